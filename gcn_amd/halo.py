"""Halo plan of a 1-D row partition: which rows of the dense operand a shard
must fetch from which peer, and which of its own rows each peer asks for.

Rank p owns a contiguous block of rows of A_hat and of every activation (blocks
ascending with the rank; equal sizes are not required).  Its
CSR block references some columns outside that range; only those rows (the
halo) are exchanged before a product, with one variable-size all-to-all (RCCL
over xGMI: every pair of GPUs exchanges its own piece concurrently on its own
link).  The CSR block is rewritten over [local rows | halo rows] so that the
SpMM reads the local activation in place and the received rows from the halo
buffer (``ops.spmm(..., b2=halo)``).  For the scrambled RMAT graphs about 18 %
(8 GPUs) to 34 % (2 GPUs) of the remote rows are referenced at all, which is
what makes this several times cheaper than an all-gather.
"""
import ctypes as C

import torch

from . import _lib
from .sparse import Csr, _ptr, _stream


def device_halo_plan(a_local, row_begin, n_loc):
    """HIP side of the plan: returns (block rewritten over [local | halo]
    columns, sorted int32 tensor of the global rows to fetch)."""
    lib = _lib.load()
    dev = a_local.device
    n_glob = a_local.n_cols
    col_new = torch.empty(a_local.nnz, dtype=torch.int32, device=dev)
    cap = min(n_glob - n_loc, a_local.nnz) + 1
    need = torch.empty(cap, dtype=torch.int32, device=dev)
    n_need = C.c_int64(0)
    with torch.cuda.device(dev):
        wsb = lib.gcnx_halo_plan_workspace_bytes(n_glob)
        ws = torch.empty(max(wsb, 256), dtype=torch.uint8, device=dev)
        _lib.check(lib.gcnx_halo_plan(a_local.struct_ref(), row_begin, n_loc, _ptr(col_new),
                                      _ptr(need), cap, C.byref(n_need), _ptr(ws), ws.numel(),
                                      _stream()))
    del ws
    n_halo = int(n_need.value)
    need = need[:n_halo].clone()
    csr = Csr(a_local.n_rows, int(n_loc) + n_halo, a_local.row_ptr, col_new, a_local.val,
              a_local.input_index)
    return csr, need


class HaloPlan(object):
    def __init__(self, a_local, row_begin, n_loc, world, rank, all_to_all, plan_fn=device_halo_plan):
        """``a_local``: the local rows with GLOBAL column ids.
        ``all_to_all(recv, send, recv_counts, send_counts)`` performs a blocking
        variable-size all-to-all along dim 0 (counts are Python lists); it is only
        used here, at set-up, for the row ranges and the request lists.  Ranks may
        own different numbers of rows (N need not divide by the rank count) as long
        as the ranges are contiguous and ascending with the rank."""
        self.world, self.rank, self.n_loc = world, rank, int(n_loc)
        self.csr, need = plan_fn(a_local, row_begin, n_loc)
        self.need = need                     # global ids of the halo rows, ascending (int32)
        dev = need.device
        self.n_halo = int(need.numel())
        # every rank's row range, gathered with the one primitive at hand
        mine = torch.tensor([[int(row_begin), int(n_loc)]] * world, dtype=torch.int64, device=dev)
        ranges = torch.empty_like(mine)
        all_to_all(ranges, mine, [1] * world, [1] * world)
        ranges = ranges.cpu()
        begins = [int(v) for v in ranges[:, 0].tolist()]
        ends = [b + int(n) for b, n in zip(begins, ranges[:, 1].tolist())]
        if begins[rank] != int(row_begin) or any(begins[p + 1] != ends[p] for p in range(world - 1)):
            raise RuntimeError("halo plan: the ranks' row ranges are not contiguous and ascending: %s" %
                               (list(zip(begins, ends)),))
        self.row_ranges = list(zip(begins, ends))
        # need[] is sorted by global id, hence grouped by owner
        edges = torch.tensor(begins + [ends[-1]], dtype=torch.int64, device=dev)
        bounds = torch.searchsorted(need.to(torch.int64), edges).cpu().tolist()
        self.recv_counts = [bounds[p + 1] - bounds[p] for p in range(world)]
        if self.recv_counts[rank] != 0 or sum(self.recv_counts) != self.n_halo:
            raise RuntimeError("halo plan: a referenced row lies outside every rank's range")
        # tell every owner how many / which of its rows this rank needs
        cnt_send = torch.tensor(self.recv_counts, dtype=torch.int64, device=dev)
        cnt_recv = torch.empty(world, dtype=torch.int64, device=dev)
        all_to_all(cnt_recv, cnt_send, [1] * world, [1] * world)
        self.send_counts = [int(v) for v in cnt_recv.cpu().tolist()]
        self.n_send = sum(self.send_counts)
        asked = torch.empty(self.n_send, dtype=torch.int32, device=dev)
        all_to_all(asked, need, self.send_counts, self.recv_counts)
        self.send_idx = (asked - int(row_begin)).contiguous()   # local row of every row to send
        if self.n_send:
            lo, hi = int(self.send_idx.min()), int(self.send_idx.max())
            if lo < 0 or hi >= n_loc:
                raise RuntimeError("halo plan: a peer asked for a row this rank does not own")
