"""The data-path collectives of the row-partitioned engine, issued through the
C ABI (``gcnx_halo_allgather_f32``, ``gcnx_comm_allreduce_sum_f32``: RCCL over
xGMI, bound inside ``libgcnhip.so``) on a communication stream of their own.

torch.distributed is used here for ONE thing: handing rank 0's 128-byte RCCL id
to the other ranks when the communicator is created.  No tensor of the training
step travels through it; a host without Python creates the communicator from
the same id with ``gcnx_comm_create`` and calls the same entry points.
"""
import ctypes as C

import torch

from . import _lib
from .sparse import _ptr


class _Pending(object):
    """Handle of an exchange in flight on the communication stream; ``wait()``
    orders the caller's current stream behind it (no host synchronisation)."""

    def __init__(self, event):
        self._event = event

    def wait(self):
        torch.cuda.current_stream().wait_event(self._event)


class RcclComm(object):
    def __init__(self, group, world, rank, device):
        """Collective over ``group`` (a torch.distributed group that is only used
        to broadcast the id).  Binds the communicator to ``device``."""
        import torch.distributed as dist
        lib = _lib.load()
        self.world, self.rank = int(world), int(rank)
        self.device = torch.device(device)
        self._comm = None
        # Every rank walks through the same collectives whatever fails locally, and
        # all of them learn the outcome before the communicator is created: a rank
        # that raised early would leave the others waiting in the broadcast or in
        # ncclCommInitRank.
        ok, why = 1, ""
        if not lib.gcnx_comm_available():
            ok, why = 0, "RCCL could not be loaded by libgcnhip.so on rank %d" % self.rank
        ident = (C.c_char * _lib.COMM_ID_BYTES)()
        with torch.cuda.device(self.device):
            if self.rank == 0 and ok:
                rc = lib.gcnx_comm_unique_id(ident)
                if rc != 0:
                    ok, why = 0, "gcnx_comm_unique_id failed on rank 0: %s" % (lib.gcnx_last_error() or b"").decode()
            box = [(ok, why, bytes(ident.raw)) if self.rank == 0 else None]
            dist.broadcast_object_list(box, src=0, group=group)
            if not box[0][0]:
                ok, why = 0, why or box[0][1]
            backend = str(dist.get_backend(group))
            agree = torch.tensor([ok], dtype=torch.int32, device=self.device if "nccl" in backend else "cpu")
            dist.all_reduce(agree, op=dist.ReduceOp.MIN, group=group)
            if int(agree.item()) == 0:
                raise RuntimeError(why or "a peer cannot create the RCCL communicator; no multi-GPU data path")
            ident = (C.c_char * _lib.COMM_ID_BYTES).from_buffer_copy(box[0][2])
            handle = C.c_void_p(0)
            _lib.check(lib.gcnx_comm_create(self.world, self.rank, ident, C.byref(handle)))
        self._comm = handle
        self.stream = torch.cuda.Stream(device=self.device)
        self._plans = {}

    def close(self):
        if self._comm is not None and self._comm.value:
            torch.cuda.synchronize(self.device)
            _lib.load().gcnx_comm_destroy(self._comm)
        self._comm = None

    def self_check(self):
        """An all-reduce and a 2-rows-per-rank exchange with known answers through
        the same entry points the training step uses (collective: every rank of
        the communicator calls it).  Returns (ok, reason)."""
        dev, world, rank = self.device, self.world, self.rank
        with torch.cuda.device(dev):
            probe = torch.full((8,), float(rank + 1), device=dev)
            self.allreduce_sum_(probe)
            counts = [2] * world
            send = torch.cat([torch.full((2, 4), float(rank * 1000 + p), device=dev) for p in range(world)])
            recv = torch.zeros((2 * world, 4), device=dev)
            self.exchange_rows(recv, send, counts, counts).wait()
            torch.cuda.synchronize(dev)
        want = torch.cat([torch.full((2, 4), float(p * 1000 + rank)) for p in range(world)])
        if float(probe[0]) != world * (world + 1) / 2:
            return False, "all-reduce returned %r, expected %r" % (float(probe[0]), world * (world + 1) / 2)
        if not torch.equal(recv.cpu(), want):
            return False, "row exchange delivered wrong rows"
        return True, ""

    # ------------------------------------------------------------------ plans
    def _plan_struct(self, send_counts, recv_counts, send_idx):
        """ctypes mirror of ``gcnx_halo_exchange`` (cached per count lists /
        index tensor; the host arrays must outlive the call)."""
        key = (tuple(send_counts), tuple(recv_counts), 0 if send_idx is None else send_idx.data_ptr())
        hit = self._plans.get(key)
        if hit is None:
            n = self.world
            hs = (C.c_int64 * n)(*[int(v) for v in send_counts])
            hr = (C.c_int64 * n)(*[int(v) for v in recv_counts])
            st = _lib.HaloExchangeStruct(self.world, self.rank, hs, hr,
                                         send_idx.data_ptr() if send_idx is not None else None)
            hit = (st, hs, hr, send_idx)
            self._plans[key] = hit
        return hit[0]

    # --------------------------------------------------------------- exchange
    def exchange_rows(self, recv, send_buf, recv_counts, send_counts, src=None, send_idx=None, col0=0):
        """Start the variable-size all-to-all of rows on the communication
        stream and return a handle whose ``wait()`` orders the current stream
        behind it.  ``recv`` [sum(recv_counts), w] and ``send_buf``
        [sum(send_counts), w] are contiguous 32-bit matrices (fp32 rows or int32
        sign words).  With ``src`` / ``send_idx`` the rows
        ``src[send_idx, col0:col0+w]`` are packed into ``send_buf`` first, on the
        communication stream; otherwise ``send_buf`` already holds them."""
        lib = _lib.load()
        w = int(recv.shape[1]) if recv.dim() == 2 else 1
        for t, name in ((recv, "recv"), (send_buf, "send_buf")):
            if t.element_size() != 4 or not t.is_contiguous() or t.device != self.device:
                raise ValueError("%s must be a contiguous 32-bit tensor on the communicator's device" % name)
        if len(recv_counts) != self.world or len(send_counts) != self.world:
            raise ValueError("one send count and one receive count per rank of the communicator")
        if recv.shape[0] != sum(recv_counts) or send_buf.shape[0] != sum(send_counts):
            raise ValueError("buffer rows do not match the exchange counts")
        if send_buf.numel() and (send_buf.dim() == 2 and int(send_buf.shape[1]) != w):
            raise ValueError("send and receive rows differ in width")
        ld_src = 0
        if src is not None:
            if send_idx is None or src.element_size() != 4 or src.stride(-1) != 1 or src.device != self.device:
                raise ValueError("src needs send_idx and unit column stride")
            if send_idx.dtype != torch.int32 or send_idx.numel() != send_buf.shape[0] or not send_idx.is_contiguous():
                raise ValueError("send_idx: one contiguous int32 per row to send")
            ld_src = src.stride(0)
            if col0 + w > src.shape[1]:
                raise ValueError("column slab outside src")
        plan = self._plan_struct(send_counts, recv_counts, send_idx if src is not None else None)
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream())        # operands (and buffer reuse) as of now
        done = torch.cuda.Event()
        with torch.cuda.device(self.device):
            self.stream.wait_event(ready)
            _lib.check(lib.gcnx_halo_allgather_f32(C.byref(plan), self._comm, _ptr(src), ld_src, int(col0), w,
                                                   _ptr(send_buf), _ptr(recv), C.c_void_p(self.stream.cuda_stream)))
            done.record(self.stream)
        return _Pending(done)

    def allreduce_sum_(self, t):
        """In-place sum over the ranks.  Issued on the communication stream like
        every other call on this communicator (one communicator, one stream: the
        order of its operations is then the issue order on every rank) and
        ordered after / before the current stream with events."""
        lib = _lib.load()
        if t.dtype != torch.float32 or not t.is_contiguous() or t.device != self.device:
            raise ValueError("allreduce: contiguous float32 tensor on the communicator's device")
        cur = torch.cuda.current_stream()
        ready, done = torch.cuda.Event(), torch.cuda.Event()
        ready.record(cur)
        with torch.cuda.device(self.device):
            self.stream.wait_event(ready)
            _lib.check(lib.gcnx_comm_allreduce_sum_f32(self._comm, _ptr(t), t.numel(),
                                                       C.c_void_p(self.stream.cuda_stream)))
            done.record(self.stream)
        cur.wait_event(done)
        return t
