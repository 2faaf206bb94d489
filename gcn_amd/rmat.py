"""Synthetic RMAT graphs, generated and normalised on the device.

Pipeline (all through the C ABI): Philox-keyed edge list -> both directions as
64-bit keys for the local row range -> sort + de-duplicate + diagonal (A + I) ->
CSR -> A_hat = D^-1/2 (A + I) D^-1/2 with the reference's fp64 rounding
(gcn/utils.py:122-135).  The degree table ``k ** -0.5`` is computed by numpy on
the host for the distinct integer degrees, exactly as the reference computes it,
so the normalised values carry the same bits as the host path.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from .sparse import Csr, _ptr, _stream

GRAPH500_ABC = (0.57, 0.19, 0.19)


def rmat_keys(scale, edge_factor, seed, device, row_begin=0, row_end=None, abc=GRAPH500_ABC,
              edge_batch=1 << 27):
    """uint64 keys (row << 32 | col), both directions, rows in
    [row_begin, row_end); unsorted, with duplicates.  Returns (keys tensor with
    ``row_end - row_begin`` spare slots at the end, number of keys)."""
    lib = _lib.load()
    dev = torch.device(device)
    n = 1 << scale
    row_end = n if row_end is None else row_end
    n_edges = n * edge_factor
    frac = (row_end - row_begin) / float(n)
    # expected keys: 2 * n_edges * frac; hubs make the share uneven, so leave room
    capacity = int(2 * n_edges * min(1.0, frac * 1.5 + 0.01)) + 1024
    keys = torch.empty(capacity + (row_end - row_begin), dtype=torch.int64, device=dev)
    ws = torch.empty(lib.gcnx_rmat_workspace_bytes(), dtype=torch.uint8, device=dev)
    total = 0
    cnt = C.c_int64(0)
    with torch.cuda.device(dev):
        for e0 in range(0, n_edges, edge_batch):
            e1 = min(n_edges, e0 + edge_batch)
            _lib.check(lib.gcnx_rmat_edges(
                scale, e0, e1, abc[0], abc[1], abc[2], int(seed), row_begin, row_end,
                C.c_void_p(keys.data_ptr() + 8 * total), capacity - total, C.byref(cnt), _ptr(ws),
                _stream()))
            total += cnt.value
    return keys, total


def csr_from_keys(keys, n_keys, row_begin, row_end, n_cols, add_identity=True):
    """Sorted, de-duplicated CSR (values 1.0) over local rows."""
    lib = _lib.load()
    dev = keys.device
    n_local = row_end - row_begin
    cap = n_keys + (n_local if add_identity else 0)
    row_ptr = torch.empty(n_local + 1, dtype=torch.int64, device=dev)
    col_idx = torch.empty(cap, dtype=torch.int32, device=dev)
    val = torch.empty(cap, dtype=torch.float32, device=dev)
    nnz = C.c_int64(0)
    with torch.cuda.device(dev):
        need = lib.gcnx_csr_from_keys_workspace_bytes(n_keys, n_local)
        ws = torch.empty(max(need, 256), dtype=torch.uint8, device=dev)
        _lib.check(lib.gcnx_csr_from_keys(_ptr(keys), n_keys, row_begin, row_end,
                                          1 if add_identity else 0, _ptr(row_ptr), _ptr(col_idx),
                                          _ptr(val), C.byref(nnz), _ptr(ws), ws.numel(), _stream()))
    del ws
    nnz = nnz.value
    # shrink to the de-duplicated size (a copy; the oversized buffers are freed)
    col_idx = col_idx[:nnz].clone()
    val = val[:nnz].clone()
    return row_ptr, col_idx, val, nnz


def degree_table(max_degree):
    """``numpy.power(k, -0.5)`` for k = 0..max_degree in fp64, with the
    reference's inf -> 0 rule for k = 0 (gcn/utils.py:126-127)."""
    k = np.arange(max_degree + 1, dtype=np.float64)
    with np.errstate(divide="ignore"):
        d = np.power(k, -0.5)
    d[np.isinf(d)] = 0.0
    return d


def normalize_csr(row_ptr, col_idx, val, n_cols, row_begin, row_sum_global):
    """In-place-safe degree normalisation of local rows; returns the new values."""
    lib = _lib.load()
    dev = col_idx.device
    n_local = row_ptr.numel() - 1
    max_deg = int(row_sum_global.max().item())
    table = torch.from_numpy(degree_table(max_deg)).to(dev)
    st = _lib.CsrStruct(n_local, n_cols, col_idx.numel(), row_ptr.data_ptr(), col_idx.data_ptr(),
                        val.data_ptr())
    out = torch.empty_like(val)
    flag = torch.zeros(1, dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(lib.gcnx_normalize_csr_f64(C.byref(st), row_begin, _ptr(row_sum_global),
                                              _ptr(table), table.numel(), _ptr(out), _ptr(flag),
                                              _stream()))
    if int(flag.item()) != 0:
        raise RuntimeError("normalize_csr: a row sum fell outside the degree table")
    return out


def rmat_ahat(scale, edge_factor=8, seed=123, device="cuda", row_begin=0, row_end=None,
              gather_degrees=None, abc=GRAPH500_ABC):
    """Rows [row_begin, row_end) of A_hat = D^-1/2 (A + I) D^-1/2 for the RMAT
    graph (2^scale vertices, ``edge_factor`` * 2^scale undirected edges before
    de-duplication) as a :class:`Csr` with GLOBAL column ids.
    ``gather_degrees(local_int32)`` must return the row sums of all vertices
    when only a row range is built (multi-GPU); it defaults to identity."""
    n = 1 << scale
    row_end = n if row_end is None else row_end
    keys, n_keys = rmat_keys(scale, edge_factor, seed, device, row_begin, row_end, abc)
    row_ptr, col_idx, val, nnz = csr_from_keys(keys, n_keys, row_begin, row_end, n)
    del keys
    local_deg = (row_ptr[1:] - row_ptr[:-1]).to(torch.int32)
    if gather_degrees is None:
        if row_begin != 0 or row_end != n:
            raise ValueError("a row range needs gather_degrees")
        deg = local_deg
    else:
        deg = gather_degrees(local_deg)
    val = normalize_csr(row_ptr, col_idx, val, n, row_begin, deg)
    return Csr(row_end - row_begin, n, row_ptr, col_idx, val, None)
