"""TEST INFRASTRUCTURE ONLY -- ctypes front end of ``oracle/_build/liboracle.so``
(gcn_oracle.c) plus numpy restatements of the counter-based generators.  Used by
tests/, ``__graft_entry__.smoke()`` and bench.py's ``cpu_baseline`` leg as the
checker; never imported by gcn_amd/.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle.so")
_lib = None

_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
_i64p = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")
_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")


def build(force=False):
    """Compile gcn_oracle.c with the committed Makefile."""
    if force or not os.path.exists(_SO) or \
            os.path.getmtime(_SO) < os.path.getmtime(os.path.join(_HERE, "gcn_oracle.c")):
        subprocess.check_call(["make", "-C", _HERE], stdout=subprocess.DEVNULL)
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        _lib.orc_masked_softmax_xent_f32.restype = C.c_float
        _lib.orc_masked_accuracy_f32.restype = C.c_float
        _lib.orc_max_threads.restype = C.c_int
    return _lib


def max_threads():
    return int(lib().orc_max_threads())


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _vp(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else C.c_void_p(0)


def spmm_coo(rows, cols, vals, b, n_rows):
    """TF CPU order: out.row(r) += v * b.row(c), nonzeros in the order given."""
    rows, cols, vals, b = _i32(rows), _i32(cols), _f32(vals), _f32(b)
    out = np.empty((n_rows, b.shape[1]), dtype=np.float32)
    lib().orc_spmm_coo_f32(C.c_int64(n_rows), C.c_int64(len(vals)), _vp(rows), _vp(cols), _vp(vals),
                           _vp(b), C.c_int64(b.shape[1]), _vp(out), C.c_int64(out.shape[1]),
                           C.c_int32(b.shape[1]))
    return out


def spmm_coo_adjoint(rows, cols, vals, g, n_cols):
    """Gradient w.r.t. the dense operand: out.row(c) += v * g.row(r)."""
    rows, cols, vals, g = _i32(rows), _i32(cols), _f32(vals), _f32(g)
    out = np.empty((n_cols, g.shape[1]), dtype=np.float32)
    lib().orc_spmm_coo_adj_f32(C.c_int64(n_cols), C.c_int64(len(vals)), _vp(rows), _vp(cols),
                               _vp(vals), _vp(g), C.c_int64(g.shape[1]), _vp(out),
                               C.c_int64(out.shape[1]), C.c_int32(g.shape[1]))
    return out


def spmm_csr(row_ptr, col_idx, val, b, omp=False):
    row_ptr = np.ascontiguousarray(row_ptr, dtype=np.int64)
    col_idx, val, b = _i32(col_idx), _f32(val), _f32(b)
    n = len(row_ptr) - 1
    out = np.empty((n, b.shape[1]), dtype=np.float32)
    fn = lib().orc_spmm_csr_f32_omp if omp else lib().orc_spmm_csr_f32
    fn(C.c_int64(n), _vp(row_ptr), _vp(col_idx), _vp(val), _vp(b), C.c_int64(b.shape[1]), _vp(out),
       C.c_int64(out.shape[1]), C.c_int32(b.shape[1]))
    return out


def spmm_csr_f64(row_ptr, col_idx, val, b):
    """(sum, sum of |terms|) accumulated in fp64: the tolerance yardstick."""
    row_ptr = np.ascontiguousarray(row_ptr, dtype=np.int64)
    col_idx, val, b = _i32(col_idx), _f32(val), _f32(b)
    n = len(row_ptr) - 1
    out = np.empty((n, b.shape[1]), dtype=np.float64)
    mag = np.empty((n, b.shape[1]), dtype=np.float64)
    lib().orc_spmm_csr_f64acc(C.c_int64(n), _vp(row_ptr), _vp(col_idx), _vp(val), _vp(b),
                              C.c_int64(b.shape[1]), _vp(out), _vp(mag), C.c_int64(out.shape[1]),
                              C.c_int32(b.shape[1]))
    return out, mag


def csr_from_coo(rows, cols, vals, shape):
    """Stable (row, col) order, duplicates kept, fp64 -> fp32 cast last.
    Returns row_ptr int64, col_idx int32, val fp32, perm int64."""
    rows, cols = _i32(rows), _i32(cols)
    vals = np.ascontiguousarray(vals)
    is64 = vals.dtype == np.float64
    if not is64:
        vals = _f32(vals)
    n_rows, n_cols = int(shape[0]), int(shape[1])
    nnz = len(vals)
    row_ptr = np.empty(n_rows + 1, dtype=np.int64)
    col_idx = np.empty(nnz, dtype=np.int32)
    val = np.empty(nnz, dtype=np.float32)
    perm = np.empty(nnz, dtype=np.int64)
    rc = lib().orc_csr_from_coo(C.c_int64(n_rows), C.c_int64(n_cols), C.c_int64(nnz), _vp(rows),
                                _vp(cols), _vp(vals), C.c_int(1 if is64 else 0), _vp(row_ptr),
                                _vp(col_idx), _vp(val), _vp(perm))
    if rc != 0:
        raise ValueError("orc_csr_from_coo failed with %d" % rc)
    return row_ptr, col_idx, val, perm


def csr_transpose(row_ptr, col_idx, val, n_cols):
    row_ptr = np.ascontiguousarray(row_ptr, dtype=np.int64)
    col_idx, val = _i32(col_idx), _f32(val)
    n_rows = len(row_ptr) - 1
    nnz = len(val)
    t_row_ptr = np.empty(n_cols + 1, dtype=np.int64)
    t_col = np.empty(nnz, dtype=np.int32)
    t_val = np.empty(nnz, dtype=np.float32)
    perm = np.empty(nnz, dtype=np.int64)
    lib().orc_csr_transpose(C.c_int64(n_rows), C.c_int64(n_cols), _vp(row_ptr), _vp(col_idx),
                            _vp(val), _vp(t_row_ptr), _vp(t_col), _vp(t_val), _vp(perm))
    return t_row_ptr, t_col, t_val, perm


def sparse_dropout(vals, mask, keep):
    vals = _f32(vals)
    mask = np.ascontiguousarray(mask, dtype=np.uint8)
    out = np.empty_like(vals)
    lib().orc_sparse_dropout_f32(C.c_int64(len(vals)), _vp(vals), _vp(mask), C.c_float(keep), _vp(out))
    return out


def gemm(a, b, trans_a=False, trans_b=False, alpha=1.0, beta=0.0, c=None):
    a, b = _f32(a), _f32(b)
    M, K = (a.shape[1], a.shape[0]) if trans_a else a.shape
    N = b.shape[0] if trans_b else b.shape[1]
    out = np.zeros((M, N), dtype=np.float32) if c is None else _f32(c).copy()
    lib().orc_gemm_f32(C.c_int(trans_a), C.c_int(trans_b), C.c_int64(M), C.c_int64(N), C.c_int64(K),
                       _vp(a), C.c_int64(a.shape[1]), _vp(b), C.c_int64(b.shape[1]), _vp(out),
                       C.c_int64(N), C.c_float(alpha), C.c_float(beta))
    return out


def gemm_f64(a, b, trans_a=False, trans_b=False):
    a, b = _f32(a), _f32(b)
    M, K = (a.shape[1], a.shape[0]) if trans_a else a.shape
    N = b.shape[0] if trans_b else b.shape[1]
    out = np.empty((M, N), dtype=np.float64)
    lib().orc_gemm_f64acc(C.c_int(trans_a), C.c_int(trans_b), C.c_int64(M), C.c_int64(N),
                          C.c_int64(K), _vp(a), C.c_int64(a.shape[1]), _vp(b),
                          C.c_int64(b.shape[1]), _vp(out), C.c_int64(N))
    return out


def masked_softmax_xent(logits, labels, mask, want_grad=True):
    logits, labels = _f32(logits), _f32(labels)
    mask = np.ascontiguousarray(mask, dtype=np.uint8)
    n, c = logits.shape
    grad = np.empty_like(logits) if want_grad else None
    loss = lib().orc_masked_softmax_xent_f32(C.c_int64(n), C.c_int32(c), _vp(logits), C.c_int64(c),
                                             _vp(labels), C.c_int64(c), _vp(mask), _vp(grad),
                                             C.c_int64(c))
    return float(loss), grad


def masked_accuracy(logits, labels, mask):
    logits, labels = _f32(logits), _f32(labels)
    mask = np.ascontiguousarray(mask, dtype=np.uint8)
    n, c = logits.shape
    return float(lib().orc_masked_accuracy_f32(C.c_int64(n), C.c_int32(c), _vp(logits),
                                               C.c_int64(c), _vp(labels), C.c_int64(c), _vp(mask)))


def adam_step(w, g, m, v, lr, t, b1=0.9, b2=0.999, eps=1e-8):
    """In place on contiguous fp32 arrays."""
    for a in (w, g, m, v):
        assert a.dtype == np.float32 and a.flags.c_contiguous
    lib().orc_adam_step_f32(C.c_int64(w.size), _vp(w), _vp(g), _vp(m), _vp(v), C.c_float(lr),
                            C.c_float(b1), C.c_float(b2), C.c_float(eps), C.c_int64(t))


def normalize_coo(rows, cols, vals, d):
    rows, cols = _i32(rows), _i32(cols)
    vals = np.ascontiguousarray(vals, dtype=np.float64)
    d = np.ascontiguousarray(d, dtype=np.float64)
    orow = np.empty_like(rows)
    ocol = np.empty_like(cols)
    oval = np.empty_like(vals)
    lib().orc_normalize_coo_f64(C.c_int64(len(vals)), _vp(rows), _vp(cols), _vp(vals), _vp(d),
                                _vp(orow), _vp(ocol), _vp(oval))
    return orow, ocol, oval


# ---------------------------------------------------------------------------
# Counter-based generators, restated in numpy (Philox4x32-10, Salmon et al.).
# ---------------------------------------------------------------------------
_M0 = np.uint64(0xD2511F53)
_M1 = np.uint64(0xCD9E8D57)
_W0 = np.uint32(0x9E3779B9)
_W1 = np.uint32(0xBB67AE85)
_LO = np.uint64(0xFFFFFFFF)


def philox4x32_10(ctr_lo, ctr_hi, key):
    """Vectorised: ctr_lo / ctr_hi uint64 arrays, key a python int.  Returns a
    [n, 4] uint32 array."""
    ctr_lo = np.asarray(ctr_lo, dtype=np.uint64)
    ctr_hi = np.broadcast_to(np.asarray(ctr_hi, dtype=np.uint64), ctr_lo.shape)
    c0 = (ctr_lo & _LO).astype(np.uint32)
    c1 = (ctr_lo >> np.uint64(32)).astype(np.uint32)
    c2 = (ctr_hi & _LO).astype(np.uint32)
    c3 = (ctr_hi >> np.uint64(32)).astype(np.uint32)
    k0 = np.uint32(key & 0xFFFFFFFF)
    k1 = np.uint32((key >> 32) & 0xFFFFFFFF)
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = _M0 * c0.astype(np.uint64)
            p1 = _M1 * c2.astype(np.uint64)
            n0 = (p1 >> np.uint64(32)).astype(np.uint32) ^ c1 ^ k0
            n1 = (p1 & _LO).astype(np.uint32)
            n2 = (p0 >> np.uint64(32)).astype(np.uint32) ^ c3 ^ k1
            n3 = (p0 & _LO).astype(np.uint32)
            c0, c1, c2, c3 = n0, n1, n2, n3
            k0 = np.uint32((int(k0) + int(_W0)) & 0xFFFFFFFF)
            k1 = np.uint32((int(k1) + int(_W1)) & 0xFFFFFFFF)
    return np.stack([c0, c1, c2, c3], axis=-1)


def bernoulli_mask(n, keep, seed, offset=0):
    """byte k = (uniform(seed, offset + k) < keep), four positions per Philox
    block; uniform = (word >> 8) * 2^-24."""
    e = np.arange(offset, offset + n, dtype=np.uint64)
    r = philox4x32_10(e >> np.uint64(2), np.uint64(0), seed)
    w = r[np.arange(n), (e & np.uint64(3)).astype(np.int64)]
    u = (w >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)
    return (u < np.float32(keep)).astype(np.uint8)


def bernoulli_mask_at(index, keep, seed):
    """Mask bits at arbitrary global element indices (uint64 array)."""
    e = np.asarray(index, dtype=np.uint64).ravel()
    r = philox4x32_10(e >> np.uint64(2), np.uint64(0), seed)
    w = r[np.arange(e.size), (e & np.uint64(3)).astype(np.int64)]
    u = (w >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)
    return (u < np.float32(keep)).astype(np.uint8).reshape(np.shape(index))


def _splitmix64(x):
    x = (x + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
    z = x
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
    return x, z ^ (z >> 31)


def rmat_scramble_params(scale, seed):
    x = (seed ^ 0x6a09e667f3bcc908) & 0xFFFFFFFFFFFFFFFF
    x, m0 = _splitmix64(x)
    x, a0 = _splitmix64(x)
    x, m1 = _splitmix64(x)
    x, a1 = _splitmix64(x)
    return dict(m0=m0 | 1, a0=a0, m1=m1 | 1, a1=a1, s0=scale // 2 + 1, s1=scale // 3 + 1,
                mask=(1 << scale) - 1)


def rmat_scramble(v, p):
    v = np.asarray(v, dtype=np.uint64)
    mask = np.uint64(p["mask"])
    with np.errstate(over="ignore"):
        v = (v * np.uint64(p["m0"]) + np.uint64(p["a0"])) & mask
        v = v ^ (v >> np.uint64(p["s0"]))
        v = (v * np.uint64(p["m1"]) + np.uint64(p["a1"])) & mask
        v = v ^ (v >> np.uint64(p["s1"]))
    return v


def _thr(p):
    if p <= 0.0:
        return 0
    if p >= 1.0:
        return 0xFFFFFFFF
    return int(p * 4294967296.0)


def rmat_edges(scale, n_edges, a, b, c, seed, edge_begin=0):
    """(u, v) uint64 arrays of the scrambled endpoints of edges
    edge_begin .. edge_begin + n_edges (self-loops NOT yet removed)."""
    e = np.arange(edge_begin, edge_begin + n_edges, dtype=np.uint64)
    ta, tab, tabc = np.uint32(_thr(a)), np.uint32(_thr(a + b)), np.uint32(_thr(a + b + c))
    u = np.zeros(n_edges, dtype=np.uint64)
    v = np.zeros(n_edges, dtype=np.uint64)
    for lvl in range(0, scale, 4):
        r = philox4x32_10(e, np.uint64(lvl >> 2), seed)
        for j in range(4):
            if lvl + j < scale:
                x = r[:, j]
                q = np.where(x < ta, 0, np.where(x < tab, 1, np.where(x < tabc, 2, 3))).astype(np.uint64)
                u = (u << np.uint64(1)) | (q >> np.uint64(1))
                v = (v << np.uint64(1)) | (q & np.uint64(1))
    p = rmat_scramble_params(scale, seed)
    return rmat_scramble(u, p), rmat_scramble(v, p)


def rmat_adjacency_plus_identity(scale, edge_factor, a, b, c, seed):
    """scipy CSR (float64 ones) of A + I for the symmetrised, de-duplicated
    RMAT graph, self-loops dropped before adding I."""
    import scipy.sparse as sp
    n = 1 << scale
    u, v = rmat_edges(scale, n * edge_factor, a, b, c, seed)
    keep = u != v
    u, v = u[keep].astype(np.int64), v[keep].astype(np.int64)
    r = np.concatenate([u, v, np.arange(n, dtype=np.int64)])
    cc = np.concatenate([v, u, np.arange(n, dtype=np.int64)])
    key = np.unique(r * n + cc)
    m = sp.csr_matrix((np.ones(len(key)), (key // n, key % n)), shape=(n, n))
    m.sort_indices()
    return m
