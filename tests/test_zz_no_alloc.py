"""The SpMM entry point that allocates nothing (gcnx_spmm_csr_static_f32: partial
sums in a device array built into the library) against the entry point it is a
variant of (gcnx_spmm_csr_f32 with a caller workspace), called through the C ABI
with ctypes.  Bar: bit-exact (same kernels, same order of additions).

Two calls in a row on one GPU, the second on a smaller input, so that the
second call runs over whatever the first one left in the shared array; then the
size limit.  Runs in the last test file on purpose: it must pass alone in a
fresh process and after every other test.
"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _both(lib, torch, a, b, width, relu=False, mask=None, mask_scale=1.0, alpha=1.0, beta=0.0, c0=None):
    """(static-scratch result, caller-workspace result) of the same product."""
    from gcn_amd.sparse import _ptr, _stream
    dev = b.device
    outs = []
    for static in (True, False):
        out = torch.full((a.n_rows, width), float("nan"), dtype=torch.float32, device=dev) if c0 is None \
            else c0.clone()
        midx = a.input_index if mask is not None else None
        with torch.cuda.device(dev):
            if static:
                rc = lib.gcnx_spmm_csr_static_f32(a.struct_ref(), _ptr(a.chunk_row), _ptr(b), b.stride(0), _ptr(out),
                                                  out.stride(0), width, alpha, beta, _ptr(mask), _ptr(midx),
                                                  mask_scale, int(relu), _stream())
            else:
                need = lib.gcnx_spmm_workspace_bytes(a.struct_ref(), width)
                ws = torch.empty(max(need, 256), dtype=torch.uint8, device=dev)
                rc = lib.gcnx_spmm_csr_f32(a.struct_ref(), _ptr(a.chunk_row), _ptr(b), b.stride(0), _ptr(out),
                                           out.stride(0), width, alpha, beta, _ptr(mask), _ptr(midx), mask_scale,
                                           int(relu), _ptr(ws), ws.numel(), _stream())
        assert rc == 0, lib.gcnx_last_error()
        torch.cuda.synchronize()
        outs.append(out.cpu().numpy())
    return outs


def test_static_scratch_spmm_matches_workspace_spmm():
    import torch
    import oracle
    from gcn_amd import _lib, rmat
    from gcn_amd.sparse import Csr, _ptr, _stream
    lib = _lib.load()
    dev = torch.device("cuda:0")
    rng = np.random.RandomState(3)

    # call 1: RMAT scale 16 (about 1.5 M nonzeros, hub rows cut by many chunk
    # boundaries), width 64 -> several MiB of partial sums in the shared array
    a1 = rmat.rmat_ahat(16, 8, 123, dev)
    assert a1.nnz > (1 << 18)                      # the chunk kernels, not the row kernel
    need1 = lib.gcnx_spmm_workspace_bytes(a1.struct_ref(), 64)
    assert (1 << 20) < need1 <= _lib.SPMM_STATIC_SCRATCH_BYTES
    b1 = torch.from_numpy(rng.standard_normal((a1.n_cols, 64)).astype(np.float32)).to(dev)
    s1, w1 = _both(lib, torch, a1, b1, 64, relu=True)
    assert np.array_equal(s1.view(np.uint32), w1.view(np.uint32))
    rp, ci, v = a1.row_ptr.cpu().numpy(), a1.col_idx.cpu().numpy(), a1.val.cpu().numpy()
    ref, mag = oracle.spmm_csr_f64(rp, ci, v, b1.cpu().numpy())
    assert np.all(np.abs(s1 - np.maximum(ref, 0.0)) <= 1e-5 * mag + 1e-30)

    # call 2, right behind it: a smaller matrix (forced down the chunk kernels
    # with 64-nonzero chunks so that it uses the array too), other width, sparse
    # dropout mask, alpha / beta -- over the contents call 1 left behind
    n2, nnz2 = 3000, 40000
    rows = rng.randint(0, n2, nnz2).astype(np.int32)
    rows[:6000] = 11                                # one row far longer than a chunk
    cols = rng.randint(0, n2, nnz2).astype(np.int32)
    vals = rng.standard_normal(nnz2)
    before = Csr.default_spmm_chunk
    Csr.default_spmm_chunk = 64
    try:
        a2 = Csr.from_coo(rows, cols, vals, (n2, n2), dev)
    finally:
        Csr.default_spmm_chunk = before
    assert 0 < lib.gcnx_spmm_workspace_bytes(a2.struct_ref(), 16) < need1
    b2 = torch.from_numpy(rng.standard_normal((n2, 16)).astype(np.float32)).to(dev)
    mask = torch.from_numpy((rng.random_sample(nnz2) < 0.5).astype(np.uint8)).to(dev)
    c0 = torch.from_numpy(rng.standard_normal((n2, 16)).astype(np.float32)).to(dev)
    s2, w2 = _both(lib, torch, a2, b2, 16, mask=mask, mask_scale=2.0, alpha=0.5, beta=-1.5, c0=c0)
    assert np.array_equal(s2.view(np.uint32), w2.view(np.uint32))
    rp, ci, v = a2.row_ptr.cpu().numpy(), a2.col_idx.cpu().numpy(), a2.val.cpu().numpy()
    perm = a2.input_index.cpu().numpy()
    vd = oracle.sparse_dropout(v, mask.cpu().numpy()[perm], 0.5)
    ref, mag = oracle.spmm_csr_f64(rp, ci, vd, b2.cpu().numpy())
    c0n = c0.cpu().numpy().astype(np.float64)
    assert np.all(np.abs(s2 - (0.5 * ref - 1.5 * c0n)) <= 1e-5 * (0.5 * mag + 1.5 * np.abs(c0n)) + 1e-30)

    # and once more the first product: same bits as before
    s3, _ = _both(lib, torch, a1, b1, 64, relu=True)
    assert np.array_equal(s3.view(np.uint32), s1.view(np.uint32))

    # beyond the array: a clear error, nothing launched
    wide = 1024
    assert lib.gcnx_spmm_workspace_bytes(a1.struct_ref(), wide) > _lib.SPMM_STATIC_SCRATCH_BYTES
    bw = torch.zeros((a1.n_cols, wide), dtype=torch.float32, device=dev)
    cw = torch.zeros((a1.n_rows, wide), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        rc = lib.gcnx_spmm_csr_static_f32(a1.struct_ref(), _ptr(a1.chunk_row), _ptr(bw), wide, _ptr(cw), wide, wide,
                                          1.0, 0.0, None, None, 1.0, 0, _stream())
    assert rc == _lib.ERR_WORKSPACE
    assert b"built-in scratch array" in lib.gcnx_last_error()


def test_static_scratch_spmm_allocates_nothing_and_records_into_a_graph():
    """The claim itself: across a call of gcnx_spmm_csr_static_f32 neither torch's
    allocator nor the device's free memory (hipMemGetInfo) moves, and the call can
    be recorded in a HIP graph (stream capture forbids allocation) and replayed:
    the replay writes the bits of the eager call."""
    import torch
    from gcn_amd import _lib, rmat
    from gcn_amd.sparse import _ptr
    lib = _lib.load()
    dev = torch.device("cuda:0")
    a = rmat.rmat_ahat(15, 8, 123, dev)            # builds its chunk table: the array's address is resolved here
    assert a.nnz > (1 << 18)
    width = 32
    assert 0 < lib.gcnx_spmm_workspace_bytes(a.struct_ref(), width) <= _lib.SPMM_STATIC_SCRATCH_BYTES
    g = torch.Generator(device=dev)
    g.manual_seed(9)
    b = torch.rand((a.n_cols, width), device=dev, generator=g) - 0.5
    out = torch.empty((a.n_rows, width), dtype=torch.float32, device=dev)
    ref = torch.empty_like(out)

    def call(dst, stream):
        with torch.cuda.device(dev):
            rc = lib.gcnx_spmm_csr_static_f32(a.struct_ref(), _ptr(a.chunk_row), _ptr(b), b.stride(0), _ptr(dst),
                                              dst.stride(0), width, 1.0, 0.0, None, None, 1.0, 1,
                                              C.c_void_p(stream.cuda_stream))
        assert rc == 0, lib.gcnx_last_error()

    call(ref, torch.cuda.current_stream())          # (a first call, so that nothing below is anybody's first)
    torch.cuda.synchronize()
    alloc0, reserved0 = torch.cuda.memory_allocated(dev), torch.cuda.memory_reserved(dev)
    free0, _ = torch.cuda.mem_get_info(dev)
    call(out, torch.cuda.current_stream())
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info(dev)
    assert torch.cuda.memory_allocated(dev) == alloc0 and torch.cuda.memory_reserved(dev) == reserved0
    assert free1 == free0, (free0, free1)
    assert torch.equal(out, ref)
    # recorded and replayed
    out.fill_(float("nan"))
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        call(out, torch.cuda.current_stream())      # the capturing stream
    torch.cuda.synchronize()
    assert bool(torch.isnan(out).all())             # recording launches nothing
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, ref)
    b.mul_(2.0)                                     # the graph reads the operand where it lies
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, ref * 2.0)
