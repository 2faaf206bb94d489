"""GPU parity tests of the HIP kernels, called through the C ABI (ctypes) and
checked against the CPU oracle on the same seeded inputs.

Bars: CSR construction / transpose / masks / RMAT keys are bit-exact; fp32
products are within 1e-5 relative to the size of the summed terms (helpers.py).
"""
import numpy as np
import pytest

from helpers import RTOL, assert_close_to_f64, random_coo, rel_maxnorm

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import torch
    import oracle
    from gcn_amd import ops, sparse
    assert torch.cuda.is_available()
    oracle.build()
    return dict(torch=torch, oracle=oracle, ops=ops, sparse=sparse, dev=torch.device("cuda:0"))


@pytest.fixture(autouse=True, params=["default", "chunk64", "chunk512"])
def spmm_path(request):
    """Every test runs three times: with the default dispatch (wavefront-per-row
    kernel for the small matrices, chunk size by matrix size), with the row kernel
    switched off (chunk kernels with 64-nonzero chunks: many rows cut by chunk
    boundaries, the fix-up pass does most stores) and with 512-nonzero chunks
    (the configuration the large graphs run: several 64-wide batches per
    wavefront)."""
    from gcn_amd.sparse import Csr
    before = Csr.default_spmm_chunk
    if request.param != "default":
        # the dispatch hint is a field of the matrix descriptor (gcnx_csr::spmm_chunk)
        Csr.default_spmm_chunk = 64 if request.param == "chunk64" else 512
    yield request.param
    Csr.default_spmm_chunk = before


def _csr_np(csr):
    return (csr.row_ptr.cpu().numpy(), csr.col_idx.cpu().numpy(), csr.val.cpu().numpy(),
            csr.input_index.cpu().numpy())


# ------------------------------------------------------------------ CSR build
@pytest.mark.parametrize("n_rows,n_cols,nnz,f64", [
    (1, 1, 1, True), (5, 3, 0, True), (64, 64, 300, True), (1000, 777, 20000, False),
    (3000, 3000, 200000, True), (10, 100000, 5000, True),
])
def test_csr_from_coo_bit_exact(env, n_rows, n_cols, nnz, f64):
    rng = np.random.RandomState(n_rows * 7 + nnz)
    rows, cols, vals = random_coo(rng, n_rows, n_cols, nnz, dup_frac=0.1)
    if not f64:
        vals = vals.astype(np.float32)
    ref = env["oracle"].csr_from_coo(rows, cols, vals, (n_rows, n_cols))
    got = _csr_np(env["sparse"].Csr.from_coo(rows, cols, vals, (n_rows, n_cols), env["dev"]))
    for g, r, name in zip(got, ref, ("row_ptr", "col_idx", "val", "perm")):
        assert g.dtype == r.dtype, name
        assert np.array_equal(g.view(np.uint8), r.view(np.uint8)), name


def test_csr_from_coo_rejects_out_of_range(env):
    from gcn_amd._lib import GcnxError, ERR_INDEX_RANGE
    with pytest.raises(GcnxError) as ei:
        env["sparse"].Csr.from_coo([0, 5], [0, 1], [1.0, 2.0], (3, 3), env["dev"])
    assert ei.value.code == ERR_INDEX_RANGE


@pytest.mark.parametrize("ds", ["cora", "citeseer"])
def test_csr_of_golden_ahat(env, ds, golden_dir):
    """The reference's own preprocess_adj output -> device CSR: same entries as
    the canonical (row, col) order of the tuple, values rounded to fp32."""
    import hashlib
    import json
    import os
    z = np.load(os.path.join(golden_dir, "ahat_%s.npz" % ds))
    known = json.load(open(os.path.join(golden_dir, "generated_answers.json")))[ds]
    csr = env["sparse"].Csr.from_tuple((z["coords"], z["values"], tuple(z["shape"])), env["dev"])
    row_ptr, col_idx, val, _ = _csr_np(csr)
    rows = np.repeat(np.arange(csr.n_rows, dtype=np.int32), np.diff(row_ptr))
    coords = np.stack([rows, col_idx], axis=1).astype(np.int32)
    assert hashlib.sha256(np.ascontiguousarray(coords).tobytes()).hexdigest() == known["ahat_coords_sha_sorted"]
    assert hashlib.sha256(val.astype("<f4").tobytes()).hexdigest() == known["ahat_vals_f32_sha_sorted"]


@pytest.mark.parametrize("n_rows,n_cols,nnz", [(1, 1, 1), (7, 5, 0), (500, 300, 9000), (4000, 4000, 150000)])
def test_transpose_bit_exact(env, n_rows, n_cols, nnz):
    rng = np.random.RandomState(nnz + 3)
    rows, cols, vals = random_coo(rng, n_rows, n_cols, nnz)
    a = env["sparse"].Csr.from_coo(rows, cols, vals, (n_rows, n_cols), env["dev"])
    rp, ci, v, perm = _csr_np(a)
    t_ref = env["oracle"].csr_transpose(rp, ci, v, n_cols)
    t = a.transpose()
    trp, tci, tv, tidx = _csr_np(t)
    assert np.array_equal(trp, t_ref[0])
    assert np.array_equal(tci, t_ref[1])
    assert np.array_equal(tv.view(np.uint32), t_ref[2].view(np.uint32))
    # input_index of the transpose = original tuple position of each slot
    assert np.array_equal(tidx, perm[t_ref[3]])
    assert t.transpose() is a


# ------------------------------------------------------------------------ SpMM
def _spmm_case(env, n_rows, n_cols, nnz, width, hub=(), empty=(), seed=0, **kw):
    torch = env["torch"]
    rng = np.random.RandomState(seed + width + nnz)
    rows, cols, vals = random_coo(rng, n_rows, n_cols, nnz, hub_rows=hub, empty_rows=empty)
    a = env["sparse"].Csr.from_coo(rows, cols, vals, (n_rows, n_cols), env["dev"])
    b = rng.standard_normal((n_cols, width)).astype(np.float32)
    rp, ci, v, perm = _csr_np(a)
    return a, b, rp, ci, v, perm, rng


@pytest.mark.parametrize("width", [1, 2, 3, 4, 7, 8, 12, 16, 20, 31, 32, 33, 60, 64, 100, 128, 132, 256, 260, 512])
@pytest.mark.parametrize("shape", [
    (1, 1, 1), (50, 40, 0), (300, 200, 511), (300, 200, 512), (300, 200, 513),
    (2000, 1500, 30000), (20000, 20000, 90000),
])
def test_spmm_matches_oracle(env, shape, width):
    n_rows, n_cols, nnz = shape
    a, b, rp, ci, v, _, _ = _spmm_case(env, n_rows, n_cols, nnz, width)
    got = env["ops"].spmm(a, env["torch"].from_numpy(b).to(env["dev"])).cpu().numpy()
    ref64, mag = env["oracle"].spmm_csr_f64(rp, ci, v, b)
    assert_close_to_f64(got, ref64, mag, what="spmm %s w=%d" % (shape, width))
    ref32 = env["oracle"].spmm_csr(rp, ci, v, b)
    assert rel_maxnorm(got, ref32) <= RTOL


@pytest.mark.parametrize("width", [1, 7, 16, 64, 128, 256])
def test_spmm_hub_and_empty_rows(env, width):
    """Rows far longer than a chunk (completed from partial sums), runs of empty
    rows at the start, the end and across chunk boundaries."""
    n = 5000
    hub = [(17, 6000), (18, 1500), (4000, 3000)]
    empty = list(range(0, 9)) + list(range(2000, 2600)) + list(range(4990, 5000))
    a, b, rp, ci, v, _, _ = _spmm_case(env, n, n, 40000, width, hub=hub, empty=empty, seed=5)
    assert np.diff(rp).max() > 4 * 512
    got = env["ops"].spmm(a, env["torch"].from_numpy(b).to(env["dev"])).cpu().numpy()
    ref64, mag = env["oracle"].spmm_csr_f64(rp, ci, v, b)
    assert_close_to_f64(got, ref64, mag, what="hub w=%d" % width)
    assert np.all(got[np.asarray(empty)] == 0.0)


def test_spmm_vs_tf_order_coo(env):
    """Against the TF-order restatement: nonzeros walked in tuple order."""
    rng = np.random.RandomState(11)
    rows, cols, vals = random_coo(rng, 900, 800, 20000)
    vals = vals.astype(np.float32)
    b = rng.standard_normal((800, 16)).astype(np.float32)
    a = env["sparse"].Csr.from_coo(rows, cols, vals, (900, 800), env["dev"])
    got = env["ops"].spmm(a, env["torch"].from_numpy(b).to(env["dev"])).cpu().numpy()
    ref = env["oracle"].spmm_coo(rows, cols, vals, b, 900)
    assert rel_maxnorm(got, ref) <= RTOL
    # adjoint (backward) product through the exact transpose
    g = rng.standard_normal((900, 16)).astype(np.float32)
    got_t = env["ops"].spmm(a.transpose(), env["torch"].from_numpy(g).to(env["dev"])).cpu().numpy()
    ref_t = env["oracle"].spmm_coo_adjoint(rows, cols, vals, g, 800)
    assert rel_maxnorm(got_t, ref_t) <= RTOL


@pytest.mark.parametrize("width", [16, 256])
def test_spmm_dropout_mask_relu_alpha_beta(env, width):
    torch = env["torch"]
    a, b, rp, ci, v, perm, rng = _spmm_case(env, 1500, 1200, 25000, width, seed=9)
    keep = 0.5
    mask = (rng.random_sample(a.nnz) < keep).astype(np.uint8)          # tuple order
    v_drop = env["oracle"].sparse_dropout(v, mask[perm], keep)        # CSR order
    bt = torch.from_numpy(b).to(env["dev"])
    mt = torch.from_numpy(mask).to(env["dev"])
    got = env["ops"].spmm(a, bt, mask=mt, mask_scale=1.0 / keep, relu=True).cpu().numpy()
    ref64, mag = env["oracle"].spmm_csr_f64(rp, ci, v_drop, b)
    assert_close_to_f64(got, np.maximum(ref64, 0.0), mag, what="mask+relu")
    # the transposed product reuses the same tuple-order mask
    g = rng.standard_normal((1500, width)).astype(np.float32)
    t = a.transpose()
    got_t = env["ops"].spmm(t, torch.from_numpy(g).to(env["dev"]), mask=mt, mask_scale=1.0 / keep).cpu().numpy()
    trp, tci, tv, tperm = env["oracle"].csr_transpose(rp, ci, v_drop, 1200)
    ref64_t, mag_t = env["oracle"].spmm_csr_f64(trp, tci, tv, g)
    assert_close_to_f64(got_t, ref64_t, mag_t, what="mask transposed")
    # alpha / beta
    c0 = rng.standard_normal((1500, width)).astype(np.float32)
    out = torch.from_numpy(c0.copy()).to(env["dev"])
    env["ops"].spmm(a, bt, out=out, alpha=0.5, beta=-2.0)
    ref64, mag = env["oracle"].spmm_csr_f64(rp, ci, v, b)
    assert_close_to_f64(out.cpu().numpy(), 0.5 * ref64 - 2.0 * c0, 0.5 * mag + 2.0 * np.abs(c0), what="alpha/beta")


def test_spmm_reproducible_and_strided(env):
    torch = env["torch"]
    a, b, rp, ci, v, _, _ = _spmm_case(env, 3000, 3000, 60000, 64, hub=[(5, 5000)], seed=2)
    big = torch.zeros((3000, 200), device=env["dev"])
    big[:, 8:72] = torch.from_numpy(b).to(env["dev"])
    bview = big[:, 8:72]                      # ldb = 200, 32-byte aligned start
    o1 = env["ops"].spmm(a, bview).cpu().numpy()
    o2 = env["ops"].spmm(a, bview).cpu().numpy()
    assert np.array_equal(o1.view(np.uint32), o2.view(np.uint32))
    o3 = env["ops"].spmm(a, torch.from_numpy(b).to(env["dev"])).cpu().numpy()
    assert np.array_equal(o1.view(np.uint32), o3.view(np.uint32))
    bodd = big[:, 9:73]                       # misaligned start -> scalar path
    bodd.copy_(torch.from_numpy(b).to(env["dev"]))
    o4 = env["ops"].spmm(a, bodd).cpu().numpy()
    ref64, mag = env["oracle"].spmm_csr_f64(rp, ci, v, b)
    assert_close_to_f64(o4, ref64, mag, what="misaligned")


# ------------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("ta,tb,M,N,K", [
    (0, 0, 1, 1, 1), (0, 0, 2708, 7, 16), (0, 0, 1000, 16, 1433), (0, 0, 4096, 256, 256),
    (0, 0, 700, 128, 64), (0, 0, 513, 100, 37), (0, 1, 2708, 16, 7), (0, 1, 3000, 256, 16),
    (1, 0, 16, 7, 2708), (1, 0, 256, 16, 100000), (1, 0, 256, 256, 20000), (1, 1, 33, 21, 65),
    (1, 0, 128, 7, 5000), (1, 0, 384, 128, 9999), (1, 0, 256, 32, 4100),
    (0, 0, 5000, 16, 256), (0, 0, 5001, 7, 64), (0, 1, 5000, 256, 16), (0, 1, 777, 128, 64), (0, 1, 4096, 20, 128),
    (0, 1, 5003, 260, 7), (0, 0, 2049, 512, 32), (0, 0, 1500, 64, 3), (0, 1, 9000, 1024, 20),
    # small K on the matrix pipe (K <= 16, N % 32 == 0), ragged rows, K not a multiple of 4, two slabs
    (0, 1, 5003, 96, 12), (0, 0, 4099, 512, 16), (0, 0, 1025, 64, 5),
    # skinny outputs: N <= 16 streaming kernels (plain and transposed A)
    (0, 0, 4099, 16, 128), (0, 0, 1024, 3, 256), (0, 0, 70001, 16, 256), (1, 0, 260, 16, 40000),
    (1, 0, 64, 1, 4096), (1, 0, 256, 6, 123457),
])
def test_gemm_matches_oracle(env, ta, tb, M, N, K):
    torch = env["torch"]
    rng = np.random.RandomState(M + N + K)
    a = rng.standard_normal((K, M) if ta else (M, K)).astype(np.float32)
    b = rng.standard_normal((N, K) if tb else (K, N)).astype(np.float32)
    got = env["ops"].gemm(torch.from_numpy(a).to(env["dev"]), torch.from_numpy(b).to(env["dev"]),
                          trans_a=bool(ta), trans_b=bool(tb)).cpu().numpy()
    ref64 = env["oracle"].gemm_f64(a, b, bool(ta), bool(tb))
    mag = env["oracle"].gemm_f64(np.abs(a), np.abs(b), bool(ta), bool(tb))
    assert_close_to_f64(got, ref64, mag, what="gemm %s" % ((ta, tb, M, N, K),))


def test_gemm_mfma_layout_asymmetric(env):
    """A = I against an asymmetric B catches a swapped accumulator row/col map."""
    torch = env["torch"]
    M = K = 512
    N = 256
    a = np.eye(M, K, dtype=np.float32)
    b = (np.arange(K * N, dtype=np.float32).reshape(K, N) % 1021) - 400.0
    got = env["ops"].gemm(torch.from_numpy(a).to(env["dev"]), torch.from_numpy(b).to(env["dev"])).cpu().numpy()
    assert np.array_equal(got, b)
    got = env["ops"].gemm(torch.from_numpy(a).to(env["dev"]), torch.from_numpy(b).to(env["dev"]),
                          alpha=2.0, relu=True).cpu().numpy()
    assert np.array_equal(got, np.maximum(2.0 * b, 0.0))


# ----------------------------------------------------------------- glue kernels
@pytest.mark.parametrize("n,c", [(1, 2), (2708, 7), (3327, 6), (100000, 16), (70000, 3),
                                 # class counts that are multiples of 4: four classes per lane
                                 (5000, 4), (5001, 8), (4097, 12), (3000, 32), (2000, 48), (999, 64), (300, 100)])
def test_masked_softmax_xent(env, n, c):
    torch = env["torch"]
    rng = np.random.RandomState(n + c)
    logits = (rng.standard_normal((n, c)) * 3).astype(np.float32)
    labels = np.eye(c, dtype=np.float32)[rng.randint(0, c, size=n)]
    mask = (rng.random_sample(n) < 0.3).astype(np.uint8)
    mask[0] = 1
    out, grad = env["ops"].masked_softmax_xent(torch.from_numpy(logits).to(env["dev"]),
                                               torch.from_numpy(labels).to(env["dev"]),
                                               torch.from_numpy(mask).to(env["dev"]))
    out = out.cpu().numpy()
    loss_ref, grad_ref = env["oracle"].masked_softmax_xent(logits, labels, mask)
    acc_ref = env["oracle"].masked_accuracy(logits, labels, mask)
    assert abs(out[0] - loss_ref) <= 1e-5 * abs(loss_ref) + 1e-7
    assert abs(out[1] - acc_ref) <= 1e-6
    g = grad.cpu().numpy()
    assert np.abs(g - grad_ref).max() <= 1e-5 * np.abs(grad_ref).max() + 1e-12
    assert np.all(g[mask == 0] == 0.0)


def test_bernoulli_mask_bit_exact(env):
    for n, keep, seed, off in [(1, 0.5, 1, 0), (1003, 0.5, 123, 0), (4097, 0.9, 7, 13), (50000, 0.25, 2 ** 40 + 5, 10 ** 12 + 3)]:
        got = env["ops"].bernoulli_mask(n, keep, seed, off, env["dev"]).cpu().numpy()
        ref = env["oracle"].bernoulli_mask(n, keep, seed, off)
        assert np.array_equal(got, ref), (n, keep, seed, off)
    m = env["ops"].bernoulli_mask(1 << 20, 0.5, 99, 0, env["dev"]).float().mean().item()
    assert abs(m - 0.5) < 5e-3


def test_dropout_relu_bwd_adam(env):
    torch = env["torch"]
    rng = np.random.RandomState(4)
    for n in (1, 5, 1027, 100003):
        x = rng.standard_normal(n).astype(np.float32)
        dy = rng.standard_normal(n).astype(np.float32)
        mask = (rng.random_sample(n) < 0.5).astype(np.uint8)
        xt, dyt, mt = (torch.from_numpy(t).to(env["dev"]) for t in (x, dy, mask))
        got = env["ops"].dropout_dense(xt, mt, 2.0).cpu().numpy()
        assert np.array_equal(got, np.where(mask != 0, x * np.float32(2.0), np.float32(0)))
        y = np.maximum(x, 0)
        got = env["ops"].relu_dropout_bwd(torch.from_numpy(y).to(env["dev"]), dyt, mt, 2.0).cpu().numpy()
        assert np.array_equal(got, np.where((y > 0) & (mask != 0), dy * np.float32(2.0), np.float32(0)))
        got = env["ops"].relu_dropout_bwd(torch.from_numpy(y).to(env["dev"]), dyt).cpu().numpy()
        assert np.array_equal(got, np.where(y > 0, dy, np.float32(0)))
    # Adam, 3 steps, with weight decay folded into the gradient
    n = 1433 * 16
    w = rng.standard_normal(n).astype(np.float32)
    m = np.zeros(n, np.float32)
    v = np.zeros(n, np.float32)
    wt, mt_, vt = (torch.from_numpy(t.copy()).to(env["dev"]) for t in (w, m, v))
    for t in (1, 2, 3):
        g = rng.standard_normal(n).astype(np.float32) * 0.01
        env["ops"].adam_step(wt, torch.from_numpy(g).to(env["dev"]), mt_, vt, 0.01, t, weight_decay=5e-4)
        env["oracle"].adam_step(w, (g + np.float32(5e-4) * w).astype(np.float32), m, v, 0.01, t)
    assert np.abs(wt.cpu().numpy() - w).max() <= 1e-6
    assert np.abs(mt_.cpu().numpy() - m).max() <= 1e-7


def test_dropout_philox_matches_mask_path(env):
    """The fused kernels regenerate exactly the mask gcnx_bernoulli_mask_u8 writes."""
    torch = env["torch"]
    rng = np.random.RandomState(8)
    for n, off in ((1, 0), (1027, 4), (1027, 3), (100003, 5 * (1 << 44) + 256 * 77)):
        x = rng.standard_normal(n).astype(np.float32)
        dy = rng.standard_normal(n).astype(np.float32)
        y = np.maximum(rng.standard_normal(n), 0).astype(np.float32)
        xt, dyt, yt = (torch.from_numpy(t).to(env["dev"]) for t in (x, dy, y))
        mask = env["oracle"].bernoulli_mask(n, 0.5, 247, off)
        mo = torch.zeros(n, dtype=torch.uint8, device=env["dev"])
        got = env["ops"].dropout_philox(xt, 0.5, 247, off, mask_out=mo).cpu().numpy()
        assert np.array_equal(mo.cpu().numpy(), mask)
        assert np.array_equal(got, np.where(mask != 0, x * np.float32(2.0), np.float32(0)))
        got = env["ops"].relu_dropout_bwd_philox(yt, dyt, 0.5, 247, off).cpu().numpy()
        assert np.array_equal(got, np.where((y > 0) & (mask != 0), dy * np.float32(2.0), np.float32(0)))


def test_philox_uniform_matches_oracle(env):
    for n, off in ((1, 0), (1001, 3), (4096, 1 << 40)):
        got = env["ops"].philox_uniform((n,), -1.0, 1.0, 124, off, env["dev"]).cpu().numpy()
        e = np.arange(off, off + n, dtype=np.uint64)
        r = env["oracle"].philox4x32_10(e >> np.uint64(2), np.uint64(0), 124)
        w = r[np.arange(n), (e & np.uint64(3)).astype(np.int64)]
        u = (w >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)
        ref = np.float32(-1.0) + np.float32(2.0) * u
        assert np.array_equal(got, ref)


@pytest.mark.parametrize("width", [7, 16, 64, 256])
def test_spmm_split_operand_bitwise(env, width):
    """[b; b2] read in place equals the assembled operand, bit for bit."""
    torch = env["torch"]
    a, b, rp, ci, v, _, _ = _spmm_case(env, 3000, 2500, 50000, width, hub=[(9, 4000)], seed=21)
    bt = torch.from_numpy(b).to(env["dev"])
    full = env["ops"].spmm(a, bt, relu=True)
    for n1 in (0, 1, 1000, 2499):
        top = bt[:n1] if n1 else bt[:0]
        got = env["ops"].spmm(a, top.contiguous() if n1 else torch.empty((0, width), device=env["dev"]),
                              b2=bt[n1:].contiguous(), relu=True)
        assert torch.equal(full, got), n1


def test_halo_plan_and_gather_rows(env):
    from gcn_amd import halo
    torch = env["torch"]
    rng = np.random.RandomState(31)
    n, n_loc, rb = 4000, 1000, 2000
    rows, cols, vals = random_coo(rng, n_loc, n, 30000)
    a = env["sparse"].Csr.from_coo(rows, cols, vals, (n_loc, n), env["dev"])
    csr, need = halo.device_halo_plan(a, rb, n_loc)
    c = a.col_idx.cpu().numpy().astype(np.int64)
    local = (c >= rb) & (c < rb + n_loc)
    need_ref = np.unique(c[~local])
    new_ref = np.where(local, c - rb, n_loc + np.searchsorted(need_ref, c))
    assert np.array_equal(need.cpu().numpy(), need_ref.astype(np.int32))
    assert np.array_equal(csr.col_idx.cpu().numpy(), new_ref.astype(np.int32))
    assert csr.n_cols == n_loc + len(need_ref)
    # product through [local | halo] equals the product with global columns
    b = torch.from_numpy(rng.standard_normal((n, 64)).astype(np.float32)).to(env["dev"])
    ref = env["ops"].spmm(a, b)
    hal = torch.empty((len(need_ref), 64), device=env["dev"])
    env["ops"].gather_rows(b, need, hal)
    assert torch.equal(hal, b[need.long()])
    got = env["ops"].spmm(csr, b[rb:rb + n_loc].contiguous(), b2=hal)
    assert torch.equal(ref, got)
    # strided source view (a column slab) and odd widths
    sub = torch.empty((len(need_ref), 7), device=env["dev"])
    env["ops"].gather_rows(b[:, 3:10], need, sub)
    assert torch.equal(sub, b[need.long(), 3:10])


@pytest.mark.parametrize("width,ld,col0", [(256, 256, 0), (64, 256, 128), (16, 16, 0), (7, 21, 7)])
def test_spmm_output_dropout_epilogue(env, width, ld, col0):
    """relu + dropout fused behind the product equals the product followed by
    the stand-alone mask (same Philox indices)."""
    torch = env["torch"]
    a, b, rp, ci, v, _, _ = _spmm_case(env, 2000, 1500, 40000, width, hub=[(3, 3000)], seed=33)
    bt = torch.from_numpy(b).to(env["dev"])
    off = 7 * (1 << 44) + 4 * 12345
    plain = env["ops"].spmm(a, bt, relu=True).cpu().numpy()
    got = env["ops"].spmm(a, bt, relu=True, dropout=(0.5, 991, off, ld, col0)).cpu().numpy()
    rows = np.arange(2000, dtype=np.uint64)[:, None]
    cols = np.arange(width, dtype=np.uint64)[None, :]
    m = env["oracle"].bernoulli_mask_at(np.uint64(off) + rows * np.uint64(ld) + np.uint64(col0) + cols, 0.5, 991)
    want = np.where(m != 0, plain * np.float32(2.0), np.float32(0))
    if np.array_equal(got, want):
        return
    # shapes whose plain product is served by another kernel (another order of
    # the k terms): the kept / dropped pattern is exact, the values agree to rounding
    _assert_dropout_fallback(env, got, want, m, rp, ci, v, b)


def _assert_dropout_fallback(env, got, want, m, rp, ci, v, b):
    """The check behind exact equality in test_spmm_output_dropout_epilogue: the
    kept / dropped pattern is exact and kept values agree to fp32 rounding,
    measured against the size of the summed terms, 2 * |A| |B| (the yardstick of
    helpers.assert_close_to_f64, from the oracle's fp64 product of the absolute
    values)."""
    assert np.array_equal(got[m == 0], want[m == 0])
    _, mag = env["oracle"].spmm_csr_f64(rp, ci, np.abs(v), np.abs(b))
    err = np.abs(got.astype(np.float64) - want.astype(np.float64))
    assert float((err / np.maximum(2.0 * mag, 1e-30)).max()) <= RTOL


def test_dropout_epilogue_fallback_check_is_live(env):
    """The tolerance branch above has to work when it is needed: a result that
    differs from the reference value by one rounding passes it, a wrong kept /
    dropped pattern or a wrong value fails it."""
    rng = np.random.RandomState(4)
    rows, cols, vals = random_coo(rng, 50, 40, 600)
    rp, ci, v, _ = env["oracle"].csr_from_coo(rows, cols, vals.astype(np.float32), (50, 40))
    b = rng.standard_normal((40, 8)).astype(np.float32)
    plain = np.maximum(env["oracle"].spmm_csr(rp, ci, v, b), 0)
    m = (rng.random_sample(plain.shape) < 0.5).astype(np.uint8)
    want = np.where(m != 0, plain * np.float32(2.0), np.float32(0))
    nudged = np.where(m != 0, np.nextafter(want, np.float32(np.inf), dtype=np.float32), want)
    _assert_dropout_fallback(env, nudged, want, m, rp, ci, v, b)
    wrong = want.copy()
    wrong[m != 0] *= np.float32(1.001)
    with pytest.raises(AssertionError):
        _assert_dropout_fallback(env, wrong, want, m, rp, ci, v, b)
    leaked = want.copy()
    leaked[m == 0] = 1.0
    with pytest.raises(AssertionError):
        _assert_dropout_fallback(env, leaked, want, m, rp, ci, v, b)


@pytest.mark.parametrize("ta,tb,M,N,K", [(0, 1, 3000, 256, 16), (0, 0, 700, 33, 20), (1, 0, 256, 16, 30000),
                                         (0, 1, 5001, 128, 16), (0, 0, 4096, 256, 128)])
def test_gemm_gated_epilogue(env, ta, tb, M, N, K):
    torch = env["torch"]
    rng = np.random.RandomState(M + K)
    a = rng.standard_normal((K, M) if ta else (M, K)).astype(np.float32)
    b = rng.standard_normal((N, K) if tb else (K, N)).astype(np.float32)
    gate = np.maximum(rng.standard_normal((M, N)), 0).astype(np.float32)
    at, bt, gt = (torch.from_numpy(t).to(env["dev"]) for t in (a, b, gate))
    plain = env["ops"].gemm(at, bt, trans_a=bool(ta), trans_b=bool(tb)).cpu().numpy()
    got = env["ops"].gemm(at, bt, trans_a=bool(ta), trans_b=bool(tb), gate=gt, gate_scale=2.0).cpu().numpy()
    assert np.array_equal(got, np.where(gate > 0, plain * np.float32(2.0), np.float32(0)))


def test_cabi_error_codes_not_exceptions(env):
    """Bad arguments come back as negative status codes with a message; nothing
    aborts or throws through the C boundary."""
    import ctypes as C
    from gcn_amd import _lib
    torch = env["torch"]
    lib = _lib.load()
    a = env["sparse"].Csr.from_coo([0, 1], [1, 0], [1.0, 2.0], (2, 2), env["dev"])
    b = torch.ones((2, 4), device=env["dev"])
    out = torch.empty((2, 4), device=env["dev"])
    P = lambda t: C.c_void_p(t.data_ptr())
    # workspace too small
    rc = lib.gcnx_spmm_csr_f32(a.struct_ref(), P(a.chunk_row), P(b), 4, P(out), 4, 4, 1.0, 0.0, None, None,
                               1.0, 0, None, 0, None)
    assert rc == _lib.ERR_WORKSPACE and b"workspace" in lib.gcnx_last_error()
    # leading dimension smaller than the width
    ws = torch.empty(4096, dtype=torch.uint8, device=env["dev"])
    rc = lib.gcnx_spmm_csr_f32(a.struct_ref(), P(a.chunk_row), P(b), 2, P(out), 4, 4, 1.0, 0.0, None, None,
                               1.0, 0, P(ws), 4096, None)
    assert rc == _lib.ERR_INVALID_ARG
    rc = lib.gcnx_gemm_f32(0, 0, 2, 2, -1, P(b), 4, P(b), 4, P(out), 4, 1.0, 0.0, 0, None, 0, None)
    assert rc == _lib.ERR_INVALID_ARG
    rc = lib.gcnx_adam_step_f32(4, P(out), P(out), P(out), P(out), 0.01, 0.9, 0.999, 1e-8, 0.0, 0, None)
    assert rc == _lib.ERR_INVALID_ARG          # t counts from 1
    cnt = C.c_int64(0)
    rc = lib.gcnx_rmat_edges(40, 0, 10, 0.57, 0.19, 0.19, 1, 0, 10, None, 0, C.byref(cnt), P(ws), None)
    assert rc == _lib.ERR_INVALID_ARG and b"scale" in lib.gcnx_last_error()
    with pytest.raises(ValueError):
        env["ops"].spmm(a, torch.ones((3, 4), device=env["dev"]))
    with pytest.raises(RuntimeError):
        env["ops"].spmm(a, torch.ones((2, 4)))          # CPU tensor: no fallback


@pytest.mark.parametrize("ta,M,N,K", [(0, 4096, 256, 256), (0, 3000, 16, 128), (1, 256, 256, 20000), (1, 128, 16, 9000)])
def test_gemm_dropout_on_a_matches_materialised(env, ta, M, N, K):
    """dropout(A) regenerated inside the product == product of the stored dropout(A)."""
    torch = env["torch"]
    rng = np.random.RandomState(M + K)
    a = rng.standard_normal((K, M) if ta else (M, K)).astype(np.float32)
    b = rng.standard_normal((K, N)).astype(np.float32)
    at, bt = torch.from_numpy(a).to(env["dev"]), torch.from_numpy(b).to(env["dev"])
    assert env["ops"].gemm_supports_a_dropout(bool(ta), M, N, K)
    off = 3 * (1 << 44) + 4 * 999
    ad = env["ops"].dropout_philox(at.view(-1), 0.5, 77, off).view(at.shape)
    ref = env["ops"].gemm(ad, bt, trans_a=bool(ta))
    got = env["ops"].gemm(at, bt, trans_a=bool(ta), a_dropout=(0.5, 77, off))
    if N > 16:
        assert torch.equal(ref, got)            # the same kernel serves both calls
    else:
        # the stored operand goes through the skinny streaming kernel, which sums
        # k in another order: same terms, fp32 rounding differs
        mag = (ad.abs().t() if ta else ad.abs()).double() @ bt.abs().double()
        assert float(((ref - got).abs().double() / mag).max()) <= RTOL
    assert not env["ops"].gemm_supports_a_dropout(False, 2708, 7, 16)


def test_device_step_counters(env):
    """The device-counter forms of the mask and Adam kernels equal the host-
    argument forms: offset + step * stride, and t read from memory."""
    torch, ops = env["torch"], env["ops"]
    dev = env["dev"]
    ctr = ops.device_counter(dev, 0)
    for step in (1, 2, 3):
        ops.counter_add(ctr, 1)
        assert int(ctr.item()) == step
        got = ops.bernoulli_mask(10007, 0.5, 99, 4, dev, step=ctr, step_stride=2 ** 40)
        want = ops.bernoulli_mask(10007, 0.5, 99, 4 + step * 2 ** 40, dev)
        assert torch.equal(got, want)
    ops.counter_add(ctr, 7)
    assert int(ctr.item()) == 10
    rng = np.random.RandomState(3)
    w0 = rng.standard_normal(5003).astype(np.float32)
    state = []
    for t_arg in ("host", "dev"):
        w = torch.from_numpy(w0.copy()).to(dev)
        m, v = torch.zeros_like(w), torch.zeros_like(w)
        tdev = ops.device_counter(dev, 0)
        grng = np.random.RandomState(4)
        for t in range(1, 6):
            g = torch.from_numpy(grng.standard_normal(5003).astype(np.float32)).to(dev)
            ops.counter_add(tdev, 1)
            ops.adam_step(w, g, m, v, 0.01, t if t_arg == "host" else tdev, weight_decay=5e-4)
        state.append((w.cpu().numpy(), m.cpu().numpy(), v.cpu().numpy()))
    for x, y in zip(*state):
        np.testing.assert_allclose(y, x, rtol=1e-6, atol=1e-8)
    with pytest.raises(ValueError):
        ops.adam_step(w, g, m, v, 0.01, torch.zeros(1, device=dev))


@pytest.mark.parametrize("n,width", [(1, 32), (1000, 256), (777, 64), (513, 100)])
def test_pack_sign_bits(env, n, width):
    torch = env["torch"]
    rng = np.random.RandomState(n + width)
    x = rng.standard_normal((n, width)).astype(np.float32)
    x[rng.random_sample(x.shape) < 0.1] = 0.0
    xt = torch.from_numpy(x).to(env["dev"])
    words = (width + 31) // 32

    def expect(v):
        on = np.zeros((v.shape[0], words * 32), dtype=np.uint64)
        on[:, :width] = v > 0
        return (on.reshape(v.shape[0], words, 32) << np.arange(32, dtype=np.uint64)).sum(axis=2).astype(np.uint32)

    got = env["ops"].pack_sign_bits(xt).cpu().numpy().view(np.uint32)
    assert np.array_equal(got, expect(x))
    rows = rng.randint(0, n, size=max(1, n // 3)).astype(np.int32)
    got = env["ops"].pack_sign_bits(xt, rows=torch.from_numpy(rows).to(env["dev"])).cpu().numpy().view(np.uint32)
    assert np.array_equal(got, expect(x[rows]))


@pytest.mark.parametrize("ta,tb,M,N,K", [(0, 1, 3000, 256, 16), (0, 1, 5001, 128, 16), (0, 0, 700, 33, 20),
                                         (0, 0, 4096, 256, 128), (0, 1, 2048, 64, 7),
                                         # split-K weight-gradient shapes and the skinny streaming kernels
                                         (1, 0, 256, 64, 5000), (1, 0, 256, 16, 40000), (0, 0, 2048, 16, 128)])
def test_gemm_bit_gate_equals_float_gate(env, ta, tb, M, N, K):
    """The gate read as one bit per element gives the bits of the float gate."""
    torch = env["torch"]
    rng = np.random.RandomState(M + K + 1)
    a = rng.standard_normal((K, M) if ta else (M, K)).astype(np.float32)
    b = rng.standard_normal((N, K) if tb else (K, N)).astype(np.float32)
    gate = np.maximum(rng.standard_normal((M, N)), 0).astype(np.float32)
    at, bt, gt = (torch.from_numpy(t).to(env["dev"]) for t in (a, b, gate))
    ref = env["ops"].gemm(at, bt, trans_a=bool(ta), trans_b=bool(tb), gate=gt, gate_scale=2.0)
    bits = env["ops"].pack_sign_bits(gt)
    got = env["ops"].gemm(at, bt, trans_a=bool(ta), trans_b=bool(tb), gate_bits=bits, gate_scale=2.0)
    assert torch.equal(ref, got)


def test_gemm_dropout_on_gathered_rows(env):
    """dropout(A) W for a gathered subset of the rows == the same rows of the
    full product (each row draws the mask of its global row id)."""
    torch = env["torch"]
    rng = np.random.RandomState(11)
    M, N, K = 6000, 256, 128
    a = torch.from_numpy(rng.standard_normal((M, K)).astype(np.float32)).to(env["dev"])
    b = torch.from_numpy(rng.standard_normal((K, N)).astype(np.float32)).to(env["dev"])
    off = 9 * (1 << 44)
    full = env["ops"].gemm(a, b, a_dropout=(0.5, 31, off))
    rows = np.sort(rng.choice(M, size=2500, replace=False)).astype(np.int32)
    rt = torch.from_numpy(rows).to(env["dev"])
    sub = a[rt.long()].contiguous()
    got = env["ops"].gemm(sub, b, a_dropout=(0.5, 31, off), a_dropout_rows=rt)
    assert torch.equal(got, full[rt.long()])


@pytest.mark.parametrize("M,N,full,kept", [(256, 256, 9000, 5000), (128, 512, 20000, 4097), (256, 256, 6000, 6000)])
def test_gemm_transposed_dropout_on_gathered_rows(env, M, N, full, kept):
    """dropout(A)^T B summed over a gathered subset of the stored rows, each drawing the
    mask of its row id (the weight gradient over the rows of dP1 that can be
    nonzero): equals the full product when B is zero in the other rows; refused
    where gcnx_gemm_dropout_at_rows_supported says no."""
    torch, ops, dev = env["torch"], env["ops"], env["dev"]
    from gcn_amd._lib import GcnxError
    rng = np.random.RandomState(M + kept)
    a = torch.from_numpy(rng.standard_normal((full, M)).astype(np.float32)).to(dev)
    b = rng.standard_normal((full, N)).astype(np.float32)
    rows = np.sort(rng.choice(full, size=kept, replace=False)).astype(np.int32)
    zero = np.ones(full, dtype=bool)
    zero[rows] = False
    b[zero] = 0.0
    bt = torch.from_numpy(b).to(dev)
    rt = torch.from_numpy(rows).to(dev)
    off = 5 * (1 << 44) + 4 * M * 3
    assert ops.gemm_supports_a_dropout_rows_t(M, N, kept)
    want = ops.gemm(a, bt, trans_a=True, a_dropout=(0.5, 77, off))
    got = ops.gemm(a[rt.long()].contiguous(), bt[rt.long()].contiguous(), trans_a=True, a_dropout=(0.5, 77, off),
                   a_dropout_rows=rt)
    # the same products summed in another grouping (the K split differs)
    assert float((got - want).abs().max()) <= 2e-6 * float(want.abs().max())
    # against the materialised mask, in fp64
    m = env["oracle"].bernoulli_mask_at(np.uint64(off) + rows.astype(np.uint64)[:, None] * np.uint64(M) +
                                        np.arange(M, dtype=np.uint64)[None, :], 0.5, 77)
    ad = np.where(m != 0, a.cpu().numpy()[rows].astype(np.float64) * 2.0, 0.0)
    ref = ad.T @ b[rows].astype(np.float64)
    mag = np.abs(ad).T @ np.abs(b[rows]).astype(np.float64)
    assert_close_to_f64(got.cpu().numpy(), ref, mag, what="transposed product over gathered rows")
    before = ops.dense_arith(1)
    try:
        assert not ops.gemm_supports_a_dropout_rows_t(M, N, kept)
        with pytest.raises(GcnxError):
            ops.gemm(a[rt.long()].contiguous(), bt[rt.long()].contiguous(), trans_a=True, a_dropout=(0.5, 77, off),
                     a_dropout_rows=rt)
    finally:
        ops.dense_arith(before)
    assert not ops.gemm_supports_a_dropout_rows_t(M, 128, kept)


@pytest.mark.parametrize("ta,tb,M,N,K", [(0, 0, 4096, 256, 256), (0, 0, 3001, 128, 64), (0, 0, 700, 33, 20),
                                         (0, 1, 2048, 64, 16), (0, 0, 5000, 16, 256), (1, 0, 128, 16, 9000)])
def test_gemm_output_dropout(env, ta, tb, M, N, K):
    """relu + dropout in the epilogue == relu of the plain product, masked with
    the Bernoulli mask at the same element indices (bit for bit)."""
    torch = env["torch"]
    rng = np.random.RandomState(M + N)
    a = rng.standard_normal((K, M) if ta else (M, K)).astype(np.float32)
    b = rng.standard_normal((N, K) if tb else (K, N)).astype(np.float32)
    at, bt = torch.from_numpy(a).to(env["dev"]), torch.from_numpy(b).to(env["dev"])
    fused = env["ops"].gemm_supports_output_dropout(bool(ta), M, N, K)
    assert fused == (not ta and M >= 512 and K % 16 == 0 and N % 4 == 0 and (N % 128 == 0 or N <= 32))
    ld, col0 = (N + 12, 8) if fused else (N, 0)
    off = 5 * (1 << 44) + 4 * 321
    plain = env["ops"].gemm(at, bt, trans_a=bool(ta), trans_b=bool(tb), relu=True).cpu().numpy()
    got = env["ops"].gemm(at, bt, trans_a=bool(ta), trans_b=bool(tb), relu=True,
                          dropout=(0.5, 4242, off, ld, col0)).cpu().numpy()
    rows = np.arange(M, dtype=np.uint64)[:, None]
    cols = np.arange(N, dtype=np.uint64)[None, :]
    m = env["oracle"].bernoulli_mask_at(np.uint64(off) + rows * np.uint64(ld) + np.uint64(col0) + cols, 0.5, 4242)
    want = np.where(m != 0, plain * np.float32(2.0), np.float32(0))
    if fused and N % 32 == 0:
        # the same call can also hand out the sign words of what it stored
        bits = torch.full((M, N // 32), -1, dtype=torch.int32, device=env["dev"])
        again = env["ops"].gemm(at, bt, trans_a=bool(ta), trans_b=bool(tb), relu=True,
                                dropout=(0.5, 4242, off, ld, col0), sign_bits=bits)
        assert np.array_equal(again.cpu().numpy(), got)
        assert torch.equal(bits, env["ops"].pack_sign_bits(again))
    if np.array_equal(got, want):
        return
    # shapes whose plain product is served by another kernel (another order of
    # the k terms): the kept / dropped pattern is exact, the values agree to rounding
    assert np.array_equal(got[m == 0], want[m == 0])
    mag = 2.0 * env["oracle"].gemm_f64(np.abs(a), np.abs(b), bool(ta), bool(tb))
    assert float((np.abs(got.astype(np.float64) - want) / np.maximum(mag, 1e-30)).max()) <= RTOL


@pytest.mark.parametrize("n,width,off", [(1000, 256, 4 * 77), (333, 100, 4 * 5), (64, 30, 3)])
def test_dropout_of_gathered_rows(env, n, width, off):
    """Row i draws the mask of row row_ids[i]: equals the same rows of the
    full-matrix dropout."""
    torch = env["torch"]
    rng = np.random.RandomState(n)
    total = 3 * n
    x = torch.from_numpy(rng.standard_normal((total, width)).astype(np.float32)).to(env["dev"])
    full = env["ops"].dropout_philox(x.view(-1), 0.5, 13, off).view(total, width)
    rows = np.sort(rng.choice(total, size=n, replace=False)).astype(np.int32)
    rt = torch.from_numpy(rows).to(env["dev"])
    got = env["ops"].dropout_rows_philox(x[rt.long()].contiguous(), rt, 0.5, 13, off)
    assert torch.equal(got, full[rt.long()])


def test_cabi_client_without_python(spmm_path):
    """tests/native/cabi_client.cpp: hipMalloc'd memory, its own stream, the C ABI
    only (no Python, no PyTorch in the process); it checks SpMM and the dense
    product against host loops and a status-code path, and exits 0."""
    import os
    import subprocess
    if spmm_path != "default":
        pytest.skip("separate process: the dispatch knobs of this process do not reach it")
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    exe = os.path.join(root, "gcn_amd", "lib", "cabi_client")
    assert os.path.exists(exe), "build it with make -C gcn_amd/csrc"
    res = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "C ABI client OK" in res.stdout


@pytest.mark.parametrize("n,nnz,width,split", [(3000, 40000, 256, False), (2000, 30000, 160, True),
                                               (900, 200000, 512, False),
                                               # the narrow kernel (lane groups per row)
                                               (2500, 50000, 128, True), (3000, 30000, 64, False),
                                               (1200, 20000, 32, False), (1500, 25000, 96, False)])
def test_spmm_sign_bits(env, n, nnz, width, split):
    """The sign pattern written by the SpMM epilogue == pack_sign_bits of the
    stored output, and the stored output is that of the plain call."""
    torch, dev = env["torch"], env["dev"]
    rng = np.random.RandomState(n + width)
    rows, cols, vals = random_coo(rng, n, n, nnz, hub_rows=((5, 3000),), empty_rows=(0, 17))
    a = env["sparse"].Csr.from_coo(rows, cols, vals, (n, n), dev)
    b = torch.from_numpy(rng.standard_normal((n, width)).astype(np.float32)).to(dev)
    assert env["ops"].spmm_supports_sign_bits(width)
    kw = dict(relu=True, dropout=(0.5, 99, 4 * 1000, width, 0))
    if split:
        kw["b2"] = b[n // 2:]
        b1 = b[:n // 2]
    else:
        b1 = b
    plain = env["ops"].spmm(a, b1, **kw)
    bits = torch.full((n, width // 32), -1, dtype=torch.int32, device=dev)
    got = env["ops"].spmm(a, b1, sign_bits=bits, **kw)
    if width > 128:
        assert torch.equal(plain, got)
    else:
        # a small matrix with a narrow operand takes the wavefront-per-row kernel
        # in the plain call and the chunk walk when sign words are asked for
        assert float((plain - got).abs().max()) <= 1e-5 * float(plain.abs().max())
    assert torch.equal(bits, env["ops"].pack_sign_bits(got))


@pytest.mark.parametrize("M,N", [(5, 3), (4096, 256), (700, 64)])
def test_gemm_empty_inner_dimension(env, M, N):
    """K = 0: the product is the zero matrix, so C = beta * C (then relu);
    shapes that would otherwise take the MFMA kernels included."""
    torch = env["torch"]
    a = torch.empty((M, 0), device=env["dev"])
    b = torch.empty((0, N), device=env["dev"])
    got = env["ops"].gemm(a, b)
    assert got.shape == (M, N) and float(got.abs().max()) == 0.0
    c = torch.randn((M, N), device=env["dev"])
    want = torch.clamp_min(c * 0.5, 0)
    got = env["ops"].gemm(a, b, out=c.clone(), beta=0.5, relu=True)
    assert torch.equal(got, want)


def test_gemm_and_spmm_with_no_rows(env):
    """M = 0 / an operand without columns: nothing is launched, shapes are kept."""
    torch = env["torch"]
    dev = env["dev"]
    got = env["ops"].gemm(torch.empty((0, 16), device=dev), torch.randn((16, 8), device=dev))
    assert got.shape == (0, 8)
    got = env["ops"].gemm(torch.randn((16, 4), device=dev), torch.empty((4, 0), device=dev))
    assert got.shape == (16, 0)


def test_degenerate_sizes_are_no_ops(env):
    """Zero rows / zero width / zero elements through every op: no launch with
    an empty grid, no fault, shapes kept."""
    import scipy.sparse as sp
    torch, ops, dev = env["torch"], env["ops"], env["dev"]
    from gcn_amd.sparse import Csr
    # a matrix with rows but no nonzeros, and an operand without columns
    a = Csr.from_scipy(sp.csr_matrix((6, 5), dtype=np.float32), device=dev)
    assert float(ops.spmm(a, torch.randn((5, 8), device=dev)).abs().max()) == 0.0
    assert ops.spmm(a, torch.empty((5, 0), device=dev)).shape == (6, 0)
    m = sp.random(50, 40, density=0.1, format="csr", dtype=np.float32, random_state=1)
    b = Csr.from_scipy(m, device=dev)
    assert ops.spmm(b, torch.empty((40, 0), device=dev)).shape == (50, 0)
    # element-wise helpers on empty tensors
    e = torch.empty((0, 16), device=dev)
    assert ops.dropout_philox(e, 0.5, 1, 0).shape == (0, 16)
    assert ops.pack_sign_bits(e).shape[0] == 0
    idx = torch.empty((0,), dtype=torch.int32, device=dev)
    out = ops.gather_rows(torch.randn((4, 16), device=dev), idx, torch.empty((0, 16), device=dev))
    assert out.shape == (0, 16)
    z = torch.empty((0, 7), device=dev)
    # the loss over zero rows is undefined (a mean over nothing): a clean error, not a launch
    from gcn_amd._lib import GcnxError
    with pytest.raises(GcnxError):
        ops.masked_softmax_xent(z, z.clone(), torch.empty((0,), dtype=torch.uint8, device=dev), want_grad=False)
    w = torch.empty((0,), device=dev)
    ops.adam_step(w, w.clone(), w.clone(), w.clone(), 0.01, 1)
    torch.cuda.synchronize()


@pytest.mark.parametrize("n,n_cols,nnz,frac", [(300, 260, 5000, 0.3), (64, 64, 64 * 64, 0.5), (1000, 1000, 64 * 7, 0.01),
                                               (50, 40, 777, 1.0), (50, 40, 777, 0.0), (10, 10, 0, 0.5),
                                               (5000, 4000, 200001, 0.1)])
def test_filter_columns_keeps_flagged_nonzeros_in_order(env, n, n_cols, nnz, frac):
    """gcnx_csr_filter_columns_count / _fill against numpy: row pointers, kept entries
    and their source slots; hub and empty rows; all / no columns flagged; nnz a
    multiple of 64 and zero.  And the product with an operand that is zero outside the
    flagged rows does not change."""
    torch, ops, dev = env["torch"], env["ops"], env["dev"]
    from gcn_amd.sparse import Csr
    rng = np.random.RandomState(n + nnz)
    if nnz:
        rows, cols, vals = random_coo(rng, n, n_cols, nnz, hub_rows=((3, nnz // 4),), empty_rows=(0, n - 1))
    else:
        rows = cols = np.zeros(0, dtype=np.int32)
        vals = np.zeros(0)
    a = Csr.from_coo(rows, cols, vals.astype(np.float32), (n, n_cols), device=dev)
    flag = (rng.random_sample(n_cols) < frac) if 0.0 < frac < 1.0 else np.full(n_cols, frac >= 1.0)
    sub = a.filter_columns(torch.from_numpy(flag).to(dev))
    rp, ci, v = a.row_ptr.cpu().numpy(), a.col_idx.cpu().numpy(), a.val.cpu().numpy()
    keep = flag[ci]
    before = np.concatenate([[0], np.cumsum(keep, dtype=np.int64)])
    assert sub.shape == a.shape and sub.nnz == int(keep.sum())
    assert np.array_equal(sub.row_ptr.cpu().numpy(), before[rp])
    assert np.array_equal(sub.col_idx.cpu().numpy(), ci[keep])
    assert np.array_equal(sub.val.cpu().numpy(), v[keep])
    assert np.array_equal(sub.input_index.cpu().numpy(), a.input_index.cpu().numpy()[keep])
    b = rng.standard_normal((n_cols, 16)).astype(np.float32)
    b[~flag] = 0.0
    bt = torch.from_numpy(b).to(dev)
    full, part = ops.spmm(a, bt), ops.spmm(sub, bt)
    assert float((full - part).abs().max()) <= 1e-6 * float(full.abs().max()) + 1e-30
    # a uint8 flag vector is taken as it is; a wrong length is refused before any launch
    sub2 = a.filter_columns(torch.from_numpy(flag.astype(np.uint8)).to(dev))
    assert sub2.nnz == sub.nnz and torch.equal(sub2.col_idx, sub.col_idx)
    with pytest.raises(ValueError):
        a.filter_columns(torch.zeros(n_cols + 1, dtype=torch.uint8, device=dev))


def test_compact_rows_then_scatter_equals_the_full_product(env):
    """Csr.compact_rows: the product with the matrix without its empty rows, put in
    place with scatter_rows into a zeroed result, equals the full product (whose
    empty rows are zero rows)."""
    torch, ops, dev = env["torch"], env["ops"], env["dev"]
    from gcn_amd.sparse import Csr
    rng = np.random.RandomState(4)
    n, n_cols = 4000, 3000
    rows, cols, vals = random_coo(rng, n, n_cols, 9000, hub_rows=((9, 2000),), empty_rows=(0, 1, n - 1))
    a = Csr.from_coo(rows, cols, vals.astype(np.float32), (n, n_cols), device=dev)
    sub, ids = a.compact_rows()
    deg = np.diff(a.row_ptr.cpu().numpy())
    assert np.array_equal(ids.cpu().numpy(), np.nonzero(deg > 0)[0]) and sub.n_rows == int((deg > 0).sum()) < n
    assert sub.nnz == a.nnz and sub.col_idx.data_ptr() == a.col_idx.data_ptr()
    for width in (16, 256):
        b = torch.from_numpy(rng.standard_normal((n_cols, width)).astype(np.float32)).to(dev)
        full = ops.spmm(a, b)
        part = ops.spmm(sub, b)
        out = torch.zeros_like(full)
        ops.scatter_rows(part, ids, out)
        assert torch.equal(out, full)


@pytest.mark.parametrize("width", [16, 64, 256])
def test_spmm_rows_subset_draws_masks_by_row_id(env, width):
    """gcnx_spmm_csr_rows_f32: the product over a subset of the rows of a matrix,
    stored next to each other (transpose -> column filter -> transpose -> keep_rows,
    the way the engine builds A_hat[N1, :]), with relu + dropout drawn by the id of
    each row and the sign words, equals the same rows of the full product: the kept /
    dropped pattern and the sign words exactly, the values to rounding (the rows are
    cut into partial sums at other places); a kept row that is empty stays a zero row."""
    torch, ops, dev = env["torch"], env["ops"], env["dev"]
    from gcn_amd.sparse import Csr
    rng = np.random.RandomState(11)
    n, n_cols = 3000, 2500
    rows, cols, vals = random_coo(rng, n, n_cols, 60000, hub_rows=((5, 4000), (1700, 900)), empty_rows=(40, 41))
    a = Csr.from_coo(rows, cols, vals.astype(np.float32), (n, n_cols), device=dev)
    pick = np.unique(np.concatenate([[5, 40, 1700, n - 1], rng.choice(n, 700, replace=False)]))
    flag = torch.zeros(n, dtype=torch.uint8, device=dev)
    flag[torch.from_numpy(pick).to(dev)] = 1
    ids = torch.from_numpy(pick.astype(np.int32)).to(dev)
    sub = ops.csr_keep_rows(ops.csr_transpose(ops.csr_filter_columns(ops.csr_transpose(a), flag)), ids)
    rp = a.row_ptr.cpu().numpy()
    assert sub.n_rows == len(pick) and sub.nnz == int((rp[pick + 1] - rp[pick]).sum())
    with pytest.raises(ValueError):
        ops.csr_keep_rows(a, ids)                              # rows that hold nonzeros would be left out
    b = torch.from_numpy(rng.standard_normal((n_cols, width)).astype(np.float32)).to(dev)
    drop = (0.5, 77, 4 * 999, width, 0)
    signs = width % 32 == 0
    bits_f = torch.zeros((n, width // 32), dtype=torch.int32, device=dev) if signs else None
    bits_s = torch.zeros((len(pick), width // 32), dtype=torch.int32, device=dev) if signs else None
    full = ops.spmm(a, b, relu=True, dropout=drop, **({"sign_bits": bits_f} if signs else {}))
    part = ops.spmm(sub, b, relu=True, dropout=drop, dropout_rows=ids, **({"sign_bits": bits_s} if signs else {}))
    ref = full[ids.long()]
    assert torch.equal(part != 0, ref != 0)
    rp_, ci_, v_, _ = _csr_np(a)
    _, mag = env["oracle"].spmm_csr_f64(rp_, ci_, v_, b.cpu().numpy())        # size of the summed terms per element
    err = (part.double() - ref.double()).abs().cpu().numpy()
    assert float((err / np.maximum(2.0 * mag[pick], 1e-30)).max()) <= RTOL
    assert not part[int(np.searchsorted(pick, 40))].any()
    if signs:
        assert torch.equal(bits_s, bits_f[ids.long()])
    # without row ids the same call draws the masks of rows 0, 1, 2, ...: another pattern
    other = ops.spmm(sub, b, relu=True, dropout=drop)
    assert not torch.equal(other != 0, ref != 0)
    # no dropout: the row ids are not consulted and the rows are those of the plain product
    plain = ops.spmm(sub, b, dropout_rows=ids)
    err = (plain.double() - ops.spmm(a, b)[ids.long()].double()).abs().cpu().numpy()
    assert float((err / np.maximum(mag[pick], 1e-30)).max()) <= RTOL


@pytest.fixture(params=["split_bf16", "f32_chain"])
def dense_arith(request, env):
    """Both arithmetic modes of the large dense products: exactly split operands
    on the bf16 matrix pipe (default) and fp32-input MFMA chains."""
    before = env["ops"].dense_arith(0 if request.param == "split_bf16" else 1)
    yield request.param
    env["ops"].dense_arith(before)


@pytest.mark.parametrize("n,nnz,width,halo", [(3000, 40000, 256, 0), (5000, 90000, 128, 700), (900, 9000, 64, 0),
                                               (4000, 200000, 512, 1500), (1500, 30000, 640, 0),
                                               # fewer nonzeros than one round / one chunk; an odd number of chunks
                                               (40, 30, 256, 0), (300, 9, 256, 5), (2000, 64 * 3 + 1, 256, 0)])
def test_spmm_factored_equals_two_kernel_path(env, dense_arith, n, nnz, width, halo):
    """A @ gate(D W^T) with the operand formed per nonzero against the stored
    intermediate (bit-gated product, then SpMM): same bits with fp32 chains, fp32
    rounding apart with split operands; and against the fp64 yardstick in both.
    Hub rows, empty rows, a halo part of the operand."""
    torch, ops, dev = env["torch"], env["ops"], env["dev"]
    from gcn_amd.sparse import Csr
    rng = np.random.RandomState(n + width)
    n_cols = n + halo
    rows, cols, vals = random_coo(rng, n, n_cols, nnz, hub_rows=((7, min(3000, nnz // 3)),), empty_rows=(0, 11, n - 1))
    a = Csr.from_coo(rows, cols, vals.astype(np.float32), (n, n_cols), device=dev)
    d = torch.from_numpy(rng.standard_normal((n_cols, 16)).astype(np.float32)).to(dev)
    w = torch.from_numpy((rng.standard_normal((width, 16)) * 0.3).astype(np.float32)).to(dev)     # W2 [hidden, classes]
    hsign = torch.from_numpy(np.maximum(rng.standard_normal((n_cols, width)), 0).astype(np.float32)).to(dev)
    bits = ops.pack_sign_bits(hsign)
    assert ops.spmm_supports_factored(width, 16)
    stored = ops.gemm(d, w, trans_b=True, gate_bits=bits, gate_scale=2.0)          # [n_cols, width]
    want = ops.spmm(a, stored)
    if halo:
        got = ops.spmm_factored(a, d[:n].contiguous(), w, bits[:n].contiguous(), gate_scale=2.0, trans_w=True,
                                d2=d[n:].contiguous(), gate_bits2=bits[n:].contiguous())
    else:
        got = ops.spmm_factored(a, d, w, bits, gate_scale=2.0, trans_w=True)
    if width > 128 and n_cols >= 1024 and dense_arith == "f32_chain":
        assert torch.equal(got, want)      # both walk a row's nonzeros in order, lane-private sums
    else:
        # at narrow widths the plain SpMM splits a row's nonzeros over lane groups, and
        # a product with few rows is served by the register-tiled kernel (other k order)
        assert float((got - want).abs().max()) <= 1e-5 * float(want.abs().max()) + 1e-30
    # W given as [inner, width]
    got2 = ops.spmm_factored(a, d[:n].contiguous(), w.t().contiguous(), bits[:n].contiguous(), gate_scale=2.0,
                             d2=d[n:].contiguous() if halo else None,
                             gate_bits2=bits[n:].contiguous() if halo else None)
    assert torch.equal(got2, got)
    # fp64 yardstick: the whole expression in fp64, error measured against the
    # size of its terms  sum_i |a_i| * scale * sum_k |d_k w_k|
    import scipy.sparse as sp
    on = hsign.cpu().numpy() > 0
    dn, wn = d.cpu().numpy().astype(np.float64), w.cpu().numpy().astype(np.float64)
    g64 = np.where(on, 2.0 * (dn @ wn.T), 0.0)
    gmag = np.where(on, 2.0 * (np.abs(dn) @ np.abs(wn).T), 0.0)
    rp, ci, v = a.row_ptr.cpu().numpy(), a.col_idx.cpu().numpy(), a.val.cpu().numpy().astype(np.float64)
    m = sp.csr_matrix((v, ci, rp), shape=(n, n_cols))
    assert_close_to_f64(got.cpu().numpy(), m @ g64, abs(m) @ gmag, what="factored spmm")


def test_spmm_factored_paths_and_general_scale(env, dense_arith):
    """Rows of D with a pitch that rules out 16-byte loads take the vector-ALU
    form of the kernel (always fp32 chains); a gate scale that is not a power of
    two multiplies the matrix value instead of the element (one rounding apart
    from the two-kernel path, inside the fp32 bar)."""
    torch, ops, dev = env["torch"], env["ops"], env["dev"]
    from gcn_amd.sparse import Csr
    rng = np.random.RandomState(77)
    n, width = 2500, 256
    rows, cols, vals = random_coo(rng, n, n, 60000, hub_rows=((3, 2000),), empty_rows=(5,))
    a = Csr.from_coo(rows, cols, vals.astype(np.float32), (n, n), device=dev)
    wide = torch.from_numpy(rng.standard_normal((n, 17)).astype(np.float32)).to(dev)
    d_odd = wide[:, :16]                                    # pitch 17 floats
    d = d_odd.contiguous()
    w = torch.from_numpy((rng.standard_normal((width, 16)) * 0.3).astype(np.float32)).to(dev)
    bits = ops.pack_sign_bits(torch.from_numpy(rng.standard_normal((n, width)).astype(np.float32)).to(dev))
    fast = ops.spmm_factored(a, d, w, bits, gate_scale=2.0, trans_w=True)
    slow = ops.spmm_factored(a, d_odd, w, bits, gate_scale=2.0, trans_w=True)
    if dense_arith == "f32_chain":
        assert torch.equal(fast, slow)
    else:
        mag2 = ops.spmm(Csr(a.n_rows, a.n_cols, a.row_ptr, a.col_idx, a.val.abs()),
                        ops.gemm(d.abs(), w.abs(), trans_b=True, gate_bits=bits, gate_scale=2.0))
        assert bool(((fast - slow).abs() <= 1e-6 * mag2 + 1e-30).all())
    for scale in (1.0 / 0.9, 3.0):
        want = ops.spmm(a, ops.gemm(d, w, trans_b=True, gate_bits=bits, gate_scale=scale))
        mag = ops.spmm(Csr(a.n_rows, a.n_cols, a.row_ptr, a.col_idx, a.val.abs()),
                       ops.gemm(d.abs(), w.abs(), trans_b=True, gate_bits=bits, gate_scale=scale))
        for dd in (d, d_odd):
            got = ops.spmm_factored(a, dd, w, bits, gate_scale=scale, trans_w=True)
            assert bool(((got - want).abs() <= 1e-6 * mag + 1e-30).all())


def test_spmm_factored_rejects_unsupported(env):
    torch, ops, dev = env["torch"], env["ops"], env["dev"]
    assert not ops.spmm_supports_factored(256, 8)
    assert not ops.spmm_supports_factored(48, 16)


@pytest.mark.parametrize("n", [1, 63, 1024, 1433 * 16, 256 * 256 + 3])
def test_l2_loss_kernel(env, n):
    """scale * sum(w^2) / 2 chained onto a device scalar (reference
    gcn/models.py:148-149), against the fp64 sum; reproducible."""
    torch = env["torch"]
    rng = np.random.RandomState(n)
    w = rng.standard_normal(n).astype(np.float32)
    wt = torch.from_numpy(w).to(env["dev"])
    want = 5e-4 * 0.5 * float((w.astype(np.float64) ** 2).sum())
    got = env["ops"].l2_loss(wt, 5e-4)
    assert got.shape == (1,) and abs(float(got.item()) - want) <= 1e-5 * want + 1e-12
    base = torch.tensor([1.25], dtype=torch.float32, device=env["dev"])
    got2 = env["ops"].l2_loss(wt, 5e-4, add=base)
    assert abs(float(got2.item()) - (1.25 + want)) <= 1e-5 * (1.25 + want)
    assert float(base.item()) == 1.25
    again = env["ops"].l2_loss(wt, 5e-4)
    assert np.array_equal(again.cpu().numpy().view(np.uint32), got.cpu().numpy().view(np.uint32))


@pytest.mark.parametrize("kind,M,N,K", [("nn", 4096, 256, 256), ("nn", 1000, 512, 64), ("nn", 513, 256, 80),
                                        ("nn", 2049, 256, 1040), ("nt", 2048, 256, 128),
                                        ("tn", 256, 256, 40000), ("tn", 128, 512, 8192 + 16)])
def test_split_bf16_products_carry_fp32_accuracy(env, kind, M, N, K):
    """The large products on the bf16 matrix pipe with exactly split operands
    (gcnx_dense_arith 0) against the fp64 yardstick, next to the fp32-input MFMA
    chains (mode 1) on the same inputs: both inside the 1e-5 bar, and the split
    form's worst error within 4x of the chain's (measured: about 1-2x; it drops
    partial products below 2^-23 of each product and adds six partial sums per
    16 values of k instead of one chain).  Includes the fused input dropout, whose
    mask must be identical in both forms, and values spread over 12 binades."""
    torch, ops, dev = env["torch"], env["ops"], env["dev"]
    rng = np.random.RandomState(M + N + K)
    ta, tb = kind == "tn", kind == "nt"
    a = (rng.standard_normal((K, M) if ta else (M, K)) * np.exp2(rng.randint(-6, 6, (K, M) if ta else (M, K)))).astype(np.float32)
    b = (rng.standard_normal((N, K) if tb else (K, N)) * 0.3).astype(np.float32)
    at, bt = torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)
    ref = env["oracle"].gemm_f64(a, b, ta, tb)
    mag = env["oracle"].gemm_f64(np.abs(a), np.abs(b), ta, tb)
    worst = {}
    outs = {}
    for mode in (0, 1):
        before = ops.dense_arith(mode)
        try:
            got = ops.gemm(at, bt, trans_a=ta, trans_b=tb).cpu().numpy()
            dropped = ops.gemm(at, bt, trans_a=ta, trans_b=tb, a_dropout=(0.5, 99, 4 * 1234)).cpu().numpy() \
                if kind != "nt" else None
        finally:
            ops.dense_arith(before)
        assert_close_to_f64(got, ref, mag, what="%s mode %d" % (kind, mode))
        worst[mode] = float((np.abs(got - ref) / np.maximum(mag, 1e-30)).max())
        outs[mode] = dropped
    assert worst[0] <= 4.0 * worst[1] + 1e-7, worst
    if kind != "nt":
        # same Philox mask in both forms: compare against the oracle's mask applied on the host
        mask = env["oracle"].bernoulli_mask(a.size, 0.5, 99, 4 * 1234).reshape(a.shape)
        ad = np.where(mask != 0, a * np.float32(2.0), np.float32(0)).astype(np.float32)
        refd = env["oracle"].gemm_f64(ad, b, ta, tb)
        magd = env["oracle"].gemm_f64(np.abs(ad), np.abs(b), ta, tb)
        for mode in (0, 1):
            assert_close_to_f64(outs[mode], refd, magd, what="%s dropout mode %d" % (kind, mode))
    print("split-bf16 vs fp32-chain worst error / summed magnitude (%s %dx%dx%d): %.3g vs %.3g" %
          (kind, M, N, K, worst[0], worst[1]))


def test_split_nn_product_on_strided_operands(env):
    """The split NN kernel streams A by LDS-DMA from (base + per-lane offset)
    addresses: A as a column slab of a wider matrix (row pitch > K, 16-byte
    aligned start), C as a column slab of a wider output, the last row tile one
    row deep, K steps: 5 (fewer than the three DMA slots' worth of look-ahead
    plus a clamped tail)."""
    torch, ops, dev = env["torch"], env["ops"], env["dev"]
    rng = np.random.RandomState(77)
    M, N, K = 1025, 256, 80
    wide = rng.standard_normal((M, K + 48)).astype(np.float32)
    b = (rng.standard_normal((K, N)) * 0.2).astype(np.float32)
    wt = torch.from_numpy(wide).to(dev)
    a_view = wt[:, 16:16 + K]                       # pitch K + 48, offset 64 bytes
    out_wide = torch.full((M, N + 64), 7.0, device=dev)
    out_view = out_wide[:, 32:32 + N]
    before = ops.dense_arith(0)
    try:
        ops.gemm(a_view, torch.from_numpy(b).to(dev), out=out_view)
    finally:
        ops.dense_arith(before)
    a = wide[:, 16:16 + K]
    assert_close_to_f64(out_view.cpu().numpy(), env["oracle"].gemm_f64(a, b, False, False),
                        env["oracle"].gemm_f64(np.abs(a), np.abs(b), False, False), what="strided split nn")
    got = out_wide.cpu().numpy()
    assert (got[:, :32] == 7.0).all() and (got[:, 32 + N:] == 7.0).all()   # nothing outside the slab is written


def test_relu_epilogues_propagate_nan(env):
    """relu in the reference's framework maps NaN to NaN; every fused relu here is
    a compare-and-select that does the same, in full and edge tiles of the dense
    kernels (both arithmetic modes, large and small shapes) and in the SpMM
    epilogues (wide, narrow and small-matrix kernels), so that a diverged run
    surfaces as a NaN loss instead of being clamped to zero."""
    torch, ops, dev = env["torch"], env["ops"], env["dev"]
    from gcn_amd.sparse import Csr
    g = torch.Generator().manual_seed(5)
    for (M, N, K) in ((1000, 256, 64), (700, 64, 48), (300, 16, 256), (130, 24, 20)):
        a = torch.randn(M, K, generator=g).to(dev)
        b = torch.randn(K, N, generator=g).to(dev)
        rows = [0, M // 2, M - 1]                     # first tile, interior, last (edge) tile
        for r in rows:
            a[r, 3] = float("nan")
        for mode in (0, 1):
            before = ops.dense_arith(mode)
            try:
                out = ops.gemm(a, b, relu=True)
            finally:
                ops.dense_arith(before)
            bad = torch.isnan(out).all(dim=1).cpu().numpy()
            assert bad[rows].all(), (M, N, K, mode)
            assert bad.sum() == len(rows)
            assert (out[~torch.isnan(out)] >= 0).all()
    # SpMM: a NaN in row c of the dense operand reaches every row that references column c
    for n, nnz, w in ((3000, 60000, 256), (3000, 60000, 16), (300, 2000, 32)):
        rng = np.random.RandomState(n + w)
        r = rng.randint(0, n, nnz).astype(np.int32)
        c = rng.randint(0, n, nnz).astype(np.int32)
        v = rng.standard_normal(nnz).astype(np.float32)
        a = Csr.from_coo(torch.from_numpy(r).to(dev), torch.from_numpy(c).to(dev), torch.from_numpy(v).to(dev), (n, n),
                         device=dev)
        b = torch.randn(n, w, generator=g).to(dev)
        b[7, :] = float("nan")
        out = ops.spmm(a, b, relu=True)
        hit = np.zeros(n, bool)
        hit[r[c == 7]] = True
        got = torch.isnan(out).all(dim=1).cpu().numpy()
        assert np.array_equal(got, hit), (n, nnz, w)


@pytest.mark.parametrize("width,pitch", [(16, 16), (5, 7), (64, 80)])
def test_scatter_rows_places_rows_and_leaves_the_rest(env, width, pitch):
    """gcnx_scatter_rows_f32: out[idx[j]] = src[j] for distinct idx, other rows
    untouched (16-byte path and scalar path), the inverse of gather_rows."""
    torch, ops, dev = env["torch"], env["ops"], env["dev"]
    g = torch.Generator().manual_seed(width)
    n, m = 5000, 1237
    idx = torch.randperm(n, generator=g)[:m].to(torch.int32).to(dev)
    src = torch.randn(m, pitch, generator=g).to(dev)[:, :width]
    out = torch.full((n, pitch), 3.0, device=dev)[:, :width]
    ops.scatter_rows(src, idx, out)
    want = torch.full((n, width), 3.0, device=dev)
    want[idx.long()] = src
    assert torch.equal(out, want)
    back = torch.empty((m, width), device=dev)
    ops.gather_rows(out.contiguous(), idx, back)
    assert torch.equal(back, src)


@pytest.mark.parametrize("chunk", [64, 512])
def test_gather_hint_changes_no_result(env, chunk):
    """gcnx_csr::gather_keep: the column degrees and the keep bits against numpy,
    and the wide SpMM with rows of rarely referenced columns streamed
    (non-temporal loads) against the same product without the hint: same bits,
    with every row streamed and with every row kept as well."""
    torch, ops, dev = env["torch"], env["ops"], env["dev"]
    from gcn_amd.sparse import Csr
    rng = np.random.RandomState(chunk)
    n, nnz, w = 6000, 150001, 256
    r = rng.randint(0, n, nnz).astype(np.int32)
    c = np.minimum((rng.pareto(1.2, nnz) * 20).astype(np.int64), n - 1).astype(np.int32)    # a few hot columns
    v = rng.standard_normal(nnz).astype(np.float32)
    a = Csr.from_coo(torch.from_numpy(r).to(dev), torch.from_numpy(c).to(dev), torch.from_numpy(v).to(dev), (n, n),
                     device=dev)
    a.spmm_chunk = chunk                                  # the chunk kernels (the small-matrix kernel has no hint)
    b = torch.randn(n, w, generator=torch.Generator().manual_seed(1)).to(dev)
    ref = ops.spmm(a, b, relu=True)
    deg = a.column_degrees().cpu().numpy()
    assert np.array_equal(deg, np.bincount(c, minlength=n))
    cols = a.col_idx.cpu().numpy()
    for thr in (None, 2, 40, 10 ** 9):
        used = a.set_gather_hint(w * 4, budget_bytes=64 * w * 4, min_degree=thr)
        if thr is None:
            # the smallest threshold (from 2, capped at 4095) whose columns fit the budget of 64 rows
            assert (deg >= used).sum() <= 64 or used == 4095
            assert used <= 2 or (deg >= used - 1).sum() > 64
        words = a.gather_keep.cpu().numpy().view(np.uint64)
        bits = np.unpackbits(words.view(np.uint8), bitorder="little")[:a.nnz].astype(bool)
        assert np.array_equal(bits, deg[cols] >= used)
        assert not np.unpackbits(words.view(np.uint8), bitorder="little")[a.nnz:].any()
        got = ops.spmm(a, b, relu=True)
        assert torch.equal(got, ref), thr
    a.clear_gather_hint()
    assert torch.equal(ops.spmm(a, b, relu=True), ref)


@pytest.mark.parametrize("n_sup,width,f64", [(1, 16, True), (2, 7, True), (4, 16, True), (4, 6, False), (3, 64, True)])
def test_interleaved_supports_equal_sum_of_products(env, n_sup, width, f64):
    """The multi-support sum of the reference, add_n over A_i . P_i
    (gcn/layers.py:213-221), as this library forms it: gcnx_coo_interleave_supports
    merges the supports' tuples into one over the columns col * S + i (checked entry
    by entry), and one SpMM of the merged matrix with P_cat viewed as [N * S, width]
    must equal the sum of the S tuple-order products of the oracle (one of the
    supports is the identity, as T_0 is; one has an empty row block)."""
    import ctypes as C
    from gcn_amd import _lib
    from gcn_amd.sparse import _ptr, _stream
    torch, dev, orc = env["torch"], env["dev"], env["oracle"]
    rng = np.random.RandomState(100 * n_sup + width)
    n = 700
    tuples = []
    for i in range(n_sup):
        if i == 0:
            r = c = np.arange(n, dtype=np.int32)
            v = np.ones(n)
        else:
            r, c, v = random_coo(rng, n, n, 2000 * (i + 1), empty_rows=list(range(100 * i, 100 * i + 40)))
        v = v.astype(np.float64 if f64 else np.float32)
        order = rng.permutation(len(v))                     # tuples arrive unsorted (gcn/utils.py:93-109)
        tuples.append((np.ascontiguousarray(np.stack([r[order], c[order]], 1).astype(np.int32)), v[order]))
    coords = [torch.from_numpy(t[0]).to(dev) for t in tuples]
    vals = [torch.from_numpy(t[1]).to(dev) for t in tuples]
    nnz = [len(t[1]) for t in tuples]
    total = sum(nnz)
    out_row = torch.empty(total, dtype=torch.int32, device=dev)
    out_col = torch.empty(total, dtype=torch.int32, device=dev)
    out_val = torch.empty(total, dtype=torch.float64 if f64 else torch.float32, device=dev)
    lib = _lib.load()
    _lib.check(lib.gcnx_coo_interleave_supports(
        n_sup, (C.c_int64 * n_sup)(*nnz), (C.c_void_p * n_sup)(*[c.data_ptr() for c in coords]),
        (C.c_void_p * n_sup)(*[c.data_ptr() + 4 for c in coords]), 2,
        (C.c_void_p * n_sup)(*[v.data_ptr() for v in vals]), 1 if f64 else 0, _ptr(out_row), _ptr(out_col),
        _ptr(out_val), _stream()))
    # entry by entry: support i's k-th tuple lands at offset(i) + k with its column moved to col * S + i
    want_row = np.concatenate([t[0][:, 0] for t in tuples])
    want_col = np.concatenate([t[0][:, 1].astype(np.int64) * n_sup + i for i, t in enumerate(tuples)]).astype(np.int32)
    want_val = np.concatenate([t[1] for t in tuples])
    assert np.array_equal(out_row.cpu().numpy(), want_row)
    assert np.array_equal(out_col.cpu().numpy(), want_col)
    assert np.array_equal(out_val.cpu().numpy().view(np.uint8), want_val.view(np.uint8))
    merged = env["sparse"].Csr.from_coo(out_row, out_col, out_val, (n, n * n_sup), dev)
    p = rng.standard_normal((n, n_sup * width)).astype(np.float32)          # P_cat = x . [W_0 | ... | W_S-1]
    got = env["ops"].spmm(merged, torch.from_numpy(p).to(dev).view(n * n_sup, width)).cpu().numpy()
    ref32 = np.zeros((n, width), dtype=np.float32)
    ref64 = np.zeros((n, width))
    mag = np.zeros((n, width))
    for i, (cd, v) in enumerate(tuples):
        pi = np.ascontiguousarray(p[:, i * width:(i + 1) * width])
        v32 = v.astype(np.float32)
        ref32 += orc.spmm_coo(cd[:, 0].copy(), cd[:, 1].copy(), v32, pi, n)   # tf.add_n of the supports' products
        rp, ci, cv, _ = orc.csr_from_coo(cd[:, 0].copy(), cd[:, 1].copy(), v32, (n, n))
        r64, m64 = orc.spmm_csr_f64(rp, ci, cv, pi)
        ref64 += r64
        mag += m64
    assert_close_to_f64(got, ref64, mag, what="merged supports S=%d w=%d" % (n_sup, width))
    assert rel_maxnorm(got, ref32) <= RTOL


def test_autograd_backward_runs_in_the_forward_products_arithmetic(env):
    """The arithmetic mode of the large products is a per-thread setting of the
    library and torch runs backward on a thread of its own: the tape node carries the
    forward call's mode.  With fp32 chains selected on the calling thread, both
    gradients of a product whose shapes the split-operand kernels would serve equal,
    bit for bit, the fp32-chain products formed directly on the calling thread -- and
    differ from the split-operand ones (so the check can tell the two apart)."""
    torch, ops, dev = env["torch"], env["ops"], env["dev"]
    g = torch.Generator(device=dev)
    g.manual_seed(21)
    a = (torch.rand((1024, 256), device=dev, generator=g) - 0.5).requires_grad_(True)
    b = (torch.rand((256, 256), device=dev, generator=g) - 0.5).requires_grad_(True)
    dy = torch.rand((1024, 256), device=dev, generator=g) - 0.5
    before = ops.dense_arith(1)
    try:
        y = ops.gemm_autograd(a, b)
        da, db = torch.autograd.grad(y, [a, b], dy)
        want_da = ops.gemm(dy, b.detach(), trans_b=True)
        want_db = ops.gemm(a.detach(), dy, trans_a=True)
        assert ops.dense_arith() == 1                 # the calling thread's mode is untouched
    finally:
        ops.dense_arith(before)
    assert torch.equal(da, want_da) and torch.equal(db, want_db)
    split_da = ops.gemm(dy, b.detach(), trans_b=True)           # default arithmetic: exactly split operands
    assert not torch.equal(split_da, want_da)
    assert float((split_da - want_da).abs().max()) <= 1e-5 * float(want_da.abs().max())
    # and the other way round: a forward product in the default arithmetic keeps it in backward
    y0 = ops.gemm_autograd(a, b)
    before = ops.dense_arith(1)
    try:
        da0, = torch.autograd.grad(y0, [a], dy)                  # backward asked for while fp32 chains are selected
    finally:
        ops.dense_arith(before)
    assert torch.equal(da0, split_da)
