"""CPU-only tests: the host-side preprocessing and the oracle against the golden
vectors captured from the reference's own gcn/utils.py, the oracle's model
restatement against an independent torch-CPU autograd computation, and the
C-ABI surface (library loads and exports what include/gcnx.h declares)."""
import ctypes
import hashlib
import io
import json
import os
import pickle
import re

import numpy as np
import pytest
import scipy.sparse as sp

import oracle
from oracle import model as omodel
from gcn_amd import planetoid, utils

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
GOLD = os.path.join(ROOT, "tests", "golden")

# Known answers recorded from a run of the reference's own load_data +
# preprocess_* when the reference was surveyed (first 16 hex digits of the
# sha256 of the little-endian bytes; coords int32 in (row, col) order).
REFERENCE_KNOWN = {
    "cora": dict(ahat_nnz=13264, coords="97df9dd060c18707", f64="2fe6db8a1469dfb3",
                 f32="3528c24a005ff79f", ahat_sum=2505.339270514625,
                 feat_nnz=49216, feat_coords="2389b411d4c7b256", feat_f32="00aca2d9db5fd0e5",
                 splits=(140, 500, 1000)),
    "citeseer": dict(ahat_nnz=12431, coords="06adf7a16f3c7109", f64="87e434675f2a24f0",
                     f32="6e9036d5605952fa", ahat_sum=3187.748582250952,
                     feat_nnz=105165, feat_coords="b7a1d79cbbb52fcb", feat_f32="008d85419358d900",
                     splits=(120, 500, 1000)),
    "pubmed": dict(ahat_nnz=108365, coords="4a9a0e839f69c501", f64="e355adb70cf118c7",
                   f32="4d81e2aaa0cde56a", ahat_sum=16352.724139980248),
}


def sha16(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]


def _sorted(coords, vals):
    o = np.lexsort((coords[:, 1], coords[:, 0]))
    return coords[o].astype(np.int32), vals[o]


# ------------------------------------------------------------ dataset reading
def test_decoder_refuses_foreign_globals():
    evil = pickle.dumps(os.getcwd, protocol=2)
    with pytest.raises(planetoid.PlanetoidFormatError):
        planetoid.decode(evil)
    with pytest.raises(planetoid.PlanetoidFormatError):
        planetoid.decode(pickle.dumps({"a": 1}, protocol=4))


def test_decoder_reads_numpy_and_csr_pickles():
    """Round trip of the three object shapes the Planetoid files hold, written
    here with pickle and read back with the non-executing decoder."""
    class _Py2Compat(pickle.Pickler):
        pass
    m = sp.random(13, 9, density=0.3, format="csr", dtype=np.float32, random_state=1)
    for obj in (np.arange(12, dtype=np.int32).reshape(3, 4), m):
        buf = io.BytesIO()
        pickle.Pickler(buf, protocol=2).dump(obj)
        try:
            got = planetoid.decode(buf.getvalue())
        except planetoid.PlanetoidFormatError:
            pytest.skip("this numpy/scipy pickles under module names the decoder does not list")
        if sp.issparse(obj):
            assert (got != obj).nnz == 0
        else:
            assert np.array_equal(got, obj)


@pytest.mark.parametrize("ds", ["cora", "citeseer", "pubmed"])
def test_ahat_matches_reference_known_answers(ds):
    k = REFERENCE_KNOWN[ds]
    adj = utils.load_graph(ds)
    coords, vals, shape = utils.preprocess_adj(adj)
    assert len(vals) == k["ahat_nnz"] and vals.dtype == np.float64
    c, v = _sorted(coords, vals)
    assert sha16(c) == k["coords"]
    assert sha16(v.astype("<f8")) == k["f64"]
    assert sha16(v.astype("<f4")) == k["f32"]
    assert abs(vals.sum() - k["ahat_sum"]) < 1e-9


@pytest.mark.parametrize("ds", ["cora", "citeseer"])
def test_load_data_and_features_match_reference_known_answers(ds):
    k = REFERENCE_KNOWN[ds]
    adj, feats, y_tr, y_va, y_te, m_tr, m_va, m_te = utils.load_data(ds)
    assert sp.isspmatrix_lil(feats)
    assert (int(m_tr.sum()), int(m_va.sum()), int(m_te.sum())) == k["splits"]
    assert m_tr.dtype == np.bool_
    coords, vals, shape = utils.preprocess_features(feats)
    assert len(vals) == k["feat_nnz"]
    c, v = _sorted(coords, vals)
    assert sha16(c) == k["feat_coords"]
    assert sha16(v.astype("<f4")) == k["feat_f32"]
    # labels only inside their mask
    assert np.all(y_tr[~m_tr] == 0) and np.all(y_tr[m_tr].sum(1) == 1)


def test_pubmed_features_unavailable():
    with pytest.raises(FileNotFoundError):
        utils.load_data("pubmed")


# -------------------------------------------------- golden vectors (reference)
@pytest.mark.parametrize("ds", ["cora", "citeseer"])
def test_preprocess_adj_golden_order_and_bits(ds):
    z = np.load(os.path.join(GOLD, "ahat_%s.npz" % ds))
    coords, vals, shape = utils.preprocess_adj(utils.load_graph(ds))
    assert np.array_equal(coords.astype(np.int32), z["coords"])      # same (unsorted) order
    assert np.array_equal(vals.view(np.uint64), z["values"].view(np.uint64))
    assert tuple(shape) == tuple(z["shape"])
    # oracle restatement of the normalisation on the same input
    a = (utils.load_graph(ds) + sp.eye(shape[0])).tocoo()
    rowsum = np.asarray(a.sum(1)).ravel()
    with np.errstate(divide="ignore"):
        d = np.power(rowsum, -0.5)
    d[np.isinf(d)] = 0
    r, c, v = oracle.normalize_coo(a.row, a.col, a.data, d)
    cs, vs = _sorted(np.stack([r, c], 1), v)
    gc, gv = _sorted(z["coords"], z["values"])
    assert np.array_equal(cs, gc) and np.array_equal(vs.view(np.uint64), gv.view(np.uint64))


def test_pubmed_ahat_sample_golden():
    z = np.load(os.path.join(GOLD, "ahat_pubmed_sample.npz"))
    coords, vals, _ = utils.preprocess_adj(utils.load_graph("pubmed"))
    assert np.array_equal(coords[z["index"]].astype(np.int32), z["coords"])
    assert np.array_equal(vals[z["index"]].view(np.uint64), z["values"].view(np.uint64))


@pytest.mark.parametrize("ds", ["cora", "citeseer"])
def test_features_and_splits_golden(ds):
    z = np.load(os.path.join(GOLD, "features_%s.npz" % ds))
    s = np.load(os.path.join(GOLD, "split_%s.npz" % ds))
    adj, feats, y_tr, y_va, y_te, m_tr, m_va, m_te = utils.load_data(ds)
    coords, vals, shape = utils.preprocess_features(feats)
    assert vals.dtype == z["values"].dtype
    assert np.array_equal(coords.astype(np.int32), z["coords"])
    assert np.array_equal(vals.view(np.uint8), z["values"].view(np.uint8))
    n = len(m_tr)
    for name, m in (("train_mask", m_tr), ("val_mask", m_va), ("test_mask", m_te)):
        assert np.array_equal(np.unpackbits(s[name])[:n].astype(bool), m)
    assert np.array_equal(y_tr.argmax(1), s["y_train_argmax"])
    assert np.array_equal(y_te.argmax(1), s["y_test_argmax"])
    assert np.array_equal((y_tr + y_va + y_te).sum(1), s["y_rowsum"])


@pytest.mark.parametrize("ds", ["cora", "citeseer"])
def test_chebyshev_golden(ds):
    """ARPACK's start vector is unseeded, so T_2/T_3 are compared with the
    eigenvalue of the golden run injected, by tolerance (never by hash)."""
    z = np.load(os.path.join(GOLD, "cheby3_%s.npz" % ds))
    adj = utils.load_graph(ds)
    polys = utils.chebyshev_polynomials(adj, 3, largest_eigval=float(z["lambda_max"]))
    assert len(polys) == 4
    for i, (c, v, shape) in enumerate(polys):
        m = sp.coo_matrix((v, (c[:, 0], c[:, 1])), shape=shape).tocsr()
        assert abs(len(v) - int(z["t%d_nnz" % i])) <= 4
        np.testing.assert_allclose(np.asarray(m.sum(1)).ravel(), z["t%d_rowsum" % i], rtol=0, atol=1e-9)
        np.testing.assert_allclose(np.asarray(abs(m).sum(1)).ravel(), z["t%d_abs_rowsum" % i], rtol=1e-9, atol=1e-9)
        sc, sv = z["t%d_sample_coords" % i], z["t%d_sample_values" % i]
        got = np.asarray(m[sc[:, 0], sc[:, 1]]).ravel()
        np.testing.assert_allclose(got, sv, rtol=1e-6, atol=1e-12)
    # without the override the eigenvalue agrees to solver precision
    polys2 = utils.chebyshev_polynomials(adj, 1)
    assert np.array_equal(polys2[1][0], polys[1][0])
    np.testing.assert_allclose(polys2[1][1], polys[1][1], rtol=1e-9, atol=1e-12)


def test_random_preprocess_golden():
    z = np.load(os.path.join(GOLD, "random_preprocess.npz"))
    ci = 0
    while "c%d_n" % ci in z.files:
        n = int(z["c%d_n" % ci])
        ic, iv = z["c%d_in_coords" % ci].reshape(-1, 2), z["c%d_in_values" % ci]
        m = sp.csr_matrix((iv, (ic[:, 0], ic[:, 1])), shape=(n, n))
        with np.errstate(divide="ignore"):
            o = utils.sparse_to_tuple(utils.normalize_adj(m.copy()))
            p = utils.preprocess_adj(m.copy())
        for got, name in ((o, "normalize_adj"), (p, "preprocess_adj")):
            assert np.array_equal(got[0].astype(np.int32).reshape(-1, 2), z["c%d_%s_coords" % (ci, name)])
            assert np.array_equal(got[1].view(np.uint64), z["c%d_%s_values" % (ci, name)].view(np.uint64))
        ci += 1
    assert ci == 6
    for fi in range(2):
        f = sp.lil_matrix(z["f%d_in" % fi])
        o = utils.preprocess_features(f)
        assert np.array_equal(o[0].astype(np.int32), z["f%d_coords" % fi])
        assert o[1].dtype == z["f%d_values" % fi].dtype
        assert np.array_equal(o[1].view(np.uint8), z["f%d_values" % fi].view(np.uint8))


def test_degree_table_matches_golden_hash():
    from gcn_amd.rmat import degree_table
    known = json.load(open(os.path.join(GOLD, "generated_answers.json")))
    t = degree_table(65536)
    assert t[0] == 0.0
    assert hashlib.sha256(t[1:].tobytes()).hexdigest() == known["pow_minus_half_1_65536_sha"]


def test_adjacency_matches_networkx():
    nx = pytest.importorskip("networkx")
    rng = np.random.RandomState(3)
    graph = {}
    for u in rng.permutation(40)[:35]:
        graph[int(u)] = [int(v) for v in rng.randint(0, 45, size=rng.randint(0, 6))]
    graph[7] = graph.get(7, []) + [7, 7, 3, 3]        # self-loop and duplicates
    ref = nx.adjacency_matrix(nx.from_dict_of_lists(graph))
    got = utils.adjacency_from_dict_of_lists(graph)
    assert got.shape == ref.shape and got.dtype == np.int64
    assert abs(got - sp.csr_matrix(ref)).sum() == 0


def test_construct_feed_dict_and_sparse_tuple():
    m = [sp.eye(3).tocsr(), sp.csr_matrix(np.array([[0, 2.0], [1.0, 0]]))]
    t = utils.sparse_to_tuple(m)
    assert t is m and t[1][0].shape == (2, 2) and t[1][0].dtype == np.int32
    ph = {"labels": "L", "labels_mask": "M", "features": "F", "support": ["S0", "S1"],
          "num_features_nonzero": "N"}
    feats = (np.zeros((5, 2), np.int32), np.ones(5), (4, 4))
    fd = utils.construct_feed_dict(feats, ["a", "b"], "y", "mask", ph)
    assert fd == {"L": "y", "M": "mask", "F": feats, "S0": "a", "S1": "b", "N": (5,)}


# ------------------------------------------------------------------ the oracle
def test_oracle_sparse_primitives_vs_scipy():
    rng = np.random.RandomState(0)
    r = rng.randint(0, 300, 5000).astype(np.int32)
    c = rng.randint(0, 200, 5000).astype(np.int32)
    v = rng.standard_normal(5000)
    rp, ci, val, perm = oracle.csr_from_coo(r, c, v, (300, 200))
    key = r.astype(np.int64) * 200 + c
    order = np.argsort(key, kind="stable")
    assert np.array_equal(perm, order)
    assert np.array_equal(val, v[order].astype(np.float32))
    m = sp.csr_matrix((val, ci, rp), shape=(300, 200))
    b = rng.standard_normal((200, 16)).astype(np.float32)
    ref = m.astype(np.float64) @ b.astype(np.float64)
    for got in (oracle.spmm_csr(rp, ci, val, b), oracle.spmm_csr(rp, ci, val, b, omp=True),
                oracle.spmm_coo(r, c, v.astype(np.float32), b, 300)):
        assert np.abs(got - ref).max() < 1e-4
    g = rng.standard_normal((300, 16)).astype(np.float32)
    adj = oracle.spmm_coo_adjoint(r, c, v.astype(np.float32), g, 200)
    assert np.abs(adj - m.T.astype(np.float64) @ g).max() < 1e-4
    trp, tci, tv, tperm = oracle.csr_transpose(rp, ci, val, 200)
    assert abs(sp.csr_matrix((tv, tci, trp), shape=(200, 300)) - m.T).sum() == 0
    with pytest.raises(ValueError):
        oracle.csr_from_coo([5], [0], [1.0], (3, 3))


def test_oracle_model_gradients_vs_torch_autograd():
    """The hand-written backward of the model restatement against torch CPU
    autograd in fp64 on the same weights and dropout masks."""
    torch = pytest.importorskip("torch")
    rng = np.random.RandomState(5)
    n, f, h, c, k = 60, 40, 8, 5, 3
    x = sp.random(n, f, density=0.2, random_state=rng, format="coo", dtype=np.float32)
    sups = []
    for _ in range(k):
        a = sp.random(n, n, density=0.1, random_state=rng, format="coo")
        sups.append((np.stack([a.row, a.col], 1).astype(np.int32), a.data, (n, n)))
    feats = (np.stack([x.row, x.col], 1).astype(np.int32), x.data, (n, f))
    w1 = [rng.standard_normal((f, h)).astype(np.float32) * 0.3 for _ in range(k)]
    w2 = [rng.standard_normal((h, c)).astype(np.float32) * 0.3 for _ in range(k)]
    labels = np.eye(c, dtype=np.float32)[rng.randint(0, c, n)]
    mask = (rng.random_sample(n) < 0.4).astype(np.uint8)
    keep = 0.5
    m1 = (rng.random_sample(x.nnz) < keep).astype(np.uint8)
    m2 = (rng.random_sample(n * h) < keep).astype(np.uint8)
    o = omodel.TwoLayerOracle("gcn_cheby", feats, sups, [w1, w2], weight_decay=5e-4)
    logits, cache = o.forward(m1, m2, keep)
    loss, acc, grad = o.loss_acc(logits, labels, mask, want_grad=True)
    g = o.backward(cache, grad)

    T = lambda a: torch.tensor(np.asarray(a), dtype=torch.float64)
    xd = T(sp.coo_matrix((x.data * m1 / keep, (x.row, x.col)), shape=(n, f)).toarray())
    A = [T(sp.coo_matrix((s[1], (s[0][:, 0], s[0][:, 1])), shape=(n, n)).toarray()) for s in sups]
    W1 = [T(w).requires_grad_(True) for w in w1]
    W2 = [T(w).requires_grad_(True) for w in w2]
    hh = torch.relu(sum(a @ (xd @ w) for a, w in zip(A, W1)))
    hd = hh * T(m2.reshape(n, h)) / keep
    out = sum(a @ (hd @ w) for a, w in zip(A, W2))
    logp = torch.log_softmax(out, 1)
    mk = T(mask)
    ce = -(T(labels) * logp).sum(1) * mk / mk.mean()
    tl = ce.mean() + sum(5e-4 * (w ** 2).sum() / 2 for w in W1)
    tl.backward()
    assert abs(loss - tl.item()) < 1e-5
    np.testing.assert_allclose(logits, out.detach().numpy(), rtol=0, atol=2e-5)
    for got, ref in zip(g[0] + g[1], W1 + W2):
        np.testing.assert_allclose(got, ref.grad.numpy(), rtol=0, atol=1e-6)
    am = (out.argmax(1) == T(labels).argmax(1)).double()
    assert abs(acc - (am * mk / mk.mean()).mean().item()) < 1e-6


def test_oracle_adam_matches_closed_form():
    rng = np.random.RandomState(1)
    w = rng.standard_normal(100).astype(np.float32)
    w0 = w.astype(np.float64).copy()
    m = np.zeros(100, np.float32)
    v = np.zeros(100, np.float32)
    mm = np.zeros(100)
    vv = np.zeros(100)
    for t in (1, 2, 3, 4):
        g = rng.standard_normal(100).astype(np.float32)
        oracle.adam_step(w, g, m, v, 0.01, t)
        mm = 0.9 * mm + 0.1 * g
        vv = 0.999 * vv + 0.001 * g.astype(np.float64) ** 2
        w0 -= 0.01 * np.sqrt(1 - 0.999 ** t) / (1 - 0.9 ** t) * mm / (np.sqrt(vv) + 1e-8)
    np.testing.assert_allclose(w, w0, rtol=0, atol=1e-5)


def test_philox_known_answer():
    """Random123 known-answer vectors for Philox4x32-10."""
    r = oracle.philox4x32_10(np.array([0], np.uint64), np.uint64(0), 0)[0]
    assert [int(x) for x in r] == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    ff = 0xFFFFFFFFFFFFFFFF
    r = oracle.philox4x32_10(np.array([ff], np.uint64), np.uint64(ff), ff)[0]
    assert [int(x) for x in r] == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    ctr_lo = (0x85a308d3 << 32) | 0x243f6a88
    ctr_hi = (0x03707344 << 32) | 0x13198a2e
    key = (0x299f31d0 << 32) | 0xa4093822
    r = oracle.philox4x32_10(np.array([ctr_lo], np.uint64), np.uint64(ctr_hi), key)[0]
    assert [int(x) for x in r] == [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_rmat_scramble_is_a_bijection():
    for scale in (1, 5, 10, 13):
        p = oracle.rmat_scramble_params(scale, 123)
        out = oracle.rmat_scramble(np.arange(1 << scale, dtype=np.uint64), p)
        assert len(np.unique(out)) == 1 << scale and out.max() < (1 << scale)


# ------------------------------------------------------------------- C ABI
def _declared_functions():
    text = open(os.path.join(ROOT, "include", "gcnx.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return set(re.findall(r"\b(gcnx_[a-z0-9_]+)\s*\(", text))


def test_cabi_exports_every_declared_symbol():
    from gcn_amd import _lib
    declared = _declared_functions()
    assert declared, "no declarations parsed"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    assert _lib.load().gcnx_abi_version() == _lib.ABI_VERSION == 5
    assert _lib.load().gcnx_spmm_chunk_nnz() == 64
    # pure host-side queries work without a GPU
    assert _lib.load().gcnx_spmm_num_chunks(1025) == 17
    m = _lib.CsrStruct(10, 10, 1025, None, None, None)
    assert _lib.load().gcnx_spmm_workspace_bytes(ctypes.byref(m), 7) == 17 * 2 * 8 * 4
    # the dispatch hint travels with the matrix descriptor (no process-wide knobs)
    m.spmm_chunk = 512
    assert _lib.load().gcnx_spmm_workspace_bytes(ctypes.byref(m), 7) == 3 * 2 * 8 * 4
    m.spmm_chunk = 100        # rounded up to 128
    assert _lib.load().gcnx_spmm_workspace_bytes(ctypes.byref(m), 7) == 9 * 2 * 8 * 4
    assert not any("row_path_limit" in n or "min_chunk" in n for n in declared)
    # the header's scratch-array size and the Python mirror agree
    text = open(os.path.join(ROOT, "include", "gcnx.h")).read()
    assert re.search(r"#define\s+GCNX_SPMM_STATIC_SCRATCH_BYTES\s+\(32u << 20\)", text)
    assert _lib.SPMM_STATIC_SCRATCH_BYTES == 32 << 20


def test_product_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "gcn_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), f
                assert "liboracle" not in text, f


def test_ops_fail_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from gcn_amd import ops, sparse
    with pytest.raises(RuntimeError):
        sparse.Csr.from_coo([0], [0], [1.0], (1, 1), "cpu")
    with pytest.raises(RuntimeError):
        ops.gemm(torch.zeros(2, 2), torch.zeros(2, 2))


# ------------------------------------------------------------- initialisers
def test_initialisers_range_dtype_and_seed_determinism():
    """glorot / uniform / zeros / ones (reference gcn/inits.py:5-27): bounds
    +-sqrt(6 / (fan_in + fan_out)) resp. +-scale, fp32, the given shape; the
    same graph seed reproduces the same bits, another layer (another random op
    of the same graph) draws other numbers, another seed too.  TensorFlow's own
    random stream cannot be reproduced, so the values themselves are not pinned
    (parity runs inject weights)."""
    from gcn_amd import inits, tfcompat as tf

    def draw(seed):
        tf.reset_default_graph()
        tf.set_random_seed(seed)
        with tf.variable_scope("m"):
            g1 = inits.glorot([1433, 16], name="weights_0")
            g2 = inits.glorot([1433, 16], name="weights_1")
            u = inits.uniform([40, 3], scale=0.05, name="u")
            z = inits.zeros([16], name="bias")
            o = inits.ones([5, 2], name="o")
        return g1, g2, u, z, o

    g1, g2, u, z, o = draw(123)
    r = np.sqrt(6.0 / (1433 + 16))
    for v, shape in ((g1, (1433, 16)), (g2, (1433, 16)), (u, (40, 3)), (z, (16,)), (o, (5, 2))):
        assert v._init_value.dtype == np.float32 and v._init_value.shape == shape == tuple(v.static_shape)
    for g in (g1, g2):
        a = g._init_value
        assert a.min() >= -r and a.max() <= r
        assert a.min() < -0.9 * r and a.max() > 0.9 * r          # the range is used, not a fraction of it
        assert abs(a.mean()) < 0.02 * r and abs(a.std() - r / np.sqrt(3.0)) < 0.02 * r   # uniform, not normal
    assert np.abs(u._init_value).max() <= 0.05 and np.abs(u._init_value).max() > 0.04
    assert not z._init_value.any() and (o._init_value == 1.0).all()
    assert g1.name == "m/weights_0:0" and z.name == "m/bias:0"
    # another layer of the same graph: another draw
    assert not np.array_equal(g1._init_value, g2._init_value)
    # same seed => same bits; other seed => other numbers
    h1, h2, hu, _, _ = draw(123)
    for a, b in ((g1, h1), (g2, h2), (u, hu)):
        assert np.array_equal(a._init_value.view(np.uint32), b._init_value.view(np.uint32))
    k1 = draw(124)[0]
    assert not np.array_equal(g1._init_value, k1._init_value)
    tf.reset_default_graph()


def test_loss_expression_splits_weight_decay_terms():
    """0 + wd * l2_loss(w) + cross-entropy (reference gcn/models.py:146-153) is
    one flat sum; minimize() hands the decay terms to the Adam kernel and keeps
    the rest for the tape."""
    from gcn_amd import tfcompat as tf
    tf.reset_default_graph()
    w = tf.Variable(tf.zeros([3, 2]), name="w")
    b = tf.Variable(tf.zeros([2]), name="b")
    ce = tf.Node(lambda run: None, (), "ce")
    loss = 0
    loss += 5e-4 * tf.nn.l2_loss(w)
    loss += tf.nn.l2_loss(b) * 2.0
    loss += ce
    assert isinstance(loss, tf._Sum) and len(loss.terms) == 3 and len(loss.l2_terms) == 2
    op = tf.train.AdamOptimizer(0.01).minimize(loss)
    assert op.rest is ce
    assert op.decay == {w: 5e-4, b: 2.0}
    only_decay = tf.train.AdamOptimizer(0.01).minimize(0 + 0.5 * tf.nn.l2_loss(w))
    assert only_decay.rest is None and only_decay.decay == {w: 0.5}
    plain = tf.train.AdamOptimizer(0.01).minimize(ce)
    assert plain.rest is ce and plain.decay == {}
    tf.reset_default_graph()


def test_model_vars_include_variables_created_outside_layers():
    """Model.vars is what save / load store (reference gcn/models.py:50-51 collects
    the global variables under the model's scope): a subclass that creates a
    variable in _build outside any layer gets it into vars; a second model of the
    same name does not pick up the first one's variables."""
    from gcn_amd import models, tfcompat as tf
    from gcn_amd.train import define_flags
    tf.reset_default_graph()
    define_flags()
    tf.app.flags.FLAGS._parse([])

    class WithTemperature(models.GCN):
        def _build(self):
            super(WithTemperature, self)._build()
            self.temperature = tf.Variable(tf.ones([1]), name="temperature")

    def placeholders():
        return {"support": [tf.sparse_placeholder(tf.float32)],
                "features": tf.sparse_placeholder(tf.float32, shape=(10, 5)),
                "labels": tf.placeholder(tf.float32, shape=(None, 3)),
                "labels_mask": tf.placeholder(tf.int32),
                "dropout": tf.placeholder_with_default(0., shape=()),
                "num_features_nonzero": tf.placeholder(tf.int32)}

    m1 = WithTemperature(placeholders(), input_dim=5, name="gcnt")
    names = sorted(m1.vars)
    assert "gcnt/temperature:0" in names and len(names) == 3, names
    assert m1.vars["gcnt/temperature:0"] is m1.temperature
    layer_vars = [v for layer in m1.layers for v in layer.vars.values()]
    assert all(m1.vars[v.name] is v for v in layer_vars) and len(layer_vars) == 2
    m2 = WithTemperature(placeholders(), input_dim=5, name="gcnt")
    assert m2.vars["gcnt/temperature:0"] is m2.temperature is not m1.temperature
    assert len(m2.vars) == 3 and not (set(map(id, m2.vars.values())) & set(map(id, m1.vars.values())))
    plain = models.GCN(placeholders(), input_dim=5)
    assert len(plain.vars) == 2
    tf.reset_default_graph()
