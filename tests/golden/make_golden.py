"""Regenerates the golden vectors in this directory from the reference's own
``gcn/utils.py`` (the only reference module that imports without TensorFlow).

Run in the build container only (needs /root/reference); the fixtures it writes
are data: inputs and the reference's outputs.

  python tests/golden/make_golden.py

What is executed from the reference: ``preprocess_adj``, ``normalize_adj``,
``preprocess_features``, ``sparse_to_tuple``, ``chebyshev_polynomials``,
``sample_mask``, ``construct_feed_dict``.  Its ``load_data`` is NOT run (it
unpickles the dataset blobs); the datasets are decoded by
``gcn_amd.planetoid`` and assembled by ``gcn_amd.utils.load_data``, which is
pinned instead by the sha256 known-answers in ``known_answers.json`` that were
recorded from a run of the reference's ``load_data`` when the reference was
surveyed.

Files over ~300 KB are reduced to sha256 digests plus a sample of entries.
"""
import hashlib
import importlib.util
import json
import os
import sys
import types

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
REF = "/root/reference/gcn-master/gcn/utils.py"
sys.path.insert(0, ROOT)

from gcn_amd import utils as mine  # noqa: E402

_LAMBDA = []


def _eigsh(*a, **kw):
    w, v = spla.eigsh(*a, **kw)
    _LAMBDA.append(float(w[0]))
    return w, v


def import_reference_utils():
    """The file asks for ``scipy.sparse.linalg.eigen.arpack.eigsh``, a module
    path modern scipy dropped; supply it as an in-memory alias of
    ``scipy.sparse.linalg.eigsh`` (wrapped to record the eigenvalue)."""
    eig = types.ModuleType("scipy.sparse.linalg.eigen")
    arp = types.ModuleType("scipy.sparse.linalg.eigen.arpack")
    arp.eigsh = _eigsh
    eig.arpack = arp
    sys.modules["scipy.sparse.linalg.eigen"] = eig
    sys.modules["scipy.sparse.linalg.eigen.arpack"] = arp
    spec = importlib.util.spec_from_file_location("ref_gcn_utils", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def tuple_record(t, full):
    coords, vals, shape = t
    rec = {
        "coords": coords.astype(np.int32),
        "shape": np.asarray(shape, dtype=np.int64),
    }
    if full:
        rec["values"] = vals
    return rec


def main():
    ref = import_reference_utils()
    known = {}
    rng = np.random.RandomState(20251111)

    # ---- datasets: A_hat and features -----------------------------------
    for ds in ("cora", "citeseer", "pubmed"):
        adj = mine.load_graph(ds)
        a_hat = ref.preprocess_adj(adj)
        coords, vals, shape = a_hat
        order = np.lexsort((coords[:, 1], coords[:, 0]))
        known[ds] = {
            "n": int(shape[0]),
            "adj_nnz": int(adj.nnz),
            "ahat_nnz": int(len(vals)),
            "ahat_coords_sha_ref_order": sha(coords.astype(np.int32)),
            "ahat_vals_f64_sha_ref_order": sha(vals.astype("<f8")),
            "ahat_coords_sha_sorted": sha(coords[order].astype(np.int32)),
            "ahat_vals_f64_sha_sorted": sha(vals[order].astype("<f8")),
            "ahat_vals_f32_sha_sorted": sha(vals[order].astype("<f4")),
            "ahat_sum": float(vals.sum()),
        }
        if ds != "pubmed":
            np.savez_compressed(os.path.join(HERE, "ahat_%s.npz" % ds),
                                coords=coords.astype(np.int32), values=vals,
                                shape=np.asarray(shape, dtype=np.int64))
        else:
            pick = np.sort(rng.choice(len(vals), 4096, replace=False))
            np.savez_compressed(os.path.join(HERE, "ahat_%s_sample.npz" % ds),
                                index=pick, coords=coords[pick].astype(np.int32),
                                values=vals[pick],
                                shape=np.asarray(shape, dtype=np.int64))

    for ds in ("cora", "citeseer"):
        out = mine.load_data(ds)
        adj, feats, y_tr, y_va, y_te, m_tr, m_va, m_te = out
        ft = ref.preprocess_features(feats.copy())
        coords, vals, shape = ft
        known[ds].update({
            "feat_nnz": int(len(vals)),
            "feat_dtype": str(vals.dtype),
            "feat_coords_sha": sha(coords.astype(np.int32)),
            "feat_vals_sha": sha(vals),
            "feat_vals_f32_sha": sha(vals.astype("<f4")),
            "feat_sum": float(vals.astype(np.float64).sum()),
            "labels_argmax_sha": sha((y_tr + y_va + y_te).argmax(1).astype(np.int32)),
            "n_train": int(m_tr.sum()), "n_val": int(m_va.sum()), "n_test": int(m_te.sum()),
        })
        np.savez_compressed(os.path.join(HERE, "features_%s.npz" % ds),
                            coords=coords.astype(np.int32), values=vals,
                            shape=np.asarray(shape, dtype=np.int64))
        # reference sample_mask on the same index sets
        n = y_tr.shape[0]
        assert np.array_equal(ref.sample_mask(range(int(m_tr.sum())), n), m_tr)
        np.savez_compressed(
            os.path.join(HERE, "split_%s.npz" % ds),
            train_mask=np.packbits(m_tr), val_mask=np.packbits(m_va),
            test_mask=np.packbits(m_te),
            y_train_argmax=y_tr.argmax(1).astype(np.int8),
            y_val_argmax=y_va.argmax(1).astype(np.int8),
            y_test_argmax=y_te.argmax(1).astype(np.int8),
            y_rowsum=(y_tr + y_va + y_te).sum(1).astype(np.int8))

        # ---- Chebyshev K=3, with the eigenvalue that produced it ----------
        del _LAMBDA[:]
        polys = ref.chebyshev_polynomials(adj, 3)
        lam = _LAMBDA[-1]
        rec = {"lambda_max": np.float64(lam)}
        for i, (c, v, s) in enumerate(polys):
            m = sp.coo_matrix((v, (c[:, 0], c[:, 1])), shape=s).tocsr()
            rec["t%d_nnz" % i] = np.int64(len(v))
            rec["t%d_sum" % i] = np.float64(v.sum())
            rec["t%d_rowsum" % i] = np.asarray(m.sum(1)).ravel()
            rec["t%d_abs_rowsum" % i] = np.asarray(abs(m).sum(1)).ravel()
            pick = np.sort(rng.choice(len(v), min(2048, len(v)), replace=False))
            rec["t%d_sample_coords" % i] = c[pick].astype(np.int32)
            rec["t%d_sample_values" % i] = v[pick]
        np.savez_compressed(os.path.join(HERE, "cheby3_%s.npz" % ds), **rec)

    # ---- small random matrices through normalize_adj / preprocess_adj ------
    cases = {}
    for ci, (n, dens, sym) in enumerate([(1, 1.0, True), (7, 0.4, True), (50, 0.1, True),
                                         (64, 0.05, False), (200, 0.02, True), (33, 0.0, True)]):
        m = sp.random(n, n, density=dens, random_state=rng, format="coo",
                      data_rvs=lambda k: rng.randint(1, 4, size=k).astype(np.float64))
        if sym:
            m = (m + m.T).tocoo()
        m = sp.csr_matrix(m)
        cases["c%d_in_coords" % ci] = np.vstack(m.tocoo().coords).T.astype(np.int32) \
            if hasattr(m.tocoo(), "coords") else np.vstack((m.tocoo().row, m.tocoo().col)).T.astype(np.int32)
        cases["c%d_in_values" % ci] = m.tocoo().data
        cases["c%d_n" % ci] = np.int64(n)
        for fn in ("normalize_adj", "preprocess_adj"):
            o = getattr(ref, fn)(m.copy())
            if fn == "normalize_adj":
                o = ref.sparse_to_tuple(o)
            cases["c%d_%s_coords" % (ci, fn)] = o[0].astype(np.int32).reshape(-1, 2)
            cases["c%d_%s_values" % (ci, fn)] = o[1]
    # features: rows with zero sum, fp32 and fp64
    for ci, dt in enumerate((np.float32, np.float64)):
        f = sp.random(40, 25, density=0.15, random_state=rng, format="lil", dtype=dt)
        f[3, :] = 0
        f[17, :] = 0
        cases["f%d_in" % ci] = f.toarray()
        o = ref.preprocess_features(f.copy())
        cases["f%d_coords" % ci] = o[0].astype(np.int32)
        cases["f%d_values" % ci] = o[1]
    np.savez_compressed(os.path.join(HERE, "random_preprocess.npz"), **cases)

    # ---- degree table: numpy's pow(k, -0.5), k = 1..65536 -------------------
    known["pow_minus_half_1_65536_sha"] = sha(np.power(np.arange(1, 65537, dtype=np.float64), -0.5))

    with open(os.path.join(HERE, "generated_answers.json"), "w") as fh:
        json.dump(known, fh, indent=1, sort_keys=True)
    for f in sorted(os.listdir(HERE)):
        print("%9d  %s" % (os.path.getsize(os.path.join(HERE, f)), f))


if __name__ == "__main__":
    main()
