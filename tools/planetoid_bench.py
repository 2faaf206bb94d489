"""Epoch time and accuracy of the drop-in path on the Planetoid configurations
(Cora, Citeseer, Pubmed; gcn, gcn_cheby, dense), next to the CPU oracle's epoch time on the host.

Pubmed's feature blob (ind.pubmed.allx) is not distributed with the reference
tree, so Pubmed runs on its REAL graph, labels and splits with shape-matched
SYNTHETIC features ([19717, 500], 50 nonzeros per row): epoch time only, its
accuracy means nothing.

    python tools/planetoid_bench.py [--epochs 200] [--out file.json]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def pubmed_with_synthetic_features():
    from gcn_amd import utils
    return utils.load_pubmed_synthetic_features()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--epochs", type=int, default=200)
    ap.add_argument("--out", default="")
    ap.add_argument("--oracle-epochs", type=int, default=5)
    ap.add_argument("--hip-graph", action="store_true", help="replay each session.run from a HIP graph")
    ap.add_argument("--chebyshev-recurrence", action="store_true",
                    help="gcn_cheby by Clenshaw recurrence over the pattern of L")
    ap.add_argument("--only", default="", help="run one configuration, e.g. pubmed:gcn_cheby")
    args = ap.parse_args()
    import oracle
    from oracle import model as omodel
    from gcn_amd import tfcompat as tf, train, utils
    train.define_flags()
    F = tf.app.flags.FLAGS
    F._parse([])
    F.hip_graph = bool(args.hip_graph)
    F.chebyshev_recurrence = bool(args.chebyshev_recurrence)
    results = []
    for ds, model in (("cora", "gcn"), ("citeseer", "gcn"), ("pubmed", "gcn"), ("cora", "gcn_cheby"),
                      ("citeseer", "gcn_cheby"), ("pubmed", "gcn_cheby"), ("cora", "dense")):
        if args.only and args.only != "%s:%s" % (ds, model):
            continue
        F.dataset, F.model, F.epochs = ds, model, args.epochs
        data = pubmed_with_synthetic_features() if ds == "pubmed" else utils.load_data(ds)
        t0 = time.time()
        res = train.run(data=data, verbose=False)
        wall = time.time() - t0
        ms = sorted(r[4] * 1e3 for r in res["trace"][5:]) or [float("nan")]
        # oracle epoch time (train step + validation forward) on the host CPU
        adj, feats, y_tr, y_va, y_te, m_tr, m_va, m_te = data
        features = utils.preprocess_features(feats.copy())
        support = utils.chebyshev_polynomials(adj, 3) if model == "gcn_cheby" else [utils.preprocess_adj(adj)]
        n_sup = len(support) if model != "dense" else 1
        rng = np.random.RandomState(0)
        f, c = features[2][1], y_tr.shape[1]
        w1 = [rng.uniform(-.1, .1, (f, 16)).astype(np.float32) for _ in range(n_sup)]
        w2 = [rng.uniform(-.1, .1, (16, c)).astype(np.float32) for _ in range(n_sup)]
        orc = omodel.TwoLayerOracle(model, features, support, [w1, w2])
        n = features[2][0]
        cpu_ms = []
        for e in range(args.oracle_epochs):
            m1 = (rng.random_sample(len(features[1])) < 0.5).astype(np.uint8)
            m2 = (rng.random_sample(n * 16) < 0.5).astype(np.uint8)
            t1 = time.perf_counter()
            orc.train_step(y_tr.astype(np.float32), m_tr.astype(np.uint8), m1, m2, 0.5)
            orc.evaluate(y_va.astype(np.float32), m_va.astype(np.uint8))
            cpu_ms.append((time.perf_counter() - t1) * 1e3)
        rec = {"dataset": ds, "model": model, "features": "synthetic" if ds == "pubmed" else "real",
               "epochs_run": res["epochs_run"], "test_acc": res["test_acc"] if ds != "pubmed" else None,
               "gpu_epoch_ms_median": ms[len(ms) // 2], "gpu_epoch_ms_min": ms[0],
               "cpu_oracle_epoch_ms_median": sorted(cpu_ms)[len(cpu_ms) // 2] if cpu_ms else None, "cpu_threads": 1, "hip_graph": bool(args.hip_graph),
               "chebyshev_recurrence": bool(args.chebyshev_recurrence),
               "nnz_support": [int(len(s[1])) for s in support], "wall_s": wall}
        print(json.dumps(rec), flush=True)
        results.append(rec)
    if args.out:
        with open(args.out, "w") as fh:
            json.dump(results, fh, indent=1)


if __name__ == "__main__":
    main()
