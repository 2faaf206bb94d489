"""Headline benchmark: full-batch two-layer GCN training steps on a synthetic
RMAT graph (the headline configuration: RMAT 2^24 nodes, avg degree 16, F=256, hidden
256, full-batch GCN on one MI355X; with N GPUs the graph grows to
2^(24+log2 N) nodes, row-partitioned, fixed 2^24 rows per GPU = weak scaling).

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
        --master-addr 127.0.0.1 --master-port P bench.py --gpus N --steps K --warmup W
    ... --gpus 8 --edge-factor 16      # RMAT 2^27 nodes, average degree 32 over 8 GPUs

A step is one optimisation step = one epoch of the full-batch model: forward
(dropout, X W1, A_hat P1 + relu, dropout, H W2, A_hat P2), masked loss,
backward (A_hat^T dZ, H'^T dP2, dP2 W2^T, A_hat^T dH, X'^T dP1) and Adam, with
every input resident in HBM.  Rank 0 prints ONE JSON line.  `value` is the
SpMM work of the whole job per second, in nnz*F/s: (nnz(A_hat) * sum of the
widths of the four SpMM of a step) * steps / time.  `roofline` is the dominant
kernel (the F=256 SpMM) timed alone with HIP events inside the timed region;
`cpu_baseline` is the oracle's restatement of the TensorFlow CPU SpMM timed on
a bounded slice of the same matrix.
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12  # bytes/s, MI355X HBM3E (spec)
MFMA_F32_PEAK = 157.3e12  # flop/s, dense fp32 matrix rate of MI355X (256 CUs x 256 flop/clk x 2.4 GHz)
MFMA_BF16_PEAK = 2.5e15   # flop/s, dense bf16 matrix rate
# The large products run on the bf16 matrix pipe with every fp32 operand split
# exactly into three bf16 pieces: six bf16 partial products per fp32 product, so
# the ceiling of that formulation in fp32-product flop/s is the bf16 peak / 6.
SPLIT_PARTIAL_PRODUCTS = 6


def algorithmic_bytes(nnz, n_rows, width):
    """No-reuse gather model of one SpMM: every nonzero reads its column index,
    its value and one row of the dense operand; every output row is written
    once; row_ptr is read once."""
    return nnz * (4 + 4 + 4 * width) + n_rows * 4 * width + 8 * (n_rows + 1)


def cpu_baseline(a, b_dev, rows, width):
    """Oracle SpMM (TensorFlow CPU order, one thread; then OpenMP over rows) on
    the first `rows` rows of the same matrix against the same dense operand."""
    import numpy as np
    import oracle
    oracle.build()
    rp = a.row_ptr[:rows + 1].cpu().numpy()
    nnz = int(rp[-1])
    ci = a.col_idx[:nnz].cpu().numpy()
    v = a.val[:nnz].cpu().numpy()
    b = b_dev.cpu().numpy()
    r = np.repeat(np.arange(rows, dtype=np.int32), np.diff(rp)).astype(np.int32)
    t0 = time.perf_counter()
    oracle.spmm_coo(r, ci, v, b, rows)
    t1 = time.perf_counter()
    oracle.spmm_csr(rp, ci, v, b, omp=True)
    t2 = time.perf_counter()
    scipy_entry = None
    try:                                     # third figure: scipy's csr @ dense (one thread, its own C++ loop)
        import scipy.sparse as sp
        m = sp.csr_matrix((v, ci, rp), shape=(rows, b.shape[0]))
        t3 = time.perf_counter()
        m @ b
        t4 = time.perf_counter()
        scipy_entry = {"value": nnz * width / (t4 - t3), "cores": 1, "seconds": t4 - t3,
                       "kind": "scipy.sparse csr_matrix @ ndarray"}
    except ImportError:
        pass
    model = ""
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return {
        "value": nnz * width / (t1 - t0), "unit": "nnz*F/s", "cores": 1, "kind": "port",
        "sample": "rows [0,%d) of the same A_hat (nnz %d) x the same [N,%d] fp32 operand, "
                  "nonzeros walked in row-major tuple order, single thread" % (rows, nnz, width),
        "seconds": t1 - t0,
        "all_cores": {"value": nnz * width / (t2 - t1), "cores": oracle.max_threads(),
                      "seconds": t2 - t1, "kind": "port (OpenMP over rows)"},
        "scipy": scipy_entry, "host_cpu": model, "host_cpu_count": os.cpu_count(),
    }


def cpu_epoch_baseline(scale, args, dev):
    """The oracle's restatement of one full training epoch of the reference's
    two-layer GCN (forward, masked loss, backward, Adam; TensorFlow CPU order,
    one thread) timed on a small RMAT instance of the benchmark's shape, next to
    the GPU engine's step on the very same instance and inputs."""
    import numpy as np
    import torch
    import oracle
    from oracle import model as omodel
    from gcn_amd import engine, ops, rmat
    oracle.build()
    n, f, h, c = 1 << scale, args.features, args.hidden, args.classes
    a = rmat.rmat_ahat(scale, 8, 123, dev)
    x = ops.philox_uniform((n, f), -1.0, 1.0, 124, 0, dev)
    r1, r2 = math.sqrt(6.0 / (f + h)), math.sqrt(6.0 / (h + c))
    w1 = ops.philox_uniform((f, h), -r1, r1, 125, 0, dev)
    w2 = ops.philox_uniform((h, c), -r2, r2, 125, 1 << 40, dev)
    cls = ops.philox_uniform((n,), 0.0, float(c), 126, 0, dev).to(torch.int64).clamp_(0, c - 1)
    labels = torch.zeros((n, c), dtype=torch.float32, device=dev)
    labels.scatter_(1, cls.view(-1, 1), 1.0)
    lmask = (torch.arange(n, device=dev) % args.label_every == 0).to(torch.uint8)
    total = int(lmask.sum().item())
    eng = engine.FullBatchGCN(a, x, labels, lmask, total, h, c, w1, w2, seed=123)
    first = eng.train_step().cpu().tolist()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 20
    for _ in range(reps):
        eng.train_step()
    torch.cuda.synchronize()
    gpu_ms = (time.perf_counter() - t0) / reps * 1e3
    # the same first step on the host: same inputs, weights and Philox masks
    rp, ci, v = a.row_ptr.cpu().numpy(), a.col_idx.cpu().numpy(), a.val.cpu().numpy()
    rows = np.repeat(np.arange(n, dtype=np.int32), np.diff(rp))
    sup = (np.stack([rows, ci], 1), v, (n, n))
    xr, xc = np.meshgrid(np.arange(n, dtype=np.int32), np.arange(f, dtype=np.int32), indexing="ij")
    xt = (np.stack([xr.ravel(), xc.ravel()], 1), x.cpu().numpy().ravel().copy(), (n, f))
    orc = omodel.TwoLayerOracle("gcn", xt, [sup], [[w1.cpu().numpy()], [w2.cpu().numpy()]], lr=0.01, weight_decay=5e-4)
    m1 = oracle.bernoulli_mask(n * f, 0.5, 123 * 2 + 1, 1 << 44)
    m2 = oracle.bernoulli_mask(n * h, 0.5, 123 * 2 + 2, 1 << 44)
    reg = 0.5 * 5e-4 * float((w1.double() ** 2).sum().item())
    yl, ml = labels.cpu().numpy(), lmask.cpu().numpy()
    t0 = time.perf_counter()
    loss, acc = orc.train_step(yl, ml, m1, m2, 0.5)
    cpu_s = time.perf_counter() - t0
    return {
        "instance": "RMAT scale %d (2^%d nodes, nnz %d), F=%d, hidden %d, %d classes, dropout 0.5, one full-batch "
                    "training epoch (forward, masked loss, backward, Adam)" % (scale, scale, a.nnz, f, h, c),
        "cpu_ms": cpu_s * 1e3, "cores": 1, "kind": "port",
        "what": "oracle.model.TwoLayerOracle.train_step: sparse products in tuple order as TensorFlow's CPU kernel "
                "walks them, dense products as plain fp32 loops, one thread; mask generation not timed",
        "gpu_ms": gpu_ms, "gpu_what": "gcn_amd.engine.FullBatchGCN.train_step on the same instance (launch-bound at "
                                      "this size), mean of %d steps" % reps,
        "first_step_loss_cpu": loss, "first_step_loss_gpu": first[0] + reg,
        "first_step_loss_agrees": bool(abs(loss - (first[0] + reg)) <= 1e-5 * abs(loss)),
    }


def planetoid_epochs(epochs):
    """Side measurement for the small-graph configurations: the drop-in path
    (gcn.utils -> gcn.models -> sess.run, each sess.run replayed from a HIP graph)
    trained for `epochs` epochs on Cora (gcn), Pubmed (gcn; real graph, labels and
    splits, shape-matched SYNTHETIC features: the feature blob is not distributed
    with the reference tree, so its accuracy means nothing) and Citeseer
    (gcn_cheby, K = 3), with the reference's flags (dropout 0.5, hidden 16, early
    stopping after 10 epochs without improvement).  Epoch = one training step + one
    validation forward, as the reference's loop runs them (gcn/train_normal.py:79-100)."""
    from gcn_amd import tfcompat as tf, train, utils
    train.define_flags()
    F = tf.app.flags.FLAGS
    F._parse([])
    F.hip_graph = True
    out = []
    for ds, model in (("cora", "gcn"), ("pubmed", "gcn"), ("citeseer", "gcn_cheby")):
        F.dataset, F.model, F.epochs = ds, model, epochs
        data = utils.load_pubmed_synthetic_features() if ds == "pubmed" else utils.load_data(ds)
        t0 = time.perf_counter()
        res = train.run(data=data, verbose=False)
        wall = time.perf_counter() - t0
        ms = sorted(r[4] * 1e3 for r in res["trace"][5:]) or [float("nan")]
        out.append({"dataset": ds, "model": model, "features": "synthetic" if ds == "pubmed" else "real",
                    "epochs": res["epochs_run"], "epoch_ms_median": ms[len(ms) // 2], "epoch_ms_min": ms[0],
                    "test_acc": res["test_acc"] if ds != "pubmed" else None, "hip_graph": True,
                    "wall_s_with_setup": wall})
        tf.reset_default_graph()
    return out


def one_gpu_denominator(args, shard_scale):
    """The single-GPU run a multi-GPU line's speed-up refers to: the same shard shape
    (2^shard_scale rows, same degree, widths and arithmetic) alone on one GPU,
    `python bench.py --scale S --edge-factor E`.  Its recorded line is looked up
    under profiles/ (newest round first); nothing is measured here."""
    want = "RMAT scale %d (2^%d nodes), avg degree %d, F=%d, hidden %d, %d classes" % (
        shard_scale, shard_scale, 2 * args.edge_factor, args.features, args.hidden, args.classes)
    out = {"command": "python bench.py --scale %d --edge-factor %d" % (shard_scale, args.edge_factor),
           "workload": want, "recorded": None}
    pdir = os.path.join(ROOT, "profiles")
    try:
        names = sorted((f for f in os.listdir(pdir) if f.endswith(".json") and "bench_line" in f), reverse=True)
    except OSError:
        names = []
    for name in names:
        try:
            with open(os.path.join(pdir, name)) as fh:
                rec = json.load(fh)
        except (OSError, ValueError):
            continue
        cfg = rec.get("config", {}) if isinstance(rec, dict) else {}
        if rec.get("n_gpus") == 1 and str(cfg.get("workload", "")).startswith(want) and \
                cfg.get("field") is None and "profiled_run" not in name and \
                cfg.get("order", "reference") == args.order and \
                cfg.get("dense_arith", "f32") == args.dense_arith and rec.get("value"):
            out["recorded"] = {"file": "profiles/" + name, "value": rec["value"], "unit": rec.get("unit"),
                               "ms_per_step": rec.get("ms_per_step")}
            break
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--scale", type=int, default=0, help="log2 of the node count (default 24 + log2 gpus)")
    ap.add_argument("--edge-factor", type=int, default=0,
                    help="undirected edges per node (avg degree = 2x).  Default: 8 on one GPU (RMAT 2^24, degree 16: "
                         "the single-GPU headline configuration), 16 on several (degree 32: RMAT 2^27 at 8 GPUs, the "
                         "row-partitioned configuration)")
    ap.add_argument("--features", type=int, default=256)
    ap.add_argument("--hidden", type=int, default=256)
    ap.add_argument("--classes", type=int, default=16)
    ap.add_argument("--cpu-rows", type=int, default=1 << 22, help="rows of the CPU-baseline slice (0 = skip)")
    ap.add_argument("--slab-cols", type=int, default=0)
    ap.add_argument("--verbose", action="store_true", help="progress lines on stderr")
    ap.add_argument("--hang-dump", type=int, default=0,
                    help="with --verbose: dump all Python stacks and exit after this many seconds")
    ap.add_argument("--halo-mode", default="auto", choices=["auto", "exchange", "recompute"],
                    help="several GPUs: send the hidden-width halo rows, or recompute them where they are needed")
    ap.add_argument("--order", default="reference", choices=["reference", "aggregate_first"],
                    help="first layer as A_hat (X' W1) like the reference (default) or as (A_hat X') W1")
    ap.add_argument("--backend", default="nccl", help="nccl (= RCCL; default) or gloo (rehearsal: several "
                    "ranks sharing one GPU)")
    ap.add_argument("--dense-arith", default="split_bf16", choices=["split_bf16", "f32"],
                    help="large dense products and the factored SpMM's row-forming: exactly split fp32 operands on "
                         "the bf16 matrix pipe (fp32 accuracy; default) or fp32-input MFMA chains")
    ap.add_argument("--exchange", default="cabi", choices=["cabi", "torch"],
                    help="several GPUs over RCCL: per-step halo exchange / all-reduce through the C ABI "
                         "(gcnx_halo_allgather_f32, default) or through torch.distributed")
    ap.add_argument("--label-every", type=int, default=100,
                    help="every how many-th node is labelled (default 100: 1 %% of the nodes)")
    ap.add_argument("--f32-side-steps", type=int, default=3,
                    help="one GPU, split_bf16 arithmetic: after the timed steps, build the engine once more with every "
                         "dense product on exact fp32 MFMA chains and time this many steps as a side measurement "
                         "(ms_per_step_f32_chains; 0 = skip)")
    ap.add_argument("--cpu-epoch-scale", type=int, default=18,
                    help="one GPU: log2 of the node count of the small RMAT instance on which the oracle's full "
                         "training epoch is timed on the host next to the GPU's (cpu_baseline.epoch; 0 = skip)")
    ap.add_argument("--planetoid-epochs", type=int, default=200,
                    help="one GPU: also train the drop-in path on Cora / Pubmed (gcn) and Citeseer (gcn_cheby) for "
                         "this many epochs and report their epoch times (planetoid; 0 = skip)")
    ap.add_argument("--dense-side-steps", type=int, default=3,
                    help="after the timed steps, time this many steps with the dense backward products as a side "
                         "measurement (ms_per_step_dense_gradients; 0 = skip)")
    ap.add_argument("--blocking-class-exchange", action="store_true",
                    help="several GPUs: form the class-width forward product in one piece behind a blocking halo "
                         "exchange (row sums in column order as on one GPU: results bitwise independent of the GPU "
                         "count); default: local columns during the exchange, halo columns after it")
    ap.add_argument("--labelled-field", action="store_true",
                    help="one GPU: the timed training step forms only the rows the masked loss depends on (the "
                         "labelled rows of Z, the rows of H' they read, the rows of P1 those read; same loss, accuracy "
                         "and weight update).  Default: every row of every activation, as the reference's step forms "
                         "them; the field step is then timed as a side measurement (ms_per_step_labelled_field)")
    ap.add_argument("--dense-loss-gradient", action="store_true",
                    help="form A_hat^T dZ over every nonzero, as the reference does; default: over the nonzeros of the "
                         "labelled columns only (the other rows of dZ are exactly zero; same result)")
    args = ap.parse_args()
    if args.label_every < 1:
        raise SystemExit("--label-every must be at least 1")
    if args.edge_factor <= 0:
        args.edge_factor = 8 if args.gpus == 1 else 16

    t_start = time.time()
    import torch
    import torch.distributed as dist
    from gcn_amd import engine, ops, rmat

    ops.dense_arith(0 if args.dense_arith == "split_bf16" else 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (args.gpus, world))
    if world & (world - 1):
        raise SystemExit("the number of GPUs must be a power of two")
    dev = torch.device("cuda", local_rank % max(1, torch.cuda.device_count()))
    torch.cuda.set_device(dev)
    group = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)
        group = dist.group.WORLD

    def note(msg):
        if args.verbose:
            print("[rank %d %.1fs] %s" % (rank, time.time() - t_start, msg), file=sys.stderr, flush=True)

    if args.verbose and args.hang_dump:
        import faulthandler
        faulthandler.dump_traceback_later(args.hang_dump, exit=True)
    note("process group up (backend %s)" % (args.backend if world > 1 else "none"))
    scale = args.scale or (24 + int(math.log2(world)))
    n = 1 << scale
    n_loc = n // world
    rb, re_ = rank * n_loc, (rank + 1) * n_loc
    seed = 123

    # ---- inputs, generated on the device (synthetic; the reference's seed) ----
    def gather_degrees(local):
        if world == 1:
            return local
        if args.backend != "nccl":
            host = torch.empty(n, dtype=torch.int32)
            dist.all_gather_into_tensor(host, local.cpu().contiguous(), group=group)
            return host.to(dev)
        out = torch.empty(n, dtype=torch.int32, device=dev)
        dist.all_gather_into_tensor(out, local.contiguous(), group=group)
        return out

    t_build = time.time()
    a = rmat.rmat_ahat(scale, args.edge_factor, seed, dev, rb, re_, gather_degrees if world > 1 else None)
    note("graph shard built: %d rows, %d nonzeros" % (a.n_rows, a.nnz))
    nnz_t = torch.tensor([a.nnz], dtype=torch.int64, device=dev if args.backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(nnz_t, group=group)
    nnz_global = int(nnz_t.item())
    # features, weights, labels: pure functions of the GLOBAL element index
    x = ops.philox_uniform((n_loc, args.features), -1.0, 1.0, 124, rb * args.features, dev)
    r1 = math.sqrt(6.0 / (args.features + args.hidden))
    r2 = math.sqrt(6.0 / (args.hidden + args.classes))
    w1 = ops.philox_uniform((args.features, args.hidden), -r1, r1, 125, 0, dev)
    w2 = ops.philox_uniform((args.hidden, args.classes), -r2, r2, 125, 1 << 40, dev)
    u = ops.philox_uniform((n_loc,), 0.0, float(args.classes), 126, rb, dev)
    cls = u.to(torch.int64).clamp_(0, args.classes - 1)
    labels = torch.zeros((n_loc, args.classes), dtype=torch.float32, device=dev)
    labels.scatter_(1, cls.view(-1, 1), 1.0)
    gid = torch.arange(rb, re_, device=dev)
    lmask = (gid % args.label_every == 0).to(torch.uint8)      # default: 1 % of the nodes are labelled
    mask_total = (n + args.label_every - 1) // args.label_every
    del cls, u, gid

    engine_cls = engine.FullBatchGCN
    if world > 1 and args.backend != "nccl":
        class HostStagedGather(engine.FullBatchGCN):
            """Rehearsal only (gloo moves host memory): stage collectives through the host."""

            def _all_to_all_start(self, recv, send, recv_counts, send_counts):
                host = torch.empty(recv.shape, dtype=recv.dtype)
                dist.all_to_all_single(host, send.cpu(), output_split_sizes=recv_counts,
                                       input_split_sizes=send_counts, group=self.group)
                recv.copy_(host)
                return None

            def _allreduce(self, t):
                host = t.cpu()
                dist.all_reduce(host, group=self.group)
                t.copy_(host)
                return t
        engine_cls = HostStagedGather

    comm = None
    if world > 1 and args.backend == "nccl" and args.exchange == "cabi":
        # the per-step exchanges and all-reduces run inside libgcnhip.so (grouped
        # ncclSend / ncclRecv on its own communicator); torch.distributed only
        # carries the communicator id and the set-up lists
        from gcn_amd.comm import RcclComm
        ok, why = 1, ""
        try:
            # (the constructor agrees among the ranks before it creates anything: if RCCL
            # cannot be loaded or the id cannot be made anywhere, it raises on every rank)
            comm = RcclComm(group, world, rank, dev)
            # self-check before anything is timed: known answers through the same entry points
            good, why = comm.self_check()
            ok = 1 if good else 0
        except Exception as exc:       # RCCL not loadable, communicator not creatable, ...
            ok, why = 0, "%s: %s" % (type(exc).__name__, exc)
        agree = torch.tensor([ok], dtype=torch.int32, device=dev)
        dist.all_reduce(agree, op=dist.ReduceOp.MIN, group=group)
        if int(agree.item()) == 0:
            # every rank falls back together: the per-step collectives go through torch.distributed
            print("[rank %d] C-ABI collectives unavailable (%s); using torch.distributed" % (rank, why or "a peer failed"),
                  file=sys.stderr, flush=True)
            if comm is not None:
                comm.close()
            comm = None
        else:
            note("C-ABI communicator up, self-check passed")
    def make_engine():
        return engine_cls(a, x, labels, lmask, mask_total, args.hidden, args.classes, w1, w2,
                          row_begin=rb, n_global=n, seed=seed, group=group, world_size=world, rank=rank,
                          slab_cols=args.slab_cols or None, halo_mode=args.halo_mode, order=args.order, comm=comm,
                          sparse_loss_gradient=not args.dense_loss_gradient,
                          overlap_class_exchange=not args.blocking_class_exchange,
                          labelled_field_only=args.labelled_field)

    model = make_engine()
    nnz_dz = torch.tensor([model.loss_gradient_nnz(), model.backward_nnz()], dtype=torch.int64,
                          device=dev if args.backend == "nccl" else "cpu")
    if world > 1:
        # the engine's set-up went through the library's communicator: nothing of it may
        # still be in flight when torch.distributed's communicator carries this sum (two
        # RCCL communicators of one process never run side by side here; every
        # torch.distributed collective below is bracketed by a device synchronisation)
        torch.cuda.synchronize()
        dist.all_reduce(nnz_dz, group=group)
    nnz_dz_global, nnz_bwd_global = (int(v) for v in nnz_dz.tolist())
    torch.cuda.synchronize()
    t_build = time.time() - t_build

    # ---- time the dominant kernel alone, inside the timed region -------------
    n_ev = 2 * args.steps
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_ev)]
    for s, e in ev:
        s.record()
        e.record()
    ev_used = []
    gemm_ev = []
    real_spmm = ops.spmm

    class Timed(object):
        """ops facade: wide SpMM launches of the timed steps get an event pair."""
        armed = False

        def __getattr__(self, name):
            return getattr(ops, name)

        def spmm(self, aa, b, **kw):
            if Timed.armed and b.shape[1] == (args.hidden if args.order == "reference" else args.features) \
                    and len(ev_used) < n_ev:
                s, e = ev[len(ev_used)]
                ops.profile_spmm_events(s, e)
                out = real_spmm(aa, b, **kw)
                ops.profile_spmm_events(None, None)
                ev_used.append((s, e, int(aa.nnz), int(aa.n_rows)))
                return out
            return real_spmm(aa, b, **kw)

        def gemm(self, x, y, **kw):
            # the two 2*N*F*hidden-flop products of the step (X' W1 and X'^T dP1)
            # get an event pair each (they run on torch's current stream)
            # (the second one sums over the rows of dP1 that can be nonzero: x then holds those rows)
            ta = bool(kw.get("trans_a"))
            big = x.shape[1] == args.features and y.shape[-1] == args.hidden and \
                ((not ta and x.shape[0] >= n_loc // 8 and y.shape[0] == args.features) or
                 (ta and y.shape[0] == x.shape[0] and x.shape[0] >= n_loc // 8))
            if Timed.armed and big and args.order == "reference" and len(gemm_ev) < n_ev:
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                out = ops.gemm(x, y, **kw)
                e.record()
                gemm_ev.append((s, e, "tn" if ta else "nn", int(x.shape[0])))
                return out
            return ops.gemm(x, y, **kw)

    model.k = Timed()

    def barrier():
        if world > 1:
            # drain this rank's streams first: the library's communicator and
            # torch's are two RCCL communicators in one process, and nothing of
            # the first should still be in flight when the second one's barrier goes out
            torch.cuda.synchronize()
            dist.barrier(group=group)
        torch.cuda.synchronize()

    note("engine ready")
    for _ in range(args.warmup):
        model.train_step()
        note("warm-up step done")
    barrier()
    Timed.armed = True
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = model.train_step()
    barrier()
    dt = time.perf_counter() - t0
    Timed.armed = False
    note("timed steps done")
    t = torch.tensor([dt], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    dt = float(t.item())
    loss_acc = out.cpu().tolist()

    # Side measurement, outside the timed region and not part of `value`: the same
    # engine with every backward product over every nonzero and row (what
    # --dense-loss-gradient times as the main number), so that both forms come from one run.
    dense_ms = every_row_ms = field_ms = None
    field = model.field
    describe = lambda fd: {"labelled_rows": fd.n_l, "rows_of_H": fd.n1, "rows_of_P1": fd.n2,
                           "nnz_hidden_width": fd.nnz_h, "nnz_class_width": fd.nnz_z}
    field_info = describe(field) if field is not None else None
    field_side = None

    def side_steps(m):
        m.train_step()
        barrier()
        t1 = time.perf_counter()
        for _ in range(args.dense_side_steps):
            m.train_step()
        barrier()
        return (time.perf_counter() - t1) / args.dense_side_steps * 1e3

    if args.dense_side_steps > 0:
        if field is not None:
            # the same engine forming every row of every activation (backward shortcuts kept)
            model.field = None
            every_row_ms = side_steps(model)
        elif world == 1 and model.use_labelled_field():
            # the same engine forming only the rows the masked loss depends on (an engine option)
            field_side = describe(model.field)
            field_ms = side_steps(model)
            model.field = None
    if not args.dense_loss_gradient and model.at_dz is not None and args.dense_side_steps > 0:
        model.use_dense_gradients()
        model.train_step()
        barrier()
        t1 = time.perf_counter()
        for _ in range(args.dense_side_steps):
            model.train_step()
        barrier()
        td = torch.tensor([time.perf_counter() - t1], dtype=torch.float64,
                          device=dev if args.backend == "nccl" else "cpu")
        if world > 1:
            dist.all_reduce(td, op=dist.ReduceOp.MAX, group=group)
        dense_ms = float(td.item()) / args.dense_side_steps * 1e3

    # Second side measurement (one GPU): the same step with every dense product and the
    # factored SpMM's row-forming on exact fp32 MFMA chains instead of the exactly split
    # bf16 operands -- a second engine built under that arithmetic (its set-up asks the
    # library which fused kernels that arithmetic serves), timed with and without the
    # backward shortcuts.
    f32_ms = f32_dense_ms = None
    if world == 1 and args.dense_arith == "split_bf16" and args.f32_side_steps > 0:
        ops.dense_arith(1)
        model_f = None
        try:
            model_f = make_engine()

            def timed_steps(m, k):
                m.train_step()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(k):
                    m.train_step()
                torch.cuda.synchronize()
                return (time.perf_counter() - t1) / k * 1e3

            f32_ms = timed_steps(model_f, args.f32_side_steps)
            if not args.dense_loss_gradient and model_f.at_dz is not None:
                model_f.use_dense_gradients()
                f32_dense_ms = timed_steps(model_f, args.f32_side_steps)
        except torch.cuda.OutOfMemoryError as exc:
            # a side measurement: a graph whose second engine does not fit next to the first
            # keeps its line (the two figures stay null)
            print("[bench] fp32-chain side measurement skipped: %s" % (str(exc).splitlines()[0],), file=sys.stderr,
                  flush=True)
        finally:
            ops.dense_arith(0)
            model_f = None
        torch.cuda.empty_cache()
        note("fp32-chain side measurement done")

    # Third side measurement (one GPU, reference order timed above): the same step with the
    # first layer associated the other way, (A_hat X') W1 -- an engine option whose results
    # agree with the reference's association to rounding, not to the bit.
    agg_ms = None
    if world == 1 and args.order == "reference" and args.dense_side_steps > 0:
        model_a = None
        try:
            model_a = engine_cls(a, x, labels, lmask, mask_total, args.hidden, args.classes, w1, w2, row_begin=rb,
                                 n_global=n, seed=seed, order="aggregate_first",
                                 sparse_loss_gradient=not args.dense_loss_gradient, labelled_field_only=True)
            model_a.train_step()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.dense_side_steps):
                model_a.train_step()
            torch.cuda.synchronize()
            agg_ms = (time.perf_counter() - t1) / args.dense_side_steps * 1e3
        except torch.cuda.OutOfMemoryError as exc:
            print("[bench] aggregate-first side measurement skipped: %s" % (str(exc).splitlines()[0],), file=sys.stderr,
                  flush=True)
        finally:
            model_a = None
        torch.cuda.empty_cache()
        note("aggregate-first side measurement done")

    if rank == 0:
        # work units: the nonzeros each SpMM multiplies (A_hat^T dZ: those of the labelled columns)
        model.field = field          # (the work of the timed steps)
        units = model.spmm_units_per_step(nnz_global, nnz_dz_global, nnz_bwd_global)
        model.field = None
        result = {
            "metric": "GCN SpMM nnz*F/s (full-batch epoch: fwd + bwd + Adam)",
            "value": units * args.steps / dt,
            "unit": "nnz*F/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": "RMAT scale %d (2^%d nodes), avg degree %d, F=%d, hidden %d, %d classes, "
                            "2-layer GCN, dropout 0.5, Adam" % (scale, scale, 2 * args.edge_factor,
                                                                args.features, args.hidden, args.classes),
                "labelled_nodes": mask_total, "nnz_a_hat": nnz_global, "rows_per_gpu": n_loc, "parallelism": "row-partition x%d" % world,
                "halo_rows_rank0": (model.plan.n_halo if model.plan is not None else 0),
                "halo_mode": ("none" if world == 1 else "recompute" if model.recompute else "exchange"),
                "exchange": ("none" if world == 1 else "gcnx_halo_allgather_f32 (RCCL inside libgcnhip.so)"
                             if comm is not None else "torch.distributed " + args.backend),
                "class_width_exchange": ("none" if world == 1 else "overlapped (local columns first)"
                                         if model.fwd_local is not None else "blocking"),
                "order": args.order, "dense_arith": args.dense_arith,
                "nnz_loss_gradient_product": nnz_dz_global, "nnz_backward_product": nnz_bwd_global,
                "gather_hint": bool(getattr(model.a if world == 1 else model.plan.csr, "gather_keep", None) is not None),
                "rows_formed": ("every row of every activation, as the reference's step forms them" if field_info is None
                                else "only what the masked loss reads: Z at the labelled rows, H' at the rows those "
                                     "read, P1 at the rows those read (same loss, accuracy and weight update)"),
                "field": field_info,
                "spmm_per_step": ("2 x width %d over the %d nonzeros of A_hat[rows of H', rows of P1] + 2 x width %d "
                                  "over the %d nonzeros of A_hat[labelled rows, rows of H'], counted as such in value"
                                  % (args.hidden, field_info["nnz_hidden_width"], args.classes,
                                     field_info["nnz_class_width"])) if field_info is not None else
                                 (("2 x width %d + 2 x width %d" % (args.hidden, args.classes)
                                   if args.order == "reference" else
                                   "1 x width %d + 2 x width %d" % (args.features, args.classes)) +
                                  ("" if nnz_dz_global == nnz_global else
                                   "; A_hat^T dZ over the %d nonzeros of the labelled columns and A_hat^T d(A P1) "
                                   "over the %d nonzeros of the columns with a labelled neighbour (the gradients "
                                   "are exactly zero elsewhere), counted as such in value"
                                   % (nnz_dz_global, nnz_bwd_global))),
            },
            "epoch_ms": dt / args.steps * 1e3,
            "ms_per_step_every_row": every_row_ms,
            "ms_per_step_labelled_field": field_ms,
            "labelled_field": field_side,
            "ms_per_step_dense_gradients": dense_ms,
            "ms_per_step_f32_chains": f32_ms,
            "ms_per_step_f32_chains_dense_gradients": f32_dense_ms,
            "ms_per_step_labelled_field_aggregate_first": agg_ms,
            "side_measurements": "after the timed region, a few steps each, not part of value: _dense_gradients = every "
                                 "product over every nonzero and row, as the reference multiplies; _labelled_field = "
                                 "the engine option that forms only the rows the masked loss reads (Z at the labelled "
                                 "rows, H' at the rows those read, P1 at the rows those read: same loss, accuracy and "
                                 "weights, `labelled_field` gives the row counts); _labelled_field_aggregate_first = "
                                 "that option with the first layer as (A_hat X') W1 (one hidden-width SpMM, no product "
                                 "over the rows of P1 but that one); _every_row = with --labelled-field, the step "
                                 "that forms every row; _f32_chains "
                                 "= a second engine with every dense product on exact fp32 MFMA chains",
            "final_xent": loss_acc[0], "final_train_acc": loss_acc[1],
            "build_seconds": t_build,
        }
        if ev_used:
            ms = [s.elapsed_time(e) for s, e, _, _ in ev_used]
            avg = sum(ms) / len(ms) * 1e-3
            k_nnz, k_rows = ev_used[0][2], ev_used[0][3]          # the matrix the timed launches multiplied
            bytes_ = algorithmic_bytes(k_nnz, k_rows, args.hidden)
            # HBM traffic of this kernel comes from rocprofv3 --pmc passes (FETCH_SIZE,
            # WRITE_SIZE; separate runs of tools/spmm_bench.py on the same matrix and
            # width), folded into bytes in profiles/spmm_hbm_traffic.json.  It is NOT
            # measured in this run: the counters need the profiler attached.
            traffic, traffic_source, l2_hit, physical_note = None, None, None, None
            try:
                with open(os.path.join(ROOT, "profiles", "spmm_hbm_traffic.json")) as fh:
                    rec = json.load(fh)
                if field_info is not None:
                    rec = rec.get("labelled_field", {})
                if rec.get("scale") == scale and rec.get("width") == args.hidden and world == 1 and \
                        rec.get("edge_factor", 8) == args.edge_factor and rec.get("nnz", k_nnz) == k_nnz:
                    traffic = rec.get("hbm_bytes_per_launch")
                    traffic_source = "static: profiles/spmm_hbm_traffic.json <- " + str(rec.get("source"))
                    l2_hit = rec.get("l2_hit_rate")
                    if rec.get("hbm_bytes_physical_note"):
                        physical_note = ("no counter on this box separates HBM service from Infinity-Cache service "
                                         "(TCC_EA0_RDREQ_DRAM == TCC_EA0_RDREQ for this kernel); details in "
                                         "profiles/spmm_hbm_traffic.json")
            except (OSError, ValueError):
                pass
            result["roofline"] = {
                "bound": "hbm",
                "kernel": "spmm_chunk_kernel<4,8> + its fix-up pass spmm_fixup_kernel<4> (%s x [%s,%d] fp32)"
                          % ((("A_hat", "N") if field_info is None else
                              ("A_hat[rows of H', rows of P1]", "rows of P1")) + (args.hidden,)),
                "matrix_rows": k_rows, "matrix_nnz": k_nnz,
                "achieved": bytes_ / avg / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                "frac": bytes_ / avg / HBM_PEAK, "traffic": traffic, "traffic_source": traffic_source,
                "traffic_is": "bytes leaving the L2s per launch (FETCH_SIZE x 2 + WRITE_SIZE), Infinity-Cache hits included",
                "hbm_bytes_physical": None, "hbm_bytes_physical_note": physical_note, "l2_hit_rate": l2_hit,
                "timed_window": "HIP events on the launch stream around the main pass and the fix-up pass",
                "algorithmic_bytes_per_launch": bytes_, "avg_launch_ms": avg * 1e3,
                "launches_timed": len(ms), "kernel_nnzF_per_s": k_nnz * args.hidden / avg,
                "compulsory_bytes_per_launch": 8 * k_nnz + 4 * (k_rows + (field_info["rows_of_P1"] if field_info else
                                                                         a.n_rows)) * args.hidden + 8 * (k_rows + 1),
            }
        if gemm_ev:
            split = ops.dense_arith() == 0
            dense_peak = MFMA_BF16_PEAK / SPLIT_PARTIAL_PRODUCTS if split else MFMA_F32_PEAK
            per, rows_summed = {}, {}
            for s_, e_, kind, k_rows in gemm_ev:
                per.setdefault(kind, []).append(s_.elapsed_time(e_) * 1e-3)
                rows_summed[kind] = k_rows
            # flops of the products as launched (the weight gradient over the rows it sums)
            flops_of = {kind: 2.0 * rows_summed[kind] * args.features * args.hidden for kind in per}
            result["roofline_dense"] = {
                kind: {"bound": "mfma",
                       "kernel": (("gemm_split_nn_kernel" if split else "gemm_mfma_kernel<256>") + " (dropout(X) W1)"
                                  if kind == "nn" else
                                  ("gemm_split_tn_kernel" if split else "gemm_mfma_tn_kernel<256>") +
                                  " (dropout(X)^T dP1)"),
                       "arith": ("fp32 operands split exactly into 3 bf16 pieces, 6 partial products per product on "
                                 "v_mfma_f32_32x32x16_bf16, fp32 accumulate (fp32 accuracy)" if split else
                                 "v_mfma_f32_32x32x2_f32 chains (exact fp32)"),
                       "achieved": flops_of[kind] / (sum(v) / len(v)) / 1e12, "peak": dense_peak / 1e12,
                       "peak_is": ("dense bf16 MFMA peak / 6 partial products" if split else "dense fp32 MFMA peak"),
                       "fp32_mfma_peak": MFMA_F32_PEAK / 1e12,
                       "unit": "TFLOP/s (fp32-product flops)", "frac": flops_of[kind] / (sum(v) / len(v)) / dense_peak,
                       "flops_per_launch": flops_of[kind], "rows_summed_or_formed": rows_summed[kind],
                       "avg_launch_ms": sum(v) / len(v) * 1e3, "launches_timed": len(v)}
                for kind, v in per.items()}
        if args.cpu_rows and world == 1:
            try:
                b_dev = model.p1  # any resident [N, hidden] fp32 operand
                result["cpu_baseline"] = cpu_baseline(a, b_dev, min(args.cpu_rows, a.n_rows), args.hidden)
            except Exception as exc:  # the baseline is a side measurement; never lose the line
                result["cpu_baseline"] = {"value": None, "unit": "nnz*F/s", "cores": 0, "kind": "port",
                                          "sample": "failed: %r" % (exc,)}
            if args.cpu_epoch_scale and "cpu_baseline" in result and result["cpu_baseline"].get("value"):
                try:
                    result["cpu_baseline"]["epoch"] = cpu_epoch_baseline(args.cpu_epoch_scale, args, dev)
                except Exception as exc:
                    result["cpu_baseline"]["epoch"] = {"cpu_ms": None, "failed": repr(exc)}
        if args.planetoid_epochs > 0 and world == 1 and scale >= 20:
            try:
                result["planetoid"] = planetoid_epochs(args.planetoid_epochs)
            except Exception as exc:    # a side measurement: never lose the line
                result["planetoid"] = {"failed": repr(exc)}
        if world > 1:
            # what the node-level speed-up of this configuration is measured against: the
            # same per-GPU shard shape (2^24 rows, this degree) run alone on one GPU
            result["one_gpu_denominator"] = one_gpu_denominator(args, scale - int(math.log2(world)))
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier(group=group)
        if comm is not None:
            comm.close()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
