"""world_size 2, 3, 4 and 8 tests of the row-partitioned engine on the gloo backend (CPU).
The kernels are answered by the oracle (tests/helpers.OracleKernels); what is
under test is the partition, the halo plan and the slabbed halo all-to-all of the dense
operand, the all-reduce of weight gradients / loss, and the rule that dropout masks depend on
the global element index only, so that 1-rank and 2-rank runs agree."""
import os
import socket
import sys

import numpy as np
import pytest
import scipy.sparse as sp
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from helpers import CpuCsr, OracleKernels  # noqa: E402


def _problem(scale=7, f=12, hidden=16, classes=5, label_rate=0.3):
    import oracle
    from gcn_amd import utils
    n = 1 << scale
    adj_i = oracle.rmat_adjacency_plus_identity(scale, 4, 0.57, 0.19, 0.19, 123)
    a_hat = utils.normalize_adj(adj_i).tocsr().astype(np.float32)
    rng = np.random.RandomState(0)
    x = rng.uniform(-1, 1, (n, f)).astype(np.float32)
    w1 = (rng.uniform(-1, 1, (f, hidden)) * 0.3).astype(np.float32)
    w2 = (rng.uniform(-1, 1, (hidden, classes)) * 0.3).astype(np.float32)
    labels = np.eye(classes, dtype=np.float32)[rng.randint(0, classes, n)]
    lmask = (rng.random_sample(n) < label_rate).astype(np.uint8)
    return n, a_hat, x, w1, w2, labels, lmask, hidden, classes


def _run(rank, world, port, slab, out_dir, halo_mode="auto", hidden=16, order="reference", tag="", dropout=0.5,
         classes=5, uneven=False, label_rate=0.3, scale=7, overlap=True):
    from gcn_amd import engine
    if world > 1:
        dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    n, a_hat, x, w1, w2, labels, lmask, hidden, classes = _problem(scale=scale, hidden=hidden, classes=classes,
                                                                   label_rate=label_rate)
    if uneven:
        # row ranges of different sizes (N does not divide by the rank count)
        cuts = [0] + [int(round((p + 1) * n / world + (3 if p % 2 == 0 and p + 1 < world else 0)))
                      for p in range(world)]
        cuts[-1] = n
        rb, re_ = cuts[rank], cuts[rank + 1]
    else:
        n_loc = n // world
        rb, re_ = rank * n_loc, (rank + 1) * n_loc
    T = torch.from_numpy
    eng = engine.FullBatchGCN(
        CpuCsr(a_hat[rb:re_]), T(x[rb:re_].copy()), T(labels[rb:re_].copy()), T(lmask[rb:re_].copy()),
        int(lmask.sum()), hidden, classes, T(w1), T(w2), row_begin=rb, n_global=n, seed=5,
        group=dist.group.WORLD if world > 1 else None, world_size=world, rank=rank, slab_cols=slab,
        kernels=OracleKernels(), halo_mode=halo_mode, order=order, dropout=dropout, overlap_class_exchange=overlap)
    if world > 1:
        assert eng.recompute == (halo_mode == "recompute" and order == "reference")
        # the class-width forward product in two parts (local columns during the exchange,
        # halo columns after it) whenever the class width fits one slab
        assert (eng.fwd_local is not None) == (overlap and eng.plan.n_halo > 0 and classes <= eng.slab)
        if eng.fwd_local is not None:
            assert eng.fwd_local.nnz + eng.fwd_halo.nnz == eng.plan.csr.nnz
            assert int(eng.fwd_local.col_idx.max(initial=0)) < eng.n_loc <= int(eng.fwd_halo.col_idx.min(initial=1 << 30))
        # the halo rows of the loss gradient that travel are exactly the labelled ones
        need = eng.plan_bwd.need.numpy().astype(np.int64)
        assert eng.dz_plan is not None and eng.dz_plan.n_halo == int(lmask[need].sum())
        assert np.array_equal(eng.dz_plan.halo_pos.numpy(), np.nonzero(lmask[need])[0])
        assert sum(eng.dz_plan.recv_counts) == eng.dz_plan.n_halo < eng.plan_bwd.n_halo
    # 16 classes and a hidden width that is a multiple of 64: the backward product
    # with the factored operand (dP2 rows + sign words, local and halo)
    assert eng.factored_bwd == (classes == 16 and hidden % 64 == 0 and order == "reference" and
                                (world == 1 or eng.recompute))
    if eng.factored_bwd:
        # local (and, with the sub-plan exchange, halo) operands of the factored product in 128-byte records
        assert eng.dp2.stride(0) == 32 == eng.hbits.stride(0)
        if world > 1 and eng.hit_plan is not None:
            assert eng.dp2_halo.stride(0) == 32 == eng.hbits_halo.stride(0)
    outs = [eng.train_step().numpy().copy() for _ in range(3)]
    ev = eng.evaluate().numpy().copy()
    np.savez(os.path.join(out_dir, "w%d_r%d%s.npz" % (world, rank, tag)), outs=np.stack(outs), ev=ev,
             w1=eng.w1.numpy(), w2=eng.w2.numpy(), z=eng.z.numpy(),
             compact=np.array([eng.at_dz is not None, eng.at_dz_c is not None, eng.at_wide is not None,
                               eng.at_wide_c is not None]),
             nnz=np.array([eng.loss_gradient_nnz(), eng.backward_nnz()]),
             halo=np.array([eng.plan.n_halo, eng.hit_plan.n_halo if eng.hit_plan is not None else -1]
                           if world > 1 else [0, 0]))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world,slab,halo_mode,hidden,classes", [
    (2, 16, "exchange", 16, 5), (2, 8, "exchange", 16, 5), (4, 4, "exchange", 16, 5),
    # halo rows of the hidden-width products recomputed by the GPU that needs them
    (2, 16, "recompute", 32, 5), (4, 8, "recompute", 64, 5),
    # ... and the backward SpMM with the factored operand over local + halo rows
    (2, 16, "recompute", 64, 16),
    # eight ranks, the count the row-partitioned RMAT-27 configuration runs at
    (8, 8, "exchange", 16, 5), (8, 16, "recompute", 64, 16)])
def test_ranks_match_one_rank(tmp_path, world, slab, halo_mode, hidden, classes):
    out = str(tmp_path)
    _run(0, 1, 0, None, out, "auto", hidden, classes=classes)
    mp.spawn(_run, args=(world, _free_port(), slab, out, halo_mode, hidden, "reference", "", 0.5, classes),
             nprocs=world, join=True)
    one = np.load(os.path.join(out, "w1_r0.npz"))
    r0 = np.load(os.path.join(out, "w%d_r0.npz" % world))
    r1 = np.load(os.path.join(out, "w%d_r1.npz" % world))
    # replicated state is identical on both ranks
    assert np.array_equal(r0["w1"], r1["w1"]) and np.array_equal(r0["w2"], r1["w2"])
    assert np.array_equal(r0["outs"], r1["outs"])
    # and equals the single-rank run up to the association of the two partial sums
    np.testing.assert_allclose(r0["outs"], one["outs"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(r0["ev"], one["ev"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(r0["w1"], one["w1"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(r0["w2"], one["w2"], rtol=0, atol=2e-6)
    z = np.concatenate([np.load(os.path.join(out, "w%d_r%d.npz" % (world, r)))["z"] for r in range(world)])
    np.testing.assert_allclose(z, one["z"], rtol=0, atol=1e-5 * np.abs(one["z"]).max())


@pytest.mark.parametrize("world,hidden,classes,dropout", [(2, 64, 16, 0.5), (2, 32, 5, 0.5), (4, 64, 16, 0.0),
                                                          (8, 64, 16, 0.5)])
def test_few_labels_ranks_match_one_rank(tmp_path, world, hidden, classes, dropout):
    """Few labelled nodes: the backward products run over the rows and columns the
    labelled nodes reach (sub-matrices, rows formed next to each other, the weight
    gradient over those rows with the dropout mask drawn by row id) -- on one rank and
    on several, where the owners tell which halo rows qualify.  Same losses, weights
    and logits; the sub-matrices of the ranks add up to the one of a single rank."""
    out = str(tmp_path)
    _run(0, 1, 0, None, out, "auto", hidden, "reference", "", dropout, classes, False, 0.03, 8)
    mp.spawn(_run, args=(world, _free_port(), 16, out, "recompute", hidden, "reference", "", dropout, classes, False,
                         0.03, 8), nprocs=world, join=True)
    one = np.load(os.path.join(out, "w1_r0.npz"))
    ranks = [np.load(os.path.join(out, "w%d_r%d.npz" % (world, r))) for r in range(world)]
    assert one["compact"].all()                       # every shortcut is in use on the single rank
    assert all(r["compact"][0] and r["compact"][2] for r in ranks)
    assert any(r["compact"][1] for r in ranks) and any(r["compact"][3] for r in ranks)
    assert np.array_equal(sum(r["nnz"] for r in ranks), one["nnz"])
    # of the halo rows of dP2 and of the sign words only those whose gradient can be nonzero travel
    assert all(0 <= r["halo"][1] < r["halo"][0] for r in ranks)
    for r in ranks[1:]:
        assert np.array_equal(ranks[0]["w1"], r["w1"]) and np.array_equal(ranks[0]["outs"], r["outs"])
    np.testing.assert_allclose(ranks[0]["outs"], one["outs"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(ranks[0]["ev"], one["ev"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(ranks[0]["w1"], one["w1"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(ranks[0]["w2"], one["w2"], rtol=0, atol=2e-6)
    z = np.concatenate([r["z"] for r in ranks])
    np.testing.assert_allclose(z, one["z"], rtol=0, atol=1e-5 * np.abs(one["z"]).max())


@pytest.mark.parametrize("world,halo_mode,hidden", [(3, "exchange", 16), (2, "recompute", 32), (3, "recompute", 32)])
def test_uneven_row_ranges(tmp_path, world, halo_mode, hidden):
    """Ranks that own different numbers of rows (three ranks over 128 rows; two
    ranks cut at 67): the halo plan takes the owners' ranges from the ranks
    themselves instead of assuming equal blocks."""
    out = str(tmp_path)
    _run(0, 1, 0, None, out, "auto", hidden)
    mp.spawn(_run, args=(world, _free_port(), 8, out, halo_mode, hidden, "reference", "", 0.5, 5, True),
             nprocs=world, join=True)
    one = np.load(os.path.join(out, "w1_r0.npz"))
    parts = [np.load(os.path.join(out, "w%d_r%d.npz" % (world, r))) for r in range(world)]
    assert len(set(p["z"].shape[0] for p in parts)) > 1          # the split really is uneven
    for p in parts[1:]:
        assert np.array_equal(parts[0]["w1"], p["w1"]) and np.array_equal(parts[0]["outs"], p["outs"])
    np.testing.assert_allclose(parts[0]["outs"], one["outs"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(parts[0]["ev"], one["ev"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(parts[0]["w1"], one["w1"], rtol=0, atol=2e-6)
    z = np.concatenate([p["z"] for p in parts])
    np.testing.assert_allclose(z, one["z"], rtol=0, atol=1e-5 * np.abs(one["z"]).max())


@pytest.mark.parametrize("world", [2, 4])
def test_aggregate_first_order(tmp_path, world):
    """(A_hat X') W1 instead of A_hat (X' W1): ranks agree with one rank, and the
    one-rank run agrees with the reference order to fp32 rounding."""
    out = str(tmp_path)
    _run(0, 1, 0, None, out, "auto", 32, "reference", "_ref")
    _run(0, 1, 0, None, out, "auto", 32, "aggregate_first", "")
    mp.spawn(_run, args=(world, _free_port(), 8, out, "auto", 32, "aggregate_first", ""), nprocs=world, join=True)
    ref = np.load(os.path.join(out, "w1_r0_ref.npz"))
    one = np.load(os.path.join(out, "w1_r0.npz"))
    r0 = np.load(os.path.join(out, "w%d_r0.npz" % world))
    r1 = np.load(os.path.join(out, "w%d_r1.npz" % world))
    assert np.array_equal(r0["w1"], r1["w1"]) and np.array_equal(r0["outs"], r1["outs"])
    np.testing.assert_allclose(r0["outs"], one["outs"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(r0["ev"], one["ev"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(r0["w1"], one["w1"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(r0["w2"], one["w2"], rtol=0, atol=2e-6)
    # against the reference association: same loss to rounding; weights after three
    # Adam steps (each +-lr per element) within a small fraction of one step
    np.testing.assert_allclose(one["outs"], ref["outs"], rtol=2e-5, atol=1e-6)
    np.testing.assert_allclose(one["ev"], ref["ev"], rtol=2e-5, atol=1e-6)
    np.testing.assert_allclose(one["w1"], ref["w1"], rtol=0, atol=1e-3)
    np.testing.assert_allclose(one["w2"], ref["w2"], rtol=0, atol=1e-3)


@pytest.mark.parametrize("halo_mode,order", [("recompute", "reference"), ("auto", "aggregate_first")])
def test_two_ranks_without_dropout(tmp_path, halo_mode, order):
    """keep = 1: no masks anywhere (the recomputed halo rows use the plain
    product, the relu gate still travels as sign bits)."""
    out = str(tmp_path)
    _run(0, 1, 0, None, out, "auto", 32, order, "", 0.0)
    mp.spawn(_run, args=(2, _free_port(), 8, out, halo_mode, 32, order, "", 0.0), nprocs=2, join=True)
    one = np.load(os.path.join(out, "w1_r0.npz"))
    r0 = np.load(os.path.join(out, "w2_r0.npz"))
    np.testing.assert_allclose(r0["outs"], one["outs"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(r0["ev"], one["ev"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(r0["w1"], one["w1"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(r0["w2"], one["w2"], rtol=0, atol=2e-6)


@pytest.mark.parametrize("order", ["reference", "aggregate_first"])
def test_dense_and_shortcut_gradients_agree(order):
    """The backward products over the rows and columns the labelled nodes reach against
    every product over every nonzero and row: same losses, same weights (sums that
    differ by exact zeros, grouped differently); and an engine switched over in
    mid-run (use_dense_gradients) continues as the dense one."""
    from gcn_amd import engine
    n, a_hat, x, w1, w2, labels, lmask, hidden, classes = _problem(scale=8, hidden=64, classes=16, label_rate=0.03)
    T = torch.from_numpy
    mk = lambda sparse: engine.FullBatchGCN(
        CpuCsr(a_hat), T(x.copy()), T(labels.copy()), T(lmask.copy()), int(lmask.sum()), hidden, classes, T(w1), T(w2),
        seed=5, kernels=OracleKernels(), sparse_loss_gradient=sparse, order=order)
    short, dense, mixed = mk(True), mk(False), mk(True)
    assert short.at_dz_c is not None and dense.at_dz is None
    assert (short.at_wide_c is not None) == (order == "reference")
    assert (short.p1_c is not None) == (order == "aggregate_first")
    for step in range(3):
        if step == 1:
            mixed.use_dense_gradients()
            assert mixed.at_dz is None and mixed.at_wide_c is None and mixed.p1_c is None
            assert (mixed.hb is not None) == (order == "reference")
        outs = [e.train_step().numpy().copy() for e in (short, dense, mixed)]
        np.testing.assert_allclose(outs[0], outs[1], rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(outs[2], outs[1], rtol=1e-5, atol=1e-7)
        for e in (short, mixed):
            np.testing.assert_allclose(e.w1.numpy(), dense.w1.numpy(), rtol=0, atol=2e-6)
            np.testing.assert_allclose(e.w2.numpy(), dense.w2.numpy(), rtol=0, atol=2e-6)
            np.testing.assert_allclose(e.g1.numpy(), dense.g1.numpy(), rtol=0, atol=1e-5 * np.abs(dense.g1.numpy()).max())


def test_modified_label_mask_is_refused():
    """The backward sub-matrices are built from the label mask at set-up: a mask
    modified in place afterwards is an error, not a silently wrong gradient."""
    from gcn_amd import engine
    n, a_hat, x, w1, w2, labels, lmask, hidden, classes = _problem(scale=7, label_rate=0.05)
    T = torch.from_numpy
    mask = T(lmask.copy())
    eng = engine.FullBatchGCN(CpuCsr(a_hat), T(x.copy()), T(labels.copy()), mask, int(lmask.sum()), hidden, classes,
                              T(w1), T(w2), seed=5, kernels=OracleKernels())
    eng.train_step()
    mask[3] = 1 - mask[3]
    with pytest.raises(RuntimeError, match="label mask"):
        eng.train_step()


def test_single_rank_matches_model_oracle():
    """The engine's schedule (buffer reuse, fused backward steps) against the
    independent model restatement in oracle/model.py."""
    import oracle
    from oracle import model as omodel
    from gcn_amd import engine
    n, a_hat, x, w1, w2, labels, lmask, hidden, classes = _problem()
    T = torch.from_numpy
    eng = engine.FullBatchGCN(CpuCsr(a_hat), T(x.copy()), T(labels.copy()), T(lmask.copy()), int(lmask.sum()),
                              hidden, classes, T(w1), T(w2), seed=5, kernels=OracleKernels())
    coo = a_hat.tocoo()
    sup = (np.stack([coo.row, coo.col], 1).astype(np.int32), coo.data, (n, n))
    f = x.shape[1]
    r, c = np.meshgrid(np.arange(n, dtype=np.int32), np.arange(f, dtype=np.int32), indexing="ij")
    feats = (np.stack([r.ravel(), c.ravel()], 1), x.ravel().copy(), (n, f))
    orc = omodel.TwoLayerOracle("gcn", feats, [sup], [[w1], [w2]], lr=0.01, weight_decay=5e-4)
    for step in (1, 2):
        off = step * (1 << 44)
        m1 = oracle.bernoulli_mask(n * f, 0.5, 5 * 2 + 1, off)
        m2 = oracle.bernoulli_mask(n * hidden, 0.5, 5 * 2 + 2, off)
        logits, cache = orc.forward(m1, m2, 0.5)
        xent, grad = oracle.masked_softmax_xent(logits, labels, lmask)
        orc.adam(orc.backward(cache, grad))
        got = eng.train_step().numpy()
        assert abs(got[0] - xent) <= 1e-5 * abs(xent)
        np.testing.assert_allclose(eng.w1.numpy(), orc.w[0][0], rtol=0, atol=2e-6)
        np.testing.assert_allclose(eng.w2.numpy(), orc.w[1][0], rtol=0, atol=2e-6)


def test_overlapped_and_blocking_class_exchange_agree(tmp_path):
    """The forward class-width product with its halo exchange hidden (two partial
    products) against the one product behind a blocking exchange: same losses and
    weights to rounding on 2 ranks; the blocking form is the one whose row sums run
    in column order as on one rank."""
    out = str(tmp_path)
    _run(0, 1, 0, None, out, "auto", 64, classes=16)
    for overlap, tag in ((True, "_ov"), (False, "_bl")):
        mp.spawn(_run, args=(2, _free_port(), 16, out, "recompute", 64, "reference", tag, 0.5, 16, False, 0.3, 7, overlap),
                 nprocs=2, join=True)
    one = np.load(os.path.join(out, "w1_r0.npz"))
    ov = np.load(os.path.join(out, "w2_r0_ov.npz"))
    bl = np.load(os.path.join(out, "w2_r0_bl.npz"))
    for got in (ov, bl):
        np.testing.assert_allclose(got["outs"], one["outs"], rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(got["w1"], one["w1"], rtol=0, atol=2e-6)
        np.testing.assert_allclose(got["w2"], one["w2"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(ov["outs"], bl["outs"], rtol=1e-5, atol=1e-7)
    z_ov = np.concatenate([np.load(os.path.join(out, "w2_r%d_ov.npz" % r))["z"] for r in range(2)])
    z_bl = np.concatenate([np.load(os.path.join(out, "w2_r%d_bl.npz" % r))["z"] for r in range(2)])
    np.testing.assert_allclose(z_ov, one["z"], rtol=0, atol=1e-5 * np.abs(one["z"]).max())
    np.testing.assert_allclose(z_bl, one["z"], rtol=0, atol=1e-5 * np.abs(one["z"]).max())


@pytest.mark.parametrize("hidden,classes,dropout,label_rate,scale,no_self_loop", [
    (64, 16, 0.5, 0.02, 9, False),      # factored backward, operands in records, N2 clearly smaller than N
    (64, 16, 0.0, 0.02, 9, False),      # no dropout at all
    (16, 5, 0.5, 0.03, 8, False),       # no sign words at this width: gate read from H'
    (32, 5, 0.5, 0.03, 8, False),       # sign words, bit-gated product then SpMM
    (64, 16, 0.5, 0.08, 8, False),      # N2 is most of the graph: X and dP1 keep their full height
    (64, 16, 0.5, 0.03, 8, True),       # a labelled node whose row of A_hat is empty: a zero row of logits
    (64, 16, 0.5, 0.02, 9, "aggregate_first"),    # first layer as (A_hat[N1, N2] X') W1: no product over N2 but the SpMM
    (16, 5, 0.0, 0.03, 8, "aggregate_first"), (64, 16, 0.5, 0.08, 8, "aggregate_first"),
    (64, 16, 0.5, 0.02, 9, "stored-dropout"),     # products that cannot regenerate dropout(X): stored over N2
    (64, 16, 0.5, 0.08, 8, "stored-dropout")])    # ... and over every row
def test_labelled_field_step_matches_full_step(hidden, classes, dropout, label_rate, scale, no_self_loop):
    """The training step over the labelled rows' field (rows of N2 / N1 / the labelled
    set stored next to each other, A_hat[N1, N2] and A_hat[labelled, N1]) against the
    step that forms every row: same losses, accuracies, gradients and weights to
    rounding; the rows it does form equal the same rows of the full step."""
    from gcn_amd import engine
    n, a_hat, x, w1, w2, labels, lmask, hidden, classes = _problem(scale=scale, hidden=hidden, classes=classes,
                                                                   label_rate=label_rate)
    kernels = OracleKernels
    order = "reference"
    if no_self_loop == "aggregate_first":
        order, no_self_loop = "aggregate_first", False
    if no_self_loop == "stored-dropout":
        no_self_loop = False

        class kernels(OracleKernels):
            def gemm_supports_a_dropout(self, trans_a, M, N, K):
                return False
    if no_self_loop:
        a_hat = a_hat.tolil()
        victim = int(np.nonzero(lmask)[0][1])
        a_hat[victim, :] = 0
        a_hat[:, victim] = 0
        a_hat = a_hat.tocsr().astype(np.float32)
        a_hat.eliminate_zeros()
    T = torch.from_numpy
    mk = lambda on: engine.FullBatchGCN(
        CpuCsr(a_hat), T(x.copy()), T(labels.copy()), T(lmask.copy()), int(lmask.sum()), hidden, classes, T(w1), T(w2),
        seed=5, kernels=kernels(), dropout=dropout, labelled_field_only=on, order=order)
    field, full = mk(True), mk(False)
    fd = field.field
    assert fd is not None and full.field is None
    if order == "aggregate_first":
        assert fd.agg and fd.a_ht is None and (fd.xd is not None) == (dropout > 0)
    else:
        assert (fd.xd is not None) == (kernels is not OracleKernels) == (dropout > 0 and not fd.fuse_x)
    lab = np.nonzero(lmask)[0]
    n1_ref = np.unique(a_hat[lab].indices)                       # columns the labelled rows read
    n2_ref = np.unique(a_hat[n1_ref].indices)
    assert np.array_equal(fd.lab_rows.numpy(), lab) and np.array_equal(fd.rows1.numpy(), n1_ref)
    assert fd.n_l == len(lab) and fd.n1 == len(n1_ref)
    if fd.rows2 is not None:
        assert np.array_equal(fd.rows2.numpy(), n2_ref) and fd.n2 == len(n2_ref) < n
    else:
        assert fd.n2 == n and len(n2_ref) > (n * 7) // 10
    assert (fd.rows2 is None) == (label_rate >= 0.08)
    assert fd.nnz_h == a_hat[n1_ref].nnz and fd.nnz_z == a_hat[lab].nnz
    assert fd.factored == (hidden == 64 and classes == 16 and order == "reference")
    if fd.factored:
        assert fd.dp2.stride(0) == 32 == fd.hbits.stride(0)
    for step in range(3):
        o_field, o_full = field.train_step().numpy().copy(), full.train_step().numpy().copy()
        np.testing.assert_allclose(o_field, o_full, rtol=1e-5, atol=1e-7)
        for name in ("g1", "g2"):
            ref = getattr(full, name).numpy()
            np.testing.assert_allclose(getattr(field, name).numpy(), ref, rtol=0, atol=1e-5 * np.abs(ref).max())
        np.testing.assert_allclose(field.w1.numpy(), full.w1.numpy(), rtol=0, atol=2e-6)
        np.testing.assert_allclose(field.w2.numpy(), full.w2.numpy(), rtol=0, atol=2e-6)
        # the rows the field step forms are the same rows of the full step
        np.testing.assert_allclose(fd.z.numpy(), full.z.numpy()[lab], rtol=0, atol=1e-5 * np.abs(full.z.numpy()).max())
        np.testing.assert_allclose(fd.hd.numpy(), full.hd.numpy()[n1_ref], rtol=0,
                                   atol=1e-5 * np.abs(full.hd.numpy()).max())
    if no_self_loop:
        assert not fd.z.numpy()[1].any()
    # evaluation forms every row in either engine
    np.testing.assert_allclose(field.evaluate().numpy(), full.evaluate().numpy(), rtol=1e-5, atol=1e-7)
    # ... and an engine switched to dense gradients leaves the field
    field.use_dense_gradients()
    assert field.field is None
    np.testing.assert_allclose(field.train_step().numpy(), full.train_step().numpy(), rtol=1e-5, atol=1e-7)
