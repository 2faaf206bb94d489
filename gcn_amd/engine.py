"""Full-batch two-layer GCN step for large graphs with DENSE input features,
written out by hand (no autograd tape, buffers allocated once) and
row-partitioned across the GPUs of one node.

It computes exactly what the reference's ``GCN`` model computes for
``sparse_inputs=False`` (gcn/models.py:159-174, gcn/layers.py:203-226,
gcn/metrics.py:4-20, Adam per gcn/models.py:142):

    X' = dropout(X)            P1 = X' W1        H  = relu(A_hat P1)
    H' = dropout(H)            P2 = H' W2        Z  = A_hat P2
    loss = wd * |W1|^2 / 2 + masked softmax cross-entropy(Z)

and the backward products  A_hat^T dZ,  H'^T dP2,  dP2 W2^T,  A_hat^T dH,
X'^T dP1.  With P ranks, rank p owns a contiguous block of rows of A_hat and of
every activation; before each SpMM the rows of the dense operand that the block
references on other ranks (the halo, gcn_amd/halo.py) are fetched with a
variable-size all-to-all -- grouped RCCL sends and receives over xGMI issued by
libgcnhip.so (gcn_amd/comm.py, gcnx_halo_allgather_f32) -- in column slabs on the
communication stream, so that the SpMM of slab s overlaps the transfer of slab
s+1.  For the two hidden-width products the halo rows are not
sent at all but recomputed by the GPU that needs them (``halo_mode``): the
features are constant, so each GPU keeps a copy of the remote feature rows it
references and forms dropout(X) W1 for them itself; in the backward pass only
the class-width rows of dP2 and the sign bits of the remote rows of H' travel
(96 bytes per row instead of 1 KiB at hidden 256) and the gated product
dP2 W2^T is formed locally.  Same operations in the same order on the same
numbers, so the result does not depend on the mode.  Of the halo rows of the
loss gradient dZ only the labelled ones travel (the others are exactly zero and
stay zero in a resident halo buffer), and the product A_hat^T dZ runs over the
nonzeros of the labelled columns only (a sub-matrix built once:
``sparse_loss_gradient``).  Weight gradients and the loss are
all-reduced; Adam runs replicated.

``order="aggregate_first"`` (opt-in) evaluates the first layer as
(A_hat X') W1 instead of A_hat (X' W1) -- the same sum, associated the other
way, so results agree with the reference order to fp32 rounding, not to the
bit.  Because the features need no gradient, the weight gradient becomes
(A_hat X')^T dZ1 with A_hat X' kept from the forward pass: the hidden-width
backward SpMM disappears, and with several GPUs so does every hidden-width
exchange (the remote feature rows are resident).  Dropout masks are a pure function of (seed, step, global element
index), so a run gives the same numbers for any GPU count.
"""
import math

import torch

from . import ops as _hip_ops


class _SubPlan(object):
    """The part of a halo plan an exchange needs (send_idx, per-peer counts), for a
    subset of the plan's rows."""
    __slots__ = ("send_idx", "send_counts", "recv_counts", "n_send", "n_halo", "halo_pos", "send_buf", "recv_buf",
                 "send_bits", "recv_bits")


class FullBatchGCN(object):
    def __init__(self, a_local, x_local, labels_local, mask_local, mask_total, n_hidden, n_class,
                 w1, w2, a_bwd_local=None, row_begin=0, n_global=None, lr=0.01, weight_decay=5e-4,
                 dropout=0.5, seed=123, group=None, world_size=1, rank=0, slab_cols=None,
                 kernels=None, keep_masks=False, halo_mode="auto", order="reference", comm=None,
                 sparse_loss_gradient=True, overlap_class_exchange=True, pack_gate_operands=True,
                 labelled_field_only=False):
        """``a_local``: Csr of the local rows of A_hat (global column ids);
        ``a_bwd_local``: local rows of A_hat^T (defaults to ``a_local``: the
        normalised adjacency of an undirected graph is bitwise symmetric).
        ``w1`` [F, hidden], ``w2`` [hidden, n_class]: initial weights (device
        tensors, identical on every rank).  ``mask_total``: number of labelled
        rows over all ranks.  ``halo_mode``: "exchange" sends the halo rows of
        every dense operand; "recompute" forms the hidden-width halo rows
        locally (see the module text); "auto" picks "recompute" when the
        kernels serve it for these shapes.  ``comm`` (a gcn_amd.comm.RcclComm):
        the per-step exchanges and all-reduces go through the C ABI
        (gcnx_halo_allgather_f32 / gcnx_comm_allreduce_sum_f32, RCCL bound inside
        libgcnhip.so) on the communicator's own stream, and so do the set-up
        exchanges of the halo plan; without it everything goes through
        torch.distributed on ``group`` (the gloo rehearsal).
        ``sparse_loss_gradient``: the rows of dZ outside the labelled set are
        exactly zero (the mask multiplies the per-row loss, gcn/metrics.py:4-11), so
        A_hat^T dZ is formed with the nonzeros of the labelled columns only, and
        A_hat^T d(A P1) with the nonzeros of the columns that have a labelled
        neighbour (the other rows of dP2, hence of d(A P1), are exactly zero) --
        sub-matrices built once, the label mask being fixed; every row of every
        gradient is still written and no sum changes (False: the full matrix,
        zeros multiplied as the reference does).
        ``overlap_class_exchange`` (several ranks): the class-width forward product
        A_hat [P2; halo rows of P2] is formed in two parts -- the nonzeros of the
        local columns while the halo rows travel, the nonzeros of the halo columns
        added once they are in -- so that no exchange of the step blocks the
        compute stream; a row's sum is then grouped (local terms, then halo terms)
        instead of running in column order, so one rank and several agree to fp32
        rounding, not to the bit (False: one product after the exchange, bitwise
        independent of the rank count).
        ``labelled_field_only`` (opt-in; one rank, with ``sparse_loss_gradient``;
        :meth:`use_labelled_field` switches an existing engine over): a training step
        forms only what its loss, its
        accuracy and its weight update depend on.  The masked loss reads the logits
        of the labelled rows; those read H' at the nodes with a labelled neighbour
        (N1); those read P1 at the nodes with a neighbour in N1 (N2).  The step then
        runs on three sets of rows stored next to each other -- X', P1, dP1 over N2;
        H', P2, dP2 over N1; Z, dZ over the labelled rows -- through the matrices
        A_hat[N1, N2] and A_hat[labelled, N1] and their transposes (built once, the
        label mask being fixed).  Every sum that is formed has the terms, in the order,
        of the full computation (dropout masks are drawn by node id), so loss,
        accuracy and weights agree with it to the rounding of where a long row is cut
        into partial sums; the rows left out are the ones no output of the step
        depends on.  ``evaluate()`` still forms every row.  With ``order="aggregate_first"``
        the field step forms (A_hat[N1, N2] X') W1: one sparse product over N2 and every
        dense product over N1.  False (default): every row of every activation, as the
        reference's step forms them."""
        self.k = kernels if kernels is not None else _hip_ops
        self.comm = comm
        self.a = a_local
        self.at = a_bwd_local if a_bwd_local is not None else a_local
        self.x = x_local
        self.y = labels_local
        self.ymask = mask_local
        self.mask_total = float(mask_total)
        self.n_loc, self.f = x_local.shape
        self.h, self.c = int(n_hidden), int(n_class)
        self.n = int(n_global if n_global is not None else a_local.n_cols)
        self.row_begin = int(row_begin)
        self.dev = x_local.device
        self.lr, self.wd, self.keep = lr, weight_decay, 1.0 - dropout
        self.seed = int(seed)
        if order not in ("reference", "aggregate_first"):
            raise ValueError("order must be 'reference' or 'aggregate_first'")
        self.order = order
        agg_first = order == "aggregate_first"
        self.group, self.world = group, int(world_size)
        self.step_no = 0
        self.keep_masks = keep_masks
        self.w1 = w1.clone().contiguous()
        self.w2 = w2.clone().contiguous()
        self.m1, self.v1 = torch.zeros_like(self.w1), torch.zeros_like(self.w1)
        self.m2, self.v2 = torch.zeros_like(self.w2), torch.zeros_like(self.w2)
        e = lambda *s, **kw: torch.empty(s, dtype=kw.get("dtype", torch.float32), device=self.dev)
        nl = self.n_loc
        # activations / gradients of the local rows, allocated once
        # dropout(X) is regenerated inside the two products that read it when
        # their shapes are served by the MFMA kernels; otherwise it is stored
        sup = getattr(self.k, "gemm_supports_a_dropout", None)
        self.fuse_x_dropout = bool(self.keep < 1.0 and not keep_masks and not agg_first and sup is not None and
                                   sup(False, nl, self.h, self.f) and sup(True, self.f, self.h, nl))
        self.xd = None if (self.fuse_x_dropout or self.keep >= 1.0) else e(nl, self.f)   # dropout(X)
        self.p1 = e(nl, self.h)        # X' W1; later the gradient wrt A P1
        self.hb = None if agg_first else e(nl, self.h)        # A^T dH (gradient wrt P1)
        self.ax = e(nl, self.f) if agg_first else None          # A_hat X' (aggregate-first order)
        self.hd = e(nl, self.h)        # H' = dropout(relu(A P1))
        self.p2 = e(nl, self.c)
        self.z = e(nl, self.c)
        self.dz = e(nl, self.c)
        self.dp2 = e(nl, self.c)
        self.g1, self.g2 = torch.empty_like(self.w1), torch.empty_like(self.w2)
        # dropout masks never touch memory (they are Philox functions of the
        # element index); the input mask is stored only on request (keep_masks)
        self.mask1 = e(nl * self.f, dtype=torch.uint8) if (self.keep < 1.0 and keep_masks) else None
        # halo exchange (only with more than one rank)
        self.plan = self.plan_bwd = None
        if self.world > 1:
            from .halo import HaloPlan
            plan_fn = getattr(self.k, "halo_plan", None)
            kw = {"plan_fn": plan_fn} if plan_fn is not None else {}
            self.plan = HaloPlan(a_local, self.row_begin, nl, self.world, rank, self._all_to_all, **kw)
            self.plan_bwd = self.plan if a_bwd_local is None else \
                HaloPlan(a_bwd_local, self.row_begin, nl, self.world, rank, self._all_to_all, **kw)
            self.slab = int(slab_cols) if slab_cols else min(64, self.h)
            wide_by_slabs = not agg_first and not self._recompute_feasible(halo_mode)
            if not wide_by_slabs and not slab_cols:
                # only class-width operands (and, once, the features) travel in slabs:
                # keep the staging buffers small
                self.slab = min(self.slab, max(16, (self.c + 3) // 4 * 4))
            n_halo = max(self.plan.n_halo, self.plan_bwd.n_halo, 1)
            n_send = max(self.plan.n_send, self.plan_bwd.n_send, 1)
            self.halobuf = [e(n_halo * self.slab), e(n_halo * self.slab)]
            self.sendbuf = [e(n_send * self.slab), e(n_send * self.slab)]
            if agg_first:
                self.recompute = False
                nh = max(self.plan.n_halo, 1)
                self.x_halo = e(nh, self.f)                     # resident remote feature rows
                self._exchange_rows(self.plan, self.x, self.x_halo)
                self.xd_halo = e(nh, self.f) if self.keep < 1.0 else None
            else:
                self.recompute = self._setup_recompute(halo_mode)
            self._setup_label_exchange()
            self._setup_class_overlap(overlap_class_exchange)
        else:
            self.slab = None
            self.recompute = False
            self.dz_plan = None
            self.fwd_local = self.fwd_halo = None
        self._setup_loss_gradient_matrix(sparse_loss_gradient)
        self._setup_backward_matrix()
        # the hidden-width forward product gathers 4 h bytes per nonzero from an operand
        # far larger than the caches: tell the kernel which rows are worth keeping there
        self._hint_forward_matrix()
        # sign bits of H' written by the SpMM that produces it: the backward gate
        # then reads 1 bit per element instead of the element
        sup_bits = getattr(self.k, "spmm_supports_sign_bits", None)
        self.hbits = None
        if not agg_first and sup_bits is not None and sup_bits(self.h) and (self.world == 1 or self.recompute):
            self.hbits = torch.empty((nl, self.h // 32), dtype=torch.int32, device=self.dev)
        if agg_first and self.keep < 1.0 and self.h % 32 == 0 and not keep_masks:
            # aggregate-first order: H' leaves the dense product, whose fused
            # relu + dropout epilogue writes the sign words
            sup_o = getattr(self.k, "gemm_supports_output_dropout", None)
            if sup_o is not None and sup_o(False, nl, self.h, self.f):
                self.hbits = torch.empty((nl, self.h // 32), dtype=torch.int32, device=self.dev)
        # with the sign words at hand the hidden-width backward product runs with a
        # factored operand: the rows of d(A P1) are formed inside the SpMM from the
        # class-width rows of dP2 and never stored
        sup_f = getattr(self.k, "spmm_supports_factored", None)
        self.factored_bwd = bool(not agg_first and self.hbits is not None and self.keep > 0.0 and sup_f is not None and
                                 sup_f(self.h, self.c))
        words = self.h // 32
        if self.factored_bwd and pack_gate_operands and self.c % 4 == 0 and 4 * (self.c + words) <= 128:
            # The factored product gathers, per nonzero, the node's row of dP2 and its sign
            # words: kept side by side in one 128-byte record per node, the two gathers of
            # a nonzero fall into one memory line instead of two (both arrays become
            # strided views of the record array; their producers write through the views)
            self._gate_records = torch.zeros((nl, 32), dtype=torch.float32, device=self.dev)
            self.dp2 = self._gate_records[:, :self.c]
            self.hbits = self._gate_records.view(torch.int32)[:, self.c:self.c + words]
            if self.world > 1 and self.recompute and self.hit_plan is not None:
                # the halo rows likewise (they are put in place row by row with
                # gcnx_scatter_rows_f32, which takes any row pitch; the rows that never
                # arrive stay zero)
                nh = max(self.plan.n_halo, 1)
                self._gate_records_halo = torch.zeros((nh, 32), dtype=torch.float32, device=self.dev)
                self.dp2_halo = self._gate_records_halo[:, :self.c]
                self.hbits_halo = self._gate_records_halo.view(torch.int32)[:, self.c:self.c + words]
        self.field = None
        self._pack_gate_operands = bool(pack_gate_operands)
        if labelled_field_only:
            self._setup_labelled_field(pack_gate_operands)

    def _hint_forward_matrix(self):
        width = self.f if self.order == "aggregate_first" else self.h
        if self.world == 1:
            m = self.a
        elif self.order == "aggregate_first" or self.recompute:
            m = self.plan.csr                 # one full-width launch over [local | halo] columns
        else:
            return                            # exchanged in 64-column slabs: the narrow kernel
        hint = getattr(m, "set_gather_hint", None)
        if hint is not None and width > 128 and m.n_cols * width * 4 > (1 << 30) and m.nnz > (1 << 24):
            hint(width * 4)

    # --------------------------------------------------- halo rows by recompute
    def _recompute_feasible(self, halo_mode):
        if halo_mode not in ("auto", "exchange", "recompute"):
            raise ValueError("halo_mode must be auto, exchange or recompute")
        k, plan = self.k, self.plan
        ok = (halo_mode != "exchange" and self.plan_bwd is plan and self.h % 32 == 0 and
              hasattr(k, "pack_sign_bits") and (self.keep >= 1.0 or self.fuse_x_dropout))
        if ok and self.keep < 1.0 and plan.n_halo:
            sup = getattr(k, "gemm_supports_a_dropout", None)
            ok = sup is not None and bool(sup(False, plan.n_halo, self.h, self.f))
        if not ok and halo_mode == "recompute":
            raise ValueError("halo_mode='recompute' is not served for this configuration")
        return ok

    def _setup_recompute(self, halo_mode):
        if not self._recompute_feasible(halo_mode):
            return False
        plan = self.plan
        e = lambda *s, **kw: torch.empty(s, dtype=kw.get("dtype", torch.float32), device=self.dev)
        nh, ns = max(plan.n_halo, 1), max(plan.n_send, 1)
        # resident copy of the remote feature rows this block references (fetched once)
        self.x_halo = e(nh, self.f)
        self._exchange_rows(plan, self.x, self.x_halo)
        self.wide_halo = e(nh, self.h)               # halo rows of P1, later of d(A P1)
        words = self.h // 32
        self.hbits_send = e(ns, words, dtype=torch.int32)
        self.hbits_halo = e(nh, words, dtype=torch.int32)
        self.dp2_send, self.dp2_halo = e(ns, self.c), e(nh, self.c)
        self._bits_work = None
        return True

    def _setup_class_overlap(self, enabled):
        """``self.fwd_local`` / ``self.fwd_halo``: the forward block cut by column into
        the nonzeros that meet local rows of the operand and those that meet halo
        rows (sub-matrices of the same shape, built once), or None."""
        self.fwd_local = self.fwd_halo = None
        filt = getattr(self.k, "csr_filter_columns", None)
        plan = self.plan
        if not enabled or filt is None or plan.n_halo == 0 or self.c > self.slab:
            return
        is_halo = torch.zeros(self.n_loc + plan.n_halo, dtype=torch.uint8, device=self.dev)
        is_halo[self.n_loc:] = 1
        self.fwd_halo = filt(plan.csr, is_halo)
        self.fwd_local = filt(plan.csr, 1 - is_halo)

    def _aggregate_class_overlapped(self, src, dst):
        """dst = A_hat [src; halo rows of src] with the exchange of the halo rows
        hidden behind the local part of the product."""
        plan = self.plan
        w = src.shape[1]
        sb = self.sendbuf[0][:plan.n_send * w].view(plan.n_send, w)
        hb = self.halobuf[0][:plan.n_halo * w].view(plan.n_halo, w)
        work = self._exchange_start(plan, src, 0, w, sb, hb)
        # the local columns' nonzeros never read the halo buffer (it is being written)
        self.k.spmm(self.fwd_local, src, out=dst, b2=hb)
        if work is not None:
            work.wait()
        self.k.spmm(self.fwd_halo, src, out=dst, b2=hb, beta=1.0)

    def _flagged_subplan(self, plan, local_flag):
        """The part of ``plan`` that concerns the flagged rows (``local_flag``: one
        bool per local row): every owner tells its requesters which of the rows they
        asked for are flagged (one exchange of a 0/1 column through the plan, at
        set-up).  Returns the sub-plan (send_idx, per-peer counts, positions of the
        flagged rows among the halo rows) and the flags of the halo rows."""
        flag = local_flag.to(torch.float32).view(-1, 1).contiguous().to(self.dev)
        got = torch.zeros((max(plan.n_halo, 1), 1), dtype=torch.float32, device=self.dev)
        self._exchange_rows(plan, flag, got)

        def per_peer(marks, counts):
            out, lo = [], 0
            m = marks.to("cpu")
            for c in counts:
                out.append(int(m[lo:lo + c].sum().item()))
                lo += c
            return out

        sub = _SubPlan()
        if plan.n_send:
            on_s = flag[plan.send_idx.long(), 0] != 0
            sub.send_idx = plan.send_idx[on_s].contiguous()
            sub.send_counts = per_peer(on_s, plan.send_counts)
        else:
            sub.send_idx = torch.empty(0, dtype=torch.int32, device=self.dev)
            sub.send_counts = [0] * self.world
        on_h = got[:plan.n_halo, 0] != 0
        sub.halo_pos = torch.nonzero(on_h).view(-1).to(torch.int32).contiguous()
        sub.recv_counts = per_peer(on_h, plan.recv_counts)
        sub.n_send, sub.n_halo = int(sub.send_idx.numel()), int(sub.halo_pos.numel())
        sub.send_buf = sub.recv_buf = None
        return sub, on_h

    # ------------------------------------------- loss gradient: labelled rows only
    def _setup_label_exchange(self):
        """dZ = (softmax - y) * mask / total is exactly zero outside the labelled
        rows, so of the halo rows of dZ only the labelled ones travel; the others
        stay zero in a resident halo buffer.  Set-up: every owner tells its
        requesters which of the requested rows are labelled (one exchange of a
        0/1 column through the backward plan); per step the labelled rows are
        exchanged packed and placed with gcnx_scatter_rows_f32.  The product that
        reads them is unchanged, so are its sums."""
        plan = self.plan_bwd
        scatter = getattr(self.k, "scatter_rows", None)
        self.dz_plan = None
        if scatter is None:
            return
        sub, lab_h = self._flagged_subplan(plan, self.ymask != 0)
        self.dz_halo_labelled = lab_h
        sub.send_buf = torch.empty((max(sub.n_send, 1), self.c), dtype=torch.float32, device=self.dev)
        sub.recv_buf = torch.empty((max(sub.n_halo, 1), self.c), dtype=torch.float32, device=self.dev)
        # halo rows of dZ: zero for good, except the labelled ones (rewritten every step)
        self.dz_halo = torch.zeros((max(plan.n_halo, 1), self.c), dtype=torch.float32, device=self.dev)
        self.dz_plan = sub

    def _setup_loss_gradient_matrix(self, enabled):
        """``self.at_dz``: the matrix of the product A^T dZ restricted to the columns
        whose dZ row can be nonzero (labelled local rows and labelled halo rows), or
        None (the full matrix is used)."""
        self.at_dz = self.at_dz_c = self.dz_rows = self.p1_c = self.ax_c = None
        # the sub-matrices below are derived from the label mask as it is now
        self._mask_version = getattr(self.ymask, "_version", None)
        filt = getattr(self.k, "csr_filter_columns", None)
        if not enabled or filt is None:
            return
        lab = self.ymask != 0
        if self.world == 1:
            self.at_dz = filt(self.at, lab)
        elif self.dz_plan is not None:
            plan = self.plan_bwd
            flag = torch.cat([lab, self.dz_halo_labelled[:plan.n_halo]]) if plan.n_halo else lab
            self.at_dz = filt(plan.csr, flag)
        # Few rows of dP2 can be nonzero (the nodes with a labelled neighbour): the
        # product writes them next to each other and they are put in place in the
        # resident dP2, whose other rows stay zero; H'^T dP2 sums over those rows only
        compact = getattr(self.k, "csr_compact_rows", None)
        if self.at_dz is not None and compact is not None and hasattr(self.k, "scatter_rows") and \
                hasattr(self.k, "gather_rows"):
            sub, rows = compact(self.at_dz)
            if 0 < sub.n_rows <= self.n_loc // 2:
                self.at_dz_c, self.dz_rows = sub, rows.to(self.dev)
                self.dp2_c = torch.empty((sub.n_rows, self.c), dtype=torch.float32, device=self.dev)
                self.hd_c = torch.empty((sub.n_rows, self.h), dtype=torch.float32, device=self.dev)
                self.dp2.zero_()
                if self.order == "aggregate_first":
                    # the gated product and the weight gradient (A_hat X')^T dZ1 over those rows only
                    self.p1_c = torch.empty((sub.n_rows, self.h), dtype=torch.float32, device=self.dev)
                    self.ax_c = torch.empty((sub.n_rows, self.f), dtype=torch.float32, device=self.dev)

    def _setup_backward_matrix(self):
        """``self.at_wide``: the matrix of the hidden-width backward product
        A^T d(A P1) restricted to the columns whose row of d(A P1) can be nonzero.
        d(A P1) = gate(dP2 W2^T) row by row, and a row of dP2 = A^T dZ is exactly
        zero unless the node has a labelled neighbour (or is labelled): the rows of
        ``self.at_dz`` that hold a nonzero.  None: the full matrix."""
        self.at_wide = self.at_wide_c = self.hb_rows = self.x_c = self.hb_c = None
        self.hit_plan = None
        filt = getattr(self.k, "csr_filter_columns", None)
        if self.at_dz is None or filt is None or self.order == "aggregate_first":
            return
        rp = torch.as_tensor(self.at_dz.row_ptr)
        hit = (rp[1:] - rp[:-1]) > 0                     # local rows of dP2 that can be nonzero
        if self.world == 1:
            self.at_wide = filt(self.at, hit)
        elif self.recompute:
            plan = self.plan
            # the owners tell which of the requested rows can be nonzero (set-up, once); only
            # those rows of dP2 and of the sign words of H' travel from here on
            sub, hit_h = self._flagged_subplan(plan, hit)
            flag = torch.cat([hit.to(hit_h.device), hit_h]) if plan.n_halo else hit
            self.at_wide = filt(plan.csr, flag)
            if hasattr(self.k, "scatter_rows"):
                words = self.h // 32
                e = lambda *sh, **kw: torch.empty(sh, dtype=kw.get("dtype", torch.float32), device=self.dev)
                sub.send_buf, sub.recv_buf = e(max(sub.n_send, 1), self.c), e(max(sub.n_halo, 1), self.c)
                sub.send_bits = e(max(sub.n_send, 1), words, dtype=torch.int32)
                sub.recv_bits = e(max(sub.n_halo, 1), words, dtype=torch.int32)
                # rows that do not qualify are never read by the filtered product; kept zero all the same
                self.dp2_halo.zero_()
                self.hbits_halo.zero_()
                self.hit_plan = sub
        self._setup_backward_rows()

    def _setup_backward_rows(self):
        """When clearly fewer than all rows of dP1 = A^T d(A P1) can be nonzero (the
        non-empty rows of ``self.at_wide``), the product writes those rows next to each
        other (``self.hb_c``) and the weight gradient X'^T dP1 sums over them only,
        against a copy of those rows of the features gathered once (``self.x_c``; the
        dropout mask of a row is drawn by its id)."""
        self.at_wide_c = self.hb_rows = self.x_c = self.hb_c = None
        compact = getattr(self.k, "csr_compact_rows", None)
        if self.at_wide is None or compact is None or not hasattr(self.k, "gather_rows"):
            return
        if self.keep < 1.0 and not self.fuse_x_dropout:
            return                                  # dropout(X) is stored per step: nothing to gather once
        sub, rows = compact(self.at_wide)
        m = int(sub.n_rows)
        if not (0 < m <= (self.n_loc * 7) // 10):
            return
        if self.keep < 1.0:
            sup = getattr(self.k, "gemm_supports_a_dropout_rows_t", None)
            if sup is None or not sup(self.f, self.h, m):
                return
        self.at_wide_c, self.hb_rows = sub, rows.to(self.dev)
        self.x_c = torch.empty((m, self.f), dtype=torch.float32, device=self.dev)
        self.k.gather_rows(self.x, self.hb_rows, self.x_c)
        self.hb_c = torch.empty((m, self.h), dtype=torch.float32, device=self.dev)
        self.hb = None                              # the full-height buffer is not needed

    # ------------------------------------ the step over the labelled rows' field only
    def _setup_labelled_field(self, pack_records):
        """``self.field``: the row sets, matrices and buffers of the training step
        restricted to what the labelled rows depend on (see ``labelled_field_only``),
        or None when it does not apply or would not pay."""
        k = self.k
        need = ("csr_transpose", "csr_keep_rows", "csr_compact_rows", "csr_filter_columns", "gather_rows")
        agg = self.order == "aggregate_first"
        if self.world != 1 or self.at_dz_c is None or self.keep_masks or any(not hasattr(k, n) for n in need):
            return
        if self.keep <= 0.0 or (agg and self.keep < 1.0 and not hasattr(k, "dropout_rows_philox")):
            return
        at_wide = self.at_wide
        if at_wide is None:
            # (the aggregate-first order has no hidden-width backward product of its own)
            rp = torch.as_tensor(self.at_dz.row_ptr)
            at_wide = k.csr_filter_columns(self.at, (rp[1:] - rp[:-1]) > 0)
        dev, c, h = self.dev, self.c, self.h
        e = lambda *sh, **kw: torch.empty(sh, dtype=kw.get("dtype", torch.float32), device=dev)
        fd = type("LabelledField", (), {})()
        lab = torch.nonzero(self.ymask != 0).view(-1)
        fd.lab_rows = lab.to(torch.int32).contiguous()
        fd.rows1 = self.dz_rows                      # N1: nodes some labelled row reads
        n_l, n1 = int(lab.numel()), int(self.dz_rows.numel())
        if n_l == 0:
            return
        # A_hat[labelled, N1]: at_dz_c is (A_hat^T)[N1, labelled columns], rows next to each other
        t = k.csr_transpose(self.at_dz_c)           # [n, n1]; only labelled rows hold nonzeros
        fd.a_z = k.csr_keep_rows(t, fd.lab_rows)    # [n_l, n1]
        fd.a_zt = k.csr_transpose(fd.a_z)           # [n1, n_l]
        # N2: the nodes some row of N1 reads (the rows of at_wide that hold a nonzero)
        if self.at_wide_c is not None:
            src, fd.rows2, fd.x2 = self.at_wide_c, self.hb_rows, self.x_c
        else:
            sub, rows = k.csr_compact_rows(at_wide)
            if 0 < int(sub.n_rows) <= (self.n_loc * 7) // 10 and (self.keep >= 1.0 or
                                                                  hasattr(k, "dropout_rows_philox")):
                src, fd.rows2 = sub, rows.to(dev)
                fd.x2 = torch.empty((int(sub.n_rows), self.f), dtype=torch.float32, device=dev)
                k.gather_rows(self.x, fd.rows2, fd.x2)
            else:
                src, fd.rows2, fd.x2 = at_wide, None, self.x
        n2 = int(src.n_rows)
        # dropout(X) over N2: regenerated inside the two products that read it where their
        # shapes are served, else stored once per step
        fd.fuse_x = fd.xd = None
        fd.agg = agg
        if agg:
            # (A_hat[N1, N2] X') W1: the sparse product reads dropout(X) over N2, stored once per step
            if self.keep < 1.0:
                fd.xd = e(n2, self.f) if fd.rows2 is not None or self.xd is None else self.xd
        elif self.keep < 1.0:
            sup = getattr(k, "gemm_supports_a_dropout", None)
            sup_t = getattr(k, "gemm_supports_a_dropout_rows_t", None)
            fd.fuse_x = bool(sup is not None and sup(False, n2, h, self.f) and sup(True, self.f, h, n2) and
                             (fd.rows2 is None or (sup_t is not None and sup_t(self.f, h, n2))))
            if not fd.fuse_x:
                fd.xd = e(n2, self.f) if fd.rows2 is not None or self.xd is None else self.xd
        t = k.csr_transpose(src)                     # [n, n2]; only rows of N1 hold nonzeros
        fd.a_h = k.csr_keep_rows(t, fd.rows1)        # A_hat[N1, N2]
        del t
        # (A_hat^T)[N2, N1], columns numbered within N1 (the aggregate-first order never forms dP1)
        fd.a_ht = None if agg else k.csr_transpose(fd.a_h)
        wide = self.f if agg else h                  # width of the operand the hidden-layer SpMM gathers
        hint = getattr(fd.a_h, "set_gather_hint", None)
        if hint is not None and wide > 128 and n2 * wide * 4 > (1 << 30) and fd.a_h.nnz > (1 << 24):
            hint(wide * 4)
        fd.n_l, fd.n1, fd.n2 = n_l, n1, n2
        fd.nnz_h, fd.nnz_z = int(fd.a_h.nnz), int(fd.a_z.nnz)
        if agg:
            fd.ax = e(n1, self.f)                    # A_hat[N1, N2] X'
            fd.p1 = fd.hb = None
        else:
            fd.p1 = e(n2, h)                         # X' W1 over N2
            # dP1 over N2
            fd.hb = self.hb_c if self.at_wide_c is not None else self.hb if fd.rows2 is None else e(n2, h)
        fd.hd = self.hd_c                            # H' over N1
        fd.p2 = e(n1, c)
        fd.z, fd.dz = e(n_l, c), e(n_l, c)
        fd.y = self.y[lab].contiguous()
        fd.ymask = self.ymask[lab].contiguous()
        words = h // 32
        sup_bits = getattr(k, "spmm_supports_sign_bits", None)
        fd.hbits = None
        fd.dp2 = None
        if self.factored_bwd and pack_records and c % 4 == 0 and 4 * (c + words) <= 128:
            # the row of dP2 and the sign words of a node of N1 side by side in one
            # 128-byte record, as in the full-height step
            fd.records = torch.zeros((n1, 32), dtype=torch.float32, device=dev)
            fd.dp2 = fd.records[:, :c]
            fd.hbits = fd.records.view(torch.int32)[:, c:c + words]
        else:
            fd.dp2 = e(n1, c)
            if sup_bits is not None and sup_bits(h):
                fd.hbits = torch.empty((n1, words), dtype=torch.int32, device=dev)
        fd.factored = bool(self.factored_bwd and fd.hbits is not None)
        if agg:
            fd.hbits = None                          # the gate is read from the few rows of H' themselves
        fd.g = None if fd.factored else e(n1, h)     # the gated product over N1 when it has to be stored
        self.field = fd

    def use_labelled_field(self):
        """From now on the training step runs over the labelled rows' field (see
        ``labelled_field_only``); returns whether it applies to this engine."""
        if self.field is None:
            self._setup_labelled_field(self._pack_gate_operands)
        return self.field is not None

    def _train_step_field(self):
        """One optimisation step over the labelled rows' field (``self.field``)."""
        k, fd = self.k, self.field
        keep = self.keep
        off = self.step_no * (1 << 44)
        xkey = (keep, self.seed * 2 + 1, off + self.row_begin * self.f) if keep < 1.0 else None
        xrows = fd.rows2
        xin = fd.x2
        if fd.agg:
            return self._train_step_field_aggregate_first(xkey, off)
        if xkey is not None and not fd.fuse_x:
            if xrows is not None:
                k.dropout_rows_philox(fd.x2, xrows, xkey[0], xkey[1], xkey[2], out=fd.xd)
            else:
                k.dropout_philox(fd.x2.view(-1), xkey[0], xkey[1], xkey[2], out=fd.xd.view(-1))
            xin, xkey = fd.xd, None
        if xkey is not None:
            k.gemm(xin, self.w1, out=fd.p1, a_dropout=xkey, a_dropout_rows=xrows)
        else:
            k.gemm(xin, self.w1, out=fd.p1)
        drop = (keep, self.seed * 2 + 2, off + self.row_begin * self.h, self.h, 0) if keep < 1.0 else None
        kw = {"sign_bits": fd.hbits} if fd.hbits is not None else {}
        k.spmm(fd.a_h, fd.p1, out=fd.hd, relu=True, dropout=drop, dropout_rows=fd.rows1 if drop is not None else None,
               **kw)
        k.gemm(fd.hd, self.w2, out=fd.p2)
        k.spmm(fd.a_z, fd.p2, out=fd.z)
        out, _ = k.masked_softmax_xent(fd.z, fd.y, fd.ymask, want_grad=True, mask_total=self.mask_total,
                                       grad_out=fd.dz)
        k.spmm(fd.a_zt, fd.dz, out=fd.dp2)                                   # A^T dZ over N1
        k.gemm(fd.hd, fd.dp2, trans_a=True, out=self.g2)                      # H'^T dP2
        scale = 1.0 / keep
        if fd.factored:
            k.spmm_factored(fd.a_ht, fd.dp2, self.w2, fd.hbits, gate_scale=scale, trans_w=True, out=fd.hb)
        else:
            if fd.hbits is not None:
                k.gemm(fd.dp2, self.w2, trans_b=True, out=fd.g, gate_bits=fd.hbits, gate_scale=scale)
            else:
                k.gemm(fd.dp2, self.w2, trans_b=True, out=fd.g, gate=fd.hd, gate_scale=scale)
            k.spmm(fd.a_ht, fd.g, out=fd.hb)
        if xkey is not None:
            k.gemm(xin, fd.hb, trans_a=True, out=self.g1, a_dropout=xkey, a_dropout_rows=xrows)
        else:
            k.gemm(xin, fd.hb, trans_a=True, out=self.g1)
        k.adam_step(self.w1, self.g1, self.m1, self.v1, self.lr, self.step_no, weight_decay=self.wd)
        k.adam_step(self.w2, self.g2, self.m2, self.v2, self.lr, self.step_no)
        return out

    def _train_step_field_aggregate_first(self, xkey, off):
        """The step over the labelled rows' field with the first layer as
        (A_hat[N1, N2] X') W1: the weight gradient is (A_hat[N1, N2] X')^T dZ1 with the
        aggregate kept from the forward pass, so no product runs over N2 but the one
        sparse product, and the dense ones run over the rows of N1."""
        k, fd = self.k, self.field
        keep = self.keep
        xin = fd.x2
        if xkey is not None:
            if fd.rows2 is not None:
                k.dropout_rows_philox(fd.x2, fd.rows2, xkey[0], xkey[1], xkey[2], out=fd.xd)
            else:
                k.dropout_philox(fd.x2.view(-1), xkey[0], xkey[1], xkey[2], out=fd.xd.view(-1))
            xin = fd.xd
        k.spmm(fd.a_h, xin, out=fd.ax)
        k.gemm(fd.ax, self.w1, out=fd.hd, relu=True)
        if keep < 1.0:
            # H' = dropout(relu(.)), the mask of a row drawn by its node id
            k.dropout_rows_philox(fd.hd, fd.rows1, keep, self.seed * 2 + 2, off + self.row_begin * self.h, out=fd.hd)
        k.gemm(fd.hd, self.w2, out=fd.p2)
        k.spmm(fd.a_z, fd.p2, out=fd.z)
        out, _ = k.masked_softmax_xent(fd.z, fd.y, fd.ymask, want_grad=True, mask_total=self.mask_total,
                                       grad_out=fd.dz)
        k.spmm(fd.a_zt, fd.dz, out=fd.dp2)                                   # A^T dZ over N1
        k.gemm(fd.hd, fd.dp2, trans_a=True, out=self.g2)                      # H'^T dP2
        k.gemm(fd.dp2, self.w2, trans_b=True, out=fd.g, gate=fd.hd, gate_scale=1.0 / keep)
        k.gemm(fd.ax, fd.g, trans_a=True, out=self.g1)                        # (A_hat X')^T dZ1
        k.adam_step(self.w1, self.g1, self.m1, self.v1, self.lr, self.step_no, weight_decay=self.wd)
        k.adam_step(self.w2, self.g2, self.m2, self.v2, self.lr, self.step_no)
        return out

    def use_dense_gradients(self):
        """From now on every backward product runs over every nonzero and row, as
        with ``sparse_loss_gradient=False`` (for measuring the two forms in one
        process; not reversible)."""
        self.field = None
        self.at_dz = self.at_dz_c = self.dz_rows = self.p1_c = self.ax_c = None
        self.at_wide = self.at_wide_c = self.hb_rows = self.x_c = self.hb_c = None
        if self.hit_plan is not None and getattr(self, "_gate_records_halo", None) is not None:
            # the full exchanges deliver straight into the halo operands: contiguous buffers again
            nh = max(self.plan.n_halo, 1)
            self.dp2_halo = torch.empty((nh, self.c), dtype=torch.float32, device=self.dev)
            self.hbits_halo = torch.empty((nh, self.h // 32), dtype=torch.int32, device=self.dev)
            self._gate_records_halo = None
        self.hit_plan = None                        # every halo row of dP2 and of the sign words travels again
        if self.hb is None and self.order != "aggregate_first":
            self.hb = torch.empty((self.n_loc, self.h), dtype=torch.float32, device=self.dev)

    def backward_nnz(self):
        """Nonzeros of the local part of A^T that the hidden-width backward product multiplies."""
        if self.at_wide is not None:
            return int(self.at_wide.nnz)
        return int((self.plan_bwd.csr if self.world > 1 else self.at).nnz)

    def loss_gradient_nnz(self):
        """Nonzeros of the local part of A^T that the product A^T dZ multiplies."""
        if self.at_dz is not None:
            return int(self.at_dz.nnz)
        return int((self.plan_bwd.csr if self.world > 1 else self.at).nnz)

    def _loss_gradient_product(self, b2=None):
        """dP2 = A^T [dZ; b2] through the sub-matrix of the labelled columns, its
        nonzero rows formed next to each other and put in place when that pays."""
        if self.at_dz_c is not None:
            self.k.spmm(self.at_dz_c, self.dz, out=self.dp2_c, b2=b2)
            self.k.scatter_rows(self.dp2_c, self.dz_rows, self.dp2)
        else:
            self.k.spmm(self.at_dz, self.dz, out=self.dp2, b2=b2)

    def _aggregate_loss_gradient(self):
        """dP2 = A^T [dZ; halo rows of dZ] with only the labelled halo rows fetched."""
        sub, plan = self.dz_plan, self.plan_bwd
        work = self._exchange_start(sub, self.dz, 0, self.c, sub.send_buf[:sub.n_send], sub.recv_buf[:sub.n_halo])
        if work is not None:
            work.wait()
        if sub.n_halo:
            self.k.scatter_rows(sub.recv_buf[:sub.n_halo], sub.halo_pos, self.dz_halo)
        b2 = self.dz_halo[:plan.n_halo] if plan.n_halo else None
        if self.at_dz is not None:
            self._loss_gradient_product(b2)
        else:
            self.k.spmm(plan.csr, self.dz, out=self.dp2, b2=b2)

    def _exchange_rows(self, plan, src, dst):
        """dst[i, :] = row plan.need[i] of the global matrix whose local rows are
        ``src`` (blocking, in column slabs through the slab buffers)."""
        w = src.shape[1]
        slab = min(self.slab, w)
        for c0 in range(0, w, slab):
            nc = min(slab, w - c0)
            sb = self.sendbuf[0][:plan.n_send * nc].view(plan.n_send, nc)
            hb = self.halobuf[0][:plan.n_halo * nc].view(plan.n_halo, nc)
            work = self._exchange_start(plan, src, c0, nc, sb, hb)
            if work is not None:
                work.wait()
            if plan.n_halo:
                dst[:plan.n_halo, c0:c0 + nc].copy_(hb)

    # ------------------------------------------------------------ collectives
    def _aggregate(self, which, src, dst, relu, dropout=None, sign_bits=None):
        """dst = act(A @ [src; halo rows of src]) for A = A_hat (which='fwd') or
        A_hat^T ('bwd'); src, dst are [n_loc, W].  With several ranks the halo
        rows are fetched in column slabs: the exchange of slab s+1 is in flight
        on the communication stream while the SpMM of slab s runs."""
        if self.world == 1:
            kw = {"sign_bits": sign_bits} if sign_bits is not None else {}
            self.k.spmm(self.a if which == "fwd" else self.at, src, out=dst, relu=relu,
                        dropout=None if dropout is None else dropout + (dst.shape[1], 0), **kw)
            return
        plan = self.plan if which == "fwd" else self.plan_bwd
        w = src.shape[1]
        slab = min(self.slab, w)
        starts = list(range(0, w, slab))
        works = [None] * len(starts)

        def launch(i):
            c0 = starts[i]
            nc = min(slab, w - c0)
            sb = self.sendbuf[i % 2][:plan.n_send * nc].view(plan.n_send, nc)
            hb = self.halobuf[i % 2][:plan.n_halo * nc].view(plan.n_halo, nc)
            works[i] = (self._exchange_start(plan, src, c0, nc, sb, hb), hb, c0, nc)

        launch(0)
        for i in range(len(starts)):
            if i + 1 < len(starts):
                launch(i + 1)
            work, hb, c0, nc = works[i]
            if work is not None:
                work.wait()
            self.k.spmm(plan.csr, src[:, c0:c0 + nc], out=dst[:, c0:c0 + nc], relu=relu,
                        b2=hb if plan.n_halo else None,
                        dropout=None if dropout is None else dropout + (dst.shape[1], c0))

    def _exchange_start(self, plan, src, c0, nc, send_buf, halo_buf):
        """Pack the rows the peers asked for (columns [c0, c0 + nc) of ``src``,
        any 32-bit type) into ``send_buf`` and start their exchange into
        ``halo_buf``; returns a handle with ``wait()`` or None."""
        if self.comm is not None:
            return self.comm.exchange_rows(halo_buf, send_buf, plan.recv_counts, plan.send_counts,
                                           src=src, send_idx=plan.send_idx, col0=c0)
        if plan.n_send:
            self.k.gather_rows(src[:, c0:c0 + nc], plan.send_idx, send_buf)
        return self._all_to_all_start(halo_buf, send_buf, plan.recv_counts, plan.send_counts)

    def _all_to_all_start(self, recv, send, recv_counts, send_counts):
        """Start the variable-size all-to-all (rows of dim 0) of already packed
        rows; returns a work handle or None when it has already completed."""
        if self.comm is not None:
            # everything that travels between the ranks once the library's communicator
            # exists goes through it -- the set-up lists of the halo plan (row ranges,
            # counts, requested row ids: 32- and 64-bit integers, moved as 32-bit words)
            # as well as the per-step rows -- so that one communicator on one stream
            # carries all traffic in issue order and torch.distributed's communicator
            # is never in flight next to it
            return self.comm.exchange_rows(self._as_words(recv), self._as_words(send), recv_counts, send_counts)
        import torch.distributed as dist
        return dist.all_to_all_single(recv, send, output_split_sizes=recv_counts,
                                      input_split_sizes=send_counts, group=self.group, async_op=True)

    @staticmethod
    def _as_words(t):
        """``t`` (contiguous, rows along dim 0, 4- or 8-byte elements) viewed as a 2-D
        matrix of 32-bit words with the same rows: the shape the C-ABI exchange moves."""
        if not t.is_contiguous():
            raise ValueError("exchange buffers must be contiguous")
        rows = int(t.shape[0]) if t.dim() else 1
        per_row = 1
        for d in t.shape[1:]:
            per_row *= int(d)
        if t.element_size() == 4:
            return t.view(rows, per_row)
        if t.element_size() == 8:
            return t.view(torch.int32).view(rows, 2 * per_row)
        raise ValueError("exchange buffers hold 4- or 8-byte elements")

    def _all_to_all(self, recv, send, recv_counts, send_counts):
        work = self._all_to_all_start(recv, send, recv_counts, send_counts)
        if work is not None:
            work.wait()

    def _allreduce(self, t):
        if self.world > 1:
            if self.comm is not None:
                return self.comm.allreduce_sum_(t)
            import torch.distributed as dist
            dist.all_reduce(t, group=self.group)
        return t

    # ------------------------------------------------------------------- step
    def _forward_aggregate_first(self, train):
        """H' = dropout(relu((A_hat X') W1)); layer 2 as in the reference order."""
        k = self.k
        keep = self.keep if train else 1.0
        off = self.step_no * (1 << 44)
        nh = self.plan.n_halo if self.world > 1 else 0
        xin, xh = self.x, (self.x_halo[:nh] if nh else None)
        if keep < 1.0:
            k.dropout_philox(self.x.view(-1), keep, self.seed * 2 + 1, off + self.row_begin * self.f,
                             out=self.xd.view(-1), mask_out=self.mask1)
            xin = self.xd
            if nh:
                k.dropout_rows_philox(self.x_halo[:nh], self.plan.need, keep, self.seed * 2 + 1, off,
                                      out=self.xd_halo[:nh])
                xh = self.xd_halo[:nh]
        if self.world == 1:
            k.spmm(self.a, xin, out=self.ax)
        else:
            k.spmm(self.plan.csr, xin, out=self.ax, b2=xh)
        drop = (keep, self.seed * 2 + 2, off + self.row_begin * self.h, self.h, 0) if keep < 1.0 else None
        if train and drop is not None and self.hbits is not None:
            k.gemm(self.ax, self.w1, out=self.hd, relu=True, dropout=drop, sign_bits=self.hbits)
        else:
            k.gemm(self.ax, self.w1, out=self.hd, relu=True, dropout=drop)
        k.gemm(self.hd, self.w2, out=self.p2)
        self._aggregate_logits()
        return xin, self.hd, None

    def forward(self, train):
        if self.order == "aggregate_first":
            return self._forward_aggregate_first(train)
        k = self.k
        keep = self.keep if train else 1.0
        xdrop = None
        off = self.step_no * (1 << 44)
        if keep < 1.0:
            xkey = (keep, self.seed * 2 + 1, off + self.row_begin * self.f)
            if self.fuse_x_dropout:
                xin, xdrop = self.x, xkey
            else:
                k.dropout_philox(self.x.view(-1), xkey[0], xkey[1], xkey[2], out=self.xd.view(-1),
                                 mask_out=self.mask1)
                xin = self.xd
        else:
            xin = self.x
        if xdrop is not None:
            k.gemm(xin, self.w1, out=self.p1, a_dropout=xdrop)
        else:
            k.gemm(xin, self.w1, out=self.p1)
        # H' = dropout(relu(A P1)) straight out of the SpMM epilogue; H itself is
        # never stored (the backward pass only needs the sign pattern of H')
        drop = (keep, self.seed * 2 + 2, off + self.row_begin * self.h) if keep < 1.0 else None
        if self.recompute:
            plan = self.plan
            nh = plan.n_halo
            if nh:
                # P1 of the halo rows from the resident remote feature rows; their
                # dropout mask is the one their owner draws (global element index)
                if keep < 1.0:
                    k.gemm(self.x_halo[:nh], self.w1, out=self.wide_halo[:nh],
                           a_dropout=(keep, self.seed * 2 + 1, off), a_dropout_rows=plan.need)
                else:
                    k.gemm(self.x_halo[:nh], self.w1, out=self.wide_halo[:nh])
            kw = {"sign_bits": self.hbits} if (train and self.hbits is not None) else {}
            k.spmm(plan.csr, self.p1, out=self.hd, relu=True, b2=self.wide_halo[:nh] if nh else None,
                   dropout=None if drop is None else drop + (self.h, 0), **kw)
            if train:
                # the backward pass gates the halo rows of d(A P1) with the sign of
                # the owners' H': 1 bit per element, sent while layer 2 runs
                if self.hbits is not None:
                    # the SpMM already wrote the sign words of every local row: the
                    # requested rows are picked (32 bytes each, a bit-preserving copy) and sent
                    sub = self.hit_plan
                    if sub is not None:
                        # only the rows whose gradient can be nonzero (placed when waited for)
                        self._bits_work = self._exchange_start(
                            sub, self.hbits.view(torch.float32), 0, self.hbits.shape[1],
                            sub.send_bits[:sub.n_send].view(torch.float32),
                            sub.recv_bits[:sub.n_halo].view(torch.float32))
                    else:
                        self._bits_work = self._exchange_start(
                            plan, self.hbits.view(torch.float32), 0, self.hbits.shape[1],
                            self.hbits_send[:plan.n_send].view(torch.float32),
                            self.hbits_halo[:nh].view(torch.float32))
                else:
                    if plan.n_send:
                        k.pack_sign_bits(self.hd, rows=plan.send_idx, out=self.hbits_send[:plan.n_send])
                    self._bits_work = self._all_to_all_start(self.hbits_halo[:nh], self.hbits_send[:plan.n_send],
                                                             plan.recv_counts, plan.send_counts)
        else:
            self._aggregate("fwd", self.p1, self.hd, relu=True, dropout=drop,
                            sign_bits=self.hbits if (train and self.world == 1) else None)
        hin = self.hd
        k.gemm(hin, self.w2, out=self.p2)
        self._aggregate_logits()
        return xin, hin, xdrop

    def _aggregate_logits(self):
        """Z = A_hat P2 (class width)."""
        if self.fwd_local is not None:
            self._aggregate_class_overlapped(self.p2, self.z)
        else:
            self._aggregate("fwd", self.p2, self.z, relu=False)

    def loss(self, want_grad):
        out, _ = self.k.masked_softmax_xent(self.z, self.y, self.ymask, want_grad=want_grad,
                                            mask_total=self.mask_total, grad_out=self.dz)
        return self._allreduce(out)

    def train_step(self):
        """One optimisation step.  Returns a 2-element device tensor
        [cross-entropy part of the loss, accuracy] (before the update)."""
        k = self.k
        if (self.at_dz is not None or self.dz_plan is not None) and \
                getattr(self.ymask, "_version", None) != self._mask_version:
            raise RuntimeError("the label mask was modified after the engine was set up; its backward "
                               "sub-matrices and exchanges are derived from it: build a new engine")
        self.step_no += 1
        if self.field is not None:
            return self._train_step_field()
        xin, hin, xdrop = self.forward(train=True)
        out = self.loss(want_grad=True)
        keep = self.keep
        # layer 2
        if self.dz_plan is not None:
            self._aggregate_loss_gradient()                                  # A^T dZ, labelled halo rows only
        elif self.at_dz is not None:
            self._loss_gradient_product()                                    # A^T dZ, labelled columns only
        else:
            self._aggregate("bwd", self.dz, self.dp2, relu=False)            # A^T dZ
        dp2_work = None
        if self.recompute:
            plan = self.plan
            sub = self.hit_plan
            if sub is not None:
                dp2_work = self._exchange_start(sub, self.dp2, 0, self.c, sub.send_buf[:sub.n_send],
                                                sub.recv_buf[:sub.n_halo])
            else:
                dp2_work = self._exchange_start(plan, self.dp2, 0, self.c, self.dp2_send[:plan.n_send],
                                                self.dp2_halo[:plan.n_halo])
        if self.at_dz_c is not None:
            # the rows of dP2 that can be nonzero, and the same rows of H'
            k.gather_rows(hin, self.dz_rows, self.hd_c)
            k.gemm(self.hd_c, self.dp2_c, trans_a=True, out=self.g2)       # H'^T dP2
        else:
            k.gemm(hin, self.dp2, trans_a=True, out=self.g2)                # H'^T dP2
        # d(A P1) = (H' > 0) ? (dP2 W2^T) / keep : 0, gate applied in the epilogue
        agg_rows = self.order == "aggregate_first" and self.p1_c is not None
        if self.factored_bwd:
            pass                            # formed per nonzero inside the SpMM below
        elif agg_rows:
            # a row of the gated product is zero where the row of dP2 is: the others only
            k.gemm(self.dp2_c, self.w2, trans_b=True, out=self.p1_c, gate=self.hd_c, gate_scale=1.0 / keep)
        elif self.hbits is not None:
            k.gemm(self.dp2, self.w2, trans_b=True, out=self.p1, gate_bits=self.hbits, gate_scale=1.0 / keep)
        else:
            k.gemm(self.dp2, self.w2, trans_b=True, out=self.p1, gate=hin, gate_scale=1.0 / keep)
        # layer 1 (rows of dP1 that can be nonzero next to each other when that is set up)
        bwd_a = self.at_wide_c if self.at_wide_c is not None else self.at_wide
        bwd_out = self.hb_c if self.at_wide_c is not None else self.hb
        if self.order == "aggregate_first":
            # the gated product is the gradient wrt (A_hat X') W1; X needs none
            if agg_rows:
                k.gather_rows(self.ax, self.dz_rows, self.ax_c)
                k.gemm(self.ax_c, self.p1_c, trans_a=True, out=self.g1)    # (A_hat X')^T dZ1 over its nonzero rows
            else:
                k.gemm(self.ax, self.p1, trans_a=True, out=self.g1)        # (A_hat X')^T dZ1
        elif self.recompute:
            plan = self.plan
            nh = plan.n_halo
            for work in (dp2_work, self._bits_work):
                if work is not None:
                    work.wait()
            self._bits_work = None
            sub = self.hit_plan
            if sub is not None and sub.n_halo:
                # the received rows go to their places among the halo rows
                k.scatter_rows(sub.recv_buf[:sub.n_halo], sub.halo_pos, self.dp2_halo)
                if self.hbits is not None:
                    k.scatter_rows(sub.recv_bits[:sub.n_halo].view(torch.float32), sub.halo_pos,
                                   self.hbits_halo.view(torch.float32))
            if self.factored_bwd:
                # local and halo rows of d(A P1) alike from dP2 rows and sign words
                k.spmm_factored(bwd_a if bwd_a is not None else plan.csr, self.dp2, self.w2,
                                self.hbits, gate_scale=1.0 / keep, trans_w=True,
                                out=bwd_out, d2=self.dp2_halo[:nh] if nh else None,
                                gate_bits2=self.hbits_halo[:nh] if nh else None)
            else:
                if nh:
                    # the same gated product for the halo rows, from the received rows
                    # of dP2 and the received sign bits of the owners' H'
                    k.gemm(self.dp2_halo[:nh], self.w2, trans_b=True, out=self.wide_halo[:nh],
                           gate_bits=self.hbits_halo[:nh], gate_scale=1.0 / keep)
                k.spmm(bwd_a if bwd_a is not None else plan.csr, self.p1, out=bwd_out,
                       b2=self.wide_halo[:nh] if nh else None)
        elif self.factored_bwd:
            k.spmm_factored(bwd_a if bwd_a is not None else self.at, self.dp2, self.w2, self.hbits,
                            gate_scale=1.0 / keep, trans_w=True, out=bwd_out)   # A^T dH -> dP1, dH never stored
        elif self.world == 1 and bwd_a is not None:
            k.spmm(bwd_a, self.p1, out=bwd_out)                          # A^T dH -> dP1
        else:
            self._aggregate("bwd", self.p1, self.hb, relu=False)         # A^T dH -> dP1
        if self.order != "aggregate_first":
            if self.at_wide_c is not None:
                # over the rows of dP1 that can be nonzero; row i of x_c is row hb_rows[i] of X
                if xdrop is not None:
                    k.gemm(self.x_c, self.hb_c, trans_a=True, out=self.g1, a_dropout=xdrop,
                           a_dropout_rows=self.hb_rows)
                else:
                    k.gemm(self.x_c, self.hb_c, trans_a=True, out=self.g1)
            elif xdrop is not None:
                k.gemm(xin, self.hb, trans_a=True, out=self.g1, a_dropout=xdrop)  # X'^T dP1, X' regenerated
            else:
                k.gemm(xin, self.hb, trans_a=True, out=self.g1)                 # X'^T dP1
        self._allreduce(self.g1)
        self._allreduce(self.g2)
        k.adam_step(self.w1, self.g1, self.m1, self.v1, self.lr, self.step_no, weight_decay=self.wd)
        k.adam_step(self.w2, self.g2, self.m2, self.v2, self.lr, self.step_no)
        return out

    def evaluate(self):
        self.forward(train=False)
        return self.loss(want_grad=False)

    def weight_decay_loss(self):
        return 0.5 * self.wd * float((self.w1.double() ** 2).sum().item())

    # work accounting for the benchmark -------------------------------------
    def spmm_units_per_step(self, nnz_global, nnz_loss_gradient_global=None, nnz_backward_global=None):
        """nnz x width summed over the SpMM launches of one training step (four
        in the reference order; the aggregate-first order has no hidden-width
        backward SpMM).  ``nnz_loss_gradient_global`` / ``nnz_backward_global``:
        the nonzeros the products A^T dZ and A^T d(A P1) actually multiply, summed
        over the ranks (loss_gradient_nnz(), backward_nnz(); default: all of them).
        A step over the labelled rows' field multiplies the nonzeros of A_hat[N1, N2]
        twice at hidden width and those of A_hat[labelled, N1] twice at class width."""
        if self.field is not None:
            if self.field.agg:
                return self.field.nnz_h * self.f + 2 * self.field.nnz_z * self.c
            return 2 * (self.field.nnz_h * self.h + self.field.nnz_z * self.c)
        nz = nnz_global if nnz_loss_gradient_global is None else nnz_loss_gradient_global
        nb = nnz_global if nnz_backward_global is None else nnz_backward_global
        if self.order == "aggregate_first":
            return nnz_global * (self.f + self.c) + nz * self.c
        return nnz_global * (self.h + self.c) + nb * self.h + nz * self.c
