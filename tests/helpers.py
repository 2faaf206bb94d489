"""Shared helpers of the parity tests."""
import numpy as np

# fp32 parity bar of the propagation path: 1e-5 relative.  "Relative" is taken
# against the size of the sum's terms (sum_k |a_ik * b_kj|, computed in fp64 by
# the oracle), which is the quantity fp32 rounding error actually scales with;
# a plain per-element ratio is meaningless where a sum cancels to ~0.
RTOL = 1e-5


def assert_close_to_f64(got, ref64, mag64, rtol=RTOL, what=""):
    got = np.asarray(got, dtype=np.float64)
    err = np.abs(got - ref64)
    bound = rtol * mag64 + 1e-30
    bad = err > bound
    if bad.any():
        i = np.unravel_index(np.argmax(err / bound), err.shape)
        raise AssertionError("%s: %d elements off; worst at %s: got %r ref %r |terms| %r" %
                             (what, int(bad.sum()), i, got[i], ref64[i], mag64[i]))


def rel_maxnorm(got, ref):
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    den = np.abs(ref).max()
    return float(np.abs(got - ref).max() / den) if den > 0 else float(np.abs(got).max())


def random_coo(rng, n_rows, n_cols, nnz, hub_rows=(), empty_rows=(), dup_frac=0.05):
    """Unsorted COO with duplicates, optional hub rows and forced-empty rows."""
    rows = rng.randint(0, n_rows, size=nnz).astype(np.int32)
    cols = rng.randint(0, n_cols, size=nnz).astype(np.int32)
    for h, cnt in hub_rows:
        idx = rng.choice(nnz, size=min(cnt, nnz), replace=False)
        rows[idx] = h
    ndup = int(nnz * dup_frac)
    if ndup and nnz > 1:
        src = rng.randint(0, nnz, size=ndup)
        dst = rng.randint(0, nnz, size=ndup)
        rows[dst] = rows[src]
        cols[dst] = cols[src]
    if len(empty_rows):
        keep = ~np.isin(rows, np.asarray(empty_rows))
        rows, cols = rows[keep], cols[keep]
    vals = rng.standard_normal(len(rows))
    return rows, cols, vals


class CpuCsr(object):
    """Host stand-in for gcn_amd.sparse.Csr used by the CPU distributed tests."""

    def __init__(self, m):
        m = m.tocsr()
        m.sort_indices()
        self.n_rows, self.n_cols = m.shape
        self.nnz = m.nnz
        self.row_ptr = m.indptr.astype(np.int64)
        self.col_idx = m.indices.astype(np.int32)
        self.val = m.data.astype(np.float32)


class OracleKernels(object):
    """The engine's kernel interface answered by the CPU oracle, so that the
    partition / all-gather / all-reduce logic of gcn_amd.engine can be exercised
    with the gloo backend on a machine without GPUs.  Test-only: the product
    default is gcn_amd.ops (HIP)."""

    def __init__(self):
        import torch
        import oracle
        self.torch, self.o = torch, oracle

    def halo_plan(self, a, row_begin, n_loc):
        """numpy counterpart of gcn_amd.halo.device_halo_plan."""
        import copy
        c = a.col_idx.astype(np.int64)
        local = (c >= row_begin) & (c < row_begin + n_loc)
        need = np.unique(c[~local])
        new = np.where(local, c - row_begin, n_loc + np.searchsorted(need, c))
        out = copy.copy(a)
        out.col_idx = new.astype(np.int32)
        out.n_cols = int(n_loc + len(need))
        return out, self.torch.from_numpy(need.astype(np.int32))

    def csr_filter_columns(self, a, col_keep):
        """numpy counterpart of gcn_amd.sparse.Csr.filter_columns."""
        import copy
        flag = np.asarray(col_keep.numpy() if hasattr(col_keep, "numpy") else col_keep) != 0
        keep = flag[a.col_idx.astype(np.int64)]
        before = np.concatenate([[0], np.cumsum(keep, dtype=np.int64)])
        out = copy.copy(a)
        out.row_ptr = before[a.row_ptr].astype(np.int64)
        out.col_idx = a.col_idx[keep]
        out.val = a.val[keep]
        out.nnz = int(keep.sum())
        return out

    def csr_compact_rows(self, a):
        """numpy counterpart of gcn_amd.sparse.Csr.compact_rows."""
        import copy
        rows = np.nonzero(np.diff(a.row_ptr) > 0)[0]
        out = copy.copy(a)
        out.row_ptr = np.concatenate([a.row_ptr[rows], a.row_ptr[-1:]]).astype(np.int64)
        out.n_rows = int(len(rows))
        return out, self.torch.from_numpy(rows.astype(np.int32))

    def csr_keep_rows(self, a, rows):
        """numpy counterpart of gcn_amd.sparse.Csr.keep_rows."""
        import copy
        r = np.asarray(rows.numpy() if hasattr(rows, "numpy") else rows).astype(np.int64)
        assert int((a.row_ptr[r + 1] - a.row_ptr[r]).sum()) == a.nnz
        out = copy.copy(a)
        out.row_ptr = np.concatenate([a.row_ptr[r], a.row_ptr[-1:]]).astype(np.int64)
        out.n_rows = int(len(r))
        return out

    def csr_transpose(self, a):
        import scipy.sparse as sp
        m = sp.csr_matrix((a.val, a.col_idx, a.row_ptr), shape=(a.n_rows, a.n_cols))
        return CpuCsr(m.T.tocsr())

    def gather_rows(self, src, idx, out):
        out.copy_(src[idx.long()])
        return out

    def scatter_rows(self, src, idx, out):
        out[idx.long()] = src
        return out

    def spmm_supports_sign_bits(self, width):
        return width % 32 == 0

    def spmm(self, a, b, out=None, relu=False, b2=None, dropout=None, sign_bits=None, alpha=1.0, beta=0.0,
             dropout_rows=None, **kw):
        if b2 is not None:
            b = self.torch.cat([b, b2], 0)
        r = self.o.spmm_csr(a.row_ptr, a.col_idx, a.val, b.contiguous().numpy())
        if alpha != 1.0:
            r = r * np.float32(alpha)
        if beta != 0.0:
            r = (r + np.float32(beta) * out.numpy()).astype(np.float32)
        if relu:
            r = np.maximum(r, 0)
        if dropout is not None and dropout[0] < 1.0:
            keep, seed, offset, ld, col0 = dropout
            rows = np.arange(r.shape[0], dtype=np.uint64)[:, None] if dropout_rows is None else \
                dropout_rows.numpy().astype(np.uint64)[:, None]
            cols = np.arange(r.shape[1], dtype=np.uint64)[None, :]
            idx = np.uint64(offset) + rows * np.uint64(ld) + np.uint64(col0) + cols
            m = self.o.bernoulli_mask_at(idx, keep, seed)
            r = np.where(m != 0, r * np.float32(1.0 / keep), np.float32(0)).astype(np.float32)
        t = self.torch.from_numpy(r)
        if sign_bits is not None:
            self.pack_sign_bits(t, out=sign_bits)
        if out is None:
            return t
        out.copy_(t)
        return out

    def spmm_supports_factored(self, width, inner):
        return inner == 16 and width >= 64 and width % 32 == 0

    def spmm_factored(self, a, d, w, gate_bits, gate_scale=1.0, trans_w=False, out=None, d2=None, gate_bits2=None):
        if d2 is not None and d2.shape[0]:
            d = self.torch.cat([d, d2], 0)
            gate_bits = self.torch.cat([gate_bits, gate_bits2], 0)
        g = self.gemm(d, w, trans_b=trans_w, gate_bits=gate_bits, gate_scale=gate_scale)
        return self.spmm(a, g, out=out)

    def gemm_supports_output_dropout(self, trans_a, M, N, K):
        return not trans_a and N % 4 == 0

    def gemm_supports_a_dropout_rows_t(self, M, N, K):
        return True

    def gemm_supports_a_dropout(self, trans_a, M, N, K):
        return True

    def pack_sign_bits(self, x, rows=None, out=None):
        v = x.numpy() if rows is None else x.numpy()[rows.numpy().astype(np.int64)]
        n, w = v.shape
        words = (w + 31) // 32
        on = np.zeros((n, words * 32), dtype=np.uint64)
        on[:, :w] = v > 0
        packed = (on.reshape(n, words, 32) << np.arange(32, dtype=np.uint64)).sum(axis=2)
        t = self.torch.from_numpy(packed.astype(np.uint32).view(np.int32))
        if out is None:
            return t
        out.copy_(t)
        return out

    def dropout_rows_philox(self, x, row_ids, keep, seed, offset, out=None):
        w = x.shape[1]
        idx = np.uint64(offset) + row_ids.numpy().astype(np.uint64)[:, None] * np.uint64(w) + \
            np.arange(w, dtype=np.uint64)[None, :]
        m = self.torch.from_numpy(self.o.bernoulli_mask_at(idx, keep, seed))
        return self.dropout_dense(x, m, 1.0 / keep, out=out)

    def gemm(self, a, b, trans_a=False, trans_b=False, out=None, gate=None, gate_scale=1.0, a_dropout=None,
             a_dropout_rows=None, gate_bits=None, relu=False, dropout=None, sign_bits=None, **kw):
        an = a.contiguous().numpy()
        if a_dropout is not None and a_dropout[0] < 1.0:
            keep, seed, offset = a_dropout
            rows = np.arange(an.shape[0], dtype=np.uint64) if a_dropout_rows is None else \
                a_dropout_rows.numpy().astype(np.uint64)
            idx = np.uint64(offset) + rows[:, None] * np.uint64(an.shape[1]) + \
                np.arange(an.shape[1], dtype=np.uint64)[None, :]
            m = self.o.bernoulli_mask_at(idx, keep, seed)
            an = np.where(m != 0, an * np.float32(1.0 / keep), np.float32(0)).astype(np.float32)
        r = self.torch.from_numpy(self.o.gemm(an, b.contiguous().numpy(), trans_a, trans_b))
        if gate_bits is not None:
            words = gate_bits.numpy().view(np.uint32)
            cols = np.arange(r.shape[1])
            on = (words[:, cols // 32] >> (cols % 32).astype(np.uint32)) & 1
            r = self.torch.where(self.torch.from_numpy(on.astype(np.bool_)), r * gate_scale,
                                 self.torch.zeros_like(r))
        if gate is not None:
            r = self.torch.where(gate > 0, r * gate_scale, self.torch.zeros_like(r))
        if relu:
            r = self.torch.clamp_min(r, 0)
        if dropout is not None and dropout[0] < 1.0:
            keep, seed, offset, ld, col0 = dropout
            rows = np.arange(r.shape[0], dtype=np.uint64)[:, None]
            cols = np.arange(r.shape[1], dtype=np.uint64)[None, :]
            m = self.o.bernoulli_mask_at(np.uint64(offset) + rows * np.uint64(ld) + np.uint64(col0) + cols, keep, seed)
            r = self.torch.where(self.torch.from_numpy(m != 0), r * np.float32(1.0 / keep), self.torch.zeros_like(r))
        if sign_bits is not None:
            self.pack_sign_bits(r, out=sign_bits)
        if out is None:
            return r
        out.copy_(r)
        return out

    def bernoulli_mask(self, n, keep, seed, offset, device, out=None):
        m = self.torch.from_numpy(self.o.bernoulli_mask(n, keep, seed, offset))
        if out is None:
            return m
        out.copy_(m)
        return out

    def dropout_dense(self, x, mask, scale, out=None):
        r = self.torch.where(mask.view(x.shape) != 0, x * scale, self.torch.zeros_like(x))
        if out is None:
            return r
        out.copy_(r)
        return out

    def dropout_philox(self, x, keep, seed, offset, out=None, mask_out=None):
        m = self.torch.from_numpy(self.o.bernoulli_mask(x.numel(), keep, seed, offset))
        if mask_out is not None:
            mask_out.copy_(m)
        return self.dropout_dense(x, m, 1.0 / keep, out=out)

    def relu_dropout_bwd_philox(self, y, dy, keep, seed, offset, out=None):
        m = self.torch.from_numpy(self.o.bernoulli_mask(y.numel(), keep, seed, offset))
        return self.relu_dropout_bwd(y, dy, m, 1.0 / keep, out=out)

    def relu_dropout_bwd(self, y, dy, mask=None, scale=1.0, out=None):
        g = dy if mask is None else self.torch.where(mask.view(dy.shape) != 0, dy * scale, self.torch.zeros_like(dy))
        r = self.torch.where(y > 0, g, self.torch.zeros_like(g))
        if out is None:
            return r
        out.copy_(r)
        return out

    def masked_softmax_xent(self, logits, labels, mask, want_grad=True, mask_total=0.0, grad_out=None):
        z, y, m = logits.numpy(), labels.numpy(), mask.numpy()
        local = float(m.sum())
        total = mask_total if mask_total > 0 else local
        loss, grad = self.o.masked_softmax_xent(z, y, m, want_grad)
        acc = self.o.masked_accuracy(z, y, m)
        share = local / total if total > 0 else 0.0
        out = self.torch.tensor([loss * share, acc * share], dtype=self.torch.float32)
        if want_grad:
            g = self.torch.from_numpy(grad * np.float32(share))
            if grad_out is not None:
                grad_out.copy_(g)
                g = grad_out
            return out, g
        return out, None

    def adam_step(self, w, grad, m, v, lr, t, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0):
        g = (grad + weight_decay * w).contiguous().numpy().astype(np.float32)
        self.o.adam_step(w.numpy(), g, m.numpy(), v.numpy(), lr, t, beta1, beta2, eps)
        return w
