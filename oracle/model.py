"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's two-layer
models and training loop, composed from the C primitives in gcn_oracle.c.

Follows, step by step:
  GraphConvolution._call   gcn/layers.py:203-226  (dropout -> per-support
                           x.W_i -> A_i.(x.W_i) -> add_n -> act)
  Dense._call              gcn/layers.py:113-129
  GCN / MLP                gcn/models.py:85-177   (relu hidden layer, linear
                           output; weight decay on layer-0 variables only)
  masked loss / accuracy   gcn/metrics.py:4-20
  training loop            gcn/train_normal.py:79-107 (eval with dropout 0,
                           early stopping on the validation loss)
Sparse products walk the nonzeros in tuple order, as TensorFlow's CPU kernel
does; sparse values are fp32 (a tf.float32 placeholder).  Initial weights and
dropout masks are inputs, because TensorFlow's random streams cannot be
reproduced.  PARITY UNPINNED against TensorFlow itself (not installable); the
backward pass is cross-checked against torch CPU autograd in tests/.
"""
import numpy as np

from . import cpu


def _tuple32(t):
    coords, vals, shape = t
    coords = np.asarray(coords).reshape(-1, 2)
    return (np.ascontiguousarray(coords[:, 0], dtype=np.int32),
            np.ascontiguousarray(coords[:, 1], dtype=np.int32),
            np.ascontiguousarray(vals, dtype=np.float32), tuple(int(s) for s in shape))


class TwoLayerOracle(object):
    """model in {'gcn', 'gcn_cheby', 'dense'}; ``weights`` = [[W1_0..], [W2_0..]]
    (one matrix per support for the graph models, one per layer for 'dense')."""

    def __init__(self, model, features, supports, weights, lr=0.01, weight_decay=5e-4):
        self.model = model
        self.x = _tuple32(features)
        self.sup = [_tuple32(s) for s in supports]
        self.w = [[np.ascontiguousarray(w, dtype=np.float32).copy() for w in layer] for layer in weights]
        self.m = [[np.zeros_like(w) for w in layer] for layer in self.w]
        self.v = [[np.zeros_like(w) for w in layer] for layer in self.w]
        self.lr = lr
        self.wd = weight_decay
        self.t = 0
        self.n = self.x[3][0]

    # ------------------------------------------------------------- forward
    def forward(self, mask1=None, mask2=None, keep=1.0):
        """Returns logits and the cache for backward.  mask1: uint8 over the
        feature nonzeros (tuple order); mask2: uint8 [N, H]."""
        xr, xc, xv, xs = self.x
        if keep < 1.0:
            xv = cpu.sparse_dropout(xv, mask1, keep)
        graph = self.model != 'dense'
        pre1 = [cpu.spmm_coo(xr, xc, xv, w, self.n) for w in self.w[0]]
        if graph:
            acc = None
            for (r, c, v, _), p in zip(self.sup, pre1):
                s = cpu.spmm_coo(r, c, v, p, self.n)
                acc = s if acc is None else acc + s
        else:
            acc = pre1[0]
        h = np.maximum(acc, np.float32(0))
        if keep < 1.0:
            scale = np.float32(1.0 / keep)
            hd = np.where(mask2.reshape(h.shape) != 0, h * scale, np.float32(0)).astype(np.float32)
        else:
            hd = h
        pre2 = [cpu.gemm(hd, w) for w in self.w[1]]
        if graph:
            out = None
            for (r, c, v, _), p in zip(self.sup, pre2):
                s = cpu.spmm_coo(r, c, v, p, self.n)
                out = s if out is None else out + s
        else:
            out = pre2[0]
        return out, dict(xv=xv, h=h, hd=hd, mask2=mask2, keep=keep)

    def loss_acc(self, logits, labels, mask, want_grad=False):
        xent, grad = cpu.masked_softmax_xent(logits, labels, mask, want_grad)
        reg = np.float32(0)
        for w in self.w[0]:
            reg = reg + np.float32(self.wd) * np.float32((w.astype(np.float32) ** 2).sum(dtype=np.float32) / 2)
        return float(np.float32(xent) + reg), cpu.masked_accuracy(logits, labels, mask), grad

    # ------------------------------------------------------------ backward
    def backward(self, cache, dout):
        xr, xc, _, xs = self.x
        xv = cache["xv"]
        graph = self.model != 'dense'
        n_sup = len(self.w[1])
        g2, dhd = [], None
        for i in range(n_sup):
            if graph:
                r, c, v, _ = self.sup[i]
                dpre = cpu.spmm_coo_adjoint(r, c, v, dout, self.n)
            else:
                dpre = dout
            g2.append(cpu.gemm(cache["hd"], dpre, trans_a=True))
            t = cpu.gemm(dpre, self.w[1][i], trans_b=True)
            dhd = t if dhd is None else dhd + t
        if cache["keep"] < 1.0:
            scale = np.float32(1.0 / cache["keep"])
            dh = np.where(cache["mask2"].reshape(dhd.shape) != 0, dhd * scale, np.float32(0)).astype(np.float32)
        else:
            dh = dhd
        dacc = np.where(cache["h"] > 0, dh, np.float32(0)).astype(np.float32)
        g1 = []
        for i in range(len(self.w[0])):
            if graph:
                r, c, v, _ = self.sup[i]
                dpre = cpu.spmm_coo_adjoint(r, c, v, dacc, self.n)
            else:
                dpre = dacc
            g = cpu.spmm_coo_adjoint(xr, xc, xv, dpre, xs[1])
            g1.append((g + np.float32(self.wd) * self.w[0][i]).astype(np.float32))
        return [g1, g2]

    def adam(self, grads):
        self.t += 1
        for layer in range(2):
            for w, g, m, v in zip(self.w[layer], grads[layer], self.m[layer], self.v[layer]):
                cpu.adam_step(w, np.ascontiguousarray(g, dtype=np.float32), m, v, self.lr, self.t)

    # ---------------------------------------------------------- train loop
    def train_step(self, labels, mask, mask1, mask2, keep):
        logits, cache = self.forward(mask1, mask2, keep)
        loss, acc, grad = self.loss_acc(logits, labels, mask, want_grad=True)
        self.adam(self.backward(cache, grad))
        return loss, acc

    def evaluate(self, labels, mask):
        logits, _ = self.forward()
        loss, acc, _ = self.loss_acc(logits, labels, mask)
        return loss, acc


def train(oracle, y_train, train_mask, y_val, val_mask, y_test, test_mask, mask_fn, epochs=200,
          dropout=0.5, early_stopping=10):
    """The loop of gcn/train_normal.py:79-107.  ``mask_fn(epoch, layer, n)``
    returns the uint8 dropout mask of that layer for that epoch."""
    keep = 1.0 - dropout
    hidden = oracle.w[0][0].shape[1]
    nnz_x = len(oracle.x[2])
    cost_val, trace = [], []
    for epoch in range(epochs):
        m1 = mask_fn(epoch, 0, nnz_x) if keep < 1.0 else None
        m2 = mask_fn(epoch, 1, oracle.n * hidden) if keep < 1.0 else None
        loss, acc = oracle.train_step(y_train, train_mask, m1, m2, keep)
        vloss, vacc = oracle.evaluate(y_val, val_mask)
        cost_val.append(vloss)
        trace.append((loss, acc, vloss, vacc))
        if epoch > early_stopping and cost_val[-1] > np.mean(cost_val[-(early_stopping + 1):-1]):
            break
    tloss, tacc = oracle.evaluate(y_test, test_mask)
    return dict(trace=trace, test_cost=tloss, test_acc=tacc, epochs_run=len(trace))


def dense_gcn_step_f64(a_hat, x, w1, w2, labels, mask, m1, m2, keep):
    """One forward + backward pass of the two-layer GCN with DENSE features
    (gcn/layers.py:203-226 with sparse_inputs=False, gcn/metrics.py:4-11 and the
    gradients TensorFlow derives from them, gcn/models.py:58) evaluated in fp64,
    together with the yardstick fp32 parity is measured against: for every
    result a bound ``M`` on the summed magnitude of the terms it is made of.
    A product C = A B of inexact operands gets M(C) = M(A) M(B) (a leaf has
    M = |value|); relu and dropout pass M through (a relu that would zero an
    element does not zero its M: the fp32 value may sit on the other side of
    zero by a rounding error).  Where the relu gate of the backward pass is
    itself within rounding of zero (|A P1| <= 1e-5 M), a flipped gate changes the
    element by its full size; those elements carry that size / 1e-5 in M, so that
    a bound of 1e-5 M admits the flip.

    ``a_hat``: scipy CSR whose values are the fp32 values of the matrix;
    ``m1`` [N, F] / ``m2`` [N, H]: uint8 dropout masks (None with keep = 1);
    weight decay is not included (it is applied inside Adam).  Returns a dict
    name -> (value, M) for "logits", "xent", "g1", "g2"."""
    a = a_hat.astype(np.float64).tocsr()
    a_abs = abs(a)
    at, at_abs = a.T.tocsr(), a_abs.T.tocsr()
    x, w1, w2 = (np.asarray(v, dtype=np.float64) for v in (x, w1, w2))
    y = np.asarray(labels, dtype=np.float64)
    msk = np.asarray(mask, dtype=np.float64)
    total = msk.sum()
    n = x.shape[0]
    if keep < 1.0:
        k1 = np.asarray(m1).reshape(x.shape) != 0
        xd = np.where(k1, x / keep, 0.0)
    else:
        xd = x
    p1, mp1 = xd @ w1, np.abs(xd) @ np.abs(w1)
    ap, map_ = a @ p1, a_abs @ mp1
    h = np.maximum(ap, 0.0)
    if keep < 1.0:
        k2 = np.asarray(m2).reshape(h.shape) != 0
        hd, mhd = np.where(k2, h / keep, 0.0), np.where(k2, map_ / keep, 0.0)
    else:
        k2 = np.ones(h.shape, dtype=bool)
        hd, mhd = h, map_
    p2, mp2 = hd @ w2, mhd @ np.abs(w2)
    z, mz = a @ p2, a_abs @ mp2
    zs = z - z.max(axis=1, keepdims=True)
    e = np.exp(zs)
    p = e / e.sum(axis=1, keepdims=True)
    logp = zs - np.log(e.sum(axis=1, keepdims=True))
    row_w = (msk / total)[:, None]
    mz_row = mz.max(axis=1, keepdims=True)
    xent = float(-(row_w * y * logp).sum())
    m_xent = float((row_w[:, 0] * (np.abs((y * logp).sum(axis=1)) + 2.0 * mz_row[:, 0])).sum())
    dz = row_w * (p - y * y.sum(axis=1, keepdims=True))
    mdz = row_w * (np.abs(p - y) + 2.0 * p * mz_row)
    dp2, mdp2 = at @ dz, at_abs @ mdz
    g2, mg2 = hd.T @ dp2, mhd.T @ mdp2
    dhd, mdhd = dp2 @ w2.T, mdp2 @ np.abs(w2).T
    gate = k2 & (ap > 0)
    dap = np.where(gate, dhd / keep, 0.0)
    mdap = np.where(k2, mdhd / keep, 0.0)
    unclear = k2 & (np.abs(ap) <= 1e-5 * map_)
    mdap = mdap + np.where(unclear, np.abs(dhd) / keep / 1e-5, 0.0)
    dp1, mdp1 = at @ dap, at_abs @ mdap
    g1, mg1 = xd.T @ dp1, np.abs(xd).T @ mdp1
    assert z.shape[0] == n
    return {"logits": (z, mz), "xent": (xent, m_xent), "g1": (g1, mg1), "g2": (g2, mg2),
            "unclear_gates": int(unclear.sum())}
