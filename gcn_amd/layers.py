"""Layers with the reference's constructor signatures, attribute names and
auto-naming (reference gcn/layers.py), computing through the HIP kernels.

What a layer builds is one fused step per stage instead of the reference's
chain of TensorFlow ops:
  * sparse dropout is a mask applied inside the product that reads the tensor
    (no compaction, no dropped-out copy of X);
  * the loop over supports plus ``add_n`` (gcn/layers.py:213-221) is ONE
    transform ``x . [W_0 | ... | W_S-1]`` followed by ONE SpMM over the supports
    merged on interleaved columns, accumulating across supports in registers;
  * ``act=tf.nn.relu`` is applied in the epilogue of the producing kernel.
The free functions ``dot`` and ``sparse_dropout`` keep their stand-alone
meaning for code that composes them by hand.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from . import ops
from . import tfcompat as tf
from .inits import *  # noqa: F401,F403  (the reference re-exports the initialisers)
from .inits import glorot, zeros
from .sparse import Csr, _ptr, _stream

flags = tf.app.flags
FLAGS = flags.FLAGS

# global unique layer ID dictionary for layer name assignment
_LAYER_UIDS = {}


def get_layer_uid(layer_name=''):
    """1, 2, 3, ... per layer class name (reference gcn/layers.py:11-18)."""
    _LAYER_UIDS[layer_name] = _LAYER_UIDS.get(layer_name, 0) + 1
    return _LAYER_UIDS[layer_name]


def sparse_dropout(x, keep_prob, noise_shape):
    """Dropout on the nonzeros of a sparse tensor: keep each with probability
    ``keep_prob``, scale survivors by 1/keep_prob (reference gcn/layers.py:21-27).
    ``noise_shape`` (the nnz) is implied by ``x`` and only kept for the
    signature."""
    return tf.sparse_retain_scaled(x, keep_prob, tf._next_seed())


def dot(x, y, sparse=False):
    """x . y with x sparse or dense (reference gcn/layers.py:30-36)."""
    if sparse:
        return tf.sparse_tensor_dense_matmul(x, y)
    return tf.matmul(x, y)


# ----------------------------------------------------------------------------
# fused building blocks shared by Dense and GraphConvolution
# ----------------------------------------------------------------------------
def _is_relu(act):
    return act is tf.nn.relu


def _transform(run, x, w, keep, key, relu):
    """act(dropout(x) . w) for sparse or dense x, dropout fused."""
    keep = float(keep)
    w = w.contiguous()
    if isinstance(x, tf.SparseValue):
        mask, scale = x.mask, x.scale
        if keep < 1.0:
            if mask is not None:
                raise NotImplementedError("sparse dropout applied twice to one tensor")
            mask = run.dropout_mask(key, x.csr.nnz, keep)
            scale = scale / keep
        return ops.spmm_autograd(x.csr, w, mask, scale, relu)
    if keep < 1.0:
        mask = run.dropout_mask(key, x.numel(), keep)
        x = ops.dropout_dense_autograd(x, mask, 1.0 / keep)
    return ops.gemm_autograd(x.contiguous(), w, relu)


_MERGED = {}
tf.on_reset(_MERGED.clear)


def _merged_support(supports):
    """One CSR for sum_i A_i . P_i: entries of support i moved to column
    ``col * S + i``, so the dense operand is ``P_cat`` viewed as [N*S, out].
    Built once per list of supports (they stay resident across steps)."""
    key = tuple(id(s.csr) for s in supports)
    hit = _MERGED.get(key)
    if hit is not None and all(a is b.csr for a, b in zip(hit[0], supports)):
        return hit[1]
    lib = _lib.load()
    dev = supports[0].csr.device
    n_sup = len(supports)
    n_rows, n_cols = supports[0].csr.shape
    if n_cols * n_sup >= 2 ** 31:
        raise ValueError("merged support has too many columns for int32 indices")
    coords, vals = [], []
    f64 = all(np.asarray(s.source[1]).dtype == np.float64 for s in supports)
    for s in supports:
        if s.csr.shape != (n_rows, n_cols):
            raise ValueError("supports differ in shape")
        c, v, _ = s.source
        coords.append(torch.from_numpy(np.ascontiguousarray(np.asarray(c).reshape(-1, 2), dtype=np.int32)).to(dev))
        vals.append(torch.from_numpy(np.ascontiguousarray(v, dtype=np.float64 if f64 else np.float32)).to(dev))
    nnz = [int(v.numel()) for v in vals]
    total = sum(nnz)
    out_row = torch.empty(total, dtype=torch.int32, device=dev)
    out_col = torch.empty(total, dtype=torch.int32, device=dev)
    out_val = torch.empty(total, dtype=torch.float64 if f64 else torch.float32, device=dev)
    h_nnz = (C.c_int64 * n_sup)(*nnz)
    h_row = (C.c_void_p * n_sup)(*[c.data_ptr() for c in coords])
    h_col = (C.c_void_p * n_sup)(*[c.data_ptr() + 4 for c in coords])
    h_val = (C.c_void_p * n_sup)(*[v.data_ptr() for v in vals])
    with torch.cuda.device(dev):
        _lib.check(lib.gcnx_coo_interleave_supports(n_sup, h_nnz, h_row, h_col, 2, h_val,
                                                    1 if f64 else 0, _ptr(out_row), _ptr(out_col),
                                                    _ptr(out_val), _stream()))
    merged = Csr.from_coo(out_row, out_col, out_val, (n_rows, n_cols * n_sup), dev)
    _MERGED[key] = ([s.csr for s in supports], merged)
    return merged


_IDENTITY_CHECKED = {}
tf.on_reset(_IDENTITY_CHECKED.clear)


def _is_identity(csr):
    hit = _IDENTITY_CHECKED.get(id(csr))
    if hit is not None and hit[0] is csr:
        return hit[1]
    ok = csr.shape[0] == csr.shape[1] == csr.nnz
    if ok:
        ar = torch.arange(csr.nnz, dtype=torch.int32, device=csr.col_idx.device)
        ok = bool(torch.equal(csr.col_idx, ar)) and bool((csr.val == 1.0).all().item())
    _IDENTITY_CHECKED[id(csr)] = (csr, ok)
    return ok


def _chebyshev_recurrence():
    return 'chebyshev_recurrence' in FLAGS and bool(FLAGS.chebyshev_recurrence)


def _aggregate_clenshaw(supports, p, n_out, relu):
    """sum_k T_k(L) . P_k without the matrices T_2..T_K: Clenshaw's recurrence

        b_k = P_k + 2 L b_{k+1} - b_{k+2}   (k = K..1),   out = P_0 + L b_1 - b_2

    with L = supports[1] (the scaled Laplacian, reference gcn/utils.py:157-158)
    and T_0 = I, T_k = 2 L T_{k-1} - T_{k-2} (gcn/utils.py:160-167).  K products
    over the pattern of L (nnz of A_hat) replace one product over the union of
    the growing patterns of T_0..T_K (35x the nonzeros for Cora, K = 3).  The
    sum is the same polynomial evaluated in fp32 on the device instead of fp64
    on the host followed by one fp32 product, so results agree to rounding, not
    to the bit; this mode is opt-in (FLAGS.chebyshev_recurrence)."""
    if not _is_identity(supports[0].csr):
        raise ValueError("chebyshev_recurrence: support[0] is not the identity")
    lap = supports[1].csr
    k_max = len(supports) - 1
    part = lambda k: p[:, k * n_out:(k + 1) * n_out]
    if k_max == 1:
        return ops.spmm_add_autograd(lap, part(1), part(0), 1.0, relu)
    b2 = part(k_max)                                              # b_K
    b1 = ops.spmm_add_autograd(lap, b2, part(k_max - 1), 2.0)      # b_{K-1}
    for k in range(k_max - 2, 0, -1):
        b1, b2 = ops.spmm_add_autograd(lap, b1, part(k) - b2, 2.0), b1
    return ops.spmm_add_autograd(lap, b1, part(0) - b2, 1.0, relu)


def _aggregate(run, supports, p, n_out, relu):
    """act(sum_i A_i . P_i) with P_i = p[:, i*n_out:(i+1)*n_out]."""
    if len(supports) == 1:
        s = supports[0]
        return ops.spmm_autograd(s.csr, p.contiguous(), s.mask, s.scale, relu)
    for s in supports:
        if s.mask is not None:
            raise NotImplementedError("dropout on a support matrix")
    if _chebyshev_recurrence():
        return _aggregate_clenshaw(supports, p, n_out, relu)
    merged = _merged_support(supports)
    n = p.shape[0]
    return ops.spmm_autograd(merged, p.contiguous().view(n * len(supports), n_out), None, 1.0, relu)


class Layer(object):
    """Base layer class (reference gcn/layers.py:39-82).

    # Properties
        name: String, defines the variable scope of the layer.
        logging: Boolean, kept for the signature (histograms are not written)

    # Methods
        _call(inputs): Defines computation graph of layer
        __call__(inputs): Wrapper for _call()
    """

    def __init__(self, **kwargs):
        allowed_kwargs = {'name', 'logging'}
        for kwarg in kwargs.keys():
            assert kwarg in allowed_kwargs, 'Invalid keyword argument: ' + kwarg
        name = kwargs.get('name')
        if not name:
            layer = self.__class__.__name__.lower()
            name = layer + '_' + str(get_layer_uid(layer))
        self.name = name
        self.vars = {}
        self.logging = kwargs.get('logging', False)
        self.sparse_inputs = False

    def _call(self, inputs):
        return inputs

    def __call__(self, inputs):
        with tf.name_scope(self.name):
            return self._call(inputs)

    def _log_vars(self):
        for var in self.vars:
            tf.summary.histogram(self.name + '/vars/' + var, self.vars[var])


class Dense(Layer):
    """Dense layer: act(dropout(x) . W + b) (reference gcn/layers.py:85-129)."""

    def __init__(self, input_dim, output_dim, placeholders, dropout=0., sparse_inputs=False,
                 act=tf.nn.relu, bias=False, featureless=False, **kwargs):
        super(Dense, self).__init__(**kwargs)

        self.dropout = placeholders['dropout'] if dropout else 0.
        self.act = act
        self.sparse_inputs = sparse_inputs
        self.featureless = featureless
        self.bias = bias
        self.output_dim = output_dim

        # helper variable for sparse dropout
        self.num_features_nonzero = placeholders['num_features_nonzero']
        self.dropout_key = tf._next_seed()

        with tf.variable_scope(self.name + '_vars'):
            self.vars['weights'] = glorot([input_dim, output_dim], name='weights')
            if self.bias:
                self.vars['bias'] = zeros([output_dim], name='bias')

        if self.logging:
            self._log_vars()

    def _call(self, inputs):
        fuse = _is_relu(self.act) and not self.bias
        key = self.dropout_key

        def fn(run, x, w, drop, *b):
            out = _transform(run, x, w, 1.0 - float(drop), key, fuse)
            if b:
                out = out + b[0]
            return out if fuse else self.act(out)

        extra = (self.vars['bias'],) if self.bias else ()
        return tf.Node(fn, (inputs, self.vars['weights'], self.dropout) + extra)


class GraphConvolution(Layer):
    """Graph convolution layer: act(sum_i A_i . (dropout(x) . W_i) + b)
    (reference gcn/layers.py:132-226), including the fork's split
    ``_update`` / ``_aggregate`` stages and its optional ``stage_hook``."""

    def __init__(self, input_dim, output_dim, placeholders, dropout=0.,
                 sparse_inputs=False, act=tf.nn.relu, bias=False,
                 featureless=False, **kwargs):
        super(GraphConvolution, self).__init__(**kwargs)

        self.dropout = placeholders['dropout'] if dropout else 0.
        self.act = act
        self.support = placeholders['support']
        self.sparse_inputs = sparse_inputs
        self.featureless = featureless
        self.bias = bias
        self.output_dim = output_dim

        # helper variable for sparse dropout
        self.num_features_nonzero = placeholders['num_features_nonzero']
        self.dropout_key = tf._next_seed()

        with tf.variable_scope(self.name + '_vars'):
            for i in range(len(self.support)):
                self.vars['weights_' + str(i)] = glorot([input_dim, output_dim],
                                                        name='weights_' + str(i))
            if self.bias:
                self.vars['bias'] = zeros([output_dim], name='bias')

        if self.logging:
            self._log_vars()

    def _hook(self, stage):
        hook = globals().get('stage_hook')
        if hook is not None:
            hook(stage, self.name.split('_')[-1])  # graphconvolution_1 -> "1"

    def _update(self, inputs):
        """Update stage of the fork: act(dropout(x) . weights_0 + b)
        (reference gcn/layers.py:164-187)."""
        self._hook("UPDATE_BEGIN")
        fuse = _is_relu(self.act) and not self.bias
        key = self.dropout_key

        def fn(run, x, w, drop, *b):
            if self.featureless:
                out = torch.relu(w) if fuse else w
            else:
                out = _transform(run, x, w, 1.0 - float(drop), key, fuse)
            if b:
                out = out + b[0]
            return out if fuse else self.act(out)

        extra = (self.vars['bias'],) if self.bias else ()
        node = tf.Node(fn, (inputs, self.vars['weights_0'], self.dropout) + extra)
        self._hook("UPDATE_END")
        return node

    def _aggregate(self, inputs):
        """Aggregate stage of the fork: sum_i A_i . inputs
        (reference gcn/layers.py:189-201)."""
        self._hook("AGG_BEGIN")
        n_sup = len(self.support)

        def fn(run, x, *sups):
            x = x.contiguous()
            if n_sup == 1:
                return _aggregate(run, sups, x, x.shape[1], False)
            # the same dense operand for every support
            return _aggregate(run, sups, x.repeat(1, n_sup), x.shape[1], False)

        node = tf.Node(fn, (inputs,) + tuple(self.support))
        self._hook("AGG_END")
        return node

    def _call(self, inputs):
        n_sup = len(self.support)
        n_out = self.output_dim
        fuse = _is_relu(self.act) and not self.bias
        key = self.dropout_key
        weights = [self.vars['weights_' + str(i)] for i in range(n_sup)]

        def fn(run, x, drop, *rest):
            ws, sups = rest[:n_sup], rest[n_sup:2 * n_sup]
            b = rest[2 * n_sup:]
            w_cat = ws[0] if n_sup == 1 else torch.cat(ws, dim=1)
            if self.featureless:
                p = w_cat
            else:
                p = _transform(run, x, w_cat, 1.0 - float(drop), key, False)
            out = _aggregate(run, sups, p, n_out, fuse)
            if b:
                out = out + b[0]
            return out if fuse else self.act(out)

        extra = (self.vars['bias'],) if self.bias else ()
        return tf.Node(fn, (inputs, self.dropout) + tuple(weights) + tuple(self.support) + extra)
