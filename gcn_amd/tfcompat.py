"""The slice of the TensorFlow-1.x surface that the reference's ``gcn`` package
and train scripts touch (gcn/layers.py, gcn/models.py, gcn/metrics.py,
gcn/inits.py, gcn/train_normal.py), re-hosted on the HIP kernels.

It keeps the reference's execution model -- build a graph of symbolic tensors
once from a ``placeholders`` dict, then ``sess.run(fetches, feed_dict)`` -- so a
train script written against the reference runs unchanged with
``import gcn_amd.tfcompat as tf``.  A symbolic tensor is a :class:`Node`; a run
evaluates the fetched nodes eagerly on the GPU, recording a torch autograd tape
when an optimizer step is among the fetches.  Sparse tuples in the feed are
turned into device CSR once and cached (the reference re-feeds them every step,
gcn/utils.py:143-144).

Only what the reference uses is provided; anything else raises AttributeError.
"""
import argparse
import collections
import contextlib
import os
import sys
import warnings

import numpy as np
import torch

from . import ops
from .sparse import Csr

float32 = "float32"
int32 = "int32"
int64 = "int64"
bool_ = "bool"

_DEVICE = None


def device():
    """The GPU this process computes on (``cuda:LOCAL_RANK``).  Raises when no
    GPU is visible: there is no CPU execution path."""
    global _DEVICE
    if _DEVICE is None:
        if not torch.cuda.is_available():
            raise RuntimeError("gcn_amd needs an AMD GPU (none visible); there is no CPU path")
        _DEVICE = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
        torch.cuda.set_device(_DEVICE)
    return _DEVICE


# --------------------------------------------------------------------- flags
class _FlagValues(object):
    """Process-global flag registry (``tf.app.flags.FLAGS``): defined by the
    entry script, read inside the library (reference gcn/models.py:95,102,114,
    142,149,162).  Parsed from ``sys.argv`` on first attribute access."""

    def __init__(self):
        object.__setattr__(self, "_defs", {})
        object.__setattr__(self, "_values", {})
        object.__setattr__(self, "_parsed", False)

    def _define(self, name, default, help_, typ):
        self._defs[name] = (default, help_, typ)
        self._values[name] = default
        object.__setattr__(self, "_parsed", False)

    def _parse(self, argv=None):
        ap = argparse.ArgumentParser(add_help=False)
        for name, (default, help_, typ) in self._defs.items():
            ap.add_argument("--" + name, type=typ, default=self._values[name], help=help_)
        ns, _ = ap.parse_known_args(sys.argv[1:] if argv is None else argv)
        self._values.update(vars(ns))
        object.__setattr__(self, "_parsed", True)

    def __getattr__(self, name):
        if name not in self._defs:
            raise AttributeError("flag %r is not defined" % name)
        if not self._parsed:
            self._parse()
        return self._values[name]

    def __setattr__(self, name, value):
        if name not in self._defs:
            raise AttributeError("flag %r is not defined" % name)
        self._values[name] = value

    def __contains__(self, name):
        return name in self._defs


def _str2bool(s):
    return str(s).lower() in ("1", "true", "t", "yes", "y")


class _Flags(object):
    FLAGS = _FlagValues()

    def DEFINE_string(self, name, default, help_=""):
        self.FLAGS._define(name, default, help_, str)

    def DEFINE_float(self, name, default, help_=""):
        self.FLAGS._define(name, default, help_, float)

    def DEFINE_integer(self, name, default, help_=""):
        self.FLAGS._define(name, default, help_, int)

    def DEFINE_boolean(self, name, default, help_=""):
        self.FLAGS._define(name, default, help_, _str2bool)

    DEFINE_bool = DEFINE_boolean


class _App(object):
    flags = _Flags()


app = _App()
flags = app.flags


# ---------------------------------------------------------------- graph nodes
class Node(object):
    """Symbolic tensor: ``fn(run, *evaluated inputs)`` computes its value."""

    def __init__(self, fn, inputs=(), name=None, static_shape=None):
        self.fn = fn
        self.inputs = tuple(inputs)
        self.name = name
        self.static_shape = static_shape

    def get_shape(self):
        return _Shape(self.static_shape)

    # scalar arithmetic used when the loss is assembled (gcn/models.py:146-153)
    def __add__(self, other):
        return _Sum([self, other])

    def __radd__(self, other):
        return _Sum([other, self])

    def __mul__(self, other):
        return Node(lambda run, a, b: a * b, (self, other))

    __rmul__ = __mul__

    def __hash__(self):
        return id(self)

    def __eq__(self, other):
        return self is other


class _Sum(Node):
    """Sum of loss terms.  Nested sums are flattened and literal zeros dropped, so
    ``0 + wd * l2_loss(w) + cross_entropy`` (reference gcn/models.py:146-153) is
    one node whose terms the optimizer can inspect: weight-decay terms
    (:class:`L2Term`) are then evaluated by ``gcnx_l2_loss_f32`` chained onto the
    other terms, and their gradient is applied inside the Adam kernel."""

    def __init__(self, terms):
        flat = []
        for t in terms:
            if isinstance(t, _Sum):
                flat.extend(t.terms)
            elif isinstance(t, (int, float)) and not isinstance(t, bool) and t == 0:
                continue
            else:
                flat.append(t)
        self.terms = flat
        self.l2_terms = [t for t in flat if isinstance(t, L2Term)]
        plain = [t for t in flat if not isinstance(t, L2Term)]
        super(_Sum, self).__init__(self._eval, plain, "sum")

    def without(self, dropped):
        """The sum of the remaining terms (None when nothing remains)."""
        rest = [t for t in self.terms if not any(t is d for d in dropped)]
        if not rest:
            return None
        return rest[0] if len(rest) == 1 else _Sum(rest)

    def _eval(self, run, *vals):
        acc = None
        for v in vals:
            acc = v if acc is None else acc + v
        for t in self.l2_terms:
            if run.folds(t):
                # value only (no tape): chained onto the running sum by the kernel
                # when that is a device scalar, so no scalar add kernels are needed
                w = run.value(t.var)
                if isinstance(acc, torch.Tensor) and acc.is_cuda and acc.dtype == torch.float32 and acc.numel() == 1:
                    acc = ops.l2_loss(w, t.coef, add=acc.reshape(1)).reshape(())
                else:
                    v = ops.l2_loss(w, t.coef).reshape(())
                    acc = v if acc is None else acc + v
            else:
                v = run.value(t)
                acc = v if acc is None else acc + v
        return acc if acc is not None else 0.0


class L2Term(Node):
    """``coef * tf.nn.l2_loss(var)`` = coef * sum(var ** 2) / 2."""

    def __init__(self, var, coef=1.0):
        super(L2Term, self).__init__(self._eval, (var,), "l2_loss")
        self.var = var
        self.coef = float(coef)

    def _eval(self, run, w):
        # stand-alone use (not folded into an optimizer step): differentiable
        return _L2Fn.apply(w, self.coef)

    def __mul__(self, other):
        if isinstance(other, (int, float)) and not isinstance(other, bool):
            return L2Term(self.var, self.coef * float(other))
        return Node.__mul__(self, other)

    __rmul__ = __mul__


class _L2Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, w, coef):
        ctx.save_for_backward(w)
        ctx.coef = coef
        return ops.l2_loss(w.contiguous(), coef).reshape(())

    @staticmethod
    def backward(ctx, dy):
        (w,) = ctx.saved_tensors
        return w * (dy * ctx.coef), None


class _Shape(object):
    def __init__(self, shape):
        self._shape = shape

    def as_list(self):
        if self._shape is None:
            raise ValueError("shape is unknown")
        return list(self._shape)


class Placeholder(Node):
    def __init__(self, dtype, shape=None, default=None, sparse=False, name=None):
        super(Placeholder, self).__init__(None, (), name, tuple(shape) if shape is not None else None)
        self.dtype = dtype
        self.default = default
        self.sparse = sparse


def placeholder(dtype, shape=None, name=None):
    return Placeholder(dtype, shape, name=name)


def placeholder_with_default(input, shape, name=None):
    return Placeholder(float32, shape, default=input, name=name)


def sparse_placeholder(dtype, shape=None, name=None):
    if isinstance(shape, Node):
        shape = None
    return Placeholder(dtype, shape, sparse=True, name=name)


def constant(value, dtype=None, name=None):
    return np.asarray(value)


class SparseValue(object):
    """A fed sparse tensor on the device, with an optional pending dropout mask
    (``sparse_dropout`` does not compact: dropped entries are zeroed inside the
    product that consumes the tensor)."""

    def __init__(self, csr, mask=None, scale=1.0, source=None):
        self.csr = csr
        self.mask = mask
        self.scale = scale
        self.source = source  # the host tuple it was fed as


# ------------------------------------------------------------------ variables
_GLOBAL_VARIABLES = []
_SCOPE = []
_SEED = [None]
_RNG_CALLS = [0]
_VAR_EPOCH = [0]   # bumped whenever a variable gets new storage (recorded graphs go stale)


class GraphKeys(object):
    GLOBAL_VARIABLES = "variables"


class Variable(Node):
    """Trainable fp32 tensor, materialised on the device by
    ``global_variables_initializer``; ``initial`` is a callable returning a numpy
    array (so that parity runs can inject weights with :meth:`load`)."""

    def __init__(self, initial, name=None):
        super(Variable, self).__init__(None, (), None)
        scope = "/".join(_SCOPE)
        self.name = (scope + "/" if scope else "") + (name or "Variable") + ":0"
        self._initial = initial
        self.tensor = None
        self.adam_m = None
        self.adam_v = None
        arr = initial() if callable(initial) else np.asarray(initial)
        self._init_value = np.ascontiguousarray(arr, dtype=np.float32)
        self.static_shape = self._init_value.shape
        _GLOBAL_VARIABLES.append(self)

    def initialize(self):
        _VAR_EPOCH[0] += 1
        self.tensor = torch.from_numpy(self._init_value.copy()).to(device()).requires_grad_(True)
        self.adam_m = torch.zeros_like(self.tensor, requires_grad=False)
        self.adam_v = torch.zeros_like(self.tensor, requires_grad=False)

    def load(self, value, sess=None):
        """Overwrite the value (resets the Adam moments)."""
        self._init_value = np.ascontiguousarray(value, dtype=np.float32).reshape(self._init_value.shape)
        self.initialize()

    def numpy(self):
        return self.tensor.detach().cpu().numpy()


def get_collection(key, scope=None):
    if key != GraphKeys.GLOBAL_VARIABLES:
        return []
    if scope is None:
        return list(_GLOBAL_VARIABLES)
    return [v for v in _GLOBAL_VARIABLES if v.name.startswith(scope + "/") or v.name.startswith(scope)]


@contextlib.contextmanager
def variable_scope(name):
    _SCOPE.append(name)
    try:
        yield
    finally:
        _SCOPE.pop()


@contextlib.contextmanager
def name_scope(name):
    yield


def set_random_seed(seed):
    _SEED[0] = int(seed)
    _RNG_CALLS[0] = 0


def _next_seed():
    """A fresh Philox key per random op instance, derived from the graph seed."""
    _RNG_CALLS[0] += 1
    base = _SEED[0] if _SEED[0] is not None else int.from_bytes(os.urandom(4), "little")
    return (base * 1000003 + _RNG_CALLS[0]) & (2 ** 63 - 1)


def random_uniform(shape, minval=0.0, maxval=1.0, dtype=float32, seed=None):
    """Initialiser form only (reference gcn/inits.py:7,14): returns a callable
    drawing the array on the host.  TensorFlow's own random stream cannot be
    reproduced; parity runs inject weights instead."""
    s = _next_seed()

    def draw():
        rng = np.random.RandomState(s % (2 ** 32))
        return rng.uniform(minval, maxval, size=tuple(int(d) for d in shape)).astype(np.float32)
    return draw


def zeros(shape, dtype=float32):
    return lambda: np.zeros(tuple(int(d) for d in shape), dtype=np.float32)


def ones(shape, dtype=float32):
    return lambda: np.ones(tuple(int(d) for d in shape), dtype=np.float32)


def global_variables_initializer():
    def init(run):
        for v in _GLOBAL_VARIABLES:
            if v.tensor is None:
                v.initialize()
        return None
    return Node(init, (), "init")


_RESET_HOOKS = []


def on_reset(fn):
    """Register a callable run by :func:`reset_default_graph` (modules that keep
    per-graph caches of device objects clear them there)."""
    _RESET_HOOKS.append(fn)
    return fn


def reset_default_graph():
    del _GLOBAL_VARIABLES[:]
    del _SCOPE[:]
    for fn in _RESET_HOOKS:
        fn()


# ------------------------------------------------------------------------ ops
def _is_sparse(v):
    return isinstance(v, SparseValue)


def sparse_tensor_dense_matmul(sp_a, b):
    def fn(run, a, bb):
        if not _is_sparse(a):
            raise TypeError("sparse_tensor_dense_matmul: first operand must be sparse")
        return ops.spmm_autograd(a.csr, bb.contiguous(), a.mask, a.scale)
    return Node(fn, (sp_a, b))


def matmul(a, b):
    return Node(lambda run, x, y: ops.gemm_autograd(x.contiguous(), y.contiguous()), (a, b))


def sparse_retain_scaled(x, keep_prob, seed_key):
    """sparse_retain(x, floor(keep + U)) * (1/keep) as one lazy step
    (reference gcn/layers.py:21-27)."""
    def fn(run, xv, keep):
        keep = float(keep)
        if keep >= 1.0:
            return xv
        if xv.mask is not None:
            raise NotImplementedError("sparse dropout applied twice to one tensor")
        mask = run.dropout_mask(seed_key, xv.csr.nnz, keep)
        return SparseValue(xv.csr, mask, xv.scale / keep, xv.source)
    return Node(fn, (x, keep_prob))


class _NN(object):
    @staticmethod
    def relu(x):
        return Node(lambda run, v: torch.relu(v), (x,)) if isinstance(x, Node) else torch.relu(x)

    @staticmethod
    def dropout(x, keep_prob, seed_key=None):
        key = seed_key if seed_key is not None else _next_seed()

        def fn(run, v, keep):
            keep = float(keep)
            if keep >= 1.0:
                return v
            mask = run.dropout_mask(key, v.numel(), keep)
            return ops.dropout_dense_autograd(v, mask, 1.0 / keep)
        return Node(fn, (x, keep_prob))

    @staticmethod
    def softmax(x):
        return Node(lambda run, v: torch.softmax(v, dim=1), (x,))

    @staticmethod
    def l2_loss(var):
        return L2Term(var)

    @staticmethod
    def softmax_cross_entropy_with_logits(logits=None, labels=None):
        raise NotImplementedError("use gcn_amd.metrics.masked_softmax_cross_entropy")


nn = _NN()


def add_n(xs):
    xs = list(xs)
    if len(xs) == 1:
        return xs[0]

    def fn(run, *vs):
        out = vs[0]
        for v in vs[1:]:
            out = out + v
        return out
    return Node(fn, xs)


class _Summary(object):
    @staticmethod
    def histogram(name, value):
        return None


summary = _Summary()


# ------------------------------------------------------------------ optimizer
class _OptOp(Node):
    """``optimizer.minimize(loss)``.  Weight-decay terms of the loss
    (``coef * l2_loss(variable)`` added at the top level) are split off: their
    gradient ``coef * w`` is added inside the Adam kernel (``decay``), and only
    ``rest`` -- the remaining terms -- goes through the autograd tape."""

    def __init__(self, optimizer, loss):
        super(_OptOp, self).__init__(None, (), "opt_op")
        self.optimizer = optimizer
        self.loss = loss
        self.folded = []
        self.rest = loss
        if isinstance(loss, _Sum):
            self.folded = [t for t in loss.l2_terms if isinstance(t.var, Variable)]
            self.rest = loss.without(self.folded)
        self.decay = {}
        for t in self.folded:
            self.decay[t.var] = self.decay.get(t.var, 0.0) + t.coef


class AdamOptimizer(object):
    """tf.train.AdamOptimizer with TF's defaults and update rule
    (reference gcn/models.py:142): lr_t = lr*sqrt(1-b2^t)/(1-b1^t),
    w -= lr_t * m / (sqrt(v) + eps)."""

    def __init__(self, learning_rate=0.001, beta1=0.9, beta2=0.999, epsilon=1e-8):
        self.learning_rate = learning_rate
        self.beta1, self.beta2, self.epsilon = beta1, beta2, epsilon
        self.step = 0          # host copy, informational
        self._step_dev = None  # the count the kernels read (advanced on the stream)

    def minimize(self, loss):
        return _OptOp(self, loss)

    def apply(self, variables, grads, decay=None):
        """One update of every variable; ``decay`` maps a variable to the
        coefficient of its weight-decay term (gradient added in the kernel).  The step count lives in device memory
        and is advanced by a kernel, so that the launches can be recorded in a
        HIP graph and replayed."""
        if self._step_dev is None:
            self._step_dev = ops.device_counter(device(), self.step)
        if not torch.cuda.is_current_stream_capturing():
            self.step += 1
        ops.counter_add(self._step_dev, 1)
        decay = decay or {}
        for v, g in zip(variables, grads):
            wd = decay.get(v, 0.0)
            if g is None:
                if not wd:
                    continue
                g = torch.zeros_like(v.tensor)     # a variable reached by its decay term only
            with torch.no_grad():
                ops.adam_step(v.tensor, g.contiguous(), v.adam_m, v.adam_v, self.learning_rate,
                              self._step_dev, self.beta1, self.beta2, self.epsilon, weight_decay=wd)


class _Saver(object):
    def __init__(self, var_dict):
        self.vars = dict(var_dict) if isinstance(var_dict, dict) else {v.name: v for v in var_dict}

    def save(self, sess, path):
        d = os.path.dirname(path)
        if d and not os.path.isdir(d):
            os.makedirs(d)
        torch.save({k: v.tensor.detach().cpu() for k, v in self.vars.items()}, path)
        return path

    def restore(self, sess, path):
        state = torch.load(path, weights_only=True)
        for k, v in self.vars.items():
            v.load(state[k].numpy())


class _Train(object):
    AdamOptimizer = AdamOptimizer
    Saver = _Saver


train = _Train()


# -------------------------------------------------------------------- session
class _Run(object):
    """State of one ``sess.run``: feed values, memo, dropout masks."""

    def __init__(self, sess, feed, folded=()):
        self.sess = sess
        self.feed = feed
        self.memo = {}
        self._folded = set(id(t) for t in folded)

    def folds(self, l2_term):
        """Whether this run takes the value of a weight-decay term without a
        tape: an optimizer step of this run applies its gradient inside Adam, or
        nothing is differentiated at all."""
        return id(l2_term) in self._folded or not torch.is_grad_enabled()

    def dropout_mask(self, key, n, keep):
        provider = self.sess.mask_provider
        if provider is not None:
            m = provider(key, n, keep)
            if m is not None:
                return torch.as_tensor(np.ascontiguousarray(m, dtype=np.uint8)).to(device())
        return ops.bernoulli_mask(n, keep, key, 0, device(), step=self.sess._run_dev,
                                  step_stride=2 ** 40)

    def value(self, node):
        if isinstance(node, np.ndarray) and node.ndim >= 1:
            # a numpy array used where a tensor is expected (e.g. the output of
            # one sess.run passed into the next graph piece) becomes a constant
            key = id(node)
            hit = self.sess._cache.get(key)
            if hit is None or hit[0] is not node:
                hit = (node, torch.from_numpy(np.ascontiguousarray(node, dtype=np.float32)).to(device()))
                self.sess._remember(key, hit)
            return hit[1]
        if not isinstance(node, Node):
            return node
        if node in self.memo:
            return self.memo[node]
        if isinstance(node, Placeholder):
            if node in self.feed:
                v = self.sess.to_device(node, self.feed[node])
            elif node.default is not None:
                v = node.default
            else:
                raise ValueError("placeholder %r was not fed" % (node.name,))
        elif isinstance(node, Variable):
            if node.tensor is None:
                raise RuntimeError("variable %s is not initialised" % node.name)
            v = node.tensor
        else:
            v = node.fn(self, *[self.value(i) for i in node.inputs])
        self.memo[node] = v
        return v


class Session(object):
    """``config={"hip_graph": True}`` (or the environment variable
    GCN_AMD_HIP_GRAPH=1) records each distinct (fetches, feed objects) call in a
    HIP graph on its second occurrence and replays it from then on: a Planetoid
    step is ~60 launches of microsecond kernels, so launch latency, not the
    kernels, sets its time.  The first occurrence runs eagerly and builds the
    cached device state (CSR uploads, transposes, merged supports).  Replays
    require that the same Python objects are fed (the reference's training
    loop does: gcn/train_normal.py:89-93); the values of fed arrays are read at
    upload time only, exactly as without graphs."""

    # fed objects whose device copies are kept (least recently used dropped first);
    # a training loop feeds a handful of objects over and over, a caller that
    # feeds fresh arrays every step must not grow device memory without bound
    CACHE_ENTRIES = 64

    def __init__(self, config=None):
        self._cache = collections.OrderedDict()
        self.run_counter = 0
        self._run_dev = None   # device copy of run_counter: the dropout kernels read it
        # hook: mask_provider(key, n, keep) -> uint8 array or None; lets a
        # parity run share dropout masks with the oracle
        self.mask_provider = None
        cfg = config if isinstance(config, dict) else {}
        self.hip_graph = bool(cfg.get("hip_graph", os.environ.get("GCN_AMD_HIP_GRAPH", "0") == "1"))
        self._graphs = {}
        self.graph_replays = 0

    def to_device(self, ph, value):
        """Upload a fed value once; later feeds of the same object reuse it."""
        if isinstance(value, (SparseValue, torch.Tensor)):
            return value
        if not ph.sparse and isinstance(value, (tuple, list, int, float)):
            return value  # e.g. num_features_nonzero = (nnz,), dropout = 0.5
        key = id(value)
        hit = self._cache.get(key)
        if hit is not None and hit[0] is value:
            self._cache.move_to_end(key)
            return hit[1]
        if ph.sparse:
            out = SparseValue(Csr.from_tuple(value, device()), source=value)
        else:
            arr = np.asarray(value)
            if arr.ndim == 0:
                out = arr[()].item()
            elif arr.dtype == np.bool_ or np.issubdtype(arr.dtype, np.integer):
                out = torch.from_numpy(np.ascontiguousarray(arr != 0).astype(np.uint8)).to(device())
                out.mask_total = float(np.count_nonzero(arr))   # read by the masked loss
            else:
                out = torch.from_numpy(np.ascontiguousarray(arr, dtype=np.float32)).to(device())
        self._remember(key, (value, out))
        return out

    def _remember(self, key, entry):
        self._cache[key] = entry
        self._cache.move_to_end(key)
        while len(self._cache) > self.CACHE_ENTRIES:
            self._cache.popitem(last=False)

    def _execute(self, flist, feed):
        """Issue every launch of one run on the current stream (no host
        synchronisation once the fed values are resident); returns the fetched
        values as device tensors / Python objects."""
        ops.counter_add(self._run_dev, 1)
        opt_ops = [f for f in flist if isinstance(f, _OptOp)]
        run = _Run(self, feed, [t for op in opt_ops for t in op.folded])
        results = [None] * len(flist)
        with torch.set_grad_enabled(bool(opt_ops)):
            for i, f in enumerate(flist):
                if not isinstance(f, _OptOp):
                    results[i] = run.value(f)
            for op in opt_ops:
                variables = [v for v in _GLOBAL_VARIABLES if v.tensor is not None]
                loss = run.value(op.rest) if op.rest is not None else None
                if isinstance(loss, torch.Tensor) and loss.requires_grad:
                    grads = torch.autograd.grad(loss, [v.tensor for v in variables], allow_unused=True)
                else:
                    grads = [None] * len(variables)
                op.optimizer.apply(variables, grads, op.decay)
        return [r.detach() if isinstance(r, torch.Tensor) else r for r in results]

    @staticmethod
    def _graph_key(flist, feed):
        sig = []
        for ph, v in feed.items():
            if isinstance(v, (bool, int, float, np.integer, np.floating)):
                sig.append((id(ph), "s", float(v)))
            elif isinstance(v, np.ndarray) and v.ndim == 0:
                sig.append((id(ph), "s", float(v)))
            elif isinstance(v, (tuple, list)) and len(v) <= 8 and \
                    all(isinstance(e, (bool, int, float, np.integer, np.floating)) for e in v):
                sig.append((id(ph), "t", tuple(float(e) for e in v)))  # e.g. num_features_nonzero
            else:
                sig.append((id(ph), "o", id(v)))
        sig.sort()
        return (tuple(id(f) for f in flist), tuple(sig), _VAR_EPOCH[0], len(_GLOBAL_VARIABLES))

    def run(self, fetches, feed_dict=None, options=None, run_metadata=None):
        single = not isinstance(fetches, (list, tuple))
        flist = [fetches] if single else list(fetches)
        device()
        self.run_counter += 1
        if self._run_dev is None:
            self._run_dev = ops.device_counter(device(), self.run_counter - 1)
        feed = feed_dict or {}
        ent = None
        if self.hip_graph and self.mask_provider is None:
            key = self._graph_key(flist, feed)
            ent = self._graphs.get(key)
            if ent is None:
                # first occurrence: eager (it also warms every cache the graph relies on);
                # the fed objects are kept alive so that their ids stay unique
                if len(self._graphs) >= 64:
                    # a caller that feeds fresh objects every step never replays:
                    # forget the calls seen once (oldest first) instead of growing
                    for old_key in [k_ for k_, v_ in self._graphs.items() if "graph" not in v_][:32]:
                        del self._graphs[old_key]
                self._graphs[key] = {"refs": (list(flist), list(feed.items()))}
            elif "graph" not in ent and not ent.get("eager_only"):
                from .sparse import WORKSPACE
                graph = torch.cuda.CUDAGraph()
                torch.cuda.synchronize()
                try:
                    with torch.cuda.graph(graph):
                        ent["results"] = self._execute(flist, feed)
                except RuntimeError as exc:
                    # something in this call cannot be recorded (a host read-back, an
                    # allocation outside the graph pool): nothing of the recording ran;
                    # run this call eagerly from now on
                    ent.pop("results", None)
                    ent["eager_only"] = True
                    warnings.warn("hip_graph: this session.run is not recordable, running it eagerly (%s)" % (exc,))
                else:
                    ent["graph"] = graph
                    ent["workspace"] = WORKSPACE.buffers()
                    # the recording reads the device copies of the fed objects: keep them
                    # alive with the graph even if the session cache drops them later
                    ent["device_refs"] = [self._cache[id(v)] for v in feed.values() if id(v) in self._cache]
        if ent is not None and "graph" in ent:
            ent["graph"].replay()
            self.graph_replays += 1
            for f in flist:
                if isinstance(f, _OptOp):
                    f.optimizer.step += 1
            results = ent["results"]
        else:
            results = self._execute(flist, feed)
        out = []
        for r in results:
            if isinstance(r, torch.Tensor):
                r = r.detach().cpu().numpy()
                if r.ndim == 0:
                    r = r[()]
            out.append(r)
        return out[0] if single else out

    def close(self):
        self._cache.clear()
        self._graphs.clear()    # recorded HIP graphs and the memory pools they hold

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()
