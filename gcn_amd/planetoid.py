"""Non-executing reader for the Planetoid ``ind.<dataset>.<part>`` files.

The upstream loader (reference gcn/utils.py:44-53) feeds these files to
``pickle.load``, which runs whatever constructors the file names.  The files
are protocol-2 pickles of exactly three shapes (scipy CSR matrix, numpy
ndarray, defaultdict of int lists), so this module *decodes* them as a data
format instead: ``pickletools.genops`` tokenises the byte stream (it executes
nothing), and the small stack machine below rebuilds only plain data.  A
``GLOBAL`` opcode never imports or calls anything; it pushes an inert tag, and
only the four tags listed in ``_KNOWN`` are given a meaning.  Anything else in
the stream raises ``PlanetoidFormatError``.

Also provides the ``.npz`` container (numpy arrays only, ``allow_pickle=False``)
that the repo ships the datasets in.
"""
import os
import pickletools

import numpy as np
import scipy.sparse as sp


class PlanetoidFormatError(ValueError):
    pass


class _Tag(object):
    """Inert stand-in for a pickled global reference (never resolved)."""
    __slots__ = ("name",)

    def __init__(self, name):
        self.name = name

    def __repr__(self):
        return "<tag %s>" % self.name


class _Node(object):
    """Inert result of REDUCE/NEWOBJ on a tag: (tag, args, state)."""
    __slots__ = ("tag", "args", "state", "items", "listitems")

    def __init__(self, tag, args):
        self.tag = tag
        self.args = args
        self.state = None
        self.items = []      # SETITEM(S) on a dict-like node
        self.listitems = []  # APPEND(S) on a list-like node


_MARK = object()

_KNOWN = {
    "numpy.core.multiarray _reconstruct",
    "numpy ndarray",
    "numpy dtype",
    "scipy.sparse.csr csr_matrix",
    "collections defaultdict",
    "__builtin__ list",
}

_DTYPES = {"i1", "i2", "i4", "i8", "u1", "u2", "u4", "u8", "f4", "f8", "b1"}


def _raw(s):
    # pickletools hands Python-2 str payloads back decoded as latin-1.
    return s.encode("latin-1") if isinstance(s, str) else bytes(s)


def _finish(obj):
    """Turn inert nodes into numpy / scipy / dict data."""
    if isinstance(obj, _Node):
        name = obj.tag.name
        if name == "numpy dtype":
            code = obj.args[0]
            if code not in _DTYPES:
                raise PlanetoidFormatError("dtype %r not allowed" % (code,))
            st = obj.state
            order = st[1] if st is not None else "<"
            if order not in ("<", "|", "="):
                raise PlanetoidFormatError("byte order %r not supported" % (order,))
            return np.dtype(("<" if code[1] != "1" else "|") + code)
        if name == "numpy.core.multiarray _reconstruct":
            st = obj.state
            if st is None or len(st) != 5:
                raise PlanetoidFormatError("ndarray without state")
            _ver, shape, dtype, fortran, raw = st
            dtype = _finish(dtype)
            if not isinstance(dtype, np.dtype):
                raise PlanetoidFormatError("bad ndarray dtype")
            shape = tuple(int(s) for s in shape)
            arr = np.frombuffer(_raw(raw), dtype=dtype)
            if arr.size != int(np.prod(shape, dtype=np.int64)):
                raise PlanetoidFormatError("ndarray payload size mismatch")
            return arr.reshape(shape, order="F" if fortran else "C").copy()
        if name == "scipy.sparse.csr csr_matrix":
            st = obj.state
            if not isinstance(st, dict):
                raise PlanetoidFormatError("csr_matrix without dict state")
            data = _finish(st["data"])
            indices = _finish(st["indices"])
            indptr = _finish(st["indptr"])
            shape = tuple(int(s) for s in st["_shape"])
            return sp.csr_matrix((data, indices, indptr), shape=shape)
        if name == "collections defaultdict":
            out = {}
            for k, v in obj.items:
                out[int(k)] = [int(x) for x in _finish(v)]
            return out
        if name == "__builtin__ list":
            return list(obj.listitems)
        raise PlanetoidFormatError("unexpected object %s" % name)
    if isinstance(obj, list):
        return [_finish(o) for o in obj]
    return obj


def decode(data):
    """Decode one Planetoid part from its bytes.  Returns a scipy CSR matrix,
    a numpy array, or a ``dict`` {int: [int, ...]} in file order."""
    stack = []
    memo = {}

    def pop_mark():
        items = []
        while True:
            if not stack:
                raise PlanetoidFormatError("mark not found")
            top = stack.pop()
            if top is _MARK:
                break
            items.append(top)
        items.reverse()
        return items

    for op, arg, _pos in pickletools.genops(data):
        n = op.name
        if n == "PROTO":
            if arg > 2:
                raise PlanetoidFormatError("protocol %d not supported" % arg)
        elif n in ("GLOBAL",):
            if arg not in _KNOWN:
                raise PlanetoidFormatError("refusing global %r" % (arg,))
            stack.append(_Tag(arg))
        elif n in ("BINPUT", "LONG_BINPUT", "PUT"):
            memo[arg] = stack[-1]
        elif n in ("BINGET", "LONG_BINGET", "GET"):
            stack.append(memo[arg])
        elif n == "MARK":
            stack.append(_MARK)
        elif n in ("BININT", "BININT1", "BININT2", "INT", "LONG", "LONG1"):
            stack.append(int(arg))
        elif n in ("BINFLOAT", "FLOAT"):
            stack.append(float(arg))
        elif n in ("SHORT_BINSTRING", "BINSTRING", "STRING", "BINUNICODE", "UNICODE"):
            stack.append(arg)
        elif n == "NONE":
            stack.append(None)
        elif n == "NEWTRUE":
            stack.append(True)
        elif n == "NEWFALSE":
            stack.append(False)
        elif n == "EMPTY_TUPLE":
            stack.append(())
        elif n == "TUPLE1":
            stack[-1:] = [tuple(stack[-1:])]
        elif n == "TUPLE2":
            stack[-2:] = [tuple(stack[-2:])]
        elif n == "TUPLE3":
            stack[-3:] = [tuple(stack[-3:])]
        elif n == "TUPLE":
            stack.append(tuple(pop_mark()))
        elif n == "EMPTY_LIST":
            stack.append([])
        elif n == "EMPTY_DICT":
            stack.append({})
        elif n == "APPEND":
            v = stack.pop()
            tgt = stack[-1]
            (tgt.listitems if isinstance(tgt, _Node) else tgt).append(v)
        elif n == "APPENDS":
            vs = pop_mark()
            tgt = stack[-1]
            (tgt.listitems if isinstance(tgt, _Node) else tgt).extend(vs)
        elif n == "SETITEM":
            v = stack.pop()
            k = stack.pop()
            tgt = stack[-1]
            if isinstance(tgt, _Node):
                tgt.items.append((k, v))
            else:
                tgt[k] = v
        elif n == "SETITEMS":
            kv = pop_mark()
            tgt = stack[-1]
            for i in range(0, len(kv), 2):
                if isinstance(tgt, _Node):
                    tgt.items.append((kv[i], kv[i + 1]))
                else:
                    tgt[kv[i]] = kv[i + 1]
        elif n in ("REDUCE", "NEWOBJ"):
            args = stack.pop()
            tag = stack.pop()
            if not isinstance(tag, _Tag):
                raise PlanetoidFormatError("REDUCE on non-global")
            stack.append(_Node(tag, args))
        elif n == "BUILD":
            state = stack.pop()
            tgt = stack[-1]
            if not isinstance(tgt, _Node):
                raise PlanetoidFormatError("BUILD on plain data")
            tgt.state = state
        elif n == "STOP":
            break
        else:
            raise PlanetoidFormatError("opcode %s not allowed" % n)
    if len(stack) != 1:
        raise PlanetoidFormatError("malformed stream")
    return _finish(stack[0])


PARTS = ("x", "y", "tx", "ty", "allx", "ally", "graph")


def read_ind_files(data_dir, dataset, missing_ok=False):
    """Read the raw ``ind.<dataset>.*`` files (reference gcn/utils.py:44-55)
    with the non-executing decoder.  Returns a dict of the seven parts plus
    ``test_index`` (file order).  With ``missing_ok`` absent parts are skipped
    (the Pubmed ``allx`` blob is not always distributed)."""
    out = {}
    for part in PARTS:
        path = os.path.join(data_dir, "ind.%s.%s" % (dataset, part))
        if missing_ok and not os.path.exists(path):
            continue
        with open(path, "rb") as f:
            out[part] = decode(f.read())
    idx = []
    with open(os.path.join(data_dir, "ind.%s.test.index" % dataset)) as f:
        for line in f:
            idx.append(int(line.strip()))
    out["test_index"] = np.asarray(idx, dtype=np.int64)
    return out


def _graph_to_arrays(graph):
    keys = np.fromiter(graph.keys(), dtype=np.int64, count=len(graph))
    lens = np.fromiter((len(v) for v in graph.values()), dtype=np.int64, count=len(graph))
    indptr = np.zeros(len(graph) + 1, dtype=np.int64)
    np.cumsum(lens, out=indptr[1:])
    nbrs = np.empty(int(indptr[-1]), dtype=np.int32)
    p = 0
    for v in graph.values():
        nbrs[p:p + len(v)] = v
        p += len(v)
    return keys, indptr, nbrs


def save_npz(path, parts):
    """Write the decoded parts as one compressed ``.npz`` of plain arrays."""
    arrs = {}
    for name in ("x", "tx", "allx"):
        if name not in parts:
            continue
        m = parts[name].tocsr()
        arrs[name + "_data"] = m.data
        arrs[name + "_indices"] = m.indices.astype(np.int32)
        arrs[name + "_indptr"] = m.indptr.astype(np.int32)
        arrs[name + "_shape"] = np.asarray(m.shape, dtype=np.int64)
    for name in ("y", "ty", "ally"):
        if name in parts:
            arrs[name] = parts[name]
    keys, indptr, nbrs = _graph_to_arrays(parts["graph"])
    arrs["graph_keys"] = keys
    arrs["graph_indptr"] = indptr
    arrs["graph_nbrs"] = nbrs
    arrs["test_index"] = parts["test_index"]
    np.savez_compressed(path, **arrs)


def load_npz(path):
    """Inverse of :func:`save_npz` (``allow_pickle=False``)."""
    z = np.load(path, allow_pickle=False)
    out = {}
    for name in ("x", "tx", "allx"):
        if name + "_data" not in z.files:
            continue
        out[name] = sp.csr_matrix(
            (z[name + "_data"], z[name + "_indices"], z[name + "_indptr"]),
            shape=tuple(int(s) for s in z[name + "_shape"]))
    for name in ("y", "ty", "ally"):
        if name in z.files:
            out[name] = z[name]
    keys, indptr, nbrs = z["graph_keys"], z["graph_indptr"], z["graph_nbrs"]
    out["graph"] = {int(k): nbrs[indptr[i]:indptr[i + 1]].tolist()
                    for i, k in enumerate(keys)}
    out["test_index"] = z["test_index"]
    return out
