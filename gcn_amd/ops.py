"""Python entry points of the HIP kernels + their autograd glue.

torch is used for tensor ownership, streams and the autograd tape; every
product, reduction, loss and update below runs in ``libgcnhip.so`` through
ctypes.  What the drop-in layer path (``gcn_amd/layers.py``) leaves to torch's
own element-wise kernels, all of them off the measured path: the concatenation
of the per-support weights of a Chebyshev layer (``torch.cat``, once per step),
a layer's optional bias add, a relu that cannot be fused into the product that
feeds it (``torch.relu``), the ``x.repeat`` of the featureless multi-support
case and the ``part(k) - b2`` of the opt-in Clenshaw recurrence.  The large-graph
engine (``gcn_amd/engine.py``) uses none of them.
All functions raise on CPU tensors -- there is no fallback path.
"""
import ctypes as C

import torch

from . import _lib
from .sparse import Csr, WORKSPACE, _ptr, _stream


def _check_dense(t, name):
    if not isinstance(t, torch.Tensor) or t.device.type != "cuda":
        raise RuntimeError("%s must be a CUDA (ROCm) tensor; gcn_amd has no CPU path" % name)
    if t.dtype != torch.float32:
        raise TypeError("%s must be float32, got %s" % (name, t.dtype))
    if t.dim() != 2 or t.stride(1) != 1:
        raise ValueError("%s must be a row-major 2-D tensor" % name)


# ------------------------------------------------------------------------ SpMM
def spmm(a, b, out=None, alpha=1.0, beta=0.0, mask=None, mask_scale=1.0, relu=False, b2=None,
         dropout=None, sign_bits=None, dropout_rows=None):
    """``out = act(alpha * A' @ b + beta * out)`` with ``A`` a :class:`Csr`.
    ``mask`` (uint8, in the order of the tuple ``A`` was built from) applies
    sparse dropout to the values on the fly.  Replaces ``dot(x, y, sparse=True)``
    (reference gcn/layers.py:30-33) and ``sparse_dropout`` (:21-27).
    With ``b2`` the dense operand is the row-wise concatenation ``[b; b2]``
    without being assembled (local rows + halo rows of a partitioned run).
    ``dropout=(keep, seed, offset, ld, col0)`` applies dropout to the output in
    the epilogue (element (r, c) has Philox index offset + r*ld + col0 + c).
    ``sign_bits`` (int32 [n_rows, width / 32]) also receives the sign pattern of
    the stored output, the layout :func:`pack_sign_bits` writes
    (:func:`spmm_supports_sign_bits`).  ``dropout_rows`` (int32 [n_rows]): output
    row r draws the dropout mask of row ``dropout_rows[r]`` instead of r -- for a
    matrix that holds a subset of the rows of A_hat next to each other
    (gcnx_spmm_csr_rows_f32)."""
    lib = _lib.load()
    _check_dense(b, "b")
    rows_b2 = 0
    if b2 is not None:
        _check_dense(b2, "b2")
        if b2.shape[1] != b.shape[1] or b2.device != b.device:
            raise ValueError("b and b2 differ in width or device")
        rows_b2 = b2.shape[0]
    if b.shape[0] + rows_b2 != a.n_cols:
        raise ValueError("shape mismatch: A is %s, operand has %d rows" % (a.shape, b.shape[0] + rows_b2))
    if b.device != a.device:
        raise ValueError("A and b live on different devices")
    width = b.shape[1]
    if out is None:
        if beta != 0.0:
            raise ValueError("beta != 0 needs an existing out")
        out = torch.empty((a.n_rows, width), dtype=torch.float32, device=b.device)
    else:
        _check_dense(out, "out")
        if tuple(out.shape) != (a.n_rows, width):
            raise ValueError("out has shape %s, expected %s" % (tuple(out.shape), (a.n_rows, width)))
    midx = None
    if mask is not None:
        if mask.dtype != torch.uint8 or mask.numel() != a.nnz or mask.device != a.device:
            raise ValueError("mask must be uint8 with one byte per nonzero on A's device")
        midx = a.input_index
    drop = None
    if dropout is not None and dropout[0] < 1.0:
        keep, seed, offset, ld, col0 = dropout
        drop = C.byref(_lib.DropoutStruct(float(keep), int(seed) & (2 ** 64 - 1), int(offset), int(ld),
                                          int(col0)))
    if dropout_rows is not None and drop is None:
        dropout_rows = None
    if dropout_rows is not None:
        dr = dropout_rows
        if mask is not None or dr.dtype != torch.int32 or dr.dim() != 1 or dr.numel() != a.n_rows or \
                not dr.is_contiguous() or dr.device != a.device:
            raise ValueError("dropout_rows: contiguous int32 [n_rows] on A's device, no value mask")
    if sign_bits is not None:
        sb = sign_bits
        if mask is not None or sb.dtype != torch.int32 or sb.dim() != 2 or sb.shape[0] != a.n_rows or \
                sb.shape[1] * 32 < width or sb.stride(1) != 1 or sb.device != a.device or \
                not spmm_supports_sign_bits(width):
            raise ValueError("sign_bits: int32 [n_rows, width / 32] on A's device, width % 32 == 0, no mask")
        if dropout_rows is None:
            with torch.cuda.device(a.device):
                need = lib.gcnx_spmm_workspace_bytes(a.struct_ref(), width)
                ws = WORKSPACE.get(need, a.device)
                _lib.check(lib.gcnx_spmm_csr_signs_f32(
                    a.struct_ref(), _ptr(a.chunk_row), _ptr(b), b.stride(0),
                    _ptr(b2 if rows_b2 else None), b2.stride(0) if rows_b2 else 0, b.shape[0],
                    _ptr(out), out.stride(0), width, float(alpha), float(beta), 1 if relu else 0, drop,
                    _ptr(sb), sb.stride(0), _ptr(ws), ws.numel(), _stream()))
            return out
    if dropout_rows is not None:
        with torch.cuda.device(a.device):
            need = lib.gcnx_spmm_workspace_bytes(a.struct_ref(), width)
            ws = WORKSPACE.get(need, a.device)
            _lib.check(lib.gcnx_spmm_csr_rows_f32(
                a.struct_ref(), _ptr(a.chunk_row), _ptr(b), b.stride(0),
                _ptr(b2 if rows_b2 else None), b2.stride(0) if rows_b2 else 0, b.shape[0],
                _ptr(out), out.stride(0), width, float(alpha), float(beta), 1 if relu else 0, drop,
                _ptr(dropout_rows), _ptr(sign_bits), sign_bits.stride(0) if sign_bits is not None else 0,
                _ptr(ws), ws.numel(), _stream()))
        return out
    with torch.cuda.device(a.device):
        need = lib.gcnx_spmm_workspace_bytes(a.struct_ref(), width)
        ws = WORKSPACE.get(need, a.device)
        _lib.check(lib.gcnx_spmm_csr_split_f32(
            a.struct_ref(), _ptr(a.chunk_row), _ptr(b), b.stride(0),
            _ptr(b2 if rows_b2 else None), b2.stride(0) if rows_b2 else 0, b.shape[0],
            _ptr(out), out.stride(0), width, float(alpha), float(beta), _ptr(mask), _ptr(midx),
            float(mask_scale), 1 if relu else 0, drop, _ptr(ws), ws.numel(), _stream()))
    return out


def spmm_supports_sign_bits(width):
    """Whether ``spmm(..., sign_bits=...)`` is served for this operand width."""
    return bool(_lib.load().gcnx_spmm_signs_supported(int(width)))


def spmm_supports_factored(width, inner):
    """Whether ``spmm_factored`` is served for this operand width / inner size."""
    return bool(_lib.load().gcnx_spmm_factored_supported(int(width), int(inner)))


def spmm_factored(a, d, w, gate_bits, gate_scale=1.0, trans_w=False, out=None, d2=None, gate_bits2=None):
    """``out = A @ G`` with ``G = where(bit, gate_scale * (D @ op(W)), 0)`` formed
    on the fly (never stored).  ``d`` [n_local_cols, inner] and ``gate_bits``
    (int32 [n_local_cols, width/32]) cover the first columns of ``a``; ``d2`` /
    ``gate_bits2`` the remaining ones (halo rows).  ``w`` is [inner, width], or
    [width, inner] with ``trans_w``."""
    lib = _lib.load()
    _check_dense(d, "d")
    _check_dense(w, "w")
    inner = d.shape[1]
    width = w.shape[0] if trans_w else w.shape[1]
    if (w.shape[1] if trans_w else w.shape[0]) != inner:
        raise ValueError("inner dimensions differ: d %s, w %s" % (tuple(d.shape), tuple(w.shape)))
    if gate_bits.dtype != torch.int32 or gate_bits.shape[0] != d.shape[0] or gate_bits.shape[1] * 32 < width:
        raise ValueError("gate_bits must be int32 [rows of d, ceil(width / 32)]")
    split = d.shape[0]
    if d2 is not None and d2.shape[0] > 0:
        _check_dense(d2, "d2")
        if gate_bits2 is None or gate_bits2.dtype != torch.int32 or gate_bits2.shape[0] != d2.shape[0] or \
                gate_bits2.shape[1] * 32 < width or d2.shape[1] != inner:
            raise ValueError("d2 / gate_bits2 do not match d / gate_bits")
        n_cols = split + d2.shape[0]
    else:
        d2, gate_bits2, n_cols = None, None, split
    if n_cols != a.n_cols:
        raise ValueError("operand rows (%d) != matrix columns (%d)" % (n_cols, a.n_cols))
    if out is None:
        out = torch.empty((a.n_rows, width), dtype=torch.float32, device=d.device)
    elif tuple(out.shape) != (a.n_rows, width) or out.stride(1) != 1:
        raise ValueError("out has shape %s, expected %s" % (tuple(out.shape), (a.n_rows, width)))
    with torch.cuda.device(d.device):
        need = lib.gcnx_spmm_workspace_bytes(a.struct_ref(), width)
        ws = WORKSPACE.get(need, d.device)
        _lib.check(lib.gcnx_spmm_csr_factored_f32(
            a.struct_ref(), _ptr(a.chunk_row), _ptr(d), d.stride(0), _ptr(gate_bits), gate_bits.stride(0),
            _ptr(d2), d2.stride(0) if d2 is not None else 0, _ptr(gate_bits2),
            gate_bits2.stride(0) if gate_bits2 is not None else 0, split, _ptr(w), w.stride(0), int(trans_w),
            inner, float(gate_scale), _ptr(out), out.stride(0), width, _ptr(ws), ws.numel(), _stream()))
    return out


class _SpmmFn(torch.autograd.Function):
    """y = act(A' @ b).  Gradient w.r.t. b only: the sparse operands of the
    reference are placeholders, no gradient reaches their values
    (reference gcn/models.py:58)."""

    @staticmethod
    def forward(ctx, b, a, mask, mask_scale, relu):
        y = spmm(a, b, mask=mask, mask_scale=mask_scale, relu=relu)
        ctx.a = a
        ctx.mask = mask
        ctx.mask_scale = mask_scale
        ctx.relu = relu
        if relu:
            ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = dy.contiguous()
        if ctx.relu:
            (y,) = ctx.saved_tensors
            dy = relu_dropout_bwd(y, dy)
        db = spmm(ctx.a.transpose(), dy, mask=ctx.mask, mask_scale=ctx.mask_scale)
        return db, None, None, None, None


def spmm_autograd(a, b, mask=None, mask_scale=1.0, relu=False):
    return _SpmmFn.apply(b, a, mask, mask_scale, relu)


class _SpmmAddFn(torch.autograd.Function):
    """y = act(alpha * A @ b + c) in one launch (c is copied into the output and
    accumulated onto).  Gradients: d b = alpha * A^T dy', d c = dy', with dy' the
    relu-gated dy."""

    @staticmethod
    def forward(ctx, b, c, a, alpha, relu):
        y = c.clone(memory_format=torch.contiguous_format)
        spmm(a, b, out=y, alpha=alpha, beta=1.0, relu=relu)
        ctx.a = a
        ctx.alpha = alpha
        ctx.relu = relu
        if relu:
            ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = dy.contiguous()
        if ctx.relu:
            (y,) = ctx.saved_tensors
            dy = relu_dropout_bwd(y, dy)
        db = spmm(ctx.a.transpose(), dy, alpha=ctx.alpha) if ctx.needs_input_grad[0] else None
        return db, (dy if ctx.needs_input_grad[1] else None), None, None, None


def spmm_add_autograd(a, b, c, alpha=1.0, relu=False):
    return _SpmmAddFn.apply(b.contiguous(), c, a, float(alpha), relu)


# ------------------------------------------------------------------------ GEMM
def gemm(a, b, trans_a=False, trans_b=False, out=None, alpha=1.0, beta=0.0, relu=False, gate=None,
         gate_scale=1.0, a_dropout=None, a_dropout_rows=None, gate_bits=None, dropout=None, sign_bits=None):
    """``out = act(alpha * op(a) @ op(b) + beta * out)``, fp32.  Replaces
    ``dot(x, y, sparse=False)`` (reference gcn/layers.py:35).  With ``gate``
    (same shape as out) the epilogue is ``gate > 0 ? out * gate_scale : 0``: the
    backward of relu + dropout fused into the product that feeds it.
    ``a_dropout=(keep, seed, offset)`` applies dropout to ``a`` (as stored,
    row-major with its own row length) while it is read, so dropout(a) is never
    written (MFMA shapes only); with ``a_dropout_rows`` (int32, one per stored row
    of ``a``) row i draws the mask of row ``a_dropout_rows[i]`` (``a`` holds a
    gathered subset of the rows the mask is defined on; with a transposed ``a``
    only where :func:`gemm_supports_a_dropout_rows_t` says so).  ``gate_bits`` (uint32 /
    int32 [M, ceil(N/32)]) is the gate as one bit per element
    (:func:`pack_sign_bits`).  ``dropout=(keep, seed, offset, ld, col0)`` applies
    dropout to the output behind the activation, as in :func:`spmm`; with it,
    ``sign_bits`` (int32 [M, N/32]) receives the sign pattern of the stored
    output (fused epilogue only)."""
    lib = _lib.load()
    _check_dense(a, "a")
    _check_dense(b, "b")
    if sign_bits is not None and (dropout is None or dropout[0] >= 1.0):
        raise ValueError("sign_bits needs the fused output dropout (dropout with keep < 1)")
    M, K = (a.shape[1], a.shape[0]) if trans_a else (a.shape[0], a.shape[1])
    Kb, N = (b.shape[1], b.shape[0]) if trans_b else (b.shape[0], b.shape[1])
    if K != Kb:
        raise ValueError("inner dimensions differ: %d vs %d" % (K, Kb))
    if out is None:
        if beta != 0.0:
            raise ValueError("beta != 0 needs an existing out")
        out = torch.empty((M, N), dtype=torch.float32, device=a.device)
    else:
        _check_dense(out, "out")
        if tuple(out.shape) != (M, N):
            raise ValueError("out has shape %s, expected %s" % (tuple(out.shape), (M, N)))
    with torch.cuda.device(a.device):
        need = lib.gcnx_gemm_workspace_bytes(int(trans_a), int(trans_b), M, N, K)
        ws = WORKSPACE.get(need, a.device)
        if dropout is not None and dropout[0] < 1.0:
            if gate is not None or gate_bits is not None or (a_dropout is not None and a_dropout[0] < 1.0) \
                    or beta != 0.0:
                raise ValueError("output dropout excludes gate, gate_bits, a_dropout and beta")
            keep, seed, offset, ld, col0 = dropout
            fused = bool(lib.gcnx_gemm_relu_dropout_supported(int(trans_a), int(trans_b), M, N, K)) and \
                out.stride(0) % 4 == 0 and out.data_ptr() % 16 == 0 and a.stride(0) % 4 == 0 and \
                a.data_ptr() % 16 == 0 and (trans_b or N <= 32 or (b.stride(0) % 4 == 0 and b.data_ptr() % 16 == 0)) \
                and int(offset) % 4 == 0 and int(ld) % 4 == 0 and int(col0) % 4 == 0
            if sign_bits is not None:
                # sign words of the stored output straight from the fused epilogue
                if not fused or N % 32 or sign_bits.dtype != torch.int32 or tuple(sign_bits.shape) != (M, N // 32) \
                        or sign_bits.stride(1) != 1:
                    raise ValueError("sign_bits: int32 [M, N / 32], fused output dropout only (see "
                                     "gemm_supports_output_dropout), N % 32 == 0")
                d = _lib.DropoutStruct(float(keep), int(seed) & (2 ** 64 - 1), int(offset), int(ld), int(col0))
                _lib.check(lib.gcnx_gemm_relu_dropout_signs_f32(
                    int(trans_a), int(trans_b), M, N, K, _ptr(a), a.stride(0), _ptr(b), b.stride(0), _ptr(out),
                    out.stride(0), float(alpha), float(beta), 1 if relu else 0, C.byref(d), _ptr(sign_bits),
                    sign_bits.stride(0), _ptr(ws), ws.numel(), _stream()))
            elif fused:
                d = _lib.DropoutStruct(float(keep), int(seed) & (2 ** 64 - 1), int(offset), int(ld), int(col0))
                _lib.check(lib.gcnx_gemm_relu_dropout_f32(int(trans_a), int(trans_b), M, N, K, _ptr(a), a.stride(0),
                                                          _ptr(b), b.stride(0), _ptr(out), out.stride(0),
                                                          float(alpha), float(beta), 1 if relu else 0, C.byref(d),
                                                          _ptr(ws), ws.numel(), _stream()))
            else:
                # shapes outside the tall MFMA kernel: the product, then the same mask in place
                if int(ld) != N or int(col0) != 0 or not out.is_contiguous():
                    raise ValueError("unfused output dropout needs ld == N, col0 == 0 and a contiguous out")
                _lib.check(lib.gcnx_gemm_f32(int(trans_a), int(trans_b), M, N, K, _ptr(a), a.stride(0),
                                             _ptr(b), b.stride(0), _ptr(out), out.stride(0), float(alpha),
                                             0.0, 1 if relu else 0, _ptr(ws), ws.numel(), _stream()))
                dropout_philox(out.view(-1), keep, seed, offset, out=out.view(-1))
        elif a_dropout is not None and a_dropout[0] < 1.0:
            if gate is not None or relu or not a.is_contiguous():
                raise ValueError("a_dropout excludes gate / relu and needs a contiguous a")
            if gate_bits is not None:
                raise ValueError("a_dropout excludes gate_bits")
            keep, seed, offset = a_dropout
            d = _lib.DropoutStruct(float(keep), int(seed) & (2 ** 64 - 1), int(offset), a.shape[1], 0)
            if a_dropout_rows is not None:
                r = a_dropout_rows
                if r.dtype != torch.int32 or r.numel() != a.shape[0] or not r.is_contiguous() or r.device != a.device:
                    raise ValueError("a_dropout_rows: one contiguous int32 per stored row of a")
                if trans_a:
                    if trans_b:
                        raise ValueError("a_dropout_rows with a transposed a needs a plain b")
                    _lib.check(lib.gcnx_gemm_dropout_at_rows_f32(M, N, K, _ptr(a), a.stride(0), C.byref(d), _ptr(r),
                                                                 _ptr(b), b.stride(0), _ptr(out), out.stride(0),
                                                                 float(alpha), float(beta), _ptr(ws), ws.numel(),
                                                                 _stream()))
                else:
                    _lib.check(lib.gcnx_gemm_dropout_a_rows_f32(int(trans_b), M, N, K, _ptr(a), a.stride(0),
                                                                C.byref(d), _ptr(r), _ptr(b), b.stride(0),
                                                                _ptr(out), out.stride(0), float(alpha),
                                                                float(beta), _ptr(ws), ws.numel(), _stream()))
            else:
                _lib.check(lib.gcnx_gemm_dropout_a_f32(int(trans_a), int(trans_b), M, N, K, _ptr(a),
                                                       a.stride(0), C.byref(d), _ptr(b), b.stride(0),
                                                       _ptr(out), out.stride(0), float(alpha), float(beta),
                                                       _ptr(ws), ws.numel(), _stream()))
        elif gate_bits is not None:
            gb = gate_bits
            if gate is not None or relu:
                raise ValueError("gate_bits excludes gate / relu")
            if gb.dtype not in (torch.int32, torch.uint32) or gb.dim() != 2 or gb.shape[0] != M or \
                    gb.shape[1] * 32 < N or gb.stride(1) != 1 or gb.device != a.device:
                raise ValueError("gate_bits must be a 32-bit integer tensor [M, ceil(N/32)]")
            _lib.check(lib.gcnx_gemm_bitgated_f32(int(trans_a), int(trans_b), M, N, K, _ptr(a),
                                                  a.stride(0), _ptr(b), b.stride(0), _ptr(out),
                                                  out.stride(0), float(alpha), float(beta), _ptr(gb),
                                                  gb.stride(0), float(gate_scale), _ptr(ws),
                                                  ws.numel(), _stream()))
        elif gate is not None:
            _check_dense(gate, "gate")
            if tuple(gate.shape) != (M, N) or relu:
                raise ValueError("gate must have the shape of out and excludes relu")
            _lib.check(lib.gcnx_gemm_gated_f32(int(trans_a), int(trans_b), M, N, K, _ptr(a),
                                               a.stride(0), _ptr(b), b.stride(0), _ptr(out),
                                               out.stride(0), float(alpha), float(beta), _ptr(gate),
                                               gate.stride(0), float(gate_scale), _ptr(ws),
                                               ws.numel(), _stream()))
        else:
            _lib.check(lib.gcnx_gemm_f32(int(trans_a), int(trans_b), M, N, K, _ptr(a), a.stride(0),
                                         _ptr(b), b.stride(0), _ptr(out), out.stride(0), float(alpha),
                                         float(beta), 1 if relu else 0, _ptr(ws), ws.numel(), _stream()))
    return out


def dense_arith(mode=None):
    """Arithmetic of the large dense products on this thread: 0 = exactly split
    operands on the bf16 matrix pipe (fp32 accuracy, default), 1 = fp32-input MFMA
    chains only.  Returns the previous mode; ``None`` only reads."""
    return int(_lib.load().gcnx_dense_arith(-1 if mode is None else int(mode)))


def dropout_rows_philox(x, row_ids, keep, seed, offset, out=None):
    """Dropout of gathered rows: row i of ``x`` draws the mask of row
    ``row_ids[i]`` of the [*, W] matrix the mask is defined on (element index
    ``offset + row_ids[i] * W + c``)."""
    lib = _lib.load()
    _check_dense(x, "x")
    if row_ids.dtype != torch.int32 or row_ids.numel() != x.shape[0] or not row_ids.is_contiguous() or \
            row_ids.device != x.device:
        raise ValueError("row_ids: one contiguous int32 per row of x")
    if out is None:
        out = torch.empty_like(x)
    else:
        _check_dense(out, "out")
        if out.shape != x.shape:
            raise ValueError("out must have the shape of x")
    with torch.cuda.device(x.device):
        _lib.check(lib.gcnx_dropout_rows_philox_f32(x.shape[0], x.shape[1], _ptr(x), x.stride(0), _ptr(row_ids),
                                                    float(keep), int(seed) & (2 ** 64 - 1), int(offset),
                                                    _ptr(out), out.stride(0), _stream()))
    return out


def pack_sign_bits(x, rows=None, out=None):
    """int32 [n, ceil(W/32)]: bit (c % 32) of word (i, c // 32) is set iff
    ``x[rows[i] if rows is not None else i, c] > 0``."""
    lib = _lib.load()
    _check_dense(x, "x")
    n = int(rows.numel()) if rows is not None else x.shape[0]
    width = x.shape[1]
    words = (width + 31) // 32
    if rows is not None and (rows.dtype != torch.int32 or not rows.is_contiguous() or rows.device != x.device):
        raise ValueError("rows must be a contiguous int32 tensor on the device of x")
    if out is None:
        out = torch.empty((n, words), dtype=torch.int32, device=x.device)
    elif out.dtype != torch.int32 or tuple(out.shape) != (n, words) or out.stride(1) != 1:
        raise ValueError("out must be int32 [n, ceil(W/32)]")
    with torch.cuda.device(x.device):
        _lib.check(lib.gcnx_pack_sign_bits_f32(n, _ptr(rows) if rows is not None else None, _ptr(x),
                                               x.stride(0), width, _ptr(out), out.stride(0), _stream()))
    return out


def gemm_supports_output_dropout(trans_a, M, N, K):
    """Whether ``gemm(..., dropout=...)`` is fused into the product's epilogue
    for this shape (otherwise the mask is applied by a second kernel)."""
    return bool(_lib.load().gcnx_gemm_relu_dropout_supported(int(trans_a), 0, int(M), int(N), int(K)))


def gemm_supports_a_dropout(trans_a, M, N, K):
    """Whether ``gemm(..., a_dropout=...)`` is served for this shape."""
    return bool(_lib.load().gcnx_gemm_dropout_a_supported(int(trans_a), 0, int(M), int(N), int(K)))


class _GemmFn(torch.autograd.Function):
    """y = act(a @ b) with both gradients.  The arithmetic mode of the large
    products (:func:`dense_arith`) is a per-thread setting of the library and
    torch runs backward on a thread of its own, so the mode the forward product
    ran under is kept with the node and put in force around the backward
    products: forward and backward of one product use the same arithmetic."""

    @staticmethod
    def forward(ctx, a, b, relu):
        ctx.arith = dense_arith()
        y = gemm(a, b, relu=relu)
        ctx.relu = relu
        if relu:
            ctx.save_for_backward(a, b, y)
        else:
            ctx.save_for_backward(a, b)
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = dy.contiguous()
        if ctx.relu:
            a, b, y = ctx.saved_tensors
            dy = relu_dropout_bwd(y, dy)
        else:
            a, b = ctx.saved_tensors
        before = dense_arith(ctx.arith)
        try:
            da = gemm(dy, b, trans_b=True) if ctx.needs_input_grad[0] else None
            db = gemm(a, dy, trans_a=True) if ctx.needs_input_grad[1] else None
        finally:
            dense_arith(before)
        return da, db, None


def gemm_supports_a_dropout_rows_t(M, N, K):
    """Whether ``gemm(a, b, trans_a=True, a_dropout=..., a_dropout_rows=...)`` is
    served for an [M, N] output summed over K stored rows."""
    return bool(_lib.load().gcnx_gemm_dropout_at_rows_supported(int(M), int(N), int(K)))


def gemm_autograd(a, b, relu=False):
    return _GemmFn.apply(a, b, relu)


# ------------------------------------------------------------- loss / accuracy
def masked_softmax_xent(logits, labels, mask, want_grad=True, mask_total=0.0, grad_out=None):
    """Returns ``(loss_acc, grad)``: ``loss_acc`` is a 2-element device tensor
    ``[loss, accuracy]`` (reference gcn/metrics.py:4-20), ``grad`` = dloss/dlogits
    or None."""
    lib = _lib.load()
    _check_dense(logits, "logits")
    _check_dense(labels, "labels")
    n, c = logits.shape
    if tuple(labels.shape) != (n, c):
        raise ValueError("labels shape %s != logits shape %s" % (tuple(labels.shape), (n, c)))
    if mask.dtype != torch.uint8 or mask.numel() != n or mask.device != logits.device:
        raise ValueError("mask must be uint8 [n] on the logits' device")
    out = torch.empty(2, dtype=torch.float32, device=logits.device)
    grad = (grad_out if grad_out is not None else torch.empty_like(logits)) if want_grad else None
    with torch.cuda.device(logits.device):
        need = lib.gcnx_xent_workspace_bytes(n)
        ws = WORKSPACE.get(need, logits.device)
        _lib.check(lib.gcnx_masked_softmax_xent_f32(
            n, c, _ptr(logits), logits.stride(0), _ptr(labels), labels.stride(0), _ptr(mask),
            float(mask_total), _ptr(out), _ptr(grad), grad.stride(0) if grad is not None else 0,
            _ptr(ws), ws.numel(), _stream()))
    return out, grad


class _XentFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, labels, mask):
        # a mask uploaded by the session carries its count of set entries, which
        # saves the counting pass (two launches) on every step
        out, grad = masked_softmax_xent(logits.contiguous(), labels, mask, want_grad=True,
                                        mask_total=float(getattr(mask, "mask_total", 0.0)))
        ctx.save_for_backward(grad)
        loss, acc = out[0], out[1]
        ctx.mark_non_differentiable(acc)
        return loss, acc

    @staticmethod
    def backward(ctx, dloss, _dacc):
        (grad,) = ctx.saved_tensors
        return grad * dloss, None, None


def masked_softmax_xent_autograd(logits, labels, mask):
    """``(loss, accuracy)`` as 0-dim tensors; ``loss`` is differentiable."""
    return _XentFn.apply(logits, labels, mask)


# ------------------------------------------------------------------ elementwise
def device_counter(device, value=0):
    """A 64-bit step counter in device memory (int64 tensor of one element),
    advanced with :func:`counter_add` on the stream."""
    dev = torch.device(device)
    if dev.type != "cuda":
        raise RuntimeError("device_counter needs a CUDA (ROCm) device")
    return torch.full((1,), int(value), dtype=torch.int64, device=dev)


def counter_add(counter, inc=1):
    lib = _lib.load()
    if counter.device.type != "cuda" or counter.dtype != torch.int64 or counter.numel() != 1:
        raise ValueError("counter must be a one-element int64 CUDA tensor")
    with torch.cuda.device(counter.device):
        _lib.check(lib.gcnx_counter_add_u64(_ptr(counter), int(inc), _stream()))
    return counter


def bernoulli_mask(n, keep, seed, offset, device, out=None, step=None, step_stride=0):
    """uint8 mask, byte k = 1 with probability ``keep``, a pure function of
    (seed, offset + k) (Philox4x32-10).  With ``step`` (a device counter) the
    offset is advanced by ``step * step_stride`` when the kernel runs, so a launch
    recorded in a HIP graph draws a new mask on every replay."""
    lib = _lib.load()
    dev = torch.device(device)
    if dev.type != "cuda":
        raise RuntimeError("bernoulli_mask needs a CUDA (ROCm) device")
    if out is not None:
        if out.dtype != torch.uint8 or out.numel() != int(n) or not out.is_contiguous():
            raise ValueError("out must be a contiguous uint8 tensor of n elements")
        mask = out
    else:
        mask = torch.empty(int(n), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        if step is not None:
            if step.device.type != "cuda" or step.dtype != torch.int64 or step.numel() != 1:
                raise ValueError("step must be a one-element int64 CUDA tensor")
            _lib.check(lib.gcnx_bernoulli_mask_dev_u8(int(n), float(keep), int(seed) & (2 ** 64 - 1),
                                                      int(offset), _ptr(step), int(step_stride),
                                                      _ptr(mask), _stream()))
        else:
            _lib.check(lib.gcnx_bernoulli_mask_u8(int(n), float(keep), int(seed) & (2 ** 64 - 1),
                                                  int(offset), _ptr(mask), _stream()))
    return mask


def dropout_dense(x, mask, scale, out=None):
    lib = _lib.load()
    if x.device.type != "cuda" or x.dtype != torch.float32 or not x.is_contiguous():
        raise ValueError("x must be a contiguous float32 CUDA tensor")
    if mask.dtype != torch.uint8 or mask.numel() != x.numel():
        raise ValueError("mask must be uint8 with one byte per element")
    if out is None:
        out = torch.empty_like(x)
    with torch.cuda.device(x.device):
        _lib.check(lib.gcnx_dropout_dense_f32(x.numel(), _ptr(x), _ptr(mask), float(scale),
                                              _ptr(out), _stream()))
    return out


def relu_dropout_bwd(y, dy, mask=None, scale=1.0, out=None):
    """``dx = (y > 0 ? dy : 0)`` optionally times the dropout mask and scale."""
    lib = _lib.load()
    if y.device.type != "cuda" or not y.is_contiguous() or not dy.is_contiguous():
        raise ValueError("y and dy must be contiguous CUDA tensors")
    if y.numel() != dy.numel() or (mask is not None and mask.numel() != y.numel()):
        raise ValueError("y, dy and mask differ in size")
    dx = out if out is not None else torch.empty_like(dy)
    if dx.numel() != dy.numel() or not dx.is_contiguous():
        raise ValueError("out must be contiguous and as large as dy")
    with torch.cuda.device(y.device):
        _lib.check(lib.gcnx_relu_dropout_bwd_f32(y.numel(), _ptr(y), _ptr(dy), _ptr(mask),
                                                 float(scale), _ptr(dx), _stream()))
    return dx


def gather_rows(src, idx, out):
    """``out[j, :] = src[idx[j], :]`` (``idx`` int32 on the device)."""
    lib = _lib.load()
    _check_dense(src, "src")
    _check_dense(out, "out")
    if idx.dtype != torch.int32 or idx.device != src.device or not idx.is_contiguous():
        raise ValueError("idx must be a contiguous int32 tensor on src's device")
    if out.shape[0] != idx.numel() or out.shape[1] != src.shape[1]:
        raise ValueError("out must be [len(idx), width]")
    with torch.cuda.device(src.device):
        _lib.check(lib.gcnx_gather_rows_f32(idx.numel(), _ptr(idx), _ptr(src), src.stride(0),
                                            src.shape[1], _ptr(out), out.stride(0), _stream()))
    return out


def scatter_rows(src, idx, out):
    """``out[idx[j], :] = src[j, :]`` for distinct ``idx`` (int32 on the device);
    the other rows of ``out`` keep their contents."""
    lib = _lib.load()
    _check_dense(src, "src")
    _check_dense(out, "out")
    if idx.dtype != torch.int32 or idx.device != src.device or not idx.is_contiguous():
        raise ValueError("idx must be a contiguous int32 tensor on src's device")
    if src.shape[0] != idx.numel() or out.shape[1] != src.shape[1]:
        raise ValueError("src must be [len(idx), width of out]")
    with torch.cuda.device(src.device):
        _lib.check(lib.gcnx_scatter_rows_f32(idx.numel(), _ptr(idx), _ptr(src), src.stride(0),
                                             src.shape[1], _ptr(out), out.stride(0), _stream()))
    return out


def csr_filter_columns(a, col_keep):
    """The sub-matrix of ``a`` (a Csr) with the nonzeros of the flagged columns only
    (Csr.filter_columns): for products whose operand is exactly zero in the other rows."""
    return a.filter_columns(col_keep)


def csr_compact_rows(a):
    """``(sub, rows)``: ``a`` without its empty rows and the int32 ids of the rows kept
    (Csr.compact_rows)."""
    return a.compact_rows()


def csr_keep_rows(a, rows):
    """``a`` restricted to the rows ``rows`` (ascending ids; every row left out must be
    empty), stored next to each other (Csr.keep_rows)."""
    return a.keep_rows(rows)


def csr_transpose(a):
    """A fresh exact transpose of ``a`` (Csr.transpose without the cache): set-up of
    derived matrices."""
    return a.transpose(cache=False)


def philox_uniform(shape, lo, hi, seed, offset, device):
    """fp32 tensor of U(lo, hi) values, element i a pure function of
    (seed, offset + i) (synthetic benchmark inputs)."""
    lib = _lib.load()
    dev = torch.device(device)
    if dev.type != "cuda":
        raise RuntimeError("philox_uniform needs a CUDA (ROCm) device")
    out = torch.empty(shape, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(lib.gcnx_philox_uniform_f32(out.numel(), float(lo), float(hi),
                                               int(seed) & (2 ** 64 - 1), int(offset), _ptr(out),
                                               _stream()))
    return out


def dropout_philox(x, keep, seed, offset, out=None, mask_out=None):
    """``out = kept ? x / keep : 0`` with the Bernoulli(keep) mask regenerated
    from Philox (seed, offset + i): same bits as :func:`bernoulli_mask`."""
    lib = _lib.load()
    if x.device.type != "cuda" or x.dtype != torch.float32 or not x.is_contiguous():
        raise ValueError("x must be a contiguous float32 CUDA tensor")
    if out is None:
        out = torch.empty_like(x)
    if out.numel() != x.numel() or not out.is_contiguous():
        raise ValueError("out must be contiguous and as large as x")
    with torch.cuda.device(x.device):
        _lib.check(lib.gcnx_dropout_philox_f32(x.numel(), _ptr(x), float(keep),
                                               int(seed) & (2 ** 64 - 1), int(offset), _ptr(out),
                                               _ptr(mask_out), _stream()))
    return out


def relu_dropout_bwd_philox(y, dy, keep, seed, offset, out=None):
    """``dx = (y > 0 and kept) ? dy / keep : 0`` with the mask of
    :func:`dropout_philox` regenerated."""
    lib = _lib.load()
    if y.device.type != "cuda" or not y.is_contiguous() or not dy.is_contiguous() or y.numel() != dy.numel():
        raise ValueError("y and dy must be contiguous CUDA tensors of one size")
    dx = out if out is not None else torch.empty_like(dy)
    with torch.cuda.device(y.device):
        _lib.check(lib.gcnx_relu_dropout_bwd_philox_f32(y.numel(), _ptr(y), _ptr(dy), float(keep),
                                                        int(seed) & (2 ** 64 - 1), int(offset),
                                                        _ptr(dx), _stream()))
    return dx


class _DropoutFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, mask, scale):
        ctx.mask = mask
        ctx.scale = scale
        return dropout_dense(x.contiguous(), mask, scale)

    @staticmethod
    def backward(ctx, dy):
        return dropout_dense(dy.contiguous(), ctx.mask, ctx.scale), None, None


def dropout_dense_autograd(x, mask, scale):
    return _DropoutFn.apply(x, mask, scale)


def adam_step(w, grad, m, v, lr, t, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0):
    """In-place TF-style Adam (reference gcn/models.py:142); ``t`` counts from 1
    and is either a Python int or a device counter (read when the kernel runs)."""
    lib = _lib.load()
    for name, x in (("w", w), ("grad", grad), ("m", m), ("v", v)):
        if x.device.type != "cuda" or x.dtype != torch.float32 or not x.is_contiguous():
            raise ValueError("%s must be a contiguous float32 CUDA tensor" % name)
    with torch.cuda.device(w.device):
        if isinstance(t, torch.Tensor):
            if t.device.type != "cuda" or t.dtype != torch.int64 or t.numel() != 1:
                raise ValueError("t must be an int or a one-element int64 CUDA tensor")
            _lib.check(lib.gcnx_adam_step_dev_f32(w.numel(), _ptr(w), _ptr(grad), _ptr(m), _ptr(v),
                                                  float(lr), float(beta1), float(beta2), float(eps),
                                                  float(weight_decay), _ptr(t), _stream()))
        else:
            _lib.check(lib.gcnx_adam_step_f32(w.numel(), _ptr(w), _ptr(grad), _ptr(m), _ptr(v),
                                              float(lr), float(beta1), float(beta2), float(eps),
                                              float(weight_decay), int(t), _stream()))
    return w


def l2_loss(w, scale=1.0, add=None):
    """One-element tensor ``(add or 0) + scale * sum(w ** 2) / 2`` (reference
    gcn/models.py:148-149: ``weight_decay * tf.nn.l2_loss(var)`` added into the
    loss).  No gradient is recorded: the decay gradient ``scale * w`` is applied
    inside :func:`adam_step` (``weight_decay``)."""
    lib = _lib.load()
    w = w.detach()
    if w.device.type != "cuda" or w.dtype != torch.float32 or not w.is_contiguous():
        raise ValueError("w must be a contiguous float32 CUDA tensor")
    if add is not None:
        add = add.detach()
        if add.device != w.device or add.dtype != torch.float32 or add.numel() != 1:
            raise ValueError("add must be a one-element float32 tensor on w's device")
    out = torch.empty(1, dtype=torch.float32, device=w.device)
    with torch.cuda.device(w.device):
        _lib.check(lib.gcnx_l2_loss_f32(w.numel(), _ptr(w), float(scale), _ptr(add), _ptr(out), _stream()))
    return out


def profile_spmm_events(start, stop):
    """Development / benchmark hook: the next SpMM calls of this thread record
    ``start`` right before and ``stop`` right after their main kernel on the
    launch stream (pass ``None, None`` to switch it off).  The arguments are
    ``torch.cuda.Event(enable_timing=True)`` objects that have been recorded at
    least once (so that their HIP handles exist)."""
    lib = _lib.load()
    a = C.c_void_p(start.cuda_event) if start is not None else C.c_void_p(0)
    b = C.c_void_p(stop.cuda_event) if stop is not None else C.c_void_p(0)
    lib.gcnx_profile_spmm_events(a, b)


__all__ = ["Csr", "spmm", "spmm_factored", "spmm_supports_factored", "spmm_autograd", "gemm", "gemm_supports_a_dropout", "gemm_autograd", "masked_softmax_xent",
           "masked_softmax_xent_autograd", "bernoulli_mask", "dropout_dense",
           "dropout_dense_autograd", "relu_dropout_bwd", "dropout_philox", "relu_dropout_bwd_philox", "philox_uniform", "gather_rows", "adam_step"]
