"""Device-resident CSR matrices built through the C ABI.

The reference hands TensorFlow an unsorted COO tuple on every ``sess.run``
(gcn/utils.py:143-144); here the tuple is uploaded once, ordered on the device
into CSR (``gcnx_csr_from_coo``) and kept resident together with its chunk
table and, on demand, its exact transpose.  torch only owns the memory.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _require_cuda(device):
    dev = torch.device(device)
    if dev.type != "cuda":
        raise RuntimeError("gcn_amd runs on the GPU only (got device %s); "
                           "there is no CPU path" % (dev,))
    return dev


class _Workspace(object):
    """Per-device scratch buffer, grown on demand and then reused, so the
    compute calls never allocate once warm."""

    def __init__(self):
        self._buf = {}

    def get(self, nbytes, device):
        key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
        buf = self._buf.get(key)
        if buf is None or buf.numel() < nbytes:
            buf = None
            self._buf[key] = None
            n = max(int(nbytes), 256)
            buf = torch.empty(n, dtype=torch.uint8, device=device)
            self._buf[key] = buf
        return buf

    def buffers(self):
        """The buffers currently handed out; a recorded HIP graph keeps these
        referenced so that growing the workspace later cannot free memory the
        graph still writes."""
        return [b for b in self._buf.values() if b is not None]


WORKSPACE = _Workspace()


class Csr(object):
    """CSR on the device: ``row_ptr`` int64 [n_rows+1], ``col_idx`` int32 [nnz],
    ``val`` fp32 [nnz]; rows ordered by column, duplicates in input order.
    ``input_index[slot]`` is the position of the entry in the tuple the matrix
    was built from (dropout masks are drawn in that order,
    reference gcn/layers.py:21-27).  ``spmm_chunk`` is the per-matrix dispatch
    hint of the SpMM entry points (0: the library chooses; 64..512: chunk
    kernels with that many nonzeros per wavefront); new matrices take the class
    attribute ``default_spmm_chunk``, which the kernel tests override to run
    every case through each code path."""

    default_spmm_chunk = 0

    def __init__(self, n_rows, n_cols, row_ptr, col_idx, val, input_index=None):
        self.n_rows = int(n_rows)
        self.n_cols = int(n_cols)
        self.nnz = int(col_idx.numel())
        self.row_ptr = row_ptr
        self.col_idx = col_idx
        self.val = val
        self.input_index = input_index
        self.device = row_ptr.device
        self._t = None
        self.gather_keep = None
        self._struct = _lib.CsrStruct(self.n_rows, self.n_cols, self.nnz,
                                      row_ptr.data_ptr(), col_idx.data_ptr(), val.data_ptr(),
                                      int(self.default_spmm_chunk), None)
        lib = _lib.load()
        n_chunks = lib.gcnx_spmm_num_chunks(self.nnz)
        self.chunk_row = torch.empty(n_chunks + 1, dtype=torch.int32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(lib.gcnx_spmm_chunk_table(C.byref(self._struct), _ptr(self.chunk_row), _stream()))

    @property
    def shape(self):
        return (self.n_rows, self.n_cols)

    def struct_ref(self):
        return C.byref(self._struct)

    @property
    def spmm_chunk(self):
        return int(self._struct.spmm_chunk)

    @spmm_chunk.setter
    def spmm_chunk(self, value):
        self._struct.spmm_chunk = int(value)

    # -------------------------------------------------------------- cache hint
    def column_degrees(self):
        """int32 [n_cols]: number of nonzeros per column (gcnx_csr_column_degrees)."""
        deg = torch.empty(self.n_cols, dtype=torch.int32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(_lib.load().gcnx_csr_column_degrees(self.struct_ref(), _ptr(deg), _stream()))
        return deg

    def set_gather_hint(self, row_bytes, budget_bytes=176 << 20, min_degree=None):
        """Attach the cache hint of the wide SpMM (``gcnx_csr::gather_keep``): the
        operand rows of the most referenced columns -- as many as fit
        ``budget_bytes`` at ``row_bytes`` per row -- stay cacheable, every other
        row is gathered with streaming loads.  Results do not change.  The default
        budget is what measured best on the 256 MiB Infinity Cache of an MI355X
        (RMAT-24, 1-KiB rows: 40.5 -> 39.1 ms with the 166 k most referenced
        columns kept).  Returns the degree threshold used."""
        lib = _lib.load()
        deg = self.column_degrees()
        if min_degree is None:
            found = C.c_int32(0)
            with torch.cuda.device(self.device):
                ws = WORKSPACE.get(lib.gcnx_csr_degree_threshold_workspace_bytes(), self.device)
                _lib.check(lib.gcnx_csr_degree_threshold(_ptr(deg), self.n_cols, int(budget_bytes) // int(row_bytes),
                                                         C.byref(found), _ptr(ws), ws.numel(), _stream()))
            min_degree = found.value
        min_degree = max(int(min_degree), 2)
        keep = torch.empty((self.nnz + 63) // 64, dtype=torch.int64, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(lib.gcnx_csr_gather_keep(self.struct_ref(), _ptr(deg), min_degree, _ptr(keep), _stream()))
        self.gather_keep = keep
        self._struct.gather_keep = keep.data_ptr() if keep.numel() else None
        return min_degree

    def clear_gather_hint(self):
        self.gather_keep = None
        self._struct.gather_keep = None

    # -------------------------------------------------------------- sub-matrix
    def filter_columns(self, col_keep):
        """The sub-matrix of the nonzeros whose column is flagged (``col_keep``: a
        uint8 / bool tensor with one entry per column), same shape, kept nonzeros in
        their order (gcnx_csr_filter_columns_count / _fill).  ``A.filter_columns(f) @ B``
        equals ``A @ B`` when the rows of ``B`` outside ``f`` are exactly zero."""
        lib = _lib.load()
        flag = col_keep
        if flag.dtype == torch.bool:
            flag = flag.to(torch.uint8)
        if flag.dtype != torch.uint8 or flag.dim() != 1 or flag.numel() != self.n_cols or flag.device != self.device:
            raise ValueError("col_keep must be a uint8 or bool vector with one entry per column, on the matrix's device")
        flag = flag.contiguous()
        dev = self.device
        row_ptr = torch.empty(self.n_rows + 1, dtype=torch.int64, device=dev)
        kept = C.c_int64(0)
        with torch.cuda.device(dev):
            ws = torch.empty(max(lib.gcnx_csr_filter_workspace_bytes(self.nnz), 256), dtype=torch.uint8, device=dev)
            _lib.check(lib.gcnx_csr_filter_columns_count(self.struct_ref(), _ptr(flag), _ptr(row_ptr), C.byref(kept),
                                                         _ptr(ws), ws.numel(), _stream()))
            n = int(kept.value)
            col = torch.empty(n, dtype=torch.int32, device=dev)
            val = torch.empty(n, dtype=torch.float32, device=dev)
            src = torch.empty(n, dtype=torch.int64, device=dev)
            _lib.check(lib.gcnx_csr_filter_columns_fill(self.struct_ref(), _ptr(ws), ws.numel(), n, _ptr(col),
                                                        _ptr(val), _ptr(src), _stream()))
        idx = self.input_index[src] if self.input_index is not None else None
        del ws
        return Csr(self.n_rows, self.n_cols, row_ptr, col, val, idx)

    def compact_rows(self):
        """``(sub, rows)``: the matrix without its empty rows (``sub`` shares the
        column and value arrays; row i of ``sub`` is row ``rows[i]`` of this matrix,
        ``rows`` int32 ascending).  A product with ``sub`` writes the rows that can be
        nonzero next to each other; gcnx_scatter_rows_f32 puts them in place."""
        deg = self.row_ptr[1:] - self.row_ptr[:-1]
        rows = torch.nonzero(deg > 0).view(-1)
        row_ptr = torch.cat([self.row_ptr[rows], self.row_ptr[-1:]]).contiguous()
        sub = Csr(int(rows.numel()), self.n_cols, row_ptr, self.col_idx, self.val, self.input_index)
        return sub, rows.to(torch.int32).contiguous()

    def keep_rows(self, rows):
        """The matrix of the rows ``rows`` (ascending ids, int32 / int64 tensor) stored
        next to each other, for a selection that leaves out EMPTY rows only -- checked;
        unlike :meth:`compact_rows` the selection may keep empty rows (they stay, as
        zero rows of every product).  Shares the column and value arrays."""
        r = rows.to(device=self.device, dtype=torch.int64)
        if r.numel() and (int(r.min()) < 0 or int(r.max()) >= self.n_rows or
                          (r.numel() > 1 and bool((r[1:] <= r[:-1]).any()))):
            raise ValueError("rows must be ascending row ids of the matrix")
        kept = int((self.row_ptr[r + 1] - self.row_ptr[r]).sum().item()) if r.numel() else 0
        if kept != self.nnz:
            raise ValueError("keep_rows: a row that is left out holds nonzeros (%d of %d kept)" % (kept, self.nnz))
        row_ptr = torch.cat([self.row_ptr[r], self.row_ptr[-1:]]).contiguous()
        return Csr(int(r.numel()), self.n_cols, row_ptr, self.col_idx, self.val, self.input_index)

    # ---------------------------------------------------------------- builders
    @classmethod
    def from_coo(cls, rows, cols, vals, shape, device="cuda", coords=None):
        """Build from COO arrays (numpy or torch).  ``vals`` may be fp32 or fp64;
        fp64 is rounded to fp32 after ordering, as feeding an fp64 tuple to a
        ``tf.float32`` sparse placeholder does (reference gcn/train_normal.py:50).
        ``coords`` (an [nnz,2] int32 array) may be given instead of rows/cols."""
        dev = _require_cuda(device)
        lib = _lib.load()
        n_rows, n_cols = int(shape[0]), int(shape[1])
        if coords is not None:
            co = torch.as_tensor(np.ascontiguousarray(coords) if isinstance(coords, np.ndarray) else coords)
            co = co.to(device=dev, dtype=torch.int32).contiguous()
            if co.dim() != 2 or co.shape[1] != 2:
                raise ValueError("coords must be [nnz, 2]")
            nnz = co.shape[0]
            row_p = co.data_ptr()
            col_p = co.data_ptr() + 4
            stride = 2
            keep = (co,)
        else:
            r = torch.as_tensor(rows).to(device=dev, dtype=torch.int32).contiguous()
            c = torch.as_tensor(cols).to(device=dev, dtype=torch.int32).contiguous()
            if r.shape != c.shape or r.dim() != 1:
                raise ValueError("rows and cols must be 1-D and equally long")
            nnz = r.numel()
            row_p, col_p, stride = r.data_ptr(), c.data_ptr(), 1
            keep = (r, c)
        v = torch.as_tensor(vals)
        if v.dtype not in (torch.float32, torch.float64):
            v = v.to(torch.float64)
        v = v.to(dev).contiguous()
        if v.numel() != nnz:
            raise ValueError("values and coordinates differ in length")
        row_ptr = torch.empty(n_rows + 1, dtype=torch.int64, device=dev)
        col_idx = torch.empty(nnz, dtype=torch.int32, device=dev)
        val = torch.empty(nnz, dtype=torch.float32, device=dev)
        perm = torch.empty(nnz, dtype=torch.int64, device=dev)
        with torch.cuda.device(dev):
            need = lib.gcnx_csr_from_coo_workspace_bytes(n_rows, nnz)
            ws = torch.empty(max(need, 256), dtype=torch.uint8, device=dev)
            _lib.check(lib.gcnx_csr_from_coo(
                n_rows, n_cols, nnz, C.c_void_p(row_p), C.c_void_p(col_p), stride, _ptr(v),
                1 if v.dtype == torch.float64 else 0, _ptr(row_ptr), _ptr(col_idx), _ptr(val),
                _ptr(perm), _ptr(ws), ws.numel(), _stream()))
        del keep, ws
        return cls(n_rows, n_cols, row_ptr, col_idx, val, perm)

    @classmethod
    def from_tuple(cls, t, device="cuda"):
        """From the reference's sparse tuple ``(coords[nnz,2], values, shape)``
        (gcn/utils.py:93-109)."""
        coords, values, shape = t
        coords = np.ascontiguousarray(np.asarray(coords).reshape(-1, 2).astype(np.int32, copy=False))
        return cls.from_coo(None, None, np.ascontiguousarray(values), shape, device, coords=coords)

    @classmethod
    def from_scipy(cls, m, device="cuda"):
        m = m.tocoo()
        return cls.from_coo(m.row, m.col, m.data, m.shape, device)

    # --------------------------------------------------------------- transpose
    def transpose(self, cache=True):
        """Exact transpose (cached).  Its ``input_index`` still refers to the
        tuple this matrix was built from, so a dropout mask drawn for the
        forward product can be reused by the backward one.  ``cache=False``: a
        fresh matrix that neither is remembered by this one nor remembers it, and
        carries no ``input_index`` (set-up of derived matrices)."""
        if cache and self._t is not None:
            return self._t
        lib = _lib.load()
        dev = self.device
        t_row_ptr = torch.empty(self.n_cols + 1, dtype=torch.int64, device=dev)
        t_col_idx = torch.empty(self.nnz, dtype=torch.int32, device=dev)
        t_val = torch.empty(self.nnz, dtype=torch.float32, device=dev)
        t_perm = torch.empty(self.nnz, dtype=torch.int64, device=dev) if cache else None
        with torch.cuda.device(dev):
            need = lib.gcnx_csr_transpose_workspace_bytes(self.n_rows, self.n_cols, self.nnz)
            ws = torch.empty(max(need, 256), dtype=torch.uint8, device=dev)
            _lib.check(lib.gcnx_csr_transpose(self.struct_ref(), _ptr(t_row_ptr), _ptr(t_col_idx),
                                              _ptr(t_val), _ptr(t_perm), _ptr(ws), ws.numel(),
                                              _stream()))
        del ws
        if not cache:
            return Csr(self.n_cols, self.n_rows, t_row_ptr, t_col_idx, t_val, None)
        idx = self.input_index[t_perm] if self.input_index is not None else t_perm
        t = Csr(self.n_cols, self.n_rows, t_row_ptr, t_col_idx, t_val, idx)
        t._t = self
        self._t = t
        return t

    # ------------------------------------------------------------------ export
    def to_scipy(self):
        import scipy.sparse as sp
        return sp.csr_matrix((self.val.cpu().numpy(), self.col_idx.cpu().numpy(),
                              self.row_ptr.cpu().numpy()), shape=self.shape)
