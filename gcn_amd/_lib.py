"""ctypes binding of ``libgcnhip.so`` (the C ABI declared in include/gcnx.h).

There is no fallback: if the library is missing or a symbol is absent, importing
an op raises.  The library is built in-tree by ``gcn_amd/csrc/Makefile``
(``__graft_entry__.build()``).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# GCNX_LIB overrides the path (kernel-variant experiments during development)
LIB_PATH = os.environ.get("GCNX_LIB") or os.path.join(_HERE, "lib", "libgcnhip.so")

ABI_VERSION = 5
SPMM_STATIC_SCRATCH_BYTES = 32 << 20   # GCNX_SPMM_STATIC_SCRATCH_BYTES of include/gcnx.h

OK = 0
ERR_INVALID_ARG = -1
ERR_WORKSPACE = -2
ERR_INDEX_RANGE = -3
ERR_HIP = -4
ERR_UNSUPPORTED = -5
ERR_COMM = -6
COMM_ID_BYTES = 128


class GcnxError(RuntimeError):
    def __init__(self, code, msg):
        super(GcnxError, self).__init__("gcnx error %d: %s" % (code, msg))
        self.code = code


class DropoutStruct(C.Structure):
    """Mirror of ``struct gcnx_dropout``."""
    _fields_ = [("keep", C.c_float), ("seed", C.c_uint64), ("offset", C.c_uint64),
                ("ld", C.c_int64), ("col0", C.c_int64)]


class HaloExchangeStruct(C.Structure):
    """Mirror of ``struct gcnx_halo_exchange``."""
    _fields_ = [("world", C.c_int32), ("rank", C.c_int32), ("h_send_rows", C.POINTER(C.c_int64)),
                ("h_recv_rows", C.POINTER(C.c_int64)), ("send_idx", C.c_void_p)]


class CsrStruct(C.Structure):
    """Mirror of ``struct gcnx_csr``."""
    _fields_ = [("n_rows", C.c_int64), ("n_cols", C.c_int64), ("nnz", C.c_int64),
                ("row_ptr", C.c_void_p), ("col_idx", C.c_void_p), ("val", C.c_void_p),
                ("spmm_chunk", C.c_int32), ("gather_keep", C.c_void_p)]


_i32, _i64, _u64, _f32, _f64 = C.c_int32, C.c_int64, C.c_uint64, C.c_float, C.c_double
_p, _sz, _int = C.c_void_p, C.c_size_t, C.c_int
_csr_p = C.POINTER(CsrStruct)

# name -> (restype, argtypes); must list every function include/gcnx.h declares.
SIGNATURES = {
    "gcnx_abi_version": (_int, []),
    "gcnx_last_error": (C.c_char_p, []),
    "gcnx_spmm_chunk_nnz": (_int, []),
    "gcnx_csr_from_coo_workspace_bytes": (_sz, [_i64, _i64]),
    "gcnx_csr_from_coo": (_int, [_i64, _i64, _i64, _p, _p, _i64, _p, _int, _p, _p, _p, _p, _p, _sz, _p]),
    "gcnx_csr_transpose_workspace_bytes": (_sz, [_i64, _i64, _i64]),
    "gcnx_csr_transpose": (_int, [_csr_p, _p, _p, _p, _p, _p, _sz, _p]),
    "gcnx_spmm_num_chunks": (_i64, [_i64]),
    "gcnx_spmm_chunk_table": (_int, [_csr_p, _p, _p]),
    "gcnx_spmm_workspace_bytes": (_sz, [_csr_p, _i32]),
    "gcnx_spmm_csr_f32": (_int, [_csr_p, _p, _p, _i64, _p, _i64, _i32, _f32, _f32, _p, _p, _f32,
                                 _int, _p, _sz, _p]),
    "gcnx_spmm_csr_static_f32": (_int, [_csr_p, _p, _p, _i64, _p, _i64, _i32, _f32, _f32, _p, _p, _f32, _int, _p]),
    "gcnx_spmm_csr_split_f32": (_int, [_csr_p, _p, _p, _i64, _p, _i64, _i64, _p, _i64, _i32, _f32, _f32,
                                       _p, _p, _f32, _int, _p, _p, _sz, _p]),
    "gcnx_spmm_signs_supported": (_int, [_i32]),
    "gcnx_spmm_csr_signs_f32": (_int, [_csr_p, _p, _p, _i64, _p, _i64, _i64, _p, _i64, _i32, _f32, _f32, _int,
                                       _p, _p, _i64, _p, _sz, _p]),
    "gcnx_spmm_csr_rows_f32": (_int, [_csr_p, _p, _p, _i64, _p, _i64, _i64, _p, _i64, _i32, _f32, _f32, _int,
                                      _p, _p, _p, _i64, _p, _sz, _p]),
    "gcnx_spmm_factored_supported": (_int, [_i32, _i32]),
    "gcnx_spmm_csr_factored_f32": (_int, [_csr_p, _p, _p, _i64, _p, _i64, _p, _i64, _p, _i64, _i64, _p, _i64, _int,
                                          _i32, _f32, _p, _i64, _i32, _p, _sz, _p]),
    "gcnx_dense_arith": (_int, [_int]),
    "gcnx_gemm_dropout_a_supported": (_int, [_int, _int, _i64, _i64, _i64]),
    "gcnx_gemm_dropout_a_f32": (_int, [_int, _int, _i64, _i64, _i64, _p, _i64, _p, _p, _i64, _p, _i64,
                                       _f32, _f32, _p, _sz, _p]),
    "gcnx_gemm_gated_f32": (_int, [_int, _int, _i64, _i64, _i64, _p, _i64, _p, _i64, _p, _i64, _f32,
                                   _f32, _p, _i64, _f32, _p, _sz, _p]),
    "gcnx_gemm_relu_dropout_supported": (_int, [_int, _int, _i64, _i64, _i64]),
    "gcnx_gemm_relu_dropout_f32": (_int, [_int, _int, _i64, _i64, _i64, _p, _i64, _p, _i64, _p, _i64, _f32,
                                          _f32, _int, _p, _p, _sz, _p]),
    "gcnx_gemm_relu_dropout_signs_f32": (_int, [_int, _int, _i64, _i64, _i64, _p, _i64, _p, _i64, _p, _i64, _f32,
                                                _f32, _int, _p, _p, _i64, _p, _sz, _p]),
    "gcnx_gemm_bitgated_f32": (_int, [_int, _int, _i64, _i64, _i64, _p, _i64, _p, _i64, _p, _i64, _f32,
                                      _f32, _p, _i64, _f32, _p, _sz, _p]),
    "gcnx_pack_sign_bits_f32": (_int, [_i64, _p, _p, _i64, _i32, _p, _i64, _p]),
    "gcnx_gemm_dropout_a_rows_f32": (_int, [_int, _i64, _i64, _i64, _p, _i64, _p, _p, _p, _i64, _p, _i64,
                                            _f32, _f32, _p, _sz, _p]),
    "gcnx_gemm_dropout_at_rows_supported": (_int, [_i64, _i64, _i64]),
    "gcnx_gemm_dropout_at_rows_f32": (_int, [_i64, _i64, _i64, _p, _i64, _p, _p, _p, _i64, _p, _i64,
                                             _f32, _f32, _p, _sz, _p]),
    "gcnx_halo_plan_workspace_bytes": (_sz, [_i64]),
    "gcnx_halo_plan": (_int, [_csr_p, _i64, _i64, _p, _p, _i64, C.POINTER(_i64), _p, _sz, _p]),
    "gcnx_gather_rows_f32": (_int, [_i64, _p, _p, _i64, _i32, _p, _i64, _p]),
    "gcnx_scatter_rows_f32": (_int, [_i64, _p, _p, _i64, _i32, _p, _i64, _p]),
    "gcnx_csr_column_degrees": (_int, [_csr_p, _p, _p]),
    "gcnx_csr_gather_keep": (_int, [_csr_p, _p, _i32, _p, _p]),
    "gcnx_csr_filter_workspace_bytes": (_sz, [_i64]),
    "gcnx_csr_filter_columns_count": (_int, [_csr_p, _p, _p, _p, _p, _sz, _p]),
    "gcnx_csr_filter_columns_fill": (_int, [_csr_p, _p, _sz, _i64, _p, _p, _p, _p]),
    "gcnx_csr_degree_threshold_workspace_bytes": (_sz, []),
    "gcnx_csr_degree_threshold": (_int, [_p, _i64, _i64, _p, _p, _sz, _p]),
    "gcnx_comm_available": (_int, []),
    "gcnx_comm_unique_id": (_int, [_p]),
    "gcnx_comm_create": (_int, [_i32, _i32, _p, C.POINTER(C.c_void_p)]),
    "gcnx_comm_destroy": (_int, [_p]),
    "gcnx_halo_allgather_f32": (_int, [C.POINTER(HaloExchangeStruct), _p, _p, _i64, _i64, _i32, _p, _p, _p]),
    "gcnx_comm_allreduce_sum_f32": (_int, [_p, _p, _i64, _p]),
    "gcnx_profile_spmm_events": (None, [_p, _p]),
    "gcnx_coo_interleave_supports": (_int, [_i32, _p, _p, _p, _i64, _p, _int, _p, _p, _p, _p]),
    "gcnx_gemm_workspace_bytes": (_sz, [_int, _int, _i64, _i64, _i64]),
    "gcnx_gemm_f32": (_int, [_int, _int, _i64, _i64, _i64, _p, _i64, _p, _i64, _p, _i64, _f32, _f32,
                             _int, _p, _sz, _p]),
    "gcnx_xent_workspace_bytes": (_sz, [_i64]),
    "gcnx_masked_softmax_xent_f32": (_int, [_i64, _i32, _p, _i64, _p, _i64, _p, _f64, _p, _p, _i64,
                                            _p, _sz, _p]),
    "gcnx_dropout_dense_f32": (_int, [_i64, _p, _p, _f32, _p, _p]),
    "gcnx_bernoulli_mask_u8": (_int, [_i64, _f32, _u64, _u64, _p, _p]),
    "gcnx_bernoulli_mask_dev_u8": (_int, [_i64, _f32, _u64, _u64, _p, _u64, _p, _p]),
    "gcnx_counter_add_u64": (_int, [_p, _u64, _p]),
    "gcnx_philox_uniform_f32": (_int, [_i64, _f32, _f32, _u64, _u64, _p, _p]),
    "gcnx_dropout_philox_f32": (_int, [_i64, _p, _f32, _u64, _u64, _p, _p, _p]),
    "gcnx_dropout_rows_philox_f32": (_int, [_i64, _i32, _p, _i64, _p, _f32, _u64, _u64, _p, _i64, _p]),
    "gcnx_relu_dropout_bwd_philox_f32": (_int, [_i64, _p, _p, _f32, _u64, _u64, _p, _p]),
    "gcnx_relu_dropout_bwd_f32": (_int, [_i64, _p, _p, _p, _f32, _p, _p]),
    "gcnx_adam_step_f32": (_int, [_i64, _p, _p, _p, _p, _f32, _f32, _f32, _f32, _f32, _i64, _p]),
    "gcnx_l2_loss_f32": (_int, [_i64, _p, _f32, _p, _p, _p]),
    "gcnx_adam_step_dev_f32": (_int, [_i64, _p, _p, _p, _p, _f32, _f32, _f32, _f32, _f32, _p, _p]),
    "gcnx_normalize_csr_f64": (_int, [_csr_p, _i64, _p, _p, _i64, _p, _p, _p]),
    "gcnx_rmat_workspace_bytes": (_sz, []),
    "gcnx_rmat_edges": (_int, [_i32, _i64, _i64, _f64, _f64, _f64, _u64, _i64, _i64, _p, _i64,
                               C.POINTER(_i64), _p, _p]),
    "gcnx_csr_from_keys_workspace_bytes": (_sz, [_i64, _i64]),
    "gcnx_csr_from_keys": (_int, [_p, _i64, _i64, _i64, _int, _p, _p, _p, C.POINTER(_i64), _p, _sz,
                                  _p]),
}

_lib = None


def load():
    """Load the shared library once and attach the prototypes."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "%s not found: the HIP extension has not been built "
            "(run `make -C gcn_amd/csrc` or __graft_entry__.build()); "
            "there is no CPU fallback" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    if lib.gcnx_abi_version() != ABI_VERSION:
        raise ImportError("libgcnhip.so ABI version %d, expected %d" % (lib.gcnx_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


def check(rc):
    if rc != OK:
        raise GcnxError(rc, load().gcnx_last_error().decode("utf-8", "replace"))
