"""ctypes binding of the C-ABI in include/gcn_spmm.h (pygcn_amd/csrc/libgcn_spmm.so).

There is NO fallback: if the HIP library is missing or does not load, every product call raises.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# (GCN_SPMM_LIB: an experiment build of the same ABI — tools/*_variant_sweep.py; never set in product use)
LIB_PATH = os.environ.get("GCN_SPMM_LIB") or os.path.join(_HERE, "csrc", "libgcn_spmm.so")

GCN_ABI_VERSION = 25
GCN_REDUCE_SUM = 0
GCN_REDUCE_MAX = 1
GCN_DEFAULT_ITEM_COST = 64
GCN_DEFAULT_LONG_THRESH = 256
GCN_DTYPE_F32 = 0
GCN_DTYPE_BF16 = 1

c_i32p = ctypes.POINTER(ctypes.c_int32)
c_i64p = ctypes.POINTER(ctypes.c_int64)
c_f32p = ctypes.POINTER(ctypes.c_float)


class GcnCsrPlan(ctypes.Structure):
    """Mirror of `struct gcn_csr_plan` (include/gcn_spmm.h).  Pointers are device addresses."""
    _fields_ = [
        ("n_rows", ctypes.c_int64), ("n_cols", ctypes.c_int64), ("nnz", ctypes.c_int64),
        ("rowptr", ctypes.c_void_p), ("rowptr_is64", ctypes.c_int32),
        ("long_thresh", ctypes.c_int32),
        ("col", ctypes.c_void_p), ("val", ctypes.c_void_p),
        ("n_items", ctypes.c_int64), ("items", ctypes.c_void_p),
        ("n_chunks", ctypes.c_int64), ("chunk_row", ctypes.c_void_p),
        ("chunk_e0", ctypes.c_void_p),
        ("n_long", ctypes.c_int64), ("long_row", ctypes.c_void_p),
        ("long_chunk0", ctypes.c_void_p),
    ]


class GcnEpilogue(ctypes.Structure):
    """Mirror of `struct gcn_epilogue` (include/gcn_spmm.h)."""
    _fields_ = [("bias", ctypes.c_void_p), ("relu", ctypes.c_int32),
                ("dropout_p", ctypes.c_float), ("seed", ctypes.c_uint64),
                ("b_row_nonzero", ctypes.c_void_p), ("b_nnz_rows", ctypes.c_void_p),
                ("b2", ctypes.c_void_p), ("ldb2", ctypes.c_int64), ("b_split", ctypes.c_int64),
                ("c_row_nonzero", ctypes.c_void_p), ("log_softmax", ctypes.c_int32),
                ("seed_dev", ctypes.c_void_p), ("c_row_select", ctypes.c_void_p),
                ("c_skip_zero_rows", ctypes.c_int32), ("drop_row_base", ctypes.c_int64),
                ("c_absmax", ctypes.c_void_p)]


class GcnGemmEpilogue(ctypes.Structure):
    """Mirror of `struct gcn_gemm_epilogue` (include/gcn_spmm.h)."""
    _fields_ = [("bias", ctypes.c_void_p), ("relu", ctypes.c_int32), ("dropout_p", ctypes.c_float),
                ("seed", ctypes.c_uint64), ("seed_dev", ctypes.c_void_p),
                ("mask_src", ctypes.c_void_p), ("ld_mask", ctypes.c_int64),
                ("mask_scale", ctypes.c_float), ("mask_rows", ctypes.c_void_p),
                ("drop_row_base", ctypes.c_int64), ("keep_bits_out", ctypes.c_void_p),
                ("mask_bits", ctypes.c_void_p)]


# every symbol include/gcn_spmm.h declares (tests check that the library exports all of them)
EXPORTS = ("gcn_abi_version", "gcn_last_error", "gcn_plan_count_host", "gcn_plan_fill_host",
           "gcn_spmm_workspace_bytes", "gcn_spmm_csr", "gcn_spmm_csr_ep",
           "gcn_relu_dropout_backward", "gcn_csr_transpose_host",
           "gcn_csr_transpose_workspace_bytes", "gcn_csr_transpose_device",
           "gcn_row_normalize_device", "gcn_gemm_xw256_workspace_bytes", "gcn_gemm_xw256_f32",
           "gcn_bwd_colsum_workspace_bytes", "gcn_relu_dropout_backward_colsum",
           "gcn_log_softmax_backward_colsum", "gcn_plan_device_workspace_bytes",
           "gcn_plan_count_device", "gcn_plan_fill_device", "gcn_coo_to_csr_workspace_bytes",
           "gcn_coo_to_csr_device", "gcn_gemm_xw256_h2_workspace_bytes", "gcn_gemm_xw256_f32_h2",
           "gcn_gemm_bf16_workspace_bytes", "gcn_gemm_xw_bf16",
           "gcn_gemm_atg256_workspace_bytes", "gcn_gemm_atg256_f32",
           "gcn_nll_log_softmax_backward_colsum", "gcn_gemm_atg_bf16_workspace_bytes",
           "gcn_gemm_atg_bf16", "gcn_sddmm_csr", "gcn_rows_pack_count", "gcn_rows_pack_values",
           "gcn_rows_unpack", "gcn_bits_row_counts", "gcn_gemm_xw256_b3_workspace_bytes",
           "gcn_gemm_xw256_f32_b3", "gcn_gemm_atg256_f32_b3", "gcn_gemm_atg256_f32_b3_colsum")

_lib = None


class NativeLibraryError(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeLibraryError(
            f"{LIB_PATH} is missing: build it with `python -m pygcn_amd.build` "
            "(hipcc --offload-arch=gfx950). pygcn_amd has no CPU or PyTorch fallback.")
    try:
        L = ctypes.CDLL(LIB_PATH)
    except OSError as e:   # e.g. libamdhip64 not found
        raise NativeLibraryError(f"cannot load {LIB_PATH}: {e}") from e
    L.gcn_abi_version.restype = ctypes.c_int
    L.gcn_last_error.restype = ctypes.c_char_p
    L.gcn_plan_count_host.restype = ctypes.c_int
    L.gcn_plan_count_host.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int64,
                                      ctypes.c_int32, ctypes.c_int32, c_i64p, c_i64p, c_i64p]
    L.gcn_plan_fill_host.restype = ctypes.c_int
    L.gcn_plan_fill_host.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int64,
                                     ctypes.c_int32, ctypes.c_int32,
                                     ctypes.c_void_p, ctypes.c_int64,
                                     ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64,
                                     ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64]
    L.gcn_spmm_workspace_bytes.restype = ctypes.c_size_t
    L.gcn_spmm_workspace_bytes.argtypes = [ctypes.POINTER(GcnCsrPlan), ctypes.c_int64]
    L.gcn_spmm_csr.restype = ctypes.c_int
    L.gcn_spmm_csr.argtypes = [ctypes.POINTER(GcnCsrPlan), ctypes.c_int, ctypes.c_void_p,
                               ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64,
                               ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t,
                               ctypes.c_void_p]
    L.gcn_spmm_csr_ep.restype = ctypes.c_int
    L.gcn_spmm_csr_ep.argtypes = [ctypes.POINTER(GcnCsrPlan), ctypes.c_int, ctypes.c_void_p,
                                  ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64,
                                  ctypes.POINTER(GcnEpilogue), ctypes.c_void_p, ctypes.c_size_t,
                                  ctypes.c_void_p]
    L.gcn_relu_dropout_backward.restype = ctypes.c_int
    L.gcn_relu_dropout_backward.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                            ctypes.c_void_p, ctypes.c_int64, ctypes.c_float,
                                            ctypes.c_void_p]
    L.gcn_csr_transpose_host.restype = ctypes.c_int
    L.gcn_csr_transpose_host.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p,
                                         ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64,
                                         ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    L.gcn_csr_transpose_workspace_bytes.restype = ctypes.c_size_t
    L.gcn_csr_transpose_workspace_bytes.argtypes = [ctypes.c_int64, ctypes.c_int64, ctypes.c_int64]
    L.gcn_csr_transpose_device.restype = ctypes.c_int
    L.gcn_csr_transpose_device.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p,
                                           ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64,
                                           ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p,
                                           ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t,
                                           ctypes.c_void_p]
    L.gcn_row_normalize_device.restype = ctypes.c_int
    L.gcn_row_normalize_device.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p,
                                           ctypes.c_int64, ctypes.c_void_p]
    L.gcn_bwd_colsum_workspace_bytes.restype = ctypes.c_size_t
    L.gcn_bwd_colsum_workspace_bytes.argtypes = [ctypes.c_int64, ctypes.c_int64, ctypes.c_int]
    L.gcn_relu_dropout_backward_colsum.restype = ctypes.c_int
    L.gcn_relu_dropout_backward_colsum.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                                   ctypes.c_void_p,
                                                   ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64,
                                                   ctypes.c_float, ctypes.c_void_p, ctypes.c_void_p,
                                                   ctypes.c_int,
                                                   ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    L.gcn_log_softmax_backward_colsum.restype = ctypes.c_int
    L.gcn_log_softmax_backward_colsum.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                                  ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64,
                                                  ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p,
                                                  ctypes.c_int,
                                                  ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    L.gcn_nll_log_softmax_backward_colsum.restype = ctypes.c_int
    L.gcn_nll_log_softmax_backward_colsum.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                                      ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                                      ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p,
                                                      ctypes.c_size_t, ctypes.c_void_p]
    L.gcn_gemm_xw256_workspace_bytes.restype = ctypes.c_size_t
    L.gcn_gemm_xw256_workspace_bytes.argtypes = []
    L.gcn_gemm_xw256_f32.restype = ctypes.c_int
    L.gcn_gemm_xw256_f32.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64,
                                     ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p,
                                     ctypes.c_size_t, ctypes.c_void_p]
    L.gcn_plan_device_workspace_bytes.restype = ctypes.c_size_t
    L.gcn_plan_device_workspace_bytes.argtypes = [ctypes.c_int64]
    L.gcn_plan_count_device.restype = ctypes.c_int
    L.gcn_plan_count_device.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_int32,
                                        ctypes.c_int32, ctypes.c_void_p, ctypes.c_size_t,
                                        ctypes.c_void_p, ctypes.c_void_p]
    L.gcn_plan_fill_device.restype = ctypes.c_int
    L.gcn_plan_fill_device.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_int32,
                                       ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p,
                                       ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64,
                                       ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]
    L.gcn_coo_to_csr_workspace_bytes.restype = ctypes.c_size_t
    L.gcn_coo_to_csr_workspace_bytes.argtypes = [ctypes.c_int64, ctypes.c_int64, ctypes.c_int64]
    L.gcn_coo_to_csr_device.restype = ctypes.c_int
    L.gcn_coo_to_csr_device.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                        ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int,
                                        ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t,
                                        ctypes.c_void_p]
    L.gcn_gemm_xw256_h2_workspace_bytes.restype = ctypes.c_size_t
    L.gcn_gemm_xw256_h2_workspace_bytes.argtypes = []
    L.gcn_gemm_xw256_f32_h2.restype = ctypes.c_int
    L.gcn_gemm_xw256_f32_h2.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p,
                                        ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64,
                                        ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(GcnGemmEpilogue),
                                        ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    L.gcn_gemm_xw256_b3_workspace_bytes.restype = ctypes.c_size_t
    L.gcn_gemm_xw256_b3_workspace_bytes.argtypes = []
    L.gcn_gemm_xw256_f32_b3.restype = ctypes.c_int
    L.gcn_gemm_xw256_f32_b3.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p,
                                        ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64,
                                        ctypes.c_void_p, ctypes.POINTER(GcnGemmEpilogue),
                                        ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    L.gcn_gemm_atg256_f32_b3.restype = ctypes.c_int
    L.gcn_gemm_atg256_f32_b3.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p,
                                         ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p,
                                         ctypes.c_int64, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    L.gcn_gemm_atg256_f32_b3_colsum.restype = ctypes.c_int
    L.gcn_gemm_atg256_f32_b3_colsum.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p,
                                                ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p,
                                                ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t,
                                                ctypes.c_void_p]
    L.gcn_gemm_bf16_workspace_bytes.restype = ctypes.c_size_t
    L.gcn_gemm_bf16_workspace_bytes.argtypes = [ctypes.c_int64, ctypes.c_int64]
    L.gcn_gemm_xw_bf16.restype = ctypes.c_int
    L.gcn_gemm_xw_bf16.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64,
                                   ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64,
                                   ctypes.c_int64, ctypes.POINTER(GcnGemmEpilogue), ctypes.c_void_p,
                                   ctypes.c_size_t, ctypes.c_void_p]
    L.gcn_gemm_atg256_workspace_bytes.restype = ctypes.c_size_t
    L.gcn_gemm_atg256_workspace_bytes.argtypes = [ctypes.c_int64]
    L.gcn_gemm_atg256_f32.restype = ctypes.c_int
    L.gcn_gemm_atg256_f32.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p,
                                      ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p,
                                      ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p,
                                      ctypes.c_size_t, ctypes.c_void_p]
    L.gcn_gemm_atg_bf16_workspace_bytes.restype = ctypes.c_size_t
    L.gcn_gemm_atg_bf16_workspace_bytes.argtypes = [ctypes.c_int64, ctypes.c_int64, ctypes.c_int64]
    L.gcn_gemm_atg_bf16.restype = ctypes.c_int
    L.gcn_gemm_atg_bf16.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p,
                                    ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64,
                                    ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p,
                                    ctypes.c_size_t, ctypes.c_void_p]
    L.gcn_sddmm_csr.restype = ctypes.c_int
    L.gcn_sddmm_csr.argtypes = [ctypes.POINTER(GcnCsrPlan), ctypes.c_int, ctypes.c_void_p, ctypes.c_int64,
                                ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p,
                                ctypes.c_void_p]
    vp, i64 = ctypes.c_void_p, ctypes.c_int64
    L.gcn_rows_pack_count.restype = ctypes.c_int
    L.gcn_rows_pack_count.argtypes = [ctypes.c_int, vp, i64, vp, i64, i64, vp, vp, vp]
    L.gcn_rows_pack_values.restype = ctypes.c_int
    L.gcn_rows_pack_values.argtypes = [ctypes.c_int, vp, i64, vp, i64, i64, vp, vp, vp]
    L.gcn_rows_unpack.restype = ctypes.c_int
    L.gcn_rows_unpack.argtypes = [ctypes.c_int, vp, vp, vp, i64, i64, vp, i64, vp]
    L.gcn_bits_row_counts.restype = ctypes.c_int
    L.gcn_bits_row_counts.argtypes = [vp, i64, i64, vp, vp]
    if L.gcn_abi_version() != GCN_ABI_VERSION:
        raise NativeLibraryError(f"{LIB_PATH}: ABI version {L.gcn_abi_version()} != "
                                 f"{GCN_ABI_VERSION}; rebuild with `python -m pygcn_amd.build`")
    _lib = L
    return L


def check(rc, what):
    """Non-zero C-ABI return -> RuntimeError (PyTorch's convention for the ops it replaces)."""
    if rc != 0:
        msg = lib().gcn_last_error()
        raise RuntimeError(f"{what} failed (code {rc}): {msg.decode() if msg else ''}")
