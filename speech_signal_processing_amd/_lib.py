"""ctypes binding of include/ssp.h (libsspgpu.so).  Fails loudly when the library is absent."""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SSP_LIB_PATH") or os.path.join(HERE, "libsspgpu.so")  # override: diagnostic builds only

SSP_OK, SSP_ERR_INVALID, SSP_ERR_UNSUPPORTED, SSP_ERR_HIP, SSP_ERR_NOMEM, SSP_ERR_NODEVICE = 0, -1, -2, -3, -4, -5
HOST, DEVICE = 0, 1
ABI_VERSION = 4
COMM_ID_BYTES = 128


class SspError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libsspgpu error %d: %s" % (code, msg))
        self.code = code


class ssp_mfcc_cfg(C.Structure):
    _fields_ = [
        ("sample_rate", C.c_int32), ("win_len", C.c_int32), ("hop", C.c_int32), ("n_fft", C.c_int32),
        ("n_filt", C.c_int32), ("n_ceps", C.c_int32), ("frame_mode", C.c_int32), ("preemph_mode", C.c_int32),
        ("preemph", C.c_float), ("spec_power", C.c_int32), ("spec_scale", C.c_float), ("log_mode", C.c_int32),
        ("floor_mode", C.c_int32), ("eps", C.c_float), ("top_db", C.c_float), ("delta_order", C.c_int32),
        ("delta_N", C.c_int32), ("cmvn", C.c_int32),
    ]


_P = C.c_void_p
_I64P = C.POINTER(C.c_int64)
_F32P = C.c_void_p   # bulk arrays are passed as raw addresses (host or device)
_MSP = C.POINTER(C.c_float)

# name -> (restype, argtypes); every symbol declared in include/ssp.h
SIGNATURES = {
    "ssp_abi_version": (C.c_int, []),
    "ssp_last_error": (C.c_char_p, []),
    "ssp_ctx_create": (C.c_int, [C.c_int, _P, C.c_int, C.POINTER(_P)]),
    "ssp_ctx_destroy": (C.c_int, [_P]),
    "ssp_debug_poison_lds": (C.c_int, [_P, C.c_uint32]),
    "ssp_ctx_sync": (C.c_int, [_P]),
    "ssp_calibrate": (C.c_int, [_P, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "ssp_ctx_wait_stream": (C.c_int, [_P, _P]),
    "ssp_ctx_signal_stream": (C.c_int, [_P, _P]),
    "ssp_comm_unique_id": (C.c_int, [_P]),
    "ssp_comm_init": (C.c_int, [_P, C.c_int, C.c_int, _P]),
    "ssp_comm_destroy": (C.c_int, [_P]),
    "ssp_comm_info": (C.c_int, [_P, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "ssp_allgather": (C.c_int, [_P, _P, _P, C.c_size_t]),
    "ssp_allreduce_sum": (C.c_int, [_P, _P, C.c_size_t, C.c_int]),
    "ssp_segments_create": (C.c_int, [_P, _I64P, C.c_int64, C.POINTER(_P)]),
    "ssp_segments_destroy": (C.c_int, [_P]),
    "ssp_segments_count": (C.c_int, [_P, _I64P, _I64P]),
    "ssp_segments_read": (C.c_int, [_P, _I64P]),
    "ssp_mfcc_plan_create": (C.c_int, [_P, C.POINTER(ssp_mfcc_cfg), _F32P, _F32P, _F32P, C.POINTER(_P)]),
    "ssp_mfcc_plan_destroy": (C.c_int, [_P]),
    "ssp_mfcc_plan_set_flags": (C.c_int, [_P, C.c_uint32]),
    "ssp_mfcc_num_frames": (C.c_int, [C.POINTER(ssp_mfcc_cfg), C.c_int64, _I64P]),
    "ssp_mfcc_out_dim": (C.c_int, [C.POINTER(ssp_mfcc_cfg), C.POINTER(C.c_int32)]),
    "ssp_mfcc_frame_segments": (C.c_int, [_P, _P, C.POINTER(_P)]),
    "ssp_mfcc_run": (C.c_int, [_P, _P, _P, _F32P, _F32P, C.c_int, C.c_int, _MSP]),
    "ssp_mfcc_run_i16": (C.c_int, [_P, _P, _P, _F32P, _F32P, C.c_int, C.c_int, _MSP]),
    "ssp_enframe": (C.c_int, [_P, _F32P, C.c_int64, C.c_int32, C.c_int32, _F32P, _F32P, C.c_int, _MSP]),
    "ssp_cepstrum": (C.c_int, [_P, _F32P, C.c_int64, C.c_int32, _F32P, C.c_int32, _F32P, C.c_int32, C.c_int32, C.c_int32,
                               C.c_float, _F32P, C.c_int, _MSP]),
    "ssp_spectrum_abs": (C.c_int, [_P, _F32P, C.c_int64, C.c_int32, C.c_float, C.c_int32, _F32P, C.c_int, _MSP]),
    "ssp_delta": (C.c_int, [_P, _F32P, _P, C.c_int32, C.c_int32, _F32P, C.c_int, _MSP]),
    "ssp_cmvn": (C.c_int, [_P, _F32P, _P, C.c_int32, _F32P, C.c_int, _MSP]),
    "ssp_plp_post": (C.c_int, [_P, _F32P, _P, C.c_int32, C.c_float, C.c_int32, C.c_int32, C.c_float, _F32P, C.c_int, _MSP]),
    "ssp_gmm_pack": (C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P, C.c_int32, C.POINTER(_P)]),
    "ssp_gmm_destroy": (C.c_int, [_P]),
    "ssp_gmm_score": (C.c_int, [_P, _F32P, _P, _F32P, _F32P, _P, C.c_int, C.c_int, _MSP]),
    "ssp_gmm_last_rescored": (C.c_int, [_P, C.POINTER(C.c_int32)]),
    "ssp_gmm_em_stats": (C.c_int, [_P, C.c_int32, C.c_int32, _P, _P, _P, _F32P, C.c_int64, _P, _P, _P, _P, C.c_int, _MSP]),
    "ssp_dense_forward": (C.c_int, [_P, _F32P, C.c_int64, C.c_int32, _F32P, _F32P, C.c_int32, C.c_int32, _F32P, C.c_int, _MSP]),
    "ssp_dnn_create": (C.c_int, [_P, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_int32), C.POINTER(_P)]),
    "ssp_dnn_destroy": (C.c_int, [_P]),
    "ssp_dnn_forward": (C.c_int, [_P, _F32P, C.c_int64, _F32P, C.c_int, _MSP]),
    "ssp_dtw_distances": (C.c_int, [_P, _F32P, _P, _F32P, _P, C.c_int32, C.c_int32, _F32P, C.c_int, _MSP]),
    "ssp_fastdtw_distances": (C.c_int, [_P, _F32P, _P, _F32P, _P, C.c_int32, C.c_void_p, _MSP]),
    "ssp_dtw_path": (C.c_int, [_P, _F32P, C.c_int64, _F32P, C.c_int64, C.c_int32, _P, _P, _P, _P]),
    "ssp_centroids": (C.c_int, [_P, _F32P, _P, C.c_int64, C.c_int32, C.c_int32, _F32P, C.c_int, _MSP]),
    "ssp_cosine_identify": (C.c_int, [_P, _F32P, C.c_int64, C.c_int32, _F32P, C.c_int32, _F32P, _P, _F32P, C.c_int, _MSP]),
    "ssp_cosine_identify2": (C.c_int, [_P, _F32P, C.c_int64, C.c_int32, _F32P, C.c_int32, _F32P, _P, _F32P, C.c_int, C.c_int, _MSP]),
    "ssp_cosine_last_auto": (C.c_int, [_P, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "ssp_gmm_last_auto": (C.c_int, [_P, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_float)]),
    "ssp_cosine_last_rescored": (C.c_int, [_P, C.POINTER(C.c_int32)]),
    "ssp_cosine_last_split_rows": (C.c_int, [_P, C.POINTER(C.c_int32)]),
}

_lib = None


def load():
    """Load libsspgpu.so and bind every symbol.  Raises if the library has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "libsspgpu.so is missing (%s). Build it with `python -m speech_signal_processing_amd.build` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback." % LIB_PATH)
    # ONE HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.so.7 / libhsa-runtime64 and whichever
    # copy initialises first owns the device, the other then reports "no ROCm-capable device".  Importing torch first
    # makes the dynamic loader resolve libsspgpu.so's NEEDED libamdhip64.so.7 (same SONAME) to torch's copy.
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    # the ABI first: an older or variant library (SSP_LIB_PATH) gets the rebuild message, not a bare AttributeError on a newer symbol
    lib.ssp_abi_version.restype = C.c_int
    lib.ssp_abi_version.argtypes = []
    if lib.ssp_abi_version() != ABI_VERSION:
        raise ImportError("libsspgpu.so ABI version %d != %d (rebuild: python -m speech_signal_processing_amd.build)" % (lib.ssp_abi_version(), ABI_VERSION))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc):
    if rc == SSP_OK:
        return
    msg = load().ssp_last_error().decode("utf-8", "replace")
    if rc == SSP_ERR_INVALID:
        raise ValueError(msg)
    if rc == SSP_ERR_UNSUPPORTED:
        raise NotImplementedError(msg)
    raise SspError(rc, msg)
