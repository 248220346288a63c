"""ctypes binding of libsfm_gpu.so (the C ABI declared in include/sfm_gpu.h).

There is no CPU fallback: if the library is missing or no GPU is present the
calls raise.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsfm_gpu.so")
# the same sources built with the kernels' timing ablations (bench.py's ceiling
# measurement and the profiling scripts only; never the product path)
TIMING_LIB_PATH = os.path.join(_HERE, "libsfm_gpu_timing.so")

DMATCH_DTYPE = np.dtype(
    [("queryIdx", "<i4"), ("trainIdx", "<i4"), ("imgIdx", "<i4"), ("distance", "<f4")]
)

OK = 0
ERR_ARG, ERR_HIP, ERR_NOT_U8, ERR_TOO_MANY, ERR_UNSUPPORTED, ERR_NOMEM, ERR_CAPACITY = (
    -1, -2, -3, -4, -5, -6, -7)
DESC_F32, DESC_U8 = 0, 1
MODEL_ESSENTIAL, MODEL_FUNDAMENTAL = 0, 1
RANSAC_ADAPTIVE, RANSAC_FIXED = 0, 1


class MatchParams(ctypes.Structure):
    _fields_ = [
        ("ratio", ctypes.c_double),
        ("model", ctypes.c_int32),
        ("mode", ctypes.c_int32),
        ("K", ctypes.c_double * 9),
        ("prob", ctypes.c_double),
        ("threshold_px", ctypes.c_double),
        ("max_iters", ctypes.c_int32),
        ("n_hyp", ctypes.c_int32),
    ]


class SfmGpuError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("sfm_gpu error %d: %s" % (code, msg))
        self.code = code


_lib = None
_timing_lib = None


def lib():
    """Loads libsfm_gpu.so; raises if it has not been built."""
    global _lib
    if _lib is None:
        _lib = _load(LIB_PATH)
    return _lib


def timing_lib():
    """Loads libsfm_gpu_timing.so (kernels with their timing ablations)."""
    global _timing_lib
    if _timing_lib is None:
        _timing_lib = _load(TIMING_LIB_PATH)
    return _timing_lib


def _load(path):
    if not os.path.exists(path):
        raise RuntimeError(
            "%s not found: build it with `python -m sfm_amd.build` (hipcc, gfx950)" % path)
    L = ctypes.CDLL(path)
    c = ctypes
    vp = c.c_void_p
    L.sfm_gpu_version.restype = c.c_char_p
    L.sfm_gpu_create.argtypes = [c.c_int, vp, c.POINTER(vp)]
    L.sfm_gpu_destroy.argtypes = [vp]
    L.sfm_gpu_destroy.restype = None
    L.sfm_gpu_last_error_string.argtypes = [vp]
    L.sfm_gpu_last_error_string.restype = c.c_char_p
    L.sfm_gpu_synchronize.argtypes = [vp]
    L.sfm_gpu_knn_match.argtypes = [vp, vp, c.c_int, c.c_size_t, vp, c.c_int, c.c_size_t,
                                    c.c_int, c.c_int, c.c_int, vp, vp]
    L.sfm_gpu_ratio_test.argtypes = [vp, vp, c.c_int, c.c_int, c.c_double, vp, vp]
    L.sfm_gpu_ransac.argtypes = [vp, vp, vp, c.c_int, vp, c.c_int, c.c_double, c.c_double,
                                 c.c_int, c.c_int, vp, c.c_int, vp, vp, vp, vp, vp]
    L.sfm_gpu_ransac_candidates.argtypes = [vp, vp, c.c_int, vp]
    L.sfm_gpu_pnp_ransac.argtypes = [vp, vp, vp, c.c_int, vp, c.c_int, c.c_double, c.c_double, vp, vp, vp, vp, vp,
                                     vp, vp]
    L.sfm_gpu_image_create.argtypes = [vp, vp, c.c_int, c.c_size_t, c.c_int, c.c_int, vp, c.c_int,
                                       c.POINTER(vp)]
    L.sfm_gpu_image_destroy.argtypes = [vp, vp]
    L.sfm_gpu_image_destroy.restype = None
    L.sfm_gpu_image_rows.argtypes = [vp]
    L.sfm_gpu_image_packed_bytes.argtypes = [c.c_int]
    L.sfm_gpu_image_packed_bytes.restype = c.c_size_t
    L.sfm_gpu_image_prepare_into.argtypes = [vp, vp, c.c_int, c.c_size_t, c.c_int, c.c_int, vp,
                                             c.c_int, vp]
    L.sfm_gpu_image_prepare_many.argtypes = [vp, vp, vp, c.c_size_t, c.c_int, c.c_int, vp, vp, c.c_int]
    L.sfm_gpu_image_wrap.argtypes = [vp, vp, c.c_int, c.c_int, c.POINTER(vp)]
    L.sfm_gpu_check_inputs.argtypes = [vp]
    L.sfm_gpu_knn_match_images.argtypes = [vp, vp, vp, c.c_int, vp, vp, c.c_int]
    L.sfm_gpu_knn_match_images_noalloc.argtypes = [vp, vp, vp, c.c_int, vp, vp]
    L.sfm_gpu_match_pairs.argtypes = [vp, vp, c.c_int, vp, c.c_int, c.POINTER(MatchParams), vp, vp, vp,
                                      c.c_int]
    L.sfm_gpu_match_pairs_pose.argtypes = [vp, vp, c.c_int, vp, c.c_int, c.POINTER(MatchParams), vp, vp, vp,
                                           c.c_int, c.c_double, vp, vp, vp, vp]
    L.sfm_gpu_match_pairs_begin.argtypes = [vp, vp, c.c_int, vp, c.c_int, c.POINTER(MatchParams)]
    L.sfm_gpu_match_pairs_begin_records.argtypes = [vp, vp, c.c_int, vp, c.c_int, c.POINTER(MatchParams), vp,
                                                    c.c_int, vp]
    L.sfm_gpu_match_pairs_end.argtypes = [vp, vp, vp, c.c_int]
    L.sfm_gpu_stream_wait_front.argtypes = [vp, vp]
    L.sfm_gpu_profile_enable.argtypes = [vp, c.c_int]
    L.sfm_gpu_profile_reset.argtypes = [vp]
    L.sfm_gpu_profile_select.argtypes = [vp, c.c_char_p]
    L.sfm_gpu_profile_query.argtypes = [vp, c.c_char_p, c.POINTER(c.c_double),
                                        c.POINTER(c.c_int64)]
    L.sfm_gpu_debug_set_knn_variant.argtypes = [vp, c.c_int]
    L.sfm_gpu_debug_redo_count.argtypes = [vp, vp]
    L.sfm_gpu_debug_score_stamps.argtypes = [vp, vp, c.c_int, vp]
    L.sfm_gpu_debug_poison_lds.argtypes = [vp, c.c_uint32]
    L.sfm_gpu_debug_copy_records.argtypes = [vp, vp, c.c_size_t]
    L.sfm_gpu_debug_sweep_stamps.argtypes = [vp, vp, c.c_int]
    L.sfm_gpu_debug_mfma_probe.argtypes = [vp, vp, vp, vp]
    L.sfm_gpu_debug_mfma16_probe.argtypes = [vp, vp, vp, vp]
    L.sfm_gpu_debug_sqrt_probe.argtypes = [vp, vp]
    L.sfm_gpu_debug_ransac_dump.argtypes = [vp, c.c_int, vp, vp, vp, vp]
    L.sfm_gpu_debug_set_score_filter.argtypes = [vp, c.c_int]
    L.sfm_gpu_debug_pose_votes.argtypes = [vp, vp]
    L.sfm_gpu_debug_set_score_prune.argtypes = [vp, c.c_int]
    L.sfm_gpu_debug_set_solver_split.argtypes = [vp, c.c_int]
    L.sfm_gpu_track_join.argtypes = [vp, vp, c.c_int, vp, c.c_int, vp, c.c_int, vp]
    L.sfm_gpu_track_join_sequence.argtypes = [vp, vp, vp, c.c_int, c.c_int, vp, c.c_int, vp]
    L.sfm_gpu_recover_pose.argtypes = [vp, vp, vp, vp, c.c_int, vp, c.c_double, vp, vp, vp, vp]
    return L


def ptr(a):
    """Host pointer of a numpy array, device pointer of an int, or NULL."""
    if a is None:
        return None
    if isinstance(a, int):
        return ctypes.c_void_p(a)
    return a.ctypes.data_as(ctypes.c_void_p)
