"""ctypes binding of libalignsdf_hip.so (the C ABI declared in include/alignsdf_hip.h).

There is no fallback: if the library is missing, or a call fails, an exception is raised.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libalignsdf_hip.so")

MAX_HEADS = 2
MAX_POINT_FEATS = 64
GRID_REFERENCE = 0
GRID_INTEGER = 1
FEATURES_AFFINE = 0
FEATURES_NERF = 1

ERANGE = -6
ENOSURF = -7
ABI_VERSION = 129        # asdf_version() of the library these bindings were written for

# every symbol include/alignsdf_hip.h declares
EXPORTS = (
    "asdf_version", "asdf_strerror", "asdf_last_hip_error", "asdf_device_count", "asdf_decoder_create",
    "asdf_decoder_destroy", "asdf_decoder_set_sample", "asdf_decode_grid", "asdf_decode_grid_box", "asdf_decode_grid_band", "asdf_decode_points", "asdf_neg_bbox",
    "asdf_mc_workspace_bytes", "asdf_mc_count", "asdf_mc_emit", "asdf_icp_workspace_bytes", "asdf_icp_ts",
    "asdf_debug_pack_host", "asdf_decoder_set_classifier", "asdf_decode_points_cls",
    "asdf_icp_ts_enqueue", "asdf_icp_ts_result", "asdf_chamfer",
    "asdf_decoder_set_math", "asdf_decoder_get_math", "asdf_debug_pack_host_f16",
    "asdf_mesh_cc_workspace_bytes", "asdf_mesh_largest_component", "asdf_decoder_status", "asdf_debug_grid_coords", "asdf_decoder_set_refine", "asdf_decoder_time_next_sweep", "asdf_decoder_set_act_scales", "asdf_decoder_get_act_scales", "asdf_mc_count_enqueue", "asdf_mc_result_status", "asdf_decoder_set_audit", "asdf_decoder_set_short_list", "asdf_decoder_set_cluster_list", "asdf_decoder_one_plane_usable", "asdf_icp_set_search", "asdf_icp_ts_enqueue_range",
    "asdf_zoom_cube", "asdf_decode_grid_band_dev", "asdf_decode_grid_dev", "asdf_mc_emit_bounded",
    "asdf_sample_surface_workspace_bytes", "asdf_sample_surface", "asdf_icp_normalise",
    "asdf_decoder_set_sample_host", "asdf_decoder_set_cluster_timeout", "asdf_set_mfma_shape", "asdf_get_mfma_shape",
    "asdf_debug_pack_host_f16w",
)
MATH_F32, MATH_F16X3 = 0, 1
MAX_CLASSES = 8


class NativeError(RuntimeError):
    def __init__(self, code, what):
        self.code = code
        super().__init__(what)


class DecoderSpec(ctypes.Structure):
    _fields_ = [("latent_size", ctypes.c_int32), ("hidden", ctypes.c_int32), ("num_heads", ctypes.c_int32),
                ("point_feats", ctypes.c_int32 * MAX_HEADS), ("outputs", ctypes.c_int32 * MAX_HEADS),
                ("feature_mode", ctypes.c_int32)]


class HeadParams(ctypes.Structure):
    _fields_ = [("w", ctypes.c_void_p * 5), ("b", ctypes.c_void_p * 5)]


_lib = None


def lib():
    """Load (once) and return the shared library; raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        # build on demand when the toolchain is here (hipcc cross-compiles without a GPU); there is no CPU fallback
        try:
            from . import build_native
            build_native.build()
        except Exception as e:      # noqa: BLE001
            raise NativeError(-100, "libalignsdf_hip.so not found at %s and building it failed (%s) - run "
                                    "`python -m alignsdf_amd.build_native`; there is no CPU fallback" % (LIB_PATH, e))
    # PyTorch ships its own HIP runtime (same soname as /opt/rocm's): it must be the one already in the process when this
    # library is loaded, or two runtimes end up side by side and the one behind this library sees no device
    import torch  # noqa: F401
    L = ctypes.CDLL(LIB_PATH)
    L.asdf_version.restype = ctypes.c_int
    if L.asdf_version() != ABI_VERSION:
        raise NativeError(-101, "libalignsdf_hip.so reports ABI version %d, these bindings need %d - rebuild with "
                                "`python -m alignsdf_amd.build_native --force`" % (L.asdf_version(), ABI_VERSION))
    vp, i32, i64, f32 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_float
    L.asdf_version.restype = ctypes.c_int
    L.asdf_strerror.restype = ctypes.c_char_p
    L.asdf_strerror.argtypes = [ctypes.c_int]
    L.asdf_last_hip_error.restype = ctypes.c_int
    L.asdf_device_count.restype = ctypes.c_int
    L.asdf_decoder_create.argtypes = [ctypes.POINTER(DecoderSpec), ctypes.POINTER(HeadParams), ctypes.POINTER(vp)]
    L.asdf_decoder_destroy.argtypes = [vp]
    L.asdf_decoder_destroy.restype = None
    L.asdf_decoder_set_sample.argtypes = [vp, vp, vp, vp]
    L.asdf_decoder_set_sample_host.argtypes = [vp, vp, vp, vp]
    L.asdf_decoder_set_cluster_timeout.argtypes = [vp, ctypes.c_uint64]
    L.asdf_set_mfma_shape.argtypes = [ctypes.c_int]
    L.asdf_set_mfma_shape.restype = ctypes.c_int
    L.asdf_get_mfma_shape.argtypes = []
    L.asdf_get_mfma_shape.restype = ctypes.c_int
    L.asdf_decode_grid.argtypes = [vp, i32, ctypes.POINTER(f32), f32, i32, vp, vp, vp, vp]
    L.asdf_decode_grid_box.argtypes = [vp, i32, ctypes.POINTER(f32), f32, i32, f32, vp, vp, vp, vp]
    L.asdf_decode_grid_band.argtypes = [vp, i32, ctypes.POINTER(f32), f32, i32, f32, vp, vp, vp, vp]
    L.asdf_zoom_cube.argtypes = [vp, i32, f32, i32, i32, vp, vp]
    L.asdf_decode_grid_band_dev.argtypes = [vp, i32, vp, i32, f32, vp, vp, vp, vp]
    L.asdf_decode_grid_dev.argtypes = [vp, i32, vp, i32, vp, vp, vp, vp]
    L.asdf_mc_emit_bounded.argtypes = [vp, i32, i32, i32, ctypes.c_double, vp, ctypes.c_size_t, vp, ctypes.c_uint32, vp, ctypes.c_uint32, vp]
    L.asdf_decode_points.argtypes = [vp, vp, i64, vp, vp, vp]
    L.asdf_decoder_set_classifier.argtypes = [vp, vp, vp, i32]
    L.asdf_decode_points_cls.argtypes = [vp, vp, i64, vp, vp, vp, vp, vp]
    L.asdf_neg_bbox.argtypes = [vp, i32, i32, i32, vp, vp]
    L.asdf_mc_workspace_bytes.argtypes = [i32, i32, i32, ctypes.POINTER(ctypes.c_size_t)]
    L.asdf_mc_count.argtypes = [vp, i32, i32, i32, ctypes.c_double, vp, ctypes.c_size_t, ctypes.POINTER(ctypes.c_uint32),
                                ctypes.POINTER(ctypes.c_uint32), vp]
    L.asdf_mc_count_enqueue.argtypes = [vp, i32, i32, i32, ctypes.c_double, vp, ctypes.c_size_t, vp, vp]
    L.asdf_mc_result_status.argtypes = [vp, ctypes.c_double]
    L.asdf_mc_emit.argtypes = [vp, i32, i32, i32, ctypes.c_double, vp, ctypes.c_size_t, vp, vp, vp]
    L.asdf_icp_set_search.argtypes = [i32]
    L.asdf_icp_workspace_bytes.argtypes = [i32, i32, ctypes.POINTER(ctypes.c_size_t)]
    L.asdf_icp_ts.argtypes = [vp, i32, vp, i32, i32, ctypes.c_double, ctypes.c_double, vp, ctypes.c_size_t,
                              ctypes.POINTER(ctypes.c_double), vp]
    L.asdf_icp_ts_enqueue.argtypes = [vp, i32, vp, i32, i32, ctypes.c_double, ctypes.c_double, vp, ctypes.c_size_t, vp, vp]
    L.asdf_icp_ts_enqueue_range.argtypes = [vp, i32, vp, i32, i32, i32, ctypes.c_double, ctypes.c_double, vp, ctypes.c_size_t, vp, vp]
    L.asdf_icp_ts_result.argtypes = [vp, ctypes.POINTER(ctypes.c_double), vp]
    L.asdf_decoder_set_math.argtypes = [vp, i32]
    L.asdf_decoder_get_math.argtypes = [vp]
    L.asdf_decoder_set_refine.argtypes = [vp, f32]
    L.asdf_decoder_set_audit.argtypes = [vp, i32, ctypes.c_uint64]
    L.asdf_decoder_set_short_list.argtypes = [vp, i32]
    L.asdf_decoder_set_cluster_list.argtypes = [vp, i32]
    L.asdf_decoder_one_plane_usable.argtypes = [vp]
    L.asdf_decoder_time_next_sweep.argtypes = [vp, vp, vp]
    L.asdf_decoder_set_act_scales.argtypes = [vp, ctypes.POINTER(f32), vp]
    L.asdf_decoder_get_act_scales.argtypes = [vp, ctypes.POINTER(f32)]
    L.asdf_decoder_status.argtypes = [vp, ctypes.POINTER(i32), i32, vp]
    L.asdf_debug_grid_coords.argtypes = [i32, ctypes.POINTER(f32), f32, i32, i64, i64, vp, vp]
    L.asdf_mesh_cc_workspace_bytes.argtypes = [i32, i32, ctypes.POINTER(ctypes.c_size_t)]
    L.asdf_mesh_largest_component.argtypes = [vp, i32, vp, i32, f32, ctypes.POINTER(f32), vp, ctypes.c_size_t, vp, vp, vp, vp]
    L.asdf_chamfer.argtypes = [vp, i32, vp, i32, vp, ctypes.c_size_t, ctypes.POINTER(ctypes.c_double), vp]
    L.asdf_sample_surface_workspace_bytes.argtypes = [i32, ctypes.POINTER(ctypes.c_size_t)]
    L.asdf_sample_surface.argtypes = [vp, vp, i32, vp, i32, f32, ctypes.POINTER(f32), vp, vp, i32, vp, vp, ctypes.c_size_t, vp]
    L.asdf_icp_normalise.argtypes = [vp, i32, vp, i32, vp, vp, vp]
    L.asdf_debug_pack_host_f16.argtypes = [ctypes.POINTER(DecoderSpec), ctypes.POINTER(HeadParams), vp, vp, vp]
    L.asdf_debug_pack_host_f16w.argtypes = [ctypes.POINTER(DecoderSpec), ctypes.POINTER(HeadParams), vp]
    L.asdf_debug_pack_host.argtypes = [ctypes.POINTER(DecoderSpec), ctypes.POINTER(HeadParams)] + [vp] * 6
    _lib = L
    return L


def check(code, what):
    if code != 0:
        L = lib()
        raise NativeError(code, "%s failed: %s (code %d, hip error %d)" % (
            what, L.asdf_strerror(code).decode(), code, L.asdf_last_hip_error()))
