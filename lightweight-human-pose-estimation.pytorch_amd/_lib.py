"""ctypes binding of the C-ABI library (include/lwpose.h).  Fails loudly when the HIP library
is missing: there is no CPU fallback anywhere in the product path."""
import ctypes as C
import os

PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(PKG, "liblwpose_hip.so")

LWP_OK, LWP_ERR_ARG, LWP_ERR_HIP, LWP_ERR_STATE, LWP_ERR_CAPACITY, LWP_ERR_NOGPU, LWP_ERR_UNBOUND = 0, -1, -2, -3, -4, -5, -6
MEM_HOST, MEM_DEVICE = 0, 1
F32, BF16 = 0, 1

EXPORTS = [
    "lwp_version", "lwp_param_count", "lwp_param_spec", "lwp_create", "lwp_destroy", "lwp_last_error",
    "lwp_set_capacity", "lwp_load_weights", "lwp_weights_blob_bytes", "lwp_weights_blob_export",
    "lwp_weights_blob_import", "lwp_forward", "lwp_upsample", "lwp_extract_keypoints", "lwp_group_keypoints",
    "lwp_infer_poses", "lwp_infer_poses_async", "lwp_fetch_poses", "lwp_time_pipeline", "lwp_profile_classes",
    "lwp_synchronize", "lwp_poses_from_maps", "lwp_layer_count", "lwp_layer_info", "lwp_debug_layer_output",
    "lwp_profile_launches", "lwp_debug_time_layer", "lwp_pipeline_submit", "lwp_pipeline_fetch", "lwp_multiscale_accumulate",
    "lwp_preprocess_dims", "lwp_preprocess_u8", "lwp_scale_dims", "lwp_preprocess_scaled_u8", "lwp_debug_layer_variant", "lwp_set_stream", "lwp_preprocess_scaled_f32", "lwp_debug_frames_per_pass", "lwp_debug_post_counts",
]


class CapacityError(RuntimeError):
    """A peak / key-point / connection / pose list overflowed its configured capacity."""


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "lwpose_amd: %s is missing — build it first (python __graft_entry__.py build, or "
            "python lightweight-human-pose-estimation.pytorch_amd/build.py). There is no CPU fallback." % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp, ip, i64p, fp, dp = C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int64), C.POINTER(C.c_float), C.POINTER(C.c_double)
    L.lwp_version.restype = C.c_int
    L.lwp_param_count.argtypes = [C.c_int] * 4
    L.lwp_param_spec.argtypes = [C.c_int] * 5 + [C.c_char_p, C.c_int, i64p, ip, ip]
    L.lwp_create.argtypes = [C.c_int] * 6 + [C.POINTER(vp)]
    L.lwp_destroy.argtypes = [vp]
    L.lwp_last_error.argtypes = [vp]
    L.lwp_last_error.restype = C.c_char_p
    L.lwp_set_capacity.argtypes = [vp] + [C.c_int] * 4
    L.lwp_load_weights.argtypes = [vp, C.POINTER(C.c_char_p), C.POINTER(vp), i64p, ip, C.c_int]
    L.lwp_weights_blob_bytes.argtypes = [vp, C.POINTER(C.c_size_t)]
    L.lwp_weights_blob_export.argtypes = [vp, vp, C.c_size_t]
    L.lwp_weights_blob_import.argtypes = [vp, vp, C.c_size_t]
    L.lwp_forward.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(vp), C.c_int]
    L.lwp_upsample.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_int]
    L.lwp_extract_keypoints.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int64, C.c_int64, i64p, i64p, fp, C.c_int, ip]
    L.lwp_group_keypoints.argtypes = [vp, vp, ip, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_int, ip]
    L.lwp_infer_poses.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, ip, vp, C.c_int, vp, C.c_int, ip]
    L.lwp_infer_poses_async.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    L.lwp_fetch_poses.argtypes = [vp, ip, vp, C.c_int, vp, C.c_int, ip]
    L.lwp_time_pipeline.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, fp]
    L.lwp_profile_classes.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, fp, ip]
    L.lwp_synchronize.argtypes = [vp]
    L.lwp_poses_from_maps.argtypes = [vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, ip, vp, C.c_int, vp, C.c_int, ip]
    L.lwp_profile_launches.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, fp, ip, C.c_int, ip]
    L.lwp_debug_time_layer.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, fp]
    L.lwp_pipeline_submit.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    L.lwp_pipeline_fetch.argtypes = [vp, C.c_int, ip, vp, C.c_int, vp, C.c_int, ip]
    L.lwp_multiscale_accumulate.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, ip, C.c_int, C.c_int, C.c_int, vp, C.c_int, C.c_int]
    dp = C.POINTER(C.c_double)
    L.lwp_preprocess_dims.argtypes = [C.c_int] * 4 + [ip] * 5 + [dp]
    L.lwp_preprocess_u8.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, dp, dp, C.c_double, vp]
    L.lwp_scale_dims.argtypes = [C.c_int, C.c_int, C.c_double, C.c_int, C.c_int] + [ip] * 5
    L.lwp_preprocess_scaled_u8.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, C.c_int, dp, dp, C.c_double, vp]
    L.lwp_preprocess_scaled_f32.argtypes = L.lwp_preprocess_scaled_u8.argtypes
    L.lwp_layer_count.argtypes = [vp]
    L.lwp_layer_info.argtypes = [vp, C.c_int, C.c_char_p, C.c_int] + [ip] * 6 + [i64p]
    L.lwp_debug_layer_output.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_size_t, ip]
    L.lwp_debug_layer_variant.argtypes = [vp, C.c_int, C.c_char_p, C.c_int]
    L.lwp_set_stream.argtypes = [vp, vp, C.c_int]
    L.lwp_debug_frames_per_pass.argtypes = [vp, C.c_int, C.c_int, C.c_int]
    L.lwp_debug_post_counts.argtypes = [vp, C.c_int] + [C.POINTER(C.c_int)] * 4
    for name in EXPORTS:
        if name not in ("lwp_last_error",):
            getattr(L, name).restype = C.c_int
    _lib = L
    return L


def check(rc, handle=None):
    if rc == LWP_OK:
        return
    msg = lib().lwp_last_error(handle)
    msg = msg.decode("utf-8", "replace") if msg else "error %d" % rc
    if rc == LWP_ERR_ARG:
        raise ValueError(msg)
    if rc == LWP_ERR_UNBOUND:
        raise UnboundLocalError(msg)
    if rc == LWP_ERR_CAPACITY:
        raise CapacityError(msg)
    raise RuntimeError("lwpose (%d): %s" % (rc, msg))


def param_spec(nref=1, num_channels=128, num_heatmaps=19, num_pafs=38):
    """[(key, shape tuple, role)] from the library's own table (no GPU needed)."""
    L = lib()
    n = L.lwp_param_count(nref, num_channels, num_heatmaps, num_pafs)
    if n < 0:
        raise ValueError("bad network shape")
    out = []
    name = C.create_string_buffer(256)
    shape = (C.c_int64 * 4)()
    nd, role = C.c_int(), C.c_int()
    for i in range(n):
        check(L.lwp_param_spec(nref, num_channels, num_heatmaps, num_pafs, i, name, 256, shape, C.byref(nd), C.byref(role)))
        out.append((name.value.decode(), tuple(shape[d] for d in range(nd.value)), role.value))
    return out


class Handle(object):
    """Owns one lwp_handle (one device, one stream)."""

    def __init__(self, device_id=0, nref=1, num_channels=128, num_heatmaps=19, num_pafs=38, dtype=F32):
        self._h = C.c_void_p()
        self.nref, self.num_channels, self.num_heatmaps, self.num_pafs = nref, num_channels, num_heatmaps, num_pafs
        self.device_id = device_id
        check(lib().lwp_create(device_id, nref, num_channels, num_heatmaps, num_pafs, dtype, C.byref(self._h)))

    @property
    def ptr(self):
        return self._h

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            lib().lwp_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
