"""ctypes binding of libtrexhip.so (include/trexhip.h) for the Python-side callers in this repo
(tests, bench.py).  The C ABI is the product boundary; this module is only plumbing.

There is NO CPU fallback: if the HIP library is missing or a call fails, this raises.
"""
import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TREXHIP_LIB_PATH") or os.path.join(_HERE, "libtrexhip.so")      # (the override: dev builds of tools/build_dev.sh)
_LIB = None

RUN_DTYPE = np.dtype([("x0", "<u2"), ("x1", "<u2"), ("y", "<u2"), ("pad", "<u2")])
BLOB_DTYPE = np.dtype([
    ("run_begin", "<u4"), ("n_runs", "<u4"), ("pix_begin", "<u4"), ("n_pixels", "<u4"),
    ("x0", "<u2"), ("y0", "<u2"), ("x1", "<u2"), ("y1", "<u2"),
    ("bid", "<u4"), ("px_min_max", "<u4"), ("parent", "<u4"), ("flags", "<u4"),
    ("m10", "<u8"), ("m01", "<u8"), ("m20", "<u8"), ("m11", "<u8"), ("m02", "<u8"),
    ("sp", "<u8"), ("spx", "<u8"), ("spy", "<u8"),
])
INFO_DTYPE = np.dtype([
    ("n_blobs", "<u4"), ("n_runs", "<u4"), ("n_pixels", "<u4"),
    ("blob_begin", "<u4"), ("run_begin", "<u4"), ("pix_begin", "<u4"),
    ("n_raw_runs", "<u4"), ("n_raw_blobs", "<u4"), ("flags", "<u4"), ("reserved", "<u4", (3,)),
])
assert BLOB_DTYPE.itemsize == 104 and RUN_DTYPE.itemsize == 8 and INFO_DTYPE.itemsize == 48

CNN_FP32, CNN_BF16X6, CNN_BF16X3, CNN_FP16X3 = 0, 1, 2, 3
STAGE_ROWS, STAGE_SEGMENT_ALL, STAGE_CONV2, STAGE_CONV3, STAGE_CNN_ALL, STAGE_CROPS, STAGE_POSTURE = 0, 1, 2, 3, 4, 5, 6
STAGE_UPLOAD_COPY, STAGE_UPLOAD_DMA = 8, 9       # host-input legs (per frame): pageable -> pinned copy (host ms), DMA (event ms)


class Params(C.Structure):
    _fields_ = [
        ("device", C.c_int32), ("width", C.c_int32), ("height", C.c_int32),
        ("max_batch", C.c_int32), ("max_runs", C.c_int32), ("max_blobs", C.c_int32), ("max_pixels", C.c_int32),
        ("threshold", C.c_int32), ("threshold_maximum", C.c_int32),
        ("enable_difference", C.c_int32), ("absolute_difference", C.c_int32),
        ("image_invert", C.c_int32), ("inclusive", C.c_int32), ("zero_is_background", C.c_int32),
        ("connectivity", C.c_int32), ("dilation_size", C.c_int32), ("use_closing", C.c_int32),
        ("closing_size", C.c_int32), ("n_ranges", C.c_int32),
        ("cm_per_pixel", C.c_double), ("ranges", C.c_double * 16),
        ("pixel_encoding", C.c_int32),
        # not implemented: any non-zero value makes trexhip_create return TREXHIP_E_UNSUPPORTED
        ("image_adjust", C.c_int32), ("blur_difference", C.c_int32), ("equalize_histogram", C.c_int32), ("correct_luminance", C.c_int32),
        ("use_adaptive_threshold", C.c_int32), ("device_color_reduce", C.c_int32), ("reserved_", C.c_int32 * 1),
    ]


class LiveParams(C.Structure):
    _fields_ = [("threshold", C.c_int32), ("threshold_maximum", C.c_int32), ("inclusive", C.c_int32),
                ("enable_difference", C.c_int32), ("absolute_difference", C.c_int32), ("image_invert", C.c_int32), ("zero_is_background", C.c_int32),
                ("n_ranges", C.c_int32), ("cm_per_pixel", C.c_double), ("ranges", C.c_double * 16)]


class PostureParams(C.Structure):
    _fields_ = [("outline_resample", C.c_float), ("outline_smooth_samples", C.c_int32), ("outline_smooth_step", C.c_int32),
                ("outline_approximate", C.c_int32), ("outline_curvature_range_ratio", C.c_float),
                ("midline_walk_offset", C.c_float), ("max_points", C.c_int32),
                ("posture_closing_steps", C.c_int32), ("peak_mode", C.c_int32), ("posture_direction_smoothing", C.c_int32)]


class MidlineParams(C.Structure):
    _fields_ = [("midline_resolution", C.c_int32), ("midline_stiff_percentage", C.c_float), ("midline_invert", C.c_int32),
                ("midline_start_with_head", C.c_int32)]


class SplitParams(C.Structure):
    _fields_ = [("track_threshold", C.c_int32), ("track_posture_threshold", C.c_int32), ("calculate_posture", C.c_int32), ("algorithm", C.c_int32),
                ("blob_split_max_shrink", C.c_float), ("blob_split_global_shrink_limit", C.c_float), ("n_ranges", C.c_int32), ("reserved_", C.c_int32),
                ("size_ranges", C.c_double * 16)]


SPLIT_INFO_DTYPE = np.dtype([("threshold", "<i4"), ("effective_threshold", "<i4"), ("status", "<i4"), ("initial_action", "<i4"), ("n_result", "<i4"),
                             ("n_evaluated", "<i4"), ("min_pixel", "<i4"), ("max_pixel", "<i4"), ("first_size", "<f4"), ("reserved_", "<f4"),
                             ("min_size_bound", "<f8")])
MIDLINE_INFO_DTYPE = np.dtype([("status", "<i4"), ("n", "<i4"), ("len", "<f4"), ("angle", "<f4"), ("offx", "<f4"), ("offy", "<f4"),
                               ("reserved", "<i4", (2,))])
POSTURE_INFO_DTYPE = np.dtype([("status", "<i4"), ("n_outline", "<i4"), ("n_segments", "<i4"), ("tail_index", "<i4"),
                               ("head_index", "<i4"), ("n_traced", "<i4"), ("reserved", "<i4", (2,))])


class BatchResult(C.Structure):
    _fields_ = [
        ("n_frames", C.c_int32), ("total_blobs", C.c_uint32), ("total_runs", C.c_uint32), ("total_pixels", C.c_uint32),
        ("frames", C.c_void_p), ("blobs", C.c_void_p), ("runs", C.c_void_p), ("pixels", C.c_void_p),
        ("pixel_channels", C.c_uint32), ("reserved_", C.c_uint32),
    ]


ENC_GRAY, ENC_R3G3B2, ENC_RGB8 = 0, 1, 2        # pixel_encoding, order of cmn::meta_encoding_t


class DeviceView(C.Structure):
    _fields_ = [("frames", C.c_void_p), ("blobs", C.c_void_p), ("runs", C.c_void_p), ("pixels", C.c_void_p),
                ("totals", C.c_void_p), ("blob_frame", C.c_void_p)]


class TrexHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libtrexhip error {code}: {msg}")
        self.code = code


# every symbol include/trexhip.h declares (tests check the library exports all of them)
class TrainParams(C.Structure):
    _fields_ = [("max_batch", C.c_int32), ("lr", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float),
                ("bn_momentum", C.c_float), ("dropout", C.c_float), ("precision", C.c_int32), ("seed", C.c_uint64)]


SYMBOLS = [
    "trexhip_abi_version", "trexhip_network_channels", "trexhip_comm_unique_id", "trexhip_comm_create", "trexhip_comm_destroy", "trexhip_comm_rank", "trexhip_comm_world", "trexhip_comm_gather_device", "trexhip_comm_gather_device_on", "trexhip_comm_count_ranks", "trexhip_last_error", "trexhip_default_params", "trexhip_create", "trexhip_destroy",
    "trexhip_set_stream", "trexhip_get_live_params", "trexhip_update_params", "trexhip_set_background", "trexhip_set_background_device", "trexhip_set_background_color", "trexhip_set_background_color_device", "trexhip_generate_average_device", "trexhip_get_background", "trexhip_segment_device",
    "trexhip_segment", "trexhip_segment_color", "trexhip_segment_color_device", "trexhip_rethreshold_device", "trexhip_rethreshold_per_blob_device", "trexhip_fetch_rethreshold", "trexhip_fetch", "trexhip_device_view_get", "trexhip_synchronize",
    "trexhip_profile_enable", "trexhip_profile_read", "trexhip_profile_reset",
    "trexhip_default_posture_params", "trexhip_posture_device", "trexhip_posture_auto_device", "trexhip_pack_frames_v6_device", "trexhip_crops_device", "trexhip_pixel_channels", "trexhip_device_alloc", "trexhip_device_free", "trexhip_copy_to_host", "trexhip_copy_to_device", "trexhip_crops_transformed_device", "trexhip_crops_posture_device", "trexhip_default_midline_params", "trexhip_midline_device", "trexhip_midline_movement_device", "trexhip_default_split_params", "trexhip_split_search_device", "trexhip_export_id_table_device", "trexhip_export_id_table_ex_device", "trexhip_load_weights", "trexhip_set_identity_precision", "trexhip_num_classes", "trexhip_identify_device", "trexhip_identify", "trexhip_identify_guard_stats",
    "trexhip_weight_blob_bytes", "trexhip_trainer_create", "trexhip_trainer_destroy", "trexhip_trainer_set_lr", "trexhip_trainer_steps", "trexhip_train_step_device", "trexhip_train_step", "trexhip_train_eval_device", "trexhip_train_eval", "trexhip_trainer_read", "trexhip_trainer_export",
    "trexhip_lzo1x_bound", "trexhip_lzo1x_compress", "trexhip_pv_write_frames",
]


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise TrexHipError(-1, f"{LIB_PATH} is missing: run __graft_entry__.build() (make -C trex_amd/csrc)")
        L = C.CDLL(LIB_PATH)
        L.trexhip_last_error.restype = C.c_char_p
        L.trexhip_default_params.argtypes = [C.POINTER(Params), C.c_int32, C.c_int32]
        L.trexhip_default_params.restype = None
        L.trexhip_create.argtypes = [C.POINTER(Params), C.POINTER(C.c_void_p)]
        L.trexhip_destroy.argtypes = [C.c_void_p]
        L.trexhip_destroy.restype = None
        L.trexhip_set_stream.argtypes = [C.c_void_p, C.c_void_p]
        L.trexhip_get_live_params.argtypes = [C.c_void_p, C.POINTER(LiveParams)]
        L.trexhip_update_params.argtypes = [C.c_void_p, C.POINTER(LiveParams)]
        L.trexhip_set_background.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
        L.trexhip_set_background_device.argtypes = [C.c_void_p, C.c_void_p]
        L.trexhip_set_background_color.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32]
        L.trexhip_set_background_color_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32]
        L.trexhip_generate_average_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32]
        L.trexhip_get_background.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
        L.trexhip_segment_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
        L.trexhip_segment.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.c_int32, C.c_int32]
        L.trexhip_segment_color.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.c_int32, C.c_int32, C.c_int32, C.c_int32]
        L.trexhip_segment_color_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32]
        L.trexhip_rethreshold_device.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32]
        L.trexhip_rethreshold_per_blob_device.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32]
        L.trexhip_fetch_rethreshold.argtypes = [C.c_void_p, C.POINTER(BatchResult)]
        L.trexhip_fetch.argtypes = [C.c_void_p, C.POINTER(BatchResult)]
        L.trexhip_device_view_get.argtypes = [C.c_void_p, C.POINTER(DeviceView)]
        L.trexhip_synchronize.argtypes = [C.c_void_p]
        L.trexhip_profile_enable.argtypes = [C.c_void_p, C.c_int32]
        L.trexhip_profile_read.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_int64)]
        L.trexhip_profile_reset.argtypes = [C.c_void_p]
        L.trexhip_default_midline_params.argtypes = [C.POINTER(MidlineParams)]
        L.trexhip_default_midline_params.restype = None
        L.trexhip_midline_device.argtypes = [C.c_void_p, C.POINTER(MidlineParams), C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.trexhip_midline_movement_device.argtypes = [C.c_void_p, C.POINTER(MidlineParams), C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.trexhip_default_split_params.argtypes = [C.POINTER(SplitParams)]
        L.trexhip_default_split_params.restype = None
        L.trexhip_split_search_device.argtypes = [C.c_void_p, C.POINTER(SplitParams), C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
        L.trexhip_crops_posture_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_float, C.c_int32]
        L.trexhip_default_posture_params.argtypes = [C.POINTER(PostureParams)]
        L.trexhip_default_posture_params.restype = None
        L.trexhip_posture_device.argtypes = [C.c_void_p, C.c_int32, C.POINTER(PostureParams), C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
        L.trexhip_posture_auto_device.argtypes = [C.c_void_p, C.POINTER(PostureParams), C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.trexhip_crops_transformed_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_float, C.c_int32]
        L.trexhip_crops_device.argtypes = [C.c_void_p, C.c_void_p] + [C.c_int32] * 5
        L.trexhip_export_id_table_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_uint32, C.c_void_p, C.c_int32]
        L.trexhip_export_id_table_ex_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32]
        L.trexhip_load_weights.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.trexhip_num_classes.argtypes = [C.c_void_p]
        L.trexhip_network_channels.argtypes = [C.c_void_p]
        L.trexhip_pack_frames_v6_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        L.trexhip_lzo1x_bound.argtypes = [C.c_size_t]; L.trexhip_lzo1x_bound.restype = C.c_size_t
        L.trexhip_lzo1x_compress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
        L.trexhip_pv_write_frames.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_uint64, C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(C.c_size_t)]
        L.trexhip_comm_unique_id.argtypes = [C.c_void_p]
        L.trexhip_comm_create.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]
        L.trexhip_comm_destroy.argtypes = [C.c_void_p]
        L.trexhip_comm_destroy.restype = None
        L.trexhip_comm_rank.argtypes = [C.c_void_p]
        L.trexhip_comm_world.argtypes = [C.c_void_p]
        L.trexhip_comm_gather_device.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        L.trexhip_comm_gather_device_on.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        L.trexhip_comm_count_ranks.argtypes = [C.c_void_p, C.POINTER(C.c_int32)]
        L.trexhip_set_identity_precision.argtypes = [C.c_void_p, C.c_int32]
        L.trexhip_identify_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
        L.trexhip_identify.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
        L.trexhip_identify_guard_stats.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.trexhip_weight_blob_bytes.argtypes = [C.c_int32, C.c_int32]
        L.trexhip_weight_blob_bytes.restype = C.c_size_t
        L.trexhip_trainer_create.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(TrainParams), C.POINTER(C.c_void_p)]
        L.trexhip_trainer_destroy.argtypes = [C.c_void_p]
        L.trexhip_trainer_destroy.restype = None
        L.trexhip_trainer_set_lr.argtypes = [C.c_void_p, C.c_float]
        L.trexhip_trainer_steps.argtypes = [C.c_void_p]
        L.trexhip_trainer_steps.restype = C.c_int64
        L.trexhip_train_step_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int32)]
        L.trexhip_train_step.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int32)]
        L.trexhip_train_eval_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(C.c_float), C.POINTER(C.c_int32)]
        L.trexhip_train_eval.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(C.c_float), C.POINTER(C.c_int32)]
        L.trexhip_trainer_read.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_size_t]
        L.trexhip_trainer_export.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
        _LIB = L
    return _LIB


def _check(rc):
    if rc != 0:
        raise TrexHipError(rc, lib().trexhip_last_error().decode())


def default_params(width, height, **kw):
    p = Params()
    lib().trexhip_default_params(C.byref(p), width, height)
    ranges = kw.pop("size_ranges", None)
    for k, v in kw.items():
        if not hasattr(p, k):
            raise KeyError(k)
        setattr(p, k, v)
    if ranges is not None:
        p.n_ranges = len(ranges)
        for i, (a, b) in enumerate(ranges):
            p.ranges[2 * i], p.ranges[2 * i + 1] = a, b
    return p


def _from_addr(addr, count, dtype):
    if count == 0 or not addr:
        return np.zeros(0, dtype)
    buf = (C.c_uint8 * (count * dtype.itemsize)).from_address(addr)
    return np.frombuffer(buf, dtype=dtype, count=count)


class FrameResult:
    """Blobs of one frame: structured arrays with frame-relative run/pixel offsets."""
    __slots__ = ("info", "blobs", "runs", "pixels")

    def __init__(self, info, blobs, runs, pixels):
        self.info, self.blobs, self.runs, self.pixels = info, blobs, runs, pixels


HIP_STREAM_LEGACY = 1        # hipStreamLegacy: the explicit handle of the legacy default stream (hip_runtime_api.h)


class Segmenter:
    """Host-side handle mirroring TRex's BackgroundSubtraction (set_background / apply / fps / deinit)."""

    def __init__(self, params, stream="torch"):
        """stream: "torch" = enqueue on torch's current stream of the device when torch is loaded (callers hand torch tensors to
        the context, so torch's own fills / copies and the context's kernels must be ordered); None = the context's own
        non-blocking stream (the library default); or a hipStream_t value."""
        self.params = params
        self._h = C.c_void_p()
        _check(lib().trexhip_create(C.byref(params), C.byref(self._h)))
        if stream == "torch":
            import sys
            torch = sys.modules.get("torch")
            if torch is not None and torch.cuda.is_available():
                s = torch.cuda.current_stream(params.device).cuda_stream
                self.set_stream(s if s else HIP_STREAM_LEGACY)     # torch's default stream is the legacy null stream
        elif stream:
            self.set_stream(stream)

    def close(self):
        if self._h:
            lib().trexhip_destroy(self._h)
            self._h = C.c_void_p()

    deinit = close

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        return self._h

    def update_params(self, size_ranges=None, **kw):
        """the settings TRex re-reads on every apply(): threshold, threshold_maximum, inclusive, enable_difference, absolute_difference,
        image_invert, zero_is_background, cm_per_pixel, size_ranges -- effective from the next segment call (trexhip_update_params)"""
        lp = LiveParams()
        _check(lib().trexhip_get_live_params(self._h, C.byref(lp)))
        for k, v in kw.items():
            if k not in dict(LiveParams._fields_):
                raise KeyError(k)
            setattr(lp, k, v)
        if size_ranges is not None:
            if len(size_ranges) > 8:
                lp.n_ranges = len(size_ranges)       # the library refuses it
            else:
                lp.n_ranges = len(size_ranges)
                for i, (a, b) in enumerate(size_ranges):
                    lp.ranges[2 * i], lp.ranges[2 * i + 1] = a, b
        _check(lib().trexhip_update_params(self._h, C.byref(lp)))
        for k, _ in LiveParams._fields_:
            if k != "ranges":
                setattr(self.params, k, getattr(lp, k))
        for i in range(16):
            self.params.ranges[i] = lp.ranges[i]

    def set_stream(self, hip_stream_ptr):
        _check(lib().trexhip_set_stream(self._h, C.c_void_p(hip_stream_ptr)))

    def set_background(self, bg):
        """bg: numpy uint8 [H,W] (host) or a torch CUDA uint8 tensor (device)."""
        if isinstance(bg, np.ndarray):
            bg = np.ascontiguousarray(bg, np.uint8)
            assert bg.shape == (self.params.height, self.params.width)
            _check(lib().trexhip_set_background(self._h, bg.ctypes.data_as(C.c_void_p), bg.shape[1]))
        else:
            assert bg.is_cuda and bg.is_contiguous() and bg.numel() == self.params.height * self.params.width
            _check(lib().trexhip_set_background_device(self._h, C.c_void_p(bg.data_ptr())))

    def generate_average(self, d_frames_ptr, n, method=0):
        """Background = per-pixel mean (0) / max (1) / min (2) of n HBM-resident gray frames; returns it as numpy."""
        _check(lib().trexhip_generate_average_device(self._h, C.c_void_p(d_frames_ptr), n, method))
        out = np.empty((self.params.height, self.params.width), np.uint8)
        _check(lib().trexhip_get_background(self._h, out.ctypes.data_as(C.c_void_p), out.shape[1]))
        return out

    def get_background(self):
        """The context's gray background as numpy uint8 [H,W]."""
        out = np.empty((self.params.height, self.params.width), np.uint8)
        _check(lib().trexhip_get_background(self._h, out.ctypes.data_as(C.c_void_p), out.shape[1]))
        return out

    def segment_device(self, d_ptr, n):
        """Enqueue the detect stage for n HBM-resident gray frames at device address d_ptr."""
        _check(lib().trexhip_segment_device(self._h, C.c_void_p(d_ptr), n))

    def segment_host(self, frames):
        frames = [np.ascontiguousarray(f, np.uint8) for f in frames]
        ptrs = (C.c_void_p * len(frames))(*[f.ctypes.data for f in frames])
        stride = frames[0].shape[1] if frames else self.params.width
        _check(lib().trexhip_segment(self._h, ptrs, stride, len(frames)))

    def set_background_color(self, bgc, color_channel=-1):
        """bgc: numpy uint8 [H,W,3|4] BGR / BGRA background (Background(image, rgb8)); also installs its gray image."""
        bgc = np.ascontiguousarray(bgc, np.uint8)
        assert bgc.shape[:2] == (self.params.height, self.params.width)
        _check(lib().trexhip_set_background_color(self._h, bgc.ctypes.data_as(C.c_void_p), bgc.shape[1] * bgc.shape[2], bgc.shape[2], color_channel))

    def segment_color_host(self, frames, color_channel=-1):
        """frames: list of uint8 [H,W,3|4] BGR/BGRA host images (what TRex's TileImage holds)."""
        frames = [np.ascontiguousarray(f, np.uint8) for f in frames]
        ch = frames[0].shape[2]
        ptrs = (C.c_void_p * len(frames))(*[f.ctypes.data for f in frames])
        _check(lib().trexhip_segment_color(self._h, ptrs, frames[0].shape[1] * ch, len(frames), ch, color_channel))

    def segment_color_device(self, d_color_ptr, n, channels, color_channel=-1):
        """n contiguous colour frames [n,H,W,channels] already in HBM."""
        _check(lib().trexhip_segment_color_device(self._h, C.c_void_p(d_color_ptr), n, channels, color_channel))

    def synchronize(self):
        _check(lib().trexhip_synchronize(self._h))

    def fetch_raw(self, rethreshold=False):
        """trexhip_fetch without building per-frame views: the BatchResult struct (counts + pointers into the context's pinned tables)."""
        r = BatchResult()
        rc = (lib().trexhip_fetch_rethreshold if rethreshold else lib().trexhip_fetch)(self._h, C.byref(r))
        if rc != 0 and rc != -3:
            _check(rc)
        if rc == -3:
            self.last_capacity_error = lib().trexhip_last_error().decode()
        return r

    def fetch(self, copy=True, rethreshold=False):
        r = BatchResult()
        rc = (lib().trexhip_fetch_rethreshold if rethreshold else lib().trexhip_fetch)(self._h, C.byref(r))
        if rc != 0 and rc != -3:
            _check(rc)
        info = _from_addr(r.frames, r.n_frames, INFO_DTYPE)
        blobs = _from_addr(r.blobs, r.total_blobs, BLOB_DTYPE)
        runs = _from_addr(r.runs, r.total_runs, RUN_DTYPE)
        pc = int(r.pixel_channels) or 1                       # bytes per pixel: 3 for the rgb8 pixel encoding
        pixels = _from_addr(r.pixels, r.total_pixels * pc, np.dtype(np.uint8))
        out = []
        for i in range(r.n_frames):
            fi = info[i]
            if fi["flags"]:
                out.append(FrameResult(fi.copy(), np.zeros(0, BLOB_DTYPE), np.zeros(0, RUN_DTYPE), np.zeros(0, np.uint8)))
                continue
            b = blobs[fi["blob_begin"]:fi["blob_begin"] + fi["n_blobs"]]
            ru = runs[fi["run_begin"]:fi["run_begin"] + fi["n_runs"]]
            px = pixels[int(fi["pix_begin"]) * pc:(int(fi["pix_begin"]) + int(fi["n_pixels"])) * pc]
            if copy:
                b, ru, px = b.copy(), ru.copy(), px.copy()
            out.append(FrameResult(fi.copy(), b, ru, px))
        if rc == -3:
            self.last_capacity_error = lib().trexhip_last_error().decode()
        return out

    def posture_device(self, n_blobs, d_outline_ptr, d_segments_ptr, d_info_ptr, table=0, **kw):
        """posture::calculate_posture for every blob of the detect (0) or re-threshold (1) table; see include/trexhip.h."""
        pp = PostureParams()
        lib().trexhip_default_posture_params(C.byref(pp))
        for k, v in kw.items():
            setattr(pp, k, v)
        _check(lib().trexhip_posture_device(self._h, table, C.byref(pp), n_blobs, C.c_void_p(d_outline_ptr),
                                            C.c_void_p(d_segments_ptr), C.c_void_p(d_info_ptr)))
        return pp

    def posture_auto_device(self, n_blobs, d_outline_ptr, d_segments_ptr, d_info_ptr, method=0, track_posture_threshold=15, d_threshold_ptr=0, d_iterations_ptr=0, **kw):
        """posture::calculate_posture with its threshold retry loop for every detect blob; see include/trexhip.h."""
        pp = PostureParams()
        lib().trexhip_default_posture_params(C.byref(pp))
        for k, v in kw.items():
            setattr(pp, k, v)
        _check(lib().trexhip_posture_auto_device(self._h, C.byref(pp), method, track_posture_threshold, n_blobs, C.c_void_p(d_outline_ptr),
                                                 C.c_void_p(d_segments_ptr), C.c_void_p(d_info_ptr), C.c_void_p(d_threshold_ptr or 0), C.c_void_p(d_iterations_ptr or 0)))

    def pack_frames_v6_device(self, d_out_ptr, capacity, d_offsets_ptr, timestamps=None):
        """pv::Frame::serialize bodies (file version V_6 layout) of the last fetched batch; see include/trexhip.h."""
        ts = np.ascontiguousarray(timestamps, np.uint64) if timestamps is not None else None
        _check(lib().trexhip_pack_frames_v6_device(self._h, ts.ctypes.data if ts is not None else None, C.c_void_p(d_out_ptr), capacity, C.c_void_p(d_offsets_ptr)))

    def crops_device(self, d_crops_ptr, n_blobs, out_w=80, out_h=80, normalization=0, difference=0):
        """constraints::diff_image for every blob of the last batch -> uint8 [n_blobs][out_h][out_w] at d_crops_ptr."""
        _check(lib().trexhip_crops_device(self._h, C.c_void_p(d_crops_ptr), n_blobs, out_w, out_h, normalization, difference))

    def midline_device(self, n_blobs, max_points, d_posture_info_ptr, d_segments_ptr, d_midline_ptr, d_midline_info_ptr, d_movement_ptr=None, **kw):
        """Midline::post_process + normalize for every blob of a posture call; d_movement_ptr: [n_blobs][2] float MovementInformation::direction
        (trexhip_midline_movement_device); see include/trexhip.h."""
        mp = MidlineParams()
        lib().trexhip_default_midline_params(C.byref(mp))
        for k, v in kw.items():
            setattr(mp, k, v)
        if d_movement_ptr is None:
            _check(lib().trexhip_midline_device(self._h, C.byref(mp), n_blobs, max_points, C.c_void_p(d_posture_info_ptr), C.c_void_p(d_segments_ptr),
                                                C.c_void_p(d_midline_ptr), C.c_void_p(d_midline_info_ptr)))
        else:
            _check(lib().trexhip_midline_movement_device(self._h, C.byref(mp), n_blobs, max_points, C.c_void_p(d_posture_info_ptr), C.c_void_p(d_segments_ptr),
                                                         C.c_void_p(d_midline_ptr), C.c_void_p(d_midline_info_ptr), C.c_void_p(d_movement_ptr)))

    def split_search_device(self, d_presumed_ptr, n_blobs, d_thresholds_ptr, d_info_ptr, method=1, size_ranges=(), **kw):
        """SplitBlob's threshold search for the detect blobs with presumed_nr > 0; see include/trexhip.h."""
        sp = SplitParams()
        lib().trexhip_default_split_params(C.byref(sp))
        for k, v in kw.items():
            setattr(sp, k, v)
        sp.n_ranges = len(size_ranges)
        for i, (a, b) in enumerate(size_ranges):
            sp.size_ranges[2 * i], sp.size_ranges[2 * i + 1] = a, b
        _check(lib().trexhip_split_search_device(self._h, C.byref(sp), method, C.c_void_p(d_presumed_ptr), n_blobs, C.c_void_p(d_thresholds_ptr),
                                                 C.c_void_p(d_info_ptr)))

    def crops_posture_device(self, d_crops_ptr, n_blobs, d_midline_info_ptr, midline_lengths=None, out_w=80, out_h=80, legacy=False, scale=1.0, difference=0):
        ln = None if midline_lengths is None else np.ascontiguousarray(midline_lengths, np.float32)
        _check(lib().trexhip_crops_posture_device(self._h, C.c_void_p(d_crops_ptr), n_blobs, out_w, out_h, C.c_void_p(d_midline_info_ptr),
                                                  None if ln is None else ln.ctypes.data_as(C.c_void_p), 1 if legacy else 0, scale, difference))

    def crops_transformed_device(self, d_crops_ptr, transforms, midline_lengths, out_w=80, out_h=80, legacy=False, scale=1.0, difference=0):
        """posture / legacy normalisation with caller-supplied Midline::transform matrices (host float32 [n,6]) and lengths [n]."""
        tr = np.ascontiguousarray(transforms, np.float32)
        ln = np.ascontiguousarray(midline_lengths, np.float32)
        _check(lib().trexhip_crops_transformed_device(self._h, C.c_void_p(d_crops_ptr), len(tr), out_w, out_h, tr.ctypes.data_as(C.c_void_p),
                                                      ln.ctypes.data_as(C.c_void_p), 1 if legacy else 0, scale, difference))

    def export_id_table_ex(self, d_probs_ptr, n_blobs, classes, frame_base, d_table_ptr, max_rows, d_midline_ptr=0, d_midline_info_ptr=0, midline_resolution=0):
        """Full per-blob record (16 header words + classes floats + 3 * midline_resolution floats per row); see include/trexhip.h."""
        _check(lib().trexhip_export_id_table_ex_device(self._h, C.c_void_p(d_probs_ptr) if d_probs_ptr else None, n_blobs, classes, frame_base,
                                                       C.c_void_p(d_midline_ptr) if d_midline_ptr else None,
                                                       C.c_void_p(d_midline_info_ptr) if d_midline_info_ptr else None, midline_resolution,
                                                       C.c_void_p(d_table_ptr), max_rows))

    def export_id_table(self, d_probs_ptr, n_blobs, classes, frame_base, d_table_ptr, max_rows):
        """Fixed-size per-blob identity table (8 header words + classes floats per row) into caller memory."""
        _check(lib().trexhip_export_id_table_device(self._h, C.c_void_p(d_probs_ptr) if d_probs_ptr else None, n_blobs,
                                                    classes, frame_base, C.c_void_p(d_table_ptr), max_rows))

    # ---- identity network (mirrors VINetwork: load_weights / probabilities) ----
    def load_weights(self, blob: bytes):
        buf = (C.c_uint8 * len(blob)).from_buffer_copy(blob)
        _check(lib().trexhip_load_weights(self._h, buf, len(blob)))

    def set_identity_precision(self, mode):
        """TREXHIP_CNN_*: 0 = exact fp32 MFMA, 1 = bf16x6 split (fp32-equivalent), 2 = bf16x3 (experiments), 3 = fp16x3 split (default:
        fp32-class on the fp16 matrix cores, range-guarded)."""
        _check(lib().trexhip_set_identity_precision(self._h, mode))

    def num_classes(self):
        return lib().trexhip_num_classes(self._h)

    def probabilities(self, crops):
        """crops: uint8 ndarray (n,80,80,C) on the host -> float32 (n,classes) softmax rows."""
        crops = np.ascontiguousarray(crops, np.uint8)
        n = crops.shape[0]
        out = np.empty((n, self.num_classes()), np.float32)
        _check(lib().trexhip_identify(self._h, crops.ctypes.data_as(C.c_void_p), n, out.ctypes.data_as(C.c_void_p)))
        return out

    def guard_stats(self):
        """(crops re-run by the fp16 range guard in the last identify call, whole batch re-run?)"""
        a, b = C.c_uint32(0), C.c_uint32(0)
        _check(lib().trexhip_identify_guard_stats(self._h, C.byref(a), C.byref(b)))
        return int(a.value), bool(b.value)

    def identify_device(self, d_crops_ptr, n, d_probs_ptr, d_logits_ptr=None):
        _check(lib().trexhip_identify_device(self._h, C.c_void_p(d_crops_ptr), n, C.c_void_p(d_probs_ptr),
                                             C.c_void_p(d_logits_ptr) if d_logits_ptr else None))

    def rethreshold_per_blob(self, d_thresholds_ptr, method=0, size_ranges=(), threshold=0):
        """SplitBlob::apply_threshold building block: one threshold per detect blob (int32 device array, pooled order; <0 skips)."""
        rng = np.ascontiguousarray(np.array(size_ranges, np.float64).reshape(-1))
        _check(lib().trexhip_rethreshold_per_blob_device(self._h, threshold, C.c_void_p(d_thresholds_ptr), method,
                                                         rng.ctypes.data_as(C.c_void_p) if len(rng) else None, len(rng) // 2))

    def rethreshold(self, threshold, method=0, size_ranges=()):
        """Tracker::prefilter's threshold_blob for every blob of the last batch; fetch(rethreshold=True) reads it."""
        rng = np.ascontiguousarray(np.array(size_ranges, np.float64).reshape(-1))
        _check(lib().trexhip_rethreshold_device(self._h, threshold, method, rng.ctypes.data_as(C.c_void_p) if len(rng) else None,
                                                len(rng) // 2))

    def device_view(self):
        v = DeviceView()
        _check(lib().trexhip_device_view_get(self._h, C.byref(v)))
        return v

    def profile_enable(self, on=True):
        _check(lib().trexhip_profile_enable(self._h, 1 if on else 0))

    def profile_read(self, stage):
        ms, n = C.c_double(), C.c_int64()
        _check(lib().trexhip_profile_read(self._h, stage, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def profile_reset(self):
        _check(lib().trexhip_profile_reset(self._h))


class Trainer:
    """Training step of the identity network (include/trexhip.h: trexhip_trainer_*, trexhip_train_step_device)."""

    def __init__(self, seg, weight_blob, max_batch, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8, bn_momentum=0.1, dropout=0.05, seed=0, precision=0):
        self._h = C.c_void_p()
        p = TrainParams(max_batch=max_batch, lr=lr, beta1=beta1, beta2=beta2, eps=eps, bn_momentum=bn_momentum, dropout=dropout, seed=seed, precision=precision)
        buf = (C.c_char * len(weight_blob)).from_buffer_copy(weight_blob)
        _check(lib().trexhip_trainer_create(seg.handle, buf, len(weight_blob), C.byref(p), C.byref(self._h)))
        hdr = np.frombuffer(weight_blob[:32], np.int32)
        self.classes, self.channels = int(hdr[2]), int(hdr[5])

    def step_device(self, d_inputs_ptr, d_targets_ptr, n, d_keep_ptr=0, want_loss=True):
        """-> (mean loss, correct count) when want_loss (synchronises), else None"""
        loss, correct = C.c_float(), C.c_int32()
        _check(lib().trexhip_train_step_device(self._h, C.c_void_p(d_inputs_ptr), C.c_void_p(d_targets_ptr), n, C.c_void_p(d_keep_ptr or 0),
                                               C.byref(loss) if want_loss else None, C.byref(correct) if want_loss else None))
        return (loss.value, correct.value) if want_loss else None

    def _check_batch(self, inputs, targets):
        """the reference asserts image size, channel count and integer class labels per batch (visual_recognition_torch.py:1109-1110); the ABI
        takes raw pointers, so a wrong shape would be an out-of-bounds read, not an error"""
        x = np.asarray(inputs)
        if x.ndim != 4 or tuple(x.shape[1:]) != (80, 80, self.channels):
            raise ValueError(f"inputs must be (n, 80, 80, {self.channels}), got {tuple(x.shape)}")
        t = np.asarray(targets)
        if t.dtype.kind not in "iu":
            raise ValueError(f"targets must be integer class indices, got dtype {t.dtype}")
        if t.shape != (x.shape[0],):
            raise ValueError(f"targets must be ({x.shape[0]},), got {tuple(t.shape)}")
        return np.ascontiguousarray(x, np.float32), np.ascontiguousarray(t, np.int32)

    def step(self, inputs, targets, keep_masks=None):
        """host arrays: inputs float32 (n,80,80,C) in [0,255], targets int (n,), keep_masks uint8 (n*308,) or None -> (loss, correct)"""
        x, y = self._check_batch(inputs, targets)
        k = np.ascontiguousarray(keep_masks, np.uint8) if keep_masks is not None else None
        if k is not None and k.size != x.shape[0] * 308:
            raise ValueError(f"keep_masks holds {k.size} entries, a batch of {x.shape[0]} needs {x.shape[0]} x 308 (16 + 64 + 128 + 100 per sample)")
        loss, correct = C.c_float(), C.c_int32()
        _check(lib().trexhip_train_step(self._h, x.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p), x.shape[0],
                                        k.ctypes.data_as(C.c_void_p) if k is not None else None, C.byref(loss), C.byref(correct)))
        return loss.value, correct.value

    def evaluate(self, inputs, targets):
        """model.eval() forward of one validation batch (host arrays) -> (mean cross entropy, correct count); the trainer is unchanged"""
        x, y = self._check_batch(inputs, targets)
        loss, correct = C.c_float(), C.c_int32()
        _check(lib().trexhip_train_eval(self._h, x.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p), x.shape[0], C.byref(loss), C.byref(correct)))
        return loss.value, correct.value

    def set_lr(self, lr):
        _check(lib().trexhip_trainer_set_lr(self._h, lr))

    @property
    def steps(self):
        return int(lib().trexhip_trainer_steps(self._h))

    def read(self, tensor, kind, shape):
        """tensor: index in state_dict order (trex_amd.weights.TENSORS); kind 0 parameter, 1 gradient, 2 / 3 Adam moments; torch layout"""
        out = np.empty(int(np.prod(shape)), np.float32)
        _check(lib().trexhip_trainer_read(self._h, tensor, kind, out.ctypes.data_as(C.c_void_p), out.size))
        return out.reshape(shape)

    def export(self):
        need = int(lib().trexhip_weight_blob_bytes(self.classes, self.channels))
        buf = (C.c_char * need)()
        got = C.c_size_t()
        _check(lib().trexhip_trainer_export(self._h, buf, need, C.byref(got)))
        return bytes(buf)

    def close(self):
        if self._h:
            lib().trexhip_trainer_destroy(self._h)
            self._h = C.c_void_p()


class Comm:
    """The library's communicator (include/trexhip.h: trexhip_comm_*): gather of every rank's table to rank 0 over RCCL."""

    def __init__(self, seg, rank=0, world=1, unique_id=None):
        self._h = C.c_void_p()
        self.rank, self.world = rank, world
        buf = (C.c_char * 128).from_buffer_copy(unique_id) if unique_id is not None else None
        _check(lib().trexhip_comm_create(seg.handle, buf, rank, world, C.byref(self._h)))

    @staticmethod
    def unique_id():
        buf = (C.c_char * 128)()
        _check(lib().trexhip_comm_unique_id(buf))
        return bytes(buf)

    def gather_device(self, d_send_ptr, nbytes, d_recv_rank0_ptr, seg=None):
        """seg: enqueue on that context's stream instead of the communicator's own (contexts of one device sharing the communicator)"""
        if seg is None:
            _check(lib().trexhip_comm_gather_device(self._h, C.c_void_p(d_send_ptr), nbytes, C.c_void_p(d_recv_rank0_ptr or 0)))
        else:
            _check(lib().trexhip_comm_gather_device_on(self._h, seg.handle, C.c_void_p(d_send_ptr), nbytes, C.c_void_p(d_recv_rank0_ptr or 0)))

    def count_ranks(self):
        """all-reduce of 1 over the communicator: the number of ranks that took part"""
        n = C.c_int32()
        _check(lib().trexhip_comm_count_ranks(self._h, C.byref(n)))
        return n.value

    def close(self):
        if self._h:
            lib().trexhip_comm_destroy(self._h)
            self._h = C.c_void_p()


def lzo1x_compress(data):
    """this library's LZO1X encoder (host code: no GPU needed) -> compressed bytes; see include/trexhip.h"""
    src = np.ascontiguousarray(np.frombuffer(bytes(data), np.uint8))
    cap = int(lib().trexhip_lzo1x_bound(len(src)))
    out = np.empty(cap, np.uint8)
    n = C.c_size_t()
    _check(lib().trexhip_lzo1x_compress(src.ctypes.data_as(C.c_void_p) if len(src) else None, len(src), out.ctypes.data_as(C.c_void_p), cap, C.byref(n)))
    return out[:n.value].copy()


def pv_write_frames(bodies, offsets, always_compress=False, file_offset=0):
    """.pv data section of the frames trexhip_pack_frames_v6_device packed (host arrays): -> (bytes, index table [n] u64); see include/trexhip.h"""
    b = np.ascontiguousarray(bodies, np.uint8)
    o = np.ascontiguousarray(offsets, np.uint64)
    n = len(o) - 1
    out = np.empty(int(o[n]) + 16, np.uint8)
    idx = np.zeros(n, np.uint64)
    used = C.c_size_t()
    _check(lib().trexhip_pv_write_frames(b.ctypes.data_as(C.c_void_p), o.ctypes.data_as(C.c_void_p), n, 1 if always_compress else 0, file_offset,
                                         out.ctypes.data_as(C.c_void_p), len(out), idx.ctypes.data_as(C.c_void_p), C.byref(used)))
    return out[:used.value].copy(), idx
