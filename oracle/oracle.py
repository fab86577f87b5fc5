"""ctypes loader for the CPU oracle (oracle/trex_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product package (trex_amd/) never imports this.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

RUN_DTYPE = np.dtype([("x0", "<u2"), ("x1", "<u2"), ("y", "<u2"), ("pad", "<u2")])
BLOB_DTYPE = np.dtype([
    ("run_begin", "<u4"), ("n_runs", "<u4"), ("pix_begin", "<u4"), ("n_pixels", "<u4"),
    ("x0", "<u2"), ("y0", "<u2"), ("x1", "<u2"), ("y1", "<u2"),
    ("bid", "<u4"), ("px_min_max", "<u4"), ("parent", "<u4"), ("flags", "<u4"),
    ("m10", "<u8"), ("m01", "<u8"), ("m20", "<u8"), ("m11", "<u8"), ("m02", "<u8"),
    ("sp", "<u8"), ("spx", "<u8"), ("spy", "<u8"),
])
assert BLOB_DTYPE.itemsize == 104 and RUN_DTYPE.itemsize == 8


class Params(C.Structure):
    _fields_ = [
        ("width", C.c_int32), ("height", C.c_int32),
        ("threshold", C.c_int32), ("threshold_maximum", C.c_int32),
        ("enable_difference", C.c_int32), ("absolute_difference", C.c_int32),
        ("image_invert", C.c_int32), ("inclusive", C.c_int32),
        ("zero_is_background", C.c_int32), ("connectivity", C.c_int32),
        ("dilation_size", C.c_int32), ("use_closing", C.c_int32), ("closing_size", C.c_int32),
        ("n_ranges", C.c_int32),
        ("cm_per_pixel", C.c_double),
        ("ranges", C.c_double * 16),
    ]


class PostureParams(C.Structure):
    """Settings read by posture::calculate_posture / Outline (core/default_config.cpp:888-898, defaults below)."""
    _fields_ = [("outline_resample", C.c_float), ("outline_smooth_samples", C.c_int32), ("outline_smooth_step", C.c_int32),
                ("outline_approximate", C.c_int32), ("outline_curvature_range_ratio", C.c_float),
                ("midline_walk_offset", C.c_float), ("max_points", C.c_int32)]


class PostureInfo(C.Structure):
    _fields_ = [("status", C.c_int32), ("n_outline", C.c_int32), ("n_segments", C.c_int32), ("tail_index", C.c_int32),
                ("head_index", C.c_int32), ("n_traced", C.c_int32), ("peak_best", C.c_float), ("peak_runner_up", C.c_float), ("peak_margin", C.c_float)]


def posture_params(outline_resample=1.0, outline_smooth_samples=4, outline_smooth_step=1, outline_approximate=3,
                   outline_curvature_range_ratio=0.03, midline_walk_offset=0.025, max_points=2048):
    return PostureParams(outline_resample, outline_smooth_samples, outline_smooth_step, outline_approximate,
                         outline_curvature_range_ratio, midline_walk_offset, max_points)


def make_params(width, height, threshold=15, threshold_maximum=255, enable_difference=1,
                absolute_difference=1, image_invert=0, inclusive=1, zero_is_background=1,
                connectivity=8, dilation_size=0, use_closing=0, closing_size=3,
                cm_per_pixel=1.0, size_ranges=()):
    """Defaults = the reference's defaults (SURVEY.md section 5 settings table)."""
    p = Params()
    p.width, p.height = width, height
    p.threshold, p.threshold_maximum = threshold, threshold_maximum
    p.enable_difference, p.absolute_difference = enable_difference, absolute_difference
    p.image_invert, p.inclusive = image_invert, inclusive
    p.zero_is_background, p.connectivity = zero_is_background, connectivity
    p.dilation_size, p.use_closing, p.closing_size = dilation_size, use_closing, closing_size
    p.cm_per_pixel = cm_per_pixel
    p.n_ranges = len(size_ranges)
    for i, (a, b) in enumerate(size_ranges):
        p.ranges[2 * i], p.ranges[2 * i + 1] = a, b
    return p


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    src = [os.path.join(_HERE, f) for f in ("trex_oracle.c", "trex_posture.c", "trex_split.c", "trex_pv.c", "trex_oracle.h")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in src if os.path.exists(s)):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(so):
            build()
        L = C.CDLL(so)
        u8p = C.POINTER(C.c_uint8)
        L.oracle_segment.restype = C.c_void_p
        L.oracle_segment.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(Params)]
        L.oracle_frame_counts.argtypes = [C.c_void_p] + [C.POINTER(C.c_int32)] * 3
        L.oracle_frame_copy.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.oracle_frame_free.argtypes = [C.c_void_p]
        L.oracle_generate_binary.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(Params)]
        L.oracle_segment_batch.restype = C.c_int64
        L.oracle_segment_batch.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.POINTER(Params), C.c_int32]
        L.oracle_line_without_grid.restype = C.c_int32
        L.oracle_line_without_grid.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32,
                                               C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                               C.POINTER(C.c_int32)]
        L.oracle_threshold_blob.restype = C.c_void_p
        L.oracle_threshold_blob.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32,
                                            C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32]
        L.oracle_rethreshold_frame.restype = C.c_void_p
        L.oracle_rethreshold_frame.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                               C.c_int32, C.c_void_p, C.c_int32, C.c_double, C.c_int32]
        L.oracle_posture_naive.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.POINTER(PostureParams), C.c_void_p, C.c_void_p, C.c_void_p]
        L.oracle_posture.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.POINTER(PostureParams), C.c_void_p, C.c_void_p,
                                     C.POINTER(PostureInfo)]
        L.oracle_outline_resample.argtypes = [C.c_void_p, C.c_int32, C.c_float, C.c_void_p, C.c_int32]
        L.oracle_trace_outline.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int32]
        L.oracle_normalize_transform.argtypes = [C.c_void_p, C.c_float, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_void_p]
        L.oracle_moments_transform.argtypes = [C.c_void_p, C.c_void_p]
        L.oracle_warp_affine_u8.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32]
        L.oracle_warp_affine_nearest_u8.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32]
        L.oracle_posture_auto.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                          C.POINTER(PostureParams), C.c_void_p, C.c_void_p, C.POINTER(PostureInfo), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        L.oracle_midline_walk.argtypes = [C.c_void_p, C.c_int32, C.c_float, C.c_void_p]
        L.oracle_midline_post_process.argtypes = [C.c_void_p, C.c_int32, C.c_float, C.c_int32, C.c_int32]
        L.oracle_midline_post_process_mv.argtypes = [C.c_void_p, C.c_int32, C.c_float, C.c_int32, C.c_int32, C.c_float, C.c_float, C.POINTER(C.c_int)]
        L.oracle_midline_normalize.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_float, C.c_void_p, C.c_void_p]
        L.oracle_midline_transform.argtypes = [C.c_float, C.c_float, C.c_float, C.c_int32, C.c_void_p]
        L.oracle_vec_to_r3g3b2.restype = C.c_uint8
        L.oracle_vec_to_r3g3b2.argtypes = [C.c_uint8] * 3
        L.oracle_r3g3b2_to_vec.argtypes = [C.c_uint8, C.c_void_p]
        L.oracle_bgr2gray.restype = C.c_uint8
        L.oracle_bgr2gray.argtypes = [C.c_uint8] * 3
        L.oracle_line_without_grid_enc.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                                   C.c_int32, C.c_void_p, C.c_void_p, C.POINTER(C.c_int32)]
        L.oracle_split_search.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                          C.POINTER(SplitParams), C.c_int32, C.POINTER(SplitInfo)]
        L.oracle_bid.restype = C.c_uint32
        L.oracle_bid.argtypes = [C.c_uint32] * 4
        del u8p
        _LIB = L
    return _LIB


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def _take_frame(h):
    L = lib()
    nb, nr, npx = C.c_int32(), C.c_int32(), C.c_int32()
    L.oracle_frame_counts(h, C.byref(nb), C.byref(nr), C.byref(npx))
    blobs = np.zeros(nb.value, BLOB_DTYPE)
    runs = np.zeros(nr.value, RUN_DTYPE)
    pixels = np.zeros(npx.value, np.uint8)
    L.oracle_frame_copy(h, _ptr(blobs), _ptr(runs), _ptr(pixels))
    L.oracle_frame_free(h)
    return blobs, runs, pixels


def segment(frame, bg, params):
    """One gray frame through the detect stage -> (blobs, runs, pixels) numpy arrays."""
    frame = np.ascontiguousarray(frame, np.uint8)
    bg = np.ascontiguousarray(bg, np.uint8)
    assert frame.shape == (params.height, params.width) == bg.shape
    h = lib().oracle_segment(_ptr(frame), _ptr(bg), C.byref(params))
    return _take_frame(h)


def generate_binary(frame, bg, params):
    frame = np.ascontiguousarray(frame, np.uint8)
    bg = np.ascontiguousarray(bg, np.uint8)
    out = np.empty_like(frame)
    lib().oracle_generate_binary(_ptr(frame), _ptr(bg), _ptr(out), C.byref(params))
    return out


def segment_batch(frames, bg, params, threads):
    frames = np.ascontiguousarray(frames, np.uint8)
    bg = np.ascontiguousarray(bg, np.uint8)
    return lib().oracle_segment_batch(_ptr(frames), frames.shape[0], _ptr(bg), C.byref(params), threads)


def line_without_grid(runs, pixels, bg, method, threshold):
    """method: 0 absolute, 1 sign, 2 none.  bg: 2-D uint8 image or None."""
    runs = np.ascontiguousarray(runs, RUN_DTYPE)
    pixels = np.ascontiguousarray(pixels, np.uint8)
    out_runs = np.zeros(max(len(pixels), 1), RUN_DTYPE)
    out_px = np.zeros(max(len(pixels), 1), np.uint8)
    n_px = C.c_int32()
    if bg is not None:
        bg = np.ascontiguousarray(bg, np.uint8)
        bgp, stride = _ptr(bg), bg.shape[1]
    else:
        bgp, stride = None, 0
    n = lib().oracle_line_without_grid(_ptr(runs), len(runs), _ptr(pixels), bgp, stride, method, threshold,
                                       _ptr(out_runs), _ptr(out_px), C.byref(n_px))
    return out_runs[:n].copy(), out_px[:n_px.value].copy()


def threshold_blob(runs, pixels, bg, method, threshold, connectivity=8):
    runs = np.ascontiguousarray(runs, RUN_DTYPE)
    pixels = np.ascontiguousarray(pixels, np.uint8)
    bg = np.ascontiguousarray(bg, np.uint8)
    h = lib().oracle_threshold_blob(_ptr(runs), len(runs), _ptr(pixels), _ptr(bg), bg.shape[1],
                                    bg.shape[1], bg.shape[0], method, threshold, connectivity)
    return _take_frame(h)


def bid(x0, x1, y, n):
    return int(lib().oracle_bid(x0, x1, y, n))


def crop_none(frame, bg, blob, runs, out_w=80, out_h=80, difference=0, invert=False):
    """constraints::diff_image with individual_image_normalization=none
    (Application/src/tracker/tracking/FilterCache.cpp:157-235 calculate_diff_image, :265-294):
    paint the blob into its bounding box, then centre-pad with zeros / centre-cut to the output size.
    difference: 0 grey values, 1 |bg-p|, 2 max(bg-p,0) (track_background_subtraction :171-175)."""
    bw = int(blob["x1"]) - int(blob["x0"]) + 1
    bh = int(blob["y1"]) - int(blob["y0"]) + 1
    img = np.zeros((bh, bw), np.uint8)
    rs = runs[blob["run_begin"]:blob["run_begin"] + blob["n_runs"]]
    for r in rs:
        y, x0, x1 = int(r["y"]), int(r["x0"]), int(r["x1"])
        p = frame[y, x0:x1 + 1].astype(np.int32)
        if invert:
            p = 255 - p
        if difference:
            b = bg[y, x0:x1 + 1].astype(np.int32)
            p = np.abs(b - p) if difference == 1 else np.maximum(b - p, 0)
        img[y - int(blob["y0"]), x0 - int(blob["x0"]):x1 - int(blob["x0"]) + 1] = p
    padded = img
    left = right = top = bottom = 0
    if padded.shape[1] < out_w:                       # :184-188
        left = out_w - padded.shape[1]; right = left // 2; left -= right
    if padded.shape[0] < out_h:                       # :190-194
        top = out_h - padded.shape[0]; bottom = top // 2; top -= bottom
    if left or right or top or bottom:
        padded = np.pad(padded, ((top, bottom), (left, right)))
    if padded.shape[1] > out_w or padded.shape[0] > out_h:   # :212-226
        left = padded.shape[1] - out_w; right = left // 2; left -= right
        top = padded.shape[0] - out_h; bottom = top // 2; top -= bottom
        padded = padded[top:padded.shape[0] - bottom, left:padded.shape[1] - right]
    assert padded.shape == (out_h, out_w)
    return padded


def bgr2gray(img):
    """cv::cvtColor(BGR2GRAY / BGRA2GRAY) on 8-bit data (BackgroundSubtraction.cpp:167,170), restated from
    OpenCV's published fixed-point path: (B*1868 + G*9617 + R*4899 + (1<<13)) >> 14.  [recalled, unpinned:
    no OpenCV in the build image; cmn::bgr2gray uses in Application/Tests/test_pixels.cpp:43,52]"""
    a = img.astype(np.uint32)
    return ((a[..., 0] * 1868 + a[..., 1] * 9617 + a[..., 2] * 4899 + 8192) >> 14).astype(np.uint8)


def rethreshold_frame(frame, bg, params, method, threshold, size_ranges=(), invert=False):
    """Detect stage followed by Tracker::prefilter's threshold_blob on every kept blob -> (blobs, runs, pixels)
    of the sub-blobs, with parent / flags filled (see trex_oracle.h)."""
    frame = np.ascontiguousarray(frame, np.uint8)
    bg = np.ascontiguousarray(bg, np.uint8)
    L = lib()
    h = L.oracle_segment(_ptr(frame), _ptr(bg), C.byref(params))
    rng = np.ascontiguousarray(np.array(size_ranges, np.float64).reshape(-1))
    h2 = L.oracle_rethreshold_frame(h, _ptr(frame), _ptr(bg), params.width, params.height, method, threshold,
                                    params.connectivity, _ptr(rng) if len(rng) else None, len(rng) // 2,
                                    params.cm_per_pixel, 1 if invert else 0)
    L.oracle_frame_free(h)
    return _take_frame(h2)


def posture(runs, origin, pp=None, naive=False):
    """posture::calculate_posture for one blob given its lines: -> (info dict, outline [n,2] f32, segments [m,4] f32
    (pos.x, pos.y, height, l_length)); coordinates relative to `origin` (the blob's bounds().pos()).
    naive=False: the EFT in the operation order the device reproduces bit for bit (refuses outline_approximate > 15 like the device);
    naive=True: the independent reading -- libm sinf / cosf per harmonic, sequential float sums, any order up to 15."""
    pp = pp or posture_params()
    runs = np.ascontiguousarray(runs, RUN_DTYPE)
    out = np.zeros((pp.max_points, 2), np.float32)
    seg = np.zeros((pp.max_points, 4), np.float32)
    info = PostureInfo()
    fn = lib().oracle_posture_naive if naive else lib().oracle_posture
    rc = fn(_ptr(runs), len(runs), int(origin[0]), int(origin[1]), C.byref(pp), _ptr(out), _ptr(seg), C.byref(info))
    if rc == 5:
        raise ValueError("oracle.posture: outline_approximate > 15 is refused by the mirrored EFT (as by trexhip_posture_device)")
    d = {k: getattr(info, k) for k, _ in PostureInfo._fields_}
    return d, out[:info.n_outline].copy(), seg[:info.n_segments].copy()


def outline_resample(points, distance, cap=100000):
    pts = np.ascontiguousarray(points, np.float32)
    out = np.zeros((cap, 2), np.float32)
    n = lib().oracle_outline_resample(_ptr(pts), len(pts), distance, _ptr(out), cap)
    return out[:n].copy()


def trace_outline(runs, origin=(0, 0), cap=100000):
    runs = np.ascontiguousarray(runs, RUN_DTYPE)
    out = np.zeros((cap, 2), np.float32)
    n = lib().oracle_trace_outline(_ptr(runs), len(runs), int(origin[0]), int(origin[1]), _ptr(out), cap)
    return out[:n].copy()


def generate_average(frames, method=0):
    """Background from sampled gray frames: mean (float32 accumulation in sample order, cv::Mat::convertTo rounding =
    half to even), max or min (averaging_method, grabber/misc/default_config.cpp:131; sampler body in commons).  The mean
    variant reproduces the reference's golden CSVs best among the candidates tried."""
    frames = np.asarray(frames)
    if method == 1:
        return frames.max(0).astype(np.uint8)
    if method == 2:
        return frames.min(0).astype(np.uint8)
    if method == 3:      # mode: most frequent value per pixel, the smallest wins a tie (commons' sampler is not in the tree: unpinned)
        n = frames.shape[0]
        flat = frames.reshape(n, -1)
        counts = np.zeros((256, flat.shape[1]), np.int32)
        for f in flat:
            counts[f, np.arange(flat.shape[1])] += 1
        return counts.argmax(0).astype(np.uint8).reshape(frames.shape[1:])
    acc = np.zeros(frames.shape[1:], np.float32)
    for f in frames:
        acc += f.astype(np.float32)
    return np.clip(np.rint(acc / np.float32(len(frames))), 0, 255).astype(np.uint8)


def crop_normalized(frame, bg, blob, runs, tr6=None, midline_length=0.0, legacy=False, out_w=80, out_h=80, scale=1.0,
                    difference=0, invert=False, nearest=False):
    """constraints::diff_image with individual_image_normalization = moments (tr6 None: FilterCache.cpp:276-288) or
    posture / legacy (tr6 = Midline::transform(...).toCV() supplied by the caller, :267-274): imageFromLines into the
    bounding box, normalize_image's transform, cv::warpAffine INTER_LINEAR (FilterCache.cpp:21-115)."""
    bw = int(blob["x1"]) - int(blob["x0"]) + 1
    bh = int(blob["y1"]) - int(blob["y0"]) + 1
    img = crop_none(frame, bg, blob, runs, out_w=bw, out_h=bh, difference=difference, invert=invert)
    L = lib()
    if tr6 is None:
        tr = np.zeros(6, np.float32)
        b1 = np.ascontiguousarray(np.array([blob], BLOB_DTYPE))
        L.oracle_moments_transform(_ptr(b1), _ptr(tr))
        midline_length = 0.0
    else:
        tr = np.ascontiguousarray(tr6, np.float32)
    M = np.zeros(6, np.float32)
    L.oracle_normalize_transform(_ptr(tr), midline_length, 1 if legacy else 0, out_w, out_h, scale, _ptr(M))
    out = np.zeros((out_h, out_w), np.uint8)
    img = np.ascontiguousarray(img)
    (L.oracle_warp_affine_nearest_u8 if nearest else L.oracle_warp_affine_u8)(_ptr(img), bw, bh, _ptr(M), _ptr(out), out_w, out_h)
    return out, M


def warp_affine(src, M6, out_w, out_h):
    """cv::warpAffine(src, M, (out_w,out_h), INTER_LINEAR, BORDER_CONSTANT 0) in OpenCV's 8-bit fixed point."""
    src = np.ascontiguousarray(src, np.uint8)
    M = np.ascontiguousarray(M6, np.float32)
    out = np.zeros((out_h, out_w), np.uint8)
    lib().oracle_warp_affine_u8(_ptr(src), src.shape[1], src.shape[0], _ptr(M), _ptr(out), out_w, out_h)
    return out


def normalize_transform(tr6, midline_length, legacy, out_w, out_h, scale):
    tr = np.ascontiguousarray(tr6, np.float32)
    M = np.zeros(6, np.float32)
    lib().oracle_normalize_transform(_ptr(tr), float(midline_length), 1 if legacy else 0, out_w, out_h, float(scale), _ptr(M))
    return M


MIDLINE_INFO_DTYPE = np.dtype([("status", "<i4"), ("n", "<i4"), ("len", "<f4"), ("angle", "<f4"), ("offx", "<f4"), ("offy", "<f4"),
                               ("reserved", "<i4", (2,))])


def midline_normalize(segments, resolution=25, stiff=0.15, invert=False, start_with_head=False, movement=None):
    """Midline::post_process followed by Midline::normalize() (Individual.cpp:1369-1372).  movement = MovementInformation::direction (x, y) or
    None (no movement information).  returns (info, processed raw segments [n,4], normalised segments [resolution,4]); info["reserved"][0] = 1
    when the midline was turned round because of the movement (`_inverted_because_previous`)."""
    s = np.ascontiguousarray(segments, np.float32).copy()
    info = np.zeros(1, MIDLINE_INFO_DTYPE)
    out = np.zeros((resolution, 4), np.float32)
    flipped = C.c_int(0)
    mv = (0.0, 0.0) if movement is None else (float(movement[0]), float(movement[1]))
    if lib().oracle_midline_post_process_mv(_ptr(s), len(s), stiff, 1 if invert else 0, 1 if start_with_head else 0, mv[0], mv[1], C.byref(flipped)) != 0:
        info["status"] = 1
        return info[0], s, out
    lib().oracle_midline_normalize(_ptr(s), len(s), resolution, stiff, _ptr(out), _ptr(info))
    info["reserved"][0, 0] = flipped.value
    return info[0], s, out


def midline_transform(angle, offx, offy, legacy=False):
    tr = np.zeros(6, np.float32)
    lib().oracle_midline_transform(float(angle), float(offx), float(offy), 1 if legacy else 0, _ptr(tr))
    return tr


def posture_auto(runs, pixels, bg, method=0, start_threshold=0, pp=None):
    """posture::calculate_posture with its threshold retry loop (Posture.cpp:305-399) for one blob given as runs + pixels
    (full-frame coordinates).  returns (info dict incl. threshold / iterations, outline [n,2], segments [m,4])."""
    runs = np.ascontiguousarray(runs, RUN_DTYPE)
    pixels = np.ascontiguousarray(pixels, np.uint8)
    bg = np.ascontiguousarray(bg, np.uint8)
    pp = pp or posture_params()
    out = np.zeros((pp.max_points, 2), np.float32)
    seg = np.zeros((pp.max_points, 4), np.float32)
    info = PostureInfo()
    thr, it = C.c_int32(), C.c_int32()
    lib().oracle_posture_auto(_ptr(runs), len(runs), _ptr(pixels), _ptr(bg), bg.shape[1], bg.shape[1], bg.shape[0], method, start_threshold,
                              C.byref(pp), _ptr(out), _ptr(seg), C.byref(info), C.byref(thr), C.byref(it))
    d = {k: getattr(info, k) for k, _ in PostureInfo._fields_}
    d["threshold"] = thr.value; d["iterations"] = it.value
    return d, out[:info.n_outline].copy(), seg[:info.n_segments].copy()


def midline_walk(outline, midline_walk_offset=0.025):
    """the two-pointer walk of Outline::calculate_midline alone on a given (rotated) outline -> segments [m,4]."""
    pts = np.ascontiguousarray(outline, np.float32)
    seg = np.zeros((max(len(pts), 1), 4), np.float32)
    ns = lib().oracle_midline_walk(_ptr(pts), len(pts), midline_walk_offset, _ptr(seg))
    return seg[:ns].copy()


ENC_GRAY, ENC_R3G3B2, ENC_RGB8 = 0, 1, 2       # order of cmn::meta_encoding_t


def vec_to_r3g3b2(c0, c1, c2):
    return int(lib().oracle_vec_to_r3g3b2(int(c0), int(c1), int(c2)))


def r3g3b2_to_vec(code):
    out = np.zeros(3, np.uint8)
    lib().oracle_r3g3b2_to_vec(int(code), _ptr(out))
    return out


def convert_to_r3g3b2(img):
    """convert_to_r3g3b2<3|4>: H x W x {3,4} (BGR[A] memory order) -> H x W codes."""
    img = np.asarray(img, np.uint8)
    return (((img[..., 0] >> 6) << 6) | ((img[..., 1] >> 5) << 3) | (img[..., 2] >> 5)).astype(np.uint8)


def line_without_grid_enc(runs, pixels, pixel_enc, bg, bg_enc, method, threshold):
    """line_without_grid for gray / r3g3b2 / rgb8 pixel arrays against a gray or rgb8 background image (H x W [x 3])."""
    runs = np.ascontiguousarray(runs, RUN_DTYPE)
    pixels = np.ascontiguousarray(pixels, np.uint8)
    pc = 3 if pixel_enc == ENC_RGB8 else 1
    out_runs = np.zeros(max(len(pixels), 1), RUN_DTYPE)
    out_px = np.zeros(max(len(pixels), 1), np.uint8)
    n_px = C.c_int32()
    bg = np.ascontiguousarray(bg, np.uint8)
    n = lib().oracle_line_without_grid_enc(_ptr(runs), len(runs), _ptr(pixels), pixel_enc, _ptr(bg), bg.shape[1], bg_enc, method, threshold,
                                           _ptr(out_runs), _ptr(out_px), C.byref(n_px))
    return out_runs[:n].copy(), out_px[:n_px.value * pc].copy()


class SplitParams(C.Structure):
    _fields_ = [("initial_threshold", C.c_int32), ("algorithm", C.c_int32), ("blob_split_max_shrink", C.c_float),
                ("blob_split_global_shrink_limit", C.c_float), ("cm_per_pixel", C.c_float), ("n_ranges", C.c_int32), ("ranges", C.c_double * 16)]


class SplitInfo(C.Structure):
    _fields_ = [("threshold", C.c_int32), ("effective_threshold", C.c_int32), ("initial_action", C.c_int32), ("n_result", C.c_int32),
                ("n_tried", C.c_int32), ("min_pixel", C.c_int32), ("max_pixel", C.c_int32), ("first_size", C.c_float), ("min_size_bound", C.c_double)]


SPLIT_ACTIONS = ("KEEP", "KEEP_ABORT", "REMOVE", "ABORT", "TOO_FEW", "SKIP", "NO_CHANCE")


def split_params(track_threshold=15, track_posture_threshold=15, calculate_posture=True, algorithm=1, max_shrink=0.2, global_shrink_limit=0.2,
                 cm_per_pixel=1.0, size_ranges=()):
    """SplitBlob settings (defaults: core/default_config.cpp:921-923); algorithm 1 = threshold, 2 = threshold_approximate"""
    p = SplitParams()
    p.initial_threshold = (max(track_threshold, track_posture_threshold) if calculate_posture else track_threshold) + 1
    p.algorithm = algorithm
    p.blob_split_max_shrink, p.blob_split_global_shrink_limit, p.cm_per_pixel = max_shrink, global_shrink_limit, cm_per_pixel
    p.n_ranges = len(size_ranges)
    for i, (a, b) in enumerate(size_ranges):
        p.ranges[2 * i], p.ranges[2 * i + 1] = a, b
    return p


def split_search(runs, pixels, bg, method, params, presumed_nr, connectivity=8):
    """SplitBlob::split's threshold search for ONE blob (full-frame runs + grey pixels).  Returns SplitInfo."""
    runs = np.ascontiguousarray(runs, RUN_DTYPE)
    pixels = np.ascontiguousarray(pixels, np.uint8)
    bg = np.ascontiguousarray(bg, np.uint8)
    out = SplitInfo()
    lib().oracle_split_search(_ptr(runs), len(runs), _ptr(pixels), _ptr(bg), bg.shape[1], bg.shape[1], bg.shape[0], method, connectivity,
                              C.byref(params), presumed_nr, C.byref(out))
    return out


def pv_serialize_v6(blobs, runs, pixels, timestamp=0):
    """pv::Frame::serialize in the on-disk layout of file version V_6 (oracle/trex_pv.c); blobs / runs / pixels of ONE frame."""
    blobs = np.ascontiguousarray(blobs, BLOB_DTYPE); runs = np.ascontiguousarray(runs, RUN_DTYPE); pixels = np.ascontiguousarray(pixels, np.uint8)
    f = lib().oracle_pv_serialize_v6
    f.restype = C.c_uint64
    f.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
    n = f(_ptr(blobs), len(blobs), _ptr(runs), _ptr(pixels), timestamp, None)
    out = np.zeros(n, np.uint8)
    f(_ptr(blobs), len(blobs), _ptr(runs), _ptr(pixels), timestamp, _ptr(out))
    return out


def lzo1x_decompress(data, out_len):
    """the oracle's restatement of the LZO1X decoder (oracle/trex_pv.c) -> bytes, or None for a malformed stream"""
    src = np.ascontiguousarray(np.frombuffer(bytes(data), np.uint8))
    dst = np.zeros(max(1, out_len), np.uint8)
    f = lib().oracle_lzo1x_decompress
    f.restype = C.c_int64
    f.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64]
    n = f(_ptr(src), len(src), _ptr(dst), out_len)
    return None if n < 0 else dst[:n].copy()


def pv_read_v6(buf):
    """pv::Frame::read_from for version V_6: (bytes consumed, timestamp, runs with y, pixels, runs per object, pixels per object)."""
    buf = np.ascontiguousarray(buf, np.uint8)
    runs = np.zeros(max(1, len(buf) // 4), RUN_DTYPE); px = np.zeros(max(1, len(buf)), np.uint8)
    br = np.zeros(65536, np.uint32); bp = np.zeros(65536, np.uint32)
    ts = C.c_uint64(); n = C.c_int32()
    f = lib().oracle_pv_read_v6
    f.restype = C.c_uint64
    f.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_int32), C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int32]
    used = f(_ptr(buf), len(buf), C.byref(ts), C.byref(n), _ptr(runs), len(runs), _ptr(px), len(px), _ptr(br), _ptr(bp), 65535)
    nb = n.value
    return int(used), int(ts.value), runs[:int(br[:nb].sum())].copy(), px[:int(bp[:nb].sum())].copy(), br[:nb].copy(), bp[:nb].copy()


def history_split(n_blobs, n_fish, blob_mappings, paired, streak=None, split_threshold=-1, manual=(), history_split_on=True):
    """HistorySplit's per-frame decision (tracking/HistorySplit.cpp:52-312, restated in trex_split.c).
    blob_mappings: {blob: iterable of individuals}, paired: {individual: [(blob, distance), ...]} (walked in the order given).
    Returns (number[n_blobs], allow_less_than[n_blobs], big[n_blobs], centers: per blob the individuals whose last positions are appended)."""
    L = lib()
    L.oracle_history_split.restype = C.c_int32
    L.oracle_history_split.argtypes = [C.c_int32, C.c_int32] + [C.c_void_p] * 6 + [C.c_int32, C.c_void_p, C.c_int32, C.c_int32] + [C.c_void_p] * 5
    map_off = np.zeros(n_blobs + 1, np.int32); map_fish = []
    for b in range(n_blobs):
        map_off[b] = len(map_fish); map_fish += sorted(set(blob_mappings.get(b, ())))
    map_off[n_blobs] = len(map_fish)
    pair_off = np.zeros(n_fish + 1, np.int32); pb = []; pd = []
    for f in range(n_fish):
        pair_off[f] = len(pb)
        for b, d in paired.get(f, ()):
            pb.append(b); pd.append(d)
    pair_off[n_fish] = len(pb)
    map_fish = np.asarray(map_fish + [0], np.int32); pb = np.asarray(pb + [0], np.int32); pd = np.asarray(pd + [0], np.float32)
    st = np.asarray(streak if streak is not None else [1] * n_fish, np.int32)
    mn = np.asarray(list(manual) + [0], np.int32)
    number = np.zeros(n_blobs, np.int32); allow = np.zeros(n_blobs, np.uint8); big = np.zeros(n_blobs, np.uint8)
    coff = np.zeros(n_blobs + 1, np.int32); cfish = np.zeros(2 * n_fish + 2, np.int32)
    L.oracle_history_split(n_blobs, n_fish, _ptr(map_off), _ptr(map_fish), _ptr(pair_off), _ptr(pb), _ptr(pd), _ptr(st), int(split_threshold),
                           _ptr(mn), len(manual), 1 if history_split_on else 0, _ptr(number), _ptr(allow), _ptr(big), _ptr(coff), _ptr(cfish))
    centers = [list(cfish[coff[b]:coff[b + 1]]) for b in range(n_blobs)]
    return number, allow, big, centers
