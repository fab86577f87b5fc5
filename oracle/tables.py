"""Host-side (numpy) restatement of the per-blob identity tables that libtrexhip's k_id_table / k_id_table_ex kernels write.
TEST INFRASTRUCTURE ONLY: the checker of tests/test_dist_*.py, never imported by trex_amd/ (the product exports the tables on the
device: trexhip_export_id_table_device / _ex_device, include/trexhip.h; consumer: Tracker::predicted, tracking/Tracker.cpp:237-247)."""
import numpy as np

HDR = 8       # header words per row of the basic table
HDR_EX = 16   # header words of the full record


def table_from_blobs(frame_results, frame_base, probs, classes, max_rows):
    """Host-side (numpy) construction of the identity table with the device kernel's exact layout;
    used by the CPU tests and as the reference for the kernel."""
    t = np.zeros((max_rows, HDR + classes), np.uint32)
    row = 0
    for f, r in enumerate(frame_results):
        for b in r.blobs if hasattr(r, "blobs") else r:
            h = t[row]
            h[0] = frame_base + f
            h[1] = b["bid"]; h[2] = b["n_pixels"]
            h[3] = int(b["x0"]) | (int(b["y0"]) << 16)
            h[4] = int(b["x1"]) | (int(b["y1"]) << 16)
            h[5] = np.float32(np.float64(b["m10"]) / np.float64(b["n_pixels"])).view(np.uint32)
            h[6] = np.float32(np.float64(b["m01"]) / np.float64(b["n_pixels"])).view(np.uint32)
            h[7] = 1
            if probs is not None:
                h[HDR:] = np.ascontiguousarray(probs[row], np.float32).view(np.uint32)
            row += 1
    return t


def table_ex_from_blobs(frame_results, frame_base, probs, classes, max_rows, midline=None, midline_info=None, resolution=0):
    """Host-side construction of the full record (trexhip_export_id_table_ex_device's exact layout): 16 header words, `classes`
    probabilities, resolution x (x, y, height) of the normalised midline.  midline: float32 [n, resolution, 4]; midline_info: the
    structured array of trexhip_midline_device (status / len / angle / offx / offy)."""
    t = np.zeros((max_rows, HDR_EX + classes + 3 * resolution), np.uint32)
    f32 = lambda v: np.float32(v).view(np.uint32)
    row = 0
    for f, r in enumerate(frame_results):
        for b in r.blobs if hasattr(r, "blobs") else r:
            h = t[row]
            n = np.float64(b["n_pixels"])
            cx, cy = np.float64(b["m10"]) / n, np.float64(b["m01"]) / n
            h[0] = frame_base + f
            h[1] = b["bid"]; h[2] = b["n_pixels"]
            h[3] = int(b["x0"]) | (int(b["y0"]) << 16)
            h[4] = int(b["x1"]) | (int(b["y1"]) << 16)
            h[5], h[6], h[7] = f32(cx), f32(cy), 1
            h[8] = f32(np.float64(b["m20"]) / n - cx * cx)
            h[9] = f32(np.float64(b["m11"]) / n - cx * cy)
            h[10] = f32(np.float64(b["m02"]) / n - cy * cy)
            h[15] = 0xffffffff
            if midline_info is not None:
                m = midline_info[row]
                h[11], h[12], h[13], h[14] = f32(m["len"]), f32(m["angle"]), f32(m["offx"]), f32(m["offy"])
                h[15] = np.int32(m["status"]).view(np.uint32)
                if m["status"] == 0 and resolution:
                    h[HDR_EX + classes:] = np.ascontiguousarray(midline[row, :, :3], np.float32).reshape(-1).view(np.uint32)
            if probs is not None:
                h[HDR_EX:HDR_EX + classes] = np.ascontiguousarray(probs[row], np.float32).view(np.uint32)
            row += 1
    return t
