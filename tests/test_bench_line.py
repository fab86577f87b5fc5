"""bench.py's final line stays below 4 KB whatever the secondary runs returned (the driver keeps only the tail of stdout; round 3's 22.8 KB
line was cut and nothing was parsed).  CPU-only: the line builders are plain functions."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def _full(n_cfg=14, err=False):
    out = {"metric": "frames/s end-to-end (segment+CNN-ID), 2048x2048 x100 individuals", "value": 26812.2, "unit": "frames/s", "n_gpus": 1, "steps": 20, "warmup": 5,
           "ms_per_step": 9.5479, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "fp16x3-split (long explanation " + "x" * 300 + ")",
           "data": "synthetic",
           "config": {"workload": "C4: 2048x2048 gray, 100 individuals/frame, 256 frames resident per step per GPU, 80x80x1 crops, 100-way V118_3 (random-init weights)",
                      "stages": "s" * 400, "frames_per_step_per_gpu": 256, "encoding": "gray", "input": "gray frames resident in HBM", "posture": False,
                      "individual_image_normalization": "none", "pipelined_lanes": 2, "blobs_per_step_rank0": 25600, "parallelism": "frame-sharded x1"},
           "roofline": {"kernel": "k_conv5_wpre (long)", "bound": "mfma", "achieved": 898.61234, "peak": 2500.0, "unit": "TFLOP/s", "frac": 0.3594, "traffic": 1.1e10,
                        "avg_launch_us": 4668.0, "launches": 20, "mfma_issue_frac": 0.43, "peak_note": "n" * 600, "traffic_note": "t" * 300},
           "roofline_kernels": {k: {"kernel": "k_x (y)", "achieved": 1.0, "frac": 0.2, "traffic": 1.0, "avg_launch_us": 2.0} for k in ("conv2", "conv3")},
           "roofline_detect": {"kernel": "k_rows32b (..)", "achieved": 5136.0, "peak": 8000.0, "frac": 0.642, "traffic": 1.1e9, "avg_launch_us": 218.2, "limiter": "l" * 900,
                               "whole_detect_pass_us": 298.5, "whole_detect_pass_frac": 0.53, "pipelined_detect_pass_us": 249.4, "pipelined_detect_pass_frac": 0.638},
           "stage_us": {"detect": 298.5, "posture": None, "crops": 105.4, "conv2": 4203.0, "conv3": 4668.0, "cnn_all": 9475.0},
           "cpu_baseline": {"value": 11.55, "unit": "frames/s", "cores": 16, "kind": "port", "value_1_thread": 7.1, "detect_frames_per_s": 2541.0,
                            "identify_frames_per_s": 11.6, "sample": "s" * 700}}
    sec = {}
    for i in range(n_cfg):
        sec["C4_some_long_configuration_name_%d" % i] = ({"error": "RuntimeError: " + "e" * 500} if err else
                                                          {"value": 1234.5678, "ms_per_step": 1.23456, "roofline": {"frac": 0.345678, "whole_detect_pass_us": 299.9, "whole_detect_pass_frac": 0.53}})
    return out, sec


def test_the_usual_line_keeps_everything_and_is_short():
    out, sec = _full()
    line = bench.fit_line(bench.compact_line(out, sec))
    j = json.loads(line)
    assert len(line) < 4096 and "dropped" not in j
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "configs"):
        assert k in j, k
    assert abs(j["roofline"]["frac"] - j["roofline"]["achieved"] / j["roofline"]["peak"]) < 1e-12
    assert j["dtype"] == "fp16x3-split" and "model" not in j["config"]


def test_a_line_that_would_be_too_long_sheds_optional_parts_never_the_contract():
    out, sec = _full(n_cfg=60, err=True)
    line = bench.fit_line(bench.compact_line(out, sec))
    j = json.loads(line)
    assert len(line) < 4096
    for k in ("metric", "value", "ms_per_step", "config", "roofline", "cpu_baseline"):
        assert k in j, k
    assert "dropped" in j or all(set(e) <= {"value", "error"} for e in j["configs"].values())
