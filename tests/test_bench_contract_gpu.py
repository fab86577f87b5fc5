"""bench.py prints ONE short JSON line with the keys the driver reads; the line is the last thing on stdout (long records come before it, prefixed)."""
import json
import os
import subprocess
import sys
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(*extra):
    if "--secondary-only" not in extra:
        extra = (*extra, "--no-secondary")          # (the secondary runs use their own, full-size shapes: only the test of them pays for that)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--batch", "16", *extra],
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.strip().splitlines() if l.strip()]
    # the driver keeps the last ~12 KB of stdout: the final line has to fit with room to spare (round 3's 22.8 KB line was cut -> unparsed)
    assert len(lines[-1]) < 4096, len(lines[-1])
    run.detail = {l.split(":", 1)[0][2:]: json.loads(l.split(":", 1)[1]) for l in lines[:-1] if l.startswith("# bench_")}
    return json.loads(lines[-1])


def test_default_line_has_the_contract_keys():
    j = run("--cpu-seconds", "2")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline"):
        assert k in j, k
    assert j["n_gpus"] == 1 and j["steps"] == 2 and j["warmup"] == 1 and j["higher_is_better"] is True and j["scaling"] == "weak"
    assert j["unit"] == "frames/s" and j["value"] > 0 and j["vs_baseline"] is None and j["data"] == "synthetic"
    assert "workload" in j["config"] and "model" not in j["config"]
    r = j["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    c = j["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("port", "reference") and c["value"] > 0 and c["cores"] >= 1
    assert abs(j["value"] - 16 * 2 / (j["ms_per_step"] * 2e-3)) / j["value"] < 1e-4      # value = frames of the timed steps / their time


def test_detect_only_line_reports_an_hbm_roofline():
    j = run("--no-cpu-baseline", "--stages", "segment")
    assert j["roofline"]["bound"] == "hbm" and j["roofline"]["unit"] == "GB/s" and j["roofline"]["peak"] == 8000.0
    assert "cpu_baseline" not in j or j["cpu_baseline"] is None or isinstance(j["cpu_baseline"], dict)


def test_two_rank_launch_on_one_gpu_runs_the_gather_path():
    # the driver's N > 1 launch line with every rank on GPU 0 (--same-gpu: RCCL over loopback sockets) -- functional check of the
    # communicator creation, the per-step gather to rank 0 and the max-over-ranks timing; the value itself means nothing
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", "29541", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "16",
                          "--same-gpu", "--no-cpu-baseline"], capture_output=True, text=True, timeout=420, cwd=ROOT)
    if out.returncode != 0 and ("Duplicate GPU" in out.stderr or "No socket interfaces" in out.stderr):
        pytest.skip("RCCL cannot run two ranks on one GPU here")
    assert out.returncode == 0, out.stderr[-3000:]
    j = json.loads([l for l in out.stdout.strip().splitlines() if l.startswith("{")][-1])
    assert j["n_gpus"] == 2 and j["scaling"] == "weak" and j["config"]["gather"].startswith("libtrexhip")
    assert len(out.stdout.strip().splitlines()[-1]) < 4096
    assert j["dist"]["ranks_seen"] == 2 and j["dist"]["gather_bytes_per_step"] > 0 and j["dist"]["ms_per_step_min"] <= j["dist"]["ms_per_step_max"]
    assert abs(j["value"] - 2 * 16 * 2 / (j["ms_per_step"] * 2e-3)) / j["value"] < 1e-4      # whole-job frames / slowest rank's time


def test_gpus_flag_spawns_its_own_ranks():
    # `python bench.py --gpus 2` without a launcher must not report an N = 1 number under n_gpus = 2: it starts the two ranks itself
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "16", "--same-gpu",
                          "--no-cpu-baseline"], capture_output=True, text=True, timeout=420, cwd=ROOT,
                         env={k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")})
    if out.returncode != 0 and ("Duplicate GPU" in out.stderr or "No socket interfaces" in out.stderr):
        pytest.skip("RCCL cannot run two ranks on one GPU here")
    assert out.returncode == 0, out.stderr[-3000:]
    j = json.loads([l for l in out.stdout.strip().splitlines() if l.startswith("{")][-1])
    assert j["n_gpus"] == 2 and abs(j["value"] - 2 * 16 * 2 / (j["ms_per_step"] * 2e-3)) / j["value"] < 1e-4


def test_a_launcher_that_started_the_wrong_number_of_ranks_is_an_error():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--batch", "4", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert out.returncode != 0 and "--gpus 2" in out.stderr


def test_final_line_is_short_and_the_long_records_come_before_it():
    # the other configurations / input paths / the N > 1 code path at N = 1 / the training step: one compact entry each in `configs` of the
    # final line; the full records (with their own rooflines) on a "# bench_secondary:" line before it and in gpurun_out/
    j = run("--no-cpu-baseline", "--secondary-only", "C2,C4_force_dist,train_step")
    s = j["configs"]
    assert set(s) == {"C2", "C4_force_dist", "train_step"}
    for name, e in s.items():
        assert "error" not in e, (name, e)
        assert e["value"] > 0 and 0 < e["frac"] < 1.5, (name, e)
    assert s["C4_force_dist"]["ranks_seen"] == 1
    assert 0 < s["train_step"]["frac_split_peak"] < 1
    full = run.detail["bench_secondary"]
    assert set(full) == set(s) and full["C2"]["roofline"]["bound"] == "hbm" and full["train_step"]["roofline"]["peak"] == 157.3
    assert json.load(open(os.path.join(ROOT, "gpurun_out", "bench_secondary.json"))).keys() == full.keys()
    # the N > 1 code path with one rank (library communicator, gather) is the same pipeline as the default run of the same shape
    assert "roofline_kernels" in j and set(j["roofline_kernels"]) == {"conv2", "conv3"}
    assert j["roofline"]["kernel"] in (j["roofline_kernels"]["conv2"]["kernel"], j["roofline_kernels"]["conv3"]["kernel"])
    d = run.detail["bench_detail"]
    assert d["value"] == pytest.approx(j["value"], rel=1e-4) and "traffic_note" in d["roofline"]


def test_full_default_line_fits_the_drivers_tail():
    # the driver's own command shape (all secondary runs, cpu baseline) at a small batch: the final line stays under 4 KB
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--cpu-seconds", "2"],
                         capture_output=True, text=True, timeout=1500, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    last = out.stdout.strip().splitlines()[-1]
    assert len(last) < 4096, len(last)
    j = json.loads(last)
    assert "roofline" in j and "cpu_baseline" in j and len(j["configs"]) >= 12
    assert len(out.stdout[-12000:].strip().splitlines()[-1]) == len(last)


@pytest.mark.parametrize("n", [2, 4, 8])
def test_strong_scaling_path_on_one_gpu(n):
    # the N > 1 strong-scaling launch (one camera's batch split between the ranks) with every rank on GPU 0: one shared communicator per
    # process, gather of the lanes' tables to rank 0 on the lanes' streams, rank 0's slab copy on its own stream.  Functional only.
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "3", "--warmup", "1", "--batch", "16", "--scaling", "strong",
                          "--same-gpu", "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, cwd=ROOT,
                         env={k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")})
    if out.returncode != 0 and ("Duplicate GPU" in out.stderr or "No socket interfaces" in out.stderr):
        pytest.skip("RCCL cannot run several ranks on one GPU here")
    assert out.returncode == 0, out.stderr[-3000:]
    j = json.loads([l for l in out.stdout.strip().splitlines() if l.startswith("{")][-1])
    assert j["n_gpus"] == n and j["scaling"] == "strong" and j["config"]["frames_per_step_per_gpu"] == 16 // n
    assert j["config"]["gather"].startswith("libtrexhip") and j["dist"]["ranks_seen"] == n
    assert abs(j["value"] - 16 * 3 / (j["ms_per_step"] * 3e-3)) / j["value"] < 1e-4        # whole-job frames (the batch is split, not multiplied)


def test_eight_rank_weak_launch_on_one_gpu_covers_every_frame_once():
    # the driver's N = 8 line (weak scaling: B frames per rank and step) with every rank on GPU 0: the gather moves 7 tables to rank 0 per step,
    # rank 0's merged table holds the step's 8 x B frames exactly once in Tracker::add order, and the N > 1 line is short and complete
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "3", "--warmup", "1", "--batch", "4", "--same-gpu",
                          "--no-cpu-baseline"], capture_output=True, text=True, timeout=1200, cwd=ROOT,
                         env={k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")})
    if out.returncode != 0 and ("Duplicate GPU" in out.stderr or "No socket interfaces" in out.stderr):
        pytest.skip("RCCL cannot run several ranks on one GPU here")
    assert out.returncode == 0, out.stderr[-3000:]
    last = [l for l in out.stdout.strip().splitlines() if l.startswith("{")][-1]
    assert len(last) < 4096, len(last)
    j = json.loads(last)
    assert j["n_gpus"] == 8 and j["scaling"] == "weak" and j["config"]["frames_per_step_per_gpu"] == 4
    d = j["dist"]
    assert d["ranks_seen"] == 8 and d["gather_bytes_per_step"] == 7 * d["table_bytes_per_rank"]
    m = d["merged"]
    assert m["frames"] == m["frames_expected"] == 8 * 4 and m["every_frame_once"] and m["tracker_order"] and m["rows"] == 8 * 4 * 100
    assert "roofline" in j and j["roofline"]["bound"] == "mfma"
    assert abs(j["value"] - 8 * 4 * 3 / (j["ms_per_step"] * 3e-3)) / j["value"] < 1e-4      # whole-job frames / the slowest rank's time
