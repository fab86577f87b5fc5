#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X TRex hot path (contract: see the task prompt).

One "step" = one pass of the hot path over one batch of synthetic frames already resident in HBM:
  detect   background subtraction + threshold + run extraction + CCL + size filter + blob gather,
           blob/run/pixel tables copied to pinned host memory (what TRex's pv::Frame needs)
  crops    80x80 crop per blob (individual_image_normalization = none)
  identify V118_3 forward + softmax for every crop, probabilities copied to the host
Prints ONE JSON line on rank 0.

    python bench.py --gpus 1 --steps 10 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_CROP_CONV3 = 2.0 * 400 * 128 * 1600      # 20x20 px, 128 out channels, K = 25*64
FLOP_PER_CROP_CONV2 = 2.0 * 1600 * 64 * 400       # 40x40 px, 64 out channels, K = 25*16
FLOP_PER_CROP_CONV1 = 2.0 * 6400 * 16 * 25        # 80x80 px, 16 out channels, K = 25 (one input channel)
FLOP_PER_CROP_TOTAL = 2.0 * 126.73e6              # 126.73 M MAC per crop at 100 classes


def usable_cores():
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = max(1, min(n, int(int(q) / int(p))))
    except Exception:
        pass
    return n


def build_parser():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="C4")
    ap.add_argument("--batch", type=int, default=0, help="frames resident in HBM per step (default 256, C5: 64): large batches amortise the latency-bound labelling / gather stages (one workgroup per frame) over the HBM-bound pixel pass")
    ap.add_argument("--stages", default="all", choices=["all", "segment"])
    ap.add_argument("--cnn-mode", default="fp16x3", choices=["fp32", "bf16x6", "bf16x3", "fp16x3"],
                    help="arithmetic of conv2/conv3: fp16x3 = 2-piece fp16 split, fp32-class error, range-guarded (default); bf16x6 = 3-piece bf16 split; fp32 = exact fp32 MFMA")
    ap.add_argument("--with-posture", action="store_true", help="also run posture (outline -> midline) for every blob inside the timed step (configs C3/C5)")
    ap.add_argument("--normalize", default="none", choices=["none", "moments", "posture"],
                    help="individual_image_normalization of the identity crops; posture = outline -> midline -> Midline::transform -> warpAffine (implies --with-posture)")
    ap.add_argument("--encoding", default="gray", choices=["gray", "rgb8"],
                    help="meta_encoding: gray = the BASELINE workload (gray frames); rgb8 = BGRA frames in HBM -> cvtColor on the device, 3-byte pixel arrays, 3-channel crops and network")
    ap.add_argument("--input", default="gray", choices=["gray", "bgra", "host-gray", "host-bgra"],
                    help="bgra: the tiles as TRex hands them over (BGRA, BackgroundSubtraction.cpp:162-180), resident in HBM: cv::cvtColor runs on the device inside the timed step; host-bgra / host-gray: the same tiles in PAGEABLE HOST memory, through trexhip_segment_color / trexhip_segment -- the PCIe-inclusive as-deployed row of SURVEY 8d (never the headline value)")
    ap.add_argument("--no-pipeline", dest="pipeline", action="store_false", help="one context, strictly serial steps (no overlap of detect(i+1) with identity(i))")
    ap.add_argument("--lanes", type=int, default=2, help="software-pipeline depth: contexts whose batches are in flight (detect of batch i+L-1 is issued while the identity network works on batch i)")
    ap.add_argument("--no-detect-priority", dest="detect_priority", action="store_false",
                    help="keep the detect stage on the lane's own stream (default: a high-priority stream per lane, so that its short kernels get compute units as soon as the other lane's identity network frees some)")
    ap.add_argument("--gather", choices=["library", "torch"], default="library",
                    help="N > 1: who owns the RCCL communicator of the per-step table gather to rank 0: libtrexhip (trexhip_comm_*, default) or torch.distributed")
    ap.add_argument("--same-gpu", action="store_true", help="dev: run all ranks of a torchrun launch on GPU 0 (RCCL over loopback sockets): checks the N > 1 code path on a 1-GPU box, not a measurement")
    ap.add_argument("--force-dist", action="store_true", help="run the multi-GPU code path (the library's communicator, trexhip_comm_*) even with a single rank")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: --batch frames per GPU per step (default); strong: --batch frames per step in total, split between the GPUs")
    ap.add_argument("--force-all", action="store_true", help="run exactly the stages given on the command line instead of the preset of the named config")
    ap.add_argument("--guard-trip", action="store_true", help="secondary row C4_guard_tripped: conv1 of the random-init network is scaled until a handful of the step's crops leave the fp16-piece range, so that the per-crop range guard re-runs them with the bf16x6 kernels inside every timed step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=16.0)
    ap.add_argument("--no-secondary", action="store_true", help="skip the short secondary runs (C2, C3, C5, host input, fp32, the N > 1 code path at N = 1, training step); their long records go to a '# bench_secondary:' line and gpurun_out/bench_secondary.json, one {value, frac} pair each to `configs` of the final line")
    ap.add_argument("--secondary-only", default="", help="dev: comma-separated subset of the secondary runs")
    return ap


def _cpulist(text):
    cpus = set()
    for part in text.strip().split(","):
        if "-" in part:
            a, b = part.split("-"); cpus.update(range(int(a), int(b) + 1))
        elif part:
            cpus.add(int(part))
    return cpus


def pin_to_gpu_node(local):
    """pins this process to the CPUs of the NUMA node GPU `local` hangs on (sysfs); falls back to the node of the CPU it is running on"""
    import ctypes
    import glob
    node = -1
    cands = sorted(glob.glob("/sys/bus/pci/devices/*/numa_node"))
    gpu_nodes = []
    for f in cands:
        d = os.path.dirname(f)
        try:
            if open(os.path.join(d, "vendor")).read().strip() != "0x1002" or not open(os.path.join(d, "class")).read().startswith("0x03") \
                    and not open(os.path.join(d, "class")).read().startswith("0x12"):
                continue
            gpu_nodes.append((os.path.basename(d), int(open(f).read().strip())))
        except OSError:
            continue
    if gpu_nodes:
        node = gpu_nodes[min(local, len(gpu_nodes) - 1)][1]
    src = "the GPU's node"
    if node < 0:
        cpu = ctypes.CDLL(None).sched_getcpu()
        for f in glob.glob("/sys/devices/system/node/node*/cpulist"):
            if cpu in _cpulist(open(f).read()):
                node = int(os.path.basename(os.path.dirname(f))[4:])
        src = "the node this process was running on"
    if node < 0:
        return "not pinned (no NUMA information)"
    cpus = _cpulist(open(f"/sys/devices/system/node/node{node}/cpulist").read()) & os.sched_getaffinity(0)
    if not cpus:
        return "not pinned (node %d has no allowed CPU)" % node
    os.sched_setaffinity(0, cpus)
    return "process pinned to NUMA node %d (%s, %d CPUs) while it allocates the tiles and drives the pipeline" % (node, src, len(cpus))


def apply_presets(args):
    # the named configurations of BASELINE.json: C2 = bg-sub + CCL only, C3 = + posture (no network), C4 = + identity network,
    # C5 = everything (posture, posture-normalised crops, network, full per-blob record)
    if args.config == "C2" and args.stages == "all" and not args.force_all:
        args.stages = "segment"
    if args.config == "C3" and args.stages == "all" and not args.force_all:
        args.stages = "segment"; args.with_posture = True
    if args.config == "C5" and args.normalize == "none" and not args.force_all:
        args.normalize = "posture"
    if args.normalize == "posture":
        args.with_posture = True
    return args


def measure(args, env):
    """one measurement of the hot path as `args` describes it -> the output dictionary (keys of the driver's contract + roofline objects)"""
    import numpy as np
    import torch
    import torch.distributed as dist
    from trex_amd import capi, synth, weights

    world, rank, local, use_dist, dev = env["world"], env["rank"], env["local"], env["use_dist"] or args.force_dist, env["dev"]
    if use_dist and not dist.is_initialized():      # a secondary run of the N > 1 code path inside a single-process launch
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    W, H, n_ind, _cid = synth.CONFIGS[args.config]
    B = args.batch or (64 if args.config == "C5" else 256)
    if args.scaling == "strong":
        from trex_amd import dist as tdist
        B = tdist.strong_split(B, world)
    classes = 256 if args.config == "C5" else 100
    with_cnn = args.stages == "all"

    # frame-sharded: rank r owns frames r*B .. r*B+B-1 of every step's block (distinct data per rank)
    frames, bg = synth.batch_torch(args.config, B, dev, t0=rank * B)
    rgb = args.encoding == "rgb8"
    rgb_unaligned = False                       # torch allocations are 16-byte aligned: the matrix-core conv1 (and with it the default chain) always runs
    host_in = args.input.startswith("host-")
    bgra_in = rgb or args.input in ("bgra", "host-bgra")
    if bgra_in:     # the same scenes as BGRA tiles (what TRex's TileImage holds)
        frames_c = torch.stack([frames, frames, frames, torch.full_like(frames, 255)], dim=-1).contiguous()
    host_frames = None
    numa_note = None
    old_affinity = None
    if host_in:
        # two-socket hosts: where the tiles' pages, the pinned ring and the copy threads lie decides the path's speed (5.5 k vs 8.3 k frames/s
        # measured with the scheduler's choice).  The producer of this benchmark is pinned to the GPU's NUMA node (else the node it runs on)
        # before it allocates anything; libtrexhip's copy threads follow the caller's node (upload.hip)
        try:
            old_affinity = os.sched_getaffinity(0)
            numa_note = pin_to_gpu_node(local)
        except Exception as ex:      # noqa: BLE001
            numa_note = "not pinned (%s)" % ex
    if host_in:     # distinct pageable buffers, one per tile, like TRex's pooled Image::Ptr
        src = frames_c if bgra_in else frames
        host_frames = [np.ascontiguousarray(src[i].cpu().numpy()) for i in range(B)]
    max_blobs = 4 * n_ind
    state = weights.synthetic_state(classes, 4242, channels=3 if args.encoding == "rgb8" else 1)
    from trex_amd.pipeline import Pipeline
    blob = weights.pack_blob(state, classes, channels=3 if rgb else 1) if with_cnn else None

    def make_pipe(gather):
        comm_ids = None
        if gather == "library" and use_dist and world > 1:   # rank 0 makes the ncclUniqueId; torch.distributed only hands it out
            box = [[capi.Comm.unique_id()]] if rank == 0 else [None]      # one communicator per process, shared by the lanes
            dist.broadcast_object_list(box, src=0)
            comm_ids = box[0]
        # the lanes, their buffers and the step schedule live in trex_amd/pipeline.py (the same object tests/test_bench_shape_gpu.py checks)
        return Pipeline(W, H, n_ind, B, classes, bg, blob, local=local, rank=rank, world=world, use_dist=use_dist, comm_ids=comm_ids,
                        with_cnn=with_cnn, with_posture=args.with_posture, normalize=args.normalize, rgb=rgb, bgra_in=bgra_in,
                        cnn_mode=args.cnn_mode, lanes=args.lanes, pipeline=args.pipeline, detect_priority=args.detect_priority,
                        host_frames=host_frames, gather=gather)

    gather_by, pipe = args.gather, None
    if args.gather == "torch":
        gather_by, pipe = "torch.distributed", make_pipe("torch")
    elif use_dist and world > 1:
        # The table gather belongs to libtrexhip's own RCCL communicator (trexhip_comm_*).  If creating it fails on any rank (e.g. no
        # loadable librccl), every rank falls back TOGETHER to the same exchange through torch.distributed's communicator -- still RCCL
        # over xGMI, reported as such in config.gather -- instead of losing the multi-GPU measurement.
        err = ""
        try:
            pipe = make_pipe("library")
        except Exception as e:      # noqa: BLE001
            err = "%s: %s" % (type(e).__name__, e)
        ok = torch.tensor([0 if err else 1], device=dev, dtype=torch.int32)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 0:
            if err:
                print("rank %d: library communicator unavailable (%s); table gather through torch.distributed" % (rank, err), file=sys.stderr)
            if pipe is not None:
                pipe.close()
            gather_by, pipe = "torch.distributed", make_pipe("torch")
    else:
        pipe = make_pipe("library")
    if use_dist and gather_by == "library":
        gather_by = "libtrexhip (trexhip_comm_gather_device)"
    lanes = pipe.lanes
    seg = lanes[0].seg
    frames_ptr = frames_c.data_ptr() if bgra_in else frames.data_ptr()

    def run(k):
        return pipe.run(k, frames_ptr)

    guard_info = None
    if args.guard_trip and with_cnn:
        # one step for the crops, then conv1 (and what follows it: the BN statistics, so that the later layers keep their scale) is scaled
        # until between 12 and 24 crops of the step leave the range the fp16 pieces hold (>= 4368 behind conv1's ReLU)
        n0 = run(1)
        torch.cuda.synchronize()
        ln0 = lanes[0]

        def scaled(m):
            st = {k: v.copy() for k, v in state.items()}
            st["conv1.weight"] *= m; st["conv1.bias"] *= m; st["bn1.running_mean"] *= m
            st["bn2.running_var"] = st["bn2.running_var"] * m * m
            return weights.pack_blob(st, classes, channels=3 if rgb else 1)

        def count(m):
            ln0.seg.load_weights(scaled(m))
            ln0.seg.set_identity_precision(3)
            ln0.seg.identify_device(ln0.crops.data_ptr(), n0, ln0.probs.data_ptr())
            return ln0.seg.guard_stats()

        lo_m, hi_m = 1.0, None
        m = 4.0
        for _ in range(40):
            c, whole = count(m)
            if c < 12 and not whole:          # (a band of 12..24 crops, so that runs compare: the cost of a trip grows with the crops re-run)
                lo_m = m
                m = m * 4.0 if hi_m is None else 0.5 * (lo_m + hi_m)
            elif whole or c > 24:
                hi_m = m
                m = 0.5 * (lo_m + hi_m)
            else:
                break
        c, whole = count(m)
        guard_info = {"conv1_scale": m, "rerun_crops_per_step": c, "whole_batch": whole, "crops_per_step": n0}
        for ln in lanes:
            ln.seg.load_weights(scaled(m))
            ln.seg.set_identity_precision(3)

    n_blobs = run(args.warmup) if args.warmup else 0

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # Batches of a few frames (TRex's default detect_batch_size is 1: core/default_config.cpp:1113) are latency chains of small launches: the
    # library replays their identify chain as a hipGraph, which it does not do while its stage timers are on, and every stage timer is two more
    # event records per stage and step.  Those rows are timed the way a deployment runs them -- timers off -- and their stage times are taken
    # from a short second run with the timers on.  The headline configurations keep the timers inside the timed region (the contract).
    timers_in_region = B > 16
    for ln in lanes:
        ln.seg.profile_enable(timers_in_region)
        ln.seg.profile_reset()
    barrier()
    t0 = time.perf_counter()
    n_blobs = run(args.steps)
    torch.cuda.synchronize()
    dt_own = time.perf_counter() - t0          # this rank's own K steps, before it waits for the others
    barrier()
    dt = time.perf_counter() - t0
    if not timers_in_region:
        for ln in lanes:
            ln.seg.profile_enable(True)
            ln.seg.profile_reset()
        run(min(args.steps, 50))
        torch.cuda.synchronize()
    dist_info = None
    if use_dist:
        # every rank's own clock around the same K steps: the job's time is the slowest rank's (the contract); the spread and rank 0's
        # extra (it alone receives the tables and copies the slab to its host) are reported beside it
        mine = torch.zeros(2 * world, device=dev, dtype=torch.float64)
        mine[rank] = dt
        mine[world + rank] = dt_own
        dist.all_reduce(mine, op=dist.ReduceOp.SUM)
        both = [float(x) for x in mine.tolist()]
        dt = max(both[:world])                  # the job's time: the slowest rank's clock around the barriers (the contract)
        per_rank = both[world:]                 # every rank's own clock, taken BEFORE the closing barrier: the spread and rank 0's extra are real
        others = per_rank[1:] or per_rank
        dist_info = {"ranks_seen": pipe.shared_comm.count_ranks() if pipe.shared_comm is not None else int(sum(1 for x in per_rank if x > 0)),
                     "ranks_seen_by": "all-reduce of 1 over libtrexhip's communicator" if pipe.shared_comm is not None else "torch.distributed",
                     "gather_bytes_per_step": (world - 1) * pipe.rows * pipe.rowlen * 4, "table_bytes_per_rank": pipe.rows * pipe.rowlen * 4,
                     "ms_per_step_min": min(per_rank) / args.steps * 1e3, "ms_per_step_max": max(per_rank) / args.steps * 1e3,
                     "rank0_extra_ms": (per_rank[0] - sum(others) / len(others)) / args.steps * 1e3}
        if rank == 0 and with_cnn:
            # what rank 0's matcher would read after the last step: the gathered slab merged into Tracker::add order (tracking/Tracker.cpp:586-587)
            # must hold every frame of the step's block of world x B frames exactly once (n_ind rows each), ordered by frame
            from trex_amd import dist as tdist
            last = lanes[(args.steps - 1) % len(lanes)]
            merged = tdist.merge_tables(last.table_host.numpy().view(np.uint32))
            fr = merged[:, 0].astype(np.int64)
            first = (args.steps - 1) * world * B
            cnt = np.bincount(fr - first, minlength=world * B) if len(fr) and fr.min() >= first and fr.max() < first + world * B else np.zeros(1)
            dist_info["merged"] = {"frames": int((cnt > 0).sum()), "frames_expected": world * B, "rows": int(len(merged)),
                                   "every_frame_once": bool(len(cnt) == world * B and (cnt == n_ind).all()),
                                   "tracker_order": bool((np.diff(fr) >= 0).all())}
    prof = {}
    for name in ("ROWS", "SEGMENT_ALL", "CONV2", "CONV3", "CNN_ALL", "CROPS", "POSTURE"):
        ms = cnt = 0
        for ln in lanes:
            a_, b_ = ln.seg.profile_read(getattr(capi, "STAGE_" + name))
            ms += a_; cnt += b_
        prof[name] = (ms, cnt)
    # the detect stage overlaps the identity network of the previous batch when two lanes are pipelined, so its in-flight
    # HIP-event durations include time lost to sharing the CUs; its roofline is taken from a short serial pass afterwards
    if len(lanes) > 1:
        ln0 = lanes[0]
        ln0.seg.profile_reset()
        for _ in range(5):
            ln0.detect(frames_ptr)
            ln0.seg.fetch(copy=False)
        prof["ROWS"] = ln0.seg.profile_read(capi.STAGE_ROWS)
        prof["SEGMENT_ALL"] = ln0.seg.profile_read(capi.STAGE_SEGMENT_ALL)
    for ln in lanes:
        ln.seg.profile_enable(False)
    # ... and as the pipeline runs it: the lanes' contexts issue alternate detect passes on their own streams, so the labelling / gather of one
    # batch (latency chains, one workgroup per frame) run under the pixel pass of the next; wall clock over 2 x 10 passes, nothing else on the GPU
    pipe_detect_s = None
    if len(lanes) > 1 and not host_in:
        for ln in lanes[:2]:
            ln.detect(frames_ptr)
        torch.cuda.synchronize()
        tp0 = time.perf_counter()
        for i in range(20):
            lanes[i % 2].detect(frames_ptr)
        torch.cuda.synchronize()
        pipe_detect_s = (time.perf_counter() - tp0) / 20
        for ln in lanes[:2]:
            ln.seg.fetch(copy=False)

    def pmc_traffic(kernel_prefix):
        """HBM bytes per launch from the committed PMC passes of this same command (profiles/rNN_pmc_summary.json, newest round;
        separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, corrected as MI355X_MICROARCH.md prescribes); only valid for the
        default C4 / 256-frame workload they were collected on, else null.  Counters cannot be read from inside this process."""
        try:
            if host_in or bgra_in or args.scaling != "weak":
                return None
            import glob
            if args.config == "C4" and B == 256:
                files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary.json")))
                j = json.load(open(files[-1]))
                for k, v in j["kernels"].items():
                    if k.startswith(kernel_prefix):
                        return v["hbm_bytes"]
                return None
            # the detect kernels of the other configurations at their default batch (profiles/rNN_pmc_detect_configs.json: FETCH / WRITE passes of
            # `bench.py --config C --stages segment --force-all`); the bytes scale with the frames per launch
            files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_detect_configs.json")))
            j = json.load(open(files[-1]))[args.config]
            for k, v in j["kernels"].items():
                if ("trexhip::" + k).startswith(kernel_prefix) or k.startswith(kernel_prefix.replace("trexhip::", "")):
                    return v["hbm_bytes"] * B / j["frames_per_launch"]
        except Exception:
            pass
        return None

    def avg_s(name):
        ms, n = prof[name]
        return (ms / n) * 1e-3 if n else 0.0

    total_frames = world * B * args.steps
    fps = total_frames / dt
    seg_bytes = 2.0 * W * H * B                       # algorithmic bytes of the pixel pass: frame + background
    rows_s, segall_s = avg_s("ROWS"), avg_s("SEGMENT_ALL")
    out = {
        "metric": "frames/s end-to-end (segment+CNN-ID), 2048x2048 x100 individuals" if (args.config == "C4" and with_cnn) else
                  f"frames/s ({'segment' + ('+posture' if args.with_posture else '') + ('+CNN-ID' if with_cnn else '')}), {W}x{H} x{n_ind} individuals ({args.config})",
        "value": fps, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": args.scaling,
        "vs_baseline": None, "dtype": ({"fp32": "f32", "bf16x6": "bf16x6-split (fp32-equivalent: 3 bf16 pieces per operand, 6 MFMA products, fp32 accumulate)", "bf16x3": "bf16x3-split", "fp16x3": "fp16x3-split (fp32-class: 2 fp16 pieces per operand = 22 mantissa bits, 3 MFMA products, fp32 accumulate, range-guarded)"}[args.cnn_mode] if with_cnn else "u8"), "data": "synthetic",
        "config": {"workload": f"{args.config}: {W}x{H} gray, {n_ind} individuals/frame, {B} frames resident per step per GPU, "
                               f"80x80x1 crops, {classes}-way V118_3 (random-init weights)",
                   "stages": "detect(bg-sub+threshold+CCL+filter+gather, tables->host) + crops(none) + identity CNN (V118_3) + per-blob ID table (gathered on rank 0 over RCCL when N>1: trexhip_comm_gather_device) -> rank-0 host"
                             if with_cnn else "detect only (bg-sub+threshold+CCL+filter+gather, tables->host)",
                   "frames_per_step_per_gpu": B, "encoding": args.encoding, "input": ("host " if host_in else "") + ("bgra tiles (cvtColor on the device inside the step)" if bgra_in else "gray frames") + (" in pageable host memory, PCIe inside the step" if host_in else " resident in HBM"), "posture": bool(args.with_posture), "individual_image_normalization": args.normalize, "pipelined_lanes": len(lanes), "stage_timers_in_timed_region": timers_in_region, "blobs_per_step_rank0": n_blobs, "parallelism": f"frame-sharded x{world}" + (" (all ranks on ONE GPU over loopback sockets: functional check only)" if args.same_gpu else ""), **({"gather": gather_by} if use_dist else {})},
    }
    if dist_info is not None:
        out["dist"] = dist_info
    if guard_info is not None:
        guard_info["rerun_crops_last_step"] = lanes[0].seg.guard_stats()[0]
        out["range_guard"] = guard_info
    rows_traffic = pmc_traffic("trexhip::k_rows")
    pass_traffic = None
    if rows_traffic is not None:
        others = [pmc_traffic("trexhip::k_ccl_lds"), pmc_traffic("trexhip::k_ccl_band"), pmc_traffic("trexhip::k_gather")]
        pass_traffic = rows_traffic + sum(x for x in others if x)
    # The primary fraction is by the HBM bytes the counters saw (the background is re-read from L2 / MALL, not from HBM); the fraction by
    # the contract's algorithmic bytes (frame + background per pixel) is kept beside it and can pass 1 for that reason.
    seg_roof = {"kernel": "k_rows32b (pixel pass: subtract / threshold / run extraction; background row in registers for 8 frames, compile-time threshold modes)", "bound": "hbm", "peak": 8000.0, "unit": "GB/s",
                "achieved": (rows_traffic / rows_s / 1e9) if (rows_traffic and rows_s) else (seg_bytes / rows_s / 1e9 if rows_s else 0.0),
                "frac": (rows_traffic / rows_s / 8e12) if (rows_traffic and rows_s) else (seg_bytes / rows_s / 8e12 if rows_s else 0.0),
                "frac_basis": "measured HBM traffic per launch (PMC)" if rows_traffic else "algorithmic bytes (no PMC summary for this workload)",
                "traffic": rows_traffic,
                "traffic_note": "HBM bytes/launch = FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE, separate --pmc passes, profiles/rNN_pmc_summary.json (collected on this command, not inside this run)",
                "avg_launch_us": rows_s * 1e6, "launches": prof["ROWS"][1], "algorithmic_bytes_per_launch": seg_bytes,
                "frac_algorithmic_bytes": seg_bytes / rows_s / 8e12 if rows_s else None,
                "whole_detect_pass_us": segall_s * 1e6,
                "whole_detect_pass_frac": (pass_traffic / segall_s / 8e12) if (pass_traffic and segall_s) else None,
                "whole_detect_pass_frac_algorithmic_bytes": seg_bytes / segall_s / 8e12 if segall_s else None,
                "whole_detect_pass_traffic": pass_traffic,
                "pipelined_detect_pass_us": pipe_detect_s * 1e6 if pipe_detect_s else None,
                "pipelined_detect_pass_frac": (pass_traffic / pipe_detect_s / 8e12) if (pass_traffic and pipe_detect_s) else None,
                "pipelined_note": "the same three kernels per pass, passes issued alternately by the pipeline's two contexts on their own streams (labelling / gather of one batch under the pixel pass of the next): wall clock per pass over 20 passes with nothing else on the GPU; whole_detect_pass_* is one context alone, serial",
                "timing": "HIP events on the kernel stream; 5 serial detect passes after the timed region when lanes are pipelined (in-flight the stage shares the GPU with the identity network)",
                "limiter": "read-only HBM streaming + VALU: the same kernel with the masks replaced by a trivial compare takes 187-200 us (a plain grid-stride streaming read of the same 1.07 GB: 167 us = 6.4 TB/s, tools/bw_probe.hip; torch's fastest read-only reduction: 183 us); exact masks and run extraction add the rest. A/B on one box (tools/rows_exp.sh): k_rows32 generic 250 us, + compile-time threshold modes 242, background row in registers for 8 frames (k_rows32b) 235, both 220, rows fastest 215. k_ccl_lds / k_gather are latency chains of one workgroup per frame / half a wave per blob",
                "frac_of_measured_copy_bw": ((rows_traffic or seg_bytes) / rows_s / 6.29e12) if rows_s else None}
    if host_in:
        bytes_step = float(B * W * H * (4 if bgra_in else 1))
        up = [ln.seg.profile_read(capi.STAGE_UPLOAD_COPY) + ln.seg.profile_read(capi.STAGE_UPLOAD_DMA) for ln in lanes]
        cms = sum(u[0] for u in up); cn = sum(u[1] for u in up); dms = sum(u[2] for u in up); dn = sum(u[3] for u in up)
        host_reduced = bgra_in and not rgb          # gray pixel arrays: the upload threads reduce BGRA tiles to gray while they copy (hostcvt.cpp)
        pcie_step = float(B * W * H) if host_reduced else bytes_step
        if old_affinity is not None:
            try:
                os.sched_setaffinity(0, old_affinity)
            except OSError:
                pass
        out["host_input"] = {"producer_numa": numa_note, "tile_bytes_per_step": bytes_step, "tile_GB_per_s": bytes_step * world * args.steps / dt / 1e9,
                             "pcie_bytes_per_step": pcie_step, "pcie_GB_per_s": pcie_step * world * args.steps / dt / 1e9, "pcie_peak_GB_per_s": 63.0,
                             "frac_of_pcie_peak": pcie_step * world * args.steps / dt / 63e9,
                             "host_copy_ms_per_frame": cms / cn if cn else None, "dma_ms_per_frame": dms / dn if dn else None,
                             "dma_GB_per_s": (pcie_step / B) / (dms / dn * 1e-3) / 1e9 if dn and dms else None,
                             "note": "pageable tiles -> pinned ring (host threads" + (", which reduce the BGRA tiles to gray on the way: a quarter of the tile bytes cross PCIe" if host_reduced else "") + ") -> HBM (async DMA per chunk of frames); the two legs overlap, the segment kernels and the identity network of the previous batch overlap both"}
    if with_cnn:
        c3_s, c2_s = avg_s("CONV3"), avg_s("CONV2")
        nprod = {"fp32": 1, "bf16x6": 6, "bf16x3": 3, "fp16x3": 3}[args.cnn_mode]
        peak = 157.3 if args.cnn_mode == "fp32" else 2500.0
        geom = int(os.environ.get("TREXHIP_CONV_GEOM", "0"))
        # fp16x3 (default chain, trex_amd/csrc/cnn_wpre.h): conv2 and conv3 run as Winograd F(4,5) convolutions along x on operand images
        # written by the producing layer -- 40 position GEMMs over 4-pixel tiles instead of 25 tap GEMMs over pixels (x0.4), 3 piece products
        # per product: 1.2 issued flops per algorithmic flop (conv2: 60 of the 64 M-slots of a pass are real tiles: 1.28)
        pre = args.cnn_mode == "fp16x3" and not rgb_unaligned and (geom & 0xfff) == 0
        wino3 = args.cnn_mode == "fp16x3" and not (geom & 256)
        if args.cnn_mode == "fp32":
            k3, k2, r3, r2 = "k_conv5<64,128,20,20,32> (conv3, fp32 MFMA)", "k_conv5<16,64,40,20,16> (conv2, fp32 MFMA)", 1.0, 1.0
        elif pre:
            k3 = "k_conv5_wpair (conv3 on the pair-major V3 operand image: Winograd F(4,5) along x walked position pair by position pair, the output transform under the next pair's taps; fp16 two-piece split, 3 MFMA products per transformed product, fp32 accumulate)"
            k2 = "k_conv2_wpre2 (conv2 on the V2 operand image, writes V3: same arithmetic, two workgroups per CU, LDS-DMA staging)"
            r3, r2 = 1.2, 1.28
            fused12 = not rgb and not (geom & (1 << 28))
            role_split = fused12 and not (geom & (1 << 30)) and ((geom & (1 << 29)) or n_blobs >= 3200)
            if fused12:
                k2 = "k_conv12_wpre (conv1 INSIDE conv2: the V2 operand image is produced into LDS from the u8 crops, never written to HBM; writes V3)"
            if role_split:
                k2 = "k_conv12_rs (conv1 INSIDE conv2, role-split: consumer waves run the tap loop and the output transform, producer waves the V3 transform of the previous pass and the V2 rows of the next; one workgroup of 8 waves per CU; writes V3)"
        elif args.cnn_mode == "fp16x3":
            k3 = "k_conv5_wino<64,128,20,2> (conv3, Winograd F(4,5), in-kernel transform)" if wino3 else "k_conv5_stream<64,128,20,20,8,persistent> (conv3, direct, fp16 MFMA x3)"
            k2, r3, r2 = "k_conv5_stream<16,64,40,8,4> (conv2, direct form, fp16 MFMA x3 per fp32 product)", (1.2 if wino3 else 3.0), 3.0
        else:
            k3, k2 = f"k_conv5_split<64,128,20,20,{args.cnn_mode}> (conv3, 16-bit MFMA x{nprod})", f"k_conv5_split<16,64,40,10,{args.cnn_mode}> (conv2, 16-bit MFMA x{nprod})"
            r3 = r2 = float(nprod)

        def conv_roof(kname, sec, flop_per_crop, ratio, launches, act_bytes):
            fl = flop_per_crop * n_blobs
            return {"kernel": kname, "bound": "mfma", "achieved": fl / sec / 1e12 if sec else 0.0, "peak": peak, "unit": "TFLOP/s",
                    "frac": fl / sec / (peak * 1e12) if sec else 0.0, "mfma_flop_issued_per_algorithmic_flop": ratio,
                    "mfma_issue_frac": ratio * fl / sec / (peak * 1e12) if sec else 0.0, "traffic": None,
                    "avg_launch_us": sec * 1e6, "launches": launches, "total_us": sec * 1e6 * launches, "algorithmic_flop_per_launch": fl,
                    "algorithmic_bytes_per_launch": act_bytes}
        # HBM bytes the layer has to move at least (operands in + results out); the default chain's images are 2 x the fp32 activation bytes
        in2 = n_blobs * 40 * 5120 if pre else n_blobs * 40 * 40 * 16 * 4
        out2 = n_blobs * 20 * 10240 if pre else n_blobs * 20 * 20 * 64 * 4
        roof3 = conv_roof(k3, c3_s, FLOP_PER_CROP_CONV3, r3, prof["CONV3"][1], out2 + n_blobs * 10 * 10 * 128 * 4)
        fused12 = pre and k2.startswith("k_conv12")
        if fused12:
            in2 = n_blobs * 6400                      # the u8 crops are all the fused kernel reads
        roof2 = conv_roof(k2, c2_s, FLOP_PER_CROP_CONV2 + (FLOP_PER_CROP_CONV1 if fused12 else 0.0), r2, prof["CONV2"][1], in2 + out2)
        t3 = pmc_traffic("trexhip::k_conv5_wpair" if pre else ("trexhip::k_conv5_wino<64, 128, 20, 2" if wino3 else "trexhip::k_conv5_stream<64, 128, 20, 20, 8"))
        t2 = pmc_traffic((("trexhip::k_conv12_rs" if k2.startswith("k_conv12_rs") else "trexhip::k_conv12_wpre") if fused12 else "trexhip::k_conv2_wpre2") if pre else "trexhip::k_conv5_stream<16, 64, 40, 8, 4")
        if args.cnn_mode == "fp16x3":
            roof3["traffic"], roof2["traffic"] = t3, t2
        # the primary roofline object is the convolution with the larger total duration in this run; both are carried
        dom = roof2 if roof2["total_us"] > roof3["total_us"] else roof3
        out["roofline"] = dict(dom)
        out["roofline"]["traffic_note"] = "HBM bytes/launch from profiles/rNN_pmc_summary.json (FETCH_SIZE x2 + WRITE_SIZE, separate --pmc passes of this command)"
        out["roofline"]["peak_note"] = "achieved counts ALGORITHMIC flops (2 per fp32 multiply-add of the dense 5x5 convolution); peak is the dense MFMA peak of the instruction used (fp32: 157.3, 16-bit: 2500 TFLOP/s, MI355X_MICROARCH.md); the matrix pipe issues `mfma_flop_issued_per_algorithmic_flop` flops per algorithmic flop, i.e. it is busy mfma_issue_frac of its peak"
        out["roofline_kernels"] = {"conv2": roof2, "conv3": roof3}
        cnn_s = avg_s("CNN_ALL")
        out["stage_us"] = {"detect": segall_s * 1e6, "posture": avg_s("POSTURE") * 1e6 if args.with_posture else None, "crops": avg_s("CROPS") * 1e6, "conv2": avg_s("CONV2") * 1e6,
                           "conv3": c3_s * 1e6, "cnn_all": cnn_s * 1e6,
                           "cnn_all_tflops": FLOP_PER_CROP_TOTAL * n_blobs / cnn_s / 1e12 if cnn_s else None}
        out["roofline_detect"] = seg_roof
    else:
        out["roofline"] = seg_roof

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # the CPU restatement (oracle/: a port, not the TRex binary) on this box's host cores, on a bounded sample of the same workload:
        # all usable cores (one frame per OpenMP thread / torch intra-op threads) and ONE thread; medians over the repetitions
        from oracle import oracle, cnn_oracle
        ncpu = usable_cores()
        budget = args.cpu_seconds
        hsrc = host_frames if (host_in and not bgra_in) else None
        k = max(2, min(B, ncpu))
        sample = np.stack(hsrc[:k]) if hsrc is not None else frames[:k].cpu().numpy()
        bgh = bg.cpu().numpy()
        op = oracle.make_params(W, H)

        def timed(fn, seconds, min_reps=3, max_reps=50):
            for _ in range(2):
                fn()
            ts = []
            t_end = time.perf_counter() + seconds
            while len(ts) < min_reps or (time.perf_counter() < t_end and len(ts) < max_reps):
                t = time.perf_counter(); fn(); ts.append(time.perf_counter() - t)
            return float(np.median(ts)), len(ts)

        share = budget / (4.0 if with_cnn else 2.0)
        t_all, r_all = timed(lambda: oracle.segment_batch(sample, bgh, op, ncpu), share)
        t_one, r_one = timed(lambda: oracle.segment(sample[0], bgh, op), share, min_reps=20)
        cpu = {"unit": "frames/s", "cores": ncpu, "kind": "port",
               "detect_frames_per_s": k / t_all, "detect_frames_per_s_1_thread": 1.0 / t_one,
               "repetitions": {"detect_all_cores": r_all, "detect_1_thread": r_one}, "statistic": "median"}
        if with_cnn:
            b_, r_, _ = oracle.segment(sample[0], bgh, op)
            cr = np.stack([oracle.crop_none(sample[0], bgh, bb, r_) for bb in b_])[..., None]
            t_call, r_call = timed(lambda: cnn_oracle.predict(state, cr, threads=ncpu), share)
            t_c1, r_c1 = timed(lambda: cnn_oracle.predict(state, cr[:10], threads=1), share, min_reps=3)
            cnn_fps, cnn_fps1 = 1.0 / t_call, 1.0 / (t_c1 * len(cr) / 10.0)
            cpu["identify_frames_per_s"] = cnn_fps
            cpu["identify_frames_per_s_1_thread"] = cnn_fps1
            cpu["repetitions"].update({"identify_all_cores": r_call, "identify_1_thread": r_c1})
            cpu["value"] = 1.0 / (t_all / k + 1.0 / cnn_fps)
            cpu["value_1_thread"] = 1.0 / (t_one + 1.0 / cnn_fps1)
            cpu["sample"] = (f"detect: {k} distinct frames of the batch, one per OpenMP thread (all cores) / frame 0 alone (1 thread); "
                             f"identify: the {len(cr)} crops of frame 0 through a torch-CPU restatement of V118_3 ({ncpu} threads) / 10 of them on 1 thread, scaled; "
                             f"medians, about {budget:.0f} s in total; restatements under oracle/, not the TRex binary")
        else:
            cpu["value"] = k / t_all
            cpu["value_1_thread"] = 1.0 / t_one
            cpu["sample"] = f"{k} distinct frames of the batch, one per OpenMP thread (all cores) / frame 0 alone (1 thread); medians, about {budget:.0f} s in total; oracle/ restatement"
        out["cpu_baseline"] = cpu
    for ln in lanes:
        ln.seg.close()
    return out


def train_step_bench(dev, n=128, classes=100, steps=100, precision=0):
    """one optimizer step of V118_3 (forward, backward, Adam) on `n` samples: SURVEY 8(f)3 -- ms per step, algorithmic TFLOP/s against
    the fp32 matrix peak (157.3).  precision 0 (the library's default): conv2 / conv3 forward and data gradients in the inference path's
    fp16 two-piece split arithmetic; 1: exact fp32 MFMA everywhere"""
    import numpy as np
    import torch
    from trex_amd import capi, weights
    state = weights.synthetic_state(classes, 1)
    x, y = weights.synthetic_train_batch(n, 2, classes, 1)
    p = capi.default_params(64, 64)
    p.max_batch = 1
    seg = capi.Segmenter(p)
    tr = capi.Trainer(seg, weights.pack_blob(state, classes, 1), max_batch=n, lr=1e-3, seed=3, precision=precision)
    dx, dy = torch.from_numpy(x).to(dev), torch.from_numpy(y.astype(np.int32)).to(dev)
    for _ in range(10):            # (a 16 ms window right behind the other secondaries read 15 % slower than 100 steps: clocks still settling)
        tr.step_device(dx.data_ptr(), dy.data_ptr(), n, 0, want_loss=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        tr.step_device(dx.data_ptr(), dy.data_ptr(), n, 0, want_loss=False)
    seg.synchronize()
    dt = (time.perf_counter() - t0) / steps
    tr.close()
    seg.close()
    # forward 3 convolutions + fc (MAC per sample) x 3 (forward, data gradient, weight gradient; conv1 has no data gradient)
    gflop = 2 * n * (2.56e6 * 2 + 40.96e6 * 3 + 81.92e6 * 3 + 1.28e6 * 3) / 1e9
    dtype = ("f32; conv2 / conv3 forward, data and weight gradients: fp16x3-split (2 fp16 pieces per operand = 22 mantissa bits, 3 MFMA products, fp32 accumulate, "
             "power-of-two scales per staged patch and per layer)") if precision == 0 else "f32 (exact fp32 MFMA)"
    kern = ("the whole step (k_t_conv5_h2 forward / data gradients and k_t_wgrad_h2 weight gradients on the 16-bit matrix cores, batch-norm / pool / head kernels, k_t_adam)"
            if precision == 0 else "the whole step (k_conv5<RAW> forward / data gradients, k_t_wgrad*, batch-norm / pool / head kernels, k_t_adam)")
    return {"metric": "training step of the identity network (V118_3: forward, cross entropy, backward, Adam)", "value": n / dt, "unit": "samples/s",
            "ms_per_step": dt * 1e3, "steps": steps, "config": {"workload": f"{n} samples of 80x80x1, {classes} classes, dropout 0.05, library-drawn masks",
                                                                 "trexhip_train_params.precision": precision},
            "dtype": dtype,
            "roofline": {"kernel": kern, "bound": "mfma",
                         "achieved": gflop / dt / 1e3, "peak": 157.3, "unit": "TFLOP/s", "frac": gflop / dt / 1e3 / 157.3, "traffic": None,
                         # against the peak of the instruction that does the work: three fp16 products per term on the 2.5 PFLOP/s cores
                         "frac_split_peak": (gflop / dt / 1e3 / (2500.0 / 3.0)) if precision == 0 else None,
                         "peak_note": "algorithmic flops of the step against the fp32 matrix peak (the yardstick of rounds 2-3); with precision 0 about 95 % of "
                                      "those flops run on the 16-bit matrix cores (peak 2500), whose three products per term are not counted",
                         "algorithmic_gflop_per_step": gflop}}


SECONDARY = [
    # name, argument overrides (on top of the defaults), steps, warmup (>= 2: both pipelined contexts have run every stage -- and allocated
    # their buffers -- before the timed region).  12 steps where a step is 10 ms: the timed region ends with a drained pipeline, and over 4
    # steps that last, un-overlapped step read as 5 % (C4_force_dist 26.7 k against the 28.1 k of the same code path over 20 steps)
    ("C2", {"config": "C2"}, 30, 4),
    # 256 frames of 1280 x 720 are 295 MB: a launch that short leaves one labelling workgroup per CU and a pixel pass of 60 us.  The C ABI takes any
    # max_batch (TRex's own detect_batch_size is a uchar: the adapter stops at 255); at 1024 frames several frames share a CU and k_ccl_lds runs in
    # its small instance (round 6: 3.6 -> 3.7 TB/s of counter bytes over the whole pass)
    ("C2_batch1024", {"config": "C2", "batch": 1024}, 16, 4),
    ("C3", {"config": "C3"}, 12, 2),
    ("C5", {"config": "C5"}, 12, 3),
    ("C4_posture_normalised", {"normalize": "posture"}, 12, 3),
    ("C4_input_bgra_device", {"input": "bgra"}, 12, 3),
    ("C4_input_host_bgra", {"input": "host-bgra"}, 6, 2),
    ("C4_input_host_gray", {"input": "host-gray"}, 6, 2),
    ("C4_encoding_rgb8", {"encoding": "rgb8"}, 12, 3),
    ("C4_cnn_fp32", {"cnn_mode": "fp32"}, 4, 2),
    ("C4_no_pipeline", {"pipeline": False}, 12, 2),
    ("C4_force_dist", {"force_dist": True}, 12, 3),
    ("C4_detect_only", {"stages": "segment", "force_all": True}, 20, 4),
    # SURVEY 8(d) / BASELINE.md fix the resident batch of the contract at 64 frames (C5: 16); TRex's own detect_batch_size is a uchar that defaults
    # to 1 (core/default_config.cpp:1113): the headline's 256 (C5: 64) is what 288 GB of HBM invites, these rows are what the contract and the
    # reference's default give
    ("C4_batch64", {"batch": 64}, 40, 6),
    ("C5_batch16", {"config": "C5", "batch": 16}, 24, 4),
    ("C4_batch1", {"batch": 1}, 300, 20),
    # a handful of the step's 25600 crops out of the fp16-piece range: the per-crop range guard re-runs them (bf16x6) inside every step
    ("C4_guard_tripped", {"guard_trip": True}, 12, 3),
]


def secondary(args, env):
    """short runs of the other configurations / input paths / precisions in this same process, so that every number DESIGN.md quotes is in
    the driver's record; each entry carries its own roofline.  Entries that fail are reported as {"error": ...}, never dropped."""
    import copy
    out = {}
    only = set(x for x in args.secondary_only.split(",") if x)
    base = build_parser().parse_args([])
    # the training step first: behind the other runs of this process it read 0.88 instead of 0.75 ms (not the clocks: a stand-alone run right
    # behind 200 bench steps reads 0.75; with only C4_detect_only in front of it 0.75 too -- what the host-input / N > 1 code-path runs leave
    # behind, worker threads, the communicator, streams, was not separated further)
    for key, prec in (("train_step", 0), ("train_step_fp32", 1)):
        if not only or key in only:
            try:
                out[key] = train_step_bench(env["dev"], precision=prec)
            except Exception as ex:      # noqa: BLE001
                out[key] = {"error": "%s: %s" % (type(ex).__name__, ex)}
    for name, over, steps, warmup in SECONDARY:
        if only and name not in only:
            continue
        a = copy.copy(base)
        for k, v in over.items():
            setattr(a, k, v)
        a.steps, a.warmup, a.no_cpu_baseline, a.no_secondary = steps, warmup, True, True
        apply_presets(a)
        try:
            r = measure(a, env)
            e = {k: r[k] for k in ("metric", "value", "unit", "ms_per_step", "steps", "warmup", "dtype") if k in r}
            e["workload"] = r["config"]["workload"]
            e["input"] = r["config"].get("input")
            e["roofline"] = {k: r["roofline"].get(k) for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_us", "frac_basis", "whole_detect_pass_us", "whole_detect_pass_frac", "whole_detect_pass_frac_algorithmic_bytes", "pipelined_detect_pass_us", "pipelined_detect_pass_frac") if k in r["roofline"]}
            for k in ("stage_us", "host_input", "dist", "range_guard"):
                if k in r:
                    e[k] = r[k]
            if name in ("C4_input_host_bgra", "C4_input_host_gray"):      # run-to-run spread of the host-input paths: two more short repetitions
                vals = [r["value"]]
                for _ in range(2):
                    vals.append(measure(a, env)["value"])
                e["repetitions"] = vals
                e["spread"] = (max(vals) - min(vals)) / (sum(vals) / len(vals))
                # the path's speed depends on where the host pages and the pinned ring happen to lie (two sockets): the entry's value is
                # the MEDIAN of the three repetitions (each a fresh pipeline with its own host frames), all three are listed
                med = sorted(vals)[1]
                e["ms_per_step"] = e["ms_per_step"] * e["value"] / med
                e["value"] = med
                e["value_is"] = "median of `repetitions`"
            out[name] = e
        except Exception as ex:      # noqa: BLE001
            out[name] = {"error": "%s: %s" % (type(ex).__name__, ex)}
    return out


def self_spawn(args, argv):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves (one process per GPU, torch.distributed.run on 127.0.0.1)
    and pass their output through, so that a plain command line can never record an N = 1 number under n_gpus = N"""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + argv
    return subprocess.call(cmd)


FINAL_LINE_MAX = 4096       # bytes: the driver keeps the last ~12 KB of stdout; round 3's 22.8 KB line was cut and nothing was parsed


def _r(x, n=4):
    """numbers of the final line: n significant digits are what the measurement carries"""
    if isinstance(x, float):
        return float("%.*g" % (n, x))
    return x


def _pick(d, keys, n=4):
    return {k: _r(d[k], n) for k in keys if d is not None and k in d}


def emit_detail(tag, obj):
    """the long record: one line on stdout BEFORE the final line (prefixed, so that no parser takes it for the result) and a file
    under gpurun_out/ (merged back by gpurun)"""
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", tag + ".json"), "w") as f:
            json.dump(obj, f, indent=1)
    except OSError:
        pass
    print("# %s: %s" % (tag, json.dumps(obj)), flush=True)


def compact_line(out, sec):
    """the final line: the contract's keys, the roofline objects as numbers, cpu_baseline, and one {value, frac} pair per secondary run.
    Prose (what each kernel is, how each figure is taken) lives in DESIGN.md section 6; the long record is in gpurun_out/bench_detail.json"""
    o = {k: _r(out[k], 6) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline") if k in out}
    o["dtype"] = out["dtype"].split(" ")[0]
    o["data"] = out["data"]
    c = out["config"]
    o["config"] = {"workload": c["workload"], **_pick(c, ("frames_per_step_per_gpu", "encoding", "posture", "individual_image_normalization", "pipelined_lanes",
                                                          "blobs_per_step_rank0", "parallelism", "gather"))}
    o["config"]["input"] = c["input"][:60]
    roof_keys = ("bound", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_us", "launches", "mfma_issue_frac")
    r = out["roofline"]
    o["roofline"] = {"kernel": r["kernel"].split(" ")[0], **_pick(r, roof_keys)}
    if o["roofline"].get("traffic") is not None:      # the counters cannot be read from inside this process: the bytes are the committed PMC passes of this same command
        o["roofline"]["traffic_source"] = "profiles/rNN_pmc_summary.json (separate --pmc passes of this command; not measured in this run)"
    o["roofline"]["frac"] = o["roofline"]["achieved"] / o["roofline"]["peak"] if o["roofline"]["peak"] else 0.0    # consistent with the rounded `achieved`
    if "roofline_kernels" in out:
        o["roofline_kernels"] = {k: {"kernel": v["kernel"].split(" ")[0], **_pick(v, ("achieved", "frac", "traffic", "avg_launch_us"))} for k, v in out["roofline_kernels"].items()}
    det_keys = ("achieved", "peak", "frac", "traffic", "avg_launch_us", "frac_algorithmic_bytes", "whole_detect_pass_us", "whole_detect_pass_frac",
                "whole_detect_pass_frac_algorithmic_bytes", "whole_detect_pass_traffic", "pipelined_detect_pass_us", "pipelined_detect_pass_frac")
    if "roofline_detect" in out:
        o["roofline_detect"] = {"kernel": out["roofline_detect"]["kernel"].split(" ")[0], "bound": "hbm", "unit": "GB/s", **_pick(out["roofline_detect"], det_keys)}
    elif r.get("bound") == "hbm":
        o["roofline"].update(_pick(r, det_keys))
    if "stage_us" in out:
        o["stage_us"] = {k: _r(v, 4) for k, v in out["stage_us"].items() if v is not None}
    if "host_input" in out:
        o["host_input"] = _pick(out["host_input"], ("pcie_GB_per_s", "frac_of_pcie_peak", "host_copy_ms_per_frame", "dma_ms_per_frame"))
    if "dist" in out:
        o["dist"] = {k: (_r(v, 4) if not isinstance(v, dict) else v) for k, v in out["dist"].items()}
    cb = out.get("cpu_baseline")
    if cb:
        o["cpu_baseline"] = {**_pick(cb, ("value", "unit", "cores", "kind", "value_1_thread", "detect_frames_per_s", "identify_frames_per_s")), "sample": cb["sample"][:300]}
    if sec is not None:
        cfgs = {}
        for name, e in sec.items():
            if "error" in e:
                cfgs[name] = {"error": e["error"][:80]}
                continue
            q = {"value": _r(e["value"]), "ms_per_step": _r(e["ms_per_step"])}
            rr = e.get("roofline") or {}
            if rr.get("bound") == "hbm":
                # detect-only runs: `frac` = the WHOLE pass (pixel kernel + labelling + gather) by the HBM bytes the counters saw, never the
                # pixel kernel by algorithmic bytes (which passes 1: the background is re-read from L2, not from HBM); null without a PMC
                # summary for the configuration
                q["frac"] = _r(rr.get("whole_detect_pass_frac"), 3) if rr.get("whole_detect_pass_frac") is not None else None
                if rr.get("frac") is not None and rr.get("frac_basis", "").startswith("measured"):
                    q["rows_frac"] = _r(rr["frac"], 3)
                if rr.get("whole_detect_pass_us") is not None:
                    q["whole_detect_pass_us"] = _r(rr["whole_detect_pass_us"], 3)
            elif rr.get("frac") is not None:
                q["frac"] = _r(rr["frac"], 3)
            if rr.get("frac_split_peak") is not None:
                q["frac_split_peak"] = _r(rr["frac_split_peak"], 3)
            if "range_guard" in e:
                q["rerun_crops"] = e["range_guard"].get("rerun_crops_last_step")
            if "host_input" in e:
                # the two legs of the as-deployed path per step: host threads pageable -> pinned ring (+ BGRA -> gray), DMA ring -> HBM;
                # they overlap each other and the kernels, so the step is about the larger of the two plus what does not overlap
                hi_ = e["host_input"]
                nfr = (e.get("ms_per_step") or 0) and (e["value"] * e["ms_per_step"] / 1e3)      # frames per step
                if hi_.get("host_copy_ms_per_frame") is not None:
                    q["copy_ms"] = _r(hi_["host_copy_ms_per_frame"] * nfr, 3)
                if hi_.get("dma_ms_per_frame") is not None:
                    q["dma_ms"] = _r(hi_["dma_ms_per_frame"] * nfr, 3)
            if "spread" in e:
                q["spread"] = _r(e["spread"], 2)
                q["min"], q["max"] = _r(min(e["repetitions"])), _r(max(e["repetitions"]))
            if "ranks_seen" in (e.get("dist") or {}):
                q["ranks_seen"] = e["dist"]["ranks_seen"]
            cfgs[name] = q
        o["configs"] = cfgs
    return o


def fit_line(o):
    """the final line as text, below FINAL_LINE_MAX bytes whatever happened in the secondary runs: optional parts are dropped in a fixed
    order (never the contract's keys, `roofline` or `cpu_baseline`) until it fits; what was dropped is named in `dropped`"""
    order = ["host_input", "stage_us", "roofline_kernels", "configs", "roofline_detect", "dist"]
    dropped = []
    while True:
        line = json.dumps(o, separators=(",", ":"))
        if len(line) < FINAL_LINE_MAX or not order:
            return line
        k = order.pop(0)
        if k == "configs" and "configs" in o:      # first only shorten the entries (errors, extras), then drop the object
            short = {n: ({"value": e["value"]} if "value" in e else {"error": e.get("error", "")[:24]}) for n, e in o["configs"].items()}
            if short != o["configs"]:
                o["configs"] = short
                order.insert(0, "configs")
                continue
        if k in o:
            del o[k]
            dropped.append(k)
            o["dropped"] = dropped


def main():
    argv = sys.argv[1:]
    args = build_parser().parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_spawn(args, argv))
    apply_presets(args)
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.same_gpu:
        # functional check of the N > 1 code path on a 1-GPU box: every rank on device 0, each claiming its own host id so that RCCL
        # accepts them and talks over its socket transport on loopback (slow; never a measurement)
        local = 0
        os.environ.update({"NCCL_HOSTID": f"trexhip-bench-rank{rank}", "NCCL_SOCKET_IFNAME": "lo", "NCCL_IB_DISABLE": "1", "NCCL_P2P_DISABLE": "1",
                           "NCCL_SHM_DISABLE": "1", "NCCL_NET_GDR_LEVEL": "0"})
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    torch.cuda.set_device(local)
    if use_dist:       # one process per GPU; backend "nccl" is RCCL over xGMI on ROCm
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    dev = torch.device("cuda", local)
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s)", file=sys.stderr)
        sys.exit(2)
    env = {"world": world, "rank": rank, "local": local, "use_dist": use_dist, "dev": dev}
    out = measure(args, env)
    sec = None
    if rank == 0 and world == 1 and not args.no_secondary:
        sec = secondary(args, env)
    if dist.is_initialized():
        sys.stdout.flush()
        dist.barrier()                      # every rank is done talking before rank 0 writes its line
        dist.destroy_process_group()
    # the ONE short JSON line is the last thing this process writes: whatever the runtime libraries left in the C stdio buffer (RCCL prints
    # a version banner there) goes out first; the long records (every key of the measurement, the secondary runs) go out BEFORE it, on
    # lines of their own that do not start with "{", and into gpurun_out/ -- the driver keeps only the tail of stdout
    import ctypes
    sys.stdout.flush()
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    if rank == 0:
        emit_detail("bench_detail", out)
        if sec is not None:
            emit_detail("bench_secondary", sec)
        print(fit_line(compact_line(out, sec)), flush=True)


if __name__ == "__main__":
    main()
