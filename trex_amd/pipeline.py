"""Host-side driver of the device pipeline over the C ABI (ctypes): detect -> (posture) -> crops -> identity -> per-blob table.

This is what bench.py times and what tests/test_bench_shape_gpu.py checks: both build the same `Pipeline`, so the benchmarked
schedule (software-pipelined lanes, detect on a high-priority stream, hand-off right after the network) is the tested one.
PyTorch supplies device buffers, streams and torch.distributed only; every kernel is launched through libtrexhip's ABI.
"""
import torch

from . import capi, dist as tdist


class Lane:
    """One context + its buffers + its stream.  Lanes are software-pipelined: while lane A's identity network works on batch i,
    lane B runs the detect stage of batch i+1 and its tables travel to the host."""

    def __init__(self, pl):
        self.pl = pl
        o = pl.opt
        p = capi.default_params(pl.W, pl.H, device=pl.local, max_batch=pl.B, max_blobs=pl.max_blobs, max_pixels=1 << 18, max_runs=32768,
                                pixel_encoding=capi.ENC_RGB8 if pl.rgb else capi.ENC_GRAY)
        self.seg = capi.Segmenter(p)
        self.seg.set_background(pl.bg)
        dev = pl.dev
        if pl.with_cnn:
            self.seg.load_weights(pl.weight_blob)
            self.seg.set_identity_precision({"fp32": 0, "bf16x6": 1, "bf16x3": 2, "fp16x3": 3}[o["cnn_mode"]])
        self.crops = torch.empty((pl.pool, 80, 80, 3) if pl.rgb else (pl.pool, 80, 80), dtype=torch.uint8, device=dev)
        self.probs = torch.empty((pl.pool, pl.classes), dtype=torch.float32, device=dev)
        # per-blob record for rank 0's matcher: header + probabilities, with posture also second moments + normalised midline
        self.table = torch.zeros((pl.rows, pl.rowlen), dtype=torch.int32, device=dev)
        self.table_host = torch.empty((pl.world * pl.rows, pl.rowlen), dtype=torch.int32).pin_memory() if pl.rank == 0 else None
        # N > 1: the library's own communicator gathers every rank's table on rank 0 (trexhip_comm_gather_device_on).  ONE communicator per
        # process, created on the first lane's context and used by every lane on its own stream: the lanes are driven by one host thread in
        # a fixed rotation, so every rank issues its gathers in the same order (what RCCL requires of operations on one communicator)
        self.comm = None
        self.table_all = None
        if pl.use_dist:
            if pl.gather == "library":
                if pl.shared_comm is None:
                    pl.shared_comm = capi.Comm(self.seg, pl.rank, pl.world, pl.next_comm_id())
                self.comm = pl.shared_comm
            self.table_all = torch.zeros((pl.world * pl.rows, pl.rowlen), dtype=torch.int32, device=dev) if pl.rank == 0 else None
            if pl.rank == 0:
                # rank 0 writes its own table straight into slot 0 of the gathered slab: trexhip_comm_gather_device skips the device copy of
                # a rank's own share when send and receive pointers coincide (14 MB per step on the lane's kernel stream otherwise)
                self.table = self.table_all[:pl.rows]
        if pl.with_posture:
            MP = pl.MP
            self.p_outline = torch.empty((pl.pool, MP, 2), dtype=torch.float32, device=dev)
            self.p_segs = torch.empty((pl.pool, MP // 2 + 1, 4), dtype=torch.float32, device=dev)
            self.p_info = torch.empty((pl.pool, 8), dtype=torch.int32, device=dev)
            self.p_mid = torch.empty((pl.pool, 25, 4), dtype=torch.float32, device=dev)
            self.p_minfo = torch.empty((pl.pool, 8), dtype=torch.int32, device=dev)
        self.stream = torch.cuda.Stream(device=dev)   # torch-side copies ride on the same stream as the kernels
        # rank 0's copy of the gathered slab (world x table) to the host has its own stream: behind the gather by an event, off the lane's
        # kernel stream, so that the lane's next posture / crops never queue behind world x 14 MB of PCIe traffic
        self.copy_stream = torch.cuda.Stream(device=dev) if (pl.rank == 0 and pl.use_dist) else None
        self.gathered = torch.cuda.Event() if self.copy_stream is not None else None
        self.seg.set_stream(self.stream.cuda_stream)
        self.hi = torch.cuda.Stream(device=dev, priority=-1) if o["detect_priority"] else None
        self.n = 0
        self.done = torch.cuda.Event()
        self.after = None                       # lane whose identity stage must finish before this lane's starts
        self.res = None                         # BatchResult of the last fetch (pointers into the context's pinned tables)

    def detect(self, frames_ptr):
        pl = self.pl
        if self.hi is not None:
            self.seg.set_stream(self.hi.cuda_stream)
        if pl.host_frames is not None:          # the as-deployed boundary: pageable host tiles, PCIe inside the step (upload.hip)
            if pl.bgra_in:
                self.seg.segment_color_host(pl.host_frames)
            else:
                self.seg.segment_host(pl.host_frames)
        elif pl.bgra_in:
            self.seg.segment_color_device(frames_ptr, pl.B, 4)
        else:
            self.seg.segment_device(frames_ptr, pl.B)

    def identify(self, step_idx):
        pl, seg = self.pl, self.seg
        o = pl.opt
        res = self.res = seg.fetch_raw()        # waits for detect; blob/run/pixel tables now on this rank's host (pinned)
        n = self.n = int(res.total_blobs)
        assert n <= pl.rows, "identity table too small"
        if self.hi is not None:
            seg.set_stream(self.stream.cuda_stream)
        if pl.with_posture and n:
            seg.posture_device(n, self.p_outline.data_ptr(), self.p_segs.data_ptr(), self.p_info.data_ptr(), max_points=pl.MP)
            seg.midline_device(n, pl.MP, self.p_info.data_ptr(), self.p_segs.data_ptr(), self.p_mid.data_ptr(), self.p_minfo.data_ptr())
        if pl.with_cnn:
            if n:
                if o["normalize"] == "posture":
                    seg.crops_posture_device(self.crops.data_ptr(), n, self.p_minfo.data_ptr())
                else:
                    seg.crops_device(self.crops.data_ptr(), n, normalization=1 if o["normalize"] == "moments" else 0)
                # one identity network on the matrix cores at a time; posture / crops above and the table hand-off below overlap the other
                # lane's network.  Not for batches of a few frames (TRex's default is ONE frame per call): their network is a chain of small
                # launches that leaves most of the chip idle, and two lanes' chains side by side are what keeps it busy
                if self.after is not None and n > 1024:
                    self.stream.wait_event(self.after.done)
                seg.identify_device(self.crops.data_ptr(), n, self.probs.data_ptr())
            self.done.record(self.stream)
            # per-blob identity table -> (gathered over RCCL/xGMI when N > 1) -> rank 0's host, for the sequential matcher
            frame_base, _ = tdist.step_frames(step_idx, pl.rank, pl.world, pl.B)
            if pl.with_posture:
                seg.export_id_table_ex(self.probs.data_ptr(), n, pl.classes, frame_base, self.table.data_ptr(), pl.rows,
                                       self.p_mid.data_ptr() if n else 0, self.p_minfo.data_ptr() if n else 0, 25)
            else:
                seg.export_id_table(self.probs.data_ptr(), n, pl.classes, frame_base, self.table.data_ptr(), pl.rows)
            if pl.use_dist and self.comm is not None:   # grouped send / recv on the context's stream, behind the table kernel
                self.comm.gather_device(self.table.data_ptr(), self.table.numel() * 4, self.table_all.data_ptr() if pl.rank == 0 else 0, seg=seg)
            elif pl.use_dist:                   # gather="torch": the same exchange through torch.distributed's RCCL communicator
                import torch.distributed as dist
                with torch.cuda.stream(self.stream):
                    dist.gather(self.table, list(self.table_all.view(pl.world, pl.rows, pl.rowlen).unbind(0)) if pl.rank == 0 else None, dst=0)
            if pl.rank == 0 and self.copy_stream is not None:
                self.gathered.record(self.stream)
                self.copy_stream.wait_event(self.gathered)
                with torch.cuda.stream(self.copy_stream):
                    self.table_host.copy_(self.table_all, non_blocking=True)
            elif pl.rank == 0:
                with torch.cuda.stream(self.stream):
                    self.table_host.copy_(self.table, non_blocking=True)
        else:
            self.done.record(self.stream)

    def drain(self):
        self.stream.synchronize()
        if self.copy_stream is not None:
            self.copy_stream.synchronize()


class Pipeline:
    def __init__(self, W, H, n_ind, B, classes, bg, weight_blob=None, *, local=0, rank=0, world=1, use_dist=False, comm_ids=None,
                 with_cnn=True, with_posture=False, normalize="none", rgb=False, bgra_in=False, cnn_mode="fp16x3",
                 lanes=2, pipeline=True, detect_priority=True, host_frames=None, gather="library"):
        self.W, self.H, self.B, self.classes, self.bg = W, H, B, classes, bg
        self.local, self.rank, self.world, self.use_dist = local, rank, world, use_dist
        assert gather in ("library", "torch")
        self.gather = gather                      # who owns the RCCL communicator of the table gather: libtrexhip (default) or torch.distributed
        self._comm_ids = list(comm_ids or [])     # the ncclUniqueId (128 bytes, made by rank 0) of the process group's one communicator
        self.dev = torch.device("cuda", local)
        self.with_cnn, self.with_posture, self.rgb, self.bgra_in = with_cnn, with_posture or normalize == "posture", rgb, bgra_in or rgb
        self.weight_blob = weight_blob
        self.host_frames = host_frames            # list of B numpy frames ([H,W] gray or [H,W,4] BGRA) in pageable host memory, or None
        self.opt = {"normalize": normalize, "cnn_mode": cnn_mode, "detect_priority": detect_priority}
        self.max_blobs = 4 * n_ind
        self.pool = B * self.max_blobs
        self.rows = B * n_ind * 5 // 4             # fixed table rows per rank per step (a gather needs equal sizes)
        self.MP = 256
        self.rowlen = (tdist.HDR_EX + classes + 3 * 25) if self.with_posture else (tdist.HDR + classes)
        self.shared_comm = None
        self.lanes = [Lane(self) for _ in range(max(2, lanes))] if pipeline else [Lane(self)]
        if len(self.lanes) > 1:
            for k, ln in enumerate(self.lanes):
                ln.after = self.lanes[(k - 1) % len(self.lanes)]
                ln.done.record(ln.stream)
        torch.cuda.synchronize()

    def next_comm_id(self):
        if self.world == 1:
            return None
        assert self._comm_ids, "Pipeline(use_dist=True, world > 1) needs comm_ids = [trexhip_comm_unique_id() of rank 0]"
        return self._comm_ids.pop(0)

    def run(self, k, frames_ptr, on_batch=None):
        """k steps = k batches through detect -> (posture) -> crops -> identity -> table on the host.  frames_ptr: device pointer of
        the B resident frames (or a callable step -> pointer).  on_batch(step, lane) is called once batch `step` is complete
        (its lane drained), before the lane is reused."""
        lanes = self.lanes
        L = len(lanes)
        D = max(1, L - 1)                           # detect runs D batches ahead of the identity network
        fp = frames_ptr if callable(frames_ptr) else (lambda i: frames_ptr)
        owner = {}

        def finish(ln):
            ln.drain()
            if on_batch is not None and id(ln) in owner:
                on_batch(owner.pop(id(ln)), ln)

        for i in range(min(D, k)):
            lanes[i % L].detect(fp(i))
        for i in range(k):
            cur = lanes[i % L]
            if not self.with_cnn and L > 1 and i + D < k:
                # no network to keep fed: issue detect(i+D) BEFORE blocking on the tables of batch i, so that their copy to the host
                # overlaps the next pixel pass (with the network the order below keeps the matrix cores' queue non-empty instead)
                nxt = lanes[(i + D) % L]
                finish(nxt)
                nxt.detect(fp(i + D))
                cur.identify(i); owner[id(cur)] = i
                continue
            cur.identify(i); owner[id(cur)] = i     # enqueue everything downstream of detect(i)
            if i + D < k:
                nxt = lanes[(i + D) % L]
                if L > 1:
                    finish(nxt)                     # its previous batch (i+D-L) is complete: table_host consumed by the matcher
                else:
                    finish(cur)
                nxt.detect(fp(i + D))               # detect(i+D) overlaps the identity network of batches i-1 / i
        for ln in lanes:
            finish(ln)
        return lanes[(k - 1) % L].n

    def close(self):
        if self.shared_comm is not None:
            self.shared_comm.close()
            self.shared_comm = None
        for ln in self.lanes:
            ln.seg.close()
