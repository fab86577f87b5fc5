// upload.hip -- host tiles -> HBM for the host-pointer entry points (trexhip_segment, trexhip_segment_color): the as-deployed boundary,
// where TRex hands over pooled BGR / BGRA tile images in pageable host memory (BackgroundSubtraction.cpp:146-180).
//
// A frame goes pageable -> pinned ring slot (host threads, rows split between them) -> HBM (one async DMA per frame on a copy stream).
// The ring has UP_SLOTS slots: while the DMA engine moves frame i, the host threads already fill the slot of frame i+1, and the compute
// stream reduces frame i-1 to gray (the caller enqueues that behind the frame's event).  The segment kernels then run over the whole
// batch.  PCIe Gen5 x16 (63 GB/s spec) bounds this path; the timings of the two legs are kept per context (trexhip_profile_read:
// TREXHIP_STAGE_UPLOAD_COPY = host milliseconds spent filling slots, TREXHIP_STAGE_UPLOAD_DMA = DMA milliseconds from HIP events).
#include "internal.h"
#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <functional>
#include <mutex>
#include <thread>
#include <cstdio>
#include <pthread.h>
#include <sched.h>

namespace trexhip {

// CPUs of the NUMA node the calling thread runs on (empty when /sys does not say).  The copy threads are pinned there, one CPU each:
// the tiles were written by the caller's side of the machine, and a copy thread on the other socket reads 16 MB per frame across the
// inter-socket link (measured on the two-socket GPU box: the copy leg of a 256-frame upload then takes 80-95 ms instead of 21)
static std::vector<int> local_node_cpus() {
    std::vector<int> out;
    const int cpu = sched_getcpu();
    if (cpu < 0) return out;
    for (int node = 0; node < 64; ++node) {
        char path[96];
        std::snprintf(path, sizeof(path), "/sys/devices/system/node/node%d/cpulist", node);
        FILE* f = std::fopen(path, "r");
        if (!f) break;
        std::vector<int> cpus;
        int a = 0, b = 0;
        bool mine = false;
        for (;;) {
            if (std::fscanf(f, "%d", &a) != 1) break;
            b = a;
            int ch = std::fgetc(f);
            if (ch == '-') { if (std::fscanf(f, "%d", &b) != 1) break; ch = std::fgetc(f); }
            for (int c = a; c <= b; ++c) { cpus.push_back(c); mine |= c == cpu; }
            if (ch != ',') break;
        }
        std::fclose(f);
        if (mine) { out = cpus; break; }
    }
    cpu_set_t allowed;
    if (!out.empty() && sched_getaffinity(0, sizeof(allowed), &allowed) == 0) {      // never outside what the process may use (cgroup / taskset)
        std::vector<int> ok;
        for (int c : out) if (CPU_ISSET(c, &allowed)) ok.push_back(c);
        out.swap(ok);
    }
    return out;
}

// a few persistent host threads that copy row ranges; one job at a time (the ctx is not re-entrant)
struct CopyPool {
    std::vector<std::thread> workers;
    std::mutex mu;
    std::condition_variable cv_go, cv_done;
    std::function<void(int, int)> job;      // (worker index, worker count)
    uint64_t generation = 0;
    int pending = 0;
    bool stop = false;

    explicit CopyPool(int n) {
        static const bool pin = [] { const char* e = std::getenv("TREXHIP_UPLOAD_PIN"); return !(e && std::atoi(e) == 0); }();
        const std::vector<int> cpus = pin ? local_node_cpus() : std::vector<int>();
        for (int w = 0; w < n; ++w)
            workers.emplace_back([this, w, n, cpus]() {
                if (!cpus.empty()) {
                    // every worker may run on any CPU of the caller's NUMA node (one mask, not one CPU each: the pools of several contexts, lanes
                    // or ranks on the same node would otherwise all stack on the node's first CPUs and leave the rest idle); TREXHIP_UPLOAD_PIN=0
                    // leaves the placement to the scheduler
                    cpu_set_t set;
                    CPU_ZERO(&set);
                    for (int c : cpus) CPU_SET(c, &set);
                    (void)pthread_setaffinity_np(pthread_self(), sizeof(set), &set);
                }
                uint64_t seen = 0;
                for (;;) {
                    std::function<void(int, int)> j;
                    {
                        std::unique_lock<std::mutex> g(mu);
                        cv_go.wait(g, [&] { return stop || generation != seen; });
                        if (stop) return;
                        seen = generation;
                        j = job;
                    }
                    j(w, n);                                           // (the calling thread does not copy: it issues the DMAs meanwhile)
                    {
                        std::lock_guard<std::mutex> g(mu);
                        if (--pending == 0) cv_done.notify_one();
                    }
                }
            });
    }
    ~CopyPool() {
        { std::lock_guard<std::mutex> g(mu); stop = true; }
        cv_go.notify_all();
        for (auto& t : workers) t.join();
    }
    void start(const std::function<void(int, int)>& f) {
        {
            std::lock_guard<std::mutex> g(mu);
            job = f; pending = (int)workers.size(); ++generation;
        }
        cv_go.notify_all();
    }
    void wait() {
        std::unique_lock<std::mutex> g(mu);
        cv_done.wait(g, [&] { return pending == 0; });
    }
};

void upload_free(trexhip_ctx* ctx) {
    Uploader& u = ctx->up;
    delete static_cast<CopyPool*>(u.pool); u.pool = nullptr;
    if (u.ring) { (void)hipHostFree(u.ring); u.ring = nullptr; }
    for (int s = 0; s < UP_SLOTS; ++s) {
        if (u.ev_done[s]) { (void)hipEventDestroy(u.ev_done[s]); u.ev_done[s] = nullptr; }
        if (u.ev_start[s]) { (void)hipEventDestroy(u.ev_start[s]); u.ev_start[s] = nullptr; }
    }
    if (u.copy_stream) { (void)hipStreamDestroy(u.copy_stream); u.copy_stream = nullptr; }
    u.slot_bytes = 0;
}

static int upload_prepare(trexhip_ctx* ctx, size_t frame_bytes) {   // frame_bytes = bytes of one ring slot (a chunk of whole frames)
    Uploader& u = ctx->up;
    if (!u.copy_stream) TH_CHECK_HIP(hipStreamCreateWithFlags(&u.copy_stream, hipStreamNonBlocking));
    for (int s = 0; s < UP_SLOTS; ++s) {
        if (!u.ev_done[s]) TH_CHECK_HIP(hipEventCreate(&u.ev_done[s]));
        if (!u.ev_start[s]) TH_CHECK_HIP(hipEventCreate(&u.ev_start[s]));
    }
    if (u.slot_bytes < frame_bytes) {
        if (u.ring) { TH_CHECK_HIP(hipStreamSynchronize(u.copy_stream)); (void)hipHostFree(u.ring); u.ring = nullptr; }
        TH_CHECK_HIP(hipHostMalloc(reinterpret_cast<void**>(&u.ring), frame_bytes * UP_SLOTS, hipHostMallocDefault));
        u.slot_bytes = frame_bytes;
        for (int s = 0; s < UP_SLOTS; ++s) u.busy[s] = false;
    }
    if (!u.pool) {
        int n = 0;
        if (const char* e = std::getenv("TREXHIP_UPLOAD_THREADS")) n = std::atoi(e);
        if (n <= 0) { const unsigned hc = std::thread::hardware_concurrency(); n = hc >= 128 ? 32 : hc >= 64 ? 16 : (hc >= 32 ? 10 : (hc >= 16 ? 6 : (hc >= 8 ? 4 : (hc >= 4 ? 2 : 1)))); }   // 6 -> 16 threads on the 256-thread box: 4.5 k -> 8 k BGRA frames/s
        u.pool = new CopyPool(n);
    }
    return TREXHIP_OK;
}

// n frames of `rows` rows x `row_bytes` bytes (source row pitch `stride`) -> d_dst, frame after frame, in chunks of whole frames of
// about UP_CHUNK_BYTES (one DMA and one event pair per chunk: small transfers do not reach the link rate).  after_chunk(first, count)
// is called once the chunk's DMA is enqueued, with ctx->stream already ordered behind it (the caller launches the device work of
// those frames there).
extern "C" void trexhip_host_reduce_row(const uint8_t* src, uint8_t* dst, size_t npix, int channels, int color_channel);   // hostcvt.cpp

// reduce_channels = 3 / 4: the upload threads reduce the colour rows to gray (or pick color_channel) on their way into the pinned ring
// (hostcvt.cpp): row_bytes is then the source row, what lands in d_dst is rows x (row_bytes / reduce_channels) bytes per frame.
int upload_frames(trexhip_ctx* ctx, const uint8_t* const* frames, int n, size_t rows, size_t row_bytes, size_t stride, uint8_t* d_dst,
                  const std::function<int(int, int)>& after_chunk, int reduce_channels, int color_channel) {
    const size_t out_row = reduce_channels ? row_bytes / (size_t)reduce_channels : row_bytes;
    const size_t frame_bytes = rows * out_row;
    const int per = (int)std::max<size_t>(1, UP_CHUNK_BYTES / frame_bytes);
    int rc = upload_prepare(ctx, frame_bytes * (size_t)per);
    if (rc) return rc;
    Uploader& u = ctx->up;
    CopyPool* pool = static_cast<CopyPool*>(u.pool);
    // the device buffer may still be read by the previous batch
    TH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    double copy_ms = 0.0;
    // chunk c is filled by the copy threads while the calling thread enqueues the DMA (and the device work) of chunk c - 1: two event
    // records, the copy, a stream wait and the caller's launches per chunk cost ~0.2 ms of API time, which used to sit between the fills
    auto issue = [&](int i0, int cnt, int s) -> int {
        uint8_t* slot = u.ring + (size_t)s * u.slot_bytes;
        TH_CHECK_HIP(hipEventRecord(u.ev_start[s], u.copy_stream));
        TH_CHECK_HIP(hipMemcpyAsync(d_dst + (size_t)i0 * frame_bytes, slot, frame_bytes * (size_t)cnt, hipMemcpyHostToDevice, u.copy_stream));
        TH_CHECK_HIP(hipEventRecord(u.ev_done[s], u.copy_stream));
        u.busy[s] = true; u.frames_in[s] = cnt;
        TH_CHECK_HIP(hipStreamWaitEvent(ctx->stream, u.ev_done[s], 0));
        if (after_chunk) { const int rc2 = after_chunk(i0, cnt); if (rc2) return rc2; }
        return TREXHIP_OK;
    };
    int chunk = 0, prev_i0 = -1, prev_cnt = 0, prev_s = 0;
    const auto t_all = std::chrono::steady_clock::now();
    for (int i0 = 0; i0 < n; i0 += per, ++chunk) {
        const int cnt = std::min(per, n - i0);
        const int s = chunk % UP_SLOTS;
        if (u.busy[s]) {                                              // the DMA that last read this slot must be done; its time is the DMA leg
            TH_CHECK_HIP(hipEventSynchronize(u.ev_done[s]));
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, u.ev_start[s], u.ev_done[s]) == hipSuccess) { u.dma_ms += ms; u.dma_n += u.frames_in[s]; }
            u.busy[s] = false;
        }
        uint8_t* slot = u.ring + (size_t)s * u.slot_bytes;
        const size_t total_rows = rows * (size_t)cnt;
        pool->start([=](int w, int nw) {                              // the rows of the whole chunk are dealt to the threads
            const size_t r0 = total_rows * (size_t)w / (size_t)nw, r1 = total_rows * (size_t)(w + 1) / (size_t)nw;
            size_t r = r0;
            while (r < r1) {
                const size_t f = r / rows, y = r - f * rows;
                const size_t run = std::min(rows - y, r1 - r);       // rows of this frame in my range
                const uint8_t* src = frames[i0 + (int)f] + y * stride;
                uint8_t* dst = slot + f * frame_bytes + y * out_row;
                if (reduce_channels) {
                    if (stride == row_bytes) trexhip_host_reduce_row(src, dst, run * out_row, reduce_channels, color_channel);
                    else for (size_t k = 0; k < run; ++k) trexhip_host_reduce_row(src + k * stride, dst + k * out_row, out_row, reduce_channels, color_channel);
                }
                else if (stride == row_bytes) std::memcpy(dst, src, run * row_bytes);
                else for (size_t k = 0; k < run; ++k) std::memcpy(dst + k * row_bytes, src + k * stride, row_bytes);
                r += run;
            }
        });
        int rc1 = TREXHIP_OK;
        if (prev_i0 >= 0) rc1 = issue(prev_i0, prev_cnt, prev_s);
        pool->wait();
        if (rc1) return rc1;
        prev_i0 = i0; prev_cnt = cnt; prev_s = s;
    }
    if (prev_i0 >= 0) { rc = issue(prev_i0, prev_cnt, prev_s); if (rc) return rc; }
    copy_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_all).count();      // the whole host leg of the call
    u.copy_ms += copy_ms; u.copy_n += n;
    return TREXHIP_OK;
}

}  // namespace trexhip
