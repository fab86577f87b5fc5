// C++ test of the host adapter (trex_amd/host) through the C ABI, checked against the CPU oracle.
// Reads like the reference's own plumbing test (Application/Tests/test_segmenter.cpp:95-235: 64x48
// frames with an 8x8 white square moving 3 px/frame) but also checks blob CONTENTS bit for bit.
#include <cassert>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <random>
#include "../../trex_amd/host/HipBackgroundSubtraction.h"
#include "../../trex_amd/host/HipVINetwork.h"
#include "../../trex_amd/host/HipPosture.h"
#include "../../trex_amd/host/HipSplitBlob.h"
#include "../../trex_amd/host/HipHistorySplit.h"
#include <cmath>
#include "../../oracle/trex_oracle.h"

using namespace track;

#define CHECK(x) do { if (!(x)) { std::fprintf(stderr, "CHECK failed %s:%d: %s\n", __FILE__, __LINE__, #x); std::exit(1); } } while (0)

static cmn::Image::Ptr gray_to_bgr(const std::vector<uint8_t>& g, int W, int H, int ch, std::mt19937& rng, bool same_channels) {
    auto im = cmn::Image::Make(H, W, ch);
    for (int i = 0; i < W * H; ++i)
        for (int c = 0; c < ch; ++c) im->data()[i * ch + c] = same_channels || c == 3 ? g[i] : (uint8_t)(rng() & 0xff);
    return im;
}
static std::vector<uint8_t> bgr2gray(const cmn::Image& im) {
    std::vector<uint8_t> g((size_t)im.rows * im.cols);
    for (size_t i = 0; i < g.size(); ++i) {
        const uint8_t* p = im.data() + i * im.dims;
        g[i] = (uint8_t)((p[0] * 1868u + p[1] * 9617u + p[2] * 4899u + 8192u) >> 14);
    }
    return g;
}

static void compare_with_oracle(const pv::Frame& frame, const std::vector<uint8_t>& gray, const std::vector<uint8_t>& bg, int W, int H,
                                const oracle_params& op) {
    oracle_frame* of = oracle_segment(gray.data(), bg.data(), &op);
    int32_t nb, nr, np;
    oracle_frame_counts(of, &nb, &nr, &np);
    std::vector<oracle_blob> blobs(nb); std::vector<oracle_run> runs(nr); std::vector<uint8_t> px(np);
    oracle_frame_copy(of, blobs.data(), runs.data(), px.data());
    oracle_frame_free(of);
    CHECK(frame.n() == nb);
    CHECK((int)frame.mask().size() == nb && (int)frame.pixels().size() == nb);
    for (int b = 0; b < nb; ++b) {
        const auto& lines = *frame.mask()[b];
        CHECK(lines.size() == blobs[b].n_runs);
        for (uint32_t j = 0; j < blobs[b].n_runs; ++j) {
            const oracle_run& r = runs[blobs[b].run_begin + j];
            CHECK(lines[j] == cmn::HorizontalLine(r.y, r.x0, r.x1));
            if (j) CHECK(lines[j - 1] < lines[j]);                     // pv.cpp:505-508 ordering invariant
        }
        const auto& p = *frame.pixels()[b];
        CHECK(p.size() == blobs[b].n_pixels);                          // pv.cpp:512
        CHECK(std::memcmp(p.data(), px.data() + blobs[b].pix_begin, p.size()) == 0);
    }
    (void)W; (void)H;
}

int main(int argc, char** argv) {
    std::mt19937 rng(7);
    const int W = 64, H = 48;
    HipBackgroundSubtraction::Settings s;
    s.max_batch = 4;
    const auto type = detect::ObjectDetectionType::hip_background_subtraction;
    HipBackgroundSubtraction::register_hip_backend(type, s, W, H);
    const detect::BackendHooks* hooks = detect::backend(type);
    CHECK(hooks && hooks->init && hooks->apply && hooks->set_background && hooks->fps && hooks->deinit);
    hooks->init();
    oracle_params op; std::memset(&op, 0, sizeof(op));
    op.width = W; op.height = H; op.threshold = 15; op.threshold_maximum = 255; op.enable_difference = 1;
    op.absolute_difference = 1; op.inclusive = 1; op.zero_is_background = 1; op.connectivity = 8; op.closing_size = 3; op.cm_per_pixel = 1.0;

    // --- no background yet: the future must carry an exception (BackgroundSubtraction.cpp:58-73 waits; we refuse) ---
    {
        TileImage t; t.images.push_back(cmn::Image::Make(H, W, 3));
        auto f = HipBackgroundSubtraction::apply(std::move(t));
        bool threw = false;
        try { f.get(); } catch (const std::exception&) { threw = true; }
        CHECK(threw);
    }
    // the backend registers its pipeline PAUSED (BackgroundSubtraction.cpp:50-56): tiles enqueued through the manager -- what
    // Detection::apply(TileImage&&) does (Detection.cpp:124-146) -- wait for the first background and are then served in order
    {
        auto& mgr = detect::pipeline_manager(type);
        CHECK(mgr.is_paused());
        std::vector<std::future<SegmentationData>> futs;
        for (int k = 0; k < 3; ++k) {
            std::vector<uint8_t> g(W * H, 100);
            for (int x = 4 + k; x < 12 + k; ++x) g[7 * W + x] = 10;
            TileImage tile; tile.images.push_back(gray_to_bgr(g, W, H, 3, rng, true));
            tile.promise = std::make_unique<std::promise<SegmentationData>>();
            futs.push_back(tile.promise->get_future());
            mgr.enqueue(std::move(tile));
        }
        CHECK(mgr.pending() == 3);
        CHECK(futs[0].wait_for(std::chrono::milliseconds(0)) != std::future_status::ready);
        std::vector<uint8_t> bgq(W * H, 100);
        auto b = cmn::Image::Make(H, W, 1); std::memcpy(b->data(), bgq.data(), bgq.size());
        hooks->set_background(b);                                       // un-pauses the manager (BackgroundSubtraction.cpp:86-99)
        CHECK(!mgr.is_paused() && mgr.pending() == 0);
        for (int k = 0; k < 3; ++k) {
            SegmentationData d = futs[k].get();
            CHECK(d.frame.n() == 1);
            CHECK((*d.frame.mask()[0])[0] == cmn::HorizontalLine(7, 4 + k, 11 + k));
        }
    }
    std::vector<uint8_t> bg(W * H, 0);
    { auto b = cmn::Image::Make(H, W, 1); std::memcpy(b->data(), bg.data(), bg.size()); hooks->set_background(b); }

    // --- the reference's moving square, handed over as BGR tiles, 4 tiles per apply() ---
    const size_t pool_before = buffers::TileBuffers::get().size();
    int callbacks = 0;
    for (int t0 = 0; t0 < 12; t0 += 4) {
        std::vector<TileImage> tiles;
        std::vector<std::future<SegmentationData>> futs;
        std::vector<std::vector<uint8_t>> grays;
        for (int t = t0; t < t0 + 4; ++t) {
            std::vector<uint8_t> g(W * H, 0);
            for (int y = 20; y < 28; ++y) for (int x = 3 * t; x < 3 * t + 8; ++x) g[y * W + x] = 255;
            TileImage tile;
            tile.images.push_back(gray_to_bgr(g, W, H, 3 + (t & 1), rng, true));
            tile.data.image = cmn::Image::Make(H, W, 3);
            tile.data.image->set_to((uint8_t)t);
            tile.promise = std::make_unique<std::promise<SegmentationData>>();
            futs.push_back(tile.promise->get_future());
            tile.callback = [&callbacks]() { ++callbacks; };
            grays.push_back(g);
            tiles.emplace_back(std::move(tile));
        }
        // mixed BGR/BGRA in one batch is refused; run them one tile at a time instead when channel counts differ
        for (size_t k = 0; k < tiles.size(); ++k) {
            std::vector<TileImage> one; one.emplace_back(std::move(tiles[k]));
            hooks->apply(std::move(one));
        }
        for (size_t k = 0; k < futs.size(); ++k) {
            SegmentationData d = futs[k].get();
            CHECK(d.frame.n() == 1);
            CHECK(d.frame.pixels()[0]->size() == 64);
            CHECK((*d.frame.mask()[0])[0] == cmn::HorizontalLine(20, 3 * (t0 + k), 3 * (t0 + k) + 7));
            CHECK(d.image && d.image->data()[0] == (uint8_t)(t0 + k));      // original frame untouched
            compare_with_oracle(d.frame, grays[k], bg, W, H, op);
        }
    }
    CHECK(callbacks == 12);
    CHECK(buffers::TileBuffers::get().size() == pool_before + 12);          // every tile image handed back to the pool
    CHECK(hooks->fps() > 0);

    // --- a whole batch of random colour tiles in ONE apply(): contents vs oracle through cvtColor ---
    {
        std::vector<uint8_t> bg2(W * H);
        for (auto& v : bg2) v = 100 + (rng() % 20);
        auto b = cmn::Image::Make(H, W, 1); std::memcpy(b->data(), bg2.data(), bg2.size()); hooks->set_background(b);
        std::vector<TileImage> tiles; std::vector<std::future<SegmentationData>> futs; std::vector<std::vector<uint8_t>> grays;
        for (int k = 0; k < 4; ++k) {
            std::vector<uint8_t> dummy(W * H, 0);
            TileImage tile; tile.images.push_back(gray_to_bgr(dummy, W, H, 3, rng, false));
            for (int i = 0; i < 40; ++i) {   // dark bars
                int y = rng() % H, x = rng() % (W - 10), l = 1 + rng() % 9;
                for (int q = 0; q < l; ++q) for (int c = 0; c < 3; ++c) tile.images[0]->data()[(y * W + x + q) * 3 + c] = rng() % 30;
            }
            grays.push_back(bgr2gray(*tile.images[0]));
            tile.promise = std::make_unique<std::promise<SegmentationData>>();
            futs.push_back(tile.promise->get_future());
            tiles.emplace_back(std::move(tile));
        }
        hooks->apply(std::move(tiles));
        for (int k = 0; k < 4; ++k) { SegmentationData d = futs[k].get(); compare_with_oracle(d.frame, grays[k], bg2, W, H, op); }
    }
    // --- wrong channel count -> exception through the future (BackgroundSubtraction.cpp:179) ---
    {
        TileImage t; t.images.push_back(cmn::Image::Make(H, W, 1));
        auto f = HipBackgroundSubtraction::apply(std::move(t));
        bool threw = false;
        try { f.get(); } catch (const std::exception&) { threw = true; }
        CHECK(threw);
    }
    // --- meta_encoding changed without re-initialising raises through the future like the reference's "Invalid image mode" (:188) ---
    {
        auto& st = HipBackgroundSubtraction::settings();
        const auto keep = st.meta_encoding;
        st.meta_encoding = cmn::meta_encoding_t::rgb8;
        TileImage t; t.images.push_back(cmn::Image::Make(H, W, 3));
        auto f = HipBackgroundSubtraction::apply(std::move(t));
        bool threw = false;
        try { f.get(); } catch (const std::exception&) { threw = true; }
        CHECK(threw);
        st.meta_encoding = keep;
    }
    // --- rgb8: the same lines, 3 bytes per pixel in memory order, frame encoding rgb8 (pv.cpp:512-517) ---
    {
        HipBackgroundSubtraction::Settings s2 = HipBackgroundSubtraction::settings();
        s2.meta_encoding = cmn::meta_encoding_t::rgb8;
        HipBackgroundSubtraction::init(s2, W, H);
        std::vector<uint8_t> bg2(W * H, 120);
        auto b = cmn::Image::Make(H, W, 1); std::memcpy(b->data(), bg2.data(), bg2.size());
        HipBackgroundSubtraction::set_background(gray_to_bgr(bg2, W, H, 3, rng, true));      // a colour average is reduced to grey like the frames
        std::vector<uint8_t> dummy(W * H, 120);
        TileImage tile; tile.images.push_back(gray_to_bgr(dummy, W, H, 4, rng, true));
        uint8_t* im = tile.images[0]->data();
        for (int x = 10; x < 20; ++x) { uint8_t* q = im + (size_t)(5 * W + x) * 4; q[0] = (uint8_t)x; q[1] = (uint8_t)(2 * x); q[2] = (uint8_t)(3 * x); }
        auto f = HipBackgroundSubtraction::apply(std::move(tile));
        SegmentationData d = f.get();
        CHECK(d.frame.encoding() == cmn::meta_encoding_t::rgb8);
        CHECK(d.frame.n() == 1);
        CHECK((*d.frame.mask()[0])[0] == cmn::HorizontalLine(5, 10, 19));
        const auto& px = *d.frame.pixels()[0];
        CHECK(px.size() == 30);
        for (int x = 10; x < 20; ++x) CHECK(px[3 * (x - 10)] == x && px[3 * (x - 10) + 1] == 2 * x && px[3 * (x - 10) + 2] == 3 * x);
        s2.meta_encoding = cmn::meta_encoding_t::gray;
        HipBackgroundSubtraction::init(s2, W, H);
        HipBackgroundSubtraction::set_background(b);
    }
    // --- a setting of the reference this backend does not implement is refused at init(), not silently ignored ---
    {
        const HipBackgroundSubtraction::Settings keep = HipBackgroundSubtraction::settings();
        for (int which = 0; which < 5; ++which) {
            HipBackgroundSubtraction::Settings s3 = keep;
            if (which == 0) s3.image_adjust = true;
            if (which == 1) s3.blur_difference = 3;
            if (which == 2) s3.equalize_histogram = true;
            if (which == 3) s3.correct_luminance = true;
            if (which == 4) s3.use_adaptive_threshold = true;
            bool threw = false; std::string what;
            try { HipBackgroundSubtraction::init(s3, W, H); } catch (const std::exception& e) { threw = true; what = e.what(); }
            CHECK(threw && what.find("not implemented") != std::string::npos);
        }
        HipBackgroundSubtraction::init(keep, W, H);
        std::vector<uint8_t> bg3(W * H, 120);
        auto b3 = cmn::Image::Make(H, W, 1); std::memcpy(b3->data(), bg3.data(), bg3.size());
        HipBackgroundSubtraction::set_background(b3);
    }
    // --- a tile of the wrong size fails through the future instead of reading past the image ---
    {
        TileImage t; t.images.push_back(cmn::Image::Make(H / 2, W, 3));
        auto f = HipBackgroundSubtraction::apply(std::move(t));
        bool threw = false;
        try { f.get(); } catch (const std::exception&) { threw = true; }
        CHECK(threw);
    }
    // --- settings changed between two apply() calls take effect on the second one, as in the reference, which re-reads cm_per_pixel /
    //     detect_size_filter (BackgroundSubtraction.cpp:137-143) and its thresholds on every call ---
    {
        auto& st = HipBackgroundSubtraction::settings();
        const HipBackgroundSubtraction::Settings keep = st;
        std::vector<uint8_t> g(W * H, 120);                                   // background is 120 everywhere (set above)
        for (int x = 2; x < 6; ++x) g[3 * W + x] = 10;                        // 4 px
        for (int y = 10; y < 13; ++y) for (int x = 2; x < 12; ++x) g[y * W + x] = 10;      // 30 px
        for (int y = 20; y < 30; ++y) for (int x = 2; x < 22; ++x) g[y * W + x] = 10;      // 200 px
        for (int x = 40; x < 48; ++x) g[40 * W + x] = 105;                    // 8 px whose difference is exactly detect_threshold = 15
        auto count = [&](oracle_params o) {
            TileImage tile; tile.images.push_back(gray_to_bgr(g, W, H, 3, rng, true));
            SegmentationData d = HipBackgroundSubtraction::apply(std::move(tile)).get();
            std::vector<uint8_t> bgl(W * H, 120);
            compare_with_oracle(d.frame, g, bgl, W, H, o);
            return (int)d.frame.n();
        };
        oracle_params o = op;
        CHECK(count(o) == 4);
        st.detect_size_filter = {{10.0, 100.0}};                              // [start, end) in cm^2, cm_per_pixel 1: only the 30 px blob
        o.n_ranges = 1; o.ranges[0] = 10; o.ranges[1] = 100;
        CHECK(count(o) == 1);
        st.cm_per_pixel = 0.5;                                                // 30 px -> 7.5 cm^2 (out), 200 px -> 50 cm^2 (in)
        o.cm_per_pixel = 0.5;
        CHECK(count(o) == 1);
        st.detect_size_filter = {{1.5, 10.0}, {40.0, 60.0}};                  // 8 px = 2, 30 px = 7.5, 200 px = 50 cm^2
        o.n_ranges = 2; o.ranges[0] = 1.5; o.ranges[1] = 10; o.ranges[2] = 40; o.ranges[3] = 60;
        CHECK(count(o) == 3);
        st.inclusive = false;                                                 // strict: the difference-15 line goes
        o.inclusive = 0;
        CHECK(count(o) == 2);
        st.inclusive = true; st.detect_threshold = 16;
        o.inclusive = 1; o.threshold = 16;
        CHECK(count(o) == 2);
        st.detect_threshold = 15; st.threshold_maximum = 100;                 // cv::inRange [15, 100]: only the difference-15 line is left
        o.threshold = 15; o.threshold_maximum = 100;
        CHECK(count(o) == 1);
        st.detect_size_filter.assign(9, {1.0, 2.0});                          // more ranges than the device holds: refused loudly, per batch
        {
            TileImage tile; tile.images.push_back(gray_to_bgr(g, W, H, 3, rng, true));
            auto f = HipBackgroundSubtraction::apply(std::move(tile));
            bool threw = false; std::string what;
            try { f.get(); } catch (const std::exception& e) { threw = true; what = e.what(); }
            CHECK(threw && what.find("more than 8 ranges") != std::string::npos);
        }
        st = keep;
        CHECK(count(op) == 4);
    }
    // --- one frame beyond the capacities fails ALONE: the other frames of the batch are delivered ---
    {
        HipBackgroundSubtraction::Settings s4 = HipBackgroundSubtraction::settings();
        const HipBackgroundSubtraction::Settings keep = s4;
        s4.max_blobs = 8; s4.max_batch = 4;
        HipBackgroundSubtraction::init(s4, W, H);
        std::vector<uint8_t> bg4(W * H, 120);
        auto b4 = cmn::Image::Make(H, W, 1); std::memcpy(b4->data(), bg4.data(), bg4.size());
        HipBackgroundSubtraction::set_background(b4);
        std::vector<TileImage> tiles; std::vector<std::future<SegmentationData>> futs;
        for (int k = 0; k < 3; ++k) {
            std::vector<uint8_t> g(W * H, 120);
            const int nb = k == 1 ? 40 : 3;                                   // frame 1 holds 40 blobs: more than the whole pool (max_batch * max_blobs = 32)
            for (int i = 0; i < nb; ++i) for (int q = 0; q < 4; ++q) g[(size_t)(2 + 6 * (i / 10)) * W + 3 + 6 * (i % 10) + q] = 10;
            TileImage tile; tile.images.push_back(gray_to_bgr(g, W, H, 3, rng, true));
            tile.promise = std::make_unique<std::promise<SegmentationData>>();
            futs.push_back(tile.promise->get_future());
            tiles.emplace_back(std::move(tile));
        }
        hooks->apply(std::move(tiles));
        for (int k = 0; k < 3; ++k) {
            bool threw = false; std::string what; int n = -1;
            try { SegmentationData d = futs[k].get(); n = (int)d.frame.n(); } catch (const std::exception& e) { threw = true; what = e.what(); }
            if (k == 1) CHECK(threw && what.find("capacity") != std::string::npos);
            else CHECK(!threw && n == 3);
        }
        HipBackgroundSubtraction::init(keep, W, H);
        HipBackgroundSubtraction::set_background(b4);
    }
    // --- full-size BGRA tiles in pageable memory: the host copy into the pinned ring and the DMA overlap (upload.hip) ---
    {
        const int BW = 2048, BH = 2048, NB = 24;
        {   // detect_batch_size is a uchar: a batch outside 1 .. 255 is refused by name, never truncated
            HipBackgroundSubtraction::Settings bad; bad.max_batch = 256;
            bool threw = false; try { HipBackgroundSubtraction::init(bad, 64, 64); } catch (const std::exception& e) { threw = std::string(e.what()).find("detect_batch_size") != std::string::npos; } CHECK(threw);
            bad.max_batch = 0;
            threw = false; try { HipBackgroundSubtraction::init(bad, 64, 64); } catch (const std::exception&) { threw = true; } CHECK(threw);
        }
        HipBackgroundSubtraction::Settings sb; sb.max_batch = NB;
        HipBackgroundSubtraction::init(sb, BW, BH);
        auto bgb = cmn::Image::Make(BH, BW, 1); std::memset(bgb->data(), 120, (size_t)BW * BH);
        HipBackgroundSubtraction::set_background(bgb);
        double wall_ms = 0;
        for (int rep = 0; rep < 2; ++rep) {                                   // rep 0 allocates the ring and starts the copy threads
            std::vector<TileImage> tiles; std::vector<std::future<SegmentationData>> futs;
            for (int k = 0; k < NB; ++k) {
                TileImage tile; tile.images.push_back(cmn::Image::Make(BH, BW, 4));
                std::memset(tile.images[0]->data(), 120, (size_t)BW * BH * 4);
                for (int q = 0; q < 40; ++q) for (int c = 0; c < 3; ++c) tile.images[0]->data()[((size_t)(100 + k) * BW + 200 + q) * 4 + c] = 10;
                tile.promise = std::make_unique<std::promise<SegmentationData>>();
                futs.push_back(tile.promise->get_future());
                tiles.emplace_back(std::move(tile));
            }
            double c0, d0, c1, d1; int64_t n0, n1;
            HipBackgroundSubtraction::upload_stats(c0, d0, n0);
            const auto t0 = std::chrono::steady_clock::now();
            hooks->apply(std::move(tiles));
            wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            HipBackgroundSubtraction::upload_stats(c1, d1, n1);
            for (int k = 0; k < NB; ++k) { SegmentationData d = futs[k].get(); CHECK(d.frame.n() == 1 && (*d.frame.mask()[0])[0] == cmn::HorizontalLine(100 + k, 200, 239)); }
            if (rep == 1) {
                const double copy = c1 - c0, dma = d1 - d0;
                std::printf("upload of %d BGRA tiles 2048x2048: apply %.1f ms, host copy %.1f ms, DMA %.1f ms (%.1f GB/s)\n", NB, wall_ms, copy, dma, NB * 16.777216 / wall_ms);
                CHECK(n1 - n0 == NB);
                CHECK(wall_ms < copy + dma);                                  // the legs overlap ...
                CHECK(NB * 16.777216 / wall_ms > 15.0);                       // ... and the tiles move at a PCIe-class rate (GB/s), pv::Frame building included
            }
        }
        // --- the same with 100 objects per tile (what C4 holds): building the pv::Frame objects of the first half of the batch overlaps the
        //     upload of the second half (Settings::split_batch) ---
        for (int split = 0; split < 2; ++split) {
            HipBackgroundSubtraction::Settings sp; sp.max_batch = NB; sp.split_batch = split != 0;
            HipBackgroundSubtraction::init(sp, BW, BH);
            HipBackgroundSubtraction::set_background(bgb);
            double best = 1e9;
            for (int rep = 0; rep < 4; ++rep) {
                std::vector<TileImage> tiles; std::vector<std::future<SegmentationData>> futs;
                for (int k = 0; k < NB; ++k) {
                    TileImage tile; tile.images.push_back(cmn::Image::Make(BH, BW, 4));
                    std::memset(tile.images[0]->data(), 120, (size_t)BW * BH * 4);
                    for (int ob = 0; ob < 100; ++ob)                               // 100 objects of 30 x 12 pixels
                        for (int yy = 0; yy < 12; ++yy)
                            for (int q = 0; q < 30; ++q)
                                for (int c = 0; c < 3; ++c) tile.images[0]->data()[((size_t)(100 + 150 * (ob / 10) + yy) * BW + 100 + 150 * (ob % 10) + q + k) * 4 + c] = 10;
                    tile.promise = std::make_unique<std::promise<SegmentationData>>();
                    futs.push_back(tile.promise->get_future());
                    tiles.emplace_back(std::move(tile));
                }
                const auto t0 = std::chrono::steady_clock::now();
                hooks->apply(std::move(tiles));
                const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
                for (int k = 0; k < NB; ++k) { SegmentationData d = futs[k].get(); CHECK(d.frame.n() == 100 && (*d.frame.mask()[0])[0] == cmn::HorizontalLine(100, 100 + k, 129 + k)); }
                if (rep > 0 && ms < best) best = ms;
            }
            std::printf("apply() of %d BGRA tiles 2048x2048 with 100 objects each, %s: %.1f ms (%.0f tiles/s)\n", NB, split ? "two half batches on two contexts" : "one batch", best, NB * 1e3 / best);
        }
        HipBackgroundSubtraction::init(HipBackgroundSubtraction::Settings{s}, W, H);
        std::vector<uint8_t> bg5(W * H, 120);
        auto b5 = cmn::Image::Make(H, W, 1); std::memcpy(b5->data(), bg5.data(), bg5.size());
        HipBackgroundSubtraction::set_background(b5);
    }
    // --- TileImage destroyed with a live promise raises inside the future (core/TileImage.cpp:13-21) ---
    {
        std::future<SegmentationData> f;
        { TileImage t; t.promise = std::make_unique<std::promise<SegmentationData>>(); f = t.promise->get_future(); }
        bool threw = false;
        try { f.get(); } catch (const std::exception&) { threw = true; }
        CHECK(threw);
    }
    // --- posture + identity crops of a batch through the adapters (posture::calculate_posture / constraints::diff_image) ---
    {
        const int PW = 256, PH = 128;
        trexhip_params p; trexhip_default_params(&p, PW, PH); p.max_batch = 2;
        trexhip_ctx* ctx = nullptr;
        CHECK(trexhip_create(&p, &ctx) == 0);
        std::vector<uint8_t> pbg((size_t)PW * PH, 200), f0 = pbg, f1 = pbg;
        auto ellipse = [&](std::vector<uint8_t>& f, float cx, float cy, float a, float b, float th) {
            for (int y = 0; y < PH; ++y) for (int x = 0; x < PW; ++x) {
                const float u = (x - cx) * std::cos(th) + (y - cy) * std::sin(th), v = -(x - cx) * std::sin(th) + (y - cy) * std::cos(th);
                if (u * u / (a * a) + v * v / (b * b) <= 1.f) f[(size_t)y * PW + x] = (uint8_t)(60 + (x + y) % 50);
            }
        };
        ellipse(f0, 60, 40, 22, 6, 0.3f); ellipse(f0, 180, 80, 18, 5, 2.0f); ellipse(f1, 120, 64, 25, 7, 1.1f);
        f1[10 * PW + 10] = 20;                                                   // a single-pixel blob: outline but no midline
        CHECK(trexhip_set_background(ctx, pbg.data(), PW) == 0);
        const uint8_t* fp[2] = {f0.data(), f1.data()};
        CHECK(trexhip_segment(ctx, fp, PW, 2) == 0);
        trexhip_batch_result res{};
        CHECK(trexhip_fetch(ctx, &res) == 0);
        CHECK(res.total_blobs == 4);
        {   // the adapter owns device buffers of the context: it goes out of scope before trexhip_destroy
        HipPosture hp(ctx);
        HipPosture::Settings ps;
        std::vector<HipPosture::Expected> r;
        try { r = hp.calculate_posture(0, (int)res.total_blobs, ps); }
        catch (const std::exception& e) { std::fprintf(stderr, "calculate_posture: %s\n", e.what()); std::exit(1); }
        CHECK(r.size() == 4);
        {   // the same blobs through the reference's own loop (posture::calculate_posture: threshold += 2 until a midline is found):
            // these bodies give a midline at the first threshold, so the results are those of the single pass; refused settings throw
            std::vector<int32_t> used;
            auto r2 = hp.calculate_posture((int)res.total_blobs, ps, &used);
            CHECK(r2.size() == 4 && used.size() == 4);
            for (size_t b = 0; b < r2.size(); ++b) {
                CHECK((bool)r2[b] == (bool)r[b] && r2[b].value.outline.size() == r[b].value.outline.size());
                CHECK((r2[b].value.midline != nullptr) == (r[b].value.midline != nullptr));
                if (r2[b].value.midline) CHECK(used[b] == ps.track_posture_threshold && r2[b].value.midline->size() == r[b].value.midline->size());
            }
            HipPosture::Settings bad = ps; bad.peak_mode_broad = true;
            bool threw = false; try { (void)hp.calculate_posture((int)res.total_blobs, bad); } catch (const std::exception&) { threw = true; }
            CHECK(threw);
            // posture_direction_smoothing > 1: the movement direction of every blob is handed to Midline::post_process (Individual.cpp:1364-1369).
            // A vector along the midline's own direction changes nothing, the opposite one turns the normalised midline round (angle + pi,
            // head and tail index swapped); without the vectors the adapter refuses instead of silently skipping the history flip
            HipPosture::Settings sm = ps; sm.posture_direction_smoothing = 5;
            threw = false; try { (void)hp.calculate_posture(0, (int)res.total_blobs, sm); } catch (const std::exception&) { threw = true; }
            CHECK(threw);
            auto base = hp.calculate_posture(0, (int)res.total_blobs, ps);
            std::vector<cmn::Vec2> along(res.total_blobs), against(res.total_blobs);
            for (size_t b = 0; b < base.size(); ++b) {
                along[b] = against[b] = cmn::Vec2(0.f, 0.f);
                if (!base[b].value.midline) continue;
                const auto& sg = base[b].value.midline->segments();
                const float dx = sg[1].pos.x - sg[0].pos.x, dy = sg[1].pos.y - sg[0].pos.y, L = std::sqrt(dx * dx + dy * dy);
                along[b] = cmn::Vec2(dx / L, dy / L); against[b] = cmn::Vec2(-dx / L, -dy / L);
            }
            auto ra = hp.calculate_posture(0, (int)res.total_blobs, sm, &along), rb = hp.calculate_posture(0, (int)res.total_blobs, sm, &against);
            int turned = 0;
            for (size_t b = 0; b < base.size(); ++b) {
                if (!base[b].value.normalized_midline) continue;
                Midline &n0 = *base[b].value.normalized_midline, &na = *ra[b].value.normalized_midline, &nb = *rb[b].value.normalized_midline;
                CHECK(na.angle() == n0.angle() && na.tail_index() == n0.tail_index() && na.head_index() == n0.head_index());
                CHECK(nb.tail_index() == n0.head_index() && nb.head_index() == n0.tail_index());
                float da = std::fabs(nb.angle() - n0.angle()); if (da > 3.14159265f) da = 6.2831853f - da;
                CHECK(da > 2.5f);                                             // the other end is the head now
                CHECK(std::fabs(nb.len() - n0.len()) < 0.1f * n0.len());            // (the straightened head part is at the other end now)
                ++turned;
            }
            CHECK(turned == 3);
            r = hp.calculate_posture(0, (int)res.total_blobs, ps);            // leave the single-pass midlines in place for the crops below
        }
        int with_midline = 0;
        for (size_t b = 0; b < r.size(); ++b) {
            CHECK((bool)r[b]);
            CHECK(!r[b].value.outline.empty());
            const trexhip_blob& B = res.blobs[b];
            if (B.n_pixels == 1) { CHECK(!r[b].value.midline && !r[b].value.normalized_midline); CHECK(r[b].value.outline.size() >= 3 && r[b].value.outline.size() <= 8); continue; }
            ++with_midline;
            CHECK(r[b].value.midline && r[b].value.midline->size() > 5 && !r[b].value.midline->is_normalized());
            Midline& nm = *r[b].value.normalized_midline;
            CHECK(nm.is_normalized() && nm.size() == 25);
            CHECK(nm.segments()[0].pos == cmn::Vec2(0.f, 0.f));                  // head at the origin (Outline.cpp:1436-1441)
            CHECK(nm.len() > 25.f && nm.len() < 60.f);                           // long axes 36..50 px
            double l = 0;
            for (size_t i = 1; i < nm.size(); ++i) l += std::hypot(nm.segments()[i].pos.x - nm.segments()[i - 1].pos.x, nm.segments()[i].pos.y - nm.segments()[i - 1].pos.y);
            CHECK(std::fabs(l - nm.len()) < 1e-2);
        }
        CHECK(with_midline == 3);
        for (int mode = 0; mode < 4; ++mode) {
            std::vector<cmn::Image::Ptr> imgs;
            try { imgs = hp.diff_images(mode, (int)res.total_blobs, 80, 80, nullptr, 1.0f, 0); }
            catch (const std::exception& e) { std::fprintf(stderr, "diff_images mode %d: %s\n", mode, e.what()); std::exit(1); }
            CHECK(imgs.size() == 4);
            for (size_t b = 0; b < imgs.size(); ++b) {
                if (mode >= 2 && res.blobs[b].n_pixels == 1) { CHECK(!imgs[b]); continue; }
                CHECK(imgs[b] && imgs[b]->rows == 80 && imgs[b]->cols == 80 && imgs[b]->dims == 1);
                uint64_t sum = 0; for (size_t i = 0; i < imgs[b]->size(); ++i) sum += imgs[b]->data()[i];
                CHECK(sum > 0);
                if (mode == 0) CHECK(sum == res.blobs[b].sp);                    // un-normalised crop holds exactly the blob's grey values
            }
        }
        }
        trexhip_destroy(ctx);
    }
    {   // --- SplitBlob::split for merged individuals (PrefilterBlobs.cpp:235-236) through HipSplitBlob ---
        const int SW = 256, SH = 128;
        trexhip_params p; trexhip_default_params(&p, SW, SH); p.max_batch = 1;
        trexhip_ctx* ctx = nullptr;
        CHECK(trexhip_create(&p, &ctx) == 0);
        std::vector<uint8_t> sbg((size_t)SW * SH), fr;
        for (int y = 0; y < SH; ++y) for (int x = 0; x < SW; ++x) sbg[(size_t)y * SW + x] = (uint8_t)(140 + ((x * 3 + y * 5) & 31) - 16);
        fr = sbg;
        auto body = [&](float cx, float cy, float a, float b, float amp) {      // darkness falls off from the body axis
            for (int y = 0; y < SH; ++y) for (int x = 0; x < SW; ++x) {
                const float r2 = (x - cx) * (x - cx) / (a * a) + (y - cy) * (y - cy) / (b * b);
                const float d = amp * std::min(1.f, std::max(0.f, 1.15f - r2));
                const int v = (int)sbg[(size_t)y * SW + x] - (int)std::lround(d);
                if (d > 0 && v < fr[(size_t)y * SW + x]) fr[(size_t)y * SW + x] = (uint8_t)std::max(0, v);
            }
        };
        body(70, 50, 18, 5, 90); body(72, 59, 17, 5, 80);                       // two touching individuals = one detect blob
        body(180, 60, 18, 5, 85);                                               // a single one
        CHECK(trexhip_set_background(ctx, sbg.data(), SW) == 0);
        const uint8_t* fp[1] = {fr.data()};
        CHECK(trexhip_segment(ctx, fp, SW, 1) == 0);
        trexhip_batch_result res{};
        CHECK(trexhip_fetch(ctx, &res) == 0);
        CHECK(res.total_blobs == 2);
        {
        HipSplitBlob sb(ctx);
        HipSplitBlob::Settings ss;
        ss.track_size_filter = {{40.0, 330.0}};
        ss.track_threshold_is_absolute = false;
        std::vector<HipSplitBlob::Split> r;
        try { r = sb.split({{0u, 2}, {1u, 2}}, res, ss); }
        catch (const std::exception& e) { std::fprintf(stderr, "HipSplitBlob::split: %s\n", e.what()); std::exit(1); }
        CHECK(r.size() == 2);
        // the CPU restatement of SplitBlob::split on the same blobs
        oracle_split_params op{};
        op.initial_threshold = 16; op.algorithm = 1; op.blob_split_max_shrink = 0.2f; op.blob_split_global_shrink_limit = 0.2f; op.cm_per_pixel = 1.f;
        op.n_ranges = 1; op.ranges[0] = 40; op.ranges[1] = 330;
        for (int b = 0; b < 2; ++b) {
            const trexhip_blob& B = res.blobs[b];
            std::vector<oracle_run> orr(B.n_runs);
            for (uint32_t i = 0; i < B.n_runs; ++i) { const trexhip_run& q = res.runs[res.frames[0].run_begin + B.run_begin + i]; orr[i].x0 = q.x0; orr[i].x1 = q.x1; orr[i].y = q.y; orr[i].pad = 0; }
            oracle_split_info oi{};
            oracle_split_search(orr.data(), (int32_t)B.n_runs, res.pixels + res.frames[0].pix_begin + B.pix_begin, sbg.data(), SW, SW, SH, 1, 8, &op, 2, &oi);
            CHECK(r[b].threshold == oi.threshold);
            CHECK((int)r[b].blobs.size() == oi.n_result);
        }
        CHECK(r[0].threshold > 16 && r[0].blobs.size() >= 2);                   // the pair separates above the track threshold
        CHECK(r[1].threshold == -1 && r[1].blobs.empty());                      // one individual cannot be split into two
        size_t prev = SIZE_MAX;
        for (auto& pr : r[0].blobs) {                                           // descending sizes, lines relative to the big blob, pixels match lines
            size_t n = 0;
            for (auto& l : *pr.lines) { n += (size_t)l.x1 - l.x0 + 1; CHECK(l.y <= res.blobs[0].y1 - res.blobs[0].y0 && l.x1 <= res.blobs[0].x1 - res.blobs[0].x0); }
            CHECK(pr.pixels && pr.pixels->size() == n && n <= prev);
            prev = n;
        }
        // --- the history split on the same frame (HistorySplit.cpp:52-312 -> PrefilterBlobs::split_big): two individuals were last seen inside the
        // merged blob, one inside the single one; each blob is the only object its individuals are paired with ---
        {
            using HS = HipHistorySplit;
            auto id_of = [&](const int b) { const trexhip_blob& B = res.blobs[b]; const trexhip_run& r0 = res.runs[res.frames[0].run_begin + B.run_begin];
                                            return HipSplitBlob::bid(r0.x0, r0.x1, r0.y, B.n_runs); };
            const HS::bid_t merged = id_of(0), single = id_of(1);
            HS::Frame F;
            F.blob_mappings[merged] = {0, 1}; F.blob_mappings[single] = {2};
            F.paired[0] = {{merged, 3.f}}; F.paired[1] = {{merged, 4.5f}}; F.paired[2] = {{single, 1.f}};
            F.last_positions[0] = {cmn::Vec2(70, 50)}; F.last_positions[1] = {cmn::Vec2(72, 59)}; F.last_positions[2] = {cmn::Vec2(180, 60)};
            for (int f = 0; f < 3; ++f) F.valid_frame_streak[f] = 20;
            F.blob_pos[merged] = cmn::Vec2((float)res.blobs[0].x0, (float)res.blobs[0].y0);
            F.blob_pos[single] = cmn::Vec2((float)res.blobs[1].x0, (float)res.blobs[1].y0);
            const HS::Decision D = HS::decide(F, HS::Settings{});
            CHECK(D.big_blobs.size() == 1 && D.big_blobs[0] == merged);
            CHECK(D.expect.at(merged).number == 2 && !D.expect.at(merged).allow_less_than && D.expect.at(merged).centers.size() == 2);
            CHECK(D.expect.at(merged).centers[0][0].x == 70.f - F.blob_pos[merged].x);           // relative to the blob's bounds().pos() (:283-284)
            // the same decision by the CPU restatement
            {
                const int32_t map_off[3] = {0, 2, 3}, map_fish[3] = {0, 1, 2}, pair_off[4] = {0, 1, 2, 3}, pair_blob[3] = {0, 0, 1}, streak[3] = {20, 20, 20};
                const float pair_d[3] = {3.f, 4.5f, 1.f};
                int32_t number[2], coff[3], cfish[8]; uint8_t allow[2], big[2];
                CHECK(oracle_history_split(2, 3, map_off, map_fish, pair_off, pair_blob, pair_d, streak, -1, nullptr, 0, 1, number, allow, big, coff, cfish) == 1);
                CHECK(number[0] == 2 && number[1] == 0 && big[0] == 1 && big[1] == 0 && cfish[0] == 0 && cfish[1] == 1);
            }
            std::map<HS::bid_t, uint32_t> pooled{{merged, 0u}, {single, 1u}};
            std::vector<HS::Outcome> o;
            try { o = HS::split_big(sb, D, pooled, res, ss, 1.f); }
            catch (const std::exception& e) { std::fprintf(stderr, "HipHistorySplit::split_big: %s\n", e.what()); std::exit(1); }
            CHECK(o.size() == 1 && o[0].blob == merged && o[0].threshold == r[0].threshold && !o[0].kept_whole && !o[0].whole_to_noise);
            CHECK(o[0].regular.size() == 2 && o[0].regular.size() + o[0].noise.size() == r[0].blobs.size());       // the two largest pieces stay objects (:286-292)
            size_t n0 = 0, n1 = 0;
            for (auto& l : *o[0].regular[0].lines) n0 += (size_t)l.x1 - l.x0 + 1;
            for (auto& l : *o[0].regular[1].lines) n1 += (size_t)l.x1 - l.x0 + 1;
            CHECK(n0 >= n1 && n1 >= 40 * 0.35);                                  // in_range_of_one(r, 0.35, 1) of the size filter 40 .. 330
            // with the switch off nothing is split (:63-68)
            HS::Settings off; off.track_do_history_split = false;
            CHECK(HS::decide(F, off).big_blobs.empty());
        }
        }
        trexhip_destroy(ctx);
    }
    hooks->deinit();
    CHECK(detect::try_pipeline_manager(type) == nullptr);               // deinit unregisters the pipeline (BackgroundSubtraction.cpp:118-120)

    // --- identity facade: files written by the pytest wrapper: weights blob, crops, expected probabilities ---
    if (argc >= 4) {
        auto slurp = [](const char* p) { std::ifstream f(p, std::ios::binary); return std::vector<char>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>()); };
        auto blob = slurp(argv[1]); auto crops = slurp(argv[2]); auto expect = slurp(argv[3]);
        HipVINetwork net(0);
        CHECK(!net.weights_loaded());
        {   // probabilities before load_weights -> error through the future
            std::vector<cmn::Image::Ptr> ims; ims.push_back(cmn::Image::Make(80, 80, 1));
            auto f = net.probabilities(std::move(ims), [](auto&&, auto&&) {});
            bool threw = false; try { f.get(); } catch (const std::exception&) { threw = true; } CHECK(threw);
        }
        net.load_weights(blob.data(), blob.size());
        const int C = net.num_classes();
        const size_t n = crops.size() / 6400;
        CHECK(expect.size() == n * C * sizeof(float));
        std::vector<cmn::Image::Ptr> ims;
        for (size_t i = 0; i < n; ++i) { auto im = cmn::Image::Make(80, 80, 1); std::memcpy(im->data(), crops.data() + i * 6400, 6400); ims.push_back(std::move(im)); }
        std::vector<std::vector<float>> values; std::vector<float> indexes;
        net.probabilities(std::move(ims), [&](std::vector<std::vector<float>>&& v, std::vector<float>&& idx) { values = std::move(v); indexes = std::move(idx); }).get();
        CHECK(values.size() == n && indexes.size() == n);
        const float* e = reinterpret_cast<const float*>(expect.data());
        double worst = 0;
        for (size_t i = 0; i < n; ++i) { CHECK(indexes[i] == (float)i); for (int c = 0; c < C; ++c) worst = std::max(worst, (double)std::fabs(values[i][c] - e[i * C + c])); }
        CHECK(worst <= 1e-4);
        auto flat = HipVINetwork::transform_results(n + 2, indexes, values);
        CHECK(flat.size() == (n + 2) * C && flat[(n + 1) * C] == -1.f && flat[0] == values[0][0]);
        {   // synchronous probabilities + paverages (VisualIdentification.h:110-180)
            std::vector<cmn::Image::Ptr> again;
            for (size_t i = 0; i < n; ++i) { auto im = cmn::Image::Make(80, 80, 1); std::memcpy(im->data(), crops.data() + i * 6400, 6400); again.emplace_back(std::move(im)); }
            std::vector<int> ids(n);
            for (size_t i = 0; i < n; ++i) ids[i] = (int)(i % 2);
            auto av = net.paverages(ids, std::move(again));
            CHECK(av.size() == 2 && av[0].values.size() == (size_t)C && av[0].samples + av[1].samples == (float)n);
            double m0 = 0; int c0 = 0;
            for (size_t i = 0; i < n; i += 2) { m0 += e[i * C + 0]; ++c0; }
            CHECK(std::fabs(av[0].values[0] - m0 / c0) <= 1e-4);
        }
        CHECK(HipVINetwork::batch_size_for(8) == 64 && HipVINetwork::batch_size_for(100) == 128 && HipVINetwork::batch_size_for(65) == 128);
        {   // wrong crop size
            std::vector<cmn::Image::Ptr> bad; bad.push_back(cmn::Image::Make(64, 64, 1));
            auto f = net.probabilities(std::move(bad), [](auto&&, auto&&) {});
            bool threw = false; try { f.get(); } catch (const std::exception&) { threw = true; } CHECK(threw);
        }
        {   // 3-channel crops into a 1-channel network: refused before anything is read past the images
            std::vector<cmn::Image::Ptr> bad; bad.push_back(cmn::Image::Make(80, 80, 3)); bad.push_back(cmn::Image::Make(80, 80, 3));
            auto f = net.probabilities(std::move(bad), [](auto&&, auto&&) {});
            bool threw = false; try { f.get(); } catch (const std::exception&) { threw = true; } CHECK(threw);
        }
        {   // training facade: a few steps on one batch lower its loss; the exported weights load back into the inference path
            HipVINetwork::Trainer tr(net, blob.data(), blob.size(), 64, 0.001f, 7);
            std::vector<float> x(n * 6400);
            std::vector<int32_t> y(n);
            for (size_t i = 0; i < x.size(); ++i) x[i] = 0.97f * (float)(unsigned char)crops[i];
            for (size_t i = 0; i < n; ++i) y[i] = (int32_t)(i % (size_t)C);
            auto first = tr.train_batch(x.data(), y.data(), (int)n);
            HipVINetwork::Trainer::Result last = first;
            for (int k = 0; k < 5; ++k) last = tr.train_batch(x.data(), y.data(), (int)n);
            CHECK(std::isfinite(first.loss) && last.loss < first.loss && tr.steps() == 6 && last.correct >= 0 && last.correct <= (int)n);
            auto val = tr.evaluate(x.data(), y.data(), (int)n);                            // eval mode: running statistics, no dropout; nothing changes
            CHECK(std::isfinite(val.loss) && val.correct >= 0 && val.correct <= (int)n && tr.steps() == 6);
            y[0] = C;                                                                   // label out of range: refused (visual_recognition_torch.py:1112)
            bool threw = false; try { tr.train_batch(x.data(), y.data(), (int)n); } catch (const std::exception&) { threw = true; } CHECK(threw);
            threw = false; try { tr.train_batch(x.data(), y.data(), 65); } catch (const std::exception&) { threw = true; } CHECK(threw);   // n > max_batch
            auto w = tr.weights();
            CHECK(w.size() == blob.size() && std::memcmp(w.data(), blob.data(), 32) == 0 && std::memcmp(w.data(), blob.data(), w.size()) != 0);
            tr.apply();
            CHECK(net.num_classes() == C);
            std::printf("training facade ok: loss %.4f -> %.4f in 6 steps\n", first.loss, last.loss);
        }
        std::printf("identity facade ok: %zu crops, %d classes, max |dp| = %.3g\n", n, C, worst);
    }
    std::printf("host adapter ok\n");
    return 0;
}
