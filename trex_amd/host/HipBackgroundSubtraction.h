// HipBackgroundSubtraction.h -- the detection backend TRex talks to, implemented on libtrexhip.
//
// Same surface as track::BackgroundSubtraction (Application/src/tracker/python/BackgroundSubtraction.h:10-26):
// set_background / apply / deinit / fps, plus register_hip_backend() which installs detect::BackendHooks
// (python/BackendRegistry.h:10-19) exactly like register_yolo_backend does (python/YOLO.cpp:1738-1747).
// The contract of apply(std::vector<TileImage>&&) follows BackgroundSubtraction.cpp:146-342 line by line:
//   read tile.images (BGR/BGRA, pooled) -> blobs in full-frame coordinates -> frame.set_encoding ->
//   frame.add_object(pair) per kept blob (< UINT16_MAX lines) -> promise->set_value(std::move(tile.data)) ;
//   on error promise->set_exception ; then ALWAYS tile.callback() (exceptions swallowed) and every image
//   back to buffers::TileBuffers ; finally one fps sample (tiles / seconds).
// Header-only; inside a TRex build define TREXHIP_WITH_TREX so TRex's own headers supply the types.
#pragma once
#ifdef TREXHIP_WITH_TREX
#include <commons.pc.h>
#include <core/TileImage.h>
#include <core/TileBuffers.h>
#include <python/BackendRegistry.h>
#include <python/PipelineRegistry.h>
#else
#include "trex_types.h"
#endif
#include <algorithm>
#include <chrono>
#include <cstring>
#include <thread>
#include <shared_mutex>
#include <stdexcept>
#include <string>
#include "../../include/trexhip.h"

namespace track {

struct HipBackgroundSubtraction {
    struct Settings {                 // the values BackgroundSubtraction::apply / RawProcessing read
        int detect_threshold = 15, threshold_maximum = 255;
        bool detect_threshold_is_absolute = true, enable_difference = true, image_invert = false;
        // true (default): keep |p| >= detect_threshold -- "disregards any pixel |p| < threshold" (core/default_config.cpp:1168, the wording
        // of the track-stage rule :1167 that Tests/test_pixels.cpp:1026-1059 pins as >=); false: strict |p| > detect_threshold (cv::threshold)
        bool inclusive = true;
        int color_channel = -1;       // std::optional<uint8_t> color_channel; <0 = none
        double cm_per_pixel = 1.0;
        std::vector<std::pair<double, double>> detect_size_filter;
        cmn::meta_encoding_t meta_encoding = cmn::meta_encoding_t::gray;
        int device = 0, max_batch = 8;
        // gray / binary encodings: colour tiles are reduced to gray by the upload threads on their way into the pinned ring (a quarter of
        // the bytes cross PCIe); true = upload the colour tile and reduce on the device (trexhip_params::device_color_reduce)
        bool device_color_reduce = false;
        // a batch of two or more tiles is handled in two halves on two device contexts: the first half's pv::Frame objects are built and its
        // promises fulfilled while the second half is copied and segmented.  false = one context (half the device memory)
        bool split_batch = true;
        // capacities per frame (0 = scaled with the frame size in init()): raw horizontal lines, kept blobs, kept foreground pixels.
        // A frame that exceeds one fails alone (its promise gets the exception); the other frames of the batch are delivered.
        int max_runs = 0, max_blobs = 0, max_pixels = 0;
        // optional pre-processing of RawProcessing::generate_binary that this backend does NOT implement: init() throws when TRex's
        // settings switch one on (grabber/misc/default_config.cpp:121-129, core/default_config.cpp:1161-1162) -- never a silent divergence
        bool image_adjust = false, equalize_histogram = false, correct_luminance = false, use_adaptive_threshold = false;
        int blur_difference = 0;
        // morphology that IS implemented (core/default_config.cpp:1163-1165)
        int dilation_size = 0, closing_size = 3;
        bool use_closing = false;
    };

    static void init(const Settings& s, uint32_t width, uint32_t height) {
        auto& d = data();
        std::unique_lock g(d.gpu_mutex);
        if (d.ctx) { trexhip_destroy(d.ctx); d.ctx = nullptr; }
        if (d.ctx2) { trexhip_destroy(d.ctx2); d.ctx2 = nullptr; }
        // TRex's detect_batch_size is a uchar (core/default_config.cpp:1113, default 1): a batch beyond 255 tiles cannot be asked for by the
        // reference's settings system, and a Settings object that carries more was filled by hand -- refuse it by name instead of truncating
        if (s.max_batch < 1 || s.max_batch > 255)
            throw std::invalid_argument("HipBackgroundSubtraction::init: max_batch = " + std::to_string(s.max_batch) + " is outside detect_batch_size's range (uchar: 1 .. 255)");
        d.settings = s;
        trexhip_params p;
        trexhip_default_params(&p, (int32_t)width, (int32_t)height);
        p.device = s.device; p.max_batch = s.max_batch; p.device_color_reduce = s.device_color_reduce;
        const trexhip_live_params lp = live_of(s);         // (throws on more than 8 size ranges)
        p.threshold = lp.threshold; p.threshold_maximum = lp.threshold_maximum; p.inclusive = lp.inclusive;
        p.absolute_difference = lp.absolute_difference; p.enable_difference = lp.enable_difference;
        p.image_invert = lp.image_invert; p.cm_per_pixel = lp.cm_per_pixel;
        p.n_ranges = lp.n_ranges;
        for (int i = 0; i < 16; ++i) p.ranges[i] = lp.ranges[i];
        p.dilation_size = s.dilation_size; p.use_closing = s.use_closing; p.closing_size = s.closing_size;
        p.image_adjust = s.image_adjust; p.blur_difference = s.blur_difference; p.equalize_histogram = s.equalize_histogram;
        p.correct_luminance = s.correct_luminance; p.use_adaptive_threshold = s.use_adaptive_threshold;   // non-default -> trexhip_create refuses
        // capacities scale with the frame: at most one line per two pixels of a row, one foreground pixel in four, one blob per 32x32 pixels
        const uint64_t px = (uint64_t)width * height;
        p.max_runs = s.max_runs > 0 ? s.max_runs : (int32_t)std::min<uint64_t>(std::max<uint64_t>(65536, px / 16), 1u << 22);
        p.max_pixels = s.max_pixels > 0 ? s.max_pixels : (int32_t)std::min<uint64_t>(std::max<uint64_t>(1u << 20, px / 4), 1u << 26);
        p.max_blobs = s.max_blobs > 0 ? s.max_blobs : (int32_t)std::min<uint64_t>(std::max<uint64_t>(2048, px / 1024), 1u << 16);
        // pixel arrays in the frame's meta_encoding (the encoding is fixed per context: it sizes the pixel pool)
        p.pixel_encoding = s.meta_encoding == cmn::meta_encoding_t::rgb8 ? TREXHIP_ENC_RGB8
                         : s.meta_encoding == cmn::meta_encoding_t::r3g3b2 ? TREXHIP_ENC_R3G3B2 : TREXHIP_ENC_GRAY;
        d.context_encoding = s.meta_encoding;
        check(trexhip_create(&p, &d.ctx));
        d.pushed = lp;
        // a second context takes the second half of a batch: its tiles are copied and segmented while the first half's tables are turned
        // into pv::Frame objects and their promises fulfilled (BackgroundSubtraction.cpp:146-342 handles the tiles one after the other and
        // fulfils each promise as it goes)
        // (the second half never holds more than half of the batch's images: its context is sized for that)
        if (s.max_batch >= 2 && s.split_batch) { p.max_batch = std::max(1, s.max_batch / 2); check(trexhip_create(&p, &d.ctx2)); }
        d.width = width; d.height = height;
        d.has_background = false;
    }

    // BackgroundSubtraction::set_background -> Data::set (BackgroundSubtraction.cpp:86-101)
    static void set_background(const cmn::Image::Ptr& average) {
        auto& d = data();
        std::unique_lock g(d.gpu_mutex);
        if (!average) { d.has_background = false; return; }
        if (!d.ctx) throw std::runtime_error("HipBackgroundSubtraction: not initialised");
        for (trexhip_ctx* c : {d.ctx, d.ctx2}) {
            if (!c) continue;
            if (average->dims == 1) check(trexhip_set_background(c, average->data(), (int32_t)average->cols));
            else if (average->dims == 3 || average->dims == 4) {
                // a colour average (Background(image, rgb8)): detection thresholds grey differences, so the library reduces the model like
                // the frames (cv::cvtColor BGR2GRAY, 8-bit fixed point -- or the selected color_channel) and keeps the colour image for
                // the per-channel difference crops
                const int cc = d.settings.color_channel;
                if (((size_t)average->rows * average->cols) % 4 == 0)
                    check(trexhip_set_background_color(c, average->data(), (int32_t)(average->cols * average->dims), (int32_t)average->dims, cc));
                else {                                                  // odd pixel counts: gray model only
                    std::vector<uint8_t> gray((size_t)average->rows * average->cols);
                    const uint8_t* p = average->data();
                    for (size_t i = 0; i < gray.size(); ++i, p += average->dims)
                        gray[i] = cc >= 0 && cc < (int)average->dims ? p[cc] : (uint8_t)((p[0] * 1868u + p[1] * 9617u + p[2] * 4899u + 8192u) >> 14);
                    check(trexhip_set_background(c, gray.data(), (int32_t)average->cols));
                }
            } else throw std::runtime_error("HipBackgroundSubtraction: background must have 1, 3 or 4 channels");
        }
        d.has_background = true;
        g.unlock();
        if (d.has_type) if (auto* m = detect::try_pipeline_manager(d.type)) m->set_paused(false);   // BackgroundSubtraction.cpp:86-99
    }

    // BackgroundSubtraction::apply(TileImage&&) (BackgroundSubtraction.cpp:107-116): synchronous variant of the
    // pipeline enqueue -- the caller's PipelineManager thread calls apply(vector) below
    static std::future<SegmentationData> apply(TileImage&& tiled) {
        if (tiled.promise) throw std::runtime_error("Tiled.promise was already set.");
        tiled.promise = std::make_unique<std::promise<SegmentationData>>();
        auto f = tiled.promise->get_future();
        std::vector<TileImage> v;
        v.emplace_back(std::move(tiled));
        apply(std::move(v));
        return f;
    }

    static void apply(std::vector<TileImage>&& tiled) {
        const auto t0 = std::chrono::steady_clock::now();
        auto& d = data();
        // the reference takes a shared lock here (BackgroundSubtraction.cpp:130) because its per-call state is thread-local; this backend
        // has ONE device context (staging buffers, pinned result tables), so concurrent apply() calls are serialised
        std::unique_lock guard(d.gpu_mutex);
        // one device batch per call: all tiles' first images (1 tile == full frame for bg-sub, DetectionTypes.cpp:294-295) -- in two halves on
        // two contexts when the batch has at least two tiles
        std::string batch_error;
        int channels = 0;
        bool ok = d.ctx && d.has_background;
        if (!ok) batch_error = "Background image not set";
        // the context was created for one pixel encoding (gray and binary share the gray arrays; add_object drops the pixels of binary)
        if (ok) {
            const auto want = d.settings.meta_encoding == cmn::meta_encoding_t::binary ? cmn::meta_encoding_t::gray : d.settings.meta_encoding;
            const auto have = d.context_encoding == cmn::meta_encoding_t::binary ? cmn::meta_encoding_t::gray : d.context_encoding;
            if (want != have) { ok = false; batch_error = "Invalid image mode: meta_encoding changed after init() (re-initialise the backend)"; }   // cf. BackgroundSubtraction.cpp:188
        }
        // the reference re-reads cm_per_pixel / detect_size_filter (and RawProcessing its thresholds) on every apply()
        // (BackgroundSubtraction.cpp:137-143): settings() changed since the last batch are pushed to the contexts before this one
        if (ok) {
            try {
                const trexhip_live_params lp = live_of(d.settings);
                if (std::memcmp(&lp, &d.pushed, sizeof(lp)) != 0) {
                    for (trexhip_ctx* c : {d.ctx, d.ctx2}) if (c) check(trexhip_update_params(c, &lp));
                    d.pushed = lp;
                }
            } catch (const std::exception& e) { ok = false; batch_error = e.what(); }
        }
        size_t n_images = 0;
        if (ok) {
            for (auto& tile : tiled)
                for (auto& image : tile.images) {
                    if (image->dims != 3 && image->dims != 4) { ok = false; batch_error = "Invalid number of channels in input image for the network."; }
                    if (channels == 0) channels = (int)image->dims;
                    if ((int)image->dims != channels) { ok = false; batch_error = "mixed channel counts in one batch"; }
                    // the reference asserts the tile size against the average image (BackgroundSubtraction.cpp:195-199); a smaller tile
                    // would make the upload read past the image
                    if (image->cols != d.width || image->rows != d.height) { ok = false; batch_error = "tile image size does not match the size the backend was initialised with"; }
                    ++n_images;
                }
            if ((int)n_images > d.settings.max_batch) { ok = false; batch_error = "more tile images than max_batch"; }
        }
        struct Part {                                                    // a run of tiles handled by one context
            size_t t0 = 0, t1 = 0;
            trexhip_ctx* ctx = nullptr;
            std::vector<const uint8_t*> ptrs;
            trexhip_batch_result res{};
            bool ok = true;
            std::string error;
        };
        Part parts[2];
        const bool two = ok && d.ctx2 && tiled.size() >= 2;
        {
            size_t split = tiled.size();
            if (two) { size_t acc = 0; for (split = 0; split < tiled.size() && 2 * acc < n_images; ++split) acc += tiled[split].images.size(); }
            parts[0].t0 = 0; parts[0].t1 = split; parts[0].ctx = d.ctx;
            parts[1].t0 = split; parts[1].t1 = tiled.size(); parts[1].ctx = d.ctx2;
        }
        auto segment = [&](Part& p) {
            p.ok = ok; p.error = batch_error;
            if (!p.ok) return;
            for (size_t t = p.t0; t < p.t1; ++t) for (auto& image : tiled[t].images) p.ptrs.push_back(image->data());
            if (p.ptrs.empty()) return;
            if (trexhip_segment_color(p.ctx, p.ptrs.data(), (int32_t)(d.width * (uint32_t)channels), (int32_t)p.ptrs.size(), channels, d.settings.color_channel) != 0) {
                p.ok = false; p.error = trexhip_last_error(); return;
            }
            // TREXHIP_E_CAPACITY is per frame: the tables stay valid for every frame whose frame_info.flags is 0, only the flagged
            // frames fail (the reference has no capacity limits and handles each tile on its own, BackgroundSubtraction.cpp:146-342)
            const int rc = trexhip_fetch(p.ctx, &p.res);
            if (rc != 0 && rc != TREXHIP_E_CAPACITY) { p.ok = false; p.error = trexhip_last_error(); }
        };
        // tables -> pv::Frame objects -> promises of the part's tiles (callbacks and buffer returns follow on the calling thread, in tile order)
        auto deliver = [&](Part& p) {
            size_t img_index = 0;
            for (size_t t = p.t0; t < p.t1; ++t) {
                auto& tile = tiled[t];
                try {
                    if (!p.ok) throw std::runtime_error(p.error);
                    tile.data.frame.set_encoding(d.settings.meta_encoding);            // :303
                    for (size_t k = 0; k < tile.images.size(); ++k) {
                        const trexhip_frame_info& fi = p.res.frames[img_index + k];
                        if (fi.flags != 0)
                            throw std::runtime_error("frame exceeds the backend's capacity (" + std::string((fi.flags & TREXHIP_FRAME_OVERFLOW_RUNS) ? "max_runs" : "max_blobs / max_pixels") +
                                                     "): raise HipBackgroundSubtraction::Settings::max_runs / max_blobs / max_pixels");
                    }
                    for (size_t k = 0; k < tile.images.size(); ++k) {
                        const trexhip_frame_info& fi = p.res.frames[img_index + k];
                        for (uint32_t b = 0; b < fi.n_blobs; ++b) {
                            const trexhip_blob& B = p.res.blobs[fi.blob_begin + b];
                            if (B.n_runs >= UINT16_MAX) continue;                       // :306-313
                            auto lines = std::make_unique<std::vector<cmn::HorizontalLine>>();
                            lines->reserve(B.n_runs);
                            const trexhip_run* r = p.res.runs + fi.run_begin + B.run_begin;
                            for (uint32_t j = 0; j < B.n_runs; ++j) lines->emplace_back(r[j].y, r[j].x0, r[j].x1);
                            const uint8_t* px = p.res.pixels + (size_t)(fi.pix_begin + B.pix_begin) * p.res.pixel_channels;
                            auto pixels = std::make_unique<cmn::PixelArray_t>(px, px + (size_t)B.n_pixels * p.res.pixel_channels);   // pv.cpp:512
                            tile.data.frame.add_object(cmn::blob::Pair(std::move(lines), std::move(pixels), 0));   // :305-314
                        }
                    }
                    tile.promise->set_value(std::move(tile.data));                      // :319
                    tile.promise = nullptr;
                } catch (...) {
                    if (tile.promise) { tile.promise->set_exception(std::current_exception()); tile.promise = nullptr; }   // :322-325
                }
                img_index += tile.images.size();
            }
        };
        segment(parts[0]);
        if (two) {
            struct Joiner {                                               // an exception below must not destroy a joinable thread
                std::thread t;
                ~Joiner() { if (t.joinable()) t.join(); }
            } first{std::thread([&] { deliver(parts[0]); })};             // the first half's consumers are served while the second half is on its way
            try {
                segment(parts[1]);
            } catch (const std::exception& e) { parts[1].ok = false; parts[1].error = e.what(); }
            deliver(parts[1]);
        } else deliver(parts[0]);
        for (auto&& tile : tiled) {
            try { if (tile.callback) tile.callback(); } catch (...) {}              // :328-334
            for (auto& image : tile.images) buffers::TileBuffers::get().move_back(std::move(image));   // :336-339
            tile.images.clear();
        }
        if (!tiled.empty()) {
            const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            d.add_time_sample(double(tiled.size()) / el);                           // :344-346
        }
    }

    static void deinit() {
        auto& d = data();
        if (d.has_type) {                                              // BackgroundSubtraction::deinit (BackgroundSubtraction.cpp:118-120)
            if (auto* m = detect::try_pipeline_manager(d.type)) m->clean_up();
            detect::unregister_pipeline(d.type);
            d.has_type = false;
        }
        std::unique_lock g(d.gpu_mutex);
        if (d.ctx) { trexhip_destroy(d.ctx); d.ctx = nullptr; }
        if (d.ctx2) { trexhip_destroy(d.ctx2); d.ctx2 = nullptr; }
        d.has_background = false;
    }
    static double fps() { return data().fps(); }
    // the two legs of the tile upload since init(): host milliseconds spent copying pageable tiles into the pinned ring and DMA
    // milliseconds (they overlap each other: apply() takes about the larger of the two), per the library's own counters
    static void upload_stats(double& host_copy_ms, double& dma_ms, int64_t& frames) {
        auto& d = data();
        std::unique_lock g(d.gpu_mutex);
        host_copy_ms = dma_ms = 0; frames = 0;
        if (!d.ctx) return;
        int64_t n2 = 0;
        for (trexhip_ctx* c : {d.ctx, d.ctx2}) {
            if (!c) continue;
            double a = 0, b = 0; int64_t fa = 0;
            (void)trexhip_profile_read(c, TREXHIP_STAGE_UPLOAD_COPY, &a, &fa);
            (void)trexhip_profile_read(c, TREXHIP_STAGE_UPLOAD_DMA, &b, &n2);
            host_copy_ms += a; dma_ms += b; frames += fa;
        }
    }
    static bool is_initializing() { return false; }

    // detect::register_backend(type, hooks) -- python/BackendRegistry.h:19; pattern of register_yolo_backend
    static void register_hip_backend(detect::ObjectDetectionType::Class type, const Settings& s, uint32_t w, uint32_t h) {
        detect::BackendHooks hooks;
        // the constructor of BackgroundSubtraction registers the pipeline PAUSED and waits for the background inside the
        // callback (BackgroundSubtraction.cpp:50-82); Detection::apply(TileImage&&) enqueues into that manager (Detection.cpp:124-146)
        hooks.init = [s, w, h, type]() {
            HipBackgroundSubtraction::init(s, w, h);
            data().type = type; data().has_type = true;
            detect::register_pipeline(type, (size_t)(s.max_batch > 0 ? s.max_batch : 1), /*start_paused=*/true, [type](std::vector<TileImage>&& images) {
                auto* m = detect::try_pipeline_manager(type);
                while (!data().has_background && m && !m->is_terminated()) std::this_thread::sleep_for(std::chrono::milliseconds(100));
                if (!m || !m->is_terminated()) HipBackgroundSubtraction::apply(std::move(images));
            });
        };
        hooks.deinit = []() { HipBackgroundSubtraction::deinit(); };
        hooks.is_initializing = []() { return HipBackgroundSubtraction::is_initializing(); };
        hooks.fps = []() { return HipBackgroundSubtraction::fps(); };
        hooks.apply = [](std::vector<TileImage>&& tiles) { HipBackgroundSubtraction::apply(std::move(tiles)); };
        hooks.set_background = [](const cmn::Image::Ptr& bg) { HipBackgroundSubtraction::set_background(bg); };
        detect::register_backend(type, std::move(hooks));
    }

    // the reference re-reads its settings on every apply() (BackgroundSubtraction.cpp:132-143); callers update these in place (not while an
    // apply() is running).  Live: detect_threshold, threshold_maximum, inclusive, detect_threshold_is_absolute, enable_difference,
    // image_invert, cm_per_pixel, detect_size_filter (pushed through trexhip_update_params before the next batch) and color_channel (read per
    // call).  meta_encoding, the capacities, morphology and device need a new init(); a changed meta_encoding fails the batch.
    static Settings& settings() { return data().settings; }

    // the live part of Settings as the ABI takes it
    static trexhip_live_params live_of(const Settings& s) {
        if (s.detect_size_filter.size() > 8) throw std::runtime_error("HipBackgroundSubtraction: detect_size_filter with more than 8 ranges is not supported");
        trexhip_live_params lp;
        std::memset(&lp, 0, sizeof(lp));
        lp.threshold = s.detect_threshold; lp.threshold_maximum = s.threshold_maximum; lp.inclusive = s.inclusive ? 1 : 0;
        lp.enable_difference = s.enable_difference ? 1 : 0; lp.absolute_difference = s.detect_threshold_is_absolute ? 1 : 0;
        lp.image_invert = s.image_invert ? 1 : 0; lp.zero_is_background = 1;
        lp.n_ranges = (int32_t)s.detect_size_filter.size(); lp.cm_per_pixel = s.cm_per_pixel;
        for (int i = 0; i < lp.n_ranges; ++i) { lp.ranges[2 * i] = s.detect_size_filter[i].first; lp.ranges[2 * i + 1] = s.detect_size_filter[i].second; }
        return lp;
    }

private:
    struct Data {
        trexhip_ctx *ctx = nullptr, *ctx2 = nullptr;
        uint32_t width = 0, height = 0;
        Settings settings;
        trexhip_live_params pushed{};             // what the contexts currently hold
        detect::ObjectDetectionType::Class type{};
        bool has_type = false;
        cmn::meta_encoding_t context_encoding = cmn::meta_encoding_t::gray;
        bool has_background = false;
        double time = 0, samples = 0;
        std::shared_mutex gpu_mutex, time_mutex;
        double fps() { std::shared_lock g(time_mutex); return samples == 0 ? 0 : time / samples; }   // :25-30
        void add_time_sample(double s) { std::unique_lock g(time_mutex); time += s; samples++; }        // :31-35
    };
    static Data& data() { static Data d; return d; }
    static void check(int rc) { if (rc != 0) throw std::runtime_error(std::string("libtrexhip: ") + trexhip_last_error()); }
};

}  // namespace track
