// HipSplitBlob.h -- batch replacement for the per-blob split of merged individuals in the tracker's preprocessing:
//   SplitBlob s(&cache, *Tracker::background(), b.get());  auto ret = s.split(ex.number, ex.centers, *Tracker::background());
//       Application/src/tracker/tracking/PrefilterBlobs.cpp:235-236, SplitBlob.h:36-73, SplitBlob.cpp:419-800
// for blob_split_algorithm threshold / threshold_approximate.  The threshold search runs on the device for all candidates of
// the context's LAST segmented (and fetched) batch at once (trexhip_split_search_device), the sub-blobs come from
// trexhip_rethreshold_per_blob_device at the thresholds found; this header only restores SplitBlob::split's result shape:
// blobs sorted by (num_pixels, blob_id) descending (:169-172), those below the shrink limit removed (:204-221), lines relative to
// the big blob's bounds().pos() (:166-167) -- PrefilterBlobs::split_big adds that offset back (PrefilterBlobs.cpp:275).
// Inside a TRex build define TREXHIP_WITH_TREX to get the real blob::Pair / HorizontalLine types.
#pragma once
#include <algorithm>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <tuple>
#include <utility>
#include <vector>
#include "../../include/trexhip.h"
#ifdef TREXHIP_WITH_TREX
#include <processing/PVBlob.h>
#else
#include "trex_types.h"
#endif

namespace track {

class HipSplitBlob {
public:
    struct Settings {                                   // names = TRex settings (SplitBlob.cpp:75-98)
        int track_threshold = 15, track_posture_threshold = 15;
        bool calculate_posture = true;
        int blob_split_algorithm = 1;                   // blob_split_algorithm_t: 0 none, 1 threshold, 2 threshold_approximate
        float blob_split_max_shrink = 0.2f, blob_split_global_shrink_limit = 0.2f;
        std::vector<std::pair<double, double>> track_size_filter;
        bool track_threshold_is_absolute = true;        // Background::diff: |bg - p| or max(bg - p, 0)
        bool track_background_subtraction = true;       // false: the grey value itself is thresholded
    };
    struct Expectation {                                // split_expectation of one big blob (PrefilterBlobs.h), blob = pooled index in the batch
        uint32_t blob = 0;
        int number = 2;
    };
    struct Split {
        uint32_t blob = 0;
        int threshold = -1;                             // best_match.threshold, -1: "could not find anything" (:797-799)
        std::vector<cmn::blob::Pair> blobs;             // what SplitBlob::split returns (empty when nothing was found)
        bool beyond_capacity = false;                   // the blob does not fit the device search (> 61440 pixels / 2048 lines): caller keeps it unsplit
    };

    explicit HipSplitBlob(trexhip_ctx* ctx) : _ctx(ctx) {}        // must be destroyed before trexhip_destroy(ctx)
    ~HipSplitBlob() { release(); }
    HipSplitBlob(const HipSplitBlob&) = delete;
    HipSplitBlob& operator=(const HipSplitBlob&) = delete;

    // one entry per expectation, in the order given; `det` = the trexhip_fetch result of the batch.  Overwrites the context's
    // re-threshold table.
    std::vector<Split> split(const std::vector<Expectation>& expect, const trexhip_batch_result& det, const Settings& s) {
        const uint32_t total_blobs = det.total_blobs;
        std::vector<Split> out(expect.size());
        for (size_t i = 0; i < expect.size(); ++i) out[i].blob = expect[i].blob;
        if (expect.empty() || total_blobs == 0 || s.blob_split_algorithm == 0) return out;       // :421-422
        reserve(total_blobs);
        std::vector<int32_t> presumed((size_t)total_blobs, 0);
        for (const Expectation& e : expect) {
            if (e.blob >= total_blobs) throw std::invalid_argument("HipSplitBlob::split: blob index outside the batch");
            presumed[e.blob] = e.number;
        }
        trexhip_split_params sp; trexhip_default_split_params(&sp);
        sp.track_threshold = s.track_threshold; sp.track_posture_threshold = s.track_posture_threshold;
        sp.calculate_posture = s.calculate_posture ? 1 : 0; sp.algorithm = s.blob_split_algorithm;
        sp.blob_split_max_shrink = s.blob_split_max_shrink; sp.blob_split_global_shrink_limit = s.blob_split_global_shrink_limit;
        if (s.track_size_filter.size() > 8) throw std::invalid_argument("HipSplitBlob::split: at most 8 size ranges");
        sp.n_ranges = (int32_t)s.track_size_filter.size();
        std::vector<double> ranges;
        for (size_t i = 0; i < s.track_size_filter.size(); ++i) {
            sp.size_ranges[2 * i] = s.track_size_filter[i].first; sp.size_ranges[2 * i + 1] = s.track_size_filter[i].second;
            ranges.push_back(s.track_size_filter[i].first); ranges.push_back(s.track_size_filter[i].second);
        }
        const int method = !s.track_background_subtraction ? 2 : (s.track_threshold_is_absolute ? 0 : 1);
        check(trexhip_copy_to_device(_ctx, _d_presumed, presumed.data(), presumed.size() * sizeof(int32_t)));
        check(trexhip_split_search_device(_ctx, &sp, method, _d_presumed, (int32_t)total_blobs, _d_thr, _d_info));
        check(trexhip_rethreshold_per_blob_device(_ctx, 0, _d_thr, method, ranges.empty() ? nullptr : ranges.data(), sp.n_ranges));
        trexhip_batch_result sub{};
        check(trexhip_fetch_rethreshold(_ctx, &sub));
        std::vector<trexhip_split_info> info((size_t)total_blobs);
        check(trexhip_copy_to_host(_ctx, info.data(), _d_info, info.size() * sizeof(trexhip_split_info)));

        // sub-blobs of each candidate: (frame, index) lists by parent
        std::vector<std::vector<std::pair<int, uint32_t>>> children((size_t)total_blobs);
        for (int f = 0; f < sub.n_frames; ++f) {
            const trexhip_frame_info& fi = sub.frames[f];
            for (uint32_t k = 0; k < fi.n_blobs; ++k) {
                const trexhip_blob& B = sub.blobs[fi.blob_begin + k];
                if (B.parent < total_blobs && presumed[B.parent] > 0) children[B.parent].emplace_back(f, fi.blob_begin + k);
            }
        }
        const uint32_t ch = sub.pixel_channels ? sub.pixel_channels : 1u;
        for (size_t i = 0; i < expect.size(); ++i) {
            const uint32_t b = expect[i].blob;
            const trexhip_split_info& I = info[b];
            Split& S = out[i];
            S.beyond_capacity = I.status == 2;
            S.threshold = I.threshold;
            if (I.threshold < 0) continue;
            const trexhip_blob& big = det.blobs[b];
            const float sqcm = _cm_per_pixel * _cm_per_pixel;
            std::vector<std::tuple<uint32_t, uint32_t, int, uint32_t>> order;       // (num_pixels, blob_id, frame, index)
            for (auto [f, k] : children[b]) {
                const trexhip_blob& B = sub.blobs[k];
                if ((double)((float)B.n_pixels * sqcm) < I.min_size_bound) continue;               // evaluate_result_multiple's removal (:204-221)
                const trexhip_run& r0 = sub.runs[sub.frames[f].run_begin + B.run_begin];
                order.emplace_back(B.n_pixels, bid((uint32_t)(r0.x0 - big.x0), (uint32_t)(r0.x1 - big.x0), (uint32_t)(r0.y - big.y0), B.n_runs), f, k);
            }
            std::sort(order.begin(), order.end(), [](const auto& a, const auto& c) {
                return std::make_tuple(std::get<0>(a), std::get<1>(a)) > std::make_tuple(std::get<0>(c), std::get<1>(c)); });
            for (const auto& o : order) {
                const trexhip_frame_info& fi = sub.frames[std::get<2>(o)];
                const trexhip_blob& B = sub.blobs[std::get<3>(o)];
                auto lines = std::make_unique<std::vector<cmn::HorizontalLine>>();
                lines->reserve(B.n_runs);
                for (uint32_t r = 0; r < B.n_runs; ++r) {
                    const trexhip_run& q = sub.runs[fi.run_begin + B.run_begin + r];
                    lines->emplace_back((uint16_t)(q.y - big.y0), (uint16_t)(q.x0 - big.x0), (uint16_t)(q.x1 - big.x0));   // add_offset(-bounds().pos())
                }
                auto px = std::make_unique<cmn::PixelArray_t>((size_t)B.n_pixels * ch);
                std::memcpy(px->data(), sub.pixels + ((size_t)fi.pix_begin + B.pix_begin) * ch, px->size());
                cmn::blob::Pair pair;
                pair.lines = std::move(lines); pair.pixels = std::move(px);
                S.blobs.emplace_back(std::move(pair));
            }
        }
        return out;
    }

    void set_cm_per_pixel(float v) { _cm_per_pixel = v; }          // the context's cm_per_pixel (trexhip_params), used for the size bound

    // pv::bid of a blob: 13/13/6-bit hash of its first line (x0 + (x1 - x0 + 1) / 2, y, number of lines)
    static uint32_t bid(uint32_t x0, uint32_t x1, uint32_t y, uint32_t n_lines) {
        uint32_t x = x0 + (x1 - x0 + 1) / 2;
        if (x > 8191) x = 8191;
        if (y > 8191) y = 8191;
        const uint32_t n = n_lines < 1 ? 1 : (n_lines > 63 ? 63 : n_lines);
        return (x << 19) | (y << 6) | n;
    }

private:
    static void check(int rc) { if (rc != 0) throw std::runtime_error(std::string("libtrexhip: ") + trexhip_last_error()); }
    void release() {
        void* p[] = {_d_presumed, _d_thr, _d_info};
        for (void* q : p) if (q) (void)trexhip_device_free(_ctx, q);
        _d_presumed = nullptr; _d_thr = nullptr; _d_info = nullptr; _cap = 0;
    }
    void reserve(uint32_t n) {
        if (n <= _cap) return;
        release();
        check(trexhip_device_alloc(_ctx, (size_t)n * sizeof(int32_t), reinterpret_cast<void**>(&_d_presumed)));
        check(trexhip_device_alloc(_ctx, (size_t)n * sizeof(int32_t), reinterpret_cast<void**>(&_d_thr)));
        check(trexhip_device_alloc(_ctx, (size_t)n * sizeof(trexhip_split_info), reinterpret_cast<void**>(&_d_info)));
        _cap = n;
    }
    trexhip_ctx* _ctx;
    int32_t *_d_presumed = nullptr, *_d_thr = nullptr;
    trexhip_split_info* _d_info = nullptr;
    uint32_t _cap = 0;
    float _cm_per_pixel = 1.f;
};

}  // namespace track
