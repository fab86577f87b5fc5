// HipHistorySplit.h -- the per-frame decision of TRex's history split (track_do_history_split, default on) as host code over the
// device split search:
//   HistorySplit(frame, need, pool)                    Application/src/tracker/tracking/HistorySplit.cpp:52-312
//     which blobs hold more individuals than the frame has objects for them, and how many each is expected to hold (`expect`, `big_blobs`)
//   PrefilterBlobs::split_big(..., expect, true, ...)   Application/src/tracker/tracking/PrefilterBlobs.cpp:152-316
//     the split of those blobs and what becomes of the pieces (regular / noise)
// The tracker state the decision reads -- PPFrame::blob_mappings / paired / last_positions (PPFrame.h:68-70, filled by
// PPFrame::init_cache from the individuals' motion caches) and IndividualCache::valid_frame_streak -- is the CALLER's: this header takes
// it as plain maps and does not restate the tracker (out of scope).  The threshold search itself runs on the device
// (HipSplitBlob::split -> trexhip_split_search_device) for all big blobs of the batch at once.
//
// The reference keeps blob_mappings / paired / probs_per_fish in robin_hood hash maps and walks them in hash order (:76, :151, :266).
// What is decided does not depend on that order except at exact ties of two distances for one blob (:206, the individual assigned
// first keeps it) and in the order of `centers` (read by the watershed algorithm only, which this library refuses).  Here the maps
// are ordered: blobs ascending, an individual's edges in the order the caller gives them.
#pragma once
#include <cstdint>
#include <map>
#include <queue>
#include <set>
#include <tuple>
#include <utility>
#include <vector>
#include "HipSplitBlob.h"

namespace track {

class HipHistorySplit {
public:
    using bid_t = uint32_t;                             // pv::bid
    using Idx_t = int32_t;                              // Idx_t; negative = invalid (what a manual split leaves in blob_mappings, :101)
    using prob_t = float;                               // Match::prob_t: the distance of the pairing (smaller = closer)

    struct Frame {                                      // the parts of PPFrame the constructor reads
        std::map<bid_t, std::set<Idx_t>> blob_mappings;                         // PPFrame.h:68
        std::map<Idx_t, std::vector<std::pair<bid_t, prob_t>>> paired;           // :69 (a map in TRex: one entry per blob)
        std::map<Idx_t, std::vector<cmn::Vec2>> last_positions;                 // :70
        std::map<Idx_t, int> valid_frame_streak;                                // frame.cached(fdx)->valid_frame_streak (:139-147)
        std::map<bid_t, cmn::Vec2> blob_pos;                                    // bounds().pos() of every blob of the frame (has_bdx / bdx_to_ptr)
        std::vector<bid_t> manual_splits;                                       // manual_splits of this frame (:9-13)
    };
    struct Settings {
        bool track_do_history_split = true;
        int track_history_split_threshold = -1;         // Frame_t; negative = invalid (the default)
    };
    struct split_expectation {                          // PrefilterBlobs.h:21-28
        size_t number = 0;
        bool allow_less_than = false;
        std::vector<std::vector<cmn::Vec2>> centers;
    };
    struct Decision {
        std::map<bid_t, split_expectation> expect;
        std::vector<bid_t> big_blobs;                   // insertion order (UnorderedVectorSet)
    };

    static Decision decide(const Frame& frame, const Settings& s) {
        Decision D;
        std::set<bid_t> already_walked, big;
        auto add_big = [&](const bid_t b) { if (big.insert(b).second) D.big_blobs.push_back(b); };
        for (const bid_t bdx : frame.manual_splits) {                            // apply_manual_matches :18-36
            if (!frame.blob_pos.count(bdx)) continue;                            // frame.has_bdx(bdx)
            add_big(bdx);
            D.expect[bdx].number = 2;
            D.expect[bdx].allow_less_than = false;
            already_walked.insert(bdx);
        }
        if (!s.track_do_history_split) return D;                                 // :63-68
        static const std::set<Idx_t> no_fish;
        static const std::vector<std::pair<bid_t, prob_t>> no_edges;
        auto mapped = [&](const bid_t b) -> const std::set<Idx_t>& { auto it = frame.blob_mappings.find(b); return it == frame.blob_mappings.end() ? no_fish : it->second; };
        auto edges = [&](const Idx_t f) -> const std::vector<std::pair<bid_t, prob_t>>& { auto it = frame.paired.find(f); return it == frame.paired.end() ? no_edges : it->second; };

        for (const auto& [bdx0, set0] : frame.blob_mappings) {                   // :76
            if (set0.size() <= 1) continue;                                      // :79
            if (already_walked.count(bdx0)) continue;                            // :82
            // the clique of this blob: every individual mapped to it, every blob those are paired with, and so on (:88-161)
            std::vector<Idx_t> available_fdx;
            std::set<Idx_t> have_fdx;
            std::set<bid_t> available_bdx;
            std::queue<bid_t> q;
            q.push(bdx0);
            while (!q.empty()) {
                const bid_t current = q.front();
                q.pop();
                for (const Idx_t fdx : mapped(current)) {
                    if (fdx < 0) continue;                                       // :101
                    if (s.track_history_split_threshold >= 0) {                  // :104-148
                        auto it = frame.valid_frame_streak.find(fdx);
                        const int length = (it != frame.valid_frame_streak.end() && it->second > 0) ? it->second : -1;
                        if (length < 0 || length < s.track_history_split_threshold) continue;
                    }
                    for (const auto& [b, d] : edges(fdx)) {                      // :151-157
                        (void)d;
                        if (!available_bdx.count(b)) { q.push(b); available_bdx.insert(b); already_walked.insert(b); }
                    }
                    if (have_fdx.insert(fdx).second) available_fdx.push_back(fdx);      // :159
                }
            }
            if (available_fdx.size() <= available_bdx.size()) continue;          // :172: no more individuals than blobs

            std::map<bid_t, std::pair<Idx_t, prob_t>> assign_blob;                                  // :176
            std::map<Idx_t, std::set<std::tuple<prob_t, bid_t>>> probs_per_fish;                      // :178
            std::map<Idx_t, std::tuple<prob_t, bid_t>> assign_fish;                                   // :179
            std::vector<Idx_t> fish_order;                                       // (the order probs_per_fish is walked in at :266)
            std::queue<Idx_t> checks;
            for (const Idx_t c : available_fdx) {                                // :227-246
                const auto& pairs = edges(c);
                if (pairs.empty()) continue;
                std::set<std::tuple<prob_t, bid_t>> combinations;
                for (const auto& [b, d] : pairs) combinations.insert({d, b});
                assign_fish[c] = *combinations.begin();
                probs_per_fish[c] = std::move(combinations);
                fish_order.push_back(c);
                checks.push(c);
            }
            // every individual takes its closest blob; a closer individual takes a blob over and sends the other one back (:190-257)
            auto check_combinations = [&](const Idx_t fdx, std::set<std::tuple<prob_t, bid_t>>& combinations) -> bool {
                if (combinations.empty()) return false;
                const bid_t b = std::get<1>(*combinations.begin());
                const prob_t d = std::get<0>(*combinations.begin());
                auto it = assign_blob.find(b);
                if (it == assign_blob.end()) { assign_blob[b] = {fdx, d}; return true; }
                if (it->second.first != fdx) {
                    if (!(it->second.second <= d)) {
                        const Idx_t oid = it->second.first;
                        it->second = {fdx, d};
                        checks.push(oid);
                        return true;
                    }
                }
                combinations.erase(combinations.begin());
                return false;
            };
            while (!checks.empty()) {
                const Idx_t c = checks.front();
                checks.pop();
                auto& combinations = probs_per_fish.at(c);
                if (!combinations.empty() && !check_combinations(c, combinations)) checks.push(c);
            }
            // individuals without an alternative left: their closest blob has to hold one more (:266-303)
            for (const Idx_t fdx : fish_order) {
                if (!probs_per_fish.at(fdx).empty()) continue;
                const bid_t max_id = std::get<1>(assign_fish.at(fdx));
                auto pit = frame.blob_pos.find(max_id);                          // frame.bdx_to_ptr(max_id)
                if (pit == frame.blob_pos.end()) continue;
                auto append_centers = [&](const Idx_t who) {
                    std::vector<cmn::Vec2> c;
                    auto lit = frame.last_positions.find(who);
                    if (lit != frame.last_positions.end()) c = lit->second;
                    for (auto& pt : c) { pt.x -= pit->second.x; pt.y -= pit->second.y; }
                    D.expect[max_id].centers.emplace_back(std::move(c));
                };
                auto ait = assign_blob.find(max_id);
                if (ait != assign_blob.end()) {
                    ++D.expect[max_id].number;
                    append_centers(ait->second.first);
                    assign_blob.erase(ait);
                }
                ++D.expect[max_id].number;
                append_centers(fdx);
                add_big(max_id);
            }
        }
        return D;
    }

    // PrefilterBlobs::split_big for the decision's big blobs (PrefilterBlobs.cpp:205-300) over the batch the context segmented last:
    // `pooled` maps a pv::bid to the blob's pooled index in `det`.  Returns per big blob what the tracker receives: the pieces that
    // stay regular objects (at most `number`, largest first) and the ones that become noise; a blob the search finds nothing in
    // goes to noise whole (:262-263; with allow_less_than it stays regular, :241-250 -- never set by the history split itself).
    struct Outcome {
        bid_t blob = 0;
        int threshold = -1;
        bool kept_whole = false, whole_to_noise = false, beyond_capacity = false;
        std::vector<cmn::blob::Pair> regular, noise;
    };
    static std::vector<Outcome> split_big(HipSplitBlob& splitter, const Decision& D, const std::map<bid_t, uint32_t>& pooled,
                                          const trexhip_batch_result& det, const HipSplitBlob::Settings& ss, const float cm_per_pixel,
                                          const bool discard_small = true) {
        std::vector<HipSplitBlob::Expectation> ex;
        std::vector<bid_t> ids;
        for (const bid_t b : D.big_blobs) {
            auto it = pooled.find(b);
            if (it == pooled.end()) continue;
            HipSplitBlob::Expectation e;
            e.blob = it->second;
            auto xit = D.expect.find(b);
            e.number = xit != D.expect.end() ? (int)xit->second.number : 2;      // split_expectation ex(2, false) (:223)
            ex.push_back(e); ids.push_back(b);
        }
        std::vector<HipSplitBlob::Split> found = splitter.split(ex, det, ss);
        std::vector<Outcome> out(found.size());
        const float cm_sq = cm_per_pixel * cm_per_pixel;
        for (size_t i = 0; i < found.size(); ++i) {
            Outcome& O = out[i];
            O.blob = ids[i]; O.threshold = found[i].threshold; O.beyond_capacity = found[i].beyond_capacity;
            const auto xit = D.expect.find(ids[i]);
            const bool allow_less = xit != D.expect.end() && xit->second.allow_less_than;
            const size_t number = (size_t)ex[i].number;
            if (found[i].beyond_capacity) { O.kept_whole = true; continue; }     // not searched: the caller keeps the blob as it is
            if (allow_less && found[i].blobs.empty()) { O.kept_whole = true; continue; }       // :241-250 (the size test there needs recount(): the caller's)
            if (found[i].blobs.empty()) { O.whole_to_noise = true; continue; }    // :262-263
            // HipSplitBlob::split returns the pieces sorted by (pixels, id) descending = std::sort(found, greater) of (recount(0), id) (:266):
            // recount(0) counts every pixel of a piece
            size_t counter = 0;
            for (auto& piece : found[i].blobs) {
                const float r = (float)(piece.pixels ? piece.pixels->size() / (ss_channels(det)) : 0) * cm_sq;
                if (in_range_of_one(ss.track_size_filter, r, 0.35f, 1.f) && (!discard_small || counter < number)) {       // :286-292
                    O.regular.emplace_back(std::move(piece));
                    ++counter;
                } else
                    O.noise.emplace_back(std::move(piece));
            }
        }
        return out;
    }

    // SizeFilters::in_range_of_one (core/SizeFilters.cpp:36-53)
    static bool in_range_of_one(const std::vector<std::pair<double, double>>& ranges, const float cmsq, const float scale_factor = -1.f, float scale_factor_r = -1.f) {
        if (ranges.empty()) return true;
        if (scale_factor_r == -1.f) scale_factor_r = 2.f - (scale_factor < 0 ? -scale_factor : scale_factor);
        for (const auto& r : ranges) {
            const double lo = scale_factor == -1.f ? r.first : r.first * scale_factor, hi = scale_factor == -1.f ? r.second : r.second * scale_factor_r;
            if ((double)cmsq >= lo && (double)cmsq < hi) return true;             // Range::contains: [start, end)
        }
        return false;
    }

private:
    static size_t ss_channels(const trexhip_batch_result& det) { return det.pixel_channels ? det.pixel_channels : 1u; }
};

}  // namespace track
