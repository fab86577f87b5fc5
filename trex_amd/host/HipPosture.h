// HipPosture.h -- batch replacements for the per-blob posture and identity-crop calls of the tracker thread pool:
//   posture::calculate_posture(Frame_t, pv::BlobWeakPtr) -> expected<Result{outline, midline, normalized_midline}>
//       Application/src/tracker/tracking/Posture.h:34-40, Posture.cpp:305-399, Individual.cpp:1369-1372 (post_process + normalize)
//   constraints::diff_image(normalize, blob, midline_transform, median_midline_length_px, output_shape, background)
//       Application/src/tracker/tracking/FilterCache.h:67-72, FilterCache.cpp:265-294
// Both work on the blobs of the context's LAST segmented (and fetched) batch, in pooled order (frame-major), so the caller
// indexes results like trexhip_batch_result.  Inside a TRex build define TREXHIP_WITH_TREX to get the real Outline / Midline /
// Image types (tracking/Outline.h, misc/Image.h); here the stand-ins of trex_types.h carry the same members.
#pragma once
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>
#include "../../include/trexhip.h"
#ifdef TREXHIP_WITH_TREX
#include <tracking/Outline.h>
#include <misc/Image.h>
#else
#include "trex_types.h"
#endif

namespace track {

class HipPosture {
public:
    struct Settings {                                   // names = TRex settings (core/default_config.cpp:888-901)
        float outline_resample = 1.f;
        int outline_smooth_samples = 4, outline_smooth_step = 1, outline_approximate = 3;
        float outline_curvature_range_ratio = 0.03f, midline_walk_offset = 0.025f;
        uint32_t midline_resolution = 25;
        float midline_stiff_percentage = 0.15f;
        bool midline_invert = false, midline_start_with_head = false;
        int max_points = 512;                           // capacity per blob: traced lattice points = 2 per pixel edge of the outline; even, up to 4096
        // calculate_posture(n_blobs, settings): the thresholds of posture::calculate_posture's own loop (Posture.cpp:318,335)
        int track_posture_threshold = 15;               // core/default_config.cpp track_posture_threshold
        int threshold_method = 0;                       // 0 |bg - p| (track_threshold_is_absolute), 1 max(bg - p, 0), 2 p: as trexhip_rethreshold_device
        // settings this backend refuses (std::runtime_error from calculate_posture) instead of ignoring them
        int posture_closing_steps = 0, posture_direction_smoothing = 0;
        bool peak_mode_broad = false;
    };
    struct Result {                                     // posture::Result (Posture.h:34-38)
        Outline outline;
        Midline::Ptr midline;
        Midline::Ptr normalized_midline;
    };
    struct Expected {                                   // std::expected<Result, const char*> without C++23
        bool ok = false;
        const char* error = nullptr;
        Result value;
        explicit operator bool() const { return ok; }
    };

    explicit HipPosture(trexhip_ctx* ctx) : _ctx(ctx) {}          // must be destroyed before trexhip_destroy(ctx): it frees buffers of that context
    ~HipPosture() { release(); }
    HipPosture(const HipPosture&) = delete;
    HipPosture& operator=(const HipPosture&) = delete;

    // posture::calculate_posture(Frame_t, pv::BlobWeakPtr) for every detect blob of the batch, WITH the reference's loop: biggest
    // sub-blob at track_posture_threshold, +2 per failed attempt, first-outline fallback (Posture.cpp:305-399).  One entry per blob,
    // pooled order.  thresholds_used (optional): the threshold whose result each blob got (-1: nothing could be traced).
    // movement (optional, one per blob): MovementInformation::direction = Individual::calculate_previous_vector(frame).direction, which the
    // reference hands to Midline::post_process when posture_direction_smoothing > 1 (Individual.cpp:1364-1369); (0, 0) = none for that blob.
    // With posture_direction_smoothing > 1 the vectors are REQUIRED (the history flip would silently be missing otherwise).
    std::vector<Expected> calculate_posture(int n_blobs, const Settings& s, std::vector<int32_t>* thresholds_used = nullptr,
                                            const std::vector<cmn::Vec2>* movement = nullptr) {
        return run(true, 0, n_blobs, s, thresholds_used, movement);
    }
    // one pass over a table as it is: table 0 = the detect blobs, table 1 = the sub-blobs of the caller's last
    // trexhip_rethreshold*_device call.  One entry per blob of that table, pooled order.
    std::vector<Expected> calculate_posture(int table, int n_blobs, const Settings& s, const std::vector<cmn::Vec2>* movement = nullptr) {
        return run(false, table, n_blobs, s, nullptr, movement);
    }

private:
    std::vector<Expected> run(bool with_loop, int table, int n_blobs, const Settings& s, std::vector<int32_t>* thresholds_used,
                              const std::vector<cmn::Vec2>* movement) {
        std::vector<Expected> out((size_t)n_blobs);
        if (n_blobs <= 0) return out;
        if (s.posture_direction_smoothing > 1 && !movement)
            throw std::runtime_error("HipPosture: posture_direction_smoothing > 1 needs the movement direction of every blob (Individual::calculate_previous_vector)");
        if (movement && (int)movement->size() != n_blobs) throw std::runtime_error("HipPosture: one movement vector per blob");
        reserve(n_blobs, s);
        trexhip_posture_params pp; trexhip_default_posture_params(&pp);
        pp.outline_resample = s.outline_resample; pp.outline_smooth_samples = s.outline_smooth_samples;
        pp.outline_smooth_step = s.outline_smooth_step; pp.outline_approximate = s.outline_approximate;
        pp.outline_curvature_range_ratio = s.outline_curvature_range_ratio; pp.midline_walk_offset = s.midline_walk_offset;
        pp.max_points = s.max_points;
        pp.posture_closing_steps = s.posture_closing_steps; pp.posture_direction_smoothing = s.posture_direction_smoothing;
        pp.peak_mode = s.peak_mode_broad ? 1 : 0;                    // non-default values are refused by the library (TREXHIP_E_UNSUPPORTED)
        trexhip_midline_params mp; trexhip_default_midline_params(&mp);
        mp.midline_resolution = (int32_t)s.midline_resolution; mp.midline_stiff_percentage = s.midline_stiff_percentage;
        mp.midline_invert = s.midline_invert; mp.midline_start_with_head = s.midline_start_with_head;
        const size_t SEG = (size_t)s.max_points / 2 + 1, R = s.midline_resolution;
        if (with_loop) {
            int32_t* d_thr = nullptr;
            if (thresholds_used) check(trexhip_device_alloc(_ctx, (size_t)n_blobs * 4, reinterpret_cast<void**>(&d_thr)));
            const int rc = trexhip_posture_auto_device(_ctx, &pp, s.threshold_method, s.track_posture_threshold, n_blobs, _d_outline, _d_segments, _d_pinfo, d_thr, nullptr);
            if (rc == 0 && thresholds_used) {
                thresholds_used->resize((size_t)n_blobs);
                (void)trexhip_synchronize(_ctx);
                (void)trexhip_copy_to_host(_ctx, thresholds_used->data(), d_thr, (size_t)n_blobs * 4);
            }
            if (d_thr) (void)trexhip_device_free(_ctx, d_thr);
            check(rc);
        } else
            check(trexhip_posture_device(_ctx, table, &pp, n_blobs, _d_outline, _d_segments, _d_pinfo));
        // the raw midline (Result::midline) is read back before post_process straightens it in place
        std::vector<float> outline((size_t)n_blobs * s.max_points * 2), raw((size_t)n_blobs * SEG * 4), norm((size_t)n_blobs * R * 4);
        std::vector<trexhip_posture_info> pinfo((size_t)n_blobs);
        std::vector<trexhip_midline_info> minfo((size_t)n_blobs);
        check(trexhip_synchronize(_ctx));
        check(trexhip_copy_to_host(_ctx, raw.data(), _d_segments, raw.size() * 4));
        if (movement) {
            std::vector<float> mv((size_t)n_blobs * 2);
            for (int b = 0; b < n_blobs; ++b) { mv[2 * (size_t)b] = (*movement)[(size_t)b].x; mv[2 * (size_t)b + 1] = (*movement)[(size_t)b].y; }
            float* d_mv = nullptr;
            check(trexhip_device_alloc(_ctx, mv.size() * 4, reinterpret_cast<void**>(&d_mv)));
            int rc = trexhip_copy_to_device(_ctx, d_mv, mv.data(), mv.size() * 4);
            if (rc == 0) rc = trexhip_midline_movement_device(_ctx, &mp, n_blobs, s.max_points, _d_pinfo, _d_segments, _d_midline, _d_minfo, d_mv);
            if (rc == 0) rc = trexhip_synchronize(_ctx);
            (void)trexhip_device_free(_ctx, d_mv);
            check(rc);
        } else
            check(trexhip_midline_device(_ctx, &mp, n_blobs, s.max_points, _d_pinfo, _d_segments, _d_midline, _d_minfo));
        check(trexhip_synchronize(_ctx));
        check(trexhip_copy_to_host(_ctx, outline.data(), _d_outline, outline.size() * 4));
        check(trexhip_copy_to_host(_ctx, norm.data(), _d_midline, norm.size() * 4));
        check(trexhip_copy_to_host(_ctx, pinfo.data(), _d_pinfo, pinfo.size() * sizeof(trexhip_posture_info)));
        check(trexhip_copy_to_host(_ctx, minfo.data(), _d_minfo, minfo.size() * sizeof(trexhip_midline_info)));
        _minfo = minfo;
        for (int b = 0; b < n_blobs; ++b) {
            Expected& e = out[(size_t)b];
            const trexhip_posture_info& pi = pinfo[(size_t)b];
            if (pi.n_outline > 0) {
                auto pts = std::make_unique<std::vector<cmn::Vec2>>((size_t)pi.n_outline);
                const float* o = outline.data() + (size_t)b * s.max_points * 2;
                for (int i = 0; i < pi.n_outline; ++i) (*pts)[(size_t)i] = cmn::Vec2(o[2 * i], o[2 * i + 1]);
                e.value.outline.replace_points(std::move(pts));
            }
            if (pi.status != 0) {
                // the reference keeps the outline when no midline can be found (Posture.cpp:383-396), else fails
                if (pi.n_outline > 0) { e.ok = true; continue; }
                e.error = pi.status == 2 ? "Blob exceeds the posture capacity of the HIP backend." : "Cannot find valid posture.";
                continue;
            }
            e.ok = true;
            auto fill = [](Midline& m, const float* s4, int n) {
                m.segments().resize((size_t)n);
                for (int i = 0; i < n; ++i) {
                    MidlineSegment& g = m.segments()[(size_t)i];
                    g.pos = cmn::Vec2(s4[4 * i], s4[4 * i + 1]); g.height = s4[4 * i + 2]; g.l_length = s4[4 * i + 3];
                }
            };
            e.value.midline = std::make_unique<Midline>();
            fill(*e.value.midline, raw.data() + (size_t)b * SEG * 4, pi.n_segments);
            e.value.midline->tail_index() = pi.tail_index; e.value.midline->head_index() = pi.head_index;
            const trexhip_midline_info& mi = minfo[(size_t)b];
            if (mi.status == 0) {                       // Midline::normalize() returned a midline (Outline.cpp:1378-1380 else nullptr)
                auto nm = std::make_unique<Midline>();
                fill(*nm, norm.data() + (size_t)b * R * 4, (int)R);
                nm->len() = mi.len; nm->angle() = mi.angle; nm->offset() = cmn::Vec2(mi.offx, mi.offy);
                nm->is_normalized() = true;
                nm->tail_index() = pi.tail_index; nm->head_index() = pi.head_index;
                if (mi.reserved[0]) std::swap(nm->head_index(), nm->tail_index());      // turned round by the movement history (Outline.cpp:957-959)
                e.value.normalized_midline = std::move(nm);
            }
        }
        return out;
    }

public:
    // constraints::diff_image for every blob of the batch: `normalize` as in individual_image_normalization
    // (0 none, 1 moments, 2 posture, 3 legacy); posture / legacy use the midlines of the last calculate_posture call and the
    // caller's per-blob median midline length (nullptr: each blob's own length); blobs without a midline yield nullptr
    // (FilterCache.cpp:268-270).  difference: track_background_subtraction (0 raw grey, 1 |bg-p|, 2 max(bg-p,0)).
    std::vector<cmn::Image::Ptr> diff_images(int normalize, int n_blobs, int out_w, int out_h, const float* median_midline_length_px,
                                             float individual_image_scale, int difference) {
        std::vector<cmn::Image::Ptr> out((size_t)n_blobs);
        if (n_blobs <= 0) return out;
        const int ch = trexhip_pixel_channels(_ctx);                 // 3 for meta_encoding rgb8
        const size_t each = (size_t)out_w * out_h * (size_t)ch;
        if (_crops_cap < each * n_blobs) {
            if (_d_crops) (void)trexhip_device_free(_ctx, _d_crops);
            _d_crops = nullptr; _crops_cap = 0;
            check(trexhip_device_alloc(_ctx, each * n_blobs, reinterpret_cast<void**>(&_d_crops)));
            _crops_cap = each * n_blobs;
        }
        if (normalize == 0 || normalize == 1)
            check(trexhip_crops_device(_ctx, _d_crops, n_blobs, out_w, out_h, normalize == 0 ? TREXHIP_NORMALIZE_NONE : TREXHIP_NORMALIZE_MOMENTS, difference));
        else {
            if ((int)_minfo.size() < n_blobs) throw std::runtime_error("HipPosture::diff_images: calculate_posture must run first for posture / legacy normalisation");
            check(trexhip_crops_posture_device(_ctx, _d_crops, n_blobs, out_w, out_h, _d_minfo, median_midline_length_px, normalize == 3 ? 1 : 0,
                                               individual_image_scale, difference));
        }
        check(trexhip_synchronize(_ctx));
        std::vector<uint8_t> host(each * n_blobs);
        check(trexhip_copy_to_host(_ctx, host.data(), _d_crops, host.size()));
        for (int b = 0; b < n_blobs; ++b) {
            if (normalize >= 2 && _minfo[(size_t)b].status != 0) continue;          // no midline -> nullptr
            auto img = cmn::Image::Make((uint32_t)out_h, (uint32_t)out_w, (uint32_t)ch);
            std::memcpy(img->data(), host.data() + (size_t)b * each, each);
            out[(size_t)b] = std::move(img);
        }
        return out;
    }

    const uint8_t* device_crops() const { return _d_crops; }      // feed trexhip_identify_device without the host round trip

private:
    static void check(int rc) { if (rc != 0) throw std::runtime_error(std::string("libtrexhip: ") + trexhip_last_error()); }
    void release() {
        void* p[] = {_d_outline, _d_segments, _d_pinfo, _d_midline, _d_minfo, _d_crops};
        for (void* q : p) if (q) (void)trexhip_device_free(_ctx, q);
        _d_outline = _d_segments = _d_midline = nullptr; _d_pinfo = nullptr; _d_minfo = nullptr; _d_crops = nullptr; _cap = 0; _crops_cap = 0;
    }
    void reserve(int n, const Settings& s) {
        if (n <= _cap && s.max_points == _cap_points && (int)s.midline_resolution == _cap_res) return;
        uint8_t* keep = _d_crops; size_t keepc = _crops_cap; _d_crops = nullptr;
        release();
        _d_crops = keep; _crops_cap = keepc;
        const size_t N = (size_t)n;
        check(trexhip_device_alloc(_ctx, N * s.max_points * 2 * 4, reinterpret_cast<void**>(&_d_outline)));
        check(trexhip_device_alloc(_ctx, N * ((size_t)s.max_points / 2 + 1) * 4 * 4, reinterpret_cast<void**>(&_d_segments)));
        check(trexhip_device_alloc(_ctx, N * sizeof(trexhip_posture_info), reinterpret_cast<void**>(&_d_pinfo)));
        check(trexhip_device_alloc(_ctx, N * s.midline_resolution * 4 * 4, reinterpret_cast<void**>(&_d_midline)));
        check(trexhip_device_alloc(_ctx, N * sizeof(trexhip_midline_info), reinterpret_cast<void**>(&_d_minfo)));
        _cap = n; _cap_points = s.max_points; _cap_res = (int)s.midline_resolution;
    }
    trexhip_ctx* _ctx;
    float *_d_outline = nullptr, *_d_segments = nullptr, *_d_midline = nullptr;
    trexhip_posture_info* _d_pinfo = nullptr;
    trexhip_midline_info* _d_minfo = nullptr;
    uint8_t* _d_crops = nullptr;
    size_t _crops_cap = 0;
    int _cap = 0, _cap_points = 0, _cap_res = 0;
    std::vector<trexhip_midline_info> _minfo;
};

}  // namespace track
