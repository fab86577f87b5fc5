// HipVINetwork.h -- identity-network facade on libtrexhip with the surface of Python::VINetwork
// (Application/src/tracker/ml/VisualIdentification.h:32,115-133,189):
//   probabilities(std::vector<Image::Ptr>&&, callback)  -> callback(values[N][classes], indexes[N])
//   transform_results (VisualIdentification.cpp:809-830)  N x M flat, -1 for missing rows
//   batch size rule (VisualIdentification.cpp:112-118)
// Errors surface as exceptions through the returned future, like SoftException does in the reference (:481-484).
#pragma once
#ifdef TREXHIP_WITH_TREX
#include <commons.pc.h>
#include <misc/Image.h>
#else
#include "trex_types.h"
#endif
#include <cstring>
#include <functional>
#include <future>
#include <map>
#include <string>
#include <vector>
#include "../../include/trexhip.h"

namespace track {

struct HipVINetwork {
    using callback_t = std::function<void(std::vector<std::vector<float>>&&, std::vector<float>&&)>;

    explicit HipVINetwork(int device = 0) {
        trexhip_params p;
        trexhip_default_params(&p, 64, 64);      // the detect buffers are unused by this facade
        p.device = device; p.max_batch = 1; p.max_runs = 64; p.max_blobs = 1; p.max_pixels = 64;
        check(trexhip_create(&p, &_ctx));
    }
    ~HipVINetwork() { trexhip_destroy(_ctx); }
    HipVINetwork(const HipVINetwork&) = delete;

    // VINetwork::load_weights: flat blob made by tools/convert_weights.py / trex_amd/weights.py
    void load_weights(const void* blob, size_t bytes) { check(trexhip_load_weights(_ctx, blob, bytes)); }
    bool weights_loaded() const { return trexhip_num_classes(_ctx) > 0; }
    int num_classes() const { return trexhip_num_classes(_ctx); }

    static size_t batch_size_for(size_t n_ids) {        // VisualIdentification.cpp:112-118
        size_t b = n_ids > 64 ? n_ids : 64;
        if (b < 128) { size_t p = 1; while (p < b) p <<= 1; return p; }
        return 128;
    }

    std::future<void> probabilities(std::vector<cmn::Image::Ptr>&& images, callback_t&& callback) {
        std::promise<void> prom;
        auto fut = prom.get_future();
        try {
            if (!weights_loaded()) throw std::runtime_error("weights not loaded");
            const int C = num_classes();
            std::vector<uint8_t> crops;
            const size_t per = (size_t)80 * 80 * (size_t)trexhip_network_channels(_ctx);     // what the network was built for
            for (auto& im : images) {
                if (!im) throw std::runtime_error("null image");
                if (im->rows != 80 || im->cols != 80) throw std::runtime_error("Invalid image size (expected individual_image_size 80x80)");  // visual_recognition_torch.py:1006-1018
                if (im->size() != per) throw std::runtime_error("Invalid image channels (the network expects " + std::to_string(trexhip_network_channels(_ctx)) + ")");
                crops.insert(crops.end(), im->data(), im->data() + im->size());
            }
            std::vector<float> flat(images.size() * (size_t)C);
            if (!images.empty()) check(trexhip_identify(_ctx, crops.data(), (int32_t)images.size(), flat.data()));
            std::vector<std::vector<float>> values(images.size());
            std::vector<float> indexes(images.size());
            for (size_t i = 0; i < images.size(); ++i) {
                values[i].assign(flat.begin() + i * C, flat.begin() + (i + 1) * C);
                indexes[i] = (float)i;                                        // visual_recognition_torch.py:1024-1028
            }
            callback(std::move(values), std::move(indexes));
            prom.set_value();
        } catch (...) { prom.set_exception(std::current_exception()); }
        return fut;
    }

    // VisualIdentification.cpp:809-830
    static std::vector<float> transform_results(size_t n_images, const std::vector<float>& indexes,
                                                const std::vector<std::vector<float>>& values) {
        const size_t M = values.empty() ? 0 : values.front().size();
        std::vector<float> out(n_images * M, -1.f);
        for (size_t i = 0; i < values.size() && i < indexes.size(); ++i) {
            const size_t idx = (size_t)indexes[i];
            if (idx < n_images) std::copy(values[i].begin(), values[i].end(), out.begin() + idx * M);
        }
        return out;
    }

    // synchronous form (VisualIdentification.h:110-133): N x M flat, rows of images without a result are -1
    std::vector<float> probabilities(std::vector<cmn::Image::Ptr>&& images) {
        const size_t N = images.size();
        std::vector<float> flat;
        probabilities(std::move(images), [&](std::vector<std::vector<float>>&& values, std::vector<float>&& indexes) {
            flat = transform_results(N, indexes, values);
        }).get();
        return flat;
    }

    // VINetwork::paverages (VisualIdentification.h:145-180): mean probability row per individual id over its images
    struct Average { float samples = 0; std::vector<float> values; };
    template <typename Idx>
    std::map<Idx, Average> paverages(const std::vector<Idx>& ids, std::vector<cmn::Image::Ptr>&& images) {
        const auto probs = probabilities(std::move(images));
        const size_t M = (size_t)num_classes();
        std::map<Idx, Average> averages;
        for (size_t i = 0; i < ids.size() && (i + 1) * M <= probs.size(); ++i) {
            Average& av = averages[ids[i]];
            if (av.values.empty()) av.values.assign(M, 0.f);
            ++av.samples;
            for (size_t k = 0; k < M; ++k) av.values[k] += probs[i * M + k];
        }
        for (auto& kv : averages) for (float& v : kv.second.values) v /= kv.second.samples;
        return averages;
    }

    // ---- training -------------------------------------------------------------------------------------------
    // VINetwork::train (VisualIdentification.cpp:496-806) hands TrainingData to Python's start_learning(), whose batch loop
    // (visual_recognition_torch.py:1100-1158) is what a Trainer step replaces; epochs, validation, the learning-rate schedule and
    // early stopping stay with the caller, as they are host logic in the reference too.
    struct Trainer {
        // weights: the same blob as load_weights (the reference starts from the network's current state_dict)
        // exact_fp32: conv2 / conv3 forward and data gradients on the fp32 matrix cores instead of the fp16 two-piece split arithmetic
        // (trexhip_train_params::precision; same parity bars, 1.6 instead of 1.2 ms per 128-sample step)
        Trainer(HipVINetwork& net, const void* blob, size_t bytes, int max_batch, float learning_rate = 0.001f, uint64_t seed = 0, bool exact_fp32 = false) : _net(net) {
            trexhip_train_params p{};
            p.precision = exact_fp32 ? 1 : 0;
            p.max_batch = max_batch; p.lr = learning_rate; p.beta1 = 0.9f; p.beta2 = 0.999f; p.eps = 1e-8f; p.bn_momentum = 0.1f; p.dropout = 0.05f; p.seed = seed;
            check(trexhip_trainer_create(net._ctx, blob, bytes, &p, &_t));
            std::memcpy(&_classes, static_cast<const char*>(blob) + 8, 4);
            std::memcpy(&_channels, static_cast<const char*>(blob) + 20, 4);
        }
        ~Trainer() { trexhip_trainer_destroy(_t); }
        Trainer(const Trainer&) = delete;
        struct Result { float loss; int correct; };
        // one optimizer step on a batch as the data loader yields it: NHWC float32 in [0, 255], class indices
        Result train_batch(const float* images, const int32_t* labels, int n) {
            Result r{};
            check(trexhip_train_step(_t, images, labels, n, nullptr, &r.loss, &r.correct));
            return r;
        }
        // validation batch in eval mode (running statistics, no dropout): the val_loss / val_acc of train() :1171-1206
        Result evaluate(const float* images, const int32_t* labels, int n) {
            Result r{};
            check(trexhip_train_eval(_t, images, labels, n, &r.loss, &r.correct));
            return r;
        }
        void set_learning_rate(float lr) { check(trexhip_trainer_set_lr(_t, lr)); }          // ReduceLROnPlateau lives with the caller
        int64_t steps() const { return trexhip_trainer_steps(_t); }
        std::vector<uint8_t> weights() const {                                                // what the reference serialises back (state_dict)
            std::vector<uint8_t> blob(trexhip_weight_blob_bytes(_classes, _channels));
            size_t got = 0;
            check(trexhip_trainer_export(_t, blob.data(), blob.size(), &got));
            return blob;
        }
        void apply() { const auto w = weights(); _net.load_weights(w.data(), w.size()); }     // the inference path continues with the trained weights
    private:
        HipVINetwork& _net;
        trexhip_trainer* _t = nullptr;
        int32_t _classes = 0, _channels = 1;
    };

private:
    trexhip_ctx* _ctx = nullptr;
    static void check(int rc) { if (rc != 0) throw std::runtime_error(std::string("libtrexhip: ") + trexhip_last_error()); }
};

}  // namespace track
