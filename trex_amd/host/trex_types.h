// trex_types.h -- minimal stand-ins for the TRex/commons types that cross the detection and identity
// boundaries.  ONLY used when this repository is built on its own (tests/cpp).  Inside a TRex build
// define TREXHIP_WITH_TREX and the adapter headers include TRex's real headers instead
// (INTEGRATION.md).  Member names and semantics follow the reference:
//   cmn::Image             commons misc/Image.h [not in tree]: rows, cols, dims, data()
//   HorizontalLine{y,x0,x1} Application/Tests/test_pixels.cpp:994-995 (ctor order), pv.cpp:505-509
//   blob::Pair             Application/src/ProcessedVideo/pv.cpp:491-529, Tests/test_matching.cpp:1577
//   pv::Frame              Application/src/ProcessedVideo/pv.h:114-192 (add_object, set_encoding, n, mask(), pixels())
//   SegmentationData       Application/src/tracker/core/TaskPipeline.h:87-118
//   TileImage              Application/src/tracker/core/TileImage.h:33-75, TileImage.cpp:13-21 (dtor contract)
//   detect::BackendHooks   Application/src/tracker/python/BackendRegistry.h:10-24
//   PipelineManager / register_pipeline  Application/src/tracker/core/TaskPipeline.h:200-330, python/PipelineRegistry.h:12-33 (synchronous stand-in)
//   track::Outline/Midline/MidlineSegment  Application/src/tracker/tracking/Outline.h:241-300 (members used by HipPosture.h only)
#pragma once
#include <cstdint>
#include <functional>
#include <future>
#include <map>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <vector>

namespace cmn {

struct Image {
    using Ptr = std::unique_ptr<Image>;
    uint32_t rows = 0, cols = 0, dims = 0;
    std::vector<uint8_t> storage;
    static Ptr Make(uint32_t rows, uint32_t cols, uint32_t dims) {
        auto p = std::make_unique<Image>();
        p->rows = rows; p->cols = cols; p->dims = dims;
        p->storage.assign((size_t)rows * cols * dims, 0);
        return p;
    }
    uint8_t* data() { return storage.data(); }
    const uint8_t* data() const { return storage.data(); }
    size_t size() const { return storage.size(); }
    void set_to(uint8_t v) { std::fill(storage.begin(), storage.end(), v); }
};

struct HorizontalLine {
    uint16_t x0 = 0, x1 = 0, y = 0, padding = 0;
    HorizontalLine() = default;
    HorizontalLine(uint16_t y_, uint16_t x0_, uint16_t x1_) : x0(x0_), x1(x1_), y(y_) {}
    bool operator==(const HorizontalLine& o) const { return x0 == o.x0 && x1 == o.x1 && y == o.y; }
    bool operator<(const HorizontalLine& o) const { return y < o.y || (y == o.y && x1 < o.x0); }
};

using PixelArray_t = std::vector<uint8_t>;

struct Vec2 {
    float x = 0, y = 0;
    Vec2() = default;
    Vec2(float x_, float y_) : x(x_), y(y_) {}
    bool operator==(const Vec2& o) const { return x == o.x && y == o.y; }
};
using Float2_t = float;

enum class meta_encoding_t { gray, r3g3b2, rgb8, binary };

namespace blob {
struct Prediction { uint8_t clid = 255, p = 0; bool valid() const { return clid != 255; } };
using line_ptr_t = std::unique_ptr<std::vector<HorizontalLine>>;
using pixel_ptr_t = std::unique_ptr<PixelArray_t>;
struct Pair {
    line_ptr_t lines;
    pixel_ptr_t pixels;
    uint8_t extra_flags = 0;
    Prediction pred;
    Pair() = default;
    Pair(line_ptr_t&& l, pixel_ptr_t&& p, uint8_t flags = 0) : lines(std::move(l)), pixels(std::move(p)), extra_flags(flags) {}
};
}  // namespace blob

}  // namespace cmn

namespace pv {
using namespace cmn;
class Frame {
    std::vector<blob::line_ptr_t> _mask;
    std::vector<blob::pixel_ptr_t> _pixels;
    std::vector<uint8_t> _flags;
    std::vector<blob::Prediction> _predictions;
    uint16_t _n = 0;
    meta_encoding_t _encoding = meta_encoding_t::gray;

public:
    void set_encoding(meta_encoding_t e) { _encoding = e; }
    meta_encoding_t encoding() const { return _encoding; }
    // pv.cpp:491-529
    void add_object(blob::Pair&& pair) {
        if (pair.lines->size() >= UINT16_MAX) throw std::invalid_argument("too many lines");
        if (pair.lines->empty()) return;
        _mask.emplace_back(std::move(pair.lines));
        if (pair.pixels && _encoding != meta_encoding_t::binary) _pixels.push_back(std::move(pair.pixels));
        _flags.push_back(pair.extra_flags);
        _predictions.resize(_flags.size());
        _predictions.back() = pair.pred;
        _n++;
    }
    uint16_t n() const { return _n; }
    const std::vector<blob::line_ptr_t>& mask() const { return _mask; }
    const std::vector<blob::pixel_ptr_t>& pixels() const { return _pixels; }
    const std::vector<uint8_t>& flags() const { return _flags; }
};
}  // namespace pv

struct SegmentationData {
    cmn::Image::Ptr image;     // the original colour frame: must be left untouched (Segmenter.cpp:1209-1212)
    pv::Frame frame;
    operator bool() const { return image != nullptr; }
};

struct Size2 { float width = 0, height = 0; };

struct TileImage {
    Size2 tile_size;
    SegmentationData data;
    std::vector<cmn::Image::Ptr> images;
    Size2 source_size, original_size;
    std::unique_ptr<std::promise<SegmentationData>> promise;
    std::function<void()> callback;
    TileImage() = default;
    TileImage(TileImage&&) = default;
    TileImage& operator=(TileImage&&) = default;
    ~TileImage() {   // core/TileImage.cpp:13-21: a live promise at destruction raises inside the future
        if (promise) {
            try { throw std::runtime_error("TileImage destroyed with a pending promise"); }
            catch (...) { promise->set_exception(std::current_exception()); }
        }
    }
};

namespace buffers {
// core/TileBuffers.h: pool of tile images; apply() must hand every image back (BackgroundSubtraction.cpp:336-339)
struct TileBuffers {
    std::mutex m;
    std::vector<cmn::Image::Ptr> pool;
    static TileBuffers& get() { static TileBuffers t; return t; }
    void move_back(cmn::Image::Ptr&& p) { std::lock_guard<std::mutex> g(m); pool.emplace_back(std::move(p)); }
    size_t size() { std::lock_guard<std::mutex> g(m); return pool.size(); }
};
}  // namespace buffers

namespace track::detect {
namespace ObjectDetectionType { enum Class { none, yolo, sam3, background_subtraction, precomputed, hip_background_subtraction }; }

struct BackendHooks {   // python/BackendRegistry.h:10-17
    std::function<void()> init;
    std::function<void()> deinit;
    std::function<bool()> is_initializing;
    std::function<double()> fps;
    std::function<void(std::vector<TileImage>&&)> apply;
    std::function<void(const cmn::Image::Ptr&)> set_background;
};
inline std::map<int, BackendHooks>& registry() { static std::map<int, BackendHooks> r; return r; }
inline void register_backend(ObjectDetectionType::Class type, BackendHooks hooks) { registry()[type] = std::move(hooks); }   // BackendRegistry.cpp:23-25
inline void unregister_backend(ObjectDetectionType::Class type) { registry().erase(type); }
inline const BackendHooks* backend(ObjectDetectionType::Class type) { auto it = registry().find(type); return it == registry().end() ? nullptr : &it->second; }
}  // namespace track::detect

// PipelineManager<Data> (core/TaskPipeline.h): collects enqueued items into batches and hands them to the backend's callback
// on a worker thread; `paused` holds them back until the backend is ready.  This stand-in keeps the same interface and
// ordering but runs the callback synchronously inside enqueue() / set_paused(false).
template <typename Data>
class PipelineManager {
public:
    PipelineManager(size_t batch_size, bool start_paused, std::function<void(std::vector<Data>&&)> cb)
        : _batch(batch_size ? batch_size : 1), _paused(start_paused), _cb(std::move(cb)) {}
    void enqueue(Data&& d) { _queue.emplace_back(std::move(d)); if (!_paused) flush(); }
    void set_paused(bool v) { _paused = v; if (!_paused) flush(); }
    bool is_paused() const { return _paused; }
    bool is_terminated() const { return _terminated; }
    void terminate() { _terminated = true; }
    void clean_up() { _queue.clear(); }
    size_t pending() const { return _queue.size(); }
private:
    void flush() {
        while (!_queue.empty() && !_terminated) {
            std::vector<Data> batch;
            while (!_queue.empty() && batch.size() < _batch) { batch.emplace_back(std::move(_queue.front())); _queue.erase(_queue.begin()); }
            _cb(std::move(batch));
        }
    }
    size_t _batch; bool _paused, _terminated = false;
    std::function<void(std::vector<Data>&&)> _cb;
    std::vector<Data> _queue;
};

namespace track::detect {
inline std::map<int, std::unique_ptr<PipelineManager<TileImage>>>& pipelines() { static std::map<int, std::unique_ptr<PipelineManager<TileImage>>> m; return m; }
inline void register_pipeline(ObjectDetectionType::Class type, size_t batch_size, bool start_paused, std::function<void(std::vector<TileImage>&&)> cb) {   // PipelineRegistry.h:12-16
    pipelines()[type] = std::make_unique<PipelineManager<TileImage>>(batch_size, start_paused, std::move(cb));
}
inline void unregister_pipeline(ObjectDetectionType::Class type) { pipelines().erase(type); }                                                                // :19
inline PipelineManager<TileImage>* try_pipeline_manager(ObjectDetectionType::Class type) { auto it = pipelines().find(type); return it == pipelines().end() ? nullptr : it->second.get(); }
inline PipelineManager<TileImage>& pipeline_manager(ObjectDetectionType::Class type) {                                                                        // :22
    auto* m = try_pipeline_manager(type);
    if (!m) throw std::runtime_error("no pipeline registered for this detection type");
    return *m;
}
}  // namespace track::detect

namespace track {
struct MidlineSegment {                                  // tracking/Outline.h:241-250
    cmn::Float2_t height = 0, l_length = 0;
    cmn::Vec2 pos;
};
class Midline {                                          // tracking/Outline.h:252-300 (GETTER_NCONST members)
public:
    using Ptr = std::unique_ptr<Midline>;
    cmn::Float2_t& len() { return _len; }
    cmn::Float2_t& angle() { return _angle; }
    cmn::Vec2& offset() { return _offset; }
    std::vector<MidlineSegment>& segments() { return _segments; }
    long& head_index() { return _head_index; }
    long& tail_index() { return _tail_index; }
    bool& is_normalized() { return _is_normalized; }
    bool empty() const { return _segments.empty(); }
    size_t size() const { return _segments.size(); }
private:
    cmn::Float2_t _len = 0, _angle = 0;
    cmn::Vec2 _offset;
    std::vector<MidlineSegment> _segments;
    long _head_index = -1, _tail_index = -1;
    bool _is_normalized = false;
};
class Outline {                                          // tracking/Outline.h (replace_points / points / size / empty)
public:
    void replace_points(std::unique_ptr<std::vector<cmn::Vec2>>&& p) { _points = std::move(p); }
    const std::vector<cmn::Vec2>& points() const { static const std::vector<cmn::Vec2> none; return _points ? *_points : none; }
    size_t size() const { return _points ? _points->size() : 0; }
    bool empty() const { return size() == 0; }
private:
    std::unique_ptr<std::vector<cmn::Vec2>> _points;
};
}  // namespace track
