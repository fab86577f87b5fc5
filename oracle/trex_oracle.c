/*
 * trex_oracle.c -- CPU restatement of the TRex background-subtraction detect stage and the
 * track-stage re-threshold.  TEST INFRASTRUCTURE ONLY (see trex_oracle.h).
 *
 * What it follows (reference = /root/reference, read-only):
 *   BackgroundSubtraction::apply         Application/src/tracker/python/BackgroundSubtraction.cpp:126-347
 *     RawProcessing::generate_binary     call at :209   (body in un-vendored commons; restated)
 *     CPULabeling::run                   call at :216   (body in un-vendored commons; restated)
 *     size filter                        :245-291, SizeFilters::in_range_of_one core/SizeFilters.cpp:37-53
 *     drop blobs with >= 65535 runs      :305-313
 *   pv::Frame::add_object invariants     Application/src/ProcessedVideo/pv.cpp:491-529
 *   threshold semantics                  Application/Tests/test_pixels.cpp:981-1071,1611-1821
 *
 * The code is deliberately plain: row scan -> runs -> classic two-pass union-find over runs ->
 * blobs ordered by their first run in raster order, runs inside a blob sorted by (y,x0).
 */
#include "trex_oracle.h"
#include <stdlib.h>
#include <string.h>
#include <math.h>
#ifdef _OPENMP
#include <omp.h>
#endif

struct oracle_frame {
    int32_t n_blobs, n_runs, n_pixels;
    oracle_blob* blobs;
    oracle_run* runs;
    uint8_t* pixels;
};

/* ------------------------------------------------------------------ pixel passes */

/* OpenCV getStructuringElement(MORPH_ELLIPSE, Size(k,k)) restated from the published
 * algorithm (opencv/modules/imgproc/src/morph.dispatch.cpp); returns k*k bytes (0/1). */
static uint8_t* ellipse_element(int k) {
    uint8_t* e = (uint8_t*)calloc((size_t)k * k, 1);
    int r = k / 2, c = k / 2;
    double inv_r2 = r ? 1.0 / ((double)r * r) : 0.0;
    for (int i = 0; i < k; ++i) {
        int j1 = 0, j2 = 0, dy = i - r;
        if (abs(dy) <= r) {
            int dx = (int)lrint(c * sqrt((r * r - dy * dy) * inv_r2)); /* cvRound */
            j1 = c - dx > 0 ? c - dx : 0;
            j2 = c + dx + 1 < k ? c + dx + 1 : k;
        }
        for (int j = j1; j < j2; ++j) e[i * k + j] = 1;
    }
    return e;
}

/* binary dilate/erode with a k*k element anchored at its centre; pixels outside the image
 * are ignored (OpenCV default morphology border). */
static void morph(const uint8_t* in, uint8_t* out, int W, int H, const uint8_t* el, int k, int dilate) {
    int a = k / 2;
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            int v = dilate ? 0 : 1;
            for (int i = 0; i < k; ++i) {
                int yy = y + i - a;
                if (yy < 0 || yy >= H) continue;
                for (int j = 0; j < k; ++j) {
                    if (!el[i * k + j]) continue;
                    int xx = x + j - a;
                    if (xx < 0 || xx >= W) continue;
                    if (dilate) v |= in[(size_t)yy * W + xx] != 0;
                    else        v &= in[(size_t)yy * W + xx] != 0;
                }
            }
            out[(size_t)y * W + x] = v ? 255 : 0;
        }
}

static inline int diff_value(int px, int bg, const oracle_params* p) {
    if (!p->enable_difference) return px;                 /* threshold raw grey values */
    if (p->absolute_difference) return abs(bg - px);      /* cv::absdiff            */
    return bg - px > 0 ? bg - px : 0;                     /* cv::subtract(avg, in), saturating */
}

static inline int passes(int d, const oracle_params* p) {
    if (p->threshold_maximum < 255)                       /* cv::inRange(thr, max): inclusive both ends */
        return d >= abs(p->threshold) && d <= p->threshold_maximum;
    return p->inclusive ? d >= abs(p->threshold) : d > abs(p->threshold); /* inclusive: "disregards any pixel |p| < threshold" (core/default_config.cpp:1168); else cv::threshold BINARY, strict */
}

/* mask = 255 where the pixel survives threshold (+ morphology); grey = (inverted) input value */
static void binary_and_grey(const uint8_t* frame, const uint8_t* bg, uint8_t* mask, uint8_t* grey,
                            const oracle_params* p) {
    const int W = p->width, H = p->height;
    const size_t N = (size_t)W * H;
    for (size_t i = 0; i < N; ++i) {
        int px = p->image_invert ? 255 - frame[i] : frame[i];
        grey[i] = (uint8_t)px;
        mask[i] = passes(diff_value(px, bg[i], p), p) ? 255 : 0;
    }
    if (p->use_closing && p->closing_size > 0) {
        uint8_t* el = ellipse_element(p->closing_size);
        uint8_t* tmp = (uint8_t*)malloc(N);
        morph(mask, tmp, W, H, el, p->closing_size, 1);
        morph(tmp, mask, W, H, el, p->closing_size, 0);
        free(tmp); free(el);
    }
    if (p->dilation_size != 0) {
        int k = 2 * abs(p->dilation_size) + 1;
        uint8_t* el = ellipse_element(k);
        uint8_t* tmp = (uint8_t*)malloc(N);
        morph(mask, tmp, W, H, el, k, p->dilation_size > 0);
        memcpy(mask, tmp, N);
        free(tmp); free(el);
    }
}

void oracle_generate_binary(const uint8_t* frame, const uint8_t* bg, uint8_t* out, const oracle_params* p) {
    const size_t N = (size_t)p->width * p->height;
    uint8_t* mask = (uint8_t*)malloc(N);
    binary_and_grey(frame, bg, mask, out, p);
    for (size_t i = 0; i < N; ++i) out[i] = mask[i] ? out[i] : 0;   /* grey under mask */
    free(mask);
}

/* ------------------------------------------------------------------ labelling */

typedef struct { int32_t* parent; } uf_t;
static int32_t uf_find(int32_t* parent, int32_t a) {
    while (parent[a] != a) { parent[a] = parent[parent[a]]; a = parent[a]; }
    return a;
}
static void uf_union(int32_t* parent, int32_t a, int32_t b) {
    a = uf_find(parent, a); b = uf_find(parent, b);
    if (a == b) return;
    if (a < b) parent[b] = a; else parent[a] = b;   /* root = smallest raster index */
}

uint32_t oracle_bid(uint32_t x0, uint32_t x1, uint32_t y, uint32_t n_runs) {
    /* pv::bid::from_blob: 13/13/6-bit hash of the first line (commons, not in tree).  Layout inferred
     * from Application/Tests/test_matching.cpp:432-435 (from_data(x,x,y,n), clamps 0..8191 / 1..63) and
     * verified numerically on videos/compare_data_automatic/test_fish*.csv: x = x0 + (x1-x0+1)/2. */
    uint32_t x = x0 + (x1 - x0 + 1) / 2;
    if (x > 8191) x = 8191;
    if (y > 8191) y = 8191;
    uint32_t n = n_runs < 1 ? 1 : (n_runs > 63 ? 63 : n_runs);
    return (x << 19) | (y << 6) | n;
}

static int size_ok(int64_t num_pixels, const oracle_params* p) {
    if (p->n_ranges <= 0) return 1;                         /* SizeFilters.cpp:38 empty => true */
    float sqcm = (float)(p->cm_per_pixel * p->cm_per_pixel); /* BackgroundSubtraction.cpp:139 (Float2_t) */
    double v = (double)((float)num_pixels * sqcm);
    for (int i = 0; i < p->n_ranges; ++i)
        if (v >= p->ranges[2 * i] && v < p->ranges[2 * i + 1]) return 1; /* Range::contains, half open */
    return 0;
}

/* label runs of a binary image (non-zero = foreground) and build the blob tables.
 * img_px supplies the grey values that are gathered into the pixel array. */
static oracle_frame* label_image(const uint8_t* bin, const uint8_t* img_px, int W, int H,
                                 int connectivity, const oracle_params* filter) {
    /* pass 1: runs in raster order */
    size_t cap = 1024, n = 0;
    oracle_run* runs = (oracle_run*)malloc(cap * sizeof(oracle_run));
    int32_t* row_start = (int32_t*)malloc((size_t)(H + 1) * sizeof(int32_t));
    for (int y = 0; y < H; ++y) {
        row_start[y] = (int32_t)n;
        const uint8_t* r = bin + (size_t)y * W;
        int x = 0;
        while (x < W) {
            if (!r[x]) { ++x; continue; }
            int x0 = x;
            while (x < W && r[x]) ++x;
            if (n == cap) { cap *= 2; runs = (oracle_run*)realloc(runs, cap * sizeof(oracle_run)); }
            runs[n].x0 = (uint16_t)x0; runs[n].x1 = (uint16_t)(x - 1); runs[n].y = (uint16_t)y; runs[n].pad = 0;
            ++n;
        }
    }
    row_start[H] = (int32_t)n;
    /* pass 2: union runs of adjacent rows that touch (8-conn: overlap +-1; 4-conn: overlap) */
    int32_t* parent = (int32_t*)malloc((n ? n : 1) * sizeof(int32_t));
    for (size_t i = 0; i < n; ++i) parent[i] = (int32_t)i;
    const int slack = connectivity == 8 ? 1 : 0;
    for (int y = 1; y < H; ++y) {
        int32_t a = row_start[y - 1], ae = row_start[y], b = row_start[y], be = row_start[y + 1];
        while (a < ae && b < be) {
            if ((int)runs[a].x1 + slack >= (int)runs[b].x0 && (int)runs[b].x1 + slack >= (int)runs[a].x0)
                uf_union(parent, a, b);
            if (runs[a].x1 < runs[b].x1) ++a; else ++b;
        }
    }
    /* pass 3: blobs in order of their first run; per-blob run lists stay in raster order */
    int32_t* blob_of = (int32_t*)malloc((n ? n : 1) * sizeof(int32_t));
    int32_t nb = 0;
    for (size_t i = 0; i < n; ++i) {
        int32_t r = uf_find(parent, (int32_t)i);
        if ((size_t)r == i) blob_of[i] = nb++;
    }
    for (size_t i = 0; i < n; ++i) blob_of[i] = blob_of[uf_find(parent, (int32_t)i)];
    uint32_t* cnt_runs = (uint32_t*)calloc((size_t)nb + 1, sizeof(uint32_t));
    uint64_t* cnt_px = (uint64_t*)calloc((size_t)nb + 1, sizeof(uint64_t));
    for (size_t i = 0; i < n; ++i) {
        cnt_runs[blob_of[i]]++;
        cnt_px[blob_of[i]] += (uint64_t)(runs[i].x1 - runs[i].x0 + 1);
    }
    /* keep = size filter (BackgroundSubtraction.cpp:259) and < UINT16_MAX lines (:306) */
    int32_t* new_idx = (int32_t*)malloc(((size_t)nb + 1) * sizeof(int32_t));
    oracle_frame* f = (oracle_frame*)calloc(1, sizeof(oracle_frame));
    int32_t kept = 0; uint64_t kept_runs = 0, kept_px = 0;
    for (int32_t b = 0; b < nb; ++b) {
        int keep = (!filter || size_ok((int64_t)cnt_px[b], filter)) && cnt_runs[b] < 65535u;
        new_idx[b] = keep ? kept++ : -1;
        if (keep) { kept_runs += cnt_runs[b]; kept_px += cnt_px[b]; }
    }
    f->n_blobs = kept; f->n_runs = (int32_t)kept_runs; f->n_pixels = (int32_t)kept_px;
    f->blobs = (oracle_blob*)calloc((size_t)kept + 1, sizeof(oracle_blob));
    f->runs = (oracle_run*)malloc((kept_runs + 1) * sizeof(oracle_run));
    f->pixels = (uint8_t*)malloc(kept_px + 1);
    uint32_t ro = 0, po = 0;
    for (int32_t b = 0; b < nb; ++b) {
        int32_t k = new_idx[b];
        if (k < 0) continue;
        oracle_blob* B = &f->blobs[k];
        B->run_begin = ro; B->pix_begin = po; B->n_runs = 0; B->n_pixels = 0; B->parent = 0xFFFFFFFFu; B->flags = 0;
        B->x0 = B->y0 = 0xFFFF; B->x1 = B->y1 = 0;
        ro += cnt_runs[b]; po += (uint32_t)cnt_px[b];
    }
    uint32_t* pmin = (uint32_t*)malloc(((size_t)kept + 1) * sizeof(uint32_t));
    uint32_t* pmax = (uint32_t*)calloc((size_t)kept + 1, sizeof(uint32_t));
    for (int32_t k = 0; k < kept; ++k) pmin[k] = 255;
    for (size_t i = 0; i < n; ++i) {
        int32_t k = new_idx[blob_of[i]];
        if (k < 0) continue;
        oracle_blob* B = &f->blobs[k];
        oracle_run r = runs[i];
        f->runs[B->run_begin + B->n_runs++] = r;
        if (r.x0 < B->x0) B->x0 = r.x0;
        if (r.x1 > B->x1) B->x1 = r.x1;
        if (r.y < B->y0) B->y0 = r.y;
        if (r.y > B->y1) B->y1 = r.y;
        for (uint32_t x = r.x0; x <= r.x1; ++x) {
            uint8_t p = img_px[(size_t)r.y * W + x];
            f->pixels[B->pix_begin + B->n_pixels++] = p;
            B->m10 += x; B->m01 += r.y;
            B->m20 += (uint64_t)x * x; B->m11 += (uint64_t)x * r.y; B->m02 += (uint64_t)r.y * r.y;
            B->sp += p; B->spx += (uint64_t)p * x; B->spy += (uint64_t)p * r.y;
            if (p < pmin[k]) pmin[k] = p;
            if (p > pmax[k]) pmax[k] = p;
        }
    }
    for (int32_t k = 0; k < kept; ++k) {
        oracle_blob* B = &f->blobs[k];
        oracle_run r0 = f->runs[B->run_begin];
        B->bid = oracle_bid(r0.x0, r0.x1, r0.y, B->n_runs);
        B->px_min_max = pmin[k] | (pmax[k] << 8);
    }
    free(pmin); free(pmax); free(new_idx); free(cnt_runs); free(cnt_px);
    free(blob_of); free(parent); free(row_start); free(runs);
    return f;
}

oracle_frame* oracle_segment(const uint8_t* frame, const uint8_t* bg, const oracle_params* p) {
    const size_t N = (size_t)p->width * p->height;
    uint8_t* mask = (uint8_t*)malloc(N);
    uint8_t* grey = (uint8_t*)malloc(N);
    binary_and_grey(frame, bg, mask, grey, p);     /* BackgroundSubtraction.cpp:209 */
    /* CPULabeling::run labels the grey-under-mask image and gathers pixel values from it (:216):
     * a masked pixel whose grey value is 0 is indistinguishable from background there. */
    if (p->zero_is_background)
        for (size_t i = 0; i < N; ++i) if (!grey[i]) mask[i] = 0;
    oracle_frame* f = label_image(mask, grey, p->width, p->height, p->connectivity, p);
    free(mask); free(grey);
    return f;
}

void oracle_frame_counts(const oracle_frame* f, int32_t* nb, int32_t* nr, int32_t* np) {
    *nb = f->n_blobs; *nr = f->n_runs; *np = f->n_pixels;
}
void oracle_frame_copy(const oracle_frame* f, oracle_blob* blobs, oracle_run* runs, uint8_t* pixels) {
    if (blobs)  memcpy(blobs, f->blobs, (size_t)f->n_blobs * sizeof(oracle_blob));
    if (runs)   memcpy(runs, f->runs, (size_t)f->n_runs * sizeof(oracle_run));
    if (pixels) memcpy(pixels, f->pixels, (size_t)f->n_pixels);
}
void oracle_frame_free(oracle_frame* f) {
    if (!f) return;
    free(f->blobs); free(f->runs); free(f->pixels); free(f);
}

int64_t oracle_segment_batch(const uint8_t* frames, int32_t n, const uint8_t* bg,
                             const oracle_params* p, int32_t threads) {
    int64_t total = 0;
    const size_t N = (size_t)p->width * p->height;
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#pragma omp parallel for schedule(dynamic) reduction(+ : total)
#endif
    for (int32_t i = 0; i < n; ++i) {
        oracle_frame* f = oracle_segment(frames + (size_t)i * N, bg, p);
        total += f->n_blobs + f->n_pixels;
        oracle_frame_free(f);
    }
    (void)threads;
    return total;
}

/* ------------------------------------------------------------------ track stage */

static inline int diff_method(int px, int bg, int method) {
    /* Background::diff<> semantics, Application/Tests/test_pixels.cpp:1642-1646 */
    if (method == 0) return abs(bg - px);
    if (method == 1) return bg - px > 0 ? bg - px : 0;
    return px;
}

int32_t oracle_line_without_grid(const oracle_run* runs, int32_t n_runs, const uint8_t* pixels,
                                 const uint8_t* bg, int32_t bg_stride, int32_t method, int32_t threshold,
                                 oracle_run* out_runs, uint8_t* out_pixels, int32_t* n_out_pixels) {
    int32_t no = 0, np = 0;
    const uint8_t* px = pixels;
    for (int32_t i = 0; i < n_runs; ++i) {
        int open = 0; oracle_run cur = {0, 0, 0, 0};
        for (int x = runs[i].x0; x <= runs[i].x1; ++x, ++px) {
            int b = bg ? bg[(size_t)runs[i].y * bg_stride + x] : 0;
            if (diff_method(*px, b, method) >= threshold) {      /* keep iff diff >= threshold */
                if (!open) { open = 1; cur.x0 = (uint16_t)x; cur.y = runs[i].y; cur.pad = 0; }
                cur.x1 = (uint16_t)x;
                out_pixels[np++] = *px;
            } else if (open) { out_runs[no++] = cur; open = 0; }  /* runs split where pixels fail */
        }
        if (open) out_runs[no++] = cur;
    }
    *n_out_pixels = np;
    return no;
}

oracle_frame* oracle_threshold_blob(const oracle_run* runs, int32_t n_runs, const uint8_t* pixels,
                                    const uint8_t* bg, int32_t bg_stride, int32_t width, int32_t height,
                                    int32_t method, int32_t threshold, int32_t connectivity) {
    /* pixel::threshold_blob (Tracker.cpp:833-837): threshold the blob's own pixels against the
     * background and re-label the survivors.  Restated by painting survivors into a scratch image
     * over the blob's bounding rows and labelling that image. */
    int y0 = 65535, y1 = -1;
    for (int32_t i = 0; i < n_runs; ++i) { if (runs[i].y < y0) y0 = runs[i].y; if (runs[i].y > y1) y1 = runs[i].y; }
    if (n_runs == 0) { y0 = 0; y1 = 0; }
    (void)height;
    size_t rows = (size_t)(y1 - y0 + 1);
    uint8_t* bin = (uint8_t*)calloc(rows * width, 1);
    uint8_t* val = (uint8_t*)calloc(rows * width, 1);
    const uint8_t* px = pixels;
    for (int32_t i = 0; i < n_runs; ++i)
        for (int x = runs[i].x0; x <= runs[i].x1; ++x, ++px) {
            int b = bg ? bg[(size_t)runs[i].y * bg_stride + x] : 0;
            if (diff_method(*px, b, method) >= threshold) {
                bin[(size_t)(runs[i].y - y0) * width + x] = 255;
                val[(size_t)(runs[i].y - y0) * width + x] = *px;
            }
        }
    oracle_frame* f = label_image(bin, val, width, (int)rows, connectivity, NULL);
    for (int32_t i = 0; i < f->n_runs; ++i) f->runs[i].y = (uint16_t)(f->runs[i].y + y0);
    for (int32_t k = 0; k < f->n_blobs; ++k) {
        oracle_blob* B = &f->blobs[k];
        uint64_t n = B->n_pixels;
        /* shift moments from scratch rows back to frame rows */
        B->m02 += 2 * (uint64_t)y0 * B->m01 + n * (uint64_t)y0 * y0;
        B->m11 += (uint64_t)y0 * B->m10;
        B->spy += (uint64_t)y0 * B->sp;
        B->m01 += n * (uint64_t)y0;
        B->y0 = (uint16_t)(B->y0 + y0); B->y1 = (uint16_t)(B->y1 + y0);
        oracle_run r0 = f->runs[B->run_begin];
        B->bid = oracle_bid(r0.x0, r0.x1, r0.y, B->n_runs);
    }
    free(bin); free(val);
    return f;
}


oracle_frame* oracle_rethreshold_frame(const oracle_frame* detect, const uint8_t* frame, const uint8_t* bg, int32_t width,
                                       int32_t height, int32_t method, int32_t threshold, int32_t connectivity,
                                       const double* ranges, int32_t n_ranges, double cm_per_pixel, int32_t invert) {
    /* paint the survivors of every kept detect blob into one scratch image tagged with (blob index + 1), label it,
     * then read each sub-blob's parent back from the tag of its first pixel.  Blobs of different parents can never
     * touch (they were separate 8-connected components), so one labelling pass over the frame is equivalent to
     * threshold_blob per blob (Tracker.cpp:833-837). */
    const size_t N = (size_t)width * height;
    uint8_t* bin = (uint8_t*)calloc(N, 1);
    uint8_t* val = (uint8_t*)calloc(N, 1);
    uint32_t* tag = (uint32_t*)calloc(N, sizeof(uint32_t));
    for (int32_t k = 0; k < detect->n_blobs; ++k) {
        const oracle_blob* B = &detect->blobs[k];
        for (uint32_t i = 0; i < B->n_runs; ++i) {
            const oracle_run r = detect->runs[B->run_begin + i];
            for (int x = r.x0; x <= r.x1; ++x) {
                const size_t o = (size_t)r.y * width + x;
                int p = frame[o];
                if (invert) p = 255 - p;
                if (diff_method(p, bg[o], method) >= threshold) { bin[o] = 255; val[o] = (uint8_t)p; tag[o] = (uint32_t)k + 1; }
            }
        }
    }
    oracle_frame* f = label_image(bin, val, width, height, connectivity, NULL);
    float sqcm = (float)(cm_per_pixel * cm_per_pixel);
    for (int32_t k = 0; k < f->n_blobs; ++k) {
        oracle_blob* B = &f->blobs[k];
        const oracle_run r0 = f->runs[B->run_begin];
        B->parent = tag[(size_t)r0.y * width + r0.x0] - 1;
        uint32_t cat = 0;
        if (n_ranges > 0) {
            double v = (double)((float)B->n_pixels * sqcm);
            int in = 0; double mn = ranges[0];
            for (int i = 0; i < n_ranges; ++i) { if (v >= ranges[2 * i] && v < ranges[2 * i + 1]) in = 1; if (ranges[2 * i] < mn) mn = ranges[2 * i]; }
            if (!in) cat = v < mn ? 1u : 2u;       /* Tracker.cpp:908-912 */
        }
        B->flags = cat;
    }
    free(bin); free(val); free(tag);
    return f;
}

/* ------------------------------------------------------------------ normalised crops (moments / posture) */
/* gui::Transform (commons, NOT IN TREE) restated as the SFML-style 3x3 affine it is used as in
 * FilterCache.cpp:50-63 / Outline.cpp:1237-1255: translate / scale / rotate(degrees) post-multiply. */
typedef struct { float m[6]; } aff_t;                        /* [m0 m1 m2; m3 m4 m5] */
static aff_t aff_identity(void) { aff_t a = {{1, 0, 0, 0, 1, 0}}; return a; }
static aff_t aff_mul(aff_t a, aff_t b) {
    aff_t c;
    c.m[0] = a.m[0] * b.m[0] + a.m[1] * b.m[3]; c.m[1] = a.m[0] * b.m[1] + a.m[1] * b.m[4]; c.m[2] = a.m[0] * b.m[2] + a.m[1] * b.m[5] + a.m[2];
    c.m[3] = a.m[3] * b.m[0] + a.m[4] * b.m[3]; c.m[4] = a.m[3] * b.m[1] + a.m[4] * b.m[4]; c.m[5] = a.m[3] * b.m[2] + a.m[4] * b.m[5] + a.m[5];
    return c;
}
static aff_t aff_translate(aff_t a, float x, float y) { aff_t t = {{1, 0, x, 0, 1, y}}; return aff_mul(a, t); }
static aff_t aff_scale(aff_t a, float s) { aff_t t = {{s, 0, 0, 0, s, 0}}; return aff_mul(a, t); }
static aff_t aff_rotate_deg(aff_t a, float deg) {
    const float rad = deg * 3.141592654f / 180.f;
    /* correctly rounded float cos / sin (double, narrowed): the one definition CPU, host and device can all meet */
    const float c = (float)cos((double)rad), s = (float)sin((double)rad);
    aff_t t = {{c, -s, 0, s, c, 0}};
    return aff_mul(a, t);
}

/* normalize_image's transform (FilterCache.cpp:50-63): translate(size/2) . scale . translate(len*0.4 | (-len/2,0)) . tr */
void oracle_normalize_transform(const float* tr6, float midline_length, int32_t use_legacy, int32_t out_w, int32_t out_h,
                                float scale, float* M6) {
    aff_t t = aff_identity();
    t = aff_translate(t, (float)out_w * 0.5f, (float)out_h * 0.5f);
    t = aff_scale(t, scale);
    if (use_legacy) t = aff_translate(t, -midline_length * 0.5f, 0.f);
    else            t = aff_translate(t, midline_length * 0.4f, midline_length * 0.4f);       /* Vec2(scalar) */
    aff_t in; memcpy(in.m, tr6, sizeof(in.m));
    t = aff_mul(t, in);
    memcpy(M6, t.m, sizeof(t.m));
}

/* individual_image_normalization = moments (FilterCache.cpp:276-288): rotate(DEGREE(-orientation + pi/4)) . translate(-size/2)
 * with pv::Blob::orientation() = 0.5 * atan2(2 mu11, mu20 - mu02) from the blob's central moments [commons, restated] */
void oracle_moments_transform(const oracle_blob* B, float* tr6) {
    const float n = (float)B->n_pixels;
    const float cx = (float)B->m10 / n, cy = (float)B->m01 / n;
    const float mu20 = (float)B->m20 / n - cx * cx, mu02 = (float)B->m02 / n - cy * cy, mu11 = (float)B->m11 / n - cx * cy;
    const float orientation = 0.5f * (float)atan2((double)(2.f * mu11), (double)(mu20 - mu02));
    const float angle = (-orientation + 3.14159265358979323846f * 0.25f) * 180.f / 3.14159265358979323846f;   /* DEGREE() */
    aff_t t = aff_identity();
    t = aff_rotate_deg(t, angle);
    t = aff_translate(t, -(float)(B->x1 - B->x0 + 1) * 0.5f, -(float)(B->y1 - B->y0 + 1) * 0.5f);
    memcpy(tr6, t.m, sizeof(t.m));
}

/* cv::warpAffine(src, dst, M, dsize, INTER_LINEAR, BORDER_CONSTANT 0) for 8-bit single channel, restated from OpenCV's
 * published fixed-point path (imgwarp.cpp: AB_BITS 10, INTER_BITS 5, INTER_REMAP_COEF_BITS 15) -- unpinned: no OpenCV here */
void oracle_warp_affine_u8(const uint8_t* src, int32_t sw, int32_t sh, const float* M6, uint8_t* dst, int32_t dw, int32_t dh) {
    double M[6];
    for (int i = 0; i < 6; ++i) M[i] = M6[i];
    double D = M[0] * M[4] - M[1] * M[3];
    D = D != 0 ? 1. / D : 0;
    const double A11 = M[4] * D, A22 = M[0] * D;
    M[0] = A11; M[1] *= -D; M[3] *= -D; M[4] = A22;
    const double b1 = -M[0] * M[2] - M[1] * M[5], b2 = -M[3] * M[2] - M[4] * M[5];
    M[2] = b1; M[5] = b2;
    const int AB_SCALE = 1024, round_delta = 16;
    for (int y = 0; y < dh; ++y) {
        const int X0 = (int)lrint((M[1] * y + M[2]) * AB_SCALE) + round_delta;
        const int Y0 = (int)lrint((M[4] * y + M[5]) * AB_SCALE) + round_delta;
        for (int x = 0; x < dw; ++x) {
            const int X = (X0 + (int)lrint(M[0] * x * AB_SCALE)) >> 5, Y = (Y0 + (int)lrint(M[3] * x * AB_SCALE)) >> 5;
            const int sx = X >> 5, sy = Y >> 5, fx = X & 31, fy = Y & 31;
            int v[4];
            for (int k = 0; k < 4; ++k) {
                const int xx = sx + (k & 1), yy = sy + (k >> 1);
                v[k] = (xx >= 0 && xx < sw && yy >= 0 && yy < sh) ? src[(size_t)yy * sw + xx] : 0;
            }
            const int w00 = (32 - fy) * (32 - fx) * 32, w01 = (32 - fy) * fx * 32, w10 = fy * (32 - fx) * 32, w11 = fy * fx * 32;
            dst[(size_t)y * dw + x] = (uint8_t)((v[0] * w00 + v[1] * w01 + v[2] * w10 + v[3] * w11 + (1 << 14)) >> 15);
        }
    }
}

/* cv::warpAffine(INTER_NEAREST, BORDER_CONSTANT 0) as OpenCV computes it: the same fixed-point coordinates with
 * round_delta = AB_SCALE / 2 and X = (X0 + adelta) >> AB_BITS (used for r3g3b2 crops, FilterCache.cpp:70-73) */
void oracle_warp_affine_nearest_u8(const uint8_t* src, int32_t sw, int32_t sh, const float* M6, uint8_t* dst, int32_t dw, int32_t dh) {
    double M[6];
    for (int i = 0; i < 6; ++i) M[i] = M6[i];
    double D = M[0] * M[4] - M[1] * M[3];
    D = D != 0 ? 1. / D : 0;
    const double A11 = M[4] * D, A22 = M[0] * D;
    M[0] = A11; M[1] *= -D; M[3] *= -D; M[4] = A22;
    const double b1 = -M[0] * M[2] - M[1] * M[5], b2 = -M[3] * M[2] - M[4] * M[5];
    M[2] = b1; M[5] = b2;
    const int AB_SCALE = 1024, round_delta = 512;
    for (int y = 0; y < dh; ++y) {
        const int X0 = (int)lrint((M[1] * y + M[2]) * AB_SCALE) + round_delta;
        const int Y0 = (int)lrint((M[4] * y + M[5]) * AB_SCALE) + round_delta;
        for (int x = 0; x < dw; ++x) {
            const int sx = (X0 + (int)lrint(M[0] * x * AB_SCALE)) >> 10, sy = (Y0 + (int)lrint(M[3] * x * AB_SCALE)) >> 10;
            dst[(size_t)y * dw + x] = (sx >= 0 && sx < sw && sy >= 0 && sy < sh) ? src[(size_t)sy * sw + sx] : 0;
        }
    }
}

/* ---- colour encodings (meta_encoding rgb8 / r3g3b2) ---------------------------------------------------------------------------
 * vec_to_r3g3b2 / r3g3b2_to_vec / convert_to_r3g3b2 live in the un-vendored commons; their bit layout is PINNED by the literal
 * vectors of Application/Tests/test_pixels.cpp:629-795: code = (c0 >> 6) << 6 | (c1 >> 5) << 3 | (c2 >> 5) with c0 the FIRST
 * channel in memory (B of a BGR image), decode = {(code >> 6) << 6, ((code >> 3) & 7) << 5, (code & 7) << 5}.
 * The value a colour pixel is thresholded on is its grey value (cmn::bgr2gray, commons): pinned to agree with
 * cv::cvtColor(BGR2GRAY) on the vectors of test_pixels.cpp:1073-1166,1289-1380,1381-1529; restated as OpenCV's 8-bit fixed point. */
uint8_t oracle_vec_to_r3g3b2(uint8_t c0, uint8_t c1, uint8_t c2) { return (uint8_t)(((c0 >> 6) << 6) | ((c1 >> 5) << 3) | (c2 >> 5)); }
void oracle_r3g3b2_to_vec(uint8_t code, uint8_t* out3) {
    out3[0] = (uint8_t)((code >> 6) << 6); out3[1] = (uint8_t)(((code >> 3) & 7) << 5); out3[2] = (uint8_t)((code & 7) << 5);
}
uint8_t oracle_bgr2gray(uint8_t b, uint8_t g, uint8_t r) { return (uint8_t)((b * 1868u + g * 9617u + r * 4899u + 8192u) >> 14); }

/* encoding: 0 gray (1 B/px), 1 r3g3b2 (1 B/px), 2 rgb8 (3 B/px) -- the order of cmn::meta_encoding_t */
static int enc_channels(int enc) { return enc == 2 ? 3 : 1; }
static int diffable(const uint8_t* p, int enc) {
    if (enc == 2) return oracle_bgr2gray(p[0], p[1], p[2]);
    if (enc == 1) { uint8_t v[3]; oracle_r3g3b2_to_vec(p[0], v); return oracle_bgr2gray(v[0], v[1], v[2]); }
    return p[0];
}

/* line_without_grid<InputInfo, OutputInfo{gray}, DifferenceMethod> for the three input encodings (test_pixels.cpp:915-1071):
 * pixels hold enc_channels(pixel_enc) bytes per pixel, the background enc_channels(bg_enc) per pixel (gray or rgb8); a pixel is
 * kept iff diff(grey(bg), grey(px)) >= threshold and keeps all of its bytes. */
int32_t oracle_line_without_grid_enc(const oracle_run* runs, int32_t n_runs, const uint8_t* pixels, int32_t pixel_enc,
                                     const uint8_t* bg, int32_t bg_stride_px, int32_t bg_enc, int32_t method, int32_t threshold,
                                     oracle_run* out_runs, uint8_t* out_pixels, int32_t* n_out_pixels) {
    const int pc = enc_channels(pixel_enc), bc = enc_channels(bg_enc);
    int32_t no = 0, np = 0;
    const uint8_t* px = pixels;
    for (int32_t i = 0; i < n_runs; ++i) {
        int open = 0; oracle_run cur = {0, 0, 0, 0};
        for (int x = runs[i].x0; x <= runs[i].x1; ++x, px += pc) {
            const int b = bg ? diffable(bg + ((size_t)runs[i].y * bg_stride_px + x) * bc, bg_enc) : 0;
            if (diff_method(diffable(px, pixel_enc), b, method) >= threshold) {
                if (!open) { open = 1; cur.x0 = (uint16_t)x; cur.y = runs[i].y; cur.pad = 0; }
                cur.x1 = (uint16_t)x;
                for (int c = 0; c < pc; ++c) out_pixels[np * pc + c] = px[c];
                ++np;
            } else if (open) { out_runs[no++] = cur; open = 0; }
        }
        if (open) out_runs[no++] = cur;
    }
    *n_out_pixels = np;
    return no;
}
