/* trex_pv.c -- CPU restatement (TEST INFRASTRUCTURE ONLY) of the frame body of a .pv file of version V_6:
 *   pv::Frame::serialize   Application/src/ProcessedVideo/pv.cpp:666-703  (timestamp, n, per object start_y / mask_size / lines / pixels)
 *   pv::Frame::read_from   pv.cpp:296-420 with header.version == V_6: compression_flag (:313-316), u64 timestamp (:347-351, V_4),
 *                          u16 n, per object u16 start_y, u16 mask_size, LegacyShortHorizontalLine[mask_size] (:377-388), pixels (:399-403)
 *   LegacyShortHorizontalLine   pv.h:17-52: u16 x0, u16 (x1 << 1) | eol; eol = last line of the current y, the next lines are on y + 1
 * (compress / uncompress themselves are declared in pv.h:27-30 and implemented outside the tree: restated from that description.)
 * Uncompressed frames only; source index / flags / predictions came with V_8 / V_9 and are not part of this layout. */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "trex_oracle.h"

static void put16(uint8_t* p, uint32_t v) { p[0] = (uint8_t)(v & 0xff); p[1] = (uint8_t)(v >> 8); }
static uint32_t get16(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8); }

/* bytes of one serialized frame; out may be NULL (size query) */
uint64_t oracle_pv_serialize_v6(const oracle_blob* blobs, int32_t n_blobs, const oracle_run* runs, const uint8_t* pixels, uint64_t timestamp, uint8_t* out) {
    uint64_t o = 0;
    if (out) { out[0] = 0; for (int k = 0; k < 8; ++k) out[1 + k] = (uint8_t)(timestamp >> (8 * k)); put16(out + 9, (uint32_t)n_blobs); }
    o = 11;
    for (int32_t b = 0; b < n_blobs; ++b) {
        const oracle_blob* B = &blobs[b];
        const oracle_run* rr = runs + B->run_begin;
        if (out) { put16(out + o, B->n_runs ? rr[0].y : 0); put16(out + o + 2, B->n_runs); }
        o += 4;
        for (uint32_t j = 0; j < B->n_runs; ++j) {
            const int eol = (j + 1 == B->n_runs) || rr[j + 1].y != rr[j].y;
            if (out) { put16(out + o, rr[j].x0); put16(out + o + 2, ((uint32_t)rr[j].x1 << 1) | (uint32_t)eol); }
            o += 4;
        }
        if (out) memcpy(out + o, pixels + B->pix_begin, B->n_pixels);
        o += B->n_pixels;
    }
    return o;
}

/* Frame::read_from for version V_6: returns the bytes consumed (0 on a malformed / compressed frame).  runs: uncompressed lines with
 * their y; blob_runs / blob_pixels: per object counts; capacities are checked. */
uint64_t oracle_pv_read_v6(const uint8_t* buf, uint64_t size, uint64_t* timestamp, int32_t* n_out, oracle_run* runs, int32_t max_runs,
                           uint8_t* pixels, int64_t max_pixels, uint32_t* blob_runs, uint32_t* blob_pixels, int32_t max_blobs) {
    if (size < 11 || buf[0] != 0) return 0;
    uint64_t ts = 0;
    for (int k = 0; k < 8; ++k) ts |= (uint64_t)buf[1 + k] << (8 * k);
    *timestamp = ts;
    const int32_t n = (int32_t)get16(buf + 9);
    if (n > max_blobs) return 0;
    uint64_t o = 11; int32_t nr = 0; int64_t np = 0;
    for (int32_t i = 0; i < n; ++i) {
        if (o + 4 > size) return 0;
        uint32_t y = get16(buf + o); const uint32_t m = get16(buf + o + 2);
        o += 4;
        if (o + 4ull * m > size || nr + (int32_t)m > max_runs) return 0;
        uint64_t px = 0;
        for (uint32_t j = 0; j < m; ++j) {
            const uint32_t x0 = get16(buf + o), w = get16(buf + o + 2);
            o += 4;
            runs[nr].x0 = (uint16_t)x0; runs[nr].x1 = (uint16_t)((w & 0xfffeu) >> 1); runs[nr].y = (uint16_t)y; runs[nr].pad = 0;
            px += (uint64_t)(runs[nr].x1 - runs[nr].x0 + 1);
            ++nr;
            if (w & 1u) ++y;                                       /* eol: the following lines are on current_y + 1 */
        }
        if (o + px > size || np + (int64_t)px > max_pixels) return 0;
        memcpy(pixels + np, buf + o, px);
        o += px; np += (int64_t)px;
        blob_runs[i] = m; blob_pixels[i] = (uint32_t)px;
    }
    *n_out = n;
    return o;
}
