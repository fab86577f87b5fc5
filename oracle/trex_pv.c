/* trex_pv.c -- CPU restatement (TEST INFRASTRUCTURE ONLY) of the frame body of a .pv file of version V_6:
 *   pv::Frame::serialize   Application/src/ProcessedVideo/pv.cpp:666-703  (timestamp, n, per object start_y / mask_size / lines / pixels)
 *   pv::Frame::read_from   pv.cpp:296-420 with header.version == V_6: compression_flag (:313-316), u64 timestamp (:347-351, V_4),
 *                          u16 n, per object u16 start_y, u16 mask_size, LegacyShortHorizontalLine[mask_size] (:377-388), pixels (:399-403)
 *   LegacyShortHorizontalLine   pv.h:17-52: u16 x0, u16 (x1 << 1) | eol; eol = last line of the current y, the next lines are on y + 1
 * (compress / uncompress themselves are declared in pv.h:27-30 and implemented outside the tree: restated from that description.)
 * Compressed frames (pv.cpp:313-340: u8 flag = 1, u32 compressed size, u32 uncompressed size, LZO1X stream) are read through
 * oracle_lzo1x_decompress below -- a restatement of the LZO1X stream format (the decoder lives in ProcessedVideo/lzo/minilzo.c; when the
 * reference tree is present the restatement is checked against that file compiled as it is: oracle/ref.mk, tests/test_pv_file.py).
 * Source index / flags / predictions came with V_8 / V_9 and are not part of this layout. */
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

/* LZO1X stream -> bytes; returns the decompressed length or -1 (malformed stream / capacity).  Instructions (first byte):
 *   0..15    no literals pending: literal run of 3 + L (L = 0: 18 + 255 per zero byte + last byte); after a literal run: 3-byte match at
 *            distance (H << 2) + D + 2049; with 1..3 literals pending: 2-byte match at distance (H << 2) + D + 1   [0000DDSS, H next byte]
 *   16..31   M4: length 2 + L (L = 0: 9 + ...), distance 16384 + (H << 14) + D; distance 16384 = end of stream            [0001HLLL, 2 bytes D/S]
 *   32..63   M3: length 2 + L (L = 0: 33 + ...), distance D + 1                                                         [001LLLLL, 2 bytes D/S]
 *   64..255  M2: length (byte >> 5) + 1 = 3..8, distance (H << 3) + D + 1                                               [LLLDDDSS, H next byte]
 * S = 0..3 literals follow every match.  A first byte above 17 is a literal run of (byte - 17). */
int64_t oracle_lzo1x_decompress(const uint8_t* in, uint64_t in_len, uint8_t* out, uint64_t cap) {
    const uint8_t* ip = in; const uint8_t* const ie = in + in_len;
    uint8_t* op = out; uint8_t* const oe = out + cap;
    int state = 0;
#define NEED_IN(k) do { if ((uint64_t)(ie - ip) < (uint64_t)(k)) return -1; } while (0)
#define NEED_OUT(k) do { if ((uint64_t)(oe - op) < (uint64_t)(k)) return -1; } while (0)
    NEED_IN(1);
    if (*ip > 17) {
        const uint64_t t = (uint64_t)(*ip++) - 17;
        NEED_IN(t); NEED_OUT(t);
        memcpy(op, ip, t); op += t; ip += t;
        state = t < 4 ? (int)t : 4;
    }
    for (;;) {
        NEED_IN(1);
        const uint32_t inst = *ip++;
        uint64_t len, dist; uint32_t S;
        if (inst < 16) {
            if (state == 0) {
                uint64_t t = inst;
                if (t == 0) { for (;;) { NEED_IN(1); if (*ip) break; t += 255; ++ip; } t += 15 + *ip++; }
                t += 3;
                NEED_IN(t); NEED_OUT(t);
                memcpy(op, ip, t); op += t; ip += t;
                state = 4;
                continue;
            }
            NEED_IN(1);
            if (state == 4) { dist = (inst >> 2) + ((uint64_t)(*ip++) << 2) + 2049; len = 3; }
            else { dist = (inst >> 2) + ((uint64_t)(*ip++) << 2) + 1; len = 2; }
            S = inst & 3;
        } else if (inst >= 64) {
            NEED_IN(1);
            len = (inst >> 5) + 1;
            dist = ((inst >> 2) & 7) + ((uint64_t)(*ip++) << 3) + 1;
            S = inst & 3;
        } else {
            const uint32_t base = inst >= 32 ? 31 : 7;
            len = inst & base;
            if (len == 0) { for (;;) { NEED_IN(1); if (*ip) break; len += 255; ++ip; } len += base + *ip++; }
            len += 2;
            NEED_IN(2);
            const uint32_t d = (uint32_t)ip[0] | ((uint32_t)ip[1] << 8);
            ip += 2;
            S = d & 3;
            if (inst >= 32) dist = (d >> 2) + 1;
            else {
                dist = 16384 + ((uint64_t)(inst & 8) << 11) + (d >> 2);
                if (dist == 16384) return ip == ie ? (int64_t)(op - out) : -1;      /* end of stream */
            }
        }
        if (dist > (uint64_t)(op - out)) return -1;
        NEED_OUT(len);
        { const uint8_t* mp = op - dist; for (uint64_t k = 0; k < len; ++k) op[k] = mp[k]; op += len; }   /* byte by byte: overlapping copies repeat */
        NEED_IN(S); NEED_OUT(S);
        for (uint32_t k = 0; k < S; ++k) *op++ = *ip++;
        state = (int)S;
    }
#undef NEED_IN
#undef NEED_OUT
}

/* Frame::read_from for version V_6: returns the bytes consumed (0 on a malformed frame).  A compressed frame is decompressed first.  runs: uncompressed lines with
 * their y; blob_runs / blob_pixels: per object counts; capacities are checked. */
uint64_t oracle_pv_read_v6(const uint8_t* buf, uint64_t size, uint64_t* timestamp, int32_t* n_out, oracle_run* runs, int32_t max_runs,
                           uint8_t* pixels, int64_t max_pixels, uint32_t* blob_runs, uint32_t* blob_pixels, int32_t max_blobs) {
    if (size >= 9 && buf[0] == 1) {
        const uint64_t csize = (uint64_t)buf[1] | ((uint64_t)buf[2] << 8) | ((uint64_t)buf[3] << 16) | ((uint64_t)buf[4] << 24);
        const uint64_t usize = (uint64_t)buf[5] | ((uint64_t)buf[6] << 8) | ((uint64_t)buf[7] << 16) | ((uint64_t)buf[8] << 24);
        if (9 + csize > size) return 0;
        uint8_t* tmp = (uint8_t*)malloc(usize + 1);
        if (!tmp) return 0;
        tmp[0] = 0;
        uint64_t used = 0;
        if (oracle_lzo1x_decompress(buf + 9, csize, tmp + 1, usize) == (int64_t)usize)
            used = oracle_pv_read_v6(tmp, usize + 1, timestamp, n_out, runs, max_runs, pixels, max_pixels, blob_runs, blob_pixels, max_blobs);
        free(tmp);
        return used == usize + 1 ? 9 + csize : 0;
    }
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
