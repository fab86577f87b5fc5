// pvfile.cpp -- host side of the .pv data section: per-frame LZO1X compression and the index table.
//
// Replaces, for the frames trexhip_pack_frames_v6_device produced:
//   pv::Frame::serialize's tail     Application/src/ProcessedVideo/pv.cpp:705-772  (frames of >= 15000 bytes -- all frames of the rgb8
//                                   encoding -- go through lzo1x_1_compress and are kept compressed when that is smaller)
//   pv::File::add_individual        pv.cpp:1488-1496  (u8 compression_flag, then the pack; the frame's file offset goes to the index table)
//   pv::Header::update              pv.cpp:1181-1192  (the index table: one u64 file offset per frame)
// and is read back by pv::Frame::read_from (pv.cpp:313-340: u8 flag, u32 compressed size, u32 uncompressed size, lzo1x_decompress).
//
// The compressor below is this library's own: a greedy hash-chain-free LZ77 matcher writing the LZO1X bit stream (literal runs, M2 / M3 / M4
// matches, end marker) that lzo1x_decompress accepts.  It does not reproduce minilzo's lzo1x_1_compress byte for byte and does not have to:
// what a reader sees is the decompressed frame.  tests/test_pv_file.py feeds its output to the reference's own lzo1x_decompress
// (ProcessedVideo/lzo/minilzo.c compiled as it lies in the reference tree, by a recipe of the test infrastructure) and gets the input back.
#include <stdint.h>
#include <string.h>
#include <string>
#include <vector>
#include "../../include/trexhip.h"

namespace trexhip { void set_error(const std::string& msg); }

namespace {

inline uint32_t load32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }

// a length beyond what the instruction byte holds: zero bytes count 255 each, the last byte the rest (1..255)
inline uint8_t* put_long(uint8_t* op, size_t rest) {
    while (rest > 255) { *op++ = 0; rest -= 255; }
    *op++ = (uint8_t)rest;
    return op;
}

// t literal bytes.  At the start of the stream a short run has its own first byte (17 + t); up to 3 literals behind a match ride in the
// two low bits of that match's second-to-last byte; longer runs are an instruction 0..15 (legal because the match before them then
// carries no literals, i.e. the decoder is in its "no trailing literals" state)
inline uint8_t* put_literals(uint8_t* op, const uint8_t* lit, size_t t, bool first) {
    if (t == 0) return op;
    if (first && t <= 238) *op++ = (uint8_t)(17 + t);
    else if (!first && t <= 3) op[-2] = (uint8_t)(op[-2] | t);
    else if (t <= 18) *op++ = (uint8_t)(t - 3);
    else { *op++ = 0; op = put_long(op, t - 18); }
    memcpy(op, lit, t);
    return op + t;
}

inline uint8_t* put_match(uint8_t* op, size_t len, size_t off) {
    if (len <= 8 && off <= 2048) {                    // M2: 3..8 bytes, distance <= 2048
        off -= 1;
        *op++ = (uint8_t)(((len - 1) << 5) | ((off & 7) << 2));
        *op++ = (uint8_t)(off >> 3);
    } else if (off <= 16384) {                        // M3: distance <= 16384
        off -= 1;
        if (len <= 33) *op++ = (uint8_t)(32 | (len - 2));
        else { *op++ = 32; op = put_long(op, len - 33); }
        *op++ = (uint8_t)((off << 2) & 0xff);
        *op++ = (uint8_t)(off >> 6);
    } else {                                          // M4: distance 16385 .. 49151
        off -= 16384;
        const uint8_t hb = (uint8_t)((off >> 11) & 8);
        if (len <= 9) *op++ = (uint8_t)(16 | hb | (len - 2));
        else { *op++ = (uint8_t)(16 | hb); op = put_long(op, len - 9); }
        *op++ = (uint8_t)(((off & 0x3fff) << 2) & 0xff);
        *op++ = (uint8_t)((off & 0x3fff) >> 6);
    }
    return op;
}

constexpr int HASH_BITS = 15;
constexpr size_t MAX_OFF = 49151;

size_t lzo1x_compress(const uint8_t* in, size_t n, uint8_t* out) {
    uint8_t* op = out;
    const uint8_t* const end = in + n;
    const uint8_t* ip = in;
    const uint8_t* lit = in;                           // start of the pending literal run
    bool first = true;
    if (n >= 8) {
        std::vector<uint32_t> table((size_t)1 << HASH_BITS, 0u);     // position + 1 of the last 4-byte group with this hash
        const uint8_t* const limit = end - 4;
        while (ip <= limit) {
            const uint32_t v = load32(ip);
            const uint32_t h = (v * 2654435761u) >> (32 - HASH_BITS);
            const uint32_t cand = table[h];
            table[h] = (uint32_t)(ip - in) + 1;
            if (cand) {
                const uint8_t* mp = in + (cand - 1);
                const size_t off = (size_t)(ip - mp);
                if (off <= MAX_OFF && load32(mp) == v) {
                    size_t len = 4;
                    while (ip + len < end && mp[len] == ip[len]) ++len;
                    op = put_literals(op, lit, (size_t)(ip - lit), first);
                    first = false;
                    op = put_match(op, len, off);
                    // the positions inside the match feed the table too (every second one: the ratio hardly moves, the time does)
                    const uint8_t* q = ip + 1;
                    ip += len;
                    for (; q + 4 <= end && q < ip; q += 2) table[(load32(q) * 2654435761u) >> (32 - HASH_BITS)] = (uint32_t)(q - in) + 1;
                    lit = ip;
                    continue;
                }
            }
            ++ip;
        }
    }
    op = put_literals(op, lit, (size_t)(end - lit), first);
    *op++ = 17; *op++ = 0; *op++ = 0;                  // end of stream: an M4 instruction with distance 16384
    return (size_t)(op - out);
}

}  // namespace

extern "C" {

size_t trexhip_lzo1x_bound(size_t n) { return n + n / 16 + 64 + 3; }       // pv.cpp:712 OUT_LEN

int trexhip_lzo1x_compress(const uint8_t* in, size_t n, uint8_t* out, size_t capacity, size_t* out_len) {
    if ((!in && n) || !out || !out_len) { trexhip::set_error("trexhip_lzo1x_compress: null argument"); return TREXHIP_E_INVALID; }
    if (capacity < trexhip_lzo1x_bound(n)) { trexhip::set_error("trexhip_lzo1x_compress: the output buffer must hold trexhip_lzo1x_bound(n) bytes"); return TREXHIP_E_INVALID; }
    if (n >= 0xffffffffull) { trexhip::set_error("trexhip_lzo1x_compress: a frame is below 4 GB (pv.cpp:726)"); return TREXHIP_E_INVALID; }
    *out_len = lzo1x_compress(in, n, out);
    return TREXHIP_OK;
}

int trexhip_pv_write_frames(const uint8_t* bodies, const uint64_t* offsets, int32_t n_frames, int32_t always_compress, uint64_t file_offset,
                            uint8_t* out, size_t capacity, uint64_t* index_table, size_t* out_bytes) {
    if (!bodies || !offsets || !out || !index_table || !out_bytes || n_frames < 0) { trexhip::set_error("trexhip_pv_write_frames: bad argument"); return TREXHIP_E_INVALID; }
    size_t o = 0;
    std::vector<uint8_t> tmp;
    for (int32_t f = 0; f < n_frames; ++f) {
        const uint8_t* b = bodies + offsets[f];
        const size_t total = (size_t)(offsets[f + 1] - offsets[f]);
        if (total < 1 || b[0] != 0) { trexhip::set_error("trexhip_pv_write_frames: a frame does not start with compression_flag 0"); return TREXHIP_E_INVALID; }
        const size_t in_len = total - 1;                                 // the pack of Frame::serialize: everything behind the flag
        index_table[f] = file_offset + o;
        bool done = false;
        if (always_compress || in_len >= 15000) {                        // pv.cpp:707-708
            tmp.resize(trexhip_lzo1x_bound(in_len));
            const size_t out_len = lzo1x_compress(b + 1, in_len, tmp.data());
            if (out_len + 8 < in_len) {                                   // pv.cpp:758: kept only when smaller, the two sizes included
                if (o + 9 + out_len > capacity) { trexhip::set_error("trexhip_pv_write_frames: output buffer too small"); return TREXHIP_E_INVALID; }
                out[o] = 1;
                const uint32_t a = (uint32_t)out_len, u = (uint32_t)in_len;
                memcpy(out + o + 1, &a, 4); memcpy(out + o + 5, &u, 4);
                memcpy(out + o + 9, tmp.data(), out_len);
                o += 9 + out_len;
                done = true;
            }
        }
        if (!done) {
            if (o + total > capacity) { trexhip::set_error("trexhip_pv_write_frames: output buffer too small"); return TREXHIP_E_INVALID; }
            memcpy(out + o, b, total);
            o += total;
        }
    }
    *out_bytes = o;
    return TREXHIP_OK;
}

}  // extern "C"
