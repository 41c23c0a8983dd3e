// Snappy raw-format codec (host side) for the KNOSSOS overlay cubes ("*.seg.sz.zip"): knossos_utils -- the volume
// I/O library SyConn calls at /root/reference/syconn/handler/prediction.py:700-702, 835-843 -- stores every 128^3
// uint64 segmentation cube as python-snappy `compress(cube.tobytes())` inside a zip member (SURVEY.md section 8f,
// row 1).  Neither knossos_utils nor snappy (pinned: snappy 1.1.8 / python-snappy 0.6.0,
// /root/reference/examples/working_env_glibc_2_27_2021_11.yml:247,270) is vendored in the reference or installed in
// this image, so this is a restatement of the PUBLISHED format (google/snappy format_description.txt):
//
//   stream   := varint(uncompressed length) element*
//   element  := literal | copy
//   tag & 3 == 0  literal: len-1 in tag>>2 if < 60, else (tag>>2) - 59 little-endian length bytes follow (len-1)
//   tag & 3 == 1  copy, len = 4 + ((tag>>2) & 7), offset = ((tag>>5) << 8) | next byte          (1..2047)
//   tag & 3 == 2  copy, len = 1 + (tag>>2), offset = next 2 bytes little-endian
//   tag & 3 == 3  copy, len = 1 + (tag>>2), offset = next 4 bytes little-endian
//   a copy may overlap its own output (offset < len repeats a pattern); offset 0 is invalid.
//
// Any stream a conforming decoder accepts is valid, so the compressor is free in its match finding; this one is a
// greedy 4-byte-hash matcher over 64 KiB blocks (matches never reach back across a block start, like the upstream
// encoder), which turns the long constant runs of label volumes into 64-byte copies at offset 8.
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../include/syconn_dense.h"

namespace {

constexpr size_t kBlock = 1u << 16;
constexpr int kHashBits = 14;

inline uint32_t load32(const uint8_t* p) { uint32_t v; std::memcpy(&v, p, 4); return v; }
inline uint32_t hash4(uint32_t v) { return (v * 0x1e35a7bdu) >> (32 - kHashBits); }

uint8_t* emit_literal(uint8_t* op, const uint8_t* lit, size_t len) {
    const size_t n = len - 1;
    if (n < 60) {
        *op++ = (uint8_t)(n << 2);
    } else {
        uint8_t* tagp = op++;
        int count = 0;
        size_t v = n;
        while (v > 0) { *op++ = (uint8_t)(v & 0xff); v >>= 8; ++count; }
        *tagp = (uint8_t)((59 + count) << 2);
    }
    std::memcpy(op, lit, len);
    return op + len;
}

uint8_t* emit_copy_upto64(uint8_t* op, size_t offset, size_t len) {   // 4 <= len <= 64, offset < 65536
    if (len < 12 && offset < 2048) {
        *op++ = (uint8_t)(1 | ((len - 4) << 2) | ((offset >> 8) << 5));
        *op++ = (uint8_t)(offset & 0xff);
    } else {
        *op++ = (uint8_t)(2 | ((len - 1) << 2));
        *op++ = (uint8_t)(offset & 0xff);
        *op++ = (uint8_t)(offset >> 8);
    }
    return op;
}

uint8_t* emit_copy(uint8_t* op, size_t offset, size_t len) {
    while (len >= 68) { op = emit_copy_upto64(op, offset, 64); len -= 64; }
    if (len > 64) { op = emit_copy_upto64(op, offset, 60); len -= 60; }    // leaves 5..8: still a valid copy
    return emit_copy_upto64(op, offset, len);
}

uint8_t* compress_block(const uint8_t* base, size_t n, uint8_t* op, uint16_t* table) {
    std::memset(table, 0, sizeof(uint16_t) << kHashBits);
    const uint8_t* ip = base;
    const uint8_t* const end = base + n;
    const uint8_t* lit = base;
    if (n >= 8) {
        const uint8_t* const limit = end - 4;          // last position a 4-byte load may start at
        ++ip;                                           // position 0 can never be a match target of itself
        while (ip <= limit) {
            const uint32_t cur = load32(ip);
            const uint32_t h = hash4(cur);
            const uint8_t* cand = base + table[h];
            table[h] = (uint16_t)(ip - base);
            if (cand < ip && load32(cand) == cur) {
                if (ip > lit) op = emit_literal(op, lit, (size_t)(ip - lit));
                size_t len = 4;
                while (ip + len < end && cand[len] == ip[len]) ++len;
                op = emit_copy(op, (size_t)(ip - cand), len);
                ip += len;
                lit = ip;
                if (ip <= limit && ip - 1 > base) table[hash4(load32(ip - 1))] = (uint16_t)(ip - 1 - base);
            } else {
                ++ip;
            }
        }
    }
    if (end > lit) op = emit_literal(op, lit, (size_t)(end - lit));
    return op;
}

}  // namespace

extern "C" {

size_t sd_snappy_max_compressed_length(size_t n) { return 32 + n + n / 6; }

int sd_snappy_compress(const void* src, size_t n, void* dst, size_t dst_capacity, size_t* dst_len) {
    if ((!src && n) || !dst || !dst_len) return SD_ERR_INVALID;
    if (n > 0xffffffffull || dst_capacity < sd_snappy_max_compressed_length(n)) return SD_ERR_INVALID;
    uint8_t* op = static_cast<uint8_t*>(dst);
    size_t v = n;
    while (v >= 0x80) { *op++ = (uint8_t)(v | 0x80); v >>= 7; }
    *op++ = (uint8_t)v;
    std::vector<uint16_t> table((size_t)1 << kHashBits);
    const uint8_t* ip = static_cast<const uint8_t*>(src);
    for (size_t done = 0; done < n; done += kBlock)
        op = compress_block(ip + done, n - done < kBlock ? n - done : kBlock, op, table.data());
    *dst_len = (size_t)(op - static_cast<uint8_t*>(dst));
    return SD_OK;
}

int sd_snappy_uncompressed_length(const void* src, size_t n, size_t* result) {
    if (!src || !result) return SD_ERR_INVALID;
    const uint8_t* p = static_cast<const uint8_t*>(src);
    uint64_t v = 0;
    for (int shift = 0, i = 0; i < 5; ++i, shift += 7) {
        if ((size_t)i >= n) return SD_ERR_INVALID;
        const uint8_t b = p[i];
        v |= (uint64_t)(b & 0x7f) << shift;
        if (!(b & 0x80)) {
            if (v > 0xffffffffull) return SD_ERR_INVALID;
            *result = (size_t)v;
            return SD_OK;
        }
    }
    return SD_ERR_INVALID;
}

int sd_snappy_uncompress(const void* src, size_t n, void* dst, size_t dst_capacity, size_t* dst_len) {
    size_t ulen = 0;
    if (!dst_len || sd_snappy_uncompressed_length(src, n, &ulen) != SD_OK) return SD_ERR_INVALID;
    if (ulen > dst_capacity || (!dst && ulen)) return SD_ERR_INVALID;
    const uint8_t* ip = static_cast<const uint8_t*>(src);
    const uint8_t* const iend = ip + n;
    while (*ip++ & 0x80) {}
    uint8_t* const out = static_cast<uint8_t*>(dst);
    size_t op = 0;
    while (ip < iend) {
        const uint8_t tag = *ip++;
        size_t len, offset;
        switch (tag & 3) {
        case 0: {
            len = (size_t)(tag >> 2) + 1;
            if (len > 60) {
                const int nb = (int)len - 60;
                if (iend - ip < nb) return SD_ERR_INVALID;
                size_t v = 0;
                for (int i = 0; i < nb; ++i) v |= (size_t)ip[i] << (8 * i);
                ip += nb;
                len = v + 1;
            }
            if ((size_t)(iend - ip) < len || ulen - op < len) return SD_ERR_INVALID;
            std::memcpy(out + op, ip, len);
            ip += len;
            op += len;
            continue;
        }
        case 1:
            if (iend - ip < 1) return SD_ERR_INVALID;
            len = 4 + ((tag >> 2) & 7);
            offset = ((size_t)(tag >> 5) << 8) | *ip++;
            break;
        case 2:
            if (iend - ip < 2) return SD_ERR_INVALID;
            len = (size_t)(tag >> 2) + 1;
            offset = (size_t)ip[0] | ((size_t)ip[1] << 8);
            ip += 2;
            break;
        default:
            if (iend - ip < 4) return SD_ERR_INVALID;
            len = (size_t)(tag >> 2) + 1;
            offset = (size_t)ip[0] | ((size_t)ip[1] << 8) | ((size_t)ip[2] << 16) | ((size_t)ip[3] << 24);
            ip += 4;
            break;
        }
        if (offset == 0 || offset > op || ulen - op < len) return SD_ERR_INVALID;
        const uint8_t* from = out + op - offset;
        if (offset >= len) std::memcpy(out + op, from, len);
        else for (size_t i = 0; i < len; ++i) out[op + i] = from[i];      // overlapping copy: byte by byte
        op += len;
    }
    if (op != ulen) return SD_ERR_INVALID;
    *dst_len = ulen;
    return SD_OK;
}

}  // extern "C"
