// SHA-256 (FIPS 180-4) device code, one message per lane.  Replaces crypto.subtle.digest('SHA-256', ..) of
// src/curves/group.ts:221-233 (hashPoints) and implements the counter-mode RNG block of the RNG contract.
#pragma once
#include "zkdev.h"

ZK_CONSTANT uint32_t SHA_K[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01,
    0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc,
    0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147,
    0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
    0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08,
    0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
    0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};

ZK_DEV uint32_t rotr32(uint32_t x, int n) { return zk_rotr32(x, n); }
ZK_DEV uint32_t bswap32(uint32_t x) { return __builtin_bswap32(x); }

ZK_DEV void sha256_iv(uint32_t h[8]) {
    h[0] = 0x6a09e667, h[1] = 0xbb67ae85, h[2] = 0x3c6ef372, h[3] = 0xa54ff53a;
    h[4] = 0x510e527f, h[5] = 0x9b05688c, h[6] = 0x1f83d9ab, h[7] = 0x5be0cd19;
}
// one compression; w[16] is the big-endian-decoded block (clobbered).  Ch is one v_bfi_b32 and Maj two instructions (bitwise
// (a ^ b) ? c : b), which the compiler does not form from the textbook expressions: 1 749 -> 1 656 vector instructions per compression
// as compiled, on ~6 000 compressions per proof.
ZK_DEV void sha256_compress(uint32_t h[8], uint32_t w[16]) {
    uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
#pragma unroll
    for (int i = 0; i < 64; i++) {
        if (i >= 16) {
            uint32_t w15 = w[(i - 15) & 15], w2 = w[(i - 2) & 15];
            uint32_t s0 = zk_xor3(rotr32(w15, 7), rotr32(w15, 18), w15 >> 3);
            uint32_t s1 = zk_xor3(rotr32(w2, 17), rotr32(w2, 19), w2 >> 10);
            w[i & 15] = w[i & 15] + s0 + w[(i - 7) & 15] + s1;
        }
        uint32_t S1 = zk_xor3(rotr32(e, 6), rotr32(e, 11), rotr32(e, 25));
        uint32_t ch = zk_bfi(e, f, g);
        uint32_t t1 = hh + S1 + ch + SHA_K[i] + w[i & 15];
        uint32_t S0 = zk_xor3(rotr32(a, 2), rotr32(a, 13), rotr32(a, 22));
        uint32_t mj = zk_maj(a, b, c);
        uint32_t t2 = S0 + mj;
        hh = g, g = f, f = e, e = d + t1, d = c, c = b, b = a, a = t1 + t2;
    }
    h[0] += a, h[1] += b, h[2] += c, h[3] += d, h[4] += e, h[5] += f, h[6] += g, h[7] += hh;
}

// One compression reading the block from a lane's word-interleaved LDS column.  Deliberately NOT inlined: the byte
// absorber below reaches it from many call sites and the 64 unrolled rounds must exist once per kernel.
struct ShaState {
    uint32_t v[8];
};
ZK_DEV_NOINLINE ShaState sha256_compress_lds(ShaState st, const uint32_t* buf, uint32_t stride) {
    uint32_t w[16];
#pragma unroll
    for (int j = 0; j < 16; j++) w[j] = buf[j * stride];
    sha256_compress(st.v, w);
    return st;
}

// Byte-stream absorber.  The 64-byte block buffer of each lane lives in LDS, word-interleaved across the
// workgroup's lanes (word j of lane t at buf[j*nthreads + t]) so same-position accesses are conflict-free.
// All lanes of a wave must absorb messages of identical structure (true for every hash on the hot path).
struct ShaStream {
    ShaState h;
    uint32_t fill;    // bytes in the current block
    uint32_t blocks;  // completed blocks
    uint32_t* buf;    // LDS, this lane's column
    uint32_t stride;  // lanes per workgroup
    uint32_t cur;     // word being assembled
    ZK_DEV void init(uint32_t* lds_base, uint32_t lane, uint32_t nlanes) {
        sha256_iv(h.v);
        fill = 0, blocks = 0, buf = lds_base + lane, stride = nlanes, cur = 0;
    }
    ZK_DEV void put_byte(uint32_t b) {
        cur = (cur << 8) | (b & 0xff);
        fill++;
        if ((fill & 3) == 0) {
            buf[((fill >> 2) - 1) * stride] = cur;
            if (fill == 64) {
                h = sha256_compress_lds(h, buf, stride);
                blocks++;
                fill = 0;
            }
        }
    }
    // four bytes at once, most significant first, at any alignment: with r = fill % 4 bytes pending in `cur`, the completed word is
    // the pending bytes followed by the top 4 - r bytes of v, and the low r bytes of v become the pending ones
    ZK_DEV void put_word(uint32_t v) {
        const uint32_t r = fill & 3;
        buf[(fill >> 2) * stride] = r ? (cur << (32 - 8 * r)) | (v >> (8 * r)) : v;
        cur = v;
        fill += 4;
        if (fill >= 64) {
            h = sha256_compress_lds(h, buf, stride);
            blocks++;
            fill -= 64;
        }
    }
    // big-endian encoding of the NBYTES least-significant bytes of a little-endian word array: the odd leading bytes one by one, the
    // rest as whole words (the points of hashPoints are 0x04 || X || Y with 32- or 33-byte coordinates, so the stream is rarely
    // word-aligned; byte-wise absorption cost ~6 instructions per byte, a fifth of the compression)
    template <int NBYTES>
    ZK_DEV void put_be(const uint32_t* w) {
#pragma unroll
        for (int i = NBYTES - 1; i >= (NBYTES / 4) * 4; i--) put_byte(w[i >> 2] >> (8 * (i & 3)));
#pragma unroll
        for (int i = NBYTES / 4 - 1; i >= 0; i--) put_word(w[i]);
    }
    ZK_DEV void finish(uint32_t out[8]) {
        uint64_t bits = ((uint64_t)blocks * 64 + fill) * 8;
        put_byte(0x80);
        while (fill != 56) put_byte(0);
#pragma unroll
        for (int i = 7; i >= 0; i--) put_byte((uint32_t)(bits >> (8 * i)));
#pragma unroll
        for (int i = 0; i < 8; i++) out[i] = h.v[i];
    }
};

// Hardened mode (include/zkattest.h, zk_ctx_set_mode): bytes appended to the Groth-Kohlweiss transcript before the points'
// hash is finished -- tag || ring digest || msgHash; the caller then absorbs R and keyXcom in their hashPoints encodings.
ZK_DEV void sha_put_gk_statement_head(ShaStream& s, const uint32_t* ring_digest8, const uint8_t* msg32) {
    const char tag[] = "ZKAttest-GK-statement-v1";
#pragma unroll 1
    for (int i = 0; i < 24; i++) s.put_byte((uint8_t)tag[i]);
#pragma unroll 1
    for (int i = 0; i < 8; i++) {
        uint32_t w = ring_digest8[i];
        s.put_byte(w >> 24), s.put_byte(w >> 16), s.put_byte(w >> 8), s.put_byte(w);
    }
#pragma unroll 1
    for (int i = 0; i < 32; i++) s.put_byte(msg32[i]);
}

