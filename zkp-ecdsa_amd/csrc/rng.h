// Deterministic RNG contract on the device (replaces crypto.getRandomValues inside rnd(), src/bignum/big.ts:171-181):
// the k-th 32-byte fill of proof p is SHA-256(seed_p || be64(k)) (mode 0) or block k of an explicit stream
// (mode 1).  rnd(order) consumes fills until one is < order; both orders on the path (n, q) have top word
// 0xffffffff, so a fill can only be rejected if its first big-endian word is 0xffffffff (p = 2^-32).  A prepass
// (k_rng_prepass) lists those fills per proof; rng_map() then turns "logical draw k" into the fill index the
// reference would have used, so every draw stays random-access for the data-parallel phases.
#pragma once
#include "sha256.h"
#include "field.h"

#define RNG_MAX_EXC 8
struct RngCtx {
    const uint8_t* seeds;    // mode 0: B x 32
    const uint8_t* stream;   // mode 1: B x stride x 32
    uint64_t stride_blocks;
    int mode;
    int sec;                 // secLevel (draw layout)
    uint32_t* exc_idx;       // [C][RNG_MAX_EXC] fill indices whose first word is 0xffffffff
    uint32_t* exc_flags;     // bit0: value >= n, bit1: value >= q
    uint32_t* exc_cnt;       // [C]
    uint32_t proof_base;     // index of the chunk's first proof inside seeds/stream
};

// fill `blk` of proof -> 8 little-endian words (w[0] least significant)
ZK_DEV void rng_block(const RngCtx& g, uint32_t proof, uint32_t blk, uint32_t w[8]) {
    const uint64_t gp = (uint64_t)g.proof_base + proof;
    if (g.mode == 0) {
        const uint32_t* s = (const uint32_t*)(g.seeds + 32 * gp);
        uint32_t m[16], h[8];
#pragma unroll
        for (int i = 0; i < 8; i++) m[i] = bswap32(s[i]);
        m[8] = 0, m[9] = blk, m[10] = 0x80000000u, m[11] = 0, m[12] = 0, m[13] = 0, m[14] = 0, m[15] = 320;
        sha256_iv(h);
        sha256_compress(h, m);
#pragma unroll
        for (int i = 0; i < 8; i++) w[i] = h[7 - i];
    } else {
        bool ok = blk < g.stride_blocks;
        const uint32_t* s = (const uint32_t*)(g.stream + 32 * (gp * g.stride_blocks + (ok ? blk : 0)));
#pragma unroll
        for (int i = 0; i < 8; i++) w[i] = ok ? bswap32(s[7 - i]) : 0;
    }
}
// modulus of logical draw j inside proveSignatureList: true = n, false = q
ZK_DEV bool rng_draw_is_n(const RngCtx& g, uint32_t j) { return j == 0 || (j >= 3 && j < 3 + 4 * (uint32_t)g.sec && ((j - 3) & 3) < 2); }

ZK_DEV uint32_t rng_map(const RngCtx& g, uint32_t proof, uint32_t k) {
    uint32_t cnt = g.exc_cnt[proof];
    if (cnt == 0) return k;
    if (cnt > RNG_MAX_EXC) cnt = RNG_MAX_EXC;
    uint32_t idx[RNG_MAX_EXC], fl[RNG_MAX_EXC];
    for (uint32_t i = 0; i < cnt; i++) {  // insertion sort (the prepass appends in arbitrary order)
        uint32_t e = g.exc_idx[proof * RNG_MAX_EXC + i], f = g.exc_flags[proof * RNG_MAX_EXC + i];
        uint32_t j = i;
        while (j > 0 && idx[j - 1] > e) idx[j] = idx[j - 1], fl[j] = fl[j - 1], j--;
        idx[j] = e, fl[j] = f;
    }
    uint32_t shift = 0;
    for (uint32_t i = 0; i < cnt; i++) {
        if (idx[i] < shift) continue;
        uint32_t j = idx[i] - shift;
        if (j > k) break;
        bool rejected = rng_draw_is_n(g, j) ? (fl[i] & 1) : ((fl[i] >> 1) & 1);
        if (rejected) shift++;
    }
    return k + shift;
}
ZK_DEV void rng_draw_words(const RngCtx& g, uint32_t proof, uint32_t k, uint32_t w[8]) { rng_block(g, proof, rng_map(g, proof, k), w); }
// logical draw k as a canonical field element (plain, not Montgomery)
template <class M>
ZK_DEV Fe<M, 1> rng_draw(const RngCtx& g, uint32_t proof, uint32_t k) {
    uint32_t w[8];
    rng_draw_words(g, proof, k, w);
    Fe<M, 1> r;
    limbs_from_words<8>(r.l, w);
    return r;
}
