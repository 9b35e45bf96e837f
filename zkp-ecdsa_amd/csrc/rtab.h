// Per-proof table of R (the ECDSA nonce point) and the scalar multiplication k*R through it.
// alpha_i * R is needed for every repetition of proveExp (exp.ts:144-149) and for the checked repetitions of
// verifyExp (exp.ts:267,299), always with the same R per proof, so each proof gets a comb table
//   entry[w][d] = d * 2^(bits w) * R,  d = 0 .. 2^(bits-1)   (d = 0: the identity, absorbed by the complete addition)
// and k is recoded into SIGNED digits in [-2^(bits-1)+1, 2^(bits-1)]: half the entries of an unsigned comb of the same
// width; a negative digit negates Y on load.  Prover: bits = 6 (43 windows x 32 entries, 43 additions per multiple,
// 80 multiples per proof); verifier: bits = 4 (65 x 8, 21 multiples per proof).  Projective entries (X, Y, Z), 28 words.
#pragma once
#include "engine.h"

ZK_DEV P256Pt ld_rtab(const uint32_t* e) {
    const uint4* q = (const uint4*)e;
    uint32_t w[28];
#pragma unroll
    for (int i = 0; i < 7; i++) {
        uint4 v = q[i];
        w[4 * i] = v.x, w[4 * i + 1] = v.y, w[4 * i + 2] = v.z, w[4 * i + 3] = v.w;
    }
    P256Pt a;
#pragma unroll
    for (int l = 0; l < 9; l++) a.x.l[l] = w[l], a.y.l[l] = w[9 + l], a.z.l[l] = w[18 + l];
    return a;
}
ZK_DEV void st_rtab(uint32_t* e, const P256Pt& a) {
#pragma unroll
    for (int l = 0; l < 9; l++) e[l] = a.x.l[l], e[9 + l] = a.y.l[l], e[18 + l] = a.z.l[l];
    e[27] = 0;
}
// -Y for a coordinate < 8q: 8q - y, again <= 8q (limbs of sub8 are >= 2^30 - 1, so no borrow)
ZK_DEV Fq8 fq8_neg(const Fq8& y) {
    Fq8 r;
#pragma unroll
    for (int i = 0; i < NLIMB; i++) r.l[i] = ModQ::sub8[i] - y.l[i];
    limbs_normalize(r.l);
    return r;
}
ZK_DEV void shr256_var(uint32_t* w, uint32_t sh) {  // sh < 32
#pragma unroll
    for (int i = 0; i < 7; i++) w[i] = __funnelshift_r(w[i], w[i + 1], sh);
    w[7] >>= sh;
}
// k * R, k < 2^256 as 8 little-endian words (destroyed)
ZK_DEV P256Pt p256_rtab_mul(const uint32_t* __restrict__ rtab, uint32_t kw[8], uint32_t bits) {
    const uint32_t nwin = rtab_nwin(bits), ent = rtab_entries(bits), half = 1u << (bits - 1), mask = (1u << bits) - 1;
    P256Pt acc;
    uint32_t carry = 0;
#pragma unroll 1
    for (uint32_t w = 0; w < nwin; w++) {
        uint32_t d = (kw[0] & mask) + carry;
        shr256_var(kw, bits);
        bool neg = d > half;
        carry = neg ? 1 : 0;
        if (neg) d = (1u << bits) - d;
        P256Pt e = ld_rtab(rtab + (size_t)RTAB_ENTRY_WORDS * (w * ent + d));
        Fq8 ny = fq8_neg(e.y);
        e.y = fe_select(neg, ny, e.y);
        if (w == 0) acc = e;   // identity + entry (entry 0 is the identity itself)
        else acc = p256_add(acc, e);
    }
    return acc;
}
// The same sum restricted to `per` windows from w0 on, for several lanes that take a range each and add their partial sums -- the table holds every
// 2^(bits w) R, so no lane needs a doubling.  Two loops on purpose: the recoding of the windows below w0 (integer carries only) has a different trip
// count per lane and costs next to nothing; the loop over the lane's own windows has the SAME trip count in every lane, so the wave executes `per`
// additions, not nwin (a single loop from 0 with the additions masked per lane runs all nwin of them: measured, profiles/r04_ab_variants.txt (10)).
ZK_DEV P256Pt p256_rtab_mul_range(const uint32_t* __restrict__ rtab, uint32_t kw[8], uint32_t bits, uint32_t w0, uint32_t per) {
    const uint32_t nwin = rtab_nwin(bits), ent = rtab_entries(bits), half = 1u << (bits - 1), mask = (1u << bits) - 1;
    uint32_t carry = 0;
#pragma unroll 1
    for (uint32_t w = 0; w < w0; w++) {
        uint32_t d = (kw[0] & mask) + carry;
        shr256_var(kw, bits);
        carry = d > half ? 1 : 0;
    }
    P256Pt acc = p256_identity();
#pragma unroll 1
    for (uint32_t j = 0; j < per; j++) {
        const uint32_t w = w0 + j;
        uint32_t d = (kw[0] & mask) + carry;
        shr256_var(kw, bits);
        bool neg = d > half;
        carry = neg ? 1 : 0;
        if (neg) d = (1u << bits) - d;
        const bool in = w < nwin;
        P256Pt e = ld_rtab(rtab + (size_t)RTAB_ENTRY_WORDS * ((in ? w : 0) * ent + (in ? d : 0)));   // past the last window: entry 0 of window 0, the identity
        Fq8 ny = fq8_neg(e.y);
        e.y = fe_select(neg && in, ny, e.y);
        acc = p256_add(acc, e);
    }
    return acc;
}
