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

// ---- sums over tables that hold EVERY window multiple need no doubling, so several lanes can take a range of windows each and add their partial sums through
// the wave's cross-lane moves (round 4: k_v_exp_points; round 5: the prover's table sums of a small chunk, k_p256.hip / k_tom.hip "wide" kernels).
#if !defined(ZK_HOST_BUILD)
ZK_DEV P256Pt p256_shfl_xor(const P256Pt& a, int m) {
    P256Pt r;
#pragma unroll
    for (int l = 0; l < NLIMB; l++) {
        r.x.l[l] = (uint32_t)__shfl_xor((int)a.x.l[l], m), r.y.l[l] = (uint32_t)__shfl_xor((int)a.y.l[l], m), r.z.l[l] = (uint32_t)__shfl_xor((int)a.z.l[l], m);
    }
    return r;
}
ZK_DEV P256Aff ld_pfix(const uint32_t* e) {
    const uint4* q = (const uint4*)e;
    uint32_t w[20];
#pragma unroll
    for (int i = 0; i < 5; i++) {
        uint4 v = q[i];
        w[4 * i] = v.x, w[4 * i + 1] = v.y, w[4 * i + 2] = v.z, w[4 * i + 3] = v.w;
    }
    P256Aff a;
#pragma unroll
    for (int l = 0; l < 9; l++) a.x.l[l] = w[l], a.y.l[l] = w[9 + l];
    return a;
}
// acc + (the comb's windows [w0, w0 + per) of k) * B: one lane's share when four lanes split a fixed-base multiplication (k_p256.hip: k_exp_commit_kt_wide, k_front_wide; k_verify.hip: k_v_p256_total_wide)
ZK_DEV P256Pt p256_fixed_mul_range(P256Pt acc, const uint32_t* __restrict__ tab, uint32_t kw[8], uint32_t w0, uint32_t per) {
#pragma unroll 1
    for (uint32_t w = 0; w < w0; w++) shr256<PFIX_WIN_BITS>(kw);
#pragma unroll 1
    for (uint32_t j = 0; j < per; j++) {
        const uint32_t w = w0 + j;
        uint32_t d = kw[0] & (PFIX_WIN_SIZE - 1);
        shr256<PFIX_WIN_BITS>(kw);
        ZK_ADD_IF(w < PFIX_NWIN && d != 0, acc, p256_add_mixed(acc, ld_pfix(tab + (size_t)PFIX_ENTRY_WORDS * ((w < PFIX_NWIN ? w : 0) * PFIX_WIN_SIZE + (d ? d : 1)))));
    }
    return acc;
}
// the sum over the lanes of an aligned group of four (every lane of the group must take part)
ZK_DEV P256Pt p256_quad_sum(P256Pt a) {
    a = p256_add(a, p256_shfl_xor(a, 1));
    return p256_add(a, p256_shfl_xor(a, 2));
}
#endif
