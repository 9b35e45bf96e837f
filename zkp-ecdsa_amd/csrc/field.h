// Device big-number field arithmetic for gfx950 (CDNA4), one field element per lane.
//
// Representation: 9 limbs of 30 bits in 32-bit VGPRs ("radix 2^30"), Montgomery domain with R = 2^270, for all
// three moduli on the hot path (Tom-256 field t [258 bit], P-256 field q = Tom scalar field, P-256 order n).
// Why 30-bit limbs: measured on MI355X (profiles/r01_valu_peak_microbench.txt) v_mad_u64_u32 issues every 4.3 cycles per
// wave at full ILP (15.5 cycles dependent latency), the same slot a 64-bit shift or add costs, and a carry out of the
// 64-bit accumulator would cost a second instruction per product; with 30-bit limbs every column of the product
// (<= 18 partial products < 2^60) fits ONE 64-bit accumulator, so a Montgomery multiplication is at most 162
// v_mad_u64_u32 + 9 v_mul_lo_u32 with no carry flags at all, and additions are plain limb-wise v_add_u32.
//
// Magnitudes are tracked in the TYPE: Fe<M, K> holds a normalised value (every limb < 2^30) that is < K*M.
//   mul : Fe<M,Ka> x Fe<M,Kb> -> Fe<M,2>   (static_assert Ka*Kb <= R/M, so the Montgomery output is < 2M)
//   add : -> Fe<M,Ka+Kb>;  sub(a,b) = a + C*M - b -> Fe<M,Ka+C>, C the smallest power of two > Kb.
// so every lazy add/sub chain is bounds-checked at compile time.  These replace every BigInt `%` of the
// reference (src/bignum/big.ts:36-42 posMod; the `% p` lines of src/curves/weier.ts and edwards.ts).
#pragma once
#include "zkdev.h"
#include "consts_gen.h"

#define LIMB_BITS 30
#define LIMB_MASK 0x3fffffffu
#define NLIMB 9
#define KCAP 512  // column-accumulator bound: top limb of a K*M value stays < 2^28

ZK_DEV uint64_t mad64(uint32_t a, uint32_t b, uint64_t c) { return (uint64_t)a * b + c; }

template <class M, int K = 2>
struct Fe {
    uint32_t l[NLIMB];
    // reinterpret with a looser bound (always safe)
    template <int K2>
    ZK_DEV Fe<M, K2> as() const {
        static_assert(K2 >= K, "cannot tighten a bound by cast");
        Fe<M, K2> r;
#pragma unroll
        for (int i = 0; i < NLIMB; i++) r.l[i] = l[i];
        return r;
    }
};

// ZK_PAR_CARRY=1 (experiment, per translation unit): ONE parallel carry step instead of the ripple -- every limb hands its excess
// to its neighbour at once, so the eight steps do not depend on each other.  The result is "nearly normalised" (limbs <= 2^30 + 2
// after a difference of three operands): fine as a product operand (the column bound of tools/radix_budget.py moves by 2^-28) and as
// an operand of another sum, NOT for fe_canon / comparisons -- only translation units that canonicalise product outputs may use it.
#ifndef ZK_PAR_CARRY
#define ZK_PAR_CARRY 0
#endif
ZK_DEV void limbs_normalize(uint32_t r[NLIMB]) {
#if ZK_PAR_CARRY
    uint32_t c[NLIMB - 1];
#pragma unroll
    for (int i = 0; i < NLIMB - 1; i++) c[i] = r[i] >> LIMB_BITS, r[i] &= LIMB_MASK;
#pragma unroll
    for (int i = 0; i < NLIMB - 1; i++) r[i + 1] += c[i];
#else
#pragma unroll
    for (int i = 0; i < NLIMB - 1; i++) {
        r[i + 1] += r[i] >> LIMB_BITS;
        r[i] &= LIMB_MASK;
    }
#endif
}

template <class M, int Ka, int Kb>
ZK_DEV Fe<M, Ka + Kb> operator+(const Fe<M, Ka>& a, const Fe<M, Kb>& b) {
    static_assert(Ka + Kb <= KCAP, "magnitude overflow");
    Fe<M, Ka + Kb> r;
#pragma unroll
    for (int i = 0; i < NLIMB; i++) r.l[i] = a.l[i] + b.l[i];
    limbs_normalize(r.l);
    return r;
}
template <int Kb>
struct SubC {
    static constexpr int value = Kb < 4 ? 4 : Kb < 8 ? 8 : Kb < 16 ? 16 : Kb < 32 ? 32 : Kb < 64 ? 64 : Kb < 128 ? 128 : 256;
    static_assert(Kb < 256, "subtrahend too large");
};
template <class M, int Ka, int Kb>
ZK_DEV Fe<M, Ka + SubC<Kb>::value> operator-(const Fe<M, Ka>& a, const Fe<M, Kb>& b) {
    constexpr int C = SubC<Kb>::value;
    static_assert(Ka + C <= KCAP, "magnitude overflow");
    Fe<M, Ka + C> r;
#pragma unroll
    for (int i = 0; i < NLIMB; i++) {
        uint32_t s;
        if constexpr (C == 4) s = M::sub4[i];
        else if constexpr (C == 8) s = M::sub8[i];
        else if constexpr (C == 16) s = M::sub16[i];
        else if constexpr (C == 32) s = M::sub32[i];
        else if constexpr (C == 64) s = M::sub64[i];
        else if constexpr (C == 128) s = M::sub128[i];
        else s = M::sub256[i];
        r.l[i] = a.l[i] + s - b.l[i];
    }
    limbs_normalize(r.l);
    return r;
}
// a - b - c with one carry pass: a + C*M - b - c, C the smallest power of two > Kb + Kc.  The redundant form of C*M used
// by operator- has limbs >= 2^30 - 1 (one subtrahend limb); lending once more along the chain (+2^30, +2^30 - 1, ..., -1:
// the value is unchanged) makes them >= 2^31 - 2, enough for two subtrahend limbs, and keeps a + s below 2^32.
template <class M, int Ka, int Kb, int Kc>
ZK_DEV Fe<M, Ka + SubC<Kb + Kc>::value> fe_sub2(const Fe<M, Ka>& a, const Fe<M, Kb>& b, const Fe<M, Kc>& c) {
    constexpr int C = SubC<Kb + Kc>::value;
    static_assert(Ka + C <= KCAP, "magnitude overflow");
    Fe<M, Ka + C> r;
#pragma unroll
    for (int i = 0; i < NLIMB; i++) {
        uint32_t s;
        if constexpr (C == 4) s = M::sub4[i];
        else if constexpr (C == 8) s = M::sub8[i];
        else if constexpr (C == 16) s = M::sub16[i];
        else if constexpr (C == 32) s = M::sub32[i];
        else if constexpr (C == 64) s = M::sub64[i];
        else if constexpr (C == 128) s = M::sub128[i];
        else s = M::sub256[i];
        s += i == 0 ? (1u << LIMB_BITS) : i < NLIMB - 1 ? LIMB_MASK : 0xffffffffu;
        r.l[i] = a.l[i] + s - b.l[i] - c.l[i];
    }
    limbs_normalize(r.l);
    return r;
}
template <class M, int Ka>
ZK_DEV Fe<M, SubC<Ka>::value> fe_neg(const Fe<M, Ka>& a) {
    Fe<M, 0> z;
#pragma unroll
    for (int i = 0; i < NLIMB; i++) z.l[i] = 0;
    auto r = z - a;
    return r;
}
template <class M, int Ka>
ZK_DEV Fe<M, 2 * Ka> fe_dbl(const Fe<M, Ka>& a) {
    return a + a;
}

// Modulus limb i as an SGPR operand the optimiser cannot see through: left to itself the compiler strength-reduces
// m * (2^a - 2^b) into 64-bit shift/add/sub sequences that cost more VALU issue slots than the one v_mad_u64_u32
// they replace (ZK_LAUNDER_MOD=0 keeps the compiler's choice).  Zero limbs stay compile-time zeros (no instruction).
#ifndef ZK_LAUNDER_MOD
#define ZK_LAUNDER_MOD 1
#endif
template <class M>
ZK_DEV void mod_limbs(uint32_t md[NLIMB]) {
#pragma unroll
    for (int i = 0; i < NLIMB; i++) {
        md[i] = M::mod[i];
#if ZK_LAUNDER_MOD
        if (M::mod[i] != 0) asm("" : "+s"(md[i]));
#endif
    }
}
#ifndef ZK_PIN_LIMBS32
#define ZK_PIN_LIMBS32 1
#endif
// A product computed ALONE: the optimiser starts every column in a fresh accumulator while the previous column is being
// finished (m_k * M_0, shift) and joins the two with a 64-bit add -- it shortens the dependent chain (a dependent
// v_mad_u64_u32 issues every 15.6 cycles, an independent one every 4.2) at the price of 16 extra VALU instructions per product,
// each at the issue cost of a multiply-add.  Where a formula has NP INDEPENDENT products (the four of a Tom-256 addition's first
// and second half, the pairs of the P-256 laws) limbs_mont_mul_n below runs them in lock-step instead: every product is ONE
// dependent chain (mad64c hides the sums from the reassociation pass), consecutive instructions belong to different products,
// so the latency is covered by the other chains and the joins disappear: 212 -> 196 VALU instructions per product.
#ifndef ZK_SINGLE_CHAIN
#define ZK_SINGLE_CHAIN 1
#endif
// ZK_MAD_VOLATILE=1 (experiment, per translation unit): the empty asm statements of mad64c are volatile, so the machine scheduler may not move one chain's multiply-adds
// past another's.  Without it the scheduler sometimes strings a product's whole chain together to get registers back: the FIRST table walk of k_exp_commit_kt has 1 046
// hazard nops per loop body against ~200 in the two walks behind it, k_msm_bucket 592 at 96 VGPRs against 153 at 128.  Measured, same box (profiles/r06_ab_variants.txt (8)):
// no gain -- at three to five waves per SIMD the other waves cover a serialised chain (p256_exp_commit 29.2 -> 29.2 ms, bucket sums 17.6 -> 16.9 ms but the P-256 pass beside
// them 9.5 -> 10.7).  Default off.
#ifndef ZK_MAD_VOLATILE
#define ZK_MAD_VOLATILE 0
#endif
ZK_DEV uint64_t mad64c(uint32_t a, uint32_t b, uint64_t c) {
    uint64_t r = (uint64_t)a * b + c;
#if ZK_SINGLE_CHAIN && !defined(ZK_HOST_BUILD)
#if ZK_MAD_VOLATILE
    asm volatile("" : "+v"(r));   // ... and, volatile, it keeps its place among the other chains' sums: the lock-step order of the source IS the schedule
#else
    asm("" : "+v"(r));   // no instruction: the sum is opaque to the reassociation pass
#endif
#endif
    return r;
}
// Montgomery product, product-scanning with a single 64-bit accumulator (no carry flags).
template <class M>
ZK_DEV void limbs_mont_mul(uint32_t out[NLIMB], const uint32_t a[NLIMB], const uint32_t b[NLIMB]) {
    uint64_t acc = 0;
    uint32_t m[NLIMB], md[NLIMB];
    mod_limbs<M>(md);
#pragma unroll
    for (int k = 0; k < NLIMB; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) acc = mad64(a[i], b[k - i], acc);
#pragma unroll
        for (int i = 0; i < k; i++) acc = mad64(m[i], md[k - i], acc);
        m[k] = ((uint32_t)acc * M::n0) & LIMB_MASK;
        acc = mad64(m[k], md[0], acc);
        acc >>= LIMB_BITS;
    }
#pragma unroll
    for (int k = NLIMB; k < 2 * NLIMB - 1; k++) {
#pragma unroll
        for (int i = k - (NLIMB - 1); i < NLIMB; i++) acc = mad64(a[i], b[k - i], acc);
#pragma unroll
        for (int i = k - (NLIMB - 1); i < NLIMB; i++) acc = mad64(m[i], md[k - i], acc);
        out[k - NLIMB] = (uint32_t)acc & LIMB_MASK;
        acc >>= LIMB_BITS;
    }
    out[NLIMB - 1] = (uint32_t)acc;
#if ZK_PIN_LIMBS32
    // Keep every result limb a 32-bit VGPR value (empty asm, no instruction).  Without this the optimiser carries some
    // products across basic blocks as the 64-bit (acc & mask) they were truncated from; instruction selection works per
    // block, cannot prove the high halves zero there, and multiplies them as 64 x 32 bits: +72 v_mad_u64_u32 and
    // +144 v_mov_b32 per table addition in k_tom_commit's loop (ISA inspection; ZK_PIN_LIMBS32=0 shows the old code).
#pragma unroll
    for (int i = 0; i < NLIMB; i++) asm("" : "+v"(out[i]));
#endif
}
// NP independent Montgomery products in lock-step (see the note above mad64c).  Row-interleaved: column k of every product
// advances by one multiply-add per row, so a product's own chain is touched every NP-th instruction.
template <class M, int NP>
ZK_DEV void limbs_mont_mul_n(uint32_t (&out)[NP][NLIMB], const uint32_t (&a)[NP][NLIMB], const uint32_t (&b)[NP][NLIMB]) {
    uint64_t acc[NP];
    uint32_t m[NP][NLIMB], md[NLIMB];
    mod_limbs<M>(md);
#pragma unroll
    for (int p = 0; p < NP; p++) acc[p] = 0;
#pragma unroll
    for (int k = 0; k < NLIMB; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++)
#pragma unroll
            for (int p = 0; p < NP; p++) acc[p] = mad64c(a[p][i], b[p][k - i], acc[p]);
#pragma unroll
        for (int i = 0; i < k; i++)
#pragma unroll
            for (int p = 0; p < NP; p++) acc[p] = mad64c(m[p][i], md[k - i], acc[p]);
#pragma unroll
        for (int p = 0; p < NP; p++) m[p][k] = ((uint32_t)acc[p] * M::n0) & LIMB_MASK;
#pragma unroll
        for (int p = 0; p < NP; p++) acc[p] = mad64c(m[p][k], md[0], acc[p]);
#pragma unroll
        for (int p = 0; p < NP; p++) acc[p] >>= LIMB_BITS;
    }
#pragma unroll
    for (int k = NLIMB; k < 2 * NLIMB - 1; k++) {
#pragma unroll
        for (int i = k - (NLIMB - 1); i < NLIMB; i++)
#pragma unroll
            for (int p = 0; p < NP; p++) acc[p] = mad64c(a[p][i], b[p][k - i], acc[p]);
#pragma unroll
        for (int i = k - (NLIMB - 1); i < NLIMB; i++)
#pragma unroll
            for (int p = 0; p < NP; p++) acc[p] = mad64c(m[p][i], md[k - i], acc[p]);
#pragma unroll
        for (int p = 0; p < NP; p++) out[p][k - NLIMB] = (uint32_t)acc[p] & LIMB_MASK;
#pragma unroll
        for (int p = 0; p < NP; p++) acc[p] >>= LIMB_BITS;
    }
#pragma unroll
    for (int p = 0; p < NP; p++) out[p][NLIMB - 1] = (uint32_t)acc[p];
#if ZK_PIN_LIMBS32
#pragma unroll
    for (int p = 0; p < NP; p++)
#pragma unroll
        for (int i = 0; i < NLIMB; i++) asm("" : "+v"(out[p][i]));
#endif
}
template <class M, int Ka, int Kb>
ZK_DEV Fe<M, 2> operator*(const Fe<M, Ka>& a, const Fe<M, Kb>& b) {
    static_assert((long)Ka * Kb <= M::kmax, "Montgomery input magnitudes too large");
    Fe<M, 2> r;
    uint32_t o[NLIMB];
    limbs_mont_mul<M>(o, a.l, b.l);
#pragma unroll
    for (int i = 0; i < NLIMB; i++) r.l[i] = o[i];
    return r;
}
// Montgomery SQUARE: the 36 off-diagonal products a_i a_j (i < j) enter once with a doubled limb (2 a_i < 2^31 still fits the
// 32-bit multiplier input), the 9 squares once: 45 + 81 multiply-adds instead of 81 + 81.  The column sums are the same
// numbers as in limbs_mont_mul(a, a), so the same bounds hold.
template <class M>
ZK_DEV void limbs_mont_sqr(uint32_t out[NLIMB], const uint32_t a[NLIMB]) {
    uint64_t acc = 0;
    uint32_t m[NLIMB], md[NLIMB], a2[NLIMB];
    mod_limbs<M>(md);
#pragma unroll
    for (int i = 0; i < NLIMB; i++) a2[i] = a[i] << 1;
#pragma unroll
    for (int k = 0; k < NLIMB; k++) {
#pragma unroll
        for (int i = 0; 2 * i < k; i++) acc = mad64(a2[i], a[k - i], acc);
        if (k % 2 == 0) acc = mad64(a[k / 2], a[k / 2], acc);
#pragma unroll
        for (int i = 0; i < k; i++) acc = mad64(m[i], md[k - i], acc);
        m[k] = ((uint32_t)acc * M::n0) & LIMB_MASK;
        acc = mad64(m[k], md[0], acc);
        acc >>= LIMB_BITS;
    }
#pragma unroll
    for (int k = NLIMB; k < 2 * NLIMB - 1; k++) {
#pragma unroll
        for (int i = k - (NLIMB - 1); 2 * i < k; i++) acc = mad64(a2[i], a[k - i], acc);
        if (k % 2 == 0) acc = mad64(a[k / 2], a[k / 2], acc);
#pragma unroll
        for (int i = k - (NLIMB - 1); i < NLIMB; i++) acc = mad64(m[i], md[k - i], acc);
        out[k - NLIMB] = (uint32_t)acc & LIMB_MASK;
        acc >>= LIMB_BITS;
    }
    out[NLIMB - 1] = (uint32_t)acc;
#if ZK_PIN_LIMBS32
#pragma unroll
    for (int i = 0; i < NLIMB; i++) asm("" : "+v"(out[i]));
#endif
}
template <class M, int Ka>
ZK_DEV Fe<M, 2> fe_sqr(const Fe<M, Ka>& a) {
    static_assert((long)Ka * Ka <= M::kmax, "Montgomery input magnitudes too large");
    Fe<M, 2> r;
    uint32_t o[NLIMB];
    limbs_mont_sqr<M>(o, a.l);
#pragma unroll
    for (int i = 0; i < NLIMB; i++) r.l[i] = o[i];
    return r;
}
// r_i = a_i * b_i for independent products: BATCH = true computes them in lock-step (limbs_mont_mul_n: fewer instructions, more
// live registers), false one after the other.  A translation unit picks per curve (ZK_BATCH_TOM / ZK_BATCH_P256, curve.h): the
// register-hungry verifier kernels keep the sequential form (k_v_p256_straus would spill 1.2 KB per lane otherwise).
template <bool BATCH, class M, int K0a, int K0b, int K1a, int K1b>
ZK_DEV void fe_mul2(Fe<M, 2>& r0, Fe<M, 2>& r1, const Fe<M, K0a>& a0, const Fe<M, K0b>& b0, const Fe<M, K1a>& a1, const Fe<M, K1b>& b1) {
    static_assert((long)K0a * K0b <= M::kmax && (long)K1a * K1b <= M::kmax, "Montgomery input magnitudes too large");
#if !defined(ZK_HOST_BUILD)
    if constexpr (BATCH) {
    uint32_t a[2][NLIMB], b[2][NLIMB], o[2][NLIMB];
#pragma unroll
    for (int i = 0; i < NLIMB; i++) a[0][i] = a0.l[i], b[0][i] = b0.l[i], a[1][i] = a1.l[i], b[1][i] = b1.l[i];
    limbs_mont_mul_n<M, 2>(o, a, b);
#pragma unroll
    for (int i = 0; i < NLIMB; i++) r0.l[i] = o[0][i], r1.l[i] = o[1][i];
    return;
    }
#endif
    r0 = a0 * b0, r1 = a1 * b1;
}
template <bool BATCH, class M, int K0a, int K0b, int K1a, int K1b, int K2a, int K2b>
ZK_DEV void fe_mul3(Fe<M, 2>& r0, Fe<M, 2>& r1, Fe<M, 2>& r2, const Fe<M, K0a>& a0, const Fe<M, K0b>& b0, const Fe<M, K1a>& a1, const Fe<M, K1b>& b1,
                    const Fe<M, K2a>& a2, const Fe<M, K2b>& b2) {
    static_assert((long)K0a * K0b <= M::kmax && (long)K1a * K1b <= M::kmax && (long)K2a * K2b <= M::kmax, "Montgomery input magnitudes too large");
#if !defined(ZK_HOST_BUILD)
    if constexpr (BATCH) {
    uint32_t a[3][NLIMB], b[3][NLIMB], o[3][NLIMB];
#pragma unroll
    for (int i = 0; i < NLIMB; i++) a[0][i] = a0.l[i], b[0][i] = b0.l[i], a[1][i] = a1.l[i], b[1][i] = b1.l[i], a[2][i] = a2.l[i], b[2][i] = b2.l[i];
    limbs_mont_mul_n<M, 3>(o, a, b);
#pragma unroll
    for (int i = 0; i < NLIMB; i++) r0.l[i] = o[0][i], r1.l[i] = o[1][i], r2.l[i] = o[2][i];
    return;
    }
#endif
    r0 = a0 * b0, r1 = a1 * b1, r2 = a2 * b2;
}
template <bool BATCH, class M, int K0a, int K0b, int K1a, int K1b, int K2a, int K2b, int K3a, int K3b>
ZK_DEV void fe_mul4(Fe<M, 2>& r0, Fe<M, 2>& r1, Fe<M, 2>& r2, Fe<M, 2>& r3, const Fe<M, K0a>& a0, const Fe<M, K0b>& b0, const Fe<M, K1a>& a1,
                    const Fe<M, K1b>& b1, const Fe<M, K2a>& a2, const Fe<M, K2b>& b2, const Fe<M, K3a>& a3, const Fe<M, K3b>& b3) {
    static_assert((long)K0a * K0b <= M::kmax && (long)K1a * K1b <= M::kmax && (long)K2a * K2b <= M::kmax && (long)K3a * K3b <= M::kmax,
                  "Montgomery input magnitudes too large");
#if !defined(ZK_HOST_BUILD)
    if constexpr (BATCH) {
    uint32_t a[4][NLIMB], b[4][NLIMB], o[4][NLIMB];
#pragma unroll
    for (int i = 0; i < NLIMB; i++)
        a[0][i] = a0.l[i], b[0][i] = b0.l[i], a[1][i] = a1.l[i], b[1][i] = b1.l[i], a[2][i] = a2.l[i], b[2][i] = b2.l[i], a[3][i] = a3.l[i], b[3][i] = b3.l[i];
    limbs_mont_mul_n<M, 4>(o, a, b);
#pragma unroll
    for (int i = 0; i < NLIMB; i++) r0.l[i] = o[0][i], r1.l[i] = o[1][i], r2.l[i] = o[2][i], r3.l[i] = o[3][i];
    return;
    }
#endif
    r0 = a0 * b0, r1 = a1 * b1, r2 = a2 * b2, r3 = a3 * b3;
}
template <class M, int K = 1>
ZK_DEV Fe<M, K> fe_const(const uint32_t c[NLIMB]) {
    Fe<M, K> r;
#pragma unroll
    for (int i = 0; i < NLIMB; i++) r.l[i] = c[i];
    return r;
}
template <class M>
ZK_DEV Fe<M, 1> fe_zero() {
    Fe<M, 1> r;
#pragma unroll
    for (int i = 0; i < NLIMB; i++) r.l[i] = 0;
    return r;
}
template <class M>
ZK_DEV Fe<M, 1> fe_one_mont() {
    return fe_const<M, 1>(M::one);
}
// bring any magnitude back to < 2M (one Montgomery multiplication by R mod M)
template <class M, int Ka>
ZK_DEV Fe<M, 2> fe_reduce(const Fe<M, Ka>& a) {
    return a * fe_one_mont<M>();
}

// value >= M ?  (normalised limbs)
template <class M>
ZK_DEV bool limbs_geq_mod(const uint32_t a[NLIMB]) {
    bool ge = true;
#pragma unroll
    for (int i = 0; i < NLIMB; i++) {  // least significant first: more significant limbs override
        if (a[i] > M::mod[i]) ge = true;
        else if (a[i] < M::mod[i]) ge = false;
    }
    return ge;
}
// canonical representative in [0, M) of a value < 4M
template <class M, int Ka>
ZK_DEV Fe<M, 1> fe_canon(const Fe<M, Ka>& a) {
    static_assert(Ka <= 4, "canonicalise only small values");
    Fe<M, 1> r;
#pragma unroll
    for (int i = 0; i < NLIMB; i++) r.l[i] = a.l[i];
#pragma unroll
    for (int rep = 0; rep < Ka - 1 + (Ka == 1); rep++) {
        if (limbs_geq_mod<M>(r.l)) {
            int32_t borrow = 0;
#pragma unroll
            for (int i = 0; i < NLIMB; i++) {
                int32_t d = (int32_t)r.l[i] - (int32_t)M::mod[i] + borrow;
                borrow = d >> 31;
                r.l[i] = (uint32_t)d & LIMB_MASK;
            }
        }
    }
    return r;
}
// Montgomery -> canonical plain value in [0, M)
template <class M, int Ka>
ZK_DEV Fe<M, 1> fe_from_mont(const Fe<M, Ka>& a) {
    Fe<M, 1> one = fe_zero<M>();
    one.l[0] = 1;
    Fe<M, 2> r = a * one;  // (a + m*M)/R <= M
    return fe_canon(r);
}
// plain (normalised) -> Montgomery
template <class M, int Ka>
ZK_DEV Fe<M, 2> fe_to_mont(const Fe<M, Ka>& a) {
    return a * fe_const<M, 1>(M::r2);
}
template <class M, int Ka>
ZK_DEV bool fe_is_zero(const Fe<M, Ka>& a) {  // value == 0 mod M ?
    Fe<M, 1> c;
    if constexpr (Ka <= 4) c = fe_canon(a);
    else c = fe_canon(fe_reduce(a));
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < NLIMB; i++) o |= c.l[i];
    return o == 0;
}
template <class M, int Ka, int Kb>
ZK_DEV bool fe_eq(const Fe<M, Ka>& a, const Fe<M, Kb>& b) {
    return fe_is_zero(fe_reduce(a - b));
}
// canonical a, b -> canonical (a - b) mod M / (a + b) mod M   (scalar arithmetic on plain values)
template <class M>
ZK_DEV Fe<M, 1> fe_sub_mod(const Fe<M, 1>& a, const Fe<M, 1>& b) {
    Fe<M, 1> r;
    int32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < NLIMB; i++) {
        int32_t t = (int32_t)a.l[i] - (int32_t)b.l[i] + borrow;
        borrow = t >> 31;
        r.l[i] = (uint32_t)t & LIMB_MASK;
    }
    if (borrow) {
        uint32_t carry = 0;
#pragma unroll
        for (int i = 0; i < NLIMB; i++) {
            uint32_t t = r.l[i] + M::mod[i] + carry;
            r.l[i] = t & LIMB_MASK;
            carry = t >> LIMB_BITS;
        }
    }
    return r;
}
template <class M>
ZK_DEV Fe<M, 1> fe_add_mod(const Fe<M, 1>& a, const Fe<M, 1>& b) {
    return fe_canon(a + b);
}
// plain canonical a, b -> plain canonical a*b mod M
template <class M>
ZK_DEV Fe<M, 1> fe_mul_mod(const Fe<M, 1>& a, const Fe<M, 1>& b) {
    return fe_canon(fe_to_mont(a) * b);
}
template <class M, int K>
ZK_DEV Fe<M, K> fe_select(bool c, const Fe<M, K>& a, const Fe<M, K>& b) {  // c ? a : b
    Fe<M, K> r;
#pragma unroll
    for (int i = 0; i < NLIMB; i++) r.l[i] = c ? a.l[i] : b.l[i];
    return r;
}

// Montgomery product for the places where ONE thread multiplies while everybody else waits for it (the Fermat inversions: thread 0 of a
// normaliser workgroup, the front end's one thread per proof): operand-scanning, one 64-bit accumulator PER COLUMN, so the nine
// multiply-adds of a row are independent of each other and issue back to back (4.2 cycles each) instead of one behind the other
// (15.8 cycles, profiles/r03_valu_peak_microbench.txt).  The column sums are those of limbs_mont_mul -- the same partial products and
// carries, added in another order -- so the result limbs are identical.  Eighteen live accumulators: not for kernels short of registers.
template <class M>
ZK_DEV void limbs_mont_mul_rows(uint32_t out[NLIMB], const uint32_t a[NLIMB], const uint32_t b[NLIMB]) {
    uint64_t T[2 * NLIMB];
    uint32_t md[NLIMB];
    mod_limbs<M>(md);
#pragma unroll
    for (int k = 0; k < 2 * NLIMB; k++) T[k] = 0;
#pragma unroll
    for (int i = 0; i < NLIMB; i++) {
#pragma unroll
        for (int j = 0; j < NLIMB; j++) T[i + j] = mad64(a[i], b[j], T[i + j]);
        if (i > 0) T[i] += T[i - 1] >> LIMB_BITS;   // column i is complete now: every a_r b_(i-r), every m_r M_(i-r) with r < i, the carry
        const uint32_t m = ((uint32_t)T[i] * M::n0) & LIMB_MASK;
#pragma unroll
        for (int j = 0; j < NLIMB; j++) T[i + j] = mad64(m, md[j], T[i + j]);
    }
    uint64_t carry = T[NLIMB - 1] >> LIMB_BITS;
#pragma unroll
    for (int k = NLIMB; k < 2 * NLIMB - 1; k++) {
        const uint64_t v = T[k] + carry;
        out[k - NLIMB] = (uint32_t)v & LIMB_MASK;
        carry = v >> LIMB_BITS;
    }
    out[NLIMB - 1] = (uint32_t)carry;
}
// Measured (profiles/r04_ab_variants.txt, same box): no effect -- tom_normalize 10.82 -> 10.96 ms, p256_front 3.16 -> 3.26 ms per step, proofs/s
// unchanged: the inversions are not what those kernels wait for.  Default off; kept as the record of the experiment.
#ifndef ZK_POW_ROWS
#define ZK_POW_ROWS 0
#endif
template <class M>
ZK_DEV Fe<M, 2> fe_mul_rows(const Fe<M, 2>& a, const Fe<M, 2>& b) {
#if ZK_POW_ROWS
    Fe<M, 2> r;
    limbs_mont_mul_rows<M>(r.l, a.l, b.l);
    return r;
#else
    return a * b;
#endif
}
// a^e for a public exponent e given as 9 little-endian 32-bit words (right-to-left binary), Montgomery domain
template <class M>
ZK_DEV_NOINLINE Fe<M, 2> fe_pow_words(Fe<M, 2> a, const uint32_t e[NLIMB]) {
    static_assert(4 <= M::kmax, "Montgomery input magnitudes too large");
    Fe<M, 2> acc = fe_one_mont<M>().template as<2>(), base = a;
    for (int w = 0; w < NLIMB; w++) {
        uint32_t ew = e[w];
        int nb = M::bits - 32 * w;
        if (nb > 32) nb = 32;
        for (int b = 0; b < nb; b++) {
            if ((ew >> b) & 1) acc = fe_mul_rows(acc, base);
#if ZK_POW_ROWS
            base = fe_mul_rows(base, base);
#else
            base = fe_sqr(base);
#endif
        }
    }
    return acc;
}
// Montgomery-domain inverse by Fermat (a^(M-2)); inv(0) = 0 like the reference's invMod (big.ts:113-119).  Until round 6 THE inversion of the engine; now the
// cross-check of fe_inv_gcd below (tests/test_host_arith.py, tools/coop_bench.hip: 164 us against 38 us for one inversion in one lane).
template <class M>
ZK_DEV Fe<M, 2> fe_inv_fermat(const Fe<M, 2>& a) {
    return fe_pow_words<M>(a, M::exp_m2);
}

// Inverse by DIVSTEPS (Bernstein-Yang 2019, "Fast constant-time gcd computation and modular inversion") instead of Fermat's ~390 dependent
// products: where ONE lane inverts while a workgroup -- at one proof per call, the whole GPU -- waits (block_inverse, the front ends), the chain
// length is the cost.  Rounds of 30 divsteps on the low 32 bits of (f, g) = (M, x) give a 2x2 transition matrix with entries |.| <= 2^30, applied
// to the full-length (f, g) and, modulo M, to (d, e) = (0, 1): the radix-2^30 limbs of this file are exactly the digits that takes (signed top
// limb).  d f^-1 = x^-1 once g = 0.  (49 * 258 + 57) / 17 = 747 divsteps suffice for inputs below 2^258 (Theorem 11.2 there), i.e. 25 rounds;
// random inputs need 18-19 and the default build stops at g = 0 (-DZK_UNIFORM_CF=1: always 25 rounds, no data-dependent branch in here).
// |d|, |e| grow by at most M per round (the multiple of M that clears the low limb is < 2^30 M), so they stay below 26 M.
// inv(0) = 0 like the reference's invMod (big.ts:113-119): g = 0 from the start leaves d = 0.
#ifndef ZK_UNIFORM_CF
#define ZK_UNIFORM_CF 0
#endif
// LOCKSTEP: every lane of the wave inverts its own element (the per-point normalisers of a small call): the fixed 30-step loop, because lanes that take
// different numbers of iterations make the wave run their union (256 chains: 39.2 us fixed, 41.1 us variable; one chain: 37.3 -> 30.6 us variable).
template <class M, bool LOCKSTEP = false>
ZK_DEV_NOINLINE Fe<M, 2> fe_inv_gcd(const Fe<M, 2> a) {
    const Fe<M, 1> x = fe_canon(a);
    int32_t f[NLIMB], g[NLIMB], d[NLIMB], e[NLIMB];
#pragma unroll
    for (int i = 0; i < NLIMB; i++) f[i] = (int32_t)M::mod[i], g[i] = (int32_t)x.l[i], d[i] = 0, e[i] = 0;
    e[0] = 1;
    int32_t eta = -1;
#pragma unroll 1
    for (int round = 0; round < 25; round++) {
#if !ZK_UNIFORM_CF
        uint32_t any = 0;
#pragma unroll
        for (int i = 0; i < NLIMB; i++) any |= (uint32_t)g[i];
        if (any == 0) break;
#endif
        uint32_t fw = (uint32_t)f[0] | ((uint32_t)f[1] << LIMB_BITS), gw = (uint32_t)g[0] | ((uint32_t)g[1] << LIMB_BITS);
        uint32_t u = 1, v = 0, q = 0, r = 1;
        if constexpr (LOCKSTEP || ZK_UNIFORM_CF) {
#pragma unroll 6
        for (int i = 0; i < LIMB_BITS; i++) {
            uint32_t c1 = (uint32_t)(eta >> 31);            // eta < 0
            const uint32_t c2 = 0u - (gw & 1u);             // g odd
            const uint32_t xf = (fw ^ c1) - c1, xu = (u ^ c1) - c1, xv = (v ^ c1) - c1;   // (f, u, v), negated if eta < 0
            gw += xf & c2, q += xu & c2, r += xv & c2;
            c1 &= c2;                                       // eta < 0 and g odd: swap
            eta = (int32_t)(((uint32_t)eta ^ c1) - (c1 + 1u));
            fw += gw & c1, u += q & c1, v += r & c1;
            gw >>= 1, u <<= 1, v <<= 1;
        }
        } else {
        // The same 30 divsteps, several per iteration (the default build's control flow depends on data anyway): a run of even g's is one shift; with g odd and
        // delta <= 0 (eta >= 0) the next eta + 1 steps are "add f if odd, halve", i.e. g <- (g + w f) / 2^k with the w that clears k low bits -- up to six at a
        // time here, w = -g f^-1 mod 2^k from one Newton step on f (f f = 1 mod 8); delta > 0 and g odd is the swap (f, g) <- (g, -f) followed by the same.
        // About 8 iterations of ~20 instructions instead of 30 of 17: one inversion in one lane 38 -> 27 us.
        for (int i = LIMB_BITS;;) {
            const int z = __builtin_ctz(gw | (0xffffffffu << i));
            gw >>= z, u <<= z, v <<= z, eta -= z, i -= z;
            if (i == 0) break;
            if (eta < 0) {
                eta = -eta;
                uint32_t t = fw;
                fw = gw, gw = 0u - t;
                t = u, u = q, q = 0u - t;
                t = v, v = r, r = 0u - t;
            }
            const int limit = eta + 1 > i ? i : eta + 1;
            const uint32_t m = (0xffffffffu >> (32 - limit)) & 63u;
            const uint32_t fi = fw * (2u - fw * fw);            // f^-1 mod 2^6
            const uint32_t w = (0u - gw * fi) & m;
            gw += fw * w, q += u * w, r += v * w;
        }
        }
        const int64_t su = (int32_t)u, sv = (int32_t)v, sq = (int32_t)q, sr = (int32_t)r;
        // (f, g) <- matrix * (f, g) / 2^30, exactly
        int64_t cf = su * f[0] + sv * g[0], cg = sq * f[0] + sr * g[0];
        cf >>= LIMB_BITS, cg >>= LIMB_BITS;
#pragma unroll
        for (int k = 1; k < NLIMB; k++) {
            cf += su * f[k] + sv * g[k], cg += sq * f[k] + sr * g[k];
            f[k - 1] = (int32_t)((uint32_t)cf & LIMB_MASK), g[k - 1] = (int32_t)((uint32_t)cg & LIMB_MASK);
            cf >>= LIMB_BITS, cg >>= LIMB_BITS;
        }
        f[NLIMB - 1] = (int32_t)cf, g[NLIMB - 1] = (int32_t)cg;
        // (d, e) <- matrix * (d, e) / 2^30 mod M: add the multiple of M that clears the low limb (the Montgomery quotient digit)
        int64_t cd = su * d[0] + sv * e[0], ce = sq * d[0] + sr * e[0];
        const int64_t md = (int64_t)(((uint32_t)cd * M::n0) & LIMB_MASK), me = (int64_t)(((uint32_t)ce * M::n0) & LIMB_MASK);
        cd += md * (int64_t)M::mod[0], ce += me * (int64_t)M::mod[0];
        cd >>= LIMB_BITS, ce >>= LIMB_BITS;
#pragma unroll
        for (int k = 1; k < NLIMB; k++) {
            cd += su * d[k] + sv * e[k] + md * (int64_t)M::mod[k], ce += sq * d[k] + sr * e[k] + me * (int64_t)M::mod[k];
            d[k - 1] = (int32_t)((uint32_t)cd & LIMB_MASK), e[k - 1] = (int32_t)((uint32_t)ce & LIMB_MASK);
            cd >>= LIMB_BITS, ce >>= LIMB_BITS;
        }
        d[NLIMB - 1] = (int32_t)cd, e[NLIMB - 1] = (int32_t)ce;
    }
    // f = +-1 (or M when x = 0, d = 0): x^-1 = f d.  +-d + 32 M is positive (|d| < 26 M) and < 64 M; one product by R^3 reduces it and brings the
    // result to the domain of the input: (x R)^-1 R^2 = x^-1 R.
    const bool neg = f[NLIMB - 1] < 0;
    Fe<M, 64> t;
#pragma unroll
    for (int i = 0; i < NLIMB; i++) t.l[i] = M::sub32[i] + (uint32_t)(neg ? -d[i] : d[i]);
    limbs_normalize(t.l);
    return t * fe_const<M, 1>(M::r3);
}

// Montgomery-domain inverse, inv(0) = 0: what every caller uses (block_inverse, the front ends, the table builders).
// (-DZK_INV_FERMAT=1 builds round 5's Fermat inversions back in: the A/B library of profiles/r06_ab_variants.txt.)
#ifndef ZK_INV_FERMAT
#define ZK_INV_FERMAT 0
#endif
template <class M, bool LOCKSTEP = false>
ZK_DEV Fe<M, 2> fe_inv(const Fe<M, 2>& a) {
#if ZK_INV_FERMAT
    return fe_inv_fermat<M>(a);
#else
    return fe_inv_gcd<M, LOCKSTEP>(a);
#endif
}

// ---- plain 32-bit-word <-> 30-bit-limb conversions ----
// words: little-endian 32-bit words (NW = 8 for 256-bit values, 9 for Tom coordinates)
template <int NW>
ZK_DEV void limbs_from_words(uint32_t l[NLIMB], const uint32_t w[NW]) {
#pragma unroll
    for (int i = 0; i < NLIMB; i++) {
        const int bit = i * LIMB_BITS, wi = bit / 32, sh = bit % 32;
        uint32_t v = 0;  // 32-bit shifts only: a 64-bit (w[wi+1]:w[wi]) pair makes the compiler park w[] in scratch to reload it as dwordx2
        if (wi < NW) v = w[wi] >> sh;
        if (sh > 32 - LIMB_BITS && wi + 1 < NW) v |= w[wi + 1] << (32 - sh);
        l[i] = v & LIMB_MASK;
    }
}
template <int NW>
ZK_DEV void words_from_limbs(uint32_t w[NW], const uint32_t l[NLIMB]) {
#pragma unroll
    for (int i = 0; i < NW; i++) {
        const int lo = (32 * i) / LIMB_BITS, sh = (32 * i) % LIMB_BITS;
        uint64_t v = (uint64_t)l[lo] >> sh;
        if (lo + 1 < NLIMB) v |= (uint64_t)l[lo + 1] << (LIMB_BITS - sh);
        if (lo + 2 < NLIMB && (2 * LIMB_BITS - sh) < 32) v |= (uint64_t)l[lo + 2] << (2 * LIMB_BITS - sh);
        w[i] = (uint32_t)v;
    }
}
template <class M, int NW>
ZK_DEV Fe<M, 1> fe_from_words(const uint32_t w[NW]) {  // caller guarantees value < M (or accepts < 2^(32 NW))
    Fe<M, 1> r;
    limbs_from_words<NW>(r.l, w);
    return r;
}
// a >= m over NW little-endian words
template <int NW>
ZK_DEV bool words_geq(const uint32_t a[NW], const uint32_t m[NW]) {
    bool ge = true;
#pragma unroll
    for (int i = 0; i < NW; i++) {
        if (a[i] > m[i]) ge = true;
        else if (a[i] < m[i]) ge = false;
    }
    return ge;
}
// reduce a 256-bit plain value (8 words) modulo a 256-bit modulus M (q or n): at most one subtraction since
// 2^256 < 2M.  Mirrors Scalar's constructor reduction (group.ts:164-167).
template <class M>
ZK_DEV Fe<M, 1> fe_from_words256_reduce(const uint32_t w[8]) {
    Fe<M, 2> r;
    limbs_from_words<8>(r.l, w);
    return fe_canon(r);
}
