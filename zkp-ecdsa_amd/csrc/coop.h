// Lane-cooperative field arithmetic for the DEPENDENT CHAINS of the engine (gfx950).
//
// Everything else in csrc/ keeps one field element per lane (field.h): right where a launch has thousands of independent elements, and useless where ONE
// chain of point operations is all there is -- the 16 w doublings behind a bucket reduction (k_msm_red_last), the Horner walk over the P-256 windows
// (k_pm_final), the 256 doublings of a proof's table of R (k_rtab_base): one lane issues ~1 600 multiplier instructions per doubling while 63 idle.
// Here a field element is spread over a 16-lane DPP ROW, limb j (radix 2^30, the limbs of field.h) in lane j of the row, lanes 9..15 holding zero, and a
// wave holds FOUR elements, one per row -- the four independent products of an Edwards addition / doubling half, or X, Y, Z of a P-256 point:
//   * Montgomery product (co_mul): nine rounds of  t += a_i * b;  m = t_0 * n0 mod 2^30;  t += m * M;  t = (t >> 30) + lo30(t of the lane above)
//     -- a_i and m reach the row by DPP row_newbcast, the limb shift by DPP row_shl:1; two v_mad_u64_u32 per lane and round instead of 18 per round in
//     one lane.  Carries never ripple: every lane keeps the high part of its column (it belongs to the column the lane holds next round) and passes
//     only the low 30 bits down, so the accumulators stay below 2^62 and operands need not be normalised limb by limb (limbs up to 2^30 + 8 are fine).
//   * add / sub: one v_add / v_sub per ELEMENT plus one parallel carry step (each lane hands its excess to the lane above through DPP row_shr:1).
//   * moving elements between rows (the x + y, E = S - A - B ... of the curve formulas): ds_bpermute_b32, a zero operand is lane 15 of the own row.
// Values are the same residues as in the one-lane code (the same formulas, other multiples of the modulus on the way), so canonical outputs are
// bit-identical.  tests/test_host_arith.py runs THIS source on the host through the SIMT emulation below (64 lanes as arrays) against the oracle.
// Reference counterparts: the BigInt `%` arithmetic of src/curves/edwards.ts:141-183 and weier.ts:133-230; the dependent chains are group.ts:133-152's
// double-and-add in the shapes the engine's batched checks leave over.
#pragma once
#include "curve.h"

#ifdef ZK_HOST_BUILD
// ---- SIMT emulation for the CPU test tier: a "register" is 64 lane values
struct CoU32 {
    uint32_t v[64];
};
struct CoU64 {
    uint64_t v[64];
};
#define CO_LANES for (int i_ = 0; i_ < 64; i_++)
inline CoU32 co_splat(uint32_t c) {
    CoU32 r;
    CO_LANES r.v[i_] = c;
    return r;
}
inline CoU32 co_lane() {
    CoU32 r;
    CO_LANES r.v[i_] = (uint32_t)i_;
    return r;
}
#define CO_BINOP(op)                                  \
    inline CoU32 operator op(CoU32 a, CoU32 b) {      \
        CoU32 r;                                      \
        CO_LANES r.v[i_] = a.v[i_] op b.v[i_];        \
        return r;                                     \
    }                                                 \
    inline CoU32 operator op(CoU32 a, uint32_t b) {   \
        CoU32 r;                                      \
        CO_LANES r.v[i_] = a.v[i_] op b;              \
        return r;                                     \
    }
CO_BINOP(+) CO_BINOP(-) CO_BINOP(&) CO_BINOP(|) CO_BINOP(^) CO_BINOP(*) CO_BINOP(>>) CO_BINOP(<<)
#undef CO_BINOP
inline CoU32 co_eq(CoU32 a, uint32_t b) {
    CoU32 r;
    CO_LANES r.v[i_] = a.v[i_] == b ? 1u : 0u;
    return r;
}
inline CoU32 co_lt(CoU32 a, uint32_t b) {
    CoU32 r;
    CO_LANES r.v[i_] = a.v[i_] < b ? 1u : 0u;
    return r;
}
inline CoU32 co_sel(CoU32 c, CoU32 a, CoU32 b) {
    CoU32 r;
    CO_LANES r.v[i_] = c.v[i_] ? a.v[i_] : b.v[i_];
    return r;
}
inline CoU64 co_zero64() {
    CoU64 r;
    CO_LANES r.v[i_] = 0;
    return r;
}
inline CoU64 co_mad(CoU32 a, CoU32 b, CoU64 c) {
    CoU64 r;
    CO_LANES r.v[i_] = (uint64_t)a.v[i_] * b.v[i_] + c.v[i_];
    return r;
}
inline CoU32 co_lo(CoU64 c) {
    CoU32 r;
    CO_LANES r.v[i_] = (uint32_t)c.v[i_];
    return r;
}
inline CoU64 co_shr30_add(CoU64 c, CoU32 a) {   // (c >> 30) + a
    CoU64 r;
    CO_LANES r.v[i_] = (c.v[i_] >> LIMB_BITS) + a.v[i_];
    return r;
}
inline CoU32 co_hi30(CoU64 c) {   // (uint32_t)(c >> 30)
    CoU32 r;
    CO_LANES r.v[i_] = (uint32_t)(c.v[i_] >> LIMB_BITS);
    return r;
}
template <int N>
inline CoU32 co_bcast(CoU32 a) {   // every lane of a row <- lane N of the row
    CoU32 r;
    CO_LANES r.v[i_] = a.v[(i_ & ~15) + N];
    return r;
}
inline CoU32 co_from_above(CoU32 a) {   // lane j <- lane j + 1 of the row, lane 15 <- 0
    CoU32 r;
    CO_LANES r.v[i_] = (i_ & 15) == 15 ? 0u : a.v[i_ + 1];
    return r;
}
inline CoU32 co_from_below(CoU32 a) {   // lane j <- lane j - 1 of the row, lane 0 <- 0
    CoU32 r;
    CO_LANES r.v[i_] = (i_ & 15) == 0 ? 0u : a.v[i_ - 1];
    return r;
}
inline CoU32 co_gather(CoU32 src, CoU32 a) {   // lane i <- lane src[i]
    CoU32 r;
    CO_LANES r.v[i_] = a.v[src.v[i_] & 63];
    return r;
}
#undef CO_LANES
#else
// ---- the device: a "register" is a VGPR
typedef uint32_t CoU32;
typedef uint64_t CoU64;
ZK_DEV CoU32 co_splat(uint32_t c) { return c; }
ZK_DEV CoU32 co_lane() { return __lane_id(); }
ZK_DEV CoU32 co_eq(CoU32 a, uint32_t b) { return a == b; }
ZK_DEV CoU32 co_lt(CoU32 a, uint32_t b) { return a < b; }
ZK_DEV CoU32 co_sel(CoU32 c, CoU32 a, CoU32 b) { return c ? a : b; }
ZK_DEV CoU64 co_zero64() { return 0; }
ZK_DEV CoU64 co_mad(CoU32 a, CoU32 b, CoU64 c) { return (uint64_t)a * b + c; }
ZK_DEV CoU32 co_lo(CoU64 c) { return (uint32_t)c; }
ZK_DEV CoU64 co_shr30_add(CoU64 c, CoU32 a) { return (c >> LIMB_BITS) + a; }
ZK_DEV CoU32 co_hi30(CoU64 c) { return (uint32_t)(c >> LIMB_BITS); }
template <int CTRL>
ZK_DEV CoU32 co_dpp(CoU32 v) {   // out-of-row sources read as zero (bound_ctrl)
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, true);
}
template <int N>
ZK_DEV CoU32 co_bcast(CoU32 a) { return co_dpp<0x150 + N>(a); }   // row_newbcast:N
ZK_DEV CoU32 co_from_above(CoU32 a) { return co_dpp<0x101>(a); }  // row_shl:1
ZK_DEV CoU32 co_from_below(CoU32 a) { return co_dpp<0x111>(a); }  // row_shr:1
ZK_DEV CoU32 co_gather(CoU32 src, CoU32 a) { return (uint32_t)__builtin_amdgcn_ds_bpermute((int)(src << 2), (int)a); }
#endif

// A CoFe limb is at most 2^30 - 1 + CO_NEAR ("nearly normalised"): a parallel carry step adds at most 3 (its input is below 2^32) to a 30-bit low part, and a
// product's last accumulators are below 2^31 + 2^30 + 4 (see co_mul), so their high parts are at most 3 as well.
#define CO_NEAR 3u

ZK_DEV CoU32 co_limb_index() { return co_lane() & 15u; }
ZK_DEV CoU32 co_row_index() { return co_lane() >> 4; }
// limb (lane & 15) of a 9-limb constant, zero in lanes 9..15
ZK_DEV CoU32 co_limbs(const uint32_t (&c)[NLIMB]) {
    const CoU32 j = co_limb_index();
    CoU32 r = co_splat(0);
#pragma unroll
    for (int i = 0; i < NLIMB; i++) r = co_sel(co_eq(j, (uint32_t)i), co_splat(c[i]), r);
    return r;
}

// One element per row: limb (lane & 15) of a value < K * M; limbs <= 2^30 - 1 + CO_NEAR; lanes 9..15 zero.
template <class M, int K = 2>
struct CoFe {
    CoU32 v;
    template <int K2>
    ZK_DEV CoFe<M, K2> as() const {
        static_assert(K2 >= K, "cannot tighten a bound by cast");
        CoFe<M, K2> r;
        r.v = v;
        return r;
    }
};
// one parallel carry step: limbs < 2^32 in, limbs <= 2^30 + 2 out, same value (limb 8 takes what limb 7 hands up; it is never split)
ZK_DEV CoU32 co_carry(CoU32 x) {
    const CoU32 top = co_eq(co_limb_index(), NLIMB - 1);
    const CoU32 lo = co_sel(top, x, x & LIMB_MASK), up = co_sel(top, co_splat(0), x >> LIMB_BITS);
    return lo + co_from_below(up);
}
template <class M, int Ka, int Kb>
ZK_DEV CoFe<M, Ka + Kb> co_add(const CoFe<M, Ka>& a, const CoFe<M, Kb>& b) {
    static_assert(Ka + Kb <= KCAP, "magnitude overflow");
    CoFe<M, Ka + Kb> r;
    r.v = co_carry(a.v + b.v);
    return r;
}
// C * M in the redundant form of field.h's fe_sub2 (every limb below the top lends 2^30 twice): limbs 0..7 >= 2^31 - 2 >= any nearly normalised
// subtrahend limb, top limb >= the top limb of any value < (C - 1) * M; a + S < 2^32 for every constant of consts_gen.h (checked below).
template <class M, int C>
ZK_DEV CoU32 co_sub_const() {
    CoU32 s;
    if constexpr (C == 4) s = co_limbs(M::sub4);
    else if constexpr (C == 8) s = co_limbs(M::sub8);
    else if constexpr (C == 16) s = co_limbs(M::sub16);
    else if constexpr (C == 32) s = co_limbs(M::sub32);
    else if constexpr (C == 64) s = co_limbs(M::sub64);
    else if constexpr (C == 128) s = co_limbs(M::sub128);
    else s = co_limbs(M::sub256);
    const CoU32 j = co_limb_index();
    const CoU32 lend = co_sel(co_eq(j, 0), co_splat(1u << LIMB_BITS), co_sel(co_lt(j, NLIMB - 1), co_splat(LIMB_MASK), co_sel(co_eq(j, NLIMB - 1), co_splat(0xffffffffu), co_splat(0))));
    return s + lend;
}
template <class M>
constexpr bool co_sub_consts_fit() {
    const uint32_t* all[7] = {M::sub4, M::sub8, M::sub16, M::sub32, M::sub64, M::sub128, M::sub256};
    for (int c = 0; c < 7; c++)
        for (int i = 0; i < NLIMB - 1; i++)
            if ((uint64_t)all[c][i] + (i == 0 ? (1u << LIMB_BITS) : LIMB_MASK) + LIMB_MASK + CO_NEAR >= (1ull << 32)) return false;
    return true;
}
static_assert(co_sub_consts_fit<ModT>() && co_sub_consts_fit<ModQ>() && co_sub_consts_fit<ModN>(), "a + C*M overflows a limb");
template <class M, int Ka, int Kb>
ZK_DEV CoFe<M, Ka + SubC<Kb>::value> co_sub(const CoFe<M, Ka>& a, const CoFe<M, Kb>& b) {
    constexpr int C = SubC<Kb>::value;
    static_assert(Ka + C <= KCAP, "magnitude overflow");
    CoFe<M, Ka + C> r;
    r.v = co_carry(a.v + co_sub_const<M, C>() - b.v);
    return r;
}
// rows where `neg` is set: a - b, elsewhere a + b (the bound is that of the subtraction)
template <class M, int Ka, int Kb>
ZK_DEV CoFe<M, Ka + SubC<Kb>::value> co_addsub(const CoFe<M, Ka>& a, const CoFe<M, Kb>& b, CoU32 neg) {
    constexpr int C = SubC<Kb>::value;
    static_assert(Ka + C <= KCAP, "magnitude overflow");
    CoFe<M, Ka + C> r;
    r.v = co_carry(a.v + co_sel(neg, co_sub_const<M, C>() - b.v, b.v));
    return r;
}

// ---- Montgomery product, one per row
template <class M, int I>
ZK_DEV void co_mont_round(CoU64& t, const CoU32& a, const CoU32& b, const CoU32& mj) {
    t = co_mad(co_bcast<I>(a), b, t);
    const CoU32 m = co_bcast<0>((co_lo(t) * M::n0) & LIMB_MASK);
    t = co_mad(m, mj, t);
    t = co_shr30_add(t, co_from_above(co_lo(t) & LIMB_MASK));   // lane 0's low part is zero by the choice of m and falls off the row
    if constexpr (I + 1 < NLIMB) co_mont_round<M, I + 1>(t, a, b, mj);
}
// Column budget: t < 2^32 + 2^30 before a round (a shifted-down t < 2^62 plus 30 low bits); a_i * b_j <= (2^30 + 2)^2 and m * M_j < 2^60 are added, so
// t < 2^61 + 2^34 throughout and t >> 30 < 2^31 + 16.
template <class M, int Ka, int Kb>
ZK_DEV CoFe<M, 2> co_mul(const CoFe<M, Ka>& a, const CoFe<M, Kb>& b, const CoU32& mj /* co_limbs(M::mod) */) {
    static_assert((long)Ka * Kb <= M::kmax, "Montgomery input magnitudes too large");
    CoU64 t = co_zero64();
    co_mont_round<M, 0>(t, a.v, b.v, mj);
    CoFe<M, 2> r;
    r.v = (co_lo(t) & LIMB_MASK) + co_from_below(co_hi30(t));   // value < 2 M < 2^260: limb 8 < 2^20, nothing leaves lane 8
    return r;
}
// exact limbs < 2^30 (for stores into the one-lane layouts): nine parallel carry steps ripple any carry to the top
template <class M, int K>
ZK_DEV CoFe<M, K> co_normalize(const CoFe<M, K>& a) {
    CoFe<M, K> r = a;
#pragma unroll
    for (int i = 0; i < NLIMB; i++) r.v = co_carry(r.v);
    return r;
}

// a^e for a public exponent (9 little-endian 32-bit words), right to left like field.h's fe_pow_words; every row raises its own element
template <class M>
ZK_DEV CoFe<M, 2> co_pow_words(const CoFe<M, 2>& a, const uint32_t (&e)[NLIMB], const CoU32& mj) {
    CoFe<M, 2> acc, base = a;
    acc.v = co_limbs(M::one);
#pragma unroll 1
    for (int w = 0; w < NLIMB; w++) {
        const uint32_t ew = e[w];
        int nb = M::bits - 32 * w;
        if (nb > 32) nb = 32;
#pragma unroll 1
        for (int b = 0; b < nb; b++) {
            if ((ew >> b) & 1) acc = co_mul(acc, base, mj);
            base = co_mul(base, base, mj);
        }
    }
    return acc;
}

// ---- rows
// row r of the result <- row R_r of a (4: zero -- lane 15 of the own row is always zero)
template <int R0, int R1, int R2, int R3>
ZK_DEV CoU32 co_row_src() {
    const CoU32 lane = co_lane(), row = lane >> 4, j = lane & 15u;
    const CoU32 sr = co_sel(co_eq(row, 0), co_splat(R0), co_sel(co_eq(row, 1), co_splat(R1), co_sel(co_eq(row, 2), co_splat(R2), co_splat(R3))));
    return co_sel(co_eq(sr, 4), lane | 15u, (sr << 4) | j);
}
template <int R0, int R1, int R2, int R3, class M, int K>
ZK_DEV CoFe<M, K> co_rows(const CoFe<M, K>& a) {
    CoFe<M, K> r;
    r.v = co_gather(co_row_src<R0, R1, R2, R3>(), a.v);
    return r;
}

// ---- Tom-256 (a = 1 image), extended coordinates: rows X, Y, T, Z (curve.h: TomPt)
struct CoTom {
    CoFe<ModT, 2> v;
};
// edwards.ts:141-160 with a = 1 (curve.h: tom_dbl): two product passes of four rows
ZK_DEV CoTom co_tom_dbl(const CoTom& p, const CoU32& mj) {
    const auto a1 = co_add(co_rows<0, 1, 3, 0>(p.v), co_rows<4, 4, 4, 1>(p.v));                 // x, y, z, x + y
    const auto s1 = co_mul(a1, a1, mj);                                                          // A, B, Cz, S
    const auto v = co_addsub(co_rows<0, 0, 2, 3>(s1), co_rows<1, 1, 2, 4>(s1), co_eq(co_row_index(), 1));   // G = A + B, H = A - B, C = 2 Cz, S
    const auto a2 = co_sub(co_rows<3, 0, 3, 0>(v), co_rows<0, 4, 0, 2>(v));                      // E = S - G, G, E, F = G - C
    const auto b2 = co_sub(co_rows<0, 1, 1, 0>(v), co_rows<2, 4, 4, 4>(v));                      // F, H, H, G
    CoTom r;
    r.v = co_mul(a2, b2, mj);                                                                    // X3 = E F, Y3 = G H, T3 = E H, Z3 = F G
    return r;
}
// edwards.ts:161-183 with a = 1 (curve.h: tom_add): three product passes
ZK_DEV CoTom co_tom_add(const CoTom& p, const CoTom& q, const CoU32& mj) {
    const auto m1 = co_mul(p.v, q.v, mj);                                                        // A = x1 x2, B = y1 y2, C0 = t1 t2, D = z1 z2
    CoFe<ModT, 1> d1;
    d1.v = co_limbs(TOM_D1_M);
    const auto pa = co_add(co_rows<0, 4, 4, 4>(p.v), co_rows<1, 4, 4, 4>(p.v));                  // x1 + y1 in row 0
    const auto qa = co_add(co_rows<0, 4, 4, 4>(q.v), co_rows<1, 4, 4, 4>(q.v));
    // second pass: row 0 (x1 + y1)(x2 + y2), row 1 C0 * d'
    CoFe<ModT, 4> a2, b2;
    a2.v = co_sel(co_eq(co_row_index(), 1), co_rows<4, 2, 4, 4>(m1).v, pa.v);
    b2.v = co_sel(co_eq(co_row_index(), 1), d1.v, qa.v);
    const auto m2 = co_mul(a2, b2, mj);                                                          // P4, C, -, -
    // E = P4 - (A + B), F = D - C, G = D + C, H = B - A
    const auto ab = co_add(co_rows<0, 4, 4, 4>(m1), co_rows<1, 4, 4, 4>(m1));                    // A + B in row 0
    CoFe<ModT, 4> x1, y1;   // x1 +- y1 per row: E = P4 - (A+B); F = D - C; G = D + C; H = B - A
    x1.v = co_sel(co_eq(co_row_index(), 0), m2.v, co_sel(co_eq(co_row_index(), 3), co_rows<4, 4, 4, 1>(m1).v, co_rows<4, 3, 3, 4>(m1).v));
    y1.v = co_sel(co_eq(co_row_index(), 0), ab.v, co_sel(co_eq(co_row_index(), 3), co_rows<4, 4, 4, 0>(m1).v, co_rows<4, 1, 1, 4>(m2).v));
    const auto w = co_addsub(x1, y1, co_eq(co_row_index(), 2) ^ 1u);                             // rows E, F, G, H
    const auto a3 = co_rows<0, 2, 0, 1>(w), b3 = co_rows<1, 3, 3, 2>(w);
    CoTom r;
    r.v = co_mul(a3, b3, mj);                                                                    // E F, G H, E H, F G
    return r;
}
// rows of four one-lane elements (limb-major arrays of 9 words each) <-> a CoFe
template <class M, int K>
ZK_DEV CoFe<M, K> co_load4(const uint32_t* e0, const uint32_t* e1, const uint32_t* e2, const uint32_t* e3) {
    const CoU32 lane = co_lane();
    CoFe<M, K> r;
#ifdef ZK_HOST_BUILD
    for (int i = 0; i < 64; i++) {
        const uint32_t* e = (i >> 4) == 0 ? e0 : (i >> 4) == 1 ? e1 : (i >> 4) == 2 ? e2 : e3;
        r.v.v[i] = (i & 15) < NLIMB ? e[i & 15] : 0u;
    }
    (void)lane;
#else
    const uint32_t row = lane >> 4, j = lane & 15u;
    const uint32_t* e = row == 0 ? e0 : row == 1 ? e1 : row == 2 ? e2 : e3;
    r.v = j < NLIMB ? e[j] : 0u;
#endif
    return r;
}
template <class M, int K>
ZK_DEV void co_store4(const CoFe<M, K>& a, uint32_t* e0, uint32_t* e1, uint32_t* e2, uint32_t* e3) {
    const CoFe<M, K> n = co_normalize(a);
#ifdef ZK_HOST_BUILD
    for (int i = 0; i < 64; i++) {
        uint32_t* e = (i >> 4) == 0 ? e0 : (i >> 4) == 1 ? e1 : (i >> 4) == 2 ? e2 : e3;
        if ((i & 15) < NLIMB && e) e[i & 15] = n.v.v[i];
    }
#else
    const uint32_t lane = co_lane(), row = lane >> 4, j = lane & 15u;
    uint32_t* e = row == 0 ? e0 : row == 1 ? e1 : row == 2 ? e2 : e3;
    if (j < NLIMB && e) e[j] = n.v;
#endif
}

// ---- small helpers of the P-256 formulas
template <class M, int K>
ZK_DEV CoFe<M, 3 * K> co_triple(const CoFe<M, K>& a) {   // limbs < 3 * 2^30 + 9 < 2^32 before the carry step
    static_assert(3 * K <= KCAP, "magnitude overflow");
    CoFe<M, 3 * K> r;
    r.v = co_carry(a.v + a.v + a.v);
    return r;
}
template <class M, int K>
ZK_DEV CoFe<M, 2 * K> co_double(const CoFe<M, K>& a) {
    static_assert(2 * K <= KCAP, "magnitude overflow");
    CoFe<M, 2 * K> r;
    r.v = co_carry(a.v + a.v);
    return r;
}
// rows where c is set: a, elsewhere b (the looser bound)
template <class M, int Ka, int Kb>
ZK_DEV CoFe<M, (Ka > Kb ? Ka : Kb)> co_pick(CoU32 c, const CoFe<M, Ka>& a, const CoFe<M, Kb>& b) {
    CoFe<M, (Ka > Kb ? Ka : Kb)> r;
    r.v = co_sel(c, a.v, b.v);
    return r;
}
template <class M>
ZK_DEV CoFe<M, 1> co_const(const uint32_t (&c)[NLIMB]) {   // the constant in every row
    CoFe<M, 1> r;
    r.v = co_limbs(c);
    return r;
}
ZK_DEV CoU32 co_row_is(uint32_t r) { return co_eq(co_row_index(), r); }

// ---- P-256, homogeneous (X : Y : Z) in rows 0, 1, 2 (row 3 of a point register is ignored), the complete Renes-Costello-Batina laws of curve.h
// (weier.ts:133-230) with their 14 / 13 products dealt to FOUR passes of at most four rows.
struct CoP256 {
    CoFe<ModQ, 8> v;
};
// weier.ts:176-230 (curve.h: p256_add).  Pass 1: t0, t1, t2, (X1 + Z1)(X2 + Z2);  pass 2: (X1 + Y1)(X2 + Y2), (Y1 + Z1)(Y2 + Z2), b t2;
// pass 3: b y3, x3d z3b, t3b x3d, t4b z3b;  pass 4: t4b y3d, t0b y3d, t3b t0b.
ZK_DEV CoP256 co_p256_add(const CoP256& p, const CoP256& q, const CoU32& mj) {
    const auto bq = co_const<ModQ>(P256_B_M);
    const auto a1 = co_add(co_rows<0, 1, 2, 0>(p.v), co_rows<4, 4, 4, 2>(p.v));                  // X1, Y1, Z1, X1 + Z1
    const auto b1 = co_add(co_rows<0, 1, 2, 0>(q.v), co_rows<4, 4, 4, 2>(q.v));
    const auto m1 = co_mul(a1, b1, mj);                                                          // t0, t1, t2, x3
    const auto a2 = co_pick(co_row_is(2), bq, co_add(co_rows<0, 1, 4, 4>(p.v), co_rows<1, 2, 4, 4>(p.v)));   // X1 + Y1, Y1 + Z1, b, 0
    const auto b2 = co_pick(co_row_is(2), m1, co_add(co_rows<0, 1, 4, 4>(q.v), co_rows<1, 2, 4, 4>(q.v)));   // X2 + Y2, Y2 + Z2, t2, 0
    const auto m2 = co_mul(a2, b2, mj);                                                          // t3, t4, z3 = b t2, 0
    // d = t3 - t0 - t1, t4 - t1 - t2, x3 - t0 - t2  (t3b, t4b, y3)
    const auto d = co_sub(co_pick(co_row_is(2), co_rows<4, 4, 3, 4>(m1), m2), co_add(co_rows<0, 1, 0, 4>(m1), co_rows<1, 2, 2, 4>(m1)));
    const auto x3c = co_triple(co_sub(d, m2));                                                   // row 2: 3 (y3 - z3)
    const auto t1r = co_rows<4, 1, 1, 1>(m1);                                                    // t1 and x3c in rows 1..3
    const auto xcr = co_rows<4, 2, 2, 2>(x3c);
    const auto x3d = co_add(t1r, xcr);
    const auto z3b = co_sub(t1r, xcr);
    const auto a3 = co_pick(co_row_is(0), bq, co_pick(co_row_is(1), x3d, co_rows<4, 4, 0, 1>(d)));            // b, x3d, t3b, t4b
    const auto b3 = co_pick(co_row_is(0), co_rows<2, 4, 4, 4>(d), co_pick(co_row_is(2), x3d, z3b));           // y3, z3b, x3d, z3b
    const auto m3 = co_mul(a3, b3, mj);                                                          // y3b, m1 = x3d z3b, m2 = t3b x3d, m3 = t4b z3b
    // row 0: t2b = 3 t2, y3d = 3 (y3b - t2b - t0), t0b = 3 t0 - t2b
    const auto t2b = co_triple(co_rows<2, 4, 4, 4>(m1));
    const auto y3d = co_triple(co_sub(m3, co_add(t2b, m1)));
    const auto t0b = co_sub(co_triple(m1), t2b);
    const auto a4 = co_pick(co_row_is(1), co_rows<4, 0, 4, 4>(t0b), co_rows<1, 4, 0, 4>(d));     // t4b, t0b, t3b, 0
    const auto b4 = co_pick(co_row_is(2), co_rows<4, 4, 0, 4>(t0b), co_rows<0, 0, 4, 4>(y3d));   // y3d, y3d, t0b, 0
    const auto m4 = co_mul(a4, b4, mj);                                                          // t1b, t2c, m4, 0
    CoP256 r;
    r.v = co_addsub(co_rows<2, 1, 3, 4>(m3), m4, co_row_is(0)).template as<8>();                 // m2 - t1b, m1 + t2c, m3 + m4
    return r;
}
// weier.ts:133-175 (curve.h: p256_dbl).  Pass 1: Y^2, Z^2, XZ, YZ;  pass 2: X^2, XY, b t2, b (2 XZ);  pass 3: x3 y3c, x3 t3b, t0b z3d, t0d z3d;  pass 4: t0d t1.
ZK_DEV CoP256 co_p256_dbl(const CoP256& p, const CoU32& mj) {
    const auto bq = co_const<ModQ>(P256_B_M);
    const auto m1 = co_mul(co_rows<1, 2, 0, 1>(p.v), co_rows<1, 2, 2, 2>(p.v), mj);              // t1 = Y^2, t2 = Z^2, z3 = XZ, t0c = YZ
    const auto z3b = co_double(m1);                                                              // row 2: 2 XZ
    const auto a2 = co_pick(co_lt(co_row_index(), 2), co_rows<0, 0, 4, 4>(p.v), bq);             // X, X, b, b
    const auto b2 = co_pick(co_lt(co_row_index(), 2), co_rows<0, 1, 4, 4>(p.v), co_pick(co_row_is(2), co_rows<4, 4, 1, 4>(m1), co_rows<4, 4, 4, 2>(z3b)));   // X, Y, t2, z3b
    const auto m2 = co_mul(a2, b2, mj);                                                          // t0 = X^2, t3 = XY, bt2, bz3
    // row 0 works out y3b = 3 (bt2 - z3b), z3d = 3 (bz3 - 3 t2 - t0), t0b = 3 t0 - 3 t2
    const auto t2b = co_triple(co_rows<1, 4, 4, 4>(m1));                                         // row 0: 3 t2
    const auto y3b = co_triple(co_sub(co_rows<2, 4, 4, 4>(m2), co_rows<2, 4, 4, 4>(z3b)));
    const auto z3d = co_triple(co_sub(co_rows<3, 4, 4, 4>(m2), co_add(t2b, m2)));
    const auto t0b = co_sub(co_triple(m2), t2b);
    const auto t1 = co_rows<0, 4, 4, 4>(m1);                                                     // row 0: t1
    const auto x3 = co_sub(t1, y3b);
    const auto y3c = co_add(t1, y3b);
    const auto t0d = co_double(co_rows<3, 4, 4, 4>(m1));                                         // row 0: 2 YZ
    const auto t3b = co_double(co_rows<1, 4, 4, 4>(m2));                                         // row 0: 2 XY
    // pass 3 rows: x3 y3c, x3 t3b, t0b z3d, t0d z3d -- every operand sits in row 0 of its register
    const auto a3 = co_pick(co_lt(co_row_index(), 2), co_rows<0, 0, 4, 4>(x3), co_pick(co_row_is(2), co_rows<4, 4, 0, 4>(t0b), co_rows<4, 4, 4, 0>(t0d)));
    const auto b3 = co_pick(co_row_is(0), y3c, co_pick(co_row_is(1), co_rows<4, 0, 4, 4>(t3b), co_rows<4, 4, 0, 0>(z3d)));
    const auto m3 = co_mul(a3, b3, mj);                                                          // y3d, x3b, m1, m2
    const auto m4 = co_mul(t0d, t1, mj);                                                         // row 0: z3e = t0d t1
    // X3 = x3b - m2, Y3 = y3d + m1, Z3 = 4 z3e
    const auto z4 = co_double(co_double(co_rows<4, 4, 0, 4>(m4)));                               // row 2: 4 z3e
    const auto xy = co_addsub(co_rows<1, 0, 4, 4>(m3), co_rows<3, 2, 4, 4>(m3), co_row_is(0));   // rows 0, 1: x3b - m2, y3d + m1
    CoP256 r;
    r.v = co_pick(co_row_is(2), z4, xy).template as<8>();
    return r;
}
// Doubling CHAINS of points of odd order (the window bases 2^(bits w) R of a proof's table: k_rtab_base): Jacobian coordinates with Z^2 kept beside Z,
// rows X, Y, Z, ZZ (x = X / ZZ, y = Y / (Z ZZ)), "dbl-2008-s-1" shape for a = -3 -- nine products in THREE passes:
//   U = 2 Y;  pass 1: V = U^2, N = (X - ZZ)(X + ZZ), Z3 = U Z;  M = 3 N;  pass 2: W = U V, S = X V, M^2, ZZ3 = V ZZ;  X3 = M^2 - 2 S;
//   pass 3: M (S - X3), W Y;  Y3 = M (S - X3) - W Y.
// Valid for every point of P-256 (no point of order two); the identity (0 : 1 : 0 : 0) stays (0 : Y' : 0 : 0) with Y' != 0.  Bounds: X < 10 q, Y < 6 q, Z, ZZ < 2 q.
struct CoP256J {
    CoFe<ModQ, 10> v;
};
ZK_DEV CoP256J co_p256_jdbl(const CoP256J& p, const CoU32& mj) {
    // pass 1: (2Y)(2Y), (X - ZZ)(X + ZZ), (2Y) Z
    const auto a1 = co_addsub(co_rows<1, 0, 1, 4>(p.v), co_rows<1, 3, 1, 4>(p.v), co_row_is(1));  // U, X - ZZ, U, 0
    const auto b1 = co_add(co_rows<1, 0, 2, 4>(p.v), co_rows<1, 3, 4, 4>(p.v));                   // U, X + ZZ, Z, 0
    const auto m1 = co_mul(a1, b1, mj);                                                           // V, N, Z3, 0
    const auto mm = co_triple(co_rows<4, 4, 1, 4>(m1));                                           // row 2: M = 3 N
    // pass 2: U V, X V, M M, V ZZ
    const auto a2 = co_pick(co_row_is(0), a1, co_pick(co_row_is(1), co_rows<4, 0, 4, 4>(p.v), co_pick(co_row_is(2), mm, co_rows<4, 4, 4, 0>(m1))));   // U, X, M, V
    const auto b2 = co_pick(co_lt(co_row_index(), 2), co_rows<0, 0, 4, 4>(m1), co_pick(co_row_is(2), mm, p.v));                                       // V, V, M, ZZ
    const auto m2 = co_mul(a2, b2, mj);                                                           // W, S, M^2, ZZ3
    const auto s1 = co_rows<4, 4, 1, 4>(m2);                                                      // row 2: S
    const auto x3 = co_sub(m2, co_double(s1));                                                    // row 2: X3 = M^2 - 2 S
    // pass 3: M (S - X3) in row 2, W Y in row 1
    const auto a3 = co_pick(co_row_is(2), mm, co_rows<4, 0, 4, 4>(m2));                           // -, W, M, -
    const auto b3 = co_pick(co_row_is(2), co_sub(s1, x3), p.v);                                   // -, Y, S - X3, -
    const auto m3 = co_mul(a3, b3, mj);                                                           // row 1: W Y, row 2: M (S - X3)
    const auto y3 = co_sub(co_rows<4, 2, 4, 4>(m3), m3);                                          // row 1: Y3
    CoP256J r;
    r.v = co_pick(co_row_is(0), co_rows<2, 4, 4, 4>(x3), co_pick(co_row_is(1), y3, co_pick(co_row_is(2), m1, m2))).template as<10>();
    return r;
}
// (X : Y : Z : ZZ) -> homogeneous (X Z : Y : Z ZZ), the form the complete additions and the table entries take; the identity becomes (0 : Y : 0)
ZK_DEV CoP256 co_p256_from_jac(const CoP256J& p, const CoU32& mj) {
    const auto one = co_const<ModQ>(ModQ::one);
    const auto a = co_rows<0, 1, 2, 4>(p.v);                                                     // X, Y, Z, 0
    const auto b = co_pick(co_row_is(1), one, co_rows<2, 4, 3, 4>(p.v));                         // Z, 1, ZZ, 0
    CoP256 r;
    r.v = co_mul(a, b, mj).template as<8>();
    return r;
}

