// Point arithmetic on the two groups of the hot path, one point per lane (gfx950).
//   * P-256 (src/curves/weier.ts): projective (X:Y:Z), a = -3, Renes-Costello-Batina COMPLETE addition and
//     doubling -- the same step sequences the reference uses (weier.ts:133-230), so identity / P = Q inputs behave
//     exactly as in the reference.
//   * Tom-256 (src/curves/edwards.ts): twisted Edwards a*x^2 + y^2 = 1 + d*x^2*y^2 in extended coordinates.  The
//     engine works on the isomorphic curve x'^2 + y^2 = 1 + (d/a) x'^2 y^2 (x' = sqrt(a) x), whose unified
//     Hisil-Wong-Carter-Dawson addition (edwards.ts:161-183 with a = 1) needs no multiplication by a.  a is a
//     square and d/a a non-square mod t (checked in tools/gen_consts.py), so the law is complete.  Only affine
//     coordinates of the original curve ever leave the engine (hash inputs, proof bytes).
#pragma once
#include "field.h"

// Independent products of a point formula in lock-step (field.h: limbs_mont_mul_n) or one after the other; chosen per
// translation unit and per curve with -DZK_BATCH_TOM=0/1, -DZK_BATCH_P256=0/1 (csrc/Makefile).
#ifndef ZK_BATCH_TOM
#define ZK_BATCH_TOM 1
#endif
#ifndef ZK_BATCH_P256
#define ZK_BATCH_P256 1
#endif

typedef Fe<ModQ, 8> Fq8;  // P-256 coordinate: < 8q
typedef Fe<ModQ, 2> Fq2;
typedef Fe<ModT, 2> Ft2;  // Tom coordinate: < 2t
typedef Fe<ModN, 2> Fn2;

struct P256Pt {
    Fq8 x, y, z;
};
struct P256Aff {  // Montgomery affine, canonical not required
    Fq2 x, y;
};
ZK_DEV P256Pt p256_identity() {  // weier.ts:50-52
    P256Pt r;
    r.x = fe_zero<ModQ>().as<8>();
    r.y = fe_one_mont<ModQ>().as<8>();
    r.z = fe_zero<ModQ>().as<8>();
    return r;
}
ZK_DEV P256Pt p256_from_affine(const P256Aff& a) {
    P256Pt r;
    r.x = a.x.as<8>();
    r.y = a.y.as<8>();
    r.z = fe_one_mont<ModQ>().as<8>();
    return r;
}
// weier.ts:176-230 (RCB 2016, Algorithm 4, a = -3): 12M + 2 mult-by-b
ZK_DEV P256Pt p256_add(const P256Pt& p, const P256Pt& q) {
    const auto b = fe_const<ModQ, 1>(P256_B_M);
    Fq2 t0, t1, t2, t3, t4, x3;
    fe_mul3<ZK_BATCH_P256 != 0>(t0, t1, t2, p.x, q.x, p.y, q.y, p.z, q.z);
    fe_mul3<ZK_BATCH_P256 != 0>(t3, t4, x3, p.x + p.y, q.x + q.y, p.y + p.z, q.y + q.z, p.x + p.z, q.x + q.z);
    auto t3b = fe_sub2(t3, t0, t1);
    auto t4b = fe_sub2(t4, t1, t2);
    auto y3 = fe_sub2(x3, t0, t2);
    Fq2 z3, y3b;
    fe_mul2<ZK_BATCH_P256 != 0>(z3, y3b, b, t2, b, y3);
    auto x3b = y3 - z3;
    auto x3c = x3b + (x3b + x3b);
    auto z3b = t1 - x3c;
    auto x3d = t1 + x3c;
    auto t2b = t2 + t2 + t2;
    auto y3c = fe_sub2(y3b, t2b, t0);
    auto y3d = y3c + (y3c + y3c);
    auto t0b = (t0 + t0 + t0) - t2b;
    Fq2 t1b, t2c, m1, m2, m3, m4;
    fe_mul3<ZK_BATCH_P256 != 0>(t1b, t2c, m1, t4b, y3d, t0b, y3d, x3d, z3b);
    fe_mul3<ZK_BATCH_P256 != 0>(m2, m3, m4, t3b, x3d, t4b, z3b, t3b, t0b);
    auto y3e = m1 + t2c;
    auto x3e = m2 - t1b;
    auto z3c = m3 + m4;
    P256Pt r;
    r.x = x3e.template as<8>();
    r.y = y3e.template as<8>();
    r.z = z3c.template as<8>();
    return r;
}
// Same law with q affine (Z2 = 1): RCB Algorithm 5 shape, 11M + 2 mult-by-b.  q must not be the identity.
ZK_DEV P256Pt p256_add_mixed(const P256Pt& p, const P256Aff& q) {
    const auto b = fe_const<ModQ, 1>(P256_B_M);
    Fq2 t0, t1, t3, m4, m5, z3;
    fe_mul3<ZK_BATCH_P256 != 0>(t0, t1, t3, p.x, q.x, p.y, q.y, p.x + p.y, q.x + q.y);
    fe_mul3<ZK_BATCH_P256 != 0>(m4, m5, z3, q.y, p.z, q.x, p.z, b, p.z);
    auto t3b = fe_sub2(t3, t0, t1);
    auto t4b = m4 + p.y;   // (y1+z1)(y2+1) - t1 - z1
    auto y3 = m5 + p.x;    // (x1+z1)(x2+1) - t0 - z1
    auto x3b = y3 - z3;
    auto x3c = x3b + (x3b + x3b);
    auto z3b = t1 - x3c;
    auto x3d = t1 + x3c;
    auto y3b = b * y3;
    auto t2b = p.z + p.z + p.z;
    auto y3c = fe_sub2(y3b, t2b, t0);
    auto y3d = y3c + (y3c + y3c);
    auto t0b = (t0 + t0 + t0) - t2b;
    Fq2 t1b, t2c, m1, m2, m3, m6;
    fe_mul3<ZK_BATCH_P256 != 0>(t1b, t2c, m1, t4b, y3d, t0b, y3d, x3d, z3b);
    fe_mul3<ZK_BATCH_P256 != 0>(m2, m3, m6, t3b, x3d, t4b, z3b, t3b, t0b);
    auto y3e = m1 + t2c;
    auto x3e = m2 - t1b;
    auto z3c = m3 + m6;
    P256Pt r;
    r.x = x3e.template as<8>();
    r.y = y3e.template as<8>();
    r.z = z3c.template as<8>();
    return r;
}
// weier.ts:133-175 (RCB Algorithm 6, a = -3): 8M + 3S incl. 2 mult-by-b
ZK_DEV P256Pt p256_dbl(const P256Pt& p) {
    const auto b = fe_const<ModQ, 1>(P256_B_M);
    Fq2 t0, t1, t2, t3, z3, t0c;
    fe_mul3<ZK_BATCH_P256 != 0>(t0, t1, t2, p.x, p.x, p.y, p.y, p.z, p.z);
    fe_mul3<ZK_BATCH_P256 != 0>(t3, z3, t0c, p.x, p.y, p.x, p.z, p.y, p.z);
    auto t3b = t3 + t3;
    auto z3b = z3 + z3;
    Fq2 bt2, bz3;
    fe_mul2<ZK_BATCH_P256 != 0>(bt2, bz3, b, t2, b, z3b);
    auto y3 = bt2 - z3b;
    auto y3b = y3 + (y3 + y3);
    auto x3 = t1 - y3b;
    auto y3c = t1 + y3b;
    auto t2b = t2 + t2 + t2;
    auto z3c = fe_sub2(bz3, t2b, t0);
    auto z3d = z3c + (z3c + z3c);
    auto t0b = (t0 + t0 + t0) - t2b;
    auto t0d = t0c + t0c;
    Fq2 y3d, x3b, m1, m2, z3e;
    fe_mul3<ZK_BATCH_P256 != 0>(y3d, x3b, m1, x3, y3c, x3, t3b, t0b, z3d);
    fe_mul2<ZK_BATCH_P256 != 0>(m2, z3e, t0d, z3d, t0d, t1);
    auto y3e = y3d + m1;
    auto x3c = x3b - m2;
    auto z3f = (z3e + z3e) + (z3e + z3e);
    P256Pt r;
    r.x = x3c.template as<8>();
    r.y = y3e.template as<8>();
    r.z = z3f.template as<8>();
    return r;
}
// Jacobian doubling for a = -3 (x = X / Z^2, y = Y / Z^3; "dbl-2001-b": 3M + 5S against the 13 products of the complete doubling above).  Only for the
// doubling CHAINS of the per-proof window tables of R (k_rtab_base: 256 doublings in a row, what a single verification waits for longest): their points
// have odd prime order, which is all the formula needs (it is wrong for Y = 0, a point of order 2; the identity (X : Y : 0) stays (0 : Y' : 0), Y' != 0).
// Magnitudes: the lazy subtractions at the end leave X, Y < 34 q and Z < 10 q; as inputs of the next doubling their products stay far inside
// ModQ::kmax ((34 + 10)^2, (34 + 4)(34 + 2)), so the chain needs no reduction of its own.
struct P256Jac {
    Fe<ModQ, 34> x, y;
    Fe<ModQ, 10> z;
};
ZK_DEV P256Jac p256_jac_from(const P256Pt& p) {   // a point with Z = 1 (p256_from_affine) or the identity (0 : 1 : 0): the same triple in both systems
    P256Jac r;
    r.x = p.x.as<34>(), r.y = p.y.as<34>(), r.z = p.z.as<10>();
    return r;
}
ZK_DEV P256Jac p256_jdbl(const P256Jac& p) {
    Fq2 delta, gamma, yz2;
    fe_mul3<ZK_BATCH_P256 != 0>(delta, gamma, yz2, p.z, p.z, p.y, p.y, p.y + p.z, p.y + p.z);
    Fq2 beta, a0, g2;
    fe_mul3<ZK_BATCH_P256 != 0>(beta, a0, g2, p.x, gamma, p.x - delta, p.x + delta, gamma, gamma);
    auto alpha = a0 + (a0 + a0);
    auto b4 = (beta + beta) + (beta + beta);
    auto g4 = (g2 + g2) + (g2 + g2);
    Fq2 al2 = alpha * alpha;
    auto x3 = fe_sub2(al2, b4, b4);
    Fq2 m = alpha * (b4 - x3);
    auto y3 = fe_sub2(m, g4, g4);
    auto z3 = fe_sub2(yz2, gamma, delta);
    P256Jac r;
    r.x = x3, r.y = y3, r.z = z3;
    return r;
}
ZK_DEV P256Pt p256_from_jac(const P256Jac& p) {   // (X : Y : Z) Jacobian -> (X Z : Y : Z^3) homogeneous, the form the complete additions take
    Fq2 z2 = p.z * p.z;
    Fq2 xz, z3;
    fe_mul2<ZK_BATCH_P256 != 0>(xz, z3, p.x, p.z, z2, p.z);
    Fq2 y = fe_reduce(p.y);   // the additions take coordinates < 8 q
    P256Pt r;
    r.x = xz.template as<8>();
    r.y = y.as<8>();
    r.z = z3.template as<8>();
    return r;
}
ZK_DEV P256Pt p256_select(bool c, const P256Pt& a, const P256Pt& b) {
    P256Pt r;
    r.x = fe_select(c, a.x, b.x);
    r.y = fe_select(c, a.y, b.y);
    r.z = fe_select(c, a.z, b.z);
    return r;
}
// A table sum skips the addition of a ZERO digit.  Default build: with a branch -- the lane idles while its neighbours add, which is what keeps the table
// sums at 124-140 registers and three waves per SIMD, but makes control flow depend on digits of secret scalars (nonces, blinders).  -DZK_UNIFORM_CF=1
// (`make uniform`: lib/libzkattest_hip_uniform.so): the addition is always computed, on a valid entry, and discarded by a select; control flow and the
// set of executed instructions no longer depend on secret data (addresses of the table gathers still do: see INTEGRATION.md, "side channels").
#ifndef ZK_UNIFORM_CF
#define ZK_UNIFORM_CF 0
#endif
#if ZK_UNIFORM_CF
#define ZK_ADD_IF(cond, acc, sum_expr)                 \
    do {                                               \
        const bool c__ = (cond);                       \
        const P256Pt s__ = (sum_expr);                 \
        (acc) = p256_select(c__, s__, (acc));          \
    } while (0)
#else
#define ZK_ADD_IF(cond, acc, sum_expr)                 \
    do {                                               \
        if (cond) (acc) = (sum_expr);                  \
    } while (0)
#endif

// y^2 == x^3 - 3x + b  (weier.ts:56-70 with Z = 1)
ZK_DEV bool p256_on_curve(const P256Aff& a) {
    const auto b = fe_const<ModQ, 1>(P256_B_M);
    auto y2 = a.y * a.y;
    auto x2 = a.x * a.x;
    auto x3 = x2 * a.x;
    auto rhs = (x3 + b) - (a.x + a.x + a.x);
    return fe_eq(y2, rhs);
}

// ---------------------------------------------------------------- Tom-256 on the a = 1 image
struct TomPt {  // extended (X:Y:T:Z), Montgomery
    Ft2 x, y, t, z;
};
template <int KX>
struct TomNielsT {  // affine precomputed: x, y, (d/a)*x*y; x and dt may carry a looser bound (negated entries)
    Fe<ModT, KX> x;
    Ft2 y;
    Fe<ModT, KX> dt;
};
typedef TomNielsT<2> TomNiels;
// -P of a table entry when neg (signed comb digits): (-x, y, -dt) with -v = 4t - v < 4t
ZK_DEV TomNielsT<4> tom_niels_neg_sel(const TomNiels& q, bool neg) {
    TomNielsT<4> r;
#pragma unroll
    for (int i = 0; i < NLIMB; i++) {
        r.x.l[i] = neg ? ModT::sub4[i] - q.x.l[i] : q.x.l[i];
        r.dt.l[i] = neg ? ModT::sub4[i] - q.dt.l[i] : q.dt.l[i];
    }
    limbs_normalize(r.x.l);
    limbs_normalize(r.dt.l);
    r.y = q.y;
    return r;
}
ZK_DEV TomPt tom_identity() {  // edwards.ts:46-48
    TomPt r;
    r.x = fe_zero<ModT>().as<2>();
    r.y = fe_one_mont<ModT>().as<2>();
    r.t = fe_zero<ModT>().as<2>();
    r.z = fe_one_mont<ModT>().as<2>();
    return r;
}
// edwards.ts:161-183 with a = 1 and Z2 = 1, d*T2 precomputed: 8M, as two lock-step groups of four independent products
template <int KX>
ZK_DEV TomPt tom_add_niels(const TomPt& p, const TomNielsT<KX>& q) {
    Ft2 A, B, C, P4;
    fe_mul4<ZK_BATCH_TOM != 0>(A, B, C, P4, p.x, q.x, p.y, q.y, p.t, q.dt, p.x + p.y, q.x + q.y);
    auto E = fe_sub2(P4, A, B);
    auto F = p.z - C;
    auto G = p.z + C;
    auto H = B - A;
    TomPt r;
    fe_mul4<ZK_BATCH_TOM != 0>(r.x, r.y, r.t, r.z, E, F, G, H, E, H, F, G);
    return r;
}
// the same addition when the result only has to be ADDED TO NOTHING ELSE (last step of a comb): no T3, 7M
template <int KX>
ZK_DEV TomPt tom_add_niels_last(const TomPt& p, const TomNielsT<KX>& q) {
    Ft2 A, B, C, P4;
    fe_mul4<ZK_BATCH_TOM != 0>(A, B, C, P4, p.x, q.x, p.y, q.y, p.t, q.dt, p.x + p.y, q.x + q.y);
    auto E = fe_sub2(P4, A, B);
    auto F = p.z - C;
    auto G = p.z + C;
    auto H = B - A;
    TomPt r;
    fe_mul3<ZK_BATCH_TOM != 0>(r.x, r.y, r.z, E, F, G, H, F, G);
    r.t = fe_zero<ModT>().as<2>();
    return r;
}
// identity + q: the niels entry as an extended point (X, Y, T = XY, Z = 1), 1M instead of the 8M of an addition
template <int KX>
ZK_DEV TomPt tom_from_niels(const TomNielsT<KX>& q) {
    TomPt r;
    if constexpr (KX == 2) r.x = q.x;
    else r.x = fe_reduce(q.x);  // a (possibly negated) entry with the looser bound: one extra product
    r.y = q.y;
    r.t = q.x * q.y;
    r.z = fe_one_mont<ModT>().as<2>();
    return r;
}
// general unified addition (both extended): 9M + 1 mult-by-d'
ZK_DEV TomPt tom_add(const TomPt& p, const TomPt& q) {
    const auto d1 = fe_const<ModT, 1>(TOM_D1_M);
    Ft2 A, B, C0, D, P4;
    fe_mul4<ZK_BATCH_TOM != 0>(A, B, C0, D, p.x, q.x, p.y, q.y, p.t, q.t, p.z, q.z);
    Ft2 C;
    fe_mul2<ZK_BATCH_TOM != 0>(C, P4, C0, d1, p.x + p.y, q.x + q.y);
    auto E = fe_sub2(P4, A, B);
    auto F = D - C;
    auto G = D + C;
    auto H = B - A;
    TomPt r;
    fe_mul4<ZK_BATCH_TOM != 0>(r.x, r.y, r.t, r.z, E, F, G, H, E, H, F, G);
    return r;
}
// edwards.ts:141-160 with a = 1: 4S + 4M
ZK_DEV TomPt tom_dbl(const TomPt& p) {
    Ft2 A, B, Cz, S;
    auto xy = p.x + p.y;
    fe_mul4<ZK_BATCH_TOM != 0>(A, B, Cz, S, p.x, p.x, p.y, p.y, p.z, p.z, xy, xy);
    auto C = Cz + Cz;
    auto E = fe_sub2(S, A, B);
    auto G = A + B;
    auto F = G - C;
    auto H = A - B;
    TomPt r;
    fe_mul4<ZK_BATCH_TOM != 0>(r.x, r.y, r.t, r.z, E, F, G, H, E, H, F, G);
    return r;
}
ZK_DEV TomPt tom_neg(const TomPt& p) {  // edwards.ts:136-140
    TomPt r;
    r.x = fe_reduce(fe_neg(p.x));
    r.y = p.y;
    r.t = fe_reduce(fe_neg(p.t));
    r.z = p.z;
    return r;
}
// Is (x, y) -- plain 9-word coordinates -- a point of a*x^2 + y^2 = 1 + d*x^2*y^2 with both coordinates < t (edwards.ts:52-65, 74-77: what
// deserialising a point checks)?  Validation only needs the verdict, so the coordinates are NOT taken to the Montgomery domain: every Montgomery
// product of plain operands divides by R once, u = x^2/R, v = y^2/R, w = u*v/R = x^2 y^2/R^3, and the equation is compared scaled by 1/R:
// u * (aR)/R + v  ==  w * (dR^3)/R + 1/R.  Five products instead of the seven of tom_from_affine_words (the verifier's validation pass
// runs it on ~1 600 points per proof).
ZK_DEV bool tom_words_on_curve(const uint32_t xw[9], const uint32_t yw[9]) {
    if (words_geq<9>(xw, ModT::mod32) || words_geq<9>(yw, ModT::mod32)) return false;
    const auto x = fe_from_words<ModT, 9>(xw), y = fe_from_words<ModT, 9>(yw);
    const auto u = x * x, v = y * y;
    const auto lhs = u * fe_const<ModT, 1>(TOM_A_M) + v;
    const auto rhs = (u * v) * fe_const<ModT, 1>(TOM_D_R3) + fe_const<ModT, 1>(TOM_RINV);
    return fe_eq(lhs, rhs);
}
// original-curve affine (plain 9-word x, y) -> a=1 image, extended Montgomery.  Returns false if a coordinate is
// >= t (edwards.ts:74-77) or the point is off the curve a*x^2 + y^2 = 1 + d*x^2*y^2 (edwards.ts:52-65).
ZK_DEV bool tom_from_affine_words(TomPt& r, const uint32_t xw[9], const uint32_t yw[9]) {
    if (words_geq<9>(xw, ModT::mod32) || words_geq<9>(yw, ModT::mod32)) return false;
    auto x = fe_to_mont(fe_from_words<ModT, 9>(xw));
    auto y = fe_to_mont(fe_from_words<ModT, 9>(yw));
    auto x2 = x * x, y2 = y * y;
    auto lhs = fe_const<ModT, 1>(TOM_A_M) * x2 + y2;
    auto rhs = fe_const<ModT, 1>(TOM_D_M) * (x2 * y2) + fe_one_mont<ModT>();
    bool ok = fe_eq(lhs, rhs);
    r.x = x * fe_const<ModT, 1>(TOM_S_M);
    r.y = y;
    r.t = r.x * r.y;
    r.z = fe_one_mont<ModT>().as<2>();
    return ok;
}
