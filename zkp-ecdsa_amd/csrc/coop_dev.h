// Device-only companions of coop.h: rows <-> the engine's one-lane memory layouts (AoS point entries, engine.h's Soa arrays), identities, and the
// table-entry addition of the verifier's Straus sums.  Included by the translation units that hold cooperative kernels (k_coop.hip, k_msm.hip, k_pmsm.hip).
#pragma once
#include "engine.h"
#include "coop.h"

// ---- device only: points between the one-lane layouts in memory and rows
// NROWS consecutive 9-limb elements at e (the AoS point entries: Tom-256 X, Y, T / d'T, Z -- k_msm.hip, VTerms::tab; P-256 X, Y, Z -- rtab.h): row r <- e[9 r ..]
template <class M, int K, int NROWS>
ZK_DEV CoFe<M, K> co_load_aos(const uint32_t* __restrict__ e) {
    const uint32_t lane = __lane_id(), row = lane >> 4, j = lane & 15u;
    CoFe<M, K> r;
    r.v = (j < NLIMB && row < (uint32_t)NROWS) ? e[row * NLIMB + j] : 0u;
    return r;
}
template <class M, int K, int NROWS>
ZK_DEV void co_store_aos(uint32_t* __restrict__ e, const CoFe<M, K>& a) {
    const CoFe<M, K> n = co_normalize(a);
    const uint32_t lane = __lane_id(), row = lane >> 4, j = lane & 15u;
    if (j < NLIMB && row < (uint32_t)NROWS) e[row * NLIMB + j] = n.v;
}
// row r -> limb-major array a_r at element e (engine.h: Soa); an array with a null pointer skips the row.  (By value: pointers to kernel arguments would
// park the structs in scratch.)
template <class M, int K>
ZK_DEV void co_store_soa(const CoFe<M, K>& a, uint32_t e, Soa a0, Soa a1, Soa a2, Soa a3) {
    const CoFe<M, K> n = co_normalize(a);
    const uint32_t lane = __lane_id(), row = lane >> 4, j = lane & 15u;
    uint32_t* p = row == 0 ? a0.p : row == 1 ? a1.p : row == 2 ? a2.p : a3.p;
    const uint32_t stride = row == 0 ? a0.stride : row == 1 ? a1.stride : row == 2 ? a2.stride : a3.stride;
    if (j < NLIMB && p) p[(size_t)j * stride + e] = n.v;
}
template <class M, int K>
ZK_DEV CoFe<M, K> co_load_soa(uint32_t e, Soa a0, Soa a1, Soa a2, Soa a3) {
    const uint32_t lane = __lane_id(), row = lane >> 4, j = lane & 15u;
    const uint32_t* p = row == 0 ? a0.p : row == 1 ? a1.p : row == 2 ? a2.p : a3.p;
    const uint32_t stride = row == 0 ? a0.stride : row == 1 ? a1.stride : row == 2 ? a2.stride : a3.stride;
    CoFe<M, K> r;
    r.v = (j < NLIMB && p) ? p[(size_t)j * stride + e] : 0u;
    return r;
}
// Affine table entries as points (X, Y, 1) in rows 0..2.  P-256 fixed-base comb (engine.h: PFIX_ENTRY_WORDS = 20: nine Montgomery limbs of x, nine of y):
ZK_DEV CoFe<ModQ, 8> co_load_pfix(const uint32_t* __restrict__ e) {
    const uint32_t lane = __lane_id(), row = lane >> 4, j = lane & 15u;
    CoFe<ModQ, 8> r;
    r.v = j >= NLIMB ? 0u : row < 2 ? e[row * NLIMB + j] : row == 2 ? co_limbs(ModQ::one) : 0u;
    return r;
}
// ... and a ring key's table (ktab.h: 8 + 8 canonical 32-bit words of x, y in Montgomery form): lane j cuts limb j out of the words; neg: Y <- 4 q - Y (ld_ktab takes
// (x, q - y): the same residue, so the same point and the same bytes later)
ZK_DEV CoFe<ModQ, 8> co_load_ktab(const uint32_t* __restrict__ e, bool neg) {
    const uint32_t lane = __lane_id(), row = lane >> 4, j = lane & 15u;
    const uint32_t bit = LIMB_BITS * (j < NLIMB ? j : 0), k = bit >> 5, sh = bit & 31u;
    uint32_t v = 0;
    if (row < 2 && j < NLIMB) {
        const uint32_t* w = e + 8 * row;
        const uint32_t lo = w[k], hi = k < 7 ? w[k + 1] : 0u;
        v = (sh ? (lo >> sh) | (hi << (32 - sh)) : lo) & ((1u << LIMB_BITS) - 1);
    }
    CoFe<ModQ, 8> r;
    r.v = row == 2 && j < NLIMB ? co_limbs(ModQ::one) : v;
    if (neg && row == 1) r.v = co_carry(co_sub_const<ModQ, 4>() - r.v);
    return r;
}
ZK_DEV CoTom co_tom_identity() {   // (0 : 1 : 0 : 1), rows X, Y, T, Z
    CoTom r;
    r.v.v = (co_row_index() & 1u) ? co_limbs(ModT::one) : 0u;
    return r;
}
ZK_DEV CoP256 co_p256_identity() {   // (0 : 1 : 0)
    CoP256 r;
    r.v.v = co_row_index() == 1 ? co_limbs(ModQ::one) : 0u;
    return r;
}
// addition of a Straus table entry (X2, Y2, d'T2, Z2 -- k_verify.hip: st_tab), negated when `neg` (-X2, -d'T2): curve.h's tom_add_tab, nine products in three passes
ZK_DEV CoTom co_tom_add_tab(const CoTom& p, const CoFe<ModT, 2>& ent, bool neg, const CoU32& mj) {
    CoFe<ModT, 0> zero;
    zero.v = 0;
    const auto e = co_pick((uint32_t)(neg && !(co_row_index() & 1u)), co_sub(zero, ent), ent);  // rows 0 and 2 negated
    const auto m1 = co_mul(p.v, e, mj);                                                          // A, B, C, D
    const auto m2 = co_mul(co_add(p.v, co_rows<1, 4, 4, 4>(p.v)), co_add(e, co_rows<1, 4, 4, 4>(e)), mj);   // row 0: (x1 + y1)(x2 + y2)
    // E = P4 - (A + B), F = D - C, G = D + C, H = B - A
    const auto x1 = co_pick(co_row_is(0), m2, co_rows<4, 3, 3, 1>(m1));
    const auto y1 = co_pick(co_row_is(0), co_add(m1, co_rows<1, 4, 4, 4>(m1)), co_rows<4, 2, 2, 0>(m1));
    const auto w = co_addsub(x1, y1, (uint32_t)(co_row_index() != 2));
    CoTom r;
    r.v = co_mul(co_rows<0, 2, 0, 1>(w), co_rows<1, 3, 3, 2>(w), mj);                            // E F, G H, E H, F G
    return r;
}
