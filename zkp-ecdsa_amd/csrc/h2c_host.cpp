// Host-only translation unit of libzkattest_hip.so: the nothing-up-my-sleeve second generators of the hardened mode
// (include/zkattest.h: zk_hardened_h).  generatePedersenParams derives h = g * r from a scalar r somebody knows
// (src/commit/pedersen.ts:61-69, with the TODO "we must generate h without using scalar mult"): whoever generated the
// parameters can open a commitment to any value.  Here h comes out of SHA-256 by try-and-increment, so nobody knows log_g h.
//
// Compiled with g++ -DZK_HOST_BUILD: the SAME field / curve / SHA-256 templates the kernels use (field.h, curve.h, sha256.h),
// instantiated for the host -- a one-time, per-deployment computation needs no GPU and no context.
//
//   dom = "ZKAttest-NUMS-h-v1"
//   P-256    : ctr = 0, 1, ...: x = SHA-256(dom || 01 || tag || be32(ctr)) mod p;  r = x^3 - 3x + b;  y = r^((p+1)/4);
//              accept when y^2 = r; y := the even one of {y, p - y};  h = (x, y)                       (cofactor 1)
//   Tom-256  : ctr = 0, 1, ...: x = (SHA-256(dom || 02 || tag || be32(ctr) || 00) * 2^256 + SHA-256(.. || 01)) mod t;
//              u = (1 - a x^2) / (1 - d x^2);  y = u^((t+1)/4);  accept when y^2 = u; y := the even one of {y, t - y};
//              h = 4 * (x, y) (cofactor), accept unless h is the identity
// The test suite holds a big-integer restatement of this procedure and compares the two (tests/test_hardened.py).
#define ZK_HOST_BUILD 1
#include <cstring>
#include <vector>
#include "../../include/zkattest.h"
#include "curve.h"
#include "sha256.h"

namespace {
void sha256_bytes(const std::vector<uint8_t>& m, uint32_t out_words_be[8]) {
    uint32_t buf[16];
    ShaStream s;
    s.init(buf, 0, 1);
    for (uint8_t b : m) s.put_byte(b);
    s.finish(out_words_be);
}
std::vector<uint8_t> h2c_msg(uint8_t curve, const uint8_t* tag, uint64_t tag_len, uint32_t ctr, int half) {
    static const char dom[] = "ZKAttest-NUMS-h-v1";
    std::vector<uint8_t> m(dom, dom + sizeof dom - 1);
    m.push_back(curve);
    m.insert(m.end(), tag, tag + tag_len);
    for (int i = 3; i >= 0; i--) m.push_back((uint8_t)(ctr >> (8 * i)));
    if (half >= 0) m.push_back((uint8_t)half);
    return m;
}
// digest (8 big-endian words, most significant first) -> 8 little-endian words
void le_words(const uint32_t d[8], uint32_t w[8]) {
    for (int i = 0; i < 8; i++) w[i] = d[7 - i];
}
template <class M, int NW>
void store_be_bytes(uint8_t* out, int nbytes, const Fe<M, 1>& v) {
    uint32_t w[NW];
    words_from_limbs<NW>(w, v.l);
    for (int i = 0; i < nbytes; i++) {
        int bi = nbytes - 1 - i;
        out[i] = bi / 4 < NW ? (uint8_t)(w[bi / 4] >> (8 * (bi % 4))) : 0;
    }
}
template <class M>
Fe<M, 1> negate_if_odd(const Fe<M, 1>& y) {
    if (!(y.l[0] & 1)) return y;
    Fe<M, 1> m;
    for (int i = 0; i < NLIMB; i++) m.l[i] = M::mod[i];
    return fe_sub_mod(m, y);
}
}  // namespace

extern "C" __attribute__((visibility("default"))) zk_status zk_hardened_h(const uint8_t* tag, uint64_t tag_len, uint8_t nist_h[64], uint8_t tom_h[72]) {
    if ((!tag && tag_len) || tag_len > 4096 || !nist_h || !tom_h) return ZK_E_ARG;
    // ---- P-256
    for (uint32_t ctr = 0;; ctr++) {
        uint32_t d[8], w[8];
        sha256_bytes(h2c_msg(1, tag, tag_len, ctr, -1), d);
        le_words(d, w);
        Fe<ModQ, 1> xp = fe_from_words256_reduce<ModQ>(w);
        Fq2 x = fe_to_mont(xp);
        Fq2 r = fe_reduce((x * x * x + fe_const<ModQ, 1>(P256_B_M)) - (x + x + x));
        Fq2 y = fe_pow_words<ModQ>(r, ModQ::exp_sqrt);
        if (!fe_eq(y * y, r)) continue;
        Fe<ModQ, 1> yp = negate_if_odd(fe_from_mont(y));
        store_be_bytes<ModQ, 8>(nist_h, 32, xp);
        store_be_bytes<ModQ, 8>(nist_h + 32, 32, yp);
        break;
    }
    // ---- Tom-256
    for (uint32_t ctr = 0;; ctr++) {
        uint32_t d0[8], d1[8], w0[9], w1[9];
        sha256_bytes(h2c_msg(2, tag, tag_len, ctr, 0), d0);
        sha256_bytes(h2c_msg(2, tag, tag_len, ctr, 1), d1);
        le_words(d0, w0), le_words(d1, w1);
        w0[8] = w1[8] = 0;
        Ft2 hi = fe_to_mont(fe_from_words<ModT, 9>(w0)), lo = fe_to_mont(fe_from_words<ModT, 9>(w1));   // both < 2^256 < t
        Ft2 x = fe_reduce(hi * fe_const<ModT, 1>(ModT::two256) + lo);
        Ft2 x2 = x * x;
        const auto one = fe_one_mont<ModT>();
        Ft2 den = fe_reduce(one - fe_const<ModT, 1>(TOM_D_M) * x2);
        if (fe_is_zero(den)) continue;
        Ft2 u = fe_reduce(one - fe_const<ModT, 1>(TOM_A_M) * x2) * fe_inv<ModT>(den);
        Ft2 y = fe_pow_words<ModT>(u, ModT::exp_sqrt);
        if (!fe_eq(y * y, u)) continue;
        Fe<ModT, 1> xp = fe_from_mont(x), yp = negate_if_odd(fe_from_mont(y));
        uint32_t xw[9], yw[9];
        words_from_limbs<9>(xw, xp.l), words_from_limbs<9>(yw, yp.l);
        TomPt P;
        if (!tom_from_affine_words(P, xw, yw)) continue;   // cannot happen: (x, y) solves the curve equation
        TomPt Q = tom_dbl(tom_dbl(P));                      // cofactor 4
        if (fe_is_zero(Q.x)) continue;                      // 4P = identity or the point of order 2: not a generator
        Ft2 zi = fe_inv<ModT>(Q.z);
        Fe<ModT, 1> hx = fe_from_mont(Q.x * zi * fe_const<ModT, 1>(TOM_SINV_M)), hy = fe_from_mont(Q.y * zi);   // back from the a = 1 image
        store_be_bytes<ModT, 9>(tom_h, 36, hx);
        store_be_bytes<ModT, 9>(tom_h + 36, 36, hy);
        break;
    }
    return ZK_OK;
}
