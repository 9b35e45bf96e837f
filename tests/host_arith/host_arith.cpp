// TEST HARNESS (not product code): compiles the engine's own arithmetic headers (zkp-ecdsa_amd/csrc/field.h, curve.h) for the
// host CPU with g++ -DZK_HOST_BUILD, so that `pytest -m "not gpu"` checks the very templates the HIP kernels instantiate --
// Montgomery product, magnitude-typed add/sub, fused double subtraction, inversion, the P-256 complete formulas, the Tom-256
// (a = 1 image) extended / niels formulas, SHA-256 with its byte absorber, and the RNG draw mapping -- against the oracle.  Built and driven by tests/test_host_arith.py.
#define ZK_HOST_BUILD 1
#include <cstring>
#include "curve.h"
#include "coop.h"
#include "rng.h"
#include "comb_digits.h"
#include "ktab.h"
#include <vector>

static uint32_t bswap32h(uint32_t v) { return __builtin_bswap32(v); }
// 40-byte big-endian operand -> 10 little-endian 32-bit words
static void be40_to_words(const uint8_t* p, uint32_t w[10]) {
    for (int i = 0; i < 10; i++) {
        uint32_t v;
        memcpy(&v, p + 4 * (9 - i), 4);
        w[i] = bswap32h(v);
    }
}
static void words_to_be(const uint32_t* w, int nw, uint8_t* out) {  // nw words -> 4*nw bytes big-endian
    for (int i = 0; i < nw; i++) {
        uint32_t v = bswap32h(w[nw - 1 - i]);
        memcpy(out + 4 * i, &v, 4);
    }
}
static void be_to_words(const uint8_t* p, int nbytes, uint32_t* w, int nw) {
    for (int i = 0; i < nw; i++) w[i] = 0;
    for (int i = 0; i < nbytes; i++) {
        int bi = nbytes - 1 - i;
        w[bi / 4] |= (uint32_t)p[i] << (8 * (bi % 4));
    }
}

template <class M>
static void field_one(int op, const uint8_t* a, const uint8_t* b, uint8_t* out) {
    uint32_t aw[10], bw[10];
    be40_to_words(a, aw), be40_to_words(b, bw);
    Fe<M, 1> x = fe_from_words<M, 9>(aw), y = fe_from_words<M, 9>(bw), r;
    if (op == 0) r = fe_mul_mod(x, y);
    else if (op == 1) r = fe_add_mod(x, y);
    else if (op == 2) r = fe_sub_mod(x, y);
    else if (op == 3) r = fe_from_mont(fe_inv_fermat<M>(fe_to_mont(x)));
    else if (op == 7) r = fe_from_mont(fe_inv_gcd<M>(fe_to_mont(x)));   // divsteps inversion
    else if (op == 4) {  // x*y - x - y through fe_sub2
        auto xm = fe_to_mont(x), ym = fe_to_mont(y);
        r = fe_from_mont(fe_sub2(xm * ym, xm, ym));
    } else if (op == 6) {  // (x + y)^2 through limbs_mont_sqr
        r = fe_from_mont(fe_sqr(fe_to_mont(x) + fe_to_mont(y)));
    } else {  // op 5: a lazy chain at large magnitudes: ((x + y) + (x + y)) * (x - y) - (y * y + x) , all in Montgomery form
        auto xm = fe_to_mont(x), ym = fe_to_mont(y);
        auto s = (xm + ym) + (xm + ym);          // K = 8
        auto d = xm - ym;                        // K = 2 + 4
        auto t = s * d - (ym * ym + xm);         // product of K 8 x 6, subtrahend K 4
        r = fe_from_mont(t);
    }
    uint32_t rw[10];
    words_from_limbs<9>(rw, r.l);
    rw[9] = 0;
    words_to_be(rw, 10, out);
}
extern "C" int ha_field_op(int which, int op, uint64_t count, const uint8_t* a, const uint8_t* b, uint8_t* out) {
    for (uint64_t i = 0; i < count; i++) {
        if (which == 0) field_one<ModQ>(op, a + 40 * i, b + 40 * i, out + 40 * i);
        else if (which == 1) field_one<ModN>(op, a + 40 * i, b + 40 * i, out + 40 * i);
        else field_one<ModT>(op, a + 40 * i, b + 40 * i, out + 40 * i);
    }
    return 0;
}

// ---------------------------------------------------------------- Tom-256
static bool tom_load(TomPt& r, const uint8_t* xy72) {
    uint32_t xw[9], yw[9];
    be_to_words(xy72, 36, xw, 9), be_to_words(xy72 + 36, 36, yw, 9);
    return tom_from_affine_words(r, xw, yw);
}
static TomNiels tom_niels_of(const TomPt& p) {  // affine image point (Z = 1) -> table-entry form (x, y, (d/a) x y)
    TomNiels n;
    n.x = p.x, n.y = p.y;
    n.dt = p.t * fe_const<ModT, 1>(TOM_D1_M);
    return n;
}
static void tom_store(const TomPt& p, uint8_t* out72) {  // image extended -> original-curve affine bytes
    Ft2 zi = fe_inv<ModT>(p.z);
    auto x = fe_from_mont((p.x * zi) * fe_const<ModT, 1>(TOM_SINV_M));
    auto y = fe_from_mont(p.y * zi);
    uint32_t w[9];
    words_from_limbs<9>(w, x.l);
    words_to_be(w, 9, out72);
    words_from_limbs<9>(w, y.l);
    words_to_be(w, 9, out72 + 36);
}
// k*P by double-and-add: tom_dbl + (odd elements: general tom_add, even elements: tom_add_niels with the precomputed entry)
extern "C" int ha_tom_mul(uint64_t count, const uint8_t* xy72, const uint8_t* k32, uint8_t* out72) {
    int bad = 0;
    for (uint64_t i = 0; i < count; i++) {
        TomPt P;
        if (!tom_load(P, xy72 + 72 * i)) {
            bad++;
            memset(out72 + 72 * i, 0xff, 72);
            continue;
        }
        TomNiels N = tom_niels_of(P);
        TomPt acc = tom_identity();
        for (int bit = 255; bit >= 0; bit--) {
            acc = tom_dbl(acc);
            if ((k32[32 * i + 31 - bit / 8] >> (bit % 8)) & 1) acc = (i & 1) ? tom_add(acc, P) : tom_add_niels(acc, N);
        }
        tom_store(acc, out72 + 72 * i);
    }
    return bad;
}
// the validation-only curve check on plain coordinates (k_verify.hip: tom_bytes_valid) beside the loader's check: out[i] = 2 * on_curve + loader_ok
extern "C" int ha_tom_on_curve(uint64_t count, const uint8_t* xy72, uint8_t* out) {
    for (uint64_t i = 0; i < count; i++) {
        uint32_t xw[9], yw[9];
        be_to_words(xy72 + 72 * i, 36, xw, 9), be_to_words(xy72 + 72 * i + 36, 36, yw, 9);
        TomPt P;
        out[i] = (uint8_t)(2 * (tom_words_on_curve(xw, yw) ? 1 : 0) + (tom_from_affine_words(P, xw, yw) ? 1 : 0));
    }
    return 0;
}
// P + Q - R through the comb's entry forms: from_niels (first step), add_niels, negated entry + add_niels_last (last step)
extern "C" int ha_tom_combo(uint64_t count, const uint8_t* p72, const uint8_t* q72, const uint8_t* r72, uint8_t* out72) {
    int bad = 0;
    for (uint64_t i = 0; i < count; i++) {
        TomPt P, Q, R;
        if (!tom_load(P, p72 + 72 * i) || !tom_load(Q, q72 + 72 * i) || !tom_load(R, r72 + 72 * i)) {
            bad++;
            continue;
        }
        TomPt acc = tom_from_niels(tom_niels_of(P));
        acc = tom_add_niels(acc, tom_niels_of(Q));
        acc = tom_add_niels_last(acc, tom_niels_neg_sel(tom_niels_of(R), true));
        tom_store(acc, out72 + 72 * i);
    }
    return bad;
}

// ---------------------------------------------------------------- P-256
static bool p256_load(P256Aff& a, const uint8_t* xy64) {
    uint32_t xw[8], yw[8];
    be_to_words(xy64, 32, xw, 8), be_to_words(xy64 + 32, 32, yw, 8);
    a.x = fe_to_mont(fe_from_words<ModQ, 8>(xw));
    a.y = fe_to_mont(fe_from_words<ModQ, 8>(yw));
    return p256_on_curve(a);
}
static void p256_store(const P256Pt& p, uint8_t* out64) {  // identity -> 64 zero bytes
    Fq2 z = fe_reduce(p.z);
    if (fe_is_zero(z)) {
        memset(out64, 0, 64);
        return;
    }
    Fq2 zi = fe_inv<ModQ>(z);
    uint32_t w[9];
    words_from_limbs<9>(w, fe_from_mont(p.x * zi).l);
    words_to_be(w, 8, out64);
    words_from_limbs<9>(w, fe_from_mont(p.y * zi).l);
    words_to_be(w, 8, out64 + 32);
}
extern "C" int ha_p256_mul(uint64_t count, const uint8_t* xy64, const uint8_t* k32, uint8_t* out64) {
    int bad = 0;
    for (uint64_t i = 0; i < count; i++) {
        P256Aff A;
        if (!p256_load(A, xy64 + 64 * i)) {
            bad++;
            memset(out64 + 64 * i, 0xff, 64);
            continue;
        }
        P256Pt P = p256_from_affine(A), acc = p256_identity();
        for (int bit = 255; bit >= 0; bit--) {
            acc = p256_dbl(acc);
            if ((k32[32 * i + 31 - bit / 8] >> (bit % 8)) & 1) {
                // the mixed formula needs a non-identity accumulator only for correctness of Z2 = 1, not of the law; use it when acc != O
                if ((i & 1) && !fe_is_zero(fe_reduce(acc.z))) acc = p256_add_mixed(acc, A);
                else acc = p256_add(acc, P);
            }
        }
        p256_store(acc, out64 + 64 * i);
    }
    return bad;
}
// complete addition on arbitrary pairs (P = Q, P = -Q included)
// 2^nd * P through the Jacobian doubling chain of the per-proof window tables (curve.h: p256_jdbl) and back to the homogeneous form
extern "C" int ha_p256_jdbl_chain(uint64_t count, const uint8_t* xy64, uint32_t nd, int from_identity, uint8_t* out64) {
    for (uint64_t i = 0; i < count; i++) {
        P256Aff A;
        if (!p256_load(A, xy64 + 64 * i)) return 1;
        P256Jac j = p256_jac_from(from_identity ? p256_identity() : p256_from_affine(A));
        for (uint32_t k = 0; k < nd; k++) j = p256_jdbl(j);
        p256_store(p256_from_jac(j), out64 + 64 * i);
    }
    return 0;
}
extern "C" int ha_p256_add(uint64_t count, const uint8_t* p64, const uint8_t* q64, uint8_t* out64) {
    int bad = 0;
    for (uint64_t i = 0; i < count; i++) {
        P256Aff A, B;
        if (!p256_load(A, p64 + 64 * i) || !p256_load(B, q64 + 64 * i)) {
            bad++;
            continue;
        }
        p256_store(p256_add(p256_from_affine(A), p256_from_affine(B)), out64 + 64 * i);
    }
    return bad;
}

// ---------------------------------------------------------------- SHA-256 byte absorber (sha256.h) and the RNG contract (rng.h)
extern "C" int ha_sha256(uint64_t count, uint64_t len, const uint8_t* msgs, uint8_t* out32) {
    for (uint64_t t = 0; t < count; t++) {
        uint32_t col[16];  // one lane's column of the LDS block buffer, stride 1
        ShaStream s;
        s.init(col, 0, 1);
        for (uint64_t i = 0; i < len; i++) s.put_byte(msgs[t * len + i]);
        uint32_t h[8];
        s.finish(h);
        for (int i = 0; i < 8; i++) {  // digest = h[0] .. h[7], each big-endian
            uint32_t v = bswap32h(h[i]);
            memcpy(out32 + 32 * t + 4 * i, &v, 4);
        }
    }
    return 0;
}
// The word-wise path of the absorber (put_be<N> -> put_word at any byte alignment): `lead` single bytes, then `count` values of
// `nbytes` (32, 33 or 36) bytes each taken from `vals` (big-endian, nbytes apiece), then `trail` single bytes -- the shape of
// hashPoints: 0x04 || X || Y per point.  The digest must be that of the same bytes absorbed one by one.
extern "C" int ha_sha256_values(uint64_t lead, const uint8_t* lead_bytes, uint64_t count, int nbytes, const uint8_t* vals, uint64_t trail,
                                const uint8_t* trail_bytes, uint8_t* out32) {
    uint32_t col[16];
    ShaStream s;
    s.init(col, 0, 1);
    for (uint64_t i = 0; i < lead; i++) s.put_byte(lead_bytes[i]);
    for (uint64_t k = 0; k < count; k++) {
        uint32_t w[9] = {0};   // little-endian words of the value
        for (int i = 0; i < nbytes; i++) w[i >> 2] |= (uint32_t)vals[k * nbytes + (nbytes - 1 - i)] << (8 * (i & 3));
        if (nbytes == 32) s.put_be<32>(w);
        else if (nbytes == 33) s.put_be<33>(w);
        else if (nbytes == 36) s.put_be<36>(w);
        else return 1;
    }
    for (uint64_t i = 0; i < trail; i++) s.put_byte(trail_bytes[i]);
    uint32_t h[8];
    s.finish(h);
    for (int i = 0; i < 8; i++) {
        uint32_t v = bswap32h(h[i]);
        memcpy(out32 + 4 * i, &v, 4);
    }
    return 0;
}
// logical draws first_k .. first_k + n_k - 1 of every proof, 32 bytes big-endian each; exc_* as k_rng_prepass would leave them
extern "C" int ha_rng_draws(int mode, int sec, uint64_t B, const uint8_t* data, uint64_t stride_blocks, uint32_t* exc_idx, uint32_t* exc_flags,
                            uint32_t* exc_cnt, uint32_t first_k, uint32_t n_k, uint8_t* out) {
    RngCtx g;
    g.seeds = mode == 0 ? data : nullptr;
    g.stream = mode == 1 ? data : nullptr;
    g.stride_blocks = stride_blocks, g.mode = mode, g.sec = sec;
    g.exc_idx = exc_idx, g.exc_flags = exc_flags, g.exc_cnt = exc_cnt, g.proof_base = 0;
    for (uint64_t p = 0; p < B; p++)
        for (uint32_t k = 0; k < n_k; k++) {
            uint32_t w[8];
            rng_draw_words(g, (uint32_t)p, first_k + k, w);
            words_to_be(w, 8, out + 32 * (p * n_k + k));
        }
    return 0;
}

// ---------------------------------------------------------------- comb digit recoding (comb_digits.h)
// digits of the 256-bit scalar k (big-endian) for a `bits`-wide comb: idx[i], neg[i] for i < nwin; returns the carry left over
// plus whatever is left of the scalar (must be 0 when nwin windows cover it)
extern "C" uint32_t ha_comb_digits(uint32_t bits, uint32_t nwin, const uint8_t* k32, uint32_t* idx, uint8_t* neg) {
    CombDigits d;
    d.init(bits);
    be_to_words(k32, 32, d.w, 8);
    for (uint32_t i = 0; i < nwin; i++) {
        bool ng;
        d.next(idx[i], ng);
        neg[i] = ng ? 1 : 0;
    }
    uint32_t rest = d.carry;
    for (int i = 0; i < 8; i++) rest |= d.w[i];
    return rest;
}

// digits of k for the per-key tables (comb_digits.h, KeyDigits): 33 signed 8-bit digits; returns what is left over (must be 0)
extern "C" uint32_t ha_key_digits(const uint8_t* k32, uint32_t* dig, uint8_t* neg) {
    KeyDigits d;
    d.init();
    be_to_words(k32, 32, d.w, 8);
    for (uint32_t i = 0; i < 33; i++) {
        bool ng;
        d.next(dig[i], ng);
        neg[i] = ng ? 1 : 0;
    }
    uint32_t rest = d.carry;
    for (int i = 0; i < 8; i++) rest |= d.w[i];
    return rest;
}

// ---------------------------------------------------------------- per-key table (ktab.h): one key's table built here the way k_ktab.hip
// builds it (window bases by doubling, 128 multiples by complete additions, affine canonical entries), then acc0 + k * (+-P) through
// p256_ktab_mul_acc for every scalar.  xy64: the table's base point; start64: the accumulator's start (64 zero bytes = identity).
extern "C" int ha_ktab_mul(const uint8_t* xy64, const uint8_t* start64, uint64_t count, const uint8_t* k32, const uint8_t* negs, uint8_t* out64) {
    P256Aff A;
    if (!p256_load(A, xy64)) return -1;
    std::vector<uint32_t> tab(KTAB_KEY_WORDS);
    P256Pt base = p256_from_affine(A);
    for (uint32_t w = 0; w < KTAB_NWIN; w++) {
        P256Pt acc = base;
        for (uint32_t d = 1; d <= KTAB_ENT; d++) {
            if (d > 1) acc = p256_add(acc, base);
            Fq2 zi = fe_inv<ModQ>(fe_reduce(acc.z));
            st_ktab(tab.data() + ((size_t)w * KTAB_ENT + d - 1) * KTAB_ENTRY_WORDS, fe_canon(acc.x * zi), fe_canon(acc.y * zi));
        }
        for (uint32_t k = 0; k < KTAB_BITS; k++) base = p256_dbl(base);
    }
    P256Pt start = p256_identity();
    bool have_start = false;
    for (int i = 0; i < 64; i++) have_start = have_start || start64[i] != 0;
    if (have_start) {
        P256Aff S;
        if (!p256_load(S, start64)) return -2;
        start = p256_from_affine(S);
    }
    int range_mismatch = 0;
    for (uint64_t i = 0; i < count; i++) {
        uint32_t kw[8];
        be_to_words(k32 + 32 * i, 32, kw, 8);
        p256_store(p256_ktab_mul_acc(start, tab.data(), kw, negs[i] != 0), out64 + 64 * i);
        // the same sum as four lanes take it (k_front_wide, k_exp_commit_kt_wide): a quarter of the windows each from the identity, then added up
        P256Pt sum = start;
        const uint32_t per = (KTAB_NWIN + 3) / 4;
        for (uint32_t part = 0; part < 4; part++) {
            be_to_words(k32 + 32 * i, 32, kw, 8);
            sum = p256_add(sum, p256_ktab_mul_range(p256_identity(), tab.data(), kw, negs[i] != 0, part * per, per));
        }
        uint8_t alt[64];
        p256_store(sum, alt);
        if (memcmp(alt, out64 + 64 * i, 64)) range_mismatch++;
    }
    return range_mismatch;
}

// ---------------------------------------------------------------- lane-cooperative arithmetic (coop.h) through its SIMT emulation
// Four operand pairs at a time, one per row.  op 0: a * b;  1: a + b;  2: a - b;  3: a lazy chain ((a + b) + (a + b)) * (a - b) - (b * b + a) as in field op 5;
// 4: a^(M - 2).  Operands and results are plain canonical values (40-byte big-endian), the arithmetic runs in the Montgomery domain.
template <class M>
static void co_field_four(int op, const uint8_t* a, const uint8_t* b, uint8_t* out) {
    uint32_t am[4][NLIMB], bm[4][NLIMB], rm[4][NLIMB];
    for (int r = 0; r < 4; r++) {
        uint32_t aw[10], bw[10];
        be40_to_words(a + 40 * r, aw), be40_to_words(b + 40 * r, bw);
        Fe<M, 2> x = fe_to_mont(fe_from_words<M, 9>(aw)), y = fe_to_mont(fe_from_words<M, 9>(bw));
        for (int i = 0; i < NLIMB; i++) am[r][i] = x.l[i], bm[r][i] = y.l[i];
    }
    const CoU32 mj = co_limbs(M::mod);
    const auto x = co_load4<M, 2>(am[0], am[1], am[2], am[3]), y = co_load4<M, 2>(bm[0], bm[1], bm[2], bm[3]);
    CoFe<M, 2> res;
    if (op == 0) res = co_mul(x, y, mj);
    else if (op == 1) res = co_mul(co_add(x, y), co_load4<M, 1>(M::one, M::one, M::one, M::one), mj);
    else if (op == 2) res = co_mul(co_sub(x, y), co_load4<M, 1>(M::one, M::one, M::one, M::one), mj);
    else if (op == 3) {
        const auto s = co_add(co_add(x, y), co_add(x, y));
        const auto d = co_sub(x, y);
        const auto t = co_sub(co_mul(s, d, mj), co_add(co_mul(y, y, mj), x));
        res = co_mul(t, co_load4<M, 1>(M::one, M::one, M::one, M::one), mj);
    } else res = co_pow_words<M>(x, M::exp_m2, mj);
    co_store4(res, rm[0], rm[1], rm[2], rm[3]);
    for (int r = 0; r < 4; r++) {
        Fe<M, 2> v;
        for (int i = 0; i < NLIMB; i++) v.l[i] = rm[r][i];
        uint32_t rw[10];
        words_from_limbs<9>(rw, fe_from_mont(v).l);
        rw[9] = 0;
        words_to_be(rw, 10, out + 40 * r);
    }
}
extern "C" int ha_co_field_op(int which, int op, uint64_t count /* multiple of 4 */, const uint8_t* a, const uint8_t* b, uint8_t* out) {
    for (uint64_t i = 0; i + 4 <= count; i += 4) {
        if (which == 0) co_field_four<ModQ>(op, a + 40 * i, b + 40 * i, out + 40 * i);
        else if (which == 1) co_field_four<ModN>(op, a + 40 * i, b + 40 * i, out + 40 * i);
        else co_field_four<ModT>(op, a + 40 * i, b + 40 * i, out + 40 * i);
    }
    return 0;
}
static CoTom co_tom_from(const TomPt& p) {
    CoTom c;
    c.v = co_load4<ModT, 2>(p.x.l, p.y.l, p.t.l, p.z.l);
    return c;
}
static TomPt co_tom_to(const CoTom& c) {
    TomPt p;
    co_store4(c.v, p.x.l, p.y.l, p.t.l, p.z.l);
    return p;
}
// k * P by double-and-add with the cooperative doubling and addition (rows X, Y, T, Z of one wave)
extern "C" int ha_co_tom_mul(uint64_t count, const uint8_t* xy72, const uint8_t* k32, uint8_t* out72) {
    int bad = 0;
    const CoU32 mj = co_limbs(ModT::mod);
    for (uint64_t i = 0; i < count; i++) {
        TomPt P;
        if (!tom_load(P, xy72 + 72 * i)) {
            bad++;
            memset(out72 + 72 * i, 0xff, 72);
            continue;
        }
        const CoTom cp = co_tom_from(P);
        CoTom acc = co_tom_from(tom_identity());
        for (int bit = 255; bit >= 0; bit--) {
            acc = co_tom_dbl(acc, mj);
            if ((k32[32 * i + 31 - bit / 8] >> (bit % 8)) & 1) acc = co_tom_add(acc, cp, mj);
        }
        tom_store(co_tom_to(acc), out72 + 72 * i);
    }
    return bad;
}
static CoP256 co_p256_from(const P256Pt& p) {
    CoP256 c;
    const uint32_t zero[NLIMB] = {0};
    c.v = co_load4<ModQ, 8>(p.x.l, p.y.l, p.z.l, zero);
    return c;
}
static P256Pt co_p256_to(const CoP256& c) {
    P256Pt p;
    co_store4(c.v, p.x.l, p.y.l, p.z.l, nullptr);
    return p;
}
// k * P by double-and-add with the cooperative complete laws (rows X, Y, Z); the accumulator starts at the identity, so O + P, P + P and P + O all occur
extern "C" int ha_co_p256_mul(uint64_t count, const uint8_t* xy64, const uint8_t* k32, uint8_t* out64) {
    const CoU32 mj = co_limbs(ModQ::mod);
    for (uint64_t i = 0; i < count; i++) {
        P256Aff A;
        if (!p256_load(A, xy64 + 64 * i)) return 1;
        const CoP256 cp = co_p256_from(p256_from_affine(A));
        CoP256 acc = co_p256_from(p256_identity());
        for (int bit = 255; bit >= 0; bit--) {
            acc = co_p256_dbl(acc, mj);
            if ((k32[32 * i + 31 - bit / 8] >> (bit % 8)) & 1) acc = co_p256_add(acc, cp, mj);
        }
        p256_store(co_p256_to(acc), out64 + 64 * i);
    }
    return 0;
}
extern "C" int ha_co_p256_add(uint64_t count, const uint8_t* p64, const uint8_t* q64, uint8_t* out64) {
    const CoU32 mj = co_limbs(ModQ::mod);
    for (uint64_t i = 0; i < count; i++) {
        P256Aff A, B;
        if (!p256_load(A, p64 + 64 * i) || !p256_load(B, q64 + 64 * i)) return 1;
        p256_store(co_p256_to(co_p256_add(co_p256_from(p256_from_affine(A)), co_p256_from(p256_from_affine(B)), mj)), out64 + 64 * i);
    }
    return 0;
}
// 2^nd * P through the cooperative Jacobian-with-ZZ doubling chain (k_rtab_base), back to the homogeneous form (X Z : Y : Z ZZ)
extern "C" int ha_co_p256_jdbl_chain(uint64_t count, const uint8_t* xy64, uint32_t nd, int from_identity, uint8_t* out64) {
    const CoU32 mj = co_limbs(ModQ::mod);
    for (uint64_t i = 0; i < count; i++) {
        P256Aff A;
        if (!p256_load(A, xy64 + 64 * i)) return 1;
        const P256Pt s = from_identity ? p256_identity() : p256_from_affine(A);
        CoP256J j;
        j.v = co_load4<ModQ, 10>(s.x.l, s.y.l, s.z.l, s.z.l);   // Z in {0, 1}: ZZ = Z
        for (uint32_t k = 0; k < nd; k++) j = co_p256_jdbl(j, mj);
        p256_store(co_p256_to(co_p256_from_jac(j, mj)), out64 + 64 * i);
    }
    return 0;
}
