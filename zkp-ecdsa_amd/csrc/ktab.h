// Per-key tables of the ring (built by k_ktab.hip at zk_ctx_set_ring) and k * P through them.  Depends on the arithmetic headers only, so
// tests/host_arith compiles it for the host as well (table of one key, every multiplication against the oracle).
#pragma once
#include "comb_digits.h"
#include "curve.h"

// per-KEY tables (k_ktab.hip): every ring key P gets d * 2^(8 w) * P, d = 1..128, w = 0..32, affine, 64 bytes per entry (slot d - 1 of
// window w): 264 KB per key, 17.7 GB for a ring of 2^16 keys.  A scalar is recoded into signed 8-bit digits in [-127, 128] (a negative
// digit negates the entry on load), so a prover's k * pk is 33 mixed additions of gathered entries, and alpha_i * R of proveExp
// (exp.ts:144-149) becomes (alpha_i u1) * G + (alpha_i u2) * pk -- no per-proof table of R, no doubling chain in the front end.
#define KTAB_BITS 8
#define KTAB_NWIN 33
#define KTAB_ENT 128
#define KTAB_ENTRY_WORDS 16
#define KTAB_MAXN 17   // 35 GB at 2^17 keys (the reference's own bench ring, 100 001 keys, pads to that); zk_ctx_set_ring falls back when the HBM is not there
#define KTAB_KEY_WORDS ((size_t)KTAB_NWIN * KTAB_ENT * KTAB_ENTRY_WORDS)

ZK_DEV P256Aff ld_ktab(const uint32_t* e, bool neg = false) {   // neg: (x, p - y), on the words (entries are canonical, y != 0 on this curve)
    uint32_t w[16];
#ifdef ZK_HOST_BUILD
    for (int i = 0; i < 16; i++) w[i] = e[i];
#else
    const uint4* q = (const uint4*)e;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        uint4 v = q[i];
        w[4 * i] = v.x, w[4 * i + 1] = v.y, w[4 * i + 2] = v.z, w[4 * i + 3] = v.w;
    }
#endif
    uint64_t br = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint64_t d = (uint64_t)ModQ::mod32[i] - w[8 + i] - br;
        br = (d >> 32) & 1;
        w[8 + i] = neg ? (uint32_t)d : w[8 + i];
    }
    P256Aff a;
    limbs_from_words<8>(a.x.l, w);
    limbs_from_words<8>(a.y.l, w + 8);
    return a;
}
ZK_DEV void st_ktab(uint32_t* e, const Fe<ModQ, 1>& x, const Fe<ModQ, 1>& y) {
    uint32_t w[16];
    words_from_limbs<8>(w, x.l);
    words_from_limbs<8>(w + 8, y.l);
#ifdef ZK_HOST_BUILD
    for (int i = 0; i < 16; i++) e[i] = w[i];
#else
    uint4* q = (uint4*)e;
#pragma unroll
    for (int i = 0; i < 4; i++) q[i] = make_uint4(w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]);
#endif
}
// acc + k * P for the ring key whose table starts at kt; k < 2^256 as 8 little-endian words (destroyed).  33 gathers of 64 bytes and
// 33 mixed complete additions (weier.ts:176-230 with Z2 = 1) over signed 8-bit digits; neg: the prover's key is the NEGATIVE of the
// table's base point (the ring holds x-coordinates only, the table was built for one of the two roots), which flips every digit's sign.
ZK_DEV P256Pt p256_ktab_mul_acc(P256Pt acc, const uint32_t* __restrict__ kt, uint32_t kw[8], bool neg) {
    KeyDigits kd;   // comb_digits.h (checked on the host by tests/test_host_arith.py)
    kd.init();
#pragma unroll
    for (int i = 0; i < 8; i++) kd.w[i] = kw[i];
#pragma unroll 1
    for (uint32_t w = 0; w < KTAB_NWIN; w++) {
        uint32_t d;
        bool dn;
        kd.next(d, dn);
        ZK_ADD_IF(d != 0, acc, p256_add_mixed(acc, ld_ktab(kt + ((size_t)w * KTAB_ENT + (d ? d - 1 : 0)) * KTAB_ENTRY_WORDS, neg != dn)));   // a zero digit (2^-8) idles its lane (curve.h: ZK_UNIFORM_CF)
    }
    return acc;
}
// The same sum restricted to windows [w0, w0 + per): the table holds every window multiple, so several lanes can take a range each (rtab.h:
// p256_rtab_mul_range has the reasoning behind the two loops: the carries below w0 are integer work with a per-lane trip count, the additions have one
// trip count for the whole wave).  Windows past the last one contribute nothing.
ZK_DEV P256Pt p256_ktab_mul_range(P256Pt acc, const uint32_t* __restrict__ kt, uint32_t kw[8], bool neg, uint32_t w0, uint32_t per) {
    KeyDigits kd;
    kd.init();
#pragma unroll
    for (int i = 0; i < 8; i++) kd.w[i] = kw[i];
    uint32_t d;
    bool dn;
#pragma unroll 1
    for (uint32_t w = 0; w < w0; w++) kd.next(d, dn);
#pragma unroll 1
    for (uint32_t j = 0; j < per; j++) {
        const uint32_t w = w0 + j;
        kd.next(d, dn);
        ZK_ADD_IF(w < KTAB_NWIN && d != 0, acc, p256_add_mixed(acc, ld_ktab(kt + ((size_t)(w < KTAB_NWIN ? w : 0) * KTAB_ENT + (d ? d - 1 : 0)) * KTAB_ENTRY_WORDS, neg != dn)));
    }
    return acc;
}
