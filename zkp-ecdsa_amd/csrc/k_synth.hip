// Seeded synthetic workload, generated on the GPU with the engine's own P-256 code so that
// bench.py needs nothing from the test oracle.  The byte layout is specified here (tests compare it with the restatement):
//   tag(t, S, i) = SHA-256(t || be64(S) || be64(i))
//   ring[i] = tag("ring") mod q;  d_b = tag("sk") mod (n-1) + 1;  msg_b = tag("msg");  k_b = tag("nonce") mod (n-1) + 1
//   pk_b = d_b G;  sig_b = ECDSA(d_b, msg_b, nonce k_b);  seed_b = tag("rng");  which_b = b mod n_keys;  ring[which_b] = pk_b.x
// (the reference's own test builds its inputs with WebCrypto: test/zkpAttestList.test.ts:28-40)
#include "engine.h"

template <int TL>
ZK_DEV void tag_hash(const char (&tag)[TL], uint64_t S, uint64_t i, uint32_t w[8]) {
    constexpr int taglen = TL - 1;
    uint8_t msg[64];
    for (int j = 0; j < 64; j++) msg[j] = 0;
    for (int j = 0; j < taglen; j++) msg[j] = (uint8_t)tag[j];
    for (int j = 0; j < 8; j++) msg[taglen + j] = (uint8_t)(S >> (56 - 8 * j)), msg[taglen + 8 + j] = (uint8_t)(i >> (56 - 8 * j));
    msg[taglen + 16] = 0x80;
    uint32_t bits = (taglen + 16) * 8;
    msg[62] = (uint8_t)(bits >> 8), msg[63] = (uint8_t)bits;
    uint32_t m[16], h[8];
    for (int j = 0; j < 16; j++) m[j] = (uint32_t)msg[4 * j] << 24 | (uint32_t)msg[4 * j + 1] << 16 | (uint32_t)msg[4 * j + 2] << 8 | msg[4 * j + 3];
    sha256_iv(h);
    sha256_compress(h, m);
    for (int j = 0; j < 8; j++) w[j] = h[7 - j];
}
// v mod (n-1) + 1 for a 256-bit v (one conditional subtraction: 2^256 < 2(n-1))
ZK_DEV Fe<ModN, 1> nonzero_scalar(const uint32_t v[8]) {
    uint32_t nm1[8], t[8];
    for (int j = 0; j < 8; j++) nm1[j] = ModN::mod32[j], t[j] = v[j];
    nm1[0] -= 1;  // n is odd
    if (words_geq<8>(t, nm1)) {
        uint64_t br = 0;
        for (int j = 0; j < 8; j++) {
            uint64_t d = (uint64_t)t[j] - nm1[j] - br;
            t[j] = (uint32_t)d, br = (d >> 32) & 1;
        }
    }
    uint64_t cy = 1;
    for (int j = 0; j < 8; j++) {
        cy += t[j];
        t[j] = (uint32_t)cy, cy >>= 32;
    }
    Fe<ModN, 1> r;
    limbs_from_words<8>(r.l, t);
    return r;
}
ZK_DEV P256Aff ld_pfix_s(const uint32_t* e) {
    P256Aff a;
    for (int l = 0; l < 9; l++) a.x.l[l] = e[l], a.y.l[l] = e[9 + l];
    return a;
}
ZK_DEV void fixed_mul_affine(const uint32_t* __restrict__ tab, const Fe<ModN, 1>& k, Fe<ModQ, 1>& x, Fe<ModQ, 1>& y) {
    uint32_t kw[8];
    words_from_limbs<8>(kw, k.l);
    P256Pt acc = p256_identity();
#pragma unroll 1
    for (int w = 0; w < PFIX_NWIN; w++) {
        uint32_t d = kw[0] & (PFIX_WIN_SIZE - 1);
        shr256<PFIX_WIN_BITS>(kw);
        P256Aff e = ld_pfix_s(tab + (size_t)PFIX_ENTRY_WORDS * (w * PFIX_WIN_SIZE + d));
        P256Pt s = p256_add_mixed(acc, e);
        acc = p256_select(d != 0, s, acc);
    }
    Fq2 zi = fe_inv<ModQ>(fe_reduce(acc.z));
    x = fe_from_mont(acc.x * zi), y = fe_from_mont(acc.y * zi);
}
__global__ void k_synth_ring(uint64_t S, uint64_t nkeys, uint8_t* ring) {
    uint64_t i = gtid();
    if (i >= nkeys) return;
    uint32_t w[8];
    tag_hash("ring", S, i, w);
    store_scalar_be(ring + 32 * i, fe_from_words256_reduce<ModQ>(w));
}
__global__ void __launch_bounds__(64) k_synth_proofs(const uint32_t* pfix_G, uint64_t S, uint64_t nkeys, uint64_t B, uint8_t* ring, uint8_t* msg, uint8_t* sig, uint8_t* pk,
                                                      uint32_t* which, uint8_t* seeds) {
    uint64_t b = gtid();
    if (b >= B) return;
    uint32_t w[8];
    tag_hash("sk", S, b, w);
    Fe<ModN, 1> d = nonzero_scalar(w);
    tag_hash("nonce", S, b, w);
    Fe<ModN, 1> k = nonzero_scalar(w);
    uint32_t mh[8];
    tag_hash("msg", S, b, mh);
    store_be<8>(msg + 32 * b, mh);
    tag_hash("rng", S, b, w);
    store_be<8>(seeds + 32 * b, w);
    Fe<ModQ, 1> px, py, kx, ky;
    fixed_mul_affine(pfix_G, d, px, py);
    fixed_mul_affine(pfix_G, k, kx, ky);
    store_scalar_be(pk + 64 * b, px);
    store_scalar_be(pk + 64 * b + 32, py);
    uint32_t xw[8];
    words_from_limbs<8>(xw, kx.l);
    Fe<ModN, 1> r = fe_from_words256_reduce<ModN>(xw);
    Fe<ModN, 1> z = fe_from_words256_reduce<ModN>(mh);
    // s = k^-1 (z + r d) mod n
    Fn2 km = fe_to_mont(k), rm = fe_to_mont(r);
    Fe<ModN, 1> rd = fe_canon(rm * d);
    Fe<ModN, 1> zr = fe_add_mod(z, rd);
    Fe<ModN, 1> s = fe_canon(fe_inv<ModN>(km) * zr);
    store_scalar_be(sig + 64 * b, r);
    store_scalar_be(sig + 64 * b + 32, s);
    uint32_t wi = (uint32_t)(b % nkeys);
    which[b] = wi;
    store_scalar_be(ring + 32 * (uint64_t)wi, px);
}
void launch_synth(hipStream_t s, const uint32_t* pfix_G, uint64_t seed, uint64_t nkeys, uint64_t B, uint8_t* ring, uint8_t* msg, uint8_t* sig, uint8_t* pk,
                  uint32_t* which, uint8_t* seeds) {
    hipLaunchKernelGGL(k_synth_ring, dim3((uint32_t)((nkeys + 255) / 256)), dim3(256), 0, s, seed, nkeys, ring);
    if (B) hipLaunchKernelGGL(k_synth_proofs, dim3((uint32_t)((B + 63) / 64)), dim3(64), 0, s, pfix_G, seed, nkeys, B, ring, msg, sig, pk, which, seeds);
}
__global__ void k_synth_param_scalars(uint64_t S, uint8_t* kn_be, uint8_t* kt_be) {
    if (gtid() != 0) return;
    uint32_t w[8];
    tag_hash("hnist", S, 0, w);
    store_scalar_be(kn_be, fe_from_words256_reduce<ModN>(w));
    tag_hash("htom", S, 0, w);
    store_scalar_be(kt_be, fe_from_words256_reduce<ModQ>(w));
}
void launch_synth_param_scalars(hipStream_t s, uint64_t seed, uint8_t* kn_be, uint8_t* kt_be) {
    hipLaunchKernelGGL(k_synth_param_scalars, dim3(1), dim3(64), 0, s, seed, kn_be, kt_be);
}
