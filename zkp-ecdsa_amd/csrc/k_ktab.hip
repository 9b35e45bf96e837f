// Per-key tables of the ring: built once per zk_ctx_set_ring for rings of up to 2^KTAB_MAXN keys (layout and purpose: engine.h).
//
// The ring holds keyToInt(publicKey) = the affine x of every key (src/zkpAttestList.ts:94-102), so the builder lifts x to a curve
// point first: y = (x^3 - 3x + b)^((p+1)/4) (p = 3 mod 4), kept when y^2 matches -- a ring value that is no x-coordinate gets no
// table (ktab_ok = 0) and proofs that name it use the per-proof table of R instead.  Which of the two roots the table belongs to does
// not matter: k * (x, -y) = -(k * (x, y)), the prover compares its own y with the table's base point (k_front).
//
//   k_ktab_base   one thread per key: the lift and the 33 window bases 2^(8 w) P (256 doublings), projective, into the temp area
//   k_ktab_fill   one thread per (key, window) of a slab of keys: d * base for d = 1..128 by 127 complete additions (weier.ts:176-230),
//                 normalised to affine with Montgomery's trick over the thread's own 128 entries (one Fermat inversion per thread)
// Entries are exact group elements in canonical Montgomery form, so every multiple taken through them equals the reference's
// Point.mul (group.ts:133-152) as a group element -- only affine coordinates are ever observable.
#include "rtab.h"   // engine.h (and with it ktab.h), the projective table entries of rtab.h

#define KTB_BASE_WORDS 28   // X, Y, Z (9 limbs each) + pad
#define KTB_TMP_WORDS 36    // X, Y, Z, prefix product of the Z's before this entry

__global__ void __launch_bounds__(64) k_ktab_base(Soa ring, uint32_t N, uint32_t* bases, uint8_t* ok) {
    uint32_t i = gtid();
    if (i >= N) return;
    const auto b = fe_const<ModQ, 1>(P256_B_M);
    Fq2 x = fe_to_mont(soa_ld<ModQ, 1>(ring, i));
    auto x2 = x * x;
    auto x3 = x2 * x;
    Fq2 rhs = fe_reduce((x3 + b) - (x + x + x));
    Fq2 y = fe_pow_words<ModQ>(rhs, ModQ::exp_sqrt);
    bool good = fe_eq(y * y, rhs);
    ok[i] = good ? 1 : 0;
    if (!good) return;
    P256Aff a;
    a.x = x, a.y = y;
    P256Pt p = p256_from_affine(a);
    uint32_t* e = bases + (size_t)i * KTAB_NWIN * KTB_BASE_WORDS;
#pragma unroll 1
    for (uint32_t w = 0; w < KTAB_NWIN; w++) {
        st_rtab(e + (size_t)w * KTB_BASE_WORDS, p);
#pragma unroll 1
        for (uint32_t k = 0; k < KTAB_BITS; k++) p = p256_dbl(p);
    }
}
ZK_DEV void ktb_st9(uint32_t* e, const uint32_t l[NLIMB]) {
#pragma unroll
    for (int k = 0; k < NLIMB; k++) e[k] = l[k];
}
template <int K>
ZK_DEV Fe<ModQ, K> ktb_ld9(const uint32_t* e) {
    Fe<ModQ, K> r;
#pragma unroll
    for (int k = 0; k < NLIMB; k++) r.l[k] = e[k];
    return r;
}
__global__ void __launch_bounds__(256) k_ktab_fill(const uint32_t* __restrict__ bases, const uint8_t* __restrict__ ok, uint32_t first, uint32_t count, uint32_t* tmp,
                                                   uint32_t* ktab) {
    uint32_t t = gtid();
    if (t >= count * KTAB_NWIN) return;
    const uint32_t key = first + t / KTAB_NWIN, w = t % KTAB_NWIN;
    if (!ok[key]) return;
    const P256Pt base = ld_rtab(bases + ((size_t)key * KTAB_NWIN + w) * KTB_BASE_WORDS);
    uint32_t* my = tmp + (size_t)t * KTAB_ENT * KTB_TMP_WORDS;
    uint32_t* out = ktab + ((size_t)key * KTAB_NWIN + w) * KTAB_ENT * KTAB_ENTRY_WORDS;
    P256Pt acc = base;
    Fq2 run = fe_one_mont<ModQ>().as<2>();
#pragma unroll 1
    for (uint32_t d = 1; d <= KTAB_ENT; d++) {   // slot d - 1 = d * base
        if (d > 1) acc = p256_add(acc, base);
        uint32_t* e = my + (size_t)(d - 1) * KTB_TMP_WORDS;
        Fq2 z = fe_reduce(acc.z);
        ktb_st9(e, acc.x.l), ktb_st9(e + 9, acc.y.l), ktb_st9(e + 18, z.l), ktb_st9(e + 27, run.l);
        run = run * z;
    }
    Fq2 inv = fe_inv<ModQ>(run);
#pragma unroll 1
    for (int d = KTAB_ENT - 1; d >= 0; d--) {
        const uint32_t* e = my + (size_t)d * KTB_TMP_WORDS;
        Fq2 zi = inv * ktb_ld9<2>(e + 27);
        inv = inv * ktb_ld9<2>(e + 18);
        Fq2 x = ktb_ld9<8>(e) * zi, y = ktb_ld9<8>(e + 9) * zi;
        st_ktab(out + (size_t)d * KTAB_ENTRY_WORDS, fe_canon(x), fe_canon(y));
    }
}
size_t ktab_temp_bytes(uint64_t N, uint32_t slab_keys) {
    return sizeof(uint32_t) * ((size_t)N * KTAB_NWIN * KTB_BASE_WORDS + (size_t)slab_keys * KTAB_NWIN * KTAB_ENT * KTB_TMP_WORDS);
}
void launch_ktab_build(hipStream_t s, const Soa& ring, uint64_t N, uint32_t* ktab, uint8_t* ok, void* temp, uint32_t slab_keys) {
    uint32_t* bases = (uint32_t*)temp;
    uint32_t* tmp = bases + (size_t)N * KTAB_NWIN * KTB_BASE_WORDS;
    hipLaunchKernelGGL(k_ktab_base, dim3((uint32_t)((N + 63) / 64)), dim3(64), 0, s, ring, (uint32_t)N, bases, ok);
    for (uint64_t first = 0; first < N; first += slab_keys) {
        uint32_t cnt = (uint32_t)std::min<uint64_t>(slab_keys, N - first);
        hipLaunchKernelGGL(k_ktab_fill, dim3((cnt * KTAB_NWIN + 255) / 256), dim3(256), 0, s, bases, ok, (uint32_t)first, cnt, tmp, ktab);
    }
}
