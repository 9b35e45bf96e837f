// Microbenchmark (gfx950): the dependent chains of the engine, one lane against a cooperating wave.
//   dbl chain : N Tom-256 doublings in a row -- curve.h's tom_dbl in ONE lane (what k_msm_red_last / k_v_straus do today) against coop.h's
//               co_tom_dbl (one limb per lane, four products per wave).  VERDICT r05 item 1's kill criterion: the cooperative chain must be >= 2x faster.
//   inversion : field.h's fe_inv (Fermat, ~390 dependent products in one lane) against fe_inv_gcd (divsteps, one lane) and co_pow_words (Fermat, one row).
// Each variant runs as `chains` independent chains (1: the latency of one chain with the GPU to itself; 136: the shape of a bucket reduction's last step).
// Results are compared (canonical limbs) before anything is timed.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I zkp-ecdsa_amd/csrc tools/coop_bench.hip -o tools/coop_bench && tools/coop_bench
#include "coop.h"
#include <cstdio>
#include <cstring>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

// in/out: 36 words per chain (X, Y, T, Z limbs)
__global__ void __launch_bounds__(64) k_dbl_lane(const uint32_t* in, uint32_t* out, uint32_t chains, uint32_t n) {
    const uint32_t c = blockIdx.x * 64 + threadIdx.x;
    if (c >= chains) return;
    TomPt p;
    for (int l = 0; l < 9; l++) p.x.l[l] = in[c * 36 + l], p.y.l[l] = in[c * 36 + 9 + l], p.t.l[l] = in[c * 36 + 18 + l], p.z.l[l] = in[c * 36 + 27 + l];
#pragma unroll 1
    for (uint32_t i = 0; i < n; i++) p = tom_dbl(p);
    const auto x = fe_canon(p.x), y = fe_canon(p.y), t = fe_canon(p.t), z = fe_canon(p.z);
    for (int l = 0; l < 9; l++) out[c * 36 + l] = x.l[l], out[c * 36 + 9 + l] = y.l[l], out[c * 36 + 18 + l] = t.l[l], out[c * 36 + 27 + l] = z.l[l];
}
// one wave per chain
__global__ void __launch_bounds__(64) k_dbl_coop(const uint32_t* in, uint32_t* out, uint32_t chains, uint32_t n) {
    const uint32_t c = blockIdx.x;
    const CoU32 mj = co_limbs(ModT::mod);
    CoTom p;
    p.v = co_load4<ModT, 2>(in + c * 36, in + c * 36 + 9, in + c * 36 + 18, in + c * 36 + 27);
#pragma unroll 1
    for (uint32_t i = 0; i < n; i++) p = co_tom_dbl(p, mj);
    co_store4(p.v, out + c * 36, out + c * 36 + 9, out + c * 36 + 18, out + c * 36 + 27);
}
__global__ void k_canon36(uint32_t* io, uint32_t chains) {   // limbs < 2 t -> canonical
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= chains * 4) return;
    Ft2 v;
    for (int l = 0; l < 9; l++) v.l[l] = io[e * 9 + l];
    const auto c = fe_canon(v);
    for (int l = 0; l < 9; l++) io[e * 9 + l] = c.l[l];
}
// inversions: x <- 1/x + 1, reps times, 9 words per chain
template <class M, int KIND>
__global__ void __launch_bounds__(64) k_inv_lane(const uint32_t* in, uint32_t* out, uint32_t chains, uint32_t reps) {
    const uint32_t c = blockIdx.x * 64 + threadIdx.x;
    if (c >= chains) return;
    Fe<M, 2> x;
    for (int l = 0; l < 9; l++) x.l[l] = in[c * 9 + l];
#pragma unroll 1
    for (uint32_t i = 0; i < reps; i++) {
        Fe<M, 2> y = KIND == 0 ? fe_inv_fermat<M>(x) : fe_inv_gcd<M>(x);
        x = fe_reduce(y + fe_one_mont<M>());
    }
    const auto r = fe_canon(x);
    for (int l = 0; l < 9; l++) out[c * 9 + l] = r.l[l];
}
template <class M>
__global__ void __launch_bounds__(64) k_inv_coop(const uint32_t* in, uint32_t* out, uint32_t chains, uint32_t reps) {
    const uint32_t c = blockIdx.x;
    const CoU32 mj = co_limbs(M::mod);
    CoFe<M, 2> x = co_load4<M, 2>(in + c * 9, in + c * 9, in + c * 9, in + c * 9);
    CoFe<M, 1> one;
    one.v = co_limbs(M::one);
#pragma unroll 1
    for (uint32_t i = 0; i < reps; i++) {
        const auto y = co_pow_words<M>(x, M::exp_m2, mj);
        x = co_mul(co_add(y, one), one, mj);
    }
    co_store4(x, out + c * 9, nullptr, nullptr, nullptr);
}
__global__ void k_canon9(uint32_t* io, uint32_t chains) {
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= chains) return;
    Ft2 v;
    for (int l = 0; l < 9; l++) v.l[l] = io[e * 9 + l];
    const auto c = fe_canon(v);
    for (int l = 0; l < 9; l++) io[e * 9 + l] = c.l[l];
}

// pure product chain: x <- x * x, n times (timing only; the cooperative kernel squares four elements at once)
__global__ void __launch_bounds__(64) k_sqr_lane(const uint32_t* in, uint32_t* out, uint32_t n) {
    if (threadIdx.x) return;
    Ft2 x;
    for (int l = 0; l < 9; l++) x.l[l] = in[l];
#pragma unroll 1
    for (uint32_t i = 0; i < n; i++) x = x * x;
    for (int l = 0; l < 9; l++) out[l] = x.l[l];
}
__global__ void __launch_bounds__(64) k_sqr_coop(const uint32_t* in, uint32_t* out, uint32_t n) {
    const CoU32 mj = co_limbs(ModT::mod);
    auto x = co_load4<ModT, 2>(in, in + 9, in + 18, in + 27);
#pragma unroll 1
    for (uint32_t i = 0; i < n; i++) x = co_mul(x, x, mj);
    co_store4(x, out, out + 9, out + 18, out + 27);
}
// P-256: n Jacobian doublings (k_rtab_base's chain), n complete additions acc <- acc + P
__global__ void __launch_bounds__(64) k_pj_lane(const uint32_t* in, uint32_t* out, uint32_t n) {
    if (threadIdx.x) return;
    P256Jac p;
    for (int l = 0; l < 9; l++) p.x.l[l] = in[l], p.y.l[l] = in[9 + l], p.z.l[l] = ModQ::one[l];
#pragma unroll 1
    for (uint32_t i = 0; i < n; i++) p = p256_jdbl(p);
    for (int l = 0; l < 9; l++) out[l] = p.x.l[l], out[9 + l] = p.y.l[l], out[18 + l] = p.z.l[l];
}
__global__ void __launch_bounds__(64) k_pj_coop(const uint32_t* in, uint32_t* out, uint32_t n) {
    const CoU32 mj = co_limbs(ModQ::mod);
    CoP256J p;
    p.v = co_load4<ModQ, 10>(in, in + 9, ModQ::one, ModQ::one);
#pragma unroll 1
    for (uint32_t i = 0; i < n; i++) p = co_p256_jdbl(p, mj);
    co_store4(p.v, out, out + 9, out + 18, out + 27);
}
__global__ void __launch_bounds__(64) k_pa_lane(const uint32_t* in, uint32_t* out, uint32_t n, int dbl) {
    if (threadIdx.x) return;
    P256Pt p, acc;
    for (int l = 0; l < 9; l++) p.x.l[l] = in[l], p.y.l[l] = in[9 + l], p.z.l[l] = ModQ::one[l];
    acc = p;
#pragma unroll 1
    for (uint32_t i = 0; i < n; i++) acc = dbl ? p256_dbl(acc) : p256_add(acc, p);
    const auto x = fe_canon(fe_reduce(acc.x)), y = fe_canon(fe_reduce(acc.y)), z = fe_canon(fe_reduce(acc.z));
    for (int l = 0; l < 9; l++) out[l] = x.l[l], out[9 + l] = y.l[l], out[18 + l] = z.l[l];
}
__global__ void __launch_bounds__(64) k_pa_coop(const uint32_t* in, uint32_t* out, uint32_t n, int dbl) {
    const CoU32 mj = co_limbs(ModQ::mod);
    CoP256 p, acc;
    p.v = co_load4<ModQ, 8>(in, in + 9, ModQ::one, ModQ::one);
    acc = p;
#pragma unroll 1
    for (uint32_t i = 0; i < n; i++) acc = dbl ? co_p256_dbl(acc, mj) : co_p256_add(acc, p, mj);
    CoFe<ModQ, 2> red = co_mul(acc.v, co_const<ModQ>(ModQ::one), mj);
    co_store4(red, out, out + 9, out + 18, nullptr);
}
__global__ void k_canon_q(uint32_t* io, uint32_t n) {
    const uint32_t e = threadIdx.x;
    if (e >= n) return;
    Fq2 v;
    for (int l = 0; l < 9; l++) v.l[l] = io[e * 9 + l];
    const auto c = fe_canon(v);
    for (int l = 0; l < 9; l++) io[e * 9 + l] = c.l[l];
}

template <class F>
static float time_ms(F launch, int reps) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    launch();
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < reps; r++) {
        hipEventRecord(e0);
        launch();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    hipEventDestroy(e0), hipEventDestroy(e1);
    return best;
}

int main() {
    const uint32_t MAXC = 4096, N = 240;
    std::vector<uint32_t> h(MAXC * 36);
    uint64_t s = 88172645463325252ull;
    for (auto& w : h) {
        s ^= s << 13, s ^= s >> 7, s ^= s << 17;
        w = (uint32_t)s & LIMB_MASK;
    }
    for (uint32_t c = 0; c < MAXC * 4; c++) h[c * 9 + 8] &= 0xffff;   // < 2^256 < t
    uint32_t *in, *o1, *o2, *o3;
    CHECK(hipMalloc(&in, MAXC * 36 * 4)); CHECK(hipMalloc(&o1, MAXC * 36 * 4)); CHECK(hipMalloc(&o2, MAXC * 36 * 4)); CHECK(hipMalloc(&o3, MAXC * 36 * 4));
    CHECK(hipMemcpy(in, h.data(), MAXC * 36 * 4, hipMemcpyHostToDevice));
    std::vector<uint32_t> r1(MAXC * 36), r2(MAXC * 36), r3(MAXC * 36);
    // ---- correctness
    {
        const uint32_t C = 64;
        hipLaunchKernelGGL(k_dbl_lane, dim3(1), dim3(64), 0, 0, in, o1, C, 37u);
        hipLaunchKernelGGL(k_dbl_coop, dim3(C), dim3(64), 0, 0, in, o2, C, 37u);
        hipLaunchKernelGGL(k_canon36, dim3(1), dim3(256), 0, 0, o2, C);
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(r1.data(), o1, C * 36 * 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(r2.data(), o2, C * 36 * 4, hipMemcpyDeviceToHost));
        printf("dbl chain (37 doublings, 64 chains): cooperative == one-lane: %s\n", memcmp(r1.data(), r2.data(), C * 36 * 4) ? "NO" : "yes");
        hipLaunchKernelGGL((k_inv_lane<ModT, 0>), dim3(1), dim3(64), 0, 0, in, o1, C, 3u);
        hipLaunchKernelGGL((k_inv_lane<ModT, 1>), dim3(1), dim3(64), 0, 0, in, o2, C, 3u);
        hipLaunchKernelGGL((k_inv_coop<ModT>), dim3(C), dim3(64), 0, 0, in, o3, C, 3u);
        hipLaunchKernelGGL(k_canon9, dim3(1), dim3(64), 0, 0, o3, C);
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(r1.data(), o1, C * 9 * 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(r2.data(), o2, C * 9 * 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(r3.data(), o3, C * 9 * 4, hipMemcpyDeviceToHost));
        printf("inversion mod t (3 in a row, 64 chains): divsteps == Fermat: %s, cooperative Fermat == Fermat: %s\n", memcmp(r1.data(), r2.data(), C * 9 * 4) ? "NO" : "yes",
               memcmp(r1.data(), r3.data(), C * 9 * 4) ? "NO" : "yes");
        hipLaunchKernelGGL((k_inv_lane<ModQ, 0>), dim3(1), dim3(64), 0, 0, in, o1, C, 2u);
        hipLaunchKernelGGL((k_inv_lane<ModQ, 1>), dim3(1), dim3(64), 0, 0, in, o2, C, 2u);
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(r1.data(), o1, C * 9 * 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(r2.data(), o2, C * 9 * 4, hipMemcpyDeviceToHost));
        printf("inversion mod q: divsteps == Fermat: %s\n", memcmp(r1.data(), r2.data(), C * 9 * 4) ? "NO" : "yes");
    }
    // ---- timing
    for (uint32_t chains : {1u, 8u, 136u, 1024u, 4096u}) {
        const float a = time_ms([&] { hipLaunchKernelGGL(k_dbl_lane, dim3((chains + 63) / 64), dim3(64), 0, 0, in, o1, chains, N); }, 5);
        const float b = time_ms([&] { hipLaunchKernelGGL(k_dbl_coop, dim3(chains), dim3(64), 0, 0, in, o2, chains, N); }, 5);
        printf("tom_dbl x %u, %4u chains: one lane %8.1f us (%6.2f us per doubling)   cooperative wave %8.1f us (%6.2f us per doubling)   ratio %.2f\n", N, chains, a * 1e3, a * 1e3 / N,
               b * 1e3, b * 1e3 / N, a / b);
    }
    for (uint32_t chains : {1u, 8u, 256u, 4096u}) {
        const uint32_t reps = 4;
        const float a = time_ms([&] { hipLaunchKernelGGL((k_inv_lane<ModT, 0>), dim3((chains + 63) / 64), dim3(64), 0, 0, in, o1, chains, reps); }, 5);
        const float b = time_ms([&] { hipLaunchKernelGGL((k_inv_lane<ModT, 1>), dim3((chains + 63) / 64), dim3(64), 0, 0, in, o2, chains, reps); }, 5);
        const float c = time_ms([&] { hipLaunchKernelGGL((k_inv_coop<ModT>), dim3(chains), dim3(64), 0, 0, in, o3, chains, reps); }, 5);
        printf("inversion mod t, %4u chains: Fermat one lane %8.1f us   divsteps one lane %8.1f us (ratio %.2f)   Fermat cooperative row %8.1f us (ratio %.2f)\n", chains, a * 1e3 / reps,
               b * 1e3 / reps, a / b, c * 1e3 / reps, a / c);
    }
    {
        const uint32_t reps = 4;
        const float a = time_ms([&] { hipLaunchKernelGGL((k_inv_lane<ModQ, 0>), dim3(1), dim3(64), 0, 0, in, o1, 1u, reps); }, 5);
        const float b = time_ms([&] { hipLaunchKernelGGL((k_inv_lane<ModQ, 1>), dim3(1), dim3(64), 0, 0, in, o2, 1u, reps); }, 5);
        printf("inversion mod q,    1 chain : Fermat one lane %8.1f us   divsteps one lane %8.1f us (ratio %.2f)\n", a * 1e3 / reps, b * 1e3 / reps, a / b);
    }
    {
        const uint32_t n = 1000;
        const float a = time_ms([&] { hipLaunchKernelGGL(k_sqr_lane, dim3(1), dim3(64), 0, 0, in, o1, n); }, 5);
        const float b = time_ms([&] { hipLaunchKernelGGL(k_sqr_coop, dim3(1), dim3(64), 0, 0, in, o2, n); }, 5);
        printf("Montgomery product mod t, a chain of %u: one lane %.3f us per product, cooperative row %.3f us per product (four at once), ratio %.2f\n", n, a * 1e3 / n, b * 1e3 / n, a / b);
    }
    {
        // values, not only times: the same points through both forms (P-256 inputs need not be on the curve for the formulas to agree)
        for (int dbl = 0; dbl < 2; dbl++) {
            hipLaunchKernelGGL(k_pa_lane, dim3(1), dim3(64), 0, 0, in, o1, 21u, dbl);
            hipLaunchKernelGGL(k_pa_coop, dim3(1), dim3(64), 0, 0, in, o2, 21u, dbl);
            hipLaunchKernelGGL(k_canon_q, dim3(1), dim3(64), 0, 0, o2, 3u);
            CHECK(hipDeviceSynchronize());
            CHECK(hipMemcpy(r1.data(), o1, 27 * 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(r2.data(), o2, 27 * 4, hipMemcpyDeviceToHost));
            printf("P-256 complete %s x 21: cooperative == one-lane: %s\n", dbl ? "doubling" : "addition", memcmp(r1.data(), r2.data(), 27 * 4) ? "NO" : "yes");
        }
        const uint32_t n = 256;
        float a = time_ms([&] { hipLaunchKernelGGL(k_pj_lane, dim3(1), dim3(64), 0, 0, in, o1, n); }, 5);
        float b = time_ms([&] { hipLaunchKernelGGL(k_pj_coop, dim3(1), dim3(64), 0, 0, in, o2, n); }, 5);
        printf("P-256 Jacobian doubling x %u: one lane (p256_jdbl, 8 products) %.1f us, cooperative wave (ZZ form, 9 products in 3 passes) %.1f us, ratio %.2f\n", n, a * 1e3, b * 1e3, a / b);
        for (int dbl = 0; dbl < 2; dbl++) {
            a = time_ms([&] { hipLaunchKernelGGL(k_pa_lane, dim3(1), dim3(64), 0, 0, in, o1, n, dbl); }, 5);
            b = time_ms([&] { hipLaunchKernelGGL(k_pa_coop, dim3(1), dim3(64), 0, 0, in, o2, n, dbl); }, 5);
            printf("P-256 complete %s x %u: one lane %.2f us each, cooperative wave (4 passes) %.2f us each, ratio %.2f\n", dbl ? "doubling" : "addition", n, a * 1e3 / n, b * 1e3 / n, a / b);
        }
    }
    return 0;
}
