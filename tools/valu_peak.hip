// Microbenchmark: issue rate of the integer/FP instructions a big-number kernel can be built from (gfx950).
// Prints ops/s per instruction kind; the v_mad_u64_u32 figure is the denominator of the VALU roofline.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int ITER = 4096, UNROLL = 16;

template <int KIND> __global__ void __launch_bounds__(256) k(uint64_t* out, uint32_t seed) {
    uint32_t a = seed + threadIdx.x, b = seed * 3 + blockIdx.x;
    uint64_t acc[8];
    double d[8];
#pragma unroll
    for (int i = 0; i < 8; i++) acc[i] = i + a, d[i] = 1.0 + i + a;
    double da = 1.0000001 + a * 1e-9, db = 1e-9 * b;
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            int j = u & 7;
            if (KIND == 0) acc[j] = (uint64_t)a * (uint32_t)(acc[j] >> 7) + acc[j];          // v_mad_u64_u32 (dependent per chain, 8 chains)
            if (KIND == 1) acc[j] = (uint32_t)acc[j] * a + b;                                 // v_mul_lo_u32 (+add)
            if (KIND == 2) acc[j] = __umulhi((uint32_t)acc[j], a) + b;                        // v_mul_hi_u32
            if (KIND == 3) acc[j] = (((uint32_t)acc[j] & 0xffffff) * (a & 0xffffff)) + b;        // v_mul_u32_u24 / mad_u32_u24
            if (KIND == 4) d[j] = __builtin_fma(d[j], da, db);                                // v_fma_f64
            if (KIND == 5) acc[j] = (uint32_t)acc[j] + a + (uint32_t)(acc[j] >> 32);          // plain adds
            if (KIND == 6) { uint32_t lo = (uint32_t)acc[j], hi = (uint32_t)(acc[j] >> 32); uint32_t s = lo + a; uint32_t c = s < lo; acc[j] = ((uint64_t)(hi + b + c) << 32) | s; } // add/addc pair
        }
    }
    uint64_t r = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) r += acc[i] + (uint64_t)d[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int KIND> int run(const char* name, uint64_t* out, double per_iter_ops) {
    int blocks = 256 * 8;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, 12345u);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < 5; r++) hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, 12345u + r);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    double ops = 5.0 * blocks * 256 * (double)ITER * UNROLL * per_iter_ops;
    printf("%-28s %8.3f ms  %10.2f Gop/s (lane-ops)  => %.2f lane-ops/clk/SIMD @2.4GHz\n", name, ms / 5, ops / (ms * 1e-3) / 1e9, ops / (ms * 1e-3) / (256 * 4 * 2.4e9));
    return 0;
}
int main() {
    uint64_t* out; CHECK(hipMalloc(&out, 256 * 8 * 256 * 8));
    run<0>("v_mad_u64_u32", out, 1);
    run<1>("v_mul_lo_u32(+add)", out, 1);
    run<2>("v_mul_hi_u32(+add)", out, 1);
    run<3>("v_mul_u32_u24(+add)", out, 1);
    run<4>("v_fma_f64", out, 1);
    run<5>("v_add x2", out, 1);
    run<6>("add/addc pair", out, 1);
    return 0;
}
