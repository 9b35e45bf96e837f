// Microbenchmark: issue rate of the integer instructions the big-number kernels are built from (gfx950).
// Each kernel keeps NACC independent 64-bit accumulators per lane; the only dependency is through the
// accumulator (exactly the pattern of the product-scanning Montgomery multiplication in csrc/field.h).
// The v_mad_u64_u32 figure at high ILP is the denominator of the VALU roofline in bench.py.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int ITER = 2048;

template <int KIND, int NACC> __global__ void __launch_bounds__(256) k(uint64_t* out, uint32_t seed) {
    uint32_t a = seed + threadIdx.x, b = seed * 3 + blockIdx.x;
    uint64_t acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; i++) acc[i] = i + a;
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
#pragma unroll
            for (int j = 0; j < NACC; j++) {
                if (KIND == 0) acc[j] = (uint64_t)a * b + acc[j];                       // v_mad_u64_u32
                if (KIND == 1) acc[j] = (uint32_t)((uint32_t)acc[j] * a);               // v_mul_lo_u32
                if (KIND == 2) acc[j] = (acc[j] >> 30) + b;                             // 64-bit shift + add
                if (KIND == 3) acc[j] = (uint32_t)acc[j] + a;                           // v_add_u32
            }
            asm volatile("" : "+v"(a), "+v"(b));
        }
    }
    uint64_t r = 0;
#pragma unroll
    for (int i = 0; i < NACC; i++) r += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
// SUSTAINED rate: the same kernel back to back for ~`ms_target` milliseconds (the figures above come from bursts of a few
// milliseconds, i.e. at boost clock; the engine's kernels run for 10-60 ms each, hundreds of ms per step)
template <int KIND, int NACC> int run_sustained(const char* name, uint64_t* out, int blocks_per_cu, float ms_target) {
    int blocks = 256 * blocks_per_cu;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k<KIND, NACC>), dim3(blocks), dim3(256), 0, 0, out, 12345u);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<KIND, NACC>), dim3(blocks), dim3(256), 0, 0, out, 1u);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float one; CHECK(hipEventElapsedTime(&one, e0, e1));
    int reps = (int)(ms_target / one) + 1;
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < reps; r++) hipLaunchKernelGGL((k<KIND, NACC>), dim3(blocks), dim3(256), 0, 0, out, 12345u + r);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    // the last tenth alone: the clock has settled by then
    int tail = reps / 10 + 1;
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < tail; r++) hipLaunchKernelGGL((k<KIND, NACC>), dim3(blocks), dim3(256), 0, 0, out, 777u + r);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms2; CHECK(hipEventElapsedTime(&ms2, e0, e1));
    double per = (double)blocks * 256 * (double)ITER * 4 * NACC;
    printf("%-22s NACC=%2d waves/SIMD=%d  SUSTAINED %7.1f ms (%d launches)  %8.2f T lane-ops/s; next %d launches %8.2f T lane-ops/s\n", name, NACC, blocks_per_cu,
           ms, reps, per * reps / (ms * 1e-3) / 1e12, tail, per * tail / (ms2 * 1e-3) / 1e12);
    return 0;
}
template <int KIND, int NACC> int run(const char* name, uint64_t* out, int blocks_per_cu) {
    int blocks = 256 * blocks_per_cu;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k<KIND, NACC>), dim3(blocks), dim3(256), 0, 0, out, 12345u);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < 5; r++) hipLaunchKernelGGL((k<KIND, NACC>), dim3(blocks), dim3(256), 0, 0, out, 12345u + r);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    double ops = 5.0 * blocks * 256 * (double)ITER * 4 * NACC;
    printf("%-22s NACC=%2d waves/SIMD=%d  %8.3f ms  %8.2f T lane-ops/s  => %.2f lane-ops/clk/SIMD @2.4GHz (%.1f cyc/wave-instr)\n", name, NACC, blocks_per_cu,
           ms / 5, ops / (ms * 1e-3) / 1e12, ops / (ms * 1e-3) / (256 * 4 * 2.4e9), 64.0 / (ops / (ms * 1e-3) / (256 * 4 * 2.4e9)));
    return 0;
}
int main() {
    uint64_t* out; CHECK(hipMalloc(&out, 256 * 8 * 256 * 8));
    run<0, 1>("v_mad_u64_u32", out, 1);
    run<0, 1>("v_mad_u64_u32", out, 2);
    run<0, 1>("v_mad_u64_u32", out, 4);
    run<0, 1>("v_mad_u64_u32", out, 8);
    run<0, 4>("v_mad_u64_u32", out, 2);
    // independent accumulator chains per SIMD = NACC x waves/SIMD: the dependent-issue behaviour the lock-step products of
    // k_tom_commit (4 chains per wave, 2 waves per SIMD) run into
    run<0, 2>("v_mad_u64_u32", out, 2);
    run<0, 8>("v_mad_u64_u32", out, 2);
    run<0, 16>("v_mad_u64_u32", out, 2);
    run<0, 2>("v_mad_u64_u32", out, 4);
    run<0, 4>("v_mad_u64_u32", out, 4);
    run<0, 8>("v_mad_u64_u32", out, 4);
    run<0, 8>("v_mad_u64_u32", out, 8);
    run<0, 16>("v_mad_u64_u32", out, 8);
    run<1, 8>("v_mul_lo_u32", out, 8);
    run<2, 8>("shr64+add64", out, 8);
    run<3, 8>("v_add_u32", out, 8);
    run_sustained<0, 16>("v_mad_u64_u32", out, 8, 300.f);
    run_sustained<0, 16>("v_mad_u64_u32", out, 8, 2000.f);
    run_sustained<0, 8>("v_mad_u64_u32", out, 2, 1000.f);
    run<0, 16>("v_mad_u64_u32", out, 8);
    return 0;
}
