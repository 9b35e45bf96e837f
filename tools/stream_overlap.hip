// Do kernels of two HIP streams overlap on this box?  (a) small grids that leave most of the GPU free, (b) full-GPU grids,
// each with and without timing events recorded around every launch (what the engine's Scope timers do).
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__global__ void spin(unsigned long long cycles, unsigned* out) {
    unsigned long long t0 = wall_clock64();
    unsigned acc = threadIdx.x;
    while (wall_clock64() - t0 < cycles) acc = acc * 1664525u + 1013904223u;
    if (acc == 12345u) out[0] = acc;
}
int main() {
    unsigned* d; CHECK(hipMalloc(&d, 64));
    hipStream_t s[2]; CHECK(hipStreamCreate(&s[0])); CHECK(hipStreamCreate(&s[1]));
    hipEvent_t ev[64]; for (int i = 0; i < 64; i++) CHECK(hipEventCreate(&ev[i]));
    const unsigned long long cyc = 100000000ull / 50;   // wall_clock64 ticks at 100 MHz: 20 ms
    for (int blocks : {64, 256 * 8, 256 * 64}) {
        for (int with_events = 0; with_events < 2; with_events++) {
            for (int nstreams = 1; nstreams <= 2; nstreams++) {
                CHECK(hipDeviceSynchronize());
                hipEvent_t t0, t1; CHECK(hipEventCreate(&t0)); CHECK(hipEventCreate(&t1));
                CHECK(hipEventRecord(t0, s[0]));
                int k = 0;
                for (int rep = 0; rep < 4; rep++)
                    for (int i = 0; i < nstreams; i++) {
                        if (with_events) hipEventRecord(ev[k++ % 64], s[i]);
                        hipLaunchKernelGGL(spin, dim3(blocks), dim3(256), 0, s[i], cyc / (blocks > 256 * 10 ? blocks / (256 * 10) : 1), d);
                        if (with_events) hipEventRecord(ev[k++ % 64], s[i]);
                    }
                CHECK(hipStreamSynchronize(s[0])); CHECK(hipStreamSynchronize(s[1]));
                CHECK(hipEventRecord(t1, s[0])); CHECK(hipEventSynchronize(t1));
                float ms; CHECK(hipEventElapsedTime(&ms, t0, t1));
                printf("blocks %6d  events %d  streams %d: %7.1f ms for 4 launches per stream\n", blocks, with_events, nstreams, ms);
            }
        }
    }
    // does a hipMemsetAsync / small async D2H in stream 1 wait for the kernels of stream 0?
    unsigned* d2; CHECK(hipMalloc(&d2, 1 << 20));
    unsigned* hp; CHECK(hipHostMalloc((void**)&hp, 4096, hipHostMallocDefault));
    for (int what = 0; what < 4; what++) {
        CHECK(hipDeviceSynchronize());
        hipEvent_t t0, t1, t2; CHECK(hipEventCreate(&t0)); CHECK(hipEventCreate(&t1)); CHECK(hipEventCreate(&t2));
        CHECK(hipEventRecord(t0, s[0]));
        for (int rep = 0; rep < 4; rep++) hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, s[0], cyc, d);   // 80 ms in stream 0
        CHECK(hipEventRecord(t1, s[1]));
        if (what == 1) CHECK(hipMemsetAsync(d2, 0, 32768, s[1]));
        if (what == 2) CHECK(hipMemcpyAsync(hp, d2, 64, hipMemcpyDeviceToHost, s[1]));
        if (what == 3) CHECK(hipMemsetAsync(d2, 0, 4, s[1]));
        hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, s[1], cyc / 20, d);                                 // 1 ms in stream 1
        CHECK(hipEventRecord(t2, s[1]));
        CHECK(hipStreamSynchronize(s[1]));
        float a, b; CHECK(hipEventElapsedTime(&a, t0, t1)); CHECK(hipEventElapsedTime(&b, t0, t2));
        CHECK(hipStreamSynchronize(s[0]));
        printf("stream 1 %s: its 1-ms kernel ended %.1f ms after stream 0 started 80 ms of small kernels\n",
               what == 0 ? "kernel only          " : what == 1 ? "memsetAsync 32 KB     " : what == 2 ? "memcpyAsync D2H pinned" : "memsetAsync 4 B       ", b);
    }
    // HIP maps streams onto a few hardware queues (GPU_MAX_HW_QUEUES, default 4): which pairs of 8 streams really overlap?
    hipStream_t st[8];
    for (int i = 0; i < 8; i++) CHECK(hipStreamCreateWithFlags(&st[i], i < 2 ? hipStreamDefault : hipStreamNonBlocking));
    for (int j = 1; j < 8; j++) {
        CHECK(hipDeviceSynchronize());
        hipEvent_t t0, t2; CHECK(hipEventCreate(&t0)); CHECK(hipEventCreate(&t2));
        CHECK(hipEventRecord(t0, st[0]));
        for (int rep = 0; rep < 4; rep++) hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, st[0], cyc, d);
        hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, st[j], cyc / 20, d);
        CHECK(hipEventRecord(t2, st[j]));
        CHECK(hipStreamSynchronize(st[j]));
        float b; CHECK(hipEventElapsedTime(&b, t0, t2));
        CHECK(hipStreamSynchronize(st[0]));
        printf("8 streams: 1-ms kernel in stream %d ended %.1f ms after stream 0 started 80 ms of kernels%s\n", j, b, b > 40 ? "   <-- SAME hardware queue" : "");
    }
    return 0;
}
