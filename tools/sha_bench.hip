// Microbenchmark (gfx950): the message -> schedule -> rounds kernels of k_hash.hip (launch_sha_msgs) at the shapes the engine uses them:
// (count, blocks) = (1, 251) the Exp challenge of one proof, (1, 68) its membership challenge, (120, 10) the PointAdd challenges of one verification.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I zkp-ecdsa_amd/csrc tools/sha_bench.hip -o tools/sha_bench -L zkp-ecdsa_amd/lib -lzkattest_hip -Wl,-rpath,$PWD/zkp-ecdsa_amd/lib
#include "engine.h"
#include <cstdio>
#include <cstring>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
int main() {
    hipStream_t s;
    CHECK(hipStreamCreate(&s));
    const size_t cap = 8192 * 10;
    uint8_t *msg, *nb;
    uint32_t *wk, *chal;
    CHECK(hipMalloc(&msg, cap * 64));
    CHECK(hipMalloc(&wk, cap * 256));
    CHECK(hipMalloc(&chal, 8192 * 16));
    CHECK(hipMalloc(&nb, 8192));
    CHECK(hipMemset(msg, 0x5a, cap * 64));
    std::vector<uint8_t> hnb(8192);
    for (int i = 0; i < 8192; i++) hnb[i] = (i % 6) < 4 ? 10 : 5;
    CHECK(hipMemcpy(nb, hnb.data(), 8192, hipMemcpyHostToDevice));
    Workspace W;
    memset(&W, 0, sizeof W);
    W.exph_msg = msg, W.exph_wk = wk, W.sec = 80;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const struct { uint32_t count, nblk; bool var; } shapes[] = {{1, 251, false}, {1, 68, false}, {1, 10, false}, {2, 10, false}, {8, 10, false}, {32, 10, false}, {33, 10, false}, {120, 10, false},
                                                                  {120, 10, true}, {480, 10, true}, {1920, 10, true}, {7680, 10, true}, {120, 68, false}};
    for (auto& sh : shapes) {
        for (int w = 0; w < 3; w++) launch_sha_msgs(s, W, sh.count, chal, sh.nblk, 3, sh.var ? nb : nullptr);
        CHECK(hipStreamSynchronize(s));
        const int reps = 50;
        CHECK(hipEventRecord(e0, s));
        for (int r = 0; r < reps; r++) launch_sha_msgs(s, W, sh.count, chal, sh.nblk, 3, sh.var ? nb : nullptr);
        CHECK(hipEventRecord(e1, s));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("count %5u  blocks %3u  %s  schedule + rounds %.1f us per launch pair\n", sh.count, sh.nblk, sh.var ? "var" : "   ", 1e3 * ms / reps);
    }
    // The same pair of kernels while OTHER streams hold work that waits for an event (what a call of a few proofs looks like to the command processor: its side
    // streams are handed their kernels early and wait for the main chain), and while another stream runs a long kernel of one wave.
    hipStream_t blocker, side[5];
    // (the side streams hash garbage in the same buffers; only the timing of the main stream's pair is read)
    CHECK(hipStreamCreateWithFlags(&blocker, hipStreamNonBlocking));
    for (auto& x : side) CHECK(hipStreamCreateWithFlags(&x, hipStreamNonBlocking));
    hipEvent_t gate;
    CHECK(hipEventCreateWithFlags(&gate, hipEventDisableTiming));
    for (int nside = 0; nside <= 5; nside++) {
        for (int longk = 0; longk < 2; longk++) {
            // the blocker stream hashes one message of 251 blocks 8 times (~3 ms in one wave), then records the gate
            for (int r = 0; r < 8; r++) launch_sha_msgs(blocker, W, 1, chal + 4096, 251, 3, nullptr);
            CHECK(hipEventRecord(gate, blocker));
            for (int i = 0; i < nside; i++) {
                CHECK(hipStreamWaitEvent(side[i], gate, 0));
                launch_sha_msgs(side[i], W, 1, chal + 8192 + 64 * i, 1, 3, nullptr);
            }
            CHECK(hipEventRecord(e0, s));
            const int reps = 20;
            for (int r = 0; r < reps; r++) launch_sha_msgs(s, W, 120, chal, 10, 3, nb);
            CHECK(hipEventRecord(e1, s));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            printf("beside a busy stream and %d streams waiting for its event: count 120 blocks 10 var  %.1f us per launch pair\n", nside, 1e3 * ms / reps);
            CHECK(hipDeviceSynchronize());
        }
    }
    // ... and beside streams that RUN long kernels of one wave each (nothing waits)
    for (int nbusy = 0; nbusy <= 5; nbusy++) {
        for (int i = 0; i < nbusy; i++)
            for (int r = 0; r < 6; r++) launch_sha_msgs(side[i], W, 1, chal + 8192 + 64 * i, 251, 3, nullptr);
        CHECK(hipEventRecord(e0, s));
        const int reps = 20;
        for (int r = 0; r < reps; r++) launch_sha_msgs(s, W, 120, chal, 10, 3, nb);
        CHECK(hipEventRecord(e1, s));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("beside %d streams that run one-wave kernels: count 120 blocks 10 var  %.1f us per launch pair\n", nbusy, 1e3 * ms / reps);
        CHECK(hipDeviceSynchronize());
    }
    // ... which stream is it?  Ten more streams, created in a row; the pair on stream a while stream b runs a one-wave kernel, for every (a, b)
    hipStream_t t[10];
    for (auto& x : t) CHECK(hipStreamCreateWithFlags(&x, hipStreamNonBlocking));
    for (auto& x : t) { launch_sha_msgs(x, W, 1, chal + 8192, 1, 3, nullptr); CHECK(hipStreamSynchronize(x)); }   // every stream has its hardware queue now
    printf("pair on stream a (row) beside a one-wave kernel on stream b (column), us per launch pair\n");
    for (int a = 0; a < 10; a++) {
        printf("a=%d:", a);
        for (int b = 0; b < 10; b++) {
            if (a == b) { printf("     -"); continue; }
            for (int r = 0; r < 3; r++) launch_sha_msgs(t[b], W, 1, chal + 8192, 251, 3, nullptr);
            CHECK(hipEventRecord(e0, t[a]));
            const int reps = 10;
            for (int r = 0; r < reps; r++) launch_sha_msgs(t[a], W, 120, chal, 10, 3, nb);
            CHECK(hipEventRecord(e1, t[a]));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            printf(" %5.1f", 1e3 * ms / reps);
            CHECK(hipDeviceSynchronize());
        }
        printf("\n");
    }
    return 0;
}
