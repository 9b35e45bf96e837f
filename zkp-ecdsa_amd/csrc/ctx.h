// Shared host-side context of libzkattest_hip.so (api.hip: prover pipeline; api_verify.hip: verifier pipeline).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include "engine.h"

struct TimerRec {
    const char* name;
    hipEvent_t e0, e1;
};
struct zk_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    // params
    DevParams P{};
    bool params_set = false;
    uint32_t* tom_tab_gen = nullptr;  // 8-bit comb table of the Tom generator (synthetic params only)
    uint32_t tom_bits = TOM_DEFAULT_BITS;  // comb width requested for g, h (zk_ctx_set_comb_bits)
    uint32_t tab_bits_alloc = 0;      // width the allocated P.tom_tab_g/h were sized for (0 = not allocated)
    size_t scratch_words = 0;
    uint32_t* tab_scratch = nullptr;
    int32_t* d_flag = nullptr;
    // ring
    uint32_t* ring_mem = nullptr;
    uint32_t* gk_etab = nullptr;   // per-ring table of the GK block transform (k_gk.hip); nullptr for small / huge rings
    bool gk_table = true;          // ZKATTEST_GK_TABLE=0 disables it (plain fold for every ring)
    uint64_t N = 0, nkeys = 0;
    uint32_t n = 0;
    // workspace
    uint32_t chunk = 4096;
    void* arena = nullptr;
    size_t arena_bytes = 0;
    uint32_t ws_C = 0, ws_sec = 0, ws_n = 0;
    Workspace W{};
    Soa gk_am{};
    uint32_t* d_totals = nullptr;
    // second pipeline lane (alternate chunks run on their own stream + workspace so that the low-occupancy front-end
    // kernels of chunk k+1 overlap the heavy phases of chunk k)
    hipStream_t stream2 = nullptr;
    Workspace W2{};
    Soa gk_am2{};
    uint32_t* d_totals2 = nullptr;
    void* arena2 = nullptr;
    size_t arena2_bytes = 0;
    bool lane2_ready = false;
    uint32_t lanes = 2;
    // verifier workspace
    VWork V{};
    void* varena = nullptr;
    size_t varena_bytes = 0;
    uint32_t vs_C = 0, vs_sec = 0, vs_n = 0;
    Soa v_res{}, v_res2{};
    MsmBuf M{}, M2{};         // batched Tom check buffers (k_msm.hip), carved with V / V2
    uint32_t verify_batch_min = 256;   // zk_ctx_set_batch_verify: chunks of at least this many proofs get the batched check (0 = never)
    VWork V2{};               // second verifier lane
    void* varena2 = nullptr;
    size_t varena2_bytes = 0;
    Soa v2_res{}, v2_res2{};
    bool vlane2_ready = false;
    // host-buffer entry points: DMA stream for page-locked caller buffers (zk_host_alloc), one event per lane
    hipStream_t copy_stream = nullptr;
    hipEvent_t copy_ev[2] = {nullptr, nullptr};
    void* io_buf = nullptr;        // device staging of the proof bytes for the host-pointer entry points (grow-only: a
    size_t io_bytes = 0;           // multi-GB hipMalloc/hipFree per call costs as much as the transfer itself)
    void* in_buf = nullptr;        // device copies of the small per-proof arrays of the host-pointer entry points (inputs,
    size_t in_bytes = 0;           // offsets, statuses, verdicts), grow-only for the same reason
    uint32_t host_taper = 1;       // host-pointer calls on page-locked buffers: tapered chunk plan (zk_ctx_set_host_taper)
    // timing
    std::vector<TimerRec> trecs;
    std::vector<hipEvent_t> epool;
    size_t eused = 0;
    std::vector<std::pair<const char*, float>> last_timing;
    float last_total_ms = 0;
};

#define HIPCHK(ctx, x)                                                                                      \
    do {                                                                                                    \
        hipError_t e_ = (x);                                                                                \
        if (e_ != hipSuccess) {                                                                             \
            char buf_[256];                                                                                 \
            snprintf(buf_, sizeof buf_, "%s failed: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
            (ctx)->err = buf_;                                                                              \
            return ZK_E_DEVICE;                                                                             \
        }                                                                                                   \
    } while (0)

static inline hipEvent_t get_event(zk_ctx* c) {
    if (c->eused == c->epool.size()) {
        hipEvent_t e;
        hipEventCreate(&e);
        c->epool.push_back(e);
    }
    return c->epool[c->eused++];
}
struct Scope {
    zk_ctx* c;
    TimerRec r;
    hipStream_t st;
    Scope(zk_ctx* c_, const char* name, hipStream_t s_ = nullptr) : c(c_), st(s_ ? s_ : c_->stream) {
        r.name = name, r.e0 = get_event(c), r.e1 = get_event(c);
        hipEventRecord(r.e0, st);
    }
    ~Scope() {
        hipEventRecord(r.e1, st);
        c->trecs.push_back(r);
    }
};
static inline void timing_begin(zk_ctx* c) { c->trecs.clear(), c->eused = 0; }
static inline void timing_end(zk_ctx* c) {
    c->last_timing.clear();
    c->last_total_ms = 0;
    for (auto& r : c->trecs) {
        float ms = 0;
        hipEventElapsedTime(&ms, r.e0, r.e1);
        bool found = false;
        for (auto& p : c->last_timing)
            if (p.first == r.name) p.second += ms, found = true;
        if (!found) c->last_timing.push_back({r.name, ms});
        c->last_total_ms += ms;
    }
}

// device allocation released on every exit path of an entry point
struct DevBuf {
    void* p = nullptr;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { hipFree(p); }
    template <class T>
    T* as() const { return (T*)p; }
};
zk_status ensure_workspace(zk_ctx* c, uint32_t C, bool second_lane = false);
zk_status ensure_in_buf(zk_ctx* c, size_t bytes);  // api.hip: c->in_buf of at least `bytes`

// One pipeline pass = one chunk of consecutive proofs.  Device-pointer calls use uniform chunks of C proofs.  Host-pointer
// calls on page-locked buffers move ~169 KB per proof across PCIe on a copy stream under the kernels of the neighbouring
// chunks, and the transfer of a step takes almost as long as its kernels (11 GB at ~55 GB/s against ~250 ms).  A prove call
// then ends at  max_k ( kernels of chunks 0..k  +  transfers of chunks k..last ),  so a chunk of c proofs with R proofs
// behind it costs about 0.9 c - 0.1 R proof-times of exposed transfer: big chunks are only harmless early, and with two lanes
// two chunks finish together.  The tapered plan: a half-sized first chunk (staggers the two lanes), chunks of at most C while
// plenty of work remains, then chunks of a sixth of what is left down to ZK_TAPER_MIN proofs.  The verifier's mirror image
// (kernels of the last chunks after the last H2D) is served by the same plan.  The bytes of a proof do not depend on the plan
// (tests/test_gpu_prove.py::test_lanes_and_chunking_do_not_change_the_bytes).
struct ChunkPlan {
    uint64_t first;
    uint32_t cnt;
};
#define ZK_TAPER_MIN 2048u
static inline std::vector<ChunkPlan> make_chunk_plan(uint64_t B, uint32_t C, bool taper) {
    std::vector<ChunkPlan> plan;
    uint64_t f = 0;
    bool first = true;
    while (f < B) {
        uint64_t left = B - f, c = C;
        if (taper && C > ZK_TAPER_MIN) {
            c = std::min<uint64_t>(C, std::max<uint64_t>(ZK_TAPER_MIN, (left / 6) & ~(uint64_t)255));
            if (first) c = std::max<uint64_t>(ZK_TAPER_MIN, c / 2);
        }
        first = false;
        c = std::min(c, left);
        plan.push_back({f, (uint32_t)c});
        f += c;
    }
    return plan;
}
zk_status ensure_io_buf(zk_ctx* c, size_t bytes);  // api.hip: c->io_buf of at least `bytes`
bool host_ptr_is_pinned(const void* p);     // api.hip: page-locked (zk_host_alloc / hipHostMalloc / hipHostRegister) host memory?
zk_status ensure_copy_stream(zk_ctx* c);    // api.hip: c->copy_stream and c->copy_ev

struct Carver {
    uint8_t* base;
    size_t off = 0;
    explicit Carver(uint8_t* b) : base(b) {}
    void* take(size_t bytes) {
        off = (off + 255) & ~(size_t)255;
        void* p = base ? base + off : nullptr;
        off += bytes;
        return p;
    }
    Soa soa(size_t elems) { return Soa{(uint32_t*)take(elems * 36), (uint32_t)elems}; }
    Soa3 soa3(size_t elems) { return Soa3{soa(elems), soa(elems), soa(elems)}; }
    TomList list(size_t cap) {
        TomList L;
        L.v = soa(cap), L.r = soa(cap), L.proj = soa3(cap), L.ax = soa(cap), L.ay = soa(cap), L.cap = (uint32_t)cap;
        return L;
    }
};
