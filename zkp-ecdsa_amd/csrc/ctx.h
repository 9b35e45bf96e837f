// Shared host-side context of libzkattest_hip.so (api.hip: prover pipeline; api_verify.hip: verifier pipeline).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <functional>
#include <string>
#include <vector>
#include "engine.h"

struct TimerRec {
    const char* name;
    hipEvent_t e0, e1;
};
#define ZK_MAX_LANES 4
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
    uint32_t* ktab = nullptr;      // per-key tables of the ring (k_ktab.hip): 264 KB per key (KTAB_KEY_WORDS), rings of up to 2^KTAB_MAXN keys
    uint8_t* ktab_ok = nullptr;    // [N] which ring values are x-coordinates and own a table
    bool key_tables = true;        // ZKATTEST_KEYTAB=0 / zk_ctx_set_key_tables(ctx, 0): per-proof tables of R for every proof (read at zk_ctx_set_ring)
    int8_t* gk_kdig = nullptr;     // the ring as int8 digit fragments (k_gk_mfma.hip), built with table E for rings of >= 2^12 keys
    int8_t* gk_edig = nullptr;     // table E's coefficient classes 2..6 as digit fragments: the prover's matrix-pipe table path (k_gk_mfma.hip)
    bool gk_mfma_prove = true;     // build gk_edig with the ring (ZKATTEST_GK_MFMA_PROVE; read at zk_ctx_set_ring)
    bool gk_mfma = true;           // verifier's ring fold on the matrix pipe where gk_kdig exists (zk_ctx_set_ring_fold, ZKATTEST_GK_MFMA)
    uint64_t N = 0, nkeys = 0;
    uint32_t n = 0;
    uint32_t* ring_digest = nullptr;   // [8] SHA-256 words of the padded ring (hardened mode), computed by every zk_ctx_set_ring
    uint32_t mode = 0;                 // zk_ctx_set_mode: ZK_MODE_REFERENCE / ZK_MODE_HARDENED
    // workspace: up to ZK_MAX_LANES pipeline lanes, each with its own HIP stream and prover / verifier workspace; consecutive
    // chunks go to consecutive lanes, so the low-occupancy per-proof kernels, the scans the host waits for and (host-pointer
    // calls) the output phases of different chunks fall into each other's heavy phases.  Lane 0 runs on `stream`.
    uint32_t chunk = 4096;
    uint32_t ws_C = 0, ws_sec = 0, ws_n = 0;
    struct ProveLane {
        hipStream_t stream = nullptr;
        Workspace W{};
        Soa gk_am{};
        uint32_t* d_totals = nullptr;
        void* arena = nullptr;
        size_t arena_bytes = 0;
        bool ready = false;
        uint32_t last_cnt = 0;          // proofs of the last chunk this lane started (zk_test_counter)
        hipStream_t side = nullptr;     // small chunks: the membership phase of stage 2 runs beside the PointAdd phase (api.hip: ProveJob::stage2)
        hipEvent_t side_fork = nullptr, side_done = nullptr;
        void* h_scan = nullptr;         // page-locked: the chunk's totals (4 x u32), item prefix sums (u32[C+1]) and byte prefix sums
        size_t h_scan_bytes = 0;        // (u64[C+1]) read back after the scan.  Pageable destinations made the runtime wait for
                                        // EVERY stream of the device (measured: the host sat 50 ms behind the other lane's kernels)
        hipEvent_t copy_ev = nullptr;   // host-buffer entry points: "this lane's bytes are final" for the lane's copy stream
        hipStream_t copy_stream = nullptr;   // D2H of this lane's finished slices: one stream per lane, so a lane whose slices
                                             // are ready never queues behind the unfinished slices of another (FIFO per stream)
    } pl[ZK_MAX_LANES];
    uint32_t lanes = 2;
    // verifier workspace
    struct VerifyLane {
        VWork V{};
        void* arena = nullptr;
        size_t arena_bytes = 0;
        Soa res{}, res2{};
        MsmBuf M{};               // batched Tom check buffers (k_msm.hip), carved with V
        uint32_t* h_msm = nullptr;   // page-locked read-back words of run_msm
        hipEvent_t msm_done = nullptr;   // recorded behind a chunk's batched passes (stage2a); stage2b waits for it before it reads the verdicts
        bool msm_pending = false, pm_pending = false;   // ... which passes of the chunk between stage2a and stage2b are in flight
        uint32_t msm_gsz = 0;
        PMsmBuf PM{};             // cross-proof P-256 pass (k_pmsm.hip); PM.aos == nullptr: not carved (chunks below p256_batch_min)
        hipStream_t aux[V_AUX_STREAMS] = {};   // small batches: the independent per-proof sums run side by side (api_verify.hip: per_proof_range)
        hipEvent_t aux_fork = nullptr, aux_done[V_AUX_STREAMS] = {};
        bool p256_launched = false;   // stage2a launched the small chunk's P-256 sums (else stage2b does, behind the Tom-256 sums' kernels)
        bool stage2_forked = false;   // the auxiliary streams already wait for this chunk's stage 1 (stage2a of a small chunk)
        bool released_by_host = false;   // ... or the host waits for it in stage2b and launches their kernels then (VerifyJob::host_release)
        bool ready = false;
    } vl[ZK_MAX_LANES];
    uint32_t vs_C = 0, vs_sec = 0, vs_n = 0;
    uint32_t p256_batch_min = 8192;   // chunks of at least this many proofs sum their P-256 relations across proofs too (ZKATTEST_P256_BATCH; 0 = never)
    uint32_t verify_groups = 8;   // groups per chunk of the batched Tom check: 8 (16-bit windows) or 64 (13-bit windows); zk_ctx_set_verify_groups
    uint32_t vs_groups = 8;       // ... the lanes' workspaces were carved for
    bool vs_pm = false;           // ... and of the cross-proof P-256 pass
    bool vs_msm = false;          // the lanes' workspaces hold the buffers of the batched Tom check
    uint32_t verify_batch_min = 256;   // zk_ctx_set_batch_verify: chunks of at least this many proofs get the batched check (0 = never)
    // host-buffer entry points: DMA stream for page-locked caller buffers (zk_host_alloc), one event per lane
    hipStream_t copy_stream = nullptr;
    void* io_buf = nullptr;        // device staging of the proof bytes for the host-pointer entry points (grow-only: a
    size_t io_bytes = 0;           // multi-GB hipMalloc/hipFree per call costs as much as the transfer itself)
    uint8_t* h_stage = nullptr;    // page-locked mirror of in_buf's head for calls of a few proofs: the input arrays cross in ONE copy that no host thread waits for, and
    size_t h_stage_bytes = 0;      // the offsets / statuses / verdicts come back in one (api.hip: ensure_h_stage; wiped with the witness)
    hipEvent_t in_ready = nullptr; // ... the lanes wait for this event instead (the job's inputs_ready)
    void* seed_buf = nullptr;      // the verifier's own seeds when the caller gives none (32 B + 32 bytes, grow-only)
    size_t seed_bytes = 0;
    void* in_buf = nullptr;        // device copies of the small per-proof arrays of the host-pointer entry points (inputs,
    size_t in_bytes = 0;           // offsets, statuses, verdicts), grow-only for the same reason
    uint32_t wire = 0;             // zk_ctx_set_wire: 0 = ZKA1 (36-byte Tom coordinates), 1 = ZKA1P (33-byte): what the prover emits and the verifier is handed
    void* unp_buf = nullptr;       // ZKA1P input of the synchronous verify calls: staging of the expanded proofs and their per-chunk offsets (grow-only)
    size_t unp_bytes = 0;
    uint64_t* unp_off = nullptr;
    size_t unp_off_entries = 0;
    uint32_t host_taper = 1;       // host-pointer calls on page-locked buffers: tapered chunk plan (zk_ctx_set_host_taper)
    uint32_t slice = 0;            // proofs per PointAdd slice of the prover (zk_ctx_set_slice): 0 = 4096 with a page-locked sink, else none
    // streamed calls (api_stream.hip): jobs submitted and not yet waited for, in submission order
    bool stream_busy = false;
    std::vector<struct zk_job*> jobs;
    uint64_t next_lane_base = 0;       // global chunk number of the next job's first chunk (lanes rotate across jobs)
    hipStream_t fin_stream = nullptr;  // collects a job's results (offsets, statuses, verdicts) behind its last kernels and copies
    struct Spare {                     // buffers of finished jobs, reused by the next ones (grow-only, like io_buf)
        void* p;
        size_t bytes;
    };
    std::vector<Spare> spare_dev, spare_pinned;
    // unit-test counters (zk_test_counter): work done, so that tests can assert on counts instead of timings
    uint64_t dbg_recheck_proofs = 0;   // proofs that went through the verifier's per-proof sums since the context was created
    uint64_t dbg_p256_batched = 0;     // proofs whose P-256 relation was accepted by the cross-proof pass (k_pmsm.hip) since the context was created
    uint64_t dbg_msm_terms = 0;        // live terms that went through the batched Tom-256 check (k_msm.hip) since the context was created
    // timing
    std::vector<TimerRec> trecs;
    std::vector<hipEvent_t> epool;
    size_t eused = 0;
    std::vector<std::pair<const char*, float>> last_timing;
    float last_total_ms = 0, last_wall_ms = 0;   // sum of the timed scopes ('+' parts excluded); first start -> last end of the same call
    int timing_mode = ZK_TIMING_AUTO;   // zk_ctx_set_timing
    bool timing_forked = false;   // this call put timed scopes on forked streams (small one-chunk calls): its scopes overlap, the total is first start -> last end
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
// Per-family events of a blocking call of B proofs?  Two events per family and chunk cost a call of a few proofs 0.15-0.45 ms of idle GPU between kernels
// (profiles/r06_ab_variants.txt (14)); a call of more than V_SIDE_MAXP proofs does not notice them.
static inline bool zk_timed(const zk_ctx* c, uint64_t B) { return c->timing_mode == ZK_TIMING_ON || (c->timing_mode == ZK_TIMING_AUTO && B > V_SIDE_MAXP); }
static inline void timing_begin(zk_ctx* c) { c->trecs.clear(), c->eused = 0, c->timing_forked = false; }
static inline void timing_end(zk_ctx* c) {
    c->last_timing.clear();
    c->last_total_ms = 0, c->last_wall_ms = 0;
    float wall = 0;
    hipEvent_t first = c->trecs.empty() ? nullptr : c->trecs[0].e0;
    for (auto& r : c->trecs) {   // the earliest start over all records: scopes of different lanes and forked streams are not in start order
        float d = 0;
        if (hipEventElapsedTime(&d, r.e0, first) == hipSuccess && d > 0) first = r.e0;
    }
    for (auto& r : c->trecs) {
        float ms = 0;
        hipEventElapsedTime(&ms, r.e0, r.e1);
        bool found = false;
        for (auto& p : c->last_timing)
            if (p.first == r.name) p.second += ms, found = true;
        if (!found) c->last_timing.push_back({r.name, ms});
        if (r.name[0] != '+') c->last_total_ms += ms;   // '+name': a part of another scope
        float end = 0;
        if (hipEventElapsedTime(&end, first, r.e1) == hipSuccess && end > wall) wall = end;
    }
    c->last_wall_ms = wall;   // first start -> last end; last_total_ms stays the SUM of the scopes (forked scopes overlap: the sum is then not a time)
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
zk_status ensure_workspace(zk_ctx* c, uint32_t C, uint32_t nlanes = 1);   // api.hip: prover workspaces of lanes 0..nlanes-1
hipError_t malloc_or_shed(zk_ctx* c, void** p, size_t bytes);   // api.hip: a workspace allocation that sheds the optional per-ring tables first
zk_status ensure_in_buf(zk_ctx* c, size_t bytes);  // api.hip: c->in_buf of at least `bytes`
#define ZK_STAGE_MAX (1u << 20)   // input arrays up to this size take the page-locked mirror
zk_status ensure_h_stage(zk_ctx* c, size_t bytes);  // api.hip: c->h_stage of at least `bytes`, c->in_ready

// One pipeline pass = one chunk of consecutive proofs.  Device-pointer calls use uniform chunks of C proofs.  Host-pointer
// calls on page-locked buffers move ~169 KB per proof across PCIe on a copy stream under the kernels of the neighbouring
// chunks, and the transfer of a step takes almost as long as its kernels (11 GB at ~57 GB/s = 194 ms against ~250 ms), so the
// copy stream has to be fed early and evenly and whatever is still in flight at the end stays exposed:
//   * prove: a chunk's PointAdd phase -- 80 % of its bytes -- runs in proof-aligned SLICES (api.hip), each followed by the
//     D2H of the proofs it completed; what stays exposed is the last slice of each lane, so a chunk's slices taper;
//   * verify: the H2D of all chunks is enqueued up front and the lanes wait per chunk; what stays exposed is the kernels of
//     the last chunks.  Small chunks are inefficient (fixed cost of the chunk-wide bucket sum: 351 k verifies/s at 8 192
//     proofs per chunk, 250 k at 4 096), so a tapered tail was measured to LOSE (263 k against 287 k verifies/s): uniform chunks;
//   * prove: the first chunks of the lanes grow C/L, 2C/L, ... so that the lanes run out of phase and one lane's output
//     phase falls into the others' compute-only phases.
// The bytes of a proof do not depend on the plan (tests/test_gpu_scale.py::test_tapered_chunk_plan_...).
struct ChunkPlan {
    uint64_t first;
    uint32_t cnt;
};
#define ZK_TAPER_MIN 2048u
#define ZK_PROVE_SIDE_MAX 2048u   // chunks up to this size: stage 2's membership phase on the lane's side stream
#define ZK_SLICE_MIN 1024u
// sizes summing to B: the first `stagger` chunks grow C/stagger, 2C/stagger, ... (the lanes then finish out of phase), then
// chunks of C, then (tail) halving chunks C/2, C/4, ... >= lo with the last size twice
static inline std::vector<ChunkPlan> make_chunk_plan(uint64_t B, uint32_t C, uint32_t stagger, bool tail, uint32_t lo = ZK_TAPER_MIN) {
    std::vector<uint32_t> sizes, tl;
    uint64_t left = B;
    if (tail && C >= 2 * lo && B >= 2ull * C) {
        for (uint32_t s = C / 2; s >= lo; s /= 2) tl.push_back(s);
        tl.push_back(tl.back());
        for (uint32_t s : tl) left -= s;
    }
    if (stagger > 1 && C >= stagger * lo && left > (uint64_t)C * (stagger + 1) / 2) {
        for (uint32_t l = 1; l < stagger; l++) {
            uint32_t s = (uint32_t)(((uint64_t)C * l / stagger) & ~255ull);
            sizes.push_back(s), left -= s;
        }
    }
    while (left) {
        uint32_t c = (uint32_t)std::min<uint64_t>(C, left);
        sizes.push_back(c), left -= c;
    }
    for (uint32_t s : tl) sizes.push_back(s);
    std::vector<ChunkPlan> plan;
    uint64_t f = 0;
    for (uint32_t s : sizes) plan.push_back({f, s}), f += s;
    return plan;
}
zk_status ensure_io_buf(zk_ctx* c, size_t bytes);  // api.hip: c->io_buf of at least `bytes`
float pinned_d2h_rate(void* p, size_t bytes);   // api.hip: slowest D2H rate over three windows of a page-locked range (the "slow pages" check)
void* alloc_fast_pinned(size_t bytes, const std::function<void*()>& alloc, const std::function<void(void*)>& release);   // api.hip: the fastest of up to three candidates
bool host_ptr_is_pinned(const void* p);     // api.hip: page-locked (zk_host_alloc / hipHostMalloc / hipHostRegister) host memory?
zk_status ensure_copy_stream(zk_ctx* c);    // api.hip: c->copy_stream and the lanes' copy events

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
