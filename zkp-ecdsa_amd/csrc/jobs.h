// Jobs of the host-side pipelines: one prove / verify call = one job (chunk plan + the two stage functions).  The synchronous entry
// points run one job to completion (api.hip, api_verify.hip); the streamed ones (api_stream.hip) keep several queued and let the
// stage-1 look-ahead cross the boundary between consecutive jobs.
#pragma once
#include <chrono>
#include <new>
#include "ctx.h"

struct ProveJob {
    zk_ctx* c = nullptr;
    uint64_t B = 0;
    const uint8_t *d_msg = nullptr, *d_sig = nullptr, *d_pk = nullptr;
    const uint32_t* d_which = nullptr;
    int rng_mode = 0;
    const uint8_t* d_rng = nullptr;
    uint64_t stride = 0;
    uint8_t* d_out = nullptr;
    uint64_t out_cap = 0;
    uint64_t* d_out_off = nullptr;
    int32_t* d_status = nullptr;
    uint8_t* host_sink = nullptr;
    uint32_t C = 0;            // chunk size of the plan
    uint32_t NL = 1;           // lanes the chunks rotate over
    uint64_t lane_base = 0;    // global number of this job's chunk 0: chunk k runs on lane (lane_base + k) % NL
    bool timed = false;        // per-family HIP events (zk_timed: zk_ctx_set_timing; never for streamed jobs: the events of overlapping jobs would interleave)
    bool more_follows = false; // streamed: another job is queued behind this one (its work hides this job's last copies)
    hipEvent_t inputs_ready = nullptr;   // streamed: the H2D of this job's input arrays; stage 1 waits for it
    std::vector<ChunkPlan> plan;
    uint64_t cursor = 0, next_s1 = 0, next_s2 = 0;
    struct Pending {
        uint32_t lane;
        uint32_t cnt;
        uint64_t first;
        ChunkIn in;
        Workspace Wgen;  // RNG view of the generator (seed mode) for the second prepass stage
        uint32_t nblk;
    } pend[ZK_MAX_LANES];
    // ZK_IO_DEBUG=1 (synchronous calls on a page-locked sink): timeline of the D2H slices on stderr
    struct IoRec {
        hipEvent_t ready, c0, c1;
        uint64_t bytes;
        uint32_t chunk, lane;
    };
    std::vector<IoRec> iorecs;
    hipEvent_t io_t0 = nullptr;
    bool io_dbg = false;
    double host_t0 = 0;
    static double host_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

    uint32_t lane_of(uint64_t k) const { return (uint32_t)((lane_base + k) % NL); }
    hipError_t sync_lanes() {
        hipError_t r = hipSuccess;
        for (uint32_t l = 0; l < NL; l++) {
            hipError_t e = hipStreamSynchronize(c->pl[l].stream);
            if (r == hipSuccess) r = e;
        }
        return r;
    }
    zk_status stage1(uint64_t chunk_no);
    zk_status stage2(uint64_t chunk_no);
};
struct MaybeScope {   // a timed scope, or nothing
    char buf[sizeof(Scope)];
    Scope* s = nullptr;
    MaybeScope(bool on, zk_ctx* c, const char* name, hipStream_t st) {
        if (on) s = new (buf) Scope(c, name, st);
    }
    ~MaybeScope() {
        if (s) s->~Scope();
    }
};


// zk_verify_batch's pipeline, same shape: stage 1 = everything up to the term lists, stage 2 = the batched Tom check (the host
// reads counters and verdicts back), the per-proof sums of the groups that failed it, and the final verdicts.
struct VerifyJob {
    zk_ctx* c = nullptr;
    uint64_t B = 0;
    const uint8_t *d_msg = nullptr, *d_proofs = nullptr;
    const uint64_t* d_off = nullptr;
    const uint8_t* d_vseeds = nullptr;
    uint8_t* d_ok = nullptr;
    int32_t* d_status = nullptr;
    // ZKA1P input (zk_ctx_set_wire): the caller's bytes / offsets are the PACKED ones; every chunk is expanded into d_proofs (staging owned by
    // the caller of the job) by launch_v_unpack and the kernels read the chunk's own offset array d_uoff + first + k (k = chunk number)
    const uint8_t* d_packed = nullptr;
    const uint64_t* d_poff = nullptr;
    uint64_t* d_uoff = nullptr;          // [B + chunks + 1]
    std::vector<uint64_t> ubase;         // per chunk: where its expanded proofs start in d_proofs
    const uint8_t* host_src = nullptr;   // page-locked source of the proof bytes (or nullptr) and the host copy of the offsets:
    const uint64_t* host_off = nullptr;  // the bytes of chunk k travel on c->copy_stream while earlier chunks are being verified
    uint32_t C = 0, NL = 1;
    uint64_t lane_base = 0;
    bool timed = false;
    // A blocking call of a few proofs (<= V_WIDE_MAXP): the auxiliary streams' stage-2 kernels are launched when the HOST has seen stage 1's last kernel end, instead of
    // being queued early behind a wait for its event.  Two or more queues that hold a wait slow every OTHER queue of the device down by ~17 us per kernel boundary
    // (tools/sha_bench.hip: 21 -> 57 us per pair of small kernels), and stage 1's tail is a chain of ten small kernels on the main stream.
    bool host_release = false;
    hipEvent_t inputs_ready = nullptr;
    std::vector<ChunkPlan> plan;
    std::vector<hipEvent_t> arrived;     // host_src: one event per chunk, recorded on the copy stream behind the chunk's bytes
    uint64_t next_s1 = 0, next_s2 = 0;
    ~VerifyJob() {
        for (auto e : arrived)
            if (e) hipEventDestroy(e);
    }
    uint32_t lane_of(uint64_t k) const { return (uint32_t)((lane_base + k) % NL); }
    // a call of ONE small chunk is a chain of latencies on an idle GPU: its independent phases go to the lane's auxiliary streams.  (Chunks of a longer job
    // overlap each other on the lanes already; there the extra streams only compete with the copy stream: 294 k -> 196 k verifies/s at 8 x 8192 proofs.)
    bool side_streams(uint32_t cnt) const { return plan.size() == 1 && cnt <= V_SIDE_MAXP; }
    // the chunk's P-256 relations are summed across proofs (k_pmsm.hip) instead of per proof
    bool p256_batched(uint32_t cnt, uint32_t lane) const {
        return !side_streams(cnt) && c->verify_batch_min && cnt >= c->verify_batch_min && c->p256_batch_min && cnt >= c->p256_batch_min && c->vl[lane].PM.aos && c->vl[lane].M.cap;
    }
    const uint64_t* off_of(uint64_t k) const { return d_packed ? d_uoff + k : d_off; }   // what the kernels index with [first + p]
    zk_status plan_unpack();             // ubase from the packed offsets of the chunks' first proofs (host copy, or read back from d_poff)
    zk_status enqueue_h2d();             // all chunks' bytes up front, one event per chunk
    zk_status stage1(uint64_t chunk_no);
    // stage 2 in two halves: 2a enqueues the chunk's batched passes (Tom-256 and P-256 bucket sums, no host round trip), 2b waits for their verdicts and enqueues
    // the per-proof sums of the groups that failed and the final kernel.  verify_device enqueues 2a of every chunk whose stage 1 is in before it blocks in 2b of
    // the oldest one: the dependent chains at the end of one chunk's pass (the bucket reductions, 3 ms) then run beside the next chunk's bucket sums.
    zk_status stage2a(uint64_t chunk_no);
    zk_status stage2b(uint64_t chunk_no);
    zk_status stage2(uint64_t chunk_no) {
        zk_status z = stage2a(chunk_no);
        return z ? z : stage2b(chunk_no);
    }
    uint64_t next_s2a = 0;
};
zk_status ensure_vworkspace(zk_ctx* c, uint32_t C, uint32_t nlanes);   // api_verify.hip
// bytes of the expansion staging and entries of the offset array for a ZKA1P batch of B proofs, `total` packed bytes, chunks of C proofs
static inline size_t unpack_stage_bytes(uint64_t B, uint64_t total, uint32_t C) { return (size_t)((total * 12 + 10) / 11 + 32 * B + 256 * (B / (C ? C : 1) + 2) + 64); }
static inline size_t unpack_off_entries(uint64_t B, uint32_t C) { return (size_t)(B + B / (C ? C : 1) + 4); }
