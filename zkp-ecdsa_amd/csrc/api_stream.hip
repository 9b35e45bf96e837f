// Streamed host-pointer calls: several batches in flight on one context (include/zkattest.h, "two batches in flight").
//
// zk_prove_batch / zk_verify_batch are synchronous: every call pays its own head (no byte of a chunk exists before its stage 1 is
// over, ~45-58 ms) and its own tail (the copies of the last slices, ~20 ms, with nothing left to hide them), which is why one
// call cannot pass ~0.88 of the device-resident rate.  Here a call is split into submit (inputs staged and
// on their way, the job queued) and wait (the job's chunks driven through the pipeline, its results handed back), and the
// stage-1 look-ahead of the chunk loop runs ACROSS the boundary between consecutive jobs: while job k's last chunks are in their
// output phase the lanes already run stage 1 of job k+1's first chunks, the copy streams never drain, and the verifier's H2D of
// job k+1 starts the moment job k's bytes have crossed the link.  Usage (one thread per context, waits in submission order):
//     submit(0); submit(1); wait(0); submit(2); wait(1); submit(3); wait(2); ...
// The reference has no counterpart (one proof per call on one thread, src/zkpAttestList.ts:104-145); the bytes and verdicts
// are those of the synchronous calls (tests/test_gpu_stream.py).
#include <chrono>
#include <cstdio>
#include <vector>
#include <cstdlib>
#include "jobs.h"

zk_status make_default_vseeds(zk_ctx* c, uint64_t B, uint8_t* d_seeds, hipStream_t s);   // api_verify.hip

#define ZK_MAX_JOBS 4

struct zk_job {
    int kind = 0;   // 0 prove, 1 verify
    zk_ctx* c = nullptr;
    ProveJob pj;
    VerifyJob vj;
    void *d_in = nullptr, *d_stage = nullptr, *h_pin = nullptr;   // device inputs / results, device proof bytes, pinned staging
    void* d_unp = nullptr;                // verify jobs on ZKA1P input: the expanded proofs (d_stage receives the packed bytes)
    size_t d_in_bytes = 0, d_stage_bytes = 0, h_pin_bytes = 0, d_unp_bytes = 0;
    size_t in_bytes = 0;                  // the input block at the head of d_in / h_pin
    hipEvent_t inputs_ready = nullptr, done = nullptr;
    hipEvent_t lane_ev[ZK_MAX_LANES] = {}, copy_ev[ZK_MAX_LANES] = {};
    // results: where they sit in d_in / h_pin and where the caller wants them
    size_t res_a_off = 0, res_a_bytes = 0, res_b_off = 0, res_b_bytes = 0;
    void *user_a = nullptr, *user_b = nullptr;
    uint32_t nl = 1;                      // lanes the job was planned over (c->lanes at submit time; frozen while jobs are queued)
    bool all_enqueued = false, finisher_enqueued = false;
    zk_status result = ZK_OK;
    std::string err;
    uint64_t nchunks() const { return kind ? vj.plan.size() : pj.plan.size(); }
    uint64_t& next_s1() { return kind ? vj.next_s1 : pj.next_s1; }
    uint64_t& next_s2() { return kind ? vj.next_s2 : pj.next_s2; }
    uint64_t lane_base() const { return kind ? vj.lane_base : pj.lane_base; }
    zk_status stage1(uint64_t k) { return kind ? vj.stage1(k) : pj.stage1(k); }
    zk_status stage2(uint64_t k) { return kind ? vj.stage2(k) : pj.stage2(k); }
};

// ---- grow-only buffer pools: a finished job's buffers serve the next one (a multi-GB hipMalloc / hipHostMalloc per call costs as
// much as the transfer itself)
static void* take_spare(std::vector<zk_ctx::Spare>& pool, size_t bytes, size_t* got) {
    int best = -1;
    for (int i = 0; i < (int)pool.size(); i++)
        if (pool[i].bytes >= bytes && (best < 0 || pool[i].bytes < pool[best].bytes)) best = i;
    if (best < 0) return nullptr;
    void* p = pool[best].p;
    *got = pool[best].bytes;
    pool.erase(pool.begin() + best);
    return p;
}
static zk_status get_dev(zk_ctx* c, size_t bytes, void** p, size_t* got) {
    bytes = bytes ? bytes : 256;
    if ((*p = take_spare(c->spare_dev, bytes, got))) return ZK_OK;
    while (!c->spare_dev.empty()) {   // too small for this job: do not hoard them
        hipFree(c->spare_dev.back().p);
        c->spare_dev.pop_back();
    }
    HIPCHK(c, hipMalloc(p, bytes));
    *got = bytes;
    return ZK_OK;
}
static zk_status get_pinned(zk_ctx* c, size_t bytes, void** p, size_t* got) {
    bytes = bytes ? bytes : 256;
    if ((*p = take_spare(c->spare_pinned, bytes, got))) return ZK_OK;
    HIPCHK(c, hipHostMalloc(p, bytes, hipHostMallocDefault));
    *got = bytes;
    return ZK_OK;
}
void* stream_take_spare_dev(zk_ctx* c, size_t bytes, size_t* got) { return take_spare(c->spare_dev, bytes, got); }   // ensure_io_buf (api.hip)
void stream_release_spares(zk_ctx* c) {   // zk_ctx_destroy: the jobs' staged inputs (signatures, seeds) are zeroed before the memory goes back
    for (auto& s : c->spare_dev) (void)hipMemset(s.p, 0, s.bytes), hipFree(s.p);
    for (auto& s : c->spare_pinned) memset(s.p, 0, s.bytes), hipHostFree(s.p);
    c->spare_dev.clear(), c->spare_pinned.clear();
    c->fin_stream = nullptr;   // borrowed: the last lane's copy stream
}
static void job_free(zk_job* j) {
    zk_ctx* c = j->c;
    if (j->d_in) c->spare_dev.push_back({j->d_in, j->d_in_bytes});
    if (j->d_stage) c->spare_dev.push_back({j->d_stage, j->d_stage_bytes});
    if (j->d_unp) c->spare_dev.push_back({j->d_unp, j->d_unp_bytes});
    if (j->h_pin) c->spare_pinned.push_back({j->h_pin, j->h_pin_bytes});
    if (j->inputs_ready) hipEventDestroy(j->inputs_ready);
    if (j->done) hipEventDestroy(j->done);
    for (auto e : j->lane_ev)
        if (e) hipEventDestroy(e);
    for (auto e : j->copy_ev)
        if (e) hipEventDestroy(e);
    delete j;
}
// Every stream a job's stages may have put work on: the lanes' compute and copy streams, the forks of small one-chunk jobs (the prover's side stream, the
// verifier's auxiliary streams), the heavy queue, the finisher.  A failed or abandoned job's buffers go back to the spare pools only after all of them are idle
// -- a HIP error between a fork and its join would otherwise leave kernels on a forked stream writing into memory the next job is handed.
static void sync_every_stream(zk_ctx* c) {
    for (auto& L : c->pl) {
        if (L.stream) hipStreamSynchronize(L.stream);
        if (L.copy_stream) hipStreamSynchronize(L.copy_stream);
        if (L.side) hipStreamSynchronize(L.side);
    }
    for (auto& L : c->vl)
        for (auto a : L.aux)
            if (a) hipStreamSynchronize(a);
    if (c->fin_stream) hipStreamSynchronize(c->fin_stream);
    if (c->copy_stream) hipStreamSynchronize(c->copy_stream);
}
void stream_abandon_jobs(zk_ctx* c) {   // zk_ctx_destroy with jobs still queued: nothing may run on, their buffers are released
    if (c->jobs.empty()) return;
    sync_every_stream(c);
    while (!c->jobs.empty()) {
        zk_job* j = c->jobs.back();
        c->jobs.pop_back();
        job_free(j);
    }
    c->stream_busy = false;
}
static void unlink_job(zk_ctx* c, zk_job* j) {
    for (size_t i = 0; i < c->jobs.size(); i++)
        if (c->jobs[i] == j) {
            c->jobs.erase(c->jobs.begin() + i);
            break;
        }
    c->stream_busy = !c->jobs.empty();
}

static zk_status stream_common(zk_ctx* c, int kind) {
    if (!c->params_set || !c->N) return ZK_E_BUFFER;
    if (c->jobs.size() >= ZK_MAX_JOBS) {
        c->err = "too many streamed jobs in flight (wait for the oldest one first)";
        return ZK_E_ARG;
    }
    if (!c->jobs.empty() && c->jobs[0]->kind != kind) {
        c->err = "prove and verify jobs cannot be in flight together on one context";
        return ZK_E_ARG;
    }
    if (!c->jobs.empty() && (!(c->ws_C == c->chunk && c->ws_sec == c->P.sec && c->ws_n == c->n) || (c->jobs[0]->kind ? c->jobs[0]->vj.NL : c->jobs[0]->pj.NL) != c->lanes ||
                              (kind == 1 && c->vs_groups != c->verify_groups))) {
        c->err = "chunk / lanes / parameters / ring changed while streamed jobs are in flight";
        return ZK_E_ARG;
    }
    // The finisher stream is the LAST lane's copy stream, not a new one: HIP multiplexes streams onto 8 hardware queues in creation
    // order and the context already owns eight (api.hip, zk_ctx_create) -- a ninth stream would share lane 0's queue, and the
    // finisher's "wait for every lane" would stall the kernels of the next job queued behind it there.
    if (!c->fin_stream) c->fin_stream = c->pl[ZK_MAX_LANES - 1].copy_stream;
    if (c->io_buf) {   // the synchronous calls' staging buffer serves as a job's (it comes back through the pool: ensure_io_buf)
        c->spare_dev.push_back({c->io_buf, c->io_bytes});
        c->io_buf = nullptr, c->io_bytes = 0;
    }
    zk_status zs = ensure_copy_stream(c);
    if (zs) return zs;
    zs = ensure_workspace(c, c->chunk, c->lanes);   // whole-chunk workspaces whatever the job's size: jobs of any size may follow
    if (zs) return zs;
    if (kind == 1) zs = ensure_vworkspace(c, c->chunk, c->lanes);
    return zs;
}
static zk_status job_events(zk_job* j) {
    zk_ctx* c = j->c;
    HIPCHK(c, hipEventCreateWithFlags(&j->inputs_ready, hipEventDisableTiming));
    HIPCHK(c, hipEventCreateWithFlags(&j->done, hipEventDisableTiming));
    j->nl = c->lanes;
    for (uint32_t l = 0; l < j->nl; l++) {
        HIPCHK(c, hipEventCreateWithFlags(&j->lane_ev[l], hipEventDisableTiming));
        HIPCHK(c, hipEventCreateWithFlags(&j->copy_ev[l], hipEventDisableTiming));
    }
    return ZK_OK;
}
// after the job's last chunk: its results travel to the pinned staging behind everything the job put on the lanes
static zk_status enqueue_finisher(zk_job* j) {
    zk_ctx* c = j->c;
    j->finisher_enqueued = true;
    for (uint32_t l = 0; l < j->nl; l++) {
        HIPCHK(c, hipEventRecord(j->lane_ev[l], c->pl[l].stream));
        HIPCHK(c, hipStreamWaitEvent(c->fin_stream, j->lane_ev[l], 0));
        HIPCHK(c, hipEventRecord(j->copy_ev[l], c->pl[l].copy_stream));
        HIPCHK(c, hipStreamWaitEvent(c->fin_stream, j->copy_ev[l], 0));
    }
    if (j->kind == 1 && j->vj.host_src) {   // the verifier's H2D stream: nothing of this job may still be in flight either
        HIPCHK(c, hipEventRecord(j->copy_ev[0], c->copy_stream));
        HIPCHK(c, hipStreamWaitEvent(c->fin_stream, j->copy_ev[0], 0));
    }
    uint8_t *d = (uint8_t*)j->d_in, *h = (uint8_t*)j->h_pin;
    if (j->res_a_bytes) HIPCHK(c, hipMemcpyAsync(h + j->res_a_off, d + j->res_a_off, j->res_a_bytes, hipMemcpyDeviceToHost, c->fin_stream));
    if (j->res_b_bytes) HIPCHK(c, hipMemcpyAsync(h + j->res_b_off, d + j->res_b_off, j->res_b_bytes, hipMemcpyDeviceToHost, c->fin_stream));
    HIPCHK(c, hipEventRecord(j->done, c->fin_stream));
    return ZK_OK;
}
static void job_fail(zk_job* j, zk_status zs) {
    if (j->result == ZK_OK) j->result = zs, j->err = j->c->err;
    j->all_enqueued = true;   // its remaining chunks are skipped
}
// Drives the queue: the chunks of jobs[0 .. upto] through stage 2 (in global chunk order), with stage 1 of the next NL - 1
// chunks enqueued first -- whichever job they belong to.  ahead > 0: after `upto` is fully enqueued, up to `ahead` further
// chunks of the following jobs go through stage 2 as well (exactly those whose stage 1 the look-ahead has already enqueued), so
// that the lanes keep running while the caller blocks on `upto`'s completion.
// stage 1 of every chunk below the horizon (a global chunk number), in order, from job ji on
static void lookahead(zk_ctx* c, size_t ji, uint64_t horizon) {
    for (size_t jj = ji; jj < c->jobs.size(); jj++) {
        zk_job* Q = c->jobs[jj];
        if (Q->all_enqueued) continue;
        zk_status zs = ZK_OK;
        while (!zs && Q->next_s1() < Q->nchunks() && Q->lane_base() + Q->next_s1() < horizon) {
            zs = Q->stage1(Q->next_s1());
            Q->next_s1()++;
        }
        if (zs) job_fail(Q, zs);
        if (Q->next_s1() < Q->nchunks()) break;
    }
}
static void drive(zk_ctx* c, zk_job* upto, uint32_t ahead) {
    const uint32_t NL = upto->nl;   // the lanes the queued jobs were planned over, not whatever c->lanes says now
    bool past = false;
    for (size_t ji = 0; ji < c->jobs.size(); ji++) {
        zk_job* J = c->jobs[ji];
        while (!J->all_enqueued && J->next_s2() < J->nchunks()) {
            if (past) {
                if (!ahead) return;
                ahead--;
            }
            lookahead(c, ji, J->lane_base() + J->next_s2() + NL);   // global chunks below that may have their stage 1 enqueued
            if (J->all_enqueued) break;
            const uint64_t k2 = J->next_s2();
            zk_status zs = J->stage2(k2);
            J->next_s2()++;
            if (zs) job_fail(J, zs);
        }
        if (J->next_s2() >= J->nchunks()) J->all_enqueued = true;
        if (J->all_enqueued && !J->finisher_enqueued) {
            zk_status zs = enqueue_finisher(J);
            if (zs && J->result == ZK_OK) J->result = zs, J->err = c->err;
        }
        if (J == upto) past = true;
    }
}

static zk_status wait_common(zk_ctx* c, zk_job* j) {
    if (!c || !j || j->c != c) return ZK_E_ARG;
    if (c->jobs.empty() || c->jobs[0] != j) {
        c->err = "streamed jobs are waited for in submission order";
        return ZK_E_ARG;
    }
    HIPCHK(c, hipSetDevice(c->device));
    drive(c, j, j->nl - 1);
    hipError_t e = j->finisher_enqueued ? hipEventSynchronize(j->done) : hipSuccess;
    if (j->result != ZK_OK || e != hipSuccess) sync_every_stream(c);   // leave nothing of this job running: its buffers go back to the pool
    zk_status zs = j->result;
    if (zs) c->err = j->err;
    else if (e != hipSuccess) {
        c->err = std::string("streamed job failed: ") + hipGetErrorString(e);
        zs = ZK_E_DEVICE;
    }
    if (!zs) {
        const uint8_t* h = (const uint8_t*)j->h_pin;
        if (j->res_a_bytes) memcpy(j->user_a, h + j->res_a_off, j->res_a_bytes);
        if (j->res_b_bytes) memcpy(j->user_b, h + j->res_b_off, j->res_b_bytes);
    }
    unlink_job(c, j);
    job_free(j);
    return zs;
}

// A job its submitter gives up without waiting for it (zk_pool_*_submit: a later device's submit failed, the shards already queued must
// not stay behind).  Two cases cover every caller: the job heads the queue -- stage 1 of its first chunks may already be out, so it simply
// runs to completion here and its results are dropped --, or it is the LAST job behind older ones that are still in flight: submit only
// staged its inputs (nothing of its stages is enqueued before the queue reaches it), so it is taken out again once those uploads are
// through; the older jobs are untouched and stay waitable in their order.  wait_common on it would refuse ("submission order") and leave a
// zombie whose result pointers dangle.
zk_status stream_cancel_job(zk_ctx* c, zk_job* j) {
    if (!c || !j || j->c != c || c->jobs.empty()) return ZK_E_ARG;
    if (c->jobs[0] == j) {
        j->res_a_bytes = j->res_b_bytes = 0;   // nobody wants the results
        (void)wait_common(c, j);
        return ZK_OK;
    }
    if (c->jobs.back() != j || j->next_s1() != 0) {
        c->err = "only the head or the untouched tail of the queue can be cancelled";
        return ZK_E_ARG;
    }
    (void)hipSetDevice(c->device);
    if (j->inputs_ready) (void)hipEventSynchronize(j->inputs_ready);           // the pinned staging block goes back to the pool
    if (j->kind == 1 && j->vj.host_src) (void)hipStreamSynchronize(c->copy_stream);   // the caller's proof bytes are no longer read
    c->next_lane_base -= j->nchunks();
    unlink_job(c, j);
    if (!c->jobs.empty() && c->jobs.back()->kind == 0) c->jobs.back()->pj.more_follows = false;
    job_free(j);
    return ZK_OK;
}

extern "C" zk_status zk_prove_submit(zk_ctx* c, uint64_t B, const uint8_t* msg, const uint8_t* sig, const uint8_t* pk, const uint32_t* which, const zk_rng* rng,
                                     uint8_t* out, uint64_t out_cap, uint64_t* out_off, int32_t* status, zk_job** job) {
    if (!c || !job || !rng || !out_off || !status || !B || !msg || !sig || !pk || !which || !rng->data || !out) return ZK_E_ARG;
    *job = nullptr;
    if (rng->mode != ZK_RNG_SEED && rng->mode != ZK_RNG_STREAM) return ZK_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    if (!host_ptr_is_pinned(out)) {
        c->err = "streamed calls need a page-locked `out` (zk_host_alloc)";
        return ZK_E_ARG;
    }
    zk_status zs = stream_common(c, 0);
    if (zs) return zs;
    const size_t rng_bytes = rng->mode == ZK_RNG_SEED ? 32 * B : 32 * B * rng->stride_blocks;
    const uint64_t cap_dev = std::min<uint64_t>(out_cap, zk_proof_max_size(c) * B);
    zk_job* j = new zk_job();
    j->kind = 0, j->c = c;
    // one block: msg | sig | pk | which | rng || offsets | statuses   (256-byte aligned pieces, same layout on both sides)
    size_t top = 0;
    auto take = [&](size_t n) {
        size_t o = (top + 255) & ~(size_t)255;
        top = o + n;
        return o;
    };
    const size_t o_msg = take(32 * B), o_sig = take(64 * B), o_pk = take(64 * B), o_w = take(4 * B), o_rng = take(rng_bytes);
    j->in_bytes = top;
    j->res_a_off = take(8 * (B + 1)), j->res_a_bytes = 8 * (B + 1), j->user_a = out_off;
    j->res_b_off = take(4 * B), j->res_b_bytes = 4 * B, j->user_b = status;
    const size_t blk = top + 256;
    if ((zs = get_dev(c, blk, &j->d_in, &j->d_in_bytes)) || (zs = get_pinned(c, blk, &j->h_pin, &j->h_pin_bytes)) ||
        (zs = get_dev(c, cap_dev ? cap_dev : 32, &j->d_stage, &j->d_stage_bytes)) || (zs = job_events(j))) {
        job_free(j);
        return zs;
    }
    uint8_t *h = (uint8_t*)j->h_pin, *d = (uint8_t*)j->d_in;
    memcpy(h + o_msg, msg, 32 * B), memcpy(h + o_sig, sig, 64 * B), memcpy(h + o_pk, pk, 64 * B), memcpy(h + o_w, which, 4 * B), memcpy(h + o_rng, rng->data, rng_bytes);
    if (hipMemcpyAsync(d, h, j->in_bytes, hipMemcpyHostToDevice, c->fin_stream) != hipSuccess || hipEventRecord(j->inputs_ready, c->fin_stream) != hipSuccess) {
        c->err = "upload of the job's inputs failed";
        job_free(j);
        return ZK_E_DEVICE;
    }
    ProveJob& J = j->pj;
    J.c = c, J.B = B, J.d_msg = d + o_msg, J.d_sig = d + o_sig, J.d_pk = d + o_pk, J.d_which = (const uint32_t*)(d + o_w), J.rng_mode = rng->mode, J.d_rng = d + o_rng;
    J.stride = rng->stride_blocks, J.d_out = (uint8_t*)j->d_stage, J.out_cap = cap_dev, J.d_out_off = (uint64_t*)(d + j->res_a_off), J.d_status = (int32_t*)(d + j->res_b_off);
    J.host_sink = out, J.timed = false, J.inputs_ready = j->inputs_ready;
    J.C = (uint32_t)std::min<uint64_t>(c->chunk, B);
    // an empty pipeline starts like a synchronous call (rising first chunks put the lanes out of phase); behind a running job the
    // lanes already are
    const bool idle = c->jobs.empty();
    J.plan = make_chunk_plan(B, J.C, idle && c->host_taper ? (c->host_taper == 1 ? c->lanes : c->host_taper) : 1, false);
    J.NL = c->lanes;
    if (idle) c->next_lane_base = 0;
    J.lane_base = c->next_lane_base;
    c->next_lane_base += J.plan.size();
    if (!idle) {
        zk_job* prev = c->jobs.back();
        if (prev->kind == 0) prev->pj.more_follows = true;   // its last slices need no taper: this job's work hides their copies
    }
    c->jobs.push_back(j);
    c->stream_busy = true;
    if (idle) lookahead(c, 0, J.lane_base + J.NL);   // nothing else is driving the queue: the first chunks' stage 1 goes out right away
    *job = j;
    return ZK_OK;
}
// The device-pointer form: inputs and outputs already in HBM (what bench.py's `value` times), same queue, same waits.  Nothing crosses
// the link; the job only owns its events.  The caller's buffers must be complete when the call is made (the lanes' streams do not know
// the caller's streams) and stay untouched until the wait returns.
extern "C" zk_status zk_prove_submit_device(zk_ctx* c, uint64_t B, const uint8_t* d_msg, const uint8_t* d_sig, const uint8_t* d_pk, const uint32_t* d_which,
                                            const zk_rng* rng, uint8_t* d_out, uint64_t out_cap, uint64_t* d_out_off, int32_t* d_status, zk_job** job) {
    if (!c || !job || !rng || !d_out_off || !d_status || !B || !d_msg || !d_sig || !d_pk || !d_which || !rng->data || !d_out) return ZK_E_ARG;
    *job = nullptr;
    if (rng->mode != ZK_RNG_SEED && rng->mode != ZK_RNG_STREAM) return ZK_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    zk_status zs = stream_common(c, 0);
    if (zs) return zs;
    zk_job* j = new zk_job();
    j->kind = 0, j->c = c;
    if ((zs = job_events(j))) {
        job_free(j);
        return zs;
    }
    if (hipEventRecord(j->inputs_ready, c->fin_stream) != hipSuccess) {
        c->err = "could not record the job's start";
        job_free(j);
        return ZK_E_DEVICE;
    }
    ProveJob& J = j->pj;
    J.c = c, J.B = B, J.d_msg = d_msg, J.d_sig = d_sig, J.d_pk = d_pk, J.d_which = d_which, J.rng_mode = rng->mode, J.d_rng = rng->data;
    J.stride = rng->stride_blocks, J.d_out = d_out, J.out_cap = out_cap, J.d_out_off = d_out_off, J.d_status = d_status;
    J.host_sink = nullptr, J.timed = false, J.inputs_ready = j->inputs_ready;
    J.C = (uint32_t)std::min<uint64_t>(c->chunk, B);
    const bool idle = c->jobs.empty();
    J.plan = make_chunk_plan(B, J.C, 1, false);
    J.NL = c->lanes;
    if (idle) c->next_lane_base = 0;
    J.lane_base = c->next_lane_base;
    c->next_lane_base += J.plan.size();
    if (!idle) {
        zk_job* prev = c->jobs.back();
        if (prev->kind == 0) prev->pj.more_follows = true;
    }
    c->jobs.push_back(j);
    c->stream_busy = true;
    if (idle) lookahead(c, 0, J.lane_base + J.NL);
    *job = j;
    return ZK_OK;
}
extern "C" zk_status zk_prove_wait(zk_ctx* c, zk_job* job) { return wait_common(c, job); }

extern "C" zk_status zk_verify_submit(zk_ctx* c, uint64_t B, const uint8_t* msg, const uint8_t* proofs, const uint64_t* off, const uint8_t* vseeds, uint8_t* ok,
                                      int32_t* status, zk_job** job) {
    if (!c || !job || !B || !msg || !proofs || !off || !ok || !status) return ZK_E_ARG;
    *job = nullptr;
    HIPCHK(c, hipSetDevice(c->device));
    if (c->params_set && c->P.sec < VK) return ZK_E_SECLEVEL;
    if (off[0] != 0) return ZK_E_ARG;
    for (uint64_t b = 0; b < B; b++)
        if (off[b + 1] < off[b]) return ZK_E_ARG;
    if (!host_ptr_is_pinned(proofs)) {
        c->err = "streamed calls need page-locked `proofs` (zk_host_alloc)";
        return ZK_E_ARG;
    }
    zk_status zs = stream_common(c, 1);
    if (zs) return zs;
    const uint64_t total = off[B];
    zk_job* j = new zk_job();
    j->kind = 1, j->c = c;
    size_t top = 0;
    auto take = [&](size_t n) {
        size_t o = (top + 255) & ~(size_t)255;
        top = o + n;
        return o;
    };
    const size_t o_msg = take(32 * B), o_off = take(8 * (B + 1)), o_seed = take(32 * B + 32);
    j->in_bytes = vseeds ? top : o_seed;   // without caller seeds the seed area is filled on the device
    const bool packed = c->wire == ZK_WIRE_ZKA1P;
    const uint32_t Cj = (uint32_t)std::min<uint64_t>(c->chunk, B);
    const size_t o_uoff = packed ? take(8 * unpack_off_entries(B, Cj)) : 0;
    j->res_a_off = take(B), j->res_a_bytes = B, j->user_a = ok;
    j->res_b_off = take(4 * B), j->res_b_bytes = 4 * B, j->user_b = status;
    const size_t blk = top + 256;
    if ((zs = get_dev(c, blk, &j->d_in, &j->d_in_bytes)) || (zs = get_pinned(c, blk, &j->h_pin, &j->h_pin_bytes)) ||
        (zs = get_dev(c, total + 64, &j->d_stage, &j->d_stage_bytes)) || (packed && (zs = get_dev(c, unpack_stage_bytes(B, total, Cj), &j->d_unp, &j->d_unp_bytes))) ||
        (zs = job_events(j))) {
        job_free(j);
        return zs;
    }
    uint8_t *h = (uint8_t*)j->h_pin, *d = (uint8_t*)j->d_in;
    memcpy(h + o_msg, msg, 32 * B), memcpy(h + o_off, off, 8 * (B + 1));
    if (vseeds) memcpy(h + o_seed, vseeds, 32 * B);
    if (hipMemcpyAsync(d, h, j->in_bytes, hipMemcpyHostToDevice, c->fin_stream) != hipSuccess) {
        c->err = "upload of the job's inputs failed";
        job_free(j);
        return ZK_E_DEVICE;
    }
    if (!vseeds && (zs = make_default_vseeds(c, B, d + o_seed, c->fin_stream))) {
        job_free(j);
        return zs;
    }
    if (hipEventRecord(j->inputs_ready, c->fin_stream) != hipSuccess) {
        c->err = "hipEventRecord failed";
        job_free(j);
        return ZK_E_DEVICE;
    }
    VerifyJob& J = j->vj;
    J.c = c, J.B = B, J.d_msg = d + o_msg, J.d_proofs = (const uint8_t*)j->d_stage, J.d_off = (const uint64_t*)(d + o_off), J.d_vseeds = d + o_seed;
    J.d_ok = d + j->res_a_off, J.d_status = (int32_t*)(d + j->res_b_off), J.host_src = proofs, J.host_off = off, J.timed = false, J.inputs_ready = j->inputs_ready;
    J.C = Cj;
    J.plan = make_chunk_plan(B, J.C, 1, false);
    J.NL = c->lanes;
    if (packed) {   // d_stage receives the packed bytes, every chunk is expanded into d_unp before the kernels read it
        J.d_packed = (const uint8_t*)j->d_stage, J.d_poff = (const uint64_t*)(d + o_off), J.d_proofs = (const uint8_t*)j->d_unp, J.d_uoff = (uint64_t*)(d + o_uoff);
        if ((zs = J.plan_unpack())) {
            job_free(j);
            return zs;
        }
    }
    const bool idle = c->jobs.empty();
    if (idle) c->next_lane_base = 0;
    J.lane_base = c->next_lane_base;
    c->next_lane_base += J.plan.size();
    // the proof bytes start crossing the link NOW, chunk by chunk, right behind the bytes of the jobs submitted earlier
    if ((zs = J.enqueue_h2d())) {
        hipStreamSynchronize(c->copy_stream);
        job_free(j);
        return zs;
    }
    c->jobs.push_back(j);
    c->stream_busy = true;
    if (idle) lookahead(c, 0, J.lane_base + J.NL);
    *job = j;
    return ZK_OK;
}
extern "C" zk_status zk_verify_wait(zk_ctx* c, zk_job* job) { return wait_common(c, job); }

extern "C" uint64_t zk_test_counter(const zk_ctx* c, int which) {
    if (!c) return 0;
    if (which == 4) return g_coop_chains.load(std::memory_order_relaxed);   // chains of the small-call paths handed to cooperating waves (k_coop.hip), process-wide
    if (which == 1) {   // proofs of lane 0's last chunk that took the key-table path (k_ktab.hip)
        const auto& L = c->pl[0];
        if (!L.ready || !L.last_cnt || !L.W.kt_use) return 0;
        std::vector<uint8_t> u(L.last_cnt);
        if (hipSetDevice(c->device) != hipSuccess || hipStreamSynchronize(L.stream) != hipSuccess ||
            hipMemcpy(u.data(), L.W.kt_use, L.last_cnt, hipMemcpyDeviceToHost) != hipSuccess)
            return ~0ull;
        uint64_t n = 0;
        for (uint8_t v : u) n += v != 0;
        return n;
    }
    return which == 0 ? c->dbg_recheck_proofs : which == 2 ? c->dbg_msm_terms : which == 3 ? c->dbg_p256_batched : 0;
}
