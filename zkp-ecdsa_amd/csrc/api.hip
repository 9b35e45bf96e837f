// C ABI of libzkattest_hip.so (include/zkattest.h) and the host-side phase pipeline of the prover.
// One context = one GPU = one HIP stream.  A batch is processed in chunks of `chunk` proofs; every phase of a chunk
// is one kernel over all proofs (or all zero-bit reps) of the chunk -- see DESIGN.md for the phase list.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <functional>
#include <string>
#include <vector>
#include "engine.h"

void launch_synth(hipStream_t s, const uint32_t* pfix_G, uint64_t seed, uint64_t nkeys, uint64_t B, uint8_t* ring, uint8_t* msg, uint8_t* sig, uint8_t* pk,
                  uint32_t* which, uint8_t* seeds);
void launch_synth_param_scalars(hipStream_t s, uint64_t seed, uint8_t* kn_be, uint8_t* kt_be);

#include <cstdlib>
#include <sys/syscall.h>
#include <unistd.h>
#include <atomic>
#include "ctx.h"
#include <array>
#include <map>
#include "jobs.h"

extern "C" const char* zk_strerror(zk_status s) {
    switch (s) {
    case ZK_OK: return "ok";
    case ZK_E_POINT_NOT_IN_GROUP: return "point not in group";
    case ZK_E_INVALID_KEY: return "invalid public key";
    case ZK_E_T_INF: return "T[i] is at infinity";
    case ZK_E_T1_INF: return "T1 is at infinity";
    case ZK_E_PADD_INF: return "P/Q/R is at infinity";
    case ZK_E_POINTS_DONT_ADD: return "Points don't add up!";
    case ZK_E_R_INF: return "R is at infinity";
    case ZK_E_PARAMS_NOT_FOUND: return "params not found";
    case ZK_E_SECLEVEL: return "security level not achieved";
    case ZK_E_BAD_ENCODING: return "error deserializing";
    case ZK_E_RNG_EXHAUSTED: return "randomness stream exhausted";
    case ZK_E_BUFFER: return "buffer too small or context not configured";
    case ZK_E_INTERPOLATION: return "incorrect interpolation";
    case ZK_E_ARG: return "invalid argument";
    case ZK_E_DEVICE: return "HIP runtime failure";
    }
    return "unknown status";
}
static thread_local std::string g_ctx_create_err;   // why this thread's last zk_ctx_create failed: zk_last_error(NULL)
extern "C" const char* zk_last_error(const zk_ctx* c) { return c ? c->err.c_str() : g_ctx_create_err.c_str(); }

// HIP multiplexes its streams onto a few hardware queues, round-robin in creation order (GPU_MAX_HW_QUEUES, default 4; measured
// with tools/stream_overlap.hip: of eight streams, numbers 3 and 7 queue behind number 0).  Two streams on one hardware queue
// run strictly one after the other, and a "wait for that event" of a copy stream stalls every compute kernel queued behind it:
// with carelessly created streams the two lanes of a context executed serially.  So (1) the library asks for 8 hardware queues
// when it is loaded before the HIP runtime starts (it cannot change a runtime that is already up: a host that initialises HIP
// first -- bench.py imports torch -- exports GPU_MAX_HW_QUEUES itself), and (2) a context creates all its streams in one go, in
// an order that gives the first two lanes and their copy streams four different queues even with the default of 4.
// ZKATTEST_NO_ENV=1 (any value but "0") keeps the library from touching the process environment at load: the host then exports
// GPU_MAX_HW_QUEUES=8 itself, or lives with lanes that share hardware queues.  An existing value is never overwritten.
__attribute__((constructor)) static void zk_more_hw_queues() {
    const char* off = getenv("ZKATTEST_NO_ENV");
    if (off && *off && strcmp(off, "0") != 0) return;
    setenv("GPU_MAX_HW_QUEUES", "8", 0);
}

static zk_status ctx_init(zk_ctx* c, int device_id);
// *out is either a fully initialised context or NULL (then zk_last_error(NULL) has the reason): a caller never holds a half-built one
extern "C" zk_status zk_ctx_create(int device_id, zk_ctx** out) {
    if (!out) return ZK_E_ARG;
    *out = nullptr;
    zk_ctx* c = new zk_ctx();
    c->device = device_id;
    zk_status zs = ctx_init(c, device_id);
    if (zs) {
        g_ctx_create_err = c->err;
        zk_ctx_destroy(c);
        (void)hipGetLastError();
        return zs;
    }
    *out = c;
    return ZK_OK;
}
static zk_status ctx_init(zk_ctx* c, int device_id) {
    HIPCHK(c, hipSetDevice(device_id));
    for (int base = 0; base < ZK_MAX_LANES; base += 2) {   // compute 0, compute 1, copy 0, copy 1; then the same for lanes 2, 3
        for (int l = base; l < base + 2; l++) HIPCHK(c, hipStreamCreate(&c->pl[l].stream));
        for (int l = base; l < base + 2; l++) HIPCHK(c, hipStreamCreateWithFlags(&c->pl[l].copy_stream, hipStreamNonBlocking));
    }
    c->stream = c->pl[0].stream;
    c->copy_stream = c->pl[0].copy_stream;   // H2D of the verifier's proofs
    if (const char* e = getenv("ZKATTEST_COMB_BITS")) {
        int b = atoi(e);
        if (b >= 8 && b <= TOM_MAX_BITS) c->tom_bits = (uint32_t)b;
    }
    if (const char* e = getenv("ZKATTEST_GK_TABLE")) c->gk_table = atoi(e) != 0;
    if (const char* e = getenv("ZKATTEST_KEYTAB")) c->key_tables = atoi(e) != 0;
    if (const char* e = getenv("ZKATTEST_GK_MFMA")) c->gk_mfma = atoi(e) != 0;
    if (const char* e = getenv("ZKATTEST_GK_MFMA_PROVE")) c->gk_mfma_prove = atoi(e) != 0;
    if (const char* e = getenv("ZKATTEST_VERIFY_GROUPS")) c->verify_groups = atoi(e) == 64 ? 64 : 8;
    if (const char* e = getenv("ZKATTEST_VERIFY_BATCH")) c->verify_batch_min = (uint32_t)atoi(e);
    if (const char* e = getenv("ZKATTEST_P256_BATCH")) c->p256_batch_min = (uint32_t)atoi(e);   // chunk size from which the P-256 relations are summed across proofs (0 = never)
    if (const char* e = getenv("ZKATTEST_LANES")) {
        int l = atoi(e);
        if (l >= 1 && l <= ZK_MAX_LANES) c->lanes = (uint32_t)l;
    }
    HIPCHK(c, hipMalloc(&c->tom_tab_gen, sizeof(uint32_t) * tom_tab_words(8)));
    HIPCHK(c, hipMalloc(&c->P.pfix_G, sizeof(uint32_t) * PFIX_TAB_WORDS));
    HIPCHK(c, hipMalloc(&c->P.pfix_H, sizeof(uint32_t) * PFIX_TAB_WORDS));
    c->scratch_words = std::max(pfix_table_scratch_words(), tom_table_scratch_words(TOM_MAX_BITS));
    HIPCHK(c, hipMalloc(&c->tab_scratch, sizeof(uint32_t) * c->scratch_words));
    HIPCHK(c, hipMalloc(&c->d_flag, 64));
    int32_t one = 1;
    HIPCHK(c, hipMemcpyAsync(c->d_flag, &one, 4, hipMemcpyHostToDevice, c->stream));
    // tables that do not depend on the parameters: P-256 generator, Tom generator (8-bit comb: only zk_synth_params uses it)
    launch_build_pfix_table(c->stream, nullptr, c->P.pfix_G, c->tab_scratch, c->d_flag);
    uint32_t genw[18];
    memcpy(genw, TOM_GX_W, 36), memcpy(genw + 9, TOM_GY_W, 36);
    uint32_t* d_xy;
    HIPCHK(c, hipMalloc(&d_xy, 18 * 4));
    HIPCHK(c, hipMemcpyAsync(d_xy, genw, 72, hipMemcpyHostToDevice, c->stream));
    launch_build_tom_table(c->stream, d_xy, 8, c->tom_tab_gen, c->tab_scratch, c->d_flag);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipFree(d_xy));
    return ZK_OK;
}
// settings a queued streamed job was planned with (its lane rotation, chunk size, workspaces, mode) stay frozen until it is waited for
static zk_status busy_refusal(zk_ctx* c) {
    c->err = "streamed jobs are in flight on this context (zk_prove_wait / zk_verify_wait them first)";
    return ZK_E_ARG;
}
extern "C" zk_status zk_ctx_set_comb_bits(zk_ctx* c, uint32_t bits) {
    if (!c || bits < 8 || bits > TOM_MAX_BITS) return ZK_E_ARG;
    if (c->stream_busy) return busy_refusal(c);
    if (bits != c->tom_bits) c->params_set = false;  // the tables are rebuilt by the next zk_ctx_set_params
    c->tom_bits = bits;
    return ZK_OK;
}
void stream_release_spares(zk_ctx* c);   // api_stream.hip
void stream_abandon_jobs(zk_ctx* c);
// Witness-derived device memory of a context: the prover lanes' workspaces (the RNG stream of every proof -- 116 KB each --, nonces, s1 = s / r, blinders,
// responses before they are written out) and the staging buffer of the host-pointer calls (signatures, seeds).  Zeroed when the context is destroyed, when a
// prove call fails, and on request (zk_ctx_wipe); a successful call leaves them as they are -- the next call overwrites them, and wiping 2.5 GB of RNG
// stream per 22 016-proof chunk would cost ~0.5 ms of every call.  (The reference leaves its BigInts to the garbage collector.)  Nothing may be in flight.
static void wipe_witness(zk_ctx* c) {
    for (int l = 0; l < ZK_MAX_LANES; l++)
        if (c->pl[l].arena && c->pl[l].stream) (void)hipMemsetAsync(c->pl[l].arena, 0, c->pl[l].arena_bytes, c->pl[l].stream);
    if (c->in_buf && c->stream) (void)hipMemsetAsync(c->in_buf, 0, c->in_bytes, c->stream);
    if (c->h_stage) {   // the page-locked mirror of the inputs (signatures, RNG blocks)
        volatile uint8_t* h = c->h_stage;
        for (size_t i = 0; i < c->h_stage_bytes; i++) h[i] = 0;
    }
    for (int l = 0; l < ZK_MAX_LANES; l++)
        if (c->pl[l].stream) (void)hipStreamSynchronize(c->pl[l].stream);
    (void)hipGetLastError();
}
extern "C" zk_status zk_ctx_wipe(zk_ctx* c) {
    if (!c) return ZK_E_ARG;
    if (c->stream_busy) {
        c->err = "streamed jobs are in flight on this context (zk_prove_wait / zk_verify_wait them first)";
        return ZK_E_ARG;
    }
    HIPCHK(c, hipSetDevice(c->device));
    wipe_witness(c);
    return ZK_OK;
}
extern "C" void zk_ctx_destroy(zk_ctx* c) {
    if (!c) return;
    hipSetDevice(c->device);
    if (c->stream) hipStreamSynchronize(c->stream);
    stream_abandon_jobs(c);
    stream_release_spares(c);
    wipe_witness(c);
    for (auto e : c->epool) hipEventDestroy(e);
    hipFree(c->P.tom_tab_g), hipFree(c->P.tom_tab_h), hipFree(c->tom_tab_gen), hipFree(c->P.pfix_G), hipFree(c->P.pfix_H);
    hipFree(c->gk_kdig), hipFree(c->gk_edig), hipFree(c->ktab), hipFree(c->ktab_ok);
    hipFree(c->tab_scratch), hipFree(c->gk_etab), hipFree(c->d_flag), hipFree(c->ring_mem), hipFree(c->ring_digest);
    hipFree(c->io_buf), hipFree(c->in_buf), hipFree(c->unp_buf), hipFree(c->unp_off), hipFree(c->seed_buf);
    if (c->h_stage) hipHostFree(c->h_stage);
    if (c->in_ready) hipEventDestroy(c->in_ready);
    for (int l = 0; l < ZK_MAX_LANES; l++) {
        hipFree(c->pl[l].arena), hipFree(c->pl[l].d_totals), hipFree(c->vl[l].arena);
        if (c->pl[l].h_scan) hipHostFree(c->pl[l].h_scan);
        if (c->vl[l].h_msm) hipHostFree(c->vl[l].h_msm);
        if (c->vl[l].aux_fork) hipEventDestroy(c->vl[l].aux_fork);
        for (int i = 0; i < V_AUX_STREAMS; i++) {
            if (c->vl[l].aux_done[i]) hipEventDestroy(c->vl[l].aux_done[i]);
            if (c->vl[l].aux[i]) hipStreamDestroy(c->vl[l].aux[i]);
        }
        if (c->pl[l].copy_ev) hipEventDestroy(c->pl[l].copy_ev);
        if (c->pl[l].side_fork) hipEventDestroy(c->pl[l].side_fork);
        if (c->pl[l].side_done) hipEventDestroy(c->pl[l].side_done);
        if (c->pl[l].side) hipStreamDestroy(c->pl[l].side);
        if (c->vl[l].msm_done) hipEventDestroy(c->vl[l].msm_done);
        if (c->pl[l].copy_stream) hipStreamDestroy(c->pl[l].copy_stream);
        if (c->pl[l].stream) hipStreamDestroy(c->pl[l].stream);
    }
    delete c;
}

// big-endian byte string -> little-endian 32-bit words
static void be_to_words(const uint8_t* be, int nbytes, uint32_t* w, int nw) {
    for (int i = 0; i < nw; i++) w[i] = 0;
    for (int i = 0; i < nbytes; i++) {
        int bi = nbytes - 1 - i;
        w[bi / 4] |= (uint32_t)be[i] << (8 * (bi % 4));
    }
}
static void words_to_limbs30(const uint32_t* w, int nw, uint32_t* l) {
    for (int i = 0; i < 9; i++) {
        int bit = 30 * i, wi = bit / 32, sh = bit % 32;
        uint64_t v = 0;
        if (wi < nw) v = w[wi];
        if (wi + 1 < nw) v |= (uint64_t)w[wi + 1] << 32;
        l[i] = (uint32_t)(v >> sh) & 0x3fffffffu;
    }
}

extern "C" zk_status zk_ctx_set_params(zk_ctx* c, const uint8_t nist_h[64], const uint8_t tom_g[72], const uint8_t tom_h[72], uint32_t sec) {
    if (!c || !nist_h || !tom_g || !tom_h) return ZK_E_ARG;
    if (sec == 0 || sec > ZK_MAXSEC) return ZK_E_SECLEVEL;
    if (c->stream_busy) {
        c->err = "streamed jobs are in flight on this context (zk_prove_wait / zk_verify_wait them first)";
        return ZK_E_ARG;
    }
    HIPCHK(c, hipSetDevice(c->device));
    uint32_t hw[16], gw[18], hw2[18];
    be_to_words(nist_h, 32, hw, 8), be_to_words(nist_h + 32, 32, hw + 8, 8);
    be_to_words(tom_g, 36, gw, 9), be_to_words(tom_g + 36, 36, gw + 9, 9);
    be_to_words(tom_h, 36, hw2, 9), be_to_words(tom_h + 36, 36, hw2 + 9, 9);
    uint32_t* d;
    HIPCHK(c, hipMalloc(&d, 64 * 4));
    HIPCHK(c, hipMemcpyAsync(d, hw, 64, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(d + 16, gw, 72, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(d + 34, hw2, 72, hipMemcpyHostToDevice, c->stream));
    int32_t one = 1;
    HIPCHK(c, hipMemcpyAsync(c->d_flag, &one, 4, hipMemcpyHostToDevice, c->stream));
    if (c->tab_bits_alloc != c->tom_bits) {
        if (c->P.tom_tab_g) HIPCHK(c, hipFree(c->P.tom_tab_g));
        if (c->P.tom_tab_h) HIPCHK(c, hipFree(c->P.tom_tab_h));
        c->P.tom_tab_g = c->P.tom_tab_h = nullptr, c->tab_bits_alloc = 0;
        HIPCHK(c, hipMalloc(&c->P.tom_tab_g, sizeof(uint32_t) * tom_tab_words(c->tom_bits)));
        HIPCHK(c, hipMalloc(&c->P.tom_tab_h, sizeof(uint32_t) * tom_tab_words(c->tom_bits)));
        c->tab_bits_alloc = c->P.tom_bits = c->tom_bits;
    }
    launch_build_pfix_table(c->stream, d, c->P.pfix_H, c->tab_scratch, c->d_flag);
    launch_build_tom_table(c->stream, d + 16, c->tom_bits, c->P.tom_tab_g, c->tab_scratch, c->d_flag);
    launch_build_tom_table(c->stream, d + 34, c->tom_bits, c->P.tom_tab_h, c->tab_scratch, c->d_flag);
    int32_t ok = 0;
    HIPCHK(c, hipMemcpyAsync(&ok, c->d_flag, 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipFree(d));
    if (!ok) {
        c->params_set = false;
        return ZK_E_POINT_NOT_IN_GROUP;
    }
    words_to_limbs30(gw, 9, c->P.tom_g_aff), words_to_limbs30(gw + 9, 9, c->P.tom_g_aff + 9);
    c->P.sec = sec;
    c->params_set = true;
    return ZK_OK;
}

static zk_status set_ring_common(zk_ctx* c, const uint8_t* d_keys, uint64_t nkeys) {
    if (c->stream_busy) {   // the queued jobs read the ring, table E and the key tables this call would free
        c->err = "streamed jobs are in flight on this context (zk_prove_wait / zk_verify_wait them first)";
        return ZK_E_ARG;
    }
    uint32_t n = 0;
    while (((uint64_t)1 << n) < nkeys) n++;
    if (n < 1 || n > ZK_MAXN - 4) return ZK_E_ARG;  // N = 1 is the reference's degenerate n = 0 case (untested there)
    uint64_t N = (uint64_t)1 << n;
    if (c->ring_mem) HIPCHK(c, hipFree(c->ring_mem));
    c->ring_mem = nullptr;
    HIPCHK(c, hipMalloc(&c->ring_mem, sizeof(uint32_t) * 9 * N));
    Soa ring = {c->ring_mem, (uint32_t)N};
    launch_ring_load(c->stream, d_keys, nkeys, N, ring);
    if (c->gk_etab) HIPCHK(c, hipFree(c->gk_etab));
    c->gk_etab = nullptr;
    if (c->gk_table && n >= GK_ETAB_MINN && n <= GK_ETAB_MAXN) {  // per-ring table of the 8 low fold levels (k_gk.hip)
        HIPCHK(c, hipMalloc(&c->gk_etab, sizeof(uint32_t) * gk_etab_words(N)));
        launch_gk_etab(c->stream, ring, (uint32_t)(N >> 8), c->gk_etab);
    }
    if (c->gk_kdig) HIPCHK(c, hipFree(c->gk_kdig));
    c->gk_kdig = nullptr;
    // the two digit-fragment tables and the per-key tables below are optimisations, not requirements: where the HBM is not there (several
    // contexts on one device, a ring of 2^20 keys next to other tenants) the allocation failure is cleared and the vector-ALU fold / the
    // per-proof tables of R serve every proof -- same bytes, same verdicts
    if (c->gk_etab && n >= GKM_MINN) {   // 33 bytes per key: the verifier's ring fold on the matrix pipe (k_gk_mfma.hip)
        if (hipMalloc(&c->gk_kdig, gkm_ring_frag_bytes(N)) == hipSuccess) launch_gkm_ring_digits(c->stream, ring, (uint32_t)(N >> 8), c->gk_kdig);
        else (void)hipGetLastError(), c->gk_kdig = nullptr;
    }
    if (c->gk_edig) HIPCHK(c, hipFree(c->gk_edig));
    c->gk_edig = nullptr;
    if (c->gk_etab && n >= GKM_MINN && c->gk_mfma_prove) {   // table E's classes 2..6 as digit fragments: the prover's matrix-pipe path
        if (hipMalloc(&c->gk_edig, gkm_etab_frag_bytes(N)) == hipSuccess) launch_gkm_etab_digits(c->stream, c->gk_etab, (uint32_t)(N >> 8), c->gk_edig);
        else (void)hipGetLastError(), c->gk_edig = nullptr;
    }
    if (c->ktab) HIPCHK(c, hipFree(c->ktab));
    if (c->ktab_ok) HIPCHK(c, hipFree(c->ktab_ok));
    c->ktab = nullptr, c->ktab_ok = nullptr;
    if (c->key_tables && n <= KTAB_MAXN) {   // multiples of every ring key (k_ktab.hip): the prover's k * pk and alpha * R become table sums
        const uint32_t slab = (uint32_t)std::min<uint64_t>(N, 4096);
        void* tmp = nullptr;
        // an optimisation, not a requirement: where the HBM is not there (several contexts on one device, a small card) the per-proof
        // tables of R serve every proof
        if (hipMalloc(&c->ktab, sizeof(uint32_t) * KTAB_KEY_WORDS * N) == hipSuccess && hipMalloc(&c->ktab_ok, N) == hipSuccess &&
            hipMalloc(&tmp, ktab_temp_bytes(N, slab)) == hipSuccess) {
            launch_ktab_build(c->stream, ring, N, c->ktab, c->ktab_ok, tmp, slab);
            hipError_t e = hipStreamSynchronize(c->stream);
            if (e != hipSuccess) {
                hipFree(tmp);
                HIPCHK(c, e);
            }
        } else {
            (void)hipGetLastError();
            hipFree(c->ktab), hipFree(c->ktab_ok);
            c->ktab = nullptr, c->ktab_ok = nullptr;
        }
        if (tmp) HIPCHK(c, hipFree(tmp));
    }
    {   // digest of the padded ring: what the hardened mode hashes into the membership challenge
        if (!c->ring_digest) HIPCHK(c, hipMalloc(&c->ring_digest, 32));
        uint32_t* leaves = nullptr;
        HIPCHK(c, hipMalloc(&leaves, 32 * ((N + 255) / 256)));
        launch_ring_digest(c->stream, ring, N, leaves, c->ring_digest);
        HIPCHK(c, hipStreamSynchronize(c->stream));
        HIPCHK(c, hipFree(leaves));
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->N = N, c->n = n, c->nkeys = nkeys;
    c->ws_C = 0;  // the workspace layout depends on the ring
    return ZK_OK;
}
extern "C" zk_status zk_ctx_set_ring_device(zk_ctx* c, const void* d_keys, uint64_t nkeys) {
    if (!c || !d_keys || nkeys < 2) return ZK_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    return set_ring_common(c, (const uint8_t*)d_keys, nkeys);
}
extern "C" zk_status zk_ctx_set_ring(zk_ctx* c, const uint8_t* keys, uint64_t nkeys) {
    if (!c || !keys || nkeys < 2) return ZK_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    uint8_t* d;
    HIPCHK(c, hipMalloc(&d, 32 * nkeys));
    HIPCHK(c, hipMemcpy(d, keys, 32 * nkeys, hipMemcpyHostToDevice));
    zk_status s = set_ring_common(c, d, nkeys);
    hipFree(d);
    return s;
}
extern "C" zk_status zk_keys_to_ints(zk_ctx* c, uint64_t n, const uint8_t* pk, uint8_t* out, int32_t* st) {
    if (!c || !pk || !out || !st || !n) return ZK_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    uint8_t *d_pk = nullptr, *d_out = nullptr;
    int32_t* d_st = nullptr;
    HIPCHK(c, hipMalloc(&d_pk, 64 * n));
    HIPCHK(c, hipMalloc(&d_out, 32 * n));
    HIPCHK(c, hipMalloc(&d_st, 4 * n));
    HIPCHK(c, hipMemcpyAsync(d_pk, pk, 64 * n, hipMemcpyHostToDevice, c->stream));
    launch_keys_to_ints(c->stream, d_pk, n, d_out, d_st);
    HIPCHK(c, hipMemcpyAsync(out, d_out, 32 * n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(st, d_st, 4 * n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    hipFree(d_pk), hipFree(d_out), hipFree(d_st);
    return ZK_OK;
}
extern "C" zk_status zk_ctx_set_chunk(zk_ctx* c, uint32_t chunk) {
    if (!c || chunk == 0 || chunk > (1u << 18)) return ZK_E_ARG;
    if (c->stream_busy) return busy_refusal(c);
    c->chunk = chunk;
    return ZK_OK;
}
extern "C" zk_status zk_ctx_set_batch_verify(zk_ctx* c, uint32_t min_chunk) {
    if (!c) return ZK_E_ARG;
    if (c->stream_busy) return busy_refusal(c);
    c->verify_batch_min = min_chunk;
    return ZK_OK;
}
extern "C" zk_status zk_ctx_set_verify_groups(zk_ctx* c, uint32_t groups) {
    if (!c || (groups != 8 && groups != 64)) return ZK_E_ARG;
    if (c->stream_busy) return busy_refusal(c);
    c->verify_groups = groups;
    return ZK_OK;
}
extern "C" zk_status zk_ctx_set_key_tables(zk_ctx* c, uint32_t on) {
    if (!c) return ZK_E_ARG;
    c->key_tables = on != 0;
    return ZK_OK;
}
extern "C" zk_status zk_ctx_set_ring_fold(zk_ctx* c, uint32_t matrix_pipe) {
    if (!c) return ZK_E_ARG;
    if (c->stream_busy) return busy_refusal(c);
    c->gk_mfma = matrix_pipe != 0;
    return ZK_OK;
}
extern "C" zk_status zk_ctx_set_mode(zk_ctx* c, uint32_t mode) {
    if (!c || (mode != ZK_MODE_REFERENCE && mode != ZK_MODE_HARDENED)) return ZK_E_ARG;
    if (c->stream_busy) return busy_refusal(c);
    c->mode = mode;
    return ZK_OK;
}
extern "C" zk_status zk_ring_digest(zk_ctx* c, uint8_t digest[32]) {
    if (!c || !digest) return ZK_E_ARG;
    if (!c->N || !c->ring_digest) return ZK_E_BUFFER;
    HIPCHK(c, hipSetDevice(c->device));
    uint32_t w[8];
    HIPCHK(c, hipMemcpy(w, c->ring_digest, 32, hipMemcpyDeviceToHost));
    for (int i = 0; i < 8; i++)
        for (int j = 0; j < 4; j++) digest[4 * i + j] = (uint8_t)(w[i] >> (24 - 8 * j));
    return ZK_OK;
}
extern "C" zk_status zk_ctx_set_wire(zk_ctx* c, uint32_t wire) {
    if (!c || (wire != ZK_WIRE_ZKA1 && wire != ZK_WIRE_ZKA1P)) return ZK_E_ARG;
    if (c->stream_busy) return busy_refusal(c);
    if (wire != c->wire) c->ws_C = 0;   // the lanes' workspaces carry the layout
    c->wire = wire;
    return ZK_OK;
}
extern "C" zk_status zk_ctx_set_slice(zk_ctx* c, uint32_t proofs) {
    if (!c || (proofs && proofs < 64)) return ZK_E_ARG;
    if (c->stream_busy) return busy_refusal(c);
    c->slice = proofs;
    return ZK_OK;
}
extern "C" zk_status zk_ctx_set_host_taper(zk_ctx* c, uint32_t on) {
    if (!c) return ZK_E_ARG;
    if (c->stream_busy) return busy_refusal(c);
    c->host_taper = on > 64 ? 64 : on;   // 0 = uniform chunks, 1 = as many rising first chunks as lanes, n >= 2 = n of them
    return ZK_OK;
}
extern "C" zk_status zk_ctx_set_lanes(zk_ctx* c, uint32_t lanes) {
    if (!c || lanes < 1 || lanes > ZK_MAX_LANES) return ZK_E_ARG;
    if (c->stream_busy) return busy_refusal(c);
    c->lanes = lanes;
    return ZK_OK;
}
extern "C" uint64_t zk_proof_max_size(const zk_ctx* c) {
    if (!c || !c->params_set || !c->N) return 0;
    return wire_proof_size(wire_make(c->wire == ZK_WIRE_ZKA1P), c->P.sec, c->n, c->P.sec);
}

// ------------------------------------------------------------------ workspace arena
static size_t carve(zk_ctx* c, Workspace& W, Soa& gk_am, uint8_t* base, uint32_t C, uint32_t sec, uint32_t n, uint64_t N) {
    Carver k(base);
    uint64_t cs = (uint64_t)C * sec;
    uint64_t items_cap = std::min<uint64_t>(cs, cs / 2 + (uint64_t)(4.0 * std::sqrt((double)cs)) + 64);
    W.C = C, W.sec = sec, W.n = n, W.N = (uint32_t)N, W.items_cap = (uint32_t)items_cap;
    W.st = (int32_t*)k.take(4 * (size_t)C);
    W.pkx = k.soa(C), W.pky = k.soa(C), W.pkxm = k.soa(C), W.pkym = k.soa(C);
    W.Rxm = k.soa(C), W.Rym = k.soa(C), W.Rx = k.soa(C), W.Ry = k.soa(C);
    W.Q = k.soa3(C), W.s1 = k.soa(C);
    W.ktab = c->ktab, W.ktab_ok = c->ktab_ok;
    W.kt_use = (uint8_t*)k.take((size_t)C), W.kt_key = (uint32_t*)k.take(4 * (size_t)C), W.r_zero = (uint8_t*)k.take((size_t)C);
    W.u1m = k.soa(C), W.u2m = k.soa(C);
    W.rtab = (uint32_t*)k.take(sizeof(uint32_t) * (size_t)std::max(rtab_words(RTAB_PROVE_BITS), rtab_words(RTAB_VERIFY_BITS)) * C);
    W.rbase = k.soa3((size_t)C * RTAB_MAX_NWIN);
    W.chal = (uint32_t*)k.take(16 * (size_t)C);
    W.zcnt = (uint32_t*)k.take(4 * (size_t)C);
    W.item_base = (uint32_t*)k.take(4 * ((size_t)C + 1));
    W.out_base = (uint64_t*)k.take(8 * ((size_t)C + 1));
    size_t ne = (size_t)C * (sec + 1);
    W.Tproj = k.soa3(ne), W.Aproj = k.soa3(ne);
    W.Tx = k.soa(ne), W.Ty = k.soa(ne), W.Ax = k.soa(ne), W.Ay = k.soa(ne);
    W.item_proof = (uint32_t*)k.take(4 * items_cap), W.item_rep = (uint32_t*)k.take(4 * items_cap), W.item_rank = (uint32_t*)k.take(4 * items_cap);
    W.T1proj = k.soa3(items_cap), W.T1x = k.soa(items_cap), W.T1y = k.soa(items_cap);
    W.padd_c = (uint32_t*)k.take(4 * 18 * items_cap);
    W.la = k.list((size_t)C * (2 + 2 * sec));
    uint8_t* const lb_begin = (uint8_t*)k.take(0);   // list B's seven arrays lie back to back from here
    const size_t lb_off0 = k.off;
    W.lb = k.list(items_cap * LB_SLOTS);
    const size_t lb_bytes = k.off - lb_off0;
    W.lc = k.list((size_t)C * 4 * n);
    W.gk_x = (uint32_t*)k.take(12 * (size_t)C);
    W.gk_coef = k.soa((size_t)(n + 1) * C);
    gk_am = k.soa((size_t)n * C);
    // fused fold (n >= 3): gk_bufA holds the tile polynomials, (T+1) coefs x N/2^T tiles per proof; the per-level
    // fallback for tiny rings ping-pongs G*N elements between gk_bufA and gk_bufB
    uint64_t g = std::max<uint64_t>(1, std::min<uint64_t>(C, ((uint64_t)1 << 16) / N));
    W.gk_group = (uint32_t)g;
    uint32_t T = c->gk_etab ? 8 : std::min<uint32_t>(n, 12);
    uint64_t tile_elems = (uint64_t)(T + 1) * C * (N >> T);
    W.gk_etab = c->gk_etab;
    W.gk_kdig = c->gk_mfma ? c->gk_kdig : nullptr;
    W.gk_edig = c->gk_mfma ? c->gk_edig : nullptr;
    W.gk_adig = c->gk_edig ? (int8_t*)k.take(gkm_asub_frag_bytes(C)) : nullptr;
    W.gk_toff = (uint32_t*)k.take(4 * 264);
    W.gk_asub = c->gk_etab ? (uint32_t*)k.take(36 * 256 * (size_t)C) : nullptr;
    W.gk_order = (uint32_t*)k.take(4 * (size_t)C);
    W.gk_goff = (uint32_t*)k.take(4 * 264);
    W.gk_bufA = (uint32_t*)k.take(36 * std::max<uint64_t>(g * N, tile_elems));
    W.gk_bufB = (uint32_t*)k.take(36 * std::max<uint64_t>(g * N, (uint64_t)(n + 1) * C * std::max<uint64_t>(1, (N >> T) / gk_finish_gsz(T, (uint32_t)(N >> T)))));
    W.rng.exc_idx = (uint32_t*)k.take(4 * RNG_MAX_EXC * (size_t)C);
    W.rng.exc_flags = (uint32_t*)k.take(4 * RNG_MAX_EXC * (size_t)C);
    W.rng.exc_cnt = (uint32_t*)k.take(4 * (size_t)C);
    W.rng_fill = (uint32_t*)k.take(32 * (size_t)(3 + 44 * sec + 5 * n + RNG_MAX_EXC) * C);
    {
        // the Exp challenge's message and schedule buffers (k_hash.hip: k_exph_*): their own memory for chunks of up to EXPH_MAXP proofs (the verifier's
        // small chunks use them while list B may be live); a prover chunk of any size borrows LIST B, which nothing touches before stage 2 -- 80 KB of
        // the 397 KB a proof's list B takes, so the three-kernel path costs no HBM
        const size_t blocks = (2 * 67 + (size_t)sec * (65 + 2 * 67) + 9 + 63) / 64, np = std::min<size_t>(C, EXPH_MAXP);
        W.exph_msg = (uint8_t*)k.take(np * blocks * 64), W.exph_wk = (uint32_t*)k.take(np * blocks * 256), W.exph_cap = (uint32_t)np;
        const size_t big_msg = ((size_t)C * blocks * 64 + 255) & ~(size_t)255;
        if (lb_bytes >= big_msg + (size_t)C * blocks * 256) W.exph_big_msg = lb_begin, W.exph_big_wk = (uint32_t*)(lb_begin ? lb_begin + big_msg : nullptr), W.exph_big_cap = C;
        else W.exph_big_msg = nullptr, W.exph_big_wk = nullptr, W.exph_big_cap = 0;
    }
    W.ring = Soa{c->ring_mem, (uint32_t)N};
    return k.off + 256;
}
// hipMalloc for a workspace that gives up the OPTIONAL per-ring tables before it gives up itself: the per-key tables (35 GB at 2^17 keys)
// are taken at zk_ctx_set_ring, before anybody knows how large the chunks will be; if a lane's arena no longer fits, the tables go and
// every proof takes the per-proof tables of R -- slower, same bytes.
hipError_t malloc_or_shed(zk_ctx* c, void** p, size_t bytes) {
    hipError_t e = hipMalloc(p, bytes);
    if (e == hipSuccess || !c->ktab) return e;
    (void)hipGetLastError();
    (void)hipDeviceSynchronize();
    hipFree(c->ktab), hipFree(c->ktab_ok);
    c->ktab = nullptr, c->ktab_ok = nullptr;
    for (auto& L : c->pl) L.W.ktab = nullptr, L.W.ktab_ok = nullptr;
    return hipMalloc(p, bytes);
}
zk_status ensure_workspace(zk_ctx* c, uint32_t C, uint32_t nlanes) {
    uint32_t sec = c->P.sec, n = c->n;
    if (!(c->ws_C == C && c->ws_sec == sec && c->ws_n == n)) {
        for (auto& L : c->pl) L.ready = false;
        c->ws_C = C, c->ws_sec = sec, c->ws_n = n;
    }
    for (uint32_t l = 0; l < nlanes && l < ZK_MAX_LANES; l++) {
        auto& L = c->pl[l];
        if (!L.ready) {
            size_t need = carve(c, L.W, L.gk_am, nullptr, C, sec, n, c->N);
            if (need > L.arena_bytes) {
                if (L.arena) HIPCHK(c, hipFree(L.arena));
                L.arena = nullptr, L.arena_bytes = 0;
                HIPCHK(c, malloc_or_shed(c, &L.arena, need));
                L.arena_bytes = need;
            }
            carve(c, L.W, L.gk_am, (uint8_t*)L.arena, C, sec, n, c->N);
            if (!L.d_totals) HIPCHK(c, hipMalloc(&L.d_totals, 64));
            const size_t hs = 64 + 16 * ((size_t)C + 2);
            if (hs > L.h_scan_bytes) {
                if (L.h_scan) HIPCHK(c, hipHostFree(L.h_scan));
                L.h_scan = nullptr, L.h_scan_bytes = 0;
                HIPCHK(c, hipHostMalloc(&L.h_scan, hs, hipHostMallocMapped | hipHostMallocCoherent));
                L.h_scan_bytes = hs;
            }
            L.ready = true;
        }
        L.W.ring = Soa{c->ring_mem, (uint32_t)c->N};
        L.W.hardened = c->mode == ZK_MODE_HARDENED, L.W.ring_digest = c->ring_digest;
        L.W.wire = wire_make(c->wire == ZK_WIRE_ZKA1P);
        L.W.ktab = c->ktab, L.W.ktab_ok = c->ktab_ok;
        L.W.gk_kdig = c->gk_mfma ? c->gk_kdig : nullptr;
        L.W.gk_edig = c->gk_mfma && L.W.gk_adig ? c->gk_edig : nullptr;
    }
    return ZK_OK;
}

// ------------------------------------------------------------------ page-locked host buffers
// zk_prove_batch / zk_verify_batch move ~169 KB per proof across PCIe.  From pageable memory the runtime stages every
// copy through its own bounce buffers (measured 8.6 GB/s, one blocking copy); from page-locked memory the copies are DMA
// transfers on their own stream, chunk by chunk, under the kernels of the neighbouring chunks.
// ---- "slow pages".  Measured on MI355X boxes (tools/exp_link_state.py, tools/exp_pool_first_call.py): a page-locked
// buffer allocated shortly after another process (or this one) released gigabytes of page-locked memory can come out of the allocator
// with pages that the device's DMA engines reach at HALF rate -- 30 instead of 57 GB/s, both directions, whatever stream, SDMA or blit,
// hipHostMalloc or mmap + hipHostRegister, huge pages or not, near or far NUMA node, with unchanged device clocks and link speed --
// while a buffer allocated next to it at the same moment copies at full rate.  The property sticks to the buffer for as long as it
// lives.  A process whose output buffer drew such pages ran every host-pointer call at 160 k instead of 262 k proofs/s ("first
// synchronous pool call anomaly" of round 3).  So the allocators below measure what they hand out and try again -- holding the slow
// buffer meanwhile, so that the allocator cannot return the same pages -- up to ZK_ALLOC_TRIES times.  ZKATTEST_HOST_ALLOC_PROBE=0
// switches the check off.
#define ZK_ALLOC_TRIES 4
// slowest device-to-host rate (GB/s) over up to three 64 MiB windows of the page-locked range [p, p + bytes); 0 = could not measure
float pinned_d2h_rate(void* p, size_t bytes) {
    const size_t win = 64u << 20;
    if (bytes < (4u << 20)) return 0.f;
    const size_t w = std::min(win, bytes);
    void* d = nullptr;
    hipStream_t s = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    float worst = 0.f;
    if (hipMalloc(&d, w) == hipSuccess && hipStreamCreateWithFlags(&s, hipStreamNonBlocking) == hipSuccess && hipEventCreate(&e0) == hipSuccess &&
        hipEventCreate(&e1) == hipSuccess) {
        const size_t offs[3] = {0, ((bytes - w) / 2) & ~(size_t)4095, (bytes - w) & ~(size_t)4095};
        const int nw = bytes >= 3 * w ? 3 : 1;
        (void)hipMemcpyAsync(p, d, 1u << 20, hipMemcpyDeviceToHost, s);   // the stream's first copy sets its queue up
        for (int k = 0; k < nw; k++) {
            float ms = 0.f;
            if (hipEventRecord(e0, s) != hipSuccess || hipMemcpyAsync((uint8_t*)p + offs[k], d, w, hipMemcpyDeviceToHost, s) != hipSuccess ||
                hipEventRecord(e1, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess || ms <= 0.f) {
                worst = 0.f;
                break;
            }
            const float r = (float)(w / 1e6 / ms);
            worst = k == 0 ? r : std::min(worst, r);
        }
    }
    (void)hipGetLastError();
    if (e0) hipEventDestroy(e0);
    if (e1) hipEventDestroy(e1);
    if (s) hipStreamDestroy(s);
    if (d) hipFree(d);
    return worst;
}
bool host_alloc_probe_enabled() {
    const char* e = getenv("ZKATTEST_HOST_ALLOC_PROBE");
    return !(e && !strcmp(e, "0"));
}
// What a page-locked copy should reach on the current device's PCIe link: 0.89 (the share measured on these boxes: 57 of 64 GB/s) of
// speed x width from sysfs (current_link_speed "32.0 GT/s PCIe", current_link_width "16"); 0 when sysfs does not say.
static float link_expected_gbps() {
    int dev = 0;
    char bus[64] = {0}, buf[128];
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetPCIBusId(bus, sizeof bus, dev) != hipSuccess) {
        (void)hipGetLastError();
        return 0.f;
    }
    for (char* q = bus; *q; q++) *q = (char)tolower(*q);
    auto rd = [&](const char* leaf) -> float {
        std::string path = std::string("/sys/bus/pci/devices/") + bus + "/" + leaf;
        FILE* f = fopen(path.c_str(), "r");
        if (!f) return 0.f;
        float v = fgets(buf, sizeof buf, f) ? (float)atof(buf) : 0.f;
        fclose(f);
        return v;
    };
    const float gts = rd("current_link_speed"), width = rd("current_link_width");
    if (gts < 2.f || width < 1.f) return 0.f;
    return 0.89f * gts * width / 8.f * (gts >= 8.f ? 128.f / 130.f : 0.8f);
}
// Allocates with `alloc`, measures, and keeps the fastest of up to ZK_ALLOC_TRIES candidates; the rejected ones are held until the choice is
// made (so that the allocator cannot hand the same pages out again) and then released with `release`.  A candidate is good when it reaches
// 0.8 of what the link should carry (sysfs); where sysfs does not say, when it is within 25 % of the best rate this process has measured
// on any page-locked buffer (the first large allocation then draws a second candidate for comparison).  Slow pages are a transient of
// the allocator (recently released page-locked memory), so a pause precedes every further attempt.
static std::atomic<float> g_best_pinned_rate{0.f};   // (both allocators may run on several host threads: zk_pool_host_alloc, callers' own threads)
// Rejected candidates are HELD while the next one is drawn, so that the allocator cannot hand the same slow pages out again -- but never more than
// ZKATTEST_HOST_ALLOC_HOLD_GB (default 48) of page-locked memory at once, this candidate included: beyond that the oldest rejected one goes back first (a
// multi-GB request must not fail, or stall the host, because three copies of it were pinned beside it).
static size_t host_alloc_hold_limit() {
    const char* e = getenv("ZKATTEST_HOST_ALLOC_HOLD_GB");
    const long gb = e ? atol(e) : 48;
    return (size_t)(gb > 0 ? gb : 1) << 30;
}
void* alloc_fast_pinned(size_t bytes, const std::function<void*()>& alloc, const std::function<void(void*)>& release) {
    const bool probe = host_alloc_probe_enabled() && bytes >= (64u << 20);
    const float expected = probe ? link_expected_gbps() : 0.f;
    void *best = nullptr, *held[ZK_ALLOC_TRIES] = {};
    float best_r = -1.f;
    int nheld = 0;
    const size_t hold_limit = host_alloc_hold_limit();
    for (int t = 0; t < ZK_ALLOC_TRIES; t++) {
        while (nheld > 0 && ((size_t)nheld + (best ? 1 : 0) + 1) * bytes > hold_limit) release(held[--nheld]);
        void* p = alloc();
        if (!p) break;
        const float r = probe ? pinned_d2h_rate(p, bytes) : 0.f;
        float seen = g_best_pinned_rate.load(std::memory_order_relaxed);
        const bool had_yardstick = seen > 0.f;
        while (r > seen && !g_best_pinned_rate.compare_exchange_weak(seen, r, std::memory_order_relaxed)) {
        }
        seen = seen > r ? seen : r;
        if (r > best_r) {
            if (best) held[nheld++] = best;
            best = p, best_r = r;
        } else {
            held[nheld++] = p;
        }
        if (!probe || r == 0.f) break;   // not measuring (or could not)
        if (expected > 0.f ? best_r >= 0.8f * expected : ((had_yardstick || t > 0) && best_r >= 0.75f * seen)) break;
        if (t + 1 < ZK_ALLOC_TRIES) usleep(300000u * (unsigned)(t + 1));
    }
    for (int i = 0; i < nheld; i++) release(held[i]);
    return best;
}
extern "C" void* zk_host_alloc(size_t bytes) {
    return alloc_fast_pinned(
        bytes,
        [&]() -> void* {
            void* p = nullptr;
            if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) {
                (void)hipGetLastError();
                return nullptr;
            }
            return p;
        },
        [](void* p) { hipHostFree(p); });
}
extern "C" void zk_host_free(void* p) {
    if (p) hipHostFree(p);
}
bool host_ptr_is_pinned(const void* p) {
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) {
        (void)hipGetLastError();  // pageable memory is "invalid value" to the runtime: not an error here
        return false;
    }
    return a.type == hipMemoryTypeHost;
}
void* stream_take_spare_dev(zk_ctx* c, size_t bytes, size_t* got);   // api_stream.hip: staging buffers of finished streamed jobs
zk_status ensure_io_buf(zk_ctx* c, size_t bytes) {
    if (bytes <= c->io_bytes) return ZK_OK;
    if (c->io_buf) hipFree(c->io_buf);
    c->io_buf = nullptr, c->io_bytes = 0;
    size_t got = 0;
    if (void* p = stream_take_spare_dev(c, bytes, &got)) {
        c->io_buf = p, c->io_bytes = got;
        return ZK_OK;
    }
    HIPCHK(c, hipMalloc(&c->io_buf, bytes));
    c->io_bytes = bytes;
    return ZK_OK;
}
zk_status ensure_in_buf(zk_ctx* c, size_t bytes) {
    if (bytes <= c->in_bytes) return ZK_OK;
    if (c->in_buf) hipFree(c->in_buf);
    c->in_buf = nullptr, c->in_bytes = 0;
    bytes += bytes / 4 + 4096;
    HIPCHK(c, hipMalloc(&c->in_buf, bytes));
    c->in_bytes = bytes;
    return ZK_OK;
}
zk_status ensure_h_stage(zk_ctx* c, size_t bytes) {
    if (!c->in_ready) HIPCHK(c, hipEventCreateWithFlags(&c->in_ready, hipEventDisableTiming));
    if (bytes <= c->h_stage_bytes) return ZK_OK;
    if (c->h_stage) {
        memset(c->h_stage, 0, c->h_stage_bytes);
        hipHostFree(c->h_stage);
    }
    c->h_stage = nullptr, c->h_stage_bytes = 0;
    bytes += bytes / 4 + 4096;
    HIPCHK(c, hipHostMalloc((void**)&c->h_stage, bytes, hipHostMallocDefault));
    c->h_stage_bytes = bytes;
    return ZK_OK;
}
zk_status ensure_copy_stream(zk_ctx* c) {   // the streams exist since zk_ctx_create; their events are made on first use
    for (auto& L : c->pl)
        if (!L.copy_ev) HIPCHK(c, hipEventCreateWithFlags(&L.copy_ev, hipEventDisableTiming));
    return ZK_OK;
}

// ------------------------------------------------------------------ the prover pipeline
// One call = one ProveJob: the chunk plan of its B proofs and the two stage functions.  The synchronous entry points run one job
// to completion; the streamed ones (zk_prove_submit / zk_prove_wait below) keep several jobs queued and let the stage-1 look-ahead
// run across the boundary between consecutive jobs, so the device never drains between calls.
//
// Every chunk has two stages.  Stage 1 (front end .. Exp challenge) needs nothing from other chunks; stage 2 starts with
// the scan, which needs the output cursor, i.e. the byte count of all earlier chunks of the job, and the host has to read the
// item count back before it can size the PointAdd launches.  Stage 1 of the next NL - 1 chunks is enqueued on the other
// lanes BEFORE the host blocks on a chunk's scan, so no stream runs dry while the host waits.
// host_sink: page-locked destination of the proof bytes (or nullptr): every slice is copied out on its lane's copy stream as soon
// as its last kernel has run, while the next slices / chunks are being proved.
static zk_status ensure_side_stream(zk_ctx* c, zk_ctx::ProveLane& PL) {
    if (!PL.side) HIPCHK(c, hipStreamCreateWithFlags(&PL.side, hipStreamNonBlocking));
    if (!PL.side_fork) HIPCHK(c, hipEventCreateWithFlags(&PL.side_fork, hipEventDisableTiming));
    if (!PL.side_done) HIPCHK(c, hipEventCreateWithFlags(&PL.side_done, hipEventDisableTiming));
    return ZK_OK;
}
zk_status ProveJob::stage1(uint64_t chunk_no) {
    const ChunkPlan& cp = plan[chunk_no];
    const DevParams& P = c->P;
    const uint32_t lane = lane_of(chunk_no);
    Workspace& W = c->pl[lane].W;
    hipStream_t s = c->pl[lane].stream;
    const uint64_t first = cp.first;
    const uint32_t cnt = cp.cnt;
    const ChunkIn in{d_msg + 32 * first, d_sig + 64 * first, d_pk + 64 * first, d_which + first, cnt};
    const uint32_t nblk = 3 + 44 * W.sec + 5 * W.n + RNG_MAX_EXC;
    Pending& pd = pend[lane];
    {
        c->pl[lane].last_cnt = cnt;
        if (io_dbg) fprintf(stderr, "host %7.1f ms: stage1 of chunk %u (%u proofs) -> lane %u\n", host_ms() - host_t0, (uint32_t)chunk_no, cnt, lane);
        if (inputs_ready) HIPCHK(c, hipStreamWaitEvent(s, inputs_ready, 0));
        W.rng.seeds = d_rng, W.rng.stream = d_rng, W.rng.stride_blocks = stride, W.rng.mode = rng_mode, W.rng.sec = (int)W.sec;
        W.rng.proof_base = (uint32_t)first;
        {
            MaybeScope t(timed, c, "rng_prepass", s);
            launch_rng_prepass(s, W, cnt, 0, 3 + 4 * W.sec + RNG_MAX_EXC, nblk, rng_mode == 0 ? W.rng_fill : nullptr, false);  // + margin: rejected fills shift later draws
        }
        pd.lane = lane, pd.cnt = cnt, pd.first = first, pd.in = in, pd.nblk = nblk;
        pd.Wgen = W;  // RNG view of the generator (seed mode) for the second prepass stage
        if (rng_mode == 0)   // from here on the chunk reads the fills the prepass wrote
            W.rng.mode = 1, W.rng.stream = (const uint8_t*)W.rng_fill, W.rng.stride_blocks = nblk, W.rng.proof_base = 0;
        {
            MaybeScope t(timed, c, "p256_front", s);
            launch_front(s, P, W, in);
        }
        {
            MaybeScope t(timed, c, "p256_rtab", s);
            launch_rtab(s, W, cnt, RTAB_PROVE_BITS, W.ktab ? W.kt_use : nullptr);   // proofs on the key-table path need no table of R
        }
    }
    {
        MaybeScope t(timed, c, "p256_exp_commit", s);
        launch_exp_commit(s, P, W, cnt);
    }
    const uint32_t na = cnt * (2 + 2 * W.sec);
    {
        {
            MaybeScope t(timed, c, "p256_normalize", s);
            auto& PL = c->pl[lane];
            const bool beside = plan.size() == 1 && cnt <= ZK_PROVE_SIDE_MAX;   // a small one-chunk call: the two lists (a workgroup inversion each) side by side
            if (beside) {
                if (timed) c->timing_forked = true;
                zk_status zs = ensure_side_stream(c, PL);
                if (zs) return zs;
                HIPCHK(c, hipEventRecord(PL.side_fork, s));
                HIPCHK(c, hipStreamWaitEvent(PL.side, PL.side_fork, 0));
            }
            launch_p256_normalize(beside ? PL.side : s, W.Aproj, cnt * (W.sec + 1), W.Ax, W.Ay, W.st, W.sec + 1, 0, nullptr);
            if (beside) HIPCHK(c, hipEventRecord(PL.side_done, PL.side));
            launch_p256_normalize(s, W.Tproj, cnt * (W.sec + 1), W.Tx, W.Ty, W.st, W.sec + 1, ZK_ST_T_INF_LATE, nullptr);
            if (beside) HIPCHK(c, hipStreamWaitEvent(s, PL.side_done, 0));
        }
        {
            MaybeScope t(timed, c, "scalars", s);
            launch_lista_scalars(s, W, cnt);
        }
    }
    {
        MaybeScope t(timed, c, "tom_commit", s);
        launch_tom_commit(s, P, W.la, na, 1, 1);
    }
    {
        {
            MaybeScope t(timed, c, "tom_normalize", s);
            launch_tom_normalize(s, W.la, na, 0, 1, 1);
        }
        {
            MaybeScope t(timed, c, "hash", s);
            launch_exp_challenge(s, W, cnt);
        }
    }
    return ZK_OK;
}
// Stage 1 of every chunk the look-ahead reaches (chunks next_s1 .. upto - 1), each on its lane's stream.
static zk_status prove_stage1_upto(ProveJob& J, uint64_t upto) {
    zk_status zs = ZK_OK;
    const uint64_t a = J.next_s1, b = std::min<uint64_t>(upto, J.plan.size());
    for (uint64_t k = a; k < b && !zs; k++) zs = J.stage1(k);
    if (b > a) J.next_s1 = b;
    return zs;
}
// Stage 2.  Order: scan -> fixed part and rep heads -> the whole Groth-Kohlweiss phase -> the PointAdd phase (80 % of the
// bytes) in proof-aligned slices.  None of the three depends on another (they share the chunk's RNG fills and the list-A
// results), and with this order every byte of a proof is final as soon as the slice holding its PointAdd items is done: a
// page-locked sink then receives the slice's proofs by DMA while the next slice is being computed.  A slice runs the
// unchanged per-item kernels on a view of the workspace whose item-indexed arrays start at the slice's first item.
zk_status ProveJob::stage2(uint64_t chunk_no) {
    const DevParams& P = c->P;
    Pending& pd = pend[lane_of(chunk_no)];
    Workspace& W = c->pl[pd.lane].W;
    hipStream_t s = c->pl[pd.lane].stream;
    const Soa& gk_am = c->pl[pd.lane].gk_am;
    uint32_t* d_totals = c->pl[pd.lane].d_totals;
    const uint32_t cnt = pd.cnt, nblk = pd.nblk;
    const uint64_t first = pd.first;
    const ChunkIn& in = pd.in;
    const Workspace& Wgen = pd.Wgen;
    // page-locked read-back area of the lane: totals, then the prefix sums the slices need
    uint32_t* totals = (uint32_t*)c->pl[pd.lane].h_scan;
    uint64_t* h_out_base = (uint64_t*)((uint8_t*)c->pl[pd.lane].h_scan + 64);
    uint32_t* h_item_base = (uint32_t*)(h_out_base + (size_t)c->ws_C + 2);
    {
        MaybeScope t(timed, c, "scan", s);
        launch_scan(s, W, cnt, cursor, out_cap, d_out_off, d_status, d_totals, first);
    }
    // proofs per slice (zk_ctx_set_slice; 0 = automatic): 4096 with a page-locked sink -- and 512 / 1024 for the small chunks of short calls, whose output (169 KB per proof:
    // 15 ms for 4096 proofs) is a large part of their time and is exposed at the end of the call or delays the lane's next chunk: a call that is ONE chunk of at most 2048 / 8192
    // proofs (tools/slice_sweep.sh: -11 % at 1024 proofs per call, -20 % at 2048, -25 % at 4096, -15 % at 8192), and the chunks of at most 2048 / 4096 proofs of a call of at
    // most 32 768 (the default chunk of 4096 on two lanes: -19 % at 8192 proofs per call, -11 % at 16 384, -5 % at 32 768).  Not beyond: chunks of 8192 in a longer call lose
    // 1 %, and the tapered first and last chunks of a 65 536-proof call lose 6-12 % -- the neighbouring chunks hide the transfers there, and small grids cost more than they gain
    // (profiles/r06_ab_variants.txt (22)).
    const uint32_t small_max = plan.size() == 1 ? 8192u : B <= 32768 ? 4096u : 0u;
    const uint32_t S = c->slice ? c->slice : !host_sink ? 0u : cnt <= small_max ? (cnt <= 2048 ? 512u : 1024u) : 4096u;
    const bool sliced = S && cnt > S;
    const bool last_chunk = first + cnt == B && !more_follows;
    launch_words_to_host(s, totals, d_totals, 4);
    if (sliced) {   // slice boundaries: the chunk's item and byte prefix sums
        launch_words_to_host(s, h_item_base, W.item_base, (size_t)cnt + 1);
        launch_words_to_host(s, h_out_base, W.out_base, 2 * ((size_t)cnt + 1));
    }
    // A small chunk leaves the GPU idle and its phases are chains of latencies: the membership phase (list C, its own hash and responses) runs on the
    // lane's side stream beside the PointAdd phase (list B); they share only what stage 1 and the scan left behind.
    auto& PL = c->pl[pd.lane];
    const bool beside = !sliced && plan.size() == 1 && cnt <= ZK_PROVE_SIDE_MAX;   // (chunks of a longer job overlap each other on the lanes already: doing it for every
                                                                                    // chunk measured -0.6 % proofs/s, profiles/r06_ab_variants.txt (9))
    hipStream_t sg = s;
    if (beside) {
        if (timed) c->timing_forked = true;
        zk_status zs = ensure_side_stream(c, PL);
        if (zs) return zs;
        sg = PL.side;
    }
    // part 0: all of it; 1: everything that stays inside the workspace (ring fold, commitments, challenge); 2: the responses and points written into the proofs
    auto membership_phase = [&](int part) -> zk_status {
        if (part != 2) {
            MaybeScope t(timed, c, "gk_fold", sg);
            launch_gk_scalars_fold(sg, W, in, gk_am);
            launch_gk_cd_scalars(sg, W, cnt);
        }
        if (part != 2) {
            MaybeScope t(timed, c, "tom_commit", sg);
            launch_tom_commit(sg, P, W.lc, cnt * 4 * W.n, 1, 1);
        }
        if (part != 2) {
            MaybeScope t(timed, c, "tom_normalize", sg);
            launch_tom_normalize(sg, W.lc, cnt * 4 * W.n, 0, 1, 1);
        }
        if (part != 2) {
            MaybeScope t(timed, c, "hash", sg);
            launch_gk_hash(sg, W, cnt, in.msg);
        }
        if (part == 1) return ZK_OK;
        {
            MaybeScope t(timed, c, "respond_write", sg);
            launch_gk_respond(sg, W, in, d_out + cursor);
        }
        if (beside) HIPCHK(c, hipEventRecord(PL.side_done, sg));
        return ZK_OK;
    };
    // A small one-chunk call: what needs nothing from the host -- the second RNG prepass and the membership phase up to its challenge -- is on the device BEFORE the host waits
    // for the scan's totals: the device does not fall idle during the host's round trip, and the side stream's chain starts ~0.1 ms earlier.
    const bool early = beside;
    if (early) {
        {
            MaybeScope t(timed, c, "rng_prepass", s);
            launch_rng_prepass(s, Wgen, cnt, 3 + 4 * W.sec + RNG_MAX_EXC, nblk, nblk, rng_mode == 0 ? W.rng_fill : nullptr, true);
        }
        HIPCHK(c, hipEventRecord(PL.side_fork, s));
        HIPCHK(c, hipStreamWaitEvent(sg, PL.side_fork, 0));
        if (zk_status zs = membership_phase(1)) return zs;
    }
    if (io_dbg) fprintf(stderr, "host %7.1f ms: stage2 lane %u waits for its scan\n", host_ms() - host_t0, pd.lane);
    HIPCHK(c, hipStreamSynchronize(s));
    if (io_dbg) fprintf(stderr, "host %7.1f ms: scan done\n", host_ms() - host_t0);
    if (totals[1]) {
        c->err = "output buffer too small";
        sync_lanes();
        if (early) (void)hipStreamSynchronize(sg);   // (the membership phase's first part is on the side stream already: workspace only)
        return ZK_E_BUFFER;
    }
    const uint32_t items_all = totals[0];
    if (items_all > W.items_cap) {
        // cannot happen for hash-derived challenges (cap = mean + 8 sigma) unless chunk*sec is tiny, where cap = chunk*sec
        c->err = "zero-bit rep count exceeds workspace capacity";
        sync_lanes();
        if (early) (void)hipStreamSynchronize(sg);
        return ZK_E_BUFFER;
    }
    uint8_t* out = d_out + cursor;
    const uint64_t chunk_bytes = (uint64_t)totals[2] | ((uint64_t)totals[3] << 32);
    {
        MaybeScope t(timed, c, "scan", s);
        launch_items(s, W, cnt);
    }
    if (!early) {
        MaybeScope t(timed, c, "rng_prepass", s);  // second stage: only the blocks a proof with z zero bits can reach
        launch_rng_prepass(s, Wgen, cnt, 3 + 4 * W.sec + RNG_MAX_EXC, nblk, nblk, rng_mode == 0 ? W.rng_fill : nullptr, true);
    }
    {
        MaybeScope t(timed, c, "respond_write", s);
        launch_write_fixed(s, W, cnt, out);
    }
    if (zk_status zs = membership_phase(early ? 2 : 0)) return zs;
    std::vector<ChunkPlan> slices;
    if (sliced) slices = make_chunk_plan(cnt, S, 1, host_sink != nullptr && !more_follows, last_chunk ? ZK_SLICE_MIN / 2 : ZK_SLICE_MIN);   // the call's very last slices stay exposed
    else slices.push_back({0, cnt});
    for (const ChunkPlan& sl : slices) {
        const uint32_t p0 = (uint32_t)sl.first, p1 = p0 + sl.cnt;
        const uint32_t i0 = sliced ? h_item_base[p0] : 0, i1 = sliced ? h_item_base[p1] : items_all;
        const uint32_t items = i1 - i0;
        if (items) {
            Workspace Ws = W;   // the slice's view: item-indexed arrays start at item i0
            Ws.item_proof += i0, Ws.item_rep += i0, Ws.item_rank += i0, Ws.padd_c += (size_t)18 * i0;
            for (Soa* a : {&Ws.T1proj.x, &Ws.T1proj.y, &Ws.T1proj.z, &Ws.T1x, &Ws.T1y, &Ws.lb.v, &Ws.lb.r, &Ws.lb.proj.x, &Ws.lb.proj.y, &Ws.lb.proj.z,
                           &Ws.lb.ax, &Ws.lb.ay})
                a->p += i0;
            {
                MaybeScope t(timed, c, "p256_t1", s);
                launch_t1(s, Ws, items);
            }
            {
                MaybeScope t(timed, c, "p256_normalize", s);
                launch_p256_normalize(s, Ws.T1proj, items, Ws.T1x, Ws.T1y, Ws.st, 1, ZK_E_T1_INF, Ws.item_proof);
            }
            {
                MaybeScope t(timed, c, "scalars", s);
                launch_padd_scalars(s, P, Ws, items);
            }
            {
                MaybeScope t(timed, c, "tom_commit", s);
                launch_tom_commit_listb(s, P, Ws.lb, items, Ws.items_cap);
            }
            {
                MaybeScope t(timed, c, "tom_normalize", s);
                launch_tom_normalize(s, Ws.lb, items * LB_COMMITS, 0, items, 0, Ws.items_cap);
            }
            {
                MaybeScope t(timed, c, "tom_derived", s);
                launch_padd_derived(s, Ws, items);
                launch_tom_normalize(s, Ws.lb, items * 5, LB_COMMITS, items, 0, Ws.items_cap);
            }
            {
                MaybeScope t(timed, c, "hash", s);
                launch_padd_hash(s, P, Ws, items);
            }
            {
                MaybeScope t(timed, c, "respond_write", s);
                launch_padd_respond(s, Ws, items, out);
                launch_write_padd_points(s, Ws, items, out);
            }
        }
        if (beside) HIPCHK(c, hipStreamWaitEvent(s, PL.side_done, 0));   // (not sliced: the one pass of this loop)
        if (host_sink) {   // every byte of proofs [p0, p1) is final: DMA them out behind the next slice's kernels
            const uint64_t b0 = sliced ? h_out_base[p0] : 0, b1 = sliced ? h_out_base[p1] : chunk_bytes;
            if (b1 > b0) {
                hipEvent_t ev = c->pl[pd.lane].copy_ev;
                IoRec r{};
                if (io_dbg) {
                    hipEventCreate(&r.ready), hipEventCreate(&r.c0), hipEventCreate(&r.c1);
                    r.bytes = b1 - b0, r.chunk = (uint32_t)(first / (C ? C : 1)), r.lane = pd.lane;
                    hipEventRecord(r.ready, s);
                }
                HIPCHK(c, hipEventRecord(ev, s));
                HIPCHK(c, hipStreamWaitEvent(c->pl[pd.lane].copy_stream, ev, 0));
                if (io_dbg) hipEventRecord(r.c0, c->pl[pd.lane].copy_stream);
                const uint64_t piece = 512ull << 20;   // one DMA command moves at most this much
                for (uint64_t o = b0; o < b1; o += piece)
                    HIPCHK(c, hipMemcpyAsync(host_sink + cursor + o, out + o, std::min<uint64_t>(piece, b1 - o), hipMemcpyDeviceToHost, c->pl[pd.lane].copy_stream));
                if (io_dbg) {
                    hipEventRecord(r.c1, c->pl[pd.lane].copy_stream);
                    iorecs.push_back(r);
                }
            }
        }
    }
    {
        MaybeScope t(timed, c, "respond_write", s);
        launch_status_out(s, W, cnt, d_status, first);   // late (cryptographically negligible) errors included: after the last slice
    }
    cursor += chunk_bytes;
    if (io_dbg) fprintf(stderr, "host %7.1f ms: stage2 lane %u enqueued\n", host_ms() - host_t0, pd.lane);
    return ZK_OK;
}

static zk_status prove_device(zk_ctx* c, uint64_t B, const uint8_t* d_msg, const uint8_t* d_sig, const uint8_t* d_pk, const uint32_t* d_which,
                              int rng_mode, const uint8_t* d_rng, uint64_t stride, uint8_t* d_out, uint64_t out_cap, uint64_t* d_out_off,
                              int32_t* d_status, uint8_t* host_sink = nullptr, hipEvent_t inputs_ready = nullptr) {
    if (!c->params_set || !c->N) return ZK_E_BUFFER;
    if (rng_mode != ZK_RNG_SEED && rng_mode != ZK_RNG_STREAM) return ZK_E_ARG;
    if (c->stream_busy) {
        c->err = "streamed jobs are in flight on this context (zk_prove_wait / zk_verify_wait them first)";
        return ZK_E_ARG;
    }
    ProveJob J;
    J.c = c, J.B = B, J.d_msg = d_msg, J.d_sig = d_sig, J.d_pk = d_pk, J.d_which = d_which, J.rng_mode = rng_mode, J.d_rng = d_rng, J.stride = stride;
    J.d_out = d_out, J.out_cap = out_cap, J.d_out_off = d_out_off, J.d_status = d_status, J.host_sink = host_sink, J.timed = zk_timed(c, B), J.inputs_ready = inputs_ready;
    J.C = (uint32_t)std::min<uint64_t>(c->chunk, B ? B : 1);
    J.plan = make_chunk_plan(B, J.C, host_sink != nullptr && c->host_taper ? (c->host_taper == 1 ? c->lanes : c->host_taper) : 1, false);
    J.NL = (uint32_t)std::min<size_t>(c->lanes, J.plan.size() ? J.plan.size() : 1);  // chunks rotate over NL streams / workspaces
    zk_status zs = ensure_workspace(c, J.C, J.NL);
    if (zs) return zs;
    timing_begin(c);
    if (B == 0) {
        HIPCHK(c, hipMemsetAsync(d_out_off, 0, 8, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        return ZK_OK;
    }
    const char* io_env = getenv("ZK_IO_DEBUG");   // the host / GPU timeline of the call on stderr (tools/exp_io_timeline.py); 2: of a device-pointer call as well
    J.io_dbg = io_env && (host_sink || atoi(io_env) >= 2);   // 2: the host / GPU timeline of a device-pointer call as well
    J.host_t0 = ProveJob::host_ms();
    if (J.io_dbg) {
        hipEventCreate(&J.io_t0);
        hipEventRecord(J.io_t0, c->stream);
    }
    const uint64_t nchunks = J.plan.size();
    for (uint64_t k = 0; k < nchunks && !zs; k++) {
        zs = prove_stage1_upto(J, k + J.NL);
        if (!zs) zs = J.stage2(k);
    }
    hipError_t e_sync = J.sync_lanes();   // nothing of this call may still be running (or writing into the caller's buffer) when it returns
    if (host_sink)
        for (uint32_t l = 0; l < J.NL; l++) {
            hipError_t e2 = hipStreamSynchronize(c->pl[l].copy_stream);
            if (e_sync == hipSuccess) e_sync = e2;
        }
    if (J.io_dbg) {
        for (auto& r : J.iorecs) {
            float a = 0, b = 0, d = 0;
            hipEventElapsedTime(&a, J.io_t0, r.ready), hipEventElapsedTime(&b, J.io_t0, r.c0), hipEventElapsedTime(&d, J.io_t0, r.c1);
            fprintf(stderr, "io: first-proof-block %4u lane %u  %8.1f MB  ready %7.1f  copy %7.1f .. %7.1f ms  (%5.1f GB/s)\n", r.chunk, r.lane, r.bytes / 1e6, a, b, d,
                    r.bytes / 1e6 / (d - b > 1e-3 ? d - b : 1e-3));
            hipEventDestroy(r.ready), hipEventDestroy(r.c0), hipEventDestroy(r.c1);
        }
        for (auto& r : c->trecs) {   // every timed scope of the call, in enqueue order
            float a = 0, b = 0;
            hipEventElapsedTime(&a, J.io_t0, r.e0), hipEventElapsedTime(&b, J.io_t0, r.e1);
            fprintf(stderr, "gpu: %-16s %7.1f .. %7.1f ms\n", r.name, a, b);
        }
        hipEventDestroy(J.io_t0);
    }
    if (zs || e_sync != hipSuccess) wipe_witness(c);   // a failed call leaves no nonce, blinder or RNG block behind
    if (zs) return zs;
    HIPCHK(c, e_sync);
    HIPCHK(c, hipGetLastError());
    timing_end(c);
    return ZK_OK;
}

extern "C" zk_status zk_prove_batch_device(zk_ctx* c, uint64_t B, const void* d_msg, const void* d_sig, const void* d_pk, const void* d_which,
                                           const zk_rng* rng, void* d_out, uint64_t out_cap, void* d_out_off, void* d_status) {
    if (!c || !rng || (B && (!d_msg || !d_sig || !d_pk || !d_which || !rng->data || !d_out)) || !d_out_off || !d_status) return ZK_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    return prove_device(c, B, (const uint8_t*)d_msg, (const uint8_t*)d_sig, (const uint8_t*)d_pk, (const uint32_t*)d_which, rng->mode, rng->data,
                        rng->stride_blocks, (uint8_t*)d_out, out_cap, (uint64_t*)d_out_off, (int32_t*)d_status);
}

extern "C" zk_status zk_prove_batch(zk_ctx* c, uint64_t B, const uint8_t* msg, const uint8_t* sig, const uint8_t* pk, const uint32_t* which,
                                    const zk_rng* rng, uint8_t* out, uint64_t out_cap, uint64_t* out_off, int32_t* status) {
    if (!c || !rng || !out_off || !status || (B && (!msg || !sig || !pk || !which || !rng->data || !out))) return ZK_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    if (!c->params_set || !c->N) return ZK_E_BUFFER;
    if (c->stream_busy) {
        c->err = "streamed jobs are in flight on this context (zk_prove_wait / zk_verify_wait them first)";
        return ZK_E_ARG;
    }
    size_t rng_bytes = rng->mode == ZK_RNG_SEED ? 32 * B : 32 * B * rng->stride_blocks;
    size_t bb = B ? B : 1;
    uint64_t cap_dev = std::min<uint64_t>(out_cap, zk_proof_max_size(c) * bb);
    // the small per-proof arrays live in the context's grow-only input buffer (no hipMalloc / hipFree per call)
    Carver k(nullptr);
    auto carve_in = [&](Carver& kk, uint8_t*& m, uint8_t*& sg, uint8_t*& p, uint32_t*& w, uint8_t*& r, uint64_t*& o, int32_t*& st_) {
        m = (uint8_t*)kk.take(32 * bb), sg = (uint8_t*)kk.take(64 * bb), p = (uint8_t*)kk.take(64 * bb), w = (uint32_t*)kk.take(4 * bb);
        r = (uint8_t*)kk.take(rng_bytes ? rng_bytes : 32), o = (uint64_t*)kk.take(8 * (bb + 1)), st_ = (int32_t*)kk.take(4 * bb);
    };
    uint8_t *d_msg, *d_sig, *d_pk, *d_rng;
    uint32_t* d_which;
    uint64_t* d_off;
    int32_t* d_st;
    carve_in(k, d_msg, d_sig, d_pk, d_which, d_rng, d_off, d_st);
    zk_status zs = ensure_in_buf(c, k.off + 256);
    if (zs) return zs;
    Carver k2((uint8_t*)c->in_buf);
    carve_in(k2, d_msg, d_sig, d_pk, d_which, d_rng, d_off, d_st);
    // `out` may itself lie in this GPU's HBM (hipMalloc'ed by the caller): the proofs are written there and never cross the link --
    // host inputs, device-resident output (zk_pool_prove_batch_device is this, per shard)
    bool out_on_device = false;
    if (B) {
        hipPointerAttribute_t a;
        if (hipPointerGetAttributes(&a, out) == hipSuccess) out_on_device = a.type == hipMemoryTypeDevice;
        else (void)hipGetLastError();
        if (out_on_device && a.device != c->device) {
            c->err = "`out` lies on another device than this context's";
            return ZK_E_ARG;
        }
    }
    if (out_on_device) cap_dev = out_cap;
    else if ((zs = ensure_io_buf(c, cap_dev ? cap_dev : 32))) return zs;  // proof bytes: the context's grow-only staging buffer
    uint8_t* d_out = out_on_device ? out : (uint8_t*)c->io_buf;
    // A call of a few proofs: the five input arrays cross in ONE copy out of the context's page-locked mirror of in_buf and the lanes wait for its event -- five
    // pageable copies and a host wait cost a one-proof call 0.1 ms before its first kernel (profiles/r06_ab_variants.txt (15)).
    const size_t in_end = (size_t)((uint8_t*)d_off - (uint8_t*)c->in_buf), res_end = (size_t)((uint8_t*)(d_st + bb) - (uint8_t*)c->in_buf);
    const bool staged = B && res_end <= ZK_STAGE_MAX;
    if (staged) {
        if ((zs = ensure_h_stage(c, res_end))) return zs;
        auto at = [&](const void* d) { return c->h_stage + ((const uint8_t*)d - (const uint8_t*)c->in_buf); };
        memcpy(at(d_msg), msg, 32 * B), memcpy(at(d_sig), sig, 64 * B), memcpy(at(d_pk), pk, 64 * B), memcpy(at(d_which), which, 4 * B), memcpy(at(d_rng), rng->data, rng_bytes);
        HIPCHK(c, hipMemcpyAsync(c->in_buf, c->h_stage, in_end, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipEventRecord(c->in_ready, c->stream));
    } else if (B) {
        HIPCHK(c, hipMemcpyAsync(d_msg, msg, 32 * B, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(d_sig, sig, 64 * B, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(d_pk, pk, 64 * B, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(d_which, which, 4 * B, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(d_rng, rng->data, rng_bytes, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));   // both lanes read these arrays
    }
    // a page-locked `out` (zk_host_alloc) receives each chunk by DMA while the next chunks are proved
    uint8_t* sink = B && !out_on_device && host_ptr_is_pinned(out) ? out : nullptr;
    if (sink) {
        zs = ensure_copy_stream(c);
        if (zs) return zs;
    }
    zs = prove_device(c, B, d_msg, d_sig, d_pk, d_which, rng->mode, d_rng, rng->stride_blocks, d_out, cap_dev, d_off, d_st, sink, staged ? c->in_ready : nullptr);
    if (zs) {
        if (staged) (void)hipStreamSynchronize(c->stream);   // the mirror is reused by the next call
        return zs;
    }
    if (staged) {   // offsets and statuses lie next to each other in in_buf: one copy
        HIPCHK(c, hipMemcpy(c->h_stage + in_end, d_off, res_end - in_end, hipMemcpyDeviceToHost));
        memcpy(out_off, c->h_stage + in_end, 8 * (B + 1));
        memcpy(status, c->h_stage + ((uint8_t*)d_st - (uint8_t*)c->in_buf), 4 * B);
    } else {
        HIPCHK(c, hipMemcpy(out_off, d_off, 8 * (B + 1), hipMemcpyDeviceToHost));
        if (B) HIPCHK(c, hipMemcpy(status, d_st, 4 * B, hipMemcpyDeviceToHost));
    }
    if (!sink && !out_on_device && out_off[B]) HIPCHK(c, hipMemcpy(out, d_out, out_off[B], hipMemcpyDeviceToHost));
    return ZK_OK;
}

// Diagnostic: what a page-locked copy of `bytes` reaches on lane `lane`'s copy stream, in both directions (GB/s), measured with HIP
// events on that stream.  The link carries ~57 GB/s on these boxes; a stream whose copies the runtime serves with shader blits instead of
// an SDMA engine -- seen in about half of the processes of some boxes, whatever the allocation of the host buffer: DESIGN.md section 9 --
// reaches ~27 GB/s, and every host-pointer call of the process then runs at 160 k instead of 262 k proofs/s.
extern "C" zk_status zk_ctx_copy_probe(zk_ctx* c, uint32_t lane, size_t bytes, int numa_node, float* d2h_gbps, float* h2d_gbps) {
    if (!c || lane >= ZK_MAX_LANES || bytes < (1u << 20)) return ZK_E_ARG;
    if (c->stream_busy) return busy_refusal(c);
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t s = c->pl[lane].copy_stream;
    void *d = nullptr, *h = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    zk_status zs = ZK_OK;
    auto fail = [&](hipError_t e, const char* what) {
        c->err = std::string(what) + ": " + hipGetErrorString(e);
        zs = ZK_E_DEVICE;
    };
    hipError_t e = hipMalloc(&d, bytes);
    if (e == hipSuccess) {
        // numa_node >= 0: the page-locked buffer is bound to that node (the policy is the calling thread's for the duration of the allocation)
        unsigned long mask[16] = {0};
        const bool bind = numa_node >= 0 && numa_node < 1024;
        if (bind) {
            mask[numa_node / (8 * sizeof(unsigned long))] |= 1ul << (numa_node % (8 * sizeof(unsigned long)));
            (void)syscall(SYS_set_mempolicy, 2 /* MPOL_BIND */, mask, 1024ul);
        }
        e = hipHostMalloc(&h, bytes, bind ? hipHostMallocNumaUser : hipHostMallocDefault);
        if (e == hipSuccess) memset(h, 1, bytes);
        if (bind) (void)syscall(SYS_set_mempolicy, 0 /* MPOL_DEFAULT */, nullptr, 0ul);
    }
    if (e == hipSuccess) e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    if (e != hipSuccess) fail(e, "copy probe set-up");
    for (int dir = 0; dir < 2 && !zs; dir++) {
        void *dst = dir ? d : h, *src = dir ? h : d;
        hipMemcpyKind kind = dir ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost;
        e = hipMemcpyAsync(dst, src, 1 << 20, kind, s);   // the stream's first copy of this kind sets its queue up
        if (e == hipSuccess) e = hipEventRecord(e0, s);
        if (e == hipSuccess) e = hipMemcpyAsync(dst, src, bytes, kind, s);
        if (e == hipSuccess) e = hipEventRecord(e1, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        float ms = 0;
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
        if (e != hipSuccess) {
            fail(e, "copy probe");
            break;
        }
        float* out = dir ? h2d_gbps : d2h_gbps;
        if (out) *out = ms > 0 ? (float)(bytes / 1e6 / ms) : 0.f;
    }
    if (e0) hipEventDestroy(e0);
    if (e1) hipEventDestroy(e1);
    if (h) hipHostFree(h);
    if (d) hipFree(d);
    return zs;
}

extern "C" uint32_t zk_last_timing(const zk_ctx* c, float* total_ms, const char** names, float* ms, uint32_t cap) {
    if (!c) return 0;
    if (total_ms) *total_ms = c->last_total_ms;
    uint32_t n = (uint32_t)c->last_timing.size();
    for (uint32_t i = 0; i < n && i < cap; i++) {
        if (names) names[i] = c->last_timing[i].first;
        if (ms) ms[i] = c->last_timing[i].second;
    }
    return n;
}

extern "C" float zk_last_wall_ms(const zk_ctx* c) { return c ? c->last_wall_ms : 0.f; }
extern "C" zk_status zk_ctx_set_timing(zk_ctx* c, int mode) {
    if (!c || mode < ZK_TIMING_OFF || mode > ZK_TIMING_AUTO) return ZK_E_ARG;
    c->timing_mode = mode;
    return ZK_OK;
}

// ------------------------------------------------------------------ synthetic workload
extern "C" zk_status zk_synth_workload(zk_ctx* c, uint64_t seed, uint64_t nkeys, uint64_t B, uint8_t* ring, uint8_t* msg, uint8_t* sig, uint8_t* pk,
                                       uint32_t* which, uint8_t* seeds) {
    if (!c || !ring || nkeys < 1 || (B && (!msg || !sig || !pk || !which || !seeds))) return ZK_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    uint8_t *d_ring, *d_msg, *d_sig, *d_pk, *d_seeds;
    uint32_t* d_which;
    size_t bb = B ? B : 1;
    HIPCHK(c, hipMalloc(&d_ring, 32 * nkeys));
    HIPCHK(c, hipMalloc(&d_msg, 32 * bb));
    HIPCHK(c, hipMalloc(&d_sig, 64 * bb));
    HIPCHK(c, hipMalloc(&d_pk, 64 * bb));
    HIPCHK(c, hipMalloc(&d_which, 4 * bb));
    HIPCHK(c, hipMalloc(&d_seeds, 32 * bb));
    launch_synth(c->stream, c->P.pfix_G, seed, nkeys, B, d_ring, d_msg, d_sig, d_pk, d_which, d_seeds);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(ring, d_ring, 32 * nkeys, hipMemcpyDeviceToHost));
    if (B) {
        HIPCHK(c, hipMemcpy(msg, d_msg, 32 * B, hipMemcpyDeviceToHost));
        HIPCHK(c, hipMemcpy(sig, d_sig, 64 * B, hipMemcpyDeviceToHost));
        HIPCHK(c, hipMemcpy(pk, d_pk, 64 * B, hipMemcpyDeviceToHost));
        HIPCHK(c, hipMemcpy(which, d_which, 4 * B, hipMemcpyDeviceToHost));
        HIPCHK(c, hipMemcpy(seeds, d_seeds, 32 * B, hipMemcpyDeviceToHost));
    }
    hipFree(d_ring), hipFree(d_msg), hipFree(d_sig), hipFree(d_pk), hipFree(d_which), hipFree(d_seeds);
    return ZK_OK;
}

static zk_status tom_commit_generic(zk_ctx* c, const uint32_t* tab_g, const uint32_t* tab_h, uint32_t bits, uint64_t count, const uint8_t* d_v, const uint8_t* d_r, uint8_t* d_out) {
    // temporary list
    TomList L;
    void* mem;
    size_t per = 36 * 7;
    HIPCHK(c, hipMalloc(&mem, per * count + 4096));
    Carver k((uint8_t*)mem);
    L = k.list(count);
    launch_bytes_to_scalars(c->stream, d_v, count, L.v);
    launch_bytes_to_scalars(c->stream, d_r, count, L.r);
    DevParams P = c->P;
    P.tom_tab_g = (uint32_t*)tab_g, P.tom_tab_h = (uint32_t*)tab_h, P.tom_bits = bits;
    launch_tom_commit(c->stream, P, L, (uint32_t)count, 1, 1);
    launch_tom_normalize(c->stream, L, (uint32_t)count, 0, 1, 1);
    launch_affine_to_bytes(c->stream, L.ax, L.ay, count, 1, d_out);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipFree(mem));
    return ZK_OK;
}
extern "C" zk_status zk_synth_params(zk_ctx* c, uint64_t seed, uint8_t nist_h[64], uint8_t tom_g[72], uint8_t tom_h[72]) {
    if (!c || !nist_h || !tom_g || !tom_h) return ZK_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    uint8_t* d;
    HIPCHK(c, hipMalloc(&d, 512));
    HIPCHK(c, hipMemsetAsync(d, 0, 512, c->stream));
    launch_synth_param_scalars(c->stream, seed, d, d + 32);  // kn, kt (big-endian); d+64: zero scalar
    launch_test_pfix(c->stream, c->P.pfix_G, 1, d, d + 128);
    zk_status zs = tom_commit_generic(c, c->tom_tab_gen, c->tom_tab_gen, 8, 1, d + 32, d + 64, d + 256);
    if (zs) return zs;
    HIPCHK(c, hipMemcpy(nist_h, d + 128, 64, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(tom_h, d + 256, 72, hipMemcpyDeviceToHost));
    HIPCHK(c, hipFree(d));
    uint32_t gw[18];
    memcpy(gw, TOM_GX_W, 36), memcpy(gw + 9, TOM_GY_W, 36);
    for (int k = 0; k < 2; k++)
        for (int i = 0; i < 36; i++) tom_g[36 * k + i] = (uint8_t)(gw[9 * k + (35 - i) / 4] >> (8 * ((35 - i) % 4)));
    return ZK_OK;
}

// ------------------------------------------------------------------ unit-test hooks
extern "C" zk_status zk_test_field_op(zk_ctx* c, int which, int op, uint64_t count, const uint8_t* a, const uint8_t* b, uint8_t* out) {
    if (!c || !a || !b || !out || which < 0 || which > 2 || op < 0 || op > 5) return ZK_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    DevBuf da, db, dout;
    HIPCHK(c, hipMalloc(&da.p, 40 * count)); HIPCHK(c, hipMalloc(&db.p, 40 * count)); HIPCHK(c, hipMalloc(&dout.p, 40 * count));
    HIPCHK(c, hipMemcpy(da.p, a, 40 * count, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(db.p, b, 40 * count, hipMemcpyHostToDevice));
    launch_test_field(c->stream, which, op, count, (uint8_t*)da.p, (uint8_t*)db.p, (uint8_t*)dout.p);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(out, dout.p, 40 * count, hipMemcpyDeviceToHost));
    return ZK_OK;
}
extern "C" zk_status zk_test_tom_commit(zk_ctx* c, uint64_t count, const uint8_t* v, const uint8_t* r, uint8_t* out) {
    if (!c || !v || !r || !out || !count) return ZK_E_ARG;
    if (!c->params_set) return ZK_E_BUFFER;
    HIPCHK(c, hipSetDevice(c->device));
    DevBuf dv, dr, dout;
    HIPCHK(c, hipMalloc(&dv.p, 32 * count)); HIPCHK(c, hipMalloc(&dr.p, 32 * count)); HIPCHK(c, hipMalloc(&dout.p, 72 * count));
    HIPCHK(c, hipMemcpy(dv.p, v, 32 * count, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(dr.p, r, 32 * count, hipMemcpyHostToDevice));
    zk_status zs = tom_commit_generic(c, c->P.tom_tab_g, c->P.tom_tab_h, c->P.tom_bits, count, (uint8_t*)dv.p, (uint8_t*)dr.p, (uint8_t*)dout.p);
    if (zs) return zs;
    HIPCHK(c, hipMemcpy(out, dout.p, 72 * count, hipMemcpyDeviceToHost));
    return ZK_OK;
}
extern "C" zk_status zk_test_p256_fixed_mul(zk_ctx* c, int base_sel, uint64_t count, const uint8_t* k, uint8_t* out) {
    if (!c || !k || !out || !count || base_sel < 0 || base_sel > 1) return ZK_E_ARG;
    if (base_sel == 1 && !c->params_set) return ZK_E_BUFFER;
    HIPCHK(c, hipSetDevice(c->device));
    DevBuf dk, dout;
    HIPCHK(c, hipMalloc(&dk.p, 32 * count)); HIPCHK(c, hipMalloc(&dout.p, 64 * count));
    HIPCHK(c, hipMemcpy(dk.p, k, 32 * count, hipMemcpyHostToDevice));
    launch_test_pfix(c->stream, base_sel ? c->P.pfix_H : c->P.pfix_G, count, (uint8_t*)dk.p, (uint8_t*)dout.p);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(out, dout.p, 64 * count, hipMemcpyDeviceToHost));
    return ZK_OK;
}
extern "C" zk_status zk_test_sha256(zk_ctx* c, uint64_t count, uint64_t len, const uint8_t* msgs, uint8_t* digests) {
    if (!c || !digests || !count || (len && !msgs)) return ZK_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    DevBuf dm, dd;
    HIPCHK(c, hipMalloc(&dm.p, count * len + 4)); HIPCHK(c, hipMalloc(&dd.p, 32 * count));
    if (len) HIPCHK(c, hipMemcpy(dm.p, msgs, count * len, hipMemcpyHostToDevice));
    launch_test_sha256(c->stream, count, len, (uint8_t*)dm.p, (uint8_t*)dd.p);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(digests, dd.p, 32 * count, hipMemcpyDeviceToHost));
    return ZK_OK;
}
extern "C" zk_status zk_test_rng_draws(zk_ctx* c, uint64_t B, const zk_rng* rng, uint32_t first_k, uint32_t n_k, uint8_t* out) {
    if (!c || !rng || !rng->data || !out || !B || !n_k) return ZK_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    uint32_t sec = c->params_set ? c->P.sec : 80;
    size_t rng_bytes = rng->mode == ZK_RNG_SEED ? 32 * B : 32 * B * rng->stride_blocks;
    DevBuf dr, dout, dexc;
    HIPCHK(c, hipMalloc(&dr.p, rng_bytes)); HIPCHK(c, hipMalloc(&dout.p, 32 * B * n_k)); HIPCHK(c, hipMalloc(&dexc.p, 4 * B * (2 * RNG_MAX_EXC + 1)));
    HIPCHK(c, hipMemcpy(dr.p, rng->data, rng_bytes, hipMemcpyHostToDevice));
    Workspace W{};
    W.sec = sec;
    W.rng.seeds = (uint8_t*)dr.p, W.rng.stream = (uint8_t*)dr.p, W.rng.stride_blocks = rng->stride_blocks, W.rng.mode = rng->mode, W.rng.sec = (int)sec;
    W.rng.exc_idx = (uint32_t*)dexc.p, W.rng.exc_flags = W.rng.exc_idx + RNG_MAX_EXC * B, W.rng.exc_cnt = W.rng.exc_flags + RNG_MAX_EXC * B;
    W.rng.proof_base = 0;
    launch_rng_prepass(c->stream, W, (uint32_t)B, 0, first_k + n_k + RNG_MAX_EXC, 0, nullptr, false);
    launch_test_rng(c->stream, W.rng, B, first_k, n_k, (uint8_t*)dout.p);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(out, dout.p, 32 * B * n_k, hipMemcpyDeviceToHost));
    return ZK_OK;
}
