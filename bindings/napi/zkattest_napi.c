/* N-API addon over the C ABI (include/zkattest.h): the binding a maintainer of the reference would put behind
 * src/zkpAttestList.ts.  Plain C, N-API version 3 (the header of the Node in this image):
 *
 *   gcc -shared -fPIC -O2 -I/usr/include/node -I../../include zkattest_napi.c -o zkattest.node \
 *       -L../../zkp-ecdsa_amd/lib -lzkattest_hip -Wl,-rpath,<abs path of zkp-ecdsa_amd/lib>
 *
 * One handle = one zk_pool = the listed GPUs of this node (one GPU: a pool of one).  A handle runs one batch at a time: the
 * synchronous calls finish before they return, the *Async calls run on a libuv worker thread and mark the handle busy
 * until their Promise settles; a second call on a busy handle, any call on a destroyed one, and destroying a busy one
 * throw instead of touching freed memory.  The garbage collector destroys a handle nobody closed.  proveSubmit / verifySubmit
 * are the streamed form: several batches of one handle in flight (zk_pool_prove_submit / _wait), one Promise per batch.
 * Large proof buffers are page-locked (zk_host_alloc) and handed to JavaScript as external Buffers, so the engine moves
 * the proof bytes by DMA under its kernels and nothing is copied on the way out. */
#define NAPI_VERSION 3
#include <math.h>
#include <node_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "zkattest.h"

#define NAPI_OK(call)                                                 \
    do {                                                              \
        if ((call) != napi_ok) {                                      \
            napi_throw_error(env, NULL, "N-API call failed: " #call); \
            return NULL;                                              \
        }                                                             \
    } while (0)
#define PINNED_MIN ((size_t)32 << 20) /* proof buffers from this size on are page-locked */

struct Job;
typedef struct {
    zk_pool *pool;
    int busy, closed;
    int orphaned; /* the external was finalized (environment teardown) while a batch was still running: job_complete frees it */
    uint32_t sec;
    /* streamed batches (proveSubmit / verifySubmit): jobs in submission order, at most `inflight` of them submitted to the engine */
    struct Job *sq_head, *sq_tail;
    int sq_submitted, sq_running;
    uint32_t inflight;
} Handle;

static napi_value throw_text(napi_env env, zk_status st, const char *detail) {
    char msg[640];
    snprintf(msg, sizeof msg, "%s%s%s", zk_strerror(st), detail && detail[0] ? ": " : "", detail ? detail : "");
    napi_throw_error(env, NULL, msg); /* the reference's error texts: 'point not in group', 'T[i] is at infinity', ... */
    return NULL;
}
/* a NULL pool asks the library why the last zk_pool_create of this thread failed */
static napi_value throw_status(napi_env env, Handle *h, zk_status st) { return throw_text(env, st, zk_pool_last_error(h ? h->pool : NULL)); }
static int get_args(napi_env env, napi_callback_info info, size_t want, napi_value *argv) {
    size_t argc = want;
    if (napi_get_cb_info(env, info, &argc, argv, NULL, NULL) != napi_ok || argc < want) {
        napi_throw_type_error(env, NULL, "wrong number of arguments");
        return 0;
    }
    return 1;
}
/* the handle behind an external; refuses destroyed and (unless allow_busy) busy ones */
static Handle *get_handle(napi_env env, napi_value v, int allow_busy) {
    void *p = NULL;
    if (napi_get_value_external(env, v, &p) != napi_ok || !p) {
        napi_throw_type_error(env, NULL, "expected an engine handle");
        return NULL;
    }
    Handle *h = p;
    if (h->closed || !h->pool) {
        napi_throw_error(env, NULL, "the engine has been destroyed");
        return NULL;
    }
    if (h->busy && !allow_busy) {
        napi_throw_error(env, NULL, "the engine is busy with an asynchronous batch (one batch at a time per engine)");
        return NULL;
    }
    return h;
}
static void handle_release(Handle *h) {
    if (h->pool) zk_pool_destroy(h->pool);
    free(h);
}
/* The worker thread of a running batch still uses h->pool and job_complete still writes h->busy: a busy handle outlives its
 * external and is released by job_complete. */
static void handle_finalize(napi_env env, void *data, void *hint) {
    Handle *h = data;
    if (h->busy) h->orphaned = 1;
    else handle_release(h);
}
/* Buffer or typed array -> pointer + byte length (NULL for null/undefined) */
static int get_bytes(napi_env env, napi_value v, uint8_t **p, size_t *len) {
    napi_valuetype t;
    *p = NULL, *len = 0;
    if (napi_typeof(env, v, &t) != napi_ok) return 0;
    if (t == napi_null || t == napi_undefined) return 1;
    bool is = false;
    if (napi_is_buffer(env, v, &is) == napi_ok && is) return napi_get_buffer_info(env, v, (void **)p, len) == napi_ok;
    if (napi_is_typedarray(env, v, &is) == napi_ok && is) {
        napi_typedarray_type ty;
        size_t n, off;
        napi_value ab;
        void *data;
        if (napi_get_typedarray_info(env, v, &ty, &n, &data, &ab, &off) != napi_ok) return 0;
        size_t w = ty == napi_uint8_array || ty == napi_int8_array || ty == napi_uint8_clamped_array ? 1
                   : ty == napi_uint16_array || ty == napi_int16_array                              ? 2
                   : ty == napi_uint32_array || ty == napi_int32_array || ty == napi_float32_array ? 4
                                                                                                    : 8;
        *p = (uint8_t *)data, *len = n * w;
        return 1;
    }
    napi_throw_type_error(env, NULL, "expected a Buffer or typed array");
    return 0;
}
static napi_value new_buffer(napi_env env, const void *src, size_t len) {
    napi_value b;
    void *dst;
    if (napi_create_buffer_copy(env, len, len ? src : "", &dst, &b) != napi_ok) return NULL;
    return b;
}
static void set_prop(napi_env env, napi_value obj, const char *name, napi_value v) {
    if (v) napi_set_named_property(env, obj, name, v);
}
static void *xmalloc(napi_env env, size_t n) {
    void *p = malloc(n ? n : 1);
    if (!p && env) napi_throw_error(env, NULL, "out of memory");
    return p;
}
/* proof buffers: page-locked from PINNED_MIN on */
typedef struct {
    uint8_t *p;
    size_t cap;
    int pinned;
} Slab;
static int slab_alloc(Slab *s, size_t cap) {
    s->cap = cap ? cap : 1, s->pinned = 0, s->p = NULL;
    if (s->cap >= PINNED_MIN) {
        s->p = zk_host_alloc(s->cap);
        s->pinned = s->p != NULL;
    }
    if (!s->p) s->p = malloc(s->cap);
    return s->p != NULL;
}
static void slab_free(Slab *s) {
    if (s->p) {
        if (s->pinned) zk_host_free(s->p);
        else free(s->p);
    }
    s->p = NULL;
}
static void slab_finalize(napi_env env, void *data, void *hint) {
    if (hint) zk_host_free(data);
    else free(data);
}
/* hands the slab to JavaScript as a Buffer of `len` bytes (ownership moves to the Buffer) */
static napi_value slab_to_buffer(napi_env env, Slab *s, size_t len) {
    napi_value b = NULL;
    if (napi_create_external_buffer(env, len, s->p, slab_finalize, s->pinned ? (void *)1 : NULL, &b) != napi_ok) {
        slab_free(s);
        return NULL;
    }
    s->p = NULL;
    return b;
}

static napi_value CreatePool(napi_env env, napi_callback_info info) { /* (deviceIds: Int32Array | Buffer of i32) -> handle */
    napi_value argv[1];
    if (!get_args(env, info, 1, argv)) return NULL;
    uint8_t *ids;
    size_t li;
    if (!get_bytes(env, argv[0], &ids, &li)) return NULL;
    if (!ids || li < 4 || li % 4) {
        napi_throw_type_error(env, NULL, "createPool: an Int32Array of device ids");
        return NULL;
    }
    Handle *h = calloc(1, sizeof *h);
    if (!h) return throw_text(env, ZK_E_BUFFER, "out of memory");
    h->sec = 80;
    zk_status st = zk_pool_create((const int *)ids, (int)(li / 4), &h->pool);
    if (st != ZK_OK) {
        napi_value r = throw_status(env, h, st);
        if (h->pool) zk_pool_destroy(h->pool);
        free(h);
        return r;
    }
    napi_value ext;
    if (napi_create_external(env, h, handle_finalize, NULL, &ext) != napi_ok) {
        zk_pool_destroy(h->pool), free(h);
        napi_throw_error(env, NULL, "could not create the handle");
        return NULL;
    }
    return ext;
}
static napi_value DestroyPool(napi_env env, napi_callback_info info) { /* idempotent; throws while an async batch is running */
    napi_value argv[1];
    if (!get_args(env, info, 1, argv)) return NULL;
    void *p = NULL;
    if (napi_get_value_external(env, argv[0], &p) != napi_ok || !p) return NULL;
    Handle *h = p;
    if (h->closed) return NULL;
    if (h->busy) {
        napi_throw_error(env, NULL, "cannot destroy an engine while an asynchronous batch is running on it");
        return NULL;
    }
    h->closed = 1;
    zk_pool_destroy(h->pool);
    h->pool = NULL;
    return NULL;
}
static napi_value PoolInfo(napi_env env, napi_callback_info info) { /* (h) -> {devices, ringTransport, proofMaxSize} */
    napi_value argv[1], o, v;
    if (!get_args(env, info, 1, argv)) return NULL;
    Handle *h = get_handle(env, argv[0], 1);
    if (!h) return NULL;
    NAPI_OK(napi_create_object(env, &o));
    NAPI_OK(napi_create_int32(env, zk_pool_size(h->pool), &v));
    set_prop(env, o, "devices", v);
    NAPI_OK(napi_create_string_utf8(env, zk_pool_ring_transport(h->pool), NAPI_AUTO_LENGTH, &v));
    set_prop(env, o, "ringTransport", v);
    NAPI_OK(napi_create_double(env, (double)zk_proof_max_size(zk_pool_ctx(h->pool, 0)), &v));
    set_prop(env, o, "proofMaxSize", v);
    return o;
}
/* (h, name, value): per-device settings applied to every device of the pool */
static napi_value SetOption(napi_env env, napi_callback_info info) {
    napi_value argv[3];
    if (!get_args(env, info, 3, argv)) return NULL;
    Handle *h = get_handle(env, argv[0], 0);
    if (!h) return NULL;
    char name[32];
    size_t ln;
    uint32_t val;
    NAPI_OK(napi_get_value_string_utf8(env, argv[1], name, sizeof name, &ln));
    NAPI_OK(napi_get_value_uint32(env, argv[2], &val));
    if (!strcmp(name, "inflight")) { /* streamed jobs inside the engine at a time (proveSubmit / verifySubmit); the addon's own knob */
        if (val < 1 || val > 8) return throw_text(env, ZK_E_ARG, name);
        h->inflight = val;
        return NULL;
    }
    for (int i = 0; i < zk_pool_size(h->pool); i++) {
        zk_ctx *c = zk_pool_ctx(h->pool, i);
        zk_status st = !strcmp(name, "chunk")         ? zk_ctx_set_chunk(c, val)
                       : !strcmp(name, "lanes")       ? zk_ctx_set_lanes(c, val)
                       : !strcmp(name, "combBits")    ? zk_ctx_set_comb_bits(c, val)
                       : !strcmp(name, "hostTaper")   ? zk_ctx_set_host_taper(c, val)
                       : !strcmp(name, "batchVerify") ? zk_ctx_set_batch_verify(c, val)
                       : !strcmp(name, "mode")        ? zk_ctx_set_mode(c, val)
                       : !strcmp(name, "slice")       ? zk_ctx_set_slice(c, val)
                       : !strcmp(name, "ringFold")    ? zk_ctx_set_ring_fold(c, val)
                       : !strcmp(name, "verifyGroups") ? zk_ctx_set_verify_groups(c, val)
                       : !strcmp(name, "wire")        ? zk_ctx_set_wire(c, val)
                       : !strcmp(name, "wipe")        ? zk_ctx_wipe(c) /* value ignored: zero the witness-derived device memory of every context now */
                                                      : ZK_E_ARG;
        if (st != ZK_OK) return throw_text(env, st, name);
    }
    return NULL;
}
static napi_value SetParams(napi_env env, napi_callback_info info) { /* (h, nistH 64, tomG 72, tomH 72, secLevel) */
    napi_value argv[5];
    if (!get_args(env, info, 5, argv)) return NULL;
    Handle *h = get_handle(env, argv[0], 0);
    uint8_t *a, *b, *c;
    size_t la, lb, lc;
    uint32_t sec;
    if (!h || !get_bytes(env, argv[1], &a, &la) || !get_bytes(env, argv[2], &b, &lb) || !get_bytes(env, argv[3], &c, &lc)) return NULL;
    NAPI_OK(napi_get_value_uint32(env, argv[4], &sec));
    if (la != 64 || lb != 72 || lc != 72) {
        napi_throw_range_error(env, NULL, "params: h_NIST is 64 bytes, g and h of Tom-256 are 72 bytes (affine, big-endian)");
        return NULL;
    }
    zk_status st = zk_pool_set_params(h->pool, a, b, c, sec);
    if (st == ZK_OK) h->sec = sec;
    return st == ZK_OK ? NULL : throw_status(env, h, st);
}
static napi_value SetRing(napi_env env, napi_callback_info info) { /* (h, keys: n x 32 bytes) -> transport */
    napi_value argv[2];
    if (!get_args(env, info, 2, argv)) return NULL;
    Handle *h = get_handle(env, argv[0], 0);
    uint8_t *k;
    size_t lk;
    if (!h || !get_bytes(env, argv[1], &k, &lk)) return NULL;
    zk_status st = zk_pool_set_ring(h->pool, k, lk / 32);
    if (st != ZK_OK) return throw_status(env, h, st);
    napi_value v;
    NAPI_OK(napi_create_string_utf8(env, zk_pool_ring_transport(h->pool), NAPI_AUTO_LENGTH, &v));
    return v;
}
static napi_value SynthParams(napi_env env, napi_callback_info info) { /* (h, seed) -> {nistH, tomG, tomH} */
    napi_value argv[2];
    if (!get_args(env, info, 2, argv)) return NULL;
    Handle *h = get_handle(env, argv[0], 0);
    uint32_t seed;
    if (!h) return NULL;
    NAPI_OK(napi_get_value_uint32(env, argv[1], &seed));
    uint8_t a[64], b[72], c[72];
    zk_ctx *ctx = zk_pool_ctx(h->pool, 0);
    zk_status st = zk_synth_params(ctx, seed, a, b, c);
    if (st != ZK_OK) return throw_text(env, st, zk_last_error(ctx));
    napi_value o;
    NAPI_OK(napi_create_object(env, &o));
    set_prop(env, o, "nistH", new_buffer(env, a, 64)), set_prop(env, o, "tomG", new_buffer(env, b, 72)), set_prop(env, o, "tomH", new_buffer(env, c, 72));
    return o;
}
static napi_value SynthWorkload(napi_env env, napi_callback_info info) { /* (h, seed, nKeys, B) -> {ring, msg, sig, pk, which, seeds} */
    napi_value argv[4];
    if (!get_args(env, info, 4, argv)) return NULL;
    Handle *h = get_handle(env, argv[0], 0);
    uint32_t seed, nk, B;
    if (!h) return NULL;
    NAPI_OK(napi_get_value_uint32(env, argv[1], &seed));
    NAPI_OK(napi_get_value_uint32(env, argv[2], &nk));
    NAPI_OK(napi_get_value_uint32(env, argv[3], &B));
    uint8_t *ring = malloc(32 * (size_t)nk + 1), *msg = malloc(32 * (size_t)B + 1), *sig = malloc(64 * (size_t)B + 1), *pk = malloc(64 * (size_t)B + 1),
            *seeds = malloc(32 * (size_t)B + 1);
    uint32_t *which = malloc(4 * (size_t)B + 4);
    napi_value o = NULL;
    zk_ctx *ctx = zk_pool_ctx(h->pool, 0);
    zk_status st = ring && msg && sig && pk && seeds && which ? zk_synth_workload(ctx, seed, nk, B, ring, msg, sig, pk, which, seeds) : ZK_E_BUFFER;
    if (st == ZK_OK && napi_create_object(env, &o) == napi_ok) {
        set_prop(env, o, "ring", new_buffer(env, ring, 32 * (size_t)nk)), set_prop(env, o, "msg", new_buffer(env, msg, 32 * (size_t)B));
        set_prop(env, o, "sig", new_buffer(env, sig, 64 * (size_t)B)), set_prop(env, o, "pk", new_buffer(env, pk, 64 * (size_t)B));
        set_prop(env, o, "which", new_buffer(env, which, 4 * (size_t)B)), set_prop(env, o, "seeds", new_buffer(env, seeds, 32 * (size_t)B));
    }
    free(ring), free(msg), free(sig), free(pk), free(seeds), free(which);
    return st == ZK_OK ? o : throw_text(env, st, zk_last_error(ctx));
}

/* ---- batches.  A Job owns copies of the small inputs; the (large) proof input of a verification is referenced, not copied. */
typedef struct Job {
    int verify, async;
    Handle *h;
    size_t B;
    uint8_t *msg, *sig, *pk, *seeds;
    const uint8_t *proofs_in;
    napi_ref proofs_ref, h_ref; /* an asynchronous job keeps the proof bytes and the handle alive */
    uint32_t *which;
    Slab out;
    uint64_t worst_cap, *off, *len;
    int32_t *status;
    uint8_t *ok;
    zk_status rc;
    char err[384];
    napi_deferred deferred;
    napi_async_work work;
    /* streamed form */
    struct Job *next;
    int stream, submitted, op; /* op of the running step: 0 = submit, 1 = wait */
    zk_pool_job *pj;
    napi_ref out_ref; /* the caller's page-locked output Buffer */
    uint8_t *out_ext;
    size_t out_ext_cap;
} Job;
static uint8_t *dup_bytes(const uint8_t *p, size_t n) {
    uint8_t *q = malloc(n ? n : 1);
    if (q && p && n) memcpy(q, p, n);
    return q;
}
static void job_free(napi_env env, Job *j) {
    if (j->proofs_ref && env) napi_delete_reference(env, j->proofs_ref);
    if (j->h_ref && env) napi_delete_reference(env, j->h_ref);
    if (j->out_ref && env) napi_delete_reference(env, j->out_ref);
    slab_free(&j->out);
    free(j->msg), free(j->sig), free(j->pk), free(j->seeds), free(j->which), free(j->off), free(j->len), free(j->status), free(j->ok);
    free(j);
}
/* output capacity for B proofs over G devices: mean + 8 sigma of the zero-bit repetitions per shard, at most the worst case */
static void prove_caps(Handle *h, size_t B, uint64_t *cap, uint64_t *worst) {
    uint64_t G = (uint64_t)zk_pool_size(h->pool), m = (B + G - 1) / G;
    uint64_t mx = zk_proof_max_size(zk_pool_ctx(h->pool, 0));
    uint64_t w = ((mx * (m ? m : 1) + 511) & ~(uint64_t)255) * G;
    double mean = (double)m * ((double)mx - 3392.0 * h->sec / 2.0), dev = 8.0 * 3392.0 * sqrt((double)m * h->sec / 4.0);
    uint64_t c = (((uint64_t)(mean + dev) + mx + 511) & ~(uint64_t)255) * G;
    *worst = w, *cap = c < w ? c : w;
}
static void job_execute(napi_env env, void *data) { /* worker thread (or inline for the synchronous calls): no N-API calls here */
    Job *j = data;
    zk_pool *p = j->h->pool;
    if (j->verify) {
        j->rc = zk_pool_verify_batch(p, j->B, j->msg, j->proofs_in, j->off, j->len, j->seeds, j->ok, j->status);
    } else {
        zk_rng rng = {ZK_RNG_SEED, j->seeds, 0};
        j->rc = zk_pool_prove_batch(p, j->B, j->msg, j->sig, j->pk, j->which, &rng, j->out.p, j->out.cap, j->off, j->len, j->status);
        if (j->rc == ZK_E_BUFFER && j->out.cap < j->worst_cap) { /* 8 sigma were not enough: worst-case buffer */
            slab_free(&j->out);
            if (slab_alloc(&j->out, j->worst_cap))
                j->rc = zk_pool_prove_batch(p, j->B, j->msg, j->sig, j->pk, j->which, &rng, j->out.p, j->out.cap, j->off, j->len, j->status);
        }
    }
    if (j->rc != ZK_OK) snprintf(j->err, sizeof j->err, "%s: %s", zk_strerror(j->rc), zk_pool_last_error(p));
}
static napi_value job_result(napi_env env, Job *j) {
    napi_value v;
    if (napi_create_object(env, &v) != napi_ok) return NULL;
    if (j->verify) {
        set_prop(env, v, "ok", new_buffer(env, j->ok, j->B));
    } else {
        uint64_t end = 0;
        for (size_t b = 0; b < j->B; b++)
            if (j->off[b] + j->len[b] > end) end = j->off[b] + j->len[b];
        set_prop(env, v, "proofs", slab_to_buffer(env, &j->out, (size_t)end));
        set_prop(env, v, "offsets", new_buffer(env, j->off, 8 * j->B)), set_prop(env, v, "lengths", new_buffer(env, j->len, 8 * j->B));
    }
    set_prop(env, v, "status", new_buffer(env, j->status, 4 * j->B));
    return v;
}
static void job_complete(napi_env env, napi_status status, void *data) { /* main thread */
    Job *j = data;
    j->h->busy = 0;
    if (j->h->orphaned) { /* finalized meanwhile: nobody can reach the handle any more */
        handle_release(j->h);
        j->h = NULL;
    }
    napi_value v = status == napi_ok && j->rc == ZK_OK ? job_result(env, j) : NULL;
    if (v) {
        napi_resolve_deferred(env, j->deferred, v);
    } else {
        napi_value msg, e;
        napi_create_string_utf8(env, j->rc != ZK_OK ? j->err : "asynchronous batch failed", NAPI_AUTO_LENGTH, &msg);
        napi_create_error(env, NULL, msg, &e);
        napi_reject_deferred(env, j->deferred, e);
    }
    napi_delete_async_work(env, j->work);
    job_free(env, j);
}
static napi_value job_run(napi_env env, Job *j, const char *name) {
    if (!j->async) {
        job_execute(env, j);
        napi_value v = j->rc == ZK_OK ? job_result(env, j) : NULL;
        if (!v) {
            if (j->rc != ZK_OK) napi_throw_error(env, NULL, j->err);
            else napi_throw_error(env, NULL, "could not build the result");
        }
        job_free(env, j);
        return v;
    }
    napi_value promise, rn;
    if (napi_create_promise(env, &j->deferred, &promise) != napi_ok || napi_create_string_utf8(env, name, NAPI_AUTO_LENGTH, &rn) != napi_ok ||
        napi_create_async_work(env, NULL, rn, job_execute, job_complete, j, &j->work) != napi_ok) {
        job_free(env, j);
        napi_throw_error(env, NULL, "could not queue the batch");
        return NULL;
    }
    j->h->busy = 1;
    if (napi_queue_async_work(env, j->work) != napi_ok) {
        j->h->busy = 0;
        napi_delete_async_work(env, j->work);
        job_free(env, j);
        napi_throw_error(env, NULL, "could not queue the batch");
        return NULL;
    }
    return promise;
}
/* (h, msg Bx32, sig Bx64, pk Bx64, which Bx4 (u32 LE), seeds Bx32)
 *   -> {proofs: Buffer, offsets: B u64 LE, lengths: B u64 LE, status: B i32}   proof b = proofs[offsets[b] .. offsets[b] + lengths[b]) */
static napi_value prove_common(napi_env env, napi_callback_info info, int async) {
    napi_value argv[6];
    if (!get_args(env, info, 6, argv)) return NULL;
    Handle *h = get_handle(env, argv[0], 0);
    uint8_t *msg, *sig, *pk, *which, *seeds;
    size_t lm, ls, lp, lw, lse;
    if (!h || !get_bytes(env, argv[1], &msg, &lm) || !get_bytes(env, argv[2], &sig, &ls) || !get_bytes(env, argv[3], &pk, &lp) ||
        !get_bytes(env, argv[4], &which, &lw) || !get_bytes(env, argv[5], &seeds, &lse))
        return NULL;
    size_t B = lm / 32;
    if (!B || lm != 32 * B || ls != 64 * B || lp != 64 * B || lw != 4 * B || lse != 32 * B) {
        napi_throw_range_error(env, NULL, "proveBatch: per proof 32-byte msgHash, 64-byte signature, 64-byte public key, u32 index, 32-byte seed");
        return NULL;
    }
    Job *j = calloc(1, sizeof *j);
    if (!j) return throw_text(env, ZK_E_BUFFER, "out of memory");
    j->h = h, j->B = B, j->async = async;
    j->msg = dup_bytes(msg, lm), j->sig = dup_bytes(sig, ls), j->pk = dup_bytes(pk, lp), j->seeds = dup_bytes(seeds, lse);
    j->which = (uint32_t *)dup_bytes(which, lw);
    uint64_t cap;
    prove_caps(h, B, &cap, &j->worst_cap);
    j->off = malloc(8 * B), j->len = malloc(8 * B), j->status = malloc(4 * B);
    if (!j->msg || !j->sig || !j->pk || !j->seeds || !j->which || !j->off || !j->len || !j->status || !slab_alloc(&j->out, cap)) {
        job_free(env, j);
        return throw_text(env, ZK_E_BUFFER, "out of memory");
    }
    if (async && napi_create_reference(env, argv[0], 1, &j->h_ref) != napi_ok) {
        job_free(env, j);
        return throw_text(env, ZK_E_BUFFER, "could not reference the handle");
    }
    return job_run(env, j, "zkattest.proveBatch");
}
static napi_value ProveBatch(napi_env env, napi_callback_info info) { return prove_common(env, info, 0); }
static napi_value ProveBatchAsync(napi_env env, napi_callback_info info) { return prove_common(env, info, 1); }
/* (h, msg Bx32, proofs, offsets (B u64 LE), lengths (B u64 LE), seeds Bx32 | null) -> {ok: B bytes, status: B i32} */
static napi_value verify_common(napi_env env, napi_callback_info info, int async) {
    napi_value argv[6];
    if (!get_args(env, info, 6, argv)) return NULL;
    Handle *h = get_handle(env, argv[0], 0);
    uint8_t *msg, *proofs, *offs, *lens, *seeds;
    size_t lm, lp, lo, ll, ls;
    if (!h || !get_bytes(env, argv[1], &msg, &lm) || !get_bytes(env, argv[2], &proofs, &lp) || !get_bytes(env, argv[3], &offs, &lo) ||
        !get_bytes(env, argv[4], &lens, &ll) || !get_bytes(env, argv[5], &seeds, &ls))
        return NULL;
    size_t B = lm / 32;
    if (!B || lm != 32 * B || lo != 8 * B || ll != 8 * B || (seeds && ls != 32 * B) || !proofs) {
        napi_throw_range_error(env, NULL, "verifyBatch: B message hashes, B offsets, B lengths, B seeds or null");
        return NULL;
    }
    Job *j = calloc(1, sizeof *j);
    if (!j) return throw_text(env, ZK_E_BUFFER, "out of memory");
    j->verify = 1, j->h = h, j->B = B, j->async = async;
    j->msg = dup_bytes(msg, lm), j->off = (uint64_t *)dup_bytes(offs, lo), j->len = (uint64_t *)dup_bytes(lens, ll);
    j->seeds = seeds ? dup_bytes(seeds, ls) : NULL;
    j->ok = malloc(B), j->status = malloc(4 * B);
    j->proofs_in = proofs;
    int okk = j->msg && j->off && j->len && j->ok && j->status && (!seeds || j->seeds);
    for (size_t b = 0; okk && b < B; b++) okk = j->off[b] <= lp && j->len[b] <= lp - j->off[b];
    if (okk && async) okk = napi_create_reference(env, argv[2], 1, &j->proofs_ref) == napi_ok && napi_create_reference(env, argv[0], 1, &j->h_ref) == napi_ok;
    if (!okk) {
        job_free(env, j);
        napi_throw_range_error(env, NULL, "verifyBatch: out of memory, or offsets / lengths beyond the proof buffer");
        return NULL;
    }
    return job_run(env, j, "zkattest.verifyBatch");
}
static napi_value VerifyBatch(napi_env env, napi_callback_info info) { return verify_common(env, info, 0); }
static napi_value VerifyBatchAsync(napi_env env, napi_callback_info info) { return verify_common(env, info, 1); }

/* ---- streamed batches: several jobs of one handle in flight (zk_pool_prove_submit / _wait).
 * proveSubmit / verifySubmit return a Promise at once and append the job to the handle's queue.  The engine wants one caller at a
 * time and its waits in submission order, so every engine call of the queue -- a submit or a wait -- is ONE libuv work item, and the
 * next one is chosen on the main thread when it completes: submit the oldest job not yet submitted while fewer than `inflight`
 * (setOption 'inflight', default 3) are inside the engine, otherwise wait for the oldest.  With three jobs queued the engine sees
 * submit(0) submit(1) submit(2) wait(0) submit(3) wait(1) ...  A handle with queued jobs is busy for every other call. */
static void stream_kick(napi_env env, Handle *h);
static void stream_execute(napi_env env, void *data) { /* worker thread: no N-API calls */
    Job *j = data;
    zk_pool *p = j->h->pool;
    if (j->op == 0) {
        if (j->verify) {
            j->rc = zk_pool_verify_submit(p, j->B, j->msg, j->proofs_in, j->off, j->len, j->seeds, j->ok, j->status, &j->pj);
        } else {
            zk_rng rng = {ZK_RNG_SEED, j->seeds, 0};
            j->rc = zk_pool_prove_submit(p, j->B, j->msg, j->sig, j->pk, j->which, &rng, j->out_ext, j->out_ext_cap, j->off, j->len, j->status, &j->pj);
        }
    } else {
        j->rc = j->verify ? zk_pool_verify_wait(p, j->pj) : zk_pool_prove_wait(p, j->pj);
    }
    if (j->rc != ZK_OK) snprintf(j->err, sizeof j->err, "%s: %s", zk_strerror(j->rc), zk_pool_last_error(p));
}
static napi_value stream_result(napi_env env, Job *j) {
    napi_value v, out = NULL;
    if (napi_create_object(env, &v) != napi_ok) return NULL;
    if (j->verify) {
        set_prop(env, v, "ok", new_buffer(env, j->ok, j->B));
    } else {
        uint64_t end = 0;
        for (size_t b = 0; b < j->B; b++)
            if (j->off[b] + j->len[b] > end) end = j->off[b] + j->len[b];
        napi_value used;
        if (napi_get_reference_value(env, j->out_ref, &out) != napi_ok || napi_create_double(env, (double)end, &used) != napi_ok) return NULL;
        set_prop(env, v, "proofs", out), set_prop(env, v, "used", used);
        set_prop(env, v, "offsets", new_buffer(env, j->off, 8 * j->B)), set_prop(env, v, "lengths", new_buffer(env, j->len, 8 * j->B));
    }
    set_prop(env, v, "status", new_buffer(env, j->status, 4 * j->B));
    return v;
}
static void stream_settle(napi_env env, Job *j, int ok) {
    napi_value v = ok ? stream_result(env, j) : NULL;
    if (v) {
        napi_resolve_deferred(env, j->deferred, v);
    } else {
        napi_value msg, e;
        napi_create_string_utf8(env, j->rc != ZK_OK ? j->err : "streamed batch failed", NAPI_AUTO_LENGTH, &msg);
        napi_create_error(env, NULL, msg, &e);
        napi_reject_deferred(env, j->deferred, e);
    }
    job_free(env, j);
}
static void stream_unlink(Handle *h, Job *j) {
    Job **pp = &h->sq_head, *prev = NULL;
    while (*pp && *pp != j) prev = *pp, pp = &(*pp)->next;
    if (*pp) *pp = j->next;
    if (h->sq_tail == j) h->sq_tail = prev;
}
static void stream_complete(napi_env env, napi_status status, void *data) { /* main thread */
    Job *j = data;
    Handle *h = j->h;
    h->sq_running = 0;
    napi_delete_async_work(env, j->work);
    int ok = status == napi_ok && j->rc == ZK_OK;
    if (j->op == 0 && ok) {
        j->submitted = 1, h->sq_submitted++;
    } else { /* a finished wait, or a submit the engine refused: the job leaves the queue */
        if (j->op == 1) h->sq_submitted--;
        stream_unlink(h, j);
        stream_settle(env, j, ok);
    }
    if (!h->sq_head) {
        h->busy = 0;
        if (h->orphaned) {
            handle_release(h);
            return;
        }
    }
    stream_kick(env, h);
}
static void stream_kick(napi_env env, Handle *h) {
    while (!h->sq_running && h->sq_head) {
        Job *u = h->sq_head;
        while (u && u->submitted) u = u->next;
        Job *j = u && (uint32_t)h->sq_submitted < (h->inflight ? h->inflight : 3) ? u : h->sq_head;
        j->op = j == u ? 0 : 1;
        napi_value rn;
        if (napi_create_string_utf8(env, "zkattest.stream", NAPI_AUTO_LENGTH, &rn) == napi_ok &&
            napi_create_async_work(env, NULL, rn, stream_execute, stream_complete, j, &j->work) == napi_ok) {
            if (napi_queue_async_work(env, j->work) == napi_ok) {
                h->sq_running = 1;
                return;
            }
            napi_delete_async_work(env, j->work);
        }
        /* could not queue: a job that is inside the engine has to be waited for here, the others are dropped */
        if (j->submitted) {
            j->rc = j->verify ? zk_pool_verify_wait(h->pool, j->pj) : zk_pool_prove_wait(h->pool, j->pj);
            h->sq_submitted--;
        }
        snprintf(j->err, sizeof j->err, "could not queue the streamed batch");
        j->rc = j->rc == ZK_OK ? ZK_E_BUFFER : j->rc;
        stream_unlink(h, j);
        stream_settle(env, j, 0);
    }
    if (!h->sq_head) h->busy = 0;
}
static napi_value stream_enqueue(napi_env env, Job *j, napi_value hv) {
    napi_value promise;
    Handle *h = j->h;
    if (napi_create_reference(env, hv, 1, &j->h_ref) != napi_ok || napi_create_promise(env, &j->deferred, &promise) != napi_ok) {
        job_free(env, j);
        return throw_text(env, ZK_E_BUFFER, "could not queue the streamed batch");
    }
    j->stream = 1;
    if (h->sq_tail) h->sq_tail->next = j;
    else h->sq_head = j;
    h->sq_tail = j;
    h->busy = 1;
    stream_kick(env, h);
    return promise;
}
/* a handle that is busy with streamed jobs accepts more of them; one that runs an exclusive batch does not */
static Handle *get_stream_handle(napi_env env, napi_value v) {
    Handle *h = get_handle(env, v, 1);
    if (h && h->busy && !h->sq_head) {
        napi_throw_error(env, NULL, "the engine is busy with an asynchronous batch (one batch at a time per engine)");
        return NULL;
    }
    return h;
}
/* (h, msg Bx32, sig Bx64, pk Bx64, which Bx4, seeds Bx32, out: page-locked Buffer from hostAlloc)
 *   -> Promise<{proofs: out, used: bytes, offsets, lengths, status}>; `out` belongs to the job until its Promise settles */
static napi_value ProveSubmit(napi_env env, napi_callback_info info) {
    napi_value argv[7];
    if (!get_args(env, info, 7, argv)) return NULL;
    Handle *h = get_stream_handle(env, argv[0]);
    uint8_t *msg, *sig, *pk, *which, *seeds, *out;
    size_t lm, ls, lp, lw, lse, lo;
    if (!h || !get_bytes(env, argv[1], &msg, &lm) || !get_bytes(env, argv[2], &sig, &ls) || !get_bytes(env, argv[3], &pk, &lp) ||
        !get_bytes(env, argv[4], &which, &lw) || !get_bytes(env, argv[5], &seeds, &lse) || !get_bytes(env, argv[6], &out, &lo))
        return NULL;
    size_t B = lm / 32;
    if (!B || lm != 32 * B || ls != 64 * B || lp != 64 * B || lw != 4 * B || lse != 32 * B || !out) {
        napi_throw_range_error(env, NULL, "proveSubmit: per proof 32-byte msgHash, 64-byte signature, 64-byte public key, u32 index, 32-byte seed; a page-locked output Buffer");
        return NULL;
    }
    Job *j = calloc(1, sizeof *j);
    if (!j) return throw_text(env, ZK_E_BUFFER, "out of memory");
    j->h = h, j->B = B, j->async = 1, j->out_ext = out, j->out_ext_cap = lo;
    j->msg = dup_bytes(msg, lm), j->sig = dup_bytes(sig, ls), j->pk = dup_bytes(pk, lp), j->seeds = dup_bytes(seeds, lse);
    j->which = (uint32_t *)dup_bytes(which, lw);
    j->off = malloc(8 * B), j->len = malloc(8 * B), j->status = malloc(4 * B);
    if (!j->msg || !j->sig || !j->pk || !j->seeds || !j->which || !j->off || !j->len || !j->status ||
        napi_create_reference(env, argv[6], 1, &j->out_ref) != napi_ok) {
        job_free(env, j);
        return throw_text(env, ZK_E_BUFFER, "out of memory");
    }
    return stream_enqueue(env, j, argv[0]);
}
/* (h, msg Bx32, proofs: page-locked Buffer, offsets, lengths, seeds | null) -> Promise<{ok, status}> */
static napi_value VerifySubmit(napi_env env, napi_callback_info info) {
    napi_value argv[6];
    if (!get_args(env, info, 6, argv)) return NULL;
    Handle *h = get_stream_handle(env, argv[0]);
    uint8_t *msg, *proofs, *offs, *lens, *seeds;
    size_t lm, lp, lo, ll, ls;
    if (!h || !get_bytes(env, argv[1], &msg, &lm) || !get_bytes(env, argv[2], &proofs, &lp) || !get_bytes(env, argv[3], &offs, &lo) ||
        !get_bytes(env, argv[4], &lens, &ll) || !get_bytes(env, argv[5], &seeds, &ls))
        return NULL;
    size_t B = lm / 32;
    if (!B || lm != 32 * B || lo != 8 * B || ll != 8 * B || (seeds && ls != 32 * B) || !proofs) {
        napi_throw_range_error(env, NULL, "verifySubmit: B message hashes, B offsets, B lengths, B seeds or null");
        return NULL;
    }
    Job *j = calloc(1, sizeof *j);
    if (!j) return throw_text(env, ZK_E_BUFFER, "out of memory");
    j->verify = 1, j->h = h, j->B = B, j->async = 1;
    j->msg = dup_bytes(msg, lm), j->off = (uint64_t *)dup_bytes(offs, lo), j->len = (uint64_t *)dup_bytes(lens, ll);
    j->seeds = seeds ? dup_bytes(seeds, ls) : NULL;
    j->ok = malloc(B), j->status = malloc(4 * B);
    j->proofs_in = proofs;
    int okk = j->msg && j->off && j->len && j->ok && j->status && (!seeds || j->seeds);
    for (size_t b = 0; okk && b < B; b++) okk = j->off[b] <= lp && j->len[b] <= lp - j->off[b];
    if (okk) okk = napi_create_reference(env, argv[2], 1, &j->proofs_ref) == napi_ok;
    if (!okk) {
        job_free(env, j);
        napi_throw_range_error(env, NULL, "verifySubmit: out of memory, or offsets / lengths beyond the proof buffer");
        return NULL;
    }
    return stream_enqueue(env, j, argv[0]);
}

static napi_value HostAlloc(napi_env env, napi_callback_info info) { /* (bytes) -> page-locked Buffer (zk_host_alloc) */
    napi_value argv[1];
    if (!get_args(env, info, 1, argv)) return NULL;
    double n;
    NAPI_OK(napi_get_value_double(env, argv[0], &n));
    if (!(n >= 1) || n > 1e12) {
        napi_throw_range_error(env, NULL, "hostAlloc: size in bytes");
        return NULL;
    }
    void *p = zk_host_alloc((size_t)n);
    if (!p) return throw_text(env, ZK_E_BUFFER, "zk_host_alloc failed");
    napi_value b;
    if (napi_create_external_buffer(env, (size_t)n, p, slab_finalize, (void *)1, &b) != napi_ok) {
        zk_host_free(p);
        napi_throw_error(env, NULL, "could not wrap the page-locked buffer");
        return NULL;
    }
    return b;
}
static napi_value HardenedH(napi_env env, napi_callback_info info) { /* (tag: Buffer) -> {nistH, tomH}   zk_hardened_h: host-only */
    napi_value argv[1];
    if (!get_args(env, info, 1, argv)) return NULL;
    uint8_t *tag;
    size_t lt;
    if (!get_bytes(env, argv[0], &tag, &lt)) return NULL;
    uint8_t a[64], b[72];
    zk_status st = zk_hardened_h(tag, lt, a, b);
    if (st != ZK_OK) return throw_text(env, st, "");
    napi_value o;
    NAPI_OK(napi_create_object(env, &o));
    set_prop(env, o, "nistH", new_buffer(env, a, 64)), set_prop(env, o, "tomH", new_buffer(env, b, 72));
    return o;
}
static napi_value ProofToJson(napi_env env, napi_callback_info info) { /* (proof: Buffer) -> string   (writeJson, src/serde.ts:34-36) */
    napi_value argv[1];
    if (!get_args(env, info, 1, argv)) return NULL;
    uint8_t *p;
    size_t lp;
    if (!get_bytes(env, argv[0], &p, &lp)) return NULL;
    uint64_t n = 0, cap = 5 * (uint64_t)lp + 4096;   /* the text is ~3.5x the binary proof: one conversion in the common case */
    char *s = xmalloc(env, cap + 1);
    if (!s) return NULL;
    zk_status st = zk_proof_to_json(p, lp, s, cap, &n);
    if (st == ZK_E_BUFFER) {
        free(s);
        s = xmalloc(env, n + 1);
        if (!s) return NULL;
        st = zk_proof_to_json(p, lp, s, n, &n);
    }
    napi_value r = NULL;
    if (st == ZK_OK) napi_create_string_utf8(env, s, n, &r);
    free(s);
    return st == ZK_OK ? r : throw_text(env, st, "");
}
static napi_value ProofFromJson(napi_env env, napi_callback_info info) { /* (text: string) -> Buffer   (readJson, src/serde.ts:21-32) */
    napi_value argv[1];
    if (!get_args(env, info, 1, argv)) return NULL;
    size_t len = 0;
    NAPI_OK(napi_get_value_string_utf8(env, argv[0], NULL, 0, &len));
    if (len > ((size_t)64 << 20)) return throw_text(env, ZK_E_BAD_ENCODING, "JSON text too long");
    char *s = xmalloc(env, len + 1);
    if (!s) return NULL;
    if (napi_get_value_string_utf8(env, argv[0], s, len + 1, &len) != napi_ok) {
        free(s);
        napi_throw_type_error(env, NULL, "expected a string");
        return NULL;
    }
    uint64_t n = 0, cap = len / 2 + 64;   /* at least two hex digits per byte: the binary proof is shorter than half the text */
    napi_value r = NULL;
    uint8_t *b = malloc(cap + 1);
    zk_status st = b ? zk_proof_from_json(s, len, b, cap, &n) : ZK_E_BUFFER;
    if (b && st == ZK_E_BUFFER && n > cap) {
        free(b);
        b = malloc(n + 1);
        st = b ? zk_proof_from_json(s, len, b, n, &n) : ZK_E_BUFFER;
    }
    if (st == ZK_OK) r = new_buffer(env, b, n);
    free(b);
    free(s);
    return st == ZK_OK && r ? r : throw_text(env, st ? st : ZK_E_BUFFER, "");
}
static napi_value KeysToInts(napi_env env, napi_callback_info info) { /* (h, pk: n x 64 bytes) -> {keys: n x 32, status}  (keyToInt) */
    napi_value argv[2];
    if (!get_args(env, info, 2, argv)) return NULL;
    Handle *h = get_handle(env, argv[0], 0);
    uint8_t *pk;
    size_t lp;
    if (!h || !get_bytes(env, argv[1], &pk, &lp)) return NULL;
    size_t n = lp / 64;
    uint8_t *keys = malloc(32 * n + 1);
    int32_t *status = malloc(4 * n + 4);
    zk_ctx *ctx = zk_pool_ctx(h->pool, 0);
    zk_status st = keys && status ? zk_keys_to_ints(ctx, n, pk, keys, status) : ZK_E_BUFFER;
    napi_value o = NULL;
    if (st == ZK_OK && napi_create_object(env, &o) == napi_ok) set_prop(env, o, "keys", new_buffer(env, keys, 32 * n)), set_prop(env, o, "status", new_buffer(env, status, 4 * n));
    free(keys), free(status);
    return st == ZK_OK ? o : throw_text(env, st, zk_last_error(ctx));
}

/* ---- whole batches of proofs <-> JSON texts on every host core (zk_proofs_to_json_batch / zk_proofs_from_json_batch), off the event
 * loop: (blob: Buffer, offsets: Buffer of n + 1 u64 LE, threads) -> Promise<{ blob, offsets, status }>.  `to` = 1: ZKA1 -> JSON. */
typedef struct {
    int to;
    uint64_t n;
    uint8_t *in;
    uint64_t *in_off;
    uint32_t threads;
    uint8_t *out;
    uint64_t out_cap, *out_off;
    int32_t *status;
    zk_status rc;
    napi_ref in_ref, off_ref;
    napi_deferred deferred;
    napi_async_work work;
} JsonJob;
static void json_job_free(napi_env env, JsonJob *j) {
    if (j->in_ref) napi_delete_reference(env, j->in_ref);
    if (j->off_ref) napi_delete_reference(env, j->off_ref);
    free(j->out), free(j->out_off), free(j->status), free(j);
}
static void json_execute(napi_env env, void *data) { /* worker thread: no N-API calls */
    JsonJob *j = data;
    for (int pass = 0; pass < 2; pass++) {
        j->rc = j->to ? zk_proofs_to_json_batch(j->n, j->in, j->in_off, (char *)j->out, j->out_cap, j->out_off, j->status, j->threads)
                      : zk_proofs_from_json_batch(j->n, (const char *)j->in, j->in_off, j->out, j->out_cap, j->out_off, j->status, j->threads);
        if (j->rc != ZK_E_BUFFER) break;
        free(j->out);                       /* the offsets are complete: come back with the exact size */
        j->out_cap = j->out_off[j->n] + 64;
        j->out = malloc(j->out_cap);
        if (!j->out) break;
    }
}
static void json_complete(napi_env env, napi_status status, void *data) { /* main thread */
    JsonJob *j = data;
    napi_value v = NULL, b = NULL;
    if (status == napi_ok && j->rc == ZK_OK && napi_create_object(env, &v) == napi_ok) {
        uint64_t len = j->out_off[j->n];
        if (napi_create_external_buffer(env, (size_t)len, j->out, slab_finalize, NULL, &b) == napi_ok) j->out = NULL; /* ownership moved */
        else b = new_buffer(env, j->out, (size_t)len);
        set_prop(env, v, "blob", b);
        set_prop(env, v, "offsets", new_buffer(env, j->out_off, 8 * (j->n + 1)));
        set_prop(env, v, "status", new_buffer(env, j->status, 4 * j->n));
        napi_resolve_deferred(env, j->deferred, v);
    } else {
        napi_value msg, e;
        napi_create_string_utf8(env, "batch JSON conversion failed", NAPI_AUTO_LENGTH, &msg);
        napi_create_error(env, NULL, msg, &e);
        napi_reject_deferred(env, j->deferred, e);
    }
    napi_delete_async_work(env, j->work);
    json_job_free(env, j);
}
static napi_value json_batch(napi_env env, napi_callback_info info, int to) {
    napi_value argv[3], promise, rn;
    if (!get_args(env, info, 3, argv)) return NULL;
    uint8_t *in, *off;
    size_t li, lo;
    uint32_t threads = 0;
    if (!get_bytes(env, argv[0], &in, &li) || !get_bytes(env, argv[1], &off, &lo) || napi_get_value_uint32(env, argv[2], &threads) != napi_ok || lo < 8 || lo % 8) {
        napi_throw_type_error(env, NULL, "expected (blob: Buffer, offsets: Buffer of n + 1 u64, threads: number)");
        return NULL;
    }
    JsonJob *j = calloc(1, sizeof *j);
    if (!j) return throw_text(env, ZK_E_BUFFER, "out of memory");
    j->to = to, j->n = lo / 8 - 1, j->in = in, j->in_off = (uint64_t *)off, j->threads = threads;
    if (j->in_off[j->n] > li) {
        free(j);
        napi_throw_range_error(env, NULL, "offsets run past the blob");
        return NULL;
    }
    j->out_cap = to ? 4 * (uint64_t)li + 4096 * (j->n + 1) : (uint64_t)li / 3 + 64 * (j->n + 1);
    j->out = malloc(j->out_cap), j->out_off = malloc(8 * (j->n + 1)), j->status = malloc(4 * (j->n ? j->n : 1));
    if (!j->out || !j->out_off || !j->status || napi_create_reference(env, argv[0], 1, &j->in_ref) != napi_ok ||
        napi_create_reference(env, argv[1], 1, &j->off_ref) != napi_ok || napi_create_promise(env, &j->deferred, &promise) != napi_ok ||
        napi_create_string_utf8(env, to ? "zkattest:toJsonBatch" : "zkattest:fromJsonBatch", NAPI_AUTO_LENGTH, &rn) != napi_ok ||
        napi_create_async_work(env, NULL, rn, json_execute, json_complete, j, &j->work) != napi_ok || napi_queue_async_work(env, j->work) != napi_ok) {
        json_job_free(env, j);
        napi_throw_error(env, NULL, "could not queue the conversion");
        return NULL;
    }
    return promise;
}
static napi_value ProofsToJsonBatch(napi_env env, napi_callback_info info) { return json_batch(env, info, 1); }
static napi_value ProofsFromJsonBatch(napi_env env, napi_callback_info info) { return json_batch(env, info, 0); }

static napi_value Init(napi_env env, napi_value exports) {
    static const struct {
        const char *name;
        napi_callback fn;
    } fns[] = {{"createPool", CreatePool},       {"destroyPool", DestroyPool},       {"poolInfo", PoolInfo},       {"setOption", SetOption},
               {"setParams", SetParams},         {"setRing", SetRing},               {"synthParams", SynthParams}, {"synthWorkload", SynthWorkload},
               {"proveBatch", ProveBatch},       {"verifyBatch", VerifyBatch},       {"proveBatchAsync", ProveBatchAsync},
               {"verifyBatchAsync", VerifyBatchAsync}, {"proofToJson", ProofToJson}, {"proofFromJson", ProofFromJson},
               {"keysToInts", KeysToInts},       {"hostAlloc", HostAlloc},           {"hardenedH", HardenedH},
               {"proofsToJsonBatch", ProofsToJsonBatch}, {"proofsFromJsonBatch", ProofsFromJsonBatch},
               {"proveSubmit", ProveSubmit},     {"verifySubmit", VerifySubmit}};
    for (size_t i = 0; i < sizeof fns / sizeof fns[0]; i++) {
        napi_value f;
        if (napi_create_function(env, fns[i].name, NAPI_AUTO_LENGTH, fns[i].fn, NULL, &f) != napi_ok) return NULL;
        napi_set_named_property(env, exports, fns[i].name, f);
    }
    return exports;
}
NAPI_MODULE(NODE_GYP_MODULE_NAME, Init)
