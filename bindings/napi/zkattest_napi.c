/* N-API addon over the C ABI (include/zkattest.h): the binding a maintainer of the reference would put behind
 * src/zkpAttestList.ts.  Plain C, N-API version 3 (the header of the Node in this image):
 *
 *   gcc -shared -fPIC -O2 -I/usr/include/node -I../../include zkattest_napi.c -o zkattest.node \
 *       -L../../zkp-ecdsa_amd/lib -lzkattest_hip -Wl,-rpath,<abs path of zkp-ecdsa_amd/lib>
 *
 * Calls are synchronous here; a production façade would wrap proveBatch / verifyBatch in napi_create_async_work so
 * that the TypeScript signatures stay Promise-returning without blocking the event loop (the context is not re-entrant:
 * one batch in flight per context).  All buffers are caller-visible Node Buffers / typed arrays; the engine copies. */
#define NAPI_VERSION 3
#include <node_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "zkattest.h"

#define NAPI_OK(call)                                          \
    do {                                                       \
        if ((call) != napi_ok) {                               \
            napi_throw_error(env, NULL, "N-API call failed: " #call); \
            return NULL;                                       \
        }                                                      \
    } while (0)

static napi_value throw_status(napi_env env, zk_ctx *ctx, zk_status st) {
    char msg[512];
    const char *detail = ctx ? zk_last_error(ctx) : "";
    snprintf(msg, sizeof msg, "%s%s%s", zk_strerror(st), detail && detail[0] ? ": " : "", detail ? detail : "");
    napi_throw_error(env, NULL, msg); /* the reference's error texts: 'point not in group', 'T[i] is at infinity', ... */
    return NULL;
}
static int get_args(napi_env env, napi_callback_info info, size_t want, napi_value *argv) {
    size_t argc = want;
    if (napi_get_cb_info(env, info, &argc, argv, NULL, NULL) != napi_ok || argc < want) {
        napi_throw_type_error(env, NULL, "wrong number of arguments");
        return 0;
    }
    return 1;
}
static zk_ctx *get_ctx(napi_env env, napi_value v) {
    void *p = NULL;
    if (napi_get_value_external(env, v, &p) != napi_ok || !p) {
        napi_throw_type_error(env, NULL, "expected a context");
        return NULL;
    }
    return (zk_ctx *)p;
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
static void set_prop(napi_env env, napi_value obj, const char *name, napi_value v) { napi_set_named_property(env, obj, name, v); }

static napi_value CreateContext(napi_env env, napi_callback_info info) {
    napi_value argv[1];
    if (!get_args(env, info, 1, argv)) return NULL;
    int32_t dev = 0;
    NAPI_OK(napi_get_value_int32(env, argv[0], &dev));
    zk_ctx *ctx = NULL;
    zk_status st = zk_ctx_create(dev, &ctx);
    if (st != ZK_OK) {
        napi_value r = throw_status(env, ctx, st);
        if (ctx) zk_ctx_destroy(ctx);
        return r;
    }
    napi_value ext;
    NAPI_OK(napi_create_external(env, ctx, NULL, NULL, &ext));
    return ext;
}
static napi_value DestroyContext(napi_env env, napi_callback_info info) {
    napi_value argv[1];
    if (!get_args(env, info, 1, argv)) return NULL;
    zk_ctx *ctx = get_ctx(env, argv[0]);
    if (ctx) zk_ctx_destroy(ctx);
    return NULL;
}
static napi_value SetParams(napi_env env, napi_callback_info info) { /* (ctx, nistH 64, tomG 72, tomH 72, secLevel) */
    napi_value argv[5];
    if (!get_args(env, info, 5, argv)) return NULL;
    zk_ctx *ctx = get_ctx(env, argv[0]);
    uint8_t *a, *b, *c;
    size_t la, lb, lc;
    uint32_t sec;
    if (!ctx || !get_bytes(env, argv[1], &a, &la) || !get_bytes(env, argv[2], &b, &lb) || !get_bytes(env, argv[3], &c, &lc)) return NULL;
    NAPI_OK(napi_get_value_uint32(env, argv[4], &sec));
    if (la != 64 || lb != 72 || lc != 72) {
        napi_throw_range_error(env, NULL, "params: h_NIST is 64 bytes, g and h of Tom-256 are 72 bytes (affine, big-endian)");
        return NULL;
    }
    zk_status st = zk_ctx_set_params(ctx, a, b, c, sec);
    return st == ZK_OK ? NULL : throw_status(env, ctx, st);
}
static napi_value SetRing(napi_env env, napi_callback_info info) { /* (ctx, keys: n x 32 bytes) */
    napi_value argv[2];
    if (!get_args(env, info, 2, argv)) return NULL;
    zk_ctx *ctx = get_ctx(env, argv[0]);
    uint8_t *k;
    size_t lk;
    if (!ctx || !get_bytes(env, argv[1], &k, &lk)) return NULL;
    zk_status st = zk_ctx_set_ring(ctx, k, lk / 32);
    return st == ZK_OK ? NULL : throw_status(env, ctx, st);
}
static napi_value SynthParams(napi_env env, napi_callback_info info) { /* (ctx, seed) -> {nistH, tomG, tomH} */
    napi_value argv[2];
    if (!get_args(env, info, 2, argv)) return NULL;
    zk_ctx *ctx = get_ctx(env, argv[0]);
    uint32_t seed;
    if (!ctx) return NULL;
    NAPI_OK(napi_get_value_uint32(env, argv[1], &seed));
    uint8_t a[64], b[72], c[72];
    zk_status st = zk_synth_params(ctx, seed, a, b, c);
    if (st != ZK_OK) return throw_status(env, ctx, st);
    napi_value o;
    NAPI_OK(napi_create_object(env, &o));
    set_prop(env, o, "nistH", new_buffer(env, a, 64)), set_prop(env, o, "tomG", new_buffer(env, b, 72)), set_prop(env, o, "tomH", new_buffer(env, c, 72));
    return o;
}
static napi_value SynthWorkload(napi_env env, napi_callback_info info) { /* (ctx, seed, nKeys, B) -> {ring, msg, sig, pk, which, seeds} */
    napi_value argv[4];
    if (!get_args(env, info, 4, argv)) return NULL;
    zk_ctx *ctx = get_ctx(env, argv[0]);
    uint32_t seed, nk, B;
    if (!ctx) return NULL;
    NAPI_OK(napi_get_value_uint32(env, argv[1], &seed));
    NAPI_OK(napi_get_value_uint32(env, argv[2], &nk));
    NAPI_OK(napi_get_value_uint32(env, argv[3], &B));
    uint8_t *ring = malloc(32 * (size_t)nk), *msg = malloc(32 * (size_t)B + 1), *sig = malloc(64 * (size_t)B + 1), *pk = malloc(64 * (size_t)B + 1), *seeds = malloc(32 * (size_t)B + 1);
    uint32_t *which = malloc(4 * (size_t)B + 4);
    zk_status st = zk_synth_workload(ctx, seed, nk, B, ring, msg, sig, pk, which, seeds);
    napi_value o = NULL;
    if (st == ZK_OK && napi_create_object(env, &o) == napi_ok) {
        set_prop(env, o, "ring", new_buffer(env, ring, 32 * (size_t)nk)), set_prop(env, o, "msg", new_buffer(env, msg, 32 * (size_t)B));
        set_prop(env, o, "sig", new_buffer(env, sig, 64 * (size_t)B)), set_prop(env, o, "pk", new_buffer(env, pk, 64 * (size_t)B));
        set_prop(env, o, "which", new_buffer(env, which, 4 * (size_t)B)), set_prop(env, o, "seeds", new_buffer(env, seeds, 32 * (size_t)B));
    }
    free(ring), free(msg), free(sig), free(pk), free(seeds), free(which);
    return st == ZK_OK ? o : throw_status(env, ctx, st);
}
/* (ctx, msg Bx32, sig Bx64, pk Bx64, which Bx4 (u32 LE), seeds Bx32) -> {proofs: Buffer, offsets: Buffer of B+1 u64 LE, status: Buffer of B i32} */
static napi_value ProveBatch(napi_env env, napi_callback_info info) {
    napi_value argv[6];
    if (!get_args(env, info, 6, argv)) return NULL;
    zk_ctx *ctx = get_ctx(env, argv[0]);
    uint8_t *msg, *sig, *pk, *which, *seeds;
    size_t lm, ls, lp, lw, lse;
    if (!ctx || !get_bytes(env, argv[1], &msg, &lm) || !get_bytes(env, argv[2], &sig, &ls) || !get_bytes(env, argv[3], &pk, &lp) ||
        !get_bytes(env, argv[4], &which, &lw) || !get_bytes(env, argv[5], &seeds, &lse))
        return NULL;
    size_t B = lm / 32;
    if (lm != 32 * B || ls != 64 * B || lp != 64 * B || lw != 4 * B || lse != 32 * B) {
        napi_throw_range_error(env, NULL, "proveBatch: per proof 32-byte msgHash, 64-byte signature, 64-byte public key, u32 index, 32-byte seed");
        return NULL;
    }
    uint64_t cap = zk_proof_max_size(ctx) * (B ? B : 1);
    uint8_t *out = malloc(cap ? cap : 1);
    uint64_t *off = malloc(8 * (B + 1));
    int32_t *status = malloc(4 * (B + 1));
    uint32_t *w32 = malloc(4 * (B + 1));
    memcpy(w32, which, 4 * B);
    zk_rng rng = {ZK_RNG_SEED, seeds, 0};
    zk_status st = zk_prove_batch(ctx, B, msg, sig, pk, w32, &rng, out, cap, off, status);
    napi_value o = NULL;
    if (st == ZK_OK && napi_create_object(env, &o) == napi_ok) {
        set_prop(env, o, "proofs", new_buffer(env, out, (size_t)off[B])), set_prop(env, o, "offsets", new_buffer(env, off, 8 * (B + 1)));
        set_prop(env, o, "status", new_buffer(env, status, 4 * B));
    }
    free(out), free(off), free(status), free(w32);
    return st == ZK_OK ? o : throw_status(env, ctx, st);
}
/* (ctx, msg Bx32, proofs, offsets (B+1 u64 LE), seeds Bx32 | null) -> {ok: Buffer of B bytes, status: Buffer of B i32} */
static napi_value VerifyBatch(napi_env env, napi_callback_info info) {
    napi_value argv[5];
    if (!get_args(env, info, 5, argv)) return NULL;
    zk_ctx *ctx = get_ctx(env, argv[0]);
    uint8_t *msg, *proofs, *offs, *seeds;
    size_t lm, lp, lo, ls;
    if (!ctx || !get_bytes(env, argv[1], &msg, &lm) || !get_bytes(env, argv[2], &proofs, &lp) || !get_bytes(env, argv[3], &offs, &lo) || !get_bytes(env, argv[4], &seeds, &ls))
        return NULL;
    size_t B = lm / 32;
    if (lo != 8 * (B + 1) || (seeds && ls != 32 * B)) {
        napi_throw_range_error(env, NULL, "verifyBatch: B message hashes, B + 1 offsets, B seeds or null");
        return NULL;
    }
    uint64_t *off = malloc(8 * (B + 1));
    memcpy(off, offs, 8 * (B + 1));
    uint8_t *ok = malloc(B + 1);
    int32_t *status = malloc(4 * (B + 1));
    zk_status st = off[B] <= lp ? zk_verify_batch(ctx, B, msg, proofs, off, seeds, ok, status) : ZK_E_ARG;
    napi_value o = NULL;
    if (st == ZK_OK && napi_create_object(env, &o) == napi_ok) set_prop(env, o, "ok", new_buffer(env, ok, B)), set_prop(env, o, "status", new_buffer(env, status, 4 * B));
    free(off), free(ok), free(status);
    return st == ZK_OK ? o : throw_status(env, ctx, st);
}
/* ---- asynchronous variants: the batch runs on a libuv worker thread, the caller gets a Promise (what keeps the reference's
 * `async function proveSignatureList(...)` signature without blocking the event loop).  One job at a time per context. */
typedef struct {
    int verify;
    zk_ctx *ctx;
    size_t B;
    uint8_t *msg, *sig, *pk, *seeds, *proofs_in;
    uint32_t *which;
    uint8_t *out;
    uint64_t cap, *off;
    int32_t *status;
    uint8_t *ok;
    zk_status rc;
    char err[256];
    napi_deferred deferred;
    napi_async_work work;
} Job;
static uint8_t *dup_bytes(const uint8_t *p, size_t n) {
    uint8_t *q = malloc(n ? n : 1);
    if (p && n) memcpy(q, p, n);
    return q;
}
static void job_free(Job *j) {
    free(j->msg), free(j->sig), free(j->pk), free(j->seeds), free(j->proofs_in), free(j->which), free(j->out), free(j->off), free(j->status), free(j->ok);
    free(j);
}
static void job_execute(napi_env env, void *data) { /* worker thread: no N-API calls here */
    Job *j = data;
    if (j->verify) j->rc = zk_verify_batch(j->ctx, j->B, j->msg, j->proofs_in, j->off, j->seeds, j->ok, j->status);
    else {
        zk_rng rng = {ZK_RNG_SEED, j->seeds, 0};
        j->rc = zk_prove_batch(j->ctx, j->B, j->msg, j->sig, j->pk, j->which, &rng, j->out, j->cap, j->off, j->status);
    }
    if (j->rc != ZK_OK) snprintf(j->err, sizeof j->err, "%s: %s", zk_strerror(j->rc), zk_last_error(j->ctx));
}
static void job_complete(napi_env env, napi_status status, void *data) { /* main thread */
    Job *j = data;
    napi_value v;
    if (status == napi_ok && j->rc == ZK_OK && napi_create_object(env, &v) == napi_ok) {
        if (j->verify) set_prop(env, v, "ok", new_buffer(env, j->ok, j->B));
        else set_prop(env, v, "proofs", new_buffer(env, j->out, (size_t)j->off[j->B])), set_prop(env, v, "offsets", new_buffer(env, j->off, 8 * (j->B + 1)));
        set_prop(env, v, "status", new_buffer(env, j->status, 4 * j->B));
        napi_resolve_deferred(env, j->deferred, v);
    } else {
        napi_value msg, e;
        napi_create_string_utf8(env, j->rc != ZK_OK ? j->err : "async work failed", NAPI_AUTO_LENGTH, &msg);
        napi_create_error(env, NULL, msg, &e);
        napi_reject_deferred(env, j->deferred, e);
    }
    napi_delete_async_work(env, j->work);
    job_free(j);
}
static napi_value job_start(napi_env env, Job *j, const char *name) {
    napi_value promise, rn;
    if (napi_create_promise(env, &j->deferred, &promise) != napi_ok || napi_create_string_utf8(env, name, NAPI_AUTO_LENGTH, &rn) != napi_ok ||
        napi_create_async_work(env, NULL, rn, job_execute, job_complete, j, &j->work) != napi_ok || napi_queue_async_work(env, j->work) != napi_ok) {
        job_free(j);
        napi_throw_error(env, NULL, "could not queue the batch");
        return NULL;
    }
    return promise;
}
static napi_value ProveBatchAsync(napi_env env, napi_callback_info info) { /* same arguments as proveBatch -> Promise of the same object */
    napi_value argv[6];
    if (!get_args(env, info, 6, argv)) return NULL;
    zk_ctx *ctx = get_ctx(env, argv[0]);
    uint8_t *msg, *sig, *pk, *which, *seeds;
    size_t lm, ls, lp, lw, lse;
    if (!ctx || !get_bytes(env, argv[1], &msg, &lm) || !get_bytes(env, argv[2], &sig, &ls) || !get_bytes(env, argv[3], &pk, &lp) ||
        !get_bytes(env, argv[4], &which, &lw) || !get_bytes(env, argv[5], &seeds, &lse))
        return NULL;
    size_t B = lm / 32;
    if (lm != 32 * B || ls != 64 * B || lp != 64 * B || lw != 4 * B || lse != 32 * B) {
        napi_throw_range_error(env, NULL, "proveBatchAsync: per proof 32-byte msgHash, 64-byte signature, 64-byte public key, u32 index, 32-byte seed");
        return NULL;
    }
    Job *j = calloc(1, sizeof *j);
    j->ctx = ctx, j->B = B;
    j->msg = dup_bytes(msg, lm), j->sig = dup_bytes(sig, ls), j->pk = dup_bytes(pk, lp), j->seeds = dup_bytes(seeds, lse);
    j->which = (uint32_t *)dup_bytes(which, lw);
    j->cap = zk_proof_max_size(ctx) * (B ? B : 1);
    j->out = malloc(j->cap ? j->cap : 1), j->off = malloc(8 * (B + 1)), j->status = malloc(4 * (B + 1));
    return job_start(env, j, "zkattest.proveBatch");
}
static napi_value VerifyBatchAsync(napi_env env, napi_callback_info info) { /* same arguments as verifyBatch -> Promise */
    napi_value argv[5];
    if (!get_args(env, info, 5, argv)) return NULL;
    zk_ctx *ctx = get_ctx(env, argv[0]);
    uint8_t *msg, *proofs, *offs, *seeds;
    size_t lm, lp, lo, ls;
    if (!ctx || !get_bytes(env, argv[1], &msg, &lm) || !get_bytes(env, argv[2], &proofs, &lp) || !get_bytes(env, argv[3], &offs, &lo) || !get_bytes(env, argv[4], &seeds, &ls))
        return NULL;
    size_t B = lm / 32;
    if (lo != 8 * (B + 1) || (seeds && ls != 32 * B)) {
        napi_throw_range_error(env, NULL, "verifyBatchAsync: B message hashes, B + 1 offsets, B seeds or null");
        return NULL;
    }
    Job *j = calloc(1, sizeof *j);
    j->verify = 1, j->ctx = ctx, j->B = B;
    j->msg = dup_bytes(msg, lm), j->proofs_in = dup_bytes(proofs, lp), j->off = (uint64_t *)dup_bytes(offs, lo);
    j->seeds = seeds ? dup_bytes(seeds, ls) : NULL;
    j->ok = malloc(B + 1), j->status = malloc(4 * (B + 1));
    if (j->off[B] > lp) {
        job_free(j);
        napi_throw_range_error(env, NULL, "verifyBatchAsync: offsets beyond the proof buffer");
        return NULL;
    }
    return job_start(env, j, "zkattest.verifyBatch");
}
static napi_value ProofToJson(napi_env env, napi_callback_info info) { /* (proof: Buffer) -> string   (writeJson, src/serde.ts:34-36) */
    napi_value argv[1];
    if (!get_args(env, info, 1, argv)) return NULL;
    uint8_t *p;
    size_t lp;
    if (!get_bytes(env, argv[0], &p, &lp)) return NULL;
    uint64_t n = 0;
    zk_status st = zk_proof_to_json(p, lp, NULL, 0, &n);
    if (st != ZK_OK && st != ZK_E_BUFFER) return throw_status(env, NULL, st);
    char *s = malloc(n + 1);
    st = zk_proof_to_json(p, lp, s, n, &n);
    napi_value r = NULL;
    if (st == ZK_OK) napi_create_string_utf8(env, s, n, &r);
    free(s);
    return st == ZK_OK ? r : throw_status(env, NULL, st);
}
static napi_value ProofFromJson(napi_env env, napi_callback_info info) { /* (text: string) -> Buffer   (readJson, src/serde.ts:21-32) */
    napi_value argv[1];
    if (!get_args(env, info, 1, argv)) return NULL;
    size_t len = 0;
    NAPI_OK(napi_get_value_string_utf8(env, argv[0], NULL, 0, &len));
    char *s = malloc(len + 1);
    NAPI_OK(napi_get_value_string_utf8(env, argv[0], s, len + 1, &len));
    uint64_t n = 0;
    zk_status st = zk_proof_from_json(s, len, NULL, 0, &n);
    napi_value r = NULL;
    if (st == ZK_OK || st == ZK_E_BUFFER) {
        uint8_t *b = malloc(n + 1);
        st = zk_proof_from_json(s, len, b, n, &n);
        if (st == ZK_OK) r = new_buffer(env, b, n);
        free(b);
    }
    free(s);
    return st == ZK_OK ? r : throw_status(env, NULL, st);
}
static napi_value KeysToInts(napi_env env, napi_callback_info info) { /* (ctx, pk: n x 64 bytes) -> {keys: n x 32, status}  (keyToInt) */
    napi_value argv[2];
    if (!get_args(env, info, 2, argv)) return NULL;
    zk_ctx *ctx = get_ctx(env, argv[0]);
    uint8_t *pk;
    size_t lp;
    if (!ctx || !get_bytes(env, argv[1], &pk, &lp)) return NULL;
    size_t n = lp / 64;
    uint8_t *keys = malloc(32 * n + 1);
    int32_t *status = malloc(4 * n + 4);
    zk_status st = zk_keys_to_ints(ctx, n, pk, keys, status);
    napi_value o = NULL;
    if (st == ZK_OK && napi_create_object(env, &o) == napi_ok) set_prop(env, o, "keys", new_buffer(env, keys, 32 * n)), set_prop(env, o, "status", new_buffer(env, status, 4 * n));
    free(keys), free(status);
    return st == ZK_OK ? o : throw_status(env, ctx, st);
}

static napi_value Init(napi_env env, napi_value exports) {
    static const struct {
        const char *name;
        napi_callback fn;
    } fns[] = {{"createContext", CreateContext}, {"destroyContext", DestroyContext}, {"setParams", SetParams}, {"setRing", SetRing},
               {"synthParams", SynthParams},     {"synthWorkload", SynthWorkload},   {"proveBatch", ProveBatch}, {"verifyBatch", VerifyBatch},
               {"proveBatchAsync", ProveBatchAsync}, {"verifyBatchAsync", VerifyBatchAsync}, {"proofToJson", ProofToJson},     {"proofFromJson", ProofFromJson},   {"keysToInts", KeysToInts}};
    for (size_t i = 0; i < sizeof fns / sizeof fns[0]; i++) {
        napi_value f;
        if (napi_create_function(env, fns[i].name, NAPI_AUTO_LENGTH, fns[i].fn, NULL, &f) != napi_ok) return NULL;
        napi_set_named_property(env, exports, fns[i].name, f);
    }
    return exports;
}
NAPI_MODULE(NODE_GYP_MODULE_NAME, Init)
