// zk_pool: the GPUs of one node behind one handle (include/zkattest.h, "several GPUs").
//
// Proofs are independent given (params, ring) -- the reference proves them one after the other on one thread
// (src/zkpAttestList.ts:104-145) -- so a batch shards by contiguous ranges: device i of G gets proofs
// [i*B/G, (i+1)*B/G), one host thread per device drives that device's zk_ctx through the ordinary single-device entry
// points, and nothing is exchanged while proving or verifying.  The only data every device needs from one place is the key
// ring: zk_pool_set_ring uploads it ONCE to the first device and broadcasts it device-to-device over xGMI with RCCL
// (ncclBroadcast, one communicator per device in this process; librccl is loaded on demand so that a single-GPU user never
// pays for it), or with hipMemcpyPeer when RCCL is not usable (library missing, the same device listed twice, init failure).
// The fixed-base tables and the per-ring table E are rebuilt locally on every device, concurrently: that is cheaper than
// moving 47 GB of tables per device across the links.
#include <dlfcn.h>
#include <thread>
#include "ctx.h"

namespace {
// the few RCCL entry points used, resolved with dlsym (no link-time dependency on librccl)
typedef struct ncclComm* ncclComm_t;
typedef int ncclResult_t;   // ncclSuccess = 0
enum { kNcclUint8 = 1 };    // ncclDataType_t: ncclInt8 = 0, ncclUint8 = 1 (rccl.h)
struct Rccl {
    void* h = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Broadcast)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool load() {
        if (h) return true;
        if (getenv("ZKATTEST_NO_RCCL")) return false;
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
            h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (h) break;
        }
        if (!h) return false;
        CommInitAll = (decltype(CommInitAll))dlsym(h, "ncclCommInitAll");
        CommDestroy = (decltype(CommDestroy))dlsym(h, "ncclCommDestroy");
        GroupStart = (decltype(GroupStart))dlsym(h, "ncclGroupStart");
        GroupEnd = (decltype(GroupEnd))dlsym(h, "ncclGroupEnd");
        Broadcast = (decltype(Broadcast))dlsym(h, "ncclBroadcast");
        GetErrorString = (decltype(GetErrorString))dlsym(h, "ncclGetErrorString");
        if (!CommInitAll || !CommDestroy || !GroupStart || !GroupEnd || !Broadcast) {
            dlclose(h), h = nullptr;
            return false;
        }
        return true;
    }
};
}  // namespace

struct zk_pool {
    std::vector<zk_ctx*> ctx;
    std::vector<int> dev;
    std::string err;
    Rccl rccl;
    std::vector<ncclComm_t> comms;     // one per device when RCCL is in use
    bool comms_tried = false;
    const char* transport = "none";    // how the last ring reached the devices: "single", "rccl", "peer-copy"
};

// runs f(i) for every device on its own host thread and returns the first non-zero status
template <class F>
static zk_status pool_each(zk_pool* p, F f) {
    const int G = (int)p->ctx.size();
    std::vector<zk_status> st(G, ZK_OK);
    if (G == 1) {
        st[0] = f(0);
    } else {
        std::vector<std::thread> th;
        for (int i = 0; i < G; i++) th.emplace_back([&, i] { st[i] = f(i); });
        for (auto& t : th) t.join();
    }
    for (int i = 0; i < G; i++)
        if (st[i]) {
            char buf[64];
            snprintf(buf, sizeof buf, "device %d (slot %d): ", p->dev[i], i);
            p->err = std::string(buf) + zk_strerror(st[i]) + " " + zk_last_error(p->ctx[i]);
            return st[i];
        }
    return ZK_OK;
}

extern "C" zk_status zk_pool_create(const int* device_ids, int n_dev, zk_pool** out) {
    if (!out || !device_ids || n_dev < 1 || n_dev > 64) return ZK_E_ARG;
    zk_pool* p = new zk_pool();
    *out = p;
    p->ctx.assign(n_dev, nullptr);
    p->dev.assign(device_ids, device_ids + n_dev);
    return pool_each(p, [&](int i) { return zk_ctx_create(p->dev[i], &p->ctx[i]); });   // the generator tables are built concurrently
}
extern "C" void zk_pool_destroy(zk_pool* p) {
    if (!p) return;
    for (size_t i = 0; i < p->comms.size(); i++)
        if (p->comms[i]) p->rccl.CommDestroy(p->comms[i]);
    for (auto c : p->ctx) zk_ctx_destroy(c);
    delete p;
}
extern "C" int zk_pool_size(const zk_pool* p) { return p ? (int)p->ctx.size() : 0; }
extern "C" zk_ctx* zk_pool_ctx(zk_pool* p, int i) { return p && i >= 0 && i < (int)p->ctx.size() ? p->ctx[i] : nullptr; }
extern "C" const char* zk_pool_last_error(const zk_pool* p) { return p ? p->err.c_str() : ""; }
extern "C" const char* zk_pool_ring_transport(const zk_pool* p) { return p ? p->transport : ""; }

extern "C" void zk_pool_shard(const zk_pool* p, uint64_t B, int i, uint64_t* first, uint64_t* count) {
    const uint64_t G = p ? p->ctx.size() : 1;
    uint64_t f0 = B * (uint64_t)i / G, f1 = B * (uint64_t)(i + 1) / G;
    if (first) *first = f0;
    if (count) *count = f1 - f0;
}

extern "C" zk_status zk_pool_set_params(zk_pool* p, const uint8_t nist_h[64], const uint8_t tom_g[72], const uint8_t tom_h[72], uint32_t sec) {
    if (!p) return ZK_E_ARG;
    return pool_each(p, [&](int i) { return zk_ctx_set_params(p->ctx[i], nist_h, tom_g, tom_h, sec); });
}

extern "C" zk_status zk_pool_set_ring(zk_pool* p, const uint8_t* keys, uint64_t nkeys) {
    if (!p || !keys || nkeys < 2) return ZK_E_ARG;
    const int G = (int)p->ctx.size();
    const size_t bytes = 32 * (size_t)nkeys;
    std::vector<void*> d(G, nullptr);
    auto release = [&] {
        for (int i = 0; i < G; i++)
            if (d[i]) {
                hipSetDevice(p->dev[i]);
                hipFree(d[i]);
            }
    };
    auto fail = [&](const char* what, hipError_t e) {
        p->err = std::string(what) + ": " + hipGetErrorString(e);
        release();
        return (zk_status)ZK_E_DEVICE;
    };
    for (int i = 0; i < G; i++) {
        hipError_t e = hipSetDevice(p->dev[i]);
        if (e == hipSuccess) e = hipMalloc(&d[i], bytes);
        if (e != hipSuccess) return fail("ring staging buffer", e);
    }
    hipSetDevice(p->dev[0]);
    hipError_t e = hipMemcpy(d[0], keys, bytes, hipMemcpyHostToDevice);   // the ring crosses PCIe once
    if (e != hipSuccess) return fail("ring upload", e);
    p->transport = "single";
    if (G > 1) {
        bool distinct = true;
        for (int i = 0; i < G; i++)
            for (int j = 0; j < i; j++) distinct = distinct && p->dev[i] != p->dev[j];
        if (distinct && !p->comms_tried) {   // one communicator per device, created once per pool
            p->comms_tried = true;
            if (p->rccl.load()) {
                p->comms.assign(G, nullptr);
                if (p->rccl.CommInitAll(p->comms.data(), G, p->dev.data()) != 0) p->comms.clear();
            }
        }
        bool done = false;
        if (distinct && !p->comms.empty()) {
            ncclResult_t r = p->rccl.GroupStart();
            for (int i = 0; i < G && r == 0; i++) {
                hipSetDevice(p->dev[i]);
                r = p->rccl.Broadcast(d[i], d[i], bytes, kNcclUint8, 0, p->comms[i], p->ctx[i]->stream);   // in place; only the root's buffer is read
            }
            ncclResult_t r2 = p->rccl.GroupEnd();
            if (r == 0 && r2 == 0) {
                done = true;
                for (int i = 0; i < G; i++) {
                    hipSetDevice(p->dev[i]);
                    if (hipStreamSynchronize(p->ctx[i]->stream) != hipSuccess) done = false;
                }
            }
            if (done) p->transport = "rccl";
        }
        if (!done) {   // device-to-device copies (xGMI where the devices are peers)
            for (int i = 1; i < G; i++) {
                e = hipMemcpyPeer(d[i], p->dev[i], d[0], p->dev[0], bytes);
                if (e != hipSuccess) return fail("ring peer copy", e);
            }
            p->transport = "peer-copy";
        }
    }
    zk_status zs = pool_each(p, [&](int i) { return zk_ctx_set_ring_device(p->ctx[i], d[i], nkeys); });   // pad, limb conversion, table E
    release();
    return zs;
}

extern "C" zk_status zk_pool_prove_batch(zk_pool* p, uint64_t B, const uint8_t* msg, const uint8_t* sig, const uint8_t* pk, const uint32_t* which,
                                         const zk_rng* rng, uint8_t* out, uint64_t out_cap, uint64_t* out_off, uint64_t* out_len, int32_t* status) {
    if (!p || !rng || !out_off || !out_len || !status || (B && (!msg || !sig || !pk || !which || !rng->data || !out))) return ZK_E_ARG;
    const uint64_t G = p->ctx.size();
    const uint64_t region = (out_cap / G) & ~(uint64_t)255;   // shard i writes into [i * region, (i+1) * region)
    return pool_each(p, [&](int i) -> zk_status {
        uint64_t first, cnt;
        zk_pool_shard(p, B, i, &first, &cnt);
        if (!cnt) return ZK_OK;
        zk_rng r = *rng;
        r.data = rng->data + (rng->mode == ZK_RNG_SEED ? 32 * first : 32 * first * rng->stride_blocks);
        std::vector<uint64_t> off(cnt + 1);
        zk_status zs = zk_prove_batch(p->ctx[i], cnt, msg + 32 * first, sig + 64 * first, pk + 64 * first, which + first, &r, out + region * i, region,
                                      off.data(), status + first);
        if (zs) return zs;
        for (uint64_t j = 0; j < cnt; j++) out_off[first + j] = region * i + off[j], out_len[first + j] = off[j + 1] - off[j];
        return ZK_OK;
    });
}

extern "C" zk_status zk_pool_verify_batch(zk_pool* p, uint64_t B, const uint8_t* msg, const uint8_t* proofs, const uint64_t* proof_off, const uint64_t* proof_len,
                                          const uint8_t* vseeds, uint8_t* ok, int32_t* status) {
    if (!p || (B && (!msg || !proofs || !proof_off || !proof_len || !ok || !status))) return ZK_E_ARG;
    return pool_each(p, [&](int i) -> zk_status {
        uint64_t first, cnt;
        zk_pool_shard(p, B, i, &first, &cnt);
        if (!cnt) return ZK_OK;
        // the single-device entry point takes proofs packed back to back: true inside a shard for the output of
        // zk_pool_prove_batch and for any fully packed buffer
        std::vector<uint64_t> off(cnt + 1);
        const uint64_t base = proof_off[first];
        if (base & 3) return ZK_E_ARG;
        for (uint64_t j = 0; j < cnt; j++) {
            if (proof_off[first + j] - base != (j ? off[j] : 0)) return ZK_E_ARG;   // a gap or an overlap inside the shard
            off[j] = proof_off[first + j] - base;
            off[j + 1] = off[j] + proof_len[first + j];
        }
        return zk_verify_batch(p->ctx[i], cnt, msg + 32 * first, proofs + base, off.data(), vseeds ? vseeds + 32 * first : nullptr, ok + first, status + first);
    });
}
