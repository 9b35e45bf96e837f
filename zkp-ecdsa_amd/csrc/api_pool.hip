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
#include <chrono>
#include <sched.h>
#include <sys/mman.h>
#include <sys/syscall.h>
#include <unistd.h>
#include <map>
#include <mutex>
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
    std::string served_by;   // the file the entry points came from (zk_pool_rccl_library)
    // Which librccl: (1) ZKATTEST_RCCL_LIB, an explicit path (the GPU tier's stub, tests/rccl_stub; a deployment's own build); (2) a librccl this process
    // has ALREADY mapped (RTLD_NOLOAD) -- under PyTorch that is the wheel's bundled one, bound to the wheel's bundled HIP runtime, which is then also the
    // runtime this library runs on: a second, system librccl next to it would talk to the other libamdhip64 (the mix DESIGN.md section 9 shows to
    // misbehave); (3) the system one.
    bool load() {
        if (h) return true;
        if (getenv("ZKATTEST_NO_RCCL")) return false;
        static const char* const names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
        if (const char* e = getenv("ZKATTEST_RCCL_LIB")) {
            h = dlopen(e, RTLD_NOW | RTLD_LOCAL);
            if (!h) return false;   // an explicit choice that cannot be loaded is not silently replaced
        }
        for (int pass = 0; pass < 2 && !h; pass++)
            for (const char* name : names) {
                h = dlopen(name, RTLD_NOW | RTLD_LOCAL | (pass == 0 ? RTLD_NOLOAD : 0));
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
        Dl_info di;
        served_by = dladdr((void*)CommInitAll, &di) && di.dli_fname ? di.dli_fname : "?";
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
    // host side of a shard: the CPUs next to the device's PCIe root (sysfs local_cpulist of its bus address).  The shard's host
    // thread runs there, and zk_pool_host_alloc has the shard's output region first-touched there, so that a device's 40 GB/s
    // of proof bytes land in the memory of its own socket instead of crossing the inter-socket link.
    std::vector<std::vector<int>> cpus;   // empty = unknown (no affinity is set)
    std::vector<int> numa;                // -1 = unknown
    bool affinity = true;                 // ZKATTEST_POOL_AFFINITY=0 switches it off
    std::vector<float> shard_ms;          // wall time of every shard's part of the last pool call (zk_pool_shard_ms)
    int test_fail_slot = -1;              // test build only (zk_test_pool_fail_next_submit): the next streamed submit fails at this device slot (one shot)
};

// "0-15,128-143" -> cpu numbers (the format of sysfs cpulist files)
static std::vector<int> parse_cpulist(const char* s) {
    std::vector<int> v;
    while (*s) {
        char* e;
        long a = strtol(s, &e, 10);
        if (e == s) break;
        long b = a;
        if (*e == '-') {
            const char* q = e + 1;
            b = strtol(q, &e, 10);
            if (e == q) break;
        }
        for (long c = a; c <= b && c < 4096 && v.size() < 4096; c++) v.push_back((int)c);
        s = *e == ',' ? e + 1 : e;
        if (*e != ',') break;
    }
    return v;
}
static bool read_small_file(const std::string& path, char* buf, size_t cap) {
    FILE* f = fopen(path.c_str(), "r");
    if (!f) return false;
    size_t n = fread(buf, 1, cap - 1, f);
    fclose(f);
    buf[n] = 0;
    return n > 0;
}
// PCI bus address of a device -> its NUMA node and local CPUs (ZKATTEST_SYSFS_ROOT lets the CPU tests point this at a fake tree)
static void locality_of_bus(const char* busid, int* numa, std::vector<int>* cpus) {
    *numa = -1;
    cpus->clear();
    const char* root = getenv("ZKATTEST_SYSFS_ROOT");
    std::string dir = std::string(root ? root : "/sys") + "/bus/pci/devices/";
    std::string id(busid);
    for (auto& ch : id) ch = (char)tolower(ch);
    char buf[4096];
    if (read_small_file(dir + id + "/numa_node", buf, sizeof buf)) *numa = atoi(buf);
    if (read_small_file(dir + id + "/local_cpulist", buf, sizeof buf)) *cpus = parse_cpulist(buf);
}
// test hook (tests/test_abi_and_host.py): the parsing above without a GPU.  Returns the number of CPUs, fills up to cap of them.
extern "C" int zk_pool_test_locality(const char* busid, int* numa, int* cpus, int cap) {
    std::vector<int> v;
    int nn = -1;
    locality_of_bus(busid, &nn, &v);
    if (numa) *numa = nn;
    for (int i = 0; i < (int)v.size() && i < cap; i++) cpus[i] = v[i];
    return (int)v.size();
}
static void bind_thread_to(const std::vector<int>& cpus) {
    if (cpus.empty()) return;
    cpu_set_t set;
    CPU_ZERO(&set);
    for (int c : cpus)
        if (c >= 0 && c < CPU_SETSIZE) CPU_SET(c, &set);
    (void)sched_setaffinity(0, sizeof set, &set);   // best effort: a cgroup that does not own these CPUs refuses, and that is fine
}

// runs f(i) for every device on its own host thread (bound to the device's local CPUs when they are known) and returns the
// first non-zero status
template <class F>
static zk_status pool_each(zk_pool* p, F f) {
    const int G = (int)p->ctx.size();
    std::vector<zk_status> st(G, ZK_OK);
    p->shard_ms.assign(G, 0.f);
    auto timed = [&](int i) {
        auto t0 = std::chrono::steady_clock::now();
        st[i] = f(i);
        p->shard_ms[i] = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    };
    if (G == 1) {
        timed(0);
    } else {
        std::vector<std::thread> th;   // a shard that gets no thread (std::system_error: thread limits) runs on the caller's: see run_threads, api_json.hip
        th.reserve(G);
        int started = 0;
        try {
            for (; started < G; started++)
                th.emplace_back([&, i = started] {
                    if (p->affinity && i < (int)p->cpus.size()) bind_thread_to(p->cpus[i]);
                    timed(i);
                });
        } catch (...) {
        }
        for (int i = started; i < G; i++) timed(i);
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

static thread_local std::string g_pool_create_err;   // why the last zk_pool_create of this thread failed (zk_pool_last_error(NULL))
extern "C" zk_status zk_pool_create(const int* device_ids, int n_dev, zk_pool** out) {
    if (!out) return ZK_E_ARG;
    *out = nullptr;   // a caller never sees a half-built pool: either every context exists or *out stays NULL
    if (!device_ids || n_dev < 1 || n_dev > 64) return ZK_E_ARG;
    zk_pool* p = new zk_pool();
    p->ctx.assign(n_dev, nullptr);
    p->dev.assign(device_ids, device_ids + n_dev);
    p->cpus.assign(n_dev, {});
    p->numa.assign(n_dev, -1);
    if (const char* e = getenv("ZKATTEST_POOL_AFFINITY")) p->affinity = atoi(e) != 0;
    for (int i = 0; i < n_dev; i++) {
        char bus[64] = {0};
        if (hipDeviceGetPCIBusId(bus, sizeof bus, p->dev[i]) == hipSuccess) locality_of_bus(bus, &p->numa[i], &p->cpus[i]);
        else (void)hipGetLastError();
    }
    std::vector<std::string> why(n_dev);   // zk_last_error(NULL) is per thread: read it on the thread that created the context
    zk_status zs = pool_each(p, [&](int i) {   // the generator tables are built concurrently
        zk_status s = zk_ctx_create(p->dev[i], &p->ctx[i]);
        if (s) why[i] = zk_last_error(nullptr);
        return s;
    });
    if (zs) {
        g_pool_create_err = p->err;
        for (int i = 0; i < n_dev; i++)
            if (!why[i].empty()) {
                g_pool_create_err += why[i];
                break;
            }
        zk_pool_destroy(p);
        return zs;
    }
    *out = p;
    return ZK_OK;
}
extern "C" void zk_pool_destroy(zk_pool* p) {
    if (!p) return;
    for (size_t i = 0; i < p->comms.size(); i++)
        if (p->comms[i]) p->rccl.CommDestroy(p->comms[i]);
    for (auto c : p->ctx) zk_ctx_destroy(c);
    delete p;
}
// Fault injection for tests/test_gpu_stream.py (the abandon path with older pool jobs in flight), compiled ONLY into the test build (`make testhooks`:
// lib/libzkattest_hip_testhooks.so, -DZK_TEST_HOOKS): the next zk_pool_prove_submit / zk_pool_verify_submit fails at device slot `slot` with ZK_E_DEVICE after the
// earlier slots were submitted (one shot).  The product library has neither the entry point nor the branch: nothing can make a production submit fail.
#ifdef ZK_TEST_HOOKS
extern "C" void zk_test_pool_fail_next_submit(zk_pool* p, int slot) {
    if (p) p->test_fail_slot = slot;
}
#define POOL_INJECTED(p, i) ((int)(i) == (p)->test_fail_slot)
#else
#define POOL_INJECTED(p, i) false
#endif
extern "C" int zk_pool_size(const zk_pool* p) { return p ? (int)p->ctx.size() : 0; }
extern "C" zk_ctx* zk_pool_ctx(zk_pool* p, int i) { return p && i >= 0 && i < (int)p->ctx.size() ? p->ctx[i] : nullptr; }
extern "C" const char* zk_pool_last_error(const zk_pool* p) { return p ? p->err.c_str() : g_pool_create_err.c_str(); }
extern "C" int zk_pool_shard_ms(const zk_pool* p, float* ms, int cap) {
    if (!p) return 0;
    for (int i = 0; i < (int)p->shard_ms.size() && i < cap; i++) ms[i] = p->shard_ms[i];
    return (int)p->shard_ms.size();
}
extern "C" int zk_pool_numa_node(const zk_pool* p, int i) { return p && i >= 0 && i < (int)p->numa.size() ? p->numa[i] : -1; }

// ---- page-locked output buffer of a pool call with per-shard placement.  zk_pool_prove_batch gives shard i the region
// [i * R, (i + 1) * R), R = (bytes / G) & ~255: the pages of region i are first touched by a thread bound to the CPUs of device
// i's socket (the kernel's default policy then places them on that node) and the whole range is page-locked afterwards, so
// every device's DMA writes stay on its own side of the machine.  Falls back to plain zk_host_alloc behaviour when the
// locality is unknown.  Free with zk_pool_host_free (NOT zk_host_free).
static std::mutex g_regs_mu;
static std::map<void*, size_t> g_regs;
// (Other strategies were measured and dropped -- plain hipHostMalloc, hipHostMallocNumaUser under a preferred-node policy, no huge pages: profiles/r04_first_call_anomaly.txt.)
static void* pool_host_alloc_once(zk_pool* p, size_t bytes);
extern "C" void zk_pool_host_free(void* mem);
// the fastest of up to three candidates: see "slow pages" in api.hip
extern "C" void* zk_pool_host_alloc(zk_pool* p, size_t bytes) {
    if (!p || !bytes) return nullptr;
    int cur = -1;   // the candidates are timed from device 0; the caller's current device is put back
    if (hipGetDevice(&cur) != hipSuccess) cur = -1, (void)hipGetLastError();
    (void)hipSetDevice(p->dev[0]);
    void* m = alloc_fast_pinned(bytes, [&]() { return pool_host_alloc_once(p, bytes); }, [](void* q) { zk_pool_host_free(q); });
    if (cur >= 0) (void)hipSetDevice(cur);
    return m;
}
static void* pool_host_alloc_once(zk_pool* p, size_t bytes) {
    const size_t page = (size_t)sysconf(_SC_PAGESIZE);
    const size_t len = (bytes + page - 1) / page * page;
    void* mem = mmap(nullptr, len, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (mem == MAP_FAILED) return nullptr;
    (void)madvise(mem, len, MADV_HUGEPAGE);
    const size_t G = p->ctx.size(), region = (bytes / G) & ~(size_t)255;
    // placement: an explicit memory policy per shard region (mbind, MPOL_PREFERRED: the device's node if it has room) -- first touch
    // alone depends on where the touching thread happens to run when the cpuset does not grant the device's CPUs -- and the touch
    // from a thread bound to those CPUs on top of it.  Measured on a two-socket box: D2H into a buffer on the far node runs at 30
    // instead of 57 GB/s.
    auto prefer_node = [&](size_t a, size_t b, int node) {
        if (node < 0 || node >= 1024) return;
        unsigned long mask[16] = {0};
        mask[node / (8 * sizeof(unsigned long))] |= 1ul << (node % (8 * sizeof(unsigned long)));
        const size_t a0 = a / page * page;
        (void)syscall(SYS_mbind, (char*)mem + a0, b - a0, 1 /* MPOL_PREFERRED */, mask, 1024ul, 0u);
    };
    for (size_t i = 0; i < G; i++) prefer_node(i * region, i + 1 == G ? len : (i + 1) * region, p->numa[i]);
    auto touch = [&](size_t a, size_t b) {   // one write per page of [a, b)
        volatile uint8_t* q = (volatile uint8_t*)mem;
        for (size_t o = a / page * page; o < b; o += page) q[o] = 0;
    };
    if (!p->affinity) {
        touch(0, len);
    } else {   // also for a single device: the pages belong on ITS node, wherever the calling thread happens to run
        std::vector<std::thread> th;
        th.reserve(G);
        size_t started = 0;
        try {
            for (; started < G; started++)
                th.emplace_back([&, i = started] {
                    bind_thread_to(p->cpus[i]);
                    touch(i * region, i + 1 == G ? len : (i + 1) * region);
                });
        } catch (...) {
        }
        for (size_t i = started; i < G; i++) touch(i * region, i + 1 == G ? len : (i + 1) * region);   // placement by policy only
        for (auto& t : th) t.join();
    }
    if (hipHostRegister(mem, len, hipHostRegisterPortable) != hipSuccess) {
        (void)hipGetLastError();
        munmap(mem, len);
        return nullptr;
    }
    std::lock_guard<std::mutex> g(g_regs_mu);
    g_regs[mem] = len;
    return mem;
}
extern "C" void zk_pool_host_free(void* mem) {
    if (!mem) return;
    size_t len = 0;
    {
        std::lock_guard<std::mutex> g(g_regs_mu);
        auto it = g_regs.find(mem);
        if (it == g_regs.end()) return;
        len = it->second;
        g_regs.erase(it);
    }
    if (!len) {   // came from hipHostMalloc (ZKATTEST_POOL_ALLOC=hostmalloc / numauser)
        (void)hipHostFree(mem);
        return;
    }
    (void)hipHostUnregister(mem);
    munmap(mem, len);
}
extern "C" const char* zk_pool_ring_transport(const zk_pool* p) { return p ? p->transport : ""; }
extern "C" const char* zk_pool_rccl_library(const zk_pool* p) { return p ? p->rccl.served_by.c_str() : ""; }

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
    int dev_at_entry = -1;   // the calling thread's current device is restored on every exit path
    if (hipGetDevice(&dev_at_entry) != hipSuccess) dev_at_entry = -1, (void)hipGetLastError();
    auto release = [&] {
        for (int i = 0; i < G; i++)
            if (d[i]) {
                hipSetDevice(p->dev[i]);
                hipFree(d[i]);
            }
        if (dev_at_entry >= 0) hipSetDevice(dev_at_entry);
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
        // RCCL refuses two ranks on one device; tests/rccl_stub does not, and ZKATTEST_RCCL_SAME_DEVICE=1 lets the one-GPU tier drive this branch through it
        if (!distinct && getenv("ZKATTEST_RCCL_SAME_DEVICE") && getenv("ZKATTEST_RCCL_LIB")) distinct = true;
        if (distinct && !p->comms_tried) {   // one communicator per device, created once per pool
            p->comms_tried = true;
            if (p->rccl.load()) {
                p->comms.assign(G, nullptr);
                ncclResult_t r = p->rccl.CommInitAll(p->comms.data(), G, p->dev.data());
                if (r != 0) {
                    p->comms.clear();
                    p->err = std::string("rccl: ncclCommInitAll failed (") + (p->rccl.GetErrorString ? p->rccl.GetErrorString(r) : "?") + "); the ring travels by peer copies";
                }
            } else {
                p->err = getenv("ZKATTEST_NO_RCCL") ? "rccl: switched off by ZKATTEST_NO_RCCL; the ring travels by peer copies"
                                                   : "rccl: librccl.so not found or incomplete; the ring travels by peer copies";
            }
        }
        bool done = false;
        if (distinct && !p->comms.empty()) {
            ncclResult_t r = p->rccl.GroupStart();
            for (int i = 0; i < G && r == 0; i++) {
                hipSetDevice(p->dev[i]);
                r = p->rccl.Broadcast(d[i], d[i], bytes, kNcclUint8, 0, p->comms[i], p->ctx[i]->stream);   // in place; only the root's buffer is read
            }
            ncclResult_t r2 = p->rccl.GroupEnd();   // closes the group also when a Broadcast was refused (its status is then an error too)
            if (r == 0 && r2 == 0) {
                done = true;
                for (int i = 0; i < G; i++) {
                    hipSetDevice(p->dev[i]);
                    if (hipStreamSynchronize(p->ctx[i]->stream) != hipSuccess) done = false;
                }
            }
            if (done) p->transport = "rccl";
            else {   // communicators that failed once are not trusted again: this and every later ring go by peer copies
                const ncclResult_t bad = r ? r : r2;   // zk_pool_last_error says why, although the call itself succeeds through the fallback
                p->err = std::string("rccl: ncclBroadcast of the ring failed (") + (bad && p->rccl.GetErrorString ? p->rccl.GetErrorString(bad) : "stream synchronisation") +
                         "); the ring travels by peer copies from now on";
                (void)hipGetLastError();
                for (auto cm : p->comms)
                    if (cm) p->rccl.CommDestroy(cm);
                p->comms.clear();
            }
        }
        if (!distinct) p->err = "rccl: a device is listed twice in this pool; the ring travels by peer copies";
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

// The same with the proofs left in HBM: shard i writes into d_out[i], a buffer of out_cap[i] bytes on device_ids[i] (zk_pool_device_alloc).
// Nothing but 160 bytes of inputs per proof crosses the link, so a scaling run through this entry point measures the GPUs and their host
// threads, not the node's host memory (8 shards emit ~55 GB/s of page-locked writes each through zk_pool_prove_batch).
extern "C" zk_status zk_pool_prove_batch_device(zk_pool* p, uint64_t B, const uint8_t* msg, const uint8_t* sig, const uint8_t* pk, const uint32_t* which,
                                                const zk_rng* rng, void* const* d_out, const uint64_t* out_cap, uint64_t* out_off, uint64_t* out_len, int32_t* status) {
    if (!p || !rng || !d_out || !out_cap || !out_off || !out_len || !status || (B && (!msg || !sig || !pk || !which || !rng->data))) return ZK_E_ARG;
    return pool_each(p, [&](int i) -> zk_status {
        uint64_t first, cnt;
        zk_pool_shard(p, B, i, &first, &cnt);
        if (!cnt) return ZK_OK;
        if (!d_out[i]) return ZK_E_ARG;
        zk_rng r = *rng;
        r.data = rng->data + (rng->mode == ZK_RNG_SEED ? 32 * first : 32 * first * rng->stride_blocks);
        std::vector<uint64_t> off(cnt + 1);
        zk_status zs = zk_prove_batch(p->ctx[i], cnt, msg + 32 * first, sig + 64 * first, pk + 64 * first, which + first, &r, (uint8_t*)d_out[i], out_cap[i], off.data(),
                                      status + first);
        if (zs) return zs;
        for (uint64_t j = 0; j < cnt; j++) out_off[first + j] = off[j], out_len[first + j] = off[j + 1] - off[j];   // relative to d_out[i]
        return ZK_OK;
    });
}
extern "C" void* zk_pool_device_alloc(zk_pool* p, int i, size_t bytes) {
    if (!p || i < 0 || i >= (int)p->ctx.size()) return nullptr;
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess) cur = -1, (void)hipGetLastError();
    void* mem = nullptr;
    if (hipSetDevice(p->dev[i]) != hipSuccess || hipMalloc(&mem, bytes ? bytes : 1) != hipSuccess) (void)hipGetLastError(), mem = nullptr;
    if (cur >= 0) (void)hipSetDevice(cur);
    return mem;
}
extern "C" void zk_pool_device_free(zk_pool* p, int i, void* mem) {
    if (!p || !mem || i < 0 || i >= (int)p->ctx.size()) return;
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess) cur = -1, (void)hipGetLastError();
    if (hipSetDevice(p->dev[i]) == hipSuccess) (void)hipFree(mem);
    if (cur >= 0) (void)hipSetDevice(cur);
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

// ---- streamed pool calls: zk_prove_submit / zk_prove_wait (api_stream.hip) on every device's shard, so that a node keeps two or three
// batches in flight per GPU.  A pool job is the set of its shards' jobs; submit runs on the caller's thread (staging the inputs is a
// memcpy and an asynchronous upload per device), wait drives every device's queue on that device's own host thread.
struct zk_pool_job {
    int kind = 0;                         // 0 prove, 1 verify
    uint64_t B = 0;
    std::vector<zk_job*> shard;           // per device (nullptr: empty shard)
    std::vector<std::vector<uint64_t>> off;   // per device: the shard's own offsets (prove: filled by wait; verify: input)
    uint64_t region = 0;
    uint64_t *out_off = nullptr, *out_len = nullptr;   // prove: where the caller wants (offset, length) per proof
};
zk_status stream_cancel_job(zk_ctx* c, zk_job* j);   // api_stream.hip
// a submit that failed half way: the shards already queued are taken out again.  Earlier pool jobs may still be in flight on those
// contexts, so this is NOT a wait (waits go in submission order; see stream_cancel_job).
static void pool_job_abandon(zk_pool* p, zk_pool_job* j) {
    for (size_t i = 0; i < j->shard.size(); i++)
        if (j->shard[i]) (void)stream_cancel_job(p->ctx[i], j->shard[i]);
    delete j;
}
extern "C" zk_status zk_pool_prove_submit(zk_pool* p, uint64_t B, const uint8_t* msg, const uint8_t* sig, const uint8_t* pk, const uint32_t* which, const zk_rng* rng,
                                          uint8_t* out, uint64_t out_cap, uint64_t* out_off, uint64_t* out_len, int32_t* status, zk_pool_job** job) {
    if (!p || !job || !rng || !out_off || !out_len || !status || !B || !msg || !sig || !pk || !which || !rng->data || !out) return ZK_E_ARG;
    *job = nullptr;
    const uint64_t G = p->ctx.size();
    zk_pool_job* j = new zk_pool_job();
    j->B = B, j->shard.assign(G, nullptr), j->off.resize(G), j->region = (out_cap / G) & ~(uint64_t)255, j->out_off = out_off, j->out_len = out_len;
    for (uint64_t i = 0; i < G; i++) {
        uint64_t first, cnt;
        zk_pool_shard(p, B, (int)i, &first, &cnt);
        if (!cnt) continue;
        zk_rng r = *rng;
        r.data = rng->data + (rng->mode == ZK_RNG_SEED ? 32 * first : 32 * first * rng->stride_blocks);
        j->off[i].assign(cnt + 1, 0);
        zk_status zs = POOL_INJECTED(p, i) ? (zk_status)ZK_E_DEVICE
                                                   : zk_prove_submit(p->ctx[i], cnt, msg + 32 * first, sig + 64 * first, pk + 64 * first, which + first, &r,
                                                                     out + j->region * i, j->region, j->off[i].data(), status + first, &j->shard[i]);
        if (zs) {
            if (POOL_INJECTED(p, i)) p->test_fail_slot = -1, p->ctx[i]->err = "(injected by zk_test_pool_fail_next_submit)";
            p->err = std::string("device slot ") + std::to_string(i) + ": " + zk_strerror(zs) + " " + zk_last_error(p->ctx[i]);
            pool_job_abandon(p, j);
            return zs;
        }
    }
    *job = j;
    return ZK_OK;
}
extern "C" zk_status zk_pool_prove_wait(zk_pool* p, zk_pool_job* j) {
    if (!p || !j || j->kind != 0 || j->shard.size() != p->ctx.size()) return ZK_E_ARG;
    zk_status zs = pool_each(p, [&](int i) -> zk_status {
        if (!j->shard[i]) return ZK_OK;
        zk_status s = zk_prove_wait(p->ctx[i], j->shard[i]);
        j->shard[i] = nullptr;
        if (s) return s;
        uint64_t first, cnt;
        zk_pool_shard(p, j->B, i, &first, &cnt);
        for (uint64_t k = 0; k < cnt; k++) j->out_off[first + k] = j->region * i + j->off[i][k], j->out_len[first + k] = j->off[i][k + 1] - j->off[i][k];
        return ZK_OK;
    });
    delete j;
    return zs;
}
extern "C" zk_status zk_pool_verify_submit(zk_pool* p, uint64_t B, const uint8_t* msg, const uint8_t* proofs, const uint64_t* proof_off, const uint64_t* proof_len,
                                           const uint8_t* vseeds, uint8_t* ok, int32_t* status, zk_pool_job** job) {
    if (!p || !job || !B || !msg || !proofs || !proof_off || !proof_len || !ok || !status) return ZK_E_ARG;
    *job = nullptr;
    const uint64_t G = p->ctx.size();
    zk_pool_job* j = new zk_pool_job();
    j->kind = 1, j->B = B, j->shard.assign(G, nullptr), j->off.resize(G);
    for (uint64_t i = 0; i < G; i++) {
        uint64_t first, cnt;
        zk_pool_shard(p, B, (int)i, &first, &cnt);
        if (!cnt) continue;
        std::vector<uint64_t>& off = j->off[i];   // lives until the job is waited for: the shard's job reads it
        off.assign(cnt + 1, 0);
        const uint64_t base = proof_off[first];
        zk_status zs = base & 3 ? ZK_E_ARG : ZK_OK;
        for (uint64_t k = 0; k < cnt && !zs; k++) {
            if (proof_off[first + k] - base != off[k]) zs = ZK_E_ARG;   // a gap or an overlap inside the shard
            off[k + 1] = off[k] + proof_len[first + k];
        }
        if (!zs && POOL_INJECTED(p, i)) zs = ZK_E_DEVICE, p->test_fail_slot = -1, p->ctx[i]->err = "(injected by zk_test_pool_fail_next_submit)";
        else if (!zs) zs = zk_verify_submit(p->ctx[i], cnt, msg + 32 * first, proofs + base, off.data(), vseeds ? vseeds + 32 * first : nullptr, ok + first, status + first, &j->shard[i]);
        if (zs) {
            p->err = std::string("device slot ") + std::to_string(i) + ": " + zk_strerror(zs) + " " + zk_last_error(p->ctx[i]);
            pool_job_abandon(p, j);
            return zs;
        }
    }
    *job = j;
    return ZK_OK;
}
extern "C" zk_status zk_pool_verify_wait(zk_pool* p, zk_pool_job* j) {
    if (!p || !j || j->kind != 1 || j->shard.size() != p->ctx.size()) return ZK_E_ARG;
    zk_status zs = pool_each(p, [&](int i) -> zk_status {
        if (!j->shard[i]) return ZK_OK;
        zk_status s = zk_verify_wait(p->ctx[i], j->shard[i]);
        j->shard[i] = nullptr;
        return s;
    });
    delete j;
    return zs;
}
