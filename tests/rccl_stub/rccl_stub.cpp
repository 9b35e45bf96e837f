// TEST INFRASTRUCTURE (not product code): a stand-in for librccl that exports the six nccl* entry points zk_pool_set_ring binds
// (zkp-ecdsa_amd/csrc/api_pool.hip: ncclCommInitAll, ncclCommDestroy, ncclGroupStart, ncclGroupEnd, ncclBroadcast, ncclGetErrorString) and implements the
// grouped in-place broadcast with hipMemcpyPeerAsync, so that the pool's RCCL branch -- success, each failure path, communicator teardown, the "not trusted
// again" state -- executes on a ONE-GPU box, where the real RCCL refuses two ranks on one device.  Selected with ZKATTEST_RCCL_LIB=<this .so> and
// ZKATTEST_RCCL_SAME_DEVICE=1.  Failure injection: RCCL_STUB_FAIL = init | broadcast | groupend (read at every call).  Counters: rccl_stub_counter(i).
// Built by tests/test_gpu_scale.py with g++ against the HIP runtime headers.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <cstring>
#include <vector>

struct StubComm {
    int rank, nranks, dev;
};
struct Pending {
    const void* send;
    void* recv;
    size_t bytes;
    int root;
    StubComm* comm;
    hipStream_t stream;
};
static thread_local std::vector<Pending> g_group;
static thread_local int g_depth = 0;
static int g_counters[8];   // 0 CommInitAll calls, 1 communicators created, 2 destroyed, 3 Broadcast calls, 4 groups completed, 5 bytes moved (KiB)

static bool fail_at(const char* what) {
    const char* e = getenv("RCCL_STUB_FAIL");
    return e && !strcmp(e, what);
}
extern "C" {
typedef StubComm* ncclComm_t;
int ncclCommInitAll(ncclComm_t* comms, int n, const int* devs) {
    g_counters[0]++;
    if (fail_at("init")) return 2;   // ncclSystemError
    for (int i = 0; i < n; i++) comms[i] = new StubComm{i, n, devs ? devs[i] : i}, g_counters[1]++;
    return 0;
}
int ncclCommDestroy(ncclComm_t c) {
    if (c) g_counters[2]++;
    delete c;
    return 0;
}
int ncclGroupStart() {
    g_depth++;
    return 0;
}
static int run_group() {
    int rc = 0;
    for (const Pending& p : g_group) {
        if (p.comm->rank == p.root) continue;
        const Pending* root = nullptr;
        for (const Pending& q : g_group)
            if (q.comm->rank == p.root) root = &q;
        if (!root) {
            rc = 5;   // ncclInvalidUsage: the root did not take part
            break;
        }
        (void)hipSetDevice(p.comm->dev);
        if (hipMemcpyPeerAsync(p.recv, p.comm->dev, root->send, root->comm->dev, p.bytes, p.stream) != hipSuccess) {
            rc = 1;   // ncclUnhandledCudaError
            break;
        }
        g_counters[5] += (int)(p.bytes >> 10);
    }
    g_group.clear();
    if (!rc) g_counters[4]++;
    return rc;
}
int ncclGroupEnd() {
    if (g_depth > 0) g_depth--;
    if (g_depth) return 0;
    if (fail_at("groupend")) {
        g_group.clear();
        return 3;   // ncclInternalError
    }
    return run_group();
}
int ncclBroadcast(const void* send, void* recv, size_t count, int dtype, int root, ncclComm_t comm, hipStream_t stream) {
    g_counters[3]++;
    if (fail_at("broadcast")) return 4;   // ncclInvalidArgument
    if (!comm || dtype != 1) return 4;
    g_group.push_back(Pending{send, recv, count, root, comm, stream});
    if (!g_depth) return run_group();
    return 0;
}
const char* ncclGetErrorString(int r) {
    switch (r) {
        case 0: return "no error";
        case 1: return "stub: unhandled hip error";
        case 2: return "stub: unhandled system error";
        case 3: return "stub: internal error";
        case 4: return "stub: invalid argument";
        case 5: return "stub: invalid usage";
    }
    return "stub: unknown";
}
int rccl_stub_counter(int i) { return i >= 0 && i < 8 ? g_counters[i] : -1; }
void rccl_stub_reset() { memset(g_counters, 0, sizeof g_counters); }
}
