// tests/hostsim/fake_rccl.cpp -- TEST INFRASTRUCTURE: the six RCCL entry points og_cluster.inl binds with dlopen, for the
// host simulator, where every "device" is host memory of one process.  Built as _build/fake_rccl/librccl.so.1; the test
// process finds it before the real library through LD_LIBRARY_PATH.  ncclReduce inside a group is recorded and executed at
// ncclGroupEnd: root's recvbuf = sum over the ranks' sendbufs in rank order (float32 sum is all the cluster uses).
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#include <mutex>
#include <vector>

extern "C" {
typedef enum { ncclSuccess = 0, ncclInvalidArgument = 4 } ncclResult_t;
struct ncclComm {
    int rank, nranks;
    uint64_t clique;
};
typedef ncclComm* ncclComm_t;

struct Pending {
    ncclComm_t comm;
    const float* send;
    float* recv;
    size_t count;
    int root;
};
static std::mutex g_lock;
static std::vector<Pending> g_pending;
static int g_group_depth = 0;
static uint64_t g_next_clique = 1;

static void run_pending()
{
    // group the recorded calls by clique
    while (!g_pending.empty()) {
        const uint64_t cl = g_pending[0].comm->clique;
        std::vector<Pending> mine, rest;
        for (const Pending& p : g_pending) (p.comm->clique == cl ? mine : rest).push_back(p);
        g_pending.swap(rest);
        const size_t n = mine[0].count;
        std::vector<float> acc(n, 0.0f);
        bool first = true;
        for (int r = 0; r < mine[0].comm->nranks; ++r)
            for (const Pending& p : mine)
                if (p.comm->rank == r) {
                    if (first) memcpy(acc.data(), p.send, n * sizeof(float));
                    else
                        for (size_t i = 0; i < n; ++i) acc[i] += p.send[i];
                    first = false;
                }
        for (const Pending& p : mine)
            if (p.comm->rank == p.root) memcpy(p.recv, acc.data(), n * sizeof(float));
    }
}

ncclResult_t ncclCommInitAll(ncclComm_t* comms, int n, const int*)
{
    std::lock_guard<std::mutex> lk(g_lock);
    const uint64_t cl = g_next_clique++;
    for (int i = 0; i < n; ++i) comms[i] = new ncclComm{i, n, cl};
    return ncclSuccess;
}
ncclResult_t ncclCommDestroy(ncclComm_t c)
{
    delete c;
    return ncclSuccess;
}
ncclResult_t ncclGroupStart()
{
    std::lock_guard<std::mutex> lk(g_lock);
    ++g_group_depth;
    return ncclSuccess;
}
ncclResult_t ncclGroupEnd()
{
    std::lock_guard<std::mutex> lk(g_lock);
    if (--g_group_depth == 0) run_pending();
    return ncclSuccess;
}
ncclResult_t ncclReduce(const void* send, void* recv, size_t count, int dtype, int op, int root, ncclComm_t comm, void*)
{
    if (dtype != 7 || op != 0 || !comm) return ncclInvalidArgument; // ncclFloat32, ncclSum
    std::lock_guard<std::mutex> lk(g_lock);
    g_pending.push_back(Pending{comm, (const float*)send, (float*)recv, count, root});
    if (g_group_depth == 0) run_pending();
    return ncclSuccess;
}
const char* ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : "fake rccl: invalid argument"; }
}
