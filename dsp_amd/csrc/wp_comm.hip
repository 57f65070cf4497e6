// wp_comm.hip — the path's one exchange step behind the C ABI: the rank-local
// partial master buses of a track-sharded render are summed with an RCCL
// reduce over xGMI (one process per GPU).  RCCL is not linked: it is opened at
// the first wp_comm_* call from the directory of the HIP runtime the process
// already uses, so single-GPU hosts (a Node process running one reverb) never
// load it.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <mutex>
#include <string>

#include "../../include/wpdsp.h"

extern "C" __attribute__((visibility("hidden"))) int wp_fail_message(int code, const char *msg);  // wp_api.hip: sets wp_last_error() for this thread

static int cfail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    return wp_fail_message(code, buf);
}

static_assert(WP_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size");

namespace {
struct Rccl {
    void *so = nullptr;
    std::string path;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Reduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
std::mutex g_rccl_mu;
Rccl g_rccl;
bool g_rccl_tried = false;
std::string g_rccl_err;

// librccl beside the HIP runtime this process runs on (a Python process that imported
// torch uses the runtime and the RCCL build torch ships; anything else the system ROCm's)
int load_rccl()
{
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    if (g_rccl.so) return WP_OK;
    if (g_rccl_tried) return cfail(WP_ERR_UNSUPPORTED, "%s", g_rccl_err.c_str());
    g_rccl_tried = true;
    std::string dir;
    Dl_info info;
    if (dladdr(reinterpret_cast<void *>(&hipGetDeviceCount), &info) && info.dli_fname) {
        dir = info.dli_fname;
        const size_t slash = dir.rfind('/');
        dir = slash == std::string::npos ? std::string() : dir.substr(0, slash + 1);
    }
    const std::string candidates[] = {dir + "librccl.so.1", dir + "librccl.so", "librccl.so.1", "librccl.so"};
    std::string tried;
    for (const std::string &c : candidates) {
        if (c.empty()) continue;
        void *so = dlopen(c.c_str(), RTLD_NOW | RTLD_LOCAL);
        if (!so) {
            tried += c + " (" + dlerror() + "); ";
            continue;
        }
        Rccl r;
        r.so = so;
        r.path = c;
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(so, "ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(so, "ncclCommInitRank"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(so, "ncclCommDestroy"));
        r.Reduce = reinterpret_cast<decltype(r.Reduce)>(dlsym(so, "ncclReduce"));
        r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(dlsym(so, "ncclAllReduce"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(so, "ncclGetErrorString"));
        if (r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.Reduce && r.AllReduce && r.GetErrorString) {
            g_rccl = r;
            return WP_OK;
        }
        tried += c + " (lacks the nccl entry points); ";
        dlclose(so);
    }
    g_rccl_err = "RCCL could not be loaded: " + tried;
    return cfail(WP_ERR_UNSUPPORTED, "%s", g_rccl_err.c_str());
}
}  // namespace

struct wp_comm {
    ncclComm_t comm;
    int nranks, rank, dev;
};

#define RCCLCHK(expr, what)                                                                              \
    do {                                                                                                 \
        ncclResult_t r__ = (expr);                                                                       \
        if (r__ != ncclSuccess) return cfail(WP_ERR_HIP, "%s failed: %s", what, g_rccl.GetErrorString(r__)); \
    } while (0)

extern "C" int wp_comm_unique_id(unsigned char *id_out)
{
    const int n = wp_init();
    if (n < 0) return n;
    if (!id_out) return cfail(WP_ERR_INVALID, "wp_comm_unique_id: null id");
    if (int rc = load_rccl()) return rc;
    ncclUniqueId id;
    RCCLCHK(g_rccl.GetUniqueId(&id), "ncclGetUniqueId");
    memcpy(id_out, id.internal, WP_COMM_ID_BYTES);
    return WP_OK;
}

extern "C" int wp_comm_available(void)
{
    const int n = wp_init();
    if (n < 0) return n;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return cfail(WP_ERR_HIP, "hipGetDevice failed");
    return load_rccl();
}

extern "C" wp_comm *wp_comm_create(const unsigned char *id_bytes, int nranks, int rank)
{
    if (wp_init() < 0) return nullptr;
    if (!id_bytes || nranks < 1 || rank < 0 || rank >= nranks) {
        cfail(WP_ERR_INVALID, "wp_comm_create: bad argument (rank %d of %d)", rank, nranks);
        return nullptr;
    }
    if (load_rccl()) return nullptr;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) {
        cfail(WP_ERR_HIP, "hipGetDevice failed");
        return nullptr;
    }
    ncclUniqueId id;
    memcpy(id.internal, id_bytes, WP_COMM_ID_BYTES);
    ncclComm_t comm = nullptr;
    const ncclResult_t r = g_rccl.CommInitRank(&comm, nranks, id, rank);
    if (r != ncclSuccess) {
        cfail(WP_ERR_HIP, "ncclCommInitRank(rank %d of %d, device %d) failed: %s", rank, nranks, dev, g_rccl.GetErrorString(r));
        return nullptr;
    }
    wp_comm *c = new wp_comm();
    c->comm = comm;
    c->nranks = nranks;
    c->rank = rank;
    c->dev = dev;
    return c;
}

extern "C" void wp_comm_destroy(wp_comm *c)
{
    if (!c) return;
    if (c->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->comm);
    delete c;
}

extern "C" int64_t wp_comm_info(const wp_comm *c, int which)
{
    if (!c) return WP_ERR_INVALID;
    switch (which) {
    case 0: return c->nranks;
    case 1: return c->rank;
    case 2: return c->dev;
    }
    return WP_ERR_INVALID;
}

static int check_reduce_args(const wp_comm *c, const float *send_dev, float *recv_dev, int64_t count, int root, bool need_recv,
                             const char *who)
{
    if (!c || !send_dev || count < 0) return cfail(WP_ERR_INVALID, "%s: bad argument", who);
    if (root < 0 || root >= c->nranks) return cfail(WP_ERR_INVALID, "%s: root %d of %d ranks", who, root, c->nranks);
    if (need_recv && !recv_dev) return cfail(WP_ERR_INVALID, "%s: the receiving rank needs a result buffer", who);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev != c->dev)
        return cfail(WP_ERR_INVALID, "%s: the communicator belongs to device %d, device %d is current", who, c->dev, dev);
    return WP_OK;
}

extern "C" int wp_bus_reduce(wp_comm *c, const float *send_dev, float *recv_dev, int64_t count, int root, void *stream)
{
    const int n = wp_init();
    if (n < 0) return n;
    if (int rc = check_reduce_args(c, send_dev, recv_dev, count, root, c && c->rank == root, "wp_bus_reduce")) return rc;
    if (count == 0) return WP_OK;
    RCCLCHK(g_rccl.Reduce(send_dev, recv_dev, (size_t)count, ncclFloat32, ncclSum, root, c->comm, (hipStream_t)stream), "ncclReduce");
    return WP_OK;
}

extern "C" int wp_bus_allreduce(wp_comm *c, const float *send_dev, float *recv_dev, int64_t count, void *stream)
{
    const int n = wp_init();
    if (n < 0) return n;
    if (int rc = check_reduce_args(c, send_dev, recv_dev, count, 0, true, "wp_bus_allreduce")) return rc;
    if (count == 0) return WP_OK;
    RCCLCHK(g_rccl.AllReduce(send_dev, recv_dev, (size_t)count, ncclFloat32, ncclSum, c->comm, (hipStream_t)stream), "ncclAllReduce");
    return WP_OK;
}

extern "C" const char *wp_comm_library(void)
{
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    return g_rccl.so ? g_rccl.path.c_str() : "";
}
