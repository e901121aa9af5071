// RCCL collectives on the library's stream (comm.hpp).  librccl is resolved at run time: the same soname PyTorch-ROCm ships
// ("librccl.so.1"), so a process that already uses torch.distributed's nccl backend shares that copy instead of loading a second one.
#include "comm.hpp"
#include "profiling.hpp"

#include <dlfcn.h>

#include <mutex>

namespace dre {

namespace {
// the five entry points of rccl.h this library uses (ncclUniqueId is passed BY VALUE: a 128-byte struct)
struct UniqueId { char internal[128]; };
using fn_get_id = int (*)(UniqueId*);
using fn_init = int (*)(void**, int, UniqueId, int);
using fn_destroy = int (*)(void*);
using fn_allgather = int (*)(const void*, void*, size_t, int, void*, hipStream_t);
using fn_allreduce = int (*)(const void*, void*, size_t, int, int, void*, hipStream_t);
using fn_errstr = const char* (*)(int);
constexpr int NCCL_FLOAT64 = 8, NCCL_SUM = 0;      // rccl.h: ncclFloat64 = 8, ncclSum = 0

struct Rccl {
    void* h = nullptr;
    fn_get_id get_id = nullptr; fn_init init = nullptr; fn_destroy destroy = nullptr;
    fn_allgather allgather = nullptr; fn_allreduce allreduce = nullptr; fn_errstr errstr = nullptr;
    std::string err;
};
Rccl& rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* names[] = {std::getenv("DRE_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* nm : names) {
            if (!nm) continue;
            r.h = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
            if (r.h) break;
            const char* e = dlerror();               // (one call: dlerror() clears the message it returns)
            r.err = e ? e : "dlopen failed";
        }
        if (!r.h) return;
        r.get_id = (fn_get_id)dlsym(r.h, "ncclGetUniqueId");
        r.init = (fn_init)dlsym(r.h, "ncclCommInitRank");
        r.destroy = (fn_destroy)dlsym(r.h, "ncclCommDestroy");
        r.allgather = (fn_allgather)dlsym(r.h, "ncclAllGather");
        r.allreduce = (fn_allreduce)dlsym(r.h, "ncclAllReduce");
        r.errstr = (fn_errstr)dlsym(r.h, "ncclGetErrorString");
        if (!(r.get_id && r.init && r.destroy && r.allgather && r.allreduce)) { r.err = "librccl lacks an expected symbol"; r.h = nullptr; }
    });
    return r;
}
void need_rccl() {
    if (!rccl().h) throw Error(ERR_INTERNAL, "RCCL is not available: " + rccl().err);
}
void chk(int rc, const char* what) {
    if (rc == 0) return;
    const char* s = rccl().errstr ? rccl().errstr(rc) : "?";
    throw Error(ERR_INTERNAL, std::string(what) + ": RCCL error " + std::to_string(rc) + " (" + s + ")");
}
}  // namespace

Comm::~Comm() {
    if (nccl && rccl().destroy) (void)rccl().destroy(nccl);
    if (stage) (void)hipHostFree(stage);
}
static void* host_stage(Comm& c, size_t bytes) {
    if (c.stage_bytes < bytes) {
        if (c.stage) (void)hipHostFree(c.stage);
        c.stage = nullptr; c.stage_bytes = 0;
        DRE_HIP(hipHostMalloc(&c.stage, bytes, hipHostMallocDefault));
        c.stage_bytes = bytes;
    }
    return c.stage;
}
std::shared_ptr<Comm> comm_init_host(Ctx*, int nranks, int rank, int (*ag)(void*, const void*, void*, size_t), int (*ar)(void*, void*, size_t), void* user) {
    DRE_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, "dre_comm_init_host: rank outside [0, nranks)");
    DRE_REQUIRE(ag != nullptr && ar != nullptr, "dre_comm_init_host: both callbacks are required");
    auto c = std::make_shared<Comm>();
    c->nranks = nranks; c->rank = rank;
    c->host_allgather = ag; c->host_allreduce = ar; c->host_user = user;
    return c;
}

void comm_unique_id(void* out128) {
    need_rccl();
    UniqueId id;
    std::memset(&id, 0, sizeof(id));
    chk(rccl().get_id(&id), "ncclGetUniqueId");
    std::memcpy(out128, &id, sizeof(id));
}

std::shared_ptr<Comm> comm_init(Ctx* ctx, int nranks, int rank, const void* id128) {
    DRE_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, "dre_comm_init: rank outside [0, nranks)");
    auto c = std::make_shared<Comm>();
    c->nranks = nranks; c->rank = rank;
    if (nranks == 1 && !id128) return c;                 // a single rank without an id: no RCCL object at all
    need_rccl();
    DRE_REQUIRE(id128 != nullptr, "dre_comm_init: the unique id of rank 0 is required for more than one rank");
    UniqueId id;
    std::memcpy(&id, id128, sizeof(id));
    DRE_HIP(hipSetDevice(ctx->device));
    chk(rccl().init(&c->nccl, nranks, id, rank), "ncclCommInitRank");
    return c;
}

// ASYNCHRONOUS host transport (option "comm_host_async"): the collective is ordered by the context's stream ONLY, like ncclAllGather — the copy
// down, the caller's collective (a host function enqueued on the stream: hipLaunchHostFunc) and the copy up are three stream operations and the
// calling thread goes on enqueueing at once.  This is the transport the two-rank tests use to expose ordering mistakes between the library's
// streams that the synchronous form (a hipStreamSynchronize on either side of the callback) would hide.  One pinned staging buffer serves every
// call: the three operations of a call are ordered with those of the next by the stream itself.
struct HostJob { Comm* c; char* stage; size_t bytes; size_t count; int kind; };
static void host_job_run(void* p) {
    HostJob* j = static_cast<HostJob*>(p);
    int rc;
    if (j->kind == 0) rc = j->c->host_allgather(j->c->host_user, j->stage + j->bytes * (size_t)j->c->rank, j->stage, j->bytes);
    else rc = j->c->host_allreduce(j->c->host_user, j->stage, j->count);
    if (rc != 0) j->c->async_failed = 1;
    delete j;
}
static void host_async_check(Comm& c) {
    if (c.async_failed) { c.async_failed = 0; throw Error(ERR_INTERNAL, "dre_comm: a host collective callback failed (asynchronous transport)"); }
}
void comm_allgather(Ctx* ctx, Comm& c, const double* send, double* recv, size_t count) {
    c.ncalls++;
    if (c.nranks > 1 && c.host_allgather && ctx->comm_host_async) {
        host_async_check(c);
        const size_t bytes = count * sizeof(double);
        if (c.stage_bytes < bytes * (size_t)c.nranks) DRE_HIP(hipStreamSynchronize(ctx->stream));      // (growing the buffer: nothing may be in flight)
        char* st = (char*)host_stage(c, bytes * (size_t)c.nranks);
        DRE_HIP(hipMemcpyAsync(st + bytes * (size_t)c.rank, send, bytes, hipMemcpyDeviceToHost, ctx->stream));
        DRE_HIP(hipLaunchHostFunc(ctx->stream, host_job_run, new HostJob{&c, st, bytes, count, 0}));
        DRE_HIP(hipMemcpyAsync(recv, st, bytes * (size_t)c.nranks, hipMemcpyHostToDevice, ctx->stream));
        c.bytes_gathered += bytes * (size_t)(c.nranks - 1);
        return;
    }
    if (c.nranks > 1 && c.host_allgather) {
        // host transport: own block down, the caller's collective on host memory, everything up — synchronous (tests, RCCL-less hosts)
        const size_t bytes = count * sizeof(double);
        char* st = (char*)host_stage(c, bytes * (size_t)c.nranks);
        DRE_HIP(hipMemcpyAsync(st + bytes * (size_t)c.rank, send, bytes, hipMemcpyDeviceToHost, ctx->stream));
        DRE_HIP(hipStreamSynchronize(ctx->stream));
        if (c.host_allgather(c.host_user, st + bytes * (size_t)c.rank, st, bytes) != 0) throw Error(ERR_INTERNAL, "dre_comm: the host all-gather callback failed");
        DRE_HIP(hipMemcpyAsync(recv, st, bytes * (size_t)c.nranks, hipMemcpyHostToDevice, ctx->stream));
        DRE_HIP(hipStreamSynchronize(ctx->stream));
        c.bytes_gathered += bytes * (size_t)(c.nranks - 1);
        return;
    }
    if (c.nranks == 1 || !c.nccl) {
        if (send != recv + (size_t)c.rank * count && count)
            DRE_HIP(hipMemcpyAsync(recv + (size_t)c.rank * count, send, count * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
        return;
    }
    c.bytes_gathered += count * sizeof(double) * (size_t)(c.nranks - 1);
    chk(rccl().allgather(send, recv, count, NCCL_FLOAT64, c.nccl, ctx->stream), "ncclAllGather");
}
void comm_allgather_inplace(Ctx* ctx, Comm& c, double* buf, size_t count) {
    comm_allgather(ctx, c, buf + (size_t)c.rank * count, buf, count);
}
void comm_allreduce_sum(Ctx* ctx, Comm& c, double* buf, size_t count) {
    c.ncalls++;
    if (c.nranks > 1 && c.host_allreduce && ctx->comm_host_async) {
        host_async_check(c);
        const size_t bytes = count * sizeof(double);
        if (c.stage_bytes < bytes) DRE_HIP(hipStreamSynchronize(ctx->stream));
        char* st = (char*)host_stage(c, bytes);
        DRE_HIP(hipMemcpyAsync(st, buf, bytes, hipMemcpyDeviceToHost, ctx->stream));
        DRE_HIP(hipLaunchHostFunc(ctx->stream, host_job_run, new HostJob{&c, st, bytes, count, 1}));
        DRE_HIP(hipMemcpyAsync(buf, st, bytes, hipMemcpyHostToDevice, ctx->stream));
        c.bytes_reduced += bytes;
        return;
    }
    if (c.nranks > 1 && c.host_allreduce) {
        const size_t bytes = count * sizeof(double);
        void* st = host_stage(c, bytes);
        DRE_HIP(hipMemcpyAsync(st, buf, bytes, hipMemcpyDeviceToHost, ctx->stream));
        DRE_HIP(hipStreamSynchronize(ctx->stream));
        if (c.host_allreduce(c.host_user, st, count) != 0) throw Error(ERR_INTERNAL, "dre_comm: the host all-reduce callback failed");
        DRE_HIP(hipMemcpyAsync(buf, st, bytes, hipMemcpyHostToDevice, ctx->stream));
        DRE_HIP(hipStreamSynchronize(ctx->stream));
        c.bytes_reduced += bytes;
        return;
    }
    if (c.nranks == 1 || !c.nccl) return;
    c.bytes_reduced += count * sizeof(double);
    chk(rccl().allreduce(buf, buf, count, NCCL_FLOAT64, NCCL_SUM, c.nccl, ctx->stream), "ncclAllReduce");
}

Roctx& Roctx::get() {
    static Roctx r = [] {
        Roctx x;
        if (!env_trace("roctx")) return x;
        void* h = dlopen("librocprofiler-sdk-roctx.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("libroctx64.so.4", RTLD_NOW | RTLD_GLOBAL);
        if (h) {
            x.push = reinterpret_cast<int (*)(const char*)>(dlsym(h, "roctxRangePushA"));
            x.pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
            if (!x.push || !x.pop) { x.push = nullptr; x.pop = nullptr; }
        }
        return x;
    }();
    return r;
}

}  // namespace dre
