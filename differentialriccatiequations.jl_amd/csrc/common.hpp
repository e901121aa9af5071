// Common infrastructure of libdre_hip: context, error handling, stream-ordered device pool,
// dense device matrix views.  gfx950 (MI355X) only.
#pragma once
#include <atomic>
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

namespace dre {

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

// status codes of the C ABI (include/dre_hip.h)
enum : int {
    ERR_OK = 0,
    ERR_INVALID = -1,
    ERR_HIP = -2,
    ERR_ALLOC = -3,
    ERR_SINGULAR = -4,
    ERR_INTERNAL = -5,
    ERR_NODEVICE = -6,
};

#define DRE_HIP(expr)                                                                       \
    do {                                                                                    \
        hipError_t e__ = (expr);                                                            \
        if (e__ != hipSuccess)                                                              \
            throw ::dre::Error(::dre::ERR_HIP, std::string(#expr) + ": " +              \
                                                       hipGetErrorString(e__));             \
    } while (0)

#define DRE_REQUIRE(cond, msg)                                                              \
    do {                                                                                    \
        if (!(cond)) throw ::dre::Error(::dre::ERR_INVALID, std::string(msg));          \
    } while (0)

// ---------------------------------------------------------------------------------------------
// Stream-ordered caching device allocator.  Everything the library launches goes to ONE private
// stream per context, so a block released by the host may be handed out again immediately: the
// next kernel that touches it is ordered after the last kernel that used it.
// ---------------------------------------------------------------------------------------------
class DevicePool {
  public:
    ~DevicePool() { trim_locked(); }
    void* alloc(size_t bytes) {
        std::lock_guard<std::mutex> lk(mu_);       // (buffers are handed out by the thread that drives the pool's stream, but released by whoever drops the last handle)
        size_t sz = round_up(bytes);
        auto it = free_.lower_bound(sz);
        if (it != free_.end() && it->first <= sz * 2) {
            void* p = it->second;
            live_[p] = it->first;
            free_.erase(it);
            return p;
        }
        void* p = nullptr;
        ++misses_;
        hipError_t e = hipMalloc(&p, sz);
        if (e != hipSuccess) {
            trim_locked();
            e = hipMalloc(&p, sz);
            if (e != hipSuccess) throw Error(ERR_ALLOC, "hipMalloc failed for " + std::to_string(sz) + " bytes");
        }
        live_[p] = sz;
        total_ += sz;
        return p;
    }
    void release(void* p) {
        if (!p) return;
        std::lock_guard<std::mutex> lk(mu_);
        auto it = live_.find(p);
        if (it == live_.end()) return;
        free_.emplace(it->second, p);
        live_.erase(it);
    }
    void trim() { std::lock_guard<std::mutex> lk(mu_); trim_locked(); }
    size_t total_bytes() const { return total_; }
    long misses() const { return misses_; }          // allocations that went to hipMalloc

  private:
    void trim_locked() {
        for (auto& kv : free_) { (void)hipFree(kv.second); total_ -= kv.first; }
        free_.clear();
    }
    std::mutex mu_;
    long misses_ = 0;
    static size_t round_up(size_t b) {
        if (b < 256) b = 256;
        size_t g = b < (1u << 20) ? 4096 : (1u << 20);
        return (b + g - 1) / g * g;
    }
    std::multimap<size_t, void*> free_;
    std::unordered_map<void*, size_t> live_;
    size_t total_ = 0;
};

#define DRE_ADI_MAX_ITERS_LIMIT 100000     // = DRE_ADI_MAX_ITERS of include/dre_hip.h
struct KernelTimer;  // profiling.hpp
struct Comm;         // comm.hpp

struct Ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    // The pool is shared with every buffer allocated from it: a device object that outlives its context (finalizers of a garbage-collected
    // host language run in no particular order) still returns its block to a live pool, which frees the memory when the last owner goes.
    std::shared_ptr<DevicePool> pool_sp = std::make_shared<DevicePool>();
    DevicePool& pool = *pool_sp;
    std::string last_error;
    int num_cus = 256;
    // optional per-kernel-class timing with HIP events (bench.py's roofline leg)
    std::unique_ptr<KernelTimer> timer;
    // pinned scratch for small device->host reads
    void* pinned = nullptr;
    size_t pinned_bytes = 0;
    // tunables (dre_ctx_set_option): real shifts of pencils with n <= dense_inv_max_n use the cached dense inverse
    int dense_inv_max_n = 1536;
    // compression: form S = L D L' (n x n) directly instead of going through QR(L) when the factor has at least
    // n / compress_direct_ratio columns and n <= compress_direct_max_n (always for n <= 512)
    int compress_direct_max_n = 2560;
    double compress_direct_ratio = 8.0;
    // compression in factor form (no QR of L, no n x n matrix) for n >= compress_factor_min_n and at least compress_factor_min_cols columns
    int compress_factor_min_n = 2561;
    int compress_factor_min_cols = 96;
    // wide factors (c >= compress_sketch_min_cols and c >= compress_sketch_ratio x sketch width) of a PSD-like sum are compressed through a randomized range
    // finder (ldlt.hip, sketch_compress): three GEMM passes over the n x c factor instead of four per 16 columns of rank; the sketch width
    // is the rank of the previous compression of this kind + compress_sketch_extra; 0 disables
    int compress_sketch = 1;
    int compress_sketch_min_cols = 320;
    int compress_sketch_extra = 48;
    double compress_sketch_ratio = 1.25; // columns >= ratio x sketch width (sweep at n = 5177 / 20209, 45 / 12 steps: 3.0 -> 257 / 246 ms, 2.0 -> 252 / 233, 1.25 -> 241 / 234)
    int ros2_tight = 1;                 // Ros2 driver: stage right-hand sides and stage solutions truncated at the reference's rank (engine.hpp, COMPRESS_TIGHT)
    int mf_swizzle = 0;                 // XCD-aware workgroup order of the multifrontal sweep kernels (sparse.hip, mf_block)
    int gemm_swizzle = 1;               // XCD-aware workgroup -> tile order of the split-K GEMM (gemm.hip, xcd_tile); 0: launch order
    int compress_sketch_sparse = 1;     // sketch with the structured sparse sign test matrix (one pass over the factor) instead of a Gaussian one (dense GEMM)
    int compress_sketch_cholqr = 1;     // orthonormalise the sketch by blocked Cholesky QR (GEMMs) instead of Householder/TSQR panels; falls back on breakdown
    // multifrontal sweeps: the top levels of the elimination tree with at most this many pivot variables are applied as one dense
    // inverse of their Schur complement (reused real factors only; 0 disables)
    int top_inverse_max_rows = 1536;
    // multifrontal sweeps below the dense top: one workgroup per subtree and 16 right-hand-side columns (sparse.hip, SubPlan) instead of one
    // launch per tree level; read when a pencil runs its first solve.  0 (default): level kernels everywhere — on the SteelProfile trees the
    // two forms measure the same (sparse.hip)
    int mf_subtree = 0;
    // Ros1, n <= 1536, no save_state: X is carried as "compressed warm start + ADI increments"; its compression runs on a second stream
    // beside the next time step (x_side_stream) or only every x_compress_every-th step (gdre.hip, gdre_solve)
    // Ros1 without save_state, real Cyclic shifts, n <= dense_x_max_n: X is carried as a dense symmetric n x n matrix between the time steps
    // (gdre.hip, ros1_dense_step); 0 disables
    int dense_x_max_n = 1536;
    // residual factors wider than this leave the dense-X loop for the factored path (0: the fast chain's own limit ADI_FAST_MAX_K); also the
    // switch the tests use to force that fallback in the middle of a run
    int dense_x_max_k = 0;
    // group chain of the dense-X loop (dense.hip k_adi_group, gdre.hip group_ops_prepare): g ADI iterations per launch.  1 = auto (largest
    // divisor of the cycle length up to 5), 0 = off (one launch per iteration), g >= 2 = that group size if it divides the cycle length
    int adi_group = 1;
    int adi_group_max_n = 768;
    // fan groups of the general path (engine.hip): up to adi_fan consecutive real-shift iterations from independent solves that share every
    // launch (0 / 1 = off, at most 10 = a whole cycle of the usual lists); a group is cut where the partial-fraction coefficients exceed adi_fan_max_coef
    int adi_fan = 8;
    double adi_fan_max_coef = 128.0;
    // pivot-free multifrontal LU: multipliers beyond pivot_growth_warn flag the ADI result (DRE_WARN_PIVOT_GROWTH) and trigger a true-residual
    // verification; beyond pivot_growth_fail the factorisation is rejected (DRE_ERR_SINGULAR)
    double pivot_growth_warn = 1e8, pivot_growth_fail = 1e13;
    // static pivoting of the multifrontal LU (sparse.hpp, Factor): relative floor for the pivots (sqrt(eps) like SuperLU_DIST; 0 = off: plain
    // pivot-free LU with growth detection only) and the number of refinement steps of a solve with a perturbed factor
    double pivot_static = 1.4901161193847656e-08;
    int pivot_refine_steps = 3;
    // Rosenbrock-1 on the general path (multifrontal solves, Cyclic real shifts): warm-start residual and feedback from the ADI's own residual
    // recurrence, X compressed on the side stream (gdre.hip, ros1_recurrence_loop); 0: the reference's order of operations
    int ros1_recurrence = 1;
    int x_side_stream = 1;
    // dense-X time loop: the side stream's set-up (SMW products, stacks) is enqueued after this many panels of the residual's band
    // reduction (the device is busy with them); -1: inside the reduction's read-back
    int side_after_panels = 1;
    // dense-X time loop: warm-started compression of the residual (warm.hip: Rayleigh-Ritz in the previous step's basis, probe verified); 0: the
    // full band reduction at every step
    int dense_warm = 1;
    // dense-X time loop: the side stream's set-up of step i + 1 is enqueued by a parked host thread at the end of step i (0: inside step i + 1)
    int recurrence_wide = 1;    // residual-recurrence loop: the increments of a solve may exceed the factor-form limit (c + 64 <= n); 0: round 4's in-loop compression
    int side_gate = 0;          // residual-recurrence loop: the parked thread enqueues the side stream only while the time loop's thread waits (launch gate below; measured at n = 5177: 85.1 ms with, 85.4 ms without — inside the noise, off)
    int side_prefetch = 0;      // (measured at n = 371: 18.3 against 17.8 ms per solve — the join in front of the chain costs more than the earlier start gains)
    // dense-inverse path: factorisations, explicit inverses and stacks of a whole Cyclic list in shared launches (gdre.hip, cycle_setup_batched)
    int setup_batched = 1;
    int xwarm_sx = 0;          // residual-recurrence loop: fresh sketch columns of the warm-started compression of X on the side stream (0 = by configuration: gdre.hip)
    // self-generated shift lists: the upcoming factorisations that are not in flight yet go out in shared launches (engine.hip, prefetch_ahead)
    int prefetch_batch = 4;    // low-water mark: refill (in one batch) when at most this many are in flight; 0 = off
    int x_compress_every = 1;
    // band reduction: number of panels the previous reduction of the same kind needed (speculation depth of the next one)
    std::map<long, int> band_hint;
    // second context (own stream, pool, hints) for work that runs beside the main stream (gdre.hip: side-stream compression of X);
    // created on first use, lives as long as this context
    std::unique_ptr<Ctx> side;
    hipEvent_t side_e1 = nullptr, side_e2 = nullptr;
    // helper contexts (own stream and pool each) for independent chains of small kernels that would otherwise queue up behind each other
    // on one stream — the factorisations and dense inverses of the shifts of a cycle (gdre.hip, cycle_ops_prepare); created on first use
    std::vector<std::unique_ptr<Ctx>> helpers;
    std::vector<hipEvent_t> helper_ev;
    hipEvent_t helper_e0 = nullptr;
    hipEvent_t aux_ev[4] = {nullptr, nullptr, nullptr, nullptr};   // fork / join events of work that runs on a helper stream beside the main one
    int setup_streams = 5;      // 0 / 1: everything on the calling context's stream
    // hipFuncSetAttribute(MaxDynamicSharedMemorySize) is per device: remembered per context, not per process
    bool attr_adi_fast = false;
    // host-visible landing zone for small device->host reads on the critical path (ctx_fetch, dense.hip): a tiny kernel copies the words
    // into pinned, device-mapped host memory and bumps a sequence number the host spins on — an order of magnitude cheaper than a copy
    // command + hipStreamSynchronize per read-back
    struct FetchZone { volatile unsigned long long seq; unsigned long long pad[7]; unsigned long long words[1024]; };
    FetchZone* fetch_host = nullptr;   // host address
    // Launch gate between the time loop's thread and the parked thread that drives the side stream (gdre.hip, ros1_recurrence_loop): two threads
    // enqueueing at once contend for the HIP runtime, and the MAIN thread's launches (the critical path) took 10-60 us each instead of 4 while the side
    // job's hundred launches went out.  The main thread raises `waiting` while it blocks on a read-back or a join; the side context (gate_follow set)
    // only enqueues then — or after gate_patience_us without such a window.
    struct LaunchGate { std::atomic<int> waiting{1}; };
    std::shared_ptr<LaunchGate> gate;          // main context: owner;  side context: follower when gate_follow
    bool gate_follow = false;
    int gate_patience_us = 300;
    // parked host threads of the time loops (gdre.hip, SideWorker): kept with the context — a thread per solve cost ~0.1 ms of creation and join
    std::shared_ptr<void> parked_worker[3];
    std::shared_ptr<struct Buf> warm_tickets;    // arrival counters of the warm-started compression's ticket kernels (warm.hip), zeroed once
    void* dense_land = nullptr;        // pinned landing zone of the dense-X time loop (gdre.hip, DenseXState): allocated once per context
    FetchZone* fetch_dev = nullptr;    // the same memory as the device sees it
    unsigned long long fetch_seq = 0;
    bool fetch_spin = true;            // false: ctx_fetch blocks in hipStreamSynchronize instead of spinning (the side context: its driver thread must not burn a core beside the main thread)
    // kernels whose dynamic-LDS limit was raised for this context's device (hipFuncSetAttribute is per device, so the record is per
    // context, not per process)
    std::unordered_map<const void*, int> lds_attr_done;
    // compression bookkeeping of this context (diagnostics only)
    struct CompressCounters { long calls = 0, cols_in = 0, order = 0, tri_steps = 0, rank_out = 0; } cstats;
    // counters of the DRE_TRACE diagnostics (per context: two contexts on two threads do not share them)
    struct TraceCounters {
        long pf_calls = 0, pf_nups = 0, pf_pend = 0; double fan_t[6] = {0, 0, 0, 0, 0, 0}; long fan_n = 0; long rl_hit = 0, rl_miss = 0, cx_hit = 0, cx_miss = 0;
        double ch_enq = 0.0, ch_wait = 0.0; long ch_n = 0, ch_it = 0; double rec_tb = 0.0, rec_ta = 0.0, rec_tf = 0.0; long rec_ns = 0; int clock_count = 0, subprobe_count = 0;
    } trace;
    bool prof_side = false;     // timing was switched on before the side context existed: it is created with its timer enabled
    // multi-GPU (comm.hpp, dre_comm_init): with a communicator of more than one rank the shifted solves of the generic ADI path are
    // column-sharded (engine.hip, adi_advance) — every rank solves its 16-column tiles of the residual block and ONE in-place all-gather per
    // ADI step, enqueued on this stream, gives every rank the full V; everything else is replicated and stays bit-identical on all ranks.
    // Residual blocks narrower than shard_min_cols are solved replicated (a rank cannot do less than one 16-column tile).
    std::shared_ptr<Comm> comm;
    int shard_min_cols = 32;
    // host transport (dre_comm_init_host): 1 = the callbacks run as host functions ON the stream (hipLaunchHostFunc), ordered by the stream only like
    // an RCCL collective; 0 = synchronously around a stream synchronisation (comm.hip)
    int comm_host_async = 0;
    // user-supplied orthogonalisation (the reference's extension point DifferentialRiccatiEquations.orthf, src/LDLt.jl:227-245, overridden in
    // test/cuda.jl:32-37): L (n x c) -> Q (n x p, orthonormal columns), R (p x c), p = min(n, c), L = Q R; device pointers, column-major.
    // Honoured by the literal compression (compress_exact / dre_ldlt_compress) and by dre_ldlt_norm.  Called with this stream idle; must
    // return when its own work is complete.
    int (*orthf_fn)(void* user, int n, int c, const double* L, int ldl, double* Q, int ldq, double* R, int ldr) = nullptr;
    void* orthf_user = nullptr;
    long orthf_calls = 0;
    void sync() { DRE_HIP(hipStreamSynchronize(stream)); }
};

struct Buf {
    std::shared_ptr<DevicePool> pool;      // keeps the pool (not the context) alive
    void* p;
    size_t bytes;
    Buf(Ctx* c, size_t b) : pool(c->pool_sp), p(c->pool_sp->alloc(b ? b : 8)), bytes(b) {}
    ~Buf() { pool->release(p); }
    Buf(const Buf&) = delete;
    Buf& operator=(const Buf&) = delete;
};
using BufP = std::shared_ptr<Buf>;

template <typename T>
struct DevArr {  // typed owning device array
    BufP buf;
    T* p = nullptr;
    size_t n = 0;
    DevArr() = default;
    DevArr(Ctx* c, size_t n_) : buf(std::make_shared<Buf>(c, n_ * sizeof(T))), p((T*)buf->p), n(n_) {}
    void upload(Ctx* c, const T* h, size_t cnt) {
        DRE_HIP(hipMemcpyAsync(p, h, cnt * sizeof(T), hipMemcpyHostToDevice, c->stream));
        DRE_HIP(hipStreamSynchronize(c->stream));
    }
    void upload(Ctx* c, const std::vector<T>& h) { if (!h.empty()) upload(c, h.data(), h.size()); }
};

// Column-major dense f64 matrix view on the device (Julia `Matrix{Float64}` layout).
struct Mat {
    BufP buf;
    double* p = nullptr;
    int rows = 0, cols = 0, ld = 0;
    Mat() = default;
    Mat(Ctx* c, int r, int cc) : buf(std::make_shared<Buf>(c, (size_t)(r > 0 ? r : 1) * (cc > 0 ? cc : 1) * sizeof(double))),
                                  p((double*)buf->p), rows(r), cols(cc), ld(r > 0 ? r : 1) {}
    Mat view(int r0, int c0, int r, int c) const {
        Mat m;
        m.buf = buf; m.p = p + (size_t)r0 + (size_t)c0 * ld; m.rows = r; m.cols = c; m.ld = ld;
        return m;
    }
    Mat colsview(int c0, int c) const { return view(0, c0, rows, c); }
    bool empty() const { return rows == 0 || cols == 0; }
};

inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

// Diagnostics on stderr: DRE_TRACE=<keyword>[,<keyword>...] (or "all").  Keywords: compress, cholqr, proj, prefetch, subtree (what a code path
// decided), rank (eigenvalues of every finished compression's band matrix on the host: its rank against the reference's), phase, chunk, rec, fan (host/device timings of the loops).  The engine's TUNABLES are options (dre_ctx_set_option; DRE_OPTIONS=
// "name=value,..." sets them for every context created by the process) — environment variables do not select kernels.
inline bool env_trace(const char* key) {
    const char* e = std::getenv("DRE_TRACE");
    if (!e || !*e) return false;
    const std::string s = std::string(",") + e + ",";
    return s.find(",all,") != std::string::npos || s.find(std::string(",") + key + ",") != std::string::npos;
}

// raise the dynamic shared memory limit of a kernel once per context (device)
// Streams of a library context by role.  HIP maps streams onto a handful of hardware queues round-robin PER PRIORITY LEVEL; with everything at
// the default priority the fifth, sixth ... stream of a process lands on the queue of the main stream and its kernels are serialised with the
// critical path.  DRE_STREAM_PRIORITIES=1 gives the main stream the highest priority (a queue of its own), the side stream the middle one and
// the helper streams the lowest.  Measured (round 3, default-ADI runs with the factorisations on helper streams): no gain at depth 5 and a
// LOSS at depth 8 (n = 1357 Ros2: 2 830 against 3 250 it/s — low-priority factorisations are starved by the main stream's kernels and arrive
// late), so equal priorities stay the default.   role: 0 main, 1 side, 2 helper.
inline hipStream_t create_stream(int role) {
    static const bool on = false;
    hipStream_t st = nullptr;
    int least = 0, greatest = 0;
    if (on && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && least != greatest) {
        // numerically lower = higher priority
        const int mid = (least + greatest) / 2;
        const int prio = role == 0 ? greatest : (role == 1 ? mid : least);
        if (hipStreamCreateWithPriority(&st, hipStreamNonBlocking, prio) == hipSuccess) return st;
        (void)hipGetLastError();
    }
    DRE_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    return st;
}

inline void lds_attr(Ctx* ctx, const void* func, int bytes) {
    auto it = ctx->lds_attr_done.find(func);
    if (it != ctx->lds_attr_done.end() && it->second >= bytes) return;
    DRE_HIP(hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    ctx->lds_attr_done[func] = bytes;
}

}  // namespace dre
