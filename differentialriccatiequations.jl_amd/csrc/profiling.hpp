// Per-kernel-class timing with HIP events on the library's own stream.
// Used by bench.py's roofline leg (torch.cuda.Event only sees torch's stream).
#pragma once
#include <chrono>
#include "common.hpp"

namespace dre {

struct KernelStat {
    double ms = 0.0;
    long launches = 0;
    double bytes = 0.0;   // algorithmic bytes attributed by the launcher
    double flops = 0.0;
};

struct KernelTimer {
    struct Pending { std::string name; hipEvent_t a, b; double bytes, flops; long count; };
    std::vector<Pending> pending;
    std::vector<hipEvent_t> free_events;
    std::map<std::string, KernelStat> stats;
    bool enabled = false;

    hipEvent_t get_event() {
        if (!free_events.empty()) { hipEvent_t e = free_events.back(); free_events.pop_back(); return e; }
        hipEvent_t e; DRE_HIP(hipEventCreate(&e)); return e;
    }
    void collect(Ctx* ctx) {
        if (pending.empty()) return;
        DRE_HIP(hipStreamSynchronize(ctx->stream));
        for (auto& p : pending) {
            float ms = 0.f;
            DRE_HIP(hipEventElapsedTime(&ms, p.a, p.b));
            auto& s = stats[p.name];
            s.ms += ms; s.launches += p.count; s.bytes += p.bytes; s.flops += p.flops;
            free_events.push_back(p.a); free_events.push_back(p.b);
        }
        pending.clear();
    }
    ~KernelTimer() {
        for (auto& p : pending) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
        for (auto e : free_events) (void)hipEventDestroy(e);
    }
};

// RAII scope: records an event pair around whatever is launched inside when timing is enabled.
struct TimedScope {
    // count > 1: the scope brackets a back-to-back chain of `count` launches of one class (one event pair for the whole chain: the
    // per-launch average then includes the launch boundaries and is not inflated by two event packets per short kernel)
    Ctx* ctx; bool on; hipEvent_t a{}, b{}; const char* name; double bytes, flops; long count;
    TimedScope(Ctx* c, const char* nm, double by = 0, double fl = 0, long cnt = 1)
        : ctx(c), on(c->timer && c->timer->enabled), name(nm), bytes(by), flops(fl), count(cnt) {
        if (c->gate_follow && c->gate && !c->gate->waiting.load(std::memory_order_relaxed)) {
            // (every kernel class of the library is launched inside a TimedScope: this is the one place all launches of a context pass)
            const auto t0 = std::chrono::steady_clock::now();
            while (!c->gate->waiting.load(std::memory_order_relaxed) &&
                   std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() < (double)c->gate_patience_us) {}
        }
        if (on) {
            a = ctx->timer->get_event(); b = ctx->timer->get_event();
            (void)hipEventRecord(a, ctx->stream);
        }
    }
    ~TimedScope() {
        if (on) {
            (void)hipEventRecord(b, ctx->stream);
            ctx->timer->pending.push_back({name, a, b, bytes, flops, count});
            if (ctx->timer->pending.size() > 4096) {
                try { ctx->timer->collect(ctx); } catch (...) {}
            }
        }
    }
};


// roctx ranges named after the reference's TimerOutputs sections (src/DifferentialRiccatiEquations.jl:22 @timeit_debug):
//   "norm(::LDLᵀ)" LDLt.jl:77 · "compress!(::LDLᵀ)" LDLt.jl:204 · "orthf" :211 · "eigen" :214 · "solve (real)" adi.jl:157 · "solve (complex)" adi.jl:196 ·
//   "Sherman-Morrison-Woodbury" smw.jl:10,33 · "solve (sparse)" smw.jl:19,36 · "solve (dense)" smw.jl:39 · "shifts" adi.jl:101 ·
//   "residual(::GALEProblem, ::LDLᵀ)" lyapunov/residual.jl:3 · "ADI" lowrank_ros1.jl:49 ("$(typeof(inner_alg))")
// They show in `rocprofv3 --marker-trace` timelines next to the kernels.  The marker library (librocprofiler-sdk-roctx.so, else libroctx64.so) is
// resolved with dlopen on first use and ONLY when DRE_TRACE lists "roctx": libdre_hip.so keeps linking against libamdhip64 alone.
struct Roctx {
    int (*push)(const char*) = nullptr;
    int (*pop)() = nullptr;
    static Roctx& get();
};
struct RoctxRange {
    bool on;
    explicit RoctxRange(const char* name) : on(false) { Roctx& r = Roctx::get(); if (r.push) { (void)r.push(name); on = true; } }
    ~RoctxRange() { if (on) (void)Roctx::get().pop(); }
    RoctxRange(const RoctxRange&) = delete;
    RoctxRange& operator=(const RoctxRange&) = delete;
};

}  // namespace dre
