#include <functional>
// C ABI of libdre_hip (declared in include/dre_hip.h).
#include "../../include/dre_hip.h"

#include <algorithm>
#include <atomic>
#include <cstring>

#include "comm.hpp"
#include "engine.hpp"
#include "hostla.hpp"
#include "profiling.hpp"

using namespace dre;

struct dre_ctx { Ctx c; std::map<std::string, KernelStat> merged; };
struct dre_dense { Mat m; };
struct dre_pencil {
    std::unique_ptr<Pencil> p;
    std::vector<double> hE, hA;
};
struct dre_factor {
    const dre_pencil* pen = nullptr;
    bool is_cplx = false;
    Factor<double> fr;
    Factor<cplx> fc;
};
struct dre_ldlt {
    LDLtP x;
    const dre_pencil* pen = nullptr;
};
struct dre_adi_result {
    AdiResult r;
    const dre_pencil* pen = nullptr;
};
struct dre_gdre_result {
    GdreResult r;
    const dre_pencil* pen = nullptr;
    int m = 0;
};

static thread_local std::string g_noctx_error;

template <typename F>
static int guarded(dre_ctx* ctx, F&& f) {
    try {
        f();
        return DRE_OK;
    } catch (const Error& e) {
        if (ctx) ctx->c.last_error = e.what(); else g_noctx_error = e.what();
        return e.code;
    } catch (const std::bad_alloc&) {
        if (ctx) ctx->c.last_error = "out of host memory";
        return DRE_ERR_ALLOC;
    } catch (const std::exception& e) {
        if (ctx) ctx->c.last_error = e.what(); else g_noctx_error = e.what();
        return DRE_ERR_INTERNAL;
    }
}

extern "C" {

int dre_version(void) { return 100; }

int dre_ctx_create(int device, dre_ctx** out) {
    if (!out) return DRE_ERR_INVALID;
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) {
        g_noctx_error = "no usable HIP device (libdre_hip has no CPU fallback)";
        return DRE_ERR_NODEVICE;
    }
    if (device < 0 || device >= ndev) { g_noctx_error = "device index out of range"; return DRE_ERR_INVALID; }
    auto* ctx = new dre_ctx();
    int rc = guarded(ctx, [&] {
        DRE_HIP(hipSetDevice(device));
        ctx->c.device = device;
        ctx->c.stream = create_stream(0);
        hipDeviceProp_t prop;
        DRE_HIP(hipGetDeviceProperties(&prop, device));
        ctx->c.num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        ctx->c.timer = std::make_unique<KernelTimer>();
    });
    if (rc != DRE_OK) { g_noctx_error = ctx->c.last_error; delete ctx; return rc; }
    // DRE_OPTIONS="name=value,name=value": the options of dre_ctx_set_option for every context of the process (tools/option_matrix.sh runs
    // the test suite under each configuration this way); an unknown name is an error like in the call
    if (const char* e = std::getenv("DRE_OPTIONS")) {
        std::string all = e;
        size_t pos = 0;
        while (pos < all.size()) {
            size_t end = all.find(',', pos);
            if (end == std::string::npos) end = all.size();
            const std::string item = all.substr(pos, end - pos);
            pos = end + 1;
            // (a typo must not run the default configuration and report green: "name:0", "name=off", "name = 1x" are errors)
            auto trim = [](std::string t) { const size_t a0 = t.find_first_not_of(" \t"); if (a0 == std::string::npos) return std::string(); return t.substr(a0, t.find_last_not_of(" \t") - a0 + 1); };
            const std::string it2 = trim(item);
            if (it2.empty()) continue;
            const size_t eq = it2.find('=');
            int rc2 = DRE_OK;
            if (eq == std::string::npos || eq == 0) { ctx->c.last_error = "item '" + it2 + "' is not of the form name=value"; rc2 = DRE_ERR_INVALID; }
            else {
                const std::string nm = trim(it2.substr(0, eq)), vs = trim(it2.substr(eq + 1));
                char* endp = nullptr;
                const double v = std::strtod(vs.c_str(), &endp);
                if (vs.empty() || endp == vs.c_str() || *endp != '\0') { ctx->c.last_error = "item '" + it2 + "': '" + vs + "' is not a number"; rc2 = DRE_ERR_INVALID; }
                else rc2 = dre_ctx_set_option(ctx, nm.c_str(), v);
            }
            if (rc2 != DRE_OK) { g_noctx_error = "DRE_OPTIONS: " + ctx->c.last_error; dre_ctx_destroy(ctx); return rc2; }
        }
    }
    *out = ctx;
    return DRE_OK;
}
int dre_ctx_destroy(dre_ctx* ctx) {
    if (!ctx) return DRE_OK;
    (void)hipSetDevice(ctx->c.device);
    (void)hipStreamSynchronize(ctx->c.stream);
    for (auto& w : ctx->c.parked_worker) w.reset();          // (parked: they hold no job between solves; joined here, before their streams go)
    ctx->c.comm.reset();
    if (ctx->c.side) {
        Ctx& sc = *ctx->c.side;
        (void)hipStreamSynchronize(sc.stream);
        for (size_t h = 0; h < sc.helpers.size(); ++h) {
            Ctx& hc = *sc.helpers[h];
            (void)hipStreamSynchronize(hc.stream);
            hc.timer.reset(); hc.pool.trim();
            (void)hipEventDestroy(sc.helper_ev[h]);
            (void)hipStreamDestroy(hc.stream);
        }
        sc.helpers.clear(); sc.helper_ev.clear();
        if (sc.helper_e0) { (void)hipEventDestroy(sc.helper_e0); sc.helper_e0 = nullptr; }
        sc.timer.reset(); sc.pool.trim();
        if (sc.fetch_host) { (void)hipHostFree((void*)sc.fetch_host); sc.fetch_host = nullptr; }
        (void)hipEventDestroy(ctx->c.side_e1); (void)hipEventDestroy(ctx->c.side_e2);
        (void)hipStreamDestroy(sc.stream);
        ctx->c.side.reset();
    }
    auto drop_helpers = [](Ctx& owner) {
        for (size_t h = 0; h < owner.helpers.size(); ++h) {
            Ctx& hc = *owner.helpers[h];
            (void)hipStreamSynchronize(hc.stream);
            hc.timer.reset(); hc.pool.trim();
            if (hc.fetch_host) { (void)hipHostFree((void*)hc.fetch_host); hc.fetch_host = nullptr; }
            (void)hipEventDestroy(owner.helper_ev[h]);
            (void)hipStreamDestroy(hc.stream);
        }
        owner.helpers.clear(); owner.helper_ev.clear();
        if (owner.helper_e0) { (void)hipEventDestroy(owner.helper_e0); owner.helper_e0 = nullptr; }
        for (auto& e : owner.aux_ev) if (e) { (void)hipEventDestroy(e); e = nullptr; }
    };
    drop_helpers(ctx->c);
    ctx->c.timer.reset();
    ctx->c.pool.trim();
    if (ctx->c.fetch_host) { (void)hipHostFree((void*)ctx->c.fetch_host); ctx->c.fetch_host = nullptr; }
    if (ctx->c.dense_land) { (void)hipHostFree(ctx->c.dense_land); ctx->c.dense_land = nullptr; }
    (void)hipStreamDestroy(ctx->c.stream);
    delete ctx;
    return DRE_OK;
}
const char* dre_last_error(dre_ctx* ctx) { return ctx ? ctx->c.last_error.c_str() : g_noctx_error.c_str(); }
int dre_ctx_sync(dre_ctx* ctx) { return guarded(ctx, [&] { ctx->c.sync(); }); }
int dre_ctx_info(dre_ctx* ctx, int64_t* info) {
    info[0] = ctx->c.num_cus; info[1] = (int64_t)ctx->c.pool.total_bytes();
    return DRE_OK;
}
// One table for dre_ctx_set_option / dre_ctx_get_option / DRE_OPTIONS: the option's storage by name (exactly one of i / d is set)
struct OptionRef { int* i = nullptr; double* d = nullptr; };
static OptionRef option_ref(Ctx& c, const std::string& key) {
    OptionRef r;
#define DRE_OPT_I(NAME, FIELD) if (key == NAME) { r.i = &c.FIELD; return r; }
#define DRE_OPT_D(NAME, FIELD) if (key == NAME) { r.d = &c.FIELD; return r; }
    DRE_OPT_I("dense_inverse_max_n", dense_inv_max_n)
    DRE_OPT_I("compress_direct_max_n", compress_direct_max_n)
    DRE_OPT_D("compress_direct_ratio", compress_direct_ratio)
    DRE_OPT_I("compress_factor_min_n", compress_factor_min_n)
    DRE_OPT_I("compress_factor_min_cols", compress_factor_min_cols)
    DRE_OPT_I("compress_sketch", compress_sketch)
    DRE_OPT_I("compress_sketch_min_cols", compress_sketch_min_cols)
    DRE_OPT_I("compress_sketch_extra", compress_sketch_extra)
    DRE_OPT_I("compress_sketch_cholqr", compress_sketch_cholqr)
    DRE_OPT_I("compress_sketch_sparse", compress_sketch_sparse)
    DRE_OPT_I("gemm_swizzle", gemm_swizzle)
    DRE_OPT_I("mf_swizzle", mf_swizzle)
    DRE_OPT_I("ros2_tight", ros2_tight)
    DRE_OPT_D("compress_sketch_ratio", compress_sketch_ratio)
    DRE_OPT_I("top_inverse_max_rows", top_inverse_max_rows)
    DRE_OPT_I("mf_subtree", mf_subtree)
    DRE_OPT_I("setup_streams", setup_streams)
    DRE_OPT_I("x_side_stream", x_side_stream)
    DRE_OPT_I("side_after_panels", side_after_panels)
    DRE_OPT_I("setup_batched", setup_batched)
    DRE_OPT_I("dense_warm", dense_warm)
    DRE_OPT_I("side_prefetch", side_prefetch)
    DRE_OPT_I("side_gate", side_gate)
    DRE_OPT_I("recurrence_wide", recurrence_wide)
    DRE_OPT_I("xwarm_sx", xwarm_sx)
    DRE_OPT_I("prefetch_batch", prefetch_batch)
    DRE_OPT_I("dense_x_max_n", dense_x_max_n)
    DRE_OPT_I("dense_x_max_k", dense_x_max_k)
    DRE_OPT_I("adi_group", adi_group)
    DRE_OPT_I("adi_group_max_n", adi_group_max_n)
    DRE_OPT_I("adi_fan", adi_fan)
    DRE_OPT_I("ros1_recurrence", ros1_recurrence)
    DRE_OPT_D("adi_fan_max_coef", adi_fan_max_coef)
    DRE_OPT_I("shard_min_cols", shard_min_cols)
    DRE_OPT_I("comm_host_async", comm_host_async)
    DRE_OPT_I("x_compress_every", x_compress_every)
    DRE_OPT_D("pivot_growth_warn", pivot_growth_warn)
    DRE_OPT_D("pivot_growth_fail", pivot_growth_fail)
    DRE_OPT_D("pivot_static", pivot_static)
    DRE_OPT_I("pivot_refine_steps", pivot_refine_steps)
#undef DRE_OPT_I
#undef DRE_OPT_D
    return r;
}
int dre_ctx_set_option(dre_ctx* ctx, const char* name, double value) {
    return guarded(ctx, [&] {
        const std::string key = name ? name : "";
        if (!(value == value)) throw Error(ERR_INVALID, "dre_ctx_set_option: the value of '" + key + "' is not a number");
        if (key == "shard_emulate") {
            // ONE process plays `value` ranks of the column-sharded ADI step one after the other (tests of the blocking logic on one GPU)
            if (!ctx->c.comm) ctx->c.comm = std::make_shared<Comm>();
            ctx->c.comm->emulate = (int)value;
            return;
        }
        const OptionRef r = option_ref(ctx->c, key);
        if (r.i) *r.i = (int)value;
        else if (r.d) *r.d = value;
        else throw Error(ERR_INVALID, "dre_ctx_set_option: unknown option '" + key + "'");
    });
}
int dre_ctx_get_option(dre_ctx* ctx, const char* name, double* value) {
    return guarded(ctx, [&] {
        const std::string key = name ? name : "";
        if (!value) throw Error(ERR_INVALID, "dre_ctx_get_option: null output");
        if (key == "shard_emulate") { *value = ctx->c.comm ? (double)ctx->c.comm->emulate : 0.0; return; }
        const OptionRef r = option_ref(ctx->c, key);
        if (r.i) *value = (double)*r.i;
        else if (r.d) *value = *r.d;
        else throw Error(ERR_INVALID, "dre_ctx_get_option: unknown option '" + key + "'");
    });
}
// Both contexts of a library context are timed (the side context carries the work that runs beside the main stream): enable/reset act on
// both, count/get see the per-class sums.
// every context that carries work of this library context: main, side, and the helper contexts of both
static void for_each_timed(dre_ctx* ctx, const std::function<void(Ctx&)>& f) {
    auto visit = [&](Ctx& c) { f(c); for (auto& h : c.helpers) if (h && h->timer) f(*h); };
    visit(ctx->c);
    if (ctx->c.side && ctx->c.side->timer) visit(*ctx->c.side);
}
static void prof_collect_merge(dre_ctx* ctx) {
    ctx->merged.clear();
    for_each_timed(ctx, [&](Ctx& c) {
        c.timer->collect(&c);
        for (auto& kv : c.timer->stats) {
            auto& d = ctx->merged[kv.first];
            d.ms += kv.second.ms; d.launches += kv.second.launches; d.bytes += kv.second.bytes; d.flops += kv.second.flops;
        }
    });
}
int dre_prof_enable(dre_ctx* ctx, int on) {
    return guarded(ctx, [&] {
        ctx->c.prof_side = on != 0;
        for_each_timed(ctx, [&](Ctx& c) { c.timer->collect(&c); c.timer->enabled = on != 0; });
    });
}
int dre_prof_reset(dre_ctx* ctx) {
    return guarded(ctx, [&] {
        for_each_timed(ctx, [&](Ctx& c) { c.timer->collect(&c); c.timer->stats.clear(); });
        ctx->merged.clear();
    });
}
int dre_prof_count(dre_ctx* ctx, int* n) {
    return guarded(ctx, [&] { prof_collect_merge(ctx); *n = (int)ctx->merged.size(); });
}
int dre_prof_get(dre_ctx* ctx, int i, char* name, int name_len, double* ms, int64_t* launches, double* bytes, double* flops) {
    return guarded(ctx, [&] {
        auto& st = ctx->merged;
        DRE_REQUIRE(i >= 0 && i < (int)st.size(), "dre_prof_get: index out of range");
        auto it = st.begin();
        std::advance(it, i);
        std::snprintf(name, name_len, "%s", it->first.c_str());
        *ms = it->second.ms; *launches = it->second.launches; *bytes = it->second.bytes; *flops = it->second.flops;
    });
}

// ---- dense -------------------------------------------------------------------------------------
int dre_dense_upload(dre_ctx* ctx, int rows, int cols, const double* host, int ld, dre_dense** out) {
    return guarded(ctx, [&] {
        DRE_REQUIRE(rows >= 0 && cols >= 0 && ld >= std::max(rows, 1), "dre_dense_upload: bad shape");
        auto* d = new dre_dense();
        d->m = Mat(&ctx->c, rows, cols);
        if (rows > 0 && cols > 0) {
            DRE_HIP(hipMemcpy2DAsync(d->m.p, (size_t)d->m.ld * sizeof(double), host, (size_t)ld * sizeof(double),
                                     (size_t)rows * sizeof(double), cols, hipMemcpyHostToDevice, ctx->c.stream));
            DRE_HIP(hipStreamSynchronize(ctx->c.stream));
        }
        *out = d;
    });
}
int dre_dense_create(dre_ctx* ctx, int rows, int cols, dre_dense** out) {
    return guarded(ctx, [&] {
        DRE_REQUIRE(rows >= 0 && cols >= 0, "dre_dense_create: bad shape");
        auto* d = new dre_dense();
        d->m = Mat(&ctx->c, rows, cols);
        fill_mat(&ctx->c, d->m, 0.0);
        *out = d;
    });
}
static void download_mat(Ctx* c, const Mat& m, double* host, int ld) {
    DRE_REQUIRE(ld >= std::max(m.rows, 1), "download: leading dimension too small");
    if (m.rows > 0 && m.cols > 0) {
        DRE_HIP(hipMemcpy2DAsync(host, (size_t)ld * sizeof(double), m.p, (size_t)m.ld * sizeof(double),
                                 (size_t)m.rows * sizeof(double), m.cols, hipMemcpyDeviceToHost, c->stream));
    }
    DRE_HIP(hipStreamSynchronize(c->stream));
}
int dre_dense_download(dre_ctx* ctx, const dre_dense* a, double* host, int ld) {
    return guarded(ctx, [&] { download_mat(&ctx->c, a->m, host, ld); });
}
// device-to-device interop with buffers the caller owns (e.g. the exchange tensors handed to RCCL): column-major, leading dimension ld
int dre_dense_from_device(dre_ctx* ctx, int rows, int cols, const double* src_dev, int ld, dre_dense** out) {
    return guarded(ctx, [&] {
        DRE_REQUIRE(rows >= 0 && cols >= 0 && ld >= std::max(rows, 1), "dre_dense_from_device: bad shape");
        auto* d = new dre_dense();
        d->m = Mat(&ctx->c, rows, cols);
        if (rows > 0 && cols > 0)
            DRE_HIP(hipMemcpy2DAsync(d->m.p, (size_t)d->m.ld * sizeof(double), src_dev, (size_t)ld * sizeof(double), (size_t)rows * sizeof(double), cols,
                                     hipMemcpyDeviceToDevice, ctx->c.stream));
        DRE_HIP(hipStreamSynchronize(ctx->c.stream));
        *out = d;
    });
}
int dre_dense_to_device(dre_ctx* ctx, const dre_dense* a, double* dst_dev, int ld) {
    return guarded(ctx, [&] {
        DRE_REQUIRE(ld >= std::max(a->m.rows, 1), "dre_dense_to_device: bad leading dimension");
        if (a->m.rows > 0 && a->m.cols > 0)
            DRE_HIP(hipMemcpy2DAsync(dst_dev, (size_t)ld * sizeof(double), a->m.p, (size_t)a->m.ld * sizeof(double), (size_t)a->m.rows * sizeof(double), a->m.cols,
                                     hipMemcpyDeviceToDevice, ctx->c.stream));
        DRE_HIP(hipStreamSynchronize(ctx->c.stream));
    });
}
int dre_dense_shape(const dre_dense* a, int* rows, int* cols) { *rows = a->m.rows; *cols = a->m.cols; return DRE_OK; }
int dre_dense_free(dre_ctx*, dre_dense* a) { delete a; return DRE_OK; }

// ---- pencil ------------------------------------------------------------------------------------
static int pencil_create_impl(dre_ctx* ctx, bool upload, int n, const int64_t* Ep, const int64_t* Ei, const double* Ev,
                              const int64_t* Ap, const int64_t* Ai, const double* Av, int base, int leaf, dre_pencil** out) {
    return guarded(ctx, [&] {
        auto* h = new dre_pencil();
        try {
            h->p = pencil_create(ctx ? &ctx->c : nullptr, n, Ep, Ei, Ev, Ap, Ai, Av, base, leaf, upload, &h->hE, &h->hA);
        } catch (...) { delete h; throw; }
        *out = h;
    });
}
int dre_pencil_create(dre_ctx* ctx, int n, const int64_t* Ep, const int64_t* Ei, const double* Ev, const int64_t* Ap,
                      const int64_t* Ai, const double* Av, int base, int leaf, dre_pencil** out) {
    return pencil_create_impl(ctx, true, n, Ep, Ei, Ev, Ap, Ai, Av, base, leaf, out);
}
int dre_pencil_create_host(int n, const int64_t* Ep, const int64_t* Ei, const double* Ev, const int64_t* Ap,
                           const int64_t* Ai, const double* Av, int base, int leaf, dre_pencil** out) {
    return pencil_create_impl(nullptr, false, n, Ep, Ei, Ev, Ap, Ai, Av, base, leaf, out);
}
int dre_pencil_free(dre_pencil* p) { delete p; return DRE_OK; }
int dre_pencil_info(const dre_pencil* p, int64_t* info) {
    const Symbolic& S = p->p->sym;
    info[0] = S.n; info[1] = (int64_t)S.idx.size(); info[2] = S.nnodes; info[3] = S.nlevels; info[4] = S.max_front;
    info[5] = S.max_sep; info[6] = S.factor_nnz; info[7] = S.fronts_size;
    return DRE_OK;
}
int dre_pencil_get_array(const dre_pencil* p, const char* name, int64_t* out, int64_t cap, int64_t* len) {
    const Symbolic& S = p->p->sym;
    std::string nm(name);
    auto put_i = [&](const std::vector<int>& v) { *len = (int64_t)v.size(); if (out) for (int64_t i = 0; i < std::min<int64_t>(cap, v.size()); ++i) out[i] = v[i]; return DRE_OK; };
    auto put_l = [&](const std::vector<int64_t>& v) { *len = (int64_t)v.size(); if (out) for (int64_t i = 0; i < std::min<int64_t>(cap, v.size()); ++i) out[i] = v[i]; return DRE_OK; };
    if (nm == "perm") return put_i(S.perm);
    if (nm == "iperm") return put_i(S.iperm);
    if (nm == "ptr") return put_i(S.ptr);
    if (nm == "idx") return put_i(S.idx);
    if (nm == "first") return put_i(S.first);
    if (nm == "size") return put_i(S.size);
    if (nm == "parent") return put_i(S.parent);
    if (nm == "level") return put_i(S.level);
    if (nm == "child_ptr") return put_i(S.child_ptr);
    if (nm == "child_idx") return put_i(S.child_idx);
    if (nm == "bptr") return put_i(S.bptr);
    if (nm == "bidx") return put_i(S.bidx);
    if (nm == "cmap_ptr") return put_i(S.cmap_ptr);
    if (nm == "cmap") return put_i(S.cmap);
    if (nm == "lvl_ptr") return put_i(S.lvl_ptr);
    if (nm == "lvl_nodes") return put_i(S.lvl_nodes);
    if (nm == "front_off") return put_l(S.front_off);
    if (nm == "inv_off") return put_l(S.inv_off);
    if (nm == "upd_off") return put_l(S.upd_off);
    if (nm == "asm_dest") return put_l(S.asm_dest);
    return DRE_ERR_INVALID;
}
int dre_pencil_get_values(const dre_pencil* p, int which, double* out, int64_t cap) {
    const std::vector<double>& v = which == 0 ? p->hE : p->hA;
    if ((int64_t)v.size() > cap) return DRE_ERR_INVALID;
    std::memcpy(out, v.data(), v.size() * sizeof(double));
    return DRE_OK;
}

// helpers: rows of user-ordered matrices <-> solver ordering
static Mat to_solver_order(Ctx* c, const dre_pencil* pen, const Mat& src) {
    if (!pen) return src;
    DRE_REQUIRE(src.rows == pen->p->n, "matrix row count does not match the pencil");
    Mat dst(c, src.rows, src.cols);
    permute_rows(c, src, pen->p->perm.p, dst);     // dst(new,:) = src(perm[new],:)
    return dst;
}
static Mat to_user_order(Ctx* c, const dre_pencil* pen, const Mat& src) {
    if (!pen) return src;
    Mat dst(c, src.rows, src.cols);
    permute_rows(c, src, pen->p->iperm.p, dst);    // dst(old,:) = src(iperm[old],:)
    return dst;
}

// ---- kernels -----------------------------------------------------------------------------------
int dre_gemm(dre_ctx* ctx, int tA, int tB, double alpha, const dre_dense* A, const dre_dense* B, double beta, dre_dense* C) {
    return guarded(ctx, [&] { gemm(&ctx->c, tA != 0, tB != 0, alpha, A->m, B->m, beta, C->m); });
}
int dre_spmm(dre_ctx* ctx, const dre_pencil* p, int which, double alpha, const dre_dense* X, double beta, dre_dense* Y) {
    return guarded(ctx, [&] {
        const Pencil& P = *p->p;
        DRE_REQUIRE(P.has_device, "pencil was created host-only");
        Mat Xs = to_solver_order(&ctx->c, p, X->m);
        Mat Ys = to_solver_order(&ctx->c, p, Y->m);
        spmm(&ctx->c, P, which == 0 ? P.valEt.p : P.valAt.p, Xs, Ys, alpha, beta);
        Mat Yu = to_user_order(&ctx->c, p, Ys);
        copy_mat(&ctx->c, Yu, Y->m);
    });
}
int dre_ctx_set_orthf(dre_ctx* ctx, dre_orthf_fn fn, void* user) {
    return guarded(ctx, [&] { ctx->c.orthf_fn = fn; ctx->c.orthf_user = user; if (ctx->c.side) { ctx->c.side->orthf_fn = fn; ctx->c.side->orthf_user = user; } });
}
int dre_orthf(dre_ctx* ctx, const dre_dense* L, dre_dense** Qo, dre_dense** Ro) {
    return guarded(ctx, [&] {
        Ctx* c = &ctx->c;
        Mat A(c, L->m.rows, L->m.cols);
        copy_mat(c, L->m, A);
        QRFact f = qr_factor(c, A);
        auto* Q = new dre_dense();
        Q->m = Mat(c, f.m, f.kq);
        set_identity(c, Q->m, 1.0);
        qr_apply_q(c, f, Q->m, false);
        auto* R = new dre_dense();
        R->m = f.R;
        *Qo = Q; *Ro = R;
    });
}
int dre_sym_eig(dre_ctx* ctx, const dre_dense* S, double tolfac, dre_dense** values, dre_dense** vectors) {
    return guarded(ctx, [&] {
        Ctx* c = &ctx->c;
        Mat A(c, S->m.rows, S->m.cols);
        copy_mat(c, S->m, A);
        SymEig e = sym_eig(c, A, tolfac > 0 ? tolfac : 4.0);
        std::vector<int> ids(e.j);
        for (int i = 0; i < e.j; ++i) ids[i] = i;
        std::sort(ids.begin(), ids.end(), [&](int a, int b) { return e.w[a] < e.w[b]; });
        std::vector<double> w(e.j);
        for (int i = 0; i < e.j; ++i) w[i] = e.w[ids[i]];
        auto* V = new dre_dense();
        V->m = sym_eig_backtransform(c, e, ids);
        auto* W = new dre_dense();
        W->m = Mat(c, e.j, 1);
        if (e.j) {
            DRE_HIP(hipMemcpyAsync(W->m.p, w.data(), e.j * sizeof(double), hipMemcpyHostToDevice, c->stream));
            DRE_HIP(hipStreamSynchronize(c->stream));
        }
        *values = W; *vectors = V;
    });
}
int dre_shift_factor(dre_ctx* ctx, const dre_pencil* p, double cA, double cE_re, double cE_im, dre_factor** out) {
    return guarded(ctx, [&] {
        const Pencil& P = *p->p;
        DRE_REQUIRE(P.has_device, "pencil was created host-only");
        auto* f = new dre_factor();
        f->pen = p;
        f->is_cplx = cE_im != 0.0;
        try {
            if (f->is_cplx) { mf_factor<cplx>(&ctx->c, P, P.valAt.p, P.valEt.p, cplx{cA, 0.0}, cplx{cE_re, cE_im}, f->fc); mf_check(&ctx->c, f->fc); }
            else { mf_factor<double>(&ctx->c, P, P.valAt.p, P.valEt.p, cA, cE_re, f->fr); mf_check(&ctx->c, f->fr); }
        } catch (...) { delete f; throw; }
        *out = f;
    });
}

__global__ void k_split_cplx(int rows, int cols, const cplx* __restrict__ src, int lds_, double* __restrict__ re, int ldr,
                             double* __restrict__ im, int ldi) {
    size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= (size_t)rows * cols) return;
    int r = id % rows, c = id / rows;
    cplx z = src[r + (size_t)c * lds_];
    re[r + (size_t)c * ldr] = z.re;
    im[r + (size_t)c * ldi] = z.im;
}
__global__ void k_make_cplx(int rows, int cols, const double* __restrict__ src, int lds_, cplx* __restrict__ dst, int ldd) {
    size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= (size_t)rows * cols) return;
    int r = id % rows, c = id / rows;
    dst[r + (size_t)c * ldd] = {src[r + (size_t)c * lds_], 0.0};
}

int dre_shift_solve(dre_ctx* ctx, const dre_factor* f, const dre_dense* B, dre_dense** X_re, dre_dense** X_im) {
    return guarded(ctx, [&] {
        Ctx* c = &ctx->c;
        const Pencil& P = *f->pen->p;
        Mat Bs = to_solver_order(c, f->pen, B->m);
        const int n = P.n, k = Bs.cols;
        if (!f->is_cplx) {
            Mat W(c, n, k);
            copy_mat(c, Bs, W);
            mf_solve<double>(c, P, f->fr, W.p, W.ld, k);
            auto* X = new dre_dense();
            X->m = Mat(c, n, k);
            Mat Xu = to_user_order(c, f->pen, W);
            copy_mat(c, Xu, X->m);
            *X_re = X;
            if (X_im) *X_im = nullptr;
        } else {
            DRE_REQUIRE(X_im != nullptr, "complex factor needs an imaginary output");
            DevArr<cplx> W(c, (size_t)n * std::max(k, 1));
            size_t tot = (size_t)n * k;
            if (tot) hipLaunchKernelGGL(k_make_cplx, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, c->stream, n, k, Bs.p, Bs.ld, W.p, n);
            mf_solve<cplx>(c, P, f->fc, W.p, n, k);
            Mat re(c, n, k), im(c, n, k);
            if (tot) hipLaunchKernelGGL(k_split_cplx, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, c->stream, n, k, W.p, n, re.p, re.ld, im.p, im.ld);
            auto* Xr = new dre_dense(); auto* Xi = new dre_dense();
            Xr->m = Mat(c, n, k); Xi->m = Mat(c, n, k);
            Mat ru = to_user_order(c, f->pen, re), iu = to_user_order(c, f->pen, im);
            copy_mat(c, ru, Xr->m); copy_mat(c, iu, Xi->m);
            c->sync();
            *X_re = Xr; *X_im = Xi;
        }
    });
}
int dre_shift_solve_smw(dre_ctx* ctx, const dre_factor* f, double alpha, const dre_dense* U, const dre_dense* Vt, const dre_dense* B,
                        dre_dense** X_re, dre_dense** X_im) {
    return guarded(ctx, [&] {
        Ctx* c = &ctx->c;
        const Pencil& P = *f->pen->p;
        DRE_REQUIRE(U && Vt && B && X_re, "dre_shift_solve_smw: null argument");
        Mat Us = to_solver_order(c, f->pen, U->m), Vs = to_solver_order(c, f->pen, Vt->m), Bs = to_solver_order(c, f->pen, B->m);
        if (!f->is_cplx) {
            Mat X = smw_solve(c, P, f->fr, alpha, Us, Vs, Bs);
            auto* Xr = new dre_dense();
            Xr->m = Mat(c, P.n, Bs.cols);
            Mat Xu = to_user_order(c, f->pen, X);
            copy_mat(c, Xu, Xr->m);
            c->sync();
            *X_re = Xr;
            if (X_im) *X_im = nullptr;
        } else {
            DRE_REQUIRE(X_im != nullptr, "complex factor needs an imaginary output");
            Mat re, im;
            smw_solve(c, P, f->fc, alpha, Us, Vs, Bs, re, im);
            auto* Xr = new dre_dense(); auto* Xi = new dre_dense();
            Xr->m = Mat(c, P.n, Bs.cols); Xi->m = Mat(c, P.n, Bs.cols);
            Mat ru = to_user_order(c, f->pen, re), iu = to_user_order(c, f->pen, im);
            copy_mat(c, ru, Xr->m); copy_mat(c, iu, Xi->m);
            c->sync();
            *X_re = Xr; *X_im = Xi;
        }
    });
}
int dre_factor_growth(dre_ctx* ctx, const dre_factor* f, double* growth) {
    return guarded(ctx, [&] { *growth = f->is_cplx ? mf_check(&ctx->c, f->fc) : mf_check(&ctx->c, f->fr); });
}
int dre_factor_perturbed(dre_ctx* ctx, const dre_factor* f, int64_t* count) {
    return guarded(ctx, [&] {
        if (f->is_cplx) { if (f->fc.nperturbed < 0) mf_check(&ctx->c, f->fc); *count = f->fc.nperturbed; }
        else { if (f->fr.nperturbed < 0) mf_check(&ctx->c, f->fr); *count = f->fr.nperturbed; }
    });
}
int dre_factor_free(dre_ctx*, dre_factor* f) { delete f; return DRE_OK; }

// ---- LDLt --------------------------------------------------------------------------------------
int dre_ldlt_create(dre_ctx* ctx, const dre_pencil* p, const dre_dense* L, const dre_dense* D, double alpha, dre_ldlt** out) {
    return guarded(ctx, [&] {
        Ctx* c = &ctx->c;
        DRE_REQUIRE(D->m.rows == D->m.cols && D->m.rows == L->m.cols, "lowrank: D must be k x k with k = size(L, 2)");
        Mat Ls = to_solver_order(c, p, L->m);
        if (!p) { Ls = Mat(c, L->m.rows, L->m.cols); copy_mat(c, L->m, Ls); }
        Mat Dc(c, D->m.rows, D->m.cols);
        copy_mat(c, D->m, Dc);
        auto* h = new dre_ldlt();
        h->x = ldlt_make(c, L->m.rows, Ls, Dc, alpha, false);
        h->pen = p;
        *out = h;
    });
}
int dre_ldlt_zero(dre_ctx* ctx, const dre_pencil* p, int n, dre_ldlt** out) {
    return guarded(ctx, [&] { auto* h = new dre_ldlt(); h->x = ldlt_zero(n); h->pen = p; *out = h; });
}
int dre_ldlt_free(dre_ctx*, dre_ldlt* x) { delete x; return DRE_OK; }
int dre_ldlt_info(const dre_ldlt* x, int* n, int* rank, int* nblocks) {
    *n = x->x->n; *rank = x->x->rank(); *nblocks = (int)x->x->blocks.size();
    return DRE_OK;
}
int dre_ldlt_add(dre_ctx* ctx, const dre_ldlt* a, const dre_ldlt* b, dre_ldlt** out) {
    return guarded(ctx, [&] {
        DRE_REQUIRE(a->pen == b->pen, "LDLt operands live in different orderings");
        auto* h = new dre_ldlt();
        h->x = ldlt_add(a->x, b->x);
        if (h->x.get() == a->x.get() || h->x.get() == b->x.get()) h->x = std::make_shared<LDLt>(*h->x);
        h->pen = a->pen;
        *out = h;
    });
}
int dre_ldlt_scale(dre_ctx* ctx, const dre_ldlt* a, double alpha, dre_ldlt** out) {
    return guarded(ctx, [&] { auto* h = new dre_ldlt(); h->x = ldlt_scale(a->x, alpha); h->pen = a->pen; *out = h; });
}
int dre_ldlt_concatenate(dre_ctx* ctx, dre_ldlt* x) { return guarded(ctx, [&] { ldlt_concatenate(&ctx->c, *x->x); }); }
int dre_ldlt_compress(dre_ctx* ctx, dre_ldlt* x) { return guarded(ctx, [&] { ldlt_compress(&ctx->c, *x->x); }); }
int dre_ldlt_compress_tol(dre_ctx* ctx, dre_ldlt* x, double abs_tol) {
    return guarded(ctx, [&] { ldlt_compress(&ctx->c, *x->x, 4.0, false, abs_tol > 0.0 ? abs_tol : -1.0); });
}
int dre_ldlt_compress_fast(dre_ctx* ctx, dre_ldlt* x) { return guarded(ctx, [&] { ldlt_compress(&ctx->c, *x->x, 4.0, false, -1.0); }); }
int dre_ldlt_canonicalize(dre_ctx* ctx, dre_ldlt* x) {
    return guarded(ctx, [&] {
        LDLt& X = *x->x;
        if (X.blocks.size() > 1 || (X.blocks.size() == 1 && !X.blocks[0].diag && X.blocks[0].L.cols > 0)) ldlt_compress(&ctx->c, X, 4.0, true);
    });
}
int dre_ldlt_norm(dre_ctx* ctx, dre_ldlt* x, double* out) { return guarded(ctx, [&] { *out = ldlt_norm_accurate(&ctx->c, *x->x); }); }
int dre_ldlt_destructure(dre_ctx* ctx, dre_ldlt* x, double* alpha, double* Lh, int ldl, double* Dh, int ldd) {
    return guarded(ctx, [&] {
        Ctx* c = &ctx->c;
        ldlt_destructure(c, *x->x);
        const LBlock& b = x->x->blocks[0];
        if (alpha) *alpha = b.alpha;
        if (Lh) { Mat Lu = to_user_order(c, x->pen, b.L); download_mat(c, Lu, Lh, ldl); }
        if (Dh) download_mat(c, b.D, Dh, ldd);
    });
}

// ---- GALE / ADI --------------------------------------------------------------------------------
int dre_adi_default_options(dre_adi_options* o) {
    std::memset(o, 0, sizeof(*o));
    o->maxiters = 100; o->reltol = -1.0; o->abstol = -1.0; o->ignore_initial_guess = 0; o->compression_interval = 10;
    o->compression = 1; o->shift_kind = 1; o->n_history = 2; o->nshifts = 0; o->shifts_re = nullptr; o->shifts_im = nullptr;
    o->compress_tolfac = 4.0;
    o->compress_exact = 0;
    o->heuristic_kplus = 0; o->heuristic_kminus = 0;
    o->shift_fn = nullptr; o->shift_user = nullptr;
    return DRE_OK;
}
static AdiOptions convert_options(const dre_adi_options* o) {
    AdiOptions a;
    if (!o) return a;
    a.maxiters = o->maxiters; a.reltol = o->reltol; a.abstol = o->abstol; a.ignore_initial_guess = o->ignore_initial_guess != 0;
    a.compression_interval = o->compression_interval; a.compression = o->compression != 0;
    a.compress_tolfac = o->compress_tolfac > 0 ? o->compress_tolfac : 4.0;
    a.compress_exact = o->compress_exact != 0;
    a.inner_solve = o->inner_solve; a.inner_user = o->inner_user;
    if (o->shift_kind == 0) {
        a.shifts.kind = ShiftSpec::CYCLIC;
        DRE_REQUIRE(o->nshifts > 0 && o->shifts_re, "Cyclic shifts need at least one value");
        for (int i = 0; i < o->nshifts; ++i) a.shifts.values.emplace_back(o->shifts_re[i], o->shifts_im ? o->shifts_im[i] : 0.0);
    } else if (o->shift_kind == 2) {
        a.shifts.kind = ShiftSpec::HEURISTIC;
        DRE_REQUIRE(o->nshifts > 0 && o->heuristic_kplus > 0 && o->heuristic_kminus > 0, "Heuristic(nshifts, k+, k-) must be positive");
        a.shifts.h_nshifts = o->nshifts; a.shifts.h_kplus = o->heuristic_kplus; a.shifts.h_kminus = o->heuristic_kminus;
    } else if (o->shift_kind == 3) {
        a.shifts.kind = ShiftSpec::USER;
        DRE_REQUIRE(o->shift_fn != nullptr, "shift_kind 3 (user-defined strategy) needs shift_fn");
        DRE_REQUIRE(o->n_history > 0, "shift_kind 3: n_history must be positive");
        a.shifts.user_fn = o->shift_fn; a.shifts.user_data = o->shift_user; a.shifts.n_history = o->n_history;
    } else {
        a.shifts.kind = ShiftSpec::PROJECTION;
        DRE_REQUIRE(o->n_history > 0 && o->n_history % 2 == 0, "History must be even");   // projection.jl:28-32
        a.shifts.n_history = o->n_history;
    }
    return a;
}
static std::atomic<uint64_t> g_tag_counter{1ull << 40};     // operator identities for the factor caches (unique across threads)
static GaleOperator make_operator(Ctx* c, const dre_pencil* p, double cA, double cE, double lr_alpha, const dre_dense* U, const dre_dense* Vt) {
    const Pencil& P = *p->p;
    DRE_REQUIRE(P.has_device, "pencil was created host-only");
    GaleOperator op;
    op.P = &P;
    op.valFt = DevArr<double>(c, P.nnz);
    vals_axpby(c, P.nnz, cA, P.valAt.p, cE, P.valEt.p, op.valFt.p);
    op.tag = g_tag_counter++;
    op.cA = cA; op.cE = cE;
    if (U && Vt) {
        DRE_REQUIRE(U->m.rows == P.n && Vt->m.rows == P.n && U->m.cols == Vt->m.cols, "low-rank factors must be n x m");
        op.has_lr = true; op.alpha = lr_alpha;
        op.U = to_solver_order(c, p, U->m);
        op.Vt = to_solver_order(c, p, Vt->m);
    }
    return op;
}
// The sketch of a wide factor (ldlt.hip sketch_compress) is chosen from what earlier compressions at the same order left in the context:
// the rank hint and the fallback counters.  A solve entered through the ABI starts from a clean slate, so that two identical calls return
// identical bits whatever ran on the context before (the explicit compression entry dre_ldlt_compress_fast keeps the history: that is
// its documented behaviour).
static void reset_sketch_history(Ctx* c, int n) {
    const long skey = -(4000000000L + (long)n);
    c->band_hint.erase(skey); c->band_hint.erase(skey - 1); c->band_hint.erase(skey - 2);
}
int dre_gale_solve(dre_ctx* ctx, const dre_pencil* p, double cA, double cE, double lr_alpha, const dre_dense* U, const dre_dense* Vt,
                   dre_ldlt* C, const dre_ldlt* X0, const dre_adi_options* opt, dre_adi_result** out) {
    return guarded(ctx, [&] {
        Ctx* c = &ctx->c;
        DRE_REQUIRE(C->pen == p && (!X0 || X0->pen == p), "LDLt operands must be created with the same pencil");
        GaleOperator op = make_operator(c, p, cA, cE, lr_alpha, U, Vt);
        AdiOptions ao = convert_options(opt);
        auto* r = new dre_adi_result();
        r->pen = p;
        reset_sketch_history(c, p->p->n);
        try { r->r = adi_solve(c, op, *C->x, X0 ? X0->x : nullptr, ao, nullptr); } catch (...) { delete r; throw; }
        *out = r;
    });
}
// ---- stepwise protocol: init / step! / isdone / solve! on the solver object (src/lyapunov/adi.jl:29-141) -------------------------
struct dre_adi_solver {
    std::shared_ptr<AdiRun> run;
    const dre_pencil* pen = nullptr;
};
int dre_adi_init(dre_ctx* ctx, const dre_pencil* p, double cA, double cE, double lr_alpha, const dre_dense* U, const dre_dense* Vt,
                 dre_ldlt* C, const dre_ldlt* X0, const dre_adi_options* opt, dre_adi_solver** out) {
    return guarded(ctx, [&] {
        Ctx* c = &ctx->c;
        DRE_REQUIRE(C->pen == p && (!X0 || X0->pen == p), "LDLt operands must be created with the same pencil");
        GaleOperator op = make_operator(c, p, cA, cE, lr_alpha, U, Vt);
        AdiOptions ao = convert_options(opt);
        auto* s = new dre_adi_solver();
        s->pen = p;
        reset_sketch_history(c, p->p->n);
        try { s->run = adi_begin(c, op, *C->x, X0 ? X0->x : nullptr, ao, nullptr); } catch (...) { delete s; throw; }
        *out = s;
    });
}
int dre_adi_step(dre_ctx* ctx, dre_adi_solver* s) { return guarded(ctx, [&] { adi_advance(*s->run, 1); }); }
int dre_adi_solve(dre_ctx* ctx, dre_adi_solver* s) {
    return guarded(ctx, [&] { while (!adi_isdone(*s->run)) adi_advance(*s->run, 1 << 30); });
}
int dre_adi_isdone(const dre_adi_solver* s, int* done) { *done = adi_isdone(*s->run) ? 1 : 0; return DRE_OK; }
int dre_adi_state(const dre_adi_solver* s, int64_t* iters, double* res_norm, double* abstol) {
    int it = 0; double rn = 0, at = 0;
    adi_peek(*s->run, &it, &rn, &at);
    if (iters) *iters = it;
    if (res_norm) *res_norm = rn;
    if (abstol) *abstol = at;
    return DRE_OK;
}
int dre_adi_shifts(const dre_adi_solver* s, int64_t from, int64_t* count, double* re, double* im) {
    auto v = adi_shifts_since(*s->run, (int)from);
    if (count) *count = (int64_t)v.size();
    for (size_t i = 0; i < v.size(); ++i) { if (re) re[i] = v[i].real(); if (im) im[i] = v[i].imag(); }
    return DRE_OK;
}
int dre_adi_snapshot(dre_ctx* ctx, dre_adi_solver* s, dre_ldlt** X, dre_ldlt** residual) {
    return guarded(ctx, [&] {
        LDLtP x, r;
        adi_snapshot(*s->run, X ? &x : nullptr, residual ? &r : nullptr);
        if (X) { auto* h = new dre_ldlt(); h->x = x; h->pen = s->pen; *X = h; }
        if (residual) { auto* h = new dre_ldlt(); h->x = r; h->pen = s->pen; *residual = h; }
    });
}
int dre_adi_finish(dre_ctx* ctx, dre_adi_solver* s, dre_adi_result** out) {
    return guarded(ctx, [&] {
        auto* r = new dre_adi_result();
        r->pen = s->pen;
        try { r->r = adi_finish(*s->run); } catch (...) { delete r; throw; }
        *out = r;
    });
}
int dre_adi_free(dre_adi_solver* s) { delete s; return DRE_OK; }

int dre_heuristic_ritz(dre_ctx* ctx, const dre_pencil* p, double cA, double cE, double lr_alpha, const dre_dense* U, const dre_dense* Vt,
                       int kplus, int kminus, double* plus_re, double* plus_im, double* minus_re, double* minus_im) {
    return guarded(ctx, [&] {
        Ctx* c = &ctx->c;
        GaleOperator op = make_operator(c, p, cA, cE, lr_alpha, U, Vt);
        std::vector<std::complex<double>> rp, rm;
        heuristic_ritz(c, op, kplus, kminus, rp, rm);
        for (int i = 0; i < kplus; ++i) { plus_re[i] = rp[i].real(); plus_im[i] = rp[i].imag(); }
        for (int i = 0; i < kminus; ++i) { minus_re[i] = rm[i].real(); minus_im[i] = rm[i].imag(); }
    });
}
int dre_gale_residual(dre_ctx* ctx, const dre_pencil* p, double cA, double cE, double lr_alpha, const dre_dense* U, const dre_dense* Vt,
                      dre_ldlt* C, dre_ldlt* X, dre_ldlt** out) {
    return guarded(ctx, [&] {
        Ctx* c = &ctx->c;
        GaleOperator op = make_operator(c, p, cA, cE, lr_alpha, U, Vt);
        auto* h = new dre_ldlt();
        h->pen = p;
        try { h->x = gale_residual(c, op, *C->x, X ? X->x : nullptr); } catch (...) { delete h; throw; }
        *out = h;
    });
}
int dre_ldlt_dot(dre_ctx* ctx, const dre_ldlt* a, const dre_ldlt* b, double* out) {
    return guarded(ctx, [&] {
        DRE_REQUIRE(a->pen == b->pen, "dot: both operands must be created with the same pencil (same row ordering)");
        *out = ldlt_dot(&ctx->c, *a->x, *b->x);
    });
}
int dre_gale_apply(dre_ctx* ctx, const dre_pencil* p, double cA, double cE, double lr_alpha, const dre_dense* U, const dre_dense* Vt,
                   const dre_ldlt* X, dre_ldlt** out) {
    return guarded(ctx, [&] {
        Ctx* c = &ctx->c;
        DRE_REQUIRE(X->pen == p, "LDLt operand must be created with the same pencil");
        GaleOperator op = make_operator(c, p, cA, cE, lr_alpha, U, Vt);
        auto* h = new dre_ldlt();
        h->pen = p;
        try { h->x = lyapunov_apply(c, op, X->x); } catch (...) { delete h; throw; }
        *out = h;
    });
}
int dre_gare_residual(dre_ctx* ctx, const dre_pencil* p, dre_ldlt* X, const dre_dense* Ct, const dre_dense* S, double gamma, const dre_dense* B,
                      const dre_dense* Rinv, double beta, dre_ldlt** out) {
    return guarded(ctx, [&] {
        Ctx* c = &ctx->c;
        DRE_REQUIRE(X->pen == p, "LDLt operand must be created with the same pencil");
        DRE_REQUIRE(Ct->m.rows == p->p->n && B->m.rows == p->p->n && S->m.rows == Ct->m.cols && Rinv->m.rows == B->m.cols, "dre_gare_residual: shape mismatch");
        Mat Cs = to_solver_order(c, p, Ct->m), Bs = to_solver_order(c, p, B->m);
        auto* h = new dre_ldlt();
        h->pen = p;
        try { h->x = gare_residual_dev(c, *p->p, *X->x, Cs, S->m, gamma, Bs, Rinv->m, beta); } catch (...) { delete h; throw; }
        *out = h;
    });
}
int dre_ldlt_feedback(dre_ctx* ctx, const dre_pencil* p, dre_ldlt* X, const dre_dense* B, dre_dense** Kt) {
    return guarded(ctx, [&] {
        Ctx* c = &ctx->c;
        DRE_REQUIRE(X->pen == p && B->m.rows == p->p->n, "dre_ldlt_feedback: operand mismatch");
        Mat Bs = to_solver_order(c, p, B->m);
        Mat Ks = ldlt_feedback_dev(c, *p->p, *X->x, Bs);
        auto* K = new dre_dense();
        Mat Ku = to_user_order(c, p, Ks);
        K->m = Mat(c, Ku.rows, Ku.cols);
        copy_mat(c, Ku, K->m);
        *Kt = K;
    });
}
int dre_adi_result_info(const dre_adi_result* r, int64_t* info, double* dinfo) {
    info[0] = r->r.iters; info[1] = r->r.converged; info[2] = r->r.warnings; info[3] = (int64_t)r->r.norms.size(); info[4] = r->r.rhs_cols;
    dinfo[0] = r->r.res_norm; dinfo[1] = r->r.abstol; dinfo[2] = r->r.initial_norm;
    return DRE_OK;
}
int dre_adi_result_history(const dre_adi_result* r, double* norms, int32_t* norm_iters, double* sre, double* sim) {
    for (size_t i = 0; i < r->r.norms.size(); ++i) { if (norms) norms[i] = r->r.norms[i]; if (norm_iters) norm_iters[i] = r->r.norm_iters[i]; }
    for (size_t i = 0; i < r->r.shifts.size(); ++i) { if (sre) sre[i] = r->r.shifts[i].real(); if (sim) sim[i] = r->r.shifts[i].imag(); }
    return DRE_OK;
}
int dre_adi_result_take_x(dre_adi_result* r, dre_ldlt** X) {
    auto* h = new dre_ldlt(); h->x = r->r.X; h->pen = r->pen; *X = h; return DRE_OK;
}
int dre_adi_result_take_residual(dre_adi_result* r, dre_ldlt** R) {
    auto* h = new dre_ldlt(); h->x = r->r.residual; h->pen = r->pen; *R = h; return DRE_OK;
}
int dre_adi_result_free(dre_adi_result* r) { delete r; return DRE_OK; }

// ---- communicator (RCCL over xGMI, comm.hip) ----------------------------------------------------
int dre_comm_unique_id(dre_ctx* ctx, void* id128) { return guarded(ctx, [&] { DRE_REQUIRE(id128, "null id buffer"); comm_unique_id(id128); }); }
int dre_comm_init(dre_ctx* ctx, int nranks, int rank, const void* id128) {
    return guarded(ctx, [&] {
        const int emu = ctx->c.comm ? ctx->c.comm->emulate : 0;
        ctx->c.comm = comm_init(&ctx->c, nranks, rank, id128);
        ctx->c.comm->emulate = emu;
    });
}
int dre_comm_init_host(dre_ctx* ctx, int nranks, int rank, dre_comm_allgather_fn allgather, dre_comm_allreduce_fn allreduce, void* user) {
    return guarded(ctx, [&] {
        const int emu = ctx->c.comm ? ctx->c.comm->emulate : 0;
        ctx->c.comm = comm_init_host(&ctx->c, nranks, rank, allgather, allreduce, user);
        ctx->c.comm->emulate = emu;
    });
}
int dre_comm_free(dre_ctx* ctx) {
    return guarded(ctx, [&] { if (ctx->c.comm) { DRE_HIP(hipStreamSynchronize(ctx->c.stream)); ctx->c.comm.reset(); } });
}
int dre_comm_info(dre_ctx* ctx, int64_t* info) {
    const Comm* c = ctx->c.comm.get();
    info[0] = c ? c->nranks : 1; info[1] = c ? c->rank : 0; info[2] = c ? (int64_t)c->ncalls : 0;
    info[3] = c ? (int64_t)c->bytes_gathered : 0; info[4] = c ? (int64_t)c->bytes_reduced : 0; info[5] = c ? c->emulate : 0;
    return DRE_OK;
}
int dre_comm_allgather(dre_ctx* ctx, const void* send_dev, void* recv_dev, size_t count) {
    return guarded(ctx, [&] {
        DRE_REQUIRE(ctx->c.comm, "dre_comm_allgather: no communicator (dre_comm_init)");
        comm_allgather(&ctx->c, *ctx->c.comm, (const double*)send_dev, (double*)recv_dev, count);
    });
}
int dre_comm_allreduce_sum(dre_ctx* ctx, void* buf_dev, size_t count) {
    return guarded(ctx, [&] {
        DRE_REQUIRE(ctx->c.comm, "dre_comm_allreduce_sum: no communicator (dre_comm_init)");
        comm_allreduce_sum(&ctx->c, *ctx->c.comm, (double*)buf_dev, count);
    });
}

// ---- GDRE --------------------------------------------------------------------------------------
int dre_gdre_solve(dre_ctx* ctx, const dre_pencil* p, const dre_dense* B, const dre_dense* C, dre_ldlt* X0, double t0, double tf,
                   double dt, int order, int save_state, const dre_adi_options* opt, dre_gdre_result** out) {
    return guarded(ctx, [&] {
        Ctx* c = &ctx->c;
        const Pencil& P = *p->p;
        DRE_REQUIRE(P.has_device, "pencil was created host-only");
        DRE_REQUIRE(X0->pen == p, "X0 must be created with the same pencil");
        DRE_REQUIRE(B->m.rows == P.n && C->m.cols == P.n, "B must be n x m and C must be q x n");
        GdreProblem prob;
        prob.P = &P;
        prob.B = to_solver_order(c, p, B->m);
        Mat Ct(c, P.n, C->m.rows);
        transpose_mat(c, C->m, Ct);
        prob.Ct = to_solver_order(c, p, Ct);
        prob.X0 = X0->x;
        prob.t0 = t0; prob.tf = tf;
        AdiOptions ao = convert_options(opt);
        auto* r = new dre_gdre_result();
        r->pen = p; r->m = B->m.cols;
        reset_sketch_history(c, p->p->n);
        try { r->r = gdre_solve(c, prob, order, dt, save_state != 0, ao); } catch (...) { delete r; throw; }
        *out = r;
    });
}
int dre_gdre_result_info(const dre_gdre_result* r, int64_t* info) {
    info[0] = (int64_t)r->r.t.size(); info[1] = (int64_t)r->r.X.size(); info[2] = r->r.adi_iters; info[3] = r->r.nfactor;
    info[4] = (int64_t)r->r.gale.size(); info[5] = r->m; info[6] = r->pen->p->n;
    return DRE_OK;
}
int dre_gdre_result_times(const dre_gdre_result* r, double* t) {
    std::memcpy(t, r->r.t.data(), r->r.t.size() * sizeof(double));
    return DRE_OK;
}
int dre_gdre_result_K(dre_ctx* ctx, const dre_gdre_result* r, int i, double* K_host, int ld) {
    return guarded(ctx, [&] {
        Ctx* c = &ctx->c;
        DRE_REQUIRE(i >= 0 && i < (int)r->r.Kt.size(), "K index out of range");
        Mat Ktu = to_user_order(c, r->pen, r->r.Kt[i]);      // n x m
        Mat K(c, Ktu.cols, Ktu.rows);
        transpose_mat(c, Ktu, K);
        download_mat(c, K, K_host, ld);
    });
}
// K(t_i)(j, old) = Kt_i(iperm[old], j) for up to 64 time points per launch (the table of factors travels as kernel arguments): the whole
// trajectory leaves the solver ordering in one pass instead of a permutation and a transposition launch per time point
struct KTrajBatch { const double* Kt[64]; int ld[64]; };
__global__ void k_traj_export(int n, int m, const int* __restrict__ iperm, KTrajBatch bt, double* __restrict__ out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n * m) return;
    const int j = idx % m, old = idx / m;
    const int src = iperm ? iperm[old] : old;
    out[(size_t)blockIdx.y * n * m + idx] = bt.Kt[blockIdx.y][src + (size_t)j * bt.ld[blockIdx.y]];
}
static void export_trajectory(Ctx* c, const dre_gdre_result* r, double* K_dev) {
    const int nt = (int)r->r.Kt.size();
    if (nt == 0) return;
    const int n = r->r.Kt[0].rows, m = r->r.Kt[0].cols;
    if (n * m == 0) return;
    for (int i0 = 0; i0 < nt; i0 += 64) {
        KTrajBatch bt;
        const int nb = std::min(64, nt - i0);
        for (int i = 0; i < 64; ++i) {
            const Mat& K = r->r.Kt[(size_t)(i0 + std::min(i, nb - 1))];
            DRE_REQUIRE(K.rows == n && K.cols == m, "K trajectory: inconsistent shapes");
            bt.Kt[i] = K.p; bt.ld[i] = K.ld;
        }
        hipLaunchKernelGGL(k_traj_export, dim3((unsigned)((n * m + 255) / 256), (unsigned)nb), dim3(256), 0, c->stream, n, m,
                           r->pen ? (const int*)r->pen->p->iperm.p : (const int*)nullptr, bt, K_dev + (size_t)i0 * n * m);
    }
    DRE_HIP(hipGetLastError());
}
int dre_gdre_result_K_device(dre_ctx* ctx, const dre_gdre_result* r, double* K_dev) {
    return guarded(ctx, [&] {
        Ctx* c = &ctx->c;
        export_trajectory(c, r, K_dev);
        c->sync();
    });
}
int dre_gdre_result_K_all(dre_ctx* ctx, const dre_gdre_result* r, double* K_host) {
    return guarded(ctx, [&] {
        Ctx* c = &ctx->c;
        const int nt = (int)r->r.Kt.size();
        if (nt == 0) return;
        const size_t tot = (size_t)nt * r->r.Kt[0].rows * r->r.Kt[0].cols;
        if (tot == 0) return;
        DevArr<double> stage(c, tot);
        export_trajectory(c, r, stage.p);
        DRE_HIP(hipMemcpyAsync(K_host, stage.p, tot * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        c->sync();
    });
}
int dre_gdre_result_X(const dre_gdre_result* r, int i, dre_ldlt** X) {
    if (i < 0 || i >= (int)r->r.X.size()) return DRE_ERR_INVALID;
    auto* h = new dre_ldlt(); h->x = r->r.X[i]; h->pen = r->pen; *X = h;
    return DRE_OK;
}
int dre_gdre_result_gale(const dre_gdre_result* r, int j, int64_t* iinfo, double* dinfo) {
    if (j < 0 || j >= (int)r->r.gale.size()) return DRE_ERR_INVALID;
    const AdiResult& a = r->r.gale[j];
    iinfo[0] = a.iters; iinfo[1] = a.converged; iinfo[2] = a.warnings; iinfo[3] = a.rhs_cols;
    dinfo[0] = a.res_norm; dinfo[1] = a.abstol;
    return DRE_OK;
}
int dre_gdre_result_gale_history(const dre_gdre_result* r, int j, int64_t* counts, double* norms, int32_t* norm_iters, double* sre, double* sim) {
    if (j < 0 || j >= (int)r->r.gale.size()) return DRE_ERR_INVALID;
    const AdiResult& g = r->r.gale[j];
    if (counts) { counts[0] = (int64_t)g.norms.size(); counts[1] = (int64_t)g.shifts.size(); }
    for (size_t i = 0; i < g.norms.size(); ++i) { if (norms) norms[i] = g.norms[i]; if (norm_iters) norm_iters[i] = g.norm_iters[i]; }
    for (size_t i = 0; i < g.shifts.size(); ++i) { if (sre) sre[i] = g.shifts[i].real(); if (sim) sim[i] = g.shifts[i].imag(); }
    return DRE_OK;
}
int dre_gdre_result_gales_all(const dre_gdre_result* r, int64_t* iinfo, double* dinfo, double* norms, int32_t* norm_iters, double* sre, double* sim) {
    size_t on = 0, os = 0;
    for (size_t j = 0; j < r->r.gale.size(); ++j) {
        const AdiResult& g = r->r.gale[j];
        if (iinfo) {
            int64_t* ii = iinfo + 6 * j;
            ii[0] = g.iters; ii[1] = g.converged; ii[2] = g.warnings; ii[3] = g.rhs_cols; ii[4] = (int64_t)g.norms.size(); ii[5] = (int64_t)g.shifts.size();
        }
        if (dinfo) { dinfo[2 * j] = g.res_norm; dinfo[2 * j + 1] = g.abstol; }
        for (size_t i = 0; i < g.norms.size(); ++i) { if (norms) norms[on + i] = g.norms[i]; if (norm_iters) norm_iters[on + i] = g.norm_iters[i]; }
        for (size_t i = 0; i < g.shifts.size(); ++i) { if (sre) sre[os + i] = g.shifts[i].real(); if (sim) sim[os + i] = g.shifts[i].imag(); }
        on += g.norms.size(); os += g.shifts.size();
    }
    return DRE_OK;
}
int dre_gdre_result_free(dre_gdre_result* r) { delete r; return DRE_OK; }

// ---- host helpers ------------------------------------------------------------------------------
int dre_host_eigvals(int n, const double* A, double* wr, double* wi) {
    return guarded(nullptr, [&] {
        std::vector<double> M(A, A + (size_t)n * n);
        auto ev = host_eigvals(n, M);
        for (int i = 0; i < n; ++i) { wr[i] = ev[i].real(); wi[i] = ev[i].imag(); }
    });
}
int dre_host_gen_eigvals(int n, const double* A, const double* E, double* wr, double* wi) {
    return guarded(nullptr, [&] {
        std::vector<double> Av(A, A + (size_t)n * n), Ev(E, E + (size_t)n * n);
        auto ev = host_gen_eigvals(n, Av, Ev);
        for (int i = 0; i < n; ++i) { wr[i] = ev[i].real(); wi[i] = ev[i].imag(); }
    });
}
int dre_host_svd_left(int p, int w, const double* R, double* U, double* sv) {
    return guarded(nullptr, [&] {
        std::vector<double> Rv(R, R + (size_t)p * w), Uv, s;
        host_svd_left(p, w, Rv, Uv, s);
        std::memcpy(U, Uv.data(), Uv.size() * sizeof(double));
        std::memcpy(sv, s.data(), s.size() * sizeof(double));
    });
}

}  // extern "C"
