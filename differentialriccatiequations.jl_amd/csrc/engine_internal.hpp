// Declarations shared by the three translation units of the low-rank engine (not part of the interface in engine.hpp):
//   ldlt.hip    LDL' objects, compress!, norms, the Lyapunov / Riccati residuals        (src/LDLt.jl, src/lyapunov/residual.jl)
//   engine.hip  Sherman-Morrison-Woodbury pieces, shift strategies, factor cache, the ADI (src/lyapunov/adi.jl, src/shifts/*.jl, src/blocklinear/*.jl)
//   gdre.hip    Rosenbrock drivers: dense-X loop, residual-recurrence loop, gdre_solve   (src/riccati/lowrank_ros1.jl, lowrank_ros2.jl)
#pragma once
#include "engine.hpp"

namespace dre {

static const double EPS = 2.220446049250313e-16;

__device__ inline double wave_sum_d(double v) {          // 64-lane sum, result in every lane
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ---- ldlt.hip ----
Mat hcat_blocks(Ctx* ctx, const LDLt& X);
void mul_blockdiag(Ctx* ctx, const Mat& M, const LDLt& X, Mat& out);
void hcat_scale_blocks(Ctx* ctx, const LDLt& X, Mat& Lcat, Mat& LD);
bool warm_compress(Ctx* ctx, LDLt& X, const Mat& Q0, double tolfac, double abs_tol, int sx, double* missed, double rel_accept = 0.0);
bool warm_compress_eig(Ctx* ctx, LDLt& X, const Mat& Qb, double abstol_lag, double frac, double est_ratio_prev, int J_prev, Mat& Qnext, int* J_out, double* est_ratio_out);
int xblocks_max_n();
double ldlt_norm_dense_small(Ctx* ctx, const LDLt& X);
void axpy_inplace(Ctx* ctx, size_t tot, double a, const double* x, double* y);          // y += a x
LDLtP gale_residual_blocks(Ctx* ctx, const GaleOperator& op, const LDLt& C, const LDLt& X, double tolfac, double abs_tol,
                           const Mat* warm_L = nullptr, const Mat* warm_EtL = nullptr, int lead_blocks = -1, double e_coeff = 0.0);
LDLtP gale_residual_impl(Ctx* ctx, const GaleOperator& op, LDLt& C, const LDLtP& X, double tolfac, bool exact, double abs_tol,
                         const Mat* warm_L, const Mat* warm_EtL, int lead_blocks = -1, double e_coeff = 0.0);

// ---- engine.hip ----
void apply_Ft(Ctx* ctx, const GaleOperator& op, const Mat& L, Mat& out);                 // out = F' L (sparse part + low-rank update)
// batched Sherman-Morrison-Woodbury set-up of up to 16 shifts: Sinv = (alpha I + U'W_U)^-1 from the stacked product, then WKS = [N K' Sinv; E'N K' Sinv]
struct SmwBatch { const double* WK; double* Sinv; double* WKS; };
void smw_sinv_fold_batched(Ctx* ctx, int n, int m, double alpha, const SmwBatch* items, int count, int* serr);
struct PendingDense { std::shared_ptr<FactorEntry<double>> fe; Mat W; Mat stack; const void* stack_U = nullptr; int stack_m = -1; };
struct DeferredDense { std::vector<PendingDense> items; DevArr<double> norms; int cap = 0; };
void finalize_dense(Ctx* ctx, DeferredDense& dd);
template <typename T>
std::shared_ptr<FactorEntry<T>> get_factor(Ctx* ctx, const GaleOperator& op, FactorCache* cache,
                                           std::map<std::tuple<uint64_t, double, double>, std::shared_ptr<FactorEntry<T>>>& store,
                                           std::complex<double> mu, bool want_dense = true, DeferredDense* defer = nullptr, bool check_now = true);
Ctx* helper_ctx(Ctx* ctx, int h);
hipEvent_t aux_event(Ctx* ctx, int i);

}  // namespace dre
