// Device-resident low-rank engine: LDL' objects, GALE operators, ADI, shift strategies, Rosenbrock drivers.
// Mirrors (not translates) the reference's hot path; each routine cites the file:line it replaces.
#pragma once
#include <complex>
#include <functional>
#include <map>

#include "common.hpp"
#include "dense.hpp"
#include "sparse.hpp"

namespace dre {

// ---- LDL' low-rank object (/root/reference/src/LDLt.jl:29-33): lazy sum_i alpha_i L_i D_i L_i' -----------
struct LBlock {
    Mat L;          // n x k
    Mat D;          // k x k (dense storage)
    double alpha = 1.0;
    bool diag = false;   // D known to be diagonal
    bool ortho = false;  // L has orthonormal columns (output of compress!): the Gram form of the norm is then fully accurate
};
struct LDLt {
    int n = 0;
    std::vector<LBlock> blocks;
    int rank() const { int r = 0; for (auto& b : blocks) r += b.L.cols; return r; }
    bool iszero() const {
        if (rank() == 0) return true;
        for (auto& b : blocks) if (b.alpha != 0.0) return false;
        return true;
    }
};
using LDLtP = std::shared_ptr<LDLt>;

LDLtP ldlt_make(Ctx* ctx, int n, const Mat& L, const Mat& D, double alpha = 1.0, bool diag = false);
LDLtP ldlt_zero(int n);
LDLtP ldlt_add(const LDLtP& a, const LDLtP& b);                 // LDLt.jl:131-148 (list append, factors shared)
LDLtP ldlt_scale(const LDLtP& a, double alpha);                  // LDLt.jl:156-159
LDLtP ldlt_deepcopy(Ctx* ctx, const LDLtP& a);
void ldlt_concatenate(Ctx* ctx, LDLt& X);                        // LDLt.jl:174-191
// LDLt.jl:204-225.  exact = true: eigen-decomposition + threshold 100*eps*max|lambda| exactly as the reference.
// exact = false (engine default): the early-terminated Householder reduction alone, D := T_j (tridiagonal); the
// dropped part is below tolfac*eps*||S||_F, i.e. not larger than what the reference's threshold discards.
// mode bits (engine's own compression only, exact = false):
//   COMPRESS_NOISE_FLOOR  the truncation tolerance is max(relative, rounding noise of FORMING the sum) — for sums whose terms cancel
//                         (||X|| << ||L||^2 ||D||: the Riccati residual that is Ros2's stage-1 right-hand side near the steady state), where
//                         the relative tolerance alone would keep the formation noise as signal; abs_tol is ignored
//   COMPRESS_KEEP_RESULT  never hand the summands back when nothing was gained: the result is always ONE block with orthonormal L, on which
//                         the Gram form of the norm (adi loop) is accurate whatever cancelled in the summands
//   COMPRESS_TIGHT        the band reduction stops at panel boundaries of a Krylov basis: its rank J lies above the number of eigenvalues the
//                         reference keeps (LDLt.jl:237-245; measured 160 - 256 against 113 - 144 columns on the stage solves of Ros2 at n = 1357).
//                         The J x J band matrix is diagonalised and the reference's own threshold 100 eps max|lambda| (or the noise floor, if
//                         larger) applied: the result is one block with orthonormal L and DIAGONAL D of the reference's rank — for factors that
//                         become the right-hand side of the next Lyapunov solve, whose cost is proportional to their width
enum { COMPRESS_NOISE_FLOOR = 1, COMPRESS_KEEP_RESULT = 2, COMPRESS_TIGHT = 4 };
void ldlt_compress(Ctx* ctx, LDLt& X, double tolfac = 4.0, bool exact = true, double abs_tol = -1.0, int mode = 0);
// residual(::GAREProblem, ::LDLt) and the feedback E'XB from the device factors of X (solver ordering throughout)
LDLtP gare_residual_dev(Ctx* ctx, const Pencil& P, LDLt& X, const Mat& Ct, const Mat& S, double gamma, const Mat& B, const Mat& Rinv, double beta);
Mat ldlt_feedback_dev(Ctx* ctx, const Pencil& P, LDLt& X, const Mat& B);
double ldlt_norm(Ctx* ctx, LDLt& X);                             // LDLt.jl:77-89 (concatenates, synchronises)
double ldlt_norm_accurate(Ctx* ctx, const LDLt& X);   // QR/compression based like LDLt.jl:77-89 (robust to cancellation); X unchanged
void ldlt_destructure(Ctx* ctx, LDLt& X, double tolfac = 4.0, bool exact = true);   // LDLt.jl:54-60: compress iff more than one block


// ---- GALE operator  F = Fs + inv(alpha) U V  with sparse Fs on the pencil's pattern ----------------------
// (/root/reference/src/LowRankUpdate.jl:18-26, src/lyapunov/types.jl:10-16)
// `dinv` (real shifts, small n only): the explicit dense inverse of the shifted sparse operator, applied with one MFMA
// GEMM per ADI step instead of the level-by-level triangular sweeps (which are launch-latency bound at small n).
// `stack` = [inv; E' inv; U' inv] ((2n + m) x n) for the low-rank factor U of the current operator: ONE GEMM with the residual
// factor R then yields the plain solve, its image under E' (for the residual recurrence) and the SMW inner products.
template <typename T> struct FactorEntry { Factor<T> f; Mat dinv; bool dense = false; Mat stack; const void* stack_U = nullptr; int stack_m = -1;
                                            bool checked = false; /* pivot-breakdown flag already read back */ double growth = 0.0; /* its pivot growth */ };
struct FactorCache {
    std::map<std::tuple<uint64_t, double, double>, std::shared_ptr<FactorEntry<double>>> real;
    std::map<std::tuple<uint64_t, double, double>, std::shared_ptr<FactorEntry<cplx>>> cplx_;
    long nfactor = 0;
    bool enabled = true;
    std::vector<std::tuple<uint64_t, double, double>> fresh;   // keys created by the running Lyapunov solve (evicted unless the shift list persists)
    int iters_hint = 0;        // ADI iterations of the previous Lyapunov solve served by this cache (speculation depth of the next one)
    int warm_sx = 0, warm_strikes = 0;   // warm-started residual compression (ldlt.hip, warm_compress): fresh directions per step, consecutive rejections
    // sharded fan groups: operators (tags) for which the ranks have agreed whether the batched solves apply to EVERY rank's factors (value: refused)
    std::map<uint64_t, bool> fan_agreed;
    void clear() { real.clear(); cplx_.clear(); }
};
struct GaleOperator {
    const Pencil* P = nullptr;
    DevArr<double> valFt;      // values of the sparse part of F' on the pencil's pattern
    uint64_t tag = 0;          // identity of valFt for the factor cache
    double cA = 1.0, cE = 0.0; // Fs = cA*A + cE*E (coefficients behind valFt; handed to a user-supplied block solver)
    bool has_lr = false;
    double alpha = 1.0;        // F = Fs + inv(alpha) * U * V
    Mat U;                     // n x m
    Mat Vt;                    // n x m  (V')
};
// Ritz values of E^-1 F (rplus) and F^-1 E (rminus), kplus / kminus Arnoldi steps from ones(n)  (shifts/heuristic.jl:39-66,103-130)
void heuristic_ritz(Ctx* ctx, const GaleOperator& op, int kplus, int kminus, std::vector<std::complex<double>>& rplus,
                    std::vector<std::complex<double>>& rminus);

// ---- shift strategies (/root/reference/src/Shifts.jl:79-116, src/shifts/*.jl) -----------------------------
// user-defined strategy (dre_shift_fn of include/dre_hip.h: Shifts.init / update! / take!, src/Shifts.jl:79-116)
typedef int (*ShiftFn)(void* user, int restart, int n, int hist_cols, const double* hist, int ldh, int capacity, double* re, double* im, int* count);
struct ShiftSpec {
    enum Kind { CYCLIC = 0, PROJECTION = 1, HEURISTIC = 2, USER = 3 } kind = PROJECTION;
    ShiftFn user_fn = nullptr;                  // USER
    void* user_data = nullptr;
    std::vector<std::complex<double>> values;   // CYCLIC
    int n_history = 2;                          // PROJECTION
    int h_nshifts = 0, h_kplus = 0, h_kminus = 0;   // HEURISTIC: Cyclic(Heuristic(nshifts, k+, k-)), recomputed per Lyapunov solve
};

// User-supplied solver of the SPARSE shifted system  (cA*A' + (cE_re + i cE_im)*E') X = B  (the ALG of ShermanMorrisonWoodbury(ALG, alg),
// src/blocklinear/types.jl:35-39; plug-in protocol types.jl:15-62, example test/cuda.jl:23-30,74).  Device pointers, caller's row
// ordering, column-major with leading dimension n; X_im is null for a real system.  Returns 0 on success.
typedef int (*BlockSolverFn)(void* user, int n, int nrhs, double cA, double cE_re, double cE_im, const double* B, double* X_re, double* X_im);

struct AdiOptions {   // /root/reference/src/lyapunov/types.jl:20-30
    int maxiters = 100;
    double reltol = -1.0;   // < 0: "nothing" -> n*eps
    double abstol = -1.0;   // < 0: "nothing" -> reltol*norm(C)
    bool ignore_initial_guess = false;
    int compression_interval = 10;
    bool compression = true;
    ShiftSpec shifts;
    double compress_tolfac = 4.0;    // Krylov-truncated compression inside the engine: remainder <= tolfac*eps*||X||_F
    double residual_abs_frac = 0.05; // the warm-start residual is truncated at this fraction of abstol (Krylov mode only)
    bool compress_exact = false;     // true: eigen-based truncation at every compression (reference arithmetic)
    BlockSolverFn inner_solve = nullptr;   // inner_alg = ShermanMorrisonWoodbury(user solver, Backslash) instead of the library's multifrontal LU
    void* inner_user = nullptr;
    bool final_compress = true;      // internal (Ros1 driver): false keeps the solution as warm start + increments (block list)
    bool tight_final = false;        // internal (Ros2 driver): the final compression of the solution truncates at the reference's rank (COMPRESS_TIGHT)
    Mat warm_L, warm_EtL;            // internal (Ros1 driver): concatenated factor of the warm start and E' times it, if already at hand
    int rhs_lead_blocks = -1;        // internal (Ros1 driver): the right-hand side is  C = (first rhs_lead_blocks blocks) + rhs_e_coeff * E'XE
    double rhs_e_coeff = 0.0;        //   with X the warm start, so the residual folds the last term into F: (F + coeff/2 E)' X E + E' X (F + coeff/2 E)
    // internal (Ros1 driver with the residual recurrence, gdre.hip ros1_recurrence_loop): the warm-start residual arrives as a block list
    // (no right-hand side, no initial guess: the solve returns the INCREMENT), the tolerance reltol * ||C||_F arrives later in device memory
    // (the side stream forms it from the compressed X) and the convergence decisions of the first chunk are taken when it is there
    std::shared_ptr<struct LDLt> given_residual;
    const double* normC_dev = nullptr;
    std::function<void()> normC_wait;    // makes the calling stream wait for *normC_dev
    // ... or the caller forms it as soon as the given residual is compressed to (Q, D, alpha): the callback makes a stream of its own wait for
    // `ready` (recorded on the solve's stream behind the compression), enqueues the work there and records `done` behind it.  It may hand all of
    // that to another host thread and return at once (a step of the general path is bound by host launches): the solve calls normC_join()
    // before it makes its stream wait for `done`.
    std::function<void(hipEvent_t ready, const Mat& Q, const Mat& D, double alpha, hipEvent_t done)> normC_build;
    std::function<void()> normC_join;
    double abstol_lag = -1.0;            // tolerance of the previous time step: truncation level of the warm-start residual
    bool keep_history = false;           // keep every iteration's V_j and R_j side by side (AdiResult::hist)
    Mat warm_basis;                      // orthonormal basis of the PREVIOUS step's compressed warm-start residual (warm-started range finder for this one's)
    Mat warm_eig_basis;                  // EIGENBASIS (with spare directions) the previous step's compression left: Rayleigh-Ritz compression (ldlt.hip, warm_compress_eig)
    double warm_est_ratio = -1.0;        //   (missed / tolerance)^2 of that compression: what this one's truncation may use of the budget
    int warm_J_prev = -1;                //   its rank
};
// iterations of one chunk of a solve with keep_history: V = [V_1 .. V_J], R = [R_1 .. R_J] (n x J k), R0 = the residual factor they started from
struct AdiHistChunk { Mat R0, Rs, Vs; std::vector<double> mu; };
struct AdiResult {
    LDLtP X;
    LDLtP residual;
    int iters = 0;
    double res_norm = 0.0, abstol = 0.0, initial_norm = 0.0;
    bool converged = false;
    int warnings = 0;       // bit 0: not converged (adi.jl:126); 4/8: Ritz values discarded/flipped (helpers.jl:133,136)
    std::vector<double> norms;                    // residual norm after each observed step (index 0 = initial)
    std::vector<int> norm_iters;                  // iteration number of each entry of `norms`
    std::vector<std::complex<double>> shifts;     // shifts consumed
    int rhs_cols = 0;
    // keep_history: the accepted iterations chunk by chunk, the inner matrix and scaling of the residual  alpha_res R T R'  (T never changes
    // during a solve, adi.jl:150-177), and whether every iteration is in there (false: an iteration ran outside the fan path or X was compressed)
    std::vector<AdiHistChunk> hist;
    bool hist_ok = false;
    Mat Tm; double alpha_res = 1.0; bool tdiag = false;
    Mat warm_basis_out; int warm_J = -1; double warm_est_ratio = -1.0;      // the eigenbasis this solve's residual compression left (empty: another path ran)
};
AdiResult adi_solve(Ctx* ctx, const GaleOperator& op, LDLt& C, const LDLtP& initial_guess, const AdiOptions& opt,
                    FactorCache* cache);
// The same solve as a resumable object (the reference's ADICache protocol init / step! / isdone / solve!, adi.jl:29-141): adi_begin = init,
// adi_advance(run, budget) enqueues up to `budget` shifts (a conjugate pair counts two and is never split) and synchronises once,
// adi_finish = final compression (adi.jl:78-80) + result.  adi_solve = begin, advance until done, finish — the stepwise and the
// one-shot form run the same kernels in the same order, so their results are identical bit for bit (test/tiny_random.jl:48-57).
struct AdiRun;
std::shared_ptr<AdiRun> adi_begin(Ctx* ctx, const GaleOperator& op, LDLt& C, const LDLtP& initial_guess, const AdiOptions& opt, FactorCache* cache);
void adi_advance(AdiRun& run, int budget);
void adi_snapshot(AdiRun& run, LDLtP* X, LDLtP* resid);     // observer payload of adi.jl:119 at the current iteration
std::vector<std::complex<double>> adi_shifts_since(const AdiRun& run, int from);   // shifts consumed by the accepted iterations from..iters
bool adi_isdone(const AdiRun& run);
void adi_peek(const AdiRun& run, int* iters, double* res_norm, double* abstol);
AdiResult adi_finish(AdiRun& run);
// X = (M + inv(alpha) Vt U')^-1 B  through Sherman-Morrison-Woodbury with a factorisation of the sparse part M
// (src/blocklinear/sherman-morrison-woodbury.jl:10-45): W = M^-1 [B, Vt], S = alpha I + U' W_Vt, X = W_B - W_Vt S^-1 (U' W_B).  Solver ordering.
Mat smw_solve(Ctx* ctx, const Pencil& P, const Factor<double>& F, double alpha, const Mat& U, const Mat& Vt, const Mat& B);
void smw_solve(Ctx* ctx, const Pencil& P, const Factor<cplx>& F, double alpha, const Mat& U, const Mat& Vt, const Mat& B, Mat& X_re, Mat& X_im);
// dot(X1, X2) = <X1, X2>_F = sum_ij a_i a_j tr(D_i (L_i' L_j) D_j (L_j' L_i))   (/root/reference/src/LDLt.jl:91-108); synchronises
double ldlt_dot(Ctx* ctx, const LDLt& X1, const LDLt& X2);
// LyapunovOperator(E, F) * X = F'XE + E'XF = [E'L, F'L] [0 D; D 0] [E'L, F'L]'   (/root/reference/src/lyapunov/gmres.jl:108-120)
LDLtP lyapunov_apply(Ctx* ctx, const GaleOperator& op, const LDLtP& X);
// residual of A'XE + E'XA + C for an LDL' iterate (/root/reference/src/lyapunov/residual.jl:3-31)
LDLtP gale_residual(Ctx* ctx, const GaleOperator& op, LDLt& C, const LDLtP& X, double tolfac = 4.0, bool exact = true, double abs_tol = -1.0);

// ---- Rosenbrock drivers (/root/reference/src/riccati/lowrank_ros1.jl, lowrank_ros2.jl) -------------------
struct GdreProblem {
    const Pencil* P = nullptr;
    Mat B;    // n x m  (solver ordering)
    Mat Ct;   // n x q  (C')
    LDLtP X0;
    double t0 = 0, tf = 0;
};
struct GdreResult {
    std::vector<double> t;
    std::vector<Mat> Kt;              // K(t_i)' stored as n x m (solver ordering)
    std::vector<LDLtP> X;             // first/last or all (save_state)
    std::vector<AdiResult> gale;      // one per Lyapunov solve (Ros2: two per step), X/residual handles dropped
    long adi_iters = 0;
    long nfactor = 0;
};
GdreResult gdre_solve(Ctx* ctx, const GdreProblem& prob, int order, double dt, bool save_state, const AdiOptions& adi);

}  // namespace dre
