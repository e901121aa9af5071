// Sparse side of the engine: the pencil (E', A' on one union pattern in nested-dissection order),
// CSR SpMM, and the multifrontal LU of the shifted operator with multi-RHS solves.
#pragma once
#include "common.hpp"
#include "dense.hpp"
#include "symbolic.hpp"

namespace dre {

struct cplx {
    double re, im;
};
__host__ __device__ inline cplx operator+(cplx a, cplx b) { return {a.re + b.re, a.im + b.im}; }
__host__ __device__ inline cplx operator-(cplx a, cplx b) { return {a.re - b.re, a.im - b.im}; }
__host__ __device__ inline cplx operator-(cplx a) { return {-a.re, -a.im}; }
__host__ __device__ inline cplx operator*(cplx a, cplx b) { return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
__host__ __device__ inline cplx operator*(cplx a, double b) { return {a.re * b, a.im * b}; }
__host__ __device__ inline cplx operator*(double b, cplx a) { return {a.re * b, a.im * b}; }
__host__ __device__ inline cplx& operator+=(cplx& a, cplx b) { a.re += b.re; a.im += b.im; return a; }
__host__ __device__ inline cplx& operator-=(cplx& a, cplx b) { a.re -= b.re; a.im -= b.im; return a; }
__host__ __device__ inline cplx& operator*=(cplx& a, cplx b) { a = a * b; return a; }
__host__ __device__ inline cplx recip(cplx a) {
    // Smith's algorithm
    if (fabs(a.re) >= fabs(a.im)) {
        double r = a.im / a.re, d = a.re + a.im * r;
        return {1.0 / d, -r / d};
    }
    double r = a.re / a.im, d = a.re * r + a.im;
    return {r / d, -1.0 / d};
}
__host__ __device__ inline double recip(double a) { return 1.0 / a; }
__host__ __device__ inline double abs1(double a) { return fabs(a); }
__host__ __device__ inline double abs1(cplx a) { return fabs(a.re) + fabs(a.im); }
template <typename T> __host__ __device__ inline T make_scalar(double re, double im);
template <> __host__ __device__ inline double make_scalar<double>(double re, double) { return re; }
template <> __host__ __device__ inline cplx make_scalar<cplx>(double re, double im) { return {re, im}; }

// Device copies of the symbolic structure
struct SymbolicDev {
    DevArr<int> first, size, bptr, bidx, cmap_ptr, cmap, child_ptr, child_idx, lvl_nodes;
    DevArr<int64_t> front_off, inv_off, upd_off, asm_dest;
};

// The top levels of the elimination tree as ONE dense operator (sparse.hip, mf_solve_mfma): with the pivot variables of the levels
// 0..T-1 collected in `topidx`, the forward and backward sweeps over those levels equal  x_top = inv(S) g,  S = Schur complement of
// the lower levels, g = right-hand side + the update rows of the level-T nodes.  inv(S) = (M^-1)[top, top] is computed once per
// factorisation that is reused (Cyclic shifts) and replaces 2 T latency-bound level launches by one GEMM.
struct TopPlan {
    bool built = false;
    int T = 0, ntop = 0;
    DevArr<int> topidx;         // ntop: solver-ordering index of top variable p
    DevArr<int> gptr;           // ntop + 1
    DevArr<int64_t> gsrc;       // row offsets into the update slab contributing to top variable p
};

// The subtrees below level Tsub of the elimination tree, each swept by ONE workgroup per 16 right-hand-side columns (sparse.hip,
// k_mf_sub_forward / k_mf_sub_backward): nodes and pivot rows of a subtree are contiguous in the postorder, so its slice of the
// right-hand-side panel (pivot rows + the rows of the root's boundary) stays in LDS for the whole sweep and the 2 x (nlevels - Tsub)
// latency-bound level launches collapse into two launches.
struct SubPlan {
    bool built = false;
    int Tsub = -1;              // level of the subtree roots; -1: no plan (the level kernels run everywhere)
    int nsub = 0;
    int prows_max = 0;          // largest panel (pivot rows of the subtree + boundary rows of its root)
    int nn_max = 0, lm_max = 0, b_max = 0, s_max = 0;
    size_t lds_fwd = 0, lds_bwd = 0;
    DevArr<int> sub;            // 4 ints per subtree: first node, number of nodes, first pivot row, number of pivot rows
    DevArr<int> lmap;           // per boundary entry (indexing of bidx): row of the subtree's LDS panel
};

// E' and A' (CSR == the caller's CSC of E and A) on one union pattern, permuted to the solver ordering.
struct Pencil {
    int n = 0, nnz = 0;
    Symbolic sym;
    SymbolicDev dev;
    DevArr<int> ptr, idx;           // permuted union pattern, CSR, 0-based
    DevArr<double> valEt, valAt;    // values of E' and A' on that pattern
    DevArr<int> perm, iperm;        // device copies (perm[new] = old)
    std::vector<int> lvl_maxfront;  // per level: largest front
    std::vector<int> lvl_maxsep;    // per level: largest number of pivot columns
    bool use_mfma_sweeps = true;    // real triangular sweeps on the matrix cores (env DRE_MF_SCALAR=1 selects the scalar kernels)
    bool has_device = false;
    mutable TopPlan top;            // built on first use
    mutable SubPlan sub;            // built on first use (after the top plan)
};

// Build from CSC arrays (1-based or 0-based) of E and A as Julia's SparseMatrixCSC stores them.
// upload = false builds the host part only (no GPU needed; used by CPU tests of the symbolic phase).
std::unique_ptr<Pencil> pencil_create(Ctx* ctx, int n, const int64_t* Ep, const int64_t* Ei, const double* Ev,
                                      const int64_t* Ap, const int64_t* Ai, const double* Av, int index_base,
                                      int leaf_size, bool upload, std::vector<double>* hostE = nullptr,
                                      std::vector<double>* hostA = nullptr);

// Y = alpha * M * X + beta * Y with M given by CSR (ptr, idx, val) of order n; X, Y are n x ncols.
void spmm(Ctx* ctx, int n, const int* ptr, const int* idx, const double* val, const Mat& X, Mat& Y, double alpha,
          double beta, const AdiState* st = nullptr, int nnz = -1, Mat* Yt = nullptr);      // Yt (optional): Y' written by the same launch
// the same on the pencil's pattern (val = one of its value arrays): the CSR segment of each 256-row workgroup is staged through LDS
// with coalesced loads, and the byte count of the timers uses the real number of nonzeros
inline void spmm(Ctx* ctx, const Pencil& P, const double* val, const Mat& X, Mat& Y, double alpha, double beta, const AdiState* st = nullptr,
                 Mat* Yt = nullptr) {
    spmm(ctx, P.n, P.ptr.p, P.idx.p, val, X, Y, alpha, beta, st, P.nnz, Yt);
}
// Y1 = M1 X and Y2 = M2 X for two value arrays on the pencil's pattern in one pass over X
void spmm_dual(Ctx* ctx, const Pencil& P, const double* val1, const double* val2, const Mat& X, Mat& Y1, Mat& Y2);
// out = a*x + b*y on value arrays of the shared pattern (shifted-operator assembly K4, values only)
void vals_axpby(Ctx* ctx, int nnz, double a, const double* x, double b, const double* y, double* out);
// row permutation helpers: dst(i,:) = src(map[i],:)
void permute_rows(Ctx* ctx, const Mat& src, const int* map_dev, Mat& dst);

template <typename T>
struct Factor {
    DevArr<T> fronts;   // all frontal matrices (L\U of the eliminated block, L21, U12, Schur complement)
    DevArr<T> inv;      // inverted diagonal blocks: strict lower = inv(L11), upper = inv(U11)
    DevArr<int> err;    // device flag: zero / NaN pivot met (checked lazily by mf_check)
    DevArr<unsigned long long> growth;   // largest multiplier of the pivot-free LU (bit pattern of a double, atomic max)
    // dense inverse of the top levels' Schur complement (real MFMA sweeps only): built at the third multi-column solve of a factor
    // the engine marked as reusable
    bool allow_topinv = false;
    mutable int uses = 0;
    mutable Mat topinv;
    // Static pivoting (round 3; the reference factorises with partial pivoting, UMFPACK / CHOLMOD, blocklinear/backslash.jl:13): a pivot smaller
    // than pivot_static x max|entry of the shifted operator| is replaced by that value with the pivot's sign (SuperLU_DIST's GESP), so the
    // multipliers stay below 1 / pivot_static and what is factorised is a slightly perturbed matrix; solves with such a factor run fixed-point
    // refinement x += F^-1 (b - M x) against the TRUE operator (values kept below).  npert counts the replaced pivots (device), nperturbed is
    // the host copy mf_check fills in (-1 = not read back yet).
    DevArr<double> pivfloor;            // [0] the floor, [1] max |entry|
    DevArr<int> npert;
    mutable int nperturbed = -1;
    const double* ref_valF = nullptr; const double* ref_valE = nullptr; T ref_cF{}, ref_cE{};
};

// Numeric multifrontal LU (no pivoting) of  M = cF * F' + cE * E'  where valF/valE live on the pencil's pattern.
template <typename T>
void mf_factor(Ctx* ctx, const Pencil& P, const double* valF, const double* valE, T cF, T cE, Factor<T>& out);
// nz real factorisations  M_z = cF F' + cE[z] E'  of the same pencil in shared launches (every tree level once for all of them), and their
// dense top inverses likewise
template <typename T>
void mf_factor_batch(Ctx* ctx, const Pencil& P, const double* valF, const double* valE, T cF, const T* cE, Factor<T>* const* outs, int nz);     // T = double, cplx
void mf_topinv_batch(Ctx* ctx, const Pencil& P, Factor<double>* const* Fs, int nz);
// Synchronises and throws ERR_SINGULAR if the factorisation met a zero pivot.
// Also returns the pivot growth (largest multiplier); throws ERR_SINGULAR beyond ctx->pivot_growth_fail.
template <typename T>
double mf_check(Ctx* ctx, const Factor<T>& F);
std::vector<double> mf_check_batch(Ctx* ctx, const std::vector<const Factor<double>*>& fs);     // several factors, one synchronisation
// In-place solve  M * X = W  for the n x nrhs panel W (column-major, leading dimension ldw), solver ordering.
template <typename T>
void mf_solve(Ctx* ctx, const Pencil& P, const Factor<T>& F, T* W, int ldw, int nrhs, const AdiState* st = nullptr);
// Builds the dense inverse of the top of the elimination tree NOW (mf_solve builds it lazily at the third multi-column solve of a reusable
// factor): for factors that are known to be reused many times (Cyclic shift lists), on whatever stream the caller runs them on.
void mf_prepare_topinv(Ctx* ctx, const Pencil& P, const Factor<double>& F);
// W = F^-1 [Win(:, 0:nin) | W(:, nin:nrhs)] — the leading right-hand sides are read in place (no copy into the work panel)
void mf_solve_from(Ctx* ctx, const Pencil& P, const Factor<double>& F, const double* Win, int ldwin, int nin, double* W, int ldw, int nrhs,
                   const AdiState* st = nullptr);

// partial-fraction coefficients of a fan group (engine.hip, fan_coefficients) and the fused  E' W + mixing  pass over the g solves
#define FAN_GMAX 10
struct FanCoef { double c[FAN_GMAX][FAN_GMAX], d[FAN_GMAX][FAN_GMAX]; };
struct FanSlots { int s[FAN_GMAX]; };       // solve s lives in columns s[s] k .. of the panel (shift-sharded groups: slab of the owning rank)
void fan_spmm_mix(Ctx* ctx, const Pencil& P, const Mat& W, const Mat& R0, Mat& V, Mat& Rc, int g, int k, const FanCoef& co, const FanSlots& sl,
                  const AdiState* st = nullptr);
// g solves with different factors of the same pencil in shared launches: W_z = F_z^-1 [Win(:, 0:nin) | W_z(:, nin:nrhs)], W_z = columns
// z nrhs .. of the n x (g nrhs) panel W.  false (nothing enqueued) where the batched form does not apply.
bool mf_solve_batch(Ctx* ctx, const Pencil& P, const Factor<double>* const* Fs, int g, const double* Win, int ldwin, int nin, double* W, int ldw, int nrhs,
                    const AdiState* st = nullptr);

}  // namespace dre
