// Dense f64 building blocks (column-major): MFMA GEMM, copies, Gram/trace norms,
// blocked Householder QR, symmetric eigensolver with early-terminating tridiagonalisation.
#pragma once
#include <cstdlib>
#include <functional>
#include "common.hpp"

namespace dre {

// Device-resident control block of one ADI solve.  Every kernel of an ADI iteration takes a
// pointer to it and returns immediately once `done` is set, so the host can enqueue iterations
// speculatively and synchronise only once per compression interval.
struct AdiState {
    int done;            // 1 = converged / collapsed / maxiters reached
    int iters;           // shifts consumed so far (a complex pair counts 2)
    int maxiters;
    int smw_singular;    // set by the capacitance-matrix inversion (SMW) when a pivot vanishes; read back with the state
    double abstol;
    double res_norm;
    double norms[512];   // residual norm after iteration i (index = shifts consumed)
    // meeting point of the g norm workgroups of a fan group (dense.hip, k_gram_norm_z): zeroed with the control block, reset by the last arrival
    int ticket, collapsed;   // collapsed: scratch of the zero-increment guard (engine.hip, k_zero_increment); 2 = the iteration collapsed (adi.jl:134-137)
    double gnorm[16];
};

// A count that only exists in device memory when the consumer is enqueued: the number of ADI iterations of a speculatively enqueued
// chunk that were accepted, clamp(st->iters - base, 0, nmax).  Kernels that take one either skip the products beyond it (batched GEMM)
// or shorten their inner dimension to per * count (split-K GEMM), so that the update of X can be enqueued before the host knows
// how many increments there are.
struct DevCount { const AdiState* st = nullptr; int base = 0, nmax = 0, per = 1; };
#if defined(__HIPCC__)
__device__ __forceinline__ int dev_count(const DevCount& dc) { const int c = dc.st->iters - dc.base; return c < 0 ? 0 : (c > dc.nmax ? dc.nmax : c); }
#endif
// C = alpha*op(A)*op(B) + beta*C   (M x N, inner K).  `st` may be null.
void gemm(Ctx* ctx, bool transA, bool transB, int M, int N, int K, double alpha, const double* A, int lda,
          const double* B, int ldb, double beta, double* C, int ldc, const AdiState* st = nullptr,
          const char* tag = "gemm_f64_mfma", double* tile_sumsq = nullptr);
// tile_sumsq (optional, K <= 64 only): receives one partial sum of squares of the updated C per 64 x 64 tile
// (gemm_num_tiles(M, N) entries) — the band reduction's termination norm comes out of the update GEMM's epilogue.
inline int gemm_num_tiles(int M, int N) { return ((M + 63) / 64) * ((N + 63) / 64); }
// C = alpha op(A) B + beta C for a SMALL C (a few dozen 16 x 16 tiles) and a long inner dimension, in one launch (no split-K slabs)
void gemm_thin(Ctx* ctx, bool tA, int M, int N, int K, double alpha, const double* A, int lda, const double* B, int ldb, double beta, double* C, int ldc,
               const AdiState* st = nullptr, const char* tag = "gemm");
inline void gemm(Ctx* ctx, bool tA, bool tB, double alpha, const Mat& A, const Mat& B, double beta, Mat& C,
                 const AdiState* st = nullptr, const char* tag = "gemm_f64_mfma", double* tile_sumsq = nullptr) {
    int M = tA ? A.cols : A.rows, K = tA ? A.rows : A.cols, N = tB ? B.rows : B.cols;
    int K2 = tB ? B.cols : B.rows;
    DRE_REQUIRE(K == K2 && C.rows == M && C.cols == N, "gemm: shape mismatch");
    gemm(ctx, tA, tB, M, N, K, alpha, A.p, A.ld, B.p, B.ld, beta, C.p, C.ld, st, tag, tile_sumsq);
}

// Split-K GEMM that leaves the per-split partial slabs (each M x N, leading dimension M) unreduced for a consumer kernel
// that sums them in a fixed order while doing its own work; returns the slab buffer, *splits_out slabs.
// Batched C_z = alpha_z A_z B_z (no transposes, no split-K) in one launch; copy_dst_z (optional) also receives A_z.
struct GemmBatchDesc {
    const double* A; const double* B; double* C; double* copy_dst;
    double alpha;
    int M, N, K, lda, ldb, ldc, ldcopy;
};
void gemm_batched(Ctx* ctx, const std::vector<GemmBatchDesc>& descs, const char* tag = "gemm_batched", DevCount dc = DevCount{});
// sum of the split-K slabs of gemm_partials (M x N each, fixed order) written to C[rowmap[row], col]
void gemm_reduce_rows(Ctx* ctx, int M, int N, int splits, const double* partial, const int* rowmap, double* C, int ldc, const AdiState* st = nullptr);
BufP gemm_partials(Ctx* ctx, bool transA, bool transB, int M, int N, int K, const double* A, int lda, const double* B, int ldb,
                   int* splits_out, const AdiState* st = nullptr, const char* tag = "gemm_f64_mfma", DevCount dc = DevCount{});
// z-batched split-K products of one shape in one launch (fan groups of the general path): slab (z, split) at (z * splits + split) * M * N
#define MF_ZMAX 16
struct GemmZ { const double* A[MF_ZMAX]; const double* B[MF_ZMAX]; };
BufP gemm_partials_z(Ctx* ctx, bool transA, bool transB, int M, int N, int K, const GemmZ& zb, int nz, int lda, int ldb, int* splits_out,
                     const AdiState* st = nullptr, const char* tag = "gemm_f64_mfma");
// C_z[rowmap ? rowmap[row] : row, col] = fixed-order sum of the slabs of product z;  C_z = C + z * cz
void gemm_reduce_z(Ctx* ctx, int M, int N, int splits, int nz, const double* partial, const int* rowmap, double* C, int ldc, long cz, const AdiState* st = nullptr);
void copy_mat(Ctx* ctx, const Mat& src, Mat& dst, double scale = 1.0, const AdiState* st = nullptr);  // dst = scale*src
struct CopyDesc { const double* src; double* dst; int rows, cols, lds, ldd; };
void copy_batched(Ctx* ctx, const std::vector<CopyDesc>& descs);                      // all blocks in one launch per 32
void fill_mat(Ctx* ctx, Mat& dst, double v);
void set_identity(Ctx* ctx, Mat& dst, double v = 1.0);       // dst = v*I (square or rectangular)
void fill_gauss(Ctx* ctx, Mat& A, unsigned long long seed);
// W (s x c) = Om' L for the structured sparse sign test matrix Om (n x s, SKETCH_ZETA entries +-1/sqrt(SKETCH_ZETA) per row; deterministic in seed)
#define SKETCH_ZETA 8
void sketch_sign(Ctx* ctx, const Mat& L, Mat& W, unsigned long long seed);
// Rinv = inv(R) for the upper Cholesky factor R of the symmetric positive semidefinite G (order <= 64), null columns (pivot below the floor
// selected by mode / *ref_dev, see k_chol_inv) zeroed; *flag_dev |= 1 when a live pivot is too small for Cholesky QR
// nullmask_dev (optional, order entries): 1 where a null column was found
// dbg_dev (optional): receives the smallest live pivot relative to the largest diagonal entry (conditioning trace)
void chol_inv(Ctx* ctx, const Mat& G, Mat& Rinv, int* flag_dev, double* ref_dev, int mode, int* nullmask_dev = nullptr, double* dbg_dev = nullptr);
// Gaussian entries of variance 1/rows into the columns of A that mask_dev marks (the others are left alone)
void fill_gauss_masked(Ctx* ctx, Mat& A, unsigned long long seed, const int* mask_dev);   // independent standard normal entries (deterministic in seed and position)
void transpose_mat(Ctx* ctx, const Mat& src, Mat& dst);      // dst = src'
void add_diag(Ctx* ctx, Mat& dst, const double* diag_dev, double scale);  // dst += scale*diag(v)
void symmetrize(Ctx* ctx, Mat& S);                           // S = (S+S')/2
void scale_cols_by_diag(Ctx* ctx, const Mat& L, const Mat& D, Mat& out, double alpha);  // out = alpha * L * diag(D_ii)
double frob_norm_host(Ctx* ctx, const Mat& A);               // synchronising
// Small device->host read-back ordered after everything enqueued on the context's stream so far (replaces hipMemcpyAsync + hipStreamSynchronize
// on the critical path): up to three device ranges (multiples of 8 bytes, together at most 8 KB) land in the given host buffers.
void ctx_fetch(Ctx* ctx, const void* d0, size_t b0, void* h0, const void* d1 = nullptr, size_t b1 = 0, void* h1 = nullptr,
               const void* d2 = nullptr, size_t b2 = 0, void* h2 = nullptr);
// the same with work enqueued by `between` right after the signal kernel, before the host starts waiting for the words
void ctx_fetch_overlap(Ctx* ctx, const std::function<void()>& between, const void* d0, size_t b0, void* h0, const void* d1 = nullptr, size_t b1 = 0,
                       void* h1 = nullptr, const void* d2 = nullptr, size_t b2 = 0, void* h2 = nullptr);
void gemm_sym_update(Ctx* ctx, const Mat& A, const Mat& B, Mat& X, const char* tag = "gemm_f64_mfma", DevCount dc = DevCount{});   // X <- sym(X + A B')
void frob2_device(Ctx* ctx, const Mat& A, double* out_dev);   // ||A||_F^2 into device memory (no synchronisation)
bool is_diagonal_host(Ctx* ctx, const Mat& D);               // synchronising (small)

// nrm = |alpha| * sqrt(trace((T*G)^2)) = |alpha| * ||R T R'||_F with G = R'R
// (/root/reference/src/LDLt.jl:77-89, evaluated through the Gram matrix instead of a pivoted QR).
// Writes st->res_norm, st->norms[st->iters], and sets st->done on convergence / maxiters.
void ldlt_norm_update_state(Ctx* ctx, const Mat& G, const Mat& T, bool tdiag, double alpha, AdiState* st, int iters_after);
// The whole residual-norm step  G = R'R,  nrm = |alpha| sqrt(tr((T G)^2)),  convergence decision  in two launches:
// the split-K Gram GEMM and one workgroup that reduces the partial slabs, forms T G in LDS and decides (k <= 88).
void residual_norm_group(Ctx* ctx, const Mat& Rcat, int g, int k, const Mat& T, bool tdiag, double alpha, AdiState* st, int iters0);   // g residuals at once
// the same from the g DIAGONAL blocks of the Gram matrix only (one z-batched product), the g norms in parallel workgroups, decisions in order
void residual_norm_group_diag(Ctx* ctx, const Mat& Rcat, int g, int k, const Mat& T, bool tdiag, double alpha, AdiState* st, int iters0);
void residual_norm_step(Ctx* ctx, const Mat& R, const Mat& T, bool tdiag, double alpha, AdiState* st, int iters_after);
void ldlt_norm_device(Ctx* ctx, const Mat& L, const Mat& D, double alpha, double* out_dev);          // |alpha| ||L D L'||_F into device memory, no synchronisation
void adi_decide_scan(Ctx* ctx, AdiState* st, int count, const double* normC_dev, double reltol, double abstol_given);   // deferred decisions for iterations 0 .. count
void ev_from_residuals(Ctx* ctx, int n, int k, int J, const Mat& R0, const Mat& Rs, Mat& EV, const double* mu);       // E'V_j = (R_{j-1} - R_j) / (2 mu_j)
// fused dense-inverse ADI step (apply + residual recurrence + Gram matrix + convergence decision), see dense.hip
// The norm kernel of iteration i can ride on the step kernel of iteration i + 1 (one more workgroup) instead of being a launch of
// its own: `pend` carries the Gram slabs of the iteration whose norm is still due.  dense_norm_flush launches it on its own (before
// a host synchronisation or a step of another kind).
struct DenseNormPending { BufP gpart; int nblk = 0; int iters_after = 0; bool valid = false; };
void dense_adi_step(Ctx* ctx, int n, int m, int k, int splits, const double* Wpart, const double* WKS, int ldwk, Mat& V, Mat& R,
                    double two_mu, const Mat& T, bool tdiag, double alpha, AdiState* st, int iters_after, DenseNormPending* pend = nullptr);
void dense_norm_flush(Ctx* ctx, int k, const Mat& T, bool tdiag, double alpha, AdiState* st, DenseNormPending* pend);
// Fast ADI chain (dense.hip): SMW-folded stacked inverse in MFMA-operand order, one launch per ADI iteration, residual norm
// pipelined over the next two launches.
#define ADI_FAST_MAX_K 512
#define ADI_FAST_NWS 1032         // doubles of the norm meeting point: (MAX_K/16) * (MAX_K/64) partial sums + the ticket word in the last slot
inline int adi_fast_nstrip(int n) { return (n + 15) / 16; }
inline int adi_fast_kst(int n) { return (n + 3) / 4; }
inline size_t adi_fast_pack_doubles(int n) { return (size_t)2 * adi_fast_nstrip(n) * adi_fast_kst(n) * 64; }
// out_j = packed [inv; E'inv]_j - WKS_j (U'inv)_j for every shift j (stack_j is (2n + m) x n with leading dimension lds_; WKS_j is 2n x m or null)
void adi_fast_build(Ctx* ctx, int n, int m, const std::vector<const double*>& stacks, int lds_, const std::vector<const double*>& wks, int ldwk,
                    const std::vector<double*>& outs);
struct AdiFastArgs {
    int n, k, nstrip, kst;
    const double* Apack;        // packed effective stack of this iteration's shift
    const double* Rcur; int ldr;
    double* Rnext; int ldr_next;
    // the residual ALSO in the lane order of the MFMA B operand ("packed": K-step t, column tile j -> 64 consecutive doubles, element
    // (row 4 t + (lane >> 4), column 16 j + (lane & 15)) at ((t * ct + j) * 64 + lane), zero padded to 4 nstrip K-steps): written by the
    // epilogue of the launch that produces it (each wave's 64 results ARE one such block), read by the next launch as single coalesced
    // 512-byte fragment loads instead of 16 separate 32-byte pieces per load.  Null: column-major gathers (mode 1, the general GALE chain).
    const double* Rpc; double* Rpn;
    double* V; int ldv;
    double two_mu;
    double* G_prev;             // receives Rcur' Rcur (Gram matrix of the residual the previous launch produced), or null
    const double* G_prev2;      // Gram matrix the previous launch produced (norm + decision now), or null
    const double* T; int ldt; int tdiag; double alpha;
    AdiState* st;
    double* nws;                // ADI_FAST_MAX_K / 16 partial sums + a ticket word (zero-initialised once per solve): meeting point of the norm workgroups
    int it_prev2;               // shifts consumed after the iteration G_prev2 belongs to
    int do_strips;              // 0: flush launch (riders only)
    int chain_timed;            // 1: the caller brackets the whole chain with one TimedScope (adi_fast_chain_cost)
    int nt;                     // column tiles per tile workgroup (1, 2 or 4; 0 = 1): adi_fast_pick
    int mode;                   // 0: K-split tile workgroups (small n: latency); 1: strip per wave over the full K, B through LDS (large n: throughput)
};
// column tiles per workgroup: as many as keep at least ~2 tile workgroups per CU in the launch
inline int adi_fast_pick_nt(int n, int k) {
    const long tiles = 2L * adi_fast_nstrip(n) * ((k + 15) / 16);
    // n = 1357 (170 half strips), sweep with DRE_ADI_FAST_NT: 1 -> 40.9 us per launch, 2 -> 35.1, 4 -> 37.6; n = 371 (48 half strips): 7.65 / 8.7 / 11.1
    return tiles > 2048 ? 4 : (tiles > 400 ? 2 : 1);
}
// kernel variant and column tiles per workgroup: the K-split tiles win while a launch is latency bound (n < 768 or residuals narrower than 128 columns); beyond that the LDS-staged
// full-K strips with two column tiles per wave keep >= 1 wave per SIMD busy without re-gathering R
inline void adi_fast_pick(int n, int k, int* mode, int* nt) {
    static const int wide_min_n = 768;
    if (n >= wide_min_n && k >= 128) { *mode = 1; *nt = 2; }       // measured at n = 1357: k = 64: 41 us (K-split) vs 53 us; k = 160: 77 vs 55 us; k = 294: 131 vs 80 us
    else {
        static const int force_nt = 0;
        *mode = 0; *nt = force_nt > 0 ? force_nt : adi_fast_pick_nt(n, k);
    }
}
void adi_fast_iter(Ctx* ctx, const AdiFastArgs& a);
// Group chain (dense.hip, k_adi_group): g ADI iterations per launch on the group stack [Om_0 .. Om_{g-1}; Pi_1 .. Pi_g]
#define ADI_GROUP_MAX_K 256
#define ADI_GROUP_MAX_G 6
struct AdiGroupArgs {
    int n, k, nstrip, kst, g;
    const double* Gpack;        // packed group stack of this launch's start position: [2 g][nstrip][kst][64]
    const double* Rpc;          // the group's input residual R_0, packed (B-operand lane order)
    double* Rring; int ldr;     // column-major residuals R_1 .. R_g: R_i at Rring + (i - 1) k ldr
    double* Rpk; size_t rpd;    // the same, packed: R_i at Rpk + (i - 1) rpd
    double* V; int ldv;         // V_0 .. V_{g-1}: V_i at V + i k ldv
    const double* Rp_prev; int n_prev;      // packed residuals the previous launch produced (Gram matrices now) ...
    double* G_prev;                         // ... into n_prev matrices of k x k
    const double* G_prev2; int n_prev2;     // Gram matrices the previous launch produced: norms + decisions now
    int it0_prev2;                          // shifts consumed after the first of those residuals
    const double* T; int ldt; double alpha;
    AdiState* st; double* nws;
    int do_strips;                          // 0: flush launch (riders only)
};
void adi_group_iter(Ctx* ctx, const AdiGroupArgs& a);
void adi_group_cost(const AdiGroupArgs& a, double* bytes, double* flops);
void adi_group_pack(Ctx* ctx, int n, int nblk, const double* src, int lds_, double* out);   // nblk row blocks of n x n (column-major, leading dimension lds_) -> packed
inline size_t adi_fast_rpack_doubles(int n, int k) { return (size_t)4 * adi_fast_nstrip(n) * ((k + 15) / 16) * 64; }
void adi_fast_pack_r(Ctx* ctx, int n, int k, const double* R, int ldr, double* Rp, const AdiState* st);      // column-major -> packed
void adi_fast_cost(const AdiFastArgs& a, double* bytes, double* flops);     // algorithmic bytes / flops of one launch
double ldlt_norm_host(Ctx* ctx, const Mat& L, const Mat& D, double alpha);  // synchronising

// --- blocked Householder QR (compact WY) ----------------------------------------------------
struct QRFact {
    int m = 0, n = 0, kq = 0, nb = 16;
    Mat V;   // m x kq, explicit unit-lower-trapezoidal reflectors (zeros above the diagonal)
    Mat VT;  // m x kq, V_p * T_p per panel (so that Q_p = I - VT_p V_p')
    Mat T;   // nb x kq, upper-triangular block-reflector factors per panel
    Mat R;   // kq x n upper trapezoid
    // tall matrices: panels are aggregated in groups of `group` columns (a multiple of nb); VTg(:, group) = V_g * T_g, so
    // that the wide trailing updates and Q applications stream the big operand once per GROUP instead of once per panel
    int group = 0;
    Mat VTg;
};
// A (m x n) is destroyed.  (/root/reference/src/LDLt.jl:237-245 `orthf`: any orthogonal-triangular
// factorisation gives the same X up to roundoff; no pivoting is needed because rank decisions are
// taken by the eigenvalue threshold afterwards.)
QRFact qr_factor(Ctx* ctx, Mat& A);
void qr_apply_q(Ctx* ctx, const QRFact& qr, Mat& B, bool transpose);  // B <- Q*B or Q'*B, B is m x r

// --- symmetric eigensolver --------------------------------------------------------------------
struct SymEig {
    int q = 0;       // order of S
    int j = 0;       // order of the reduced tridiagonal problem (early termination), j <= q
    int nref = 0;    // Householder reflectors generated
    std::vector<double> w;  // j eigenvalues (host), UNSORTED (column i of Z belongs to w[i])
    Mat Z;           // j x j eigenvectors of the tridiagonal matrix (device)
    Mat V;           // q x q Householder vectors of the reduction (device, column i = reflector i)
    DevArr<double> tau;
    DevArr<double> d, e;   // the tridiagonal matrix T_j (device)
    double snorm = 0.0;    // ||S||_F
};
// Householder tridiagonalisation of S (q x q, full symmetric storage, destroyed) that stops as soon as
// the not-yet-reduced trailing block is below tolfac*eps*||S||_F, then implicit QL on the tridiagonal.
// want_eig = false stops after the reduction: S ~ Q_h(:,1:j) T_j Q_h(:,1:j)' with T_j = tridiag(d, e).
// deflate: an off-diagonal entry of the tridiagonal problem at or below deflate * eps * ||S||_F counts as zero (orders above 128, where the QL iteration's
// scalar chain runs on the host; 1e-3 = the device kernels' constant).  1: a perturbation of the size of the reduction's own backward error — for callers
// that only want the eigenvalues ABOVE 100 eps max|lambda| (COMPRESS_TIGHT): the iteration no longer resolves the cluster of rounding-noise eigenvalues
SymEig sym_eig(Ctx* ctx, Mat& S, double tolfac = 4.0, bool want_eig = true, double abs_tol = -1.0, bool tol_is_floor = false, double deflate = 1e-3);
Mat sym_tridiag_dense(Ctx* ctx, const SymEig& e);    // T_j as a dense j x j matrix
// B (q x r) <- Q_h * [Zsel; 0]   where Zsel = Z(:, ids) (ids on host); with an empty Z the identity is used
Mat sym_eig_backtransform(Ctx* ctx, const SymEig& e, const std::vector<int>& ids);

// --- blocked two-sided Householder band reduction with early termination (multi-CU, GEMM-rich) ------------
// S ~ Qb(:,1:J) * Dband * Qb(:,1:J)'  with Dband the leading J x J block (band width nb) of Qb' S Qb; the reduction
// stops at the first panel boundary J where everything not yet reduced is below tolfac*eps*||S||_F.
struct SymBand {
    int q = 0, J = 0, nb = 16, npanels = 0;
    Mat V;      // q x q explicit reflectors; panel p occupies columns [p*nb, ...) and rows >= (p+1)*nb
    Mat VT;     // V_p * T_p per panel
    Mat T;      // nb x q block-reflector factors
    Mat D;      // J x J symmetric band matrix (dense storage)
    Mat V0, VT0; // optional leading reflector block Q0 = I - VT0 V0' (factor-form reduction with a warm start): Qb <- Q0 Qb
    Mat B0;      // optional q x J identity block written by the launch that extracted D (start of sym_band_basis: saves its fill launch)
    mutable bool B0_used = false;      // the basis is built in place: the block serves ONE call
};
// abs_tol > 0 replaces the relative criterion by ||remainder||_F <= abs_tol
// spec (optional): the band matrix and the basis for the predicted result are enqueued while the control block is read back (spec->hit
// tells whether they are valid: then out.D == spec->D and spec->B == sym_band_basis(out)); ext_part: ext_nparts partial sums of
// ||S||_F^2 that came with the assembly of S (saves the first norm launch)
struct BandSpec {
    int J = -1; bool hit = false; Mat B, D;
    // further host work to slot into the wait for the control block (runs once, after the speculative kernels were enqueued; ran = true)
    std::function<void()> extra;
    bool ran = false;
    int extra_after = -1;      // >= 1: run `extra` right after that many panels (with their updates) were enqueued instead of inside the read-back
    // optional job of the control-block launch (the dense time loop's tolerances; saves a launch): with nc = sqrt(sum tol_parts),
    // at = tol_abstol >= 0 ? tol_abstol : tol_reltol nc  ->  tols_out = {at, tol_frac at, nc}; the reduction's absolute tolerance is tols_out[1]
    const double* tol_parts = nullptr; int tol_nparts = 0; double tol_reltol = 0.0, tol_abstol = -1.0, tol_frac = 1.0; double* tols_out = nullptr;
};
SymBand sym_band_reduce(Ctx* ctx, Mat& S, double tolfac, double abs_tol = -1.0, const double* abs_tol_dev = nullptr, BandSpec* spec = nullptr,
                        const double* ext_part = nullptr, int ext_nparts = 0, bool tol_is_floor = false);   // abs_tol_dev: the tolerance lives in device memory;
                        // tol_is_floor: tolerance = max(relative, abs_tol) (formation noise of sums with cancellation)
Mat sym_band_basis(Ctx* ctx, const SymBand& b);     // q x J, the first J columns of Qb
// The same reduction for S = L blockdiag(alpha_b D_b) L' given in factor form (L: n x c, overwritten), c + 64 <= n:
// neither S nor a QR of L is formed; the termination norm is a 16-probe randomized estimate (dense.hip).
void lead_rotate(Ctx* ctx, Mat& L, Mat& V0, Mat& VT0);                        // L <- Q0' L,  Q0 = I - VT0 V0' from QR(L[:, 0:16])
void lead_rotate_back(Ctx* ctx, const Mat& V0, const Mat& VT0, Mat& B);      // B <- Q0 B
struct LrBlockD { int off, k, ldd, diag; const double* D; double alpha; };
SymBand lr_band_reduce(Ctx* ctx, Mat& Lx, const std::vector<LrBlockD>& blocks, double tolfac, double abs_tol = -1.0, bool tol_is_floor = false);   // Lx = [L | 16 spare columns]

// warm-started compression of the dense-X time loop's residual (warm.hip): fresh directions Z_1 = (I - QQ') Y_f with their whitening factor, Z and
// Res Z, the small generalized eigenproblem with the truncation decision and the solve's tolerances, R = B Uc with the probe's decision
void warm_project(Ctx* ctx, int n, int q, const Mat& Q, const Mat& Yf, const Mat& Pf, Mat& Z1, double* slab, int* ticket, double* Cw);
void warm_z(Ctx* ctx, int n, const Mat& Z1, const double* Cw, const Mat& Res, Mat& Zb, Mat& Zy, Mat& W2);
void warm_small(Ctx* ctx, int q, int m, int kl, int qn, const Mat& Cc, const double* parts, int nparts, double reltol, double abstol, double frac,
                double budget_frac, double* tols, Mat& Uc, Mat& T, Mat& Cp, int* ticket, Mat* Mout = nullptr, Mat* LTout = nullptr);
void warm_finish(Ctx* ctx, int n, int q, int kl, const Mat& Q, const Mat& Wk, const Mat& Yp, const Mat& Pp, Mat& R, double* slab, int* ticket, double* tols);
void warm_ctl(Ctx* ctx, double* tols, int* ticket, int J);
void warm_eig(Ctx* ctx, int m, const Mat& S, Mat& U);
int warm_project_slabs(int n);                                 // workgroups of warm_project (slab: 256 doubles each)
void warm_zapply(Ctx* ctx, int n, const Mat& Z1, const double* Cw, Mat& Zb, Mat& Zy);

}  // namespace dre
