/* libdre_hip — C ABI of the MI355X-native low-rank Rosenbrock/ADI engine.
 *
 * Drop-in boundary for the hot path of mpimd-csc/DifferentialRiccatiEquations.jl v0.5.5
 *   solve(GDREProblem{<:LDLᵀ}, Ros1/Ros2(ADI(...)))  and everything below it.
 * The reference has no FFI of its own (pure Julia, multiple dispatch); each entry point cites the
 * reference method it replaces (paths relative to the reference tree).  The Julia shim that binds
 * these symbols with `ccall` is differentialriccatiequations.jl_amd/julia/DREHip.jl; the ctypes
 * binding used by the tests is differentialriccatiequations.jl_amd/_lib.py.  See INTEGRATION.md.
 *
 * Conventions
 *  - every function returns an int32 status: 0 ok, <0 error (dre_last_error(ctx) has the text);
 *    no exception crosses the boundary;
 *  - dense matrices are column-major Float64 (Julia `Matrix{Float64}`), sparse matrices are passed
 *    as the three arrays of a Julia `SparseMatrixCSC{Float64,Int64}` (colptr, rowval, nzval; 1-based
 *    with index_base = 1) — CSC of M is byte-for-byte CSR of M', which is what every product on
 *    the path needs (E'V, A'L, (A'+pE')\R);
 *  - host buffers belong to the caller, device objects to the context; every object has a *_free;
 *  - one context per GPU / host thread; all work of a context is ordered on its private HIP stream;
 *  - there is NO CPU fallback: without a usable HIP device dre_ctx_create fails with DRE_ERR_NODEVICE.
 */
#ifndef DRE_HIP_H
#define DRE_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DRE_OK 0
#define DRE_ERR_INVALID (-1)
#define DRE_ERR_HIP (-2)
#define DRE_ERR_ALLOC (-3)
#define DRE_ERR_SINGULAR (-4)
#define DRE_ERR_INTERNAL (-5)
#define DRE_ERR_NODEVICE (-6)

/* hard limits of the engine */
#define DRE_ADI_MAX_ITERS 100000      /* largest dre_adi_options.maxiters (the device keeps the norm history as a ring that the host empties per chunk) */
#define DRE_SMW_MAX_RANK 32           /* columns of the low-rank factors U, V of F = cA*A + cE*E + inv(alpha)*U*V */

/* warning bits reported by ADI (AdiResult.warnings) */
#define DRE_WARN_NOT_CONVERGED 1      /* src/lyapunov/adi.jl:125-126 */
#define DRE_WARN_ZERO_INCREMENT 2     /* src/lyapunov/adi.jl:134-137,200-204: a complex pair's solve returned V = 0 exactly; the increment is zero,
                                         X and the residual are left alone and the iteration stops with the pair's two shifts counted */
#define DRE_WARN_RITZ_DISCARDED 4     /* src/shifts/helpers.jl:133 */
#define DRE_WARN_RITZ_FLIPPED 8       /* src/shifts/helpers.jl:136 */
#define DRE_WARN_PIVOT_GROWTH 16      /* the pivot-free sparse LU met multipliers above "pivot_growth_warn" (default 1e8); the convergence claim of
                                         this solve was re-checked against the residual evaluated from scratch (see dre_shift_factor) */

typedef struct dre_ctx dre_ctx;
typedef struct dre_dense dre_dense;
typedef struct dre_pencil dre_pencil;
typedef struct dre_factor dre_factor;
typedef struct dre_ldlt dre_ldlt;
typedef struct dre_adi_result dre_adi_result;
typedef struct dre_gdre_result dre_gdre_result;

/* ---- context ------------------------------------------------------------------------------ */
int dre_version(void);
int dre_ctx_create(int device, dre_ctx** out);
int dre_ctx_destroy(dre_ctx* ctx);
const char* dre_last_error(dre_ctx* ctx);
int dre_ctx_sync(dre_ctx* ctx);
int dre_ctx_info(dre_ctx* ctx, int64_t* info /* [0]=CUs [1]=pool bytes */);
/* Tunables of the engine (no counterpart in the reference; they select between device code paths that compute the same result):
 *   "dense_inverse_max_n"  real ADI shifts on pencils with n <= value apply a cached dense inverse of A' + mu E' by MFMA GEMM
 *                          instead of the multifrontal triangular sweeps (default 1536, or env DRE_DENSE_INV_MAX_N; 0 disables).
 *   compress! (LDLt.jl:204-225) without a QR of the factor — the band reduction runs in the natural coordinates:
 *   "compress_direct_max_n"     n <= value and columns >= n / compress_direct_ratio: S = L D L' (n x n) is formed directly
 *                               (default 2560; n <= 512 always);  "compress_direct_ratio" (default 8)
 *   "compress_factor_min_n"     n >= value: band reduction in factor form, reflectors applied to L, randomized termination estimate
 *                               (default 2561; a huge value disables);  "compress_factor_min_cols" (default 96) fewer columns: QR path
 *   "compress_sketch"           wide factors (columns >= "compress_sketch_min_cols", default 320, and >= "compress_sketch_ratio", default 1.25, x the sketch width) of sums without
 *                               cancellation: randomized range finder, three GEMM passes over the factor; sketch width = rank of the previous
 *                               compression of this kind + "compress_sketch_extra" (48); rejected sketches fall back (default 1, 0 disables)
 *   "compress_sketch_sparse"    1 (default): the test matrix of the sketch is a structured sparse sign matrix (8 entries +-1/sqrt(8) per row), applied
 *                               in one pass over the factor; 0: Gaussian (dense GEMM).  The acceptance test uses 16 independent Gaussian probe
 *                               columns either way; two probe rejections at an order n switch that order to the Gaussian matrix
 *   "compress_sketch_cholqr"    1 (default): the sketch is orthonormalised in 64-column blocks by block Gram-Schmidt + Cholesky QR, both twice
 *                               (GEMMs and a 64 x 64 Cholesky kernel); a block with cond > ~3e6 is detected on the device and the compression
 *                               redone, after two such events at an order n Householder/TSQR panels are used (0: always)
 *   (env: DRE_COMPRESS_DIRECT_MAX_N, DRE_COMPRESS_DIRECT_RATIO, DRE_COMPRESS_FACTOR_MIN_N, DRE_COMPRESS_FACTOR_MIN_COLS, DRE_COMPRESS_SKETCH*)
 *   "mf_subtree"                1: the multifrontal sweeps below the dense top run as one workgroup per subtree (default 0: one launch per level)
 *   "setup_streams"             helper streams for the factorisations / dense inverses of the shifts of a Cyclic list (default 5; 0, 1: none)
 *   "top_inverse_max_rows"      multifrontal solves with a real factor that keeps being reused (third multi-column solve on): the top
 *                               levels of the elimination tree with at most this many pivot variables are applied as ONE dense inverse
 *                               of their Schur complement (MFMA GEMM) instead of level-by-level sweeps (default 1536, 0 disables;
 *                               env DRE_TOP_INVERSE_MAX_ROWS)
 *   "pivot_growth_warn" / "pivot_growth_fail"   thresholds on the largest multiplier of the pivot-free sparse LU (defaults 1e8 / 1e13), see dre_shift_factor
 *   "dense_x_max_n"            Ros1 without save_state, real Cyclic shifts, n <= value (default 1536): X is carried as a dense symmetric n x n matrix
 *                               between the time steps and converted to LDL' form once at the end (env DRE_DENSE_X_MAX_N; 0 disables)
 *   "x_side_stream"             Ros1, n <= 1536, no save_state: X is carried as "compressed warm start + ADI increments" and its
 *                               compression (adi.jl:78-80) runs on a second stream beside the next time step (default 1; env
 *                               DRE_X_SIDE_STREAM);  "x_compress_every" = s (default 1) with x_side_stream = 0: single stream,
 *                               X compressed every s-th step only */
int dre_ctx_set_option(dre_ctx* ctx, const char* name, double value);
/* the current value of an option of dre_ctx_set_option (tests snapshot and restore what they change; no reference counterpart: the reference's
 * tunables are keyword arguments) */
int dre_ctx_get_option(dre_ctx* ctx, const char* name, double* value);
/* per-kernel-class timing with HIP events on the library stream (replaces TimerOutputs.@timeit_debug,
 * src/DifferentialRiccatiEquations.jl:22 and the sections listed in SURVEY.md §5) */
int dre_prof_enable(dre_ctx* ctx, int on);
int dre_prof_reset(dre_ctx* ctx);
int dre_prof_count(dre_ctx* ctx, int* n);
int dre_prof_get(dre_ctx* ctx, int i, char* name, int name_len, double* ms, int64_t* launches, double* bytes, double* flops);

/* ---- dense matrices (array backend: similar/adapt/copyto!) ------------------- */
int dre_dense_upload(dre_ctx* ctx, int rows, int cols, const double* host, int ld, dre_dense** out);
int dre_dense_create(dre_ctx* ctx, int rows, int cols, dre_dense** out);            /* zero-filled */
int dre_dense_download(dre_ctx* ctx, const dre_dense* a, double* host, int ld);
/* device-to-device interop with buffers the caller owns (the exchange buffers of the multi-GPU layer, e.g. a torch tensor's data_ptr):
 * column-major with leading dimension ld; both synchronise the context's stream */
int dre_dense_from_device(dre_ctx* ctx, int rows, int cols, const double* src_dev, int ld, dre_dense** out);
int dre_dense_to_device(dre_ctx* ctx, const dre_dense* a, double* dst_dev, int ld);
int dre_dense_shape(const dre_dense* a, int* rows, int* cols);
int dre_dense_free(dre_ctx* ctx, dre_dense* a);

/* ---- the pencil (E, A): union pattern, nested-dissection ordering, symbolic multifrontal analysis,
 *      done once per problem because the pattern of A' + pE' never changes
 *      (replaces the per-step `factorize` analysis of src/blocklinear/backslash.jl:13) --------- */
int dre_pencil_create(dre_ctx* ctx, int n, const int64_t* E_colptr, const int64_t* E_rowval, const double* E_nzval,
                      const int64_t* A_colptr, const int64_t* A_rowval, const double* A_nzval, int index_base,
                      int leaf_size, dre_pencil** out);
/* host-only variant (no GPU touched): symbolic analysis for inspection / CPU tests */
int dre_pencil_create_host(int n, const int64_t* E_colptr, const int64_t* E_rowval, const double* E_nzval,
                           const int64_t* A_colptr, const int64_t* A_rowval, const double* A_nzval, int index_base,
                           int leaf_size, dre_pencil** out);
int dre_pencil_free(dre_pencil* p);
/* info: [0]=n [1]=nnz(union) [2]=tree nodes [3]=levels [4]=max front [5]=max separator [6]=factor nnz [7]=fronts slab entries */
int dre_pencil_info(const dre_pencil* p, int64_t* info);
/* integer arrays of the symbolic structure by name ("perm","iperm","ptr","idx","first","size","parent","level",
 * "child_ptr","child_idx","bptr","bidx","cmap_ptr","cmap","front_off","inv_off","upd_off","asm_dest","lvl_ptr","lvl_nodes") */
int dre_pencil_get_array(const dre_pencil* p, const char* name, int64_t* out, int64_t cap, int64_t* len);
int dre_pencil_get_values(const dre_pencil* p, int which /*0 = E', 1 = A'*/, double* out, int64_t cap);

/* ---- kernels exposed one by one (extension points of SURVEY.md §8b; parity tests call these) -- */
/* C = alpha*op(A)*op(B) + beta*C on the f64 MFMA path (LinearAlgebra.mul!) */
int dre_gemm(dre_ctx* ctx, int transA, int transB, double alpha, const dre_dense* A, const dre_dense* B, double beta, dre_dense* C);
/* Y = alpha*M'*X + beta*Y, M = E (which=0) or A (which=1): mul!(R, E', V, a, b) src/lyapunov/adi.jl:171,217 */
int dre_spmm(dre_ctx* ctx, const dre_pencil* p, int which, double alpha, const dre_dense* X, double beta, dre_dense* Y);
/* orthf(L) -> Q, R  (src/LDLt.jl:237-245) */
int dre_orthf(dre_ctx* ctx, const dre_dense* L, dre_dense** Q, dre_dense** R);
/* The reference's EXTENSION POINT `DifferentialRiccatiEquations.orthf(L) -> (Q, R)` (src/LDLt.jl:227-245; test/cuda.jl:32-37 substitutes an SVD):
 * a user-supplied orthogonalisation for this context.  L is n x c (device, column-major, leading dimension ldl); the callback fills Q (n x p,
 * orthonormal columns) and R (p x c) with p = min(n, c) and L = Q R, and returns 0.  It is called with the context's stream idle and must return
 * when its own work is complete.  Honoured wherever the engine runs the reference's arithmetic: the literal compression (ADI option
 * compress_exact, dre_ldlt_compress: src/LDLt.jl:211) and dre_ldlt_norm (src/LDLt.jl:84).  fn = NULL restores the library's Householder QR. */
typedef int (*dre_orthf_fn)(void* user, int n, int c, const double* L, int ldl, double* Q, int ldq, double* R, int ldr);
int dre_ctx_set_orthf(dre_ctx* ctx, dre_orthf_fn fn, void* user);
/* eigen(Symmetric(S)) with early-terminating tridiagonalisation (src/LDLt.jl:214); returns the j computed
 * eigenpairs (all those above tolfac*eps*||S||_F in magnitude), values ascending */
int dre_sym_eig(dre_ctx* ctx, const dre_dense* S, double tolfac, dre_dense** values, dre_dense** vectors);
/* factorize(cA*A' + (cE_re + i cE_im)*E')  (src/blocklinear/backslash.jl:8-15); complex iff cE_im != 0.
 * The multifrontal LU does NOT pivot (the reference's UMFPACK / CHOLMOD do): it is exact-arithmetic safe for the pencils of the path
 * (-(A + pE) symmetric positive definite for real p < 0, complex symmetric with definite parts otherwise) and for diagonally dominant
 * ones.  For anything else the largest multiplier |l_ik| is tracked: above "pivot_growth_fail" (default 1e13; dre_ctx_set_option) the
 * factorisation is rejected with DRE_ERR_SINGULAR, above "pivot_growth_warn" (default 1e8) ADI results carry DRE_WARN_PIVOT_GROWTH and
 * their convergence claim is verified against the true residual.  Pencils that need pivoting take a user block solver
 * (dre_adi_options.inner_solve). */
int dre_shift_factor(dre_ctx* ctx, const dre_pencil* p, double cA, double cE_re, double cE_im, dre_factor** out);
/* X = F \ B  (src/blocklinear/backslash.jl:17-21); X_im may be NULL for a real factor */
int dre_shift_solve(dre_ctx* ctx, const dre_factor* f, const dre_dense* B, dre_dense** X_re, dre_dense** X_im);
/* X = (M + inv(alpha) * Vt * U') \ B  with M the factorised sparse matrix: ShermanMorrisonWoodbury(Backslash) applied to
 * BlockLinearProblem(LowRankUpdate(M, alpha, Vt, U'), B)  (src/blocklinear/sherman-morrison-woodbury.jl:10-45, src/LowRankUpdate.jl:61-64).
 * U and Vt are n x m (m <= DRE_SMW_MAX_RANK); complex iff the factor is. */
int dre_shift_solve_smw(dre_ctx* ctx, const dre_factor* f, double alpha, const dre_dense* U, const dre_dense* Vt, const dre_dense* B,
                        dre_dense** X_re, dre_dense** X_im);
int dre_factor_growth(dre_ctx* ctx, const dre_factor* f, double* growth);   /* largest multiplier met by the LU */
/* Static pivoting (the reference's factorize pivots: UMFPACK / CHOLMOD, src/blocklinear/backslash.jl:13): pivots below
 * pivot_static * max|entry| (option "pivot_static", default sqrt(eps); 0 = plain pivot-free LU) are replaced by that value, which bounds
 * the multipliers by 1 / pivot_static; solves with a factor that has replaced pivots run "pivot_refine_steps" (default 3) steps of
 * fixed-point refinement against the true operator (real shifts; a complex factor with replaced pivots is refused: user block solver).
 * count = number of replaced pivots of this factor. */
int dre_factor_perturbed(dre_ctx* ctx, const dre_factor* f, int64_t* count);
int dre_factor_free(dre_ctx* ctx, dre_factor* f);

/* ---- LDLᵀ objects (src/LDLt.jl) ------------------------------------------------------------ */
/* lowrank(L, D) scaled by alpha; rows are permuted to the pencil's ordering when p != NULL */
int dre_ldlt_create(dre_ctx* ctx, const dre_pencil* p, const dre_dense* L, const dre_dense* D, double alpha, dre_ldlt** out);
int dre_ldlt_zero(dre_ctx* ctx, const dre_pencil* p, int n, dre_ldlt** out);
int dre_ldlt_free(dre_ctx* ctx, dre_ldlt* x);
int dre_ldlt_info(const dre_ldlt* x, int* n, int* rank, int* nblocks);
int dre_ldlt_add(dre_ctx* ctx, const dre_ldlt* a, const dre_ldlt* b, dre_ldlt** out);          /* LDLt.jl:131-148 */
int dre_ldlt_scale(dre_ctx* ctx, const dre_ldlt* a, double alpha, dre_ldlt** out);             /* LDLt.jl:156-159 */
int dre_ldlt_concatenate(dre_ctx* ctx, dre_ldlt* x);                                           /* LDLt.jl:174-191 */
int dre_ldlt_compress(dre_ctx* ctx, dre_ldlt* x);                                              /* LDLt.jl:204-225 */
/* compress! with an ABSOLUTE truncation tolerance (Frobenius norm of what may be dropped) instead of the relative one: used where the
 * caller only compares norm(x) with a tolerance (Riccati / Lyapunov residuals near convergence are pure cancellation noise relative to
 * their own size, so a relative criterion keeps all of it).  abs_tol <= 0 behaves like dre_ldlt_compress_fast. */
int dre_ldlt_compress_tol(dre_ctx* ctx, dre_ldlt* x, double abs_tol);
/* The engine's own compression (what the ADI and Rosenbrock loops use between the API boundaries): early-terminating band reduction — on
 * S = L D L', in factor form, or through a randomized range finder for wide factors — truncated at 4 eps ||X||_F; the result has an
 * orthonormal factor and a band matrix D instead of diag(eigenvalues).  Same X up to that tolerance. */
int dre_ldlt_compress_fast(dre_ctx* ctx, dre_ldlt* x);
int dre_ldlt_norm(dre_ctx* ctx, dre_ldlt* x, double* out);                                     /* LDLt.jl:77-89 */
/* bring an engine result to the reference's canonical form: one component, D = diag(eigenvalues), |lambda| >= 100 eps max|lambda| */
int dre_ldlt_canonicalize(dre_ctx* ctx, dre_ldlt* x);
/* dot(X1, X2) = <X1, X2>_F (LDLt.jl:91-108): Gram products and the two small congruences on the device; both operands on the same pencil */
int dre_ldlt_dot(dre_ctx* ctx, const dre_ldlt* a, const dre_ldlt* b, double* out);
/* alpha, L, D = X  (LDLt.jl:54-60; compresses when more than one component); pass NULL buffers to query sizes */
int dre_ldlt_destructure(dre_ctx* ctx, dre_ldlt* x, double* alpha, double* L_host, int ldl, double* D_host, int ldd);

/* ---- GALE / ADI (src/lyapunov/types.jl:10-32, adi.jl:29-225) --------------------------------- */
/* Pluggable inner solver (src/blocklinear/types.jl:15-62; src/lyapunov/types.jl:26 `inner_alg`; example test/cuda.jl:23-30,74):
 * a user-supplied solver of the SPARSE shifted system
 *      (cA*A' + (cE_re + i*cE_im)*E') X = B,        B real n x nrhs, column-major, leading dimension n,
 * i.e. the ALG of inner_alg = ShermanMorrisonWoodbury(ALG, Backslash): the engine keeps doing the rank-m Sherman-Morrison-Woodbury
 * correction for F = Fs + inv(alpha) U V around it.  All pointers are DEVICE pointers in the caller's row ordering; X_im is NULL for a
 * real system.  The engine synchronises its stream before the call; the callback must have completed its own device work when it
 * returns.  Return 0 on success (anything else aborts the solve with DRE_ERR_INTERNAL).  With a user solver the engine neither caches
 * factorisations nor takes the dense-inverse fast paths. */
typedef int (*dre_block_solver_fn)(void* user, int n, int nrhs, double cA, double cE_re, double cE_im, const double* B, double* X_re, double* X_im);

/* User-defined shift strategy: the reference's extension point  Shifts.init(strategy, prob) / Shifts.update!(shifts, X, R, Vs...) /
 * Shifts.take!(shifts)  (src/Shifts.jl:79-116; a strategy that produces batches implements take_many! behind a BufferedIterator,
 * src/shifts/helpers.jl:60-89; example: the Dummy strategy of test/Shifts.jl:133-163).  With shift_kind = 3 the engine calls shift_fn whenever
 * its buffer of shifts is empty:
 *     restart   1 on the first call of a Lyapunov solve (= init), 0 afterwards
 *     hist      DEVICE pointer, n x hist_cols column-major (leading dimension ldh), rows in the CALLER's ordering: what update! has handed over — the residual factor R at the
 *               start of a solve, afterwards the last n_history increments V (a conjugate pair contributes its two real blocks), oldest first
 *     capacity  room in re / im (HOST arrays); write *count >= 1 shifts, in the order in which they are to be used
 * Every shift must have a negative real part; a complex shift mu is followed by conj(mu) — the pair is handled by ONE double step and the
 * second take! is checked against the first (adi.jl:181-225, :190).  The engine synchronises its stream before the call.  Return 0 on success (anything else aborts the solve with DRE_ERR_INTERNAL). */
typedef int (*dre_shift_fn)(void* user, int restart, int n, int hist_cols, const double* hist, int ldh, int capacity, double* re, double* im, int* count);

typedef struct dre_adi_options {
    int32_t maxiters;              /* 100 */
    double reltol;                 /* < 0 = nothing -> n*eps */
    double abstol;                 /* < 0 = nothing -> reltol*norm(C) */
    int32_t ignore_initial_guess;  /* 0 */
    int32_t compression_interval;  /* 10 */
    int32_t compression;           /* 1 */
    int32_t shift_kind;            /* 0 = Cyclic(values), 1 = Projection(n_history), 2 = Cyclic(Heuristic(nshifts, kplus, kminus)) recomputed
                                      on the device from (E, F) at the start of every Lyapunov solve (adi.jl:54, heuristic.jl:39-66),
                                      3 = user-defined strategy: shift_fn below (n_history = number of increments it is shown) */
    int32_t n_history;             /* 2 */
    int32_t nshifts;               /* Cyclic: number of values (conjugate pairs adjacent); Heuristic: number of shifts to select */
    const double* shifts_re;
    const double* shifts_im;       /* may be NULL (all real) */
    double compress_tolfac;        /* <= 0 -> 4: the engine's compressions drop what lies below tolfac*eps*||S||_F */
    int32_t compress_exact;        /* 1: eigen-decomposition + 100*eps*max|lambda| threshold at every compression (reference arithmetic);
                                      0 (default): Krylov-truncated compression, D stays tridiagonal inside the engine */
    int32_t heuristic_kplus;       /* shift_kind 2: Arnoldi steps with E^-1 F */
    int32_t heuristic_kminus;      /* shift_kind 2: Arnoldi steps with F^-1 E */
    dre_block_solver_fn inner_solve; /* NULL (default): inner_alg = Backslash() on the library's multifrontal LU (src/blocklinear/backslash.jl) */
    void* inner_user;              /* passed back to inner_solve */
    dre_shift_fn shift_fn;         /* shift_kind 3 only */
    void* shift_user;              /* passed back to shift_fn */
} dre_adi_options;
int dre_adi_default_options(dre_adi_options* opt);

/* solve(GALEProblem(E, F, C), ADI(...); initial_guess) with F = cA*A + cE*E + inv(lr_alpha)*U*V
 * (LowRankUpdate, src/LowRankUpdate.jl:18-39).  U is n x m, Vt = V' is n x m; both NULL for a plain sparse F. */
int dre_gale_solve(dre_ctx* ctx, const dre_pencil* p, double cA, double cE, double lr_alpha, const dre_dense* U,
                   const dre_dense* Vt, dre_ldlt* C, const dre_ldlt* X0, const dre_adi_options* opt, dre_adi_result** out);
/* The same solve as the reference's stepwise protocol on the solver object (ADICache, src/lyapunov/adi.jl:5-21):
 *   dre_adi_init   = CommonSolve.init(::GALEProblem, ::ADI; initial_guess)   adi.jl:29-69   (residual, tolerances, shift oracle)
 *   dre_adi_step   = step!(cache)          adi.jl:97-128  — ONE shift, or one conjugate pair (counts two, adi.jl:181-225), then the
 *                                                           residual norm and the convergence test (adi.jl:115-123)
 *   dre_adi_isdone = isdone(cache)         adi.jl:130-141
 *   dre_adi_solve  = solve!(cache)         adi.jl:71-76   — steps until done (whole chunks are enqueued speculatively)
 *   dre_adi_finish = the tail of solve!    adi.jl:78-89   — final compression, observe_gale_done! payload; call once
 * Stepping to the end and solving in one go run the same kernels in the same order: the results are identical bit for bit
 * (test/tiny_random.jl:48-57).  dre_gale_solve = init + solve + finish. */
typedef struct dre_adi_solver dre_adi_solver;
int dre_adi_init(dre_ctx* ctx, const dre_pencil* p, double cA, double cE, double lr_alpha, const dre_dense* U, const dre_dense* Vt,
                 dre_ldlt* C, const dre_ldlt* X0, const dre_adi_options* opt, dre_adi_solver** out);
int dre_adi_step(dre_ctx* ctx, dre_adi_solver* s);
int dre_adi_solve(dre_ctx* ctx, dre_adi_solver* s);
int dre_adi_isdone(const dre_adi_solver* s, int* done);
int dre_adi_state(const dre_adi_solver* s, int64_t* iters, double* res_norm, double* abstol);     /* shifts consumed, last residual norm, abstol */
/* observe_gale_step!(observer, i, X, residual, residual_norm) (src/lyapunov/adi.jl:119, src/Callbacks.jl:97-107): the iterate X and the
 * residual object of the solver's CURRENT iteration as LDLᵀ handles (either pointer may be NULL).  X shares its factors with the solver
 * (lazy list of increments, LDLt.jl:131-148); the residual factor is a copy (the iteration updates it in place, adi.jl:171). */
int dre_adi_snapshot(dre_ctx* ctx, dre_adi_solver* s, dre_ldlt** X, dre_ldlt** residual);
/* observe_gale_metadata!(observer, "ADI shifts", mu) (adi.jl:103,192) while stepping: the shifts of the accepted iterations from `from`
 * (0-based) on; re/im may be NULL to ask for the count only (they must hold *count doubles otherwise). */
int dre_adi_shifts(const dre_adi_solver* s, int64_t from, int64_t* count, double* re, double* im);
int dre_adi_finish(dre_ctx* ctx, dre_adi_solver* s, dre_adi_result** out);
int dre_adi_free(dre_adi_solver* s);
/* Penzl's heuristic, device part (src/shifts/heuristic.jl:39-66,103-130): Ritz values of E^-1 F (kplus Arnoldi steps) and of F^-1 E
 * (kminus steps), both from ones(n), for F = cA*A + cE*E + inv(lr_alpha)*U*V (products and solves see the low-rank part, the solves
 * through Sherman-Morrison-Woodbury like heuristic.jl:51-60).  Outputs: kplus + kminus complex numbers, raw (unsorted, unstabilised);
 * the selection `heuristic(R, nshifts)` (heuristic.jl:22-37,82-101) is host logic of the shim. */
int dre_heuristic_ritz(dre_ctx* ctx, const dre_pencil* p, double cA, double cE, double lr_alpha, const dre_dense* U, const dre_dense* Vt,
                       int kplus, int kminus, double* plus_re, double* plus_im, double* minus_re, double* minus_im);
/* residual(GALEProblem, X)  (src/lyapunov/residual.jl:3-31) */
int dre_gale_residual(dre_ctx* ctx, const dre_pencil* p, double cA, double cE, double lr_alpha, const dre_dense* U,
                      const dre_dense* Vt, dre_ldlt* C, dre_ldlt* X, dre_ldlt** out);
/* LyapunovOperator(E, F) * X = F'XE + E'XF as an LDL' object with factor [E'L, F'L] (src/lyapunov/gmres.jl:108-120); F as in dre_gale_solve */
/* residual(::GAREProblem, ::LDLt) = gamma C'SC + A'XE + E'XA - beta^2 E'XB Rinv B'XE as one LDL' object [C', A'L, E'L] T [...]'
 * (src/riccati/residual.jl:5-52), and the feedback K' = E'XB (newton.jl:104-112), both formed from the device factors of X */
int dre_gare_residual(dre_ctx* ctx, const dre_pencil* p, dre_ldlt* X, const dre_dense* Ct, const dre_dense* S, double gamma, const dre_dense* B,
                      const dre_dense* Rinv, double beta, dre_ldlt** out);
int dre_ldlt_feedback(dre_ctx* ctx, const dre_pencil* p, dre_ldlt* X, const dre_dense* B, dre_dense** Kt);
int dre_gale_apply(dre_ctx* ctx, const dre_pencil* p, double cA, double cE, double lr_alpha, const dre_dense* U, const dre_dense* Vt,
                   const dre_ldlt* X, dre_ldlt** out);
/* info: [0]=iters [1]=converged [2]=warnings [3]=number of recorded norms [4]=rhs columns;  dinfo: [0]=res_norm [1]=abstol [2]=initial norm */
int dre_adi_result_info(const dre_adi_result* r, int64_t* info, double* dinfo);
int dre_adi_result_history(const dre_adi_result* r, double* norms, int32_t* norm_iters, double* shifts_re, double* shifts_im);
int dre_adi_result_take_x(dre_adi_result* r, dre_ldlt** X);            /* observe_gale_done!(…, X, …) */
int dre_adi_result_take_residual(dre_adi_result* r, dre_ldlt** R);
int dre_adi_result_free(dre_adi_result* r);

/* ---- multi-GPU: RCCL over xGMI inside the library (SURVEY.md section 8b `dre_comm_init`, 8e) ----
 * One process per GPU, one context per process.  Rank 0 makes the 128-byte unique id (ncclGetUniqueId), the host program hands it to
 * the other ranks (torch.distributed / MPI / a file), every rank calls dre_comm_init(ctx, nranks, rank, id): the communicator belongs
 * to the context and its collectives are enqueued on the context's stream (no host synchronisation, no staging copies).
 * With a communicator of more than one rank attached, every solve entered through this ABI (dre_gdre_solve, dre_gale_solve,
 * dre_adi_*) runs the SAME device-resident loop on every rank with the shifted solves of the ADI iteration
 * (perform_single_step!, src/lyapunov/adi.jl:149-179; the multifrontal sweeps + SMW of src/blocklinear) sharded:
 *  - Cyclic real shifts (fan groups: g consecutive iterations from g INDEPENDENT solves with the same right-hand side): BY SHIFT — rank r
 *    solves the group positions s = r (mod P), so it only ever factorises the shifts it owns (src/blocklinear/backslash.jl:13 once per
 *    owned shift and run), and ONE in-place all-gather per GROUP of g iterations completes the panel [W_1 .. W_g] on every rank;
 *  - any other real-shift step: BY COLUMN — rank r solves its 16-column tiles of the residual block, one all-gather per ADI step;
 * residual update, norm, compression, shifts and the feedback K(t) are replicated and bit-identical on all ranks (the reference has
 * no multi-device path; test/cuda.jl runs one GPU).  librccl is loaded lazily by the first dre_comm_* call.
 * option "shard_min_cols" (default 32): narrower residual blocks are solved replicated; option "shard_emulate" = P: one process plays
 * P ranks one after the other (tests). */
int dre_comm_unique_id(dre_ctx* ctx, void* id128);
int dre_comm_init(dre_ctx* ctx, int nranks, int rank, const void* id128);      /* id128 may be NULL for nranks == 1 (no RCCL object) */
/* The same communicator over a HOST transport: the collectives are staged through pinned host memory and handed to the caller's callbacks
 * (MPI, gloo, ... — hosts without RCCL, and the two-rank tests of the sharded solve on ONE GPU: RCCL refuses two ranks on one device).
 * allgather(user, send, recv, bytes_per_rank): recv holds nranks blocks, send points at this rank's block inside recv (in place);
 * allreduce(user, buf, count): sum of `count` doubles in place.  Both return 0 on success; they are called from the thread that drives
 * the solve, between two stream synchronisations. */
typedef int (*dre_comm_allgather_fn)(void* user, const void* send, void* recv, size_t bytes_per_rank);
typedef int (*dre_comm_allreduce_fn)(void* user, void* buf, size_t count);
int dre_comm_init_host(dre_ctx* ctx, int nranks, int rank, dre_comm_allgather_fn allgather, dre_comm_allreduce_fn allreduce, void* user);
int dre_comm_free(dre_ctx* ctx);
/* info: [0]=nranks [1]=rank [2]=collective calls [3]=bytes received by all-gathers [4]=bytes reduced [5]=emulated ranks */
int dre_comm_info(dre_ctx* ctx, int64_t* info);
/* generic collectives on device buffers of doubles (the K(t) gather of independent replicas uses them: recv holds nranks blocks of count) */
int dre_comm_allgather(dre_ctx* ctx, const void* send_dev, void* recv_dev, size_t count);
int dre_comm_allreduce_sum(dre_ctx* ctx, void* buf_dev, size_t count);

/* ---- GDRE (src/riccati/lowrank_ros1.jl:3-66, lowrank_ros2.jl:3-89) ---------------------------- */
/* solve(GDREProblem(E, A, B, C, X0, (t0, tf)), Ros<order>(ADI(opt)); dt, save_state).  B is n x m, C is q x n. */
int dre_gdre_solve(dre_ctx* ctx, const dre_pencil* p, const dre_dense* B, const dre_dense* C, dre_ldlt* X0, double t0,
                   double tf, double dt, int order, int save_state, const dre_adi_options* opt, dre_gdre_result** out);
/* info: [0]=time points [1]=stored states [2]=total ADI iterations [3]=sparse factorisations [4]=Lyapunov solves [5]=m [6]=n */
int dre_gdre_result_info(const dre_gdre_result* r, int64_t* info);
int dre_gdre_result_times(const dre_gdre_result* r, double* t);
int dre_gdre_result_K(dre_ctx* ctx, const dre_gdre_result* r, int i, double* K_host /* m x n */, int ld);
/* whole trajectory K(t_0..t_end) as nt consecutive m x n column-major blocks written to DEVICE memory owned by the
 * caller (e.g. a torch tensor's data_ptr) so that it can be handed to RCCL without a host round trip */
int dre_gdre_result_K_device(dre_ctx* ctx, const dre_gdre_result* r, double* K_dev);
/* the same nt blocks written to HOST memory (nt * m * n doubles): one export launch and one copy instead of nt calls of dre_gdre_result_K
 * (sol.K of DRESolution, src/riccati/lowrank_ros1.jl:65) */
int dre_gdre_result_K_all(dre_ctx* ctx, const dre_gdre_result* r, double* K_host);
int dre_gdre_result_X(const dre_gdre_result* r, int i, dre_ldlt** X);   /* shares the factors (sol.X[1] === prob.X0) */
/* per Lyapunov solve j: iinfo [0]=iters [1]=converged [2]=warnings [3]=rhs columns; dinfo [0]=res_norm [1]=abstol */
int dre_gdre_result_gale(const dre_gdre_result* r, int j, int64_t* iinfo, double* dinfo);
/* per-iteration record of Lyapunov solve j for the observer replay (observe_gale_step! / observe_gale_metadata!, src/lyapunov/adi.jl:65,103,119,192):
 * counts [0]=number of recorded norms (index 0 = initial residual) [1]=shifts consumed; pass NULL arrays to query the counts */
int dre_gdre_result_gale_history(const dre_gdre_result* r, int j, int64_t* counts, double* norms, int32_t* norm_iters, double* shifts_re, double* shifts_im);
/* every Lyapunov solve of the result at once: iinfo 6 per solve ([0..3] as dre_gdre_result_gale, [4]=recorded norms, [5]=shifts consumed), dinfo 2 per
 * solve; the histories concatenated in solve order.  Call with NULL history arrays first to size them from iinfo. */
int dre_gdre_result_gales_all(const dre_gdre_result* r, int64_t* iinfo, double* dinfo, double* norms, int32_t* norm_iters, double* shifts_re,
                              double* shifts_im);
int dre_gdre_result_free(dre_gdre_result* r);

/* ---- host helpers exposed for CPU tests of the Projection shift pipeline ---------------------- */
int dre_host_eigvals(int n, const double* A, double* wr, double* wi);
int dre_host_gen_eigvals(int n, const double* A, const double* E, double* wr, double* wi);
int dre_host_svd_left(int p, int w, const double* R, double* U, double* sv);

#ifdef __cplusplus
}
#endif
#endif
