#include <chrono>
#include <thread>
#include <mutex>
#include <condition_variable>
#include <exception>
// Device-resident low-rank Rosenbrock/ADI engine (see engine.hpp).
#include "engine.hpp"
#include "comm.hpp"
#include <functional>

#include <algorithm>
#include <numeric>

#include "hostla.hpp"
#include "profiling.hpp"

namespace dre {

static const double EPS = 2.220446049250313e-16;

// =============================================================================================
// LDL' objects
// =============================================================================================
LDLtP ldlt_make(Ctx*, int n, const Mat& L, const Mat& D, double alpha, bool diag) {
    auto X = std::make_shared<LDLt>();
    X->n = n;
    X->blocks.push_back({L, D, alpha, diag});
    return X;
}
LDLtP ldlt_zero(int n) {
    auto X = std::make_shared<LDLt>();
    X->n = n;
    return X;
}
LDLtP ldlt_add(const LDLtP& a, const LDLtP& b) {
    DRE_REQUIRE(a->n == b->n, "outer dimensions must match");
    if (a->iszero()) return b;
    if (b->iszero()) return a;
    auto X = std::make_shared<LDLt>();
    X->n = a->n;
    X->blocks = a->blocks;
    X->blocks.insert(X->blocks.end(), b->blocks.begin(), b->blocks.end());
    return X;
}
LDLtP ldlt_scale(const LDLtP& a, double alpha) {
    auto X = std::make_shared<LDLt>();
    X->n = a->n;
    X->blocks = a->blocks;
    for (auto& b : X->blocks) b.alpha *= alpha;
    return X;
}
LDLtP ldlt_deepcopy(Ctx* ctx, const LDLtP& a) {
    auto X = std::make_shared<LDLt>();
    X->n = a->n;
    for (auto& b : a->blocks) {
        LBlock nb;
        nb.L = Mat(ctx, b.L.rows, b.L.cols);
        nb.D = Mat(ctx, b.D.rows, b.D.cols);
        copy_mat(ctx, b.L, nb.L);
        copy_mat(ctx, b.D, nb.D);
        nb.alpha = b.alpha; nb.diag = b.diag;
        X->blocks.push_back(nb);
    }
    return X;
}

static Mat hcat_blocks(Ctx* ctx, const LDLt& X) {
    const int c = X.rank();
    Mat L(ctx, X.n, c);
    int off = 0;
    std::vector<CopyDesc> cd;
    for (auto& b : X.blocks) {
        if (b.L.cols == 0) continue;
        Mat dst = L.colsview(off, b.L.cols);
        cd.push_back({b.L.p, dst.p, X.n, b.L.cols, b.L.ld, dst.ld});
        off += b.L.cols;
    }
    copy_batched(ctx, cd);
    return L;
}

void ldlt_concatenate(Ctx* ctx, LDLt& X) {
    if (X.blocks.size() <= 1) return;
    const int c = X.rank();
    Mat L = hcat_blocks(ctx, X);
    Mat D(ctx, c, c);
    fill_mat(ctx, D, 0.0);
    int off = 0;
    bool diag = true;
    for (auto& b : X.blocks) {
        const int k = b.L.cols;
        if (k == 0) continue;
        Mat dst = D.view(off, off, k, k);
        copy_mat(ctx, b.D, dst, b.alpha);
        diag = diag && b.diag;
        off += k;
    }
    X.blocks.clear();
    X.blocks.push_back({L, D, 1.0, diag});
}

// out(:, blk) = alpha_blk * M(:, blk) * D_blk  for every block of X (M has X.rank() columns): one batched launch
static void mul_blockdiag(Ctx* ctx, const Mat& M, const LDLt& X, Mat& out) {
    std::vector<GemmBatchDesc> descs;
    int off = 0;
    for (auto& b : X.blocks) {
        const int k = b.L.cols;
        if (k == 0) continue;
        Mat src = M.colsview(off, k), dst = out.colsview(off, k);
        descs.push_back({src.p, b.D.p, dst.p, nullptr, b.alpha, M.rows, k, k, src.ld, b.D.ld, dst.ld, 0});
        off += k;
    }
    gemm_batched(ctx, descs, "gemm_compress");
}
// Lcat = [L_1 ... L_p] and LD(:, blk) = alpha_blk L_blk D_blk in one batched launch (the copy rides on the product)
static void hcat_scale_blocks(Ctx* ctx, const LDLt& X, Mat& Lcat, Mat& LD) {
    std::vector<GemmBatchDesc> descs;
    int off = 0;
    for (auto& b : X.blocks) {
        const int k = b.L.cols;
        if (k == 0) continue;
        Mat dl = Lcat.colsview(off, k), dd = LD.colsview(off, k);
        descs.push_back({b.L.p, b.D.p, dd.p, dl.p, b.alpha, X.n, k, k, b.L.ld, b.D.ld, dd.ld, dl.ld});
        off += k;
    }
    gemm_batched(ctx, descs, "gemm_compress");
}


// Compression of a WIDE factor (c >> rank) through a randomized range finder (LDLt.jl:204-225 replaced for this regime; same result up
// to the truncation tolerance).  With Om an n x s Gaussian test matrix, Y = X Om = L (Dt (L' Om)) spans the numerical range of
// X = L Dt L' as soon as s exceeds the numerical rank by a modest oversampling, so  X ~ Q (Q'XQ) Q'  with Q = orth(Y):  three GEMM
// passes over the n x c factor (L'Om, L W, Q'L) instead of four per 16 columns of rank in the factor-form band reduction.  The small
// s x s matrix Q'XQ then goes through the usual band reduction, which fixes the final rank J.  Two independent acceptance tests:
// (1) J leaves at least 32 of the s sketch directions unused (they carry nothing above the truncation tolerance), (2) 16 further
// probe columns G, independent of Q:  ||(I - QQ') X G||_F <= 64 eps ||X G||_F  (the floor of that difference in f64 is ~ 20 eps).
// A rejected sketch costs its three passes and the caller falls back to the factor-form reduction.  Only for sums without
// cancellation (the ADI solution factors): the relative accuracy of Y is eps ||L||^2 ||Dt||.
// Orthonormal basis of the columns of Y (n x s, destroyed) into Q (n x s) without Householder panels: 64-column blocks, each projected
// against the finished blocks and orthonormalised by Cholesky QR (Gram matrix -> k_chol_inv -> GEMM with inv(R)), both TWICE (block
// Gram-Schmidt with re-orthogonalisation, the second round on the already well-conditioned block).  Everything is a GEMM over the n rows plus a 64 x 64 workgroup
// kernel, 14 launches per block, against 16 dependent latency-bound column steps per 16 columns of a Householder/TSQR panel.  Valid
// while every block has cond <= ~3e6 AFTER the projections (columns of X Om: the decay of the spectrum over 64 indices); k_chol_inv raises
// *flag otherwise and the caller redoes the factorisation with Householder panels.
// j_start > 0: the first j_start columns of Q are ALREADY orthonormal (a warm-start basis); only the columns from there on are taken from Y,
// projected against everything before them and orthonormalised
static void orth_cholqr(Ctx* ctx, Mat& Y, Mat& Q, int* flag_dev, int j_start = 0, bool permissive = false) {
    const int n = Y.rows, s = Y.cols, bs = 64;
    Mat G(ctx, bs, bs), Ri(ctx, bs, bs), T(ctx, n, bs);
    DevArr<double> ref(ctx, 1);              // scale of the sketch: largest squared column norm of the first block
    DevArr<int> nullmask(ctx, bs);
    static const bool trace = env_trace("cholqr");
    DevArr<double> dbg(ctx, 64);
    int nblk = 0;
    for (int j0 = j_start; j0 < s; j0 += bs) {
        const int b = std::min(bs, s - j0);
        Mat Yb = Y.colsview(j0, b), Qb = Q.colsview(j0, b), Tb = T.colsview(0, b), Gb = G.view(0, 0, b, b), Rb = Ri.view(0, 0, b, b);
        // project, normalise, project AGAIN, normalise again: the second projection acts on the well-conditioned T, so the loss of
        // orthogonality against the earlier blocks is O(eps) instead of O(eps cond(Y_b))
        auto project = [&](Mat& V) {
            if (j0 == 0) return;
            Mat Qp = Q.colsview(0, j0), W(ctx, j0, b);
            gemm(ctx, true, false, 1.0, Qp, V, 0.0, W, nullptr, "gemm_orth");
            gemm(ctx, false, false, -1.0, Qp, W, 1.0, V, nullptr, "gemm_orth");
        };
        project(Yb);
        gemm(ctx, true, false, 1.0, Yb, Yb, 0.0, Gb, nullptr, "gemm_orth");
        chol_inv(ctx, Gb, Rb, flag_dev, ref.p, permissive ? 3 : (j0 == j_start ? 0 : 1), nullmask.p, trace && nblk < 32 ? dbg.p + 2 * nblk : nullptr);
        gemm(ctx, false, false, 1.0, Yb, Rb, 0.0, Tb, nullptr, "gemm_orth");
        // a column that was rounding noise relative to the whole sketch (sketch wider than the numerical rank) becomes a fresh random direction:
        // Q stays orthonormal in all its columns, as a Householder Q would, and the band reduction of Q'XQ sorts the direction out
        fill_gauss_masked(ctx, Tb, 0x9E3779B97F4A7C15ull + (unsigned long long)j0, nullmask.p);
        project(Tb);
        gemm(ctx, true, false, 1.0, Tb, Tb, 0.0, Gb, nullptr, "gemm_orth");
        chol_inv(ctx, Gb, Rb, flag_dev, ref.p, 2, nullptr, trace && nblk < 32 ? dbg.p + 2 * nblk + 1 : nullptr);
        ++nblk;
        gemm(ctx, false, false, 1.0, Tb, Rb, 0.0, Qb, nullptr, "gemm_orth");
    }
    if (trace) {
        double h[64];
        ctx_fetch(ctx, dbg.p, sizeof(double) * 2 * std::min(nblk, 32), h);
        std::fprintf(stderr, "[cholqr] n=%d s=%d  min pivot / max diagonal per block (pass 1, pass 2):", n, s);
        for (int b2 = 0; b2 < std::min(nblk, 32); ++b2) std::fprintf(stderr, "  %.1e %.1e", h[2 * b2], h[2 * b2 + 1]);
        std::fprintf(stderr, "\n");
    }
}

static bool sketch_compress(Ctx* ctx, LDLt& X, double tolfac, int s, long skey) {
    const int n = X.n, c = X.rank(), sp = s + 16;
    static const bool trace = env_trace("compress");
    Mat Lcat = (X.blocks.size() == 1) ? X.blocks[0].L : hcat_blocks(ctx, X);
    Mat W1(ctx, sp, c), W2(ctx, sp, c), Y(ctx, n, sp);
    const long spkey = skey - 2;                       // band_hint: sketches with the sparse sign test matrix the probe rejected at this order (two strikes: Gaussian)
    const bool use_sparse = ctx->compress_sketch_sparse && s <= 1024 && ctx->band_hint[spkey] < 2;
    if (use_sparse) {
        // Om = [sparse sign matrix (s columns) | Gaussian probe (16 columns, independent of it)]
        Mat Ws = W1.view(0, 0, s, c), Wg = W1.view(s, 0, 16, c), G(ctx, n, 16);
        sketch_sign(ctx, Lcat, Ws, 0x2545F4914F6CDD1Dull);
        fill_gauss(ctx, G, 0x5851F42D4C957F2Dull);
        gemm(ctx, true, false, 1.0, G, Lcat, 0.0, Wg, nullptr, "gemm_sketch");
    } else {
        Mat Om(ctx, n, sp);
        fill_gauss(ctx, Om, 0x2545F4914F6CDD1Dull);
        gemm(ctx, true, false, 1.0, Om, Lcat, 0.0, W1, nullptr, "gemm_sketch");       // Om' L
    }
    mul_blockdiag(ctx, W1, X, W2);                                                     // Om' L Dt
    gemm(ctx, false, true, 1.0, Lcat, W2, 0.0, Y, nullptr, "gemm_sketch");            // X Om  (Dt symmetric)
    Mat Yr = Y.colsview(0, s), Z = Y.colsview(s, 16);
    DevArr<double> nrm(ctx, 2);
    frob2_device(ctx, Z, nrm.p);
    Mat Q(ctx, n, s);
    DevArr<long long> cflag(ctx, 1);
    DRE_HIP(hipMemsetAsync(cflag.p, 0, sizeof(long long), ctx->stream));
    const long ckey = skey - 1;                        // band_hint: Cholesky-QR breakdowns seen at this order (two strikes: Householder panels from then on)
    const bool use_chol = ctx->compress_sketch_cholqr && ctx->band_hint[ckey] < 2;
    if (use_chol) orth_cholqr(ctx, Yr, Q, reinterpret_cast<int*>(cflag.p));
    else {
        QRFact qr = qr_factor(ctx, Yr);
        set_identity(ctx, Q, 1.0);
        qr_apply_q(ctx, qr, Q, false);
    }
    {
        Mat QtZ(ctx, s, 16);
        gemm(ctx, true, false, 1.0, Q, Z, 0.0, QtZ, nullptr, "gemm_sketch");
        gemm(ctx, false, false, -1.0, Q, QtZ, 1.0, Z, nullptr, "gemm_sketch");
        frob2_device(ctx, Z, nrm.p + 1);
    }
    Mat B(ctx, s, c), BD(ctx, s, c), S(ctx, s, s);
    gemm(ctx, true, false, 1.0, Q, Lcat, 0.0, B, nullptr, "gemm_sketch");             // Q' L
    mul_blockdiag(ctx, B, X, BD);
    gemm(ctx, false, true, 1.0, BD, B, 0.0, S, nullptr, "gemm_compress");             // Q' X Q
    symmetrize(ctx, S);
    SymBand sb = sym_band_reduce(ctx, S, tolfac);
    double h[2] = {0.0, 0.0};
    long long cf = 0;
    ctx_fetch(ctx, nrm.p, 2 * sizeof(double), h, cflag.p, sizeof(long long), &cf);
    const double est = h[0] > 0.0 ? std::sqrt(h[1] / h[0]) : 0.0;
    const bool chol_bad = use_chol && (cf & 1) != 0;
    const bool ok = !chol_bad && sb.J + 32 <= s && est <= 64.0 * EPS;
    if (trace) std::fprintf(stderr, "[compress] n=%d c=%d sketch s=%d (%s, %s) -> J=%d  probe residual %.2e  %s\n", n, c, s, use_sparse ? "sparse sign" : "Gaussian",
                            use_chol ? "CholQR2 blocks" : "Householder", sb.J, est, ok ? "accepted" : (chol_bad ? "REJECTED (Cholesky breakdown)" : "REJECTED"));
    if (chol_bad) { ctx->band_hint[ckey] += 1; return false; }
    if (use_sparse && !ok && sb.J + 32 <= s) ctx->band_hint[spkey] += 1;       // enough room in the sketch, yet the probe sees a miss: the test matrix's fault
    if (!ok) { ctx->band_hint[skey] = std::max(ctx->band_hint[skey], std::min(sb.J + 16, s)); return false; }
    ctx->cstats.calls++; ctx->cstats.cols_in += c; ctx->cstats.order += s; ctx->cstats.tri_steps += sb.J; ctx->cstats.rank_out += sb.J;
    ctx->band_hint[skey] = sb.J;
    X.blocks.clear();
    if (sb.J == 0) { X.blocks.push_back({Mat(ctx, n, 0), Mat(ctx, 0, 0), 1.0, true}); return true; }
    Mat Bq = sym_band_basis(ctx, sb);                 // s x J
    Mat Lnew(ctx, n, sb.J);
    gemm(ctx, false, false, 1.0, Q, Bq, 0.0, Lnew, nullptr, "gemm_sketch");
    X.blocks.push_back({Lnew, sb.D, 1.0, false, true});
    return true;
}

// Compression of a wide factored sum X = L blockdiag(alpha_b D_b) L' whose range is KNOWN to lie close to that of an orthonormal basis Q0 (n x q0):
// the warm-start residual of a Rosenbrock step against the previous step's (ros1_recurrence_loop) — the subspace moves slowly between time
// steps.  Range finder with a warm start: Q = [Q0, orth((I - Q0 Q0') X Om)] with only sx fresh directions, S = Q'XQ (a hundred rows), its
// early-terminating band reduction with the absolute tolerance, L <- Q Qb.  Six passes over the factor as GEMMs + one 64-column Cholesky-QR
// block, against ~5 panels x 13 dependent launches of the factor-form reduction (1.5 ms -> ~0.6 ms at n = 5177, c = 2300).  A 16-column Gaussian
// probe measures what the basis missed, ||(I - QQ') X||_F ~ sqrt(n / 16) ||(I - QQ') X Om_p||_F: accepted below abs_tol (the level the caller
// truncates at anyway); otherwise the caller runs the full reduction.  *missed returns the estimate.
// rel_accept > 0: the probe is judged RELATIVE to its own size (what the basis missed of X Om_p over X Om_p <= rel_accept, the criterion of
// sketch_compress: 64 eps is the rounding floor of the projection) instead of against abs_tol — the compression of X itself, whose
// tolerance 4 eps ||X|| lies below that floor.
static bool warm_compress(Ctx* ctx, LDLt& X, const Mat& Q0, double tolfac, double abs_tol, int sx, double* missed, double rel_accept = 0.0) {
    const int n = X.n, c = X.rank(), q0 = Q0.cols, sp = sx + 16, s = q0 + sx;
    static const bool trace = env_trace("compress");
    if (c == 0 || q0 < 16 || s <= 64 || s + 80 > n || abs_tol <= 0.0 || Q0.rows != n) {
        if (trace) std::fprintf(stderr, "[warm compress] not applicable: c=%d q0=%d sx=%d abs_tol=%g\n", c, q0, sx, abs_tol);
        return false;
    }
    Mat Lcat = (X.blocks.size() == 1) ? X.blocks[0].L : hcat_blocks(ctx, X);
    Mat Om(ctx, n, sp), W1(ctx, sp, c), W2(ctx, sp, c), Y(ctx, n, s + 16);
    fill_gauss(ctx, Om, 0x6A09E667F3BCC909ull);
    gemm(ctx, true, false, 1.0, Om, Lcat, 0.0, W1, nullptr, "gemm_sketch");             // Om' L
    mul_blockdiag(ctx, W1, X, W2);                                                       // Om' L Dt
    Mat Yx = Y.colsview(q0, sp);                                                         // X Om lands behind the warm-start columns
    gemm(ctx, false, true, 1.0, Lcat, W2, 0.0, Yx, nullptr, "gemm_sketch");
    Mat Yr = Y.colsview(0, s), Z = Y.colsview(s, 16);
    DevArr<double> nrm(ctx, 2);
    frob2_device(ctx, Z, nrm.p);
    DevArr<long long> cflag(ctx, 1);
    DRE_HIP(hipMemsetAsync(cflag.p, 0, sizeof(long long), ctx->stream));
    Mat Q(ctx, n, s);
    { Mat d = Q.colsview(0, q0); copy_mat(ctx, Q0, d); }
    orth_cholqr(ctx, Yr, Q, reinterpret_cast<int*>(cflag.p), q0, true);
    {
        Mat QtZ(ctx, s, 16);
        gemm(ctx, true, false, 1.0, Q, Z, 0.0, QtZ, nullptr, "gemm_sketch");
        gemm(ctx, false, false, -1.0, Q, QtZ, 1.0, Z, nullptr, "gemm_sketch");
        frob2_device(ctx, Z, nrm.p + 1);
    }
    Mat B(ctx, s, c), BD(ctx, s, c), S(ctx, s, s);
    gemm(ctx, true, false, 1.0, Q, Lcat, 0.0, B, nullptr, "gemm_sketch");               // Q' L
    mul_blockdiag(ctx, B, X, BD);
    gemm(ctx, false, true, 1.0, BD, B, 0.0, S, nullptr, "gemm_compress");               // Q' X Q
    symmetrize(ctx, S);
    SymBand sb = sym_band_reduce(ctx, S, tolfac, abs_tol);
    double h[2] = {0.0, 0.0};
    long long cf = 0;
    ctx_fetch(ctx, nrm.p, 2 * sizeof(double), h, cflag.p, sizeof(long long), &cf);
    const double est = std::sqrt(std::max(h[1], 0.0) * (double)n / 16.0);
    if (missed) *missed = est;
    const double est_rel = h[0] > 0.0 ? std::sqrt(std::max(h[1], 0.0) / h[0]) : 0.0;
    const bool ok = (cf & 1) == 0 && (rel_accept > 0.0 ? est_rel <= rel_accept : est <= abs_tol) && sb.J + 16 <= s;
    if (trace) std::fprintf(stderr, "[warm compress] n=%d c=%d q0=%d sx=%d -> J=%d  missed %.2e (tolerance %.2e)  %s\n", n, c, q0, sx, sb.J, est, abs_tol, ok ? "accepted" : "REJECTED");
    if (!ok) return false;
    X.blocks.clear();
    if (sb.J == 0) { X.blocks.push_back({Mat(ctx, n, 0), Mat(ctx, 0, 0), 1.0, true}); return true; }
    Mat Bq = sym_band_basis(ctx, sb);                 // s x J
    Mat Lnew(ctx, n, sb.J);
    gemm(ctx, false, false, 1.0, Q, Bq, 0.0, Lnew, nullptr, "gemm_sketch");
    X.blocks.push_back({Lnew, sb.D, 1.0, false, true});
    return true;
}

// Rounding noise of forming S = A B' (A, B: n x c) in the probabilistic model: entry (i, j) is off by ~ eps sqrt(sum_k (a_ik b_jk)^2), hence
// ||noise||_F^2 ~ eps^2 sum_k ||A[:,k]||^2 ||B[:,k]||^2 — the pairing of the columns matters (for Ros2's stage-1 right-hand side the large
// blocks A'L never meet each other: a bound by ||A||_F ||B||_F overestimates the noise by orders of magnitude and truncates signal).
__global__ __launch_bounds__(256) void k_colpair_noise(int n, const double* __restrict__ A, int lda, const double* __restrict__ B, int ldb, double* __restrict__ part) {
    __shared__ double red[8];
    const int k = blockIdx.x;
    double sa = 0.0, sb = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) { const double a = A[i + (size_t)k * lda], b = B[i + (size_t)k * ldb]; sa += a * a; sb += b * b; }
    for (int o = 32; o > 0; o >>= 1) { sa += __shfl_xor(sa, o, 64); sb += __shfl_xor(sb, o, 64); }
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = sa; red[4 + (threadIdx.x >> 6)] = sb; }
    __syncthreads();
    if (threadIdx.x == 0) part[k] = ((red[0] + red[1]) + (red[2] + red[3])) * ((red[4] + red[5]) + (red[6] + red[7]));
}
__global__ __launch_bounds__(256) void k_noise_floor(int c, double fac, const double* __restrict__ part, double* __restrict__ floor_out) {
    __shared__ double red[4];
    double s = 0.0;
    for (int i = threadIdx.x; i < c; i += 256) s += part[i];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) floor_out[0] = fac * 2.220446049250313e-16 * sqrt((red[0] + red[1]) + (red[2] + red[3]));
}
static double noise_floor_fac() {
    static const double f = 4.0;    // 0.03 ... 4: same K(t) to 2e-14 (tools/dbg_ros2_full.py)
    return f;
}
void ldlt_compress(Ctx* ctx, LDLt& X, double tolfac, bool exact, double abs_tol, int mode) {
    const int n = X.n, c = X.rank();
    const bool nfloor = !exact && (mode & COMPRESS_NOISE_FLOOR), keep_result = !exact && (mode & COMPRESS_KEEP_RESULT);
    if (nfloor) abs_tol = -1.0;
    DevArr<double> nf;                       // [0] the floor, [1..] per-column products
    auto floor_of = [&](const Mat& A, const Mat& B) {       // device-side floor for S = A B'
        nf = DevArr<double>(ctx, (size_t)A.cols + 1);
        hipLaunchKernelGGL(k_colpair_noise, dim3(A.cols), dim3(256), 0, ctx->stream, A.rows, (const double*)A.p, A.ld, (const double*)B.p, B.ld, nf.p + 1);
        hipLaunchKernelGGL(k_noise_floor, dim3(1), dim3(256), 0, ctx->stream, A.cols, noise_floor_fac(), (const double*)(nf.p + 1), nf.p);
    };
    auto floor_host = [&]() { double h = 0.0; ctx_fetch(ctx, nf.p, sizeof(double), &h); return h; };
    auto set_empty = [&]() {
        X.blocks.clear();
        X.blocks.push_back({Mat(ctx, n, 0), Mat(ctx, 0, 0), 1.0, true});
    };
    if (c == 0) { set_empty(); return; }
    // Q = I is admissible whenever the n x n matrix S = L D L' is affordable; for small n this skips the whole QR
    // (two thirds of all panel factorisations at n = 371) at the price of one GEMM.
    // (a handful of columns: the QR path keeps the rank <= c, the direct form can only stop at panel boundaries of the n x n problem)
    const bool wide = c >= n || (!exact && ((n <= 512 && c > 64) || (n <= ctx->compress_direct_max_n && (double)c * ctx->compress_direct_ratio >= (double)n)));
    if (env_trace("compress")) std::fprintf(stderr, "[compress enter] n=%d c=%d wide=%d exact=%d abs_tol=%g factor_min_n=%d min_cols=%d sketch=%d/%d\n", n, c, (int)wide, (int)exact, abs_tol, ctx->compress_factor_min_n, ctx->compress_factor_min_cols, ctx->compress_sketch, ctx->compress_sketch_min_cols);
    const long skey = -(4000000000L + (long)n);          // band_hint: rank of the previous wide-factor compression at this order
    const bool sketchable = !wide && !exact && !nfloor && abs_tol <= 0.0 && ctx->compress_sketch && n >= ctx->compress_factor_min_n && c >= ctx->compress_sketch_min_cols && c + 64 <= n;
    if (sketchable) {
        auto hit = ctx->band_hint.find(skey);
        if (hit != ctx->band_hint.end() && hit->second > 0) {
            const int s = ((hit->second + ctx->compress_sketch_extra + 15) / 16) * 16;
            if ((double)c >= ctx->compress_sketch_ratio * s && s + 80 <= n && sketch_compress(ctx, X, tolfac, s, skey)) return;
        }
    }
    if (!wide && !exact && n >= ctx->compress_factor_min_n && c >= ctx->compress_factor_min_cols && c + 64 <= n) {
        // large n: the band reduction works on the factor itself (dense.hip, lr_band_reduce): rank/16 panel steps on n x c data
        // instead of a QR of all c columns followed by the reduction of R D R'
        Mat Lw(ctx, n, c + 16);                  // 16 spare columns: the probe vectors of the termination estimate
        std::vector<LrBlockD> tab;
        std::vector<CopyDesc> cd;
        int off = 0;
        for (auto& b : X.blocks) {
            const int k = b.L.cols;
            if (k == 0) continue;
            Mat dst = Lw.colsview(off, k);
            cd.push_back({b.L.p, dst.p, n, k, b.L.ld, dst.ld});
            tab.push_back({off, k, b.D.ld, b.diag ? 1 : 0, b.D.p, b.alpha});
            off += k;
        }
        copy_batched(ctx, cd);
        double lr_tol = abs_tol;
        if (nfloor) {
            // noise of  L blockdiag(alpha_b D_b) L'  (host value: the factor-form reduction keeps its control block on the host side anyway)
            Mat Lc = Lw.colsview(0, c), LDc(ctx, n, c);
            mul_blockdiag(ctx, Lc, X, LDc);
            floor_of(LDc, Lc);
            lr_tol = floor_host();
        }
        SymBand sb = lr_band_reduce(ctx, Lw, tab, tolfac, lr_tol, nfloor);
        ctx->cstats.calls++; ctx->cstats.cols_in += c; ctx->cstats.order += n; ctx->cstats.tri_steps += sb.J; ctx->cstats.rank_out += sb.J;
        if (sketchable) ctx->band_hint[skey] = std::max(sb.J, 16);
        if (sb.J == 0) { set_empty(); return; }
        if (sb.J >= c && !keep_result) { ldlt_concatenate(ctx, X); return; }        // nothing gained: keep the summands
        Mat Bq = sym_band_basis(ctx, sb);
        X.blocks.clear();
        X.blocks.push_back({Bq, sb.D, 1.0, false, true});
        return;
    }
    Mat Lcat, S, V0, VT0, Quser;
    QRFact qr;
    const bool userq = exact && ctx->orthf_fn != nullptr;
    if (userq) {
        // the caller's orthf (LDLt.jl:211: Q, R = orthf(L)), then S = R D R' as in the reference
        Lcat = (X.blocks.size() == 1) ? X.blocks[0].L : hcat_blocks(ctx, X);
        const int pq = std::min(n, c);
        Quser = Mat(ctx, n, pq);
        Mat Ru(ctx, pq, c), RD(ctx, pq, c);
        DRE_HIP(hipStreamSynchronize(ctx->stream));
        const int rc = ctx->orthf_fn(ctx->orthf_user, n, c, Lcat.p, Lcat.ld, Quser.p, Quser.ld, Ru.p, Ru.ld);
        ctx->orthf_calls++;
        if (rc != 0) throw Error(ERR_INTERNAL, "the user-supplied orthf returned " + std::to_string(rc));
        mul_blockdiag(ctx, Ru, X, RD);
        S = Mat(ctx, pq, pq);
        gemm(ctx, false, true, 1.0, RD, Ru, 0.0, S, nullptr, "gemm_compress");
    } else if (wide) {
        // more columns than rows: Q = I, "R" = L (any orthogonal-times-anything factorisation is admissible)
        Mat LD;
        static const int rot_min_n = 65;
        if (!exact && lead_rotation_enabled() && c >= 32 && n >= rot_min_n) {
            // start the reduction from the dominant directions (dense.hip, lead_rotate): Q = Q0 instead of I.  L and L D sit side
            // by side so that one batched launch fills both and one block reflector rotates both
            Mat both(ctx, n, 2 * c);
            Lcat = both.colsview(0, c); LD = both.colsview(c, c);
            hcat_scale_blocks(ctx, X, Lcat, LD);
            lead_rotate(ctx, both, V0, VT0);
        }
        else if (X.blocks.size() == 1) { LD = Mat(ctx, n, c); Lcat = X.blocks[0].L; mul_blockdiag(ctx, Lcat, X, LD); }
        else { LD = Mat(ctx, n, c); Lcat = Mat(ctx, n, c); hcat_scale_blocks(ctx, X, Lcat, LD); }
        S = Mat(ctx, n, n);
        gemm(ctx, false, true, 1.0, LD, Lcat, 0.0, S, nullptr, "gemm_compress");
        if (nfloor) floor_of(LD, Lcat);
    } else {
        Lcat = (X.blocks.size() == 1) ? X.blocks[0].L : hcat_blocks(ctx, X);
        Mat A(ctx, n, c);
        copy_mat(ctx, Lcat, A);
        qr = qr_factor(ctx, A);
        Mat RD(ctx, c, c);
        mul_blockdiag(ctx, qr.R, X, RD);
        S = Mat(ctx, c, c);
        gemm(ctx, false, true, 1.0, RD, qr.R, 0.0, S, nullptr, "gemm_compress");
        if (nfloor) floor_of(RD, qr.R);
    }
    symmetrize(ctx, S);
    Mat B, Dnew;
    int r = 0;
    if (exact) {
        SymEig e = sym_eig(ctx, S, tolfac, true);
        ctx->cstats.calls++; ctx->cstats.cols_in += c; ctx->cstats.order += S.rows; ctx->cstats.tri_steps += e.j;
        if (e.j == 0) { set_empty(); return; }
        double wmax = 0.0;
        for (double w : e.w) wmax = std::max(wmax, std::fabs(w));
        const double thr = 100.0 * wmax * EPS;
        std::vector<int> ids;
        for (int i = 0; i < e.j; ++i)
            if (std::fabs(e.w[i]) >= thr && wmax > 0.0) ids.push_back(i);
        std::sort(ids.begin(), ids.end(), [&](int a, int b) { return e.w[a] < e.w[b]; });
        r = (int)ids.size();
        ctx->cstats.rank_out += r;
        if (r == 0) { set_empty(); return; }
        B = sym_eig_backtransform(ctx, e, ids);
        std::vector<double> hd((size_t)r * r, 0.0);
        for (int i = 0; i < r; ++i) hd[i + (size_t)i * r] = e.w[ids[i]];
        Dnew = Mat(ctx, r, r);
        DRE_HIP(hipMemcpyAsync(Dnew.p, hd.data(), hd.size() * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        DRE_HIP(hipStreamSynchronize(ctx->stream));
    } else {
        if (S.rows <= 64) {
            // small problems: unblocked reduction with per-column termination gives the exact truncation rank
            // (the blocked variant can only stop at multiples of the panel width)
            SymEig e = nfloor ? sym_eig(ctx, S, tolfac, false, floor_host(), true) : sym_eig(ctx, S, tolfac, false, abs_tol);
            ctx->cstats.calls++; ctx->cstats.cols_in += c; ctx->cstats.order += S.rows; ctx->cstats.tri_steps += e.j;
            r = e.j;
            ctx->cstats.rank_out += r;
            if (r == 0) { set_empty(); return; }
            std::vector<int> ids(r);
            for (int i = 0; i < r; ++i) ids[i] = i;
            B = sym_eig_backtransform(ctx, e, ids);
            Dnew = sym_tridiag_dense(ctx, e);
        } else {
            SymBand sb = nfloor ? sym_band_reduce(ctx, S, tolfac, -1.0, nf.p, nullptr, nullptr, 0, true) : sym_band_reduce(ctx, S, tolfac, abs_tol);
            ctx->cstats.calls++; ctx->cstats.cols_in += c; ctx->cstats.order += S.rows; ctx->cstats.tri_steps += sb.J;
            r = sb.J;
            ctx->cstats.rank_out += r;
            if (r == 0) { set_empty(); return; }
            B = sym_band_basis(ctx, sb);
            Dnew = sb.D;
        }
    }
    Mat Lnew;
    if (userq) {
        Lnew = Mat(ctx, n, r);
        gemm(ctx, false, false, 1.0, Quser, B, 0.0, Lnew, nullptr, "gemm_compress");          // L <- Q V_keep  (LDLt.jl:220-221)
    } else if (wide) {
        Lnew = B;
        lead_rotate_back(ctx, V0, VT0, Lnew);
    } else {
        Lnew = Mat(ctx, n, r);
        fill_mat(ctx, Lnew, 0.0);
        Mat top = Lnew.view(0, 0, c, r);
        copy_mat(ctx, B, top);
        qr_apply_q(ctx, qr, Lnew, false);
    }
    if (!exact && r >= c && !keep_result) {
        // nothing gained (numerical rank = number of columns, or a remainder that stays above the tolerance because S itself
        // carries cancellation): keep the summands as they are, concatenated
        ldlt_concatenate(ctx, X);
        return;
    }
    static const bool trace = env_trace("compress");
    if (trace) std::fprintf(stderr, "[compress] n=%d c=%d -> r=%d  (%s, order %d)\n", n, c, r, wide ? "direct" : "qr", S.rows);
    X.blocks.clear();
    X.blocks.push_back({Lnew, Dnew, 1.0, exact, true});
}

void ldlt_destructure(Ctx* ctx, LDLt& X, double tolfac, bool exact) {
    if (X.blocks.size() > 1) ldlt_compress(ctx, X, tolfac, exact);
    if (X.blocks.empty()) X.blocks.push_back({Mat(ctx, X.n, 0), Mat(ctx, 0, 0), 1.0, true});
}

double ldlt_norm(Ctx* ctx, LDLt& X) {
    if (X.rank() == 0) return 0.0;
    ldlt_concatenate(ctx, X);
    auto& b = X.blocks[0];
    return ldlt_norm_host(ctx, b.L, b.D, b.alpha);
}

// norm(::LDLt) as the reference defines it (LDLt.jl:77-89: through an orthogonal-triangular factorisation of L): accurate
// relative to the RESULT even when the terms of X cancel.  The Gram form used inside the ADI loop is only accurate relative to
// the largest term (error ~ sqrt(eps) ||L||^2 ||D|| under cancellation), which is harmless there but not for user-level sums
// such as the Arnoldi vectors of the low-rank GMRES.  X itself is left untouched (the compression works on a shallow copy).
double ldlt_norm_accurate(Ctx* ctx, const LDLt& X) {
    if (X.rank() == 0) return 0.0;
    if (ctx->orthf_fn) {
        // LDLt.jl:77-89 with the caller's orthf: |alpha| ||R D R'||_F  (= the norm of the literally compressed object)
        LDLt Y = X;
        ldlt_compress(ctx, Y, 4.0, true);
        if (Y.rank() == 0) return 0.0;
        auto& b = Y.blocks[0];
        return ldlt_norm_host(ctx, b.L, b.D, b.alpha);
    }
    if (X.blocks.size() == 1 && X.blocks[0].ortho) {        // already compressed: L'L = I, nothing can cancel
        auto& b0 = X.blocks[0];
        return ldlt_norm_host(ctx, b0.L, b0.D, b0.alpha);
    }
    LDLt Y = X;
    ldlt_compress(ctx, Y, 4.0, false);
    if (Y.rank() == 0) return 0.0;
    if (!(Y.blocks.size() == 1 && Y.blocks[0].ortho)) {     // the compression kept the summands (nothing to gain): orthogonalise exactly
        Y = X;
        ldlt_compress(ctx, Y, 4.0, true);
        if (Y.rank() == 0) return 0.0;
    }
    auto& b = Y.blocks[0];
    return ldlt_norm_host(ctx, b.L, b.D, b.alpha);
}

// =============================================================================================
// Sherman-Morrison-Woodbury pieces (/root/reference/src/blocklinear/sherman-morrison-woodbury.jl:10-45)
// with  F' + mu E' = M + inv(alpha) Vt U'  :  W = M^-1 [R, Vt];  S = alpha I + U' W_U;  X = W_R - W_U S^-1 (U' W_R)
// =============================================================================================
template <typename T> __device__ inline T wave_sum_t(T v);
template <> __device__ inline double wave_sum_t<double>(double v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
template <> __device__ inline cplx wave_sum_t<cplx>(cplx v) {
    for (int o = 32; o > 0; o >>= 1) { v.re += __shfl_xor(v.re, o, 64); v.im += __shfl_xor(v.im, o, 64); }
    return v;
}

// small(0:m, c) = U' * W(:, c); one workgroup per column c
template <typename T>
__global__ __launch_bounds__(256) void k_smw_small(int n, int m, const double* __restrict__ U, int ldu, const T* __restrict__ W,
                                                   int ldw, T* __restrict__ small, int lds_, const AdiState* st) {
    if (st && st->done) return;
    __shared__ double redbuf[4 * 2 * 8];
    T* red = reinterpret_cast<T*>(redbuf);
    const int c = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const T* w = W + (size_t)c * ldw;
    for (int j0 = 0; j0 < m; j0 += 8) {
        T acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = make_scalar<T>(0.0, 0.0);
        for (int i = tid; i < n; i += 256) {
            const T wi = w[i];
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (j0 + j < m) acc[j] += wi * U[i + (size_t)(j0 + j) * ldu];
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            T s = wave_sum_t<T>(acc[j]);
            if (lane == 0) red[wave * 8 + j] = s;
        }
        __syncthreads();
        if (tid < 8 && j0 + tid < m) small[(j0 + tid) + (size_t)c * lds_] = red[tid] + red[8 + tid] + red[16 + tid] + red[24 + tid];
    }
}

// Sinv = inv(alpha I + Smat) by Gauss-Jordan with partial pivoting; one workgroup, m <= 32
template <typename T>
__device__ __forceinline__ void sinv_body(int m, const T* __restrict__ Smat, int lds_, double alpha, T* __restrict__ Sinv, int* err) {
    __shared__ double abuf[32 * 64 * 2];
    __shared__ int piv;
    __shared__ double colkbuf[32 * 2];
    T* colk = reinterpret_cast<T*>(colkbuf);
    T* A = reinterpret_cast<T*>(abuf);   // m x 2m, column-major, ld = 32
    const int tid = threadIdx.x;
    for (int id = tid; id < m * 2 * m; id += 64) {
        const int i = id % m, j = id / m;
        T v = make_scalar<T>(0.0, 0.0);
        if (j < m) { v = Smat[i + (size_t)j * lds_]; if (i == j) v = v + make_scalar<T>(alpha, 0.0); }
        else if (j - m == i) v = make_scalar<T>(1.0, 0.0);
        A[i + j * 32] = v;
    }
    __syncthreads();
    for (int k = 0; k < m; ++k) {
        if (tid == 0) {
            int p = k; double best = abs1(A[k + k * 32]);
            for (int i = k + 1; i < m; ++i) { double a = abs1(A[i + k * 32]); if (a > best) { best = a; p = i; } }
            piv = p;
            if (best == 0.0) *err = 1;
        }
        __syncthreads();
        const int p = piv;
        if (tid < 2 * m && p != k) { T t = A[k + tid * 32]; A[k + tid * 32] = A[p + tid * 32]; A[p + tid * 32] = t; }
        __syncthreads();
        const T rp = recip(A[k + k * 32]);
        __syncthreads();
        if (tid < 2 * m) A[k + tid * 32] *= rp;
        if (tid < m) colk[tid] = A[tid + k * 32];   // multipliers, read before column k is touched
        __syncthreads();
        if (tid < 2 * m) {
            const T akj = A[k + tid * 32];
            for (int i = 0; i < m; ++i)
                if (i != k) A[i + tid * 32] -= colk[i] * akj;
        }
        __syncthreads();
    }
    for (int id = tid; id < m * m; id += 64) {
        const int i = id % m, j = id / m;
        Sinv[i + (size_t)j * m] = A[i + (m + j) * 32];
    }
}
template <typename T>
__global__ __launch_bounds__(64) void k_sinv(int m, const T* __restrict__ Smat, int lds_, double alpha, T* __restrict__ Sinv,
                                             const AdiState* st, int* err) {
    if (st && st->done) return;
    sinv_body<T>(m, Smat, lds_, alpha, Sinv, err);
}
// All shifts of a Cyclic list at once (dense-inverse path): blockIdx.x = shift.  WK_j = [inv; E'inv; U'inv]_j * Vt is (2n+m) x m.
struct SmwBatch { const double* WK; double* Sinv; double* WKS; };
__global__ __launch_bounds__(64) void k_sinv_batched(int n, int m, int ldwk, double alpha, const SmwBatch* __restrict__ bt, const AdiState* st, int* err) {
    if (st && st->done) return;
    const SmwBatch b = bt[blockIdx.x];
    sinv_body<double>(m, b.WK + 2 * (size_t)n, ldwk, alpha, b.Sinv, err);
}
__global__ __launch_bounds__(256) void k_fold_sinv_batched(int nrows, int m, int ldwk, const SmwBatch* __restrict__ bt, const AdiState* st) {
    if (st && st->done) return;
    const SmwBatch b = bt[blockIdx.y];
    const size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= (size_t)nrows * m) return;
    const int r = id % nrows, j = id / nrows;
    double a0 = 0.0, a1 = 0.0;
    int l = 0;
    for (; l + 1 < m; l += 2) {
        a0 += b.WK[r + (size_t)l * ldwk] * b.Sinv[l + (size_t)j * m];
        a1 += b.WK[r + (size_t)(l + 1) * ldwk] * b.Sinv[l + 1 + (size_t)j * m];
    }
    if (l < m) a0 += b.WK[r + (size_t)l * ldwk] * b.Sinv[l + (size_t)j * m];
    b.WKS[r + (size_t)j * nrows] = a0 + a1;
}

// the same with the batch table passed by value (no upload): up to 16 shifts per launch
struct SmwBatchArgs { SmwBatch b[16]; };
__global__ __launch_bounds__(64) void k_sinv_batched_args(int n, int m, int ldwk, double alpha, SmwBatchArgs bt, int* err) {
    const SmwBatch b = bt.b[blockIdx.x];
    sinv_body<double>(m, b.WK + 2 * (size_t)n, ldwk, alpha, b.Sinv, err);
}
__global__ __launch_bounds__(256) void k_fold_sinv_batched_args(int nrows, int m, int ldwk, SmwBatchArgs bt) {
    const SmwBatch b = bt.b[blockIdx.y];
    const size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= (size_t)nrows * m) return;
    const int r = id % nrows, j = id / nrows;
    double a0 = 0.0, a1 = 0.0;
    int l = 0;
    for (; l + 1 < m; l += 2) {
        a0 += b.WK[r + (size_t)l * ldwk] * b.Sinv[l + (size_t)j * m];
        a1 += b.WK[r + (size_t)(l + 1) * ldwk] * b.Sinv[l + 1 + (size_t)j * m];
    }
    if (l < m) a0 += b.WK[r + (size_t)l * ldwk] * b.Sinv[l + (size_t)j * m];
    b.WKS[r + (size_t)j * nrows] = a0 + a1;
}

// WKS = WK(0:nrows, :) * Sinv  — folds the capacitance inverse into the low-rank solve products once per shift, so that the
// per-step apply kernel needs no inner m x m solve.  One thread per output entry.
__global__ __launch_bounds__(256) void k_fold_sinv(int nrows, int m, const double* __restrict__ WK, int ldwk, const double* __restrict__ Sinv,
                                                   double* __restrict__ WKS, int ldo, const AdiState* st) {
    if (st && st->done) return;
    const size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= (size_t)nrows * m) return;
    const int r = id % nrows, j = id / nrows;
    double a0 = 0.0, a1 = 0.0;
    int l = 0;
    for (; l + 1 < m; l += 2) {
        a0 += WK[r + (size_t)l * ldwk] * Sinv[l + (size_t)j * m];
        a1 += WK[r + (size_t)(l + 1) * ldwk] * Sinv[l + 1 + (size_t)j * m];
    }
    if (l < m) a0 += WK[r + (size_t)l * ldwk] * Sinv[l + (size_t)j * m];
    WKS[r + (size_t)j * ldo] = a0 + a1;
}

#define SMW_CB 8
// real:    V  = W_R - W_U * (Sinv * small_R)
// complex: V1 = sqrt2 (Re V + delta Im V),  V2 = sqrt(2 delta^2 + 2) Im V   (/root/reference/src/lyapunov/adi.jl:205-211)
template <typename T, bool HAS_LR>
__global__ __launch_bounds__(256) void k_smw_apply(int n, int m, int k, const T* __restrict__ W, int ldw, const T* __restrict__ WU,
                                                   int ldwu, const T* __restrict__ Sinv, const T* __restrict__ small, int lds_,
                                                   double* __restrict__ V1, int ldv1, double* __restrict__ V2, int ldv2,
                                                   double delta, const AdiState* st) {
    if (st && st->done) return;
    __shared__ double ybuf[SMW_CB * 32 * 2];
    T* y = reinterpret_cast<T*>(ybuf);
    const int c0 = blockIdx.y * SMW_CB, kc = min(SMW_CB, k - c0);
    const int tid = threadIdx.x;
    if (HAS_LR) {
        for (int id = tid; id < kc * m; id += 256) {
            const int j = id % m, c = id / m;
            T acc = make_scalar<T>(0.0, 0.0);
            for (int l = 0; l < m; ++l) acc += Sinv[j + (size_t)l * m] * small[l + (size_t)(c0 + c) * lds_];
            y[j + c * 32] = acc;
        }
        __syncthreads();
    }
    const int i = blockIdx.x * 256 + tid;
    if (i >= n) return;
    for (int c = 0; c < kc; ++c) {
        T v = W[i + (size_t)(c0 + c) * ldw];
        if (HAS_LR)
            for (int j = 0; j < m; ++j) v -= WU[i + (size_t)j * ldwu] * y[j + c * 32];
        if constexpr (sizeof(T) == 8) {
            V1[i + (size_t)(c0 + c) * ldv1] = *reinterpret_cast<double*>(&v);
        } else {
            const cplx z = *reinterpret_cast<cplx*>(&v);
            V1[i + (size_t)(c0 + c) * ldv1] = 1.4142135623730951 * z.re + (1.4142135623730951 * delta) * z.im;
            V2[i + (size_t)(c0 + c) * ldv2] = sqrt(2.0 * delta * delta + 2.0) * z.im;
        }
    }
}

// Dense-inverse ADI step (real shift).  Wst = [inv; E' inv; U' inv] * R holds W (n rows), EW = E' W (n rows) and
// small = U' W (m rows);  WKst the same products for the low-rank factor Vt.  With y = Sinv * small:
//   V = W - WK y,      R <- R - 2 mu E' V = R - 2 mu (EW - EWK y)          (adi.jl:166-171 with LowRankUpdate.jl:29-39)
template <bool HAS_LR>
__global__ __launch_bounds__(256) void k_dense_apply(int n, int m, int k, const double* __restrict__ Wst, int ldw,
                                                     const double* __restrict__ WKS, int ldwk,
                                                     double* __restrict__ V, int ldv, double* __restrict__ R, int ldr,
                                                     double two_mu, const AdiState* st) {
    if (st && st->done) return;
    // WKS = [WK; EWK] * Sinv (folded once per shift), small = rows 2n.. of Wst:  V = W - WKS_top small,  R -= 2 mu (EW - WKS_mid small)
    const int i = blockIdx.x * 64 + (threadIdx.x & 63), c = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (i >= n || c >= k) return;
    const double* wc = Wst + (size_t)c * ldw;
    double v = wc[i], ev = wc[n + i];
    if (HAS_LR) {
        const double* small = wc + 2 * (size_t)n;
        for (int j = 0; j < m; ++j) {
            const double sj = small[j];
            v -= WKS[i + (size_t)j * ldwk] * sj;
            ev -= WKS[n + i + (size_t)j * ldwk] * sj;
        }
    }
    V[i + (size_t)c * ldv] = v;
    R[i + (size_t)c * ldr] -= two_mu * ev;
}

// X(:, c) = W(:, c) - WU (Sinv small(:, c))  for a complex system, split into real and imaginary parts; one workgroup column per c
__global__ __launch_bounds__(256) void k_smw_plain_cplx(int n, int m, const cplx* __restrict__ W, int ldw, const cplx* __restrict__ WU, int ldwu,
                                                        const cplx* __restrict__ Sinv, const cplx* __restrict__ small, int lds_,
                                                        double* __restrict__ Xre, double* __restrict__ Xim, int ldx) {
    __shared__ double ybuf[64];
    cplx* y = reinterpret_cast<cplx*>(ybuf);
    const int c = blockIdx.y, tid = threadIdx.x;
    if (tid < m) {
        cplx acc = {0.0, 0.0};
        for (int l = 0; l < m; ++l) acc += Sinv[tid + (size_t)l * m] * small[l + (size_t)c * lds_];
        y[tid] = acc;
    }
    __syncthreads();
    const int i = blockIdx.x * 256 + tid;
    if (i >= n) return;
    cplx v = W[i + (size_t)c * ldw];
    for (int j = 0; j < m; ++j) v -= WU[i + (size_t)j * ldwu] * y[j];
    Xre[i + (size_t)c * ldx] = v.re; Xim[i + (size_t)c * ldx] = v.im;
}
__global__ void k_real_to_cplx(int rows, int cols, const double* __restrict__ src, int lds_, cplx* __restrict__ dst, int ldd,
                               const AdiState* st) {
    if (st && st->done) return;
    size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= (size_t)rows * cols) return;
    int r = id % rows, c = id / rows;
    dst[r + (size_t)c * ldd] = {src[r + (size_t)c * lds_], 0.0};
}

// =============================================================================================
// Shift oracles
// =============================================================================================
struct ShiftOracle {
    virtual ~ShiftOracle() {}
    virtual std::complex<double> take(int* warn) = 0;
    virtual void update(const Mat& R, const std::vector<Mat>& Vs) {}
    // the shifts that take() will return next, as far as they are already known (never triggers a computation)
    virtual std::vector<std::complex<double>> peek(size_t) const { return {}; }
};
struct CyclicOracle : ShiftOracle {   // shifts/helpers.jl:19-21,91-93
    std::vector<std::complex<double>> v;
    size_t i = 0;
    std::complex<double> take(int*) override { auto x = v[i % v.size()]; ++i; return x; }
    std::vector<std::complex<double>> peek(size_t count) const override {
        std::vector<std::complex<double>> out;
        for (size_t j = 0; j < count && !v.empty(); ++j) out.push_back(v[(i + j) % v.size()]);
        return out;
    }
};

static void apply_Ft(Ctx* ctx, const GaleOperator& op, const Mat& L, Mat& out) {
    // out = F' L = Fs' L + inv(alpha) Vt (U' L)      (LowRankUpdate.jl:51-54,82-85)
    const Pencil& P = *op.P;
    spmm(ctx, P, op.valFt.p, L, out, 1.0, 0.0);
    if (op.has_lr && L.cols > 0) {
        Mat tmp(ctx, op.U.cols, L.cols);
        gemm(ctx, true, false, 1.0, op.U, L, 0.0, tmp);
        gemm(ctx, false, false, 1.0 / op.alpha, op.Vt, tmp, 1.0, out);
    }
}

struct ProjectionOracle : ShiftOracle {   // shifts/projection.jl:34-73
    Ctx* ctx; const GaleOperator* op; int n_history;
    const AdiState* st_dev = nullptr;   // speculative enqueueing: never compute Ritz values from skipped steps
    std::vector<Mat> Vs;   // handles, NOT snapshots
    std::vector<std::complex<double>> buffer;
    size_t pos = 0;
    void update(const Mat& R, const std::vector<Mat>& newVs) override {
        if (newVs.empty()) Vs.push_back(R);
        for (auto& v : newVs) Vs.push_back(v);
        const int lst = (int)Vs.size();
        const int fst = std::max(0, lst - n_history);
        Vs.erase(Vs.begin(), Vs.begin() + fst);
    }
    void take_many(int* warn) {
        const Pencil& P = *op->P;
        if (st_dev) {
            int done = 0;
            DRE_HIP(hipMemcpyAsync(&done, &st_dev->done, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
            DRE_HIP(hipStreamSynchronize(ctx->stream));
            if (done) { buffer.assign(1, std::complex<double>(-1.0, 0.0)); pos = 0; return; }
        }
        int w = 0;
        for (auto& v : Vs) w += v.cols;
        DRE_REQUIRE(w > 0, "Projection shifts: empty history");
        Mat N(ctx, P.n, w);
        int off = 0;
        for (auto& v : Vs) { Mat d = N.colsview(off, v.cols); copy_mat(ctx, v, d); off += v.cols; }
        // orth(N): SVD with absolute cut n*eps (Stuff.jl:13-19) realised as QR + SVD of the small R
        QRFact qr = qr_factor(ctx, N);
        const int kq = qr.kq;
        std::vector<double> hR((size_t)kq * w);
        DRE_HIP(hipMemcpy2DAsync(hR.data(), kq * sizeof(double), qr.R.p, qr.R.ld * sizeof(double), kq * sizeof(double), w, hipMemcpyDeviceToHost, ctx->stream));
        DRE_HIP(hipStreamSynchronize(ctx->stream));
        std::vector<double> Us, sv;
        static const bool trace = env_trace("proj");
        auto t0 = std::chrono::steady_clock::now();
        // orth(N) keeps the left singular vectors with sigma > n eps (Stuff.jl:13-19).  With N = Q R and R square, every sigma(R) > n eps means
        // span(N) = span(Q): the Ritz values of the projected pencil do not depend on WHICH orthonormal basis of that subspace is used, so the
        // SVD of R (one-sided Jacobi on the host: 59 % of the wall-clock of a default-ADI run at n = 371, measured with DRE_TRACE_PROJ) is
        // only run when R may be rank deficient: sigma_min is estimated by four steps of inverse iteration on R'R (triangular solves) and
        // compared with 100 n eps.
        bool full_rank = false;
        static const bool skip_svd = true;
        if (skip_svd && kq == w && w >= 1) {
            double dmin = 1e300;
            for (int i = 0; i < w; ++i) dmin = std::min(dmin, std::fabs(hR[i + (size_t)i * kq]));
            if (dmin > 100.0 * P.n * EPS) {
                std::vector<double> x((size_t)w, 1.0 / std::sqrt((double)w)), y((size_t)w);
                double zn = 0.0;
                for (int it = 0; it < 4; ++it) {
                    for (int i = 0; i < w; ++i) {                  // R' y = x  (forward substitution, R upper triangular column-major)
                        double acc = x[(size_t)i];
                        for (int j = 0; j < i; ++j) acc -= hR[j + (size_t)i * kq] * y[(size_t)j];
                        y[(size_t)i] = acc / hR[i + (size_t)i * kq];
                    }
                    for (int i = w - 1; i >= 0; --i) {             // R z = y  (back substitution; z overwrites x)
                        double acc = y[(size_t)i];
                        for (int j = i + 1; j < w; ++j) acc -= hR[i + (size_t)j * kq] * x[(size_t)j];
                        x[(size_t)i] = acc / hR[i + (size_t)i * kq];
                    }
                    zn = 0.0;
                    for (int i = 0; i < w; ++i) zn += x[(size_t)i] * x[(size_t)i];
                    zn = std::sqrt(zn);
                    if (!(zn > 0.0) || !std::isfinite(zn)) break;
                    for (int i = 0; i < w; ++i) x[(size_t)i] /= zn;
                }
                const double smin_est = (zn > 0.0 && std::isfinite(zn)) ? 1.0 / std::sqrt(zn) : 0.0;      // ||(R'R)^-1 x|| -> 1 / sigma_min^2
                full_rank = smin_est > 100.0 * P.n * EPS;
            }
        }
        if (full_rank) {
            Us.assign((size_t)kq * kq, 0.0);
            for (int i = 0; i < kq; ++i) Us[i + (size_t)i * kq] = 1.0;
            sv.assign((size_t)kq, 1.0);
        }
        // A TINY rank-deficient R (the last two increments of a converging solve: ||R|| ~ 1e-12, a handful of singular values above n eps) took
        // 55 ms of host Jacobi per batch.  Its left singular vectors are the eigenvectors of R R' and the eigenvalues sigma^2 are resolved to
        // eps sigma_max^2, i.e. sigma to ~1.5e-8 sigma_max: exact enough for the absolute cut n eps whenever 3e-8 ||R||_F < n eps — then the
        // device eigensolver (early-terminating tridiagonalisation + QL, dense.hip) does it in ~2 ms and the basis never leaves the device.
        Mat Qdev;
        int r = 0;
        double fro = 0.0;
        for (double v : hR) fro += v * v;
        fro = std::sqrt(fro);
        if (!full_rank && skip_svd && fro * 3e-8 < P.n * EPS && fro > 0.0) {
            Mat G(ctx, kq, kq);
            gemm(ctx, false, true, 1.0, qr.R, qr.R, 0.0, G, nullptr, "gemm_proj");
            symmetrize(ctx, G);
            SymEig e = sym_eig(ctx, G, 4.0, true);
            std::vector<int> ids;
            const double thr2 = (P.n * EPS) * (P.n * EPS);
            for (int i = 0; i < e.j; ++i) if (e.w[(size_t)i] > thr2) ids.push_back(i);
            r = (int)ids.size();
            DRE_REQUIRE(r > 0, "Projection shifts: residual factor is numerically zero");
            Mat B = sym_eig_backtransform(ctx, e, ids);          // kq x r, orthonormal columns
            Qdev = Mat(ctx, P.n, r);
            fill_mat(ctx, Qdev, 0.0);
            Mat top = Qdev.view(0, 0, kq, r);
            copy_mat(ctx, B, top);
        } else if (!full_rank && skip_svd) host_rrqr_svd_left(kq, w, hR, P.n * EPS, Us, sv);      // steeply graded history block: pivoted QR + small SVD
        else if (!full_rank) host_svd_left(kq, w, hR, Us, sv);      // R = Us diag(sv) W'
        auto t1 = std::chrono::steady_clock::now();
        if (Qdev.empty()) {
            std::vector<int> keep;
            for (int i = 0; i < (int)sv.size(); ++i) if (std::fabs(sv[i]) > P.n * EPS) keep.push_back(i);
            r = (int)keep.size();
            DRE_REQUIRE(r > 0, "Projection shifts: residual factor is numerically zero");
            std::vector<double> hB((size_t)P.n * r, 0.0);
            for (int c = 0; c < r; ++c) for (int i = 0; i < kq; ++i) hB[i + (size_t)c * P.n] = Us[i + (size_t)keep[c] * kq];
            Qdev = Mat(ctx, P.n, r);
            DRE_HIP(hipMemcpyAsync(Qdev.p, hB.data(), hB.size() * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
            DRE_HIP(hipStreamSynchronize(ctx->stream));
        }
        Mat Q = Qdev;
        qr_apply_q(ctx, qr, Q, false);
        // restrictions: Q'EQ = (Q'E'Q)',  Q'FQ = (Q'F'Q)'
        Mat EQ(ctx, P.n, r), FQ(ctx, P.n, r), Et(ctx, r, r), Ft(ctx, r, r);
        spmm(ctx, P, P.valEt.p, Q, EQ, 1.0, 0.0);
        apply_Ft(ctx, *op, Q, FQ);
        gemm(ctx, true, false, 1.0, Q, EQ, 0.0, Et);
        gemm(ctx, true, false, 1.0, Q, FQ, 0.0, Ft);
        std::vector<double> hE((size_t)r * r), hF((size_t)r * r), hEt((size_t)r * r), hFt((size_t)r * r);
        DRE_HIP(hipMemcpyAsync(hEt.data(), Et.p, hEt.size() * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        DRE_HIP(hipMemcpyAsync(hFt.data(), Ft.p, hFt.size() * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        DRE_HIP(hipStreamSynchronize(ctx->stream));
        for (int i = 0; i < r; ++i) for (int j = 0; j < r; ++j) { hE[i + (size_t)j * r] = hEt[j + (size_t)i * r]; hF[i + (size_t)j * r] = hFt[j + (size_t)i * r]; }
        auto t2 = std::chrono::steady_clock::now();
        std::vector<std::complex<double>> lam = host_gen_eigvals(r, hF, hE);
        if (trace) {
            auto t3 = std::chrono::steady_clock::now();
            auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
            double svmax = 0.0; for (double v : sv) svmax = std::max(svmax, std::fabs(v));
            std::fprintf(stderr, "[projection] w=%d kq=%d r=%d  svd %.1f ms  device %.1f ms  eig %.1f ms  ||R||_F %.2e sv_max %.2e full_rank %d\n", w, kq, r, ms(t0, t1), ms(t1, t2), ms(t2, t3),
                         fro, svmax, (int)full_rank);
        }
        // stabilize_ritz_values! + safe_sort!  (helpers.jl:122-140)
        int nun = 0;
        for (auto& l : lam) if (!(l.real() < 0)) ++nun;
        if (nun > 0 && nun < (int)lam.size()) {
            if (warn) *warn |= 4;
            std::vector<std::complex<double>> f;
            for (auto& l : lam) if (l.real() < 0) f.push_back(l);
            lam.swap(f);
        } else if (nun == (int)lam.size()) {
            if (warn) *warn |= 8;
            for (auto& l : lam) l = std::complex<double>(-l.real(), l.imag());
        }
        std::stable_sort(lam.begin(), lam.end(), [](const std::complex<double>& a, const std::complex<double>& b) {
            if (a.real() != b.real()) return a.real() < b.real();
            return std::fabs(a.imag()) < std::fabs(b.imag());
        });
        buffer = lam; pos = 0;
    }
    std::complex<double> take(int* warn) override {
        if (pos >= buffer.size()) take_many(warn);
        return buffer[pos++];
    }
    std::vector<std::complex<double>> peek(size_t count) const override {
        std::vector<std::complex<double>> out;
        for (size_t i = pos; i < buffer.size() && out.size() < count; ++i) out.push_back(buffer[i]);
        return out;
    }
};

// =============================================================================================
// GALE residual (/root/reference/src/lyapunov/residual.jl:3-31)
// =============================================================================================
// largest n for which the Ros1 driver carries X as a block list (right-hand side, feedback and residual on the summands)
static int xblocks_max_n() {
    static const int v = 1536;
    return v;
}
// P = F / s + s E,  M = F / s - s E  with  s^4 = ||F||_F^2 / ||E||_F^2  read from device memory (nrm2[0], nrm2[1])
__global__ void k_balance_pm(size_t tot, const double* __restrict__ F, const double* __restrict__ E, const double* __restrict__ nrm2,
                             double* __restrict__ Pm, double* __restrict__ Mm) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= tot) return;
    const double a2 = nrm2[0], b2 = nrm2[1];
    const double s = (a2 > 0.0 && b2 > 0.0) ? sqrt(sqrt(a2 / b2)) : 1.0;
    const double f = F[i] / s, e = E[i] * s;
    Pm[i] = f + e; Mm[i] = f - e;
}
// ||X||_F of a block list through the n x n matrix (small n only; no compression, X untouched); synchronises
static double ldlt_norm_dense_small(Ctx* ctx, const LDLt& X) {
    const int n = X.n, c = X.rank();
    if (c == 0) return 0.0;
    Mat Lcat(ctx, n, c), LD(ctx, n, c), S(ctx, n, n);
    hcat_scale_blocks(ctx, X, Lcat, LD);
    gemm(ctx, false, true, 1.0, LD, Lcat, 0.0, S, nullptr, "gemm_compress");
    return frob_norm_host(ctx, S);
}
// Block lists on both sides (small n, Krylov mode): the summands of C and of the warm start X are used as they are, nothing is
// compressed on the way in.  F'XE + E'XF = (P D P' - M D M') / 2 with P = F'L / s + s E'L, M = F'L / s - s E'L (s balances the
// two terms, so the rounding error stays at eps ||F'L|| ||E'L|| ||D|| like in the [E'L, F'L] form), i.e. per block of X two
// blocks that share its D.
__global__ void k_axpy_inplace(size_t tot, double a, const double* __restrict__ x, double* __restrict__ y) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < tot) y[i] += a * x[i];
}
static LDLtP gale_residual_blocks(Ctx* ctx, const GaleOperator& op, const LDLt& C, const LDLt& X, double tolfac, double abs_tol,
                                  const Mat* warm_L = nullptr, const Mat* warm_EtL = nullptr, int lead_blocks = -1, double e_coeff = 0.0) {
    const Pencil& P = *op.P;
    const int n = P.n, c = X.rank();
    const bool have = warm_L && warm_EtL && warm_L->cols == c && warm_EtL->cols == c && warm_L->rows == n && c > 0;
    Mat Lall = have ? *warm_L : ((X.blocks.size() == 1) ? X.blocks[0].L : hcat_blocks(ctx, X));
    Mat EtL = have ? *warm_EtL : Mat(ctx, n, c);
    Mat FtL(ctx, n, c), Pm(ctx, n, c), Mm(ctx, n, c);
    if (!have) spmm(ctx, P, P.valEt.p, Lall, EtL, 1.0, 0.0);
    apply_Ft(ctx, op, Lall, FtL);
    const bool fold = lead_blocks >= 0 && lead_blocks <= (int)C.blocks.size() && e_coeff != 0.0;
    if (fold) {      // C = lead blocks + e_coeff E'XE:  the last term joins F  (one third fewer columns in the compression below)
        const size_t tot0 = (size_t)n * c;
        hipLaunchKernelGGL(k_axpy_inplace, dim3((unsigned)((tot0 + 255) / 256)), dim3(256), 0, ctx->stream, tot0, 0.5 * e_coeff, (const double*)EtL.p, FtL.p);
    }
    DevArr<double> nrm2(ctx, 2);
    frob2_device(ctx, FtL, nrm2.p);
    frob2_device(ctx, EtL, nrm2.p + 1);
    const size_t tot = (size_t)n * c;
    hipLaunchKernelGGL(k_balance_pm, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, ctx->stream, tot, (const double*)FtL.p, (const double*)EtL.p,
                       (const double*)nrm2.p, Pm.p, Mm.p);
    auto res = std::make_shared<LDLt>();
    res->n = n;
    if (fold) res->blocks.assign(C.blocks.begin(), C.blocks.begin() + lead_blocks);
    else res->blocks = C.blocks;
    int off = 0;
    for (auto& b : X.blocks) {
        const int k = b.L.cols;
        if (k == 0) continue;
        res->blocks.push_back({Pm.colsview(off, k), b.D, 0.5 * b.alpha, b.diag, false});
        res->blocks.push_back({Mm.colsview(off, k), b.D, -0.5 * b.alpha, b.diag, false});
        off += k;
    }
    ldlt_compress(ctx, *res, tolfac, false, abs_tol);
    return res;
}

static LDLtP gale_residual_impl(Ctx* ctx, const GaleOperator& op, LDLt& C, const LDLtP& X, double tolfac, bool exact, double abs_tol,
                                const Mat* warm_L, const Mat* warm_EtL, int lead_blocks = -1, double e_coeff = 0.0);
LDLtP gale_residual(Ctx* ctx, const GaleOperator& op, LDLt& C, const LDLtP& X, double tolfac, bool exact, double abs_tol) {
    return gale_residual_impl(ctx, op, C, X, tolfac, exact, abs_tol, nullptr, nullptr);
}
static LDLtP gale_residual_impl(Ctx* ctx, const GaleOperator& op, LDLt& C, const LDLtP& X, double tolfac, bool exact, double abs_tol,
                                const Mat* warm_L, const Mat* warm_EtL, int lead_blocks, double e_coeff) {
    auto Cp = std::make_shared<LDLt>(C);
    if (!X || X->iszero()) return ldlt_deepcopy(ctx, Cp);
    const Pencil& P = *op.P;
    if (!exact && P.n <= xblocks_max_n() && (C.blocks.size() > 1 || X->blocks.size() > 1))
        return gale_residual_blocks(ctx, op, C, *X, tolfac, abs_tol, warm_L, warm_EtL, lead_blocks, e_coeff);
    ldlt_destructure(ctx, C, tolfac, exact);
    ldlt_destructure(ctx, *X, tolfac, exact);
    const LBlock& cb = C.blocks[0];
    const LBlock& xb = X->blocks[0];
    const int nG = cb.L.cols, n0 = xb.L.cols, dim = nG + 2 * n0;
    Mat R(ctx, P.n, dim);
    { Mat d = R.colsview(0, nG); copy_mat(ctx, cb.L, d); }
    { Mat d = R.colsview(nG, n0); spmm(ctx, P, P.valEt.p, xb.L, d, 1.0, 0.0); }
    { Mat d = R.colsview(nG + n0, n0); apply_Ft(ctx, op, xb.L, d); }
    Mat T(ctx, dim, dim);
    fill_mat(ctx, T, 0.0);
    { Mat d = T.view(0, 0, nG, nG); copy_mat(ctx, cb.D, d, cb.alpha); }
    { Mat d = T.view(nG, nG + n0, n0, n0); copy_mat(ctx, xb.D, d, xb.alpha); }
    { Mat d = T.view(nG + n0, nG, n0, n0); copy_mat(ctx, xb.D, d, xb.alpha); }
    LDLtP res = ldlt_make(ctx, P.n, R, T, 1.0, false);
    ldlt_compress(ctx, *res, tolfac, exact, abs_tol);
    return res;
}

// =============================================================================================
// dot and LyapunovOperator on the device (the two pieces of the low-rank GMRES that are not ADI, gmres.jl:108-120, LDLt.jl:91-108)
// =============================================================================================
__global__ __launch_bounds__(256) void k_dot_hadamard(int r, int c, const double* __restrict__ A, int lda, const double* __restrict__ B, int ldb, double* out) {
    __shared__ double red[4];
    double s = 0.0;
    for (size_t id = threadIdx.x; id < (size_t)r * c; id += 256) { const int i = id % r, j = id / r; s += A[i + (size_t)j * lda] * B[i + (size_t)j * ldb]; }
    s = wave_sum_t<double>(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = (red[0] + red[1]) + (red[2] + red[3]);
}
double ldlt_dot(Ctx* ctx, const LDLt& X1, const LDLt& X2) {
    DRE_REQUIRE(X1.n == X2.n, "dot: outer dimensions must match");
    if (X1.rank() == 0 || X2.rank() == 0) return 0.0;
    LDLt A = X1, B = X2;                      // shallow copies: concatenation builds new factors, the operands stay untouched
    ldlt_concatenate(ctx, A); ldlt_concatenate(ctx, B);
    const LBlock& a = A.blocks[0]; const LBlock& b = B.blocks[0];
    const int r1 = a.L.cols, r2 = b.L.cols;
    Mat M(ctx, r1, r2), T1(ctx, r1, r2), T2(ctx, r1, r2);
    gemm(ctx, true, false, 1.0, a.L, b.L, 0.0, M, nullptr, "gemm_dot");          // L1' L2
    gemm(ctx, false, false, a.alpha, a.D, M, 0.0, T1, nullptr, "gemm_dot");      // a1 D1 (L1' L2)
    gemm(ctx, false, false, b.alpha, T1, b.D, 0.0, T2, nullptr, "gemm_dot");     // ... a2 D2     (D2 symmetric)
    DevArr<double> out(ctx, 1);
    hipLaunchKernelGGL(k_dot_hadamard, dim3(1), dim3(256), 0, ctx->stream, r1, r2, (const double*)T2.p, T2.ld, (const double*)M.p, M.ld, out.p);
    double h = 0.0;
    DRE_HIP(hipMemcpyAsync(&h, out.p, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    DRE_HIP(hipStreamSynchronize(ctx->stream));
    return h;
}
LDLtP lyapunov_apply(Ctx* ctx, const GaleOperator& op, const LDLtP& X) {
    const Pencil& P = *op.P;
    const int n = P.n;
    if (!X || X->rank() == 0) return ldlt_zero(n);
    LDLt A = *X;
    ldlt_concatenate(ctx, A);
    const LBlock& a = A.blocks[0];
    const int r = a.L.cols;
    Mat L2(ctx, n, 2 * r), D2(ctx, 2 * r, 2 * r);
    { Mat d = L2.colsview(0, r); spmm(ctx, P, P.valEt.p, a.L, d, 1.0, 0.0); }
    { Mat d = L2.colsview(r, r); apply_Ft(ctx, op, a.L, d); }
    fill_mat(ctx, D2, 0.0);
    { Mat d = D2.view(0, r, r, r); copy_mat(ctx, a.D, d, a.alpha); }
    { Mat d = D2.view(r, 0, r, r); copy_mat(ctx, a.D, d, a.alpha); }
    return ldlt_make(ctx, n, L2, D2, 1.0, false);
}

// =============================================================================================
// Algebraic Riccati pieces on the device (riccati/residual.jl:5-52, newton.jl:104-112): the residual
//   R(X) = gamma C'S C + A'XE + E'XA - beta^2 E'XB Rinv B'XE   as ONE LDL' block  [C', A'L, E'L] T [...]'
// and the feedback K' = E'XB, both from the factors of X where they live (no download of L).
// =============================================================================================
static void single_block(Ctx* ctx, LDLt& X) { if (X.blocks.size() > 1) ldlt_concatenate(ctx, X); }
LDLtP gare_residual_dev(Ctx* ctx, const Pencil& P, LDLt& X, const Mat& Ct, const Mat& S, double gamma, const Mat& B, const Mat& Rinv, double beta) {
    const int n = P.n, h = Ct.cols, m = B.cols;
    single_block(ctx, X);
    const int z = X.blocks.empty() ? 0 : X.blocks[0].L.cols;
    Mat R(ctx, n, h + 2 * z), T(ctx, h + 2 * z, h + 2 * z);
    fill_mat(ctx, T, 0.0);
    { Mat d = R.colsview(0, h); copy_mat(ctx, Ct, d); }
    { Mat d = T.view(0, 0, h, h); copy_mat(ctx, S, d, gamma); }
    if (z > 0) {
        auto& b = X.blocks[0];
        { Mat d = R.colsview(h, z); spmm(ctx, P, P.valAt.p, b.L, d, 1.0, 0.0); }
        { Mat d = R.colsview(h + z, z); spmm(ctx, P, P.valEt.p, b.L, d, 1.0, 0.0); }
        { Mat d = T.view(h, h + z, z, z); copy_mat(ctx, b.D, d, b.alpha); }
        { Mat d = T.view(h + z, h, z, z); copy_mat(ctx, b.D, d, b.alpha); }
        Mat BtL(ctx, m, z), BtLD(ctx, m, z), RB(ctx, m, z);
        gemm(ctx, true, false, 1.0, B, b.L, 0.0, BtL);
        gemm(ctx, false, false, b.alpha * beta, BtL, b.D, 0.0, BtLD);
        gemm(ctx, false, false, 1.0, Rinv, BtLD, 0.0, RB);
        Mat d = T.view(h + z, h + z, z, z);
        gemm(ctx, true, false, -1.0, BtLD, RB, 0.0, d);
    }
    return ldlt_make(ctx, n, R, T, 1.0, false);
}
Mat ldlt_feedback_dev(Ctx* ctx, const Pencil& P, LDLt& X, const Mat& B) {
    const int n = P.n, m = B.cols;
    Mat Kt(ctx, n, m);
    single_block(ctx, X);
    if (X.blocks.empty() || X.blocks[0].L.cols == 0) { fill_mat(ctx, Kt, 0.0); return Kt; }
    auto& b = X.blocks[0];
    const int z = b.L.cols;
    Mat LtB(ctx, z, m), DLtB(ctx, z, m), XB(ctx, n, m);
    gemm(ctx, true, false, 1.0, b.L, B, 0.0, LtB);
    gemm(ctx, false, false, b.alpha, b.D, LtB, 0.0, DLtB);
    gemm(ctx, false, false, 1.0, b.L, DLtB, 0.0, XB);
    spmm(ctx, P, P.valEt.p, XB, Kt, 1.0, 0.0);
    return Kt;
}

// =============================================================================================
// ADI (/root/reference/src/lyapunov/adi.jl:29-225)
// =============================================================================================
// Dense inverses whose acceptance test (condition estimate ||M||_F ||inv(M)||_F, two norms in device memory) is still outstanding: a run
// over all shifts of a cycle enqueues every factorisation and inverse first and reads all norms back with ONE synchronisation
// (finalize_dense) instead of two per shift.
struct PendingDense { std::shared_ptr<FactorEntry<double>> fe; Mat W; Mat stack; const void* stack_U = nullptr; int stack_m = -1; };
struct DeferredDense { std::vector<PendingDense> items; DevArr<double> norms; int cap = 0; };
static void finalize_dense(Ctx* ctx, DeferredDense& dd) {
    const int cnt = (int)dd.items.size();
    if (!cnt) return;
    std::vector<double> h((size_t)2 * cnt);
    ctx_fetch(ctx, dd.norms.p, (size_t)2 * cnt * sizeof(double), h.data());
    {   // breakdown flag, pivot growth and static-pivot count of all factors of the cycle with one synchronisation (sparse.hpp)
        std::vector<const Factor<double>*> fs;
        std::vector<int> who;
        for (int i = 0; i < cnt; ++i) if (!dd.items[i].fe->checked) { fs.push_back(&dd.items[i].fe->f); who.push_back(i); }
        if (!fs.empty()) {
            const std::vector<double> gr = mf_check_batch(ctx, fs);
            for (size_t j = 0; j < who.size(); ++j) { dd.items[who[j]].fe->growth = gr[j]; dd.items[who[j]].fe->checked = true; }
        }
    }
    for (int i = 0; i < cnt; ++i) {
        const double cond_est = std::sqrt(h[2 * i]) * std::sqrt(h[2 * i + 1]);
        auto& fe = *dd.items[i].fe;
        // (a factor with replaced pivots belongs to a perturbed matrix: its explicit inverse is not the operator's — sweeps + refinement instead)
        if (cond_est == cond_est && cond_est < 1e7 && fe.f.nperturbed <= 0) {
            fe.dinv = dd.items[i].W; fe.dense = true;
            if (!dd.items[i].stack.empty()) { fe.stack = dd.items[i].stack; fe.stack_U = dd.items[i].stack_U; fe.stack_m = dd.items[i].stack_m; }
        }
    }
    dd.items.clear();
}
template <typename T>
static std::shared_ptr<FactorEntry<T>> get_factor(Ctx* ctx, const GaleOperator& op, FactorCache* cache,
                                                  std::map<std::tuple<uint64_t, double, double>, std::shared_ptr<FactorEntry<T>>>& store,
                                                  std::complex<double> mu, bool want_dense = true, DeferredDense* defer = nullptr, bool check_now = true) {
    auto key = std::make_tuple(op.tag, mu.real(), mu.imag());
    if (cache->enabled) {
        auto it = store.find(key);
        if (it != store.end()) return it->second;
    }
    auto fe = std::make_shared<FactorEntry<T>>();
    mf_factor<T>(ctx, *op.P, op.valFt.p, op.P->valEt.p, make_scalar<T>(1.0, 0.0), make_scalar<T>(mu.real(), mu.imag()), fe->f);
    cache->nfactor++;
    // static pivoting: whether pivots were replaced decides how this factor may be used (refinement, no explicit inverse), so the count is
    // read back here — one synchronisation per NEW factorisation (ten per run with a Cyclic list); the deferred set-up of a whole cycle
    // reads it with its acceptance norms instead (finalize_dense)
    // Single-use factors (self-generated shifts: one new factorisation per ADI iteration) are checked with their chunk instead (AdiRun::check_used):
    // a read-back per iteration would undo the speculative enqueue; a late-detected replaced pivot is handled like a growth warning there.
    if (ctx->pivot_static > 0.0 && !defer && check_now) { fe->growth = mf_check(ctx, fe->f); fe->checked = true; }
    if constexpr (sizeof(T) == sizeof(double)) {
        const int n = op.P->n;
        fe->f.allow_topinv = cache->enabled;     // a factor that keeps being reused gets the dense top-level inverse (sparse.hip)
        if (want_dense && n <= ctx->dense_inv_max_n && fe->f.nperturbed <= 0) {      // only for shifts that will be reused (Cyclic): the inverse costs n solves
            // explicit inverse through n unit right-hand sides; kept only if the operator is well conditioned enough
            // that inverse-times-vector is as accurate as the triangular solves for the ADI recurrences
            Mat W(ctx, n, n);
            set_identity(ctx, W, 1.0);
            mf_solve<double>(ctx, *op.P, fe->f, W.p, W.ld, n, nullptr);
            Mat fv(ctx, op.P->nnz, 1);
            vals_axpby(ctx, op.P->nnz, 1.0, op.valFt.p, mu.real(), op.P->valEt.p, fv.p);
            if (defer && (int)defer->items.size() < defer->cap) {
                const size_t slot = defer->items.size();
                frob2_device(ctx, fv, defer->norms.p + 2 * slot);
                frob2_device(ctx, W, defer->norms.p + 2 * slot + 1);
                PendingDense pd{fe, W};
                {
                    // the stacked inverse [N; E'N; B'N] of this shift in the same helper chain (it was a serial tail of four launches per shift
                    // on the side stream: 0.5 ms of the first time step)
                    const int mm = op.has_lr ? op.U.cols : 0;
                    Mat stk(ctx, 2 * n + mm, n);
                    { Mat top = stk.view(0, 0, n, n); copy_mat(ctx, W, top); }
                    { Mat mid = stk.view(n, 0, n, n); spmm(ctx, *op.P, op.P->valEt.p, W, mid, 1.0, 0.0, nullptr); }
                    if (mm) { Mat bot = stk.view(2 * n, 0, mm, n); gemm(ctx, true, false, 1.0, op.U, W, 0.0, bot, nullptr, "gemm_dinv"); }
                    pd.stack = stk; pd.stack_U = (const void*)op.U.p; pd.stack_m = mm;
                }
                defer->items.push_back(pd);
            } else {
                const double cond_est = frob_norm_host(ctx, fv) * frob_norm_host(ctx, W);
                if (cond_est == cond_est && cond_est < 1e7) { fe->dinv = W; fe->dense = true; }
            }
        }
    }
    if (cache->enabled) { store[key] = fe; cache->fresh.push_back(key); }
    return fe;
}

__global__ void k_join_cplx(int rows, int cols, const double* __restrict__ re, const double* __restrict__ im, int lds_, cplx* __restrict__ dst, int ldd) {
    size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= (size_t)rows * cols) return;
    int r = id % rows, c = id / rows;
    dst[r + (size_t)c * ldd] = {re[r + (size_t)c * lds_], im ? im[r + (size_t)c * lds_] : 0.0};
}
// The plug-in point of the reference's BlockLinearSolver protocol: the user solves the sparse shifted system for the block right-hand
// side W (n x ncols, solver ordering, overwritten by the solution).  The panel is handed over in the caller's row ordering after a
// stream synchronisation; the callback must have finished its own device work when it returns.
static void user_block_solve(Ctx* ctx, const GaleOperator& op, const AdiOptions& opt, std::complex<double> mu, Mat& W, cplx* Wc) {
    const Pencil& P = *op.P;
    const int n = P.n, nc = W.cols;
    Mat Bu(ctx, n, nc), Xr(ctx, n, nc), Xi;
    permute_rows(ctx, W, P.iperm.p, Bu);                  // Bu(old, :) = W(iperm[old], :)
    const bool cx = mu.imag() != 0.0;
    if (cx) Xi = Mat(ctx, n, nc);
    ctx->sync();
    const int rc = opt.inner_solve(opt.inner_user, n, nc, op.cA, op.cE + mu.real(), mu.imag(), Bu.p, Xr.p, cx ? Xi.p : nullptr);
    if (rc != 0) throw Error(ERR_INTERNAL, "user block solver (inner_alg) failed with code " + std::to_string(rc));
    if (!cx) { permute_rows(ctx, Xr, P.perm.p, W); return; }
    Mat Xr2(ctx, n, nc), Xi2(ctx, n, nc);
    permute_rows(ctx, Xr, P.perm.p, Xr2); permute_rows(ctx, Xi, P.perm.p, Xi2);
    const size_t tot = (size_t)n * nc;
    hipLaunchKernelGGL(k_join_cplx, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, ctx->stream, n, nc, (const double*)Xr2.p, (const double*)Xi2.p, n, Wc, n);
}

Mat smw_solve(Ctx* ctx, const Pencil& P, const Factor<double>& F, double alpha, const Mat& U, const Mat& Vt, const Mat& B) {
    const int n = P.n, k = B.cols, m = U.cols;
    DRE_REQUIRE(U.rows == n && Vt.rows == n && Vt.cols == m && B.rows == n, "smw_solve: shape mismatch");
    DRE_REQUIRE(m >= 1 && m <= 32, "SMW: between 1 and 32 low-rank columns (DRE_SMW_MAX_RANK)");
    Mat W(ctx, n, k + m), X(ctx, n, k);
    { Mat d = W.colsview(0, k); copy_mat(ctx, B, d); }
    { Mat d = W.colsview(k, m); copy_mat(ctx, Vt, d); }
    mf_solve<double>(ctx, P, F, W.p, W.ld, k + m, nullptr);
    Mat small(ctx, m, k + m);
    gemm(ctx, true, false, 1.0, U, W, 0.0, small, nullptr, "smw_small");
    DevArr<double> sinv(ctx, (size_t)m * m);
    DevArr<int> serr(ctx, 1);
    DRE_HIP(hipMemsetAsync(serr.p, 0, sizeof(int), ctx->stream));
    hipLaunchKernelGGL((k_sinv<double>), dim3(1), dim3(64), 0, ctx->stream, m, small.p + (size_t)k * small.ld, small.ld, alpha, sinv.p, (const AdiState*)nullptr, serr.p);
    hipLaunchKernelGGL((k_smw_apply<double, true>), dim3(ceil_div(n, 256), ceil_div(k, SMW_CB)), dim3(256), 0, ctx->stream, n, m, k, W.p, W.ld,
                       (const double*)(W.p + (size_t)k * W.ld), W.ld, (const double*)sinv.p, (const double*)small.p, small.ld, X.p, X.ld, (double*)nullptr, 0, 0.0,
                       (const AdiState*)nullptr);
    int herr = 0;
    DRE_HIP(hipMemcpyAsync(&herr, serr.p, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    DRE_HIP(hipStreamSynchronize(ctx->stream));
    if (herr) throw Error(ERR_SINGULAR, "SMW: capacitance matrix is singular");
    return X;
}
void smw_solve(Ctx* ctx, const Pencil& P, const Factor<cplx>& F, double alpha, const Mat& U, const Mat& Vt, const Mat& B, Mat& X_re, Mat& X_im) {
    const int n = P.n, k = B.cols, m = U.cols;
    DRE_REQUIRE(U.rows == n && Vt.rows == n && Vt.cols == m && B.rows == n, "smw_solve: shape mismatch");
    DRE_REQUIRE(m >= 1 && m <= 32, "SMW: between 1 and 32 low-rank columns (DRE_SMW_MAX_RANK)");
    const int nc = k + m;
    DevArr<cplx> W(ctx, (size_t)n * nc), small(ctx, (size_t)m * nc), sinv(ctx, (size_t)m * m), Xc(ctx, (size_t)n * k);
    DevArr<int> serr(ctx, 1);
    DRE_HIP(hipMemsetAsync(serr.p, 0, sizeof(int), ctx->stream));
    size_t tot = (size_t)n * k;
    hipLaunchKernelGGL(k_real_to_cplx, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, ctx->stream, n, k, B.p, B.ld, W.p, n, (const AdiState*)nullptr);
    tot = (size_t)n * m;
    hipLaunchKernelGGL(k_real_to_cplx, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, ctx->stream, n, m, Vt.p, Vt.ld, W.p + (size_t)k * n, n, (const AdiState*)nullptr);
    mf_solve<cplx>(ctx, P, F, W.p, n, nc, nullptr);
    hipLaunchKernelGGL((k_smw_small<cplx>), dim3(nc), dim3(256), 0, ctx->stream, n, m, U.p, U.ld, (const cplx*)W.p, n, small.p, m, (const AdiState*)nullptr);
    hipLaunchKernelGGL((k_sinv<cplx>), dim3(1), dim3(64), 0, ctx->stream, m, (const cplx*)(small.p + (size_t)k * m), m, alpha, sinv.p, (const AdiState*)nullptr, serr.p);
    // X = W_B - W_Vt (Sinv small_B): the apply kernel's complex epilogue produces the real ADI pair, so the plain complex result is formed here
    X_re = Mat(ctx, n, k); X_im = Mat(ctx, n, k);
    hipLaunchKernelGGL(k_smw_plain_cplx, dim3(ceil_div(n, 256), k), dim3(256), 0, ctx->stream, n, m, (const cplx*)W.p, n, (const cplx*)(W.p + (size_t)k * n), n,
                       (const cplx*)sinv.p, (const cplx*)small.p, m, X_re.p, X_im.p, X_re.ld);
    int herr = 0;
    DRE_HIP(hipMemcpyAsync(&herr, serr.p, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    DRE_HIP(hipStreamSynchronize(ctx->stream));
    if (herr) throw Error(ERR_SINGULAR, "SMW: capacitance matrix is singular");
}

struct SmwCacheEntry { BufP keep; void* WU; int ldwu; BufP sinv; BufP keep2; };
std::vector<std::complex<double>> heuristic_shift_values(Ctx* ctx, const GaleOperator& op, int nshifts, int kplus, int kminus, int* warnings);

// The solver object of one Lyapunov solve (the reference's ADICache, adi.jl:5-21): adi_begin = init (adi.jl:29-69), adi_advance = step!
// / solve! (adi.jl:71-128; one call enqueues up to `budget` shifts speculatively and synchronises once), adi_finish = the tail of solve!
// (final compression adi.jl:78-80, result).  adi_solve runs the three in sequence.
// =============================================================================================
// Fan groups (round 3, general path): g consecutive real-shift ADI iterations from g INDEPENDENT solves with the same right-hand side.
// With Z_s = (A' + mu_s E')^-1 (low-rank term included) the resolvent identity  Z_a - Z_b = (mu_b - mu_a) Z_b E' Z_a  turns the iterates of
// adi.jl:158-171,   V_j = Z_j R_{j-1},  R_j = R_{j-1} - 2 mu_j E' V_j   (j = 1..g, from R_0),   into partial fractions of  W_s = Z_s R_0:
//     V_j = sum_{s<=j} c_js W_s,     c_js = prod_{i<j} (-mu_s - mu_i) / prod_{i<=j, i!=s} (mu_i - mu_s),
//     R_j = R_0 - E' Y_j,            Y_j = 2 sum_{i<=j} mu_i V_i = sum_{s<=j} d_js W_s,   d_js = 2 sum_{i=s..j} mu_i c_is.
// The g multifrontal solves (+ SMW corrections) are latency bound and use a fraction of the chip each: they SHARE every launch (round 4:
// blockIdx.z = shift, sparse.hip mf_solve_batch; round 3 ran them side by side on g streams, 13 launches each); one pass over E' forms all V_j and
// all residuals R_j = R_0 - sum_s d_js E' W_s (sparse.hip, k_fan_spmm_mix), and the norms/decisions follow in iteration order (one batched Gram
// product + one decision launch for the group).  The coefficients grow when shifts of a group are close (~ mu / delta mu per pair): groups are cut so that max_j sum_s |c_js| stays
// below fan_max_coef (the products W_s are accurate to ~eps cond, the combination amplifies that by the coefficient sum).
// =============================================================================================
static double fan_coefficients(const double* mu, int g, FanCoef* out) {
    long double c[FAN_GMAX][FAN_GMAX] = {{0}}, d[FAN_GMAX][FAN_GMAX] = {{0}};
    double worst = 0.0;
    for (int j = 0; j < g; ++j) {
        long double sum = 0.0L;
        for (int s = 0; s <= j; ++s) {
            long double num = 1.0L, den = 1.0L;
            for (int i = 0; i < j; ++i) num *= -(long double)mu[s] - (long double)mu[i];
            for (int i = 0; i <= j; ++i) if (i != s) den *= (long double)mu[i] - (long double)mu[s];
            c[j][s] = num / den;
            sum += fabsl(c[j][s]);
        }
        worst = std::max(worst, (double)sum);
    }
    for (int j = 0; j < g; ++j)
        for (int s = 0; s <= j; ++s) {
            long double acc = 0.0L;
            for (int i = s; i <= j; ++i) acc += 2.0L * (long double)mu[i] * c[i][s];
            d[j][s] = acc;
        }
    for (int j = 0; j < FAN_GMAX; ++j) for (int s = 0; s < FAN_GMAX; ++s) { out->c[j][s] = (double)c[j][s]; out->d[j][s] = (double)d[j][s]; }
    return worst;
}
// SMW of the g solves of a fan group in one launch (blockIdx.z = solve): W_z <- W_z - W_U,z (Sinv_z small_z) in place;  W_z / small_z = columns
// z k .. of the n x (g k) panel / of small = U' W  (smw.jl:36-43)
struct SmwZ { const double* WU[MF_ZMAX]; const double* Sinv[MF_ZMAX]; int ldwu[MF_ZMAX]; };
__global__ __launch_bounds__(256) void k_smw_apply_z(int n, int m, int k, double* __restrict__ W, int ldw, SmwZ sz, const double* __restrict__ small, int lds_,
                                                     const AdiState* st) {
    if (st && st->done) return;
    __shared__ double y[SMW_CB * 32];
    const int z = blockIdx.z;
    const int c0 = blockIdx.y * SMW_CB, kc = min(SMW_CB, k - c0);
    const int tid = threadIdx.x;
    const double* __restrict__ Sinv = sz.Sinv[z];
    const double* __restrict__ WU = sz.WU[z];
    const int ldwu = sz.ldwu[z];
    for (int id = tid; id < kc * m; id += 256) {
        const int j = id % m, c = id / m;
        double acc = 0.0;
        for (int l = 0; l < m; ++l) acc += Sinv[j + (size_t)l * m] * small[l + (size_t)(z * k + c0 + c) * lds_];
        y[j + c * 32] = acc;
    }
    __syncthreads();
    const int i = blockIdx.x * 256 + tid;
    if (i >= n) return;
    double* __restrict__ w = W + i + (size_t)(z * k + c0) * ldw;
    double v[SMW_CB];
#pragma unroll
    for (int c = 0; c < SMW_CB; ++c) v[c] = w[(size_t)min(c, kc - 1) * ldw];
    for (int j = 0; j < m; ++j) {
        const double wuj = WU[i + (size_t)j * ldwu];
#pragma unroll
        for (int c = 0; c < SMW_CB; ++c) v[c] -= wuj * y[j + c * 32];
    }
#pragma unroll
    for (int c = 0; c < SMW_CB; ++c) if (c < kc) w[(size_t)c * ldw] = v[c];
}
// the capacitance matrices of several shifts inverted in one launch (blockIdx.x = shift): Sinv_z = inv(alpha I + small[:, z m .. (z+1) m))
struct SinvZ { double* out[MF_ZMAX]; };
__global__ __launch_bounds__(64) void k_sinv_z(int m, const double* __restrict__ small, int lds_, double alpha, SinvZ iz, const AdiState* st, int* err) {
    if (st && st->done) return;
    sinv_body<double>(m, small + (size_t)blockIdx.x * m * lds_, lds_, alpha, iz.out[blockIdx.x], err);
}
// helper context h of a context (own stream and pool), created on first use
static Ctx* helper_ctx(Ctx* ctx, int h) {
    while ((int)ctx->helpers.size() <= h) {
        auto hc = std::make_unique<Ctx>();
        hc->device = ctx->device; hc->num_cus = ctx->num_cus;
        hc->stream = create_stream(2);
        hc->timer = std::make_unique<KernelTimer>();
        hipEvent_t ev;
        DRE_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        ctx->helpers.push_back(std::move(hc)); ctx->helper_ev.push_back(ev);
    }
    Ctx* hc = ctx->helpers[(size_t)h].get();
    hc->timer->enabled = ctx->timer && ctx->timer->enabled;
    return hc;
}
static hipEvent_t aux_event(Ctx* ctx, int i) {
    if (!ctx->aux_ev[i]) DRE_HIP(hipEventCreateWithFlags(&ctx->aux_ev[i], hipEventDisableTiming));
    return ctx->aux_ev[i];
}
struct StepRec { int iters_after; size_t nblocks; int nshifts; Mat Rafter; };      // Rafter: the residual factor after this iteration where it is NOT updated in place (fan groups)
struct AdiRun {
    Ctx* ctx = nullptr;
    GaleOperator op;
    AdiOptions opt_in, opt;
    FactorCache local;
    FactorCache* cache = nullptr;
    AdiResult res;
    LDLtP X, resid;
    Mat R, Tm;
    double alpha_res = 1.0, abstol = 0.0, ctf = 4.0;
    bool cex = false, tdiag = false;
    int n = 0, k = 0, m = 0;
    std::unique_ptr<ShiftOracle> oracle;
    DevArr<AdiState> st;
    AdiState h0;
    std::map<std::pair<double, double>, SmwCacheEntry> smw_cache;
    std::shared_ptr<LDLt> Xw;
    int iters_host = 0, last_compression = 0, chunk_limit = 10;
    std::vector<std::complex<double>> all_shifts;
    bool finished = false, finalized = false;
    std::vector<std::shared_ptr<FactorEntry<double>>> used_real;
    std::vector<std::shared_ptr<FactorEntry<cplx>>> used_cplx;
    DenseNormPending npend;
    // fast chain
    bool fast = false, fast_ready = false;
    std::vector<std::shared_ptr<FactorEntry<double>>> fast_fe;
    std::vector<double*> fast_pack;
    std::vector<Mat> fast_keep;
    Mat Gm;
    DevArr<double> nws;
    size_t cyc = 0;
    // every factorisation is checked once per chunk, after the chunk's synchronisation (the breakdown flag is written by the
    // factorisation kernels only); the handles are dropped then, so single-use factors are freed chunk by chunk
    double max_growth = 0.0;         // largest pivot growth among the factorisations this solve used
    LDLt Crhs;                       // the right-hand side (shallow copy) for the true-residual verification after a growth warning
    // single-use factors (Projection / per-solve Heuristic shifts): factorised AHEAD on the helper streams while the current iteration solves
    // (all shifts of a batch are known at once and their factorisations are independent), checked lazily with their chunk
    struct Prefetched { hipEvent_t ev; long ticket; };
    std::map<std::pair<double, double>, Prefetched> prefetch_ev;     // factor of this shift is being made on a helper stream: wait for the event before use
    std::vector<hipEvent_t> ev_pool;
    bool check_now = false;           // a lazily checked factor turned out to have replaced pivots: from now on every new factor is checked at once
    size_t prefetch_rr = 0;
    bool helpers_ready = false;
    bool chunk_from_hint = false;
    hipEvent_t tol_event = nullptr;   // the tolerance is being formed on a helper stream: wait for this event, then decide iteration 0 (apply_tolerance)
    bool abstol_pending = false;      // the tolerance was formed on the device (AdiOptions::normC_build): the host copy follows with the first chunk
    bool defer = false;               // the tolerance is still on its way (AdiOptions::normC_dev): kernels record norms, decisions follow at the chunk end
    double reltol = 0.0;
    bool hist_ok = true;
    bool fan_off = false;             // the batched fan form does not apply to this solve's factors (sparse.hip, mf_solve_batch): one iteration at a time
    bool fan_smw_all = false;         // the SMW products of every cached shift of the cycle were formed with the first group's
    void check_used() {
        {   // the unchecked real factors with ONE synchronisation (the ten factorisations of a freshly factorised cycle: ten read-backs before)
            std::vector<const Factor<double>*> fs; std::vector<FactorEntry<double>*> es;
            for (auto& f : used_real) {
                bool seen = false;
                for (auto* e : es) seen = seen || e == f.get();
                if (!f->checked && !seen) { fs.push_back(&f->f); es.push_back(f.get()); }
            }
            if (fs.size() >= 2) {
                const std::vector<double> gr = mf_check_batch(ctx, fs);
                for (size_t i = 0; i < es.size(); ++i) {
                    es[i]->growth = gr[i]; es[i]->checked = true;
                    if (es[i]->f.nperturbed > 0) { check_now = true; max_growth = std::max(max_growth, 1e300); }
                }
            }
        }
        for (auto& f : used_real) {
            if (!f->checked) { f->growth = mf_check(ctx, f->f); f->checked = true; if (f->f.nperturbed > 0) { check_now = true; max_growth = std::max(max_growth, 1e300); } }
            max_growth = std::max(max_growth, f->growth);
        }
        for (auto& f : used_cplx) {
            if (!f->checked) { f->growth = mf_check(ctx, f->f); f->checked = true; if (f->f.nperturbed > 0) { check_now = true; max_growth = std::max(max_growth, 1e300); } }
            max_growth = std::max(max_growth, f->growth);
        }
        used_real.clear(); used_cplx.clear();
    }
    ~AdiRun() {
        for (auto& kv : prefetch_ev) { if (ctx) (void)hipStreamWaitEvent(ctx->stream, kv.second.ev, 0); (void)hipEventDestroy(kv.second.ev); }
        for (auto e : ev_pool) (void)hipEventDestroy(e);
    }
};

std::shared_ptr<AdiRun> adi_begin(Ctx* ctx, const GaleOperator& op_in, LDLt& C, const LDLtP& initial_guess, const AdiOptions& opt_in,
                                  FactorCache* cache) {
    auto runp = std::make_shared<AdiRun>();
    AdiRun& run = *runp;
    run.ctx = ctx; run.op = op_in; run.opt_in = opt_in;
    const GaleOperator& op = run.op;
    const Pencil& P = *op.P;
    const int n = P.n;
    if (!cache) cache = &run.local;
    run.cache = cache;
    AdiResult& res = run.res;
    AdiOptions opt_h;                           // Cyclic(Heuristic(...)): the values are recomputed from (E, F) for this solve (adi.jl:54)
    const AdiOptions* optp = &opt_in;
    if (opt_in.shifts.kind == ShiftSpec::HEURISTIC) {
        opt_h = opt_in;
        opt_h.shifts.kind = ShiftSpec::CYCLIC;
        opt_h.shifts.values = heuristic_shift_values(ctx, op, opt_in.shifts.h_nshifts, opt_in.shifts.h_kplus, opt_in.shifts.h_kminus, &res.warnings);
        optp = &opt_h;
    }
    run.opt = *optp;
    const AdiOptions& opt = run.opt;
    const double ctf = opt.compress_tolfac;
    const bool cex = opt.compress_exact;
    const bool keep_blocks = !cex && n <= xblocks_max_n() && C.blocks.size() > 1 && initial_guess && !opt.ignore_initial_guess && !initial_guess->iszero();
    double normC = 0.0;
    const bool defer = opt.given_residual && opt.abstol < 0 && opt.normC_dev != nullptr;      // ||C|| arrives in device memory, later
    DRE_REQUIRE(!opt.given_residual || defer || opt.abstol >= 0, "ADI with a given residual needs abstol or a device-side ||C||");
    if (opt.given_residual) { /* no right-hand side object */ }
    else if (keep_blocks) normC = ldlt_norm_dense_small(ctx, C);       // the summands go into the residual as they are (gale_residual_blocks)
    else { ldlt_destructure(ctx, C, ctf, cex); normC = ldlt_norm(ctx, C); }
    const double reltol = opt.reltol >= 0 ? opt.reltol : n * EPS;
    const double abstol = defer ? -1.0 : (opt.abstol >= 0 ? opt.abstol : reltol * normC);      // (-1: no norm is ever at or below it — the kernels only record)
    run.defer = defer; run.reltol = reltol;
    LDLtP X = (opt.ignore_initial_guess || !initial_guess) ? ldlt_zero(n) : initial_guess;
    LDLtP resid;
    if (opt.given_residual) {
        // the caller knows the warm-start residual in factored form (Rosenbrock-1 recurrence): truncated at the PREVIOUS step's level while this
        // step's tolerance is still being formed; the solve returns the increment
        resid = opt.given_residual;
        X = ldlt_zero(n);
        const double lag = opt.abstol_lag > 0.0 ? opt.abstol_lag : (opt.abstol >= 0.0 ? opt.abstol : -1.0);
        if (resid->blocks.size() > 1) {
            bool done = false;
            static const bool warm_on = true;
            // (not attempted where it cannot apply — a basis of fewer than 16 columns, q0 + sx <= 64 — so that such a step does not count as a rejection)
            if (warm_on && opt.warm_basis.cols >= 16 && lag > 0.0 && cache->warm_strikes < 2 &&
                opt.warm_basis.cols + (cache->warm_sx > 0 ? cache->warm_sx : 32) > 64) {
                const int sx = cache->warm_sx > 0 ? cache->warm_sx : 32;
                double missed = 0.0;
                done = warm_compress(ctx, *resid, opt.warm_basis, ctf, opt.residual_abs_frac * lag, sx, &missed);
                if (done) cache->warm_strikes = 0;
                else if (sx < 64) cache->warm_sx = 64;          // more fresh directions next time; two failures in a row at 64: the full reduction from then on
                else cache->warm_strikes += 1;
            }
            if (!done) ldlt_compress(ctx, *resid, ctf, false, lag > 0.0 ? opt.residual_abs_frac * lag : -1.0);
        }
    } else {
    // Krylov mode: components of the warm-start residual far below the convergence tolerance are dropped
    resid = gale_residual_impl(ctx, op, C, X, ctf, cex, cex ? -1.0 : opt.residual_abs_frac * abstol,
                                     opt.warm_L.empty() ? nullptr : &opt.warm_L, opt.warm_EtL.empty() ? nullptr : &opt.warm_EtL,
                                     opt.rhs_lead_blocks, opt.rhs_e_coeff);
    }
    ldlt_destructure(ctx, *resid, ctf, cex);
    LBlock rb = resid->blocks[0];
    Mat& R = run.R; Mat& Tm = run.Tm;
    R = rb.L; Tm = rb.D;
    const double alpha_res = rb.alpha;
    const int k = R.cols;
    const bool tdiag = rb.diag;      // a numerically diagonal T that is not flagged takes the general (dense-T) kernels: same result
    const double norm0 = defer ? 1e300 : ldlt_norm_host(ctx, R, Tm, alpha_res);      // deferred: formed on the device below, read with the first chunk
    res.abstol = abstol; res.initial_norm = norm0; res.rhs_cols = k;
    res.norms.push_back(norm0); res.norm_iters.push_back(0);
    res.residual = resid;
    res.X = X;
    res.res_norm = norm0;
    DRE_REQUIRE(opt.maxiters >= 0 && opt.maxiters <= DRE_ADI_MAX_ITERS_LIMIT, "ADI: maxiters out of range (dre_hip.h, DRE_ADI_MAX_ITERS)");
    run.X = X; run.resid = resid; run.alpha_res = alpha_res; run.abstol = abstol; run.ctf = ctf; run.cex = cex; run.tdiag = tdiag;
    run.n = n; run.k = k; run.m = op.has_lr ? op.U.cols : 0;
    run.Xw = std::make_shared<LDLt>(*X);        // the iterate: never mutate the caller's initial guess (adi.jl:174 builds a new list)
    run.Crhs = C;
    if (norm0 <= abstol || k == 0) { res.converged = true; run.finished = true; return runp; }

    std::unique_ptr<ShiftOracle>& oracle = run.oracle;
    if (opt.shifts.kind == ShiftSpec::CYCLIC) {
        DRE_REQUIRE(!opt.shifts.values.empty(), "Cyclic shifts: empty list");
        auto o = std::make_unique<CyclicOracle>();
        o->v = opt.shifts.values;
        oracle = std::move(o);
    } else {
        auto o = std::make_unique<ProjectionOracle>();
        o->ctx = ctx; o->op = &op; o->n_history = opt.shifts.n_history;
        oracle = std::move(o);
    }
    oracle->update(R, {});

    // device-resident control block
    run.st = DevArr<AdiState>(ctx, 1);
    DevArr<AdiState>& st = run.st;
    if (auto* po = dynamic_cast<ProjectionOracle*>(oracle.get())) po->st_dev = st.p;
    AdiState& h0 = run.h0;           // stays alive with the solver object (source of an asynchronous upload)
    std::memset(&h0, 0, sizeof(h0));
    h0.maxiters = opt.maxiters; h0.abstol = abstol; h0.res_norm = norm0; h0.norms[0] = norm0;
    DRE_HIP(hipMemcpyAsync(st.p, &h0, sizeof(AdiState), hipMemcpyHostToDevice, ctx->stream));      // (whole block: the ticket of the fan groups' norm launch starts at 0)
    if (defer) {
        // norm of the initial residual as iteration 0 of the record, on the device (no host round trip)
        Mat G0(ctx, k, k);
        gemm(ctx, true, false, 1.0, R, R, 0.0, G0, nullptr, "gemm_gram");
        ldlt_norm_update_state(ctx, G0, Tm, tdiag, alpha_res, st.p, 0);
        if (opt.normC_build) {
            // the tolerance is formed right here, on this stream: the decisions are live from iteration 0 on; the host reads it with the first chunk
            // ... on a helper stream, beside the SMW set-up and the first group's sweeps; the main stream picks it up in front of its first norm
            Ctx* hc = helper_ctx(ctx, 0);
            hipEvent_t e0 = aux_event(ctx, 0), e1 = aux_event(ctx, 1);
            DRE_HIP(hipEventRecord(e0, ctx->stream));
            DRE_HIP(hipStreamWaitEvent(hc->stream, e0, 0));
            opt.normC_build(hc, R, Tm, alpha_res);
            DRE_HIP(hipEventRecord(e1, hc->stream));
            run.tol_event = e1;
            run.defer = false; run.abstol_pending = true;
        }
    }
    const int m = op.has_lr ? op.U.cols : 0;
    DRE_REQUIRE(m <= 32, "SMW: more than 32 low-rank columns not supported (dre_hip.h, DRE_SMW_MAX_RANK)");
    auto& smw_cache = run.smw_cache;
    int* const serr = &st.p->smw_singular;     // lives in the control block: comes back with every chunk synchronisation
    if (op.has_lr && opt.shifts.kind == ShiftSpec::CYCLIC) {
        // Dense-inverse path: the SMW products of ALL shifts of the cycle whose stacked inverses already exist (i.e. from the
        // second time step on) are formed up front — one batched GEMM, one batched capacitance inversion, one batched fold —
        // instead of three launches per shift inside the loop.
        std::vector<GemmBatchDesc> descs;
        std::vector<SmwBatch> hb;
        std::vector<std::pair<double, SmwCacheEntry>> pending;
        for (auto& mu : opt.shifts.values) {
            if (mu.imag() != 0.0 || smw_cache.count({mu.real(), 0.0})) continue;
            bool dup = false;
            for (auto& pe : pending) dup = dup || pe.first == mu.real();
            if (dup) continue;
            auto it = cache->real.find(std::make_tuple(op.tag, mu.real(), 0.0));
            if (it == cache->real.end()) continue;
            auto& fe = it->second;
            if (!fe->dense || fe->stack.empty() || fe->stack_U != (const void*)op.U.p || fe->stack_m != m) continue;
            Mat WK(ctx, 2 * n + m, m), WKS(ctx, 2 * n, m);
            SmwCacheEntry en;
            en.keep = WKS.buf; en.WU = WKS.p; en.ldwu = WKS.ld;
            en.sinv = std::make_shared<Buf>(ctx, (size_t)m * m * sizeof(double));
            en.keep2 = WK.buf;
            descs.push_back({fe->stack.p, op.Vt.p, WK.p, nullptr, 1.0, 2 * n + m, m, n, fe->stack.ld, op.Vt.ld, WK.ld, 0});
            hb.push_back({WK.p, (double*)en.sinv->p, WKS.p});
            pending.push_back({mu.real(), en});
        }
        if (!descs.empty()) {
            gemm_batched(ctx, descs, "gemm_dinv");
            DevArr<SmwBatch> db(ctx, hb.size());
            DRE_HIP(hipMemcpyAsync(db.p, hb.data(), hb.size() * sizeof(SmwBatch), hipMemcpyHostToDevice, ctx->stream));
            hipLaunchKernelGGL(k_sinv_batched, dim3((unsigned)hb.size()), dim3(64), 0, ctx->stream, n, m, 2 * n + m, op.alpha, (const SmwBatch*)db.p,
                               (const AdiState*)st.p, serr);
            hipLaunchKernelGGL(k_fold_sinv_batched, dim3(ceil_div(2 * n * m, 256), (unsigned)hb.size()), dim3(256), 0, ctx->stream, 2 * n, m, 2 * n + m,
                               (const SmwBatch*)db.p, (const AdiState*)st.p);
            for (auto& pe : pending) smw_cache.emplace(std::make_pair(pe.first, 0.0), pe.second);
        }
    }

    // General path, Cyclic real list, one rank: every shift of the cycle that has no factor yet is factorised NOW, all of them in shared
    // launches (sparse.hip, mf_factor_batch: one launch per tree level for the whole list) together with their dense top inverses — round 3
    // ran them as ten chains on five helper streams: 4 + 1.6 + 1.6 ms of the first time step at n = 5177 went into waiting for them.
    if (opt.shifts.kind == ShiftSpec::CYCLIC && !opt.inner_solve && cache->enabled && n > ctx->dense_inv_max_n && ctx->adi_fan >= 2 && P.use_mfma_sweeps &&
        !(ctx->comm && ctx->comm->nranks > 1 && ctx->comm->emulate <= 1)) {
        std::vector<double> todo;
        for (auto& mu : opt.shifts.values) {
            if (mu.imag() != 0.0) { todo.clear(); break; }
            bool dup = cache->real.count(std::make_tuple(op.tag, mu.real(), 0.0)) > 0;
            for (double t : todo) dup = dup || t == mu.real();
            if (!dup && (int)todo.size() < MF_ZMAX) todo.push_back(mu.real());
        }
        if (todo.size() >= 2) {
            std::vector<std::shared_ptr<FactorEntry<double>>> fes;
            std::vector<Factor<double>*> fp;
            for (double t : todo) { (void)t; fes.push_back(std::make_shared<FactorEntry<double>>()); fp.push_back(&fes.back()->f); }
            mf_factor_batch<double>(ctx, P, op.valFt.p, P.valEt.p, 1.0, todo.data(), fp.data(), (int)todo.size());
            for (size_t z = 0; z < todo.size(); ++z) {
                fes[z]->f.allow_topinv = true;
                const auto key = std::make_tuple(op.tag, todo[z], 0.0);
                cache->real[key] = fes[z]; cache->fresh.push_back(key); cache->nfactor++;
                run.used_real.push_back(fes[z]);                 // pivots / growth are read back with the first chunk
            }
            mf_topinv_batch(ctx, P, fp.data(), (int)todo.size());
        }
    }

    // chunk length: compression_interval, or — where the intermediate compressions are deferred anyway — the iteration count of the
    // previous solve (+2), so that a whole Lyapunov solve is enqueued before the first host synchronisation
    run.chunk_limit = (!cex && n <= xblocks_max_n() && cache->iters_hint > 0) ? std::max(opt.compression_interval, cache->iters_hint + 2)
                                                                          : opt.compression_interval;
    // The same on the multifrontal path with a Cyclic list (fan groups): the intermediate compressions are deferred there as long as the factor
    // fits the factor form (adi_advance), so a whole solve of the previous length is enqueued before the first synchronisation — provided the
    // uncompressed iterate still fits afterwards with a further compression interval to spare.
    if (!cex && opt.compression && opt.shifts.kind == ShiftSpec::CYCLIC && !opt.inner_solve && cache->enabled && cache->iters_hint > 0 &&
        n > ctx->dense_inv_max_n && n >= ctx->compress_factor_min_n && k > 0) {
        const long room = ((long)n - 64 - X->rank()) / k - 2L * opt.compression_interval - FAN_GMAX;
        run.chunk_limit = (int)std::max<long>(opt.compression_interval, std::min<long>(cache->iters_hint + 1, room));
        run.chunk_from_hint = true;       // exactly that many iterations: the last group of the chunk is cut short (a speculative group costs 14 launches)
    }
    // ---- fast chain (dense.hip, k_adi_fast): every shift of the cycle is real and already has its stacked dense inverse for the
    // current low-rank factor (i.e. from the second time step of a run on) and the residual is at most 96 columns wide ----------
    static const bool fast_env = true;
    bool fast = fast_env && !opt_in.inner_solve && opt_in.shifts.kind == ShiftSpec::CYCLIC && k >= 1 && k <= ADI_FAST_MAX_K && n <= ctx->dense_inv_max_n && cache->enabled;
    auto& fast_fe = run.fast_fe;            // per position of the cycle
    auto& fast_pack = run.fast_pack;
    auto& fast_keep = run.fast_keep;
    if (fast) {
        const int mm = op.has_lr ? m : 0;
        std::map<double, double*> by_mu;
        std::vector<const double*> stacks, wks; std::vector<double*> outs;
        for (auto& mu : opt.shifts.values) {
            if (mu.imag() != 0.0) { fast = false; break; }
            auto it = cache->real.find(std::make_tuple(op.tag, mu.real(), 0.0));
            if (it == cache->real.end() || !it->second->dense || it->second->stack.empty() || it->second->stack_m != mm ||
                (mm && it->second->stack_U != (const void*)op.U.p)) { fast = false; break; }
            auto sc = smw_cache.find({mu.real(), 0.0});
            if (mm && sc == smw_cache.end()) { fast = false; break; }
            fast_fe.push_back(it->second);
            auto bm = by_mu.find(mu.real());
            if (bm == by_mu.end()) {
                Mat pk(ctx, (int)adi_fast_pack_doubles(n) / 64, 64);
                fast_keep.push_back(pk);
                stacks.push_back(it->second->stack.p);
                wks.push_back(mm ? (const double*)sc->second.WU : nullptr);
                outs.push_back(pk.p);
                bm = by_mu.emplace(mu.real(), pk.p).first;
            }
            fast_pack.push_back(bm->second);
        }
        if (fast) adi_fast_build(ctx, n, mm, stacks, 2 * n + mm, wks, 2 * n, outs);
    }
    run.fast = fast;
    return runp;
}

// One chunk: up to `budget` shifts are enqueued speculatively (at least one; a conjugate pair counts two and is never split), then one
// synchronisation tells how far the device got.  budget = 1 is the reference's step! (adi.jl:97-128).
void adi_advance(AdiRun& run, int budget) {
    if (run.finished) return;
    Ctx* ctx = run.ctx;
    const GaleOperator& op = run.op;
    const Pencil& P = *op.P;
    const int n = run.n, k = run.k, m = run.m;
    FactorCache* cache = run.cache;
    AdiResult& res = run.res;
    const AdiOptions& opt = run.opt;
    const AdiOptions& opt_in = run.opt_in;
    const double ctf = run.ctf, alpha_res = run.alpha_res;
    const bool cex = run.cex, tdiag = run.tdiag;
    Mat& R = run.R; Mat& Tm = run.Tm;
    auto& oracle = run.oracle;
    auto& st = run.st;
    auto& smw_cache = run.smw_cache;
    int* const serr = &st.p->smw_singular;     // lives in the control block: comes back with every chunk synchronisation
    auto& Xw = run.Xw;
    int& iters_host = run.iters_host; int& last_compression = run.last_compression;
    auto& all_shifts = run.all_shifts;
    bool& finished = run.finished;
    auto& used_real = run.used_real; auto& used_cplx = run.used_cplx;
    auto& npend = run.npend;
    auto& resid = run.resid;
    const int chunk_limit = std::max(1, std::min(std::min(run.chunk_limit, budget), 480));     // (< 512: ring of the norm history)
    auto check_used = [&]() { run.check_used(); };
    static const bool lazy_norm = true;
    (void)P; (void)opt_in; (void)serr; (void)lazy_norm;
    if (run.fast) {
        const int nstrip = adi_fast_nstrip(n), kst = adi_fast_kst(n);
        if (!run.fast_ready) {
            run.Gm = Mat(ctx, k * k, 2);
            run.nws = DevArr<double>(ctx, ADI_FAST_NWS);
            DRE_HIP(hipMemsetAsync(run.nws.p, 0, (ADI_FAST_NWS) * sizeof(double), ctx->stream));
            run.fast_ready = true;
        }
        Mat& Gm = run.Gm; auto& nws = run.nws;
        size_t& cyc = run.cyc;                // position in the cycle
        auto& fast_fe = run.fast_fe; auto& fast_pack = run.fast_pack;
        {
            const int base_it = iters_host;
            // one more iteration than the previous solve needed; the two flush launches deliver the decisions of the last two
            const int fast_chunk = cache->iters_hint > 0 ? std::max(opt.compression_interval, cache->iters_hint + 1) : run.chunk_limit;
            const int nit = std::min(std::min(std::min(std::max(1, fast_chunk), std::max(1, budget)), opt.maxiters - iters_host), 480);
            if (nit <= 0) { finished = true; resid->blocks[0].L = R; return; }
            Mat Rring(ctx, n, k * nit), Vall(ctx, n, k * nit);
            const size_t blocks_before = Xw->blocks.size();
            const size_t cyc_before = cyc;
            AdiFastArgs a;
            std::memset(&a, 0, sizeof(a));
            a.n = n; a.k = k; a.nstrip = nstrip; a.kst = kst; adi_fast_pick(n, k, &a.mode, &a.nt);
            a.T = Tm.p; a.ldt = Tm.ld; a.tdiag = tdiag ? 1 : 0; a.alpha = alpha_res; a.st = st.p; a.nws = nws.p;
            // the residual also in the B-operand lane order (dense.hpp, AdiFastArgs::Rpc): slot 0 = the chunk's input, slot j = after iteration j
            static const bool packed_r = true;
            const bool use_pk = packed_r && a.mode == 0;
            const size_t rpd = adi_fast_rpack_doubles(n, k);
            DevArr<double> Rpk(ctx, use_pk ? rpd * (size_t)(nit + 1) : 1);
            if (use_pk) adi_fast_pack_r(ctx, n, k, R.p, R.ld, Rpk.p, st.p);
            for (int j = 1; j <= nit; ++j) {
                const std::complex<double> mu = opt.shifts.values[cyc % opt.shifts.values.size()];
                all_shifts.push_back(mu);
                used_real.push_back(fast_fe[cyc % fast_fe.size()]);
                a.Apack = fast_pack[cyc % fast_pack.size()];
                if (j == 1) { a.Rcur = R.p; a.ldr = R.ld; } else { a.Rcur = Rring.p + (size_t)(j - 2) * k * Rring.ld; a.ldr = Rring.ld; }
                a.Rnext = Rring.p + (size_t)(j - 1) * k * Rring.ld; a.ldr_next = Rring.ld;
                a.Rpc = use_pk ? Rpk.p + (size_t)(j - 1) * rpd : nullptr;
                a.Rpn = use_pk ? Rpk.p + (size_t)j * rpd : nullptr;
                Mat Vj = Vall.colsview((j - 1) * k, k);
                a.V = Vj.p; a.ldv = Vj.ld;
                a.two_mu = 2.0 * mu.real();
                const int g = base_it + j;                                  // shifts consumed after this iteration
                // this launch also forms the Gram matrix of its INPUT residual (iteration g - 1) and decides on iteration g - 2
                a.G_prev = j >= 2 ? Gm.p + (size_t)((g - 1) & 1) * k * k : nullptr;
                a.G_prev2 = j >= 3 ? Gm.p + (size_t)((g - 2) & 1) * k * k : nullptr;
                a.it_prev2 = g - 2; a.do_strips = 1;
                adi_fast_iter(ctx, a);
                Xw->blocks.push_back({Vj, Tm, -2.0 * mu.real() * alpha_res, tdiag});
                ++cyc; ++iters_host;
            }
            {   // drain the norm pipeline: Gram matrix of the last residual, decisions for the last two iterations of the chunk
                const int g = base_it + nit;
                a.do_strips = 0; a.Apack = nullptr; a.Rnext = nullptr; a.V = nullptr;
                a.Rpc = use_pk ? Rpk.p + (size_t)nit * rpd : nullptr; a.Rpn = nullptr;
                a.Rcur = Rring.p + (size_t)(nit - 1) * k * Rring.ld; a.ldr = Rring.ld;
                a.G_prev = Gm.p + (size_t)(g & 1) * k * k;
                a.G_prev2 = nit >= 2 ? Gm.p + (size_t)((g - 1) & 1) * k * k : nullptr;
                a.it_prev2 = g - 1;
                adi_fast_iter(ctx, a);
                a.G_prev = nullptr;
                a.G_prev2 = Gm.p + (size_t)(g & 1) * k * k;
                a.it_prev2 = g;
                adi_fast_iter(ctx, a);
            }
            AdiState h;
            DRE_HIP(hipMemcpyAsync(&h, st.p, sizeof(AdiState), hipMemcpyDeviceToHost, ctx->stream));
            DRE_HIP(hipStreamSynchronize(ctx->stream));
            const int acc_it = std::min(std::max(h.iters - base_it, 0), nit);          // accepted iterations of this chunk
            for (int j = 1; j <= acc_it; ++j) { res.norms.push_back(h.norms[(base_it + j) & 511]); res.norm_iters.push_back(base_it + j); }
            Xw->blocks.resize(blocks_before + acc_it);
            check_used();
            if (acc_it > 0) R = Rring.colsview((acc_it - 1) * k, k);
            iters_host = base_it + acc_it;
            all_shifts.resize(iters_host);
            cyc = cyc_before + acc_it;
            last_compression += acc_it;
            res.iters = h.iters;
            res.res_norm = h.res_norm;
            if (h.done || acc_it < nit) finished = true;
            // (literal mode: the interval compression also runs after the LAST step, before the observer looks — adi.jl:111-119; it is the
            // compression adi.jl:78-80 would do anyway)
            if (opt.compression && last_compression >= opt.compression_interval && (!finished || cex)) {
                const long rk = Xw->rank();
                const bool defer = !cex && (n <= 512 ? rk <= 16L * n : (n <= ctx->compress_direct_max_n && rk <= 16L * n));
                if (!defer) { ldlt_compress(ctx, *Xw, ctf, cex); last_compression = 0; }
            }
        }
        resid->blocks[0].L = R;                 // the residual factor after the last accepted iteration
        return;
    }
    {
        std::vector<StepRec> recs;
        const size_t blocks_before = Xw->blocks.size();
        const int lc_before = last_compression;
        const Mat R_chunk_start = R;
        Mat hist_V, hist_R;                       // keep_history: the chunk's V_j and R_j side by side
        const int iters_chunk_start = iters_host;
        // the tolerance reltol ||C|| was not known when the solve began (AdiOptions::normC_dev): wait for it on this stream and take the decisions
        // of adi.jl:115-123 for everything recorded so far, in iteration order
        auto apply_tolerance = [&]() {
            if (!run.tol_event) return;
            DRE_HIP(hipStreamWaitEvent(ctx->stream, run.tol_event, 0));
            adi_decide_scan(ctx, st.p, 0, opt.normC_dev, run.reltol, -1.0);
            run.tol_event = nullptr;
        };
        auto resolve_deferred = [&]() {
            apply_tolerance();
            if (!run.defer) return;
            if (opt.normC_wait) opt.normC_wait();
            adi_decide_scan(ctx, st.p, iters_host, opt.normC_dev, run.reltol, opt.abstol);
            run.defer = false;
        };
        static const int chunk_timing = (env_trace("chunk") ? 1 : 0);
        const auto ct0 = std::chrono::steady_clock::now();
        int since_sync = 0, chunk_shifts = 0;
        const bool single_use = opt_in.shifts.kind != ShiftSpec::CYCLIC && !opt.inner_solve && cache->enabled;
        // Cyclic lists on the multifrontal path: the factorisations of the FIRST pass through the cycle (one workgroup per front: ~0.5 ms each at
        // n = 5177, 1.4 ms at n = 20209, a handful of CUs busy) also run ahead on the helper streams instead of one after the other in front of
        // their first use; from the second pass on every factor is cached and the look-ahead finds nothing to do.
        static const bool cyc_ahead_env = true;
        const bool sharded_la = ctx->comm && std::max(ctx->comm->nranks, ctx->comm->emulate) > 1;
        const bool cyc_ahead = cyc_ahead_env && opt_in.shifts.kind == ShiftSpec::CYCLIC && !opt.inner_solve && cache->enabled && n > ctx->dense_inv_max_n && !sharded_la;
        const bool lookahead = single_use || cyc_ahead;
        // the shift about to be used may have been factorised ahead on a helper stream: the main stream waits for that factorisation's event
        auto wait_prefetched = [&](std::complex<double> mu) {
            auto it = run.prefetch_ev.find({mu.real(), mu.imag()});
            if (it == run.prefetch_ev.end()) return;
            DRE_HIP(hipStreamWaitEvent(ctx->stream, it->second.ev, 0));
            run.ev_pool.push_back(it->second.ev);
            run.prefetch_ev.erase(it);
        };
        // factorise the next few shifts of the batch on the helper streams (one workgroup per front: a factorisation uses a handful of CUs
        // for ~200 us at n = 371 — 45 % of the kernel time of a default-ADI run when it sits on the main stream)
        auto prefetch_ahead = [&](std::complex<double> cur, bool fan_call = false) {
            static const int depth_env = -1;
            const int nh = depth_env >= 0 ? std::min(depth_env, 16) : (ctx->setup_streams >= 1 ? std::max(8, ctx->setup_streams) : 0);
            if (!lookahead || nh < 1) return;
            auto ups = oracle->peek((size_t)2 * nh + 3);
            // `cur` was just taken; when it opens a conjugate pair its partner is still the FIRST upcoming shift (the double step takes it later,
            // adi.jl:190) and must not be read as the start of a new pair — that shifted every later pair by one: the partners were factorised
            // ahead (never used, their slots never freed) and the shifts really needed were factorised inline (round 3 finding: 88 % of the
            // complex factorisations of a default-ADI run sat on the main stream).
            if (cur.imag() != 0.0 && !ups.empty() && ups[0] == std::conj(cur)) ups.erase(ups.begin());
            // slots whose shift is no longer ahead (never the case when the pairs are read correctly; cheap insurance against a leak)
            for (auto it = run.prefetch_ev.begin(); it != run.prefetch_ev.end();) {
                bool ahead = false;
                for (auto& u : ups) ahead = ahead || (u.real() == it->first.first && u.imag() == it->first.second);
                if (ahead) { ++it; continue; }
                run.ev_pool.push_back(it->second.ev);
                it = run.prefetch_ev.erase(it);
            }
            int scheduled = (int)run.prefetch_ev.size();
            static const bool trace_pf = env_trace("prefetch");
            if (trace_pf) {
                static long calls = 0, nups = 0, pend = 0;
                ++calls; nups += (long)ups.size(); pend += scheduled;
                if (calls % 500 == 0) std::fprintf(stderr, "[prefetch] %ld calls: %.2f upcoming shifts known per call, %.2f factorisations already in flight per call (depth %d)\n",
                                                   calls, (double)nups / calls, (double)pend / calls, nh);
            }
            auto helpers_up = [&]() {
                if (run.helpers_ready) return;
                while ((int)ctx->helpers.size() < nh) {
                    auto hc = std::make_unique<Ctx>();
                    hc->device = ctx->device; hc->num_cus = ctx->num_cus;
                    hc->stream = create_stream(2);
                    hc->timer = std::make_unique<KernelTimer>();
                    hipEvent_t ev;
                    DRE_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
                    ctx->helpers.push_back(std::move(hc)); ctx->helper_ev.push_back(ev);
                }
                if (!ctx->helper_e0) DRE_HIP(hipEventCreateWithFlags(&ctx->helper_e0, hipEventDisableTiming));
                DRE_HIP(hipEventRecord(ctx->helper_e0, ctx->stream));         // the operator's value arrays are ready here
                for (int h = 0; h < nh; ++h) {
                    Ctx* hc = ctx->helpers[(size_t)h].get();
                    hc->pivot_static = ctx->pivot_static; hc->pivot_growth_warn = ctx->pivot_growth_warn; hc->pivot_growth_fail = ctx->pivot_growth_fail;
                    hc->top_inverse_max_rows = ctx->top_inverse_max_rows; hc->dense_inv_max_n = ctx->dense_inv_max_n; hc->mf_subtree = ctx->mf_subtree;
                    hc->timer->enabled = ctx->timer && ctx->timer->enabled;
                    DRE_HIP(hipStreamWaitEvent(hc->stream, ctx->helper_e0, 0));
                }
                run.helpers_ready = true;
            };
            // Self-generated shift lists (Projection: ~13 upcoming shifts are known at every call): the factorisations that are not yet in flight
            // go out TOGETHER, one set of level launches per kind (real / complex) on one helper stream, instead of one chain of ~15 launches
            // per shift — the host spends ~250 us of every iteration of such a run on launch calls, and a single factorisation (~370 us at
            // n = 1357, one workgroup per front) leaves the device as idle as sixteen of them.
            if (single_use && !run.check_now && ctx->prefetch_batch > 0) {
                // (refilled only when few are left in flight: the window of known shifts advances by one per iteration, so filling it at every call
                // would again send the factorisations out one by one)
                if (scheduled > ctx->prefetch_batch) return;
                std::vector<std::complex<double>> tr, tc;
                const int room = MF_ZMAX - scheduled;
                for (size_t i = 0; i < ups.size() && (int)(tr.size() + tc.size()) < room; ++i) {
                    const std::complex<double> nx = ups[i];
                    const bool cx = nx.imag() != 0.0;
                    const auto ck = std::make_tuple(op.tag, nx.real(), nx.imag());
                    const bool known = cx ? cache->cplx_.count(ck) > 0 : cache->real.count(ck) > 0;
                    if (cx) ++i;
                    if (known || (nx == cur && !fan_call)) continue;
                    auto& lst = cx ? tc : tr;
                    bool dup = false;
                    for (auto& t : lst) dup = dup || t == nx;
                    if (!dup) lst.push_back(nx);
                }
                auto mark = [&](Ctx* hc, std::complex<double> nx) {
                    hipEvent_t ev;
                    if (!run.ev_pool.empty()) { ev = run.ev_pool.back(); run.ev_pool.pop_back(); }
                    else DRE_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
                    DRE_HIP(hipEventRecord(ev, hc->stream));
                    run.prefetch_ev[{nx.real(), nx.imag()}] = AdiRun::Prefetched{ev, 0};
                    ++scheduled;
                };
                if (tr.size() >= 2) {
                    helpers_up();
                    Ctx* hc = ctx->helpers[run.prefetch_rr++ % (size_t)nh].get();
                    std::vector<std::shared_ptr<FactorEntry<double>>> fes;
                    std::vector<Factor<double>*> fp;
                    std::vector<double> ce;
                    for (auto& t : tr) { fes.push_back(std::make_shared<FactorEntry<double>>()); fp.push_back(&fes.back()->f); ce.push_back(t.real()); }
                    mf_factor_batch<double>(hc, P, op.valFt.p, P.valEt.p, 1.0, ce.data(), fp.data(), (int)tr.size());
                    for (size_t z = 0; z < tr.size(); ++z) {
                        fes[z]->f.allow_topinv = cache->enabled;
                        const auto key = std::make_tuple(op.tag, tr[z].real(), 0.0);
                        cache->real[key] = fes[z]; cache->fresh.push_back(key); cache->nfactor++;
                        mark(hc, tr[z]);
                    }
                }
                if (tc.size() >= 2) {
                    helpers_up();
                    Ctx* hc = ctx->helpers[run.prefetch_rr++ % (size_t)nh].get();
                    std::vector<std::shared_ptr<FactorEntry<cplx>>> fes;
                    std::vector<Factor<cplx>*> fp;
                    std::vector<cplx> ce;
                    for (auto& t : tc) { fes.push_back(std::make_shared<FactorEntry<cplx>>()); fp.push_back(&fes.back()->f); ce.push_back(make_scalar<cplx>(t.real(), t.imag())); }
                    mf_factor_batch<cplx>(hc, P, op.valFt.p, P.valEt.p, make_scalar<cplx>(1.0, 0.0), ce.data(), fp.data(), (int)tc.size());
                    for (size_t z = 0; z < tc.size(); ++z) {
                        const auto key = std::make_tuple(op.tag, tc[z].real(), tc[z].imag());
                        cache->cplx_[key] = fes[z]; cache->fresh.push_back(key); cache->nfactor++;
                        mark(hc, tc[z]);
                    }
                }
            }
            for (size_t i = 0; i < ups.size() && scheduled < nh; ++i) {
                const std::complex<double> nx = ups[i];
                const bool cx = nx.imag() != 0.0;
                const auto ck = std::make_tuple(op.tag, nx.real(), nx.imag());
                const bool known = cx ? cache->cplx_.count(ck) > 0 : cache->real.count(ck) > 0;
                if (cx) ++i;                                   // the conjugate partner follows (adi.jl:190) and needs no factorisation of its own
                if (known) continue;
                // the shift the caller is about to use RIGHT NOW comes round again inside the horizon when the cycle is shorter than it: the
                // caller factorises it inline on its own stream — handing it to a helper here would put an unfinished factor into the cache that
                // the caller picks up without an event to wait for
                if (nx == cur && !fan_call) continue;
                helpers_up();
                Ctx* hc = ctx->helpers[run.prefetch_rr++ % (size_t)nh].get();
                hipEvent_t ev;
                if (!run.ev_pool.empty()) { ev = run.ev_pool.back(); run.ev_pool.pop_back(); }
                else DRE_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
                // (Issuing these factorisations from a host thread of their own was tried — the loop spends ~250 us of host time per iteration on
                // launch calls: no gain, 3 230 against 3 250 it/s at n = 1357 Ros2; launches from two threads serialise inside the HIP runtime.)
                if (cx) (void)get_factor<cplx>(hc, op, cache, cache->cplx_, nx, false, nullptr, run.check_now);
                else {
                    auto fnew = get_factor<double>(hc, op, cache, cache->real, nx, false, nullptr, run.check_now);
                    // a Cyclic factor will be used in every time step: its dense top-of-tree inverse is built right behind the factorisation, on
                    // the helper stream (mf_solve would build it inside the third solve that uses the factor, on the critical path)
                    static const bool top_ahead = true;
                    if (cyc_ahead && top_ahead) mf_prepare_topinv(hc, *op.P, fnew->f);
                }
                DRE_HIP(hipEventRecord(ev, hc->stream));
                const long ticket = 0;
                run.prefetch_ev[{nx.real(), nx.imag()}] = AdiRun::Prefetched{ev, ticket};
                ++scheduled;
            }
        };
        // Fan groups (round 4: batched, and sharded over the ranks of a communicator BY SHIFT — north_star's "independent ADI shifts farmed across
        // the GPUs"): the g solves of a group are independent, so rank r takes the group positions s = r (mod P), i.e. only ever factorises and
        // keeps the shifts it owns (the factor farm of SURVEY 8e-4 without shipping factors), writes its W_s into slab r of the gathered panel and
        // ONE in-place all-gather per GROUP (n g k doubles in all; round 3: one per iteration, columns sharded) completes it on every rank;
        // mixing, norms, decisions, compression and K(t) are replicated and bit-identical on all ranks.
        const int fan_P = ctx->comm ? std::max(1, std::max(ctx->comm->nranks, ctx->comm->emulate)) : 1;
        const int fan_max = (opt_in.shifts.kind == ShiftSpec::CYCLIC && !opt.inner_solve && cache->enabled && n > ctx->dense_inv_max_n)
                                ? std::min(ctx->adi_fan, FAN_GMAX) : 0;
        while (iters_host < opt.maxiters) {
            // ---- fan group: the next g real shifts of the cycle at once (independent solves that share every launch) -------------------------
            if (fan_max >= 2 && !run.fan_off) {
                // (a group may cross the chunk limit — 3, 3, 3, 3 instead of 3, 3, 3, 1 iterations per chunk of 10 — unless the caller steps
                // with a budget, the literal mode compresses at exact intervals or the chunk length is the previous solve's iteration count)
                const bool strict = cex || budget < (1 << 29) || run.chunk_from_hint;
                const int room = std::min(std::min(fan_max, opt.maxiters - iters_host), strict ? chunk_limit - chunk_shifts : fan_max);
                const int gmin = opt.keep_history ? 1 : 2;          // with a history even a single iteration takes this path (its R_j stays in the slab)
                const auto ups = room >= gmin ? oracle->peek((size_t)room) : std::vector<std::complex<double>>();
                int g = 0;
                double mus[FAN_GMAX];
                while (g < (int)ups.size() && g < room && ups[(size_t)g].imag() == 0.0) {
                    bool dup = false;
                    for (int i = 0; i < g; ++i) dup = dup || mus[i] == ups[(size_t)g].real();
                    if (dup) break;
                    mus[g] = ups[(size_t)g].real(); ++g;
                }
                FanCoef co;
                while (g >= 2 && fan_coefficients(mus, g, &co) > ctx->adi_fan_max_coef) --g;
                if (g == 1) (void)fan_coefficients(mus, 1, &co);
                if (g < gmin) g = 0;
                // which ranks this process plays: its own, or all of them one after the other (shard_emulate)
                const bool emu = ctx->comm && ctx->comm->emulate > 1;
                const int my_rank = ctx->comm ? ctx->comm->rank : 0;
                auto mine = [&](int s_) { return fan_P == 1 || emu || (s_ % fan_P) == my_rank; };
                std::vector<std::shared_ptr<FactorEntry<double>>> fes((size_t)std::max(g, 0));
                if (lookahead && g >= gmin) prefetch_ahead(std::complex<double>(0.0, 0.0), true);      // (first pass through the cycle: this group's and the next groups' factors; all waited for below)
                for (int s_ = 0; s_ < g && g >= gmin; ++s_) {
                    if (!mine(s_)) continue;
                    if (lookahead) wait_prefetched(std::complex<double>(mus[s_], 0.0));
                    auto fe = get_factor<double>(ctx, op, cache, cache->real, std::complex<double>(mus[s_], 0.0), true, nullptr, !lookahead || run.check_now);
                    if (fe->dense) { g = 0; break; }              // the dense-inverse step has its own fused kernels
                    fes[(size_t)s_] = fe;
                }
                if (g >= gmin) {
                    static const bool fht = env_trace("fan");
                    static double ft[6] = {0, 0, 0, 0, 0, 0}; static long fn_ = 0;
                    auto fnow = []() { return std::chrono::steady_clock::now(); };
                    auto fus = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
                    const auto f0 = fnow();
                    const AdiState* dst = st.p;
                    dense_norm_flush(ctx, k, Tm, tdiag, alpha_res, st.p, &npend);
                    const int gp = ceil_div(g, fan_P);                 // group positions per rank; W_s lives in slot (s mod P) gp + s div P
                    FanSlots slots; std::memset(&slots, 0, sizeof(slots));
                    for (int s_ = 0; s_ < g; ++s_) slots.s[s_] = (s_ % fan_P) * gp + s_ / fan_P;
                    Mat Wcat(ctx, n, fan_P * gp * k);
                    bool ok = true;
                    for (int r = 0; r < fan_P && ok; ++r) {
                        if (!(fan_P == 1 || emu || r == my_rank)) continue;
                        std::vector<int> pos;                          // this rank's group positions
                        for (int s_ = r; s_ < g; s_ += fan_P) pos.push_back(s_);
                        const int gr = (int)pos.size();
                        if (gr == 0) continue;
                        SmwZ sz; std::memset(&sz, 0, sizeof(sz));
                        if (op.has_lr) {
                            // SMW products of this rank's shifts that this solve has not formed yet: W_U = M_s^-1 Vt for all of them in one batched
                            // solve, the capacitance matrices inverted in one launch (smw.jl:19-28); at the first group of a solve (one rank) every
                            // other shift of the cycle whose factor exists rides along, so that a time step pays this chain once
                            std::vector<std::pair<double, std::shared_ptr<FactorEntry<double>>>> need;
                            for (int s_ : pos) if (!smw_cache.count({mus[s_], 0.0})) need.push_back({mus[s_], fes[(size_t)s_]});
                            if (!need.empty() && !run.fan_smw_all && fan_P == 1) {
                                run.fan_smw_all = true;
                                for (auto& mv : opt.shifts.values) {
                                    if (mv.imag() != 0.0 || (int)need.size() >= MF_ZMAX) continue;
                                    bool dup = smw_cache.count({mv.real(), 0.0}) > 0;
                                    for (auto& nd : need) dup = dup || nd.first == mv.real();
                                    if (dup) continue;
                                    auto itf = cache->real.find(std::make_tuple(op.tag, mv.real(), 0.0));
                                    if (itf == cache->real.end() || itf->second->dense || run.prefetch_ev.count({mv.real(), 0.0})) continue;
                                    need.push_back({mv.real(), itf->second});
                                }
                            }
                            if (!need.empty()) {
                                const int np = (int)need.size();
                                const Factor<double>* Fp[MF_ZMAX];
                                for (int z = 0; z < np; ++z) Fp[z] = &need[(size_t)z].second->f;
                                Mat WUcat(ctx, n, np * m);
                                ok = mf_solve_batch(ctx, P, Fp, np, op.Vt.p, op.Vt.ld, m, WUcat.p, WUcat.ld, m, dst);
                                if (!ok) break;
                                Mat smu(ctx, m, np * m);
                                gemm(ctx, true, false, 1.0, op.U, WUcat, 0.0, smu, dst, "smw_small");
                                SinvZ iz; std::memset(&iz, 0, sizeof(iz));
                                for (int z = 0; z < np; ++z) {
                                    SmwCacheEntry en;
                                    en.keep = WUcat.buf; en.WU = WUcat.p + (size_t)z * m * WUcat.ld; en.ldwu = WUcat.ld;
                                    en.sinv = std::make_shared<Buf>(ctx, (size_t)m * m * sizeof(double));
                                    iz.out[z] = (double*)en.sinv->p;
                                    smw_cache.emplace(std::make_pair(need[(size_t)z].first, 0.0), en);
                                    used_real.push_back(need[(size_t)z].second);
                                }
                                hipLaunchKernelGGL(k_sinv_z, dim3(np), dim3(64), 0, ctx->stream, m, (const double*)smu.p, smu.ld, op.alpha, iz, dst, serr);
                            }
                            for (int z = 0; z < gr; ++z) {
                                const auto& en = smw_cache.find({mus[pos[(size_t)z]], 0.0})->second;
                                sz.WU[z] = (const double*)en.WU; sz.ldwu[z] = en.ldwu; sz.Sinv[z] = (const double*)en.sinv->p;
                            }
                        }
                        const Factor<double>* Fs[MF_ZMAX];
                        for (int z = 0; z < gr; ++z) Fs[z] = &fes[(size_t)pos[(size_t)z]]->f;
                        Mat Wr = Wcat.colsview(r * gp * k, gr * k);          // slab r: its solves side by side
                        const auto f1 = fnow();
                        ok = mf_solve_batch(ctx, P, Fs, gr, R.p, R.ld, k, Wr.p, Wr.ld, k, dst);
                        if (fht) { ft[0] += fus(f0, f1); ft[1] += fus(f1, fnow()); }
                        if (!ok) break;
                        for (int s_ : pos) used_real.push_back(fes[(size_t)s_]);
                        if (op.has_lr) {
                            // W_s <- W_s - W_U,s (S_s^-1 (U' W_s))  for the rank's solves: one product U' [W_s ...], one apply launch (in place)
                            Mat sm(ctx, m, gr * k);
                            gemm(ctx, true, false, 1.0, op.U, Wr, 0.0, sm, dst, "smw_small");
                            TimedScope ts(ctx, "smw_apply", 8.0 * n * gr * (2.0 * k + m), 2.0 * n * gr * (double)k * m, gr);
                            hipLaunchKernelGGL(k_smw_apply_z, dim3(ceil_div(n, 256), ceil_div(k, SMW_CB), gr), dim3(256), 0, ctx->stream, n, m, k, Wr.p, Wr.ld, sz,
                                               (const double*)sm.p, sm.ld, dst);
                        }
                    }
                    // (nothing was enqueued where the batched form does not apply — the same decision on every rank: it depends on the pencil and
                    // the context's options only — and this solve goes on one iteration at a time)
                    if (!ok) {
                        // (with real ranks the decision must not differ between them — it can only where a rank's own factors needed static pivots:
                        // fail loudly instead of leaving the other ranks inside a collective)
                        if (fan_P > 1 && !emu) throw Error(ERR_INTERNAL, "sharded fan group: the batched solves do not apply to this rank's factors (static pivots); "
                                                                         "run with adi_fan = 0 for this pencil");
                        run.fan_off = true; continue;
                    }
                    if (fan_P > 1 && !emu) {
                        TimedScope ts(ctx, "comm_allgather_w", 8.0 * n * (double)fan_P * gp * k, 0.0);
                        comm_allgather_inplace(ctx, *ctx->comm, Wcat.p, (size_t)n * gp * k);
                    }
                    const auto f2 = fnow();
                    // V_j = sum_s c_js W_s,  R_j = R_0 - sum_s d_js E' W_s  for the g iterations: one pass over E' and the panels
                    Mat Vcat, Rcat;
                    if (opt.keep_history) {
                        // every iteration of the chunk side by side (the Rosenbrock-1 driver reads E'V_j = (R_{j-1} - R_j) / (2 mu_j) off these slabs)
                        const int cap = chunk_limit + FAN_GMAX;
                        if (hist_V.empty()) { hist_V = Mat(ctx, n, cap * k); hist_R = Mat(ctx, n, cap * k); }
                        if (chunk_shifts + g <= cap) { Vcat = hist_V.colsview(chunk_shifts * k, g * k); Rcat = hist_R.colsview(chunk_shifts * k, g * k); }
                        else run.hist_ok = false;
                    }
                    if (Vcat.p == nullptr) { Vcat = Mat(ctx, n, g * k); Rcat = Mat(ctx, n, g * k); }
                    fan_spmm_mix(ctx, P, Wcat, R, Vcat, Rcat, g, k, co, slots, dst);
                    const auto f3 = fnow();
                    for (int j = 0; j < g; ++j) {
                        const std::complex<double> muj = oracle->take(&res.warnings);
                        all_shifts.push_back(muj);
                        Mat Vj = Vcat.colsview(j * k, k), Rj = Rcat.colsview(j * k, k);
                        Xw->blocks.push_back({Vj, Tm, -2.0 * muj.real() * alpha_res, tdiag});
                        iters_host += 1; last_compression += 1;
                        oracle->update(Rj, {Vj});
                        recs.push_back({iters_host, Xw->blocks.size(), 1, Rj});
                        ++since_sync; ++chunk_shifts;
                    }
                    const auto f4 = fnow();
                    apply_tolerance();
                    residual_norm_group_diag(ctx, Rcat, g, k, Tm, tdiag, alpha_res, st.p, iters_host - g);
                    R = Rcat.colsview((g - 1) * k, k);
                    if (fht) {
                        const auto f5 = fnow();
                        ft[2] += fus(f2, f3); ft[3] += fus(f3, f4); ft[4] += fus(f4, f5); ft[5] += fus(f0, f5);
                        if (++fn_ % 128 == 0) std::fprintf(stderr, "[fan host, us per group] before the solves %.1f | batched solves %.1f | mix %.1f | bookkeeping %.1f | norms %.1f | total %.1f\n",
                                                           ft[0] / fn_, ft[1] / fn_, ft[2] / fn_, ft[3] / fn_, ft[4] / fn_, ft[5] / fn_);
                    }
                    if (opt.compression && chunk_shifts >= chunk_limit) break;
                    if (!opt.compression && since_sync >= std::min(10, chunk_limit)) break;
                    continue;
                }
            }
            apply_tolerance();
            if (run.defer) {
                // an iteration outside the fan path while the tolerance is still on its way: close the chunk first (its end resolves the tolerance)
                if (!recs.empty()) break;
                resolve_deferred();
            }
            run.hist_ok = false;                  // (its residual factor is updated in place: no history of this solve)
            std::complex<double> mu = oracle->take(&res.warnings);
            all_shifts.push_back(mu);
            const bool is_real = (mu.imag() == 0.0);
            const AdiState* dst = st.p;
            if (lookahead) { wait_prefetched(mu); prefetch_ahead(mu); }
            Mat V1, V2;
            bool norm_done = false, rode = false;
            if (is_real) {
                // dense inverses pay off only for shift lists that persist across Lyapunov solves (user-given Cyclic values)
                const bool user_inner = opt.inner_solve != nullptr;
                std::shared_ptr<FactorEntry<double>> fe;
                if (!user_inner) {
                    if (single_use && env_trace("prefetch")) {
                        static long hit = 0, miss = 0;
                        (cache->real.count(std::make_tuple(op.tag, mu.real(), 0.0)) ? hit : miss)++;
                        if ((hit + miss) % 500 == 0) std::fprintf(stderr, "[prefetch] real shifts: %ld found ready, %ld factorised inline\n", hit, miss);
                    }
                    fe = get_factor<double>(ctx, op, cache, cache->real, mu, opt_in.shifts.kind == ShiftSpec::CYCLIC, nullptr, !lookahead || run.check_now);
                    used_real.push_back(fe);
                }
                auto key = std::make_pair(mu.real(), 0.0);
                auto sc = smw_cache.find(key);
                const bool have = op.has_lr && sc != smw_cache.end();
                const int ncols = k + ((op.has_lr && !have) ? m : 0);
                if (fe && fe->dense) {
                    // dense-inverse step: one stacked GEMM + one fused apply (V and the residual recurrence)
                    const int mm = op.has_lr ? m : 0;
                    if (fe->stack.empty() || fe->stack_U != (const void*)op.U.p || fe->stack_m != mm) {
                        Mat stk(ctx, 2 * n + mm, n);
                        { Mat top = stk.view(0, 0, n, n); copy_mat(ctx, fe->dinv, top); }
                        { Mat mid = stk.view(n, 0, n, n); spmm(ctx, P, P.valEt.p, fe->dinv, mid, 1.0, 0.0, nullptr); }
                        if (mm) { Mat bot = stk.view(2 * n, 0, mm, n); gemm(ctx, true, false, 1.0, op.U, fe->dinv, 0.0, bot, nullptr, "gemm_dinv"); }
                        fe->stack = stk; fe->stack_U = (const void*)op.U.p; fe->stack_m = mm;
                    }
                    const int lds_ = 2 * n + mm;
                    V1 = Mat(ctx, n, k);
                    if (op.has_lr && !have) {
                        Mat WK(ctx, lds_, m);
                        gemm(ctx, false, false, 1.0, fe->stack, op.Vt, 0.0, WK, dst, "gemm_dinv");
                        SmwCacheEntry en;
                        Mat WKS(ctx, 2 * n, m);
                        en.keep = WKS.buf; en.WU = WKS.p; en.ldwu = WKS.ld;
                        en.sinv = std::make_shared<Buf>(ctx, (size_t)m * m * sizeof(double));
                        hipLaunchKernelGGL((k_sinv<double>), dim3(1), dim3(64), 0, ctx->stream, m, WK.p + 2 * (size_t)n, WK.ld, op.alpha, (double*)en.sinv->p, dst, serr);
                        hipLaunchKernelGGL(k_fold_sinv, dim3(ceil_div(2 * n * m, 256)), dim3(256), 0, ctx->stream, 2 * n, m, WK.p, WK.ld,
                                           (const double*)en.sinv->p, WKS.p, WKS.ld, dst);
                        sc = smw_cache.emplace(key, en).first;
                    }
                    if (k <= 96) {
                        // split-K slabs of the stacked GEMM are consumed directly by the fused step kernel
                        int zs = 1;
                        BufP wpart = gemm_partials(ctx, false, false, lds_, k, n, fe->stack.p, fe->stack.ld, R.p, R.ld, &zs, dst, "gemm_dinv");
                        dense_adi_step(ctx, n, mm, k, zs, (const double*)wpart->p, op.has_lr ? (const double*)sc->second.WU : nullptr,
                                       op.has_lr ? sc->second.ldwu : 0, V1, R, 2.0 * mu.real(), Tm, tdiag, alpha_res, st.p, iters_host + 1,
                                       lazy_norm ? &npend : nullptr);
                        norm_done = true; rode = true;
                    } else {
                        Mat Wst(ctx, lds_, k);
                        gemm(ctx, false, false, 1.0, fe->stack, R, 0.0, Wst, dst, "gemm_dinv");
                        TimedScope ts(ctx, "dense_apply", 8.0 * n * (4.0 * k + 2.0 * m), 4.0 * n * k * m);
                        if (op.has_lr)
                            hipLaunchKernelGGL((k_dense_apply<true>), dim3(ceil_div(n, 64), ceil_div(k, 4)), dim3(256), 0, ctx->stream,
                                               n, m, k, Wst.p, Wst.ld, (const double*)sc->second.WU, sc->second.ldwu,
                                               V1.p, V1.ld, R.p, R.ld, 2.0 * mu.real(), dst);
                        else
                            hipLaunchKernelGGL((k_dense_apply<false>), dim3(ceil_div(n, 64), ceil_div(k, 4)), dim3(256), 0, ctx->stream,
                                               n, 0, k, Wst.p, Wst.ld, (const double*)nullptr, 0,
                                               V1.p, V1.ld, R.p, R.ld, 2.0 * mu.real(), dst);
                    }
                } else if (!user_inner && ctx->comm && std::max(ctx->comm->nranks, ctx->comm->emulate) > 1 && k >= ctx->shard_min_cols) {
                    // ---- column-sharded step: V = (F' + mu E')^-1 R acts column by column (adi.jl:158-159), so rank g
                    // solves its 16-column tiles of R only (multifrontal sweeps + SMW, both column local), writes them into its block of the
                    // gathered panel and ONE in-place all-gather (RCCL, on this stream) completes V on every rank.  The m columns M^-1 Vt of
                    // the SMW correction are computed by every rank that owns columns (replicated: m = 7 against k/P).
                    Comm& cm = *ctx->comm;
                    const int Pn = std::max(cm.nranks, cm.emulate);
                    const ColBlocks cb(k, Pn);
                    Mat Vg(ctx, n, cb.padded());
                    auto local = [&](int g) {
                        const int c0 = cb.c0(g), kc = cb.c1(g) - c0;
                        if (kc <= 0) return;
                        auto scl = smw_cache.find(key);
                        const bool havel = op.has_lr && scl != smw_cache.end();
                        const int extra = (op.has_lr && !havel) ? m : 0;
                        Mat Vloc = Vg.colsview(c0, kc);
                        if (!op.has_lr) {
                            mf_solve_from(ctx, P, fe->f, R.p + (size_t)c0 * R.ld, R.ld, kc, Vloc.p, Vloc.ld, kc, dst);
                            return;
                        }
                        Mat W(ctx, n, kc + extra);
                        if (extra) { Mat d = W.colsview(kc, m); copy_mat(ctx, op.Vt, d, 1.0, dst); }
                        mf_solve_from(ctx, P, fe->f, R.p + (size_t)c0 * R.ld, R.ld, kc, W.p, W.ld, kc + extra, dst);
                        Mat small(ctx, m, kc + extra);
                        gemm(ctx, true, false, 1.0, op.U, W, 0.0, small, dst, "smw_small");
                        if (!havel) {
                            SmwCacheEntry en;
                            en.keep = W.buf; en.WU = W.p + (size_t)kc * W.ld; en.ldwu = W.ld;
                            en.sinv = std::make_shared<Buf>(ctx, (size_t)m * m * sizeof(double));
                            hipLaunchKernelGGL((k_sinv<double>), dim3(1), dim3(64), 0, ctx->stream, m, small.p + (size_t)kc * small.ld, small.ld, op.alpha, (double*)en.sinv->p, dst, serr);
                            scl = smw_cache.emplace(key, en).first;
                        }
                        TimedScope ts(ctx, "smw_apply", 8.0 * n * (2.0 * kc + m), 2.0 * n * kc * m);
                        hipLaunchKernelGGL((k_smw_apply<double, true>), dim3(ceil_div(n, 256), ceil_div(kc, SMW_CB)), dim3(256), 0, ctx->stream,
                                           n, m, kc, W.p, W.ld, (const double*)scl->second.WU, scl->second.ldwu, (const double*)scl->second.sinv->p,
                                           small.p, small.ld, Vloc.p, Vloc.ld, (double*)nullptr, 0, 0.0, dst);
                    };
                    if (cm.emulate > 1) { for (int g = 0; g < Pn; ++g) local(g); }
                    else {
                        local(cm.rank);
                        TimedScope ts(ctx, "comm_allgather_v", 8.0 * n * (double)cb.padded(), 0.0);
                        comm_allgather_inplace(ctx, cm, Vg.p, (size_t)n * cb.width());
                    }
                    V1 = Vg.colsview(0, k);
                    spmm(ctx, P, P.valEt.p, V1, R, -2.0 * mu.real(), 1.0, dst);          // R <- R - 2 mu E' V   (adi.jl:171), replicated
                } else {
                    Mat W(ctx, n, ncols);
                    if (user_inner) { Mat d = W.colsview(0, k); copy_mat(ctx, R, d, 1.0, dst); }
                    if (op.has_lr && !have) { Mat d = W.colsview(k, m); copy_mat(ctx, op.Vt, d, 1.0, dst); }
                    if (user_inner) user_block_solve(ctx, op, opt, mu, W, nullptr);
                    else mf_solve_from(ctx, P, fe->f, R.p, R.ld, k, W.p, W.ld, ncols, dst);     // the residual block is read where it is
                    if (op.has_lr) {
                        V1 = Mat(ctx, n, k);
                        // small = U' W: a skinny split-K MFMA GEMM (the one-workgroup-per-column kernel is latency bound at large n).  (Letting the
                        // SMW kernels sum the slabs themselves saves the reduction launch and is neutral at n = 5177, but every one of the 1264
                        // workgroups of the apply kernel then sums its 56 entries over 80 slabs at n = 20209: 59 us against 20 + 10 — reverted.)
                        Mat small(ctx, m, ncols);
                        gemm(ctx, true, false, 1.0, op.U, W, 0.0, small, dst, "smw_small");
                        if (!have) {
                            SmwCacheEntry en;
                            en.keep = W.buf; en.WU = W.p + (size_t)k * W.ld; en.ldwu = W.ld;
                            en.sinv = std::make_shared<Buf>(ctx, (size_t)m * m * sizeof(double));
                            hipLaunchKernelGGL((k_sinv<double>), dim3(1), dim3(64), 0, ctx->stream, m, small.p + (size_t)k * small.ld, small.ld, op.alpha, (double*)en.sinv->p, dst, serr);
                            sc = smw_cache.emplace(key, en).first;
                        }
                        TimedScope ts(ctx, "smw_apply", 8.0 * n * (2.0 * k + m), 2.0 * n * k * m);
                        hipLaunchKernelGGL((k_smw_apply<double, true>), dim3(ceil_div(n, 256), ceil_div(k, SMW_CB)), dim3(256), 0, ctx->stream,
                                           n, m, k, W.p, W.ld, (const double*)sc->second.WU, sc->second.ldwu, (const double*)sc->second.sinv->p,
                                           small.p, small.ld, V1.p, V1.ld, (double*)nullptr, 0, 0.0, dst);
                    } else {
                        V1 = W;
                    }
                    // R <- R - 2 mu E' V   (adi.jl:171)
                    spmm(ctx, P, P.valEt.p, V1, R, -2.0 * mu.real(), 1.0, dst);
                }
                Xw->blocks.push_back({V1, Tm, -2.0 * mu.real() * alpha_res, tdiag});
                iters_host += 1; last_compression += 1;
                oracle->update(R, {V1});
            } else {
                std::complex<double> mu2 = oracle->take(&res.warnings);
                all_shifts.push_back(mu2);
                DRE_REQUIRE(std::abs(mu2 - std::conj(mu)) <= 1e-8 * std::abs(mu), "complex shifts must come in conjugate pairs (adi.jl:190)");
                V1 = Mat(ctx, n, k);
                V2 = Mat(ctx, n, k);
                const bool user_inner = opt.inner_solve != nullptr;
                std::shared_ptr<FactorEntry<cplx>> fe;
                if (!user_inner) {
                    if (single_use && env_trace("prefetch")) {
                        static long hit = 0, miss = 0;
                        (cache->cplx_.count(std::make_tuple(op.tag, mu.real(), mu.imag())) ? hit : miss)++;
                        if ((hit + miss) % 500 == 0) std::fprintf(stderr, "[prefetch] complex pairs: %ld found ready, %ld factorised inline\n", hit, miss);
                    }
                    fe = get_factor<cplx>(ctx, op, cache, cache->cplx_, mu, true, nullptr, !lookahead || run.check_now);
                    used_cplx.push_back(fe);
                }
                auto key = std::make_pair(mu.real(), mu.imag());
                auto sc = smw_cache.find(key);
                const bool have = op.has_lr && sc != smw_cache.end();
                const int ncols = k + ((op.has_lr && !have) ? m : 0);
                auto Wb = std::make_shared<Buf>(ctx, (size_t)n * ncols * sizeof(cplx));
                cplx* W = (cplx*)Wb->p;
                {
                    size_t tot = (size_t)n * k;
                    hipLaunchKernelGGL(k_real_to_cplx, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, ctx->stream, n, k, R.p, R.ld, W, n, dst);
                    if (op.has_lr && !have) {
                        tot = (size_t)n * m;
                        hipLaunchKernelGGL(k_real_to_cplx, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, ctx->stream, n, m, op.Vt.p, op.Vt.ld, W + (size_t)k * n, n, dst);
                    }
                }
                if (user_inner) {
                    // the real right-hand side [R, Vt] goes to the user's solver; the complex solution comes back interleaved
                    Mat Wr(ctx, n, ncols);
                    { Mat d = Wr.colsview(0, k); copy_mat(ctx, R, d, 1.0, dst); }
                    if (op.has_lr && !have) { Mat d = Wr.colsview(k, m); copy_mat(ctx, op.Vt, d, 1.0, dst); }
                    user_block_solve(ctx, op, opt, mu, Wr, W);
                } else mf_solve<cplx>(ctx, P, fe->f, W, n, ncols, dst);
                const double delta = mu.real() / mu.imag();
                if (op.has_lr) {
                    auto sb = std::make_shared<Buf>(ctx, (size_t)m * ncols * sizeof(cplx));
                    cplx* small = (cplx*)sb->p;
                    hipLaunchKernelGGL((k_smw_small<cplx>), dim3(ncols), dim3(256), 0, ctx->stream, n, m, op.U.p, op.U.ld, W, n, small, m, dst);
                    if (!have) {
                        SmwCacheEntry en;
                        en.keep = Wb; en.WU = W + (size_t)k * n; en.ldwu = n;
                        en.sinv = std::make_shared<Buf>(ctx, (size_t)m * m * sizeof(cplx));
                        hipLaunchKernelGGL((k_sinv<cplx>), dim3(1), dim3(64), 0, ctx->stream, m, small + (size_t)k * m, m, op.alpha, (cplx*)en.sinv->p, dst, serr);
                        sc = smw_cache.emplace(key, en).first;
                    }
                    hipLaunchKernelGGL((k_smw_apply<cplx, true>), dim3(ceil_div(n, 256), ceil_div(k, SMW_CB)), dim3(256), 0, ctx->stream,
                                       n, m, k, W, n, (const cplx*)sc->second.WU, sc->second.ldwu, (const cplx*)sc->second.sinv->p,
                                       small, m, V1.p, V1.ld, V2.p, V2.ld, delta, dst);
                } else {
                    hipLaunchKernelGGL((k_smw_apply<cplx, false>), dim3(ceil_div(n, 256), ceil_div(k, SMW_CB)), dim3(256), 0, ctx->stream,
                                       n, 0, k, W, n, (const cplx*)nullptr, 0, (const cplx*)nullptr, (const cplx*)nullptr, 0,
                                       V1.p, V1.ld, V2.p, V2.ld, delta, dst);
                }
                // R <- R - 2 sqrt2 Re(mu) E' V1   (adi.jl:217)
                spmm(ctx, P, P.valEt.p, V1, R, -2.0 * 1.4142135623730951 * mu.real(), 1.0, dst);
                Xw->blocks.push_back({V1, Tm, -2.0 * mu.real() * alpha_res, tdiag});
                Xw->blocks.push_back({V2, Tm, -2.0 * mu.real() * alpha_res, tdiag});
                iters_host += 2; last_compression += 2;
                oracle->update(R, {V1, V2});
            }
            // residual norm through the Gram matrix, convergence decision on the device
            if (!rode) dense_norm_flush(ctx, k, Tm, tdiag, alpha_res, st.p, &npend);     // a step of another kind: norms stay in order
            if (!norm_done) residual_norm_step(ctx, R, Tm, tdiag, alpha_res, st.p, iters_host);
            recs.push_back({iters_host, Xw->blocks.size(), is_real ? 1 : 2, R});
            ++since_sync; chunk_shifts += is_real ? 1 : 2;
            if (opt.compression && chunk_shifts >= chunk_limit) break;
            if (!opt.compression && since_sync >= std::min(10, chunk_limit)) break;
        }
        // synchronise once per chunk and find out how far the device really got
        dense_norm_flush(ctx, k, Tm, tdiag, alpha_res, st.p, &npend);
        const bool was_deferred = run.defer;
        resolve_deferred();
        AdiState h;
        const auto ct1 = std::chrono::steady_clock::now();
        ctx_fetch(ctx, st.p, sizeof(AdiState), &h);
        if (was_deferred || run.abstol_pending) {
            run.abstol = h.abstol; res.abstol = h.abstol; res.initial_norm = h.norms[0];
            if (!res.norms.empty()) res.norms[0] = h.norms[0];
            run.abstol_pending = false;
        }
        if (chunk_timing) {
            static double enq = 0.0, wait = 0.0; static long nch = 0, nit_ = 0;
            const auto ct2 = std::chrono::steady_clock::now();
            enq += std::chrono::duration<double, std::micro>(ct1 - ct0).count(); wait += std::chrono::duration<double, std::micro>(ct2 - ct1).count();
            nit_ += (long)recs.size();
            if (++nch % 64 == 0) std::fprintf(stderr, "[chunk timing] %ld chunks, %ld iterations: host enqueue %.1f us / iteration, wait at the synchronisation %.1f us / iteration\n",
                                              nch, nit_, enq / nit_, wait / nit_);
        }
        size_t nblocks = blocks_before;
        int lc = lc_before;
        R = R_chunk_start;
        for (auto& r : recs) {
            if (r.iters_after <= h.iters) {
                nblocks = r.nblocks; lc += r.nshifts;
                R = r.Rafter;                                  // (fan groups leave every residual in a buffer of its own)
                res.norms.push_back(h.norms[r.iters_after & 511]);
                res.norm_iters.push_back(r.iters_after);
            }
        }
        Xw->blocks.resize(nblocks);
        check_used();
        last_compression = lc;
        res.iters = h.iters;
        res.res_norm = h.res_norm;
        if (opt.keep_history && run.hist_ok) {
            const int acc = std::max(0, std::min(h.iters, iters_host) - iters_chunk_start);        // accepted iterations of this chunk
            if (acc > 0 && !hist_V.empty()) {
                AdiHistChunk hc;
                hc.R0 = R_chunk_start; hc.Rs = hist_R.colsview(0, acc * k); hc.Vs = hist_V.colsview(0, acc * k);
                for (int j = 0; j < acc; ++j) hc.mu.push_back(all_shifts[(size_t)iters_chunk_start + j].real());
                res.hist.push_back(std::move(hc));
            } else if (acc > 0) run.hist_ok = false;
        }
        if (h.smw_singular) throw Error(ERR_SINGULAR, "SMW: capacitance matrix is singular");
        if (h.done || recs.empty()) finished = true;
        if (opt.compression && last_compression >= opt.compression_interval && (!finished || cex)) {
            // Small n: the compression works on the n x n matrix L D L' whatever the number of columns, and the increments
            // never depend on X, so the intermediate compressions of adi.jl:72-76 are deferred to the final one
            // (adi.jl:78-80) as long as the uncompressed factor stays small.
            // The same holds for the direct form up to compress_direct_max_n (one GEMM over all columns) and for the factor form
            // (panel steps ~ rank, GEMM traffic ~ columns: one late compression costs the GEMMs of two early ones and half the
            // panels) while the factor still fits the factor-form limit c + 64 <= n after the next chunk.
            static const bool defer_on = true;
            const long rk = Xw->rank(), next = (long)std::max(opt.compression_interval * 2, run.chunk_limit + FAN_GMAX) * k;
            const bool defer = !cex && (n <= 512 ? rk <= 16L * n
                                       : defer_on && ((n <= ctx->compress_direct_max_n && rk <= 16L * n) ||
                                                      (n >= ctx->compress_factor_min_n && rk + next + 64 <= n)));
            if (!defer) {
                ldlt_compress(ctx, *Xw, ctf, cex);
                last_compression = 0;
                run.hist_ok = false;              // (the increments are no longer the V_j)
            }
        }
        if (resid && !resid->blocks.empty()) resid->blocks[0].L = R;      // (the factor may have moved to a fan group's buffer)
    }
}

AdiResult adi_finish(AdiRun& run) {
    Ctx* ctx = run.ctx;
    FactorCache* cache = run.cache;
    AdiResult& res = run.res;
    const AdiOptions& opt = run.opt;
    const AdiOptions& opt_in = run.opt_in;
    const double ctf = run.ctf, abstol = run.abstol;
    const bool cex = run.cex;
    auto& Xw = run.Xw;
    int& last_compression = run.last_compression;
    auto& all_shifts = run.all_shifts;
    // look-ahead factorisations the solve did not get to use stay in a persistent cache (Cyclic lists): whatever runs on the main stream from
    // here on — the next Lyapunov solve finds them "known" — is ordered behind them
    for (auto& kv : run.prefetch_ev) (void)hipStreamWaitEvent(ctx->stream, kv.second.ev, 0);
    res.hist_ok = opt.keep_history && run.hist_ok; res.Tm = run.Tm; res.alpha_res = run.alpha_res; res.tdiag = run.tdiag;
    if (run.finalized || run.res.rhs_cols == 0 || (run.iters_host == 0 && run.res.converged)) { run.finalized = true; return res; }
    run.finalized = true;
    auto check_used = [&]() { run.check_used(); };
    check_used();
    if (opt_in.shifts.kind != ShiftSpec::CYCLIC) {
        // self-generated shifts (Projection, per-solve Heuristic) never come back: their factors must not outlive the solve,
        // or device memory grows with the total number of ADI iterations of a time loop
        for (auto& key : cache->fresh) { cache->real.erase(key); cache->cplx_.erase(key); }
    }
    cache->fresh.clear();
    if (opt.compression && last_compression > 0 && (opt.final_compress || cex)) ldlt_compress(ctx, *Xw, ctf, cex);   // adi.jl:78-80
    cache->iters_hint = res.iters;
    all_shifts.resize(res.iters);
    res.shifts = all_shifts;
    res.X = Xw;
    res.converged = res.res_norm <= abstol;
    if (run.max_growth > ctx->pivot_growth_warn && opt.given_residual) res.warnings |= 16;      // (no right-hand side object to verify against)
    else if (run.max_growth > ctx->pivot_growth_warn) {
        // The pivot-free LU met huge multipliers: the residual recurrence R <- R - 2 mu E'V may not describe X any more.  The claim is
        // checked once against the residual evaluated from scratch (lyapunov/residual.jl:3-31); a solve that only looks converged is
        // reported as not converged.
        res.warnings |= 16;
        LDLt Ccopy = run.Crhs;
        LDLtP tr = gale_residual(ctx, run.op, Ccopy, Xw, 4.0, true);
        const double tn = tr->rank() ? ldlt_norm(ctx, *tr) : 0.0;
        if (tn > 10.0 * std::max(res.res_norm, abstol)) { res.res_norm = tn; res.converged = tn <= abstol; }
    }
    if (!res.converged) res.warnings |= 1;
    return res;
}

AdiResult adi_solve(Ctx* ctx, const GaleOperator& op, LDLt& C, const LDLtP& initial_guess, const AdiOptions& opt_in,
                    FactorCache* cache) {
    static const bool tm = env_trace("rec");
    static double tb = 0, ta = 0, tf = 0; static long ns = 0;
    const auto t0 = std::chrono::steady_clock::now();
    auto run = adi_begin(ctx, op, C, initial_guess, opt_in, cache);
    const auto t1 = std::chrono::steady_clock::now();
    while (!run->finished) adi_advance(*run, 1 << 30);
    const auto t2 = std::chrono::steady_clock::now();
    AdiResult r = adi_finish(*run);
    if (tm) {
        const auto t3 = std::chrono::steady_clock::now();
        auto us = [](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
        tb += us(t0, t1); ta += us(t1, t2); tf += us(t2, t3);
        if (++ns % 12 == 0) std::fprintf(stderr, "[adi_solve host, us per solve] begin (residual) %.0f | advance %.0f | finish %.0f\n", tb / ns, ta / ns, tf / ns);
    }
    return r;
}
// The iterate and the residual object of a running solve as the reference's observer sees them at adi.jl:119 (Callbacks.jl:97-107):
// X shares its factors with the solver (increments are never modified; a later compression replaces the list, not the buffers), the
// residual factor is updated in place by the iteration and is therefore copied.
void adi_snapshot(AdiRun& run, LDLtP* X, LDLtP* resid) {
    Ctx* ctx = run.ctx;
    if (X) *X = std::make_shared<LDLt>(*run.Xw);
    if (resid) {
        auto r = std::make_shared<LDLt>();
        r->n = run.n;
        if (run.k > 0 && run.resid && !run.resid->blocks.empty()) {
            Mat Rc(ctx, run.n, run.k);
            copy_mat(ctx, run.R, Rc);
            r->blocks.push_back({Rc, run.Tm, run.alpha_res, run.tdiag, false});
        }
        *resid = r;
    }
}
std::vector<std::complex<double>> adi_shifts_since(const AdiRun& run, int from) {
    std::vector<std::complex<double>> out;
    const int upto = std::min<int>(run.res.iters, (int)run.all_shifts.size());
    for (int i = std::max(from, 0); i < upto; ++i) out.push_back(run.all_shifts[(size_t)i]);
    return out;
}
bool adi_isdone(const AdiRun& run) { return run.finished; }
void adi_peek(const AdiRun& run, int* iters, double* res_norm, double* abstol) { *iters = run.res.iters; *res_norm = run.res.res_norm; *abstol = run.abstol; }


// =============================================================================================
// Penzl's heuristic: Ritz values of E^-1 F and F^-1 E by two Arnoldi runs from ones(n), everything on the device
// (/root/reference/src/shifts/heuristic.jl:39-66,103-130).  The engine holds the transposed operators, whose spectra are the same.
// F' = Fs' + inv(alpha) Vt U' ; products add the rank-m term, solves go through Sherman-Morrison-Woodbury (heuristic.jl:51-60).
// The Hessenberg matrix is accumulated on the device and downloaded once; its eigenvalues come from the host QR (hostla.hpp).
// =============================================================================================
__global__ void k_scale_inv_norm(int n, const double* __restrict__ w, const double* __restrict__ nrm2, double* __restrict__ out, double* __restrict__ hslot) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const double nr = sqrt(nrm2[0]);
    if (i == 0) *hslot = nr;
    if (i < n) out[i] = w[i] / nr;
}

static std::vector<std::complex<double>> arnoldi_ritz(Ctx* ctx, int n, int k, const std::function<void(const Mat&, Mat&)>& apply) {
    DRE_REQUIRE(k >= 1 && k < n, "heuristic shifts: Krylov dimension out of range");
    Mat V(ctx, n, k + 1), H(ctx, k + 1, k), w(ctx, n, 1);
    DevArr<double> nr(ctx, 1);
    fill_mat(ctx, H, 0.0);
    { Mat v0 = V.colsview(0, 1); fill_mat(ctx, v0, 1.0 / std::sqrt((double)n)); }         // b0 = ones(n) (heuristic.jl:68-80), normalised
    for (int j = 0; j < k; ++j) {
        Mat x = V.colsview(j, 1);
        apply(x, w);
        Mat Vj = V.colsview(0, j + 1);
        Mat hcol = H.view(0, j, j + 1, 1);
        gemm(ctx, true, false, 1.0, Vj, w, 0.0, hcol, nullptr, "gemm_arnoldi");           // classical Gram-Schmidt, twice
        gemm(ctx, false, false, -1.0, Vj, hcol, 1.0, w, nullptr, "gemm_arnoldi");
        Mat g2(ctx, j + 1, 1);
        gemm(ctx, true, false, 1.0, Vj, w, 0.0, g2, nullptr, "gemm_arnoldi");
        gemm(ctx, false, false, -1.0, Vj, g2, 1.0, w, nullptr, "gemm_arnoldi");
        vals_axpby(ctx, j + 1, 1.0, hcol.p, 1.0, g2.p, hcol.p);
        frob2_device(ctx, w, nr.p);
        Mat vn = V.colsview(j + 1, 1);
        hipLaunchKernelGGL(k_scale_inv_norm, dim3(ceil_div(n, 256)), dim3(256), 0, ctx->stream, n, w.p, nr.p, vn.p, H.p + (size_t)(j + 1) + (size_t)j * H.ld);
    }
    std::vector<double> hh((size_t)(k + 1) * k), hk((size_t)k * k);
    DRE_HIP(hipMemcpyAsync(hh.data(), H.p, hh.size() * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    DRE_HIP(hipStreamSynchronize(ctx->stream));
    for (int c = 0; c < k; ++c) for (int r = 0; r < k; ++r) hk[r + (size_t)c * k] = hh[r + (size_t)c * (k + 1)];
    return host_eigvals(k, hk);
}

void heuristic_ritz(Ctx* ctx, const GaleOperator& op, int kplus, int kminus, std::vector<std::complex<double>>& rplus,
                    std::vector<std::complex<double>>& rminus) {
    const Pencil& P = *op.P;
    const int n = P.n, m = op.has_lr ? op.U.cols : 0;
    Factor<double> fE, fF;
    mf_factor<double>(ctx, P, op.valFt.p, P.valEt.p, 0.0, 1.0, fE);        // E'
    mf_factor<double>(ctx, P, op.valFt.p, P.valEt.p, 1.0, 0.0, fF);        // Fs'
    mf_check(ctx, fE); mf_check(ctx, fF);
    Mat W, Sinv, t1, t2;
    DevArr<int> serr(ctx, 1);
    if (m) {
        DRE_REQUIRE(m <= 32, "SMW: more than 32 low-rank columns not supported");
        W = Mat(ctx, n, m); copy_mat(ctx, op.Vt, W);
        mf_solve<double>(ctx, P, fF, W.p, W.ld, m, nullptr);               // Fs'^-1 Vt
        Mat S(ctx, m, m); Sinv = Mat(ctx, m, m); t1 = Mat(ctx, m, 1); t2 = Mat(ctx, m, 1);
        gemm(ctx, true, false, 1.0, op.U, W, 0.0, S, nullptr, "gemm_arnoldi");
        DRE_HIP(hipMemsetAsync(serr.p, 0, sizeof(int), ctx->stream));
        hipLaunchKernelGGL((k_sinv<double>), dim3(1), dim3(64), 0, ctx->stream, m, S.p, S.ld, op.alpha, Sinv.p, (const AdiState*)nullptr, serr.p);
    }
    // w = E'^-1 (F' x)
    rplus = arnoldi_ritz(ctx, n, kplus, [&](const Mat& x, Mat& w) {
        spmm(ctx, P, op.valFt.p, x, w, 1.0, 0.0, nullptr);
        if (m) {
            gemm(ctx, true, false, 1.0, op.U, x, 0.0, t1, nullptr, "gemm_arnoldi");
            gemm(ctx, false, false, 1.0 / op.alpha, op.Vt, t1, 1.0, w, nullptr, "gemm_arnoldi");
        }
        mf_solve<double>(ctx, P, fE, w.p, w.ld, 1, nullptr);
    });
    // w = F'^-1 (E' x)
    rminus = arnoldi_ritz(ctx, n, kminus, [&](const Mat& x, Mat& w) {
        spmm(ctx, P, P.valEt.p, x, w, 1.0, 0.0, nullptr);
        mf_solve<double>(ctx, P, fF, w.p, w.ld, 1, nullptr);
        if (m) {
            gemm(ctx, true, false, 1.0, op.U, w, 0.0, t1, nullptr, "gemm_arnoldi");
            gemm(ctx, false, false, 1.0, Sinv, t1, 0.0, t2, nullptr, "gemm_arnoldi");
            gemm(ctx, false, false, -1.0, W, t2, 1.0, w, nullptr, "gemm_arnoldi");
        }
    });
    if (m) {
        int herr = 0;
        DRE_HIP(hipMemcpyAsync(&herr, serr.p, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        DRE_HIP(hipStreamSynchronize(ctx->stream));
        if (herr) throw Error(ERR_SINGULAR, "heuristic shifts: SMW capacitance matrix is singular");
    }
}

// Host logic of Penzl's heuristic: stabilisation of the Ritz values (shifts/helpers.jl:115-140) and the greedy min-max
// selection (shifts/heuristic.jl:82-101); conjugate pairs are appended adjacently.
static std::vector<std::complex<double>> stabilize_ritz(std::vector<std::complex<double>> v, int* warnings) {
    std::vector<std::complex<double>> stable;
    for (auto& x : v) if (x.real() < 0.0) stable.push_back(x);
    if (stable.size() == v.size()) return v;
    if (stable.empty()) {                      // all unstable: flip them (helpers.jl:136-138)
        if (warnings) *warnings |= 8;
        for (auto& x : v) x = std::complex<double>(-x.real(), x.imag());
        return v;
    }
    if (warnings) *warnings |= 4;              // some unstable: discard them (helpers.jl:133-134)
    return stable;
}
static std::vector<std::complex<double>> heuristic_select(const std::vector<std::complex<double>>& R, int nshifts) {
    DRE_REQUIRE(!R.empty(), "heuristic shifts: no Ritz values");
    auto sfun = [](std::complex<double> t, const std::vector<std::complex<double>>& P) {
        double out = 1.0;
        for (auto& p : P) out *= std::abs(t - p) / std::abs(t + p);
        return out;
    };
    size_t best = 0; double bestv = 0.0;
    for (size_t i = 0; i < R.size(); ++i) {
        double mx = 0.0;
        for (auto& t : R) mx = std::max(mx, sfun(t, {R[i]}));
        if (i == 0 || mx < bestv) { best = i; bestv = mx; }
    }
    std::vector<std::complex<double>> P;
    auto push = [&](std::complex<double> p) { P.push_back(p); if (p.imag() != 0.0) P.push_back(std::conj(p)); };
    push(R[best]);
    while ((int)P.size() < nshifts) {
        size_t arg = 0; double mv = -1.0;
        for (size_t i = 0; i < R.size(); ++i) { const double v = sfun(R[i], P); if (v > mv) { mv = v; arg = i; } }
        push(R[arg]);
    }
    return P;
}
std::vector<std::complex<double>> heuristic_shift_values(Ctx* ctx, const GaleOperator& op, int nshifts, int kplus, int kminus, int* warnings) {
    std::vector<std::complex<double>> rp, rm;
    heuristic_ritz(ctx, op, kplus, kminus, rp, rm);
    rp = stabilize_ritz(rp, warnings);
    rm = stabilize_ritz(rm, warnings);
    for (auto& v : rm) rp.push_back(1.0 / v);
    return heuristic_select(rp, nshifts);
}

// =============================================================================================
// Rosenbrock drivers
// =============================================================================================
struct Feedback { Mat L, D, BtLD, EtL, Kt; double alpha; bool diag; };

static Feedback feedback(Ctx* ctx, const GdreProblem& prob, LDLt& X, double ctf, bool cex) {
    // alpha, L, D = X;  BtLD = (B'L) D [*alpha];  K = BtLD (L'E)     (lowrank_ros1.jl:25-28,53-56)
    const Pencil& P = *prob.P;
    ldlt_destructure(ctx, X, ctf, cex);
    const LBlock& b = X.blocks[0];
    Feedback f;
    f.L = b.L; f.D = b.D; f.alpha = b.alpha; f.diag = b.diag;
    const int r = b.L.cols, m = prob.B.cols;
    Mat BtL(ctx, m, r);
    gemm(ctx, true, false, 1.0, prob.B, b.L, 0.0, BtL);
    f.BtLD = Mat(ctx, m, r);
    gemm(ctx, false, false, b.alpha, BtL, b.D, 0.0, f.BtLD);
    f.EtL = Mat(ctx, P.n, r);
    spmm(ctx, P, P.valEt.p, b.L, f.EtL, 1.0, 0.0);
    f.Kt = Mat(ctx, P.n, m);
    if (r > 0) gemm(ctx, false, true, 1.0, f.EtL, f.BtLD, 0.0, f.Kt);
    else fill_mat(ctx, f.Kt, 0.0);
    return f;
}

// The same for a block list X = sum_b alpha_b L_b D_b L_b' that is NOT compressed first (small n, Ros1 between two compressions of X):
// L = [L_1 ... L_p] concatenated, BtLD = (B'L) blockdiag(alpha_b D_b), K' = (E'L) BtLD'.
static Feedback feedback_blocks(Ctx* ctx, const GdreProblem& prob, const LDLt& X) {
    const Pencil& P = *prob.P;
    Feedback f;
    const int c = X.rank(), m = prob.B.cols;
    f.L = (X.blocks.size() == 1) ? X.blocks[0].L : hcat_blocks(ctx, X);
    f.alpha = 1.0; f.diag = false;
    Mat BtL(ctx, m, c);
    gemm(ctx, true, false, 1.0, prob.B, f.L, 0.0, BtL);
    f.BtLD = Mat(ctx, m, c);
    mul_blockdiag(ctx, BtL, X, f.BtLD);
    f.EtL = Mat(ctx, P.n, c);
    spmm(ctx, P, P.valEt.p, f.L, f.EtL, 1.0, 0.0);
    f.Kt = Mat(ctx, P.n, m);
    gemm(ctx, false, true, 1.0, f.EtL, f.BtLD, 0.0, f.Kt);
    return f;
}

// =============================================================================================
// Ros1 with X carried as a dense symmetric n x n matrix between the time steps (small n).
// At these sizes every compression already forms the n x n matrix L D L' (the factors have more columns than rows: warm start
// + ~17 ADI increments of ~64 columns each), so the factored form buys nothing between two steps: the compression of X after
// every Lyapunov solve (adi.jl:78-80) — a strictly sequential chain of ~9 Householder panels per step that bounded the whole time
// loop — disappears, and the warm-start residual of the step's Lyapunov equation (lyapunov/residual.jl:3-31 applied to
// lowrank_ros1.jl:39-47) collapses to the Riccati residual
//     Res = C'C + K'K + E'XE/tau + F'XE + E'XF = C'C - K'K + A'XE + (A'XE)'          (F = A - E/(2 tau) - B K,  K = B'XE),
// three SpMMs and one fused assembly kernel.  The ADI iteration itself is unchanged (low-rank residual factor, low-rank increments,
// adi.jl:97-179); X_i = X_{i-1} + sum_j (-2 mu_j) V_j T V_j' is one GEMM.  The LDL' form of X is produced once at the end (and by
// the generic path whenever save_state asks for every X(t)).
// =============================================================================================
__global__ __launch_bounds__(256) void k_dense_residual(int n, int q, int m, const double* __restrict__ Ct, int ldc, const double* __restrict__ Kt, int ldk,
                                                        const double* __restrict__ M, int ldm, const double* __restrict__ EY, int ldey, double inv_tau,
                                                        double* __restrict__ Res, int ldres, double* __restrict__ part) {
    __shared__ double red[17];
    const int i = blockIdx.x * 16 + (threadIdx.x & 15), j = blockIdx.y * 16 + (threadIdx.x >> 4);
    double s = 0.0, s2 = 0.0;
    if (i < n && j < n) {
        double cc = 0.0, kk = 0.0;
        for (int l = 0; l < q; ++l) cc += Ct[i + (size_t)l * ldc] * Ct[j + (size_t)l * ldc];
        for (int l = 0; l < m; ++l) kk += Kt[i + (size_t)l * ldk] * Kt[j + (size_t)l * ldk];
        const double mm = M[i + (size_t)j * ldm] + M[j + (size_t)i * ldm];
        const double ey = 0.5 * (EY[i + (size_t)j * ldey] + EY[j + (size_t)i * ldey]);
        const double res = (cc - kk) + mm;
        Res[i + (size_t)j * ldres] = res;
        const double rhs = (cc + kk) + inv_tau * ey;              // right-hand side of the step's Lyapunov equation (lowrank_ros1.jl:42-43)
        s = rhs * rhs;
        s2 = res * res;
    }
    // block_sum (dense.hip) is not visible here: fixed-order reduction through LDS
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    s = wave_sum_t<double>(s);
    s2 = wave_sum_t<double>(s2);
    if (lane == 0) { red[wave] = s; red[4 + wave] = s2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const size_t slot = blockIdx.x + (size_t)gridDim.x * blockIdx.y, ntot = (size_t)gridDim.x * gridDim.y;
        part[slot] = (red[0] + red[1]) + (red[2] + red[3]);
        part[ntot + slot] = (red[4] + red[5]) + (red[6] + red[7]);      // ||Res||_F^2: the first termination norm of the band reduction
    }
}
// control block of the Lyapunov solve: residual = R D R' with orthonormal R, so its norm is ||D||_F
// Workgroup 0: control block of the Lyapunov solve.  Workgroups 1..: the initial residual R (n x J) in the B-operand lane order of the fast
// chain (dense.hpp, AdiFastArgs::Rpc; four 64-entry blocks per workgroup) — rides on this launch instead of a launch of its own.
__global__ __launch_bounds__(256) void k_adi_init_state(int J, const double* __restrict__ D, int ldd, const double* __restrict__ tols, int maxiters, AdiState* st,
                                                        double* __restrict__ nws, int nws_n, int n, int ct, int nblk, const double* __restrict__ R, int ldr,
                                                        double* __restrict__ Rp) {
    if (blockIdx.x > 0) {
        const int blk = (blockIdx.x - 1) * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
        if (blk >= nblk) return;
        const int t = blk / ct, j = blk - t * ct;
        const int row = 4 * t + (lane >> 4), col = 16 * j + (lane & 15);
        Rp[(size_t)blk * 64 + lane] = (row < n && col < J) ? R[row + (size_t)col * ldr] : 0.0;
        return;
    }
    __shared__ double red[4];
    for (int i = threadIdx.x; i < nws_n; i += 256) nws[i] = 0.0;        // meeting point of the fast chain's norm workgroups (was a memset of its own)
    double s = 0.0;
    for (int id = threadIdx.x; id < J * J; id += 256) { const double x = D[id % J + (size_t)(id / J) * ldd]; s += x * x; }
    s = wave_sum_t<double>(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const double nrm = sqrt((red[0] + red[1]) + (red[2] + red[3]));
        st->iters = 0; st->maxiters = maxiters; st->smw_singular = 0;
        st->abstol = tols[0]; st->res_norm = nrm; st->norms[0] = nrm;
        st->done = (nrm <= tols[0]) ? 1 : 0;
    }
}

typedef double v4d __attribute__((ext_vector_type(4)));
// 16-row strip of  W = Apk X  (X: n x ncols <= 32, column-major), K split over the four waves; returns this thread's element (row lk + 4 wave,
// column 16 j + lr) of both column tiles in v0 / v1
__device__ __forceinline__ void group_thin_tile(const double* __restrict__ Apk_strip, const double* __restrict__ X, int ldx, int n, int ncols,
                                                double (*part)[2][4][64], double& v0, double& v1) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lk = lane >> 4, lr = lane & 15;
    const int kst = (n + 3) >> 2, per = (kst + 3) >> 2;
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const int t0 = wv * per, t1 = min(kst, t0 + per);
    const bool c0ok = lr < ncols, c1ok = 16 + lr < ncols;
    const double* __restrict__ ap = Apk_strip + lane;
    const double* __restrict__ x0 = X + (size_t)(c0ok ? lr : 0) * ldx;
    const double* __restrict__ x1 = X + (size_t)(c1ok ? 16 + lr : 0) * ldx;
    v4d acc0 = (v4d){0.0, 0.0, 0.0, 0.0}, acc1 = (v4d){0.0, 0.0, 0.0, 0.0};
    for (int tb = t0; tb < t1; tb += 24) {
        double av[24], b0[24], b1[24];
#pragma unroll
        for (int u = 0; u < 24; ++u) {
            const int t = min(tb + u, t1 - 1);
            const int c = min(4 * t + lk, n - 1);
            av[u] = ap[(size_t)t * 64]; b0[u] = x0[c]; b1[u] = x1[c];
        }
#pragma unroll
        for (int u = 0; u < 24; ++u) {
            const bool kok = (tb + u < t1) && 4 * (tb + u) + lk < n;
            const double a = kok ? av[u] : 0.0;
            acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, (kok && c0ok) ? b0[u] : 0.0, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, (kok && c1ok) ? b1[u] : 0.0, acc1, 0, 0, 0);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) { part[wave][0][r][lane] = acc0[r]; part[wave][1][r][lane] = acc1[r]; }
    __syncthreads();
    v0 = ((part[0][0][wave][lane] + part[1][0][wave][lane]) + part[2][0][wave][lane]) + part[3][0][wave][lane];
    v1 = ((part[0][1][wave][lane] + part[1][1][wave][lane]) + part[2][1][wave][lane]) + part[3][1][wave][lane];
}
// W_s = stack_s X for every shift s of the cycle in one launch (X: n x ncols <= 32, the same for all): the SMW products N K', E'N K', B'N K'
struct StackThinBatch { const double* Apk[16]; double* W[16]; };
__global__ __launch_bounds__(256) void k_stack_thin(int M, int n, int ncols, const double* __restrict__ X, int ldx, int ldw, StackThinBatch bt) {
    __shared__ double part[4][2][4][64];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lk = lane >> 4, lr = lane & 15;
    const int kst = (n + 3) >> 2;
    double v0, v1;
    group_thin_tile(bt.Apk[blockIdx.y] + (size_t)blockIdx.x * kst * 64, X, ldx, n, ncols, part, v0, v1);
    const int orow = blockIdx.x * 16 + lk + 4 * wave;
    if (orow >= M) return;
    double* __restrict__ W = bt.W[blockIdx.y];
    if (lr < ncols) W[orow + (size_t)lr * ldw] = v0;
    if (16 + lr < ncols) W[orow + (size_t)(16 + lr) * ldw] = v1;
}
// SMW products of every shift of a real Cyclic list for the low-rank factor (U, Vt) of `op`, and the SMW-folded packed stacks of the
// fast chain (dense.hip).  Factors, dense inverses and stacked inverses come from the cache (built on first use).  false: some shift
// cannot take the dense-inverse path (complex, or its inverse was rejected by the condition estimate).
struct CycleOps {
    std::vector<const double*> wks_pos; // per position of the cycle: [N K' Sinv; E'N K' Sinv] (2n x m, leading dimension 2n) of its shift, or null (no low-rank part)
    std::vector<const double*> gpack;   // group chain: packed group stack per start position (index = start / g), empty = not available
    int group_g = 0;
    bool single_built = true;           // the single-iteration packed stacks (k_adi_fast) exist; false: build_single makes them on demand
    std::function<void(Ctx*)> build_single;
    std::vector<double*> pack;          // per position of the cycle
    std::vector<std::shared_ptr<FactorEntry<double>>> fe;
    std::vector<Mat> keep;
    std::vector<BufP> keepb;
    DevArr<int> serr;
    DevArr<long long> serr8;      // the breakdown flag as an 8-byte word (read back together with the control block)
};
static void ensure_stack(Ctx* ctx, const GaleOperator& op, FactorEntry<double>& fe) {
    const Pencil& P = *op.P;
    const int n = P.n, mm = op.has_lr ? op.U.cols : 0;
    if (!fe.stack.empty() && fe.stack_m == mm && (!mm || fe.stack_U == (const void*)op.U.p)) return;
    Mat stk(ctx, 2 * n + mm, n);
    { Mat top = stk.view(0, 0, n, n); copy_mat(ctx, fe.dinv, top); }
    { Mat mid = stk.view(n, 0, n, n); spmm(ctx, P, P.valEt.p, fe.dinv, mid, 1.0, 0.0, nullptr); }
    if (mm) { Mat bot = stk.view(2 * n, 0, mm, n); gemm(ctx, true, false, 1.0, op.U, fe.dinv, 0.0, bot, nullptr, "gemm_dinv"); }
    fe.stack = stk; fe.stack_U = (const void*)op.U.p; fe.stack_m = mm;
}
// Acceptance norms of the batched set-up below, as partial sums the host adds up: out[17 z] = ||F + mu_z E||_F^2 (value arrays),
// out[17 z + 1 .. 17 z + 16] = sums of squares of sixteen column slabs of N_z (n x n, leading dimension ldw)
struct SetupNormZ { double mu[MF_ZMAX]; };
__global__ __launch_bounds__(256) void k_setup_norms_z(int nnz, const double* __restrict__ valF, const double* __restrict__ valE, SetupNormZ mz, int n,
                                                       const double* __restrict__ W, int ldw, long wz, double* __restrict__ out) {
    const int z = blockIdx.x, part = blockIdx.y, tid = threadIdx.x;
    double s = 0.0;
    if (part == 0) {
        const double mu = mz.mu[z];
        for (int i = tid; i < nnz; i += 256) { const double v = valF[i] + mu * valE[i]; s += v * v; }
    } else {
        const double* __restrict__ Wz = W + (size_t)z * wz;
        const int cw = (n + 15) / 16, c0 = (part - 1) * cw, c1 = min(n, c0 + cw);
        for (int c = c0; c < c1; ++c)
            for (int r = tid; r < n; r += 256) { const double v = Wz[r + (size_t)c * ldw]; s += v * v; }
    }
    __shared__ double sh[4];
    s = wave_sum_t<double>(s);
    if ((tid & 63) == 0) sh[tid >> 6] = s;
    __syncthreads();
    if (tid == 0) out[17 * z + part] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}
// Set-up of a whole real Cyclic list on the dense-inverse path (n <= dense_inv_max_n) in SHARED launches: every missing factorisation
// (mf_factor_batch: one launch per tree level for all shifts), the explicit inverses N_z = M_z^-1 through one batched solve with the identity
// as common right-hand side (mf_solve_batch), the stacked inverses [N_z; E'N_z; U'N_z] of all shifts as ONE matrix (one SpMM, one GEMM) and
// the acceptance norms — ~25 launches instead of ~30 per shift on helper streams (1.6 of the 2.2 ms of the first time step at n = 371).
// Fills the factor cache; shifts it cannot take (already cached, more than MF_ZMAX, sweeps without the matrix cores) go through get_factor.
static void cycle_setup_batched(Ctx* ctx, const GaleOperator& op, const std::vector<std::complex<double>>& values, FactorCache* cache) {
    const Pencil& P = *op.P;
    const int n = P.n, mm = op.has_lr ? op.U.cols : 0;
    if (!cache->enabled || !P.use_mfma_sweeps || n > ctx->dense_inv_max_n || ctx->setup_batched <= 0) return;
    std::vector<double> todo;
    for (auto& mu : values) {
        bool dup = cache->real.count(std::make_tuple(op.tag, mu.real(), 0.0)) > 0;
        for (double t : todo) dup = dup || t == mu.real();
        if (!dup && (int)todo.size() < MF_ZMAX) todo.push_back(mu.real());
    }
    const int g = (int)todo.size();
    if (g < 2) return;
    std::vector<std::shared_ptr<FactorEntry<double>>> fes;
    std::vector<Factor<double>*> fp;
    std::vector<const Factor<double>*> cf;
    for (int z = 0; z < g; ++z) { fes.push_back(std::make_shared<FactorEntry<double>>()); fp.push_back(&fes.back()->f); cf.push_back(&fes.back()->f); }
    mf_factor_batch<double>(ctx, P, op.valFt.p, P.valEt.p, 1.0, todo.data(), fp.data(), g);
    Mat STK(ctx, 2 * n + mm, n * g), Id(ctx, n, n);
    set_identity(ctx, Id, 1.0);
    if (!mf_solve_batch(ctx, P, cf.data(), g, Id.p, Id.ld, n, STK.p, STK.ld, n)) return;       // (the factors are dropped: get_factor redoes them one by one)
    DevArr<double> nr(ctx, (size_t)17 * g);
    SetupNormZ mz; std::memset(&mz, 0, sizeof(mz));
    for (int z = 0; z < g; ++z) mz.mu[z] = todo[(size_t)z];
    hipLaunchKernelGGL(k_setup_norms_z, dim3((unsigned)g, 17), dim3(256), 0, ctx->stream, P.nnz, (const double*)op.valFt.p, (const double*)P.valEt.p, mz, n,
                       (const double*)STK.p, STK.ld, (long)n * STK.ld, nr.p);
    {
        Mat top = STK.view(0, 0, n, n * g), mid = STK.view(n, 0, n, n * g);
        spmm(ctx, P, P.valEt.p, top, mid, 1.0, 0.0, nullptr);
        if (mm) { Mat bot = STK.view(2 * n, 0, mm, n * g); gemm(ctx, true, false, 1.0, op.U, top, 0.0, bot, nullptr, "gemm_dinv"); }
    }
    std::vector<double> hp((size_t)17 * g), h((size_t)2 * g, 0.0);
    ctx_fetch(ctx, nr.p, hp.size() * sizeof(double), hp.data());
    for (int z = 0; z < g; ++z) { h[2 * z] = hp[17 * z]; for (int i = 1; i <= 16; ++i) h[2 * z + 1] += hp[17 * z + i]; }
    const std::vector<double> gr = mf_check_batch(ctx, cf);
    for (int z = 0; z < g; ++z) {
        auto& fe = *fes[(size_t)z];
        fe.growth = gr[(size_t)z]; fe.checked = true;
        fe.f.allow_topinv = true;
        const double cond_est = std::sqrt(h[2 * z]) * std::sqrt(h[2 * z + 1]);
        if (cond_est == cond_est && cond_est < 1e7 && fe.f.nperturbed <= 0) {
            fe.stack = STK.view(0, z * n, 2 * n + mm, n); fe.stack_U = (const void*)op.U.p; fe.stack_m = mm;
            fe.dinv = STK.view(0, z * n, n, n); fe.dense = true;
        }
        const auto key = std::make_tuple(op.tag, todo[(size_t)z], 0.0);
        cache->real[key] = fes[(size_t)z]; cache->fresh.push_back(key); cache->nfactor++;
    }
}
static bool cycle_ops_prepare(Ctx* ctx, const GaleOperator& op, const std::vector<std::complex<double>>& values, FactorCache* cache, CycleOps& co,
                              std::vector<Mat>* wks_store = nullptr /* persistent 2n x m buffers per position of the cycle (group chain) */,
                              const std::vector<Mat>* spack = nullptr /* packed stacks per position (group chain): thin products without k_gemm */,
                              bool defer_single = false /* the single-iteration packs are only needed by a fallback chunk: build on demand */) {
    const Pencil& P = *op.P;
    const int n = P.n, m = op.has_lr ? op.U.cols : 0;
    if (m > 32) return false;
    std::map<double, double*> by_mu;
    std::map<double, const double*> wks_by_mu;
    std::vector<GemmBatchDesc> descs;
    std::vector<SmwBatch> hb;
    std::vector<const double*> stacks, wks; std::vector<double*> outs;
    for (auto& mu : values) if (mu.imag() != 0.0) return false;
    cycle_setup_batched(ctx, op, values, cache);
    {
        // Every factorisation and dense inverse of the cycle is enqueued before the single read-back of the acceptance norms.  Each of
        // them is a chain of ~25 small kernels (assembly, one factorisation launch per tree level, n unit right-hand sides, norms) that
        // uses a few CUs: the chains of different shifts go to different helper streams and run side by side.
        DeferredDense dd;
        dd.cap = std::min<int>((int)values.size(), 256);
        dd.norms = DevArr<double>(ctx, (size_t)2 * dd.cap);
        int todo = 0;
        for (auto& mu : values) if (!cache->enabled || !cache->real.count(std::make_tuple(op.tag, mu.real(), mu.imag()))) ++todo;
        const int nh = (cache->enabled && todo > 1) ? std::min(todo, std::max(0, ctx->setup_streams)) : 0;
        if (nh > 1) {
            while ((int)ctx->helpers.size() < nh) {
                auto hc = std::make_unique<Ctx>();
                hc->device = ctx->device; hc->num_cus = ctx->num_cus;
                hc->stream = create_stream(2);
                hc->timer = std::make_unique<KernelTimer>();
                hipEvent_t ev;
                DRE_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
                ctx->helpers.push_back(std::move(hc)); ctx->helper_ev.push_back(ev);
            }
            if (!ctx->helper_e0) DRE_HIP(hipEventCreateWithFlags(&ctx->helper_e0, hipEventDisableTiming));
            DRE_HIP(hipEventRecord(ctx->helper_e0, ctx->stream));         // the operator's value arrays are ready here
            for (int h = 0; h < nh; ++h) {
                Ctx* hc = ctx->helpers[(size_t)h].get();
                hc->dense_inv_max_n = ctx->dense_inv_max_n; hc->top_inverse_max_rows = ctx->top_inverse_max_rows; hc->mf_subtree = ctx->mf_subtree;
                hc->pivot_growth_warn = ctx->pivot_growth_warn; hc->pivot_growth_fail = ctx->pivot_growth_fail;
                hc->timer->enabled = ctx->timer && ctx->timer->enabled;
                DRE_HIP(hipStreamWaitEvent(hc->stream, ctx->helper_e0, 0));
            }
            int j = 0;
            for (auto& mu : values) {
                if (cache->real.count(std::make_tuple(op.tag, mu.real(), mu.imag()))) continue;
                (void)get_factor<double>(ctx->helpers[(size_t)(j % nh)].get(), op, cache, cache->real, mu, true, &dd);
                ++j;
            }
            for (int h = 0; h < nh; ++h) {
                DRE_HIP(hipEventRecord(ctx->helper_ev[(size_t)h], ctx->helpers[(size_t)h]->stream));
                DRE_HIP(hipStreamWaitEvent(ctx->stream, ctx->helper_ev[(size_t)h], 0));
            }
        } else {
            for (auto& mu : values) (void)get_factor<double>(ctx, op, cache, cache->real, mu, true, &dd);
        }
        finalize_dense(ctx, dd);
    }
    size_t pos_idx = 0;
    for (auto& mu : values) {
        const size_t pos = pos_idx++;
        auto fe = get_factor<double>(ctx, op, cache, cache->real, mu, true);
        if (!fe->dense) return false;
        ensure_stack(ctx, op, *fe);
        co.fe.push_back(fe);
        auto bm = by_mu.find(mu.real());
        if (bm == by_mu.end()) {
            Mat pk(ctx, (int)(adi_fast_pack_doubles(n) / 64), 64);
            co.keep.push_back(pk);
            const double* wksp = nullptr;
            if (m) {
                Mat WK(ctx, 2 * n + m, m);
                Mat WKS = (wks_store && pos < wks_store->size()) ? (*wks_store)[pos] : Mat(ctx, 2 * n, m);
                auto sinv = std::make_shared<Buf>(ctx, (size_t)m * m * sizeof(double));
                co.keep.push_back(WK); co.keep.push_back(WKS); co.keepb.push_back(sinv);
                descs.push_back({fe->stack.p, op.Vt.p, WK.p, nullptr, 1.0, 2 * n + m, m, n, fe->stack.ld, op.Vt.ld, WK.ld, 0});
                hb.push_back({WK.p, (double*)sinv->p, WKS.p});
                wksp = WKS.p;
            }
            stacks.push_back(fe->stack.p); wks.push_back(wksp); outs.push_back(pk.p);
            bm = by_mu.emplace(mu.real(), pk.p).first;
            wks_by_mu[mu.real()] = wksp;
        }
        co.pack.push_back(bm->second);
        co.wks_pos.push_back(wks_by_mu[mu.real()]);
    }
    if (m) {
        co.serr8 = DevArr<long long>(ctx, 1);
        co.serr = DevArr<int>();
        co.serr.buf = co.serr8.buf; co.serr.p = (int*)co.serr8.p; co.serr.n = 2;     // the kernels write the low word
        DRE_HIP(hipMemsetAsync(co.serr8.p, 0, sizeof(long long), ctx->stream));
        if (spack && spack->size() == values.size() && m <= 32 && descs.size() <= 16 && descs.size() == values.size()) {
            // WK_s = stack_s K' for every shift in one launch on the packed stacks (the general batched GEMM spends 31 us on these 749 x 371 x 7 products)
            StackThinBatch tb;
            for (size_t i = 0; i < 16; ++i) { const size_t j = i < descs.size() ? i : 0; tb.Apk[i] = (*spack)[j].p; tb.W[i] = descs[j].C; }
            TimedScope ts(ctx, "gemm_dinv", 8.0 * descs.size() * (2.0 * n + m) * n, 2.0 * descs.size() * (2.0 * n + m) * n * (double)m);
            hipLaunchKernelGGL(k_stack_thin, dim3(ceil_div(2 * n + m, 16), (unsigned)descs.size()), dim3(256), 0, ctx->stream, 2 * n + m, n, m, (const double*)op.Vt.p,
                               op.Vt.ld, 2 * n + m, tb);
        } else gemm_batched(ctx, descs, "gemm_dinv");
        for (size_t b0 = 0; b0 < hb.size(); b0 += 16) {
            SmwBatchArgs ba;
            const unsigned nb = (unsigned)std::min<size_t>(16, hb.size() - b0);
            for (unsigned i = 0; i < 16; ++i) ba.b[i] = hb[b0 + (i < nb ? i : 0)];
            TimedScope ts(ctx, "smw_batched", 8.0 * nb * (2.0 * n * m * 2.0 + 3.0 * m * m), 4.0 * nb * n * (double)m * m);
            hipLaunchKernelGGL(k_sinv_batched_args, dim3(nb), dim3(64), 0, ctx->stream, n, m, 2 * n + m, op.alpha, ba, co.serr.p);
            hipLaunchKernelGGL(k_fold_sinv_batched_args, dim3(ceil_div(2 * n * m, 256), nb), dim3(256), 0, ctx->stream, 2 * n, m, 2 * n + m, ba);
        }
    }
    if (defer_single) {
        co.single_built = false;
        co.build_single = [n, m, stacks, wks, outs](Ctx* c) { adi_fast_build(c, n, m, stacks, 2 * n + m, wks, 2 * n, outs); };
    } else adi_fast_build(ctx, n, m, stacks, 2 * n + m, wks, 2 * n, outs);
    return true;
}

// ---- group chain (dense.hip, k_adi_group): operator products of g consecutive ADI iterations --------------------------------------
// With N_s = (A' + (mu_s - 1/(2 tau)) E')^-1 (K independent, kept per shift as stack_s = [N_s; E'N_s; B'N_s]) the shifted operator of a time
// step is a rank-m correction (Sherman-Morrison-Woodbury, smw.jl:20-43):  A_s = N_s + a_s b_s',  P_s = I - 2 mu_s E'A_s = P0_s + c_s b_s'
// with a_s = -(N_s K' Sinv_s), c_s = 2 mu_s (E'N_s K' Sinv_s), b_s' = B'N_s.  For the g shifts s_0 .. s_{g-1} of a group (index i for s_i):
//   Pi_i = P_{i-1} ... P_0 = Pi0_i + X_i Y_i',      Om_i = A_i Pi_i = Om0_i + [N_i X_i, a_i] Y_{i+1}',
//   X_i  block j (j < i)  = Phi(j+1, i) c_j,        Phi(a, b) = P0_{b-1} ... P0_a   (K independent; Pi0_i = Phi(0, i), Om0_i = N_i Phi(0, i)),
//   Y_{i+1} = [Y_i, y_i],  y_i = Pi_i' b_i = D_i' + sum_{j<i} y_j M(j, i),   D_i = b_i' Phi(0, i),   M(j, i) = c_j' Psi(j, i),   Psi(j, i) = Phi(j+1, i)' b_i.
// Everything K independent (Phi, N Phi, Psi, D: GroupBase) is formed ONCE per run — one stacked product stack_b Phi(a, b) delivers
// N_b Phi(a, b), the next Phi(a, b+1) and Psi(a-1, b)' at once — and a time step costs three launches on the side stream: the left-factor
// blocks (independent thin products, no recursion), the rows of Y, and the fold + packing of the effective group stack.
struct GroupLeftDesc { const double* Mtx; int ldm; const double* v; int ldv; double* out; int ldo; double scale; int kind; int pad; };   // kind 0: out = scale Mtx v (null Mtx: copy), 1: M = scale v' PsiT'
struct GroupBase {
    int g = 0, n = 0, m = 0;
    uint64_t tag = 0;
    std::vector<double> mus;            // the cycle this base was built for
    std::vector<int> starts;            // start positions (multiples of g)
    std::vector<Mat> G0;                // per start: (2 g n) x n  rows [i n, (i+1) n) = Om0_i, rows [(g + i) n, ...) = Pi0_{i+1}
    std::vector<Mat> D;                 // per start: (g m) x n    rows [i m, (i+1) m) = D_i
    std::vector<Mat> spack;             // per position of the cycle: stack_s = [N_s; E'N_s; B'N_s] in the MFMA A-operand order
    std::vector<Mat> wks;               // persistent [N K' Sinv; E'N K' Sinv] per position of the cycle (2n x m): the descriptors point into them
    std::vector<Mat> XL, Yt, Mb;        // per start: left factors (2 g n) x (m g), Y' (m g) x n, the M(j, i) blocks (m x m each, g*g slots)
    std::vector<Mat> pack;              // packed effective group stacks
    DevArr<GroupLeftDesc> table;        // descriptors of the left-factor launch (built once per run)
    std::vector<GroupLeftDesc> host_table;
    int ndesc = 0;
};
// out(16-row strip) = scale * Mtx(strip, :) v   (v: n x m, m <= 16), K split over the four waves;  kind 1: the m x m matrix scale * v' P' with P = Mtx (m x n)
__global__ __launch_bounds__(256) void k_group_left(int n, int m, const GroupLeftDesc* __restrict__ table) {
    __shared__ double part[4][4][64];
    const GroupLeftDesc d = table[blockIdx.y];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lk = lane >> 4, lr = lane & 15;
    if (d.kind == 1) {
        if (blockIdx.x > 0) return;
        // M[u, v] = scale * sum_r vv[r, u] * PsiT[v, r]
        for (int e = tid; e < m * m; e += 256) {
            const int u = e % m, vv = e / m;
            double sacc = 0.0;
            for (int r = 0; r < n; ++r) sacc += d.v[r + (size_t)u * d.ldv] * d.Mtx[vv + (size_t)r * d.ldm];
            d.out[u + (size_t)vv * d.ldo] = d.scale * sacc;
        }
        return;
    }
    const int row0 = blockIdx.x * 16;
    if (row0 >= n) return;
    if (!d.Mtx) {
        for (int e = tid; e < 16 * m; e += 256) {
            const int r = row0 + (e & 15), c = e >> 4;
            if (r < n) d.out[r + (size_t)c * d.ldo] = d.scale * d.v[r + (size_t)c * d.ldv];
        }
        return;
    }
    const int row = row0 + lr;
    const bool rok = row < n, cok = lr < m;
    const int kst = (n + 3) >> 2, per = (kst + 3) >> 2;
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const int t0 = wv * per, t1 = min(kst, t0 + per);
    const double* __restrict__ ap = d.Mtx + (rok ? row : 0);
    const double* __restrict__ xp = d.v + (size_t)(cok ? lr : 0) * d.ldv;
    v4d acc = (v4d){0.0, 0.0, 0.0, 0.0};
    for (int tb = t0; tb < t1; tb += 24) {
        double av[24], bv[24];
#pragma unroll
        for (int u = 0; u < 24; ++u) {
            const int c = min(4 * min(tb + u, t1 - 1) + lk, n - 1);
            av[u] = ap[(size_t)c * d.ldm]; bv[u] = xp[c];
        }
#pragma unroll
        for (int u = 0; u < 24; ++u) {
            const bool kok = (tb + u < t1) && 4 * (tb + u) + lk < n;
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64((kok && rok) ? av[u] : 0.0, (kok && cok) ? bv[u] : 0.0, acc, 0, 0, 0);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) part[wave][r][lane] = acc[r];
    __syncthreads();
    const double vsum = ((part[0][wave][lane] + part[1][wave][lane]) + part[2][wave][lane]) + part[3][wave][lane];
    const int orow = row0 + lk + 4 * wave;
    if (orow < n && lr < m) d.out[orow + (size_t)lr * d.ldo] = d.scale * vsum;
}
// rows of Y' = [y_0 .. y_{g-1}]':  y_i = D_i' + sum_{j<i} y_j M(j, i).  One thread per (row, component v); the g levels are sequential, the
// row's earlier y_j go through LDS (32 rows per workgroup, 8 component slots per row).
struct GroupYOne { const double* D; const double* Mb; double* Yt; };
struct GroupYBatch { GroupYOne s[8]; };
__global__ __launch_bounds__(256) void k_group_y(int n, int m, int g, int ldd, GroupYBatch bt) {
    __shared__ double ysh[32][ADI_GROUP_MAX_G][8];
    __shared__ double msh[ADI_GROUP_MAX_G * ADI_GROUP_MAX_G][8][8];
    const GroupYOne& o = bt.s[blockIdx.y];
    const int rl = threadIdx.x >> 3, v = threadIdx.x & 7;
    const int rr = blockIdx.x * 32 + rl;
    const int r = m * g;
    for (int e = threadIdx.x; e < g * g * 64; e += 256) {
        const int ji = e >> 6, u = (e >> 3) & 7, vv = e & 7;
        msh[ji][u][vv] = (u < m && vv < m) ? o.Mb[(size_t)ji * m * m + u + (size_t)vv * m] : 0.0;
    }
    __syncthreads();
    const bool ok = rr < n && v < m;
    for (int i = 0; i < g; ++i) {
        double acc = ok ? o.D[(size_t)i * m + v + (size_t)rr * ldd] : 0.0;
        for (int j = 0; j < i; ++j)
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += ysh[rl][j][u] * msh[j * g + i][u][v];
        ysh[rl][i][v] = ok ? acc : 0.0;
        if (ok) o.Yt[(size_t)i * m + v + (size_t)rr * r] = acc;
        __syncthreads();
    }
}
// Packed effective group stack in one pass:  out = pack(G0 + XL Yt).  The rank-r product of a 16 x 16 tile is computed TRANSPOSED, so that the
// accumulator layout is the packed layout (acc[q] = K-step 4 tt + q, position lane) — the trick of k_eff_stack_mfma (dense.hip).
struct GroupFoldOne { const double* G0; const double* XL; const double* Yt; double* out; };
struct GroupFoldBatch { GroupFoldOne s[8]; };
__global__ __launch_bounds__(256) void k_group_fold(int n, int nblk, int nstrip, int kst, int r, int ldg, int ldx, GroupFoldBatch bt) {
    const GroupFoldOne& o = bt.s[blockIdx.z];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, lk = lane >> 4, lr = lane & 15;
    const int tt = blockIdx.x * 4 + wave;                          // column tile: K-steps 4 tt .. 4 tt + 3
    const int ntile = (kst + 3) >> 2;
    if (tt >= ntile) return;
    const int hs = blockIdx.y, b = hs / nstrip, s = hs - b * nstrip;
    const int rowl = 16 * s + lr;                                  // row within the block (B-operand column index j = lr)
    const bool rok = rowl < n;
    const size_t grow = (size_t)b * n + (rok ? rowl : 0);
    const int colA = 16 * tt + lr;                                 // A operand: A[i = lr][k] = Yt[k, 16 tt + lr]
    const bool cok = colA < n;
    v4d acc = (v4d){0.0, 0.0, 0.0, 0.0};
    const int ksteps = (r + 3) >> 2;
    for (int kk = 0; kk < ksteps; ++kk) {
        const int q = 4 * kk + lk;
        const bool qok = q < r;
        const double ya = (qok && cok) ? o.Yt[q + (size_t)colA * r] : 0.0;
        const double xb = (qok && rok) ? o.XL[grow + (size_t)q * ldx] : 0.0;
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(ya, xb, acc, 0, 0, 0);
    }
    // acc[q] = (XL Yt)[row = rowl, col = 16 tt + 4 q + lk]   (transposed tile: D'[i = lk + 4 q][j = lr])
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int t = 4 * tt + q, col = 4 * t + lk;
        if (t < kst) {
            const double base = (rok && col < n) ? o.G0[grow + (size_t)col * ldg] : 0.0;
            o.out[((size_t)hs * kst + t) * 64 + lane] = (rok && col < n) ? base + acc[q] : 0.0;
        }
    }
}
// next = prev - two_mu * W_mid  (n x n);  prev = null: identity
__global__ __launch_bounds__(256) void k_group_next_phi(int n, double two_mu, const double* __restrict__ prev, int ldp, const double* __restrict__ Wmid, int ldw,
                                                       double* __restrict__ next, int ldn) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (size_t)n * n) return;
    const int r = idx % n, c = idx / n;
    const double p = prev ? prev[r + (size_t)c * ldp] : (r == c ? 1.0 : 0.0);
    next[r + (size_t)c * ldn] = p - two_mu * Wmid[r + (size_t)c * ldw];
}
static int group_size_for(Ctx* ctx, int ncycle, int n, int m) {
    if (ctx->adi_group == 0 || n > ctx->adi_group_max_n || m < 1 || m > 8) return 0;
    if (ctx->adi_group > 1) return (ncycle % ctx->adi_group == 0 && ctx->adi_group <= ADI_GROUP_MAX_G && m * (ctx->adi_group - 1) <= 32) ? ctx->adi_group : 0;
    for (int g = std::min(5, ADI_GROUP_MAX_G); g >= 2; --g) if (ncycle % g == 0 && m * (g - 1) <= 32) return g;      // auto: the largest divisor of the cycle length up to 5
    return 0;
}
// (2n + m) x n stack -> A-operand order: strip of 16 rows x K-step of 4 columns = 64 consecutive doubles (one coalesced 512-byte fragment load)
__global__ __launch_bounds__(256) void k_pack_rows(int M, int n, int kst, const double* __restrict__ src, int lds_, double* __restrict__ out) {
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (t >= kst) return;
    const int row = 16 * blockIdx.y + (lane & 15), col = 4 * t + (lane >> 4);
    out[((size_t)blockIdx.y * kst + t) * 64 + lane] = (row < M && col < n) ? src[row + (size_t)col * lds_] : 0.0;
}
// Level i of the thin recursion, one launch for all start positions:  W = stack_{s_i} X_i  (X_i: n x m i, the left factor of Pi_i, in XL block
// row g + i - 1), one workgroup per 16-row strip of the packed stack, K split over the four waves, two column tiles; the epilogue writes the
// three row ranges of W where they are needed:  top -> XV_i (XL block row i),  mid -> X_{i+1} = X_i - 2 mu W_mid (XL block row g + i),
// bottom (b_i' X_i, m x m i) -> the blocks M(j, i) of the Y recursion.
struct GroupLevelOne { const double* Apk; double* XL; double* Mb; double two_mu; };
struct GroupLevelBatch { GroupLevelOne s[8]; };
__global__ __launch_bounds__(256) void k_group_level_gemm(int n, int m, int g, int i, int ldx, GroupLevelBatch bt) {
    __shared__ double part[4][2][4][64];
    const GroupLevelOne& o = bt.s[blockIdx.y];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lk = lane >> 4, lr = lane & 15;
    const int M = 2 * n + m, ncols = m * i, kst = (n + 3) >> 2;
    const double* __restrict__ Xi = o.XL + (size_t)(g + i - 1) * n;
    double vv2[2];
    group_thin_tile(o.Apk + (size_t)blockIdx.x * kst * 64, Xi, ldx, n, ncols, part, vv2[0], vv2[1]);
    const int orow = blockIdx.x * 16 + lk + 4 * wave;
    if (orow >= M) return;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const double v = vv2[j];
        const int col = 16 * j + lr;
        if (col >= ncols) continue;
        if (orow < n) o.XL[(size_t)i * n + orow + (size_t)col * ldx] = v;                                            // XV_i
        else if (orow < 2 * n) {
            const int r = orow - n;
            o.XL[(size_t)(g + i) * n + r + (size_t)col * ldx] = Xi[r + (size_t)col * ldx] - o.two_mu * v;           // X_{i+1}
        } else {
            const int vv = orow - 2 * n, jj = col / m, u = col - jj * m;
            o.Mb[(size_t)(jj * g + i) * m * m + u + (size_t)vv * m] = v;                                               // M(jj, i)[u, vv]
        }
    }
}
// K-independent part, once per run (every shift of the cycle has its stacked inverse)
static void group_base_build(Ctx* ctx, const GaleOperator& op, const std::vector<std::complex<double>>& values, const CycleOps& co, GroupBase& gb, int g) {
    const int n = op.P->n, m = op.U.cols, c = (int)values.size();
    gb = GroupBase();
    gb.g = g; gb.n = n; gb.m = m; gb.tag = op.tag;
    for (auto& v : values) gb.mus.push_back(v.real());
    const int r = m * g, M = 2 * n + m, kst = adi_fast_kst(n), nstripM = ceil_div(M, 16);
    std::vector<GroupLeftDesc> tab;
    const unsigned nn_blocks = (unsigned)(((size_t)n * n + 255) / 256);
    for (int pos = 0; pos < c; ++pos) {
        gb.wks.push_back(Mat(ctx, 2 * n, m));
        const FactorEntry<double>& fe = *co.fe[(size_t)pos];
        Mat pk(ctx, nstripM * kst, 64);
        hipLaunchKernelGGL(k_pack_rows, dim3(ceil_div(kst, 4), nstripM), dim3(256), 0, ctx->stream, M, n, kst, (const double*)fe.stack.p, fe.stack.ld, pk.p);
        gb.spack.push_back(pk);
    }
    for (int p = 0; p < c; p += g) {
        gb.starts.push_back(p);
        Mat G0(ctx, 2 * g * n, n), D(ctx, g * m, n), XL(ctx, 2 * g * n, r), Yt(ctx, r, n), Mb(ctx, m * m, g * g);
        fill_mat(ctx, XL, 0.0); fill_mat(ctx, Yt, 0.0); fill_mat(ctx, Mb, 0.0);
        Mat W(ctx, M, n);
        for (int b = 0; b < g; ++b) {       // Phi(0, b) = Pi0_b:  stack_b Pi0_b = [Om0_b; E'N_b Pi0_b; D_b],  Pi0_{b+1} = Pi0_b - 2 mu_b (mid)
            const FactorEntry<double>& fe = *co.fe[(size_t)(p + b)];
            const double two_mu = 2.0 * values[(size_t)(p + b)].real();
            Mat top, mid, bot, prev;
            if (b == 0) { top = fe.stack.view(0, 0, n, n); mid = fe.stack.view(n, 0, n, n); bot = fe.stack.view(2 * n, 0, m, n); }
            else {
                prev = G0.view((g + b - 1) * n, 0, n, n);
                gemm(ctx, false, false, 1.0, fe.stack, prev, 0.0, W, nullptr, "gemm_group_base");
                top = W.view(0, 0, n, n); mid = W.view(n, 0, n, n); bot = W.view(2 * n, 0, m, n);
            }
            Mat nxt = G0.view((g + b) * n, 0, n, n);
            hipLaunchKernelGGL(k_group_next_phi, dim3(nn_blocks), dim3(256), 0, ctx->stream, n, two_mu, b == 0 ? (const double*)nullptr : (const double*)prev.p,
                               b == 0 ? 0 : prev.ld, (const double*)mid.p, mid.ld, nxt.p, nxt.ld);
            Mat om = G0.view(b * n, 0, n, n); copy_mat(ctx, top, om);
            Mat dd = D.view(b * m, 0, m, n); copy_mat(ctx, bot, dd);
        }
        // the K-dependent blocks that are plain copies:  XV_i block i = a_i = -(N K' Sinv)_i,  X_{i+1} block i = c_i = 2 mu_i (E'N K' Sinv)_i
        for (int i = 0; i < g; ++i) {
            const Mat& w = gb.wks[(size_t)(p + i)];
            tab.push_back({nullptr, 0, w.p, w.ld, XL.p + (size_t)i * n + (size_t)(i * m) * XL.ld, XL.ld, -1.0, 0, 0});
            tab.push_back({nullptr, 0, w.p + n, w.ld, XL.p + (size_t)(g + i) * n + (size_t)(i * m) * XL.ld, XL.ld, 2.0 * values[(size_t)(p + i)].real(), 0, 0});
        }
        gb.G0.push_back(G0); gb.D.push_back(D); gb.XL.push_back(XL); gb.Yt.push_back(Yt); gb.Mb.push_back(Mb);
        gb.pack.push_back(Mat(ctx, (int)((size_t)g * adi_fast_pack_doubles(n) / 64), 64));
    }
    gb.ndesc = (int)tab.size();
    gb.table = DevArr<GroupLeftDesc>(ctx, tab.size());
    gb.host_table = std::move(tab);                      // stays alive with the base: the upload is asynchronous
    DRE_HIP(hipMemcpyAsync(gb.table.p, gb.host_table.data(), gb.host_table.size() * sizeof(GroupLeftDesc), hipMemcpyHostToDevice, ctx->stream));
}
// K-dependent part, once per time step (side stream): copies, g - 1 thin levels, the rows of Y, fold + packing
static void group_ops_prepare(Ctx* ctx, const std::vector<std::complex<double>>& values, CycleOps& co, GroupBase& gb) {
    const int n = gb.n, m = gb.m, g = gb.g, r = m * g;
    const int ns = (int)gb.starts.size();
    TimedScope tsall(ctx, "group_prepare", 8.0 * ns * ((g - 1) * (2.0 * n + m) * n + 4.0 * g * n * (double)n), 2.0 * ns * 2.0 * g * n * (double)n * r, g + 2);
    hipLaunchKernelGGL(k_group_left, dim3(ceil_div(n, 16), gb.ndesc), dim3(256), 0, ctx->stream, n, m, (const GroupLeftDesc*)gb.table.p);
    for (int i = 1; i < g; ++i) {
        GroupLevelBatch lb;
        for (int q = 0; q < 8; ++q) {
            const int qq = q < ns ? q : 0, pos = gb.starts[(size_t)qq] + i;
            lb.s[q] = {gb.spack[(size_t)pos].p, gb.XL[(size_t)qq].p, gb.Mb[(size_t)qq].p, 2.0 * values[(size_t)pos].real()};
        }
        hipLaunchKernelGGL(k_group_level_gemm, dim3(ceil_div(2 * n + m, 16), ns), dim3(256), 0, ctx->stream, n, m, g, i, gb.XL[0].ld, lb);
    }
    {
        GroupYBatch yb;
        for (int q = 0; q < 8; ++q) { const int qq = q < ns ? q : 0; yb.s[q] = {gb.D[(size_t)qq].p, gb.Mb[(size_t)qq].p, gb.Yt[(size_t)qq].p}; }
        hipLaunchKernelGGL(k_group_y, dim3(ceil_div(n, 32), ns), dim3(256), 0, ctx->stream, n, m, g, gb.D[0].ld, yb);
    }
    co.gpack.clear();
    {
        const int nstrip = adi_fast_nstrip(n), kst = adi_fast_kst(n);
        GroupFoldBatch ft;
        for (int q = 0; q < 8; ++q) { const int qq = q < ns ? q : 0; ft.s[q] = {gb.G0[(size_t)qq].p, gb.XL[(size_t)qq].p, gb.Yt[(size_t)qq].p, gb.pack[(size_t)qq].p}; }
        hipLaunchKernelGGL(k_group_fold, dim3(ceil_div((kst + 3) / 4, 4), 2 * g * nstrip, ns), dim3(256), 0, ctx->stream, n, 2 * g, nstrip, kst, r,
                           gb.G0[0].ld, gb.XL[0].ld, ft);
        for (int q = 0; q < ns; ++q) co.gpack.push_back(gb.pack[(size_t)q].p);
    }
    co.group_g = g;
    DRE_HIP(hipGetLastError());
}

// One parked host thread per GDRE solve for work that is DRIVEN beside the main loop (the side-stream compression of X has host read-backs of
// its own, so it cannot simply be enqueued): jobs are handed over through a condition variable — no thread is spawned per time step.
class SideWorker {
  public:
    ~SideWorker() { { std::lock_guard<std::mutex> lk(m_); quit_ = true; } cv_.notify_all(); if (th_.joinable()) th_.join(); }
    void submit(std::function<void()> job) {
        if (!th_.joinable()) th_ = std::thread([this] { run(); });
        { std::lock_guard<std::mutex> lk(m_); job_ = std::move(job); busy_ = true; err_ = nullptr; }
        cv_.notify_all();
    }
    bool pending() { std::lock_guard<std::mutex> lk(m_); return busy_; }
    void wait() {       // returns when the submitted job is finished; rethrows what it threw
        std::unique_lock<std::mutex> lk(m_);
        done_.wait(lk, [this] { return !busy_; });
        if (err_) { auto e = err_; err_ = nullptr; std::rethrow_exception(e); }
    }
  private:
    void run() {
        for (;;) {
            std::function<void()> job;
            { std::unique_lock<std::mutex> lk(m_); cv_.wait(lk, [this] { return quit_ || (busy_ && job_); }); if (quit_) return; job = std::move(job_); job_ = nullptr; }
            std::exception_ptr e;
            try { job(); } catch (...) { e = std::current_exception(); }
            { std::lock_guard<std::mutex> lk(m_); busy_ = false; err_ = e; }
            done_.notify_all();
        }
    }
    std::thread th_;
    std::mutex m_;
    std::condition_variable cv_, done_;
    std::function<void()> job_;
    std::exception_ptr err_;
    bool busy_ = false, quit_ = false;
};

struct DenseXState {
    Mat X;        // n x n, symmetric
    Mat P1;       // E' X
    Mat P1t;      // X E (= P1'), written by the same SpMM launch
    Mat Kt;       // K' = E' X B  (n x m)
    int hint = 0; // ADI iterations of the previous step
    GroupBase gb; // K-independent operator products of the group chain (built at the first dense step)
    // pinned host landing zone: control block, tolerances and the SMW breakdown flag come back with ONE synchronisation per chunk
    struct Landing { AdiState st; double tols[4]; int serr; };
    Landing* land = nullptr;
    // optional phase timing (DRE_PHASE_TIMING=1): events at the phase boundaries of every step, summed at the end
    bool phase_on = env_trace("phase");
    std::vector<hipEvent_t> pev;
    std::vector<int> ptag;
    void mark(Ctx* ctx, int tag) {
        if (!phase_on) return;
        hipEvent_t e; (void)hipEventCreate(&e); (void)hipEventRecord(e, ctx->stream);
        pev.push_back(e); ptag.push_back(tag);
    }
    void report() {
        if (!phase_on || pev.size() < 2) return;
        (void)hipEventSynchronize(pev.back());
        static const char* names[] = {"start", "assembly", "band_reduce", "basis+init", "adi_chain", "sync+xupdate", "feedback"};
        double acc[8] = {0};
        for (size_t i = 1; i < pev.size(); ++i) { float ms = 0; (void)hipEventElapsedTime(&ms, pev[i - 1], pev[i]); if (ptag[i] > 0 && ptag[i] < 8) acc[ptag[i]] += ms; }
        double tot = 0; for (int i = 1; i < 7; ++i) tot += acc[i];
        std::fprintf(stderr, "[phase timing, ms per solve] ");
        for (int i = 1; i < 7; ++i) std::fprintf(stderr, "%s %.2f | ", names[i], acc[i]);
        std::fprintf(stderr, "sum %.2f\n", tot);
        for (auto e : pev) (void)hipEventDestroy(e);
        pev.clear(); ptag.clear();
    }
    // (the zone belongs to the context: a pinned allocation and its release cost ~0.1 ms each, per solve)
    void attach(Ctx* ctx) {
        if (!ctx->dense_land && hipHostMalloc(&ctx->dense_land, 16384, hipHostMallocDefault) != hipSuccess) ctx->dense_land = nullptr;
        static_assert(sizeof(Landing) <= 16384, "landing zone");
        land = (Landing*)ctx->dense_land;
    }
    DenseXState() = default;
    DenseXState(const DenseXState&) = delete;
    DenseXState& operator=(const DenseXState&) = delete;
};
// One Ros1 step on the dense state.  Returns false (state untouched) when the fast chain cannot take the step.
static bool ros1_dense_step(Ctx* ctx, const GdreProblem& prob, const GaleOperator& op_base, double tau, const AdiOptions& adi, FactorCache* cache,
                            DenseXState& sx, AdiResult& ar) {
    const Pencil& P = *prob.P;
    const int n = P.n, q = prob.Ct.cols, m = prob.B.cols;
    GaleOperator op = op_base;
    op.Vt = sx.Kt;
    DRE_REQUIRE(sx.land != nullptr, "pinned host memory unavailable");
    // SMW products and the folded stacks depend on K only: they are built on the side stream while the main stream assembles and
    // compresses the residual; the ADI chain waits for them through an event
    Ctx* const wctx = (ctx->side && ctx->x_side_stream) ? ctx->side.get() : ctx;
    if (wctx != ctx) DRE_HIP(hipEventRecord(ctx->side_e1, ctx->stream));          // K of the previous step is ready here
    sx.mark(ctx, 0);
    // The side stream is set up (event wait, ~6 launches, allocations: 30-40 us of host time) while the host would otherwise WAIT for the
    // control block of the band reduction: the device is busy with the panels then, and the chain that needs the result is 150 us away.
    CycleOps co;
    bool co_ok = true, build_group_base = false;
    auto side_setup = [&]() {
        if (wctx != ctx) DRE_HIP(hipStreamWaitEvent(wctx->stream, ctx->side_e1, 0));
        // group chain: the K-independent operator products are built at the first dense step of a run; from then on the SMW products of
        // every step land in the persistent buffers the left-factor descriptors point into
        const int gwant = group_size_for(ctx, (int)adi.shifts.values.size(), n, m);       // (the MAIN context's options)
        bool gb_ok = gwant >= 2 && (int)adi.shifts.values.size() / gwant <= 8 && sx.gb.g == gwant && sx.gb.tag == op.tag && sx.gb.m == m &&
                     sx.gb.mus.size() == adi.shifts.values.size();
        for (size_t i = 0; gb_ok && i < sx.gb.mus.size(); ++i) gb_ok = sx.gb.mus[i] == adi.shifts.values[i].real();
        co_ok = cycle_ops_prepare(wctx, op, adi.shifts.values, cache, co, gb_ok ? &sx.gb.wks : nullptr, gb_ok ? &sx.gb.spack : nullptr, gb_ok);
        if (co_ok && gb_ok) group_ops_prepare(wctx, adi.shifts.values, co, sx.gb);
        build_group_base = co_ok && !gb_ok && gwant >= 2 && (int)adi.shifts.values.size() / gwant <= 8;
        if (wctx != ctx) DRE_HIP(hipEventRecord(ctx->side_e2, wctx->stream));
        if (build_group_base) {
            // first dense step of a run: the K-independent operator products (8 GEMMs + ~30 small launches per run) are formed on the side
            // stream BEHIND the event the chain waits for — this step runs the one-iteration chain, the group chain takes over from the next
            bool distinct = true;                       // (a cycle with repeated values keeps the single-iteration chain)
            for (size_t i = 0; i < adi.shifts.values.size(); ++i)
                for (size_t j = 0; j < i; ++j) distinct = distinct && adi.shifts.values[i].real() != adi.shifts.values[j].real();
            if (distinct) group_base_build(wctx, op, adi.shifts.values, co, sx.gb, gwant);
        }
    };
    // DRE_SIDE_EARLY=1: enqueue the side stream's work (SMW products, thin recursion, fold) before the assembly instead of inside the band
    // reduction's first read-back.  Measured at n = 371 with the group chain: 21.7 ms per solve early against 21.2 ms in the read-back slot (the
    // host calls delay the main stream's first kernels by more than the earlier start gains) — off by default.
    static const int side_early_env = -1;
    const bool side_early = wctx != ctx && side_early_env > 0;
    if (side_early) side_setup();
    // Riccati residual at X (= warm-start residual of the step's Lyapunov equation) and the norm of the equation's right-hand side.
    // The main stream's kernels are enqueued BEFORE the side stream is set up: the host calls for the side stream (event wait, six
    // launches) would otherwise sit in front of them while the main stream idles.
    Mat Mx(ctx, n, n), EY(ctx, n, n), Res(ctx, n, n);
    const Mat& Y = sx.P1t;                                                      // Y = X E
    spmm_dual(ctx, P, P.valAt.p, P.valEt.p, Y, Mx, EY);       // A' X E and E' X E in one pass over X E
    const int nt = ceil_div(n, 16);
    DevArr<double> part(ctx, (size_t)2 * nt * nt), tols(ctx, 4);
    hipLaunchKernelGGL(k_dense_residual, dim3(nt, nt), dim3(256), 0, ctx->stream, n, q, m, (const double*)prob.Ct.p, prob.Ct.ld, (const double*)sx.Kt.p, sx.Kt.ld,
                       (const double*)Mx.p, Mx.ld, (const double*)EY.p, EY.ld, 1.0 / tau, Res.p, Res.ld, part.p);
    const double reltol = adi.reltol >= 0 ? adi.reltol : n * EPS;
    sx.mark(ctx, 1);
    // residual factor: Res ~ Q D Q' (band reduction, truncated at a fraction of abstol like the warm-start residual of the generic path)
    BandSpec spec;
    // tols[0] = abstol = reltol ||C_rhs||_F (adi.jl:61-62), tols[1] = truncation tolerance of the residual compression, tols[2] = ||C_rhs||_F:
    // computed by the reduction's control-block launch
    spec.tol_parts = part.p; spec.tol_nparts = nt * nt; spec.tol_reltol = reltol; spec.tol_abstol = adi.abstol; spec.tol_frac = adi.residual_abs_frac;
    spec.tols_out = tols.p;
    static const bool side_in_fetch = true;
    const bool defer_side = wctx != ctx && side_in_fetch && !side_early;      // (on ONE context the set-up's own read-backs would nest inside the reduction's: it runs first then)
    if (defer_side) { spec.extra = side_setup; spec.extra_after = ctx->side_after_panels; }
    else if (!side_early) side_setup();
    SymBand sb = sym_band_reduce(ctx, Res, adi.compress_tolfac, -1.0, tols.p + 1, &spec, part.p + (size_t)nt * nt, nt * nt);
    if (defer_side && !spec.ran) side_setup();
    // Leaving early (refusal or exception) after the side stream was set up: the caller falls back to the generic ADI on the MAIN stream
    // with the same factor cache, whose entries (stacks, dense inverses, SMW products) and op.Vt the side stream may still be touching
    // (ADVICE round 2).  Join both streams on the host before anything of this frame is released or reused.
    auto join_side = [&]() {
        if (wctx != ctx) (void)hipStreamSynchronize(wctx->stream);
        (void)hipStreamSynchronize(ctx->stream);
    };
    struct SideGuard { std::function<void()> f; bool armed = true; ~SideGuard() { if (armed) f(); } } side_guard{join_side};
    if (!co_ok) return false;
    sx.mark(ctx, 2);
    const int k = sb.J;
    DevArr<AdiState> st(ctx, 1);
    AdiState h;
    std::memset(&h, 0, sizeof(int) * 4 + sizeof(double) * 3);
    ar = AdiResult();
    ar.rhs_cols = k;
    if (k > ADI_FAST_MAX_K || (ctx->dense_x_max_k > 0 && k > ctx->dense_x_max_k)) return false;
    std::vector<Mat> keepV;
    std::vector<BufP> keepRpk;
    double init_norm = 0.0;
    Mat Vall, Wall;
    int acc_total = 0;
    std::vector<double> coef;
    if (k > 0) {
        Mat R = spec.hit ? spec.B : sym_band_basis(ctx, sb);        // predicted rank: the basis was enqueued during the read-back
        Mat Tm = sb.D;
        DevArr<double> nws(ctx, ADI_FAST_NWS);
        static const bool packed_r = true;
        int mode0 = 0, nt0 = 0;
        adi_fast_pick(n, k, &mode0, &nt0);
        const bool use_pk = packed_r && mode0 == 0;
        const size_t rpd = adi_fast_rpack_doubles(n, k);
        DevArr<double> Rp0(ctx, use_pk ? rpd : 1);                  // the initial residual in the fast chain's B-operand order (slot 0 of the first chunk)
        const int pk_ct = (k + 15) / 16, pk_nblk = use_pk ? 4 * adi_fast_nstrip(n) * pk_ct : 0;
        hipLaunchKernelGGL(k_adi_init_state, dim3(1 + (pk_nblk + 3) / 4), dim3(256), 0, ctx->stream, k, (const double*)Tm.p, Tm.ld, (const double*)tols.p, adi.maxiters, st.p, nws.p,
                           ADI_FAST_NWS, n, pk_ct, pk_nblk, (const double*)R.p, R.ld, Rp0.p);
        if (wctx != ctx) DRE_HIP(hipStreamWaitEvent(ctx->stream, ctx->side_e2, 0));
        sx.mark(ctx, 3);
        const int nstrip = adi_fast_nstrip(n), kst = adi_fast_kst(n);
        Mat Gm(ctx, k * k, 2);
        // the whole solve is enqueued at once (one more iteration than the previous step needed); further chunks only if that was not enough
        int iters_host = 0;
        size_t cyc = 0;
        bool finished = false;
        const int cap = std::max(1, adi.maxiters);
        const int ctk = (k + 15) / 16;
        const int gsz = co.group_g;
        const bool grp = gsz >= 2 && use_pk && k <= ADI_GROUP_MAX_K && gsz * ctk * ((ctk + 3) / 4) <= ADI_FAST_NWS - 1 && !co.gpack.empty();
        Vall = Mat(ctx, n, k * (std::min(cap, std::max(adi.compression_interval, sx.hint + 1) + 64) + (grp ? gsz : 0)));
        int vcols_used = 0;
        // specx: every chunk's update of X is enqueued during its read-back (the batched product takes its descriptors as kernel arguments,
        // 48 at most; chunks never grow within a solve)
        const bool specx = std::min(std::max(adi.compression_interval, sx.hint + 1), adi.maxiters) + (grp ? gsz : 0) <= 48;
        bool any_plain = false;
        while (!finished) {
            const int base_it = iters_host;
            const size_t cyc0 = cyc;
            int nit = std::min(std::max(adi.compression_interval, sx.hint + 1), adi.maxiters - iters_host);
            nit = std::min(std::min(nit, 480), (Vall.cols - vcols_used) / k);        // (< 512: the device keeps the norm history as a ring)
            if (nit <= 0) break;
            // group chain (first chunk of a solve: it starts at position 0 of the cycle): whole groups of gsz iterations per launch; the
            // iteration count of the previous time step (counts fall from step to step) rounded up to a multiple of gsz is enqueued
            const bool grp_now = grp && base_it == 0 && cyc == 0;
            if (grp_now) {
                const int want = std::min(std::min(std::max(sx.hint, 1), adi.maxiters), 480);
                nit = ((want + gsz - 1) / gsz) * gsz;
                nit = std::min(nit, ((Vall.cols - vcols_used) / k / gsz) * gsz);
            }
            Mat Rring(ctx, n, k * nit);
            keepV.push_back(Rring);
            if (grp_now && nit >= gsz) {
                const int NG = nit / gsz, ncyc = (int)adi.shifts.values.size();
                DevArr<double> Rpk(ctx, rpd * (size_t)(nit + 1));
                Mat Gg(ctx, k * k, 2 * gsz);
                AdiGroupArgs ga;
                std::memset(&ga, 0, sizeof(ga));
                ga.n = n; ga.k = k; ga.nstrip = nstrip; ga.kst = kst; ga.g = gsz;
                ga.ldr = Rring.ld; ga.rpd = rpd; ga.ldv = Vall.ld;
                ga.T = Tm.p; ga.ldt = Tm.ld; ga.alpha = 1.0; ga.st = st.p; ga.nws = nws.p;
                double by1 = 0.0, fl1 = 0.0;
                ga.do_strips = 1; ga.n_prev = gsz;
                adi_group_cost(ga, &by1, &fl1);
                {
                    TimedScope chain_ts(ctx, "adi_group_iter", by1 * NG, fl1 * NG, NG + 2);
                    for (int L = 0; L <= NG + 1; ++L) {
                        ga.do_strips = L < NG ? 1 : 0;
                        if (L < NG) {
                            ga.Gpack = co.gpack[(size_t)(((L * gsz) % ncyc) / gsz)];
                            ga.Rpc = L == 0 ? Rp0.p : Rpk.p + (size_t)(L * gsz) * rpd;
                            ga.Rring = Rring.p + (size_t)(L * gsz) * k * Rring.ld;
                            ga.Rpk = Rpk.p + (size_t)(L * gsz + 1) * rpd;
                            ga.V = Vall.p + (size_t)(vcols_used + L * gsz * k) * Vall.ld;
                        }
                        // Gram matrices of the residuals launch L - 1 produced; norms + decisions for those of launch L - 2
                        ga.n_prev = (L >= 1 && L <= NG) ? gsz : 0;
                        ga.Rp_prev = ga.n_prev ? Rpk.p + (size_t)((L - 1) * gsz + 1) * rpd : nullptr;
                        ga.G_prev = Gg.p + (size_t)(L & 1) * gsz * k * k;
                        ga.n_prev2 = L >= 2 ? gsz : 0;
                        ga.G_prev2 = Gg.p + (size_t)((L - 1) & 1) * gsz * k * k;
                        ga.it0_prev2 = base_it + (L - 2) * gsz + 1;
                        adi_group_iter(ctx, ga);
                    }
                }
                for (int j = 0; j < nit; ++j) {
                    const std::complex<double> mu = adi.shifts.values[cyc % adi.shifts.values.size()];
                    ar.shifts.push_back(mu);
                    coef.push_back(-2.0 * mu.real());
                    ++cyc; ++iters_host;
                }
                keepRpk.push_back(Rpk.buf);
                sx.mark(ctx, 4);
            } else {
            if (!co.single_built) { co.build_single(ctx); co.single_built = true; }
            AdiFastArgs a;
            std::memset(&a, 0, sizeof(a));
            a.n = n; a.k = k; a.nstrip = nstrip; a.kst = kst; adi_fast_pick(n, k, &a.mode, &a.nt);
            DevArr<double> Rpk(ctx, use_pk ? rpd * (size_t)(nit + 1) : 1);
            const double* slot0 = Rp0.p;                        // first chunk: packed by the init launch
            if (use_pk && base_it > 0) { adi_fast_pack_r(ctx, n, k, R.p, R.ld, Rpk.p, st.p); slot0 = Rpk.p; }
            a.T = Tm.p; a.ldt = Tm.ld; a.tdiag = 0; a.alpha = 1.0; a.st = st.p; a.nws = nws.p;
            a.chain_timed = 1; a.do_strips = 1; a.G_prev = Gm.p;
            double by1 = 0.0, fl1 = 0.0;
            adi_fast_cost(a, &by1, &fl1);
            auto chain_ts = std::make_unique<TimedScope>(ctx, "adi_fast_iter", by1 * nit, fl1 * nit, nit + 2);   // the two flush launches ride along (riders only)
            for (int j = 1; j <= nit; ++j) {
                const std::complex<double> mu = adi.shifts.values[cyc % adi.shifts.values.size()];
                a.Apack = co.pack[cyc % co.pack.size()];
                if (j == 1) { a.Rcur = R.p; a.ldr = R.ld; } else { a.Rcur = Rring.p + (size_t)(j - 2) * k * Rring.ld; a.ldr = Rring.ld; }
                a.Rnext = Rring.p + (size_t)(j - 1) * k * Rring.ld; a.ldr_next = Rring.ld;
                a.Rpc = use_pk ? (j == 1 ? slot0 : Rpk.p + (size_t)(j - 1) * rpd) : nullptr;
                a.Rpn = use_pk ? Rpk.p + (size_t)j * rpd : nullptr;
                Mat Vj = Vall.colsview(vcols_used + (j - 1) * k, k);
                a.V = Vj.p; a.ldv = Vj.ld;
                a.two_mu = 2.0 * mu.real();
                const int g = base_it + j;
                a.G_prev = j >= 2 ? Gm.p + (size_t)((g - 1) & 1) * k * k : nullptr;
                a.G_prev2 = j >= 3 ? Gm.p + (size_t)((g - 2) & 1) * k * k : nullptr;
                a.it_prev2 = g - 2; a.do_strips = 1;
                adi_fast_iter(ctx, a);
                ar.shifts.push_back(mu);
                coef.push_back(-2.0 * mu.real());
                ++cyc; ++iters_host;
            }
            {
                const int g = base_it + nit;
                a.do_strips = 0; a.Apack = nullptr; a.Rnext = nullptr; a.V = nullptr;
                a.Rpc = use_pk ? Rpk.p + (size_t)nit * rpd : nullptr; a.Rpn = nullptr;
                a.Rcur = Rring.p + (size_t)(nit - 1) * k * Rring.ld; a.ldr = Rring.ld;
                a.G_prev = Gm.p + (size_t)(g & 1) * k * k;
                a.G_prev2 = nit >= 2 ? Gm.p + (size_t)((g - 1) & 1) * k * k : nullptr;
                a.it_prev2 = g - 1;
                adi_fast_iter(ctx, a);
                a.G_prev = nullptr;
                a.G_prev2 = Gm.p + (size_t)(g & 1) * k * k;
                a.it_prev2 = g;
                adi_fast_iter(ctx, a);
            }
            chain_ts.reset();
            sx.mark(ctx, 4);
            }
            {
                // control block (header + the norms of this chunk), tolerances and the SMW breakdown flag in ONE read-back.  The update
                //   X <- X + sum_j (-2 mu_j) V_j T V_j'      (adi.jl:166-174 accumulated: one batched product + one GEMM)
                // of this chunk is enqueued right behind the read-back kernel: how many of the speculatively enqueued iterations count is
                // read from the control block ON THE DEVICE (DevCount), so the device works on X while the host waits for the words.
                const size_t stb = sizeof(int) * 4 + sizeof(double) * (2 + (size_t)std::min(512, base_it + nit + 1 + 8));
                long long serr8 = 0;
                std::function<void()> between;
                if (specx) between = [&]() {
                    DevCount dc{st.p, base_it, nit, k};
                    Mat Wc(ctx, n, k * nit);
                    std::vector<GemmBatchDesc> descs;
                    for (int j = 0; j < nit; ++j) {
                        Mat Vj = Vall.colsview(vcols_used + j * k, k), Wj = Wc.colsview(j * k, k);
                        const double cj = -2.0 * adi.shifts.values[(cyc0 + (size_t)j) % adi.shifts.values.size()].real();
                        descs.push_back({Vj.p, Tm.p, Wj.p, nullptr, cj, n, k, k, Vj.ld, Tm.ld, Wj.ld, 0});
                    }
                    DevCount dcb = dc; dcb.per = 1;
                    gemm_batched(ctx, descs, "gemm_xupdate", dcb);
                    Mat Vch = Vall.colsview(vcols_used, k * nit);
                    gemm_sym_update(ctx, Wc, Vch, sx.X, "gemm_xupdate", dc);
                };
                ctx_fetch_overlap(ctx, between, st.p, stb, &sx.land->st, tols.p, 4 * sizeof(double), sx.land->tols, m ? (const void*)co.serr8.p : nullptr, m ? 8 : 0, &serr8);
                sx.land->serr = (int)serr8;
            }
            h = sx.land->st;
            const int acc_it = std::min(std::max(h.iters - base_it, 0), nit);
            for (int j = 1; j <= acc_it; ++j) { ar.norms.push_back(h.norms[(base_it + j) & 511]); ar.norm_iters.push_back(base_it + j); }
            if (base_it == 0) init_norm = h.norms[0];
            if (!specx && acc_it > 0) any_plain = true;
            iters_host = base_it + acc_it;
            vcols_used += acc_it * k;
            cyc = cyc - nit + acc_it;
            ar.shifts.resize(iters_host); coef.resize(iters_host);
            if (acc_it > 0) R = Rring.colsview((acc_it - 1) * k, k);
            if (h.done || acc_it < nit || iters_host >= adi.maxiters) finished = true;
        }
        acc_total = iters_host;
        for (auto& f : co.fe) if (!f->checked) { f->growth = mf_check(ctx, f->f); f->checked = true; }
        if (sx.land->serr) throw Error(ERR_SINGULAR, "SMW: capacitance matrix is singular");
        // chunks whose update was not enqueued during the read-back (more than 48 iterations at once): all increments in one go
        if (acc_total > 0 && any_plain) {
            Wall = Mat(ctx, n, k * acc_total);
            std::vector<GemmBatchDesc> descs;
            for (int j = 0; j < acc_total; ++j) {
                Mat Vj = Vall.colsview(j * k, k), Wj = Wall.colsview(j * k, k);
                descs.push_back({Vj.p, Tm.p, Wj.p, nullptr, coef[j], n, k, k, Vj.ld, Tm.ld, Wj.ld, 0});
            }
            gemm_batched(ctx, descs, "gemm_xupdate");
            Mat Vacc = Vall.colsview(0, k * acc_total);
            gemm_sym_update(ctx, Wall, Vacc, sx.X, "gemm_xupdate");
        }
    } else {
        // zero residual: read the tolerances back for the record
        DRE_HIP(hipMemcpyAsync(sx.land->tols, tols.p, 4 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        DRE_HIP(hipStreamSynchronize(ctx->stream));
        if (wctx != ctx) DRE_HIP(hipStreamWaitEvent(ctx->stream, ctx->side_e2, 0));
    }
    ar.abstol = sx.land->tols[0];
    ar.iters = acc_total;
    ar.initial_norm = k > 0 ? init_norm : 0.0;
    ar.res_norm = k > 0 ? (acc_total > 0 ? h.res_norm : init_norm) : 0.0;
    ar.norms.insert(ar.norms.begin(), ar.initial_norm); ar.norm_iters.insert(ar.norm_iters.begin(), 0);
    ar.converged = ar.res_norm <= ar.abstol;
    if (!ar.converged) ar.warnings |= 1;
    sx.hint = acc_total;
    cache->iters_hint = acc_total;
    sx.mark(ctx, 5);
    // feedback of the new X:  P1 = E' X,  K' = P1 B      (lowrank_ros1.jl:53-56)
    spmm(ctx, P, P.valEt.p, sx.X, sx.P1, 1.0, 0.0, nullptr, &sx.P1t);
    Mat Kt(ctx, n, m);
    gemm_thin(ctx, false, n, m, n, 1.0, sx.P1.p, sx.P1.ld, prob.B.p, prob.B.ld, 0.0, Kt.p, Kt.ld);
    sx.Kt = Kt;
    sx.mark(ctx, 6);
    side_guard.armed = false;          // the main stream waited for side_e2 above: nothing of the side stream is pending
    return true;
}

static uint64_t tag_of(int order, double tau) {
    uint64_t bits;
    std::memcpy(&bits, &tau, sizeof(bits));
    return bits * 1315423911ull + (uint64_t)order * 0x9E3779B97F4A7C15ull + 1;
}

// LDL' form of a dense symmetric X: compress!(lowrank(I, X))  (LDLt.jl:204-225; S = I X I' is X itself)
static LDLtP dense_to_ldlt(Ctx* ctx, int n, const Mat& Xd, double ctf) {
    Mat I(ctx, n, n), D(ctx, n, n);
    set_identity(ctx, I, 1.0);
    copy_mat(ctx, Xd, D);
    LDLtP X = ldlt_make(ctx, n, I, D, 1.0, false);
    ldlt_compress(ctx, *X, ctf, false);
    return X;
}


// =============================================================================================
// Rosenbrock-1 time loop with the RESIDUAL RECURRENCE (round 4; general path: multifrontal solves, Cyclic real shifts).
// Between two time steps the reference (lowrank_ros1.jl:35-60) compresses X (LDLt.jl:204-225), forms the feedback K = B'XE, the right-hand side
// and the warm-start residual [G, E'L, F'L] from that compressed X (lyapunov/residual.jl:3-31) and compresses the residual: at n = 5177 that is
// 4.7 of the 6 ms of a time step, all of it short dependent kernels, with the ADI iteration itself at 1.3 ms.  But step i's ADI already holds
// everything step i + 1 needs.  With X_i = X_{i-1} + sum_j V_j (c_j T) V_j', c_j = -2 mu_j alpha, and the recurrence R_j = R_{j-1} - 2 mu_j E'V_j
// (adi.jl:166-177):
//     E'V_j = (R_{j-1} - R_j) / (2 mu_j)                                  — E' times every increment, without touching E or V
//     K_i   = K_{i-1} + sum_j (B'V_j) (c_j T) (E'V_j)'                    — the feedback (lowrank_ros1.jl:53-57)
//     Res_{i+1}(X_i) = alpha R_J T R_J'  -  dK' dK  +  (1/tau) sum_j (E'V_j) (c_j T) (E'V_j)',      dK = K_i - K_{i-1}
// the last line because F_{i+1} = F_i - B dK and rhs_{i+1} - rhs_i = K_i'K_i - K_{i-1}'K_{i-1} + E'(X_i - X_{i-1})E/tau, and the cross terms with
// K_i = B'X_iE collapse to -dK'dK.  (alpha R_J T R_J' is step i's residual at its last iterate: the ADI invariant.)  So the critical path of a
// time step is: compress that factored residual (the same factor-form reduction as before, on [R_J, dK', E'V_1 .. E'V_J]) -> ADI.  X itself is
// only an output: its compression (adi.jl:78-80, lowrank_ros1.jl:53) runs on the SIDE stream, driven by a parked host thread, beside the next
// step, and the one thing the next solve needs from it — the tolerance reltol ||rhs_{i+1}||_F (adi.jl:61-62) — arrives in device memory
// and is applied to the recorded norms at the end of the solve's first chunk (adi_advance, deferred decisions).
// Same iterates as the reference's loop up to rounding: the identity is exact; what differs is that the truncation error of compressing X
// (4 eps ||X||) no longer re-enters the next residual.  A step whose ADI leaves the fan path (complex shift, user solver, in-loop compression)
// falls back to the reference's order for the next step.
// =============================================================================================
static bool ros1_recurrence_ok(Ctx* ctx, const GdreProblem& prob, int order, const AdiOptions& adi) {
    static const bool env_on = true;
    const int n = prob.P->n;
    if (!env_on || !ctx->ros1_recurrence || order != 1 || adi.compress_exact || adi.inner_solve || adi.ignore_initial_guess || !adi.compression) return false;
    if (adi.shifts.kind != ShiftSpec::CYCLIC || adi.shifts.values.empty() || adi.abstol >= 0.0) return false;
    for (auto& mu : adi.shifts.values) if (mu.imag() != 0.0) return false;
    if (n <= ctx->dense_inv_max_n || n < ctx->compress_factor_min_n || ctx->adi_fan < 2 || !prob.P->use_mfma_sweeps) return false;
    return true;
}
// state of X as the side stream holds it: compressed up to time step `step` (one block), E' times its factor, completion event
struct SideState { int step = 0; LDLtP X; Mat EtL; hipEvent_t ev = nullptr; };
// what step s added to  E'XE / tau_{s+1}:  Q Dq Q' - aT-weighted R_J R_J' + dK'dK   (the compressed warm-start residual of step s + 1, the final
// residual factor of step s, the change of the feedback) — kept until the side stream's X includes step s
struct StepDelta { int s; double tau; Mat Q, Dq, Rj, Tj; double aj; Mat dKt; };

static void ros1_recurrence_loop(Ctx* ctx, const GdreProblem& prob, double dt, bool save_state, const AdiOptions& adi, int nsteps, GdreResult& out,
                                 FactorCache& cache, const Feedback& fb0) {
    (void)dt;
    const Pencil& P = *prob.P;
    const int n = P.n, q = prob.Ct.cols, m = prob.B.cols;
    const double ctf = adi.compress_tolfac;
    // side context (own stream, pool, hints) + the parked thread that drives it
    if (!ctx->side) {
        auto sc = std::make_unique<Ctx>();
        sc->device = ctx->device; sc->num_cus = ctx->num_cus;
        sc->stream = create_stream(1);
        DRE_HIP(hipEventCreateWithFlags(&ctx->side_e1, hipEventDisableTiming));
        DRE_HIP(hipEventCreateWithFlags(&ctx->side_e2, hipEventDisableTiming));
        sc->timer = std::make_unique<KernelTimer>();
        sc->timer->enabled = ctx->prof_side;
        ctx->side = std::move(sc);
    }
    Ctx* const side = ctx->side.get();
    side->dense_inv_max_n = ctx->dense_inv_max_n; side->compress_direct_max_n = ctx->compress_direct_max_n;
    side->compress_direct_ratio = ctx->compress_direct_ratio; side->compress_factor_min_n = ctx->compress_factor_min_n;
    side->compress_factor_min_cols = ctx->compress_factor_min_cols; side->compress_sketch = ctx->compress_sketch;
    side->compress_sketch_min_cols = ctx->compress_sketch_min_cols; side->compress_sketch_extra = ctx->compress_sketch_extra;
    side->compress_sketch_ratio = ctx->compress_sketch_ratio; side->compress_sketch_sparse = ctx->compress_sketch_sparse;
    side->compress_sketch_cholqr = ctx->compress_sketch_cholqr;
    side->fetch_spin = false;
    side->orthf_fn = ctx->orthf_fn; side->orthf_user = ctx->orthf_user;
    SideWorker worker;
    // events: a ring (at most one job is in flight; a slot is reused eight jobs later)
    hipEvent_t ring[16];
    for (auto& e : ring) DRE_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    struct EvGuard { hipEvent_t* v; ~EvGuard() { for (int i = 0; i < 16; ++i) (void)hipEventDestroy(v[i]); } } evguard{ring};
    long njobs = 0;
    std::mutex smu;
    auto cur = std::make_shared<SideState>();
    cur->step = 0; cur->X = prob.X0; cur->EtL = fb0.EtL;
    auto get_state = [&]() { std::lock_guard<std::mutex> lk(smu); return cur; };
    std::vector<LDLtP> saved((size_t)nsteps + 1);
    int xwarm_sx = 32, xwarm_strikes = 0;           // warm-started compression of X on the side stream (touched by the worker only)
    std::vector<LBlock> pend;                       // increments the side stream has not been handed yet
    int pend_upto = 0;
    bool job_pending = false;
    static const bool rec_timing = env_trace("rec");
    double t_join = 0.0, t_solve = 0.0, t_tail = 0.0; long n_join = 0, n_jobs_t = 0;
    auto now = []() { return std::chrono::steady_clock::now(); };
    auto us = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
    auto join_side = [&]() {                        // host: the running job is finished (its read-backs are synchronous) and its state published
        if (!job_pending) return;
        const auto a = now();
        worker.wait();
        if (rec_timing) { t_join += us(a, now()); ++n_join; }
        job_pending = false;
    };
    struct JoinGuard { SideWorker& w; bool& p; ~JoinGuard() { if (p) { try { w.wait(); } catch (...) {} } } } jguard{worker, job_pending};
    // hand everything pending to the side stream: X_upto = compress(X_base + increments), E' times its factor
    auto submit_job = [&]() {
        if (pend.empty() && pend_upto == get_state()->step) return;
        join_side();
        hipEvent_t e_main = ring[(2 * njobs) % 16], e_side = ring[(2 * njobs + 1) % 16];
        ++njobs; ++n_jobs_t;
        DRE_HIP(hipEventRecord(e_main, ctx->stream));
        const auto base = get_state();
        const std::vector<LBlock> blocks = pend;
        const int target = pend_upto;
        pend.clear();
        LDLtP* const saved_slot = save_state ? &saved[(size_t)target] : nullptr;
        auto* curp = &cur; auto* mup = &smu;
        int* const xw_sx = &xwarm_sx; int* const xw_strikes = &xwarm_strikes;
        worker.submit([=, &P]() {
            DRE_HIP(hipSetDevice(side->device));
            DRE_HIP(hipStreamWaitEvent(side->stream, e_main, 0));
            if (base->ev) DRE_HIP(hipStreamWaitEvent(side->stream, base->ev, 0));
            auto Xs = std::make_shared<LDLt>();
            Xs->n = n;
            for (auto& b : base->X->blocks) if (b.L.cols > 0) Xs->blocks.push_back(b);
            for (auto& b : blocks) Xs->blocks.push_back(b);
            if (Xs->blocks.empty()) Xs->blocks.push_back({Mat(side, n, 0), Mat(side, 0, 0), 1.0, true});
            // X_b's basis nearly spans the new X (the solution moves slowly between time steps): the warm-started range finder of the residual
            // compression, with the relative tolerance of compress! as an absolute one (||X_b||_F = ||D_b||_F: L_b is orthonormal); the full
            // sketch compression where the probe rejects it
            bool done = false;
            if (!blocks.empty() && base->X->blocks.size() == 1 && base->X->blocks[0].ortho && base->X->blocks[0].L.cols >= 64 && *xw_strikes < 2) {
                const LBlock& bb = base->X->blocks[0];
                const double nb = frob_norm_host(side, bb.D) * std::fabs(bb.alpha);
                double missed = 0.0;
                done = nb > 0.0 && warm_compress(side, *Xs, bb.L, ctf, ctf * EPS * nb, *xw_sx, &missed, 64.0 * EPS);
                if (done) *xw_strikes = 0;
                else if (*xw_sx < 64) *xw_sx = 64;
                else *xw_strikes += 1;
            }
            if (!done) ldlt_destructure(side, *Xs, ctf, false);
            auto st = std::make_shared<SideState>();
            st->step = target; st->X = Xs; st->ev = e_side;
            const LBlock& b = Xs->blocks[0];
            st->EtL = Mat(side, n, b.L.cols);
            if (b.L.cols > 0) spmm(side, P, P.valEt.p, b.L, st->EtL, 1.0, 0.0);
            DRE_HIP(hipEventRecord(e_side, side->stream));
            DRE_HIP(hipStreamSynchronize(side->stream));       // the increments' buffers go back to the MAIN pool when this closure dies
            if (saved_slot) *saved_slot = Xs;
            { std::lock_guard<std::mutex> lk(*mup); *curp = st; }
        });
        job_pending = true;
    };

    std::map<uint64_t, DevArr<double>> valF_by_tau;
    Mat Kt = fb0.Kt;                                // K(t_{i-1})'
    bool have_hist = false;
    AdiResult prev;                                 // pieces of the previous solve the recurrence needs (hist, Tm, alpha_res, residual)
    Mat prev_dKt;
    double abstol_prev = -1.0;
    std::vector<StepDelta> deltas;                  // steps the side stream's X does not include yet
    DevArr<double> normC_dev(ctx, 1);
    Mat Im(ctx, m, m);
    set_identity(ctx, Im, 1.0);
    for (int i = 1; i <= nsteps; ++i) {
        const double tau = out.t[i - 1] - out.t[i];
        GaleOperator op;
        op.P = &P;
        op.tag = tag_of(1, tau);
        auto it = valF_by_tau.find(op.tag);
        if (it == valF_by_tau.end()) {
            DevArr<double> v(ctx, P.nnz);
            vals_axpby(ctx, P.nnz, 1.0, P.valAt.p, -1.0 / (2.0 * tau), P.valEt.p, v.p);      // A - E/(2 tau)   (lowrank_ros1.jl:39)
            it = valF_by_tau.emplace(op.tag, v).first;
        }
        op.valFt = it->second; op.cA = 1.0; op.cE = -1.0 / (2.0 * tau);
        op.has_lr = true; op.U = prob.B; op.Vt = Kt; op.alpha = -1.0;
        AdiOptions a2 = adi;
        a2.final_compress = false; a2.keep_history = true;
        AdiResult ar;
        std::vector<LBlock> incr;
        const auto ts0 = now();
        if (!have_hist) {
            // the reference's order (first step, or after a solve without history): compressed X -> feedback pieces -> right-hand side -> ADI
            if (!pend.empty() || pend_upto != get_state()->step) submit_job();
            join_side();
            const auto st = get_state();
            if (st->ev) DRE_HIP(hipStreamWaitEvent(ctx->stream, st->ev, 0));
            deltas.clear();
            LDLtP X = st->X;
            Feedback fb = feedback(ctx, prob, *X, ctf, false);
            const int r = fb.L.cols;
            Mat G(ctx, n, q + r);
            { Mat d = G.colsview(0, q); copy_mat(ctx, prob.Ct, d); }
            { Mat d = G.colsview(q, r); copy_mat(ctx, fb.EtL, d); }
            Mat S(ctx, q + r, q + r);
            set_identity(ctx, S, 0.0);
            { Mat d = S.view(0, 0, q, q); set_identity(ctx, d, 1.0); }
            if (r > 0) {
                Mat d = S.view(q, q, r, r);
                copy_mat(ctx, fb.D, d, fb.alpha / tau);
                gemm(ctx, true, false, 1.0, fb.BtLD, fb.BtLD, 1.0, d);
            }
            LDLtP rhs = ldlt_make(ctx, n, G, S, 1.0, false);
            if (X->iszero()) ldlt_compress(ctx, *rhs, ctf, false);
            const size_t nb_prev = X->blocks.size();
            ar = adi_solve(ctx, op, *rhs, X, a2, &cache);
            bool intact = ar.X->blocks.size() >= nb_prev;
            for (size_t bi = 0; intact && bi < nb_prev; ++bi) intact = ar.X->blocks[bi].L.p == X->blocks[bi].L.p;
            if (intact) incr.assign(ar.X->blocks.begin() + (long)nb_prev, ar.X->blocks.end());
            else {
                // the solve compressed in between: its X stands (published as the side state of this step; nothing pending)
                ar.hist_ok = false;
                auto st2 = std::make_shared<SideState>();
                st2->step = i; st2->X = ar.X;
                { std::lock_guard<std::mutex> lk(smu); cur = st2; }
                pend_upto = i;
            }
        } else {
            // the recurrence: [R_J, dK', E'V_1 .. E'V_J] with their inner blocks IS the warm-start residual of this step
            auto resid = std::make_shared<LDLt>();
            resid->n = n;
            Mat RJ;                                   // final residual factor of the previous solve
            if (prev.residual && !prev.residual->blocks.empty() && prev.residual->blocks[0].L.cols > 0) RJ = prev.residual->blocks[0].L;
            if (RJ.cols > 0) resid->blocks.push_back({RJ, prev.Tm, prev.alpha_res, prev.tdiag, false});
            if (!prev.hist.empty()) {
                resid->blocks.push_back({prev_dKt, Im, -1.0, true, false});
                const int k = prev.Tm.rows;
                for (auto& hc : prev.hist) {
                    const int J = (int)hc.mu.size();
                    Mat EV(ctx, n, J * k);
                    ev_from_residuals(ctx, n, k, J, hc.R0, hc.Rs, EV, hc.mu.data());
                    for (int j = 0; j < J; ++j)
                        resid->blocks.push_back({EV.colsview(j * k, k), prev.Tm, -2.0 * hc.mu[(size_t)j] * prev.alpha_res / tau, prev.tdiag, false});
                }
            }
            a2.given_residual = resid;
            a2.abstol_lag = abstol_prev;
            if (!prev.hist.empty()) a2.warm_basis = prev.hist[0].R0;      // the compressed residual the previous solve started from: orthonormal
            a2.normC_dev = normC_dev.p;
            // ||rhs_i||_F for the tolerance (adi.jl:61-62), on this stream, as soon as the residual is compressed:
            //   rhs_i = C'C + K'K + E'X_b E / tau + sum_{s = b+1 .. i-1} (tau_{s+1} / tau) (Q_s Dq_s Q_s' - a_s R_s T_s R_s' + dK_s'dK_s)
            // with X_b the latest X the side stream has finished (b >= i - 4) — one Gram matrix of a few hundred columns
            a2.normC_build = [&, i, tau, RJ](Ctx* hc, const Mat& Q, const Mat& Dq, double aq) {
                StepDelta dl;
                dl.s = i - 1; dl.tau = tau; dl.Q = Q; dl.Dq = Mat(hc, Dq.rows, Dq.cols); copy_mat(hc, Dq, dl.Dq, aq);
                dl.Rj = RJ; dl.Tj = prev.Tm; dl.aj = prev.alpha_res; dl.dKt = prev.hist.empty() ? Mat() : prev_dKt;
                deltas.push_back(dl);
                auto st = get_state();
                if ((i - 1) - st->step > 5) { join_side(); st = get_state(); }
                if (st->ev) DRE_HIP(hipStreamWaitEvent(hc->stream, st->ev, 0));
                while (!deltas.empty() && deltas.front().s <= st->step) deltas.erase(deltas.begin());
                const LBlock& xb = st->X->blocks[0];
                const int r = xb.L.cols;
                int cols = q + m + r;
                for (auto& d : deltas) cols += d.Q.cols + d.Rj.cols + (d.dKt.cols > 0 ? m : 0);
                Mat F(hc, n, cols), S(hc, cols, cols);
                fill_mat(hc, S, 0.0);
                std::vector<CopyDesc> cd;
                int off = 0;
                auto put = [&](const Mat& L, const Mat* D, double scale, bool identity) {
                    if (L.cols == 0) return;
                    Mat dst = F.colsview(off, L.cols);
                    cd.push_back({L.p, dst.p, n, L.cols, L.ld, dst.ld});
                    Mat ds = S.view(off, off, L.cols, L.cols);
                    if (identity) set_identity(hc, ds, scale); else copy_mat(hc, *D, ds, scale);
                    off += L.cols;
                };
                put(prob.Ct, nullptr, 1.0, true);
                put(Kt, nullptr, 1.0, true);
                put(st->EtL, &xb.D, xb.alpha / tau, false);
                for (auto& d : deltas) {
                    const double sc = d.tau / tau;
                    put(d.Q, &d.Dq, sc, false);
                    put(d.Rj, &d.Tj, -sc * d.aj, false);
                    if (d.dKt.cols > 0) put(d.dKt, nullptr, sc, true);
                }
                copy_batched(hc, cd);
                ldlt_norm_device(hc, F, S, 1.0, normC_dev.p);
            };
            LDLt none; none.n = n;
            ar = adi_solve(ctx, op, none, nullptr, a2, &cache);
            for (auto& b : ar.X->blocks) if (b.L.cols > 0) incr.push_back(b);
        }
        out.adi_iters += ar.iters;
        const auto ts1 = now();
        if (rec_timing) t_solve += us(ts0, ts1);
        size_t hist_its = 0;
        for (auto& hc : ar.hist) hist_its += hc.mu.size();
        const bool hist = ar.hist_ok && hist_its == (size_t)ar.iters && (int)incr.size() == ar.iters;
        // the increments go to the side stream (a job per step with save_state; otherwise whenever the previous job is done)
        if (pend_upto < i) { for (auto& b : incr) pend.push_back(b); pend_upto = i; }
        Mat Kt_new;
        if (hist) {
            // side jobs: every step with save_state (every X(t) is an output); otherwise when the previous one is done AND `batch` steps have
            // gathered (the tolerance formula above tolerates a lag of 3 steps; one compression of 2-3 steps' increments costs little more
            // than one step's — its latency chains depend on the rank, not on the number of columns)
            static const int batch = 2;
            if (save_state || (!worker.pending() && i - get_state()->step >= batch) || i == nsteps) submit_job();
            // K_i' = K_{i-1}' + sum_j (E'V_j) (c_j T) (V_j'B)
            Mat dKt(ctx, n, m);
            Kt_new = Mat(ctx, n, m);
            copy_mat(ctx, Kt, Kt_new);
            const int k = ar.Tm.rows;
            bool first = true;
            for (auto& hc : ar.hist) {
                const int J = (int)hc.mu.size();
                Mat EV(ctx, n, J * k), VtB(ctx, J * k, m), Mx(ctx, J * k, m);
                ev_from_residuals(ctx, n, k, J, hc.R0, hc.Rs, EV, hc.mu.data());
                gemm(ctx, true, false, 1.0, hc.Vs, prob.B, 0.0, VtB, nullptr, "gemm_feedback");
                std::vector<GemmBatchDesc> descs;
                for (int j = 0; j < J; ++j)
                    descs.push_back({ar.Tm.p, VtB.p + (size_t)j * k, Mx.p + (size_t)j * k, nullptr, -2.0 * hc.mu[(size_t)j] * ar.alpha_res, k, m, k, ar.Tm.ld, VtB.ld, Mx.ld, 0});
                gemm_batched(ctx, descs, "gemm_feedback");
                gemm(ctx, false, false, 1.0, EV, Mx, first ? 0.0 : 1.0, dKt, nullptr, "gemm_feedback");
                first = false;
            }
            if (first) fill_mat(ctx, dKt, 0.0);
            else { const size_t tot = (size_t)n * m; hipLaunchKernelGGL(k_axpy_inplace, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, ctx->stream, tot, 1.0, (const double*)dKt.p, Kt_new.p); }
            prev_dKt = dKt;
        } else {
            // no history (an iteration outside the fan path, a compression inside the solve): the reference's order for this step's tail
            submit_job();
            join_side();
            const auto st = get_state();
            if (st->ev) DRE_HIP(hipStreamWaitEvent(ctx->stream, st->ev, 0));
            LDLtP X = st->X;
            if (X.get() == prob.X0.get()) X = std::make_shared<LDLt>(*prob.X0);
            Feedback fb = feedback(ctx, prob, *X, ctf, false);
            Kt_new = fb.Kt;
            if (save_state) saved[(size_t)i] = X;
            deltas.clear();
        }
        if (rec_timing) t_tail += us(ts1, now());
        abstol_prev = ar.abstol;
        have_hist = hist;
        Kt = Kt_new;
        out.Kt.push_back(Kt);
        prev = ar;                                   // (shallow: the slabs stay alive until the next residual is built)
        prev.X.reset();
        AdiResult rec = std::move(ar);
        rec.X.reset(); rec.residual.reset(); rec.hist.clear();
        out.gale.push_back(std::move(rec));
    }
    submit_job();
    join_side();
    {
        const auto st = get_state();
        if (st->ev) DRE_HIP(hipStreamWaitEvent(ctx->stream, st->ev, 0));
        DRE_HIP(hipStreamSynchronize(ctx->stream));
        if (rec_timing) std::fprintf(stderr, "[rec timing, us per step] solve (residual + ADI) %.0f | tail (feedback, job hand-over) %.0f | blocked in join %.0f (%ld joins, %ld jobs)\n",
                                     t_solve / nsteps, t_tail / nsteps, t_join / nsteps, n_join, n_jobs_t);
        if (save_state) for (int i = 1; i <= nsteps; ++i) out.X.push_back(saved[(size_t)i] ? saved[(size_t)i] : st->X);
        else out.X.push_back(st->X);
    }
}

GdreResult gdre_solve(Ctx* ctx, const GdreProblem& prob, int order, double dt, bool save_state, const AdiOptions& adi) {
    DRE_REQUIRE(order == 1 || order == 2, "only Ros1 and Ros2 have a low-rank formulation");
    DRE_REQUIRE(dt != 0.0, "dt must be nonzero");
    const Pencil& P = *prob.P;
    const int n = P.n, q = prob.Ct.cols, m = prob.B.cols;
    GdreResult out;
    const int nsteps = (int)std::floor((prob.tf - prob.t0) / dt + 1e-9);
    DRE_REQUIRE(nsteps >= 0, "tspan and dt point in opposite directions");
    for (int i = 0; i <= nsteps; ++i) out.t.push_back(prob.t0 + i * dt);
    LDLtP X = prob.X0;
    out.X.push_back(X);
    const double ctf = adi.compress_tolfac;
    const bool cex = adi.compress_exact;
    Feedback fb = feedback(ctx, prob, *X, ctf, cex);
    out.Kt.push_back(fb.Kt);
    FactorCache cache;
    if (ros1_recurrence_ok(ctx, prob, order, adi) && nsteps >= 1) {
        ros1_recurrence_loop(ctx, prob, dt, save_state, adi, nsteps, out, cache, fb);
        out.nfactor = cache.nfactor;
        return out;
    }
    // Ros1 at small n: X stays "warm start + increments" between two compressions (every xevery-th step and at the end); right-hand side,
    // feedback and warm-start residual work on the block list (the direct-form compression does not care about the number of columns)
    const bool xside_env = ctx->x_side_stream != 0;
    const int xevery = std::max(1, ctx->x_compress_every);
    const bool xblocks = order == 1 && (xevery > 1 || xside_env) && !cex && !save_state && n <= xblocks_max_n() && !adi.ignore_initial_guess;
    // Side stream: the compression of X_{i-1} is not on the critical path of step i (right-hand side, feedback and residual take the
    // block list), so it runs on a second stream, driven by a second host thread with its own context (stream, pool, hints), while the
    // main stream does the residual compression and the ADI iteration of step i; its result replaces the uncompressed summands at the
    // end of step i:  X_i = compress(X_{i-1}) + increments_i.  A single latency-bound solve leaves most of the chip idle.
    const bool xside = xblocks && xside_env;
    if (xside_env && !ctx->side) {
        auto sc = std::make_unique<Ctx>();
        sc->device = ctx->device; sc->num_cus = ctx->num_cus;
        sc->stream = create_stream(1);
        DRE_HIP(hipEventCreateWithFlags(&ctx->side_e1, hipEventDisableTiming));
        DRE_HIP(hipEventCreateWithFlags(&ctx->side_e2, hipEventDisableTiming));
        sc->timer = std::make_unique<KernelTimer>();
        sc->timer->enabled = ctx->prof_side;
        ctx->side = std::move(sc);
    }
    Ctx* const side = ctx->side.get();
    if (side) {
        side->dense_inv_max_n = ctx->dense_inv_max_n; side->compress_direct_max_n = ctx->compress_direct_max_n;
        side->compress_direct_ratio = ctx->compress_direct_ratio; side->compress_factor_min_n = ctx->compress_factor_min_n;
        side->compress_factor_min_cols = ctx->compress_factor_min_cols;
    }
    std::map<uint64_t, DevArr<double>> valF_by_tau;
    const double gamma = 1.0 + 1.0 / std::sqrt(2.0);
    // Ros1, small n, real Cyclic shifts, no save_state: from the second step on X is carried as a dense symmetric matrix (ros1_dense_step)
    bool densex = order == 1 && !cex && !save_state && !adi.inner_solve && !adi.ignore_initial_guess && adi.compression && adi.shifts.kind == ShiftSpec::CYCLIC &&
                  n <= ctx->dense_x_max_n && n <= ctx->dense_inv_max_n && m <= 32 && !adi.shifts.values.empty();
    for (auto& mu : adi.shifts.values) if (mu.imag() != 0.0) densex = false;
    DenseXState sx;
    sx.attach(ctx);
    SideWorker side_worker;        // parked thread that drives the side-stream compression of the block-list loop (created on first use)
    bool sx_init = false, x_is_dense = false;

    const bool wall_on = env_trace("phase");
    auto wall_now = [&]() { if (wall_on) DRE_HIP(hipStreamSynchronize(ctx->stream)); return std::chrono::steady_clock::now(); };
    const auto w_begin = wall_now();
    auto w_first = w_begin;
    for (int i = 1; i <= nsteps; ++i) {
        if (wall_on && i == 2) w_first = wall_now();
        const double tau = out.t[i - 1] - out.t[i];
        GaleOperator op;
        op.P = &P;
        op.tag = tag_of(order, tau);
        auto it = valF_by_tau.find(op.tag);
        if (it == valF_by_tau.end()) {
            DevArr<double> v(ctx, P.nnz);
            if (order == 1) vals_axpby(ctx, P.nnz, 1.0, P.valAt.p, -1.0 / (2.0 * tau), P.valEt.p, v.p);      // A - E/(2 tau)
            else vals_axpby(ctx, P.nnz, gamma * tau, P.valAt.p, -0.5, P.valEt.p, v.p);                         // gamma tau A - E/2
            it = valF_by_tau.emplace(op.tag, v).first;
        }
        op.valFt = it->second;
        if (order == 1) { op.cA = 1.0; op.cE = -1.0 / (2.0 * tau); } else { op.cA = gamma * tau; op.cE = -0.5; }
        op.has_lr = true;
        op.U = prob.B;
        op.Vt = fb.Kt;
        op.alpha = order == 1 ? -1.0 : 1.0 / (-gamma * tau);
        const int r = fb.L.cols;
        if (densex) {
            if (!sx_init) {
                // the current X (X0, or a block list: warm start + increments) as a dense matrix
                const int c = X->rank();
                sx.X = Mat(ctx, n, n);
                if (c > 0) {
                    Mat Lcat(ctx, n, c), LD(ctx, n, c);
                    hcat_scale_blocks(ctx, *X, Lcat, LD);
                    gemm(ctx, false, true, 1.0, LD, Lcat, 0.0, sx.X, nullptr, "gemm_xupdate");
                    symmetrize(ctx, sx.X);
                } else fill_mat(ctx, sx.X, 0.0);
                sx.P1 = Mat(ctx, n, n);
                sx.P1t = Mat(ctx, n, n);
                spmm(ctx, P, P.valEt.p, sx.X, sx.P1, 1.0, 0.0, nullptr, &sx.P1t);
                sx.Kt = fb.Kt;
                sx.hint = cache.iters_hint > 0 ? cache.iters_hint : 3 * adi.compression_interval;    // first solve: a few chunks at most
                sx_init = true;
            }
            AdiResult ar;
            if (ros1_dense_step(ctx, prob, op, tau, adi, &cache, sx, ar)) {
                out.adi_iters += ar.iters;
                out.gale.push_back(std::move(ar));
                out.Kt.push_back(sx.Kt);
                x_is_dense = true;
                continue;
            }
            // the fast chain refused (residual too wide, ill-conditioned shifted operator): back to the factored form for good
            densex = false;
            if (x_is_dense) { X = dense_to_ldlt(ctx, n, sx.X, ctf); x_is_dense = false; fb = feedback(ctx, prob, *X, ctf, false); op.Vt = fb.Kt; }
        }
        if (order == 1 && xblocks) {
            // X is a block list (compressed every `xevery` steps only):  rhs = C'C + K'K + sum_b (E'L_b) (alpha_b D_b / tau) (E'L_b)'
            // as a block list of its own; nothing is compressed before the warm-start residual (gale_residual_blocks)
            auto rhs = std::make_shared<LDLt>();
            rhs->n = n;
            Mat Iq(ctx, q, q), Im(ctx, m, m);
            set_identity(ctx, Iq, 1.0); set_identity(ctx, Im, 1.0);
            rhs->blocks.push_back({prob.Ct, Iq, 1.0, true, false});
            rhs->blocks.push_back({fb.Kt, Im, 1.0, true, false});
            int off = 0;
            for (auto& b : X->blocks) {
                const int k = b.L.cols;
                if (k == 0) continue;
                rhs->blocks.push_back({fb.EtL.colsview(off, k), b.D, b.alpha / tau, b.diag, false});
                off += k;
            }
            AdiOptions a2 = adi;
            const bool last = (i == nsteps);
            a2.final_compress = xside ? false : (last || (i % xevery == 0));
            a2.warm_L = fb.L; a2.warm_EtL = fb.EtL;            // the feedback already concatenated X and applied E'
            static const bool fold_e = true;
            if (fold_e) { a2.rhs_lead_blocks = 2; a2.rhs_e_coeff = 1.0 / tau; }     // rhs = [C'C, K'K] + E'XE / tau
            // side stream: compress the warm start X_{i-1} concurrently (only worth it once it carries increments)
            LDLtP Xc;
            const size_t nb_prev = X->blocks.size();
            bool side_job = false;
            if (xside && nb_prev > 1) {
                DRE_HIP(hipEventRecord(ctx->side_e1, ctx->stream));
                Xc = std::make_shared<LDLt>(*X);               // shallow: shares the summands, which stay alive in X until the join
                side_worker.submit([&, Xc]() {
                    DRE_HIP(hipSetDevice(side->device));
                    DRE_HIP(hipStreamWaitEvent(side->stream, ctx->side_e1, 0));
                    ldlt_compress(side, *Xc, ctf, false);
                    DRE_HIP(hipEventRecord(ctx->side_e2, side->stream));
                });
                side_job = true;
            }
            AdiResult ar;
            try { ar = adi_solve(ctx, op, *rhs, X, a2, &cache); }
            catch (...) { if (side_job) { try { side_worker.wait(); } catch (...) {} } throw; }
            if (side_job) {
                side_worker.wait();
                DRE_HIP(hipStreamWaitEvent(ctx->stream, ctx->side_e2, 0));
                // X_i = compress(X_{i-1}) + increments_i  (unless the ADI loop had to compress in between: then its X stands)
                bool intact = ar.X->blocks.size() >= nb_prev;
                for (size_t bi = 0; intact && bi < nb_prev; ++bi) intact = ar.X->blocks[bi].L.p == X->blocks[bi].L.p;
                if (intact) {
                    auto Xn = std::make_shared<LDLt>();
                    Xn->n = n;
                    Xn->blocks = Xc->blocks;
                    for (size_t bi = nb_prev; bi < ar.X->blocks.size(); ++bi) Xn->blocks.push_back(ar.X->blocks[bi]);
                    ar.X = Xn;
                }
            }
            if (xside && last) ldlt_compress(ctx, *ar.X, ctf, false);
            X = ar.X;
            out.adi_iters += ar.iters;
            ar.X.reset(); ar.residual.reset();
            out.gale.push_back(std::move(ar));
            if (save_state) out.X.push_back(X);
            fb = feedback_blocks(ctx, prob, *X);
            out.Kt.push_back(fb.Kt);
            continue;
        }
        if (order == 1) {
            // G = [C', E'L];  S = blkdiag(I_q, BtLD' BtLD + D/tau);  R = compress!(lowrank(G, S))   (lowrank_ros1.jl:42-44)
            Mat G(ctx, n, q + r);
            { Mat d = G.colsview(0, q); copy_mat(ctx, prob.Ct, d); }
            { Mat d = G.colsview(q, r); copy_mat(ctx, fb.EtL, d); }
            Mat S(ctx, q + r, q + r);
            set_identity(ctx, S, 0.0);
            { Mat d = S.view(0, 0, q, q); set_identity(ctx, d, 1.0); }
            if (r > 0) {
                Mat d = S.view(q, q, r, r);
                // E'XE / tau with X = alpha L D L': the reference writes D/tau here (lowrank_ros1.jl:43), which is only right for alpha = 1
                // (SURVEY Appendix B.3); the engine keeps alpha so that every code path solves the same equation as the dense solver
                copy_mat(ctx, fb.D, d, fb.alpha / tau);
                gemm(ctx, true, false, 1.0, fb.BtLD, fb.BtLD, 1.0, d);
            }
            LDLtP rhs = ldlt_make(ctx, n, G, S, 1.0, false);
            // The reference compresses the right-hand side here (lowrank_ros1.jl:44) and again, together with the warm start,
            // inside residual() (lyapunov/residual.jl:30).  In Krylov mode the second compression truncates at a fraction of
            // abstol, which also removes what the first one would have filtered, so the first is skipped for warm starts.
            if (cex || adi.ignore_initial_guess || X->iszero()) ldlt_compress(ctx, *rhs, ctf, cex);
            AdiResult ar = adi_solve(ctx, op, *rhs, X, adi, &cache);
            X = ar.X;
            out.adi_iters += ar.iters;
            ar.X.reset(); ar.residual.reset();
            out.gale.push_back(std::move(ar));
        } else {
            // stage 1: G = [C', A'L, E'L], S = [I 0 0; 0 0 D; 0 D -(BtLD)'BtLD]      (lowrank_ros2.jl:44-58)
            const int nG = q + 2 * r;
            Mat G(ctx, n, nG);
            { Mat d = G.colsview(0, q); copy_mat(ctx, prob.Ct, d); }
            if (r > 0) {
                Mat d = G.colsview(q, r); spmm(ctx, P, P.valAt.p, fb.L, d, 1.0, 0.0);
                Mat d2 = G.colsview(q + r, r); copy_mat(ctx, fb.EtL, d2);
            }
            Mat S(ctx, nG, nG);
            set_identity(ctx, S, 0.0);
            { Mat d = S.view(0, 0, q, q); set_identity(ctx, d, 1.0); }
            if (r > 0) {
                Mat d23 = S.view(q, q + r, r, r); copy_mat(ctx, fb.D, d23, fb.alpha);       // A'XE + E'XA with X = alpha L D L'
                Mat d32 = S.view(q + r, q, r, r); copy_mat(ctx, fb.D, d32, fb.alpha);
                Mat d33 = S.view(q + r, q + r, r, r); gemm(ctx, true, false, -1.0, fb.BtLD, fb.BtLD, 0.0, d33);
            }
            LDLtP R1 = ldlt_make(ctx, n, G, S, 1.0, false);
            // lowrank_ros2.jl:58 compresses here; [C', A'L, E'L] has full numerical rank q + 2r generically, so the engine's
            // compression would run to the end and hand G back (ldlt_compress): it is only attempted in the literal (exact) mode
            // Default mode (round 3; found by the 45-step fixture): R1 is the Riccati residual at X, whose terms cancel as X approaches the steady
            // state (||R1|| / (||G||^2 ||S||) falls below 1e-8 within ~20 steps).  The Gram form of the norm inside the ADI loop is only accurate
            // relative to the largest term, so on the raw summands abstol = n eps ||R1|| and every residual norm became rounding noise (the solves
            // stopped after 0 iterations and K(t) froze 1e-6 away from the oracle).  R1 is therefore always brought to ONE block with orthonormal
            // factor, truncated at max(relative tolerance, formation noise of G S G'): nothing can cancel in that form.
            if (cex) ldlt_compress(ctx, *R1, ctf, cex);
            else ldlt_compress(ctx, *R1, ctf, false, -1.0, COMPRESS_NOISE_FLOOR | COMPRESS_KEEP_RESULT);
            AdiResult a1 = adi_solve(ctx, op, *R1, nullptr, adi, &cache);
            LDLtP K1 = a1.X;
            // stage 2: G2 = E'T1, S2 = (tau^2 B'T1D1)'(B'T1D1) + (2 - 1/gamma) D1     (lowrank_ros2.jl:61-69)
            ldlt_destructure(ctx, *K1, ctf, cex);
            const LBlock kb = K1->blocks[0];
            const int r1 = kb.L.cols;
            Mat BtT1(ctx, m, r1), BtT1D1(ctx, m, r1);
            gemm(ctx, true, false, 1.0, prob.B, kb.L, 0.0, BtT1);
            gemm(ctx, false, false, kb.alpha, BtT1, kb.D, 0.0, BtT1D1);
            Mat G2(ctx, n, r1);
            spmm(ctx, P, P.valEt.p, kb.L, G2, 1.0, 0.0);
            Mat S2(ctx, r1, r1);
            copy_mat(ctx, kb.D, S2, 2.0 - 1.0 / gamma);
            if (r1 > 0) gemm(ctx, true, false, tau * tau, BtT1D1, BtT1D1, 1.0, S2);
            LDLtP R2 = ldlt_make(ctx, n, G2, S2, 1.0, false);
            AdiResult a2 = adi_solve(ctx, op, *R2, nullptr, adi, &cache);
            LDLtP K2 = a2.X;
            // X = X + ((2 - 1/(2 gamma)) tau) K1 + (-tau/2) K2     (lowrank_ros2.jl:72)
            X = ldlt_add(ldlt_add(X, ldlt_scale(K1, (2.0 - 1.0 / (2.0 * gamma)) * tau)), ldlt_scale(K2, -tau / 2.0));
            if (X.get() == prob.X0.get()) X = std::make_shared<LDLt>(*X);   // never compress the caller's X0 in place
            out.adi_iters += a1.iters + a2.iters;
            a1.X.reset(); a1.residual.reset(); a2.X.reset(); a2.residual.reset();
            out.gale.push_back(std::move(a1));
            out.gale.push_back(std::move(a2));
        }
        if (save_state) out.X.push_back(X);
        fb = feedback(ctx, prob, *X, ctf, cex);
        out.Kt.push_back(fb.Kt);
    }
    const auto w_loop = wall_now();
    if (env_trace("pool"))
        std::fprintf(stderr, "[pool] main: %ld misses, %.1f MB; side: %ld misses, %.1f MB\n", ctx->pool.misses(), ctx->pool.total_bytes() / 1048576.0,
                     side ? side->pool.misses() : 0L, side ? side->pool.total_bytes() / 1048576.0 : 0.0);
    sx.report();
    if (x_is_dense) X = dense_to_ldlt(ctx, n, sx.X, ctf);
    if (!save_state) out.X.push_back(X);
    out.nfactor = cache.nfactor;
    if (wall_on) {
        const auto w_end = wall_now();
        auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        std::fprintf(stderr, "[wall, ms] first step %.2f | steps 2..%d %.2f | final form %.2f\n", ms(w_begin, w_first), nsteps, ms(w_first, w_loop), ms(w_loop, w_end));
    }
    return out;
}

}  // namespace dre
