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

#include "engine_internal.hpp"

namespace dre {


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

Mat hcat_blocks(Ctx* ctx, const LDLt& X) {
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
    RoctxRange roctx_range("concatenate!(::LDLᵀ)");
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
void mul_blockdiag(Ctx* ctx, const Mat& M, const LDLt& X, Mat& out) {
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
void hcat_scale_blocks(Ctx* ctx, const LDLt& X, Mat& Lcat, Mat& LD) {
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
        // (warm start: what the known basis leaves of the sketch can be 1e-14 of it — the first projection's own rounding, eps ||Y_b||, is then as large as
        // the remainder, the normalised block is NOT orthogonal to the basis, loses most of its norm in the second round and the second pass reports a
        // breakdown (pivots 0.08 ... 0.17 seen) although the probe confirms the basis to 2e-15: a second projection BEFORE the normalisation acts on
        // the small remainder and leaves it orthogonal to the basis relative to itself)
        if (permissive) project(Yb);
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

// debug (DRE_TRACE=rank): how far the band reduction's rank (a multiple of the panel width, in Krylov order) lies above the rank the reference keeps
// (LDLt.jl:237-245: eigenvalues of at least 100 eps max|lambda|) — the eigenvalues of the J x J band matrix on the host
static void trace_band_rank(Ctx* ctx, const Mat& D, const char* tag, int n, int c) {
    if (!env_trace("rank") || D.rows == 0) return;
    const int J = D.rows;
    std::vector<double> h((size_t)J * J);
    DRE_HIP(hipMemcpy2DAsync(h.data(), (size_t)J * sizeof(double), D.p, (size_t)D.ld * sizeof(double), (size_t)J * sizeof(double), J, hipMemcpyDeviceToHost, ctx->stream));
    DRE_HIP(hipStreamSynchronize(ctx->stream));
    auto ev = host_eigvals(J, h);
    double mx = 0.0;
    for (auto& e : ev) mx = std::max(mx, std::abs(e));
    int r100 = 0, r1 = 0;
    for (auto& e : ev) { if (std::abs(e) >= 100.0 * EPS * mx) ++r100; if (std::abs(e) >= 1e-12 * mx) ++r1; }
    std::fprintf(stderr, "[rank] %s n=%d c=%d: band rank J=%d, eigenvalues >= 100 eps max: %d, >= 1e-12 max: %d\n", tag, n, c, J, r100, r1);
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
    // band_hint[ckey]: Cholesky-QR breakdowns seen at this order.  Two strikes: the blocks are projected TWICE before their first normalisation and a
    // column 14 orders below its block's scale counts as dependent, not as a breakdown (the form of the warm-started range finder: a sketch wider than
    // the numerical rank leaves later blocks almost inside the span of the earlier ones — s = 240 against a rank of 144 in the stage solves of Ros2
    // at n = 5177 — and the first projection's own rounding is then as large as what it leaves).  Two more strikes: Householder panels from then on
    // (round 5 before this: after the first two, 30 % of the device time of a Ros2 run at n = 5177 in k_tsqr_*).  The probe below judges every form.
    const long ckey = skey - 1;
    const int strikes = (int)ctx->band_hint[ckey];
    const bool use_chol = ctx->compress_sketch_cholqr && strikes < 4;
    if (use_chol) orth_cholqr(ctx, Yr, Q, reinterpret_cast<int*>(cflag.p), 0, strikes >= 2);
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
                            use_chol ? (strikes >= 2 ? "CholQR2 blocks, projected twice" : "CholQR2 blocks") : "Householder", sb.J, est, ok ? "accepted" : (chol_bad ? "REJECTED (Cholesky breakdown)" : "REJECTED"));
    if (chol_bad) { ctx->band_hint[ckey] += 1; return false; }
    if (use_sparse && !ok && sb.J + 32 <= s) ctx->band_hint[spkey] += 1;       // enough room in the sketch, yet the probe sees a miss: the test matrix's fault
    if (!ok) { ctx->band_hint[skey] = std::max(ctx->band_hint[skey], std::min(sb.J + 16, s)); return false; }
    ctx->cstats.calls++; ctx->cstats.cols_in += c; ctx->cstats.order += s; ctx->cstats.tri_steps += sb.J; ctx->cstats.rank_out += sb.J;
    ctx->band_hint[skey] = sb.J;
    trace_band_rank(ctx, sb.D, "sketch", n, c);
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
bool warm_compress(Ctx* ctx, LDLt& X, const Mat& Q0, double tolfac, double abs_tol, int sx, double* missed, double rel_accept) {
    const int n = X.n, c = X.rank(), q0 = Q0.cols, sp = sx + 16, s = q0 + sx;
    static const bool trace = env_trace("compress");
    if (c == 0 || q0 < 16 || s < 48 || s + 80 > n || abs_tol <= 0.0 || Q0.rows != n) {
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
    if (trace) std::fprintf(stderr, "[warm compress] n=%d c=%d q0=%d sx=%d -> J=%d  missed %.2e (tolerance %.2e; relative %.2e against %.2e; Cholesky flag %d; room %d)  %s\n", n, c, q0, sx,
                            sb.J, est, abs_tol, est_rel, rel_accept, (int)(cf & 1), s - sb.J, ok ? "accepted" : "REJECTED");
    if (!ok) return false;
    X.blocks.clear();
    if (sb.J == 0) { X.blocks.push_back({Mat(ctx, n, 0), Mat(ctx, 0, 0), 1.0, true}); return true; }
    Mat Bq = sym_band_basis(ctx, sb);                 // s x J
    Mat Lnew(ctx, n, sb.J);
    gemm(ctx, false, false, 1.0, Q, Bq, 0.0, Lnew, nullptr, "gemm_sketch");
    X.blocks.push_back({Lnew, sb.D, 1.0, false, true});
    return true;
}

// The warm-started compression of warm.hip for a residual in FACTORED form (general path: n > dense_inverse_max_n, Res = L Dt L' with a few
// thousand columns): Rayleigh-Ritz in the basis B = [Q, Z], Q = the eigenbasis the previous step's compression left (q <= 64 columns), Z = 16 fresh
// directions.  Products with Res are two GEMMs over the factor (L'V, then L (Dt .)); the small generalized eigenproblem, the truncation and the
// probe are the dense path's kernels.  Replaces the range finder above + band reduction (~30 launches, rank on 16-column panel boundaries of a
// Krylov basis) by ~20 launches and the numerical rank itself.  Returns false (X untouched) when the probe or the whitening rejects the basis;
// on success X is ONE block (R = the J leading eigen-directions zero padded to a multiple of 16, T diagonal-dominant dense), Qnext the next basis.
bool warm_compress_eig(Ctx* ctx, LDLt& X, const Mat& Qb, double abstol_lag, double frac, double est_ratio_prev, int J_prev, Mat& Qnext, int* J_out, double* est_ratio_out) {
    const int n = X.n, c = X.rank(), q = Qb.cols, m = q + 16;
    static const bool trace = env_trace("compress");
    if (c == 0 || q < 16 || q > 64 || q + 32 > n || n < 96 || abstol_lag <= 0.0 || Qb.rows != n) return false;
    RoctxRange roctx_range("compress!(::LDLᵀ)");
    Mat Lcat = (X.blocks.size() == 1) ? X.blocks[0].L : hcat_blocks(ctx, X);
    // the chain's width and the next basis follow the PREVIOUS rank (it moves by a few directions per step): kl = its multiple of 16, 16 spare directions
    const int kl = J_prev >= 0 ? std::min(q, std::max(16, ((J_prev + 15) / 16) * 16 + (J_prev % 16 > 12 ? 16 : 0))) : q, qn = std::min(std::min(m, 64), kl + 16);
    Mat V0(ctx, n, 32 + m), YB(ctx, n, 64 + q), W1(ctx, 32 + q, c), W2(ctx, 32 + q, c), Pf(ctx, q, 16), Z1(ctx, n, 16);
    { Mat om = V0.colsview(0, 32); fill_gauss(ctx, om, 0x7F4A7C159E3779B9ull); }
    { Mat d = V0.colsview(32, q); copy_mat(ctx, Qb, d); }
    Mat V0q = V0.colsview(0, 32 + q), Yq = YB.colsview(0, 32 + q);
    gemm(ctx, true, false, 1.0, V0q, Lcat, 0.0, W1, nullptr, "gemm_sketch");                  // [Om_p, Om_f, Q]' L
    mul_blockdiag(ctx, W1, X, W2);                                                            // . Dt
    gemm(ctx, false, true, 1.0, Lcat, W2, 0.0, Yq, nullptr, "gemm_sketch");                  // [Y_p, Y_f, W] = Res [Om_p, Om_f, Q]
    Mat Qv = V0.colsview(32, q), Bb = V0.colsview(32, m), Yp = YB.colsview(0, 16), Yf = YB.colsview(16, 16);
    gemm(ctx, true, false, 1.0, Qv, Yf, 0.0, Pf, nullptr, "gemm_sketch");                    // P_f = Q'Y_f
    const int nslab = warm_project_slabs(n), nstrip = ceil_div(n, 16);
    DevArr<double> slab(ctx, (size_t)nslab * 256 + nstrip + 256 + 16);
    double* const slabF = slab.p + (size_t)nslab * 256;
    double* const Cw = slabF + nstrip;
    double* const tols = Cw + 256;                        // 12 doubles
    if (!ctx->warm_tickets) {
        ctx->warm_tickets = std::make_shared<Buf>(ctx, 4 * sizeof(int));
        DRE_HIP(hipMemsetAsync(ctx->warm_tickets->p, 0, 4 * sizeof(int), ctx->stream));
    }
    int* const tickets = (int*)ctx->warm_tickets->p;
    warm_project(ctx, n, q, Qv, Yf, Pf, Z1, slab.p, tickets, Cw);
    Mat Zb = V0.colsview(32 + q, 16), Zy = YB.colsview(32 + q, 16), Wz = YB.colsview(48 + q, 16);
    warm_zapply(ctx, n, Z1, Cw, Zb, Zy);
    {
        Mat A1(ctx, 16, c), A2(ctx, 16, c);
        gemm(ctx, true, false, 1.0, Zb, Lcat, 0.0, A1, nullptr, "gemm_sketch");              // Z'L
        mul_blockdiag(ctx, A1, X, A2);
        gemm(ctx, false, true, 1.0, Lcat, A2, 0.0, Wz, nullptr, "gemm_sketch");              // W_2 = Res Z
    }
    Mat Cc(ctx, m, 64 + q), Uc(ctx, m, qn), Tm(ctx, kl, kl), Cp(ctx, m, 16);
    gemm(ctx, true, false, 1.0, Bb, YB, 0.0, Cc, nullptr, "gemm_sketch");                    // B' [Y_p, Y_f, W, Z, W_2]
    const double bf = est_ratio_prev < 0.0 ? 0.4 : std::min(0.6, std::max(0.05, 1.0 - 2.6 * est_ratio_prev));
    warm_small(ctx, q, m, kl, qn, Cc, nullptr, 0, 0.0, abstol_lag, frac, bf, tols, Uc, Tm, Cp, tickets + 1);
    Mat Rfull(ctx, n, qn);
    warm_finish(ctx, n, m, qn, Bb, Uc, Yp, Cp, Rfull, slabF, tickets + 1, tols);
    double h[12];
    ctx_fetch(ctx, tols, 12 * sizeof(double), h);
    const int J = (int)h[4];
    const bool ok = h[5] == 0.0 && J >= 0 && J <= kl;
    if (trace) std::fprintf(stderr, "[warm eig] n=%d c=%d q=%d -> J=%d  missed^2 %.2e dropped^2 %.2e tol^2 %.2e  %s\n", n, c, q, J, h[6], h[7], h[1] * h[1], ok ? "accepted" : "REJECTED");
    if (!ok) return false;
    const int k = std::min(kl, std::max(16, ((J + 15) / 16) * 16));
    X.blocks.clear();
    X.blocks.push_back({Rfull.colsview(0, k), Tm.view(0, 0, k, k), 1.0, false, true});
    Qnext = Rfull;
    if (J_out) *J_out = J;
    if (est_ratio_out) *est_ratio_out = h[1] > 0.0 ? h[6] / (h[1] * h[1]) : -1.0;
    ctx->cstats.calls++; ctx->cstats.cols_in += c; ctx->cstats.order += m; ctx->cstats.rank_out += J;
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
static void ldlt_compress_core(Ctx* ctx, LDLt& X, double tolfac, bool exact, double abs_tol, int mode, double* floor_used);
// COMPRESS_TIGHT (engine.hpp): X = L D L' with orthonormal L (n x J) and the J x J band matrix D of a finished compression -> eigenvalues of D,
// the reference's threshold, L <- L U_keep, D <- diag(lambda_keep)
static void ldlt_tighten(Ctx* ctx, LDLt& X, double tolfac, double floor_abs) {
    if (X.blocks.size() != 1) return;
    LBlock& b = X.blocks[0];
    const int n = X.n, J = b.L.cols;
    if (J < 2 || b.diag || !b.ortho || b.D.rows != J) return;
    Mat Dc(ctx, J, J);
    copy_mat(ctx, b.D, Dc, b.alpha);
    SymEig e = sym_eig(ctx, Dc, tolfac, true, -1.0, false, 1.0);          // (deflation at eps ||D||: only the eigenvalues above 100 eps max|lambda| are wanted)
    if (e.j == 0) return;
    double wmax = 0.0;
    for (double w : e.w) wmax = std::max(wmax, std::fabs(w));
    const double thr = std::max(100.0 * wmax * EPS, floor_abs);
    std::vector<int> ids;
    for (int i = 0; i < e.j; ++i)
        if (std::fabs(e.w[i]) >= thr && wmax > 0.0) ids.push_back(i);
    std::sort(ids.begin(), ids.end(), [&](int a, int c) { return e.w[a] < e.w[c]; });
    const int r = (int)ids.size();
    static const bool trace = env_trace("compress");
    if (trace) std::fprintf(stderr, "[compress tight] n=%d band rank %d -> %d (threshold %.2e, largest %.2e)\n", n, J, r, thr, wmax);
    if (r == 0 || r >= J) return;          // (nothing above the threshold: the band form stays, as before; nothing gained: keep it)
    Mat U = sym_eig_backtransform(ctx, e, ids);          // J x r
    Mat Lnew(ctx, n, r);
    gemm(ctx, false, false, 1.0, b.L, U, 0.0, Lnew, nullptr, "gemm_compress");
    std::vector<double> hd((size_t)r * r, 0.0);
    for (int i = 0; i < r; ++i) hd[i + (size_t)i * r] = e.w[ids[i]];
    Mat Dnew(ctx, r, r);
    DRE_HIP(hipMemcpyAsync(Dnew.p, hd.data(), hd.size() * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    DRE_HIP(hipStreamSynchronize(ctx->stream));
    X.blocks.clear();
    X.blocks.push_back({Lnew, Dnew, 1.0, true, true});
}
void ldlt_compress(Ctx* ctx, LDLt& X, double tolfac, bool exact, double abs_tol, int mode) {
    RoctxRange roctx_range("compress!(::LDLᵀ)");
    double floor_used = 0.0;
    const bool tight = !exact && (mode & COMPRESS_TIGHT);
    ldlt_compress_core(ctx, X, tolfac, exact, abs_tol, mode & ~COMPRESS_TIGHT, tight ? &floor_used : nullptr);
    if (tight) ldlt_tighten(ctx, X, tolfac, floor_used);
}
static void ldlt_compress_core(Ctx* ctx, LDLt& X, double tolfac, bool exact, double abs_tol, int mode, double* floor_used) {
    const int n = X.n, c = X.rank();
    const bool nfloor = !exact && (mode & COMPRESS_NOISE_FLOOR), keep_result = !exact && (mode & COMPRESS_KEEP_RESULT);
    if (nfloor) abs_tol = -1.0;
    DevArr<double> nf;                       // [0] the floor, [1..] per-column products
    auto floor_of = [&](const Mat& A, const Mat& B) {       // device-side floor for S = A B'
        nf = DevArr<double>(ctx, (size_t)A.cols + 1);
        hipLaunchKernelGGL(k_colpair_noise, dim3(A.cols), dim3(256), 0, ctx->stream, A.rows, (const double*)A.p, A.ld, (const double*)B.p, B.ld, nf.p + 1);
        hipLaunchKernelGGL(k_noise_floor, dim3(1), dim3(256), 0, ctx->stream, A.cols, noise_floor_fac(), (const double*)(nf.p + 1), nf.p);
    };
    auto floor_host = [&]() { double h = 0.0; ctx_fetch(ctx, nf.p, sizeof(double), &h); if (floor_used) *floor_used = h; return h; };
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
    // More columns than rows at large n (round 5: the increments of a whole Lyapunov solve with a 144-column residual at n = 5177 — the
    // residual-recurrence loop no longer compresses inside the solve): the direct form would reduce the n x n matrix (60 ms and a rank of 896 at
    // n = 5177: its relative tolerance sits below the noise of forming S); the sketch only ever multiplies with the factor.
    const bool wide_sk = wide && !exact && !nfloor && abs_tol <= 0.0 && ctx->compress_sketch && n >= ctx->compress_factor_min_n && n > ctx->compress_direct_max_n &&
                         c >= ctx->compress_sketch_min_cols;
    if (sketchable || wide_sk) {
        auto hit = ctx->band_hint.find(skey);
        int s = 0;
        if (hit != ctx->band_hint.end() && hit->second > 0) s = ((hit->second + ctx->compress_sketch_extra + 15) / 16) * 16;
        else if (wide_sk) s = 320;
        if (s > 0 && (double)c >= ctx->compress_sketch_ratio * s && s + 80 <= n) {
            if (sketch_compress(ctx, X, tolfac, s, skey)) return;
            if (wide_sk) {          // one retry with a doubled sketch (a rejected attempt leaves the rank it saw in the hint)
                const int s2 = std::min(((std::max(ctx->band_hint[skey] + ctx->compress_sketch_extra, 2 * s) + 15) / 16) * 16, ((n - 96) / 16) * 16);
                if (s2 > s && sketch_compress(ctx, X, tolfac, s2, skey)) return;
            }
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
        trace_band_rank(ctx, sb.D, "factor form", n, c);
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
        if (!exact && c >= 32 && n >= rot_min_n) {
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
            if (nfloor && floor_used) (void)floor_host();
            ctx->cstats.calls++; ctx->cstats.cols_in += c; ctx->cstats.order += S.rows; ctx->cstats.tri_steps += sb.J;
            r = sb.J;
            ctx->cstats.rank_out += r;
            if (r == 0) { set_empty(); return; }
            trace_band_rank(ctx, sb.D, wide ? "direct" : "qr", n, c);
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
    RoctxRange roctx_range("norm(::LDLᵀ)");
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
// GALE residual (/root/reference/src/lyapunov/residual.jl:3-31)
// =============================================================================================
// largest n for which the Ros1 driver carries X as a block list (right-hand side, feedback and residual on the summands)
int xblocks_max_n() {
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
double ldlt_norm_dense_small(Ctx* ctx, const LDLt& X) {
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
void axpy_inplace(Ctx* ctx, size_t tot, double a, const double* x, double* y) {
    if (!tot) return;
    hipLaunchKernelGGL(k_axpy_inplace, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, ctx->stream, tot, a, x, y);
}
LDLtP gale_residual_blocks(Ctx* ctx, const GaleOperator& op, const LDLt& C, const LDLt& X, double tolfac, double abs_tol,
                           const Mat* warm_L, const Mat* warm_EtL, int lead_blocks, double e_coeff) {
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

LDLtP gale_residual(Ctx* ctx, const GaleOperator& op, LDLt& C, const LDLtP& X, double tolfac, bool exact, double abs_tol) {
    RoctxRange roctx_range("residual(::GALEProblem, ::LDLᵀ)");
    return gale_residual_impl(ctx, op, C, X, tolfac, exact, abs_tol, nullptr, nullptr);
}
LDLtP gale_residual_impl(Ctx* ctx, const GaleOperator& op, LDLt& C, const LDLtP& X, double tolfac, bool exact, double abs_tol,
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
    s = wave_sum_d(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = (red[0] + red[1]) + (red[2] + red[3]);
}
double ldlt_dot(Ctx* ctx, const LDLt& X1, const LDLt& X2) {
    RoctxRange roctx_range("dot(::LDLᵀ, ::LDLᵀ)");
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
    RoctxRange roctx_range("residual(::GAREProblem, ::LDLᵀ)");
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


}  // namespace dre
