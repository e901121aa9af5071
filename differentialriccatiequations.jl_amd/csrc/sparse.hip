// Sparse kernels for gfx950: CSR SpMM, multifrontal LU (assembly, extend-add, dense partial LU with
// inverted diagonal blocks) and level-scheduled multi-RHS triangular solves, real and complex.
#include "sparse.hpp"
#include "profiling.hpp"

#include <algorithm>
#include <cstdlib>

namespace dre {

// =============================================================================================
// Pencil construction (host)
// =============================================================================================
std::unique_ptr<Pencil> pencil_create(Ctx* ctx, int n, const int64_t* Ep, const int64_t* Ei, const double* Ev,
                                      const int64_t* Ap, const int64_t* Ai, const double* Av, int base, int leaf_size,
                                      bool upload, std::vector<double>* hostE, std::vector<double>* hostA) {
    DRE_REQUIRE(n > 0, "pencil_create: n must be positive");
    DRE_REQUIRE(base == 0 || base == 1, "pencil_create: index base must be 0 or 1");
    auto P = std::make_unique<Pencil>();
    P->n = n;
    // The CSC arrays of a matrix are the CSR arrays of its transpose: "row" c below is column c of E / A.
    std::vector<int> uptr(n + 1, 0), uidx;
    {
        std::vector<int> tmp;
        for (int c = 0; c < n; ++c) {
            tmp.clear();
            for (int64_t p = Ep[c] - base; p < Ep[c + 1] - base; ++p) tmp.push_back((int)(Ei[p] - base));
            for (int64_t p = Ap[c] - base; p < Ap[c + 1] - base; ++p) tmp.push_back((int)(Ai[p] - base));
            std::sort(tmp.begin(), tmp.end());
            tmp.erase(std::unique(tmp.begin(), tmp.end()), tmp.end());
            for (int r : tmp) DRE_REQUIRE(r >= 0 && r < n, "pencil_create: row index out of range");
            uidx.insert(uidx.end(), tmp.begin(), tmp.end());
            uptr[c + 1] = (int)uidx.size();
        }
    }
    int leaf = leaf_size > 0 ? leaf_size : (n <= 2000 ? 24 : 32);
    P->sym = symbolic_analyze(n, uptr, uidx, leaf);
    const Symbolic& S = P->sym;
    P->nnz = (int)S.idx.size();
    // scatter values into the permuted pattern
    std::vector<double> vE(P->nnz, 0.0), vA(P->nnz, 0.0);
    auto scatter = [&](const int64_t* Cp, const int64_t* Ci, const double* Cv, std::vector<double>& out) {
        for (int c = 0; c < n; ++c) {
            const int pr = S.iperm[c];   // row of the transposed, permuted matrix
            const int* rb = S.idx.data() + S.ptr[pr];
            const int* re = S.idx.data() + S.ptr[pr + 1];
            for (int64_t p = Cp[c] - base; p < Cp[c + 1] - base; ++p) {
                const int pc = S.iperm[(int)(Ci[p] - base)];
                const int* it = std::lower_bound(rb, re, pc);
                if (it == re || *it != pc) throw Error(ERR_INTERNAL, "pencil_create: entry missing from union pattern");
                out[S.ptr[pr] + (it - rb)] += Cv[p];
            }
        }
    };
    scatter(Ep, Ei, Ev, vE);
    scatter(Ap, Ai, Av, vA);
    if (hostE) *hostE = vE;
    if (hostA) *hostA = vA;
    P->lvl_maxfront.assign(S.nlevels, 0);
    P->lvl_maxsep.assign(S.nlevels, 0);
    for (int t = 0; t < S.nnodes; ++t) {
        P->lvl_maxfront[S.level[t]] = std::max(P->lvl_maxfront[S.level[t]], S.fsize(t));
        P->lvl_maxsep[S.level[t]] = std::max(P->lvl_maxsep[S.level[t]], S.size[t]);
    }
    if (!upload) return P;

    auto up_i = [&](DevArr<int>& d, const std::vector<int>& h) { d = DevArr<int>(ctx, std::max<size_t>(h.size(), 1)); d.upload(ctx, h); };
    auto up_l = [&](DevArr<int64_t>& d, const std::vector<int64_t>& h) { d = DevArr<int64_t>(ctx, std::max<size_t>(h.size(), 1)); d.upload(ctx, h); };
    up_i(P->ptr, S.ptr); up_i(P->idx, S.idx);
    P->valEt = DevArr<double>(ctx, P->nnz); P->valEt.upload(ctx, vE);
    P->valAt = DevArr<double>(ctx, P->nnz); P->valAt.upload(ctx, vA);
    up_i(P->perm, S.perm); up_i(P->iperm, S.iperm);
    up_i(P->dev.first, S.first); up_i(P->dev.size, S.size); up_i(P->dev.bptr, S.bptr); up_i(P->dev.bidx, S.bidx);
    up_i(P->dev.cmap_ptr, S.cmap_ptr); up_i(P->dev.cmap, S.cmap); up_i(P->dev.child_ptr, S.child_ptr);
    up_i(P->dev.child_idx, S.child_idx); up_i(P->dev.lvl_nodes, S.lvl_nodes);
    up_l(P->dev.front_off, S.front_off); up_l(P->dev.inv_off, S.inv_off); up_l(P->dev.upd_off, S.upd_off);
    up_l(P->dev.asm_dest, S.asm_dest);
    P->has_device = true;
    return P;
}

// =============================================================================================
// SpMM: one thread per row, 8 columns per workgroup column-slice.  Rows of a 7-point FEM operator
// are short (<= 7 nnz), so a row-per-thread mapping keeps loads of Y/X coalesced along the rows
// (column-major panels) and the gathered X rows hit L2.
// =============================================================================================
#define SPMM_CB 8
// LDS-staged variant: the CSR segment (column indices + values) of the workgroup's 256 consecutive rows is one contiguous range of
// the arrays; it is loaded cooperatively (fully coalesced) into LDS once and every thread then walks its own row in LDS.  The gathered
// rows of the dense panel X stay L2 gathers (the nested-dissection ordering keeps the neighbours of a row close).
#define SPMM_LDS_NNZ 3584
__global__ __launch_bounds__(256) void k_spmm_lds(int n, const int* __restrict__ ptr, const int* __restrict__ idx,
                                                  const double* __restrict__ val, const double* __restrict__ X, int ldx,
                                                  double* __restrict__ Y, int ldy, int ncols, double alpha, double beta,
                                                  const AdiState* st, double* __restrict__ Yt, int ldyt) {
    if (st && st->done) return;
    __shared__ double vs[SPMM_LDS_NNZ];
    __shared__ int is[SPMM_LDS_NNZ];
    const int r0 = blockIdx.x * 256, r1 = min(n, r0 + 256);
    const int p0 = ptr[r0], p1 = ptr[r1];
    const bool staged = (p1 - p0) <= SPMM_LDS_NNZ;
    if (staged)
        for (int p = p0 + threadIdx.x; p < p1; p += 256) { vs[p - p0] = val[p]; is[p - p0] = idx[p]; }
    __syncthreads();
    const int i = r0 + threadIdx.x;
    if (i >= n) return;
    const int c0 = blockIdx.y * SPMM_CB;
    const int c1 = min(ncols, c0 + SPMM_CB);
    const int pb = ptr[i], pe = ptr[i + 1];
    double acc[SPMM_CB];
#pragma unroll
    for (int c = 0; c < SPMM_CB; ++c) acc[c] = 0.0;
    if (staged && c1 - c0 == SPMM_CB) {
        // full column block: no predicate inside the loop, so the 8 gathers of an entry (and, two entries at a time, 16) are issued
        // back to back and the old values of Y are requested before the loop — the predicated form made the compiler wait per load
        double yold[SPMM_CB];
#pragma unroll
        for (int c = 0; c < SPMM_CB; ++c) yold[c] = (beta != 0.0) ? Y[i + (size_t)(c0 + c) * ldy] : 0.0;
        const double* __restrict__ xb = X + (size_t)c0 * ldx;
        int p = pb - p0;
        const int pend = pe - p0;
        for (; p + 1 < pend; p += 2) {
            const double v0 = vs[p], v1 = vs[p + 1];
            const double* x0 = xb + is[p];
            const double* x1 = xb + is[p + 1];
            double t0[SPMM_CB], t1[SPMM_CB];
#pragma unroll
            for (int c = 0; c < SPMM_CB; ++c) { t0[c] = x0[(size_t)c * ldx]; t1[c] = x1[(size_t)c * ldx]; }
#pragma unroll
            for (int c = 0; c < SPMM_CB; ++c) { acc[c] += v0 * t0[c]; acc[c] += v1 * t1[c]; }
        }
        if (p < pend) {
            const double v0 = vs[p];
            const double* x0 = xb + is[p];
#pragma unroll
            for (int c = 0; c < SPMM_CB; ++c) acc[c] += v0 * x0[(size_t)c * ldx];
        }
#pragma unroll
        for (int c = 0; c < SPMM_CB; ++c) {
            const double o = (beta == 0.0) ? alpha * acc[c] : alpha * acc[c] + beta * yold[c];
            Y[i + (size_t)(c0 + c) * ldy] = o;
            if (Yt) Yt[(c0 + c) + (size_t)i * ldyt] = o;          // the transposed copy rides along (64 contiguous bytes per thread)
        }
        return;
    }
    for (int p = pb; p < pe; ++p) {
        const double v = staged ? vs[p - p0] : val[p];
        const double* x = X + (staged ? is[p - p0] : idx[p]) + (size_t)c0 * ldx;
#pragma unroll
        for (int c = 0; c < SPMM_CB; ++c)
            if (c0 + c < c1) acc[c] += v * x[(size_t)c * ldx];
    }
#pragma unroll
    for (int c = 0; c < SPMM_CB; ++c)
        if (c0 + c < c1) {
            double* y = Y + i + (size_t)(c0 + c) * ldy;
            const double o = (beta == 0.0) ? alpha * acc[c] : alpha * acc[c] + beta * (*y);
            *y = o;
            if (Yt) Yt[(c0 + c) + (size_t)i * ldyt] = o;
        }
}
void spmm(Ctx* ctx, int n, const int* ptr, const int* idx, const double* val, const Mat& X, Mat& Y, double alpha,
          double beta, const AdiState* st, int nnz, Mat* Yt) {
    DRE_REQUIRE(X.rows == n && Y.rows == n && X.cols == Y.cols, "spmm: shape mismatch");
    DRE_REQUIRE(!Yt || (Yt->rows == Y.cols && Yt->cols == n), "spmm: shape of the transposed copy");
    if (X.cols == 0) return;
    // algorithmic bytes: CSR (12 B/nnz + 4 B/row) + X read + Y read/write; nnz < 0: unknown to the caller, 7-point estimate
    const double z = nnz >= 0 ? (double)nnz : 7.0 * n;
    TimedScope ts(ctx, "spmm_csr", 12.0 * z + 4.0 * n + 8.0 * n * X.cols * (beta == 0.0 ? 2.0 : 3.0), 2.0 * z * X.cols);
    hipLaunchKernelGGL(k_spmm_lds, dim3(ceil_div(n, 256), ceil_div(X.cols, SPMM_CB)), dim3(256), 0, ctx->stream, n, ptr, idx, val,
                       X.p, X.ld, Y.p, Y.ld, X.cols, alpha, beta, st, Yt ? Yt->p : (double*)nullptr, Yt ? Yt->ld : 0);
    DRE_HIP(hipGetLastError());
}
// Fan groups (engine.hip): W = [W_1 .. W_g] (n x g k) are g independent solves with the same right-hand side R_0; the g ADI iterates they stand for are
//   V_j = sum_{s<=j} c_js W_s,   R_j = R_0 - sum_{s<=j} d_js (E' W_s)      (adi.jl:158-171 re-associated through the resolvent identity)
// — the product with E' and the mixing in ONE pass: a thread owns a row and FAN_CB columns of all g blocks, gathers the neighbour rows of the g
// panels (CSR segment of the workgroup's 256 rows staged in LDS as in k_spmm_lds) and writes the 2 g outputs.  Replaces SpMM + k_fan_mix (two
// launches, E' W written and read back).
#define FAN_CB 2
template <int G>
__global__ __launch_bounds__(256) void k_fan_spmm_mix(int n, const int* __restrict__ ptr, const int* __restrict__ idx, const double* __restrict__ val, int k,
                                                      const double* __restrict__ W, int ldw, const double* __restrict__ R0, int ldr,
                                                      double* __restrict__ V, int ldv, double* __restrict__ Rc, int ldrc, FanCoef co, FanSlots sl, const AdiState* st) {
    if (st && st->done) return;
    __shared__ double vs[SPMM_LDS_NNZ];
    __shared__ int is[SPMM_LDS_NNZ];
    const int r0 = blockIdx.x * 256, r1 = min(n, r0 + 256);
    const int p0 = ptr[r0], p1 = ptr[r1];
    const bool staged = (p1 - p0) <= SPMM_LDS_NNZ;
    if (staged)
        for (int p = p0 + threadIdx.x; p < p1; p += 256) { vs[p - p0] = val[p]; is[p - p0] = idx[p]; }
    __syncthreads();
    const int i = r0 + threadIdx.x;
    if (i >= n) return;
    const int c0 = blockIdx.y * FAN_CB;
    int col[FAN_CB];
#pragma unroll
    for (int c = 0; c < FAN_CB; ++c) col[c] = min(c0 + c, k - 1);        // (clamped: a ragged last block recomputes the last column and does not store it)
    double ew[FAN_CB][G], w[FAN_CB][G], r[FAN_CB];
#pragma unroll
    for (int c = 0; c < FAN_CB; ++c) {
        r[c] = R0[i + (size_t)col[c] * ldr];
#pragma unroll
        for (int s = 0; s < G; ++s) { ew[c][s] = 0.0; w[c][s] = W[i + (size_t)(sl.s[s] * k + col[c]) * ldw]; }
    }
    const int pb = ptr[i], pe = ptr[i + 1];
    for (int p = pb; p < pe; ++p) {
        const double v = staged ? vs[p - p0] : val[p];
        const double* __restrict__ x = W + (staged ? is[p - p0] : idx[p]);
        double t[FAN_CB][G];
#pragma unroll
        for (int c = 0; c < FAN_CB; ++c)
#pragma unroll
            for (int s = 0; s < G; ++s) t[c][s] = x[(size_t)(sl.s[s] * k + col[c]) * ldw];
#pragma unroll
        for (int c = 0; c < FAN_CB; ++c)
#pragma unroll
            for (int s = 0; s < G; ++s) ew[c][s] += v * t[c][s];
    }
#pragma unroll
    for (int c = 0; c < FAN_CB; ++c) {
        if (c0 + c >= k) break;
#pragma unroll
        for (int j = 0; j < G; ++j) {
            double vv = 0.0, y = 0.0;
#pragma unroll
            for (int s = 0; s <= j; ++s) { vv += co.c[j][s] * w[c][s]; y += co.d[j][s] * ew[c][s]; }
            V[i + (size_t)(j * k + c0 + c) * ldv] = vv;
            Rc[i + (size_t)(j * k + c0 + c) * ldrc] = r[c] - y;
        }
    }
}
void fan_spmm_mix(Ctx* ctx, const Pencil& P, const Mat& W, const Mat& R0, Mat& V, Mat& Rc, int g, int k, const FanCoef& co, const FanSlots& sl, const AdiState* st) {
    const int n = P.n;
    DRE_REQUIRE(g >= 1 && g <= FAN_GMAX && W.rows == n && W.cols >= g * k && V.cols == g * k && Rc.cols == g * k && R0.cols == k, "fan_spmm_mix: shapes");
    for (int s = 0; s < g; ++s) DRE_REQUIRE(sl.s[s] >= 0 && (sl.s[s] + 1) * k <= W.cols, "fan_spmm_mix: slot outside the panel");
    TimedScope ts(ctx, "fan_spmm_mix", 12.0 * P.nnz + 4.0 * n + 8.0 * n * k * (3.0 * g + 1.0), 2.0 * (double)P.nnz * g * k + 2.0 * n * k * (double)g * (g + 1));
    const dim3 grid(ceil_div(n, 256), ceil_div(k, FAN_CB)), block(256);
#define DRE_FAN_CASE(G_) case G_: hipLaunchKernelGGL((k_fan_spmm_mix<G_>), grid, block, 0, ctx->stream, n, (const int*)P.ptr.p, (const int*)P.idx.p, (const double*)P.valEt.p, k, \
                                                    (const double*)W.p, W.ld, (const double*)R0.p, R0.ld, V.p, V.ld, Rc.p, Rc.ld, co, sl, st); break;
    switch (g) { DRE_FAN_CASE(1) DRE_FAN_CASE(2) DRE_FAN_CASE(3) DRE_FAN_CASE(4) DRE_FAN_CASE(5) DRE_FAN_CASE(6) DRE_FAN_CASE(7) DRE_FAN_CASE(8) DRE_FAN_CASE(9) DRE_FAN_CASE(10) }
#undef DRE_FAN_CASE
    DRE_HIP(hipGetLastError());
}

// Two operators on the SAME pattern applied to the same panel in one pass (the pencil keeps E' and A' on one union pattern):
// Y1 = M1 X, Y2 = M2 X.  128 rows x 4 columns per workgroup: the panel rows are gathered once for both products, and the finer
// decomposition fills the chip at small n (n = 371, 371 columns: 279 workgroups instead of 2 x 94).
#define SPMM2_ROWS 128
#define SPMM2_CB 4
#define SPMM2_NNZ 2048
__global__ __launch_bounds__(SPMM2_ROWS) void k_spmm_dual(int n, const int* __restrict__ ptr, const int* __restrict__ idx, const double* __restrict__ val1,
                                                          const double* __restrict__ val2, const double* __restrict__ X, int ldx, double* __restrict__ Y1,
                                                          int ldy1, double* __restrict__ Y2, int ldy2, int ncols) {
    __shared__ double v1s[SPMM2_NNZ], v2s[SPMM2_NNZ];
    __shared__ int is[SPMM2_NNZ];
    const int r0 = blockIdx.x * SPMM2_ROWS, r1 = min(n, r0 + SPMM2_ROWS);
    const int p0 = ptr[r0], p1 = ptr[r1];
    const bool staged = (p1 - p0) <= SPMM2_NNZ;
    if (staged)
        for (int p = p0 + threadIdx.x; p < p1; p += SPMM2_ROWS) { v1s[p - p0] = val1[p]; v2s[p - p0] = val2[p]; is[p - p0] = idx[p]; }
    __syncthreads();
    const int i = r0 + threadIdx.x;
    if (i >= n) return;
    const int c0 = blockIdx.y * SPMM2_CB, c1 = min(ncols, c0 + SPMM2_CB);
    const int pb = ptr[i], pe = ptr[i + 1];
    double a1[SPMM2_CB], a2[SPMM2_CB];
#pragma unroll
    for (int c = 0; c < SPMM2_CB; ++c) { a1[c] = 0.0; a2[c] = 0.0; }
    if (staged && c1 - c0 == SPMM2_CB) {
        // full column block, predicate-free (see k_spmm_lds): the gathers of two entries in flight together
        const double* __restrict__ xb = X + (size_t)c0 * ldx;
        int p = pb - p0;
        const int pend = pe - p0;
        for (; p + 1 < pend; p += 2) {
            const double* x0 = xb + is[p];
            const double* x1 = xb + is[p + 1];
            double t0[SPMM2_CB], t1[SPMM2_CB];
#pragma unroll
            for (int c = 0; c < SPMM2_CB; ++c) { t0[c] = x0[(size_t)c * ldx]; t1[c] = x1[(size_t)c * ldx]; }
#pragma unroll
            for (int c = 0; c < SPMM2_CB; ++c) {
                a1[c] += v1s[p] * t0[c]; a2[c] += v2s[p] * t0[c];
                a1[c] += v1s[p + 1] * t1[c]; a2[c] += v2s[p + 1] * t1[c];
            }
        }
        if (p < pend) {
            const double* x0 = xb + is[p];
#pragma unroll
            for (int c = 0; c < SPMM2_CB; ++c) { const double xv = x0[(size_t)c * ldx]; a1[c] += v1s[p] * xv; a2[c] += v2s[p] * xv; }
        }
#pragma unroll
        for (int c = 0; c < SPMM2_CB; ++c) { Y1[i + (size_t)(c0 + c) * ldy1] = a1[c]; Y2[i + (size_t)(c0 + c) * ldy2] = a2[c]; }
        return;
    }
    for (int p = pb; p < pe; ++p) {
        const double w1 = staged ? v1s[p - p0] : val1[p], w2 = staged ? v2s[p - p0] : val2[p];
        const double* x = X + (staged ? is[p - p0] : idx[p]) + (size_t)c0 * ldx;
#pragma unroll
        for (int c = 0; c < SPMM2_CB; ++c)
            if (c0 + c < c1) { const double xv = x[(size_t)c * ldx]; a1[c] += w1 * xv; a2[c] += w2 * xv; }
    }
#pragma unroll
    for (int c = 0; c < SPMM2_CB; ++c)
        if (c0 + c < c1) { Y1[i + (size_t)(c0 + c) * ldy1] = a1[c]; Y2[i + (size_t)(c0 + c) * ldy2] = a2[c]; }
}
void spmm_dual(Ctx* ctx, const Pencil& P, const double* val1, const double* val2, const Mat& X, Mat& Y1, Mat& Y2) {
    const int n = P.n;
    DRE_REQUIRE(X.rows == n && Y1.rows == n && Y2.rows == n && X.cols == Y1.cols && X.cols == Y2.cols, "spmm_dual: shape mismatch");
    if (X.cols == 0) return;
    TimedScope ts(ctx, "spmm_csr", 20.0 * P.nnz + 4.0 * n + 8.0 * n * X.cols * 3.0, 4.0 * P.nnz * X.cols);
    hipLaunchKernelGGL(k_spmm_dual, dim3(ceil_div(n, SPMM2_ROWS), ceil_div(X.cols, SPMM2_CB)), dim3(SPMM2_ROWS), 0, ctx->stream, n, (const int*)P.ptr.p,
                       (const int*)P.idx.p, val1, val2, (const double*)X.p, X.ld, Y1.p, Y1.ld, Y2.p, Y2.ld, X.cols);
    DRE_HIP(hipGetLastError());
}
__global__ void k_axpby(int n, double a, const double* __restrict__ x, double b, const double* __restrict__ y, double* __restrict__ out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = a * x[i] + b * y[i];
}
void vals_axpby(Ctx* ctx, int nnz, double a, const double* x, double b, const double* y, double* out) {
    hipLaunchKernelGGL(k_axpby, dim3(ceil_div(nnz, 256)), dim3(256), 0, ctx->stream, nnz, a, x, b, y, out);
}
__global__ void k_permute_rows(int rows, int cols, const double* __restrict__ src, int lds, const int* __restrict__ map,
                               double* __restrict__ dst, int ldd) {
    size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= (size_t)rows * cols) return;
    int r = id % rows, c = id / rows;
    dst[r + (size_t)c * ldd] = src[map[r] + (size_t)c * lds];
}
void permute_rows(Ctx* ctx, const Mat& src, const int* map, Mat& dst) {
    DRE_REQUIRE(src.rows == dst.rows && src.cols == dst.cols, "permute_rows: shape mismatch");
    size_t tot = (size_t)src.rows * src.cols;
    if (!tot) return;
    hipLaunchKernelGGL(k_permute_rows, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, ctx->stream, src.rows, src.cols, src.p, src.ld, map, dst.p, dst.ld);
}

// =============================================================================================
// Multifrontal numeric factorisation
// =============================================================================================
template <typename T>
__global__ void k_assemble(int nnz, const int64_t* __restrict__ dest, const double* __restrict__ vF,
                           const double* __restrict__ vE, T cF, T cE, T* __restrict__ fronts, double rel, double* __restrict__ pivfloor) {
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    double m = 0.0;
    if (p < nnz) { const T v = cF * vF[p] + cE * vE[p]; fronts[dest[p]] = v; m = abs1(v); }
    if (pivfloor) {
        // static pivoting: floor = rel * max |entry| of the shifted operator (non-negative doubles order like their bit patterns), as [0] of the
        // factor's control block; every workgroup contributes its maximum
        for (int o = 32; o > 0; o >>= 1) m = fmax(m, __shfl_xor(m, o, 64));
        if ((threadIdx.x & 63) == 0 && m > 0.0) atomicMax(reinterpret_cast<unsigned long long*>(pivfloor), (unsigned long long)__double_as_longlong(rel * m));
    }
}

// the same for up to MF_ZMAX factors of a batch (blockIdx.y = factor): consecutive blocks of `stride` entries in ONE allocation, control blocks 64 bytes apart
template <typename T>
struct AssembleZ { T cE[MF_ZMAX]; };
template <typename T>
__global__ void k_assemble_z(int nnz, const int64_t* __restrict__ dest, const double* __restrict__ vF, const double* __restrict__ vE, T cF, AssembleZ<T> ce,
                             T* __restrict__ fronts0, size_t stride, double rel, char* __restrict__ ctrl0) {
    const int z = blockIdx.y;
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    double m = 0.0;
    if (p < nnz) { const T v = cF * vF[p] + ce.cE[z] * vE[p]; fronts0[(size_t)z * stride + dest[p]] = v; m = abs1(v); }
    if (ctrl0) {
        double* pivfloor = reinterpret_cast<double*>(ctrl0 + (size_t)z * 64 + 8);
        for (int o = 32; o > 0; o >>= 1) m = fmax(m, __shfl_xor(m, o, 64));
        if ((threadIdx.x & 63) == 0 && m > 0.0) atomicMax(reinterpret_cast<unsigned long long*>(pivfloor), (unsigned long long)__double_as_longlong(rel * m));
    }
}

struct MfArgs {
    const int *first, *size, *bptr, *bidx, *cmap_ptr, *cmap, *child_ptr, *child_idx, *lvl_nodes;
    const int64_t *front_off, *inv_off, *upd_off;
};
// Batched sweeps (round 4): blockIdx.z selects one of up to MF_ZMAX factorisations of the SAME pencil (same elimination tree, different shift)
// with its own block of the work panel (z * wz doubles further) and of the update slab (z * uz) — the g independent solves of a fan group
// (engine.hip) share every launch instead of running side by side on g streams.
struct MfZ { const double* fronts[MF_ZMAX]; const double* inv[MF_ZMAX]; long wz, uz; };

// One workgroup per front of the level: extend-add the children's Schur complements, eliminate the
// s pivot columns (right-looking, no pivoting), then invert the two triangular diagonal blocks.
// Pivot growth of the pivot-free LU: the largest multiplier |l_ik| = |a_ik / pivot| met anywhere in the factorisation.  The reference
// factorises with pivoting (UMFPACK / CHOLMOD, blocklinear/backslash.jl:13); without it a tiny non-zero pivot gives huge multipliers
// and a silently inaccurate solve, so the growth is recorded (bit pattern of a non-negative double, atomic max) and judged by mf_check.
__device__ __forceinline__ void growth_commit(double g, unsigned long long* out) {
    for (int o = 32; o > 0; o >>= 1) g = fmax(g, __shfl_xor(g, o, 64));
    if ((threadIdx.x & 63) == 0 && g > 0.0) atomicMax(out, (unsigned long long)__double_as_longlong(g));
}
// static pivoting: a pivot below the floor is replaced by the floor with the pivot's direction (sign for real, phase for complex)
__device__ __forceinline__ double static_pivot(double piv, double fl) { return piv >= 0.0 ? fl : -fl; }
__device__ __forceinline__ cplx static_pivot(cplx piv, double fl) {
    const double a = sqrt(piv.re * piv.re + piv.im * piv.im);
    if (a == 0.0) return cplx{fl, 0.0};
    return cplx{piv.re / a * fl, piv.im / a * fl};
}
// Batched factorisations (round 4): blockIdx.y selects one of up to MF_ZMAX factors of the same pencil (same tree, different shift) — the ten
// factorisations of a Cyclic list share every level launch instead of running as ten chains on five streams (4 ms -> ~1 ms of the first time step
// at n = 5177).  ctrl = the factor's 64-byte control block: [0] growth (u64)  [8] floor, max|entry|  [24] err  [28] npert (mf_factor).
struct FactorZ { void* fronts[MF_ZMAX]; void* inv[MF_ZMAX]; void* ctrl[MF_ZMAX]; };
template <typename T>
__global__ void k_front_factor(MfArgs a, int lvl_begin, FactorZ fz, int use_floor) {
    T* __restrict__ fronts = (T*)fz.fronts[blockIdx.y];
    T* __restrict__ inv = (T*)fz.inv[blockIdx.y];
    char* const cb = (char*)fz.ctrl[blockIdx.y];
    unsigned long long* __restrict__ growth = (unsigned long long*)cb;
    const double* __restrict__ pivfloor = use_floor ? (const double*)(cb + 8) : nullptr;
    int* __restrict__ err = (int*)(cb + 24);
    int* __restrict__ npert = (int*)(cb + 28);
    const int t = a.lvl_nodes[lvl_begin + blockIdx.x];
    const int s = a.size[t], b = a.bptr[t + 1] - a.bptr[t], f = s + b;
    const int tid = threadIdx.x, nt = blockDim.x;
    T* F = fronts + a.front_off[t];
    for (int ci = a.child_ptr[t]; ci < a.child_ptr[t + 1]; ++ci) {
        const int c = a.child_idx[ci];
        const int sc = a.size[c], bc = a.bptr[c + 1] - a.bptr[c], fc = sc + bc;
        const T* Fc = fronts + a.front_off[c];
        const int* map = a.cmap + a.cmap_ptr[c];
        for (int id = tid; id < bc * bc; id += nt) {
            const int i = id % bc, j = id / bc;
            F[map[i] + (size_t)map[j] * f] += Fc[(sc + i) + (size_t)(sc + j) * fc];
        }
        __syncthreads();
    }
    double gmax = 0.0;
    for (int k = 0; k < s; ++k) {
        __syncthreads();
        T piv = F[k + (size_t)k * f];
        if (!(abs1(piv) == abs1(piv))) { if (tid == 0) *err = 1; return; }
        const double fl = pivfloor ? pivfloor[0] : 0.0;
        if (abs1(piv) < fl) {                       // (every thread takes the same branch)
            piv = static_pivot(piv, fl);
            __syncthreads();
            if (tid == 0) { F[k + (size_t)k * f] = piv; atomicAdd(npert, 1); }
        }
        if (abs1(piv) == 0.0) { if (tid == 0) *err = 1; return; }
        const T rp = recip(piv);
        for (int i = k + 1 + tid; i < f; i += nt) { const T l = F[i + (size_t)k * f] * rp; F[i + (size_t)k * f] = l; gmax = fmax(gmax, abs1(l)); }
        __syncthreads();
        const int nr = f - k - 1;
        for (int id = tid; id < nr * nr; id += nt) {
            const int i = k + 1 + id % nr, j = k + 1 + id / nr;
            F[i + (size_t)j * f] -= F[i + (size_t)k * f] * F[k + (size_t)j * f];
        }
    }
    growth_commit(gmax, growth);
    __syncthreads();
    T* Ti = inv + a.inv_off[t];
    for (int j = tid; j < s; j += nt) {
        // column j of inv(L11) (unit lower), stored strictly below the diagonal
        for (int i = j + 1; i < s; ++i) {
            T acc = F[i + (size_t)j * f];
            for (int k = j + 1; k < i; ++k) acc += F[i + (size_t)k * f] * Ti[k + (size_t)j * s];
            Ti[i + (size_t)j * s] = -acc;
        }
    }
    __syncthreads();
    for (int j = tid; j < s; j += nt) {
        // column j of inv(U11), stored on and above the diagonal
        T dj = recip(F[j + (size_t)j * f]);
        Ti[j + (size_t)j * s] = dj;
        for (int i = j - 1; i >= 0; --i) {
            T acc = make_scalar<T>(0.0, 0.0);
            for (int k = i + 1; k <= j; ++k) acc += F[i + (size_t)k * f] * Ti[k + (size_t)j * s];
            Ti[i + (size_t)j * s] = -(acc * recip(F[i + (size_t)i * f]));
        }
    }
}

// sum over the 16 consecutive lanes of a group (result in every lane of the group)
template <typename T> __device__ __forceinline__ T group16_sum(T v);
template <> __device__ __forceinline__ double group16_sum<double>(double v) {
    for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 16);
    return v;
}
template <> __device__ __forceinline__ cplx group16_sum<cplx>(cplx v) {
    for (int o = 8; o > 0; o >>= 1) { v.re += __shfl_xor(v.re, o, 16); v.im += __shfl_xor(v.im, o, 16); }
    return v;
}
// Blocked variant: panels of 16 pivot columns.  The column panel (f-kb) x 16 and the row panel 16 x (f-kb-16) live in LDS,
// so the trailing matrix in global memory is read and written once per PANEL (not once per pivot) and the per-pivot work
// runs at LDS latency.  The inverses of the triangular diagonal blocks are formed in LDS as well when s^2 entries fit
// (inv_lds).  Same arithmetic (right-looking LU without pivoting), same outputs as k_front_factor.
#define FF_NB 16
template <typename T>
__global__ __launch_bounds__(1024) void k_front_factor_blocked(MfArgs a, int lvl_begin, FactorZ fz, int inv_lds, int use_floor) {
    T* __restrict__ fronts = (T*)fz.fronts[blockIdx.y];
    T* __restrict__ inv = (T*)fz.inv[blockIdx.y];
    char* const cb = (char*)fz.ctrl[blockIdx.y];
    unsigned long long* __restrict__ growth = (unsigned long long*)cb;
    const double* __restrict__ pivfloor = use_floor ? (const double*)(cb + 8) : nullptr;
    int* __restrict__ err = (int*)(cb + 24);
    int* __restrict__ npert = (int*)(cb + 28);
    extern __shared__ double ffraw[];
    double gmax = 0.0;
    T* sm = reinterpret_cast<T*>(ffraw);
    __shared__ int bad;
    const int t = a.lvl_nodes[lvl_begin + blockIdx.x];
    const int s = a.size[t], b = a.bptr[t + 1] - a.bptr[t], f = s + b;
    const int tid = threadIdx.x, nt = blockDim.x;
    T* F = fronts + a.front_off[t];
    if (tid == 0) bad = 0;
    for (int ci = a.child_ptr[t]; ci < a.child_ptr[t + 1]; ++ci) {
        const int c = a.child_idx[ci];
        const int sc = a.size[c], bc = a.bptr[c + 1] - a.bptr[c], fc = sc + bc;
        const T* Fc = fronts + a.front_off[c];
        const int* map = a.cmap + a.cmap_ptr[c];
        for (int id = tid; id < bc * bc; id += nt) {
            const int i = id % bc, j = id / bc;
            F[map[i] + (size_t)map[j] * f] += Fc[(sc + i) + (size_t)(sc + j) * fc];
        }
        __syncthreads();
    }
    __syncthreads();
    for (int kb = 0; kb < s; kb += FF_NB) {
        const int jb = min(FF_NB, s - kb), rows = f - kb, nc = f - kb - jb;     // panel rows, trailing columns
        T* Pn = sm;                                  // rows x jb, ld rows:  F[kb + r, kb + c]
        T* U = sm + (size_t)rows * FF_NB;            // jb x nc, ld FF_NB:   F[kb + l, kb + jb + c]
        for (int id = tid; id < rows * jb; id += nt) { const int r = id % rows, c = id / rows; Pn[r + c * rows] = F[(kb + r) + (size_t)(kb + c) * f]; }
        for (int id = tid; id < jb * nc; id += nt) { const int l = id % jb, c = id / jb; U[l + c * FF_NB] = F[(kb + l) + (size_t)(kb + jb + c) * f]; }
        __syncthreads();
        // unblocked LU of the column panel
        for (int c = 0; c < jb; ++c) {
            T piv = Pn[c + c * rows];
            const double fl = pivfloor ? pivfloor[0] : 0.0;
            const bool small = abs1(piv) < fl;       // every thread sees the same value
            if (small) piv = static_pivot(piv, fl);
            if (abs1(piv) == 0.0 || !(abs1(piv) == abs1(piv))) bad = 1;
            const T rp = recip(piv);
            __syncthreads();
            if (small && tid == 0) { Pn[c + c * rows] = piv; atomicAdd(npert, 1); }
            for (int r = c + 1 + tid; r < rows; r += nt) { const T l = Pn[r + c * rows] * rp; Pn[r + c * rows] = l; gmax = fmax(gmax, abs1(l)); }
            __syncthreads();
            const int nr = rows - c - 1, ncc = jb - c - 1;
            for (int id = tid; id < nr * ncc; id += nt) {
                const int r = c + 1 + id % nr, c2 = c + 1 + id / nr;
                Pn[r + c2 * rows] -= Pn[r + c * rows] * Pn[c + c2 * rows];
            }
            __syncthreads();
        }
        if (bad) { if (tid == 0) *err = 1; return; }
        if (kb + FF_NB >= s) growth_commit(gmax, growth);       // last panel of this front
        // row panel: U12 = inv(L11) A12, one thread per column
        for (int c = tid; c < nc; c += nt) {
            T* u = U + c * FF_NB;
            for (int l = 1; l < jb; ++l) {
                T acc = u[l];
                for (int q = 0; q < l; ++q) acc -= Pn[l + q * rows] * u[q];
                u[l] = acc;
            }
        }
        __syncthreads();
        // write back the panel and the row panel, update the trailing matrix once
        for (int id = tid; id < rows * jb; id += nt) { const int r = id % rows, c = id / rows; F[(kb + r) + (size_t)(kb + c) * f] = Pn[r + c * rows]; }
        for (int id = tid; id < jb * nc; id += nt) { const int l = id % jb, c = id / jb; F[(kb + l) + (size_t)(kb + jb + c) * f] = U[l + c * FF_NB]; }
        for (int id = tid; id < nc * nc; id += nt) {
            const int r = id % nc, c = id / nc;
            T* dst = F + (size_t)(kb + jb + r) + (size_t)(kb + jb + c) * f;
            T a0 = *dst, a1 = make_scalar<T>(0.0, 0.0);
            const T* lrow = Pn + (jb + r);
            const T* ucol = U + c * FF_NB;
            int l = 0;
            for (; l + 1 < jb; l += 2) { a0 -= lrow[l * rows] * ucol[l]; a1 -= lrow[(l + 1) * rows] * ucol[l + 1]; }
            if (l < jb) a0 -= lrow[l * rows] * ucol[l];
            *dst = a0 + a1;
        }
        __syncthreads();
    }
    T* Ti = inv + a.inv_off[t];
    if (inv_lds) {
        // both triangular inverses from an LDS copy of F11 (L\U), one thread per column, results straight to global
        T* A11 = sm;                                   // s x s, ld s
        T* X = sm + (size_t)s * s;                     // s x s scratch for the inverse being built
        for (int id = tid; id < s * s; id += nt) A11[id] = F[(id % s) + (size_t)(id / s) * f];
        __syncthreads();
        // One column of an inverse per 16-lane group; the dot product of every substitution step is spread over the 16 lanes
        // (strided over k) and reduced with __shfl_xor inside the group, so a column costs s short steps instead of s^2/2
        // dependent multiply-adds of a single thread.
        const int grp = tid >> 4, l = tid & 15, ngrp = nt >> 4;
        for (int j = grp; j < s; j += ngrp) {          // column j of inv(L11) (unit lower), stored strictly below the diagonal
            for (int i = j + 1; i < s; ++i) {
                T acc = make_scalar<T>(0.0, 0.0);
                for (int k = j + 1 + l; k < i; k += 16) acc += A11[i + k * s] * X[k + j * s];
                acc = group16_sum<T>(acc);
                if (l == 0) {
                    const T v = -(acc + A11[i + j * s]);
                    X[i + j * s] = v;
                    Ti[i + (size_t)j * s] = v;
                }
            }
        }
        __syncthreads();
        for (int j = grp; j < s; j += ngrp) {          // column j of inv(U11), stored on and above the diagonal
            if (l == 0) {
                const T dj = recip(A11[j + j * s]);
                X[j + j * s] = dj;
                Ti[j + (size_t)j * s] = dj;
            }
            for (int i = j - 1; i >= 0; --i) {
                T acc = make_scalar<T>(0.0, 0.0);
                for (int k = i + 1 + l; k <= j; k += 16) acc += A11[i + k * s] * X[k + j * s];
                acc = group16_sum<T>(acc);
                if (l == 0) {
                    const T v = -(acc * recip(A11[i + i * s]));
                    X[i + j * s] = v;
                    Ti[i + (size_t)j * s] = v;
                }
            }
        }
        return;
    }
    for (int j = tid; j < s; j += nt) {
        for (int i = j + 1; i < s; ++i) {
            T acc = F[i + (size_t)j * f];
            for (int k = j + 1; k < i; ++k) acc += F[i + (size_t)k * f] * Ti[k + (size_t)j * s];
            Ti[i + (size_t)j * s] = -acc;
        }
    }
    __syncthreads();
    for (int j = tid; j < s; j += nt) {
        T dj = recip(F[j + (size_t)j * f]);
        Ti[j + (size_t)j * s] = dj;
        for (int i = j - 1; i >= 0; --i) {
            T acc = make_scalar<T>(0.0, 0.0);
            for (int k = i + 1; k <= j; ++k) acc += F[i + (size_t)k * f] * Ti[k + (size_t)j * s];
            Ti[i + (size_t)j * s] = -(acc * recip(F[i + (size_t)i * f]));
        }
    }
}

static MfArgs mf_args(const Pencil& P) {
    MfArgs a;
    a.first = P.dev.first.p; a.size = P.dev.size.p; a.bptr = P.dev.bptr.p; a.bidx = P.dev.bidx.p;
    a.cmap_ptr = P.dev.cmap_ptr.p; a.cmap = P.dev.cmap.p; a.child_ptr = P.dev.child_ptr.p; a.child_idx = P.dev.child_idx.p;
    a.lvl_nodes = P.dev.lvl_nodes.p; a.front_off = P.dev.front_off.p; a.inv_off = P.dev.inv_off.p; a.upd_off = P.dev.upd_off.p;
    return a;
}

template <typename T>
static void mf_factor_levels(Ctx* ctx, const Pencil& P, const FactorZ& fz, int nz);
template <typename T>
void mf_factor(Ctx* ctx, const Pencil& P, const double* valF, const double* valE, T cF, T cE, Factor<T>& out) {
    DRE_REQUIRE(P.has_device, "mf_factor: pencil has no device data");
    const Symbolic& S = P.sym;
    if (!out.fronts.p) out.fronts = DevArr<T>(ctx, (size_t)std::max<int64_t>(S.fronts_size, 1));
    if (!out.inv.p) out.inv = DevArr<T>(ctx, (size_t)std::max<int64_t>(S.inv_size, 1));
    TimedScope ts(ctx, sizeof(T) == 8 ? "mf_factor_real" : "mf_factor_complex", (double)sizeof(T) * 2.0 * S.fronts_size, 0);
    DRE_HIP(hipMemsetAsync(out.fronts.p, 0, (size_t)S.fronts_size * sizeof(T), ctx->stream));
    if (!out.err.p) {
        // the four small control words of a factor live in ONE 64-byte block (one memset per factorisation instead of four):
        //   [0] growth (u64)  [8] floor, max|entry| (2 doubles)  [24] err (int)  [28] npert (int)
        auto blk = std::make_shared<Buf>(ctx, 64);
        out.growth.buf = blk; out.growth.p = (unsigned long long*)blk->p; out.growth.n = 1;
        out.pivfloor.buf = blk; out.pivfloor.p = (double*)((char*)blk->p + 8); out.pivfloor.n = 2;
        out.err.buf = blk; out.err.p = (int*)((char*)blk->p + 24); out.err.n = 1;
        out.npert.buf = blk; out.npert.p = (int*)((char*)blk->p + 28); out.npert.n = 1;
    }
    out.topinv = Mat(); out.uses = 0;          // a new factorisation invalidates the dense top inverse
    DRE_HIP(hipMemsetAsync(out.growth.p, 0, 64, ctx->stream));
    out.nperturbed = -1;
    out.ref_valF = valF; out.ref_valE = valE; out.ref_cF = cF; out.ref_cE = cE;
    hipLaunchKernelGGL((k_assemble<T>), dim3(ceil_div(P.nnz, 256)), dim3(256), 0, ctx->stream, P.nnz, P.dev.asm_dest.p, valF, valE, cF, cE, out.fronts.p,
                       ctx->pivot_static, ctx->pivot_static > 0.0 ? out.pivfloor.p : (double*)nullptr);
    FactorZ fz; std::memset(&fz, 0, sizeof(fz));
    fz.fronts[0] = out.fronts.p; fz.inv[0] = out.inv.p; fz.ctrl[0] = out.growth.p;
    mf_factor_levels<T>(ctx, P, fz, 1);
    DRE_HIP(hipGetLastError());
}
template <typename T>
static void mf_factor_levels(Ctx* ctx, const Pencil& P, const FactorZ& fz, int nz) {
    const Symbolic& S = P.sym;
    const int use_floor = ctx->pivot_static > 0.0 ? 1 : 0;
    MfArgs a = mf_args(P);
    for (int l = S.nlevels - 1; l >= 0; --l) {
        const int nb = S.lvl_ptr[l + 1] - S.lvl_ptr[l];
        const int nt = P.lvl_maxfront[l] > 96 ? 1024 : 256;
        const int fmax = P.lvl_maxfront[l], smax = P.lvl_maxsep[l];
        const size_t panel_b = (size_t)2 * fmax * FF_NB * sizeof(T), inv_b = (size_t)2 * smax * smax * sizeof(T);
        const size_t lim = 150 * 1024;
        if (panel_b <= lim) {
            const int inv_lds = inv_b <= lim ? 1 : 0;
            const size_t shm = std::max(panel_b, inv_lds ? inv_b : (size_t)0);
            lds_attr(ctx, (const void*)k_front_factor_blocked<double>, 150 * 1024); lds_attr(ctx, (const void*)k_front_factor_blocked<cplx>, 150 * 1024);
            hipLaunchKernelGGL((k_front_factor_blocked<T>), dim3(nb, nz), dim3(fmax > 64 ? 1024 : 256), shm, ctx->stream, a, S.lvl_ptr[l], fz, inv_lds, use_floor);
        } else {
            hipLaunchKernelGGL((k_front_factor<T>), dim3(nb, nz), dim3(nt), 0, ctx->stream, a, S.lvl_ptr[l], fz, use_floor);
        }
    }
}
// nz real factorisations  M_z = cF F' + cE_z E'  of the same pencil in shared launches (assembly per factor, every tree level once)
template <typename T>
void mf_factor_batch(Ctx* ctx, const Pencil& P, const double* valF, const double* valE, T cF, const T* cE, Factor<T>* const* outs, int nz) {
    DRE_REQUIRE(P.has_device && nz >= 1 && nz <= MF_ZMAX, "mf_factor_batch: batch size");
    const Symbolic& S = P.sym;
    TimedScope ts(ctx, sizeof(T) == 8 ? "mf_factor_real" : "mf_factor_complex", (double)sizeof(T) * 2.0 * S.fronts_size * nz, 0, nz);
    FactorZ fz; std::memset(&fz, 0, sizeof(fz));
    // Fresh factors (the usual case: a whole batch of new shifts) share ONE allocation for their fronts and ONE for their control blocks: one
    // memset each and one assembly launch for the batch instead of three memsets and a launch per factor (40 host-paced launches in front of the
    // ten factorisations of a Cyclic list: 160 us of the first time step at n = 371).
    bool fresh = nz >= 2 && MF_ZMAX <= 16;
    for (int z = 0; z < nz; ++z) fresh = fresh && !outs[z]->fronts.p && !outs[z]->err.p;
    if (fresh) {
        const size_t stride = (size_t)std::max<int64_t>(S.fronts_size, 1);
        DevArr<T> all(ctx, stride * nz);
        auto ctrl = std::make_shared<Buf>(ctx, (size_t)64 * nz);
        DRE_HIP(hipMemsetAsync(all.p, 0, stride * nz * sizeof(T), ctx->stream));
        DRE_HIP(hipMemsetAsync(ctrl->p, 0, (size_t)64 * nz, ctx->stream));
        AssembleZ<T> ce;
        for (int z = 0; z < MF_ZMAX; ++z) ce.cE[z] = cE[z < nz ? z : 0];
        for (int z = 0; z < nz; ++z) {
            Factor<T>& out = *outs[z];
            out.fronts.buf = all.buf; out.fronts.p = all.p + (size_t)z * stride; out.fronts.n = stride;
            out.inv = DevArr<T>(ctx, (size_t)std::max<int64_t>(S.inv_size, 1));
            char* blk = (char*)ctrl->p + (size_t)64 * z;
            out.growth.buf = ctrl; out.growth.p = (unsigned long long*)blk; out.growth.n = 1;
            out.pivfloor.buf = ctrl; out.pivfloor.p = (double*)(blk + 8); out.pivfloor.n = 2;
            out.err.buf = ctrl; out.err.p = (int*)(blk + 24); out.err.n = 1;
            out.npert.buf = ctrl; out.npert.p = (int*)(blk + 28); out.npert.n = 1;
            out.topinv = Mat(); out.uses = 0;
            out.nperturbed = -1;
            out.ref_valF = valF; out.ref_valE = valE; out.ref_cF = cF; out.ref_cE = cE[z];
            fz.fronts[z] = out.fronts.p; fz.inv[z] = out.inv.p; fz.ctrl[z] = out.growth.p;
        }
        hipLaunchKernelGGL((k_assemble_z<T>), dim3(ceil_div(P.nnz, 256), nz), dim3(256), 0, ctx->stream, P.nnz, P.dev.asm_dest.p, valF, valE, cF, ce, all.p, stride,
                           ctx->pivot_static, ctx->pivot_static > 0.0 ? (char*)ctrl->p : (char*)nullptr);
        mf_factor_levels<T>(ctx, P, fz, nz);
        DRE_HIP(hipGetLastError());
        return;
    }
    for (int z = 0; z < nz; ++z) {
        Factor<T>& out = *outs[z];
        if (!out.fronts.p) out.fronts = DevArr<T>(ctx, (size_t)std::max<int64_t>(S.fronts_size, 1));
        if (!out.inv.p) out.inv = DevArr<T>(ctx, (size_t)std::max<int64_t>(S.inv_size, 1));
        DRE_HIP(hipMemsetAsync(out.fronts.p, 0, (size_t)S.fronts_size * sizeof(T), ctx->stream));
        if (!out.err.p) {
            auto blk = std::make_shared<Buf>(ctx, 64);
            out.growth.buf = blk; out.growth.p = (unsigned long long*)blk->p; out.growth.n = 1;
            out.pivfloor.buf = blk; out.pivfloor.p = (double*)((char*)blk->p + 8); out.pivfloor.n = 2;
            out.err.buf = blk; out.err.p = (int*)((char*)blk->p + 24); out.err.n = 1;
            out.npert.buf = blk; out.npert.p = (int*)((char*)blk->p + 28); out.npert.n = 1;
        }
        out.topinv = Mat(); out.uses = 0;
        DRE_HIP(hipMemsetAsync(out.growth.p, 0, 64, ctx->stream));
        out.nperturbed = -1;
        out.ref_valF = valF; out.ref_valE = valE; out.ref_cF = cF; out.ref_cE = cE[z];
        hipLaunchKernelGGL((k_assemble<T>), dim3(ceil_div(P.nnz, 256)), dim3(256), 0, ctx->stream, P.nnz, P.dev.asm_dest.p, valF, valE, cF, cE[z], out.fronts.p,
                           ctx->pivot_static, ctx->pivot_static > 0.0 ? out.pivfloor.p : (double*)nullptr);
        fz.fronts[z] = out.fronts.p; fz.inv[z] = out.inv.p; fz.ctrl[z] = out.growth.p;
    }
    mf_factor_levels<T>(ctx, P, fz, nz);
    DRE_HIP(hipGetLastError());
}
template void mf_factor_batch<double>(Ctx*, const Pencil&, const double*, const double*, double, const double*, Factor<double>* const*, int);
template void mf_factor_batch<cplx>(Ctx*, const Pencil&, const double*, const double*, cplx, const cplx*, Factor<cplx>* const*, int);
template <typename T>
double mf_check(Ctx* ctx, const Factor<T>& F) {
    if (!F.err.p) return 0.0;
    unsigned long long w[4] = {0, 0, 0, 0};       // the factor's control block (mf_factor): growth | floor, max|entry| | err, npert — one copy
    DRE_HIP(hipMemcpyAsync(w, F.growth.p, sizeof(w), hipMemcpyDeviceToHost, ctx->stream));
    DRE_HIP(hipStreamSynchronize(ctx->stream));
    const int herr = (int)(w[3] & 0xffffffffull);
    F.nperturbed = (int)(w[3] >> 32);
    if (herr) throw Error(ERR_SINGULAR, "mf_factor: zero or NaN pivot (shifted operator numerically singular)");
    double g;
    std::memcpy(&g, &w[0], sizeof(g));
    if (!(g == g) || g > ctx->pivot_growth_fail)
        throw Error(ERR_SINGULAR, "mf_factor: pivot growth " + std::to_string(g) + " of the pivot-free LU exceeds the limit (pivot_growth_fail): the shifted operator "
                                  "needs pivoting; use a user block solver (dre_adi_options.inner_solve) for this pencil");
    return g;
}
// the same for several factors with ONE synchronisation AND one copy (set-up of a whole Cyclic list): the 64-byte control blocks are
// gathered by a tiny kernel (three separate 4/8-byte copies per factor cost ~19 us each in the first time step's critical path)
struct MetaPtrs { const unsigned long long* p[16]; };
__global__ void k_gather_meta(MetaPtrs m, int nf, unsigned long long* __restrict__ out) {
    const int i = threadIdx.x;
    if (i < nf * 8) out[i] = m.p[i >> 3][i & 7];
}
std::vector<double> mf_check_batch(Ctx* ctx, const std::vector<const Factor<double>*>& fs) {
    const size_t nf = fs.size();
    std::vector<double> out(nf, 0.0);
    for (size_t b0 = 0; b0 < nf; b0 += 16) {
        const int nb = (int)std::min<size_t>(16, nf - b0);
        MetaPtrs mp;
        for (int i = 0; i < 16; ++i) mp.p[i] = fs[b0 + (size_t)(i < nb ? i : 0)]->growth.p;      // start of the control block (mf_factor)
        DevArr<unsigned long long> g(ctx, 16 * 8);
        hipLaunchKernelGGL(k_gather_meta, dim3(1), dim3(128), 0, ctx->stream, mp, nb, g.p);
        unsigned long long h[16 * 8];
        DRE_HIP(hipMemcpyAsync(h, g.p, (size_t)nb * 64, hipMemcpyDeviceToHost, ctx->stream));
        DRE_HIP(hipStreamSynchronize(ctx->stream));
        for (int i = 0; i < nb; ++i) {
            const unsigned long long* w = h + 8 * i;       // [0] growth, [1..2] floor / max|entry|, [3] = err (low word) | npert (high word)
            const int herr = (int)(w[3] & 0xffffffffull), hnp = (int)(w[3] >> 32);
            if (herr) throw Error(ERR_SINGULAR, "mf_factor: zero or NaN pivot (shifted operator numerically singular)");
            double gr;
            std::memcpy(&gr, &w[0], sizeof(gr));
            if (!(gr == gr) || gr > ctx->pivot_growth_fail)
                throw Error(ERR_SINGULAR, "mf_factor: pivot growth " + std::to_string(gr) + " of the LU exceeds the limit (pivot_growth_fail)");
            fs[b0 + (size_t)i]->nperturbed = hnp;
            out[b0 + (size_t)i] = gr;
        }
    }
    return out;
}
template double mf_check<double>(Ctx*, const Factor<double>&);
template double mf_check<cplx>(Ctx*, const Factor<cplx>&);
template void mf_factor<double>(Ctx*, const Pencil&, const double*, const double*, double, double, Factor<double>&);
template void mf_factor<cplx>(Ctx*, const Pencil&, const double*, const double*, cplx, cplx, Factor<cplx>&);

// =============================================================================================
// Multi-RHS triangular solves, one launch per tree level; a workgroup owns (front, slice of KC columns).
// =============================================================================================
#define MF_KC 8

template <typename T>
__global__ __launch_bounds__(256) void k_mf_forward(MfArgs a, int lvl_begin, const T* __restrict__ fronts,
                                                    const T* __restrict__ inv, T* __restrict__ W, int ldw, int nrhs,
                                                    T* __restrict__ upd, int64_t ldu, const AdiState* st) {
    if (st && st->done) return;
    extern __shared__ double smraw[];
    T* sm = reinterpret_cast<T*>(smraw);
    const int t = a.lvl_nodes[lvl_begin + blockIdx.x];
    const int s = a.size[t], b = a.bptr[t + 1] - a.bptr[t], f = s + b, first = a.first[t];
    const int c0 = blockIdx.y * MF_KC, kc = min(MF_KC, nrhs - c0);
    const int tid = threadIdx.x, nt = blockDim.x;
    T* w = sm;               // f x kc
    T* y = sm + (size_t)f * MF_KC;   // s x kc
    const T* F = fronts + a.front_off[t];
    const T* Ti = inv + a.inv_off[t];
    for (int id = tid; id < f * kc; id += nt) {
        const int i = id % f, c = id / f;
        w[i + c * f] = (i < s) ? W[(first + i) + (size_t)(c0 + c) * ldw] : make_scalar<T>(0.0, 0.0);
    }
    __syncthreads();
    for (int ci = a.child_ptr[t]; ci < a.child_ptr[t + 1]; ++ci) {
        const int ch = a.child_idx[ci];
        const int bc = a.bptr[ch + 1] - a.bptr[ch];
        const int* map = a.cmap + a.cmap_ptr[ch];
        const T* uc = upd + a.upd_off[ch];
        for (int id = tid; id < bc * kc; id += nt) {
            const int i = id % bc, c = id / bc;
            w[map[i] + c * f] += uc[i + (size_t)(c0 + c) * ldu];
        }
        __syncthreads();
    }
    for (int id = tid; id < s * kc; id += nt) {
        const int i = id % s, c = id / s;
        T a0 = w[i + c * f], a1 = make_scalar<T>(0.0, 0.0), a2 = a1, a3 = a1;
        int k = 0;
        for (; k + 3 < i; k += 4) {       // four independent chains: a dependent f64 FMA costs 32 cycles on gfx950
            a0 += Ti[i + (size_t)k * s] * w[k + c * f];
            a1 += Ti[i + (size_t)(k + 1) * s] * w[k + 1 + c * f];
            a2 += Ti[i + (size_t)(k + 2) * s] * w[k + 2 + c * f];
            a3 += Ti[i + (size_t)(k + 3) * s] * w[k + 3 + c * f];
        }
        for (; k < i; ++k) a0 += Ti[i + (size_t)k * s] * w[k + c * f];
        const T acc = (a0 + a1) + (a2 + a3);
        y[i + c * s] = acc;
        W[(first + i) + (size_t)(c0 + c) * ldw] = acc;
    }
    __syncthreads();
    T* ut = upd + a.upd_off[t];
    for (int id = tid; id < b * kc; id += nt) {
        const int i = id % b, c = id / b;
        T a0 = w[s + i + c * f], a1 = make_scalar<T>(0.0, 0.0), a2 = a1, a3 = a1;
        int k = 0;
        for (; k + 3 < s; k += 4) {
            a0 -= F[(s + i) + (size_t)k * f] * y[k + c * s];
            a1 -= F[(s + i) + (size_t)(k + 1) * f] * y[k + 1 + c * s];
            a2 -= F[(s + i) + (size_t)(k + 2) * f] * y[k + 2 + c * s];
            a3 -= F[(s + i) + (size_t)(k + 3) * f] * y[k + 3 + c * s];
        }
        for (; k < s; ++k) a0 -= F[(s + i) + (size_t)k * f] * y[k + c * s];
        ut[i + (size_t)(c0 + c) * ldu] = (a0 + a1) + (a2 + a3);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void k_mf_backward(MfArgs a, int lvl_begin, const T* __restrict__ fronts,
                                                     const T* __restrict__ inv, T* __restrict__ W, int ldw, int nrhs,
                                                     const AdiState* st) {
    if (st && st->done) return;
    extern __shared__ double smraw[];
    T* sm = reinterpret_cast<T*>(smraw);
    const int t = a.lvl_nodes[lvl_begin + blockIdx.x];
    const int s = a.size[t], b = a.bptr[t + 1] - a.bptr[t], f = s + b, first = a.first[t];
    const int c0 = blockIdx.y * MF_KC, kc = min(MF_KC, nrhs - c0);
    const int tid = threadIdx.x, nt = blockDim.x;
    T* xb = sm;                       // b x kc
    T* z = sm + (size_t)b * MF_KC;    // s x kc
    const T* F = fronts + a.front_off[t];
    const T* Ti = inv + a.inv_off[t];
    const int* B = a.bidx + a.bptr[t];
    for (int id = tid; id < b * kc; id += nt) {
        const int i = id % b, c = id / b;
        xb[i + c * b] = W[B[i] + (size_t)(c0 + c) * ldw];
    }
    __syncthreads();
    for (int id = tid; id < s * kc; id += nt) {
        const int i = id % s, c = id / s;
        T a0 = W[(first + i) + (size_t)(c0 + c) * ldw], a1 = make_scalar<T>(0.0, 0.0), a2 = a1, a3 = a1;
        int k = 0;
        for (; k + 3 < b; k += 4) {
            a0 -= F[i + (size_t)(s + k) * f] * xb[k + c * b];
            a1 -= F[i + (size_t)(s + k + 1) * f] * xb[k + 1 + c * b];
            a2 -= F[i + (size_t)(s + k + 2) * f] * xb[k + 2 + c * b];
            a3 -= F[i + (size_t)(s + k + 3) * f] * xb[k + 3 + c * b];
        }
        for (; k < b; ++k) a0 -= F[i + (size_t)(s + k) * f] * xb[k + c * b];
        z[i + c * s] = (a0 + a1) + (a2 + a3);
    }
    __syncthreads();
    for (int id = tid; id < s * kc; id += nt) {
        const int i = id % s, c = id / s;
        T a0 = make_scalar<T>(0.0, 0.0), a1 = a0, a2 = a0, a3 = a0;
        int k = i;
        for (; k + 3 < s; k += 4) {
            a0 += Ti[i + (size_t)k * s] * z[k + c * s];
            a1 += Ti[i + (size_t)(k + 1) * s] * z[k + 1 + c * s];
            a2 += Ti[i + (size_t)(k + 2) * s] * z[k + 2 + c * s];
            a3 += Ti[i + (size_t)(k + 3) * s] * z[k + 3 + c * s];
        }
        for (; k < s; ++k) a0 += Ti[i + (size_t)k * s] * z[k + c * s];
        W[(first + i) + (size_t)(c0 + c) * ldw] = (a0 + a1) + (a2 + a3);
    }
}

// ---------------------------------------------------------------------------------------------
// Real sweeps on the matrix cores: a workgroup owns (front, 16 right-hand-side columns); every 16 x 16 output tile of
// y = inv(L11) w_S, upd = w_B - L21 y (forward) and z = w_S - U12 x_B, x = inv(U11) z (backward) is a chain of
// v_mfma_f64_16x16x4 with the factor entries streamed from HBM/L2 straight into the A operand (coalesced 128 B rows,
// eight loads in flight) and the right-hand-side panel staged in LDS as the B operand.
// ---------------------------------------------------------------------------------------------
typedef double mf_v4d __attribute__((ext_vector_type(4)));
#define MFM_KC 16

// Workgroup -> (front of the level, block of 16 right-hand-side columns, shift).  The column blocks of one (front, shift) read the SAME factor
// entries; in launch order they are gridDim.x workgroups apart and land on different XCDs (round-robin placement: blocks b and b + 8 share
// one), so every XCD's L2 fetches the front for itself.  swz != 0: every XCD takes a contiguous chunk of the list ordered (shift, front,
// column block) — the column blocks of a front are resident together on one XCD, and with 8 shifts an XCD holds (mostly) ONE shift's factors.
// A speed matter only (placement is not a contract); bijective for any grid.
__device__ __forceinline__ void mf_block(int swz, int& bx, int& by, int& bz) {
    if (!swz) { bx = blockIdx.x; by = blockIdx.y; bz = blockIdx.z; return; }
    const unsigned gx = gridDim.x, gy = gridDim.y, per = gx * gy, T = per * gridDim.z;
    unsigned L = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
    const unsigned xcd = L & 7u, slot = L >> 3, q = T >> 3, r = T & 7u;
    L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    const unsigned z = L / per, rem = L - z * per, x = rem / gy;
    bz = (int)z; bx = (int)x; by = (int)(rem - x * gy);
}

// THROUGHPUT form (the forward kernel, and the backward kernel of levels with many fronts: thousands of workgroups, the memory round trips
// of one workgroup hide behind the others): predicated loads, so masked lanes (outside the triangle / past the end of the K range) generate
// no traffic, element-per-thread staging, fewer registers.  The LATENCY form of the backward kernel further down wins where a level has few
// fronts (measured at n = 20209, 10 column blocks: up to ~1300 workgroups 16 vs 22 us, 320 workgroups 10 vs 28 us; at the 504-leaf level the
// throughput form is 34 vs 71 us).
// acc += A(rows r0.., K range [kbeg, kend)) * Bs   with A(row, k) = sign * Aglob[row + k * lda] where keep(row, k), else 0
template <typename Keep>
__device__ __forceinline__ mf_v4d mfma_rowtile(mf_v4d acc, const double* __restrict__ Aglob, int lda, int r0, int kbeg, int kend, double sign,
                                               const double* __restrict__ Bs, int ldb, Keep keep) {
    const int lane = threadIdx.x & 63, lr = lane & 15, lk = lane >> 4;
    const int row = r0 + lr;
    for (int k0 = kbeg; k0 < kend; k0 += 32) {
        double av[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int k = k0 + 4 * u + lk;
            av[u] = (k < kend && keep(row, k)) ? Aglob[row + (size_t)k * lda] : 0.0;
        }
        // the sign is applied when the value is consumed: `sign * load` inside the predicated region made the compiler wait for every
        // single load (eight memory round trips in a row per batch)
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int k = k0 + 4 * u + lk;
            if (k0 + 4 * u < kend) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(sign * av[u], Bs[k + lr * ldb], acc, 0, 0, 0);
        }
    }
    return acc;
}

// Win / ldwin / nin: the first nin right-hand-side columns are READ from Win (the caller's residual block) instead of W, so that the
// caller does not have to copy them into the work panel first; everything is written to W (every row of the panel belongs to exactly one
// front or to the dense top, so the sweeps write all of it).
__global__ __launch_bounds__(1024) void k_mf_forward_mfma(MfArgs a, int lvl_begin, MfZ zb,
                                                          double* __restrict__ W, int ldw, int nrhs, double* __restrict__ upd, int64_t ldu,
                                                          const AdiState* st, const double* __restrict__ Win, int ldwin, int nin, int swz) {
    if (st && st->done) return;
    extern __shared__ double sm[];
    int bx, by, bz;
    mf_block(swz, bx, by, bz);
    const double* __restrict__ fronts = zb.fronts[bz];
    const double* __restrict__ inv = zb.inv[bz];
    W += (size_t)bz * zb.wz; upd += (size_t)bz * zb.uz;
    const int t = a.lvl_nodes[lvl_begin + bx];
    const int s = a.size[t], b = a.bptr[t + 1] - a.bptr[t], f = s + b, first = a.first[t];
    const int c0 = by * MFM_KC, kc = min(MFM_KC, nrhs - c0);
    const int tid = threadIdx.x, nt = blockDim.x, wave = tid >> 6, lane = tid & 63, nw = nt >> 6;
    const int sp = (s + 15) & ~15, wr = max(s + ((b + 15) & ~15), sp) + 4, ldl = wr | 1, ldy = (sp + 4) | 1;
    double* w = sm;                          // wr x 16 (ld ldl): rows 0..s-1 = w_S, s..f-1 = assembled w_B, rest 0
    double* y = sm + (size_t)ldl * MFM_KC;   // (sp + 4) x 16 (ld ldy), zero padded
    const double* F = fronts + a.front_off[t];
    const double* Ti = inv + a.inv_off[t];
    // staging, four elements per thread and pass: the four (predicated) loads are issued together and waited for once — one load per pass
    // made every pass a memory round trip of its own (ldl * 16 / nt = 6..10 in a row on a narrow level)
    for (int id0 = tid; id0 < ldl * MFM_KC; id0 += 4 * nt) {
        double v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int id = id0 + q * nt, i = id % ldl, c = id / ldl;
            const double* __restrict__ src = (c0 + c < nin) ? Win + (size_t)(c0 + c) * ldwin : W + (size_t)(c0 + c) * ldw;
            v[q] = (i < s && c < kc) ? src[first + i] : 0.0;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) { const int id = id0 + q * nt; if (id < ldl * MFM_KC) w[id] = v[q]; }
    }
    for (int id = tid; id < ldy * MFM_KC; id += nt) y[id] = 0.0;
    __syncthreads();
    for (int ci = a.child_ptr[t]; ci < a.child_ptr[t + 1]; ++ci) {
        const int ch = a.child_idx[ci];
        const int bc = a.bptr[ch + 1] - a.bptr[ch];
        const int* map = a.cmap + a.cmap_ptr[ch];
        const double* uc = upd + a.upd_off[ch];
        for (int id0 = tid; id0 < bc * kc; id0 += 4 * nt) {
            double v[4];
            int mi[4], cc[4];
            // (no arithmetic on a loaded value inside its predicate: the eight loads go out together)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int id = min(id0 + q * nt, bc * kc - 1), i = id % bc;
                cc[q] = id / bc;
                v[q] = uc[i + (size_t)(c0 + cc[q]) * ldu];
                mi[q] = map[i];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) if (id0 + q * nt < bc * kc) w[mi[q] + cc[q] * ldl] += v[q];
        }
        __syncthreads();
    }
    const int lr = lane & 15, lq = lane >> 4;
    // y = inv(L11) w_S :  unit diagonal (accumulator starts at w), strictly lower part of Ti
    for (int rt = wave; rt * 16 < s; rt += nw) {
        const int r0 = rt * 16;
        mf_v4d acc;
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] = w[(r0 + lq + 4 * r) + lr * ldl];
        acc = mfma_rowtile(acc, Ti, s, r0, 0, min(s, r0 + 16), 1.0, w, ldl, [s](int row, int k) { return row < s && k < row; });
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = r0 + lq + 4 * r;
            if (row < s) {
                y[row + lr * ldy] = acc[r];
                if (lr < kc) W[(first + row) + (size_t)(c0 + lr) * ldw] = acc[r];
            }
        }
    }
    __syncthreads();
    // upd = w_B - L21 y
    double* ut = upd + a.upd_off[t];
    for (int rt = wave; rt * 16 < b; rt += nw) {
        const int r0 = rt * 16;
        mf_v4d acc;
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] = w[(s + r0 + lq + 4 * r) + lr * ldl];
        acc = mfma_rowtile(acc, F + s, f, r0, 0, s, -1.0, y, ldy, [b](int row, int) { return row < b; });
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = r0 + lq + 4 * r;
            if (row < b && lr < kc) ut[row + (size_t)(c0 + lr) * ldu] = acc[r];
        }
    }
}

__global__ __launch_bounds__(1024) void k_mf_backward_tp(MfArgs a, int lvl_begin, MfZ zb,
                                                           double* __restrict__ W, int ldw, int nrhs, const AdiState* st, int swz) {
    if (st && st->done) return;
    extern __shared__ double sm[];
    int bx, by, bz;
    mf_block(swz, bx, by, bz);
    const double* __restrict__ fronts = zb.fronts[bz];
    const double* __restrict__ inv = zb.inv[bz];
    W += (size_t)bz * zb.wz;
    const int t = a.lvl_nodes[lvl_begin + bx];
    const int s = a.size[t], b = a.bptr[t + 1] - a.bptr[t], f = s + b, first = a.first[t];
    const int c0 = by * MFM_KC, kc = min(MFM_KC, nrhs - c0);
    const int tid = threadIdx.x, nt = blockDim.x, wave = tid >> 6, lane = tid & 63, nw = nt >> 6;
    const int sp = (s + 15) & ~15, ldx = (b + 36) | 1, ldz = (sp + 36) | 1;
    double* xb = sm;                          // (b + pad) x 16: the already known ancestor unknowns, zero padded
    double* z = sm + (size_t)ldx * MFM_KC;    // (sp + pad) x 16
    const double* F = fronts + a.front_off[t];
    const double* Ti = inv + a.inv_off[t];
    const int* B = a.bidx + a.bptr[t];
    // staging: four predicated loads in flight per thread and pass (see the forward kernel)
    for (int id0 = tid; id0 < ldx * MFM_KC; id0 += 4 * nt) {
        double v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int id = id0 + q * nt, i = id % ldx, c = id / ldx;
            v[q] = (i < b && c < kc) ? W[B[i] + (size_t)(c0 + c) * ldw] : 0.0;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) { const int id = id0 + q * nt; if (id < ldx * MFM_KC) xb[id] = v[q]; }
    }
    for (int id0 = tid; id0 < ldz * MFM_KC; id0 += 4 * nt) {
        double v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int id = id0 + q * nt, i = id % ldz, c = id / ldz;
            v[q] = (i < s && c < kc) ? W[(first + i) + (size_t)(c0 + c) * ldw] : 0.0;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) { const int id = id0 + q * nt; if (id < ldz * MFM_KC) z[id] = v[q]; }
    }
    __syncthreads();
    const int lr = lane & 15, lq = lane >> 4;
    // z = w_S - U12 x_B   (in place in LDS: every tile reads and writes only its own rows of z)
    for (int rt = wave; rt * 16 < s; rt += nw) {
        const int r0 = rt * 16;
        mf_v4d acc;
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] = z[(r0 + lq + 4 * r) + lr * ldz];
        acc = mfma_rowtile(acc, F + (size_t)s * f, f, r0, 0, b, -1.0, xb, ldx, [s](int row, int) { return row < s; });
#pragma unroll
        for (int r = 0; r < 4; ++r) z[(r0 + lq + 4 * r) + lr * ldz] = acc[r];
    }
    __syncthreads();
    // x_S = inv(U11) z : upper triangle of Ti including the diagonal
    for (int rt = wave; rt * 16 < s; rt += nw) {
        const int r0 = rt * 16;
        mf_v4d acc = {0.0, 0.0, 0.0, 0.0};
        acc = mfma_rowtile(acc, Ti, s, r0, r0, s, 1.0, z, ldz, [s](int row, int k) { return row < s && k >= row; });
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = r0 + lq + 4 * r;
            if (row < s && lr < kc) W[(first + row) + (size_t)(c0 + lr) * ldw] = acc[r];
        }
    }
}

// LATENCY form (backward levels with few fronts).
// A operand of a 16-row tile: element (row, k) = A[row + k * lda]; addressable for row < nrow and k < kcap (both >= 1).
struct RtSrc { const double* A; int lda, nrow, kcap; };
// One batch = 8 K-steps (32 columns of the factor block) of the tile at rows r0..r0+15, starting at column k0.  BRANCH-FREE: every lane
// loads from a clamped, always valid address and the predicate (range, triangle, sign) is applied when the value is consumed (rt_mma) —
// with predicated loads the compiler waits for each load before it issues the next one (the in-order vmcnt counter cannot express "maybe
// issued"), which made a batch cost eight memory round trips instead of one.
__device__ __forceinline__ void rt_load(double (&raw)[8], const RtSrc& s, int r0, int k0, int kend) {
    const int lane = threadIdx.x & 63, lr = lane & 15, lk = lane >> 4;
    const double* __restrict__ p = s.A + min(r0 + lr, s.nrow - 1);
#pragma unroll
    for (int u = 0; u < 8; ++u)
        if (k0 + 4 * u < kend) raw[u] = p[(size_t)min(k0 + 4 * u + lk, s.kcap - 1) * s.lda];       // wave-uniform guard: K-steps past the end are not requested
}
// acc += sign * A(tile rows, [k0, k0 + 32) below kend, where keep(row, k)) * Bs
template <typename Keep>
__device__ __forceinline__ mf_v4d rt_mma(mf_v4d acc, const double (&raw)[8], int r0, int k0, int kend, double sign, const double* __restrict__ Bs, int ldb,
                                         Keep keep) {
    const int lane = threadIdx.x & 63, lr = lane & 15, lk = lane >> 4;
    const int row = r0 + lr;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int k = k0 + 4 * u + lk;
        if (k0 + 4 * u < kend) {
            const double v = (k < kend && keep(row, k)) ? sign * raw[u] : 0.0;
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(v, Bs[k + lr * ldb], acc, 0, 0, 0);
        }
    }
    return acc;
}
// The whole K range [kbeg, kend) of one tile, software pipelined: `cur` holds the batch at kbeg (issued by the caller, as early as it
// likes); while a batch multiplies the next one is in flight (two register sets, no copies).
template <typename Keep>
__device__ __forceinline__ mf_v4d rt_tile(mf_v4d acc, const RtSrc& s, int r0, int kbeg, int kend, double sign, const double* __restrict__ Bs, int ldb,
                                          Keep keep, double (&cur)[8]) {
    double nxt[8];
    for (int k0 = kbeg; k0 < kend; k0 += 64) {
        rt_load(nxt, s, r0, k0 + 32, kend);
        acc = rt_mma(acc, cur, r0, k0, kend, sign, Bs, ldb, keep);
        rt_load(cur, s, r0, k0 + 64, kend);
        acc = rt_mma(acc, nxt, r0, k0 + 32, kend, sign, Bs, ldb, keep);
    }
    return acc;
}

// Backward: the long product is z = w_S - U12 x_B (K = b, the boundary: up to several hundred near the top of the tree) on only
// ceil(s / 16) row tiles, so the waves of a workgroup SPLIT K: G = nw / tiles waves per tile, each over a 32-aligned share of the
// boundary, partial tiles summed in a fixed order through LDS (`part`, nw x 256 doubles behind z).
__global__ __launch_bounds__(1024) void k_mf_backward_mfma(MfArgs a, int lvl_begin, MfZ zb,
                                                           double* __restrict__ W, int ldw, int nrhs, const AdiState* st, int split, int swz) {
    if (st && st->done) return;
    extern __shared__ double sm[];
    int bx, by, bz;
    mf_block(swz, bx, by, bz);
    const double* __restrict__ fronts = zb.fronts[bz];
    const double* __restrict__ inv = zb.inv[bz];
    W += (size_t)bz * zb.wz;
    const int t = a.lvl_nodes[lvl_begin + bx];
    const int s = a.size[t], b = a.bptr[t + 1] - a.bptr[t], f = s + b, first = a.first[t];
    const int c0 = by * MFM_KC, kc = min(MFM_KC, nrhs - c0);
    const int tid = threadIdx.x, nt = blockDim.x, wave = tid >> 6, lane = tid & 63, nw = nt >> 6;
    const int sp = (s + 15) & ~15, ldx = (b + 36) | 1, ldz = (sp + 36) | 1;
    double* xb = sm;                          // (b + pad) x 16: the already known ancestor unknowns, zero padded
    double* z = sm + (size_t)ldx * MFM_KC;    // (sp + pad) x 16
    double* part = z + (size_t)ldz * MFM_KC;  // nw x 256: K-split partial tiles
    const double* F = fronts + a.front_off[t];
    const double* Ti = inv + a.inv_off[t];
    const int* B = a.bidx + a.bptr[t];
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const int nts = sp >> 4, G = split ? max(1, nw / nts) : 1;        // split = 0: no room for the partial tiles in LDS (huge fronts)
    // split form (G >= 2): wave -> (tile, share of K);  plain form: wave -> tiles wave, wave + nw, ...
    const int ztile = G >= 2 ? wv / G : wv, zpart = G >= 2 ? wv - ztile * G : 0;
    const int kshare = G >= 2 ? ((((b + G - 1) / G) + 31) & ~31) : b;
    const int zk0 = zpart * kshare, zk1 = min(b, zk0 + kshare);
    const bool zwork = ztile < nts && zk0 < zk1;
    const RtSrc sz{F + (size_t)s * f, f, s, max(b, 1)}, sx{Ti, s, s, s};
    double za[8], xa[8];
    if (zwork) rt_load(za, sz, ztile * 16, zk0, zk1);
    if (wv * 16 < s) rt_load(xa, sx, wv * 16, wv * 16, s);
    // staging: one row per thread, 16 columns in flight (see the forward kernel)
    for (int i = tid; i < b; i += nt) {
        double v[MFM_KC];
        const double* __restrict__ src = W + B[i] + (size_t)c0 * ldw;
#pragma unroll
        for (int c = 0; c < MFM_KC; ++c) v[c] = src[(size_t)min(c, kc - 1) * ldw];
#pragma unroll
        for (int c = 0; c < MFM_KC; ++c) xb[i + c * ldx] = c < kc ? v[c] : 0.0;
    }
    for (int id = tid; id < (ldx - b) * MFM_KC; id += nt) { const int i = b + id % (ldx - b), c = id / (ldx - b); xb[i + c * ldx] = 0.0; }
    for (int i = tid; i < s; i += nt) {
        double v[MFM_KC];
        const double* __restrict__ src = W + (first + i) + (size_t)c0 * ldw;
#pragma unroll
        for (int c = 0; c < MFM_KC; ++c) v[c] = src[(size_t)min(c, kc - 1) * ldw];
#pragma unroll
        for (int c = 0; c < MFM_KC; ++c) z[i + c * ldz] = c < kc ? v[c] : 0.0;
    }
    for (int id = tid; id < (ldz - s) * MFM_KC; id += nt) { const int i = s + id % (ldz - s), c = id / (ldz - s); z[i + c * ldz] = 0.0; }
    __syncthreads();
    const int lr = lane & 15, lq = lane >> 4;
    // z = w_S - U12 x_B   (in place in LDS: every tile reads and writes only its own rows of z)
    if (G >= 2) {
        mf_v4d acc = {0.0, 0.0, 0.0, 0.0};
        const int r0 = ztile * 16;
        if (zwork) acc = rt_tile(acc, sz, r0, zk0, zk1, -1.0, xb, ldx, [s](int row, int) { return row < s; }, za);
        if (ztile < nts && zpart > 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) part[(size_t)wv * 256 + r * 64 + lane] = acc[r];
        }
        __syncthreads();
        if (ztile < nts && zpart == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                double v = z[(r0 + lq + 4 * r) + lr * ldz] + acc[r];
                for (int g = 1; g < G; ++g) v += part[(size_t)(wv + g) * 256 + r * 64 + lane];
                z[(r0 + lq + 4 * r) + lr * ldz] = v;
            }
        }
    } else {
        for (int rt = wv; rt * 16 < s; rt += nw) {
            const int r0 = rt * 16;
            if (rt != wv && b > 0) rt_load(za, sz, r0, 0, b);
            mf_v4d acc;
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r] = z[(r0 + lq + 4 * r) + lr * ldz];
            acc = rt_tile(acc, sz, r0, 0, b, -1.0, xb, ldx, [s](int row, int) { return row < s; }, za);
#pragma unroll
            for (int r = 0; r < 4; ++r) z[(r0 + lq + 4 * r) + lr * ldz] = acc[r];
        }
    }
    __syncthreads();
    // x_S = inv(U11) z : upper triangle of Ti including the diagonal
    for (int rt = wv; rt * 16 < s; rt += nw) {
        const int r0 = rt * 16;
        if (rt != wv) rt_load(xa, sx, r0, r0, s);
        mf_v4d acc = {0.0, 0.0, 0.0, 0.0};
        acc = rt_tile(acc, sx, r0, r0, s, 1.0, z, ldz, [s](int row, int k) { return row < s && k >= row; }, xa);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = r0 + lq + 4 * r;
            if (row < s && lr < kc) W[(first + row) + (size_t)(c0 + lr) * ldw] = acc[r];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Subtree sweeps (SubPlan, sparse.hpp).  The level kernels above pay one launch and one chain of dependent memory round trips per tree
// level and direction (16-18 us each at n = 20209, 14 of them per solve below the dense top).  Below level Tsub a subtree is a contiguous
// range of nodes and of pivot rows, small enough that its slice of the right-hand-side panel (16 columns) fits in LDS: ONE workgroup then
// walks the whole subtree in postorder.  Forward: y_S = inv(L11) w_S, and -L21 y_S is scattered straight into the LDS rows of the
// ancestors (supernodal form: no update vectors between the nodes of a subtree; what leaves the subtree is accumulated in the rows of the
// root's boundary and written out as the root's update vector, so the dense top / the level kernels above see exactly what they saw
// before).  Backward: reverse postorder, x_B gathered from LDS.  The factor entries are the only HBM/L2 traffic inside the loop; every
// wave prefetches the A fragments of its tile of the NEXT node (8 MFMA K-steps, enough for pivot blocks up to 32) while the current node
// computes, so the loop is bound by the MFMA chains and two (four) workgroup barriers per node, not by memory latency.
// MEASURED (MI355X, n = 20209, 99 right-hand sides, wall-clock stamps inside the kernel): 2.2 us per node (0.4 issuing the prefetch,
// 0.65 y tile, 0.9 update tile + scatter, 0.25 barriers and fragment hand-over) — a chain of two dependent 8-step f64 MFMA products
// with their LDS operands; 43 nodes per subtree at Tsub = 5 give 110 us forward / 155 us backward against 122 / 122 us for the seven
// level launches they replace, and deeper roots (shorter chains, more workgroups than CUs x occupancy) land at the same total.  The
// subtree sweeps are therefore OFF by default (dre_ctx_set_option "mf_subtree"); they are kept, tested against the level kernels and
// SuperLU, for pencils whose trees are deep and narrow.
// ---------------------------------------------------------------------------------------------
#define SUB_NW 8
#define SUB_NT (SUB_NW * 64)
#define SUB_LD 17
struct SubArgs { const int* sub; const int* lmap; int prows_cap, nn_cap, lm_cap, b_cap, s_cap; };
struct FragSrc {
    const double* base;   // element (row, k) = base[row + k * ld]
    int ld, nrow, kend, mode;      // valid: row < nrow, k < kend, and (mode 0: all | 1: k < row | 2: k >= row); everything else reads as 0
};
// Branch-free: every lane loads from a clamped (always valid) address; the predicate is applied by frag_mask when the values are CONSUMED,
// so the eight loads of a batch are in flight together and a prefetched batch is not waited for before the node that uses it.
__device__ __forceinline__ void frag_load(double (&av)[8], const FragSrc& f, int r0, int kb) {
    const int lane = threadIdx.x & 63, row = r0 + (lane & 15), lk = lane >> 4;
    const int rc = max(min(row, f.nrow - 1), 0);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int k = kb + 4 * u + lk;
        av[u] = f.base[rc + (size_t)max(min(k, f.kend - 1), 0) * f.ld];
    }
}
__device__ __forceinline__ void frag_mask(double (&av)[8], const FragSrc& f, int r0, int kb) {
    const int lane = threadIdx.x & 63, row = r0 + (lane & 15), lk = lane >> 4;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int k = kb + 4 * u + lk;
        const bool ok = row < f.nrow && k < f.kend && (f.mode == 0 || (f.mode == 1 ? k < row : k >= row));
        av[u] = ok ? av[u] : 0.0;
    }
}
__device__ __forceinline__ mf_v4d frag_mma(mf_v4d acc, const double (&av)[8], const double* __restrict__ Bs, int brow0, int kb, int kend) {
    const int lane = threadIdx.x & 63, lr = lane & 15, lk = lane >> 4;
#pragma unroll
    for (int u = 0; u < 8; ++u)
        if (kb + 4 * u < kend) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], Bs[(brow0 + kb + 4 * u + lk) * SUB_LD + lr], acc, 0, 0, 0);
    return acc;
}
struct SubLds {
    double* panel; long long* m_fo; long long* m_io; int* m_s; int* m_b; int* m_prow; int* m_lm; int* lml; double* extra;
};
__device__ __forceinline__ SubLds sub_carve(double* sm, const SubArgs& sa) {
    SubLds L;
    L.panel = sm;
    L.m_fo = reinterpret_cast<long long*>(sm + (size_t)(sa.prows_cap + 4) * SUB_LD);
    L.m_io = L.m_fo + sa.nn_cap;
    L.m_s = reinterpret_cast<int*>(L.m_io + sa.nn_cap);
    L.m_b = L.m_s + sa.nn_cap; L.m_prow = L.m_b + sa.nn_cap; L.m_lm = L.m_prow + sa.nn_cap;
    L.lml = L.m_lm + sa.nn_cap;
    L.extra = reinterpret_cast<double*>(L.lml + ((sa.lm_cap + 1) & ~1));
    return L;
}
static size_t sub_lds_base(const SubArgs& sa) {
    return (size_t)(sa.prows_cap + 4) * SUB_LD * 8 + (size_t)sa.nn_cap * (2 * 8 + 4 * 4) + (size_t)((sa.lm_cap + 1) & ~1) * 4;
}
// one prefetched batch (8 K-steps from kb) of a tile
__device__ __forceinline__ mf_v4d tile_pf(mf_v4d acc, const FragSrc& f, int r0, int kb, const double* __restrict__ Bs, int brow0, const double (&pf)[8]) {
    double av[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) av[u] = pf[u];
    frag_mask(av, f, r0, kb);
    return frag_mma(acc, av, Bs, brow0, kb, f.kend);      // (the branch-free variant with eight unconditional MFMAs measured 8-10 % slower)
}
// the rest of a tile's K range with loads issued on the spot (fronts beyond the prefetch depth; never instantiated in the SMALL kernels)
__device__ __forceinline__ mf_v4d tile_rest(mf_v4d acc, const FragSrc& f, int r0, int kb, const double* __restrict__ Bs, int brow0) {
    for (; kb < f.kend; kb += 32) {
        double av[8];
        frag_load(av, f, r0, kb);
        frag_mask(av, f, r0, kb);
        acc = frag_mma(acc, av, Bs, brow0, kb, f.kend);
    }
    return acc;
}

// SMALL: every node of every subtree has s <= 32 and b <= 256, so that all factor fragments of a node are covered by the prefetch (one
// batch for the y tile, two update tiles per wave) and the node loop contains no other global load: the only vector-memory wait of an
// iteration is the one at its top, for fragments requested one node earlier.
template <bool SMALL>
__global__ __launch_bounds__(SUB_NT) void k_mf_sub_forward(MfArgs a, SubArgs sa, const double* __restrict__ fronts, const double* __restrict__ inv,
                                                           double* __restrict__ W, int ldw, int nrhs, double* __restrict__ upd, int64_t ldu,
                                                           const AdiState* st, long long* probe) {
    if (st && st->done) return;
    int pslot = 0;
#define SUB_STAMP() do { if (probe && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0 && pslot < 60) probe[pslot++] = (long long)wall_clock64(); } while (0)
    SUB_STAMP();
    extern __shared__ double sm[];
    const SubLds L = sub_carve(sm, sa);
    double* yb = L.extra;                                         // (round16(s_cap) + 4) x SUB_LD: y of the current node
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lr = lane & 15, lq = lane >> 4;
    const int n0 = sa.sub[4 * blockIdx.x], nn = sa.sub[4 * blockIdx.x + 1], row0 = sa.sub[4 * blockIdx.x + 2], nrows = sa.sub[4 * blockIdx.x + 3];
    const int root = n0 + nn - 1, lm0 = a.bptr[n0], broot = a.bptr[root + 1] - a.bptr[root], prows = nrows + broot;
    const int c0 = blockIdx.y * MFM_KC, kc = min(MFM_KC, nrhs - c0);
    for (int i = tid; i < nn; i += SUB_NT) {
        const int t = n0 + i;
        L.m_s[i] = a.size[t]; L.m_b[i] = a.bptr[t + 1] - a.bptr[t]; L.m_prow[i] = a.first[t] - row0; L.m_lm[i] = a.bptr[t] - lm0;
        L.m_fo[i] = a.front_off[t]; L.m_io[i] = a.inv_off[t];
    }
    for (int i = tid; i < a.bptr[root + 1] - lm0; i += SUB_NT) L.lml[i] = sa.lmap[lm0 + i];
    for (int id0 = tid; id0 < (prows + 4) * MFM_KC; id0 += 8 * SUB_NT) {       // eight loads in flight per thread
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int id = id0 + u * SUB_NT, i = id % (prows + 4), c = id / (prows + 4);
            const bool ok = i < nrows && c < kc;
            v[u] = W[ok ? (size_t)(row0 + i) + (size_t)(c0 + c) * ldw : (size_t)row0];
            v[u] = ok ? v[u] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int id = id0 + u * SUB_NT, i = id % (prows + 4), c = id / (prows + 4);
            if (id < (prows + 4) * MFM_KC) L.panel[i * SUB_LD + c] = v[u];
        }
    }
    __syncthreads();
    SUB_STAMP();
    const int r0 = wave * 16;
    double nY[8], nU[8], nU2[8];
    auto prefetch = [&](int i) {
        const int s = L.m_s[i], b = L.m_b[i];
        if (r0 < s) frag_load(nY, FragSrc{inv + L.m_io[i], s, s, min(s, r0 + 16), 1}, r0, 0);
        if (r0 < b) frag_load(nU, FragSrc{fronts + L.m_fo[i] + s, s + b, b, s, 0}, r0, 0);
        if (SMALL && r0 + 16 * SUB_NW < b) frag_load(nU2, FragSrc{fronts + L.m_fo[i] + s, s + b, b, s, 0}, r0 + 16 * SUB_NW, 0);
    };
    prefetch(0);
    for (int i = 0; i < nn; ++i) {
        double cY[8], cU[8], cU2[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {     // the wait for this node's fragments sits HERE, before the next prefetch is issued (vmcnt counts in order)
            cY[u] = nY[u]; cU[u] = nU[u]; cU2[u] = nU2[u];
            asm volatile("" : "+v"(cY[u]), "+v"(cU[u]), "+v"(cU2[u]));
        }
        SUB_STAMP();
        if (i + 1 < nn) prefetch(i + 1);
        SUB_STAMP();
        const int s = L.m_s[i], b = L.m_b[i], f = s + b, prow = L.m_prow[i], lm = L.m_lm[i];
        const double* F = fronts + L.m_fo[i];
        const double* Ti = inv + L.m_io[i];
        // y_S = inv(L11) w_S  (unit diagonal: the accumulator starts at w; strictly lower part of Ti)
        // (y goes to its own buffer: the tiles of a node read each other's w rows, and one barrier per phase is saved)
        if (r0 < s) {
            mf_v4d yacc;
#pragma unroll
            for (int r = 0; r < 4; ++r) { const int row = r0 + lq + 4 * r; yacc[r] = row < s ? L.panel[(prow + row) * SUB_LD + lr] : 0.0; }
            const FragSrc fy{Ti, s, s, min(s, r0 + 16), 1};
            yacc = tile_pf(yacc, fy, r0, 0, L.panel, prow, cY);
            if (!SMALL) yacc = tile_rest(yacc, fy, r0, 32, L.panel, prow);
#pragma unroll
            for (int r = 0; r < 4; ++r) yb[(r0 + lq + 4 * r) * SUB_LD + lr] = yacc[r];       // rows >= s of the last tile: never read with a nonzero A
        }
        SUB_STAMP();
        __syncthreads();
        SUB_STAMP();
        for (int id = tid; id < s * MFM_KC; id += SUB_NT) { const int row = id >> 4, c = id & 15; L.panel[(prow + row) * SUB_LD + c] = yb[row * SUB_LD + c]; }
        // rows of the ancestors -= L21 y_S   (the tile holds +L21 y_S)
        const FragSrc fu{F + s, f, b, s, 0};
        auto scatter = [&](int rt, const mf_v4d& acc) {
#pragma unroll
            for (int r = 0; r < 4; ++r) { const int row = rt * 16 + lq + 4 * r; if (row < b) L.panel[L.lml[lm + row] * SUB_LD + lr] -= acc[r]; }
        };
        if (r0 < b) {
            mf_v4d acc = {0.0, 0.0, 0.0, 0.0};
            acc = tile_pf(acc, fu, r0, 0, yb, 0, cU);
            if (!SMALL) acc = tile_rest(acc, fu, r0, 32, yb, 0);
            scatter(wave, acc);
        }
        if (SMALL) {
            if (r0 + 16 * SUB_NW < b) {
                mf_v4d acc = {0.0, 0.0, 0.0, 0.0};
                acc = tile_pf(acc, fu, r0 + 16 * SUB_NW, 0, yb, 0, cU2);
                scatter(wave + SUB_NW, acc);
            }
        } else {
            for (int rt = wave + SUB_NW; rt * 16 < b; rt += SUB_NW) {
                mf_v4d acc = {0.0, 0.0, 0.0, 0.0};
                acc = tile_rest(acc, fu, rt * 16, 0, yb, 0);
                scatter(rt, acc);
            }
        }
        SUB_STAMP();
        __syncthreads();
        SUB_STAMP();
    }
    for (int id = tid; id < nrows * kc; id += SUB_NT) {
        const int i = id % nrows, c = id / nrows;
        W[(size_t)(row0 + i) + (size_t)(c0 + c) * ldw] = L.panel[i * SUB_LD + c];
    }
    double* ut = upd + a.upd_off[root];
    for (int id = tid; id < broot * kc; id += SUB_NT) {
        const int j = id % broot, c = id / broot;
        ut[j + (size_t)(c0 + c) * ldu] = L.panel[(nrows + j) * SUB_LD + c];
    }
    SUB_STAMP();
    if (probe && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) probe[63] = pslot;
#undef SUB_STAMP
}

template <bool SMALL>
__global__ __launch_bounds__(SUB_NT) void k_mf_sub_backward(MfArgs a, SubArgs sa, const double* __restrict__ fronts, const double* __restrict__ inv,
                                                            double* __restrict__ W, int ldw, int nrhs, const AdiState* st) {
    if (st && st->done) return;
    extern __shared__ double sm[];
    const SubLds L = sub_carve(sm, sa);
    double* xb = L.extra;                                         // (b_cap + 4) x SUB_LD: the boundary unknowns of the current node
    double* zb = xb + (size_t)(sa.b_cap + 4) * SUB_LD;            // (s_cap + 4) x SUB_LD
    double* part = zb + (size_t)(sa.s_cap + 4) * SUB_LD;          // SUB_NW x 256: K-split partial tiles
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lr = lane & 15, lq = lane >> 4;
    const int n0 = sa.sub[4 * blockIdx.x], nn = sa.sub[4 * blockIdx.x + 1], row0 = sa.sub[4 * blockIdx.x + 2], nrows = sa.sub[4 * blockIdx.x + 3];
    const int root = n0 + nn - 1, lm0 = a.bptr[n0], broot = a.bptr[root + 1] - a.bptr[root], prows = nrows + broot;
    const int c0 = blockIdx.y * MFM_KC, kc = min(MFM_KC, nrhs - c0);
    for (int i = tid; i < nn; i += SUB_NT) {
        const int t = n0 + i;
        L.m_s[i] = a.size[t]; L.m_b[i] = a.bptr[t + 1] - a.bptr[t]; L.m_prow[i] = a.first[t] - row0; L.m_lm[i] = a.bptr[t] - lm0;
        L.m_fo[i] = a.front_off[t]; L.m_io[i] = a.inv_off[t];
    }
    for (int i = tid; i < a.bptr[root + 1] - lm0; i += SUB_NT) L.lml[i] = sa.lmap[lm0 + i];
    const int* Broot = a.bidx + a.bptr[root];
    for (int id0 = tid; id0 < (prows + 4) * MFM_KC; id0 += 8 * SUB_NT) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int id = id0 + u * SUB_NT, i = id % (prows + 4), c = id / (prows + 4);
            const bool ok = i < prows && c < kc;
            const int grow = i < nrows ? row0 + i : Broot[min(max(i - nrows, 0), max(broot - 1, 0))];
            v[u] = W[ok ? (size_t)grow + (size_t)(c0 + c) * ldw : (size_t)row0];
            v[u] = ok ? v[u] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int id = id0 + u * SUB_NT, i = id % (prows + 4), c = id / (prows + 4);
            if (id < (prows + 4) * MFM_KC) L.panel[i * SUB_LD + c] = v[u];
        }
    }
    __syncthreads();
    double nZ[8], nZ2[8], nX[8];
    auto prefetch = [&](int i) {
        const int s = L.m_s[i], b = L.m_b[i], f = s + b;
        const int ny = (s + 15) >> 4, nch = max(1, SUB_NW / ny), ks = (b + 3) >> 2, cks = (ks + nch - 1) / nch;
        const int tile = wave % ny, ch = wave / ny;
        if (wave < ny * nch && b > 0) {
            const int kb0 = ch * cks * 4, ke = min(b, (ch + 1) * cks * 4);
            const FragSrc fz{fronts + L.m_fo[i] + (size_t)s * f, f, s, ke, 0};
            if (kb0 < ke) frag_load(nZ, fz, tile * 16, kb0);
            if (SMALL && kb0 + 32 < ke) frag_load(nZ2, fz, tile * 16, kb0 + 32);
        }
        if (wave < ny) frag_load(nX, FragSrc{inv + L.m_io[i], s, s, s, 2}, wave * 16, wave * 16);
    };
    prefetch(nn - 1);
    for (int i = nn - 1; i >= 0; --i) {
        double cZ[8], cZ2[8], cX[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            cZ[u] = nZ[u]; cZ2[u] = nZ2[u]; cX[u] = nX[u];
            asm volatile("" : "+v"(cZ[u]), "+v"(cZ2[u]), "+v"(cX[u]));
        }
        if (i > 0) prefetch(i - 1);
        const int s = L.m_s[i], b = L.m_b[i], f = s + b, prow = L.m_prow[i], lm = L.m_lm[i];
        const double* F = fronts + L.m_fo[i];
        const double* Ti = inv + L.m_io[i];
        const int ny = (s + 15) >> 4, nch = max(1, SUB_NW / ny), ks = (b + 3) >> 2, cks = (ks + nch - 1) / nch;
        const int bp = (b + 3) & ~3, sp = (s + 3) & ~3;
        for (int id = tid; id < bp * MFM_KC; id += SUB_NT) {
            const int k = id >> 4, c = id & 15;
            xb[k * SUB_LD + c] = k < b ? L.panel[L.lml[lm + k] * SUB_LD + c] : 0.0;
        }
        __syncthreads();
        // partial tiles of  U12 x_B,  K split over the waves (subtracted in the reduction below)
        if (wave < ny * nch && b > 0) {
            const int tile = wave % ny, ch = wave / ny;
            const int kb0 = ch * cks * 4, ke = min(b, (ch + 1) * cks * 4);
            mf_v4d acc = {0.0, 0.0, 0.0, 0.0};
            if (kb0 < ke) {
                const FragSrc fz{F + (size_t)s * f, f, s, ke, 0};
                acc = tile_pf(acc, fz, tile * 16, kb0, xb, 0, cZ);
                if (SMALL) { if (kb0 + 32 < ke) acc = tile_pf(acc, fz, tile * 16, kb0 + 32, xb, 0, cZ2); }
                else acc = tile_rest(acc, fz, tile * 16, kb0 + 32, xb, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) part[wave * 256 + (lq + 4 * r) * 16 + lr] = acc[r];
        }
        __syncthreads();
        for (int id = tid; id < sp * MFM_KC; id += SUB_NT) {
            const int row = id >> 4, c = id & 15;
            double v = 0.0;
            if (row < s) {
                v = L.panel[(prow + row) * SUB_LD + c];
                if (b > 0) { const int tile = row >> 4; for (int ch = 0; ch < nch; ++ch) v -= part[(tile + ch * ny) * 256 + (row & 15) * 16 + c]; }
            }
            zb[row * SUB_LD + c] = v;
        }
        __syncthreads();
        // x_S = inv(U11) z : upper triangle of Ti including the diagonal
        if (wave < ny) {
            const int r0 = wave * 16;
            mf_v4d acc = {0.0, 0.0, 0.0, 0.0};
            const FragSrc fx{Ti, s, s, s, 2};
            acc = tile_pf(acc, fx, r0, r0, zb, 0, cX);
            if (!SMALL) acc = tile_rest(acc, fx, r0, r0 + 32, zb, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) { const int row = r0 + lq + 4 * r; if (row < s) L.panel[(prow + row) * SUB_LD + lr] = acc[r]; }
        }
        __syncthreads();
    }
    for (int id = tid; id < nrows * kc; id += SUB_NT) {
        const int i = id % nrows, c = id / nrows;
        W[(size_t)(row0 + i) + (size_t)(c0 + c) * ldw] = L.panel[i * SUB_LD + c];
    }
}

// host: choose the root level and build the per-subtree tables
static void sub_plan_build(Ctx* ctx, const Pencil& P, int Tmin) {
    SubPlan& sp = P.sub;
    sp.built = true; sp.Tsub = -1;
    if (!ctx->mf_subtree) return;
    const Symbolic& S = P.sym;
    const size_t lds_limit = 156 * 1024;
    std::vector<int> nsubnodes((size_t)S.nnodes, 1);            // nodes of the subtree rooted at t (postorder: children first)
    for (int t = 0; t < S.nnodes; ++t) if (S.parent[t] >= 0) nsubnodes[(size_t)S.parent[t]] += nsubnodes[(size_t)t];
    // root level: the chain of a workgroup costs ~2.2 us per node (measured: MFMA / LDS latency of two dependent tile products) plus ~6 us of
    // set-up, a level launch above the subtrees ~17.5 us; deeper roots mean shorter chains and more level launches
    int forced = -1;
    int bestL = -1; double bestcost = 1e300;
    for (int Lv = std::max(Tmin, 1); Lv < S.nlevels; ++Lv) {
        int nnmax = 0;
        for (int q = S.lvl_ptr[Lv]; q < S.lvl_ptr[Lv + 1]; ++q) nnmax = std::max(nnmax, nsubnodes[(size_t)S.lvl_nodes[q]]);
        const double cost = 6.0 + 2.2 * nnmax + 17.5 * (Lv - std::max(Tmin, 0));
        if (cost < bestcost) { bestcost = cost; bestL = Lv; }
    }
    if (forced >= 0) bestL = std::max(forced, std::max(Tmin, 1));
    for (int Lv = std::max(bestL, 1); Lv < S.nlevels; ++Lv) {
        SubArgs sa{nullptr, nullptr, 0, 0, 0, 0, 0};
        bool ok = true;
        std::vector<int> tab;
        for (int q = S.lvl_ptr[Lv]; q < S.lvl_ptr[Lv + 1] && ok; ++q) {
            const int r = S.lvl_nodes[q], nn = nsubnodes[(size_t)r], n0 = r - nn + 1;
            const int rowb = S.first[n0], nrows = S.first[r] + S.size[r] - rowb, broot = S.bptr[r + 1] - S.bptr[r];
            sa.prows_cap = std::max(sa.prows_cap, nrows + broot); sa.nn_cap = std::max(sa.nn_cap, nn);
            sa.lm_cap = std::max(sa.lm_cap, S.bptr[r + 1] - S.bptr[n0]);
            for (int t = n0; t <= r; ++t) {
                sa.b_cap = std::max(sa.b_cap, S.bptr[t + 1] - S.bptr[t]); sa.s_cap = std::max(sa.s_cap, S.size[t]);
                if (S.level[t] < Lv) ok = false;                 // not a subtree in the postorder (cannot happen for a tree; guard)
            }
            tab.push_back(n0); tab.push_back(nn); tab.push_back(rowb); tab.push_back(nrows);
        }
        if (!ok || tab.empty() || sa.s_cap > 16 * SUB_NW) continue;
        const size_t base = sub_lds_base(sa), fwd = base + (size_t)(((sa.s_cap + 15) & ~15) + 4) * SUB_LD * 8;
        const size_t bwd = base + ((size_t)(sa.b_cap + 4) * SUB_LD + (size_t)(sa.s_cap + 4) * SUB_LD + (size_t)SUB_NW * 256) * 8;
        if (bwd > lds_limit) continue;
        // rows of the subtree's LDS panel for every boundary entry
        std::vector<int> lmap((size_t)std::max<int>(S.bptr[S.nnodes], 1), 0);
        for (size_t z = 0; z < tab.size(); z += 4) {
            const int n0 = tab[z], nn = tab[z + 1], rowb = tab[z + 2], nrows = tab[z + 3], r = n0 + nn - 1;
            const int* Br = S.bidx.data() + S.bptr[r];
            const int broot = S.bptr[r + 1] - S.bptr[r];
            for (int t = n0; t <= r; ++t)
                for (int e = S.bptr[t]; e < S.bptr[t + 1]; ++e) {
                    const int j = S.bidx[e];
                    if (j < rowb + nrows) { lmap[(size_t)e] = j - rowb; continue; }
                    const int* it = std::lower_bound(Br, Br + broot, j);
                    if (it == Br + broot || *it != j) return;     // boundary entry outside the root's boundary: keep the level kernels
                    lmap[(size_t)e] = nrows + (int)(it - Br);
                }
        }
        sp.sub = DevArr<int>(ctx, tab.size()); sp.sub.upload(ctx, tab);
        sp.lmap = DevArr<int>(ctx, lmap.size()); sp.lmap.upload(ctx, lmap);
        sp.Tsub = Lv; sp.nsub = (int)(tab.size() / 4);
        sp.prows_max = sa.prows_cap; sp.nn_max = sa.nn_cap; sp.lm_max = sa.lm_cap; sp.b_max = sa.b_cap; sp.s_max = sa.s_cap;
        sp.lds_fwd = fwd; sp.lds_bwd = bwd;
        if (env_trace("subtree"))
            std::fprintf(stderr, "[subtree sweeps] n=%d levels=%d Tsub=%d subtrees=%d panel rows<=%d nodes<=%d b<=%d s<=%d LDS fwd %zu bwd %zu\n", P.n, S.nlevels, Lv,
                         sp.nsub, sa.prows_cap, sa.nn_cap, sa.b_cap, sa.s_cap, fwd, bwd);
        return;
    }
}
static void mf_sub_sweep(Ctx* ctx, const Pencil& P, const Factor<double>& Fc, double* W, int ldw, int nrhs, double* upd, int64_t ldu, const AdiState* st,
                         bool forward) {
    const SubPlan& sp = P.sub;
    MfArgs a = mf_args(P);
    SubArgs sa{sp.sub.p, sp.lmap.p, sp.prows_max, sp.nn_max, sp.lm_max, sp.b_max, sp.s_max};
    const int ncb = ceil_div(nrhs, MFM_KC);
    const bool small = sp.s_max <= 32 && sp.b_max <= 32 * SUB_NW;        // every fragment of every node fits the prefetch depth
    static const bool probe_on = env_trace("subprobe");   // debug: wall-clock stamps (10 ns ticks) of workgroup (0, 0) at the phase boundaries
    int& probe_count = ctx->trace.subprobe_count;
    DevArr<long long> probe;
    long long* pp = nullptr;
    if (probe_on && forward && nrhs >= 64 && probe_count < 3) { probe = DevArr<long long>(ctx, 64); DRE_HIP(hipMemsetAsync(probe.p, 0, 64 * 8, ctx->stream)); pp = probe.p; }
    if (small) {
        lds_attr(ctx, (const void*)k_mf_sub_forward<true>, 160 * 1024); lds_attr(ctx, (const void*)k_mf_sub_backward<true>, 160 * 1024);
        if (forward) hipLaunchKernelGGL((k_mf_sub_forward<true>), dim3(sp.nsub, ncb), dim3(SUB_NT), sp.lds_fwd, ctx->stream, a, sa, Fc.fronts.p, Fc.inv.p, W, ldw, nrhs, upd, ldu, st, pp);
        else hipLaunchKernelGGL((k_mf_sub_backward<true>), dim3(sp.nsub, ncb), dim3(SUB_NT), sp.lds_bwd, ctx->stream, a, sa, Fc.fronts.p, Fc.inv.p, W, ldw, nrhs, st);
    } else {
        lds_attr(ctx, (const void*)k_mf_sub_forward<false>, 160 * 1024); lds_attr(ctx, (const void*)k_mf_sub_backward<false>, 160 * 1024);
        if (forward) hipLaunchKernelGGL((k_mf_sub_forward<false>), dim3(sp.nsub, ncb), dim3(SUB_NT), sp.lds_fwd, ctx->stream, a, sa, Fc.fronts.p, Fc.inv.p, W, ldw, nrhs, upd, ldu, st, pp);
        else hipLaunchKernelGGL((k_mf_sub_backward<false>), dim3(sp.nsub, ncb), dim3(SUB_NT), sp.lds_bwd, ctx->stream, a, sa, Fc.fronts.p, Fc.inv.p, W, ldw, nrhs, st);
    }
    if (pp) {
        long long h[64];
        DRE_HIP(hipMemcpyAsync(h, pp, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
        DRE_HIP(hipStreamSynchronize(ctx->stream));
        ++probe_count;
        std::fprintf(stderr, "[sub probe] nrhs=%d stamps=%lld (us since start):", nrhs, h[63]);
        for (int i = 1; i < (int)h[63] && i < 60; ++i) std::fprintf(stderr, " %.2f", (double)(h[i] - h[0]) * 0.01);
        std::fprintf(stderr, "\n");
    }
}

// ---- top levels as one dense operator (TopPlan, sparse.hpp) --------------------------------------------------------
static void top_plan_build(Ctx* ctx, const Pencil& P, int max_rows) {
    TopPlan& tp = P.top;
    tp.built = true; tp.T = 0; tp.ntop = 0;
    const Symbolic& S = P.sym;
    int T = 0, rows = 0;
    while (T < S.nlevels - 2) {
        int add = 0;
        for (int q = S.lvl_ptr[T]; q < S.lvl_ptr[T + 1]; ++q) add += S.size[S.lvl_nodes[q]];
        if (rows + add > max_rows) break;
        rows += add; ++T;
    }
    if (T < 2 || rows < 32) return;
    std::vector<int> topidx; topidx.reserve(rows);
    std::vector<int> pos((size_t)S.n, -1);
    for (int l = 0; l < T; ++l)
        for (int q = S.lvl_ptr[l]; q < S.lvl_ptr[l + 1]; ++q) {
            const int t = S.lvl_nodes[q];
            for (int i = 0; i < S.size[t]; ++i) { pos[(size_t)S.first[t] + i] = (int)topidx.size(); topidx.push_back(S.first[t] + i); }
        }
    // update rows of the level-T nodes (children of the top) by destination
    std::vector<std::vector<int64_t>> src((size_t)rows);
    for (int q = S.lvl_ptr[T]; q < S.lvl_ptr[T + 1]; ++q) {
        const int c = S.lvl_nodes[q];
        for (int i = S.bptr[c]; i < S.bptr[c + 1]; ++i) {
            const int pp = pos[(size_t)S.bidx[i]];
            if (pp < 0) return;                       // border variable outside the top levels: unexpected tree shape, keep the sweeps
            src[(size_t)pp].push_back(S.upd_off[c] + (i - S.bptr[c]));
        }
    }
    std::vector<int> gptr((size_t)rows + 1, 0);
    std::vector<int64_t> gsrc;
    for (int pp = 0; pp < rows; ++pp) { gsrc.insert(gsrc.end(), src[pp].begin(), src[pp].end()); gptr[(size_t)pp + 1] = (int)gsrc.size(); }
    tp.topidx = DevArr<int>(ctx, topidx.size()); tp.topidx.upload(ctx, topidx);
    tp.gptr = DevArr<int>(ctx, gptr.size()); tp.gptr.upload(ctx, gptr);
    tp.gsrc = DevArr<int64_t>(ctx, std::max<size_t>(gsrc.size(), 1)); tp.gsrc.upload(ctx, gsrc);
    tp.T = T; tp.ntop = rows;
}
__global__ void k_copy_cols(int n, const double* __restrict__ src, int lds_, double* __restrict__ dst, int ldd, const AdiState* st) {
    if (st && st->done) return;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i + (size_t)blockIdx.y * ldd] = src[i + (size_t)blockIdx.y * lds_];
}
// g[p, c] = W[topidx[p], c] + sum of the update rows that the level-T nodes send to top variable p
__global__ void k_top_gather(int ntop, int nrhs, const int* __restrict__ topidx, const int* __restrict__ gptr, const int64_t* __restrict__ gsrc,
                             const double* __restrict__ W, int ldw, const double* __restrict__ upd, int64_t ldu, double* __restrict__ g, int ldg,
                             const AdiState* st, const double* __restrict__ Win, int ldwin, int nin, long wz, long uz, long gz) {
    if (st && st->done) return;
    W += (size_t)blockIdx.y * wz; upd += (size_t)blockIdx.y * uz; g += (size_t)blockIdx.y * gz;      // (batched solves: blockIdx.y = factor)
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)ntop * nrhs) return;
    const int p = idx % ntop; const size_t c = idx / ntop;
    double v = (int)c < nin ? Win[topidx[p] + c * ldwin] : W[topidx[p] + c * ldw];
    for (int j = gptr[p]; j < gptr[p + 1]; ++j) v += upd[gsrc[j] + c * (size_t)ldu];
    g[p + c * ldg] = v;
}
// (A fused top product — x = Sinv g scattered into the panel by one K-split tile kernel, 4 / 8 / 16 waves — was built in round 3 and measured at
// the same wall-clock as split-K GEMM + slab reduction (22-24 us against 17.5 + 3.8 at n = 5177); removed in round 4.)
__global__ void k_top_unit_rows(int nrhs, int c0, const int* __restrict__ topidx, double* __restrict__ W, int ldw) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < nrhs) W[topidx[c0 + j] + (size_t)j * ldw] = 1.0;
}
__global__ void k_top_collect(int ntop, int nrhs, int c0, const int* __restrict__ topidx, const double* __restrict__ W, int ldw, double* __restrict__ Ti, int ldt) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)ntop * nrhs) return;
    const int p = idx % ntop; const size_t c = idx / ntop;
    Ti[p + (c0 + c) * ldt] = W[topidx[p] + c * ldw];
}

struct MfIn;
static void mf_solve_mfma(Ctx* ctx, const Pencil& P, const Factor<double>& Fc, double* W, int ldw, int nrhs, const AdiState* st, const MfIn* inp = nullptr);
// inv(S) = (M^-1)[top, top]: unit right-hand sides on the top variables, sweeps over the top levels only (everything below stays zero)
static void mf_build_topinv(Ctx* ctx, const Pencil& P, const Factor<double>& Fc);

// Backward levels with at most this many workgroups (fronts x column blocks) run the latency variant of the sweep kernel (env
// DRE_MF_LAT_BWD overrides; 0 = throughput variant everywhere).  Sweep at n = 20209 / 5177 (tools/mf_latency_sweep.sh): 700, 1500 and 3000 are
// within 2 % of each other, 0 costs 15 % / 9 % of the solve time.  A latency variant of the FORWARD kernel (same loads, early prefetch)
// was built and lost at every threshold (its K ranges are short: s <= 64), so the forward sweep has the one kernel.
static long mf_latency_max_wg(bool) {
    static const long b = 1500;
    return b;
}
struct MfIn { const double* p = nullptr; int ld = 0, n = 0; };       // leading right-hand-side columns that live outside the work panel
static MfZ mf_single(const Factor<double>& Fc) {
    MfZ zb; std::memset(&zb, 0, sizeof(zb));
    zb.fronts[0] = Fc.fronts.p; zb.inv[0] = Fc.inv.p;
    return zb;
}
static void mf_sweep_levels(Ctx* ctx, const Pencil& P, const MfZ& zb, int nz, double* W, int ldw, int nrhs, double* upd, int64_t ldu,
                            const AdiState* st, bool forward, int l_from, int l_to, MfIn in = MfIn()) {
    // forward: levels l_from down to l_to (l_from >= l_to);  backward: levels l_from up to l_to
    const Symbolic& S = P.sym;
    MfArgs a = mf_args(P);
    const int ncb = ceil_div(nrhs, MFM_KC);
    const int swz = ctx->mf_swizzle != 0 && ncb >= 2 ? 1 : 0;          // (one column block: nothing is shared)
    lds_attr(ctx, (const void*)k_mf_forward_mfma, 150 * 1024); lds_attr(ctx, (const void*)k_mf_backward_mfma, 150 * 1024);
    lds_attr(ctx, (const void*)k_mf_backward_tp, 150 * 1024);
    if (forward) {
        for (int l = l_from; l >= l_to; --l) {
            const int nb = S.lvl_ptr[l + 1] - S.lvl_ptr[l];
            const int fm = P.lvl_maxfront[l], sm_ = P.lvl_maxsep[l], spm = (sm_ + 15) & ~15;
            const size_t shm = ((size_t)((std::max(fm + 16, spm) + 4) | 1) + (size_t)((spm + 4) | 1)) * MFM_KC * sizeof(double);
            // wide levels (more workgroups than the chip holds at once) are throughput bound: four waves per front keep more fronts resident
            // (n = 20209: 32.9 -> 31.5 ms of sweeps per 4 steps against 512 threads, 34.2 ms with 1024)
            static const int tp_threads = 256;
            int nthreads = fm > 128 ? 1024 : (fm > 48 ? 512 : 256);
            if (tp_threads > 0 && (long)nb * ncb * nz > 2000) nthreads = tp_threads;
            hipLaunchKernelGGL(k_mf_forward_mfma, dim3(nb, ncb, nz), dim3(nthreads), shm, ctx->stream, a, S.lvl_ptr[l], zb, W, ldw, nrhs, upd, ldu, st, in.p, in.ld, in.n, swz);
        }
    } else {
        for (int l = l_from; l <= l_to; ++l) {
            const int nb = S.lvl_ptr[l + 1] - S.lvl_ptr[l];
            const int fm = P.lvl_maxfront[l], sm_ = P.lvl_maxsep[l], spm = (sm_ + 15) & ~15;
            if ((long)nb * ncb * nz > mf_latency_max_wg(false)) {
                const size_t shm = ((size_t)((fm + 36) | 1) + (size_t)((spm + 36) | 1)) * MFM_KC * sizeof(double);
                const int nthreads = sm_ > 64 ? 512 : 256;
                hipLaunchKernelGGL(k_mf_backward_tp, dim3(nb, ncb, nz), dim3(nthreads), shm, ctx->stream, a, S.lvl_ptr[l], zb, W, ldw, nrhs, st, swz);
                continue;
            }
            // waves: one per 16-row tile of the separator times up to four shares of the boundary (K-split of z = w_S - U12 x_B)
            const int nts = spm / 16, bmax = std::max(fm - 1, 0);
            const int nwv = std::min(16, std::max(sm_ > 64 ? 8 : 4, nts * std::min(4, std::max(1, bmax / 64))));
            const int nthreads = 64 * nwv;
            const size_t base = ((size_t)((fm + 36) | 1) + (size_t)((spm + 36) | 1)) * MFM_KC * sizeof(double), parts = (size_t)nwv * 256 * sizeof(double);
            const int split = base + parts <= (size_t)150 * 1024;
            const size_t shm = base + (split ? parts : 0);
            hipLaunchKernelGGL(k_mf_backward_mfma, dim3(nb, ncb, nz), dim3(nthreads), shm, ctx->stream, a, S.lvl_ptr[l], zb, W, ldw, nrhs, st, split, swz);
        }
    }
}

static void mf_build_topinv(Ctx* ctx, const Pencil& P, const Factor<double>& Fc) {
    const TopPlan& tp = P.top;
    const Symbolic& S = P.sym;
    const int ntop = tp.ntop, T = tp.T, n = P.n;
    TimedScope ts(ctx, "mf_top_inverse", 8.0 * (double)ntop * ntop, 0.0);
    Mat Ti(ctx, ntop, ntop);
    const int64_t ldu = std::max<int64_t>(S.upd_rows, 1);
    const int chunk = 512;
    for (int c0 = 0; c0 < ntop; c0 += chunk) {
        const int ch = std::min(chunk, ntop - c0);
        Mat Wk(ctx, n, ch);
        DevArr<double> upd(ctx, (size_t)ldu * ch);
        DRE_HIP(hipMemsetAsync(Wk.p, 0, (size_t)n * ch * sizeof(double), ctx->stream));
        DRE_HIP(hipMemsetAsync(upd.p, 0, (size_t)ldu * ch * sizeof(double), ctx->stream));
        hipLaunchKernelGGL(k_top_unit_rows, dim3(ceil_div(ch, 256)), dim3(256), 0, ctx->stream, ch, c0, (const int*)tp.topidx.p, Wk.p, Wk.ld);
        mf_sweep_levels(ctx, P, mf_single(Fc), 1, Wk.p, Wk.ld, ch, upd.p, ldu, nullptr, true, T - 1, 0);
        mf_sweep_levels(ctx, P, mf_single(Fc), 1, Wk.p, Wk.ld, ch, upd.p, ldu, nullptr, false, 0, T - 1);
        const size_t tot = (size_t)ntop * ch;
        hipLaunchKernelGGL(k_top_collect, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, ctx->stream, ntop, ch, c0, (const int*)tp.topidx.p,
                           (const double*)Wk.p, Wk.ld, Ti.p, Ti.ld);
    }
    DRE_HIP(hipGetLastError());
    Fc.topinv = Ti;
}

static void mf_solve_mfma(Ctx* ctx, const Pencil& P, const Factor<double>& Fc, double* W, int ldw, int nrhs, const AdiState* st, const MfIn* inp) {
    const Symbolic& S = P.sym;
    MfIn in = inp ? *inp : MfIn();
    const int64_t ldu = std::max<int64_t>(S.upd_rows, 1);
    if (Fc.allow_topinv && ctx->top_inverse_max_rows > 0 && nrhs >= 8 && Fc.topinv.empty() && ++Fc.uses == 3) {
        if (!P.top.built) top_plan_build(ctx, P, ctx->top_inverse_max_rows);
        if (P.top.T >= 2) mf_build_topinv(ctx, P, Fc);
    }
    DevArr<double> upd(ctx, (size_t)ldu * nrhs);
    const double bytes = 2.0 * 8.0 * (double)S.factor_nnz + 4.0 * 8.0 * (double)P.n * nrhs;
    const double flops = 2.0 * 2.0 * (double)S.factor_nnz * nrhs;
    // subtree sweeps below level Tsub (>= the number of dense top levels, so that both plans agree on who owns a level)
    if (!P.sub.built) {
        if (!P.top.built && ctx->top_inverse_max_rows > 0) top_plan_build(ctx, P, ctx->top_inverse_max_rows);
        sub_plan_build(ctx, P, P.top.T);
    }
    const int Tsub = P.sub.Tsub;            // -1: level kernels everywhere
    if (in.n > 0 && Tsub >= 0) {
        // the subtree kernels read their panel in place: bring the outside columns in first
        hipLaunchKernelGGL(k_copy_cols, dim3(ceil_div(P.n, 256), in.n), dim3(256), 0, ctx->stream, P.n, in.p, in.ld, W, ldw, st);
        in = MfIn();
    }
    auto forward_to = [&](int l_to) {        // levels nlevels-1 .. l_to
        if (Tsub >= 0 && Tsub >= l_to) {
            mf_sub_sweep(ctx, P, Fc, W, ldw, nrhs, upd.p, ldu, st, true);
            if (Tsub - 1 >= l_to) mf_sweep_levels(ctx, P, mf_single(Fc), 1, W, ldw, nrhs, upd.p, ldu, st, true, Tsub - 1, l_to);
        } else mf_sweep_levels(ctx, P, mf_single(Fc), 1, W, ldw, nrhs, upd.p, ldu, st, true, S.nlevels - 1, l_to, in);
    };
    auto backward_from = [&](int l_from) {   // levels l_from .. nlevels-1
        if (Tsub >= 0 && Tsub >= l_from) {
            if (Tsub - 1 >= l_from) mf_sweep_levels(ctx, P, mf_single(Fc), 1, W, ldw, nrhs, upd.p, ldu, st, false, l_from, Tsub - 1);
            mf_sub_sweep(ctx, P, Fc, W, ldw, nrhs, upd.p, ldu, st, false);
        } else mf_sweep_levels(ctx, P, mf_single(Fc), 1, W, ldw, nrhs, upd.p, ldu, st, false, l_from, S.nlevels - 1);
    };
    if (Fc.topinv.empty()) {
        TimedScope ts(ctx, "mf_solve_real", bytes, flops);
        forward_to(0);
        backward_from(0);
    } else {
        const TopPlan& tp = P.top;
        const int ntop = tp.ntop, T = tp.T;
        Mat g(ctx, ntop, nrhs);
        const size_t tot = (size_t)ntop * nrhs;
        {
            TimedScope ts(ctx, "mf_solve_real", bytes, flops);
            forward_to(T);
            hipLaunchKernelGGL(k_top_gather, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, ctx->stream, ntop, nrhs, (const int*)tp.topidx.p,
                               (const int*)tp.gptr.p, (const int64_t*)tp.gsrc.p, (const double*)W, ldw, (const double*)upd.p, ldu, g.p, g.ld, st, in.p, in.ld, in.n, 0L, 0L, 0L);
        }
        // x_top = inv(S) g as split-K slabs; their fixed-order sum lands directly in the rows of the panel (no x, no scatter launch)
        int zs = 1;
        BufP xpart = gemm_partials(ctx, false, false, ntop, nrhs, ntop, Fc.topinv.p, Fc.topinv.ld, g.p, g.ld, &zs, st, "gemm_mf_top");
        {
            TimedScope ts(ctx, "mf_solve_real", 0.0, 0.0);
            gemm_reduce_rows(ctx, ntop, nrhs, zs, (const double*)xpart->p, (const int*)tp.topidx.p, W, ldw, st);
            backward_from(T);
        }
    }
    DRE_HIP(hipGetLastError());
}

// the dense top inverses of nz factors in shared launches (unit right-hand sides on the top variables, sweeps over the top levels only)
__global__ void k_top_unit_rows_z(int nrhs, int c0, const int* __restrict__ topidx, double* __restrict__ W, int ldw, long wz) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < nrhs) W[(size_t)blockIdx.y * wz + topidx[c0 + j] + (size_t)j * ldw] = 1.0;
}
struct TopOutZ { double* Ti[MF_ZMAX]; };
__global__ void k_top_collect_z(int ntop, int nrhs, int c0, const int* __restrict__ topidx, const double* __restrict__ W, int ldw, long wz, TopOutZ out, int ldt) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)ntop * nrhs) return;
    const int p = idx % ntop; const size_t c = idx / ntop;
    out.Ti[blockIdx.y][p + (c0 + c) * ldt] = W[(size_t)blockIdx.y * wz + topidx[p] + c * ldw];
}
void mf_topinv_batch(Ctx* ctx, const Pencil& P, Factor<double>* const* Fs, int nz) {
    if (!P.use_mfma_sweeps || ctx->top_inverse_max_rows <= 0 || nz < 1) return;
    if (!P.top.built) top_plan_build(ctx, P, ctx->top_inverse_max_rows);
    if (!P.sub.built) sub_plan_build(ctx, P, P.top.T);
    if (P.top.T < 2) return;
    const TopPlan& tp = P.top;
    const Symbolic& S = P.sym;
    const int ntop = tp.ntop, T = tp.T, n = P.n;
    const int64_t ldu = std::max<int64_t>(S.upd_rows, 1);
    // at most ~2 GB of work panels at a time
    const int chunk = 512;
    const size_t per_z = ((size_t)n + (size_t)ldu) * chunk * sizeof(double);
    const int zmax = (int)std::max<size_t>(1, std::min<size_t>((size_t)MF_ZMAX, ((size_t)2 << 30) / std::max<size_t>(per_z, 1)));
    for (int z0 = 0; z0 < nz; z0 += zmax) {
        const int g = std::min(zmax, nz - z0);
        TimedScope ts(ctx, "mf_top_inverse", 8.0 * (double)ntop * ntop * g, 0.0, g);
        MfZ zb; std::memset(&zb, 0, sizeof(zb));
        TopOutZ out; std::memset(&out, 0, sizeof(out));
        std::vector<Mat> Tis;
        for (int z = 0; z < g; ++z) {
            Tis.push_back(Mat(ctx, ntop, ntop));
            out.Ti[z] = Tis.back().p;
            zb.fronts[z] = Fs[z0 + z]->fronts.p; zb.inv[z] = Fs[z0 + z]->inv.p;
        }
        for (int c0 = 0; c0 < ntop; c0 += chunk) {
            const int ch = std::min(chunk, ntop - c0);
            Mat Wk(ctx, n, ch * g);
            DevArr<double> upd(ctx, (size_t)ldu * ch * g);
            zb.wz = (long)ch * Wk.ld; zb.uz = (long)ldu * ch;
            DRE_HIP(hipMemsetAsync(Wk.p, 0, (size_t)n * ch * g * sizeof(double), ctx->stream));
            DRE_HIP(hipMemsetAsync(upd.p, 0, (size_t)ldu * ch * g * sizeof(double), ctx->stream));
            hipLaunchKernelGGL(k_top_unit_rows_z, dim3(ceil_div(ch, 256), g), dim3(256), 0, ctx->stream, ch, c0, (const int*)tp.topidx.p, Wk.p, Wk.ld, zb.wz);
            mf_sweep_levels(ctx, P, zb, g, Wk.p, Wk.ld, ch, upd.p, ldu, nullptr, true, T - 1, 0);
            mf_sweep_levels(ctx, P, zb, g, Wk.p, Wk.ld, ch, upd.p, ldu, nullptr, false, 0, T - 1);
            const size_t tot = (size_t)ntop * ch;
            hipLaunchKernelGGL(k_top_collect_z, dim3((unsigned)((tot + 255) / 256), g), dim3(256), 0, ctx->stream, ntop, ch, c0, (const int*)tp.topidx.p,
                               (const double*)Wk.p, Wk.ld, zb.wz, out, ntop);
        }
        for (int z = 0; z < g; ++z) { Fs[z0 + z]->topinv = Tis[(size_t)z]; Fs[z0 + z]->uses = 3; }
    }
    DRE_HIP(hipGetLastError());
}
void mf_prepare_topinv(Ctx* ctx, const Pencil& P, const Factor<double>& Fc) {
    if (!P.use_mfma_sweeps || !Fc.allow_topinv || ctx->top_inverse_max_rows <= 0 || !Fc.topinv.empty()) return;
    if (!P.top.built) top_plan_build(ctx, P, ctx->top_inverse_max_rows);
    if (!P.sub.built) sub_plan_build(ctx, P, P.top.T);
    if (P.top.T >= 2) mf_build_topinv(ctx, P, Fc);
    Fc.uses = 3;
}
// W = F^-1 [Win(:, 0:nin) | W(:, nin:nrhs)]: the leading right-hand sides are read where they are (no copy into the panel) when the
// matrix-core sweeps run; otherwise they are copied in and the in-place solve follows.
void mf_solve_from(Ctx* ctx, const Pencil& P, const Factor<double>& Fc, const double* Win, int ldwin, int nin, double* W, int ldw, int nrhs, const AdiState* st) {
    if (nrhs <= 0) return;
    DRE_REQUIRE(nin >= 0 && nin <= nrhs && Win != W, "mf_solve_from: bad column split");
    if (P.use_mfma_sweeps && Fc.nperturbed <= 0) {
        MfIn in; in.p = Win; in.ld = ldwin; in.n = nin;
        mf_solve_mfma(ctx, P, Fc, W, ldw, nrhs, st, &in);
        return;
    }
    if (nin > 0) hipLaunchKernelGGL(k_copy_cols, dim3(ceil_div(P.n, 256), nin), dim3(256), 0, ctx->stream, P.n, Win, ldwin, W, ldw, st);
    mf_solve<double>(ctx, P, Fc, W, ldw, nrhs, st);
}
// g solves with DIFFERENT factors of the same pencil and the same leading right-hand sides, in shared launches (blockIdx.z = factor):
//   W_z = F_z^-1 [Win(:, 0:nin) | W_z(:, nin:nrhs)],   W_z = columns z nrhs .. (z + 1) nrhs of the n x (g nrhs) panel W.
// The sweeps are latency bound (a level launch keeps a fraction of the chip busy for ~9 us): g of them per launch cost about the same as one.
// Returns false — nothing enqueued — where the batch form does not apply (scalar sweeps, statically pivoted factors, subtree plan).
bool mf_solve_batch(Ctx* ctx, const Pencil& P, const Factor<double>* const* Fs, int g, const double* Win, int ldwin, int nin, double* W, int ldw, int nrhs,
                    const AdiState* st) {
    if (g < 1 || g > MF_ZMAX || nrhs <= 0 || !P.use_mfma_sweeps) return false;
    for (int z = 0; z < g; ++z) if (Fs[z]->nperturbed > 0) return false;
    const Symbolic& S = P.sym;
    if (!P.sub.built) {
        if (!P.top.built && ctx->top_inverse_max_rows > 0) top_plan_build(ctx, P, ctx->top_inverse_max_rows);
        sub_plan_build(ctx, P, P.top.T);
    }
    if (P.sub.Tsub >= 0) return false;
    // the dense top: all factors with it or none (a reusable factor that lacks it gets it now, on this stream)
    bool all_allowed = ctx->top_inverse_max_rows > 0 && P.top.built && P.top.T >= 2 && nrhs >= 8;
    int have = 0;
    for (int z = 0; z < g; ++z) { all_allowed = all_allowed && Fs[z]->allow_topinv; have += Fs[z]->topinv.empty() ? 0 : 1; }
    if (have != g) {
        if (all_allowed) { for (int z = 0; z < g; ++z) if (Fs[z]->topinv.empty()) { mf_build_topinv(ctx, P, *Fs[z]); Fs[z]->uses = 3; } have = g; }
        else if (have != 0) return false;
    }
    const bool top = have == g;
    MfIn in; in.p = Win; in.ld = ldwin; in.n = nin;
    const int64_t ldu = std::max<int64_t>(S.upd_rows, 1);
    MfZ zb; std::memset(&zb, 0, sizeof(zb));
    for (int z = 0; z < g; ++z) { zb.fronts[z] = Fs[z]->fronts.p; zb.inv[z] = Fs[z]->inv.p; }
    zb.wz = (long)nrhs * ldw; zb.uz = (long)ldu * nrhs;
    DevArr<double> upd(ctx, (size_t)ldu * nrhs * g);
    const double bytes = g * (2.0 * 8.0 * (double)S.factor_nnz + 4.0 * 8.0 * (double)P.n * nrhs);
    const double flops = g * (2.0 * 2.0 * (double)S.factor_nnz * nrhs);
    if (!top) {
        TimedScope ts(ctx, "mf_solve_real", bytes, flops, g);
        mf_sweep_levels(ctx, P, zb, g, W, ldw, nrhs, upd.p, ldu, st, true, S.nlevels - 1, 0, in);
        mf_sweep_levels(ctx, P, zb, g, W, ldw, nrhs, upd.p, ldu, st, false, 0, S.nlevels - 1);
        DRE_HIP(hipGetLastError());
        return true;
    }
    const TopPlan& tp = P.top;
    const int ntop = tp.ntop, T = tp.T;
    Mat gb(ctx, ntop, nrhs * g);
    const size_t tot = (size_t)ntop * nrhs;
    {
        TimedScope ts(ctx, "mf_solve_real", bytes, flops, g);
        mf_sweep_levels(ctx, P, zb, g, W, ldw, nrhs, upd.p, ldu, st, true, S.nlevels - 1, T, in);
        hipLaunchKernelGGL(k_top_gather, dim3((unsigned)((tot + 255) / 256), g), dim3(256), 0, ctx->stream, ntop, nrhs, (const int*)tp.topidx.p,
                           (const int*)tp.gptr.p, (const int64_t*)tp.gsrc.p, (const double*)W, ldw, (const double*)upd.p, ldu, gb.p, gb.ld, st, in.p, in.ld, in.n,
                           zb.wz, zb.uz, (long)ntop * nrhs);
    }
    GemmZ gz; std::memset(&gz, 0, sizeof(gz));
    for (int z = 0; z < g; ++z) { gz.A[z] = Fs[z]->topinv.p; gz.B[z] = gb.p + (size_t)z * ntop * nrhs; DRE_REQUIRE(Fs[z]->topinv.ld == Fs[0]->topinv.ld, "mf_solve_batch: top inverses of different shape"); }
    int zs = 1;
    BufP xpart = gemm_partials_z(ctx, false, false, ntop, nrhs, ntop, gz, g, Fs[0]->topinv.ld, gb.ld, &zs, st, "gemm_mf_top");
    {
        TimedScope ts(ctx, "mf_solve_real", 0.0, 0.0, 0);
        gemm_reduce_z(ctx, ntop, nrhs, zs, g, (const double*)xpart->p, (const int*)tp.topidx.p, W, ldw, zb.wz, st);
        mf_sweep_levels(ctx, P, zb, g, W, ldw, nrhs, upd.p, ldu, st, false, T, S.nlevels - 1);
    }
    DRE_HIP(hipGetLastError());
    return true;
}
template <typename T>
static void mf_solve_once(Ctx* ctx, const Pencil& P, const Factor<T>& Fc, T* W, int ldw, int nrhs, const AdiState* st);
__global__ void k_refine_axpy(int n, int nrhs, const double* __restrict__ d, int ldd, double* __restrict__ x, int ldx, const AdiState* st) {
    if (st && st->done) return;
    const size_t id = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (id >= (size_t)n * nrhs) return;
    const int r = id % n, c = id / n;
    x[r + (size_t)c * ldx] += d[r + (size_t)c * ldd];
}
template <typename T>
void mf_solve(Ctx* ctx, const Pencil& P, const Factor<T>& Fc, T* W, int ldw, int nrhs, const AdiState* st) {
    if (nrhs <= 0) return;
    if (Fc.nperturbed > 0) {
        // the factor belongs to a perturbed matrix (static pivoting): fixed-point refinement against the true operator M = cF F' + cE E'
        if constexpr (sizeof(T) == sizeof(double)) {
            const int n = P.n;
            Mat B0(ctx, n, nrhs), Rr(ctx, n, nrhs), X(ctx, 0, 0);
            Mat Wm; Wm.p = W; Wm.rows = n; Wm.cols = nrhs; Wm.ld = ldw;       // header over the caller's panel
            copy_mat(ctx, Wm, B0, 1.0, st);
            mf_solve_once<T>(ctx, P, Fc, W, ldw, nrhs, st);
            for (int it = 0; it < ctx->pivot_refine_steps; ++it) {
                copy_mat(ctx, B0, Rr, 1.0, st);
                spmm(ctx, P, Fc.ref_valF, Wm, Rr, -Fc.ref_cF, 1.0, st);         // r = b - M x
                spmm(ctx, P, Fc.ref_valE, Wm, Rr, -Fc.ref_cE, 1.0, st);
                mf_solve_once<T>(ctx, P, Fc, Rr.p, Rr.ld, nrhs, st);
                hipLaunchKernelGGL(k_refine_axpy, dim3((unsigned)(((size_t)n * nrhs + 255) / 256)), dim3(256), 0, ctx->stream, n, nrhs, (const double*)Rr.p, Rr.ld, W, ldw, st);
            }
            return;
        } else {
            throw Error(ERR_SINGULAR, "mf_solve: the complex shifted operator needed static pivots; refinement is implemented for real shifts only — "
                                      "use a user block solver (dre_adi_options.inner_solve) for this pencil");
        }
    }
    mf_solve_once<T>(ctx, P, Fc, W, ldw, nrhs, st);
}
template <typename T>
static void mf_solve_once(Ctx* ctx, const Pencil& P, const Factor<T>& Fc, T* W, int ldw, int nrhs, const AdiState* st) {
    if (nrhs <= 0) return;
    if constexpr (sizeof(T) == sizeof(double)) {
        if (P.use_mfma_sweeps) { mf_solve_mfma(ctx, P, Fc, W, ldw, nrhs, st); return; }
    }
    const Symbolic& S = P.sym;
    MfArgs a = mf_args(P);
    const int64_t ldu = std::max<int64_t>(S.upd_rows, 1);
    DevArr<T> upd(ctx, (size_t)ldu * nrhs);
    const int ncb = ceil_div(nrhs, MF_KC);
    // algorithmic bytes: factor entries once per sweep + panel read/write per sweep
    const double bytes = 2.0 * sizeof(T) * (double)S.factor_nnz + 4.0 * sizeof(T) * (double)P.n * nrhs;
    const double flops = (sizeof(T) == 8 ? 2.0 : 8.0) * 2.0 * (double)S.factor_nnz * nrhs;
    TimedScope ts(ctx, sizeof(T) == 8 ? "mf_solve_real" : "mf_solve_complex", bytes, flops);
    for (int l = S.nlevels - 1; l >= 0; --l) {
        const int nb = S.lvl_ptr[l + 1] - S.lvl_ptr[l];
        const size_t shm = (size_t)(P.lvl_maxfront[l] * 2) * MF_KC * sizeof(T);
        hipLaunchKernelGGL((k_mf_forward<T>), dim3(nb, ncb), dim3(256), shm, ctx->stream, a, S.lvl_ptr[l], Fc.fronts.p, Fc.inv.p, W, ldw, nrhs, upd.p, ldu, st);
    }
    for (int l = 0; l < S.nlevels; ++l) {
        const int nb = S.lvl_ptr[l + 1] - S.lvl_ptr[l];
        const size_t shm = (size_t)(P.lvl_maxfront[l]) * MF_KC * sizeof(T);
        hipLaunchKernelGGL((k_mf_backward<T>), dim3(nb, ncb), dim3(256), shm, ctx->stream, a, S.lvl_ptr[l], Fc.fronts.p, Fc.inv.p, W, ldw, nrhs, st);
    }
    DRE_HIP(hipGetLastError());
}
template void mf_solve<double>(Ctx*, const Pencil&, const Factor<double>&, double*, int, int, const AdiState*);
template void mf_solve<cplx>(Ctx*, const Pencil&, const Factor<cplx>&, cplx*, int, int, const AdiState*);

}  // namespace dre
