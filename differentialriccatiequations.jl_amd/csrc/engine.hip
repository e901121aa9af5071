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
void smw_sinv_fold_batched(Ctx* ctx, int n, int m, double alpha, const SmwBatch* items, int count, int* serr) {
    for (int b0 = 0; b0 < count; b0 += 16) {
        SmwBatchArgs ba;
        const unsigned nb = (unsigned)std::min(16, count - b0);
        for (unsigned i = 0; i < 16; ++i) ba.b[i] = items[b0 + (int)(i < nb ? i : 0)];
        TimedScope ts(ctx, "smw_batched", 8.0 * nb * (2.0 * n * m * 2.0 + 3.0 * m * m), 4.0 * nb * n * (double)m * m);
        hipLaunchKernelGGL(k_sinv_batched_args, dim3(nb), dim3(64), 0, ctx->stream, n, m, 2 * n + m, alpha, ba, serr);
        hipLaunchKernelGGL(k_fold_sinv_batched_args, dim3(ceil_div(2 * n * m, 256), nb), dim3(256), 0, ctx->stream, 2 * n, m, 2 * n + m, ba);
    }
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
    // index (in the strategy's own list) of the shift take() returns next; 0 where there is no list
    virtual size_t position() const { return 0; }
    virtual size_t list_size() const { return 0; }
    // the strategy saw the device-side `done` flag while refilling: nothing further needs to be enqueued
    virtual bool stopped() const { return false; }
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
    size_t position() const override { return v.empty() ? 0 : i % v.size(); }
    size_t list_size() const override { return v.size(); }
};

void apply_Ft(Ctx* ctx, const GaleOperator& op, const Mat& L, Mat& out) {
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
    bool saw_done = false;
    bool stopped() const override { return saw_done; }
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
            if (done) { buffer.assign(1, std::complex<double>(-1.0, 0.0)); pos = 0; saw_done = true; return; }
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
        if (kq == w && w >= 1) {
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
        if (!full_rank && fro * 3e-8 < P.n * EPS && fro > 0.0) {
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
        } else if (!full_rank) host_rrqr_svd_left(kq, w, hR, P.n * EPS, Us, sv);      // steeply graded history block: pivoted QR + small SVD (R = Us diag(sv) W')
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

// User-defined strategy (dre_shift_fn, include/dre_hip.h): the reference's protocol init / update! / take!  (src/Shifts.jl:79-116) with the batch
// form of a BufferedIterator (shifts/helpers.jl:60-89).  update() keeps handles of what the reference hands to update! (R at the start, then the
// increments; the last n_history of them); when the buffer runs dry the history is laid out as one n x w block and the callback fills the buffer.
struct UserOracle : ShiftOracle {
    Ctx* ctx = nullptr; int n = 0; int n_history = 2;
    const int* iperm = nullptr;         // device: solver ordering -> caller's ordering (Pencil::iperm, as for the user block solver)
    ShiftFn fn = nullptr; void* user = nullptr;
    const AdiState* st_dev = nullptr;   // speculative enqueueing: a converged solve asks for nothing further
    std::vector<Mat> Vs;
    std::vector<std::complex<double>> buffer;
    size_t pos = 0;
    bool first = true, saw_done = false;
    bool stopped() const override { return saw_done; }
    void update(const Mat& R, const std::vector<Mat>& newVs) override {
        if (newVs.empty()) Vs.push_back(R);
        for (auto& v : newVs) Vs.push_back(v);
        const int lst = (int)Vs.size();
        Vs.erase(Vs.begin(), Vs.begin() + std::max(0, lst - n_history));
    }
    void take_many() {
        int done = 0;
        if (st_dev) DRE_HIP(hipMemcpyAsync(&done, &st_dev->done, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        DRE_HIP(hipStreamSynchronize(ctx->stream));
        if (done) { buffer.assign(1, std::complex<double>(-1.0, 0.0)); pos = 0; saw_done = true; return; }
        int w = 0;
        for (auto& v : Vs) w += v.cols;
        Mat N(ctx, n, std::max(w, 1));
        int off = 0;
        for (auto& v : Vs) { if (v.cols == 0) continue; Mat d = N.colsview(off, v.cols); permute_rows(ctx, v, iperm, d); off += v.cols; }      // caller's row ordering
        DRE_HIP(hipStreamSynchronize(ctx->stream));
        constexpr int CAP = 1024;
        std::vector<double> re((size_t)CAP, 0.0), im((size_t)CAP, 0.0);
        int count = 0;
        const int rc = fn(user, first ? 1 : 0, n, w, N.p, N.ld, CAP, re.data(), im.data(), &count);
        first = false;
        DRE_REQUIRE(rc == 0, "user-defined shift strategy failed (shift_fn returned non-zero)");
        DRE_REQUIRE(count >= 1 && count <= CAP, "user-defined shift strategy: count out of range");
        buffer.clear();
        for (int i = 0; i < count; ++i) {
            DRE_REQUIRE(std::isfinite(re[(size_t)i]) && std::isfinite(im[(size_t)i]) && re[(size_t)i] < 0.0, "user-defined shift strategy: shifts need a negative real part");
            buffer.emplace_back(re[(size_t)i], im[(size_t)i]);
        }
        pos = 0;
    }
    std::complex<double> take(int*) override {
        if (pos >= buffer.size()) take_many();
        return buffer[pos++];
    }
    std::vector<std::complex<double>> peek(size_t count) const override {
        std::vector<std::complex<double>> out;
        for (size_t i = pos; i < buffer.size() && out.size() < count; ++i) out.push_back(buffer[i]);
        return out;
    }
};


// =============================================================================================
// ADI (/root/reference/src/lyapunov/adi.jl:29-225)
// =============================================================================================
// Dense inverses whose acceptance test (condition estimate ||M||_F ||inv(M)||_F, two norms in device memory) is still outstanding: a run
// over all shifts of a cycle enqueues every factorisation and inverse first and reads all norms back with ONE synchronisation
// (finalize_dense) instead of two per shift.
void finalize_dense(Ctx* ctx, DeferredDense& dd) {
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
std::shared_ptr<FactorEntry<T>> get_factor(Ctx* ctx, const GaleOperator& op, FactorCache* cache,
                                           std::map<std::tuple<uint64_t, double, double>, std::shared_ptr<FactorEntry<T>>>& store,
                                           std::complex<double> mu, bool want_dense, DeferredDense* defer, bool check_now) {
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
template std::shared_ptr<FactorEntry<double>> get_factor<double>(Ctx*, const GaleOperator&, FactorCache*, std::map<std::tuple<uint64_t, double, double>, std::shared_ptr<FactorEntry<double>>>&,
                                                                 std::complex<double>, bool, DeferredDense*, bool);
template std::shared_ptr<FactorEntry<cplx>> get_factor<cplx>(Ctx*, const GaleOperator&, FactorCache*, std::map<std::tuple<uint64_t, double, double>, std::shared_ptr<FactorEntry<cplx>>>&,
                                                             std::complex<double>, bool, DeferredDense*, bool);

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
    RoctxRange roctx_range("Sherman-Morrison-Woodbury");
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
Ctx* helper_ctx(Ctx* ctx, int h) {
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
    hc->gemm_swizzle = ctx->gemm_swizzle; hc->mf_swizzle = ctx->mf_swizzle;
    return hc;
}
hipEvent_t aux_event(Ctx* ctx, int i) {
    if (!ctx->aux_ev[i]) DRE_HIP(hipEventCreateWithFlags(&ctx->aux_ev[i], hipEventDisableTiming));
    return ctx->aux_ev[i];
}
// Zero-increment guard (adi.jl:200-204 for every complex pair; :161-165 for a real step under mixed precision, which this engine does not
// have): the solve returned V = 0 exactly although the residual is above the tolerance.  The reference warns, sets the increment to zero,
// leaves X and the residual alone and stops (isdone, adi.jl:134-137) — with the pair's two shifts counted.  V = 0 <=> V1 = V2 = 0.  Every
// workgroup ORs what it saw into the control block; the last arrival takes the decision: done, iters += nshifts, the norm recorded unchanged,
// `collapsed` = 2 for the host (DRE_WARN_ZERO_INCREMENT).  The step's remaining kernels see `done` and leave R and the norms as they are.
__global__ __launch_bounds__(256) void k_zero_increment(size_t tot, int n, int k, const double* __restrict__ V1, int ld1, const double* __restrict__ V2, int ld2,
                                                        AdiState* st, int nshifts) {
    if (st->done) return;
    __shared__ int any_sh;
    if (threadIdx.x == 0) any_sh = 0;
    __syncthreads();
    int any = 0;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < tot; e += (size_t)gridDim.x * 256) {
        const int r = (int)(e % (size_t)n), c = (int)(e / (size_t)n);
        if (V1[r + (size_t)c * ld1] != 0.0 || V2[r + (size_t)c * ld2] != 0.0) { any = 1; break; }
    }
    if (any) any_sh = 1;
    __syncthreads();
    if (threadIdx.x != 0) return;
    if (any_sh) atomicOr(&st->collapsed, 1);
    __threadfence();
    if (atomicAdd(&st->ticket, 1) != (int)gridDim.x - 1) return;
    __threadfence();
    st->ticket = 0;
    const int seen = __hip_atomic_load(&st->collapsed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (seen & 1) { st->collapsed = 0; return; }
    st->collapsed = 2;
    st->iters += nshifts;
    st->norms[st->iters & 511] = st->res_norm;
    st->done = 1;
}
void zero_increment_guard(Ctx* ctx, const Mat& V1, const Mat& V2, AdiState* st, int nshifts) {
    const size_t tot = (size_t)V1.rows * V1.cols;
    if (tot == 0) return;
    const unsigned nb = (unsigned)std::min<size_t>(256, (tot + 4095) / 4096);
    hipLaunchKernelGGL(k_zero_increment, dim3(nb), dim3(256), 0, ctx->stream, tot, V1.rows, V1.cols, (const double*)V1.p, V1.ld, (const double*)V2.p, V2.ld, st, nshifts);
    DRE_HIP(hipGetLastError());
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
            if (opt.warm_eig_basis.cols >= 16 && lag > 0.0 && ctx->dense_warm != 0) {
                int J = -1; double er = -1.0;
                Mat Qn;
                done = warm_compress_eig(ctx, *resid, opt.warm_eig_basis, lag, opt.residual_abs_frac, opt.warm_est_ratio, opt.warm_J_prev, Qn, &J, &er);
                if (done) { res.warm_basis_out = Qn; res.warm_J = J; res.warm_est_ratio = er; }
            }
            // (not attempted where it cannot apply — a basis of fewer than 16 columns, q0 + sx <= 64 — so that such a step does not count as a rejection)
            if (!done && opt.warm_basis.cols >= 16 && lag > 0.0 && cache->warm_strikes < 2 &&
                opt.warm_basis.cols + (cache->warm_sx > 0 ? cache->warm_sx : 32) >= 48) {
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
    } else if (opt.shifts.kind == ShiftSpec::USER) {
        DRE_REQUIRE(opt.shifts.user_fn != nullptr, "user-defined shift strategy: no callback");
        auto o = std::make_unique<UserOracle>();
        o->ctx = ctx; o->n = n; o->n_history = opt.shifts.n_history; o->fn = opt.shifts.user_fn; o->user = opt.shifts.user_data;
        o->iperm = op.P->iperm.p;
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
    if (auto* uo = dynamic_cast<UserOracle*>(oracle.get())) uo->st_dev = st.p;
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
            // the tolerance is formed beside the SMW set-up and the first group's sweeps (a stream and a host thread of the caller's); this stream
            // picks it up in front of its first decision, the host reads it with the first chunk
            hipEvent_t e0 = aux_event(ctx, 0), e1 = aux_event(ctx, 1);
            DRE_HIP(hipEventRecord(e0, ctx->stream));
            opt.normC_build(e0, R, Tm, alpha_res, e1);
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
    // Sharded by shift (a communicator with real ranks): a rank owns the shifts at the list positions i = rank (mod P) — fixed for the run, whatever
    // the group boundaries turn out to be — and factorises exactly those here, in shared launches, like the single rank does with the whole list
    // (round 4 switched the batched set-up off in sharded mode: every rank factorised its shifts one by one inside the first groups).
    if (opt.shifts.kind == ShiftSpec::CYCLIC && !opt.inner_solve && cache->enabled && n > ctx->dense_inv_max_n && ctx->adi_fan >= 2 && P.use_mfma_sweeps) {
        const bool real_ranks = ctx->comm && ctx->comm->nranks > 1 && ctx->comm->emulate <= 1;
        std::vector<double> todo;
        for (size_t iv = 0; iv < opt.shifts.values.size(); ++iv) {
            const auto& mu = opt.shifts.values[iv];
            if (mu.imag() != 0.0) { todo.clear(); break; }
            if (real_ranks && (int)(iv % (size_t)ctx->comm->nranks) != ctx->comm->rank) continue;
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
        if (real_ranks) {
            // The batched solves refuse factors with replaced pivots (static pivoting) — a property of a rank's OWN shifts.  One rank leaving the
            // fan path alone would leave the others inside the group's all-gather, so the ranks agree once per operator: every
            // rank checks the factors it owns (one read-back) and the flags are summed across the communicator; any refusal switches the fan
            // groups off on ALL ranks for this operator (the column-sharded iterations take over).
            auto ag = cache->fan_agreed.find(op.tag);
            if (ag == cache->fan_agreed.end()) {
                std::vector<const Factor<double>*> fs;
                std::vector<std::shared_ptr<FactorEntry<double>>> mine_fe;
                for (size_t iv = 0; iv < opt.shifts.values.size(); ++iv) {
                    if ((int)(iv % (size_t)ctx->comm->nranks) != ctx->comm->rank) continue;
                    auto itf = cache->real.find(std::make_tuple(op.tag, opt.shifts.values[iv].real(), 0.0));
                    if (itf == cache->real.end()) continue;
                    mine_fe.push_back(itf->second);
                    if (!itf->second->checked) fs.push_back(&itf->second->f);
                }
                if (!fs.empty()) {
                    const std::vector<double> gr = mf_check_batch(ctx, fs);
                    size_t j = 0;
                    for (auto& fe : mine_fe) if (!fe->checked) { fe->growth = gr[j++]; fe->checked = true; }
                }
                double flag = 0.0;
                for (auto& fe : mine_fe) if (fe->f.nperturbed > 0) flag = 1.0;
                DevArr<double> fl(ctx, 1);
                DRE_HIP(hipMemcpyAsync(fl.p, &flag, sizeof(double), hipMemcpyHostToDevice, ctx->stream));
                comm_allreduce_sum(ctx, *ctx->comm, fl.p, 1);
                double tot = 0.0;
                ctx_fetch(ctx, fl.p, sizeof(double), &tot);
                ag = cache->fan_agreed.emplace(op.tag, tot > 0.0).first;
            }
            if (ag->second) run.fan_off = true;
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
        // (The residual-recurrence caller takes the increments as they are — its side stream compresses X whatever the number of columns and the next
        // residual is compressed by the range finder, both GEMM passes over the slabs — so neither the chunk nor the iterate is held to the
        // factor-form limit there: at n = 5177 the first steps (k = 144 ... 64, 30-39 iterations: up to 5 600 columns) lost their history to an
        // in-loop compression and fell back to the reference's order, 6.4 instead of ~4 ms per step.)
        const bool free_form = opt.given_residual && opt.keep_history && !opt.final_compress && ctx->recurrence_wide != 0;
        const long room = free_form ? (1L << 20) : ((long)n - 64 - X->rank()) / k - 2L * opt.compression_interval - FAN_GMAX;
        run.chunk_limit = (int)std::max<long>(opt.compression_interval, std::min<long>(cache->iters_hint + 1, room));
        run.chunk_from_hint = true;       // exactly that many iterations: the last group of the chunk is cut short (a speculative group costs 14 launches)
    }
    // ---- fast chain (dense.hip, k_adi_fast): every shift of the cycle is real and already has its stacked dense inverse for the
    // current low-rank factor (i.e. from the second time step of a run on) and the residual is at most 96 columns wide ----------
    bool fast = !opt_in.inner_solve && opt_in.shifts.kind == ShiftSpec::CYCLIC && k >= 1 && k <= ADI_FAST_MAX_K && n <= ctx->dense_inv_max_n && cache->enabled;
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
    static const bool trace_prefetch = env_trace("prefetch");
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
    (void)P; (void)opt_in; (void)serr;
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
            const bool use_pk = a.mode == 0;
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
            if (opt.normC_join) opt.normC_join();          // (`done` is recorded by then)
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
        const bool sharded_la = ctx->comm && std::max(ctx->comm->nranks, ctx->comm->emulate) > 1;
        const bool cyc_ahead = opt_in.shifts.kind == ShiftSpec::CYCLIC && !opt.inner_solve && cache->enabled && n > ctx->dense_inv_max_n && !sharded_la;
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
            const int nh = ctx->setup_streams >= 1 ? std::max(8, ctx->setup_streams) : 0;
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
                long &calls = ctx->trace.pf_calls, &nups = ctx->trace.pf_nups, &pend = ctx->trace.pf_pend;
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
                // the shift in hand rides along when it has no factor yet (the first shift of a refilled batch: factorised inline on the main stream
                // it was ~230 us of eight dependent launches per refill; in the batch it shares them with the next fifteen)
                if (!fan_call) {
                    const bool ccx = cur.imag() != 0.0;
                    const auto cck = std::make_tuple(op.tag, cur.real(), cur.imag());
                    const bool cknown = ccx ? cache->cplx_.count(cck) > 0 : cache->real.count(cck) > 0;
                    if (!cknown && !run.prefetch_ev.count({cur.real(), cur.imag()})) (ccx ? tc : tr).push_back(cur);
                }
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
                    if (cyc_ahead) mf_prepare_topinv(hc, *op.P, fnew->f);
                }
                DRE_HIP(hipEventRecord(ev, hc->stream));
                const long ticket = 0;
                run.prefetch_ev[{nx.real(), nx.imag()}] = AdiRun::Prefetched{ev, ticket};
                ++scheduled;
            }
        };
        // Fan groups (round 4: batched, and sharded over the ranks of a communicator BY SHIFT — north_star's "independent ADI shifts farmed across
        // the GPUs"): the g solves of a group are independent, so rank r takes the shifts at the LIST positions i = r (mod P), i.e. only ever factorises and
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
                // ownership by the shift's position in the LIST (not in the group): the same shift always meets the same rank, so a rank factorises
                // and keeps only its share of the list (adi_begin factorises that share up front)
                const size_t pos0 = oracle->position(), lsz = oracle->list_size();
                auto owner = [&](int s_) { return (int)((lsz ? (pos0 + (size_t)s_) % lsz : (size_t)s_) % (size_t)fan_P); };
                auto mine = [&](int s_) { return fan_P == 1 || emu || owner(s_) == my_rank; };
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
                    double* const ft = ctx->trace.fan_t; long& fn_ = ctx->trace.fan_n;
                    auto fnow = []() { return std::chrono::steady_clock::now(); };
                    auto fus = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
                    const auto f0 = fnow();
                    const AdiState* dst = st.p;
                    dense_norm_flush(ctx, k, Tm, tdiag, alpha_res, st.p, &npend);
                    // group positions per rank (a group that wraps around the end of the list may give one rank more than g / P); W_s lives in slab
                    // owner(s), in the owner's order
                    int gp = 0;
                    FanSlots slots; std::memset(&slots, 0, sizeof(slots));
                    {
                        int seen[64] = {0};
                        for (int s_ = 0; s_ < g; ++s_) gp = std::max(gp, ++seen[owner(s_) & 63]);
                        std::memset(seen, 0, sizeof(seen));
                        for (int s_ = 0; s_ < g; ++s_) { const int o = owner(s_); slots.s[s_] = o * gp + seen[o & 63]++; }
                    }
                    Mat Wcat(ctx, n, fan_P * gp * k);
                    bool ok = true;
                    for (int r = 0; r < fan_P && ok; ++r) {
                        if (!(fan_P == 1 || emu || r == my_rank)) continue;
                        std::vector<int> pos;                          // this rank's group positions
                        for (int s_ = 0; s_ < g; ++s_) if (owner(s_) == r) pos.push_back(s_);
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
            std::complex<double> mu;
            { RoctxRange rr("shifts"); mu = oracle->take(&res.warnings); }        // adi.jl:101
            // (a refill that found the solve converged: the rest of the chunk would be early-exit launches with a synchronisation per refill)
            if (oracle->stopped() && !recs.empty()) break;
            all_shifts.push_back(mu);
            const bool is_real = (mu.imag() == 0.0);
            RoctxRange roctx_solve(is_real ? "solve (real)" : "solve (complex)");        // adi.jl:157,196 (the range covers the whole step)
            const AdiState* dst = st.p;
            if (lookahead) { wait_prefetched(mu); prefetch_ahead(mu); wait_prefetched(mu); }        // (the batch may carry the shift in hand: its event is waited for behind it)
            Mat V1, V2;
            bool norm_done = false, rode = false;
            if (is_real) {
                // dense inverses pay off only for shift lists that persist across Lyapunov solves (user-given Cyclic values)
                const bool user_inner = opt.inner_solve != nullptr;
                std::shared_ptr<FactorEntry<double>> fe;
                if (!user_inner) {
                    if (single_use && trace_prefetch) {
                        long &hit = ctx->trace.rl_hit, &miss = ctx->trace.rl_miss;
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
                                       &npend);
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
                    if (single_use && trace_prefetch) {
                        long &hit = ctx->trace.cx_hit, &miss = ctx->trace.cx_miss;
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
                zero_increment_guard(ctx, V1, V2, st.p, 2);       // adi.jl:200-204
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
            double &enq = ctx->trace.ch_enq, &wait = ctx->trace.ch_wait; long &nch = ctx->trace.ch_n, &nit_ = ctx->trace.ch_it;
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
        if (h.collapsed == 2) res.warnings |= 2;          // DRE_WARN_ZERO_INCREMENT: the iteration collapsed (adi.jl:134-137)
        if (h.done || recs.empty()) finished = true;
        if (opt.compression && last_compression >= opt.compression_interval && (!finished || cex)) {
            // Small n: the compression works on the n x n matrix L D L' whatever the number of columns, and the increments
            // never depend on X, so the intermediate compressions of adi.jl:72-76 are deferred to the final one
            // (adi.jl:78-80) as long as the uncompressed factor stays small.
            // The same holds for the direct form up to compress_direct_max_n (one GEMM over all columns) and for the factor form
            // (panel steps ~ rank, GEMM traffic ~ columns: one late compression costs the GEMMs of two early ones and half the
            // panels) while the factor still fits the factor-form limit c + 64 <= n after the next chunk.
            const long rk = Xw->rank(), next = (long)std::max(opt.compression_interval * 2, run.chunk_limit + FAN_GMAX) * k;
            const bool free_form = opt.given_residual && opt.keep_history && !opt.final_compress && ctx->recurrence_wide != 0 && run.hist_ok;
            const bool defer = free_form || (!cex && (n <= 512 ? rk <= 16L * n
                                       : ((n <= ctx->compress_direct_max_n && rk <= 16L * n) ||
                                          (n >= ctx->compress_factor_min_n && rk + next + 64 <= n))));
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
    if (opt.compression && last_compression > 0 && (opt.final_compress || cex)) ldlt_compress(ctx, *Xw, ctf, cex, -1.0, opt.tight_final ? COMPRESS_TIGHT : 0);   // adi.jl:78-80
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
    double &tb = ctx->trace.rec_tb, &ta = ctx->trace.rec_ta, &tf = ctx->trace.rec_tf; long& ns = ctx->trace.rec_ns;
    const auto t0 = std::chrono::steady_clock::now();
    RoctxRange roctx_range("ADI");
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
    // (k = n is allowed as in the reference: the last normalisation divides a vanishing remainder, its column is never used — test/Shifts.jl runs 3 x 3 pencils)
    DRE_REQUIRE(k >= 1 && k <= n, "heuristic shifts: Krylov dimension out of range");
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


}  // namespace dre
