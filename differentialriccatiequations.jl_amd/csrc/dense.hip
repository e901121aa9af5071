// Dense f64 kernels for gfx950: small helpers (copies, norms, traces, read-backs) and the ADI chains of the dense-inverse regime
// (k_adi_fast, k_adi_group).  GEMM family: gemm.hip; Householder QR / TSQR, symmetric eigensolver, band reductions: qr_band.hip.
#include "dense_device.hpp"
#include <functional>
#include "profiling.hpp"
#include <atomic>
#include <chrono>

namespace dre {

// =============================================================================================
// small helpers
// =============================================================================================
__global__ void k_copy(int rows, int cols, const double* __restrict__ src, int lds, double* __restrict__ dst, int ldd,
                       double scale, const AdiState* st) {
    if (st && st->done) return;
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)rows * cols) return;
    int r = idx % rows, c = idx / rows;
    dst[r + (size_t)c * ldd] = scale * src[r + (size_t)c * lds];
}
void copy_mat(Ctx* ctx, const Mat& src, Mat& dst, double scale, const AdiState* st) {
    DRE_REQUIRE(src.rows == dst.rows && src.cols == dst.cols, "copy_mat: shape mismatch");
    size_t tot = (size_t)src.rows * src.cols;
    if (!tot) return;
    TimedScope ts(ctx, "copy", 16.0 * tot, 0);
    hipLaunchKernelGGL(k_copy, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, ctx->stream, src.rows, src.cols, src.p, src.ld, dst.p, dst.ld, scale, st);
}
// Several column blocks copied by ONE launch (horizontal concatenation of the summands of an LDL' object): the descriptors travel
// as kernel arguments, no upload.
struct CopyBatchArgs { CopyDesc d[32]; int n; };
__global__ __launch_bounds__(256) void k_copy_batched(CopyBatchArgs a) {
    const CopyDesc d = a.d[blockIdx.y];
    const size_t tot = (size_t)d.rows * d.cols;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < tot; idx += (size_t)gridDim.x * blockDim.x) {
        const size_t r = idx % d.rows, c = idx / d.rows;
        d.dst[r + c * d.ldd] = d.src[r + c * d.lds];
    }
}
void copy_batched(Ctx* ctx, const std::vector<CopyDesc>& descs) {
    for (size_t b0 = 0; b0 < descs.size(); b0 += 32) {
        CopyBatchArgs a;
        a.n = (int)std::min<size_t>(32, descs.size() - b0);
        size_t mx = 0; double by = 0.0;
        for (int i = 0; i < a.n; ++i) { a.d[i] = descs[b0 + i]; const size_t t = (size_t)a.d[i].rows * a.d[i].cols; mx = std::max(mx, t); by += 16.0 * t; }
        if (mx == 0) continue;
        TimedScope ts(ctx, "copy", by, 0);
        const unsigned gx = (unsigned)std::min<size_t>((mx + 255) / 256, 2048);
        hipLaunchKernelGGL(k_copy_batched, dim3(gx, (unsigned)a.n), dim3(256), 0, ctx->stream, a);
    }
    DRE_HIP(hipGetLastError());
}
__global__ void k_fill(int rows, int cols, double* __restrict__ dst, int ldd, double v, double dv) {
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)rows * cols) return;
    int r = idx % rows, c = idx / rows;
    dst[r + (size_t)c * ldd] = (r == c) ? dv : v;
}
void fill_mat(Ctx* ctx, Mat& dst, double v) {
    size_t tot = (size_t)dst.rows * dst.cols;
    if (!tot) return;
    hipLaunchKernelGGL(k_fill, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, ctx->stream, dst.rows, dst.cols, dst.p, dst.ld, v, v);
}
void set_identity(Ctx* ctx, Mat& dst, double v) {
    size_t tot = (size_t)dst.rows * dst.cols;
    if (!tot) return;
    hipLaunchKernelGGL(k_fill, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, ctx->stream, dst.rows, dst.cols, dst.p, dst.ld, 0.0, v);
}
__global__ void k_transpose(int rows, int cols, const double* __restrict__ src, int lds, double* __restrict__ dst, int ldd) {
    __shared__ double tile[32][33];
    int bx = blockIdx.x * 32, by = blockIdx.y * 32;
    int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 256 threads: ty in 0..7
    for (int j = ty; j < 32; j += 8) {
        int r = bx + tx, c = by + j;
        tile[j][tx] = (r < rows && c < cols) ? src[r + (size_t)c * lds] : 0.0;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        int r = by + tx, c = bx + j;   // dst is cols x rows
        if (r < cols && c < rows) dst[r + (size_t)c * ldd] = tile[tx][j];
    }
}
void transpose_mat(Ctx* ctx, const Mat& src, Mat& dst) {
    DRE_REQUIRE(src.rows == dst.cols && src.cols == dst.rows, "transpose: shape mismatch");
    if (src.empty()) return;
    hipLaunchKernelGGL(k_transpose, dim3(ceil_div(src.rows, 32), ceil_div(src.cols, 32)), dim3(256), 0, ctx->stream, src.rows, src.cols, src.p, src.ld, dst.p, dst.ld);
}
__global__ void k_add_diag(int n, double* __restrict__ dst, int ld, const double* __restrict__ v, double scale) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i + (size_t)i * ld] += scale * (v ? v[i] : 1.0);
}
void add_diag(Ctx* ctx, Mat& dst, const double* v, double scale) {
    int n = std::min(dst.rows, dst.cols);
    if (!n) return;
    hipLaunchKernelGGL(k_add_diag, dim3(ceil_div(n, 256)), dim3(256), 0, ctx->stream, n, dst.p, dst.ld, v, scale);
}
__global__ void k_symmetrize(int n, double* __restrict__ S, int ld) {
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)n * n) return;
    int r = idx % n, c = idx / n;
    if (r > c) {
        double a = S[r + (size_t)c * ld], b = S[c + (size_t)r * ld];
        double m = 0.5 * (a + b);
        S[r + (size_t)c * ld] = m;
        S[c + (size_t)r * ld] = m;
    }
}
// X <- sym(X + A B')  (A, B: n x K): the split-K slabs of A B' are reduced, added and symmetrised in one pass (same arithmetic as
// gemm(beta = 1) followed by symmetrize)
// one workgroup per pair of mirror tiles (16 x 16, one entry of each per thread): both are reduced with eight slab loads in flight and
// exchanged through LDS
__global__ __launch_bounds__(256) void k_reduce_sym_update(int n, int splits, const double* __restrict__ partial, double* __restrict__ X, int ldx) {
    __shared__ double sa[16][17], sb[16][17];
    const int I = blockIdx.x, J = blockIdx.y;
    if (I < J) return;
    const int tr = threadIdx.x & 15, tc = threadIdx.x >> 4;
    const size_t slab = (size_t)n * n;
    auto reduce_at = [&](int r, int c) {
        if (r >= n || c >= n) return 0.0;
        const double* p = partial + r + (size_t)c * n;
        double v = 0.0;
        int z = 0;
        for (; z + 7 < splits; z += 8) {
            double q[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) q[u] = p[(size_t)(z + u) * slab];
#pragma unroll
            for (int u = 0; u < 8; ++u) v += q[u];
        }
        for (; z < splits; ++z) v += p[(size_t)z * slab];
        return v + X[r + (size_t)c * ldx];
    };
    sa[tr][tc] = reduce_at(I * 16 + tr, J * 16 + tc);
    if (I != J) sb[tr][tc] = reduce_at(J * 16 + tr, I * 16 + tc);
    __syncthreads();
    {
        const int r = I * 16 + tr, c = J * 16 + tc;
        if (r < n && c < n) {
            const double a = sa[tr][tc], b = (I == J) ? sa[tc][tr] : sb[tc][tr];
            X[r + (size_t)c * ldx] = (r == c) ? a : 0.5 * (a + b);
        }
    }
    if (I != J) {
        const int r = J * 16 + tr, c = I * 16 + tc;
        if (r < n && c < n) X[r + (size_t)c * ldx] = 0.5 * (sb[tr][tc] + sa[tc][tr]);
    }
}
void gemm_sym_update(Ctx* ctx, const Mat& A, const Mat& B, Mat& X, const char* tag, DevCount dc) {
    const int n = X.rows;
    DRE_REQUIRE(X.cols == n && A.rows == n && B.rows == n && A.cols == B.cols, "gemm_sym_update: shape mismatch");
    if (A.cols == 0) return;
    int splits = 1;
    BufP pb = gemm_partials(ctx, false, true, n, n, A.cols, A.p, A.ld, B.p, B.ld, &splits, nullptr, tag, dc);
    const int nt = ceil_div(n, 16);
    hipLaunchKernelGGL(k_reduce_sym_update, dim3(nt, nt), dim3(256), 0, ctx->stream, n, splits, (const double*)pb->p, X.p, X.ld);
    DRE_HIP(hipGetLastError());
}
void symmetrize(Ctx* ctx, Mat& S) {
    DRE_REQUIRE(S.rows == S.cols, "symmetrize: square matrix expected");
    size_t tot = (size_t)S.rows * S.rows;
    if (!tot) return;
    hipLaunchKernelGGL(k_symmetrize, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, ctx->stream, S.rows, S.p, S.ld);
}
__global__ void k_scale_cols(int rows, int cols, const double* __restrict__ L, int ldl, const double* __restrict__ D, int ldd,
                             double* __restrict__ out, int ldo, double alpha) {
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)rows * cols) return;
    int r = idx % rows, c = idx / rows;
    out[r + (size_t)c * ldo] = alpha * D[c + (size_t)c * ldd] * L[r + (size_t)c * ldl];
}
void scale_cols_by_diag(Ctx* ctx, const Mat& L, const Mat& D, Mat& out, double alpha) {
    size_t tot = (size_t)L.rows * L.cols;
    if (!tot) return;
    hipLaunchKernelGGL(k_scale_cols, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, ctx->stream, L.rows, L.cols, L.p, L.ld, D.p, D.ld, out.p, out.ld, alpha);
}


__global__ __launch_bounds__(1024) void k_frob2(int rows, int cols, const double* __restrict__ A, int ld, double* out) {
    __shared__ double red[17];
    double s = 0.0;
    size_t tot = (size_t)rows * cols;
    for (size_t idx = threadIdx.x; idx < tot; idx += blockDim.x) {
        double v = A[idx % rows + (idx / rows) * (size_t)ld];
        s += v * v;
    }
    s = block_sum(s, red);
    if (threadIdx.x == 0) out[0] = s;
}
static double read_scalar(Ctx* ctx, const double* dev) {
    double h;
    ctx_fetch(ctx, dev, sizeof(double), &h);        // (signal kernel + spin on pinned memory: a fraction of a copy command + stream synchronisation)
    return h;
}
// large operands: partial sums over 64 workgroups, then a fixed-order sum (deterministic)
__global__ __launch_bounds__(256) void k_frob2_parts(int rows, int cols, const double* __restrict__ A, int ld, double* __restrict__ part) {
    __shared__ double red[17];
    double s0 = 0.0, s1 = 0.0;
    const size_t tot = (size_t)rows * cols, stride = (size_t)gridDim.x * blockDim.x;
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; idx + stride < tot; idx += 2 * stride) {
        const size_t i2 = idx + stride;
        const double v = A[idx % rows + (idx / rows) * (size_t)ld], w = A[i2 % rows + (i2 / rows) * (size_t)ld];
        s0 += v * v; s1 += w * w;
    }
    if (idx < tot) { const double v = A[idx % rows + (idx / rows) * (size_t)ld]; s0 += v * v; }
    const double s = block_sum(s0 + s1, red);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
}
__global__ __launch_bounds__(64) void k_frob2_finish(int nparts, const double* __restrict__ part, double* __restrict__ out) {
    double s = 0.0;
    for (int i = threadIdx.x; i < nparts; i += 64) s += part[i];
    s = wave_sum(s);
    if (threadIdx.x == 0) out[0] = s;
}
void frob2_device(Ctx* ctx, const Mat& A, double* out_dev) {
    if ((size_t)A.rows * A.cols <= 65536) {
        hipLaunchKernelGGL(k_frob2, dim3(1), dim3(1024), 0, ctx->stream, A.rows, A.cols, A.p, A.ld, out_dev);
        return;
    }
    DevArr<double> part(ctx, 64);
    hipLaunchKernelGGL(k_frob2_parts, dim3(64), dim3(256), 0, ctx->stream, A.rows, A.cols, (const double*)A.p, A.ld, part.p);
    hipLaunchKernelGGL(k_frob2_finish, dim3(1), dim3(64), 0, ctx->stream, 64, (const double*)part.p, out_dev);
}
struct FetchSrc { const unsigned long long* p[3]; int n[3]; };
__global__ __launch_bounds__(256) void k_fetch(FetchSrc s, unsigned long long* __restrict__ dst, unsigned long long* seq, unsigned long long value) {
    int off = 0;
    for (int a = 0; a < 3; ++a) {
        for (int i = threadIdx.x; i < s.n[a]; i += 256) __hip_atomic_store(dst + off + i, s.p[a][i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        off += s.n[a];
    }
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(seq, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
void ctx_fetch(Ctx* ctx, const void* d0, size_t b0, void* h0, const void* d1, size_t b1, void* h1, const void* d2, size_t b2, void* h2) {
    ctx_fetch_overlap(ctx, std::function<void()>(), d0, b0, h0, d1, b1, h1, d2, b2, h2);
}
void ctx_fetch_overlap(Ctx* ctx, const std::function<void()>& between, const void* d0, size_t b0, void* h0, const void* d1, size_t b1, void* h1,
                       const void* d2, size_t b2, void* h2) {
    const size_t tot = (b0 + b1 + b2) / 8;
    DRE_REQUIRE(b0 % 8 == 0 && b1 % 8 == 0 && b2 % 8 == 0 && tot <= 1024, "ctx_fetch: ranges must be multiples of 8 bytes, 8 KB in all");
    if (ctx->fetch_spin && !ctx->fetch_host) {       // (spinning is neutral on a fast host and saves the wake-up latency of hipStreamSynchronize on a slow one)
        void* hp = nullptr;
        if (hipHostMalloc(&hp, sizeof(Ctx::FetchZone), hipHostMallocMapped) == hipSuccess) {
            void* dp = nullptr;
            if (hipHostGetDevicePointer(&dp, hp, 0) == hipSuccess) {
                ctx->fetch_host = (Ctx::FetchZone*)hp; ctx->fetch_dev = (Ctx::FetchZone*)dp;
                ctx->fetch_host->seq = 0; ctx->fetch_seq = 0;
            } else (void)hipHostFree(hp);
        }
    }
    if (!ctx->fetch_spin || !ctx->fetch_host) {
        if (b0) DRE_HIP(hipMemcpyAsync(h0, d0, b0, hipMemcpyDeviceToHost, ctx->stream));
        if (b1) DRE_HIP(hipMemcpyAsync(h1, d1, b1, hipMemcpyDeviceToHost, ctx->stream));
        if (b2) DRE_HIP(hipMemcpyAsync(h2, d2, b2, hipMemcpyDeviceToHost, ctx->stream));
        if (between) between();
        DRE_HIP(hipStreamSynchronize(ctx->stream));
        return;
    }
    FetchSrc s;
    s.p[0] = (const unsigned long long*)d0; s.n[0] = (int)(b0 / 8);
    s.p[1] = (const unsigned long long*)d1; s.n[1] = (int)(b1 / 8);
    s.p[2] = (const unsigned long long*)d2; s.n[2] = (int)(b2 / 8);
    const unsigned long long want = ++ctx->fetch_seq;
    hipLaunchKernelGGL(k_fetch, dim3(1), dim3(256), 0, ctx->stream, s, (unsigned long long*)ctx->fetch_dev->words, (unsigned long long*)&ctx->fetch_dev->seq, want);
    DRE_HIP(hipGetLastError());
    if (between) between();          // work the host enqueues while the words are on their way (the device runs it right behind the signal kernel)
    // bounded spin on the host-visible sequence number, then the plain synchronisation as a safety net
    const auto t0 = std::chrono::steady_clock::now();
    bool ok = false;
    struct GateWindow { Ctx* c; bool on; GateWindow(Ctx* cc) : c(cc), on(cc->gate && !cc->gate_follow) { if (on) c->gate->waiting.store(1, std::memory_order_relaxed); }
                        ~GateWindow() { if (on) c->gate->waiting.store(0, std::memory_order_relaxed); } } gate_window(ctx);       // the side thread may enqueue while this one waits
    for (long it = 0;; ++it) {
        if (ctx->fetch_host->seq == want) { ok = true; break; }
        if ((it & 1023) == 1023 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 2.0) break;
    }
    if (!ok) {
        DRE_HIP(hipStreamSynchronize(ctx->stream));
        DRE_REQUIRE(ctx->fetch_host->seq == want, "ctx_fetch: the signal kernel did not complete");
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    const unsigned long long* w = (const unsigned long long*)ctx->fetch_host->words;
    if (b0) std::memcpy(h0, w, b0);
    if (b1) std::memcpy(h1, w + b0 / 8, b1);
    if (b2) std::memcpy(h2, w + (b0 + b1) / 8, b2);
}
double frob_norm_host(Ctx* ctx, const Mat& A) {
    if (A.empty()) return 0.0;
    DevArr<double> out(ctx, 1);
    frob2_device(ctx, A, out.p);
    return std::sqrt(read_scalar(ctx, out.p));
}
__global__ __launch_bounds__(256) void k_offdiag_max(int n, const double* __restrict__ D, int ld, double* out) {
    __shared__ double red[17];
    double s = 0.0;
    for (size_t idx = threadIdx.x; idx < (size_t)n * n; idx += blockDim.x) {
        int r = idx % n, c = idx / n;
        if (r != c) s += fabs(D[r + (size_t)c * ld]);
    }
    s = block_sum(s, red);
    if (threadIdx.x == 0) out[0] = s;
}
bool is_diagonal_host(Ctx* ctx, const Mat& D) {
    if (D.empty()) return true;
    DevArr<double> out(ctx, 1);
    hipLaunchKernelGGL(k_offdiag_max, dim3(1), dim3(256), 0, ctx->stream, D.rows, D.p, D.ld, out.p);
    return read_scalar(ctx, out.p) == 0.0;
}

// nrm^2 = sum_ij (T G)_ij (T G)_ji ; single block, k <= a few hundred
__global__ __launch_bounds__(1024) void k_ldlt_norm(int k, const double* __restrict__ G, int ldg, const double* __restrict__ T, int ldt,
                                                    int tdiag, double alpha, AdiState* st, int iters_after, double* out) {
    __shared__ double red[17];
    if (st && st->done) return;
    double s = 0.0;
    if (tdiag) {
        for (size_t idx = threadIdx.x; idx < (size_t)k * k; idx += blockDim.x) {
            int i = idx % k, j = idx / k;
            double g = G[i + (size_t)j * ldg];
            s += T[i + (size_t)i * ldt] * T[j + (size_t)j * ldt] * g * g;
        }
    } else {
        // (TG)_ij = sum_l T_il G_lj ; (TG)_ji = sum_l T_jl G_li
        for (size_t idx = threadIdx.x; idx < (size_t)k * k; idx += blockDim.x) {
            int i = idx % k, j = idx / k;
            double a = 0.0, b = 0.0;
            for (int l = 0; l < k; ++l) {
                a += T[i + (size_t)l * ldt] * G[l + (size_t)j * ldg];
                b += T[j + (size_t)l * ldt] * G[l + (size_t)i * ldg];
            }
            s += a * b;
        }
    }
    s = block_sum(s, red);
    if (threadIdx.x == 0) {
        double nrm = fabs(alpha) * sqrt(fmax(s, 0.0));
        if (out) out[0] = nrm;
        if (st) {
            st->res_norm = nrm;
            st->iters = iters_after;
            st->norms[iters_after & 511] = nrm;          // ring: the host reads every chunk (< 512 iterations) before it wraps
            if (nrm <= st->abstol || iters_after >= st->maxiters) st->done = 1;
        }
    }
}
// s = sum_ij M_ij M_ji  for M = T G given explicitly
__global__ __launch_bounds__(1024) void k_trace_sq(int k, const double* __restrict__ M, int ldm, double alpha, AdiState* st, int iters_after, double* out) {
    __shared__ double red[17];
    if (st && st->done) return;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    double s = 0.0;
    for (int c = wave; c < k; c += nw)
        for (int r = lane; r < k; r += 64) s += M[r + (size_t)c * ldm] * M[c + (size_t)r * ldm];
    s = block_sum(s, red);
    if (threadIdx.x == 0) {
        double nrm = fabs(alpha) * sqrt(fmax(s, 0.0));
        if (out) out[0] = nrm;
        if (st) {
            st->res_norm = nrm;
            st->iters = iters_after;
            st->norms[iters_after & 511] = nrm;          // ring: the host reads every chunk (< 512 iterations) before it wraps
            if (nrm <= st->abstol || iters_after >= st->maxiters) st->done = 1;
        }
    }
}
// g matrices M_j = T G_jj (k x k each, stored side by side: M_j at columns j k of TGall) in iteration order, with the decisions of adi.jl:115-123:
// the first residual at or below abstol ends the loop (fan groups, residual wider than 96 columns).  One workgroup per matrix: trace(M_j^2) =
// sum over 32 x 32 tile pairs (bi <= bj) of <M[bi, bj], M[bj, bi]'>, both tiles read coalesced and met through LDS (the single workgroup that
// walked all g matrices with a strided second operand took 86 us for g = 8, k = 200: 4.4 % of a Ros2 run at n = 5177); the norms meet in
// st->gnorm and the last arrival takes the decisions in iteration order (the meeting point of k_gram_norm_z).
__global__ __launch_bounds__(512) void k_trace_sq_multi(int k, int g, const double* __restrict__ TGall, int ldm, double alpha, AdiState* st, int iters0) {
    __shared__ double A[2][32][33], B[2][32][33];
    __shared__ double red[17];
    if (st->done) return;
    const int j = blockIdx.x;
    const double* __restrict__ M = TGall + (size_t)j * k * ldm;
    const int q = threadIdx.x >> 8, t = threadIdx.x & 255, tx = t & 31, ty = t >> 5;      // two tile pairs in flight, 32 x 8 threads each
    const int nt = (k + 31) / 32, npair = nt * (nt + 1) / 2;
    double s = 0.0;
    for (int p0 = 0; p0 < npair; p0 += 2) {
        const int p = p0 + q;
        int bi = 0, rem = p;
        if (p < npair) { while (rem >= nt - bi) { rem -= nt - bi; ++bi; } }
        const int bj = bi + rem;
        if (p < npair) {
            for (int c = ty; c < 32; c += 8) {
                const int ra = bi * 32 + tx, ca = bj * 32 + c;           // A = M[bi rows, bj cols]
                A[q][c][tx] = (ra < k && ca < k) ? M[ra + (size_t)ca * ldm] : 0.0;
                const int rb = bj * 32 + tx, cb = bi * 32 + c;           // B = M[bj rows, bi cols]
                B[q][c][tx] = (rb < k && cb < k) ? M[rb + (size_t)cb * ldm] : 0.0;
            }
        }
        __syncthreads();
        if (p < npair) {
            double a = 0.0;
            for (int c = ty; c < 32; c += 8) a += A[q][c][tx] * B[q][tx][c];     // M[bi*32 + tx, bj*32 + c] * M[bj*32 + c, bi*32 + tx]
            s += (bi == bj) ? a : 2.0 * a;
        }
        __syncthreads();
    }
    s = block_sum(s, red);
    if (threadIdx.x != 0) return;
    __hip_atomic_store(&st->gnorm[j], fabs(alpha) * sqrt(fmax(s, 0.0)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __threadfence();
    if (atomicAdd(&st->ticket, 1) != g - 1) return;
    __threadfence();
    st->ticket = 0;
    for (int i = 0; i < g; ++i) {
        const double nrm = __hip_atomic_load(&st->gnorm[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int iters_after = iters0 + i + 1;
        st->res_norm = nrm;
        st->iters = iters_after;
        st->norms[iters_after & 511] = nrm;
        if (nrm <= st->abstol || iters_after >= st->maxiters) { st->done = 1; break; }
    }
}
// The same for large k in two launches: 32 x 32 tile pairs (bi <= bj) through LDS, so that both M_ij and M_ji are read coalesced,
// one partial sum per pair; then the fixed-order sum and the decision.
__global__ __launch_bounds__(256) void k_trace_sq_tiles(int k, const double* __restrict__ M, int ldm, double* __restrict__ part, const AdiState* st) {
    if (st && st->done) return;
    __shared__ double A[32][33], B[32][33];
    __shared__ double red[17];
    // tile pair index -> (bi, bj), bi <= bj
    const int nt = (k + 31) / 32;
    int bi = 0, rem = blockIdx.x;
    while (rem >= nt - bi) { rem -= nt - bi; ++bi; }
    const int bj = bi + rem;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 32 x 8
    for (int c = ty; c < 32; c += 8) {
        const int ra = bi * 32 + tx, ca = bj * 32 + c;           // A = M[bi-block rows, bj-block cols]
        A[c][tx] = (ra < k && ca < k) ? M[ra + (size_t)ca * ldm] : 0.0;
        const int rb = bj * 32 + tx, cb = bi * 32 + c;           // B = M[bj-block rows, bi-block cols]
        B[c][tx] = (rb < k && cb < k) ? M[rb + (size_t)cb * ldm] : 0.0;
    }
    __syncthreads();
    double s = 0.0;
    for (int c = ty; c < 32; c += 8) s += A[c][tx] * B[tx][c];   // M[i, j] * M[j, i]
    s = block_sum(s, red);
    if (threadIdx.x == 0) part[blockIdx.x] = (bi == bj) ? s : 2.0 * s;
}
__global__ __launch_bounds__(64) void k_trace_finish(int nparts, const double* __restrict__ part, double alpha, AdiState* st, int iters_after, double* out) {
    if (st && st->done) return;
    double s = 0.0;
    for (int i = threadIdx.x; i < nparts; i += 64) s += part[i];
    s = wave_sum(s);
    if (threadIdx.x == 0) {
        double nrm = fabs(alpha) * sqrt(fmax(s, 0.0));
        if (out) out[0] = nrm;
        if (st) {
            st->res_norm = nrm;
            st->iters = iters_after;
            st->norms[iters_after & 511] = nrm;          // ring: the host reads every chunk (< 512 iterations) before it wraps
            if (nrm <= st->abstol || iters_after >= st->maxiters) st->done = 1;
        }
    }
}
static void trace_sq(Ctx* ctx, const Mat& TG, double alpha, AdiState* st, int iters_after, double* out) {
    const int k = TG.rows;
    if (k <= 128) {
        hipLaunchKernelGGL(k_trace_sq, dim3(1), dim3(1024), 0, ctx->stream, k, TG.p, TG.ld, alpha, st, iters_after, out);
        return;
    }
    const int nt = (k + 31) / 32, np = nt * (nt + 1) / 2;
    DevArr<double> part(ctx, (size_t)np);
    hipLaunchKernelGGL(k_trace_sq_tiles, dim3(np), dim3(256), 0, ctx->stream, k, (const double*)TG.p, TG.ld, part.p, (const AdiState*)st);
    hipLaunchKernelGGL(k_trace_finish, dim3(1), dim3(64), 0, ctx->stream, np, (const double*)part.p, alpha, st, iters_after, out);
}
void ldlt_norm_update_state(Ctx* ctx, const Mat& G, const Mat& T, bool tdiag, double alpha, AdiState* st, int iters_after) {
    if (tdiag) {
        TimedScope ts(ctx, "ldlt_norm", 16.0 * G.rows * G.cols, 4.0 * G.rows * G.cols);
        hipLaunchKernelGGL(k_ldlt_norm, dim3(1), dim3(1024), 0, ctx->stream, G.rows, G.p, G.ld, T.p, T.ld, 1, alpha, st, iters_after, (double*)nullptr);
    } else {
        Mat TG(ctx, G.rows, G.cols);
        gemm(ctx, false, false, 1.0, T, G, 0.0, TG, st, "gemm_norm");
        TimedScope ts(ctx, "ldlt_norm", 16.0 * G.rows * G.cols, 4.0 * G.rows * G.cols);
        trace_sq(ctx, TG, alpha, st, iters_after, nullptr);
    }
}
// Workgroup-wide: G = sum of `splits` k x k slabs (fixed order), M = T G (or diag(T) G), nrm = |alpha| sqrt(sum_ij M_ij M_ji),
// then the convergence decision of adi.jl:115-123 on the device.  G and T live in LDS (gsm: 2 kp^2 doubles, kp = k rounded
// up to 32, k <= 96).  Dense T: M = T G and N = G T' (= M') are formed 32 x 32 block-wise on the matrix cores — both in the
// same lane layout, so tr(M M) = sum_ij M_ij N_ij needs no transposition; only blocks bi <= bj are computed.
template <bool COHERENT>
__device__ __forceinline__ void gram_norm_body(int k, int splits, const double* part, const double* __restrict__ T, int ldt,
                                               int tdiag, double alpha, AdiState* st, int iters_after, double* gsm, double* red, int ldp = 0, size_t slab = 0,
                                               double* nrm_out = nullptr) {
    const int kp = (k + 31) & ~31, ld = kp;
    if (ldp == 0) { ldp = k; slab = (size_t)k * k; }       // default: dense k x k slabs
    double* G = gsm;                        // kp x kp
    double* Ts = gsm + (size_t)kp * kp;     // kp x kp (or k diagonal entries)
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, nw = blockDim.x >> 6;
    for (int idx = tid; idx < kp * kp; idx += blockDim.x) {
        const int r = idx % kp, c = idx / kp;
        const bool in = r < k && c < k;
        double s = 0.0;
        if (in) {
            // fixed summation order; up to four slab loads in flight
            const double* q = part + r + (size_t)c * ldp;
            const size_t sl = slab;
            int z = 0;
            if (!COHERENT)
                for (; z + 3 < splits; z += 4) {
                    const double p0 = q[z * sl], p1 = q[(z + 1) * sl], p2 = q[(z + 2) * sl], p3 = q[(z + 3) * sl];
                    s = (((s + p0) + p1) + p2) + p3;
                }
            for (; z < splits; ++z)   // COHERENT: slabs written by other workgroups of the same launch are read past the L1
                s += COHERENT ? __hip_atomic_load(q + z * sl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : q[z * sl];
        }
        G[idx] = s;
        if (!tdiag) Ts[idx] = in ? T[r + (size_t)c * ldt] : 0.0;
    }
    if (tdiag) for (int i = tid; i < k; i += blockDim.x) Ts[i] = T[i + (size_t)i * ldt];
    __syncthreads();
    double s = 0.0;
    if (tdiag) {
        for (int c = wave; c < k; c += nw)
            for (int r = lane; r < k; r += 64) { const double g = G[r + c * ld]; s += Ts[r] * Ts[c] * g * g; }
    } else {
        const int nbk = kp / 32, lr = lane & 15, lk = lane >> 4;
        int p = 0;
        for (int bi = 0; bi < nbk; ++bi)
            for (int bj = bi; bj < nbk; ++bj, ++p) {
                if (p % nw != wave) continue;
                v4d m[2][2], nn[2][2];
#pragma unroll
                for (int x = 0; x < 2; ++x)
#pragma unroll
                    for (int y = 0; y < 2; ++y) { m[x][y] = (v4d){0.0, 0.0, 0.0, 0.0}; nn[x][y] = (v4d){0.0, 0.0, 0.0, 0.0}; }
                const double* ti = Ts + bi * 32 + lr; const double* tj = Ts + bj * 32 + lr;
                const double* gi = G + bi * 32 + lr;  const double* gj = G + bj * 32 + lr;
                for (int kk = 0; kk < kp / 4; ++kk) {
                    const size_t off = (size_t)(kk * 4 + lk) * ld;
                    const double ta0 = ti[off], ta1 = ti[off + 16], gb0 = gj[off], gb1 = gj[off + 16];
                    const double ga0 = gi[off], ga1 = gi[off + 16], tb0 = tj[off], tb1 = tj[off + 16];
                    m[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(ta0, gb0, m[0][0], 0, 0, 0);
                    m[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(ta0, gb1, m[0][1], 0, 0, 0);
                    m[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(ta1, gb0, m[1][0], 0, 0, 0);
                    m[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(ta1, gb1, m[1][1], 0, 0, 0);
                    nn[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(ga0, tb0, nn[0][0], 0, 0, 0);
                    nn[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(ga0, tb1, nn[0][1], 0, 0, 0);
                    nn[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(ga1, tb0, nn[1][0], 0, 0, 0);
                    nn[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(ga1, tb1, nn[1][1], 0, 0, 0);
                }
                double sum = 0.0;
#pragma unroll
                for (int x = 0; x < 2; ++x)
#pragma unroll
                    for (int y = 0; y < 2; ++y)
#pragma unroll
                        for (int r = 0; r < 4; ++r) sum += m[x][y][r] * nn[x][y][r];
                s += (bi == bj) ? sum : 2.0 * sum;
            }
    }
    s = block_sum(s, red);
    if (tid == 0) {
        const double nrm = fabs(alpha) * sqrt(fmax(s, 0.0));
        if (nrm_out) { *nrm_out = nrm; return; }          // (the caller decides: fan groups take their g norms in iteration order)
        st->res_norm = nrm;
        st->iters = iters_after;
        st->norms[iters_after & 511] = nrm;          // ring: the host reads every chunk (< 512 iterations) before it wraps
        if (nrm <= st->abstol || iters_after >= st->maxiters) st->done = 1;
    }
}
// Fan groups, round 4: the norms of the g residuals of a group from their g Gram blocks Gd = [G_11 .. G_gg] (k x g k) in g workgroups side by side;
// the LAST one to arrive (ticket in the control block) takes the decisions of adi.jl:115-123 in iteration order — the first residual at or
// below abstol ends the loop.  (Round 3: one workgroup, the g norms one after the other: 42 us at g = 5, k = 64.)
__global__ __launch_bounds__(1024) void k_gram_norm_z(int k, int g, const double* __restrict__ Gd, const double* __restrict__ T, int ldt, int tdiag, double alpha,
                                                      AdiState* st, int iters0) {
    if (st->done) return;
    extern __shared__ double gsm[];
    __shared__ double red[17];
    const int j = blockIdx.x;
    gram_norm_body<false>(k, 1, Gd + (size_t)j * k * k, T, ldt, tdiag, alpha, st, 0, gsm, red, k, 0, &st->gnorm[j]);
    if (threadIdx.x != 0) return;
    __threadfence();
    if (atomicAdd(&st->ticket, 1) != g - 1) return;
    __threadfence();
    st->ticket = 0;
    for (int i = 0; i < g; ++i) {
        const double nrm = __hip_atomic_load(&st->gnorm[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int iters_after = iters0 + i + 1;
        st->res_norm = nrm;
        st->iters = iters_after;
        st->norms[iters_after & 511] = nrm;
        if (nrm <= st->abstol || iters_after >= st->maxiters) { st->done = 1; break; }
    }
}
__global__ __launch_bounds__(1024) void k_gram_norm(int k, int splits, const double* __restrict__ part, const double* __restrict__ T, int ldt,
                                                    int tdiag, double alpha, AdiState* st, int iters_after) {
    if (st->done) return;
    extern __shared__ double gsm[];
    __shared__ double red[17];
    gram_norm_body<false>(k, splits, part, T, ldt, tdiag, alpha, st, iters_after, gsm, red);
}

// The norms and decisions of g consecutive iterations in ONE launch (fan groups, engine.hip): Gall is the (g k) x (g k) Gram matrix of
// [R_1 .. R_g]; its diagonal blocks are taken in iteration order, and the first residual at or below abstol ends the loop (adi.jl:115-123).
__global__ __launch_bounds__(1024) void k_gram_norm_multi(int k, int g, const double* __restrict__ Gall, int ldg, const double* __restrict__ T, int ldt,
                                                          int tdiag, double alpha, AdiState* st, int iters0) {
    if (st->done) return;
    extern __shared__ double gsm[];
    __shared__ double red[17];
    __shared__ int stop;
    for (int j = 0; j < g; ++j) {
        gram_norm_body<false>(k, 1, Gall + (size_t)j * k + (size_t)j * k * ldg, T, ldt, tdiag, alpha, st, iters0 + j + 1, gsm, red, ldg, 0);
        if (threadIdx.x == 0) stop = __hip_atomic_load(&st->done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (stop) return;
        __syncthreads();
    }
}

// Dense-inverse ADI step for a real shift: everything after the stacked GEMM except the final norm reduction (k <= 96):
//   Wpart: split-K slabs of [inv; E' inv; U' inv] * R  ((2n + m) x k each);  WKS = [inv Vt; E' inv Vt] * Sinv  (2n x m)
//   V = W - WKS_top small,   R <- R - 2 mu (EW - WKS_mid small)                 (adi.jl:166-171, LowRankUpdate.jl:29-39)
//   Gpart[blockIdx] = R_new(rows of this workgroup)' R_new(rows)                 (Gram slabs for the residual norm)
// One workgroup per 64 rows (grid-stride over row chunks).
template <bool HAS_LR>
__global__ __launch_bounds__(1024) void k_dense_step(int n, int m, int k, int splits, const double* __restrict__ Wpart,
                                                     const double* __restrict__ WKS, int ldwk, double* __restrict__ V, int ldv,
                                                     double* __restrict__ R, int ldr, double two_mu, double* __restrict__ Gpart,
                                                     AdiState* st, int nblk_main, const double* __restrict__ Gprev, int nblk_prev,
                                                     const double* __restrict__ Tn, int ldtn, int tdiag, double alpha_n, int iters_prev) {
    if (st->done) return;
    extern __shared__ double dsm[];
    if ((int)blockIdx.x >= nblk_main) {
        // rider: the residual norm and convergence decision of the PREVIOUS iteration (its Gram slabs are complete since the last launch)
        // run beside this iteration's step instead of in a launch of their own; a positive decision stops the loop one launch later
        __shared__ double nred[17];
        gram_norm_body<false>(k, nblk_prev, Gprev, Tn, ldtn, tdiag, alpha_n, st, iters_prev, dsm, nred);
        return;
    }
    const int M = 2 * n + m;
    const size_t slab = (size_t)M * k;
    const int kp16 = (k + 15) & ~15, ldk = kp16 + 1;
    double* Rt = dsm;                                  // Rt[c + il * ldk] = R_new(i0 + il, c)
    double* small = Rt + (size_t)64 * ldk;             // m x k
    double* WKs = small + (size_t)(HAS_LR ? m : 0) * k;   // 64 x m
    double* EWKs = WKs + (size_t)64 * (HAS_LR ? m : 0);   // 64 x m
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, nt = blockDim.x, nw = nt >> 6;
    const int nt16 = kp16 / 16, lr = lane & 15, lk = lane >> 4;
    // Gram tiles owned by this wave (at most 3 for k <= 96 with 16 waves), accumulated over all row chunks of the workgroup
    v4d acc[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) acc[q] = (v4d){0.0, 0.0, 0.0, 0.0};
    if (HAS_LR) {
        for (int id = tid; id < m * k; id += nt) {
            const int j = id % m, c = id / m;
            double sv = 0.0;
            const double* wp = Wpart + (size_t)(2 * n + j) + (size_t)c * M;
            int z = 0;
            for (; z + 3 < splits; z += 4) {
                const double p0 = wp[z * slab], p1 = wp[(z + 1) * slab], p2 = wp[(z + 2) * slab], p3 = wp[(z + 3) * slab];
                sv = (((sv + p0) + p1) + p2) + p3;
            }
            for (; z < splits; ++z) sv += wp[z * slab];
            small[id] = sv;
        }
    }
    for (int i0 = blockIdx.x * 64; i0 < n; i0 += nblk_main * 64) {
        if (HAS_LR) {
            for (int id = tid; id < 64 * m; id += nt) {
                const int il = id & 63, j = id >> 6, i = i0 + il;
                WKs[id] = i < n ? WKS[i + (size_t)j * ldwk] : 0.0;
                EWKs[id] = i < n ? WKS[n + i + (size_t)j * ldwk] : 0.0;
            }
        }
        __syncthreads();          // small / WKs ready; previous chunk's Rt consumed
        {
            // every thread owns row i and the columns wave, wave + 16, ...: the slab loads of a split are independent
            // (three columns at a time keeps the kernel inside 128 registers without spills)
            const int il = lane, i = i0 + il;
#pragma unroll 1
            for (int qb = 0; qb < 6; qb += 3) {
                if (wave + qb * 16 >= kp16) break;
                double w[3], ew[3];
#pragma unroll
                for (int q = 0; q < 3; ++q) { w[q] = 0.0; ew[q] = 0.0; }
#pragma unroll 1
                for (int z = 0; z < splits; ++z) {
                    const double* wp = Wpart + z * slab + i;
#pragma unroll
                    for (int q = 0; q < 3; ++q) {
                        const int c = wave + (qb + q) * 16;
                        if (c < k && i < n) { w[q] += wp[(size_t)c * M]; ew[q] += wp[(size_t)c * M + n]; }
                    }
                }
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const int c = wave + (qb + q) * 16;
                    if (c >= kp16) continue;
                    double rn = 0.0;
                    if (c < k && i < n) {
                        double wv = w[q], ev = ew[q];
                        if (HAS_LR)
                            for (int j = 0; j < m; ++j) {
                                const double sj = small[j + c * m];
                                wv -= WKs[il + j * 64] * sj;
                                ev -= EWKs[il + j * 64] * sj;
                            }
                        V[i + (size_t)c * ldv] = wv;
                        double* rp = R + i + (size_t)c * ldr;
                        rn = *rp - two_mu * ev;
                        *rp = rn;
                    }
                    Rt[c + il * ldk] = rn;
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int t = wave + q * nw;
            if (t < nt16 * nt16) {
                const int ti = t % nt16, tj = t / nt16;
                const double* pa = Rt + ti * 16 + lr; const double* pb = Rt + tj * 16 + lr;
#pragma unroll 4
                for (int l0 = 0; l0 < 64; l0 += 4) {
                    const double a = pa[(l0 + lk) * ldk], bb = pb[(l0 + lk) * ldk];
                    acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bb, acc[q], 0, 0, 0);
                }
            }
        }
    }
    {
        double* gp = Gpart + (size_t)blockIdx.x * k * k;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int t = wave + q * nw;
            if (t < nt16 * nt16) {
                const int ti = t % nt16, tj = t / nt16;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = ti * 16 + (lane >> 4) + 4 * r, col = tj * 16 + (lane & 15);
                    if (row < k && col < k) gp[row + (size_t)col * k] = acc[q][r];
                }
            }
        }
    }
}

static void launch_gram_norm(Ctx* ctx, int k, int nblk, const double* gpart, const Mat& T, bool tdiag, double alpha, AdiState* st, int iters_after) {
    const int kp32 = (k + 31) & ~31;
    TimedScope ts(ctx, "ldlt_norm", 8.0 * nblk * k * k, 4.0 * (double)k * k * k);
    const size_t shm = 2 * (size_t)kp32 * kp32 * sizeof(double);
    lds_attr(ctx, (const void*)k_gram_norm, 150 * 1024);
    hipLaunchKernelGGL(k_gram_norm, dim3(1), dim3(1024), shm, ctx->stream, k, nblk, gpart, T.p, T.ld, tdiag ? 1 : 0, alpha, st, iters_after);
}
void dense_norm_flush(Ctx* ctx, int k, const Mat& T, bool tdiag, double alpha, AdiState* st, DenseNormPending* pend) {
    if (!pend || !pend->valid) return;
    launch_gram_norm(ctx, k, pend->nblk, (const double*)pend->gpart->p, T, tdiag, alpha, st, pend->iters_after);
    pend->valid = false;
    DRE_HIP(hipGetLastError());
}
// pend != nullptr: the norm of THIS iteration is not launched; it rides on the step kernel of the next iteration (or dense_norm_flush),
// and the pending norm of the previous iteration rides on this step.
void dense_adi_step(Ctx* ctx, int n, int m, int k, int splits, const double* Wpart, const double* WKS, int ldwk, Mat& V, Mat& R,
                    double two_mu, const Mat& T, bool tdiag, double alpha, AdiState* st, int iters_after, DenseNormPending* pend) {
    DRE_REQUIRE(k <= 96 && m <= 32, "dense_adi_step: k <= 96 and m <= 32 expected");
    const int nblk = ceil_div(n, 64), kp16 = (k + 15) & ~15, kp32 = (k + 31) & ~31;
    auto gpart = std::make_shared<Buf>(ctx, (size_t)nblk * k * k * sizeof(double));
    const bool ride = pend && pend->valid;
    const size_t step_lds = ((size_t)64 * (kp16 + 1) + (size_t)m * k + 2 * (size_t)64 * m) * sizeof(double);
    const size_t lds = ride ? std::max(step_lds, 2 * (size_t)kp32 * kp32 * sizeof(double)) : step_lds;
    lds_attr(ctx, (const void*)k_dense_step<true>, 150 * 1024); lds_attr(ctx, (const void*)k_dense_step<false>, 150 * 1024);
    {
        TimedScope ts(ctx, "dense_step", 8.0 * (2.0 * n * k * splits + 3.0 * n * k + 2.0 * n * m + (double)nblk * k * k), 4.0 * n * k * m + 2.0 * n * (double)k * k);
        const double* gprev = ride ? (const double*)pend->gpart->p : nullptr;
        const int nprev = ride ? pend->nblk : 0, iprev = ride ? pend->iters_after : 0;
        if (m > 0)
            hipLaunchKernelGGL((k_dense_step<true>), dim3(nblk + (ride ? 1 : 0)), dim3(1024), lds, ctx->stream, n, m, k, splits, Wpart, WKS, ldwk, V.p, V.ld, R.p, R.ld,
                               two_mu, (double*)gpart->p, st, nblk, gprev, nprev, (const double*)T.p, T.ld, tdiag ? 1 : 0, alpha, iprev);
        else
            hipLaunchKernelGGL((k_dense_step<false>), dim3(nblk + (ride ? 1 : 0)), dim3(1024), lds, ctx->stream, n, 0, k, splits, Wpart, (const double*)nullptr, 0, V.p, V.ld,
                               R.p, R.ld, two_mu, (double*)gpart->p, st, nblk, gprev, nprev, (const double*)T.p, T.ld, tdiag ? 1 : 0, alpha, iprev);
    }
    if (pend) {
        pend->gpart = gpart; pend->nblk = nblk; pend->iters_after = iters_after; pend->valid = true;
    } else {
        // the Gram slabs of the workgroups are summed (fixed order) by the norm kernel; a hand-over inside one launch would
        // need agent-scope fences, which cost more than a launch on a multi-XCD part
        launch_gram_norm(ctx, k, nblk, (const double*)gpart->p, T, tdiag, alpha, st, iters_after);
    }
    DRE_HIP(hipGetLastError());
}

// =============================================================================================
// Fast ADI chain (dense-inverse regime, real Cyclic shifts whose stacked inverses persist across the Lyapunov solves of a run).
// Once per Lyapunov solve and shift the rank-m Sherman-Morrison-Woodbury correction (smw.jl:20-43) is folded into the
// stacked inverse,
//     Seff = [inv; E' inv] - ([inv; E' inv] Vt Sinv) (U' inv)                      (2n x n),
// so that an ADI iteration (adi.jl:149-179) is ONE launch of pure matrix-core work without split-K slabs or a separate
// apply pass:   V = Seff_top R,   R_next = R - 2 mu Seff_bot R.
// Seff is stored in the lane order of the MFMA A operand ("packed": strip of 16 rows x K-step of 4 columns = 64 consecutive
// doubles), so every A fragment is one fully coalesced 512-byte load.  One workgroup owns a 16-row strip of V or of R_next,
// one wave per 16-column tile, K = n in registers (no LDS staging: the B fragments come straight from L2).
// The residual norm (LDLt.jl:77-89 in Gram form) is pipelined over the following two launches: the strip workgroups leave
// per-strip Gram slabs of R_next, rider workgroups of the next launch sum them (fixed order), and one rider of the launch
// after that forms tr((T G)^2) and takes the convergence decision of adi.jl:115-123.  A positive decision therefore arrives
// two launches late; the speculative iterations are discarded by the host exactly like every other speculatively enqueued one.
// =============================================================================================
struct EffStackBatch { const double* stack[16]; const double* WKS[16]; double* out[16]; };
__global__ __launch_bounds__(256) void k_eff_stack(int n, int m, int nstrip, int kst, int lds_, int ldwk, EffStackBatch bt) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= kst * 64) return;
    const int t = idx >> 6, lane = idx & 63;
    const int b = blockIdx.y, half = b / nstrip, s = b - half * nstrip;
    const int row = s * 16 + (lane & 15), col = 4 * t + (lane >> 4);
    const double* __restrict__ stack = bt.stack[blockIdx.z];
    const double* __restrict__ WKS = bt.WKS[blockIdx.z];
    double v = 0.0;
    if (row < n && col < n) {
        const size_t r = (size_t)half * n + row;
        v = stack[r + (size_t)col * lds_];
        if (WKS) {
            const double* ui = stack + 2 * (size_t)n + (size_t)col * lds_;     // (U' inv)(:, col)
            double a0 = 0.0, a1 = 0.0;
            int l = 0;
            for (; l + 1 < m; l += 2) { a0 += WKS[r + (size_t)l * ldwk] * ui[l]; a1 += WKS[r + (size_t)(l + 1) * ldwk] * ui[l + 1]; }
            if (l < m) a0 += WKS[r + (size_t)l * ldwk] * ui[l];
            v -= a0 + a1;
        }
    }
    bt.out[blockIdx.z][((size_t)b * kst + t) * 64 + lane] = v;
}
// The same on the matrix cores (m <= 8): a wave owns 16 x 16 tiles (16 rows of a strip x 4 K-steps of the packed layout).  The correction
// is computed TRANSPOSED,  D = (U' inv)(:, cols)' (WKS(rows, :))'  (16 x m times m x 16: two v_mfma_f64_16x16x4), because the accumulator
// layout of D — lane = 16 lk + lr holds D[lk + 4 r][lr] = (col 4 r + lk, row lr) — IS the packed layout: acc[r] is the entry of K-step
// t0 + r at position `lane`.  Two loads per output entry instead of fifteen (the scalar kernel re-reads the m entries of WKS and of U' inv
// for every entry and is bound by the vector-memory instruction rate: 317 us per step for ten shifts at n = 1357, 590 MB).
#define EFF_TPW 4
__global__ __launch_bounds__(256) void k_eff_stack_mfma(int n, int m, int nstrip, int kst, int lds_, int ldwk, EffStackBatch bt) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, lr = lane & 15, lk = lane >> 4;
    const int b = blockIdx.y, half = b / nstrip, s = b - half * nstrip;
    const double* __restrict__ stack = bt.stack[blockIdx.z];
    const double* __restrict__ WKS = bt.WKS[blockIdx.z];
    double* __restrict__ out = bt.out[blockIdx.z];
    const int ntile = (kst + 3) >> 2;
    const int row = s * 16 + lr;
    const size_t r = (size_t)half * n + min(row, n - 1);
    // B operand (shared by all tiles of the strip): B[k][j = row] = WKS(row, k)
    double bw[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const int k = 4 * kk + lk;
        const double v = WKS[r + (size_t)min(k, m - 1) * ldwk];
        bw[kk] = (k < m && row < n) ? v : 0.0;
    }
#pragma unroll
    for (int i = 0; i < EFF_TPW; ++i) {
        const int tt = (blockIdx.x * EFF_TPW + i) * 4 + wave;           // tile = K-steps 4 tt .. 4 tt + 3 = columns 16 tt .. 16 tt + 15
        if (tt >= ntile) break;                                          // wave-uniform
        // A operand: A[i = col][k] = (U' inv)(k, col), col = 16 tt + lr
        const int colA = 16 * tt + lr;
        double au[2], sv[4];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int k = 4 * kk + lk;
            const double v = stack[2 * (size_t)n + min(k, m - 1) + (size_t)min(colA, n - 1) * lds_];
            au[kk] = (k < m && colA < n) ? v : 0.0;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int col = 16 * tt + 4 * q + lk;
            const double v = stack[r + (size_t)min(col, n - 1) * lds_];
            sv[q] = (row < n && col < n) ? v : 0.0;
        }
        v4d acc = (v4d){0.0, 0.0, 0.0, 0.0};
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(au[0], bw[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(au[1], bw[1], acc, 0, 0, 0);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int t = 4 * tt + q;
            if (t < kst) out[((size_t)b * kst + t) * 64 + lane] = sv[q] - acc[q];
        }
    }
}
void adi_fast_build(Ctx* ctx, int n, int m, const std::vector<const double*>& stacks, int lds_, const std::vector<const double*>& wks, int ldwk,
                    const std::vector<double*>& outs) {
    const int nstrip = adi_fast_nstrip(n), kst = adi_fast_kst(n);
    bool all_lr = m >= 1 && m <= 8;
    for (auto w : wks) if (!w) all_lr = false;
    if (all_lr) {
        const int ntile = (kst + 3) >> 2;
        for (size_t b0 = 0; b0 < stacks.size(); b0 += 16) {
            EffStackBatch bt;
            const int nb = (int)std::min<size_t>(16, stacks.size() - b0);
            for (int i = 0; i < 16; ++i) { const int j = i < nb ? i : 0; bt.stack[i] = stacks[b0 + j]; bt.WKS[i] = wks[b0 + j]; bt.out[i] = outs[b0 + j]; }
            TimedScope ts(ctx, "adi_eff_stack", 8.0 * nb * (2.0 * n * n + (double)m * n + 2.0 * n * m + 2.0 * nstrip * 16.0 * kst * 4.0), 4.0 * nb * n * n * (double)m);
            hipLaunchKernelGGL(k_eff_stack_mfma, dim3(ceil_div(ntile, 4 * EFF_TPW), 2 * nstrip, nb), dim3(256), 0, ctx->stream, n, m, nstrip, kst, lds_, ldwk, bt);
        }
        DRE_HIP(hipGetLastError());
        return;
    }
    for (size_t b0 = 0; b0 < stacks.size(); b0 += 16) {
        EffStackBatch bt;
        const int nb = (int)std::min<size_t>(16, stacks.size() - b0);
        for (int i = 0; i < 16; ++i) { const int j = i < nb ? i : 0; bt.stack[i] = stacks[b0 + j]; bt.WKS[i] = wks[b0 + j]; bt.out[i] = outs[b0 + j]; }
        TimedScope ts(ctx, "adi_eff_stack", 8.0 * nb * (2.0 * n * n + (double)m * n + 2.0 * n * m + 2.0 * nstrip * 16.0 * kst * 4.0), 4.0 * nb * n * n * (double)m);
        hipLaunchKernelGGL(k_eff_stack, dim3(ceil_div(kst * 64, 256), 2 * nstrip, nb), dim3(256), 0, ctx->stream, n, m, nstrip, kst, lds_, ldwk, bt);
    }
    DRE_HIP(hipGetLastError());
}

// tr((T G)^2) with G and T read straight from global memory (k x k, leading dimensions k and ldt), no LDS: one workgroup,
// 32 x 32 block pairs bi <= bj spread over the waves (same arithmetic as gram_norm_body); then the decision of adi.jl:115-123.
__device__ __forceinline__ void gram_norm_global(int k, const double* __restrict__ G, const double* __restrict__ T, int ldt, int tdiag,
                                                 double alpha, AdiState* st, int iters_after, double* red) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, nw = blockDim.x >> 6;
    const int kp = (k + 31) & ~31;
    double s = 0.0;
    if (tdiag) {
        for (int idx = tid; idx < k * k; idx += blockDim.x) {
            const int r = idx % k, c = idx / k;
            const double g = G[idx];
            s += T[r + (size_t)r * ldt] * T[c + (size_t)c * ldt] * g * g;
        }
    } else {
        const int nbk = kp / 32, lr = lane & 15, lk = lane >> 4;
        int p = 0;
        for (int bi = 0; bi < nbk; ++bi)
            for (int bj = bi; bj < nbk; ++bj, ++p) {
                if (p % nw != wave) continue;
                v4d m[2][2], nn[2][2];
#pragma unroll
                for (int x = 0; x < 2; ++x)
#pragma unroll
                    for (int y = 0; y < 2; ++y) { m[x][y] = (v4d){0.0, 0.0, 0.0, 0.0}; nn[x][y] = (v4d){0.0, 0.0, 0.0, 0.0}; }
                const int ri0 = bi * 32 + lr, ri1 = ri0 + 16, rj0 = bj * 32 + lr, rj1 = rj0 + 16;
                for (int kk = 0; kk < kp / 4; ++kk) {
                    const int c = kk * 4 + lk;                    // inner index
                    const bool cok = c < k;
                    // M = T G: A operand T[row, c], B operand G[c, col] = G[col, c] (G symmetric);  N = G T': A operand G[row, c], B operand T[col, c]
                    const double ta0 = (cok && ri0 < k) ? T[ri0 + (size_t)c * ldt] : 0.0, ta1 = (cok && ri1 < k) ? T[ri1 + (size_t)c * ldt] : 0.0;
                    const double gb0 = (cok && rj0 < k) ? G[rj0 + (size_t)c * k] : 0.0,  gb1 = (cok && rj1 < k) ? G[rj1 + (size_t)c * k] : 0.0;
                    const double ga0 = (cok && ri0 < k) ? G[ri0 + (size_t)c * k] : 0.0,  ga1 = (cok && ri1 < k) ? G[ri1 + (size_t)c * k] : 0.0;
                    const double tb0 = (cok && rj0 < k) ? T[rj0 + (size_t)c * ldt] : 0.0, tb1 = (cok && rj1 < k) ? T[rj1 + (size_t)c * ldt] : 0.0;
                    m[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(ta0, gb0, m[0][0], 0, 0, 0);
                    m[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(ta0, gb1, m[0][1], 0, 0, 0);
                    m[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(ta1, gb0, m[1][0], 0, 0, 0);
                    m[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(ta1, gb1, m[1][1], 0, 0, 0);
                    nn[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(ga0, tb0, nn[0][0], 0, 0, 0);
                    nn[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(ga0, tb1, nn[0][1], 0, 0, 0);
                    nn[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(ga1, tb0, nn[1][0], 0, 0, 0);
                    nn[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(ga1, tb1, nn[1][1], 0, 0, 0);
                }
                double sum = 0.0;
#pragma unroll
                for (int x = 0; x < 2; ++x)
#pragma unroll
                    for (int y = 0; y < 2; ++y)
#pragma unroll
                        for (int r = 0; r < 4; ++r) sum += m[x][y][r] * nn[x][y][r];
                s += (bi == bj) ? sum : 2.0 * sum;
            }
    }
    s = block_sum(s, red);
    if (tid == 0) {
        const double nrm = fabs(alpha) * sqrt(fmax(s, 0.0));
        st->res_norm = nrm;
        st->iters = iters_after;
        st->norms[iters_after & 511] = nrm;          // ring: the host reads every chunk (< 512 iterations) before it wraps
        if (nrm <= st->abstol || iters_after >= st->maxiters) st->done = 1;
    }
}

// 16 x 16 output tile  C = A_strip B  over the K-steps [t0, t1) of this wave:  A packed (64 consecutive doubles per K-step),
// B = X[4 t + lk, col] column-major; all loads of a batch of 24 K-steps are issued before its first MFMA.
#define ADI_FAST_KB 24
// PACKED: bp points at (column tile, lane) of the packed residual, K-step stride pstride doubles
template <bool PACKED>
__device__ __forceinline__ v4d adi_fast_tile(const double* __restrict__ ap, const double* __restrict__ bp, bool colok, int lk, int n, int t0, int t1, size_t pstride) {
    // branch-free: every load address is clamped into range (the packed A strip is padded, rows of B are clamped to n - 1) and
    // out-of-range operands are zeroed by a select, so the 48 loads of a batch issue back to back without exec-mask branches
    v4d acc = (v4d){0.0, 0.0, 0.0, 0.0};
    if (t1 <= t0) return acc;
    for (int tb = t0; tb < t1; tb += ADI_FAST_KB) {
        double av[ADI_FAST_KB], bv[ADI_FAST_KB];
#pragma unroll
        for (int u = 0; u < ADI_FAST_KB; ++u) {
            const int t = min(tb + u, t1 - 1);
            const int row = 4 * t + lk;
            av[u] = ap[(size_t)t * 64];
            bv[u] = PACKED ? bp[(size_t)t * pstride] : bp[min(row, n - 1) - lk];
        }
#pragma unroll
        for (int u = 0; u < ADI_FAST_KB; ++u) {
            // lane & 15 is the ROW of the A fragment but the COLUMN of the B fragment: the column mask applies to B only
            const int row = 4 * (tb + u) + lk;
            const bool kok = (tb + u < t1) && row < n;
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(kok ? av[u] : 0.0, (kok && colok) ? bv[u] : 0.0, acc, 0, 0, 0);
        }
    }
    return acc;
}
// The same for NT column tiles per wave (wide residuals / larger n): one A fragment feeds NT MFMAs, so the packed strip is read once per
// NT tiles, and the operand batches are double buffered — the loads of batch i + 1 are in flight while batch i multiplies.
template <int NT, bool PACKED>
__device__ __forceinline__ void adi_fast_tiles(const double* __restrict__ ap, const double* __restrict__ R, int ldr, int col0, int k, int lk, int lr, int n,
                                               int t0, int t1, v4d (&acc)[NT], const double* __restrict__ Rp, size_t pstride) {
    constexpr int KB = 24 / NT;
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[j] = (v4d){0.0, 0.0, 0.0, 0.0};
    if (t1 <= t0) return;
    const double* bp[NT];
    bool cok[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int col = col0 + j * 16 + lr;
        cok[j] = col < k;
        bp[j] = PACKED ? Rp + ((size_t)(col0 / 16 + j)) * 64 + (threadIdx.x & 63) : R + (size_t)(cok[j] ? col : 0) * ldr;
    }
    double a0[KB], a1[KB], b0[NT][KB], b1[NT][KB];
    auto load = [&](double (&av)[KB], double (&bv)[NT][KB], int tb) {
#pragma unroll
        for (int u = 0; u < KB; ++u) {
            const int t = min(tb + u, t1 - 1);
            const int row = min(4 * t + lk, n - 1);
            av[u] = ap[(size_t)t * 64];
#pragma unroll
            for (int j = 0; j < NT; ++j) bv[j][u] = PACKED ? bp[j][(size_t)t * pstride] : bp[j][row];
        }
    };
    auto mma = [&](const double (&av)[KB], const double (&bv)[NT][KB], int tb) {
#pragma unroll
        for (int u = 0; u < KB; ++u) {
            const bool kok = (tb + u < t1) && 4 * (tb + u) + lk < n;
            const double x = kok ? av[u] : 0.0;
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, (kok && cok[j]) ? bv[j][u] : 0.0, acc[j], 0, 0, 0);
        }
    };
    load(a0, b0, t0);
    for (int tb = t0; tb < t1; tb += 2 * KB) {
        load(a1, b1, tb + KB);
        mma(a0, b0, tb);
        load(a0, b0, tb + 2 * KB);
        mma(a1, b1, tb + KB);
    }
}
template <int NT>
__device__ __forceinline__ void adi_fast_strip_group(const AdiFastArgs& a, int hs, int tg, double* part /* [4][4*NT][64] */) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lk = lane >> 4, lr = lane & 15;
    const int k = a.k, n = a.n;
    const int half = hs / a.nstrip, s = hs - half * a.nstrip;
    const int col0 = tg * NT * 16;
    const int erow = s * 16 + lk + 4 * wave;
    double rold[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int col = col0 + j * 16 + lr;
        rold[j] = (half == 1 && col < k && erow < n) ? a.Rcur[erow + (size_t)col * a.ldr] : 0.0;
    }
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const int per = (a.kst + 3) >> 2, t0 = wv * per, t1 = min(a.kst, t0 + per);
    v4d acc[NT];
    const int ctp = (k + 15) >> 4;
    if (a.Rpc) adi_fast_tiles<NT, true>(a.Apack + (size_t)hs * a.kst * 64 + lane, a.Rcur, a.ldr, col0, k, lk, lr, n, t0, t1, acc, a.Rpc, (size_t)ctp * 64);
    else adi_fast_tiles<NT, false>(a.Apack + (size_t)hs * a.kst * 64 + lane, a.Rcur, a.ldr, col0, k, lk, lr, n, t0, t1, acc, nullptr, 0);
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) part[((size_t)wave * 4 * NT + j * 4 + r) * 64 + lane] = acc[j][r];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int col = col0 + j * 16 + lr;
        const size_t o = ((size_t)j * 4 + wave) * 64 + lane, ws = (size_t)4 * NT * 64;
        const double v = ((part[o] + part[ws + o]) + part[2 * ws + o]) + part[3 * ws + o];
        const bool ok = col < k && erow < n;
        const double rnew = rold[j] - a.two_mu * v;
        if (ok) {
            if (half == 0) a.V[erow + (size_t)col * a.ldv] = v;
            else a.Rnext[erow + (size_t)col * a.ldr_next] = rnew;
        }
        // this wave's 64 results are K-step 4 s + wave of column tile col0 / 16 + j of the packed residual
        if (half == 1 && a.Rpn && col0 / 16 + j < ctp) a.Rpn[((size_t)(4 * s + wave) * ctp + (col0 / 16 + j)) * 64 + lane] = ok ? rnew : 0.0;
    }
}
// Large n (mode 1): every wave owns one 16-row strip over the FULL K range and NT column tiles; the four waves of a workgroup share the
// B operand (64 rows x NT*16 columns of R per K-chunk), staged through LDS with coalesced loads and double buffered — without it every
// wave re-gathers R in 32-byte pieces and the vector memory pipeline, not the matrix cores, bounds the launch (77 us at n = 1357, k = 160
// against a matrix-core floor of 15 us).  The A fragments stay single 512-byte loads of the packed strip, register double buffered.
#define ADI_WIDE_KC 64
#define ADI_WIDE_LDB 68
template <int NT>
__device__ __forceinline__ void adi_fast_wide(const AdiFastArgs& a, int sg, int tg, double* lds /* 2 x NT*16 x ADI_WIDE_LDB */) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lk = lane >> 4, lr = lane & 15;
    const int k = a.k, n = a.n, kst = a.kst;
    const int hs = sg * 4 + wave;
    const bool active = hs < 2 * a.nstrip;
    const int hsc = active ? hs : 0;
    const int half = hsc / a.nstrip, s = hsc - half * a.nstrip;
    const int col0 = tg * NT * 16, ncolw = NT * 16;
    const double* __restrict__ ap = a.Apack + (size_t)hsc * kst * 64 + lane;
    v4d acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[j] = (v4d){0.0, 0.0, 0.0, 0.0};
    const int nchunk = (n + ADI_WIDE_KC - 1) / ADI_WIDE_KC;
    constexpr int PER = NT * 16 * ADI_WIDE_KC / 256;           // doubles of the B chunk per thread
    double breg[PER], areg[16];
    auto load_b = [&](int ch) {
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int idx = tid + 256 * i, row = ch * ADI_WIDE_KC + (idx & (ADI_WIDE_KC - 1)), col = col0 + (idx >> 6);
            breg[i] = a.Rcur[min(row, n - 1) + (size_t)min(col, k - 1) * a.ldr];
            if (row >= n || col >= k) breg[i] = 0.0;
        }
    };
    auto store_b = [&](int buf) {
        double* B = lds + (size_t)buf * ncolw * ADI_WIDE_LDB;
#pragma unroll
        for (int i = 0; i < PER; ++i) { const int idx = tid + 256 * i; B[(idx >> 6) * ADI_WIDE_LDB + (idx & (ADI_WIDE_KC - 1))] = breg[i]; }
    };
    auto load_a = [&](int ch) {
#pragma unroll
        for (int u = 0; u < 16; ++u) { const int t = min(ch * 16 + u, kst - 1); areg[u] = ap[(size_t)t * 64]; }
    };
    load_b(0); load_a(0);
    store_b(0);
    __syncthreads();
    for (int ch = 0; ch < nchunk; ++ch) {
        const double* B = lds + (size_t)(ch & 1) * ncolw * ADI_WIDE_LDB;
        double acur[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) acur[u] = (ch * 16 + u < kst) ? areg[u] : 0.0;
        if (ch + 1 < nchunk) { load_b(ch + 1); load_a(ch + 1); }         // in flight while this chunk multiplies
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int j = 0; j < NT; ++j)
                acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(acur[u], B[(j * 16 + lr) * ADI_WIDE_LDB + 4 * u + lk], acc[j], 0, 0, 0);
        if (ch + 1 < nchunk) store_b((ch + 1) & 1);
        __syncthreads();
    }
    if (!active) return;
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = s * 16 + lk + 4 * r, col = col0 + j * 16 + lr;
            if (row < n && col < k) {
                if (half == 0) a.V[row + (size_t)col * a.ldv] = acc[j][r];
                else a.Rnext[row + (size_t)col * a.ldr_next] = a.Rcur[row + (size_t)col * a.ldr] - a.two_mu * acc[j][r];
            }
        }
}
// Workgroups of one launch (256 threads = 4 waves that split K):
//   [0, 2 nstrip ct)            tile (half, strip, column tile) of V = Seff_top R (half 0) or R_next = R - 2 mu Seff_bot R (half 1)
//   [.., + ct ct)               Gram tile (ta, tb) of the INPUT residual R (= output of the previous launch) -> G_prev
//   last                        norm + decision from G_prev2 (the Gram matrix the previous launch produced)
__global__ __launch_bounds__(256) void k_adi_fast(AdiFastArgs a) {
    if (a.st->done) return;
    __shared__ double partbuf[4 * 16 * 64];                       // K-quarter partial tiles: [4 waves][4 NT][64 lanes], NT <= 4
    double (*part)[4][64] = reinterpret_cast<double (*)[4][64]>(partbuf);
    __shared__ double nred[17];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lk = lane >> 4;
    const int k = a.k, ct = (k + 15) >> 4, n = a.n;
    int b = blockIdx.x;
    // XCD-aware order of the tile workgroups: workgroups are dealt round-robin over the 8 XCDs, so the ct column tiles that re-read the same
    // packed strip get indices that differ by a multiple of 8 (strip-major with a stride padded to 8; the padding workgroups exit) — the
    // re-reads then hit that XCD's L2 instead of going out to the fabric ct times
    const int hstride = (2 * a.nstrip + 7) & ~7;
    const int ntile = a.nt > 0 ? a.nt : 1, ngroups = (ct + ntile - 1) / ntile;
    // grid order: norm workgroups, Gram tiles, then the strip tiles — the riders are the longer dependent chains, so they start first and
    // run beside the strips instead of queueing behind them
    const int nnorm = ct * ((ct + 3) >> 2), nrider = nnorm + ct * ct;
    const int nsw = a.do_strips ? (a.mode == 1 ? ((2 * a.nstrip + 3) / 4) * ngroups : hstride * ngroups) : 0;
    b = (b >= nrider) ? b - nrider : b + nsw;                     // strips occupy [0, nsw) of the logical index, riders follow
    if (b < nsw && a.mode == 1) {
        extern __shared__ double wide_lds[];
        const int nsg = (2 * a.nstrip + 3) / 4;
        const int tg = b / nsg, sg = b - tg * nsg;
        if (ntile == 1) adi_fast_wide<1>(a, sg, tg, wide_lds);
        else if (ntile == 2) adi_fast_wide<2>(a, sg, tg, wide_lds);
        else adi_fast_wide<4>(a, sg, tg, wide_lds);
        return;
    }
    if (b < nsw && ntile > 1) {
        const int tg = b / hstride, hs = b - tg * hstride;
        if (hs >= 2 * a.nstrip) return;
        if (ntile == 2) adi_fast_strip_group<2>(a, hs, tg, partbuf);
        else adi_fast_strip_group<4>(a, hs, tg, partbuf);
        return;
    }
    if (b < nsw) {
        const int tc = b / hstride, hs = b - tc * hstride;
        if (hs >= 2 * a.nstrip) return;
        const int half = hs / a.nstrip, s = hs - half * a.nstrip;
        const int col = tc * 16 + (lane & 15);
        const bool colok = col < k;
        // old residual entry of the element this thread finishes in the epilogue (requested early)
        const int erow = s * 16 + (lane >> 4) + 4 * wave;
        const double rold = (half == 1 && colok && erow < n) ? a.Rcur[erow + (size_t)col * a.ldr] : 0.0;
        const int wv = __builtin_amdgcn_readfirstlane(wave);
        const int per = (a.kst + 3) >> 2, t0 = wv * per, t1 = min(a.kst, t0 + per);
        const v4d acc = a.Rpc ? adi_fast_tile<true>(a.Apack + (size_t)hs * a.kst * 64 + lane, a.Rpc + (size_t)tc * 64 + lane, colok, lk, n, t0, t1, (size_t)ct * 64)
                              : adi_fast_tile<false>(a.Apack + (size_t)hs * a.kst * 64 + lane, a.Rcur + (size_t)(colok ? col : 0) * a.ldr + lk, colok, lk, n, t0, t1, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) part[wave][r][lane] = acc[r];
        __syncthreads();
        // thread (wave = r, lane) finishes element (row lk + 4 r, column lane & 15) of the tile: fixed-order sum over the K quarters
        const double v = ((part[0][wave][lane] + part[1][wave][lane]) + part[2][wave][lane]) + part[3][wave][lane];
        const bool ok = colok && erow < n;
        const double rnew = rold - a.two_mu * v;
        if (ok) {
            if (half == 0) a.V[erow + (size_t)col * a.ldv] = v;
            else a.Rnext[erow + (size_t)col * a.ldr_next] = rnew;
        }
        if (half == 1 && a.Rpn) a.Rpn[((size_t)(4 * s + wave) * ct + tc) * 64 + lane] = ok ? rnew : 0.0;
        return;
    }
    b -= nsw;
    if (b < ct * ct) {
        if (!a.G_prev) return;
        // Gram tile (ta, tb) of the input residual:  G[ta-cols, tb-cols] = R(:, ta)' R(:, tb)
        const int ta = b % ct, tb = b / ct;
        if (ta > tb) return;                                      // the mirrored tile is written by (tb, ta)'s partner below
        const int ca = ta * 16 + (lane & 15), cb = tb * 16 + (lane & 15);
        const bool aok = ca < k, bok = cb < k;
        const double* __restrict__ pa = a.Rcur + (size_t)(aok ? ca : 0) * a.ldr + lk;
        const double* __restrict__ pb = a.Rcur + (size_t)(bok ? cb : 0) * a.ldr + lk;
        const int wv = __builtin_amdgcn_readfirstlane(wave);
        const int per = (a.kst + 3) >> 2, t0 = wv * per, t1 = min(a.kst, t0 + per);
        v4d acc = (v4d){0.0, 0.0, 0.0, 0.0};
        for (int tb0 = t0; tb0 < t1; tb0 += ADI_FAST_KB) {
            double av[ADI_FAST_KB], bv[ADI_FAST_KB];
#pragma unroll
            for (int u = 0; u < ADI_FAST_KB; ++u) {
                const int t = min(tb0 + u, t1 - 1);
                const int off = min(4 * t + lk, n - 1) - lk;
                if (a.Rpc) {                                   // (launch-uniform)
                    av[u] = a.Rpc[((size_t)t * ct + ta) * 64 + lane];
                    bv[u] = a.Rpc[((size_t)t * ct + tb) * 64 + lane];
                } else {
                    av[u] = pa[off];
                    bv[u] = pb[off];
                }
            }
#pragma unroll
            for (int u = 0; u < ADI_FAST_KB; ++u) {
                const bool ok = (tb0 + u < t1) && 4 * (tb0 + u) + lk < n;
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64((ok && aok) ? av[u] : 0.0, (ok && bok) ? bv[u] : 0.0, acc, 0, 0, 0);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) part[wave][r][lane] = acc[r];
        __syncthreads();
        const double v = ((part[0][wave][lane] + part[1][wave][lane]) + part[2][wave][lane]) + part[3][wave][lane];
        const int gr = ta * 16 + (lane >> 4) + 4 * wave, gc = tb * 16 + (lane & 15);
        if (gr < k && gc < k) {
            a.G_prev[gr + (size_t)gc * k] = v;
            if (ta != tb) a.G_prev[gc + (size_t)gr * k] = v;
        }
        return;
    }
    b -= ct * ct;
    if (!a.G_prev2) return;
    // norm of the residual whose Gram matrix the previous launch left: tr((T G)^2) = sum_ij M_ij N_ij with M = T G, N = G T' (= M');
    // workgroup b owns tile row b of M and N (one 16 x 16 tile per wave and pass), all operand loads are 128-byte column segments
    // issued before the first MFMA.  The ct partial sums meet through a ticket: the workgroup whose atomic add comes last sums
    // them in a fixed order and takes the decision of adi.jl:115-123.
    // workgroup b = (tile row I, group of four tile columns): one 16 x 16 tile pair per wave
    const int ngrp = (ct + 3) >> 2;
    const int I = b / ngrp, J0 = (b - I * ngrp) * 4;
    double sloc = 0.0;
    const double* __restrict__ G = a.G_prev2; const double* __restrict__ T = a.T;
    const int lr = lane & 15;
    for (int J = J0 + wave; J < min(ct, J0 + 4); J += 4) {
        v4d mm = (v4d){0.0, 0.0, 0.0, 0.0}, nn = (v4d){0.0, 0.0, 0.0, 0.0};
        const int ri = I * 16 + lr, rj = J * 16 + lr;
        const bool iok = ri < k, jok = rj < k;
        const int ric = iok ? ri : 0, rjc = jok ? rj : 0;
        for (int kk0 = 0; kk0 < ct * 4; kk0 += 16) {
            double ta[16], gb[16], ga[16], tb[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int c = min((kk0 + u) * 4 + lk, k - 1);
                ta[u] = T[ric + (size_t)c * a.ldt]; gb[u] = G[rjc + (size_t)c * k];
                ga[u] = G[ric + (size_t)c * k];     tb[u] = T[rjc + (size_t)c * a.ldt];
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const bool cok = (kk0 + u) < ct * 4 && (kk0 + u) * 4 + lk < k;
                mm = __builtin_amdgcn_mfma_f64_16x16x4f64((cok && iok) ? ta[u] : 0.0, (cok && jok) ? gb[u] : 0.0, mm, 0, 0, 0);
                nn = __builtin_amdgcn_mfma_f64_16x16x4f64((cok && iok) ? ga[u] : 0.0, (cok && jok) ? tb[u] : 0.0, nn, 0, 0, 0);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) sloc += mm[r] * nn[r];
    }
    sloc = block_sum(sloc, nred);
    if (tid == 0) {
        const int nrb = ct * ngrp;                       // norm workgroups of this launch
        __hip_atomic_store(a.nws + b, sloc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unsigned* ticket = reinterpret_cast<unsigned*>(a.nws + ADI_FAST_NWS - 1);
        const unsigned tk = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tk == (unsigned)(nrb - 1)) {
            double tot = 0.0;
            for (int i = 0; i < nrb; ++i) tot += __hip_atomic_load(a.nws + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);       // ready for the next launch
            const double nrm = fabs(a.alpha) * sqrt(fmax(tot, 0.0));
            AdiState* st = a.st;
            st->res_norm = nrm;
            st->iters = a.it_prev2;
            st->norms[a.it_prev2 & 511] = nrm;
            if (nrm <= st->abstol || a.it_prev2 >= st->maxiters) st->done = 1;
        }
    }
}
// algorithmic traffic of one launch: the packed effective stack once (2 n x n), R read, V and R_next written (+ the Gram pass over R)
void adi_fast_cost(const AdiFastArgs& a, double* bytes, double* flops) {
    *flops = (a.do_strips ? 4.0 * a.n * (double)a.n * a.k : 0.0) + (a.G_prev ? 2.0 * a.n * (double)a.k * a.k : 0.0);
    *bytes = a.do_strips ? 8.0 * (2.0 * a.nstrip * 16.0 * a.kst * 4.0 + 3.0 * a.n * a.k) : 8.0 * (double)a.n * a.k;
}
__global__ void k_adi_pack_r(int n, int k, int ct, const double* __restrict__ R, int ldr, double* __restrict__ Rp, const AdiState* st) {
    if (st && st->done) return;
    const int t = blockIdx.x, j = blockIdx.y, lane = threadIdx.x;
    const int row = 4 * t + (lane >> 4), col = 16 * j + (lane & 15);
    Rp[((size_t)t * ct + j) * 64 + lane] = (row < n && col < k) ? R[row + (size_t)col * ldr] : 0.0;
}
void adi_fast_pack_r(Ctx* ctx, int n, int k, const double* R, int ldr, double* Rp, const AdiState* st) {
    const int ct = (k + 15) >> 4;
    hipLaunchKernelGGL(k_adi_pack_r, dim3(4 * adi_fast_nstrip(n), ct), dim3(64), 0, ctx->stream, n, k, ct, R, ldr, Rp, st);
}
void adi_fast_iter(Ctx* ctx, const AdiFastArgs& a) {
    DRE_REQUIRE(a.k >= 1 && a.k <= ADI_FAST_MAX_K, "adi_fast_iter: residual too wide");
    const int ct = (a.k + 15) >> 4;
    const int ntile = a.nt > 0 ? a.nt : 1;
    DRE_REQUIRE(ntile == 1 || ntile == 2 || ntile == 4, "adi_fast_iter: nt must be 1, 2 or 4");
    const int ngroups = (ct + ntile - 1) / ntile;
    const int nsw = a.do_strips ? (a.mode == 1 ? ((2 * a.nstrip + 3) / 4) * ngroups : ((2 * a.nstrip + 7) & ~7) * ngroups) : 0;
    const size_t lds = (a.do_strips && a.mode == 1) ? (size_t)2 * ntile * 16 * ADI_WIDE_LDB * sizeof(double) : 0;
    if (lds > 48 * 1024) lds_attr(ctx, (const void*)k_adi_fast, 80 * 1024);
    if (a.chain_timed) {
        hipLaunchKernelGGL(k_adi_fast, dim3(nsw + ct * ct + ct * ((ct + 3) / 4)), dim3(256), lds, ctx->stream, a);
    } else {
        double by, fl;
        adi_fast_cost(a, &by, &fl);
        TimedScope ts(ctx, a.do_strips ? "adi_fast_iter" : "adi_fast_flush", by, fl);
        hipLaunchKernelGGL(k_adi_fast, dim3(nsw + ct * ct + ct * ((ct + 3) / 4)), dim3(256), lds, ctx->stream, a);
    }
    DRE_HIP(hipGetLastError());
}


// =============================================================================================
// Group ADI chain (round 3): g consecutive ADI iterations in ONE launch.
//   R_i = Pi_i R_0,  V_i = Om_i R_0   with  Pi_i = P_{p+i-1} ... P_p,  P_s = I - 2 mu_s E' A_s,  Om_i = A_{p+i} Pi_i,  A_s = (F' + mu_s E')^-1
// (perform_single_step!, adi.jl:149-179, applied g times: same iterates, the operator products are formed once per time step on the side
// stream — gdre.hip, group_ops_prepare).  The launch-per-iteration chain at n = 371 is bound by the dependent kernel boundary and the
// memory round trips of a 6-us kernel, not by its 32 MFLOP; the group stack [Om_0 .. Om_{g-1}; Pi_1 .. Pi_g] (2 g blocks of n x n, packed in
// the MFMA A-operand order like the single-iteration stack) turns g of those launches into one with 2 g times the tile workgroups.
// Riders: the Gram matrices of the g residuals the PREVIOUS launch produced, and the norms + decisions (adi.jl:115-123, taken in iteration
// order by the last norm workgroup to arrive) for the g residuals of the launch before that.
// =============================================================================================
__global__ __launch_bounds__(256) void k_pack_blocks(int n, int nblk, int nstrip, int kst, const double* __restrict__ src, int lds_, double* __restrict__ out) {
    // out[((b * nstrip + s) * kst + t) * 64 + lane] = src[b n + 16 s + (lane & 15), 4 t + (lane >> 4)]   (zero padded)
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int hs = blockIdx.y;                                     // b * nstrip + s
    if (t >= kst) return;
    const int b = hs / nstrip, sidx = hs - b * nstrip;
    const int row = 16 * sidx + (lane & 15), col = 4 * t + (lane >> 4);
    out[((size_t)hs * kst + t) * 64 + lane] = (row < n && col < n) ? src[(size_t)b * n + row + (size_t)col * lds_] : 0.0;
}
void adi_group_pack(Ctx* ctx, int n, int nblk, const double* src, int lds_, double* out) {
    const int nstrip = adi_fast_nstrip(n), kst = adi_fast_kst(n);
    TimedScope ts(ctx, "adi_group_pack", 16.0 * nblk * n * (double)n, 0.0);
    hipLaunchKernelGGL(k_pack_blocks, dim3(ceil_div(kst, 4), nblk * nstrip), dim3(256), 0, ctx->stream, n, nblk, nstrip, kst, src, lds_, out);
    DRE_HIP(hipGetLastError());
}

__global__ __launch_bounds__(256) void k_adi_group(AdiGroupArgs a) {
    // the flag is REQUESTED here and looked at after the operand loads of the tile product are in flight: a dependent ~1 us round trip to L2
    // in front of every workgroup's first load otherwise (launches enqueued past the end of the solve do their loads for nothing)
    const int done_flag = a.st->done;
    __shared__ double partbuf[4 * 4 * 64];
    double (*part)[4][64] = reinterpret_cast<double (*)[4][64]>(partbuf);
    __shared__ double nred[17];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lk = lane >> 4;
    const int k = a.k, ct = (k + 15) >> 4, n = a.n, g = a.g;
    const int ngrp = (ct + 3) >> 2;
    const int nnorm = a.n_prev2 * ct * ngrp, ngram = a.n_prev * ct * ct;
    const int hstride = (2 * g * a.nstrip + 7) & ~7;
    const int nsw = a.do_strips ? hstride * ct : 0;
    int b = blockIdx.x;
    const int nrider = nnorm + ngram;
    b = (b >= nrider) ? b - nrider : b + nsw;                     // riders first in the grid, strips occupy [0, nsw) of the logical index
    if (b < nsw) {
        const int tc = b / hstride, hs = b - tc * hstride;
        if (hs >= 2 * g * a.nstrip) return;
        const int blk = hs / a.nstrip, s = hs - blk * a.nstrip;    // blk < g: V of iteration blk;  blk >= g: residual after iteration blk - g + 1
        const int col = tc * 16 + (lane & 15);
        const bool colok = col < k;
        const int erow = s * 16 + (lane >> 4) + 4 * wave;
        const int wv = __builtin_amdgcn_readfirstlane(wave);
        const int per = (a.kst + 3) >> 2, t0 = wv * per, t1 = min(a.kst, t0 + per);
        const v4d acc = adi_fast_tile<true>(a.Gpack + (size_t)hs * a.kst * 64 + lane, a.Rpc + (size_t)tc * 64 + lane, colok, lk, n, t0, t1, (size_t)ct * 64);
        if (done_flag) return;
#pragma unroll
        for (int r = 0; r < 4; ++r) part[wave][r][lane] = acc[r];
        __syncthreads();
        const double v = ((part[0][wave][lane] + part[1][wave][lane]) + part[2][wave][lane]) + part[3][wave][lane];
        const bool ok = colok && erow < n;
        if (blk < g) {
            if (ok) a.V[(size_t)blk * k * a.ldv + erow + (size_t)col * a.ldv] = v;
        } else {
            const int i = blk - g;
            if (ok) a.Rring[(size_t)i * k * a.ldr + erow + (size_t)col * a.ldr] = v;
            a.Rpk[(size_t)i * a.rpd + ((size_t)(4 * s + wave) * ct + tc) * 64 + lane] = ok ? v : 0.0;
        }
        return;
    }
    b -= nsw;
    if (b < ngram) {
        // Gram tile (ta, tb) of residual i of the previous launch
        const int i = b / (ct * ct), bb = b - i * ct * ct;
        const int ta = bb % ct, tb = bb / ct;
        if (ta > tb) return;
        const double* __restrict__ Rp = a.Rp_prev + (size_t)i * a.rpd;
        const bool aok = ta * 16 + (lane & 15) < k, bok = tb * 16 + (lane & 15) < k;
        const int wv = __builtin_amdgcn_readfirstlane(wave);
        const int per = (a.kst + 3) >> 2, t0 = wv * per, t1 = min(a.kst, t0 + per);
        v4d acc = (v4d){0.0, 0.0, 0.0, 0.0};
        for (int tb0 = t0; tb0 < t1; tb0 += ADI_FAST_KB) {
            double av[ADI_FAST_KB], bv[ADI_FAST_KB];
#pragma unroll
            for (int u = 0; u < ADI_FAST_KB; ++u) {
                const int t = min(tb0 + u, t1 - 1);
                av[u] = Rp[((size_t)t * ct + ta) * 64 + lane];
                bv[u] = Rp[((size_t)t * ct + tb) * 64 + lane];
            }
#pragma unroll
            for (int u = 0; u < ADI_FAST_KB; ++u) {
                const bool ok = (tb0 + u < t1) && 4 * (tb0 + u) + lk < n;
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64((ok && aok) ? av[u] : 0.0, (ok && bok) ? bv[u] : 0.0, acc, 0, 0, 0);
            }
        }
        if (done_flag) return;
#pragma unroll
        for (int r = 0; r < 4; ++r) part[wave][r][lane] = acc[r];
        __syncthreads();
        const double v = ((part[0][wave][lane] + part[1][wave][lane]) + part[2][wave][lane]) + part[3][wave][lane];
        const int gr = ta * 16 + (lane >> 4) + 4 * wave, gc = tb * 16 + (lane & 15);
        double* __restrict__ G = a.G_prev + (size_t)i * k * k;
        if (gr < k && gc < k) {
            G[gr + (size_t)gc * k] = v;
            if (ta != tb) G[gc + (size_t)gr * k] = v;
        }
        return;
    }
    b -= ngram;
    if (b >= nnorm) return;
    // norm workgroup (residual i of the launch before the previous one, tile row I, four tile columns): tr((T G)^2) partial sums
    const int per_it = ct * ngrp;
    const int i = b / per_it, bb = b - i * per_it;
    const int I = bb / ngrp, J0 = (bb - I * ngrp) * 4;
    double sloc = 0.0;
    const double* __restrict__ G = a.G_prev2 + (size_t)i * k * k; const double* __restrict__ T = a.T;
    const int lr = lane & 15;
    for (int J = J0 + wave; J < min(ct, J0 + 4); J += 4) {
        v4d mm = (v4d){0.0, 0.0, 0.0, 0.0}, nn = (v4d){0.0, 0.0, 0.0, 0.0};
        const int ri = I * 16 + lr, rj = J * 16 + lr;
        const bool iok = ri < k, jok = rj < k;
        const int ric = iok ? ri : 0, rjc = jok ? rj : 0;
        for (int kk0 = 0; kk0 < ct * 4; kk0 += 16) {
            double ta[16], gb[16], ga[16], tb[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int c = min((kk0 + u) * 4 + lk, k - 1);
                ta[u] = T[ric + (size_t)c * a.ldt]; gb[u] = G[rjc + (size_t)c * k];
                ga[u] = G[ric + (size_t)c * k];     tb[u] = T[rjc + (size_t)c * a.ldt];
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const bool cok = (kk0 + u) < ct * 4 && (kk0 + u) * 4 + lk < k;
                mm = __builtin_amdgcn_mfma_f64_16x16x4f64((cok && iok) ? ta[u] : 0.0, (cok && jok) ? gb[u] : 0.0, mm, 0, 0, 0);
                nn = __builtin_amdgcn_mfma_f64_16x16x4f64((cok && iok) ? ga[u] : 0.0, (cok && jok) ? tb[u] : 0.0, nn, 0, 0, 0);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) sloc += mm[r] * nn[r];
    }
    if (done_flag) return;
    sloc = block_sum(sloc, nred);
    if (tid == 0) {
        __hip_atomic_store(a.nws + b, sloc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unsigned* ticket = reinterpret_cast<unsigned*>(a.nws + ADI_FAST_NWS - 1);
        const unsigned tk = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tk == (unsigned)(nnorm - 1)) {
            // the last norm workgroup of the launch: the decisions of adi.jl:115-123 in ITERATION order — the first residual at or below
            // abstol (or at maxiters) ends the solve; the norms of later, speculatively computed iterations are not recorded
            AdiState* st = a.st;
            for (int it = 0; it < a.n_prev2; ++it) {
                double tot = 0.0;
                for (int q = 0; q < per_it; ++q) tot += __hip_atomic_load(a.nws + it * per_it + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const double nrm = fabs(a.alpha) * sqrt(fmax(tot, 0.0));
                const int iters_after = a.it0_prev2 + it;
                st->res_norm = nrm;
                st->iters = iters_after;
                st->norms[iters_after & 511] = nrm;          // ring: the host reads every chunk (< 512 iterations) before it wraps
                if (nrm <= st->abstol || iters_after >= st->maxiters) { st->done = 1; break; }
            }
            __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);       // ready for the next launch
        }
    }
}
void adi_group_cost(const AdiGroupArgs& a, double* bytes, double* flops) {
    *flops = (a.do_strips ? 2.0 * (2.0 * a.g) * a.n * (double)a.n * a.k : 0.0) + 2.0 * a.n_prev * a.n * (double)a.k * a.k;
    *bytes = a.do_strips ? 8.0 * (2.0 * a.g * a.nstrip * 16.0 * a.kst * 4.0 + (1.0 + 3.0 * a.g) * a.n * a.k) : 8.0 * (double)a.n_prev * a.n * a.k;
}
void adi_group_iter(Ctx* ctx, const AdiGroupArgs& a) {
    DRE_REQUIRE(a.k >= 1 && a.k <= ADI_GROUP_MAX_K && a.g >= 2 && a.g <= ADI_GROUP_MAX_G, "adi_group_iter: residual too wide or bad group size");
    const int ct = (a.k + 15) >> 4, ngrp = (ct + 3) >> 2;
    DRE_REQUIRE(a.g * ct * ngrp <= ADI_FAST_NWS - 1, "adi_group_iter: norm meeting point too small");
    const int nsw = a.do_strips ? ((2 * a.g * a.nstrip + 7) & ~7) * ct : 0;
    const int grid = nsw + a.n_prev * ct * ct + a.n_prev2 * ct * ngrp;
    if (grid == 0) return;
    hipLaunchKernelGGL(k_adi_group, dim3(grid), dim3(256), 0, ctx->stream, a);
    DRE_HIP(hipGetLastError());
}

// norms + decisions for the g residuals Rcat = [R_1 .. R_g] (n x g k) of a fan group, iterations iters0 + 1 .. iters0 + g: one Gram product
// (cross blocks included: the product is latency bound, the extra tiles ride along), its slab reduction and one decision launch
void residual_norm_group(Ctx* ctx, const Mat& Rcat, int g, int k, const Mat& T, bool tdiag, double alpha, AdiState* st, int iters0) {
    if (k > 96 && k <= 256 && !tdiag && g * k <= 1024) {
        // wide residual: one Gram product for the group, one batched product T G_jj, one decision launch (4 launches instead of 4 g)
        Mat Gall(ctx, g * k, g * k), TGall(ctx, k, g * k);
        // the Gram product of the whole group computes g^2 blocks for g: fine while it is latency bound, not when it is compute bound
        // (n = 20209, k = 112, g = 3: 4.6 GFLOP = 179 us against 3 x 23 us for the diagonal blocks alone)
        if (2.0 * (double)g * k * g * k * Rcat.rows > 1.5e9) {
            for (int j = 0; j < g; ++j)
                gemm(ctx, true, false, k, k, Rcat.rows, 1.0, Rcat.p + (size_t)j * k * Rcat.ld, Rcat.ld, Rcat.p + (size_t)j * k * Rcat.ld, Rcat.ld, 0.0,
                     Gall.p + (size_t)j * k + (size_t)j * k * Gall.ld, Gall.ld, st, "gemm_gram");
        } else gemm(ctx, true, false, 1.0, Rcat, Rcat, 0.0, Gall, st, "gemm_gram");
        std::vector<GemmBatchDesc> descs;
        for (int j = 0; j < g; ++j)
            descs.push_back({T.p, Gall.p + (size_t)j * k + (size_t)j * k * Gall.ld, TGall.p + (size_t)j * k * TGall.ld, nullptr, 1.0, k, k, k, T.ld, Gall.ld, TGall.ld, 0});
        gemm_batched(ctx, descs, "gemm_norm");
        TimedScope ts(ctx, "ldlt_norm", 16.0 * g * k * k, 4.0 * g * k * k);
        hipLaunchKernelGGL(k_trace_sq_multi, dim3(g), dim3(512), 0, ctx->stream, k, g, (const double*)TGall.p, TGall.ld, alpha, st, iters0);
        DRE_HIP(hipGetLastError());
        return;
    }
    if (k > 96 || g * k > 512) {
        for (int j = 0; j < g; ++j) { Mat Rj = Rcat.colsview(j * k, k); residual_norm_step(ctx, Rj, T, tdiag, alpha, st, iters0 + j + 1); }
        return;
    }
    Mat Gall(ctx, g * k, g * k);
    if (2.0 * (double)g * k * g * k * Rcat.rows > 1.5e9) {
        for (int j = 0; j < g; ++j)
            gemm(ctx, true, false, k, k, Rcat.rows, 1.0, Rcat.p + (size_t)j * k * Rcat.ld, Rcat.ld, Rcat.p + (size_t)j * k * Rcat.ld, Rcat.ld, 0.0,
                 Gall.p + (size_t)j * k + (size_t)j * k * Gall.ld, Gall.ld, st, "gemm_gram");
    } else gemm(ctx, true, false, 1.0, Rcat, Rcat, 0.0, Gall, st, "gemm_gram");
    TimedScope ts(ctx, "ldlt_norm", 8.0 * g * k * k, 4.0 * g * (double)k * k * k);
    const int kp = (k + 31) & ~31;
    const size_t shm = 2 * (size_t)kp * kp * sizeof(double);
    lds_attr(ctx, (const void*)k_gram_norm_multi, 150 * 1024);
    hipLaunchKernelGGL(k_gram_norm_multi, dim3(1), dim3(1024), shm, ctx->stream, k, g, (const double*)Gall.p, Gall.ld, T.p, T.ld, tdiag ? 1 : 0, alpha, st, iters0);
    DRE_HIP(hipGetLastError());
}
void residual_norm_group_diag(Ctx* ctx, const Mat& Rcat, int g, int k, const Mat& T, bool tdiag, double alpha, AdiState* st, int iters0) {
    DRE_REQUIRE(g >= 1 && g <= 16 && Rcat.cols == g * k, "residual_norm_group_diag: shapes");
    // G_jj = R_j' R_j for the g residuals: one z-batched split-K product + one reduction of its slabs
    GemmZ gz; std::memset(&gz, 0, sizeof(gz));
    for (int j = 0; j < g; ++j) gz.A[j] = gz.B[j] = Rcat.p + (size_t)j * k * Rcat.ld;
    int zs = 1;
    BufP part = gemm_partials_z(ctx, true, false, k, k, Rcat.rows, gz, g, Rcat.ld, Rcat.ld, &zs, st, "gemm_gram");
    Mat Gd(ctx, k, g * k);
    gemm_reduce_z(ctx, k, k, zs, g, (const double*)part->p, nullptr, Gd.p, Gd.ld, (long)k * k, st);
    if (k <= 96) {
        TimedScope ts(ctx, "ldlt_norm", 8.0 * g * k * k, 4.0 * g * (double)k * k * k);
        const int kp = (k + 31) & ~31;
        const size_t shm = 2 * (size_t)kp * kp * sizeof(double);
        lds_attr(ctx, (const void*)k_gram_norm_z, 150 * 1024);
        hipLaunchKernelGGL(k_gram_norm_z, dim3(g), dim3(1024), shm, ctx->stream, k, g, (const double*)Gd.p, T.p, T.ld, tdiag ? 1 : 0, alpha, st, iters0);
        DRE_HIP(hipGetLastError());
        return;
    }
    if (tdiag) {
        for (int j = 0; j < g; ++j) { Mat Gj = Gd.colsview(j * k, k); ldlt_norm_update_state(ctx, Gj, T, true, alpha, st, iters0 + j + 1); }
        return;
    }
    // wide residual: one batched product T G_jj, one decision launch
    Mat TGall(ctx, k, g * k);
    std::vector<GemmBatchDesc> descs;
    for (int j = 0; j < g; ++j) descs.push_back({T.p, Gd.p + (size_t)j * k * Gd.ld, TGall.p + (size_t)j * k * TGall.ld, nullptr, 1.0, k, k, k, T.ld, Gd.ld, TGall.ld, 0});
    gemm_batched(ctx, descs, "gemm_norm");
    TimedScope ts(ctx, "ldlt_norm", 16.0 * g * k * k, 4.0 * g * k * k);
    hipLaunchKernelGGL(k_trace_sq_multi, dim3(g), dim3(512), 0, ctx->stream, k, g, (const double*)TGall.p, TGall.ld, alpha, st, iters0);
    DRE_HIP(hipGetLastError());
}
void residual_norm_step(Ctx* ctx, const Mat& R, const Mat& T, bool tdiag, double alpha, AdiState* st, int iters_after) {
    const int k = R.cols;
    if (k > 96) {
        Mat G(ctx, k, k);
        gemm(ctx, true, false, 1.0, R, R, 0.0, G, st, "gemm_gram");
        ldlt_norm_update_state(ctx, G, T, tdiag, alpha, st, iters_after);
        return;
    }
    int splits = 1;
    BufP part = gemm_partials(ctx, true, false, k, k, R.rows, R.p, R.ld, R.p, R.ld, &splits, st, "gemm_gram");
    if (splits > 12) {
        // tall R: dozens of k x k slabs would be summed by the single norm workgroup (6 MB at n = 5177); reduce them on many CUs first
        auto one = std::make_shared<Buf>(ctx, (size_t)k * k * sizeof(double));
        const size_t tot = (size_t)k * k;
        hipLaunchKernelGGL(k_gemm_reduce, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, ctx->stream, k, k, splits, 1.0, (const double*)part->p, 0.0,
                           (double*)one->p, k, (const AdiState*)st);
        part = one; splits = 1;
    }
    TimedScope ts(ctx, "ldlt_norm", 8.0 * splits * k * k, 4.0 * (double)k * k * k);
    const int kp = (k + 31) & ~31;
    const size_t shm = 2 * (size_t)kp * kp * sizeof(double);
    lds_attr(ctx, (const void*)k_gram_norm, 150 * 1024);
    hipLaunchKernelGGL(k_gram_norm, dim3(1), dim3(1024), shm, ctx->stream, k, splits, (const double*)part->p, T.p, T.ld, tdiag ? 1 : 0, alpha, st, iters_after);
}

double ldlt_norm_host(Ctx* ctx, const Mat& L, const Mat& D, double alpha) {
    if (L.cols == 0) return 0.0;
    Mat G(ctx, L.cols, L.cols);
    gemm(ctx, true, false, 1.0, L, L, 0.0, G, nullptr, "gemm_gram");
    DevArr<double> out(ctx, 1);
    Mat TG(ctx, G.rows, G.cols);
    gemm(ctx, false, false, 1.0, D, G, 0.0, TG, nullptr, "gemm_norm");
    trace_sq(ctx, TG, alpha, nullptr, 0, out.p);
    return read_scalar(ctx, out.p);
}

// |alpha| ||L D L'||_F through the Gram matrix into DEVICE memory (no synchronisation): the tolerance of the next Lyapunov solve is formed
// on the side stream while the main stream already iterates (gdre.hip, Rosenbrock-1 loop with the residual recurrence)
void ldlt_norm_device(Ctx* ctx, const Mat& L, const Mat& D, double alpha, double* out_dev) {
    if (L.cols == 0) { DRE_HIP(hipMemsetAsync(out_dev, 0, sizeof(double), ctx->stream)); return; }
    Mat G(ctx, L.cols, L.cols);
    gemm(ctx, true, false, 1.0, L, L, 0.0, G, nullptr, "gemm_gram");
    Mat TG(ctx, G.rows, G.cols);
    gemm(ctx, false, false, 1.0, D, G, 0.0, TG, nullptr, "gemm_norm");
    trace_sq(ctx, TG, alpha, nullptr, 0, out_dev);
}
// Deferred convergence decisions (adi.jl:115-123) for iterations 0 .. count whose norms were recorded while the tolerance was still being
// formed elsewhere: abstol = reltol * (*normC) (or abstol_given >= 0), then the first recorded norm at or below it ends the solve.
__global__ void k_decide_scan(AdiState* st, int count, const double* __restrict__ normC, double reltol, double abstol_given) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const double abstol = abstol_given >= 0.0 ? abstol_given : reltol * normC[0];
    st->abstol = abstol;
    int it = count;
    for (int i = 0; i <= count; ++i)
        if (st->norms[i & 511] <= abstol) { it = i; break; }
    st->iters = it;
    st->res_norm = st->norms[it & 511];
    st->done = (st->norms[it & 511] <= abstol || it >= st->maxiters) ? 1 : 0;
}
void adi_decide_scan(Ctx* ctx, AdiState* st, int count, const double* normC_dev, double reltol, double abstol_given) {
    hipLaunchKernelGGL(k_decide_scan, dim3(1), dim3(64), 0, ctx->stream, st, count, normC_dev, reltol, abstol_given);
    DRE_HIP(hipGetLastError());
}
// EV_j = (R_{j-1} - R_j) * inv2mu_j for up to 64 consecutive iterations whose residual factors lie side by side (Rs = [R_1 .. R_J], R_0 given
// for the first one): E'V_j of the residual recurrence R_j = R_{j-1} - 2 mu_j E'V_j (adi.jl:171) without touching E or V
struct EvScale { double inv2mu[64]; };
__global__ __launch_bounds__(256) void k_ev_from_residuals(int n, int k, int J, const double* __restrict__ R0, int ldr0, const double* __restrict__ Rs, int ldrs,
                                                           double* __restrict__ EV, int ldev, EvScale sc) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (size_t)n * k * J) return;
    const int i = idx % n; const size_t col = idx / n;
    const int j = (int)(col / k), c = (int)(col % k);
    const double prev = j == 0 ? R0[i + (size_t)c * ldr0] : Rs[i + (size_t)(col - k) * ldrs];
    EV[i + col * ldev] = (prev - Rs[i + col * ldrs]) * sc.inv2mu[j];
}
void ev_from_residuals(Ctx* ctx, int n, int k, int J, const Mat& R0, const Mat& Rs, Mat& EV, const double* mu) {
    for (int j0 = 0; j0 < J; j0 += 64) {
        const int jj = std::min(64, J - j0);
        EvScale sc;
        for (int j = 0; j < jj; ++j) sc.inv2mu[j] = 1.0 / (2.0 * mu[j0 + j]);
        const double* r0 = j0 == 0 ? R0.p : Rs.p + (size_t)(j0 - 1) * k * Rs.ld;
        const int ld0 = j0 == 0 ? R0.ld : Rs.ld;
        const size_t tot = (size_t)n * k * jj;
        hipLaunchKernelGGL(k_ev_from_residuals, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, ctx->stream, n, k, jj, r0, ld0,
                           (const double*)(Rs.p + (size_t)j0 * k * Rs.ld), Rs.ld, EV.p + (size_t)j0 * k * EV.ld, EV.ld, sc);
    }
    DRE_HIP(hipGetLastError());
}

}  // namespace dre
