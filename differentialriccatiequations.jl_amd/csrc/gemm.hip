// Dense f64 GEMM family for gfx950 (v_mfma_f64_16x16x4_f64): split-K GEMM, batched and z-batched products, the thin single-launch form.
#include "dense_device.hpp"
#include <functional>
#include "profiling.hpp"
#include <atomic>
#include <chrono>

namespace dre {

// =============================================================================================
// GEMM: 64x64 block tile, 4 waves (2x2), each wave 2x2 tiles of v_mfma_f64_16x16x4_f64, BK = 16.
// MFMA operand map (cdna_hip_programming.md §3): A: lane l holds A[l&15][l>>4]; B: B[l>>4][l&15];
// C/D (f64 form): col = lane&15, row = (lane>>4) + 4*reg.
// LDS rows are padded to 81 doubles so that both the k-fastest (transposed) and the m-fastest
// staging writes and the fragment reads stay (nearly) bank-conflict free.
// =============================================================================================
#define GB_M 64
#define GB_N 64
#define GB_K 32
#define GB_LD 81

// One 64 x 64 output tile over the K range [kbeg, kend): C = alpha A B + beta C, or (partial != nullptr) the raw
// product into the slab `partial` (M x N, ld M).
template <bool TA, bool TB>
__device__ __forceinline__ void gemm_tile(int M, int N, int K, double alpha, const double* __restrict__ A, int lda,
                                          const double* __restrict__ B, int ldb, double beta, double* __restrict__ C, int ldc,
                                          int m0, int n0, int kbeg, int kend, double* __restrict__ partial,
                                          double* __restrict__ tile_sumsq = nullptr) {
    __shared__ double As[GB_K][GB_LD];
    __shared__ double Bs[GB_K][GB_LD];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    v4d acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (v4d){0.0, 0.0, 0.0, 0.0};
    const int wm = (wave & 1) * 32, wn = (wave >> 1) * 32;
    const int lr = lane & 15, lk = lane >> 4;

    // Software pipeline: the 16 global loads of K-tile t+1 are issued right after tile t went to LDS, so their latency
    // overlaps the MFMAs of tile t (these GEMMs are bound by the memory round trip per K-tile, not by the matrix cores).
    double ra[8], rb[8];
    auto load_tile = [&](int k0) {
        if (!TA) {
            const int m = tid & 63, kq = tid >> 6, gm = m0 + m;
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                const int gk = k0 + kq + 4 * p;
                ra[p] = (gm < M && gk < kend) ? A[gm + (size_t)gk * lda] : 0.0;
            }
        } else {
            const int k = tid & 31, mq = tid >> 5, gk = k0 + k;
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                const int gm = m0 + mq + 8 * p;
                ra[p] = (gm < M && gk < kend) ? A[gk + (size_t)gm * lda] : 0.0;
            }
        }
        if (!TB) {
            const int k = tid & 31, nq = tid >> 5, gk = k0 + k;
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                const int gn = n0 + nq + 8 * p;
                rb[p] = (gn < N && gk < kend) ? B[gk + (size_t)gn * ldb] : 0.0;
            }
        } else {
            const int n = tid & 63, kq = tid >> 6, gn = n0 + n;
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                const int gk = k0 + kq + 4 * p;
                rb[p] = (gn < N && gk < kend) ? B[gn + (size_t)gk * ldb] : 0.0;
            }
        }
    };
    if (kbeg < kend) load_tile(kbeg);
    // beta != 0: the old C tile is requested now, its latency hides behind the K loop
    double cpre[2][2][4];
    const bool want_c = !partial && beta != 0.0;
    if (want_c) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = m0 + wm + i * 16 + (lane >> 4) + 4 * r, col = n0 + wn + j * 16 + (lane & 15);
                    cpre[i][j][r] = (row < M && col < N) ? C[row + (size_t)col * ldc] : 0.0;
                }
    }
    for (int k0 = kbeg; k0 < kend; k0 += GB_K) {
        if (!TA) { const int m = tid & 63, kq = tid >> 6;
#pragma unroll
            for (int p = 0; p < 8; ++p) As[kq + 4 * p][m] = ra[p];
        } else { const int k = tid & 31, mq = tid >> 5;
#pragma unroll
            for (int p = 0; p < 8; ++p) As[k][mq + 8 * p] = ra[p];
        }
        if (!TB) { const int k = tid & 31, nq = tid >> 5;
#pragma unroll
            for (int p = 0; p < 8; ++p) Bs[k][nq + 8 * p] = rb[p];
        } else { const int n = tid & 63, kq = tid >> 6;
#pragma unroll
            for (int p = 0; p < 8; ++p) Bs[kq + 4 * p][n] = rb[p];
        }
        __syncthreads();
        if (k0 + GB_K < kend) load_tile(k0 + GB_K);
#pragma unroll
        for (int kk = 0; kk < GB_K / 4; ++kk) {
            const int k = kk * 4 + lk;
            const double a0 = As[k][wm + lr], a1 = As[k][wm + 16 + lr];
            const double b0 = Bs[k][wn + lr], b1 = Bs[k][wn + 16 + lr];
            acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
        }
        __syncthreads();
    }
    double ssq = 0.0;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + wm + i * 16 + (lane >> 4) + 4 * r;
                const int col = n0 + wn + j * 16 + (lane & 15);
                if (row < M && col < N) {
                    if (partial) {
                        partial[row + (size_t)col * M] = acc[i][j][r];
                    } else {
                        double* c = C + row + (size_t)col * ldc;
                        const double v = (beta == 0.0) ? alpha * acc[i][j][r] : alpha * acc[i][j][r] + beta * cpre[i][j][r];
                        *c = v;
                        ssq += v * v;
                    }
                }
            }
    if (tile_sumsq) {
        // one partial per tile, summed wave by wave in a fixed order (As is free again after the last K-tile)
        ssq = wave_sum(ssq);
        if (lane == 0) As[0][wave] = ssq;
        __syncthreads();
        if (tid == 0) *tile_sumsq = (As[0][0] + As[0][1]) + (As[0][2] + As[0][3]);
    }
}


// (A 128 x 128-tile variant — 4 x 4 MFMA tiles per wave, 16 flop per byte staged through LDS — was built and measured in round 2:
// 45.1 TFLOP/s at 4096^3 against 47.9 for this kernel, 22-30 against 29-34 on the skinny passes of the randomized compression
// (tools/gemm_probe.py).  With one wave per SIMD its global round trips are not covered; the 64 x 64 tiles keep four workgroups per CU.
// It was removed again.)
// Workgroup -> output tile.  Workgroups are dealt round-robin over the 8 XCDs (blocks b and b + 8 share one; observed, a speed matter only), each
// with its own L2: in launch order the tiles that share an operand panel sit on DIFFERENT L2s and every XCD streams the large operand for
// itself (the 5 row tiles of a 304 x 6400 x 20209 sketch product: the 1 GB factor crosses the fabric five times).  swz != 0: every XCD takes
// a CONTIGUOUS chunk of the tile list (bijective remap for any grid size), and the list runs fastest along the dimension with FEWER tiles,
// so the tiles that are resident together on one XCD share the panels of the large operand; splits slowest (they share nothing).
__device__ __forceinline__ void xcd_tile(int swz, int& bx, int& by, int& bz) {
    if (!swz) { bx = blockIdx.x; by = blockIdx.y; bz = blockIdx.z; return; }
    const unsigned gx = gridDim.x, gy = gridDim.y, per = gx * gy, T = per * gridDim.z;
    unsigned L = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
    const unsigned xcd = L & 7u, slot = L >> 3, q = T >> 3, r = T & 7u;
    L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    const unsigned z = L / per, rem = L - z * per;
    bz = (int)z;
    if (gx <= gy) { const unsigned y = rem / gx; by = (int)y; bx = (int)(rem - y * gx); }
    else { const unsigned x = rem / gy; bx = (int)x; by = (int)(rem - x * gy); }
}
// Used where the larger operand does not fit the 256 MB Infinity Cache (measured, tools/gemm_probe.py: 304 x 6400 x 20209 TN 39.1 -> 42.4 TFLOP/s,
// 20209 x 304 x 6400 NT 35.8 -> 39.2, 20209 x 304 x 3500 NT 34.3 -> 38.1; cache-resident operands: within +-2 %, 2976 x 112 x 2976 7 % slower).
static inline int gemm_swizzle(const Ctx* ctx, int M, int N, int K) {
    return ctx->gemm_swizzle != 0 && 8.0 * (double)K * (double)std::max(M, N) > 256.0 * 1048576.0 ? 1 : 0;
}
template <bool TA, bool TB>
__global__ __launch_bounds__(256) void k_gemm(int M, int N, int K, double alpha, const double* __restrict__ A,
                                              int lda, const double* __restrict__ B, int ldb, double beta,
                                              double* __restrict__ C, int ldc, int kchunk,
                                              double* __restrict__ partial, const AdiState* st, double* __restrict__ tile_sumsq, DevCount dc, int swz) {
    if (st && st->done) return;
    if (dc.st) K = min(K, dc.per * dev_count(dc));          // inner dimension decided on the device (accepted ADI iterations x columns)
    int bx, by, bz;
    xcd_tile(swz, bx, by, bz);
    const int kbeg = bz * kchunk;
    gemm_tile<TA, TB>(M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, bx * GB_M, by * GB_N, kbeg, min(K, kbeg + kchunk),
                      partial ? partial + (size_t)bz * M * N : nullptr,
                      tile_sumsq ? tile_sumsq + bx + (size_t)gridDim.x * by : nullptr);
}

// Batched NN GEMM with per-batch operands (blockIdx.z = batch): C_z = alpha_z A_z B_z; optionally A_z is also copied to
// copy_dst_z (the column-concatenation of LDL' blocks comes for free with the product by the block-diagonal factor).
__global__ __launch_bounds__(256) void k_gemm_batched(const GemmBatchDesc* __restrict__ descs) {
    const GemmBatchDesc d = descs[blockIdx.z];
    const int m0 = blockIdx.x * GB_M, n0 = blockIdx.y * GB_N;
    if (m0 >= d.M) return;
    if (d.copy_dst && blockIdx.y == 0) {
        for (int id = threadIdx.x; id < GB_M * d.K; id += blockDim.x) {
            const int r = m0 + id % GB_M, c = id / GB_M;
            if (r < d.M) d.copy_dst[r + (size_t)c * d.ldcopy] = d.A[r + (size_t)c * d.lda];
        }
    }
    if (n0 >= d.N) return;
    gemm_tile<false, false>(d.M, d.N, d.K, d.alpha, d.A, d.lda, d.B, d.ldb, 0.0, d.C, d.ldc, m0, n0, 0, d.K, nullptr);
}
// up to 48 products: the descriptors travel as kernel arguments (no upload, no staging copy on the host)
struct GemmBatchArgs { GemmBatchDesc d[48]; DevCount dc; };
__global__ __launch_bounds__(256) void k_gemm_batched_args(GemmBatchArgs a) {
    if (a.dc.st && (int)blockIdx.z >= dev_count(a.dc)) return;       // products beyond the device-side count are not formed
    const GemmBatchDesc d = a.d[blockIdx.z];
    const int m0 = blockIdx.x * GB_M, n0 = blockIdx.y * GB_N;
    if (m0 >= d.M) return;
    if (d.copy_dst && blockIdx.y == 0) {
        for (int id = threadIdx.x; id < GB_M * d.K; id += blockDim.x) {
            const int r = m0 + id % GB_M, c = id / GB_M;
            if (r < d.M) d.copy_dst[r + (size_t)c * d.ldcopy] = d.A[r + (size_t)c * d.lda];
        }
    }
    if (n0 >= d.N) return;
    gemm_tile<false, false>(d.M, d.N, d.K, d.alpha, d.A, d.lda, d.B, d.ldb, 0.0, d.C, d.ldc, m0, n0, 0, d.K, nullptr);
}
void gemm_batched(Ctx* ctx, const std::vector<GemmBatchDesc>& descs, const char* tag, DevCount dc) {
    DRE_REQUIRE(!dc.st || descs.size() <= 48, "gemm_batched: a device-side count needs at most 48 products");
    if (!descs.empty() && descs.size() <= 48) {
        int maxM = 0, maxN = 0; double fl = 0.0, by = 0.0;
        GemmBatchArgs a;
        a.dc = dc;
        for (size_t i = 0; i < descs.size(); ++i) {
            const auto& d = descs[i];
            a.d[i] = d;
            maxM = std::max(maxM, d.M); maxN = std::max(maxN, d.N);
            fl += 2.0 * d.M * d.N * (double)d.K; by += 8.0 * ((double)d.M * d.K * (d.copy_dst ? 2.0 : 1.0) + (double)d.K * d.N + (double)d.M * d.N);
        }
        TimedScope ts(ctx, tag, by, fl);
        hipLaunchKernelGGL(k_gemm_batched_args, dim3(ceil_div(maxM, GB_M), std::max(1, ceil_div(maxN, GB_N)), (unsigned)descs.size()), dim3(256), 0, ctx->stream, a);
        DRE_HIP(hipGetLastError());
        return;
    }
    if (descs.empty()) return;
    int maxM = 0, maxN = 0; double fl = 0.0, by = 0.0;
    for (auto& d : descs) {
        maxM = std::max(maxM, d.M); maxN = std::max(maxN, d.N);
        fl += 2.0 * d.M * d.N * (double)d.K; by += 8.0 * ((double)d.M * d.K * (d.copy_dst ? 2.0 : 1.0) + (double)d.K * d.N + (double)d.M * d.N);
    }
    DevArr<GemmBatchDesc> dd(ctx, descs.size());
    DRE_HIP(hipMemcpyAsync(dd.p, descs.data(), descs.size() * sizeof(GemmBatchDesc), hipMemcpyHostToDevice, ctx->stream));
    TimedScope ts(ctx, tag, by, fl);
    hipLaunchKernelGGL(k_gemm_batched, dim3(ceil_div(maxM, GB_M), std::max(1, ceil_div(maxN, GB_N)), (unsigned)descs.size()), dim3(256), 0, ctx->stream,
                       (const GemmBatchDesc*)dd.p);
    DRE_HIP(hipGetLastError());
}

// fixed-order sum of split-K slabs written to C[rowmap[row], col] (the scatter of a gathered sub-system rides on the reduction)
__global__ void k_gemm_reduce_rows(int M, int N, int splits, const double* __restrict__ partial, const int* __restrict__ rowmap, double* __restrict__ C, int ldc,
                                   const AdiState* st) {
    if (st && st->done) return;
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)M * N) return;
    const size_t slab = (size_t)M * N;
    double s = 0.0;
    int z = 0;
    for (; z + 7 < splits; z += 8) {
        double q[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) q[u] = partial[(z + u) * slab + idx];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += q[u];
    }
    for (; z < splits; ++z) s += partial[z * slab + idx];
    const int row = idx % M, col = idx / M;
    C[rowmap[row] + (size_t)col * ldc] = s;
}
void gemm_reduce_rows(Ctx* ctx, int M, int N, int splits, const double* partial, const int* rowmap, double* C, int ldc, const AdiState* st) {
    const size_t tot = (size_t)M * N;
    if (!tot) return;
    hipLaunchKernelGGL(k_gemm_reduce_rows, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, ctx->stream, M, N, splits, partial, rowmap, C, ldc, st);
    DRE_HIP(hipGetLastError());
}
__global__ void k_gemm_reduce(int M, int N, int splits, double alpha, const double* __restrict__ partial,
                              double beta, double* __restrict__ C, int ldc, const AdiState* st) {
    if (st && st->done) return;
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)M * N) return;
    // fixed summation order (deterministic); four slab loads in flight at a time
    const size_t slab = (size_t)M * N;
    double s = 0.0;
    int z = 0;
    for (; z + 7 < splits; z += 8) {
        double q[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) q[u] = partial[(z + u) * slab + idx];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += q[u];
    }
    for (; z + 3 < splits; z += 4) {
        const double p0 = partial[z * slab + idx], p1 = partial[(z + 1) * slab + idx], p2 = partial[(z + 2) * slab + idx], p3 = partial[(z + 3) * slab + idx];
        s = (((s + p0) + p1) + p2) + p3;
    }
    for (; z < splits; ++z) s += partial[z * slab + idx];
    int row = idx % M, col = idx / M;
    double* c = C + row + (size_t)col * ldc;
    *c = (beta == 0.0) ? alpha * s : alpha * s + beta * (*c);
}

// (A tall-skinny  C = alpha A'B  kernel — every wave walks groups of 16 rows, up to 4 x 4 accumulator tiles, one partial per wave, no LDS — was
// built in round 3 for the Gram matrices / U'W / V'Z products and measured NOT faster than the split-K GEMM below: Gram of a 5177 x 64 factor
// 21 us with either.  Removed in round 4; CHANGELOG.)
void gemm(Ctx* ctx, bool tA, bool tB, int M, int N, int K, double alpha, const double* A, int lda, const double* B,
          int ldb, double beta, double* C, int ldc, const AdiState* st, const char* tag, double* tile_sumsq) {
    if (M <= 0 || N <= 0) return;
    TimedScope ts(ctx, tag, 8.0 * ((double)M * K + (double)K * N + 2.0 * M * N), 2.0 * M * N * (double)K);
    const int tm = ceil_div(M, GB_M), tn = ceil_div(N, GB_N);
    // These GEMMs are latency bound (one memory round trip per K-tile), so K is split until the grid fills the
    // chip or every block is down to two K-tiles; partial slabs are reduced in a fixed order (deterministic).
    DRE_REQUIRE(!tile_sumsq || K <= 2 * GB_K, "gemm: tile_sumsq needs an unsplit K");
    int splits = 1;
    if (K > 2 * GB_K) {
        int want = ceil_div(2 * ctx->num_cus, tm * tn);
        splits = std::max(1, std::min(want, K / (2 * GB_K)));
    }
    int kchunk = K > 0 ? ceil_div(ceil_div(K, splits), GB_K) * GB_K : GB_K;
    splits = K > 0 ? ceil_div(K, kchunk) : 1;
    dim3 grid(tm, tn, splits), block(256);
    BufP pb;
    double* partial = nullptr;
    if (splits > 1) {
        pb = std::make_shared<Buf>(ctx, (size_t)splits * M * N * sizeof(double));
        partial = (double*)pb->p;
    }
    if (!tA && !tB) hipLaunchKernelGGL((k_gemm<false, false>), grid, block, 0, ctx->stream, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, kchunk, partial, st, tile_sumsq, DevCount{}, gemm_swizzle(ctx, M, N, K));
    else if (tA && !tB) hipLaunchKernelGGL((k_gemm<true, false>), grid, block, 0, ctx->stream, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, kchunk, partial, st, tile_sumsq, DevCount{}, gemm_swizzle(ctx, M, N, K));
    else if (!tA && tB) hipLaunchKernelGGL((k_gemm<false, true>), grid, block, 0, ctx->stream, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, kchunk, partial, st, tile_sumsq, DevCount{}, gemm_swizzle(ctx, M, N, K));
    else hipLaunchKernelGGL((k_gemm<true, true>), grid, block, 0, ctx->stream, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, kchunk, partial, st, tile_sumsq, DevCount{}, gemm_swizzle(ctx, M, N, K));
    if (splits > 1) {
        size_t tot = (size_t)M * N;
        hipLaunchKernelGGL(k_gemm_reduce, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, ctx->stream, M, N, splits, alpha, partial, beta, C, ldc, st);
    }
    DRE_HIP(hipGetLastError());
}

// Small outputs with a long inner dimension in ONE launch: one workgroup per 16 x 16 tile of C = alpha op(A) B + beta C, the four waves split K and
// meet in LDS (fixed order) — the split-K GEMM above needs a second launch to sum its slabs, and at a few dozen tiles both are pure latency.
template <bool TA>
__global__ __launch_bounds__(256) void k_gemm_thin(int M, int N, int K, double alpha, const double* __restrict__ A, int lda, const double* __restrict__ B, int ldb,
                                                   double beta, double* __restrict__ C, int ldc, const AdiState* st) {
    const int done_flag = st ? st->done : 0;
    __shared__ double part[4][4][64];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lr = lane & 15, lk = lane >> 4;
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const int i0 = blockIdx.x * 16, j0 = blockIdx.y * 16;
    const int ai = min(i0 + lr, M - 1), bj = min(j0 + lr, N - 1);
    const bool aok = i0 + lr < M, bok = j0 + lr < N;
    const int kst = (K + 3) >> 2, per = (kst + 3) >> 2, t0 = wv * per, t1 = min(kst, t0 + per);
    v4d acc = (v4d){0.0, 0.0, 0.0, 0.0};
    for (int tb = t0; tb < t1; tb += 24) {
        double av[24], bv[24];
#pragma unroll
        for (int u = 0; u < 24; ++u) {
            const int t = min(tb + u, t1 - 1), kk = min(4 * t + lk, K - 1);
            av[u] = TA ? A[kk + (size_t)ai * lda] : A[ai + (size_t)kk * lda];
            bv[u] = B[kk + (size_t)bj * ldb];
        }
#pragma unroll
        for (int u = 0; u < 24; ++u) {
            const bool ok = (tb + u < t1) && 4 * (tb + u) + lk < K;
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64((ok && aok) ? av[u] : 0.0, (ok && bok) ? bv[u] : 0.0, acc, 0, 0, 0);
        }
    }
    if (done_flag) return;
#pragma unroll
    for (int r = 0; r < 4; ++r) part[wave][r][lane] = acc[r];
    __syncthreads();
    const double v = ((part[0][wave][lane] + part[1][wave][lane]) + part[2][wave][lane]) + part[3][wave][lane];
    const int row = i0 + lk + 4 * wave, col = j0 + lr;
    if (row < M && col < N) {
        double* c = C + row + (size_t)col * ldc;
        *c = (beta == 0.0) ? alpha * v : alpha * v + beta * (*c);
    }
}
void gemm_thin(Ctx* ctx, bool tA, int M, int N, int K, double alpha, const double* A, int lda, const double* B, int ldb, double beta, double* C, int ldc,
               const AdiState* st, const char* tag) {
    if (M <= 0 || N <= 0) return;
    DRE_REQUIRE(K >= 1, "gemm_thin: empty inner dimension");
    TimedScope ts(ctx, tag, 8.0 * ((double)M * K + (double)K * N + 2.0 * M * N), 2.0 * M * N * (double)K);
    const dim3 grid(ceil_div(M, 16), ceil_div(N, 16));
    if (tA) hipLaunchKernelGGL((k_gemm_thin<true>), grid, dim3(256), 0, ctx->stream, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, st);
    else hipLaunchKernelGGL((k_gemm_thin<false>), grid, dim3(256), 0, ctx->stream, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, st);
    DRE_HIP(hipGetLastError());
}

BufP gemm_partials(Ctx* ctx, bool tA, bool tB, int M, int N, int K, const double* A, int lda, const double* B, int ldb,
                   int* splits_out, const AdiState* st, const char* tag, DevCount dc) {
    TimedScope ts(ctx, tag, 8.0 * ((double)M * K + (double)K * N + 2.0 * M * N), 2.0 * M * N * (double)K);
    const int tm = ceil_div(M, GB_M), tn = ceil_div(N, GB_N);
    int splits = 1;
    if (K > 2 * GB_K) {
        int want = ceil_div(2 * ctx->num_cus, tm * tn);
        splits = std::max(1, std::min(want, K / (2 * GB_K)));
    }
    int kchunk = K > 0 ? ceil_div(ceil_div(K, splits), GB_K) * GB_K : GB_K;
    splits = K > 0 ? ceil_div(K, kchunk) : 1;
    dim3 grid(tm, tn, splits), block(256);
    auto pb = std::make_shared<Buf>(ctx, (size_t)splits * M * N * sizeof(double));
    double* partial = (double*)pb->p;
    double* none = nullptr;
    if (!tA && !tB) hipLaunchKernelGGL((k_gemm<false, false>), grid, block, 0, ctx->stream, M, N, K, 1.0, A, lda, B, ldb, 0.0, none, 0, kchunk, partial, st, (double*)nullptr, dc, gemm_swizzle(ctx, M, N, K));
    else if (tA && !tB) hipLaunchKernelGGL((k_gemm<true, false>), grid, block, 0, ctx->stream, M, N, K, 1.0, A, lda, B, ldb, 0.0, none, 0, kchunk, partial, st, (double*)nullptr, dc, gemm_swizzle(ctx, M, N, K));
    else if (!tA && tB) hipLaunchKernelGGL((k_gemm<false, true>), grid, block, 0, ctx->stream, M, N, K, 1.0, A, lda, B, ldb, 0.0, none, 0, kchunk, partial, st, (double*)nullptr, dc, gemm_swizzle(ctx, M, N, K));
    else hipLaunchKernelGGL((k_gemm<true, true>), grid, block, 0, ctx->stream, M, N, K, 1.0, A, lda, B, ldb, 0.0, none, 0, kchunk, partial, st, (double*)nullptr, dc, gemm_swizzle(ctx, M, N, K));
    DRE_HIP(hipGetLastError());
    *splits_out = splits;
    return pb;
}

// ---------------------------------------------------------------------------------------------
// z-batched split-K products (round 4): nz products C_z = op(A_z) op(B_z) of one shape in ONE launch — the g solves of a fan group
// (engine.hip) share every launch instead of running side by side on g streams.  Slab (z, split) lies at partial + (z * splits + split) M N.
// ---------------------------------------------------------------------------------------------
template <bool TA, bool TB>
__global__ __launch_bounds__(256) void k_gemm_z(int M, int N, int K, GemmZ zb, int lda, int ldb, int kchunk, int splits, double* __restrict__ partial,
                                                const AdiState* st) {
    if (st && st->done) return;
    const int z = blockIdx.z / splits, sp = blockIdx.z - z * splits;
    const int kbeg = sp * kchunk;
    gemm_tile<TA, TB>(M, N, K, 1.0, zb.A[z], lda, zb.B[z], ldb, 0.0, nullptr, 0, blockIdx.x * GB_M, blockIdx.y * GB_N, kbeg, min(K, kbeg + kchunk),
                      partial + (size_t)blockIdx.z * M * N);
}
BufP gemm_partials_z(Ctx* ctx, bool tA, bool tB, int M, int N, int K, const GemmZ& zb, int nz, int lda, int ldb, int* splits_out, const AdiState* st,
                     const char* tag) {
    DRE_REQUIRE(nz >= 1 && nz <= MF_ZMAX, "gemm_partials_z: batch size");
    TimedScope ts(ctx, tag, 8.0 * nz * ((double)M * K + (double)K * N + 2.0 * M * N), 2.0 * nz * M * N * (double)K);
    const int tm = ceil_div(M, GB_M), tn = ceil_div(N, GB_N);
    int splits = 1;
    if (K > 2 * GB_K) {
        int want = ceil_div(2 * ctx->num_cus, tm * tn * nz);
        splits = std::max(1, std::min(want, K / (2 * GB_K)));
    }
    int kchunk = K > 0 ? ceil_div(ceil_div(K, splits), GB_K) * GB_K : GB_K;
    splits = K > 0 ? ceil_div(K, kchunk) : 1;
    dim3 grid(tm, tn, splits * nz), block(256);
    auto pb = std::make_shared<Buf>(ctx, (size_t)splits * nz * M * N * sizeof(double));
    double* partial = (double*)pb->p;
    if (!tA && !tB) hipLaunchKernelGGL((k_gemm_z<false, false>), grid, block, 0, ctx->stream, M, N, K, zb, lda, ldb, kchunk, splits, partial, st);
    else if (tA && !tB) hipLaunchKernelGGL((k_gemm_z<true, false>), grid, block, 0, ctx->stream, M, N, K, zb, lda, ldb, kchunk, splits, partial, st);
    else DRE_REQUIRE(false, "gemm_partials_z: only NN and TN products");
    DRE_HIP(hipGetLastError());
    *splits_out = splits;
    return pb;
}
// fixed-order sums of the slabs of gemm_partials_z: C_z[rowmap ? rowmap[row] : row, col] = sum_split slab(z, split)[row, col],  C_z = C + z cz
__global__ void k_gemm_reduce_z(int M, int N, int splits, const double* __restrict__ partial, const int* __restrict__ rowmap, double* __restrict__ C, int ldc,
                                long cz, const AdiState* st) {
    if (st && st->done) return;
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)M * N) return;
    const size_t slab = (size_t)M * N;
    const double* __restrict__ p = partial + (size_t)blockIdx.y * splits * slab + idx;
    double s = 0.0;
    int z = 0;
    for (; z + 7 < splits; z += 8) {
        double q[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) q[u] = p[(z + u) * slab];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += q[u];
    }
    for (; z < splits; ++z) s += p[z * slab];
    const int row = idx % M, col = idx / M;
    C[(size_t)blockIdx.y * cz + (rowmap ? rowmap[row] : row) + (size_t)col * ldc] = s;
}
void gemm_reduce_z(Ctx* ctx, int M, int N, int splits, int nz, const double* partial, const int* rowmap, double* C, int ldc, long cz, const AdiState* st) {
    const size_t tot = (size_t)M * N;
    if (!tot) return;
    hipLaunchKernelGGL(k_gemm_reduce_z, dim3((unsigned)((tot + 255) / 256), nz), dim3(256), 0, ctx->stream, M, N, splits, partial, rowmap, C, ldc, cz, st);
    DRE_HIP(hipGetLastError());
}

}  // namespace dre
