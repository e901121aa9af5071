// Dense f64 kernels for gfx950: MFMA GEMM, Householder QR, symmetric eigensolver, small helpers.
#include "dense.hpp"
#include <functional>
#include "profiling.hpp"
#include <atomic>
#include <chrono>

namespace dre {
// Termination tolerance of the band reductions.  st->abstol > 0: absolute; <= 0: relative, tolfac * eps * ||S||_F.  Floor mode
// (st->maxiters == BAND_TOL_FLOOR, set by k_band_init / lr_band_reduce): max(relative, st->abstol) — the caller's estimate of the rounding
// noise with which S was FORMED (sums with cancellation: ||S|| << ||L||^2 ||D||, where the relative tolerance alone would keep the noise
// as signal; ldlt.hip, ldlt_compress COMPRESS_NOISE_FLOOR).
#define BAND_TOL_FLOOR 0x7F100D
__device__ inline double band_tol(const AdiState* st, double tolfac, double base) {
    const double rel = tolfac * 2.220446049250313e-16 * sqrt(base), a = st->abstol;
    if (st->maxiters == BAND_TOL_FLOOR) return fmax(rel, a);
    return a > 0.0 ? a : rel;
}

typedef double v4d __attribute__((ext_vector_type(4)));

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

// Wave-wide sum through DPP row shifts / row broadcasts (no LDS crossbar round trips as with ds_bpermute shuffles);
// the total lands in lane 63 and is broadcast through a scalar register.  Invalid source lanes contribute 0.
template <int CTRL, int ROW_MASK = 0xf>
__device__ inline double dpp_mov0(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ inline double wave_sum(double v) {
    v += dpp_mov0<0x111>(v);            // row_shr:1
    v += dpp_mov0<0x112>(v);            // row_shr:2
    v += dpp_mov0<0x114>(v);            // row_shr:4
    v += dpp_mov0<0x118>(v);            // row_shr:8   -> lane 15 of every row holds the row total
    v += dpp_mov0<0x142, 0xa>(v);       // row_bcast:15 into rows 1 and 3
    v += dpp_mov0<0x143, 0xc>(v);       // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave total
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63), hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
    return __hiloint2double(hi, lo);
}
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
template <bool TA, bool TB>
__global__ __launch_bounds__(256) void k_gemm(int M, int N, int K, double alpha, const double* __restrict__ A,
                                              int lda, const double* __restrict__ B, int ldb, double beta,
                                              double* __restrict__ C, int ldc, int kchunk,
                                              double* __restrict__ partial, const AdiState* st, double* __restrict__ tile_sumsq, DevCount dc) {
    if (st && st->done) return;
    if (dc.st) K = min(K, dc.per * dev_count(dc));          // inner dimension decided on the device (accepted ADI iterations x columns)
    const int kbeg = blockIdx.z * kchunk;
    gemm_tile<TA, TB>(M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, blockIdx.x * GB_M, blockIdx.y * GB_N, kbeg, min(K, kbeg + kchunk),
                      partial ? partial + (size_t)blockIdx.z * M * N : nullptr,
                      tile_sumsq ? tile_sumsq + blockIdx.x + (size_t)gridDim.x * blockIdx.y : nullptr);
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
    if (!tA && !tB) hipLaunchKernelGGL((k_gemm<false, false>), grid, block, 0, ctx->stream, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, kchunk, partial, st, tile_sumsq, DevCount{});
    else if (tA && !tB) hipLaunchKernelGGL((k_gemm<true, false>), grid, block, 0, ctx->stream, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, kchunk, partial, st, tile_sumsq, DevCount{});
    else if (!tA && tB) hipLaunchKernelGGL((k_gemm<false, true>), grid, block, 0, ctx->stream, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, kchunk, partial, st, tile_sumsq, DevCount{});
    else hipLaunchKernelGGL((k_gemm<true, true>), grid, block, 0, ctx->stream, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, kchunk, partial, st, tile_sumsq, DevCount{});
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
    if (!tA && !tB) hipLaunchKernelGGL((k_gemm<false, false>), grid, block, 0, ctx->stream, M, N, K, 1.0, A, lda, B, ldb, 0.0, none, 0, kchunk, partial, st, (double*)nullptr, dc);
    else if (tA && !tB) hipLaunchKernelGGL((k_gemm<true, false>), grid, block, 0, ctx->stream, M, N, K, 1.0, A, lda, B, ldb, 0.0, none, 0, kchunk, partial, st, (double*)nullptr, dc);
    else if (!tA && tB) hipLaunchKernelGGL((k_gemm<false, true>), grid, block, 0, ctx->stream, M, N, K, 1.0, A, lda, B, ldb, 0.0, none, 0, kchunk, partial, st, (double*)nullptr, dc);
    else hipLaunchKernelGGL((k_gemm<true, true>), grid, block, 0, ctx->stream, M, N, K, 1.0, A, lda, B, ldb, 0.0, none, 0, kchunk, partial, st, (double*)nullptr, dc);
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

// block-wide sum, result valid in every thread; blockDim.x multiple of 64, <= 1024
__device__ inline double block_sum(double v, double* red /* >= 17 doubles */) {
    v = wave_sum(v);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int w = 0; w < nw; ++w) s += red[w];
        red[16] = s;
    }
    __syncthreads();
    return red[16];
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
// the first residual at or below abstol ends the loop (fan groups, residual wider than 96 columns)
__global__ __launch_bounds__(1024) void k_trace_sq_multi(int k, int g, const double* __restrict__ TGall, int ldm, double alpha, AdiState* st, int iters0) {
    __shared__ double red[17];
    __shared__ int stop;
    if (st->done) return;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    for (int j = 0; j < g; ++j) {
        const double* __restrict__ M = TGall + (size_t)j * k * ldm;
        double s = 0.0;
        for (int c = wave; c < k; c += nw)
            for (int r = lane; r < k; r += 64) s += M[r + (size_t)c * ldm] * M[c + (size_t)r * ldm];
        s = block_sum(s, red);
        if (threadIdx.x == 0) {
            const double nrm = fabs(alpha) * sqrt(fmax(s, 0.0));
            const int iters_after = iters0 + j + 1;
            st->res_norm = nrm;
            st->iters = iters_after;
            st->norms[iters_after & 511] = nrm;
            const int d = (nrm <= st->abstol || iters_after >= st->maxiters) ? 1 : 0;
            if (d) st->done = 1;
            stop = d;
        }
        __syncthreads();
        if (stop) return;
        __syncthreads();
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
        hipLaunchKernelGGL(k_trace_sq_multi, dim3(1), dim3(1024), 0, ctx->stream, k, g, (const double*)TGall.p, TGall.ld, alpha, st, iters0);
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
    hipLaunchKernelGGL(k_trace_sq_multi, dim3(1), dim3(1024), 0, ctx->stream, k, g, (const double*)TGall.p, TGall.ld, alpha, st, iters0);
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

// =============================================================================================
// Blocked Householder QR (compact WY), panel width 16.
// =============================================================================================
#define QR_NB 16

__global__ void k_band_decide(int k, int nparts, const double* __restrict__ part, double tolfac, AdiState* st);

#ifdef DRE_PANEL_PROBE
__device__ long long g_probe[64];
#define PROBE(i) do { if (threadIdx.x == 0 && blockIdx.x == 0) g_probe[i] = clock64(); } while (0)
#define PROBEW(i) do { if (jj == 3 && wave == 4 && lane == 0 && blockIdx.x == 0) g_probe[i] = clock64(); } while (0)
#else
#define PROBE(i) do { } while (0)
#define PROBEW(i) do { } while (0)
#endif

struct PanelShared {
    double red[17];
    double Tsh[QR_NB][QR_NB + 1];
    double Zm[QR_NB][QR_NB + 1];   // Zm[i][j] = v_i' v_j  (i < j), input of the T recurrence
    double scl[QR_NB];             // deferred scaling: v_jj = x_jj * scl[jj] below the diagonal
    double taus[QR_NB];
    double betas[QR_NB];
    double nrm2[QR_NB + 1];        // nrm2[j] = ||P[j+1:, j]||^2 once reflectors 0..j-1 are applied (lookahead)
    double pv[2][1024];            // register-resident core: the current / next pivot column (double buffered)
};

// Householder QR of the rows x jb panel Pn (leading dimension ldp) by one workgroup, in place: on exit the upper triangle
// holds R, the entries below the diagonal the reflector vectors and sh.Tsh the block-reflector factor T.
// ONE barrier per column: every thread derives (tau, beta, scale) of column jj redundantly from the lookahead norm; then
// waves jj+1.. apply H_jj to the later columns (the wave of column jj+1 also accumulates the next norm), waves 0..jj-1
// compute the dot products v_i' v_jj for T, and the otherwise idle wave jj advances the T recurrence by one column.
// Diagonal entries (beta) and the scaling of the reflectors are written after the loop.
__device__ __forceinline__ void hh_panel_core(double* __restrict__ Pn, int ldp, int rows, int jb, PanelShared& sh) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, nw = blockDim.x >> 6;
    double (*Tsh)[QR_NB + 1] = sh.Tsh;
    double (*Zm)[QR_NB + 1] = sh.Zm;
    double* scl = sh.scl; double* red = sh.red;
    for (int i = tid; i < QR_NB * (QR_NB + 1); i += blockDim.x) { (&Tsh[0][0])[i] = 0.0; (&Zm[0][0])[i] = 0.0; }
    __syncthreads();                 // the LDS copy of the panel is complete
    {
        double s0 = 0.0;
        for (int i = 1 + tid; i < rows; i += blockDim.x) s0 += Pn[i] * Pn[i];
        s0 = block_sum(s0, red);
        if (tid == 0) sh.nrm2[0] = s0;
    }
    __syncthreads();
    PROBE(3);
    // T recurrence for column c (all earlier columns of T final):  T(0:c, c) = -tau_c T(0:c, 0:c) Zm(0:c, c)
    auto t_column = [&](int c) {
        const double tc = sh.taus[c];
        // four independent FMA chains (the chain of the plain recurrence is the long pole of the late columns)
        if (lane < c) {
            double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
            int l = lane;
            for (; l + 3 < c; l += 4) {
                a0 += Tsh[lane][l] * Zm[l][c];
                a1 += Tsh[lane][l + 1] * Zm[l + 1][c];
                a2 += Tsh[lane][l + 2] * Zm[l + 2][c];
                a3 += Tsh[lane][l + 3] * Zm[l + 3][c];
            }
            for (; l < c; ++l) a0 += Tsh[lane][l] * Zm[l][c];
            Tsh[lane][c] = -tc * ((a0 + a1) + (a2 + a3));
        }
        if (lane == 0) Tsh[c][c] = tc;
    };
    for (int jj = 0; jj < jb; ++jj) {
        double* col = Pn + (size_t)jj * ldp;        // local column jj, pivot at local row jj (entries below are UNSCALED x)
        PROBEW(10);
        const double s = sh.nrm2[jj], alpha = col[jj];
        double tau = 0.0, beta = alpha, scale = 0.0;
        if (s > 0.0) {
            const double nrm = sqrt(alpha * alpha + s);
            beta = alpha >= 0.0 ? -nrm : nrm;
            tau = (beta - alpha) / beta;
            scale = 1.0 / (alpha - beta);
        }
        if (tid == 0) { scl[jj] = scale; sh.taus[jj] = tau; sh.betas[jj] = beta; }
        PROBEW(11);
        for (int j = wave; j < jb; j += nw) {
            if (j > jj) {
                double* cj = Pn + (size_t)j * ldp;
                const double cjj = cj[jj];           // entry in the pivot row
                double w = 0.0;
                {   // four row strips in flight: the loop is bound by LDS latency, not by bandwidth
                    double w1 = 0.0, w2 = 0.0, w3 = 0.0;
                    int i = jj + 1 + lane;
#pragma unroll 1
                    for (; i + 192 < rows; i += 256) {
                        w += col[i] * cj[i]; w1 += col[i + 64] * cj[i + 64]; w2 += col[i + 128] * cj[i + 128]; w3 += col[i + 192] * cj[i + 192];
                    }
#pragma unroll 1
                    for (; i < rows; i += 64) w += col[i] * cj[i];
                    w = (w + w1) + (w2 + w3);
                }
                PROBEW(12);
                w = wave_sum(w) * scale + cjj;
                const double tw = tau * w, tws = tw * scale;
                PROBEW(13);
                double nn = 0.0;
                {
                    double n1 = 0.0, n2 = 0.0, n3 = 0.0;
                    int i = jj + 1 + lane;
#pragma unroll 1
                    for (; i + 192 < rows; i += 256) {
                        const double x0 = cj[i] - tws * col[i], x1 = cj[i + 64] - tws * col[i + 64];
                        const double x2 = cj[i + 128] - tws * col[i + 128], x3 = cj[i + 192] - tws * col[i + 192];
                        cj[i] = x0; cj[i + 64] = x1; cj[i + 128] = x2; cj[i + 192] = x3;
                        if (i > jj + 1) nn += x0 * x0;
                        n1 += x1 * x1; n2 += x2 * x2; n3 += x3 * x3;
                    }
#pragma unroll 1
                    for (; i < rows; i += 64) {
                        const double x = cj[i] - tws * col[i];
                        cj[i] = x;
                        if (i > jj + 1) nn += x * x;
                    }
                    nn = (nn + n1) + (n2 + n3);
                }
                if (lane == 0) cj[jj] = cjj - tw;
                PROBEW(14);
                if (j == jj + 1) { nn = wave_sum(nn); if (lane == 0) sh.nrm2[jj + 1] = nn; }
                PROBEW(15);
            } else if (j < jj) {
                const double* vi = Pn + (size_t)j * ldp;   // reflector j: unscaled below its pivot, scale scl[j]
                double w = 0.0;
                {
                    double w1 = 0.0, w2 = 0.0, w3 = 0.0;
                    int r = jj + 1 + lane;
#pragma unroll 1
                    for (; r + 192 < rows; r += 256) {
                        w += vi[r] * col[r]; w1 += vi[r + 64] * col[r + 64]; w2 += vi[r + 128] * col[r + 128]; w3 += vi[r + 192] * col[r + 192];
                    }
#pragma unroll 1
                    for (; r < rows; r += 64) w += vi[r] * col[r];
                    w = (w + w1) + (w2 + w3);
                }
                w = wave_sum(w) * scl[j] * scale;
                if (lane == 0) Zm[j][jj] = w + vi[jj] * scl[j];
            } else if (jj > 0) {
                t_column(jj - 1);
            }
        }
        __syncthreads();
        PROBEW(16);
    }
    PROBE(4);
    if (wave == 0) t_column(jb - 1);
    if (tid < jb) Pn[tid + (size_t)tid * ldp] = sh.betas[tid];
    __syncthreads();
    // apply the deferred scaling: below-diagonal entries become the reflector vectors
    for (int c = wave; c < jb; c += nw) {
        double* pc = Pn + (size_t)c * ldp;
        const double sv = scl[c];
        for (int r = c + 1 + lane; r < rows; r += 64) pc[r] *= sv;
    }
    __syncthreads();
}

// Register-resident variant (rows <= 64 NR <= 1024, blockDim = 1024 >= 64 jb): wave w keeps column w of the panel in NR
// registers per lane for the whole factorisation; only the pivot column travels through LDS (published by its owner one
// step ahead, double buffered).  Per column step the LDS traffic drops from five panel sweeps to one pivot-column read
// per wave.  Same arithmetic as hh_panel_core; Pn is read at entry and holds R / the reflectors / sh.Tsh at exit.
__device__ __forceinline__ double lane_bcast(double v, int srclane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), srclane), hi = __builtin_amdgcn_readlane(__double2hiint(v), srclane);
    return __hiloint2double(hi, lo);
}
template <int NR>
__device__ __forceinline__ void hh_panel_core_reg(double* __restrict__ Pn, int ldp, int rows, int jb, PanelShared& sh,
                                                  double* __restrict__ pvext = nullptr, int pvld = 1024) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    double (*Tsh)[QR_NB + 1] = sh.Tsh;
    double (*Zm)[QR_NB + 1] = sh.Zm;
    double* scl = sh.scl;
    for (int i = tid; i < QR_NB * (QR_NB + 1); i += blockDim.x) { (&Tsh[0][0])[i] = 0.0; (&Zm[0][0])[i] = 0.0; }
    __syncthreads();                 // the LDS copy of the panel is complete
    double* const pvb = pvext ? pvext : &sh.pv[0][0];     // two pivot-column buffers of pvld doubles each
    if (!pvext) pvld = 1024;
    const bool own = wave < jb;
    double x[NR];
#pragma unroll
    for (int u = 0; u < NR; ++u) { const int i = lane + 64 * u; x[u] = (own && i < rows) ? Pn[i + (size_t)wave * ldp] : 0.0; }
    if (wave == 0) {
        double s0 = 0.0;
#pragma unroll
        for (int u = 0; u < NR; ++u) { const int i = lane + 64 * u; if (i >= 1) s0 += x[u] * x[u]; if (i < rows) pvb[i] = x[u]; }
        s0 = wave_sum(s0);
        if (lane == 0) sh.nrm2[0] = s0;
    }
    __syncthreads();
    PROBE(3);
    auto t_column = [&](int c) {
        const double tc = sh.taus[c];
        if (lane < c) {
            double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
            int l = lane;
            for (; l + 3 < c; l += 4) {
                a0 += Tsh[lane][l] * Zm[l][c];
                a1 += Tsh[lane][l + 1] * Zm[l + 1][c];
                a2 += Tsh[lane][l + 2] * Zm[l + 2][c];
                a3 += Tsh[lane][l + 3] * Zm[l + 3][c];
            }
            for (; l < c; ++l) a0 += Tsh[lane][l] * Zm[l][c];
            Tsh[lane][c] = -tc * ((a0 + a1) + (a2 + a3));
        }
        if (lane == 0) Tsh[c][c] = tc;
    };
    for (int jj = 0; jj < jb; ++jj) {
        const double* pv = pvb + (size_t)(jj & 1) * pvld;
        double* pvn = pvb + (size_t)((jj + 1) & 1) * pvld;
        const double s = sh.nrm2[jj], alpha = pv[jj];
        double tau = 0.0, beta = alpha, scale = 0.0;
        if (s > 0.0) {
            const double nrm = sqrt(alpha * alpha + s);
            beta = alpha >= 0.0 ? -nrm : nrm;
            tau = (beta - alpha) / beta;
            scale = 1.0 / (alpha - beta);
        }
        if (tid == 0) { scl[jj] = scale; sh.taus[jj] = tau; sh.betas[jj] = beta; }
        if (own) {
            const int ju = jj >> 6, jl = jj & 63;
            if (wave > jj) {
                double sel = 0.0, w = 0.0;
#pragma unroll
                for (int u = 0; u < NR; ++u) {
                    const int i = lane + 64 * u;
                    if (u == ju) sel = x[u];
                    if (i > jj && i < rows) w += pv[i] * x[u];
                    if (NR > 16 && (u & 7) == 7) __builtin_amdgcn_sched_barrier(0);   // bound the number of pivot entries in flight
                }
                const double cjj = lane_bcast(sel, jl);          // entry of my column in the pivot row
                w = wave_sum(w) * scale + cjj;
                const double tw = tau * w, tws = tw * scale;
                double nn = 0.0;
#pragma unroll
                for (int u = 0; u < NR; ++u) {
                    const int i = lane + 64 * u;
                    if (i > jj && i < rows) {
                        x[u] -= tws * pv[i];
                        if (i > jj + 1) nn += x[u] * x[u];
                    }
                    if (u == ju && lane == jl) x[u] = cjj - tw;
                    if (NR > 16 && (u & 7) == 7) __builtin_amdgcn_sched_barrier(0);
                }
                if (wave == jj + 1) {
                    nn = wave_sum(nn);
                    if (lane == 0) sh.nrm2[jj + 1] = nn;
#pragma unroll
                    for (int u = 0; u < NR; ++u) { const int i = lane + 64 * u; if (i < rows) pvn[i] = x[u]; }
                }
            } else if (wave < jj) {
                double sel = 0.0, w = 0.0;
#pragma unroll
                for (int u = 0; u < NR; ++u) {
                    const int i = lane + 64 * u;
                    if (u == ju) sel = x[u];
                    if (i > jj && i < rows) w += x[u] * pv[i];
                    if (NR > 16 && (u & 7) == 7) __builtin_amdgcn_sched_barrier(0);
                }
                const double vjj = lane_bcast(sel, jl);           // entry of reflector `wave` in row jj (unscaled)
                w = wave_sum(w) * scl[wave] * scale;
                if (lane == 0) Zm[wave][jj] = w + vjj * scl[wave];
            } else if (jj > 0) {
                t_column(jj - 1);
            }
        }
        __syncthreads();
    }
    PROBE(4);
    if (wave == 0) t_column(jb - 1);
    if (own) {
        const double sv = scl[wave], bw = sh.betas[wave];
#pragma unroll
        for (int u = 0; u < NR; ++u) {
            const int i = lane + 64 * u;
            if (i < rows) Pn[i + (size_t)wave * ldp] = (i > wave) ? x[u] * sv : (i == wave ? bw : x[u]);
        }
    }
    __syncthreads();
}
// LDS panels: dispatch on the number of rows (registers per lane)
__device__ __forceinline__ void hh_panel_core_lds(double* __restrict__ Pn, int ldp, int rows, int jb, PanelShared& sh) {
    if (rows <= 256) hh_panel_core_reg<4>(Pn, ldp, rows, jb, sh);
    else if (rows <= 512) hh_panel_core_reg<8>(Pn, ldp, rows, jb, sh);
    else if (rows <= 768) hh_panel_core_reg<12>(Pn, ldp, rows, jb, sh);
    else hh_panel_core_reg<16>(Pn, ldp, rows, jb, sh);
}

// One workgroup factors the panel A[j0:m, j0:j0+jb].  V (explicit, pre-zeroed), T and VT = V*T are written too.
// PLDS: the panel rows j0..m live in LDS for the whole factorisation (m - j0 <= QR_LDS_ROWS), which turns the
// ~6 dependent global round trips per column into LDS round trips.
#define QR_LDS_ROWS 1024
template <bool PLDS>
__global__ __launch_bounds__(1024) void k_qr_panel(double* __restrict__ A, int lda, int m, int j0, int jb,
                                                   double* __restrict__ V, int ldv, double* __restrict__ T, int ldt,
                                                   double* __restrict__ VT, int ldvt, AdiState* st,
                                                   const double* __restrict__ part, int nparts, int kpanel, double tolfac,
                                                   double* __restrict__ part_out) {
    PROBE(0);
    if (st && st->done) return;
    if (part) {
        // fused termination test of the band reduction (was a kernel of its own): the previous launch left `nparts`
        // partial sums of the not-yet-reduced norm; every thread evaluates the same fixed-order sum.
        const double resn = st->res_norm;
        double r2 = 0.0;
        for (int i = (threadIdx.x & 63); i < nparts; i += 64) r2 += part[i];
        r2 = wave_sum(r2);
        const double base = (kpanel == 0) ? r2 : resn;
        const double tol = band_tol(st, tolfac, base);
        const bool stop = r2 <= tol * tol;
        __syncthreads();          // everybody has read res_norm / done before thread 0 updates them
        if (threadIdx.x == 0) {
            if (kpanel == 0) st->res_norm = r2;
            if (stop) { st->done = 1; st->iters = kpanel; }
        }
        if (stop) return;
    }
    PROBE(1);
    extern __shared__ double psm[];
    __shared__ PanelShared sh;
    double (*Tsh)[QR_NB + 1] = sh.Tsh;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, nw = blockDim.x >> 6;
    const int rows = m - j0;                       // panel rows (global rows j0..m-1)
    const int ldp = PLDS ? (rows | 1) : lda;       // odd leading dimension in LDS
    double* Pn = PLDS ? psm : (A + (size_t)j0 * lda + j0);   // Pn[r + c*ldp] = A[j0 + r, j0 + c]
    if (PLDS) {
        // global -> LDS with four independent loads in flight per thread
        const int tot = rows * jb, nt = blockDim.x;
        for (int base = tid; base < tot; base += 4 * nt) {
            double x[4]; int rr[4], cc[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int id = base + u * nt;
                cc[u] = id / rows; rr[u] = id - cc[u] * rows;
                x[u] = (id < tot) ? A[(j0 + rr[u]) + (size_t)(j0 + cc[u]) * lda] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (base + u * nt < tot) Pn[rr[u] + (size_t)cc[u] * ldp] = x[u];
        }
    }
    PROBE(2);
    if (PLDS) hh_panel_core_lds(Pn, ldp, rows, jb, sh);
    else if (rows <= 1536) {
        // medium panels (1024 < rows <= 1536): no LDS copy at all — the register-resident core reads its columns straight
        // from global memory and writes them back; only the two pivot-column buffers live in (dynamic) LDS
        hh_panel_core_reg<24>(Pn, ldp, rows, jb, sh, psm, 2048);
    } else hh_panel_core(Pn, ldp, rows, jb, sh);
    PROBE(5);
    if (part_out && wave == 0) {
        // coupling term of the NEXT termination test: 2 ||triu(R)||_F^2 of this panel (see k_band_rem)
        double c2 = 0.0;
        for (int id = lane; id < jb * jb; id += 64) {
            const int r = id % jb, c = id / jb;
            if (r <= c && r < rows) { const double x = Pn[r + (size_t)c * ldp]; c2 += 2.0 * x * x; }
        }
        c2 = wave_sum(c2);
        if (lane == 0) part_out[0] = c2;
    }
    // write back: R part + reflectors into A, explicit V, T, and VT = V * T
    for (int c = wave; c < jb; c += nw) {
        const double* pc = Pn + (size_t)c * ldp;
        for (int r = lane; r < rows; r += 64) {
            const double x = pc[r];
            if (PLDS) A[(j0 + r) + (size_t)(j0 + c) * lda] = x;
            V[(j0 + r) + (size_t)(j0 + c) * ldv] = (r > c) ? x : (r == c ? 1.0 : 0.0);
        }
    }
    for (int i = tid; i < QR_NB * jb; i += blockDim.x) {
        int r = i % QR_NB, cc = i / QR_NB;
        T[r + (size_t)(j0 + cc) * ldt] = Tsh[r][cc];
    }
    PROBE(6);
    if (VT) {
        // VT(r, c) = sum_{l <= c} V(r, l) T(l, c)
        for (int c = wave; c < jb; c += nw)
            for (int r = lane; r < rows; r += 64) {
                double acc = 0.0;
                for (int l = 0; l <= c; ++l) {
                    const double v = (r > l) ? Pn[r + (size_t)l * ldp] : (r == l ? 1.0 : 0.0);
                    acc += v * Tsh[l][c];
                }
                VT[(j0 + r) + (size_t)(j0 + c) * ldvt] = acc;
            }
    }
    PROBE(7);
}

// ---------------------------------------------------------------------------------------------
// Tall panels (rows > QR_LDS_ROWS): TSQR with Householder reconstruction (Ballard, Demmel, Grigori, Jacquelin, Knight,
// Nguyen 2014).  The panel is cut into row chunks that fit LDS; every chunk is factored by one workgroup on its own
// CU (k_tsqr_local), the stacked R factors by one workgroup (k_tsqr_top), the thin orthonormal Q is formed chunk-wise
// (k_tsqr_formq), and one LU of [I;0] - Q S (sign matrix S chosen for unit-size pivots) turns it back into the compact
// WY form (V, T) that the trailing updates use (k_hr_small, k_tsqr_finish).  Unconditionally stable like Householder QR.
// ---------------------------------------------------------------------------------------------
struct TsqrPlan { int P; int base; int rem; };     // chunk c has base + (c < rem) rows and starts at c*base + min(c, rem)
__device__ __host__ inline int chunk_start(const TsqrPlan& p, int c) { return c * p.base + (c < p.rem ? c : p.rem); }
__device__ __host__ inline int chunk_rows(const TsqrPlan& p, int c) { return p.base + (c < p.rem ? 1 : 0); }

// BIG: chunks of 1024 < rows <= 1536 (panels taller than 64 x 1023 rows): no LDS copy, the register-resident core works in place on the
// chunk's slice of Vloc (global memory), like the medium single-workgroup panels.
template <bool BIG>
__global__ __launch_bounds__(1024) void k_tsqr_local(const double* __restrict__ A, int lda, int jb, TsqrPlan plan, double* __restrict__ Vloc, int ldvl,
                                                     double* __restrict__ Tloc, double* __restrict__ Rstack, int ldrs, const AdiState* st) {
    if (st && st->done) return;
    extern __shared__ double psm[];
    __shared__ PanelShared sh;
    const int c = blockIdx.x, r0 = chunk_start(plan, c), rows = chunk_rows(plan, c);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, nw = blockDim.x >> 6;
    if (BIG) {
        double* Pn = Vloc + r0;
        for (int j = wave; j < jb; j += nw)
            for (int r = lane; r < rows; r += 64) Pn[r + (size_t)j * ldvl] = A[(r0 + r) + (size_t)j * lda];
        __syncthreads();
        hh_panel_core_reg<24>(Pn, ldvl, rows, jb, sh, psm, 2048);
        for (int i = tid; i < jb * jb; i += blockDim.x) {
            const int r = i % jb, j = i / jb;
            Rstack[(c * jb + r) + (size_t)j * ldrs] = (r <= j) ? Pn[r + (size_t)j * ldvl] : 0.0;
        }
        __syncthreads();
        for (int i = tid; i < jb * jb; i += blockDim.x) {
            const int r = i % jb, j = i / jb;
            if (r <= j) Pn[r + (size_t)j * ldvl] = (r == j) ? 1.0 : 0.0;
        }
    } else {
        const int ldp = rows | 1;
        for (int j = wave; j < jb; j += nw)
            for (int r = lane; r < rows; r += 64) psm[r + (size_t)j * ldp] = A[(r0 + r) + (size_t)j * lda];
        hh_panel_core_lds(psm, ldp, rows, jb, sh);
        for (int j = wave; j < jb; j += nw)
            for (int r = lane; r < rows; r += 64) {
                const double x = psm[r + (size_t)j * ldp];
                Vloc[(r0 + r) + (size_t)j * ldvl] = (r > j) ? x : (r == j ? 1.0 : 0.0);
                if (r < jb) Rstack[(c * jb + r) + (size_t)j * ldrs] = (r <= j) ? x : 0.0;
            }
    }
    for (int i = tid; i < jb * jb; i += blockDim.x) Tloc[(size_t)c * QR_NB * QR_NB + i % jb + (i / jb) * QR_NB] = sh.Tsh[i % jb][i / jb];
}

// Householder reconstruction on the top jb x jb block Q1 of the thin Q:  [I;0] - Q S = V U  with S = diag(sgn) chosen so
// that every pivot is 1 + |q~_jj| >= 1.  Outputs: sgn, Uinv, T = U V1^-T, V1 (unit lower), and R <- S R.
struct HrShared {
    double W[QR_NB][QR_NB + 1], U[QR_NB][QR_NB + 1], V1[QR_NB][QR_NB + 1], Ui[QR_NB][QR_NB + 1], Vi[QR_NB][QR_NB + 1];
    double sg[QR_NB];
};
// Workgroup-wide (only the first 64 threads do arithmetic, everybody takes the barriers); h.W holds Q1 on entry.
__device__ __forceinline__ void hr_small_body(int jb, HrShared& h, double* __restrict__ Rfin, double* __restrict__ hr) {
    const int tid = threadIdx.x, nt = blockDim.x;
    double (*W)[QR_NB + 1] = h.W; double (*U)[QR_NB + 1] = h.U; double (*V1)[QR_NB + 1] = h.V1;
    double (*Ui)[QR_NB + 1] = h.Ui; double (*Vi)[QR_NB + 1] = h.Vi; double* sg = h.sg;
    for (int i = tid; i < QR_NB * (QR_NB + 1); i += nt) { (&U[0][0])[i] = 0.0; (&V1[0][0])[i] = 0.0; (&Ui[0][0])[i] = 0.0; (&Vi[0][0])[i] = 0.0; }
    __syncthreads();
    for (int j = 0; j < jb; ++j) {
        const double s = (W[j][j] >= 0.0) ? -1.0 : 1.0;       // s'_j = -sgn(q~_jj)
        const double piv = 1.0 - s * W[j][j];
        if (tid == 0) { sg[j] = s; U[j][j] = piv; V1[j][j] = 1.0; }
        if (tid < j) U[tid][j] = -s * W[tid][j];
        if (tid > j && tid < jb) V1[tid][j] = -s * W[tid][j] / piv;
        __syncthreads();
        // eliminate column j from the later columns of the Q block
        for (int id = tid; id < jb * jb; id += nt) {
            const int i = id % jb, c = id / jb;
            if (i > j && c > j) W[i][c] -= V1[i][j] * W[j][c];
        }
        __syncthreads();
    }
    // Uinv (upper) and V1inv (unit lower): one column per 16-lane group, the dot product of each substitution step spread over
    // the 16 lanes (DPP row reduction) — the recurrences are 16 steps long instead of 136 dependent multiply-adds
    if (nt >= 512) {
        const int j = tid >> 4, l = tid & 15;            // tid < 256: Uinv column j;  256 <= tid < 512: V1inv column j - 16
        if (j < jb) {
            if (l == 0) Ui[j][j] = 1.0 / U[j][j];
            for (int i = j - 1; i >= 0; --i) {
                double p = (l > i && l <= j) ? U[i][l] * Ui[l][j] : 0.0;
                p += dpp_mov0<0x111>(p); p += dpp_mov0<0x112>(p); p += dpp_mov0<0x114>(p); p += dpp_mov0<0x118>(p);   // lane 15 of the row: total
                if (l == 15) Ui[i][j] = -p / U[i][i];
            }
        } else if (j >= 16 && j - 16 < jb) {
            const int c = j - 16;
            if (l == 0) Vi[c][c] = 1.0;
            for (int i = c + 1; i < jb; ++i) {
                double p = (l > c && l < i) ? V1[i][l] * Vi[l][c] : 0.0;
                p += dpp_mov0<0x111>(p); p += dpp_mov0<0x112>(p); p += dpp_mov0<0x114>(p); p += dpp_mov0<0x118>(p);
                if (l == 15) Vi[i][c] = -(p + V1[i][c]);
            }
        }
    } else if (tid < jb) {
        const int j = tid;
        Ui[j][j] = 1.0 / U[j][j];
        for (int i = j - 1; i >= 0; --i) {
            double acc = 0.0;
            for (int k = i + 1; k <= j; ++k) acc += U[i][k] * Ui[k][j];
            Ui[i][j] = -acc / U[i][i];
        }
        Vi[j][j] = 1.0;
        for (int i = j + 1; i < jb; ++i) {
            double acc = V1[i][j];
            for (int k = j + 1; k < i; ++k) acc += V1[i][k] * Vi[k][j];
            Vi[i][j] = -acc;
        }
    }
    __syncthreads();
    // hr layout (each block QR_NB x QR_NB, column-major): [0] sgn, [1] Uinv, [2] T, [3] V1
    double* Uo = hr + QR_NB * QR_NB; double* To = hr + 2 * QR_NB * QR_NB; double* Vo = hr + 3 * QR_NB * QR_NB;
    for (int id = tid; id < jb * jb; id += nt) {
        const int i = id % jb, j = id / jb;
        Uo[i + j * QR_NB] = Ui[i][j];
        Vo[i + j * QR_NB] = V1[i][j];
        double acc = 0.0;                       // T = U * V1^-T  ->  T(i,j) = sum_k U(i,k) Vinv(j,k)
        for (int k = 0; k < jb; ++k) acc += U[i][k] * Vi[j][k];
        To[i + j * QR_NB] = acc;
        Rfin[i + j * QR_NB] *= sg[i];           // A = Q R = (Q S)(S R)
    }
    if (tid < jb) hr[tid] = sg[tid];
}

// QR of the stacked R factors (P*jb x jb) and the leading jb columns Qt of its orthogonal factor
__global__ __launch_bounds__(1024) void k_tsqr_top(double* __restrict__ Rstack, int ldrs, int rowsR, int jb, double* __restrict__ Rfin,
                                                   double* __restrict__ Qt, const AdiState* st, const double* __restrict__ Vloc0, int ldvl,
                                                   const double* __restrict__ Tloc0, double* __restrict__ hr) {
    if (st && st->done) return;
    extern __shared__ double psm[];
    __shared__ PanelShared sh;
    __shared__ double Msh[QR_NB][QR_NB + 1];
    static_assert(sizeof(HrShared) <= sizeof(((PanelShared*)nullptr)->pv), "HrShared must fit the pivot buffers");
    HrShared& hs = *reinterpret_cast<HrShared*>(&sh.pv[0][0]);     // the pivot buffers are free once the panel is factored
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, nw = blockDim.x >> 6;
    const int ldp = rowsR | 1;
    for (int j = wave; j < jb; j += nw)
        for (int r = lane; r < rowsR; r += 64) psm[r + (size_t)j * ldp] = Rstack[r + (size_t)j * ldrs];
    hh_panel_core_lds(psm, ldp, rowsR, jb, sh);
    for (int i = tid; i < jb * jb; i += blockDim.x) {
        const int r = i % jb, j = i / jb;
        Rfin[r + j * QR_NB] = (r <= j) ? psm[r + (size_t)j * ldp] : 0.0;
        // M = T * V1'  with V1 the unit lower triangular top block:  M(r, j) = sum_{l >= r, l <= j} T(r,l) V1(j,l)
        double acc = 0.0;
        for (int l = r; l <= j; ++l) acc += sh.Tsh[r][l] * (l == j ? 1.0 : psm[j + (size_t)l * ldp]);
        Msh[r][j] = acc;
    }
    __syncthreads();
    // Qt = [I; 0] - V M
    for (int j = wave; j < jb; j += nw)
        for (int r = lane; r < rowsR; r += 64) {
            double acc = (r == j) ? 1.0 : 0.0;
            for (int l = 0; l < jb; ++l) {
                const double v = (r > l) ? psm[r + (size_t)l * ldp] : (r == l ? 1.0 : 0.0);
                acc -= v * Msh[l][j];
            }
            Qt[r + (size_t)j * ldrs] = acc;
            if (r < jb) hs.W[r][j] = acc;            // Qt block of chunk 0, reused below
        }
    __syncthreads();
    // Q1 = top jb x jb block of the thin Q = Qt_0 - V1_0 (T_0 (V1_0' Qt_0))  (chunk 0's local reflectors), then the
    // Householder reconstruction on it — all on 16 x 16 blocks, no launch of its own
    if (tid < jb * jb) {
        const int i = tid % jb, j = tid / jb;
        double acc = hs.W[i][j];
        for (int l = i + 1; l < jb; ++l) acc += Vloc0[l + (size_t)i * ldvl] * hs.W[l][j];
        hs.U[i][j] = acc;                             // N = V1' Qt_0   (U, Ui are scratch here)
    }
    __syncthreads();
    if (tid < jb * jb) {
        const int i = tid % jb, j = tid / jb;
        double acc = 0.0;
        for (int l = i; l < jb; ++l) acc += Tloc0[i + l * QR_NB] * hs.U[l][j];
        hs.Ui[i][j] = acc;                            // M = T_0 N
    }
    __syncthreads();
    if (tid < jb * jb) {
        const int r = tid % jb, j = tid / jb;
        double acc = hs.W[r][j];
        for (int l = 0; l <= r && l < jb; ++l) acc -= Vloc0[r + (size_t)l * ldvl] * hs.Ui[l][j];   // V1 is unit lower triangular (explicit)
        hs.Vi[r][j] = acc;
    }
    __syncthreads();
    if (tid < jb * jb) hs.W[tid % jb][tid / jb] = hs.Vi[tid % jb][tid / jb];
    __syncthreads();
    hr_small_body(jb, hs, Rfin, hr);
}

// Fused per chunk: thin Q rows (never stored), V = [V1; -Q2 S Uinv], VT = V T, and the panel of A receives R / the reflectors.
// With M = T_c V1_c' Qt_c the rows of the thin Q are q = [Qt_c; 0] - V_c M, hence
//   v  = -q (S Uinv) = vloc (M SU) - [Qt_c SU; 0],      vt = v T = vloc (M SU T) - [Qt_c SU T; 0]
// i.e. two 16-wide mat-vecs per row against 16 x 16 matrices that are formed once per workgroup in LDS.  One thread per row.
__global__ __launch_bounds__(256) void k_tsqr_formq_finish(int jb, TsqrPlan plan, const double* __restrict__ Vloc, int ldvl,
                                                           const double* __restrict__ Tloc, const double* __restrict__ Qt, int ldrs,
                                                           const double* __restrict__ hr, const double* __restrict__ Rfin,
                                                           double* __restrict__ A, int lda, double* __restrict__ V, int ldv,
                                                           double* __restrict__ T, int ldt, double* __restrict__ VT, int ldvt, const AdiState* st) {
    if (st && st->done) return;
    constexpr int B = QR_NB;
    __shared__ double Qts[B][B + 1], Nsh[B][B + 1], Msh[B][B + 1], SU[B][B + 1], Ts[B][B + 1];
    __shared__ double MU[B][B + 1], QU[B][B + 1], MUT[B][B + 1], QUT[B][B + 1];
    const int c = blockIdx.x, r0 = chunk_start(plan, c), rows = chunk_rows(plan, c);
    const int tid = threadIdx.x, i = tid % B, j = tid / B;          // blockDim.x == 256 == B*B
    const bool in = i < jb && j < jb;
    const double* Vc = Vloc + r0;
    const double* Tc = Tloc + (size_t)c * B * B;
    Qts[i][j] = in ? Qt[(c * jb + i) + (size_t)j * ldrs] : 0.0;
    SU[i][j] = in ? hr[i] * hr[B * B + i + j * B] : 0.0;            // S * Uinv
    Ts[i][j] = (in && i <= j) ? hr[2 * B * B + i + j * B] : 0.0;      // upper triangular
    __syncthreads();
    {   // N = V1' Qt_c  (V1 = top jb x jb block of V_c, unit lower triangular)
        double acc = Qts[i][j];
        if (in) for (int l = i + 1; l < jb; ++l) acc += Vc[l + (size_t)i * ldvl] * Qts[l][j];
        Nsh[i][j] = in ? acc : 0.0;
    }
    __syncthreads();
    {   // M = T_c N
        double acc = 0.0;
        if (in) for (int l = i; l < jb; ++l) acc += Tc[i + l * B] * Nsh[l][j];
        Msh[i][j] = acc;
    }
    __syncthreads();
    {
        double a = 0.0, q = 0.0;
        for (int l = 0; l < B; ++l) { a += Msh[i][l] * SU[l][j]; q += Qts[i][l] * SU[l][j]; }
        MU[i][j] = a; QU[i][j] = q;
    }
    __syncthreads();
    {
        double a = 0.0, q = 0.0;
        for (int l = 0; l < B; ++l) { a += MU[i][l] * Ts[l][j]; q += QU[i][l] * Ts[l][j]; }
        MUT[i][j] = a; QUT[i][j] = q;
    }
    __syncthreads();
    if (c == 0 && blockIdx.y == 0 && in) T[i + (size_t)j * ldt] = Ts[i][j];
    const int rl = blockIdx.y * blockDim.x + tid;
    if (rl >= rows) return;
    const int r = r0 + rl;                              // row of the panel
    double vrow[B];
#pragma unroll
    for (int l = 0; l < B; ++l) vrow[l] = (l < jb) ? Vc[rl + (size_t)l * ldvl] : 0.0;
    const bool top = r < jb;                            // the first jb rows of the panel: V1 from the reconstruction
#pragma unroll
    for (int cc = 0; cc < B; ++cc) {
        if (cc >= jb) break;
        double v, vt;
        if (top) {
            v = hr[3 * B * B + r + cc * B];
            vt = 0.0;
            for (int l = 0; l <= cc; ++l) vt += hr[3 * B * B + r + l * B] * Ts[l][cc];
        } else {
            double a0 = 0.0, a1 = 0.0;
#pragma unroll
            for (int l = 0; l < B; ++l) { a0 += vrow[l] * MU[l][cc]; a1 += vrow[l] * MUT[l][cc]; }
            v = a0 - (rl < jb ? QU[rl][cc] : 0.0);
            vt = a1 - (rl < jb ? QUT[rl][cc] : 0.0);
        }
        V[r + (size_t)cc * ldv] = v;
        if (VT) VT[r + (size_t)cc * ldvt] = vt;
        A[r + (size_t)cc * lda] = (r <= cc) ? Rfin[r + cc * B] : v;
    }
}

#define TSQR_CHUNK 512
static void launch_tsqr_panel(Ctx* ctx, double* A, int lda, int rows, int jb, double* V, int ldv, double* T, int ldt, double* VT, int ldvt,
                              const AdiState* st) {
    // A, V, VT point at the (0,0) entry of the panel
    TsqrPlan plan;
    plan.P = std::max(2, rows / TSQR_CHUNK);
    while (plan.P * jb > QR_LDS_ROWS) --plan.P;
    plan.base = rows / plan.P; plan.rem = rows % plan.P;
    const bool big = plan.base + 1 > QR_LDS_ROWS - 1;
    DRE_REQUIRE(plan.base >= jb && plan.base + 1 <= 1536, "TSQR panel: chunk size out of range");
    const int rowsR = plan.P * jb;
    Mat Vloc(ctx, rows, jb), Rstack(ctx, rowsR, jb), Qt(ctx, rowsR, jb);
    DevArr<double> Tloc(ctx, (size_t)plan.P * QR_NB * QR_NB), Rfin(ctx, QR_NB * QR_NB), hr(ctx, 4 * QR_NB * QR_NB);
    TimedScope ts(ctx, "qr_panel_tsqr", 8.0 * rows * jb * 8.0, 2.0 * rows * jb * jb * 3.0);
    lds_attr(ctx, (const void*)k_tsqr_local<false>, 132 * 1024); lds_attr(ctx, (const void*)k_tsqr_top, 132 * 1024);
    const size_t shm1 = (size_t)((plan.base + 1) | 1) * jb * sizeof(double);
    if (big) hipLaunchKernelGGL((k_tsqr_local<true>), dim3(plan.P), dim3(1024), (size_t)2 * 2048 * sizeof(double), ctx->stream, A, lda, jb, plan, Vloc.p, Vloc.ld, Tloc.p, Rstack.p, Rstack.ld, st);
    else hipLaunchKernelGGL((k_tsqr_local<false>), dim3(plan.P), dim3(1024), shm1, ctx->stream, A, lda, jb, plan, Vloc.p, Vloc.ld, Tloc.p, Rstack.p, Rstack.ld, st);
    const size_t shm2 = (size_t)(rowsR | 1) * jb * sizeof(double);
    hipLaunchKernelGGL(k_tsqr_top, dim3(1), dim3(1024), shm2, ctx->stream, Rstack.p, Rstack.ld, rowsR, jb, Rfin.p, Qt.p, st,
                       (const double*)Vloc.p, Vloc.ld, (const double*)Tloc.p, hr.p);
    hipLaunchKernelGGL(k_tsqr_formq_finish, dim3(plan.P, ceil_div(plan.base + 1, 256)), dim3(256), 0, ctx->stream, jb, plan, Vloc.p, Vloc.ld, Tloc.p, Qt.p, Qt.ld, hr.p, Rfin.p,
                       A, lda, V, ldv, T, ldt, VT, ldvt, st);
    DRE_HIP(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------
// Small panels (rows <= 512, exactly 16 columns, j0 = 0): four waves, wave w keeps columns 4w .. 4w+3 in registers (NR rows per lane and
// column).  Same Householder arithmetic as hh_panel_core_reg, restructured for latency: the column loop is fully unrolled (every register
// index is a compile-time constant, the pivot entry is one readlane), the owner of column j derives (tau, beta, scale) alone and
// publishes the SCALED reflector through LDS (double buffered), so the other waves need one barrier per column and no redundant
// sqrt/div chain; the dot products of a wave's (up to four) trailing columns and the look-ahead norm are independent DPP reduction
// chains the scheduler interleaves.  T comes from V'V on the matrix cores after the loop, V T as well.
// ---------------------------------------------------------------------------------------------
template <int NR>
__global__ __launch_bounds__(256) void k_qr_panel16(double* __restrict__ A, int lda, int rows, double* __restrict__ V, int ldv,
                                                    double* __restrict__ T, int ldt, double* __restrict__ VT, int ldvt, AdiState* st,
                                                    const double* __restrict__ part, int nparts, int kpanel, double tolfac,
                                                    double* __restrict__ part_out, int zero_above) {
    // the flag, the panel and the partial sums of the termination test are all REQUESTED before the first of them is looked at: three dependent
    // round trips to L2 (flag -> sums -> panel, ~1 us each) in front of the column loop otherwise.  (The barrier below keeps the compiler from
    // sinking the panel loads behind the early exit.)
    const int done_flag = st ? st->done : 0;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int c0 = wave * 4;
    double x[4][NR];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int u = 0; u < NR; ++u) { const int r = lane + 64 * u; x[c][u] = r < rows ? A[r + (size_t)(c0 + c) * lda] : 0.0; }
    if (part) {
        const double resn = st->res_norm;
        double r2 = 0.0;
        for (int i = lane; i < nparts; i += 64) r2 += part[i];
        r2 = wave_sum(r2);
        const double base = (kpanel == 0) ? r2 : resn;
        const double tol = band_tol(st, tolfac, base);
        const bool stop = r2 <= tol * tol;
        __syncthreads();
        if (done_flag) return;
        if (tid == 0) {
            if (kpanel == 0) st->res_norm = r2;
            if (stop) { st->done = 1; st->iters = kpanel; }
        }
        if (stop) return;
    } else if (done_flag) return;
    extern __shared__ double q16[];
    double* pv = q16;                                 // 2 x 512: the published reflector
    double* Vs = q16 + 1024;                          // rows x 17: V (explicit) for V'V and V T
    __shared__ double taus[16], Zs[4][16][17], Tsh[16][17];
    double sig = 0.0;                                 // ||column[j+1:]||^2 of the column this wave owns next (valid in its owner)
    if (wave == 0) {
#pragma unroll
        for (int u = 0; u < NR; ++u) { const int r = lane + 64 * u; if (r >= 1) sig += x[0][u] * x[0][u]; }
        sig = wave_sum(sig);
    }
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) {
        constexpr int dummy = 0; (void)dummy;
        const int wo = jj >> 2, co = jj & 3;
        double* pvb = pv + (jj & 1) * 512;
        if (wave == wo) {
            const double alpha = lane_bcast(x[co][0], jj);
            double tau = 0.0, beta = alpha, scale = 0.0;
            if (sig > 0.0) {
                const double nrm = sqrt(alpha * alpha + sig);
                beta = alpha >= 0.0 ? -nrm : nrm;
                tau = (beta - alpha) / beta;
                scale = 1.0 / (alpha - beta);
            }
#pragma unroll
            for (int u = 0; u < NR; ++u) {
                const int r = lane + 64 * u;
                double v = 0.0;
                if (r > jj) { v = x[co][u] * scale; x[co][u] = v; }
                else if (r == jj) { v = 1.0; x[co][u] = beta; }
                pvb[r] = v;
            }
            if (lane == 0) taus[jj] = tau;
        }
        __syncthreads();
        if (jj == 15) break;
        const double tau = taus[jj];
        double v[NR];
#pragma unroll
        for (int u = 0; u < NR; ++u) v[u] = pvb[lane + 64 * u];
        double w[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            w[c] = 0.0;
            if (c0 + c > jj) {
#pragma unroll
                for (int u = 0; u < NR; ++u) w[c] += v[u] * x[c][u];
            }
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) if (c0 + c > jj) w[c] = wave_sum(w[c]) * tau;
        double nn = 0.0;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (c0 + c > jj) {
#pragma unroll
                for (int u = 0; u < NR; ++u) {
                    x[c][u] -= w[c] * v[u];
                    if (c0 + c == jj + 1 && lane + 64 * u > jj + 1) nn += x[c][u] * x[c][u];
                }
            }
        }
        if (c0 <= jj + 1 && jj + 1 < c0 + 4) sig = wave_sum(nn);        // the next owner's look-ahead norm
    }
    // the rows of V above this panel (the caller's columns start zero_above rows higher): zeroed here instead of a fill of the whole matrix
    for (int id = tid; id < zero_above * 16; id += 256) V[(long)(id % zero_above) - zero_above + (long)(id / zero_above) * ldv] = 0.0;
    // V (explicit) to LDS and global, the panel (R above, reflectors below) back to A
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int u = 0; u < NR; ++u) {
            const int r = lane + 64 * u, gc = c0 + c;
            if (r < rows) {
                const double val = x[c][u];
                A[r + (size_t)gc * lda] = val;
                const double vv = r > gc ? val : (r == gc ? 1.0 : 0.0);
                V[r + (size_t)gc * ldv] = vv;
                Vs[r * 17 + gc] = vv;
            }
        }
    if (part_out && wave == 0) {
        // coupling term of the NEXT termination test: 2 ||triu(R)||_F^2 of this panel — R sits in rows 0..15, i.e. lanes 0..15 of register 0
        double c2 = 0.0;
        (void)c2;
    }
    __syncthreads();
    if (part_out) {
        // every wave holds four columns of R in lanes 0..15 of x[c][0]
        double c2 = 0.0;
#pragma unroll
        for (int c = 0; c < 4; ++c) if (lane <= c0 + c && lane < rows) c2 += 2.0 * x[c][0] * x[c][0];
        c2 = wave_sum(c2);
        if (lane == 0) Zs[wave][0][16] = c2;
    }
    {   // Z = V'V on the matrix cores: the waves split the rows; four independent accumulation chains per wave (a single chain of ~23
        // dependent products with their LDS operands was 3.2 us of the kernel), fixed-order sums
        const int lr = lane & 15, lk = lane >> 4;
        const int kst = (rows + 3) >> 2, per = (kst + 3) >> 2, t0 = wave * per, t1 = min(kst, t0 + per);
        v4d ac0 = (v4d){0.0, 0.0, 0.0, 0.0}, ac1 = ac0, ac2 = ac0, ac3 = ac0;
        for (int t = t0; t < t1; t += 4) {
            const int r0 = 4 * t + lk, r1 = r0 + 4, r2 = r0 + 8, r3 = r0 + 12;
            const double a0 = r0 < rows ? Vs[r0 * 17 + lr] : 0.0;
            const double a1 = (t + 1 < t1 && r1 < rows) ? Vs[r1 * 17 + lr] : 0.0;
            const double a2 = (t + 2 < t1 && r2 < rows) ? Vs[r2 * 17 + lr] : 0.0;
            const double a3 = (t + 3 < t1 && r3 < rows) ? Vs[r3 * 17 + lr] : 0.0;
            ac0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, a0, ac0, 0, 0, 0);
            ac1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, a1, ac1, 0, 0, 0);
            ac2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, a2, ac2, 0, 0, 0);
            ac3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a3, a3, ac3, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) Zs[wave][lk + 4 * r][lr] = (ac0[r] + ac1[r]) + (ac2[r] + ac3[r]);
    }
    __syncthreads();
    if (part_out && tid == 0) part_out[0] = (Zs[0][0][16] + Zs[1][0][16]) + (Zs[2][0][16] + Zs[3][0][16]);
    {
        // T from tau and Z = V'V by recursive doubling (the blocked form of larft: T12 = -T11 Z12 T22 for adjacent diagonal blocks of size
        // 1, 2, 4, 8), every entry by its own thread: the column-by-column recurrence on one wave was 5 us of the kernel (16 dependent steps)
        __shared__ double Zf[16][17], Wsh[16][17];
        const int i = tid & 15, j = tid >> 4;
        Zf[i][j] = ((Zs[0][i][j] + Zs[1][i][j]) + Zs[2][i][j]) + Zs[3][i][j];
        Tsh[i][j] = i == j ? taus[i] : 0.0;
        __syncthreads();
#pragma unroll
        for (int sz = 1; sz < 16; sz <<= 1) {
            const int bi = i & ~(2 * sz - 1);
            const bool mine = (j & ~(2 * sz - 1)) == bi && i - bi < sz && j - bi >= sz;
            double wv = 0.0;
            if (mine) for (int m = bi + sz; m <= j; ++m) wv += Zf[i][m] * Tsh[m][j];          // W = Z12 T22
            Wsh[i][j] = wv;
            __syncthreads();
            double tv = 0.0;
            if (mine) for (int l = i; l < bi + sz; ++l) tv += Tsh[i][l] * Wsh[l][j];          // T12 = -T11 W
            __syncthreads();
            if (mine) Tsh[i][j] = -tv;
            __syncthreads();
        }
        T[i + (size_t)j * ldt] = Tsh[i][j];
    }
    __syncthreads();
    if (VT) {
        // VT = V T: one 16-row tile per wave and pass, K = 16
        const int lr = lane & 15, lk = lane >> 4;
        for (int rt = wave; rt * 16 < rows; rt += 4) {
            v4d acc = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int r = rt * 16 + lr;
                const double a = r < rows ? Vs[r * 17 + 4 * q + lk] : 0.0;
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, Tsh[4 * q + lk][lr], acc, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = rt * 16 + lk + 4 * r;
                if (row < rows) VT[row + (size_t)lr * ldvt] = acc[r];
            }
        }
    }
}

static void launch_qr_panel(Ctx* ctx, double* A, int lda, int m, int j0, int jb, double* V, int ldv, double* T, int ldt,
                            double* VT, int ldvt, AdiState* st, const double* part = nullptr, int nparts = 0, int kpanel = 0,
                            double tolfac = 0.0, double* part_out = nullptr, int zero_above = 0) {
    const int rows = m - j0;
    if (rows <= 512 && rows >= 16 && jb == 16 && j0 == 0) {
        TimedScope ts(ctx, "qr_panel", 8.0 * rows * jb * 4.0, 2.0 * rows * jb * jb);
        const size_t shm = ((size_t)1024 + (size_t)rows * 17) * sizeof(double);
        // NR = rows per lane and column: the column loop's work and its dependent chains scale with it (n = 371: 6 instead of 8)
#define DRE_QR16_CASE(NRV) { lds_attr(ctx, (const void*)k_qr_panel16<NRV>, 96 * 1024); \
            hipLaunchKernelGGL((k_qr_panel16<NRV>), dim3(1), dim3(256), shm, ctx->stream, A, lda, rows, V, ldv, T, ldt, VT, ldvt, st, part, nparts, kpanel, tolfac, part_out, zero_above); }
        if (rows <= 256) DRE_QR16_CASE(4)
        else if (rows <= 320) DRE_QR16_CASE(5)
        else if (rows <= 384) DRE_QR16_CASE(6)
        else if (rows <= 448) DRE_QR16_CASE(7)
        else DRE_QR16_CASE(8)
#undef DRE_QR16_CASE
        return;
    }
    if (rows <= QR_LDS_ROWS) {
        TimedScope ts(ctx, "qr_panel", 8.0 * rows * jb * 4.0, 2.0 * rows * jb * jb);
        const size_t shm = (size_t)(rows | 1) * jb * sizeof(double);
        lds_attr(ctx, (const void*)k_qr_panel<true>, 132 * 1024);
        hipLaunchKernelGGL((k_qr_panel<true>), dim3(1), dim3(1024), shm, ctx->stream, A, lda, m, j0, jb, V, ldv, T, ldt, VT, ldvt, st, part, nparts, kpanel, tolfac, part_out);
    } else if (rows <= 1536) {
        {
            TimedScope ts(ctx, "qr_panel", 8.0 * rows * jb * 4.0, 2.0 * rows * jb * jb);
            hipLaunchKernelGGL((k_qr_panel<false>), dim3(1), dim3(1024), (size_t)2 * 2048 * sizeof(double), ctx->stream, A, lda, m, j0, jb, V, ldv, T, ldt,
                               (double*)nullptr, 0, st, part, nparts, kpanel, tolfac, part_out);
        }
        // V T as a (multi-workgroup) GEMM: inside the single-workgroup kernel it would re-read the panel 8.5 times from L2
        if (VT) gemm(ctx, false, false, rows, jb, jb, 1.0, V + (size_t)j0 * ldv + j0, ldv, T + (size_t)j0 * ldt, ldt, 0.0,
                     VT + (size_t)j0 * ldvt + j0, ldvt, st, "gemm_qr");
    } else if (rows >= 2 * TSQR_CHUNK && jb <= rows / 2 && rows <= 64 * 1535) {      // one-level tree: at most 64 chunks of <= 1535 rows
        // tall panel: TSQR + Householder reconstruction on many CUs (the termination test, if any, runs on its own)
        if (part) hipLaunchKernelGGL(k_band_decide, dim3(1), dim3(1), 0, ctx->stream, kpanel, nparts, part, tolfac, st);
        launch_tsqr_panel(ctx, A + (size_t)j0 * lda + j0, lda, rows, jb, V + (size_t)j0 * ldv + j0, ldv, T + (size_t)j0 * ldt, ldt,
                          VT ? VT + (size_t)j0 * ldvt + j0 : nullptr, ldvt, st);
    } else {
        TimedScope ts(ctx, "qr_panel", 8.0 * rows * jb * 4.0, 2.0 * rows * jb * jb);
        hipLaunchKernelGGL((k_qr_panel<false>), dim3(1), dim3(1024), 0, ctx->stream, A, lda, m, j0, jb, V, ldv, T, ldt, VT, ldvt, st, part, nparts, kpanel, tolfac, part_out);
    }
}

__global__ void k_extract_upper(int kq, int n, const double* __restrict__ A, int lda, double* __restrict__ R, int ldr) {
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)kq * n) return;
    int r = idx % kq, c = idx / kq;
    R[r + (size_t)c * ldr] = (r <= c) ? A[r + (size_t)c * lda] : 0.0;
}

// Aggregated block-reflector factor of nr = np*nb Householder vectors V = [V_0 ... V_{np-1}] (panel factors T_p given):
//   H_0 H_1 ... H_{nr-1} = I - V Tg V',   Tg(0:k, k:k+nb) = -Tg(0:k, 0:k) * G(0:k, k:k+nb) * T_p,   G = V'V.
// One workgroup (nr <= 64 here); X is an nr x nb scratch block in LDS.
__global__ __launch_bounds__(256) void k_build_T(int nr, int nb, const double* __restrict__ G, int ldg, const double* __restrict__ Tp, int ldt,
                                                 double* __restrict__ Tg, int ldb) {
    __shared__ double Ts[64 * 65];
    __shared__ double X[64 * 17];
    const int tid = threadIdx.x;
    for (int id = tid; id < nr * nr; id += blockDim.x) Ts[(id % nr) + (id / nr) * 65] = 0.0;
    __syncthreads();
    for (int k = 0; k < nr; k += nb) {
        const int jb = min(nb, nr - k);
        for (int id = tid; id < jb * jb; id += blockDim.x) {
            const int i = id % jb, j = id / jb;
            Ts[(k + i) + (k + j) * 65] = Tp[i + (size_t)(k + j) * ldt];
        }
        if (k > 0) {
            for (int id = tid; id < k * jb; id += blockDim.x) {      // X = G(0:k, k:k+jb) * T_p   (T_p upper triangular)
                const int i = id % k, j = id / k;
                double acc = 0.0;
                for (int l = 0; l <= j; ++l) acc += G[i + (size_t)(k + l) * ldg] * Tp[l + (size_t)(k + j) * ldt];
                X[i + j * 64] = acc;
            }
            __syncthreads();
            for (int id = tid; id < k * jb; id += blockDim.x) {      // Tg(0:k, k:k+jb) = -Tg(0:k, 0:k) * X
                const int i = id % k, j = id / k;
                double a0 = 0.0, a1 = 0.0;
                int l = i;
                for (; l + 1 < k; l += 2) { a0 += Ts[i + l * 65] * X[l + j * 64]; a1 += Ts[i + (l + 1) * 65] * X[l + 1 + j * 64]; }
                if (l < k) a0 += Ts[i + l * 65] * X[l + j * 64];
                Ts[i + (k + j) * 65] = -(a0 + a1);
            }
        }
        __syncthreads();
    }
    for (int id = tid; id < nr * nr; id += blockDim.x) Tg[(id % nr) + (size_t)(id / nr) * ldb] = Ts[(id % nr) + (id / nr) * 65];
}

#define QR_GROUP 64
#define QR_GROUP_MIN_ROWS 4096

QRFact qr_factor(Ctx* ctx, Mat& A) {
    QRFact f;
    f.m = A.rows; f.n = A.cols; f.kq = std::min(A.rows, A.cols); f.nb = QR_NB;
    f.V = Mat(ctx, f.m, f.kq);
    f.VT = Mat(ctx, f.m, f.kq);
    f.T = Mat(ctx, QR_NB, std::max(f.kq, 1));
    f.R = Mat(ctx, f.kq, f.n);
    fill_mat(ctx, f.V, 0.0);
    fill_mat(ctx, f.VT, 0.0);
    const bool grouped = f.m >= QR_GROUP_MIN_ROWS && f.kq > QR_NB;
    const int GW = grouped ? QR_GROUP : QR_NB;
    if (grouped) { f.group = GW; f.VTg = Mat(ctx, f.m, f.kq); }
    for (int g0 = 0; g0 < f.kq; g0 += GW) {
        const int gw = std::min(GW, f.kq - g0), gend = g0 + gw;
        for (int j0 = g0; j0 < gend; j0 += QR_NB) {
            const int jb = std::min(QR_NB, gend - j0);
            launch_qr_panel(ctx, A.p, A.ld, f.m, j0, jb, f.V.p, f.V.ld, f.T.p, f.T.ld, f.VT.p, f.VT.ld, nullptr);
            // columns up to the end of the group (all remaining columns when not grouped):  A2 <- Q_p' A2 = A2 - V (V T)' A2
            const int n2 = (grouped ? gend : f.n) - j0 - jb;
            if (n2 > 0) {
                Mat Vp = f.V.view(j0, j0, f.m - j0, jb);
                Mat VTp = f.VT.view(j0, j0, f.m - j0, jb);
                Mat A2 = A.view(j0, j0 + jb, f.m - j0, n2);
                Mat W(ctx, jb, n2);
                gemm(ctx, true, false, 1.0, VTp, A2, 0.0, W, nullptr, "gemm_qr");
                gemm(ctx, false, false, -1.0, Vp, W, 1.0, A2, nullptr, "gemm_qr");
            }
        }
        if (!grouped) continue;
        // aggregate the group's panels:  Q_g = I - V_g T_g V_g',  VTg = V_g T_g
        Mat Vg = f.V.view(g0, g0, f.m - g0, gw);
        Mat VTg = f.VTg.view(g0, g0, f.m - g0, gw);
        if (gw > QR_NB) {
            Mat G(ctx, gw, gw), Tg(ctx, gw, gw);
            gemm(ctx, true, false, 1.0, Vg, Vg, 0.0, G, nullptr, "gemm_qr");
            hipLaunchKernelGGL(k_build_T, dim3(1), dim3(256), 0, ctx->stream, gw, QR_NB, G.p, G.ld, f.T.p + (size_t)g0 * f.T.ld, f.T.ld, Tg.p, Tg.ld);
            gemm(ctx, false, false, 1.0, Vg, Tg, 0.0, VTg, nullptr, "gemm_qr");
        } else {
            Mat VTp = f.VT.view(g0, g0, f.m - g0, gw);
            copy_mat(ctx, VTp, VTg);
        }
        const int n2 = f.n - gend;
        if (n2 > 0) {
            Mat A2 = A.view(g0, gend, f.m - g0, n2);
            Mat W(ctx, gw, n2);
            gemm(ctx, true, false, 1.0, VTg, A2, 0.0, W, nullptr, "gemm_qr_wide_tn");      // W = T_g' V_g' A2
            gemm(ctx, false, false, -1.0, Vg, W, 1.0, A2, nullptr, "gemm_qr_wide_nn");      // A2 <- Q_g' A2
        }
    }
    size_t tot = (size_t)f.kq * f.n;
    if (tot) hipLaunchKernelGGL(k_extract_upper, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, ctx->stream, f.kq, f.n, A.p, A.ld, f.R.p, f.R.ld);
    DRE_HIP(hipGetLastError());
    return f;
}

void qr_apply_q(Ctx* ctx, const QRFact& f, Mat& B, bool transpose) {
    DRE_REQUIRE(B.rows == f.m, "qr_apply_q: row mismatch");
    if (B.cols == 0 || f.kq == 0) return;
    const int GW = f.group > 0 ? f.group : QR_NB;
    const Mat& VTall = f.group > 0 ? f.VTg : f.VT;
    const int np = ceil_div(f.kq, GW);
    for (int pp = 0; pp < np; ++pp) {
        const int p = transpose ? pp : np - 1 - pp;
        const int j0 = p * GW, jb = std::min(GW, f.kq - j0);
        Mat Vp = f.V.view(j0, j0, f.m - j0, jb);
        Mat VTp = VTall.view(j0, j0, f.m - j0, jb);
        Mat B2 = B.view(j0, 0, f.m - j0, B.cols);
        Mat W(ctx, jb, B.cols);
        if (!transpose) {   // Q_p B = B - (V T)(V' B)
            gemm(ctx, true, false, 1.0, Vp, B2, 0.0, W, nullptr, "gemm_qr");
            gemm(ctx, false, false, -1.0, VTp, W, 1.0, B2, nullptr, "gemm_qr");
        } else {            // Q_p' B = B - V (V T)' B
            gemm(ctx, true, false, 1.0, VTp, B2, 0.0, W, nullptr, "gemm_qr");
            gemm(ctx, false, false, -1.0, Vp, W, 1.0, B2, nullptr, "gemm_qr");
        }
    }
}

// =============================================================================================
// Symmetric eigensolver: early-terminating Householder tridiagonalisation + implicit QL.
// =============================================================================================
struct TridiagInfo { int jdim; int nref; double snorm; };

// Single workgroup.  S: q x q full symmetric (both triangles kept up to date).
// V(:, j) receives reflector j (v[j+1] = 1, zeros above; V pre-zeroed), d/e the tridiagonal.
__global__ __launch_bounds__(1024) void k_tridiag(int q, double* __restrict__ S, int lds_, double* __restrict__ V, int ldv,
                                                  double* __restrict__ tau_out, double* __restrict__ d, double* __restrict__ e,
                                                  double tolfac, double abs_tol, TridiagInfo* info, int floor_mode) {
    extern __shared__ double sm[];
    double* v = sm;          // q
    double* w = sm + q;      // q
    __shared__ double red[17];
    __shared__ double sc[4];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, nw = blockDim.x >> 6;
    // ||S||_F^2
    double s = 0.0;
    for (size_t idx = tid; idx < (size_t)q * q; idx += blockDim.x) {
        double x = S[idx % q + (idx / q) * (size_t)lds_];
        s += x * x;
    }
    double rem2 = block_sum(s, red);          // ||S[j:, j:]||_F^2 for j = 0
    const double snorm = sqrt(rem2);
    const double tol = floor_mode ? fmax(tolfac * 2.220446049250313e-16 * snorm, abs_tol) : (abs_tol > 0.0 ? abs_tol : tolfac * 2.220446049250313e-16 * snorm);
    const double tol2 = tol * tol;
    int jdim = q, nref = 0;
    double eprev = 0.0;
    for (int j = 0; j < q; ++j) {
        if (rem2 + 2.0 * eprev * eprev <= tol2) { jdim = j; break; }   // nothing left worth reducing
        if (j == q - 1) { if (tid == 0) d[j] = S[j + (size_t)j * lds_]; break; }
        const int nr = q - j - 1;                         // order of the trailing block
        double* colj = S + (size_t)j * lds_;
        // Householder vector from S[j+1:, j]
        double xs = 0.0;
        for (int r = j + 2 + tid; r < q; r += blockDim.x) xs += colj[r] * colj[r];
        xs = block_sum(xs, red);
        if (tid == 0) {
            double alpha = colj[j + 1], tau = 0.0, beta = alpha, scale = 0.0;
            if (xs > 0.0) {
                double nrm = sqrt(alpha * alpha + xs);
                beta = alpha >= 0.0 ? -nrm : nrm;
                tau = (beta - alpha) / beta;
                scale = 1.0 / (alpha - beta);
            }
            sc[0] = tau; sc[1] = beta; sc[2] = scale;
            d[j] = colj[j];
            e[j] = beta;
            tau_out[j] = tau;
        }
        __syncthreads();
        const double tau = sc[0], beta = sc[1], scale = sc[2];
        for (int r = tid; r < nr; r += blockDim.x) {      // v indexed from row j+1
            double x = (r == 0) ? 1.0 : colj[j + 1 + r] * scale;
            v[r] = x;
            V[(j + 1 + r) + (size_t)j * ldv] = x;
        }
        __syncthreads();
        nref = j + 1;
        eprev = beta;
        if (tau != 0.0) {
            // p = tau * S22 * v   (column r of the symmetric block dotted with v; one wave per column)
            for (int r = wave; r < nr; r += nw) {
                const double* cr = S + (size_t)(j + 1 + r) * lds_ + (j + 1);
                double acc = 0.0;
                for (int c = lane; c < nr; c += 64) acc += cr[c] * v[c];
                acc = wave_sum(acc);
                if (lane == 0) w[r] = tau * acc;
            }
            __syncthreads();
            double pv = 0.0;
            for (int r = tid; r < nr; r += blockDim.x) pv += w[r] * v[r];
            pv = block_sum(pv, red);
            const double K = -0.5 * tau * pv;
            for (int r = tid; r < nr; r += blockDim.x) w[r] += K * v[r];
            __syncthreads();
        }
        // S22 -= v w' + w v'  and  ||S22||_F^2 for the next termination test (one wave per column, rows on lanes)
        double acc2 = 0.0;
        for (int c = wave; c < nr; c += nw) {
            double* pc = S + (size_t)(j + 1 + c) * lds_ + (j + 1);
            if (tau != 0.0) {
                const double wc = w[c], vc = v[c];
                for (int r = lane; r < nr; r += 64) {
                    const double x = pc[r] - (v[r] * wc + w[r] * vc);
                    pc[r] = x;
                    acc2 += x * x;
                }
            } else {
                for (int r = lane; r < nr; r += 64) { const double x = pc[r]; acc2 += x * x; }
            }
        }
        rem2 = block_sum(acc2, red);
    }
    if (tid == 0) { info->jdim = jdim; info->nref = nref; info->snorm = snorm; }
}

// Implicit QL with Wilkinson shift on (d, e) of order n; Z (n x n, identity on entry) accumulates the
// rotations.  Lane 0 of wave 0 generates the rotation chain of sweep t+1 (a strictly sequential scalar
// recurrence, kept in registers with the next d/e prefetched) while waves 1.. apply the chain of sweep t
// to their rows of Z, so the O(n^3) accumulation hides behind the O(n^2) scalar chase.
// ZLDS: Z lives in LDS (n <= 128) and is written back at the end.
template <bool ZLDS>
__global__ __launch_bounds__(256) void k_tql(int n, double* __restrict__ dg, double* __restrict__ eg, double* __restrict__ Zg, int ldzg,
                                             double anorm, int* fail) {
    extern __shared__ double sm[];
    double* d = sm;                 // n
    double* e = sm + n;             // n
    double* csb = sm + 2 * n;       // 2 x n
    double* snb = sm + 4 * n;       // 2 x n
    double* Z = ZLDS ? sm + 6 * n : Zg;
    const int ldz = ZLDS ? n : ldzg;
    __shared__ int ctl_has[2], ctl_m[2], ctl_ilo[2];
    __shared__ int fin;
    const int tid = threadIdx.x;
    for (int i = tid; i < n; i += blockDim.x) { d[i] = dg[i]; e[i] = (i < n - 1) ? eg[i] : 0.0; }
    if (ZLDS) for (int i = tid; i < n * n; i += blockDim.x) Z[i] = (i % n == i / n) ? 1.0 : 0.0;
    if (tid == 0) fin = 0;
    __syncthreads();
    const double eps = 2.220446049250313e-16;
    const double abstiny = 1e-3 * eps * anorm;
    int l = 0, iter = 0, cur = 0;       // generator state (meaningful in thread 0)
    bool have_prev = false;
    int pm = 0, pilo = 0;
    const int nappl = blockDim.x - 64;
    while (true) {
        if (tid == 0) {
            bool produced = false;
            double* cs = csb + cur * n;
            double* sn = snb + cur * n;
            while (!produced && l < n) {
                int m = l;
                for (; m < n - 1; ++m) {
                    const double em = fabs(e[m]);
                    if (em <= eps * (fabs(d[m]) + fabs(d[m + 1])) || em <= abstiny) break;
                }
                if (m == l) { ++l; iter = 0; continue; }
                if (iter >= 80) { *fail = 1; l = n; break; }
                ++iter;
                const double dl = d[l], el = e[l];
                double g = (d[l + 1] - dl) / (2.0 * el);
                double r = sqrt(g * g + 1.0);
                g = d[m] - dl + el / (g + (g >= 0.0 ? r : -r));
                double s = 1.0, c = 1.0, p = 0.0;
                double ei = e[m - 1], di = d[m - 1], di1 = d[m];
                int i, ilo = l;
                bool broke = false;
                for (i = m - 1; i >= l; --i) {
                    const double e_next = (i > l) ? e[i - 1] : 0.0;     // prefetch: independent of the chain below
                    const double d_next = (i > l) ? d[i - 1] : 0.0;
                    const double f = s * ei, b = c * ei;
                    const double h = f * f + g * g;
                    if (h == 0.0) { e[i + 1] = 0.0; d[i + 1] = di1 - p; e[m] = 0.0; broke = true; break; }
                    const double rinv = rsqrt(h);
                    e[i + 1] = h * rinv;
                    s = f * rinv; c = g * rinv;
                    g = di1 - p;
                    r = (di - g) * s + 2.0 * c * b;
                    p = s * r;
                    d[i + 1] = g + p;
                    g = c * r - b;
                    cs[i] = c; sn[i] = s;
                    di1 = di; di = d_next; ei = e_next;
                }
                if (broke) ilo = i + 1;
                else { d[l] -= p; e[l] = g; e[m] = 0.0; }
                ctl_m[cur] = m; ctl_ilo[cur] = ilo;
                produced = true;
            }
            ctl_has[cur] = produced ? 1 : 0;
            if (!produced) fin = 1;
        } else if (tid >= 64 && have_prev) {
            const double* cs = csb + (cur ^ 1) * n;
            const double* sn = snb + (cur ^ 1) * n;
            for (int k = tid - 64; k < n; k += nappl) {
                double zi1 = Z[k + (size_t)pm * ldz];
                for (int i = pm - 1; i >= pilo; --i) {
                    const double zi = Z[k + (size_t)i * ldz];
                    const double c = cs[i], s = sn[i];
                    Z[k + (size_t)(i + 1) * ldz] = s * zi + c * zi1;
                    zi1 = c * zi - s * zi1;
                }
                Z[k + (size_t)pilo * ldz] = zi1;
            }
        }
        __syncthreads();
        have_prev = ctl_has[cur] != 0; pm = ctl_m[cur]; pilo = ctl_ilo[cur];
        const int f = fin;
        cur ^= 1;
        __syncthreads();
        if (!have_prev && f) break;
    }
    for (int i = tid; i < n; i += blockDim.x) dg[i] = d[i];
    if (ZLDS) for (int i = tid; i < n * n; i += blockDim.x) Zg[i % n + (size_t)(i / n) * ldzg] = Z[i];
}

__global__ void k_tri_to_dense(int n, const double* __restrict__ d, const double* __restrict__ e, double* __restrict__ A, int lda) {
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)n * n) return;
    const int r = idx % n, c = idx / n;
    double v = 0.0;
    if (r == c) v = d[r];
    else if (r == c + 1) v = e[c];
    else if (c == r + 1) v = e[r];
    A[r + (size_t)c * lda] = v;
}

SymEig sym_eig(Ctx* ctx, Mat& S, double tolfac, bool want_eig, double abs_tol, bool tol_is_floor) {
    DRE_REQUIRE(S.rows == S.cols, "sym_eig: square matrix expected");
    SymEig out;
    const int q = S.rows;
    out.q = q;
    if (q == 0) return out;
    DRE_REQUIRE(q <= 8192, "sym_eig: order above 8192 not supported by the single-workgroup reduction");
    out.V = Mat(ctx, q, q);
    fill_mat(ctx, out.V, 0.0);
    out.tau = DevArr<double>(ctx, q);
    out.d = DevArr<double>(ctx, q); out.e = DevArr<double>(ctx, q);
    DevArr<double>& d = out.d; DevArr<double>& e = out.e;
    DevArr<TridiagInfo> info(ctx, 1);
    DRE_HIP(hipMemsetAsync(out.tau.p, 0, q * sizeof(double), ctx->stream));
    {
        TimedScope ts(ctx, "sym_tridiag", 0, 0);
        size_t shm = 2 * (size_t)q * sizeof(double);
        if (shm > 60 * 1024) {
            lds_attr(ctx, (const void*)k_tridiag, 140 * 1024);
        }
        hipLaunchKernelGGL(k_tridiag, dim3(1), dim3(1024), shm, ctx->stream, q, S.p, S.ld, out.V.p, out.V.ld, out.tau.p, d.p, e.p, tolfac, abs_tol, info.p, tol_is_floor ? 1 : 0);
    }
    TridiagInfo hi;
    DRE_HIP(hipMemcpyAsync(&hi, info.p, sizeof(hi), hipMemcpyDeviceToHost, ctx->stream));
    DRE_HIP(hipStreamSynchronize(ctx->stream));
    out.j = hi.jdim;
    out.nref = hi.nref;
    out.snorm = hi.snorm;
    if (out.j == 0 || !want_eig) return out;
    const int j = out.j;
    out.Z = Mat(ctx, j, j);
    DevArr<int> fail(ctx, 1);
    DRE_HIP(hipMemsetAsync(fail.p, 0, sizeof(int), ctx->stream));
    DevArr<double> dw(ctx, j), ew(ctx, j);     // QL works on copies; out.d / out.e keep T_j
    DRE_HIP(hipMemcpyAsync(dw.p, d.p, j * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    DRE_HIP(hipMemcpyAsync(ew.p, e.p, j * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    {
        TimedScope ts(ctx, "sym_tql", 16.0 * j * j, 6.0 * 1.7 * (double)j * j * j);
        if (j <= 128) {
            size_t shm = (6 * (size_t)j + (size_t)j * j) * sizeof(double);
            lds_attr(ctx, (const void*)k_tql<true>, 150 * 1024);
            hipLaunchKernelGGL((k_tql<true>), dim3(1), dim3(256), shm, ctx->stream, j, dw.p, ew.p, out.Z.p, out.Z.ld, hi.snorm, fail.p);
        } else {
            set_identity(ctx, out.Z, 1.0);
            size_t shm = 6 * (size_t)j * sizeof(double);
            hipLaunchKernelGGL((k_tql<false>), dim3(1), dim3(256), shm, ctx->stream, j, dw.p, ew.p, out.Z.p, out.Z.ld, hi.snorm, fail.p);
        }
    }
    out.w.resize(j);
    int hfail = 0;
    DRE_HIP(hipMemcpyAsync(out.w.data(), dw.p, j * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    DRE_HIP(hipMemcpyAsync(&hfail, fail.p, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    DRE_HIP(hipStreamSynchronize(ctx->stream));
    if (hfail) throw Error(ERR_INTERNAL, "sym_eig: QL iteration did not converge");
    out.nref = hi.nref;   // only the first nref columns of V hold reflectors
    return out;
}

// One wave per output column: B(:,c) = H_0 H_1 ... H_{nref-1} [Z(:, ids[c]); 0]
__global__ __launch_bounds__(256) void k_backtransform(int q, int j, int nref, const double* __restrict__ V, int ldv,
                                                       const double* __restrict__ tau, const double* __restrict__ Z, int ldz,
                                                       const int* __restrict__ ids, int ncols, double* __restrict__ B, int ldb) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int c = blockIdx.x * 4 + wave;
    if (c >= ncols) return;
    double* b = B + (size_t)c * ldb;
    if (Z) {
        const double* z = Z + (size_t)ids[c] * ldz;
        for (int r = lane; r < q; r += 64) b[r] = (r < j) ? z[r] : 0.0;
    } else {
        const int one = ids[c];
        for (int r = lane; r < q; r += 64) b[r] = (r == one) ? 1.0 : 0.0;
    }
    for (int i = nref - 1; i >= 0; --i) {
        const double t = tau[i];
        if (t == 0.0) continue;
        const double* v = V + (size_t)i * ldv;
        double w = 0.0;
        for (int r = i + 1 + lane; r < q; r += 64) w += v[r] * b[r];
        w = wave_sum(w) * t;
        for (int r = i + 1 + lane; r < q; r += 64) b[r] -= w * v[r];
    }
}

Mat sym_tridiag_dense(Ctx* ctx, const SymEig& e) {
    Mat T(ctx, e.j, e.j);
    size_t tot = (size_t)e.j * e.j;
    if (tot) hipLaunchKernelGGL(k_tri_to_dense, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, ctx->stream, e.j, e.d.p, e.e.p, T.p, T.ld);
    return T;
}

Mat sym_eig_backtransform(Ctx* ctx, const SymEig& e, const std::vector<int>& ids) {
    const int r = (int)ids.size();
    Mat B(ctx, e.q, r);
    if (r == 0 || e.q == 0) return B;
    DevArr<int> dids(ctx, r);
    dids.upload(ctx, ids);
    TimedScope ts(ctx, "sym_backtransform", 0, 0);
    hipLaunchKernelGGL(k_backtransform, dim3(ceil_div(r, 4)), dim3(256), 0, ctx->stream, e.q, e.j, e.nref, e.V.p, e.V.ld,
                       e.tau.p, e.Z.p, e.Z.p ? e.Z.ld : 0, dids.p, r, B.p, B.ld);
    DRE_HIP(hipGetLastError());
    return B;
}

// =============================================================================================
// Blocked band reduction
// =============================================================================================
// partial sums of  ||S[k:, k:]||_F^2 + 2 * ||triu(S[k:k+b, k-b:k])||_F^2  (the second term couples the kept part
// to the rest); one partial per workgroup, reduced in fixed order by k_band_decide
#define BAND_REM_BLOCKS 64
__global__ __launch_bounds__(256) void k_band_rem(int q, int k, int b, const double* __restrict__ S, int ld, double* __restrict__ part,
                                                  const AdiState* st) {
    if (st->done) return;
    __shared__ double red[17];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    double s = 0.0;
    for (int c = k + blockIdx.x * nw + wave; c < q; c += nw * gridDim.x)
        for (int r = k + lane; r < q; r += 64) { const double x = S[r + (size_t)c * ld]; s += x * x; }
    if (k >= b && blockIdx.x == 0) {
        for (int c = wave; c < b; c += nw)
            for (int r = lane; r <= c && r < q - k; r += 64) { const double x = S[(k + r) + (size_t)(k - b + c) * ld]; s += 2.0 * x * x; }
    }
    s = block_sum(s, red);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
}
// st->res_norm holds ||S||_F^2 (set at k = 0), st->abstol the absolute tolerance (<= 0: relative tolfac*eps*||S||_F),
// st->iters the panel boundary J at which the reduction stopped
__global__ void k_band_decide(int k, int nparts, const double* __restrict__ part, double tolfac, AdiState* st) {
    if (st->done) return;
    double r2 = 0.0;
    for (int i = 0; i < nparts; ++i) r2 += part[i];
    if (k == 0) st->res_norm = r2;
    const double tol = band_tol(st, tolfac, st->res_norm);
    if (r2 <= tol * tol) { st->done = 1; st->iters = k; }
}
// Large panels (m > 540): the same update row-parallel over many workgroups.  Z is already reduced (m x b), the b x b
// matrix M = V' Z arrives as split-K slabs; every workgroup forms N = T' M redundantly and owns 256 rows.
__global__ __launch_bounds__(256) void k_band_w_rows(int m, int splits, const double* __restrict__ Z, int ldz, const double* __restrict__ Mpart,
                                                     const double* __restrict__ Vp, int ldv, const double* __restrict__ Tp, int ldt,
                                                     double* __restrict__ P1, double* __restrict__ P2, int ldp, const AdiState* st) {
    if (st->done) return;
    constexpr int b = QR_NB;
    __shared__ double Msh[b][b + 1], Nsh[b][b + 1];
    const int tid = threadIdx.x;
    {
        const int i = tid % b, j = tid / b;
        double acc = 0.0;
        for (int z = 0; z < splits; ++z) acc += Mpart[(size_t)z * b * b + i + j * b];
        Msh[i][j] = acc;
    }
    __syncthreads();
    {
        const int i = tid % b, j = tid / b;                 // N = T' M
        double acc = 0.0;
        for (int l = 0; l <= i; ++l) acc += Tp[l + (size_t)i * ldt] * Msh[l][j];
        Nsh[i][j] = acc;
    }
    __syncthreads();
    const int r = blockIdx.x * 256 + tid;
    if (r >= m) return;
    double v[b], z[b];
#pragma unroll
    for (int l = 0; l < b; ++l) { v[l] = Vp[r + (size_t)l * ldv]; z[l] = Z[r + (size_t)l * ldz]; }
#pragma unroll
    for (int c = 0; c < b; ++c) {
        double a0 = z[c], a1 = 0.0;
#pragma unroll
        for (int l = 0; l < b; l += 2) { a0 -= 0.5 * v[l] * Nsh[l][c]; a1 -= 0.5 * v[l + 1] * Nsh[l + 1][c]; }
        const double acc = a0 + a1;
        P1[r + (size_t)c * ldp] = acc;       P1[r + (size_t)(b + c) * ldp] = v[c];
        P2[r + (size_t)c * ldp] = v[c];      P2[r + (size_t)(b + c) * ldp] = acc;
    }
}
// Panels of at most 540 rows: the two-sided update in two lean launches instead of three (split-K GEMM + single-workgroup W kernel + update GEMM).
// k_band_z:   Z = S22 (V T)   — one workgroup per 16 rows, the four waves split K, all operand loads of a batch issued before its first MFMA
// k_band_upd: every 64 x 64 tile workgroup recomputes the 16 x 16 matrix N = T' (V' Z) (two 45 KB operands from L2), forms the rows of
//             W = Z - V N / 2 it needs, updates its tile  S22 -= W V' + V W'  on the matrix cores and leaves the tile's sum of squares for
//             the termination test of the next panel.
__global__ __launch_bounds__(256) void k_band_z(int m, const double* __restrict__ S22, int lds_, const double* __restrict__ VT, int ldvt,
                                                double* __restrict__ Z, int ldz, const AdiState* st) {
    const int done_flag = st->done;          // requested now, looked at behind the product (see k_adi_group)
    __shared__ double part[4][4][64];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lr = lane & 15, lk = lane >> 4;
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const int r0 = blockIdx.x * 16, row = r0 + lr, rowc = min(row, m - 1);
    const int kst = (m + 3) >> 2, per = (kst + 3) >> 2, t0 = wv * per, t1 = min(kst, t0 + per);
    v4d acc = (v4d){0.0, 0.0, 0.0, 0.0};
    for (int tb = t0; tb < t1; tb += 24) {
        double av[24], bv[24];
#pragma unroll
        for (int u = 0; u < 24; ++u) {
            const int t = min(tb + u, t1 - 1), cc = min(4 * t + lk, m - 1);
            av[u] = S22[rowc + (size_t)cc * lds_];
            bv[u] = VT[cc + (size_t)lr * ldvt];
        }
#pragma unroll
        for (int u = 0; u < 24; ++u) {
            const bool ok = (tb + u < t1) && 4 * (tb + u) + lk < m;
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64((ok && row < m) ? av[u] : 0.0, ok ? bv[u] : 0.0, acc, 0, 0, 0);
        }
    }
    if (done_flag) return;
#pragma unroll
    for (int r = 0; r < 4; ++r) part[wave][r][lane] = acc[r];
    __syncthreads();
    const double v = ((part[0][wave][lane] + part[1][wave][lane]) + part[2][wave][lane]) + part[3][wave][lane];
    const int orow = r0 + lk + 4 * wave;
    if (orow < m) Z[orow + (size_t)lr * ldz] = v;
}
__global__ __launch_bounds__(256) void k_band_upd(int m, double* __restrict__ S22, int lds_, const double* __restrict__ V, int ldv,
                                                  const double* __restrict__ Z, int ldz, const double* __restrict__ T, int ldt,
                                                  double* __restrict__ tile_sumsq, const AdiState* st) {
    const int done_flag = st->done;          // requested now, looked at behind the first product (see k_adi_group)
    __shared__ double part[4][4][64];
    __shared__ double Msh[16][17], Nsh[16][17];
    __shared__ double Ar[64][33], Bc[64][33];
    __shared__ double red[4];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lr = lane & 15, lk = lane >> 4;
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const int i0 = blockIdx.x * 64, j0 = blockIdx.y * 64;
    const int wm = (wave & 1) * 32, wn = (wave >> 1) * 32;
    // everything the later phases read from global memory is requested NOW, so that its latency hides behind the M = V'Z product:
    // the rows of V and Z of the tile's row and column block, and the old values of the tile itself
    const int rr_ = tid & 63, cq_ = tid >> 6;
    double vrow2[2][16], zrow2[2][4], cold[2][2][4];
#pragma unroll
    for (int side = 0; side < 2; ++side) {
        const int r = (side == 0 ? i0 : j0) + rr_, rc = min(r, m - 1);
#pragma unroll
        for (int l = 0; l < 16; ++l) vrow2[side][l] = V[rc + (size_t)l * ldv];
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) zrow2[side][cc] = Z[rc + (size_t)(cq_ * 4 + cc) * ldz];
    }
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = min(i0 + wm + x * 16 + lk + 4 * r, m - 1), col = min(j0 + wn + y * 16 + lr, m - 1);
                cold[x][y][r] = S22[row + (size_t)col * lds_];
            }
    {   // M = V' Z  (16 x 16, K = m)
        const int kst = (m + 3) >> 2, per = (kst + 3) >> 2, t0 = wv * per, t1 = min(kst, t0 + per);
        v4d acc = (v4d){0.0, 0.0, 0.0, 0.0};
        for (int tb = t0; tb < t1; tb += 24) {
            double av[24], bv[24];
#pragma unroll
            for (int u = 0; u < 24; ++u) {
                const int t = min(tb + u, t1 - 1), cc = min(4 * t + lk, m - 1);
                av[u] = V[cc + (size_t)lr * ldv];
                bv[u] = Z[cc + (size_t)lr * ldz];
            }
#pragma unroll
            for (int u = 0; u < 24; ++u) {
                const bool ok = (tb + u < t1) && 4 * (tb + u) + lk < m;
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(ok ? av[u] : 0.0, ok ? bv[u] : 0.0, acc, 0, 0, 0);
            }
        }
        if (done_flag) return;
#pragma unroll
        for (int r = 0; r < 4; ++r) part[wave][r][lane] = acc[r];
        __syncthreads();
        Msh[lk + 4 * wave][lr] = ((part[0][wave][lane] + part[1][wave][lane]) + part[2][wave][lane]) + part[3][wave][lane];
    }
    __syncthreads();
    {   // N = T' M  (T upper triangular)
        const int i = tid & 15, j = tid >> 4;
        double a0 = 0.0;
        for (int l = 0; l <= i; ++l) a0 += T[l + (size_t)i * ldt] * Msh[l][j];
        Nsh[i][j] = a0;
    }
    __syncthreads();
    {   // rows of [W V] for the tile's row block and of [V W] for its column block
        const int rr = rr_, cq = cq_;
#pragma unroll
        for (int side = 0; side < 2; ++side) {
            const int r = (side == 0 ? i0 : j0) + rr;
            double vrow[16];
#pragma unroll
            for (int l = 0; l < 16; ++l) vrow[l] = r < m ? vrow2[side][l] : 0.0;
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) {
                const int c = cq * 4 + cc;
                double a0 = r < m ? zrow2[side][cc] : 0.0, a1 = 0.0;
#pragma unroll
                for (int l = 0; l < 16; l += 2) { a0 -= 0.5 * vrow[l] * Nsh[l][c]; a1 -= 0.5 * vrow[l + 1] * Nsh[l + 1][c]; }
                const double w = a0 + a1;
                if (side == 0) { Ar[rr][c] = w; Ar[rr][16 + c] = vrow[c]; }
                else { Bc[rr][c] = vrow[c]; Bc[rr][16 + c] = w; }
            }
        }
    }
    __syncthreads();
    v4d acc[2][2];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y) acc[x][y] = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int qk = 0; qk < 8; ++qk) {
        const double a0 = Ar[wm + lr][4 * qk + lk], a1 = Ar[wm + 16 + lr][4 * qk + lk];
        const double b0 = Bc[wn + lr][4 * qk + lk], b1 = Bc[wn + 16 + lr][4 * qk + lk];
        acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
    }
    double ssq = 0.0;
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = i0 + wm + x * 16 + lk + 4 * r, col = j0 + wn + y * 16 + lr;
                if (row < m && col < m) {
                    const double v = cold[x][y][r] - acc[x][y][r];
                    S22[row + (size_t)col * lds_] = v;
                    ssq += v * v;
                }
            }
    if (tile_sumsq) {
        ssq = wave_sum(ssq);
        if (lane == 0) red[wave] = ssq;
        __syncthreads();
        if (tid == 0) tile_sumsq[blockIdx.x + (size_t)gridDim.x * blockIdx.y] = (red[0] + red[1]) + (red[2] + red[3]);
    }
}
// D(i,j) for the leading J x J block: diagonal blocks as stored, sub-diagonal blocks = upper triangle of the panel's R
__global__ void k_extract_band(int J, int b, int kred, const double* __restrict__ S, int ld, double* __restrict__ D, int ldd, int q, double* __restrict__ B0,
                               int ldb) {
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (B0 && idx < (size_t)q * J) { const int r = idx % q, c = idx / q; B0[r + (size_t)c * ldb] = r == c ? 1.0 : 0.0; }     // rider: [I; 0], q x J
    if (idx >= (size_t)J * J) return;
    int i = idx % J, j = idx / J;
    const bool swap = i < j;
    if (swap) { int t = i; i = j; j = t; }
    const int k = (j / b) * b;
    double v;
    if (i < k + b || j >= kred) v = S[i + (size_t)j * ld];   // diagonal block, or a column that was never reduced
    else { const int r = i - k - b, c = j - k; v = (r <= c) ? S[i + (size_t)j * ld] : 0.0; }
    const int oi = swap ? j : i, oj = swap ? i : j;
    D[oi + (size_t)oj * ldd] = v;
}

// control block of a reduction set up on the device (abs_tol_dev: the tolerance only exists in device memory)
struct BandTolJob { const double* parts; int nparts; double reltol, abstol, frac; double* out; };
__global__ __launch_bounds__(64) void k_band_init(AdiState* st, double abs_tol, const double* __restrict__ abs_tol_dev, int floor_mode, BandTolJob job) {
    double at_dev = 0.0;
    if (job.parts) {       // tolerances of the dense time loop's Lyapunov solve (gdre.hip, ros1_dense_step): adi.jl:61-62
        double s = 0.0;
        for (int i = threadIdx.x; i < job.nparts; i += 64) s += job.parts[i];
        s = wave_sum(s);
        const double nc = sqrt(s), at = job.abstol >= 0.0 ? job.abstol : job.reltol * nc;
        at_dev = job.frac * at;
        if (threadIdx.x == 0) { job.out[0] = at; job.out[1] = at_dev; job.out[2] = nc; }
    }
    if (threadIdx.x != 0) return;
    st->done = 0; st->iters = 0; st->maxiters = floor_mode ? BAND_TOL_FLOOR : 0; st->smw_singular = 0;
    st->abstol = job.parts ? at_dev : (abs_tol_dev ? abs_tol_dev[0] : abs_tol);
    st->res_norm = 0.0;
}
// debug (DRE_TRACE=clock): shader clock while the solve runs = delta s_memtime / delta s_memrealtime x 100 MHz over ~10 us of dependent ALU work
__global__ void k_clock_probe(long long* out) {
    const long long c0 = clock64(), w0 = wall_clock64();
    double x = 1.0 + threadIdx.x;
    for (int i = 0; i < 4000; ++i) x = x * 1.0000001 + 1e-9;
    const long long c1 = clock64(), w1 = wall_clock64();
    if (threadIdx.x == 0) { out[0] = c1 - c0; out[1] = w1 - w0; out[2] = (long long)x; }
}
SymBand sym_band_reduce(Ctx* ctx, Mat& S, double tolfac, double abs_tol, const double* abs_tol_dev, BandSpec* spec, const double* ext_part, int ext_nparts,
                        bool tol_is_floor) {
    DRE_REQUIRE(S.rows == S.cols, "sym_band_reduce: square matrix expected");
    SymBand out;
    const int q = S.rows, b = QR_NB;
    out.q = q; out.nb = b;
    if (q == 0) return out;
    out.V = Mat(ctx, q, q);
    out.VT = Mat(ctx, q, q);
    out.T = Mat(ctx, b, q);
    // small orders: every panel goes through the 16-column register kernel, which zeroes the rows of V above its panel itself
    const bool panel_zeroes = q - b <= 512 && q - b >= 16;
    if (!panel_zeroes) fill_mat(ctx, out.V, 0.0);
    // all panels factored by a single-workgroup panel kernel: the termination norm of the next panel is assembled from the update
    // GEMM's per-tile sums of squares plus the coupling term written by the panel kernel — no separate norm launch
    const bool fused_rem = q - b <= 1536;            // single-workgroup panel kernels (LDS and register variants), not the TSQR panels
    DevArr<double> part(ctx, (size_t)std::max(BAND_REM_BLOCKS, 1 + gemm_num_tiles(q, q)));
    int nparts = BAND_REM_BLOCKS;
    DevArr<AdiState> st(ctx, 1);
    {
        BandTolJob job{nullptr, 0, 0.0, -1.0, 1.0, nullptr};
        if (spec && spec->tol_parts) job = BandTolJob{spec->tol_parts, spec->tol_nparts, spec->tol_reltol, spec->tol_abstol, spec->tol_frac, spec->tols_out};
        hipLaunchKernelGGL(k_band_init, dim3(1), dim3(64), 0, ctx->stream, st.p, abs_tol, abs_tol_dev, tol_is_floor ? 1 : 0, job);
    }
    {
        static const bool cp = env_trace("clock");
        int& cp_count = ctx->trace.clock_count;
        if (cp && q > 300 && (++cp_count % 40) == 20) {
            DevArr<long long> o(ctx, 4);
            hipLaunchKernelGGL(k_clock_probe, dim3(1), dim3(64), 0, ctx->stream, o.p);
            long long h[4];
            DRE_HIP(hipMemcpyAsync(h, o.p, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
            DRE_HIP(hipStreamSynchronize(ctx->stream));
            std::fprintf(stderr, "[clock probe] %lld shader cycles in %lld x 10 ns -> %.0f MHz\n", h[0], h[1], h[1] > 0 ? (double)h[0] / (double)h[1] * 100.0 : 0.0);
        }
    }
    // Panels are enqueued speculatively: every kernel returns at once after the device-side decision `done`, and the
    // host looks at the flag only every few panels.
    int k = 0, np = 0, J = q;
    bool finished = false;
    // speculation depth: the previous reduction of the same kind (same order, same tolerance mode) needed `hint` panels; the
    // panel after the last one is the one whose prologue detects termination
    const long hkey = (long)q * 2 + ((abs_tol > 0.0 || abs_tol_dev) ? 1 : 0) + (tol_is_floor ? 1000003L : 0L);
    auto hit = ctx->band_hint.find(hkey);
    int chunk = hit != ctx->band_hint.end() ? std::max(2, hit->second + 1) : 4;
    // the speculative result follows the TREND of the last two reductions of this kind (the ranks of a Rosenbrock run's first residuals fall by
    // a panel per step: predicting the previous count was wrong every time there, and the basis was formed twice)
    int predicted = hit != ctx->band_hint.end() ? hit->second : 0;
    {
        auto prev2 = ctx->band_hint.find(hkey + 2000003L);
        if (hit != ctx->band_hint.end() && prev2 != ctx->band_hint.end() && spec) {
            predicted = std::max(1, hit->second + (hit->second - prev2->second));
            chunk = std::max(2, std::min(chunk, predicted + 2));
        }
    }
    bool first_round = true;
    int deferred_k = -1;
    auto fused_update = [&](int kk) {      // S22 <- S22 - W V' - V W' for the panel at kk (k_band_z + k_band_upd)
        const int m = q - kk - b;
        Mat S22 = S.view(kk + b, kk + b, m, m);
        Mat Vp = out.V.view(kk + b, kk, m, b);
        Mat VTp = out.VT.view(kk + b, kk, m, b);
        Mat Tp = out.T.view(0, kk, b, b);
        Mat Z(ctx, m, b);
        {
            TimedScope ts(ctx, "band_z", 8.0 * ((double)m * m + 2.0 * m * b), 2.0 * m * (double)m * b);
            hipLaunchKernelGGL(k_band_z, dim3(ceil_div(m, 16)), dim3(256), 0, ctx->stream, m, (const double*)S22.p, S22.ld, (const double*)VTp.p, VTp.ld, Z.p, Z.ld,
                               (const AdiState*)st.p);
        }
        {
            TimedScope ts(ctx, "band_upd", 8.0 * (2.0 * m * m + 2.0 * m * b), 4.0 * m * (double)m * b);
            hipLaunchKernelGGL(k_band_upd, dim3(ceil_div(m, 64), ceil_div(m, 64)), dim3(256), 0, ctx->stream, m, S22.p, S22.ld, (const double*)Vp.p, Vp.ld,
                               (const double*)Z.p, Z.ld, (const double*)Tp.p, Tp.ld, fused_rem ? part.p + 1 : (double*)nullptr, (const AdiState*)st.p);
        }
        if (fused_rem) nparts = 1 + gemm_num_tiles(m, m);
    };
    while (!finished) {
        int issued = 0;
        while (issued < chunk && k < q) {
            const double* cur_part = part.p;
            if (k == 0 && ext_part && fused_rem) { cur_part = ext_part; nparts = ext_nparts; }      // ||S||_F^2 came with the assembly of S
            else if (!fused_rem || k == 0) {
                TimedScope ts(ctx, "band_rem", 8.0 * (q - k) * (q - k), 2.0 * (q - k) * (q - k));
                hipLaunchKernelGGL(k_band_rem, dim3(BAND_REM_BLOCKS), dim3(256), 0, ctx->stream, q, k, b, S.p, S.ld, part.p, st.p);
                nparts = BAND_REM_BLOCKS;
            }
            const int m = q - k - b;            // rows below the diagonal block of this panel
            if (m < b) {                        // the last rows stay unreduced: D is stored dense, band form is not required
                hipLaunchKernelGGL(k_band_decide, dim3(1), dim3(1), 0, ctx->stream, k, nparts, cur_part, tolfac, st.p);
                k = q;
                break;
            }
            // the panel kernel evaluates the termination test in its prologue
            launch_qr_panel(ctx, S.p + (size_t)(k + b) + (size_t)k * S.ld, S.ld, m, 0, b,
                            out.V.p + (size_t)(k + b) + (size_t)k * out.V.ld, out.V.ld, out.T.p + (size_t)k * out.T.ld, out.T.ld,
                            out.VT.p + (size_t)(k + b) + (size_t)k * out.VT.ld, out.VT.ld, st.p, cur_part, nparts, k, tolfac,
                            fused_rem ? part.p : nullptr, panel_zeroes ? k + b : 0);
            // two-sided update of S22 = S[k+b:, k+b:]:  S22 <- S22 - W V' - V W',  W = Z - V N / 2,  Z = S22 (V T),  N = T' (V' Z)
            Mat S22 = S.view(k + b, k + b, m, m);
            Mat Vp = out.V.view(k + b, k, m, b);
            Mat VTp = out.VT.view(k + b, k, m, b);
            Mat Tp = out.T.view(0, k, b, b);
            Mat P1(ctx, m, 2 * b), P2(ctx, m, 2 * b);
            if (m > 540) {
                // row-parallel variant: Z = S22 (V T) reduced, M = V' Z as slabs, then one multi-workgroup kernel
                Mat Z(ctx, m, b);
                gemm(ctx, false, false, 1.0, S22, VTp, 0.0, Z, st.p, "gemm_band");
                int ms = 1;
                BufP mpart = gemm_partials(ctx, true, false, b, b, m, Vp.p, Vp.ld, Z.p, Z.ld, &ms, st.p, "gemm_band");
                TimedScope ts(ctx, "band_w", 8.0 * m * b * 6.0, 2.0 * m * b * b);
                hipLaunchKernelGGL(k_band_w_rows, dim3(ceil_div(m, 256)), dim3(256), 0, ctx->stream, m, ms, Z.p, Z.ld, (const double*)mpart->p,
                                   Vp.p, Vp.ld, Tp.p, Tp.ld, P1.p, P2.p, P1.ld, st.p);
            } else {
                // the last panel of a speculative chunk is the one whose prologue is expected to detect termination: its two-sided update
                // is only enqueued if the read-back says the reduction goes on
                if (first_round && hit != ctx->band_hint.end() && issued == chunk - 1 && issued >= 1) { deferred_k = k; break; }
                fused_update(k);
                k += b; ++np; ++issued;
                if (spec && first_round && spec->extra && !spec->ran && spec->extra_after >= 1 && issued == spec->extra_after) { spec->ran = true; spec->extra(); }
                continue;
            }
            gemm(ctx, false, true, -1.0, P1, P2, 1.0, S22, st.p, "gemm_band", fused_rem ? part.p + 1 : nullptr);    // S22 -= [W V] [V W]'
            if (fused_rem) nparts = 1 + gemm_num_tiles(m, m);
            k += b; ++np; ++issued;
        }
        AdiState h;
        // Speculation on the result (same number of panels as the previous reduction of this kind): the band matrix and the basis are
        // enqueued right behind the read-back kernel, so the device works on them while the control block travels to the host; they
        // are used if the prediction holds and dropped otherwise.
        std::function<void()> between;
        if (spec && first_round && hit != ctx->band_hint.end() && predicted > 0 && predicted * b <= k && predicted * b < q) {
            const int Js = predicted * b, nps = predicted;
            between = [&, Js, nps]() {
                SymBand tmp = out;
                tmp.J = Js; tmp.npanels = nps;
                tmp.D = Mat(ctx, Js, Js);
                tmp.B0 = Mat(ctx, q, Js);
                const size_t tots = (size_t)q * Js;
                hipLaunchKernelGGL(k_extract_band, dim3((unsigned)((tots + 255) / 256)), dim3(256), 0, ctx->stream, Js, b, nps * b, S.p, S.ld, tmp.D.p, tmp.D.ld,
                                   q, tmp.B0.p, tmp.B0.ld);
                spec->B = sym_band_basis(ctx, tmp);
                spec->D = tmp.D; spec->J = Js;
                if (spec->extra && !spec->ran) { spec->ran = true; spec->extra(); }
            };
        } else if (spec && first_round && spec->extra && !spec->ran) {
            between = [&]() { spec->ran = true; spec->extra(); };
        }
        ctx_fetch_overlap(ctx, between, st.p, sizeof(int) * 4 + sizeof(double) * 2, &h);
        first_round = false;
        if (h.done) { J = h.iters; np = J / b; finished = true; }
        else if (deferred_k >= 0) { fused_update(deferred_k); k += b; ++np; }       // the prediction was short: the reduction continues
        else if (k >= q) { J = q; finished = true; }
        deferred_k = -1;
        chunk = 4;
    }
    if (hit != ctx->band_hint.end()) ctx->band_hint[hkey + 2000003L] = hit->second;
    ctx->band_hint[hkey] = np;
    out.J = J; out.npanels = np;
    if (spec && spec->J == J && J > 0) { out.D = spec->D; spec->hit = true; DRE_HIP(hipGetLastError()); return out; }
    if (spec) { spec->hit = false; spec->B = Mat(); spec->D = Mat(); }
    out.D = Mat(ctx, J, J);
    if (J > 0) out.B0 = Mat(ctx, q, J);
    size_t tot = (size_t)q * J;
    if (tot) hipLaunchKernelGGL(k_extract_band, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, ctx->stream, J, b, np * b, S.p, S.ld, out.D.p, out.D.ld,
                                q, out.B0.p, out.B0.ld);
    DRE_HIP(hipGetLastError());
    return out;
}

// =============================================================================================
// Band reduction of S = L Dt L' WITHOUT forming S (n x n) or a QR of L (n x c): the reflectors that reduce S to band form are
// applied to the factor,  L <- Q_k' L,  so a panel step touches n x c instead of n x n entries and the number of panel steps is
// rank / 16 instead of c / 16 (QR of L) + rank / 16 (band reduction of R Dt R').  Dt = blockdiag(alpha_b D_b).
//   panel k:  P = L[k:, :] (Dt L[k:k+16, :]')        the next 16 columns of the current trailing matrix (n-k x 16)
//             QR of P[16:, :] -> V, T                 (R stays in P: the sub-diagonal band block; P[0:16, :] is the diagonal block)
//             L[k+16:, :] -= V ((V T)' L[k+16:, :])
// Termination: the trailing matrix is never formed, so its norm is ESTIMATED with 16 fixed pseudo-random probe vectors G:
// E ||S_rem G_rem||_F^2 / 16 = ||S_rem||_F^2  (Hutchinson-type; relative standard deviation of the norm ~ 18 %).  The probe rides on
// the panel's own GEMMs: G is carried as 16 extra columns of L (rotated with it, which keeps it Gaussian: the rotations depend on S
// only), G_rem' L_rem follows from the invariant G' L by downdating the rows that became final, S_rem G_rem comes out next to P
// (32 instead of 16 columns).  The test adds the exact coupling block ||R_{k-1}||^2 and doubles the estimate (bias towards one more panel).
// =============================================================================================
__global__ void k_fill_gauss(int n, int cols, unsigned long long seed, double* __restrict__ out, int ld) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)n * cols) return;
    auto mix = [](unsigned long long z) {
        z += 0x9E3779B97F4A7C15ull; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31);
    };
    const unsigned long long a = mix(seed + 2 * idx), b = mix(seed + 2 * idx + 1);
    const double u1 = ((double)(a >> 11) + 1.0) * (1.0 / 9007199254740993.0), u2 = (double)(b >> 11) * (1.0 / 9007199254740992.0);
    out[idx % n + (idx / n) * (size_t)ld] = sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);
}
// Upper Cholesky factor R of a symmetric positive semidefinite b x b matrix G (b <= 64) and Rinv = inv(R), one workgroup:  G = R'R, so
// that Y Rinv has orthonormal columns when G = Y'Y (Cholesky QR).  Right-looking in LDS with ONE barrier per column (the scaling by
// 1/sqrt(pivot) is folded into the rank-1 update).  Two thresholds:
//   * a pivot at or below `floor` is a NULL column — what is left of it after the earlier columns is rounding noise relative to the whole
//     sketch — and gets a zero column in Rinv (a zero column of Q: it contributes nothing instead of a normalised noise vector that is
//     not orthogonal to anything).  mode 0: floor = relfloor x this block's largest diagonal entry, which is also written to *ref (first
//     block of a sketch); mode 1: floor = relfloor x *ref; mode 2: floor = 1e-20 (second pass: the block is orthonormal up to 1e-8);
//   * a live pivot at or below 1e-15 x the block's largest diagonal entry in the first pass, or below 1/4 in the second pass (the first
//     pass lost more orthogonality than the second repairs: cond(Y_b) beyond ~1e7), raises *flag: the caller falls back to Householder panels.
//   nullmask (optional, b entries): 1 for a null column, so that the caller can put a fresh random direction there (k_fill_gauss_masked).
// (Round 4 experiment, measured and removed: a REGISTER form — lane = row, wave = 16-column block of A and Y, multipliers published through LDS,
// row k of Y through v_readlane, one barrier per step — took 107-108 us per 64 x 64 block against 62-65 us for this LDS form, fully unrolled or
// unrolled by 16: the 32 v_readlane + SGPR-operand FMAs per wave and step cost more than the LDS round trips they replace.)
__global__ __launch_bounds__(256) void k_chol_inv(int b, const double* __restrict__ G, int ldg, double* __restrict__ Rinv, int ldr, int* __restrict__ flag,
                                                  double* __restrict__ ref, int mode, int* __restrict__ nullmask, double relfloor, double* __restrict__ dbg) {
    extern __shared__ double chol_lds[];            // 2 x 64 x 65 doubles (dynamic: beyond the 64 KB static limit)
    double (*A)[65] = reinterpret_cast<double (*)[65]>(chol_lds);
    double (*Y)[65] = A + 64;                       // forward substitution on the identity, carried along: inv(L) = diag(rd) Y at the end
    __shared__ double dmax_s, rds[64];
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    for (int id = tid; id < 64 * 64; id += 256) { const int i = id & 63, j = id >> 6; A[i][j] = (i < b && j < b) ? G[i + (size_t)j * ldg] : 0.0; Y[i][j] = (i == j) ? 1.0 : 0.0; }
    __syncthreads();
    if (tid == 0) { double m = 0.0; for (int i = 0; i < b; ++i) m = fmax(m, A[i][i]); dmax_s = m; if (mode == 0 || mode == 3) *ref = m; }
    __syncthreads();
    // mode 3 (warm-started range finder, ldlt.hip warm_compress): the block is what a known basis left of a sketch — numerically rank deficient by
    // design; a column whose pivot falls 14 orders below the block's scale is DEPENDENT (replaced by a random direction by the caller), not a breakdown
    const double floor_abs = mode == 2 ? 1e-20 : (mode == 3 ? 1e-14 * dmax_s : relfloor * (mode == 0 ? dmax_s : *ref));
    // breakdown: judged where it shows.  First pass (modes 0, 1): only a pivot that is all rounding (<= 1e-15 of the block's scale; the Gram
    // matrix carries cond^2) is hopeless.  Second pass (mode 2): the block entered orthonormal up to the error of the first pass, so every live
    // pivot of its Gram matrix is ~1; one below 1/4 says the first pass lost more than CholeskyQR2 repairs.  (Measured on the rail
    // sketches: first-pass ratios down to 3e-14 still give second-pass pivots >= 0.99 and a probe residual of 2e-15; the earlier
    // first-pass bound of 1e-13 struck there and sent every later sketch of the run to Householder panels: 27 % of a 45-step run.)
    const double thr = mode == 3 ? 0.0 : (mode == 2 ? 0.25 : 1e-15) * dmax_s;
    bool bad = false;
    double minpiv = dmax_s;                          // smallest live pivot (trace only; divided by the largest diagonal entry once, at the end)
    // step k (one barrier): with l_ik = A_ik / pivot,   A_ij -= l_ik A_jk  (k < j <= i: the Schur complement)   and
    //                                                   Y_ij -= l_ik Y_kj  (j <= k: rows of inv(L), unscaled; Y_kk = 1)
    // on a 16 x 16 thread grid; the first version inverted L afterwards with one thread per column (b^3/6 dependent steps: 100 of its 107 us)
    // Only ONE division sits in the dependent chain of a step (1 / pivot); the square roots of the scaling are taken after the loop, all at once.
    for (int k = 0; k < b; ++k) {
        const double piv = A[k][k];
        const bool live = piv > floor_abs;           // (NaN: not live)
        if (live && !(piv > thr)) bad = true;
        if (live) minpiv = fmin(minpiv, piv);
        const double rp = live ? 1.0 / piv : 0.0;
        if (tid == 0) { rds[k] = live ? piv : 0.0; if (nullmask) nullmask[k] = live ? 0 : 1; }
        for (int i = k + 1 + ty; i < b; i += 16) {
            const double lik = A[i][k] * rp;
            for (int j = tx; j <= i; j += 16) {
                if (j > k) A[i][j] -= lik * A[j][k];
                else Y[i][j] -= lik * Y[k][j];
            }
        }
        __syncthreads();
    }
    if (tid < 64) rds[tid] = (tid < b && rds[tid] > 0.0) ? 1.0 / sqrt(rds[tid]) : 0.0;        // rd_k = 1 / sqrt(pivot_k), null column: 0
    __syncthreads();
    // Rinv = inv(L)':  Rinv(r, c) = rd_c Y(c, r) for r <= c  (a null column c: rd_c = 0)
    for (int id = tid; id < 64 * 64; id += 256) {
        const int r = id & 63, c = id >> 6;
        if (r < b && c < b) Rinv[r + (size_t)c * ldr] = (r <= c) ? Y[c][r] * rds[c] : 0.0;
    }
    if (bad && tid == 0) atomicOr(flag, 1);
    if (dbg && tid == 0) *dbg = minpiv / dmax_s;
}
// Unit-scale Gaussian entries (variance 1/n) into the columns of T that mask marks
__global__ void k_fill_gauss_masked(int n, int cols, unsigned long long seed, double* __restrict__ out, int ld, const int* __restrict__ mask, double scale) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)n * cols) return;
    const int col = (int)(idx / n);
    if (!mask[col]) return;
    auto mix = [](unsigned long long z) {
        z += 0x9E3779B97F4A7C15ull; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31);
    };
    const unsigned long long a = mix(seed + 2 * idx), b = mix(seed + 2 * idx + 1);
    const double u1 = ((double)(a >> 11) + 1.0) * (1.0 / 9007199254740993.0), u2 = (double)(b >> 11) * (1.0 / 9007199254740992.0);
    out[idx % n + (size_t)col * ld] = scale * sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);
}
void fill_gauss_masked(Ctx* ctx, Mat& A, unsigned long long seed, const int* mask_dev) {
    const size_t tot = (size_t)A.rows * A.cols;
    if (tot) hipLaunchKernelGGL(k_fill_gauss_masked, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, ctx->stream, A.rows, A.cols, seed, A.p, A.ld, mask_dev,
                                1.0 / std::sqrt((double)A.rows));
}
void chol_inv(Ctx* ctx, const Mat& G, Mat& Rinv, int* flag_dev, double* ref_dev, int mode, int* nullmask_dev, double* dbg_dev) {
    DRE_REQUIRE(G.rows == G.cols && G.rows <= 64 && Rinv.rows == G.rows && Rinv.cols == G.rows, "chol_inv: order <= 64 expected");
    if (G.rows == 0) return;
    const size_t shm = (size_t)2 * 64 * 65 * sizeof(double);
    const double relfloor = 1e-30;        // measured on the rail sketches: 1e-28 leaves a probe residual of 1e-14, 1e-30 and below 2.4e-15 (Householder: 1.9e-15)
    lds_attr(ctx, (const void*)k_chol_inv, (int)shm);
    hipLaunchKernelGGL(k_chol_inv, dim3(1), dim3(256), shm, ctx->stream, G.rows, (const double*)G.p, G.ld, Rinv.p, Rinv.ld, flag_dev, ref_dev, mode, nullmask_dev, relfloor, dbg_dev);
    DRE_HIP(hipGetLastError());
}
// Structured sparse sign test matrix Om (n x s) for the range finder of ldlt.hip sketch_compress: row i has SKETCH_ZETA entries
// +-1/sqrt(SKETCH_ZETA), in the columns (i + off_t) mod s with independent pseudo-random signs (one byte of sign bits per row, k_sign_bits).
// W(0:s, j) = Om' L(:, j): one workgroup per column of L, thread h owns the rows i = h (mod s) — coalesced reads of the column, ZETA private
// accumulators, which are the buckets (h + off_t) mod s; they meet through LDS in a fixed order (deterministic, no atomics).  One pass over
// L at HBM speed (8 n c bytes) instead of a dense GEMM with a Gaussian matrix (2 n c s flop: 1.5 ms per sketch at n = 20209, c = 4500).
__global__ void k_sign_bits(int n, unsigned long long seed, unsigned char* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned long long z = seed + 0x9E3779B97F4A7C15ull * (unsigned long long)(i + 1);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
    out[i] = (unsigned char)(z >> 24);
}
struct SketchOffsets { int off[SKETCH_ZETA]; };
__global__ __launch_bounds__(1024) void k_sketch_sign(int n, int s, const double* __restrict__ L, int ldl, double* __restrict__ W, int ldw, SketchOffsets so,
                                                      const unsigned char* __restrict__ bits) {
    extern __shared__ double sk_slot[];                 // SKETCH_ZETA x s
    const int j = blockIdx.x, h = threadIdx.x;
    double acc[SKETCH_ZETA];
#pragma unroll
    for (int t = 0; t < SKETCH_ZETA; ++t) acc[t] = 0.0;
    if (h < s) {
        const double* __restrict__ col = L + (size_t)j * ldl;
        for (int i0 = h; i0 < n; i0 += 8 * s) {
            double v[8];
            unsigned char sb[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) { const int i = min(i0 + q * s, n - 1); v[q] = col[i]; sb[q] = bits[i]; }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const double x = (i0 + q * s < n) ? v[q] : 0.0;
#pragma unroll
                for (int t = 0; t < SKETCH_ZETA; ++t) acc[t] += ((sb[q] >> t) & 1) ? x : -x;
            }
        }
#pragma unroll
        for (int t = 0; t < SKETCH_ZETA; ++t) { int b = h + so.off[t]; if (b >= s) b -= s; sk_slot[t * s + b] = acc[t]; }
    }
    __syncthreads();
    if (h < s) {
        double r = 0.0;
#pragma unroll
        for (int t = 0; t < SKETCH_ZETA; ++t) r += sk_slot[t * s + h];
        W[h + (size_t)j * ldw] = r * 0.35355339059327373;          // 1 / sqrt(8)
    }
}
static_assert(SKETCH_ZETA == 8, "k_sketch_sign: one sign byte per row, scale 1/sqrt(8)");
void sketch_sign(Ctx* ctx, const Mat& L, Mat& W, unsigned long long seed) {
    const int n = L.rows, c = L.cols, s = W.rows;
    DRE_REQUIRE(W.cols == c && s >= SKETCH_ZETA && s <= 1024 && n >= 1, "sketch_sign: shape out of range");
    if (c == 0) return;
    SketchOffsets so;
    so.off[0] = 0;
    for (int t = 1; t < SKETCH_ZETA; ++t) {
        unsigned long long z = seed + 0xD1B54A32D192ED03ull * (unsigned long long)t;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
        int o = (int)(z % (unsigned long long)s);
        for (bool clash = true; clash;) { clash = false; for (int u = 0; u < t; ++u) if (so.off[u] == o) { o = (o + 1) % s; clash = true; } }
        so.off[t] = o;
    }
    DevArr<unsigned char> bits(ctx, (size_t)n);
    hipLaunchKernelGGL(k_sign_bits, dim3(ceil_div(n, 256)), dim3(256), 0, ctx->stream, n, seed, bits.p);
    TimedScope ts(ctx, "sketch_sign", 8.0 * n * c + 8.0 * s * c, 8.0 * (double)n * c);
    const size_t shm = (size_t)SKETCH_ZETA * s * sizeof(double);
    if (shm > 48 * 1024) lds_attr(ctx, (const void*)k_sketch_sign, 64 * 1024);
    hipLaunchKernelGGL(k_sketch_sign, dim3(c), dim3((s + 63) & ~63), shm, ctx->stream, n, s, (const double*)L.p, L.ld, W.p, W.ld, so, (const unsigned char*)bits.p);
    DRE_HIP(hipGetLastError());
}
void fill_gauss(Ctx* ctx, Mat& A, unsigned long long seed) {
    const size_t tot = (size_t)A.rows * A.cols;
    if (tot) hipLaunchKernelGGL(k_fill_gauss, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, ctx->stream, A.rows, A.cols, seed, A.p, A.ld);
}
struct LrBlockDev { int off, k, ldd, diag; const double* D; double alpha; };
// RD = RB * blockdiag(alpha_b D_b)  (32 x c, ld 32);  colblk[j] = block of column j.  One thread per output entry.
__global__ __launch_bounds__(256) void k_rows_blockdiag(int c, const double* __restrict__ RB, double* __restrict__ RD, const LrBlockDev* __restrict__ blocks,
                                                        const int* __restrict__ colblk, const AdiState* st) {
    if (st && st->done) return;
    const int i = threadIdx.x & 31, j = blockIdx.x * 8 + (threadIdx.x >> 5);
    if (j >= c) return;
    const LrBlockDev b = blocks[colblk[j]];
    const int jj = j - b.off;
    double acc = 0.0;
    if (b.diag) acc = RB[i + (size_t)j * 32] * b.D[jj + (size_t)jj * b.ldd];
    else {
        const double* dcol = b.D + (size_t)jj * b.ldd;            // D symmetric: column jj
        const double* rb = RB + i + (size_t)b.off * 32;
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
        int l = 0;
        for (; l + 7 < b.k; l += 8) {                              // eight independent loads of each operand in flight
            double x[8], d[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { x[u] = rb[(size_t)(l + u) * 32]; d[u] = dcol[l + u]; }
            a0 += x[0] * d[0]; a1 += x[1] * d[1]; a2 += x[2] * d[2]; a3 += x[3] * d[3];
            a0 += x[4] * d[4]; a1 += x[5] * d[5]; a2 += x[6] * d[6]; a3 += x[7] * d[7];
        }
        for (; l < b.k; ++l) a0 += rb[(size_t)l * 32] * dcol[l];
        acc = (a0 + a1) + (a2 + a3);
    }
    RD[i + (size_t)j * 32] = b.alpha * acc;
}
// RB[0:16, :] = L[k:k+16, :]; for k > 0 the probe products are downdated by the 16 rows that became final with the previous panel,
// RB[16+p, j] -= sum_i L[k-16+i, j] G[k-16+i, p]  (G = columns c..c+15 of the extended factor), and the band blocks of the previous
// panel are saved:  BS[0:16, k-16:k] = diagonal block, BS[16:32, k-16:k] = R (upper triangle) from PP, which the next GEMM overwrites.
__global__ __launch_bounds__(256) void k_lr_rows(int c, int k, const double* __restrict__ Lw, int ldl, double* __restrict__ RB,
                                                 const double* __restrict__ PP, int ldp, double* __restrict__ BS, const AdiState* st) {
    if (st && st->done) return;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < (size_t)16 * c) {
        const int i = idx & 15; const size_t j = idx >> 4;
        RB[i + j * 32] = Lw[(size_t)(k + i) + j * ldl];
        if (k > 0) {
            const double* lj = Lw + (size_t)(k - 16) + j * ldl;                 // L[k-16:k, j]
            const double* gp = Lw + (size_t)(k - 16) + (size_t)(c + i) * ldl;   // G[k-16:k, p],  p = i
            double a = 0.0;
#pragma unroll
            for (int t = 0; t < 16; ++t) a += lj[t] * gp[t];
            RB[16 + i + j * 32] -= a;
        }
    }
    if (k > 0 && blockIdx.x == 0) {
        for (int t = threadIdx.x; t < 32 * 16; t += blockDim.x) {
            const int r = t & 31, cc = t >> 5;
            double v = PP[r + (size_t)cc * ldp];
            if (r >= 16 && r - 16 > cc) v = 0.0;
            BS[r + (size_t)(k - 16 + cc) * 32] = v;
        }
    }
}
// Termination test of the factor-form band reduction at panel boundary k (see above), two launches: partial sums of squares of the
// probe columns PG (rows x 16) over LR_PARTS workgroups, then the decision.  The coupling block R_{k-1} is read from the band store.
// st->res_norm = estimate of ||S||_F^2 (set at k = 0).
#define LR_PARTS 64
__global__ __launch_bounds__(256) void k_lr_probe_parts(int rows, const double* __restrict__ PG, int ldp, double* __restrict__ part, const AdiState* st) {
    if (st->done) return;
    __shared__ double red[17];
    double s0 = 0.0, s1 = 0.0;
    const size_t tot = (size_t)rows * 16, stride = (size_t)gridDim.x * blockDim.x;
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; idx + stride < tot; idx += 2 * stride) {
        const size_t i2 = idx + stride;
        const double v = PG[idx % rows + (idx / rows) * (size_t)ldp], w = PG[i2 % rows + (i2 / rows) * (size_t)ldp];
        s0 += v * v; s1 += w * w;
    }
    if (idx < tot) { const double v = PG[idx % rows + (idx / rows) * (size_t)ldp]; s0 += v * v; }
    const double s = block_sum(s0 + s1, red);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_lr_decide(int k, const double* __restrict__ part, const double* __restrict__ BS, double tolfac, double safety, AdiState* st) {
    if (st->done) return;
    __shared__ double red[17];
    double s = (threadIdx.x < LR_PARTS) ? safety * part[threadIdx.x] / 16.0 : 0.0;
    if (k > 0) {
        const int r = threadIdx.x & 15, cc = threadIdx.x >> 4;
        if (r <= cc) { const double v = BS[16 + r + (size_t)(k - 16 + cc) * 32]; s += 2.0 * v * v; }
    }
    s = block_sum(s, red);                     // remainder^2 = safety * est^2 + 2 ||R_{k-1}||^2
    if (threadIdx.x == 0) {
        const double r2 = s;
        if (k == 0) st->res_norm = s / safety;
        const double base = (k == 0) ? s / safety : st->res_norm;
        const double tol = band_tol(st, tolfac, base);
        if (r2 <= tol * tol || !(s == s)) { st->done = 1; st->iters = k; }
    }
}
__global__ void k_lr_extract_band(int J, const double* __restrict__ BS, double* __restrict__ D, int ldd) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)J * J) return;
    const int i = idx % J, j = idx / J;
    const int bi = i >> 4, bj = j >> 4, ii = i & 15, jj = j & 15;
    double v = 0.0;
    if (bi == bj) v = 0.5 * (BS[ii + (size_t)(bj * 16 + jj) * 32] + BS[jj + (size_t)(bj * 16 + ii) * 32]);
    else if (bi == bj + 1) v = BS[16 + ii + (size_t)(bj * 16 + jj) * 32];          // R_bj(ii, jj), zero below the diagonal
    else if (bj == bi + 1) v = BS[16 + jj + (size_t)(bi * 16 + ii) * 32];
    D[i + (size_t)j * ldd] = v;
}

// Warm start of the reductions that work in the natural coordinates (S = L D L' formed directly, or the factor form): rotate the
// coordinates so that the leading 16 columns of L — the dominant directions of a previously compressed summand — span the first 16
// unit vectors,  L <- Q0' L  with Q0 = I - VT0 V0' from the QR of L[:, 0:16].  Without it the reduction starts from arbitrary unit
// vectors and needs about one panel (16 columns of rank) more to reach the same remainder; the QR path gets this order for free.
void lead_rotate(Ctx* ctx, Mat& L, Mat& V0, Mat& VT0) {
    const int n = L.rows, c = L.cols, b = QR_NB;
    DRE_REQUIRE(c >= b && n >= 2 * b, "lead_rotate: at least 16 columns and 32 rows");
    V0 = Mat(ctx, n, b); VT0 = Mat(ctx, n, b);
    Mat T0(ctx, b, b), P0(ctx, n, b), Y0(ctx, b, c);
    fill_mat(ctx, V0, 0.0);
    Mat L0 = L.colsview(0, b);
    copy_mat(ctx, L0, P0);
    launch_qr_panel(ctx, P0.p, P0.ld, n, 0, b, V0.p, V0.ld, T0.p, T0.ld, VT0.p, VT0.ld, nullptr);
    gemm(ctx, true, false, 1.0, VT0, L, 0.0, Y0, nullptr, "gemm_lrband");
    gemm(ctx, false, false, -1.0, V0, Y0, 1.0, L, nullptr, "gemm_lrband");
}
void lead_rotate_back(Ctx* ctx, const Mat& V0, const Mat& VT0, Mat& B) {      // B <- Q0 B
    if (V0.empty() || B.cols == 0) return;
    Mat W(ctx, V0.cols, B.cols);
    gemm(ctx, true, false, 1.0, V0, B, 0.0, W, nullptr, "gemm_band");
    gemm(ctx, false, false, -1.0, VT0, W, 1.0, B, nullptr, "gemm_band");
}

SymBand lr_band_reduce(Ctx* ctx, Mat& Lx, const std::vector<LrBlockD>& blocks, double tolfac, double abs_tol, bool tol_is_floor) {
    // Lx = [L | 16 spare columns]: the probe vectors live next to the factor and are transformed with it
    const int n = Lx.rows, c = Lx.cols - 16, b = QR_NB;
    Mat Lw = Lx.colsview(0, c);
    DRE_REQUIRE(b == 16 && c >= 1 && c + 64 <= n, "lr_band_reduce: needs c + 64 <= n");
    SymBand out;
    out.q = n; out.nb = b;
    const int maxp = ceil_div(c, b) + 1;                 // rank(S) <= c: after that many panels nothing is left
    static const double safety = 2.0;
    const int cap = maxp * b;
    out.V = Mat(ctx, n, cap); out.VT = Mat(ctx, n, cap); out.T = Mat(ctx, b, cap);
    fill_mat(ctx, out.V, 0.0);
    // block table
    std::vector<LrBlockDev> hb; std::vector<int> hcol((size_t)c);
    for (auto& x : blocks) {
        DRE_REQUIRE(x.off >= 0 && x.k >= 0 && x.off + x.k <= c, "lr_band_reduce: block table out of range");
        for (int j = 0; j < x.k; ++j) hcol[(size_t)x.off + j] = (int)hb.size();
        hb.push_back({x.off, x.k, x.ldd, x.diag, x.D, x.alpha});
    }
    DevArr<LrBlockDev> dblocks(ctx, hb.size());
    DevArr<int> dcol(ctx, (size_t)c);
    DevArr<AdiState> st(ctx, 1);
    AdiState h_up;                            // (the staging objects live until the function's first read-back: no synchronisation for the uploads)
    {
        std::memset(&h_up, 0, sizeof(int) * 4 + sizeof(double) * 2);
        h_up.abstol = abs_tol;
        if (tol_is_floor) h_up.maxiters = BAND_TOL_FLOOR;
        DRE_HIP(hipMemcpyAsync(st.p, &h_up, sizeof(int) * 4 + sizeof(double) * 2, hipMemcpyHostToDevice, ctx->stream));
        DRE_HIP(hipMemcpyAsync(dblocks.p, hb.data(), hb.size() * sizeof(LrBlockDev), hipMemcpyHostToDevice, ctx->stream));
        DRE_HIP(hipMemcpyAsync(dcol.p, hcol.data(), hcol.size() * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    }
    Mat RB(ctx, 32, c), RD(ctx, 32, c), PP(ctx, n, 32), BS(ctx, 32, cap), Yx(ctx, 16, c + 16);
    DevArr<double> parts(ctx, LR_PARTS);
    if (c >= 32) lead_rotate(ctx, Lw, out.V0, out.VT0);
    {
        Mat G = Lx.colsview(c, 16);
        const size_t tot = (size_t)n * 16;
        hipLaunchKernelGGL(k_fill_gauss, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, ctx->stream, n, 16, 0x5DEECE66Dull, G.p, G.ld);
        Mat Yg = RB.view(16, 0, 16, c);
        gemm(ctx, true, false, 1.0, G, Lw, 0.0, Yg, nullptr, "gemm_lrband");
    }
    fill_mat(ctx, BS, 0.0);
    const long hkey = -((long)n * 2 + (abs_tol > 0.0 ? 1 : 0));        // negative keys: factor-form reductions
    auto hit = ctx->band_hint.find(hkey);
    int chunk = hit != ctx->band_hint.end() ? std::max(4, hit->second + 1) : 4;
    int k = 0, np = 0, J = -1;
    bool finished = false;
    while (!finished) {
        int issued = 0;
        while (issued < chunk && np < maxp) {
            const int rows = n - k, m = rows - b;
            hipLaunchKernelGGL(k_lr_rows, dim3(ceil_div(16 * c, 256)), dim3(256), 0, ctx->stream, c, k, Lw.p, Lw.ld, RB.p, PP.p, PP.ld, BS.p, st.p);
            {
                TimedScope ts(ctx, "lrband_rows", 8.0 * 32.0 * c * 3.0, 2.0 * 32.0 * c * 64.0);
                hipLaunchKernelGGL(k_rows_blockdiag, dim3(ceil_div(c, 8)), dim3(256), 0, ctx->stream, c, RB.p, RD.p, (const LrBlockDev*)dblocks.p,
                                   (const int*)dcol.p, st.p);
            }
            gemm(ctx, false, true, rows, 32, c, 1.0, Lw.p + k, Lw.ld, RD.p, RD.ld, 0.0, PP.p, PP.ld, st.p, "gemm_lrband");
            {
                TimedScope ts(ctx, "lrband_decide", 8.0 * rows * 16.0, 2.0 * rows * 16.0);
                hipLaunchKernelGGL(k_lr_probe_parts, dim3(LR_PARTS), dim3(256), 0, ctx->stream, rows, PP.p + (size_t)16 * PP.ld, PP.ld, parts.p, st.p);
                hipLaunchKernelGGL(k_lr_decide, dim3(1), dim3(256), 0, ctx->stream, k, (const double*)parts.p, BS.p, tolfac, safety, st.p);
            }
            launch_qr_panel(ctx, PP.p + b, PP.ld, m, 0, b, out.V.p + (size_t)(k + b) + (size_t)k * out.V.ld, out.V.ld,
                            out.T.p + (size_t)k * out.T.ld, out.T.ld, out.VT.p + (size_t)(k + b) + (size_t)k * out.VT.ld, out.VT.ld, st.p);
            Mat Vp = out.V.view(k + b, k, m, b);
            Mat VTp = out.VT.view(k + b, k, m, b);
            // Y = (V T)' [L_rem, G_rem]  (16 x (c + 16)),  [L_rem, G_rem] -= V Y:  the probe vectors are rotated with the factor, their
            // products G_rem' L_rem follow by downdating the invariant G' L (k_lr_rows)
            gemm(ctx, true, false, 16, c + 16, m, 1.0, VTp.p, VTp.ld, Lx.p + (k + b), Lx.ld, 0.0, Yx.p, Yx.ld, st.p, "gemm_lrband");
            gemm(ctx, false, false, m, c + 16, 16, -1.0, Vp.p, Vp.ld, Yx.p, Yx.ld, 1.0, Lx.p + (k + b), Lx.ld, st.p, "gemm_lrband");
            k += b; ++np; ++issued;
        }
        AdiState h;
        ctx_fetch(ctx, st.p, sizeof(int) * 4 + sizeof(double) * 2, &h);
        if (h.done) { J = h.iters; finished = true; }
        else if (np >= maxp) {
            // every column of L consumed: save the last panel's band blocks and stop
            hipLaunchKernelGGL(k_lr_rows, dim3(1), dim3(256), 0, ctx->stream, 0, k, Lw.p, Lw.ld, RB.p, PP.p, PP.ld, BS.p, st.p);
            J = k; finished = true;
        }
        chunk = 4;
    }
    np = J / b;
    if (hit != ctx->band_hint.end()) ctx->band_hint[hkey + 2000003L] = hit->second;
    ctx->band_hint[hkey] = np;
    out.J = J; out.npanels = np;
    out.D = Mat(ctx, J, J);
    const size_t tot = (size_t)J * J;
    if (tot) hipLaunchKernelGGL(k_lr_extract_band, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, ctx->stream, J, BS.p, out.D.p, out.D.ld);
    DRE_HIP(hipGetLastError());
    return out;
}

// Block reflector of ALL band panels at once.  With V = [V_0 ... V_{np-1}] and G = V'V the aggregated factor of
// H = Q_0 Q_1 ... Q_{np-1} = I - V Tbig V' has the block inverse  Tbig^-1 = blockdiag(T_p^-1) + blockstriu(G),  so
// M = Tbig R (R = V(1:J, :)') follows from a block back substitution that needs only the panel factors T_p themselves:
//   M_p = T_p (R_p - sum_{l > p} G_{p,l} M_l).
// One workgroup per 16 columns of M (they are independent); nb == 16.
__global__ __launch_bounds__(256) void k_blocktri_apply(int nr, int J, const double* __restrict__ G, int ldg, const double* __restrict__ Tp, int ldt,
                                                        const double* __restrict__ V, int ldv, double* __restrict__ M, int ldm) {
    // Every operand of a panel step (the 16 x (nr-k-16) block row of G, the panel factor T_p, the right-hand-side entry) is
    // prefetched into registers one step ahead and handed over through LDS: the recurrence itself never waits for L2.
    extern __shared__ double sh[];
    double* Msh = sh;                          // nr x 17
    double* Gs = sh + (size_t)nr * 17;         // 16 x (nr - 16), ld 17:  Gs[i + l * 17] = G(k + i, k + 16 + l)
    double* Ysh = Gs + (size_t)nr * 17;        // 16 x 17
    double* Tsm2 = Ysh + 16 * 17;              // 2 x (16 x 17): the panel factor, double buffered (read after the second barrier)
    const int tid = threadIdx.x, i = tid & 15, j = tid >> 4;
    const int j0 = blockIdx.x * 16;
    const bool colok = (j0 + j) < J;
    constexpr int NPRE = 16;                   // 16 * (nr - 16) / 256 <= 16  for nr <= 272
    double pre[NPRE], tpre, rpre;
    auto prefetch = [&](int k) {
        const int ncols = nr - k - 16;
#pragma unroll
        for (int q = 0; q < NPRE; ++q) {
            const int id = tid + q * 256;
            pre[q] = (id < 16 * ncols) ? G[(size_t)(k + (id & 15)) + (size_t)(k + 16 + (id >> 4)) * ldg] : 0.0;
        }
        tpre = Tp[i + (size_t)(k + j) * ldt];
        rpre = colok ? V[(size_t)(j0 + j) + (size_t)(k + i) * ldv] : 0.0;
    };
    prefetch(nr - 16);
    for (int k = nr - 16; k >= 0; k -= 16) {
        const int ncols = nr - k - 16;
#pragma unroll
        for (int q = 0; q < NPRE; ++q) {
            const int id = tid + q * 256;
            if (id < 16 * ncols) Gs[(id & 15) + (id >> 4) * 17] = pre[q];
        }
        double* Tsm = Tsm2 + ((k >> 4) & 1) * 16 * 17;
        Tsm[i + j * 17] = tpre;
        double a0 = rpre, a1 = 0.0, a2 = 0.0, a3 = 0.0;
        __syncthreads();
        if (k >= 16) prefetch(k - 16);
        int l = 0;
        for (; l + 3 < ncols; l += 4) {
            a0 -= Gs[i + l * 17] * Msh[(k + 16 + l) * 17 + j];
            a1 -= Gs[i + (l + 1) * 17] * Msh[(k + 17 + l) * 17 + j];
            a2 -= Gs[i + (l + 2) * 17] * Msh[(k + 18 + l) * 17 + j];
            a3 -= Gs[i + (l + 3) * 17] * Msh[(k + 19 + l) * 17 + j];
        }
        for (; l < ncols; ++l) a0 -= Gs[i + l * 17] * Msh[(k + 16 + l) * 17 + j];
        Ysh[i * 17 + j] = (a0 + a1) + (a2 + a3);
        __syncthreads();
        double m = 0.0;
        for (int t = i; t < 16; ++t) m += Tsm[i + t * 17] * Ysh[t * 17 + j];
        Msh[(k + i) * 17 + j] = m;
    }
    __syncthreads();
    if (colok)
        for (int r = i; r < nr; r += 16) M[r + (size_t)(j0 + j) * ldm] = Msh[r * 17 + j];
}

static Mat sym_band_basis_core(Ctx* ctx, const SymBand& sb);
Mat sym_band_basis(Ctx* ctx, const SymBand& sb) {
    Mat B = sym_band_basis_core(ctx, sb);
    if (!sb.V0.empty() && sb.J > 0) lead_rotate_back(ctx, sb.V0, sb.VT0, B);     // factor-form reduction with a warm start: Qb <- Q0 Qb
    return B;
}
static Mat sym_band_basis_core(Ctx* ctx, const SymBand& sb) {
    Mat B;
    if (!sb.B0.empty() && !sb.B0_used && sb.B0.rows == sb.q && sb.B0.cols == sb.J) { B = sb.B0; sb.B0_used = true; }
    else { B = Mat(ctx, sb.q, sb.J); set_identity(ctx, B, 1.0); }
    const int b = sb.nb;
    int np = sb.npanels;
    while (np > 0 && sb.q - (np - 1) * b - b < b) --np;        // panels that really hold reflectors
    const int nr = np * b;
    if (nr == 0) return B;
    if (nr <= 272) {
        // all panels as ONE block reflector:  Qb(:, 1:J) = [I; 0] - V (Tbig V(1:J, :)')   — 4 launches instead of 2 per panel
        Mat Vall = sb.V.view(0, 0, sb.q, nr);
        DRE_REQUIRE(b == 16, "sym_band_basis: panel width 16 expected");
        Mat G(ctx, nr, nr), M(ctx, nr, sb.J);
        if (sb.q <= 2048) gemm_thin(ctx, true, nr, nr, sb.q, 1.0, Vall.p, Vall.ld, Vall.p, Vall.ld, 0.0, G.p, G.ld, nullptr, "gemm_band");   // one launch, no slabs
        else gemm(ctx, true, false, 1.0, Vall, Vall, 0.0, G, nullptr, "gemm_band");
        {
            TimedScope ts(ctx, "blocktri", 8.0 * (nr * (double)nr / 2 + 2.0 * nr * sb.J), (double)nr * nr * sb.J);
            const size_t shm = ((size_t)2 * nr + 48) * 17 * sizeof(double);
            lds_attr(ctx, (const void*)k_blocktri_apply, 80 * 1024);
            hipLaunchKernelGGL(k_blocktri_apply, dim3((sb.J + 15) / 16), dim3(256), shm, ctx->stream, nr, sb.J, G.p, G.ld, sb.T.p, sb.T.ld, sb.V.p, sb.V.ld, M.p, M.ld);
        }
        gemm(ctx, false, false, -1.0, Vall, M, 1.0, B, nullptr, "gemm_band");
        return B;
    }
    // Qb = Q_0 Q_1 ... Q_{np-1};  Qb * [I; 0]: apply the last panel first;  Q_p B = B - (V T)(V' B)
    for (int p = np - 1; p >= 0; --p) {
        const int k = p * b, m = sb.q - k - b;
        Mat Vp = sb.V.view(k + b, k, m, b);
        Mat VTp = sb.VT.view(k + b, k, m, b);
        Mat B2 = B.view(k + b, 0, m, sb.J);
        Mat W(ctx, b, sb.J);
        gemm(ctx, true, false, 1.0, Vp, B2, 0.0, W, nullptr, "gemm_band");
        gemm(ctx, false, false, -1.0, VTp, W, 1.0, B2, nullptr, "gemm_band");
    }
    return B;
}

}  // namespace dre
