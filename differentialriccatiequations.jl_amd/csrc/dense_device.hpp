// Device-side helpers shared by the dense translation units (gemm.hip, dense.hip, qr_band.hip): wave / workgroup sums, the band reductions'
// termination tolerance, the MFMA accumulator type; and the one kernel of gemm.hip that dense.hip launches itself.
#pragma once
#include "dense.hpp"

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

// fixed-order sum of split-K slabs (gemm.hip)
__global__ void k_gemm_reduce(int M, int N, int splits, double alpha, const double* __restrict__ partial, double beta, double* __restrict__ C, int ldc,
                              const AdiState* st);

}  // namespace dre
