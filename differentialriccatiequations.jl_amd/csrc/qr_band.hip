// Dense f64 kernels for gfx950: blocked Householder QR (compact WY) and TSQR, the symmetric eigensolver, the two-sided band reductions
// (dense and factor form).
#include "dense_device.hpp"
#include <functional>
#include "profiling.hpp"
#include <atomic>
#include <chrono>

namespace dre {

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

// n > 128 (Z no longer fits one workgroup's LDS).  The generator is a scalar dependent chain — ~500 cycles per rotation in one GPU thread: 7.5 ms for
// the ~34 000 rotations of a 208 x 208 problem, wherever Z lives (a row-slab form with Z in LDS and the generator run redundantly per workgroup took the
// same 7.6 ms) — so it runs on the HOST (10 - 15 ns per rotation; d and e are 2 j doubles, and sym_eig synchronises for the reduction's dimension
// anyway), which logs every rotation; the device replays the log on row slabs of Z held in LDS: lane = row, the carried element in a register, one
// LDS read and one LDS write per rotation and row, (c, s) through scalar loads.  Same recurrences and deflation constants as k_tql.
struct QlSweep { int m, ilo; long off; };          // rotations i = m - 1 ... ilo of one implicit QL sweep, (c, s) pairs at log[2 (off + m - 1 - i)]
static bool host_tql_log(int n, std::vector<double>& d, std::vector<double>& e, double anorm, double deflate, std::vector<double>& log, std::vector<QlSweep>& sweeps) {
    const double eps = 2.220446049250313e-16, abstiny = deflate * eps * anorm;
    e.resize(n, 0.0);
    e[n - 1] = 0.0;
    int l = 0, iter = 0;
    while (l < n) {
        int m = l;
        for (; m < n - 1; ++m) {
            const double em = std::fabs(e[m]);
            if (em <= eps * (std::fabs(d[m]) + std::fabs(d[m + 1])) || em <= abstiny) break;
        }
        if (m == l) { ++l; iter = 0; continue; }
        if (iter >= 80) return false;
        ++iter;
        const double dl = d[l], el = e[l];
        double g = (d[l + 1] - dl) / (2.0 * el);
        double r = std::sqrt(g * g + 1.0);
        g = d[m] - dl + el / (g + (g >= 0.0 ? r : -r));
        double s = 1.0, c = 1.0, p = 0.0;
        const long off = (long)(log.size() / 2);
        int i, ilo = l;
        bool broke = false;
        for (i = m - 1; i >= l; --i) {
            const double f = s * e[i], b = c * e[i];
            const double h = f * f + g * g;
            if (h == 0.0) { e[i + 1] = 0.0; d[i + 1] -= p; e[m] = 0.0; broke = true; break; }
            const double rr = std::sqrt(h);
            e[i + 1] = rr;
            s = f / rr; c = g / rr;
            g = d[i + 1] - p;
            r = (d[i] - g) * s + 2.0 * c * b;
            p = s * r;
            d[i + 1] = g + p;
            g = c * r - b;
            log.push_back(c); log.push_back(s);
        }
        if (broke) ilo = i + 1;
        else { d[l] -= p; e[l] = g; e[m] = 0.0; }
        if (m - 1 >= ilo) sweeps.push_back({m, ilo, off});
    }
    return true;
}
__global__ __launch_bounds__(64) void k_tql_replay(int n, int rows, const double* __restrict__ log, const QlSweep* __restrict__ sweeps, int nsweeps,
                                                   double* __restrict__ Zg, int ldzg) {
    extern __shared__ double Zs[];          // rows x n, Zs[k + i * rows]
    const int k = threadIdx.x, r0 = blockIdx.x * rows, nr = min(rows, n - r0);
    for (int id = k; id < rows * n; id += 64) { const int kk = id % rows, i = id / rows; Zs[id] = (r0 + kk == i) ? 1.0 : 0.0; }
    __syncthreads();
    const int vz = (int)(__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) >> 6);          // lane id >> 6 = 0
    if (k < nr) {
        for (int t = 0; t < nsweeps; ++t) {
            // (the sweep's parameters through scalar registers)
            const int m = __builtin_amdgcn_readfirstlane(sweeps[t].m), ilo = __builtin_amdgcn_readfirstlane(sweeps[t].ilo);
            const long off = ((long)__builtin_amdgcn_readfirstlane((int)(sweeps[t].off >> 32)) << 32) |
                             (unsigned long)(unsigned)__builtin_amdgcn_readfirstlane((int)(sweeps[t].off & 0xffffffffL));
            // the pairs through VECTOR loads (every lane the same address: one request): scalar loads share the LDS counter (lgkmcnt) and return out of
            // order, so every wait for an LDS operand would also wait for all pairs in flight; `vz` is zero, but not to the compiler
            const double* __restrict__ cs = log + 2 * off + vz;
            double zi1 = Zs[k + (size_t)m * rows];
            // eight rotations per batch, two batches in registers: while the dependent chain of one batch runs, the operands of the next (the untouched
            // columns i - 8 ... i - 15 and their pairs) are in flight — one wave per workgroup has nothing else to hide its LDS and scalar-load latencies
            // behind.  (Column i - u is first written by rotation u + 1: the early reads see the values the sequential order sees.)
            double z[2][8], cc[2][8], ss[2][8];
            auto fetch = [&](int b, int i) {
                const int cnt = min(8, i - ilo + 1);
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (u < cnt) { z[b][u] = Zs[k + (size_t)(i - u) * rows]; cc[b][u] = cs[2 * u]; ss[b][u] = cs[2 * u + 1]; }
                cs += 2 * max(cnt, 0);
            };
            auto chain = [&](int b, int i) {
                const int cnt = min(8, i - ilo + 1);
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (u < cnt) { Zs[k + (size_t)(i - u + 1) * rows] = ss[b][u] * z[b][u] + cc[b][u] * zi1; zi1 = cc[b][u] * z[b][u] - ss[b][u] * zi1; }
            };
            int i = m - 1;
            if (i >= ilo) fetch(0, i);
            for (; i >= ilo; i -= 16) {
                if (i - 8 >= ilo) fetch(1, i - 8);
                chain(0, i);
                if (i - 8 >= ilo) {
                    if (i - 16 >= ilo) fetch(0, i - 16);
                    chain(1, i - 8);
                }
            }
            Zs[k + (size_t)ilo * rows] = zi1;
        }
    }
    __syncthreads();
    for (int id = k; id < rows * n; id += 64) { const int kk = id % rows, i = id / rows; if (kk < nr) Zg[(r0 + kk) + (size_t)i * ldzg] = Zs[id]; }
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

SymEig sym_eig(Ctx* ctx, Mat& S, double tolfac, bool want_eig, double abs_tol, bool tol_is_floor, double deflate) {
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
        } else if (j <= 2048) {
            // generator on the host, replay on row slabs of Z in LDS (k_tql_replay)
            std::vector<double> hd(j), he(j, 0.0), hlog;
            std::vector<QlSweep> hsw;
            DRE_HIP(hipMemcpyAsync(hd.data(), d.p, j * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
            if (j > 1) DRE_HIP(hipMemcpyAsync(he.data(), e.p, (j - 1) * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
            DRE_HIP(hipStreamSynchronize(ctx->stream));
            hlog.reserve(std::min<size_t>((size_t)4 * j * j, (size_t)4 << 20));
            const auto t_h0 = std::chrono::steady_clock::now();
            if (!host_tql_log(j, hd, he, hi.snorm, deflate, hlog, hsw)) throw Error(ERR_INTERNAL, "sym_eig: QL iteration did not converge");
            if (env_trace("compress"))
                std::fprintf(stderr, "[sym_eig] j=%d: %zu sweeps, %zu rotations on the host in %.2f ms (deflation at %.0e eps ||S||)\n", j, hsw.size(), hlog.size() / 2,
                             std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_h0).count(), deflate);
            DevArr<double> dlog(ctx, std::max<size_t>(hlog.size(), 2));
            DevArr<QlSweep> dsw(ctx, std::max<size_t>(hsw.size(), 1));
            if (!hlog.empty()) DRE_HIP(hipMemcpyAsync(dlog.p, hlog.data(), hlog.size() * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
            if (!hsw.empty()) DRE_HIP(hipMemcpyAsync(dsw.p, hsw.data(), hsw.size() * sizeof(QlSweep), hipMemcpyHostToDevice, ctx->stream));
            DRE_HIP(hipMemcpyAsync(dw.p, hd.data(), j * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
            int rows = 64;
            while (rows > 4 && (size_t)rows * j * sizeof(double) > (size_t)150 * 1024) rows >>= 1;
            lds_attr(ctx, (const void*)k_tql_replay, 150 * 1024);
            hipLaunchKernelGGL(k_tql_replay, dim3(ceil_div(j, rows)), dim3(64), (size_t)rows * j * sizeof(double), ctx->stream, j, rows, (const double*)dlog.p,
                               (const QlSweep*)dsw.p, (int)hsw.size(), out.Z.p, out.Z.ld);
            DRE_HIP(hipStreamSynchronize(ctx->stream));          // (the host vectors of the uploads die with this scope)
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
