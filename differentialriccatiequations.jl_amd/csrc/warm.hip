// Warm-started compression of the DENSE Riccati residual of the dense-X time loop (gdre.hip, ros1_dense_step): round 5.
//
// The reference compresses the warm-start residual of every Lyapunov solve from scratch (lyapunov/residual.jl:3-31 -> compress!, LDLt.jl:204-225).
// Rounds 1-4 did the same on the device: a two-sided band reduction of the n x n matrix Res, 3 dependent launches per 16 columns of rank
// (36 % of the device time at n = 371).  Between two Rosenbrock steps the RANGE of the residual hardly moves: what the previous step's
// basis Q misses of the new residual is a handful of directions (8 - 14 in the first steps of a SteelProfile run, 1 - 6 later) plus the
// rounding noise of forming Res.  So the compression of step i + 1 is a Rayleigh-Ritz step in the basis B = [Q, Z]:
//     [Y_p, Y_f, W] = Res [Om_p, Om_f, Q]       one thin product; Om_p: 16 fixed Gaussian probe columns, Om_f: 16 fixed Gaussian sketch columns
//     Z_1 = (I - QQ') Y_f,  Z = Z_1 C            16 fresh directions, whitened by a thresholded Cholesky factor of Z_1'Z_1 (k_warm_project:
//                                                last-arrival ticket) — only ROUGHLY orthonormal (kappa(Z_1) ~ 1e5): the defect is measured, not assumed away
//     W_2 = Res Z                                (k_warm_z)
//     [P_p, ., H_1, G_z, H_2] = B' [Y_p, Y_f, W, Z, W_2]      one thin product:  H = B'Res B,  G = B'B = [I, Q'Z; Z'Q, Z'Z]
//     G = L L',  M = L^-1 H L^-T = U diag(lam) U'              in the LDS of ONE workgroup (k_warm_small): block Cholesky (only the 16 x 16 Schur
//                                                complement is factorised), cyclic Jacobi — M is nearly diagonal, because Q is the previous eigenbasis
//     keep the J largest |lam| with  dropped^2 + off^2 <= budget;   R_full = B L^-T U (the qn leading eigen-directions: the next basis),
//     R = its first kl columns, T = diag(lam_J, 0);   probe:  est^2 = ||Y_p - B G^-1 B'Y_p||_F^2 / 16 ~ ||(I - P_B) Res||_F^2;
//     accepted iff  2 est^2 + dropped^2 <= tol^2                                                              (k_warm_finish: last-arrival ticket)
// Six launches and NO host read-back in front of the ADI chain: the chain is launched at the PREVIOUS step's width kl (R and T are zero padded
// beyond J), the host learns J with the chain's control block.  A rejected attempt sets the solve's `done` flag before its first iteration:
// nothing is applied to X and the step is redone with the full band reduction, which also gives the next step a fresh basis.
#include "dense.hpp"
#include "profiling.hpp"

namespace dre {

typedef double v4d __attribute__((ext_vector_type(4)));

__device__ inline double warm_wave_sum(double v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// sum over the 1024 threads of the workgroup, result in every thread (red: 17 doubles of LDS)
__device__ inline double warm_block_sum(double v, double* red) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    v = warm_wave_sum(v);
    if (lane == 0) red[wave] = v;
    __syncthreads();
    if (wave == 0) {
        double x = lane < 16 ? red[lane] : 0.0;
        x = warm_wave_sum(x);
        if (lane == 0) red[16] = x;
    }
    __syncthreads();
    const double r = red[16];
    __syncthreads();
    return r;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// k_warm_project: one workgroup per 16-row strip.  P_f = Q'Y_f (q x 16, every workgroup forms it: 3 us of MFMA against a launch boundary),
// Z_1[strip] = Y_f[strip] - Q[strip] P_f, the strip's share of Z_1'Z_1; the LAST workgroup sums the shares in fixed order, factorises
// (right-looking Cholesky, pivots at or below 1e-10 x the largest diagonal entry give a zero column of C) and writes C = L^-T.
// ---------------------------------------------------------------------------------------------------------------------------------
struct WarmProjectArgs {
    int n, q, rows_per_wg;        // rows_per_wg: a multiple of 16 (large n: at most 256 workgroups meet at the ticket)
    const double* Q; int ldq;
    const double* Yf; int ldy;
    const double* Pf; int ldpf;   // q x 16: Q'Y_f
    double* Z1; int ldz;
    double* slab;          // gridDim.x x 256
    int* ticket;
    double* Cw;            // 16 x 16, column-major: Z = Z_1 Cw
};
__global__ __launch_bounds__(256) void k_warm_project(WarmProjectArgs a) {
    __shared__ double Pf[64][17];
    __shared__ double Zs[16][17];
    __shared__ double Gs[16][17], Ls[16][17];
    __shared__ int last_sh;
    const int tid = threadIdx.x;
    const int n = a.n, q = a.q;
    for (int e = tid; e < q * 16; e += 256) { const int i = e % q, c = e / q; Pf[i][c] = a.Pf[i + (size_t)c * a.ldpf]; }
    __syncthreads();
    // Z_1 in strips of 16 rows: thread (row r = tid & 15, column c = tid >> 4); the workgroup's share of Z_1'Z_1 accumulates over its strips
    double gacc = 0.0;
    for (int sub = 0; sub < a.rows_per_wg; sub += 16) {
        const int row0 = blockIdx.x * a.rows_per_wg + sub;
        if (row0 >= n) break;
        {
            const int r = tid & 15, c = tid >> 4, row = row0 + r;
            double z = 0.0;
            if (row < n) {
                double s0 = a.Yf[row + (size_t)c * a.ldy], s1 = 0.0;
                int j = 0;
                for (; j + 1 < q; j += 2) { s0 -= a.Q[row + (size_t)j * a.ldq] * Pf[j][c]; s1 -= a.Q[row + (size_t)(j + 1) * a.ldq] * Pf[j + 1][c]; }
                if (j < q) s0 -= a.Q[row + (size_t)j * a.ldq] * Pf[j][c];
                z = s0 + s1;
                a.Z1[row + (size_t)c * a.ldz] = z;
            }
            Zs[r][c] = z;
        }
        __syncthreads();
        {
            const int i = tid & 15, j = tid >> 4;
            double s = 0.0;
#pragma unroll
            for (int r = 0; r < 16; ++r) s += Zs[r][i] * Zs[r][j];
            gacc += s;
        }
        __syncthreads();
    }
    a.slab[(size_t)blockIdx.x * 256 + tid] = gacc;
    __syncthreads();
    // (ONE release fence per workgroup, behind the barrier: on a multi-XCD part a device-scope fence writes the XCD's L2 back, and 256 of them
    // per workgroup made this kernel 33 us long)
    if (tid == 0) { __threadfence(); last_sh = (atomicAdd(a.ticket, 1) == (int)gridDim.x - 1) ? 1 : 0; if (last_sh) __threadfence(); }
    __syncthreads();
    if (!last_sh) return;
    {
        // the strips' shares in fixed order; the loads of a batch are all in flight before the first addition
        double s = 0.0;
        for (unsigned b0 = 0; b0 < gridDim.x; b0 += 16) {
            double v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const unsigned b = min(b0 + u, gridDim.x - 1);
                v[u] = __hip_atomic_load(&a.slab[(size_t)b * 256 + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) s += (b0 + u < gridDim.x) ? v[u] : 0.0;
        }
        Gs[tid & 15][tid >> 4] = s;
        Ls[tid & 15][tid >> 4] = 0.0;
    }
    __syncthreads();
    if (tid < 64) {
        // Right-looking Cholesky of the 16 x 16 Gram matrix by ONE wave (LDS operations of a wave execute in order: no workgroup barriers),
        // thresholded: pivots at or below 1e-10 x the largest diagonal entry are dead.  Lane l owns the entries (i, c) = (l & 15, (l >> 4) + 4 k).
        const int lane = tid;
        double mx = 0.0;
        for (int i = 0; i < 16; ++i) mx = fmax(mx, Gs[i][i]);
        const double floor_ = 1e-10 * mx;
        const int i = lane & 15, c0 = lane >> 4;
        for (int j = 0; j < 16; ++j) {
            const double piv = Gs[j][j];
            const bool live = piv > floor_ && piv > 0.0;
            const double inv = live ? 1.0 / sqrt(piv) : 0.0;
            const double gij = Gs[i][j] * inv;
            double upd[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) { const int c = c0 + 4 * k; upd[k] = (i > j && c > j) ? gij * (Gs[c][j] * inv) : 0.0; }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
#pragma unroll
            for (int k = 0; k < 4; ++k) { const int c = c0 + 4 * k; if (i > j && c > j) Gs[i][c] -= upd[k]; }
            if (lane < 16 && lane >= j) Ls[lane][j] = live ? Gs[lane][j] * inv : 0.0;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
        // C = L^-T restricted to the live pivots (a dead pivot gives a zero row and column): lane c < 16 forms row c of L^-1 — fully unrolled
        // with compile-time indices, so that x stays in registers (a run-time index put it into scratch memory: 15 us of dependent loads)
        if (lane < 16) {
            const int cc = lane;
            double x[16];
#pragma unroll
            for (int ii = 0; ii < 16; ++ii) x[ii] = 0.0;
            const double dcc = Ls[cc][cc];
            const double xcc = dcc > 0.0 ? 1.0 / dcc : 0.0;
#pragma unroll
            for (int ii = 0; ii < 16; ++ii) x[ii] = ii == cc ? xcc : 0.0;
#pragma unroll
            for (int j = 14; j >= 0; --j) {
                const double d = Ls[j][j];
                double s0 = 0.0, s1 = 0.0;
#pragma unroll
                for (int ii = j + 1; ii < 16; ii += 2) {
                    s0 += x[ii] * Ls[ii][j];                    // (x[ii] = 0 beyond cc, Ls is zero above the diagonal)
                    if (ii + 1 < 16) s1 += x[ii + 1] * Ls[ii + 1][j];
                }
                const double xj = (j < cc && d > 0.0 && dcc > 0.0) ? -(s0 + s1) / d : 0.0;
                if (j < cc) x[j] = xj;
            }
#pragma unroll
            for (int j = 0; j < 16; ++j) a.Cw[j + 16 * cc] = x[j];       // C[j][c] = (L^-1)[c][j]
        }
    }
    if (tid == 0) *a.ticket = 0;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// k_warm_z: one workgroup per 16-row strip.  Z = Z_1 C for ALL rows into LDS (n x 16), the strip's rows of Z to the basis buffer and to the
// right-hand block of the Gram product, W_2[strip] = Res[strip, :] Z.
// ---------------------------------------------------------------------------------------------------------------------------------
struct WarmZArgs {
    int n;
    const double* Z1; int ldz;
    const double* Cw;
    const double* Res; int ldres;
    double* Zb; int ldzb;      // Z inside the basis buffer
    double* Zy; int ldzy;      // Z inside the right-hand block
    double* W2; int ldw2;
};
__global__ __launch_bounds__(256) void k_warm_z(WarmZArgs a) {
    extern __shared__ double zsm[];       // n x 16, row-major with stride 17
    __shared__ double Cs[16][17];
    __shared__ double part[4][4][64];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lr = lane & 15, lk = lane >> 4;
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const int n = a.n;
    Cs[tid & 15][tid >> 4] = a.Cw[tid];
    __syncthreads();
    for (int r = tid; r < n; r += 256) {
        double z1[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) z1[j] = a.Z1[r + (size_t)j * a.ldz];
#pragma unroll 2
        for (int c = 0; c < 16; ++c) {
            double s0 = 0.0, s1 = 0.0;
#pragma unroll
            for (int j = 0; j < 16; j += 2) { s0 += z1[j] * Cs[j][c]; s1 += z1[j + 1] * Cs[j + 1][c]; }
            zsm[r * 17 + c] = s0 + s1;
        }
    }
    __syncthreads();
    const int row0 = blockIdx.x * 16;
    {
        const int r = tid & 15, c = tid >> 4, row = row0 + r;
        if (row < n) { const double z = zsm[row * 17 + c]; a.Zb[row + (size_t)c * a.ldzb] = z; a.Zy[row + (size_t)c * a.ldzy] = z; }
    }
    const int arow = min(row0 + lr, n - 1);
    const bool aok = row0 + lr < n;
    const int kst = (n + 3) >> 2, per = (kst + 3) >> 2, t0 = wv * per, t1 = min(kst, t0 + per);
    v4d acc = (v4d){0.0, 0.0, 0.0, 0.0};
    for (int tb = t0; tb < t1; tb += 24) {
        double av[24];
#pragma unroll
        for (int u = 0; u < 24; ++u) {
            const int kk = min(4 * min(tb + u, t1 - 1) + lk, n - 1);
            av[u] = a.Res[arow + (size_t)kk * a.ldres];
        }
#pragma unroll
        for (int u = 0; u < 24; ++u) {
            const int kk = 4 * (tb + u) + lk;
            const bool ok = (tb + u < t1) && kk < n;
            const double bv = ok ? zsm[kk * 17 + lr] : 0.0;
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64((ok && aok) ? av[u] : 0.0, bv, acc, 0, 0, 0);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) part[wave][r][lane] = acc[r];
    __syncthreads();
    const double v = ((part[0][wave][lane] + part[1][wave][lane]) + part[2][wave][lane]) + part[3][wave][lane];
    const int orow = row0 + lk + 4 * wave;
    if (orow < n) a.W2[orow + (size_t)lr * a.ldw2] = v;
}

// Z = Z_1 C (n x 16) for large n, where the rows do not fit one workgroup's LDS: a thread per row, into the basis buffer and the right-hand block
__global__ __launch_bounds__(256) void k_warm_zapply(int n, const double* __restrict__ Z1, int ldz, const double* __restrict__ Cw, double* __restrict__ Zb, int ldzb,
                                                     double* __restrict__ Zy, int ldzy) {
    __shared__ double Cs[16][17];
    Cs[threadIdx.x & 15][threadIdx.x >> 4] = Cw[threadIdx.x];
    __syncthreads();
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= n) return;
    double z1[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) z1[j] = Z1[r + (size_t)j * ldz];
#pragma unroll 2
    for (int c = 0; c < 16; ++c) {
        double s0 = 0.0, s1 = 0.0;
#pragma unroll
        for (int j = 0; j < 16; j += 2) { s0 += z1[j] * Cs[j][c]; s1 += z1[j + 1] * Cs[j + 1][c]; }
        const double z = s0 + s1;
        Zb[r + (size_t)c * ldzb] = z; Zy[r + (size_t)c * ldzy] = z;
    }
}
void warm_zapply(Ctx* ctx, int n, const Mat& Z1, const double* Cw, Mat& Zb, Mat& Zy) {
    hipLaunchKernelGGL(k_warm_zapply, dim3(ceil_div(n, 256)), dim3(256), 0, ctx->stream, n, (const double*)Z1.p, Z1.ld, Cw, Zb.p, Zb.ld, Zy.p, Zy.ld);
    DRE_HIP(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------------------------------------------
// k_warm_small: the small generalized eigenproblem in one workgroup.
// tols layout (8 doubles): [0] abstol of the Lyapunov solve (adi.jl:61-62), [1] truncation tolerance of the residual compression, [2] ||rhs||_F,
// [3] unused, [4] J (rank kept), [5] reject flag, [6] estimate of ||(I - P_B) Res||_F^2, [7] dropped^2 + off^2
// ---------------------------------------------------------------------------------------------------------------------------------
struct WarmSmallArgs {
    int q, m, kl, qn;                  // basis columns, m = q + 16 (q when there are no fresh directions), chain width, columns of the next basis
    const double* Cc; int ldc;         // m x (64 + q): [P_p | . | B'W | B'Z | B'W_2]
    const double* parts; int nparts; double reltol, abstol, frac;
    double budget_frac;                // share of tol^2 the truncation may use
    int max_sweeps;                    // Jacobi sweeps per call
    int mode;                          // 1: stop behind the set-up and hand M = L^-1 H L^-T (m x m) and L^-T (m x m) over (wide bases: band reduction by the caller)
                                       // 2: eigenvectors of the symmetric matrix Mout (m x m) by decreasing |eigenvalue| into Uc (m x m): no tolerances, no deflation
    double* Mout; int ldm;
    double* LTout; int ldl;
    double* tols;
    double* Uc; int ldu;               // m x qn: coefficients of the next basis in B
    double* T; int ldt;                // kl x kl
    double* Cp; int ldp;               // m x 16: G^-1 B'Y_p
    int* ticket;
};
// sum over the NT threads of the workgroup, result in every thread (red: 17 doubles of LDS)
template <int NT>
__device__ inline double warm_block_sum_t(double v, double* red) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    v = warm_wave_sum(v);
    if (lane == 0) red[wave] = v;
    __syncthreads();
    if (wave == 0) {
        double x = lane < NT / 64 ? red[lane] : 0.0;
        x = warm_wave_sum(x);
        if (lane == 0) red[16] = x;
    }
    __syncthreads();
    const double r = red[16];
    __syncthreads();
    return r;
}
// C = op(X) Y for m x m matrices in LDS (leading dimension LD, zero beyond m up to the next multiple of 16) on the matrix cores: one 16 x 16
// tile per wave and pass.  TX: op(X) = X'.  The caller separates reads and writes of a buffer by barriers.
template <int LD, bool TX>
__device__ __forceinline__ void warm_lds_mm(int m, const double (*X)[LD], const double (*Y)[LD], double (*C)[LD], int nwaves) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, lr = lane & 15, lk = lane >> 4;
    const int mt = (m + 15) >> 4, ks = (m + 3) >> 2;
    for (int tile = wave; tile < mt * mt; tile += nwaves) {
        const int ti = tile / mt, tj = tile - ti * mt;
        v4d acc = (v4d){0.0, 0.0, 0.0, 0.0};
        for (int t = 0; t < ks; ++t) {
            const int kk = 4 * t + lk;
            const double av = TX ? X[kk][16 * ti + lr] : X[16 * ti + lr][kk];
            const double bv = Y[kk][16 * tj + lr];
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) C[16 * ti + lk + 4 * r][16 * tj + lr] = acc[r];
    }
}
template <int NT, int MM>
__global__ __launch_bounds__(NT) void k_warm_small(WarmSmallArgs a) {
    constexpr int WARM_LD = MM + 1;
    constexpr int WARM_MMAX = MM;
    extern __shared__ double wsm[];
    double (*A)[WARM_LD] = (double (*)[WARM_LD])wsm;
    double (*W)[WARM_LD] = (double (*)[WARM_LD])(wsm + WARM_MMAX * WARM_LD);
    __shared__ double red[17];
    __shared__ double rc[40], rs[40], lam[WARM_MMAX];
    __shared__ int rp[40], rq[40], pos[WARM_MMAX];
    __shared__ double Gzs[WARM_MMAX][17], Pps[WARM_MMAX][17], Tmp[WARM_MMAX < 32 ? 32 : WARM_MMAX][17], Xm[16][65], S22[16][17], Li[16][17];
    __shared__ int J_sh, na_sh;
    __shared__ double defl_sh, rho[WARM_MMAX];
    __shared__ int ord[WARM_MMAX], act[WARM_MMAX];
    const int tid = threadIdx.x, q = a.q, m = a.m, nf = m - q;
    int ns_bad = 0;
    const long long t_begin = wall_clock64();
    const bool eig_only = a.mode == 2;
    // tolerances of the Lyapunov solve (what k_band_init's job does on the cold path)
    // (the partial sums are requested here and reduced behind the loads of H: their round trip hides behind those)
    double s = 0.0;
    if (!eig_only) for (int i = tid; i < a.nparts; i += NT) s += a.parts[i];
    // H = B'Res B (symmetrised) into A, identity into W, B'Z and B'Y_p into LDS
    if (eig_only) {
        for (int e = tid; e < MM * MM; e += NT) {
            const int i = e % MM, j = e / MM;
            A[i][j] = (i < m && j < m) ? 0.5 * (a.Mout[i + (size_t)j * a.ldm] + a.Mout[j + (size_t)i * a.ldm]) : 0.0;
            W[i][j] = i == j ? 1.0 : 0.0;
        }
    } else {
        const double* __restrict__ Hq = a.Cc + (size_t)32 * a.ldc;                // columns of B'W
        const double* __restrict__ H2 = a.Cc + (size_t)(48 + q) * a.ldc;          // B'W_2
        for (int e = tid; e < MM * MM; e += NT) {
            const int i = e % MM, j = e / MM;
            double v = 0.0;
            if (i < m && j < m) {
                const double hij = j < q ? Hq[i + (size_t)j * a.ldc] : H2[i + (size_t)(j - q) * a.ldc];
                const double hji = i < q ? Hq[j + (size_t)i * a.ldc] : H2[j + (size_t)(i - q) * a.ldc];
                v = 0.5 * (hij + hji);
            }
            A[i][j] = v;
            W[i][j] = i == j ? 1.0 : 0.0;
        }
        const double* __restrict__ Gz = a.Cc + (size_t)(32 + q) * a.ldc;          // B'Z
        for (int e = tid; e < m * 16; e += NT) {
            const int i = e % m, c = e / m;
            Gzs[i][c] = nf > 0 ? Gz[i + (size_t)c * a.ldc] : 0.0;
            Pps[i][c] = a.Cc[i + (size_t)c * a.ldc];
        }
    }
    s = warm_block_sum_t<NT>(s, red);          // (ends with a barrier: A, W, Gzs, Pps are complete behind it)
    const double nc = sqrt(s), at = a.abstol >= 0.0 ? a.abstol : a.reltol * nc, tolc = eig_only ? 0.0 : a.frac * at;
    if (nf > 0) {
        // G = B'B = [I, G12; G12', G22] = L L' with L = [I, 0; G12', L22], L22 L22' = G22 - G12'G12 (16 x 16).  M = L^-1 H L^-T with
        // L^-1 = [I, 0; X, Li], Li = L22^-1, X = -Li G12'.  A dead direction (zero column of Z) gets a unit pivot and stays decoupled.
        const int i16 = tid & 15, c16 = (tid >> 4) & 15;
        double sc = 0.0;
        if (tid < 256) {
            sc = 0.5 * (Gzs[q + i16][c16] + Gzs[q + c16][i16]);
            double s2 = 0.0;
            for (int r = 0; r < q; ++r) s2 += Gzs[r][i16] * Gzs[r][c16];
            sc -= s2;
            if (!(Gzs[q + i16][i16] > 1e-30) || !(Gzs[q + c16][c16] > 1e-30)) sc = (i16 == c16) ? 1.0 : 0.0;
            S22[i16][c16] = sc; Li[i16][c16] = 0.0;
        }
        __syncthreads();
        // N = S^(-1/2) (symmetric) by Newton-Schulz,  N <- N (3 I - S N^2) / 2  from N = I: S = I + E with a small E (Z is orthonormal up to the
        // loss of its single Cholesky-QR pass), quadratic convergence; six iterations reach 1e-16 from ||E|| = 0.5.  Li holds N.
        if (tid < 256) Li[i16][c16] = i16 == c16 ? 1.0 : 0.0;
        __syncthreads();
        double ns_res = 0.0;
        for (int itn = 0; itn < 6; ++itn) {
            double v = 0.0;
            if (tid < 256) {      // Tmp16 = S N
#pragma unroll
                for (int r = 0; r < 16; ++r) v += S22[i16][r] * Li[r][c16];
                Tmp[i16][c16] = v;
            }
            __syncthreads();
            double w = 0.0;
            if (tid < 256) {      // R = 3 I - N (S N)
#pragma unroll
                for (int r = 0; r < 16; ++r) w += Li[i16][r] * Tmp[r][c16];
                ns_res = fabs((i16 == c16 ? 1.0 : 0.0) - w);
                w = (i16 == c16 ? 3.0 : 0.0) - w;
                Tmp[16 + i16][c16] = w;
            }
            __syncthreads();
            double nn = 0.0;
            if (tid < 256) {
#pragma unroll
                for (int r = 0; r < 16; ++r) nn += Li[i16][r] * Tmp[16 + r][c16];
                nn *= 0.5;
            }
            // (the residual belongs to the iterate BEFORE this update: 1e-8 there is 1e-16 after it)
            const int unconv = __syncthreads_or(tid < 256 && !(ns_res < 1e-8));
            if (tid < 256) Li[i16][c16] = nn;
            __syncthreads();
            ns_bad = unconv ? 1 : 0;
            if (!unconv) break;
        }
        {
            const double v = tid < 256 ? 0.5 * (Li[i16][c16] + Li[c16][i16]) : 0.0;
            __syncthreads();
            if (tid < 256) Li[i16][c16] = v;
            __syncthreads();
        }
        for (int e = tid; e < 16 * q; e += NT) {      // X = -Li G12'   (16 x q)
            const int i = e & 15, c = e >> 4;
            double s2 = 0.0;
            for (int r = 0; r < 16; ++r) s2 += Li[i][r] * Gzs[c][r];
            Xm[i][c] = -s2;
        }
        __syncthreads();
        for (int e = tid; e < 16 * m; e += NT) {      // bottom rows of T1 = L^-1 H:  X H[:q, :] + Li H[q:, :]   (16 x m)
            const int i = e & 15, c = e >> 4;
            double s2 = 0.0, s3 = 0.0;
            int r = 0;
            for (; r + 1 < q; r += 2) { s2 += Xm[i][r] * A[r][c]; s3 += Xm[i][r + 1] * A[r + 1][c]; }
            if (r < q) s2 += Xm[i][r] * A[r][c];
            for (r = 0; r < 16; ++r) s3 += Li[i][r] * A[q + r][c];
            Tmp[c][i] = s2 + s3;
        }
        __syncthreads();
        for (int e = tid; e < 16 * m; e += NT) { const int i = e & 15, c = e >> 4; A[q + i][c] = Tmp[c][i]; }
        __syncthreads();
        for (int e = tid; e < 16 * m; e += NT) {      // right columns of M = T1 L^-T:  T1[:, :q] X' + T1[:, q:] Li'   (m x 16)
            const int c = e & 15, r = e >> 4;
            double s2 = 0.0, s3 = 0.0;
            int j = 0;
            for (; j + 1 < q; j += 2) { s2 += A[r][j] * Xm[c][j]; s3 += A[r][j + 1] * Xm[c][j + 1]; }
            if (j < q) s2 += A[r][j] * Xm[c][j];
            for (j = 0; j < 16; ++j) s3 += A[r][q + j] * Li[c][j];
            Tmp[r][c] = s2 + s3;
        }
        __syncthreads();
        for (int e = tid; e < 16 * m; e += NT) { const int c = e & 15, r = e >> 4; A[r][q + c] = Tmp[r][c]; }
        __syncthreads();
        for (int e = tid; e < m * 16; e += NT) {      // symmetrise (only the last 16 rows / columns changed)
            const int c = e & 15, r = e >> 4;
            if (r < q + c) { const double v = 0.5 * (A[r][q + c] + A[q + c][r]); A[r][q + c] = v; A[q + c][r] = v; }
        }
        __syncthreads();
    }
    // C_p = G^-1 B'Y_p = L^-T (L^-1 P):  v = L^-1 P (top = P_top, bottom = X P_top + N P_bot),  C_p = (top = v_top + X' v_bot, bottom = N v_bot)
    if (eig_only) {
    } else if (nf > 0) {
        if (tid < 256) {
            const int i = tid & 15, c = tid >> 4;
            double s2 = 0.0;
            for (int r = 0; r < q; ++r) s2 += Xm[i][r] * Pps[r][c];
            for (int r = 0; r < 16; ++r) s2 += Li[i][r] * Pps[q + r][c];
            Tmp[i][c] = s2;           // v_bot
        }
        __syncthreads();
        for (int e = tid; e < m * 16; e += NT) {
            const int i = e % m, c = e / m;
            double v;
            if (i < q) { v = Pps[i][c]; for (int r = 0; r < 16; ++r) v += Xm[r][i] * Tmp[r][c]; }
            else { v = 0.0; const int ii = i - q; for (int r = 0; r < 16; ++r) v += Li[r][ii] * Tmp[r][c]; }
            a.Cp[i + (size_t)c * a.ldp] = v;
        }
    } else {
        for (int e = tid; e < m * 16; e += NT) { const int i = e % m, c = e / m; a.Cp[i + (size_t)c * a.ldp] = Pps[i][c]; }
    }
    __syncthreads();
    if (a.mode == 1) {
        for (int e = tid; e < m * m; e += NT) {
            const int i = e % m, j = e / m;
            a.Mout[i + (size_t)j * a.ldm] = A[i][j];
            double lt;                                           // L^-T = [I, X'; 0, N]
            if (j < q) lt = i == j ? 1.0 : 0.0;
            else if (nf == 0) lt = i == j ? 1.0 : 0.0;
            else lt = i < q ? Xm[j - q][i] : Li[i - q][j - q];
            a.LTout[i + (size_t)j * a.ldl] = lt;
        }
        if (tid == 0) {
            a.tols[0] = at; a.tols[1] = tolc; a.tols[2] = nc; a.tols[3] = sqrt(a.budget_frac) * tolc;        // [3]: the band reduction's tolerance
            a.tols[4] = 0.0; a.tols[5] = (ns_bad || !(nc == nc)) ? 1.0 : 0.0; a.tols[6] = 0.0; a.tols[7] = a.budget_frac * tolc * tolc;
            *a.ticket = 0;
        }
        return;
    }
    double f = 0.0;
    for (int e = tid; e < m * m; e += NT) { const double v = A[e % m][e / m]; f += v * v; }
    const double fro2 = warm_block_sum_t<NT>(f, red);
    // (mode 2: a BASIS is wanted, not eigenvalues: the iteration stops at a relative 1e-10 of the norm)
    const double budget = a.budget_frac * tolc * tolc;
    // (the off-diagonal mass the iteration may leave: T stays dense and the truncation below accounts for every entry it drops, so this only decides
    // how sharply the diagonal orders the coordinates — a quarter of the truncation budget instead of 1 % of tol^2 saves a third of the rounds)
    const double thr2 = eig_only ? 1e-20 * fro2 : fmax(0.25 * budget, 64.0 * 4.930380657631324e-32 * fro2);
    // Deflation: rows whose norms add up to less than a quarter of the truncation budget (x 2: a dropped row takes its column along) leave the
    // eigenproblem untouched — they are the noise directions at the bottom of the spectrum and most of the fresh directions; what is left is a
    // handful of coordinates (the dominant eigen-directions and what really couples to them)
    if (tid < m) {
        double s2 = 0.0;
        for (int j = 0; j < m; ++j) { const double v = A[tid][j]; s2 += v * v; }
        rho[tid] = s2;
    }
    __syncthreads();
    int myrank = 0;                               // position of coordinate tid in the ascending order of the row norms
    if (tid < m) {
        const double ri = rho[tid];
        int r = 0;
        for (int j = 0; j < m; ++j) { const double rj = rho[j]; r += (rj < ri || (rj == ri && j < tid)) ? 1 : 0; }
        ord[r] = tid;
        myrank = r;
    }
    __syncthreads();
    if (m <= 64) {
        // one wave: prefix sums of the sorted row norms by shuffles (fixed association), the deflated set is a prefix of the order, the active
        // coordinates are compacted by ballot — the serial form of this block was 5 us of dependent LDS reads on one lane
        if (tid < 64) {
            const double r2 = tid < m ? rho[ord[tid]] : 0.0;
            double cum = r2;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const double t = __shfl_up(cum, o, 64); if (tid >= o) cum += t; }
            const bool in = tid < m && 2.0 * cum <= 0.25 * budget;
            const unsigned long long inb = __ballot(in);
            // (the sums never decrease: `in` is a prefix of the order unless a NaN sits in it — then everything stays active)
            int nN = __popcll(inb);
            if (nN < 64 && (inb >> nN) != 0ull) nN = 0;
            const double cumN = __shfl(cum, nN > 0 ? nN - 1 : 0, 64);
            const bool active = tid < m && myrank >= nN;
            const unsigned long long ab = __ballot(active);
            if (tid < m) pos[tid] = active ? 1 : 0;
            if (active) act[__popcll(ab & ((1ull << tid) - 1ull))] = tid;
            if (tid == 0) { na_sh = __popcll(ab); defl_sh = nN > 0 ? 2.0 * cumN : 0.0; }
        }
    } else if (tid == 0) {
        double cum = 0.0;
        int nN = 0;
        for (int k = 0; k < m; ++k) {
            const double r2 = rho[ord[k]];
            if (2.0 * (cum + r2) <= 0.25 * budget) { cum += r2; ++nN; } else break;
        }
        for (int k = 0; k < m; ++k) pos[ord[k]] = k >= nN ? 1 : 0;        // pos: active flag for now
        int na = 0;
        for (int i = 0; i < m; ++i) if (pos[i]) act[na++] = i;
        na_sh = na; defl_sh = 2.0 * cum;
    }
    __syncthreads();
    const int na = na_sh, nae = na + (na & 1), half = nae >> 1;
    const double defl2 = defl_sh;
    // compact copy of the active block (through W), identity into W
    for (int e = tid; e < nae * nae; e += NT) {
        const int x = e % nae, y = e / nae;
        W[x][y] = (x < na && y < na) ? A[act[x]][act[y]] : 0.0;
    }
    __syncthreads();
    for (int e = tid; e < nae * nae; e += NT) { const int x = e % nae, y = e / nae; A[x][y] = W[x][y]; }
    __syncthreads();
    for (int e = tid; e < nae * nae; e += NT) { const int x = e % nae, y = e / nae; W[x][y] = x == y ? 1.0 : 0.0; }
    __syncthreads();
    // Cyclic Jacobi on the active block, round-robin pairing (the pairs of a round are disjoint).  A pair whose off-diagonal entry is at or below
    // sqrt(thr2) / m is not rotated (all such entries together stay within thr2); a round without rotations costs one barrier.  A round's
    // rotations are applied to A as 2 x 2 blocks  J_i' A[{p_i,q_i}, {p_j,q_j}] J_j  (one thread per pair of pairs: no entry is touched by two
    // threads) and to the columns of W.
    const double skip = sqrt(thr2) / (double)m;
    int hsh = 0;                                  // half <= 2^hsh
    while ((1 << hsh) < half) ++hsh;
    const int hmask = (1 << hsh) - 1;
    double off2 = 0.0;
    bool conv = false;
    const long long t_setup = wall_clock64();
    const long long c_setup = clock64();
    int nsweeps = 0, nactive = 0;
    const int maxsw = a.max_sweeps;
    for (int sweep = 0; sweep <= maxsw; ++sweep) {
        double o = 0.0;
        for (int e = tid; e < na * na; e += NT) {
            const int i = e % na, j = e / na;
            if (i != j) { const double v = A[i][j]; o += v * v; }
        }
        off2 = warm_block_sum_t<NT>(o, red);
        if (off2 <= thr2) { conv = true; break; }
        if (sweep == maxsw) break;
        ++nsweeps;
        for (int r = 0; r < nae - 1; ++r) {
            // pair b of round r: (nae - 1, r) for b = 0, else (r + b, r - b) mod (nae - 1); every thread derives the indices it needs itself, so that
            // the rotation table holds c and s only and all loads of a phase are independent of each other
            auto pair_of = [&](int b, int& p, int& qq) {
                if (b == 0) { p = nae - 1; qq = r; }
                else { p = r + b; if (p >= nae - 1) p -= nae - 1; qq = r - b; if (qq < 0) qq += nae - 1; }
                if (p > qq) { const int t = p; p = qq; qq = t; }
            };
            int mine = 0;
            if (tid < half) {
                int p, qq;
                pair_of(tid, p, qq);
                const double apq = A[p][qq], app = A[p][p], aqq = A[qq][qq];
                double c = 1.0, sn = 0.0;
                if (fabs(apq) > skip) {
                    // the ANGLE may be approximate (hardware reciprocal / square root without refinement: the pair's entry drops by 1e-8 instead of
                    // to zero); c and s are exact to rounding for the t they use, so the rotation stays orthogonal
                    const double tau = (aqq - app) * __builtin_amdgcn_rcp(2.0 * apq);
                    const double t0 = __builtin_amdgcn_rcp(fabs(tau) + __builtin_amdgcn_sqrt(1.0 + tau * tau));
                    const double t = tau >= 0.0 ? t0 : -t0;
                    if (t == t && fabs(t) <= 1.0) {
                        const double x = 1.0 + t * t;
                        double y = __builtin_amdgcn_rsq(x);
                        y = y * (1.5 - 0.5 * x * y * y);
                        y = y * (1.5 - 0.5 * x * y * y);
                        c = y; sn = t * y;
                    }
                }
                rc[tid] = c; rs[tid] = sn;        // (index na for odd na is the zero padding row / column: never rotated)
                mine = sn != 0.0 ? 1 : 0;
            }
            const int any = __syncthreads_or(mine);
            if (!any) continue;
            ++nactive;
            for (int e0 = tid; e0 < (half << hsh) || e0 < (na << hsh); e0 += NT) {
                // one block of A and one entry pair of W per thread and pass: everything is loaded before anything is stored
                const int bi = e0 >> hsh, bj = e0 & hmask;
                const bool blk = bi < half && bj < half, wit = bi < na && bj < half;
                int pi = 0, qi = 0, pj = 0, qj = 0;
                if (bj < half) pair_of(bj, pj, qj);
                if (blk) pair_of(bi, pi, qi);
                const double cj = bj < half ? rc[bj] : 1.0, sj = bj < half ? rs[bj] : 0.0;
                const double ci = blk ? rc[bi] : 1.0, si = blk ? rs[bi] : 0.0;
                double a00 = 0.0, a01 = 0.0, a10 = 0.0, a11 = 0.0, wx = 0.0, wy = 0.0;
                if (blk) { a00 = A[pi][pj]; a01 = A[pi][qj]; a10 = A[qi][pj]; a11 = A[qi][qj]; }
                if (wit) { wx = W[bi][pj]; wy = W[bi][qj]; }
                if (blk && (si != 0.0 || sj != 0.0)) {
                    const double b00 = ci * a00 - si * a10, b01 = ci * a01 - si * a11, b10 = si * a00 + ci * a10, b11 = si * a01 + ci * a11;
                    A[pi][pj] = cj * b00 - sj * b01; A[pi][qj] = sj * b00 + cj * b01;
                    A[qi][pj] = cj * b10 - sj * b11; A[qi][qj] = sj * b10 + cj * b11;
                }
                if (wit && sj != 0.0) { W[bi][pj] = cj * wx - sj * wy; W[bi][qj] = sj * wx + cj * wy; }
            }
            __syncthreads();
        }
    }
    const long long t_jac = wall_clock64();
    const long long c_jac = clock64();
    if (eig_only) {
        if (tid < na) {
            const double li = fabs(A[tid][tid]);
            int r = 0;
            for (int j = 0; j < na; ++j) { const double lj = fabs(A[j][j]); r += (lj > li || (lj == li && j < tid)) ? 1 : 0; }
            pos[r] = tid;
        }
        __syncthreads();
        // columns: the active coordinates by decreasing |diagonal|, then the (exactly zero) deflated rows as unit vectors
        for (int e = tid; e < m * m; e += NT) {
            const int i = e % m, c = e / m;
            double v = 0.0;
            if (c < na) { for (int x = 0; x < na; ++x) if (act[x] == i) v = W[x][pos[c]]; }
            else v = (ord[m - 1 - c] == i) ? 1.0 : 0.0;
            a.Uc[i + (size_t)c * a.ldu] = v;
        }
        return;
    }
    // The rotated block stays DENSE in T, so an unconverged sweep costs nothing in accuracy: with the coordinates ordered by decreasing |diagonal|
    // (pos), keeping the leading J of them drops  ||A||_F^2 - ||A[K, K]||_F^2 = sum_{c >= J} g[c]  exactly,  g[c] = a_cc^2 + 2 sum_{c' < c} a_cc'^2.  J = the fewest coordinates that leave dropped^2 + deflated^2 within the budget.
    if (tid < na) {
        const double li = fabs(A[tid][tid]);
        int r = 0;
        for (int j = 0; j < na; ++j) { const double lj = fabs(A[j][j]); r += (lj > li || (lj == li && j < tid)) ? 1 : 0; }
        pos[r] = tid;
    }
    __syncthreads();
    if (tid < na) {
        const int pc = pos[tid];
        double g = A[pc][pc] * A[pc][pc], g2 = 0.0;
        for (int c2 = 0; c2 < tid; ++c2) { const double v = A[pc][pos[c2]]; g2 += v * v; }
        lam[tid] = g + 2.0 * g2;
    }
    __syncthreads();
    if (tid == 0) {
        double d2 = defl2;
        int J = na;
        for (int c = na - 1; c >= 0; --c) {       // suffix sums, smallest terms first: no cancellation
            if (d2 + lam[c] <= budget) { d2 += lam[c]; J = c; } else break;
        }
        int rej = ns_bad ? 1 : 0;
        if (J > a.kl) { J = a.kl; rej = 1; }
        if (!(nc == nc) || !(fro2 == fro2)) rej = 1;
        J_sh = J;
        a.tols[0] = at; a.tols[1] = tolc; a.tols[2] = nc; a.tols[3] = (double)nsweeps + 1e-4 * nactive + 1e-7 * na;
        a.tols[4] = (double)J; a.tols[5] = (double)rej; a.tols[6] = 0.0; a.tols[7] = d2;
        a.tols[8] = 0.01 * (double)(t_setup - t_begin); a.tols[9] = 0.01 * (double)(t_jac - t_setup); a.tols[10] = (double)(c_jac - c_setup);
        *a.ticket = 0;
    }
    __syncthreads();
    // T: the kept block, zero padded to the chain's width
    for (int e = tid; e < a.kl * a.kl; e += NT) {
        const int i = e % a.kl, j = e / a.kl;
        a.T[i + (size_t)j * a.ldt] = (i < J_sh && j < J_sh) ? 0.5 * (A[pos[i]][pos[j]] + A[pos[j]][pos[i]]) : 0.0;
    }
    __syncthreads();
    const int J = J_sh;
    // eigenvectors in the coordinates of M into A (m x qn): the active ones by decreasing |lambda|, then the deflated coordinates (unit vectors,
    // largest row norm first)
    for (int e = tid; e < m * a.qn; e += NT) { const int i = e % m, c = e / m; A[i][c] = 0.0; }
    __syncthreads();
    for (int e = tid; e < na * a.qn; e += NT) {
        const int x = e % na, c = e / na;
        if (c < na) A[act[x]][c] = W[x][pos[c]];
    }
    if (tid < a.qn && tid >= na) A[ord[m - 1 - tid]][tid] = 1.0;        // ord ascends in the row norm; its top na entries are the active rows
    __syncthreads();
    // coefficients of the next basis in B:  Uc = L^-T U'  (top = U'_top + X' U'_bot,  bottom = N U'_bot)
    for (int e = tid; e < m * a.qn; e += NT) {
        const int i = e % m, c = e / m;
        double v;
        if (nf == 0) v = A[i][c];
        else if (i < q) { v = A[i][c]; for (int r = 0; r < 16; ++r) v += Xm[r][i] * A[q + r][c]; }
        else { v = 0.0; const int ii = i - q; for (int r = 0; r < 16; ++r) v += Li[r][ii] * A[q + r][c]; }
        a.Uc[i + (size_t)c * a.ldu] = v;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// k_warm_finish: R = B Uc (n x qn) and the probe  ||Y_p - B C_p||_F^2  in one launch: one workgroup per 16-row strip, the waves take the column
// tiles (the last "tile" is the probe); the LAST workgroup to arrive sums the strips' partial sums in fixed order and takes the decision.
// ---------------------------------------------------------------------------------------------------------------------------------
struct WarmFinishArgs {
    int n, q, kl;              // q: columns of B, kl: columns of R
    const double* Q; int ldq;
    const double* Wk; int ldw;
    const double* Yp; int ldy;
    const double* Pp; int ldp;
    double* R; int ldr;
    double* slab; int* ticket; double* tols;
};
__global__ __launch_bounds__(256) void k_warm_finish(WarmFinishArgs a) {
    __shared__ double red[4];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lr = lane & 15, lk = lane >> 4;
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const int row0 = blockIdx.x * 16;
    const int ct = (a.kl + 15) >> 4, ksteps = (a.q + 3) >> 2;
    const int arow = min(row0 + lr, a.n - 1);
    const bool aok = row0 + lr < a.n;
    double ssq = 0.0;
    for (int j = wv; j <= ct; j += 4) {
        const bool probe = j == ct;
        const double* __restrict__ Bm = probe ? a.Pp : a.Wk;
        const int ldb = probe ? a.ldp : a.ldw, ncol = probe ? 16 : a.kl;
        const int col = (probe ? 0 : 16 * j) + lr;
        const bool cok = col < ncol;
        const int colc = cok ? col : 0;
        v4d acc = (v4d){0.0, 0.0, 0.0, 0.0};
        for (int t = 0; t < ksteps; ++t) {
            const int kk = 4 * t + lk;
            const bool ok = kk < a.q;
            const int kc = ok ? kk : 0;
            const double av = a.Q[arow + (size_t)kc * a.ldq], bv = Bm[kc + (size_t)colc * ldb];
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64((ok && aok) ? av : 0.0, (ok && cok) ? bv : 0.0, acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = row0 + lk + 4 * r;
            if (row < a.n && cok) {
                if (!probe) a.R[row + (size_t)col * a.ldr] = acc[r];
                else { const double d = a.Yp[row + (size_t)lr * a.ldy] - acc[r]; ssq += d * d; }
            }
        }
    }
    __shared__ int last_sh;
    ssq = warm_wave_sum(ssq);
    if (lane == 0) red[wave] = ssq;
    __syncthreads();
    if (tid == 0) {
        a.slab[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
        __threadfence();
        last_sh = (atomicAdd(a.ticket, 1) == (int)gridDim.x - 1) ? 1 : 0;
    }
    __syncthreads();
    if (!last_sh || wave != 0) return;
    __threadfence();
    double sum = 0.0;
    for (unsigned i0 = 0; i0 < gridDim.x; i0 += 64) {       // (fixed order: lane = strip, then the wave's reduction tree)
        const unsigned i = i0 + lane;
        sum += i < gridDim.x ? __hip_atomic_load(&a.slab[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
    }
    sum = warm_wave_sum(sum);
    if (lane != 0) return;
    const double est2 = sum / 16.0, tolc = a.tols[1], d2 = a.tols[7];
    a.tols[6] = est2;
    if (!(2.0 * est2 + d2 <= tolc * tolc)) a.tols[5] = 1.0;
    *a.ticket = 0;
}

// control words of the wide-basis variant (the band reduction of S ran with its own read-back): J, no rejection yet, the reduction's share
// of the error budget (its remainder is at most tol), the ticket
__global__ void k_warm_ctl(double* tols, int* ticket, int J) {
    tols[4] = (double)J; tols[6] = 0.0;          // ([5] and [7] were set by the small kernel: whitening breakdown, the reduction's share of the budget)
    *ticket = 0;
}
void warm_ctl(Ctx* ctx, double* tols, int* ticket, int J) {
    hipLaunchKernelGGL(k_warm_ctl, dim3(1), dim3(1), 0, ctx->stream, tols, ticket, J);
    DRE_HIP(hipGetLastError());
}
// rows of Z_1 per workgroup of k_warm_project: 16 while at most 64 workgroups result, else the multiple of 16 that gives 64
// (at most 256 workgroups meet at the ticket: the last arrival sums their 256-entry shares in batches of 16 loads; 64 workgroups of 320 rows each
// made the kernel 130-210 us long at n = 20209)
int warm_project_rows(int n) { const int strips = (n + 15) / 16; return strips <= 256 ? 16 : 16 * ((strips + 255) / 256); }
int warm_project_slabs(int n) { return ceil_div(n, warm_project_rows(n)); }
void warm_project(Ctx* ctx, int n, int q, const Mat& Q, const Mat& Yf, const Mat& Pf, Mat& Z1, double* slab, int* ticket, double* Cw) {
    DRE_REQUIRE(q >= 1 && q <= 64 && Q.rows == n && Q.cols >= q && Yf.rows == n && Yf.cols >= 16 && Z1.rows == n && Z1.cols >= 16 && Pf.rows >= q && Pf.cols >= 16,
                "warm_project: shapes");
    const int rpw = warm_project_rows(n);
    WarmProjectArgs a{n, q, rpw, Q.p, Q.ld, Yf.p, Yf.ld, Pf.p, Pf.ld, Z1.p, Z1.ld, slab, ticket, Cw};
    TimedScope ts(ctx, "warm_project", 8.0 * ((double)n * (q + 32)), 2.0 * (double)n * q * 16);
    hipLaunchKernelGGL(k_warm_project, dim3(ceil_div(n, rpw)), dim3(256), 0, ctx->stream, a);
    DRE_HIP(hipGetLastError());
}
void warm_z(Ctx* ctx, int n, const Mat& Z1, const double* Cw, const Mat& Res, Mat& Zb, Mat& Zy, Mat& W2) {
    DRE_REQUIRE(n <= 1024 && Zb.rows == n && Zy.rows == n && W2.rows == n && Res.rows == n && Res.cols == n, "warm_z: shapes");
    WarmZArgs a{n, Z1.p, Z1.ld, Cw, Res.p, Res.ld, Zb.p, Zb.ld, Zy.p, Zy.ld, W2.p, W2.ld};
    const int shm = n * 17 * (int)sizeof(double);
    lds_attr(ctx, (const void*)k_warm_z, shm);
    TimedScope ts(ctx, "warm_project", 8.0 * ((double)n * n + 48.0 * n), 2.0 * n * (double)n * 16);
    hipLaunchKernelGGL(k_warm_z, dim3(ceil_div(n, 16)), dim3(256), shm, ctx->stream, a);
    DRE_HIP(hipGetLastError());
}
void warm_small(Ctx* ctx, int q, int m, int kl, int qn, const Mat& Cc, const double* parts, int nparts, double reltol, double abstol, double frac,
                double budget_frac, double* tols, Mat& Uc, Mat& T, Mat& Cp, int* ticket, Mat* Mout, Mat* LTout) {
    DRE_REQUIRE(q >= 1 && q <= 64 && (m == q || m == q + 16) && m <= 80 && kl >= 1 && kl <= qn && qn <= m && Uc.rows >= m && Uc.cols >= qn &&
                T.rows >= kl && T.cols >= kl && Cp.rows >= m && Cp.cols >= 16 && Cc.rows >= m && Cc.cols >= 64 + q, "warm_small: shapes");
    WarmSmallArgs a{q, m, kl, qn, Cc.p, Cc.ld, parts, nparts, reltol, abstol, frac, budget_frac, 24, Mout ? 1 : 0, Mout ? Mout->p : nullptr, Mout ? Mout->ld : 0, LTout ? LTout->p : nullptr, LTout ? LTout->ld : 0, tols, Uc.p, Uc.ld, T.p, T.ld, Cp.p, Cp.ld, ticket};
    TimedScope ts(ctx, "warm_small", 8.0 * (2.0 * m * m + (double)m * qn), 12.0 * m * (double)m * m);
    if (m <= 32) {
        const int shm = 2 * 48 * 49 * (int)sizeof(double);
        lds_attr(ctx, (const void*)k_warm_small<256, 48>, shm);
        hipLaunchKernelGGL((k_warm_small<256, 48>), dim3(1), dim3(256), shm, ctx->stream, a);
    } else if (m <= 48) {
        // 33 .. 48 coordinates (the first warm steps of a run: 32 basis columns, two to three full sweeps): sixteen waves — a Jacobi round's block
        // updates are 1 536 items at 48 active coordinates, six passes of 256 threads (2.8 us per round against 1.2 at 16 coordinates)
        const int shm = 2 * 48 * 49 * (int)sizeof(double);
        lds_attr(ctx, (const void*)k_warm_small<1024, 48>, shm);
        hipLaunchKernelGGL((k_warm_small<1024, 48>), dim3(1), dim3(1024), shm, ctx->stream, a);
    } else {
        const int shm = 2 * 80 * 81 * (int)sizeof(double);
        lds_attr(ctx, (const void*)k_warm_small<1024, 80>, shm);
        hipLaunchKernelGGL((k_warm_small<1024, 80>), dim3(1), dim3(1024), shm, ctx->stream, a);
    }
    DRE_HIP(hipGetLastError());
}
// eigenvectors of the symmetric m x m matrix S (m <= 80) by decreasing |eigenvalue| into U (m x m), one workgroup on ctx's stream: the basis of
// a cold compression is turned into an eigenbasis BESIDE the time loop (gdre.hip), so that the warm-started compression starts from a nearly diagonal matrix
void warm_eig(Ctx* ctx, int m, const Mat& S, Mat& U) {
    DRE_REQUIRE(m >= 1 && m <= 80 && S.rows >= m && S.cols >= m && U.rows >= m && U.cols >= m, "warm_eig: shapes");
    WarmSmallArgs a{m, m, m, m, nullptr, 0, nullptr, 0, 0.0, 0.0, 0.0, 0.0, 8, 2, S.p, S.ld, nullptr, 0, nullptr, U.p, U.ld, nullptr, 0, nullptr, 0, nullptr};
    const int shm = 2 * 80 * 81 * (int)sizeof(double);
    lds_attr(ctx, (const void*)k_warm_small<1024, 80>, shm);
    TimedScope ts(ctx, "warm_eig", 16.0 * m * m, 12.0 * m * (double)m * m);
    hipLaunchKernelGGL((k_warm_small<1024, 80>), dim3(1), dim3(1024), shm, ctx->stream, a);
    DRE_HIP(hipGetLastError());
}
void warm_finish(Ctx* ctx, int n, int q, int kl, const Mat& Q, const Mat& Wk, const Mat& Yp, const Mat& Pp, Mat& R, double* slab, int* ticket, double* tols) {
    DRE_REQUIRE(Q.rows == n && Q.cols >= q && R.rows == n && R.cols >= kl && Yp.rows == n && Yp.cols >= 16 && Pp.rows >= q && Pp.cols >= 16, "warm_finish: shapes");
    WarmFinishArgs a{n, q, kl, Q.p, Q.ld, Wk.p, Wk.ld, Yp.p, Yp.ld, Pp.p, Pp.ld, R.p, R.ld, slab, ticket, tols};
    TimedScope ts(ctx, "warm_finish", 8.0 * ((double)n * q + (double)n * kl + 16.0 * n), 2.0 * n * (double)q * (kl + 16));
    hipLaunchKernelGGL(k_warm_finish, dim3(ceil_div(n, 16)), dim3(256), 0, ctx->stream, a);
    DRE_HIP(hipGetLastError());
}

}  // namespace dre
