// Small dense host linear algebra used by the Projection shift strategy, whose Ritz values the
// reference also computes on the CPU (/root/reference/src/shifts/projection.jl:63-67, Stuff.jl:13-19).
// Column-major storage, plain C++ (no LAPACK in this image).
#pragma once
#include <algorithm>
#include <cmath>
#include <complex>
#include <stdexcept>
#include <vector>

namespace dre {

// Left singular vectors and singular values of the p x w matrix R (column-major, ld = p) by one-sided
// Jacobi on R' (rows of R are orthogonalised).  On return U is p x p (column-major), sv has p entries
// (not sorted); if w < p the trailing directions get singular value 0.
inline void host_svd_left(int p, int w, const std::vector<double>& R, std::vector<double>& U, std::vector<double>& sv) {
    // B = R' is w x p (columns of B = rows of R); rotate columns of B, accumulate J (p x p) -> U = J
    std::vector<double> B((size_t)w * p);
    for (int i = 0; i < p; ++i)
        for (int j = 0; j < w; ++j) B[j + (size_t)i * w] = R[i + (size_t)j * p];
    U.assign((size_t)p * p, 0.0);
    for (int i = 0; i < p; ++i) U[i + (size_t)i * p] = 1.0;
    const double eps = 2.220446049250313e-16;
    for (int sweep = 0; sweep < 60; ++sweep) {
        bool rotated = false;
        for (int a = 0; a < p - 1; ++a)
            for (int b = a + 1; b < p; ++b) {
                double* ca = &B[(size_t)a * w];
                double* cb = &B[(size_t)b * w];
                double aa = 0, bb = 0, ab = 0;
                for (int i = 0; i < w; ++i) { aa += ca[i] * ca[i]; bb += cb[i] * cb[i]; ab += ca[i] * cb[i]; }
                if (std::fabs(ab) <= eps * std::sqrt(aa * bb) || ab == 0.0) continue;
                rotated = true;
                const double zeta = (bb - aa) / (2.0 * ab);
                const double t = (zeta >= 0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
                const double c = 1.0 / std::sqrt(1.0 + t * t), s = c * t;
                for (int i = 0; i < w; ++i) { double x = ca[i], y = cb[i]; ca[i] = c * x - s * y; cb[i] = s * x + c * y; }
                double* ua = &U[(size_t)a * p];
                double* ub = &U[(size_t)b * p];
                for (int i = 0; i < p; ++i) { double x = ua[i], y = ub[i]; ua[i] = c * x - s * y; ub[i] = s * x + c * y; }
            }
        if (!rotated) break;
    }
    sv.assign(p, 0.0);
    for (int a = 0; a < p; ++a) {
        double s = 0;
        for (int i = 0; i < w; ++i) s += B[i + (size_t)a * w] * B[i + (size_t)a * w];
        sv[a] = std::sqrt(s);
    }
}

// The same for a matrix whose singular values fall off steeply (the history block of a converging ADI solve: sigma from 1e-3 to below
// 1e-13 within a few directions): Householder QR with column pivoting, stopped as soon as everything not yet eliminated is below
// 0.1 * tol in the Frobenius norm (those directions cannot hold a singular value above tol), then the Jacobi SVD of the rp leading ROWS only.
// One-sided Jacobi on the full 224 x 224 block took 55 ms per call; this takes well under a millisecond when rp is a handful.
// On return U is p x rp (column-major, ld p), sv has rp entries.
inline void host_rrqr_svd_left(int p, int w, const std::vector<double>& R, double tol, std::vector<double>& U, std::vector<double>& sv) {
    std::vector<double> A(R);                              // p x w, ld p
    std::vector<double> cn((size_t)w, 0.0);
    std::vector<std::vector<double>> vs;                   // Householder vectors (length p, zero above their pivot row), unit scaling in betas
    std::vector<double> betas;
    const int kmax = p < w ? p : w;
    int rp = 0;
    for (int step = 0; step < kmax; ++step) {
        double tot = 0.0; int piv = step; double best = -1.0;
        for (int j = step; j < w; ++j) {
            double sacc = 0.0;
            for (int i = step; i < p; ++i) sacc += A[i + (size_t)j * p] * A[i + (size_t)j * p];
            cn[(size_t)j] = sacc; tot += sacc;
            if (sacc > best) { best = sacc; piv = j; }
        }
        if (tot <= (0.1 * tol) * (0.1 * tol)) break;
        if (piv != step) for (int i = 0; i < p; ++i) { const double t = A[i + (size_t)step * p]; A[i + (size_t)step * p] = A[i + (size_t)piv * p]; A[i + (size_t)piv * p] = t; }
        // Householder reflector for A[step:, step]
        double alpha = A[step + (size_t)step * p], sig = 0.0;
        for (int i = step + 1; i < p; ++i) sig += A[i + (size_t)step * p] * A[i + (size_t)step * p];
        std::vector<double> v((size_t)p, 0.0);
        double beta = 0.0;
        if (sig > 0.0 || alpha < 0.0) {
            const double nrm = std::sqrt(alpha * alpha + sig);
            const double b = alpha >= 0.0 ? -nrm : nrm;
            v[(size_t)step] = alpha - b;
            for (int i = step + 1; i < p; ++i) v[(size_t)i] = A[i + (size_t)step * p];
            double vv = 0.0;
            for (int i = step; i < p; ++i) vv += v[(size_t)i] * v[(size_t)i];
            beta = vv > 0.0 ? 2.0 / vv : 0.0;
            for (int j = step; j < w; ++j) {
                double d = 0.0;
                for (int i = step; i < p; ++i) d += v[(size_t)i] * A[i + (size_t)j * p];
                d *= beta;
                for (int i = step; i < p; ++i) A[i + (size_t)j * p] -= d * v[(size_t)i];
            }
        }
        vs.push_back(std::move(v)); betas.push_back(beta);
        rp = step + 1;
    }
    if (rp == 0) { U.clear(); sv.clear(); return; }
    std::vector<double> B((size_t)rp * w);
    for (int j = 0; j < w; ++j)
        for (int i = 0; i < rp; ++i) B[i + (size_t)j * rp] = (i <= j || true) ? A[i + (size_t)j * p] : 0.0;
    std::vector<double> Us;
    host_svd_left(rp, w, B, Us, sv);                     // rp x rp
    U.assign((size_t)p * rp, 0.0);
    for (int c = 0; c < rp; ++c) for (int i = 0; i < rp; ++i) U[i + (size_t)c * p] = Us[i + (size_t)c * rp];
    for (int step = rp - 1; step >= 0; --step) {          // U <- H_step U
        const std::vector<double>& v = vs[(size_t)step];
        const double beta = betas[(size_t)step];
        if (beta == 0.0) continue;
        for (int c = 0; c < rp; ++c) {
            double d = 0.0;
            for (int i = step; i < p; ++i) d += v[(size_t)i] * U[i + (size_t)c * p];
            d *= beta;
            for (int i = step; i < p; ++i) U[i + (size_t)c * p] -= d * v[(size_t)i];
        }
    }
}

// Solve E X = A for X (n x n) by LU with partial pivoting; A is overwritten by X.
inline void host_lu_solve(int n, std::vector<double> E, std::vector<double>& A) {
    std::vector<int> piv(n);
    for (int k = 0; k < n; ++k) {
        int p = k; double best = std::fabs(E[k + (size_t)k * n]);
        for (int i = k + 1; i < n; ++i) if (std::fabs(E[i + (size_t)k * n]) > best) { best = std::fabs(E[i + (size_t)k * n]); p = i; }
        if (best == 0.0) throw std::runtime_error("host_lu_solve: singular matrix");
        piv[k] = p;
        if (p != k) {
            for (int j = 0; j < n; ++j) std::swap(E[k + (size_t)j * n], E[p + (size_t)j * n]);
            for (int j = 0; j < n; ++j) std::swap(A[k + (size_t)j * n], A[p + (size_t)j * n]);
        }
        const double rp = 1.0 / E[k + (size_t)k * n];
        for (int i = k + 1; i < n; ++i) E[i + (size_t)k * n] *= rp;
        for (int j = k + 1; j < n; ++j) {
            const double ekj = E[k + (size_t)j * n];
            if (ekj != 0.0) for (int i = k + 1; i < n; ++i) E[i + (size_t)j * n] -= E[i + (size_t)k * n] * ekj;
        }
        for (int j = 0; j < n; ++j) {
            const double akj = A[k + (size_t)j * n];
            if (akj != 0.0) for (int i = k + 1; i < n; ++i) A[i + (size_t)j * n] -= E[i + (size_t)k * n] * akj;
        }
    }
    for (int j = 0; j < n; ++j)
        for (int k = n - 1; k >= 0; --k) {
            double x = A[k + (size_t)j * n] / E[k + (size_t)k * n];
            A[k + (size_t)j * n] = x;
            if (x != 0.0) for (int i = 0; i < k; ++i) A[i + (size_t)j * n] -= E[i + (size_t)k * n] * x;
        }
}

// Eigenvalues of a real general matrix (column-major n x n, destroyed): elimination to Hessenberg form
// followed by the Francis double-shift QR iteration (the classical EISPACK elmhes/hqr pair).
inline std::vector<std::complex<double>> host_eigvals(int n, std::vector<double>& M) {
    std::vector<std::complex<double>> out(n);
    if (n == 0) return out;
    // (the row operations below walk a row with the stride of a column: with a leading dimension that is a multiple of 32 doubles every entry of
    // a row maps to the same few cache sets — the 256 x 256 projected pencil of a Projection batch took 47 ms against 10 ms for 224 x 224 —
    // so such matrices are iterated on in a copy with an odd leading dimension)
    const size_t ld = (n % 32 == 0 && n >= 64) ? (size_t)n + 3 : (size_t)n;
    std::vector<double> padded;
    if (ld != (size_t)n) {
        padded.assign(ld * n, 0.0);
        for (int j = 0; j < n; ++j) for (int i = 0; i < n; ++i) padded[i + (size_t)j * ld] = M[i + (size_t)j * n];
    }
    double* const Mp = ld != (size_t)n ? padded.data() : M.data();
    auto a = [&](int i, int j) -> double& { return Mp[(i - 1) + (size_t)(j - 1) * ld]; };   // 1-based accessor
    const double eps = 2.220446049250313e-16;
    // --- reduction to upper Hessenberg form
    for (int m = 2; m < n; ++m) {
        double x = 0.0; int i = m;
        for (int j = m; j <= n; ++j) if (std::fabs(a(j, m - 1)) > std::fabs(x)) { x = a(j, m - 1); i = j; }
        if (i != m) {
            for (int j = m - 1; j <= n; ++j) std::swap(a(i, j), a(m, j));
            for (int j = 1; j <= n; ++j) std::swap(a(j, i), a(j, m));
        }
        if (x != 0.0) {
            for (i = m + 1; i <= n; ++i) {
                double y = a(i, m - 1);
                if (y != 0.0) {
                    y /= x; a(i, m - 1) = y;
                    for (int j = m; j <= n; ++j) a(i, j) -= y * a(m, j);
                    for (int j = 1; j <= n; ++j) a(j, m) += y * a(j, i);
                }
            }
        }
    }
    for (int i = 3; i <= n; ++i) for (int j = 1; j <= i - 2; ++j) a(i, j) = 0.0;
    // --- QR iteration
    std::vector<double> wr(n + 1, 0.0), wi(n + 1, 0.0);
    double anorm = 0.0;
    for (int i = 1; i <= n; ++i) for (int j = std::max(i - 1, 1); j <= n; ++j) anorm += std::fabs(a(i, j));
    int nn = n; double t = 0.0;
    double p = 0, q = 0, r = 0, s, x, y, z, w, u, v;
    while (nn >= 1) {
        int its = 0, l;
        do {
            for (l = nn; l >= 2; --l) {
                s = std::fabs(a(l - 1, l - 1)) + std::fabs(a(l, l));
                if (s == 0.0) s = anorm;
                if (std::fabs(a(l, l - 1)) <= eps * s) { a(l, l - 1) = 0.0; break; }
            }
            x = a(nn, nn);
            if (l == nn) { wr[nn] = x + t; wi[nn--] = 0.0; }
            else {
                y = a(nn - 1, nn - 1);
                w = a(nn, nn - 1) * a(nn - 1, nn);
                if (l == nn - 1) {
                    p = 0.5 * (y - x); q = p * p + w; z = std::sqrt(std::fabs(q)); x += t;
                    if (q >= 0.0) {
                        z = p + (p >= 0.0 ? std::fabs(z) : -std::fabs(z));
                        wr[nn - 1] = wr[nn] = x + z;
                        if (z != 0.0) wr[nn] = x - w / z;
                        wi[nn - 1] = wi[nn] = 0.0;
                    } else {
                        wr[nn - 1] = wr[nn] = x + p;
                        wi[nn] = z; wi[nn - 1] = -z;
                    }
                    nn -= 2;
                } else {
                    if (its == 120) throw std::runtime_error("host_eigvals: too many QR iterations");
                    if (its > 0 && its % 10 == 0) {
                        t += x;
                        for (int i = 1; i <= nn; ++i) a(i, i) -= x;
                        s = std::fabs(a(nn, nn - 1)) + std::fabs(a(nn - 1, nn - 2));
                        y = x = 0.75 * s; w = -0.4375 * s * s;
                    }
                    ++its;
                    int m;
                    for (m = nn - 2; m >= l; --m) {
                        z = a(m, m); r = x - z; s = y - z;
                        p = (r * s - w) / a(m + 1, m) + a(m, m + 1);
                        q = a(m + 1, m + 1) - z - r - s;
                        r = a(m + 2, m + 1);
                        s = std::fabs(p) + std::fabs(q) + std::fabs(r);
                        p /= s; q /= s; r /= s;
                        if (m == l) break;
                        u = std::fabs(a(m, m - 1)) * (std::fabs(q) + std::fabs(r));
                        v = std::fabs(p) * (std::fabs(a(m - 1, m - 1)) + std::fabs(z) + std::fabs(a(m + 1, m + 1)));
                        if (u <= eps * v) break;
                    }
                    for (int i = m + 2; i <= nn; ++i) { a(i, i - 2) = 0.0; if (i != m + 2) a(i, i - 3) = 0.0; }
                    for (int k = m; k <= nn - 1; ++k) {
                        if (k != m) {
                            p = a(k, k - 1); q = a(k + 1, k - 1); r = 0.0;
                            if (k != nn - 1) r = a(k + 2, k - 1);
                            if ((x = std::fabs(p) + std::fabs(q) + std::fabs(r)) != 0.0) { p /= x; q /= x; r /= x; }
                        }
                        const double nrm = std::sqrt(p * p + q * q + r * r);
                        s = p >= 0.0 ? nrm : -nrm;
                        if (s != 0.0) {
                            if (k == m) { if (l != m) a(k, k - 1) = -a(k, k - 1); }
                            else a(k, k - 1) = -s * x;
                            p += s; x = p / s; y = q / s; z = r / s; q /= p; r /= p;
                            for (int j = k; j <= nn; ++j) {
                                p = a(k, j) + q * a(k + 1, j);
                                if (k != nn - 1) { p += r * a(k + 2, j); a(k + 2, j) -= p * z; }
                                a(k + 1, j) -= p * y; a(k, j) -= p * x;
                            }
                            const int mmin = nn < k + 3 ? nn : k + 3;
                            for (int i = l; i <= mmin; ++i) {
                                p = x * a(i, k) + y * a(i, k + 1);
                                if (k != nn - 1) { p += z * a(i, k + 2); a(i, k + 2) -= p * r; }
                                a(i, k + 1) -= p * q; a(i, k) -= p;
                            }
                        }
                    }
                }
            }
        } while (l < nn - 1);
    }
    for (int i = 1; i <= n; ++i) out[i - 1] = std::complex<double>(wr[i], wi[i]);
    return out;
}

// Generalised eigenvalues of (A, E) with nonsingular E: eigenvalues of E^-1 A.
inline std::vector<std::complex<double>> host_gen_eigvals(int n, const std::vector<double>& A, const std::vector<double>& E) {
    std::vector<double> M = A;
    host_lu_solve(n, E, M);
    return host_eigvals(n, M);
}

}  // namespace dre
