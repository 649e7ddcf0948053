// Minimum-norm least squares on centred normal equations (device code shared by the analog regression epilogue and
// the batched linear regression).
#pragma once
#include <hip/hip_runtime.h>

namespace sdlsq {

constexpr int kMaxF = 8;

// A = [ S | b ]: S the symmetric F x F matrix of centred cross products, b = column F.  coef = pinv(S) b like the
// lstsq inside sklearn's LinearRegression: eigen-decomposition S = V diag(lam) V^T by cyclic Jacobi rotations, coef =
// sum over the non-null directions of v (v . b) / lam.  Under-determined and collinear designs then give the
// pseudo-inverse solution.  The system is equilibrated first (S_fg / sqrt(S_ff S_gg): the correlation matrix), so
// features in very different units (precipitation ~1e-5 next to pressure ~1e5) are resolved alike -- lstsq works on the
// data matrix, where a small-scale column is not lost either; a direction is null when its eigenvalue of the
// equilibrated matrix is <= 1e-12 of the largest (correlation beyond 1 - 1e-12: the normal equations cannot resolve
// more), a constant feature gets coefficient 0; among the solutions of a rank-deficient system the one of minimum norm in
// the original coordinates is returned, like lstsq does.  S is overwritten.
__device__ inline void minnorm_solve(int F, double (&A)[kMaxF][kMaxF + 1], double* coef) {
    double V[kMaxF][kMaxF], bvec[kMaxF], sc[kMaxF];
    for (int f = 0; f < F; ++f) sc[f] = A[f][f] > 0.0 ? 1.0 / sqrt(A[f][f]) : 0.0;
    for (int f = 0; f < F; ++f) {
        for (int g = 0; g < F; ++g) A[f][g] = f == g ? (sc[f] > 0.0 ? 1.0 : 0.0) : A[f][g] * sc[f] * sc[g];
        A[f][F] *= sc[f];
    }
    for (int f = 0; f < F; ++f) {
        bvec[f] = A[f][F];
        for (int g = 0; g < F; ++g) V[f][g] = f == g ? 1.0 : 0.0;
    }
    for (int sweep = 0; sweep < 12; ++sweep) {
        double off = 0.0, diag = 0.0;
        for (int f = 0; f < F; ++f) {
            diag += A[f][f] * A[f][f];
            for (int g = f + 1; g < F; ++g) off += A[f][g] * A[f][g];
        }
        if (off <= 1e-30 * diag || off == 0.0) break;
        for (int p = 0; p < F - 1; ++p)
            for (int r = p + 1; r < F; ++r) {
                const double apr = A[p][r];
                if (apr == 0.0) continue;
                const double theta = (A[r][r] - A[p][p]) / (2.0 * apr);
                const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double cs = 1.0 / sqrt(t * t + 1.0), sn = t * cs;
                for (int g = 0; g < F; ++g) {  // S <- J^T S J on rows / columns p, r
                    const double agp = A[g][p], agr = A[g][r];
                    A[g][p] = cs * agp - sn * agr;
                    A[g][r] = sn * agp + cs * agr;
                }
                for (int g = 0; g < F; ++g) {
                    const double apg = A[p][g], arg = A[r][g];
                    A[p][g] = cs * apg - sn * arg;
                    A[r][g] = sn * apg + cs * arg;
                }
                for (int g = 0; g < F; ++g) {
                    const double vgp = V[g][p], vgr = V[g][r];
                    V[g][p] = cs * vgp - sn * vgr;
                    V[g][r] = sn * vgp + cs * vgr;
                }
            }
    }
    double lam_max = 0.0;
    for (int f = 0; f < F; ++f) lam_max = fmax(lam_max, A[f][f]);
    for (int f = 0; f < F; ++f) coef[f] = 0.0;
    for (int e = 0; e < F; ++e) {
        const double lam = A[e][e];
        if (!(lam > 1e-12 * lam_max)) continue;
        double vb = 0.0;
        for (int f = 0; f < F; ++f) vb += V[f][e] * bvec[f];
        const double w = vb / lam;
        for (int f = 0; f < F; ++f) coef[f] += V[f][e] * w;
    }
    for (int f = 0; f < F; ++f) coef[f] *= sc[f];
    // coef solves the original system; lstsq returns its minimum-norm solution *in the original coordinates*: remove the
    // components along the null space of S, spanned by sc * v over the null directions v found above (Gram-Schmidt)
    double nb[kMaxF][kMaxF];
    int nn = 0;
    for (int e = 0; e < F; ++e) {
        if (A[e][e] > 1e-12 * lam_max) continue;
        double nrm = 0.0;
        for (int f = 0; f < F; ++f) {
            nb[nn][f] = sc[f] * V[f][e];
            nrm += nb[nn][f] * nb[nn][f];
        }
        for (int p = 0; p < nn; ++p) {
            double dot = 0.0;
            for (int f = 0; f < F; ++f) dot += nb[nn][f] * nb[p][f];
            for (int f = 0; f < F; ++f) nb[nn][f] -= dot * nb[p][f];
        }
        double n2 = 0.0;
        for (int f = 0; f < F; ++f) n2 += nb[nn][f] * nb[nn][f];
        if (!(n2 > 1e-24 * nrm) || nrm == 0.0) continue;  // a constant feature (sc = 0) or a dependent direction
        const double inv = 1.0 / sqrt(n2);
        for (int f = 0; f < F; ++f) nb[nn][f] *= inv;
        ++nn;
    }
    for (int p = 0; p < nn; ++p) {
        double dot = 0.0;
        for (int f = 0; f < F; ++f) dot += coef[f] * nb[p][f];
        for (int f = 0; f < F; ++f) coef[f] -= dot * nb[p][f];
    }
}

// Cholesky solve of the symmetric positive definite n x n system H d = r (n <= kMaxF + 1); false if a pivot is not positive
__device__ inline bool chol_solve(int n, double (&H)[kMaxF + 1][kMaxF + 1], const double* r, double* d) {
    for (int j = 0; j < n; ++j) {
        double s = H[j][j];
        for (int p = 0; p < j; ++p) s -= H[j][p] * H[j][p];
        if (!(s > 0.0)) return false;
        const double l = sqrt(s);
        H[j][j] = l;
        for (int i = j + 1; i < n; ++i) {
            double t = H[i][j];
            for (int p = 0; p < j; ++p) t -= H[i][p] * H[j][p];
            H[i][j] = t / l;
        }
    }
    for (int i = 0; i < n; ++i) {
        double t = r[i];
        for (int p = 0; p < i; ++p) t -= H[i][p] * d[p];
        d[i] = t / H[i][i];
    }
    for (int i = n - 1; i >= 0; --i) {
        double t = d[i];
        for (int p = i + 1; p < n; ++p) t -= H[p][i] * d[p];
        d[i] = t / H[i][i];
    }
    return true;
}

// log(1 + exp(z)) and the logistic function without overflow
__device__ inline double softplus(double z) { return z > 0.0 ? z + log1p(exp(-z)) : log1p(exp(z)); }
__device__ inline double sigmoid(double z) { return 0.5 * (1.0 + tanh(0.5 * z)); }

}  // namespace sdlsq
