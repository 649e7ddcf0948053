// Minimum-norm least squares on centred normal equations (device code shared by the analog regression epilogue and
// the batched linear regression).
#pragma once
#include <hip/hip_runtime.h>

namespace sdlsq {

constexpr int kMaxF = 8;

// A = [ S | b ]: S the symmetric F x F matrix of centred cross products, b = column F.  coef = pinv(S) b like the
// lstsq inside sklearn's LinearRegression: eigen-decomposition S = V diag(lam) V^T by cyclic Jacobi rotations, coef =
// sum over the non-null directions of v (v . b) / lam.  Under-determined and collinear designs then give the
// pseudo-inverse solution.  A direction is null when lam <= 1e-12 * lam_max (singular value below 1e-6 of the largest:
// the normal equations cannot resolve more).  S is overwritten.
__device__ inline void minnorm_solve(int F, double (&A)[kMaxF][kMaxF + 1], double* coef) {
    double V[kMaxF][kMaxF], bvec[kMaxF];
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
}

}  // namespace sdlsq
