// gtr_model.hpp -- per-locus GTR eigen-system, computed on the device (one thread per locus).
//
// Model of tapir/data/models_and_rates.bf:978-1001: Q_ij = r_ij * pi_j (i != j), rows sum to zero, NOT
// normalised to one expected substitution; HyPhy's BranchLength() therefore reports expected
// substitutions kappa * t * s with kappa = sum_i pi_i * (-Q_ii) = 2 sum_{i<j} pi_i pi_j r_ij, which is
// why the rate tapir reads from the JSON is kappa * siteRate (bf:1056-1061).
//
// exp(Q t) = U diag(exp(lam t)) U^-1 via the symmetrised matrix S = D^1/2 Q D^-1/2 (D = diag(pi)) and a
// cyclic Jacobi iteration.  Eigenvalue 0 is moved to slot 0 and made exact: lam[0] = 0, U[:,0] = 1,
// U^-1[0,:] = pi.  The kernels rely on that structure (one exp less per branch, no derivative terms).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace tphip {

// Packed to the 32 doubles the kernels actually read (eigenvalue 0 and its trivial eigenvectors are implicit),
// so a wave can hold the whole model in 64 SGPRs instead of re-fetching it in every tree op.
struct LocusModel {
    double lam[3];   // lam_1..lam_3 (all < 0); lam_0 == 0 is implicit
    double U[12];    // U[i*3 + (k-1)], k = 1..3;      U[i][0] == 1 is implicit
    double Ui[12];   // Ui[(k-1)*4 + j], k = 1..3;     U^-1[0][j] == pi[j]
    double pi[4];
    double kappa;
};
static_assert(sizeof(LocusModel) == 256, "LocusModel layout");

}  // namespace tphip
