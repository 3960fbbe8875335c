// pic_solve.h -- field_solve_kernel: density from an accumulator row, two-scan periodic Poisson solve, energies
// (DESIGN.md 4.2).  Runs once per step (the post-step refresh of pic.py:145-146) and for the probes; the force
// evaluations inside a step are solved in the sweep prologues (pic_sweep.h: prologue_field) with the same scans.
#pragma once
#include "pic_device.h"

namespace {

constexpr int SBLOCK = BLOCK;        // field-solve workgroup: the sweeps' size, so that every solve of the library (sweep prologue,
                                     // resident kernel, this one) splits its scans the same way and rounds the same way
constexpr int SWAVES = SBLOCK / 64;

struct SolveIO {
  const acc_t* acc;        // [env][Ng] deposit (weight sums, 2^-fg units), or null when rhs is given
  const double* rhs;       // [env][Ng] right-hand side taken as it is (pic_solve_poisson)
  const double* ext;       // E_ext [env][Ng] or null
  const double* ke_part;   // [env][nblk] or null
  double *n, *E, *phi, *KE, *PE, *PEr;   // any may be null
};

__global__ __launch_bounds__(SBLOCK) void field_solve_kernel(SolveIO io, SolveArgs a) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  double* sb = reinterpret_cast<double*>(smem_raw);   // b, then G_{j+1/2}
  double* se = sb + a.Ng;                             // phi
  __shared__ double ws[SWAVES];

  const int tid = threadIdx.x;
  const int env = blockIdx.x;
  const int Ng = a.Ng;
  const size_t row = (size_t)env * Ng;

  // density (interpolate.py:16-18), b = n - n0 (pic.py:116)
  if (io.acc) {
    const double unit = ldexp(1.0, -a.fg);
    for (int j = tid; j < Ng; j += SBLOCK) {
      const double nj = ((double)io.acc[row + j] * unit) * a.scale;
      if (io.n) io.n[row + j] = nj;
      sb[j] = nj - a.n0;
    }
  } else {
    for (int j = tid; j < Ng; j += SBLOCK) sb[j] = io.rhs[row + j];
  }
  __syncthreads();
  const double gmean = scan_gradient<SWAVES>(sb, Ng, a.dx, ws);

  // E_j = -(G_{j+1/2} + G_{j-1/2}) / 2, plus the external field where the caller evaluates a force (util.py:102-103)
  double e2 = 0.0;
  for (int j = tid; j < Ng; j += SBLOCK) {
    const double gp = sb[j] - gmean;
    const double gm = sb[j == 0 ? Ng - 1 : j - 1] - gmean;
    const double E = -0.5 * (gp + gm);
    const double Et = io.ext ? E + io.ext[row + j] : E;
    if (io.E) io.E[row + j] = Et;
    e2 += Et * Et;
  }
  const double S = block_sum<SWAVES>(e2, ws);
  if (tid == 0) {
    const double pe = 0.5 * S * a.dx;                 // objective.py:33 / util.py:129
    if (io.PEr) io.PEr[env] = pe;
    if (io.PE) io.PE[env] = pe * a.N_over_L;          // util.py:130
  }

  if (io.KE) {
    double k = 0.0;
    for (int b = tid; b < a.nblk; b += SBLOCK) k += io.ke_part[(size_t)env * a.nblk + b];
    k = block_sum<SWAVES>(k, ws);
    if (tid == 0) io.KE[env] = 0.5 * k;               // util.py:144
  }

  if (io.phi) {
    // phi_{j+1} = phi_j + dx G_{j+1/2}: exclusive scan, then remove the mean
    const int m = (Ng + SBLOCK - 1) / SBLOCK;
    const int lo = min(tid * m, Ng), hi = min(lo + m, Ng);
    double loc = 0.0, tot;
    for (int j = lo; j < hi; ++j) loc += (sb[j] - gmean) * a.dx;
    double run = block_excl_scan<SWAVES>(loc, ws, tot);
    double ploc = 0.0;
    for (int j = lo; j < hi; ++j) {
      se[j] = run;
      ploc += run;
      run += (sb[j] - gmean) * a.dx;
    }
    const double pmean = block_sum<SWAVES>(ploc, ws) / (double)Ng;
    for (int j = tid; j < Ng; j += SBLOCK) io.phi[row + j] = se[j] - pmean;
  }
}

}  // namespace
