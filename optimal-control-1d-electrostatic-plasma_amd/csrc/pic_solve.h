// pic_solve.h -- field_solve_kernel: slab reduction, density, two-scan periodic Poisson solve, energies.
#pragma once
#include "pic_device.h"

namespace {

// ---------------------------------------------------------------------------------------------
// Field solve (one workgroup per environment).
// Replaces Gaussian_Elimination_Periodic + dense grad matvec (src/env/solve.py:27-53,
// src/env/util.py:99-103, pic.py:116-117).  With G_{j+1/2} = (phi_{j+1}-phi_j)/dx the 3-point
// periodic Poisson equation reads G_{j+1/2} - G_{j-1/2} = b_j dx, so G = cumsum(b) dx - mean and
// E_j = -(phi_{j+1}-phi_{j-1})/(2dx) = -(G_{j+1/2} + G_{j-1/2})/2.  phi follows from a second
// scan and is returned with zero mean.
// ---------------------------------------------------------------------------------------------
constexpr int SBLOCK = 1024;         // field-solve workgroup: 16 waves
constexpr int SWAVES = SBLOCK / 64;
constexpr int SGROUPS = 4;           // slab rows are summed by 4 groups of 256 lanes

// inputs / outputs of one field solve; a launch carries up to two independent ones (blockIdx.y), e.g. the
// post-step refresh of step s and the first force evaluation of step s+1
struct SolveIO {
  const double* part;      // slab [env][nblk][Ng] to reduce
  const double* ext;       // E_ext [env][Ng] or null
  const double* ke_part;   // [env][nblk] or null
  double *n, *Ef, *E, *phi, *KE, *PE, *PEr;   // any may be null
};

__global__ __launch_bounds__(SBLOCK) void field_solve_kernel(SolveIO io0, SolveIO io1, SolveArgs a) {
  const SolveIO io = blockIdx.y == 0 ? io0 : io1;
  const double* __restrict__ part = io.part;
  const double* __restrict__ E_ext = io.ext;
  const double* __restrict__ ke_part = io.ke_part;
  double* __restrict__ n_out = io.n;
  double* __restrict__ Ef_out = io.Ef;
  double* __restrict__ E_out = io.E;
  double* __restrict__ phi_out = io.phi;
  double* __restrict__ KE_out = io.KE;
  double* __restrict__ PE_out = io.PE;
  double* __restrict__ PEr_out = io.PEr;
  extern __shared__ __align__(16) unsigned char smem_raw[];
  double* sb = reinterpret_cast<double*>(smem_raw);   // b, then G_{j+1/2}
  double* se = sb + a.Ng;                             // E, then phi
  double* sp = se + a.Ng;                             // [SGROUPS][Ng] partial row sums
  __shared__ double ws[SWAVES];

  const int tid = threadIdx.x;
  const int env = a.env0 + blockIdx.x;
  const int Ng = a.Ng;
  const int m = (Ng + SBLOCK - 1) / SBLOCK;
  const int lo = min(tid * m, Ng), hi = min(lo + m, Ng);

  // density: slab rows summed in a fixed order (group g takes rows g, g+4, ...; 4 loads in flight per
  // lane), scaled (interpolate.py:16-18), b = n - n0 (pic.py:116)
  const double* slab = part + (size_t)env * a.nblk * Ng;
  const int g = tid >> 8, lane = tid & 255;
  for (int j = lane; j < Ng; j += 256) {
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0, s4 = 0.0, s5 = 0.0, s6 = 0.0, s7 = 0.0;
    int b = g;
    for (; b + 7 * SGROUPS < a.nblk; b += 8 * SGROUPS) {       // 8 independent loads in flight per lane
      s0 += slab[(size_t)b * Ng + j];
      s1 += slab[(size_t)(b + SGROUPS) * Ng + j];
      s2 += slab[(size_t)(b + 2 * SGROUPS) * Ng + j];
      s3 += slab[(size_t)(b + 3 * SGROUPS) * Ng + j];
      s4 += slab[(size_t)(b + 4 * SGROUPS) * Ng + j];
      s5 += slab[(size_t)(b + 5 * SGROUPS) * Ng + j];
      s6 += slab[(size_t)(b + 6 * SGROUPS) * Ng + j];
      s7 += slab[(size_t)(b + 7 * SGROUPS) * Ng + j];
    }
    for (; b < a.nblk; b += SGROUPS) s0 += slab[(size_t)b * Ng + j];
    sp[g * Ng + j] = ((s0 + s1) + (s2 + s3)) + ((s4 + s5) + (s6 + s7));
  }
  __syncthreads();
  for (int j = tid; j < Ng; j += SBLOCK) {
    double s = (sp[j] + sp[Ng + j]) + (sp[2 * Ng + j] + sp[3 * Ng + j]);
    double nj = s * a.scale;
    if (n_out) n_out[(size_t)env * Ng + j] = nj;
    sb[j] = nj - a.n0;
  }
  __syncthreads();

  // G_{j+1/2} = dx * inclusive_scan(b)
  double loc = 0.0;
  for (int j = lo; j < hi; ++j) loc += sb[j];
  double tot;
  double run = block_excl_scan<SWAVES>(loc, ws, tot);
  loc = 0.0;
  for (int j = lo; j < hi; ++j) {
    run += sb[j];
    double gj = run * a.dx;
    sb[j] = gj;
    loc += gj;
  }
  const double gmean = block_sum<SWAVES>(loc, ws) / (double)Ng;   // syncs: all of sb is G now

  // E_j = -(G_{j+1/2} + G_{j-1/2}) / 2, plus the external field for force evaluations (util.py:102-103)
  double e2 = 0.0;
  for (int j = tid; j < Ng; j += SBLOCK) {
    double gp = sb[j] - gmean;
    double gm = sb[j == 0 ? Ng - 1 : j - 1] - gmean;
    double E = -0.5 * (gp + gm);
    se[j] = E;
    double Et = E_ext ? E + E_ext[(size_t)env * Ng + j] : E;
    if (Ef_out) Ef_out[(size_t)env * Ng + j] = Et;
    if (E_out) E_out[(size_t)env * Ng + j] = Et;
    e2 += Et * Et;
  }
  const double S = block_sum<SWAVES>(e2, ws);
  if (tid == 0) {
    double pe = 0.5 * S * a.dx;                       // objective.py:33 / util.py:129
    if (PEr_out) PEr_out[env] = pe;
    if (PE_out) PE_out[env] = pe * a.N_over_L;        // util.py:130
  }

  if (KE_out) {
    double k = 0.0;
    for (int b = tid; b < a.nblk; b += SBLOCK) k += ke_part[(size_t)env * a.nblk + b];
    k = block_sum<SWAVES>(k, ws);
    if (tid == 0) KE_out[env] = 0.5 * k;              // util.py:144
  }

  if (phi_out) {
    // phi_{j+1} = phi_j + dx G_{j+1/2}: exclusive scan, then remove the mean
    loc = 0.0;
    for (int j = lo; j < hi; ++j) loc += (sb[j] - gmean) * a.dx;
    run = block_excl_scan<SWAVES>(loc, ws, tot);
    double ploc = 0.0;
    for (int j = lo; j < hi; ++j) {
      se[j] = run;
      ploc += run;
      run += (sb[j] - gmean) * a.dx;
    }
    const double pmean = block_sum<SWAVES>(ploc, ws) / (double)Ng;
    for (int j = tid; j < Ng; j += SBLOCK) phi_out[(size_t)env * Ng + j] = se[j] - pmean;
  }
}

}  // namespace
