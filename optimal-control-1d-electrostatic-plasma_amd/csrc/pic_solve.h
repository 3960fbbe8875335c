// pic_solve.h -- field_solve_kernel: density from an accumulator row (or a given right-hand side), then solve_block
// (pic_device.h): two-scan periodic Poisson solve, energies.  Runs once per step of the streaming
// schedule (the post-step refresh of pic.py:145-146) and for the probes; the force evaluations inside a step are solved in
// the sweep prologues (pic_sweep.h: prologue_field) and everything of a resident step in pic_resident.h, all with
// the same scans.
#pragma once
#include "pic_device.h"

namespace {

constexpr int SBLOCK = BLOCK;        // 8 waves, as the sweeps and the resident kernel: the float64 sums of E^2 and of the
constexpr int SWAVES = SBLOCK / 64;  // kinetic-energy partials are then grouped the same way in all three

struct SolveIO {
  const acc_t* acc;        // [S][env][Ng] deposit (weight sums, 2^-fg units, S sub-rows), or null when rhs is given
  acc_t* acc_clear;        // = acc, or null: the row is zeroed once it has been read (the probes' own row: no memset per probe)
  const double* rhs;       // [env][Ng] right-hand side taken as it is (pic_solve_poisson)
  const double* ke_part;   // [env][nblk] or null
  double* n;               // [env][Ng] density out, or null
  SolveOut out;
};

// Density from the accumulator row (or the given right-hand side) of environment `env`, then solve_block.  Called by a whole
// workgroup of SBLOCK threads; smem: 2 Ng doubles of LDS, ws: 2 SWAVES doubles, slot: 2 doubles.
__device__ __forceinline__ void solve_environment(const SolveIO& io, int env, int Ng, int nblk, int fg, int S, long long sub,
                                                  double scale, double n0, double dx, double N_over_L, unsigned char* smem,
                                                  double* ws, double* slot) {
  double* sb = reinterpret_cast<double*>(smem);       // b, then G_{j+1/2}
  double* se = sb + Ng;                               // phi
  const int tid = threadIdx.x;
  const size_t row = (size_t)env * Ng;
  // density (interpolate.py:16-18), b = n - n0 (pic.py:116)
  if (io.acc) {
    const double unit = ldexp(1.0, -fg);
    for (int j = tid; j < Ng; j += SBLOCK) {
      const double nj = ((double)acc_row_sum(io.acc + row, j, S, sub) * unit) * scale;
      if (io.n) io.n[row + j] = nj;
      sb[j] = nj - n0;
      if (io.acc_clear)
        for (int s = 0; s < S; ++s) io.acc_clear[row + (size_t)s * sub + j] = 0;
    }
  } else {
    for (int j = tid; j < Ng; j += SBLOCK) sb[j] = io.rhs[row + j];
  }
  double k = 0.0;
  if (io.ke_part)
    for (int b = tid; b < nblk; b += SBLOCK) k += io.ke_part[(size_t)env * nblk + b];
  __syncthreads();
  SolveOut o = io.out;
  if (!io.ke_part) o.KE = nullptr;
  solve_block<SWAVES>(o, env, Ng, dx, N_over_L, k, sb, se, ws, slot);
}

__global__ __launch_bounds__(SBLOCK) void field_solve_kernel(SolveIO io, SolveArgs a) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  __shared__ double ws[2 * SWAVES];
  __shared__ double slot[2];
  solve_environment(io, blockIdx.x, a.Ng, a.nblk, a.fg, a.S, a.sub, a.scale, a.n0, a.dx, a.N_over_L, smem_raw, ws, slot);
}

}  // namespace
