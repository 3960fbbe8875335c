// pic_resident.h -- resident_kernel<P, A, SHAPE, PPT, NW>: whole environment steps inside ONE workgroup.
//
// The reference's own workload is many small environments (N = 5000, Ng = 250: run_wo_oc.py:33-34, the RL
// trainers).  There a step of the streaming schedule is four dependent launches of a few microseconds each and
// most of that is latency (accumulator round trips, kernel boundaries).  When an environment's particles fit
// the registers of one workgroup (N <= 64 NW PPT, up to 8192), this kernel keeps them there for all `nsteps` steps
// of a pic_step call: the meshes live in LDS, the field solves are workgroup scans, nothing crosses a kernel
// boundary and nothing but the final particles, fields and per-step energies goes to HBM.
//
// Arithmetic is the streaming path's, helper by helper (locate, gather_field, drift, deposit, mesh_node_sum,
// scan_fields; where registers allow, the gather re-uses the cell and weights the previous deposit located instead of
// locating the same position again), and the deposits are the same integer sums, so particles and fields come out bit for bit as
// from the sweeps (tests/test_gpu_resident.py).  Only KE, a float64 sum whose order follows the launch geometry,
// may differ in its last bits.
#pragma once
#include "pic_device.h"
#include "pic_sweep.h"

namespace {

struct ResidentIO {
  const double* ext;       // [env][Ng] external field of the force evaluations, or null
  double *n, *E, *phi;     // [env][Ng] post-step refresh of the LAST step (pic.py:145-146)
  double *KE, *PE, *PEr;   // [env]
  double* hist;            // [nsteps][3][env] KE, PE, PE_reward after every step, or null
  void* snap;              // [nsteps][2][env][N] positions (as floats of the particle dtype) and velocities after every step, or null
  unsigned long long* bad;
  int nsteps;
  int num_envs;
  double c1, c2, d1, d2;   // Yoshida-4 (integration.py:62-69): c = (c1, c2, c2, c1), d = (0, d1, d2, d1)
};

// What the post-step refresh of a step needs besides the meshes (pic.py:145-146): where its results go, and the LDS
// scratch of the second scanning wave.
struct RefreshCtx {
  double* s2;              // Ng doubles: b = n - n0 of the refreshed density, then its G
  double* se;              // Ng doubles: phi
  double* slot2;           // 2 doubles
  double* ws;              // 2 NW doubles
  size_t row;              // env * Ng
  int env;
  double N_over_L;
};

// Field tile Es (gather layout: Ng + 2 slots, OFF for TSC) from the LDS mesh `acc_all`; sb: Ng doubles of scratch;
// xt: the external field of this environment in LDS, or null.
// While the scanning wave works, the others clear the meshes the coming particle phase deposits into (`z0`, and
// `z1` = the mesh just read, in sub-stage D), so that the phase needs no barrier of its own for that.
//
// kRefresh (sub-stage B of every step but the first of a launch): the post-step refresh of the PREVIOUS step -- density
// from the mesh `z0` still holds (sub-stage D's deposit of the final positions), E, phi and the three energies -- rides
// along: its sums go with the force field's sums, its two scans run in wave 1 next to wave 0's, its outputs with the
// field tile; it costs no barrier of its own.  Nothing in the step that follows reads what it produces.
template <typename T, typename A, int SHAPE, int NW, bool kRefresh>
__device__ __forceinline__ void resident_field(const A* __restrict__ acc_all, int R, int stride, const double* __restrict__ xt,
                                               int Ng, int fg, double scale, double n0, double dx, double* __restrict__ sb,
                                               double* __restrict__ slot, T* __restrict__ Es, A* __restrict__ z0,
                                               A* __restrict__ z1, const ResidentIO& io, const RefreshCtx& rc, double ke_prev,
                                               int prev_step) {
  constexpr int NT = NW * 64;
  constexpr int OFF = (SHAPE == PIC_TSC) ? 1 : 0;
  const int tid = threadIdx.x;
  const double unit = ldexp(1.0, -fg);
  for (int j = tid; j < Ng; j += NT)
    sb[j] = ((double)mesh_node_sum<A, SHAPE>(acc_all, R, stride, Ng, fg, j) * unit) * scale - n0;
  if (kRefresh)
    for (int j = tid; j < Ng; j += NT) {
      const double nj = ((double)mesh_node_sum<A, SHAPE>(z0, R, stride, Ng, fg, j) * unit) * scale;
      io.n[rc.row + j] = nj;
      rc.s2[j] = nj - n0;
    }
  __syncthreads();                                   // the meshes have been read: they may be cleared now
  scan_fields(sb, nullptr, Ng, dx, slot, 0);
  if (kRefresh) scan_fields(rc.s2, rc.se, Ng, dx, rc.slot2, 1);
  for (int i = tid; i < R * stride; i += NT) z0[i] = A{};
  if (z1) for (int i = tid; i < R * stride; i += NT) z1[i] = A{};
  __syncthreads();
  const double gmean = slot[0];
  for (int i = tid; i < Ng + 2; i += NT) {
    int node = i - OFF;
    node = node < 0 ? node + Ng : (node >= Ng ? node - Ng : node);
    const double gp = sb[node] - gmean;
    const double gm = sb[node == 0 ? Ng - 1 : node - 1] - gmean;
    double E = -0.5 * (gp + gm);
    if (xt) E += xt[node];
    Es[i] = (T)E;
  }
  if (kRefresh) {                                    // solve_block's outputs, reduction order included
    const double g2 = rc.slot2[0], pmean = rc.slot2[1];
    double e2 = 0.0;
    for (int j = tid; j < Ng; j += NT) {
      const double gp = rc.s2[j] - g2;
      const double gm = rc.s2[j == 0 ? Ng - 1 : j - 1] - g2;
      const double E = -0.5 * (gp + gm);
      io.E[rc.row + j] = E;
      e2 += E * E;
      io.phi[rc.row + j] = rc.se[j] - pmean;
    }
    const double wa = wave_sum(e2), wb = wave_sum(ke_prev);
    if ((tid & 63) == 0) {
      rc.ws[tid >> 6] = wa;
      rc.ws[NW + (tid >> 6)] = wb;
    }
  }
  __syncthreads();
  if (kRefresh && tid == 0) {                        // ws is not written again before the next field phase's barriers
    double S = 0.0, K = 0.0;
    for (int i = 0; i < NW; ++i) {
      S += rc.ws[i];
      K += rc.ws[NW + i];
    }
    const double pe = 0.5 * S * dx;
    io.PEr[rc.env] = pe;
    io.PE[rc.env] = pe * rc.N_over_L;
    io.KE[rc.env] = 0.5 * K;
    if (io.hist) {
      double* h3 = io.hist + (size_t)prev_step * 3 * io.num_envs;
      h3[rc.env] = 0.5 * K;
      h3[io.num_envs + rc.env] = pe * rc.N_over_L;
      h3[2 * (size_t)io.num_envs + rc.env] = pe;
    }
  }
}

template <typename P, typename A, int SHAPE, int PPT, int NW, bool kCarry>
__global__ __launch_bounds__(NW * 64) void resident_kernel(typename P::X* __restrict__ x, typename P::V* __restrict__ v,
                                                           ResidentIO io, SweepArgs a) {
  constexpr int NT = NW * 64;
  using T = typename P::W;
  using X = typename P::X;
  using V = typename P::V;

  // LDS: [R meshes: accA][R meshes: accB][sb][se][s2][xt: Ng doubles each][field tile Es]
  extern __shared__ __align__(16) unsigned char smem_raw[];
  const int Ng = a.Ng;
  const int stride = Ng + 2;
  const int R = a.R;
  A* accA = reinterpret_cast<A*>(smem_raw);
  A* accB = accA + (size_t)R * stride;
  double* sb = reinterpret_cast<double*>(accB + (size_t)R * stride);
  double* se = sb + Ng;
  double* s2 = se + Ng;
  double* xt_lds = s2 + Ng;
  T* Es = reinterpret_cast<T*>(xt_lds + Ng);
  __shared__ double ws[2 * NW];
  __shared__ double slot[2], slot2[2];

  const int tid = threadIdx.x;
  const int env = blockIdx.x;
  const int rep = (tid >> 6) & (R - 1);
  const Consts<P> k(a);
  const T c1 = (T)io.c1, c2 = (T)io.c2, d1 = (T)io.d1, d2 = (T)io.d2;
  X* xe = x + (size_t)env * a.ld;
  V* ve = v + (size_t)env * a.ld;
  const size_t row = (size_t)env * Ng;
  // the external field of the force evaluations is the same for every sub-stage and step of the call: one copy in LDS
  const double* xt = io.ext ? xt_lds : nullptr;
  if (io.ext)
    for (int j = tid; j < Ng; j += NT) xt_lds[j] = io.ext[row + j];
  RefreshCtx rc{s2, se, slot2, ws, row, env, a.N_over_L};

  // particle tid + s NT lives in slot s of lane tid
  X xs[PPT];
  V vs[PPT];
#pragma unroll
  for (int s = 0; s < PPT; ++s) {
    const long long i = (long long)s * NT + tid;
    xs[s] = i < a.N ? xe[i] : X(0);
    vs[s] = i < a.N ? ve[i] : V(0);
  }
  unsigned bad = 0u;

  // Where a sub-stage deposits a particle is where the next one gathers its field.  kCarry keeps cell and weights of
  // that position in registers from one sub-stage to the next (one locate per sub-stage instead of two, -15 % per
  // step); without it the gather locates again and the kernel needs ~50 registers fewer, so that two workgroups
  // share a CU -- the better trade once there are more environments than CUs (host: launch_resident).
  constexpr int NCAR = kCarry ? PPT : 1;
  int js[NCAR];
  T wgt[NCAR][SHAPE == PIC_TSC ? 3 : 2];

  // deposit of the first drift position q1 = x + (c1 v) dt (integration.py:42 with d1 = 0) into accA
  for (int i = tid; i < 2 * R * stride; i += NT) accA[i] = A{};
  __syncthreads();
#pragma unroll
  for (int s = 0; s < PPT; ++s) {
    if (kCarry) js[kCarry ? s : 0] = 0;
    if ((long long)s * NT + tid < a.N) {
      T w[3];
      X xw;
      int j;
      unsigned frac;
      const X q = drift<P>(xs[s], vs[s], c1, k, bad);
      locate<P, SHAPE>(q, k, xw, j, w, frac, bad);
      deposit<A, P, SHAPE>(accA + (size_t)rep * stride, j, w, frac, k.magic);
      if (kCarry) {
        js[kCarry ? s : 0] = j;
        wgt[kCarry ? s : 0][0] = w[0]; wgt[kCarry ? s : 0][1] = w[1];
        if (SHAPE == PIC_TSC) wgt[kCarry ? s : 0][SHAPE == PIC_TSC ? 2 : 0] = w[2];
      }
    }
  }
  __syncthreads();

  double ke = 0.0;               // sum of p^2 of this lane's particles after the step just made
  for (int step = 0; step < io.nsteps; ++step) {
    const double ke_prev = ke;
    ke = 0.0;
    for (int st = ST_B; st <= ST_D; ++st) {
      // sub-stage st reads the field of the deposit in `in` and deposits into `out` (D: also the next q1 into `in`)
      A* in = (st == ST_C) ? accB : accA;
      A* out = (st == ST_C) ? accA : accB;
      // (the previous step's post-step refresh goes with sub-stage B's field phase; the last step's follows the loop)
      if (st == ST_B && step > 0)
        resident_field<T, A, SHAPE, NW, true>(in, R, stride, xt, Ng, a.fg, a.scale, a.n0, a.dx, sb, slot, Es, out, nullptr,
                                              io, rc, ke_prev, step - 1);
      else
        resident_field<T, A, SHAPE, NW, false>(in, R, stride, xt, Ng, a.fg, a.scale, a.n0, a.dx, sb, slot, Es, out,
                                               st == ST_D ? in : nullptr, io, rc, 0.0, 0);
      // Yoshida coefficients of this sub-stage (integration.py:62-69): (c, d) = (c2, d1), (c3, d2), (c4, d3)
      const T c_cur = (st == ST_D) ? c1 : c2;
      const T d_cur = (st == ST_C) ? d2 : d1;
      A* acc = out + (size_t)rep * stride;
      A* acc2 = in + (size_t)rep * stride;
#pragma unroll
      for (int s = 0; s < PPT; ++s) {
        if ((long long)s * NT + tid < a.N) {
          T w[3];
          X xw;
          int j;
          unsigned frac;
          X q = xs[s];
          V p = vs[s];
          if (st == ST_B) q = drift<P>(q, p, c1, k, bad);         // q1 again (it is never stored)
          if (kCarry) {                                           // cell and weights of q, located by the previous deposit
            j = js[kCarry ? s : 0];
            w[0] = wgt[kCarry ? s : 0][0]; w[1] = wgt[kCarry ? s : 0][1];
            w[2] = (SHAPE == PIC_TSC) ? wgt[kCarry ? s : 0][SHAPE == PIC_TSC ? 2 : 0] : T(0);
          } else {
            locate<P, SHAPE>(q, k, xw, j, w, frac, bad);
          }
          const T E = gather_field<T, SHAPE>(Es, j, w);           // util.py:105 / pic.py:120
          p = p + (V)((d_cur * (-E)) * k.dt);                     // integration.py:32, pic.py:127
          q = drift<P>(q, p, c_cur, k, bad);                      // integration.py:42
          locate<P, SHAPE>(q, k, xw, j, w, frac, bad);
          deposit<A, P, SHAPE>(acc, j, w, frac, k.magic);
          if (st == ST_D) {
            q = xw;                                               // pic.py:139 (+ util.py:51)
            ke += (double)p * (double)p;
            const X qn = drift<P>(q, p, c1, k, bad);              // next step's q1
            X xn;
            locate<P, SHAPE>(qn, k, xn, j, w, frac, bad);
            deposit<A, P, SHAPE>(acc2, j, w, frac, k.magic);
          }
          if (kCarry) {
            js[kCarry ? s : 0] = j;
            wgt[kCarry ? s : 0][0] = w[0]; wgt[kCarry ? s : 0][1] = w[1];
            if (SHAPE == PIC_TSC) wgt[kCarry ? s : 0][SHAPE == PIC_TSC ? 2 : 0] = w[2];
          }
          xs[s] = q;
          vs[s] = p;
        }
      }
      __syncthreads();
    }

    if (io.snap) {                    // PIC.simulate's particle snapshots (pic.py:175-223), written from the registers
      using F = typename P::V;        // positions leave as floats of the particle dtype whatever their format
      F* sx = static_cast<F*>(io.snap) + ((size_t)step * 2 * io.num_envs + env) * (size_t)a.N;
      F* sv = sx + (size_t)io.num_envs * (size_t)a.N;
#pragma unroll
      for (int s = 0; s < PPT; ++s) {
        const long long i = (long long)s * NT + tid;
        if (i < a.N) {
          F xf = (F)pos_to_length<P>(xs[s], a.L);
          if (P::kFixed && xf >= (F)a.L) xf = F(0);
          sx[i] = xf;
          sv[i] = vs[s];
        }
      }
    }
  }

  // post-step refresh of the last step (pic.py:145-146; no external field: pic.py:114-117) from the deposit in accB
  if (io.nsteps > 0) {
    const double unit = ldexp(1.0, -a.fg);
    for (int j = tid; j < Ng; j += NT) {
      const double nj = ((double)mesh_node_sum<A, SHAPE>(accB, R, stride, Ng, a.fg, j) * unit) * a.scale;
      io.n[row + j] = nj;
      sb[j] = nj - a.n0;
    }
    __syncthreads();
    SolveOut o{};
    o.E = io.E; o.phi = io.phi; o.KE = io.KE; o.PE = io.PE; o.PEr = io.PEr;
    solve_block<NW>(o, env, Ng, a.dx, a.N_over_L, ke, sb, se, ws, slot);
    if (io.hist && tid == 0) {        // this thread wrote the three energies a moment ago
      double* h3 = io.hist + (size_t)(io.nsteps - 1) * 3 * io.num_envs;
      h3[env] = io.KE[env];
      h3[io.num_envs + env] = io.PE[env];
      h3[2 * (size_t)io.num_envs + env] = io.PEr[env];
    }
  }

#pragma unroll
  for (int s = 0; s < PPT; ++s) {
    const long long i = (long long)s * NT + tid;
    if (i < a.N) {
      xe[i] = xs[s];
      ve[i] = vs[s];
    }
  }
  if (bad) atomicAdd(io.bad, (unsigned long long)bad);
}

}  // namespace
