// pic_sweep.h -- the push sweeps: in-prologue field solve, one particle through one Yoshida sub-stage, slab-row
// flush and sweep_kernel<T, A, SHAPE, STAGE> itself.
#pragma once
#include "pic_device.h"

namespace {

// Field tile for a sweep workgroup, computed in its own prologue (256 threads) from the slab the previous
// sweep wrote: density -> G = dx cumsum(n - n0) - mean -> E_j = -(G_{j+1/2} + G_{j-1/2})/2 (+ E_ext), the same
// scan solve as field_solve_kernel.  Every workgroup of an environment repeats it (the rows come from L2);
// in exchange a step needs no field-solve launch between sweeps.  sb: Ng doubles of LDS scratch.
template <typename T, int OFF>
__device__ __forceinline__ void prologue_field(const double* __restrict__ slab, int nblk, const double* __restrict__ ext,
                                               int Ng, double scale, double n0, double dx, double* __restrict__ sb,
                                               double* __restrict__ ws, T* __restrict__ Es) {
  const int tid = threadIdx.x;
  for (int j = tid; j < Ng; j += BLOCK) {
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0, s4 = 0.0, s5 = 0.0, s6 = 0.0, s7 = 0.0;
    int b = 0;
    for (; b + 7 < nblk; b += 8) {          // 8 independent loads in flight per lane
      s0 += slab[(size_t)b * Ng + j];
      s1 += slab[(size_t)(b + 1) * Ng + j];
      s2 += slab[(size_t)(b + 2) * Ng + j];
      s3 += slab[(size_t)(b + 3) * Ng + j];
      s4 += slab[(size_t)(b + 4) * Ng + j];
      s5 += slab[(size_t)(b + 5) * Ng + j];
      s6 += slab[(size_t)(b + 6) * Ng + j];
      s7 += slab[(size_t)(b + 7) * Ng + j];
    }
    for (; b < nblk; ++b) s0 += slab[(size_t)b * Ng + j];
    sb[j] = (((s0 + s1) + (s2 + s3)) + ((s4 + s5) + (s6 + s7))) * scale - n0;
  }
  __syncthreads();
  const int m = (Ng + BLOCK - 1) / BLOCK;
  const int lo = min(tid * m, Ng), hi = min(lo + m, Ng);
  double loc = 0.0;
  for (int j = lo; j < hi; ++j) loc += sb[j];
  double tot;
  double run = block_excl_scan<WAVES>(loc, ws, tot);
  loc = 0.0;
  for (int j = lo; j < hi; ++j) {
    run += sb[j];
    const double g = run * dx;
    sb[j] = g;
    loc += g;
  }
  const double gmean = block_sum<WAVES>(loc, ws) / (double)Ng;     // syncs: sb holds G everywhere
  for (int i = tid; i < Ng + 2; i += BLOCK) {
    int node = i - OFF;
    node = node < 0 ? node + Ng : (node >= Ng ? node - Ng : node);
    const double gp = sb[node] - gmean;
    const double gm = sb[node == 0 ? Ng - 1 : node - 1] - gmean;
    double E = -0.5 * (gp + gm);
    if (ext) E += ext[node];
    Es[i] = (T)E;
  }
  __syncthreads();
}

// One particle through one sub-stage.  Stages D / REFRESH also deposit the NEXT step's first drift
// position q1 = x' + (c1 p) dt into a second mesh (acc2), which is exactly what sweep A of the next
// step would deposit from the stored x', p -- so that sweep (a full read of x and v) is skipped.
template <typename T, typename A, int SHAPE, int STAGE>
__device__ __forceinline__ void push_one(T& xq, T& vp, const T* __restrict__ Es, A* __restrict__ acc,
                                         A* __restrict__ acc2, T L, T dx, T rdx, T dt, T c_prev, T c_cur, T d_cur,
                                         T c_next, int Ng, double& ke, unsigned& bad) {
#if defined(PIC_EXP_LEVEL) && PIC_EXP_LEVEL <= 1      // timing experiment: stream only (results are wrong)
  xq = xq + T(0); vp = vp + T(0);
  return;
#endif
  T w[3];
  T xw;
  int j;
  T q = xq, p = vp;
  if (STAGE == ST_A) {
    q = q + (c_cur * p) * dt;                                   // integration.py:42, c1
  } else if (STAGE == ST_B || STAGE == ST_C || STAGE == ST_D) {
    if (STAGE == ST_B) q = q + (c_prev * p) * dt;               // q1 again (it is never stored)
    locate<T, SHAPE>(q, L, dx, rdx, Ng, xw, j, w, bad);
#if defined(PIC_EXP_LEVEL) && PIC_EXP_LEVEL == 2      // timing experiment: arithmetic only, no LDS traffic
    T E = w[0] * T(0.25) + w[1] * T(0.5) + T(j) * T(1e-30);
#else
    T E = gather_field<T, SHAPE>(Es, j, w);                     // util.py:105 / pic.py:120
#endif
    p = p + (d_cur * (-E)) * dt;                                // integration.py:32, pic.py:127
    q = q + (c_cur * p) * dt;                                   // integration.py:42
  }
  locate<T, SHAPE>(q, L, dx, rdx, Ng, xw, j, w, bad);
#if defined(PIC_EXP_LEVEL) && (PIC_EXP_LEVEL == 2 || PIC_EXP_LEVEL == 3)   // no deposit (3: gather kept)
  asm volatile("" ::"v"(w[0]), "v"(w[1]), "v"(j));
#else
  deposit<A, T, SHAPE>(acc, j, w);
#endif
  if (STAGE == ST_D || STAGE == ST_REFRESH) {
    q = xw;                                                     // pic.py:139 (+ util.py:51)
    ke += (double)p * (double)p;
#if !defined(PIC_EXP_D) || PIC_EXP_D < 2               // timing experiments: 1 = no second deposit, 2 = no second locate either
    T qn = q + (c_next * p) * dt;                               // next step's q1 (integration.py:42, c1)
    T xn;
    locate<T, SHAPE>(qn, L, dx, rdx, Ng, xn, j, w, bad);
#if defined(PIC_EXP_D) && PIC_EXP_D == 1
    asm volatile("" ::"v"(w[0]), "v"(w[1]), "v"(j));
#else
    deposit<A, T, SHAPE>(acc2, j, w);
#endif
#endif
  }
  xq = q;
  vp = p;
}

// fold the periodic ghost slots and the replicas of one LDS mesh, store it as this workgroup's slab row
template <typename A, int SHAPE>
__device__ __forceinline__ void flush_mesh(const A* __restrict__ acc_all, int R, int stride, int Ng,
                                           double* __restrict__ row) {
  constexpr int OFF = (SHAPE == PIC_TSC) ? 1 : 0;
  if constexpr (std::is_same<A, fix_t>::value) {
    for (int c = threadIdx.x; c < Ng; c += BLOCK) {
      const int cm = c == 0 ? Ng - 1 : c - 1;
      unsigned long long own = 0ull, left = 0ull;
      for (int r = 0; r < R; ++r) {
        own += acc_all[(size_t)r * stride + c];
        left += acc_all[(size_t)r * stride + cm];
      }
      const long long mask = (1ll << FX_LOW) - 1;
      const long long q = ((long long)(own >> FX_LOW) << FX_FRAC) - ((long long)own & mask) + ((long long)left & mask);
      row[c] = (double)q * (1.0 / (double)(1 << FX_FRAC));      // exact: |q| < 2^45
    }
    return;
  }
  for (int c = threadIdx.x; c < Ng; c += BLOCK) {
    double s = 0.0;
    for (int r = 0; r < R; ++r) {
      const A* ar = acc_all + (size_t)r * stride;
      double t = (double)ar[c + OFF];
      if (SHAPE == PIC_CIC) {
        if (c == 0) t += (double)ar[Ng];
      } else {
        if (c == 0) t += (double)ar[Ng + 1];
        if (c == Ng - 1) t += (double)ar[0];
      }
      s += t;
    }
    row[c] = s;
  }
}

template <typename T, typename A, int SHAPE, int STAGE>
__global__ __launch_bounds__(BLOCK) void sweep_kernel(T* __restrict__ x, T* __restrict__ v,
                                                      const double* __restrict__ Ef,
                                                      const double* __restrict__ slab_in,
                                                      const double* __restrict__ ext_in,
                                                      double* __restrict__ part, double* __restrict__ part2,
                                                      double* __restrict__ ke_part,
                                                      unsigned long long* __restrict__ bad_count, SweepArgs a) {
  constexpr int OFF = (SHAPE == PIC_TSC) ? 1 : 0;
  constexpr int VEC = VecOf<T>::n;
  using V = typename VecOf<T>::type;
  constexpr bool kGather = (STAGE == ST_B || STAGE == ST_C || STAGE == ST_D);
  constexpr bool kStore = (STAGE == ST_B || STAGE == ST_C || STAGE == ST_D || STAGE == ST_REFRESH);
  constexpr bool kStoreV = (STAGE == ST_B || STAGE == ST_C || STAGE == ST_D);
  constexpr bool kReadV = (STAGE != ST_PROBE);
  constexpr bool kDual = (STAGE == ST_D || STAGE == ST_REFRESH);

  // LDS: [R meshes: acc][R meshes: acc2 (dual stages)][field tile Es]
  extern __shared__ __align__(16) unsigned char smem_raw[];
  const int Ng = a.Ng;
  const int stride = Ng + 2;
  const int nacc = (kDual ? 2 : 1) * a.R * stride;
  A* acc_all = reinterpret_cast<A*>(smem_raw);
  A* acc2_all = acc_all + (size_t)a.R * stride;
  T* Es = reinterpret_cast<T*>(smem_raw + (size_t)2 * a.R * stride * sizeof(A));
  __shared__ double red[WAVES];

  const int tid = threadIdx.x;
  // Consecutive sweeps walk memory in opposite directions: what the previous sweep wrote last (still
  // in the 256 MB Infinity Cache) is what this one reads first.
  const int env = a.env0 + (a.reverse ? (int)gridDim.y - 1 - (int)blockIdx.y : (int)blockIdx.y);
  const int blk = a.reverse ? (int)gridDim.x - 1 - (int)blockIdx.x : (int)blockIdx.x;

#ifdef PIC_EXP_STAMPB
  const unsigned long long sb0 = wall_clock64();
#endif
#if defined(PIC_EXP_LEVEL) && PIC_EXP_LEVEL <= 0
  constexpr bool kPrologue = false;      // timing experiment: no LDS zeroing / field tile / barrier
#else
  constexpr bool kPrologue = true;
#endif
  // the mesh region doubles as scratch of the in-prologue field solve, so it is zeroed after that solve
  const bool solve_here = kGather && slab_in != nullptr;
  if (kPrologue && !solve_here) for (int i = tid; i < nacc; i += BLOCK) acc_all[i] = A(0);
#ifdef PIC_EXP_STAMPB
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const unsigned long long sbz = wall_clock64();
#endif
  if (kGather && kPrologue) {
    if (slab_in) {
      // no field-solve launch ran before this sweep: solve here, with the (not yet zeroed) mesh region as scratch
      prologue_field<T, OFF>(slab_in + (size_t)env * a.nblk * Ng, a.nblk, ext_in ? ext_in + (size_t)env * Ng : nullptr,
                             Ng, a.scale, a.n0, a.dx, reinterpret_cast<double*>(smem_raw), red, Es);
      for (int i = tid; i < nacc; i += BLOCK) acc_all[i] = A(0);
    } else {
      const double* Ee = Ef + (size_t)env * Ng;
      for (int i = tid; i < stride; i += BLOCK) {
        int node = i - OFF;
        node = node < 0 ? node + Ng : (node >= Ng ? node - Ng : node);
        Es[i] = (T)Ee[node];
      }
    }
  }
#ifdef PIC_EXP_STAMPB
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  const unsigned long long sbe = wall_clock64();
#endif
  if (kPrologue) __syncthreads();

  const int rep = (tid >> 6) & (a.R - 1);
  A* acc = acc_all + (size_t)rep * stride;
  A* acc2 = acc2_all + (size_t)rep * stride;
  const T L = (T)a.L, dx = (T)a.dx, rdx = (T)a.rdx, dt = (T)a.dt;
  const T c_prev = (T)a.c_prev, c_cur = (T)a.c_cur, d_cur = (T)a.d_cur, c_next = (T)a.c_next;

  T* xe = x + (size_t)env * a.ld;
  T* ve = v + (size_t)env * a.ld;
  const long long step = (long long)BLOCK * VEC;

  double ke = 0.0;
  unsigned bad = 0u;
#ifdef PIC_EXP_STAMPB
  const unsigned long long sb1 = wall_clock64();
#endif
  // A workgroup owns the runs blk, blk + nblk, blk + 2 nblk, ... of `chunk` particles of its environment.
  // chunk = ceil(N / nblk) gives every workgroup one contiguous region; a chunk of a few tiles interleaves
  // the workgroups of an environment, so that the addresses in flight form a compact moving window.
  for (long long begin = (long long)blk * a.chunk; begin < a.N; begin += (long long)a.nblk * a.chunk) {
  long long end = begin + a.chunk;
  if (end > a.N) end = a.N;
  long long i = begin + (long long)tid * VEC;
#if PIC_PIPE == 0
  // PIC_TILES tiles per lane per iteration: all their loads are issued before the first particle is
  // pushed, so a wave keeps PIC_TILES x 2 KB of requests in flight while it waits.
#ifdef PIC_EXP_STAMP   // diagnostic build: wall-clock (10 ns ticks) spent waiting for loads vs pushing, per wave 0
  unsigned long long st_mem = 0, st_cmp = 0, st_n = 0;
#endif
  for (; i + (long long)(PIC_TILES - 1) * step + VEC <= end; i += (long long)PIC_TILES * step) {
    V xv[PIC_TILES], vv[PIC_TILES];
#ifdef PIC_EXP_STAMP
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long st0 = wall_clock64();
    __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
    for (int t = 0; t < PIC_TILES; ++t) {
      xv[t] = PIC_LOAD(reinterpret_cast<const V*>(xe + i + (long long)t * step));
      vv[t] = V{};
      if (kReadV) vv[t] = PIC_LOAD(reinterpret_cast<const V*>(ve + i + (long long)t * step));
    }
#ifdef PIC_EXP_STAMP
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long st1 = wall_clock64();
    __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
    for (int t = 0; t < PIC_TILES; ++t) {
      T* xs = reinterpret_cast<T*>(&xv[t]);
      T* vs = reinterpret_cast<T*>(&vv[t]);
#pragma unroll
      for (int k = 0; k < VEC; ++k) {
        T pv = kReadV ? vs[k] : T(0);
        push_one<T, A, SHAPE, STAGE>(xs[k], pv, Es, acc, acc2, L, dx, rdx, dt, c_prev, c_cur, d_cur, c_next, Ng, ke, bad);
        if (kReadV) vs[k] = pv;
      }
      if (kStore) {
        PIC_STORE(xv[t], reinterpret_cast<V*>(xe + i + (long long)t * step));
        if (kStoreV) PIC_STORE(vv[t], reinterpret_cast<V*>(ve + i + (long long)t * step));
      }
    }
#ifdef PIC_EXP_STAMP
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned long long st2 = wall_clock64();
    __builtin_amdgcn_sched_barrier(0);
    st_mem += st1 - st0;
    st_cmp += st2 - st1;
    st_n += 1;
#endif
  }
#ifdef PIC_EXP_STAMP
  if (tid == 0 && (STAGE == ST_C)) {
    atomicAdd(&bad_count[1], st_mem);
    atomicAdd(&bad_count[2], st_cmp);
    atomicAdd(&bad_count[3], st_n);
  }
#endif
  for (; i + VEC <= end; i += step) {          // leftover whole tiles
    V xv = PIC_LOAD(reinterpret_cast<const V*>(xe + i));
    V vv = {};
    if (kReadV) vv = PIC_LOAD(reinterpret_cast<const V*>(ve + i));
    T* xs = reinterpret_cast<T*>(&xv);
    T* vs = reinterpret_cast<T*>(&vv);
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      T pv = kReadV ? vs[k] : T(0);
      push_one<T, A, SHAPE, STAGE>(xs[k], pv, Es, acc, acc2, L, dx, rdx, dt, c_prev, c_cur, d_cur, c_next, Ng, ke, bad);
      if (kReadV) vs[k] = pv;
    }
    if (kStore) {
      PIC_STORE(xv, reinterpret_cast<V*>(xe + i));
      if (kStoreV) PIC_STORE(vv, reinterpret_cast<V*>(ve + i));
    }
  }
#else
  // Software pipeline (double buffer).  The next tile's loads are issued by inline asm BEFORE the current
  // tile is pushed: written as plain C loads, hipcc proves they cannot alias the stores and sinks them back
  // down to their use, so memory wait and push never overlap (stamped: 2.0 us + 1.8 us per iteration, all
  // waves of a SIMD in lockstep).  hipcc does not count asm loads in its own s_waitcnt, so the wait is
  // explicit: in issue order the younger VMEM operations at that point are exactly this iteration's stores
  // (kNumStores), hence vmcnt(kNumStores).  The "+v" ties keep every use of the prefetched registers behind
  // the wait (cdna_hip_programming.md 5.7).
  constexpr int kNumStores = kStore ? (kStoreV ? 2 : 1) : 0;
  bool have = (i + VEC <= end);
  V cx = {}, cv = {};
  if (have) {
    cx = PIC_LOAD(reinterpret_cast<const V*>(xe + i));
    if (kReadV) cv = PIC_LOAD(reinterpret_cast<const V*>(ve + i));
  }
#ifdef PIC_EXP_STAMP
  unsigned long long st_mem = 0, st_cmp = 0, st_n = 0;
#endif
  while (have) {
    const long long in = i + step;
    const bool hn = (in + VEC <= end);
    V nx = {}, nv = {};
#ifdef PIC_EXP_STAMP
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long st0 = wall_clock64();
    __builtin_amdgcn_sched_barrier(0);
#endif
    if (hn) {
      if (kReadV)
        asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dwordx4 %1, %3, off"
                     : "=&v"(nx), "=&v"(nv) : "v"(xe + in), "v"(ve + in) : "memory");
      else
        asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(nx) : "v"(xe + in) : "memory");
    }
    T* xs = reinterpret_cast<T*>(&cx);
    T* vs = reinterpret_cast<T*>(&cv);
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      T pv = kReadV ? vs[k] : T(0);
      push_one<T, A, SHAPE, STAGE>(xs[k], pv, Es, acc, acc2, L, dx, rdx, dt, c_prev, c_cur, d_cur, c_next, Ng, ke, bad);
      if (kReadV) vs[k] = pv;
    }
    if (kStore) {
      PIC_STORE(cx, reinterpret_cast<V*>(xe + i));
      if (kStoreV) PIC_STORE(cv, reinterpret_cast<V*>(ve + i));
    }
#ifdef PIC_EXP_STAMP
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned long long st1 = wall_clock64();
    __builtin_amdgcn_sched_barrier(0);
#endif
    if (kNumStores == 2) asm volatile("s_waitcnt vmcnt(2)" : "+v"(nx), "+v"(nv) : : "memory");
    else if (kNumStores == 1) asm volatile("s_waitcnt vmcnt(1)" : "+v"(nx), "+v"(nv) : : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" : "+v"(nx), "+v"(nv) : : "memory");
#ifdef PIC_EXP_STAMP
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long st2 = wall_clock64();
    __builtin_amdgcn_sched_barrier(0);
    st_cmp += st1 - st0;      // issue prefetch + push + issue stores
    st_mem += st2 - st1;      // residual wait for the prefetched tile
    st_n += 1;
#endif
    cx = nx;
    cv = nv;
    i = in;
    have = hn;
  }
#ifdef PIC_EXP_STAMP
  if (tid == 0 && (STAGE == ST_C)) {
    atomicAdd(&bad_count[1], st_mem);
    atomicAdd(&bad_count[2], st_cmp);
    atomicAdd(&bad_count[3], st_n);
  }
#endif
#endif
  for (long long k = i; k < end; ++k) {       // ragged tail (fewer than VEC particles left for this lane)
    T xq = xe[k];
    T pv = kReadV ? ve[k] : T(0);
    push_one<T, A, SHAPE, STAGE>(xq, pv, Es, acc, acc2, L, dx, rdx, dt, c_prev, c_cur, d_cur, c_next, Ng, ke, bad);
    if (kStore) {
      xe[k] = xq;
      if (kStoreV) ve[k] = pv;
    }
  }
  }   // runs
#ifdef PIC_EXP_STAMPB
  const unsigned long long sb2 = wall_clock64();      // this wave's loop is done
#endif
  __syncthreads();
#ifdef PIC_EXP_STAMPB
  const unsigned long long sb3 = wall_clock64();      // every wave's loop is done
#endif

#if defined(PIC_EXP_LEVEL) && PIC_EXP_LEVEL <= -1
  if (ke < -1.0) part[0] = ke;           // timing experiment: no flush, no KE reduction
  return;
#endif
  const size_t rowi = ((size_t)env * a.nblk + blk) * Ng;
  flush_mesh<A, SHAPE>(acc_all, a.R, stride, Ng, part + rowi);
  if (kDual) flush_mesh<A, SHAPE>(acc2_all, a.R, stride, Ng, part2 + rowi);

  if (kDual) {
    double w = wave_sum(ke);
    if ((tid & 63) == 0) red[tid >> 6] = w;
    __syncthreads();
    if (tid == 0) {
      double s = 0.0;
      for (int k = 0; k < WAVES; ++k) s += red[k];
      ke_part[(size_t)env * a.nblk + blk] = s;
    }
  }
  if (bad) atomicAdd(bad_count, (unsigned long long)bad);
#ifdef PIC_EXP_STAMPB
  if (STAGE == ST_C && (tid & 63) == 0) {
    const unsigned long long sb4 = wall_clock64();
    // [1] prologue+loop of this wave, [2] wait for the slowest wave of the workgroup, [3] flush; counts in [0]'s upper bits unused
    const bool late = (unsigned)(blockIdx.y * gridDim.x + blockIdx.x) >= 2048u;   // not in the first resident set
    if (late) {
      atomicAdd(&bad_count[1], ((sbz - sb0) << 32) | (sbe - sbz));     // zero LDS | field tile load
      atomicAdd(&bad_count[2], ((sb1 - sbe) << 32) | (sb2 - sb1));     // barrier | loop
      atomicAdd(&bad_count[3], ((sb3 - sb2) << 32) | 1ull);            // straggler wait | count
    }
    (void)sb4;
  }
#endif
}

}  // namespace
