// picstep_timeline.hip -- DIAGNOSTIC build of libpicstep: the product translation unit with its timeline hooks
// (PIC_STAMP in csrc/pic_device.h) switched on.  Thread 0 of every workgroup writes the shader clock (s_memtime) at each
// hook into a buffer of its own that no kernel reads; slot 1 / 27 hold the 100 MHz wall clock (s_memrealtime) at entry /
// exit for aligning workgroups with each other, slot 28 the XCC id.  Never shipped: built by profiles/timeline/build.sh into
// profiles/bin/, loaded by profiles/timeline/timeline.py only.  Stamping costs time itself (~ +10 % wave cycles): read
// the intervals against each other, not against the product's durations.
#include <hip/hip_runtime.h>

constexpr int TL_LAUNCHES = 64, TL_WGS = 1024, TL_SLOTS = 32;
__device__ unsigned long long g_tl[TL_LAUNCHES][TL_WGS][TL_SLOTS];
__device__ int g_tl_launch;      // index of the launch being recorded; its workgroup (0, 0) bumps it on exit

// Called by thread 0 of a workgroup only.  The launch index is read once, at the first hook, and kept in LDS: a stamp is then
// s_memtime + one store (no load in front of it), and a workgroup records into the launch it started in even when workgroup 0
// of that launch has already bumped the index.
__device__ __forceinline__ void tl_stamp(int slot) {
  __shared__ int tl_l;
  const unsigned long long t = __builtin_amdgcn_s_memtime();
  if (slot == 0) tl_l = __builtin_nontemporal_load(&g_tl_launch);
  const int l = tl_l;
  const int wg = blockIdx.y * gridDim.x + blockIdx.x;
  if ((unsigned)l < (unsigned)TL_LAUNCHES && (unsigned)wg < (unsigned)TL_WGS && (unsigned)slot < (unsigned)TL_SLOTS) {   // (unsigned: a kernel without hook 0 reads an unset tl_l)
    g_tl[l][wg][slot] = t;
    if (slot == 0) {
      g_tl[l][wg][1] = __builtin_amdgcn_s_memrealtime();
      g_tl[l][wg][28] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) & 0xF;   // HW_REG_XCC_ID
      g_tl[l][wg][29] = ((unsigned long long)gridDim.x << 32) | gridDim.y;
    }
    if (slot == 26) {
      g_tl[l][wg][27] = __builtin_amdgcn_s_memrealtime();
      if (wg == 0) g_tl_launch = l + 1;
    }
  }
}
#define PIC_STAMP(slot) do { if (threadIdx.x == 0) tl_stamp(slot); } while (0)
#define PIC_STAMP_LOADS(slot) do { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); if (threadIdx.x == 0) tl_stamp(slot); } while (0)

#include "../../optimal-control-1d-electrostatic-plasma_amd/csrc/picstep.hip"

extern "C" int pic_timeline_reset(void) {
  int zero = 0;
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  void* p = nullptr;
  if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_tl)) != hipSuccess) return -1;
  if (hipMemset(p, 0, sizeof(unsigned long long) * TL_LAUNCHES * TL_WGS * TL_SLOTS) != hipSuccess) return -1;
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_tl_launch), &zero, sizeof(int)) != hipSuccess) return -1;
  return hipDeviceSynchronize() == hipSuccess ? 0 : -1;
}

// dst: [TL_LAUNCHES][TL_WGS][TL_SLOTS] uint64; returns the number of launches recorded
extern "C" int pic_timeline_read(unsigned long long* dst) {
  int n = 0;
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  if (hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_tl), sizeof(unsigned long long) * TL_LAUNCHES * TL_WGS * TL_SLOTS) != hipSuccess) return -1;
  if (hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_tl_launch), sizeof(int)) != hipSuccess) return -1;
  return n;
}
