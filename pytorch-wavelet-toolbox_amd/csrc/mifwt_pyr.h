// mifwt_pyr.h — building blocks of the multi-level 2-D analysis kernel (mifwt_dwt2_fwd_pyr.hip): lags between the levels,
// rolling vertical pass, packed-FMA forms, compile-time loops over runtime phases, per-wave cycle profiling.
#pragma once
#include <type_traits>

#include "mifwt_stream.h"

namespace mifwt {

extern unsigned long long* g_pyr_prof;

enum PyrRole { kRoleL1 = 0, kRoleL2 = 1, kRoleL3 = 2, kRoleLoad = 3, kRoleXchg = 4 };  // what a wave of a workgroup does

constexpr uint32_t kPyrOob = 0x80000000u;  // byte offset beyond every buffer resource: loads return 0, stores are dropped

// Lags in 8-row steps.  Level 2 of step s runs in the step's SECOND half (after the barrier that follows level 1's first two
// rows of the step), and reads ring-1 rows up to pair index 4 (s - D2) + L/2 + 2: it must stay below the pair 4 s + 2 being
// written meanwhile (4 D2 > L/2); at the top of the plane the mirrored row L - 2 (pair L - 2 + L/2 - 1) must exist (4 D2 >= L - 2
// + L/2 - 2).  Level 3 of step s runs BEFORE level 2 of step s in the same wave: filter delay 2 (D3 - D2 - 1) >= L/2 - 1, at
// the top 2 (D3 - D2 - 1) >= L - 2 + L/2 - 2.  (Ring depth 16 covers all of them for L <= 8: DESIGN.md §4.1c.)
constexpr int pyr_cdiv(int a, int b) { return a <= 0 ? 0 : (a + b - 1) / b; }
constexpr int pyr_lag2_inner(int L) { return (L / 2) / 4 + 1; }
constexpr int pyr_lag2(int L) { const int b = pyr_cdiv(L - 2 + L / 2 - 2, 4); return b > pyr_lag2_inner(L) ? b : pyr_lag2_inner(L); }
constexpr int pyr_lag3_inner(int L) { return pyr_lag2_inner(L) + 1 + pyr_cdiv(L / 2 - 1, 2); }
constexpr int pyr_lag3(int L) { const int a = L / 2 - 1, b = L - 2 + L / 2 - 2; return pyr_lag2(L) + 1 + pyr_cdiv(a > b ? a : b, 2); }

// cache policy of the sub-band / output stores of the streaming kernels (aux operand of the buffer stores: sc0 = 1, nt = 2, sc1 = 16;
// tools/wbench.hip)
#ifndef MIFWT_ST_AUX
#define MIFWT_ST_AUX 0
#endif
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t pyr_rsrc(const void* p, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
// 64 lanes x 16 B -> LDS [lds_addr + 16 lane); global address = resource base + voff (per lane) + soff; non-temporal
__device__ __forceinline__ void pyr_store1(float v, rsrc_t rsrc, uint32_t voff, uint32_t soff) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v), rsrc, voff, soff, MIFWT_ST_AUX);
}
// workgroup barrier; with profiling on, the cycles spent in it are added to `waited`
template <bool PROF>
__device__ __forceinline__ void pyr_barrier(unsigned long long& waited) {
  if constexpr (PROF) {
    const unsigned long long t0 = __builtin_readcyclecounter();
    __syncthreads();
    waited += __builtin_readcyclecounter() - t0;
  } else {
    __syncthreads();
  }
}
template <int N>
__device__ __forceinline__ void pyr_wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// packed FMAs acc (+)= (tap.x, tap.y) * pair.x / pair.y with the tap pair in an SGPR pair: with the three-operand pattern of the
// passes (accumulator, tap, sample all distinct) 4.7 cycles per wave-instruction at two waves per SIMD against 5.5 for taps held
// in VGPR pairs (tools/ubench.hip "distinct" rows, profiles/r02_ubench_valu_copy.txt)
__device__ __forceinline__ void vfma_lo(f2& acc, const f2 tap, const f2 pair) {
  asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc) : "s"(tap), "v"(pair));
}
__device__ __forceinline__ void vfma_hi(f2& acc, const f2 tap, const f2 pair) {
  asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(acc) : "s"(tap), "v"(pair));
}
__device__ __forceinline__ f2 vmul_lo(const f2 tap, const f2 pair) {
  f2 r;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(r) : "s"(tap), "v"(pair));
  return r;
}
__device__ __forceinline__ f2 vmul_hi(const f2 tap, const f2 pair) {
  f2 r;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(r) : "s"(tap), "v"(pair));
  return r;
}

// rolling vertical pass: the L/2 outputs in flight of NC columns; lo = (aa, da), hi = (ad, dd) per column.  Output i lives in
// slot i mod L/2 for its whole life, so nothing is ever copied: the pair index modulo L/2 (R) is a compile-time constant at
// every call site (the callers unroll or switch over it)
template <int L, int NC>
struct PyrAcc {
  static constexpr int HP = L / 2;
  f2 lo[HP][NC], hi[HP][NC];
  __device__ __forceinline__ void clear() {
#pragma unroll
    for (int q = 0; q < HP; ++q)
#pragma unroll
      for (int c = 0; c < NC; ++c) lo[q][c] = hi[q][c] = (f2){0.f, 0.f};
  }
  // one row of horizontally filtered samples hv[c] = (h_lo, h_hi) of pair p (R = p mod HP); PH = 0: first row of the pair, 1: second
  template <int PH, int R>
  __device__ __forceinline__ void feed(const f2 (&tap)[L], const f2 (&hv)[NC]) {
#pragma unroll
    for (int q = 0; q < HP; ++q) {
      const int sl = (R - q + HP) % HP;  // output p - q
      const int m = L - 1 - 2 * q - PH;
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        if (q == 0 && PH == 0) {
          lo[sl][c] = vmul_lo(tap[m], hv[c]);
          hi[sl][c] = vmul_hi(tap[m], hv[c]);
        } else {
          vfma_lo(lo[sl][c], tap[m], hv[c]);
          vfma_hi(hi[sl][c], tap[m], hv[c]);
        }
      }
    }
  }
  // slot of the output that pair p completes (p - (HP - 1))
  static constexpr int done(int R) { return (R + 1) % HP; }
};

// f(integral_constant<int, i>) for i = 0 .. N - 1, unrolled at compile time (loop indices that feed template arguments)
template <int N, int I = 0, typename F>
__device__ __forceinline__ void pyr_static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    pyr_static_for<N, I + 1>(f);
  }
}

// f(integral_constant<int, r>) for the runtime r in [0, N)
template <int N, typename F>
__device__ __forceinline__ void pyr_dispatch(int r, F&& f) {
  if constexpr (N == 1) {
    f(std::integral_constant<int, 0>{});
  } else if constexpr (N == 2) {
    if (r == 0) f(std::integral_constant<int, 0>{});
    else f(std::integral_constant<int, 1>{});
  } else if constexpr (N == 3) {
    if (r == 0) f(std::integral_constant<int, 0>{});
    else if (r == 1) f(std::integral_constant<int, 1>{});
    else f(std::integral_constant<int, 2>{});
  } else {
    static_assert(N == 4, "filter lengths up to 8");
    if (r < 2) {
      if (r == 0) f(std::integral_constant<int, 0>{});
      else f(std::integral_constant<int, 1>{});
    } else {
      if (r == 2) f(std::integral_constant<int, 2>{});
      else f(std::integral_constant<int, 3>{});
    }
  }
}

}  // namespace mifwt
