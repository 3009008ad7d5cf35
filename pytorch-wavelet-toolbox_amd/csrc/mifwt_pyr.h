// mifwt_pyr.h — building blocks of the multi-level 2-D analysis kernel (mifwt_dwt2_fwd_pyr.hip): lags between the levels,
// rolling vertical pass, packed-FMA forms, compile-time loops over runtime phases, per-wave cycle profiling.
#pragma once
#include <type_traits>

#include "mifwt_stream.h"

namespace mifwt {

extern unsigned long long* g_pyr_prof;

enum PyrRole { kRoleL1 = 0, kRoleL2 = 1, kRoleL3 = 2, kRoleLoad = 3, kRoleTail = 4 };  // what a wave of a workgroup does

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
// 16-byte store, non-temporal (measured on config 2 with 516-float rows and rotating output sets, profiles/r04b_st16_ab.txt: nt 141 us,
// default policy 150, write-through sc0 sc1 166 — against 148 for the 8-byte stores on the same planes; the policy is a build-time
// switch for A/B runs).  The wait states behind it: a VALU write of the store's LAST data register in the next cycle
// corrupts that dword on gfx950 (mifwt_dwt2_inv_pyr.hip, ipyr_store4)
#ifndef MIFWT_PYR_ST16_POLICY
#define MIFWT_PYR_ST16_POLICY "nt"
#endif
__device__ __forceinline__ void pyr_store4(const f4 data, rsrc_t rsrc, uint32_t voff, uint32_t soff) {
  asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen " MIFWT_PYR_ST16_POLICY "\n\ts_nop 1" ::"v"(data), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}
// Lane-pair exchange in front of a 16-byte store (v_permlane32_swap, new on gfx950: lanes 32-63 of the first operand change places with
// lanes 0-31 of the second).  Lane l < 32 and lane l + 32 hold NEIGHBOURING column pairs (l: columns c, c + 1; l + 32: c + 2, c + 3) of
// the same two rows: a = row i, b = row i + 1.  Afterwards lane l holds columns c .. c + 3 of row i, lane l + 32 columns c .. c + 3 of
// row i + 1: one buffer_store_dwordx4 per lane writes 512 contiguous bytes of each of the two rows.
// (inline assembly: hipcc 7.2 folds the two results of __builtin_amdgcn_permlane32_swap into one register inside this kernel — the
// store data came out as (x, y, x, y).  Two wait states in front: a VALU write of either operand must be two instructions old.)
__device__ __forceinline__ void pyr_permlane32_swap(float& x, float& y) {
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 0" : "+v"(x), "+v"(y));
}
__device__ __forceinline__ f4 pyr_swap_rows(float a0, float a1, float b0, float b1) {
  pyr_permlane32_swap(a0, b0);
  pyr_permlane32_swap(a1, b1);
  return (f4){a0, a1, b0, b1};
}
// Per-lane constants of the 16-byte store path of one output plane family (one row pitch): the lane pair (l, l + 32) owns columns
// cq .. cq + 3 of the two rows a half step completes.  Full groups leave as one 16-byte store per lane, a ragged last group (1 .. 3
// columns inside the plane) as single dwords; rows a segment does not own go to an offset beyond every resource.
struct PyrSt16 {
  uint32_t v16;  // byte offset of the lane's four columns if all of them lie inside the plane, else beyond every resource
  int cq0, wend;  // (uniform) first column of the wave's lane grid, end of the plane's columns
  bool rag;      // (uniform) some lane of the wave holds a ragged group
  // (everything else is recomputed where it is needed: the level-1 / level-2 waves of the 8-tap kernel have no registers to spare)
  __device__ __forceinline__ void set(int lane, int cq0_, int wend_) {
    cq0 = cq0_;
    wend = wend_;
    const int cq = cq0 + 4 * (lane & 31), nin = wend - cq;
    v16 = nin >= 4 ? 4u * (uint32_t)cq : kPyrOob;
    rag = __builtin_amdgcn_readfirstlane(__builtin_amdgcn_ballot_w64(nin > 0 && nin < 4) != 0);
  }
  // rows i0 (lanes 0-31) and i0 + 1 (lanes 32-63) of a plane whose segment owns rows [oA, oB): the lane's row is `dx` bytes
  // behind the scalar row offset `so` (`mine`: the segment owns the lane's row)
  __device__ __forceinline__ void rows(int lane, int i0, int oA, int oB, uint32_t pitch_bytes, uint32_t& v4, uint32_t& dx, bool& mine, uint32_t& so) const {
    const bool own0 = i0 >= oA && i0 < oB, own1 = i0 + 1 >= oA && i0 + 1 < oB, upper = lane >= 32;
    so = (own0 || own1) ? (uint32_t)(own0 ? i0 : i0 + 1) * pitch_bytes : 0u;
    mine = upper ? own1 : own0;
    dx = (upper && own0) ? pitch_bytes : 0u;
    v4 = mine ? v16 + dx : kPyrOob;  // (kPyrOob + a pitch is still beyond every resource)
  }
  // one band: a = the lane's two columns of row i0, b = of row i0 + 1
  __device__ __forceinline__ void band(int lane, uint32_t v4, uint32_t dx, bool mine, float a0, float a1, float b0, float b1, rsrc_t r, uint32_t soff) const {
    const f4 t = pyr_swap_rows(a0, a1, b0, b1);
    pyr_store4(t, r, v4, soff);
    if (rag) {
      const int cq = cq0 + 4 * (lane & 31), nin = wend - cq;
      const uint32_t vrl = (mine && nin > 0 && nin < 4) ? 4u * (uint32_t)cq + dx : kPyrOob;
      pyr_store1(t.x, r, vrl, soff);
      pyr_store1(t.y, r, nin >= 2 ? vrl : kPyrOob, soff + 4u);
      pyr_store1(t.z, r, nin >= 3 ? vrl : kPyrOob, soff + 8u);
    }
  }
};

// cache policy of the streamed input requests (build-time switch for A/B runs: tools/pyr_ab.py)
#ifndef MIFWT_PYR_DMA_POLICY
#define MIFWT_PYR_DMA_POLICY "nt"
#endif
// (LDS-DMA) a loader's share of one staged row: NCH requests of 1 KiB, LDS addresses lds0 + 2048 j (the two loaders take the even / the
// odd 1-KiB pieces of a row), global offsets voff[j]
template <int NCH, int STR>
__device__ __forceinline__ void pyr_dma_row(const uint32_t (&voff)[3], rsrc_t rsrc, uint32_t soff, uint32_t lds0) {
  // (STR = LDS distance of a loader's consecutive pieces: 2 KiB with two loader waves — they take the even / the odd pieces — 1 KiB with one)
  uint32_t keep;
  if constexpr (NCH == 1) {
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen " MIFWT_PYR_DMA_POLICY " lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff[0]), "s"(rsrc), "s"(soff), "s"(lds0) : "memory");
  } else if constexpr (NCH == 2) {
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %5\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, %4 offen " MIFWT_PYR_DMA_POLICY " lds\n\t"
                 "s_add_u32 m0, m0, %6\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen " MIFWT_PYR_DMA_POLICY " lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff[0]), "v"(voff[1]), "s"(rsrc), "s"(soff), "s"(lds0), "n"(STR) : "memory", "scc");
  } else {
    static_assert(NCH == 3, "at most three requests per row and loader");
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %6\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %4, %5 offen " MIFWT_PYR_DMA_POLICY " lds\n\t"
                 "s_add_u32 m0, m0, %7\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %4, %5 offen " MIFWT_PYR_DMA_POLICY " lds\n\t"
                 "s_add_u32 m0, m0, %7\n\ts_nop 0\n\tbuffer_load_dwordx4 %3, %4, %5 offen " MIFWT_PYR_DMA_POLICY " lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff[0]), "v"(voff[1]), "v"(voff[2]), "s"(rsrc), "s"(soff), "s"(lds0), "n"(STR) : "memory", "scc");
  }
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
// (one definition for every kernel: mifwt_stream.h)
__device__ __forceinline__ void vfma_lo(f2& acc, const f2 tap, const f2 pair) { pkfma_lo(acc, tap, pair); }
__device__ __forceinline__ void vfma_hi(f2& acc, const f2 tap, const f2 pair) { pkfma_hi(acc, tap, pair); }
__device__ __forceinline__ f2 vmul_lo(const f2 tap, const f2 pair) { return pkmul_lo(tap, pair); }
__device__ __forceinline__ f2 vmul_hi(const f2 tap, const f2 pair) { return pkmul_hi(tap, pair); }

// rolling vertical pass: the L/2 outputs in flight of NC columns; lo = (aa, da), hi = (ad, dd) per column.  Output i lives in
// slot i mod L/2 for its whole life, so nothing is ever copied: the pair index modulo L/2 (R) is a compile-time constant at
// every call site (the callers unroll or switch over it)
// (f64 callers: the same roles with two v_fma_f64 per step, mifwt_stream.h)
__device__ __forceinline__ void vfma_lo(d2& acc, const d2 tap, const d2 pair) { afma_lo(acc, tap, pair); }
__device__ __forceinline__ void vfma_hi(d2& acc, const d2 tap, const d2 pair) { afma_hi(acc, tap, pair); }
__device__ __forceinline__ d2 vmul_lo(const d2 tap, const d2 pair) { return amul_lo(tap, pair); }
__device__ __forceinline__ d2 vmul_hi(const d2 tap, const d2 pair) { return amul_hi(tap, pair); }

template <int L, int NC, typename V = f2>
struct PyrAcc {
  static constexpr int HP = L / 2;
  V lo[HP][NC], hi[HP][NC];
  __device__ __forceinline__ void clear() {
#pragma unroll
    for (int q = 0; q < HP; ++q)
#pragma unroll
      for (int c = 0; c < NC; ++c) lo[q][c] = hi[q][c] = V{};
  }
  // one row of horizontally filtered samples hv[c] = (h_lo, h_hi) of pair p (R = p mod HP); PH = 0: first row of the pair, 1: second
  template <int PH, int R>
  __device__ __forceinline__ void feed(const V (&tap)[L], const V (&hv)[NC]) {
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
  } else if constexpr (N == 4) {
    if (r < 2) {
      if (r == 0) f(std::integral_constant<int, 0>{});
      else f(std::integral_constant<int, 1>{});
    } else {
      if (r == 2) f(std::integral_constant<int, 2>{});
      else f(std::integral_constant<int, 3>{});
    }
  } else {
    static_assert(N == 5, "filter lengths up to 10");
    if (r < 2) {
      if (r == 0) f(std::integral_constant<int, 0>{});
      else f(std::integral_constant<int, 1>{});
    } else if (r == 2) {
      f(std::integral_constant<int, 2>{});
    } else {
      if (r == 3) f(std::integral_constant<int, 3>{});
      else f(std::integral_constant<int, 4>{});
    }
  }
}

}  // namespace mifwt
