// mifwt_dwt2_fwd_pyr.hip — UP TO THREE consecutive 2-D analysis levels in one launch (gfx950), kernel id 16.
//
// Seam: NLEV trips of the reference's level loop (src/ptwt/conv_transform_2.py:142-149: _fwt_pad2 + F.conv2d(stride 2) +
// split); a pyramid returns only the detail bands of every level but the last (conv_transform_2.py:150-156), so the
// approximations in between never reach HBM here: they live in LDS rings.
//
// Shape of the work (measured first with tools/ubench.hip, profiles/r02_ubench.txt):
//   * a workgroup = kPyrNW COMPUTE waves + one LOADER wave, and owns one row segment of one image;
//   * compute wave w = one column STRIP: 256 level-0 columns -> <= 128 level-1 columns (two per lane) -> <= 64 level-2
//     columns (one per lane) -> <= 32 level-3 columns.  A strip computes its own left halo at every level (the lanes
//     exist anyway), so strips never exchange data and there is no barrier between levels;
//   * rows STREAM through the strip: the vertical pass of every level keeps the L/2 outputs in flight in registers
//     (rolling accumulators, 2 packed FMAs per sample and band pair), nothing is re-read; level l+1 consumes the rows of
//     level l from a 16-row LDS ring through the boundary index map (mirrored rows at the top / bottom of the plane are
//     ring rows), lagging by a fixed number of 8-row steps;
//   * the LOADER wave issues every global load of the workgroup as LDS-DMA (buffer_load_dwordx4 ... lds, 1 KiB per
//     instruction, NON-TEMPORAL so that the streamed input does not evict the half-written output lines from L2 — 107 ->
//     85 us on the traffic skeleton) three 4-row sub-steps ahead; its vmcnt queue holds loads only.  The compute waves'
//     queue holds stores only and is never waited on: with both in one queue the in-order counter made every load wait
//     for the acknowledgement of older stores (150 us for the same traffic).  One s_barrier per 4-row sub-step hands a
//     landed sub-buffer over;
//     (Measured and dropped: progress counters in LDS instead of the barriers — waves that spin on a counter steal issue
//     slots from the waves they wait for: 145-170 us against 111-119 us with barriers on config 2.)
//   * boundary extension: level-0 pad columns are copied inside LDS after the rows land, ring pad columns after a ring
//     row is written (edge strips only, one read + one write per step); out-of-plane rows in zero mode are zero rows.
// Results agree with the per-level kernels to rounding (different summation order), with the fp64 oracle within 1e-6.
// f32, even L <= 8, modes zero / constant / reflect / symmetric (periodic needs the far side of the plane).
// Algorithmic traffic: 4 B H W read + 4 B (3 H1 W1 [+ 3 H2 W2] + 4 H_N W_N) written.
#include <type_traits>

#include "mifwt_stream.h"

namespace mifwt {

extern unsigned long long* g_pyr_prof;

constexpr int kPyrNW = 5;                     // compute waves = column strips per workgroup
constexpr int kPyrSub = 4;                    // level-0 rows per sub-step (one barrier each)
constexpr int kPyrNBuf = 4;                   // staging sub-buffers per strip
constexpr int kPyrPad = 8;                    // floats in front of a staged / ring row (left extension, 16-byte aligned body)
constexpr int kPyrSlotB = (kPyrPad + 256 + 8) * 4;   // one staged level-0 row: pad + 256 columns + right extension
constexpr int kPyrR1B = (kPyrPad + 128 + 8) * 4;     // one ring row of level-1 approximations
constexpr int kPyrR2B = (kPyrPad + 64 + 8) * 4;      // ... of level-2 approximations
constexpr int kPyrRing = 16;                  // ring rows (+ one zero row at slot 16)
constexpr int kPyrStageB = kPyrSub * kPyrNBuf * kPyrSlotB;
constexpr uint32_t kPyrOob = 0x80000000u;

constexpr int pyr_wave_bytes(int nlev) { return kPyrStageB + (nlev >= 2 ? (kPyrRing + 1) * kPyrR1B : 0) + (nlev >= 3 ? (kPyrRing + 1) * kPyrR2B : 0); }
constexpr int pyr_lds_bytes(int nlev) { return 64 + kPyrNW * pyr_wave_bytes(nlev); }
// steps by which level l + 1 lags level l (see the derivation in DESIGN.md §4.1c): the rows its first / mirrored taps
// need must have been produced
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

template <int L, int NLEV>
struct PyrArgs {
  const float* x;
  float* det[NLEV][3];  // [level - 1][band ad, da, dd]
  float* approx;        // band aa of level NLEV
  int64_t xs_b, ds_b[NLEV], as_b;  // image strides (elements)
  int xs_h, ds_h[NLEV], as_h;      // row strides (elements)
  int H[NLEV + 1], W[NLEV + 1];    // extents of level 0 (the input) .. NLEV
  int nstrips, ngroups, nseg, seg_rows;  // strips per plane, workgroups per row segment, segments, level-NLEV rows per segment
  int cpw0, cpw;                          // level-NLEV columns of strip 0 / of the other strips
  int mode;
  unsigned long long* prof;  // optional per-wave cycle counts [workgroup][wave][total, in barriers] (mifwt_pyr_profile_buffer)
  int dbg;  // A/B measurement switches (MIFWT_OPT_DEBUG): 1 = no stores, 2 = no loads, 4 = deep waves idle, 16 = loader at default priority
  f2 tap[L];  // (dec_lo[m], dec_hi[m])
};

typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t pyr_rsrc(const void* p, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
// 64 lanes x 16 B -> LDS [lds_addr + 16 lane); global address = resource base + voff (per lane) + soff; non-temporal
// (M0 carries the LDS address and belongs to the compiler: saved and restored inside the statement.)  One row of the
// workgroup: the same level-0 row for the five strips, LDS addresses lds0 + w * step
__device__ __forceinline__ void pyr_dma_row(const uint32_t (&voff)[kPyrNW], rsrc_t rsrc, uint32_t soff, uint32_t lds0, uint32_t step) {
  uint32_t keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %8\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %6, %7 offen nt lds\n\t"
      "s_add_u32 m0, m0, %9\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %6, %7 offen nt lds\n\t"
      "s_add_u32 m0, m0, %9\n\ts_nop 0\n\tbuffer_load_dwordx4 %3, %6, %7 offen nt lds\n\t"
      "s_add_u32 m0, m0, %9\n\ts_nop 0\n\tbuffer_load_dwordx4 %4, %6, %7 offen nt lds\n\t"
      "s_add_u32 m0, m0, %9\n\ts_nop 0\n\tbuffer_load_dwordx4 %5, %6, %7 offen nt lds\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff[0]), "v"(voff[1]), "v"(voff[2]), "v"(voff[3]), "v"(voff[4]), "s"(rsrc), "s"(soff), "s"(lds0), "s"(step)
      : "memory", "scc");
}
__device__ __forceinline__ void pyr_store1(float v, rsrc_t rsrc, uint32_t voff, uint32_t soff) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v), rsrc, voff, soff, 0);
}
// workgroup barrier; with profiling on, the cycles spent in it are added to `waited`
__device__ __forceinline__ void pyr_barrier(const unsigned long long* prof, unsigned long long& waited) {
  if (prof) {
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
// in VGPR pairs (tools/ubench.hip "distinct" rows, profiles/r02_ubench.txt)
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

// Roles of the waves of a workgroup: one wave per (strip, level) + the loader.
//   level-1 wave of strip k   staged rows -> level-1 details (HBM) + approximation rows (ring 1 of the strip)
//   level-2 wave              ring 1 -> level-2 details + ring 2          level-3 wave   ring 2 -> the four level-3 bands
// A ring row is written in step s and read from a later step on, with at least one barrier in between.
// A workgroup's waves land on the four SIMDs round-robin (wave i -> class i mod 4); a level-1 wave issues about 2.3x the
// instructions of a level-2 wave and 5x those of a level-3 wave, and a SIMD with only two long waves on it cannot hide
// their latencies (measured: the SIMD hosting two level-1 waves and nothing else set the pace of the workgroup while the
// others idled a third of the time).  The tables put on the four classes:
//   three levels (16 waves):  {L1, L1, L3, L3}  {L1, L2, L2, L3}  {L1, L2, L2, L3}  {L1, L2, L3, loader}
//   two levels   (11 waves):  {L1, L1, L2}      {L1, L2, L2}      {L1, L2, L2}      {L1, loader}
enum PyrRole { kRoleL1 = 0, kRoleL2 = 1, kRoleL3 = 2, kRoleLoad = 3 };
constexpr int pyr_nwaves(int nlev) { return nlev == 1 ? 6 : (nlev == 2 ? 11 : 16); }
constexpr int pyr_role(int nlev, int w) {
  if (w < 5) return kRoleL1;
  if (nlev == 1) return kRoleLoad;
  if (nlev == 2) return w == 7 ? kRoleLoad : kRoleL2;
  return w == 15 ? kRoleLoad : ((w == 5 || w == 6 || w == 7 || w == 9 || w == 10) ? kRoleL2 : kRoleL3);
}
constexpr int pyr_strip_of(int nlev, int w) {
  if (w < 5) return w;
  if (nlev == 2) return w < 7 ? w - 5 : w - 6;                     // 5 6 . 8 9 10 -> 0 1 . 2 3 4
  if (w == 5 || w == 6 || w == 7) return w - 5;                     // L2: 5 6 7 9 10 -> 0 1 2 3 4
  if (w == 9 || w == 10) return w - 6;
  return w == 8 ? 0 : w - 10;                                       // L3: 8 11 12 13 14 -> 0 1 2 3 4
}
template <int L, int NLEV>
__global__ void __launch_bounds__(64 * pyr_nwaves(NLEV)) dwt2_fwd_pyr_kernel(const PyrArgs<L, NLEV> a) {
  constexpr int HL = L - 2, HP = L / 2;
  constexpr int NW = kPyrNW, NWAVES = pyr_nwaves(NLEV);
  static_assert(kPyrNW == 5, "the wave -> role tables are written for five strips");
  constexpr int WB = pyr_wave_bytes(NLEV);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int bid = blockIdx.x;
  const int grp = bid % a.ngroups;
  bid /= a.ngroups;
  const int seg = bid % a.nseg, img = bid / a.nseg;
  const bool zero_mode = a.mode == MIFWT_MODE_ZERO;
  Fold1 fold;
  fold.set(a.mode);

  // ---- row ranges of this segment: computed rows [rA, rB) and owned rows [oA, oB) per level (index = level) -------------
  int rA[NLEV + 1], rB[NLEV + 1], oA[NLEV + 1], oB[NLEV + 1];
  oA[NLEV] = rA[NLEV] = seg * a.seg_rows;
  oB[NLEV] = rB[NLEV] = seg == a.nseg - 1 ? a.H[NLEV] : min(a.H[NLEV], rA[NLEV] + a.seg_rows);
#pragma unroll
  for (int l = NLEV - 1; l >= 1; --l) {
    oA[l] = 2 * oA[l + 1];
    oB[l] = oB[l + 1] == a.H[l + 1] ? a.H[l] : min(a.H[l], 2 * oB[l + 1]);
    rA[l] = max(0, 2 * rA[l + 1] - HL);
    rB[l] = min(a.H[l], 2 * rB[l + 1]);
  }
  // lags (in steps) of levels 2 and 3 behind level 1: a segment at the top of the plane waits for the rows its mirrored
  // taps need, the others only for the filter delay; level 2 never reads a ring row in the step that writes it
  const bool top = seg == 0;
  const int D2 = top ? pyr_lag2(L) : pyr_lag2_inner(L);
  const int D3 = top ? pyr_lag3(L) : pyr_lag3_inner(L);
  const int E0 = 2 * rA[1] - HL;                         // first level-0 row of the stream (extended index)
  const int e0_end = 2 * rB[1];                          // level-0 rows from here on feed nothing
  const int npair1 = rB[1] - rA[1] + HP - 1;             // row pairs level 1 must see
  const int nsteps1 = (npair1 + 3) / 4;
  int nsteps = nsteps1;
  int npair2 = 0, npair3 = 0;
  if constexpr (NLEV >= 2) {
    npair2 = rB[2] - rA[2] + HP - 1;
    nsteps = max(nsteps, D2 + (npair2 + 1) / 2);
  }
  if constexpr (NLEV >= 3) {
    npair3 = rB[3] - rA[3] + HP - 1;
    nsteps = max(nsteps, D3 + npair3);
  }
  const int nsub = 2 * nsteps, nsub1 = 2 * nsteps1;

  // =====================================================================================================================
  // loader wave
  if (wave == (NLEV == 1 ? 5 : (NLEV == 2 ? 7 : 15))) {  // = the wave the role tables give kRoleLoad
    const uint32_t img_bytes = ((uint32_t)(a.H[0] - 1) * (uint32_t)a.xs_h + (uint32_t)a.W[0]) * 4u;
    const rsrc_t xr = pyr_rsrc(a.x + (int64_t)img * a.xs_b, img_bytes);
    const rsrc_t xr_dead = pyr_rsrc(a.x + (int64_t)img * a.xs_b, 0);  // every lane out of range: a row of zeros lands
    const uint32_t row_bytes = (uint32_t)a.xs_h * 4u;
    uint32_t voff[NW];
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      const int k = grp * NW + w;
      // level-1 columns the strip computes start at cA1; its staged row starts at level-0 column g0 (16-byte aligned)
      int cA = k == 0 ? 0 : a.cpw0 + (k - 1) * a.cpw;
#pragma unroll
      for (int l = NLEV - 1; l >= 1; --l) cA = max(0, 2 * cA - HL);
      const int g0 = max(0, 2 * cA - HL) & ~3;
      const int c = g0 + 4 * lane;
      voff[w] = (k < a.nstrips && c < a.W[0]) ? 4u * (uint32_t)c : kPyrOob;
    }
    auto issue = [&](int t) {
      if (a.dbg & 2) return;
      const uint32_t buf = (uint32_t)(t & (kPyrNBuf - 1)) * (kPyrSub * kPyrSlotB) + 64u + kPyrPad * 4u;
#pragma unroll
      for (int kk = 0; kk < kPyrSub; ++kk) {
        const int e = E0 + kPyrSub * t + kk;
        const bool dead = e >= e0_end || (zero_mode && (unsigned)e >= (unsigned)a.H[0]);
        const uint32_t soff = dead ? 0u : (uint32_t)fold(e, a.H[0]) * row_bytes;
        pyr_dma_row(voff, dead ? xr_dead : xr, soff, buf + (uint32_t)kk * kPyrSlotB, (uint32_t)WB);
      }
    };
    constexpr int PER = kPyrSub * NW;  // DMA instructions per sub-step
    // the loader is the youngest wave on its SIMD and instruction issue goes by priority, then age: without this the older
    // waves beside it delay the one wave every other wave waits for (measured: 115.6 -> 111.4 us on config 2)
    if (!(a.dbg & 16)) __builtin_amdgcn_s_setprio(3);
    __syncthreads();  // the other waves have initialised their LDS
#pragma unroll
    for (int t = 0; t < kPyrNBuf - 1; ++t)
      if (t < nsub1) issue(t);
#pragma unroll 1
    for (int t = 0; t < nsub; ++t) {
      // sub-step t must have landed; t + 1 and t + 2 may still be in flight
      if (t + 2 < nsub1) pyr_wait_vm<2 * PER>();
      else if (t + 1 < nsub1) pyr_wait_vm<PER>();
      else pyr_wait_vm<0>();
      __syncthreads();
      if (t + kPyrNBuf - 1 < nsub1) issue(t + kPyrNBuf - 1);  // into the buffer sub-step t - 1 was read from
    }
    return;
  }

  // =====================================================================================================================
  int role = kRoleL1, slot_w = 0;
#pragma unroll
  for (int w = 0; w < NWAVES; ++w)
    if (wave == w) {
      role = pyr_role(NLEV, w);
      slot_w = pyr_strip_of(NLEV, w);
    }
  const bool deep = role != kRoleL1;
  const int strip = grp * NW + slot_w;
  if (strip >= a.nstrips) {
#pragma unroll 1
    for (int t = 0; t <= nsub; ++t) __syncthreads();
    return;
  }
  // ---- column ranges: computed [cA, cB), owned [pA, pB) per level -------------------------------------------------------
  int cA[NLEV + 1], cB[NLEV + 1], pA[NLEV + 1], pB[NLEV + 1];
  pA[NLEV] = cA[NLEV] = strip == 0 ? 0 : a.cpw0 + (strip - 1) * a.cpw;
  pB[NLEV] = cB[NLEV] = strip == a.nstrips - 1 ? a.W[NLEV] : min(a.W[NLEV], a.cpw0 + strip * a.cpw);
#pragma unroll
  for (int l = NLEV - 1; l >= 1; --l) {
    pA[l] = 2 * pA[l + 1];
    pB[l] = pB[l + 1] == a.W[l + 1] ? a.W[l] : min(a.W[l], 2 * pB[l + 1]);
    cA[l] = max(0, 2 * cA[l + 1] - HL);
    cB[l] = min(a.W[l], 2 * cB[l + 1]);
  }
  unsigned long long waited = 0;
  const unsigned long long t_start = a.prof ? __builtin_readcyclecounter() : 0;
  auto prof_out = [&]() {
    if (a.prof && lane == 0) {
      unsigned long long* o = a.prof + ((size_t)blockIdx.x * NWAVES + wave) * 2;
      o[0] = __builtin_readcyclecounter() - t_start;
      o[1] = waited;
    }
  };
  unsigned char* const wbase = smem + 64 + slot_w * WB;
  unsigned char* const stage = wbase;
  unsigned char* const ring1 = wbase + kPyrStageB;
  unsigned char* const ring2 = ring1 + (kPyrRing + 1) * kPyrR1B;
  f2 tap[L];
#pragma unroll
  for (int m = 0; m < L; ++m) tap[m] = a.tap[m];

  // =====================================================================================================================
  // level-1 wave
  if (!deep) {
    const int g0 = max(0, 2 * cA[1] - HL) & ~3;
    // lane -> columns c1, c1 + 1; its window = staged floats [win1, win1 + L + 2)
    const int c1 = cA[1] + 2 * lane;
    const uint32_t win1 = 4u * (uint32_t)(kPyrPad - HL + 2 * cA[1] - g0 + 4 * lane);
    const uint32_t sv1_x2 = (c1 >= pA[1] && c1 + 1 < pB[1]) ? 4u * (uint32_t)c1 : kPyrOob;
    const uint32_t sv1_x1 = (c1 >= pA[1] && c1 + 1 == pB[1]) ? 4u * (uint32_t)c1 : kPyrOob;
    const bool ragged1 = ((pB[1] - cA[1]) & 1) != 0;
    constexpr int NP = 2 * HL + 1;  // extension columns of a row: HL on the left, HL + 1 on the right
    // level-0 pad fill (strips at the plane's left / right edge): lane -> (row kk of the sub-step, pad column)
    uint32_t f0_src = 0, f0_dst = 0;
    bool f0_on = false;
    {
      const int kk = lane / NP, p = lane - kk * NP;
      const bool left = p < HL;
      const int e = left ? p - HL : a.W[0] + (p - HL);  // extended level-0 column
      if (kk < kPyrSub && !zero_mode && (left ? cA[1] == 0 : cB[1] == a.W[1])) {
        f0_on = true;
        f0_src = (uint32_t)kk * kPyrSlotB + 4u * (uint32_t)(kPyrPad + fold(e, a.W[0]) - g0);
        f0_dst = (uint32_t)kk * kPyrSlotB + 4u * (uint32_t)(kPyrPad + e - g0);
      }
    }
    const bool f0_any = __builtin_amdgcn_readfirstlane(__builtin_amdgcn_ballot_w64(f0_on) != 0);
    // ring-1 pad fill: lane -> (row j of the sub-step's two, pad column); idle lanes copy float 0 of the pair's first row
    // onto itself (never used: the pad area starts at kPyrPad - HL >= 2)
    uint32_t f1_src = 0, f1_dst = 0;
    bool f1_any = false;
    if constexpr (NLEV >= 2) {
      const int j = lane / NP, p = lane - j * NP;
      const bool left = p < HL;
      const int e = left ? p - HL : a.W[1] + (p - HL);
      const bool on = j < 2 && !zero_mode && (left ? cA[1] == 0 : cB[1] == a.W[1]);
      f1_src = on ? (uint32_t)j * kPyrR1B + 4u * (uint32_t)(kPyrPad + fold(e, a.W[1]) - cA[1]) : 0u;
      f1_dst = on ? (uint32_t)j * kPyrR1B + 4u * (uint32_t)(kPyrPad + e - cA[1]) : 0u;
      f1_any = __builtin_amdgcn_readfirstlane(__builtin_amdgcn_ballot_w64(on) != 0);
    }
    // stores are never branched around (a branch per row pair would cut the step into small scheduling regions and the
    // LDS latency of every row would be exposed): rows this segment does not own go through a resource of size 0, which
    // drops them
    const uint32_t dbytes = (a.dbg & 1) ? 0u : ((uint32_t)(a.H[1] - 1) * (uint32_t)a.ds_h[0] + (uint32_t)a.W[1]) * 4u;
    const uint32_t abytes = (a.dbg & 1) ? 0u : ((uint32_t)(a.H[NLEV] - 1) * (uint32_t)a.as_h + (uint32_t)a.W[NLEV]) * 4u;
    const float* const dp0 = a.det[0][0] + (int64_t)img * a.ds_b[0];
    const float* const dp1 = a.det[0][1] + (int64_t)img * a.ds_b[0];
    const float* const dp2 = a.det[0][2] + (int64_t)img * a.ds_b[0];
    const float* const app = a.approx + (int64_t)img * a.as_b;

    // the staged rows' pad columns must read as zero in zero mode (nothing ever writes them)
    for (int i = lane; i < kPyrStageB / 16; i += 64) reinterpret_cast<f4*>(stage)[i] = (f4){0.f, 0.f, 0.f, 0.f};
    __syncthreads();  // ... before the loader's first row lands

    PyrAcc<L, 2> acc;
    acc.clear();
    // horizontal pass of one row: staged row -> (lo, hi) of the lane's two columns
    // the L + 2 staged samples under a lane's two columns
    auto load_win = [&](const unsigned char* row, f2 (&w)[HP + 1]) {
      if constexpr ((HL & 3) == 2) {
        w[0] = *reinterpret_cast<const f2*>(row);
#pragma unroll
        for (int j = 0; j < HP / 2; ++j) {
          const f4 v = *reinterpret_cast<const f4*>(row + 8 + 16 * j);
          w[1 + 2 * j] = (f2){v.x, v.y};
          w[2 + 2 * j] = (f2){v.z, v.w};
        }
      } else {
#pragma unroll
        for (int j = 0; j < (HP + 1) / 2; ++j) {
          const f4 v = *reinterpret_cast<const f4*>(row + 16 * j);
          w[2 * j] = (f2){v.x, v.y};
          w[2 * j + 1] = (f2){v.z, v.w};
        }
      }
    };
    // horizontal pass of the two rows of a pair, interleaved: four independent chains of packed FMAs (one wave per SIMD has
    // nobody to hide the dependent-issue latency of a single chain behind)
    auto h_pair = [&](const f2 (&wa)[HP + 1], const f2 (&wb)[HP + 1], f2 (&ha)[2], f2 (&hb)[2]) {
#pragma unroll
      for (int k = 0; k < HP; ++k) {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          if (k == 0) {
            ha[c] = vmul_lo(tap[L - 1], wa[c]);
            hb[c] = vmul_lo(tap[L - 1], wb[c]);
          } else {
            vfma_lo(ha[c], tap[L - 1 - 2 * k], wa[c + k]);
            vfma_lo(hb[c], tap[L - 1 - 2 * k], wb[c + k]);
          }
        }
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          vfma_hi(ha[c], tap[L - 2 - 2 * k], wa[c + k]);
          vfma_hi(hb[c], tap[L - 2 - 2 * k], wb[c + k]);
        }
      }
    };

    // one 8-row step; SM = the step index modulo 3 when L/2 = 3 (four pairs per step rotate three accumulator slots), else 0;
    // RAG: this strip's last owned column is the first of a lane's two (one more 4-byte store per band)
    auto step1 = [&](auto sm_tag, auto rag_tag, int s) {
      constexpr int SM = decltype(sm_tag)::value;
      constexpr bool RAG = decltype(rag_tag)::value;
      pyr_static_for<2>([&](auto half_tag) {
        constexpr int half = decltype(half_tag)::value;
        pyr_barrier(a.prof, waited);  // the loader has seen this sub-step land
        if (s < nsteps1) {
          const int t = 2 * s + half;
          unsigned char* sb = stage + (t & (kPyrNBuf - 1)) * (kPyrSub * kPyrSlotB);
          if (f0_any) {
            const float v = *reinterpret_cast<const float*>(sb + f0_src);
            wave_lds_fence();
            if (f0_on) *reinterpret_cast<float*>(sb + f0_dst) = v;
            wave_lds_fence();
          }
          // every window of the sub-step is requested before the first one is used
          f2 w[kPyrSub][HP + 1];
#pragma unroll
          for (int kk = 0; kk < kPyrSub; ++kk) load_win(sb + kk * kPyrSlotB + win1, w[kk]);
          pyr_static_for<kPyrSub / 2>([&](auto jj_tag) {
            constexpr int jj = decltype(jj_tag)::value;
            constexpr int j = 2 * half + jj;      // pair of the step
            constexpr int R = (4 * SM + j) % HP;  // its index modulo L/2
            f2 ha[2], hb[2];
            h_pair(w[2 * jj], w[2 * jj + 1], ha, hb);
            acc.template feed<0, R>(tap, ha);
            acc.template feed<1, R>(tap, hb);
            {
              const int i = rA[1] + 4 * s + j - (HP - 1);  // the level-1 row it completes
              const f2 (&lo)[2] = acc.lo[PyrAcc<L, 2>::done(R)];
              const f2 (&hi)[2] = acc.hi[PyrAcc<L, 2>::done(R)];
              if constexpr (NLEV >= 2) {
                *reinterpret_cast<f2*>(ring1 + ((4 * s + j) & (kPyrRing - 1)) * kPyrR1B + 4 * (kPyrPad + 2 * lane)) = (f2){lo[0].x, lo[1].x};
              }
              {
                const bool own = i >= oA[1] && i < oB[1];
                const uint32_t nb = own ? dbytes : 0u;
                const rsrc_t r0 = pyr_rsrc(dp0, nb), r1 = pyr_rsrc(dp1, nb), r2 = pyr_rsrc(dp2, nb);
                const uint32_t so = own ? (uint32_t)i * (uint32_t)a.ds_h[0] * 4u : 0u;
                __builtin_amdgcn_raw_buffer_store_b64((f2){hi[0].x, hi[1].x}, r0, sv1_x2, so, 0);
                __builtin_amdgcn_raw_buffer_store_b64((f2){lo[0].y, lo[1].y}, r1, sv1_x2, so, 0);
                __builtin_amdgcn_raw_buffer_store_b64((f2){hi[0].y, hi[1].y}, r2, sv1_x2, so, 0);
                if constexpr (RAG) {
                  pyr_store1(hi[0].x, r0, sv1_x1, so);
                  pyr_store1(lo[0].y, r1, sv1_x1, so);
                  pyr_store1(hi[0].y, r2, sv1_x1, so);
                }
                if constexpr (NLEV == 1) {
                  const rsrc_t ra = pyr_rsrc(app, own ? abytes : 0u);
                  const uint32_t sa = own ? (uint32_t)i * (uint32_t)a.as_h * 4u : 0u;
                  __builtin_amdgcn_raw_buffer_store_b64((f2){lo[0].x, lo[1].x}, ra, sv1_x2, sa, 0);
                  if constexpr (RAG) pyr_store1(lo[0].x, ra, sv1_x1, sa);
                }
              }
            }
          });
          if constexpr (NLEV >= 2) {
            if (f1_any) {  // extension columns of the two ring rows just written
              wave_lds_fence();
              unsigned char* rb = ring1 + ((4 * s + 2 * half) & (kPyrRing - 1)) * kPyrR1B;
              const float v = *reinterpret_cast<const float*>(rb + f1_src);
              wave_lds_fence();
              *reinterpret_cast<float*>(rb + f1_dst) = v;
            }
          }
        }
      });
    };
    auto run1 = [&](auto rag_tag) {
      int sm = 0;
#pragma unroll 1
      for (int s = 0; s < nsteps; ++s) {
        if constexpr (HP == 3) {
          pyr_dispatch<3>(sm, [&](auto t) { step1(t, rag_tag, s); });
          sm = sm == 2 ? 0 : sm + 1;
        } else {
          step1(std::integral_constant<int, 0>{}, rag_tag, s);
        }
      }
    };
    if (ragged1) run1(std::true_type{});
    else run1(std::false_type{});
    prof_out();
    return;
  }

  // =====================================================================================================================
  // deep wave: levels 2 (and 3) of its strip
  if constexpr (NLEV >= 2) {
    constexpr int NP = 2 * HL + 1;
    const int c2 = cA[2] + lane;
    const uint32_t win2 = 4u * (uint32_t)(kPyrPad - HL + 2 * cA[2] - cA[1] + 2 * lane);
    const uint32_t sv2 = (c2 >= pA[2] && c2 < pB[2]) ? 4u * (uint32_t)c2 : kPyrOob;
    uint32_t win3 = 0, sv3 = kPyrOob, f2_src = 0, f2_dst = 0;
    bool f2_any = false;
    if constexpr (NLEV >= 3) {
      const int c3 = cA[3] + lane;
      win3 = 4u * (uint32_t)(kPyrPad - HL + 2 * cA[3] - cA[2] + 2 * lane);
      sv3 = (c3 >= pA[3] && c3 < pB[3]) ? 4u * (uint32_t)c3 : kPyrOob;
      const int j = lane / NP, p = lane - j * NP;
      const bool left = p < HL;
      const int e = left ? p - HL : a.W[2] + (p - HL);
      const bool on = j < 2 && !zero_mode && (left ? cA[2] == 0 : cB[2] == a.W[2]);
      f2_src = on ? (uint32_t)j * kPyrR2B + 4u * (uint32_t)(kPyrPad + fold(e, a.W[2]) - cA[2]) : 0u;
      f2_dst = on ? (uint32_t)j * kPyrR2B + 4u * (uint32_t)(kPyrPad + e - cA[2]) : 0u;
      f2_any = __builtin_amdgcn_readfirstlane(__builtin_amdgcn_ballot_w64(on) != 0);
    }
    const uint32_t svN = NLEV == 2 ? sv2 : sv3;  // approximation band: same lanes as the last level's details
    // (stores of rows the segment does not own go through a resource of size 0: see the level-1 wave)
    uint32_t dbytes[NLEV];
    const float* dp[NLEV][3];
#pragma unroll
    for (int l = 1; l < NLEV; ++l) {
      dbytes[l] = (a.dbg & 1) ? 0u : ((uint32_t)(a.H[l + 1] - 1) * (uint32_t)a.ds_h[l] + (uint32_t)a.W[l + 1]) * 4u;
#pragma unroll
      for (int b = 0; b < 3; ++b) dp[l][b] = a.det[l][b] + (int64_t)img * a.ds_b[l];
    }
    const uint32_t abytes = (a.dbg & 1) ? 0u : ((uint32_t)(a.H[NLEV] - 1) * (uint32_t)a.as_h + (uint32_t)a.W[NLEV]) * 4u;
    const float* const app = a.approx + (int64_t)img * a.as_b;

    const bool do2 = role == kRoleL2, do3 = NLEV >= 3 && role == kRoleL3;
    // rings: pad columns in zero mode and the zero rows (slot kPyrRing) must read as zero
    if (do2)
      for (int i = lane; i < (WB - kPyrStageB) / 16; i += 64) reinterpret_cast<f4*>(ring1)[i] = (f4){0.f, 0.f, 0.f, 0.f};
    __syncthreads();

    // ring slots: level-1 row i lives in slot (i + ro1) & 15, level-2 row i in (i + ro2) & 15
    const int ro1 = HP - 1 - rA[1], ro2 = HP - 1 - rA[2];
    const int E1 = 2 * rA[2] - HL;
    int E2 = 0;
    if constexpr (NLEV >= 3) E2 = 2 * rA[3] - HL;
    PyrAcc<L, 1> acc2, acc3;
    acc2.clear();
    acc3.clear();
    int ph2 = 0, ph3 = 0;  // pair index modulo L/2 of the next level-2 / level-3 block
    // horizontal pass of one row: ring row of the level above -> (lo, hi) of the lane's column
    auto load_win = [&](const unsigned char* row, f2 (&w)[HP]) {
#pragma unroll
      for (int k = 0; k < HP; ++k) w[k] = *reinterpret_cast<const f2*>(row + 8 * k);
    };
    // horizontal pass of the two rows of a pair, interleaved (two independent chains)
    auto h_pair = [&](const f2 (&wa)[HP], const f2 (&wb)[HP], f2 (&ha)[1], f2 (&hb)[1]) {
#pragma unroll
      for (int k = 0; k < HP; ++k) {
        if (k == 0) {
          ha[0] = vmul_lo(tap[L - 1], wa[0]);
          hb[0] = vmul_lo(tap[L - 1], wb[0]);
        } else {
          vfma_lo(ha[0], tap[L - 1 - 2 * k], wa[k]);
          vfma_lo(hb[0], tap[L - 1 - 2 * k], wb[k]);
        }
        vfma_hi(ha[0], tap[L - 2 - 2 * k], wa[k]);
        vfma_hi(hb[0], tap[L - 2 - 2 * k], wb[k]);
      }
    };

#pragma unroll 1
    for (int s = 0; s < nsteps; ++s) {
      pyr_barrier(a.prof, waited);
      if constexpr (NLEV >= 3) {
        // ===== level 3: two rows of the level-2 ring = one pair = one level-3 row =====
        if (do3 && s >= D3 && s - D3 < npair3 && !(a.dbg & 4)) {
          pyr_dispatch<HP>(ph3, [&](auto r_tag) {
          constexpr int R = decltype(r_tag)::value;  // (s - D3) mod L/2
          f2 w3[2][HP];
#pragma unroll
          for (int r = 0; r < 2; ++r) {
            const int e = E2 + 2 * (s - D3) + r;
            const bool dead = zero_mode && (unsigned)e >= (unsigned)a.H[2];
            const int slot = dead ? kPyrRing : ((fold(e, a.H[2]) + ro2) & (kPyrRing - 1));
            load_win(ring2 + slot * kPyrR2B + win3, w3[r]);
          }
          {
            f2 ha[1], hb[1];
            h_pair(w3[0], w3[1], ha, hb);
            acc3.template feed<0, R>(tap, ha);
            acc3.template feed<1, R>(tap, hb);
            {
              const int i = rA[3] + (s - D3) - (HP - 1);
              const f2 lo = acc3.lo[PyrAcc<L, 1>::done(R)][0], hi = acc3.hi[PyrAcc<L, 1>::done(R)][0];
              {
                const bool own = i >= oA[3] && i < oB[3];
                const uint32_t nb = own ? dbytes[2] : 0u;
                const uint32_t so = own ? (uint32_t)i * (uint32_t)a.ds_h[2] * 4u : 0u;
                pyr_store1(hi.x, pyr_rsrc(dp[2][0], nb), sv3, so);
                pyr_store1(lo.y, pyr_rsrc(dp[2][1], nb), sv3, so);
                pyr_store1(hi.y, pyr_rsrc(dp[2][2], nb), sv3, so);
                pyr_store1(lo.x, pyr_rsrc(app, own ? abytes : 0u), svN, own ? (uint32_t)i * (uint32_t)a.as_h * 4u : 0u);
              }
            }
          }
          });
          ph3 = ph3 + 1 == HP ? 0 : ph3 + 1;
        }
      }
      pyr_barrier(a.prof, waited);  // level 1's first two rows of this step (and everything before) are in ring 1
      // ===== level 2: four rows of the level-1 ring = two pairs = two level-2 rows =====
      if (do2 && s >= D2 && 2 * (s - D2) < npair2 && !(a.dbg & 4)) {
        pyr_dispatch<HP>(ph2, [&](auto r0_tag) {
        constexpr int R0 = decltype(r0_tag)::value;  // (2 (s - D2)) mod L/2
        f2 w2[4][HP];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int e = E1 + 4 * (s - D2) + r;
          const bool dead = zero_mode && (unsigned)e >= (unsigned)a.H[1];
          const int slot = dead ? kPyrRing : ((fold(e, a.H[1]) + ro1) & (kPyrRing - 1));
          load_win(ring1 + slot * kPyrR1B + win2, w2[r]);
        }
        pyr_static_for<2>([&](auto jj_tag) {
          constexpr int jj = decltype(jj_tag)::value;
          constexpr int R = (R0 + jj) % HP;
          f2 ha[1], hb[1];
          h_pair(w2[2 * jj], w2[2 * jj + 1], ha, hb);
          acc2.template feed<0, R>(tap, ha);
          acc2.template feed<1, R>(tap, hb);
          {
            const int p = 2 * (s - D2) + jj;
            const int i = rA[2] + p - (HP - 1);
            const f2 lo = acc2.lo[PyrAcc<L, 1>::done(R)][0], hi = acc2.hi[PyrAcc<L, 1>::done(R)][0];
            if constexpr (NLEV >= 3) *reinterpret_cast<float*>(ring2 + (p & (kPyrRing - 1)) * kPyrR2B + 4 * (kPyrPad + lane)) = lo.x;
            {
              const bool own = i >= oA[2] && i < oB[2];
              const uint32_t nb = own ? dbytes[1] : 0u;
              const uint32_t so = own ? (uint32_t)i * (uint32_t)a.ds_h[1] * 4u : 0u;
              pyr_store1(hi.x, pyr_rsrc(dp[1][0], nb), sv2, so);
              pyr_store1(lo.y, pyr_rsrc(dp[1][1], nb), sv2, so);
              pyr_store1(hi.y, pyr_rsrc(dp[1][2], nb), sv2, so);
              if constexpr (NLEV == 2) pyr_store1(lo.x, pyr_rsrc(app, own ? abytes : 0u), svN, own ? (uint32_t)i * (uint32_t)a.as_h * 4u : 0u);
            }
          }
        });
        });
        ph2 = (ph2 + 2) % HP;
        if constexpr (NLEV >= 3) {
          wave_lds_fence();
          if (f2_any) {
            unsigned char* rb = ring2 + ((2 * (s - D2)) & (kPyrRing - 1)) * kPyrR2B;
            const float v = *reinterpret_cast<const float*>(rb + f2_src);
            wave_lds_fence();
            *reinterpret_cast<float*>(rb + f2_dst) = v;
            wave_lds_fence();
          }
        }
      }
    }
    prof_out();
  }
}

unsigned long long* g_pyr_prof = nullptr;

// ---- host side ------------------------------------------------------------------------------------------------------------
struct PyrPlan {
  int nstrips, ngroups, nseg, seg_rows, cpw0, cpw;
};

// level-NLEV columns a strip can own: interior strips carry HL halo columns per level on their left, strip 0 none
static void pyr_strip_widths(int L, int nlev, int* cpw0, int* cpw) {
  const int HL = L - 2;
  int n = (253 - HL) / 2;  // level-1 columns from 256 staged level-0 columns (3 of them lost to the 16-byte alignment)
  n = n > 128 ? 128 : n;
  n &= ~1;
  for (int l = 2; l <= nlev; ++l) {
    n = (n - HL) / 2;
    if (n > 64) n = 64;
  }
  *cpw = n;
  int n0 = 128;
  for (int l = 2; l <= nlev; ++l) n0 = n0 / 2 > 64 ? 64 : n0 / 2;
  *cpw0 = n0;
}

static bool pyr_plan(int nlev, const mifwt_level_desc* const* d, PyrPlan* p) {
  const int L = d[0]->filt_len, HL = L - 2;
  const int WN = (int)d[nlev - 1]->coef_extent[1], HN = (int)d[nlev - 1]->coef_extent[0];
  pyr_strip_widths(L, nlev, &p->cpw0, &p->cpw);
  const int min_cols = HL + 2;  // an edge strip mirrors its own columns: it must own at least these at the last level
  if (WN < min_cols || HN < 2 * (HL + 2)) return false;
  if (WN <= p->cpw0) {
    p->nstrips = 1;
  } else {
    p->nstrips = 1 + (WN - p->cpw0 + p->cpw - 1) / p->cpw;
    const int last = WN - p->cpw0 - (p->nstrips - 2) * p->cpw;
    if (last < min_cols) p->cpw0 -= min_cols - last;  // shift the strip boundaries left so that the last strip is wide enough
    if (p->cpw0 < min_cols) return false;
  }
  p->ngroups = (p->nstrips + kPyrNW - 1) / kPyrNW;
  // row segments: about one workgroup per CU, at least 8 rows of the last level each, the last segment not shorter than 8
  int dev = 0, ncu = 256;
  if (hipGetDevice(&dev) == hipSuccess) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ncu = v;
  }
  const int64_t per_seg = d[0]->batch * p->ngroups;
  int nseg = g_options[MIFWT_OPT_PAIR_ROWS] > 0 ? (HN + g_options[MIFWT_OPT_PAIR_ROWS] - 1) / g_options[MIFWT_OPT_PAIR_ROWS]
                                                  : (int)((ncu + per_seg / 2) / (per_seg > 0 ? per_seg : 1));
  const int max_seg = HN / 8 > 0 ? HN / 8 : 1;
  nseg = nseg < 1 ? 1 : (nseg > max_seg ? max_seg : nseg);
  p->seg_rows = (HN + nseg - 1) / nseg;
  p->nseg = (HN + p->seg_rows - 1) / p->seg_rows;
  if (p->nseg > 1 && HN - (p->nseg - 1) * p->seg_rows < 8) --p->nseg;  // the kernel gives the last segment everything up to H
  return true;
}

bool dwt2_fwd_pyr_supported(int nlev, const mifwt_level_desc* const* d) {
  if (nlev < 1 || nlev > 3 || g_options[MIFWT_OPT_PAIR_MODE] == 2 || g_options[MIFWT_OPT_PYRAMID_MODE] == 2) return false;
  const mifwt_level_desc* d0 = d[0];
  const int L = d0->filt_len;
  if (d0->ndim != 2 || d0->dtype != MIFWT_F32 || L < 2 || L > 8 || (L & 1)) return false;
  if (d0->mode == MIFWT_MODE_PERIODIC || d0->mode < 0 || d0->mode > MIFWT_MODE_SYMMETRIC) return false;
  if (d0->batch < 1 || d0->sig_stride[2] != 1) return false;
  // LDS-DMA moves 16 aligned bytes per lane: rows must start on 16-byte boundaries and hold a multiple of 4 samples
  if ((d0->sig_extent[1] & 3) || (d0->sig_stride[1] & 3) || (d0->sig_stride[0] & 3)) return false;
  const int64_t lim = int64_t(1) << 29;  // byte offsets inside one image stay below 2^31
  if (d0->sig_extent[0] * d0->sig_stride[1] >= lim) return false;
  for (int l = 0; l < nlev; ++l) {
    const mifwt_level_desc* dl = d[l];
    if (dl->ndim != 2 || dl->dtype != MIFWT_F32 || dl->filt_len != L || dl->mode != d0->mode || dl->batch != d0->batch) return false;
    if (dl->detail_stride[2] != 1 || dl->coef_extent[0] * dl->detail_stride[1] >= lim) return false;
    for (int ax = 0; ax < 2; ++ax) {
      const int64_t n = l == 0 ? d0->sig_extent[ax] : d[l - 1]->coef_extent[ax];
      if (dl->sig_extent[ax] != n || dl->coef_extent[ax] != (n + L - 1) / 2) return false;
      if (n < 2 * L) return false;  // single-fold boundary map, pads mirrored from inside the first / last strip
    }
  }
  const mifwt_level_desc* dn = d[nlev - 1];
  if (dn->approx_stride[2] != 1 || dn->coef_extent[0] * dn->approx_stride[1] >= lim) return false;
  PyrPlan p;
  return pyr_plan(nlev, d, &p);
}

template <int L, int NLEV>
static int launch_pyr(const mifwt_level_desc* const* d, const void* x, void* const* const* details, void* approx, const double* lo,
                      const double* hi, hipStream_t stream) {
  PyrPlan p;
  if (!pyr_plan(NLEV, d, &p)) return MIFWT_ERR_UNSUPPORTED;
  PyrArgs<L, NLEV> a;
  a.x = static_cast<const float*>(x);
  a.xs_b = d[0]->sig_stride[0];
  a.xs_h = (int)d[0]->sig_stride[1];
  a.H[0] = (int)d[0]->sig_extent[0];
  a.W[0] = (int)d[0]->sig_extent[1];
  for (int l = 0; l < NLEV; ++l) {
    for (int b = 0; b < 3; ++b) a.det[l][b] = static_cast<float*>(details[l][b]);
    a.ds_b[l] = d[l]->detail_stride[0];
    a.ds_h[l] = (int)d[l]->detail_stride[1];
    a.H[l + 1] = (int)d[l]->coef_extent[0];
    a.W[l + 1] = (int)d[l]->coef_extent[1];
  }
  a.approx = static_cast<float*>(approx);
  a.as_b = d[NLEV - 1]->approx_stride[0];
  a.as_h = (int)d[NLEV - 1]->approx_stride[1];
  a.nstrips = p.nstrips;
  a.ngroups = p.ngroups;
  a.nseg = p.nseg;
  a.seg_rows = p.seg_rows;
  a.cpw0 = p.cpw0;
  a.cpw = p.cpw;
  a.mode = d[0]->mode;
  a.dbg = g_options[MIFWT_OPT_DEBUG];
  a.prof = g_pyr_prof;
  for (int m = 0; m < L; ++m) a.tap[m] = (f2){(float)lo[m], (float)hi[m]};
  const int64_t nwg = d[0]->batch * p.nseg * p.ngroups;
  if (nwg > INT32_MAX / 8) return MIFWT_ERR_UNSUPPORTED;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&dwt2_fwd_pyr_kernel<L, NLEV>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            pyr_lds_bytes(NLEV)) != hipSuccess)
      return MIFWT_ERR_LAUNCH;
    attr_set = true;
  }
  hipLaunchKernelGGL((dwt2_fwd_pyr_kernel<L, NLEV>), dim3((unsigned)nwg), dim3(64 * pyr_nwaves(NLEV)), pyr_lds_bytes(NLEV), stream, a);
  return hipGetLastError() == hipSuccess ? MIFWT_OK : MIFWT_ERR_LAUNCH;
}

template <int L>
static int launch_pyr_l(int nlev, const mifwt_level_desc* const* d, const void* x, void* const* const* details, void* approx,
                        const double* lo, const double* hi, hipStream_t stream) {
  switch (nlev) {
    case 1: return launch_pyr<L, 1>(d, x, details, approx, lo, hi, stream);
    case 2: return launch_pyr<L, 2>(d, x, details, approx, lo, hi, stream);
    case 3: return launch_pyr<L, 3>(d, x, details, approx, lo, hi, stream);
    default: return MIFWT_ERR_UNSUPPORTED;
  }
}

int dwt2_fwd_pyr(int nlev, const mifwt_level_desc* const* d, const void* x, void* const* const* details, void* approx,
                 const double* lo, const double* hi, hipStream_t stream) {
  if (!dwt2_fwd_pyr_supported(nlev, d)) return MIFWT_ERR_UNSUPPORTED;
  switch (d[0]->filt_len) {
    case 2: return launch_pyr_l<2>(nlev, d, x, details, approx, lo, hi, stream);
    case 4: return launch_pyr_l<4>(nlev, d, x, details, approx, lo, hi, stream);
    case 6: return launch_pyr_l<6>(nlev, d, x, details, approx, lo, hi, stream);
    case 8: return launch_pyr_l<8>(nlev, d, x, details, approx, lo, hi, stream);
    default: return MIFWT_ERR_UNSUPPORTED;
  }
}

}  // namespace mifwt
