// mifwt_dwt2_inv_pyr.hip — UP TO THREE consecutive 2-D synthesis levels of a big plane in one launch (gfx950), kernel id 22.
//
// Seam: NLEV trips of waverec2's level loop (src/ptwt/conv_transform_2.py:222-249: torch.stack + F.conv_transpose2d(stride 2) +
// crops, the result fed back as the next approximation).  The approximations between the levels are intermediates that
// one-kernel-per-level writes to HBM and reads straight back; here they only exist as LDS rings.  The synthesis mirror of
// mifwt_dwt2_fwd_pyr.hip (kernel id 16), coarse -> fine; tests/test_ipyr_model.py is the CPU model of everything below (row
// ranges, schedule, staging entries, rings) and checks every LDS read for "written in an earlier sub-step".
//
// Shape of the work:
//   * a workgroup owns one ROW SEGMENT of the output plane of one image (whole rows: planes up to ~1500 columns) and runs one wave
//     per role: level-1 waves (the finest level: 4 output columns = 2 coefficient columns per lane, 256 output columns per
//     wave), level-2 waves, level-3 waves and three LOADER waves.  Config 2 (64 x 1024^2, db4, 3 levels): 4 + 3 + 2 + 3 waves,
//     4 segments per image = 256 workgroups, one per CU.
//   * coefficient rows STREAM top-down.  Per level and coefficient row: the horizontal polyphase pass first (the four bands of
//     the row are read from LDS through a window of L/2 + 1 coefficients per lane: (aa, ad) -> the vertically-low image row,
//     (da, dd) -> the vertically-high one), then the vertical pass with the L/2 output row PAIRS in flight in registers (rolling
//     accumulators, slot rotation resolved at compile time): 8 packed FMAs per output sample, the direct-form minimum.  A
//     finished pair of a coarse level goes into that level's LDS ring as two rows of the next finer level's approximation;
//     the finest level stores 16 bytes per lane and row.
//   * synthesis halos are small (L/2 - 1 coefficient rows per level and segment, against 3 (L - 2) input rows per level in the
//     analysis kernel): a segment re-reads 2-3 % of the coefficients.
//   * the LOADER waves issue every global load as LDS-DMA (buffer_load_dwordx4 ... lds, non-temporal), `nbuf - 2` sub-steps
//     ahead, into a ring of staging entries; rows of ANY alignment and width (rows of 515 floats start on 4-byte boundaries
//     only: tools/dma_probe.hip — the DMA engine takes them, lanes switched off in EXEC leave LDS alone).  One s_barrier per
//     sub-step (2 coefficient rows of level 1 = 4 output rows) hands a landed entry over.  The compute waves' vmcnt queues hold
//     stores only and are never waited on.
//   * schedule (sub-steps t = 0 .. nsub - 1): level 3 consumes one coefficient row per STEP (two sub-steps), level 2 two rows per
//     step D2 steps later (both at odd sub-steps), level 1 two rows per sub-step from sub-step T1 on (mifwt ipyr_lags()).
// Results agree with the per-level kernels to rounding (different summation order), with the fp64 oracle within 1e-6.
// f32, even L <= 8, unit innermost strides; the three detail bands of a level share their strides.
// Algorithmic traffic: 4 B (4 M_N + 3 sum_{l<N} M_l) read + 4 B H W written.
#include "mifwt_pyr.h"

namespace mifwt {

constexpr int kIpWaves = 16;
constexpr int kIpRing1 = 16;  // rows of the level-1 approximation ring (written by the level-2 waves)
constexpr int kIpRing2 = 8;   // rows of the level-2 approximation ring
constexpr int kIpLoaders = 3;
constexpr int kIpFirstLoader = kIpWaves - kIpLoaders;

// lags of the schedule (tests/test_ipyr_model.py: ipyr_schedule)
constexpr int ipyr_d2(int L, int nlev) { return nlev >= 3 ? L / 2 : 0; }
constexpr int ipyr_t1(int L, int nlev) { return nlev == 1 ? 0 : 2 * ipyr_d2(L, nlev) + L / 2 + 1; }

template <int L, int NLEV>
struct IPyrArgs {
  const float* band[NLEV][4];  // [l - 1][b]: band b (aa, ad, da, dd) of level l (1 = finest); aa only for the coarsest
  int64_t bs_b[NLEV][2];       // image strides (elements) of [approximation, details]
  int bs_h[NLEV][2];           // row strides
  float* y;
  int64_t ys_b;
  int ys_h;
  int Mh[NLEV], Mw[NLEV];  // coefficient extents
  int H, W;                // output extents of the finest level
  int nseg, seg_rows;
  int nbuf;
  int pitchS[NLEV];  // bytes of a staged row
  int pitchR[2];     // bytes of a row of the level-1 / level-2 approximation ring
  int offL[NLEV];    // byte offset of a level's rows inside a staging entry
  int entry_bytes;
  int nS[NLEV];      // waves per level
  int tw;            // fast warm-up: entries 0 .. tw - 1 (tw = T1 or 0) have slots of their own (their level-2 / level-3 rows), all
                     // requested when the workgroup starts; the ring of nbuf entries begins with entry tw
  int wu2_off, wu3_off;  // LDS byte offsets of those slots: level-2 rows of entries 2 D2 .. tw - 1, level-3 rows of entries 0 .. tw - 1
  int dbg;
  f2 tlo[L / 2];  // (rec_lo[2j], rec_lo[2j+1])
  f2 thi[L / 2];
  DevTapArg dt;   // device-resident taps (mifwt_common.h); dt.lo == nullptr: tlo / thi count
};

// where the tap pairs live: SGPR pairs (default) or VGPR pairs (experiment build -DMIFWT_IPYR_TAPV: a lone wave issues a packed FMA with
// an SGPR-pair operand every ~6.9 cycles, with VGPR operands every ~5.3 — tools/ubench.hip — at the price of L registers)
#ifdef MIFWT_IPYR_TAPV
#define MIFWT_TAPC "v"
#else
#define MIFWT_TAPC "s"
#endif
__device__ __forceinline__ void ivfma_lo(f2& acc, const f2 tap, const f2 pair) {
  asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc) : MIFWT_TAPC(tap), "v"(pair));
}
__device__ __forceinline__ void ivfma_hi(f2& acc, const f2 tap, const f2 pair) {
  asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(acc) : MIFWT_TAPC(tap), "v"(pair));
}
__device__ __forceinline__ f2 ivmul_lo(const f2 tap, const f2 pair) {
  f2 r;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(r) : MIFWT_TAPC(tap), "v"(pair));
  return r;
}
__device__ __forceinline__ f2 ivmul_hi(const f2 tap, const f2 pair) {
  f2 r;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(r) : MIFWT_TAPC(tap), "v"(pair));
  return r;
}
// acc (+)= tap.x * v / tap.y * v with the tap pair in an SGPR pair and BOTH halves of v (two neighbouring columns)
__device__ __forceinline__ void vfma_tx(f2& acc, const f2 tap, const f2 v) {
  asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc) : MIFWT_TAPC(tap), "v"(v));
}
__device__ __forceinline__ void vfma_ty(f2& acc, const f2 tap, const f2 v) {
  asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc) : MIFWT_TAPC(tap), "v"(v));
}
__device__ __forceinline__ f2 vmul_tx(const f2 tap, const f2 v) {
  f2 r;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(r) : MIFWT_TAPC(tap), "v"(v));
  return r;
}
__device__ __forceinline__ f2 vmul_ty(const f2 tap, const f2 v) {
  f2 r;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]" : "=v"(r) : MIFWT_TAPC(tap), "v"(v));
  return r;
}

// rolling vertical pass of the synthesis: the L/2 output row pairs in flight of a lane's two column pairs.  r0 = the even row of a
// pair, r1 = the odd one, each an f2 over two neighbouring columns.  Pair p lives in slot p mod L/2 for its whole life.
template <int L>
struct IpAcc {
  static constexpr int HL = L / 2;
  f2 r0[HL][2], r1[HL][2];
  // coefficient row k (R = k mod HL), horizontally synthesised: vl = vertically-low image row, vh = vertically-high one
  template <int R>
  __device__ __forceinline__ void feed(const f2 (&tlo)[HL], const f2 (&thi)[HL], const f2 (&vl)[2], const f2 (&vh)[2]) {
#pragma unroll
    for (int i = 0; i < HL; ++i) {
      const int sl = (R - i + HL) % HL;  // pair k - i
      const int j = HL - 1 - i;
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        if (i == 0) {
          r0[sl][e] = vmul_tx(tlo[j], vl[e]);
          r1[sl][e] = vmul_ty(tlo[j], vl[e]);
        } else {
          vfma_tx(r0[sl][e], tlo[j], vl[e]);
          vfma_ty(r1[sl][e], tlo[j], vl[e]);
        }
        vfma_tx(r0[sl][e], thi[j], vh[e]);
        vfma_ty(r1[sl][e], thi[j], vh[e]);
      }
    }
  }
  static constexpr int done(int R) { return (R + 1) % HL; }  // slot of the pair row k completes (k - (HL - 1))
};

// horizontal synthesis of one coefficient row under a lane: cl / ch = the windows of the low / high branch (f2 pieces from
// coefficient column 2 G on), v[e] = output columns 4 G + 2 e, + 1:  y[2 q + r] = sum_t g[2 (HL-1-t) + r] c[q + t]
template <int L>
__device__ __forceinline__ void ipyr_hsyn(const f2 (&tlo)[L / 2], const f2 (&thi)[L / 2], const f2 (&cl)[L / 4 + 1], const f2 (&ch)[L / 4 + 1],
                                          f2 (&v)[2]) {
  constexpr int HL = L / 2;
#pragma unroll
  for (int t = 0; t < HL; ++t) {
    const int j = HL - 1 - t;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int u = e + t;
      if (t == 0) {
        v[e] = (u & 1) ? ivmul_hi(tlo[j], cl[u >> 1]) : ivmul_lo(tlo[j], cl[u >> 1]);
      } else {
        if (u & 1) ivfma_hi(v[e], tlo[j], cl[u >> 1]);
        else ivfma_lo(v[e], tlo[j], cl[u >> 1]);
      }
      if (u & 1) ivfma_hi(v[e], thi[j], ch[u >> 1]);
      else ivfma_lo(v[e], thi[j], ch[u >> 1]);
    }
  }
}

// f(integral_constant<int, ph>) for the phase ph = (2 k) mod HL of a block of two coefficient rows 2 k, 2 k + 1
template <int HL, typename F>
__device__ __forceinline__ void ipyr_phase2(int ph, F&& f) {
  if constexpr (HL == 1 || HL == 2) {
    f(std::integral_constant<int, 0>{});
  } else if constexpr (HL == 3) {
    if (ph == 0) f(std::integral_constant<int, 0>{});
    else if (ph == 1) f(std::integral_constant<int, 1>{});
    else f(std::integral_constant<int, 2>{});
  } else if constexpr (HL == 4) {
    if (ph == 0) f(std::integral_constant<int, 0>{});
    else f(std::integral_constant<int, 2>{});
  } else {
    static_assert(HL == 5, "filter lengths up to 10");
    if (ph == 0) f(std::integral_constant<int, 0>{});
    else if (ph == 1) f(std::integral_constant<int, 1>{});
    else if (ph == 2) f(std::integral_constant<int, 2>{});
    else if (ph == 3) f(std::integral_constant<int, 3>{});
    else f(std::integral_constant<int, 4>{});
  }
}

// f(integral_constant<int, r>) for the runtime r in [0, HL): the phase of a single coefficient row
template <int HL, typename F>
__device__ __forceinline__ void ipyr_phase1(int r, F&& f) {
  if constexpr (HL <= 4) {
    pyr_dispatch<HL>(r, f);
  } else {
    static_assert(HL == 5, "filter lengths up to 10");
    if (r < 2) {
      if (r == 0) f(std::integral_constant<int, 0>{});
      else f(std::integral_constant<int, 1>{});
    } else if (r == 2) {
      f(std::integral_constant<int, 2>{});
    } else if (r == 3) {
      f(std::integral_constant<int, 3>{});
    } else {
      f(std::integral_constant<int, 4>{});
    }
  }
}

// 16-byte store.  A VALU write of the store's LAST data register in the cycle after a store of more than 8 bytes corrupts that
// dword (the compiler's hazard table assumes a scalar soffset lifts this hazard; on gfx950 it does not: the ragged-column path's
// v_cndmask landed in that slot and every few rows a lane's fourth column came out wrong) — the wait states travel with the store
__device__ __forceinline__ void ipyr_store4(const f4 data, rsrc_t rsrc, uint32_t voff, uint32_t soff) {
  // write-through to memory (sc0 sc1): these are full 1-KiB pieces per wave and row, nothing reads them back — 97.8-98.5 against
  // 100.9-101.0 us per launch on config 2 with the default policy, in one run (profiles/r03g_store_policy_nbuf.txt; a plain copy
  // kernel gains the same 3 %, profiles/r03f_wbench.txt).  (The analysis kernel's 8-byte stores of partial lines LOSE 6 % with it.)
#if MIFWT_ST_AUX == 99
  asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen\n\ts_nop 1" ::"v"(data), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
#elif defined(MIFWT_IPYR_ST_NT)
  asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen nt\n\ts_nop 1" ::"v"(data), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
#else
  asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen sc0 sc1\n\ts_nop 1" ::"v"(data), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
#endif
}

// one LDS-DMA piece: lanes whose 16 bytes start inside the row (voff < limit) move them to LDS [lds + 16 lane)
__device__ __forceinline__ void ipyr_dma(uint32_t voff, uint32_t limit, rsrc_t rsrc, uint32_t soff, uint32_t lds) {
  if (voff < limit) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen " MIFWT_PYR_DMA_POLICY " lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(soff), "s"(lds) : "memory");
  }
}

__device__ __forceinline__ void ipyr_wait_vm(int n) {
  // (s_waitcnt takes an immediate: the loaders' request counts depend on the plane widths)
  switch (n < 63 ? n : 63) {
#define MIFWT_W(k) case k: pyr_wait_vm<k>(); break;
#define MIFWT_W8(k) MIFWT_W(k) MIFWT_W(k + 1) MIFWT_W(k + 2) MIFWT_W(k + 3) MIFWT_W(k + 4) MIFWT_W(k + 5) MIFWT_W(k + 6) MIFWT_W(k + 7)
    MIFWT_W8(0) MIFWT_W8(8) MIFWT_W8(16) MIFWT_W8(24) MIFWT_W8(32) MIFWT_W8(40) MIFWT_W8(48) MIFWT_W8(56)
#undef MIFWT_W8
#undef MIFWT_W
    default: pyr_wait_vm<0>(); break;
  }
}

// DT: the taps come from device memory (a.dt; a learnable filter bank that lives on the GPU) — an instance of its own: the synthesis waves
// of the default instance run at the 128-register limit, and the loads' temporaries cost it 56 bytes of scratch and a quarter of its speed
// (waverec2 of config 2 0.0926 -> 0.1152 ms with one instance for both)
template <int L, int NLEV, bool DT>
__global__ void __launch_bounds__(64 * kIpWaves) idwt2_pyr_kernel(const IPyrArgs<L, NLEV> a) {
  constexpr int HL = L / 2;
  constexpr int NW = L / 4 + 1;  // f2 pieces of a window: coefficient columns 2 G .. 2 G + HL
  constexpr int D2 = ipyr_d2(L, NLEV), T1 = ipyr_t1(L, NLEV);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // wave -> (role, index): role 1 .. NLEV = synthesis level, 0 = loader, -1 = nothing to do
  int role = -1, widx = 0;
  {
    int w = wave;
    if (w >= kIpFirstLoader) {
      role = 0;
      widx = w - kIpFirstLoader;
    } else {
#pragma unroll
      for (int l = 1; l <= NLEV; ++l) {
        if (role < 0 && w < a.nS[l - 1]) {
          role = l;
          widx = w;
        }
        w -= a.nS[l - 1];
      }
    }
  }
  if (role < 0) return;  // (a wave that has ended does not take part in the barriers of the others)

  f2 tlo_in[L / 2], thi_in[L / 2];
  if constexpr (DT) {
#pragma unroll
    for (int j = 0; j < L / 2; ++j) {
      tlo_in[j] = (f2){dtap_lo<float>(a.dt, 2 * j), dtap_lo<float>(a.dt, 2 * j + 1)};
      thi_in[j] = (f2){dtap_hi<float>(a.dt, 2 * j), dtap_hi<float>(a.dt, 2 * j + 1)};
    }
  }

  const int seg = blockIdx.x % a.nseg, img = blockIdx.x / a.nseg;
  const int y0 = seg * a.seg_rows, y1 = min(a.H, y0 + a.seg_rows);
  // coefficient rows [ra[l], rb[l]] (inclusive) of level l this segment consumes
  int ra[NLEV + 1], rb[NLEV + 1];
  ra[1] = y0 >> 1;
  rb[1] = ((y1 + 1) >> 1) - 1 + HL - 1;
#pragma unroll
  for (int l = 2; l <= NLEV; ++l) {
    ra[l] = ra[l - 1] >> 1;
    rb[l] = (rb[l - 1] >> 1) + HL - 1;
  }
  const int nsub = T1 + ((rb[1] - ra[1] + 2) >> 1);
  const int ahead = a.nbuf - 2;
  const int TW = a.tw;
  const uint32_t stage_off = 0;
  const uint32_t ring1_off = stage_off + (uint32_t)(a.nbuf * a.entry_bytes);
  const uint32_t ring2_off = ring1_off + (NLEV >= 2 ? (uint32_t)(kIpRing1 * a.pitchR[0]) : 0u);

  // =====================================================================================================================
  // loader waves: item i of a staging entry belongs to loader i mod 3
  if (role == 0) {
    // items of an entry, in a fixed order: level 1: (band, row j) ...; level 2: (band) ...; level 3: two band rows
    constexpr int NB1 = NLEV == 1 ? 4 : 3, NB2 = NLEV == 2 ? 4 : 3;
    constexpr int N1 = 2 * NB1, N2 = NLEV >= 2 ? NB2 : 0, N3 = NLEV >= 3 ? 2 : 0;
    constexpr int NITEMS = N1 + N2 + N3;
    const uint32_t lane16 = 16u * (uint32_t)lane;
    if (!(MIFWT_DBG(a) & 16)) __builtin_amdgcn_s_setprio(3);  // (config 2: 90.8-92.0 us per call with, 93.6-97.1 without, tools/pyr_prio_ab.py)
    auto run = [&](auto w_tag) {
      constexpr int WI = decltype(w_tag)::value;
      // resources of this loader's items (a level-3 item alternates between two bands with the parity of the sub-step)
      constexpr int NMINE = (NITEMS - WI + kIpLoaders - 1) / kIpLoaders;
      rsrc_t rs[NMINE][2];
      uint32_t rowb[NMINE][2];  // bytes between rows
      int per = 0;              // requests of this loader per sub-step
      pyr_static_for<NMINE>([&](auto k_tag) {
        constexpr int K = decltype(k_tag)::value, I = WI + kIpLoaders * K;
        constexpr int LV = I < N1 ? 1 : (I < N1 + N2 ? 2 : 3);
        const int mw = a.Mw[LV - 1], mh = a.Mh[LV - 1];
#pragma unroll
        for (int v = 0; v < 2; ++v) {
          int b;
          if constexpr (LV == 1) b = (4 - NB1) + I / 2;
          else if constexpr (LV == 2) b = (4 - NB2) + (I - N1);
          else b = 2 * v + (I - N1 - N2);
          const int sx = b == 0 ? 0 : 1;
          const uint32_t bytes = ((uint32_t)(mh - 1) * (uint32_t)a.bs_h[LV - 1][sx] + (uint32_t)mw) * 4u;
          rs[K][v] = pyr_rsrc(a.band[LV - 1][b] + (int64_t)img * a.bs_b[LV - 1][sx], (MIFWT_DBG(a) & 2) ? 0u : bytes);
          rowb[K][v] = (uint32_t)a.bs_h[LV - 1][sx] * 4u;
        }
        per += (mw + 255) >> 8;
      });
      const rsrc_t dead = pyr_rsrc(a.band[0][1], 0);  // every lane out of range: zeros land
      int ib = 0;
      auto issue = [&](int t) {
        const bool warm = t < TW;  // (the level-1 rows of such an entry are before the segment: nothing to fetch)
        const uint32_t ent = stage_off + (uint32_t)ib * (uint32_t)a.entry_bytes;
        if (!warm) ib = ib + 1 == a.nbuf ? 0 : ib + 1;
        pyr_static_for<NMINE>([&](auto k_tag) {
          constexpr int K = decltype(k_tag)::value, I = WI + kIpLoaders * K;
          constexpr int LV = I < N1 ? 1 : (I < N1 + N2 ? 2 : 3);
          int r, v = 0;
          uint32_t dst;
          bool alive;
          if constexpr (LV == 1) {
            constexpr int SB = I / 2, J = I & 1;
            r = ra[1] + 2 * (t - T1) + J;
            alive = t >= T1 && r <= rb[1];
            dst = ent + (uint32_t)a.offL[0] + (uint32_t)((2 * SB + J) * a.pitchS[0]);
            if (warm) return;
          } else if constexpr (LV == 2) {
            constexpr int SB = I - N1;
            r = ra[NLEV >= 2 ? 2 : 1] + t - 2 * D2;
            alive = r >= ra[NLEV >= 2 ? 2 : 1] && r <= rb[NLEV >= 2 ? 2 : 1];
            dst = ent + (uint32_t)a.offL[NLEV >= 2 ? 1 : 0] + (uint32_t)(SB * a.pitchS[NLEV >= 2 ? 1 : 0]);
            if (warm) {
              if (t < 2 * D2) return;
              dst = (uint32_t)a.wu2_off + (uint32_t)(((t - 2 * D2) * NB2 + SB) * a.pitchS[NLEV >= 2 ? 1 : 0]);
            }
          } else {
            constexpr int SI = I - N1 - N2;
            r = ra[NLEV >= 3 ? 3 : 1] + (t >> 1);
            v = t & 1;
            alive = r <= rb[NLEV >= 3 ? 3 : 1];
            dst = ent + (uint32_t)a.offL[NLEV >= 3 ? 2 : 0] + (uint32_t)(SI * a.pitchS[NLEV >= 3 ? 2 : 0]);
            if (warm) dst = (uint32_t)a.wu3_off + (uint32_t)((2 * t + SI) * a.pitchS[NLEV >= 3 ? 2 : 0]);
          }
          const int mw = a.Mw[LV - 1];
          const uint32_t limit = 4u * (uint32_t)mw;
          const rsrc_t rr = alive ? (v ? rs[K][1] : rs[K][0]) : dead;
          const uint32_t soff = alive ? (uint32_t)r * (v ? rowb[K][1] : rowb[K][0]) : 0u;
          const int nch = (mw + 255) >> 8;
#pragma unroll 1
          for (int c = 0; c < nch; ++c) ipyr_dma(lane16 + 1024u * (uint32_t)c, limit, rr, soff, dst + 1024u * (uint32_t)c);
        });
      };
      for (int t = 0; t < TW + ahead; ++t)
        if (t < nsub) issue(t);
#pragma unroll 1
      for (int t = 0; t < nsub; ++t) {
        // entry t must have landed; the ones requested after it may still be in flight (before sub-step TW: every warm-up entry —
        // they were requested first — but none of the ring's entries)
        const int later = t < TW ? min(ahead, nsub - TW) : min(ahead - 1, nsub - 1 - t);
        ipyr_wait_vm(later * per);
        __syncthreads();
        if (t >= TW && t + ahead < nsub) issue(t + ahead);  // into the buffer of entry t - 2, which nobody reads any more
      }
    };
    if (widx == 0) run(std::integral_constant<int, 0>{});
    else if (widx == 1) run(std::integral_constant<int, 1>{});
    else run(std::integral_constant<int, 2>{});
    return;
  }

  // =====================================================================================================================
  // synthesis waves
  f2 tlo[HL], thi[HL];
#pragma unroll
  for (int j = 0; j < HL; ++j) {
    tlo[j] = DT ? tlo_in[j] : a.tlo[j];
    thi[j] = DT ? thi_in[j] : a.thi[j];
  }
  IpAcc<L> acc;
#pragma unroll
  for (int q = 0; q < HL; ++q)
#pragma unroll
    for (int e = 0; e < 2; ++e) acc.r0[q][e] = acc.r1[q][e] = (f2){0.f, 0.f};
  const int G = 64 * widx + lane;  // lane of the level's grid: coefficient columns 2 G, 2 G + 1 -> output columns 4 G .. 4 G + 3

  // the two rows of a block: windows of the four bands, horizontal passes, vertical pass, finished pairs handed to `emit(j, p, slot)`
  auto rows2 = [&](auto r0_tag, const unsigned char* (&src)[2][4], auto&& emit) {
    constexpr int R0 = decltype(r0_tag)::value;
    f2 vl[2][2], vh[2][2];
#ifdef MIFWT_IPYR_TAPV
    constexpr bool kTwoRows = L <= 6;
#else
    constexpr bool kTwoRows = L <= 8;
#endif
    if constexpr (kTwoRows) {
      f2 w[2][4][NW];
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
          for (int m = 0; m < NW; ++m) w[j][b][m] = *reinterpret_cast<const f2*>(src[j][b] + 8 * m);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        ipyr_hsyn<L>(tlo, thi, w[j][0], w[j][1], vl[j]);
        ipyr_hsyn<L>(tlo, thi, w[j][2], w[j][3], vh[j]);
      }
    } else {  // ten taps: 40 registers of accumulators — one row's windows at a time
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        f2 w[4][NW];
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
          for (int m = 0; m < NW; ++m) w[b][m] = *reinterpret_cast<const f2*>(src[j][b] + 8 * m);
        ipyr_hsyn<L>(tlo, thi, w[0], w[1], vl[j]);
        ipyr_hsyn<L>(tlo, thi, w[2], w[3], vh[j]);
        asm volatile("" ::: "memory");  // (keeps the second row's window loads behind the first row's passes)
      }
    }
    acc.template feed<R0 % HL>(tlo, thi, vl[0], vh[0]);
    emit(std::integral_constant<int, 0>{}, std::integral_constant<int, IpAcc<L>::done(R0 % HL)>{});
    acc.template feed<(R0 + 1) % HL>(tlo, thi, vl[1], vh[1]);
    emit(std::integral_constant<int, 1>{}, std::integral_constant<int, IpAcc<L>::done((R0 + 1) % HL)>{});
  };

  if (MIFWT_DBG(a) & 32) __builtin_amdgcn_s_setprio(1);  // (experiment: the synthesis waves above the default priority as well — no change)
  // ---- level 1 (the finest): output rows to global memory -----------------------------------------------------------------
  if (role == 1) {
    const int nq = (a.W + 3) >> 2;  // lanes with an output column
    const int Gr = min(G, nq - 1);
    const bool full = 4 * G + 3 < a.W;
    const int nrag = (!full && 4 * G < a.W) ? a.W - 4 * G : 0;  // 1 .. 3 columns of the last lane of a ragged plane
    const bool rag_any = __builtin_amdgcn_readfirstlane(__builtin_amdgcn_ballot_w64(nrag != 0) != 0);
    const uint32_t ybytes = (MIFWT_DBG(a) & 1) ? 0u : ((uint32_t)(a.H - 1) * (uint32_t)a.ys_h + (uint32_t)a.W) * 4u;
    const rsrc_t ry = pyr_rsrc(a.y + (int64_t)img * a.ys_b, ybytes);
    const uint32_t sv4 = full ? 16u * (uint32_t)G : kPyrOob;
    const uint32_t svr = nrag ? 16u * (uint32_t)G : kPyrOob;
    const uint32_t win = 8u * (uint32_t)Gr;
    constexpr int NB1 = NLEV == 1 ? 4 : 3;
    int eb = 0;
    // (round 5: the phase of a sub-step's two coefficient rows — the accumulators' slot rotation — is a COMPILE-TIME constant: the
    // sub-step loop is unrolled over the period of 2 k mod L/2 instead of switching between L/2 variants of the body at run time,
    // which cost a branch tree and a dozen accumulator copies per sub-step where the variants' register assignments met again;
    // as in the analysis kernel, mifwt_dwt2_fwd_pyr.hip)
    auto substep1 = [&](auto r0_tag, int t) {
      __syncthreads();
      if (!(MIFWT_DBG(a) & 4)) {
        const int r1 = ra[1] + 2 * (t - T1);
        const unsigned char* ent = smem + stage_off + eb * a.entry_bytes + a.offL[0];
        const unsigned char* src[2][4];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int b = 0; b < 4; ++b) {
            if (b == 0 && NLEV >= 2) src[j][b] = smem + ring1_off + ((r1 + j) & (kIpRing1 - 1)) * a.pitchR[0] + win;
            else src[j][b] = ent + (2 * (b - (4 - NB1)) + j) * a.pitchS[0] + win;
          }
        {
          rows2(r0_tag, src, [&](auto j_tag, auto slot_tag) {
            constexpr int j = decltype(j_tag)::value, slot = decltype(slot_tag)::value;
            const int p = r1 + j - (HL - 1);
            const bool on = p >= ra[1];
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
              const int row = 2 * p + rr;
              const bool own = on && row < y1;
              const uint32_t so = own ? (uint32_t)row * (uint32_t)a.ys_h * 4u : 0u;
              const f2 (&q)[2] = rr ? acc.r1[slot] : acc.r0[slot];
              ipyr_store4((f4){q[0].x, q[0].y, q[1].x, q[1].y}, ry, own ? sv4 : kPyrOob, so);
              if (rag_any) {
                const uint32_t v = own ? svr : kPyrOob;
                pyr_store1(q[0].x, ry, v, so);
                pyr_store1(q[0].y, ry, nrag >= 2 ? v : kPyrOob, so + 4u);
                pyr_store1(q[1].x, ry, nrag >= 3 ? v : kPyrOob, so + 8u);
              }
            }
          });
        }
      }
      if (t >= TW) eb = eb + 1 == a.nbuf ? 0 : eb + 1;
    };
    {
      int t = 0;
#pragma unroll 1
      for (; t < T1 && t < nsub; ++t) {  // (the lag: the coarser levels have not produced the first rows yet)
        __syncthreads();
        if (t >= TW) eb = eb + 1 == a.nbuf ? 0 : eb + 1;
      }
      constexpr int PERIOD = (HL % 2 == 0) ? (HL / 2 > 0 ? HL / 2 : 1) : HL;  // sub-steps until 2 k mod L/2 repeats
#pragma unroll 1
      for (; t < nsub; t += PERIOD) {
        pyr_static_for<PERIOD>([&](auto k_tag) {
          constexpr int k = decltype(k_tag)::value;
          if (k == 0 || t + k < nsub) substep1(std::integral_constant<int, (2 * k) % HL>{}, t + k);
        });
      }
    }
    return;
  }

  // ---- level 2: two coefficient rows per step (at odd sub-steps), output rows into ring 1 ---------------------------------------
  if constexpr (NLEV >= 2) {
    if (role == 2) {
      const int nq = (a.Mw[0] + 3) >> 2;
      const int Gr = min(G, nq - 1);
      const bool wr = G < nq;
      const uint32_t win = 8u * (uint32_t)Gr;
      const uint32_t wq = 16u * (uint32_t)Gr;
      constexpr int NB2 = NLEV == 2 ? 4 : 3;
      int eb = 0, ph = 0;
#pragma unroll 1
      for (int t = 0; t < nsub; ++t) {
        __syncthreads();
        const int s = t >> 1;
        const int r2 = ra[2] + 2 * (s - D2);
        if ((t & 1) && s >= D2 && r2 <= rb[2] && !(MIFWT_DBG(a) & 4)) {
          const int ebp = eb == 0 ? a.nbuf - 1 : eb - 1;  // entry t - 1
          const unsigned char* src[2][4];
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const int e = t - 1 + j;
            const unsigned char* ent = e < TW ? smem + a.wu2_off + (e - 2 * D2) * NB2 * a.pitchS[1]
                                              : smem + stage_off + (j ? eb : ebp) * a.entry_bytes + a.offL[1];
#pragma unroll
            for (int b = 0; b < 4; ++b) {
              if (b == 0 && NLEV >= 3) src[j][b] = smem + ring2_off + ((r2 + j) & (kIpRing2 - 1)) * a.pitchR[1] + win;
              else src[j][b] = ent + (b - (4 - NB2)) * a.pitchS[1] + win;
            }
          }
          ipyr_phase2<HL>(ph, [&](auto r0_tag) {
            rows2(r0_tag, src, [&](auto j_tag, auto slot_tag) {
              constexpr int j = decltype(j_tag)::value, slot = decltype(slot_tag)::value;
              const int p = r2 + j - (HL - 1);
              if (p >= ra[2] && wr) {
#pragma unroll
                for (int rr = 0; rr < 2; ++rr) {
                  const f2 (&q)[2] = rr ? acc.r1[slot] : acc.r0[slot];
                  *reinterpret_cast<f4*>(smem + ring1_off + ((2 * p + rr) & (kIpRing1 - 1)) * a.pitchR[0] + wq) = (f4){q[0].x, q[0].y, q[1].x, q[1].y};
                }
              }
            });
          });
          ph = (ph + 2) % HL;
        }
        if (t >= TW) eb = eb + 1 == a.nbuf ? 0 : eb + 1;
      }
      return;
    }
  }

  // ---- level 3: one coefficient row per step (at odd sub-steps; bands aa, ad from entry t - 1, da, dd from entry t) -----------------
  if constexpr (NLEV >= 3) {
    const int nq = (a.Mw[1] + 3) >> 2;
    const int Gr = min(G, nq - 1);
    const bool wr = G < nq;
    const uint32_t win = 8u * (uint32_t)Gr;
    const uint32_t wq = 16u * (uint32_t)Gr;
    int eb = 0, ph = 0;
#pragma unroll 1
    for (int t = 0; t < nsub; ++t) {
      __syncthreads();
      const int r3 = ra[3] + (t >> 1);
      if ((t & 1) && r3 <= rb[3] && !(MIFWT_DBG(a) & 4)) {
        const int ebp = eb == 0 ? a.nbuf - 1 : eb - 1;
        const unsigned char* e0 = (t - 1 < TW ? smem + a.wu3_off + 2 * (t - 1) * a.pitchS[2] : smem + stage_off + ebp * a.entry_bytes + a.offL[2]) + win;
        const unsigned char* e1 = (t < TW ? smem + a.wu3_off + 2 * t * a.pitchS[2] : smem + stage_off + eb * a.entry_bytes + a.offL[2]) + win;
        ipyr_phase1<HL>(ph, [&](auto r_tag) {
          constexpr int R = decltype(r_tag)::value;
          f2 w[4][NW];
#pragma unroll
          for (int m = 0; m < NW; ++m) {
            w[0][m] = *reinterpret_cast<const f2*>(e0 + 8 * m);
            w[1][m] = *reinterpret_cast<const f2*>(e0 + a.pitchS[2] + 8 * m);
            w[2][m] = *reinterpret_cast<const f2*>(e1 + 8 * m);
            w[3][m] = *reinterpret_cast<const f2*>(e1 + a.pitchS[2] + 8 * m);
          }
          f2 vl[2], vh[2];
          ipyr_hsyn<L>(tlo, thi, w[0], w[1], vl);
          ipyr_hsyn<L>(tlo, thi, w[2], w[3], vh);
          acc.template feed<R>(tlo, thi, vl, vh);
          constexpr int slot = IpAcc<L>::done(R);
          const int p = r3 - (HL - 1);
          if (p >= ra[3] && wr) {
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
              const f2 (&q)[2] = rr ? acc.r1[slot] : acc.r0[slot];
              *reinterpret_cast<f4*>(smem + ring2_off + ((2 * p + rr) & (kIpRing2 - 1)) * a.pitchR[1] + wq) = (f4){q[0].x, q[0].y, q[1].x, q[1].y};
            }
          }
        });
        ph = ph + 1 == HL ? 0 : ph + 1;
      }
      if (t >= TW) eb = eb + 1 == a.nbuf ? 0 : eb + 1;
    }
  }
}

// ---- host side ------------------------------------------------------------------------------------------------------------
struct IPyrPlan {
  int nseg, seg_rows, nbuf, pitchS[3], pitchR[2], offL[3], entry_bytes, nS[3], lds, tw, wu2_off, wu3_off;
};

// d[0] = the coarsest level, d[nlev - 1] = the finest (the order of mifwt_dwt2_inv_pyramid); level l (1 = finest) = d[nlev - l]
static bool ipyr_plan(int nlev, const mifwt_level_desc* const* d, IPyrPlan* p) {
  const int L = d[0]->filt_len;
  const mifwt_level_desc* fin = d[nlev - 1];
  const int H = (int)fin->sig_extent[0];
  int nwaves = 0;
  for (int l = 1; l <= nlev; ++l) {
    const mifwt_level_desc* dl = d[nlev - l];
    const int nw = (int)dl->sig_extent[1];  // output columns of the level
    p->nS[l - 1] = ((nw + 3) / 4 + 63) / 64;
    nwaves += p->nS[l - 1];
    p->pitchS[l - 1] = (((int)dl->coef_extent[1] + 8 + 3) & ~3) * 4;
  }
  for (int l = nlev; l < 3; ++l) p->nS[l] = 0, p->pitchS[l] = 0;
  if (nwaves > kIpFirstLoader) return false;
  p->pitchR[0] = nlev >= 2 ? (((int)d[nlev - 1]->coef_extent[1] + 12 + 3) & ~3) * 4 : 0;
  p->pitchR[1] = nlev >= 3 ? (((int)d[nlev - 2]->coef_extent[1] + 12 + 3) & ~3) * 4 : 0;
  int off = 0;
  for (int l = 1; l <= nlev; ++l) {
    p->offL[l - 1] = off;
    const int nb = l == nlev ? 4 : 3;
    off += (l == 1 ? 2 * nb : (l == 2 ? nb : 2)) * p->pitchS[l - 1];
  }
  for (int l = nlev; l < 3; ++l) p->offL[l] = 0;
  p->entry_bytes = (off + 15) & ~15;
  const int rings = kIpRing1 * p->pitchR[0] + kIpRing2 * p->pitchR[1];
  // the lanes of the last wave of a level read their windows from the position of the level's last lane, plus L/2 + 1 samples:
  // some slack behind the last ring row
  const int slack = 256;
  // requests in flight per loader: (nbuf - 2) sub-steps of at most ceil(items / 3) rows each
  int per_max = 0;
  {
    int per[3] = {0, 0, 0}, idx = 0;
    for (int l = 1; l <= nlev; ++l) {
      const int nb = l == nlev ? 4 : 3;
      const int items = l == 1 ? 2 * nb : (l == 2 ? nb : 2);
      const int nch = ((int)d[nlev - l]->coef_extent[1] + 255) / 256;
      for (int i = 0; i < items; ++i, ++idx) per[idx % 3] += nch;
    }
    per_max = std::max(per[0], std::max(per[1], per[2]));
  }
  // fast warm-up (MIFWT_OPT_DEBUG bit 8 switches it off): slots of their own for the level-2 / level-3 rows of the entries before
  // level 1 starts, if they fit beside at least four ring entries
  const int t1 = ipyr_t1(L, nlev), d2 = ipyr_d2(L, nlev);
  const int wu3 = nlev >= 3 ? t1 * 2 * p->pitchS[2] : 0;
  const int wu2 = nlev >= 2 ? (t1 - 2 * d2) * (nlev == 2 ? 4 : 3) * p->pitchS[1] : 0;
  int wu = (g_options[MIFWT_OPT_DEBUG] & 8) ? 0 : wu2 + wu3;
  if (wu && 4 * p->entry_bytes + rings + slack + wu > 160 * 1024) wu = 0;
  p->nbuf = g_options[MIFWT_OPT_PREFETCH_PAIRS] > 3 ? std::min(8, g_options[MIFWT_OPT_PREFETCH_PAIRS]) : 5;
  while (p->nbuf > 4 && (p->nbuf * p->entry_bytes + rings + slack + wu > 160 * 1024 || (p->nbuf - 2) * per_max > 63)) --p->nbuf;
  p->lds = p->nbuf * p->entry_bytes + rings + slack + wu;
  if (p->lds > 160 * 1024 || (p->nbuf - 2) * per_max > 63) return false;
  p->tw = wu ? t1 : 0;
  p->wu3_off = p->nbuf * p->entry_bytes + rings + slack;
  p->wu2_off = p->wu3_off + wu3;
  if (p->lds < 82 * 1024) p->lds = 82 * 1024;  // one workgroup per CU
  int dev = 0, ncu = 256;
  if (hipGetDevice(&dev) == hipSuccess) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ncu = v;
  }
  const int64_t batch = d[0]->batch;
  int nseg = g_options[MIFWT_OPT_PAIR_ROWS] > 0 ? (H + g_options[MIFWT_OPT_PAIR_ROWS] - 1) / g_options[MIFWT_OPT_PAIR_ROWS]
                                                  : (int)((ncu + batch / 2) / (batch > 0 ? batch : 1));
  const int max_seg = H / 32 > 0 ? H / 32 : 1;
  nseg = nseg < 1 ? 1 : (nseg > max_seg ? max_seg : nseg);
  p->seg_rows = (((H + nseg - 1) / nseg) + 7) & ~7;
  p->nseg = (H + p->seg_rows - 1) / p->seg_rows;
  return true;
}

bool dwt2_inv_pyr_supported(int nlev, const mifwt_level_desc* const* d) {
  if (nlev < 1 || nlev > 3 || g_options[MIFWT_OPT_FORCE_GENERIC] || g_options[MIFWT_OPT_PAIR_MODE] == 2 || g_options[MIFWT_OPT_PYRAMID_MODE] == 2)
    return false;
  const int L = d[0]->filt_len;
  if (L < 2 || L > 10 || (L & 1)) return false;
  const int64_t lim = int64_t(1) << 29;  // byte offsets inside one image stay below 2^31
  for (int i = 0; i < nlev; ++i) {
    const mifwt_level_desc* dl = d[i];
    if (dl->ndim != 2 || dl->dtype != MIFWT_F32 || dl->filt_len != L || dl->batch != d[0]->batch || dl->batch < 1) return false;
    if (dl->detail_stride[2] != 1 || dl->detail_stride[0] < 0 || dl->detail_stride[1] < 0) return false;
    if ((dl->coef_extent[0] - 1) * dl->detail_stride[1] + dl->coef_extent[1] >= lim) return false;
    for (int ax = 0; ax < 2; ++ax) {
      const int64_t out = dl->sig_extent[ax];  // (cropped) output extent of the level = the next level's coefficient extent
      if (out < 1 || out > 2 * dl->coef_extent[ax] - L + 2) return false;
      if (i + 1 < nlev && d[i + 1]->coef_extent[ax] != out) return false;
    }
  }
  if (d[0]->approx_stride[2] != 1 || d[0]->approx_stride[0] < 0 || d[0]->approx_stride[1] < 0) return false;
  if ((d[0]->coef_extent[0] - 1) * d[0]->approx_stride[1] + d[0]->coef_extent[1] >= lim) return false;
  const mifwt_level_desc* fin = d[nlev - 1];
  if (fin->sig_stride[2] != 1 || fin->sig_stride[0] < 0 || fin->sig_stride[1] < 0) return false;
  if (fin->sig_extent[0] * fin->sig_stride[1] >= lim) return false;
  if (fin->sig_extent[0] < 32) return false;
  IPyrPlan p;
  if (!ipyr_plan(nlev, d, &p)) return false;
  // where it pays (MIFWT_OPT_PYRAMID_MODE 1 overrides): planes a workgroup streams as whole rows of a useful length — from 384
  // columns on (256 x 384^2 db4: 80 against 96 us for the per-level / two-level launches; 256^2: 53 against 49; 192^2: 69 against 44,
  // profiles/r03l_ipyr_where.txt)
  if (g_options[MIFWT_OPT_PYRAMID_MODE] != 1 && fin->sig_extent[1] < 384) return false;
  return true;
}

template <int L, int NLEV>
static int launch_ipyr(const mifwt_level_desc* const* d, const void* approx, const void* const* const* details, void* y, const double* lo,
                       const double* hi, hipStream_t stream) {
  IPyrPlan p;
  if (!ipyr_plan(NLEV, d, &p)) return MIFWT_ERR_UNSUPPORTED;
  IPyrArgs<L, NLEV> a;
  for (int l = 1; l <= NLEV; ++l) {
    const mifwt_level_desc* dl = d[NLEV - l];
    a.band[l - 1][0] = l == NLEV ? static_cast<const float*>(approx) : nullptr;
    for (int b = 0; b < 3; ++b) a.band[l - 1][1 + b] = static_cast<const float*>(details[NLEV - l][b]);
    a.bs_b[l - 1][0] = d[0]->approx_stride[0];
    a.bs_h[l - 1][0] = (int)d[0]->approx_stride[1];
    a.bs_b[l - 1][1] = dl->detail_stride[0];
    a.bs_h[l - 1][1] = (int)dl->detail_stride[1];
    a.Mh[l - 1] = (int)dl->coef_extent[0];
    a.Mw[l - 1] = (int)dl->coef_extent[1];
    a.pitchS[l - 1] = p.pitchS[l - 1];
    a.offL[l - 1] = p.offL[l - 1];
    a.nS[l - 1] = p.nS[l - 1];
  }
  const mifwt_level_desc* fin = d[NLEV - 1];
  a.y = static_cast<float*>(y);
  a.ys_b = fin->sig_stride[0];
  a.ys_h = (int)fin->sig_stride[1];
  a.H = (int)fin->sig_extent[0];
  a.W = (int)fin->sig_extent[1];
  a.nseg = p.nseg;
  a.seg_rows = p.seg_rows;
  a.nbuf = p.nbuf;
  a.pitchR[0] = p.pitchR[0];
  a.pitchR[1] = p.pitchR[1];
  a.entry_bytes = p.entry_bytes;
  a.tw = p.tw;
  a.wu2_off = p.wu2_off;
  a.wu3_off = p.wu3_off;
  a.dbg = g_options[MIFWT_OPT_DEBUG];
  for (int j = 0; j < L / 2; ++j) {
    a.tlo[j] = (f2){(float)lo[2 * j], (float)lo[2 * j + 1]};
    a.thi[j] = (f2){(float)hi[2 * j], (float)hi[2 * j + 1]};
  }
  a.dt = dev_tap_arg(L);
  const int64_t nwg = d[0]->batch * p.nseg;
  if (nwg > INT32_MAX / 8) return MIFWT_ERR_UNSUPPORTED;
  static DynLdsOnce lds_once;
  if (a.dt.lo) {  // device-resident taps: the one-level form only (a learnable bank goes level by level)
    if constexpr (NLEV == 1) {
      static DynLdsOnce lds_once_dt;
      if (!lds_once_dt.ensure(reinterpret_cast<const void*>(&idwt2_pyr_kernel<L, 1, true>), 160 * 1024)) return MIFWT_ERR_LAUNCH;
      hipLaunchKernelGGL((idwt2_pyr_kernel<L, 1, true>), dim3((unsigned)nwg), dim3(64 * kIpWaves), p.lds, stream, a);
      return hipGetLastError() == hipSuccess ? MIFWT_OK : MIFWT_ERR_LAUNCH;
    }
    return MIFWT_ERR_UNSUPPORTED;
  }
  if (!lds_once.ensure(reinterpret_cast<const void*>(&idwt2_pyr_kernel<L, NLEV, false>), 160 * 1024)) return MIFWT_ERR_LAUNCH;
  hipLaunchKernelGGL((idwt2_pyr_kernel<L, NLEV, false>), dim3((unsigned)nwg), dim3(64 * kIpWaves), p.lds, stream, a);
  return hipGetLastError() == hipSuccess ? MIFWT_OK : MIFWT_ERR_LAUNCH;
}

template <int L>
static int launch_ipyr_l(int nlev, const mifwt_level_desc* const* d, const void* approx, const void* const* const* details, void* y,
                         const double* lo, const double* hi, hipStream_t stream) {
  switch (nlev) {
    case 1: return launch_ipyr<L, 1>(d, approx, details, y, lo, hi, stream);
    case 2: return launch_ipyr<L, 2>(d, approx, details, y, lo, hi, stream);
    case 3: return launch_ipyr<L, 3>(d, approx, details, y, lo, hi, stream);
    default: return MIFWT_ERR_UNSUPPORTED;
  }
}

int dwt2_inv_pyr(int nlev, const mifwt_level_desc* const* d, const void* approx, const void* const* const* details, void* y,
                 const double* lo, const double* hi, hipStream_t stream) {
  if (!dwt2_inv_pyr_supported(nlev, d)) return MIFWT_ERR_UNSUPPORTED;
  switch (d[0]->filt_len) {
    case 2: return launch_ipyr_l<2>(nlev, d, approx, details, y, lo, hi, stream);
    case 4: return launch_ipyr_l<4>(nlev, d, approx, details, y, lo, hi, stream);
    case 6: return launch_ipyr_l<6>(nlev, d, approx, details, y, lo, hi, stream);
    case 8: return launch_ipyr_l<8>(nlev, d, approx, details, y, lo, hi, stream);
    case 10: return launch_ipyr_l<10>(nlev, d, approx, details, y, lo, hi, stream);
    default: return MIFWT_ERR_UNSUPPORTED;
  }
}

}  // namespace mifwt
