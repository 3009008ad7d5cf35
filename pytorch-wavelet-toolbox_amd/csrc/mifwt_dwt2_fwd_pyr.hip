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
//   * boundary extension: level-0 pad columns are copied inside LDS after the rows land, ring pad columns after a ring
//     row is written (edge strips only, one read + one write per step); out-of-plane rows in zero mode are zero rows.
// Results agree with the per-level kernels to rounding (different summation order), with the fp64 oracle within 1e-6.
// f32, even L <= 8, modes zero / constant / reflect / symmetric (periodic needs the far side of the plane).
// Algorithmic traffic: 4 B H W read + 4 B (3 H1 W1 [+ 3 H2 W2] + 4 H_N W_N) written.
#include <type_traits>

#include "mifwt_stream.h"

namespace mifwt {

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
constexpr int pyr_lag2(int L) { const int hp = L / 2, hl = L - 2; const int m = hp - 1 > hl + hp - 4 ? hp - 1 : hl + hp - 4; return m <= 0 ? 0 : (m + 3) / 4; }
constexpr int pyr_lag3(int L) { const int hp = L / 2, hl = L - 2; const int m = hp - 1 > hl + hp - 2 ? hp - 1 : hl + hp - 2; return pyr_lag2(L) + (m <= 0 ? 0 : (m + 1) / 2); }

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
  f2 tap[L];  // (dec_lo[m], dec_hi[m])
};

typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t pyr_rsrc(const void* p, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
// 64 lanes x 16 B -> LDS [lds_addr + 16 lane); global address = resource base + voff (per lane) + soff; non-temporal
// (M0 carries the LDS address and belongs to the compiler: saved and restored inside the statement)
__device__ __forceinline__ void pyr_dma(uint32_t voff, rsrc_t rsrc, uint32_t soff, uint32_t lds_addr) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %4 offen nt lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(voff), "s"(rsrc), "s"(lds_addr), "s"(soff)
               : "memory");
}
__device__ __forceinline__ void pyr_store1(float v, rsrc_t rsrc, uint32_t voff, uint32_t soff) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v), rsrc, voff, soff, 0);
}
template <int N>
__device__ __forceinline__ void pyr_wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// rolling vertical pass: the L/2 outputs in flight of NC columns; lo = (aa, da), hi = (ad, dd) per column
template <int L, int NC>
struct PyrAcc {
  f2 lo[L / 2][NC], hi[L / 2][NC];
  // one row of horizontally filtered samples hv[c] = (h_lo, h_hi); PH = 0: first row of a pair, 1: second
  template <int PH>
  __device__ __forceinline__ void feed(const f2 (&tap)[L], const f2 (&hv)[NC]) {
#pragma unroll
    for (int q = 0; q < L / 2; ++q) {
      const int m = L - 1 - 2 * q - PH;
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        if (q == 0 && PH == 0) {
          lo[0][c] = pkmul_lo(tap[m], hv[c]);
          hi[0][c] = pkmul_hi(tap[m], hv[c]);
        } else {
          pkfma_lo(lo[q][c], tap[m], hv[c]);
          pkfma_hi(hi[q][c], tap[m], hv[c]);
        }
      }
    }
  }
  // after the second row of a pair the oldest output is complete (read it at index L/2 - 1 first)
  __device__ __forceinline__ void shift() {
#pragma unroll
    for (int q = L / 2 - 1; q > 0; --q)
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        lo[q][c] = lo[q - 1][c];
        hi[q][c] = hi[q - 1][c];
      }
  }
};

template <int L, int NLEV>
__global__ void __launch_bounds__(64 * (kPyrNW + 1), 2) dwt2_fwd_pyr_kernel(const PyrArgs<L, NLEV> a) {
  constexpr int HL = L - 2, HP = L / 2;
  constexpr int D2 = pyr_lag2(L), D3 = pyr_lag3(L);
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
  if (wave == kPyrNW) {
    const uint32_t img_bytes = ((uint32_t)(a.H[0] - 1) * (uint32_t)a.xs_h + (uint32_t)a.W[0]) * 4u;
    const rsrc_t xr = pyr_rsrc(a.x + (int64_t)img * a.xs_b, img_bytes);
    const uint32_t row_bytes = (uint32_t)a.xs_h * 4u;
    uint32_t voff[kPyrNW];
#pragma unroll
    for (int w = 0; w < kPyrNW; ++w) {
      const int k = grp * kPyrNW + w;
      // level-1 columns the strip computes start at cA1; its staged row starts at level-0 column g0 (16-byte aligned)
      int cA = k == 0 ? 0 : a.cpw0 + (k - 1) * a.cpw;
#pragma unroll
      for (int l = NLEV - 1; l >= 1; --l) cA = max(0, 2 * cA - HL);
      const int g0 = max(0, 2 * cA - HL) & ~3;
      const int c = g0 + 4 * lane;
      voff[w] = (k < a.nstrips && c < a.W[0]) ? 4u * (uint32_t)c : kPyrOob;
    }
    auto issue = [&](int t) {
      const uint32_t buf = (uint32_t)(t & (kPyrNBuf - 1)) * (kPyrSub * kPyrSlotB) + 64u + kPyrPad * 4u;
#pragma unroll
      for (int kk = 0; kk < kPyrSub; ++kk) {
        const int e = E0 + kPyrSub * t + kk;
        const bool dead = e >= e0_end || (zero_mode && (unsigned)e >= (unsigned)a.H[0]);
        const uint32_t soff = dead ? 0u : (uint32_t)fold(e, a.H[0]) * row_bytes;
#pragma unroll
        for (int w = 0; w < kPyrNW; ++w)
          pyr_dma(dead ? kPyrOob : voff[w], xr, soff, (uint32_t)w * WB + buf + (uint32_t)kk * kPyrSlotB);
      }
    };
    constexpr int PER = kPyrSub * kPyrNW;  // DMA instructions per sub-step
    __syncthreads();  // the compute waves have initialised their LDS
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
  // compute waves
  const int strip = grp * kPyrNW + wave;
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
  const int g0 = max(0, 2 * cA[1] - HL) & ~3;

  unsigned char* const wbase = smem + 64 + wave * WB;
  unsigned char* const stage = wbase;
  unsigned char* const ring1 = wbase + kPyrStageB;
  unsigned char* const ring2 = ring1 + (kPyrRing + 1) * kPyrR1B;

  // ---- per-lane constants ------------------------------------------------------------------------------------------------
  // level 1: lane -> columns c1, c1 + 1; its window = staged floats [win1, win1 + L + 2)
  const int c1 = cA[1] + 2 * lane;
  const uint32_t win1 = 4u * (uint32_t)(kPyrPad - HL + 2 * cA[1] - g0 + 4 * lane);
  const uint32_t sv1_x2 = (c1 >= pA[1] && c1 + 1 < pB[1]) ? 4u * (uint32_t)c1 : kPyrOob;
  const uint32_t sv1_x1 = (c1 >= pA[1] && c1 + 1 == pB[1]) ? 4u * (uint32_t)c1 : kPyrOob;
  const bool ragged1 = ((pB[1] - cA[1]) & 1) != 0;
  // level-0 pad fill (strips at the plane's left / right edge): lane -> (row kk of the sub-step, pad column)
  uint32_t f0_src = 0, f0_dst = 0;
  bool f0_on = false;
  {
    constexpr int NP = 2 * HL + 1;
    const int kk = lane / NP, p = lane - kk * NP;
    const bool left = p < HL;
    const int e = left ? p - HL : a.W[0] + (p - HL);  // extended level-0 column
    const bool need = kk < kPyrSub && !zero_mode && (left ? cA[1] == 0 : cB[1] == a.W[1]);
    if (need) {
      f0_on = true;
      f0_src = (uint32_t)kk * kPyrSlotB + 4u * (uint32_t)(kPyrPad + fold(e, a.W[0]) - g0);
      f0_dst = (uint32_t)kk * kPyrSlotB + 4u * (uint32_t)(kPyrPad + e - g0);
    }
  }
  const bool f0_any = __builtin_amdgcn_readfirstlane(__builtin_amdgcn_ballot_w64(f0_on) != 0);

  // deeper levels: lane -> column c; window = ring floats [winN, winN + L) of the level above
  int c2 = 0, c3 = 0;
  uint32_t win2 = 0, win3 = 0, sv2 = kPyrOob, sv3 = kPyrOob;
  uint32_t f1_src = 0, f1_dst = 0, f2_src = 0, f2_dst = 0;
  bool f1_any = false, f2_any = false;
  if constexpr (NLEV >= 2) {
    c2 = cA[2] + lane;
    win2 = 4u * (uint32_t)(kPyrPad - HL + 2 * cA[2] - cA[1] + 2 * lane);
    sv2 = (c2 >= pA[2] && c2 < pB[2]) ? 4u * (uint32_t)c2 : kPyrOob;
    // ring-1 pad fill: lane -> (row j of the step's four, pad column)
    constexpr int NP = 2 * HL + 1;
    const int j = lane / NP, p = lane - j * NP;
    const bool left = p < HL;
    const int e = left ? p - HL : a.W[1] + (p - HL);
    const bool on = j < 4 && !zero_mode && (left ? cA[1] == 0 : cB[1] == a.W[1]);
    // idle lanes copy float 0 of the group's first row onto itself (never used: the pad area starts at kPyrPad - HL >= 2)
    f1_src = on ? (uint32_t)j * kPyrR1B + 4u * (uint32_t)(kPyrPad + fold(e, a.W[1]) - cA[1]) : 0u;
    f1_dst = on ? (uint32_t)j * kPyrR1B + 4u * (uint32_t)(kPyrPad + e - cA[1]) : 0u;
    f1_any = __builtin_amdgcn_readfirstlane(__builtin_amdgcn_ballot_w64(on) != 0);
  }
  if constexpr (NLEV >= 3) {
    c3 = cA[3] + lane;
    win3 = 4u * (uint32_t)(kPyrPad - HL + 2 * cA[3] - cA[2] + 2 * lane);
    sv3 = (c3 >= pA[3] && c3 < pB[3]) ? 4u * (uint32_t)c3 : kPyrOob;
    constexpr int NP = 2 * HL + 1;
    const int j = lane / NP, p = lane - j * NP;
    const bool left = p < HL;
    const int e = left ? p - HL : a.W[2] + (p - HL);
    const bool on = j < 2 && !zero_mode && (left ? cA[2] == 0 : cB[2] == a.W[2]);
    f2_src = on ? (uint32_t)j * kPyrR2B + 4u * (uint32_t)(kPyrPad + fold(e, a.W[2]) - cA[2]) : 0u;
    f2_dst = on ? (uint32_t)j * kPyrR2B + 4u * (uint32_t)(kPyrPad + e - cA[2]) : 0u;
    f2_any = __builtin_amdgcn_readfirstlane(__builtin_amdgcn_ballot_w64(on) != 0);
  }
  const uint32_t svN = NLEV == 1 ? 0u : (NLEV == 2 ? sv2 : sv3);  // approximation band: same lanes as the last level's details

  // ---- output resources ---------------------------------------------------------------------------------------------------
  rsrc_t dr[NLEV][3];
#pragma unroll
  for (int l = 0; l < NLEV; ++l) {
    const uint32_t bytes = ((uint32_t)(a.H[l + 1] - 1) * (uint32_t)a.ds_h[l] + (uint32_t)a.W[l + 1]) * 4u;
#pragma unroll
    for (int b = 0; b < 3; ++b) dr[l][b] = pyr_rsrc(a.det[l][b] + (int64_t)img * a.ds_b[l], bytes);
  }
  const rsrc_t ar = pyr_rsrc(a.approx + (int64_t)img * a.as_b, ((uint32_t)(a.H[NLEV] - 1) * (uint32_t)a.as_h + (uint32_t)a.W[NLEV]) * 4u);

  // ---- LDS initialisation: everything this wave may read before it is written (pads in zero mode, the zero rows) -------
  for (int i = lane; i < WB / 16; i += 64) reinterpret_cast<f4*>(wbase)[i] = (f4){0.f, 0.f, 0.f, 0.f};
  __syncthreads();  // ... before the loader's first row lands

  // ring slots: level-1 row i lives in slot (i + ro1) & 15, level-2 row i in (i + ro2) & 15
  const int ro1 = HP - 1 - rA[1];
  int ro2 = 0, E1 = 0, E2 = 0;
  if constexpr (NLEV >= 2) {
    ro2 = HP - 1 - rA[2];
    E1 = 2 * rA[2] - HL;
  }
  if constexpr (NLEV >= 3) E2 = 2 * rA[3] - HL;

  PyrAcc<L, 2> acc1;
  PyrAcc<L, 1> acc2, acc3;
#pragma unroll
  for (int q = 0; q < HP; ++q) {
    acc1.lo[q][0] = acc1.lo[q][1] = acc1.hi[q][0] = acc1.hi[q][1] = (f2){0.f, 0.f};
    acc2.lo[q][0] = acc2.hi[q][0] = acc3.lo[q][0] = acc3.hi[q][0] = (f2){0.f, 0.f};
  }

  // horizontal pass of one level-1 row: staged row -> (lo, hi) of the lane's two columns
  auto h1_row = [&](const unsigned char* row, f2 (&hv)[2]) {
    f2 w[HP + 1];
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
      if constexpr ((HP & 1) == 0) w[HP] = *reinterpret_cast<const f2*>(row + 8 * HP);
    }
#pragma unroll
    for (int c = 0; c < 2; ++c) {
#pragma unroll
      for (int k = 0; k < HP; ++k) {
        if (k == 0) hv[c] = pkmul_lo(a.tap[L - 1], w[c]);
        else pkfma_lo(hv[c], a.tap[L - 1 - 2 * k], w[c + k]);
        pkfma_hi(hv[c], a.tap[L - 2 - 2 * k], w[c + k]);
      }
    }
  };
  // horizontal pass of one deeper row: ring row of the level above -> (lo, hi) of the lane's column
  auto hN_row = [&](const unsigned char* row, f2 (&hv)[1]) {
#pragma unroll
    for (int k = 0; k < HP; ++k) {
      const f2 w = *reinterpret_cast<const f2*>(row + 8 * k);
      if (k == 0) hv[0] = pkmul_lo(a.tap[L - 1], w);
      else pkfma_lo(hv[0], a.tap[L - 1 - 2 * k], w);
      pkfma_hi(hv[0], a.tap[L - 2 - 2 * k], w);
    }
  };

  // ---- steps -------------------------------------------------------------------------------------------------------------
#pragma unroll 1
  for (int s = 0; s < nsteps; ++s) {
    // ===== level 1: two sub-steps of four level-0 rows = four row pairs = four level-1 rows =====
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      __syncthreads();  // the loader has seen this sub-step land
      if (s < nsteps1) {
        const int t = 2 * s + half;
        const unsigned char* sb = stage + (t & (kPyrNBuf - 1)) * (kPyrSub * kPyrSlotB);
        if (f0_any) {
          const float v = *reinterpret_cast<const float*>(sb + f0_src);
          wave_lds_fence();
          if (f0_on) *reinterpret_cast<float*>(const_cast<unsigned char*>(sb) + f0_dst) = v;
          wave_lds_fence();
        }
#pragma unroll
        for (int kk = 0; kk < kPyrSub; ++kk) {
          f2 hv[2];
          h1_row(sb + kk * kPyrSlotB + win1, hv);
          if ((kk & 1) == 0) {
            acc1.template feed<0>(a.tap, hv);
          } else {
            acc1.template feed<1>(a.tap, hv);
            const int j = 2 * half + (kk >> 1);          // pair of the step
            const int i = rA[1] + 4 * s + j - (HP - 1);  // the level-1 row it completes
            const f2 (&lo)[2] = acc1.lo[HP - 1];
            const f2 (&hi)[2] = acc1.hi[HP - 1];
            if constexpr (NLEV >= 2) {
              *reinterpret_cast<f2*>(ring1 + ((4 * s + j) & (kPyrRing - 1)) * kPyrR1B + 4 * (kPyrPad + 2 * lane)) = (f2){lo[0].x, lo[1].x};
            }
            if (i >= oA[1] && i < oB[1]) {
              const uint32_t so = (uint32_t)i * (uint32_t)a.ds_h[0] * 4u;
              __builtin_amdgcn_raw_buffer_store_b64((f2){hi[0].x, hi[1].x}, dr[0][0], sv1_x2, so, 0);
              __builtin_amdgcn_raw_buffer_store_b64((f2){lo[0].y, lo[1].y}, dr[0][1], sv1_x2, so, 0);
              __builtin_amdgcn_raw_buffer_store_b64((f2){hi[0].y, hi[1].y}, dr[0][2], sv1_x2, so, 0);
              if (ragged1) {
                pyr_store1(hi[0].x, dr[0][0], sv1_x1, so);
                pyr_store1(lo[0].y, dr[0][1], sv1_x1, so);
                pyr_store1(hi[0].y, dr[0][2], sv1_x1, so);
              }
              if constexpr (NLEV == 1) {
                const uint32_t sa = (uint32_t)i * (uint32_t)a.as_h * 4u;
                __builtin_amdgcn_raw_buffer_store_b64((f2){lo[0].x, lo[1].x}, ar, sv1_x2, sa, 0);
                if (ragged1) pyr_store1(lo[0].x, ar, sv1_x1, sa);
              }
            }
            acc1.shift();
          }
        }
      }
    }
    if constexpr (NLEV >= 2) {
      wave_lds_fence();
      if (f1_any && s < nsteps1) {  // extension columns of the four ring rows just written
        unsigned char* rb = ring1 + ((4 * s) & (kPyrRing - 1)) * kPyrR1B;
        const float v = *reinterpret_cast<const float*>(rb + f1_src);
        wave_lds_fence();
        *reinterpret_cast<float*>(rb + f1_dst) = v;
        wave_lds_fence();
      }
      // ===== level 2: four rows of the level-1 ring = two pairs = two level-2 rows =====
      if (s >= D2 && 2 * (s - D2) < npair2) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int e = E1 + 4 * (s - D2) + r;
          const bool dead = zero_mode && (unsigned)e >= (unsigned)a.H[1];
          const int slot = dead ? kPyrRing : ((fold(e, a.H[1]) + ro1) & (kPyrRing - 1));
          f2 hv[1];
          hN_row(ring1 + slot * kPyrR1B + win2, hv);
          if ((r & 1) == 0) {
            acc2.template feed<0>(a.tap, hv);
          } else {
            acc2.template feed<1>(a.tap, hv);
            const int p = 2 * (s - D2) + (r >> 1);
            const int i = rA[2] + p - (HP - 1);
            const f2 lo = acc2.lo[HP - 1][0], hi = acc2.hi[HP - 1][0];
            if constexpr (NLEV >= 3) *reinterpret_cast<float*>(ring2 + (p & (kPyrRing - 1)) * kPyrR2B + 4 * (kPyrPad + lane)) = lo.x;
            if (i >= oA[2] && i < oB[2]) {
              const uint32_t so = (uint32_t)i * (uint32_t)a.ds_h[1] * 4u;
              pyr_store1(hi.x, dr[1][0], sv2, so);
              pyr_store1(lo.y, dr[1][1], sv2, so);
              pyr_store1(hi.y, dr[1][2], sv2, so);
              if constexpr (NLEV == 2) pyr_store1(lo.x, ar, svN, (uint32_t)i * (uint32_t)a.as_h * 4u);
            }
            acc2.shift();
          }
        }
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
    if constexpr (NLEV >= 3) {
      // ===== level 3: two rows of the level-2 ring = one pair = one level-3 row =====
      if (s >= D3 && s - D3 < npair3) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          const int e = E2 + 2 * (s - D3) + r;
          const bool dead = zero_mode && (unsigned)e >= (unsigned)a.H[2];
          const int slot = dead ? kPyrRing : ((fold(e, a.H[2]) + ro2) & (kPyrRing - 1));
          f2 hv[1];
          hN_row(ring2 + slot * kPyrR2B + win3, hv);
          if (r == 0) {
            acc3.template feed<0>(a.tap, hv);
          } else {
            acc3.template feed<1>(a.tap, hv);
            const int i = rA[3] + (s - D3) - (HP - 1);
            const f2 lo = acc3.lo[HP - 1][0], hi = acc3.hi[HP - 1][0];
            if (i >= oA[3] && i < oB[3]) {
              const uint32_t so = (uint32_t)i * (uint32_t)a.ds_h[2] * 4u;
              pyr_store1(hi.x, dr[2][0], sv3, so);
              pyr_store1(lo.y, dr[2][1], sv3, so);
              pyr_store1(hi.y, dr[2][2], sv3, so);
              pyr_store1(lo.x, ar, svN, (uint32_t)i * (uint32_t)a.as_h * 4u);
            }
            acc3.shift();
          }
        }
      }
    }
  }
}

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
  if (nlev < 1 || nlev > 3 || g_options[MIFWT_OPT_PAIR_MODE] == 2) return false;
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
  hipLaunchKernelGGL((dwt2_fwd_pyr_kernel<L, NLEV>), dim3((unsigned)nwg), dim3(64 * (kPyrNW + 1)), pyr_lds_bytes(NLEV), stream, a);
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
