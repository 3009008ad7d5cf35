// mifwt_dwt2_fwd_roll.hip — TWO consecutive 2-D analysis levels in one launch, rolling column strips (gfx950).
//
// Same seam and same results as mifwt_dwt2_fwd_pair.hip (two trips of src/ptwt/conv_transform_2.py:142-149, the
// level-1 approximation never reaches HBM), different shape of the work.  The tile version recomputes, per tile, the
// 3 (L-2) level-0 halo rows under a level-2 tile: with 8 level-2 rows per tile that is 50 input rows loaded,
// staged and filtered for every 32 it owns.  Here a 256-thread workgroup owns a column strip (64 level-1 columns: 64 -
// (L-2) owned + L-2 halo) of one SEGMENT of level-2 rows and walks down it in steps of 8 level-2 rows = 16 level-1 rows
// = 32 level-0 rows; the horizontally filtered rows and the level-1 approximation live in two LDS windows whose
// last L-2 rows are copied to the top for the next step (fixed slots: no ring arithmetic), so inside a segment nothing is
// loaded or filtered twice:
//   per step   wave w:  its 8 prefetched level-0 rows -> h-window; request the NEXT step's rows (in flight
//                       during everything below); level-1 horizontal pass over its own rows, in place
//              barrier  level-1 vertical pass, 4 rows per wave from a register window: details -> HBM, approximation
//                       -> a-window; level-2 horizontal pass over the rows it just produced, in place
//              barrier  level-2 vertical pass, one row per half-wave -> the four level-2 bands to HBM.
// A segment starts with a prologue of 3 (L-2) level-0 rows that fills the rings.  The first segment of a plane is
// aligned to its top, all others to its bottom (the boundary extension of level 2 reads ACTUAL level-1 rows through
// the index map, and those must still be in the window: tests/test_roll_model.py models exactly this bookkeeping).
// Column handling (shifted window at the plane's edges, per-lane index map for the extension) is the tile version's.
// Bit-identical to the per-level kernels.  Algorithmic traffic: 4 B H W read + 4 B (3 H1 W1 + 4 H2 W2) written.
#include <type_traits>

#include "mifwt_dwt2_tile.h"

namespace mifwt {

template <int L>
struct Dwt2RollArgs {
  const float* x;
  float* d1[3];  // level-1 bands ad, da, dd
  float* o2[4];  // level-2 bands aa, ad, da, dd
  int64_t xs_b, d1s_b, a2s_b, d2s_b;  // image strides (elements)
  int xs_h, d1s_h, a2s_h, d2s_h;      // row strides (elements; one image spans < 2^31 elements)
  int H0, W0, H1, W1, H2, W2;
  int strips, nseg, seg;  // column strips per plane, row segments per plane, level-2 rows per segment (multiple of 8)
  FastDiv div_s, div_g;   // by strips, nseg
  int mode;
  f2 tap[L];  // (dec_lo[m], dec_hi[m])
};

constexpr int roll_lds_bytes(int L) { return ((32 + L - 2) * (128 + L - 2) + (16 + L - 2) * 64) * 4; }
constexpr int roll_occupancy(int L) {
  // 6 workgroups' windows fit; the prefetch registers of the longer instances need the 5-wave register budget
  const int n = (160 * 1024) / roll_lds_bytes(L), cap = L <= 2 ? 6 : 5;
  return n > cap ? cap : (n < 1 ? 1 : n);
}

template <int L>
__global__ void __launch_bounds__(256, roll_occupancy(L)) dwt2_fwd_roll_kernel(const Dwt2RollArgs<L> a) {
  constexpr int HL = L - 2;
  constexpr int C1 = 64;              // level-1 columns of a strip, halo included = lanes
  constexpr int T2C = (C1 - HL) / 2;  // level-2 columns of a strip
  constexpr int OC1 = 2 * T2C;        // level-1 columns a strip owns
  constexpr int C0 = 2 * C1 + HL;     // level-0 columns
  constexpr int XP = C0;              // pitch of the h-window (floats, even)
  constexpr int NQ = (C0 + 63) / 64;
  constexpr int S2 = 8, S1 = 16, S0 = 32;  // rows per step at levels 2 / 1 / 0
  constexpr int RH = S0 + HL;  // h-window: slot t < HL = level-0 row 4 j - HL + t (kept from the previous step), then the step's 32
  constexpr int RL = S1 + HL;  // a-window: slot t < HL = level-1 row 2 j - HL + t, then the step's 16
  constexpr int LP = 64;
  constexpr int PR0 = 3 * HL;         // prologue: level-0 rows
  constexpr int PW0 = (PR0 + 3) / 4;  //           per wave
  constexpr int PW1 = (HL + 3) / 4;   // prologue: level-1 rows per wave
  static_assert(T2C <= 32 && T2C >= 1 && PR0 <= RH, "window geometry");
  __shared__ __attribute__((aligned(16))) float hr[RH * XP];
  __shared__ __attribute__((aligned(16))) float lr[RL * LP + 8];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  uint32_t utc, usg;
  const int img = (int)a.div_g.divmod(a.div_s.divmod((uint32_t)bid, utc), usg);
  const int tc = (int)utc, sg = (int)usg;
  // level-2 rows [ja, jb) of this segment; segment 0 is top-aligned (its last step is masked beyond jb), the others
  // end exactly at jb
  const int jb = a.H2 - (a.nseg - 1 - sg) * a.seg;
  const int ja = sg > 0 ? jb - a.seg : 0;
  const int nsteps = (jb - ja + S2 - 1) / S2;
  const int own_lo = 2 * ja, own_hi = min(2 * jb, a.H1);  // level-1 rows whose details this segment writes

  const int k2_0 = tc * T2C;
  const int s1c = min(max(2 * k2_0 - HL, 0), a.W1 - C1);
  const int c_first = 2 * s1c - HL;
  const bool zero_mode = a.mode == MIFWT_MODE_ZERO;
  Fold1 fold;  // mifwt_stream.h
  fold.set(a.mode);
  __builtin_assume(wave >= 0 && wave < 4);

  // ---- level-0 column offsets, once per workgroup ---------------------------------------------------------------------
  const uint32_t img_bytes = ((uint32_t)(a.H0 - 1) * (uint32_t)a.xs_h + (uint32_t)a.W0) * 4u;
  const __amdgpu_buffer_rsrc_t xrsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x + (int64_t)img * a.xs_b), 0, img_bytes, 0x00020000);
  constexpr uint32_t kOob = 0x80000000u;  // >= num_records: the load returns 0 without a memory request
  const uint32_t row_bytes = (uint32_t)a.xs_h * 4u;
  uint32_t coff[NQ];  // byte offset of this lane's column in load q; lanes beyond the window's C0 columns request nothing
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int c = c_first + lane + 64 * q;
    const bool dead = lane + 64 * q >= C0 || (zero_mode && (unsigned)c >= (unsigned)a.W0);
    coff[q] = dead ? kOob : 4u * (uint32_t)fold(c, a.W0);
  }
  constexpr bool kTail = (C0 & 63) != 0;  // the last load covers only C0 - 64 (NQ - 1) columns
  const bool tail_lane = lane + 64 * (NQ - 1) < C0;

  // level-0 rows r_first + wave + 4 i (i < N, row < r_end) of the extended plane -> registers.  Rows the level-1 plane
  // does not need (above -HL: segment 0's prologue; below 2 H1 - 1: the bottom-aligned last step) request nothing.
  const int r_valid_hi = 2 * a.H1;
  auto request = [&](auto n_tag, float (&v)[decltype(n_tag)::value][NQ], int r_first, int r_end) {
    constexpr int N = decltype(n_tag)::value;
    if (r_first >= 0 && r_first + 4 * N <= min(a.H0, r_end)) {  // interior: no map, one scalar add per row
      uint32_t soff = (uint32_t)(r_first + wave) * row_bytes;
#pragma unroll
      for (int i = 0; i < N; ++i) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) v[i][q] = tile_load<float>(xrsrc, coff[q], soff);
        soff += 4u * row_bytes;
      }
    } else {
#pragma unroll
      for (int i = 0; i < N; ++i) {
        const int r = r_first + wave + 4 * i;
        const bool dead = r >= r_end || r < -HL || r >= r_valid_hi || (zero_mode && (unsigned)r >= (unsigned)a.H0);
        const uint32_t soff = __builtin_amdgcn_readfirstlane(dead ? 0u : (uint32_t)fold(r, a.H0) * row_bytes);
        // a dead row is requested at per-lane offsets beyond num_records: nothing is fetched, zeros come back
#pragma unroll
        for (int q = 0; q < NQ; ++q) v[i][q] = tile_load<float>(xrsrc, dead ? kOob : coff[q], soff);
      }
    }
  };
  // registers of N rows -> h-window slots slot(i); the partial last load is stored under one exec mask for all rows.
  // nrows_tag: rows that exist (compile time; when it equals 4 N every wave owns N rows and no row test is emitted —
  // the compiler cannot know that wave < 4)
  auto stage = [&](auto n_tag, auto nrows_tag, const float (&v)[decltype(n_tag)::value][NQ], auto slot) {
    constexpr int N = decltype(n_tag)::value, NROWS = decltype(nrows_tag)::value;
    constexpr bool kAll = NROWS == 4 * N;
    constexpr int NF = kTail ? NQ - 1 : NQ;
#pragma unroll
    for (int i = 0; i < N; ++i)
      if (kAll || wave + 4 * i < NROWS) {
#pragma unroll
        for (int q = 0; q < NF; ++q) hr[slot(i) * XP + lane + 64 * q] = v[i][q];
      }
    if constexpr (kTail) {
      if (tail_lane) {
#pragma unroll
        for (int i = 0; i < N; ++i)
          if (kAll || wave + 4 * i < NROWS) hr[slot(i) * XP + lane + 64 * (NQ - 1)] = v[i][NQ - 1];
      }
    }
  };

  // level-1 horizontal pass over h-window slots s0 and s1, in place (two rows at once: two independent accumulation
  // chains, back-to-back dependent v_pk_fma_f32 cost a wait state each)
  auto h1_rows = [&](int s0, int s1) {
    const f2* row0 = reinterpret_cast<const f2*>(&hr[s0 * XP + 2 * lane]);
    const f2* row1 = reinterpret_cast<const f2*>(&hr[s1 * XP + 2 * lane]);
    f2 acc0, acc1;
#pragma unroll
    for (int p = 0; p < L / 2; ++p) {
      const f2 x0 = row0[p], x1 = row1[p];
      if (p == 0) {
        acc0 = pkmul_lo(a.tap[L - 1], x0);
        acc1 = pkmul_lo(a.tap[L - 1], x1);
      } else {
        pkfma_lo(acc0, a.tap[L - 1 - 2 * p], x0);
        pkfma_lo(acc1, a.tap[L - 1 - 2 * p], x1);
      }
      pkfma_hi(acc0, a.tap[L - 2 - 2 * p], x0);
      pkfma_hi(acc1, a.tap[L - 2 - 2 * p], x1);
    }
    wave_lds_fence();
    *reinterpret_cast<f2*>(&hr[s0 * XP + 2 * lane]) = acc0;
    *reinterpret_cast<f2*>(&hr[s1 * XP + 2 * lane]) = acc1;
  };
  auto h1_row = [&](int s) { h1_rows(s, s); };

  // level-2 column map (edge strips only), once per workgroup
  const int half = lane >> 5, kk = lane & 31;
  const int k2 = k2_0 + kk;
  const bool col_live = kk < T2C && k2 < a.W2;
  const bool cols_in2 = 2 * k2_0 - HL >= 0 && 2 * k2_0 + OC1 <= a.W1;  // then s1c == 2 k2_0 - HL, no extension
  int cidx[L];
#pragma unroll
  for (int p = 0; p < L; ++p) {
    const int e = 2 * k2 - HL + p;
    const bool dead = !col_live || (zero_mode && (unsigned)e >= (unsigned)a.W1);
    cidx[p] = dead ? -1 : fold(e, a.W1) - s1c;
  }
  // level-2 horizontal pass over a-window slot `s` (lanes 0-31 and 32-63 may be given different slots), in place
  auto h2_row = [&](int s) {
    f2 acc;
    if (cols_in2) {
      const f2* row = reinterpret_cast<const f2*>(&lr[s * LP + 2 * kk]);
#pragma unroll
      for (int p = 0; p < L / 2; ++p) {
        const f2 xx = row[p];
        if (p == 0) {
          acc = pkmul_lo(a.tap[L - 1], xx);
        } else {
          pkfma_lo(acc, a.tap[L - 1 - 2 * p], xx);
        }
        pkfma_hi(acc, a.tap[L - 2 - 2 * p], xx);
      }
    } else {
#pragma unroll
      for (int p = 0; p < L / 2; ++p) {
        f2 xx;
        xx.x = lr[s * LP + max(cidx[2 * p], 0)];
        xx.y = lr[s * LP + max(cidx[2 * p + 1], 0)];
        xx.x = cidx[2 * p] >= 0 ? xx.x : 0.0f;
        xx.y = cidx[2 * p + 1] >= 0 ? xx.y : 0.0f;
        if (p == 0) {
          acc = pkmul_lo(a.tap[L - 1], xx);
        } else {
          pkfma_lo(acc, a.tap[L - 1 - 2 * p], xx);
        }
        pkfma_hi(acc, a.tap[L - 2 - 2 * p], xx);
      }
    }
    wave_lds_fence();
    // lanes beyond the strip's T2C columns store too: columns 2 T2C .. 63 of the row are dead once it has been read
    *reinterpret_cast<f2*>(&lr[s * LP + 2 * kk]) = acc;
  };

  const int m1c = s1c + lane;
  const bool own_c = m1c >= 2 * k2_0 && m1c < 2 * k2_0 + OC1;
  // band bases stay on the scalar unit; lanes add 32-bit element offsets
  float* const d1b0 = a.d1[0] + (int64_t)img * a.d1s_b;
  float* const d1b1 = a.d1[1] + (int64_t)img * a.d1s_b;
  float* const d1b2 = a.d1[2] + (int64_t)img * a.d1s_b;
  float* const o2b0 = a.o2[0] + (int64_t)img * a.a2s_b;
  float* const o2b1 = a.o2[1] + (int64_t)img * a.d2s_b;
  float* const o2b2 = a.o2[2] + (int64_t)img * a.d2s_b;
  float* const o2b3 = a.o2[3] + (int64_t)img * a.d2s_b;

  // ---- prologue: level-0 rows [4 ja - 3 HL, 4 ja): the last HL of them -> h-window slots [0, HL) (where step 0 expects the
  // rows kept from "the previous step"), the first 2 HL -> slots [HL, 3 HL); level-1 rows [2 ja - HL, 2 ja) -> a-window
  // slots [0, HL).  (For segment 0 those level-1 rows lie above the plane and are never read.)
  float pv[S0 / 4][NQ];
  if constexpr (HL > 0) {
    float pp[PW0][NQ];
    request(std::integral_constant<int, PW0>{}, pp, 4 * ja - PR0, 4 * ja);
    request(std::integral_constant<int, S0 / 4>{}, pv, 4 * ja, 4 * ja + S0);
    auto pslot = [&](int i) {
      const int q = wave + 4 * i;  // prologue row index
      return q < 2 * HL ? q + HL : q - 2 * HL;
    };
    stage(std::integral_constant<int, PW0>{}, std::integral_constant<int, PR0>{}, pp, pslot);
    wave_lds_fence();
#pragma unroll
    for (int i = 0; i < PW0; ++i)
      if (wave + 4 * i < PR0) h1_row(pslot(i));
    __syncthreads();
#pragma unroll
    for (int i = 0; i < PW1; ++i) {
      const int il = wave + 4 * i;  // level-1 row 2 ja - HL + il -> a-window slot il; its h-rows: prologue rows 2 il .. 2 il + L - 1
      if (il < HL) {
        float aa;
#pragma unroll
        for (int m = 0; m < L; ++m) {
          const int q = 2 * il + (L - 1) - m;
          const int s = q < 2 * HL ? q + HL : q - 2 * HL;
          const float hv = hr[s * XP + 2 * lane];
          aa = m == 0 ? a.tap[0].x * hv : __builtin_fmaf(a.tap[m].x, hv, aa);  // the low half of the packed vertical pass
        }
        lr[il * LP + lane] = aa;
        wave_lds_fence();
        h2_row(il);  // both half-waves compute and store the same row
      }
    }
    __syncthreads();  // the prologue's first 2 HL h-rows are dead: step 0 overwrites their slots
  } else {
    request(std::integral_constant<int, S0 / 4>{}, pv, 4 * ja, 4 * ja + S0);
  }

  // ---- steps ------------------------------------------------------------------------------------------------------------
#pragma unroll 1
  for (int st = 0; st < nsteps; ++st) {
    const int j = ja + S2 * st;
    // this wave's 8 level-0 rows -> slots HL + wave + 4 i, then their horizontal pass (same wave: DS order suffices)
    stage(std::integral_constant<int, S0 / 4>{}, std::integral_constant<int, S0>{}, pv, [&](int i) { return HL + wave + 4 * i; });
    wave_lds_fence();
    if (st + 1 < nsteps) request(std::integral_constant<int, S0 / 4>{}, pv, 4 * (j + S2), 4 * (j + S2) + S0);
#pragma unroll
    for (int i = 0; i < S0 / 4; i += 2) h1_rows(HL + wave + 4 * i, HL + wave + 4 * i + 4);
    __syncthreads();

    // level-1 vertical pass: wave w -> level-1 rows 2 j + 4 w + i (i < 4) from h-window slots 8 w + t, t < HL + 8
    {
      if (st > 0) {
        // keep the last HL rows of the previous step's a-window: slot S1 + t -> slot t, by the wave that is about to
        // overwrite slot S1 + t (every wave finished reading the old window before the barrier above)
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          if (wave == w) {
#pragma unroll
            for (int t = 0; t < HL; ++t)
              if (((S1 + t - HL) >> 2) == w) {
                const float keep = lr[(S1 + t) * LP + lane];
                wave_lds_fence();
                lr[t * LP + lane] = keep;
              }
          }
        }
        wave_lds_fence();
      }
      f2 win[HL + 8];
      const float* wbase = &hr[(8 * wave) * XP + 2 * lane];
#pragma unroll
      for (int t = 0; t < HL + 8; ++t) win[t] = *reinterpret_cast<const f2*>(wbase + t * XP);
      float* lrow = &lr[(HL + 4 * wave) * LP + lane];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int m1r = 2 * j + 4 * wave + i;
        const bool own_r = m1r >= own_lo && m1r < own_hi;
        f2 lo2, hi2;  // (aa, da), (ad, dd)
#pragma unroll
        for (int m = 0; m < L; ++m) {
          const f2 hv = win[2 * i + (L - 1) - m];
          if (m == 0) {
            lo2 = pkmul_lo(a.tap[0], hv);
            hi2 = pkmul_hi(a.tap[0], hv);
          } else {
            pkfma_lo(lo2, a.tap[m], hv);
            pkfma_hi(hi2, a.tap[m], hv);
          }
        }
        if (own_r && own_c) {  // own_r fails only on rows beyond the plane / beyond this segment
          const int off = m1r * a.d1s_h + m1c;
          d1b0[off] = hi2.x;
          d1b1[off] = lo2.y;
          d1b2[off] = hi2.y;
        }
        lrow[i * LP] = lo2.x;
      }
      // level-2 horizontal pass over the four rows this wave just wrote: half-wave h takes rows h and 2 + h
      wave_lds_fence();
#pragma unroll
      for (int i = 0; i < 2; ++i) h2_row(HL + 4 * wave + 2 * i + half);
    }
    __syncthreads();

    // keep the last HL rows of the h-window for the next step: slot S0 + t -> slot t, by the wave whose next staging
    // overwrites slot S0 + t (all waves are past the vertical pass)
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      if (wave == w) {
#pragma unroll
        for (int t = 0; t < HL; ++t)
          if (((S0 + t - HL) & 3) == w) {
            const f2 keep = *reinterpret_cast<const f2*>(&hr[(S0 + t) * XP + 2 * lane]);
            wave_lds_fence();
            *reinterpret_cast<f2*>(&hr[t * XP + 2 * lane]) = keep;
          }
      }
    }

    // level-2 vertical pass: half-wave (2 wave + half) -> level-2 row j + 2 wave + half
    {
      const int j2 = j + 2 * wave + half;
      const bool live = col_live && j2 < jb;
      const bool rows_in2 = 2 * j - HL >= 0 && 2 * j + S1 <= a.H1;  // no extension anywhere in this step
      f2 lo2, hi2;
#pragma unroll
      for (int m = 0; m < L; ++m) {
        const int t = (L - 1) - m;  // extended level-1 row e = 2 j2 - HL + t; a-window slot = (actual row) - (2 j - HL)
        int sl = 2 * (2 * wave + half) + t;
        bool zero = false;
        if (!rows_in2) {
          const int e = 2 * j2 - HL + t;
          zero = !live || (zero_mode && (unsigned)e >= (unsigned)a.H1);
          sl = zero ? 0 : fold(e, a.H1) - (2 * j - HL);
        }
        f2 hv = *reinterpret_cast<const f2*>(&lr[sl * LP + 2 * kk]);
        if (zero) hv = (f2){0.0f, 0.0f};
        if (m == 0) {
          lo2 = pkmul_lo(a.tap[0], hv);
          hi2 = pkmul_hi(a.tap[0], hv);
        } else {
          pkfma_lo(lo2, a.tap[m], hv);
          pkfma_hi(hi2, a.tap[m], hv);
        }
      }
      if (live) {
        const int off_a = j2 * a.a2s_h + k2, off_d = j2 * a.d2s_h + k2;
        o2b0[off_a] = lo2.x;
        o2b1[off_d] = hi2.x;
        o2b2[off_d] = lo2.y;
        o2b3[off_d] = hi2.y;
      }
    }
  }
}

// level-2 rows per segment: the option's value, else 32; always a multiple of 8 that leaves at least two segments
static int roll_segment(const mifwt_level_desc* d2) {
  int seg = g_options[MIFWT_OPT_PAIR_ROWS];
  if (seg <= 0) seg = 24;  // measured best on 1024^2 planes (16 .. 64 tried): more, shorter segments balance the chip better than they cost in prologues
  seg = (seg + 7) / 8 * 8;
  const int cap = (int)((d2->coef_extent[0] - 1) / 8 * 8);
  return seg < cap ? seg : cap;
}

bool dwt2_fwd_roll_supported(const mifwt_level_desc* d1, const mifwt_level_desc* d2) {
  if (!dwt2_fwd_pair_supported(d1, d2)) return false;
  // one step is 16 level-1 rows and the bottom-aligned last step must find its mirrored rows in the window; 32-bit
  // element offsets inside one image of every band
  const int64_t lim = int64_t(1) << 31;
  if (d1->coef_extent[0] * d1->detail_stride[1] >= lim || d2->coef_extent[0] * d2->approx_stride[1] >= lim ||
      d2->coef_extent[0] * d2->detail_stride[1] >= lim)
    return false;
  return d1->coef_extent[0] >= 32 && d2->coef_extent[0] >= 9;
}

template <int L>
static int launch_roll(const mifwt_level_desc* d1, const mifwt_level_desc* d2, const void* x, void* const* details1,
                       void* approx2, void* const* details2, const double* lo, const double* hi, hipStream_t stream) {
  constexpr int T2C = (64 - (L - 2)) / 2;
  Dwt2RollArgs<L> a;
  a.x = static_cast<const float*>(x);
  for (int s = 0; s < 3; ++s) a.d1[s] = static_cast<float*>(details1[s]);
  a.o2[0] = static_cast<float*>(approx2);
  for (int s = 1; s < 4; ++s) a.o2[s] = static_cast<float*>(details2[s - 1]);
  a.xs_b = d1->sig_stride[0];
  a.xs_h = (int)d1->sig_stride[1];
  a.d1s_b = d1->detail_stride[0];
  a.d1s_h = (int)d1->detail_stride[1];
  a.a2s_b = d2->approx_stride[0];
  a.a2s_h = (int)d2->approx_stride[1];
  a.d2s_b = d2->detail_stride[0];
  a.d2s_h = (int)d2->detail_stride[1];
  a.H0 = (int)d1->sig_extent[0];
  a.W0 = (int)d1->sig_extent[1];
  a.H1 = (int)d1->coef_extent[0];
  a.W1 = (int)d1->coef_extent[1];
  a.H2 = (int)d2->coef_extent[0];
  a.W2 = (int)d2->coef_extent[1];
  a.mode = d1->mode;
  for (int m = 0; m < L; ++m) a.tap[m] = (f2){(float)lo[m], (float)hi[m]};
  a.strips = (a.W2 + T2C - 1) / T2C;
  a.seg = roll_segment(d2);
  a.nseg = (a.H2 + a.seg - 1) / a.seg;
  a.div_s = make_fastdiv((uint32_t)a.strips);
  a.div_g = make_fastdiv((uint32_t)a.nseg);
  const int64_t nwg = (int64_t)d1->batch * a.strips * a.nseg;
  if (nwg > INT32_MAX / 8) return MIFWT_ERR_UNSUPPORTED;
  hipLaunchKernelGGL((dwt2_fwd_roll_kernel<L>), dim3((unsigned)nwg), dim3(256), 0, stream, a);
  return hipGetLastError() == hipSuccess ? MIFWT_OK : MIFWT_ERR_LAUNCH;
}

int dwt2_fwd_roll(const mifwt_level_desc* d1, const mifwt_level_desc* d2, const void* x, void* const* details1,
                  void* approx2, void* const* details2, const double* lo, const double* hi, hipStream_t stream) {
  if (!dwt2_fwd_roll_supported(d1, d2)) return MIFWT_ERR_UNSUPPORTED;
  switch (d1->filt_len) {
    case 2: return launch_roll<2>(d1, d2, x, details1, approx2, details2, lo, hi, stream);
    case 4: return launch_roll<4>(d1, d2, x, details1, approx2, details2, lo, hi, stream);
    case 6: return launch_roll<6>(d1, d2, x, details1, approx2, details2, lo, hi, stream);
    case 8: return launch_roll<8>(d1, d2, x, details1, approx2, details2, lo, hi, stream);
    default: return MIFWT_ERR_UNSUPPORTED;
  }
}

}  // namespace mifwt
