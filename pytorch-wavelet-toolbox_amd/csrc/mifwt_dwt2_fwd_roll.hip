// mifwt_dwt2_fwd_roll.hip — TWO consecutive 2-D analysis levels in one launch, rolling column strips (gfx950).
//
// Same seam and same results as mifwt_dwt2_fwd_pair.hip (two trips of src/ptwt/conv_transform_2.py:142-149, the
// level-1 approximation never reaches HBM), different shape of the work.  The tile version recomputes, per tile, the
// 3 (L-2) level-0 halo rows under a level-2 tile: with 8 level-2 rows per tile that is 50 input rows loaded,
// staged and filtered for every 32 it owns.  Here a 256-thread workgroup owns a column strip (64 level-1 columns: 64 -
// (L-2) owned + L-2 halo) of one SEGMENT of level-2 rows and walks down it in steps of 8 level-2 rows = 16 level-1 rows
// = 32 level-0 rows; the horizontally filtered rows and the level-1 approximation live in two LDS rings that keep
// the last L-2 rows of the previous step, so inside a segment nothing is loaded or filtered twice:
//   per step   wave w:  its 8 prefetched level-0 rows -> h-ring slots; request the NEXT step's rows (in flight
//                       during everything below); level-1 horizontal pass over its own rows, in place
//              barrier  level-1 vertical pass, 4 rows per wave from a register window: details -> HBM, approximation
//                       -> a-ring; level-2 horizontal pass over the rows it just produced, in place
//              barrier  level-2 vertical pass, one row per half-wave -> the four level-2 bands to HBM.
// A segment starts with a prologue of 3 (L-2) level-0 rows that fills the rings.  The first segment of a plane is
// aligned to its top, all others to its bottom (the boundary extension of level 2 reads ACTUAL level-1 rows through
// the index map, and those must still be in the ring: oracle-checked in tests/test_host_logic.py's numpy model).
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
  int64_t xs_b, xs_h;
  int64_t d1s_b, d1s_h;
  int64_t a2s_b, a2s_h, d2s_b, d2s_h;
  int H0, W0, H1, W1, H2, W2;
  int strips, nseg, seg;  // column strips per plane, row segments per plane, level-2 rows per segment (multiple of 8)
  int mode;
  f2 tap[L];  // (dec_lo[m], dec_hi[m])
};

constexpr int roll_lds_bytes(int L) { return ((32 + L - 2) * (128 + L - 2) + (16 + L - 2) * 64) * 4; }
constexpr int roll_occupancy(int L) {
  // 6 workgroups' rings fit; the prefetch registers of the 6- and 8-tap instances need the 5-wave register budget
  const int n = (160 * 1024) / roll_lds_bytes(L), cap = L <= 4 ? 6 : 5;
  return n > cap ? cap : (n < 1 ? 1 : n);
}

template <int L>
__global__ void __launch_bounds__(256, roll_occupancy(L)) dwt2_fwd_roll_kernel(const Dwt2RollArgs<L> a) {
  constexpr int HL = L - 2;
  constexpr int C1 = 64;              // level-1 columns of a strip, halo included = lanes
  constexpr int T2C = (C1 - HL) / 2;  // level-2 columns of a strip
  constexpr int OC1 = 2 * T2C;        // level-1 columns a strip owns
  constexpr int C0 = 2 * C1 + HL;     // level-0 columns
  constexpr int XP = C0;              // pitch of the h-ring (floats, even)
  constexpr int NQ = (C0 + 63) / 64;
  constexpr int S2 = 8, S1 = 16, S0 = 32;  // rows per step at levels 2 / 1 / 0
  constexpr int RH = S0 + HL;              // h-ring rows (level-0 row index space, horizontally filtered)
  constexpr int RL = S1 + HL;              // a-ring rows (level-1 approximation, then its horizontal (lo, hi) image)
  constexpr int LP = 64;
  constexpr int PR0 = 3 * HL;              // prologue: level-0 rows
  constexpr int PW0 = (PR0 + 3) / 4;       //           per wave
  constexpr int PW1 = (HL + 3) / 4;        // prologue: level-1 rows per wave
  static_assert(T2C <= 32 && T2C >= 1 && PR0 <= RH, "ring geometry");
  __shared__ __attribute__((aligned(16))) float hr[RH * XP];
  __shared__ __attribute__((aligned(16))) float lr[RL * LP + 8];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int tc = bid % a.strips;
  const int sg = (bid / a.strips) % a.nseg;
  const int img = bid / (a.strips * a.nseg);
  // level-2 rows [ja, jb) of this segment; segment 0 is top-aligned (its last step is masked beyond jb), the others
  // end exactly at jb
  const int jb = a.H2 - (a.nseg - 1 - sg) * a.seg;
  const int ja = sg > 0 ? jb - a.seg : 0;
  const int nsteps = (jb - ja + S2 - 1) / S2;
  const int own_lo = 2 * ja, own_hi = min(2 * jb, a.H1);  // level-1 rows whose details this segment writes

  const int k2_0 = tc * T2C;
  const int s1c = min(max(2 * k2_0 - HL, 0), a.W1 - C1);
  const int c_first = 2 * s1c - HL;

  // ---- level-0 column offsets, once per workgroup ---------------------------------------------------------------------
  const uint32_t img_bytes = ((uint32_t)(a.H0 - 1) * (uint32_t)a.xs_h + (uint32_t)a.W0) * 4u;
  const __amdgpu_buffer_rsrc_t xrsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x + (int64_t)img * a.xs_b), 0, img_bytes, 0x00020000);
  constexpr uint32_t kOob = 0x80000000u;
  const uint32_t row_bytes = (uint32_t)a.xs_h * 4u;
  uint32_t coff[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int c = lane + 64 * q;
    const int m = c < C0 ? ext_index_near(c_first + c, a.W0, a.mode) : -1;
    coff[q] = m < 0 ? kOob : 4u * (uint32_t)m;
  }

  // level-0 rows r_first + wave + 4 i (i < N) of the extended plane -> registers
  auto request = [&](auto n_tag, float (&v)[decltype(n_tag)::value][NQ], int r_first, int r_end) {
    constexpr int N = decltype(n_tag)::value;
    if (r_first >= 0 && r_first + 4 * N <= a.H0 && r_first + 4 * N <= r_end) {
#pragma unroll
      for (int i = 0; i < N; ++i) {
        const uint32_t soff = (uint32_t)(r_first + wave + 4 * i) * row_bytes;
#pragma unroll
        for (int q = 0; q < NQ; ++q) v[i][q] = tile_load<float>(xrsrc, coff[q], soff);
      }
    } else {
#pragma unroll
      for (int i = 0; i < N; ++i) {
        const int r = r_first + wave + 4 * i;
        const int m = r < r_end ? ext_index_near(r, a.H0, a.mode) : -1;
        const uint32_t soff = m < 0 ? 0u : (uint32_t)m * row_bytes;
#pragma unroll
        for (int q = 0; q < NQ; ++q) v[i][q] = tile_load<float>(xrsrc, m < 0 ? kOob : coff[q], soff);
      }
    }
  };

  // registers -> ring slots slot0 + wave + 4 i (mod RH), then the level-1 horizontal pass over the same rows, in place
  // (a row is staged, read and overwritten by ONE wave, whose DS operations execute in order: no barrier in between)
  auto stage_h1 = [&](auto n_tag, float (&v)[decltype(n_tag)::value][NQ], int slot0, int nrows) {
    constexpr int N = decltype(n_tag)::value;
#pragma unroll
    for (int i = 0; i < N; ++i) {
      if (wave + 4 * i < nrows) {
        int s = slot0 + wave + 4 * i;
        s = s >= RH ? s - RH : s;
#pragma unroll
        for (int q = 0; q < NQ; ++q)
          if (lane + 64 * q < XP) hr[s * XP + lane + 64 * q] = v[i][q];
      }
    }
    wave_lds_fence();
#pragma unroll
    for (int i = 0; i < N; ++i) {
      if (wave + 4 * i < nrows) {
        int s = slot0 + wave + 4 * i;
        s = s >= RH ? s - RH : s;
        const f2* row = reinterpret_cast<const f2*>(&hr[s * XP + 2 * lane]);
        f2 acc;
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
        wave_lds_fence();
        *reinterpret_cast<f2*>(&hr[s * XP + 2 * lane]) = acc;
      }
    }
  };

  // level-2 column map (edge strips only), once per workgroup
  const int half = lane >> 5, kk = lane & 31;
  const int k2 = k2_0 + kk;
  const bool col_live = kk < T2C && k2 < a.W2;
  const bool cols_in2 = 2 * k2_0 - HL >= 0 && 2 * k2_0 + OC1 <= a.W1;  // then s1c == 2 k2_0 - HL, no extension
  int cidx[L];
#pragma unroll
  for (int p = 0; p < L; ++p) {
    const int m = col_live ? ext_index_near(2 * k2 - HL + p, a.W1, a.mode) : -1;
    cidx[p] = cols_in2 ? 2 * kk + p : (m < 0 ? -1 : m - s1c);
  }
  // level-2 horizontal pass over a-ring row `s` for the lanes of one half-wave (or all lanes with both = true), in place
  auto h2_row = [&](int s, bool active) {
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
        xx.x = cidx[2 * p] >= 0 ? lr[s * LP + cidx[2 * p]] : 0.0f;
        xx.y = cidx[2 * p + 1] >= 0 ? lr[s * LP + cidx[2 * p + 1]] : 0.0f;
        if (p == 0) {
          acc = pkmul_lo(a.tap[L - 1], xx);
        } else {
          pkfma_lo(acc, a.tap[L - 1 - 2 * p], xx);
        }
        pkfma_hi(acc, a.tap[L - 2 - 2 * p], xx);
      }
    }
    wave_lds_fence();
    if (active && kk < T2C) *reinterpret_cast<f2*>(&lr[s * LP + 2 * kk]) = acc;
  };

  const int m1c = s1c + lane;
  const bool own_c = m1c >= 2 * k2_0 && m1c < 2 * k2_0 + OC1;
  // band bases stay on the scalar unit (wave-uniform); the lanes add their column
  const int64_t d1_img = (int64_t)img * a.d1s_b, a2_img = (int64_t)img * a.a2s_b, d2_img = (int64_t)img * a.d2s_b;

  // ---- prologue: level-0 rows [4 ja - 3 HL, 4 ja) -> h-ring slots [0, 3 HL); level-1 rows [2 ja - HL, 2 ja) -> a-ring
  // slots [0, HL).  (For segment 0 those level-1 rows lie above the plane and are never read; their level-0 rows are
  // real: the level-0 extension.)
  float pv[S0 / 4][NQ];
  int hs = PR0 % RH;  // h-ring slot of the first row of the coming step
  int ls = HL;        // a-ring slot of the first level-1 row of the coming step
  if constexpr (HL > 0) {
    float pp[PW0][NQ];
    request(std::integral_constant<int, PW0>{}, pp, 4 * ja - PR0, 4 * ja);
    request(std::integral_constant<int, S0 / 4>{}, pv, 4 * ja, 4 * ja + S0);
    stage_h1(std::integral_constant<int, PW0>{}, pp, 0, PR0);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < PW1; ++i) {
      const int il = wave + 4 * i;  // level-1 row 2 ja - HL + il, a-ring slot il; its h-rows: slots 2 il .. 2 il + L - 1
      if (il < HL) {
        float aa;
#pragma unroll
        for (int m = 0; m < L; ++m) {
          const float hv = hr[(2 * il + (L - 1) - m) * XP + 2 * lane];
          // same operation as the low half of the packed vertical pass
          aa = m == 0 ? a.tap[0].x * hv : __builtin_fmaf(a.tap[m].x, hv, aa);
        }
        lr[il * LP + lane] = aa;
        wave_lds_fence();
        h2_row(il, half == 0);
      }
    }
    __syncthreads();  // the prologue's h-rows are dead: step 0 may overwrite their slots
  } else {
    request(std::integral_constant<int, S0 / 4>{}, pv, 4 * ja, 4 * ja + S0);
  }

  // ---- steps ------------------------------------------------------------------------------------------------------------
#pragma unroll 1
  for (int st = 0; st < nsteps; ++st) {
    const int j = ja + S2 * st;
    stage_h1(std::integral_constant<int, S0 / 4>{}, pv, hs, S0);
    if (st + 1 < nsteps) request(std::integral_constant<int, S0 / 4>{}, pv, 4 * (j + S2), 4 * (j + S2) + S0);
    __syncthreads();

    // level-1 vertical pass: wave w -> level-1 rows 2 j + 4 w + i, i < 4; h-rows 4 j + 8 w - HL + t, t < HL + 8
    {
      f2 win[HL + 8];
      int base = hs + 8 * wave - HL;
      base = base < 0 ? base + RH : base;
#pragma unroll
      for (int t = 0; t < HL + 8; ++t) {
        int s = base + t;
        s = s >= RH ? s - RH : s;
        s = s >= RH ? s - RH : s;
        win[t] = *reinterpret_cast<const f2*>(&hr[s * XP + 2 * lane]);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int m1r = 2 * j + 4 * wave + i;
        int sl = ls + 4 * wave + i;
        sl = sl >= RL ? sl - RL : sl;
        const bool own_r = m1r >= own_lo && m1r < own_hi;
        f2 lo2;  // (aa, da)
#pragma unroll
        for (int m = 0; m < L; ++m) {
          const f2 hv = win[2 * i + (L - 1) - m];
          if (m == 0) {
            lo2 = pkmul_lo(a.tap[0], hv);
          } else {
            pkfma_lo(lo2, a.tap[m], hv);
          }
        }
        lr[sl * LP + lane] = lo2.x;
        if (own_r) {
          f2 hi2;  // (ad, dd)
#pragma unroll
          for (int m = 0; m < L; ++m) {
            const f2 hv = win[2 * i + (L - 1) - m];
            if (m == 0) {
              hi2 = pkmul_hi(a.tap[0], hv);
            } else {
              pkfma_hi(hi2, a.tap[m], hv);
            }
          }
          if (own_c) {
            const int64_t off = d1_img + (int64_t)m1r * a.d1s_h;
            (a.d1[0] + off)[m1c] = hi2.x;
            (a.d1[1] + off)[m1c] = lo2.y;
            (a.d1[2] + off)[m1c] = hi2.y;
          }
        }
      }
      // level-2 horizontal pass over the four rows this wave just wrote: half-wave h takes rows (h, 2 + h)
      wave_lds_fence();
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        int sl = ls + 4 * wave + 2 * i + half;
        sl = sl >= RL ? sl - RL : sl;
        h2_row(sl, true);
      }
    }
    __syncthreads();

    // level-2 vertical pass: half-wave (2 wave + half) -> level-2 row j + 2 wave + half
    {
      const int j2 = j + 2 * wave + half;
      const bool live = col_live && j2 < jb;
      const bool rows_in2 = 2 * j - HL >= 0 && 2 * j + S1 <= a.H1;
      f2 lo2, hi2;
#pragma unroll
      for (int m = 0; m < L; ++m) {
        const int t = (L - 1) - m;  // extended level-1 row 2 j2 - HL + t
        int rel;                    // relative to the first level-1 row of this step, in [-HL, 16)
        bool zero = false;
        if (rows_in2) {
          rel = 2 * (2 * wave + half) - HL + t;
        } else {
          const int e = live ? ext_index_near(2 * j2 - HL + t, a.H1, a.mode) : -1;
          zero = e < 0;
          rel = zero ? 0 : e - 2 * j;
        }
        int sl = ls + rel;
        sl = sl < 0 ? sl + RL : sl;
        sl = sl >= RL ? sl - RL : sl;
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
        // j2 differs between the half-waves: 32-bit per-lane offsets onto scalar bases (planes < 2^31 elements)
        const int off_a = j2 * (int)a.a2s_h + k2, off_d = j2 * (int)a.d2s_h + k2;
        (a.o2[0] + a2_img)[off_a] = lo2.x;
        (a.o2[1] + d2_img)[off_d] = hi2.x;
        (a.o2[2] + d2_img)[off_d] = lo2.y;
        (a.o2[3] + d2_img)[off_d] = hi2.y;
      }
    }
    hs += S0;
    hs = hs >= RH ? hs - RH : hs;
    ls += S1;
    ls = ls >= RL ? ls - RL : ls;
  }
}

// level-2 rows per segment: the option's value, else 32; always a multiple of 8 that leaves at least two segments
static int roll_segment(const mifwt_level_desc* d2) {
  int seg = g_options[MIFWT_OPT_PAIR_ROWS];
  if (seg <= 0) seg = 32;
  seg = (seg + 7) / 8 * 8;
  const int cap = (int)((d2->coef_extent[0] - 1) / 8 * 8);
  return seg < cap ? seg : cap;
}

bool dwt2_fwd_roll_supported(const mifwt_level_desc* d1, const mifwt_level_desc* d2) {
  if (!dwt2_fwd_pair_supported(d1, d2)) return false;
  // rings: one step is 16 level-1 rows; the bottom-aligned last step must find its mirrored rows in the ring
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
  a.xs_h = d1->sig_stride[1];
  a.d1s_b = d1->detail_stride[0];
  a.d1s_h = d1->detail_stride[1];
  a.a2s_b = d2->approx_stride[0];
  a.a2s_h = d2->approx_stride[1];
  a.d2s_b = d2->detail_stride[0];
  a.d2s_h = d2->detail_stride[1];
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
