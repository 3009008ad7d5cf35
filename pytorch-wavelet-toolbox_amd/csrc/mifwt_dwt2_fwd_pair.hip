// mifwt_dwt2_fwd_pair.hip — TWO consecutive 2-D analysis levels in one launch (gfx950), kernel id 12.
//
// Reference seam: two trips of the level loop of src/ptwt/conv_transform_2.py:142-149 (F.pad + F.conv2d(stride 2) +
// split, the approximation fed back as the next input).  A pyramid level only RETURNS its three detail bands; the
// approximation of every level but the last is an intermediate that the one-kernel-per-level path writes to HBM and
// reads straight back (config 2: 68 MB out + 68 MB in between levels 1 and 2, a fifth of all traffic).  Here a
// 256-thread workgroup owns a T2R x T2C tile of LEVEL-2 coefficients and everything below it:
//   1. burst-load the level-0 window (4 T2R + 3 (L-2)) x (2*64 + L-2), boundary extension as an index map -> LDS
//   2. level-1 horizontal pass, in place (lane = level-1 column, 64 of them: 2 T2C owned + L-2 halo to the left)
//   3. level-1 vertical pass from a register window: the three detail bands of the OWNED 2 T2R x 2 T2C block go to
//      HBM, the approximation (owned block + L-2 halo rows / columns) stays in LDS
//   4. level-2 horizontal pass over the LDS approximation, in place (two rows per wave step, 32 lanes each)
//   5. level-2 vertical pass -> the four level-2 bands to HBM.
// The level-1 approximation never exists in HBM.  Level-2's boundary extension acts on the level-1 approximation:
// every workgroup computes a window of ACTUAL level-1 rows / columns (the nominal window shifted into the plane at an
// edge), and edge tiles read it through the extension index map; periodic extension would need the far side of
// the plane and is left to the per-level kernels.  Same arithmetic, in the same order, as two calls of the tile
// kernel (mifwt_dwt2_tile.h): results are bit-identical to the per-level path.  tests/test_pair_model.py models the window
// bookkeeping on the CPU.
// Algorithmic traffic: 4 B H W read + 4 B (3 H1 W1 + 4 H2 W2) written.
#include "mifwt_dwt2_tile.h"

namespace mifwt {

template <int L>
struct Dwt2PairArgs {
  const float* x;
  float* d1[3];  // level-1 bands ad, da, dd
  float* o2[4];  // level-2 bands aa, ad, da, dd
  int64_t xs_b, xs_h;
  int64_t d1s_b, d1s_h;
  int64_t a2s_b, a2s_h, d2s_b, d2s_h;
  int H0, W0, H1, W1, H2, W2;
  int tiles_c, tiles_r;
  FastDiv div_c, div_r;  // by tiles_c, tiles_r
  int mode;
  int sync_stage;
  f2 tap[L];  // (dec_lo[m], dec_hi[m])
};

constexpr int pair_lds_bytes(int L, int T2R) { return (2 * (2 * T2R + L - 2) + L - 2) * (2 * 64 + L - 2) * 4; }
constexpr int pair_occupancy(int L, int T2R) {
  const int n = (160 * 1024) / pair_lds_bytes(L, T2R);
  return n > 8 ? 8 : (n < 1 ? 1 : n);
}

template <int L, int T2R>
__global__ void __launch_bounds__(256, pair_occupancy(L, T2R)) dwt2_fwd_pair_kernel(const Dwt2PairArgs<L> a) {
  constexpr int HL = L - 2;
  constexpr int C1 = 64;              // level-1 columns of a tile, halo included = lanes
  constexpr int T2C = (C1 - HL) / 2;  // level-2 columns of a tile
  constexpr int OC1 = 2 * T2C;        // level-1 columns a tile owns (writes details for)
  constexpr int R1 = 2 * T2R + HL;    // level-1 rows of a tile, halo included
  constexpr int R0 = 2 * R1 + HL;     // level-0 rows
  constexpr int C0 = 2 * C1 + HL;     // level-0 columns
  constexpr int XP = C0;              // LDS pitch of the level-0 window (even)
  constexpr int NQ = (C0 + 63) / 64;
  constexpr int RPW0 = (R0 + 3) / 4;  // level-0 rows per wave (load, horizontal pass)
  constexpr int RW1 = (R1 + 3) / 4;   // level-1 rows per wave (vertical pass)
  constexpr int LP = 64;              // LDS pitch of the level-1 approximation
  constexpr int RW2 = (T2R + 7) / 8;  // level-2 rows per half-wave (vertical pass)
  static_assert(T2C <= 32 && T2C >= 1, "half-wave layout of the level-2 passes");
  __shared__ __attribute__((aligned(16))) float xt[R0 * XP];
  float* const ll = xt;  // the level-1 approximation reuses the window once the vertical pass holds it in registers
  static_assert(R1 * LP + 8 <= R0 * XP, "approximation tile must fit into the window it replaces");

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  uint32_t utc, utr;
  const int img = (int)a.div_r.divmod(a.div_c.divmod((uint32_t)bid, utc), utr);
  const int tc = (int)utc, tr = (int)utr;
  const int k2_0 = tc * T2C, j2_0 = tr * T2R;
  // window of actual level-1 rows / columns: nominal [2 j2_0 - HL, 2 j2_0 + 2 T2R), shifted into [0, H1)
  const int s1r = min(max(2 * j2_0 - HL, 0), a.H1 - R1);
  const int s1c = min(max(2 * k2_0 - HL, 0), a.W1 - C1);

  // ---- 1. level-0 window -> LDS -------------------------------------------------------------------------------------
  const uint32_t img_bytes = ((uint32_t)(a.H0 - 1) * (uint32_t)a.xs_h + (uint32_t)a.W0) * 4u;
  const __amdgpu_buffer_rsrc_t xrsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x + (int64_t)img * a.xs_b), 0, img_bytes, 0x00020000);
  constexpr uint32_t kOob = 0x80000000u;
  const int c_first = 2 * s1c - HL, r_first = 2 * s1r - HL;
  // boundary extension = the branch-free single-fold map (Fold1, mifwt_stream.h): the window of actual level-1 rows /
  // columns keeps every requested level-0 index within one period of the plane
  __builtin_assume(wave >= 0 && wave < 4);
  const bool zero_mode = a.mode == MIFWT_MODE_ZERO;
  Fold1 fold;
  fold.set(a.mode);
  const bool rows_inside = r_first >= 0 && r_first + R0 <= a.H0;
  const uint32_t row_bytes = (uint32_t)a.xs_h * 4u;
  uint32_t coff[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int c = lane + 64 * q, ci = c_first + c;
    const bool dead = c >= C0 || (zero_mode && (unsigned)ci >= (unsigned)a.W0);
    coff[q] = dead ? kOob : 4u * (uint32_t)fold(ci, a.W0);
  }
  float v[RPW0][NQ];
  if (rows_inside) {
    uint32_t soff = (uint32_t)(r_first + wave) * row_bytes;
#pragma unroll
    for (int i = 0; i < RPW0; ++i) {
      const uint32_t so = (4 * i + 3 < R0 || wave + 4 * i < R0) ? soff : (uint32_t)(r_first + R0 - 1) * row_bytes;
#pragma unroll
      for (int q = 0; q < NQ; ++q) v[i][q] = tile_load<float>(xrsrc, coff[q], so);
      soff += 4u * row_bytes;
    }
  } else {
#pragma unroll
    for (int i = 0; i < RPW0; ++i) {
      const int r = wave + 4 * i, ri = r_first + r;
      const bool dead = r >= R0 || (zero_mode && (unsigned)ri >= (unsigned)a.H0);
      const uint32_t soff = __builtin_amdgcn_readfirstlane(dead ? 0u : (uint32_t)fold(ri, a.H0) * row_bytes);
#pragma unroll
      for (int q = 0; q < NQ; ++q) v[i][q] = tile_load<float>(xrsrc, dead ? kOob : coff[q], soff);
    }
  }
#pragma unroll
  for (int i = 0; i < RPW0; ++i) {
    const int r = wave + 4 * i;
    if (4 * i + 3 < R0 || r < R0) {  // first clause: compile time
#pragma unroll
      for (int q = 0; q < NQ; ++q)
        if (64 * q + 63 < XP || lane + 64 * q < XP) xt[r * XP + lane + 64 * q] = v[i][q];
    }
  }
  // no workgroup barrier here: row r is staged, filtered and overwritten by the same wave (rows wave + 4 i), whose DS
  // operations execute in order; the option keeps the barrier for A/B measurements
  if (a.sync_stage) __syncthreads(); else wave_lds_fence();

  // ---- 2. level-1 horizontal pass, in place: row r becomes (lo, hi)[c] of level-1 column s1c + c --------------------
#pragma unroll
  for (int i = 0; i < RPW0; ++i) {
    const int r = wave + 4 * i;
    if (4 * i + 3 < R0 || r < R0) {
      const f2* row = reinterpret_cast<const f2*>(&xt[r * XP + 2 * lane]);
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
      *reinterpret_cast<f2*>(&xt[r * XP + 2 * lane]) = acc;
    }
  }
  __syncthreads();

  // ---- 3. level-1 vertical pass: details of the owned block -> HBM, approximation -> LDS ---------------------------
  {
    const int i1b = wave * RW1;
    f2 win[2 * RW1 + HL];
#pragma unroll
    for (int t = 0; t < 2 * RW1 + HL; ++t) {
      const int row = min(2 * i1b + t, R0 - 1);
      win[t] = *reinterpret_cast<const f2*>(&xt[row * XP + 2 * lane]);
    }
    __syncthreads();  // every wave holds its rows: the window storage is free for the approximation tile
    const int m1c = s1c + lane;
    const bool own_c = m1c >= 2 * k2_0 && m1c < 2 * k2_0 + OC1;
    float* db[3];  // wave-uniform bases; lanes add 32-bit element offsets
#pragma unroll
    for (int s = 0; s < 3; ++s) db[s] = a.d1[s] + (int64_t)img * a.d1s_b;
#pragma unroll
    for (int i = 0; i < RW1; ++i) {
      const int i1 = i1b + i;  // wave-uniform
      if (i1 < R1) {
        const int m1r = s1r + i1;
        const bool own_r = m1r >= 2 * j2_0 && m1r < 2 * j2_0 + 2 * T2R;
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
        ll[i1 * LP + lane] = lo2.x;
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
            const int off = m1r * (int)a.d1s_h + m1c;
            db[0][off] = hi2.x;
            db[1][off] = lo2.y;
            db[2][off] = hi2.y;
          }
        }
      }
    }
  }
  __syncthreads();

  // ---- 4. level-2 horizontal pass over the approximation tile, in place; half-wave h takes row 2q + h -----------------
  const int half = lane >> 5, kk = lane & 31;
  const int k2 = k2_0 + kk;
  const bool col_live = kk < T2C && k2 < a.W2;
  {
    const bool cols_in2 = 2 * k2_0 - HL >= 0 && 2 * k2_0 + OC1 <= a.W1;  // then s1c == 2 k2_0 - HL, no extension
    if (cols_in2) {
#pragma unroll 1
      for (int q = wave; q < R1 / 2; q += 4) {
        const int r = 2 * q + half;
        const f2* row = reinterpret_cast<const f2*>(&ll[r * LP + 2 * kk]);
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
        if (kk < T2C) *reinterpret_cast<f2*>(&ll[r * LP + 2 * kk]) = acc;
      }
    } else {
      int cidx[L];
#pragma unroll
      for (int p = 0; p < L; ++p) {
        const int e = 2 * k2 - HL + p;
        const bool dead = !col_live || (zero_mode && (unsigned)e >= (unsigned)a.W1);
        cidx[p] = dead ? -1 : fold(e, a.W1) - s1c;
      }
#pragma unroll 1
      for (int q = wave; q < R1 / 2; q += 4) {
        const int r = 2 * q + half;
        f2 acc;
#pragma unroll
        for (int p = 0; p < L / 2; ++p) {
          f2 xx;
          xx.x = cidx[2 * p] >= 0 ? ll[r * LP + cidx[2 * p]] : 0.0f;
          xx.y = cidx[2 * p + 1] >= 0 ? ll[r * LP + cidx[2 * p + 1]] : 0.0f;
          if (p == 0) {
            acc = pkmul_lo(a.tap[L - 1], xx);
          } else {
            pkfma_lo(acc, a.tap[L - 1 - 2 * p], xx);
          }
          pkfma_hi(acc, a.tap[L - 2 - 2 * p], xx);
        }
        wave_lds_fence();
        if (kk < T2C) *reinterpret_cast<f2*>(&ll[r * LP + 2 * kk]) = acc;
      }
    }
  }
  __syncthreads();

  // ---- 5. level-2 vertical pass + stores: half-wave (2 wave + half) owns RW2 level-2 rows ---------------------------
  {
    const bool rows_in2 = 2 * j2_0 - HL >= 0 && 2 * j2_0 + 2 * T2R <= a.H1;  // then s1r == 2 j2_0 - HL
    float* ob[4];
    ob[0] = a.o2[0] + (int64_t)img * a.a2s_b;
#pragma unroll
    for (int s = 1; s < 4; ++s) ob[s] = a.o2[s] + (int64_t)img * a.d2s_b;
#pragma unroll
    for (int i = 0; i < RW2; ++i) {
      const int j2l = (2 * wave + half) * RW2 + i;
      const int j2 = j2_0 + j2l;
      const bool live = col_live && j2l < T2R && j2 < a.H2;
      f2 lo2, hi2;
#pragma unroll
      for (int m = 0; m < L; ++m) {
        const int t = (L - 1) - m;  // extended level-1 row 2 j2 + 1 - m = 2 j2 - HL + t
        int local;
        if (rows_in2) {
          local = min(2 * j2l + t, R1 - 1);
        } else {
          const int e = 2 * j2 - HL + t;
          const bool dead = !live || (zero_mode && (unsigned)e >= (unsigned)a.H1);
          local = dead ? -1 : fold(e, a.H1) - s1r;
        }
        f2 hv = (f2){0.0f, 0.0f};
        if (local >= 0) hv = *reinterpret_cast<const f2*>(&ll[local * LP + 2 * kk]);
        if (m == 0) {
          lo2 = pkmul_lo(a.tap[0], hv);
          hi2 = pkmul_hi(a.tap[0], hv);
        } else {
          pkfma_lo(lo2, a.tap[m], hv);
          pkfma_hi(hi2, a.tap[m], hv);
        }
      }
      if (live) {
        const int off_a = j2 * (int)a.a2s_h + k2, off_d = j2 * (int)a.d2s_h + k2;
        ob[0][off_a] = lo2.x;
        ob[1][off_d] = hi2.x;
        ob[2][off_d] = lo2.y;
        ob[3][off_d] = hi2.y;
      }
    }
  }
}

template <int L, int T2R>
static int launch_pair(const mifwt_level_desc* d1, const mifwt_level_desc* d2, const void* x, void* const* details1,
                       void* approx2, void* const* details2, const double* lo, const double* hi, hipStream_t stream) {
  constexpr int T2C = (64 - (L - 2)) / 2;
  Dwt2PairArgs<L> a;
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
  a.sync_stage = g_options[MIFWT_OPT_SYNC_STAGE];
  for (int m = 0; m < L; ++m) a.tap[m] = (f2){(float)lo[m], (float)hi[m]};
  a.tiles_c = (a.W2 + T2C - 1) / T2C;
  a.tiles_r = (a.H2 + T2R - 1) / T2R;
  a.div_c = make_fastdiv((uint32_t)a.tiles_c);
  a.div_r = make_fastdiv((uint32_t)a.tiles_r);
  const int64_t ntiles = (int64_t)d1->batch * a.tiles_c * a.tiles_r;
  if (ntiles > INT32_MAX / 8) return MIFWT_ERR_UNSUPPORTED;
  hipLaunchKernelGGL((dwt2_fwd_pair_kernel<L, T2R>), dim3((unsigned)ntiles), dim3(256), 0, stream, a);
  return hipGetLastError() == hipSuccess ? MIFWT_OK : MIFWT_ERR_LAUNCH;
}

// level-2 rows per tile: 8 unless the option overrides (A/B) or the level-1 plane is too short for that window
static int pair_rows(const mifwt_level_desc* d1) {
  int t = g_options[MIFWT_OPT_PAIR_ROWS];
  // measured on 64 x 1024^2 (4 / 6 / 8 / 12 rows): haar 101 / 104 / 106 / 107 us, db2 121 / 113 / 116 / 116, db3 135 / 128 /
  // 127 / 129, db4 170 / 150 / 143 / 153 — the longer the filter, the more rows amortise the 3 (L - 2)-row halo
  if (t <= 0) t = d1->filt_len <= 2 ? 4 : (d1->filt_len <= 4 ? 6 : 8);
  t = t <= 4 ? 4 : (t <= 6 ? 6 : (t <= 8 ? 8 : 12));
  const int64_t h1 = d1->coef_extent[0];
  const int hl = d1->filt_len - 2;
  while (t > 4 && 2 * t + hl > h1) t = t == 12 ? 8 : (t == 8 ? 6 : 4);
  return t;
}

bool dwt2_fwd_pair_supported(const mifwt_level_desc* d1, const mifwt_level_desc* d2) {
  if (g_options[MIFWT_OPT_FORCE_GENERIC] || g_options[MIFWT_OPT_PAIR_MODE] == 2) return false;
  if (d1->ndim != 2 || d2->ndim != 2 || d1->dtype != MIFWT_F32 || d2->dtype != MIFWT_F32) return false;
  const int L = d1->filt_len;
  if (d2->filt_len != L || L < 2 || L > 8 || (L & 1)) return false;
  if (d1->mode != d2->mode || d1->mode == MIFWT_MODE_PERIODIC || d1->batch != d2->batch) return false;
  for (int i = 0; i < 2; ++i)
    if (d2->sig_extent[i] != d1->coef_extent[i]) return false;
  if (d1->sig_stride[2] != 1 || d1->detail_stride[2] != 1 || d2->approx_stride[2] != 1 || d2->detail_stride[2] != 1)
    return false;
  const int64_t span = (d1->sig_extent[0] - 1) * d1->sig_stride[1] + d1->sig_extent[1];
  if (d1->sig_stride[1] < 0 || span >= (int64_t(1) << 29)) return false;
  if (d1->detail_stride[0] < 0 || d1->detail_stride[1] < 0) return false;
  for (int i = 0; i < 2; ++i)
    if (d2->approx_stride[i] < 0 || d2->detail_stride[i] < 0) return false;
  const int64_t lim = int64_t(1) << 31;  // 32-bit element offsets inside one image of a band
  if (d1->coef_extent[0] * d1->detail_stride[1] >= lim || d2->coef_extent[0] * d2->approx_stride[1] >= lim ||
      d2->coef_extent[0] * d2->detail_stride[1] >= lim)
    return false;
  // the level-1 window of a tile (halo included) must fit into the level-1 plane
  if (d1->coef_extent[1] < 64 || d1->coef_extent[0] < 2 * 4 + (L - 2)) return false;
  return true;
}

template <int L>
static int launch_pair_rows(const mifwt_level_desc* d1, const mifwt_level_desc* d2, const void* x, void* const* details1,
                            void* approx2, void* const* details2, const double* lo, const double* hi, hipStream_t stream) {
  switch (pair_rows(d1)) {
    case 4: return launch_pair<L, 4>(d1, d2, x, details1, approx2, details2, lo, hi, stream);
    case 6: return launch_pair<L, 6>(d1, d2, x, details1, approx2, details2, lo, hi, stream);
    case 8: return launch_pair<L, 8>(d1, d2, x, details1, approx2, details2, lo, hi, stream);
    default: return launch_pair<L, 12>(d1, d2, x, details1, approx2, details2, lo, hi, stream);
  }
}

int dwt2_fwd_pair(const mifwt_level_desc* d1, const mifwt_level_desc* d2, const void* x, void* const* details1,
                  void* approx2, void* const* details2, const double* lo, const double* hi, hipStream_t stream) {
  if (!dwt2_fwd_pair_supported(d1, d2)) return MIFWT_ERR_UNSUPPORTED;
  switch (d1->filt_len) {
    case 2: return launch_pair_rows<2>(d1, d2, x, details1, approx2, details2, lo, hi, stream);
    case 4: return launch_pair_rows<4>(d1, d2, x, details1, approx2, details2, lo, hi, stream);
    case 6: return launch_pair_rows<6>(d1, d2, x, details1, approx2, details2, lo, hi, stream);
    case 8: return launch_pair_rows<8>(d1, d2, x, details1, approx2, details2, lo, hi, stream);
    default: return MIFWT_ERR_UNSUPPORTED;
  }
}

}  // namespace mifwt
