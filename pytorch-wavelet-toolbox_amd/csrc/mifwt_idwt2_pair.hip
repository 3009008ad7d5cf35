// mifwt_idwt2_pair.hip — TWO consecutive 2-D synthesis levels in one launch (gfx950), kernel id 13.
//
// Reference seam: two trips of waverec2's level loop (src/ptwt/conv_transform_2.py:222-249: torch.stack +
// F.conv_transpose2d(stride 2) + crops, the result fed back as the next approximation).  The approximation between the two
// levels is an intermediate that one-kernel-per-level writes to HBM and reads straight back; here it only exists as an LDS
// tile.  Mirror of mifwt_dwt2_fwd_pair.hip, built on the level tile kernel (mifwt_idwt2_tile.h): a 256-thread workgroup
// owns 32 output rows x 2 * NQ output columns of the FINER level (NQ = 64 - (L/2 - 1) coefficient columns) and
//   A. requests, in one burst, the three detail tiles of the finer level and the four (19 + ...)/2 x ~36 coefficient
//      tiles of the COARSER level that its approximation tile depends on;
//   B. synthesises the approximation tile (19 rows x 64 columns for L = 8) from the coarser level — vertical pass, then
//      horizontal pass, the same polyphase formulas and operation order as a stand-alone level — into the LDS slot the
//      level kernel loads its band 0 into;
//   C. continues exactly like the level kernel (vertical synthesis, horizontal synthesis, 8-byte stores).
// Synthesis halos are small (L/2 - 1 coefficients per level and axis), so the approximation tile costs about a quarter of
// the tile's arithmetic on top; the coarser level's scratch tiles live in the LDS array phase C overwrites later.
// Results are bit-identical to two per-level launches (tests/test_idwt_pair_model.py models the tile geometry on the CPU).
// f32, even L <= 8.
// Algorithmic traffic: 4 B (4 M2h M2w + 3 M1h M1w) read + 4 B H W written.
#include "mifwt_idwt2_tile.h"

namespace mifwt {

template <int L>
struct Idwt2PairArgs {
  const float* in2[4];  // coarser level: bands aa, ad, da, dd
  const float* in1[3];  // finer level: details ad, da, dd
  float* y;
  int64_t i2s_b[4], i1s_b, ys_b;  // image strides (elements)
  int i2s_h[4], i1s_h, ys_h;      // row strides
  int M2h, M2w;                   // coarser coefficient extents
  int M1h, M1w;                   // finer coefficient extents = (cropped) output extents of the coarser level
  int H, W;                       // output extents (already trimmed)
  int tiles_c, tiles_r;
  FastDiv div_c, div_r;
  f2 tlo[L / 2];  // (rec_lo[2j], rec_lo[2j+1])
  f2 thi[L / 2];  // (rec_hi[2j], rec_hi[2j+1])
};

constexpr int kPairTRO = 32;

template <int L>
__global__ void __launch_bounds__(256, idwt_tile_occupancy(L, kPairTRO)) idwt2_pair_kernel(const Idwt2PairArgs<L> a) {
  constexpr int TRO = kPairTRO;
  constexpr int HL = L / 2;
  constexpr int NQ = 64 - (HL - 1);      // finer coefficient columns whose outputs a tile stores
  constexpr int CR = TRO / 2 + HL - 1;   // finer coefficient rows of a tile
  constexpr int RPW = (CR + 3) / 4;
  constexpr int PPW = TRO / 8;
  constexpr int NR2 = (CR - 1) / 2 + HL;  // coarser coefficient rows under the approximation tile (its first row is even)
  constexpr int NC2 = 32 + HL;            // coarser coefficient columns (the tile's first column may be odd)
  constexpr int P2 = 44;                  // pitch of the coarser scratch tiles
  constexpr int NP2 = (CR + 1) / 2;       // row pairs of the approximation tile
  constexpr int RPW2 = (NR2 + 3) / 4;
  static_assert(NC2 <= P2 && NC2 <= 64, "coarser tile width");
  __shared__ __attribute__((aligned(16))) float ct[4][CR][64];  // finer coefficient tiles; [0] is computed here
  __shared__ __attribute__((aligned(16))) f2 xt[TRO][64];       // phase C's vertical image; before that: c2, xt2
  float* const c2 = reinterpret_cast<float*>(&xt[0][0]);                  // [4][NR2][P2]
  f2* const xt2 = reinterpret_cast<f2*>(c2 + 4 * NR2 * P2);               // [2 NP2][P2]
  static_assert((4 * NR2 * P2) % 2 == 0 && 4 * NR2 * P2 + 2 * (2 * NP2 * P2) <= TRO * 64 * 2, "coarser scratch must fit into xt");

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  __builtin_assume(wave >= 0 && wave < 4);
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  uint32_t utc, utr;
  const int img = (int)a.div_r.divmod(a.div_c.divmod((uint32_t)bid, utc), utr);
  const int q0 = (int)utc * NQ;   // first finer coefficient column
  const int y0 = (int)utr * TRO;  // first output row
  const int m0 = y0 >> 1;         // first finer coefficient row (even: TRO / 2 is)
  const int p0r = m0 >> 1, p0c = q0 >> 1;  // first coarser coefficient row / column

  // ---- A. loads: finer details -> ct[1..3], coarser bands -> c2 ---------------------------------------------------------
  constexpr uint32_t kOob = 0x80000000u;
  const int qc = q0 + lane;
  const uint32_t coff1 = qc < a.M1w ? 4u * (uint32_t)qc : kOob;
  float v1[3][RPW];
  {
    const uint32_t bytes = ((uint32_t)(a.M1h - 1) * (uint32_t)a.i1s_h + (uint32_t)a.M1w) * 4u;
    const uint32_t row_bytes = (uint32_t)a.i1s_h * 4u;
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      const __amdgpu_buffer_rsrc_t rs =
          __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in1[s] + (int64_t)img * a.i1s_b), 0, bytes, 0x00020000);
#pragma unroll
      for (int i = 0; i < RPW; ++i) {
        const int m = m0 + wave + 4 * i;
        v1[s][i] = idwt_tile_load<float>(rs, (wave + 4 * i < CR && m < a.M1h) ? coff1 : kOob, (uint32_t)(m < a.M1h ? m : 0) * row_bytes);
      }
    }
  }
  const int pc = p0c + lane;
  const uint32_t coff2 = (lane < NC2 && pc < a.M2w) ? 4u * (uint32_t)pc : kOob;
  float v2[4][RPW2];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const uint32_t bytes = ((uint32_t)(a.M2h - 1) * (uint32_t)a.i2s_h[s] + (uint32_t)a.M2w) * 4u;
    const __amdgpu_buffer_rsrc_t rs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in2[s] + (int64_t)img * a.i2s_b[s]), 0, bytes, 0x00020000);
    const uint32_t row_bytes = (uint32_t)a.i2s_h[s] * 4u;
#pragma unroll
    for (int i = 0; i < RPW2; ++i) {
      const int p = p0r + wave + 4 * i;
      v2[s][i] = idwt_tile_load<float>(rs, (wave + 4 * i < NR2 && p < a.M2h) ? coff2 : kOob, (uint32_t)(p < a.M2h ? p : 0) * row_bytes);
    }
  }
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int i = 0; i < RPW2; ++i)
      if (wave + 4 * i < NR2 && lane < NC2) c2[(s * NR2 + wave + 4 * i) * P2 + lane] = v2[s][i];
#pragma unroll
  for (int s = 0; s < 3; ++s)
#pragma unroll
    for (int i = 0; i < RPW; ++i)
      if (wave + 4 * i < CR) ct[1 + s][wave + 4 * i][lane] = v1[s][i];
  __syncthreads();

  // ---- B1. coarser level, vertical synthesis: xt2[2pp + r][c] = (X_lo, X_hi) of approximation row m0 + 2pp + r ----------
#pragma unroll
  for (int j = 0; j < (NP2 + 3) / 4; ++j) {
    const int pp = wave + 4 * j;
    if (pp < NP2 && lane < NC2) {
      f2 xl, xh;
#pragma unroll
      for (int i = 0; i < HL; ++i) {
        const f2 tl = a.tlo[HL - 1 - i], th = a.thi[HL - 1 - i];
        const f2 caa = {c2[(0 * NR2 + pp + i) * P2 + lane], c2[(1 * NR2 + pp + i) * P2 + lane]};  // .x = aa, .y = ad
        const f2 cda = {c2[(2 * NR2 + pp + i) * P2 + lane], c2[(3 * NR2 + pp + i) * P2 + lane]};  // .x = da, .y = dd
        if (i == 0) {
          xl = pkmul_lo(tl, caa);
          xh = pkmul_hi(tl, caa);
        } else {
          pkfma_lo(xl, tl, caa);
          pkfma_hi(xh, tl, caa);
        }
        pkfma_lo(xl, th, cda);
        pkfma_hi(xh, th, cda);
      }
      xt2[(2 * pp) * P2 + lane] = (f2){xl.x, xh.x};
      xt2[(2 * pp + 1) * P2 + lane] = (f2){xl.y, xh.y};
    }
  }
  __syncthreads();

  // ---- B2. coarser level, horizontal synthesis -> ct[0]: lane cl -> approximation columns 2 (p0c + cl), + 1 ---------------
  {
    const int xo = 2 * lane - (q0 & 1);  // tile column of the first of the lane's two outputs
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
      const int r = wave + 4 * i;
      if (r < CR && lane <= 32) {
        f2 o;
#pragma unroll
        for (int t = 0; t < HL; ++t) {
          const f2 w = xt2[r * P2 + lane + t];
          if (t == 0) {
            o = pkmul_lo(a.tlo[HL - 1], w);
          } else {
            pkfma_lo(o, a.tlo[HL - 1 - t], w);
          }
          pkfma_hi(o, a.thi[HL - 1 - t], w);
        }
        if (xo >= 0 && xo < 64) ct[0][r][xo] = o.x;
        if (xo + 1 >= 0 && xo + 1 < 64) ct[0][r][xo + 1] = o.y;
      }
    }
  }
  __syncthreads();  // the approximation tile is complete; c2 / xt2 are dead (xt is rewritten below)

  // ---- C. finer level: vertical synthesis, horizontal synthesis, stores (as idwt2_tile_kernel) ------------------------------
#pragma unroll
  for (int j = 0; j < PPW; ++j) {
    const int pp = wave * PPW + j;
    f2 xl, xh;
#pragma unroll
    for (int i = 0; i < HL; ++i) {
      const f2 tl = a.tlo[HL - 1 - i], th = a.thi[HL - 1 - i];
      const f2 caa = {ct[0][pp + i][lane], ct[1][pp + i][lane]};
      const f2 cda = {ct[2][pp + i][lane], ct[3][pp + i][lane]};
      if (i == 0) {
        xl = pkmul_lo(tl, caa);
        xh = pkmul_hi(tl, caa);
      } else {
        pkfma_lo(xl, tl, caa);
        pkfma_hi(xh, tl, caa);
      }
      pkfma_lo(xl, th, cda);
      pkfma_hi(xh, th, cda);
    }
    xt[2 * pp][lane] = (f2){xl.x, xh.x};
    xt[2 * pp + 1][lane] = (f2){xl.y, xh.y};
  }
  __syncthreads();

  const int x = 2 * (q0 + lane);
  const bool lane_on = lane < NQ && x < a.W;
  float* const ybase = a.y + (int64_t)img * a.ys_b;
#pragma unroll
  for (int j = 0; j < TRO / 4; ++j) {
    const int r = wave * (TRO / 4) + j;
    f2 o;
#pragma unroll
    for (int i = 0; i < HL; ++i) {
      const f2 w = xt[r][(lane + i) & 63];
      if (i == 0) {
        o = pkmul_lo(a.tlo[HL - 1], w);
      } else {
        pkfma_lo(o, a.tlo[HL - 1 - i], w);
      }
      pkfma_hi(o, a.thi[HL - 1 - i], w);
    }
    const int yr = y0 + r;
    if (lane_on && yr < a.H) {
      float* dst = ybase + yr * a.ys_h + x;
      typedef float pair_t __attribute__((ext_vector_type(2), aligned(4)));
      if (x + 1 < a.W)
        *reinterpret_cast<pair_t*>(dst) = (pair_t){o.x, o.y};
      else
        dst[0] = o.x;
    }
  }
}

bool dwt2_inv_pair_supported(const mifwt_level_desc* d2, const mifwt_level_desc* d1) {
  if (g_options[MIFWT_OPT_FORCE_GENERIC] || g_options[MIFWT_OPT_PAIR_MODE] == 2) return false;
  if (d1->ndim != 2 || d2->ndim != 2 || d1->dtype != MIFWT_F32 || d2->dtype != MIFWT_F32) return false;
  const int L = d1->filt_len;
  if (d2->filt_len != L || L < 2 || L > 8 || (L & 1) || d1->batch != d2->batch) return false;
  for (int i = 0; i < 2; ++i)
    if (d2->sig_extent[i] != d1->coef_extent[i]) return false;  // the coarser level's (cropped) output is the finer approximation
  if (d1->sig_stride[2] != 1 || d1->detail_stride[2] != 1 || d2->approx_stride[2] != 1 || d2->detail_stride[2] != 1) return false;
  for (int i = 0; i < 2; ++i)
    if (d1->sig_stride[i] < 0 || d1->detail_stride[i] < 0 || d2->approx_stride[i] < 0 || d2->detail_stride[i] < 0) return false;
  const int64_t lim = int64_t(1) << 29;  // 32-bit byte offsets inside one image of every band
  if ((d1->coef_extent[0] - 1) * d1->detail_stride[1] + d1->coef_extent[1] >= lim) return false;
  if ((d2->coef_extent[0] - 1) * d2->approx_stride[1] + d2->coef_extent[1] >= lim) return false;
  if ((d2->coef_extent[0] - 1) * d2->detail_stride[1] + d2->coef_extent[1] >= lim) return false;
  if (d1->sig_extent[0] * d1->sig_stride[1] >= (int64_t(1) << 31)) return false;
  // tiles are 32 output rows tall: the fusion pays on planes of at least a few tiles (measured on 1024^2)
  return d1->sig_extent[0] >= 64 && d1->sig_extent[1] >= 64;
}

template <int L>
static int launch_idwt_pair(const mifwt_level_desc* d2, const mifwt_level_desc* d1, const void* approx2, const void* const* details2,
                            const void* const* details1, void* y, const double* lo, const double* hi, hipStream_t stream) {
  constexpr int NQ = 64 - (L / 2 - 1);
  Idwt2PairArgs<L> a;
  a.in2[0] = static_cast<const float*>(approx2);
  for (int s = 1; s < 4; ++s) a.in2[s] = static_cast<const float*>(details2[s - 1]);
  for (int s = 0; s < 3; ++s) a.in1[s] = static_cast<const float*>(details1[s]);
  for (int s = 0; s < 4; ++s) {
    a.i2s_b[s] = s == 0 ? d2->approx_stride[0] : d2->detail_stride[0];
    a.i2s_h[s] = (int)(s == 0 ? d2->approx_stride[1] : d2->detail_stride[1]);
  }
  a.i1s_b = d1->detail_stride[0];
  a.i1s_h = (int)d1->detail_stride[1];
  a.y = static_cast<float*>(y);
  a.ys_b = d1->sig_stride[0];
  a.ys_h = (int)d1->sig_stride[1];
  a.M2h = (int)d2->coef_extent[0];
  a.M2w = (int)d2->coef_extent[1];
  a.M1h = (int)d1->coef_extent[0];
  a.M1w = (int)d1->coef_extent[1];
  a.H = (int)d1->sig_extent[0];
  a.W = (int)d1->sig_extent[1];
  for (int j = 0; j < L / 2; ++j) {
    a.tlo[j] = (f2){(float)lo[2 * j], (float)lo[2 * j + 1]};
    a.thi[j] = (f2){(float)hi[2 * j], (float)hi[2 * j + 1]};
  }
  a.tiles_c = (a.W + 2 * NQ - 1) / (2 * NQ);
  a.tiles_r = (a.H + kPairTRO - 1) / kPairTRO;
  a.div_c = make_fastdiv((uint32_t)a.tiles_c);
  a.div_r = make_fastdiv((uint32_t)a.tiles_r);
  const int64_t ntiles = (int64_t)d1->batch * a.tiles_c * a.tiles_r;
  if (ntiles > INT32_MAX / 8) return MIFWT_ERR_UNSUPPORTED;
  hipLaunchKernelGGL((idwt2_pair_kernel<L>), dim3((unsigned)ntiles), dim3(256), 0, stream, a);
  return hipGetLastError() == hipSuccess ? MIFWT_OK : MIFWT_ERR_LAUNCH;
}

int dwt2_inv_pair(const mifwt_level_desc* d2, const mifwt_level_desc* d1, const void* approx2, const void* const* details2,
                  const void* const* details1, void* y, const double* lo, const double* hi, hipStream_t stream) {
  if (!dwt2_inv_pair_supported(d2, d1)) return MIFWT_ERR_UNSUPPORTED;
  switch (d1->filt_len) {
    case 2: return launch_idwt_pair<2>(d2, d1, approx2, details2, details1, y, lo, hi, stream);
    case 4: return launch_idwt_pair<4>(d2, d1, approx2, details2, details1, y, lo, hi, stream);
    case 6: return launch_idwt_pair<6>(d2, d1, approx2, details2, details1, y, lo, hi, stream);
    case 8: return launch_idwt_pair<8>(d2, d1, approx2, details2, details1, y, lo, hi, stream);
    default: return MIFWT_ERR_UNSUPPORTED;
  }
}

}  // namespace mifwt
