// mifwt_idwt2_tile.h — LDS-tiled fused 2-D synthesis level (gfx950), kernel id 8.
//
// Replaces, for one level of waverec2 / fswaverec2: torch.stack + F.conv_transpose2d([4,1,L,L], stride 2) + the four
// crops (reference src/ptwt/conv_transform_2.py:222-249) — separably, in polyphase (gather) form (only the L/2
// non-zero products per output sample, only the cropped interior), mirror of mifwt_dwt2_tile.h:
//   per axis, output index n = 2p + r:   y[2p + r] = sum_{i < L/2} g_lo[L-2-2i+r] a[p+i] + g_hi[L-2-2i+r] d[p+i]
// A 256-thread workgroup owns TRO output rows x 2*NQ output columns (NQ = 64 - (L/2 - 1) coefficient columns, so that
// the NQ + L/2 - 1 coefficient columns a tile needs are one lane each):
//   1. all four waves request the four bands' (TRO/2 + L/2 - 1) x (NQ + L/2 - 1) coefficient tiles in one burst -> LDS;
//   2. vertical synthesis LDS -> LDS: lane = coefficient column, X_lo from (aa, da), X_hi from (ad, dd), output-row pair
//      per step with the tap PAIRS (g[2j], g[2j+1]) packed (v_pk_fma_f32, accumulator = (row 2p, row 2p+1));
//   3. horizontal synthesis LDS -> registers -> global: lane = output column pair, one 8-byte store per row.
// Algorithmic traffic: 4*4*B*Mh*Mw read + 4*B*H*W written (f32; f16 storage: half of that).
#pragma once
#include "mifwt_stream.h"

namespace mifwt {

template <typename T, int L>
struct Idwt2TileArgs {
  const T* in[4];  // bands aa, ad, da, dd
  T* y;
  int64_t is_b[4], is_h[4];  // band strides (elements); innermost stride 1
  int64_t ys_b, ys_h;
  int Mh, Mw;  // coefficient extents
  int H, W;    // output extents (already trimmed: 2M - L + 2 - t)
  int tiles_c, tiles_r, ntiles;
  FastDiv div_c, div_r;  // by tiles_c, tiles_r
  typename TileArith<T>::vec2 tlo[L / 2];  // (rec_lo[2j], rec_lo[2j+1]) in the arithmetic type
  typename TileArith<T>::vec2 thi[L / 2];  // (rec_hi[2j], rec_hi[2j+1])
  DevTapArg dt;                            // device-resident taps (mifwt_common.h); dt.lo == nullptr: tlo / thi count
};

constexpr int idwt_tile_occupancy(int L, int TRO, int esz = 4) {
  const int cr = TRO / 2 + L / 2 - 1;
  const int lds = (4 * cr + 2 * TRO) * 64 * esz;
  const int n = (160 * 1024) / lds;
  return n > 8 ? 8 : (n < 1 ? 1 : n);
}

template <typename T>
__device__ __forceinline__ typename TileArith<T>::type idwt_tile_load(__amdgpu_buffer_rsrc_t rsrc, uint32_t voff, uint32_t soff);
template <>
__device__ __forceinline__ double idwt_tile_load<double>(__amdgpu_buffer_rsrc_t rsrc, uint32_t voff, uint32_t soff) {
  return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rsrc, voff, soff, 0));
}
template <>
__device__ __forceinline__ float idwt_tile_load<float>(__amdgpu_buffer_rsrc_t rsrc, uint32_t voff, uint32_t soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, soff, 0));
}
template <>
__device__ __forceinline__ float idwt_tile_load<_Float16>(__amdgpu_buffer_rsrc_t rsrc, uint32_t voff, uint32_t soff) {
  return (float)__builtin_bit_cast(_Float16, (unsigned short)__builtin_amdgcn_raw_buffer_load_b16(rsrc, voff, soff, 0));
}

template <typename T, int L, int TRO>
__global__ void __launch_bounds__(256, idwt_tile_occupancy(L, TRO, sizeof(typename TileArith<T>::type))) idwt2_tile_kernel(const Idwt2TileArgs<T, L> a) {
  constexpr uint32_t ES = sizeof(T);
  typedef typename TileArith<T>::type A;   // arithmetic / LDS element type
  typedef typename TileArith<T>::vec2 A2;
  constexpr int HL = L / 2;
  constexpr int NQ = 64 - (HL - 1);      // coefficient columns whose outputs a tile stores
  constexpr int CR = TRO / 2 + HL - 1;   // coefficient rows of a tile
  constexpr int RPW = (CR + 3) / 4;      // coefficient rows per wave (load phase)
  constexpr int PPW = TRO / 8;           // output row PAIRS per wave (vertical phase): TRO/2 pairs over 4 waves
  static_assert(TRO % 8 == 0, "TRO must be a multiple of 8");
  __shared__ __attribute__((aligned(16))) A ct[4][CR][64];  // coefficient tiles
  __shared__ __attribute__((aligned(16))) A2 xt[TRO][64];       // (X_lo, X_hi) per output row and coefficient column

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  uint32_t utc, utr;
  const int img = (int)a.div_r.divmod(a.div_c.divmod((uint32_t)bid, utc), utr);
  const int tc = (int)utc, tr = (int)utr;
  __builtin_assume(wave >= 0 && wave < 4);
  // the taps: by value, or (a learnable filter bank that lives on the GPU) read once from device memory
  A2 tlo[HL], thi[HL];
  if (a.dt.lo) {
#pragma unroll
    for (int j = 0; j < HL; ++j) {
      tlo[j] = (A2){dtap_lo<A>(a.dt, 2 * j), dtap_lo<A>(a.dt, 2 * j + 1)};
      thi[j] = (A2){dtap_hi<A>(a.dt, 2 * j), dtap_hi<A>(a.dt, 2 * j + 1)};
    }
  } else {
#pragma unroll
    for (int j = 0; j < HL; ++j) tlo[j] = a.tlo[j], thi[j] = a.thi[j];
  }
  const int q0 = tc * NQ;        // first coefficient column
  const int y0 = tr * TRO;       // first output row (even)
  const int m0 = y0 >> 1;        // first coefficient row

  // ---- 1. coefficient tiles -> LDS -------------------------------------------------------------------------------------
  constexpr uint32_t kOob = 0x80000000u;
  const int qc = q0 + lane;
  const uint32_t coff = qc < a.Mw ? ES * (uint32_t)qc : kOob;  // columns past the band read 0 (never used by a stored output)
  A v[4][RPW];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const uint32_t bytes = ((uint32_t)(a.Mh - 1) * (uint32_t)a.is_h[s] + (uint32_t)a.Mw) * ES;
    const __amdgpu_buffer_rsrc_t rs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(a.in[s] + (int64_t)img * a.is_b[s]), 0, bytes, 0x00020000);
    const uint32_t row_bytes = (uint32_t)a.is_h[s] * ES;
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
      const int m = m0 + wave + 4 * i;  // wave-uniform
      v[s][i] = idwt_tile_load<T>(rs, (wave + 4 * i < CR && m < a.Mh) ? coff : kOob, (uint32_t)(m < a.Mh ? m : 0) * row_bytes);
    }
  }
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int i = 0; i < RPW; ++i)
      if (wave + 4 * i < CR) ct[s][wave + 4 * i][lane] = v[s][i];
  __syncthreads();

  // ---- 2. vertical synthesis: xt[2pp + r][c] = (X_lo, X_hi) of output row y0 + 2pp + r at coefficient column c ------------
#pragma unroll
  for (int j = 0; j < PPW; ++j) {
    const int pp = wave * PPW + j;  // output row pair of the tile; coefficient rows pp .. pp + HL - 1
    A2 xl, xh;                      // (row 2pp, row 2pp + 1) of X_lo / X_hi
#pragma unroll
    for (int i = 0; i < HL; ++i) {
      const A2 tl = tlo[HL - 1 - i], th = thi[HL - 1 - i];
      const A2 caa = {ct[0][pp + i][lane], ct[1][pp + i][lane]};  // .x = aa, .y = ad
      const A2 cda = {ct[2][pp + i][lane], ct[3][pp + i][lane]};  // .x = da, .y = dd
      if (i == 0) {
        xl = amul_lo(tl, caa);
        xh = amul_hi(tl, caa);
      } else {
        afma_lo(xl, tl, caa);
        afma_hi(xh, tl, caa);
      }
      afma_lo(xl, th, cda);
      afma_hi(xh, th, cda);
    }
    xt[2 * pp][lane] = (A2){xl.x, xh.x};
    xt[2 * pp + 1][lane] = (A2){xl.y, xh.y};
  }
  __syncthreads();

  // ---- 3. horizontal synthesis + stores: wave w owns output rows y0 + w*TRO/4 .. ; lane -> output columns 2(q0+lane), +1 -------
  const int x = 2 * (q0 + lane);
  const bool lane_on = lane < NQ && x < a.W;
#pragma unroll
  for (int j = 0; j < TRO / 4; ++j) {
    const int r = wave * (TRO / 4) + j;
    A2 o;  // (y[x], y[x + 1])
#pragma unroll
    for (int i = 0; i < HL; ++i) {
      const A2 w = xt[r][(lane + i) & 63];  // lanes >= NQ wrap harmlessly (not stored)
      if (i == 0) {
        o = amul_lo(tlo[HL - 1], w);
      } else {
        afma_lo(o, tlo[HL - 1 - i], w);
      }
      afma_hi(o, thi[HL - 1 - i], w);
    }
    const int yr = y0 + r;
    if (lane_on && yr < a.H) {
      T* dst = a.y + (int64_t)img * a.ys_b + (int64_t)yr * a.ys_h + x;
      typedef T pair_t __attribute__((ext_vector_type(2), aligned(sizeof(T))));
      if (x + 1 < a.W)
        *reinterpret_cast<pair_t*>(dst) = (pair_t){(T)o.x, (T)o.y};
      else
        dst[0] = (T)o.x;
    }
  }
}

template <typename T, int L, int TRO>
int launch_idwt_tile(const mifwt_level_desc* d, const void* approx, const void* const* details, void* y, const double* lo,
                     const double* hi, hipStream_t stream) {
  constexpr int NQ = 64 - (L / 2 - 1);
  Idwt2TileArgs<T, L> a;
  a.in[0] = static_cast<const T*>(approx);
  for (int s = 1; s < 4; ++s) a.in[s] = static_cast<const T*>(details[s - 1]);
  for (int s = 0; s < 4; ++s) {
    a.is_b[s] = s == 0 ? d->approx_stride[0] : d->detail_stride[0];
    a.is_h[s] = s == 0 ? d->approx_stride[1] : d->detail_stride[1];
  }
  a.y = static_cast<T*>(y);
  a.ys_b = d->sig_stride[0];
  a.ys_h = d->sig_stride[1];
  a.Mh = (int)d->coef_extent[0];
  a.Mw = (int)d->coef_extent[1];
  a.H = (int)d->sig_extent[0];
  a.W = (int)d->sig_extent[1];
  for (int j = 0; j < L / 2; ++j) {
    typedef typename TileArith<T>::type A;
    a.tlo[j] = (typename TileArith<T>::vec2){(A)lo[2 * j], (A)lo[2 * j + 1]};
    a.thi[j] = (typename TileArith<T>::vec2){(A)hi[2 * j], (A)hi[2 * j + 1]};
  }
  a.dt = dev_tap_arg(L);
  a.tiles_c = (a.W + 2 * NQ - 1) / (2 * NQ);
  a.tiles_r = (a.H + TRO - 1) / TRO;
  a.div_c = make_fastdiv((uint32_t)a.tiles_c);
  a.div_r = make_fastdiv((uint32_t)a.tiles_r);
  const int64_t ntiles = (int64_t)d->batch * a.tiles_c * a.tiles_r;
  if (ntiles > INT32_MAX / 8) return MIFWT_ERR_UNSUPPORTED;
  a.ntiles = (int)ntiles;
  hipLaunchKernelGGL((idwt2_tile_kernel<T, L, TRO>), dim3((unsigned)ntiles), dim3(256), 0, stream, a);
  return hipGetLastError() == hipSuccess ? MIFWT_OK : MIFWT_ERR_LAUNCH;
}

template <typename T, int L>
int launch_idwt_tr(const mifwt_level_desc* d, const void* approx, const void* const* details, void* y, const double* lo,
                   const double* hi, hipStream_t stream) {
  constexpr int NQ = 64 - (L / 2 - 1);
  int tro = g_options[MIFWT_OPT_TILE_ROWS];
  if (tro <= 0) {
    const int64_t tiles_c = (d->sig_extent[1] + 2 * NQ - 1) / (2 * NQ);
    // several rounds anyway: taller tiles = longer load bursts (32 rows measured best on 1024^2 in f32; f64 tiles of 32 rows leave fewer
    // workgroups per CU: waverec2 db4 level 3 on 64 x 1024^2 f64 8 / 16 / 24 / 32 rows = 291 / 284 / 322 / 306-311 us, profiles/r05zz_f64_tile_rows.txt)
    tro = sizeof(typename TileArith<T>::type) == 8 ? 16 : 32;
    for (int cand = 8; cand <= 32; cand += 8) {  // smallest tile height whose grid is resident in one round
      const int64_t blocks = d->batch * tiles_c * ((d->sig_extent[0] + cand - 1) / cand);
      if (blocks <= 256 * idwt_tile_occupancy(L, cand, sizeof(typename TileArith<T>::type))) {
        tro = cand;
        break;
      }
    }
  }
  if (tro <= 8) return launch_idwt_tile<T, L, 8>(d, approx, details, y, lo, hi, stream);
  if (tro <= 16) return launch_idwt_tile<T, L, 16>(d, approx, details, y, lo, hi, stream);
  if (tro <= 24) return launch_idwt_tile<T, L, 24>(d, approx, details, y, lo, hi, stream);
  return launch_idwt_tile<T, L, 32>(d, approx, details, y, lo, hi, stream);
}

}  // namespace mifwt
