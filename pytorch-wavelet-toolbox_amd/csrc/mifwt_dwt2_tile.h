// mifwt_dwt2_tile.h — LDS-tiled fused 2-D analysis level (gfx950), kernel id 7.
//
// Same maths and the same reference seam as mifwt_dwt2_fwd.hip (F.pad + F.conv2d([4,1,L,L], stride 2) + split,
// src/ptwt/conv_transform_2.py:142-149), different shape of parallelism.  The streaming kernel walks down
// 256-column strips with the vertical filter window in registers; its per-wave critical path (ring prefill, then a
// few row pairs) dominates once a plane is only a few hundred pixels wide — levels 2-3 of a pyramid ran at 3.6 and
// 2.1 TB/s against 4.7 TB/s on level 1.  Here a 256-thread workgroup owns an output tile of TR x 64 coefficients:
//   1. all four waves request the (2 TR + L - 2) x (2*64 + L - 2) input tile, boundary extension applied as an
//      index map per row / column, in ONE burst (every load of the tile is in flight before the first is needed)
//      and park it in LDS;
//   2. horizontal pass LDS -> LDS, IN PLACE (row r of the (lo, hi) image overwrites row r of the input tile: it is
//      shorter, and a row is read and written by one wave only, whose DS operations execute in order): lane = output
//      column, 8-byte conflict-free reads, (lo, hi) packed v_pk_fma_f32;
//   3. vertical pass LDS -> registers -> global: lane = output column, wave = a block of output rows, 16 packed
//      FMAs per coefficient position, one coalesced 256-byte store per band and row.
// Algorithmic traffic: 4*B*H*W read + 4*4*B*Ho*Wo written; the tile halo ((2 TR + L - 2) / 2 TR rows,
// (128 + L - 2) / 128 columns) is re-read through L2.
#pragma once
#include "mifwt_stream.h"

namespace mifwt {

constexpr int kTC = 64;  // output columns per tile = lanes

template <typename T, int L>
struct Dwt2TileArgs {
  const T* x;
  T* out[4];  // bands aa, ad, da, dd
  int64_t xs_b, xs_h;
  int64_t xs_outer;  // two-level batch of the INPUT (the depth slices of the volumes of a 3-D level whose batch stride is not depth x
  FastDiv div_in;    // slice stride): image i = outer (i / inner) x xs_outer + (i % inner) x xs_b; div_in.d == 0: one level
  int64_t os_b[4], os_h[4];
  int H, W, Ho, Wo;
  int tiles_c, tiles_r, ntiles;
  FastDiv div_c, div_r;  // by tiles_c, tiles_r
  int mode;
  int sync_stage;
  typename TileArith<T>::vec2 tap[L];  // (dec_lo[m], dec_hi[m]) in the arithmetic type
  DevTapArg dt;                        // device-resident taps (mifwt_common.h); dt.lo == nullptr: `tap` counts
};

// element offset of input image `img` (one- or two-level batch)
template <typename T, int L>
__device__ __forceinline__ int64_t tile_image_offset(const Dwt2TileArgs<T, L>& a, int img) {
  if (a.div_in.d == 0) return (int64_t)img * a.xs_b;
  uint32_t in;
  const uint32_t out = a.div_in.divmod((uint32_t)img, in);
  return (int64_t)out * a.xs_outer + (int64_t)in * a.xs_b;
}

// workgroups per CU that the tile's LDS footprint admits = waves per SIMD to allocate registers for (a 256-thread
// workgroup puts one wave on each SIMD)
constexpr int tile_occupancy(int L, int TR, int esz = 4) {
  const int lds = (2 * TR + L - 2) * ((2 * kTC + L - 2 + 1) & ~1) * esz;
  const int n = (160 * 1024) / lds;
  return n > 8 ? 8 : (n < 1 ? 1 : n);
}

// one element of storage type T through the buffer resource (byte offset), widened to float
template <typename T>
__device__ __forceinline__ typename TileArith<T>::type tile_load(__amdgpu_buffer_rsrc_t rsrc, uint32_t voff, uint32_t soff);
template <>
__device__ __forceinline__ double tile_load<double>(__amdgpu_buffer_rsrc_t rsrc, uint32_t voff, uint32_t soff) {
  return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rsrc, voff, soff, 0));
}
template <>
__device__ __forceinline__ float tile_load<float>(__amdgpu_buffer_rsrc_t rsrc, uint32_t voff, uint32_t soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, soff, 0));
}
template <>
__device__ __forceinline__ float tile_load<_Float16>(__amdgpu_buffer_rsrc_t rsrc, uint32_t voff, uint32_t soff) {
  return (float)__builtin_bit_cast(_Float16, (unsigned short)__builtin_amdgcn_raw_buffer_load_b16(rsrc, voff, soff, 0));
}

template <typename T, int L, int TR>
__global__ void __launch_bounds__(256, tile_occupancy(L, TR, sizeof(typename TileArith<T>::type))) dwt2_fwd_tile_kernel(const Dwt2TileArgs<T, L> a) {
  constexpr uint32_t ES = sizeof(T);
  typedef typename TileArith<T>::type A;   // arithmetic / LDS element type
  typedef typename TileArith<T>::vec2 A2;  // (lo, hi) pair
  constexpr int IR = 2 * TR + L - 2;    // input rows of a tile
  constexpr int IC = 2 * kTC + L - 2;   // input columns of a tile
  constexpr int XP = (IC + 1) & ~1;     // LDS pitch of the tile (floats, even: 8-byte aligned pairs; >= 2 * kTC)
  constexpr int NQ = (IC + 63) / 64;    // column loads per lane and row
  constexpr int RW = TR / 4;            // output rows per wave in the vertical pass
  __shared__ __attribute__((aligned(16))) A xt[IR * XP];
  static_assert(XP >= 2 * kTC, "the (lo, hi) row image must fit into the input row it replaces");

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  uint32_t utc, utr;
  const int img = (int)a.div_r.divmod(a.div_c.divmod((uint32_t)bid, utc), utr);
  const int tc = (int)utc, tr = (int)utr;
  const int k0 = tc * kTC, j0 = tr * TR;
  // the taps: by value, or (a learnable filter bank that lives on the GPU) read once from device memory
  A2 tapv[L];
  if (a.dt.lo) {
#pragma unroll
    for (int m = 0; m < L; ++m) tapv[m] = (A2){dtap_lo<A>(a.dt, m), dtap_hi<A>(a.dt, m)};
  } else {
#pragma unroll
    for (int m = 0; m < L; ++m) tapv[m] = a.tap[m];
  }

  // ---- 1. input tile -> LDS ------------------------------------------------------------------------------------------
  const uint32_t img_bytes = ((uint32_t)(a.H - 1) * (uint32_t)a.xs_h + (uint32_t)a.W) * ES;
  const __amdgpu_buffer_rsrc_t xrsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(a.x + tile_image_offset(a, img)), 0, img_bytes, 0x00020000);
  constexpr uint32_t kOob = 0x80000000u;  // >= num_records: the load returns 0 without a memory request
  // columns / rows the tile's real outputs need (ragged last tiles request nothing beyond them)
  const int nc_need = 2 * (min(k0 + kTC, a.Wo) - k0) + L - 2;
  const int nr_need = 2 * (min(j0 + TR, a.Ho) - j0) + L - 2;
  const int c_first = 2 * k0 - (L - 2), r_first = 2 * j0 - (L - 2);
  // Boundary extension = the branch-free single-fold map (the launcher routes planes shorter than the filter elsewhere).
  // A tile whose rows lie inside the image skips the per-row map: one scalar add per row.
  __builtin_assume(wave >= 0 && wave < 4);
  const bool zero_mode = a.mode == MIFWT_MODE_ZERO;
  Fold1 fold;
  fold.set(a.mode);
  const bool rows_inside = r_first >= 0 && r_first + IR <= a.H;
  const uint32_t row_bytes = (uint32_t)a.xs_h * ES;
  constexpr int RPW = (IR + 3) / 4;  // rows per wave
  uint32_t coff[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int c = lane + 64 * q, ci = c_first + c;
    const bool dead = c >= nc_need || (zero_mode && (unsigned)ci >= (unsigned)a.W);
    coff[q] = dead ? kOob : ES * (uint32_t)fold(ci, a.W);
  }
  A v[RPW][NQ];
  if (rows_inside) {
    uint32_t soff = (uint32_t)(r_first + wave) * row_bytes;
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
      // rows beyond the tile (wave + 4 i >= IR, last i only) re-read the tile's last row: in range, never used
      const uint32_t so = (4 * i + 3 < IR || wave + 4 * i < IR) ? soff : (uint32_t)(r_first + IR - 1) * row_bytes;
#pragma unroll
      for (int q = 0; q < NQ; ++q) v[i][q] = tile_load<T>(xrsrc, coff[q], so);
      soff += 4u * row_bytes;
    }
  } else {
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
      const int r = wave + 4 * i, ri = r_first + r;
      const bool dead = r >= nr_need || (zero_mode && (unsigned)ri >= (unsigned)a.H);
      const uint32_t soff = __builtin_amdgcn_readfirstlane(dead ? 0u : (uint32_t)fold(ri, a.H) * row_bytes);
#pragma unroll
      for (int q = 0; q < NQ; ++q) v[i][q] = tile_load<T>(xrsrc, dead ? kOob : coff[q], soff);
    }
  }
#pragma unroll
  for (int i = 0; i < RPW; ++i) {
    const int r = wave + 4 * i;
    if (4 * i + 3 < IR || r < IR) {  // first clause: compile time, true for every i but possibly the last
#pragma unroll
      for (int q = 0; q < NQ; ++q)
        if (64 * q + 63 < XP || lane + 64 * q < XP) xt[r * XP + lane + 64 * q] = v[i][q];
    }
  }
  // no workgroup barrier here: row r is staged, filtered and overwritten by the same wave (rows wave + 4 i), whose DS
  // operations execute in order; the option keeps the barrier for A/B measurements
  if (a.sync_stage) __syncthreads(); else wave_lds_fence();

  // ---- 2. horizontal pass, in place: row r becomes (lo, hi)[k] of output column k0 + k ---------------------------------
  // c[k] = sum_m h[m] x_ext[2k + 1 - m]; tile column of x_ext[2k + 1 - m] is 2k + (L - 1) - m
#pragma unroll
  for (int i = 0; i < RPW; ++i) {
    const int r = wave + 4 * i;
    if (4 * i + 3 < IR || r < IR) {
      const A2* row = reinterpret_cast<const A2*>(&xt[r * XP + 2 * lane]);
      A2 acc;
#pragma unroll
      for (int p = 0; p < L / 2; ++p) {
        const A2 xx = row[p];  // tile columns 2k + 2p, 2k + 2p + 1  <->  taps L-1-2p, L-2-2p
        if (p == 0) {
          acc = amul_lo(tapv[L - 1], xx);
        } else {
          afma_lo(acc, tapv[L - 1 - 2 * p], xx);
        }
        afma_hi(acc, tapv[L - 2 - 2 * p], xx);
      }
      wave_lds_fence();  // every lane's reads of row r are issued (DS ops of a wave run in order) before its overwrite
      *reinterpret_cast<A2*>(&xt[r * XP + 2 * lane]) = acc;
    }
  }
  __syncthreads();

  // ---- 3. vertical pass + stores: wave w owns output rows j0 + w*RW .. + RW - 1 ----------------------------------------
  const int k = k0 + lane;
  // band bases once per workgroup, row offsets once per row and stride set (bands 1..3 share the detail strides): the
  // scalar unit is shared by the CU, per-store 64-bit address arithmetic was a third of its load on small planes
  T* obase[4];  // wave-uniform (scalar registers); lanes add a 32-bit element offset
#pragma unroll
  for (int s = 0; s < 4; ++s) obase[s] = a.out[s] + (int64_t)img * a.os_b[s];
  A2 win[2 * RW + L - 2];
#pragma unroll
  for (int t = 0; t < 2 * RW + L - 2; ++t) win[t] = *reinterpret_cast<const A2*>(&xt[(2 * wave * RW + t) * XP + 2 * lane]);
#pragma unroll
  for (int i = 0; i < RW; ++i) {
    const int j = j0 + wave * RW + i;
    A2 lo2, hi2;  // lo2 = (aa, da), hi2 = (ad, dd)
#pragma unroll
    for (int m = 0; m < L; ++m) {
      const A2 hv = win[2 * i + (L - 1) - m];  // row 2j + 1 - m of the extended plane
      if (m == 0) {
        lo2 = amul_lo(tapv[0], hv);
        hi2 = amul_hi(tapv[0], hv);
      } else {
        afma_lo(lo2, tapv[m], hv);
        afma_hi(hi2, tapv[m], hv);
      }
    }
    if (j < a.Ho && k < a.Wo) {
      const int off_a = j * (int)a.os_h[0] + k, off_d = j * (int)a.os_h[1] + k;
      obase[0][off_a] = (T)lo2.x;
      obase[1][off_d] = (T)hi2.x;
      obase[2][off_d] = (T)lo2.y;
      obase[3][off_d] = (T)hi2.y;
    }
  }
}

template <typename T, int L, int TR>
int launch_tile(const mifwt_level_desc* d, const void* x, void* approx, void* const* details, const double* lo,
                const double* hi, hipStream_t stream) {
  Dwt2TileArgs<T, L> a;
  a.x = static_cast<const T*>(x);
  a.out[0] = static_cast<T*>(approx);
  for (int s = 1; s < 4; ++s) a.out[s] = static_cast<T*>(details[s - 1]);
  a.xs_b = d->sig_stride[0];
  a.xs_h = d->sig_stride[1];
  a.xs_outer = 0;
  a.div_in.mul = a.div_in.shift = a.div_in.d = 0;
  if (g_batch_split.inner > 0) {  // (set by the 3-D composed route around this one call: mifwt_compose.hip plane3_fwd)
    a.div_in = make_fastdiv((uint32_t)g_batch_split.inner);
    a.xs_outer = g_batch_split.outer_stride;
  }
  for (int s = 0; s < 4; ++s) {
    a.os_b[s] = s == 0 ? d->approx_stride[0] : d->detail_stride[0];
    a.os_h[s] = s == 0 ? d->approx_stride[1] : d->detail_stride[1];
  }
  a.H = (int)d->sig_extent[0];
  a.W = (int)d->sig_extent[1];
  a.Ho = (int)d->coef_extent[0];
  a.Wo = (int)d->coef_extent[1];
  a.mode = d->mode;
  a.sync_stage = g_options[MIFWT_OPT_SYNC_STAGE];
  for (int m = 0; m < L; ++m)
    a.tap[m] = (typename TileArith<T>::vec2){(typename TileArith<T>::type)lo[m], (typename TileArith<T>::type)hi[m]};
  a.dt = dev_tap_arg(L);
  a.tiles_c = (a.Wo + kTC - 1) / kTC;
  a.tiles_r = (a.Ho + TR - 1) / TR;
  a.div_c = make_fastdiv((uint32_t)a.tiles_c);
  a.div_r = make_fastdiv((uint32_t)a.tiles_r);
  const int64_t ntiles = (int64_t)d->batch * a.tiles_c * a.tiles_r;
  if (ntiles > INT32_MAX / 8) return MIFWT_ERR_UNSUPPORTED;
  a.ntiles = (int)ntiles;
  hipLaunchKernelGGL((dwt2_fwd_tile_kernel<T, L, TR>), dim3((unsigned)ntiles), dim3(256), 0, stream, a);
  return hipGetLastError() == hipSuccess ? MIFWT_OK : MIFWT_ERR_LAUNCH;
}

// Tile height for a level: the smallest one whose grid is resident at once (one "round" of workgroups: a second
// round costs a full tile latency again), else 12 rows (measured best on 1024^2 / 515^2 planes for L <= 8;
// taller tiles carry less row halo, which matters for long filters: 16 / 24 rows there).
template <typename T, int L>
int launch_tr(const mifwt_level_desc* d, const void* x, void* approx, void* const* details, const double* lo,
              const double* hi, hipStream_t stream) {
  int tr = g_options[MIFWT_OPT_TILE_ROWS];
  if (tr <= 0) {
    const int64_t tiles_c = (d->coef_extent[1] + kTC - 1) / kTC;
    tr = L <= 10 ? 12 : (L <= 20 ? 16 : 24);
    for (int cand = 8; cand <= 24; cand += 4) {
      const int64_t blocks = d->batch * tiles_c * ((d->coef_extent[0] + cand - 1) / cand);
      const int64_t lds = (int64_t)(2 * cand + L - 2) * (2 * kTC + L - 2) * (int64_t)sizeof(typename TileArith<T>::type);
      int64_t per_cu = (160 * 1024) / lds;
      if (per_cu > 8) per_cu = 8;
      if (blocks <= 256 * per_cu) {
        tr = cand;
        break;
      }
    }
  }
  if constexpr (L <= 16) {
    if (tr <= 8) return launch_tile<T, L, 8>(d, x, approx, details, lo, hi, stream);
    if (tr <= 12) return launch_tile<T, L, 12>(d, x, approx, details, lo, hi, stream);
    if (tr <= 16) return launch_tile<T, L, 16>(d, x, approx, details, lo, hi, stream);
    if (tr <= 20) return launch_tile<T, L, 20>(d, x, approx, details, lo, hi, stream);
    return launch_tile<T, L, 24>(d, x, approx, details, lo, hi, stream);
  } else {  // long filters: fewer instantiations
    if (tr <= 8) return launch_tile<T, L, 8>(d, x, approx, details, lo, hi, stream);
    if (tr <= 16) return launch_tile<T, L, 16>(d, x, approx, details, lo, hi, stream);
    return launch_tile<T, L, 24>(d, x, approx, details, lo, hi, stream);
  }
}

}  // namespace mifwt
