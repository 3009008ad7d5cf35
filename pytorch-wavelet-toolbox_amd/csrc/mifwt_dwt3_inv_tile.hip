// mifwt_dwt3_inv_tile.hip — fully fused LDS-brick 3-D synthesis level for short filters (gfx950), kernel id 10.
//
// Replaces, for one level of waverec3 / fswaverec3: torch.stack + F.conv_transpose3d([8,1,L,L,L], stride 2) + the crops (reference
// src/ptwt/conv_transform_3.py:205-249) — separably, in polyphase (gather) form, reading the eight sub-band volumes once and writing
// the reconstruction once (the composed route of mifwt_compose.hip moves 2x the algorithmic bytes through scratch).  Mirror of the
// slice-per-wave analysis brick (mifwt_dwt3_fwd_tile.hip), passes in the opposite order:
//   per axis, output index n = 2p + r:   y[2p + r] = sum_{i < L/2} g_lo[L-2-2i+r] a[p+i] + g_hi[L-2-2i+r] d[p+i]
// A 256-thread workgroup owns 2*CZ x 2*CY x 2*NQ output samples (depth x rows x columns; CZ = 2, CY = 4, NQ = 64 - (L/2 - 1), so that
// the NQ + L/2 - 1 coefficient columns a brick needs are one lane each):
//   1. D pass, global -> registers -> LDS: a wave takes (coefficient row, (H, W) band) units; per unit the CZ + L/2 - 1 slices of the
//      depth-low and the depth-high band of that (H, W) band, lane = coefficient column, all requested in one burst; packed
//      accumulators = (output slice 2p, 2p + 1); results parked as V[output slice][(H, W) band][row][column];
//   2. one barrier; then wave w owns output slice w: H pass from LDS into registers ((W-low, W-high) x 2*CY rows per lane), parked
//      back into the slice's own slot as (lo, hi) pairs (wave-local ordering only);
//   3. W pass LDS -> registers -> global: lane = output column pair, one 8-byte store per row.
// LDS per workgroup = 2*CZ slots of 4 x (CY + L/2 - 1) x 64 floats (L = 4: 20 KB).  Envelope: f32, L in {2, 4, 6, 8}, unit innermost
// strides (the halo of a synthesis brick is L/2 - 1 COEFFICIENTS per axis, so eight taps still fit: 28 KB, 70 registers of inputs);
// longer filters use the composed route.  Algorithmic traffic: 8*4*B*Md*Mh*Mw read + 4*B*D*H*W written.
#include "mifwt_stream.h"

namespace mifwt {

namespace {

constexpr int kCZ3 = 2;

template <int L>
struct Idwt3TileArgs {
  const float* in[8];  // band s: bit 2 = depth high, bit 1 = row high, bit 0 = column high
  float* y;
  int64_t is_b[8];
  int is_d[2], is_h[2];  // [0]: the approximation's strides, [1]: the detail bands'
  int64_t ys_b;
  int ys_d, ys_h;
  int Md, Mh, Mw;  // coefficient extents
  int D, H, W;     // output extents (already trimmed: 2M - L + 2 - t)
  int tiles_c, tiles_r, tiles_d;
  FastDiv div_c, div_r, div_d;
  int yvec;  // 8-byte stores into y are aligned
  int nt;    // non-zero: non-temporal output stores (MIFWT_OPT_NT_STORE)
  f2 tlo[L / 2], thi[L / 2];  // (rec_lo[2j], rec_lo[2j+1]), (rec_hi[2j], rec_hi[2j+1])
};

template <int L, int CY>
__global__ void __launch_bounds__(256, 2) idwt3_tile_kernel(const Idwt3TileArgs<L> a) {
  constexpr int HL = L / 2;
  constexpr int CZ = kCZ3, NQ = 64 - (HL - 1);
  constexpr int IZ = CZ + HL - 1, IY = CY + HL - 1;
  constexpr int SLOT = 4 * IY * 64;  // floats of one output slice's V image: [(H, W) band][row][column]
  static_assert(SLOT >= 2 * CY * 64 * 2, "the (W-low, W-high) image of a slice must fit into the slot it replaces");
  static_assert(2 * CZ == 4, "one output slice per wave");
  extern __shared__ __attribute__((aligned(16))) float vimg[];  // [2 CZ][4][IY][64]

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  __builtin_assume(wave >= 0 && wave < 4);
  uint32_t utc, utr, utd;
  const int img = (int)a.div_d.divmod(a.div_r.divmod(a.div_c.divmod((uint32_t)xcd_remap(blockIdx.x, gridDim.x), utc), utr), utd);
  const int px0 = (int)utc * NQ, py0 = (int)utr * CY, pz0 = (int)utd * CZ;  // first polyphase index of the brick per axis
  constexpr uint32_t kOob = 0x80000000u;

  // ---- 1. D pass: unit = (coefficient row y, (H, W) band hw); IY units per wave ---------------------------------------------------
  {
    float v[IY][2][IZ];
    const int x = px0 + lane;
#pragma unroll
    for (int k = 0; k < IY; ++k) {
      const int u = wave * IY + k, yy = u >> 2, hw = u & 3;
      const int yc = py0 + yy;
#pragma unroll
      for (int dh = 0; dh < 2; ++dh) {
        const int s = 4 * dh + hw;
        const int sd = s == 0 ? a.is_d[0] : a.is_d[1], sh = s == 0 ? a.is_h[0] : a.is_h[1];
        const uint32_t bytes = (uint32_t)(((int64_t)(a.Md - 1) * sd + (int64_t)(a.Mh - 1) * sh + a.Mw) * 4);
        const __amdgpu_buffer_rsrc_t rs =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in[s] + (int64_t)img * a.is_b[s]), 0, bytes, 0x00020000);
        const bool row_ok = yc < a.Mh && x < a.Mw;
        const uint32_t voff = row_ok ? (uint32_t)(yc * sh + x) * 4u : kOob;
#pragma unroll
        for (int z = 0; z < IZ; ++z) {
          const int zc = pz0 + z;
          const uint32_t soff = zc < a.Md ? (uint32_t)(zc * sd) * 4u : 0u;
          v[k][dh][z] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, zc < a.Md ? voff : kOob, soff, 0));
        }
      }
    }
#pragma unroll
    for (int k = 0; k < IY; ++k) {
      const int u = wave * IY + k, yy = u >> 2, hw = u & 3;
#pragma unroll
      for (int p = 0; p < CZ; ++p) {
        f2 acc = {0.f, 0.f};  // (output slice 2p, 2p + 1)
#pragma unroll
        for (int i = 0; i < HL; ++i) acc += a.tlo[HL - 1 - i] * v[k][0][p + i] + a.thi[HL - 1 - i] * v[k][1][p + i];
        vimg[(2 * p) * SLOT + (hw * IY + yy) * 64 + lane] = acc.x;
        vimg[(2 * p + 1) * SLOT + (hw * IY + yy) * 64 + lane] = acc.y;
      }
    }
  }
  __syncthreads();

  // ---- 2. H pass of output slice `wave`: (W-low, W-high) x 2 CY rows per lane ---------------------------------------------------------
  float* slot = &vimg[wave * SLOT];
  f2 urow[2 * CY];  // (W-low, W-high) of column px0 + lane, output row r
  {
    float h[4][IY];
#pragma unroll
    for (int hw = 0; hw < 4; ++hw)
#pragma unroll
      for (int yy = 0; yy < IY; ++yy) h[hw][yy] = slot[(hw * IY + yy) * 64 + lane];
#pragma unroll
    for (int p = 0; p < CY; ++p) {
      f2 lo2 = {0.f, 0.f}, hi2 = {0.f, 0.f};  // (row 2p, row 2p + 1) of the W-low / W-high image
#pragma unroll
      for (int i = 0; i < HL; ++i) {
        lo2 += a.tlo[HL - 1 - i] * h[0][p + i] + a.thi[HL - 1 - i] * h[2][p + i];  // (H low, W low) + (H high, W low)
        hi2 += a.tlo[HL - 1 - i] * h[1][p + i] + a.thi[HL - 1 - i] * h[3][p + i];  // (H low, W high) + (H high, W high)
      }
      urow[2 * p] = (f2){lo2.x, hi2.x};
      urow[2 * p + 1] = (f2){lo2.y, hi2.y};
    }
  }
  wave_lds_fence();  // every value of the slot has been read before the (lo, hi) image overwrites it
#pragma unroll
  for (int r = 0; r < 2 * CY; ++r) *reinterpret_cast<f2*>(&slot[(r * 64 + lane) * 2]) = urow[r];
  wave_lds_fence();

  // ---- 3. W pass + stores: lane = output column pair ---------------------------------------------------------------------------------
  const int zo = 2 * pz0 + wave;
  const int xo = 2 * (px0 + lane);
  if (zo < a.D && lane < NQ && xo < a.W) {
    float* yb = a.y + (int64_t)img * a.ys_b + (int64_t)zo * a.ys_d + xo;
    const bool both = xo + 1 < a.W;
#pragma unroll
    for (int r = 0; r < 2 * CY; ++r) {
      const int yo = 2 * py0 + r;
      f2 acc = {0.f, 0.f};  // (column 2q, column 2q + 1)
#pragma unroll
      for (int i = 0; i < HL; ++i) {
        const f2 pr = *reinterpret_cast<const f2*>(&slot[(r * 64 + lane + i) * 2]);
        acc += a.tlo[HL - 1 - i] * pr.x + a.thi[HL - 1 - i] * pr.y;
      }
      if (yo < a.H) {
        float* dst = yb + (int64_t)yo * a.ys_h;
        if (a.yvec && both) {
          if (a.nt) __builtin_nontemporal_store(acc, reinterpret_cast<f2*>(dst));
          else *reinterpret_cast<f2*>(dst) = acc;
        } else {
          dst[0] = acc.x;
          if (both) dst[1] = acc.y;
        }
      }
    }
  }
}

template <int L, int kCY3>
int launch_i3(const mifwt_level_desc* d, const void* approx, const void* const* details, void* y, const double* lo, const double* hi,
              hipStream_t stream) {
  constexpr int HL = L / 2, NQ = 64 - (HL - 1), IY = kCY3 + HL - 1;
  constexpr size_t lds_bytes = (size_t)2 * kCZ3 * 4 * IY * 64 * sizeof(float);
  Idwt3TileArgs<L> a;
  a.nt = g_options[MIFWT_OPT_NT_STORE];
  for (int s = 0; s < 8; ++s) {
    a.in[s] = static_cast<const float*>(s == 0 ? approx : details[s - 1]);
    a.is_b[s] = s == 0 ? d->approx_stride[0] : d->detail_stride[0];
  }
  a.is_d[0] = (int)d->approx_stride[1];
  a.is_h[0] = (int)d->approx_stride[2];
  a.is_d[1] = (int)d->detail_stride[1];
  a.is_h[1] = (int)d->detail_stride[2];
  a.y = static_cast<float*>(y);
  a.ys_b = d->sig_stride[0];
  a.ys_d = (int)d->sig_stride[1];
  a.ys_h = (int)d->sig_stride[2];
  a.Md = (int)d->coef_extent[0];
  a.Mh = (int)d->coef_extent[1];
  a.Mw = (int)d->coef_extent[2];
  a.D = (int)d->sig_extent[0];
  a.H = (int)d->sig_extent[1];
  a.W = (int)d->sig_extent[2];
  a.tiles_c = ((a.W + 1) / 2 + NQ - 1) / NQ;
  a.tiles_r = ((a.H + 1) / 2 + kCY3 - 1) / kCY3;
  a.tiles_d = ((a.D + 1) / 2 + kCZ3 - 1) / kCZ3;
  a.div_c = make_fastdiv((uint32_t)a.tiles_c);
  a.div_r = make_fastdiv((uint32_t)a.tiles_r);
  a.div_d = make_fastdiv((uint32_t)a.tiles_d);
  a.yvec = (a.ys_h % 2 == 0 && a.ys_d % 2 == 0 && a.ys_b % 2 == 0 && reinterpret_cast<uintptr_t>(y) % 8 == 0) ? 1 : 0;
  for (int j = 0; j < HL; ++j) {
    a.tlo[j] = (f2){(float)lo[2 * j], (float)lo[2 * j + 1]};
    a.thi[j] = (f2){(float)hi[2 * j], (float)hi[2 * j + 1]};
  }
  const int64_t ntiles = (int64_t)d->batch * a.tiles_c * a.tiles_r * a.tiles_d;
  if (ntiles > INT32_MAX / 8) return MIFWT_ERR_UNSUPPORTED;
  hipLaunchKernelGGL((idwt3_tile_kernel<L, kCY3>), dim3((unsigned)ntiles), dim3(256), lds_bytes, stream, a);
  return hipGetLastError() == hipSuccess ? MIFWT_OK : MIFWT_ERR_LAUNCH;
}

}  // namespace

bool dwt3_inv_tile_supported(const mifwt_level_desc* d) {
  if (d->ndim != 3 || d->dtype != MIFWT_F32) return false;
  const int L = d->filt_len;
  if (L != 2 && L != 4 && L != 6 && L != 8) return false;
  if (d->sig_stride[3] != 1 || d->approx_stride[3] != 1 || d->detail_stride[3] != 1) return false;
  for (int i = 0; i < 3; ++i) {
    if (d->sig_stride[i] < 0 || d->approx_stride[i] < 0 || d->detail_stride[i] < 0) return false;
    if (d->coef_extent[i] < L / 2 || d->sig_extent[i] < 1 || d->sig_extent[i] > 2 * d->coef_extent[i] - L + 2) return false;
  }
  // one batch element of a band / of the output must be addressable with 32-bit byte offsets (buffer resources, int strides)
  for (const int64_t* st : {d->approx_stride, d->detail_stride}) {
    const int64_t span = (d->coef_extent[0] - 1) * st[1] + (d->coef_extent[1] - 1) * st[2] + d->coef_extent[2];
    if (span >= (int64_t(1) << 29) || st[1] >= (int64_t(1) << 29) || st[2] >= (int64_t(1) << 29)) return false;
  }
  if (d->sig_stride[1] >= (int64_t(1) << 31) || d->sig_stride[2] >= (int64_t(1) << 31)) return false;
  return true;
}

int dwt3_inv_tile(const mifwt_level_desc* d, const void* approx, const void* const* details, void* y, const double* lo,
                  const double* hi, hipStream_t stream) {
  switch (d->filt_len) {
    // (MIFWT_OPT_TILE_ROWS 8: bricks of 16 output rows instead of 8, A/B)
    case 2: return g_options[MIFWT_OPT_TILE_ROWS] == 8 ? launch_i3<2, 8>(d, approx, details, y, lo, hi, stream) : launch_i3<2, 4>(d, approx, details, y, lo, hi, stream);
    case 4: return g_options[MIFWT_OPT_TILE_ROWS] == 8 ? launch_i3<4, 8>(d, approx, details, y, lo, hi, stream) : launch_i3<4, 4>(d, approx, details, y, lo, hi, stream);
    case 6: return g_options[MIFWT_OPT_TILE_ROWS] == 8 ? launch_i3<6, 8>(d, approx, details, y, lo, hi, stream) : launch_i3<6, 4>(d, approx, details, y, lo, hi, stream);
    case 8: return g_options[MIFWT_OPT_TILE_ROWS] == 8 ? launch_i3<8, 8>(d, approx, details, y, lo, hi, stream) : launch_i3<8, 4>(d, approx, details, y, lo, hi, stream);
    default: return MIFWT_ERR_UNSUPPORTED;
  }
}

}  // namespace mifwt
