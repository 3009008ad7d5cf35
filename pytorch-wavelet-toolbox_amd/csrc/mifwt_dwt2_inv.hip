// mifwt_dwt2_inv.hip — fused single-launch 2-D synthesis level for gfx950.
//
// Replaces, for one level of waverec2 / fswaverec2:  torch.stack + F.conv_transpose2d([4,1,L,L], stride 2) +
// the four crops (reference src/ptwt/conv_transform_2.py:222-249) — separably, in polyphase (gather) form:
// only the L/2 non-zero products per output sample are formed and only the cropped interior
// [L-2, L-2 + out_extent) of the transposed convolution is computed.  The four sub-band planes are read
// once, the reconstructed plane is written once.
//
// Per axis, for output index n = 2p + r, r in {0,1}:
//     y[2p + r] = sum_{i=0}^{L/2-1}  g_lo[L-2-2i+r] * a[p+i]  +  g_hi[L-2-2i+r] * d[p+i]
// i.e. the output PAIR (y[2p], y[2p+1]) is the tap PAIRS (g[2j], g[2j+1]), j = L/2-1-i, times one broadcast
// coefficient — exactly the packed v_pk_fma_f32 form of the analysis kernel (mifwt_stream.h).
//
// Design (mirror of mifwt_dwt2_fwd.hip, "streaming wave strips"):
//   * one wavefront = one strip of KQ coefficient columns (= 2*KQ output columns) x one chunk of output rows;
//     no workgroup barriers.
//   * lane l owns coefficient columns q0+2l, q0+2l+1 of all four bands: one 8-byte buffer load per band per
//     coefficient row; a register ring holds the L/2+1 rows of the current window plus the prefetched rows.
//   * vertical synthesis in registers: X_lo(rows 2p,2p+1) from (aa, da), X_hi from (ad, dd) — accumulator pair
//     = the two output rows, taps = SGPR pairs, coefficient = broadcast VGPR.
//   * horizontal synthesis through a per-wave LDS slab ([column](row0,row1) pairs): each lane reads the
//     L/2+1 neighbouring columns and produces output columns 4l..4l+3 of both rows = two 16-byte stores.
//   * one loop step consumes 2 coefficient rows and emits 4 output rows.
//
// Algorithmic traffic per level: 4*4*B*Mh*Mw bytes read + 4*B*H*W bytes written (f32).
#include "mifwt_stream.h"

namespace mifwt {

namespace {

template <int L>
struct Dwt2InvArgs {
  const float* in[4];  // bands aa, ad, da, dd
  float* y;
  int64_t is_b[4], is_h[4];  // band strides (elements); innermost stride 1
  int64_t ys_b, ys_h;
  int Mh, Mw;          // coefficient extents
  int H, W;            // output extents (already trimmed: 2M - L + 2 - t)
  int nstrips, nchunks, ntasks;
  int rows_per_chunk;  // output rows per chunk (multiple of 4)
  f2 tlo[L / 2];       // (rec_lo[2j], rec_lo[2j+1])
  f2 thi[L / 2];       // (rec_hi[2j], rec_hi[2j+1])
};

template <int L, int D>
struct ICfg {
  static constexpr int HL = L / 2;
  static constexpr int KQ = (L == 2) ? 128 : 112;            // coefficient columns per strip whose outputs are stored
  static constexpr int RING = ((HL + 1 + 2 * D) + 1) & ~1;   // coefficient rows held per lane (even)
  static constexpr int U = RING / 2;                         // loop steps per unrolled body
  static constexpr int NRD = (2 * (HL + 1) + 3) / 4;         // 16-byte LDS reads per array per row pair
  static_assert(RING >= HL + 3, "ring must hold a step's window plus the rows being refilled");
  static_assert(2 * (KQ / 2 - 1) + 1 + HL <= 128 + 3, "strip window exceeds the LDS row");
};

constexpr int kInvLdsRow = (128 + 8) * 2;  // floats: 128 columns (+ slack) x (row0,row1)

template <int L, int D>
__global__ void __launch_bounds__(256, 3) dwt2_inv_stream_kernel(const Dwt2InvArgs<L> a) {
  using C = ICfg<L, D>;
  constexpr int HL = C::HL, KQ = C::KQ, RING = C::RING, U = C::U, NRD = C::NRD;
  __shared__ __attribute__((aligned(16))) float lds_all[4][2][2][kInvLdsRow];  // [wave][p parity][lo/hi][col x 2]

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int task = xcd_remap(blockIdx.x, gridDim.x) * 4 + wave;
  if (task >= a.ntasks) return;
  const int strip = task % a.nstrips;
  const int chunk = (task / a.nstrips) % a.nchunks;
  const int img = task / (a.nstrips * a.nchunks);
  float(*lds)[2][kInvLdsRow] = lds_all[wave];

  const int y0 = chunk * a.rows_per_chunk;                 // first output row of the chunk (multiple of 4)
  const int y1 = min(y0 + a.rows_per_chunk, a.H);
  const int nsteps = (y1 - y0 + 3) >> 2;                   // 4 output rows per step
  const int m0 = y0 >> 1;                                  // coefficient row of ring index t = 0
  const int q0 = strip * KQ;                               // first coefficient column of the strip
  const int x0 = 2 * q0;                                   // first output column of the strip

  // band images as buffer resources (out-of-range reads return 0, so a float2 that straddles the end of the
  // last row is harmless); per-lane byte offset of this lane's two coefficient columns
  __amdgpu_buffer_rsrc_t rs[4];
  uint32_t row_bytes[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const uint32_t bytes = ((uint32_t)(a.Mh - 1) * (uint32_t)a.is_h[s] + (uint32_t)a.Mw) * 4u;
    rs[s] = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in[s] + (int64_t)img * a.is_b[s]), 0, bytes, 0x00020000);
    row_bytes[s] = (uint32_t)a.is_h[s] * 4u;
  }
  const int qc = q0 + 2 * lane;
  const uint32_t coff = (qc < a.Mw) ? 4u * (uint32_t)qc : 0u;  // lanes right of the band re-read column 0 (unused)

  struct Row {
    f2 b[4];  // two coefficient columns of the four bands
  };
  auto load_row = [&](int t) -> Row {
    int m = m0 + t;
    m = m < a.Mh ? m : a.Mh - 1;  // rows past the band are never used; keep the load in range
    Row r;
#pragma unroll
    for (int s = 0; s < 4; ++s)
      r.b[s] = __builtin_bit_cast(f2, __builtin_amdgcn_raw_buffer_load_b64(rs[s], coff, (uint32_t)m * row_bytes[s], 0));
    return r;
  };

  Row ring[RING];
#pragma unroll
  for (int t = 0; t < RING - 2; ++t) ring[t] = load_row(t);

  const bool sactive = lane < KQ / 2 && x0 + 4 * lane < a.W;
  const bool full4 = x0 + 4 * lane + 3 < a.W;
  char* const ybase = reinterpret_cast<char*>(a.y + (int64_t)img * a.ys_b + (int64_t)y0 * a.ys_h);
  const int64_t yrow_bytes = a.ys_h * 4;
  const uint32_t yoff = 4u * (uint32_t)(x0 + 4 * lane);

  for (int g = 0;; ++g) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int st = g * U + u;  // step: output rows y0 + 4 st .. + 3, coefficient rows m0 + 2 st .. + HL
      if (st >= nsteps) return;
      ring[(2 * u + RING - 2) % RING] = load_row(2 * st + RING - 2);
      ring[(2 * u + RING - 1) % RING] = load_row(2 * st + RING - 1);

      // ---- vertical synthesis: output row pairs pp = 0, 1 -> X_lo / X_hi for this lane's two columns ------
#pragma unroll
      for (int pp = 0; pp < 2; ++pp) {
        f2 xl[2], xh[2];  // (row 2p, row 2p+1) of X_lo / X_hi at columns qc, qc+1
#pragma unroll
        for (int i = 0; i < HL; ++i) {
          const Row& r = ring[(2 * u + pp + i) % RING];
          const f2 tl = a.tlo[HL - 1 - i], th = a.thi[HL - 1 - i];
          if (i == 0) {
            xl[0] = pkmul_lo(tl, r.b[0]);  // aa: low vertical, low horizontal
            xl[1] = pkmul_hi(tl, r.b[0]);
            xh[0] = pkmul_lo(tl, r.b[1]);  // ad: low vertical, high horizontal
            xh[1] = pkmul_hi(tl, r.b[1]);
          } else {
            pkfma_lo(xl[0], tl, r.b[0]);
            pkfma_hi(xl[1], tl, r.b[0]);
            pkfma_lo(xh[0], tl, r.b[1]);
            pkfma_hi(xh[1], tl, r.b[1]);
          }
          pkfma_lo(xl[0], th, r.b[2]);  // da: high vertical, low horizontal
          pkfma_hi(xl[1], th, r.b[2]);
          pkfma_lo(xh[0], th, r.b[3]);  // dd
          pkfma_hi(xh[1], th, r.b[3]);
        }
        *reinterpret_cast<f4*>(&lds[pp][0][4 * lane]) = (f4){xl[0].x, xl[0].y, xl[1].x, xl[1].y};
        *reinterpret_cast<f4*>(&lds[pp][1][4 * lane]) = (f4){xh[0].x, xh[0].y, xh[1].x, xh[1].y};
      }
      wave_lds_fence();

      // ---- horizontal synthesis: output columns 4l .. 4l+3 of the four rows ------------------------------------
#pragma unroll
      for (int pp = 0; pp < 2; ++pp) {
        f2 wl[2 * NRD], wh[2 * NRD];  // column pairs (row0,row1) from column 2l on
#pragma unroll
        for (int c = 0; c < NRD; ++c) {
          const f4 tl = *reinterpret_cast<const f4*>(&lds[pp][0][4 * lane + 4 * c]);
          const f4 th = *reinterpret_cast<const f4*>(&lds[pp][1][4 * lane + 4 * c]);
          wl[2 * c] = (f2){tl.x, tl.y};
          wl[2 * c + 1] = (f2){tl.z, tl.w};
          wh[2 * c] = (f2){th.x, th.y};
          wh[2 * c + 1] = (f2){th.z, th.w};
        }
        f2 o0[2], o1[2];  // o0[e] = outputs (4l+2e, 4l+2e+1) of row 2p, o1[e] of row 2p+1
#pragma unroll
        for (int e = 0; e < 2; ++e) {
#pragma unroll
          for (int i = 0; i < HL; ++i) {
            const f2 tl = a.tlo[HL - 1 - i], th = a.thi[HL - 1 - i];
            if (i == 0) {
              o0[e] = pkmul_lo(tl, wl[e + i]);
              o1[e] = pkmul_hi(tl, wl[e + i]);
            } else {
              pkfma_lo(o0[e], tl, wl[e + i]);
              pkfma_hi(o1[e], tl, wl[e + i]);
            }
            pkfma_lo(o0[e], th, wh[e + i]);
            pkfma_hi(o1[e], th, wh[e + i]);
          }
        }
        const int yr = y0 + 4 * st + 2 * pp;
        if (sactive) {
          char* const rp = ybase + (int64_t)(4 * st + 2 * pp) * yrow_bytes;
          const f4 v0 = {o0[0].x, o0[0].y, o0[1].x, o0[1].y};
          const f4 v1 = {o1[0].x, o1[0].y, o1[1].x, o1[1].y};
          if (full4) {
            if (yr < y1) *reinterpret_cast<f4u*>(rp + yoff) = v0;
            if (yr + 1 < y1) *reinterpret_cast<f4u*>(rp + yrow_bytes + yoff) = v1;
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              if (x0 + 4 * lane + e < a.W) {
                if (yr < y1) *reinterpret_cast<float*>(rp + yoff + 4 * e) = v0[e];
                if (yr + 1 < y1) *reinterpret_cast<float*>(rp + yrow_bytes + yoff + 4 * e) = v1[e];
              }
            }
          }
        }
      }
      wave_lds_fence();
    }
  }
}

template <int L, int D>
int launch(const mifwt_level_desc* d, const void* approx, const void* const* details, void* y, const double* lo,
           const double* hi, hipStream_t stream) {
  using C = ICfg<L, D>;
  Dwt2InvArgs<L> a;
  a.in[0] = static_cast<const float*>(approx);
  for (int s = 1; s < 4; ++s) a.in[s] = static_cast<const float*>(details[s - 1]);
  for (int s = 0; s < 4; ++s) {
    a.is_b[s] = s == 0 ? d->approx_stride[0] : d->detail_stride[0];
    a.is_h[s] = s == 0 ? d->approx_stride[1] : d->detail_stride[1];
  }
  a.y = static_cast<float*>(y);
  a.ys_b = d->sig_stride[0];
  a.ys_h = d->sig_stride[1];
  a.Mh = (int)d->coef_extent[0];
  a.Mw = (int)d->coef_extent[1];
  a.H = (int)d->sig_extent[0];
  a.W = (int)d->sig_extent[1];
  for (int j = 0; j < L / 2; ++j) {
    a.tlo[j] = (f2){(float)lo[2 * j], (float)lo[2 * j + 1]};
    a.thi[j] = (f2){(float)hi[2 * j], (float)hi[2 * j + 1]};
  }
  a.nstrips = (a.W + 2 * C::KQ - 1) / (2 * C::KQ);
  // output rows per task: 16; 32 for long filters on big planes — a task re-reads L/2 - 1 coefficient rows of every band as its halo
  // (config 4's finest level, 16 taps on 64 x 4096^2: 2.57 ms per waverec2 with 16, 2.45 with 32, 2.53 with 64; profiles/r04j_c4_rpc_sweep.txt)
  int rpc = (L >= 12 && a.H >= 2048) ? 32 : 16;
  if (g_options[MIFWT_OPT_ROWS_PER_CHUNK] > 0) rpc = (g_options[MIFWT_OPT_ROWS_PER_CHUNK] + 3) & ~3;
  a.rows_per_chunk = rpc;
  a.nchunks = (a.H + rpc - 1) / rpc;
  const int64_t ntasks = (int64_t)d->batch * a.nstrips * a.nchunks;
  if (ntasks > INT32_MAX / 8) return MIFWT_ERR_UNSUPPORTED;
  a.ntasks = (int)ntasks;
  hipLaunchKernelGGL((dwt2_inv_stream_kernel<L, D>), dim3((unsigned)((ntasks + 3) / 4)), dim3(256), 0, stream, a);
  return hipGetLastError() == hipSuccess ? MIFWT_OK : MIFWT_ERR_LAUNCH;
}

}  // namespace

bool dwt2_inv_stream_supported(const mifwt_level_desc* d) {
  if (d->ndim != 2 || d->dtype != MIFWT_F32) return false;
  const int L = d->filt_len;
  if (L < 2 || L > 16 || (L & 1)) return false;
  if (d->sig_stride[2] != 1 || d->approx_stride[2] != 1 || d->detail_stride[2] != 1) return false;
  for (int i = 0; i < 2; ++i)
    if (d->approx_stride[i] < 0 || d->detail_stride[i] < 0 || d->sig_stride[i] < 0) return false;
  // one band image must be addressable with 32-bit byte offsets (buffer-resource loads)
  const int64_t span_a = (d->coef_extent[0] - 1) * d->approx_stride[1] + d->coef_extent[1];
  const int64_t span_d = (d->coef_extent[0] - 1) * d->detail_stride[1] + d->coef_extent[1];
  if (span_a >= (int64_t(1) << 29) || span_d >= (int64_t(1) << 29)) return false;
  if (d->sig_extent[1] >= (int64_t(1) << 29)) return false;
  return true;
}

int dwt2_inv_stream(const mifwt_level_desc* d, const void* approx, const void* const* details, void* y,
                    const double* lo, const double* hi, hipStream_t stream) {
  switch (d->filt_len) {
    case 2: return launch<2, 1>(d, approx, details, y, lo, hi, stream);
    case 4: return launch<4, 1>(d, approx, details, y, lo, hi, stream);
    case 6: return launch<6, 1>(d, approx, details, y, lo, hi, stream);
    case 8: return launch<8, 1>(d, approx, details, y, lo, hi, stream);
    case 10: return launch<10, 1>(d, approx, details, y, lo, hi, stream);
    case 12: return launch<12, 1>(d, approx, details, y, lo, hi, stream);
    case 14: return launch<14, 1>(d, approx, details, y, lo, hi, stream);
    case 16: return launch<16, 1>(d, approx, details, y, lo, hi, stream);  // (prefetch depth 2 / 3: 1.71 / 1.67-1.76 against 1.66 ms on config 4, EXPERIMENTS R5.10)
    default: return MIFWT_ERR_UNSUPPORTED;
  }
}

}  // namespace mifwt
