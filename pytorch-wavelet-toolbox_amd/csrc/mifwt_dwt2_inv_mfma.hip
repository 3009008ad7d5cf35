// mifwt_dwt2_inv_mfma.hip — fused 2-D synthesis level for LONG filters on f16 data with the matrix cores (gfx950), id 23.
//
// Same seam as the other fused 2-D synthesis kernels: torch.stack + F.conv_transpose2d(stride 2) + the crops of one level of
// waverec2 / fswaverec2 (reference src/ptwt/conv_transform_2.py:222-249; separable form: separable_conv_transform.py:75-110).
// With 18..32 taps the vector LDS-tile kernel (id 8) is FMA-bound: level 1 of BASELINE config 5's 32-image slice took 10.1 ms
// against 1.1 ms of HBM time, the whole fswaverec2 12.7 ms against 4.0 ms for the analysis on the matrix cores.
//
// Per axis the cropped polyphase form  y[2p + r] = sum_{i < L/2} g_lo[L-2-2i+r] a[p+i] + g_hi[L-2-2i+r] d[p+i]  over a block of 16
// output pairs is a banded product
//     y[2 p0 .. 2 p0 + 32)  =  S (32 x 64) . [a[p0 .. p0 + 32); d[p0 .. p0 + 32)],     S[2q + r][32 b + q + i] = g_b[L-2-2i+r]
// (zero elsewhere: half of S is structural zeros), i.e. the GEMM shape of the analysis kernel (mifwt_dwt2_fwd_mfma.hip): M = 32,
// K = 64, N = rows / columns, v_mfma_f32_32x32x16_f16, four K-steps, taps as f16 pairs (t = t_hi + t_lo: f32-accurate filters).
//
// The mirror of the analysis WALK: a workgroup walks down seg_tiles stacked tiles (32 output rows x 128 output columns) of one
// column panel of an image.  Per tile the loader wave fetches the 16 NEW coefficient rows x 80 columns of the four bands (a "chunk":
// 40 LDS-DMA pieces of 1 KB; two chunk buffers); the four matrix waves
//   1. filter the chunk along the rows — A = 32 (band pair, coefficient row) x 64 (low band | high band columns) straight from the
//      chunk, B = S — and write the (vertical low, vertical high) images TRANSPOSED ([band][output column][coefficient row]) into
//      one half of a 32-row LDS ring;
//   2. filter the ring window (16 older + 16 new coefficient rows) along the columns with the operands swapped (D^T: a lane owns an
//      output ROW and four groups of four adjacent columns), park the block in the dead older half of the wave's own ring columns
//      and store 16 bytes per lane, four lanes a row.
// One extra chunk per unit primes the ring.  Coefficients beyond the planes' extents count as zero (the crop of the reference):
// rows past the end are requested out of range, columns past the end are zeroed in LDS by the loader (the padding of a row pitch
// may hold anything).  The intermediate image is rounded to f16 once (5e-4 norm-wise per level, like the analysis kernel).
// Envelope: f16 storage, even L in [18, 32], unit innermost strides; calls of fewer than 16 tiles stay with the vector tile kernel.
#include "mifwt_stream.h"

// cache policy of the chunk requests (experiment builds: -DMIFWT_MFMA_DMA_NT=1 non-temporal, =2 sc1)
#if MIFWT_MFMA_DMA_NT == 1
#define MIFWT_MFMA_DMA_POLICY " nt"
#elif MIFWT_MFMA_DMA_NT == 2
#define MIFWT_MFMA_DMA_POLICY " sc1"
#else
#define MIFWT_MFMA_DMA_POLICY ""
#endif

namespace mifwt {

namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f16x __attribute__((ext_vector_type(16)));

constexpr int kSR = 16;                        // coefficient rows of a chunk (= output rows of a tile / 2)
constexpr int kSC = 80;                        // coefficient columns of a chunk: 64 + L/2 - 1 <= 79
constexpr int kSOC = 128;                      // output columns of a tile
constexpr int kSOR = 2 * kSR;                  // output rows of a tile
constexpr int kSPieces = kSC / 8;              // 16-byte pieces of a chunk row (no padding: the pitch is 160 bytes)
constexpr int kSBand = kSR * kSC;              // halfs of one band of a chunk
constexpr int kSChunkBytes = 4 * kSBand * 2;   // 10240
constexpr int kSDma = 3;                       // requests per band and chunk (64 + 64 + 32 lanes)
constexpr int kVP = 40;                        // halfs between ring columns: 32 rows + 8 (16-byte aligned fragments)
constexpr int kSRingBytes = 2 * kSOC * kVP * 2;  // 20480
constexpr int kSLdsBytes = 2 * kSChunkBytes + kSRingBytes;

struct MfmaInvArgs {
  const _Float16* in[4];  // bands aa, ad, da, dd (second letter = along the rows)
  _Float16* y;
  int64_t is_b[4], is_h[4];
  int64_t ys_b, ys_h;
  int Mh, Mw, H, W;
  int tiles_c, tiles_r;
  int seg_tiles, segs, nunits;
  int L, dbg;
  float lo[32], hi[32];  // rec taps, zero-padded to 32
};

__device__ __forceinline__ void imfma_dma16(uint32_t voff, __amdgpu_buffer_rsrc_t rsrc, uint32_t lds_addr) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen" MIFWT_MFMA_DMA_POLICY " lds" ::"v"(voff), "s"(rsrc), "s"(lds_addr) : "memory");
}

__global__ void __launch_bounds__(320, 5) idwt2_mfma_walk_kernel(const MfmaInvArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char wsm[];
  _Float16* const xt = reinterpret_cast<_Float16*>(wsm);                     // two chunks: [band][row][80]
  _Float16* const hv = reinterpret_cast<_Float16*>(wsm + 2 * kSChunkBytes);  // ring: [vertical band][output column][kVP]
  float* const taps = reinterpret_cast<float*>(hv);                          // (until the first horizontal pass)

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  __builtin_assume(wave >= 0 && wave < 5);
  const int L = a.L, HL = L >> 1;

  // units (image, row segment, panel), the panel index fastest; one contiguous eighth of the sequence per XCD, staggered starts
  // (as in the analysis walk)
  const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3, nq = gridDim.x >> 3;
  const int u_begin = (int)(((int64_t)a.nunits * xcd) >> 3), u_end = (int)(((int64_t)a.nunits * (xcd + 1)) >> 3);
  const int panel = u_end - u_begin, rot = (int)(((int64_t)panel * xcd) >> 3);
  if (q >= panel) return;
  if (threadIdx.x < 64) taps[threadIdx.x] = threadIdx.x < 32 ? a.lo[threadIdx.x] : a.hi[threadIdx.x - 32];
  __syncthreads();
  struct Unit {
    int img, tc, tr0, nt;  // image, panel, first tile row, tiles
  };
  auto locate = [&](int it) -> Unit {
    int pos = it + rot;
    if (pos >= panel) pos -= panel;
    const int idx = u_begin + pos;
    const int rest = idx / a.tiles_c;
    Unit u;
    u.tc = idx - rest * a.tiles_c;
    u.img = rest / a.segs;
    u.tr0 = (rest - u.img * a.segs) * a.seg_tiles;
    u.nt = min(a.seg_tiles, a.tiles_r - u.tr0);
    return u;
  };
  int it = q;
  // (unit, chunk) after (u, gg); chunk g of a unit = coefficient rows 16 (tr0 + g) .. + 15; tile tr0 + g - 1 (output rows
  // 32 (tr0 + g - 1) .. + 31) needs chunks g - 1 and g
  auto advance = [&](Unit& u, int& gg) -> bool {
    if (gg < u.nt) {
      ++gg;
      return true;
    }
    it += nq;
    if (it >= panel) return false;
    u = locate(it);
    gg = 0;
    return true;
  };

  // =============================================================================================================================
  if (wave == 4) {
    constexpr uint32_t kOob = 0x80000000u;
    // request j of a band: lane -> piece 64 j + lane = (row, piece of the row); the lanes past the band's 160 pieces are switched off
    int prow[kSDma];
    uint32_t pcol[kSDma];
#pragma unroll
    for (int j = 0; j < kSDma; ++j) {
      const int P = 64 * j + lane;
      prow[j] = P / kSPieces;
      pcol[j] = 16u * (uint32_t)(P - prow[j] * kSPieces);
    }
    const bool last_live = lane < kSR * kSPieces - 64 * (kSDma - 1);
    const uint32_t lds0 = (uint32_t)(uintptr_t)xt;
    auto rsrc_of = [&](const Unit& u, int s) {
      const uint32_t bytes = (MIFWT_DBG(a) & 2) ? 0u : ((uint32_t)(a.Mh - 1) * (uint32_t)a.is_h[s] + (uint32_t)a.Mw) * 2u;
      return __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(a.in[s] + (int64_t)u.img * a.is_b[s]), 0, bytes, 0x00020000);
    };
    auto issue_dma = [&](const Unit& u, int g, int buf) {
      const int r0 = kSR * (u.tr0 + g), c0 = (kSOC / 2) * u.tc;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const __amdgpu_buffer_rsrc_t rs = rsrc_of(u, s);
        const uint32_t rowb = (uint32_t)a.is_h[s] * 2u;
        const uint32_t dst = lds0 + (uint32_t)(buf * kSChunkBytes + s * (kSBand * 2));
#pragma unroll
        for (int j = 0; j < kSDma; ++j) {
          const int r = r0 + prow[j];
          const uint32_t v = r >= a.Mh ? kOob : (uint32_t)r * rowb + (uint32_t)(2 * c0) + pcol[j];
          if (j < kSDma - 1 || last_live) imfma_dma16(v, rs, dst + 1024u * (uint32_t)j);
        }
      }
    };
    // columns past the planes' width (last panel): zeros, whatever the memory behind a row's end holds
    auto zero_cols = [&](const Unit& u, int buf) {
      const int c0 = (kSOC / 2) * u.tc;
      const int z0 = max(0, a.Mw - c0);
      if (z0 >= kSC) return;
      const int nz = kSC - z0;
      _Float16* xb = xt + buf * (4 * kSBand);
      for (int e = lane; e < 4 * kSR * nz; e += 64) {
        const int rb = e / nz;
        xb[rb * kSC + z0 + (e - rb * nz)] = (_Float16)0.f;
      }
    };

    // The pieces are ranges of whole dwords counted from a piece's (even) first column: the LAST sample of a plane of odd width shares
    // its dword with two bytes past the plane's end, and that dword is refused as out of range.  The chunk that holds it gets the sample
    // by a 2-byte load of its own (one lane per band, once per plane and panel).
    auto last_sample = [&](const Unit& u, int g, int buf) {
      const int r0 = kSR * (u.tr0 + g), c0 = (kSOC / 2) * u.tc;
      const int rr = a.Mh - 1 - r0, cc = a.Mw - 1 - c0;
      if (!(a.Mw & 1) || rr < 0 || rr >= kSR || cc < 0 || cc >= kSC) return;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const __amdgpu_buffer_rsrc_t rs = rsrc_of(u, s);
        const uint32_t v = __builtin_amdgcn_raw_buffer_load_b16(rs, ((uint32_t)(a.Mh - 1) * (uint32_t)a.is_h[s] + (uint32_t)(a.Mw - 1)) * 2u, 0, 0);
        if (lane == 0) xt[buf * (4 * kSBand) + s * kSBand + rr * kSC + cc] = __builtin_bit_cast(_Float16, (unsigned short)v);
      }
    };

    // chunk s of this block's sequence lives in buffer s & 1; two chunks are in flight (chunk s + 2 is requested as soon as barrier
    // B(s) has released the buffer of chunk s); requests complete in order
    Unit u0 = locate(it), u1 = u0, u2;
    int g0 = 0, g1 = 0, g2;
    bool has1 = advance(u1, g1);
    u2 = u1;
    g2 = g1;
    bool has2 = has1 && advance(u2, g2);
    issue_dma(u0, 0, 0);
    if (has1) issue_dma(u1, g1, 1);
    for (int s = 0;; ++s) {
      if (has1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * kSDma) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      zero_cols(u0, s & 1);
      last_sample(u0, g0, s & 1);
      __syncthreads();  // A(s): chunk s complete in LDS
      __syncthreads();  // B(s): the horizontal pass has read it
      if (!has1) break;
      if (has2) issue_dma(u2, g2, s & 1);
      u0 = u1;
      g0 = g1;
      u1 = u2;
      g1 = g2;
      has1 = has2;
      if (has2) has2 = advance(u2, g2);
    }
    return;
  }

  // =============================================================================================================================
  // matrix waves
  // the matrix waves run above the loader's priority: finest level of the config-5 slice 2.17 -> 1.95 ms with any raised level
  // (1, 2 or 3), alternating rounds on one box; raising the LOADER instead: 2.33 (MIFWT_OPT_DEBUG 64 = all waves at the default)
  if (!(MIFWT_DBG(a) & 64)) __builtin_amdgcn_s_setprio(1);
  const int n = lane & 31, half = lane >> 5;
  // S fragments: S[m][k], m = n = 2 q + r, k = 16 c + 8 half + e = 32 b + kk: g_b[L - 2 - 2 (kk - q) + r] for 0 <= kk - q < L/2;
  // f16 pairs (t = t_hi + t_lo)
  h8 shi[4], slo[4];
  {
    const int qq = n >> 1, r = n & 1;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int k = 16 * c + 8 * half + e;
        const int b = k >> 5, i = (k & 31) - qq;
        const float t = (i >= 0 && i < HL) ? taps[32 * b + L - 2 - 2 * i + r] : 0.f;
        const _Float16 th = (_Float16)t;
        shi[c][e] = th;
        slo[c][e] = (_Float16)(t - (float)th);
      }
    }
  }

  const int kb = wave;  // this wave's 32 output columns of the panel (both passes)
  Unit cur = locate(it);
  int g = 0, s = 0;
  for (;;) {
    Unit un = cur;
    int gn = g;
    const bool has_next = advance(un, gn);
    __syncthreads();  // A(s): chunk s is in LDS, the ring half it goes to is no longer read
    // ---- horizontal pass of the chunk: A[(vertical band, coefficient row) n][k]: K-steps 0, 1 = the band that is low along the
    // rows (aa | da), 2, 3 = the high one (ad | dd), coefficient columns 16 kb + (0 .. 31);  D[n][output column 32 kb + j]
    {
      const _Float16* xb = xt + (s & 1) * (4 * kSBand) + (n & 15) * kSC + 16 * kb + 8 * half;
      const int bvn = n >> 4;
      f16x acc;
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const h8 xf = *reinterpret_cast<const h8*>(xb + (2 * bvn + (c >> 1)) * kSBand + 16 * (c & 1));
        if (!(MIFWT_DBG(a) & 4)) {
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(xf, shi[c], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(xf, slo[c], acc, 0, 0, 0);
        }
      }
      // D lane (n, half): output column 32 kb + n; rows i = (e & 3) + 8 (e >> 2) + 4 half: vertical band i >> 4, coefficient row i & 15
#pragma unroll
      for (int gg = 0; gg < 4; ++gg)
        *reinterpret_cast<h4*>(&hv[((gg >> 1) * kSOC + 32 * kb + n) * kVP + 16 * (g & 1) + 8 * (gg & 1) + 4 * half]) =
            (h4){(_Float16)acc[4 * gg], (_Float16)acc[4 * gg + 1], (_Float16)acc[4 * gg + 2], (_Float16)acc[4 * gg + 3]};
    }
    __syncthreads();  // B(s): ring half complete, the chunk buffer released

    // ---- vertical pass of tile tr0 + g - 1, operands swapped: A[output column n][k]: K-steps 0, 1 = the vertical-low image, 2, 3 =
    // the high one, window rows 16 (c & 1) + (0 .. 15): 0 .. 15 = chunk g - 1, 16 .. 31 = chunk g;  D[column i][output row n]
    if (g >= 1) {
      const int oldh = (g - 1) & 1;
      f16x acc;
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const h8 b = *reinterpret_cast<const h8*>(&hv[((c >> 1) * kSOC + 32 * kb + n) * kVP + 16 * (((c & 1) + oldh) & 1) + 8 * half]);
        if (!(MIFWT_DBG(a) & 4)) {
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, shi[c], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, slo[c], acc, 0, 0, 0);
        }
      }
      const int row0 = kSOR * (cur.tr0 + g - 1), col0 = kSOC * cur.tc + 32 * kb;
      if (MIFWT_DBG(a) & 1) {
      } else if (col0 + 32 <= a.W) {
        // Transposed through LDS, wave-local: this wave is the only reader of ring columns 32 kb + (0 .. 31) of both images, and their
        // OLDER half is dead once the fragments above are in registers.  Output row m of the 32 x 32 block = 64 bytes = the older
        // halves (16 halfs each) of ring columns 2 m and 2 m + 1 of the wave's 64 (image, column) slots.
        auto slot = [&](int sl) -> _Float16* { return &hv[((sl >> 5) * kSOC + 32 * kb + (sl & 31)) * kVP + 16 * oldh]; };
#pragma unroll
        for (int gg = 0; gg < 4; ++gg)
          *reinterpret_cast<h4*>(slot(2 * n + (gg >> 1)) + 8 * (gg & 1) + 4 * half) =
              (h4){(_Float16)acc[4 * gg], (_Float16)acc[4 * gg + 1], (_Float16)acc[4 * gg + 2], (_Float16)acc[4 * gg + 3]};
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int m = 16 * i + (lane >> 2), pc = lane & 3;
          const h8 v = *reinterpret_cast<const h8*>(slot(2 * m + (pc >> 1)) + 8 * (pc & 1));
          if (row0 + m < a.H) {
            _Float16* dst = a.y + (int64_t)cur.img * a.ys_b + (int64_t)(row0 + m) * a.ys_h + col0 + 8 * pc;
            // (16-byte stores; rows of an odd pitch start 2-byte aligned: works, slowly — tools/align_probe.hip)
            // non-temporal: nothing reads these lines back (config-5 slice, level 1: 2.29 -> 2.15 ms analysis, 2.21 -> 2.18 synthesis;
            // MIFWT_OPT_DEBUG 8 / 16 = write-through / default policy, tools/mfma_policy_ab.py)
            if (MIFWT_DBG(a) & 8) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(dst), "v"(v) : "memory");
            else if (MIFWT_DBG(a) & 16) asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(dst), "v"(v) : "memory");
            else asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(dst), "v"(v) : "memory");
          }
        }
      } else if (col0 < a.W) {  // the last columns of a plane: sample by sample
        const int row = row0 + n;
        if (row < a.H) {
          _Float16* base = a.y + (int64_t)cur.img * a.ys_b + (int64_t)row * a.ys_h + col0 + 4 * half;
#pragma unroll
          for (int gg = 0; gg < 4; ++gg)
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (col0 + 4 * half + 8 * gg + e < a.W) base[8 * gg + e] = (_Float16)acc[4 * gg + e];
        }
      }
    }
    if (!has_next) break;
    cur = un;
    g = gn;
    ++s;
  }
}

}  // namespace

bool dwt2_inv_mfma_supported(const mifwt_level_desc* d) {
  if (d->ndim != 2 || d->dtype != MIFWT_F16 || g_options[MIFWT_OPT_MFMA_MODE] == 2) return false;
  const int L = d->filt_len;
  if (L < 18 || L > 32 || (L & 1)) return false;
  if (d->sig_stride[2] != 1 || d->approx_stride[2] != 1 || d->detail_stride[2] != 1) return false;
  for (int i = 0; i < 2; ++i)
    if (d->approx_stride[i] < 0 || d->detail_stride[i] < 0 || d->sig_stride[i] < 0) return false;
  const int64_t lim = int64_t(1) << 29;  // 32-bit byte offsets inside a plane
  if ((d->coef_extent[0] - 1) * d->approx_stride[1] + d->coef_extent[1] >= lim) return false;
  if ((d->coef_extent[0] - 1) * d->detail_stride[1] + d->coef_extent[1] >= lim) return false;
  for (int ax = 0; ax < 2; ++ax)
    if (d->sig_extent[ax] < 1 || d->sig_extent[ax] > 2 * d->coef_extent[ax] - L + 2) return false;
  // where it pays: wherever there is a tile per workgroup or so (32 x 542^2 sym16: 0.026 against 0.205 ms for the vector tile kernel,
  // 32 x 1052^2: 0.049 against 0.172, tools/c5_rec_time.py; MIFWT_OPT_MFMA_MODE 4: always)
  // The threshold looks at ONE image's tiles, never at the batch: this kernel rounds the intermediate image to f16, the vector tile
  // kernel keeps it in f32 — a batch and its images one by one must take the same path to agree to the bit.
  const int64_t tiles_per_image = ((d->sig_extent[0] + kSOR - 1) / kSOR) * ((d->sig_extent[1] + kSOC - 1) / kSOC);
  return g_options[MIFWT_OPT_MFMA_MODE] == 4 || tiles_per_image >= 4;
}

int dwt2_inv_mfma(const mifwt_level_desc* d, const void* approx, const void* const* details, void* y, const double* lo,
                  const double* hi, hipStream_t stream) {
  MfmaInvArgs a;
  a.in[0] = static_cast<const _Float16*>(approx);
  for (int s = 1; s < 4; ++s) a.in[s] = static_cast<const _Float16*>(details[s - 1]);
  for (int s = 0; s < 4; ++s) {
    a.is_b[s] = s == 0 ? d->approx_stride[0] : d->detail_stride[0];
    a.is_h[s] = s == 0 ? d->approx_stride[1] : d->detail_stride[1];
  }
  a.y = static_cast<_Float16*>(y);
  a.ys_b = d->sig_stride[0];
  a.ys_h = d->sig_stride[1];
  a.Mh = (int)d->coef_extent[0];
  a.Mw = (int)d->coef_extent[1];
  a.H = (int)d->sig_extent[0];
  a.W = (int)d->sig_extent[1];
  a.L = d->filt_len;
  a.dbg = g_options[MIFWT_OPT_DEBUG];
  for (int m = 0; m < 32; ++m) {
    a.lo[m] = m < d->filt_len ? (float)lo[m] : 0.f;
    a.hi[m] = m < d->filt_len ? (float)hi[m] : 0.f;
  }
  a.tiles_c = (a.W + kSOC - 1) / kSOC;
  a.tiles_r = (a.H + kSOR - 1) / kSOR;
  int64_t grid = 256 * 4;  // four workgroups per CU (LDS)
  const int64_t panels = (int64_t)d->batch * a.tiles_c;
  int64_t segs = (16 * grid + panels - 1) / panels;  // about 16 units per workgroup, at least 4 tiles each
  segs = std::max<int64_t>(1, std::min<int64_t>(segs, (a.tiles_r + 3) / 4));
  a.seg_tiles = (int)((a.tiles_r + segs - 1) / segs);
  if (g_options[MIFWT_OPT_TILE_ROWS] > 0) a.seg_tiles = std::max(1, std::min(a.tiles_r, g_options[MIFWT_OPT_TILE_ROWS]));  // (experiments)
  a.segs = (a.tiles_r + a.seg_tiles - 1) / a.seg_tiles;
  const int64_t nunits = panels * a.segs;
  if (nunits > INT32_MAX / 8) return MIFWT_ERR_UNSUPPORTED;
  a.nunits = (int)nunits;
  if (nunits < grid) grid = (nunits + 7) & ~int64_t(7);
  hipLaunchKernelGGL(idwt2_mfma_walk_kernel, dim3((unsigned)grid), dim3(320), kSLdsBytes, stream, a);
  return hipGetLastError() == hipSuccess ? MIFWT_OK : MIFWT_ERR_LAUNCH;
}

}  // namespace mifwt
