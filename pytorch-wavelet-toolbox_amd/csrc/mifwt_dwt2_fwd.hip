// mifwt_dwt2_fwd.hip — fused single-launch 2-D analysis level for gfx950 (the north-star kernel).
//
// Replaces, for one level of wavedec2 / fswavedec2:  F.pad + F.conv2d([4,1,L,L], stride 2) + split
// (reference src/ptwt/conv_transform_2.py:142-149) — separably, with the boundary extension as an index
// map, reading the input once and writing the four sub-band planes once.
//
// Design ("streaming wave strips"; the level is HBM-bound: ~8 B moved per 16 FMA):
//   * The unit of work is ONE WAVEFRONT (64 lanes), not a workgroup: a wave owns a vertical strip of
//     256 extended input columns (lane l owns 4 consecutive columns -> one 16-byte load per row, 1 KiB per
//     wave-row, fully coalesced) and walks down a chunk of rows.  Waves never synchronise with each other:
//     no s_barrier anywhere, a workgroup is just four independent waves sharing a CU.
//   * Column (vertical) pass in REGISTERS: each lane keeps a ring of the most recent input rows of its
//     4 columns; rows are requested several rows ahead of use, which is what keeps HBM busy.
//   * Row (horizontal) pass through a 4 KiB per-wave LDS slab: the vertical low/high rows of two
//     consecutive output rows are written with ds_write_b128, then lanes 0-31 / 32-63 each produce 4
//     consecutive output columns of all four bands for one of the two rows and store them with one
//     16-byte store per band.
//   * Boundary handling never touches the bulk of the image.  Output columns are split into
//       - INTERIOR strips: every tap of every output lies inside the image and inside fully valid 4-column
//         groups -> no column index map at all, plain vector loads;
//       - EDGE strips (left: outputs [0, KL), right: outputs [KE, Wo)): per-element index-mapped loads
//         (zero / constant / reflect / periodic / symmetric by ext_index), a handful of columns wide.
//     Rows: the source row of every ring index of the chunk is tabulated once per wave in LDS (boundary
//     map included), so the streaming loop carries no index arithmetic beyond one table read per row pair.
//   * Neighbouring strips overlap by R4 = roundup4(L-2) columns and neighbouring row chunks by L-2 rows
//     (re-read through L2); blockIdx is remapped so that neighbours share an XCD's L2.
//
// Algorithmic traffic per level: 4*B*H*W bytes read + 4*4*B*Ho*Wo bytes written (f32).
#include "mifwt_common.h"

namespace mifwt {

namespace {

typedef float f4 __attribute__((ext_vector_type(4)));
// 16-byte vectors that are only guaranteed 4-byte aligned (odd row pitches such as 515 floats)
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));

template <int L>
struct Dwt2FwdArgs {
  const float* x;
  float* out[4];       // bands aa, ad, da, dd
  int64_t xs_b, xs_h;  // input strides (elements); innermost stride is 1
  int64_t os_b[4], os_h[4];
  int H, W, Ho, Wo;
  int kl;              // outputs [0, kl) belong to the left edge strip
  int ke;              // outputs [ke, Wo) belong to the right edge strips; [kl, ke) is interior
  int n_int, n_edge_r; // number of interior strips / right edge strips (left edge strip: kl > 0)
  int nstrips, nchunks, ntasks;
  int rows_per_chunk;  // output rows per chunk (even)
  int mode;
  float lo[L], hi[L];  // dec_lo / dec_hi in PyWavelets order
};

constexpr int round4(int v) { return (v + 3) & ~3; }
constexpr int kMaxRowsPerChunk = 64;
constexpr int kRowTab = 2 * kMaxRowsPerChunk + 4 + 28;  // >= 4 * npairs + RING for every configuration

template <int L>
struct Cfg {
  static constexpr int R4 = round4(L - 2);                 // left overlap of a strip (columns)
  static constexpr int KS = ((256 - R4) / 2) & ~3;         // output columns per strip (multiple of 4)
  static constexpr int KL = round4((L - 2) / 2);           // outputs whose taps reach columns < 0
  static constexpr int NCH = (R4 + 8) / 4;                 // float4 chunks a lane reads per filter row
  static constexpr int RING = L <= 8 ? 16 : (L <= 12 ? 20 : 24);  // register ring depth (rows)
  static constexpr int U = RING / 4;                       // row pairs per unrolled loop body
  static_assert(RING >= L + 6, "ring must hold a row pair's window plus the rows being refilled");
  static_assert(2 * KL >= R4, "interior strips must start at a non-negative column");
};

// One wave: output columns [k_base, k_end) x output rows [j0, j1) of image `img`, all four bands.
//   EDGE : columns go through the per-element boundary index map (interior strips need none)
template <int L, bool EDGE>
__device__ __forceinline__ void strip_body(const Dwt2FwdArgs<L>& a, float (*lds)[2][256], int* rowtab, const int lane,
                                           const int img, const int k_base, const int k_end, const int j0,
                                           const int j1) {
  using C = Cfg<L>;
  constexpr int R4 = C::R4, KS = C::KS, NCH = C::NCH, RING = C::RING, U = C::U;

  const int c_first = 2 * k_base - R4 + 4 * lane;  // first extended input column of this lane
  const int npairs = (j1 - j0 + 1) >> 1;
  const int row_first = 2 * j0 - (L - 2);          // extended input row of ring index t = 0
  const int nrows_in = 2 * (j1 - j0) + L - 2;      // ring indices t in [0, nrows_in) are needed

  const float* __restrict__ xb = a.x + (int64_t)img * a.xs_b;

  // per-lane column addressing
  int coff[4];     // EDGE: mapped source column per element (clamped to 0 when it is an implicit zero)
  float cmask[4];  // EDGE: 0 for implicit zeros, else 1
  int cvec = 0;    // !EDGE: column of the 16-byte load (lanes right of the image re-load column 0; unused)
  if (EDGE) {
    // lanes right of the last column this strip needs all read column 0 (one broadcast line, unused)
    const bool needed = c_first <= 2 * (k_end - 1) + 1;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int m = needed ? ext_index(c_first + e, a.W, a.mode) : 0;
      coff[e] = m < 0 ? 0 : m;
      cmask[e] = m < 0 ? 0.f : 1.f;
    }
  } else {
    cvec = (c_first + 3 < a.W) ? c_first : 0;
  }

  // source row (or -1: all-zero row / not needed) of every ring index this chunk can touch
  for (int t = lane; t < kRowTab; t += 64) rowtab[t] = t < nrows_in ? ext_index(row_first + t, a.H, a.mode) : -1;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

  auto load_row = [&](int src) -> f4 {  // src is wave-uniform
    f4 v = {0.f, 0.f, 0.f, 0.f};
    if (src >= 0) {
      const float* __restrict__ rp = xb + (int64_t)src * a.xs_h;
      if (EDGE) {
        v.x = rp[coff[0]] * cmask[0];
        v.y = rp[coff[1]] * cmask[1];
        v.z = rp[coff[2]] * cmask[2];
        v.w = rp[coff[3]] * cmask[3];
      } else {
        v = *reinterpret_cast<const f4u*>(rp + cvec);
      }
    }
    return v;
  };
  auto load_rows4 = [&](int t0, f4& r0, f4& r1, f4& r2, f4& r3) {  // t0 is a multiple of 4
    const int4 src = *reinterpret_cast<const int4*>(&rowtab[t0]);
    r0 = load_row(__builtin_amdgcn_readfirstlane(src.x));
    r1 = load_row(__builtin_amdgcn_readfirstlane(src.y));
    r2 = load_row(__builtin_amdgcn_readfirstlane(src.z));
    r3 = load_row(__builtin_amdgcn_readfirstlane(src.w));
  };

  f4 ring[RING];
#pragma unroll
  for (int t = 0; t < RING - 4; t += 4) load_rows4(t, ring[t], ring[t + 1], ring[t + 2], ring[t + 3]);

  // horizontal-pass role of this lane
  const int hrow = lane >> 5;       // which of the two output rows of a pair
  const int q = lane & 31;          // group of 4 output columns inside the strip
  const int kcol = k_base + 4 * q;  // first output column of this lane
  const bool hactive = q < KS / 4 && kcol < k_end;
  const bool full4 = kcol + 3 < k_end;

  float* __restrict__ ob[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) ob[s] = a.out[s] + (int64_t)img * a.os_b[s] + kcol;

  for (int g = 0;; ++g) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int p = g * U + u;  // pair index inside the chunk
      if (p >= npairs) return;
      // refill the four ring slots that the previous pair released
      load_rows4(4 * p + RING - 4, ring[(4 * u + RING - 4) % RING], ring[(4 * u + RING - 3) % RING],
                 ring[(4 * u + RING - 2) % RING], ring[(4 * u + RING - 1) % RING]);

      // ---- vertical pass: two output rows, both filters, this lane's 4 columns -----------------------
#pragma unroll
      for (int rr = 0; rr < 2; ++rr) {
        f4 vlo = {0.f, 0.f, 0.f, 0.f}, vhi = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int m = 0; m < L; ++m) {
          // c[j] = sum_m h[m] * x_ext[2j + 1 - m];  ring index of row 2j+1-m is 4p + 2rr + (L-1) - m
          const f4 xv = ring[(4 * u + 2 * rr + (L - 1) - m) % RING];
          vlo += a.lo[m] * xv;
          vhi += a.hi[m] * xv;
        }
        *reinterpret_cast<f4*>(&lds[rr][0][4 * lane]) = vlo;
        *reinterpret_cast<f4*>(&lds[rr][1][4 * lane]) = vhi;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

      // ---- horizontal pass: 4 output columns x 4 bands of one row --------------------------------------
      float wl[4 * NCH], wh[4 * NCH];
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const f4 tl = *reinterpret_cast<const f4*>(&lds[hrow][0][(8 * q + 4 * c) & 255]);
        const f4 th = *reinterpret_cast<const f4*>(&lds[hrow][1][(8 * q + 4 * c) & 255]);
        wl[4 * c + 0] = tl.x; wl[4 * c + 1] = tl.y; wl[4 * c + 2] = tl.z; wl[4 * c + 3] = tl.w;
        wh[4 * c + 0] = th.x; wh[4 * c + 1] = th.y; wh[4 * c + 2] = th.z; wh[4 * c + 3] = th.w;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

      f4 o[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float s_aa = 0.f, s_ad = 0.f, s_da = 0.f, s_dd = 0.f;
#pragma unroll
        for (int m = 0; m < L; ++m) {
          const int idx = 2 * e + 1 + R4 - m;  // extended column 2k+1-m relative to this lane's chunk base
          s_aa = fmaf(a.lo[m], wl[idx], s_aa);
          s_ad = fmaf(a.hi[m], wl[idx], s_ad);
          s_da = fmaf(a.lo[m], wh[idx], s_da);
          s_dd = fmaf(a.hi[m], wh[idx], s_dd);
        }
        o[0][e] = s_aa; o[1][e] = s_ad; o[2][e] = s_da; o[3][e] = s_dd;
      }
      const int j = j0 + 2 * p + hrow;
      if (hactive && j < j1) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          float* __restrict__ op = ob[s] + (int64_t)j * a.os_h[s];
          if (!EDGE || full4) {
            *reinterpret_cast<f4u*>(op) = o[s];
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (kcol + e < k_end) op[e] = o[s][e];
          }
        }
      }
    }
  }
}

template <int L>
__global__ void __launch_bounds__(256) dwt2_fwd_stream_kernel(const Dwt2FwdArgs<L> a) {
  using C = Cfg<L>;
  __shared__ __attribute__((aligned(16))) float lds_all[4][2][2][256];  // [wave][row][lo/hi][col]
  __shared__ __attribute__((aligned(16))) int rowtab_all[4][kRowTab];     // [wave][ring index] -> source row

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // scalar: everything below is wave-uniform

  // XCD-aware block remap (block b runs on XCD b % 8): give each XCD a contiguous range of tasks so that
  // strips / chunks that share halo columns / rows meet in the same L2.
  int bid = blockIdx.x;
  {
    const int nblk = gridDim.x;
    const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int task = bid * 4 + wave;
  if (task >= a.ntasks) return;
  const int strip = task % a.nstrips;
  const int chunk = (task / a.nstrips) % a.nchunks;
  const int img = task / (a.nstrips * a.nchunks);

  const int j0 = chunk * a.rows_per_chunk;
  const int j1 = min(j0 + a.rows_per_chunk, a.Ho);
  float(*lds)[2][256] = lds_all[wave];
  int* rowtab = rowtab_all[wave];

  if (strip < a.n_int) {
    const int k_base = a.kl + strip * C::KS;
    const int k_end = min(k_base + C::KS, a.ke);
    strip_body<L, false>(a, lds, rowtab, lane, img, k_base, k_end, j0, j1);
  } else {
    const int e = strip - a.n_int;  // right edge strips first, the left edge strip (if any) last
    int k_base, k_end;
    if (e < a.n_edge_r) {
      k_base = a.ke + e * C::KS;
      k_end = min(k_base + C::KS, a.Wo);
    } else {
      k_base = 0;
      k_end = a.kl;
    }
    strip_body<L, true>(a, lds, rowtab, lane, img, k_base, k_end, j0, j1);
  }
}

template <int L>
int launch(const mifwt_level_desc* d, const void* x, void* approx, void* const* details, const double* lo,
           const double* hi, hipStream_t stream) {
  using C = Cfg<L>;
  Dwt2FwdArgs<L> a;
  a.x = static_cast<const float*>(x);
  a.out[0] = static_cast<float*>(approx);
  for (int s = 1; s < 4; ++s) a.out[s] = static_cast<float*>(details[s - 1]);
  a.xs_b = d->sig_stride[0];
  a.xs_h = d->sig_stride[1];
  for (int s = 0; s < 4; ++s) {
    a.os_b[s] = s == 0 ? d->approx_stride[0] : d->detail_stride[0];
    a.os_h[s] = s == 0 ? d->approx_stride[1] : d->detail_stride[1];
  }
  a.H = (int)d->sig_extent[0];
  a.W = (int)d->sig_extent[1];
  a.Ho = (int)d->coef_extent[0];
  a.Wo = (int)d->coef_extent[1];
  a.mode = d->mode;
  for (int m = 0; m < L; ++m) {
    a.lo[m] = (float)lo[m];
    a.hi[m] = (float)hi[m];
  }
  // column partition: [0, kl) left edge | [kl, ke) interior | [ke, Wo) right edge.
  // interior outputs k need extended columns 2k-(L-2) .. 2k+1, all inside [0, W4) with W4 = W & ~3, so
  // that every 4-column group an interior lane loads is completely inside the row.
  const int w4 = a.W & ~3;
  int kl = C::KL < a.Wo ? C::KL : a.Wo;
  int ke = kl;
  if (w4 / 2 > kl) ke = kl + ((w4 / 2 - kl) & ~3);
  if (ke > a.Wo) ke = kl + ((a.Wo - kl) & ~3);
  a.kl = kl;
  a.ke = ke;
  a.n_int = (ke - kl + C::KS - 1) / C::KS;
  a.n_edge_r = (a.Wo - ke + C::KS - 1) / C::KS;
  a.nstrips = a.n_int + a.n_edge_r + (kl > 0 ? 1 : 0);
  // rows per chunk: enough tasks to spread over 256 CUs x ~12 waves, but chunks tall enough that the
  // L-2 rows of vertical overlap stay a small fraction
  int rpc = 32;
  const int64_t per_chunk_units = (int64_t)d->batch * a.nstrips;
  while (rpc > 8 && per_chunk_units * ((a.Ho + rpc - 1) / rpc) < 256 * 12) rpc >>= 1;
  if (g_options[MIFWT_OPT_ROWS_PER_CHUNK] > 0) rpc = (g_options[MIFWT_OPT_ROWS_PER_CHUNK] + 1) & ~1;
  if (rpc > kMaxRowsPerChunk) rpc = kMaxRowsPerChunk;
  a.rows_per_chunk = rpc;
  a.nchunks = (a.Ho + rpc - 1) / rpc;
  const int64_t ntasks = per_chunk_units * a.nchunks;
  if (ntasks > INT32_MAX / 8) return MIFWT_ERR_UNSUPPORTED;
  a.ntasks = (int)ntasks;
  const unsigned nblk = (unsigned)((ntasks + 3) / 4);
  hipLaunchKernelGGL(dwt2_fwd_stream_kernel<L>, dim3(nblk), dim3(256), 0, stream, a);
  return hipGetLastError() == hipSuccess ? MIFWT_OK : MIFWT_ERR_LAUNCH;
}

}  // namespace

bool dwt2_fwd_stream_supported(const mifwt_level_desc* d) {
  if (d->ndim != 2 || d->dtype != MIFWT_F32) return false;
  const int L = d->filt_len;
  if (L < 2 || L > 16 || (L & 1)) return false;
  if (d->sig_stride[2] != 1 || d->approx_stride[2] != 1 || d->detail_stride[2] != 1) return false;
  if (d->sig_extent[0] > (1 << 28) || d->sig_extent[1] > (1 << 28)) return false;
  return true;
}

int dwt2_fwd_stream(const mifwt_level_desc* d, const void* x, void* approx, void* const* details,
                    const double* lo, const double* hi, hipStream_t stream) {
  switch (d->filt_len) {
    case 2: return launch<2>(d, x, approx, details, lo, hi, stream);
    case 4: return launch<4>(d, x, approx, details, lo, hi, stream);
    case 6: return launch<6>(d, x, approx, details, lo, hi, stream);
    case 8: return launch<8>(d, x, approx, details, lo, hi, stream);
    case 10: return launch<10>(d, x, approx, details, lo, hi, stream);
    case 12: return launch<12>(d, x, approx, details, lo, hi, stream);
    case 14: return launch<14>(d, x, approx, details, lo, hi, stream);
    case 16: return launch<16>(d, x, approx, details, lo, hi, stream);
    default: return MIFWT_ERR_UNSUPPORTED;
  }
}

// synthesis fast path: not built yet — the dispatcher falls back to the generic axis passes
bool dwt2_inv_stream_supported(const mifwt_level_desc*) { return false; }
int dwt2_inv_stream(const mifwt_level_desc*, const void*, const void* const*, void*, const double*, const double*,
                    hipStream_t) {
  return MIFWT_ERR_UNSUPPORTED;
}

}  // namespace mifwt
