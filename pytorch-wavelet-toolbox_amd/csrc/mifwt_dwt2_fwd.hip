// mifwt_dwt2_fwd.hip — fused single-launch 2-D analysis level for gfx950, streaming wave strips (kernel id 1).
//
// The first design of the north-star kernel.  The LDS-tile kernel (mifwt_dwt2_tile.h, kernel id 7) has since overtaken
// it for every filter up to 14 taps and for 16 taps below ~1500^2 planes; this one stays the choice for 16-tap filters
// on big planes, where its register ring re-reads no row halo (dispatch: dwt2_fwd_choice in mifwt_compose.hip).
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
//   * Row (horizontal) pass through a 4.25 KiB per-wave LDS slab: the vertical (low, high) results of two
//     consecutive output rows are written with ds_write_b128 (bank-conflict-free padded layout), then
//     lanes 0-31 / 32-63 each produce 4 consecutive output columns of all four bands for one of the two
//     rows and store them with one 16-byte store per band.
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
#include "mifwt_stream.h"

namespace mifwt {

namespace {


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
  int nt_store;        // non-zero: streaming (nontemporal) stores for the sub-band planes
  f2 tap[L];           // (dec_lo[m], dec_hi[m]) in PyWavelets order: one SGPR pair per tap
};

constexpr int round4(int v) { return (v + 3) & ~3; }
constexpr int kMaxRowsPerChunk = 64;
constexpr int kRowTab = 2 * kMaxRowsPerChunk + 4 + 28;  // >= 4 * npairs + RING for every configuration

template <int L, int D = 1>
struct Cfg {
  static constexpr int R4 = round4(L - 2);                 // left overlap of a strip (columns)
  static constexpr int KS = ((256 - R4) / 2) & ~3;         // output columns per strip (multiple of 4)
  static constexpr int KL = round4((L - 2) / 2);           // outputs whose taps reach columns < 0
  static constexpr int NCH = (R4 + 8) / 4;                 // float4 chunks a lane reads per filter row
  static constexpr int RING = round4(L + 2 + 4 * D);        // register ring depth (rows): window + D pairs ahead
  static constexpr int U = RING / 4;                       // row pairs per unrolled loop body
  static_assert(RING >= L + 6, "ring must hold a row pair's window plus the rows being refilled");
  static_assert(4 * (kMaxRowsPerChunk / 2) + RING <= kRowTab, "row table too small");
  static_assert(2 * KL >= R4, "interior strips must start at a non-negative column");
};

// One wave: output columns [k_base, k_end) x output rows [j0, j1) of image `img`, all four bands.
//   EDGE : columns go through the per-element boundary index map (interior strips need none)
//   D    : prefetch depth in row pairs (ring = window + 4*D rows)
// Arithmetic layout: both passes issue v_pk_fma_f32 with the FILTER PAIR packed — accumulator pair
// (low-pass, high-pass) += (lo[m], hi[m]) * broadcast(sample) — so a tap is one SGPR pair, the sample is a
// single (op_sel-broadcast) VGPR with no pairing/alignment constraint, and the vertical results leave the
// registers already interleaved (lo, hi) per column, which is the LDS layout the horizontal pass reads.
// The streaming loop is branch-free around its loads (rows come from a per-wave table of clamped source
// rows; implicit-zero rows/columns are handled by multiplying with 0/1 masks at first use), so that the
// compiler keeps counted s_waitcnt vmcnt(N) and the prefetched rows stay in flight across iterations.
// All global addresses are (wave-uniform SGPR base) + (per-lane 32-bit byte offset).

// LDS row image: 256 columns x (lo, hi) = 128 16-byte slots, one pad slot after every 4 (physical slot =
// s + (s >> 2)): ds_write_b128 (lane l -> slots 2l, 2l+1) and ds_read_b128 (lane q -> slots 4q + c) are
// bank-conflict-free and the read offsets are compile-time constants relative to one per-lane base.
constexpr int kLdsRowFloats = 168 * 4;

template <int L, int D, bool EDGE>
__device__ __forceinline__ void strip_body(const Dwt2FwdArgs<L>& a, float (*lds)[kLdsRowFloats], int* rowtab,
                                           const int lane, const int img, const int k_base, const int k_end,
                                           const int j0, const int j1) {
  using C = Cfg<L, D>;
  constexpr int R4 = C::R4, KS = C::KS, NCH = C::NCH, RING = C::RING, U = C::U;

  const int c_first = 2 * k_base - R4 + 4 * lane;  // first extended input column of this lane
  const int npairs = (j1 - j0 + 1) >> 1;
  const int row_first = 2 * j0 - (L - 2);          // extended input row of ring index t = 0
  const int nrows_in = 2 * (j1 - j0) + L - 2;      // ring indices t in [0, nrows_in) are needed

  // input image as a buffer resource: loads are (SGPR descriptor) + (per-lane byte offset) + (uniform row
  // offset), no 64-bit address arithmetic in the loop
  const uint32_t row_bytes = (uint32_t)a.xs_h * 4u;
  const uint32_t img_bytes = ((uint32_t)(a.H - 1) * (uint32_t)a.xs_h + (uint32_t)a.W) * 4u;
  const __amdgpu_buffer_rsrc_t xrsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x + (int64_t)img * a.xs_b), 0, img_bytes, 0x00020000);

  // per-lane column addressing (byte offsets inside a row)
  uint32_t coff[4];  // EDGE: mapped source column per element (0 when it is an implicit zero or not needed)
  float cmask[4];    // EDGE: 0 for implicit zeros, else 1
  uint32_t cvec = 0; // !EDGE: offset of the 16-byte load (lanes right of the image re-load column 0; unused)
  if (EDGE) {
    // lanes right of the last column this strip needs all read column 0 (one broadcast line, unused)
    const bool needed = c_first <= 2 * (k_end - 1) + 1;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int m = needed ? ext_index(c_first + e, a.W, a.mode) : 0;
      coff[e] = m < 0 ? 0u : 4u * (uint32_t)m;
      cmask[e] = m < 0 ? 0.f : 1.f;
    }
  } else {
    cvec = (c_first + 3 < a.W) ? 4u * (uint32_t)c_first : 0u;
  }

  // Per-wave row table: byte offset of the clamped source row of every ring index this chunk can touch
  // (rows past the chunk re-read its last row: an L1/L2 hit, never used) and, for zero mode, whether the
  // row is an implicit zero.
  const bool rows_inside = row_first >= 0 && row_first + nrows_in <= a.H;
  const bool zero_rows = a.mode == MIFWT_MODE_ZERO && !rows_inside;
  float* rowmask = reinterpret_cast<float*>(rowtab + kRowTab);
  if (rows_inside) {
    for (int t = lane; t < kRowTab; t += 64) rowtab[t] = (row_first + (t < nrows_in ? t : nrows_in - 1)) * (int)row_bytes;
  } else {
    for (int t = lane; t < kRowTab; t += 64) {
      const int m = ext_index(row_first + (t < nrows_in ? t : nrows_in - 1), a.H, a.mode);
      rowtab[t] = (m < 0 ? 0 : m) * (int)row_bytes;
      rowmask[t] = m < 0 ? 0.f : 1.f;
    }
  }
  wave_lds_fence();

  auto load_row = [&](int soff) -> f4 {  // soff: wave-uniform byte offset of a valid row
    f4 v;
    if (EDGE) {
      v.x = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrsrc, coff[0], soff, 0));
      v.y = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrsrc, coff[1], soff, 0));
      v.z = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrsrc, coff[2], soff, 0));
      v.w = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrsrc, coff[3], soff, 0));
    } else {
      v = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, cvec, soff, 0));
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
  auto mask_rows4 = [&](int t0, f4& r0, f4& r1, f4& r2, f4& r3) {  // zero mode, image top / bottom only
    const f4 mk = *reinterpret_cast<const f4*>(&rowmask[t0]);
    r0 *= mk.x;
    r1 *= mk.y;
    r2 *= mk.z;
    r3 *= mk.w;
  };

  f4 ring[RING];
#pragma unroll
  for (int t = 0; t < RING - 4; t += 4) load_rows4(t, ring[t], ring[t + 1], ring[t + 2], ring[t + 3]);
  if (zero_rows) {
#pragma unroll
    for (int t = 0; t < RING - 4; t += 4) mask_rows4(t, ring[t], ring[t + 1], ring[t + 2], ring[t + 3]);
  }

  // horizontal-pass role of this lane
  const int hrow = lane >> 5;       // which of the two output rows of a pair
  const int q = lane & 31;          // group of 4 output columns inside the strip
  const int kcol = k_base + 4 * q;  // first output column of this lane
  const bool hactive = q < KS / 4 && kcol < k_end;
  const bool full4 = kcol + 3 < k_end;

  // output addressing: uniform base (band, image, row pair) + per-lane byte offset (row of the pair, column)
  char* __restrict__ obase[4];
  int64_t opair_bytes[4];
  uint32_t ooff[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    obase[s] = reinterpret_cast<char*>(a.out[s] + (int64_t)img * a.os_b[s] + (int64_t)j0 * a.os_h[s]);
    opair_bytes[s] = a.os_h[s] * 8;
    ooff[s] = 4u * ((uint32_t)hrow * (uint32_t)a.os_h[s] + (uint32_t)kcol);
  }

  // LDS addressing (float indices, padded layout): this lane writes its 4 columns (logical slots 2l, 2l+1
  // -> 32 contiguous bytes) and reads the 2*NCH logical slots from 4q on
  float* const wr = &lds[0][(2 * lane + (lane >> 1)) * 4];
  const float* const rdp = &lds[hrow][5 * q * 4];

  for (int g = 0;; ++g) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int p = g * U + u;  // pair index inside the chunk
      if (p >= npairs) return;
      // refill the four ring slots that the previous pair released (rows 4p+RING-4 .. 4p+RING-1)
      load_rows4(4 * p + RING - 4, ring[(4 * u + RING - 4) % RING], ring[(4 * u + RING - 3) % RING],
                 ring[(4 * u + RING - 2) % RING], ring[(4 * u + RING - 1) % RING]);
      if (zero_rows && p > 0) {
        // rows that were requested one iteration ago are masked just before their first use
        mask_rows4(4 * p + RING - 8, ring[(4 * u + RING - 8) % RING], ring[(4 * u + RING - 7) % RING],
                   ring[(4 * u + RING - 6) % RING], ring[(4 * u + RING - 5) % RING]);
      }

      // ---- vertical pass: two output rows, this lane's 4 columns, (lo, hi) packed -----------------------
      // (the two rows INTERLEAVED, tap by tap: eight independent accumulation chains — with four, the compiler's hazard recogniser puts
      // one s_nop behind every round of inline-assembly FMAs, mifwt_stream.h; same products in the same order per chain)
      f2 v[2][4];
#pragma unroll
      for (int m = 0; m < L; ++m) {
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
          // c[j] = sum_m h[m] * x_ext[2j + 1 - m];  ring index of row 2j+1-m is 4p + 2rr + (L-1) - m
          const f4 xv = ring[(4 * u + 2 * rr + (L - 1) - m) % RING];
          const f2 x01 = {xv.x, xv.y}, x23 = {xv.z, xv.w};
          if (m == 0) {
            v[rr][0] = pkmul_lo(a.tap[0], x01);
            v[rr][1] = pkmul_hi(a.tap[0], x01);
            v[rr][2] = pkmul_lo(a.tap[0], x23);
            v[rr][3] = pkmul_hi(a.tap[0], x23);
          } else {
            pkfma_lo(v[rr][0], a.tap[m], x01);
            pkfma_hi(v[rr][1], a.tap[m], x01);
            pkfma_lo(v[rr][2], a.tap[m], x23);
            pkfma_hi(v[rr][3], a.tap[m], x23);
          }
        }
      }
#pragma unroll
      for (int rr = 0; rr < 2; ++rr) {
        if (EDGE) {  // implicit-zero columns (the vertical filter commutes with the column extension)
#pragma unroll
          for (int c = 0; c < 4; ++c) v[rr][c] *= cmask[c];
        }
        *reinterpret_cast<f4*>(wr + rr * kLdsRowFloats) = (f4){v[rr][0].x, v[rr][0].y, v[rr][1].x, v[rr][1].y};
        *reinterpret_cast<f4*>(wr + rr * kLdsRowFloats + 4) = (f4){v[rr][2].x, v[rr][2].y, v[rr][3].x, v[rr][3].y};
      }
      wave_lds_fence();

      // ---- horizontal pass: 4 output columns x 4 bands of one row --------------------------------------
      f2 w[4 * NCH];  // w[i] = vertical (low, high) result of column 8q + i
#pragma unroll
      for (int c = 0; c < 2 * NCH; ++c) {
        const f4 t = *reinterpret_cast<const f4*>(rdp + (c + (c >> 2)) * 4);
        w[2 * c] = (f2){t.x, t.y};
        w[2 * c + 1] = (f2){t.z, t.w};
      }
      wave_lds_fence();

      f2 ol[4], oh[4];  // ol[e] = (aa, ad), oh[e] = (da, dd) of output column e
#pragma unroll
      for (int m = 0; m < L; ++m) {  // (the four columns interleaved, tap by tap: eight chains, see the vertical pass)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int idx = 2 * e + 1 + R4 - m;  // extended column 2k+1-m relative to this lane's chunk base
          if (m == 0) {
            ol[e] = pkmul_lo(a.tap[0], w[idx]);
            oh[e] = pkmul_hi(a.tap[0], w[idx]);
          } else {
            pkfma_lo(ol[e], a.tap[m], w[idx]);
            pkfma_hi(oh[e], a.tap[m], w[idx]);
          }
        }
      }
      const f4 o[4] = {{ol[0].x, ol[1].x, ol[2].x, ol[3].x},
                       {ol[0].y, ol[1].y, ol[2].y, ol[3].y},
                       {oh[0].x, oh[1].x, oh[2].x, oh[3].x},
                       {oh[0].y, oh[1].y, oh[2].y, oh[3].y}};
      if (hactive && j0 + 2 * p + hrow < j1) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          char* __restrict__ op = obase[s] + (int64_t)p * opair_bytes[s];
          if (!EDGE || full4) {
            if (a.nt_store)
              __builtin_nontemporal_store(o[s], reinterpret_cast<f4u*>(op + ooff[s]));
            else
              *reinterpret_cast<f4u*>(op + ooff[s]) = o[s];
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (kcol + e < k_end) *reinterpret_cast<float*>(op + ooff[s] + 4 * e) = o[s][e];
          }
        }
      }
    }
  }
}

template <int L, int D>
__global__ void __launch_bounds__(256, (L <= 8 && D <= 2) ? 3 : 2) dwt2_fwd_stream_kernel(const Dwt2FwdArgs<L> a) {
  using C = Cfg<L, D>;
  __shared__ __attribute__((aligned(16))) float lds_all[4][2][kLdsRowFloats];  // [wave][row][col x (lo,hi), padded]
  __shared__ __attribute__((aligned(16))) int rowtab_all[4][2 * kRowTab];  // [wave]: source rows, then 0/1 row masks

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // scalar: everything below is wave-uniform

  // XCD-aware block remap (block b runs on XCD b % 8): give each XCD a contiguous range of tasks so that
  // strips / chunks that share halo columns / rows meet in the same L2.
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int task = bid * 4 + wave;
  if (task >= a.ntasks) return;
  const int strip = task % a.nstrips;
  const int chunk = (task / a.nstrips) % a.nchunks;
  const int img = task / (a.nstrips * a.nchunks);

  const int j0 = chunk * a.rows_per_chunk;
  const int j1 = min(j0 + a.rows_per_chunk, a.Ho);
  float(*lds)[kLdsRowFloats] = lds_all[wave];
  int* rowtab = rowtab_all[wave];

  if (strip < a.n_int) {
    const int k_base = a.kl + strip * C::KS;
    const int k_end = min(k_base + C::KS, a.ke);
    strip_body<L, D, false>(a, lds, rowtab, lane, img, k_base, k_end, j0, j1);
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
    strip_body<L, D, true>(a, lds, rowtab, lane, img, k_base, k_end, j0, j1);
  }
}

template <int L, int D>
int launch(const mifwt_level_desc* d, const void* x, void* approx, void* const* details, const double* lo,
           const double* hi, hipStream_t stream) {
  using C = Cfg<L, D>;
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
  a.nt_store = g_options[MIFWT_OPT_NT_STORE];
  for (int m = 0; m < L; ++m) {
    a.tap[m].x = (float)lo[m];
    a.tap[m].y = (float)hi[m];
  }
  // column partition: [0, kl) left edge | [kl, ke) interior | [ke, Wo) right edge.
  // interior outputs k need extended columns 2k-(L-2) .. 2k+1, all inside [0, W4) with W4 = W & ~3, so
  // that every 4-column group an interior lane loads is completely inside the row.
  const int w4 = a.W & ~3;
  int kl = C::KL < a.Wo ? C::KL : a.Wo;
  int ke = kl;
  if (w4 / 2 > kl) ke = kl + ((w4 / 2 - kl) & ~3);
  if (ke > a.Wo) ke = kl + ((a.Wo - kl) & ~3);
  // a short interior remainder rides with the right edge strip instead of occupying a wave of its own
  {
    const int rem = (ke - kl) % C::KS;
    if (rem > 0 && rem <= 32 && rem + (a.Wo - ke) <= C::KS) ke -= rem;
  }
  a.kl = kl;
  a.ke = ke;
  a.n_int = (ke - kl + C::KS - 1) / C::KS;
  a.n_edge_r = (a.Wo - ke + C::KS - 1) / C::KS;
  a.nstrips = a.n_int + a.n_edge_r + (kl > 0 ? 1 : 0);
  // (A workgroup-cooperative full-line output writer — one workgroup spanning the row, per-band LDS stage, only complete
  // 128-byte lines stored — was built and measured: 0.139 vs 0.117 ms on config 2 level 1; the per-iteration workgroup
  // barrier and the lower count of streaming waves per CU cost more than full-line stores gain.  Removed.)
  // Independent-wave kernel.  Rows per chunk: short chunks win on MI355X (8 rows beats 16/32/64 by 10-30 %):
  // the L-2 halo rows are re-read from L2, not HBM, while many short tasks keep every CU's wave slots
  // refilled and balanced.
  int rpc = 8;
  const int64_t per_chunk_units = (int64_t)d->batch * a.nstrips;
  if (g_options[MIFWT_OPT_ROWS_PER_CHUNK] > 0) rpc = (g_options[MIFWT_OPT_ROWS_PER_CHUNK] + 1) & ~1;
  if (rpc > kMaxRowsPerChunk) rpc = kMaxRowsPerChunk;
  a.rows_per_chunk = rpc;
  a.nchunks = (a.Ho + rpc - 1) / rpc;
  const int64_t ntasks = per_chunk_units * a.nchunks;
  if (ntasks > INT32_MAX / 8) return MIFWT_ERR_UNSUPPORTED;
  a.ntasks = (int)ntasks;
  const unsigned nblk = (unsigned)((ntasks + 3) / 4);
  hipLaunchKernelGGL((dwt2_fwd_stream_kernel<L, D>), dim3(nblk), dim3(256), 0, stream, a);
  return hipGetLastError() == hipSuccess ? MIFWT_OK : MIFWT_ERR_LAUNCH;
}

}  // namespace

bool dwt2_fwd_stream_supported(const mifwt_level_desc* d) {
  if (d->ndim != 2 || d->dtype != MIFWT_F32) return false;
  const int L = d->filt_len;
  if (L < 2 || L > 16 || (L & 1)) return false;
  if (d->sig_stride[2] != 1 || d->approx_stride[2] != 1 || d->detail_stride[2] != 1) return false;
  // one image must be addressable with 32-bit byte offsets (buffer-resource loads)
  const int64_t span = (d->sig_extent[0] - 1) * d->sig_stride[1] + d->sig_extent[1];
  if (d->sig_stride[1] < 0 || span >= (int64_t(1) << 29)) return false;
  if (d->approx_stride[1] < 0 || d->detail_stride[1] < 0) return false;
  if (d->approx_stride[1] >= (int64_t(1) << 29) || d->detail_stride[1] >= (int64_t(1) << 29)) return false;
  return true;
}

template <int L>
int launch_depth(const mifwt_level_desc* d, const void* x, void* approx, void* const* details, const double* lo,
                 const double* hi, hipStream_t stream) {
  // prefetch depth (row pairs in flight beyond the current window); MIFWT_OPT_PREFETCH_PAIRS overrides
  int depth = g_options[MIFWT_OPT_PREFETCH_PAIRS];
  // two pairs ahead pays on big planes; small planes (later levels) prefer the higher occupancy of depth 1
  if (depth <= 0) depth = (L <= 8 && d->sig_extent[0] * d->sig_extent[1] >= (1 << 19)) ? 2 : 1;
  if (L <= 8 && depth >= 3) return launch<L, 3>(d, x, approx, details, lo, hi, stream);
  if (depth >= 2) return launch<L, 2>(d, x, approx, details, lo, hi, stream);
  return launch<L, 1>(d, x, approx, details, lo, hi, stream);
}

int dwt2_fwd_stream(const mifwt_level_desc* d, const void* x, void* approx, void* const* details,
                    const double* lo, const double* hi, hipStream_t stream) {
  switch (d->filt_len) {
    case 2: return launch_depth<2>(d, x, approx, details, lo, hi, stream);
    case 4: return launch_depth<4>(d, x, approx, details, lo, hi, stream);
    case 6: return launch_depth<6>(d, x, approx, details, lo, hi, stream);
    case 8: return launch_depth<8>(d, x, approx, details, lo, hi, stream);
    case 10: return launch_depth<10>(d, x, approx, details, lo, hi, stream);
    case 12: return launch_depth<12>(d, x, approx, details, lo, hi, stream);
    case 14: return launch_depth<14>(d, x, approx, details, lo, hi, stream);
    case 16: return launch_depth<16>(d, x, approx, details, lo, hi, stream);
    default: return MIFWT_ERR_UNSUPPORTED;
  }
}

}  // namespace mifwt
