// mifwt_dwt2_fwd_mfma.hip — fused 2-D analysis level for LONG filters on f16 data with the matrix cores (gfx950), id 11.
//
// Same seam as the other fused 2-D analysis kernels (F.pad + F.conv2d + split, reference
// src/ptwt/conv_transform_2.py:142-149; separable form: separable_conv_transform.py:38-72).  With 18..32 taps the
// level is no longer HBM- but FMA-bound on the vector ALUs (64 FMA per input sample; the vector LDS-tile kernel spends
// 6.0 ms on level 1 of BASELINE config 5's 32-image slice against 1.9 ms of HBM time).  A stride-2 filter bank over a
// block of 16 outputs is a banded-Toeplitz product:
//     [lo[0..16), hi[0..16)] (32)  =  T (32 x 64)  .  x_window (64),      T[(band, k), j] = h_band[2k + L - 1 - j]
// (zero outside 0 <= 2k + L - 1 - j < L; half of T is structural zeros — the price of the stride), i.e. a GEMM
// with M = 32, K = 64 and N = as many independent rows / columns as one likes: v_mfma_f32_32x32x16_f16, four K-steps.
//
// A 256-thread workgroup owns 16 x 64 coefficients (all four bands) of one image:
//   1. the 64 x 160 input tile (f16; boundary extension as index maps, out-of-range = 0) -> LDS, one burst;
//   2. horizontal pass on the matrix cores: B = 16-byte row fragments of the tile straight from LDS (row pitch chosen
//      conflict-free), A = T from registers, f32 accumulate; result written TRANSPOSED as f16 ([band][column][row]) so
//      that
//   3. the vertical pass is the same GEMM with the same T: B = 16-byte column fragments, D -> global stores.
// Taps enter the matrix cores as f16 PAIRS (t = t_hi + t_lo, two MFMAs per K-step): f32-accurate filters; the data are
// f16 by definition of this storage type, the intermediate (lo, hi) image is rounded to f16 once.
// Envelope: f16 storage, even L in [18, 32] (shorter filters are HBM-bound in the vector kernels already).
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
extern unsigned long long* g_pyr_prof;  // (mifwt_dwt2_fwd_pyr.hip; set by mifwt_pyr_profile_buffer)
}

namespace mifwt {

namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16x __attribute__((ext_vector_type(16)));

constexpr int kMR = 16;    // output rows per tile
constexpr int kMC = 64;    // output columns per tile
constexpr int kIR = 64;    // input rows of a tile (>= 2 * 16 + L - 2)
constexpr int kIC = 160;   // input columns of a tile (>= 2 * 64 + L - 2)
constexpr int kXP = 168;   // LDS pitch of the input tile in halfs: 336 B, conflict-free 16-byte row fragments
constexpr int kHP = 72;    // LDS pitch of the transposed (lo, hi) image in halfs: 144 B, conflict-free as well

struct MfmaArgs {
  const _Float16* x;
  _Float16* out[4];  // bands aa, ad, da, dd
  int64_t xs_b, xs_h;
  int64_t os_b[4], os_h[4];
  int H, W, Ho, Wo;
  int tiles_c, tiles_r, ntiles;
  unsigned long long* prof;     // walk kernel, profiling build: 8 counters per wave (tools/mfma_walk_prof.py)
  int dbg;                      // walk kernel, MIFWT_OPT_DEBUG: 1 = no stores, 2 = no loads, 4 = no matrix work
  int seg_tiles, segs, nunits;  // walk kernel: a unit = seg_tiles vertically stacked tiles of one 64-column panel of an image
  int mode, L;
  float lo[32], hi[32];  // dec taps, zero-padded to 32
};

__global__ void __launch_bounds__(256, 4) dwt2_fwd_mfma_kernel(const MfmaArgs a) {
  __shared__ __attribute__((aligned(16))) _Float16 xt[kIR * kXP];
  __shared__ __attribute__((aligned(16))) _Float16 ht[2 * kMC * kHP];
  __shared__ float taps[64];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  __builtin_assume(wave >= 0 && wave < 4);
  // boundary extension = the branch-free single-fold map (Fold1, mifwt_stream.h; the launcher requires extents >= L)
  const bool zero_mode = a.mode == MIFWT_MODE_ZERO;
  Fold1 fold;
  fold.set(a.mode);
  const int L = a.L;
  const int n = lane & 31, half = lane >> 5;
  // (the window starts P = L - 2 + s samples before the first output's pair, as in the walk kernel below: the same sums in the same
  // order, so that the two kernels agree to the bit)
  const int P = L - 2 + ((8 - ((L - 2) & 7)) & 7);

  // ---- T fragments, once per (persistent) workgroup: T[i][j] = h_band(i)[2 (i & 15) + L - 1 - j] with i = l & 31,
  // j = 16 c + 8 (l >> 5) + e; f16 pairs (t = t_hi + t_lo)
  if (threadIdx.x < 64) taps[threadIdx.x] = threadIdx.x < 32 ? a.lo[threadIdx.x] : a.hi[threadIdx.x - 32];
  __syncthreads();
  h8 ahi[4], alo[4];
  {
    const int band = n >> 4, kq = n & 15;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int j = 16 * c + 8 * half + e;
        const int m = 2 * kq + P + 1 - j;
        const float t = (m >= 0 && m < L) ? taps[32 * band + m] : 0.f;
        const _Float16 th = (_Float16)t;
        ahi[c][e] = th;
        alo[c][e] = (_Float16)(t - (float)th);
      }
    }
  }

  // Persistent workgroups (the T fragments above cost about as much as one tile's MFMAs).  Block b runs on XCD b % 8;
  // every XCD gets one contiguous eighth of the (image, tile row, tile column) sequence and its blocks walk through it
  // together, so that tiles stacked vertically — which share half of their input rows — meet in one L2.
  const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3, nq = gridDim.x >> 3;  // the grid is a multiple of 8
  const int t_begin = (int)(((int64_t)a.ntiles * xcd) >> 3), t_end = (int)(((int64_t)a.ntiles * (xcd + 1)) >> 3);
  // Each XCD starts an eighth further into its panel: with one image per panel (8 images) all XCDs would otherwise sit
  // at the same tile of their image, i.e. at addresses a whole image stride (a power of two) apart — measured 15x
  // slower (every access of the chip lands in the same few memory channels).
  const int panel = t_end - t_begin, rot = (int)(((int64_t)panel * xcd) >> 3);
  constexpr uint32_t kOob = 0x80000000u;
  const uint32_t row_bytes = (uint32_t)a.xs_h * 2u;
  const uint32_t img_bytes = ((uint32_t)(a.H - 1) * (uint32_t)a.xs_h + (uint32_t)a.W) * 2u;
  const bool aligned4 = (a.xs_h & 1) == 0 && (a.xs_b & 1) == 0 && (reinterpret_cast<uintptr_t>(a.x) & 3) == 0;

  struct Tile {
    int img, k0, j0;
    bool pairs;  // staged as column pairs (dwords) instead of single halfs
  };
  auto locate = [&](int it) -> Tile {
    int pos = it + rot;
    if (pos >= panel) pos -= panel;
    const int tile = t_begin + pos;
    const int trow = tile / a.tiles_c, tc = tile - trow * a.tiles_c;  // trow = img * tiles_r + tr
    Tile t;
    t.img = trow / a.tiles_r;
    t.k0 = tc * kMC;
    t.j0 = (trow - t.img * a.tiles_r) * kMR;
    const int c_first = 2 * t.k0 - P;
    // column pairs as dwords when the tile's columns lie inside the image and every row starts 4-byte aligned
    t.pairs = aligned4 && c_first >= 0 && c_first + kIC <= a.W;
    return t;
  };
  // stage 1a: request a tile's 64 x 160 input window into registers (boundary extension as index maps; everything
  // outside the needed window and all implicit zeros are out-of-range offsets of the buffer resource)
  auto request = [&](const Tile& t, uint32_t (&v)[16][3]) {
    const __amdgpu_buffer_rsrc_t xrsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(a.x + (int64_t)t.img * a.xs_b), 0, img_bytes, 0x00020000);
    const int nc_need = 2 * (min(t.k0 + kMC, a.Wo) - t.k0) + P;
    const int nr_need = 2 * (min(t.j0 + kMR, a.Ho) - t.j0) + P;
    const int c_first = 2 * t.k0 - P, r_first = 2 * t.j0 - P;
    if (t.pairs) {
      uint32_t poff[2];
#pragma unroll
      for (int qq = 0; qq < 2; ++qq) poff[qq] = lane + 64 * qq < kIC / 2 ? 2u * (uint32_t)(c_first + 2 * (lane + 64 * qq)) : kOob;
      if (r_first >= 0 && r_first + kIR <= a.H) {  // rows inside the image: no map, one add per row
        uint32_t soff = (uint32_t)(r_first + wave) * row_bytes;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
#pragma unroll
          for (int qq = 0; qq < 2; ++qq) v[i][qq] = __builtin_amdgcn_raw_buffer_load_b32(xrsrc, poff[qq], soff, 0);
          soff += 4u * row_bytes;
        }
      } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int r = wave + 4 * i;  // wave-uniform
          const int ri = r_first + r;
          const bool dead = r >= nr_need || (zero_mode && (unsigned)ri >= (unsigned)a.H);
          const uint32_t soff = __builtin_amdgcn_readfirstlane(dead ? 0u : (uint32_t)fold(ri, a.H) * row_bytes);
#pragma unroll
          for (int qq = 0; qq < 2; ++qq) v[i][qq] = __builtin_amdgcn_raw_buffer_load_b32(xrsrc, dead ? kOob : poff[qq], soff, 0);
        }
      }
    } else {
      uint32_t coff[3];
#pragma unroll
      for (int qq = 0; qq < 3; ++qq) {
        const int c = lane + 64 * qq;
        const int ci = c_first + c;
        const bool dead = c >= nc_need || (zero_mode && (unsigned)ci >= (unsigned)a.W);
        coff[qq] = dead ? kOob : 2u * (uint32_t)fold(ci, a.W);
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int r = wave + 4 * i;  // wave-uniform
        const int ri = r_first + r;
        const bool dead = r >= nr_need || (zero_mode && (unsigned)ri >= (unsigned)a.H);
        const uint32_t soff = __builtin_amdgcn_readfirstlane(dead ? 0u : (uint32_t)fold(ri, a.H) * row_bytes);
#pragma unroll
        for (int qq = 0; qq < 3; ++qq) v[i][qq] = __builtin_amdgcn_raw_buffer_load_b16(xrsrc, dead ? kOob : coff[qq], soff, 0);
      }
    }
  };
  // stage 1b: park the requested window in LDS
  auto commit = [&](const Tile& t, const uint32_t (&v)[16][3]) {
    if (t.pairs) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int r = wave + 4 * i;
#pragma unroll
        for (int qq = 0; qq < 2; ++qq)
          if (lane + 64 * qq < kXP / 2) *reinterpret_cast<uint32_t*>(&xt[r * kXP + 2 * (lane + 64 * qq)]) = v[i][qq];
      }
    } else {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int r = wave + 4 * i;
#pragma unroll
        for (int qq = 0; qq < 3; ++qq)
          if (lane + 64 * qq < kXP) xt[r * kXP + lane + 64 * qq] = __builtin_bit_cast(_Float16, (unsigned short)v[i][qq]);
      }
    }
  };

  // Software pipeline over this block's tiles: the NEXT tile's window is requested into registers before the matrix
  // work of the current one and parked in LDS once the horizontal pass has released the tile buffer.
  uint32_t stage[16][3];
  if (q >= panel) return;
  Tile cur = locate(q);
  request(cur, stage);
  commit(cur, stage);
  __syncthreads();
  for (int it = q; it < panel; it += nq) {
    const bool has_next = it + nq < panel;
    Tile nxt = cur;
    if (has_next) {
      nxt = locate(it + nq);
      request(nxt, stage);
    }

    // ---- 2. horizontal pass: (row group rg, output block kb) jobs, two per wave.  D[r][(band, kq)] = X[r][j] . T^T: the
    // tile rows are the A operand, T the B operand, so that a lane ends up with ONE output column and four consecutive rows
    // per register quad — the transposed (lo, hi) image is written with 8-byte stores
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      const int job = wave * 2 + jj;
      const int rg = job >> 2, kb = job & 3;
      f16x acc;
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const h8 xf = *reinterpret_cast<const h8*>(&xt[(rg * 32 + n) * kXP + 32 * kb + 16 * c + 8 * half]);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(xf, ahi[c], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(xf, alo[c], acc, 0, 0, 0);
      }
      // D[i][col]: col = n -> (band, kq);  i = (e & 3) + 8 (e >> 2) + 4 half -> tile row rg * 32 + i
      const int band = n >> 4, kq = n & 15;
      _Float16* hrow = &ht[(band * kMC + kb * 16 + kq) * kHP + rg * 32 + 4 * half];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        typedef _Float16 h4 __attribute__((ext_vector_type(4)));
        *reinterpret_cast<h4*>(hrow + 8 * g) =
            (h4){(_Float16)acc[4 * g], (_Float16)acc[4 * g + 1], (_Float16)acc[4 * g + 2], (_Float16)acc[4 * g + 3]};
      }
    }
    __syncthreads();  // ht complete, xt released

    if (has_next) commit(nxt, stage);

    // ---- 3. vertical pass: one (horizontal band, column group) job per wave -------------------------------------------------
    {
      const int bh = wave >> 1, cg = wave & 1;
      f16x acc;
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const h8 b = *reinterpret_cast<const h8*>(&ht[(bh * kMC + cg * 32 + n) * kHP + 16 * c + 8 * half]);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi[c], b, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(alo[c], b, acc, 0, 0, 0);
      }
      // D[i][col]: i = (e & 3) + 8 (e >> 2) + 4 half -> vertical band bv = e >> 3, output row jr + 4 half with
      // jr = (e & 3) + 8 ((e >> 2) & 1);  one per-lane base pointer per vertical band, the rest is wave-uniform
      const int k = cur.k0 + cg * 32 + n;
      if (k < a.Wo) {
#pragma unroll
        for (int bv = 0; bv < 2; ++bv) {
          const int s = 2 * bv + bh;  // band: bit 1 = vertical (axis -2) high, bit 0 = horizontal high
          _Float16* base = a.out[s] + (int64_t)cur.img * a.os_b[s] + (int64_t)(cur.j0 + 4 * half) * a.os_h[s] + k;
          const int jlim = a.Ho - cur.j0 - 4 * half;  // rows jr of this lane with jr < jlim exist
#pragma unroll
          for (int g = 0; g < 8; ++g) {
            const int jr = (g & 3) + 8 * (g >> 2);
            if (jr < jlim) base[(int64_t)jr * a.os_h[s]] = (_Float16)acc[8 * bv + g];
          }
        }
      }
    }
    __syncthreads();  // next tile's window parked, ht released
    cur = nxt;
  }
}


// ---- the same tile, WALKED down a 64-column panel -----------------------------------------------------------------------------
// Vertically stacked tiles share half of their 64 input rows, and so do their horizontally filtered (lo, hi) images.  A workgroup
// walks down seg_tiles stacked tiles of one panel: per tile only the 32 NEW input rows (a "chunk") are fetched and filtered along the
// rows (one MFMA job per wave instead of two) into one half of a 64-row ring of the transposed (lo, hi) image; the vertical pass reads
// its 64-row window from the ring (older half first).  One extra chunk per unit primes the ring.  Against the tile kernel above: half
// the window reads (2.5x -> 1.25x of the plane through L2), two thirds of the MFMAs, the same sums in the same order.
//
// The tile kernel and the first version of the walk were bound by instruction issue and by waits, not by the matrix cores or HBM (with
// loads, stores and MFMAs switched off the launch still took 1.35 of 3.1 ms; tools/mfma_walk_parts.py), so this one is built to issue
// little:
//   * a fifth wave is the LOADER: a chunk = 32 rows x 21 sixteen-byte pieces (20 of data, one of padding = the LDS pitch) travels as
//     11 LDS-DMA requests (buffer_load_dwordx4 ... lds; global addresses need only 2-byte alignment, tools/dma_probe.hip), two chunks
//     double-buffered; the matrix waves never wait for a load and their stores never delay one;
//   * columns a panel needs from outside the plane (first / last panel) are patched into the landed chunk by the loader: their
//     boundary-mapped samples are requested together with the chunk (one 2-byte load per element into registers);
//   * the vertical pass runs with the operands swapped (D^T): a lane owns ONE output row and four groups of four adjacent columns —
//     four 8-byte stores per lane and band pair instead of sixteen 2-byte ones (rows of an odd pitch start 2-byte aligned: such
//     stores work, tools/align_probe.hip).
constexpr int kWR = 32;                            // input rows of a chunk = 2 kMR
constexpr int kWPieces = kXP / 8;                  // 16-byte pieces of an LDS row
constexpr int kWDma = (kWR * kWPieces + 63) / 64;  // requests per chunk
constexpr int kWChunkBytes = kWR * kXP * 2;
constexpr int kWLdsBytes = 2 * kWChunkBytes + 2 * kMC * kHP * 2;
constexpr int kWPatch = 16;                        // patched samples per loader lane requested ahead (32 rows x 32 columns; more: on the spot)

__device__ __forceinline__ void mfma_dma16(uint32_t voff, __amdgpu_buffer_rsrc_t rsrc, uint32_t lds_addr) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen" MIFWT_MFMA_DMA_POLICY " lds" ::"v"(voff), "s"(rsrc), "s"(lds_addr) : "memory");
}

template <bool PROF>
__global__ void __launch_bounds__(320, 5) dwt2_fwd_mfma_walk_kernel(const MfmaArgs a) {
  unsigned long long pc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pt = PROF ? __builtin_readcyclecounter() : 0;
  auto lap = [&](int k) {
    if constexpr (PROF) {
      const unsigned long long now = __builtin_readcyclecounter();
      pc[k] += now - pt;
      pt = now;
    }
  };
  auto dump = [&]() {
    if constexpr (PROF) {
      if ((threadIdx.x & 63) == 0)
        for (int k = 0; k < 8; ++k) MIFWT_PROFP(a)[((size_t)blockIdx.x * 5 + (threadIdx.x >> 6)) * 8 + k] = pc[k];
    }
  };
  extern __shared__ __attribute__((aligned(16))) unsigned char wsm[];
  _Float16* const xt = reinterpret_cast<_Float16*>(wsm);                     // two chunks
  _Float16* const ht = reinterpret_cast<_Float16*>(wsm + 2 * kWChunkBytes);  // ring of the transposed (lo, hi) image
  float* const taps = reinterpret_cast<float*>(ht);                          // (until the first horizontal pass)

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  __builtin_assume(wave >= 0 && wave < 5);
  const bool zero_mode = a.mode == MIFWT_MODE_ZERO;
  Fold1 fold;
  fold.set(a.mode);
  const int L = a.L;
  // the window of a tile starts P = L - 2 + s samples before its first output's pair, s = the fewest samples that make P a multiple
  // of 8: the 16-byte pieces of a chunk row then start on 16-byte boundaries (planes with 16-byte aligned rows); the 64-sample window
  // of 16 outputs still holds all their taps (2 * 15 + L - 1 + s <= 63 for every L <= 32).  tests/test_mfma_walk_model.py
  const int P = L - 2 + ((8 - ((L - 2) & 7)) & 7);

  // units (image, row segment, panel) with the panel index fastest: the blocks of an XCD walk down neighbouring panels (which share
  // 32 of their 160 columns) at the same time; staggered starting points as in the tile kernel
  const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3, nq = gridDim.x >> 3;
  const int u_begin = (int)(((int64_t)a.nunits * xcd) >> 3), u_end = (int)(((int64_t)a.nunits * (xcd + 1)) >> 3);
  const int panel = u_end - u_begin, rot = (int)(((int64_t)panel * xcd) >> 3);
  if (q >= panel) return;
  if (threadIdx.x < 64) taps[threadIdx.x] = threadIdx.x < 32 ? a.lo[threadIdx.x] : a.hi[threadIdx.x - 32];
  __syncthreads();
  struct Unit {
    int img, k0, tr0, nt;  // image, first output column, first tile row, tiles
  };
  auto locate = [&](int it) -> Unit {
    int pos = it + rot;
    if (pos >= panel) pos -= panel;
    const int idx = u_begin + pos;
    const int rest = idx / a.tiles_c, tc = idx - rest * a.tiles_c;
    Unit u;
    u.img = rest / a.segs;
    u.tr0 = (rest - u.img * a.segs) * a.seg_tiles;
    u.nt = min(a.seg_tiles, a.tiles_r - u.tr0);
    u.k0 = tc * kMC;
    return u;
  };
  int it = q;
  // (unit, chunk) after (u, gg) in this block's sequence; false at the end.  Chunk g of a unit = extended input rows
  // r_first + 32 g .. + 31 with r_first = 2 kMR tr0 - P; tile tr0 + g - 1 needs chunks g - 1 and g.
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
    const uint32_t row_bytes = (uint32_t)a.xs_h * 2u;
    const uint32_t img_bytes = (MIFWT_DBG(a) & 2) ? 0u : ((uint32_t)(a.H - 1) * (uint32_t)a.xs_h + (uint32_t)a.W) * 2u;
    // request j of a chunk: lane -> piece 64 j + lane = (row, piece of the row); the padding piece requests nothing, the lanes past
    // the chunk's end are switched off
    uint32_t vfast[kWDma];
#pragma unroll
    for (int j = 0; j < kWDma; ++j) {
      const int P = 64 * j + lane, row = P / kWPieces, piece = P - row * kWPieces;
      vfast[j] = piece == kWPieces - 1 ? kOob : (uint32_t)row * row_bytes + 16u * (uint32_t)piece;
    }
    const bool last_live = lane < kWR * kWPieces - 64 * (kWDma - 1);
    const uint32_t lds0 = (uint32_t)(uintptr_t)xt;  // (LDS offset of the first chunk buffer)

    struct Geo {
      int c_first, r_first;
    };
    auto geo = [&](const Unit& u, int g) -> Geo { return {2 * u.k0 - P, 2 * kMR * u.tr0 - P + kWR * g}; };
    auto rsrc_of = [&](const Unit& u) {
      return __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(a.x + (int64_t)u.img * a.xs_b), 0, img_bytes, 0x00020000);
    };
    const int r_end = 2 * a.Ho;  // extended rows from here on feed no stored output
    auto issue_dma = [&](const Unit& u, int g, int buf) {
      const __amdgpu_buffer_rsrc_t xrsrc = rsrc_of(u);
      const Geo ge = geo(u, g);
      const uint32_t dst = lds0 + (uint32_t)(buf * kWChunkBytes);
      if (ge.r_first >= 0 && ge.r_first + kWR <= a.H) {
        // (columns left of the plane: the sum wraps into the row above or out of range; those samples are patched)
        const uint32_t base = (uint32_t)ge.r_first * row_bytes + (uint32_t)(2 * ge.c_first);
#pragma unroll
        for (int j = 0; j < kWDma; ++j) {
          const uint32_t v = vfast[j] == kOob ? kOob : vfast[j] + base;
          if (j < kWDma - 1 || last_live) mfma_dma16(v, xrsrc, dst + 1024u * (uint32_t)j);
        }
      } else {
#pragma unroll
        for (int j = 0; j < kWDma; ++j) {
          const int P = 64 * j + lane, row = P / kWPieces, piece = P - row * kWPieces;
          const int ri = ge.r_first + row;
          const bool dead = piece == kWPieces - 1 || ri >= r_end || (zero_mode && (unsigned)ri >= (unsigned)a.H);
          const uint32_t v = dead ? kOob : (uint32_t)fold(ri, a.H) * row_bytes + (uint32_t)(2 * ge.c_first) + 16u * (uint32_t)piece;
          if (j < kWDma - 1 || last_live) mfma_dma16(v, xrsrc, dst + 1024u * (uint32_t)j);
        }
      }
    };
    // Samples outside the plane's columns (first / last panel): element e = lane + 64 t of the list (row, patched column); window
    // columns [0, nl) on the left, [nr0, nr1) on the right; the window columns from nr1 on feed no stored coefficient and are
    // zeroed.  The first kWPatch per lane are requested a step ahead.
    uint32_t pv[kWPatch];
    auto patch_cols = [&](const Unit& u, const Geo& ge, int& nl, int& nr0, int& nr1) -> int {
      // (whole 16-byte pieces on the left: a piece that starts before the plane's first sample is out of range as a whole; whole
      // dwords on the right: the dword that holds the last sample of an odd-width plane's last row ends out of range)
      nl = min(kIC, (max(0, -ge.c_first) + 7) & ~7);
      nr0 = max(nl, min(kIC, (a.W - ge.c_first) & ~1));
      nr1 = max(nr0, min(kIC, 2 * (min(u.k0 + kMC, a.Wo) - u.k0) + P));  // (nc_need of the tile kernel)
      return nl + (nr1 - nr0);
    };
    auto patch_off = [&](const Geo& ge, int nl, int nr0, int ncols, int e, int& row, int& wc) -> uint32_t {
      row = e / ncols;
      const int ce = e - row * ncols;
      wc = ce < nl ? ce : nr0 + (ce - nl);
      const int ri = ge.r_first + row, ci = ge.c_first + wc;
      const bool dead = row >= kWR || ri >= r_end || (zero_mode && ((unsigned)ri >= (unsigned)a.H || (unsigned)ci >= (unsigned)a.W));
      return dead ? kOob : (uint32_t)fold(ri, a.H) * row_bytes + 2u * (uint32_t)fold(ci, a.W);
    };
    auto issue_patch = [&](const Unit& u, int g) {
      const Geo ge = geo(u, g);
      int nl, nr0, nr1;
      const int ncols = patch_cols(u, ge, nl, nr0, nr1);
      if (ncols == 0) return;
      const __amdgpu_buffer_rsrc_t xrsrc = rsrc_of(u);
      const int nt = min(kWPatch, (kWR * ncols + 63) >> 6);
#pragma unroll
      for (int t = 0; t < kWPatch; ++t) {
        if (t < nt) {
          int row, wc;
          pv[t] = __builtin_amdgcn_raw_buffer_load_b16(xrsrc, patch_off(ge, nl, nr0, ncols, lane + 64 * t, row, wc), 0, 0);
        }
      }
    };
    auto write_patch = [&](const Unit& u, int g, int buf) {
      const Geo ge = geo(u, g);
      int nl, nr0, nr1;
      const int ncols = patch_cols(u, ge, nl, nr0, nr1);
      _Float16* xb = xt + buf * (kWR * kXP);
      if (nr1 < kIC) {
        const int nz = kIC - nr1;
        for (int e = lane; e < kWR * nz; e += 64) {
          const int row = e / nz;
          xb[row * kXP + nr1 + (e - row * nz)] = (_Float16)0.f;
        }
      }
      if (ncols == 0) return;
      const int ntot = (kWR * ncols + 63) >> 6, nt = min(kWPatch, ntot);
#pragma unroll
      for (int t = 0; t < kWPatch; ++t) {
        if (t < nt) {
          int row, wc;
          (void)patch_off(ge, nl, nr0, ncols, lane + 64 * t, row, wc);
          if (row < kWR) xb[row * kXP + wc] = __builtin_bit_cast(_Float16, (unsigned short)pv[t]);
        }
      }
      if (ntot > kWPatch) {  // (a last panel that lies mostly outside the plane: the rest on the spot)
        const __amdgpu_buffer_rsrc_t xrsrc = rsrc_of(u);
        for (int t = kWPatch; t < ntot; ++t) {
          int row, wc;
          const uint32_t v = __builtin_amdgcn_raw_buffer_load_b16(xrsrc, patch_off(ge, nl, nr0, ncols, lane + 64 * t, row, wc), 0, 0);
          if (row < kWR) xb[row * kXP + wc] = __builtin_bit_cast(_Float16, (unsigned short)v);
        }
      }
    };

    // Chunk s of this block's sequence lives in buffer s & 1.  Two chunks are in flight: chunk s + 2 is requested as soon as barrier
    // B(s) has released the buffer of chunk s; the patched samples of chunk s + 1 are requested once those of chunk s are written.
    // Requests complete in order: DMA(s), patch(s), DMA(s + 1) — waiting for all but the last kWDma leaves chunk s complete.
    Unit u0 = locate(it), u1 = u0, u2;
    int g0 = 0, g1 = 0, g2;
    bool has1 = advance(u1, g1);
    u2 = u1;
    g2 = g1;
    bool has2 = has1 && advance(u2, g2);
    issue_dma(u0, 0, 0);
    issue_patch(u0, 0);
    if (has1) issue_dma(u1, g1, 1);
    lap(0);
    for (int s = 0;; ++s) {
      if (has1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kWDma) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      lap(1);  // waiting for the chunk
      write_patch(u0, g0, s & 1);
      if (has1) issue_patch(u1, g1);
      lap(2);  // patch
      __syncthreads();  // A(s): chunk s complete in LDS
      lap(3);
      __syncthreads();  // B(s): the horizontal pass has read it
      lap(4);
      if (!has1) break;
      if (has2) issue_dma(u2, g2, s & 1);
      u0 = u1;
      g0 = g1;
      u1 = u2;
      g1 = g2;
      has1 = has2;
      if (has2) has2 = advance(u2, g2);
      lap(5);  // requests + bookkeeping
    }
    dump();
    return;
  }

  // =============================================================================================================================
  // matrix waves
  // (wave priorities measured: the loader raised: 2.29 against 2.10 ms; the matrix waves raised: 2.16-2.19 against 2.09-2.20 — none)
  const int n = lane & 31, half = lane >> 5;
  // T fragments: T[i][j] = h_band(i)[2 (i & 15) + P + 1 - j] with i = l & 31, j = 16 c + 8 (l >> 5) + e; f16 pairs (t = t_hi + t_lo)
  h8 ahi[4], alo[4];
  {
    const int band = n >> 4, kq = n & 15;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int j = 16 * c + 8 * half + e;
        const int m = 2 * kq + P + 1 - j;
        const float t = (m >= 0 && m < L) ? taps[32 * band + m] : 0.f;
        const _Float16 th = (_Float16)t;
        ahi[c][e] = th;
        alo[c][e] = (_Float16)(t - (float)th);
      }
    }
  }

  Unit cur = locate(it);
  int g = 0, s = 0;
  lap(0);
  for (;;) {
    Unit un = cur;
    int gn = g;
    const bool has_next = advance(un, gn);
    lap(1);  // bookkeeping
    __syncthreads();  // A(s): chunk s is in LDS, the ring half it goes to is no longer read
    lap(2);
    // ---- horizontal pass of the chunk: wave = output block of 16 columns; D[row][(band, kq)] -> ring half g & 1, transposed
    {
      const int kb = wave;
      const _Float16* xb = xt + (s & 1) * (kWR * kXP);
      f16x acc;
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const h8 xf = *reinterpret_cast<const h8*>(&xb[n * kXP + 32 * kb + 16 * c + 8 * half]);
        if (!(MIFWT_DBG(a) & 4)) {
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(xf, ahi[c], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(xf, alo[c], acc, 0, 0, 0);
        }
      }
      const int band = n >> 4, kq = n & 15;
      _Float16* hrow = &ht[(band * kMC + kb * 16 + kq) * kHP + 32 * (g & 1) + 4 * half];
#pragma unroll
      for (int gg = 0; gg < 4; ++gg) {
        typedef _Float16 h4 __attribute__((ext_vector_type(4)));
        *reinterpret_cast<h4*>(hrow + 8 * gg) =
            (h4){(_Float16)acc[4 * gg], (_Float16)acc[4 * gg + 1], (_Float16)acc[4 * gg + 2], (_Float16)acc[4 * gg + 3]};
      }
    }
    lap(3);  // horizontal pass
    __syncthreads();  // B(s): ring half complete, the chunk buffer released
    lap(4);

    // ---- vertical pass of tile tr0 + g - 1 (window rows 0 .. 31 = chunk g - 1, 32 .. 63 = chunk g), operands swapped:
    // D[column i][(vertical band, row) n]: lane (n, half) owns output row n & 15 of vertical band n >> 4 and the columns
    // cg * 32 + 8 gg + 4 half + (0 .. 3), gg = 0 .. 3
    if (g >= 1) {
      const int bh = wave >> 1, cg = wave & 1;
      const int old = 32 * ((g - 1) & 1);
      f16x acc;
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const h8 b = *reinterpret_cast<const h8*>(&ht[(bh * kMC + cg * 32 + n) * kHP + ((16 * c + 8 * half + old) & 63)]);
        if (!(MIFWT_DBG(a) & 4)) {
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, ahi[c], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, alo[c], acc, 0, 0, 0);
        }
      }
      typedef _Float16 h4 __attribute__((ext_vector_type(4)));
      if (MIFWT_DBG(a) & 1) {
      } else {
        // Transposed through LDS, wave-local: this wave is the only reader of ring columns bh * 64 + cg * 32 + (0 .. 31) in the
        // vertical pass, and their OLDER half is dead once its fragments above are in registers (the next horizontal pass rewrites it
        // behind barrier A).  Row r = (vertical band, output row) of the 32 x 32 block goes to the older half of ring column r:
        // 64 bytes; then every lane stores 16 bytes, four lanes a row.
        _Float16* colbase = &ht[(bh * kMC + cg * 32) * kHP + old];
#pragma unroll
        for (int gg = 0; gg < 4; ++gg)
          *reinterpret_cast<h4*>(colbase + n * kHP + 8 * gg + 4 * half) =
              (h4){(_Float16)acc[4 * gg], (_Float16)acc[4 * gg + 1], (_Float16)acc[4 * gg + 2], (_Float16)acc[4 * gg + 3]};
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int r = 16 * i + (lane >> 2), pc = lane & 3;
          const int sb = 2 * (r >> 4) + bh;  // band: bit 1 = vertical (axis -2) high, bit 0 = horizontal high
          const int j = (cur.tr0 + g - 1) * kMR + (r & 15);
          const h8 v = *reinterpret_cast<const h8*>(colbase + r * kHP + 8 * pc);
          const int kc = cur.k0 + cg * 32 + 8 * pc;  // first of the piece's eight columns
          if (j < a.Ho && kc + 8 > a.Wo) {
            // the LAST panel of a plane: the piece that straddles the plane's last column leaves sample by sample, pieces beyond it do
            // not leave at all.  (Until round 5 the whole last panel was stored column by column, sixteen 2-byte stores per lane and tile:
            // its units were the long pole of every launch on planes of few panels.)
            _Float16* dst = a.out[sb] + (int64_t)cur.img * a.os_b[sb] + (int64_t)j * a.os_h[sb] + kc;
#pragma unroll
            for (int e = 0; e < 8; ++e)
              if (kc + e < a.Wo) dst[e] = v[e];
          } else if (j < a.Ho) {
            _Float16* dst = a.out[sb] + (int64_t)cur.img * a.os_b[sb] + (int64_t)j * a.os_h[sb] + kc;
            // (16-byte stores to 2-byte aligned addresses — rows of an odd pitch — are fine on this hardware, tools/align_probe.hip;
            // the compiler would split them)
            // non-temporal: nothing reads these lines back (config-5 slice, level 1: 2.29 -> 2.15 ms analysis, 2.21 -> 2.18 synthesis;
            // MIFWT_OPT_DEBUG 8 / 16 = write-through / default policy, tools/mfma_policy_ab.py)
            if (MIFWT_DBG(a) & 8) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(dst), "v"(v) : "memory");
            else if (MIFWT_DBG(a) & 16) asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(dst), "v"(v) : "memory");
            else asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(dst), "v"(v) : "memory");
          }
        }
      }
    }
    lap(5);  // vertical pass + stores issued
    if (!has_next) break;
    cur = un;
    g = gn;
    ++s;
  }
  dump();
}

}  // namespace

bool dwt2_fwd_mfma_supported(const mifwt_level_desc* d) {
  if (d->ndim != 2 || d->dtype != MIFWT_F16) return false;
  const int L = d->filt_len;
  if (L < 18 || L > 32 || (L & 1)) return false;
  if (d->sig_stride[2] != 1 || d->approx_stride[2] != 1 || d->detail_stride[2] != 1) return false;
  const int64_t span = (d->sig_extent[0] - 1) * d->sig_stride[1] + d->sig_extent[1];
  if (d->sig_stride[1] < 0 || span >= (int64_t(1) << 29)) return false;
  for (int i = 0; i < 2; ++i)
    if (d->approx_stride[i] < 0 || d->detail_stride[i] < 0) return false;
  if (d->sig_extent[0] < L || d->sig_extent[1] < L) return false;  // single-fold boundary map
  return true;
}

int dwt2_fwd_mfma(const mifwt_level_desc* d, const void* x, void* approx, void* const* details, const double* lo,
                  const double* hi, hipStream_t stream) {
  MfmaArgs a;
  a.x = static_cast<const _Float16*>(x);
  a.out[0] = static_cast<_Float16*>(approx);
  for (int s = 1; s < 4; ++s) a.out[s] = static_cast<_Float16*>(details[s - 1]);
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
  a.L = d->filt_len;
  a.dbg = g_options[MIFWT_OPT_DEBUG];
  for (int m = 0; m < 32; ++m) {
    a.lo[m] = m < d->filt_len ? (float)lo[m] : 0.f;
    a.hi[m] = m < d->filt_len ? (float)hi[m] : 0.f;
  }
  a.tiles_c = (a.Wo + kMC - 1) / kMC;
  a.tiles_r = (a.Ho + kMR - 1) / kMR;
  const int64_t ntiles = (int64_t)d->batch * a.tiles_c * a.tiles_r;
  if (ntiles > INT32_MAX / 8) return MIFWT_ERR_UNSUPPORTED;
  a.ntiles = (int)ntiles;
  // 4 workgroups per CU (LDS) on 256 CUs; the grid is a multiple of 8 (one contiguous panel of tiles per XCD)
  int64_t grid = 256 * 4;
  // the walk wins wherever it was measured (32 x 8192^2: 2.0 against 3.6 ms; 32 x 1051^2: 0.080 against 0.098; 32 x 541^2: 0.049 against
  // 0.051; tools/mfma_walk_ab.py) and serves every call, so that a batch and its images one by one take the same path;
  // MIFWT_OPT_MFMA_MODE 3 = the tile-at-a-time kernel of round 2 (kept for comparisons: bit-identical results)
  const bool walk = g_options[MIFWT_OPT_MFMA_MODE] != 3;
  if (walk) {
    // the walk: units of seg_tiles stacked tiles, about 16 units per workgroup (the priming chunk costs 1 / seg_tiles), at least 4 tiles each
    const int64_t panels = (int64_t)d->batch * a.tiles_c;
    int64_t segs = (16 * grid + panels - 1) / panels;
    segs = std::max<int64_t>(1, std::min<int64_t>(segs, (a.tiles_r + 3) / 4));
    a.seg_tiles = (int)((a.tiles_r + segs - 1) / segs);
    if (g_options[MIFWT_OPT_TILE_ROWS] > 0) a.seg_tiles = std::max(1, std::min(a.tiles_r, g_options[MIFWT_OPT_TILE_ROWS]));  // (experiments)
    a.segs = (a.tiles_r + a.seg_tiles - 1) / a.seg_tiles;
    const int64_t nunits = panels * a.segs;
    if (nunits > INT32_MAX / 8) return MIFWT_ERR_UNSUPPORTED;
    a.nunits = (int)nunits;
    if (nunits < grid) grid = (nunits + 7) & ~int64_t(7);
    a.prof = g_pyr_prof;
    count_launch(MIFWT_VARIANT_FWD_MFMA_WALK);
    if constexpr (kDiag) {
      if (MIFWT_PROFP(a)) {
        hipLaunchKernelGGL(dwt2_fwd_mfma_walk_kernel<kDiag>, dim3((unsigned)grid), dim3(320), kWLdsBytes, stream, a);
        return hipGetLastError() == hipSuccess ? MIFWT_OK : MIFWT_ERR_LAUNCH;
      }
    }
    hipLaunchKernelGGL(dwt2_fwd_mfma_walk_kernel<false>, dim3((unsigned)grid), dim3(320), kWLdsBytes, stream, a);
    return hipGetLastError() == hipSuccess ? MIFWT_OK : MIFWT_ERR_LAUNCH;
  }
  a.seg_tiles = a.segs = a.nunits = 0;
  if (ntiles < grid) grid = (ntiles + 7) & ~int64_t(7);
  count_launch(MIFWT_VARIANT_FWD_MFMA_TILE);
  hipLaunchKernelGGL(dwt2_fwd_mfma_kernel, dim3((unsigned)grid), dim3(256), 0, stream, a);
  return hipGetLastError() == hipSuccess ? MIFWT_OK : MIFWT_ERR_LAUNCH;
}

}  // namespace mifwt
