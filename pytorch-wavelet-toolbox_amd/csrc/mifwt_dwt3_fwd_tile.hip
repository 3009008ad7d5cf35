// mifwt_dwt3_fwd_tile.hip — fully fused LDS-tile 3-D analysis level for short filters (gfx950), kernel id 9.
//
// Replaces, for one level of wavedec3 / fswavedec3: F.pad + F.conv3d([8,1,L,L,L], stride 2) + split (reference
// src/ptwt/conv_transform_3.py:121-141) — separably, reading the volume once and writing the eight sub-band volumes
// once (the composed route of mifwt_compose.hip moves 2x the algorithmic bytes through scratch).
//
// A 256-thread workgroup owns TD x TR x 64 coefficients (depth x rows x columns) of all eight bands:
//   1. the (2 TD + L - 2) x (2 TR + L - 2) x (128 + L - 2) input brick, boundary extension as index maps per slice /
//      row / column, requested in ONE burst and parked in LDS;
//   2. W pass in place: every brick row becomes 64 (lo, hi) pairs (one wave per row, DS operations in order);
//   3. H pass in place: every brick slice becomes TR x 64 x (aa, da, ad, dd) — one wave per slice, all of the slice
//      read into registers before the first overwrite;
//   4. D pass LDS -> registers -> global: lane = output column, 4 packed accumulators = 8 bands, coalesced 256-byte
//      stores per band, row and slice.
// LDS per workgroup = the brick (L = 4, TD = 2, TR = 4: 31 KB, 5 workgroups per CU); the row / slice halo
// ((2T + L - 2) / 2T per axis) is re-read through L2.  Envelope: f32, L in {2, 4, 6}; longer filters use the composed
// route (the brick would no longer allow two workgroups per CU).
// Algorithmic traffic: 4*B*D*H*W read + 8*4*B*Do*Ho*Wo written.
#include <type_traits>

#include "mifwt_stream.h"

namespace mifwt {

namespace {

constexpr int kTC3 = 64;
constexpr int kExtra3 = 2;  // leftover columns the last column tile of the slice-per-wave kernel can take along

template <int L>
struct Dwt3TileArgs {
  const float* x;
  float* out[8];  // band s: bit 2 = depth high, bit 1 = row high, bit 0 = column high
  int64_t xs_b, xs_d, xs_h;
  int64_t os_b[8], os_d[8], os_h[8];
  int D, H, W, Do, Ho, Wo;
  int tiles_c, tiles_r, tiles_d;
  int segd;                     // unused
  FastDiv div_c, div_r, div_d;  // slice-per-wave kernel: by tiles_c, tiles_r, tiles_d
  int nt;       // non-zero: non-temporal sub-band stores (MIFWT_OPT_NT_STORE)
  int k_limit;  // the brick kernels store columns < k_limit; dwt3_fwd_tail_kernel the few beyond (see launch3)
  int extra;    // slice-per-wave kernel: the last column tile also makes columns k_limit .. k_limit + extra - 1 (<= kExtra3)
  int mode;
  f2 tap[L];  // (dec_lo[m], dec_hi[m])
};

template <int L, int TD, int TR>
__global__ void __launch_bounds__(256, 2) dwt3_fwd_tile_kernel(const Dwt3TileArgs<L> a) {
  constexpr int ID = 2 * TD + L - 2, IR = 2 * TR + L - 2, IC = 2 * kTC3 + L - 2;
  constexpr int XP = (IC + 1) & ~1;   // row pitch (floats)
  constexpr int SP = IR * XP;         // slice pitch (floats); >= TR * 256 (the slice's H-pass image)
  constexpr int NQ = (IC + 63) / 64;
  constexpr int NROWS = ID * IR;
  constexpr int RPW = (NROWS + 3) / 4;
  static_assert(SP >= TR * 4 * kTC3, "the H-pass image of a slice must fit into the slice it replaces");
  static_assert(XP >= 2 * kTC3, "the W-pass image of a row must fit into the row it replaces");
  static_assert((TD * TR) % 4 == 0, "output (slice, row) pairs are dealt to four waves");
  extern __shared__ __attribute__((aligned(16))) float brick[];  // [ID][IR][XP]

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int tc = bid % a.tiles_c;
  bid /= a.tiles_c;
  const int tr = bid % a.tiles_r;
  bid /= a.tiles_r;
  const int td = bid % a.tiles_d;
  const int img = bid / a.tiles_d;
  const int k0 = tc * kTC3, j0 = tr * TR, z0 = td * TD;

  // ---- 1. input brick -> LDS ---------------------------------------------------------------------------------------------
  // buffer resource over one batch element (32-bit offsets; the host checks the volume fits)
  const uint32_t vol_bytes =
      (uint32_t)(((int64_t)(a.D - 1) * a.xs_d + (int64_t)(a.H - 1) * a.xs_h + a.W) * 4);
  const __amdgpu_buffer_rsrc_t xrsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x + (int64_t)img * a.xs_b), 0, vol_bytes, 0x00020000);
  constexpr uint32_t kOob = 0x80000000u;
  const int nc_need = 2 * (min(k0 + kTC3, a.k_limit) - k0) + L - 2;
  const int nr_need = 2 * (min(j0 + TR, a.Ho) - j0) + L - 2;
  const int nd_need = 2 * (min(z0 + TD, a.Do) - z0) + L - 2;
  const int c_first = 2 * k0 - (L - 2), r_first = 2 * j0 - (L - 2), d_first = 2 * z0 - (L - 2);
  // boundary extension = the branch-free single-fold map (Fold1, mifwt_stream.h; the launcher requires extents >= L)
  __builtin_assume(wave >= 0 && wave < 4);
  const bool zero_mode = a.mode == MIFWT_MODE_ZERO;
  Fold1 fold;
  fold.set(a.mode);
  uint32_t coff[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int c = lane + 64 * q;
    const int ci = c_first + c;
    const bool dead = c >= nc_need || (zero_mode && (unsigned)ci >= (unsigned)a.W);
    coff[q] = dead ? kOob : 4u * (uint32_t)fold(ci, a.W);
  }
  const uint32_t row_bytes = (uint32_t)a.xs_h * 4u, slice_bytes = (uint32_t)a.xs_d * 4u;
  // bricks whose slices and rows lie inside the volume need no per-row boundary maps (scalar-unit work per brick row)
  const bool inside = d_first >= 0 && d_first + ID <= a.D && r_first >= 0 && r_first + IR <= a.H;
  float v[RPW][NQ];
  if (inside) {
    const uint32_t base = (uint32_t)d_first * slice_bytes + (uint32_t)r_first * row_bytes;
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
      const int rid = min(wave + 4 * i, NROWS - 1);  // wave-uniform brick row id = d * IR + r
      const int d = rid / IR, r = rid - d * IR;
      const uint32_t soff = base + (uint32_t)d * slice_bytes + (uint32_t)r * row_bytes;
#pragma unroll
      for (int q = 0; q < NQ; ++q)
        v[i][q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrsrc, coff[q], soff, 0));
    }
  } else {
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
      const int rid = wave + 4 * i;
      const int d = rid / IR, r = rid - d * IR;
      const int di = d_first + d, ri = r_first + r;
      const bool on = rid < NROWS && d < nd_need && r < nr_need &&
                      !(zero_mode && ((unsigned)di >= (unsigned)a.D || (unsigned)ri >= (unsigned)a.H));
      const uint32_t soff = __builtin_amdgcn_readfirstlane(
          on ? (uint32_t)fold(di, a.D) * slice_bytes + (uint32_t)fold(ri, a.H) * row_bytes : 0u);
#pragma unroll
      for (int q = 0; q < NQ; ++q)
        v[i][q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrsrc, on ? coff[q] : kOob, soff, 0));
    }
  }
#pragma unroll
  for (int i = 0; i < RPW; ++i) {
    const int rid = wave + 4 * i;
    if (rid < NROWS) {
#pragma unroll
      for (int q = 0; q < NQ; ++q)
        if (lane + 64 * q < XP) brick[rid * XP + lane + 64 * q] = v[i][q];
    }
  }
  // no workgroup barrier: a brick row is staged and then filtered by the same wave (rows wave + 4 i), DS order suffices
  wave_lds_fence();

  // ---- 2. W pass, in place: brick row -> (lo, hi)[k] -----------------------------------------------------------------------
  // rows are wave-private (row id = wave mod 4), so a wave may overwrite a row as soon as ITS reads of that row are
  // issued; rows go in batches of WB (reads of a batch, then its writes) to keep the dependent read -> write chains few
  constexpr int WB = 5;
#pragma unroll
  for (int i0 = 0; i0 < RPW; i0 += WB) {
    f2 acc[WB];
#pragma unroll
    for (int ib = 0; ib < WB; ++ib) {
      const int rid = wave + 4 * (i0 + ib);
      if (i0 + ib < RPW && rid < NROWS) {
        const f2* row = reinterpret_cast<const f2*>(&brick[rid * XP + 2 * lane]);
#pragma unroll
        for (int p = 0; p < L / 2; ++p) {
          const f2 xx = row[p];  // tile columns 2k + 2p, 2k + 2p + 1  <->  taps L-1-2p, L-2-2p
          if (p == 0) {
            acc[ib] = pkmul_lo(a.tap[L - 1], xx);
          } else {
            pkfma_lo(acc[ib], a.tap[L - 1 - 2 * p], xx);
          }
          pkfma_hi(acc[ib], a.tap[L - 2 - 2 * p], xx);
        }
      }
    }
    wave_lds_fence();
#pragma unroll
    for (int ib = 0; ib < WB; ++ib) {
      const int rid = wave + 4 * (i0 + ib);
      if (i0 + ib < RPW && rid < NROWS) *reinterpret_cast<f2*>(&brick[rid * XP + 2 * lane]) = acc[ib];
    }
  }
  __syncthreads();

  // ---- 3. H pass, in place: slice d -> [j][k] x (aa, da, ad, dd) ---------------------------------------------------------
  for (int d = wave; d < ID; d += 4) {
    float* slice = &brick[d * SP];
    f2 rowv[IR];
#pragma unroll
    for (int r = 0; r < IR; ++r) rowv[r] = *reinterpret_cast<const f2*>(&slice[r * XP + 2 * lane]);
    wave_lds_fence();  // the whole slice is in registers before its image overwrites it
#pragma unroll
    for (int j = 0; j < TR; ++j) {
      f2 lo2, hi2;  // lo2 = (row-low, row-high) of the column-low value; hi2 of the column-high value
#pragma unroll
      for (int m = 0; m < L; ++m) {
        const f2 hv = rowv[2 * j + (L - 1) - m];
        if (m == 0) {
          lo2 = pkmul_lo(a.tap[0], hv);
          hi2 = pkmul_hi(a.tap[0], hv);
        } else {
          pkfma_lo(lo2, a.tap[m], hv);
          pkfma_hi(hi2, a.tap[m], hv);
        }
      }
      // components: .x = (H a, W a), .y = (H d, W a), .z = (H a, W d), .w = (H d, W d)
      *reinterpret_cast<f4*>(&slice[(j * kTC3 + lane) * 4]) = (f4){lo2.x, lo2.y, hi2.x, hi2.y};
    }
  }
  __syncthreads();

  // ---- 4. D pass + stores: (slice, row) pairs dealt to the waves; lane = output column ------------------------------------------
  // The scalar unit is shared by the whole CU and was this kernel's busiest resource (934 scalar instructions per wave, most of
  // them 64-bit address arithmetic per band and store): band bases are formed once per workgroup, the (slice, row) offset once
  // per pair and stride set (band 0 = approximation strides, bands 1..7 share the detail strides).
  const int k = k0 + lane;
  float* obase[8];
#pragma unroll
  for (int b = 0; b < 8; ++b) obase[b] = a.out[b] + (int64_t)img * a.os_b[b] + k;
#pragma unroll
  for (int i = 0; i < (TD * TR) / 4; ++i) {
    const int pr = wave * ((TD * TR) / 4) + i;
    const int dz = pr / TR, j = pr - dz * TR;
    f2 acc[4];  // acc[c] = (depth-low, depth-high) of component c
#pragma unroll
    for (int m = 0; m < L; ++m) {
      const f4 hv = *reinterpret_cast<const f4*>(&brick[(2 * dz + (L - 1) - m) * SP + (j * kTC3 + lane) * 4]);
      const f2 h01 = {hv.x, hv.y}, h23 = {hv.z, hv.w};
      if (m == 0) {
        acc[0] = pkmul_lo(a.tap[0], h01);
        acc[1] = pkmul_hi(a.tap[0], h01);
        acc[2] = pkmul_lo(a.tap[0], h23);
        acc[3] = pkmul_hi(a.tap[0], h23);
      } else {
        pkfma_lo(acc[0], a.tap[m], h01);
        pkfma_hi(acc[1], a.tap[m], h01);
        pkfma_lo(acc[2], a.tap[m], h23);
        pkfma_hi(acc[3], a.tap[m], h23);
      }
    }
    const int z = z0 + dz, y = j0 + j;
    if (z < a.Do && y < a.Ho && k < a.k_limit) {
      const int64_t off_a = (int64_t)z * a.os_d[0] + (int64_t)y * a.os_h[0];  // approximation strides
      const int64_t off_d = (int64_t)z * a.os_d[1] + (int64_t)y * a.os_h[1];  // detail strides (bands 1..7)
      // component c = (H bit = c & 1, W bit = c >> 1)  ->  band = 4 * depth + 2 * H + W
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int hw = 2 * (c & 1) + (c >> 1);
        if (a.nt) {
          __builtin_nontemporal_store(acc[c].x, &obase[hw][hw == 0 ? off_a : off_d]);
          __builtin_nontemporal_store(acc[c].y, &obase[4 + hw][off_d]);
        } else {
          obase[hw][hw == 0 ? off_a : off_d] = acc[c].x;
          obase[4 + hw][off_d] = acc[c].y;
        }
      }
    }
  }
}

// ---- slice-per-wave bricks (the default) ---------------------------------------------------------------------------------
// Same brick as dwt3_fwd_tile_kernel (TD x 4 x 64 coefficients, all eight bands), different dealing of the work: wave w
// takes input SLICES w, w + 4, ... whole.  Its rows are staged, filtered along W and then along H without leaving the
// wave — the W-pass result of row r, column k sits in lane k's registers, which is exactly what the H pass of column k
// needs — so the W-pass image is never written to LDS and read back, and the barrier between the two passes is gone
// (dwt3_fwd_tile_kernel deals ROWS to waves and pays both).  One barrier: all slices filtered -> D pass.
template <int L, int TD>
__global__ void __launch_bounds__(256, 2) dwt3_fwd_slice_kernel(const Dwt3TileArgs<L> a) {
  constexpr int TR = 4, HL = L - 2;
  constexpr int ID = 2 * TD + HL, IR = 2 * TR + HL, IC = 2 * (kTC3 + kExtra3) + HL;
  constexpr int XP = (IC + 1) & ~1;  // row pitch (floats)
  constexpr int SP = IR * XP;        // slice pitch (floats); >= the slice's (H, W) image, TR x (64 + kExtra3) columns x 4
  constexpr int NQ = (IC + 63) / 64;
  constexpr int SPW = (ID + 3) / 4;  // slices per wave
  constexpr int XIMG = TR * 4 * kTC3;  // where the (H, W) image of the extra columns starts: [TR][kExtra3] x (aa, da, ad, dd)
  static_assert(SP >= TR * 4 * (kTC3 + kExtra3), "the (H, W) image of a slice must fit into the slice it replaces");
  static_assert((TD * TR) % 4 == 0, "output (slice, row) pairs are dealt to four waves");
  extern __shared__ __attribute__((aligned(16))) float ring[];  // [ID][IR][XP]

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  __builtin_assume(wave >= 0 && wave < 4);
  uint32_t utc, utr, utd;
  const int img = (int)a.div_d.divmod(a.div_r.divmod(a.div_c.divmod((uint32_t)xcd_remap(blockIdx.x, gridDim.x), utc), utr), utd);
  const int k0 = (int)utc * kTC3, j0 = (int)utr * TR, z0 = (int)utd * TD;
  const int zb = min(a.Do, z0 + TD);
  const uint32_t vol_bytes = (uint32_t)(((int64_t)(a.D - 1) * a.xs_d + (int64_t)(a.H - 1) * a.xs_h + a.W) * 4);
  const __amdgpu_buffer_rsrc_t xrsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x + (int64_t)img * a.xs_b), 0, vol_bytes, 0x00020000);
  constexpr uint32_t kOob = 0x80000000u;
  const bool zero_mode = a.mode == MIFWT_MODE_ZERO;
  Fold1 fold;
  fold.set(a.mode);
  const uint32_t row_bytes = (uint32_t)a.xs_h * 4u, slice_bytes = (uint32_t)a.xs_d * 4u;
  // column and row maps are the same for every slice of the walk: once per workgroup
  // The last column tile takes the one or two leftover columns of a plane just over a multiple of 64 columns along (129 = 2 * 64 + 1
  // for 256^3 with db2): their input samples are two more columns of its brick, and a handful of lanes filter them beside
  // the 64 lane-columns.  (A separate kernel for them read every row tail as its own 64-byte transaction and scattered single floats:
  // 56 us for 1 / 129 of level 1 of config 3, a fifth of the level.)
  const int ex = (int)utc == a.tiles_c - 1 ? a.extra : 0;
  const int nc_need = 2 * (min(k0 + kTC3, a.k_limit) - k0 + ex) + HL;
  const int nr_need = 2 * (min(j0 + TR, a.Ho) - j0) + HL;
  const int c_first = 2 * k0 - HL, r_first = 2 * j0 - HL;
  uint32_t coff[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int c = lane + 64 * q, ci = c_first + c;
    const bool dead = c >= nc_need || (zero_mode && (unsigned)ci >= (unsigned)a.W);
    coff[q] = dead ? kOob : 4u * (uint32_t)fold(ci, a.W);
  }
  uint32_t roff[IR];
  uint32_t rdead = 0;
#pragma unroll
  for (int i = 0; i < IR; ++i) {
    const int ri = r_first + i;
    const bool dead = i >= nr_need || (zero_mode && (unsigned)ri >= (unsigned)a.H);
    roff[i] = dead ? 0u : (uint32_t)fold(ri, a.H) * row_bytes;
    rdead |= dead ? 1u << i : 0u;
  }

  // the IR rows of extended input slice e -> registers (nothing is requested for slices no stored output needs)
  auto request = [&](float (&v)[IR][NQ], int e, bool on) {
    const bool sdead = !on || e >= 2 * zb || (zero_mode && (unsigned)e >= (unsigned)a.D);  // slices past the brick's last real output: nothing
    const uint32_t sbase = __builtin_amdgcn_readfirstlane(sdead ? 0u : (uint32_t)fold(e, a.D) * slice_bytes);
    if (!sdead && rdead == 0) {
#pragma unroll
      for (int i = 0; i < IR; ++i)
#pragma unroll
        for (int q = 0; q < NQ; ++q)
          v[i][q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrsrc, coff[q], sbase + roff[i], 0));
    } else {
#pragma unroll
      for (int i = 0; i < IR; ++i) {
        const bool dead = sdead || ((rdead >> i) & 1u);
#pragma unroll
        for (int q = 0; q < NQ; ++q)
          v[i][q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrsrc, dead ? kOob : coff[q], sbase + roff[i], 0));
      }
    }
  };
  // park the raw rows of a slice in ring slot `slot`
  auto stage = [&](const float (&v)[IR][NQ], int slot) {
    float* sl = &ring[slot * SP];
#pragma unroll
    for (int i = 0; i < IR; ++i)
#pragma unroll
      for (int q = 0; q < NQ; ++q)
        if (64 * q + 63 < XP || lane + 64 * q < XP) sl[i * XP + lane + 64 * q] = v[i][q];
  };
  // W pass + H pass of the slice in ring slot `slot`, registers in between; the (H, W) image replaces the raw rows
  auto filter_slice = [&](int slot) {
    float* sl = &ring[slot * SP];
    wave_lds_fence();  // this wave staged the slice: its own DS order is all that is needed
    f2 rowv[IR];       // (W-low, W-high) of column k0 + lane, row i
#pragma unroll
    for (int i = 0; i < IR; ++i) {
      const f2* row = reinterpret_cast<const f2*>(&sl[i * XP + 2 * lane]);
#pragma unroll
      for (int p = 0; p < L / 2; ++p) {
        const f2 xx = row[p];
        if (p == 0) {
          rowv[i] = pkmul_lo(a.tap[L - 1], xx);
        } else {
          pkfma_lo(rowv[i], a.tap[L - 1 - 2 * p], xx);
        }
        pkfma_hi(rowv[i], a.tap[L - 2 - 2 * p], xx);
      }
    }
    // extra columns: lane = (output row j, extra column e) filters its own L rows along W, then along H
    f4 ximg = {0.f, 0.f, 0.f, 0.f};
    const int xj = min(lane >> 1, TR - 1), xe = lane & 1;
    if (ex) {
      f2 lo2, hi2;
#pragma unroll
      for (int m = 0; m < L; ++m) {
        const f2* row = reinterpret_cast<const f2*>(&sl[(2 * xj + (L - 1) - m) * XP + 2 * (kTC3 + xe)]);
        f2 hv;
#pragma unroll
        for (int p = 0; p < L / 2; ++p) {
          const f2 xx = row[p];
          if (p == 0) {
            hv = pkmul_lo(a.tap[L - 1], xx);
          } else {
            pkfma_lo(hv, a.tap[L - 1 - 2 * p], xx);
          }
          pkfma_hi(hv, a.tap[L - 2 - 2 * p], xx);
        }
        if (m == 0) {
          lo2 = pkmul_lo(a.tap[0], hv);
          hi2 = pkmul_hi(a.tap[0], hv);
        } else {
          pkfma_lo(lo2, a.tap[m], hv);
          pkfma_hi(hi2, a.tap[m], hv);
        }
      }
      ximg = (f4){lo2.x, lo2.y, hi2.x, hi2.y};
    }
    wave_lds_fence();  // every raw row has been read before the image overwrites the slot
    if (ex && lane < 2 * TR && xe < ex) *reinterpret_cast<f4*>(&sl[XIMG + (xj * kExtra3 + xe) * 4]) = ximg;
#pragma unroll
    for (int j = 0; j < TR; ++j) {
      f2 lo2, hi2;
#pragma unroll
      for (int m = 0; m < L; ++m) {
        const f2 hv = rowv[2 * j + (L - 1) - m];
        if (m == 0) {
          lo2 = pkmul_lo(a.tap[0], hv);
          hi2 = pkmul_hi(a.tap[0], hv);
        } else {
          pkfma_lo(lo2, a.tap[m], hv);
          pkfma_hi(hi2, a.tap[m], hv);
        }
      }
      // components: .x = (H a, W a), .y = (H d, W a), .z = (H a, W d), .w = (H d, W d)
      *reinterpret_cast<f4*>(&sl[(j * kTC3 + lane) * 4]) = (f4){lo2.x, lo2.y, hi2.x, hi2.y};
    }
  };

  const int k = k0 + lane;
  float* obase[8];  // wave-uniform; lanes add 32-bit element offsets
#pragma unroll
  for (int b = 0; b < 8; ++b) obase[b] = a.out[b] + (int64_t)img * a.os_b[b];

  // every load of the wave's slices is in flight before the first slice is staged
  float v[SPW][IR][NQ];
#pragma unroll
  for (int t = 0; t < SPW; ++t) request(v[t], 2 * z0 - HL + wave + 4 * t, wave + 4 * t < ID);
#pragma unroll
  for (int t = 0; t < SPW; ++t) {
    if (4 * t + 3 < ID || wave + 4 * t < ID) {
      stage(v[t], wave + 4 * t);
      filter_slice(wave + 4 * t);
    }
  }
  __syncthreads();

  // D pass + stores: (output slice, row) pairs dealt to the waves, lane = output column (second call: lane = extra column)
  auto dpass = [&](auto xtag) {
    constexpr bool kX = decltype(xtag)::value;
    const int kk = kX ? a.k_limit + (lane & 1) : k;
    const bool on = kX ? lane < ex : k < a.k_limit;
#pragma unroll
    for (int i = 0; i < (TD * TR) / 4; ++i) {
      const int pr = wave * ((TD * TR) / 4) + i, dz = pr / TR, j = pr - dz * TR;
      const int ioff = kX ? XIMG + (j * kExtra3 + (lane & 1)) * 4 : (j * kTC3 + lane) * 4;
      f2 acc[4];  // acc[c] = (depth-low, depth-high) of component c
#pragma unroll
      for (int m = 0; m < L; ++m) {
        const f4 hv = *reinterpret_cast<const f4*>(&ring[(2 * dz + (L - 1) - m) * SP + ioff]);
        const f2 h01 = {hv.x, hv.y}, h23 = {hv.z, hv.w};
        if (m == 0) {
          acc[0] = pkmul_lo(a.tap[0], h01);
          acc[1] = pkmul_hi(a.tap[0], h01);
          acc[2] = pkmul_lo(a.tap[0], h23);
          acc[3] = pkmul_hi(a.tap[0], h23);
        } else {
          pkfma_lo(acc[0], a.tap[m], h01);
          pkfma_hi(acc[1], a.tap[m], h01);
          pkfma_lo(acc[2], a.tap[m], h23);
          pkfma_hi(acc[3], a.tap[m], h23);
        }
      }
      const int zo = z0 + dz, y = j0 + j;
      if (zo < zb && y < a.Ho && on) {
        const int off_a = zo * (int)a.os_d[0] + y * (int)a.os_h[0] + kk;  // approximation strides
        const int off_d = zo * (int)a.os_d[1] + y * (int)a.os_h[1] + kk;  // detail strides (bands 1..7)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int hw = 2 * (c & 1) + (c >> 1);  // component c = (H bit = c & 1, W bit = c >> 1) -> band = 4 depth + 2 H + W
          if (a.nt) {
            __builtin_nontemporal_store(acc[c].x, &obase[hw][hw == 0 ? off_a : off_d]);
            __builtin_nontemporal_store(acc[c].y, &obase[4 + hw][off_d]);
          } else {
            obase[hw][hw == 0 ? off_a : off_d] = acc[c].x;
            obase[4 + hw][off_d] = acc[c].y;
          }
        }
      }
    }
  };
  dpass(std::false_type{});
  if (ex) dpass(std::true_type{});
}

// The last few columns of a plane whose width is just over a multiple of 64 (129 = 2 * 64 + 1 for 256^3 with db2)
// would cost a whole extra column of bricks with one active lane in 64.  They go to this kernel instead: a workgroup owns
// TZ output slices x TY output rows of the `rem` leftover columns and runs the three passes through LDS with work items
// instead of lanes-as-columns:
//   1. W pass: item = (input slice, input row): reads the row's last few samples (one cache line) -> (lo, hi) per column
//   2. H pass: item = (input slice, output row, column) -> the four (H, W) components
//   3. D pass: item = (output slice, output row, column) -> eight bands, stored.
// (The first version was one thread per output position with L^3 mapped loads: 16 lines per position instead of the
// row tails once — 77 us for the one leftover column of config 3, a fifth of the level.)
constexpr int kTailTZ = 4, kTailTY = 32;

template <int L>
__global__ void __launch_bounds__(256) dwt3_fwd_tail_kernel(const Dwt3TileArgs<L> a, const int k_begin, const int rem, const int nzb,
                                                            const int nyb) {
  constexpr int HL = L - 2, TZ = kTailTZ, TY = kTailTY;
  constexpr int ID = 2 * TZ + HL, IRY = 2 * TY + HL;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  f2* const W1 = reinterpret_cast<f2*>(sm);                      // [ID][IRY][rem]: (W-low, W-high)
  f4* const H1 = reinterpret_cast<f4*>(sm + 2 * ID * IRY * rem);  // [ID][TY][rem]: (Ha Wa, Hd Wa, Ha Wd, Hd Wd)
  const int tid = threadIdx.x;
  int bid = blockIdx.x;
  const int yb = bid % nyb;
  bid /= nyb;
  const int zb = bid % nzb;
  const int img = bid / nzb;
  const int z0 = zb * TZ, y0 = yb * TY;
  const int d_first = 2 * z0 - HL, r_first = 2 * y0 - HL;
  const bool zero_mode = a.mode == MIFWT_MODE_ZERO;
  Fold1 fold;
  fold.set(a.mode);
  const float* xb = a.x + (int64_t)img * a.xs_b;

  // ---- 1. W pass ---------------------------------------------------------------------------------------------------------
  for (int item = tid; item < ID * IRY; item += 256) {
    const int s = item / IRY, r = item - s * IRY;
    const int ed = d_first + s, eh = r_first + r;
    // rows / slices no stored output needs (ragged last blocks) and implicit zeros read nothing
    const bool dead = ed >= 2 * a.Do || eh >= 2 * a.Ho || (zero_mode && ((unsigned)ed >= (unsigned)a.D || (unsigned)eh >= (unsigned)a.H));
    const float* row = xb + (dead ? 0 : (int64_t)fold(ed, a.D) * a.xs_d + (int64_t)fold(eh, a.H) * a.xs_h);
    for (int e = 0; e < rem; ++e) {
      const int k = k_begin + e;
      float lo = 0.f, hi = 0.f;
#pragma unroll
      for (int p = 0; p < L; ++p) {
        const int m = L - 1 - p, ew = 2 * k + 1 - m;  // same order as the bricks' W pass
        const bool zw = dead || (zero_mode && (unsigned)ew >= (unsigned)a.W);
        const float xv = zw ? 0.f : row[fold(ew, a.W)];
        lo = p == 0 ? a.tap[m].x * xv : __builtin_fmaf(a.tap[m].x, xv, lo);
        hi = p == 0 ? a.tap[m].y * xv : __builtin_fmaf(a.tap[m].y, xv, hi);
      }
      W1[item * rem + e] = (f2){lo, hi};
    }
  }
  __syncthreads();

  // ---- 2. H pass ---------------------------------------------------------------------------------------------------------
  for (int item = tid; item < ID * TY * rem; item += 256) {
    const int e = item % rem, sj = item / rem;
    const int s = sj / TY, j = sj - s * TY;
    f2 lo2, hi2;  // lo2 = (H-low, H-high) of the W-low value, hi2 of the W-high value
#pragma unroll
    for (int m = 0; m < L; ++m) {
      const f2 hv = W1[(s * IRY + 2 * j + (L - 1) - m) * rem + e];
      if (m == 0) {
        lo2 = (f2){a.tap[0].x * hv.x, a.tap[0].y * hv.x};
        hi2 = (f2){a.tap[0].x * hv.y, a.tap[0].y * hv.y};
      } else {
        lo2 = (f2){__builtin_fmaf(a.tap[m].x, hv.x, lo2.x), __builtin_fmaf(a.tap[m].y, hv.x, lo2.y)};
        hi2 = (f2){__builtin_fmaf(a.tap[m].x, hv.y, hi2.x), __builtin_fmaf(a.tap[m].y, hv.y, hi2.y)};
      }
    }
    H1[item] = (f4){lo2.x, lo2.y, hi2.x, hi2.y};  // item == (s * TY + j) * rem + e
  }
  __syncthreads();

  // ---- 3. D pass + stores ------------------------------------------------------------------------------------------------
  for (int item = tid; item < TZ * TY * rem; item += 256) {
    const int e = item % rem, zj = item / rem;
    const int dz = zj / TY, j = zj - dz * TY;
    float lo[4], hi[4];  // depth-low / depth-high of component c
#pragma unroll
    for (int m = 0; m < L; ++m) {
      const f4 hv = H1[((2 * dz + (L - 1) - m) * TY + j) * rem + e];
      const float c4[4] = {hv.x, hv.y, hv.z, hv.w};
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        lo[c] = m == 0 ? a.tap[0].x * c4[c] : __builtin_fmaf(a.tap[m].x, c4[c], lo[c]);
        hi[c] = m == 0 ? a.tap[0].y * c4[c] : __builtin_fmaf(a.tap[m].y, c4[c], hi[c]);
      }
    }
    const int z = z0 + dz, y = y0 + j, k = k_begin + e;
    if (z < a.Do && y < a.Ho) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int hw = 2 * (c & 1) + (c >> 1);  // component c = (H bit = c & 1, W bit = c >> 1) -> band = 4 depth + 2 H + W
        a.out[hw][(int64_t)img * a.os_b[hw] + (int64_t)z * a.os_d[hw] + (int64_t)y * a.os_h[hw] + k] = lo[c];
        a.out[4 + hw][(int64_t)img * a.os_b[4 + hw] + (int64_t)z * a.os_d[4 + hw] + (int64_t)y * a.os_h[4 + hw] + k] = hi[c];
      }
    }
  }
}

template <int L, int TD, int TR>
int launch3(const mifwt_level_desc* d, const void* x, void* approx, void* const* details, const double* lo,
            const double* hi, hipStream_t stream) {
  // TD < 0 selects the slice-per-wave kernel with bricks of -TD output slices
  constexpr bool kRoll = TD < 0;
  constexpr int TDA = TD < 0 ? -TD : TD;
  constexpr int ID = 2 * TDA + L - 2, IR = 2 * TR + L - 2, XP = (2 * (kTC3 + (kRoll ? kExtra3 : 0)) + L - 2 + 1) & ~1;
  constexpr size_t lds_bytes = (size_t)ID * IR * XP * sizeof(float);
  Dwt3TileArgs<L> a;
  a.nt = g_options[MIFWT_OPT_NT_STORE];
  a.x = static_cast<const float*>(x);
  for (int s = 0; s < 8; ++s) {
    a.out[s] = static_cast<float*>(s == 0 ? approx : details[s - 1]);
    const int64_t* st = s == 0 ? d->approx_stride : d->detail_stride;
    a.os_b[s] = st[0];
    a.os_d[s] = st[1];
    a.os_h[s] = st[2];
  }
  a.xs_b = d->sig_stride[0];
  a.xs_d = d->sig_stride[1];
  a.xs_h = d->sig_stride[2];
  a.D = (int)d->sig_extent[0];
  a.H = (int)d->sig_extent[1];
  a.W = (int)d->sig_extent[2];
  a.Do = (int)d->coef_extent[0];
  a.Ho = (int)d->coef_extent[1];
  a.Wo = (int)d->coef_extent[2];
  a.mode = d->mode;
  for (int m = 0; m < L; ++m) a.tap[m] = (f2){(float)lo[m], (float)hi[m]};
  // columns: whole bricks of 64; a remainder of at most 8 columns behind at least one full brick goes to the direct
  // kernel (a brick column for it would run with <= 8 of 64 lanes)
  const int rem = a.Wo % kTC3;
  const bool behind = a.Wo > kTC3 && rem > 0 && rem <= 8;   // a few columns behind at least one full brick
  const bool along = behind && kRoll && rem <= kExtra3;     // the last column tile of the slice-per-wave kernel takes them along
  const bool split = behind && !along;                      // ... or they go to the tail kernel
  a.extra = along ? rem : 0;
  a.k_limit = behind ? a.Wo - rem : a.Wo;
  a.tiles_c = (a.k_limit + kTC3 - 1) / kTC3;
  a.tiles_r = (a.Ho + TR - 1) / TR;
  a.segd = 0;
  a.tiles_d = (a.Do + TDA - 1) / TDA;
  a.div_c = make_fastdiv((uint32_t)a.tiles_c);
  a.div_r = make_fastdiv((uint32_t)a.tiles_r);
  a.div_d = make_fastdiv((uint32_t)a.tiles_d);
  const int64_t ntiles = (int64_t)d->batch * a.tiles_c * a.tiles_r * a.tiles_d;
  if (ntiles > INT32_MAX / 8) return MIFWT_ERR_UNSUPPORTED;
  if (split) {
    constexpr int ID = 2 * kTailTZ + L - 2, IRY = 2 * kTailTY + L - 2;
    const int nzb = (a.Do + kTailTZ - 1) / kTailTZ, nyb = (a.Ho + kTailTY - 1) / kTailTY;
    const size_t tail_lds = (size_t)(2 * ID * IRY + 4 * ID * kTailTY) * rem * sizeof(float);
    static DynLdsOnce tail_once;
    if (!tail_once.ensure(reinterpret_cast<const void*>(&dwt3_fwd_tail_kernel<L>), (int)((size_t)(2 * ID * IRY + 4 * ID * kTailTY) * 8 * sizeof(float))))
      return MIFWT_ERR_LAUNCH;
    const int64_t nblk = (int64_t)d->batch * nzb * nyb;
    if (nblk > INT32_MAX / 8) return MIFWT_ERR_UNSUPPORTED;
    hipLaunchKernelGGL((dwt3_fwd_tail_kernel<L>), dim3((unsigned)nblk), dim3(256), tail_lds, stream, a, a.k_limit, rem, nzb, nyb);
    if (hipGetLastError() != hipSuccess) return MIFWT_ERR_LAUNCH;
  }
  static DynLdsOnce lds_once;
  if constexpr (kRoll) {
    static_assert(TR == 4, "the slice-per-wave kernel owns 4 rows");
    if (!lds_once.ensure(reinterpret_cast<const void*>(&dwt3_fwd_slice_kernel<L, TDA>), (int)lds_bytes)) return MIFWT_ERR_LAUNCH;
    hipLaunchKernelGGL((dwt3_fwd_slice_kernel<L, TDA>), dim3((unsigned)ntiles), dim3(256), lds_bytes, stream, a);
  } else {
    if (!lds_once.ensure(reinterpret_cast<const void*>(&dwt3_fwd_tile_kernel<L, TD, TR>), (int)lds_bytes)) return MIFWT_ERR_LAUNCH;
    hipLaunchKernelGGL((dwt3_fwd_tile_kernel<L, TD, TR>), dim3((unsigned)ntiles), dim3(256), lds_bytes, stream, a);
  }
  return hipGetLastError() == hipSuccess ? MIFWT_OK : MIFWT_ERR_LAUNCH;
}

}  // namespace

bool dwt3_fwd_tile_supported(const mifwt_level_desc* d) {
  if (d->ndim != 3 || d->dtype != MIFWT_F32) return false;
  const int L = d->filt_len;
  if (L != 2 && L != 4 && L != 6) return false;
  if (d->sig_stride[3] != 1 || d->approx_stride[3] != 1 || d->detail_stride[3] != 1) return false;
  for (int i = 0; i < 3; ++i)
    if (d->sig_stride[i] < 0 || d->approx_stride[i] < 0 || d->detail_stride[i] < 0) return false;
  // one batch element must be addressable with 32-bit byte offsets below 2^31 (buffer-resource loads)
  const int64_t span = (d->sig_extent[0] - 1) * d->sig_stride[1] + (d->sig_extent[1] - 1) * d->sig_stride[2] + d->sig_extent[2];
  // single-fold boundary map: every extent at least as long as the filter
  for (int i = 0; i < 3; ++i)
    if (d->sig_extent[i] < L) return false;
  // 32-bit element offsets inside one batch element of a band (rolling kernel)
  if (d->coef_extent[0] * d->approx_stride[1] >= (int64_t(1) << 31) || d->coef_extent[0] * d->detail_stride[1] >= (int64_t(1) << 31))
    return false;
  return span < (int64_t(1) << 29);
}

int dwt3_fwd_tile(const mifwt_level_desc* d, const void* x, void* approx, void* const* details, const double* lo,
                  const double* hi, hipStream_t stream) {
  switch (d->filt_len) {
    // brick = 2 slices x 4 rows x 64 columns of coefficients: measured best on 8 x 256^3 db2 (0.38 ms per level-1 call vs
    // 0.46 for 4 x 4 x 64, 0.39 for 4 x 2 x 64, 0.44 for 2 x 2 x 64, 0.53 for 2 x 8 x 64; composed route 0.48)
    // TILE_ROWS option: 0 = slice-per-wave bricks 2 x 4 x 64 (default), 3 = slice-per-wave 3 x 4 x 64; row-dealt bricks: 2 = 2 x 4 x 64, 1 = 4 x 4 x 64
    case 2: return g_options[MIFWT_OPT_TILE_ROWS] == 1   ? launch3<2, 4, 4>(d, x, approx, details, lo, hi, stream)
                   : g_options[MIFWT_OPT_TILE_ROWS] == 2 ? launch3<2, 2, 4>(d, x, approx, details, lo, hi, stream)
                                                         : (g_options[MIFWT_OPT_TILE_ROWS] == 3 ? launch3<2, -3, 4>(d, x, approx, details, lo, hi, stream)
                                                                                             : launch3<2, -2, 4>(d, x, approx, details, lo, hi, stream));
    case 4: return g_options[MIFWT_OPT_TILE_ROWS] == 1   ? launch3<4, 4, 4>(d, x, approx, details, lo, hi, stream)
                   : g_options[MIFWT_OPT_TILE_ROWS] == 2 ? launch3<4, 2, 4>(d, x, approx, details, lo, hi, stream)
                                                         : (g_options[MIFWT_OPT_TILE_ROWS] == 3 ? launch3<4, -3, 4>(d, x, approx, details, lo, hi, stream)
                                                                                             : launch3<4, -2, 4>(d, x, approx, details, lo, hi, stream));
    case 6: return g_options[MIFWT_OPT_TILE_ROWS] == 1   ? launch3<6, 4, 4>(d, x, approx, details, lo, hi, stream)
                   : g_options[MIFWT_OPT_TILE_ROWS] == 2 ? launch3<6, 2, 4>(d, x, approx, details, lo, hi, stream)
                                                         : (g_options[MIFWT_OPT_TILE_ROWS] == 3 ? launch3<6, -3, 4>(d, x, approx, details, lo, hi, stream)
                                                                                             : launch3<6, -2, 4>(d, x, approx, details, lo, hi, stream));
    default: return MIFWT_ERR_UNSUPPORTED;
  }
}

}  // namespace mifwt
