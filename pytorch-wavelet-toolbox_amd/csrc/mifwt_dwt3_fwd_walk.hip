// mifwt_dwt3_fwd_walk.hip — fully fused 3-D analysis level, workgroups WALK along the depth axis (gfx950), kernel id 24.
//
// Replaces, for one level of wavedec3 / fswavedec3: F.pad + F.conv3d([8,1,L,L,L], stride 2) + split (reference
// src/ptwt/conv_transform_3.py:121-141) — separably, reading the volume once and writing the eight sub-band volumes once.
//
// The brick kernel (mifwt_dwt3_fwd_tile.hip, id 9) stages (2 TD + L - 2) (2 TR + L - 2) (128 + L - 2) samples per 2 TD x 2 TR x 128
// it consumes — 1.9 x for db2 — in phases (request, wait, three passes, store) that only five workgroups per CU overlap.  Here the
// depth halo is gone and nothing happens in phases:
//   * a workgroup owns TR output rows x EVERY output column of one depth segment of one volume and walks down the segment's input
//     slices; the only halo is L - 2 rows per 2 TR (re-read through L2 by the row group next door, which runs beside it on the same XCD);
//   * one LOADER wave requests the 2 TR + L - 2 rows of a slice as LDS-DMA (buffer_load_dwordx4 ... lds: a whole 256-sample row per
//     instruction, non-temporal), a ring of slices ahead of the compute waves; its vmcnt queue holds loads only, the compute waves'
//     queues hold stores only and are never waited on (the scheme of the 2-D kernel 16); one s_barrier per slice;
//   * a COMPUTE wave owns a strip of columns (lane = output column): W pass from the staged rows (8-byte LDS reads), H pass in
//     registers, D pass as rolling accumulators — the L/2 output slices in flight x 8 bands x TR rows live in registers, every
//     filtered slice is fed to them and forgotten; an output slice leaves every second step as 8 TR coalesced row pieces;
//   * boundary extension: slices and rows through the index map of the request (zero mode: requests through an empty resource land
//     zeros), pad columns filled in LDS by the waves that read them;  every mode, periodic included (a level needs nothing but
//     index maps), even L <= 10.
// f32 and f64 (template parameter T; f64: rows of at most 256 samples — a staged row is one or two 1-KiB requests either way — and two
// v_fma_f64 where f32 has one packed FMA).
// Algorithmic traffic: sizeof(T) (B D H W read + 8 B Do Ho Wo written).
#include "mifwt_pyr.h"

namespace mifwt {

namespace {

constexpr int kW3Pad = 8;        // samples behind a staged row's body (the right pad samples: at most L - 1)
// samples in front of it (the L - 2 left pad samples; the body stays 16-byte aligned).  Four for filters up to six taps: five staged slices
// of ten 256-sample rows are then 53 600 bytes, and THREE workgroups fit a CU's 160 KB (with eight: 54 400, two)
constexpr int walk3_lpad(int L) { return L <= 6 ? 4 : 8; }
constexpr int kW3MaxStrips = 4;  // compute waves per workgroup

template <typename T, int L>
struct Walk3Args {
  const T* x;
  T* out[8];  // band s: bit 2 = depth high, bit 1 = row high, bit 0 = column high
  int64_t xs_b, os_b[2];  // batch strides: input; [0] approximation, [1] details
  uint32_t xs_d, xs_h;
  uint32_t os_d[2], os_h[2];
  int D, H, W, Do, Ho, Wo;
  int nstrips, nq;    // compute waves, output columns per wave
  int ngroups;        // row groups of TR output rows
  int nseg, seg_out;  // depth segments, output slices per segment
  int nslots;         // staged slices
  int mode, nt, dbg;
  FastDiv div_g, div_s;
  typename TileArith<T>::vec2 tap[L];
};

template <int N>
__device__ __forceinline__ void walk3_wait(int later) {  // at most `later` slices of N requests each may still be in flight
  if (later >= 6) pyr_wait_vm<(6 * N > 63 ? 63 : 6 * N)>();
  else if (later == 5) pyr_wait_vm<(5 * N > 63 ? 63 : 5 * N)>();
  else if (later == 4) pyr_wait_vm<(4 * N > 63 ? 63 : 4 * N)>();
  else if (later == 3) pyr_wait_vm<(3 * N > 63 ? 63 : 3 * N)>();
  else if (later == 2) pyr_wait_vm<(2 * N > 63 ? 63 : 2 * N)>();
  else if (later == 1) pyr_wait_vm<(N > 63 ? 63 : N)>();
  else pyr_wait_vm<0>();
}

// one staged row: NCH requests of 1 KiB; NT: non-temporal (a row no other workgroup asks for), else default policy (halo rows: the
// row group next door requests them too, about now)
template <int NCH, bool NT>
__device__ __forceinline__ void walk3_dma_row(const uint32_t (&voff)[3], rsrc_t rsrc, uint32_t soff, uint32_t lds0) {
  if constexpr (NT) {
    pyr_dma_row<NCH, 0x400>(voff, rsrc, soff, lds0);
  } else {
    uint32_t keep;
    if constexpr (NCH == 1) {
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(voff[0]), "s"(rsrc), "s"(soff), "s"(lds0) : "memory");
    } else {
      static_assert(NCH == 2, "rows of at most 512 samples");
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %5\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, %4 offen lds\n\t"
                   "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(voff[0]), "v"(voff[1]), "s"(rsrc), "s"(soff), "s"(lds0) : "memory", "scc");
    }
  }
}

// a staged row: BODY samples of ES bytes (f32: 128, 160, 256 or 512; f64: 128 or 256 — one request of BODY ES / 16 lanes, or two of
// 1 KiB) between two pads
constexpr int walk3_pitch(int L, int BODY, int ES) { return (walk3_lpad(L) + BODY + kW3Pad) * ES; }
constexpr int walk3_nreq(int BODY, int ES) { return BODY * ES > 1024 ? 2 : 1; }
// loader waves: the requests of four slices ahead must fit a wave's vmcnt counter (63)
constexpr int walk3_nload(int IRW, int BODY, int ES) { return IRW * walk3_nreq(BODY, ES) * 4 > 63 ? 2 : 1; }
constexpr int kW3MaxWaves = 10;  // compute waves (column strips x row sub-groups, at most 8) + loaders

// (f64: one row sub-group, so at most 4 + 2 waves — and the 256 VGPRs that leaves a wave hold the accumulators without scratch)
template <typename T> constexpr int walk3_max_waves() { return sizeof(T) == 8 ? 6 : kW3MaxWaves; }

template <typename T, int L, int TR, int BODY, int NRG>
__global__ void __launch_bounds__(64 * walk3_max_waves<T>()) dwt3_fwd_walk_kernel(const Walk3Args<T, L> a) {
  using V2 = typename TileArith<T>::vec2;
  constexpr int ES = (int)sizeof(T);
  // a workgroup owns NRG row sub-groups of TR output rows: compute wave w filters strip w % nstrips of sub-group w / nstrips
  constexpr int HL = L - 2, HP = L / 2, IR = 2 * TR + HL, NC = 2 * TR;
  constexpr int IRW = 2 * TR * NRG + HL;  // staged rows of a slice
  constexpr int PITCH = walk3_pitch(L, BODY, ES);
  constexpr int SLAB = IRW * PITCH;
  constexpr int NLOAD = walk3_nload(IRW, BODY, ES), NCHE = walk3_nreq(BODY, ES);
  static_assert(BODY * ES <= 2048 && PITCH % 16 == 0, "a staged row is at most two 1-KiB requests; rows stay 16-byte aligned");
  static_assert(IRW % NLOAD == 0, "the loaders take every NLOAD-th row: equal shares");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  uint32_t ug, us;
  const int img = __builtin_amdgcn_readfirstlane((int)a.div_s.divmod(a.div_g.divmod((uint32_t)xcd_remap(blockIdx.x, gridDim.x), ug), us));
  const int j0 = __builtin_amdgcn_readfirstlane((int)ug * (TR * NRG));
  const int zA = __builtin_amdgcn_readfirstlane((int)us * a.seg_out), zB = min(a.Do, zA + a.seg_out);
  const int E0 = 2 * zA - HL;            // first input slice (extended index) of the walk
  const int nsl = 2 * (zB - zA) + HL;    // slices = steps (even)
  const bool zero_mode = a.mode == MIFWT_MODE_ZERO;
  Fold1 fold;
  fold.set(a.mode);

  // =====================================================================================================================
  // loader waves: loader l requests the staged rows l, l + NLOAD, ...
  const int ncomp = a.nstrips * NRG;
  if (wave >= ncomp) {
    const int l = wave - ncomp;
    constexpr int NR = IRW / NLOAD;
    const uint32_t vol_bytes = (uint32_t)(((int64_t)(a.D - 1) * a.xs_d + (int64_t)(a.H - 1) * a.xs_h + a.W) * ES);
    const rsrc_t xr = pyr_rsrc(a.x + (int64_t)img * a.xs_b, vol_bytes);
    const rsrc_t xr_dead = pyr_rsrc(a.x + (int64_t)img * a.xs_b, 0);  // every lane out of range: zeros land
    const uint32_t row_bytes = a.xs_h * (uint32_t)ES, slice_bytes = a.xs_d * (uint32_t)ES;
    const int r_first = 2 * j0 - HL;
    const int nr_need = 2 * (min(j0 + TR * NRG, a.Ho) - j0) + HL;
    uint32_t roff[NR];
    uint32_t rdead = 0;
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      const int ri = r_first + l + NLOAD * i;
      const bool dead = l + NLOAD * i >= nr_need || (zero_mode && (unsigned)ri >= (unsigned)a.H);
      roff[i] = __builtin_amdgcn_readfirstlane(dead ? 0u : (uint32_t)fold(ri, a.H) * row_bytes);
      rdead |= dead ? 1u << i : 0u;
    }
    uint32_t voff[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int cb = 1024 * j + 16 * lane;  // byte offset of this lane's 16 bytes inside the row
      voff[j] = (j < NCHE && cb < a.W * ES) ? (uint32_t)cb : kPyrOob;
    }
    __builtin_amdgcn_s_setprio(3);
    constexpr int PER = NR * NCHE;
    int ib = 0;  // slot of the next slice to be requested
    auto issue = [&](int t) {
      const uint32_t buf = (uint32_t)ib * (uint32_t)SLAB + (uint32_t)(walk3_lpad(L) * ES);
      ib = ib + 1 == a.nslots ? 0 : ib + 1;
      if (MIFWT_DBG(a) & 2) return;
      const int e = E0 + t;
      const bool sdead = zero_mode && (unsigned)e >= (unsigned)a.D;
      const uint32_t sbase = sdead ? 0u : (uint32_t)fold(e, a.D) * slice_bytes;
#pragma unroll
      for (int i = 0; i < NR; ++i) {
        const bool dead = __builtin_amdgcn_readfirstlane((int)(sdead || ((rdead >> i) & 1u))) != 0;
        const uint32_t so = __builtin_amdgcn_readfirstlane(sbase + roff[i]), la = __builtin_amdgcn_readfirstlane(buf + (uint32_t)((l + NLOAD * i) * PITCH));
        // default cache policy: the first and last L - 2 rows are staged by the row groups next door as well, about now, and a non-temporal
        // request does not leave them in L2 for the neighbour (config 3, level 1: HBM reads 1.23 x the volume and 261 us, against 227 us)
        if constexpr (BODY * ES < 1024) {
          if (lane < BODY * ES / 16) walk3_dma_row<1, false>(voff, dead ? xr_dead : xr, so, la);  // masked lanes write nothing: BODY ES bytes land
        } else {
          if (MIFWT_DBG(a) & 16) walk3_dma_row<NCHE, true>(voff, dead ? xr_dead : xr, so, la);
          else walk3_dma_row<NCHE, false>(voff, dead ? xr_dead : xr, so, la);
        }
      }
    };
    const int ahead = a.nslots - 1;  // slices requested ahead (<= 4)
    for (int t = 0; t < ahead; ++t)
      if (t < nsl) issue(t);
#pragma unroll 1
    for (int t = 0; t < nsl; ++t) {
      walk3_wait<PER>(min(ahead - 1, nsl - 1 - t));  // slice t has landed
      __syncthreads();
      if (t + ahead < nsl) issue(t + ahead);  // into the slot slice t - 1 was read from
    }
    return;
  }

  // =====================================================================================================================
  // compute wave: output columns [k0, k1)
  const int sub = __builtin_amdgcn_readfirstlane(wave / a.nstrips), strip = wave - sub * a.nstrips;
  const int jw = j0 + sub * TR;  // first output row of this wave
  const int k0 = strip * a.nq, k1 = min(a.Wo, k0 + a.nq);
  const int k = k0 + lane;
  const bool active = k < k1;
  const int kk = min(k, a.Wo - 1);  // idle lanes filter the plane's last column (their windows stay inside the row)
  const bool first = strip == 0, last = strip == a.nstrips - 1;
  const int nrp = 2 * a.Wo - a.W;  // pad samples behind a row (0 .. L - 1)

  T* obase[8];
#pragma unroll
  for (int b = 0; b < 8; ++b) obase[b] = a.out[b] + (int64_t)img * a.os_b[b == 0 ? 0 : 1];

  PyrAcc<L, NC, V2> acc;
  acc.clear();
  int slot = 0;

  // one slice: pads, W pass, H pass -> hv[2 j] = (Ha Wa, Hd Wa), hv[2 j + 1] = (Ha Wd, Hd Wd) of row j
  auto filter_slice = [&](V2 (&hv)[NC]) {
    unsigned char* const sl = smem + slot * SLAB + sub * (2 * TR * PITCH);  // the wave's 2 TR + L - 2 rows
    slot = slot + 1 == a.nslots ? 0 : slot + 1;
    if constexpr (HL > 0) {
      if (first)
      for (int it = lane; it < IR * HL; it += 64) {
        const int r = it / HL, i = it - r * HL;
        T* row = reinterpret_cast<T*>(sl + r * PITCH) + walk3_lpad(L);
        row[i - HL] = zero_mode ? T(0) : row[fold(i - HL, a.W)];
      }
    }
    if (last && nrp > 0) {
      for (int it = lane; it < IR * nrp; it += 64) {
        const int r = it / nrp, i = it - r * nrp;
        T* row = reinterpret_cast<T*>(sl + r * PITCH) + walk3_lpad(L);
        row[a.W + i] = zero_mode ? T(0) : row[fold(a.W + i, a.W)];
      }
    }
    wave_lds_fence();
    if (MIFWT_DBG(a) & 4) {
#pragma unroll
      for (int c = 0; c < NC; ++c) hv[c] = (V2){T(1), T(2)};
      return;
    }
    V2 rowv[IR];  // (W-low, W-high) of column k, row i
    const unsigned char* wb = sl + (walk3_lpad(L) + 2 * kk - HL) * ES;
#pragma unroll
    for (int i = 0; i < IR; ++i) {
      const V2* row = reinterpret_cast<const V2*>(wb + i * PITCH);
#pragma unroll
      for (int p = 0; p < HP; ++p) {
        const V2 xx = row[p];  // samples 2 k - HL + 2 p, + 1  <->  taps L - 1 - 2 p, L - 2 - 2 p
        if (p == 0) rowv[i] = vmul_lo(a.tap[L - 1], xx);
        else vfma_lo(rowv[i], a.tap[L - 1 - 2 * p], xx);
        vfma_hi(rowv[i], a.tap[L - 2 - 2 * p], xx);
      }
    }
#pragma unroll
    for (int j = 0; j < TR; ++j) {
#pragma unroll
      for (int m = 0; m < L; ++m) {
        const V2 v = rowv[2 * j + (L - 1) - m];
        if (m == 0) {
          hv[2 * j] = vmul_lo(a.tap[0], v);
          hv[2 * j + 1] = vmul_hi(a.tap[0], v);
        } else {
          vfma_lo(hv[2 * j], a.tap[m], v);
          vfma_hi(hv[2 * j + 1], a.tap[m], v);
        }
      }
    }
  };

  // output slice z from accumulator slot SL.  (Measured and dropped: lanes l / l + 32 owning neighbouring columns and exchanging the rows of
  // a row pair for 8-byte stores — on 129-sample rows they are only 4-byte aligned: 235 against 229 us, profiles/r04w16_walk3_st8.txt)
  auto emit = [&](auto sl_tag, int z) {
    constexpr int SL = decltype(sl_tag)::value;
    if (MIFWT_DBG(a) & 1) return;
    const uint32_t za = (uint32_t)z * a.os_d[0] + (uint32_t)k, zd = (uint32_t)z * a.os_d[1] + (uint32_t)k;
#pragma unroll
    for (int j = 0; j < TR; ++j) {
      const int y = jw + j;
      if (y < a.Ho && active) {
        uint32_t oa = za + (uint32_t)y * a.os_h[0], od = zd + (uint32_t)y * a.os_h[1];
        if (MIFWT_DBG(a) & 8) {  // A/B (wrong results): rows on a 128-sample pitch, i.e. line-aligned 256-byte stores
          if (k >= 128) continue;
          oa = od = ((uint32_t)z * (uint32_t)a.Ho + (uint32_t)y) * 128u + (uint32_t)k;
        }
        // acc.lo[.][2 j] = (D a, D d) of Ha Wa; acc.hi[.][2 j] of Hd Wa; acc.lo[.][2 j + 1] of Ha Wd; acc.hi[.][2 j + 1] of Hd Wd
        if (a.nt) {
          __builtin_nontemporal_store(acc.lo[SL][2 * j].x, &obase[0][oa]);
          __builtin_nontemporal_store(acc.lo[SL][2 * j + 1].x, &obase[1][od]);
          __builtin_nontemporal_store(acc.hi[SL][2 * j].x, &obase[2][od]);
          __builtin_nontemporal_store(acc.hi[SL][2 * j + 1].x, &obase[3][od]);
          __builtin_nontemporal_store(acc.lo[SL][2 * j].y, &obase[4][od]);
          __builtin_nontemporal_store(acc.lo[SL][2 * j + 1].y, &obase[5][od]);
          __builtin_nontemporal_store(acc.hi[SL][2 * j].y, &obase[6][od]);
          __builtin_nontemporal_store(acc.hi[SL][2 * j + 1].y, &obase[7][od]);
        } else {
          obase[0][oa] = acc.lo[SL][2 * j].x;
          obase[1][od] = acc.lo[SL][2 * j + 1].x;
          obase[2][od] = acc.hi[SL][2 * j].x;
          obase[3][od] = acc.hi[SL][2 * j + 1].x;
          obase[4][od] = acc.lo[SL][2 * j].y;
          obase[5][od] = acc.lo[SL][2 * j + 1].y;
          obase[6][od] = acc.hi[SL][2 * j].y;
          obase[7][od] = acc.hi[SL][2 * j + 1].y;
        }
      }
    }
  };

  // pairs of slices; the pair index modulo L/2 is a compile-time constant inside the unrolled body
  for (int pb = 0; 2 * pb < nsl; pb += HP) {
    bool done = false;
    pyr_static_for<HP>([&](auto r_tag) {
      constexpr int R = decltype(r_tag)::value;
      const int p = pb + R;
      if (done || 2 * p >= nsl) {
        done = true;
        return;
      }
      V2 hv[NC];
      __syncthreads();
      filter_slice(hv);
      acc.template feed<0, R>(a.tap, hv);
      __syncthreads();
      filter_slice(hv);
      acc.template feed<1, R>(a.tap, hv);
      const int z = zA + p - (HP - 1);
      if (p >= HP - 1 && z < zB) emit(std::integral_constant<int, PyrAcc<L, NC, V2>::done(R)>{}, z);
    });
  }
}

template <typename T, int L, int TR, int BODY, int NRG>
int launch_walk3(const mifwt_level_desc* d, const void* x, void* approx, void* const* details, const double* lo, const double* hi,
                 hipStream_t stream) {
  constexpr int HL = L - 2, IRW = 2 * TR * NRG + HL;
  constexpr int ES = (int)sizeof(T);
  constexpr int PITCH = walk3_pitch(L, BODY, ES), SLAB = IRW * PITCH;
  constexpr int NLOAD = walk3_nload(IRW, BODY, ES), PER = IRW / NLOAD * walk3_nreq(BODY, ES);
  Walk3Args<T, L> a;
  a.x = static_cast<const T*>(x);
  for (int s = 0; s < 8; ++s) a.out[s] = static_cast<T*>(s == 0 ? approx : details[s - 1]);
  a.xs_b = d->sig_stride[0];
  a.xs_d = (uint32_t)d->sig_stride[1];
  a.xs_h = (uint32_t)d->sig_stride[2];
  a.os_b[0] = d->approx_stride[0];
  a.os_b[1] = d->detail_stride[0];
  a.os_d[0] = (uint32_t)d->approx_stride[1];
  a.os_d[1] = (uint32_t)d->detail_stride[1];
  a.os_h[0] = (uint32_t)d->approx_stride[2];
  a.os_h[1] = (uint32_t)d->detail_stride[2];
  a.D = (int)d->sig_extent[0];
  a.H = (int)d->sig_extent[1];
  a.W = (int)d->sig_extent[2];
  a.Do = (int)d->coef_extent[0];
  a.Ho = (int)d->coef_extent[1];
  a.Wo = (int)d->coef_extent[2];
  a.mode = d->mode;
  a.nt = g_options[MIFWT_OPT_NT_STORE];
  a.dbg = g_options[MIFWT_OPT_DEBUG] & 31;  // (bit 6: strips of 64 columns, below)
  for (int m = 0; m < L; ++m) a.tap[m] = (typename TileArith<T>::vec2){(T)lo[m], (T)hi[m]};
  a.nstrips = (a.Wo + 63) / 64;
  a.nq = (g_options[MIFWT_OPT_DEBUG] & 64) ? 64 : (a.Wo + a.nstrips - 1) / a.nstrips;
  if (const int ns = exp_word() & 15; ns >= a.nstrips && ns * NRG <= (ES == 8 ? 4 : 8)) {  // (A/B: narrower strips, more waves — 8 x 256^3 db2, 3 / 4 / 5 / 6 / 8 strips: 232 / 233 / 243 / 249 / 347 us)
    a.nstrips = ns;
    a.nq = (a.Wo + ns - 1) / ns;
  }
  a.ngroups = (a.Ho + TR * NRG - 1) / (TR * NRG);
  // staged slices: four ahead of the one being filtered (config 3 level 1: 2 / 3 / 4 / 5 / 6 ahead = 271 / 259 / 241 / 252 / 256 us)
  // (volumes below ~4 M samples are latency-bound: workgroups per CU beat depth — 8 x 129^3, 1 / 2 / 3 / 4 ahead: 37 / 34 / 50 / 46 us)
  const int64_t in_vol = d->sig_extent[0] * d->sig_extent[1] * d->sig_extent[2];
  int nslots = (L == 10 || in_vol < (int64_t(1) << 22) || ES == 8) ? 3 : 5;  // (f64: three slices of 2-KiB rows leave two workgroups per CU)
  if (g_options[MIFWT_OPT_PREFETCH_PAIRS] > 0) nslots = g_options[MIFWT_OPT_PREFETCH_PAIRS] + 1;
  if (nslots < 2) nslots = 2;
  if (nslots > 7) nslots = 7;
  while (nslots > 2 && (PER * (nslots - 1) > 63 || nslots * SLAB > 150 * 1024)) --nslots;
  a.nslots = nslots;
  const size_t lds_bytes = (size_t)nslots * SLAB;
  // depth segments: enough workgroups for ~3 per slot (workgroups per CU: by LDS), at least 6 output slices each
  int ncu = 256;
  {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ncu = v;
  }
  int wpc = (int)((size_t)(160 * 1024) / lds_bytes);
  if (wpc > 8) wpc = 8;
  if (wpc < 1) wpc = 1;
  const int64_t base = (int64_t)d->batch * a.ngroups;
  int nseg = (int)((3 * (int64_t)ncu * wpc + base - 1) / base);
  if (g_options[MIFWT_OPT_ROWS_PER_CHUNK] > 0) nseg = (a.Do + g_options[MIFWT_OPT_ROWS_PER_CHUNK] - 1) / g_options[MIFWT_OPT_ROWS_PER_CHUNK];
  if (nseg > a.Do / 6 && g_options[MIFWT_OPT_ROWS_PER_CHUNK] <= 0) nseg = a.Do / 6;
  if (nseg < 1) nseg = 1;
  a.seg_out = (a.Do + nseg - 1) / nseg;
  a.nseg = (a.Do + a.seg_out - 1) / a.seg_out;
  a.div_g = make_fastdiv((uint32_t)a.ngroups);
  a.div_s = make_fastdiv((uint32_t)a.nseg);
  const int64_t nblk = base * a.nseg;
  if (nblk > INT32_MAX / 8) return MIFWT_ERR_UNSUPPORTED;
  static DynLdsOnce lds_once;
  if (!lds_once.ensure(reinterpret_cast<const void*>(&dwt3_fwd_walk_kernel<T, L, TR, BODY, NRG>), 7 * SLAB > 160 * 1024 ? 160 * 1024 : 7 * SLAB))
    return MIFWT_ERR_LAUNCH;
  hipLaunchKernelGGL((dwt3_fwd_walk_kernel<T, L, TR, BODY, NRG>), dim3((unsigned)nblk), dim3(64 * (a.nstrips * NRG + NLOAD)), lds_bytes, stream, a);
  return hipGetLastError() == hipSuccess ? MIFWT_OK : MIFWT_ERR_LAUNCH;
}

template <int L>
int launch_walk3_l(const mifwt_level_desc* d, const void* x, void* approx, void* const* details, const double* lo, const double* hi,
                   hipStream_t stream) {
  constexpr int TR = L <= 6 ? 4 : 2;
  const int W = (int)d->sig_extent[2], nstrips = ((int)d->coef_extent[2] + 63) / 64;
  // row sub-groups per workgroup (several sub-groups share a staged slice: fewer halo rows, more compute waves per loader): pays for ten
  // taps only, where a lone sub-group stages 12 rows per 2 it completes (32 x 100^3 db5, 1 / 2 / 4 sub-groups: 174 / 164 / 187 us;
  // 16 x 128^3 db4: 129 / 147 / 146; 8 x 256^3 db2: 228 / 234; 8 x 66^3 db2: 20.7 / 21.4 / 25.3); MIFWT_OPT_PAIR_ROWS overrides (1, 2)
  int nrg = (L == 10 && nstrips <= 2) ? 2 : 1;
  if (g_options[MIFWT_OPT_PAIR_ROWS] > 0) nrg = g_options[MIFWT_OPT_PAIR_ROWS];
  const bool two = nrg >= 2 && L == 10 && nstrips <= 4;  // (instantiated for ten taps only)
  if (W <= 128) return two ? launch_walk3<float, L, TR, 128, (L == 10 ? 2 : 1)>(d, x, approx, details, lo, hi, stream) : launch_walk3<float, L, TR, 128, 1>(d, x, approx, details, lo, hi, stream);
  if (W <= 160) return two ? launch_walk3<float, L, TR, 160, (L == 10 ? 2 : 1)>(d, x, approx, details, lo, hi, stream) : launch_walk3<float, L, TR, 160, 1>(d, x, approx, details, lo, hi, stream);
  if (W <= 256) return two ? launch_walk3<float, L, TR, 256, (L == 10 ? 2 : 1)>(d, x, approx, details, lo, hi, stream) : launch_walk3<float, L, TR, 256, 1>(d, x, approx, details, lo, hi, stream);
  return launch_walk3<float, L, TR, 512, 1>(d, x, approx, details, lo, hi, stream);
}

// f64: rows of at most 256 samples (2 KiB = two requests), one row sub-group
template <int L>
int launch_walk3_l64(const mifwt_level_desc* d, const void* x, void* approx, void* const* details, const double* lo, const double* hi,
                     hipStream_t stream) {
  constexpr int TR = L <= 4 ? 4 : 2;  // (six taps with four rows: 255 VGPRs and scratch)
  if constexpr (L <= 4) {
    // two output rows per workgroup on small volumes (8 x 66^3 db2: 17.0 against 22.5 us; 129^3: 68.2 against 66.6; 256^3: 465-505 against 443)
    const int64_t vol = d->sig_extent[0] * d->sig_extent[1] * d->sig_extent[2];
    if (g_options[MIFWT_OPT_TILE_ROWS] == 2 || (g_options[MIFWT_OPT_TILE_ROWS] == 0 && vol < (int64_t(1) << 20))) {
      if (d->sig_extent[2] <= 128) return launch_walk3<double, L, 2, 128, 1>(d, x, approx, details, lo, hi, stream);
      return launch_walk3<double, L, 2, 256, 1>(d, x, approx, details, lo, hi, stream);
    }
  }
  if (d->sig_extent[2] <= 128) return launch_walk3<double, L, TR, 128, 1>(d, x, approx, details, lo, hi, stream);
  return launch_walk3<double, L, TR, 256, 1>(d, x, approx, details, lo, hi, stream);
}

}  // namespace

bool dwt3_fwd_walk_supported(const mifwt_level_desc* d) {
  if (d->ndim != 3 || (d->dtype != MIFWT_F32 && d->dtype != MIFWT_F64)) return false;
  const bool f64 = d->dtype == MIFWT_F64;
  const int L = d->filt_len;
  if (L < 2 || L > 10 || (L & 1)) return false;
  if (d->sig_stride[3] != 1 || d->approx_stride[3] != 1 || d->detail_stride[3] != 1) return false;
  for (int i = 0; i < 3; ++i)
    if (d->sig_stride[i] < 0 || d->approx_stride[i] < 0 || d->detail_stride[i] < 0) return false;
  // single-fold boundary map: every extent at least as long as the filter
  for (int i = 0; i < 3; ++i)
    if (d->sig_extent[i] < L) return false;
  // a row is one or two 1-KiB requests; at most four column strips of 64
  if (d->sig_extent[2] > (f64 ? 256 : 512) || d->coef_extent[2] > 64 * kW3MaxStrips) return false;
  // one batch element addressable with 32-bit byte offsets (buffer-resource requests), 32-bit element offsets inside a band
  const int64_t span = (d->sig_extent[0] - 1) * d->sig_stride[1] + (d->sig_extent[1] - 1) * d->sig_stride[2] + d->sig_extent[2];
  if (span >= (int64_t(1) << (f64 ? 28 : 29))) return false;
  if (d->coef_extent[0] * d->approx_stride[1] >= (int64_t(1) << 31) || d->coef_extent[0] * d->detail_stride[1] >= (int64_t(1) << 31))
    return false;
  return true;
}

int dwt3_fwd_walk(const mifwt_level_desc* d, const void* x, void* approx, void* const* details, const double* lo, const double* hi,
                  hipStream_t stream) {
  // eight / ten taps on rows of at most 128 samples: the slab form (mifwt_dwt3_fwd_slab.hip; MIFWT_OPT_DEBUG bit 21 keeps the strips)
  // where it is ahead (or wherever it can run, MIFWT_OPT_TILE_MODE 4)
  if (!(g_options[MIFWT_OPT_DEBUG] & 2097152) && (g_options[MIFWT_OPT_TILE_MODE] == 4 ? dwt3_fwd_slab_supported(d) : dwt3_fwd_slab_pays(d)))
    return dwt3_fwd_slab(d, x, approx, details, lo, hi, stream);
  if (d->dtype == MIFWT_F64) {
    switch (d->filt_len) {
      case 2: return launch_walk3_l64<2>(d, x, approx, details, lo, hi, stream);
      case 4: return launch_walk3_l64<4>(d, x, approx, details, lo, hi, stream);
      case 6: return launch_walk3_l64<6>(d, x, approx, details, lo, hi, stream);
      case 8: return launch_walk3_l64<8>(d, x, approx, details, lo, hi, stream);
      case 10: return launch_walk3_l64<10>(d, x, approx, details, lo, hi, stream);
      default: return MIFWT_ERR_UNSUPPORTED;
    }
  }
  switch (d->filt_len) {
    case 2: return launch_walk3_l<2>(d, x, approx, details, lo, hi, stream);
    case 4: return launch_walk3_l<4>(d, x, approx, details, lo, hi, stream);
    case 6: return launch_walk3_l<6>(d, x, approx, details, lo, hi, stream);
    case 8: return launch_walk3_l<8>(d, x, approx, details, lo, hi, stream);
    case 10: return launch_walk3_l<10>(d, x, approx, details, lo, hi, stream);
    default: return MIFWT_ERR_UNSUPPORTED;
  }
}

}  // namespace mifwt
