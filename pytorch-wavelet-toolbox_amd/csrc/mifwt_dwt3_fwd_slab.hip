// mifwt_dwt3_fwd_slab.hip — fully fused 3-D analysis level for LONG filters on volumes of short rows (gfx950): the SLAB form of the
// depth-walking kernel, kernel id 24.
//
// Seam: F.pad + F.conv3d([8,1,L,L,L], stride 2) + split of one level of wavedec3 / fswavedec3 (reference
// src/ptwt/conv_transform_3.py:121-141), the shape of the reference's own 3-D speed test (examples/speed_tests/timeitconv_3d.py:54-64:
// 32 x 100^3, db5, periodic).
//
// The strip form (mifwt_dwt3_fwd_walk.hip) gives a wave TR output rows of a 64-column strip and lets it filter its own 2 TR + L - 2 input
// rows along the row axis.  With ten taps TR is 2 (the depth pass holds L/2 x 8 bands x TR rows in registers), so a wave filters 12 rows
// per 4 it consumes, the workgroups are three waves, and a depth segment of 6 output slices walks 20 — five times the arithmetic of the
// level: 32 x 100^3 db5 199.7 us against 134.9 us for the composed route (planes + depth pass, 2.25 x the bytes).  Here
//   * a workgroup owns a SLAB: up to 28 output rows x every column x a long depth segment of one volume (one workgroup per CU);
//   * four LOADER waves bring the slab's rows of a slice in by LDS-DMA (several rows per 1-KiB request: lane -> (row, 16-byte piece), the
//     boundary map of rows and slices in the request's offsets) and fill the pad columns of what they loaded; everything but the slice
//     offset of a request and of a pad sample is worked out once per workgroup (a wave issues one instruction per four cycles: what a
//     loader executes per step is on the workgroup's critical path);
//   * ALL waves filter every staged row ONCE along the row axis (work item = four neighbouring outputs of a row, rows fastest over the
//     lanes; row pitches of 4 x odd floats keep the 16-byte LDS accesses of neighbouring rows in different banks) into an LDS image of
//     (W-low, W-high) pairs; then each lane of the compute waves — one (row pair, column) — runs the column pass from that image (L + 2
//     8-byte reads) and feeds the depth pass (rolling accumulators, as in the strip form); an output slice leaves every second step;
//   * step t = { column + depth pass of slice t | barrier | row pass of slice t + 1 | barrier }; the loaders request slice t + 3 (three
//     staged slices) and pad slice t + 1 during the first half: a request has two steps to land.
// Measured (32 x 100^3 db5 periodic, one level): 81-86 us against 135-142 us composed and 200 us in the strip form; with parts switched
// off (tools/slab_parts.py, diagnostics build): launch + barriers 18 us, row pass +12, column + depth pass +11, stores +7 .. 12, the
// requests +17 (results wrong in every such run).
// f32, L = 8 / 10, rows of at most 128 samples, every boundary mode.  Same sums in the same order as the strip form.
// Algorithmic traffic: 4 (B D H W read + 8 B Do Ho Wo written).
#include <atomic>

#include "mifwt_pyr.h"

namespace mifwt {

namespace {

constexpr int kSlabLpad = 8;       // floats in front of a staged row's body (the L - 2 left pad samples)
constexpr int kSlabLoaders = 4;
constexpr int kSlabMaxWaves = 16;  // compute + loader waves
constexpr int kSlabMaxReq = 20;    // requests of a loader wave per slice (76 staged rows, one row per request, four loaders)
constexpr int kSlabMaxPad = 8;     // pad samples per loader lane and slice (<= 24 rows x 17 samples over 64 lanes)

template <int L>
struct Slab3Args {
  const float* x;
  float* out[8];  // band s: bit 2 = depth high, bit 1 = row high, bit 0 = column high
  int64_t xs_b, os_b[2];
  uint32_t xs_d, xs_h;
  uint32_t os_d[2], os_h[2];
  int D, H, W, Do, Ho, Wo;
  int rw, ngroups;    // output rows of a slab (even), slabs per volume and segment
  int nseg, seg_out;  // depth segments, output slices per segment
  int ncw;            // compute waves
  int nq;             // row-pass items (four outputs) of a row
  int pitch_f, ppl, rpq;  // floats of a staged row, 16-byte pieces of it (= DMA lanes per row), rows per request
  int rin_max;        // staged rows of a slice: 2 rw + L - 2
  int raw_slot;       // bytes of a staged slice in LDS
  int wfp;            // (low, high) pairs of a row of the filtered image
  int npad;           // pad samples of a row: L - 2 in front, 2 Wo - W behind
  int mode, exp, dbg;  // exp: MIFWT_OPT_EXP (A/B runs of diagnostics builds: 1 loaders at default priority, 2 row pass on the compute waves only); dbg: MIFWT_OPT_DEBUG of -DMIFWT_DIAG builds (timing experiments, results wrong): 1 no stores, 2 no requests, 4 no row pass, 8 no column / depth pass, 16 no pad fill, 32 loaders keep only the barriers, 64 no depth pass either
  FastDiv div_wo, div_rin, div_g, div_s, div_ppl, div_npad, div_preq;  // (div_preq: by rpq * npad)
  f2 tap[L];
};

// at most n requests of this wave may still be in flight (they complete in order: everything older has landed); the count of an
// s_waitcnt is an immediate: a binary dispatch over 0 .. 2 kSlabMaxReq (two slices in flight)
template <int LO, int HI>
__device__ __forceinline__ void slab_wait_range(int n) {
  if constexpr (LO == HI) {
    pyr_wait_vm<LO>();
  } else {
    constexpr int MID = (LO + HI) / 2;
    if (n <= MID) slab_wait_range<LO, MID>(n);
    else slab_wait_range<MID + 1, HI>(n);
  }
}
__device__ __forceinline__ void slab_wait(int n) { slab_wait_range<0, 2 * kSlabMaxReq>(n); }  // (two slices in flight)

template <int L>
__global__ void __launch_bounds__(64 * kSlabMaxWaves) dwt3_fwd_slab_kernel(const Slab3Args<L> a) {
  constexpr int HL = L - 2, HP = L / 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int raw_slot = a.raw_slot;                 // bytes of a staged slice (its rows rounded up to whole requests, 32 spare bytes)
  unsigned char* const raw0 = smem;
  unsigned char* const wf0 = smem + 3 * raw_slot;  // (three staged slices: one being filtered, one landing, one requested)

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  uint32_t ug, us;
  const int img = __builtin_amdgcn_readfirstlane((int)a.div_s.divmod(a.div_g.divmod((uint32_t)xcd_remap(blockIdx.x, gridDim.x), ug), us));
  const int j0 = __builtin_amdgcn_readfirstlane((int)ug * a.rw);  // first output row of the slab
  const int nrows = min(a.rw, a.Ho - j0);
  const int rin = 2 * nrows + HL;  // staged rows the slab needs
  const int zA = __builtin_amdgcn_readfirstlane((int)us * a.seg_out), zB = min(a.Do, zA + a.seg_out);
  const int E0 = 2 * zA - HL;          // first input slice (extended index) of the walk
  const int nsl = 2 * (zB - zA) + HL;  // slices = steps (even)
  const bool zero_mode = a.mode == MIFWT_MODE_ZERO;
  Fold1 fold;
  fold.set(a.mode);

  // ROW PASS of one staged slice -> filtered image, every staged row once, by ALL waves of the workgroup (the loaders are idle most of a
  // step).  A work item = four neighbouring outputs of a row: L + 6 samples for the four (a lane that made one output read L; the pass
  // was bound by its LDS round trips, one per item and wave: 34 of 91 us with two outputs an item on the compute waves alone)
  const int gl = wave * 64 + lane;  // lane of the workgroup
  const int nl_all = (MIFWT_EXPW(a) & 2) ? 64 * a.ncw : 64 * (a.ncw + kSlabLoaders);
  const int nitems = a.rin_max * a.nq;  // (row fastest: neighbouring lanes read neighbouring ROWS — with a row pitch of 4 x odd floats their
                                         // 16-byte reads fall into different banks; column fastest, lanes 32 bytes apart, had two lanes a bank)
  auto row_pass = [&](int t) {
    const unsigned char* const rs = raw0 + (t % 3) * raw_slot;
    unsigned char* const ws = wf0;  // (one filtered image: written in the second half of a step, read in the first half of the next)
    if (MIFWT_DBG(a) & 4) return;
    for (int id = gl; id < nitems; id += nl_all) {
      uint32_t ru;
      const int kq = (int)a.div_rin.divmod((uint32_t)id, ru), r = (int)ru;
      if (r >= rin) continue;
      const f2* row = reinterpret_cast<const f2*>(rs + (r * a.pitch_f + kSlabLpad + 8 * kq - HL) * 4);
      f2 xx[HP + 3];
#pragma unroll
      for (int p = 0; p < HP + 3; ++p) xx[p] = row[p];  // samples 8 kq - HL + 2 p, + 1
      f2 v[4];
#pragma unroll
      for (int p = 0; p < HP; ++p) {  // output 4 kq + o: samples xx[o + p] <-> taps L - 1 - 2 p, L - 2 - 2 p
#pragma unroll
        for (int o = 0; o < 4; ++o) {
          if (p == 0) v[o] = vmul_lo(a.tap[L - 1], xx[o]);
          else vfma_lo(v[o], a.tap[L - 1 - 2 * p], xx[o + p]);
          vfma_hi(v[o], a.tap[L - 2 - 2 * p], xx[o + p]);
        }
      }
      // (the outputs past Wo of a row's last item land in the spare columns of the image, which nobody reads)
      f4* dst = reinterpret_cast<f4*>(ws + (r * a.wfp + 4 * kq) * 8);
      dst[0] = (f4){v[0].x, v[0].y, v[1].x, v[1].y};
      dst[1] = (f4){v[2].x, v[2].y, v[3].x, v[3].y};
    }
  };

  // =====================================================================================================================
  // loader waves: loader l takes the requests l, l + kSlabLoaders, ... of every slice.  Everything a request needs but the slice is the
  // same for all slices and is worked out ONCE: per-lane global offsets (row through the boundary map + piece), the lanes that take
  // part, and the (destination, source) pairs of the pad samples of the rows this wave loads.  (A first version did that arithmetic
  // per slice — 45 instructions a request, 40 a pad sample: the loaders were the launch, 215 us for 32 x 100^3 db5 with 87 us left when
  // every load, pass and store was switched off.)
  if (wave >= a.ncw) {
    const int l = wave - a.ncw;
    const uint32_t vol_bytes = (MIFWT_DBG(a) & 2) ? 0u : (uint32_t)(((int64_t)(a.D - 1) * a.xs_d + (int64_t)(a.H - 1) * a.xs_h + a.W) * 4);
    const rsrc_t xr = pyr_rsrc(a.x + (int64_t)img * a.xs_b, vol_bytes);
    const rsrc_t xr_dead = pyr_rsrc(a.x + (int64_t)img * a.xs_b, 0);  // every lane out of range: zeros land
    const uint32_t row_bytes = a.xs_h * 4u, slice_bytes = a.xs_d * 4u;
    const int r_first = 2 * j0 - HL;
    uint32_t piece;
    const int q = (int)a.div_ppl.divmod((uint32_t)lane, piece);  // lane -> (row of the request, 16-byte piece of the row)
    const bool lane_on = q < a.rpq && (int)piece * 4 < a.W;
    const int nreq = (rin + a.rpq - 1) / a.rpq;
    const int nown = nreq > l ? (nreq - l + kSlabLoaders - 1) / kSlabLoaders : 0;  // requests of this wave per slice (<= kSlabMaxReq)
    const uint32_t pitch_b = (uint32_t)a.pitch_f * 4u;
    // (a wave issues one instruction per four cycles at best, and a step is over when its slowest wave is: what a loader executes per
    // step is on the critical path of the workgroup — ~400 instructions a step in the second version of this kernel, 0.65 us of 1.9)
    uint32_t voff[kSlabMaxReq];
#pragma unroll
    for (int j = 0; j < kSlabMaxReq; ++j) {
      const int row = (l + kSlabLoaders * j) * a.rpq + q;  // staged row of this lane in request j
      const int ri = r_first + row;
      const bool dead = row >= rin || (zero_mode && (unsigned)ri >= (unsigned)a.H);  // (rows past the slab's: zeros into the slot's spare rows)
      voff[j] = dead ? kPyrOob : (uint32_t)fold(ri, a.H) * row_bytes + 16u * piece;
    }
    // pad samples of this wave's rows, flattened over its lanes: item -> (own request, row of the request, pad sample); byte offsets of
    // destination and source inside a staged slice (items past the end: the slot's spare 16 bytes, both ways)
    uint32_t pdst[kSlabMaxPad], psrc[kSlabMaxPad];
    {
      const int per_req = a.rpq * a.npad, nitems = nown * per_req;
      const uint32_t spare = (uint32_t)raw_slot - 16u;
#pragma unroll
      for (int u = 0; u < kSlabMaxPad; ++u) {
        const int it = lane + 64 * u;
        uint32_t p, rest;
        const int j = (int)a.div_preq.divmod((uint32_t)it, rest);
        const int qq = (int)a.div_npad.divmod(rest, p);
        const int row = (l + kSlabLoaders * j) * a.rpq + qq;
        const int c = (int)p < HL ? (int)p - HL : a.W + ((int)p - HL);
        const bool on = it < nitems && row < rin;
        pdst[u] = on ? 4u * (uint32_t)(row * a.pitch_f + kSlabLpad + c) : spare;
        psrc[u] = on ? 4u * (uint32_t)(row * a.pitch_f + kSlabLpad + fold(c, a.W)) : spare;
      }
    }
    if (!(MIFWT_EXPW(a) & 1)) __builtin_amdgcn_s_setprio(3);
    const uint32_t la0 = (uint32_t)(kSlabLpad * 4) + (uint32_t)(l * a.rpq) * pitch_b, la_step = (uint32_t)(kSlabLoaders * a.rpq) * pitch_b;
    // requests of slice t into slot t mod 3; returns how many this wave issued
    auto issue = [&](int t) -> int {
      if (t >= nsl) return 0;
      const int e = E0 + t;
      const bool sdead = zero_mode && (unsigned)e >= (unsigned)a.D;
      const uint32_t sbase = __builtin_amdgcn_readfirstlane(sdead ? 0u : (uint32_t)fold(e, a.D) * slice_bytes);
      const rsrc_t rs = sdead ? xr_dead : xr;
      uint32_t la = (uint32_t)(t % 3) * (uint32_t)raw_slot + la0;
      if (lane_on) {
        uint32_t keep;
        asm volatile("s_mov_b32 %0, m0" : "=s"(keep));  // (M0 = LDS base of a request; nothing else in between uses it)
#pragma unroll
        for (int j = 0; j < kSlabMaxReq; ++j) {
          if (j < nown) {
            asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(voff[j]), "s"(rs), "s"(sbase), "s"(la) : "memory");
            la += la_step;
          }
        }
        asm volatile("s_mov_b32 m0, %0" ::"s"(keep));
      }
      return nown;
    };
    // pad columns of slice t (its requests of THIS wave have landed)
    auto pads = [&](int t) {
      if (t >= nsl || (MIFWT_DBG(a) & 16)) return;
      unsigned char* const base = raw0 + (t % 3) * raw_slot;
      if (zero_mode) {
#pragma unroll
        for (int u = 0; u < kSlabMaxPad; ++u) *reinterpret_cast<float*>(base + pdst[u]) = 0.f;
        return;
      }
      float v[kSlabMaxPad];
#pragma unroll
      for (int u = 0; u < kSlabMaxPad; ++u) v[u] = *reinterpret_cast<const float*>(base + psrc[u]);
#pragma unroll
      for (int u = 0; u < kSlabMaxPad; ++u) *reinterpret_cast<float*>(base + pdst[u]) = v[u];
    };
    if (MIFWT_DBG(a) & 32) {  // (timing experiment: the loaders only keep the barriers)
      for (int t = 0; t < nsl + 2; ++t) __syncthreads();
      return;
    }
    issue(0);
    const int n1 = issue(1);
    int n2 = issue(2);
    slab_wait(n1 + n2);
    pads(0);
    __syncthreads();
    if (!(MIFWT_EXPW(a) & 2)) row_pass(0);  // (this wave's share)
    __syncthreads();
#pragma unroll 1
    for (int t = 0; t < nsl; ++t) {
      const int n3 = issue(t + 3);  // into the slot of slice t, filtered in the second half of step t - 1
      slab_wait(n2 + n3);           // slice t + 1 has landed
      pads(t + 1);
      n2 = n3;
      __syncthreads();
      if (t + 1 < nsl && !(MIFWT_EXPW(a) & 2)) row_pass(t + 1);
      __syncthreads();
    }
    return;
  }

  // =====================================================================================================================
  // compute waves
  uint32_t ku;
  const int jp = (int)a.div_wo.divmod((uint32_t)gl, ku);  // row pair of the slab, column
  const int k = (int)ku;
  const bool active = 2 * jp < nrows;

  float* obase[8];
#pragma unroll
  for (int b = 0; b < 8; ++b) obase[b] = a.out[b] + (int64_t)img * a.os_b[b == 0 ? 0 : 1];

  PyrAcc<L, 4, f2> acc;
  acc.clear();

  // column pass of the lane's two output rows from the filtered image: hv[2 j] = (Ha Wa, Hd Wa), hv[2 j + 1] = (Ha Wd, Hd Wd) of row j
  auto col_pass = [&](f2 (&hv)[4]) {
    const f2* col = reinterpret_cast<const f2*>(wf0) + (4 * jp) * a.wfp + k;
    if (MIFWT_DBG(a) & 8) {
#pragma unroll
      for (int c = 0; c < 4; ++c) hv[c] = (f2){1.f, 2.f};
      return;
    }
    // (rows from the last to the first: the strip form adds its terms in the order of the taps, and so does this)
#pragma unroll
    for (int i = L + 1; i >= 0; --i) {
      const f2 v = col[i * a.wfp];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int m = 2 * j + (L - 1) - i;  // tap of staged row i for output row j
        if (m < 0 || m >= L) continue;
        if (m == 0) {
          hv[2 * j] = vmul_lo(a.tap[m], v);
          hv[2 * j + 1] = vmul_hi(a.tap[m], v);
        } else {
          vfma_lo(hv[2 * j], a.tap[m], v);
          vfma_hi(hv[2 * j + 1], a.tap[m], v);
        }
      }
    }
  };

  auto emit = [&](auto sl_tag, int z) {
    constexpr int SL = decltype(sl_tag)::value;
    if (MIFWT_DBG(a) & 1) return;
    const uint32_t za = (uint32_t)z * a.os_d[0] + (uint32_t)k, zd = (uint32_t)z * a.os_d[1] + (uint32_t)k;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int y = j0 + 2 * jp + j;
      if (2 * jp + j < nrows) {
        const uint32_t oa = za + (uint32_t)y * a.os_h[0], od = zd + (uint32_t)y * a.os_h[1];
        // acc.lo[.][2 j] = (D a, D d) of Ha Wa; acc.hi[.][2 j] of Hd Wa; acc.lo[.][2 j + 1] of Ha Wd; acc.hi[.][2 j + 1] of Hd Wd
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
  };

  // step t: { column + depth pass of slice t | barrier | row pass of slice t + 1 | barrier } — the loaders request slice t + 3 and pad slice
  // t + 1 during the first half, so a request has two whole steps to land (one step, with a single barrier a step, left every step waiting
  // for its requests: 1.7 against 1.35 us a step)
  __syncthreads();  // slice 0 staged
  row_pass(0);
  __syncthreads();
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
      f2 hv[4];
      if (active && !(MIFWT_DBG(a) & 64)) {
        col_pass(hv);
        acc.template feed<0, R>(a.tap, hv);
      }
      __syncthreads();
      row_pass(2 * p + 1);  // (nsl is even: slice 2 p + 1 exists)
      __syncthreads();
      if (active && !(MIFWT_DBG(a) & 64)) {
        col_pass(hv);
        acc.template feed<1, R>(a.tap, hv);
        const int z = zA + p - (HP - 1);
        if (p >= HP - 1 && z < zB) emit(std::integral_constant<int, PyrAcc<L, 4, f2>::done(R)>{}, z);
      }
      __syncthreads();
      if (2 * p + 2 < nsl) row_pass(2 * p + 2);
      __syncthreads();
    });
  }
}

struct SlabPlan {
  int rw, ngroups, ncw, pitch_f, ppl, rpq, rin_max, raw_slot, wfp, npad, nseg, seg_out, lds, slots;
};

bool slab_plan(const mifwt_level_desc* d, SlabPlan* p) {
  const int L = d->filt_len, HL = L - 2;
  const int W = (int)d->sig_extent[2], Ho = (int)d->coef_extent[1], Wo = (int)d->coef_extent[2], Do = (int)d->coef_extent[0];
  p->npad = HL + (2 * Wo - W);
  // a staged row: left pad, body, right pad (2 Wo samples behind the left pad), whole 16-byte pieces; 4 x ODD floats, so that the
  // 16-byte reads of neighbouring rows fall into different LDS banks.  (The row pass reads whole groups of four outputs: the last group
  // of a row may read up to four floats of the next row, or of the slot's spare bytes, for outputs nobody uses.)
  p->pitch_f = (kSlabLpad + std::max(2 * Wo, W) + 3) & ~3;
  if (!((p->pitch_f / 4) & 1)) p->pitch_f += 4;
  p->ppl = p->pitch_f / 4;
  if (p->ppl > 64) return false;
  p->rpq = 64 / p->ppl;
  p->wfp = (Wo + 3) & ~3;  // (16-byte stores of four neighbouring outputs; 2 x odd pairs a row: bank spread as above)
  p->wfp += 2;
  if (!((p->wfp / 2) & 1)) p->wfp += 2;
  // the tallest slab: one lane per (row pair, column), two staged and two filtered slices in LDS
  const int max_lanes = 64 * (kSlabMaxWaves - kSlabLoaders);
  int pairs = std::min(max_lanes / Wo, (Ho + 1) / 2);
  auto raw_of = [&](int prs) { return ((4 * prs + HL + p->rpq - 1) / p->rpq * p->rpq) * p->pitch_f * 4 + 32; };  // (rows of whole requests, 32 spare bytes)
  auto lds_of = [&](int prs) { return 3 * raw_of(prs) + (4 * prs + HL) * p->wfp * 8; };
  // ... and what a loader wave keeps per lane: at most kSlabMaxReq requests and 64 kSlabMaxPad pad samples a slice, 16-bit offsets
  auto fits = [&](int prs) {
    const int rin = 4 * prs + HL, nreq = (rin + p->rpq - 1) / p->rpq, nown = (nreq + kSlabLoaders - 1) / kSlabLoaders;
    return lds_of(prs) <= 156 * 1024 && nown <= kSlabMaxReq && nown * p->rpq * p->npad <= 64 * kSlabMaxPad && raw_of(prs) < (1 << 30);
  };
  while (pairs > 1 && !fits(pairs)) --pairs;
  if (pairs < 1 || !fits(pairs)) return false;
  p->ngroups = (Ho + 2 * pairs - 1) / (2 * pairs);
  pairs = ((Ho + p->ngroups - 1) / p->ngroups + 1) / 2;  // slabs of equal height
  p->rw = 2 * pairs;
  p->ngroups = (Ho + p->rw - 1) / p->rw;
  p->ncw = (pairs * Wo + 63) / 64;
  p->rin_max = 2 * p->rw + HL;
  p->lds = lds_of(pairs);
  p->raw_slot = raw_of(pairs);
  // depth segments: one workgroup per CU and LDS share; a segment of n output slices walks 2 n + L - 2
  int ncu = 256;
  {
    static std::atomic<int> ncu_of[64];  // (per device, asked once)
    int dev = 0;
    if (hipGetDevice(&dev) == hipSuccess) {
      int v = ncu_of[dev & 63].load(std::memory_order_relaxed);
      if (v == 0 && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ncu_of[dev & 63].store(v, std::memory_order_relaxed);
      if (v > 0) ncu = v;
    }
  }
  // (workgroups a CU holds: by LDS and by waves — the kernel takes ~115 registers a lane: sixteen waves a CU)
  const int wpc = std::max(1, std::min((160 * 1024) / (p->lds + 1024), 16 / (p->ncw + kSlabLoaders)));
  p->slots = ncu * wpc;
  const int64_t base = d->batch * p->ngroups;
  int nseg = (int)(((int64_t)ncu * wpc + base - 1) / base);
  if (g_options[MIFWT_OPT_ROWS_PER_CHUNK] > 0) nseg = (Do + g_options[MIFWT_OPT_ROWS_PER_CHUNK] - 1) / g_options[MIFWT_OPT_ROWS_PER_CHUNK];
  if (nseg > Do / 2) nseg = Do / 2;
  if (nseg < 1) nseg = 1;
  p->seg_out = (Do + nseg - 1) / nseg;
  p->nseg = (Do + p->seg_out - 1) / p->seg_out;
  return true;
}

template <int L>
int launch_slab3(const mifwt_level_desc* d, const SlabPlan& p, const void* x, void* approx, void* const* details, const double* lo,
                 const double* hi, hipStream_t stream) {
  Slab3Args<L> a;
  a.x = static_cast<const float*>(x);
  for (int s = 0; s < 8; ++s) a.out[s] = static_cast<float*>(s == 0 ? approx : details[s - 1]);
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
  a.dbg = g_options[MIFWT_OPT_DEBUG];
  a.exp = exp_word();
  a.rw = p.rw;
  a.ngroups = p.ngroups;
  a.nseg = p.nseg;
  a.seg_out = p.seg_out;
  a.ncw = p.ncw;
  a.pitch_f = p.pitch_f;
  a.ppl = p.ppl;
  a.rpq = p.rpq;
  a.rin_max = p.rin_max;
  a.raw_slot = p.raw_slot;
  a.wfp = p.wfp;
  a.npad = p.npad;
  a.div_wo = make_fastdiv((uint32_t)a.Wo);
  a.div_rin = make_fastdiv((uint32_t)a.rin_max);
  a.nq = (a.Wo + 3) / 4;  // row-pass items of a row
  a.div_g = make_fastdiv((uint32_t)a.ngroups);
  a.div_s = make_fastdiv((uint32_t)a.nseg);
  a.div_ppl = make_fastdiv((uint32_t)a.ppl);
  a.div_npad = make_fastdiv((uint32_t)std::max(1, a.npad));
  a.div_preq = make_fastdiv((uint32_t)std::max(1, a.npad * a.rpq));
  for (int m = 0; m < L; ++m) a.tap[m] = (f2){(float)lo[m], (float)hi[m]};
  const int64_t nblk = d->batch * a.ngroups * a.nseg;
  if (nblk > INT32_MAX / 8) return MIFWT_ERR_UNSUPPORTED;
  static DynLdsOnce lds_once;
  if (!lds_once.ensure(reinterpret_cast<const void*>(&dwt3_fwd_slab_kernel<L>), 160 * 1024)) return MIFWT_ERR_LAUNCH;
  hipLaunchKernelGGL((dwt3_fwd_slab_kernel<L>), dim3((unsigned)nblk), dim3(64 * (a.ncw + kSlabLoaders)), (size_t)p.lds, stream, a);
  return hipGetLastError() == hipSuccess ? MIFWT_OK : MIFWT_ERR_LAUNCH;
}

}  // namespace

bool dwt3_fwd_slab_supported(const mifwt_level_desc* d) {
  if (d->ndim != 3 || d->dtype != MIFWT_F32) return false;
  const int L = d->filt_len;
  if (L != 8 && L != 10) return false;
  if (d->mode < 0 || d->mode > MIFWT_MODE_SYMMETRIC || d->batch < 1) return false;
  if (d->sig_stride[3] != 1 || d->approx_stride[3] != 1 || d->detail_stride[3] != 1) return false;
  for (int i = 0; i < 3; ++i)
    if (d->sig_stride[i] < 0 || d->approx_stride[i] < 0 || d->detail_stride[i] < 0) return false;
  // single-fold boundary map: every extent at least as long as the filter; rows of at most 128 samples
  for (int i = 0; i < 3; ++i)
    if (d->sig_extent[i] < L || d->coef_extent[i] != (d->sig_extent[i] + L - 1) / 2) return false;
  if (d->sig_extent[2] > 128) return false;
  // one batch element addressable with 32-bit byte offsets (buffer-resource requests), 32-bit element offsets inside a band
  const int64_t span = (d->sig_extent[0] - 1) * d->sig_stride[1] + (d->sig_extent[1] - 1) * d->sig_stride[2] + d->sig_extent[2];
  if (span >= (int64_t(1) << 29)) return false;
  if (d->coef_extent[0] * d->approx_stride[1] >= (int64_t(1) << 31) || d->coef_extent[0] * d->detail_stride[1] >= (int64_t(1) << 31))
    return false;
  SlabPlan p;
  return slab_plan(d, &p);
}

// Where the slab form is ahead of the composed route (2-D planes + depth pass: 2.25 x the bytes, two launches) — measured, GPU time of a
// level from kernel traces, slab against composed, us (profiles/r06p_walk3_routes.txt):
//   db5   100^3  32 / 16 / 8 / 4 volumes:  98.5 / 53.2 / 36.5 / 27.8  against  142.2 / 67.3 / 37.9 / 25.5
//          54^3                            29.1 / 24.4 / 22.6 / 21.9  against   37.6 / 27.5 / 20.6 / 16.9
//          31^3                            21.4 / 16.0 / 15.4 / 15.1  against   20.4 / 14.9 / 12.5 / 12.0
//   db4   128^3                           167.4 / 93.4 / 47.8 / 32.4  against  329.4 / 172.9 / 80.0 / 44.3   (strip form, 16 volumes: 144)
//         100^3                            85.2 / 44.0 / 30.5 / 24.9  against  140.0 / 67.7 / 36.8 / 27.1
//          53^3                            24.4 / 18.5 / 17.1 / 16.3  against   32.2 / 22.6 / 16.6 / 13.0
//          30^3                            16.1 / 13.3 / 13.0 / 12.8  against   18.7 / 13.8 / 11.6 / 11.4
// i.e. it wants a volume of 10^5 samples and enough of them to fill the chip: thresholds on the volume and on the samples of the batch.
bool dwt3_fwd_slab_pays(const mifwt_level_desc* d) {
  if (!dwt3_fwd_slab_supported(d)) return false;
  const int64_t vol = d->sig_extent[0] * d->sig_extent[1] * d->sig_extent[2], total = vol * d->batch;
  if (vol < 100000) return false;
  if (d->filt_len == 10) return vol >= 500000 ? total >= (int64_t(1) << 23) : total >= 2400000;
  return vol >= 500000 ? total >= 4000000 : total >= 2300000;
}

int dwt3_fwd_slab_plan_query(const mifwt_level_desc* d, int* out, int capacity) {
  if (capacity < 12) return MIFWT_ERR_BADARG;
  if (!dwt3_fwd_slab_supported(d)) return MIFWT_ERR_UNSUPPORTED;
  SlabPlan p;
  if (!slab_plan(d, &p)) return MIFWT_ERR_UNSUPPORTED;
  const int v[12] = {p.rw, p.ngroups, p.ncw, p.pitch_f, p.rpq, p.rin_max, p.wfp, p.npad, p.nseg, p.seg_out, p.lds, dwt3_fwd_slab_pays(d) ? 1 : 0};
  for (int i = 0; i < 12; ++i) out[i] = v[i];
  return 12;
}

int dwt3_fwd_slab(const mifwt_level_desc* d, const void* x, void* approx, void* const* details, const double* lo, const double* hi,
                  hipStream_t stream) {
  if (!dwt3_fwd_slab_supported(d)) return MIFWT_ERR_UNSUPPORTED;
  SlabPlan p;
  if (!slab_plan(d, &p)) return MIFWT_ERR_UNSUPPORTED;
  return d->filt_len == 8 ? launch_slab3<8>(d, p, x, approx, details, lo, hi, stream) : launch_slab3<10>(d, p, x, approx, details, lo, hi, stream);
}

}  // namespace mifwt
