// mifwt_dwt2_fwd_pyr.hip — UP TO THREE consecutive 2-D analysis levels in one launch (gfx950), kernel id 16.
//
// Seam: NLEV trips of the reference's level loop (src/ptwt/conv_transform_2.py:142-149: _fwt_pad2 + F.conv2d(stride 2) +
// split); a pyramid returns only the detail bands of every level but the last (conv_transform_2.py:150-156), so the
// approximations in between never reach HBM here: they live in LDS rings.
//
// Shape of the work (every choice measured: tools/ubench.hip, tools/pyr_time.py, tools/pyr_prof.py; profiles/r02_*):
//   * a workgroup owns one row segment of one column GROUP of one image (the whole width of a plane up to 1280 columns;
//     wider planes are cut into groups that recompute L - 2 halo columns per level at their left seam) and runs one wave per
//     role: level-1 waves (two columns per lane, 128 columns per wave), level-2 waves (two per lane), level-3 waves (one per
//     lane), two LOADER waves.  Config 2: 5 + 3 + 3 + 2 waves.  The waves of a level tile the group's columns densely and
//     SHARE one staged level-0 row and one ring row per level (private 256-column strips with private halos, the first
//     design, requested every row 1.25 times and needed 5 + 5 + 5 waves: 4-9 % slower on every shape tried);
//   * rows STREAM through a wave: the vertical pass of every level keeps the L/2 outputs in flight in registers (rolling
//     accumulators with compile-time slot rotation, 2 packed FMAs per sample and band pair), nothing is re-read; level l+1
//     consumes the rows of level l from a 16-row LDS ring through the boundary index map (mirrored rows at the top / bottom
//     of the plane are ring rows), lagging by a fixed number of 8-row steps; the two rows of a pair are filtered interleaved
//     (four independent chains) and a sub-step's windows are all requested before the first is used;
//   * the LOADER waves issue every global load of the workgroup as LDS-DMA (buffer_load_dwordx4 ... lds, 1 KiB per
//     instruction, NON-TEMPORAL so that the streamed input does not evict the half-written output lines from L2: 107 -> 85
//     us on the traffic skeleton), three 4-row sub-steps ahead, at raised priority (they are the youngest waves of their
//     SIMDs); their vmcnt queues hold loads only.  The compute waves' queues hold stores only and are never waited on: with
//     both in one queue the in-order counter made every load wait for the acknowledgement of older stores (150 us for the
//     same traffic).  One s_barrier per 4-row sub-step hands a landed sub-buffer over.  (Measured and dropped: progress
//     counters in LDS instead of the barriers — waves that spin on a counter steal issue slots from the waves they wait
//     for: 145-170 against 111-119 us; three columns per level-1 lane, i.e. 3 + 3 + 3 waves: 112.6 against 108.5 us — a
//     lone wave runs at 8-9 cycles per instruction whatever shares its SIMD, so fewer, longer waves lose; 3 to 6 staging
//     sub-buffers: alike; pacing the requests with s_sleep instead of issuing a sub-step as one burst: 106-142 against 104 us.)
//   * stores are never branched around: rows a segment does not own go through a buffer resource of size 0;
//   * boundary extension: pad columns are filled inside LDS by the waves that read them, right before they do (a row is
//     complete one barrier after it was written, whoever wrote its columns); out-of-plane rows in zero mode are zero rows.
// Results agree with the per-level kernels to rounding (different summation order), with the fp64 oracle within 1e-6.
// f32, even L <= 8, modes zero / constant / reflect / symmetric (periodic needs the far side of the plane); input rows of any length
// and alignment.
// Algorithmic traffic: 4 B H W read + 4 B (3 H1 W1 [+ 3 H2 W2] + 4 H_N W_N) written.
#include "mifwt_pyr.h"

namespace mifwt {

unsigned long long* g_pyr_prof = nullptr;

constexpr int kPyrSub = 4;    // level-0 rows per sub-step (one barrier each)
constexpr int kPyrPad = 8;    // floats in front of a staged / ring row
constexpr int kPyrRing = 16;  // ring rows (+ one zero row at slot 16)
constexpr int kPyrCtl = 64;   // bytes in front of the staging area
constexpr int kPyrWaves = 16;
constexpr int kPyrMaxChunks = 5;  // 1 KiB requests per staged row (three for the first loader wave, two for the second)

template <int L, int NLEV>
struct PyrArgs {
  const float* x;
  float* det[NLEV];         // [level - 1]: the lowest of the level's three detail planes
  uint32_t doff[NLEV][3];   // byte offsets of the bands ad, da, dd from it (one buffer resource serves the three)
  uint32_t dspan[NLEV];     // the largest of them
  float* approx;        // band aa of level NLEV
  int64_t xs_b, ds_b[NLEV], as_b;
  int xs_h, ds_h[NLEV], as_h;
  int H[NLEV + 1], W[NLEV + 1];
  int ngroups, nseg, seg_rows;  // column groups per plane, row segments, level-NLEV rows per segment
  int seg0_rows;                // ... of the FIRST segment (= seg_rows unless MIFWT_OPT_DEBUG bit 7 asks for a longer one)
  int cpg0, cpg;                // level-NLEV columns of group 0 / of the other groups
  int nchunks, nbuf;            // 1 KiB requests per level-0 row, staging sub-buffers of kPyrSub rows
  int pitch0, pitch1, pitch2;   // bytes of a staged row / a ring-1 row / a ring-2 row
  int nl1, nl2, nl3;            // waves of level 1 / 2 / 3
  int mode;
  // HANDOVER between the row segments of an image (a.handover; needs a.ws): a segment starts its deep levels where its own level-1 rows
  // suffice and takes the last L - 2 (+ 1) approximation rows of levels 1 and 2 it needs from the segment BELOW it, which deposits its
  // first rows in the workspace — instead of streaming a prologue of (2^NLEV - 1) (L - 2) input rows
  int handover;
  unsigned long long nonce;   // value of a set flag of this call
  unsigned long long* flags;  // [image][segment]
  float* xrows;               // [image][segment][(L - 1) rows of W1 floats, (L - 1) rows of W2 floats]
  int xrow_stride;            // floats per (image, segment)
  int xcd_map;
  int exp;      // MIFWT_OPT_EXP: experiment word of the current A/B run (0 in the product)
  int l2split;  // the level-2 waves take one row pair in each half of a step (else both behind the step's second barrier)
  int compact;  // 0: 16 waves (6 + 3 + 3 level waves at most, two loaders), one workgroup per CU; 1: 8 waves (3 + 2 + 2, one loader), two per CU
  unsigned long long* prof;
  int dbg;
  f2 tap[L];
};

// one LDS-DMA piece for rows of any length, past every cache (sc0 sc1: the rows were written by another XCD): lanes whose 16 bytes
// start inside the row (voff < limit) move them to LDS [lds + 16 lane)
__device__ __forceinline__ void pyr_dma_masked(uint32_t voff, uint32_t limit, rsrc_t rsrc, uint32_t soff, uint32_t lds) {
  if (voff < limit) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen sc0 sc1 lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(soff), "s"(lds) : "memory");
  }
}

// the L + 2 samples under a lane's two columns, from an LDS row whose float index of the first sample is congruent to
// -(L - 2) modulo 4 (8-byte aligned for L = 4, 8: one 8-byte and then 16-byte reads; 16-byte aligned for L = 2, 6)
template <int L>
__device__ __forceinline__ void pyr_load_win2(const unsigned char* row, f2 (&w)[L / 2 + 1]) {
  constexpr int HP = L / 2;
  if constexpr (((L - 2) & 3) == 2) {
    w[0] = *reinterpret_cast<const f2*>(row);
#pragma unroll
    for (int j = 0; j < HP / 2; ++j) {
      const f4 v = *reinterpret_cast<const f4*>(row + 8 + 16 * j);
      w[1 + 2 * j] = (f2){v.x, v.y};
      w[2 + 2 * j] = (f2){v.z, v.w};
    }
  } else {
#pragma unroll
    for (int j = 0; j < (HP + 1) / 2; ++j) {
      const f4 v = *reinterpret_cast<const f4*>(row + 16 * j);
      w[2 * j] = (f2){v.x, v.y};
      w[2 * j + 1] = (f2){v.z, v.w};
    }
  }
}

// wave -> (role, index within the role).  Waves land on the four SIMDs round-robin (class = wave mod 4): level 1 = waves 0-4 and
// 13, level 2 = 5-7, level 3 = 9-11, loaders = 15 and 14; with five level-1 waves the classes hold {L1, L1} {L1, L2, L3}
// {L1, L2, L3, loader} {L1, L2, L3, loader}.  Waves 0 and 4 share a SIMD: they take interior columns, not the first / last
// level-1 wave, which also copy the boundary extension of every staged row.
__device__ __forceinline__ void pyr_role(int wave, int nl1, int nchunks, int compact, int& role, int& idx) {
  role = -1;
  idx = 0;
  if (compact) {  // eight waves: level 1 = waves 0-2, level 2 = 3-4, level 3 = 5-6, the loader = 7 (two per SIMD)
    if (wave < 3) role = kRoleL1, idx = wave;
    else if (wave < 5) role = kRoleL2, idx = wave - 3;
    else if (wave < 7) role = kRoleL3, idx = wave - 5;
    else if (wave == 7) role = kRoleLoad;
    return;
  }
  if (wave < 5) {
    role = kRoleL1;
    idx = nl1 < 5 ? wave : (wave == 0 ? 1 : (wave == 1 ? 0 : (wave == 2 ? 3 : (wave == 3 ? 4 : 2))));
  } else if (wave == 13) {
    role = kRoleL1;
    idx = 5;
  } else if (wave >= 5 && wave <= 7) {
    role = kRoleL2;
    idx = wave - 5;
  } else if (wave >= 9 && wave <= 11) {
    role = kRoleL3;
    idx = wave - 9;
  } else if (wave == 15 || (wave == 14 && nchunks >= 2)) {
    role = kRoleLoad;
    idx = 15 - wave;
  }
}

template <int L, int NLEV, bool PROF, bool ST16>
__global__ void __launch_bounds__(64 * kPyrWaves) dwt2_fwd_pyr_kernel(const PyrArgs<L, NLEV> a) {
  constexpr int HL = L - 2, HP = L / 2;
  constexpr int NC1 = 2;          // columns per level-1 lane (three were measured: 112.6 against 108.5 us on config 2)
  constexpr int NP = 2 * HL + 1;  // extension columns of a row: HL on the left, HL + 1 on the right
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int role, widx;
  pyr_role(wave, a.nl1, a.nchunks, a.compact, role, widx);
  if (wave == 12 && a.handover && !a.compact) role = kRoleXchg;
  if (role < 0 || (role == kRoleL1 && widx >= a.nl1) || (role == kRoleL2 && (NLEV < 2 || widx >= a.nl2)) ||
      (role == kRoleL3 && (NLEV < 3 || widx >= a.nl3)))
    return;  // (a wave that has ended does not take part in the barriers of the others)

  int bid = blockIdx.x;
  if (a.xcd_map) {  // workgroups go to the 8 XCDs round-robin: give every XCD whole images (all row segments of an image share an L2)
    const int per = gridDim.x >> 3;
    bid = (bid & 7) * per + (bid >> 3);
  }
  const int grp = bid % a.ngroups;
  bid /= a.ngroups;
  // (handover: a segment waits, at its end, for the segment below it, which therefore gets the LOWER block index — workgroups start in
  // index order, so whatever a workgroup waits for is already running or done)
  const int seg = a.handover ? a.nseg - 1 - bid % a.nseg : bid % a.nseg, img = bid / a.nseg;
  const bool zero_mode = a.mode == MIFWT_MODE_ZERO;
  Fold1 fold;
  fold.set(a.mode);

  // ---- row ranges of this segment: computed rows [rA, rB) and owned rows [oA, oB) per level (index = level) -------------
  int rA[NLEV + 1], rB[NLEV + 1], oA[NLEV + 1], oB[NLEV + 1];
  int lim[NLEV + 1];  // rows of level l below lim[l] come out of this workgroup's own passes (the rest of [rA, rB): handed over)
  if (!a.handover) {
    oA[NLEV] = rA[NLEV] = seg == 0 ? 0 : a.seg0_rows + (seg - 1) * a.seg_rows;
    oB[NLEV] = rB[NLEV] = seg == a.nseg - 1 ? a.H[NLEV] : min(a.H[NLEV], rA[NLEV] + (seg == 0 ? a.seg0_rows : a.seg_rows));
#pragma unroll
    for (int l = NLEV - 1; l >= 1; --l) {
      oA[l] = 2 * oA[l + 1];
      oB[l] = oB[l + 1] == a.H[l + 1] ? a.H[l] : min(a.H[l], 2 * oB[l + 1]);
      rA[l] = max(0, 2 * rA[l + 1] - HL);
      rB[l] = min(a.H[l], 2 * rB[l + 1]);
    }
#pragma unroll
    for (int l = 1; l <= NLEV; ++l) lim[l] = rB[l];
  } else {
    // ownership boundaries bottom-up: the last level as without handover, level l from the first row that level l + 1 of this segment
    // reads (2 o - (L - 2)): a segment OWNS what it used to compute as its prologue, and stops where the next one starts
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int sg = seg + e;
      int o = sg == 0 ? 0 : (sg >= a.nseg ? a.H[NLEV] : min(a.H[NLEV], a.seg0_rows + (sg - 1) * a.seg_rows));
#pragma unroll
      for (int l = NLEV; l >= 1; --l) {
        if (l < NLEV) o = sg == 0 ? 0 : (sg >= a.nseg ? a.H[l] : max(0, 2 * o - HL));
        if (e == 0) oA[l] = rA[l] = o;
        else oB[l] = lim[l] = o;
      }
    }
    rB[NLEV] = oB[NLEV];
#pragma unroll
    for (int l = NLEV - 1; l >= 1; --l) rB[l] = min(a.H[l], 2 * lim[l + 1]);  // (beyond lim[l]: rows the segment below hands over)
  }
  const bool top = seg == 0;
  const int D2 = top ? pyr_lag2(L) : pyr_lag2_inner(L);
  const int D3 = top ? pyr_lag3(L) : pyr_lag3_inner(L);
  const int E0 = 2 * rA[1] - HL, e0_end = 2 * lim[1];
  const int npair1 = lim[1] - rA[1] + HP - 1;  // (pairs of the workgroup's OWN passes: rows from lim[l] on are handed over)
  const int nsteps1 = (npair1 + 3) / 4;
  int nsteps = nsteps1, npair2 = 0, npair3 = 0;
  if constexpr (NLEV >= 2) {
    npair2 = lim[2] - rA[2] + HP - 1;
    nsteps = max(nsteps, D2 + (npair2 + 1) / 2);
  }
  if constexpr (NLEV >= 3) {
    npair3 = lim[3] - rA[3] + HP - 1;
    nsteps = max(nsteps, D3 + npair3);
  }
  const int nsub = 2 * nsteps, nsub1 = 2 * nsteps1;

  // ---- column ranges of this group: computed [cA, cB), owned [pA, pB) per level -----------------------------------------
  int cA[NLEV + 1], cB[NLEV + 1], pA[NLEV + 1], pB[NLEV + 1];
  pA[NLEV] = cA[NLEV] = grp == 0 ? 0 : a.cpg0 + (grp - 1) * a.cpg;
  pB[NLEV] = cB[NLEV] = grp == a.ngroups - 1 ? a.W[NLEV] : min(a.W[NLEV], a.cpg0 + grp * a.cpg);
#pragma unroll
  for (int l = NLEV - 1; l >= 1; --l) {
    pA[l] = 2 * pA[l + 1];
    pB[l] = pB[l + 1] == a.W[l + 1] ? a.W[l] : min(a.W[l], 2 * pB[l + 1]);
    cA[l] = max(0, 2 * cA[l + 1] - HL);
    cB[l] = min(a.W[l], 2 * cB[l + 1]);
  }
  // lane grids: level-1 lanes own 3 columns from o1 on, placed so that a lane lies entirely inside or outside [pA1, ...)
  const int o1 = pA[1] - NC1 * ((pA[1] - cA[1] + NC1 - 1) / NC1);
  const int g0 = max(0, 2 * o1 - HL) & ~3;  // level-0 column at the start of a staged row's body (16-byte aligned)

  // ring-1 rows start sh1 floats later where that gives the level-2 windows (first sample: column 2 cA2 - HL) the alignment
  // pyr_load_win2 expects, whatever the group's position
  int sh1 = 0;
  if constexpr (NLEV >= 2) sh1 = (2 * cA[2] - cA[1]) & 2;  // 2 cA2 - cA1 is 0 or L - 2
  unsigned char* const stage = smem + kPyrCtl;
  unsigned char* const ring1 = stage + a.nbuf * kPyrSub * a.pitch0;
  unsigned char* const ring2 = ring1 + (kPyrRing + 1) * a.pitch1;

  // =====================================================================================================================
  // loader wave
  if (role == kRoleLoad) {
    const uint32_t img_bytes = ((uint32_t)(a.H[0] - 1) * (uint32_t)a.xs_h + (uint32_t)a.W[0]) * 4u;
    const rsrc_t xr = pyr_rsrc(a.x + (int64_t)img * a.xs_b, img_bytes);
    const rsrc_t xr_dead = pyr_rsrc(a.x + (int64_t)img * a.xs_b, 0);  // every lane out of range: a row of zeros lands
    const uint32_t row_bytes = (uint32_t)a.xs_h * 4u;
    // loader `widx` of `nload` requests the 1-KiB pieces widx, widx + nload, widx + 2 nload of every row
    const int nload = a.compact ? 1 : 2;
    const int mych = (a.nchunks - widx + nload - 1) / nload;
    uint32_t voff[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int c = g0 + 256 * (widx + nload * j) + 4 * lane;
      voff[j] = (j < mych && c < a.W[0]) ? 4u * (uint32_t)c : kPyrOob;
    }
    if (!(a.dbg & 16)) __builtin_amdgcn_s_setprio(3);  // the youngest wave of its SIMD, and the one everybody waits for
    __syncthreads();  // the other waves have initialised their LDS
    auto run = [&](auto nch_tag, auto str_tag) {
      constexpr int NCH = decltype(nch_tag)::value;
      constexpr int STR = decltype(str_tag)::value;
      constexpr int PER = kPyrSub * NCH;
      int ib = 0;  // staging sub-buffer of the next sub-step to be requested (sub-steps are requested in order)
      auto issue = [&](int t) {
        const uint32_t buf = (uint32_t)ib * (uint32_t)(kPyrSub * a.pitch0) + (uint32_t)kPyrCtl + kPyrPad * 4u + 1024u * (uint32_t)widx;
        ib = ib + 1 == a.nbuf ? 0 : ib + 1;
        if (a.dbg & 2) return;
#pragma unroll
        for (int kk = 0; kk < kPyrSub; ++kk) {
          const int e = E0 + kPyrSub * t + kk;
          const bool dead = e >= e0_end || (zero_mode && (unsigned)e >= (unsigned)a.H[0]);
          const uint32_t soff = dead ? 0u : (uint32_t)fold(e, a.H[0]) * row_bytes;
          pyr_dma_row<NCH, STR>(voff, dead ? xr_dead : xr, soff, buf + (uint32_t)(kk * a.pitch0));
        }
      };
      // nbuf - 1 sub-steps are requested ahead; at most 63 requests of a wave can be in flight
      const int ahead = a.nbuf - 1;
      for (int t = 0; t < ahead; ++t)
        if (t < nsub1) issue(t);
#pragma unroll 1
      for (int t = 0; t < nsub; ++t) {
        // sub-step t must have landed; the ones requested after it may still be in flight
        const int later = min(ahead - 1, nsub1 - 1 - t);
        if (later >= 6) pyr_wait_vm<(6 * PER > 63 ? 63 : 6 * PER)>();
        else if (later == 5) pyr_wait_vm<(5 * PER > 63 ? 63 : 5 * PER)>();
        else if (later == 4) pyr_wait_vm<(4 * PER > 63 ? 63 : 4 * PER)>();
        else if (later == 3) pyr_wait_vm<(3 * PER > 63 ? 63 : 3 * PER)>();
        else if (later == 2) pyr_wait_vm<(2 * PER > 63 ? 63 : 2 * PER)>();
        else if (later == 1) pyr_wait_vm<PER>();
        else pyr_wait_vm<0>();
        __syncthreads();
        if (t + ahead < nsub1) issue(t + ahead);  // into the buffer sub-step t - 1 was read from
      }
    };
    if (a.compact) pyr_dispatch<3>(mych - 1, [&](auto k) { run(std::integral_constant<int, decltype(k)::value + 1>{}, std::integral_constant<int, 0x400>{}); });
    else pyr_dispatch<3>(mych - 1, [&](auto k) { run(std::integral_constant<int, decltype(k)::value + 1>{}, std::integral_constant<int, 0x800>{}); });
    return;
  }

  unsigned long long waited = 0;
  const unsigned long long t_start = PROF ? __builtin_readcyclecounter() : 0;
  // (profiling build: wave 0 also leaves the workgroup's start / end on the 100 MHz wall clock and where it ran — HW_ID, XCC_ID — in
  // the slots of the unused waves 8 and 12: ramp, tail and the gap between launches, tools/pyr_prof.py)
  const unsigned long long w_start = PROF ? __builtin_amdgcn_s_memrealtime() : 0;
  auto prof_out = [&]() {
    if (PROF && lane == 0) {
      unsigned long long* o = a.prof + ((size_t)blockIdx.x * kPyrWaves + wave) * 2;
      o[0] = __builtin_readcyclecounter() - t_start;
      o[1] = waited;
      if (wave == 0) {
        unsigned long long* w = a.prof + ((size_t)blockIdx.x * kPyrWaves + 8) * 2;
        w[0] = w_start;
        w[1] = __builtin_amdgcn_s_memrealtime();
        w[8] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
        w[9] = __builtin_amdgcn_s_getreg((31 << 11) | 20);
      }
    }
  };
  f2 tap[L];
#pragma unroll
  for (int m = 0; m < L; ++m) tap[m] = a.tap[m];
  // LDS initialisation (pads in zero mode, the zero rows of the rings): the level-1 waves clear the staging area, the
  // level-2 waves the rings
  {
    const int nst = a.nbuf * kPyrSub * a.pitch0 / 16, nrg = ((kPyrRing + 1) * (a.pitch1 + a.pitch2)) / 16;
    if (role == kRoleL1)
      for (int i = widx * 64 + lane; i < nst; i += 64 * a.nl1) reinterpret_cast<f4*>(stage)[i] = (f4){0.f, 0.f, 0.f, 0.f};
    if (NLEV >= 2 && role == kRoleL2)
      for (int i = widx * 64 + lane; i < nrg; i += 64 * a.nl2) reinterpret_cast<f4*>(ring1)[i] = (f4){0.f, 0.f, 0.f, 0.f};
  }
  __syncthreads();  // ... before the loader's first row lands

  // =====================================================================================================================
  // exchange wave (handover): deposits this segment's first approximation rows of levels 1 and 2 for the segment above, and puts the
  // rows the segment below deposited into the rings when the level-1 / level-2 waves of this workgroup run out of rows of their own.
  // It only moves rows between LDS and the workspace; the passes do not know where a ring row came from.
  if constexpr (NLEV >= 2) {
    if (role == kRoleXchg) {
      constexpr int XM = HL + 1;  // rows handed over per level at most
      const int W1 = a.W[1], W2 = NLEV >= 3 ? a.W[2] : 0;
      const int nk1 = (W1 + 63) >> 6, nk2 = (W2 + 63) >> 6;
      const bool produce = seg > 0, consume = seg < a.nseg - 1;
      if (lane == 0) *reinterpret_cast<volatile unsigned long long*>(smem) = 0;  // (where the flag of the segment below will land)
      float* const mine = a.xrows + ((int64_t)img * a.nseg + seg) * a.xrow_stride;
      const float* const theirs = a.xrows + ((int64_t)img * a.nseg + seg + 1) * a.xrow_stride;
      // deposit: level-1 rows oA[1] + r (r < np1), level-2 rows oA[2] + r (r < np2) — what the segment above lacks
      const int np1 = produce ? 2 * oA[2] - oA[1] : 0;
      const int np2 = (produce && NLEV >= 3) ? 2 * oA[NLEV >= 3 ? 3 : 2] - oA[2] : 0;
      // take over: level-1 rows lim[1] + r (r < nc1), level-2 rows lim[2] + r (r < nc2)
      const int nc1 = consume ? rB[1] - lim[1] : 0;  // (rB[l] = 2 lim[l + 1]: what the level above needs)
      const int nc2 = (consume && NLEV >= 3) ? rB[2] - lim[2] : 0;
      // LDS area the deposited rows of the segment below are copied into (LDS-DMA), behind the rings
      const int pitchX1 = ((W1 + 3) & ~3) * 4 + 16, pitchX2 = ((W2 + 3) & ~3) * 4 + 16;
      unsigned char* const xa1 = ring2 + (NLEV >= 3 ? (kPyrRing + 1) * a.pitch2 : 0);
      unsigned char* const xa2 = xa1 + XM * pitchX1;
      const uint32_t xa1_off = (uint32_t)(xa1 - smem), xa2_off = (uint32_t)(xa2 - smem);
      // sub-steps: a level-1 row with sequence number q is written by its level-1 wave during sub-step q / 2, a level-2 row with
      // sequence number p during sub-step 2 (D2 + p / 2) + 1; a row is readable one barrier later
      const int q0c = lim[1] - rA[1] + HP - 1;  // sequence number of the first level-1 row taken over
      const int t_acq = max(0, q0c / 2 - 5);
      int t_rel = -1;
      if (produce) {
        t_rel = (np1 - 1 + HP - 1) / 2 + 1;
        if (NLEV >= 3 && np2 > 0) t_rel = max(t_rel, 2 * (D2 + (np2 - 1 + HP - 1) / 2) + 2);
      }
      const rsrc_t rmine = pyr_rsrc(mine, (uint32_t)a.xrow_stride * 4u);
      const rsrc_t rtheirs = pyr_rsrc(theirs, (uint32_t)a.xrow_stride * 4u);
      bool loaded = false;
#pragma unroll 1
      for (int t = 0; t < nsub; ++t) {
        __syncthreads();
        // ---- deposit -----------------------------------------------------------------------------------------------------
        if (produce && t <= t_rel + 6) {
#pragma unroll 1
          for (int r = 0; r < np1; ++r) {
            const int q = r + HP - 1;
            if (t != q / 2 + 1) continue;
            const unsigned char* row = ring1 + (q & (kPyrRing - 1)) * a.pitch1 + 4 * (kPyrPad + sh1 - cA[1]);
#pragma unroll 1
            for (int kk = 0; kk < nk1; ++kk) {
              const int c = 64 * kk + lane;
              const float v = *reinterpret_cast<const float*>(row + 4 * min(c, W1 - 1));
              __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v), rmine, c < W1 ? 4u * (uint32_t)(r * W1 + c) : kPyrOob, 0, 17);
            }
          }
          if constexpr (NLEV >= 3) {
#pragma unroll 1
            for (int r = 0; r < np2; ++r) {
              const int pq = r + HP - 1;
              if (t != 2 * (D2 + pq / 2) + 2) continue;
              const unsigned char* row = ring2 + (pq & (kPyrRing - 1)) * a.pitch2 + 4 * (kPyrPad - cA[2]);
#pragma unroll 1
              for (int kk = 0; kk < nk2; ++kk) {
                const int c = 64 * kk + lane;
                const float v = *reinterpret_cast<const float*>(row + 4 * min(c, W2 - 1));
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v), rmine, c < W2 ? 4u * (uint32_t)(XM * W1 + r * W2 + c) : kPyrOob, 0, 17);
              }
            }
          }
          if (t == t_rel + 6 || (t == nsub - 1 && t_rel + 6 > nsub - 1 && t >= t_rel)) {
            // Everything was deposited six sub-steps ago (the wait below is over before it starts: a wave that waits in here keeps the
            // whole workgroup at its next barrier).  The deposits are write-through stores (sc0 sc1: they bypass the XCD's L2, which is not coherent
            // with the other XCDs'), acknowledged once they are in memory; the flag goes the same way afterwards, and the reader uses
            // cache-bypassing loads — no cache-wide write-back / invalidate (an agent-scope release / acquire fence does exactly that:
            // buffer_wbl2 / buffer_inv on an L2 full of dirty output lines cost 19 us per launch, profiles/r03k_handover.txt)
            pyr_wait_vm<0>();
            if (lane == 0) __hip_atomic_store(a.flags + (int64_t)img * a.nseg + seg, a.nonce, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          }
        }
        // ---- take over ---------------------------------------------------------------------------------------------------
        if (consume && t_acq >= 4 && t == t_acq - 4) {
          // the flag of the segment below, requested past the caches into LDS four sub-steps before it is looked at (a wave that waits
          // for memory in here keeps the whole workgroup at its next barrier)
          const rsrc_t rf = pyr_rsrc(a.flags + (int64_t)img * a.nseg + seg + 1, 8);
          if (lane < 2) {
            uint32_t keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dword %1, %2, %3 offen sc0 sc1 lds\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(4u * (uint32_t)lane), "s"(rf), "s"(0u), "s"(0u) : "memory");
          }
        }
        if (consume && t >= t_acq) {
          if (!loaded) {
            loaded = true;
            // the segment below raised its flag long ago (it deposits within its first steps; this is the end of ours)
            bool seen = false;
            if (t_acq >= 4) {
              pyr_wait_vm<0>();
              seen = *reinterpret_cast<volatile unsigned long long*>(smem) == a.nonce;
            }
            if (!seen) {
              const unsigned long long* f = a.flags + (int64_t)img * a.nseg + seg + 1;
              int spins = 0;
              while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != a.nonce && ++spins < (1 << 20)) __builtin_amdgcn_s_sleep(32);
              // the segment below never raised its flag (its workgroup is not resident, or the workspace / call id is shared with another
              // launch): abort the launch — the host sees a failed kernel at its next synchronisation — rather than copy rows that were
              // never deposited
              if (spins >= (1 << 20)) __builtin_trap();
            }
            asm volatile("" ::: "memory");  // (the requests below are issued after the flag was seen)
            const uint32_t lane16 = 16u * (uint32_t)lane;
#pragma unroll 1
            for (int r = 0; r < nc1; ++r)
#pragma unroll 1
              for (int c = 0; c < (W1 + 255) >> 8; ++c)
                pyr_dma_masked(lane16 + 1024u * (uint32_t)c, 4u * (uint32_t)W1, rtheirs, 4u * (uint32_t)(r * W1), xa1_off + (uint32_t)(r * pitchX1) + 1024u * (uint32_t)c);
            if constexpr (NLEV >= 3) {
#pragma unroll 1
              for (int r = 0; r < nc2; ++r)
#pragma unroll 1
                for (int c = 0; c < (W2 + 255) >> 8; ++c)
                  pyr_dma_masked(lane16 + 1024u * (uint32_t)c, 4u * (uint32_t)W2, rtheirs, 4u * (uint32_t)(XM * W1 + r * W2),
                                 xa2_off + (uint32_t)(r * pitchX2) + 1024u * (uint32_t)c);
            }
          }
#pragma unroll 1
          for (int r = 0; r < nc1; ++r) {
            const int q = q0c + r;
            if (t != q / 2) continue;
            pyr_wait_vm<0>();  // (requested several sub-steps ago)
            unsigned char* row = ring1 + (q & (kPyrRing - 1)) * a.pitch1 + 4 * (kPyrPad + sh1 - cA[1]);
#pragma unroll 1
            for (int kk = 0; kk < nk1; ++kk) {
              const int c = 64 * kk + lane;
              const float v = *reinterpret_cast<const float*>(xa1 + r * pitchX1 + 4 * min(c, W1 - 1));
              if (c < W1) *reinterpret_cast<float*>(row + 4 * c) = v;
            }
          }
          if constexpr (NLEV >= 3) {
#pragma unroll 1
            for (int r = 0; r < nc2; ++r) {
              const int pq = lim[2] - rA[2] + HP - 1 + r;
              if (t != 2 * (D2 + pq / 2) + 1) continue;
              pyr_wait_vm<0>();
              unsigned char* row = ring2 + (pq & (kPyrRing - 1)) * a.pitch2 + 4 * (kPyrPad - cA[2]);
#pragma unroll 1
              for (int kk = 0; kk < nk2; ++kk) {
                const int c = 64 * kk + lane;
                const float v = *reinterpret_cast<const float*>(xa2 + r * pitchX2 + 4 * min(c, W2 - 1));
                if (c < W2) *reinterpret_cast<float*>(row + 4 * c) = v;
              }
            }
          }
        }
      }
      return;
    }
  }

  // =====================================================================================================================
  // level-1 wave: NC1 columns per lane
  if (role == kRoleL1) {
    const int gmax = (cB[1] - o1 + NC1 - 1) / NC1 - 1;               // last lane of the grid that has a column
    // 16-byte stores: lanes l and l + 32 hold neighbouring column pairs (they exchange rows in front of a store, pyr_swap_rows)
    constexpr bool st16 = ST16;
    const int glane = 64 * widx + (st16 ? 2 * (lane & 31) + (lane >> 5) : lane);
    const int G = min(glane, gmax);                           // (lanes beyond it repeat that one and store nothing)
    const bool real = glane <= gmax;
    const int c0 = o1 + NC1 * G;
    const uint32_t win = 4u * (uint32_t)(kPyrPad - HL + 2 * o1 - g0 + 2 * NC1 * G);  // the lane's L - 2 + 2 NC1 staged samples
    const bool full = real && c0 >= pA[1] && c0 + 1 < pB[1];
    const bool rag1 = real && c0 >= pA[1] && c0 + 1 == pB[1];  // the lane that holds the last owned column alone
    const uint32_t sv2 = full ? 4u * (uint32_t)c0 : kPyrOob, sv1 = rag1 ? 4u * (uint32_t)c0 : kPyrOob;
    const bool rag = __builtin_amdgcn_readfirstlane(__builtin_amdgcn_ballot_w64(rag1) != 0);
    // ring-1 positions of the lane's columns (columns outside the computed range go to float 0 of the row, which nobody reads)
    uint32_t rw[NC1];
#pragma unroll
    for (int k = 0; k < NC1; ++k) rw[k] = (real && c0 + k >= cA[1] && c0 + k < cB[1]) ? 4u * (uint32_t)(kPyrPad + sh1 + c0 + k - cA[1]) : 0u;
    // level-0 pad fill, by the waves whose windows reach the pads: lane -> (row kk of the sub-step, pad column)
    uint32_t f_src = 0, f_dst = 0;
    bool f_on = false;
    {
      const int wlo = 2 * (o1 + NC1 * min(64 * widx, gmax)) - HL, whi = 2 * (o1 + NC1 * min(64 * widx + 63, gmax) + NC1 - 1) + 1;
      const int kk = lane / NP, p = lane - kk * NP;
      const bool left = p < HL;
      const int e = left ? p - HL : a.W[0] + (p - HL);
      // (zero mode: the pads are zeros from the LDS initialisation — except the right one of rows that are not a multiple of 4
      // samples long, where the last lane of a row's DMA request brings up to three samples of whatever follows the row)
      const bool zfix = zero_mode && !left && (a.W[0] & 3) != 0;
      if (kk < kPyrSub && (!zero_mode || zfix) && (left ? wlo < 0 : whi >= a.W[0])) {
        f_on = true;
        f_src = (uint32_t)(kk * a.pitch0) + 4u * (uint32_t)(kPyrPad + (zero_mode ? 0 : fold(e, a.W[0])) - g0);
        f_dst = (uint32_t)(kk * a.pitch0) + 4u * (uint32_t)(kPyrPad + e - g0);
      }
    }
    const bool f_any = __builtin_amdgcn_readfirstlane(__builtin_amdgcn_ballot_w64(f_on) != 0) && !(a.dbg & 16384);
    // one buffer resource for the three detail planes of the image (band = scalar offset), one for the approximation; a row
    // the segment does not own is stored at a per-lane offset beyond every resource (dropped) — the resources never change
    const uint32_t dbytes = (a.dbg & 1) ? 0u : a.dspan[0] + ((uint32_t)(a.H[1] - 1) * (uint32_t)a.ds_h[0] + (uint32_t)a.W[1]) * 4u;
    const uint32_t abytes = (a.dbg & 1) ? 0u : ((uint32_t)(a.H[NLEV] - 1) * (uint32_t)a.as_h + (uint32_t)a.W[NLEV]) * 4u;
    const rsrc_t rd = pyr_rsrc(a.det[0] + (int64_t)img * a.ds_b[0], dbytes);
    const rsrc_t ra = pyr_rsrc(a.approx + (int64_t)img * a.as_b, NLEV == 1 ? abytes : 0u);
    const uint32_t o0 = a.doff[0][0], o1 = a.doff[0][1], o2 = a.doff[0][2];
    PyrSt16 gd, ga;  // detail planes / approximation plane (one level only)
    const uint32_t dpitch = (uint32_t)a.ds_h[0] * 4u, apitch = (uint32_t)a.as_h * 4u;
    if constexpr (st16) {
      gd.set(lane, NC1 * 64 * widx, pB[1]);  // (one column group: the lane grid starts at column 0)
      if constexpr (NLEV == 1) ga = gd;
    }

    PyrAcc<L, NC1> acc;
    acc.clear();
    int bi = 0;  // staging sub-buffer of the next sub-step
    constexpr int NW2 = HP + 1;  // 8-byte pieces of a window (L + 2 samples; 2 o1 - g0 is a multiple of 4)
    auto load_win = [&](const unsigned char* row, f2 (&w)[NW2]) { pyr_load_win2<L>(row, w); };
    // horizontal pass of the two rows of a pair, interleaved (2 NC1 independent chains)
    auto h_pair = [&](const f2 (&wa)[NW2], const f2 (&wb)[NW2], f2 (&ha)[NC1], f2 (&hb)[NC1]) {
#pragma unroll
      for (int k = 0; k < HP; ++k) {
#pragma unroll
        for (int c = 0; c < NC1; ++c) {
          if (k == 0) {
            ha[c] = vmul_lo(tap[L - 1], wa[c]);
            hb[c] = vmul_lo(tap[L - 1], wb[c]);
          } else {
            vfma_lo(ha[c], tap[L - 1 - 2 * k], wa[c + k]);
            vfma_lo(hb[c], tap[L - 1 - 2 * k], wb[c + k]);
          }
        }
#pragma unroll
        for (int c = 0; c < NC1; ++c) {
          vfma_hi(ha[c], tap[L - 2 - 2 * k], wa[c + k]);
          vfma_hi(hb[c], tap[L - 2 - 2 * k], wb[c + k]);
        }
      }
    };

    auto step1 = [&](auto sm_tag, int s) {
      constexpr int SM = decltype(sm_tag)::value;
      pyr_static_for<2>([&](auto half_tag) {
        constexpr int half = decltype(half_tag)::value;
        pyr_barrier<PROF>(waited);  // the loader has seen this sub-step land
        if (s < nsteps1) {
          unsigned char* sb = stage + bi * (kPyrSub * a.pitch0);
          bi = bi + 1 == a.nbuf ? 0 : bi + 1;
          if (f_any) {
            const float v = zero_mode ? 0.f : *reinterpret_cast<const float*>(sb + f_src);
            wave_lds_fence();
            if (f_on) *reinterpret_cast<float*>(sb + f_dst) = v;
            wave_lds_fence();
          }
          f2 w[kPyrSub][NW2];
#pragma unroll
          for (int kk = 0; kk < kPyrSub; ++kk) load_win(sb + kk * a.pitch0 + win, w[kk]);
          float k_ad[2], k_da[2], k_dd[2], k_aa[2];  // (16-byte stores: the first row of the half step waits for the second)
          pyr_static_for<kPyrSub / 2>([&](auto jj_tag) {
            constexpr int jj = decltype(jj_tag)::value;
            constexpr int j = 2 * half + jj;      // pair of the step
            constexpr int R = (4 * SM + j) % HP;  // its index modulo L/2
            f2 ha[NC1], hb[NC1];
            h_pair(w[2 * jj], w[2 * jj + 1], ha, hb);
#ifdef MIFWT_PYR_EXP
            if (!(a.dbg & 64)) acc.template feed<0, R>(tap, ha);
            if (!(a.dbg & (64 | 8))) acc.template feed<1, R>(tap, hb);
#else
            acc.template feed<0, R>(tap, ha);
            acc.template feed<1, R>(tap, hb);
#endif
            const int i = rA[1] + 4 * s + j - (HP - 1);  // the level-1 row it completes
            const f2 (&lo)[NC1] = acc.lo[PyrAcc<L, NC1>::done(R)];
            const f2 (&hi)[NC1] = acc.hi[PyrAcc<L, NC1>::done(R)];
            if constexpr (NLEV >= 2) {
              unsigned char* rr = ring1 + ((4 * s + j) & (kPyrRing - 1)) * a.pitch1;
              const bool mine = i < lim[1];  // (handover: rows from lim[1] on are put into the ring by the exchange wave; float 0 is never read)
#pragma unroll
              for (int k = 0; k < NC1; ++k) *reinterpret_cast<float*>(rr + (mine ? rw[k] : 0u)) = lo[k].x;
            }
            if constexpr (st16) {
              if constexpr (jj == 0) {
                k_ad[0] = hi[0].x, k_ad[1] = hi[1].x, k_da[0] = lo[0].y, k_da[1] = lo[1].y, k_dd[0] = hi[0].y, k_dd[1] = hi[1].y;
                if constexpr (NLEV == 1) k_aa[0] = lo[0].x, k_aa[1] = lo[1].x;
              } else {
                uint32_t v4, dx, so;
                bool mine;
                gd.rows(lane, i - 1, oA[1], oB[1], dpitch, v4, dx, mine, so);
                gd.band(lane, v4, dx, mine, k_ad[0], k_ad[1], hi[0].x, hi[1].x, rd, so + o0);
                gd.band(lane, v4, dx, mine, k_da[0], k_da[1], lo[0].y, lo[1].y, rd, so + o1);
                gd.band(lane, v4, dx, mine, k_dd[0], k_dd[1], hi[0].y, hi[1].y, rd, so + o2);
                if constexpr (NLEV == 1) {
                  ga.rows(lane, i - 1, oA[1], oB[1], apitch, v4, dx, mine, so);
                  ga.band(lane, v4, dx, mine, k_aa[0], k_aa[1], lo[0].x, lo[1].x, ra, so);
                }
              }
            } else {
              const bool own = i >= oA[1] && i < oB[1];
              const uint32_t v2 = own ? sv2 : kPyrOob;
              const uint32_t so = own ? (uint32_t)i * (uint32_t)a.ds_h[0] * 4u : 0u;
              __builtin_amdgcn_raw_buffer_store_b64((f2){hi[0].x, hi[1].x}, rd, v2, so + o0, MIFWT_ST_AUX);
              __builtin_amdgcn_raw_buffer_store_b64((f2){lo[0].y, lo[1].y}, rd, v2, so + o1, MIFWT_ST_AUX);
              __builtin_amdgcn_raw_buffer_store_b64((f2){hi[0].y, hi[1].y}, rd, v2, so + o2, MIFWT_ST_AUX);
              if (rag) {
                const uint32_t v1 = own ? sv1 : kPyrOob;
                pyr_store1(hi[0].x, rd, v1, so + o0);
                pyr_store1(lo[0].y, rd, v1, so + o1);
                pyr_store1(hi[0].y, rd, v1, so + o2);
              }
              if constexpr (NLEV == 1) {
                const uint32_t sa = own ? (uint32_t)i * (uint32_t)a.as_h * 4u : 0u;
                __builtin_amdgcn_raw_buffer_store_b64((f2){lo[0].x, lo[1].x}, ra, v2, sa, MIFWT_ST_AUX);
                if (rag) pyr_store1(lo[0].x, ra, own ? sv1 : kPyrOob, sa);
              }
            }
          });
        }
      });
    };
    int sm = 0;
#pragma unroll 1
    for (int s = 0; s < nsteps; ++s) {
      if constexpr (HP == 3) {
        pyr_dispatch<3>(sm, [&](auto t) { step1(t, s); });
        sm = sm == 2 ? 0 : sm + 1;
      } else {
        step1(std::integral_constant<int, 0>{}, s);
      }
    }
    prof_out();
    return;
  }

  // =====================================================================================================================
  // level-2 wave: two columns per lane, rows from ring 1
  if constexpr (NLEV >= 2) {
    if (role == kRoleL2) {
      const int gmax = (cB[2] - cA[2] + 1) / 2 - 1;
      constexpr bool st16 = ST16;  // (as at level 1)
      const int glane = 64 * widx + (st16 ? 2 * (lane & 31) + (lane >> 5) : lane);
      const int G = min(glane, gmax);
      const bool real = glane <= gmax;
      const int c0 = cA[2] + 2 * G;  // (pA2 - cA2 is even: a lane lies inside or outside the owned range)
      const uint32_t win = 4u * (uint32_t)(kPyrPad + sh1 - HL + 2 * c0 - cA[1]);
      const bool full = real && c0 >= pA[2] && c0 + 1 < pB[2];
      const bool rag1 = real && c0 >= pA[2] && c0 + 1 == pB[2];
      const uint32_t sv2 = full ? 4u * (uint32_t)c0 : kPyrOob, sv1 = rag1 ? 4u * (uint32_t)c0 : kPyrOob;
      const bool rag = __builtin_amdgcn_readfirstlane(__builtin_amdgcn_ballot_w64(rag1) != 0);
      uint32_t rw[2];
#pragma unroll
      for (int k = 0; k < 2; ++k) rw[k] = (real && c0 + k < cB[2]) ? 4u * (uint32_t)(kPyrPad + c0 + k - cA[2]) : 0u;
      // ring-1 pad fill by the waves whose windows reach the pads: lane -> (row r of the step's four, pad column)
      uint32_t f_src = 0, f_dst = 0;
      bool f_on = false;
      const int f_r = lane / NP;
      {
        const int wlo = 2 * (cA[2] + 2 * min(64 * widx, gmax)) - HL, whi = 2 * (cA[2] + 2 * min(64 * widx + 63, gmax) + 1) + 1;
        const int p = lane - f_r * NP;
        const bool left = p < HL;
        const int e = left ? p - HL : a.W[1] + (p - HL);
        if (f_r < 4 && !zero_mode && (left ? wlo < 0 : whi >= a.W[1])) {
          f_on = true;
          f_src = 4u * (uint32_t)(kPyrPad + sh1 + fold(e, a.W[1]) - cA[1]);
          f_dst = 4u * (uint32_t)(kPyrPad + sh1 + e - cA[1]);
        }
      }
      const bool f_any = __builtin_amdgcn_readfirstlane(__builtin_amdgcn_ballot_w64(f_on) != 0) && !(a.dbg & 32768);
      const uint32_t dbytes = (a.dbg & 1) ? 0u : a.dspan[1] + ((uint32_t)(a.H[2] - 1) * (uint32_t)a.ds_h[1] + (uint32_t)a.W[2]) * 4u;
      const uint32_t abytes = (a.dbg & 1) ? 0u : ((uint32_t)(a.H[NLEV] - 1) * (uint32_t)a.as_h + (uint32_t)a.W[NLEV]) * 4u;
      const rsrc_t rd = pyr_rsrc(a.det[1] + (int64_t)img * a.ds_b[1], dbytes);
      const rsrc_t ra = pyr_rsrc(a.approx + (int64_t)img * a.as_b, NLEV == 2 ? abytes : 0u);
      const uint32_t o0 = a.doff[1][0], o1 = a.doff[1][1], o2 = a.doff[1][2];
      PyrSt16 gd, ga;
      const uint32_t dpitch = (uint32_t)a.ds_h[1] * 4u, apitch = (uint32_t)a.as_h * 4u;
      if constexpr (st16) {
        gd.set(lane, 128 * widx, pB[2]);
        if constexpr (NLEV == 2) ga = gd;
      }
      const int ro1 = HP - 1 - rA[1];
      const int E1 = 2 * rA[2] - HL;
      PyrAcc<L, 2> acc;
      acc.clear();
      constexpr int NW2 = HP + 1;
      auto load_win = [&](const unsigned char* row, f2 (&w)[NW2]) { pyr_load_win2<L>(row, w); };
      auto h_pair = [&](const f2 (&wa)[NW2], const f2 (&wb)[NW2], f2 (&ha)[2], f2 (&hb)[2]) {
#pragma unroll
        for (int k = 0; k < HP; ++k) {
#pragma unroll
          for (int c = 0; c < 2; ++c) {
            if (k == 0) {
              ha[c] = vmul_lo(tap[L - 1], wa[c]);
              hb[c] = vmul_lo(tap[L - 1], wb[c]);
            } else {
              vfma_lo(ha[c], tap[L - 1 - 2 * k], wa[c + k]);
              vfma_lo(hb[c], tap[L - 1 - 2 * k], wb[c + k]);
            }
          }
#pragma unroll
          for (int c = 0; c < 2; ++c) {
            vfma_hi(ha[c], tap[L - 2 - 2 * k], wa[c + k]);
            vfma_hi(hb[c], tap[L - 2 - 2 * k], wb[c + k]);
          }
        }
      };
      if ((a.exp & 3) == 1) __builtin_amdgcn_s_setprio(1);
      if ((a.exp & 3) == 2) __builtin_amdgcn_s_setprio(2);
      if ((a.exp & 3) == 3) __builtin_amdgcn_s_setprio(3);
      int ph = 0;  // pair index modulo L/2 of the next block
      // Round 5: the step's two row pairs in the step's two HALVES, one each (a.l2split).  Until then a level-2 wave did a whole step's
      // work (two pairs, ~330 instructions) behind the step's second barrier and sat out the first half: per-wave clocks had it waiting
      // for about half of its life, i.e. never in its own half — it was what the second half of every step waited for (a level-1 wave
      // needs ~290 instructions per half).  The rows pair 0 reads (pair indices up to 4 (s - D2) + L/2 of level 1) are complete one
      // barrier earlier than those of pair 1 wherever 4 D2 >= L/2 + 1, which the lags guarantee — except where pair 0 reads MIRRORED rows
      // at the top of the plane (the first step of a top segment): that pair then runs with pair 1, as before.  Bit-identical.
      if constexpr (!ST16) {
        if (a.l2split) {
          auto do_pair = [&](auto r_tag, auto jj_tag, int s, const uint32_t (&so_r)[4]) {
            constexpr int R = decltype(r_tag)::value, jj = decltype(jj_tag)::value;
            f2 wa[NW2], wb[NW2];
            load_win(ring1 + so_r[2 * jj] + win, wa);
            load_win(ring1 + so_r[2 * jj + 1] + win, wb);
            f2 ha[2], hb[2];
            h_pair(wa, wb, ha, hb);
            acc.template feed<0, R>(tap, ha);
            acc.template feed<1, R>(tap, hb);
            const int p = 2 * (s - D2) + jj;
            const int i = rA[2] + p - (HP - 1);
            const f2 (&lo)[2] = acc.lo[PyrAcc<L, 2>::done(R)];
            const f2 (&hi)[2] = acc.hi[PyrAcc<L, 2>::done(R)];
            if constexpr (NLEV >= 3) {
              unsigned char* rr = ring2 + (p & (kPyrRing - 1)) * a.pitch2;
              const bool mine = i < lim[2];
              *reinterpret_cast<float*>(rr + (mine ? rw[0] : 0u)) = lo[0].x;
              *reinterpret_cast<float*>(rr + (mine ? rw[1] : 0u)) = lo[1].x;
            }
            const bool own = i >= oA[2] && i < oB[2];
            const uint32_t v2 = own ? sv2 : kPyrOob;
            const uint32_t so = own ? (uint32_t)i * (uint32_t)a.ds_h[1] * 4u : 0u;
            __builtin_amdgcn_raw_buffer_store_b64((f2){hi[0].x, hi[1].x}, rd, v2, so + o0, MIFWT_ST_AUX);
            __builtin_amdgcn_raw_buffer_store_b64((f2){lo[0].y, lo[1].y}, rd, v2, so + o1, MIFWT_ST_AUX);
            __builtin_amdgcn_raw_buffer_store_b64((f2){hi[0].y, hi[1].y}, rd, v2, so + o2, MIFWT_ST_AUX);
            if (rag) {
              const uint32_t v1 = own ? sv1 : kPyrOob;
              pyr_store1(hi[0].x, rd, v1, so + o0);
              pyr_store1(lo[0].y, rd, v1, so + o1);
              pyr_store1(hi[0].y, rd, v1, so + o2);
            }
            if constexpr (NLEV == 2) {
              const uint32_t sa = own ? (uint32_t)i * (uint32_t)a.as_h * 4u : 0u;
              __builtin_amdgcn_raw_buffer_store_b64((f2){lo[0].x, lo[1].x}, ra, v2, sa, MIFWT_ST_AUX);
              if (rag) pyr_store1(lo[0].x, ra, own ? sv1 : kPyrOob, sa);
            }
          };
          auto half = [&](auto jj_tag, int s, const uint32_t (&so_r)[4]) {
            constexpr int jj = decltype(jj_tag)::value;
            if (f_any) {  // the pads of the pair's two rows
              const uint32_t fo = f_r == 0 ? so_r[0] : (f_r == 1 ? so_r[1] : (f_r == 2 ? so_r[2] : so_r[3]));
              const float v = *reinterpret_cast<const float*>(ring1 + fo + f_src);
              wave_lds_fence();
              if (f_on && (f_r >> 1) == jj) *reinterpret_cast<float*>(ring1 + fo + f_dst) = v;
              wave_lds_fence();
            }
            pyr_dispatch<HP>((ph + jj) % HP, [&](auto r_tag) { do_pair(r_tag, jj_tag, s, so_r); });
          };
#pragma unroll 1
          for (int s = 0; s < nsteps; ++s) {
            pyr_barrier<PROF>(waited);
            const bool act = s >= D2 && 2 * (s - D2) < npair2 && !(a.dbg & 4);
            uint32_t so_r[4] = {0u, 0u, 0u, 0u};
            bool early = false;
            if (act) {
              int qmax0 = 0;  // the newest level-1 pair index pair 0 reads
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const int e = E1 + 4 * (s - D2) + r;
                const bool dead = zero_mode && (unsigned)e >= (unsigned)a.H[1];
                const int q = fold(e, a.H[1]) + ro1;
                so_r[r] = (uint32_t)((dead ? kPyrRing : (q & (kPyrRing - 1))) * a.pitch1);
                if (r < 2 && !dead) qmax0 = max(qmax0, q);
              }
              early = qmax0 <= 4 * s - 1;  // (written in an earlier step: complete behind this step's first barrier)
              if (early) half(std::integral_constant<int, 0>{}, s, so_r);
            }
            pyr_barrier<PROF>(waited);
            if (act) {
              if (!early) half(std::integral_constant<int, 0>{}, s, so_r);
              half(std::integral_constant<int, 1>{}, s, so_r);
              ph = (ph + 2) % HP;
            }
          }
          prof_out();
          return;
        }
      }
#pragma unroll 1
      for (int s = 0; s < nsteps; ++s) {
        pyr_barrier<PROF>(waited);
        pyr_barrier<PROF>(waited);  // level 1's first two rows of this step (and everything before) are in ring 1
        if (s >= D2 && 2 * (s - D2) < npair2 && !(a.dbg & 4)) {
          uint32_t so_r[4];  // ring-1 byte offsets of the step's four rows
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int e = E1 + 4 * (s - D2) + r;
            const bool dead = zero_mode && (unsigned)e >= (unsigned)a.H[1];
            so_r[r] = (uint32_t)((dead ? kPyrRing : ((fold(e, a.H[1]) + ro1) & (kPyrRing - 1))) * a.pitch1);
          }
          if (f_any) {
            const uint32_t fo = f_r == 0 ? so_r[0] : (f_r == 1 ? so_r[1] : (f_r == 2 ? so_r[2] : so_r[3]));
            const float v = *reinterpret_cast<const float*>(ring1 + fo + f_src);
            wave_lds_fence();
            if (f_on) *reinterpret_cast<float*>(ring1 + fo + f_dst) = v;
            wave_lds_fence();
          }
          pyr_dispatch<HP>(ph, [&](auto r0_tag) {
            constexpr int R0 = decltype(r0_tag)::value;  // (2 (s - D2)) mod L/2
            // (16-byte stores of 8-tap filters: the first row of a step waits in registers for the second — the windows of a
            // step's two row pairs are requested pair by pair then, the wave has no registers for all four next to them)
            constexpr bool kSplit = st16 && L >= 8;
            f2 w[4][NW2];
            if constexpr (!kSplit) {
#pragma unroll
              for (int r = 0; r < 4; ++r) load_win(ring1 + so_r[r] + win, w[r]);
            }
            float k_ad[2], k_da[2], k_dd[2], k_aa[2];
            pyr_static_for<2>([&](auto jj_tag) {
              constexpr int jj = decltype(jj_tag)::value;
              constexpr int R = (R0 + jj) % HP;
              if constexpr (kSplit) {
                load_win(ring1 + so_r[2 * jj] + win, w[2 * jj]);
                load_win(ring1 + so_r[2 * jj + 1] + win, w[2 * jj + 1]);
              }
              f2 ha[2], hb[2];
              h_pair(w[2 * jj], w[2 * jj + 1], ha, hb);
              acc.template feed<0, R>(tap, ha);
              acc.template feed<1, R>(tap, hb);
              const int p = 2 * (s - D2) + jj;
              const int i = rA[2] + p - (HP - 1);
              const f2 (&lo)[2] = acc.lo[PyrAcc<L, 2>::done(R)];
              const f2 (&hi)[2] = acc.hi[PyrAcc<L, 2>::done(R)];
              if constexpr (NLEV >= 3) {
                unsigned char* rr = ring2 + (p & (kPyrRing - 1)) * a.pitch2;
                const bool mine = i < lim[2];
                *reinterpret_cast<float*>(rr + (mine ? rw[0] : 0u)) = lo[0].x;
                *reinterpret_cast<float*>(rr + (mine ? rw[1] : 0u)) = lo[1].x;
              }
              if constexpr (st16) {
                if constexpr (jj == 0) {
                  k_ad[0] = hi[0].x, k_ad[1] = hi[1].x, k_da[0] = lo[0].y, k_da[1] = lo[1].y, k_dd[0] = hi[0].y, k_dd[1] = hi[1].y;
                  if constexpr (NLEV == 2) k_aa[0] = lo[0].x, k_aa[1] = lo[1].x;
                } else {
                  uint32_t v4, dx, so;
                  bool mine;
                  gd.rows(lane, i - 1, oA[2], oB[2], dpitch, v4, dx, mine, so);
                  gd.band(lane, v4, dx, mine, k_ad[0], k_ad[1], hi[0].x, hi[1].x, rd, so + o0);
                  gd.band(lane, v4, dx, mine, k_da[0], k_da[1], lo[0].y, lo[1].y, rd, so + o1);
                  gd.band(lane, v4, dx, mine, k_dd[0], k_dd[1], hi[0].y, hi[1].y, rd, so + o2);
                  if constexpr (NLEV == 2) {
                    ga.rows(lane, i - 1, oA[2], oB[2], apitch, v4, dx, mine, so);
                    ga.band(lane, v4, dx, mine, k_aa[0], k_aa[1], lo[0].x, lo[1].x, ra, so);
                  }
                }
              } else {
                const bool own = i >= oA[2] && i < oB[2];
                const uint32_t v2 = own ? sv2 : kPyrOob;
                const uint32_t so = own ? (uint32_t)i * (uint32_t)a.ds_h[1] * 4u : 0u;
                __builtin_amdgcn_raw_buffer_store_b64((f2){hi[0].x, hi[1].x}, rd, v2, so + o0, MIFWT_ST_AUX);
                __builtin_amdgcn_raw_buffer_store_b64((f2){lo[0].y, lo[1].y}, rd, v2, so + o1, MIFWT_ST_AUX);
                __builtin_amdgcn_raw_buffer_store_b64((f2){hi[0].y, hi[1].y}, rd, v2, so + o2, MIFWT_ST_AUX);
                if (rag) {
                  const uint32_t v1 = own ? sv1 : kPyrOob;
                  pyr_store1(hi[0].x, rd, v1, so + o0);
                  pyr_store1(lo[0].y, rd, v1, so + o1);
                  pyr_store1(hi[0].y, rd, v1, so + o2);
                }
                if constexpr (NLEV == 2) {
                  const uint32_t sa = own ? (uint32_t)i * (uint32_t)a.as_h * 4u : 0u;
                  __builtin_amdgcn_raw_buffer_store_b64((f2){lo[0].x, lo[1].x}, ra, v2, sa, MIFWT_ST_AUX);
                  if (rag) pyr_store1(lo[0].x, ra, own ? sv1 : kPyrOob, sa);
                }
              }
            });
          });
          ph = (ph + 2) % HP;
        }
      }
      prof_out();
      return;
    }
  }

  // =====================================================================================================================
  // level-3 wave: one column per lane, rows from ring 2
  if constexpr (NLEV >= 3) {
    const int gmax = cB[3] - cA[3] - 1;
    const int G = min(64 * widx + lane, gmax);
    const bool real = 64 * widx + lane <= gmax;
    const int c = cA[3] + G;
    const uint32_t win = 4u * (uint32_t)(kPyrPad - HL + 2 * c - cA[2]);
    const uint32_t sv = real && c >= pA[3] && c < pB[3] ? 4u * (uint32_t)c : kPyrOob;
    uint32_t f_src = 0, f_dst = 0;
    bool f_on = false;
    const int f_r = lane / NP;
    {
      const int wlo = 2 * (cA[3] + min(64 * widx, gmax)) - HL, whi = 2 * (cA[3] + min(64 * widx + 63, gmax)) + 1;
      const int p = lane - f_r * NP;
      const bool left = p < HL;
      const int e = left ? p - HL : a.W[2] + (p - HL);
      if (f_r < 2 && !zero_mode && (left ? wlo < 0 : whi >= a.W[2])) {
        f_on = true;
        f_src = 4u * (uint32_t)(kPyrPad + fold(e, a.W[2]) - cA[2]);
        f_dst = 4u * (uint32_t)(kPyrPad + e - cA[2]);
      }
    }
    const bool f_any = __builtin_amdgcn_readfirstlane(__builtin_amdgcn_ballot_w64(f_on) != 0) && !(a.dbg & 32768);
    const uint32_t dbytes = (a.dbg & 1) ? 0u : a.dspan[2] + ((uint32_t)(a.H[3] - 1) * (uint32_t)a.ds_h[2] + (uint32_t)a.W[3]) * 4u;
    const uint32_t abytes = (a.dbg & 1) ? 0u : ((uint32_t)(a.H[3] - 1) * (uint32_t)a.as_h + (uint32_t)a.W[3]) * 4u;
    const rsrc_t rd = pyr_rsrc(a.det[2] + (int64_t)img * a.ds_b[2], dbytes);
    const rsrc_t ra = pyr_rsrc(a.approx + (int64_t)img * a.as_b, abytes);
    const uint32_t o0 = a.doff[2][0], o1 = a.doff[2][1], o2 = a.doff[2][2];
    const int ro2 = HP - 1 - rA[2];
    const int E2 = 2 * rA[3] - HL;
    PyrAcc<L, 1> acc;
    acc.clear();
    auto load_win = [&](const unsigned char* row, f2 (&w)[HP]) {
#pragma unroll
      for (int k = 0; k < HP; ++k) w[k] = *reinterpret_cast<const f2*>(row + 8 * k);
    };
    auto h_pair = [&](const f2 (&wa)[HP], const f2 (&wb)[HP], f2 (&ha)[1], f2 (&hb)[1]) {
#pragma unroll
      for (int k = 0; k < HP; ++k) {
        if (k == 0) {
          ha[0] = vmul_lo(tap[L - 1], wa[0]);
          hb[0] = vmul_lo(tap[L - 1], wb[0]);
        } else {
          vfma_lo(ha[0], tap[L - 1 - 2 * k], wa[k]);
          vfma_lo(hb[0], tap[L - 1 - 2 * k], wb[k]);
        }
        vfma_hi(ha[0], tap[L - 2 - 2 * k], wa[k]);
        vfma_hi(hb[0], tap[L - 2 - 2 * k], wb[k]);
      }
    };
    if (((a.exp >> 2) & 3) == 1) __builtin_amdgcn_s_setprio(1);
    if (((a.exp >> 2) & 3) == 2) __builtin_amdgcn_s_setprio(2);
    int ph = 0;
#pragma unroll 1
    for (int s = 0; s < nsteps; ++s) {
      pyr_barrier<PROF>(waited);
      if (s >= D3 && s - D3 < npair3 && !(a.dbg & 4)) {
        uint32_t so_r[2];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          const int e = E2 + 2 * (s - D3) + r;
          const bool dead = zero_mode && (unsigned)e >= (unsigned)a.H[2];
          so_r[r] = (uint32_t)((dead ? kPyrRing : ((fold(e, a.H[2]) + ro2) & (kPyrRing - 1))) * a.pitch2);
        }
        if (f_any) {
          const uint32_t fo = f_r == 0 ? so_r[0] : so_r[1];
          const float v = *reinterpret_cast<const float*>(ring2 + fo + f_src);
          wave_lds_fence();
          if (f_on) *reinterpret_cast<float*>(ring2 + fo + f_dst) = v;
          wave_lds_fence();
        }
        pyr_dispatch<HP>(ph, [&](auto r_tag) {
          constexpr int R = decltype(r_tag)::value;  // (s - D3) mod L/2
          f2 w[2][HP];
          load_win(ring2 + so_r[0] + win, w[0]);
          load_win(ring2 + so_r[1] + win, w[1]);
          f2 ha[1], hb[1];
          h_pair(w[0], w[1], ha, hb);
          acc.template feed<0, R>(tap, ha);
          acc.template feed<1, R>(tap, hb);
          const int i = rA[3] + (s - D3) - (HP - 1);
          const f2 lo = acc.lo[PyrAcc<L, 1>::done(R)][0], hi = acc.hi[PyrAcc<L, 1>::done(R)][0];
          const bool own = i >= oA[3] && i < oB[3];
          const uint32_t v = own ? sv : kPyrOob;
          const uint32_t so = own ? (uint32_t)i * (uint32_t)a.ds_h[2] * 4u : 0u;
          pyr_store1(hi.x, rd, v, so + o0);
          pyr_store1(lo.y, rd, v, so + o1);
          pyr_store1(hi.y, rd, v, so + o2);
          pyr_store1(lo.x, ra, v, own ? (uint32_t)i * (uint32_t)a.as_h * 4u : 0u);
        });
        ph = ph + 1 == HP ? 0 : ph + 1;
      }
      pyr_barrier<PROF>(waited);
    }
    prof_out();
  }
}

// ---- host side ------------------------------------------------------------------------------------------------------------
struct PyrPlan {
  int compact;  // eight-wave workgroups, two per CU (MIFWT_OPT_DEBUG bit 11; see PyrArgs)
  int ngroups, nseg, seg_rows, seg0_rows, cpg0, cpg, nchunks, nbuf, pitch0, pitch1, pitch2, nl1, nl2, nl3, lds;
  int handover, xrow_stride;  // segments hand their first approximation rows over (needs a workspace); floats per (image, segment)
  size_t ws_bytes;
};

// columns of level NLEV a group may own: the level-1 / 2 / 3 lane grids hold 4 x 192 / 3 x 128 / 3 x 64 columns, a staged row
// at most kPyrMaxChunks x 256 level-0 columns; interior groups recompute (L - 2) halo columns per level on their left
static int pyr_group_cols(int L, int nlev, bool first, bool compact) {
  const int nc1 = 2;
  const int HL = first ? 0 : L - 2;
  int best = 0;
  const int w1 = compact ? 3 : 6, w2 = compact ? 2 : 3, w3 = compact ? 2 : 3, maxch = compact ? 3 : kPyrMaxChunks;
  for (int n = 1; n <= 4 * 192; ++n) {
    int m = n;  // columns computed at the level below, walking down to level 1
    bool ok = true;
    for (int l = nlev; l >= 1 && ok; --l) {
      const int cap = l == 1 ? w1 * 128 - nc1 : (l == 2 ? w2 * 128 : w3 * 64);
      ok = m <= cap;
      if (l > 1) m = 2 * m + HL;
    }
    // m = level-1 columns now; level-0 span incl. the 16-byte alignment slack
    if (ok && 2 * m + HL + 3 <= maxch * 256) best = n;
  }
  return best;
}

static bool pyr_plan(int nlev, const mifwt_level_desc* const* d, PyrPlan* p) {
  const int nc1 = 2;  // columns per level-1 lane
  const int L = d[0]->filt_len, HL = L - 2;
  const int WN = (int)d[nlev - 1]->coef_extent[1], HN = (int)d[nlev - 1]->coef_extent[0];
  const int min_cols = HL + 2;
  if (WN < min_cols || HN < 2 * (HL + 2)) return false;
  p->compact = (g_options[MIFWT_OPT_DEBUG] & 2048) ? 1 : 0;
  p->cpg0 = pyr_group_cols(L, nlev, true, p->compact);
  p->cpg = pyr_group_cols(L, nlev, false, p->compact);
  if (nlev == 1) {  // level-1 lanes hold column pairs that start on even columns
    p->cpg0 &= ~1;
    p->cpg &= ~1;
  }
  if (p->cpg < min_cols) return false;
  if (WN <= p->cpg0) {
    p->ngroups = 1;
  } else {
    // as few groups as the lane grids allow, of EQUAL width (groups of the maximum width plus a narrow last one left a
    // quarter of the workgroups with a third of the work: 4096-column planes ran 28 % slower per byte than 1024-column ones)
    p->ngroups = 1 + (WN - p->cpg0 + p->cpg - 1) / p->cpg;
    int base = (WN + p->ngroups - 1) / p->ngroups;
    if (nlev == 1) base = (base + 1) & ~1;
    if (base < p->cpg) {
      p->cpg0 = base;
      p->cpg = base;
    }
    const int last = WN - p->cpg0 - (p->ngroups - 2) * p->cpg;
    if (last < min_cols) p->cpg0 -= min_cols - last;
  }
  // the widest group decides the row pitches and the wave counts (same recurrences as the kernel)
  int n[4] = {0, 0, 0, 0}, body = 0;
  const int W0 = (int)d[0]->sig_extent[1];
  for (int g = 0; g < p->ngroups; ++g) {
    int cA = g == 0 ? 0 : p->cpg0 + (g - 1) * p->cpg, cB = g == p->ngroups - 1 ? WN : std::min(WN, p->cpg0 + g * p->cpg);
    int pA = cA;
    for (int l = nlev; l >= 1; --l) {
      n[l] = std::max(n[l], cB - cA + (l == 1 ? 2 : 0));  // (+ up to two columns of lane-grid alignment at level 1)
      if (l > 1) {
        const int Wl = (int)d[l - 2]->coef_extent[1];
        pA = 2 * pA;
        cA = std::max(0, 2 * cA - HL);
        cB = std::min(Wl, 2 * cB);
      }
    }
    const int o1 = pA - nc1 * ((pA - cA + nc1 - 1) / nc1);  // first column of the level-1 lane grid
    if (o1 < 0) return false;
    const int g0 = std::max(0, 2 * o1 - HL) & ~3;
    body = std::max(body, std::min(W0, 2 * cB) - g0);
  }
  p->nl1 = (n[1] + 64 * nc1 - 1) / (64 * nc1);
  p->nl2 = nlev >= 2 ? (n[2] + 127) / 128 : 0;
  p->nl3 = nlev >= 3 ? (n[3] + 63) / 64 : 0;
  if (p->nl1 > (p->compact ? 3 : 6) || p->nl2 > (p->compact ? 2 : 3) || p->nl3 > (p->compact ? 2 : 3)) return false;
  p->nchunks = (body + 255) / 256;
  if (p->nchunks < 1 || p->nchunks > (p->compact ? 3 : kPyrMaxChunks)) return false;
  p->pitch0 = (kPyrPad + 256 * p->nchunks + 8) * 4;
  p->pitch1 = nlev >= 2 ? ((kPyrPad + n[1] + HL + 8 + 3) & ~3) * 4 : 0;
  p->pitch2 = nlev >= 3 ? ((kPyrPad + n[2] + HL + 8 + 3) & ~3) * 4 : 0;
  const int rings = (kPyrRing + 1) * (p->pitch1 + p->pitch2);
  // as many staging sub-buffers as fit (the loaders run nbuf - 1 sub-steps ahead; a wave holds at most 63 requests in flight)
  p->nbuf = g_options[MIFWT_OPT_PREFETCH_PAIRS] > 1 ? std::min(8, g_options[MIFWT_OPT_PREFETCH_PAIRS]) : 4;  // (3 .. 6 measured alike on config 2)
  const int lds_cap = p->compact ? 80 * 1024 : 160 * 1024, per_loader = p->compact ? p->nchunks : (p->nchunks + 1) / 2;
  while (p->nbuf > 2 && (kPyrCtl + p->nbuf * kPyrSub * p->pitch0 + rings > lds_cap || (p->nbuf - 1) * kPyrSub * per_loader > 63)) --p->nbuf;
  p->lds = kPyrCtl + p->nbuf * kPyrSub * p->pitch0 + rings;
  if (p->lds > lds_cap) return false;
  // one workgroup per CU (the segment count below is chosen for that; two small workgroups on one CU leave others idle) — or, compact,
  // exactly two
  const int lds_used = p->lds;
  if (!p->compact && p->lds < 82 * 1024) p->lds = 82 * 1024;
  if (p->compact && p->lds < 54 * 1024) p->lds = 54 * 1024;  // (never three)
  int dev = 0, ncu = 256;
  if (hipGetDevice(&dev) == hipSuccess) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ncu = v;
  }
  const int64_t per_seg = d[0]->batch * p->ngroups;
  int nseg = g_options[MIFWT_OPT_PAIR_ROWS] > 0 ? (HN + g_options[MIFWT_OPT_PAIR_ROWS] - 1) / g_options[MIFWT_OPT_PAIR_ROWS]
                                                  : (int)(((p->compact ? 2 : 1) * ncu + per_seg / 2) / (per_seg > 0 ? per_seg : 1));
  const int max_seg = HN / 8 > 0 ? HN / 8 : 1;
  nseg = nseg < 1 ? 1 : (nseg > max_seg ? max_seg : nseg);
  p->seg_rows = (HN + nseg - 1) / nseg;
  p->nseg = (HN + p->seg_rows - 1) / p->seg_rows;
  if (p->nseg > 1 && HN - (p->nseg - 1) * p->seg_rows < 8) --p->nseg;
  p->seg0_rows = p->seg_rows;
  // Round 5: segments of equal TIME, not of equal rows.  Per-workgroup wall clocks (tools/pyr_clock.py, profiles/r05a_clock.txt) on
  // config 2 with 34 / 34 / 34 / 32 level-3 rows: the four segments of an image end 97.5 / 104.3 / 101.1 / 90.5 us after the launch
  // starts, and the launch lasts as long as its longest workgroup (+ 4.5 us until the next one's first wave).  A step in which the
  // level-1 waves run (prologue or owned rows) costs 2.5 us, a drain step (deep levels only, no memory traffic) 1.1 us — that fits the
  // top (35 + 9 steps), middle (40 + 2) and bottom (32 + 8) segments within 1 %.  So: first-segment rows and inner-segment rows are
  // searched around the equal split for the smallest modelled maximum (config 2: 34 / 32 / 32 / 36).
  if (p->nseg > 1 && g_options[MIFWT_OPT_PAIR_ROWS] <= 0 && !(g_options[MIFWT_OPT_DEBUG] & 4096)) {
    const int HP = L / 2;
    int Hl[4] = {0, 0, 0, 0};
    for (int l = 1; l <= nlev; ++l) Hl[l] = (int)d[l - 1]->coef_extent[0];
    auto seg_time = [&](int oA, int oB) {
      int rA[4], rB[4];
      rA[nlev] = oA;
      rB[nlev] = oB;
      for (int l = nlev - 1; l >= 1; --l) {
        rA[l] = std::max(0, 2 * rA[l + 1] - HL);
        rB[l] = std::min(Hl[l], 2 * rB[l + 1]);
      }
      const bool top = oA == 0;
      const int D2 = top ? pyr_lag2(L) : pyr_lag2_inner(L), D3 = top ? pyr_lag3(L) : pyr_lag3_inner(L);
      const int nsteps1 = (rB[1] - rA[1] + HP - 1 + 3) / 4;
      int nsteps = nsteps1;
      if (nlev >= 2) nsteps = std::max(nsteps, D2 + (rB[2] - rA[2] + HP - 1 + 1) / 2);
      if (nlev >= 3) nsteps = std::max(nsteps, D3 + rB[3] - rA[3] + HP - 1);
      return nsteps1 + 0.45 * (nsteps - nsteps1);
    };
    const int base = p->seg_rows, ns = p->nseg;
    double best = 1e30;
    int best0 = base, bestr = base;
    for (int r = std::max(8, base - 6); r <= base + 2; ++r)
      for (int a0 = std::max(8, base - 6); a0 <= base + 6; ++a0) {
        const int last = HN - a0 - (ns - 2) * r;
        if (last < 8 || (ns > 2 && a0 + (ns - 3) * r >= HN)) continue;
        double t = seg_time(0, a0);
        if (ns > 2) t = std::max(t, seg_time(a0, a0 + r));  // (the inner segments are alike)
        t = std::max(t, seg_time(a0 + (ns - 2) * r, HN));
        if (t < best - 1e-9) best = t, best0 = a0, bestr = r;
      }
    p->seg0_rows = best0;
    p->seg_rows = bestr;
  }
  // (experiments: explicit first / inner segment rows, tools/pyr_segs.py)
  if (p->nseg > 1 && g_options[MIFWT_OPT_PYR_SEG0_ROWS] > 0 && g_options[MIFWT_OPT_PYR_SEG_ROWS] > 0) {
    const int a0 = g_options[MIFWT_OPT_PYR_SEG0_ROWS], r = g_options[MIFWT_OPT_PYR_SEG_ROWS];
    if (a0 >= 8 && r >= 8 && HN - a0 - (p->nseg - 2) * r >= 8) p->seg0_rows = a0, p->seg_rows = r;
  }
  // (Round 4: a first segment SHORTER by the two level-3 rows its longer top-of-plane lags cost — 42 / 42 / 42 / 42 steps of 8 rows per
  // workgroup instead of 44 / 42 / 42 / 40 — measured the same, 104.4-106.5 against 104.4-105.1 us in three alternating runs,
  // profiles/r04t_segment_balance.txt: as with round 3's longer first segment, the launch does not wait for its longest workgroup.)
  // handover between the row segments of an image instead of prologues (kernel: exchange wave): two levels at least, more than one
  // segment, one column group, a spare wave (at most five level-1 waves), room in LDS for the rows taken over
  p->handover = 0;
  p->xrow_stride = 0;
  p->ws_bytes = 0;
  // MEASURED SLOWER on config 2 and therefore OFF unless MIFWT_OPT_DEBUG bit 8 asks for it: 108.0 against 102.5-104.9 us per launch with
  // 4.5 % less traffic and 12 % fewer level-1 rows per workgroup (profiles/r03k_handover.txt).  Without prologues every workgroup is in
  // its full read + write steps at the same time and then all of them drain their deep levels together (7 steps without memory
  // traffic); with prologues the segments are staggered by what they are — 5 read-only steps at the start of three workgroups in four.
  if (nlev >= 2 && p->nseg > 1 && p->ngroups == 1 && p->nl1 <= 5 && !p->compact && (g_options[MIFWT_OPT_DEBUG] & 256) && g_options[MIFWT_OPT_PAIR_ROWS] <= 0) {
    const int W1 = (int)d[0]->coef_extent[1], W2 = nlev >= 3 ? (int)d[1]->coef_extent[1] : 0;
    const int XM = HL + 1;
    const int xlds = XM * ((((W1 + 3) & ~3) * 4 + 16) + (nlev >= 3 ? ((W2 + 3) & ~3) * 4 + 16 : 0));
    // every segment must be long enough for the shifted ownership boundaries (2 (L - 2) rows of the last level is plenty)
    // the first segment runs with the longer lags of the plane's top (mirrored rows must exist before they are read): it gets that
    // many rows of the last level fewer, so that the workgroups of an image finish together
    const int dtop = nlev >= 3 ? pyr_lag3(L) - pyr_lag3_inner(L) : 2 * (pyr_lag2(L) - pyr_lag2_inner(L));
    const int y = (HN + dtop + p->nseg - 1) / p->nseg, x0 = y - dtop;
    if (lds_used + xlds <= 160 * 1024 && x0 >= 2 * HL + 4 && y >= 2 * HL + 4 && HN - x0 - (p->nseg - 2) * y >= 2 * HL + 4) {
      p->seg_rows = y;
      p->seg0_rows = x0;
      p->handover = 1;
      p->lds = std::max(p->lds, lds_used + xlds);
      p->xrow_stride = (XM * (W1 + W2) + 3) & ~3;
      const size_t nsegs = (size_t)d[0]->batch * p->nseg;
      p->ws_bytes = ((nsegs * 8 + 255) & ~size_t(255)) + nsegs * (size_t)p->xrow_stride * 4;
    }
  }
  // Every segment but the first streams a prologue of (2^nlev - 1) (L - 2) level-0 rows.  Giving the first one that many rows more
  // (config 2: 38 + 3 x 32 level-3 rows = 304 / 298 level-0 rows per workgroup, against 272 / 314 / 314 / 298 for equal segments)
  // was measured SLOWER in the whole kernel, twice: 104.2 against 100.8 us (profiles/r03e_seg0.txt; round 2 saw the same with the
  // arithmetic alone) — the kernel is bound by the memory system, not by its longest workgroup, and the early finishers of the top
  // segments hand their bandwidth to the rest.  Equal segments stay the default; MIFWT_OPT_DEBUG bit 7 switches the long first one on.
  if (p->nseg > 1 && g_options[MIFWT_OPT_PAIR_ROWS] <= 0 && (g_options[MIFWT_OPT_DEBUG] & 128)) {
    const double halo = ((1 << nlev) - 1) * (double)HL / (double)(1 << nlev);
    int y = (int)((HN - halo) / p->nseg + 0.5);
    int x = HN - (p->nseg - 1) * y;
    if (y >= 8 && x >= y) {
      p->seg_rows = y;
      p->seg0_rows = x;
    }
  }
  return true;
}

bool dwt2_fwd_pyr_supported(int nlev, const mifwt_level_desc* const* d) {
  if (nlev < 1 || nlev > 3 || g_options[MIFWT_OPT_PAIR_MODE] == 2 || g_options[MIFWT_OPT_PYRAMID_MODE] == 2) return false;
  const mifwt_level_desc* d0 = d[0];
  const int L = d0->filt_len;
  if (d0->ndim != 2 || d0->dtype != MIFWT_F32 || L < 2 || L > 8 || (L & 1)) return false;
  if (d0->mode == MIFWT_MODE_PERIODIC || d0->mode < 0 || d0->mode > MIFWT_MODE_SYMMETRIC) return false;
  if (d0->batch < 1 || d0->sig_stride[2] != 1) return false;
  // (rows of any length and alignment: the LDS-DMA engine takes 16 bytes per lane from 4-byte aligned addresses, tools/dma_probe.hip)
  const int64_t lim = int64_t(1) << 29;  // byte offsets inside one image stay below 2^31
  if (d0->sig_extent[0] * d0->sig_stride[1] >= lim) return false;
  for (int l = 0; l < nlev; ++l) {
    const mifwt_level_desc* dl = d[l];
    if (dl->ndim != 2 || dl->dtype != MIFWT_F32 || dl->filt_len != L || dl->mode != d0->mode || dl->batch != d0->batch) return false;
    if (dl->detail_stride[2] != 1 || dl->coef_extent[0] * dl->detail_stride[1] >= lim) return false;
    for (int ax = 0; ax < 2; ++ax) {
      const int64_t n = l == 0 ? d0->sig_extent[ax] : d[l - 1]->coef_extent[ax];
      if (dl->sig_extent[ax] != n || dl->coef_extent[ax] != (n + L - 1) / 2) return false;
      if (n < 2 * L) return false;  // single-fold boundary map, pads mirrored from inside the group
    }
  }
  const mifwt_level_desc* dn = d[nlev - 1];
  if (dn->approx_stride[2] != 1 || dn->coef_extent[0] * dn->approx_stride[1] >= lim) return false;
  PyrPlan p;
  if (!pyr_plan(nlev, d, &p)) return false;
  // Where it pays (tools/pyr_where.py, profiles/r03m_pyr_where.txt; round 2: tools/pyr_matrix.py, tools/pyr_big.py;
  // MIFWT_OPT_PYRAMID_MODE 1 overrides): planes of 448 .. ~2560 columns, i.e. one or two column groups.  A workgroup then reads whole
  // rows (or halves of them), one after the other.  Four column groups (4096 columns: 4 KB pieces 16 KB apart) ran at 0.44 of the HBM
  // peak against 0.65 for the per-level tile kernel; narrower planes leave most lanes of the level-2 / 3 waves idle (256^2: 75 against 55 us).
  if (g_options[MIFWT_OPT_PYRAMID_MODE] != 1 && (p.ngroups > (p.compact ? 4 : 2) || d0->sig_extent[1] < 448)) return false;
  return true;
}


template <int L, int NLEV>
static int launch_pyr(const mifwt_level_desc* const* d, const void* x, void* const* const* details, void* approx, const double* lo,
                       const double* hi, void* ws, size_t ws_bytes, unsigned long long nonce, hipStream_t stream) {
  PyrPlan p;
  if (!pyr_plan(NLEV, d, &p)) return MIFWT_ERR_UNSUPPORTED;
  PyrArgs<L, NLEV> a;
  a.x = static_cast<const float*>(x);
  a.xs_b = d[0]->sig_stride[0];
  a.xs_h = (int)d[0]->sig_stride[1];
  a.H[0] = (int)d[0]->sig_extent[0];
  a.W[0] = (int)d[0]->sig_extent[1];
  for (int l = 0; l < NLEV; ++l) {
    uintptr_t lo_p = reinterpret_cast<uintptr_t>(details[l][0]);
    for (int b = 1; b < 3; ++b) lo_p = std::min(lo_p, reinterpret_cast<uintptr_t>(details[l][b]));
    a.det[l] = reinterpret_cast<float*>(lo_p);
    a.dspan[l] = 0;
    for (int b = 0; b < 3; ++b) {
      const uintptr_t off = reinterpret_cast<uintptr_t>(details[l][b]) - lo_p;
      // one resource spans a level's three planes of an image: they must lie within 1 GiB of one another (they are the
      // planes of one level buffer in every caller of this library)
      if (off >= (uintptr_t(1) << 30) || (off & 3)) return MIFWT_ERR_UNSUPPORTED;
      a.doff[l][b] = (uint32_t)off;
      a.dspan[l] = std::max(a.dspan[l], (uint32_t)off);
    }
    a.ds_b[l] = d[l]->detail_stride[0];
    a.ds_h[l] = (int)d[l]->detail_stride[1];
    a.H[l + 1] = (int)d[l]->coef_extent[0];
    a.W[l + 1] = (int)d[l]->coef_extent[1];
  }
  a.approx = static_cast<float*>(approx);
  a.as_b = d[NLEV - 1]->approx_stride[0];
  a.as_h = (int)d[NLEV - 1]->approx_stride[1];
  a.ngroups = p.ngroups;
  a.nseg = p.nseg;
  a.seg_rows = p.seg_rows;
  a.seg0_rows = p.seg0_rows;
  a.cpg0 = p.cpg0;
  a.cpg = p.cpg;
  a.nchunks = p.nchunks;
  a.compact = p.compact;
  a.nbuf = p.nbuf;
  a.pitch0 = p.pitch0;
  a.pitch1 = p.pitch1;
  a.pitch2 = p.pitch2;
  a.nl1 = p.nl1;
  a.nl2 = p.nl2;
  a.nl3 = p.nl3;
  a.mode = d[0]->mode;
  a.dbg = g_options[MIFWT_OPT_DEBUG];
  a.prof = g_pyr_prof;
  const int64_t nwg_all = d[0]->batch * p.nseg * p.ngroups;
  a.xcd_map = (a.dbg & 32) && (nwg_all % 8 == 0) ? 1 : 0;
  // 16-byte stores (lane-pair exchange): one column group, and every plane the level's waves write has rows, images and bands on
  // 16-byte boundaries (the host layer pads the row pitch of these planes to a multiple of four floats; dense odd-width planes of
  // other callers keep the 8-byte stores).  MIFWT_OPT_DEBUG bit 9 switches them off (A/B runs).
  bool st16 = p.ngroups == 1 && !(a.dbg & 512) && !p.handover;
  {
    auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    for (int l = 0; l < NLEV && l < 2; ++l) {
      st16 = st16 && al16(a.det[l]) && a.ds_h[l] % 4 == 0 && a.ds_b[l] % 4 == 0;
      for (int b = 0; b < 3; ++b) st16 = st16 && (a.doff[l][b] & 15) == 0;
      if (l + 1 == NLEV) st16 = st16 && al16(a.approx) && a.as_h % 4 == 0 && a.as_b % 4 == 0;
    }
  }
  a.l2split = (g_options[MIFWT_OPT_DEBUG] & 8192) ? 0 : 1;
  a.exp = g_options[MIFWT_OPT_EXP];
  a.handover = 0;
  a.nonce = 0;
  a.flags = nullptr;
  a.xrows = nullptr;
  a.xrow_stride = 0;
  if (p.handover && ws && ws_bytes >= p.ws_bytes && !a.xcd_map) {
    const size_t nsegs = (size_t)d[0]->batch * p.nseg;
    a.handover = 1;
    a.l2split = 0;  // (the exchange wave places handed-over rows by the old schedule)
    a.nonce = nonce;
    a.flags = static_cast<unsigned long long*>(ws);
    a.xrows = reinterpret_cast<float*>(static_cast<unsigned char*>(ws) + ((nsegs * 8 + 255) & ~size_t(255)));
    a.xrow_stride = p.xrow_stride;
  }
  for (int m = 0; m < L; ++m) a.tap[m] = (f2){(float)lo[m], (float)hi[m]};
  const int64_t nwg = d[0]->batch * p.nseg * p.ngroups;
  if (nwg > INT32_MAX / 8) return MIFWT_ERR_UNSUPPORTED;
  // (the per-wave cycle profile of tools/pyr_prof.py exists for the three-level 8-tap kernel only)
  constexpr bool kCanProf = L == 8 && NLEV == 3;
  static DynLdsOnce lds_once, lds_once16, lds_once_prof;
  if (!lds_once.ensure(reinterpret_cast<const void*>(&dwt2_fwd_pyr_kernel<L, NLEV, false, false>), 160 * 1024)) return MIFWT_ERR_LAUNCH;
  if (!lds_once16.ensure(reinterpret_cast<const void*>(&dwt2_fwd_pyr_kernel<L, NLEV, false, true>), 160 * 1024)) return MIFWT_ERR_LAUNCH;
  if (kCanProf && !lds_once_prof.ensure(reinterpret_cast<const void*>(&dwt2_fwd_pyr_kernel<L, NLEV, kCanProf, false>), 160 * 1024))
    return MIFWT_ERR_LAUNCH;
  if (kCanProf && a.prof)
    hipLaunchKernelGGL((dwt2_fwd_pyr_kernel<L, NLEV, kCanProf, false>), dim3((unsigned)nwg), dim3(p.compact ? 512 : 64 * kPyrWaves), p.lds, stream, a);
  else if (st16) {
    count_launch(MIFWT_VARIANT_FWD_PYR_ST16);
    hipLaunchKernelGGL((dwt2_fwd_pyr_kernel<L, NLEV, false, true>), dim3((unsigned)nwg), dim3(p.compact ? 512 : 64 * kPyrWaves), p.lds, stream, a);
  } else {
    count_launch(MIFWT_VARIANT_FWD_PYR_ST8);
    hipLaunchKernelGGL((dwt2_fwd_pyr_kernel<L, NLEV, false, false>), dim3((unsigned)nwg), dim3(p.compact ? 512 : 64 * kPyrWaves), p.lds, stream, a);
  }
  return hipGetLastError() == hipSuccess ? MIFWT_OK : MIFWT_ERR_LAUNCH;
}

template <int L>
static int launch_pyr_l(int nlev, const mifwt_level_desc* const* d, const void* x, void* const* const* details, void* approx,
                         const double* lo, const double* hi, void* ws, size_t ws_bytes, unsigned long long nonce, hipStream_t stream) {
  switch (nlev) {
    case 1: return launch_pyr<L, 1>(d, x, details, approx, lo, hi, ws, ws_bytes, nonce, stream);
    case 2: return launch_pyr<L, 2>(d, x, details, approx, lo, hi, ws, ws_bytes, nonce, stream);
    case 3: return launch_pyr<L, 3>(d, x, details, approx, lo, hi, ws, ws_bytes, nonce, stream);
    default: return MIFWT_ERR_UNSUPPORTED;
  }
}

size_t dwt2_fwd_pyr_workspace(int nlev, const mifwt_level_desc* const* d) {
  if (!dwt2_fwd_pyr_supported(nlev, d)) return 0;
  PyrPlan p;
  return pyr_plan(nlev, d, &p) ? p.ws_bytes : 0;
}

int dwt2_fwd_pyr(int nlev, const mifwt_level_desc* const* d, const void* x, void* const* const* details, void* approx,
                  const double* lo, const double* hi, void* ws, size_t ws_bytes, unsigned long long nonce, hipStream_t stream) {
  if (!dwt2_fwd_pyr_supported(nlev, d)) return MIFWT_ERR_UNSUPPORTED;
  switch (d[0]->filt_len) {
    case 2: return launch_pyr_l<2>(nlev, d, x, details, approx, lo, hi, ws, ws_bytes, nonce, stream);
    case 4: return launch_pyr_l<4>(nlev, d, x, details, approx, lo, hi, ws, ws_bytes, nonce, stream);
    case 6: return launch_pyr_l<6>(nlev, d, x, details, approx, lo, hi, ws, ws_bytes, nonce, stream);
    case 8: return launch_pyr_l<8>(nlev, d, x, details, approx, lo, hi, ws, ws_bytes, nonce, stream);
    default: return MIFWT_ERR_UNSUPPORTED;
  }
}

}  // namespace mifwt
