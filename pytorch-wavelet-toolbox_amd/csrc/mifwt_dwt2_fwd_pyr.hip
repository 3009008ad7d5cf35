// mifwt_dwt2_fwd_pyr.hip — UP TO THREE consecutive 2-D analysis levels in one launch (gfx950), kernel id 16.
//
// Seam: NLEV trips of the reference's level loop (src/ptwt/conv_transform_2.py:142-149: _fwt_pad2 + F.conv2d(stride 2) +
// split); a pyramid returns only the detail bands of every level but the last (conv_transform_2.py:150-156), so the
// approximations in between never reach HBM here: they live in LDS rings.
//
// Shape of the work (every choice measured: tools/ubench.hip, tools/pyr_time.py, tools/pyr_prof.py, tools/pyr_clock.py;
// profiles/r02_* .. r05*):
//   * PERSISTENT workgroups (round 5): the launch has one workgroup per CU (and column group); the rows of the last level of all
//     images, laid end to end, are cut into one CHUNK per workgroup (PyrArgs::wg_start, cut by the host so that the chunks take equal
//     modelled TIME, not equal rows), and a workgroup runs the UNITS of its chunk — its part of one image each — one after the other.
//     A batch of 64 images of 1024^2 on 256 CUs is four units per image as before (rows 34 / 32 / 32 / 36 of 134); 65 images are 256
//     chunks of ~34 rows, most of them the end of one image and the start of the next (until round 4: 260 one-unit workgroups, the
//     last four of them alone on the chip: 1.7 x the time of 64 images);
//   * a unit is one row segment of one column GROUP of one image (the whole width of a plane up to 1280 columns; wider planes are
//     cut into groups that recompute L - 2 halo columns per level at their left seam); the workgroup runs one wave per role:
//     level-1 waves (two columns per lane, 128 columns per wave), level-2 waves (two per lane), level-3 waves (one per
//     lane), two LOADER waves, and (round 5) a TAIL wave for the last few columns of each level where they would leave a level
//     wave nearly empty.  Config 2: 4 + 2 + 2 level waves + tail + 2 loaders in the TWELVE-wave form (three waves per SIMD: 168
//     registers a lane; until round 4: 5 + 3 + 3 + 2 of sixteen).  The waves of a level tile the group's columns densely and
//     SHARE one staged level-0 row and one ring row per level (private 256-column strips with private halos, the first
//     design, requested every row 1.25 times and needed 5 + 5 + 5 waves: 4-9 % slower on every shape tried);
//   * rows STREAM through a wave: the vertical pass of every level keeps the L/2 outputs in flight in registers (rolling
//     accumulators with compile-time slot rotation, 2 packed FMAs per sample and band pair), nothing is re-read; level l+1
//     consumes the rows of level l from a 16-row LDS ring through the boundary index map (mirrored rows at the top / bottom
//     of the plane are ring rows), lagging by a fixed number of 8-row steps; the two rows of a pair are filtered interleaved
//     (four independent chains); a unit below the top of its image starts with a prologue of (2^NLEV - 1) (L - 2) input rows;
//   * the LOADER waves issue every global load of the workgroup as LDS-DMA (buffer_load_dwordx4 ... lds, 1 KiB per
//     instruction, NON-TEMPORAL so that the streamed input does not evict the half-written output lines from L2: 107 -> 85
//     us on the traffic skeleton), three 4-row sub-steps ahead, at raised priority (they are the youngest waves of their
//     SIMDs); their vmcnt queues hold loads only.  The compute waves' queues hold stores only and are never waited on: with
//     both in one queue the in-order counter made every load wait for the acknowledgement of older stores (150 us for the
//     same traffic).  One s_barrier per 4-row sub-step hands a landed sub-buffer over.  (Measured and dropped: progress
//     counters in LDS instead of the barriers — waves that spin on a counter steal issue slots from the waves they wait
//     for: 145-170 against 111-119 us; three columns per level-1 lane, i.e. 3 + 3 + 3 waves: 112.6 against 108.5 us;
//     3 to 6 staging sub-buffers: alike; pacing the requests with s_sleep: 106-142 against 104 us; 16-byte stores after a lane-pair
//     exchange, segments that hand rows over through a workspace instead of prologues, eight-wave workgroups two per CU: rounds 3-4,
//     EXPERIMENTS.md 0.1 / I.1, in the history of this file up to round 4.)
//   * stores are never branched around: rows a unit does not own go through a per-lane offset beyond every buffer resource;
//   * boundary extension: pad columns are filled inside LDS by the waves that read them, right before they do (a row is
//     complete one barrier after it was written, whoever wrote its columns); out-of-plane rows in zero mode are zero rows.
// Results agree with the per-level kernels to rounding (different summation order), with the fp64 oracle within 1e-6; they do not
// depend on how the rows are cut into units (every output is the same chain of FMAs wherever its unit starts).
// f32, even L <= 8, modes zero / constant / reflect / symmetric (periodic needs the far side of the plane); input rows of any length
// and alignment.
// Algorithmic traffic: 4 B H W read + 4 B (3 H1 W1 [+ 3 H2 W2] + 4 H_N W_N) written.
#include <atomic>
#include <cstring>
#include <mutex>

#include "mifwt_pyr.h"

namespace mifwt {

unsigned long long* g_pyr_prof = nullptr;

constexpr int kPyrSub = 4;    // level-0 rows per sub-step (one barrier each)
constexpr int kPyrPad = 8;    // floats in front of a staged / ring row
constexpr int kPyrRing = 16;  // ring rows (+ one zero row at slot 16)
constexpr int kPyrCtl = 64;   // bytes in front of the staging area
constexpr int kPyrWaves = 16;
constexpr int kPyrMaxChunks = 5;  // 1 KiB requests per staged row (three for the first loader wave, two for the second)
constexpr int kPyrMaxWG = 320;    // row chunks (= workgroups per column group) of a launch
constexpr int kPyrTailCols = 8;                                   // columns of a level the tail wave can take (its job grids: 4 / 6 / 6)
constexpr int kPyrTailRows = kPyrRing + 8;                        // rows of a tail image: 16 slots + the first 8 once more
constexpr int kPyrTailBytes = kPyrTailRows * kPyrTailCols * 8;    // one level's image of horizontally filtered (lo, hi) pairs

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
  int ngroups;                  // column groups per plane
  int cpg0, cpg;                // level-NLEV columns of group 0 / of the other groups
  int nchunks, nbuf;            // 1 KiB requests per level-0 row, staging sub-buffers of kPyrSub rows
  int pitch0, pitch1, pitch2;   // bytes of a staged row / a ring-1 row / a ring-2 row
  int nl1, nl2, nl3;            // waves of level 1 / 2 / 3
  int nt[3], tc0[3];            // TAIL columns of level l + 1: the nt last columns, from tc0 on, are the tail wave's (0: none)
  int twave;                    // the wave that runs them (-1: none)
  int mode;
  int exp;      // MIFWT_OPT_EXP: experiment word of the current A/B run (0 in the product)
  int l2split;  // the level-2 waves take one row pair in each half of a step (else both behind the step's second barrier)
  unsigned long long* prof;
  int dbg;
  DevTapArg dt;                        // device-resident taps (mifwt_common.h); dt.lo == nullptr: `tap` counts
  FastDiv hn_div;                      // division by H[NLEV]
  uint32_t wg_start[kPyrMaxWG + 1];    // chunk k = rows [wg_start[k], wg_start[k + 1]) of the batch's level-NLEV rows laid end to end
  f2 tap[L];
};

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
template <int NW>
__device__ __forceinline__ void pyr_role(int wave, int nl1, int nchunks, int twave, int& role, int& idx) {
  role = -1;
  idx = 0;
  if constexpr (NW == 12) {
    // TWELVE waves (at most 4 + 2 + 2 level waves: a workgroup of three waves per SIMD may use 168 registers a lane, and the
    // three-level 8-tap kernel wants more than the 128 of a sixteen-wave workgroup): level 1 = waves 0-3, level 2 = 4-5, level 3 =
    // 6-7, the tail wave = 9, loaders = 11 and 10; SIMD classes {L1, L2} {L1, L2, tail} {L1, L3, loader} {L1, L3, loader}
    if (wave == twave) role = kRoleTail;
    else if (wave < 4) role = kRoleL1, idx = wave;
    else if (wave < 6) role = kRoleL2, idx = wave - 4;
    else if (wave < 8) role = kRoleL3, idx = wave - 6;
    else if (wave == 11 || (wave == 10 && nchunks >= 2)) role = kRoleLoad, idx = 11 - wave;
    return;
  }
  if (wave == twave) {
    role = kRoleTail;
  } else if (wave < 5) {
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

// DT: the taps come from device memory (a.dt; a learnable filter bank that lives on the GPU) — an instance of its own, so that the default
// instance's registers are exactly what they were (mifwt_dwt2_inv_pyr.hip)
template <int L, int NLEV, bool PROF, int NW, bool DT = false>
__global__ void __launch_bounds__(64 * NW) dwt2_fwd_pyr_kernel(const PyrArgs<L, NLEV> a) {
  constexpr int HL = L - 2, HP = L / 2;
  constexpr int NC1 = 2;          // columns per level-1 lane (three were measured: 112.6 against 108.5 us on config 2)
  constexpr int NP = 2 * HL + 1;  // extension columns of a row: HL on the left, HL + 1 on the right
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int role, widx;
  pyr_role<NW>(wave, a.nl1, a.nchunks, a.twave, role, widx);
  if (role < 0 || (role == kRoleL1 && widx >= a.nl1) || (role == kRoleL2 && (NLEV < 2 || widx >= a.nl2)) ||
      (role == kRoleL3 && (NLEV < 3 || widx >= a.nl3)))
    return;  // (a wave that has ended does not take part in the barriers of the others)

  // the chunk of this workgroup: rows [g_lo, g_hi) of the batch's level-NLEV rows laid end to end
  const int grp = blockIdx.x % a.ngroups;
  int chunk = blockIdx.x / a.ngroups;
  if (MIFWT_EXPW(a) >> 16) chunk = (chunk + ((MIFWT_EXPW(a) >> 16) & 15)) % (int)(gridDim.x / a.ngroups);  // (experiment: which chunk runs on which XCD)
  const uint32_t g_lo = a.wg_start[chunk], g_hi = a.wg_start[chunk + 1];
  if (g_lo >= g_hi) return;
  const bool zero_mode = a.mode == MIFWT_MODE_ZERO;
  Fold1 fold;
  fold.set(a.mode);

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
  // lane grids: level-1 lanes own NC1 columns from o1 on, placed so that a lane lies entirely inside or outside [pA1, ...)
  const int o1 = pA[1] - NC1 * ((pA[1] - cA[1] + NC1 - 1) / NC1);
  const int g0 = max(0, 2 * o1 - HL) & ~3;  // level-0 column at the start of a staged row's body (16-byte aligned)

  // ring-1 rows start sh1 floats later where that gives the level-2 windows (first sample: column 2 cA2 - HL) the alignment
  // pyr_load_win2 expects, whatever the group's position
  int sh1 = 0;
  if constexpr (NLEV >= 2) sh1 = (2 * cA[2] - cA[1]) & 2;  // 2 cA2 - cA1 is 0 or L - 2
  unsigned char* const stage = smem + kPyrCtl;
  unsigned char* const ring1 = stage + a.nbuf * kPyrSub * a.pitch0;
  unsigned char* const ring2 = ring1 + (kPyrRing + 1) * a.pitch1;
  unsigned char* const tailimg = ring2 + (kPyrRing + 1) * a.pitch2;  // (three images of kPyrTailBytes, if there is a tail wave)

  unsigned long long waited = 0;
  const unsigned long long t_start = PROF ? __builtin_readcyclecounter() : 0;
  // (profiling build: wave 0 also leaves the workgroup's start / end on the 100 MHz wall clock and where it ran — HW_ID, XCC_ID — in
  // the slots of the unused waves 8 and 12: ramp, tail and the gap between launches, tools/pyr_clock.py)
  const unsigned long long w_start = PROF ? __builtin_amdgcn_s_memrealtime() : 0;

  f2 tap[L];
  if (role != kRoleLoad) {
    if constexpr (DT) {  // (a learnable filter bank that lives on the GPU: read once from device memory)
#pragma unroll
      for (int m = 0; m < L; ++m) tap[m] = (f2){dtap_lo<float>(a.dt, m), dtap_hi<float>(a.dt, m)};
    } else {
#pragma unroll
      for (int m = 0; m < L; ++m) tap[m] = a.tap[m];
    }
    // LDS initialisation (pads in zero mode, the zero rows of the rings), once per workgroup — every unit has the same geometry and
    // nothing ever writes these places: the level-1 waves clear the staging area, the level-2 waves the rings
    const int nst = a.nbuf * kPyrSub * a.pitch0 / 16, nrg = ((kPyrRing + 1) * (a.pitch1 + a.pitch2)) / 16;
    if (role == kRoleL1)
      for (int i = widx * 64 + lane; i < nst; i += 64 * a.nl1) reinterpret_cast<f4*>(stage)[i] = (f4){0.f, 0.f, 0.f, 0.f};
    if (NLEV >= 2 && role == kRoleL2)
      for (int i = widx * 64 + lane; i < nrg; i += 64 * a.nl2) reinterpret_cast<f4*>(ring1)[i] = (f4){0.f, 0.f, 0.f, 0.f};
  } else {
    if (!(MIFWT_DBG(a) & 16)) __builtin_amdgcn_s_setprio(3);  // the youngest wave of its SIMD, and the one everybody waits for
  }
  __syncthreads();  // ... before the loader's first row lands

  // =====================================================================================================================
  // the units of the chunk, one after the other.  The unit loop is instantiated ONCE PER ROLE (round 6): with one loop around the
  // chain of role branches every value any role reads in its step loop was live across the step loops of all the others — 144-218
  // spilled scalars per instance, reloaded with v_readlane inside the loops, and uniform store offsets that ended up in vector
  // registers behind readfirstlane loops.  Inside a role's own loop only that role's values are alive.
  auto run_units = [&](auto role_tag) {
  constexpr int role = decltype(role_tag)::value;
#pragma unroll 1
  for (uint32_t g = g_lo; g < g_hi;) {
    const int img = (int)a.hn_div.div(g);
    const int u_lo = (int)(g - (uint32_t)img * (uint32_t)a.H[NLEV]);
    const int u_rows = min(a.H[NLEV] - u_lo, (int)(g_hi - g));
    g += (uint32_t)u_rows;

    // ---- row ranges of this unit: computed rows [rA, rB) and owned rows [oA, oB) per level (index = level) --------------
    int rA[NLEV + 1], rB[NLEV + 1], oA[NLEV + 1], oB[NLEV + 1];
    oA[NLEV] = rA[NLEV] = u_lo;
    oB[NLEV] = rB[NLEV] = u_lo + u_rows;
#pragma unroll
    for (int l = NLEV - 1; l >= 1; --l) {
      oA[l] = 2 * oA[l + 1];
      oB[l] = oB[l + 1] == a.H[l + 1] ? a.H[l] : min(a.H[l], 2 * oB[l + 1]);
      rA[l] = max(0, 2 * rA[l + 1] - HL);
      rB[l] = min(a.H[l], 2 * rB[l + 1]);
    }
    const bool top = u_lo == 0;
    const int D2 = top ? pyr_lag2(L) : pyr_lag2_inner(L);
    const int D3 = top ? pyr_lag3(L) : pyr_lag3_inner(L);
    const int E0 = 2 * rA[1] - HL, e0_end = 2 * rB[1];
    const int npair1 = rB[1] - rA[1] + HP - 1;
    const int nsteps1 = (npair1 + 3) / 4;
    int nsteps = nsteps1, npair2 = 0, npair3 = 0;
    if constexpr (NLEV >= 2) {
      npair2 = rB[2] - rA[2] + HP - 1;
      nsteps = max(nsteps, D2 + (npair2 + 1) / 2);
    }
    if constexpr (NLEV >= 3) {
      npair3 = rB[3] - rA[3] + HP - 1;
      nsteps = max(nsteps, D3 + npair3);
    }
    const int nsub = 2 * nsteps, nsub1 = 2 * nsteps1;

    if (role == kRoleLoad) {
      // ===================================================================================================================
      // loader wave
      const uint32_t img_bytes = ((uint32_t)(a.H[0] - 1) * (uint32_t)a.xs_h + (uint32_t)a.W[0]) * 4u;
      const rsrc_t xr = pyr_rsrc(a.x + (int64_t)img * a.xs_b, img_bytes);
      const rsrc_t xr_dead = pyr_rsrc(a.x + (int64_t)img * a.xs_b, 0);  // every lane out of range: a row of zeros lands
      const uint32_t row_bytes = (uint32_t)a.xs_h * 4u;
      // loader `widx` of two requests the 1-KiB pieces widx, widx + 2, widx + 4 of every row
      const int mych = (a.nchunks - widx + 1) / 2;
      uint32_t voff[3];
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int c = g0 + 256 * (widx + 2 * j) + 4 * lane;
        voff[j] = (j < mych && c < a.W[0]) ? 4u * (uint32_t)c : kPyrOob;
      }
      auto run = [&](auto nch_tag) {
        constexpr int NCH = decltype(nch_tag)::value;
        constexpr int PER = kPyrSub * NCH;
        int ib = 0;  // staging sub-buffer of the next sub-step to be requested (sub-steps are requested in order)
        auto issue = [&](int t) {
          const uint32_t buf = (uint32_t)ib * (uint32_t)(kPyrSub * a.pitch0) + (uint32_t)kPyrCtl + kPyrPad * 4u + 1024u * (uint32_t)widx;
          ib = ib + 1 == a.nbuf ? 0 : ib + 1;
          if (MIFWT_DBG(a) & 2) return;
#pragma unroll
          for (int kk = 0; kk < kPyrSub; ++kk) {
            const int e = E0 + kPyrSub * t + kk;
            const bool dead = e >= e0_end || (zero_mode && (unsigned)e >= (unsigned)a.H[0]);
            const uint32_t soff = dead ? 0u : (uint32_t)fold(e, a.H[0]) * row_bytes;
            pyr_dma_row<NCH, 0x800>(voff, dead ? xr_dead : xr, soff, buf + (uint32_t)(kk * a.pitch0));
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
      pyr_dispatch<3>(mych - 1, [&](auto k) { run(std::integral_constant<int, decltype(k)::value + 1>{}); });
    } else if (role == kRoleL1) {
      // ===================================================================================================================
      // level-1 wave: NC1 columns per lane
      const int gmax = (cB[1] - o1 + NC1 - 1) / NC1 - 1;               // last lane of the grid that has a column
      const int glane = 64 * widx + lane;
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
      // level-0 pad fill, by the waves whose windows reach the pads: lane -> (row kk of the sub-step, pad column); a second pass
      // where the four rows' pads are more than a wave's lanes (ten taps: 4 x 17)
      constexpr int NFP = (kPyrSub * NP + 63) / 64;
      uint32_t f_src[NFP], f_dst[NFP];
      bool f_on[NFP];
      bool f_some = false;
#pragma unroll
      for (int ps = 0; ps < NFP; ++ps) {
        f_src[ps] = f_dst[ps] = 0;
        f_on[ps] = false;
        const int wlo = 2 * (o1 + NC1 * min(64 * widx, gmax)) - HL, whi = 2 * (o1 + NC1 * min(64 * widx + 63, gmax) + NC1 - 1) + 1;
        const int fl = lane + 64 * ps;
        const int kk = fl / NP, p = fl - kk * NP;
        const bool left = p < HL;
        const int e = left ? p - HL : a.W[0] + (p - HL);
        // (zero mode: the pads are zeros from the LDS initialisation — except the right one of rows that are not a multiple of 4
        // samples long, where the last lane of a row's DMA request brings up to three samples of whatever follows the row)
        const bool zfix = zero_mode && !left && (a.W[0] & 3) != 0;
        if (kk < kPyrSub && (!zero_mode || zfix) && (left ? wlo < 0 : whi >= a.W[0])) {
          f_on[ps] = true;
          f_some = true;
          f_src[ps] = (uint32_t)(kk * a.pitch0) + 4u * (uint32_t)(kPyrPad + (zero_mode ? 0 : fold(e, a.W[0])) - g0);
          f_dst[ps] = (uint32_t)(kk * a.pitch0) + 4u * (uint32_t)(kPyrPad + e - g0);
        }
      }
      const bool f_any = __builtin_amdgcn_readfirstlane(__builtin_amdgcn_ballot_w64(f_some) != 0) && !(MIFWT_DBG(a) & 16384);
      // one buffer resource for the three detail planes of the image (band = scalar offset), one for the approximation; a row
      // the unit does not own is stored at a per-lane offset beyond every resource (dropped) — the resources never change
      const uint32_t dbytes = (MIFWT_DBG(a) & 1) ? 0u : a.dspan[0] + ((uint32_t)(a.H[1] - 1) * (uint32_t)a.ds_h[0] + (uint32_t)a.W[1]) * 4u;
      const uint32_t abytes = (MIFWT_DBG(a) & 1) ? 0u : ((uint32_t)(a.H[NLEV] - 1) * (uint32_t)a.as_h + (uint32_t)a.W[NLEV]) * 4u;
      const rsrc_t rd = pyr_rsrc(a.det[0] + (int64_t)img * a.ds_b[0], dbytes);
      const rsrc_t ra = pyr_rsrc(a.approx + (int64_t)img * a.as_b, NLEV == 1 ? abytes : 0u);
      const uint32_t o0 = a.doff[0][0], o1b = a.doff[0][1], o2 = a.doff[0][2];

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
              float v[NFP];
#pragma unroll
              for (int ps = 0; ps < NFP; ++ps) v[ps] = zero_mode ? 0.f : *reinterpret_cast<const float*>(sb + f_src[ps]);
              wave_lds_fence();
#pragma unroll
              for (int ps = 0; ps < NFP; ++ps)
                if (f_on[ps]) *reinterpret_cast<float*>(sb + f_dst[ps]) = v[ps];
              wave_lds_fence();
            }
            f2 w[kPyrSub][NW2];
#pragma unroll
            for (int kk = 0; kk < kPyrSub; ++kk) load_win(sb + kk * a.pitch0 + win, w[kk]);
            pyr_static_for<kPyrSub / 2>([&](auto jj_tag) {
              constexpr int jj = decltype(jj_tag)::value;
              constexpr int j = 2 * half + jj;      // pair of the step
              constexpr int R = (4 * SM + j) % HP;  // its index modulo L/2
              f2 ha[NC1], hb[NC1];
              h_pair(w[2 * jj], w[2 * jj + 1], ha, hb);
              acc.template feed<0, R>(tap, ha);
              acc.template feed<1, R>(tap, hb);
              const int i = rA[1] + 4 * s + j - (HP - 1);  // the level-1 row it completes
              const f2 (&lo)[NC1] = acc.lo[PyrAcc<L, NC1>::done(R)];
              const f2 (&hi)[NC1] = acc.hi[PyrAcc<L, NC1>::done(R)];
              if constexpr (NLEV >= 2) {
                unsigned char* rr = ring1 + ((4 * s + j) & (kPyrRing - 1)) * a.pitch1;
#pragma unroll
                for (int k = 0; k < NC1; ++k) *reinterpret_cast<float*>(rr + (i < rB[1] ? rw[k] : 0u)) = lo[k].x;  // (float 0 of a row is never read)
              }
              const bool own = i >= oA[1] && i < oB[1];
              const uint32_t v2 = own ? sv2 : kPyrOob;
              const uint32_t so = own ? (uint32_t)i * (uint32_t)a.ds_h[0] * 4u : 0u;
              __builtin_amdgcn_raw_buffer_store_b64((f2){hi[0].x, hi[1].x}, rd, v2, so + o0, MIFWT_ST_AUX);
              __builtin_amdgcn_raw_buffer_store_b64((f2){lo[0].y, lo[1].y}, rd, v2, so + o1b, MIFWT_ST_AUX);
              __builtin_amdgcn_raw_buffer_store_b64((f2){hi[0].y, hi[1].y}, rd, v2, so + o2, MIFWT_ST_AUX);
              if (rag) {
                const uint32_t v1 = own ? sv1 : kPyrOob;
                pyr_store1(hi[0].x, rd, v1, so + o0);
                pyr_store1(lo[0].y, rd, v1, so + o1b);
                pyr_store1(hi[0].y, rd, v1, so + o2);
              }
              if constexpr (NLEV == 1) {
                const uint32_t sa = own ? (uint32_t)i * (uint32_t)a.as_h * 4u : 0u;
                __builtin_amdgcn_raw_buffer_store_b64((f2){lo[0].x, lo[1].x}, ra, v2, sa, MIFWT_ST_AUX);
                if (rag) pyr_store1(lo[0].x, ra, own ? sv1 : kPyrOob, sa);
              }
            });
          }
        });
      };
      int sm = 0;
#pragma unroll 1
      for (int s = 0; s < nsteps; ++s) {
        if constexpr (HP == 3 || HP == 5) {  // (4 s + j) mod L/2 depends on s
          pyr_dispatch<HP>(sm, [&](auto t) { step1(t, s); });
          sm = sm == HP - 1 ? 0 : sm + 1;
        } else {
          step1(std::integral_constant<int, 0>{}, s);
        }
      }
    } else if (NLEV >= 2 && role == kRoleL2) {
      // ===================================================================================================================
      // level-2 wave: two columns per lane, rows from ring 1
      if constexpr (NLEV >= 2) {
        const int gmax = (cB[2] - cA[2] + 1) / 2 - 1;
        const int glane = 64 * widx + lane;
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
        const bool f_any = __builtin_amdgcn_readfirstlane(__builtin_amdgcn_ballot_w64(f_on) != 0) && !(MIFWT_DBG(a) & 32768);
        const uint32_t dbytes = (MIFWT_DBG(a) & 1) ? 0u : a.dspan[1] + ((uint32_t)(a.H[2] - 1) * (uint32_t)a.ds_h[1] + (uint32_t)a.W[2]) * 4u;
        const uint32_t abytes = (MIFWT_DBG(a) & 1) ? 0u : ((uint32_t)(a.H[NLEV] - 1) * (uint32_t)a.as_h + (uint32_t)a.W[NLEV]) * 4u;
        const rsrc_t rd = pyr_rsrc(a.det[1] + (int64_t)img * a.ds_b[1], dbytes);
        const rsrc_t ra = pyr_rsrc(a.approx + (int64_t)img * a.as_b, NLEV == 2 ? abytes : 0u);
        const uint32_t o0 = a.doff[1][0], o1b = a.doff[1][1], o2 = a.doff[1][2];
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
        if ((MIFWT_EXPW(a) & 3) == 1) __builtin_amdgcn_s_setprio(1);
        if ((MIFWT_EXPW(a) & 3) == 2) __builtin_amdgcn_s_setprio(2);
        if ((MIFWT_EXPW(a) & 3) == 3) __builtin_amdgcn_s_setprio(3);
        // One row pair in each HALF of a step (a.l2split, round 5).  Until then a level-2 wave did a whole step's work (two pairs, ~330
        // instructions) behind the step's second barrier and sat out the first half: per-wave clocks had it waiting for about half of
        // its life, i.e. never in its own half — it was what the second half of every step waited for (a level-1 wave needs ~290
        // instructions per half).  The rows pair 0 reads (pair indices up to 4 (s - D2) + L/2 of level 1) are complete one barrier
        // earlier than those of pair 1 wherever 4 D2 >= L/2 + 1, which the lags guarantee — except where pair 0 reads MIRRORED rows at
        // the top of the plane (the first step of a top unit): that pair then runs with pair 1, as before.  Bit-identical.
        unsigned long long tsec[4] = {0, 0, 0, 0};  // (profiling build: cycles in window loads / arithmetic / ring write + stores / pad fill)
        auto do_pair = [&](auto r_tag, auto jj_tag, int s, const uint32_t (&so_r)[4]) {
          constexpr int R = decltype(r_tag)::value, jj = decltype(jj_tag)::value;
          unsigned long long t0 = 0;
          if constexpr (PROF) t0 = __builtin_readcyclecounter();
          f2 wa[NW2], wb[NW2];
          load_win(ring1 + so_r[2 * jj] + win, wa);
          load_win(ring1 + so_r[2 * jj + 1] + win, wb);
          if constexpr (PROF) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const unsigned long long t1 = __builtin_readcyclecounter();
            tsec[0] += t1 - t0;
            t0 = t1;
          }
          f2 ha[2], hb[2];
          h_pair(wa, wb, ha, hb);
          acc.template feed<0, R>(tap, ha);
          acc.template feed<1, R>(tap, hb);
          if constexpr (PROF) {
            asm volatile("" ::: "memory");
            const unsigned long long t1 = __builtin_readcyclecounter();
            tsec[1] += t1 - t0;
            t0 = t1;
          }
          const int p = 2 * (s - D2) + jj;
          const int i = rA[2] + p - (HP - 1);
          const f2 (&lo)[2] = acc.lo[PyrAcc<L, 2>::done(R)];
          const f2 (&hi)[2] = acc.hi[PyrAcc<L, 2>::done(R)];
          if constexpr (NLEV >= 3) {
            unsigned char* rr = ring2 + (p & (kPyrRing - 1)) * a.pitch2;
            const bool mine = i < rB[2];
            *reinterpret_cast<float*>(rr + (mine ? rw[0] : 0u)) = lo[0].x;
            *reinterpret_cast<float*>(rr + (mine ? rw[1] : 0u)) = lo[1].x;
          }
          const bool own = i >= oA[2] && i < oB[2];
          const uint32_t v2 = own ? sv2 : kPyrOob;
          const uint32_t so = own ? (uint32_t)i * (uint32_t)a.ds_h[1] * 4u : 0u;
          __builtin_amdgcn_raw_buffer_store_b64((f2){hi[0].x, hi[1].x}, rd, v2, so + o0, MIFWT_ST_AUX);
          __builtin_amdgcn_raw_buffer_store_b64((f2){lo[0].y, lo[1].y}, rd, v2, so + o1b, MIFWT_ST_AUX);
          __builtin_amdgcn_raw_buffer_store_b64((f2){hi[0].y, hi[1].y}, rd, v2, so + o2, MIFWT_ST_AUX);
          if (rag) {
            const uint32_t v1 = own ? sv1 : kPyrOob;
            pyr_store1(hi[0].x, rd, v1, so + o0);
            pyr_store1(lo[0].y, rd, v1, so + o1b);
            pyr_store1(hi[0].y, rd, v1, so + o2);
          }
          if constexpr (NLEV == 2) {
            const uint32_t sa = own ? (uint32_t)i * (uint32_t)a.as_h * 4u : 0u;
            __builtin_amdgcn_raw_buffer_store_b64((f2){lo[0].x, lo[1].x}, ra, v2, sa, MIFWT_ST_AUX);
            if (rag) pyr_store1(lo[0].x, ra, own ? sv1 : kPyrOob, sa);
          }
          if constexpr (PROF) {
            asm volatile("" ::: "memory");
            tsec[2] += __builtin_readcyclecounter() - t0;
          }
        };
        // (the pair index modulo L/2 — the accumulators' slot rotation — is a COMPILE-TIME constant of every pair: a step takes two pairs,
        // so the step loop is unrolled over the period of 2 k mod L/2.  Until round 5 a runtime switch picked one of L/2 variants of the
        // pair's code per pair: a branch tree, and 14 accumulator copies per pair where the variants' register assignments met again)
        auto half = [&](auto r_tag, auto jj_tag, int s, const uint32_t (&so_r)[4]) {
          constexpr int jj = decltype(jj_tag)::value;
          unsigned long long tp = 0;
          if constexpr (PROF) tp = __builtin_readcyclecounter();
          if (f_any) {  // the pads of the pair's two rows
            const uint32_t fo = f_r == 0 ? so_r[0] : (f_r == 1 ? so_r[1] : (f_r == 2 ? so_r[2] : so_r[3]));
            const float v = *reinterpret_cast<const float*>(ring1 + fo + f_src);
            wave_lds_fence();
            if (f_on && (f_r >> 1) == jj) *reinterpret_cast<float*>(ring1 + fo + f_dst) = v;
            wave_lds_fence();
          }
          if constexpr (PROF) tsec[3] += __builtin_readcyclecounter() - tp;
          do_pair(r_tag, jj_tag, s, so_r);
        };
        // one step: its two pairs have indices R0, R0 + 1 modulo L/2
        auto step2 = [&](auto r0_tag, int s) {
          constexpr int R0 = decltype(r0_tag)::value;
          using RA = std::integral_constant<int, R0 % HP>;
          using RB = std::integral_constant<int, (R0 + 1) % HP>;
          pyr_barrier<PROF>(waited);
          const bool act = 2 * (s - D2) < npair2 && !(MIFWT_DBG(a) & 4);
          uint32_t so_r[4] = {0u, 0u, 0u, 0u};  // ring-1 byte offsets of the step's four rows
          bool early = false;
          if (act) {
            int qmax0 = 0;  // the newest level-1 pair index pair 0 reads
            const int e0 = E1 + 4 * (s - D2);
            if (e0 >= 0 && e0 + 3 < a.H[1]) {
              // rows inside the plane (every step but the first / last few of a plane): consecutive ring slots, no index map — the map
              // and its four selections were 130 of the ~530 instructions of a level-2 wave's step
              const int q0 = e0 + ro1;
#pragma unroll
              for (int r = 0; r < 4; ++r) so_r[r] = (uint32_t)(((q0 + r) & (kPyrRing - 1)) * a.pitch1);
              qmax0 = q0 + 1;
            } else {
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const int e = e0 + r;
                const bool dead = zero_mode && (unsigned)e >= (unsigned)a.H[1];
                const int q = fold(e, a.H[1]) + ro1;
                so_r[r] = (uint32_t)((dead ? kPyrRing : (q & (kPyrRing - 1))) * a.pitch1);
                if (r < 2 && !dead) qmax0 = max(qmax0, q);
              }
            }
            // (for the tail wave, which filters the same rows behind the step's second barrier)
            if (NW == 12 && widx == 0 && a.twave >= 0 && lane == 0) *reinterpret_cast<u4*>(smem) = (u4){so_r[0], so_r[1], so_r[2], so_r[3]};
            early = a.l2split && qmax0 <= 4 * s - 1;  // (written in an earlier step: complete behind this step's first barrier)
            if (early) half(RA{}, std::integral_constant<int, 0>{}, s, so_r);
          }
          pyr_barrier<PROF>(waited);  // level 1's first two rows of this step (and everything before) are in ring 1
          if (act) {
            if (!early) half(RA{}, std::integral_constant<int, 0>{}, s, so_r);
            half(RB{}, std::integral_constant<int, 1>{}, s, so_r);
          }
        };
        {
          int s = 0;
#pragma unroll 1
          for (; s < D2 && s < nsteps; ++s) {  // (the lag: level 1 has not produced the first rows yet)
            pyr_barrier<PROF>(waited);
            pyr_barrier<PROF>(waited);
          }
          constexpr int PERIOD = (HP % 2 == 0) ? (HP / 2 > 0 ? HP / 2 : 1) : HP;  // steps until 2 k mod L/2 repeats
#pragma unroll 1
          for (; s < nsteps; s += PERIOD) {
            pyr_static_for<PERIOD>([&](auto k_tag) {
              constexpr int k = decltype(k_tag)::value;
              if (k == 0 || s + k < nsteps) step2(std::integral_constant<int, (2 * k) % HP>{}, s + k);
            });
          }
        }
        if (PROF && widx == 0 && lane == 0) {
          unsigned long long* o = MIFWT_PROFP(a) + ((size_t)blockIdx.x * kPyrWaves + 14) * 2;
          o[0] += tsec[0], o[1] += tsec[1], o[2] += tsec[2], o[3] += tsec[3];
        }
      }
    } else if (NLEV >= 3 && role == kRoleL3) {
      // ===================================================================================================================
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
        const bool f_any = __builtin_amdgcn_readfirstlane(__builtin_amdgcn_ballot_w64(f_on) != 0) && !(MIFWT_DBG(a) & 32768);
        const uint32_t dbytes = (MIFWT_DBG(a) & 1) ? 0u : a.dspan[2] + ((uint32_t)(a.H[3] - 1) * (uint32_t)a.ds_h[2] + (uint32_t)a.W[3]) * 4u;
        const uint32_t abytes = (MIFWT_DBG(a) & 1) ? 0u : ((uint32_t)(a.H[3] - 1) * (uint32_t)a.as_h + (uint32_t)a.W[3]) * 4u;
        const rsrc_t rd = pyr_rsrc(a.det[2] + (int64_t)img * a.ds_b[2], dbytes);
        const rsrc_t ra = pyr_rsrc(a.approx + (int64_t)img * a.as_b, abytes);
        const uint32_t o0 = a.doff[2][0], o1b = a.doff[2][1], o2 = a.doff[2][2];
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
        // (as in the level-2 waves: the pair index modulo L/2 is a compile-time constant, the step loop unrolled over its period)
        auto step3 = [&](auto r_tag, int s) {
          constexpr int R = decltype(r_tag)::value;  // (s - D3) mod L/2
          pyr_barrier<PROF>(waited);
          if (s - D3 < npair3 && !(MIFWT_DBG(a) & 4)) {
            uint32_t so_r[2];
            const int e0 = E2 + 2 * (s - D3);
            if (e0 >= 0 && e0 + 1 < a.H[2]) {
              so_r[0] = (uint32_t)(((e0 + ro2) & (kPyrRing - 1)) * a.pitch2);
              so_r[1] = (uint32_t)(((e0 + 1 + ro2) & (kPyrRing - 1)) * a.pitch2);
            } else {
#pragma unroll
              for (int r = 0; r < 2; ++r) {
                const int e = e0 + r;
                const bool dead = zero_mode && (unsigned)e >= (unsigned)a.H[2];
                so_r[r] = (uint32_t)((dead ? kPyrRing : ((fold(e, a.H[2]) + ro2) & (kPyrRing - 1))) * a.pitch2);
              }
            }
            if (NW == 12 && widx == 0 && a.twave >= 0 && lane == 0) *reinterpret_cast<u2*>(smem + 16) = (u2){so_r[0], so_r[1]};
            if (f_any) {
              const uint32_t fo = f_r == 0 ? so_r[0] : so_r[1];
              const float v = *reinterpret_cast<const float*>(ring2 + fo + f_src);
              wave_lds_fence();
              if (f_on) *reinterpret_cast<float*>(ring2 + fo + f_dst) = v;
              wave_lds_fence();
            }
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
            pyr_store1(lo.y, rd, v, so + o1b);
            pyr_store1(hi.y, rd, v, so + o2);
            pyr_store1(lo.x, ra, v, own ? (uint32_t)i * (uint32_t)a.as_h * 4u : 0u);
          }
          pyr_barrier<PROF>(waited);
        };
        {
          int s = 0;
#pragma unroll 1
          for (; s < D3 && s < nsteps; ++s) {
            pyr_barrier<PROF>(waited);
            pyr_barrier<PROF>(waited);
          }
#pragma unroll 1
          for (; s < nsteps; s += HP) {
            pyr_static_for<HP>([&](auto k_tag) {
              constexpr int k = decltype(k_tag)::value;
              if (k == 0 || s + k < nsteps) step3(k_tag, s + k);
            });
          }
        }
      }
    }
    else if (NW == 12 && role == kRoleTail) {  // (the sixteen-wave form has no registers for it: 128 a lane)
      if constexpr (NW == 12) {
      // ===================================================================================================================
      // TAIL wave (round 5).  515 / 261 / 134 columns are 4 x 128 + 3, 2 x 128 + 5, 2 x 64 + 6: a fifth level-1 wave, a third level-2
      // and a third level-3 wave ran the whole instruction stream of their level for two, three and six lanes (1066 of the 4600
      // instructions a workgroup issues per 8-row step).  The last few columns of a level are cheap in DIRECT form, one lane per
      // (row, column): stage 1 = the horizontal pass of every row that arrived this half step, for each tail column, into a small LDS
      // image of (lo, hi) pairs; stage 2 = the vertical pass of every row the half step completes, from the L newest rows of that
      // image — the same products in the same order as the rolling accumulators of the level waves (bit-identical), all three levels
      // in one instruction stream (lanes differ in where they read and write).  The samples come through the boundary index map
      // (a window that reaches beyond the plane reads the mirrored / clamped column itself), so nobody fills pads for the tail.
      // Level 1 rows are filtered in the half step they land in (a staging buffer is recycled after it), levels 2 and 3 behind the
      // step's second barrier — where the level-2 waves used to run their whole step.
      const int nt1 = a.nt[0], nt2 = NLEV >= 2 ? a.nt[1] : 0, nt3 = NLEV >= 3 ? a.nt[2] : 0;
      // stage-1 job of the lane: (level, row of the half step's new rows, tail column)
      int j_lvl = 0, j_r = 0, j_c = 0;
      {
        int idx = lane;
        if (idx < 4 * nt1) {
          j_lvl = 1, j_r = idx / max(nt1, 1), j_c = idx - j_r * nt1;
        } else {
          idx -= 4 * nt1;
          if (idx < 4 * nt2) {
            j_lvl = 2, j_r = idx / max(nt2, 1), j_c = idx - j_r * nt2;
          } else {
            idx -= 4 * nt2;
            if (idx < 2 * nt3) j_lvl = 3, j_r = idx / max(nt3, 1), j_c = idx - j_r * nt3;
          }
        }
      }
      // byte offsets of the job's L samples from the start of its source row (zero mode: float 1 of every row is a zero nobody writes)
      uint32_t soff[L];
      {
        const int Wp = j_lvl <= 1 ? a.W[0] : (j_lvl == 2 ? a.W[NLEV >= 2 ? 1 : 0] : a.W[NLEV >= 3 ? 2 : 0]);
        const int tc = j_lvl <= 1 ? a.tc0[0] : (j_lvl == 2 ? a.tc0[1] : a.tc0[2]);
        const int body0 = j_lvl <= 1 ? kPyrPad - g0 : (j_lvl == 2 ? kPyrPad + sh1 - cA[1] : kPyrPad - cA[NLEV >= 2 ? 2 : 1]);
#pragma unroll
        for (int k = 0; k < L; ++k) {
          const int x = 2 * (tc + j_c) - HL + k;
          const bool oob = (unsigned)x >= (unsigned)Wp;
          soff[k] = (oob && zero_mode) ? 4u : 4u * (uint32_t)(body0 + (zero_mode ? x : fold(x, Wp)));
        }
      }
      const uint32_t m_l1 = j_lvl == 1 ? ~0u : 0u, m_l2 = j_lvl == 2 ? ~0u : 0u, m_l3 = j_lvl == 3 ? ~0u : 0u;
      const uint32_t m_r0 = j_r == 0 ? ~0u : 0u, m_r1 = j_r == 1 ? ~0u : 0u, m_r2 = j_r == 2 ? ~0u : 0u, m_r3 = j_r == 3 ? ~0u : 0u;
      const uint32_t j_rp0 = (uint32_t)j_r * (uint32_t)a.pitch0, r1off = (uint32_t)(ring1 - smem), r2off = (uint32_t)(ring2 - smem);
      const uint32_t img1 = (uint32_t)(tailimg - smem) + (uint32_t)(max(j_lvl, 1) - 1) * (uint32_t)kPyrTailBytes + 8u * (uint32_t)j_c;
      // stage-2 job of the lane: (level, output row of the half step, tail column)
      int k_lvl = 0, k_o = 0, k_c = 0;
      {
        int idx = lane;
        if (idx < 2 * nt1) {
          k_lvl = 1, k_o = idx / max(nt1, 1), k_c = idx - k_o * nt1;
        } else {
          idx -= 2 * nt1;
          if (idx < 2 * nt2) {
            k_lvl = 2, k_o = idx / max(nt2, 1), k_c = idx - k_o * nt2;
          } else {
            idx -= 2 * nt2;
            if (idx < nt3) k_lvl = 3, k_c = idx;
          }
        }
      }
      constexpr int I2 = NLEV >= 2 ? 1 : 0, I3 = NLEV >= 3 ? 2 : 0;
      const int kC = (k_lvl <= 1 ? a.tc0[0] : (k_lvl == 2 ? a.tc0[1] : a.tc0[2])) + k_c;  // the column of the job's level
      const int k_rA = k_lvl <= 1 ? rA[1] : (k_lvl == 2 ? rA[I2 + 1] : rA[I3 + 1]);
      const int k_rB = k_lvl <= 1 ? rB[1] : (k_lvl == 2 ? rB[I2 + 1] : rB[I3 + 1]);
      const int k_oA = k_lvl <= 1 ? oA[1] : (k_lvl == 2 ? oA[I2 + 1] : oA[I3 + 1]);
      const int k_oB = k_lvl <= 1 ? oB[1] : (k_lvl == 2 ? oB[I2 + 1] : oB[I3 + 1]);
      const bool k_last = k_lvl == NLEV;  // the level whose approximation leaves the workgroup
      // the job's column in the three detail planes / the approximation plane of its image, and in the ring its approximation feeds
      const int kl = max(k_lvl, 1) - 1;
      unsigned char* const k_dptr = reinterpret_cast<unsigned char*>((kl == 0 ? a.det[0] : (kl == 1 ? a.det[I2] : a.det[I3])) +
                                                                      (int64_t)img * (kl == 0 ? a.ds_b[0] : (kl == 1 ? a.ds_b[I2] : a.ds_b[I3])) + kC);
      const uint32_t k_dpitch = 4u * (uint32_t)(kl == 0 ? a.ds_h[0] : (kl == 1 ? a.ds_h[I2] : a.ds_h[I3]));
      const uint32_t k_o0 = kl == 0 ? a.doff[0][0] : (kl == 1 ? a.doff[I2][0] : a.doff[I3][0]);
      const uint32_t k_o1 = kl == 0 ? a.doff[0][1] : (kl == 1 ? a.doff[I2][1] : a.doff[I3][1]);
      const uint32_t k_o2 = kl == 0 ? a.doff[0][2] : (kl == 1 ? a.doff[I2][2] : a.doff[I3][2]);
      unsigned char* const k_aptr = reinterpret_cast<unsigned char*>(a.approx + (int64_t)img * a.as_b + kC);
      const uint32_t k_apitch = 4u * (uint32_t)a.as_h;
      const uint32_t k_ring = k_lvl <= 1 ? (uint32_t)(ring1 - smem) + 4u * (uint32_t)(kPyrPad + sh1 + kC - cA[1])
                                         : (uint32_t)(ring2 - smem) + 4u * (uint32_t)(kPyrPad + kC - cA[I2 + 1]);
      const uint32_t k_rpitch = k_lvl <= 1 ? (uint32_t)a.pitch1 : (uint32_t)a.pitch2;
      const uint32_t img2 = (uint32_t)(tailimg - smem) + (uint32_t)kl * (uint32_t)kPyrTailBytes + 8u * (uint32_t)k_c;
      const uint32_t n_l1 = k_lvl == 1 ? ~0u : 0u, n_l2 = k_lvl == 2 ? ~0u : 0u, n_l3 = k_lvl == 3 ? ~0u : 0u;
      const bool nostore = MIFWT_DBG(a) & 1;
      int bi = 0;
      if (((MIFWT_EXPW(a) >> 4) & 3) == 0) __builtin_amdgcn_s_setprio(2);  // (short dependent chains that every half step waits for)
      if (((MIFWT_EXPW(a) >> 4) & 3) == 1) __builtin_amdgcn_s_setprio(1);
      if (((MIFWT_EXPW(a) >> 4) & 3) == 3) __builtin_amdgcn_s_setprio(3);

      // stage 1 then stage 2 of one half step; rb1 = LDS offset of the staged rows of the half step (level 1), v2 / v3 = ring offsets
      // of the rows levels 2 / 3 consume in this step (published by the first level-2 / level-3 wave); act = which levels run
      auto run_half = [&](int s, int half, bool act1, bool act2, bool act3, uint32_t rb1, const u4 v2, const u2 v3) {
        // ---- stage 1 ------------------------------------------------------------------------------------------------------
        const bool on1 = (j_lvl == 1 && act1) || (j_lvl == 2 && act2) || (j_lvl == 3 && act3);
        if (on1) {
          // (selections as mask arithmetic: a chain of conditionals over these became a switch over a scratch array)
          const uint32_t rowb = (m_l1 & (rb1 + j_rp0)) | (m_l2 & (r1off + ((m_r0 & v2.x) | (m_r1 & v2.y) | (m_r2 & v2.z) | (m_r3 & v2.w)))) |
                                (m_l3 & (r2off + ((m_r0 & v3.x) | (m_r1 & v3.y))));
          const int seq = j_r + (int)((m_l1 & (uint32_t)(8 * s + 4 * half)) | (m_l2 & (uint32_t)(4 * (s - D2))) | (m_l3 & (uint32_t)(2 * (s - D3))));
          float xs[L];
#pragma unroll
          for (int k = 0; k < L; ++k) xs[k] = *reinterpret_cast<const float*>(smem + rowb + soff[k]);
          f2 h;
#pragma unroll
          for (int k = 0; k < HP; ++k) {
            const f2 pr = (f2){xs[2 * k], xs[2 * k + 1]};
            if (k == 0) h = vmul_lo(tap[L - 1], pr);
            else vfma_lo(h, tap[L - 1 - 2 * k], pr);
            vfma_hi(h, tap[L - 2 - 2 * k], pr);
          }
          const uint32_t slot = (uint32_t)seq & (kPyrRing - 1);
          *reinterpret_cast<f2*>(smem + img1 + slot * (8u * kPyrTailCols)) = h;
          *reinterpret_cast<f2*>(smem + img1 + (slot < 8u ? slot + kPyrRing : slot) * (8u * kPyrTailCols)) = h;
        }
        wave_lds_fence();
        // ---- stage 2 ------------------------------------------------------------------------------------------------------
        const int P = k_o + (int)((n_l1 & (uint32_t)(4 * s + 2 * half)) | (n_l2 & (uint32_t)(2 * (s - D2))) | (n_l3 & (uint32_t)(s - D3)));  // pair index of the output row
        const bool on2 = ((k_lvl == 1 && act1) || (k_lvl == 2 && act2) || (k_lvl == 3 && act3)) && P >= HP - 1;
        if (on2) {
          const uint32_t b = (uint32_t)(2 * P + 2 - L) & (kPyrRing - 1);
          const unsigned char* hp = smem + img2 + b * (8u * kPyrTailCols);
          f2 hv[L];
#pragma unroll
          for (int n = 0; n < L; ++n) hv[n] = *reinterpret_cast<const f2*>(hp + n * (8 * kPyrTailCols));
          f2 lo = vmul_lo(tap[L - 1], hv[0]), hi = vmul_hi(tap[L - 1], hv[0]);
#pragma unroll
          for (int n = 1; n < L; ++n) {
            vfma_lo(lo, tap[L - 1 - n], hv[n]);
            vfma_hi(hi, tap[L - 1 - n], hv[n]);
          }
          const int i = k_rA + P - (HP - 1);
          if (!k_last && i < k_rB) *reinterpret_cast<float*>(smem + k_ring + ((uint32_t)P & (kPyrRing - 1)) * k_rpitch) = lo.x;
          if (i >= k_oA && i < k_oB && !nostore) {
            unsigned char* drow = k_dptr + (size_t)i * k_dpitch;
            *reinterpret_cast<float*>(drow + k_o0) = hi.x;
            *reinterpret_cast<float*>(drow + k_o1) = lo.y;
            *reinterpret_cast<float*>(drow + k_o2) = hi.y;
            if (k_last) *reinterpret_cast<float*>(k_aptr + (size_t)i * k_apitch) = lo.x;
          }
        }
      };
#pragma unroll 1
      for (int s = 0; s < nsteps; ++s) {
        const bool act1 = s < nsteps1;
        const bool act2 = NLEV >= 2 && s >= D2 && 2 * (s - D2) < npair2 && !(MIFWT_DBG(a) & 4);
        const bool act3 = NLEV >= 3 && s >= D3 && s - D3 < npair3 && !(MIFWT_DBG(a) & 4);
        pyr_barrier<PROF>(waited);
        uint32_t rb1 = (uint32_t)(stage - smem) + (uint32_t)bi * (uint32_t)(kPyrSub * a.pitch0);
        if (act1) bi = bi + 1 == a.nbuf ? 0 : bi + 1;
        run_half(s, 0, act1, false, false, rb1, (u4){0u, 0u, 0u, 0u}, (u2){0u, 0u});
        pyr_barrier<PROF>(waited);
        const u4 v2 = *reinterpret_cast<const u4*>(smem);
        const u2 v3 = *reinterpret_cast<const u2*>(smem + 16);
        rb1 = (uint32_t)(stage - smem) + (uint32_t)bi * (uint32_t)(kPyrSub * a.pitch0);
        if (act1) bi = bi + 1 == a.nbuf ? 0 : bi + 1;
        run_half(s, 1, act1, act2, act3, rb1, v2, v3);
      }
      }
    }
    // the next unit's first requests land in staging buffers, its level-1 rows in ring slots, that this unit's last sub-step may
    // still be read from: everybody is through with it behind this barrier
    __syncthreads();
  }
  };
  if (role == kRoleLoad) run_units(std::integral_constant<int, kRoleLoad>{});
  else if (role == kRoleL1) run_units(std::integral_constant<int, kRoleL1>{});
  else if (role == kRoleL2) run_units(std::integral_constant<int, kRoleL2>{});
  else if (role == kRoleL3) run_units(std::integral_constant<int, kRoleL3>{});
  else run_units(std::integral_constant<int, kRoleTail>{});

  if (PROF && lane == 0 && role != kRoleLoad) {
    unsigned long long* o = MIFWT_PROFP(a) + ((size_t)blockIdx.x * kPyrWaves + wave) * 2;
    o[0] = __builtin_readcyclecounter() - t_start;
    o[1] = waited;
    if (wave == 0) {
      unsigned long long* w = MIFWT_PROFP(a) + ((size_t)blockIdx.x * kPyrWaves + 8) * 2;
      w[0] = w_start;
      w[1] = __builtin_amdgcn_s_memrealtime();
      w[8] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
      w[9] = __builtin_amdgcn_s_getreg((31 << 11) | 20);
      w[10] = g_lo;  // (slot of wave 13)
      w[11] = g_hi;
    }
  }
}

// ---- host side ------------------------------------------------------------------------------------------------------------
struct PyrPlan {
  int ngroups, cpg0, cpg, nchunks, nbuf, pitch0, pitch1, pitch2, nl1, nl2, nl3, lds;
  int nt[3], tc0[3], twave;            // tail columns per level and the wave that runs them (PyrArgs)
  int nwaves;                          // waves of a workgroup: 12 or 16 (pyr_role)
  int nwg;                             // row chunks = workgroups per column group
  uint32_t wg_start[kPyrMaxWG + 1];
};

// columns of level NLEV a group may own: the level-1 / 2 / 3 lane grids hold 6 x 128 / 3 x 128 / 3 x 64 columns, a staged row
// at most kPyrMaxChunks x 256 level-0 columns; interior groups recompute (L - 2) halo columns per level on their left
static int pyr_group_cols_search(int L, int nlev, bool first);
// (memoised: three plans per call — the route query, the envelope test, the launch — each asked twice; the search below is ~2 us, and
// the host time in front of the first launch of a short timed loop is time the GPU idles)
static int pyr_group_cols(int L, int nlev, bool first) {
  static std::atomic<int> memo[6][4][2];  // [L / 2][nlev][first]: value + 1 (0 = not computed); L <= 10, nlev <= 3
  const int li = L / 2;
  if (li < 0 || li > 5 || nlev < 0 || nlev > 3) return pyr_group_cols_search(L, nlev, first);
  std::atomic<int>& m = memo[li][nlev][first ? 1 : 0];
  int v = m.load(std::memory_order_relaxed);
  if (v == 0) {
    v = pyr_group_cols_search(L, nlev, first) + 1;
    m.store(v, std::memory_order_relaxed);
  }
  return v - 1;
}
static int pyr_group_cols_search(int L, int nlev, bool first) {
  const int nc1 = 2;
  const int HL = first ? 0 : L - 2;
  int best = 0;
  for (int n = 1; n <= 4 * 192; ++n) {
    int m = n;  // columns computed at the level below, walking down to level 1
    bool ok = true;
    for (int l = nlev; l >= 1 && ok; --l) {
      const int cap = l == 1 ? 6 * 128 - nc1 : (l == 2 ? 3 * 128 : 3 * 64);
      ok = m <= cap;
      if (l > 1) m = 2 * m + HL;
    }
    // m = level-1 columns now; level-0 span incl. the 16-byte alignment slack
    if (ok && 2 * m + HL + 3 <= kPyrMaxChunks * 256) best = n;
  }
  return best;
}

// ---- the schedule: which rows of which images a workgroup takes ---------------------------------------------------------------
// Modelled time of one unit = rows [u_lo, u_hi) of the last level of one image, in 8-row steps (the kernel's own formulas).
// Round 5, per-workgroup wall clocks on config 2 (tools/pyr_clock.py, profiles/r05a_clock.txt): a step in which the level-1 waves
// run (prologue or owned rows) costs 2.5 us, a drain step (deep levels only, no memory traffic) 1.1 us — that fits the top (35 + 9
// steps), middle (40 + 2) and bottom (32 + 8) segments of an image within 1 %; a unit that is not the first of its workgroup
// starts behind one more barrier with cold staging buffers (kUnitStart).
struct PyrGeom {
  int L, nlev, H[4];
  int parity;  // 1: chunk k runs on XCD k mod 8 (one column group) — the odd XCDs are the slower ones, see pyr_cut
};
constexpr double kUnitStart = 1.0;
static double pyr_unit_time(const PyrGeom& gm, int u_lo, int u_hi) {
  // (calibration runs, tools/pyr_calib.py: MIFWT_OPT_EXP bits 8-11 = drain weight in twentieths over 0.25, bits 12-15 = what a unit
  // at the top of its image saves — no prologue to request, nothing to wait for — in tenths of a step)
  const int ex = exp_word();
  const double kDrainStep = ((ex >> 8) & 15) ? 0.25 + 0.05 * ((ex >> 8) & 15) : 0.45;
  const double kTopBonus = ((ex >> 12) & 15) ? 0.1 * ((ex >> 12) & 15) : 1.0;  // (35 / 32 / 32 / 35 rows on config 2: 0.5-1 % ahead of 34 / 32 / 32 / 36, profiles/r05i_calib.txt)
  const int L = gm.L, HL = L - 2, HP = L / 2, nlev = gm.nlev;
  int rA[4], rB[4];
  rA[nlev] = u_lo;
  rB[nlev] = u_hi;
  for (int l = nlev - 1; l >= 1; --l) {
    rA[l] = std::max(0, 2 * rA[l + 1] - HL);
    rB[l] = std::min(gm.H[l], 2 * rB[l + 1]);
  }
  const bool top = u_lo == 0;
  const int D2 = top ? pyr_lag2(L) : pyr_lag2_inner(L), D3 = top ? pyr_lag3(L) : pyr_lag3_inner(L);
  const int nsteps1 = (rB[1] - rA[1] + HP - 1 + 3) / 4;
  int nsteps = nsteps1;
  if (nlev >= 2) nsteps = std::max(nsteps, D2 + (rB[2] - rA[2] + HP - 1 + 1) / 2);
  if (nlev >= 3) nsteps = std::max(nsteps, D3 + rB[3] - rA[3] + HP - 1);
  return nsteps1 + kDrainStep * (nsteps - nsteps1) + kUnitStart - (top ? kTopBonus : 0.0);
}

// Cuts the B x HN rows into at most `gmax` chunks whose modelled times do not exceed `budget`: greedy, a chunk takes whole rests of
// images while they fit and then as many rows of the next image as fit (at least kMinRows, leaving at least kMinRows).  Returns the
// number of chunks (gmax + 1 if the rows do not fit), the cuts in `cut` if it is not null.
constexpr int kMinRows = 8;
static int pyr_cut(const PyrGeom& gm, int64_t B, double budget, int gmax, uint32_t* cut) {
  const int HN = gm.H[gm.nlev];
  int64_t img = 0;
  int row = 0, n = 0;
  if (cut) cut[0] = 0;
  // (XCD-aware budgets were measured and dropped, round 5: workgroup b runs on XCD b mod 8 and the workgroups of the ODD XCDs end 3-5 us
  // (of ~92) behind those of the even ones whatever rows they are given — but giving odd chunks 2-4 % less to do bought nothing with
  // the results dropped, 99.1 -> 98.8 us, and cost 2 us with rotating outputs: profiles/r05u_xcd_parity.txt)
  while (img < B) {
    if (n == gmax) return gmax + 1;
    double used = 0;
    bool first = true;
    while (img < B) {
      const double rest = pyr_unit_time(gm, row, HN);
      if (used + rest <= budget) {  // the rest of this image
        used += rest;
        ++img;
        row = 0;
        first = false;
        continue;
      }
      // a part of it: the most rows that fit (the unit time is monotone in the rows)
      int lo = 0, hi = HN - row - kMinRows;  // rows taken; lo fits (nothing), hi is the most that leaves kMinRows
      if (hi >= kMinRows) {
        while (lo < hi) {
          const int mid = (lo + hi + 1) / 2;
          if (used + pyr_unit_time(gm, row, row + mid) <= budget) lo = mid;
          else hi = mid - 1;
        }
      } else {
        lo = 0;
      }
      if (lo >= kMinRows) {
        row += lo;
        first = false;
      } else if (first) {
        // nothing fits into an empty chunk: the budget is too small
        return gmax + 1;
      }
      break;
    }
    ++n;
    if (cut) cut[n] = (uint32_t)(img * HN + row);
  }
  return n;
}

static void pyr_schedule(const PyrGeom& gm, int64_t B, int gwant, PyrPlan* p) {
  const int HN = gm.H[gm.nlev];
  // a small cache (a call loop asks for the same few schedules over and over — a five-level call of the reference's shape takes this
  // kernel for two of its levels; the search below is ~0.1 ms)
  struct Entry {
    int L, nlev, H[4], gwant, ex, n, parity;
    int64_t B;
    uint32_t cut[kPyrMaxWG + 1];
  };
  constexpr int kEntries = 16;
  static std::mutex mu;
  static Entry cache[kEntries];
  static int used = 0, next = 0;
  const int ex = exp_word();
  auto same = [&](const Entry& e) {
    return e.L == gm.L && e.nlev == gm.nlev && e.H[1] == gm.H[1] && e.H[2] == gm.H[2] && e.H[3] == gm.H[3] && e.gwant == gwant && e.ex == ex && e.parity == gm.parity && e.B == B;
  };
  {
    std::lock_guard<std::mutex> lk(mu);
    for (int i = 0; i < used; ++i)
      if (same(cache[i])) {
        p->nwg = cache[i].n;
        std::copy(cache[i].cut, cache[i].cut + cache[i].n + 1, p->wg_start);
        return;
      }
  }
  const int g = std::max(1, std::min(gwant, kPyrMaxWG));
  // the smallest budget the rows fit into g chunks with
  double lo = 0, hi = pyr_unit_time(gm, 0, HN);
  hi = hi * (double)((B + g - 1) / g) + hi;  // whole images dealt round-robin always fit
  for (int it = 0; it < 40 && hi - lo > 0.02; ++it) {
    const double mid = 0.5 * (lo + hi);
    if (pyr_cut(gm, B, mid, g, nullptr) <= g) hi = mid;
    else lo = mid;
  }
  p->nwg = pyr_cut(gm, B, hi, g, p->wg_start);
  {
    std::lock_guard<std::mutex> lk(mu);
    Entry& e = cache[next];
    next = (next + 1) % kEntries;
    used = std::min(used + 1, kEntries);
    e.L = gm.L, e.nlev = gm.nlev, e.gwant = gwant, e.ex = ex, e.B = B, e.n = p->nwg, e.parity = gm.parity;
    for (int l = 0; l < 4; ++l) e.H[l] = gm.H[l];
    std::copy(p->wg_start, p->wg_start + p->nwg + 1, e.cut);
  }
}

static bool pyr_plan_compute(int nlev, const mifwt_level_desc* const* d, PyrPlan* p) {
  const int nc1 = 2;  // columns per level-1 lane
  const int L = d[0]->filt_len, HL = L - 2;
  const int WN = (int)d[nlev - 1]->coef_extent[1], HN = (int)d[nlev - 1]->coef_extent[0];
  const int min_cols = HL + 2;
  if (WN < min_cols || HN < 2 * (HL + 2)) return false;
  if (d[0]->batch * (int64_t)HN >= (int64_t(1) << 31)) return false;
  p->cpg0 = pyr_group_cols(L, nlev, true);
  p->cpg = pyr_group_cols(L, nlev, false);
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
  if (p->nl1 > 6 || p->nl2 > 3 || p->nl3 > 3) return false;
  // The last few columns of a level (one column group: 1024-column planes have 515 = 4 x 128 + 3, 261 = 2 x 128 + 5, 134 = 2 x 64 + 6)
  // go to the TAIL wave instead of a nearly empty level wave: at most 4 / 6 / 6 columns (its lane grid holds 4 rows x 4 + 4 x 6 +
  // 2 x 6 jobs).  MIFWT_OPT_DEBUG bit 19 keeps the level waves (A/B runs, parity of the two forms).
  p->twave = -1;
  for (int l = 0; l < 3; ++l) p->nt[l] = p->tc0[l] = 0;
  if (p->ngroups == 1 && !(g_options[MIFWT_OPT_DEBUG] & 524288)) {
    const int cap[3] = {64 * nc1, 128, 64}, ntmax[3] = {4, 6, 6};
    int* nl[3] = {&p->nl1, &p->nl2, &p->nl3};
    bool any = false;
    for (int l = 0; l < nlev; ++l) {
      const int cols = (int)d[l]->coef_extent[1], full = cols / cap[l], rem = cols - full * cap[l];
      if (full >= 1 && rem >= 1 && rem <= ntmax[l]) {
        p->nt[l] = rem;
        p->tc0[l] = full * cap[l];
        *nl[l] = full;
        any = true;
      }
    }
    // ... in the twelve-wave form only (pyr_role<12>: 168 registers a lane; the sixteen-wave form keeps its level waves)
    const bool fits12 = p->nl1 <= 4 && p->nl2 <= 2 && p->nl3 <= 2 && !(g_options[MIFWT_OPT_DEBUG] & 1048576);
    if (any && fits12) {
      p->twave = 9;
    } else if (any) {
      for (int l = 0; l < 3; ++l) p->nt[l] = p->tc0[l] = 0;
      p->nl1 = (n[1] + 64 * nc1 - 1) / (64 * nc1);
      p->nl2 = nlev >= 2 ? (n[2] + 127) / 128 : 0;
      p->nl3 = nlev >= 3 ? (n[3] + 63) / 64 : 0;
    }
  }
  // twelve waves where they suffice
  p->nwaves = (p->nl1 <= 4 && p->nl2 <= 2 && p->nl3 <= 2 && !(g_options[MIFWT_OPT_DEBUG] & 1048576)) ? 12 : 16;
  p->nchunks = (body + 255) / 256;
  if (p->nchunks < 1 || p->nchunks > kPyrMaxChunks) return false;
  p->pitch0 = (kPyrPad + 256 * p->nchunks + 8) * 4;
  p->pitch1 = nlev >= 2 ? ((kPyrPad + n[1] + HL + 8 + 3) & ~3) * 4 : 0;
  p->pitch2 = nlev >= 3 ? ((kPyrPad + n[2] + HL + 8 + 3) & ~3) * 4 : 0;
  const int rings = (kPyrRing + 1) * (p->pitch1 + p->pitch2) + (p->twave >= 0 ? 3 * kPyrTailBytes : 0);
  // as many staging sub-buffers as fit (the loaders run nbuf - 1 sub-steps ahead; a wave holds at most 63 requests in flight)
  p->nbuf = g_options[MIFWT_OPT_PREFETCH_PAIRS] > 1 ? std::min(8, g_options[MIFWT_OPT_PREFETCH_PAIRS]) : 4;  // (3 .. 6 measured alike on config 2)
  const int lds_cap = 160 * 1024, per_loader = (p->nchunks + 1) / 2;
  while (p->nbuf > 2 && (kPyrCtl + p->nbuf * kPyrSub * p->pitch0 + rings > lds_cap || (p->nbuf - 1) * kPyrSub * per_loader > 63)) --p->nbuf;
  p->lds = kPyrCtl + p->nbuf * kPyrSub * p->pitch0 + rings;
  if (p->lds > lds_cap) return false;
  // one workgroup per CU (the chunk count below is chosen for that; two small workgroups on one CU leave others idle)
  if (p->lds < 82 * 1024) p->lds = 82 * 1024;
  int dev = 0, ncu = 256;
  if (hipGetDevice(&dev) == hipSuccess) {
    static std::atomic<int> ncu_of[64];  // (per device, asked once: the attribute query is a runtime call per plan otherwise)
    int v = ncu_of[dev & 63].load(std::memory_order_relaxed);
    if (v == 0 && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ncu_of[dev & 63].store(v, std::memory_order_relaxed);
    if (v > 0) ncu = v;
  }
  // Row chunks: one per CU and column group, none shorter than kMinRows rows of the last level; MIFWT_OPT_PAIR_ROWS asks for chunks
  // of about that many rows, MIFWT_OPT_PYR_WGS for that many chunks (parity tests of units that start / end anywhere)
  PyrGeom gm;
  gm.L = L;
  gm.nlev = nlev;
  gm.parity = p->ngroups == 1 ? 1 : 0;
  gm.H[0] = (int)d[0]->sig_extent[0];
  for (int l = 1; l <= 3; ++l) gm.H[l] = l <= nlev ? (int)d[l - 1]->coef_extent[0] : 0;
  const int64_t rows = d[0]->batch * (int64_t)HN;
  int64_t gwant = std::max(1, ncu / p->ngroups);
  if (g_options[MIFWT_OPT_PAIR_ROWS] > 0) gwant = (rows + g_options[MIFWT_OPT_PAIR_ROWS] - 1) / g_options[MIFWT_OPT_PAIR_ROWS];
  if (g_options[MIFWT_OPT_PYR_WGS] > 0) gwant = g_options[MIFWT_OPT_PYR_WGS];
  gwant = std::min<int64_t>(gwant, std::max<int64_t>(1, rows / kMinRows));
  pyr_schedule(gm, d[0]->batch, (int)std::min<int64_t>(gwant, kPyrMaxWG), p);
  return p->nwg >= 1 && p->nwg <= kPyrMaxWG && p->wg_start[p->nwg] == (uint32_t)rows;
}

// One call asks for the plan of its geometry three times (the route query of the C ABI, the envelope test, the launch), a call loop
// asks for the same one over and over: the last plan of this thread is kept (descriptors and every option the planner reads compared
// byte for byte).  Host time in front of a launch is time the GPU idles at the start of a short timed loop.
static bool pyr_plan(int nlev, const mifwt_level_desc* const* d, PyrPlan* p) {
  struct Memo {
    bool valid = false, ok = false;
    int nlev = 0, dev = -1;
    int opts[6] = {0, 0, 0, 0, 0, 0};
    mifwt_level_desc desc[3];
    PyrPlan plan;
  };
  static thread_local Memo m;
  const int opts[6] = {g_options[MIFWT_OPT_DEBUG], g_options[MIFWT_OPT_PREFETCH_PAIRS], g_options[MIFWT_OPT_PAIR_ROWS], g_options[MIFWT_OPT_PYR_WGS],
                       exp_word(), 0};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (nlev >= 1 && nlev <= 3 && m.valid && m.nlev == nlev && m.dev == dev && memcmp(m.opts, opts, sizeof(opts)) == 0) {
    bool same = true;
    for (int l = 0; l < nlev && same; ++l) same = memcmp(&m.desc[l], d[l], sizeof(mifwt_level_desc)) == 0;
    if (same) {
      if (m.ok) *p = m.plan;
      return m.ok;
    }
  }
  const bool ok = pyr_plan_compute(nlev, d, p);
  if (nlev >= 1 && nlev <= 3) {
    m.nlev = nlev;
    m.dev = dev;
    memcpy(m.opts, opts, sizeof(opts));
    for (int l = 0; l < nlev; ++l) m.desc[l] = *d[l];
    m.ok = ok;
    if (ok) m.plan = *p;
    m.valid = true;
  }
  return ok;
}

bool dwt2_fwd_pyr_supported(int nlev, const mifwt_level_desc* const* d) {
  if (nlev < 1 || nlev > 3 || g_options[MIFWT_OPT_PAIR_MODE] == 2 || g_options[MIFWT_OPT_PYRAMID_MODE] == 2) return false;
  const mifwt_level_desc* d0 = d[0];
  const int L = d0->filt_len;
  if (d0->ndim != 2 || d0->dtype != MIFWT_F32 || L < 2 || L > 10 || (L & 1)) return false;
  // ten taps and the periodic extension: ONE level per launch (round 5) — a single level needs nothing from the far side of the plane
  // but index maps (rows through the loader's map, pad columns from the staged row itself); the rings of a second level would need
  // rows of the plane's other end (periodic) / 32 rows a level (ten taps: more LDS than a CU has)
  if ((L == 10 || d0->mode == MIFWT_MODE_PERIODIC) && nlev != 1) return false;
  if (d0->mode < 0 || d0->mode > MIFWT_MODE_SYMMETRIC) return false;
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
  if (L == 10 && p.nwaves != 12) return false;
  // Where it pays (tools/pyr_where.py, profiles/r03m_pyr_where.txt; round 2: tools/pyr_matrix.py, tools/pyr_big.py;
  // MIFWT_OPT_PYRAMID_MODE 1 overrides): planes of 448 .. ~2560 columns, i.e. one or two column groups.  A workgroup then reads whole
  // rows (or halves of them), one after the other.  Four column groups (4096 columns: 4 KB pieces 16 KB apart) ran at 0.44 of the HBM
  // peak against 0.65 for the per-level tile kernel; narrower planes leave most lanes of the level-2 / 3 waves idle (256^2: 75 against 55 us).
  if (g_options[MIFWT_OPT_PYRAMID_MODE] != 1 && (p.ngroups > 2 || d0->sig_extent[1] < 448)) return false;
  // ONE level alone pays on planes of about a thousand columns (64 x 1024^2 db4: 83.6 against 105 us for the tile kernel; equal at
  // 515^2, behind at 1400^2: tools/fwd1_probe.py, profiles/r04r_fwd1_probe.txt)
  if (g_options[MIFWT_OPT_PYRAMID_MODE] != 1 && nlev == 1 && (d0->sig_extent[1] < 896 || d0->sig_extent[1] > 1280 || d0->sig_extent[0] < 256)) return false;
  return true;
}

int dwt2_fwd_pyr_schedule(int nlev, const mifwt_level_desc* const* d, uint32_t* wg_start, int capacity) {
  if (!dwt2_fwd_pyr_supported(nlev, d)) return MIFWT_ERR_UNSUPPORTED;
  PyrPlan p;
  if (!pyr_plan(nlev, d, &p)) return MIFWT_ERR_UNSUPPORTED;
  if (capacity < p.nwg + 1) return MIFWT_ERR_BADARG;
  std::copy(p.wg_start, p.wg_start + p.nwg + 1, wg_start);
  return p.nwg;
}

template <int L, int NLEV>
static int launch_pyr(const mifwt_level_desc* const* d, const void* x, void* const* const* details, void* approx, const double* lo,
                       const double* hi, hipStream_t stream) {
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
  a.cpg0 = p.cpg0;
  a.cpg = p.cpg;
  a.nchunks = p.nchunks;
  a.nbuf = p.nbuf;
  a.pitch0 = p.pitch0;
  a.pitch1 = p.pitch1;
  a.pitch2 = p.pitch2;
  a.nl1 = p.nl1;
  a.nl2 = p.nl2;
  a.nl3 = p.nl3;
  for (int l = 0; l < 3; ++l) a.nt[l] = p.nt[l], a.tc0[l] = p.tc0[l];
  a.twave = p.twave;
  a.mode = d[0]->mode;
  a.dbg = g_options[MIFWT_OPT_DEBUG];
  // (timing experiments, results wrong: the nearly empty last wave of level 1 / 2 / 3 switched off)
  if ((MIFWT_DBG(a) & 65536) && a.nl1 > 1) --a.nl1;
  if ((MIFWT_DBG(a) & 131072) && a.nl2 > 1) --a.nl2;
  if ((MIFWT_DBG(a) & 262144) && a.nl3 > 1) --a.nl3;
  a.prof = g_pyr_prof;
  a.l2split = (g_options[MIFWT_OPT_DEBUG] & 8192) ? 0 : 1;
  a.exp = exp_word();
  a.hn_div = make_fastdiv((uint32_t)a.H[NLEV]);
  std::copy(p.wg_start, p.wg_start + p.nwg + 1, a.wg_start);
  for (int k = p.nwg + 1; k <= kPyrMaxWG; ++k) a.wg_start[k] = a.wg_start[p.nwg];
  for (int m = 0; m < L; ++m) a.tap[m] = (f2){(float)lo[m], (float)hi[m]};
  a.dt = dev_tap_arg(L);
  const int64_t nwg = (int64_t)p.nwg * p.ngroups;
  // (the per-wave cycle profile of tools/pyr_prof.py exists for the three-level 8-tap kernel only)
  constexpr bool kCanProf = kDiag && L == 8 && NLEV == 3;  // (-DMIFWT_DIAG builds)
  constexpr bool kHas16 = L <= 8;  // (ten taps: the twelve-wave form only — the level-1 waves want 157 registers a lane)
  static DynLdsOnce lds_once12, lds_once16, lds_once_prof12, lds_once_prof16;
  if (!lds_once12.ensure(reinterpret_cast<const void*>(&dwt2_fwd_pyr_kernel<L, NLEV, false, 12>), 160 * 1024)) return MIFWT_ERR_LAUNCH;
  if (kCanProf && !lds_once_prof12.ensure(reinterpret_cast<const void*>(&dwt2_fwd_pyr_kernel<L, NLEV, kCanProf, 12>), 160 * 1024)) return MIFWT_ERR_LAUNCH;
  count_launch(MIFWT_VARIANT_FWD_PYR_ST8);
  const dim3 grid((unsigned)nwg), block(64 * p.nwaves);
  if (a.dt.lo) {  // device-resident taps: the one-level form only (a learnable bank goes level by level), twelve or sixteen waves
    if constexpr (NLEV == 1) {
      static DynLdsOnce lds_dt12, lds_dt16;
      if (p.nwaves == 12) {
        if (!lds_dt12.ensure(reinterpret_cast<const void*>(&dwt2_fwd_pyr_kernel<L, 1, false, 12, true>), 160 * 1024)) return MIFWT_ERR_LAUNCH;
        hipLaunchKernelGGL((dwt2_fwd_pyr_kernel<L, 1, false, 12, true>), grid, block, p.lds, stream, a);
      } else if constexpr (kHas16) {
        if (!lds_dt16.ensure(reinterpret_cast<const void*>(&dwt2_fwd_pyr_kernel<L, 1, false, 16, true>), 160 * 1024)) return MIFWT_ERR_LAUNCH;
        hipLaunchKernelGGL((dwt2_fwd_pyr_kernel<L, 1, false, 16, true>), grid, block, p.lds, stream, a);
      } else {
        return MIFWT_ERR_UNSUPPORTED;
      }
      return hipGetLastError() == hipSuccess ? MIFWT_OK : MIFWT_ERR_LAUNCH;
    }
    return MIFWT_ERR_UNSUPPORTED;
  }
  if (p.nwaves == 12) {
    if (kCanProf && MIFWT_PROFP(a)) hipLaunchKernelGGL((dwt2_fwd_pyr_kernel<L, NLEV, kCanProf, 12>), grid, block, p.lds, stream, a);
    else hipLaunchKernelGGL((dwt2_fwd_pyr_kernel<L, NLEV, false, 12>), grid, block, p.lds, stream, a);
  } else if constexpr (kHas16) {
    if (!lds_once16.ensure(reinterpret_cast<const void*>(&dwt2_fwd_pyr_kernel<L, NLEV, false, 16>), 160 * 1024)) return MIFWT_ERR_LAUNCH;
    if (kCanProf && !lds_once_prof16.ensure(reinterpret_cast<const void*>(&dwt2_fwd_pyr_kernel<L, NLEV, kCanProf, 16>), 160 * 1024)) return MIFWT_ERR_LAUNCH;
    if (kCanProf && MIFWT_PROFP(a)) hipLaunchKernelGGL((dwt2_fwd_pyr_kernel<L, NLEV, kCanProf, 16>), grid, block, p.lds, stream, a);
    else hipLaunchKernelGGL((dwt2_fwd_pyr_kernel<L, NLEV, false, 16>), grid, block, p.lds, stream, a);
  } else {
    return MIFWT_ERR_UNSUPPORTED;
  }
  return hipGetLastError() == hipSuccess ? MIFWT_OK : MIFWT_ERR_LAUNCH;
}

template <int L>
static int launch_pyr_l(int nlev, const mifwt_level_desc* const* d, const void* x, void* const* const* details, void* approx,
                         const double* lo, const double* hi, hipStream_t stream) {
  switch (nlev) {
    case 1: return launch_pyr<L, 1>(d, x, details, approx, lo, hi, stream);
    case 2: return launch_pyr<L, 2>(d, x, details, approx, lo, hi, stream);
    case 3: return launch_pyr<L, 3>(d, x, details, approx, lo, hi, stream);
    default: return MIFWT_ERR_UNSUPPORTED;
  }
}

int dwt2_fwd_pyr(int nlev, const mifwt_level_desc* const* d, const void* x, void* const* const* details, void* approx,
                  const double* lo, const double* hi, hipStream_t stream) {
  if (!dwt2_fwd_pyr_supported(nlev, d)) return MIFWT_ERR_UNSUPPORTED;
  switch (d[0]->filt_len) {
    case 2: return launch_pyr_l<2>(nlev, d, x, details, approx, lo, hi, stream);
    case 4: return launch_pyr_l<4>(nlev, d, x, details, approx, lo, hi, stream);
    case 6: return launch_pyr_l<6>(nlev, d, x, details, approx, lo, hi, stream);
    case 8: return launch_pyr_l<8>(nlev, d, x, details, approx, lo, hi, stream);
    case 10: return nlev == 1 ? launch_pyr<10, 1>(d, x, details, approx, lo, hi, stream) : MIFWT_ERR_UNSUPPORTED;
    default: return MIFWT_ERR_UNSUPPORTED;
  }
}

}  // namespace mifwt
