// mifwt_dwt2_fwd_small.hip — EVERY level of a 2-D decomposition of a small plane in one launch (gfx950), kernel id 20.
//
// Reference seam: all trips of wavedec2's level loop (src/ptwt/conv_transform_2.py:142-149: _fwt_pad2 + F.conv2d(stride 2) +
// split per level) for planes small enough to live in LDS — image patches, the deep levels of a pyramid, packet nodes
// (SURVEY.md §8f-4).  One kernel per level spends a launch (and a write + read of the approximation) on a few kilobytes per
// image: 1024 x 128^2 db4 level 3 took 81 us in three launches for 134 MB of compulsory traffic.  Here a workgroup (256-1024
// threads by its LDS share) owns one image at a time: it parks the plane in LDS WITH its boundary extension materialised and
// runs every level on it, so no inner loop ever looks at the boundary:
//   image A [H][PA]       the plane; sample s of a row sits in column O + s, s in [-(L-2), 2 Wo), the pads copied from the samples the
//                         boundary index map names (tables of (destination, source) offsets per level, built once per workgroup;
//                         a first version evaluated the map per tap of every edge output and spent 40 % of its time there)
//   horizontal pass       (lo, hi)[r][k] = sum_m (h_lo, h_hi)[m] A[r][2k + 1 - m]  ->  image B row L-2 + r, (lo, hi) of a column side by side;
//                         a lane owns one (r, k): L/2 aligned 8-byte LDS reads, L packed FMAs
//   image B [2 Ho + L-2][2 Wo]   pad ROWS filled by copying the row the map names
//   vertical pass         a lane owns one (kr, c) and makes all four bands: (aa, da) from the lo, (ad, dd) from the hi;
//                         the three details go to HBM through buffer stores, aa into image A as the next level's plane (to HBM on
//                         the last level)
// Work items are flattened (magic-number division), so narrow planes still fill the lanes.  The grid is persistent (one
// workgroup per resident slot); the NEXT image's samples are loaded into registers (<= 8 quads per lane) before the levels of the
// current one run, so its HBM latency is hidden even where one workgroup fills a CU's LDS.
//   c_lo/hi[k] = sum_m h_lo/hi[m] x_ext[2k + 1 - m],  k < floor((n + L - 1) / 2)  per axis     
// Same sums as one mifwt_dwt_fwd call per level (summation order differs: agreement to rounding).
// f32, even L <= 20, any boundary mode (periodic included), unit innermost strides, up to 8 levels, planes whose two LDS images
// fit into 160 KB (128 x 128 up to 12 taps).  Algorithmic traffic: the input once, every returned coefficient once.
#include <type_traits>

#include "mifwt_pyr.h"
#include "mifwt_stream.h"

namespace mifwt {

namespace {

constexpr int kSmallMaxThreads = 1024;
constexpr int kSmallMaxLevels = 8;
constexpr int kSmallLdsBytes = 160 * 1024;  // all of a CU's LDS
constexpr int kParkDepth = 8;  // quads (or single samples) of the next image a lane holds in registers

template <int L>
struct SmallArgs {
  const float* x;
  float* det[kSmallMaxLevels][3];  // [level][band ad, da, dd]
  float* approx;                   // band aa of the last level
  int64_t xs_b, ds_b[kSmallMaxLevels], as_b;
  int xs_h, ds_h[kSmallMaxLevels], as_h;
  int H[kSmallMaxLevels + 1], W[kSmallMaxLevels + 1];
  int PA[kSmallMaxLevels];                            // row pitch of image A at each level (multiple of 4)
  FastDiv div_park;                                   // by W[0] / 4 (vec) or W[0]
  FastDiv div_wo[kSmallMaxLevels], div_pc[kSmallMaxLevels];  // by Wo, by the pad samples of a row of A
  int nlevels, mode, cap_a, vec, dbg;  // cap_a: floats of LDS image A (image B follows)
  int64_t batch;
  f2 tap[L];                           // (dec_lo[m], dec_hi[m])
};

#define MIFWT_PARK8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
static_assert(kParkDepth == 8, "MIFWT_PARK8");

template <int L>
__global__ void __launch_bounds__(kSmallMaxThreads) dwt2_fwd_small_kernel(const SmallArgs<L> a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char small_lds[];
  constexpr int TP = 2 * (L - 2) + 2;  // entries of one pad table (a row of image A has at most 2 (L - 2) + 1 pad samples)
  int2* const tbl = reinterpret_cast<int2*>(small_lds);  // [level][pad columns of A | pad rows of B][TP] x (destination, source) byte offsets
  float* A = reinterpret_cast<float*>(small_lds + a.nlevels * 2 * TP * (int)sizeof(int2));
  float* B = A + a.cap_a;
  const uint32_t tid = threadIdx.x, nt = blockDim.x;
  constexpr int P = L - 2;          // samples a window reaches past either end of a row (one more at the far end of odd rows)
  constexpr int O = (P + 3) & ~3;   // column of sample 0 in image A
  const int W0 = a.W[0], PA0 = a.PA[0];
  const FastDiv dvp = a.div_park;
  const uint32_t n4 = (uint32_t)(a.H[0] * (W0 >> 2)), n1 = (uint32_t)(a.H[0] * W0);

  // The quads of the plane a lane moves: tid, tid + nt, ... (at most kParkDepth); (row, column) of the first by one division, of the
  // others by fixed steps plus a carry.  Lanes past the end re-read quad 0 (a conditional load pushes the registers into scratch).
  uint32_t pk_c;
  const uint32_t pk_r = dvp.divmod(tid, pk_c);
  uint32_t pk_dc;
  const uint32_t pk_dr = dvp.divmod(nt, pk_dc);
  const uint32_t w4 = (uint32_t)W0 >> 2;
  const uint32_t pk_g = pk_r * (uint32_t)a.xs_h + 4 * pk_c, pk_l = (pk_r * PA0 + O + 4 * pk_c) * 4;  // floats in HBM, bytes in LDS
  const uint32_t pk_g0 = pk_dr * (uint32_t)a.xs_h + 4 * pk_dc, pk_g1 = pk_g0 + (uint32_t)a.xs_h - W0;
  const uint32_t pk_l0 = (pk_dr * PA0 + 4 * pk_dc) * 4, pk_l1 = pk_l0 + (PA0 - W0) * 4;
  // The boundary extension of every level, once per workgroup (the geometry is the same for every image): where each pad sample of
  // a row of image A / each pad row of image B goes and which sample / row the boundary map names for it (-1: zero).  The fills
  // then cost two LDS accesses and a table entry per pad sample; with the map evaluated per sample they took as long as both filter
  // passes together (4096 x 64^2 db4: 30 of 80 us).
  for (int l = 0; l < a.nlevels; ++l) {
    const int H = a.H[l], W = a.W[l], Ho = a.H[l + 1], Wo = a.W[l + 1];
    const int npc = (L - 2) + 2 * Wo - W, npr = (L - 2) + 2 * Ho - H, PBb = 2 * Wo * 4;
    constexpr int P = L - 2, O = (P + 3) & ~3;
    int2* ta = tbl + l * 2 * TP;
    int2* tb = ta + TP;
    for (int j = tid; j < npc; j += nt) {
      const int s_ = j < P ? j - P : W + j - P;
      const int src = ext_index(s_, W, a.mode);
      ta[j] = make_int2((O + s_) * 4, src >= 0 ? (O + src) * 4 : -1);
    }
    for (int j = tid; j < npr; j += nt) {
      const int s_ = j < P ? j - P : H + j - P;
      const int src = ext_index(s_, H, a.mode);
      tb[j] = make_int2((P + s_) * PBb, src >= 0 ? (P + src) * PBb : -1);
    }
  }
  __syncthreads();
#define MIFWT_PARK_DECL(u) float4 q##u = {0.f, 0.f, 0.f, 0.f};
  MIFWT_PARK8(MIFWT_PARK_DECL)
#undef MIFWT_PARK_DECL
#define MIFWT_PARK_LD(u)                                                                    \
  q##u = *reinterpret_cast<const float4*>(xi + (tid + u * nt < n4 ? g : 0u));               \
  c += pk_dc;                                                                               \
  g += c >= w4 ? pk_g1 : pk_g0;                                                             \
  c -= c >= w4 ? w4 : 0u;
  if (a.vec && !(MIFWT_DBG(a) & 2)) {
    const float* xi = a.x + (int64_t)blockIdx.x * a.xs_b;
    uint32_t c = pk_c, g = pk_g;
    MIFWT_PARK8(MIFWT_PARK_LD)
  }

  for (int64_t img = blockIdx.x; img < a.batch; img += gridDim.x) {
    // ---- park the plane: A[r * PA0 + O + c] --------------------------------------------------------------------------------
    if (MIFWT_DBG(a) & 2) {
    } else if (a.vec) {  // rows start on 16-byte boundaries and hold whole quads; the plane is at most kParkDepth quads per lane
#define MIFWT_PARK_ST(u)                                                                    \
  if (tid + u * nt < n4) *reinterpret_cast<float4*>(reinterpret_cast<char*>(A) + lo) = q##u; \
  c += pk_dc;                                                                               \
  lo += c >= w4 ? pk_l1 : pk_l0;                                                            \
  c -= c >= w4 ? w4 : 0u;
      {
        uint32_t c = pk_c, lo = pk_l;
        MIFWT_PARK8(MIFWT_PARK_ST)
      }
#undef MIFWT_PARK_ST
      const int64_t img_ld = img + gridDim.x;  // in flight while this image's levels run
      if (img_ld < a.batch) {
        const float* xi = a.x + img_ld * a.xs_b;
        uint32_t c = pk_c, g = pk_g;
        MIFWT_PARK8(MIFWT_PARK_LD)
      }
    } else {
      const float* __restrict__ xb = a.x + img * a.xs_b;
      auto one = [&](uint32_t it) {
        uint32_t c;
        const uint32_t r = dvp.divmod(min(it, n1 - 1), c);
        return xb[(int64_t)r * a.xs_h + c];
      };
      for (uint32_t base = tid; base < n1; base += kParkDepth * nt) {
#define MIFWT_PARK_LD1(u) const float v##u = one(base + u * nt);
        MIFWT_PARK8(MIFWT_PARK_LD1)
#undef MIFWT_PARK_LD1
#define MIFWT_PARK_ST1(u)                                   \
  if (base + u * nt < n1) {                                 \
    uint32_t c;                                             \
    const uint32_t r = dvp.divmod(base + u * nt, c);        \
    A[r * PA0 + O + c] = v##u;                              \
  }
        MIFWT_PARK8(MIFWT_PARK_ST1)
#undef MIFWT_PARK_ST1
      }
    }
    __syncthreads();

    for (int l = 0; l < a.nlevels; ++l) {
      const int H = a.H[l], W = a.W[l], Ho = a.H[l + 1], Wo = a.W[l + 1];
      const int PA = a.PA[l], PB = 2 * Wo;
      const bool last = l == a.nlevels - 1;
      // ---- pad columns of image A: item = (row r, pad sample j) ---------------------------------------------------------------
      {
        const uint32_t npc = (uint32_t)(P + 2 * Wo - W);
        const int2* ta = tbl + l * 2 * TP;
        uint32_t j;
        const uint32_t r0 = a.div_pc[l].divmod(tid, j);
        uint32_t dj;
        const uint32_t dr = a.div_pc[l].divmod(nt, dj);
        uint32_t rowb = r0 * PA * 4;
        const uint32_t rowb0 = dr * PA * 4, rowb1 = rowb0 + PA * 4;
        char* Ab = reinterpret_cast<char*>(A);
        for (uint32_t it = tid; it < (uint32_t)((MIFWT_DBG(a) & 32) ? 0 : H * (int)npc); it += nt) {
          const int2 e = ta[j];
          const float v = e.y >= 0 ? *reinterpret_cast<const float*>(Ab + rowb + e.y) : 0.f;
          *reinterpret_cast<float*>(Ab + rowb + e.x) = v;
          j += dj;
          const bool carry = j >= npc;
          j -= carry ? npc : 0u;
          rowb += carry ? rowb1 : rowb0;
        }
      }
      __syncthreads();
      // ---- horizontal pass: item = (row r, output column k) -------------------------------------------------------------------
      // (One division per pass: the next item of a lane is nt further on, (r, k) and the byte offsets derived from them advance by
      // fixed steps plus a carry.  With a division and the index products per item the integer multiplies — quarter rate — took
      // as long as the filter.)
      {
        uint32_t k;
        const uint32_t r0 = a.div_wo[l].divmod(tid, k);
        uint32_t dk;
        const uint32_t dr = a.div_wo[l].divmod(nt, dk);
        uint32_t src = (r0 * PA + O + 2 * k - P) * 4;          // samples 2k - (L-2) .. 2k + 1 of row r
        uint32_t dst = ((P + r0) * PB + 2 * k) * 4;            // (lo, hi)[r][k], interleaved
        const uint32_t src0 = (dr * PA + 2 * dk) * 4, src1 = src0 + (PA - 2 * Wo) * 4;
        const uint32_t dst0 = (dr * PB + 2 * dk) * 4, dst1 = dst0 + (PB - 2 * Wo) * 4;
        const char* Ab = reinterpret_cast<const char*>(A);
        char* Bb = reinterpret_cast<char*>(B);
        for (uint32_t it = tid; it < (uint32_t)((MIFWT_DBG(a) & 64) ? 0 : H * Wo); it += nt) {
          const f2* w = reinterpret_cast<const f2*>(Ab + src);
          f2 pr[L / 2];
#pragma unroll
          for (int j = 0; j < L / 2; ++j) pr[j] = w[j];  // samples 2k + 1 - m for m = L-1-2j (x), L-2-2j (y)
          f2 acc = {0.f, 0.f};
#pragma unroll
          for (int j = 0; j < L / 2; ++j) {
            vfma_lo(acc, a.tap[L - 1 - 2 * j], pr[j]);
            vfma_hi(acc, a.tap[L - 2 - 2 * j], pr[j]);
          }
          *reinterpret_cast<f2*>(Bb + dst) = acc;
          k += dk;
          const bool carry = k >= (uint32_t)Wo;
          k -= carry ? (uint32_t)Wo : 0u;
          src += carry ? src1 : src0;
          dst += carry ? dst1 : dst0;
        }
      }
      __syncthreads();
      // ---- pad rows of image B: item = (pad row j, column c): the (lo, hi) pair of a column at a time ---------------------------
      {
        const uint32_t npr = (uint32_t)(P + 2 * Ho - H);
        const int2* tb = tbl + l * 2 * TP + TP;
        uint32_t c;
        uint32_t j = a.div_wo[l].divmod(tid, c);
        uint32_t dc;
        const uint32_t dj = a.div_wo[l].divmod(nt, dc);
        char* Bb = reinterpret_cast<char*>(B);
        for (uint32_t it = tid; it < (uint32_t)((MIFWT_DBG(a) & 32) ? 0 : (int)npr * Wo); it += nt) {
          const int2 e = tb[j];
          const f2 v = e.y >= 0 ? *reinterpret_cast<const f2*>(Bb + e.y + c * 8) : (f2){0.f, 0.f};
          *reinterpret_cast<f2*>(Bb + e.x + c * 8) = v;
          c += dc;
          const bool carry = c >= (uint32_t)Wo;
          c -= carry ? (uint32_t)Wo : 0u;
          j += dj + (carry ? 1u : 0u);
        }
      }
      __syncthreads();
      // ---- vertical pass: item = (output row kr, column c); all four bands ------------------------------------------------------
      {
        const uint32_t dspan = (MIFWT_DBG(a) & 1) ? 0u : (uint32_t)Ho * (uint32_t)a.ds_h[l] * 4u;  // (a resource of no bytes drops its stores)
        const rsrc_t r_ad = pyr_rsrc(a.det[l][0] + img * a.ds_b[l], dspan);
        const rsrc_t r_da = pyr_rsrc(a.det[l][1] + img * a.ds_b[l], dspan);
        const rsrc_t r_dd = pyr_rsrc(a.det[l][2] + img * a.ds_b[l], dspan);
        const rsrc_t r_aa = pyr_rsrc(a.approx + img * a.as_b, last && !(MIFWT_DBG(a) & 1) ? (uint32_t)Ho * (uint32_t)a.as_h * 4u : 0u);
        const uint32_t pa_n = last ? (uint32_t)a.as_h : (uint32_t)a.PA[l + 1];  // row pitch of where aa goes
        const uint32_t ds_h = (uint32_t)a.ds_h[l], pb4 = (uint32_t)PB * 4u;
        uint32_t c;
        const uint32_t kr0 = a.div_wo[l].divmod(tid, c);
        uint32_t dc;
        const uint32_t dkr = a.div_wo[l].divmod(nt, dc);
        uint32_t src = ((P + 2 * kr0 + 1) * PB + 2 * c) * 4;                  // (lo, hi) of column c in sample row 2 kr + 1
        uint32_t dst = (kr0 * ds_h + c) * 4;                                   // the detail bands
        uint32_t dsa = (kr0 * pa_n + (last ? 0u : (uint32_t)O) + c) * 4;       // aa: HBM on the last level, image A before
        const uint32_t src0 = (2 * dkr * PB + 2 * dc) * 4, src1 = src0 + (2 * PB - 2 * Wo) * 4;
        const uint32_t dst0 = (dkr * ds_h + dc) * 4, dst1 = dst0 + (ds_h - Wo) * 4;
        const uint32_t dsa0 = (dkr * pa_n + dc) * 4, dsa1 = dsa0 + (pa_n - Wo) * 4;
        const char* Bb = reinterpret_cast<const char*>(B);
        char* Ab = reinterpret_cast<char*>(A);
        auto run = [&](auto last_tag) {  // (two copies of the loop, so that its body is one straight block)
          constexpr bool kLast = decltype(last_tag)::value;
          for (uint32_t it = tid; it < (uint32_t)((MIFWT_DBG(a) & 128) ? 0 : Ho * Wo); it += nt) {
            f2 pr[L];
            uint32_t rd = src;
#pragma unroll
            for (int m = 0; m < L; ++m) {  // sample row 2 kr + 1 - m sits one row of B further up per m
              pr[m] = *reinterpret_cast<const f2*>(Bb + rd);
              rd -= pb4;
            }
            f2 lo = {0.f, 0.f}, hi = {0.f, 0.f};
#pragma unroll
            for (int m = 0; m < L; ++m) {
              lo += a.tap[m] * pr[m].x;  // (aa, da) from the horizontal lo
              hi += a.tap[m] * pr[m].y;  // (ad, dd) from the horizontal hi
            }
            __builtin_amdgcn_sched_group_barrier(0x100, L, 0);  // all L reads in flight before the first FMA waits for one
            __builtin_amdgcn_sched_group_barrier(0x002, 2 * L, 0);
            if constexpr (kLast)
              pyr_store1(lo.x, r_aa, dsa, 0);
            else
              *reinterpret_cast<float*>(Ab + dsa) = lo.x;  // the next level's plane (image A is dead since the horizontal pass)
            pyr_store1(hi.x, r_ad, dst, 0);
            pyr_store1(lo.y, r_da, dst, 0);
            pyr_store1(hi.y, r_dd, dst, 0);
            c += dc;
            const bool carry = c >= (uint32_t)Wo;
            c -= carry ? (uint32_t)Wo : 0u;
            src += carry ? src1 : src0;
            dst += carry ? dst1 : dst0;
            dsa += carry ? dsa1 : dsa0;
          }
        };
        if (last)
          run(std::true_type{});
        else
          run(std::false_type{});
      }
      __syncthreads();
    }
  }
#undef MIFWT_PARK_LD
}
#undef MIFWT_PARK8

struct SmallPlan {
  int cap_a, lds, threads, grid;
};

constexpr int small_origin(int L) { return (L - 2 + 3) & ~3; }

bool small_plan(int nlev, const mifwt_level_desc* const* d, SmallPlan* p) {
  if (nlev < 1 || nlev > kSmallMaxLevels || g_options[MIFWT_OPT_FORCE_GENERIC] || g_options[MIFWT_OPT_PAIR_MODE] == 2 ||
      g_options[MIFWT_OPT_PYRAMID_MODE] == 2)
    return false;
  const mifwt_level_desc* d0 = d[0];
  const int L = d0->filt_len;
  if (d0->ndim != 2 || d0->dtype != MIFWT_F32 || L < 2 || L > 20 || (L & 1)) return false;
  if (d0->mode < 0 || d0->mode > MIFWT_MODE_SYMMETRIC) return false;
  if (d0->batch < 1 || d0->batch > (int64_t(1) << 30) || d0->sig_stride[2] != 1) return false;
  if (d0->sig_extent[0] < 1 || d0->sig_extent[1] < 1 || d0->sig_extent[0] > 4096 || d0->sig_extent[1] > 4096) return false;
  if (d0->sig_stride[1] < 0 || d0->sig_extent[0] * d0->sig_stride[1] >= (int64_t(1) << 30)) return false;  // 32-bit offsets inside an image
  for (int l = 0; l < nlev; ++l) {
    const mifwt_level_desc* dl = d[l];
    if (dl->ndim != 2 || dl->dtype != MIFWT_F32 || dl->filt_len != L || dl->mode != d0->mode || dl->batch != d0->batch) return false;
    if (dl->detail_stride[2] != 1) return false;
    for (int ax = 0; ax < 2; ++ax) {
      const int64_t n = l == 0 ? d0->sig_extent[ax] : d[l - 1]->coef_extent[ax];
      if (dl->sig_extent[ax] != n || dl->coef_extent[ax] != (n + L - 1) / 2) return false;
    }
    if (dl->coef_extent[0] * dl->detail_stride[1] >= (int64_t(1) << 29)) return false;  // byte offsets of a band fit 31 bits
  }
  const mifwt_level_desc* dn = d[nlev - 1];
  if (dn->approx_stride[2] != 1 || dn->coef_extent[0] * dn->approx_stride[1] >= (int64_t(1) << 29)) return false;
  int64_t cap_a = 0, cap_b = 0;  // the largest image of either kind (planes shorter than the filter GROW from level to level)
  for (int l = 0; l < nlev; ++l) {
    const int64_t Ho = d[l]->coef_extent[0], Wo = d[l]->coef_extent[1];
    const int64_t pa = (small_origin(L) + 2 * Wo + 3) & ~int64_t(3);
    cap_a = std::max(cap_a, d[l]->sig_extent[0] * pa);
    cap_b = std::max(cap_b, (L - 2 + 2 * Ho) * 2 * Wo);
  }
  cap_a = (cap_a + 3) & ~int64_t(3);
  const int64_t tbl_bytes = (int64_t)nlev * 2 * (2 * (L - 2) + 2) * 8;  // the pad tables (multiple of 16 bytes)
  if ((cap_a + cap_b) * 4 + tbl_bytes > kSmallLdsBytes) return false;
  p->cap_a = (int)cap_a;
  p->lds = (int)((cap_a + cap_b) * 4 + tbl_bytes);
  // Threads: about four trips of a lane through a filter pass of the finest level (a quarter of its coefficient plane, rounded to
  // a power of two: 64 for 32^2, 128 for 48^2, 256 for 64^2, 512 for 88^2, 1024 for 128^2) — measured with forced counts: more
  // threads idle at the barriers (16384 x 32^2 db2: 59 us with 64 threads against 103 with 256), fewer leave the passes too long
  // (88^2 db4: 65 us with 512 against 74 with 256).  Resident workgroups per CU: by LDS (1 KB + 1/64 of slack: two workgroups of
  // 79.4 KB share a CU, two of 80.8 KB did not) and by waves (the kernels take up to 128 VGPRs: 16 waves per CU; a grid of three 512-thread
  // workgroups per CU ran as two and then one).
  {
    const double quarter = (double)(d0->coef_extent[0] * d0->coef_extent[1]) / 4.0;
    p->threads = 64;
    while (p->threads < 1024 && quarter >= p->threads * 1.4142) p->threads *= 2;
  }
  const int slots = 160 * 1024 / (p->lds + 1024 + p->lds / 64);
  const int per_cu = std::max(1, std::min(slots, 1024 / p->threads));
  // A plane that keeps a CU's LDS to itself runs its phases back to back; that pays only when the CU gets several images (the next
  // one's load overlaps: 256 x 131^2, one image per CU, 20 us against 17 us for a launch per level) and the plane is big enough for
  // the per-image fixed costs (2048 x 96^2: 82-97 us against 59-75; 1024 x 112^2 even; 120^2 and up ahead: tools/small_time.py probe).
  if (per_cu == 1 && (d0->batch < 2 * 256 || p->lds < 112 * 1024) && g_options[MIFWT_OPT_PYRAMID_MODE] != 3) return false;
  p->grid = (int)std::min<int64_t>(d0->batch, int64_t(256) * per_cu);
  return true;
}

template <int L>
int launch_small(int nlev, const mifwt_level_desc* const* d, const SmallPlan& p, const void* x, void* const* const* details, void* approx,
                 const double* lo, const double* hi, hipStream_t stream) {
  SmallArgs<L> a;
  a.x = static_cast<const float*>(x);
  a.xs_b = d[0]->sig_stride[0];
  a.xs_h = (int)d[0]->sig_stride[1];
  a.H[0] = (int)d[0]->sig_extent[0];
  a.W[0] = (int)d[0]->sig_extent[1];
  for (int l = 0; l < nlev; ++l) {
    for (int b = 0; b < 3; ++b) a.det[l][b] = static_cast<float*>(details[l][b]);
    a.ds_b[l] = d[l]->detail_stride[0];
    a.ds_h[l] = (int)d[l]->detail_stride[1];
    a.H[l + 1] = (int)d[l]->coef_extent[0];
    a.W[l + 1] = (int)d[l]->coef_extent[1];
    a.PA[l] = (small_origin(L) + 2 * a.W[l + 1] + 3) & ~3;
    a.div_wo[l] = make_fastdiv((uint32_t)a.W[l + 1]);
    a.div_pc[l] = make_fastdiv((uint32_t)std::max(1, L - 2 + 2 * a.W[l + 1] - a.W[l]));
  }
  a.approx = static_cast<float*>(approx);
  a.as_b = d[nlev - 1]->approx_stride[0];
  a.as_h = (int)d[nlev - 1]->approx_stride[1];
  a.nlevels = nlev;
  a.mode = d[0]->mode;
  a.batch = d[0]->batch;
  a.cap_a = p.cap_a;
  a.dbg = g_options[MIFWT_OPT_DEBUG];
  a.vec = (a.W[0] % 4 == 0 && a.xs_h % 4 == 0 && a.xs_b % 4 == 0 && reinterpret_cast<uintptr_t>(x) % 16 == 0 &&
           (int64_t)a.H[0] * (a.W[0] / 4) <= (int64_t)kParkDepth * p.threads)
              ? 1
              : 0;
  a.div_park = make_fastdiv((uint32_t)(a.vec ? a.W[0] / 4 : a.W[0]));
  for (int m = 0; m < L; ++m) a.tap[m] = (f2){(float)lo[m], (float)hi[m]};
  static DynLdsOnce lds_once;
  if (!lds_once.ensure(reinterpret_cast<const void*>(&dwt2_fwd_small_kernel<L>), kSmallLdsBytes)) return MIFWT_ERR_LAUNCH;
  hipLaunchKernelGGL((dwt2_fwd_small_kernel<L>), dim3((unsigned)p.grid), dim3(p.threads), p.lds, stream, a);
  return hipGetLastError() == hipSuccess ? MIFWT_OK : MIFWT_ERR_LAUNCH;
}

}  // namespace

bool dwt2_fwd_small_supported(int nlev, const mifwt_level_desc* const* d) {
  SmallPlan p;
  return small_plan(nlev, d, &p);
}

int dwt2_fwd_small(int nlev, const mifwt_level_desc* const* d, const void* x, void* const* const* details, void* approx, const double* lo,
                   const double* hi, hipStream_t stream) {
  SmallPlan p;
  if (!small_plan(nlev, d, &p)) return MIFWT_ERR_UNSUPPORTED;
#define MIFWT_SMALL_CASE(LL) \
  case LL: return launch_small<LL>(nlev, d, p, x, details, approx, lo, hi, stream);
  switch (d[0]->filt_len) {
    MIFWT_SMALL_CASE(2)
    MIFWT_SMALL_CASE(4)
    MIFWT_SMALL_CASE(6)
    MIFWT_SMALL_CASE(8)
    MIFWT_SMALL_CASE(10)
    MIFWT_SMALL_CASE(12)
    MIFWT_SMALL_CASE(14)
    MIFWT_SMALL_CASE(16)
    MIFWT_SMALL_CASE(18)
    MIFWT_SMALL_CASE(20)
    default: return MIFWT_ERR_UNSUPPORTED;
  }
#undef MIFWT_SMALL_CASE
}

}  // namespace mifwt
