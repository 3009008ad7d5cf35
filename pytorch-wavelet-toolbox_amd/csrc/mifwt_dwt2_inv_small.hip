// mifwt_dwt2_inv_small.hip — EVERY level of a 2-D reconstruction of a small plane in one launch (gfx950), kernel id 21.
//
// Reference seam: all trips of waverec2's level loop (src/ptwt/conv_transform_2.py:222-249: torch.stack + F.conv_transpose2d(stride 2)
// + the crops, per level) for planes small enough to live in LDS — the mirror of mifwt_dwt2_fwd_small.hip.  One
// kernel per level spends a launch (and a write + read of the running approximation) on a few kilobytes per image: 4096 x 64^2 db2
// level 3 took 60 us in two launches.  Here a workgroup (256-1024 threads by its LDS share) owns one image at a time:
//   region C [4 planes of Mh x Mw]   the level's coefficients, (aa, da) and (ad, dd) of a position side by side
//   vertical pass     lane = one (row pair p, column c): L/2 8-byte reads per pair plane, 2 L packed FMAs (tap PAIRS (g[2j], g[2j+1]) in SGPRs,
//                     accumulator = (row 2p, row 2p+1)) -> (X_lo, X_hi)[2p .. 2p+1][c] into region X
//   horizontal pass   lane = one (row n, column pair q): L/2 8-byte reads, L packed FMAs -> y[n][2q .. 2q+1]: the next level's aa
//                     (cropped to that level's coefficient extents) back into region C, or HBM on the last level
// per axis, output index n = 2p + r:   y[2p + r] = sum_{i < L/2} g_lo[L-2-2i+r] a[p+i] + g_hi[L-2-2i+r] d[p+i]
// (polyphase form of the transposed convolution, only the cropped interior; same sums as one mifwt_dwt_inv call per level, the
// summation order differs: agreement to rounding).  The detail bands of the NEXT level (the coarsest set of the next image after the
// last level) are requested into registers before the passes of the current one run; the grid is persistent.  Work items are flattened
// over the lanes with one magic-number division per pass (fixed steps plus a carry, as in the analysis kernel).
// f32, even L <= 20, up to 8 levels, dense coefficient planes, the two LDS regions of the finest level <= 160 KB (128 x 128 outputs
// for 8 taps).  Algorithmic traffic: every coefficient once, the output once.
#include <type_traits>

#include "mifwt_pyr.h"
#include "mifwt_stream.h"

namespace mifwt {

namespace {

constexpr int kISMaxThreads = 1024;
constexpr int kISMaxLevels = 8;
constexpr int kISLdsBytes = 160 * 1024;
constexpr int kISDepth = 8;  // samples of one band plane a lane holds in registers (plane <= kISDepth x threads)

template <int L>
struct ISmallArgs {
  const float* aa;                       // coarsest approximation
  const float* det[kISMaxLevels][3];     // [level, coarsest first][band ad, da, dd]
  float* y;
  int64_t aa_b, det_b[kISMaxLevels], ys_b;  // batch strides (floats); planes are dense
  int ys_h;
  int Mh[kISMaxLevels], Mw[kISMaxLevels];  // coefficient extents of a level
  int Hn[kISMaxLevels], Wn[kISMaxLevels];  // output extents of a level (= the next level's coefficient extents; the last one's: y)
  FastDiv div_mw[kISMaxLevels], div_qc[kISMaxLevels];  // by Mw, by ceil(Wn / 2)
  int nlevels, cap_c, yvec;                // cap_c: floats of region C (region X follows); yvec: 8-byte stores into y are aligned
  int64_t batch;
  f2 tlo[L / 2], thi[L / 2];               // (rec_lo[2j], rec_lo[2j+1]), (rec_hi[2j], rec_hi[2j+1])
};

typedef unsigned int u2v __attribute__((ext_vector_type(2)));

template <int L>
__global__ void __launch_bounds__(kISMaxThreads) idwt2_small_kernel(const ISmallArgs<L> a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char ismall_lds[];
  float* C = reinterpret_cast<float*>(ismall_lds);
  float* X = C + a.cap_c;
  const uint32_t tid = threadIdx.x, nt = blockDim.x;
  constexpr int HL = L / 2;
  const int nl = a.nlevels;

  // Band planes of one level -> registers: band b, samples tid, tid + nt, ... (a resource of the plane's bytes: lanes past its end get 0).
  // Slot 3 is the approximation (coarsest level only).
  float q[4][kISDepth];
  auto request = [&](int64_t img, int l, bool with_aa) {
    const uint32_t bytes = (uint32_t)(a.Mh[l] * a.Mw[l]) * 4u;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const float* base = b < 3 ? a.det[l][b] + img * a.det_b[l] : a.aa + img * a.aa_b;
      const rsrc_t rs = pyr_rsrc(base, b < 3 || with_aa ? bytes : 0u);
#pragma unroll
      for (int u = 0; u < kISDepth; ++u)
        q[b][u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (tid + u * nt) * 4u, 0, 0));
    }
  };
  // ... -> region C: (aa, da) pairs first, (ad, dd) pairs behind them
  auto park = [&](int l, bool with_aa) {
    const uint32_t plane = (uint32_t)(a.Mh[l] * a.Mw[l]);
    float* lo = C;
    float* hi = C + 2 * plane;
#pragma unroll
    for (int u = 0; u < kISDepth; ++u) {
      const uint32_t idx = tid + u * nt;
      if (idx < plane) {
        hi[2 * idx] = q[0][u];      // ad
        lo[2 * idx + 1] = q[1][u];  // da
        hi[2 * idx + 1] = q[2][u];  // dd
        if (with_aa) lo[2 * idx] = q[3][u];
      }
    }
  };

  if ((int64_t)blockIdx.x < a.batch) request(blockIdx.x, 0, true);

  for (int64_t img = blockIdx.x; img < a.batch; img += gridDim.x) {
    for (int l = 0; l < nl; ++l) {
      const int Mh = a.Mh[l], Mw = a.Mw[l], Hn = a.Hn[l], Wn = a.Wn[l];
      const bool last = l == nl - 1;
      park(l, l == 0);
      __syncthreads();
      // in flight while this level's passes run: the next level's details, or the coarsest set of the next image
      if (!last)
        request(img, l + 1, false);
      else if (img + gridDim.x < a.batch)
        request(img + gridDim.x, 0, true);
      // ---- vertical pass: item = (output row pair p, column c) --------------------------------------------------------------
      {
        const int PR = (Hn + 1) >> 1;
        uint32_t c;
        const uint32_t p0 = a.div_mw[l].divmod(tid, c);
        uint32_t dc;
        const uint32_t dp = a.div_mw[l].divmod(nt, dc);
        const uint32_t pitch = (uint32_t)Mw * 8u, hi_off = (uint32_t)(Mh * Mw) * 8u;
        uint32_t src = (p0 * Mw + c) * 8;           // (aa, da)[p][c]
        uint32_t dst = (2 * p0 * Mw + c) * 8;       // (X_lo, X_hi)[2p][c]
        const uint32_t src0 = (dp * Mw + dc) * 8, src1 = src0 + 0u;                  // a carry moves to the next row: + Mw - Mw
        const uint32_t dst0 = (2 * dp * Mw + dc) * 8, dst1 = dst0 + (uint32_t)Mw * 8u;  // ... two output rows: + 2 Mw - Mw
        const char* Cb = reinterpret_cast<const char*>(C);
        char* Xb = reinterpret_cast<char*>(X);
        for (uint32_t it = tid; it < (uint32_t)(PR * Mw); it += nt) {
          f2 lo[HL], hi[HL];
          uint32_t rd = src;
#pragma unroll
          for (int i = 0; i < HL; ++i) {
            lo[i] = *reinterpret_cast<const f2*>(Cb + rd);
            hi[i] = *reinterpret_cast<const f2*>(Cb + rd + hi_off);
            rd += pitch;
          }
          f2 xl = {0.f, 0.f}, xh = {0.f, 0.f};  // (row 2p, row 2p + 1) of X_lo / X_hi
#pragma unroll
          for (int i = 0; i < HL; ++i) {
            xl += a.tlo[HL - 1 - i] * lo[i].x + a.thi[HL - 1 - i] * lo[i].y;
            xh += a.tlo[HL - 1 - i] * hi[i].x + a.thi[HL - 1 - i] * hi[i].y;
          }
          __builtin_amdgcn_sched_group_barrier(0x100, 2 * HL, 0);  // all reads in flight before the first FMA waits for one
          __builtin_amdgcn_sched_group_barrier(0x002, 4 * HL, 0);
          *reinterpret_cast<f2*>(Xb + dst) = (f2){xl.x, xh.x};
          *reinterpret_cast<f2*>(Xb + dst + pitch) = (f2){xl.y, xh.y};  // (region X holds an even number of rows)
          c += dc;
          const bool carry = c >= (uint32_t)Mw;
          c -= carry ? (uint32_t)Mw : 0u;
          src += carry ? src1 : src0;
          dst += carry ? dst1 : dst0;
        }
      }
      __syncthreads();
      // ---- horizontal pass: item = (row n, output column pair q) -----------------------------------------------------------
      {
        const int QC = (Wn + 1) >> 1;
        uint32_t qq;
        const uint32_t n0 = a.div_qc[l].divmod(tid, qq);
        uint32_t dq;
        const uint32_t dn = a.div_qc[l].divmod(nt, dq);
        uint32_t src = (n0 * Mw + qq) * 8;  // (X_lo, X_hi)[n][q]
        const uint32_t src0 = (dn * Mw + dq) * 8, src1 = src0 + (uint32_t)(Mw - QC) * 8u;
        // where y[n][2q] goes: floats from the image's first output sample (HBM), or the .x slots of the next level's (aa, da) pairs
        const uint32_t opitch = last ? (uint32_t)a.ys_h : (uint32_t)Wn;
        uint32_t dst = n0 * opitch + 2 * qq;
        const uint32_t dst0 = dn * opitch + 2 * dq, dst1 = dst0 + opitch - 2 * (uint32_t)QC;
        const rsrc_t ry = pyr_rsrc(a.y + img * a.ys_b, last ? (uint32_t)(Hn - 1) * (uint32_t)a.ys_h * 4u + (uint32_t)Wn * 4u : 0u);
        const char* Xb = reinterpret_cast<const char*>(X);
        const bool odd_w = Wn & 1;
        auto run = [&](auto last_tag, auto vec_tag) {
          constexpr bool kLast = decltype(last_tag)::value, kVec = decltype(vec_tag)::value;
          for (uint32_t it = tid; it < (uint32_t)(Hn * QC); it += nt) {
            f2 pr[HL];
#pragma unroll
            for (int i = 0; i < HL; ++i) pr[i] = *reinterpret_cast<const f2*>(Xb + src + i * 8);
            f2 acc = {0.f, 0.f};  // (column 2q, column 2q + 1)
#pragma unroll
            for (int i = 0; i < HL; ++i) acc += a.tlo[HL - 1 - i] * pr[i].x + a.thi[HL - 1 - i] * pr[i].y;
            const bool both = !odd_w || qq + 1 < (uint32_t)QC;  // the last pair of an odd row has one column
            if constexpr (kLast) {
              if constexpr (kVec) {
                if (both)
                  __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2v, acc), ry, dst * 4u, 0, 0);
                else
                  pyr_store1(acc.x, ry, dst * 4u, 0);
              } else {
                pyr_store1(acc.x, ry, dst * 4u, 0);
                pyr_store1(acc.y, ry, both ? dst * 4u + 4u : kPyrOob, 0);
              }
            } else {
              C[2 * dst] = acc.x;
              if (both) C[2 * dst + 2] = acc.y;
            }
            qq += dq;
            const bool carry = qq >= (uint32_t)QC;
            qq -= carry ? (uint32_t)QC : 0u;
            src += carry ? src1 : src0;
            dst += carry ? dst1 : dst0;
          }
        };
        if (!last)
          run(std::false_type{}, std::false_type{});
        else if (a.yvec)
          run(std::true_type{}, std::true_type{});
        else
          run(std::true_type{}, std::false_type{});
      }
      __syncthreads();
    }
  }
}

struct ISmallPlan {
  int cap_c, lds, threads, grid;
};

bool ismall_plan(int nlev, const mifwt_level_desc* const* d, ISmallPlan* p) {
  if (nlev < 1 || nlev > kISMaxLevels || g_options[MIFWT_OPT_FORCE_GENERIC] || g_options[MIFWT_OPT_PAIR_MODE] == 2 ||
      g_options[MIFWT_OPT_PYRAMID_MODE] == 2)
    return false;
  const mifwt_level_desc* d0 = d[0];
  const int L = d0->filt_len;
  if (d0->ndim != 2 || d0->dtype != MIFWT_F32 || L < 2 || L > 20 || (L & 1)) return false;
  if (d0->batch < 1 || d0->batch > (int64_t(1) << 30)) return false;
  int64_t cap_c = 0, cap_x = 0, plane_max = 0;
  for (int l = 0; l < nlev; ++l) {
    const mifwt_level_desc* dl = d[l];
    if (dl->ndim != 2 || dl->dtype != MIFWT_F32 || dl->filt_len != L || dl->batch != d0->batch) return false;
    const int64_t Mh = dl->coef_extent[0], Mw = dl->coef_extent[1];
    if (Mh < L / 2 || Mw < L / 2 || Mh > 4096 || Mw > 4096) return false;
    // dense planes
    if (dl->detail_stride[2] != 1 || dl->detail_stride[1] != Mw) return false;
    if (l == 0 && (dl->approx_stride[2] != 1 || dl->approx_stride[1] != Mw)) return false;
    // output extents: what the next level takes as its approximation (a crop of the full 2M - L + 2), the last one's: y
    for (int ax = 0; ax < 2; ++ax) {
      const int64_t full = 2 * dl->coef_extent[ax] - L + 2;
      const int64_t out = l + 1 < nlev ? d[l + 1]->coef_extent[ax] : dl->sig_extent[ax];
      if (out < 1 || out > full) return false;
      if (l + 1 == nlev && dl->sig_extent[ax] != out) return false;
    }
    const int64_t Hn = l + 1 < nlev ? d[l + 1]->coef_extent[0] : dl->sig_extent[0];
    cap_c = std::max(cap_c, 4 * Mh * Mw);
    cap_x = std::max(cap_x, 2 * ((Hn + 1) & ~int64_t(1)) * Mw);
    plane_max = std::max(plane_max, Mh * Mw);
  }
  const mifwt_level_desc* dn = d[nlev - 1];
  if (dn->sig_stride[2] != 1 || dn->sig_stride[1] < dn->sig_extent[1] || dn->sig_extent[0] * dn->sig_stride[1] >= (int64_t(1) << 29)) return false;
  cap_c = (cap_c + 3) & ~int64_t(3);
  if ((cap_c + cap_x) * 4 > kISLdsBytes) return false;
  p->cap_c = (int)cap_c;
  p->lds = (int)((cap_c + cap_x) * 4);
  // Threads: about four trips of a lane through a filter pass of the finest level (a quarter of its coefficient plane, rounded to
  // a power of two: 64 for 32^2, 128 for 48^2, 256 for 64^2, 512 for 88^2, 1024 for 128^2) — measured with forced counts: more
  // threads idle at the barriers (16384 x 32^2 db2: 59 us with 64 threads against 103 with 256), fewer leave the passes too long
  // (88^2 db4: 65 us with 512 against 74 with 256).  Resident workgroups per CU: by LDS (1 KB + 1/64 of slack: two workgroups of
  // 79.4 KB share a CU, two of 80.8 KB did not) and by waves (the kernels take up to 128 VGPRs: 16 waves per CU; a grid of three 512-thread
  // workgroups per CU ran as two and then one).
  {
    const double quarter = (double)(plane_max) / 4.0;
    p->threads = 64;
    while (p->threads < 1024 && quarter >= p->threads * 1.4142) p->threads *= 2;
    while (p->threads < kISMaxThreads && plane_max > (int64_t)kISDepth * p->threads) p->threads *= 2;  // (the register prefetch holds a plane)
    if (plane_max > (int64_t)kISDepth * p->threads) return false;
  }
  const int slots = 160 * 1024 / (p->lds + 1024 + p->lds / 64);
  const int per_cu = std::max(1, std::min(slots, 1024 / p->threads));
  // (as in the analysis kernel: a plane that keeps a CU's LDS to itself pays only when the CU gets several images and the plane is big —
  // 1024 x 128^2 db4 55 us against 63 level by level, 1024 x 120^2 and 112^2 1-3 us behind, 2048 x 96^2 sym4 77 against 55)
  if (per_cu == 1 && (d0->batch < 2 * 256 || p->lds < 128 * 1024) && g_options[MIFWT_OPT_PYRAMID_MODE] != 3) return false;
  p->grid = (int)std::min<int64_t>(d0->batch, int64_t(256) * per_cu);
  return true;
}

template <int L>
int launch_ismall(int nlev, const mifwt_level_desc* const* d, const ISmallPlan& p, const void* approx, const void* const* const* details,
                  void* y, const double* lo, const double* hi, hipStream_t stream) {
  ISmallArgs<L> a;
  a.aa = static_cast<const float*>(approx);
  a.aa_b = d[0]->approx_stride[0];
  for (int l = 0; l < nlev; ++l) {
    for (int b = 0; b < 3; ++b) a.det[l][b] = static_cast<const float*>(details[l][b]);
    a.det_b[l] = d[l]->detail_stride[0];
    a.Mh[l] = (int)d[l]->coef_extent[0];
    a.Mw[l] = (int)d[l]->coef_extent[1];
    a.Hn[l] = (int)(l + 1 < nlev ? d[l + 1]->coef_extent[0] : d[l]->sig_extent[0]);
    a.Wn[l] = (int)(l + 1 < nlev ? d[l + 1]->coef_extent[1] : d[l]->sig_extent[1]);
    a.div_mw[l] = make_fastdiv((uint32_t)a.Mw[l]);
    a.div_qc[l] = make_fastdiv((uint32_t)((a.Wn[l] + 1) / 2));
  }
  a.y = static_cast<float*>(y);
  a.ys_b = d[nlev - 1]->sig_stride[0];
  a.ys_h = (int)d[nlev - 1]->sig_stride[1];
  a.nlevels = nlev;
  a.cap_c = p.cap_c;
  a.yvec = (a.ys_h % 2 == 0 && a.ys_b % 2 == 0 && reinterpret_cast<uintptr_t>(y) % 8 == 0) ? 1 : 0;
  a.batch = d[0]->batch;
  for (int j = 0; j < L / 2; ++j) {
    a.tlo[j] = (f2){(float)lo[2 * j], (float)lo[2 * j + 1]};
    a.thi[j] = (f2){(float)hi[2 * j], (float)hi[2 * j + 1]};
  }
  static DynLdsOnce lds_once;
  if (!lds_once.ensure(reinterpret_cast<const void*>(&idwt2_small_kernel<L>), kISLdsBytes)) return MIFWT_ERR_LAUNCH;
  hipLaunchKernelGGL((idwt2_small_kernel<L>), dim3((unsigned)p.grid), dim3(p.threads), p.lds, stream, a);
  return hipGetLastError() == hipSuccess ? MIFWT_OK : MIFWT_ERR_LAUNCH;
}

}  // namespace

bool dwt2_inv_small_supported(int nlev, const mifwt_level_desc* const* d) {
  ISmallPlan p;
  return ismall_plan(nlev, d, &p);
}

int dwt2_inv_small(int nlev, const mifwt_level_desc* const* d, const void* approx, const void* const* const* details, void* y,
                   const double* lo, const double* hi, hipStream_t stream) {
  ISmallPlan p;
  if (!ismall_plan(nlev, d, &p)) return MIFWT_ERR_UNSUPPORTED;
#define MIFWT_ISMALL_CASE(LL) \
  case LL: return launch_ismall<LL>(nlev, d, p, approx, details, y, lo, hi, stream);
  switch (d[0]->filt_len) {
    MIFWT_ISMALL_CASE(2)
    MIFWT_ISMALL_CASE(4)
    MIFWT_ISMALL_CASE(6)
    MIFWT_ISMALL_CASE(8)
    MIFWT_ISMALL_CASE(10)
    MIFWT_ISMALL_CASE(12)
    MIFWT_ISMALL_CASE(14)
    MIFWT_ISMALL_CASE(16)
    MIFWT_ISMALL_CASE(18)
    MIFWT_ISMALL_CASE(20)
    default: return MIFWT_ERR_UNSUPPORTED;
  }
#undef MIFWT_ISMALL_CASE
}

}  // namespace mifwt
