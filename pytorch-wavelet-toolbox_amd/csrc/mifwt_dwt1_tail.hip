// mifwt_dwt1_tail.hip — the deep levels of a 1-D decomposition in ONE launch (gfx950), kernel id 14.
//
// Reference seam: the trailing trips of wavedec's level loop (src/ptwt/conv_transform.py:133-140: _fwt_pad + F.conv1d(stride 2)
// per level, the approximation fed back).  Once a row is a few thousand samples long a level is a few microseconds of
// work behind ~5 us of launch latency (the reference's own 1-D speed test, 32 x 10^6 samples at level 10, spends 45 of its
// 166 us in the last five levels; a 1 x 4096 signal at level 12 is launch latency only).  Here a 512-thread workgroup owns one
// row: it parks the row in LDS and runs every remaining level on it — detail coefficients go to HBM, the approximation
// ping-pongs between two LDS buffers and only the last one is stored.  Any boundary mode (the general index map: deep levels
// are shorter than the filter and fold repeatedly), any even filter length up to 32 taps, f32 / f64 in their own precision.
//   c_lo/hi[k] = sum_m h_lo/hi[m] x_ext[2k + 1 - m],  k < floor((n + L - 1) / 2)     
// Traffic is negligible; the point is one launch instead of nlevels.  The same for the COARSE levels of waverec (kernel id 15,
// src/ptwt/conv_transform.py:184-199: stack + conv_transpose1d(stride 2) + crop per level): the row grows in LDS until the
// next level's output would no longer fit.
#include "mifwt_common.h"

namespace mifwt {

namespace {

constexpr int kTailMaxLevels = 24;
constexpr int kTailMaxTaps = 32;
constexpr int kTailThreads = 512;  // at most; a row is one workgroup, short rows get fewer lanes (more workgroups per CU: 16 384 rows
                                   // of 1024 samples ran one latency chain per CU slot with 512 mostly idle lanes each)
static int tail_threads(int64_t n) { return n <= 2048 ? 128 : (n <= 6144 ? 256 : kTailThreads); }

struct Dwt1TailArgs {
  const void* x;
  void* approx;                  // final approximation [rows, m_last]
  void* det[kTailMaxLevels];     // per fused level: detail coefficients [rows, m_l]
  int64_t x_rs, approx_rs, det_rs[kTailMaxLevels];  // row strides (elements)
  int n0, nlevels, filt_len, mode, cap;             // cap: elements of LDS buffer A (buffer B follows)
  double lo[kTailMaxTaps], hi[kTailMaxTaps];        // dec_lo / dec_hi, PyWavelets order
};

template <typename T>
__global__ void __launch_bounds__(kTailThreads) dwt1_tail_kernel(const Dwt1TailArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char tail_lds[];
  T* A = reinterpret_cast<T*>(tail_lds);
  T* B = A + a.cap;
  __shared__ T tlo[kTailMaxTaps], thi[kTailMaxTaps];
  const int tid = threadIdx.x, nt = blockDim.x, L = a.filt_len;
  const int64_t row = blockIdx.x;
  if (tid < L) {
    tlo[tid] = (T)a.lo[tid];
    thi[tid] = (T)a.hi[tid];
  }
  const T* __restrict__ xr = static_cast<const T*>(a.x) + row * a.x_rs;
  for (int i = tid; i < a.n0; i += nt) A[i] = xr[i];
  __syncthreads();
  int n = a.n0;
  for (int lvl = 0; lvl < a.nlevels; ++lvl) {
    const int m = (n + L - 1) >> 1;
    T* __restrict__ dr = static_cast<T*>(a.det[lvl]) + row * a.det_rs[lvl];
    const bool near = n >= L;  // then every index lies within one period: no division in the map
    // interior outputs k in [k_lo, k_hi): every tap reads inside the row (2k + 1 - (L - 1) >= 0 and 2k + 1 < n) — no boundary
    // map, and four outputs per thread and step so that their LDS reads are in flight together (one output at a time ran
    // at LDS latency: 10 us per level on a 15 000-sample row)
    const int k_lo = min((L - 2) >> 1, m), k_hi = max(min(n >> 1, m), k_lo);
    for (int k = k_lo + tid; k < k_hi; k += 4 * nt) {
      T clo[4], chi[4];
      const T* xp[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        clo[u] = chi[u] = T(0);
        const int ku = k + u * nt;
        xp[u] = A + 2 * (ku < k_hi ? ku : k) + 1;  // lanes past the end recompute output k (stored once, below)
      }
      for (int t = 0; t < L; ++t) {
        const T wl = tlo[t], wh = thi[t];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const T xv = xp[u][-t];
          clo[u] = __builtin_fma(wl, xv, clo[u]);
          chi[u] = __builtin_fma(wh, xv, chi[u]);
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int ku = k + u * nt;
        if (ku < k_hi) {
          dr[ku] = chi[u];
          B[ku] = clo[u];
        }
      }
    }
    // the few outputs at the two ends, through the boundary map
    for (int idx = tid; idx < k_lo + (m - k_hi); idx += nt) {
      const int k = idx < k_lo ? idx : k_hi + (idx - k_lo);
      T clo = T(0), chi = T(0);
      for (int t = 0; t < L; ++t) {
        const int s = near ? ext_index_near(2 * k + 1 - t, n, a.mode) : ext_index(2 * k + 1 - t, n, a.mode);
        const T xv = s < 0 ? T(0) : A[s];
        clo = __builtin_fma(tlo[t], xv, clo);
        chi = __builtin_fma(thi[t], xv, chi);
      }
      dr[k] = chi;
      B[k] = clo;
    }
    __syncthreads();
    T* tmp = A;
    A = B;
    B = tmp;
    n = m;
  }
  T* __restrict__ ar = static_cast<T*>(a.approx) + row * a.approx_rs;
  for (int i = tid; i < n; i += nt) ar[i] = A[i];
}

struct Idwt1TailArgs {
  const void* approx;            // coarsest approximation [rows, m0]
  const void* det[kTailMaxLevels];  // per fused level (coarsest first): detail coefficients [rows, m_l]
  void* y;                       // output of the last fused level [rows, out_len[nlevels - 1]]
  int64_t approx_rs, y_rs, det_rs[kTailMaxLevels];
  int out_len[kTailMaxLevels];   // output samples per level (2 m - L + 2 - trim): the next level's coefficient count
  int m0, nlevels, filt_len, cap, cap_small;  // cap / cap_small: elements of the BIG / SMALL (and detail) LDS buffers
  double lo[kTailMaxTaps], hi[kTailMaxTaps];  // rec_lo / rec_hi, PyWavelets order
};

// polyphase synthesis, cropped:  y[2p + r] = sum_{i < L/2} g_lo[L-2-2i+r] a[p+i] + g_hi[L-2-2i+r] d[p+i]  (coefficients past the
// end read as zero), the same formula as the 2-D synthesis kernels apply per axis
template <typename T>
__global__ void __launch_bounds__(kTailThreads) idwt1_tail_kernel(const Idwt1TailArgs a) {
  // LDS: a BIG buffer (the last level's output at most), a SMALL one (its input) and the detail row of the level at hand.  The
  // walk alternates between BIG and SMALL and starts so that the last level reads SMALL; the last level stores straight to
  // global memory.  (Reading the detail coefficients from global memory inside the tap loop — one dependent load per tap —
  // made the four coarse levels of 32 rows of 15 633 samples a 93 us launch.)
  extern __shared__ __attribute__((aligned(16))) unsigned char tail_lds[];
  T* big = reinterpret_cast<T*>(tail_lds);
  T* small = big + a.cap;
  T* D = small + a.cap_small;
  __shared__ T tlo[kTailMaxTaps], thi[kTailMaxTaps];
  const int tid = threadIdx.x, nt = blockDim.x, L = a.filt_len, HLn = a.filt_len >> 1;
  const int64_t row = blockIdx.x;
  if (tid < L) {
    tlo[tid] = (T)a.lo[tid];
    thi[tid] = (T)a.hi[tid];
  }
  T* A = (a.nlevels & 1) ? small : big;  // level l reads SMALL when nlevels - 1 - l is even
  T* B = (a.nlevels & 1) ? big : small;
  const T* __restrict__ ar = static_cast<const T*>(a.approx) + row * a.approx_rs;
  for (int i = tid; i < a.m0 + HLn; i += nt) A[i] = i < a.m0 ? ar[i] : T(0);  // (zeros behind the row: what the windows read past it)
  T* __restrict__ yr = static_cast<T*>(a.y) + row * a.y_rs;
  int m = a.m0;
  for (int lvl = 0; lvl < a.nlevels; ++lvl) {
    const int n = a.out_len[lvl];
    const bool last = lvl == a.nlevels - 1;
    const T* __restrict__ dr = static_cast<const T*>(a.det[lvl]) + row * a.det_rs[lvl];
    for (int i = tid; i < m + HLn; i += nt) D[i] = i < m ? dr[i] : T(0);
    __syncthreads();  // (also: A complete)
    // a thread owns a position p: the outputs 2p and 2p + 1 share their L/2 coefficient pairs
    for (int p = tid; 2 * p < n + (last ? 0 : HLn); p += nt) {
      T acc0 = T(0), acc1 = T(0);
      if (2 * p < n) {
        for (int i = 0; i < HLn; ++i) {
          const T av = A[p + i], dv = D[p + i];
          acc0 = __builtin_fma(tlo[L - 2 - 2 * i], av, acc0);
          acc0 = __builtin_fma(thi[L - 2 - 2 * i], dv, acc0);
          acc1 = __builtin_fma(tlo[L - 1 - 2 * i], av, acc1);
          acc1 = __builtin_fma(thi[L - 1 - 2 * i], dv, acc1);
        }
      }
      if (last) {
        yr[2 * p] = acc0;
        if (2 * p + 1 < n) yr[2 * p + 1] = acc1;
      } else {  // (+ zeros behind the row for the next level's windows)
        B[2 * p] = acc0;
        B[2 * p + 1] = 2 * p + 1 < n ? acc1 : T(0);
      }
    }
    __syncthreads();
    T* tmp = A;
    A = B;
    B = tmp;
    m = n;
  }
}

}  // namespace

// rows of at most this many samples can start the fused tail (both LDS buffers within 96 KB)
int dwt1_tail_max_n(int dtype) { return dtype == MIFWT_F64 ? 8192 : 16384; }

bool dwt1_tail_supported(int dtype, int filt_len, int mode, int64_t rows, int64_t n0, int nlevels) {
  if (g_options[MIFWT_OPT_FORCE_GENERIC] || g_options[MIFWT_OPT_PAIR_MODE] == 2) return false;
  if (dtype != MIFWT_F32 && dtype != MIFWT_F64) return false;
  if (filt_len < 2 || filt_len > kTailMaxTaps || (filt_len & 1)) return false;
  if (mode < MIFWT_MODE_ZERO || mode > MIFWT_MODE_SYMMETRIC) return false;
  if (rows < 1 || rows > (int64_t(1) << 30) || nlevels < 2 || nlevels > kTailMaxLevels) return false;
  return n0 >= 1 && n0 <= dwt1_tail_max_n(dtype);
}

int dwt1_tail(int dtype, int filt_len, int mode, int64_t rows, int64_t n0, int nlevels, const void* x, int64_t x_row_stride,
              void* approx, int64_t approx_row_stride, void* const* details, const int64_t* detail_row_strides, const double* lo,
              const double* hi, hipStream_t stream) {
  if (!dwt1_tail_supported(dtype, filt_len, mode, rows, n0, nlevels)) return MIFWT_ERR_UNSUPPORTED;
  Dwt1TailArgs a;
  a.x = x;
  a.approx = approx;
  a.x_rs = x_row_stride;
  a.approx_rs = approx_row_stride;
  for (int l = 0; l < nlevels; ++l) {
    a.det[l] = details[l];
    a.det_rs[l] = detail_row_strides[l];
  }
  a.n0 = (int)n0;
  a.nlevels = nlevels;
  a.filt_len = filt_len;
  a.mode = mode;
  for (int t = 0; t < filt_len; ++t) {
    a.lo[t] = lo[t];
    a.hi[t] = hi[t];
  }
  const int esz = dtype == MIFWT_F64 ? 8 : 4;
  // buffer A holds the row, buffer B its first approximation; later levels alternate.  A level of fewer than L - 1 samples
  // GROWS (m = (n + L - 1) / 2 > n), but never beyond L: both buffers hold at least 32 elements
  a.cap = (int)n0 < 32 ? 32 : (((int)n0 + 3) & ~3);
  const int m0 = ((int)n0 + filt_len - 1) >> 1;
  const size_t lds = (size_t)(a.cap + (m0 < 32 ? 32 : ((m0 + 3) & ~3))) * esz;
  static DynLdsOnce lds_once[2];
  const int ti = dtype == MIFWT_F64 ? 1 : 0;
  {
    const int max_lds = (dwt1_tail_max_n(dtype) + (dwt1_tail_max_n(dtype) + kTailMaxTaps) / 2 + 8) * esz;
    const void* fn = ti ? reinterpret_cast<const void*>(&dwt1_tail_kernel<double>) : reinterpret_cast<const void*>(&dwt1_tail_kernel<float>);
    if (!lds_once[ti].ensure(fn, max_lds)) return MIFWT_ERR_LAUNCH;
  }
  if (ti)
    hipLaunchKernelGGL((dwt1_tail_kernel<double>), dim3((unsigned)rows), dim3(tail_threads(n0)), lds, stream, a);
  else
    hipLaunchKernelGGL((dwt1_tail_kernel<float>), dim3((unsigned)rows), dim3(tail_threads(n0)), lds, stream, a);
  return hipGetLastError() == hipSuccess ? MIFWT_OK : MIFWT_ERR_LAUNCH;
}

bool idwt1_tail_supported(int dtype, int filt_len, int64_t rows, int64_t m0, int nlevels, const int* out_len) {
  if (g_options[MIFWT_OPT_FORCE_GENERIC] || g_options[MIFWT_OPT_PAIR_MODE] == 2) return false;
  if (dtype != MIFWT_F32 && dtype != MIFWT_F64) return false;
  if (filt_len < 2 || filt_len > kTailMaxTaps || (filt_len & 1)) return false;
  if (rows < 1 || rows > (int64_t(1) << 30) || nlevels < 2 || nlevels > kTailMaxLevels || m0 < 1 || !out_len) return false;
  int64_t m = m0;
  for (int l = 0; l < nlevels; ++l) {
    const int64_t full = 2 * m - filt_len + 2;  // a level's output is this or one less (the reference's end-crop)
    if (out_len[l] < 1 || (out_len[l] != full && out_len[l] != full - 1)) return false;
    m = out_len[l];
    if (m > dwt1_tail_max_n(dtype)) return false;
  }
  return m0 <= dwt1_tail_max_n(dtype);
}

int idwt1_tail(int dtype, int filt_len, int64_t rows, int64_t m0, int nlevels, const void* approx, int64_t approx_row_stride,
               const void* const* details, const int64_t* detail_row_strides, const int* out_len, void* y, int64_t y_row_stride,
               const double* lo, const double* hi, hipStream_t stream) {
  if (!idwt1_tail_supported(dtype, filt_len, rows, m0, nlevels, out_len)) return MIFWT_ERR_UNSUPPORTED;
  Idwt1TailArgs a;
  a.approx = approx;
  a.y = y;
  a.approx_rs = approx_row_stride;
  a.y_rs = y_row_stride;
  int big = (int)m0;
  for (int l = 0; l < nlevels; ++l) {
    a.det[l] = details[l];
    a.det_rs[l] = detail_row_strides[l];
    a.out_len[l] = out_len[l];
    big = out_len[l] > big ? out_len[l] : big;
  }
  a.m0 = (int)m0;
  a.nlevels = nlevels;
  a.filt_len = filt_len;
  for (int t = 0; t < filt_len; ++t) {
    a.lo[t] = lo[t];
    a.hi[t] = hi[t];
  }
  const int esz = dtype == MIFWT_F64 ? 8 : 4;
  // BIG holds the outputs of the levels nlevels - 1, nlevels - 3, ... (and the input of level 0 if nlevels is even), SMALL the
  // others; every row is followed by L/2 zeros
  int need_big = 32, need_small = 32;
  {
    int len = (int)m0;  // row parked for level l
    for (int l = 0; l <= nlevels; ++l) {
      const bool in_small = ((nlevels - 1 - l) & 1) == 0;  // level l reads SMALL when nlevels - 1 - l is even (l = nlevels: the output, never parked)
      if (l < nlevels) {
        int& need = in_small ? need_small : need_big;
        need = std::max(need, len + filt_len / 2 + 4);
        len = out_len[l];
      }
    }
  }
  a.cap = (need_big + 3) & ~3;
  a.cap_small = (need_small + 3) & ~3;
  const size_t lds = (size_t)(a.cap + 2 * a.cap_small) * esz;
  static DynLdsOnce lds_once[2];
  const int ti = dtype == MIFWT_F64 ? 1 : 0;
  {
    const int max_lds = 2 * (dwt1_tail_max_n(dtype) + 64) * esz;
    const void* fn = ti ? reinterpret_cast<const void*>(&idwt1_tail_kernel<double>) : reinterpret_cast<const void*>(&idwt1_tail_kernel<float>);
    if (!lds_once[ti].ensure(fn, max_lds)) return MIFWT_ERR_LAUNCH;
  }
  if (ti)
    hipLaunchKernelGGL((idwt1_tail_kernel<double>), dim3((unsigned)rows), dim3(tail_threads(out_len[nlevels - 1])), lds, stream, a);
  else
    hipLaunchKernelGGL((idwt1_tail_kernel<float>), dim3((unsigned)rows), dim3(tail_threads(out_len[nlevels - 1])), lds, stream, a);
  return hipGetLastError() == hipSuccess ? MIFWT_OK : MIFWT_ERR_LAUNCH;
}

}  // namespace mifwt
