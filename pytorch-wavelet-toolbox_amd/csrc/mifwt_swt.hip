// mifwt_swt.hip — stationary (undecimated, "a trous") wavelet transform levels for gfx950.
//
// Replaces, per level of ptwt.swt / ptwt.iswt (reference src/ptwt/stationary_transform.py:95-107 and :142-156):
//   analysis : _circular_pad + F.conv1d(stride 1, dilation D = 2^level) + split
//   synthesis: stack + _circular_pad + F.conv_transpose1d(groups 2, dilation D, padding) + mean over the pair
// with the circular extension as an index map (no padded tensor, any number of wraps) and both bands in one pass:
//   analysis   lo/hi[n] = s * sum_m h_lo/hi[m] * x[(n + D (L/2 - m)) mod N]
//   synthesis  y[n]     = s * sum_j g_lo[j] a[(n + D (L/2 - 1 - j)) mod N] + g_hi[j] d[(n + D (L/2 - 1 - j)) mod N]
// (s = 1 resp. 1/2 in the transform; the free scale makes each kernel the other's adjoint with reversed taps).
// Bound: HBM (stride 1: N in, 2N out per level).  A lane produces 4 (f32 / f16) or 2 (f64) consecutive samples of one
// row; every tap is one vector load of that run shifted by a multiple of D — the L-fold re-reads are served by the
// vector L1 / L2.  Lanes whose window wraps around the row ends take a per-element modulo path.
// Filter lengths 2..20 are compile-time instantiations (taps in SGPRs, fully unrolled); every other even length up to
// MIFWT_MAX_FILT (db11+, sym11+, coif4+, dmey) runs the same kernel with L = 0: a run-time tap loop over a 128-entry tap table.
#include "mifwt_axis_stream.h"

namespace mifwt {

namespace {

template <typename A, int L>
struct SwtArgs {
  const void* in0;  // analysis: x       synthesis: a
  const void* in1;  // analysis: unused  synthesis: d
  void* out0;       // analysis: lo      synthesis: y
  void* out1;       // analysis: hi      synthesis: unused
  int64_t in0_rs, in1_rs, out0_rs, out1_rs;  // row strides (elements); samples are contiguous
  int rows, n, dilation, nsegs;
  int64_t ntasks;
  int filt_len;  // read by the run-time-length instantiation (L == 0) only
  A scale;
  A lo[L ? L : MIFWT_MAX_FILT], hi[L ? L : MIFWT_MAX_FILT];
};

__device__ __forceinline__ int wrap(int i, int n) {
  i %= n;
  return i < 0 ? i + n : i;
}

template <typename T, int L, bool INVERSE>
__global__ void __launch_bounds__(256) swt_kernel(const SwtArgs<typename ElemTraits<T>::Acc, L> a) {
  using A = typename ElemTraits<T>::Acc;
  constexpr int E = ElemTraits<T>::EO;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t task = (int64_t)blockIdx.x * 4 + wave;
  if (task >= a.ntasks) return;
  const int seg = (int)(task % a.nsegs);
  const int row = (int)(task / a.nsegs);
  const int n0 = (seg * 64 + lane) * E;
  if (n0 >= a.n) return;
  const int D = a.dilation, N = a.n;
  const int FL = L ? L : a.filt_len;
  constexpr int kUnroll = L ? L : 1;
  // tap t reads the run starting at n0 + off(t):  analysis off = D (L/2 - t),  synthesis off = D (L/2 - 1 - t)
  const int off_max = D * (FL / 2 - (INVERSE ? 1 : 0));
  const int off_min = off_max - D * (FL - 1);
  const bool interior = n0 + off_min >= 0 && n0 + off_max + E <= N;
  const T* __restrict__ p0 = static_cast<const T*>(a.in0) + (int64_t)row * a.in0_rs;
  const T* __restrict__ p1 = INVERSE ? static_cast<const T*>(a.in1) + (int64_t)row * a.in1_rs : nullptr;
  A acc0[E], acc1[E];
#pragma unroll
  for (int e = 0; e < E; ++e) acc0[e] = acc1[e] = A(0);
#pragma unroll kUnroll
  for (int t = 0; t < FL; ++t) {
    const int s = n0 + off_max - D * t;
    A v0[E], v1[E];
    if (interior) {
      load_run<T, A, E>(p0 + s, v0);
      if (INVERSE) load_run<T, A, E>(p1 + s, v1);
    } else {
#pragma unroll
      for (int e = 0; e < E; ++e) {
        const int i = wrap(s + e, N);
        v0[e] = (A)p0[i];
        if (INVERSE) v1[e] = (A)p1[i];
      }
    }
#pragma unroll
    for (int e = 0; e < E; ++e) {
      if (INVERSE) {
        acc0[e] = fma(a.lo[t], v0[e], acc0[e]);
        acc0[e] = fma(a.hi[t], v1[e], acc0[e]);
      } else {
        acc0[e] = fma(a.lo[t], v0[e], acc0[e]);
        acc1[e] = fma(a.hi[t], v0[e], acc1[e]);
      }
    }
  }
#pragma unroll
  for (int e = 0; e < E; ++e) {
    acc0[e] *= a.scale;
    acc1[e] *= a.scale;
  }
  T* o0 = static_cast<T*>(a.out0) + (int64_t)row * a.out0_rs + n0;
  T* o1 = INVERSE ? nullptr : static_cast<T*>(a.out1) + (int64_t)row * a.out1_rs + n0;
  if (n0 + E <= N) {
    store_run<T, A, E>(o0, acc0);
    if (!INVERSE) store_run<T, A, E>(o1, acc1);
  } else {
#pragma unroll
    for (int e = 0; e < E; ++e)
      if (n0 + e < N) {
        o0[e] = (T)acc0[e];
        if (!INVERSE) o1[e] = (T)acc1[e];
      }
  }
}

struct SwtCall {
  int inverse, filt_len;
  int64_t rows, n, dilation;
  const void* in0;
  const void* in1;
  void* out0;
  void* out1;
  int64_t in0_rs, in1_rs, out0_rs, out1_rs;
  const double* lo;
  const double* hi;
  double scale;
  hipStream_t stream;
};

template <typename T, int L>
int swt_launch(const SwtCall& c) {
  using A = typename ElemTraits<T>::Acc;
  SwtArgs<A, L> a;
  a.in0 = c.in0;
  a.in1 = c.in1;
  a.out0 = c.out0;
  a.out1 = c.out1;
  a.in0_rs = c.in0_rs;
  a.in1_rs = c.in1_rs;
  a.out0_rs = c.out0_rs;
  a.out1_rs = c.out1_rs;
  a.rows = (int)c.rows;
  a.n = (int)c.n;
  a.dilation = (int)c.dilation;
  const int64_t per_seg = 64 * ElemTraits<T>::EO;
  const int64_t nsegs = (c.n + per_seg - 1) / per_seg;
  a.nsegs = (int)nsegs;
  a.ntasks = nsegs * c.rows;
  a.scale = (A)c.scale;
  a.filt_len = c.filt_len;
  for (int t = 0; t < c.filt_len; ++t) {
    a.lo[t] = (A)c.lo[t];
    a.hi[t] = (A)c.hi[t];
  }
  if (a.ntasks == 0) return MIFWT_OK;
  const int64_t nblk = (a.ntasks + 3) / 4;
  if (nblk > INT32_MAX) return MIFWT_ERR_UNSUPPORTED;
  if (c.inverse)
    hipLaunchKernelGGL((swt_kernel<T, L, true>), dim3((unsigned)nblk), dim3(256), 0, c.stream, a);
  else
    hipLaunchKernelGGL((swt_kernel<T, L, false>), dim3((unsigned)nblk), dim3(256), 0, c.stream, a);
  return hipGetLastError() == hipSuccess ? MIFWT_OK : MIFWT_ERR_LAUNCH;
}

template <typename T>
int swt_dispatch(const SwtCall& c) {
  switch (c.filt_len) {
    case 2: return swt_launch<T, 2>(c);
    case 4: return swt_launch<T, 4>(c);
    case 6: return swt_launch<T, 6>(c);
    case 8: return swt_launch<T, 8>(c);
    case 10: return swt_launch<T, 10>(c);
    case 12: return swt_launch<T, 12>(c);
    case 14: return swt_launch<T, 14>(c);
    case 16: return swt_launch<T, 16>(c);
    case 18: return swt_launch<T, 18>(c);
    case 20: return swt_launch<T, 20>(c);
    default: return swt_launch<T, 0>(c);  // run-time tap loop
  }
}

int swt_level(int inverse, int dtype, int filt_len, int64_t rows, int64_t n, int64_t dilation, const void* in0,
              const void* in1, int64_t in0_rs, int64_t in1_rs, void* out0, void* out1, int64_t out0_rs, int64_t out1_rs,
              const double* lo, const double* hi, double scale, void* stream) {
  if (!in0 || !out0 || !lo || !hi || (inverse && !in1) || (!inverse && !out1)) return MIFWT_ERR_BADARG;
  if (filt_len < 2 || (filt_len & 1) || filt_len > MIFWT_MAX_FILT || rows < 0 || n < 1 || dilation < 1) return MIFWT_ERR_BADARG;
  if (n > INT32_MAX / 8 || rows > INT32_MAX / 8 || dilation * filt_len > INT32_MAX / 8) return MIFWT_ERR_UNSUPPORTED;
  SwtCall c = {inverse, filt_len, rows, n, dilation, in0, in1, out0, out1, in0_rs, in1_rs, out0_rs, out1_rs, lo, hi, scale,
               static_cast<hipStream_t>(stream)};
  switch (dtype) {
    case MIFWT_F32: return swt_dispatch<float>(c);
    case MIFWT_F64: return swt_dispatch<double>(c);
    case MIFWT_F16: return swt_dispatch<_Float16>(c);
    default: return MIFWT_ERR_BADARG;
  }
}

}  // namespace

}  // namespace mifwt

extern "C" {

int mifwt_swt_fwd(int dtype, int filt_len, int64_t rows, int64_t n, int64_t dilation, const void* x, int64_t x_row_stride,
                  void* lo, void* hi, int64_t lo_row_stride, int64_t hi_row_stride, const double* dec_lo,
                  const double* dec_hi, double scale, void* stream) {
  return mifwt::swt_level(0, dtype, filt_len, rows, n, dilation, x, nullptr, x_row_stride, 0, lo, hi, lo_row_stride,
                          hi_row_stride, dec_lo, dec_hi, scale, stream);
}

int mifwt_swt_inv(int dtype, int filt_len, int64_t rows, int64_t n, int64_t dilation, const void* a, const void* d,
                  int64_t a_row_stride, int64_t d_row_stride, void* y, int64_t y_row_stride, const double* rec_lo,
                  const double* rec_hi, double scale, void* stream) {
  return mifwt::swt_level(1, dtype, filt_len, rows, n, dilation, a, d, a_row_stride, d_row_stride, y, nullptr, y_row_stride,
                          0, rec_lo, rec_hi, scale, stream);
}

}  // extern "C"
