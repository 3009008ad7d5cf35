// mifwt_common.h — internal declarations shared by the gfx950 kernels and the C-ABI dispatcher.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

#include "mifwt.h"

namespace mifwt {

constexpr int kMaxFilt = MIFWT_MAX_FILT;
extern int g_options[16];  // mifwt_set_option() switches

// DIAGNOSTICS.  The measurement switches that break results (no stores / no loads / no deep levels ..., MIFWT_OPT_DEBUG), the experiment
// word of A/B runs (MIFWT_OPT_EXP) and the per-wave profiling instances exist only in builds with -DMIFWT_DIAG (tools/: experiment builds
// through __graft_entry__.build_variant("diag", "-DMIFWT_DIAG"), selected with MIFWT_LIB).  In the product build the kernels never read
// those words — MIFWT_DBG / MIFWT_EXPW are the constant 0, MIFWT_PROFP a null pointer — and mifwt_set_option refuses them; what
// MIFWT_OPT_DEBUG keeps in the product are the ROUTING bits (kRouteBits: alternative code paths with the same results, pinned by parity tests).
#ifdef MIFWT_DIAG
constexpr bool kDiag = true;
#define MIFWT_DBG(a) ((a).dbg)
#define MIFWT_EXPW(a) ((a).exp)
#define MIFWT_PROFP(a) ((a).prof)
#else
constexpr bool kDiag = false;
#define MIFWT_DBG(a) 0
#define MIFWT_EXPW(a) 0
#define MIFWT_PROFP(a) (static_cast<unsigned long long*>(nullptr))
#endif
// 8 = kernel 22 without its fast warm-up, 64 = column strips of 64 in the 3-D analysis walk, 512 = 8-byte stores in the 3-D synthesis walk,
// 1024 = analysis adjoints on the generic passes, 4096 = the per-sample border kernel, 8192 = level-2 waves of kernel 16 behind the step's
// second barrier, bits 19 / 20 = kernel 16 without its tail wave / in its sixteen-wave form
constexpr int kRouteBits = 8 | 64 | 512 | 1024 | 4096 | 8192 | 524288 | 1048576 | 2097152;
inline int exp_word() { return kDiag ? g_options[MIFWT_OPT_EXP] : 0; }
extern unsigned long long g_launch_counts[16];  // mifwt_launch_count(): launches per kernel variant (MIFWT_VARIANT_*)
inline void count_launch(int variant) { __atomic_fetch_add(&g_launch_counts[variant], 1ull, __ATOMIC_RELAXED); }

// Two-level batch for ONE call of the LDS-tile 2-D analysis kernel (thread-local, set and cleared by the 3-D composed route around that
// call): the input images are `inner` slices per volume, `outer_stride` elements between volumes; inner == 0: off.
struct BatchSplit {
  int64_t inner, outer_stride;
};
extern thread_local BatchSplit g_batch_split;

// hipFuncAttributeMaxDynamicSharedMemorySize is a property of a kernel ON A DEVICE: set it once per (kernel, device).  One of these
// as a function-local static next to each launch; safe from threads that call with the GIL released (two racing threads at worst
// both set the same value).
struct DynLdsOnce {
  std::atomic<uint64_t> done{0};
  bool ensure(const void* fn, int bytes) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    const uint64_t bit = uint64_t(1) << (dev & 63);
    if (done.load(std::memory_order_acquire) & bit) return true;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return false;
    done.fetch_or(bit, std::memory_order_release);
    return true;
  }
};

// Boundary extension as an index map (replaces F.pad / _pad_symmetric of the reference:
// src/ptwt/conv_transform.py:59-66, src/ptwt/_util.py:163-195).  Returns the source index in [0, n) of
// extended-signal index i, or -1 for an implicit zero.  reflect = whole-sample mirror, symmetric =
// half-sample mirror; both fold repeatedly, so any |i| is legal (the host layer reproduces torch's
// refusal of reflect pad >= n / circular pad > n before a kernel is ever launched).
__host__ __device__ __forceinline__ int ext_index(int i, int n, int mode) {
  if ((unsigned)i < (unsigned)n) return i;
  switch (mode) {
    case MIFWT_MODE_ZERO:
      return -1;
    case MIFWT_MODE_CONSTANT:
      return i < 0 ? 0 : n - 1;
    case MIFWT_MODE_PERIODIC: {
      int p = i % n;
      return p < 0 ? p + n : p;
    }
    case MIFWT_MODE_SYMMETRIC: {
      const int t = 2 * n;
      int p = i % t;
      if (p < 0) p += t;
      return p < n ? p : t - 1 - p;
    }
    default: {  // MIFWT_MODE_REFLECT
      if (n == 1) return 0;
      const int t = 2 * (n - 1);
      int p = i % t;
      if (p < 0) p += t;
      return p < n ? p : t - p;
    }
  }
}

// Same map, arranged so that the common case — an index at most one period outside [0, n) — needs no integer
// division (a division costs ~25 scalar / vector instructions on gfx950); falls back to ext_index otherwise.
__host__ __device__ __forceinline__ int ext_index_near(int i, int n, int mode) {
  if ((unsigned)i < (unsigned)n) return i;
  int j;
  switch (mode) {
    case MIFWT_MODE_ZERO:
      return -1;
    case MIFWT_MODE_CONSTANT:
      return i < 0 ? 0 : n - 1;
    case MIFWT_MODE_PERIODIC:
      j = i < 0 ? i + n : i - n;
      break;
    case MIFWT_MODE_SYMMETRIC:
      j = i < 0 ? -i - 1 : 2 * n - 1 - i;
      break;
    default:  // MIFWT_MODE_REFLECT
      j = i < 0 ? -i : 2 * (n - 1) - i;
      break;
  }
  return (unsigned)j < (unsigned)n ? j : ext_index(i, n, mode);
}

// ---- generic per-axis passes (any L <= 128, any strides, f32/f64) ---------------------------------
struct AxisJob {
  const void* in0;  // analysis: input          synthesis: low-pass band
  const void* in1;  // analysis: unused         synthesis: high-pass band
  void* out0;       // analysis: low-pass band  synthesis: output
  void* out1;       // analysis: high-pass band synthesis: unused
  int64_t in0_stride[4], in1_stride[4], out0_stride[4], out1_stride[4];
};

// DEVICE-RESIDENT TAPS (round 5): while `g_dtaps.lo` is set (mifwt_*_dtaps entry points, this thread only) the generic axis kernels
// read their filter from these device arrays of L doubles instead of their launch arguments — a learnable filter bank that lives on
// the GPU then needs no device-to-host copy, no stream synchronisation, and the calls can be captured into a HIP graph.  `rev`:
// tap m of the pass is element L - 1 - m of the arrays (the adjoint of a synthesis level is an analysis level with reversed taps).
struct DeviceTaps {
  const double* lo;
  const double* hi;
  int rev;
};
extern thread_local DeviceTaps g_dtaps;
// ... and (round 6) the FUSED kernels that serve the learnable-wavelet training loop: their launch code copies this thread's
// `g_dtaps` into the kernel arguments (DevTapArg; lo == nullptr: the taps travel by value as before) and the kernel reads the L doubles
// once, when it starts, into the registers its by-value taps would occupy — same conversions, same packing, bit-identical results.
struct DevTapArg {
  const double* lo;
  const double* hi;
  int rev, len;
};
extern thread_local int g_dtaps_taken;  // launches that copied g_dtaps into their arguments (checked by the mifwt_*_dtaps entry points)
inline DevTapArg dev_tap_arg(int filt_len) {
  if (g_dtaps.lo) ++g_dtaps_taken;
  return {g_dtaps.lo, g_dtaps.hi, g_dtaps.rev, filt_len};
}
#ifdef __HIPCC__
// wave-uniform value in a scalar register (the fused kernels keep their taps in SGPR pairs)
// (v_readfirstlane through inline assembly: the compiler knows the converted tap is uniform, drops the builtin as redundant and keeps
// the value — every tap of every kernel, both branches merged — in a VECTOR register: 79 -> 134 registers in the streaming axis kernels.
// The wait states around it are the ones the compiler would insert for a readfirstlane of its own and cannot for an opaque asm
// statement on gfx950: one between a VALU write of the source register and the read — without it the low word of a double tap came
// out stale, 1e-7 off — and two between the SGPR write and a VALU that reads it.  Kernel entry only.)
__device__ __forceinline__ int dtap_rfl(int v) {
  int r;
  asm volatile("s_nop 0\n\tv_readfirstlane_b32 %0, %1\n\ts_nop 1" : "=s"(r) : "v"(v));
  return r;
}
__device__ __forceinline__ float dtap_uniform(float v) { return __builtin_bit_cast(float, dtap_rfl(__builtin_bit_cast(int, v))); }
// (a double tap is loaded by s_load and never converted: it is in scalar registers already)
__device__ __forceinline__ double dtap_uniform(double v) { return v; }
// tap m of the low-pass / high-pass filter of the pass, in the kernel's arithmetic type S (the conversion the host does for by-value taps)
template <typename S>
__device__ __forceinline__ S dtap_lo(const DevTapArg& dt, int m) { return dtap_uniform((S)dt.lo[dt.rev ? dt.len - 1 - m : m]); }
template <typename S>
__device__ __forceinline__ S dtap_hi(const DevTapArg& dt, int m) { return dtap_uniform((S)dt.hi[dt.rev ? dt.len - 1 - m : m]); }
#endif
int launch_axis_fwd(int dtype, const AxisJob* jobs, int njobs, const int64_t out_ext[4], int taxis, int64_t n_in,
                    int mode, int filt_len, const double* lo, const double* hi, hipStream_t stream);
int launch_axis_inv(int dtype, const AxisJob* jobs, int njobs, const int64_t out_ext[4], int taxis, int64_t m_in,
                    int filt_len, const double* lo, const double* hi, hipStream_t stream);

// adjoint of one analysis axis pass (g_lo, g_hi -> g_x, halo folded back through the boundary map); lo / hi are
// the DEC taps in PyWavelets order, m_in the coefficient extent, n_sig the signal extent along taxis
int launch_axis_adj(int dtype, const AxisJob* jobs, int njobs, const int64_t out_ext[4], int taxis, int64_t m_in,
                    int64_t n_sig, int mode, int filt_len, const double* lo, const double* hi, hipStream_t stream);

// ---- streaming single-axis kernels (mifwt_axis_stream.h; any mode, L in the instantiated set) ----------
// One (input -> low, high) or (low, high -> output) job; up to four jobs of identical geometry per launch.
// Strides are in elements.  outer kernels: s[0] = batch, s[1] = transformed axis (the inner run is dense);
// inner kernels: s[0..2] = the three row dims (the transformed axis has stride 1).
struct StreamJob {
  const void* in0;  // analysis: x            synthesis: low-pass band
  const void* in1;  // analysis: unused       synthesis: high-pass band
  void* out0;       // analysis: low-pass     synthesis: y
  void* out1;       // analysis: high-pass    synthesis: unused
  int64_t in0_s[3], in1_s[3], out0_s[3], out1_s[3];
};

struct StreamCall {
  int filt_len, mode;
  const StreamJob* jobs;
  int njobs;
  int64_t batch;       // outer
  int64_t rows[3];     // inner
  int64_t n_in, n_out;
  int64_t inner;       // outer
  const double* lo;
  const double* hi;
  hipStream_t stream;
};

enum StreamKind { kOuterFwd = 0, kOuterInv = 1, kInnerFwd = 2, kInnerInv = 3 };
bool stream_filter_supported(int filt_len);
int stream_call(int dtype, int kind, const StreamCall& c);  // dispatches on dtype (f32 / f64 / f16) and filt_len

// ---- fused fast paths --------------------------------------------------------------------------------
// Each returns MIFWT_ERR_UNSUPPORTED when the descriptor is outside its envelope (the dispatcher then
// falls back to the generic passes) and never touches the workspace.
enum KernelId { kGeneric = 0, kDwt2FwdStream = 1, kDwt2InvStream = 2, kDwt1FwdRow = 3, kDwt1InvRow = 4, kDwt3FwdStream = 5, kDwt3InvStream = 6, kDwt2FwdTile = 7, kDwt2InvTile = 8, kDwt3FwdTile = 9, kDwt3InvTile = 10, kDwt2FwdMfma = 11, kDwt2FwdPair = 12, kDwt2InvPair = 13, kDwt1FwdTail = 14, kDwt1InvTail = 15, kDwt2FwdPyr = 16,
  kDwt1FwdLong = 17, kDwt1InvLong = 18, kDwt2FwdSmall = 20, kDwt2InvSmall = 21, kDwt2InvPyr = 22, kDwt2InvMfma = 23, kDwt3FwdWalk = 24, kDwt3InvWalk = 25 };

bool dwt2_fwd_stream_supported(const mifwt_level_desc* d);
int dwt2_fwd_stream(const mifwt_level_desc* d, const void* x, void* approx, void* const* details,
                    const double* dec_lo, const double* dec_hi, hipStream_t stream);

// LDS-tile fused 2-D analysis level (mifwt_dwt2_tile.h): f32 / f16 storage, even L <= 16 and L in {18, 20, 24, 32}
bool dwt2_fwd_tile_supported(const mifwt_level_desc* d);
int dwt2_fwd_tile(const mifwt_level_desc* d, const void* x, void* approx, void* const* details,
                  const double* dec_lo, const double* dec_hi, hipStream_t stream);

// matrix-core (banded-Toeplitz MFMA) fused 2-D analysis level: f16 storage, even L in [18, 32]
bool dwt2_fwd_mfma_supported(const mifwt_level_desc* d);
// ... and its synthesis mirror (mifwt_dwt2_inv_mfma.hip, kernel id 23)
bool dwt2_inv_mfma_supported(const mifwt_level_desc* d);
int dwt2_inv_mfma(const mifwt_level_desc* d, const void* approx, const void* const* details, void* y, const double* lo,
                  const double* hi, hipStream_t stream);
int dwt2_fwd_mfma(const mifwt_level_desc* d, const void* x, void* approx, void* const* details,
                  const double* dec_lo, const double* dec_hi, hipStream_t stream);

// two consecutive 2-D analysis levels in one launch, the intermediate approximation kept in LDS
// (mifwt_dwt2_fwd_pair.hip): f32, even L <= 8, every mode but periodic
bool dwt2_fwd_pair_supported(const mifwt_level_desc* d1, const mifwt_level_desc* d2);
int dwt2_fwd_pair(const mifwt_level_desc* d1, const mifwt_level_desc* d2, const void* x, void* const* details1,
                  void* approx2, void* const* details2, const double* dec_lo, const double* dec_hi, hipStream_t stream);

// the same two levels as rolling column strips (mifwt_dwt2_fwd_roll.hip): needs a level-1 plane of >= 32 rows
bool dwt2_fwd_roll_supported(const mifwt_level_desc* d1, const mifwt_level_desc* d2);

// boundary part of the adjoint of an analysis level with a boundary extension (mifwt_adjoint_border.hip): recomputes the samples near
// the borders of g_x, pad positions folded back, after the zero-mode adjoint (a synthesis launch) has written all of g_x
bool adjoint_border_supported(const mifwt_level_desc* d);
int adjoint_border(const mifwt_level_desc* d, const void* g_approx, const void* const* g_details, void* g_x, const double* dec_lo,
                   const double* dec_hi, hipStream_t stream);
int dwt2_fwd_roll(const mifwt_level_desc* d1, const mifwt_level_desc* d2, const void* x, void* const* details1,
                  void* approx2, void* const* details2, const double* dec_lo, const double* dec_hi, hipStream_t stream);

// up to three consecutive 2-D analysis levels in one launch: cooperative column groups + loader waves, rolling vertical
// passes (mifwt_dwt2_fwd_pyr.hip): f32, even L <= 8, every mode but periodic, 16-byte aligned input rows; d[l] = level l + 1,
// every level of a 2-D decomposition of a small plane in one launch, a workgroup per image (mifwt_dwt2_fwd_small.hip)
bool dwt2_fwd_small_supported(int nlevels, const mifwt_level_desc* const* d);
int dwt2_fwd_small(int nlevels, const mifwt_level_desc* const* d, const void* x, void* const* const* details, void* approx, const double* lo,
                   const double* hi, hipStream_t stream);
// every level of a 2-D reconstruction of a small plane in one launch (mifwt_dwt2_inv_small.hip); d[0] = the coarsest level
bool dwt2_inv_small_supported(int nlevels, const mifwt_level_desc* const* d);
int dwt2_inv_small(int nlevels, const mifwt_level_desc* const* d, const void* approx, const void* const* const* details, void* y,
                   const double* lo, const double* hi, hipStream_t stream);
// details[l] = its three detail planes, approx = the last level's approximation
// up to three consecutive 2-D synthesis levels of a big plane in one launch, rows streamed through registers and LDS rings
// (mifwt_dwt2_inv_pyr.hip): f32, even L <= 8; d[0] = the coarsest level
bool dwt2_inv_pyr_supported(int nlev, const mifwt_level_desc* const* d);
int dwt2_inv_pyr(int nlev, const mifwt_level_desc* const* d, const void* approx, const void* const* const* details, void* y,
                 const double* rec_lo, const double* rec_hi, hipStream_t stream);
bool dwt2_fwd_pyr_supported(int nlev, const mifwt_level_desc* const* d);
int dwt2_fwd_pyr(int nlev, const mifwt_level_desc* const* d, const void* x, void* const* const* details, void* approx,
                  const double* dec_lo, const double* dec_hi, hipStream_t stream);
int dwt2_fwd_pyr_schedule(int nlev, const mifwt_level_desc* const* d, uint32_t* wg_start, int capacity);  // the row chunks of the launch

// two consecutive 2-D synthesis levels in one launch (mifwt_idwt2_pair.hip): f32, even L <= 8; d2 = the coarser level
bool dwt2_inv_pair_supported(const mifwt_level_desc* d2, const mifwt_level_desc* d1);
int dwt2_inv_pair(const mifwt_level_desc* d2, const mifwt_level_desc* d1, const void* approx2, const void* const* details2,
                  const void* const* details1, void* y, const double* rec_lo, const double* rec_hi, hipStream_t stream);

// the deep levels of a 1-D decomposition in one launch, one workgroup per row (mifwt_dwt1_tail.hip)
int dwt1_tail_max_n(int dtype);
bool dwt1_tail_supported(int dtype, int filt_len, int mode, int64_t rows, int64_t n0, int nlevels);
int dwt1_tail(int dtype, int filt_len, int mode, int64_t rows, int64_t n0, int nlevels, const void* x, int64_t x_row_stride,
              void* approx, int64_t approx_row_stride, void* const* details, const int64_t* detail_row_strides, const double* lo,
              const double* hi, hipStream_t stream);

// several levels of a 1-D decomposition of rows too long for one workgroup, chunk per workgroup (mifwt_dwt1_long.hip)
int dwt1_long_levels(int dtype, int filt_len, int mode, int64_t rows, int64_t n0, int want);
int dwt1_long(int dtype, int filt_len, int mode, int64_t rows, int64_t n0, int nlevels, const void* x, int64_t x_row_stride,
              void* approx, int64_t approx_row_stride, void* const* details, const int64_t* detail_row_strides, const double* lo,
              const double* hi, hipStream_t stream);
// ... and the coarse levels of a 1-D reconstruction (same file); out_len[l] = output samples of fused level l
// the finest levels of a 1-D reconstruction in one launch, a chunk of the output row per workgroup (mifwt_dwt1_long.hip);
// m[s] = coefficients per row entering fused step s (coarsest first), m[nlevels] = output length
int idwt1_long_supported(int dtype, int filt_len, int64_t rows, int nlevels, const int* m);
int dwt1_long_plan_query(int dtype, int filt_len, int mode, int64_t rows, int64_t n0, int want, int* out);
int idwt1_long_plan_query(int dtype, int filt_len, int64_t rows, int nlevels, const int* m, int* out);
int idwt1_long(int dtype, int filt_len, int64_t rows, int nlevels, const int* m, const void* approx, int64_t approx_row_stride,
               const void* const* details, const int64_t* detail_row_strides, void* y, int64_t y_row_stride, const double* lo,
               const double* hi, hipStream_t stream);
bool idwt1_tail_supported(int dtype, int filt_len, int64_t rows, int64_t m0, int nlevels, const int* out_len);
int idwt1_tail(int dtype, int filt_len, int64_t rows, int64_t m0, int nlevels, const void* approx, int64_t approx_row_stride,
               const void* const* details, const int64_t* detail_row_strides, const int* out_len, void* y, int64_t y_row_stride,
               const double* lo, const double* hi, hipStream_t stream);

// which fused 2-D analysis kernel serves this descriptor: kDwt2FwdTile, kDwt2FwdStream, or -1 (neither)
int dwt2_fwd_choice(const mifwt_level_desc* d);
int dwt2_fwd_fused(const mifwt_level_desc* d, const void* x, void* approx, void* const* details,
                   const double* dec_lo, const double* dec_hi, hipStream_t stream);  // runs that choice

// LDS-tile fused 2-D synthesis level (mifwt_idwt2_tile.h): f32 / f16, even L <= 16 and L in {18, 20, 24, 32}
bool dwt2_inv_tile_supported(const mifwt_level_desc* d);
int dwt2_inv_tile(const mifwt_level_desc* d, const void* approx, const void* const* details, void* y,
                  const double* rec_lo, const double* rec_hi, hipStream_t stream);
int dwt2_inv_choice(const mifwt_level_desc* d);  // kDwt2InvTile, kDwt2InvStream, or -1
int dwt2_inv_fused(const mifwt_level_desc* d, const void* approx, const void* const* details, void* y,
                   const double* rec_lo, const double* rec_hi, hipStream_t stream);

bool dwt2_inv_stream_supported(const mifwt_level_desc* d);
int dwt2_inv_stream(const mifwt_level_desc* d, const void* approx, const void* const* details, void* y,
                    const double* rec_lo, const double* rec_hi, hipStream_t stream);

// fully fused LDS-tile 3-D analysis level (mifwt_dwt3_fwd_tile.hip): f32, L in {2, 4, 6}
bool dwt3_fwd_tile_supported(const mifwt_level_desc* d);
int dwt3_fwd_tile(const mifwt_level_desc* d, const void* x, void* approx, void* const* details,
                  const double* dec_lo, const double* dec_hi, hipStream_t stream);
// fully fused 3-D analysis level, workgroups walking along the depth axis (mifwt_dwt3_fwd_walk.hip): f32, even L <= 10, every mode
bool dwt3_fwd_walk_supported(const mifwt_level_desc* d);
// ... its SLAB form for 8 / 10 taps on rows of at most 128 samples (mifwt_dwt3_fwd_slab.hip): f32, every mode; dwt3_fwd_walk runs it
// where it applies (MIFWT_OPT_DEBUG bit 21 keeps the strip form)
bool dwt3_fwd_slab_supported(const mifwt_level_desc* d);
int dwt3_fwd_slab_plan_query(const mifwt_level_desc* d, int* out, int capacity);  // (mifwt_dwt3_fwd_slab_plan)
bool dwt3_fwd_slab_pays(const mifwt_level_desc* d);  // ... and is ahead of the composed route (a cost model fitted to measurements)
int dwt3_fwd_slab(const mifwt_level_desc* d, const void* x, void* approx, void* const* details, const double* lo, const double* hi,
                  hipStream_t stream);
int dwt3_fwd_walk(const mifwt_level_desc* d, const void* x, void* approx, void* const* details, const double* lo, const double* hi,
                  hipStream_t stream);
// ... and its synthesis mirror (mifwt_dwt3_inv_walk.hip): f32, even L <= 8, dense coefficient rows
bool dwt3_inv_walk_supported(const mifwt_level_desc* d);
int dwt3_inv_walk(const mifwt_level_desc* d, const void* approx, const void* const* details, void* y, const double* lo, const double* hi,
                  hipStream_t stream);
// fully fused LDS-brick 3-D synthesis level (mifwt_dwt3_inv_tile.hip): f32, L in {2, 4, 6}
bool dwt3_inv_tile_supported(const mifwt_level_desc* d);
int dwt3_inv_tile(const mifwt_level_desc* d, const void* approx, const void* const* details, void* y, const double* lo,
                  const double* hi, hipStream_t stream);

// ---- composed routes (mifwt_compose.hip) --------------------------------------------------------------------
bool plane3_route_ok(const mifwt_level_desc* d, int direction);  // ndim 3 f32: fused 2-D planes + depth pass
bool rows_route_ok(const mifwt_level_desc* d, int direction);    // streaming inner pass + outer passes
size_t plane3_ws_bytes(const mifwt_level_desc* d, int direction);
size_t rows_ws_bytes(const mifwt_level_desc* d, int direction);
int plane3_fwd(const mifwt_level_desc* d, const void* x, void* approx, void* const* details, const double* lo,
               const double* hi, void* ws, hipStream_t stream);
int plane3_inv(const mifwt_level_desc* d, const void* approx, const void* const* details, void* y, const double* lo,
               const double* hi, void* ws, hipStream_t stream);
int rows_fwd(const mifwt_level_desc* d, const void* x, void* approx, void* const* details, const double* lo,
             const double* hi, void* ws, hipStream_t stream);
int rows_inv(const mifwt_level_desc* d, const void* approx, const void* const* details, void* y, const double* lo,
             const double* hi, void* ws, hipStream_t stream);

}  // namespace mifwt
