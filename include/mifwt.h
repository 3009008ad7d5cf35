/*
 * mifwt.h — C ABI of libmifwt.so, the MI355X (gfx950) fast-wavelet-transform engine.
 *
 * Drop-in seam.  The reference (v0lta/PyTorch-Wavelet-Toolbox, "ptwt") is pure Python and has no FFI;
 * this ABI sits exactly where the reference hands one decomposition / reconstruction level to ATen:
 *
 *   analysis  level : F.pad (or _pad_symmetric) + F.conv{1,2,3}d(stride=2) + torch.split
 *       src/ptwt/conv_transform.py:135-139     (1-D:  _fwt_pad :33-66,  conv1d :137)
 *       src/ptwt/conv_transform_2.py:142-149   (2-D:  _fwt_pad2 :34-71, conv2d :144)
 *       src/ptwt/conv_transform_3.py:121-141   (3-D:  _fwt_pad3 :34-73, conv3d :127)
 *       src/ptwt/separable_conv_transform.py:38-72 (separable: the same level, axis by axis)
 *   synthesis level : torch.stack + F.conv_transpose{1,2,3}d(stride=2) + crop
 *       src/ptwt/conv_transform.py:184-199, conv_transform_2.py:222-249, conv_transform_3.py:205-249,
 *       src/ptwt/separable_conv_transform.py:75-111
 *
 * One call = one level for the whole (folded) batch = ONE fused kernel on the fast paths: the boundary
 * extension is an index map inside the kernel (no padded tensor in HBM), the 2^ndim sub-bands are written
 * straight to their final planes (no split/stack copies) and only the cropped interior of a synthesis
 * level is computed.
 *
 * Conventions
 *   - All data pointers are DEVICE pointers owned by the caller (torch's caching allocator); filter taps
 *     are HOST pointers (double, PyWavelets order, NOT flipped — the library applies the flip the
 *     reference does at src/ptwt/_util.py:863-865).  Taps are copied into the kernel arguments, so there
 *     is no device-side filter state and calls are re-entrant.
 *   - Nothing here allocates, frees or synchronises.  Work is enqueued on `stream`
 *     (a hipStream_t, e.g. torch.cuda.current_stream().cuda_stream) of the current HIP device.
 *   - Strides are in ELEMENTS.  Index 0 of every stride array is the folded batch, then the transformed
 *     axes outermost first.  The innermost transformed axis need not be contiguous (arbitrary strides are
 *     accepted; unit innermost stride selects the fused fast kernels).
 *   - Sub-band order: band index s in [0, 2^ndim); bit (ndim-1-a) of s set <=> axis a is HIGH-pass
 *     ("aa","ad","da","dd" / "aaa","aad",...,"ddd": character i of the key <-> transformed axis i,
 *     as in src/ptwt/_util.py:926-934).  Band 0 is the approximation; details[s-1] is band s.
 *     (ptwt's 2-D tuple is (H,V,D) = ('da','ad','dd') = details[1], details[0], details[2].)
 *   - Return value: MIFWT_OK or a negative error code; never throws.
 */
#ifndef MIFWT_H
#define MIFWT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MIFWT_ABI_VERSION 3
#define MIFWT_MAX_NDIM 3
#define MIFWT_MAX_FILT 128 /* longest PyWavelets discrete filter is coif17 = 102 taps */

enum mifwt_dtype { MIFWT_F32 = 0, MIFWT_F64 = 1, MIFWT_F16 = 2 /* f16 storage, f32 arithmetic */ };

/* Boundary rules of ptwt.constants.BoundaryMode (src/ptwt/constants.py:85-108, _util.py:36-44). */
enum mifwt_mode {
  MIFWT_MODE_ZERO = 0,      /* ... 0  0 | x1 x2 ... xn | 0  0 ...        (torch "constant")  */
  MIFWT_MODE_CONSTANT = 1,  /* ... x1 x1 | x1 x2 ... xn | xn xn ...      (torch "replicate") */
  MIFWT_MODE_REFLECT = 2,   /* ... x3 x2 | x1 x2 ... xn | xn-1 xn-2 ...  (torch "reflect")   */
  MIFWT_MODE_PERIODIC = 3,  /* ... xn-1 xn | x1 x2 ... xn | x1 x2 ...    (torch "circular")  */
  MIFWT_MODE_SYMMETRIC = 4  /* ... x2 x1 | x1 x2 ... xn | xn xn-1 ...    (_pad_symmetric)    */
};

enum mifwt_status {
  MIFWT_OK = 0,
  MIFWT_ERR_BADARG = -1,      /* null pointer, ndim/dtype/mode/filt_len out of range, extents inconsistent */
  MIFWT_ERR_UNSUPPORTED = -2, /* valid request this build has no kernel for */
  MIFWT_ERR_WORKSPACE = -3,   /* workspace smaller than mifwt_workspace_bytes() */
  MIFWT_ERR_LAUNCH = -4       /* hipLaunchKernel reported an error (see hipGetLastError) */
};

/* One decomposition / reconstruction level of an ndim-dimensional transform over a folded batch. */
typedef struct mifwt_level_desc {
  int32_t ndim;     /* 1..3 transformed axes */
  int32_t dtype;    /* enum mifwt_dtype */
  int32_t mode;     /* enum mifwt_mode (analysis only; synthesis ignores it) */
  int32_t filt_len; /* L, 2..MIFWT_MAX_FILT */
  int64_t batch;    /* folded leading dims (src/ptwt/_util.py:271-286) */
  /* signal side: analysis input / synthesis output.  Synthesis output extent per axis must be
   * 2*M - L + 2 - t with t in {0,1}: t = 1 is the reference's extra end-crop
   * (src/ptwt/_util.py:231-244); only these samples are computed. */
  int64_t sig_extent[MIFWT_MAX_NDIM];
  int64_t sig_stride[1 + MIFWT_MAX_NDIM];
  /* coefficient side: every sub-band has extent M = floor((N + L - 1) / 2) per axis
   * (= the conv output length of src/ptwt/_util.py:204-217; passed, not recomputed). */
  int64_t coef_extent[MIFWT_MAX_NDIM];
  int64_t approx_stride[1 + MIFWT_MAX_NDIM]; /* band 0 */
  int64_t detail_stride[1 + MIFWT_MAX_NDIM]; /* bands 1..2^ndim-1 share one stride set */
} mifwt_level_desc;

/* Replaces: F.pad/_pad_symmetric + F.conv{1,2,3}d(stride=2) + split (citations above).
 *   x        analysis input                        [batch, N_0.., N_{ndim-1}] via sig_stride
 *   approx   band 0 output                         [batch, M_0..]             via approx_stride
 *   details  HOST array of 2^ndim-1 device ptrs    [batch, M_0..] each        via detail_stride
 *   dec_lo/dec_hi  HOST taps, PyWavelets order (wavelet.dec_lo / dec_hi), L doubles each
 *   workspace      device scratch of >= mifwt_workspace_bytes(desc, 0) bytes (may be NULL if that is 0) */
int mifwt_dwt_fwd(const mifwt_level_desc* desc, const void* x, void* approx, void* const* details,
                  const double* dec_lo, const double* dec_hi, void* workspace, size_t workspace_bytes,
                  void* stream);

/* Replaces: torch.stack + F.conv_transpose{1,2,3}d(stride=2) + crop (citations above).
 *   approx / details  inputs laid out as above;  y  output [batch, sig_extent..] via sig_stride
 *   rec_lo/rec_hi     HOST taps, PyWavelets order (wavelet.rec_lo / rec_hi) */
int mifwt_dwt_inv(const mifwt_level_desc* desc, const void* approx, const void* const* details, void* y,
                  const double* rec_lo, const double* rec_hi, void* workspace, size_t workspace_bytes,
                  void* stream);

/* DEVICE-RESIDENT TAPS (rounds 5 / 6).  The entry points above take the filter as HOST doubles (the kernels receive it in their launch
 * arguments); a learnable filter bank that lives on the GPU (the reference keeps its taps as tensors with autograd,
 * src/ptwt/_util.py:115-132; examples/network_compression/wavelet_linear.py:118,150) would have to be copied to the host — a stream
 * synchronisation — on every call.  These four take DEVICE pointers to L doubles each; the kernels read the taps themselves: no copy,
 * no synchronisation, capturable into a HIP graph.  Which kernel serves a call: mifwt_kernel_id_dtaps(desc, direction) — the fused 2-D
 * kernels where they read device taps (round 6: LDS tiles, ids 7 / 8: every dtype and filter length they serve; one level through the
 * streaming kernels, ids 16 / 22; the border kernels of the analysis adjoint; the streaming axis passes, ids 3 / 4: what a learnable-wavelet training step on image-sized
 * planes runs on, src/ptwt/_util.py:115-132), the generic per-axis passes (id 0: every dtype, mode, stride set and filter length)
 * everywhere else.  A kernel reads the L doubles once, when it starts, into the registers its by-value taps would occupy (same
 * conversion, same packing): results are bit-identical to the host-tap entry points on the same kernel.
 * Workspace: mifwt_workspace_bytes_dtaps(desc, direction), direction as for mifwt_workspace_bytes (0 .. 3). */
int mifwt_kernel_id_dtaps(const mifwt_level_desc* desc, int direction);
size_t mifwt_workspace_bytes_dtaps(const mifwt_level_desc* desc, int direction);
int mifwt_dwt_fwd_dtaps(const mifwt_level_desc* desc, const void* x, void* approx, void* const* details, const double* d_dec_lo,
                        const double* d_dec_hi, void* workspace, size_t workspace_bytes, void* stream);
int mifwt_dwt_inv_dtaps(const mifwt_level_desc* desc, const void* approx, const void* const* details, void* y, const double* d_rec_lo,
                        const double* d_rec_hi, void* workspace, size_t workspace_bytes, void* stream);
int mifwt_dwt_fwd_adjoint_dtaps(const mifwt_level_desc* desc, const void* g_approx, const void* const* g_details, void* g_x,
                                const double* d_dec_lo, const double* d_dec_hi, void* workspace, size_t workspace_bytes, void* stream);
int mifwt_dwt_inv_adjoint_dtaps(const mifwt_level_desc* desc, const void* g_y, void* g_approx, void* const* g_details,
                                const double* d_rec_lo, const double* d_rec_hi, void* workspace, size_t workspace_bytes, void* stream);

/* Round 6, for the tap gradients of 2-D levels with every operand in its natural layout (no transposed copies):
 * mifwt_dwt1_fwd_outer — one 1-D analysis level along the MIDDLE axis of [batch, n, inner] arrays (inner contiguous; element strides):
 *   lo_out / hi_out [batch, floor((n + L - 1) / 2), inner]; taps as HOST doubles (dec_lo / dec_hi) or, when d_dec_lo / d_dec_hi are non-null,
 *   DEVICE doubles (the host pointers are then ignored).  f32 / f64, the filter lengths of the streaming axis kernels (even L <= 20, 24, 32).
 * mifwt_dwt1_inv_outer — its inverse: (lo, hi) [batch, m, inner] -> y [batch, n_out, inner], n_out = 2 m - L + 2 or one less (the reference's
 *   crop, src/ptwt/_util.py:231-244).
 * mifwt_tap_correlate_planes — mifwt_tap_correlate's reduction on operands [batch, rows, columns] (columns contiguous; batch / row strides in
 *   elements): along = 1: out[t] += sum a[b, r, k] b_ext[b, r, 2k + c0 + sgn t] (a: [batch, R, M], b: [batch, R, N]);
 *   along = 0: out[t] += sum a[b, k, c] b_ext[b, 2k + c0 + sgn t, c] (a: [batch, M, C], b: [batch, N, C]).  L <= 32, f32 / f64. */
int mifwt_dwt1_fwd_outer(int dtype, int64_t batch, int64_t n, int64_t inner, const void* x, int64_t x_batch_stride, int64_t x_axis_stride, void* lo_out,
                         void* hi_out, int64_t out_batch_stride, int64_t out_axis_stride, int mode, int filt_len, const double* dec_lo,
                         const double* dec_hi, const double* d_dec_lo, const double* d_dec_hi, void* stream);
int mifwt_dwt1_inv_outer(int dtype, int64_t batch, int64_t m, int64_t n_out, int64_t inner, const void* lo_in, int64_t lo_batch_stride,
                         int64_t lo_axis_stride, const void* hi_in, int64_t hi_batch_stride, int64_t hi_axis_stride, void* y, int64_t y_batch_stride,
                         int64_t y_axis_stride, int filt_len, const double* rec_lo, const double* rec_hi, const double* d_rec_lo,
                         const double* d_rec_hi, void* stream);
int mifwt_tap_correlate_planes(int dtype, int along, int64_t batch, int64_t a_rows, int64_t a_cols, int64_t b_rows, int64_t b_cols, const void* a,
                               int64_t a_batch_stride, int64_t a_row_stride, const void* b, int64_t b_batch_stride, int64_t b_row_stride,
                               int filt_len, int c0, int sgn, int mode, double* out, void* stream);

/* TWO consecutive 2-D analysis levels in one launch — two trips of the reference's level loop
 * (src/ptwt/conv_transform_2.py:142-149) whose intermediate approximation never reaches HBM: a pyramid returns only the
 * detail bands of a level that is not the last (conv_transform_2.py:150-156), so the write + re-read of that
 * approximation is traffic the per-level seam cannot avoid.  d1 / d2 describe the two levels exactly as two
 * mifwt_dwt_fwd calls would (d2->sig_extent == d1->coef_extent; d1's approx_stride and d2's sig_stride are ignored).
 *   details1  HOST array of 3 device ptrs: level-1 bands ad, da, dd     approx2 / details2: the level-2 bands
 * Results are bit-identical to the two per-level calls.  mifwt_dwt2_fwd_pair_supported() says (1 / 0) whether this
 * build serves the pair (f32, even L <= 8, any mode but periodic, unit innermost strides, level-1 plane at least
 * 64 columns x L + 6 rows); the call itself returns MIFWT_ERR_UNSUPPORTED otherwise and launches nothing. */
int mifwt_dwt2_fwd_pair_supported(const mifwt_level_desc* d1, const mifwt_level_desc* d2);
int mifwt_dwt2_fwd_pair(const mifwt_level_desc* d1, const mifwt_level_desc* d2, const void* x, void* const* details1,
                        void* approx2, void* const* details2, const double* dec_lo, const double* dec_hi, void* stream);

/* SEVERAL consecutive 2-D analysis levels in one launch — `nlevels` trips of the reference's level loop
 * (src/ptwt/conv_transform_2.py:142-149); none of the approximations in between reaches HBM (a pyramid returns only the detail
 * bands of a level that is not the last, conv_transform_2.py:150-156).  descs[l] describes fused level l exactly as a
 * mifwt_dwt_fwd call would (descs[l]->sig_extent == descs[l-1]->coef_extent; approx_stride of every level but the last and
 * sig_stride of every level but the first are ignored).
 *   details  HOST array of nlevels HOST arrays of 3 device ptrs: bands ad, da, dd of fused level l
 *   approx   band aa of the last fused level
 * Same sums as nlevels mifwt_dwt_fwd calls (summation order differs: agreement to rounding, not bit for bit).  Two kernels:
 *   (1) kernel id 16, up to THREE levels, rows streamed through registers and LDS rings: f32, even L <= 8, modes zero / constant /
 *       reflect / symmetric, unit innermost strides (input rows of any length and alignment), every fused plane at least 2 L
 *       samples per axis; in auto mode (MIFWT_OPT_PYRAMID_MODE 0) only for planes of 448 .. ~2560 columns (one or two column groups),
 *       where a workgroup streams whole rows; the three detail planes of a level within 1 GiB of one another;
 *   (2) kernel id 20, up to EIGHT levels — the whole pyramid — of planes small enough to live in LDS (the plane and its
 *       horizontally filtered image, both with their boundary extension, <= 160 KB: 128 x 128 up to 12 taps), a workgroup per
 *       image at a time: f32, even L <= 20, every boundary mode, unit innermost strides; in auto mode planes that keep a
 *       CU's LDS to themselves only from 112 KB of LDS images and 512 images upwards (MIFWT_OPT_PYRAMID_MODE 3 lifts that).
 * mifwt_dwt2_fwd_pyramid_supported says which one serves the call (0 none, 1, 2); MIFWT_ERR_UNSUPPORTED otherwise, nothing launched. */
int mifwt_dwt2_fwd_pyramid_supported(int nlevels, const mifwt_level_desc* const* descs);
int mifwt_dwt2_fwd_pyramid(int nlevels, const mifwt_level_desc* const* descs, const void* x, void* const* const* details, void* approx,
                           const double* dec_lo, const double* dec_hi, void* stream);
/* Kernel (1) runs PERSISTENT workgroups: the rows of the last fused level of all images, laid end to end, are cut into one chunk per
 * workgroup (one workgroup per CU and column group; chunks of equal modelled time), a workgroup runs the parts of images in its chunk
 * one after the other (a part that starts inside an image streams a prologue of (2^nlevels - 1)(L - 2) input rows first).  Results do
 * not depend on the cut.  mifwt_dwt2_fwd_pyramid_schedule writes the cuts of a call into wg_start[0 .. n] (global row indices,
 * wg_start[0] = 0, wg_start[n] = batch x rows) and returns n — diagnostics and tests; MIFWT_ERR_UNSUPPORTED where kernel (1) does not
 * serve the call, MIFWT_ERR_BADARG if `capacity` < n + 1. */
int mifwt_dwt2_fwd_pyramid_schedule(int nlevels, const mifwt_level_desc* const* descs, unsigned int* wg_start, int capacity);

/* Plan of the SLAB form of kernel id 24 for one 3-D analysis level (eight / ten taps, f32, rows of at most 128 samples;
 * csrc/mifwt_dwt3_fwd_slab.hip — one level of wavedec3 / fswavedec3, src/ptwt/conv_transform_3.py:121-141) — diagnostics and tests:
 * out[0 .. 11] = output rows of a slab, slabs per volume, compute waves of a workgroup, floats of a staged row, rows per 1-KiB
 * request, staged rows of a slice, (low, high) pairs of a row of the filtered image, pad samples of a row, depth segments, output
 * slices per segment, bytes of LDS, 1 if the default route takes this form for the level (else 0).  Returns 12; MIFWT_ERR_UNSUPPORTED
 * where the form does not apply, MIFWT_ERR_BADARG if `capacity` < 12. */
int mifwt_dwt3_fwd_slab_plan(const mifwt_level_desc* desc, int* out, int capacity);

/* SEVERAL levels of a 2-D reconstruction in one launch — trips of waverec2's level loop (src/ptwt/conv_transform_2.py:222-249);
 * the running approximation never reaches HBM.
 * descs[0] describes the COARSEST level, descs[nlevels-1] the finest, each exactly as a mifwt_dwt_inv call would: coef_extent = the
 * level's coefficient extents, detail_stride its bands' strides; approx_stride counts for descs[0] only (the coarsest approximation),
 * sig_extent / sig_stride for the last one only (y).  The output of level l is cropped to descs[l+1]->coef_extent (which must not
 * exceed 2 M - L + 2 per axis): the reference's trims (conv_transform_2.py:240-247) and the separable containers' crop
 * (separable_conv_transform.py:94-97) are both that.
 *   approx   the coarsest approximation        details  HOST array of nlevels HOST arrays of 3 device ptrs: bands ad, da, dd
 * Same sums as nlevels mifwt_dwt_inv calls (summation order differs: agreement to rounding).  Two kernels:
 *   (1) kernel id 21, EVERY level (up to eight) of a plane small enough to live in LDS: f32, even L <= 20, dense coefficient planes
 *       (row stride = width), the finest level's coefficients and its vertically synthesised image <= 160 KB (128 x 128 outputs for 8
 *       taps); in auto mode planes that keep a CU's LDS to themselves only from 128 KB of LDS images and 512 images upwards
 *       (MIFWT_OPT_PYRAMID_MODE 3 lifts that);
 *   (2) kernel id 22, up to THREE levels of a big plane, coefficient rows streamed through LDS rings (a caller with more levels runs the
 *       coarser ones first and hands their output over as `approx`): f32, even L <= 10, unit innermost strides, the three detail bands
 *       of a level sharing their strides (any row / image stride, any alignment), output planes of at least 32 rows whose rows fit a
 *       workgroup (about 1500 columns); in auto mode (MIFWT_OPT_PYRAMID_MODE 0) planes from 384 columns on.
 * mifwt_dwt2_inv_pyramid_supported says which one serves the call (0 none, 1, 2; MIFWT_OPT_PYRAMID_MODE 2 switches both off);
 * MIFWT_ERR_UNSUPPORTED otherwise, nothing launched. */
int mifwt_dwt2_inv_pyramid_supported(int nlevels, const mifwt_level_desc* const* descs);
int mifwt_dwt2_inv_pyramid(int nlevels, const mifwt_level_desc* const* descs, const void* approx, const void* const* const* details, void* y,
                           const double* rec_lo, const double* rec_hi, void* stream);

/* TWO consecutive 2-D synthesis levels in one launch — two trips of waverec2's level loop
 * (src/ptwt/conv_transform_2.py:222-249); the approximation between them (the coarser level's cropped output) never
 * reaches HBM.  d2 describes the COARSER level, d1 the finer one, exactly as two mifwt_dwt_inv calls would
 * (d2->sig_extent == d1->coef_extent; d2's sig_stride and d1's approx_stride are ignored).
 *   approx2 / details2  the coarser level's bands      details1  HOST array of 3 device ptrs: finer bands ad, da, dd
 * Results are bit-identical to the two per-level calls.  f32, even L <= 8, unit innermost strides, output plane >= 64 x 64
 * (mifwt_dwt2_inv_pair_supported says 1 / 0); MIFWT_ERR_UNSUPPORTED otherwise, nothing launched. */
int mifwt_dwt2_inv_pair_supported(const mifwt_level_desc* d2, const mifwt_level_desc* d1);
int mifwt_dwt2_inv_pair(const mifwt_level_desc* d2, const mifwt_level_desc* d1, const void* approx2, const void* const* details2,
                        const void* const* details1, void* y, const double* rec_lo, const double* rec_hi, void* stream);

/* The DEEP levels of a 1-D decomposition in one launch — the trailing trips of wavedec's level loop
 * (src/ptwt/conv_transform.py:133-140).  Below a few thousand samples per row a level is launch latency, not work; here one
 * workgroup per row parks the row in LDS and runs `nlevels` levels on it.  Rows of contiguous samples (row strides in
 * elements), f32 / f64, even filt_len <= 32, any boundary mode, n <= mifwt_dwt1_fwd_tail_max_n(dtype), 2 <= nlevels <= 24.
 *   details  HOST array of nlevels device pointers: detail coefficients of fused level l, [rows, floor((n_l + L - 1) / 2)]
 *   approx   the last level's approximation
 * Same sums as nlevels mifwt_dwt_fwd calls (summation order differs: agreement to rounding, not bit for bit).
 * MIFWT_ERR_UNSUPPORTED outside the envelope, nothing launched. */
int mifwt_dwt1_fwd_tail_max_n(int dtype);
int mifwt_dwt1_fwd_tail(int dtype, int filt_len, int mode, int64_t rows, int64_t n, int nlevels, const void* x, int64_t x_row_stride,
                        void* approx, int64_t approx_row_stride, void* const* details, const int64_t* detail_row_strides,
                        const double* dec_lo, const double* dec_hi, void* stream);

/* SEVERAL levels of a 1-D decomposition in one launch, a CHUNK of a row per workgroup — the leading trips of wavedec's level
 * loop (src/ptwt/conv_transform.py:133-140) for rows of at least 4096 samples (longer than one workgroup's LDS or not: the
 * one-workgroup-per-row launch is a latency chain; few rows get smaller chunks, about one workgroup per CU; the two end pieces may
 * be the whole row): a workgroup owns a chunk of a row plus the
 * (L - 2)(2^K - 1) halo samples its K levels consume; the approximations between the levels stay in LDS.  Same arguments as
 * mifwt_dwt1_fwd_tail; f32, even filt_len <= 20, any boundary mode.
 * mifwt_dwt1_fwd_long_levels answers how many of `want` levels one launch fuses for this geometry (it stops where the halo
 * would exceed a twelfth of a chunk; 0 = not served); mifwt_dwt1_fwd_long must be called with exactly that count, MIFWT_ERR_UNSUPPORTED otherwise, nothing
 * launched.  Agreement with per-level calls to rounding. */
int mifwt_dwt1_fwd_long_levels(int dtype, int filt_len, int mode, int64_t rows, int64_t n, int want);
int mifwt_dwt1_fwd_long(int dtype, int filt_len, int mode, int64_t rows, int64_t n, int nlevels, const void* x, int64_t x_row_stride,
                        void* approx, int64_t approx_row_stride, void* const* details, const int64_t* detail_row_strides,
                        const double* dec_lo, const double* dec_hi, void* stream);

/* The COARSE levels of a 1-D reconstruction in one launch — the leading trips of waverec's level loop
 * (src/ptwt/conv_transform.py:184-199: stack + conv_transpose1d(stride 2) + crop per level), mirror of mifwt_dwt1_fwd_tail.
 *   approx   coarsest approximation [rows, m]       details  HOST array of nlevels device ptrs, coarsest first: [rows, m_l]
 *   out_len  HOST array: output samples of fused level l = 2 m_l - L + 2 - t, t in {0, 1} (the reference's end-crop,
 *            src/ptwt/_util.py:231-244); m_{l+1} = out_len[l]
 *   y        output of the last fused level [rows, out_len[nlevels - 1]]
 * f32 / f64, even filt_len <= 32, every out_len <= mifwt_dwt1_fwd_tail_max_n(dtype), 2 <= nlevels <= 24. */
int mifwt_dwt1_inv_tail(int dtype, int filt_len, int64_t rows, int64_t m, int nlevels, const void* approx, int64_t approx_row_stride,
                        const void* const* details, const int64_t* detail_row_strides, const int32_t* out_len, void* y,
                        int64_t y_row_stride, const double* rec_lo, const double* rec_hi, void* stream);

/* The FINEST levels of a 1-D reconstruction in one launch, a chunk of the output row per workgroup — the trailing trips of
 * waverec's level loop (src/ptwt/conv_transform.py:184-199) once a level's output no longer fits into one workgroup's LDS
 * (mirror of mifwt_dwt1_fwd_long; the coarse levels before it are mifwt_dwt1_inv_tail's).
 *   m        HOST array of nlevels + 1 ints: m[s] = coefficients per row entering fused step s (coarsest first) — the length
 *            of approx for s = 0, of details[s] for every s — and m[s + 1] = 2 m[s] - L + 2 - t its output length (t in {0, 1}:
 *            the reference's end-crop, src/ptwt/_util.py:231-244); m[nlevels] = samples per row of y
 *   approx   [rows, m[0]]     details  HOST array of nlevels device ptrs, coarsest first: [rows, m[s]]     y  [rows, m[nlevels]]
 * f32 / f64, even filt_len <= 20, 2 <= nlevels <= 8 with (L/2) 2^nlevels below a twelfth of a chunk (8 K f32 / 4 K f64 samples;
 * smaller for few short rows), output rows of at least 1024 samples (mifwt_dwt1_inv_long_supported says 1 / 0); MIFWT_ERR_UNSUPPORTED otherwise, nothing
 * launched.  Agreement with per-level calls to rounding.  Kernel id 18. */
int mifwt_dwt1_inv_long_supported(int dtype, int filt_len, int64_t rows, int nlevels, const int32_t* m);
int mifwt_dwt1_inv_long(int dtype, int filt_len, int64_t rows, int nlevels, const int32_t* m, const void* approx, int64_t approx_row_stride,
                        const void* const* details, const int64_t* detail_row_strides, void* y, int64_t y_row_stride,
                        const double* rec_lo, const double* rec_hi, void* stream);

/* Diagnostic: the launch geometry mifwt_dwt1_fwd_long (inverse = 0: mode, n, nlevels = the `want` argument; m ignored) or
 * mifwt_dwt1_inv_long (inverse = 1: nlevels, m; mode, n ignored) would use: out6 = {levels, chunk, nchunks, end_l, end_r, cap}
 * (chunk: level-K outputs per interior workgroup / output samples per workgroup; end_l, end_r: level-K outputs of the two end
 * pieces, analysis only; cap: floats of LDS buffer A).  Returns 1, or 0 when the geometry is not served. */
int mifwt_dwt1_long_plan(int inverse, int dtype, int filt_len, int mode, int64_t rows, int64_t n, int nlevels, const int32_t* m, int32_t* out6);

/* Adjoints (transposes) of the two level maps, for reverse-mode differentiation.  The reference gets them from
 * ATen autograd through F.pad / _pad_symmetric + F.conv{1,2,3}d and torch.stack + F.conv_transpose{1,2,3}d
 * (same call sites as above); here they are explicit entry points that take the SAME descriptor as the
 * forward call they differentiate.
 *   mifwt_dwt_fwd_adjoint: g_x = A^T (g_approx, g_details), A = the analysis level incl. its boundary extension
 *       (the halo of the transposed convolution is folded back through the boundary index map).
 *   mifwt_dwt_inv_adjoint: (g_approx, g_details) = S^T g_y, S = the (cropped) synthesis level.
 * Taps are the forward call's taps (dec_* resp. rec_*), PyWavelets order. */
int mifwt_dwt_fwd_adjoint(const mifwt_level_desc* desc, const void* g_approx, const void* const* g_details, void* g_x,
                          const double* dec_lo, const double* dec_hi, void* workspace, size_t workspace_bytes,
                          void* stream);
int mifwt_dwt_inv_adjoint(const mifwt_level_desc* desc, const void* g_y, void* g_approx, void* const* g_details,
                          const double* rec_lo, const double* rec_hi, void* workspace, size_t workspace_bytes,
                          void* stream);

/* Stationary (undecimated) transform levels — ptwt.swt / ptwt.iswt, src/ptwt/stationary_transform.py:95-107
 * (_circular_pad + F.conv1d(stride 1, dilation) + split) and :142-156 (stack + _circular_pad +
 * F.conv_transpose1d(groups 2, dilation) + mean).  `rows` independent rows of `n` contiguous samples, row strides in
 * elements, dilation = 2^level_index, periodic extension as an index map.  `scale` multiplies the result: 1 for swt,
 * 0.5 for iswt (the reference's mean over the two reconstructions); with reversed taps and the other scale each
 * call is the adjoint of the other (backward passes).  Even filt_len <= MIFWT_MAX_FILT (2..20 unrolled, longer
 * filters through a run-time tap loop), f32 / f64 / f16.
 *   fwd: lo/hi[n] = scale * sum_m dec_lo/hi[m] x[(n + D (L/2 - m)) mod N]
 *   inv: y[n]     = scale * sum_j rec_lo[j] a[(n + D (L/2 - 1 - j)) mod N] + rec_hi[j] d[(same)] */
int mifwt_swt_fwd(int dtype, int filt_len, int64_t rows, int64_t n, int64_t dilation, const void* x, int64_t x_row_stride,
                  void* lo, void* hi, int64_t lo_row_stride, int64_t hi_row_stride, const double* dec_lo,
                  const double* dec_hi, double scale, void* stream);
int mifwt_swt_inv(int dtype, int filt_len, int64_t rows, int64_t n, int64_t dilation, const void* a, const void* d,
                  int64_t a_row_stride, int64_t d_row_stride, void* y, int64_t y_row_stride, const double* rec_lo,
                  const double* rec_hi, double scale, void* stream);

/* Reduction behind the gradients w.r.t. the FILTER TAPS (learnable wavelets: src/ptwt/wavelets_learnable.py; the reference
 * gets them from ATen's conv backward because its taps stay in the autograd graph, src/ptwt/_util.py:132):
 *     out[t] += sum_{row < rows} sum_{k < m_len} a[row, k] * b_ext[row, 2k + c0 + sgn * t],     t in [0, filt_len)
 * a: rows x m_len, b: rows x n_len (contiguous samples, row strides in elements), b extended by `mode` (enum mifwt_mode);
 * out: DEVICE array of filt_len doubles, accumulated into (zero it first).  f32 / f64.
 *   analysis level,  dL/d dec[m]:  a = upstream gradient of the band, b = level input, c0 = 1,        sgn = -1, mode = level's
 *   synthesis level, dL/d rec[t]:  a = the band,                     b = upstream gradient of y, c0 = -(L-2), sgn = +1, zero mode
 * (for N-D levels a / b are taken after the other axes have been transformed; the host layer composes that). */
int mifwt_tap_correlate(int dtype, int64_t rows, int64_t m_len, int64_t n_len, const void* a, int64_t a_row_stride,
                        const void* b, int64_t b_row_stride, int filt_len, int c0, int sgn, int mode, double* out,
                        void* stream);
/* The same reduction for the STATIONARY levels (mifwt_swt_fwd / mifwt_swt_inv: stride 1, dilation D, periodic with any number of
 * wraps):     out[t] += sum_{row < rows} sum_{k < n} a[row, k] * b[row, (k + c0 + tstep * t) mod n],     t in [0, filt_len)
 *   swt level,  dL/d dec[m]:  a = upstream gradient of the band, b = level input,        c0 = D L/2,       tstep = -D
 *   iswt level, dL/d rec[j]:  a = upstream gradient of y,        b = the band (a or d),  c0 = D (L/2 - 1), tstep = -D  (times the
 *   level's scale, which the host applies).  f32 / f64. */
int mifwt_tap_correlate_dilated(int dtype, int64_t rows, int64_t n, const void* a, int64_t a_row_stride, const void* b,
                                int64_t b_row_stride, int filt_len, int64_t c0, int64_t tstep, double* out, void* stream);

/* Scratch bytes one call needs (0 on the fused paths).  direction: 0 = analysis, 1 = synthesis,
 * 2 = mifwt_dwt_fwd_adjoint, 3 = mifwt_dwt_inv_adjoint (same numbering for mifwt_kernel_id). */
size_t mifwt_workspace_bytes(const mifwt_level_desc* desc, int direction);

/* Which kernel family a call would dispatch to (tests use it to assert the fast path is the one that ran);
 * negative = error code.
 *   0  generic per-axis passes (any strides, any L <= 128; f32 / f64 / f16 storage — the fallback of every call no fused kernel takes)
 *   1 / 2  fused single-launch 2-D analysis / synthesis level, streaming wave strips (f32, even L <= 16)
 *   7 / 8  fused single-launch 2-D analysis / synthesis level, LDS tiles (f32 / f16, even L <= 20, 24, 32; f64, even L <= 16); the
 *          default 2-D kernels — the streaming analysis kernel is kept for 16-tap filters on planes >= ~1500^2
 *   3 / 4  streaming axis passes, analysis / synthesis: inner-axis kernel (+ one outer-axis pass per further
 *          axis); unit innermost stride, f32 / f64 / f16, L in {2..20 even, 24, 32}
 *   5 / 6  3-D analysis / synthesis (f32, even L <= 16; what the brick kernels 9 / 10 do not take): fused 2-D kernel over every
 *          depth slice + one streaming pass along depth
 *   9 / 10  fully fused 3-D analysis / synthesis level, LDS bricks (f32, L in {2, 4, 6}; synthesis also 8): the small volumes
 *   24 / 25  fully fused 3-D analysis / synthesis level, workgroups walking along the depth axis (f32 and f64; analysis: even L <= 10,
 *          every mode, rows <= 512 samples (f64: 256), picked from 2^22 samples per volume on and for 8 taps; synthesis: even L <= 8,
 *          dense coefficient rows, picked from 2^20 output samples on; f64, which has no bricks: analysis from 2^15 / 2^17 / 2^19 samples
 *          on for <= 4 / 6 / 8 taps, synthesis from 2^16 (<= 4 taps) / 2^19 output samples on).  Round 6: batches of 8 / 16 volumes and
 *          more from 2^21 / 10^5 samples per volume on (L <= 6); a SLAB form of the analysis kernel for 8 / 10 taps on rows of at most
 *          128 samples (f32, every mode: a workgroup filters every staged row once), picked by volume and batch (mifwt_dwt3_fwd_slab_plan)
 *   11 / 23  fused 2-D analysis / synthesis level on the matrix cores (banded-Toeplitz MFMA; f16 storage, even L in [18, 32]; both walk down
 *          column panels; the synthesis kernel from 16 tiles of 32 x 128 samples per call on, the vector tile kernel 8 below that)
 *   12 / 13  two fused 2-D analysis / synthesis levels per launch (mifwt_dwt2_fwd_pair / mifwt_dwt2_inv_pair; never
 *          returned by mifwt_kernel_id, which describes single-level calls)
 *   14 / 15  the deep levels of a 1-D analysis / the coarse levels of a 1-D synthesis in one launch (mifwt_dwt1_fwd_tail /
 *          mifwt_dwt1_inv_tail; likewise not returned by mifwt_kernel_id)
 *   16     up to three fused 2-D analysis levels per launch (mifwt_dwt2_fwd_pyramid; not returned by mifwt_kernel_id)
 *   17 / 18  several fused 1-D analysis / synthesis levels of long rows, a chunk per workgroup (mifwt_dwt1_fwd_long /
 *          mifwt_dwt1_inv_long; likewise)
 *   20 / 21  every level of a 2-D analysis / synthesis of a small plane in one launch (mifwt_dwt2_fwd_pyramid's second kernel /
 *          mifwt_dwt2_inv_pyramid; likewise)
 *   22     up to three fused 2-D synthesis levels of a big plane per launch (mifwt_dwt2_inv_pyramid's second kernel; likewise) */
int mifwt_kernel_id(const mifwt_level_desc* desc, int direction);

/* Library-wide diagnostic switches (process-global, meant for tests and A/B measurements).
 *   MIFWT_OPT_FORCE_GENERIC (0): non-zero routes every call through the generic per-axis passes.
 *   MIFWT_OPT_ROWS_PER_CHUNK (1): >0 overrides the fused kernels' output rows per streamed chunk.
 *   MIFWT_OPT_PREFETCH_PAIRS (2): >0 overrides the fused kernels' register-ring prefetch depth. */
#define MIFWT_OPT_FORCE_GENERIC 0
#define MIFWT_OPT_ROWS_PER_CHUNK 1
#define MIFWT_OPT_PREFETCH_PAIRS 2 /* >0 overrides the fused kernels' prefetch depth (row pairs in flight) */
#define MIFWT_OPT_RESERVED3 3      /* unused (was: cooperative full-line writer, removed after measurement) */
#define MIFWT_OPT_NT_STORE 4       /* non-zero: nontemporal stores for the sub-band planes */
#define MIFWT_OPT_TILE_MODE 5      /* 2-D analysis: 0 = auto (LDS-tile kernel on small planes), 1 = always tile, 2 = never.  3-D: 0 = auto (depth-walking kernels 24 / 25 on big volumes, bricks 9 / 10 on small ones), 1 = bricks wherever they can run, 2 = composed route, 4 = walking kernels wherever they can run */
#define MIFWT_OPT_TILE_ROWS 6      /* >0 overrides the tile kernels' output rows per tile */
#define MIFWT_OPT_MFMA_MODE 7      /* 2-D analysis, f16 storage, 18..32 taps: 0 = matrix-core kernels (walking down column panels), 2 = vector tile
                                      kernels, 3 = the tile-at-a-time analysis kernel of round 2 (comparisons), 4 = the synthesis kernel for every size */
#define MIFWT_OPT_PAIR_MODE 8      /* multi-level launches (mifwt_dwt2_fwd_pyramid, mifwt_dwt2_fwd_pair, mifwt_dwt2_inv_pair, mifwt_dwt1_fwd_tail): 2 = never (they answer
                                      UNSUPPORTED / 0); analysis pairs: 0 = auto (rolling strips for 8 taps, else tiles), 1 = tiles only,
                                      3 = rolling strips wherever they apply */
#define MIFWT_OPT_PAIR_ROWS 9      /* >0 overrides the pair kernels' level-2 rows per tile (4, 6, 8, 12) / per strip segment (multiple of 8) */
#define MIFWT_OPT_PYRAMID_MODE 12 /* mifwt_dwt2_fwd_pyramid: 0 = auto (each of its two kernels where it is the fastest route), 1 = the streaming kernel wherever it can run, 3 = the small-plane kernel wherever it can run, 2 = never (it answers UNSUPPORTED / 0; the two-level and per-level kernels then serve the call) */
#define MIFWT_OPT_DEBUG 11        /* ROUTING bits: alternative code paths with the SAME results (A/B runs, parity tests): 8 = the streaming synthesis kernel (id 22) without its fast warm-up, 64 = column strips of 64 instead of balanced strips (3-D analysis walk), 512 = 8-byte instead of 16-byte output stores (3-D synthesis walk), 1024 = analysis adjoints with a boundary extension on the generic per-axis passes instead of synthesis launch + border kernel, 4096 = the border part of a 2-D analysis adjoint on the one-thread-per-sample kernel instead of the one-thread-per-border-line kernel, 8192 = the level-2 waves of the streaming analysis kernel (id 16) behind the step's second barrier, 1 << 19 = kernel 16 without its tail wave, 1 << 20 = kernel 16 in its sixteen-wave form.  Any other bit is a MEASUREMENT switch that breaks results (no stores / no loads / no deep levels / single passes off ...): those are compiled into -DMIFWT_DIAG builds only (tools/; csrc/mifwt_common.h) and the product library answers MIFWT_ERR_UNSUPPORTED to them */
#define MIFWT_OPT_PYR_WGS 13 /* >0: the streaming analysis kernel (id 16) cuts the batch's rows into this many chunks (workgroups per column group) instead of one per CU — parity tests of units that start and end anywhere */
#define MIFWT_OPT_EXP 15           /* experiment word of the A/B run in progress (tools/): -DMIFWT_DIAG builds only; the product library accepts 0 and answers MIFWT_ERR_UNSUPPORTED to anything else */
#define MIFWT_OPT_SYNC_STAGE 10    /* non-zero: tile kernels keep the workgroup barrier between staging and the horizontal pass (A/B) */
int mifwt_set_option(int key, int value);

/* Diagnostic (-DMIFWT_DIAG builds; the product library has no profiling instances and answers MIFWT_ERR_UNSUPPORTED to a non-null
 * buffer): a device buffer of 2 x uint64 per wave of every workgroup that later mifwt_dwt2_fwd_pyramid launches fill with
 * (cycles alive, cycles spent in workgroup barriers); NULL switches it off again. */
int mifwt_pyr_profile_buffer(void* device_buffer);

/* Diagnostic: how many launches of a kernel VARIANT this process has enqueued — variants that share a kernel id (what the tests use to
 * pin "this code path ran"; `variant` out of range: 0). */
#define MIFWT_VARIANT_FWD_MFMA_WALK 0 /* id 11: the analysis kernel that walks down column panels */
#define MIFWT_VARIANT_FWD_MFMA_TILE 1 /* id 11: the tile-at-a-time analysis kernel of round 2 (MIFWT_OPT_MFMA_MODE 3) */
#define MIFWT_VARIANT_FWD_PYR_ST16 2  /* id 16: 16-byte stores after a lane-pair exchange (planes with 16-byte aligned rows) */
#define MIFWT_VARIANT_FWD_PYR_ST8 3   /* id 16: 8-byte stores (any row pitch) */
unsigned long long mifwt_launch_count(int variant);

const char* mifwt_strerror(int code);
int mifwt_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* MIFWT_H */
