// mifwt_axis_stream_f32_b.hip — streaming single-axis kernels (mifwt_axis_stream.h): float storage, L = 10, 12.
#include "mifwt_axis_stream.h"

MIFWT_STREAM_DEFINE(f32, float, 10)
MIFWT_STREAM_DEFINE(f32, float, 12)
