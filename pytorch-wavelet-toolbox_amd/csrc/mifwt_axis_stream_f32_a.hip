// mifwt_axis_stream_f32_a.hip — streaming single-axis kernels (mifwt_axis_stream.h): float storage, L = 2, 4, 6, 8.
#include "mifwt_axis_stream.h"

MIFWT_STREAM_DEFINE(f32, float, 2)
MIFWT_STREAM_DEFINE(f32, float, 4)
MIFWT_STREAM_DEFINE(f32, float, 6)
MIFWT_STREAM_DEFINE(f32, float, 8)
