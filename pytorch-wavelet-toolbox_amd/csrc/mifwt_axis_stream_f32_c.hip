// mifwt_axis_stream_f32_c.hip — streaming single-axis kernels (mifwt_axis_stream.h): float storage, L = 14, 16.
#include "mifwt_axis_stream.h"

MIFWT_STREAM_DEFINE(f32, float, 14)
MIFWT_STREAM_DEFINE(f32, float, 16)
