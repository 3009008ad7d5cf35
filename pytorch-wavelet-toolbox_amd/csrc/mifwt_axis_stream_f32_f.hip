// mifwt_axis_stream_f32_f.hip — streaming single-axis kernels (mifwt_axis_stream.h): float storage, L = 24.
#include "mifwt_axis_stream.h"

MIFWT_STREAM_DEFINE(f32, float, 24)
