// mifwt_axis_stream_f32_e.hip — streaming single-axis kernels (mifwt_axis_stream.h): float storage, L = 20.
#include "mifwt_axis_stream.h"

MIFWT_STREAM_DEFINE(f32, float, 20)
