// mifwt_axis_stream_f64_g.hip — streaming single-axis kernels (mifwt_axis_stream.h): double storage, L = 32.
#include "mifwt_axis_stream.h"

MIFWT_STREAM_DEFINE(f64, double, 32)
