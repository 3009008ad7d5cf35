// mifwt_axis_stream_f64_f.hip — streaming single-axis kernels (mifwt_axis_stream.h): double storage, L = 24.
#include "mifwt_axis_stream.h"

MIFWT_STREAM_DEFINE(f64, double, 24)
