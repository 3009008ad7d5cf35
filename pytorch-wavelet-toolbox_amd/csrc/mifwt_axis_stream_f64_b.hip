// mifwt_axis_stream_f64_b.hip — streaming single-axis kernels (mifwt_axis_stream.h): double storage, L = 10, 12.
#include "mifwt_axis_stream.h"

MIFWT_STREAM_DEFINE(f64, double, 10)
MIFWT_STREAM_DEFINE(f64, double, 12)
