// mifwt_axis_stream_f64_a.hip — streaming single-axis kernels (mifwt_axis_stream.h): double storage, L = 2, 4, 6, 8.
#include "mifwt_axis_stream.h"

MIFWT_STREAM_DEFINE(f64, double, 2)
MIFWT_STREAM_DEFINE(f64, double, 4)
MIFWT_STREAM_DEFINE(f64, double, 6)
MIFWT_STREAM_DEFINE(f64, double, 8)
