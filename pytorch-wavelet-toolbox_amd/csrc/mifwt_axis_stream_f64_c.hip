// mifwt_axis_stream_f64_c.hip — streaming single-axis kernels (mifwt_axis_stream.h): double storage, L = 14, 16.
#include "mifwt_axis_stream.h"

MIFWT_STREAM_DEFINE(f64, double, 14)
MIFWT_STREAM_DEFINE(f64, double, 16)
