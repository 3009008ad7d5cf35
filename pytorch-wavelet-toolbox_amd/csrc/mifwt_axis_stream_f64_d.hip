// mifwt_axis_stream_f64_d.hip — streaming single-axis kernels (mifwt_axis_stream.h): double storage, L = 18.
#include "mifwt_axis_stream.h"

MIFWT_STREAM_DEFINE(f64, double, 18)
