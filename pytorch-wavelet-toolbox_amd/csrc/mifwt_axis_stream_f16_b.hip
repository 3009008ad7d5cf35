// mifwt_axis_stream_f16_b.hip — streaming single-axis kernels (mifwt_axis_stream.h): _Float16 storage, L = 10, 12.
#include "mifwt_axis_stream.h"

MIFWT_STREAM_DEFINE(f16, _Float16, 10)
MIFWT_STREAM_DEFINE(f16, _Float16, 12)
