// mifwt_axis_stream_f16_a.hip — streaming single-axis kernels (mifwt_axis_stream.h): _Float16 storage, L = 2, 4, 6, 8.
#include "mifwt_axis_stream.h"

MIFWT_STREAM_DEFINE(f16, _Float16, 2)
MIFWT_STREAM_DEFINE(f16, _Float16, 4)
MIFWT_STREAM_DEFINE(f16, _Float16, 6)
MIFWT_STREAM_DEFINE(f16, _Float16, 8)
