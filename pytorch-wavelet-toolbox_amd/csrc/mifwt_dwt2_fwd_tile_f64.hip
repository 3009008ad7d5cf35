// mifwt_dwt2_fwd_tile_f64.hip — instantiations of the LDS-tile 2-D analysis kernel (mifwt_dwt2_tile.h): f64 storage and
// arithmetic (the reference's second supported dtype, src/ptwt/constants.py:27), L <= 16.
#include "mifwt_dwt2_tile.h"

namespace mifwt {

int dwt2_fwd_tile_f64_short(const mifwt_level_desc* d, const void* x, void* approx, void* const* details,
                            const double* lo, const double* hi, hipStream_t stream) {
  switch (d->filt_len) {
    case 2: return launch_tr<double, 2>(d, x, approx, details, lo, hi, stream);
    case 4: return launch_tr<double, 4>(d, x, approx, details, lo, hi, stream);
    case 6: return launch_tr<double, 6>(d, x, approx, details, lo, hi, stream);
    case 8: return launch_tr<double, 8>(d, x, approx, details, lo, hi, stream);
    case 10: return launch_tr<double, 10>(d, x, approx, details, lo, hi, stream);
    case 12: return launch_tr<double, 12>(d, x, approx, details, lo, hi, stream);
    case 14: return launch_tr<double, 14>(d, x, approx, details, lo, hi, stream);
    case 16: return launch_tr<double, 16>(d, x, approx, details, lo, hi, stream);
    default: return MIFWT_ERR_UNSUPPORTED;
  }
}

}  // namespace mifwt
