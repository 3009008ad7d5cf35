// mifwt_idwt2_tile_f64.hip — LDS-tile 2-D synthesis kernel (mifwt_idwt2_tile.h): f64 storage and arithmetic, L <= 16.
#include "mifwt_idwt2_tile.h"

namespace mifwt {

int idwt2_tile_f64_short(const mifwt_level_desc* d, const void* approx, const void* const* details, void* y,
                         const double* lo, const double* hi, hipStream_t stream) {
  switch (d->filt_len) {
    case 2: return launch_idwt_tr<double, 2>(d, approx, details, y, lo, hi, stream);
    case 4: return launch_idwt_tr<double, 4>(d, approx, details, y, lo, hi, stream);
    case 6: return launch_idwt_tr<double, 6>(d, approx, details, y, lo, hi, stream);
    case 8: return launch_idwt_tr<double, 8>(d, approx, details, y, lo, hi, stream);
    case 10: return launch_idwt_tr<double, 10>(d, approx, details, y, lo, hi, stream);
    case 12: return launch_idwt_tr<double, 12>(d, approx, details, y, lo, hi, stream);
    case 14: return launch_idwt_tr<double, 14>(d, approx, details, y, lo, hi, stream);
    case 16: return launch_idwt_tr<double, 16>(d, approx, details, y, lo, hi, stream);
    default: return MIFWT_ERR_UNSUPPORTED;
  }
}

}  // namespace mifwt
