// mifwt_idwt2_tile_long_b.hip — LDS-tile 2-D synthesis kernel (mifwt_idwt2_tile.h): 24- and 32-tap filters, f32 / f16.
#include "mifwt_idwt2_tile.h"

namespace mifwt {

int idwt2_tile_long_b(const mifwt_level_desc* d, const void* approx, const void* const* details, void* y,
                          const double* lo, const double* hi, hipStream_t stream) {
  switch (d->filt_len) {
    case 24:
      return d->dtype == MIFWT_F16 ? launch_idwt_tr<_Float16, 24>(d, approx, details, y, lo, hi, stream)
                                   : launch_idwt_tr<float, 24>(d, approx, details, y, lo, hi, stream);
    case 32:
      return d->dtype == MIFWT_F16 ? launch_idwt_tr<_Float16, 32>(d, approx, details, y, lo, hi, stream)
                                   : launch_idwt_tr<float, 32>(d, approx, details, y, lo, hi, stream);
    default: return MIFWT_ERR_UNSUPPORTED;
  }
}

}  // namespace mifwt
