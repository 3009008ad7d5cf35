// mifwt_dwt2_fwd_tile_long20.hip — LDS-tile 2-D analysis kernel (mifwt_dwt2_tile.h): 20-tap filters, f32 and f16 storage.
#include "mifwt_dwt2_tile.h"

namespace mifwt {

int dwt2_fwd_tile_long20(const mifwt_level_desc* d, const void* x, void* approx, void* const* details, const double* lo,
                         const double* hi, hipStream_t stream) {
  if (d->dtype == MIFWT_F16) return launch_tr<_Float16, 20>(d, x, approx, details, lo, hi, stream);
  return launch_tr<float, 20>(d, x, approx, details, lo, hi, stream);
}

}  // namespace mifwt
