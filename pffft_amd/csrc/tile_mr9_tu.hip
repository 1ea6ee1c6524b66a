// libpffft_hip.so, translation unit of the tile kernels with an odd first stage of radix 9 (fft_tile.h, tile_host.h).
#include "tile_host.h"

namespace pf {
PF_TILE_MR_TU(9)
}  // namespace pf
