// libpffft_hip.so, translation unit of the tile kernels with two odd stages, odd part 25 (fft_tile.h, tile_host.h).
#include "tile_host.h"

namespace pf {
PF_TILE_MR_TU(25)
}  // namespace pf
