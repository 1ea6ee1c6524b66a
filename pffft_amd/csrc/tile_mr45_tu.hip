// libpffft_hip.so, translation unit of the tile kernels with two odd stages, odd part 45 (fft_tile.h, tile_host.h).
#include "tile_host.h"

namespace pf {
PF_TILE_MR_TU(45)
}  // namespace pf
