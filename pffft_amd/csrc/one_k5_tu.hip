// libpffft_hip.so: fft_one.h instantiated for flag set 5 (bit 0 internal layout in, bit 1 out, bit 2 backward, bit 3 real)
#include "fft_one.h"
#include "one_k.h"
namespace pf {
PF_ONE_K_TU(5)
}
