"""pffft_amd — host-side Python mirror of the pffft C API over libpffft_hip.so (MI355X / gfx950).

The product is the C-ABI shared library (include/pffft_hip.h).  This package is the thin ctypes
layer the tests and bench.py use to reach it: same names, argument meaning and error behaviour as
the reference API (include/pffft/pffft.h:124-250 in the reference tree).  There is no CPU fallback:
importing works anywhere, but every transform needs the HIP library and a GPU and raises otherwise.
"""
from .api import (BACKWARD, COMPLEX, FORWARD, REAL, FastConv, Setup, device_count, error_count, is_valid_size, last_error,
                  kernel_name, describe, setup_devices, tile_plan, lib, lib_path, min_fft_size, nearest_transform_size, set_variant, has_variants,
                  simd_arch, simd_size)

__all__ = ["Setup", "FastConv", "FORWARD", "BACKWARD", "REAL", "COMPLEX", "lib", "lib_path",
           "device_count", "simd_size", "simd_arch", "min_fft_size", "is_valid_size",
           "nearest_transform_size", "kernel_name", "describe", "setup_devices", "tile_plan", "set_variant", "has_variants", "error_count", "last_error"]
