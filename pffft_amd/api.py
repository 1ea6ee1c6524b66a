"""ctypes binding of libpffft_hip.so.  Mirrors the reference's operator interface:

    reference (C)                                  here
    pffft_new_setup(N, PFFFT_COMPLEX)              Setup(N, COMPLEX, dtype=np.float32)
    pffft_transform(s, in, out, work, dir)         s.transform(x, direction)            (internal layout)
    pffft_transform_ordered(...)                   s.transform_ordered(x, direction)
    pffft_zreorder(s, in, out, dir)                s.zreorder(x, direction)
    pffft_zconvolve_accumulate / _no_accu          s.zconvolve(a, b, ab, scaling, accumulate=...)
    pffastconv_new_setup / _apply                  FastConv(h, block_len, flags).apply(x, flush)

numpy arrays go through the legacy single-vector entries (host pointers, staged by the library);
torch CUDA tensors go through the batched device entries (`*_hip_*_batch`) on the current stream.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

FORWARD, BACKWARD = 0, 1
REAL, COMPLEX = 0, 1

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def lib_path() -> str:
    return os.path.join(_HERE, "libpffft_hip.so")


def lib():
    """Load the C-ABI library; fails loudly when it has not been built."""
    global _LIB
    if _LIB is not None:
        return _LIB
    p = lib_path()
    if not os.path.exists(p):
        raise RuntimeError(f"{p} is missing — build it with `python -m pffft_amd.build` "
                           "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
    # One HIP runtime per process: PyTorch bundles its own libamdhip64.so.7 and hands us its streams
    # and device pointers, so when torch is installed it must be loaded FIRST — the dynamic loader then
    # binds libpffft_hip.so's NEEDED libamdhip64.so.7 to that same instance (matching SONAME).  Loading
    # in the other order gives two runtimes in one process; the second one finds no device.
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    L = C.CDLL(p, mode=getattr(os, "RTLD_LOCAL", 0))
    for pfx, ct in (("pffft", C.c_float), ("pffftd", C.c_double)):
        g = lambda n: getattr(L, f"{pfx}_{n}")
        g("new_setup").restype = C.c_void_p; g("new_setup").argtypes = [C.c_int, C.c_int]
        g("destroy_setup").restype = None; g("destroy_setup").argtypes = [C.c_void_p]
        for n in ("transform", "transform_ordered"):
            g(n).restype = None; g(n).argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        g("zreorder").restype = None; g("zreorder").argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        for n in ("zconvolve_accumulate", "zconvolve_no_accu"):
            g(n).restype = None; g(n).argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, ct]
        g("simd_size").restype = C.c_int; g("simd_size").argtypes = []
        g("simd_arch").restype = C.c_char_p; g("simd_arch").argtypes = []
        g("min_fft_size").restype = C.c_int; g("min_fft_size").argtypes = [C.c_int]
        g("is_valid_size").restype = C.c_int; g("is_valid_size").argtypes = [C.c_int, C.c_int]
        g("nearest_transform_size").restype = C.c_int
        g("nearest_transform_size").argtypes = [C.c_int, C.c_int, C.c_int]
        g("next_power_of_two").restype = C.c_int; g("next_power_of_two").argtypes = [C.c_int]
        g("is_power_of_two").restype = C.c_int; g("is_power_of_two").argtypes = [C.c_int]
        g("aligned_malloc").restype = C.c_void_p; g("aligned_malloc").argtypes = [C.c_size_t]
        g("aligned_free").restype = None; g("aligned_free").argtypes = [C.c_void_p]
        g("hip_transform_batch").restype = C.c_int
        g("hip_transform_batch").argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int,
                                             C.c_int, C.c_void_p]
        g("hip_zreorder_batch").restype = C.c_int
        g("hip_zreorder_batch").argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
        g("hip_zconvolve_batch").restype = C.c_int
        g("hip_zconvolve_batch").argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, ct, C.c_size_t,
                                             C.c_int, C.c_int, C.c_void_p]
        g("hip_convolve_batch").restype = C.c_int
        g("hip_convolve_batch").argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, ct, C.c_size_t,
                                            C.c_int, C.c_int, C.c_void_p]
        getattr(L, f"validate_{pfx}_simd").restype = C.c_int
        getattr(L, f"validate_{pfx}_simd_ex").restype = C.c_int
        getattr(L, f"validate_{pfx}_simd_ex").argtypes = [C.c_void_p]
    L.pffastconv_new_setup.restype = C.c_void_p
    L.pffastconv_new_setup.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.c_int]
    L.pffastconv_destroy_setup.restype = None; L.pffastconv_destroy_setup.argtypes = [C.c_void_p]
    L.pffastconv_apply.restype = C.c_int
    L.pffastconv_apply.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
    L.pffastconv_hip_apply_device.restype = C.c_int
    L.pffastconv_hip_apply_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
    L.pffastconv_hip_apply_batch.restype = C.c_int
    L.pffastconv_hip_apply_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int,
                                             C.c_int, C.c_void_p]
    L.pffft_hip_error_count.restype = C.c_uint
    L.pffastconv_simd_size.restype = C.c_int
    L.pffft_hip_shift_transform_batch.restype = C.c_int
    L.pffft_hip_shift_transform_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_double,
                                                  C.c_double, C.c_void_p]
    L.pffft_hip_kernel_name.restype = C.c_char_p; L.pffft_hip_kernel_name.argtypes = [C.c_void_p]
    L.pffft_hip_describe.restype = C.c_int; L.pffft_hip_describe.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
    L.pffft_hip_setup_devices.restype = C.c_int; L.pffft_hip_setup_devices.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.c_int]
    L.pffft_hip_last_error.restype = C.c_char_p
    L.pffft_hip_device_count.restype = C.c_int
    L.pffft_hip_set_variant.restype = None; L.pffft_hip_set_variant.argtypes = [C.c_int]
    L.pffft_hip_has_variants.restype = C.c_int; L.pffft_hip_has_variants.argtypes = []
    L.pffft_hip_tile_plan.restype = C.c_int
    L.pffft_hip_tile_plan.argtypes = [C.c_longlong, C.c_int, C.c_int, C.POINTER(C.c_int)]
    _LIB = L
    return L


def _pfx(dtype) -> str:
    return "pffftd" if np.dtype(dtype) == np.float64 else "pffft"


def device_count() -> int:
    return lib().pffft_hip_device_count()


def simd_size(dtype=np.float32) -> int:
    return getattr(lib(), f"{_pfx(dtype)}_simd_size")()


def simd_arch(dtype=np.float32) -> str:
    return getattr(lib(), f"{_pfx(dtype)}_simd_arch")().decode()


def min_fft_size(transform, dtype=np.float32) -> int:
    return getattr(lib(), f"{_pfx(dtype)}_min_fft_size")(transform)


def is_valid_size(N, transform, dtype=np.float32) -> bool:
    return bool(getattr(lib(), f"{_pfx(dtype)}_is_valid_size")(N, transform))


def nearest_transform_size(N, transform, higher, dtype=np.float32) -> int:
    return getattr(lib(), f"{_pfx(dtype)}_nearest_transform_size")(N, transform, int(bool(higher)))


def error_count() -> int:
    """Legacy (void) entries that failed soft in this process (include/pffft_hip.h)."""
    return int(lib().pffft_hip_error_count())


def last_error() -> str:
    return lib().pffft_hip_last_error().decode()


def set_variant(v: int) -> None:
    lib().pffft_hip_set_variant(int(v))


def has_variants() -> bool:
    """True for a development build (PFFFT_HIP_VARIANTS=1): both variants of every Stockham plan are instantiated."""
    return bool(lib().pffft_hip_has_variants())


def tile_plan(n, is_double=False, deep=False):
    """Tile lengths of the passes over HBM of a complex core transform of n points beyond LDS ([] = streaming passes / not planned
    by the tile planner): include/pffft_hip.h pffft_hip_tile_plan.  Host arithmetic only."""
    buf = (C.c_int * 3)()
    k = lib().pffft_hip_tile_plan(int(n), int(bool(is_double)), int(deep), buf)
    return [int(buf[i]) for i in range(k)]


def kernel_name(setup: "Setup") -> str:
    return lib().pffft_hip_kernel_name(setup.handle).decode()


def describe(setup: "Setup") -> str:
    """pffft_hip_describe: the routes the planner chose for this setup, one line per (direction, layout)."""
    buf = C.create_string_buffer(4096)
    n = lib().pffft_hip_describe(setup.handle, buf, len(buf))
    if n < 0:
        raise ValueError("pffft_hip_describe: invalid handle")
    return buf.value.decode()


def setup_devices(setup: "Setup"):
    """pffft_hip_setup_devices: the devices this setup holds tables / counters / scratch on (first-use device first)."""
    buf = (C.c_int * 80)()
    n = lib().pffft_hip_setup_devices(setup.handle, buf, 80)
    return [buf[i] for i in range(min(n, 80))]


def _is_torch(x) -> bool:
    return type(x).__module__.startswith("torch")


def _check(rc: int, what: str):
    if rc != 0:
        raise RuntimeError(f"{what} failed ({rc}): {lib().pffft_hip_last_error().decode()}")


def _aligned_empty(n, dtype):
    dtype = np.dtype(dtype)
    raw = np.empty(n * dtype.itemsize + 64, dtype=np.uint8)
    off = (-raw.ctypes.data) % 64
    return raw[off:off + n * dtype.itemsize].view(dtype)


def _aligned(a, dtype):
    a = np.asarray(a, dtype=dtype).ravel()
    if a.ctypes.data % 64 == 0 and a.flags.c_contiguous:
        return a
    out = _aligned_empty(a.size, dtype)
    out[:] = a
    return out


class Setup:
    """PFFFT_Setup / PFFFTD_Setup.  Raises ValueError where pffft_new_setup returns NULL
    (src/pffft_priv_impl.h:1066-1078,1105-1109)."""

    def __init__(self, N: int, transform: int, dtype=np.float32):
        self.N, self.transform_type, self.dtype = int(N), int(transform), np.dtype(dtype)
        self._pfx = _pfx(dtype)
        self._L = lib()
        self.handle = getattr(self._L, f"{self._pfx}_new_setup")(self.N, self.transform_type)
        if not self.handle:
            raise ValueError(f"pffft_new_setup({N}, {transform}) returned NULL")
        self.vec_scalars = self.N * (2 if transform == COMPLEX else 1)

    def close(self):
        if getattr(self, "handle", None):
            getattr(self._L, f"{self._pfx}_destroy_setup")(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---------------- device (torch CUDA tensors): batched entries ----------------
    def _stream(self):
        import torch
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def _tcheck(self, t):
        import torch
        want = torch.float64 if self.dtype == np.float64 else torch.float32
        assert t.is_cuda and t.dtype == want and t.is_contiguous(), "need contiguous CUDA tensor of the setup dtype"
        assert t.numel() % self.vec_scalars == 0
        return t.numel() // self.vec_scalars

    def transform_batch(self, x, out=None, direction=FORWARD, ordered=False):
        """x: CUDA tensor holding `batch` contiguous vectors.  out may be x (in place)."""
        import torch
        batch = self._tcheck(x)
        if out is None:
            out = torch.empty_like(x)
        assert self._tcheck(out) == batch
        fn = getattr(self._L, f"{self._pfx}_hip_transform_batch")
        _check(fn(self.handle, x.data_ptr(), out.data_ptr(), batch, direction, int(bool(ordered)), self._stream()),
               "hip_transform_batch")
        return out

    def shift_transform_batch(self, x, rate, phase_rad=0.0, out=None, ordered=False):
        """pffft_hip_shift_transform_batch: x is ONE stream of batch*N complex samples; sample g is multiplied by
        exp(j (phase_rad + 2 pi rate g)) (src/pf_mixer.cpp:142-165) and every N samples are forward-transformed."""
        import torch
        assert self.transform_type == COMPLEX and self.dtype == np.float32
        batch = self._tcheck(x)
        if out is None:
            out = torch.empty_like(x)
        _check(self._L.pffft_hip_shift_transform_batch(self.handle, x.data_ptr(), out.data_ptr(), batch,
                                                       int(bool(ordered)), float(rate), float(phase_rad), self._stream()),
               "hip_shift_transform_batch")
        return out

    def zreorder_batch(self, x, out=None, direction=FORWARD):
        import torch
        batch = self._tcheck(x)
        if out is None:
            out = torch.empty_like(x)
        fn = getattr(self._L, f"{self._pfx}_hip_zreorder_batch")
        _check(fn(self.handle, x.data_ptr(), out.data_ptr(), batch, direction, self._stream()), "hip_zreorder_batch")
        return out

    def zconvolve_batch(self, a, b, ab, scaling, accumulate=True, b_broadcast=False):
        batch = self._tcheck(a)
        self._tcheck(ab)
        fn = getattr(self._L, f"{self._pfx}_hip_zconvolve_batch")
        _check(fn(self.handle, a.data_ptr(), b.data_ptr(), ab.data_ptr(), scaling, batch, int(bool(accumulate)),
                  int(bool(b_broadcast)), self._stream()), "hip_zconvolve_batch")
        return ab

    def convolve_batch(self, x, H, out=None, scaling=1.0, accumulate=False):
        """pffft_hip_convolve_batch: out (+)= backward(forward(x) . H) * scaling; H = ONE spectrum in the internal layout
        (1-D tensor, broadcast) or one per vector (same shape as x)."""
        import torch
        batch = self._tcheck(x)
        if out is None:
            assert not accumulate
            out = torch.empty_like(x)
        self._tcheck(out)
        bc = H.numel() == self.vec_scalars
        assert bc or H.numel() == x.numel()
        assert H.is_cuda and H.dtype == x.dtype and H.is_contiguous()
        fn = getattr(self._L, f"{self._pfx}_hip_convolve_batch")
        _check(fn(self.handle, x.data_ptr(), H.data_ptr(), out.data_ptr(), scaling, batch, int(bool(accumulate)), int(bc),
                  self._stream()), "hip_convolve_batch")
        return out

    # ---------------- host (numpy): the legacy single-vector entries ----------------
    def _legacy(self, name, x, direction):
        xin = _aligned(x, self.dtype)
        assert xin.size == self.vec_scalars
        out = _aligned_empty(self.vec_scalars, self.dtype)
        getattr(self._L, f"{self._pfx}_{name}")(self.handle, xin.ctypes.data, out.ctypes.data, None, direction)
        return out

    def transform(self, x, direction=FORWARD):
        if _is_torch(x):
            return self.transform_batch(x, None, direction, ordered=False)
        return self._legacy("transform", x, direction)

    def transform_ordered(self, x, direction=FORWARD):
        if _is_torch(x):
            return self.transform_batch(x, None, direction, ordered=True)
        return self._legacy("transform_ordered", x, direction)

    def transform_inplace(self, buf: np.ndarray, direction=FORWARD, ordered=False):
        """input and output alias (allowed: include/pffft/pffft.h:157)."""
        assert buf.dtype == self.dtype and buf.size == self.vec_scalars and buf.ctypes.data % 32 == 0
        name = "transform_ordered" if ordered else "transform"
        getattr(self._L, f"{self._pfx}_{name}")(self.handle, buf.ctypes.data, buf.ctypes.data, None, direction)
        return buf

    def zreorder(self, x, direction=FORWARD):
        if _is_torch(x):
            return self.zreorder_batch(x, None, direction)
        xin = _aligned(x, self.dtype)
        out = _aligned_empty(self.vec_scalars, self.dtype)
        getattr(self._L, f"{self._pfx}_zreorder")(self.handle, xin.ctypes.data, out.ctypes.data, direction)
        return out

    def zconvolve(self, a, b, ab, scaling, accumulate=True):
        if _is_torch(a):
            return self.zconvolve_batch(a, b, ab, scaling, accumulate)
        pa, pb = _aligned(a, self.dtype), _aligned(b, self.dtype)
        pab = _aligned_empty(self.vec_scalars, self.dtype)
        pab[:] = np.asarray(ab, dtype=self.dtype).ravel()
        name = "zconvolve_accumulate" if accumulate else "zconvolve_no_accu"
        getattr(self._L, f"{self._pfx}_{name}")(self.handle, pa.ctypes.data, pb.ctypes.data, pab.ctypes.data,
                                                 self.dtype.type(scaling))
        return pab


class FastConv:
    """PFFASTCONV_Setup (src/pffastconv.c:58-116); `block_len` is updated like *blockLen."""

    def __init__(self, taps, block_len: int = 0, flags: int = 0):
        self._L = lib()
        h = _aligned(taps, np.float32)
        self.filter_len = h.size
        bl = C.c_int(block_len)
        self.handle = self._L.pffastconv_new_setup(h.ctypes.data, h.size, C.byref(bl), flags)
        if not self.handle:
            raise ValueError("pffastconv_new_setup returned NULL")
        self.block_len, self.flags = bl.value, flags

    def close(self):
        if getattr(self, "handle", None):
            self._L.pffastconv_destroy_setup(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def apply(self, x, flush: bool = True, out=None):
        """Returns (y[:n_out], n_out) like pffastconv_apply (src/pffastconv.c:133-263)."""
        cpl = 2 if (self.flags & 1) else 1
        if _is_torch(x):
            import torch
            assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous()
            n_in = x.numel() // cpl
            y = out if out is not None else torch.empty_like(x)
            st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
            n = self._L.pffastconv_hip_apply_device(self.handle, x.data_ptr(), n_in, y.data_ptr(), int(bool(flush)), st)
            if n < 0:
                raise RuntimeError("pffastconv_hip_apply_device failed: " + self._L.pffft_hip_last_error().decode())
            return y[:n * cpl], n
        xin = _aligned(x, np.float32)
        n_in = xin.size // cpl
        y = _aligned_empty(max(xin.size, 1), np.float32)
        n = self._L.pffastconv_apply(self.handle, xin.ctypes.data, n_in, y.ctypes.data, int(bool(flush)))
        if n < 0:   # the drop-in's failure value (include/pffft_hip.h): the C entry has already failed soft, the mirror returns it as is
            return y[:0].copy(), n
        return y[:n * cpl].copy(), n

    def apply_batch(self, x, flush: bool = True, out=None):
        """pffastconv_hip_apply_batch: x is a [nsignals, floats-per-signal] CUDA tensor of independent signals; every row
        is filtered exactly as apply() would.  Returns (y[:, :n_out*cpl], n_out); n_out is per signal."""
        import torch
        cpl = 2 if (self.flags & 1) else 1
        assert x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1
        nsig, fl = x.shape
        y = out if out is not None else torch.empty_like(x)
        assert y.dim() == 2 and y.shape[0] == nsig and y.stride(1) == 1
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        n = self._L.pffastconv_hip_apply_batch(self.handle, x.data_ptr(), fl // cpl, x.stride(0) if nsig > 1 else fl,
                                               y.data_ptr(), y.stride(0) if nsig > 1 else fl, nsig, int(bool(flush)), st)
        if n < 0:
            raise RuntimeError("pffastconv_hip_apply_batch failed: " + self._L.pffft_hip_last_error().decode())
        return y[:, :n * cpl], n
