"""ctypes binding to oracle/_ref/libpffft_ref.so — the REAL reference (marton78/pffft)
compiled from its own sources by oracle/Makefile.

TEST INFRASTRUCTURE ONLY.  Imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py as the checker / timed CPU baseline.  The product
(pffft_amd, libpffft_hip.so) never imports, links or executes anything in oracle/.

The symbols bound here are the reference's public C API:
  include/pffft/pffft.h:124-250, include/pffft/pffft_double.h, include/pffft/pffastconv.h:145-180.
The library is loaded through its own ctypes.CDLL handle (RTLD_LOCAL) because the
drop-in exports the very same symbol names (SURVEY.md finding 10).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
REF_SO = os.path.join(_HERE, "_ref", "libpffft_ref.so")
FFTPACK_SO = os.path.join(_HERE, "_ref", "libfftpack_ref.so")
# BASELINE configs[0] builds of the same sources (oracle/Makefile): PFFFT_USE_SIMD=OFF = "4xScalar" vectors, and the
# SIMD_SZ == 1 build (PFFFT_USE_SCALAR_VECT=OFF too) whose unordered output is FFTPACK's order
REF_4XSCALAR_SO = os.path.join(_HERE, "_ref", "libpffft_ref_4xscalar.so")
REF_SCALAR_SO = os.path.join(_HERE, "_ref", "libpffft_ref_scalar.so")
REFERENCE_ROOT = os.environ.get("PFFFT_REFERENCE_ROOT", "/root/reference")

FORWARD, BACKWARD = 0, 1
REAL, COMPLEX = 0, 1


def build(force: bool = False) -> bool:
    """Compile oracle/_ref from /root/reference when the sources are present.
    Returns True when the .so exists afterwards.  On the GPU box /root/reference is
    absent and the prebuilt object shipped with the snapshot is used as is."""
    want = [REF_SO, os.path.join(_HERE, "_ref", "libpfdsp_ref.so"), REF_4XSCALAR_SO, REF_SCALAR_SO,
            os.path.join(_HERE, "_ref", "libcpubase.so")]
    stale = any(not os.path.exists(f) for f in want) or (
        os.path.getmtime(os.path.join(_HERE, "cpu_baseline.c")) > os.path.getmtime(want[-1]))
    if os.path.isdir(os.path.join(REFERENCE_ROOT, "src")) and (force or stale):
        subprocess.run(["make", "-C", _HERE, f"REF={REFERENCE_ROOT}"], check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return os.path.exists(REF_SO)


def available() -> bool:
    return os.path.exists(REF_SO)


class _Api:
    """One precision of the reference API (prefix 'pffft' float / 'pffftd' double)."""

    def __init__(self, lib, prefix: str, ctype, dtype):
        self.lib, self.prefix, self.ctype, self.dtype = lib, prefix, ctype, np.dtype(dtype)
        P = C.POINTER(ctype)
        f = lambda name: getattr(lib, f"{prefix}_{name}")
        self.new_setup = f("new_setup"); self.new_setup.restype = C.c_void_p
        self.new_setup.argtypes = [C.c_int, C.c_int]
        self.destroy_setup = f("destroy_setup"); self.destroy_setup.restype = None
        self.destroy_setup.argtypes = [C.c_void_p]
        for name in ("transform", "transform_ordered"):
            fn = f(name); fn.restype = None
            fn.argtypes = [C.c_void_p, P, P, P, C.c_int]
            setattr(self, name, fn)
        self.zreorder = f("zreorder"); self.zreorder.restype = None
        self.zreorder.argtypes = [C.c_void_p, P, P, C.c_int]
        for name in ("zconvolve_accumulate", "zconvolve_no_accu"):
            fn = f(name); fn.restype = None
            fn.argtypes = [C.c_void_p, P, P, P, ctype]
            setattr(self, name, fn)
        self.simd_size = f("simd_size"); self.simd_size.restype = C.c_int; self.simd_size.argtypes = []
        self.simd_arch = f("simd_arch"); self.simd_arch.restype = C.c_char_p; self.simd_arch.argtypes = []
        self.min_fft_size = f("min_fft_size"); self.min_fft_size.restype = C.c_int
        self.min_fft_size.argtypes = [C.c_int]
        self.is_valid_size = f("is_valid_size"); self.is_valid_size.restype = C.c_int
        self.is_valid_size.argtypes = [C.c_int, C.c_int]
        self.nearest_transform_size = f("nearest_transform_size")
        self.nearest_transform_size.restype = C.c_int
        self.nearest_transform_size.argtypes = [C.c_int, C.c_int, C.c_int]
        self.next_power_of_two = f("next_power_of_two"); self.next_power_of_two.restype = C.c_int
        self.next_power_of_two.argtypes = [C.c_int]
        self.is_power_of_two = f("is_power_of_two"); self.is_power_of_two.restype = C.c_int
        self.is_power_of_two.argtypes = [C.c_int]
        self.aligned_malloc = f("aligned_malloc"); self.aligned_malloc.restype = C.c_void_p
        self.aligned_malloc.argtypes = [C.c_size_t]
        self.aligned_free = f("aligned_free"); self.aligned_free.restype = None
        self.aligned_free.argtypes = [C.c_void_p]

    # ---- numpy conveniences (64-byte aligned buffers as the reference demands) ----
    def empty(self, n: int) -> np.ndarray:
        raw = np.empty(n * self.dtype.itemsize + 64, dtype=np.uint8)
        off = (-raw.ctypes.data) % 64
        return raw[off:off + n * self.dtype.itemsize].view(self.dtype)

    def aligned(self, a) -> np.ndarray:
        a = np.asarray(a, dtype=self.dtype).ravel()
        out = self.empty(a.size)
        out[:] = a
        return out

    def ptr(self, a: np.ndarray):
        assert a.dtype == self.dtype and a.ctypes.data % 32 == 0
        return a.ctypes.data_as(C.POINTER(self.ctype))


class Setup:
    """RAII wrapper around PFFFT_Setup* of the reference (src/pffft_priv_impl.h:1051-1120)."""

    def __init__(self, api: _Api, N: int, transform: int):
        self.api, self.N, self.transform = api, N, transform
        self.h = api.new_setup(N, transform)
        if not self.h:
            raise ValueError(f"reference rejected N={N} transform={transform}")
        self.nfloats = N * (2 if transform == COMPLEX else 1)

    def close(self):
        if self.h:
            self.api.destroy_setup(self.h)
            self.h = None

    __del__ = close

    def _run(self, fn, x, direction):
        a = self.api
        xin = a.aligned(x)
        assert xin.size == self.nfloats, (xin.size, self.nfloats)
        out = a.empty(self.nfloats)
        work = a.empty(self.nfloats)
        fn(self.h, a.ptr(xin), a.ptr(out), a.ptr(work), direction)
        return out.copy()

    def transform_unordered(self, x, direction=FORWARD):
        return self._run(self.api.transform, x, direction)

    def transform_ordered(self, x, direction=FORWARD):
        return self._run(self.api.transform_ordered, x, direction)

    def zreorder(self, x, direction=FORWARD):
        a = self.api
        xin = a.aligned(x)
        out = a.empty(self.nfloats)
        a.zreorder(self.h, a.ptr(xin), a.ptr(out), direction)
        return out.copy()

    def zconvolve(self, fa, fb, fab, scaling, accumulate: bool):
        a = self.api
        pa, pb, pab = a.aligned(fa), a.aligned(fb), a.aligned(fab)
        fn = a.zconvolve_accumulate if accumulate else a.zconvolve_no_accu
        fn(self.h, a.ptr(pa), a.ptr(pb), a.ptr(pab), a.ctype(scaling))
        return pab.copy()

    def batch(self, x2d: np.ndarray, direction=FORWARD, ordered=False) -> np.ndarray:
        """Loop of single-vector calls — the reference has no batch entry (SURVEY finding 1)."""
        a = self.api
        x2d = np.ascontiguousarray(x2d, dtype=a.dtype).reshape(-1, self.nfloats)
        fn = a.transform_ordered if ordered else a.transform
        xin, out, work = a.empty(self.nfloats), a.empty(self.nfloats), a.empty(self.nfloats)
        res = np.empty_like(x2d)
        for i in range(x2d.shape[0]):
            xin[:] = x2d[i]
            fn(self.h, a.ptr(xin), a.ptr(out), a.ptr(work), direction)
            res[i] = out
        return res


class Reference:
    def __init__(self, path: str = REF_SO):
        if not os.path.exists(path):
            raise FileNotFoundError(
                f"{path} missing: run `make -C oracle` (needs /root/reference) first")
        self.lib = C.CDLL(path, mode=getattr(os, "RTLD_LOCAL", 0))
        self.f32 = _Api(self.lib, "pffft", C.c_float, np.float32)
        self.f64 = _Api(self.lib, "pffftd", C.c_double, np.float64)
        L = self.lib
        L.pffastconv_new_setup.restype = C.c_void_p
        L.pffastconv_new_setup.argtypes = [C.POINTER(C.c_float), C.c_int, C.POINTER(C.c_int), C.c_int]
        L.pffastconv_destroy_setup.restype = None
        L.pffastconv_destroy_setup.argtypes = [C.c_void_p]
        L.pffastconv_apply.restype = C.c_int
        L.pffastconv_apply.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_int,
                                       C.POINTER(C.c_float), C.c_int]

    def api(self, dtype) -> _Api:
        return self.f64 if np.dtype(dtype) == np.float64 else self.f32

    def setup(self, N, transform, dtype=np.float32) -> Setup:
        return Setup(self.api(dtype), N, transform)

    def fastconv(self, x, h, block_len=0, flags=0, flush=1):
        """pffastconv_new_setup + pffastconv_apply over the whole signal
        (src/pffastconv.c:58-116,133-263).  Returns (y[:n_out], n_out, block_len_used)."""
        a = self.f32
        hh = a.aligned(h)
        bl = C.c_int(block_len)
        s = self.lib.pffastconv_new_setup(a.ptr(hh), len(h), C.byref(bl), flags)
        if not s:
            raise ValueError("pffastconv_new_setup returned NULL")
        try:
            cplx = 2 if (flags & 1) else 1
            xx = a.aligned(x)
            n_in = xx.size // cplx
            y = a.empty(max(xx.size, 1))
            y[:] = 0
            n_out = self.lib.pffastconv_apply(s, a.ptr(xx), n_in, a.ptr(y), flush)
            return y[:n_out * cplx].copy(), n_out, bl.value
        finally:
            self.lib.pffastconv_destroy_setup(s)


_REF = None


def get() -> Reference:
    global _REF
    if _REF is None:
        _REF = Reference()
    return _REF
