"""ctypes binding of the PFDSP mixer ABI (include/pfdsp_hip.h = the reference's include/pffft/pf_mixer.h:61-280).

`MixerABI(path)` binds every reference-named entry of ANY shared object with that ABI — the product
(`libpfdsp_hip.so`, via `lib()`) and, in tests only, the compiled reference (oracle/pfdsp_ref.py hands it
oracle/_ref/libpfdsp_ref.so).  The struct classes below are the ABI's by-value state blocks.

`Mixer` is the small operator-style mirror the tests use:  m = Mixer("addfast", rate); y = m(x)  keeps the
algorithm's state between calls exactly as a C caller would (returned phase fed back / struct advanced).
numpy complex64 arrays go in as host pointers; torch CUDA complex64 tensors as device pointers.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
UNROLL, SIMD = 128, 4      # PF_SHIFT_LIMITED_UNROLL_SIZE, PF_SHIFT_LIMITED_SIMD_SZ (pf_mixer.h:135-136)


class complexf(C.Structure):
    _fields_ = [("i", C.c_float), ("q", C.c_float)]


class shift_table_data_t(C.Structure):
    _fields_ = [("table", C.POINTER(C.c_float)), ("table_size", C.c_int)]


class shift_addfast_data_t(C.Structure):
    _fields_ = [("dsin", C.c_float * 4), ("dcos", C.c_float * 4), ("phase_increment", C.c_float)]


class shift_unroll_data_t(C.Structure):
    _fields_ = [("dsin", C.POINTER(C.c_float)), ("dcos", C.POINTER(C.c_float)), ("phase_increment", C.c_float),
                ("size", C.c_int)]


class shift_limited_unroll_data_t(C.Structure):
    _fields_ = [("dcos", C.c_float * UNROLL), ("dsin", C.c_float * UNROLL), ("complex_phase", complexf),
                ("phase_increment", C.c_float)]


class shift_limited_unroll_A_sse_data_t(C.Structure):
    _fields_ = [("dcos", C.c_float * (UNROLL + SIMD)), ("dsin", C.c_float * (UNROLL + SIMD)),
                ("phase_state_i", C.c_float * SIMD), ("phase_state_q", C.c_float * SIMD),
                ("dcos_blk", C.c_float), ("dsin_blk", C.c_float), ("phase_increment", C.c_float)]


class shift_limited_unroll_B_sse_data_t(C.Structure):
    _fields_ = [("dtrig", C.c_float * (UNROLL + SIMD)),
                ("phase_state_i", C.c_float * SIMD), ("phase_state_q", C.c_float * SIMD),
                ("dcos_blk", C.c_float), ("dsin_blk", C.c_float), ("phase_increment", C.c_float)]


class shift_limited_unroll_C_sse_data_t(C.Structure):
    _fields_ = [("dinterl_trig", C.c_float * (2 * (UNROLL + SIMD))),
                ("phase_state_i", C.c_float * SIMD), ("phase_state_q", C.c_float * SIMD),
                ("dcos_blk", C.c_float), ("dsin_blk", C.c_float), ("phase_increment", C.c_float)]


class shift_recursive_osc_t(C.Structure):
    _fields_ = [("u_cos", C.c_float * 8), ("v_sin", C.c_float * 8)]


class shift_recursive_osc_conf_t(C.Structure):
    _fields_ = [("k1", C.c_float), ("k2", C.c_float)]


class shift_recursive_osc_sse_t(C.Structure):
    _fields_ = [("u_cos", C.c_float * 4), ("v_sin", C.c_float * 4)]


class shift_recursive_osc_sse_conf_t(C.Structure):
    _fields_ = [("k1", C.c_float), ("k2", C.c_float)]


_SSE_DATA = {"A": shift_limited_unroll_A_sse_data_t, "B": shift_limited_unroll_B_sse_data_t,
             "C": shift_limited_unroll_C_sse_data_t}

# every reference-named entry: name -> (restype, argtypes)
_P, _F, _I = C.c_void_p, C.c_float, C.c_int
REFERENCE_ENTRIES = {
    "have_sse_shift_mixer_impl": (_I, []),
    "shift_math_cc": (_F, [_P, _P, _I, _F, _F]),
    "shift_table_init": (shift_table_data_t, [_I]),
    "shift_table_deinit": (None, [shift_table_data_t]),
    "shift_table_cc": (_F, [_P, _P, _I, _F, shift_table_data_t, _F]),
    "shift_addfast_init": (shift_addfast_data_t, [_F]),
    "shift_addfast_cc": (_F, [_P, _P, _I, C.POINTER(shift_addfast_data_t), _F]),
    "shift_addfast_inp_c": (_F, [_P, _I, C.POINTER(shift_addfast_data_t), _F]),
    "shift_unroll_init": (shift_unroll_data_t, [_F, _I]),
    "shift_unroll_deinit": (None, [C.POINTER(shift_unroll_data_t)]),
    "shift_unroll_cc": (_F, [_P, _P, _I, C.POINTER(shift_unroll_data_t), _F]),
    "shift_unroll_inp_c": (_F, [_P, _I, C.POINTER(shift_unroll_data_t), _F]),
    "shift_limited_unroll_init": (shift_limited_unroll_data_t, [_F]),
    "shift_limited_unroll_cc": (None, [_P, _P, _I, C.POINTER(shift_limited_unroll_data_t)]),
    "shift_limited_unroll_inp_c": (None, [_P, _I, C.POINTER(shift_limited_unroll_data_t)]),
    "shift_recursive_osc_init": (None, [_F, _F, C.POINTER(shift_recursive_osc_conf_t), C.POINTER(shift_recursive_osc_t)]),
    "shift_recursive_osc_update_rate": (None, [_F, C.POINTER(shift_recursive_osc_conf_t), C.POINTER(shift_recursive_osc_t)]),
    "shift_recursive_osc_cc": (None, [_P, _P, _I, C.POINTER(shift_recursive_osc_conf_t), C.POINTER(shift_recursive_osc_t)]),
    "shift_recursive_osc_inp_c": (None, [_P, _I, C.POINTER(shift_recursive_osc_conf_t), C.POINTER(shift_recursive_osc_t)]),
    "gen_recursive_osc_c": (None, [_P, _I, C.POINTER(shift_recursive_osc_conf_t), C.POINTER(shift_recursive_osc_t)]),
    "shift_recursive_osc_sse_init": (None, [_F, _F, C.POINTER(shift_recursive_osc_sse_conf_t), C.POINTER(shift_recursive_osc_sse_t)]),
    "shift_recursive_osc_sse_update_rate": (None, [_F, C.POINTER(shift_recursive_osc_sse_conf_t), C.POINTER(shift_recursive_osc_sse_t)]),
    "shift_recursive_osc_sse_inp_c": (None, [_P, _I, C.POINTER(shift_recursive_osc_sse_conf_t), C.POINTER(shift_recursive_osc_sse_t)]),
}
for _k, _T in _SSE_DATA.items():
    REFERENCE_ENTRIES[f"shift_limited_unroll_{_k}_sse_init"] = (_T, [_F, _F])
    REFERENCE_ENTRIES[f"shift_limited_unroll_{_k}_sse_inp_c"] = (None, [_P, _I, C.POINTER(_T)])


class MixerABI:
    """All reference-named mixer entries of one shared object, typed."""

    def __init__(self, path: str):
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing — build it with `python -m pffft_amd.build`.  There is no CPU fallback.")
        self.path = path
        self.dll = C.CDLL(path, mode=getattr(os, "RTLD_LOCAL", 0))
        for name, (res, args) in REFERENCE_ENTRIES.items():
            fn = getattr(self.dll, name)
            fn.restype, fn.argtypes = res, args
            setattr(self, name, fn)


_LIB = None


def lib_path() -> str:
    return os.path.join(_HERE, "libpfdsp_hip.so")


def lib() -> MixerABI:
    """The product library (HIP).  torch first, so that both share one HIP runtime (see api.lib)."""
    global _LIB
    if _LIB is None:
        try:
            import torch  # noqa: F401
        except Exception:
            pass
        L = MixerABI(lib_path())
        L.dll.pfdsp_hip_shift_device.restype = C.c_int
        L.dll.pfdsp_hip_shift_device.argtypes = [_P, _P, C.c_size_t, C.c_double, C.c_double, _P]
        L.dll.pfdsp_hip_last_error.restype = C.c_char_p
        _LIB = L
    return _LIB


def _is_torch(x) -> bool:
    return type(x).__module__.startswith("torch")


def _ptr(x):
    if _is_torch(x):
        import torch
        assert x.is_cuda and x.dtype == torch.complex64 and x.is_contiguous()
        return x.data_ptr(), x.numel()
    assert isinstance(x, np.ndarray) and x.dtype == np.complex64 and x.flags.c_contiguous
    return x.ctypes.data, x.size


def _like(x):
    if _is_torch(x):
        import torch
        return torch.empty_like(x)
    return np.empty_like(x)


def shift_device(x, rate: float, phase_rad: float = 0.0, out=None):
    """pfdsp_hip_shift_device on torch CUDA complex64 tensors (current stream); x=None + out: oscillator only."""
    import torch
    L = lib()
    if out is None:
        out = torch.empty_like(x)
    po, n = _ptr(out)
    pi = _ptr(x)[0] if x is not None else None
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    rc = L.dll.pfdsp_hip_shift_device(pi, po, n, float(rate), float(phase_rad), st)
    if rc:
        raise RuntimeError(f"pfdsp_hip_shift_device failed ({rc}): {L.dll.pfdsp_hip_last_error().decode()}")
    return out


ALGOS = ("math", "table", "addfast", "unroll", "limited_unroll", "limited_unroll_A_sse", "limited_unroll_B_sse",
         "limited_unroll_C_sse", "recursive_osc", "recursive_osc_sse")


class Mixer:
    """One reference mixer algorithm with its state, over a MixerABI (default: the HIP product).

    __call__(x, inplace=False) shifts the next len(x) samples of the stream and advances the state the way a C
    caller of that algorithm would.  `abi` lets the tests run the very same driver over the compiled reference."""

    def __init__(self, algo: str, rate: float, phase: float = 0.0, abi: MixerABI | None = None, table_size: int = 65536,
                 unroll_size: int = 0):
        assert algo in ALGOS, algo
        self.L = abi if abi is not None else lib()
        self.algo, self.rate, self.phase = algo, float(rate), float(phase)
        L = self.L
        self.data = self.conf = None
        if algo == "table":
            self.data = L.shift_table_init(table_size)
        elif algo == "addfast":
            self.data = L.shift_addfast_init(rate)
        elif algo == "unroll":
            self.unroll_size = unroll_size
            self.data = L.shift_unroll_init(rate, unroll_size) if unroll_size else None
        elif algo == "limited_unroll":
            self.data = L.shift_limited_unroll_init(rate)
            if phase:   # the reference's init starts at phase 0 (:427-428); a caller presets the phasor
                self.data.complex_phase.i, self.data.complex_phase.q = np.float32(np.cos(phase)), np.float32(np.sin(phase))
        elif algo.endswith("_sse") and algo.startswith("limited"):
            self.data = getattr(L, f"shift_{algo}_init")(rate, phase)
        elif algo == "recursive_osc":
            self.conf, self.data = shift_recursive_osc_conf_t(), shift_recursive_osc_t()
            L.shift_recursive_osc_init(rate, phase, C.byref(self.conf), C.byref(self.data))
        elif algo == "recursive_osc_sse":
            self.conf, self.data = shift_recursive_osc_sse_conf_t(), shift_recursive_osc_sse_t()
            L.shift_recursive_osc_sse_init(rate, phase, C.byref(self.conf), C.byref(self.data))

    def close(self):
        if self.algo == "table" and self.data is not None:
            self.L.shift_table_deinit(self.data)
        if self.algo == "unroll" and self.data is not None:
            self.L.shift_unroll_deinit(C.byref(self.data))
        self.data = None

    def __call__(self, x, inplace: bool = False):
        L, a = self.L, self.algo
        pi, n = _ptr(x)
        has_cc = a in ("math", "table", "addfast", "unroll", "limited_unroll", "recursive_osc")
        if inplace or not has_cc:
            if not inplace:          # only an in-place entry exists: work on a copy
                y = x.clone() if _is_torch(x) else x.copy()
            else:
                y = x
            po = _ptr(y)[0]
        else:
            y = _like(x)
            po = _ptr(y)[0]
        if a == "math":
            assert not inplace
            self.phase = L.shift_math_cc(pi, po, n, self.rate, self.phase)
        elif a == "table":
            assert not inplace
            self.phase = L.shift_table_cc(pi, po, n, self.rate, self.data, self.phase)
        elif a == "addfast":
            self.phase = (L.shift_addfast_inp_c(po, n, C.byref(self.data), self.phase) if inplace
                          else L.shift_addfast_cc(pi, po, n, C.byref(self.data), self.phase))
        elif a == "unroll":
            if self.data is None or self.data.size < n:      # the table must cover the call (:363-376)
                if self.data is not None:
                    L.shift_unroll_deinit(C.byref(self.data))
                self.data = L.shift_unroll_init(self.rate, n)
            self.phase = (L.shift_unroll_inp_c(po, n, C.byref(self.data), self.phase) if inplace
                          else L.shift_unroll_cc(pi, po, n, C.byref(self.data), self.phase))
        elif a == "limited_unroll":
            (L.shift_limited_unroll_inp_c(po, n, C.byref(self.data)) if inplace
             else L.shift_limited_unroll_cc(pi, po, n, C.byref(self.data)))
        elif a.startswith("limited_unroll_"):
            getattr(L, f"shift_{a}_inp_c")(po, n, C.byref(self.data))
        elif a == "recursive_osc":
            (L.shift_recursive_osc_inp_c(po, n, C.byref(self.conf), C.byref(self.data)) if inplace
             else L.shift_recursive_osc_cc(pi, po, n, C.byref(self.conf), C.byref(self.data)))
        elif a == "recursive_osc_sse":
            L.shift_recursive_osc_sse_inp_c(po, n, C.byref(self.conf), C.byref(self.data))
        return y

    def generate(self, out):
        """gen_recursive_osc_c: writes the oscillator itself (src/pf_mixer.cpp:1008-1030)."""
        assert self.algo == "recursive_osc"
        po, n = _ptr(out)
        self.L.gen_recursive_osc_c(po, n, C.byref(self.conf), C.byref(self.data))
        return out
