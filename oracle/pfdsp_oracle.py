"""numpy restatement of the reference's frequency-shift mixers (src/pf_mixer.cpp, marton78/pffft).

TEST INFRASTRUCTURE ONLY.  Two things live here:

* `exact(...)`: the function all ten reference algorithms approximate,
  out[i] = in[i] * exp(j (phase0 + (i + first) * inc)), evaluated in float64 — the ground truth both
  the reference and the HIP kernels are held against;
* the reference's own recurrences, restated operation by operation in float32 (np.float32 scalars,
  no FMA, same evaluation order), each citing the lines it follows.  They are sequential Python loops:
  use them at n <= a few thousand.  Pinned against the compiled reference (oracle/_ref/libpfdsp_ref.so)
  by tests/test_pfdsp_oracle.py — exact up to the last bit of sinf/cosf (numpy vs glibc).

Parity pinned: yes (against the reference's own object code, same inputs).
"""
from __future__ import annotations

import math

import numpy as np

f32 = np.float32
PI = f32(3.14159265358979323846)          # src/pf_mixer.cpp:40  #define PI ((float)3.14159...)
TWO_PI = f32(2) * PI


def DRIFT(n):
    """Measured bound on |reference - exact| after n samples of |x| <= sqrt(2) (tests/test_pfdsp_oracle.py)."""
    return 1e-6 + 2e-7 * n


def RETURN_BAR(n, inc):
    """Bound on the error of the phase algorithms C and D RETURN after n samples: they form it as the float product n*inc
    and wrap it with a float 2*pi, one rounded subtraction per turn (src/pf_mixer.cpp:280-283) — a chained call starts
    from that value, so its whole block is off by as much."""
    tot = abs(n * float(inc))
    return 1e-6 + (2 + tot / (2 * math.pi)) * float(np.spacing(np.float32(tot)))


def _sinf(x):
    return f32(math.sin(float(x)))


def _cosf(x):
    return f32(math.cos(float(x)))


def _wrap_pm_pi(p):                        # while(p>PI) p-=2*PI; while(p<-PI) p+=2*PI;   (e.g. :281-283)
    while p > PI:
        p = f32(p - TWO_PI)
    while p < -PI:
        p = f32(p + TWO_PI)
    return p


def _cmul_into(out, k, x, c, s):           # out = (c + j s) * x, the way every variant writes it (:157-158)
    xi, xq = f32(x.real), f32(x.imag)
    out[k] = complex(f32(f32(c * xi) - f32(s * xq)), f32(f32(s * xi) + f32(c * xq)))


def exact(x, inc, phase0=0.0, first=0):
    """float64 ground truth; `inc` is the per-sample increment in radians AS THE ALGORITHM HOLDS IT (a float32)."""
    x = np.asarray(x)
    i = np.arange(x.size, dtype=np.float64) + first
    ph = float(phase0) + i * float(inc)
    return x.astype(np.complex128) * np.exp(1j * ph)


def increment(rate, recursive=False):
    """The float32 per-sample increment the reference forms: 2*rate*PI (:146-147, :235, :336, :416, :524) —
    or rate*PI for the recursive oscillators (:901, :1046)."""
    r = f32(rate)
    return f32(r * PI) if recursive else f32(f32(f32(2) * r) * PI)


# ---- A: shift_math_cc (:142-165) ----
def shift_math_cc(x, rate, starting_phase):
    inc = increment(rate)
    out = np.empty(x.size, np.complex64)
    phase = f32(starting_phase)
    for i in range(x.size):
        _cmul_into(out, i, x[i], _cosf(phase), _sinf(phase))
        phase = f32(phase + inc)
        while phase > TWO_PI:
            phase = f32(phase - TWO_PI)
        while phase < 0:
            phase = f32(phase + TWO_PI)
    return out, phase


# ---- C: shift_addfast_cc (:232-286) ----
def shift_addfast_cc(x, rate, starting_phase):
    inc = increment(rate)
    ds = [_sinf(f32(inc * f32(i + 1))) for i in range(4)]
    dc = [_cosf(f32(inc * f32(i + 1))) for i in range(4)]
    out = np.array(x, np.complex64)
    cs, ss = _cosf(f32(starting_phase)), _sinf(f32(starting_phase))
    for g in range(x.size // 4):
        cv = [f32(f32(cs * dc[j]) - f32(ss * ds[j])) for j in range(4)]     # SADF_L1 (:244-246)
        sv = [f32(f32(ss * dc[j]) + f32(cs * ds[j])) for j in range(4)]
        for j in range(4):
            _cmul_into(out, 4 * g + j, x[4 * g + j], cv[j], sv[j])          # SADF_L2 (:247-249)
        cs, ss = cv[3], sv[3]
    ph = f32(f32(starting_phase) + f32(f32(x.size) * inc))                  # :280
    return out, _wrap_pm_pi(ph)


def _unroll_table(inc, size):              # :339-347, :418-426
    dc, ds = np.empty(size, f32), np.empty(size, f32)
    my = f32(0)
    for i in range(size):
        my = _wrap_pm_pi(f32(my + inc))
        ds[i], dc[i] = _sinf(my), _cosf(my)
    return dc, ds


# ---- D: shift_unroll_cc (:333-381) ----
def shift_unroll_cc(x, rate, starting_phase):
    inc = increment(rate)
    dc, ds = _unroll_table(inc, x.size)
    out = np.empty(x.size, np.complex64)
    cs, ss = _cosf(f32(starting_phase)), _sinf(f32(starting_phase))
    cv, sv = cs, ss
    for i in range(x.size):
        _cmul_into(out, i, x[i], cv, sv)
        cv = f32(f32(cs * dc[i]) - f32(ss * ds[i]))
        sv = f32(f32(ss * dc[i]) + f32(cs * ds[i]))
    ph = f32(f32(starting_phase) + f32(f32(x.size) * inc))
    return out, _wrap_pm_pi(ph)


# ---- E: shift_limited_unroll_cc (:413-464); state = (cos, sin) phasor ----
def shift_limited_unroll_cc(x, rate, state=(1.0, 0.0)):
    inc = increment(rate)
    dc, ds = _unroll_table(inc, 128)
    out = np.array(x, np.complex64)
    cs, ss = f32(state[0]), f32(state[1])
    cv, sv = cs, ss
    pos, size = 0, x.size
    while size > 0:
        n = 128 if size >= 128 else size
        for i in range(n // 4 * 4):
            _cmul_into(out, pos + i, x[pos + i], cv, sv)
            cv = f32(f32(cs * dc[i]) - f32(ss * ds[i]))
            sv = f32(f32(ss * dc[i]) + f32(cs * ds[i]))
        mag = f32(np.sqrt(f32(f32(cv * cv) + f32(sv * sv))))
        cv, sv = f32(cv / mag), f32(sv / mag)
        cs, ss = cv, sv
        pos += 128
        size -= 128
    return out, (cv, sv)


# ---- F (= G = H up to the table layout): shift_limited_unroll_A_sse_inp_c (:519-615) ----
def shift_limited_unroll_sse(x, rate, phase_start):
    inc = increment(rate)
    ng = (128 + 4) // 4
    tc, ts = np.empty(ng, f32), np.empty(ng, f32)
    my = f32(0)
    for g in range(ng):                       # :527-541: one entry per 4 increments
        for _ in range(4):
            my = _wrap_pm_pi(f32(my + inc))
        tc[g], ts[g] = _cosf(my), _sinf(my)
    st_c, st_s = np.empty(4, f32), np.empty(4, f32)
    my = f32(phase_start)
    for k in range(4):                        # :546-554
        st_c[k], st_s[k] = _cosf(my), _sinf(my)
        my = _wrap_pm_pi(f32(my + inc))
    out = np.array(x, np.complex64)
    cv, sv = st_c.copy(), st_s.copy()
    pos, left = 0, x.size
    while left:
        nb = 128 if left >= 128 else left
        for g in range(nb // 4):
            for k in range(4):
                _cmul_into(out, pos + 4 * g + k, x[pos + 4 * g + k], cv[k], sv[k])
            cv = (tc[g] * st_c - ts[g] * st_s).astype(f32)     # "vals := d[] * starts" (:593-598); f32 array ops round per op
            sv = (ts[g] * st_c + tc[g] * st_s).astype(f32)
        left -= nb
        pos += nb
        mag = np.sqrt((cv * cv + sv * sv).astype(f32)).astype(f32)
        cv, sv = (cv / mag).astype(f32), (sv / mag).astype(f32)
        st_c, st_s = cv.copy(), sv.copy()
    return out, (st_c, st_s)


# ---- I / J: recursive quadrature oscillator (:898-1030, :1043-1126) ----
def recursive_osc_init(rate, starting_phase, lanes):
    u, v = np.empty(lanes, f32), np.empty(lanes, f32)
    if f32(starting_phase) != 0:
        u[0], v[0] = _cosf(f32(starting_phase)), _sinf(f32(starting_phase))
    else:
        u[0], v[0] = f32(1), f32(0)
    inc_s = increment(rate, recursive=True)
    k1 = f32(math.tan(float(f32(f32(0.5) * inc_s))))
    k2 = f32(f32(f32(2) * k1) / f32(f32(1) + f32(k1 * k1)))
    for j in range(1, lanes):                 # :904-913
        u[j], v[j] = u[j - 1], v[j - 1]
        tmp = f32(u[j] - f32(k1 * v[j]))
        v[j] = f32(v[j] + f32(k2 * tmp))
        u[j] = f32(tmp - f32(k1 * v[j]))
    inc_b = _wrap_pm_pi(f32(inc_s * f32(lanes)))
    K1 = f32(math.tan(float(f32(f32(0.5) * inc_b))))
    K2 = f32(f32(f32(2) * K1) / f32(f32(1) + f32(K1 * K1)))
    return (K1, K2), (u, v)


def recursive_osc_run(x, conf, state, lanes, gen=False):
    k1, k2 = conf
    u, v = state[0].copy(), state[1].copy()
    out = np.array(x, np.complex64)
    for b in range(x.size // lanes):
        for j in range(lanes):
            if gen:
                out[lanes * b + j] = complex(u[j], v[j])
            else:
                _cmul_into(out, lanes * b + j, x[lanes * b + j], u[j], v[j])
        tmp = (u - (k1 * v).astype(f32)).astype(f32)           # :962-967
        v = (v + (k2 * tmp).astype(f32)).astype(f32)
        u = (tmp - (k1 * v).astype(f32)).astype(f32)
    return out, (u, v)


def osc_step_angle(k1, k2):
    """Rotation angle per block of the recurrence above for the GIVEN float constants: the update is a product of
    three shears with trace 2 - 2 k1 k2 = 2 cos(theta)."""
    h = min(max(0.5 * float(k1) * float(k2), 0.0), 1.0)
    th = 2.0 * math.asin(math.sqrt(h))
    return -th if k1 < 0 else th
