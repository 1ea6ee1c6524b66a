"""numpy restatement of the reference pffft algorithm (marton78/pffft) for the hot path.

TEST INFRASTRUCTURE ONLY — this module is the parity checker's second leg (the first is the real
reference compiled into oracle/_ref by oracle/Makefile).  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import it; the product (pffft_amd / libpffft_hip.so) never does.

Pinning: tests/test_oracle.py checks every function here against (a) oracle/_ref, the reference's own
code, when it is present, and (b) the committed fixtures tests/golden/*.npz that were generated from
oracle/_ref by tests/golden/make_golden.py.  The reference ships no stored golden vectors
(SURVEY.md §4); its generative checks (single-tone spectra, FFTPACK cross-check, naive FIR) are
restated in tests/ as well.

What is restated, with the reference lines each function follows (paths relative to the reference
root, SIMD_SZ == 4 build):
  size helpers          src/pffft_priv_impl.h:78-114, src/pffft_common.c:25-43
  decompose             src/pffft_priv_impl.h:904-928          (factor order, the "2 goes first" rule)
  cfft_lane             cfftf1_ps :1004-1048 + passf2/3/4/5_ps :122-321 + cffti1_ps :965-1001
                        (pass structure and twiddle generation in working precision; the radix-r
                        butterfly is written as a small DFT matrix instead of the hand-scheduled adds)
  cplx_forward/backward pffft_transform_internal :1465-1532 with pffft_cplx_finalize :1195-1237 /
                        pffft_cplx_preprocess :1239-1270 and the e[] table of pffft_new_setup :1089-1097
  real_forward/backward rfftf1_ps/rfftb1_ps :809-901 + pffft_real_finalize :1330-1372 /
                        pffft_real_preprocess :1423-1462 — restated at the level of what they compute:
                        four lane transforms of length N/4 combined by a radix-4 step.  The lane real
                        transform is evaluated with the complex passes on a real sequence (the
                        half-complex in-place arithmetic of radf*/radb* is NOT restated).
  internal layout       pffft_zreorder :1158-1193 incl. reversed_copy :1125-1139 — closed form,
                        checked against the reference's own zreorder permutation
  zconvolve             :1534-1684 (accumulate / no_accu, DC+Nyquist fix-up :1626-1629, :1680-1683)
  fastconv              src/pffastconv.c:58-116 (setup), :133-263 (overlap-save loop)
"""
from __future__ import annotations

import numpy as np

FORWARD, BACKWARD = 0, 1
REAL, COMPLEX = 0, 1
SIMD_SZ = 4


# ------------------------------------------------------------------ size helpers
def min_fft_size(transform: int) -> int:  # :78-89
    return 2 * SIMD_SZ * SIMD_SZ if transform == REAL else SIMD_SZ * SIMD_SZ


def is_valid_size(N: int, transform: int) -> bool:  # :91-98
    nmin, r = min_fft_size(transform), N
    while r >= 5 * nmin and r % 5 == 0:
        r //= 5
    while r >= 3 * nmin and r % 3 == 0:
        r //= 3
    while r >= 2 * nmin and r % 2 == 0:
        r //= 2
    return r == nmin


def nearest_transform_size(N: int, transform: int, higher: bool) -> int:  # :100-114
    nmin = min_fft_size(transform)
    N = max(N, nmin)
    N = nmin * ((N + nmin - 1) // nmin) if higher else nmin * (N // nmin)
    d = nmin if higher else -nmin
    while not is_valid_size(N, transform):
        N += d
    return N


def next_power_of_two(N: int) -> int:  # src/pffft_common.c:25-37 (32-bit unsigned bit smearing)
    v = (N - 1) & 0xFFFFFFFF
    for s in (1, 2, 4, 8, 16):
        v |= v >> s
    v = (v + 1) & 0xFFFFFFFF
    return v - (1 << 32) if v >= (1 << 31) else v


def is_power_of_two(N: int) -> bool:  # src/pffft_common.c:39-43
    return bool(N) and not (N & (N - 1))


def new_setup_ok(N: int, transform: int) -> bool:
    """True where pffft_new_setup returns non-NULL (:1066-1078, :1105-1109)."""
    if N <= 0 or N > (1 << 26):
        return False
    if N % (2 * SIMD_SZ * SIMD_SZ if transform == REAL else SIMD_SZ * SIMD_SZ):
        return False
    r = N // SIMD_SZ
    for f in (2, 3, 5):
        while r % f == 0:
            r //= f
    return r == 1


def decompose(n: int, ntryh) -> list:  # :904-928
    nl, fac = n, []
    for ntry in ntryh:
        while nl != 1 and nl % ntry == 0:
            fac.append(ntry)
            nl //= ntry
            if ntry == 2 and len(fac) != 1:  # a factor 2 is rotated to the front (:915-921)
                fac = [2] + fac[:-1]
    return fac


NTRYH_CPLX = (5, 3, 4, 2)  # cffti1_ps :967
NTRYH_REAL = (4, 2, 3, 5)  # rffti1_ps :934


# ------------------------------------------------------------------ lane transform (complex passes)
def _cdtype(dtype):
    return np.complex64 if np.dtype(dtype) == np.float32 else np.complex128


def cfft_lane(x: np.ndarray, factors, sign: int, dtype) -> np.ndarray:
    """cfftf1_ps (:1004-1048) for one lane: pass k1 views the data as cc[l1][ip][ido], does the
    radix-ip butterfly over the middle axis, multiplies output j by the pass twiddle
    wa_j[i] = cos/sin(i * (j*l1) * 2*pi/n) (cffti1_ps :974-1000, applied with sign `isign`), and
    writes ch[ip][l1][ido]."""
    rdt, cdt = np.dtype(dtype), _cdtype(dtype)
    n = x.size
    c = np.asarray(x, dtype=cdt)
    argh = rdt.type(2 * rdt.type(np.pi)) / rdt.type(n)
    l1 = 1
    for ip in factors:
        ido = n // (l1 * ip)
        cc = c.reshape(l1, ip, ido)
        jj = np.arange(ip)
        D = np.exp(sign * 2j * np.pi * np.outer(jj, jj) / ip).astype(cdt)
        y = np.einsum("pj,kji->pki", D, cc).astype(cdt)
        fi = np.arange(ido).astype(rdt)
        for j in range(1, ip):
            argld = rdt.type(j * l1) * argh
            ang = (fi * argld).astype(rdt)
            w = (np.cos(ang).astype(rdt) + 1j * sign * np.sin(ang).astype(rdt)).astype(cdt)
            y[j] *= w[None, :]
        c = y.reshape(-1)
        l1 *= ip
    return c


def _e_table(N: int, dtype) -> np.ndarray:
    """e[] of pffft_new_setup (:1089-1097) as complex W(m, k) = exp(-2 pi i (m+1) k / N), k < N/4."""
    rdt = np.dtype(dtype)
    k = np.arange(N // 4).astype(rdt)
    out = []
    for m in range(3):
        A = (rdt.type(-2) * rdt.type(np.pi) * rdt.type(m + 1) * k / rdt.type(N)).astype(rdt)
        out.append((np.cos(A).astype(rdt) + 1j * np.sin(A).astype(rdt)).astype(_cdtype(dtype)))
    return np.stack(out)


# ------------------------------------------------------------------ internal ("unordered") layout
def bin_of(v: np.ndarray, l: np.ndarray, n: int, is_real: bool) -> np.ndarray:
    """Bin stored at slot l of 4-scalar group v (pffft_zreorder :1158-1193 in closed form).
    n = number of complex bins of the vector (N complex / N/2 real)."""
    b, q = v >> 3, (v >> 1) & 3
    t = 4 * b + l
    if not is_real:
        return q * (n // 4) + t
    return np.select([q == 0, q == 2, q == 1],
                     [t, n // 2 + t, np.where(t > 0, n // 2 - t, n // 4)],
                     np.where(t > 0, n - t, 3 * (n // 4)))


def internal_index_table(N: int, transform: int) -> np.ndarray:
    """perm with canonical[j] = internal[perm[j]] (what pffft_zreorder(FORWARD) applies)."""
    n = N if transform == COMPLEX else N // 2
    v, l = np.meshgrid(np.arange(n // 2), np.arange(4), indexing="ij")
    canon = 2 * bin_of(v, l, n, transform == REAL) + (v & 1)
    perm = np.empty(2 * n, dtype=np.int64)
    perm[canon.ravel()] = (4 * v + l).ravel()
    return perm


def zreorder(x: np.ndarray, N: int, transform: int, direction: int) -> np.ndarray:
    perm = internal_index_table(N, transform)
    x = np.asarray(x)
    if direction == FORWARD:
        return x[perm]
    out = np.empty_like(x)
    out[perm] = x
    return out


# ------------------------------------------------------------------ transforms
def _canon_to_c(x):
    return x[0::2] + 1j * x[1::2]


def _c_to_canon(z, dtype):
    out = np.empty(2 * z.size, dtype=dtype)
    out[0::2], out[1::2] = z.real, z.imag
    return out


def cplx_forward_canonical(x: np.ndarray, N: int, dtype) -> np.ndarray:
    """pffft_transform_internal forward, complex (:1488-1496): uninterleave, four lane FFTs of
    length N/4 over x[l::4], pffft_cplx_finalize (:1195-1237: twiddle sub-FFT m by e[], radix-4)."""
    cdt = _cdtype(dtype)
    z = _canon_to_c(np.asarray(x, dtype=dtype)).astype(cdt)
    n4 = N // 4
    fac = decompose(n4, NTRYH_CPLX)
    Y = np.stack([cfft_lane(z[l::4], fac, -1, dtype) for l in range(4)])  # Y[l][k'], k' < N/4
    e = _e_table(N, dtype)
    Y[1:] *= e
    D4 = np.array([[1, 1, 1, 1], [1, -1j, -1, 1j], [1, -1, 1, -1], [1, 1j, -1, -1j]], dtype=cdt)
    X = (D4 @ Y).astype(cdt)  # X[q][k'] = X[q N/4 + k']
    return _c_to_canon(X.reshape(-1), dtype)


def cplx_backward_canonical(X: np.ndarray, N: int, dtype) -> np.ndarray:
    """Backward, complex (:1513-1518): pffft_cplx_preprocess (:1239-1270: inverse radix-4, conj e[]),
    lane FFTs with isign=+1, re-interleave.  Unscaled: backward(forward(x)) = N x."""
    cdt = _cdtype(dtype)
    Z = _canon_to_c(np.asarray(X, dtype=dtype)).astype(cdt).reshape(4, N // 4)
    D4 = np.array([[1, 1, 1, 1], [1, 1j, -1, -1j], [1, -1, 1, -1], [1, -1j, -1, 1j]], dtype=cdt)
    Y = (D4 @ Z).astype(cdt)
    Y[1:] *= np.conj(_e_table(N, dtype))
    fac = decompose(N // 4, NTRYH_CPLX)
    z = np.empty(N, dtype=cdt)
    for l in range(4):
        z[l::4] = cfft_lane(Y[l], fac, +1, dtype)
    return _c_to_canon(z, dtype)


def real_forward_canonical(x: np.ndarray, N: int, dtype) -> np.ndarray:
    """Forward, real (:1484-1487): four lane real transforms of length N/4 (rfftf1_ps over x[l::4],
    factors from decompose(N/4, {4,2,3,5}) walked in reverse, :818-819) then pffft_real_finalize
    (:1330-1372): X[k] = sum_l W_N^(l k) Y_l[k mod N/4], k = 0..N/2.  Canonical output: N/2 complex
    bins, bin 0 = (DC, Nyquist) (include/pffft/pffft.h:144-152)."""
    cdt = _cdtype(dtype)
    x = np.asarray(x, dtype=dtype)
    n4 = N // 4
    fac = decompose(n4, NTRYH_REAL)[::-1]
    Y = np.stack([cfft_lane(x[l::4].astype(cdt), fac, -1, dtype) for l in range(4)])
    k = np.arange(N // 2 + 1)
    e = _e_table(N, dtype)  # W_N^((m+1) k') for k' < N/4
    W = np.exp(-2j * np.pi * np.outer(np.arange(4), k) / N)
    W[1:, : n4] = e  # working-precision table where the reference has one
    Xf = (W.astype(cdt) * Y[:, k % n4]).sum(axis=0).astype(cdt)
    out = np.empty(N, dtype=dtype)
    out[0], out[1] = Xf[0].real, Xf[N // 2].real
    out[2::2], out[3::2] = Xf[1: N // 2].real, Xf[1: N // 2].imag
    return out


def real_backward_canonical(X: np.ndarray, N: int, dtype) -> np.ndarray:
    """Backward, real (:1508-1511): pffft_real_preprocess + rfftb1_ps.  Restated as the complex
    backward algorithm applied to the Hermitian extension of the half spectrum; result is N*x."""
    X = np.asarray(X, dtype=dtype)
    h = np.empty(N // 2 + 1, dtype=_cdtype(dtype))
    h[0], h[N // 2] = X[0], X[1]
    h[1: N // 2] = X[2::2] + 1j * X[3::2]
    full = np.concatenate([h, np.conj(h[-2:0:-1])])
    z = cplx_backward_canonical(_c_to_canon(full, dtype), N, dtype)
    return z[0::2].astype(dtype)


def transform(x, N: int, transform_: int, direction: int, ordered: bool, dtype=np.float32) -> np.ndarray:
    """pffft_transform (ordered=False) / pffft_transform_ordered (ordered=True) (:1816-1822)."""
    x = np.asarray(x, dtype=dtype)
    if direction == FORWARD:
        c = (cplx_forward_canonical if transform_ == COMPLEX else real_forward_canonical)(x, N, dtype)
        return c if ordered else zreorder(c, N, transform_, BACKWARD)
    c = x if ordered else zreorder(x, N, transform_, FORWARD)
    return (cplx_backward_canonical if transform_ == COMPLEX else real_backward_canonical)(c, N, dtype)


# ------------------------------------------------------------------ spectral multiply
def zconvolve(a, b, ab, scaling, transform_: int, accumulate: bool, dtype=np.float32) -> np.ndarray:
    """pffft_zconvolve_accumulate (:1534-1630) / _no_accu (:1632-1684) on internal-layout vectors:
    per (re-vector, im-vector) pair a complex product, ab (+)= prod*scaling; for real transforms
    lane 0 of the first pair holds DC and Nyquist, both real, fixed up separately (:1626-1629)."""
    a, b = np.asarray(a, dtype=dtype), np.asarray(b, dtype=dtype)
    out = np.array(ab, dtype=dtype, copy=True)
    sc = np.dtype(dtype).type(scaling)
    A, B = a.reshape(-1, 2, 4), b.reshape(-1, 2, 4)
    pr = A[:, 0] * B[:, 0] - A[:, 1] * B[:, 1]
    pi = A[:, 0] * B[:, 1] + A[:, 1] * B[:, 0]
    O = out.reshape(-1, 2, 4)
    ab_r0, ab_i0 = O[0, 0, 0], O[0, 1, 0]
    if accumulate:
        O[:, 0] = pr * sc + O[:, 0]
        O[:, 1] = pi * sc + O[:, 1]
    else:
        O[:, 0], O[:, 1] = pr * sc, pi * sc
    if transform_ == REAL:
        O[0, 0, 0] = (ab_r0 if accumulate else 0) + A[0, 0, 0] * B[0, 0, 0] * sc
        O[0, 1, 0] = (ab_i0 if accumulate else 0) + A[0, 1, 0] * B[0, 1, 0] * sc
    return out


# ------------------------------------------------------------------ overlap-save FIR
CPLX_INP_OUT, CPLX_FILTER, DIRECT_INP, DIRECT_OUT, CPLX_SINGLE_FFT, SYMMETRIC, CORRELATION = 1, 2, 4, 8, 16, 32, 64


def fastconv_setup(taps, block_len: int, flags: int):
    """pffastconv_new_setup (src/pffastconv.c:58-116).  Returns dict or None (NULL)."""
    taps = np.asarray(taps, dtype=np.float32)
    filter_len = taps.size
    cf = 2 if (flags & CPLX_INP_OUT) and (flags & CPLX_SINGLE_FFT) else 1
    nfft = max(2 * next_power_of_two(filter_len - 1), 2 * SIMD_SZ * SIMD_SZ)
    if flags & CPLX_FILTER:
        return None
    if block_len > nfft:
        nfft = next_power_of_two(block_len)
    out_block = nfft
    nfft *= cf
    xt = np.zeros(nfft, dtype=np.float32)
    for i in range(filter_len):
        xt[(nfft - cf * i) & (nfft - 1)] = taps[i] if (flags & CORRELATION) else taps[filter_len - 1 - i]
    hf = transform(xt, nfft, REAL, FORWARD, False, np.float32)
    return dict(Nfft=nfft, blockLen=out_block, filterLen=(2 * filter_len - 1) if cf == 2 else filter_len,
                flags=flags, cf=cf, Hf=hf, scale=np.float32(1.0 / nfft))


def fastconv_apply(s, x, flush: bool):
    """pffastconv_apply (src/pffastconv.c:133-263).  Returns (y, produced)."""
    x = np.asarray(x, dtype=np.float32)
    nfft, flen, cf, flags = s["Nfft"], s["filterLen"], s["cf"], s["flags"]
    cplx = bool(flags & CPLX_INP_OUT)
    input_len = x.size if (cf == 2 or not cplx) else x.size // 2
    y = np.zeros(x.size, dtype=np.float32)
    max_off = (input_len - flen + 1) if flush else (input_len - nfft + 1)
    parts = 2 if (cplx and cf == 1) else 1
    off = 0
    while off < max_off:
        proc = min(nfft, input_len - off)
        num_out = proc - flen + 1
        if cf == 2:
            num_out &= ~1
            if not num_out:
                break
        for part in range(parts):
            xt = np.zeros(nfft, dtype=np.float32)
            xt[:proc] = x[2 * off + part: 2 * (off + proc): 2] if parts == 2 else x[off: off + proc]
            xf = transform(xt, nfft, REAL, FORWARD, False)
            mf = zconvolve(xf, s["Hf"], np.zeros(nfft, np.float32), s["scale"], REAL, False)
            yt = transform(mf, nfft, REAL, BACKWARD, False)
            if parts == 2:
                y[2 * off + part: 2 * (off + num_out): 2] = yt[:num_out]
            else:
                y[off: off + num_out] = yt[:num_out]
        off += num_out
    produced = off // cf
    return y[: (2 * produced if cplx else produced)], produced


def slow_conv(x, taps) -> np.ndarray:
    """Naive time-domain FIR — the reference test's ground truth slow_conv_R
    (tests/test_pffastconv.c:175-213): y[i] = sum_j x[i+j] * h_rev... = valid-mode convolution."""
    return np.convolve(np.asarray(x, np.float64), np.asarray(taps, np.float64), mode="valid")
