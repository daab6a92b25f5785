"""NumPy twin of the oracle: the same cascade written in the DILATED form
(out[t] = sum_l f[l] * x[idx(t - l*2^(j-1))], no zero stuffing), vectorised over
t and accumulated tap by tap in ascending l.  TEST INFRASTRUCTURE ONLY.

It exists to cross-check oracle/vw_oracle.c, which follows the reference
literally (zero-stuffed filters, dense loops): both must agree bit for bit,
which also demonstrates that skipping the stuffed zeros changes nothing.

Reference formulas: internal/ScalarOps.java:743-789 (dilated circular form),
:799-845 (zero padding / symmetric), modwt/MultiLevelMODWTTransform.java:621-689
(inverse), :801-806 (truncation of filters longer than the signal).
"""
import math

import numpy as np

PERIODIC, ZERO_PADDING, SYMMETRIC = 0, 1, 2


def _reflect(idx, n):
    period = 2 * n
    m = np.mod(idx, period)
    return np.where(m >= n, period - m - 1, m)


def _taps_used(length, stride, n):
    # a zero-stuffed filter truncated to n entries keeps the taps with l*stride < n
    return min(length, (n - 1) // stride + 1)


def _convolve(x, taps, stride, boundary):
    """x [B][N] -> [B][N]; taps already scaled."""
    b, n = x.shape
    t = np.arange(n)
    out = np.zeros_like(x)
    for l, f in enumerate(taps):
        idx = t - l * stride
        if boundary == PERIODIC:
            v = x[:, np.mod(idx, n)]
            out = out + v * f
        elif boundary == ZERO_PADDING:
            ok = idx >= 0
            v = np.where(ok[None, :], x[:, np.clip(idx, 0, n - 1)], 0.0)
            # skipped terms add nothing: adding +0.0 keeps the running sum's bits unless it is -0.0,
            # which cannot occur because every sum starts from +0.0
            out = np.where(ok[None, :], out + v * f, out)
        else:
            out = out + x[:, _reflect(idx, n)] * f
    return out


def decompose(x, wavelet, levels, boundary):
    scale = 1.0 / math.sqrt(2.0)
    lo = [c * scale for c in wavelet.dec_lo]
    hi = [c * scale for c in wavelet.dec_hi]
    x = np.asarray(x, dtype=np.float64)
    b, n = x.shape
    out = np.empty((b, levels + 1, n))
    cur = x
    for j in range(1, levels + 1):
        s = 1 << (j - 1)
        nlo, nhi = _taps_used(len(lo), s, n), _taps_used(len(hi), s, n)
        out[:, j - 1] = _convolve(cur, hi[:nhi], s, boundary)
        cur = _convolve(cur, lo[:nlo], s, boundary)
    out[:, levels] = cur
    return out


def reconstruct(coeffs, wavelet, boundary):
    scale = 1.0 / math.sqrt(2.0)
    lo = [c * scale for c in wavelet.rec_lo]
    hi = [c * scale for c in wavelet.rec_hi]
    c = np.asarray(coeffs, dtype=np.float64)
    b, jp1, n = c.shape
    levels = jp1 - 1
    t = np.arange(n)
    rec = c[:, levels]
    for j in range(levels, 0, -1):
        s = 1 << (j - 1)
        d = c[:, j - 1]
        nlo, nhi = _taps_used(len(lo), s, n), _taps_used(len(hi), s, n)
        out = np.zeros_like(rec)
        if boundary == SYMMETRIC:
            for l in range(max(nlo, nhi)):
                idx = _reflect(t - l * s, n)
                if l < nlo:
                    out = out + lo[l] * rec[:, idx]
                if l < nhi:
                    out = out + hi[l] * d[:, idx]
        else:
            for l in range(nlo):
                idx = t + l * s
                if boundary == PERIODIC:
                    idx = np.mod(idx, n)
                    out = out + (lo[l] * rec[:, idx] + hi[l] * d[:, idx])
                else:
                    ok = idx < n
                    ic = np.clip(idx, 0, n - 1)
                    out = np.where(ok[None, :], out + (lo[l] * rec[:, ic] + hi[l] * d[:, ic]), out)
        rec = out
    return rec
