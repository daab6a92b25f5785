"""Wavelet filter banks for the MODWT/SWT path.

Tap tables live in wavelet_tables.json (low-pass literals only, as C99 hex
floats).  The derived filters are rebuilt here with the reference's arithmetic:

* orthogonal families: g[i] = (+1 if i even else -1) * h[L-1-i]
  (reference api/Daubechies.java:320-327, api/Symlet.java:462-470,
  api/Coiflet.java:629-637, api/DiscreteMeyer.java:102-110); reconstruction
  filters are the decomposition filters (api/OrthogonalWavelet.java:21-33).
* Haar: h = {s, s}, g = {s, -s}, s = 1/sqrt(2) (api/Haar.java:40-44).
* biorthogonal splines: g = qmf(rec_lo), g~ = qmf(dec_lo) with sign
  (+1 if (len-1-i) even else -1) (api/BiorthogonalSpline.java:297-327);
  reverse biorthogonal swaps the decomposition and reconstruction pairs
  (api/ReverseBiorthogonalSpline.java:95-117).
"""
from __future__ import annotations

import json
import os
from dataclasses import dataclass
from functools import lru_cache

import numpy as np

_TABLES = os.path.join(os.path.dirname(__file__), "wavelet_tables.json")


@dataclass(frozen=True)
class Wavelet:
    """Mirror of the reference's DiscreteWavelet accessors (api/DiscreteWavelet.java:10)."""

    name: str
    dec_lo: tuple
    dec_hi: tuple
    rec_lo: tuple
    rec_hi: tuple
    orthogonal: bool

    def lowPassDecomposition(self):
        return np.array(self.dec_lo, dtype=np.float64)

    def highPassDecomposition(self):
        return np.array(self.dec_hi, dtype=np.float64)

    def lowPassReconstruction(self):
        return np.array(self.rec_lo, dtype=np.float64)

    def highPassReconstruction(self):
        return np.array(self.rec_hi, dtype=np.float64)


def _qmf_orthogonal(h):
    n = len(h)
    return [(1 if i % 2 == 0 else -1) * h[n - 1 - i] for i in range(n)]


def _qmf_biorthogonal(h):
    n = len(h)
    return [(1 if (n - 1 - i) % 2 == 0 else -1) * h[n - 1 - i] for i in range(n)]


@lru_cache(maxsize=None)
def _load():
    with open(_TABLES) as f:
        doc = json.load(f)
    out = {}
    for name, taps in doc["orthogonal"].items():
        h = [float.fromhex(t) for t in taps]
        if name == "haar":
            g = [h[0], -h[1]]
        else:
            g = _qmf_orthogonal(h)
        out[name] = Wavelet(name, tuple(h), tuple(g), tuple(h), tuple(g), True)
    for name, pair in doc["biorthogonal"].items():
        dec_lo = [float.fromhex(t) for t in pair["dec_lo"]]
        rec_lo = [float.fromhex(t) for t in pair["rec_lo"]]
        dec_hi = _qmf_biorthogonal(rec_lo)
        rec_hi = _qmf_biorthogonal(dec_lo)
        out[name] = Wavelet(name, tuple(dec_lo), tuple(dec_hi), tuple(rec_lo), tuple(rec_hi), False)
        rname = "rbio" + name[4:]  # rbioX.Y
        out[rname] = Wavelet(rname, tuple(rec_lo), tuple(rec_hi), tuple(dec_lo), tuple(dec_hi), False)
    return out


def get_wavelet(name: str) -> Wavelet:
    """Look a wavelet up by the reference's code ("haar", "db4", "sym8", "coif3",
    "bior4.4", "rbio2.2", "dmey"; reference api/WaveletName.java:9-121)."""
    table = _load()
    key = name.lower()
    if key not in table:
        raise KeyError("unknown wavelet %r" % name)
    return table[key]


def available_wavelets():
    return sorted(_load().keys())


HAAR = "haar"
