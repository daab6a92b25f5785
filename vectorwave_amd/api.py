"""Host-side mirror of the reference's MODWT / SWT API over the C ABI.

Same class and method names, argument meaning and error behaviour as the
reference's Java classes, so that parity tests read like the reference's own
tests.  Every method that computes goes through libvwgpu.so (include/vwgpu.h);
nothing here computes a transform on the CPU.

Reference classes mirrored (paths under src/main/java/ai/prophetizo/wavelet/):
  modwt/MODWTTransform.java:69            -> MODWTTransform
  modwt/MultiLevelMODWTTransform.java:66  -> MultiLevelMODWTTransform
  swt/VectorWaveSwtAdapter.java:61        -> VectorWaveSwtAdapter
  WaveletOperations.java:12               -> WaveletOperations
  modwt/MODWTResult.java:24, modwt/MultiLevelMODWTResult.java:24,
  modwt/MutableMultiLevelMODWTResult.java:19 -> result containers
  exception/ErrorCode.java:18             -> ErrorCode strings on the exceptions
"""
from __future__ import annotations

import ctypes as C
import enum
import math
from typing import List, Optional, Sequence

import numpy as np

from . import _native as N
from .wavelets import Wavelet, get_wavelet


class BoundaryMode(enum.IntEnum):
    """Boundary modes the MODWT classes accept (modwt/MODWTTransform.java:88-90)."""

    PERIODIC = 0
    ZERO_PADDING = 1
    SYMMETRIC = 2


class WaveletTransformException(Exception):
    def __init__(self, message, error_code=None):
        super().__init__(message)
        self.error_code = error_code

    def getErrorCode(self):
        return self.error_code


class InvalidSignalException(WaveletTransformException):
    """exception/InvalidSignalException.java"""


class InvalidArgumentException(WaveletTransformException):
    """exception/InvalidArgumentException.java"""


_STATUS_TO_CODE = {
    N.VW_E_NULL: "VAL_001",
    N.VW_E_NONFINITE: "VAL_003",
    N.VW_E_NOT_POSITIVE: "VAL_004",
    N.VW_E_TOO_LARGE: "VAL_005",
    N.VW_E_EMPTY: "VAL_006",
    N.VW_E_LENGTH_MISMATCH: "VAL_007",
    N.VW_E_BOUNDARY: "CFG_003",
    N.VW_E_LEVEL: "CFG_004",
    N.VW_E_UNSUPPORTED: "CFG_005",
}


def _raise_for_status(ctx: N.Context, st: int):
    """Map a vw_status back onto the exception class the reference throws."""
    if st == N.VW_OK:
        return
    msg = ctx.last_error()
    code = _STATUS_TO_CODE.get(st)
    if st in (N.VW_E_NONFINITE, N.VW_E_EMPTY):
        raise InvalidSignalException(msg, code)
    if st in (N.VW_E_LEVEL, N.VW_E_BOUNDARY, N.VW_E_TOO_LARGE, N.VW_E_UNSUPPORTED):
        raise InvalidArgumentException(msg, code)
    if st == N.VW_E_NULL:
        raise TypeError(msg)  # NullPointerException
    if st == N.VW_E_NOT_POSITIVE:
        raise ValueError(msg)  # IllegalArgumentException
    if st == N.VW_E_LENGTH_MISMATCH:
        raise IndexError(msg)  # ArrayIndexOutOfBoundsException in the reference
    raise N.NativeLibraryError("libvwgpu status %d: %s" % (st, msg))


def _dp(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _vp(a: np.ndarray):
    return C.c_void_p(a.ctypes.data)


def _taps(t) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(t, dtype=np.float64))


def _require_signal(signal, name="signal") -> np.ndarray:
    if signal is None:
        raise TypeError("%s cannot be null" % name)
    return np.ascontiguousarray(np.asarray(signal, dtype=np.float64))


def _as_wavelet(w) -> Wavelet:
    if w is None:
        raise TypeError("wavelet cannot be null")
    return get_wavelet(w) if isinstance(w, str) else w


# ---------------------------------------------------------------------------
# result containers
# ---------------------------------------------------------------------------

class MODWTResult:
    """modwt/MODWTResult.java:24 + MODWTResultImpl.java:31-63 (finite check, defensive copies)."""

    def __init__(self, approximationCoeffs, detailCoeffs, _validated=False):
        if approximationCoeffs is None:
            raise TypeError("approximationCoeffs cannot be null")
        if detailCoeffs is None:
            raise TypeError("detailCoeffs cannot be null")
        a = np.array(approximationCoeffs, dtype=np.float64)
        d = np.array(detailCoeffs, dtype=np.float64)
        if a.shape != d.shape:
            raise ValueError("Approximation and detail coefficients must have the same length.")
        if a.size == 0:
            raise ValueError("Coefficient arrays cannot be empty")
        if not _validated:
            for arr, nm in ((a, "approximationCoeffs"), (d, "detailCoeffs")):
                bad = np.flatnonzero(~np.isfinite(arr))
                if bad.size:
                    raise InvalidSignalException("%s contains a non-finite value at index %d" % (nm, bad[0]), "VAL_003")
        self._a, self._d = a, d

    @staticmethod
    def create(approximationCoeffs, detailCoeffs):
        return MODWTResult(approximationCoeffs, detailCoeffs)

    def approximationCoeffs(self):
        return self._a.copy()

    def detailCoeffs(self):
        return self._d.copy()

    def getSignalLength(self):
        return int(self._a.size)

    def isValid(self):
        return bool(np.isfinite(self._a).all() and np.isfinite(self._d).all())


class MultiLevelMODWTResult:
    """Immutable J-level result (modwt/MultiLevelMODWTResultImpl.java:9), backed by one
    contiguous [J+1][N] block ordered d_1..d_J, a_J."""

    _mutable = False

    def __init__(self, block: np.ndarray):
        assert block.ndim == 2
        self._block = block
        self._levels = block.shape[0] - 1
        self._n = block.shape[1]

    def getLevels(self):
        return self._levels

    def getSignalLength(self):
        return self._n

    def _validate_level(self, level):
        if level < 1 or level > self._levels:
            raise ValueError("Level must be between 1 and %d, got: %d" % (self._levels, level))

    def getDetailCoeffsAtLevel(self, level):
        self._validate_level(level)
        return self._block[level - 1].copy()

    def getApproximationCoeffs(self):
        return self._block[self._levels].copy()

    @staticmethod
    def _energy(c):
        # sequential sum of squares, like the reference's computeEnergy loops
        e = 0.0
        for v in c.tolist():
            e += v * v
        return e

    def getDetailEnergyAtLevel(self, level):
        self._validate_level(level)
        return self._energy(self._block[level - 1])

    def getApproximationEnergy(self):
        return self._energy(self._block[self._levels])

    def getTotalEnergy(self):
        total = self.getApproximationEnergy()
        for level in range(1, self._levels + 1):
            total += self.getDetailEnergyAtLevel(level)
        return total

    def getRelativeEnergyDistribution(self):
        # immutable impl order: [approx, d1..dJ] (modwt/MultiLevelMODWTResultImpl.java:131-136)
        total = self.getTotalEnergy()
        dist = np.zeros(self._levels + 1)
        if total > 0:
            dist[0] = self.getApproximationEnergy() / total
            for level in range(1, self._levels + 1):
                dist[level] = self.getDetailEnergyAtLevel(level) / total
        return dist

    def copy(self):
        return type(self)(self._block.copy())

    def isValid(self):
        return bool(np.isfinite(self._block).all())

    def block(self) -> np.ndarray:
        """The [J+1][N] coefficient block (d_1..d_J, a_J); a view, not a copy."""
        return self._block


class MutableMultiLevelMODWTResult(MultiLevelMODWTResult):
    """modwt/MutableMultiLevelMODWTResultImpl.java:16"""

    _mutable = True

    def getMutableDetailCoeffs(self, level):
        self._validate_level(level)
        return self._block[level - 1]

    def getMutableApproximationCoeffs(self):
        return self._block[self._levels]

    def _validate_coeffs(self, coeffs, name):
        if coeffs is None:
            raise TypeError("%s cannot be null" % name)
        c = np.asarray(coeffs, dtype=np.float64)
        if c.shape != (self._n,):
            raise ValueError("%s length %d does not match signal length %d" % (name, c.size, self._n))
        return c

    def setDetailCoeffs(self, level, coeffs):
        self._validate_level(level)
        self._block[level - 1] = self._validate_coeffs(coeffs, "detail coefficients at level %d" % level)

    def setApproximationCoeffs(self, coeffs):
        self._block[self._levels] = self._validate_coeffs(coeffs, "approximation coefficients")

    def clearCaches(self):
        pass  # energies are not cached here

    def applyThreshold(self, level, threshold, soft, ctx: Optional[N.Context] = None):
        """modwt/MutableMultiLevelMODWTResult.java:83-114; level 0 = approximation."""
        if level == 0:
            row = self._block[self._levels]
        else:
            self._validate_level(level)
            row = self._block[level - 1]
        if threshold < 0:
            # the reference's loop with a negative threshold: soft shrinks away from zero
            # by |thr|, hard keeps everything; mirrored on the host because the C ABI
            # (like WaveletOperations) rejects negative thresholds
            if soft:
                row[:] = np.where(np.abs(row) > threshold, np.sign(row) * (np.abs(row) - threshold), 0.0)
            return
        ctx = ctx or N.default_context()
        out = np.empty_like(row)
        st = ctx.lib.vw_threshold(ctx.handle, _vp(np.ascontiguousarray(row)), row.size, float(threshold),
                                  1 if soft else 0, _vp(out))
        _raise_for_status(ctx, st)
        row[:] = out

    def getRelativeEnergyDistribution(self):
        # mutable impl order: [d1..dJ, approx] (modwt/MutableMultiLevelMODWTResultImpl.java:203-207)
        total = self.getTotalEnergy()
        dist = np.zeros(self._levels + 1)
        if total > 0:
            for level in range(1, self._levels + 1):
                dist[level - 1] = self.getDetailEnergyAtLevel(level) / total
            dist[self._levels] = self.getApproximationEnergy() / total
        return dist

    def toImmutable(self):
        return MultiLevelMODWTResult(self._block.copy())


# ---------------------------------------------------------------------------
# WaveletOperations facade
# ---------------------------------------------------------------------------

class WaveletOperations:
    """WaveletOperations.java:12: the static operator facade, now one downcall per batch."""

    @staticmethod
    def _convolve(signal, filt, output, boundary, level=1, ctx=None):
        ctx = ctx or N.default_context()
        s = _require_signal(signal)
        f = _taps(filt)
        x = s.reshape(1, -1) if s.ndim == 1 else s
        out = np.empty_like(x)
        st = ctx.lib.vw_modwt_convolve_batch(ctx.handle, _vp(x), x.shape[0], x.shape[1], _dp(f), f.size, level,
                                             int(boundary), _vp(out))
        _raise_for_status(ctx, st)
        out = out.reshape(s.shape)
        if output is not None:
            output[...] = out
            return output
        return out

    @staticmethod
    def circularConvolveMODWT(signal, filt, output=None, ctx=None):
        return WaveletOperations._convolve(signal, filt, output, BoundaryMode.PERIODIC, 1, ctx)

    @staticmethod
    def zeroPaddingConvolveMODWT(signal, filt, output=None, ctx=None):
        return WaveletOperations._convolve(signal, filt, output, BoundaryMode.ZERO_PADDING, 1, ctx)

    @staticmethod
    def symmetricConvolveMODWT(signal, filt, output=None, ctx=None):
        return WaveletOperations._convolve(signal, filt, output, BoundaryMode.SYMMETRIC, 1, ctx)

    @staticmethod
    def circularConvolveMODWTLevel(signal, filt, output=None, level=1, ctx=None):
        """internal/ScalarOps.java:743-789"""
        return WaveletOperations._convolve(signal, filt, output, BoundaryMode.PERIODIC, level, ctx)

    @staticmethod
    def _threshold(coefficients, threshold, soft, ctx=None):
        if coefficients is None:
            raise ValueError("Coefficients array cannot be null")
        if threshold < 0:
            raise ValueError("Threshold must be non-negative")
        ctx = ctx or N.default_context()
        c = np.ascontiguousarray(np.asarray(coefficients, dtype=np.float64))
        out = np.empty_like(c)
        st = ctx.lib.vw_threshold(ctx.handle, _vp(c), c.size, float(threshold), 1 if soft else 0, _vp(out))
        _raise_for_status(ctx, st)
        return out

    @staticmethod
    def softThreshold(coefficients, threshold, ctx=None):
        return WaveletOperations._threshold(coefficients, threshold, True, ctx)

    @staticmethod
    def hardThreshold(coefficients, threshold, ctx=None):
        return WaveletOperations._threshold(coefficients, threshold, False, ctx)

    @staticmethod
    def getPerformanceInfo(ctx=None):
        ctx = ctx or N.default_context()
        return ctx.performance_info()


# ---------------------------------------------------------------------------
# MODWTTransform (single level)
# ---------------------------------------------------------------------------

class MODWTTransform:
    """modwt/MODWTTransform.java:69"""

    def __init__(self, wavelet, boundaryMode, ctx: Optional[N.Context] = None):
        self.wavelet = _as_wavelet(wavelet)
        if boundaryMode is None:
            raise TypeError("boundaryMode cannot be null")
        try:
            self.boundaryMode = BoundaryMode(boundaryMode)
        except ValueError:
            raise InvalidArgumentException(
                "MODWT only supports PERIODIC, ZERO_PADDING, and SYMMETRIC boundary modes", "CFG_003")
        self._ctx = ctx

    @property
    def ctx(self) -> N.Context:
        if self._ctx is None:
            self._ctx = N.default_context()
        return self._ctx

    def getWavelet(self):
        return self.wavelet

    def getBoundaryMode(self):
        return self.boundaryMode

    def getPerformanceInfo(self):
        return self.ctx.performance_info()

    # -- raw batch calls over contiguous blocks --------------------------------
    def forward_block(self, x: np.ndarray, check_finite: bool) -> np.ndarray:
        """x [B][N] -> coeffs [B][2][N] (detail, approximation)."""
        ctx = self.ctx
        b, n = x.shape
        h, g = _taps(self.wavelet.dec_lo), _taps(self.wavelet.dec_hi)
        out = np.empty((b, 2, n))
        bad = C.c_int64(-1)
        flags = N.VW_FLAG_SINGLE_LEVEL | (N.VW_FLAG_CHECK_FINITE if check_finite else 0)
        st = ctx.lib.vw_modwt_forward_batch(ctx.handle, _vp(x), b, n, 1, _dp(h), h.size, _dp(g), g.size,
                                            int(self.boundaryMode), flags, _vp(out), C.byref(bad))
        _raise_for_status(ctx, st)
        return out

    def inverse_block(self, coeffs: np.ndarray, check_finite: bool, batch_variant: bool) -> np.ndarray:
        ctx = self.ctx
        b, two, n = coeffs.shape
        hr, gr = _taps(self.wavelet.rec_lo), _taps(self.wavelet.rec_hi)
        out = np.empty((b, n))
        bad = C.c_int64(-1)
        flags = N.VW_FLAG_SINGLE_LEVEL | (N.VW_FLAG_CHECK_FINITE if check_finite else 0) | (
            N.VW_FLAG_SYMMETRIC_BATCH_VARIANT if batch_variant else 0)
        st = ctx.lib.vw_modwt_inverse_batch(ctx.handle, _vp(coeffs), b, n, 1, _dp(hr), hr.size, _dp(gr), gr.size,
                                            int(self.boundaryMode), flags, _vp(out), C.byref(bad))
        _raise_for_status(ctx, st)
        return out

    # -- reference API -----------------------------------------------------------
    def forward(self, signal) -> MODWTResult:
        x = _require_signal(signal)
        if x.size == 0:
            raise InvalidSignalException("Signal cannot be empty", "VAL_006")
        c = self.forward_block(x.reshape(1, -1), check_finite=True)
        return MODWTResult(c[0, 1], c[0, 0], _validated=True)

    def inverse(self, modwtResult: MODWTResult):
        if modwtResult is None:
            raise TypeError("modwtResult cannot be null")
        if not modwtResult.isValid():
            raise InvalidSignalException("MODWTResult contains invalid coefficients", "VAL_003")
        c = np.stack([modwtResult._d, modwtResult._a])[None]
        return self.inverse_block(np.ascontiguousarray(c), check_finite=False, batch_variant=False)[0]

    def forwardBatch(self, signals) -> List[MODWTResult]:
        if signals is None:
            raise TypeError("signals array cannot be null")
        if len(signals) == 0:
            return []
        rows = [_require_signal(s) for s in signals]
        first = rows[0].size
        same = all(r.size == first for r in rows)
        if same and len(rows) >= 4 and first >= 64:
            # forwardBatchOptimized (modwt/MODWTTransform.java:537-587): no input validation; a
            # non-finite coefficient surfaces when the per-signal result objects are built
            c = self.forward_block(np.ascontiguousarray(np.stack(rows)), check_finite=False)
            return [MODWTResult(c[i, 1], c[i, 0]) for i in range(len(rows))]
        return [self.forward(r) for r in rows]

    def inverseBatch(self, results: Sequence[MODWTResult]):
        if results is None:
            raise TypeError("results array cannot be null")
        if len(results) == 0:
            return []
        first = results[0].getSignalLength()
        same = all(r.getSignalLength() == first for r in results)
        if same and len(results) >= 4 and first >= 64:
            c = np.ascontiguousarray(np.stack([np.stack([r._d, r._a]) for r in results]))
            out = self.inverse_block(c, check_finite=False, batch_variant=True)
            return [out[i] for i in range(len(results))]
        return [self.inverse(r) for r in results]


# ---------------------------------------------------------------------------
# MultiLevelMODWTTransform
# ---------------------------------------------------------------------------

class MultiLevelMODWTTransform:
    """modwt/MultiLevelMODWTTransform.java:66"""

    MAX_DECOMPOSITION_LEVELS = 10

    def __init__(self, wavelet, boundaryMode, ctx: Optional[N.Context] = None):
        self.wavelet = _as_wavelet(wavelet)
        if boundaryMode is None:
            raise TypeError("boundaryMode cannot be null")
        self.singleLevelTransform = MODWTTransform(self.wavelet, boundaryMode, ctx)
        self.boundaryMode = self.singleLevelTransform.boundaryMode
        self._ctx = ctx

    @property
    def ctx(self) -> N.Context:
        if self._ctx is None:
            self._ctx = N.default_context()
        return self._ctx

    def getWavelet(self):
        return self.wavelet

    def getBoundaryMode(self):
        return self.boundaryMode

    @staticmethod
    def getMaxDecompositionLevels():
        return MultiLevelMODWTTransform.MAX_DECOMPOSITION_LEVELS

    def getMaximumLevels(self, signalLength):
        lib = N.load()
        return min(int(lib.vw_max_levels(int(signalLength), len(self.wavelet.dec_lo))),
                   self.MAX_DECOMPOSITION_LEVELS)

    # -- batch surface (new: modelled on MODWTTransform.forwardBatch) ----------
    def decompose_block(self, x: np.ndarray, levels: int, check_finite: bool = True) -> np.ndarray:
        """x [B][N] -> coeffs [B][J+1][N]."""
        ctx = self.ctx
        b, n = x.shape
        h, g = _taps(self.wavelet.dec_lo), _taps(self.wavelet.dec_hi)
        out = np.empty((b, levels + 1 if levels >= 0 else 1, n))
        bad = C.c_int64(-1)
        flags = N.VW_FLAG_CHECK_FINITE if check_finite else 0
        st = ctx.lib.vw_modwt_forward_batch(ctx.handle, _vp(x), b, n, int(levels), _dp(h), h.size, _dp(g), g.size,
                                            int(self.boundaryMode), flags, _vp(out), C.byref(bad))
        if st == N.VW_E_LEVEL and check_finite and b > 0 and n > 0:
            # decompose() validates finiteness before the level count
            # (modwt/MultiLevelMODWTTransform.java:216-244)
            st2 = ctx.lib.vw_first_non_finite(ctx.handle, _vp(x), x.size, C.byref(bad))
            if st2 == N.VW_E_NONFINITE:
                _raise_for_status(ctx, st2)
            raise InvalidArgumentException("Invalid number of decomposition levels", "CFG_004")
        _raise_for_status(ctx, st)
        return out

    def reconstruct_block(self, coeffs: np.ndarray, band_mask: Optional[int] = None) -> np.ndarray:
        ctx = self.ctx
        b, jp1, n = coeffs.shape
        hr, gr = _taps(self.wavelet.rec_lo), _taps(self.wavelet.rec_hi)
        out = np.empty((b, n))
        if band_mask is None:
            bad = C.c_int64(-1)
            st = ctx.lib.vw_modwt_inverse_batch(ctx.handle, _vp(coeffs), b, n, jp1 - 1, _dp(hr), hr.size, _dp(gr),
                                                gr.size, int(self.boundaryMode), 0, _vp(out), C.byref(bad))
        else:
            st = ctx.lib.vw_modwt_inverse_bands_batch(ctx.handle, _vp(coeffs), b, n, jp1 - 1, _dp(hr), hr.size, _dp(gr),
                                                      gr.size, int(self.boundaryMode), int(band_mask), _vp(out))
        _raise_for_status(ctx, st)
        return out

    def decomposeBatch(self, signals, levels: Optional[int] = None) -> List[MultiLevelMODWTResult]:
        x = np.ascontiguousarray(np.asarray(signals, dtype=np.float64))
        if x.ndim != 2:
            raise ValueError("decomposeBatch needs equal-length signals [B][N]")
        if levels is None:
            levels = self.getMaximumLevels(x.shape[1])
        c = self.decompose_block(x, levels)
        return [MultiLevelMODWTResult(c[i]) for i in range(x.shape[0])]

    def reconstructBatch(self, results: Sequence[MultiLevelMODWTResult]):
        if len(results) == 0:
            return []
        c = np.ascontiguousarray(np.stack([r.block() for r in results]))
        out = self.reconstruct_block(c)
        return [out[i] for i in range(len(results))]

    # -- reference API -----------------------------------------------------------
    def _decompose(self, signal, levels, cls):
        x = _require_signal(signal)
        if levels is None:
            levels = self.getMaximumLevels(x.size)
        if x.size == 0:
            raise InvalidSignalException("Signal cannot be empty for multi-level MODWT", "VAL_006")
        c = self.decompose_block(x.reshape(1, -1), int(levels))
        return cls(c[0])

    def decompose(self, signal, levels: Optional[int] = None) -> MultiLevelMODWTResult:
        return self._decompose(signal, levels, MultiLevelMODWTResult)

    def decomposeMutable(self, signal, levels: Optional[int] = None) -> MutableMultiLevelMODWTResult:
        return self._decompose(signal, levels, MutableMultiLevelMODWTResult)

    def reconstruct(self, result: MultiLevelMODWTResult):
        if result is None:
            raise TypeError("result cannot be null")
        return self.reconstruct_block(np.ascontiguousarray(result.block())[None])[0]

    def reconstructFromLevel(self, result, startLevel):
        """modwt/MultiLevelMODWTTransform.java:385-410: details finer than startLevel are zeros."""
        if result is None:
            raise TypeError("result cannot be null")
        J = result.getLevels()
        if startLevel < 1 or startLevel > J:
            raise InvalidArgumentException("Invalid start level: %d. Must be between 1 and %d" % (startLevel, J))
        mask = (1 << J)  # approximation
        for level in range(startLevel, J + 1):
            mask |= 1 << (level - 1)
        return self.reconstruct_block(np.ascontiguousarray(result.block())[None], mask)[0]

    def reconstructLevels(self, result, minLevel, maxLevel):
        """modwt/MultiLevelMODWTTransform.java:420-468: band-pass; the approximation takes part
        only when the top level is inside the range."""
        if result is None:
            raise TypeError("result cannot be null")
        J = result.getLevels()
        if minLevel < 1 or maxLevel > J or minLevel > maxLevel:
            raise InvalidArgumentException("Invalid level range for partial reconstruction", "CFG_004")
        mask = 0
        for level in range(minLevel, maxLevel + 1):
            mask |= 1 << (level - 1)
        if J <= maxLevel:
            mask |= 1 << J
        return self.reconstruct_block(np.ascontiguousarray(result.block())[None], mask)[0]


# ---------------------------------------------------------------------------
# VectorWaveSwtAdapter
# ---------------------------------------------------------------------------

class VectorWaveSwtAdapter:
    """swt/VectorWaveSwtAdapter.java:61"""

    PARALLEL_THRESHOLD = 4096

    def __init__(self, wavelet, boundaryMode=BoundaryMode.PERIODIC, ctx: Optional[N.Context] = None):
        if wavelet is None:
            raise TypeError("Wavelet cannot be null")
        if boundaryMode is None:
            raise TypeError("Boundary mode cannot be null")
        self.wavelet = _as_wavelet(wavelet)
        self.modwtTransform = MultiLevelMODWTTransform(self.wavelet, boundaryMode, ctx)
        self.boundaryMode = self.modwtTransform.boundaryMode
        self._ctx = ctx

    @property
    def ctx(self) -> N.Context:
        if self._ctx is None:
            self._ctx = N.default_context()
        return self._ctx

    def getWavelet(self):
        return self.wavelet

    def getBoundaryMode(self):
        return self.boundaryMode

    def forward(self, signal, levels: Optional[int] = None) -> MutableMultiLevelMODWTResult:
        return self.modwtTransform.decomposeMutable(signal, levels)

    def inverse(self, result: MutableMultiLevelMODWTResult):
        return self.modwtTransform.reconstruct(result)

    def applyThreshold(self, result, level, threshold, soft):
        if result is None:
            raise TypeError("Result cannot be null")
        result.applyThreshold(level, threshold, soft, self.ctx)

    def universalThreshold(self, result) -> float:
        """The threshold applyUniversalThreshold derives (swt/VectorWaveSwtAdapter.java:255-264)."""
        ctx = self.ctx
        blk = np.ascontiguousarray(result.block())[None]
        n = result.getSignalLength()
        thr = np.empty(1)
        st = ctx.lib.vw_swt_universal_threshold_batch(ctx.handle, _vp(blk), 1, n, result.getLevels(),
                                                      math.sqrt(2 * math.log(n)), _vp(thr))
        _raise_for_status(ctx, st)
        return float(thr[0])

    def applyUniversalThreshold(self, result, soft):
        if result is None:
            raise TypeError("Result cannot be null")
        thr = self.universalThreshold(result)
        for level in range(1, result.getLevels() + 1):
            self.applyThreshold(result, level, thr, soft)

    def denoise_block(self, x: np.ndarray, levels: int, threshold: float = -1.0, soft: bool = True,
                      return_thresholds: bool = False):
        """x [B][N] -> denoised [B][N] in one downcall (new batch surface)."""
        ctx = self.ctx
        b, n = x.shape
        w = self.wavelet
        h, g, hr, gr = _taps(w.dec_lo), _taps(w.dec_hi), _taps(w.rec_lo), _taps(w.rec_hi)
        out = np.empty((b, n))
        thr = np.empty(b)
        s2 = math.sqrt(2 * math.log(n)) if n > 0 else 0.0
        st = ctx.lib.vw_swt_denoise_batch(ctx.handle, _vp(x), b, n, int(levels), _dp(h), h.size, _dp(g), g.size,
                                          _dp(hr), hr.size, _dp(gr), gr.size, int(self.boundaryMode),
                                          float(threshold), s2, 1 if soft else 0, _vp(out), _vp(thr))
        _raise_for_status(ctx, st)
        return (out, thr) if return_thresholds else out

    def denoise(self, signal, levels, threshold: float = -1, soft: bool = True):
        x = _require_signal(signal)
        if x.size == 0:
            raise InvalidSignalException("Signal cannot be empty for multi-level MODWT", "VAL_006")
        # forward() -> decomposeMutable() validates finiteness first
        bad = C.c_int64(-1)
        _raise_for_status(self.ctx, self.ctx.lib.vw_first_non_finite(self.ctx.handle, _vp(x), x.size, C.byref(bad)))
        return self.denoise_block(x.reshape(1, -1), levels, threshold, soft)[0]

    def denoiseBatch(self, signals, levels, threshold: float = -1, soft: bool = True):
        x = np.ascontiguousarray(np.asarray(signals, dtype=np.float64))
        return self.denoise_block(x, levels, threshold, soft)

    def extractLevel(self, signal, levels, targetLevel):
        """swt/VectorWaveSwtAdapter.java:326-348: keep one band, reconstruct."""
        result = self.forward(signal, levels)
        mask = (1 << levels) if targetLevel == 0 else (1 << (targetLevel - 1) if 1 <= targetLevel <= levels else 0)
        return self.modwtTransform.reconstruct_block(np.ascontiguousarray(result.block())[None], mask)[0]

    def cleanup(self):
        pass

    def getCacheStatistics(self):
        return {"filterCacheSize": 0, "parallelExecutorActive": False, "parallelThreshold": self.PARALLEL_THRESHOLD}


# ---------------------------------------------------------------------------
# denoising/WaveletDenoiser.java
# ---------------------------------------------------------------------------

class ThresholdMethod(enum.IntEnum):
    """denoising/WaveletDenoiser.java:588-617"""
    UNIVERSAL = N.VW_THRESHOLD_UNIVERSAL
    SURE = N.VW_THRESHOLD_SURE
    MINIMAX = N.VW_THRESHOLD_MINIMAX
    BAYES = N.VW_THRESHOLD_BAYES
    FIXED = N.VW_THRESHOLD_FIXED


class ThresholdType(enum.IntEnum):
    """denoising/WaveletDenoiser.java:622-634"""
    SOFT = 1
    HARD = 0


class WaveletDenoiser:
    """Mirror of denoising/WaveletDenoiser.java: MAD noise estimate from the finest detail band, threshold
    selection (UNIVERSAL, SURE, MINIMAX, BAYES), soft/hard thresholding, single- and multi-level forms.
    Everything between the input and the output signal runs on the GPU in one downcall per batch."""

    ThresholdMethod = ThresholdMethod
    ThresholdType = ThresholdType

    def __init__(self, wavelet, boundaryMode, ctx: Optional[N.Context] = None):
        # :79-90
        if wavelet is None:
            raise InvalidArgumentException("wavelet cannot be null", "VAL_001")
        if boundaryMode is None:
            raise InvalidArgumentException("boundaryMode cannot be null", "VAL_001")
        self.wavelet = _as_wavelet(wavelet)
        self.boundaryMode = BoundaryMode(boundaryMode)
        self._ctx = ctx

    @property
    def ctx(self) -> N.Context:
        if self._ctx is None:
            self._ctx = N.default_context()
        return self._ctx

    @staticmethod
    def forFinancialData(ctx: Optional[N.Context] = None):
        """:96-98: DB4, PERIODIC"""
        return WaveletDenoiser("db4", BoundaryMode.PERIODIC, ctx)

    def denoise_block(self, x: np.ndarray, levels: int, method, type=ThresholdType.SOFT, fixed_threshold: float = 0.0,
                      return_thresholds: bool = False):
        """x [B][N] -> denoised [B][N] (new batch surface).  levels == 0: the single-level form
        (denoise / denoiseFixed); levels >= 1: denoiseMultiLevel."""
        ctx = self.ctx
        b, n = x.shape
        w = self.wavelet
        h, g, hr, gr = _taps(w.dec_lo), _taps(w.dec_hi), _taps(w.rec_lo), _taps(w.rec_hi)
        nl = max(int(levels), 1)
        out = np.empty((b, n))
        thr = np.empty((b, nl))
        flags = N.VW_FLAG_SINGLE_LEVEL if levels == 0 else 0
        log_n = math.log(n) if n > 0 else 0.0
        st = ctx.lib.vw_wavelet_denoise_batch(ctx.handle, _vp(x), b, n, nl, _dp(h), h.size, _dp(g), g.size, _dp(hr),
                                              hr.size, _dp(gr), gr.size, int(self.boundaryMode), flags, int(method),
                                              1 if int(type) == int(ThresholdType.SOFT) else 0, float(fixed_threshold),
                                              log_n, _vp(out), _vp(thr))
        _raise_for_status(ctx, st)
        return (out, thr) if return_thresholds else out

    def _checked(self, signal) -> np.ndarray:
        # MODWTTransform.forward / MultiLevelMODWTTransform.decompose validate first (null, empty, finite)
        x = _require_signal(signal)
        if x.size == 0:
            raise InvalidSignalException("Signal cannot be empty", "VAL_006")
        bad = C.c_int64(-1)
        _raise_for_status(self.ctx, self.ctx.lib.vw_first_non_finite(self.ctx.handle, _vp(x), x.size, C.byref(bad)))
        return x

    def denoise(self, signal, method, type=ThresholdType.SOFT):
        """:108-139"""
        x = self._checked(signal)
        if int(method) == int(ThresholdMethod.FIXED):
            raise InvalidArgumentException("Fixed threshold method requires explicit threshold value", "CFG_005")
        return self.denoise_block(x.reshape(1, -1), 0, method, type)[0]

    def denoiseMultiLevel(self, signal, levels, method, type):
        """:152-167"""
        x = self._checked(signal)
        return self.denoise_block(x.reshape(1, -1), int(levels), method, type)[0]

    def denoiseFixed(self, signal, threshold, type):
        """:350-360"""
        x = self._checked(signal)
        return self.denoise_block(x.reshape(1, -1), 0, ThresholdMethod.FIXED, type, float(threshold))[0]

    def denoiseBatch(self, signals, method, type=ThresholdType.SOFT):
        return self.denoise_block(np.ascontiguousarray(np.asarray(signals, dtype=np.float64)), 0, method, type)

    def denoiseMultiLevelBatch(self, signals, levels, method, type=ThresholdType.SOFT):
        return self.denoise_block(np.ascontiguousarray(np.asarray(signals, dtype=np.float64)), int(levels), method, type)


# ---------------------------------------------------------------------------
# modwt/streaming/MODWTStreamingDenoiser.java
# ---------------------------------------------------------------------------

class NoiseEstimation(enum.IntEnum):
    """modwt/streaming/MODWTStreamingDenoiser.java:58-65"""
    MAD = N.VW_NOISE_MAD
    STD = N.VW_NOISE_STD
    FIXED = N.VW_NOISE_FIXED


class IllegalStateException(RuntimeError):
    """java.lang.IllegalStateException"""


class MODWTStreamingDenoiser:
    """Mirror of modwt/streaming/MODWTStreamingDenoiser.java: denoises a signal block by block while a ring of
    |detail| samples tracks the noise level.  `streams(k)` on the builder (not in the reference) runs k such
    denoisers in lock step, one block per stream per call, with their rings resident in GPU memory."""

    NoiseEstimation = NoiseEstimation

    class Builder:
        """:367-469, same defaults (:368-375) and the same argument checks"""

        def __init__(self):
            self._wavelet = "db4"
            self._boundaryMode = BoundaryMode.PERIODIC
            self._bufferSize = 256
            self._thresholdType = ThresholdType.SOFT
            self._thresholdMethod = ThresholdMethod.UNIVERSAL
            self._thresholdMultiplier = 1.0
            self._noiseEstimation = NoiseEstimation.MAD
            self._noiseWindowSize = 1024
            self._streams = 1
            self._ctx = None

        def wavelet(self, wavelet):
            if wavelet is None:
                raise InvalidArgumentException("Wavelet cannot be null")
            self._wavelet = wavelet
            return self

        def boundaryMode(self, boundaryMode):
            if boundaryMode is None:
                raise InvalidArgumentException("Boundary mode cannot be null")
            self._boundaryMode = BoundaryMode(boundaryMode)
            return self

        def bufferSize(self, bufferSize):
            if bufferSize <= 0:
                raise InvalidArgumentException("Buffer size must be positive")
            self._bufferSize = int(bufferSize)
            return self

        def thresholdType(self, thresholdType):
            if thresholdType is None:
                raise InvalidArgumentException("Threshold type cannot be null")
            self._thresholdType = ThresholdType(thresholdType)
            return self

        def thresholdMethod(self, thresholdMethod):
            if thresholdMethod is None:
                raise InvalidArgumentException("Threshold method cannot be null")
            self._thresholdMethod = ThresholdMethod(thresholdMethod)
            return self

        def thresholdMultiplier(self, thresholdMultiplier):
            if not thresholdMultiplier > 0:
                raise InvalidArgumentException("Threshold multiplier must be positive")
            self._thresholdMultiplier = float(thresholdMultiplier)
            return self

        def noiseEstimation(self, noiseEstimation):
            if noiseEstimation is None:
                raise InvalidArgumentException("Noise estimation cannot be null")
            self._noiseEstimation = NoiseEstimation(noiseEstimation)
            return self

        def noiseWindowSize(self, noiseWindowSize):
            if noiseWindowSize <= 0:
                raise InvalidArgumentException("Noise window size must be positive")
            self._noiseWindowSize = int(noiseWindowSize)
            return self

        def streams(self, streams):
            if streams <= 0:
                raise InvalidArgumentException("Stream count must be positive")
            self._streams = int(streams)
            return self

        def context(self, ctx):
            self._ctx = ctx
            return self

        def build(self):
            return MODWTStreamingDenoiser(self)

    def __init__(self, builder: "MODWTStreamingDenoiser.Builder"):
        # :67-83
        self.wavelet = _as_wavelet(builder._wavelet)
        self.boundaryMode = builder._boundaryMode
        self.bufferSize = builder._bufferSize
        self.thresholdType = builder._thresholdType
        self.thresholdMethod = builder._thresholdMethod
        self.thresholdMultiplier = builder._thresholdMultiplier
        self.noiseEstimation = builder._noiseEstimation
        self.noiseWindowSize = builder._noiseWindowSize
        self.streams = builder._streams
        self._ctx = builder._ctx if builder._ctx is not None else N.default_context()
        self._closed = False
        self._final_samples = 0
        self._subscribers = []
        self._state = C.c_void_p()
        st = self._ctx.lib.vw_stream_state_create(self._ctx.handle, self.streams, self.noiseWindowSize, C.byref(self._state))
        _raise_for_status(self._ctx, st)

    def denoise(self, samples):
        """:93-127.  One block: a 1-D array for a single stream, or [streams][n] for lock-step streams."""
        if self._closed:
            raise IllegalStateException("Denoiser is closed")
        if samples is None:
            raise InvalidArgumentException("Samples cannot be null or empty")
        x = np.ascontiguousarray(np.asarray(samples, dtype=np.float64))
        if x.size == 0:
            raise InvalidArgumentException("Samples cannot be null or empty")
        single = x.ndim == 1
        if single:
            x = x.reshape(1, -1)
        if x.ndim != 2 or x.shape[0] != self.streams:
            raise InvalidArgumentException("expected one block per stream: %d rows" % self.streams)
        ctx = self._ctx
        bad = C.c_int64(-1)
        _raise_for_status(ctx, ctx.lib.vw_first_non_finite(ctx.handle, _vp(x), x.size, C.byref(bad)))
        b, n = x.shape
        w = self.wavelet
        h, g, hr, gr = _taps(w.dec_lo), _taps(w.dec_hi), _taps(w.rec_lo), _taps(w.rec_hi)
        out = np.empty((b, n))
        log_n = math.log(n)
        st = ctx.lib.vw_streaming_denoise_batch(ctx.handle, self._state, _vp(x), b, n, _dp(h), h.size, _dp(g), g.size,
                                                _dp(hr), hr.size, _dp(gr), gr.size, int(self.boundaryMode), 0,
                                                int(self.thresholdMethod),
                                                1 if int(self.thresholdType) == int(ThresholdType.SOFT) else 0,
                                                self.thresholdMultiplier, int(self.noiseEstimation), log_n,
                                                math.sqrt(2.0 * log_n), _vp(out), None)
        _raise_for_status(ctx, st)
        result = out[0] if single else out
        for s in self._subscribers:
            item = result.copy()  # :122-124: subscribers receive a clone
            (s.onNext if hasattr(s, "onNext") else s)(item)
        return result

    def getEstimatedNoiseLevel(self):
        """:332-334; an array with one level per stream when there are several"""
        lv = np.empty(self.streams)
        _raise_for_status(self._ctx, self._ctx.lib.vw_stream_state_noise_levels(self._ctx.handle, self._state, _vp(lv)))
        return float(lv[0]) if self.streams == 1 else lv

    def getNoiseWindow(self):
        """Copy of the |detail| rings [streams][noiseWindowSize] and the ring position (for tests)."""
        win = np.empty((self.streams, self.noiseWindowSize))
        idx = C.c_int64(0)
        _raise_for_status(self._ctx, self._ctx.lib.vw_stream_state_window(self._ctx.handle, self._state, _vp(win), C.byref(idx)))
        return win, int(idx.value)

    def getSamplesProcessed(self):
        """:341-343"""
        if not self._state:
            return self._final_samples
        return int(self._ctx.lib.vw_stream_state_samples_processed(self._state))

    def subscribe(self, subscriber):
        """:346-348: `subscriber` is a callable or has onNext(array)"""
        self._subscribers.append(subscriber)

    def close(self):
        """:351-355"""
        if not self._closed:
            self._closed = True
            for s in self._subscribers:
                if hasattr(s, "onComplete"):
                    s.onComplete()
            self._subscribers = []
            if self._state:
                self._final_samples = int(self._ctx.lib.vw_stream_state_samples_processed(self._state))
                self._ctx.lib.vw_stream_state_destroy(self._ctx.handle, self._state)
                self._state = C.c_void_p()

    def isClosed(self):
        return self._closed

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class InvalidStateException(WaveletTransformException):
    """exception/InvalidStateException.java"""


class StreamingStatistics:
    """MODWTStreamingTransformImpl.StreamingStatisticsImpl (modwt/streaming/MODWTStreamingTransformImpl.java:312-391)"""

    def __init__(self):
        self.reset()

    def reset(self):
        import time
        self._samples = 0
        self._blocks = 0
        self._total_ns = 0
        self._max_ns = 0
        self._min_ns = None
        self._start = time.perf_counter_ns()

    def _record_blocks(self, blocks, total_ns):
        if blocks <= 0:
            return
        per = total_ns // blocks
        self._blocks += blocks
        self._total_ns += total_ns
        self._max_ns = max(self._max_ns, per)
        self._min_ns = per if self._min_ns is None else min(self._min_ns, per)

    def getSamplesProcessed(self):
        return self._samples

    def getBlocksProcessed(self):
        return self._blocks

    def getAverageProcessingTimeNanos(self):
        return self._total_ns // self._blocks if self._blocks > 0 else 0

    def getMaxProcessingTimeNanos(self):
        return self._max_ns

    def getMinProcessingTimeNanos(self):
        return 0 if self._min_ns is None else self._min_ns

    def getThroughputSamplesPerSecond(self):
        import time
        elapsed = (time.perf_counter_ns() - self._start) / 1e9
        return self._samples / elapsed if elapsed > 0 else 0.0


class MODWTStreamingTransform:
    """Mirror of modwt/streaming/MODWTStreamingTransform.java + MODWTStreamingTransformImpl.java: samples are
    buffered, and every time bufferSize unconsumed samples are present the last bufferSize samples are transformed
    (single level) and published; consecutive windows overlap by filter length - 1 samples.  All windows that one
    process() call completes go to the GPU as ONE batch of rows; `streams` > 1 (not in the reference) feeds that
    many lock-step streams per call, data [streams][len], and publishes coefficient blocks [streams][2][bufferSize]
    (detail, approximation) instead of MODWTResult objects."""

    @staticmethod
    def create(wavelet, boundaryMode, bufferSize=256, streams=1, ctx: Optional[N.Context] = None):
        """MODWTStreamingTransform.create (:66-81)"""
        return MODWTStreamingTransform(wavelet, boundaryMode, bufferSize, streams, ctx)

    @staticmethod
    def createMultiLevel(wavelet, boundaryMode, bufferSize, levels, streams=1, ctx: Optional[N.Context] = None):
        """MODWTStreamingTransform.createMultiLevel (:92-95)"""
        return MultiLevelMODWTStreamingTransform(wavelet, boundaryMode, bufferSize, levels, streams, ctx)

    def __init__(self, wavelet, boundaryMode, bufferSize, streams=1, ctx: Optional[N.Context] = None):
        # MODWTStreamingTransformImpl.java:79-143
        if wavelet is None:
            raise InvalidArgumentException("Wavelet cannot be null")
        if boundaryMode is None:
            raise InvalidArgumentException("Boundary mode cannot be null")
        if bufferSize <= 0:
            raise InvalidArgumentException("Buffer size must be positive, got: %d" % bufferSize)
        self.wavelet = _as_wavelet(wavelet)
        self.filterLength = len(self.wavelet.dec_lo)
        self.overlapSize = self.filterLength - 1
        if bufferSize > 2 ** 31 - 1 - self.overlapSize:
            raise InvalidArgumentException("Buffer size too large, would cause integer overflow: bufferSize=%d, overlapSize=%d"
                                           % (bufferSize, self.overlapSize))
        mb = (bufferSize + self.overlapSize) * 8 // (1024 * 1024)
        if (bufferSize + self.overlapSize) * 8 > 100 * 1024 * 1024:
            raise InvalidArgumentException("Buffer size too large, would require %dMB. Maximum allowed is 100MB" % mb)
        if bufferSize < self.filterLength:
            raise InvalidArgumentException("Buffer size must be at least as large as filter length: bufferSize=%d, filterLength=%d"
                                           % (bufferSize, self.filterLength))
        if streams <= 0:
            raise InvalidArgumentException("Stream count must be positive")
        self.boundaryMode = BoundaryMode(boundaryMode)
        self.bufferSize = int(bufferSize)
        self.streams = int(streams)
        self._transform = MODWTTransform(self.wavelet, self.boundaryMode, ctx)
        self._history = np.empty((self.streams, 0))  # the last samples seen (at most bufferSize): the circular buffer
        self._samples_in_buffer = 0                  # samplesInBuffer of the reference
        self._closed = False
        self._subscribers = []
        self._statistics = StreamingStatistics()

    # -- Flow.Publisher ------------------------------------------------------
    def subscribe(self, subscriber):
        """`subscriber` is a callable or has onNext(item) (and optionally onComplete())"""
        self._subscribers.append(subscriber)

    def _submit(self, blocks):
        # blocks [streams][k][2][bufferSize]
        for w in range(blocks.shape[1]):
            if self.streams == 1:
                item = MODWTResult(blocks[0, w, 1], blocks[0, w, 0], _validated=True)
            else:
                item = blocks[:, w].copy()
            for s in self._subscribers:
                (s.onNext if hasattr(s, "onNext") else s)(item)

    def _transform_windows(self, windows):
        """windows [streams][k][bufferSize] (a strided view is fine) -> published blocks; one GPU batch."""
        import time
        t0 = time.perf_counter_ns()
        k = windows.shape[1]
        rows = np.ascontiguousarray(windows).reshape(self.streams * k, self.bufferSize)
        coeffs = self._transform.forward_block(rows, True).reshape(self.streams, k, 2, self.bufferSize)
        self._submit(coeffs)
        self._statistics._record_blocks(k, time.perf_counter_ns() - t0)

    # -- MODWTStreamingTransform ----------------------------------------------
    def process(self, data):
        """:145-158 with processSampleInternal/processBuffer (:161-203) applied to the whole array at once"""
        if self._closed:
            raise InvalidStateException("Transform is closed")
        if data is None:
            raise InvalidSignalException("Data cannot be null or empty")
        d = np.asarray(data, dtype=np.float64)
        if d.size == 0:
            raise InvalidSignalException("Data cannot be null or empty")
        if d.ndim == 1:
            d = d.reshape(1, -1)
        if d.ndim != 2 or d.shape[0] != self.streams:
            raise InvalidArgumentException("expected data for %d stream(s)" % self.streams)
        self._feed(d)

    def processSample(self, sample):
        """:160-167"""
        if self._closed:
            raise InvalidStateException("Transform is closed")
        self._feed(np.asarray(sample, dtype=np.float64).reshape(self.streams, 1))

    def _after_window(self, count):
        """samplesInBuffer after a window went out (MODWTStreamingTransformImpl.java:216-226)"""
        return count - (self.bufferSize - self.overlapSize)

    def _feed(self, d):
        """The reference's sample loop (:154-180) for a whole array: a window goes out whenever samplesInBuffer reaches
        bufferSize, and holds the last bufferSize samples.  All windows that the array completes are transformed as
        one batch.  A window that holds a NaN or an infinity fails like MODWTTransform.forward fails in the
        reference (:194): the windows before it are published, the sample that completed it is stored but neither
        counted nor consumed, the rest of the array is dropped, and the exception propagates."""
        m = d.shape[1]
        seen = self._history.shape[1]
        hist = np.concatenate([self._history, d], axis=1)
        ends, counts = [], []  # data index of the sample that completes each window; samplesInBuffer at that moment
        count, i = self._samples_in_buffer, -1
        while True:
            need = max(1, self.bufferSize - count)
            if i + need >= m:
                break
            i += need
            ends.append(i)
            counts.append(count + need)
            count = self._after_window(count + need)
        bad = None
        if ends:
            nonfinite = np.concatenate([[0], np.cumsum(~np.isfinite(hist).all(axis=0))])
            for k, e in enumerate(ends):
                if nonfinite[seen + e + 1] - nonfinite[seen + e + 1 - self.bufferSize] > 0:
                    bad = k
                    break
        good = ends if bad is None else ends[:bad]
        starts = np.asarray(good, dtype=np.int64) + (seen + 1 - self.bufferSize)
        all_windows = np.lib.stride_tricks.sliding_window_view(hist, self.bufferSize, axis=1) if ends else None
        if bad is None:
            self._samples_in_buffer = count + (m - 1 - i)
            self._history = hist[:, -self.bufferSize:].copy()
            self._statistics._samples += m
            if good:
                self._transform_windows(all_windows[:, starts])
            return
        failed_at = ends[bad]
        self._samples_in_buffer = counts[bad]
        self._history = hist[:, : seen + failed_at + 1][:, -self.bufferSize:].copy()
        self._statistics._samples += failed_at
        if good:
            self._transform_windows(all_windows[:, starts])
        # the failing window goes through the same validating call and raises what forward() raises
        self._transform_windows(all_windows[:, [seen + failed_at + 1 - self.bufferSize]])
        raise InvalidSignalException("window contains non-finite values", "VAL_003")

    def _drain(self):
        level = self._samples_in_buffer
        if level > 0:
            if level > self.bufferSize:
                # only after a failed window: the reference's copy loop runs past finalBuffer (:240-242)
                raise IndexError("Index %d out of bounds for length %d" % (self.bufferSize, self.bufferSize))
            final = np.zeros((self.streams, 1, self.bufferSize))
            final[:, 0, :level] = self._history[:, self._history.shape[1] - level:]
            self._history = np.empty((self.streams, 0))
            self._samples_in_buffer = 0
            self._transform_windows(final)

    def flush(self):
        """:205-231: the unconsumed samples, zero-padded to bufferSize, make one last block"""
        if self._closed:
            raise InvalidStateException("Transform is closed")
        self._drain()

    def getStatistics(self):
        return self._statistics

    def reset(self):
        """:238-250"""
        if self._closed:
            raise InvalidStateException("Transform is closed")
        self._history = np.empty((self.streams, 0))
        self._samples_in_buffer = 0
        self._statistics.reset()

    def getBufferLevel(self):
        return int(self._samples_in_buffer)

    def isClosed(self):
        return self._closed

    def close(self):
        """:262-291: the remaining samples are processed, then the publisher completes"""
        if not self._closed:
            self._closed = True
            self._drain()
            for s in self._subscribers:
                if hasattr(s, "onComplete"):
                    s.onComplete()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


class MultiLevelMODWTStreamingTransform(MODWTStreamingTransform):
    """Mirror of modwt/streaming/MultiLevelMODWTStreamingTransform.java: non-overlapping blocks of bufferSize
    samples, each decomposed to `levels` levels.  Per block the reference publishes one MODWTResult per level
    (details of that level; the approximation with the last level, an empty array before it, :146-152); the blocks
    that one process() call completes are decomposed in ONE GPU batch.  With `streams` > 1 each published item is
    the coefficient block [streams][levels+1][bufferSize] of one window instead."""

    def __init__(self, wavelet, boundaryMode, bufferSize, levels, streams=1, ctx: Optional[N.Context] = None):
        # :60-91
        if wavelet is None:
            raise InvalidArgumentException("Wavelet cannot be null")
        if boundaryMode is None:
            raise InvalidArgumentException("Boundary mode cannot be null")
        if bufferSize <= 0:
            raise InvalidArgumentException("Buffer size must be positive, got: %d" % bufferSize)
        if levels < 1:
            raise InvalidArgumentException("Levels must be at least 1, got: %d" % levels)
        if streams <= 0:
            raise InvalidArgumentException("Stream count must be positive")
        self.wavelet = _as_wavelet(wavelet)
        self.boundaryMode = BoundaryMode(boundaryMode)
        self.bufferSize = int(bufferSize)
        self.levels = int(levels)
        self.streams = int(streams)
        self.filterLength = len(self.wavelet.dec_lo)
        self.overlapSize = 0  # blocks do not overlap here (:161 resets the sample count to zero)
        self._ml = MultiLevelMODWTTransform(self.wavelet, self.boundaryMode, ctx)
        self._history = np.empty((self.streams, 0))
        self._samples_in_buffer = 0
        self._closed = False
        self._subscribers = []
        self._statistics = StreamingStatistics()

    def _submit(self, blocks):
        # blocks [streams][k][levels+1][bufferSize]
        empty = np.empty(0)
        for w in range(blocks.shape[1]):
            if self.streams == 1:
                items = [_LevelResult(blocks[0, w, self.levels] if level == self.levels else empty, blocks[0, w, level - 1])
                         for level in range(1, self.levels + 1)]
            else:
                items = [blocks[:, w].copy()]
            for item in items:
                for s in self._subscribers:
                    (s.onNext if hasattr(s, "onNext") else s)(item)

    def _transform_windows(self, windows):
        import time
        t0 = time.perf_counter_ns()
        k = windows.shape[1]
        rows = np.ascontiguousarray(windows).reshape(self.streams * k, self.bufferSize)
        coeffs = self._ml.decompose_block(rows, self.levels, True).reshape(self.streams, k, self.levels + 1, self.bufferSize)
        self._submit(coeffs)
        self._statistics._record_blocks(k, time.perf_counter_ns() - t0)

    def _after_window(self, count):
        """:166: the first-level counter restarts at zero after every block"""
        return 0

    def flush(self):
        """:166-196 (no closed check in the reference)"""
        self._drain()


class _LevelResult:
    """MultiLevelMODWTStreamingTransform.MODWTResultWrapper (:262-310): one level of one block"""

    def __init__(self, approx, details):
        self._a, self._d = np.array(approx, dtype=np.float64), np.array(details, dtype=np.float64)

    def approximationCoeffs(self):
        return self._a.copy()

    def detailCoeffs(self):
        return self._d.copy()

    def getSignalLength(self):
        return int(self._a.size)

    def isValid(self):
        return bool(np.isfinite(self._a).all() and np.isfinite(self._d).all())


# ---------------------------------------------------------------------------
# swt/SWTResult.java
# ---------------------------------------------------------------------------

class SWTResult:
    """Mirror of swt/SWTResult.java: immutable approximation + per-level details (level index 0-based, as in the
    reference), lazily cached energies, and the sparse form whose compaction runs on the GPU."""

    def __init__(self, approximation, details, levels, ctx: Optional[N.Context] = None):
        # :43-60
        if approximation is None:
            raise TypeError("Approximation cannot be null")
        if details is None:
            raise TypeError("Details cannot be null")
        if len(details) != levels:
            raise ValueError("Details array length must match levels")
        self._a = np.array(approximation, dtype=np.float64)
        self._levels = int(levels)
        self._n = int(self._a.size)
        for d in details:
            if d is None or len(d) != self._n:
                raise ValueError("All detail levels must have same length as approximation")
        self._d = np.array(details, dtype=np.float64).reshape(self._levels, self._n)
        self._ctx = ctx
        self._detail_energies = [None] * self._levels
        self._total_energy = None

    def getApproximation(self):
        return self._a.copy()

    def getDetail(self, level):
        if level < 0 or level >= self._levels:
            raise ValueError("Level must be between 0 and %d" % (self._levels - 1))
        return self._d[level].copy()

    def getAllDetails(self):
        return self._d.copy()

    def getLevels(self):
        return self._levels

    def getSignalLength(self):
        return self._n

    def getDetailEnergy(self, level):
        # :124-147, summed in index order like the Java loop
        if level < 0 or level >= self._levels:
            raise ValueError("Level must be between 0 and %d" % (self._levels - 1))
        if self._detail_energies[level] is None:
            self._detail_energies[level] = MultiLevelMODWTResult._energy(self._d[level])
        return self._detail_energies[level]

    def getApproximationEnergy(self):
        return MultiLevelMODWTResult._energy(self._a)

    def getTotalEnergy(self):
        if self._total_energy is None:
            e = self.getApproximationEnergy()
            for level in range(self._levels):
                e += self.getDetailEnergy(level)
            self._total_energy = e
        return self._total_energy

    def toSparse(self, threshold):
        """:190-192"""
        return SparseSWTResult(self, threshold)

    def __eq__(self, other):
        return (isinstance(other, SWTResult) and self._levels == other._levels and self._n == other._n
                and _java_arrays_equal(self._a, other._a) and _java_arrays_equal(self._d, other._d))

    def __hash__(self):
        return hash((self._levels, self._n, self._a.tobytes(), self._d.tobytes()))

    def __repr__(self):
        return "SWTResult[levels=%d, signalLength=%d, totalEnergy=%.6f]" % (self._levels, self._n, self.getTotalEnergy())


def _java_arrays_equal(a, b):
    """Arrays.equals(double[], double[]): element-wise Double.doubleToLongBits equality (NaN equals NaN, 0.0 != -0.0)"""
    if a.shape != b.shape:
        return False
    an, bn = np.isnan(a), np.isnan(b)
    return bool(np.array_equal(an, bn) and np.array_equal(a[~an].view(np.uint64), b[~bn].view(np.uint64)))


class SparseSWTResult:
    """SWTResult.SparseSWTResult (:198-257): dense approximation, per-level (index, value) lists of the detail
    entries with |v| > threshold.  The compaction (count pass, prefix sum, ordered fill) and the expansion back to
    dense bands are GPU kernels."""

    def __init__(self, full: SWTResult, threshold):
        self._ctx = full._ctx if full._ctx is not None else N.default_context()
        self._n, self._levels, self._threshold = full._n, full._levels, float(threshold)
        self._a = full._a.copy()
        ctx = self._ctx
        block = np.empty((1, self._levels + 1, self._n))
        block[0, :self._levels] = full._d
        block[0, self._levels] = full._a
        counts = np.zeros(self._levels, dtype=np.int64)
        if self._n > 0:
            _raise_for_status(ctx, ctx.lib.vw_swt_sparsify_batch(ctx.handle, _vp(block), 1, self._n, self._levels, self._threshold,
                                                                 _vp(counts), None, None, None))
        self._offsets = np.zeros(self._levels + 1, dtype=np.int64)
        np.cumsum(counts, out=self._offsets[1:])
        total = int(self._offsets[-1])
        self._indices = np.empty(total, dtype=np.int32)
        self._values = np.empty(total)
        if total > 0:
            _raise_for_status(ctx, ctx.lib.vw_swt_sparsify_batch(ctx.handle, _vp(block), 1, self._n, self._levels, self._threshold,
                                                                 None, _vp(self._offsets), _vp(self._indices), _vp(self._values)))

    def getNonZeroCount(self, level):
        return int(self._offsets[level + 1] - self._offsets[level])

    def getSparseDetail(self, level):
        """(indices, values) of one level (not in the reference's public surface; for inspection and tests)"""
        lo, hi = int(self._offsets[level]), int(self._offsets[level + 1])
        return self._indices[lo:hi].copy(), self._values[lo:hi].copy()

    def toFull(self):
        """:225-231"""
        ctx = self._ctx
        block = np.empty((1, self._levels + 1, self._n))
        block[0, self._levels] = self._a
        if self._n > 0:
            _raise_for_status(ctx, ctx.lib.vw_swt_densify_batch(ctx.handle, _vp(self._offsets), _vp(self._indices), _vp(self._values),
                                                                1, self._n, self._levels, _vp(block)))
        return SWTResult(self._a, block[0, :self._levels], self._levels, ctx)

    def getCompressionRatio(self):
        """:238-247"""
        original = self._n * (self._levels + 1)
        compressed = self._n + 2 * int(self._offsets[-1])
        return original / compressed

    def getThreshold(self):
        return self._threshold


# ---------------------------------------------------------------------------
# modwt/BatchSIMDMODWT.java and modwt/ParallelMultiLevelMODWT.java
# ---------------------------------------------------------------------------

class BatchSIMDMODWT:
    """Structure-of-arrays batch surface (modwt/BatchSIMDMODWT.java): signals stored sample-major,
    soa[t * batchSize + b].  Static methods, like the reference."""

    @staticmethod
    def batchMODWTSoA(soaSignals, soaApprox, soaDetail, wavelet, batchSize, signalLength, ctx: Optional[N.Context] = None):
        """:64-81: single-level PERIODIC MODWT of the whole batch; fills soaApprox / soaDetail in place."""
        ctx = ctx or N.default_context()
        w = _as_wavelet(wavelet)
        x = _require_signal(soaSignals, "soaSignals")
        if soaApprox is None or soaDetail is None:
            raise TypeError("output arrays cannot be null")
        cnt = int(batchSize) * int(signalLength)
        if x.size < cnt or soaApprox.size < cnt or soaDetail.size < cnt:
            raise IndexError("SoA arrays shorter than batchSize * signalLength")
        h, g = _taps(w.dec_lo), _taps(w.dec_hi)
        a = np.empty(cnt)
        d = np.empty(cnt)
        st = ctx.lib.vw_batch_modwt_soa(ctx.handle, _vp(x), int(batchSize), int(signalLength), _dp(h), h.size, _dp(g),
                                        g.size, 1 if w.name == "haar" else 0, _vp(a), _vp(d))
        _raise_for_status(ctx, st)
        soaApprox[:cnt] = a
        soaDetail[:cnt] = d

    @staticmethod
    def convertToSoA(signals, soaOutput, ctx: Optional[N.Context] = None):
        """:282-291: soaOutput[t * batchSize + b] = signals[b][t]"""
        ctx = ctx or N.default_context()
        x = np.ascontiguousarray(np.asarray(signals, dtype=np.float64))
        b, n = x.shape
        out = np.empty(b * n)
        _raise_for_status(ctx, ctx.lib.vw_transpose(ctx.handle, _vp(x), b, n, _vp(out)))
        soaOutput[: b * n] = out

    @staticmethod
    def convertFromSoA(soaData, output, ctx: Optional[N.Context] = None):
        """:299-308: output[b][t] = soaData[t * batchSize + b]"""
        ctx = ctx or N.default_context()
        b, n = len(output), len(output[0])
        s = np.ascontiguousarray(np.asarray(soaData, dtype=np.float64)[: b * n])
        out = np.empty((b, n))
        _raise_for_status(ctx, ctx.lib.vw_transpose(ctx.handle, _vp(s), n, b, _vp(out)))
        for i in range(b):
            output[i][:] = out[i]


class ParallelMultiLevelMODWT:
    """modwt/ParallelMultiLevelMODWT.java: the same cascade as MultiLevelMODWTTransform.decompose with its own
    level limit (:218-248) and with every non-PERIODIC mode treated as zero padding (:100-131).  The
    executor of the reference has no counterpart: the levels of one signal run fused in one kernel."""

    def __init__(self, executor=None, ctx: Optional[N.Context] = None):
        self._ctx = ctx

    @staticmethod
    def _max_levels(n, filter_len):
        if n <= filter_len:
            return 0
        level = 1
        while level < 10 and (filter_len - 1) * (1 << (level - 1)) + 1 <= n:
            level += 1
        return level - 1

    def decompose(self, signal, wavelet, mode, levels) -> MultiLevelMODWTResult:
        x = _require_signal(signal)
        ctx = self._ctx or N.default_context()
        # :62-66: finiteness first, then emptiness
        bad = C.c_int64(-1)
        if x.size:
            _raise_for_status(ctx, ctx.lib.vw_first_non_finite(ctx.handle, _vp(x), x.size, C.byref(bad)))
        if x.size == 0:
            raise InvalidSignalException("Signal cannot be empty", "VAL_006")
        w = _as_wavelet(wavelet)
        maxl = self._max_levels(x.size, len(w.dec_lo))
        if levels < 1 or levels > maxl:
            raise InvalidArgumentException("Invalid number of levels: %d. Must be between 1 and %d" % (levels, maxl), "CFG_004")
        eff = BoundaryMode.PERIODIC if BoundaryMode(mode) == BoundaryMode.PERIODIC else BoundaryMode.ZERO_PADDING
        t = MultiLevelMODWTTransform(w, eff, ctx)
        return MultiLevelMODWTResult(t.decompose_block(x.reshape(1, -1), int(levels), check_finite=False)[0])
