"""CPU: the oracle's restatement of modwt/streaming/MODWTStreamingDenoiser.java against a line-by-line
Python re-reading of the same Java methods and against the properties the reference's own tests assert
(src/test/.../modwt/streaming/MODWTStreamingDenoiserNoiseWindowTest.java, ...ThresholdMultiplierTest.java,
...OverflowTest.java).  Those tests hold no numeric golden outputs, so this part of the oracle is pinned by
formula only."""
import math

import numpy as np
import pytest

from oracle import oracle as o
from vectorwave_amd.wavelets import get_wavelet


def _py_median(v):
    a = sorted(v)
    n = len(a)
    return a[n // 2] if n % 2 == 1 else (a[n // 2 - 1] + a[n // 2]) / 2.0


def _py_mad(v):
    if not any(x != 0.0 for x in v):
        return 0.0
    m = _py_median(v)
    return _py_median([abs(x - m) for x in v])


def _py_std(v):
    n = len(v)
    s = 0.0
    for x in v:
        s += x
    mean = s / n
    q = 0.0
    for x in v:
        q += (x - mean) * (x - mean)
    return math.sqrt(q / (n - 1))


def _py_window_update(details, window, index):
    """MODWTStreamingDenoiser.java:140-191, statement by statement"""
    n, w = len(details), len(window)
    if n <= w:
        for d in details:
            window[index] = abs(d)
            index = (index + 1) % w
        return index
    strata = min(w, 10)
    per, extra, size = w // strata, w % strata, n // strata
    taken = 0
    for s in range(strata):
        start = s * size
        end = n if s == strata - 1 else (s + 1) * size
        take = per + (1 if s < extra else 0)
        if take > 0:
            length = end - start
            step = max(1, length // take)
            for i in range(take):
                if taken >= w:
                    break
                idx = start + (i * step) % length
                if idx < n:
                    window[index] = abs(details[idx])
                    index = (index + 1) % w
                    taken += 1
    remaining = w - taken
    if remaining > 0:
        for i in range(max(0, n - remaining), n):
            if taken >= w:
                break
            window[index] = abs(details[i])
            index = (index + 1) % w
            taken += 1
    return index


def _blocks(streams, n, count, seed, noise=0.1):
    rng = np.random.default_rng(seed)
    t = np.arange(count * n).reshape(count, n)
    clean = np.sin(2 * np.pi * t / 32.0)
    return clean[:, None, :] + noise * rng.standard_normal((count, streams, n))


@pytest.mark.parametrize("n", [1, 2, 7, 100, 101])
def test_mad_and_std_match_the_formulas(n):
    rng = np.random.default_rng(n)
    v = np.abs(rng.standard_normal(n))
    assert o.stream_mad(v) == _py_mad(v.tolist())
    if n >= 2:
        assert o.stream_std(v) == _py_std(v.tolist())
    else:
        assert o.stream_std(v) == 0.0
    assert o.stream_mad(np.zeros(n)) == 0.0


@pytest.mark.parametrize("n,w", [(64, 1024), (256, 1024), (480, 100), (1000, 7), (4096, 512), (300, 300), (301, 300),
                                 (2048, 1024), (50, 3), (37, 10), (45, 11)])
@pytest.mark.parametrize("est", [o.NOISE_MAD, o.NOISE_STD])
def test_window_update_follows_the_java_loops(n, w, est):
    wv = get_wavelet("db4")
    state = o.StreamingDenoiserState(2, w)
    win = [[0.0] * w, [0.0] * w]
    index = 0
    xs = _blocks(2, n, 4, seed=n + w)
    for blk in xs:
        st, _, _ = o.streaming_denoise_batch(state, blk, wv, estimation=est)
        assert st == 0
        det = o.modwt_forward_batch(blk, wv)[:, 0, :]
        for r in range(2):
            nxt = _py_window_update(det[r].tolist(), win[r], index)
        index = nxt
        assert state.index == index
        assert np.array_equal(state.window, np.array(win))
        for r in range(2):
            want = _py_mad(win[r]) / 0.6745 if est == o.NOISE_MAD else _py_std(win[r])
            assert state.level[r] == want
    assert state.samples_processed == 4 * n


def test_unit_multiplier_is_the_plain_denoiser():
    # :108-110: with multiplier 1 the block goes through WaveletDenoiser.denoise unchanged
    wv = get_wavelet("db4")
    state = o.StreamingDenoiserState(3, 256)
    for blk in _blocks(3, 200, 3, seed=3):
        for method in (o.UNIVERSAL, o.SURE, o.MINIMAX, o.BAYES):
            st, got, thr = o.streaming_denoise_batch(state, blk, wv, method=method)
            st2, want, want_thr = o.wavelet_denoise_batch(blk, wv, 0, method)
            assert st == 0 and st2 == 0
            assert np.array_equal(got, want) and np.array_equal(thr, want_thr[:, 0])


@pytest.mark.parametrize("method,factor", [(o.UNIVERSAL, 1.0), (o.SURE, 0.8)])
def test_multiplier_path_thresholds(method, factor):
    # :277-324 with an estimated level: threshold = level * sqrt(2 ln n) [* 0.8] * multiplier
    wv = get_wavelet("db4")
    n = 128
    state = o.StreamingDenoiserState(2, 512)
    for blk in _blocks(2, n, 3, seed=9):
        st, got, thr = o.streaming_denoise_batch(state, blk, wv, method=method, multiplier=1.5)
        assert st == 0
        det = o.modwt_forward_batch(blk, wv)[:, 0, :]
        for r in range(2):
            # a ring that is still mostly zeros has MAD 0: the block's own details give sigma then (:282-293)
            sigma = state.level[r] if state.level[r] > 0 else _py_mad(np.abs(det[r]).tolist()) / 0.6745
            base = sigma * math.sqrt(2.0 * math.log(n))
            if method == o.SURE:
                base = base * 0.8
            assert thr[r] == base * 1.5
        st2, want, _ = o.wavelet_denoise_batch(blk[:1], wv, 0, o.FIXED, fixed_threshold=float(thr[0]))
        assert np.array_equal(got[0], want[0])


def test_multiplier_path_minimax_fixed_and_fixed_estimation():
    wv = get_wavelet("haar")
    for n, want_fn in ((32, lambda s, ln: 0.0), (64, lambda s, ln: s * (0.3936 + 0.1829 * ln)),
                       (100, lambda s, ln: s * (0.4745 + 0.1148 * ln))):
        state = o.StreamingDenoiserState(1, 64)
        blk = _blocks(1, n, 1, seed=n)[0]
        st, _, thr = o.streaming_denoise_batch(state, blk, wv, method=o.MINIMAX, multiplier=2.0)
        assert st == 0 and thr[0] == want_fn(state.level[0], math.log(n)) * 2.0
    # FIXED method: the level itself; FIXED estimation: sigma from this block's own details, state untouched
    state = o.StreamingDenoiserState(1, 64)
    blk = _blocks(1, 100, 1, seed=4)[0]
    st, _, thr = o.streaming_denoise_batch(state, blk, wv, method=o.FIXED, multiplier=0.5, estimation=o.NOISE_FIXED)
    d = o.modwt_forward_batch(blk, wv)[0, 0]
    assert st == 0 and thr[0] == _py_mad(np.abs(d).tolist()) / 0.6745 * 0.5
    assert state.index == 0 and not state.window.any() and state.level[0] == 0.0


def test_methods_the_reference_rejects():
    wv = get_wavelet("db4")
    state = o.StreamingDenoiserState(1, 64)
    blk = _blocks(1, 50, 1, seed=1)[0]
    assert o.streaming_denoise_batch(state, blk, wv, method=o.FIXED)[0] == 105          # denoise(..., FIXED, ...) throws
    assert o.streaming_denoise_batch(state, blk, wv, method=o.BAYES, multiplier=2.0)[0] == 105  # switch default branch


def test_reference_test_properties():
    # NoiseWindowTest.testNoiseEstimationConsistency: two denoisers fed the same blocks agree
    wv = get_wavelet("db4")
    xs = _blocks(1, 256, 5, seed=2, noise=0.05)
    s1, s2 = o.StreamingDenoiserState(1, 128), o.StreamingDenoiserState(1, 128)
    for blk in xs:
        a = o.streaming_denoise_batch(s1, blk, wv, multiplier=1.2)[1]
        b = o.streaming_denoise_batch(s2, blk, wv, multiplier=1.2)[1]
        assert np.array_equal(a, b)
    # OverflowTest.testNoiseEstimationConsistency: level positive and of the order of the injected noise
    assert 0 < s1.level[0] < 0.2
    # ThresholdMultiplierTest.testThresholdMultiplierEffect: larger multipliers remove more energy
    energies = []
    for mult in (0.25, 0.5, 3.0):
        st = o.StreamingDenoiserState(1, 128)
        y = o.streaming_denoise_batch(st, xs[0], wv, multiplier=mult)[1]
        energies.append(float((y * y).sum()))
    assert energies[0] > energies[1] > energies[2]  # 3.0 removes the whole detail band


# ---- modwt/streaming/MODWTStreamingTransformImpl.java ---------------------------------------------------------

def _py_stream_windows(data, buffer_size, overlap):
    """processSampleInternal / processBuffer (:161-203) statement by statement; returns the windows and the
    number of unconsumed samples"""
    ring = [0.0] * (buffer_size + overlap)
    write, level, windows = 0, 0, []
    for s in data:
        ring[write] = s
        write = (write + 1) % len(ring)
        level += 1
        if level >= buffer_size:
            read = (write - buffer_size + len(ring)) % len(ring)
            windows.append([ring[(read + i) % len(ring)] for i in range(buffer_size)])
            level -= buffer_size - overlap
    return windows, level


@pytest.mark.parametrize("wn,bsz", [("haar", 2), ("haar", 16), ("db4", 8), ("db4", 64), ("sym8", 100), ("db4", 256)])
def test_streaming_transform_windows(wn, bsz):
    wv = get_wavelet(wn)
    rng = np.random.default_rng(bsz)
    data = rng.standard_normal(5 * bsz + 3)
    st = o.StreamingTransformState(wv, bsz)
    got = np.concatenate([st.process(data[:7]), st.process(data[7:7 + bsz]), st.process(data[7 + bsz:])])
    windows, level = _py_stream_windows(data.tolist(), bsz, len(wv.dec_lo) - 1)
    assert got.shape[0] == len(windows) and st.level() == level
    want = o.modwt_forward_batch(np.array(windows), wv)
    assert np.array_equal(got, want)
    # window k starts at sample k * (bufferSize - overlap): consecutive windows share filter length - 1 samples
    hop = bsz - (len(wv.dec_lo) - 1)
    for k, w in enumerate(windows):
        assert w == data[k * hop:k * hop + bsz].tolist()
    # flush: the unconsumed samples, zero-padded (:205-231)
    tail = st.flush()
    assert tail.shape[0] == (1 if level > 0 else 0)
    if level > 0:
        padded = np.zeros(bsz)
        padded[:level] = data[data.size - level:]
        assert np.array_equal(tail, o.modwt_forward_batch(padded[None], wv))
    assert st.level() == 0 and st.flush().shape[0] == 0


# ---- modwt/streaming/MultiLevelMODWTStreamingTransform.java ---------------------------------------------------

@pytest.mark.parametrize("wn,bsz,levels", [("haar", 16, 3), ("db4", 64, 3), ("db4", 256, 5), ("sym8", 100, 2)])
def test_multilevel_streaming_blocks(wn, bsz, levels):
    wv = get_wavelet(wn)
    rng = np.random.default_rng(bsz)
    data = rng.standard_normal(3 * bsz + 5)
    st, got = o.stream_multilevel(data, wv, bsz, levels)
    assert st == 0 and got.shape == (4, levels + 1, bsz)
    tail = np.zeros(bsz)
    tail[:5] = data[3 * bsz:]
    blocks = np.concatenate([data[:3 * bsz].reshape(3, bsz), tail[None]])  # :138-146 and :166-176: no overlap, zero-padded tail
    st, want = o.decompose_batch(blocks, wv, levels)
    assert st == 0 and np.array_equal(got, want)
    st, got = o.stream_multilevel(data, wv, bsz, levels, flush=False)
    assert got.shape[0] == 3
    # more levels than the block length allows: the reference's decompose throws
    assert o.stream_multilevel(data[:40], get_wavelet("db4"), 16, 5)[0] != 0


# ---- the host-side mirror's buffering (vectorwave_amd/api.py), with the transform stubbed: no GPU -------------------

def _py_stream_checked(state, data, buffer_size, overlap, multilevel=False):
    """The reference's sample loop with forward()'s validation (MODWTStreamingTransformImpl.java:168-227,
    MultiLevelMODWTStreamingTransform.java:124-167): state = [history list, samplesInBuffer, samples counted].
    Returns (windows emitted, index of the failing sample or -1)."""
    hist, level, counted = state
    windows = []
    for t, s in enumerate(data):
        hist.append(s)
        level += 1
        if level >= buffer_size:
            w = hist[-buffer_size:]
            if not np.isfinite(w).all():
                state[:] = [hist, level, counted]
                return windows, t
            windows.append(list(w))
            level = 0 if multilevel else level - (buffer_size - overlap)
        counted += 1
    state[:] = [hist, level, counted]
    return windows, -1


class _StubbedTransform:
    """Mixin: windows go to a list instead of the GPU; a non-finite window raises like the validating forward."""

    def _transform_windows(self, windows):
        from vectorwave_amd.api import InvalidSignalException
        for k in range(windows.shape[1]):
            w = np.array(windows[0, k])
            if not np.isfinite(w).all():
                raise InvalidSignalException("non-finite", "VAL_003")
            self.seen_windows.append(w.tolist())


@pytest.mark.parametrize("multilevel", [False, True])
@pytest.mark.parametrize("bsz,bad_at", [(16, [40]), (16, [15]), (16, [0]), (8, [8, 9]), (32, [100, 164]), (16, [])])
def test_mirror_buffering_follows_the_reference_through_failures(bsz, bad_at, multilevel):
    from vectorwave_amd import api

    base = api.MultiLevelMODWTStreamingTransform if multilevel else api.MODWTStreamingTransform
    cls = type("Stubbed", (_StubbedTransform, base), {})
    t = cls.__new__(cls)
    if multilevel:
        base.__init__(t, "db2", api.BoundaryMode.PERIODIC, bsz, 2, 1, ctx=object())
    else:
        base.__init__(t, "db2", api.BoundaryMode.PERIODIC, bsz, 1, ctx=object())
    t.seen_windows = []
    overlap = 0 if multilevel else 3
    rng = np.random.default_rng(bsz + len(bad_at))
    data = rng.standard_normal(12 * bsz + 5)
    for b in bad_at:
        data[b] = np.nan if b % 2 == 0 else np.inf
    state = [[], 0, 0]
    want_windows = []
    pos, chunk = 0, 0
    chunks = [5, bsz, 3 * bsz + 1, 1, 2, 7 * bsz]
    while pos < data.size:
        size = chunks[chunk % len(chunks)]
        chunk += 1
        piece = data[pos:pos + size]
        w, failed = _py_stream_checked(state, piece.tolist(), bsz, overlap, multilevel)
        want_windows += w
        if failed < 0:
            t.process(piece)
            pos += piece.size
        else:
            with pytest.raises(api.InvalidSignalException):
                t.process(piece)
            pos += failed + 1  # the caller carries on after the sample that failed
        assert t.seen_windows == want_windows
        assert t.getBufferLevel() == state[1]
        assert t.getStatistics().getSamplesProcessed() == state[2]
