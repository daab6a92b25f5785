"""GPU parity of the streaming block transform (modwt/streaming/MODWTStreamingTransformImpl.java) against the
oracle's sample-by-sample restatement: same blocks, bit for bit, whatever the chunking of the input; flush,
close, reset, statistics and error behaviour of the Python mirror."""
import numpy as np
import pytest

from oracle import oracle as o
from vectorwave_amd import (BoundaryMode, InvalidArgumentException, InvalidSignalException, InvalidStateException,
                            MODWTResult, MODWTStreamingTransform)
from vectorwave_amd.wavelets import get_wavelet

from util import P, S, Z, assert_bits_equal

pytestmark = pytest.mark.gpu


class _Collector:
    def __init__(self):
        self.items, self.completed = [], False

    def onNext(self, item):
        self.items.append(item)

    def onComplete(self):
        self.completed = True

    def blocks(self):
        return np.array([[r.detailCoeffs(), r.approximationCoeffs()] for r in self.items]).reshape(len(self.items), 2, -1)


@pytest.mark.parametrize("mode", [P, Z, S])
@pytest.mark.parametrize("wn,bsz", [("haar", 2), ("haar", 16), ("db4", 8), ("db4", 256), ("sym8", 100), ("coif3", 480)])
def test_blocks_match_whatever_the_chunking(ctx, wn, bsz, mode):
    w = get_wavelet(wn)
    rng = np.random.default_rng(bsz + mode)
    data = rng.standard_normal(7 * bsz + 5)
    ref = o.StreamingTransformState(w, bsz, mode)
    want = np.concatenate([ref.process(data), ref.flush()])
    sink = _Collector()
    t = MODWTStreamingTransform.create(w, BoundaryMode(mode), bsz, ctx=ctx)
    t.subscribe(sink)
    pos = 0
    for chunk in (1, 3, bsz - 1, bsz, 2 * bsz + 1, 5):
        t.process(data[pos:pos + chunk])
        pos += chunk
    t.processSample(data[pos])
    t.process(data[pos + 1:])
    hop = bsz - (len(w.dec_lo) - 1)
    assert t.getBufferLevel() == data.size - (len(sink.items)) * hop
    assert t.getStatistics().getSamplesProcessed() == data.size
    assert t.getStatistics().getBlocksProcessed() == len(sink.items)
    t.flush()
    assert t.getBufferLevel() == 0
    assert all(isinstance(r, MODWTResult) for r in sink.items)
    assert_bits_equal(sink.blocks(), want, "%s bufferSize=%d mode=%d" % (wn, bsz, mode))
    t.close()
    assert sink.completed and t.isClosed()


def test_lockstep_streams(ctx):
    w = get_wavelet("db4")
    streams, bsz = 6, 128
    rng = np.random.default_rng(3)
    data = rng.standard_normal((streams, 1000))
    got = []
    with MODWTStreamingTransform.create(w, BoundaryMode.PERIODIC, bsz, streams=streams, ctx=ctx) as t:
        t.subscribe(got.append)
        t.process(data[:, :333])
        t.process(data[:, 333:])
    got = np.array(got)  # [blocks][streams][2][bsz], the last one from close()
    for s in range(streams):
        ref = o.StreamingTransformState(w, bsz)
        want = np.concatenate([ref.process(data[s]), ref.flush()])
        assert_bits_equal(got[:, s], want, "stream %d" % s)


def test_close_drains_and_reset_clears(ctx):
    w = get_wavelet("db4")
    sink = _Collector()
    t = MODWTStreamingTransform.create(w, BoundaryMode.PERIODIC, 64, ctx=ctx)
    t.subscribe(sink)
    t.process(np.arange(10.0))
    t.reset()
    assert t.getBufferLevel() == 0 and t.getStatistics().getSamplesProcessed() == 0
    t.process(np.arange(10.0))
    t.close()
    t.close()
    padded = np.zeros(64)
    padded[:10] = np.arange(10.0)
    assert len(sink.items) == 1
    assert_bits_equal(sink.blocks(), o.modwt_forward_batch(padded[None], w), "drained block")
    for call in (lambda: t.process(np.zeros(4)), lambda: t.processSample(1.0), t.flush, t.reset):
        with pytest.raises(InvalidStateException):
            call()


def test_argument_checks(ctx):
    for bad in (lambda: MODWTStreamingTransform.create(None, BoundaryMode.PERIODIC),
                lambda: MODWTStreamingTransform.create("db4", None),
                lambda: MODWTStreamingTransform.create("db4", BoundaryMode.PERIODIC, 0),
                lambda: MODWTStreamingTransform.create("db4", BoundaryMode.PERIODIC, 7),          # below the filter length
                lambda: MODWTStreamingTransform.create("db4", BoundaryMode.PERIODIC, 20000000)):  # above 100 MB
        with pytest.raises(InvalidArgumentException):
            bad()
    t = MODWTStreamingTransform.create("db4", BoundaryMode.PERIODIC, ctx=ctx)
    assert t.bufferSize == 256
    with pytest.raises(InvalidSignalException):
        t.process(None)
    with pytest.raises(InvalidSignalException):
        t.process(np.zeros(0))
    with pytest.raises(InvalidSignalException):
        t.process(np.full(300, np.nan))


@pytest.mark.parametrize("mode", [P, Z, S])
@pytest.mark.parametrize("wn,bsz,levels", [("haar", 16, 3), ("db4", 64, 3), ("db4", 480, 4), ("sym8", 1000, 5)])
def test_multilevel_streaming_blocks(ctx, wn, bsz, levels, mode):
    w = get_wavelet(wn)
    rng = np.random.default_rng(bsz + levels)
    data = rng.standard_normal(5 * bsz + 7)
    st, want = o.stream_multilevel(data, w, bsz, levels, mode)
    assert st == 0
    sink = _Collector()
    t = MODWTStreamingTransform.createMultiLevel(w, BoundaryMode(mode), bsz, levels, ctx=ctx)
    t.subscribe(sink)
    t.process(data[:3])
    t.process(data[3:2 * bsz + 1])
    t.processSample(data[2 * bsz + 1])
    t.process(data[2 * bsz + 2:])
    assert t.getBufferLevel() == 7 and t.getStatistics().getBlocksProcessed() == 5
    t.close()
    assert sink.completed and len(sink.items) == 6 * levels
    for blk in range(6):
        for level in range(1, levels + 1):
            item = sink.items[blk * levels + level - 1]
            assert_bits_equal(item.detailCoeffs(), want[blk, level - 1], "block %d level %d" % (blk, level))
            if level == levels:
                assert_bits_equal(item.approximationCoeffs(), want[blk, levels], "block %d approximation" % blk)
            else:
                assert item.approximationCoeffs().size == 0  # :247-249
    with pytest.raises(InvalidStateException):
        t.process(np.zeros(4))


def test_multilevel_streaming_lockstep_and_errors(ctx):
    w = get_wavelet("db4")
    streams, bsz, levels = 4, 128, 3
    data = np.random.default_rng(11).standard_normal((streams, 300))
    got = []
    with MODWTStreamingTransform.createMultiLevel(w, BoundaryMode.PERIODIC, bsz, levels, streams=streams, ctx=ctx) as t:
        t.subscribe(got.append)
        t.process(data)
    got = np.array(got)  # [blocks][streams][levels+1][bsz]
    for s in range(streams):
        st, want = o.stream_multilevel(data[s], w, bsz, levels)
        assert_bits_equal(got[:, s], want, "stream %d" % s)
    for bad in (lambda: MODWTStreamingTransform.createMultiLevel(None, BoundaryMode.PERIODIC, 64, 2),
                lambda: MODWTStreamingTransform.createMultiLevel("db4", None, 64, 2),
                lambda: MODWTStreamingTransform.createMultiLevel("db4", BoundaryMode.PERIODIC, 0, 2),
                lambda: MODWTStreamingTransform.createMultiLevel("db4", BoundaryMode.PERIODIC, 64, 0)):
        with pytest.raises(InvalidArgumentException):
            bad()
    t = MODWTStreamingTransform.createMultiLevel("db4", BoundaryMode.PERIODIC, 16, 5, ctx=ctx)
    with pytest.raises(InvalidArgumentException):  # five levels do not fit a 16-sample block
        t.process(np.zeros(16))


@pytest.mark.parametrize("wn,bsz,bad_at", [("db4", 64, [200]), ("haar", 16, [15]), ("db4", 32, [0]), ("sym8", 100, [250, 251]),
                                           ("db2", 16, [37, 90])])
def test_non_finite_window_fails_where_the_reference_fails(ctx, wn, bsz, bad_at):
    """A window with a NaN or an infinity: the blocks before it are published, the call raises, and the buffer is left
    as the reference leaves it (sample stored, nothing consumed), so that what the caller feeds next produces the
    reference's blocks again.  Checked against the oracle's walk of the reference's sample loop."""
    w = get_wavelet(wn)
    rng = np.random.default_rng(bsz)
    data = rng.standard_normal(9 * bsz + 7)
    for k, b in enumerate(bad_at):
        data[b] = np.nan if k % 2 == 0 else -np.inf
    ref = o.StreamingTransformState(w, bsz, P)
    sink = _Collector()
    t = MODWTStreamingTransform.create(w, BoundaryMode.PERIODIC, bsz, ctx=ctx)
    t.subscribe(sink)
    want = []
    pos, failures = 0, 0
    chunks = [3 * bsz + 1, 5, bsz, 1, 4 * bsz]
    k = 0
    while pos < data.size:
        piece = data[pos:pos + chunks[k % len(chunks)]]
        k += 1
        blocks, failed = ref.process_checked(piece)
        want += list(blocks)
        if failed < 0:
            t.process(piece)
            pos += piece.size
        else:
            failures += 1
            with pytest.raises(InvalidSignalException):
                t.process(piece)
            pos += failed + 1
        assert len(sink.items) == len(want)
        assert t.getBufferLevel() == ref.level()
    assert failures >= len(bad_at)
    if want:
        assert_bits_equal(sink.blocks(), np.array(want), "%s bufferSize=%d" % (wn, bsz))
