package ai.prophetizo.wavelet.modwt.streaming;

import ai.prophetizo.wavelet.api.BoundaryMode;
import ai.prophetizo.wavelet.api.Daubechies;
import ai.prophetizo.wavelet.api.Wavelet;
import ai.prophetizo.wavelet.denoising.WaveletDenoiser.ThresholdMethod;
import ai.prophetizo.wavelet.denoising.WaveletDenoiser.ThresholdType;
import ai.prophetizo.wavelet.exception.InvalidArgumentException;
import ai.prophetizo.wavelet.gpu.GpuModwtBackend;
import ai.prophetizo.wavelet.modwt.MODWTTransform;
import ai.prophetizo.wavelet.util.ValidationUtils;

import java.lang.foreign.MemorySegment;
import java.util.concurrent.Flow;
import java.util.concurrent.SubmissionPublisher;
import java.util.concurrent.atomic.AtomicBoolean;
import java.util.concurrent.atomic.AtomicLong;

/**
 * Drop-in for the reference's block-by-block denoiser (modwt/streaming/MODWTStreamingDenoiser.java:35): same
 * package, builder, public methods, nested enum and exception rules.  The ring of |detail| samples that tracks the
 * noise level (:132-206) and the level itself live in GPU memory; one {@link #denoise} is one downcall
 * (vw_streaming_denoise_batch) that transforms the block, feeds the ring, re-estimates the noise (MAD :211-241 or
 * standard deviation :249-268), picks the threshold (:276-321), thresholds and inverts.
 *
 * <p>{@link Builder#streams(int)} is an addition: k denoisers advanced in lock step, one block per stream and call
 * ({@link #denoiseAll}), which is how many instruments share one launch.  With the default of one stream the class
 * behaves as the reference's.  There is no CPU fallback.
 */
public class MODWTStreamingDenoiser implements Flow.Publisher<double[]>, AutoCloseable {

    private final Wavelet wavelet;
    private final BoundaryMode boundaryMode;
    private final int bufferSize;
    private final ThresholdType thresholdType;
    private final ThresholdMethod thresholdMethod;
    private final double thresholdMultiplier;
    private final NoiseEstimation noiseEstimation;
    private final int noiseWindowSize;
    private final int streams;

    private final SubmissionPublisher<double[]> publisher;
    private final AtomicBoolean closed = new AtomicBoolean(false);
    private final AtomicLong samplesProcessed = new AtomicLong(0);

    private final GpuModwtBackend backend;
    /** the native stream state; guarded by {@code this} together with {@link #closed} */
    private MemorySegment state;
    private double lastNoiseLevel;

    /** :58-65; the ordinals are the native VW_NOISE_* values. */
    public enum NoiseEstimation {
        /** median absolute deviation of the noise window / 0.6745 */
        MAD,
        /** sample standard deviation of the noise window */
        STD,
        /** no window, no estimate */
        FIXED
    }

    private MODWTStreamingDenoiser(Builder builder) {
        // :68-69: both constructors of the reference run here, with their argument checks
        new MODWTTransform(builder.wavelet, builder.boundaryMode);
        this.wavelet = builder.wavelet;
        this.boundaryMode = builder.boundaryMode;
        this.bufferSize = builder.bufferSize;
        this.thresholdType = builder.thresholdType;
        this.thresholdMethod = builder.thresholdMethod;
        this.thresholdMultiplier = builder.thresholdMultiplier;
        this.noiseEstimation = builder.noiseEstimation;
        this.noiseWindowSize = builder.noiseWindowSize;
        this.streams = builder.streams;
        this.publisher = new SubmissionPublisher<>();
        this.backend = GpuModwtBackend.shared();
        this.state = backend.createStreamState(streams, noiseWindowSize);
    }

    /**
     * :93-127.  IllegalStateException("Denoiser is closed") after close(); InvalidArgumentException for a null or
     * empty block; non-finite samples are rejected where the reference's MODWTTransform.forward rejects them.
     */
    public double[] denoise(double[] samples) {
        if (closed.get()) {
            throw new IllegalStateException("Denoiser is closed");
        }
        if (samples == null || samples.length == 0) {
            throw new InvalidArgumentException("Samples cannot be null or empty");
        }
        if (streams != 1) {
            throw new InvalidArgumentException("This denoiser advances " + streams + " streams per call: use denoiseAll");
        }
        double[] denoised = denoiseAll(new double[][] {samples})[0];
        if (publisher.hasSubscribers()) {
            publisher.submit(denoised.clone());   // :122-124
        }
        return denoised;
    }

    /**
     * One block per stream, blocks[streams][n], all of one length; every stream's ring, noise level and sample count
     * advance by one block (not in the reference).  Subscribers are not fed from here.
     */
    public double[][] denoiseAll(double[][] blocks) {
        if (closed.get()) {
            throw new IllegalStateException("Denoiser is closed");
        }
        if (blocks == null || blocks.length != streams) {
            throw new InvalidArgumentException("Expected one block per stream: " + streams + " rows");
        }
        for (double[] b : blocks) {
            if (b == null || b.length == 0) {
                throw new InvalidArgumentException("Samples cannot be null or empty");
            }
            ValidationUtils.validateFiniteValues(b, "signal");
        }
        double[][] out;
        synchronized (this) {
            if (state == null) {
                throw new IllegalStateException("Denoiser is closed");
            }
            out = backend.streamingDenoise(state, blocks, wavelet, boundaryMode, thresholdMethod.ordinal(),
                thresholdType == ThresholdType.SOFT, thresholdMultiplier, noiseEstimation.ordinal());
        }
        samplesProcessed.addAndGet(blocks[0].length);
        return out;
    }

    /** :332-334: the level after the last block (of stream 0 when there are several). */
    public double getEstimatedNoiseLevel() {
        return getEstimatedNoiseLevels()[0];
    }

    /** The level of every stream (not in the reference). */
    public synchronized double[] getEstimatedNoiseLevels() {
        if (state == null) {
            double[] lv = new double[streams];
            java.util.Arrays.fill(lv, lastNoiseLevel);
            return lv;
        }
        return backend.estimatedNoiseLevels(state, streams);
    }

    /** :341-343: samples per stream. */
    public long getSamplesProcessed() {
        return samplesProcessed.get();
    }

    public int getBufferSize() {
        return bufferSize;
    }

    @Override
    public void subscribe(Flow.Subscriber<? super double[]> subscriber) {
        publisher.subscribe(subscriber);
    }

    /** :351-355, and the GPU state goes with it. */
    @Override
    public void close() {
        if (closed.compareAndSet(false, true)) {
            publisher.close();
            synchronized (this) {
                if (state != null) {
                    lastNoiseLevel = backend.estimatedNoiseLevels(state, streams)[0];
                    backend.destroyStreamState(state);
                    state = null;
                }
            }
        }
    }

    public boolean isClosed() {
        return closed.get();
    }

    /** :369-506: same defaults and the same argument checks. */
    public static class Builder {
        private Wavelet wavelet = Daubechies.DB4;
        private BoundaryMode boundaryMode = BoundaryMode.PERIODIC;
        private int bufferSize = 256;
        private ThresholdType thresholdType = ThresholdType.SOFT;
        private ThresholdMethod thresholdMethod = ThresholdMethod.UNIVERSAL;
        private double thresholdMultiplier = 1.0;
        private NoiseEstimation noiseEstimation = NoiseEstimation.MAD;
        private int noiseWindowSize = 1024;
        private int streams = 1;

        public Builder wavelet(Wavelet wavelet) {
            if (wavelet == null) {
                throw new InvalidArgumentException("Wavelet cannot be null");
            }
            this.wavelet = wavelet;
            return this;
        }

        public Builder boundaryMode(BoundaryMode boundaryMode) {
            if (boundaryMode == null) {
                throw new InvalidArgumentException("Boundary mode cannot be null");
            }
            this.boundaryMode = boundaryMode;
            return this;
        }

        public Builder bufferSize(int bufferSize) {
            if (bufferSize <= 0) {
                throw new InvalidArgumentException("Buffer size must be positive");
            }
            this.bufferSize = bufferSize;
            return this;
        }

        public Builder thresholdType(ThresholdType thresholdType) {
            if (thresholdType == null) {
                throw new InvalidArgumentException("Threshold type cannot be null");
            }
            this.thresholdType = thresholdType;
            return this;
        }

        public Builder thresholdMethod(ThresholdMethod thresholdMethod) {
            if (thresholdMethod == null) {
                throw new InvalidArgumentException("Threshold method cannot be null");
            }
            this.thresholdMethod = thresholdMethod;
            return this;
        }

        public Builder thresholdMultiplier(double thresholdMultiplier) {
            if (thresholdMultiplier <= 0) {
                throw new InvalidArgumentException("Threshold multiplier must be positive");
            }
            this.thresholdMultiplier = thresholdMultiplier;
            return this;
        }

        public Builder noiseEstimation(NoiseEstimation noiseEstimation) {
            if (noiseEstimation == null) {
                throw new InvalidArgumentException("Noise estimation cannot be null");
            }
            this.noiseEstimation = noiseEstimation;
            return this;
        }

        public Builder noiseWindowSize(int noiseWindowSize) {
            if (noiseWindowSize <= 0) {
                throw new InvalidArgumentException("Noise window size must be positive");
            }
            this.noiseWindowSize = noiseWindowSize;
            return this;
        }

        /** Number of lock-step streams (not in the reference; default 1). */
        public Builder streams(int streams) {
            if (streams <= 0) {
                throw new InvalidArgumentException("Stream count must be positive");
            }
            this.streams = streams;
            return this;
        }

        public MODWTStreamingDenoiser build() {
            return new MODWTStreamingDenoiser(this);
        }
    }
}
