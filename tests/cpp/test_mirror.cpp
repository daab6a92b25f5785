// GPU test of the C++ host mirror (include/vectorwave/modwt.hpp): the reference's unit tests
// restated in C++, plus bit-for-bit comparison with the CPU oracle (linked as the checker).
// Built by tests/cpp/Makefile, run by tests/test_gpu_cpp_mirror.py.  Reference test citations are
// relative to src/test/java/ai/prophetizo/wavelet/.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>

#include "vectorwave/modwt.hpp"

extern "C" {
int vwo_multilevel_decompose_batch(const double*, int64_t, int64_t, int, const double*, int, const double*, int, int, double*);
void vwo_multilevel_reconstruct_batch(const double*, int64_t, int64_t, int, const double*, int, const double*, int, int, double*);
int vwo_swt_denoise_batch(const double*, int64_t, int64_t, int, const double*, int, const double*, int, const double*, int, const double*,
                          int, int, double, int, double*);
void vwo_modwt_forward_batch(const double*, int64_t, int64_t, const double*, int, const double*, int, int, double*);
int vwo_wavelet_denoise_batch(const double*, int64_t, int64_t, int, const double*, int, const double*, int, const double*, int, const double*, int,
                              int, int, double, int, double*, double*);
int vwo_streaming_denoise_batch(const double*, int64_t, int64_t, const double*, int, const double*, int, const double*, int, const double*, int, int, int,
                                int, double, int, double*, int64_t, int64_t*, double*, double*, double*);
void vwo_batch_modwt_soa(const double*, double*, double*, const double*, const double*, int, int, int64_t, int64_t);
int64_t vwo_sparse_vector(const double*, int64_t, double, int32_t*, double*);
}

using namespace vectorwave;
static int failures = 0;
#define CHECK(cond)                                                     \
    do {                                                                \
        if (!(cond)) {                                                  \
            std::printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond); \
            ++failures;                                                 \
        }                                                               \
    } while (0)

static std::vector<double> make_signal(int64_t n, unsigned seed) {
    std::mt19937_64 rng(seed);
    std::normal_distribution<double> nd(0.0, 1.0);
    std::vector<double> x(n);
    for (int64_t i = 0; i < n; ++i) x[i] = std::sin(2 * M_PI * i / 32.0) + 0.5 * std::sin(2 * M_PI * i / 64.0) + 0.1 * nd(rng);
    return x;
}
static bool bits_equal(const std::vector<double>& a, const std::vector<double>& b) {
    return a.size() == b.size() && std::memcmp(a.data(), b.data(), a.size() * sizeof(double)) == 0;
}
static double max_abs_diff(const std::vector<double>& a, const std::vector<double>& b) {
    double m = 0;
    for (size_t i = 0; i < a.size(); ++i) m = std::fmax(m, std::fabs(a[i] - b[i]));
    return m;
}

int main() {
    // modwt/MODWTPercivalWaldenValidationTest.java:73-82: Haar {1,2,3,4} -> approx {2.5,1.5,2.5,3.5}
    {
        MODWTTransform t(getWavelet("haar"), BoundaryMode::PERIODIC);
        MODWTResult r = t.forward({1, 2, 3, 4});
        const double want[4] = {2.5, 1.5, 2.5, 3.5};
        for (int i = 0; i < 4; ++i) CHECK(std::fabs(r.approximationCoeffs()[i] - want[i]) < 1e-12);
        CHECK(max_abs_diff(t.inverse(r), {1, 2, 3, 4}) < 1e-12);
    }
    // internal/ScalarOpsTest.java:225-274
    {
        auto c = WaveletOperations::circularConvolveMODWT({1, 2, 3, 4}, {0.6, 0.4});
        auto z = WaveletOperations::zeroPaddingConvolveMODWT({1, 2, 3, 4}, {0.6, 0.4});
        CHECK(std::fabs(c[0] - 2.2) < 1e-10 && std::fabs(c[1] - 1.6) < 1e-10);
        CHECK(std::fabs(z[0] - 0.6) < 1e-10 && std::fabs(z[3] - 3.6) < 1e-10);
        auto l2 = WaveletOperations::convolve({1, 2, 3, 4, 5, 6, 7, 8}, {0.7, 0.3}, BoundaryMode::PERIODIC, 2);
        CHECK(std::fabs(l2[0] - (1 * 0.7 + 7 * 0.3)) < 1e-10 && std::fabs(l2[1] - (2 * 0.7 + 8 * 0.3)) < 1e-10);
    }
    // modwt/MODWTTransformTest.java:131-154: empty / NaN / Inf rejected
    {
        MODWTTransform t(getWavelet("haar"), BoundaryMode::PERIODIC);
        bool threw = false;
        try { t.forward({}); } catch (const InvalidSignalException& e) { threw = e.getErrorCode() == "VAL_006"; }
        CHECK(threw);
        threw = false;
        try { t.forward({1.0, NAN, 3.0}); } catch (const InvalidSignalException& e) { threw = e.getErrorCode() == "VAL_003"; }
        CHECK(threw);
    }
    // modwt/MultiLevelMODWTTransformTest.java:203-218, 254-291
    {
        MultiLevelMODWTTransform t(getWavelet("haar"), BoundaryMode::PERIODIC);
        CHECK(t.getMaximumLevels(10000) == 9 && MultiLevelMODWTTransform::getMaxDecompositionLevels() == 10);
        auto x = make_signal(64, 1);
        for (int bad : {0, 20}) {
            bool threw = false;
            try { t.decompose(x, bad); } catch (const InvalidArgumentException& e) { threw = e.getErrorCode() == "CFG_004"; }
            CHECK(threw);
        }
    }
    // multi-level: bit-for-bit vs the oracle, and the reference's reconstruction tolerance
    struct Case { const char* w; int64_t n; int j; int mode; };
    const Case cases[] = {{"haar", 8, 3, 0}, {"db4", 32, 3, 0}, {"db4", 1000, 3, 0}, {"db4", 4096, 6, 0}, {"sym8", 777, 5, 1},
                          {"coif3", 301, 4, 2}, {"db10", 5001, 6, 0}, {"bior4.4", 500, 4, 0}};
    for (const Case& c : cases) {
        const Wavelet& w = getWavelet(c.w);
        MultiLevelMODWTTransform t(w, static_cast<BoundaryMode>(c.mode));
        auto x = make_signal(c.n, 7 + static_cast<unsigned>(c.n));
        MultiLevelMODWTResult r = t.decompose(x, c.j);
        std::vector<double> want(static_cast<size_t>(c.j + 1) * c.n), wantx(c.n);
        CHECK(vwo_multilevel_decompose_batch(x.data(), 1, c.n, c.j, w.decLo.data(), (int)w.decLo.size(), w.decHi.data(), (int)w.decHi.size(), c.mode, want.data()) == 0);
        CHECK(bits_equal(r.block(), want));
        vwo_multilevel_reconstruct_batch(want.data(), 1, c.n, c.j, w.recLo.data(), (int)w.recLo.size(), w.recHi.data(), (int)w.recHi.size(), c.mode, wantx.data());
        auto xr = t.reconstruct(r);
        CHECK(bits_equal(xr, wantx));
        if (c.mode == 0 && (std::string(c.w) == "haar" || std::string(c.w) == "db4")) CHECK(max_abs_diff(xr, x) < 1e-10);
    }
    // batch == per-signal (modwt/BatchSIMDMODWTTest.java:55-150), DB4 B=16 N=64
    {
        const Wavelet& w = getWavelet("db4");
        MODWTTransform t(w, BoundaryMode::PERIODIC);
        std::vector<std::vector<double>> xs;
        for (int b = 0; b < 16; ++b) xs.push_back(make_signal(64, 100 + b));
        auto batch = t.forwardBatch(xs);
        for (int b = 0; b < 16; ++b) {
            MODWTResult one = t.forward(xs[b]);
            CHECK(bits_equal(batch[b].approximationCoeffs(), one.approximationCoeffs()));
            CHECK(bits_equal(batch[b].detailCoeffs(), one.detailCoeffs()));
        }
        auto back = t.inverseBatch(batch);
        for (int b = 0; b < 16; ++b) CHECK(max_abs_diff(back[b], xs[b]) < 1e-10);
    }
    // swt/VectorWaveSwtAdapterTest.java:23-42, 243-268, 336-356, 205-217
    {
        const Wavelet& w = getWavelet("db4");
        VectorWaveSwtAdapter a(w);
        auto x = make_signal(256, 42);
        auto r = a.forward(x, 4);
        CHECK(max_abs_diff(a.inverse(r), x) < 1e-10);
        auto y1 = a.denoise(x, 4, -1, true), y2 = a.denoise(x, 4, -999, true);
        CHECK(bits_equal(y1, y2));
        std::vector<double> want(256);
        CHECK(vwo_swt_denoise_batch(x.data(), 1, 256, 4, w.decLo.data(), 8, w.decHi.data(), 8, w.recLo.data(), 8, w.recHi.data(), 8, 0, -1.0, 1, want.data()) == 0);
        CHECK(bits_equal(y1, want));
        a.applyUniversalThreshold(r, true);
        CHECK(bits_equal(a.inverse(r), y1));
        std::vector<double> total(256, 0.0);
        for (int k = 0; k < 4; ++k) {
            auto e = a.extractLevel(x, 3, k);
            for (int i = 0; i < 256; ++i) total[i] += e[i];
        }
        CHECK(max_abs_diff(total, x) < 1e-8);
    }
    {
        // WaveletDenoiser: every selection method, single- and multi-level, against the oracle
        const Wavelet& w = getWavelet("db4");
        WaveletDenoiser d = WaveletDenoiser::forFinancialData();
        auto x = make_signal(512, 7);
        using M = WaveletDenoiser::ThresholdMethod;
        using T = WaveletDenoiser::ThresholdType;
        for (int method = 0; method < 4; ++method) {
            std::vector<double> want(512), wthr(4);
            CHECK(vwo_wavelet_denoise_batch(x.data(), 1, 512, 4, w.decLo.data(), 8, w.decHi.data(), 8, w.recLo.data(), 8, w.recHi.data(), 8, 0, method,
                                            0.0, 0, want.data(), wthr.data()) == 0);
            CHECK(bits_equal(d.denoiseMultiLevel(x, 4, static_cast<M>(method), T::HARD), want));
            CHECK(vwo_wavelet_denoise_batch(x.data(), 1, 512, 0, w.decLo.data(), 8, w.decHi.data(), 8, w.recLo.data(), 8, w.recHi.data(), 8, 0, method,
                                            0.0, 1, want.data(), wthr.data()) == 0);
            CHECK(bits_equal(d.denoise(x, static_cast<M>(method)), want));
        }
        std::vector<double> want(512), wthr(1);
        CHECK(vwo_wavelet_denoise_batch(x.data(), 1, 512, 0, w.decLo.data(), 8, w.decHi.data(), 8, w.recLo.data(), 8, w.recHi.data(), 8, 0, 4, 0.3, 1,
                                        want.data(), wthr.data()) == 0);
        CHECK(bits_equal(d.denoiseFixed(x, 0.3, T::SOFT), want));
        bool threw = false;
        try { d.denoise(x, M::FIXED); } catch (const InvalidArgumentException&) { threw = true; }
        CHECK(threw);
        threw = false;
        try { d.denoiseMultiLevel(x, 3, M::FIXED, T::SOFT); } catch (const InvalidArgumentException&) { threw = true; }
        CHECK(threw);
    }
    {
        // MODWTStreamingDenoiser: three lock-step streams over four blocks, MAD and STD, with and without a multiplier
        const Wavelet& w = getWavelet("db4");
        using E = MODWTStreamingDenoiser::NoiseEstimation;
        const int64_t streams = 3, n = 200, window = 128;
        for (int est = 0; est < 2; ++est)
            for (double mult : {1.0, 1.5}) {
                MODWTStreamingDenoiser den = MODWTStreamingDenoiser::Builder().noiseWindowSize(window).streams(streams).thresholdMultiplier(mult)
                                                 .noiseEstimation(static_cast<E>(est)).build();
                std::vector<double> ring(streams * window, 0.0), level(streams, 0.0);
                int64_t index = 0;
                for (int blk = 0; blk < 4; ++blk) {
                    auto x = make_signal(streams * n, 100 + blk);
                    std::vector<double> want(x.size()), thr(streams);
                    CHECK(vwo_streaming_denoise_batch(x.data(), streams, n, w.decLo.data(), 8, w.decHi.data(), 8, w.recLo.data(), 8, w.recHi.data(), 8, 0, 0,
                                                      1, mult, est, ring.data(), window, &index, level.data(), want.data(), thr.data()) == 0);
                    CHECK(bits_equal(den.denoise(x), want));
                    CHECK(bits_equal(den.getEstimatedNoiseLevels(), level));
                }
                CHECK(den.getSamplesProcessed() == 4 * n);
                den.close();
                bool threw = false;
                try { den.denoise(std::vector<double>(streams * n, 0.0)); } catch (const std::logic_error&) { threw = true; }
                CHECK(threw);
            }
    }
    {
        // BatchSIMDMODWT: the three arithmetic flavours against the oracle, and the layout round trip
        const int64_t batch = 37, n = 64;
        auto aos = make_signal(batch * n, 9);
        std::vector<double> soa, back;
        BatchSIMDMODWT::convertToSoA(aos, batch, n, soa);
        for (int64_t b = 0; b < batch; ++b)
            for (int64_t t = 0; t < n; t += 13) CHECK(soa[t * batch + b] == aos[b * n + t]);
        BatchSIMDMODWT::convertFromSoA(soa, batch, n, back);
        CHECK(bits_equal(back, aos));
        for (const char* name : {"haar", "db2", "db4"}) {
            const Wavelet& w = getWavelet(name);
            std::vector<double> a, d, wa(batch * n), wd(batch * n);
            BatchSIMDMODWT::batchMODWTSoA(soa, a, d, w, batch, n);
            vwo_batch_modwt_soa(soa.data(), wa.data(), wd.data(), w.decLo.data(), w.decHi.data(), (int)w.decLo.size(), std::string(name) == "haar" ? 1 : 0,
                                batch, n);
            CHECK(bits_equal(a, wa));
            CHECK(bits_equal(d, wd));
        }
    }
    {
        // SWTResult: sparse form against the oracle's SparseVector, and the round trip
        const int levels = 3;
        const int64_t n = 1500;
        auto a = make_signal(n, 21);
        std::vector<std::vector<double>> d;
        for (int l = 0; l < levels; ++l) d.push_back(make_signal(n, 30 + l));
        SWTResult full(a, d, levels);
        SparseSWTResult sp = full.toSparse(0.8);
        int64_t kept = 0;
        for (int l = 0; l < levels; ++l) {
            std::vector<int32_t> wi(n);
            std::vector<double> wv(n);
            const int64_t cnt = vwo_sparse_vector(d[l].data(), n, 0.8, wi.data(), wv.data());
            CHECK(sp.getNonZeroCount(l) == cnt);
            bool same = true;
            for (int64_t i = 0; i < cnt; ++i) same = same && sp.indices()[sp.offsets()[l] + i] == wi[i] && sp.values()[sp.offsets()[l] + i] == wv[i];
            CHECK(same);
            kept += cnt;
        }
        CHECK(sp.getCompressionRatio() == static_cast<double>(n * (levels + 1)) / static_cast<double>(n + 2 * kept));
        SWTResult back = sp.toFull();
        CHECK(bits_equal(back.getApproximation(), a));
        for (int l = 0; l < levels; ++l) {
            std::vector<double> want(n);
            for (int64_t i = 0; i < n; ++i) want[i] = std::fabs(d[l][i]) > 0.8 ? d[l][i] : 0.0;
            CHECK(bits_equal(back.getDetail(l), want));
        }
        bool threw = false;
        try { SWTResult bad(a, d, levels + 1); } catch (const std::invalid_argument&) { threw = true; }
        CHECK(threw);
    }
    std::printf(failures == 0 ? "cpp mirror: all checks passed\n" : "cpp mirror: %d checks FAILED\n", failures);
    return failures == 0 ? 0 : 1;
}
