// vectorwave/modwt.hpp -- header-only C++17 host mirror of the reference's MODWT / SWT API over
// the C ABI in vwgpu.h.
//
// The reference's host language is Java 23; where no JVM is available this header plays the role
// of the Java facade: same class and method names, argument meaning and error behaviour
// (reference paths under src/main/java/ai/prophetizo/wavelet/):
//   modwt/MODWTTransform.java:69            -> vectorwave::MODWTTransform
//   modwt/MultiLevelMODWTTransform.java:66  -> vectorwave::MultiLevelMODWTTransform
//   swt/VectorWaveSwtAdapter.java:61        -> vectorwave::VectorWaveSwtAdapter
//   WaveletOperations.java:12               -> vectorwave::WaveletOperations
//   api/*.java (Haar, Daubechies, Symlet, Coiflet, DiscreteMeyer, BiorthogonalSpline,
//   ReverseBiorthogonalSpline)              -> vectorwave::Wavelet / getWavelet()
//   exception/*.java                         -> vectorwave::InvalidSignalException etc.
// Every method that computes is one downcall into libvwgpu; nothing is computed on the host.
#pragma once

#include <cmath>
#include <cstdint>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../vwgpu.h"

namespace vectorwave {

enum class BoundaryMode : int32_t { PERIODIC = VW_BOUNDARY_PERIODIC, ZERO_PADDING = VW_BOUNDARY_ZERO_PADDING, SYMMETRIC = VW_BOUNDARY_SYMMETRIC };

// ---- exceptions (exception/ErrorCode.java:18-118) ---------------------------------------------
class WaveletTransformException : public std::runtime_error {
  public:
    WaveletTransformException(const std::string& msg, std::string code) : std::runtime_error(msg), code_(std::move(code)) {}
    const std::string& getErrorCode() const { return code_; }

  private:
    std::string code_;
};
class InvalidSignalException : public WaveletTransformException {
    using WaveletTransformException::WaveletTransformException;
};
class InvalidArgumentException : public WaveletTransformException {
    using WaveletTransformException::WaveletTransformException;
};
class NullPointerException : public std::invalid_argument {
    using std::invalid_argument::invalid_argument;
};

// ---- wavelets ------------------------------------------------------------------------------------
struct Wavelet {
    std::string name;
    std::vector<double> decLo, decHi, recLo, recHi;
    const std::vector<double>& lowPassDecomposition() const { return decLo; }
    const std::vector<double>& highPassDecomposition() const { return decHi; }
    const std::vector<double>& lowPassReconstruction() const { return recLo; }
    const std::vector<double>& highPassReconstruction() const { return recHi; }
};

namespace detail {
// g[i] = (i even ? +1 : -1) * h[L-1-i]  (api/Daubechies.java:320-327, Symlet.java:462-470, Coiflet.java:629-637)
inline std::vector<double> qmfOrthogonal(const std::vector<double>& h) {
    std::vector<double> g(h.size());
    for (size_t i = 0; i < h.size(); ++i) g[i] = (i % 2 == 0 ? 1 : -1) * h[h.size() - 1 - i];
    return g;
}
// sign = ((len-1-i) even ? +1 : -1)  (api/BiorthogonalSpline.java:297-327)
inline std::vector<double> qmfBiorthogonal(const std::vector<double>& h) {
    std::vector<double> g(h.size());
    for (size_t i = 0; i < h.size(); ++i) g[i] = (((h.size() - 1 - i) % 2 == 0) ? 1 : -1) * h[h.size() - 1 - i];
    return g;
}
inline const std::map<std::string, Wavelet>& table() {
    static const std::map<std::string, Wavelet> t = [] {
        std::map<std::string, Wavelet> m;
#define VW_TAPS(...) std::vector<double> { __VA_ARGS__ }
#define VW_ORTHO(NAME, H)                                                      \
    {                                                                          \
        Wavelet w;                                                             \
        w.name = NAME;                                                         \
        w.decLo = H;                                                           \
        w.decHi = std::string(NAME) == "haar" ? std::vector<double>{w.decLo[0], -w.decLo[1]} : qmfOrthogonal(w.decLo); \
        w.recLo = w.decLo;                                                     \
        w.recHi = w.decHi;                                                     \
        m[NAME] = w;                                                           \
    }
#define VW_BIOR(NAME, DLO, RLO)                                                \
    {                                                                          \
        Wavelet w;                                                             \
        w.name = NAME;                                                         \
        w.decLo = DLO;                                                         \
        w.recLo = RLO;                                                         \
        w.decHi = qmfBiorthogonal(w.recLo);                                    \
        w.recHi = qmfBiorthogonal(w.decLo);                                    \
        m[NAME] = w;                                                           \
        Wavelet r; /* reverse biorthogonal swaps the pairs (api/ReverseBiorthogonalSpline.java:95-117) */ \
        r.name = std::string("rbio") + std::string(NAME).substr(4);            \
        r.decLo = w.recLo;                                                     \
        r.decHi = w.recHi;                                                     \
        r.recLo = w.decLo;                                                     \
        r.recHi = w.decHi;                                                     \
        m[r.name] = r;                                                         \
    }
#include "wavelet_tables.inc"
#undef VW_TAPS
#undef VW_ORTHO
#undef VW_BIOR
        return m;
    }();
    return t;
}
}  // namespace detail

// Lookup by the reference's wavelet code ("haar", "db4", "sym8", "coif3", "bior4.4", "rbio2.2", "dmey";
// api/WaveletName.java:9-121).
inline const Wavelet& getWavelet(const std::string& name) {
    auto it = detail::table().find(name);
    if (it == detail::table().end()) throw InvalidArgumentException("unknown wavelet " + name, "CFG_001");
    return it->second;
}

// ---- context ---------------------------------------------------------------------------------------
class Context {
  public:
    explicit Context(int device = 0) {
        const int32_t st = vw_init(device, &ctx_);
        if (st != VW_OK) throw std::runtime_error("vw_init failed (status " + std::to_string(st) + "): no usable HIP device; there is no CPU fallback");
    }
    ~Context() {
        if (ctx_ != nullptr) vw_destroy(ctx_);
    }
    Context(const Context&) = delete;
    Context& operator=(const Context&) = delete;
    vw_ctx* get() const { return ctx_; }
    std::string lastError() const { return vw_last_error(ctx_); }
    std::string performanceInfo() const { return vw_performance_info(ctx_); }

    // Status -> the exception class the reference throws.
    void check(int32_t st) const {
        if (st == VW_OK) return;
        const std::string msg = lastError();
        switch (st) {
            case VW_E_NONFINITE: throw InvalidSignalException(msg, "VAL_003");
            case VW_E_EMPTY: throw InvalidSignalException(msg, "VAL_006");
            case VW_E_LEVEL: throw InvalidArgumentException(msg, "CFG_004");
            case VW_E_BOUNDARY: throw InvalidArgumentException(msg, "CFG_003");
            case VW_E_TOO_LARGE: throw InvalidArgumentException(msg, "VAL_005");
            case VW_E_UNSUPPORTED: throw InvalidArgumentException(msg, "CFG_005");
            case VW_E_NULL: throw NullPointerException(msg);
            case VW_E_NOT_POSITIVE: throw std::invalid_argument(msg);  // IllegalArgumentException
            case VW_E_LENGTH_MISMATCH: throw std::out_of_range(msg);  // ArrayIndexOutOfBoundsException
            default: throw std::runtime_error("libvwgpu status " + std::to_string(st) + ": " + msg);
        }
    }

  private:
    vw_ctx* ctx_ = nullptr;
};

inline std::shared_ptr<Context> defaultContext() {
    static std::shared_ptr<Context> c = std::make_shared<Context>(0);
    return c;
}

// ---- result containers -------------------------------------------------------------------------------
// modwt/MODWTResult.java:24 (+ MODWTResultImpl.java:31-63: getters return copies)
class MODWTResult {
  public:
    MODWTResult(std::vector<double> approx, std::vector<double> detail) : a_(std::move(approx)), d_(std::move(detail)) {
        if (a_.size() != d_.size()) throw std::invalid_argument("Approximation and detail coefficients must have the same length.");
        if (a_.empty()) throw std::invalid_argument("Coefficient arrays cannot be empty");
    }
    std::vector<double> approximationCoeffs() const { return a_; }
    std::vector<double> detailCoeffs() const { return d_; }
    int64_t getSignalLength() const { return static_cast<int64_t>(a_.size()); }
    bool isValid() const {
        for (double v : a_)
            if (!std::isfinite(v)) return false;
        for (double v : d_)
            if (!std::isfinite(v)) return false;
        return true;
    }

  private:
    std::vector<double> a_, d_;
};

// modwt/MultiLevelMODWTResult.java:24 / MutableMultiLevelMODWTResult.java:19: one contiguous
// [J+1][N] block ordered d_1..d_J, a_J (level 1 = finest).
class MultiLevelMODWTResult {
  public:
    MultiLevelMODWTResult(std::vector<double> block, int levels, int64_t n) : block_(std::move(block)), levels_(levels), n_(n) {}
    int getLevels() const { return levels_; }
    int64_t getSignalLength() const { return n_; }
    std::vector<double> getDetailCoeffsAtLevel(int level) const {
        validateLevel(level);
        return std::vector<double>(block_.begin() + (level - 1) * n_, block_.begin() + level * n_);
    }
    std::vector<double> getApproximationCoeffs() const { return std::vector<double>(block_.begin() + levels_ * n_, block_.end()); }
    double* getMutableDetailCoeffs(int level) {
        validateLevel(level);
        return block_.data() + (level - 1) * n_;
    }
    double* getMutableApproximationCoeffs() { return block_.data() + levels_ * n_; }
    bool isValid() const {
        for (double v : block_)
            if (!std::isfinite(v)) return false;
        return true;
    }
    const std::vector<double>& block() const { return block_; }
    std::vector<double>& block() { return block_; }

  private:
    void validateLevel(int level) const {
        if (level < 1 || level > levels_) throw std::invalid_argument("Level must be between 1 and " + std::to_string(levels_));
    }
    std::vector<double> block_;
    int levels_;
    int64_t n_;
};
using MutableMultiLevelMODWTResult = MultiLevelMODWTResult;

// ---- WaveletOperations (WaveletOperations.java:12) ---------------------------------------------------
struct WaveletOperations {
    static std::vector<double> convolve(const std::vector<double>& signal, const std::vector<double>& filter, BoundaryMode mode,
                                        int level = 1, std::shared_ptr<Context> ctx = defaultContext()) {
        std::vector<double> out(signal.size());
        ctx->check(vw_modwt_convolve_batch(ctx->get(), signal.data(), 1, static_cast<int64_t>(signal.size()), filter.data(),
                                           static_cast<int32_t>(filter.size()), level, static_cast<int32_t>(mode), out.data()));
        return out;
    }
    static std::vector<double> circularConvolveMODWT(const std::vector<double>& s, const std::vector<double>& f) { return convolve(s, f, BoundaryMode::PERIODIC); }
    static std::vector<double> zeroPaddingConvolveMODWT(const std::vector<double>& s, const std::vector<double>& f) { return convolve(s, f, BoundaryMode::ZERO_PADDING); }
    static std::vector<double> symmetricConvolveMODWT(const std::vector<double>& s, const std::vector<double>& f) { return convolve(s, f, BoundaryMode::SYMMETRIC); }
    static std::vector<double> threshold(const std::vector<double>& c, double thr, bool soft, std::shared_ptr<Context> ctx = defaultContext()) {
        if (thr < 0) throw std::invalid_argument("Threshold must be non-negative");
        std::vector<double> out(c.size());
        ctx->check(vw_threshold(ctx->get(), c.data(), static_cast<int64_t>(c.size()), thr, soft ? 1 : 0, out.data()));
        return out;
    }
    static std::vector<double> softThreshold(const std::vector<double>& c, double thr) { return threshold(c, thr, true); }
    static std::vector<double> hardThreshold(const std::vector<double>& c, double thr) { return threshold(c, thr, false); }
};

// ---- MODWTTransform (modwt/MODWTTransform.java:69) ------------------------------------------------------
class MODWTTransform {
  public:
    MODWTTransform(const Wavelet& wavelet, BoundaryMode mode, std::shared_ptr<Context> ctx = defaultContext())
        : w_(wavelet), mode_(mode), ctx_(std::move(ctx)) {}
    const Wavelet& getWavelet() const { return w_; }
    BoundaryMode getBoundaryMode() const { return mode_; }
    std::string getPerformanceInfo() const { return ctx_->performanceInfo(); }

    MODWTResult forward(const std::vector<double>& signal) const {
        if (signal.empty()) throw InvalidSignalException("Signal cannot be empty", "VAL_006");
        std::vector<double> c = forwardBlock(signal.data(), 1, static_cast<int64_t>(signal.size()), true);
        const int64_t n = static_cast<int64_t>(signal.size());
        return MODWTResult(std::vector<double>(c.begin() + n, c.end()), std::vector<double>(c.begin(), c.begin() + n));
    }
    std::vector<double> inverse(const MODWTResult& r) const {
        if (!r.isValid()) throw InvalidSignalException("MODWTResult contains invalid coefficients", "VAL_003");
        const int64_t n = r.getSignalLength();
        std::vector<double> c = r.detailCoeffs();
        const std::vector<double> a = r.approximationCoeffs();
        c.insert(c.end(), a.begin(), a.end());
        return inverseBlock(c.data(), 1, n, false);
    }
    // forwardBatch / inverseBatch (modwt/MODWTTransform.java:459-532): equal-length batches of at least 4
    // signals of at least 64 samples take the unvalidated batch path, everything else goes signal by signal.
    std::vector<MODWTResult> forwardBatch(const std::vector<std::vector<double>>& signals) const {
        std::vector<MODWTResult> out;
        if (signals.empty()) return out;
        const size_t n = signals[0].size();
        bool same = true;
        for (const auto& s : signals) same = same && s.size() == n;
        if (same && signals.size() >= 4 && n >= 64) {
            std::vector<double> x;
            x.reserve(signals.size() * n);
            for (const auto& s : signals) x.insert(x.end(), s.begin(), s.end());
            std::vector<double> c = forwardBlock(x.data(), static_cast<int64_t>(signals.size()), static_cast<int64_t>(n), false);
            for (size_t b = 0; b < signals.size(); ++b) {
                const double* p = c.data() + b * 2 * n;
                MODWTResult r(std::vector<double>(p + n, p + 2 * n), std::vector<double>(p, p + n));
                if (!r.isValid()) throw InvalidSignalException("approximationCoeffs contains a non-finite value", "VAL_003");
                out.push_back(std::move(r));
            }
            return out;
        }
        for (const auto& s : signals) out.push_back(forward(s));
        return out;
    }
    std::vector<std::vector<double>> inverseBatch(const std::vector<MODWTResult>& results) const {
        std::vector<std::vector<double>> out;
        if (results.empty()) return out;
        const int64_t n = results[0].getSignalLength();
        bool same = true;
        for (const auto& r : results) same = same && r.getSignalLength() == n;
        if (same && results.size() >= 4 && n >= 64) {
            std::vector<double> c;
            c.reserve(results.size() * 2 * n);
            for (const auto& r : results) {
                const auto d = r.detailCoeffs(), a = r.approximationCoeffs();
                c.insert(c.end(), d.begin(), d.end());
                c.insert(c.end(), a.begin(), a.end());
            }
            std::vector<double> x = inverseBlock(c.data(), static_cast<int64_t>(results.size()), n, true);
            for (size_t b = 0; b < results.size(); ++b) out.emplace_back(x.begin() + b * n, x.begin() + (b + 1) * n);
            return out;
        }
        for (const auto& r : results) out.push_back(inverse(r));
        return out;
    }

    // Raw block calls: x [B][N] -> coeffs [B][2][N] (detail, approximation) and back.
    std::vector<double> forwardBlock(const double* x, int64_t b, int64_t n, bool checkFinite) const {
        std::vector<double> c(static_cast<size_t>(b) * 2 * n);
        int64_t bad = -1;
        ctx_->check(vw_modwt_forward_batch(ctx_->get(), x, b, n, 1, w_.decLo.data(), static_cast<int32_t>(w_.decLo.size()), w_.decHi.data(),
                                           static_cast<int32_t>(w_.decHi.size()), static_cast<int32_t>(mode_),
                                           VW_FLAG_SINGLE_LEVEL | (checkFinite ? VW_FLAG_CHECK_FINITE : 0u), c.data(), &bad));
        return c;
    }
    std::vector<double> inverseBlock(const double* c, int64_t b, int64_t n, bool batchVariant) const {
        std::vector<double> x(static_cast<size_t>(b) * n);
        int64_t bad = -1;
        ctx_->check(vw_modwt_inverse_batch(ctx_->get(), c, b, n, 1, w_.recLo.data(), static_cast<int32_t>(w_.recLo.size()), w_.recHi.data(),
                                           static_cast<int32_t>(w_.recHi.size()), static_cast<int32_t>(mode_),
                                           VW_FLAG_SINGLE_LEVEL | (batchVariant ? VW_FLAG_SYMMETRIC_BATCH_VARIANT : 0u), x.data(), &bad));
        return x;
    }

  private:
    Wavelet w_;
    BoundaryMode mode_;
    std::shared_ptr<Context> ctx_;
};

// ---- MultiLevelMODWTTransform (modwt/MultiLevelMODWTTransform.java:66) -------------------------------------
class MultiLevelMODWTTransform {
  public:
    MultiLevelMODWTTransform(const Wavelet& wavelet, BoundaryMode mode, std::shared_ptr<Context> ctx = defaultContext())
        : w_(wavelet), mode_(mode), ctx_(std::move(ctx)) {}
    static int getMaxDecompositionLevels() { return 10; }
    int getMaximumLevels(int64_t signalLength) const {
        const int m = vw_max_levels(signalLength, static_cast<int32_t>(w_.decLo.size()));
        return m < 10 ? m : 10;
    }
    const Wavelet& getWavelet() const { return w_; }
    BoundaryMode getBoundaryMode() const { return mode_; }

    MultiLevelMODWTResult decompose(const std::vector<double>& signal) const { return decompose(signal, getMaximumLevels(static_cast<int64_t>(signal.size()))); }
    MultiLevelMODWTResult decompose(const std::vector<double>& signal, int levels) const {
        // decompose() checks finiteness, then emptiness, then the level count (:216-244)
        const int64_t n = static_cast<int64_t>(signal.size());
        if (n > 0) {
            int64_t bad = -1;
            ctx_->check(vw_first_non_finite(ctx_->get(), signal.data(), n, &bad));
        } else {
            throw InvalidSignalException("Signal cannot be empty for multi-level MODWT", "VAL_006");
        }
        if (levels < 1 || levels > getMaximumLevels(n)) throw InvalidArgumentException("Invalid number of decomposition levels", "CFG_004");
        return MultiLevelMODWTResult(decomposeBlock(signal.data(), 1, n, levels), levels, n);
    }
    MutableMultiLevelMODWTResult decomposeMutable(const std::vector<double>& signal, int levels) const { return decompose(signal, levels); }
    std::vector<double> reconstruct(const MultiLevelMODWTResult& r) const { return reconstructBlock(r.block().data(), 1, r.getSignalLength(), r.getLevels()); }
    // reconstructFromLevel / reconstructLevels (:385-468) as band masks
    std::vector<double> reconstructFromLevel(const MultiLevelMODWTResult& r, int startLevel) const {
        const int J = r.getLevels();
        if (startLevel < 1 || startLevel > J) throw InvalidArgumentException("Invalid start level", "CFG_004");
        uint32_t mask = 1u << J;
        for (int l = startLevel; l <= J; ++l) mask |= 1u << (l - 1);
        return reconstructBands(r, mask);
    }
    std::vector<double> reconstructLevels(const MultiLevelMODWTResult& r, int minLevel, int maxLevel) const {
        const int J = r.getLevels();
        if (minLevel < 1 || maxLevel > J || minLevel > maxLevel) throw InvalidArgumentException("Invalid level range for partial reconstruction", "CFG_004");
        uint32_t mask = 0;
        for (int l = minLevel; l <= maxLevel; ++l) mask |= 1u << (l - 1);
        if (J <= maxLevel) mask |= 1u << J;
        return reconstructBands(r, mask);
    }
    std::vector<double> reconstructBands(const MultiLevelMODWTResult& r, uint32_t mask) const {
        std::vector<double> x(static_cast<size_t>(r.getSignalLength()));
        ctx_->check(vw_modwt_inverse_bands_batch(ctx_->get(), r.block().data(), 1, r.getSignalLength(), r.getLevels(), w_.recLo.data(),
                                                 static_cast<int32_t>(w_.recLo.size()), w_.recHi.data(), static_cast<int32_t>(w_.recHi.size()),
                                                 static_cast<int32_t>(mode_), mask, x.data()));
        return x;
    }
    // Batch surface (new; modelled on MODWTTransform.forwardBatch): x [B][N] <-> coeffs [B][J+1][N].
    std::vector<double> decomposeBlock(const double* x, int64_t b, int64_t n, int levels, bool checkFinite = false) const {
        std::vector<double> c(static_cast<size_t>(b) * (levels + 1) * n);
        int64_t bad = -1;
        ctx_->check(vw_modwt_forward_batch(ctx_->get(), x, b, n, levels, w_.decLo.data(), static_cast<int32_t>(w_.decLo.size()), w_.decHi.data(),
                                           static_cast<int32_t>(w_.decHi.size()), static_cast<int32_t>(mode_), checkFinite ? VW_FLAG_CHECK_FINITE : 0u,
                                           c.data(), &bad));
        return c;
    }
    std::vector<double> reconstructBlock(const double* c, int64_t b, int64_t n, int levels) const {
        std::vector<double> x(static_cast<size_t>(b) * n);
        int64_t bad = -1;
        ctx_->check(vw_modwt_inverse_batch(ctx_->get(), c, b, n, levels, w_.recLo.data(), static_cast<int32_t>(w_.recLo.size()), w_.recHi.data(),
                                           static_cast<int32_t>(w_.recHi.size()), static_cast<int32_t>(mode_), 0u, x.data(), &bad));
        return x;
    }

  private:
    Wavelet w_;
    BoundaryMode mode_;
    std::shared_ptr<Context> ctx_;
};

// ---- VectorWaveSwtAdapter (swt/VectorWaveSwtAdapter.java:61) ---------------------------------------------------
class VectorWaveSwtAdapter {
  public:
    explicit VectorWaveSwtAdapter(const Wavelet& wavelet, BoundaryMode mode = BoundaryMode::PERIODIC, std::shared_ptr<Context> ctx = defaultContext())
        : w_(wavelet), mode_(mode), ctx_(ctx), t_(wavelet, mode, ctx) {}
    const Wavelet& getWavelet() const { return w_; }
    BoundaryMode getBoundaryMode() const { return mode_; }
    MutableMultiLevelMODWTResult forward(const std::vector<double>& signal, int levels) const { return t_.decomposeMutable(signal, levels); }
    MutableMultiLevelMODWTResult forward(const std::vector<double>& signal) const { return t_.decompose(signal); }
    std::vector<double> inverse(const MutableMultiLevelMODWTResult& r) const { return t_.reconstruct(r); }
    // level 0 = approximation (modwt/MutableMultiLevelMODWTResult.java:83-114)
    void applyThreshold(MutableMultiLevelMODWTResult& r, int level, double threshold, bool soft) const {
        double* p = level == 0 ? r.getMutableApproximationCoeffs() : r.getMutableDetailCoeffs(level);
        ctx_->check(vw_threshold(ctx_->get(), p, r.getSignalLength(), threshold, soft ? 1 : 0, p));
    }
    double universalThreshold(const MutableMultiLevelMODWTResult& r) const {
        double thr = 0.0;
        const int64_t n = r.getSignalLength();
        ctx_->check(vw_swt_universal_threshold_batch(ctx_->get(), r.block().data(), 1, n, r.getLevels(), std::sqrt(2 * std::log(static_cast<double>(n))), &thr));
        return thr;
    }
    void applyUniversalThreshold(MutableMultiLevelMODWTResult& r, bool soft) const {
        const double thr = universalThreshold(r);
        for (int level = 1; level <= r.getLevels(); ++level) applyThreshold(r, level, thr, soft);
    }
    std::vector<double> denoise(const std::vector<double>& signal, int levels, double threshold = -1, bool soft = true) const {
        return denoiseBlock(signal.data(), 1, static_cast<int64_t>(signal.size()), levels, threshold, soft);
    }
    // x [B][N] -> y [B][N], one downcall
    std::vector<double> denoiseBlock(const double* x, int64_t b, int64_t n, int levels, double threshold = -1, bool soft = true) const {
        std::vector<double> y(static_cast<size_t>(b) * n);
        ctx_->check(vw_swt_denoise_batch(ctx_->get(), x, b, n, levels, w_.decLo.data(), static_cast<int32_t>(w_.decLo.size()), w_.decHi.data(),
                                         static_cast<int32_t>(w_.decHi.size()), w_.recLo.data(), static_cast<int32_t>(w_.recLo.size()), w_.recHi.data(),
                                         static_cast<int32_t>(w_.recHi.size()), static_cast<int32_t>(mode_), threshold,
                                         n > 0 ? std::sqrt(2 * std::log(static_cast<double>(n))) : 0.0, soft ? 1 : 0, y.data(), nullptr));
        return y;
    }
    std::vector<double> extractLevel(const std::vector<double>& signal, int levels, int targetLevel) const {
        const MultiLevelMODWTResult r = forward(signal, levels);
        const uint32_t mask = targetLevel == 0 ? (1u << levels) : ((targetLevel >= 1 && targetLevel <= levels) ? (1u << (targetLevel - 1)) : 0u);
        return t_.reconstructBands(r, mask);
    }

  private:
    Wavelet w_;
    BoundaryMode mode_;
    std::shared_ptr<Context> ctx_;
    MultiLevelMODWTTransform t_;
};

// ---- denoising/WaveletDenoiser.java ---------------------------------------------------------------
class WaveletDenoiser {
  public:
    // :588-617, :622-634
    enum class ThresholdMethod : int32_t { UNIVERSAL = VW_THRESHOLD_UNIVERSAL, SURE = VW_THRESHOLD_SURE, MINIMAX = VW_THRESHOLD_MINIMAX, BAYES = VW_THRESHOLD_BAYES, FIXED = VW_THRESHOLD_FIXED };
    enum class ThresholdType : int32_t { SOFT = 1, HARD = 0 };

    WaveletDenoiser(const Wavelet& wavelet, BoundaryMode mode, std::shared_ptr<Context> ctx = defaultContext()) : w_(wavelet), mode_(mode), ctx_(ctx) {}
    // :96-98: DB4, PERIODIC
    static WaveletDenoiser forFinancialData(std::shared_ptr<Context> ctx = defaultContext()) { return WaveletDenoiser(getWavelet("db4"), BoundaryMode::PERIODIC, ctx); }

    // :108-139 (single-level MODWT, threshold from the detail band)
    std::vector<double> denoise(const std::vector<double>& signal, ThresholdMethod method, ThresholdType type = ThresholdType::SOFT) const {
        validate(signal);
        if (method == ThresholdMethod::FIXED) throw InvalidArgumentException("Fixed threshold method requires explicit threshold value", "CFG_005");
        return denoiseBlock(signal.data(), 1, static_cast<int64_t>(signal.size()), 0, method, type);
    }
    // :152-167 (J levels, sigma from level 1, threshold_j from sigma / sqrt(1 << (j-1)))
    std::vector<double> denoiseMultiLevel(const std::vector<double>& signal, int levels, ThresholdMethod method, ThresholdType type) const {
        validate(signal);
        return denoiseBlock(signal.data(), 1, static_cast<int64_t>(signal.size()), levels, method, type);
    }
    // :350-360
    std::vector<double> denoiseFixed(const std::vector<double>& signal, double threshold, ThresholdType type) const {
        validate(signal);
        return denoiseBlock(signal.data(), 1, static_cast<int64_t>(signal.size()), 0, ThresholdMethod::FIXED, type, threshold);
    }
    // x [B][N] -> y [B][N], one downcall; levels == 0 selects the single-level form; thresholds (optional) [B][max(levels,1)]
    std::vector<double> denoiseBlock(const double* x, int64_t b, int64_t n, int levels, ThresholdMethod method, ThresholdType type,
                                     double fixedThreshold = 0.0, std::vector<double>* thresholds = nullptr) const {
        const int nl = levels > 0 ? levels : 1;
        std::vector<double> y(static_cast<size_t>(b) * n);
        if (thresholds != nullptr) thresholds->assign(static_cast<size_t>(b) * nl, 0.0);
        ctx_->check(vw_wavelet_denoise_batch(ctx_->get(), x, b, n, nl, w_.decLo.data(), static_cast<int32_t>(w_.decLo.size()), w_.decHi.data(),
                                             static_cast<int32_t>(w_.decHi.size()), w_.recLo.data(), static_cast<int32_t>(w_.recLo.size()), w_.recHi.data(),
                                             static_cast<int32_t>(w_.recHi.size()), static_cast<int32_t>(mode_), levels == 0 ? VW_FLAG_SINGLE_LEVEL : 0u,
                                             static_cast<int32_t>(method), type == ThresholdType::SOFT ? 1 : 0, fixedThreshold,
                                             n > 0 ? std::log(static_cast<double>(n)) : 0.0, y.data(), thresholds != nullptr ? thresholds->data() : nullptr));
        return y;
    }

  private:
    // MODWTTransform.forward / MultiLevelMODWTTransform.decompose validate before transforming
    void validate(const std::vector<double>& signal) const {
        if (signal.empty()) throw InvalidSignalException("Signal cannot be empty", "VAL_006");
        int64_t bad = -1;
        ctx_->check(vw_first_non_finite(ctx_->get(), signal.data(), static_cast<int64_t>(signal.size()), &bad));
    }
    Wavelet w_;
    BoundaryMode mode_;
    std::shared_ptr<Context> ctx_;
};

// modwt/streaming/MODWTStreamingDenoiser.java: block-by-block denoiser whose noise rings live in GPU memory.
// `streams` > 1 (not in the reference) runs that many denoisers in lock step, one block per stream per call.
class MODWTStreamingDenoiser {
  public:
    enum class NoiseEstimation : int32_t { MAD = VW_NOISE_MAD, STD = VW_NOISE_STD, FIXED = VW_NOISE_FIXED };  // :58-65

    // :367-469, same defaults and argument checks
    class Builder {
      public:
        Builder& wavelet(const Wavelet& w) { wavelet_ = w; return *this; }
        Builder& boundaryMode(BoundaryMode m) { mode_ = m; return *this; }
        Builder& bufferSize(int v) { if (v <= 0) throw InvalidArgumentException("Buffer size must be positive", ""); bufferSize_ = v; return *this; }
        Builder& thresholdType(WaveletDenoiser::ThresholdType t) { type_ = t; return *this; }
        Builder& thresholdMethod(WaveletDenoiser::ThresholdMethod m) { method_ = m; return *this; }
        Builder& thresholdMultiplier(double v) { if (!(v > 0)) throw InvalidArgumentException("Threshold multiplier must be positive", ""); multiplier_ = v; return *this; }
        Builder& noiseEstimation(NoiseEstimation e) { estimation_ = e; return *this; }
        Builder& noiseWindowSize(int v) { if (v <= 0) throw InvalidArgumentException("Noise window size must be positive", ""); window_ = v; return *this; }
        Builder& streams(int64_t v) { if (v <= 0) throw InvalidArgumentException("Stream count must be positive", ""); streams_ = v; return *this; }
        Builder& context(std::shared_ptr<Context> c) { ctx_ = c; return *this; }
        MODWTStreamingDenoiser build() const { return MODWTStreamingDenoiser(*this); }

      private:
        friend class MODWTStreamingDenoiser;
        Wavelet wavelet_ = getWavelet("db4");
        BoundaryMode mode_ = BoundaryMode::PERIODIC;
        int bufferSize_ = 256;
        WaveletDenoiser::ThresholdType type_ = WaveletDenoiser::ThresholdType::SOFT;
        WaveletDenoiser::ThresholdMethod method_ = WaveletDenoiser::ThresholdMethod::UNIVERSAL;
        double multiplier_ = 1.0;
        NoiseEstimation estimation_ = NoiseEstimation::MAD;
        int window_ = 1024;
        int64_t streams_ = 1;
        std::shared_ptr<Context> ctx_ = defaultContext();
    };

    MODWTStreamingDenoiser(MODWTStreamingDenoiser&& o) noexcept : cfg_(o.cfg_), state_(o.state_), closed_(o.closed_) { o.state_ = nullptr; }
    MODWTStreamingDenoiser(const MODWTStreamingDenoiser&) = delete;
    MODWTStreamingDenoiser& operator=(const MODWTStreamingDenoiser&) = delete;
    ~MODWTStreamingDenoiser() { close(); }

    // :93-127; samples holds one block per stream, [streams][n] row-major
    std::vector<double> denoise(const std::vector<double>& samples) {
        if (closed_) throw std::logic_error("Denoiser is closed");  // IllegalStateException
        if (samples.empty()) throw InvalidArgumentException("Samples cannot be null or empty", "");
        if (samples.size() % static_cast<size_t>(cfg_.streams_) != 0) throw InvalidArgumentException("expected one block per stream", "");
        const int64_t n = static_cast<int64_t>(samples.size()) / cfg_.streams_;
        int64_t bad = -1;
        cfg_.ctx_->check(vw_first_non_finite(cfg_.ctx_->get(), samples.data(), static_cast<int64_t>(samples.size()), &bad));
        std::vector<double> y(samples.size());
        const Wavelet& w = cfg_.wavelet_;
        const double logN = std::log(static_cast<double>(n));
        cfg_.ctx_->check(vw_streaming_denoise_batch(cfg_.ctx_->get(), state_, samples.data(), cfg_.streams_, n, w.decLo.data(), static_cast<int32_t>(w.decLo.size()),
                                                    w.decHi.data(), static_cast<int32_t>(w.decHi.size()), w.recLo.data(), static_cast<int32_t>(w.recLo.size()),
                                                    w.recHi.data(), static_cast<int32_t>(w.recHi.size()), static_cast<int32_t>(cfg_.mode_), 0u,
                                                    static_cast<int32_t>(cfg_.method_), cfg_.type_ == WaveletDenoiser::ThresholdType::SOFT ? 1 : 0,
                                                    cfg_.multiplier_, static_cast<int32_t>(cfg_.estimation_), logN, std::sqrt(2.0 * logN), y.data(), nullptr));
        return y;
    }
    // :332-334, one level per stream
    std::vector<double> getEstimatedNoiseLevels() const {
        std::vector<double> lv(static_cast<size_t>(cfg_.streams_), 0.0);
        if (state_ != nullptr) cfg_.ctx_->check(vw_stream_state_noise_levels(cfg_.ctx_->get(), state_, lv.data()));
        return lv;
    }
    double getEstimatedNoiseLevel() const { return getEstimatedNoiseLevels()[0]; }
    int64_t getSamplesProcessed() const { return state_ != nullptr ? vw_stream_state_samples_processed(state_) : finalSamples_; }  // :341-343
    bool isClosed() const { return closed_; }
    void close() {  // :351-355
        if (state_ != nullptr) {
            finalSamples_ = vw_stream_state_samples_processed(state_);
            vw_stream_state_destroy(cfg_.ctx_->get(), state_);
            state_ = nullptr;
        }
        closed_ = true;
    }

  private:
    explicit MODWTStreamingDenoiser(const Builder& b) : cfg_(b) {
        cfg_.ctx_->check(vw_stream_state_create(cfg_.ctx_->get(), cfg_.streams_, cfg_.window_, &state_));
    }
    Builder cfg_;
    vw_stream_state* state_ = nullptr;
    bool closed_ = false;
    int64_t finalSamples_ = 0;
};

// modwt/BatchSIMDMODWT.java: single-level PERIODIC MODWT over a sample-major batch, soa[t * batch + b]
struct BatchSIMDMODWT {
    // :64-81
    static void batchMODWTSoA(const std::vector<double>& soaSignals, std::vector<double>& soaApprox, std::vector<double>& soaDetail, const Wavelet& wavelet,
                              int64_t batchSize, int64_t signalLength, std::shared_ptr<Context> ctx = defaultContext()) {
        const size_t count = static_cast<size_t>(batchSize) * static_cast<size_t>(signalLength);
        if (soaSignals.size() < count) throw std::out_of_range("soaSignals shorter than batchSize * signalLength");
        soaApprox.resize(count);
        soaDetail.resize(count);
        ctx->check(vw_batch_modwt_soa(ctx->get(), soaSignals.data(), batchSize, signalLength, wavelet.decLo.data(), static_cast<int32_t>(wavelet.decLo.size()),
                                      wavelet.decHi.data(), static_cast<int32_t>(wavelet.decHi.size()), wavelet.name == "haar" ? 1 : 0, soaApprox.data(),
                                      soaDetail.data()));
    }
    // :282-291: soa[t * batch + b] = signals[b][t]
    static void convertToSoA(const std::vector<double>& signals, int64_t batchSize, int64_t signalLength, std::vector<double>& soaOutput,
                             std::shared_ptr<Context> ctx = defaultContext()) {
        soaOutput.resize(signals.size());
        ctx->check(vw_transpose(ctx->get(), signals.data(), batchSize, signalLength, soaOutput.data()));
    }
    // :299-308
    static void convertFromSoA(const std::vector<double>& soaData, int64_t batchSize, int64_t signalLength, std::vector<double>& output,
                               std::shared_ptr<Context> ctx = defaultContext()) {
        output.resize(soaData.size());
        ctx->check(vw_transpose(ctx->get(), soaData.data(), signalLength, batchSize, output.data()));
    }
};

// swt/SWTResult.java: dense container (level index 0-based as in the reference) and its sparse form, whose
// compaction and expansion run on the GPU
class SparseSWTResult;
class SWTResult {
  public:
    SWTResult(std::vector<double> approximation, std::vector<std::vector<double>> details, int levels, std::shared_ptr<Context> ctx = defaultContext())
        : a_(std::move(approximation)), d_(std::move(details)), levels_(levels), ctx_(ctx) {
        // :43-60
        if (static_cast<int>(d_.size()) != levels) throw std::invalid_argument("Details array length must match levels");
        for (const auto& d : d_)
            if (d.size() != a_.size()) throw std::invalid_argument("All detail levels must have same length as approximation");
    }
    std::vector<double> getApproximation() const { return a_; }
    std::vector<double> getDetail(int level) const { checkLevel(level); return d_[static_cast<size_t>(level)]; }
    int getLevels() const { return levels_; }
    int getSignalLength() const { return static_cast<int>(a_.size()); }
    double getDetailEnergy(int level) const { checkLevel(level); return energy(d_[static_cast<size_t>(level)]); }  // :124-147
    double getApproximationEnergy() const { return energy(a_); }
    double getTotalEnergy() const {  // :168-181
        double e = getApproximationEnergy();
        for (int l = 0; l < levels_; ++l) e += getDetailEnergy(l);
        return e;
    }
    inline SparseSWTResult toSparse(double threshold) const;  // :190-192

  private:
    friend class SparseSWTResult;
    void checkLevel(int level) const {
        if (level < 0 || level >= levels_) throw std::invalid_argument("Level must be between 0 and " + std::to_string(levels_ - 1));
    }
    static double energy(const std::vector<double>& v) {
        double e = 0.0;
        for (double c : v) e += c * c;
        return e;
    }
    std::vector<double> a_;
    std::vector<std::vector<double>> d_;
    int levels_;
    std::shared_ptr<Context> ctx_;
};

// SWTResult.SparseSWTResult (:198-257)
class SparseSWTResult {
  public:
    SparseSWTResult(const SWTResult& full, double threshold) : a_(full.a_), levels_(full.levels_), n_(static_cast<int64_t>(full.a_.size())), threshold_(threshold), ctx_(full.ctx_) {
        std::vector<double> block(static_cast<size_t>(levels_ + 1) * n_);
        for (int l = 0; l < levels_; ++l) std::copy(full.d_[static_cast<size_t>(l)].begin(), full.d_[static_cast<size_t>(l)].end(), block.begin() + static_cast<int64_t>(l) * n_);
        std::vector<int64_t> counts(static_cast<size_t>(levels_), 0);
        if (n_ > 0) ctx_->check(vw_swt_sparsify_batch(ctx_->get(), block.data(), 1, n_, levels_, threshold_, counts.data(), nullptr, nullptr, nullptr));
        offsets_.assign(static_cast<size_t>(levels_) + 1, 0);
        for (int l = 0; l < levels_; ++l) offsets_[static_cast<size_t>(l) + 1] = offsets_[static_cast<size_t>(l)] + counts[static_cast<size_t>(l)];
        indices_.resize(static_cast<size_t>(offsets_.back()));
        values_.resize(static_cast<size_t>(offsets_.back()));
        if (offsets_.back() > 0)
            ctx_->check(vw_swt_sparsify_batch(ctx_->get(), block.data(), 1, n_, levels_, threshold_, nullptr, offsets_.data(), indices_.data(), values_.data()));
    }
    // :225-231
    SWTResult toFull() const {
        std::vector<double> block(static_cast<size_t>(levels_ + 1) * n_, 0.0);
        if (n_ > 0) ctx_->check(vw_swt_densify_batch(ctx_->get(), offsets_.data(), indices_.data(), values_.data(), 1, n_, levels_, block.data()));
        std::vector<std::vector<double>> d(static_cast<size_t>(levels_));
        for (int l = 0; l < levels_; ++l) d[static_cast<size_t>(l)].assign(block.begin() + static_cast<int64_t>(l) * n_, block.begin() + static_cast<int64_t>(l + 1) * n_);
        return SWTResult(a_, d, levels_, ctx_);
    }
    // :238-247
    double getCompressionRatio() const {
        const int64_t original = n_ * (levels_ + 1);
        const int64_t compressed = n_ + 2 * offsets_.back();
        return static_cast<double>(original) / static_cast<double>(compressed);
    }
    double getThreshold() const { return threshold_; }
    int64_t getNonZeroCount(int level) const { return offsets_[static_cast<size_t>(level) + 1] - offsets_[static_cast<size_t>(level)]; }
    const std::vector<int32_t>& indices() const { return indices_; }
    const std::vector<double>& values() const { return values_; }
    const std::vector<int64_t>& offsets() const { return offsets_; }

  private:
    std::vector<double> a_;
    int levels_;
    int64_t n_;
    double threshold_;
    std::shared_ptr<Context> ctx_;
    std::vector<int64_t> offsets_;
    std::vector<int32_t> indices_;
    std::vector<double> values_;
};

inline SparseSWTResult SWTResult::toSparse(double threshold) const { return SparseSWTResult(*this, threshold); }

}  // namespace vectorwave
