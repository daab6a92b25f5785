// vw_kernels.hip.h -- gfx950 device code of the MODWT / SWT engine.
//
// The transform is an a-trous stencil: at level j (stride s = 2^(j-1))
//   V_j[t] = sum_l lo[l] * V_{j-1}[t - l*s]      W_j[t] = sum_l hi[l] * V_{j-1}[t - l*s]
// and the inverse walks back with a look-ahead
//   R_{j-1}[t] = sum_l ( lo~[l] * R_j[t + l*s] + hi~[l] * W_j[t + l*s] ).
// It is HBM-bound fp64 work (no MFMA): every kernel here is built around
// coalesced 16-byte-per-lane global accesses, an LDS-resident signal tile with a
// halo sized for the dilated support, taps read through the scalar unit, and all
// levels of a group fused per tile so intermediates never travel to HBM.
//
// Arithmetic contract (the reference's scalar Java path): every product and every
// sum is rounded separately, taps are accumulated in ascending l starting from
// +0.0.  FP contraction is therefore disabled for the whole translation unit.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#pragma clang fp contract(off)

namespace vw {

constexpr int kThreads = 256;           // 4 wavefronts per workgroup
constexpr int kChunk = 2 * kThreads;    // samples covered by one pass of the workgroup (a pair per lane)
constexpr int kMaxTaps = 104;           // longest reference filter is coif17 (102 taps)
constexpr int kMaxLevels = 12;          // reference cap is 9

constexpr int kPeriodic = 0;
constexpr int kZeroPad = 1;
constexpr int kSymmetric = 2;

// Filter pair, already multiplied by 1/sqrt(2) on the host.  Passed by value in
// the kernel argument segment so the taps are fetched with scalar loads.  Two
// spare slots let the tap loop prefetch one tap ahead without a bounds check.
struct Taps {
    double lo[kMaxTaps + 2];
    double hi[kMaxTaps + 2];
};

struct FwdArgs {
    const double* in;      // V_{lev0-1}: rows of length n
    int64_t in_stride;     // elements between rows of `in`
    double* wout;          // detail bands: row r, level j at wout + r*w_row_stride + (j-1)*w_band_stride
    int64_t w_row_stride;
    int64_t w_band_stride;
    double* vout;          // V_{lev1}
    int64_t v_stride;
    int64_t n;
    int64_t n_store;       // == n; 0 in the store-ablation build of the tuning harness
    int64_t nrows;
    int32_t ntiles;
    int32_t lev0, lev1;    // levels computed by this launch (inclusive)
    int32_t lh, lg;        // tap counts of lo / hi
    int32_t pstart[kMaxLevels + 1];  // first buffer position at which V_j is computed (even)
    // streaming kernel only: a workgroup walks `nchunks` consecutive tiles of its row and carries the
    // last hlen[j] samples of V_{j-1} (the left context level j needs) from tile to tile in LDS
    int32_t tiles_per_row, nchunks;
    // k_fwd_stream only: the row's tiles are dealt evenly over its `ntiles` runs: run_base tiles each, one more in
    // the first run_rem runs (a short last run would pay a whole halo for little work and end early)
    int32_t run_base, run_rem;
    int32_t hoff[kMaxLevels + 2], hlen[kMaxLevels + 2];
    int32_t ndry, hist_cap;  // k_fwd_sdry: warm-up tiles per run, doubles of history storage
    unsigned long long* first_bad;   // nullptr = no finite check
    int32_t nt_store;      // results are stored with the streaming hint (launches far larger than the caches)
    Taps taps;
};

struct InvArgs {
    const double* rin;     // R_{lev1}: rows of length n (a_J for the top group)
    int64_t rin_stride;
    const double* win;     // detail bands, addressed like FwdArgs::wout
    int64_t w_row_stride;
    int64_t w_band_stride;
    double* rout;          // R_{lev0-1}
    int64_t rout_stride;
    int64_t n;
    int64_t n_store;       // == n; 0 in the store-ablation build of the tuning harness
    int64_t nrows;
    int32_t ntiles;
    int32_t lev0, lev1;    // levels undone by this launch, from lev1 down to lev0
    int32_t lh;            // loop bound (low-pass reconstruction length)
    int32_t lg;            // k_inv_sym only: high-pass reconstruction length (the other kernels index both filters with l < lh)
    int32_t pend[kMaxLevels + 1];    // R_{j} is valid on buffer positions [0, pend[j]) (even)
    // k_inv_sdry only: runs of `nchunks` tiles walked right to left, `ndry` warm-up tiles, and the first
    // hlen[j] samples of R_j and W_j carried from tile to tile (hist_cap doubles of storage each)
    int32_t tiles_per_row, nchunks, ndry, hist_cap;
    int32_t hoff[kMaxLevels + 2], hlen[kMaxLevels + 2];
    uint32_t band_mask;    // bit (j-1): use W_j, else zeros; bit 31: use rin, else zeros
    int32_t thr_mode;      // 0 none, else a mode of apply_thr: applied to W_j as it is loaded
    const double* thr;     // threshold of row r, level j at thr[r*thr_row_stride + (j-1)*thr_level_stride] (thr_mode != 0)
    int64_t thr_row_stride, thr_level_stride;
    unsigned long long* first_bad;  // flat index into the coefficient block of the first NaN/Inf
    int32_t check_rin;     // rin is the a_J band of the block (top group): include it in the check
    int32_t nt_store;      // results are stored with the streaming hint (launches far larger than the caches)
    int32_t ablate;        // tuning build only (bit 0: no arithmetic, bit 1: no band loads); always 0 in the product
    Taps taps;
};

// Parts of a kernel that the tuning build (-DVW_TUNING, tools/ only) can switch off to time what is left.  In the
// product build the test is a compile-time false and the code below it is the only code.
#ifdef VW_TUNING
#define VW_ABLATED(args, bit) (((args).ablate >> (bit)) & 1)
#else
#define VW_ABLATED(args, bit) false
#endif

// ---------------------------------------------------------------------------
// helpers
// ---------------------------------------------------------------------------

// Workgroups are dealt round-robin over the 8 XCDs, so ids b and b+8 share an
// L2.  Remap so that each XCD walks a contiguous run of (row, tile) ids: adjacent
// tiles of one row overlap in their halo and then hit in the same L2.  Bijective
// for any grid size.
__device__ __forceinline__ int64_t xcd_remap(int64_t bid, int64_t nblk) {
    const int64_t q = nblk >> 3, r = nblk & 7;
    const int64_t x = bid & 7, k = bid >> 3;
    const int64_t base = (x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q;
    return base + k;
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains the
// vector-memory counter, i.e. it would wait for every global store (and every
// prefetch load) issued so far -- here those must stay in flight across the
// per-level barriers.  LDS is coherent within the CU, so "my LDS operations have
// completed" + s_barrier is all the tile hand-off between waves needs.
// 16-byte global store of a result that this kernel never reads back.  nt: the launch writes far more than
// the last-level caches hold, so the lines are marked streaming (measured on 69 GB launches: the inverse's
// memory-only time drops 5 %); small launches keep their results cacheable for the kernel that reads them next.
typedef double vw_f64x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void gstore2(double* p, double2 v, int nt) {
    if (nt) {
        vw_f64x2 x;
        x.x = v.x;
        x.y = v.y;
#ifdef VW_TUNING
        // tuning build: the other cache-policy bits (plan option nt_stores = 2..5); the s_nop keeps a later VALU write of
        // the data registers two wait states away from the store, which the compiler does not do for inline assembly
        if (nt == 2) {
            asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(p), "v"(x) : "memory");
            return;
        }
        if (nt == 3) {
            asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" : : "v"(p), "v"(x) : "memory");
            return;
        }
        if (nt == 4) {
            asm volatile("global_store_dwordx4 %0, %1, off sc1 nt\n\ts_nop 1" : : "v"(p), "v"(x) : "memory");
            return;
        }
        if (nt == 5) {
            asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt\n\ts_nop 1" : : "v"(p), "v"(x) : "memory");
            return;
        }
#endif
        __builtin_nontemporal_store(x, reinterpret_cast<vw_f64x2*>(p));
    } else {
        *reinterpret_cast<double2*>(p) = v;
    }
}

// Global access modes of the long-context kernels (template parameter VEC2 there): 0 = one sample per access,
// 1 = 16-byte accesses (even n, 16-byte-aligned rows), 2 = 16-byte accesses at 8-byte alignment.  Mode 2 serves odd
// row lengths, where every second row starts 8 bytes off: gfx950 executes a dwordx4 global access at any dword
// alignment, so only the pair that straddles the end of the row needs the one-sample form.
typedef double vw_f64x2u __attribute__((ext_vector_type(2), aligned(8)));
template <int VM>
__device__ __forceinline__ double2 gload_pair(const double* __restrict__ row, int64_t t, int64_t n) {  // 0 <= t; zeros beyond n
    double2 x = make_double2(0.0, 0.0);
    if (VM == 1) {
        if (t < n) x = *reinterpret_cast<const double2*>(row + t);
    } else if (VM == 2) {
        if (t + 1 < n) {
            const vw_f64x2u v = *reinterpret_cast<const vw_f64x2u*>(row + t);
            x.x = v.x;
            x.y = v.y;
        } else if (t < n) {
            x.x = row[t];
        }
    } else {
        if (t < n) x.x = row[t];
        if (t + 1 < n) x.y = row[t + 1];
    }
    return x;
}
__device__ __forceinline__ int64_t mod_wrap(int64_t i, int64_t n);

// One double from a wave-uniform address through the scalar unit.  As a vector load the value would sit behind
// every older vector-memory request of the wave (the counter retires in order), so waiting for it also drains
// the prefetched bands -- the LDS-DMA ring would lose its two levels of run-ahead to one threshold per level.
// Read only: written by an earlier kernel.
__device__ __forceinline__ double uniform_load_f64(const double* p) {
    const uint64_t u = reinterpret_cast<uint64_t>(p);
    // (the builtin returns int: without the casts the low word would be sign-extended over the high one)
    const uint32_t uhi = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(u >> 32)));
    const uint32_t ulo = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(u & 0xffffffffu)));
    const uint64_t su = (static_cast<uint64_t>(uhi) << 32) | static_cast<uint64_t>(ulo);
    double v;
    asm volatile("s_load_dwordx2 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(su) : "memory");
    return v;
}
template <int VM>
__device__ __forceinline__ double2 gload_pair_wrapped(const double* __restrict__ row, int64_t t, int64_t n) {  // any t, periodic
    double2 x;
    const int64_t tw = mod_wrap(t, n);
    if (VM == 1) {
        x = *reinterpret_cast<const double2*>(row + tw);
    } else if (VM == 2 && tw + 1 < n) {
        const vw_f64x2u v = *reinterpret_cast<const vw_f64x2u*>(row + tw);
        x.x = v.x;
        x.y = v.y;
    } else {
        x.x = row[tw];
        x.y = row[tw + 1 < n ? tw + 1 : mod_wrap(t + 1, n)];
    }
    return x;
}
template <int VM>
__device__ __forceinline__ void gstore_pair(double* __restrict__ row, int64_t t, int64_t ns, double v0, double v1, int nt) {
    if (VM == 1) {
        if (t < ns) gstore2(row + t, make_double2(v0, v1), nt);
    } else if (VM == 2) {
        if (t + 1 < ns) {
            vw_f64x2u x;
            x.x = v0;
            x.y = v1;
            if (nt)
                __builtin_nontemporal_store(x, reinterpret_cast<vw_f64x2u*>(row + t));
            else
                *reinterpret_cast<vw_f64x2u*>(row + t) = x;
        } else if (t < ns) {
            row[t] = v0;
        }
    } else {
        if (t < ns) row[t] = v0;
        if (t + 1 < ns) row[t + 1] = v1;
    }
}

__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

__device__ __forceinline__ int64_t mod_wrap(int64_t i, int64_t n) {
    int64_t m = i % n;
    return m < 0 ? m + n : m;
}

// util/MathUtils.symmetricBoundaryExtension of the reference.
__device__ __forceinline__ int64_t sym_reflect(int64_t idx, int64_t n) {
    if (idx >= 0 && idx < n) return idx;
    const int64_t period = 2 * n;
    idx = ((idx % period) + period) % period;
    if (idx >= n) idx = period - idx - 1;
    return idx;
}

__device__ __forceinline__ bool finite_f64(double v) {
    return (__double_as_longlong(v) & 0x7ff0000000000000LL) != 0x7ff0000000000000LL;
}

// The reference thresholds in two forms that differ only for NaN operands (a NaN threshold or coefficient):
// mode 1: soft, MutableMultiLevelMODWTResult.applyThreshold (modwt/MutableMultiLevelMODWTResult.java:97-114):
//         |c| > thr ? signum(c) * (|c| - thr) : 0.0            -- a NaN comparison zeroes the coefficient
// mode 2: hard, every caller:  |c| <= thr ? 0.0 : c
// mode 3: soft, WaveletDenoiser.applyThreshold / VectorOps.Denoising.softThreshold
//         (denoising/WaveletDenoiser.java:554-559, internal/VectorOps.java:454-459):
//         |c| <= thr ? 0.0 : signum(c) * (|c| - thr)           -- a NaN comparison propagates the NaN
__device__ __forceinline__ double apply_thr(double c, double thr, int mode) {
    const double a = fabs(c);
    if (mode == 2) return (a <= thr) ? 0.0 : c;
    const bool keep = (mode == 1) ? (a > thr) : !(a <= thr);
    if (!keep) return 0.0;
    const double m = a - thr;
    // signum(c) * m with signum in {+1, -1, +-0, NaN}
    return c > 0.0 ? m : (c < 0.0 ? -m : c * m);
}

// Thresholding when coefficient and threshold are known to be finite and the threshold is not negative: the three
// forms of apply_thr then agree, and soft thresholding is max(|c| - thr, 0) with the sign of c -- three instructions
// instead of a dozen.  A coefficient that shrinks to zero may come out as -0.0 where apply_thr gives +0.0; callers use
// this only where the value feeds an inverse cascade and nothing else: every reconstruction sum starts from +0.0,
// (+0.0) + (-0.0) is +0.0 and a nonzero term is not changed by a zero of either sign, so the sign of a zero
// coefficient cannot reach an output.
__device__ __forceinline__ double shrink_finite(double c, double thr, bool soft) {
    if (!soft) return (fabs(c) <= thr) ? 0.0 : c;
    return __builtin_copysign(__builtin_fmax(fabs(c) - thr, 0.0), c);
}

// ---------------------------------------------------------------------------
// fused forward: levels lev0..lev1 of one (row, tile) per workgroup
// ---------------------------------------------------------------------------
//
// LDS buffer: M = (HPT + PPT) * 512 doubles, position p <-> sample t = t0 - HALO + p.
// Each lane owns the aligned pair (p, p+1), p = chunk*512 + 2*lane, of every
// chunk: PPT tile chunks (V and W computed) and HPT halo chunks (V only, and only
// where the shrinking halo still needs it).  Results stay in registers across
// the barrier and overwrite the buffer in place, so one buffer serves all levels.

// Halo chunks are switched on per WAVE (a scalar branch), never per lane: a lane
// of an active wave that lies left of the needed zone recomputes the first
// needed pair instead (its address is clamped, its result is not written back).
// Per-lane predicates inside the tap loop would split every tap into
// exec-masked blocks with their own LDS waits.
// FMA = false is the reference's arithmetic (product and sum rounded separately: bit parity with the
// Java scalar path).  FMA = true fuses each multiply-add (one rounding per tap instead of two): half
// the fp64 instructions, results within ~1e-15 relative of the exact path, not bit-identical.
template <bool FMA>
__device__ __forceinline__ double mac(double acc, double x, double f) {
    return FMA ? __builtin_fma(x, f, acc) : acc + x * f;
}

template <int PPT, int HPT, bool S1, bool DO_LO, bool DO_HI, bool FMA, bool FIRST = false, int CH = kChunk, bool HALO_HERE = true>
__device__ __forceinline__ void fwd_tap(const char* lds, int baseT, const int (&hbase)[HPT], int off, double flo,
                                        double fhi, double (&av)[PPT][2], double (&aw)[PPT][2],
                                        double (&ah)[HPT][2], double (&xp)[PPT], double (&xph)[HPT],
                                        const bool (&wact)[HPT]) {
    double2 xt[PPT];
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        if (S1) {
            xt[k].x = *reinterpret_cast<const double*>(lds + baseT + k * (CH * 8) - off);
            xt[k].y = xp[k];
            xp[k] = xt[k].x;
        } else {
            xt[k] = *reinterpret_cast<const double2*>(lds + baseT + k * (CH * 8) - off);
        }
    }
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        // FIRST: the reference's `sum = 0.0; sum += x*f` -- the add of +0.0 is kept (it turns a -0.0
        // product into +0.0), only the separate zero-initialisation of the accumulator is saved
        if (DO_LO) {
            av[k][0] = mac<FMA>(FIRST ? 0.0 : av[k][0], xt[k].x, flo);
            av[k][1] = mac<FMA>(FIRST ? 0.0 : av[k][1], xt[k].y, flo);
        }
        if (DO_HI) {
            aw[k][0] = mac<FMA>(FIRST ? 0.0 : aw[k][0], xt[k].x, fhi);
            aw[k][1] = mac<FMA>(FIRST ? 0.0 : aw[k][1], xt[k].y, fhi);
        }
    }
    if (DO_LO && HALO_HERE) {
#pragma unroll
        for (int c = 0; c < HPT; ++c) {
            if (wact[c]) {  // wave-uniform
                double x0, x1;
                if (S1) {
                    x0 = *reinterpret_cast<const double*>(lds + hbase[c] - off);
                    x1 = xph[c];
                    xph[c] = x0;
                } else {
                    const double2 x = *reinterpret_cast<const double2*>(lds + hbase[c] - off);
                    x0 = x.x;
                    x1 = x.y;
                }
                ah[c][0] = mac<FMA>(ah[c][0], x0, flo);
                ah[c][1] = mac<FMA>(ah[c][1], x1, flo);
            }
        }
    }
}

// LT > 0: both filters have exactly LT taps; the tap loop is unrolled completely and the 2*LT tap
// values are fetched in one go, so no scalar load (which shares its completion counter with the LDS
// reads) sits inside the loop.  LT == 0: any lengths, one tap per iteration, taps fetched one ahead.
template <int PPT, int HPT, bool S1, bool FMA, int CH = kChunk, int LT = 0, int HOFF = HPT * kChunk>
__device__ __forceinline__ void fwd_level(const char* lds, const FwdArgs& a, int s, int tid,
                                          double (&av)[PPT][2], double (&aw)[PPT][2],
                                          double (&ah)[HPT][2], const int (&hbase)[HPT], const bool (&wact)[HPT]) {
    constexpr int HALO = HOFF;  // buffer position of the tile
    const int baseT = (HALO + 2 * tid) * 8;
    double xp[PPT], xph[HPT];
    if (S1) {
#pragma unroll
        for (int k = 0; k < PPT; ++k)
            xp[k] = *reinterpret_cast<const double*>(lds + baseT + k * (CH * 8) + 8);
#pragma unroll
        for (int c = 0; c < HPT; ++c)
            xph[c] = wact[c] ? *reinterpret_cast<const double*>(lds + hbase[c] + 8) : 0.0;
    }
    const int sb = s * 8;
    if (LT > 0) {
        double tl[LT > 0 ? LT : 1], th[LT > 0 ? LT : 1];
#pragma unroll
        for (int i = 0; i < LT; ++i) {
            tl[i] = a.taps.lo[i];
            th[i] = a.taps.hi[i];
        }
        fwd_tap<PPT, HPT, S1, true, true, FMA, true, CH, false>(lds, baseT, hbase, 0, tl[0], th[0], av, aw, ah, xp, xph, wact);
#pragma unroll
        for (int i = 1; i < LT; ++i)
            fwd_tap<PPT, HPT, S1, true, true, FMA, false, CH, false>(lds, baseT, hbase, i * sb, tl[i], th[i], av, aw, ah, xp,
                                                                     xph, wact);
        // The halo chunks (low-pass only, wave-uniform switch) after the tile, each behind ONE branch with its
        // operands requested together: inside the tap loop every tap cost every wave a branch, and a wave with an
        // active halo chunk waited for its tile operands tap by tap.
#pragma unroll
        for (int c = 0; c < HPT; ++c) {
            if (wact[c]) {
                if (S1) {
                    double d[LT > 0 ? LT + 1 : 1];
#pragma unroll
                    for (int m = 1; m <= LT; ++m) d[m] = *reinterpret_cast<const double*>(lds + hbase[c] + 8 - m * 8);
                    d[0] = xph[c];
#pragma unroll
                    for (int i = 0; i < LT; ++i) {
                        ah[c][0] = mac<FMA>(ah[c][0], d[i + 1], tl[i]);
                        ah[c][1] = mac<FMA>(ah[c][1], d[i], tl[i]);
                    }
                } else {
                    double2 x[LT > 0 ? LT : 1];
#pragma unroll
                    for (int i = 0; i < LT; ++i) x[i] = *reinterpret_cast<const double2*>(lds + hbase[c] - i * sb);
#pragma unroll
                    for (int i = 0; i < LT; ++i) {
                        ah[c][0] = mac<FMA>(ah[c][0], x[i].x, tl[i]);
                        ah[c][1] = mac<FMA>(ah[c][1], x[i].y, tl[i]);
                    }
                }
            }
        }
        return;
    }
    const int lh = a.lh, lg = a.lg;
    const int lmin = lh < lg ? lh : lg;
    int off = 0;
    int l = 0;
    double flo = a.taps.lo[0], fhi = a.taps.hi[0];
    if (lmin > 0) {  // tap 0 defines the tile accumulators
        const double flo_n = a.taps.lo[1], fhi_n = a.taps.hi[1];
        fwd_tap<PPT, HPT, S1, true, true, FMA, true, CH>(lds, baseT, hbase, off, flo, fhi, av, aw, ah, xp, xph, wact);
        off += sb;
        flo = flo_n;
        fhi = fhi_n;
        l = 1;
    }
    for (; l < lmin; ++l) {
        const double flo_n = a.taps.lo[l + 1], fhi_n = a.taps.hi[l + 1];
        fwd_tap<PPT, HPT, S1, true, true, FMA, false, CH>(lds, baseT, hbase, off, flo, fhi, av, aw, ah, xp, xph, wact);
        off += sb;
        flo = flo_n;
        fhi = fhi_n;
    }
    for (; l < lh; ++l) {  // low-pass longer than high-pass
        const double flo_n = a.taps.lo[l + 1];
        fwd_tap<PPT, HPT, S1, true, false, FMA, false, CH>(lds, baseT, hbase, off, flo, 0.0, av, aw, ah, xp, xph, wact);
        off += sb;
        flo = flo_n;
    }
    for (; l < lg; ++l) {  // high-pass longer than low-pass
        const double fhi_n = a.taps.hi[l + 1];
        fwd_tap<PPT, HPT, S1, false, true, FMA, false, CH>(lds, baseT, hbase, off, 0.0, fhi, av, aw, ah, xp, xph, wact);
        off += sb;
        fhi = fhi_n;
    }
}

template <int PPT, int HPT, int MODE, bool VEC2, bool FMA>
__global__ void __launch_bounds__(kThreads) k_fwd_fused(const FwdArgs a) {
    extern __shared__ double2 vw_smem[];
    constexpr int HALO = HPT * kChunk;
    constexpr int T = PPT * kChunk;
    constexpr int M = HALO + T;
    // two buffers: level j reads one and writes V_j into the other, so a single barrier per level
    // is enough (nobody can still be reading the buffer that is being overwritten)
    double* buf = reinterpret_cast<double*>(vw_smem);
    double* nxt = buf + M;
    const int tid = threadIdx.x;

    const int64_t id = xcd_remap(blockIdx.x, gridDim.x);
    const int64_t row = id / a.ntiles;
    const int tile = static_cast<int>(id - row * a.ntiles);
    const int64_t t0 = static_cast<int64_t>(tile) * T;
    const int64_t n = a.n;
    const int64_t ns = a.n_store;
    const double* __restrict__ inrow = a.in + row * a.in_stride;

    // ---- stage V_{lev0-1}[t0-HALO, t0+T) into LDS --------------------------------
    {
        double2 v[HPT + PPT];
#pragma unroll
        for (int c = 0; c < HPT + PPT; ++c) {
            const int p = c * kChunk + 2 * tid;
            const int64_t t = t0 - HALO + p;
            double2 x = make_double2(0.0, 0.0);
            if (t >= 0) {
                if (VEC2) {
                    if (t < n) x = *reinterpret_cast<const double2*>(inrow + t);
                } else {
                    if (t < n) x.x = inrow[t];
                    if (t + 1 < n) x.y = inrow[t + 1];
                }
            } else if (MODE == kPeriodic) {
                if (VEC2) {
                    x = *reinterpret_cast<const double2*>(inrow + mod_wrap(t, n));
                } else {
                    x.x = inrow[mod_wrap(t, n)];
                    x.y = inrow[mod_wrap(t + 1, n)];
                }
            }
            v[c] = x;
        }
        if (a.first_bad != nullptr) {
#pragma unroll
            for (int c = HPT; c < HPT + PPT; ++c) {
                const int64_t t = t0 - HALO + c * kChunk + 2 * tid;
                if (t < n && !finite_f64(v[c].x))
                    atomicMin(a.first_bad, static_cast<unsigned long long>(row * n + t));
                if (t + 1 < n && !finite_f64(v[c].y))
                    atomicMin(a.first_bad, static_cast<unsigned long long>(row * n + t + 1));
            }
        }
#pragma unroll
        for (int c = 0; c < HPT + PPT; ++c)
            *reinterpret_cast<double2*>(buf + c * kChunk + 2 * tid) = v[c];
    }
    lds_barrier();

    for (int j = a.lev0; j <= a.lev1; ++j) {
        const int s = 1 << (j - 1);
        const int ps = a.pstart[j];
        double av[PPT][2], aw[PPT][2], ah[HPT][2];
        bool wact[HPT], hwrite[HPT];
        int hbase[HPT];
        const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#pragma unroll
        for (int c = 0; c < HPT; ++c) {
            ah[c][0] = ah[c][1] = 0.0;
            const int p = c * kChunk + 2 * tid;
            // the last level needs no halo: only the tile is stored
            wact[c] = (j < a.lev1) && (c * kChunk + 128 * wave + 128 > ps);
            hwrite[c] = wact[c] && (p >= ps);
            hbase[c] = (p >= ps ? p : ps) * 8;
        }
        const char* lds = reinterpret_cast<const char*>(buf);
        if (s == 1)
            fwd_level<PPT, HPT, true, FMA>(lds, a, s, tid, av, aw, ah, hbase, wact);
        else
            fwd_level<PPT, HPT, false, FMA>(lds, a, s, tid, av, aw, ah, hbase, wact);

        // ---- W_j straight to HBM -------------------------------------------------
        double* __restrict__ wrow = a.wout + row * a.w_row_stride + static_cast<int64_t>(j - 1) * a.w_band_stride;
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            const int64_t t = t0 + k * kChunk + 2 * tid;
            if (VEC2) {
                if (t < ns) gstore2(wrow + t, make_double2(aw[k][0], aw[k][1]), a.nt_store);
            } else {
                if (t < ns) wrow[t] = aw[k][0];
                if (t + 1 < ns) wrow[t + 1] = aw[k][1];
            }
        }
        if (j == a.lev1) {
            double* __restrict__ vrow = a.vout + row * a.v_stride;
#pragma unroll
            for (int k = 0; k < PPT; ++k) {
                const int64_t t = t0 + k * kChunk + 2 * tid;
                if (VEC2) {
                    if (t < ns) gstore2(vrow + t, make_double2(av[k][0], av[k][1]), a.nt_store);
                } else {
                    if (t < ns) vrow[t] = av[k][0];
                    if (t + 1 < ns) vrow[t + 1] = av[k][1];
                }
            }
        } else {
#pragma unroll
            for (int k = 0; k < PPT; ++k)
                *reinterpret_cast<double2*>(nxt + HALO + k * kChunk + 2 * tid) = make_double2(av[k][0], av[k][1]);
#pragma unroll
            for (int c = 0; c < HPT; ++c) {
                if (hwrite[c]) {
                    double2 v = make_double2(ah[c][0], ah[c][1]);
                    if (MODE == kZeroPad) {
                        // samples left of the signal are zero at every level
                        const int64_t t = t0 - HALO + c * kChunk + 2 * tid;
                        if (t < 0) v.x = 0.0;
                        if (t + 1 < 0) v.y = 0.0;
                    }
                    *reinterpret_cast<double2*>(nxt + c * kChunk + 2 * tid) = v;
                }
            }
            lds_barrier();
            double* tmp = buf;
            buf = nxt;
            nxt = tmp;
        }
    }
}

// ---------------------------------------------------------------------------
// streaming forward: one workgroup walks a run of consecutive tiles of one row
// ---------------------------------------------------------------------------
//
// Same tile engine as k_fwd_fused with a 512-sample halo, but only the first tile of the run pays
// for the halo (loaded from HBM and recomputed level by level).  For every later tile the left
// context of level j is the tail of V_{j-1} of the previous tile, kept in a small LDS history array,
// so no halo sample is loaded or recomputed again; the next tile's input is fetched from HBM into
// registers while the current tile is computed.  The lane that owns a tail pair both consumes the
// old history entry and replaces it, so the history needs no second copy and no extra barrier.
//
// NT lanes per workgroup (256 or 512); a tile chunk is 2*NT samples, the halo is owned by the first
// 256 lanes.  Buffer position p <-> sample t0 - 512 + p.
template <int NT, int PPT, int MODE, bool VEC2, bool FMA, int LT = 0>
__global__ void __launch_bounds__(NT) k_fwd_stream(const FwdArgs a) {
    extern __shared__ double2 vw_smem[];
    constexpr int HPT = 1;
    constexpr int HALO = kChunk;
    constexpr int CH = 2 * NT;
    constexpr int T = PPT * CH;
    constexpr int M = HALO + T;
    double* buf = reinterpret_cast<double*>(vw_smem);
    double* hst = buf + M;  // kChunk doubles
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool hwave = wave < kThreads / 64;  // waves whose lanes own a halo pair

    const int64_t id = xcd_remap(blockIdx.x, gridDim.x);
    const int64_t row = id / a.ntiles;
    const int seg = static_cast<int>(id - row * a.ntiles);
    const int tile0 = seg * a.run_base + (seg < a.run_rem ? seg : a.run_rem);
    const int ntl = a.run_base + (seg < a.run_rem ? 1 : 0);
    const int64_t n = a.n;
    const int64_t ns = a.n_store;
    const double* __restrict__ inrow = a.in + row * a.in_stride;
    const bool carry = ntl > 1;

    double2 pre[PPT];
    for (int ch = 0; ch < ntl; ++ch) {
        const int64_t t0 = static_cast<int64_t>(tile0 + ch) * T;
        const bool warm = (ch == 0);
        // ---- stage V_{lev0-1} -------------------------------------------------------
        if (warm) {
            if (hwave) {
                const int64_t t = t0 - HALO + 2 * tid;
                double2 x = make_double2(0.0, 0.0);
                if (t >= 0) {
                    if (VEC2) {
                        if (t < n) x = *reinterpret_cast<const double2*>(inrow + t);
                    } else {
                        if (t < n) x.x = inrow[t];
                        if (t + 1 < n) x.y = inrow[t + 1];
                    }
                } else if (MODE == kPeriodic) {
                    if (VEC2) {
                        x = *reinterpret_cast<const double2*>(inrow + mod_wrap(t, n));
                    } else {
                        x.x = inrow[mod_wrap(t, n)];
                        x.y = inrow[mod_wrap(t + 1, n)];
                    }
                } else if (MODE == kSymmetric) {
                    // half-sample reflection: sample t < 0 is sample -t - 1 (rows of at least HALO samples)
                    if (VEC2) {
                        const double2 m = *reinterpret_cast<const double2*>(inrow + (-t - 2));
                        x = make_double2(m.y, m.x);
                    } else {
                        x.x = inrow[-t - 1];
                        x.y = inrow[-t - 2];
                    }
                }
                *reinterpret_cast<double2*>(buf + 2 * tid) = x;
            }
#pragma unroll
            for (int k = 0; k < PPT; ++k) {
                const int64_t t = t0 + k * CH + 2 * tid;
                double2 x = make_double2(0.0, 0.0);
                if (VEC2) {
                    if (t < n) x = *reinterpret_cast<const double2*>(inrow + t);
                } else {
                    if (t < n) x.x = inrow[t];
                    if (t + 1 < n) x.y = inrow[t + 1];
                }
                pre[k] = x;
            }
        }
        if (a.first_bad != nullptr) {
#pragma unroll
            for (int k = 0; k < PPT; ++k) {
                const int64_t t = t0 + k * CH + 2 * tid;
                if (t < n && !finite_f64(pre[k].x))
                    atomicMin(a.first_bad, static_cast<unsigned long long>(row * n + t));
                if (t + 1 < n && !finite_f64(pre[k].y))
                    atomicMin(a.first_bad, static_cast<unsigned long long>(row * n + t + 1));
            }
        }
#pragma unroll
        for (int k = 0; k < PPT; ++k) *reinterpret_cast<double2*>(buf + HALO + k * CH + 2 * tid) = pre[k];
        if (carry) {
            const int hl = a.hlen[a.lev0];
            const int i = 2 * tid - (CH - hl);
            if (i >= 0) {
                double2* h = reinterpret_cast<double2*>(hst + a.hoff[a.lev0] + i);
                const double2 old = *h;
                *h = pre[PPT - 1];
                if (!warm) *reinterpret_cast<double2*>(buf + HALO - hl + i) = old;
            }
        }
        lds_barrier();

        // ---- next tile's input: in flight while this tile is computed ---------------
        if (ch + 1 < ntl) {
#pragma unroll
            for (int k = 0; k < PPT; ++k) {
                const int64_t t = t0 + T + k * CH + 2 * tid;
                double2 x = make_double2(0.0, 0.0);
                if (VEC2) {
                    if (t < n) x = *reinterpret_cast<const double2*>(inrow + t);
                } else {
                    if (t < n) x.x = inrow[t];
                    if (t + 1 < n) x.y = inrow[t + 1];
                }
                pre[k] = x;
            }
        }

        // SYMMETRIC: left of the row start V_j is the mirror image of its own first samples at EVERY level (the reference
        // reflects the index into the level's input, internal/ScalarOps.java:827-845), which is not what recomputing
        // from a mirrored V_{j-1} gives: the first tile of a row copies its halo from the tile after each level instead
        const bool sym0 = (MODE == kSymmetric) && (t0 == 0);
        for (int j = a.lev0; j <= a.lev1; ++j) {
            const int s = 1 << (j - 1);
            const int ps = a.pstart[j];
            double av[PPT][2], aw[PPT][2], ah[HPT][2];
            bool wact[HPT], hwrite[HPT];
            int hbase[HPT];
            {
                ah[0][0] = ah[0][1] = 0.0;
                const int p = 2 * tid;
                // halo samples are recomputed for the first tile only, and never for the last level
                wact[0] = warm && hwave && (j < a.lev1) && (128 * wave + 128 > ps) && !sym0;
                hwrite[0] = wact[0] && (p >= ps);
                hbase[0] = (p >= ps ? p : ps) * 8;
            }
            const char* lds = reinterpret_cast<const char*>(buf);
            if (s == 1)
                fwd_level<PPT, HPT, true, FMA, CH, LT>(lds, a, s, tid, av, aw, ah, hbase, wact);
            else
                fwd_level<PPT, HPT, false, FMA, CH, LT>(lds, a, s, tid, av, aw, ah, hbase, wact);

            double* __restrict__ wrow =
                a.wout + row * a.w_row_stride + static_cast<int64_t>(j - 1) * a.w_band_stride;
#pragma unroll
            for (int k = 0; k < PPT; ++k) {
                const int64_t t = t0 + k * CH + 2 * tid;
                if (VEC2) {
                    if (t < ns) gstore2(wrow + t, make_double2(aw[k][0], aw[k][1]), a.nt_store);
                } else {
                    if (t < ns) wrow[t] = aw[k][0];
                    if (t + 1 < ns) wrow[t + 1] = aw[k][1];
                }
            }
            if (j == a.lev1) {
                double* __restrict__ vrow = a.vout + row * a.v_stride;
#pragma unroll
                for (int k = 0; k < PPT; ++k) {
                    const int64_t t = t0 + k * CH + 2 * tid;
                    if (VEC2) {
                        if (t < ns) gstore2(vrow + t, make_double2(av[k][0], av[k][1]), a.nt_store);
                    } else {
                        if (t < ns) vrow[t] = av[k][0];
                        if (t + 1 < ns) vrow[t + 1] = av[k][1];
                    }
                }
                if (ch + 1 < ntl) lds_barrier();  // last reads of this tile done before it is restaged
            } else {
                lds_barrier();  // every lane has finished reading V_{j-1}
#pragma unroll
                for (int k = 0; k < PPT; ++k)
                    *reinterpret_cast<double2*>(buf + HALO + k * CH + 2 * tid) = make_double2(av[k][0], av[k][1]);
                if (hwrite[0]) {
                    double2 v = make_double2(ah[0][0], ah[0][1]);
                    if (MODE == kZeroPad) {
                        // samples left of the signal are zero at every level
                        const int64_t t = t0 - HALO + 2 * tid;
                        if (t < 0) v.x = 0.0;
                        if (t + 1 < 0) v.y = 0.0;
                    }
                    *reinterpret_cast<double2*>(buf + 2 * tid) = v;
                }
                if (carry) {
                    const int hl = a.hlen[j + 1];
                    const int i = 2 * tid - (CH - hl);
                    if (i >= 0) {
                        double2* h = reinterpret_cast<double2*>(hst + a.hoff[j + 1] + i);
                        const double2 old = *h;
                        *h = make_double2(av[PPT - 1][0], av[PPT - 1][1]);
                        if (!warm) *reinterpret_cast<double2*>(buf + HALO - hl + i) = old;
                    }
                }
                lds_barrier();
                if (sym0) {
                    if (hwave) {  // V_j[-1 - i] = V_j[i]: halo pair (2 tid, 2 tid + 1) is tile pair HALO - 2 - 2 tid, swapped
                        const double2 m = *reinterpret_cast<const double2*>(buf + HALO + (HALO - 2 - 2 * tid));
                        *reinterpret_cast<double2*>(buf + 2 * tid) = make_double2(m.y, m.x);
                    }
                    lds_barrier();
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------
// streaming forward for long contexts (long filters, deep cascades)
// ---------------------------------------------------------------------------
//
// Same run-of-tiles scheme as k_fwd_stream, for level groups whose context (L-1)(2^J-1) runs to
// thousands of samples.  Recomputing such a halo per tile costs more than the tile itself, and holding
// it in registers costs the occupancy; here nothing is recomputed per tile and the first tile of a run
// is not special either: the run simply starts `ndry` tiles early with the stores switched off.  The
// cascade is FIR, so after sum_j (L-1) 2^(j-1) samples every history holds exact values, whatever it
// held before.  Histories: hlen[j] <= min(T, HB*512) samples of V_{j-1} per level in LDS.
template <int NT, int PPT, int HB, int MODE, int VEC2, bool FMA, int LT = 0>
__global__ void __launch_bounds__(NT) k_fwd_sdry(const FwdArgs a) {
    extern __shared__ double2 vw_smem[];
    constexpr int HALO = HB * kChunk;
    constexpr int CH = 2 * NT;
    constexpr int T = PPT * CH;
    constexpr int M = HALO + T;
    double* buf = reinterpret_cast<double*>(vw_smem);
    double* hst = buf + M;  // a.hist_cap doubles
    const int tid = threadIdx.x;

    const int64_t id = xcd_remap(blockIdx.x, gridDim.x);
    const int64_t row = id / a.ntiles;
    const int seg = static_cast<int>(id - row * a.ntiles);
    const int tile0 = seg * a.nchunks;
    int ntl = a.tiles_per_row - tile0;
    if (ntl > a.nchunks) ntl = a.nchunks;
    const int64_t n = a.n;
    const int64_t ns = a.n_store;
    const double* __restrict__ inrow = a.in + row * a.in_stride;
    // left of a ZERO_PADDING row everything is zero at every level: zeroed histories are already exact
    const int ndry = (MODE == kZeroPad && tile0 == 0) ? 0 : a.ndry;

    for (int i = 2 * tid; i < a.hist_cap; i += 2 * NT) *reinterpret_cast<double2*>(hst + i) = make_double2(0.0, 0.0);
    lds_barrier();

    // Tile loader for any position.  The loads are to stay in flight while the tile before is computed, so nothing may
    // touch the loaded registers until the tile is staged: every lane loads from an address inside the row -- wrapped
    // (PERIODIC) or clamped -- and `settle` puts in the zeros (left of a ZERO_PADDING row, right of any row) when the tile
    // is staged.  (A zero fill at load time, or a load site inside a branch, meets the loaded value in a register copy
    // right behind the load, and the copy waits for the data.)
    auto load_tile = [&](int64_t t0, double2(&v)[PPT]) {
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            int64_t ia = t0 + k * CH + 2 * tid, ib = ia + 1;
            if (MODE == kPeriodic) {
                if (ia < 0) ia = (ia >= -n) ? ia + n : mod_wrap(ia, n);
                if (ib < 0) ib = (ib >= -n) ? ib + n : mod_wrap(ib, n);
            } else {
                if (ia < 0) ia = 0;
                if (ib < 0) ib = 0;
            }
            if (ia >= n) ia = (VEC2 == 1) ? n - 2 : n - 1;  // right of the row: zeros in every mode (settle)
            if (ib >= n) ib = n - 1;
            if (VEC2 == 1) {  // even n, aligned rows: a pair never straddles a row end
                v[k] = *reinterpret_cast<const double2*>(inrow + ia);
            } else {
                v[k].x = inrow[ia];
                v[k].y = inrow[ib];
            }
        }
        __builtin_amdgcn_sched_barrier(0);  // issued here, not where the scheduler finds room for them
    };
    auto settle = [&](int64_t t0, double2(&v)[PPT]) {
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            const int64_t t = t0 + k * CH + 2 * tid;
            if (t >= n || (MODE != kPeriodic && t < 0)) v[k].x = 0.0;
            if (t + 1 >= n || (MODE != kPeriodic && t + 1 < 0)) v[k].y = 0.0;
        }
    };
    // the lanes that own the last hlen[j] samples of the tile swap them with the stored history, which
    // becomes the left context of this tile
    auto hist_swap = [&](int j, const double2(&vals)[PPT]) {
        const int hl = a.hlen[j];
        double* h0 = hst + a.hoff[j];
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            const int i = k * CH + 2 * tid - (T - hl);
            if (i >= 0) {
                double2* h = reinterpret_cast<double2*>(h0 + i);
                const double2 old = *h;
                *h = vals[k];
                *reinterpret_cast<double2*>(buf + HALO - hl + i) = old;
            }
        }
    };

    double2 pre[PPT];
    for (int ch = -ndry; ch < ntl; ++ch) {
        const int64_t t0 = static_cast<int64_t>(tile0 + ch) * T;
        const bool dry = ch < 0;
        if (ch == -ndry) load_tile(t0, pre);
        settle(t0, pre);
        if (a.first_bad != nullptr && !dry) {
#pragma unroll
            for (int k = 0; k < PPT; ++k) {
                const int64_t t = t0 + k * CH + 2 * tid;
                if (t < n && !finite_f64(pre[k].x))
                    atomicMin(a.first_bad, static_cast<unsigned long long>(row * n + t));
                if (t + 1 < n && !finite_f64(pre[k].y))
                    atomicMin(a.first_bad, static_cast<unsigned long long>(row * n + t + 1));
            }
        }
#pragma unroll
        for (int k = 0; k < PPT; ++k) *reinterpret_cast<double2*>(buf + HALO + k * CH + 2 * tid) = pre[k];
        hist_swap(a.lev0, pre);
        lds_barrier();
        load_tile(ch + 1 < ntl ? t0 + T : t0, pre);  // in flight while this tile is computed (after the last tile: itself, unused)

        for (int j = a.lev0; j <= a.lev1; ++j) {
            const int s = 1 << (j - 1);
            double av[PPT][2], aw[PPT][2], ah[1][2] = {{0.0, 0.0}};
            const bool wact[1] = {false};
            const int hbase[1] = {0};
            const char* lds = reinterpret_cast<const char*>(buf);
            if (s == 1)
                fwd_level<PPT, 1, true, FMA, CH, LT, HALO>(lds, a, s, tid, av, aw, ah, hbase, wact);
            else
                fwd_level<PPT, 1, false, FMA, CH, LT, HALO>(lds, a, s, tid, av, aw, ah, hbase, wact);

            if (!dry) {
                double* __restrict__ wrow =
                    a.wout + row * a.w_row_stride + static_cast<int64_t>(j - 1) * a.w_band_stride;
#pragma unroll
                for (int k = 0; k < PPT; ++k) {
                    const int64_t t = t0 + k * CH + 2 * tid;
                    gstore_pair<VEC2>(wrow, t, ns, aw[k][0], aw[k][1], a.nt_store);
                }
            }
            if (j == a.lev1) {
                if (!dry) {
                    double* __restrict__ vrow = a.vout + row * a.v_stride;
#pragma unroll
                    for (int k = 0; k < PPT; ++k) {
                        const int64_t t = t0 + k * CH + 2 * tid;
                        gstore_pair<VEC2>(vrow, t, ns, av[k][0], av[k][1], a.nt_store);
                    }
                }
                if (ch + 1 < ntl) lds_barrier();  // last reads of this tile done before it is restaged
            } else {
                lds_barrier();  // every lane has finished reading V_{j-1}
                double2 v2[PPT];
#pragma unroll
                for (int k = 0; k < PPT; ++k) {
                    v2[k] = make_double2(av[k][0], av[k][1]);
                    *reinterpret_cast<double2*>(buf + HALO + k * CH + 2 * tid) = v2[k];
                }
                hist_swap(j + 1, v2);
                lds_barrier();
            }
        }
    }
}

// ---------------------------------------------------------------------------
// fused inverse: levels lev1 down to lev0 of one (row, tile) per workgroup
// ---------------------------------------------------------------------------
//
// Two LDS buffers of M = (PPT + HPT) * 512 doubles: R (running reconstruction)
// and W (detail band of the level being undone); position p <-> t = t0 + p, the
// halo is on the right and shrinks as the levels descend.  The next band is
// fetched from HBM into registers while the current level is computed.

// Tile chunks (k < PPT) are always computed; halo chunks are switched on per
// wave, lanes right of the needed zone recompute the last needed pair (clamped
// address, result not written back) -- see fwd_tap.
template <int PPT, int NP, bool S1>
__device__ __forceinline__ void inv_read(const char* ldsR, const char* ldsW, const int (&pb)[NP], int off,
                                         const bool (&wact)[NP], const double2 (&rc)[NP], const double2 (&wc)[NP],
                                         double2 (&r)[NP], double2 (&w)[NP]) {
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        if ((k < PPT) || wact[k]) {
            if (S1) {
                // pair (p, p+1) reads positions p+l and p+1+l: slide by one per tap
                r[k].x = rc[k].y;
                w[k].x = wc[k].y;
                r[k].y = *reinterpret_cast<const double*>(ldsR + pb[k] + off + 8);
                w[k].y = *reinterpret_cast<const double*>(ldsW + pb[k] + off + 8);
            } else {
                r[k] = *reinterpret_cast<const double2*>(ldsR + pb[k] + off);
                w[k] = *reinterpret_cast<const double2*>(ldsW + pb[k] + off);
            }
        }
    }
}

template <int PPT, int NP, bool FMA>
__device__ __forceinline__ void inv_mac(double flo, double fhi, const double2 (&rc)[NP], const double2 (&wc)[NP],
                                        const bool (&wact)[NP], double (&acc)[NP][2]) {
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        if ((k < PPT) || wact[k]) {
            // sum += lo[l]*a[idx] + hi[l]*d[idx]
            if (FMA) {
                acc[k][0] = __builtin_fma(flo, rc[k].x, __builtin_fma(fhi, wc[k].x, acc[k][0]));
                acc[k][1] = __builtin_fma(flo, rc[k].y, __builtin_fma(fhi, wc[k].y, acc[k][1]));
            } else {
                acc[k][0] = acc[k][0] + (flo * rc[k].x + fhi * wc[k].x);
                acc[k][1] = acc[k][1] + (flo * rc[k].y + fhi * wc[k].y);
            }
        }
    }
}

// One level of one tile: taps in ascending order, the next tap's filter values fetched one ahead.
// LT > 0: the loop bound is exactly LT; the tap loop is unrolled completely with all 2*LT tap values fetched up
// front, so that no scalar load (which shares its completion counter with the LDS reads) sits inside it.
template <int PPT, int NP, bool S1, bool FMA, int LT = 0>
__device__ __forceinline__ void inv_level(const char* ldsR, const char* ldsW, const InvArgs& a, int s,
                                          double (&acc)[NP][2], const int (&pb)[NP], const bool (&wact)[NP]) {
    const int lh = a.lh;
    const int sb = s * 8;
    double2 r[NP], w[NP];
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        r[k] = make_double2(0.0, 0.0);
        w[k] = make_double2(0.0, 0.0);
        if (S1 && ((k < PPT) || wact[k])) {  // seed of the sliding window: position p
            r[k].y = *reinterpret_cast<const double*>(ldsR + pb[k]);
            w[k].y = *reinterpret_cast<const double*>(ldsW + pb[k]);
        }
    }
    if (LT > 0) {
        double tl[LT > 0 ? LT : 1], th[LT > 0 ? LT : 1];
#pragma unroll
        for (int i = 0; i < LT; ++i) {
            tl[i] = a.taps.lo[i];
            th[i] = a.taps.hi[i];
        }
#pragma unroll
        for (int l = 0; l < LT; ++l) {
            inv_read<PPT, NP, S1>(ldsR, ldsW, pb, l * sb, wact, r, w, r, w);
            inv_mac<PPT, NP, FMA>(tl[l], th[l], r, w, wact, acc);
        }
        return;
    }
    // any length: one tap per trip, the next tap's filter values fetched one ahead
    int off = 0;
    double flo = a.taps.lo[0], fhi = a.taps.hi[0];
    for (int l = 0; l < lh; ++l) {
        const double flo_n = a.taps.lo[l + 1], fhi_n = a.taps.hi[l + 1];
        inv_read<PPT, NP, S1>(ldsR, ldsW, pb, off, wact, r, w, r, w);
        inv_mac<PPT, NP, FMA>(flo, fhi, r, w, wact, acc);
        off += sb;
        flo = flo_n;
        fhi = fhi_n;
    }
}

template <int NP, int MODE, bool VEC2>
__device__ __forceinline__ void inv_fetch(const double* __restrict__ src, bool use, int64_t t0,
                                          int64_t n, int tid, int plimit, double2 (&v)[NP]) {
#pragma unroll
    for (int c = 0; c < NP; ++c) {
        const int p = c * kChunk + 2 * tid;
        double2 x = make_double2(0.0, 0.0);
        if (use && p < plimit) {
            const int64_t t = t0 + p;
            // kSymmetric (single-level launches only: MODWTTransform.inverseBatchOptimized reflects t + l at the row end,
            // modwt/MODWTTransform.java:644-657): sample t >= n is sample 2n - 1 - t; rows of at least one chunk
            if (VEC2) {
                if (t < n) {
                    x = *reinterpret_cast<const double2*>(src + t);
                } else if (MODE == kPeriodic) {
                    x = *reinterpret_cast<const double2*>(src + mod_wrap(t, n));
                } else if (MODE == kSymmetric) {
                    // (a tile that starts near the row end reaches beyond the mirror image of the whole row: nothing
                    // that is stored depends on those positions, and nothing may be read for them)
                    if (t <= 2 * n - 2) {
                        const double2 m = *reinterpret_cast<const double2*>(src + (2 * n - 2 - t));
                        x = make_double2(m.y, m.x);
                    }
                }
            } else {
                if (t < n)
                    x.x = src[t];
                else if (MODE == kPeriodic)
                    x.x = src[mod_wrap(t, n)];
                else if (MODE == kSymmetric && t <= 2 * n - 1)
                    x.x = src[2 * n - 1 - t];
                if (t + 1 < n)
                    x.y = src[t + 1];
                else if (MODE == kPeriodic)
                    x.y = src[mod_wrap(t + 1, n)];
                else if (MODE == kSymmetric && t <= 2 * n - 2)
                    x.y = src[2 * n - 2 - t];
            }
        }
        v[c] = x;
    }
}

template <int PPT, int HPT, int MODE, bool VEC2, bool FMA>
__global__ void __launch_bounds__(kThreads) k_inv_fused(const InvArgs a) {
    extern __shared__ double2 vw_smem[];
    constexpr int NP = PPT + HPT;
    constexpr int M = NP * kChunk;
    double* bufR = reinterpret_cast<double*>(vw_smem);
    double* bufW = bufR + M;
    const char* ldsR = reinterpret_cast<const char*>(bufR);
    const char* ldsW = reinterpret_cast<const char*>(bufW);
    constexpr int T = PPT * kChunk;
    const int tid = threadIdx.x;

    const int64_t id = xcd_remap(blockIdx.x, gridDim.x);
    const int64_t row = id / a.ntiles;
    const int tile = static_cast<int>(id - row * a.ntiles);
    const int64_t t0 = static_cast<int64_t>(tile) * T;
    const int64_t n = a.n;
    const double* __restrict__ wbase = a.win + row * a.w_row_stride;

    double2 pre[NP];
    // R_{lev1}
    inv_fetch<NP, MODE, VEC2>(a.rin + row * a.rin_stride, (a.band_mask >> 31) & 1u, t0, n, tid,
                              a.pend[a.lev1], pre);
    if (a.first_bad != nullptr && a.check_rin) {
        const int64_t flat0 = row * a.w_row_stride + static_cast<int64_t>(a.lev1) * a.w_band_stride;
#pragma unroll
        for (int c = 0; c < PPT; ++c) {
            const int64_t t = t0 + c * kChunk + 2 * tid;
            if (t < n && !finite_f64(pre[c].x)) atomicMin(a.first_bad, static_cast<unsigned long long>(flat0 + t));
            if (t + 1 < n && !finite_f64(pre[c].y)) atomicMin(a.first_bad, static_cast<unsigned long long>(flat0 + t + 1));
        }
    }
#pragma unroll
    for (int c = 0; c < NP; ++c) *reinterpret_cast<double2*>(bufR + c * kChunk + 2 * tid) = pre[c];
    // W_{lev1}
    inv_fetch<NP, MODE, VEC2>(wbase + static_cast<int64_t>(a.lev1 - 1) * a.w_band_stride,
                              (a.band_mask >> (a.lev1 - 1)) & 1u, t0, n, tid, a.pend[a.lev1], pre);

    for (int j = a.lev1; j >= a.lev0; --j) {
        const int s = 1 << (j - 1);
        const double thr = (a.thr_mode != 0) ? uniform_load_f64(a.thr + row * a.thr_row_stride + (j - 1) * a.thr_level_stride) : 0.0;
        // publish W_j (thresholded on the way in when denoising)
#pragma unroll
        for (int c = 0; c < NP; ++c) {
            double2 w = pre[c];
            if (a.thr_mode != 0) {
                w.x = apply_thr(w.x, thr, a.thr_mode);
                w.y = apply_thr(w.y, thr, a.thr_mode);
            }
            if (a.first_bad != nullptr && c < PPT) {
                const int64_t flat0 = row * a.w_row_stride + static_cast<int64_t>(j - 1) * a.w_band_stride;
                const int64_t t = t0 + c * kChunk + 2 * tid;
                if (t < n && !finite_f64(w.x)) atomicMin(a.first_bad, static_cast<unsigned long long>(flat0 + t));
                if (t + 1 < n && !finite_f64(w.y)) atomicMin(a.first_bad, static_cast<unsigned long long>(flat0 + t + 1));
            }
            *reinterpret_cast<double2*>(bufW + c * kChunk + 2 * tid) = w;
        }
        lds_barrier();  // R_j and W_j visible
        if (j > a.lev0)
            inv_fetch<NP, MODE, VEC2>(wbase + static_cast<int64_t>(j - 2) * a.w_band_stride,
                                      (a.band_mask >> (j - 2)) & 1u, t0, n, tid, a.pend[j - 1], pre);

        const int pe = (j > a.lev0) ? a.pend[j - 1] : T;  // the last level only needs the tile
        double acc[NP][2];
        bool wact[NP], wr[NP];
        int pb[NP];
        const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            acc[k][0] = acc[k][1] = 0.0;
            const int p = k * kChunk + 2 * tid;
            wact[k] = (k * kChunk + 128 * wave) < pe;
            wr[k] = p < pe;
            pb[k] = ((k < PPT || p < pe) ? p : pe - 2) * 8;
        }
        if (s == 1)
            inv_level<PPT, NP, true, FMA>(ldsR, ldsW, a, s, acc, pb, wact);
        else
            inv_level<PPT, NP, false, FMA>(ldsR, ldsW, a, s, acc, pb, wact);

        if (j == a.lev0) {
            double* __restrict__ orow = a.rout + row * a.rout_stride;
#pragma unroll
            for (int k = 0; k < PPT; ++k) {
                const int64_t t = t0 + k * kChunk + 2 * tid;
                if (VEC2) {
                    if (t < a.n_store) gstore2(orow + t, make_double2(acc[k][0], acc[k][1]), a.nt_store);
                } else {
                    if (t < a.n_store) orow[t] = acc[k][0];
                    if (t + 1 < a.n_store) orow[t + 1] = acc[k][1];
                }
            }
        } else {
            lds_barrier();  // every lane has finished reading R_j / W_j
#pragma unroll
            for (int k = 0; k < NP; ++k) {
                if (wr[k]) {
                    double2 v = make_double2(acc[k][0], acc[k][1]);
                    if (MODE == kZeroPad) {
                        // samples right of the signal are zero at every level
                        const int64_t t = t0 + k * kChunk + 2 * tid;
                        if (t >= n) v.x = 0.0;
                        if (t + 1 >= n) v.y = 0.0;
                    }
                    *reinterpret_cast<double2*>(bufR + k * kChunk + 2 * tid) = v;
                }
            }
            // the barrier at the top of the next iteration publishes R_{j-1}
        }
    }
}

// ---------------------------------------------------------------------------
// register-blocked inverse: the tile kernel with sliding operand windows
// ---------------------------------------------------------------------------
//
// k_inv_fused reads both operands of every tap of every output pair from LDS: 2*L 16-byte reads per
// pair and level.  Outputs that lie a multiple of the level's stride apart share operands -- output
// t + k*s at tap l reads position t + (k + l)*s -- so here a lane owns, at a level with stride s = 2*sp
// samples, the four pairs P0 + k*sp (k = 0..3, pair index = position / 2) and walks the window
// c = k + l = 0..L+2 once: L + 3 reads per operand for four pairs instead of 4*L, each operand used by
// up to four accumulators straight from registers.  Every accumulator still receives its taps in
// ascending order, so the bits do not change.  The ownership differs from level to level (it follows
// the stride); that costs nothing because R_j travels through LDS between levels anyway.
//
// LDS layout.  Ownership in blocks of 4*sp pairs makes the 16 lanes that one LDS pass serves hit
// addresses that are multiples of 64*sp bytes apart, i.e. the same banks.  The buffers that level j
// reads are therefore stored swizzled for that level: pair index p sits in 16-byte slot
// p ^ (((p >> 4) & 3) << S) for sp = 2^S <= 4, p ^ (((p >> 5) & 1) << 3) for sp = 8, p itself for
// sp >= 16 and for level 1 -- a permutation inside each aligned block of 16 slots that spreads the 16
// lanes of a pass over all 16 slot residues.  Writers (the staging of W_j, the write-back of R_{j-1})
// apply the swizzle of the level that will read.
//
// Level 1 (stride one sample) keeps the chunk ownership (pair tid + 256*k) and reads its window as
// L/2 + 1 aligned 16-byte pairs per operand.  The halo pair of a lane (pair 1024 + tid, needed only
// while lower levels remain) is computed tap by tap as in k_inv_fused, switched on per wave.
// Shape: 256 lanes, 2048-sample tile + 512-sample halo, both filters of exactly LT taps, levels <= 8.

template <int S>
__device__ __forceinline__ int blk_slot(int pair) {  // S = log2(sp) of the reading level; S < 0: level 1
    if (S < 0 || S >= 4) return pair;
    if (S == 3) return pair ^ (((pair >> 5) & 1) << 3);
    return pair ^ (((pair >> 4) & 3) << S);
}
__device__ __forceinline__ int blk_slot_rt(int pair, int S) {
    if (S < 0 || S >= 4) return pair;
    const int sh = (S == 3) ? 5 : 4, mask = (S == 3) ? 1 : 3;
    return pair ^ (((pair >> sh) & mask) << S);
}

constexpr int kBlkPairs = 5 * kThreads;        // pairs per buffer: 4 tile chunks + 1 halo chunk
constexpr int kBlkWOff = kBlkPairs * 16;       // byte offset of the W buffer behind the R buffer

// Thresholding of a detail coefficient as the inverse consumes it (CT = 0: none), for a finite threshold >= 0:
//   CT 1  apply_thr mode 1 (soft, a NaN coefficient becomes zero): max(|c| - thr, 0) with the sign of c -- a zero may
//         come out with the other sign, which cannot reach an output (see shrink_finite);
//   CT 2  apply_thr mode 2 (hard): |c| <= thr ? 0 : c, a NaN coefficient stays;
//   CT 3  apply_thr mode 3 (soft, a NaN coefficient stays): c - clamp(c, -thr, thr), bit for bit the reference's value
//         for every finite, infinite or NaN c (|c| <= thr gives c - c = +0.0).
// Identity settings for a level whose threshold is not such a number (the level is then thresholded in LDS by the
// general rule first): thr = 0 for CT 1 and CT 3, thr = -1 for CT 2.
template <int CT>
__device__ __forceinline__ double consume_thr(double c, double thr) {
    if (CT == 1) return shrink_finite(c, thr, true);
    if (CT == 2) return (fabs(c) <= thr) ? 0.0 : c;
    if (CT == 3) return c - __builtin_fmin(__builtin_fmax(c, -thr), thr);
    return c;
}

template <bool FMA>
__device__ __forceinline__ double inv_pair_mac(double acc, double flo, double r, double fhi, double w) {
    // sum += lo[l]*a[idx] + hi[l]*d[idx]
    return FMA ? __builtin_fma(flo, r, __builtin_fma(fhi, w, acc)) : acc + (flo * r + fhi * w);
}

// main round of a level with stride 2^S pairs (S compile-time: swizzled addresses fold to shifts and xors;
// S >= 4: SRT carries the run-time log2 and no swizzle applies)
// PRELOAD: the whole operand window is requested before the first multiply (a scheduling fence keeps the
// compiler from sinking the reads next to their uses), so that one wave covers its own LDS latency; costs
// 8 * (LT + 3) registers and is meant for the two-waves-per-SIMD ring kernel.
template <int LT, int S, bool FMA, int WOFF = kBlkWOff, bool PRELOAD = false, int CT = 0>
__device__ __forceinline__ void inv_blk_main(const char* lds, int tid, int SRT, const double (&tl)[LT], const double (&th)[LT],
                                             double (&acc)[5][2], int& P0out, double cthr = 0.0) {
    const int sl = (S >= 4) ? SRT : S;
    const int sp = 1 << sl;
    const int P0 = ((tid >> sl) << (sl + 2)) + (tid & (sp - 1));
    P0out = P0;
    if (PRELOAD) {
        double2 r[LT + 3], w[LT + 3];
#pragma unroll
        for (int c = 0; c < LT + 3; ++c) {
            const int adr = blk_slot<S>(P0 + c * sp) << 4;
            r[c] = *reinterpret_cast<const double2*>(lds + adr);
            w[c] = *reinterpret_cast<const double2*>(lds + WOFF + adr);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 0; c < LT + 3; ++c) {
            if (CT != 0) {
                w[c].x = consume_thr<CT>(w[c].x, cthr);
                w[c].y = consume_thr<CT>(w[c].y, cthr);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int l = c - k;
                if (l >= 0 && l < LT) {
                    acc[k][0] = inv_pair_mac<FMA>(l == 0 ? 0.0 : acc[k][0], tl[l], r[c].x, th[l], w[c].x);
                    acc[k][1] = inv_pair_mac<FMA>(l == 0 ? 0.0 : acc[k][1], tl[l], r[c].y, th[l], w[c].y);
                }
            }
        }
        return;
    }
    static_assert(CT == 0 || PRELOAD, "consume-time thresholding is wired into the preloading forms only");
#pragma unroll
    for (int c = 0; c < LT + 3; ++c) {
        const int adr = blk_slot<S>(P0 + c * sp) << 4;
        const double2 r = *reinterpret_cast<const double2*>(lds + adr);
        const double2 w = *reinterpret_cast<const double2*>(lds + WOFF + adr);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int l = c - k;
            if (l >= 0 && l < LT) {
                acc[k][0] = inv_pair_mac<FMA>(l == 0 ? 0.0 : acc[k][0], tl[l], r.x, th[l], w.x);
                acc[k][1] = inv_pair_mac<FMA>(l == 0 ? 0.0 : acc[k][1], tl[l], r.y, th[l], w.y);
            }
        }
    }
}

// halo pair of a lane at a blocked level: one pair, tap by tap (wave-uniform switch outside)
template <int LT, int S, bool FMA, int WOFF = kBlkWOff, bool PRELOAD = false, int CT = 0>
__device__ __forceinline__ void inv_blk_halo(const char* lds, int ph, int SRT, const double (&tl)[LT], const double (&th)[LT],
                                             double (&acc)[5][2], double cthr = 0.0) {
    const int sl = (S >= 4) ? SRT : S;
    const int sp = 1 << sl;
    if (PRELOAD) {
        double2 r[LT], w[LT];
#pragma unroll
        for (int l = 0; l < LT; ++l) {
            const int adr = blk_slot<S>(ph + l * sp) << 4;
            r[l] = *reinterpret_cast<const double2*>(lds + adr);
            w[l] = *reinterpret_cast<const double2*>(lds + WOFF + adr);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int l = 0; l < LT; ++l) {
            acc[4][0] = inv_pair_mac<FMA>(l == 0 ? 0.0 : acc[4][0], tl[l], r[l].x, th[l], consume_thr<CT>(w[l].x, cthr));
            acc[4][1] = inv_pair_mac<FMA>(l == 0 ? 0.0 : acc[4][1], tl[l], r[l].y, th[l], consume_thr<CT>(w[l].y, cthr));
        }
        return;
    }
    static_assert(CT == 0 || PRELOAD, "consume-time thresholding is wired into the preloading forms only");
#pragma unroll
    for (int l = 0; l < LT; ++l) {
        const int adr = blk_slot<S>(ph + l * sp) << 4;
        const double2 r = *reinterpret_cast<const double2*>(lds + adr);
        const double2 w = *reinterpret_cast<const double2*>(lds + WOFF + adr);
        acc[4][0] = inv_pair_mac<FMA>(l == 0 ? 0.0 : acc[4][0], tl[l], r.x, th[l], w.x);
        acc[4][1] = inv_pair_mac<FMA>(l == 0 ? 0.0 : acc[4][1], tl[l], r.y, th[l], w.y);
    }
}

// level 1: chunk ownership, window of LT/2 + 1 aligned pairs per operand
template <int LT, bool FMA, int WOFF = kBlkWOff, bool PRELOAD = false, int CT = 0>
__device__ __forceinline__ void inv_blk_level1(const char* lds, int tid, const double (&tl)[LT], const double (&th)[LT],
                                               double (&acc)[5][2], double cthr = 0.0) {
    constexpr int NW = LT / 2 + 1;
    if (PRELOAD) {
        // two chunks' windows at a time
#pragma unroll
        for (int k0 = 0; k0 < 4; k0 += 2) {
            double2 rr[2][NW], ww[2][NW];
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int q = 0; q < NW; ++q) {
                    const int adr = ((k0 + kk) * kThreads + tid + q) << 4;
                    rr[kk][q] = *reinterpret_cast<const double2*>(lds + adr);
                    ww[kk][q] = *reinterpret_cast<const double2*>(lds + WOFF + adr);
                }
            __builtin_amdgcn_sched_barrier(0);
            if (CT != 0) {
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                    for (int q = 0; q < NW; ++q) {
                        ww[kk][q].x = consume_thr<CT>(ww[kk][q].x, cthr);
                        ww[kk][q].y = consume_thr<CT>(ww[kk][q].y, cthr);
                    }
            }
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int l = 0; l < LT; ++l) {
                    const double r0 = (l & 1) ? rr[kk][l / 2].y : rr[kk][l / 2].x;
                    const double r1 = (l & 1) ? rr[kk][l / 2 + 1].x : rr[kk][l / 2].y;
                    const double w0 = (l & 1) ? ww[kk][l / 2].y : ww[kk][l / 2].x;
                    const double w1 = (l & 1) ? ww[kk][l / 2 + 1].x : ww[kk][l / 2].y;
                    acc[k0 + kk][0] = inv_pair_mac<FMA>(l == 0 ? 0.0 : acc[k0 + kk][0], tl[l], r0, th[l], w0);
                    acc[k0 + kk][1] = inv_pair_mac<FMA>(l == 0 ? 0.0 : acc[k0 + kk][1], tl[l], r1, th[l], w1);
                }
            __builtin_amdgcn_sched_barrier(0);
        }
        return;
    }
    static_assert(CT == 0 || PRELOAD, "consume-time thresholding is wired into the preloading forms only");
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        double r[2 * NW], w[2 * NW];
#pragma unroll
        for (int q = 0; q < NW; ++q) {
            const int adr = (k * kThreads + tid + q) << 4;
            const double2 rr = *reinterpret_cast<const double2*>(lds + adr);
            const double2 ww = *reinterpret_cast<const double2*>(lds + WOFF + adr);
            r[2 * q] = rr.x;
            r[2 * q + 1] = rr.y;
            w[2 * q] = ww.x;
            w[2 * q + 1] = ww.y;
        }
#pragma unroll
        for (int l = 0; l < LT; ++l) {
            acc[k][0] = inv_pair_mac<FMA>(l == 0 ? 0.0 : acc[k][0], tl[l], r[l], th[l], w[l]);
            acc[k][1] = inv_pair_mac<FMA>(l == 0 ? 0.0 : acc[k][1], tl[l], r[l + 1], th[l], w[l + 1]);
        }
    }
}

template <int LT, int MODE, bool VEC2, bool FMA>
__global__ void __launch_bounds__(kThreads, 3) k_inv_blk(const InvArgs a) {
    extern __shared__ double2 vw_smem[];
    constexpr int PPT = 4, NP = 5;
    constexpr int T = PPT * kChunk;
    char* lds = reinterpret_cast<char*>(vw_smem);
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    const int64_t id = xcd_remap(blockIdx.x, gridDim.x);
    const int64_t row = id / a.ntiles;
    const int tile = static_cast<int>(id - row * a.ntiles);
    const int64_t t0 = static_cast<int64_t>(tile) * T;
    const int64_t n = a.n;
    const double* __restrict__ wbase = a.win + row * a.w_row_stride;

    double tl[LT], th[LT];
#pragma unroll
    for (int i = 0; i < LT; ++i) {
        tl[i] = a.taps.lo[i];
        th[i] = a.taps.hi[i];
    }

    double2 pre[NP];
    // R_{lev1}, staged in the layout of the level that reads it.  The swizzle only looks at bits 4..5 of the pair
    // index, which the chunk term 256*c does not touch: one swizzled base per lane and level serves all chunks.
    inv_fetch<NP, MODE, VEC2>(a.rin + row * a.rin_stride, (a.band_mask >> 31) & 1u, t0, n, tid, a.pend[a.lev1], pre);
    if (a.first_bad != nullptr && a.check_rin) {
        const int64_t flat0 = row * a.w_row_stride + static_cast<int64_t>(a.lev1) * a.w_band_stride;
#pragma unroll
        for (int c = 0; c < PPT; ++c) {
            const int64_t t = t0 + c * kChunk + 2 * tid;
            if (t < n && !finite_f64(pre[c].x)) atomicMin(a.first_bad, static_cast<unsigned long long>(flat0 + t));
            if (t + 1 < n && !finite_f64(pre[c].y)) atomicMin(a.first_bad, static_cast<unsigned long long>(flat0 + t + 1));
        }
    }
    {
        const int base = blk_slot_rt(tid, a.lev1 - 2) << 4;
#pragma unroll
        for (int c = 0; c < NP; ++c) *reinterpret_cast<double2*>(lds + base + c * (kThreads * 16)) = pre[c];
    }
    inv_fetch<NP, MODE, VEC2>(wbase + static_cast<int64_t>(a.lev1 - 1) * a.w_band_stride, (a.band_mask >> (a.lev1 - 1)) & 1u, t0, n,
                              tid, a.pend[a.lev1], pre);

    for (int j = a.lev1; j >= a.lev0; --j) {
        const int S = j - 2;
        const double thr = (a.thr_mode != 0) ? uniform_load_f64(a.thr + row * a.thr_row_stride + (j - 1) * a.thr_level_stride) : 0.0;
        // publish W_j (thresholded on the way in when denoising) in this level's layout
        {
            const int base = (blk_slot_rt(tid, S) << 4) + kBlkWOff;
#pragma unroll
            for (int c = 0; c < NP; ++c) {
                double2 w = pre[c];
                if (a.thr_mode != 0) {
                    w.x = apply_thr(w.x, thr, a.thr_mode);
                    w.y = apply_thr(w.y, thr, a.thr_mode);
                }
                if (a.first_bad != nullptr && c < PPT) {
                    const int64_t flat0 = row * a.w_row_stride + static_cast<int64_t>(j - 1) * a.w_band_stride;
                    const int64_t t = t0 + c * kChunk + 2 * tid;
                    if (t < n && !finite_f64(w.x)) atomicMin(a.first_bad, static_cast<unsigned long long>(flat0 + t));
                    if (t + 1 < n && !finite_f64(w.y)) atomicMin(a.first_bad, static_cast<unsigned long long>(flat0 + t + 1));
                }
                *reinterpret_cast<double2*>(lds + base + c * (kThreads * 16)) = w;
            }
        }
        lds_barrier();  // R_j and W_j visible
        if (j > a.lev0)
            inv_fetch<NP, MODE, VEC2>(wbase + static_cast<int64_t>(j - 2) * a.w_band_stride, (a.band_mask >> (j - 2)) & 1u, t0, n, tid,
                                      a.pend[j - 1], pre);

        const int pe = (j > a.lev0) ? a.pend[j - 1] : T;  // the last level only needs the tile
        const bool hact = (T + 128 * wave) < pe;          // this wave owns halo pairs that lower levels need
        const int ph_own = kThreads * PPT + tid;          // the lane's halo pair
        const bool hwr = 2 * ph_own < pe;
        const int ph = hwr ? ph_own : (pe >> 1) - 1;      // lanes right of the needed zone recompute its last pair
        double acc[NP][2];
        int P0 = 0;
        acc[4][0] = acc[4][1] = 0.0;
        switch (j) {
            case 1:
                inv_blk_level1<LT, FMA>(lds, tid, tl, th, acc);
                break;
            case 2:
                inv_blk_main<LT, 0, FMA>(lds, tid, 0, tl, th, acc, P0);
                if (hact) inv_blk_halo<LT, 0, FMA>(lds, ph, 0, tl, th, acc);
                break;
            case 3:
                inv_blk_main<LT, 1, FMA>(lds, tid, 1, tl, th, acc, P0);
                if (hact) inv_blk_halo<LT, 1, FMA>(lds, ph, 1, tl, th, acc);
                break;
            case 4:
                inv_blk_main<LT, 2, FMA>(lds, tid, 2, tl, th, acc, P0);
                if (hact) inv_blk_halo<LT, 2, FMA>(lds, ph, 2, tl, th, acc);
                break;
            case 5:
                inv_blk_main<LT, 3, FMA>(lds, tid, 3, tl, th, acc, P0);
                if (hact) inv_blk_halo<LT, 3, FMA>(lds, ph, 3, tl, th, acc);
                break;
            default:
                inv_blk_main<LT, 4, FMA>(lds, tid, S, tl, th, acc, P0);
                if (hact) inv_blk_halo<LT, 4, FMA>(lds, ph, S, tl, th, acc);
                break;
        }
        // pair index of accumulator k in this level's ownership
        const int sp = (j >= 2) ? (1 << S) : 0;

        if (j == a.lev0) {
            double* __restrict__ orow = a.rout + row * a.rout_stride;
#pragma unroll
            for (int k = 0; k < PPT; ++k) {
                const int pair = (j == 1) ? k * kThreads + tid : P0 + k * sp;
                const int64_t t = t0 + 2 * pair;
                if (VEC2) {
                    if (t < a.n_store) gstore2(orow + t, make_double2(acc[k][0], acc[k][1]), a.nt_store);
                } else {
                    if (t < a.n_store) orow[t] = acc[k][0];
                    if (t + 1 < a.n_store) orow[t + 1] = acc[k][1];
                }
            }
        } else {
            lds_barrier();  // every lane has finished reading R_j / W_j
            // j > lev0 >= 1, so this is a blocked level; R_{j-1} goes into the layout of level j-1
#pragma unroll
            for (int k = 0; k < NP; ++k) {
                if (k < PPT || hwr) {
                    const int pair = (k < PPT) ? P0 + k * sp : ph_own;
                    double2 v = make_double2(acc[k][0], acc[k][1]);
                    if (MODE == kZeroPad) {
                        // samples right of the signal are zero at every level
                        const int64_t t = t0 + 2 * pair;
                        if (t >= n) v.x = 0.0;
                        if (t + 1 >= n) v.y = 0.0;
                    }
                    *reinterpret_cast<double2*>(lds + (blk_slot_rt(pair, S - 1) << 4)) = v;
                }
            }
            // the barrier at the top of the next iteration publishes R_{j-1}
        }
    }
}

// ---------------------------------------------------------------------------
// k_inv_ring: the register-blocked inverse fed by LDS-DMA, two detail bands ahead
// ---------------------------------------------------------------------------
//
// The inverse reads seven planes per tile and every level depends on its band having arrived.  With the band
// of the next level landing in registers (k_inv_fused, k_inv_blk) a workgroup has at most one band in flight
// and waits out a full memory latency per level.  Here the bands go from HBM straight into LDS
// (global_load_lds_dwordx4: no registers, no ds_write pass) into a ring of three band buffers, issued two
// levels ahead: W_{j-2} is requested before level j starts, so two bands per workgroup are always in flight.
// LDS: R + 3 band buffers of 20 KiB = 80 KiB, two workgroups per CU.
//
// An LDS-DMA instruction writes 64 consecutive 16-byte slots (wave-uniform base + lane * 16); the SOURCE
// address is per lane.  The swizzled layouts of the blocked levels are therefore produced on the source side:
// the lane that fills slot s fetches the pair whose swizzled position is s (the swizzle is an involution inside
// aligned blocks of 16 slots).  Wave w fills the 1 KiB blocks w, w+4, ... of a band: always five instructions
// per band and wave, so the number of younger requests at each wait is a constant (vmcnt counts in order):
// before level j everything but the two youngest bands must have landed.  Slots right of the zone a level
// needs are filled from a clamped address (valid memory, values never used).  Bands travel by slot j % 3.
//
// PLAIN = false adds a pass over the freshly landed slots of the wave (threshold-on-load for the denoisers,
// band mask, finite check): it touches only W_j (and R on the first level), which the counted wait of the level
// covers, so the two younger bands stay in flight across it; the plain form runs that pass only on the
// ZERO_PADDING tiles that reach beyond their row (zeros right of the signal).
// Needs 16-byte-aligned rows with even n >= 2560 (one wrap at most inside a tile's halo).

__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst)
                 : "memory");
}
// the same with a wave-uniform 64-bit base and a 32-bit byte offset per lane
__device__ __forceinline__ void glds16_so(const void* sbase, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(lds_dst)
                 : "memory");
}

constexpr int kRingBand = kBlkPairs * 16;  // bytes per band buffer

// One band (R or W_j of this tile) into the buffer at byte offset `dst`, in the layout of the level with log2
// stride S.  Wave w fills the 1 KiB blocks w, w+4, ..., w+16: block index and chunk differ by multiples of 256
// slots, which the swizzle (bits 4..5 of the slot index) does not see, so the five source addresses of a lane are
// one swizzled offset plus multiples of 4096 bytes.  Only the halo chunk can lie right of the needed zone (clamped)
// and only the last tile of a row reaches beyond the row (`edge`: wrapped or clamped lane by lane).
template <int MODE>
__device__ __forceinline__ void ring_issue(const double* __restrict__ src, unsigned lds_base, int dst, int S, int pl /*pairs needed*/,
                                           int64_t t0, int64_t n, int wave, int lane, bool edge) {
    const int pair0 = blk_slot_rt(64 * wave + lane, S);
    const unsigned ldst = lds_base + dst + 1024 * wave;
    if (!edge) {
        const double* tsrc = src + t0;  // wave-uniform
        const unsigned off0 = static_cast<unsigned>(pair0) << 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) glds16_so(tsrc, off0 + 4096u * i, ldst + 4096u * i);
        const int p4 = pair0 + 4 * kThreads;
        glds16_so(tsrc, static_cast<unsigned>(p4 < pl ? p4 : pl - 1) << 4, ldst + 4096u * 4);
        return;
    }
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        int pair = pair0 + i * kThreads;
        pair = pair < pl ? pair : pl - 1;
        int64_t t = t0 + 2 * pair;
        if (t >= n) t = (MODE == kPeriodic) ? t - n : n - 2;
        glds16(src + t, ldst + 4096u * i);
    }
}

// write-back of a blocked level's accumulators as R_{j-1}, in the layout of the level (log2 stride SR) that reads it
template <int SR, int MODE>
__device__ __forceinline__ void ring_writeback(char* lds, const double (&acc)[5][2], int P0, int sp, int ph_own, bool hwr, int64_t t0,
                                               int64_t n) {
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        if (k < 4 || hwr) {
            const int pair = (k < 4) ? P0 + k * sp : ph_own;
            double2 v = make_double2(acc[k][0], acc[k][1]);
            if (MODE == kZeroPad) {
                // samples right of the signal are zero at every level
                const int64_t t = t0 + 2 * pair;
                if (t >= n) v.x = 0.0;
                if (t + 1 >= n) v.y = 0.0;
            }
            *reinterpret_cast<double2*>(lds + (blk_slot<SR>(pair) << 4)) = v;
        }
    }
}

template <int LT, int MODE, bool FMA, bool PLAIN, int TH = 0>
__global__ void __launch_bounds__(kThreads, 2) k_inv_ring(const InvArgs a) {
    static_assert(TH == 0 || PLAIN, "consume-time thresholding goes with the plain staging");
    extern __shared__ double2 vw_smem[];
    constexpr int PPT = 4, NP = 5;
    constexpr int T = PPT * kChunk;
    char* lds = reinterpret_cast<char*>(vw_smem);
    const unsigned lds_base = static_cast<unsigned>(reinterpret_cast<uintptr_t>(lds));  // LDS byte address of the R buffer
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;

    const int64_t id = xcd_remap(blockIdx.x, gridDim.x);
    const int64_t row = id / a.ntiles;
    const int tile = static_cast<int>(id - row * a.ntiles);
    const int64_t t0 = static_cast<int64_t>(tile) * T;
    const int64_t n = a.n;
    const double* __restrict__ wbase = a.win + row * a.w_row_stride;

    double tl[LT], th[LT];
#pragma unroll
    for (int i = 0; i < LT; ++i) {
        tl[i] = a.taps.lo[i];
        th[i] = a.taps.hi[i];
    }

    // the tile (with its halo) reaches beyond its row: the loads wrap or clamp lane by lane, and a ZERO_PADDING tile
    // needs zeros there even in the plain form
    const bool redge = t0 + 2 * kBlkPairs > n;
    const bool edge = (MODE == kZeroPad) && redge;

    // prologue: R_{lev1}, W_{lev1}, W_{lev1-1}
    if (!VW_ABLATED(a, 1)) ring_issue<MODE>(a.rin + row * a.rin_stride, lds_base, 0, a.lev1 - 2, a.pend[a.lev1] >> 1, t0, n, wave, lane, redge);
    if (!VW_ABLATED(a, 1)) ring_issue<MODE>(wbase + static_cast<int64_t>(a.lev1 - 1) * a.w_band_stride, lds_base, (1 + a.lev1 % 3) * kRingBand, a.lev1 - 2,
                     a.pend[a.lev1] >> 1, t0, n, wave, lane, redge);
    if (a.lev1 - 1 >= a.lev0 && !VW_ABLATED(a, 1))
        ring_issue<MODE>(wbase + static_cast<int64_t>(a.lev1 - 2) * a.w_band_stride, lds_base, (1 + (a.lev1 - 1) % 3) * kRingBand,
                         a.lev1 - 3, a.pend[a.lev1 - 1] >> 1, t0, n, wave, lane, redge);

    for (int j = a.lev1; j >= a.lev0; --j) {
        const int S = j - 2;
        // two levels ahead; its slot was last read at level j+1, whose readers all passed that level's second barrier
        if (j - 2 >= a.lev0 && !VW_ABLATED(a, 1))
            ring_issue<MODE>(wbase + static_cast<int64_t>(j - 3) * a.w_band_stride, lds_base, (1 + (j - 2) % 3) * kRingBand, j - 4,
                             a.pend[j - 2] >> 1, t0, n, wave, lane, redge);
        // everything older than the two youngest bands has landed: R (first pass) and W_j
        if (j - 2 >= a.lev0) {
            asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
        } else if (j - 1 >= a.lev0) {
            asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        // TH != 0 (a denoise with nothing else to do to the bands): the level's threshold is applied to the W operands
        // as the arithmetic consumes them, and the pass below only runs where it is needed anyway (zeros right of a
        // ZERO_PADDING row) or where the threshold is no finite non-negative number -- then it applies the general
        // rule in LDS and the consumers get the identity setting
        const double thr_j = (TH != 0 || (!PLAIN && a.thr_mode != 0))
                                 ? uniform_load_f64(a.thr + row * a.thr_row_stride + (j - 1) * a.thr_level_stride) : 0.0;
        const bool thr_plain = finite_f64(thr_j) && thr_j >= 0.0;
        const double cthr = (TH == 0) ? 0.0 : (thr_plain ? thr_j : (TH == 2 ? -1.0 : 0.0));
        if (!PLAIN || edge || (TH != 0 && !thr_plain)) {
            // pass over the slots this wave has filled: W_j (and R on the first level).  Kept lean -- it runs once per
            // level on 5 slots per lane: finite values under a finite non-negative threshold take the three-instruction
            // form of the thresholding rule (a wave-uniform choice), and the non-finite report, which needs 64-bit
            // index arithmetic and atomics, is a second loop that only a wave holding such a value enters.
            const int mode = (TH != 0) ? (thr_plain ? 0 : a.thr_mode) : (PLAIN ? 0 : a.thr_mode);
            const double thr = thr_j;
            const bool use_w = (a.band_mask >> (j - 1)) & 1u;
            const bool use_r = (a.band_mask >> 31) & 1u;
            const int wofs = (1 + j % 3) * kRingBand;
            const int pl = a.pend[j] >> 1;
            const bool top = (j == a.lev1);
            // all five slots are requested before the first is used; the decisions that do not depend on the slot are
            // taken once per level, outside the loops
            double2 wv[5];
            double2* pw[5];
            bool live[5];
            bool bad = false;
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                const int slot = 64 * (wave + 4 * i) + lane;
                const int pair = blk_slot_rt(slot, S);
                const bool inrow = (MODE == kPeriodic) || (t0 + 2 * pair < n);
                live[i] = inrow && pair < pl;
                pw[i] = reinterpret_cast<double2*>(lds + wofs + 16 * slot);
                wv[i] = *pw[i];
            }
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                if (!use_w || !live[i]) wv[i] = make_double2(0.0, 0.0);
                bad = bad || !finite_f64(wv[i].x) || !finite_f64(wv[i].y);
            }
            if (mode != 0) {
                if (thr_plain && __builtin_amdgcn_ballot_w64(bad) == 0) {
                    if (mode != 2) {
#pragma unroll
                        for (int i = 0; i < 5; ++i) {
                            wv[i].x = shrink_finite(wv[i].x, thr, true);
                            wv[i].y = shrink_finite(wv[i].y, thr, true);
                        }
                    } else {
#pragma unroll
                        for (int i = 0; i < 5; ++i) {
                            wv[i].x = shrink_finite(wv[i].x, thr, false);
                            wv[i].y = shrink_finite(wv[i].y, thr, false);
                        }
                    }
                } else {
                    bad = false;
#pragma unroll
                    for (int i = 0; i < 5; ++i) {  // (a rolled loop would index wv dynamically and put it into scratch)
                        wv[i].x = apply_thr(wv[i].x, thr, mode);
                        wv[i].y = apply_thr(wv[i].y, thr, mode);
                        bad = bad || !finite_f64(wv[i].x) || !finite_f64(wv[i].y);
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < 5; ++i) *pw[i] = wv[i];
            if (top) {
#pragma unroll
                for (int i = 0; i < 5; ++i) {
                    double2* pr = reinterpret_cast<double2*>(reinterpret_cast<char*>(pw[i]) - wofs);
                    double2 r = *pr;
                    if (!use_r || !live[i]) r = make_double2(0.0, 0.0);
                    bad = bad || !finite_f64(r.x) || !finite_f64(r.y);
                    *pr = r;
                }
            }
            if (a.first_bad != nullptr && __builtin_amdgcn_ballot_w64(bad) != 0) {
                // the wave holds a NaN or an infinity: report the first one of the tile proper by its flat index
#pragma unroll 1
                for (int i = 0; i < 4; ++i) {  // the fifth block of a wave is halo: pairs >= kThreads * PPT
                    const int slot = 64 * (wave + 4 * i) + lane;
                    const int pair = blk_slot_rt(slot, S);
                    const int64_t t = t0 + 2 * pair;
                    if (pair >= kThreads * PPT || t >= n) continue;
                    const double2 w = *reinterpret_cast<const double2*>(lds + wofs + 16 * slot);
                    const int64_t flat0 = row * a.w_row_stride + static_cast<int64_t>(j - 1) * a.w_band_stride;
                    if (!finite_f64(w.x)) atomicMin(a.first_bad, static_cast<unsigned long long>(flat0 + t));
                    if (!finite_f64(w.y)) atomicMin(a.first_bad, static_cast<unsigned long long>(flat0 + t + 1));
                    if (top && a.check_rin) {
                        const double2 r = *reinterpret_cast<const double2*>(lds + 16 * slot);
                        const int64_t flat1 = row * a.w_row_stride + static_cast<int64_t>(a.lev1) * a.w_band_stride;
                        if (!finite_f64(r.x)) atomicMin(a.first_bad, static_cast<unsigned long long>(flat1 + t));
                        if (!finite_f64(r.y)) atomicMin(a.first_bad, static_cast<unsigned long long>(flat1 + t + 1));
                    }
                }
            }
        }
        lds_barrier();  // every wave's part of W_j has landed, R_j is written

        const int pe = (j > a.lev0) ? a.pend[j - 1] : T;  // the last level only needs the tile
        const bool hact = (T + 128 * wave) < pe;
        const int ph_own = kThreads * PPT + tid;
        const bool hwr = 2 * ph_own < pe;
        const int ph = hwr ? ph_own : (pe >> 1) - 1;
        double acc[NP][2];
        int P0 = 0;
        acc[4][0] = acc[4][1] = 0.0;
        if (VW_ABLATED(a, 0)) {
#pragma unroll
            for (int k = 0; k < NP; ++k) acc[k][0] = acc[k][1] = 1.0;
            P0 = ((tid >> (S < 0 ? 0 : S)) << ((S < 0 ? 0 : S) + 2)) + (tid & ((1 << (S < 0 ? 0 : S)) - 1));
        } else
#define VW_RING_LEVEL(SS, SLOT)                                                                         \
    inv_blk_main<LT, SS, FMA, (1 + SLOT) * kRingBand, true, TH>(lds, tid, S, tl, th, acc, P0, cthr);    \
    if (hact) inv_blk_halo<LT, SS, FMA, (1 + SLOT) * kRingBand, true, TH>(lds, ph, S, tl, th, acc, cthr);
        switch (j) {
            case 1: inv_blk_level1<LT, FMA, (1 + 1) * kRingBand, true, TH>(lds, tid, tl, th, acc, cthr); break;
            case 2: VW_RING_LEVEL(0, 2) break;
            case 3: VW_RING_LEVEL(1, 0) break;
            case 4: VW_RING_LEVEL(2, 1) break;
            case 5: VW_RING_LEVEL(3, 2) break;
            case 6: VW_RING_LEVEL(4, 0) break;
            case 7: VW_RING_LEVEL(4, 1) break;
            default: VW_RING_LEVEL(4, 2) break;  // level 8
        }
#undef VW_RING_LEVEL
        const int sp = (j >= 2) ? (1 << S) : 0;

        if (j == a.lev0) {
            double* __restrict__ orow = a.rout + row * a.rout_stride;
#pragma unroll
            for (int k = 0; k < PPT; ++k) {
                const int pair = (j == 1) ? k * kThreads + tid : P0 + k * sp;
                const int64_t t = t0 + 2 * pair;
                if (t < a.n_store) gstore2(orow + t, make_double2(acc[k][0], acc[k][1]), a.nt_store);
            }
        } else {
            lds_barrier();  // every lane has finished reading R_j / W_j
            // R_{j-1} in the layout of level j-1 (j > lev0 >= 1: a blocked level wrote these accumulators)
            switch (j) {
                case 2: ring_writeback<-1, MODE>(lds, acc, P0, sp, ph_own, hwr, t0, n); break;
                case 3: ring_writeback<0, MODE>(lds, acc, P0, sp, ph_own, hwr, t0, n); break;
                case 4: ring_writeback<1, MODE>(lds, acc, P0, sp, ph_own, hwr, t0, n); break;
                case 5: ring_writeback<2, MODE>(lds, acc, P0, sp, ph_own, hwr, t0, n); break;
                case 6: ring_writeback<3, MODE>(lds, acc, P0, sp, ph_own, hwr, t0, n); break;
                default: ring_writeback<4, MODE>(lds, acc, P0, sp, ph_own, hwr, t0, n); break;
            }
        }
    }
}

// ---------------------------------------------------------------------------
// k_inv_carry: k_inv_ring over a run of tiles, with the right context carried from tile to tile
// ---------------------------------------------------------------------------
//
// k_inv_ring pays for the right context of every tile twice: a fifth 512-sample chunk of every band is loaded and
// the running reconstruction is recomputed on it level by level (a quarter more loads, LDS reads and arithmetic).
// Here a workgroup walks a run of tiles of one row from right to left.  Only the first tile of the run is treated
// as in k_inv_ring; for every later tile the right context of level j -- the first (L-1) 2^(j-1) samples of R_j
// and of W_j of the tile just done, its "heads" -- is taken from LDS.  A buffer is used as
//     pairs [0, 1024)     the tile
//     pairs [1024, 1152)  right context (the heads of the previous tile; the first tile's loaded / recomputed halo)
//     pairs [1152, 1280)  head storage
// so the four buffers keep their 20 KiB.  Only the top level of a run's first tile needs more than 128 pairs of
// context; it overruns the head storage of the R buffer and of its own band slot before anything is stored there,
// and its own heads go to the other two buffers.  Heads are exchanged by the lanes that already hold the values:
// the two waves whose LDS-DMA filled pairs 0..127 of a band (after their own wait for it), and the lanes that
// write R_{j-1} back; the owner takes the old head out, stores the new one and puts the old one into the context
// zone, so no extra barrier is needed.  Later tiles issue four LDS-DMA instructions per band and wave, the first
// tile four or five (a wave skips the fifth when its block lies right of the needed zone, which also keeps the head
// storage intact); the counted waits follow.
constexpr int kCarryCtx = 128;                               // pairs of right context behind the tile
constexpr int kCarrySpare = 4 * kThreads + kCarryCtx;        // first pair of a buffer's head storage

struct CarryHeads {
    unsigned order;  // buffer index (0 = R, 1..3 = band slots) of head-storage area q in byte q
};
// LDS byte offset of pair v of the head storage (four areas of 128 pairs)
__device__ __forceinline__ int carry_head_addr(const CarryHeads& h, int v) {
    const int buf = (h.order >> ((v >> 7) << 3)) & 0xff;
    return buf * kRingBand + ((kCarrySpare + (v & 127)) << 4);
}

// One band of this tile into the buffer at `dst`; returns nothing, issues 4 instructions, or 5 on the first tile of a
// run when the wave's block of the fifth chunk holds needed pairs.
template <int MODE>
__device__ __forceinline__ void carry_issue(const double* __restrict__ src, unsigned lds_base, int dst, int S, int pl, int64_t t0,
                                            int64_t n, int wave, int lane, bool first, bool redge) {
    const int pair0 = blk_slot_rt(64 * wave + lane, S);
    const unsigned ldst = lds_base + dst + 1024 * wave;
    const bool fifth = first && (4 * kThreads + 64 * wave < pl);
    if (!redge) {
        const double* tsrc = src + t0;  // wave-uniform
        const unsigned off0 = static_cast<unsigned>(pair0) << 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) glds16_so(tsrc, off0 + 4096u * i, ldst + 4096u * i);
        if (fifth) {
            const int p4 = pair0 + 4 * kThreads;
            glds16_so(tsrc, static_cast<unsigned>(p4 < pl ? p4 : pl - 1) << 4, ldst + 4096u * 4);
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        if (i < 4 || fifth) {
            int pair = pair0 + i * kThreads;
            pair = pair < pl ? pair : pl - 1;
            int64_t t = t0 + 2 * pair;
            if (t >= n) t = (MODE == kPeriodic) ? t - n : n - 2;
            glds16(src + t, ldst + 4096u * i);
        }
    }
}

// write-back of R_{j-1} in the layout of the level that reads it (log2 stride SR); the owners of its first hp pairs also
// store them as the head for the next tile and put the previous tile's head (`oldh`, read out before the level's
// arithmetic so that no LDS round trip sits between the two barriers) into the context zone
template <int SR, int MODE>
__device__ __forceinline__ void carry_writeback(char* lds, const double (&acc)[5][2], int P0, int sp, int ph_own, bool hwr, int64_t t0,
                                                int64_t n, bool first, bool heads /*wave-uniform: this wave owns head pairs*/,
                                                const CarryHeads& hb, int vhead, int hp, const double2 (&oldh)[4]) {
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        if (k < 4 || hwr) {
            const int pair = (k < 4) ? P0 + k * sp : ph_own;
            double2 v = make_double2(acc[k][0], acc[k][1]);
            if (MODE == kZeroPad) {
                // samples right of the signal are zero at every level
                const int64_t t = t0 + 2 * pair;
                if (t >= n) v.x = 0.0;
                if (t + 1 >= n) v.y = 0.0;
            }
            *reinterpret_cast<double2*>(lds + (blk_slot<SR>(pair) << 4)) = v;
            if (heads && k < 4 && pair < hp) {
                *reinterpret_cast<double2*>(lds + carry_head_addr(hb, vhead + pair)) = v;
                if (!first) *reinterpret_cast<double2*>(lds + (blk_slot<SR>(4 * kThreads + pair) << 4)) = oldh[k];
            }
        }
    }
}

template <int LT, int MODE, bool FMA, bool PLAIN>
__global__ void __launch_bounds__(kThreads, 2) k_inv_carry(const InvArgs a) {
    extern __shared__ double2 vw_smem[];
    constexpr int PPT = 4, NP = 5;
    constexpr int T = PPT * kChunk;
    char* lds = reinterpret_cast<char*>(vw_smem);
    const unsigned lds_base = static_cast<unsigned>(reinterpret_cast<uintptr_t>(lds));  // LDS byte address of the R buffer
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;

    // a.ntiles: runs per row, counted from the right end of the row; a.nchunks: tiles per run
    const int64_t id = xcd_remap(blockIdx.x, gridDim.x);
    const int64_t row = id / a.ntiles;
    const int seg = static_cast<int>(id - row * a.ntiles);
    const int hi_tile = a.tiles_per_row - 1 - seg * a.nchunks;
    const int ntl = (hi_tile + 1 < a.nchunks) ? hi_tile + 1 : a.nchunks;
    const int64_t n = a.n;
    const double* __restrict__ wbase = a.win + row * a.w_row_stride;

    double tl[LT], th[LT];
#pragma unroll
    for (int i = 0; i < LT; ++i) {
        tl[i] = a.taps.lo[i];
        th[i] = a.taps.hi[i];
    }
    // head storage in the order: slot of band lev1 - 1, slot of band lev1 - 2, R buffer, slot of band lev1
    CarryHeads hb;
    hb.order = static_cast<unsigned>(1 + (a.lev1 + 2) % 3) | (static_cast<unsigned>(1 + (a.lev1 + 1) % 3) << 8) |
               (static_cast<unsigned>(1 + a.lev1 % 3) << 24);
    // R_j is valid on buffer positions [0, pend(j)) in the first tile of a run: the tile plus the right context that
    // levels j .. lev0 still need (the host's InvArgs::pend, recomputed here so that no scalar load sits in the level loop)
    const int lev0m = 1 << (a.lev0 - 1);
    auto pend = [&](int j) {
        if (j < a.lev0) return T;
        const int e = (LT - 1) * ((1 << j) - lev0m);
        const int pe = T + ((e + 1) & ~1);
        return pe < 2 * kBlkPairs ? pe : 2 * kBlkPairs;
    };

    for (int ch = 0; ch < ntl; ++ch) {
        const bool first = (ch == 0);
        const int64_t t0 = static_cast<int64_t>(hi_tile - ch) * T;
        // the first tile (with its halo) may reach beyond its row: loads wrap or clamp lane by lane, and a
        // ZERO_PADDING tile needs zeros there even in the plain form; later tiles lie inside the row
        const bool redge = first && (t0 + 2 * kBlkPairs > n);
        const bool edge = (MODE == kZeroPad) && redge;
        auto pairs_of = [&](int j) { return first ? (pend(j) >> 1) : 4 * kThreads; };
        auto issued = [&](int j) { return j >= a.lev0 ? 4 + ((first && 4 * kThreads + 64 * wave < (pend(j) >> 1)) ? 1 : 0) : 0; };

        // prologue: R_{lev1}, W_{lev1}, W_{lev1-1}
        carry_issue<MODE>(a.rin + row * a.rin_stride, lds_base, 0, a.lev1 - 2, pairs_of(a.lev1), t0, n, wave, lane, first, redge);
        carry_issue<MODE>(wbase + static_cast<int64_t>(a.lev1 - 1) * a.w_band_stride, lds_base, (1 + a.lev1 % 3) * kRingBand, a.lev1 - 2,
                          pairs_of(a.lev1), t0, n, wave, lane, first, redge);
        if (a.lev1 - 1 >= a.lev0)
            carry_issue<MODE>(wbase + static_cast<int64_t>(a.lev1 - 2) * a.w_band_stride, lds_base, (1 + (a.lev1 - 1) % 3) * kRingBand,
                              a.lev1 - 3, pairs_of(a.lev1 - 1), t0, n, wave, lane, first, redge);

        int vR = 0, vW = kCarryCtx;  // head storage of level j: R_j at pair vR, W_j at pair vW
        for (int j = a.lev1; j >= a.lev0; --j) {
            const int S = j - 2;
            const int hpj = (((LT - 1) << (j - 1)) + 1) >> 1;  // pairs of right context of level j
            // two levels ahead; its slot was last read at level j+1, whose readers all passed that level's second barrier
            if (j - 2 >= a.lev0)
                carry_issue<MODE>(wbase + static_cast<int64_t>(j - 3) * a.w_band_stride, lds_base, (1 + (j - 2) % 3) * kRingBand, j - 4,
                                  pairs_of(j - 2), t0, n, wave, lane, first, redge);
            // the previous tile's heads are read out while the band is still on its way: those of W_j (and of R_j at the top
            // level) by the lanes of the two waves whose first block holds pairs 0..127, that of R_{j-1} by the lanes that will
            // write those pairs back (the ownership of a level is the same in every tile)
            const int hi = 64 * wave + lane;
            const bool hx = (wave < 2) && (hi < hpj);
            const int vRn = (j == a.lev1) ? 2 * kCarryCtx : vW + hpj;  // head storage of the next level down
            const int hpn = (j >= 2) ? ((((LT - 1) << (j - 2)) + 1) >> 1) : 0;
            const int spj = (j >= 2) ? (1 << S) : 0;
            const int P0j = (j >= 2) ? (((tid >> S) << (S + 2)) + (tid & (spj - 1))) : 0;
            double2 oldw = make_double2(0.0, 0.0), oldr = make_double2(0.0, 0.0), oldn[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) oldn[k] = make_double2(0.0, 0.0);
            // (wave-uniform conditions and clamped addresses: no lane-dependent control flow around the reads)
            if (!first) {
                if (wave < 2) {
                    const int hc = hx ? hi : 0;
                    oldw = *reinterpret_cast<const double2*>(lds + carry_head_addr(hb, vW + hc));
                    if (j == a.lev1) oldr = *reinterpret_cast<const double2*>(lds + carry_head_addr(hb, vR + hc));
                }
                if (wave == 0 && j > a.lev0) {  // pairs below 128 all belong to lanes of the first wave
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int pair = P0j + k * spj;
                        oldn[k] = *reinterpret_cast<const double2*>(lds + carry_head_addr(hb, vRn + (pair < hpn ? pair : 0)));
                    }
                }
            }
            // everything older than the two youngest bands has landed: R (first pass) and W_j
            if (!PLAIN) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            } else {
                switch (issued(j - 1) + issued(j - 2)) {
                    case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
                    case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
                    case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
                    case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
                    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
                    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
                }
            }
            const int wofs = (1 + j % 3) * kRingBand;
            if (!PLAIN || edge) {
                // pass over the slots this wave has filled: W_j (and R on the first level)
                const double thr = (a.thr_mode != 0) ? uniform_load_f64(a.thr + row * a.thr_row_stride + (j - 1) * a.thr_level_stride) : 0.0;
                const bool use_w = (a.band_mask >> (j - 1)) & 1u;
                const bool use_r = (a.band_mask >> 31) & 1u;
                const int pl = pairs_of(j);
                const int nfill = issued(j);
#pragma unroll
                for (int i = 0; i < 5; ++i) {
                    if (i < nfill) {
                        const int slot = 64 * (wave + 4 * i) + lane;
                        const int pair = blk_slot_rt(slot, S);
                        const int64_t t = t0 + 2 * pair;
                        const bool inrow = (MODE == kPeriodic) || (t < n);
                        double2* pw = reinterpret_cast<double2*>(lds + wofs + 16 * slot);
                        double2 w = *pw;
                        if (!use_w || !inrow || pair >= pl) w = make_double2(0.0, 0.0);
                        if (a.thr_mode != 0) {
                            w.x = apply_thr(w.x, thr, a.thr_mode);
                            w.y = apply_thr(w.y, thr, a.thr_mode);
                        }
                        if (a.first_bad != nullptr && pair < kThreads * PPT && t < n) {
                            const int64_t flat0 = row * a.w_row_stride + static_cast<int64_t>(j - 1) * a.w_band_stride;
                            if (!finite_f64(w.x)) atomicMin(a.first_bad, static_cast<unsigned long long>(flat0 + t));
                            if (!finite_f64(w.y)) atomicMin(a.first_bad, static_cast<unsigned long long>(flat0 + t + 1));
                        }
                        *pw = w;
                        if (j == a.lev1) {
                            double2* pr = reinterpret_cast<double2*>(lds + 16 * slot);
                            double2 r = *pr;
                            if (!use_r || !inrow || pair >= pl) r = make_double2(0.0, 0.0);
                            if (a.first_bad != nullptr && a.check_rin && pair < kThreads * PPT && t < n) {
                                const int64_t flat0 = row * a.w_row_stride + static_cast<int64_t>(a.lev1) * a.w_band_stride;
                                if (!finite_f64(r.x)) atomicMin(a.first_bad, static_cast<unsigned long long>(flat0 + t));
                                if (!finite_f64(r.y)) atomicMin(a.first_bad, static_cast<unsigned long long>(flat0 + t + 1));
                            }
                            *pr = r;
                        }
                    }
                }
            }
            // the previous tile's heads become this tile's right context
            const int hown = blk_slot_rt(hi, S) << 4, hctx = blk_slot_rt(4 * kThreads + hi, S) << 4;
            if (hx && !first) {
                *reinterpret_cast<double2*>(lds + wofs + hctx) = oldw;
                if (j == a.lev1) *reinterpret_cast<double2*>(lds + hctx) = oldr;
            }
            lds_barrier();  // every wave's part of W_j has landed, R_j and both contexts are written

            // this tile's heads, requested ahead of the operand windows and stored behind the arithmetic
            double2 neww = make_double2(0.0, 0.0), newr = make_double2(0.0, 0.0);
            if (hx) {
                neww = *reinterpret_cast<const double2*>(lds + wofs + hown);
                if (j == a.lev1) newr = *reinterpret_cast<const double2*>(lds + hown);
            }

            const int pe = (first && j > a.lev0) ? pend(j - 1) : T;  // later tiles and the last level only need the tile
            const bool hact = (T + 128 * wave) < pe;
            const int ph_own = kThreads * PPT + tid;
            const bool hwr = 2 * ph_own < pe;
            const int ph = hwr ? ph_own : (pe >> 1) - 1;
            double acc[NP][2];
            int P0 = 0;
            acc[4][0] = acc[4][1] = 0.0;
#define VW_CARRY_LEVEL(SS, SLOT)                                                                        \
    inv_blk_main<LT, SS, FMA, (1 + SLOT) * kRingBand, true>(lds, tid, S, tl, th, acc, P0);              \
    if (hact) inv_blk_halo<LT, SS, FMA, (1 + SLOT) * kRingBand, true>(lds, ph, S, tl, th, acc);
            switch (j) {
                case 1: inv_blk_level1<LT, FMA, (1 + 1) * kRingBand, true>(lds, tid, tl, th, acc); break;
                case 2: VW_CARRY_LEVEL(0, 2) break;
                case 3: VW_CARRY_LEVEL(1, 0) break;
                case 4: VW_CARRY_LEVEL(2, 1) break;
                case 5: VW_CARRY_LEVEL(3, 2) break;
                case 6: VW_CARRY_LEVEL(4, 0) break;
                case 7: VW_CARRY_LEVEL(4, 1) break;
                default: VW_CARRY_LEVEL(4, 2) break;  // level 8
            }
#undef VW_CARRY_LEVEL
            const int sp = spj;
            if (hx) {
                *reinterpret_cast<double2*>(lds + carry_head_addr(hb, vW + hi)) = neww;
                if (j == a.lev1) *reinterpret_cast<double2*>(lds + carry_head_addr(hb, vR + hi)) = newr;
            }

            if (j == a.lev0) {
                double* __restrict__ orow = a.rout + row * a.rout_stride;
#pragma unroll
                for (int k = 0; k < PPT; ++k) {
                    const int pair = (j == 1) ? k * kThreads + tid : P0 + k * sp;
                    const int64_t t = t0 + 2 * pair;
                    if (t < a.n_store) gstore2(orow + t, make_double2(acc[k][0], acc[k][1]), a.nt_store);
                }
            } else {
                lds_barrier();  // every lane has finished reading R_j / W_j
                // R_{j-1} in the layout of level j-1 (j > lev0 >= 1: a blocked level wrote these accumulators)
                switch (j) {
                    case 2: carry_writeback<-1, MODE>(lds, acc, P0, sp, ph_own, hwr, t0, n, first, wave == 0, hb, vRn, hpn, oldn); break;
                    case 3: carry_writeback<0, MODE>(lds, acc, P0, sp, ph_own, hwr, t0, n, first, wave == 0, hb, vRn, hpn, oldn); break;
                    case 4: carry_writeback<1, MODE>(lds, acc, P0, sp, ph_own, hwr, t0, n, first, wave == 0, hb, vRn, hpn, oldn); break;
                    case 5: carry_writeback<2, MODE>(lds, acc, P0, sp, ph_own, hwr, t0, n, first, wave == 0, hb, vRn, hpn, oldn); break;
                    case 6: carry_writeback<3, MODE>(lds, acc, P0, sp, ph_own, hwr, t0, n, first, wave == 0, hb, vRn, hpn, oldn); break;
                    default: carry_writeback<4, MODE>(lds, acc, P0, sp, ph_own, hwr, t0, n, first, wave == 0, hb, vRn, hpn, oldn); break;
                }
                vR = vRn;
                vW = vRn + hpn;
            }
        }
        if (ch + 1 < ntl) lds_barrier();  // the last level's reads are done before the next tile's bands land
    }
}

// streaming inverse for long contexts: mirror image of k_fwd_sdry
// ---------------------------------------------------------------------------
//
// A workgroup walks a run of tiles of one row from right to left.  The right context of level j is the
// head (first hlen[j] samples) of R_j and of the thresholded W_j of the tile processed just before,
// kept in LDS; the run starts `ndry` tiles to the right of its last tile with the stores switched off,
// after which every history is exact (FIR cascade).  While a level is computed, the next detail band
// (or, at the last level, the next tile's R and first detail band) is in flight from HBM.
template <int NT, int PPT, int HB, int MODE, int VEC2, bool FMA, int LT = 0>
__global__ void __launch_bounds__(NT) k_inv_sdry(const InvArgs a) {
    extern __shared__ double2 vw_smem[];
    constexpr int HALO = HB * kChunk;
    constexpr int CH = 2 * NT;
    constexpr int T = PPT * CH;
    constexpr int M = T + HALO;
    double* bufR = reinterpret_cast<double*>(vw_smem);
    double* bufW = bufR + M;
    double* hstR = bufW + M;           // a.hist_cap doubles
    double* hstW = hstR + a.hist_cap;  // a.hist_cap doubles
    const char* ldsR = reinterpret_cast<const char*>(bufR);
    const char* ldsW = reinterpret_cast<const char*>(bufW);
    const int tid = threadIdx.x;

    const int64_t id = xcd_remap(blockIdx.x, gridDim.x);
    const int64_t row = id / a.ntiles;
    const int seg = static_cast<int>(id - row * a.ntiles);
    const int tile0 = seg * a.nchunks;
    int ntl = a.tiles_per_row - tile0;
    if (ntl > a.nchunks) ntl = a.nchunks;
    const int64_t n = a.n;
    const double* __restrict__ wbase = a.win + row * a.w_row_stride;
    const double* __restrict__ rrow = a.rin + row * a.rin_stride;
    const bool use_r = (a.band_mask >> 31) & 1u;
    // right of a ZERO_PADDING row everything is zero at every level: zeroed histories are already exact
    const int ndry = (MODE == kZeroPad && tile0 + ntl == a.tiles_per_row) ? 0 : a.ndry;

    for (int i = 2 * tid; i < 2 * a.hist_cap; i += 2 * NT) *reinterpret_cast<double2*>(hstR + i) = make_double2(0.0, 0.0);
    lds_barrier();

    // Tile loader for any position.  The loads must stay in flight across a whole level (that is their point), so nothing
    // may touch the loaded registers until the band is published one level later: every lane loads from an address inside
    // the row -- wrapped (PERIODIC) or clamped (ZERO_PADDING) -- and zeros for samples right of a ZERO_PADDING row or for
    // a band that the mask switches off are put in by `settle` when the band is published.  (With the zero fill at load
    // time the compiler merged the loaded value with the constant right behind the load and waited for it there: the
    // memory latency of every level lay open, and the kernel took its memory time plus its arithmetic time.)
    // (the byte offsets of a lane's samples inside a row depend on the tile alone: they are worked out once per tile, not
    // once per level -- with sixteen waves in lock-step every instruction of the level loop costs the CU sixteen issue slots)
    unsigned boa[PPT], bob[PPT];
    auto tile_offsets = [&](int64_t t0) {
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            int64_t ia = t0 + k * CH + 2 * tid, ib = ia + 1;
            if (MODE == kPeriodic) {
                if (ia >= n) ia = (ia < 2 * n) ? ia - n : ia % n;
                if (ib >= n) ib = (ib < 2 * n) ? ib - n : ib % n;
            } else {
                if (ia >= n) ia = (VEC2 == 1) ? n - 2 : n - 1;
                if (ib >= n) ib = n - 1;
            }
            if (VW_ABLATED(a, 1)) {  // (tuning build: every lane reads the start of the row -- the same loads, no memory time)
                ia = 0;
                ib = 1 < n ? 1 : 0;
            }
            boa[k] = static_cast<unsigned>(ia) * 8u;  // rows of less than 2^29 samples (the planner checks)
            bob[k] = static_cast<unsigned>(ib) * 8u;
        }
    };
    auto load_tile = [&](const double* __restrict__ src, double2(&v)[PPT]) {
        const char* __restrict__ base = reinterpret_cast<const char*>(src);  // wave-uniform
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            if (VEC2 == 1) {  // even n, aligned rows: a pair never straddles the row end
                v[k] = *reinterpret_cast<const double2*>(base + boa[k]);
            } else {
                v[k].x = *reinterpret_cast<const double*>(base + boa[k]);
                v[k].y = *reinterpret_cast<const double*>(base + bob[k]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);  // issued here, not where the scheduler finds room for them
    };
    auto settle = [&](bool use, int64_t t0, double2(&v)[PPT]) {
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            const int64_t t = t0 + k * CH + 2 * tid;
            if (!use || (MODE != kPeriodic && t >= n)) v[k].x = 0.0;
            if (!use || (MODE != kPeriodic && t + 1 >= n)) v[k].y = 0.0;
        }
    };
    // the lanes that own the first hlen[j] samples of the tile swap them with the stored history, which
    // becomes the right context of this tile
    auto head_swap = [&](double* hst, double* buf, int j, const double2(&vals)[PPT]) {
        const int hl = a.hlen[j];
        double* h0 = hst + a.hoff[j];
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            const int i = k * CH + 2 * tid;
            if (i < hl) {
                double2* h = reinterpret_cast<double2*>(h0 + i);
                const double2 old = *h;
                *h = vals[k];
                *reinterpret_cast<double2*>(buf + T + i) = old;
            }
        }
    };

    bool wact[PPT];
    int pb[PPT];
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        wact[k] = true;
        pb[k] = (k * CH + 2 * tid) * 8;
    }
    double2 pre[PPT];   // detail band on its way into bufW
    double2 preR[PPT];  // R_{lev1} of the tile about to be staged
    for (int ch = -ndry; ch < ntl; ++ch) {
        const int64_t t0 = static_cast<int64_t>(tile0 + ntl - 1 - ch) * T;
        const bool dry = ch < 0;
        if (ch == -ndry) {
            tile_offsets(t0);
            load_tile(rrow, preR);
            load_tile(wbase + static_cast<int64_t>(a.lev1 - 1) * a.w_band_stride, pre);
        }
        settle(use_r, t0, preR);
        if (a.first_bad != nullptr && a.check_rin && !dry) {
            const int64_t flat0 = row * a.w_row_stride + static_cast<int64_t>(a.lev1) * a.w_band_stride;
#pragma unroll
            for (int c = 0; c < PPT; ++c) {
                const int64_t t = t0 + c * CH + 2 * tid;
                if (t < n && !finite_f64(preR[c].x))
                    atomicMin(a.first_bad, static_cast<unsigned long long>(flat0 + t));
                if (t + 1 < n && !finite_f64(preR[c].y))
                    atomicMin(a.first_bad, static_cast<unsigned long long>(flat0 + t + 1));
            }
        }
#pragma unroll
        for (int c = 0; c < PPT; ++c) *reinterpret_cast<double2*>(bufR + c * CH + 2 * tid) = preR[c];
        head_swap(hstR, bufR, a.lev1, preR);

        for (int j = a.lev1; j >= a.lev0; --j) {
            const int s = 1 << (j - 1);
            const double thr = (a.thr_mode != 0) ? uniform_load_f64(a.thr + row * a.thr_row_stride + (j - 1) * a.thr_level_stride) : 0.0;
            // ---- publish W_j (thresholded on the way in when denoising) ---------------
            settle(((a.band_mask >> (j - 1)) & 1u) != 0, t0, pre);
#pragma unroll
            for (int c = 0; c < PPT; ++c) {
                double2 w = pre[c];
                if (a.thr_mode != 0) {
                    w.x = apply_thr(w.x, thr, a.thr_mode);
                    w.y = apply_thr(w.y, thr, a.thr_mode);
                }
                if (a.first_bad != nullptr && !dry) {
                    const int64_t flat0 = row * a.w_row_stride + static_cast<int64_t>(j - 1) * a.w_band_stride;
                    const int64_t t = t0 + c * CH + 2 * tid;
                    if (t < n && !finite_f64(w.x))
                        atomicMin(a.first_bad, static_cast<unsigned long long>(flat0 + t));
                    if (t + 1 < n && !finite_f64(w.y))
                        atomicMin(a.first_bad, static_cast<unsigned long long>(flat0 + t + 1));
                }
                pre[c] = w;
                *reinterpret_cast<double2*>(bufW + c * CH + 2 * tid) = w;
            }
            head_swap(hstW, bufW, j, pre);
            lds_barrier();  // R_j and W_j visible

            // ---- loads that fly while this level is computed --------------------------
            // the next detail band of this tile, or at the last level the first band of the next tile (and its R).  ONE
            // load site for the band, taken on every path: two sites, or a site inside a branch, meet in a register
            // copy right behind the load, and the copy waits for the data.  (Nothing is left to fetch at the last level of
            // the last tile: it loads its own band again, unused.)
            {
                const bool nxt = (j == a.lev0), more = (ch + 1 < ntl);
                const int jb = nxt ? a.lev1 - 1 : j - 2;
                if (nxt && more) tile_offsets(t0 - T);
                load_tile(wbase + static_cast<int64_t>(jb) * a.w_band_stride, pre);
                if (nxt && more) load_tile(rrow, preR);
            }

            double acc[PPT][2];
#pragma unroll
            for (int k = 0; k < PPT; ++k) acc[k][0] = acc[k][1] = 0.0;
            if (s == 1)
                inv_level<PPT, PPT, true, FMA, LT>(ldsR, ldsW, a, s, acc, pb, wact);
            else
                inv_level<PPT, PPT, false, FMA, LT>(ldsR, ldsW, a, s, acc, pb, wact);

            if (j == a.lev0) {
                if (!dry) {
                    double* __restrict__ orow = a.rout + row * a.rout_stride;
#pragma unroll
                    for (int k = 0; k < PPT; ++k) {
                        const int64_t t = t0 + k * CH + 2 * tid;
                        gstore_pair<VEC2>(orow, t, a.n_store, acc[k][0], acc[k][1], a.nt_store);
                    }
                }
                if (ch + 1 < ntl) lds_barrier();  // last reads of this tile done before it is restaged
            } else {
                lds_barrier();  // every lane has finished reading R_j / W_j
                double2 v2[PPT];
#pragma unroll
                for (int k = 0; k < PPT; ++k) {
                    v2[k] = make_double2(acc[k][0], acc[k][1]);
                    if (MODE == kZeroPad) {
                        // samples right of the signal are zero at every level
                        const int64_t t = t0 + k * CH + 2 * tid;
                        if (t >= n) v2[k].x = 0.0;
                        if (t + 1 >= n) v2[k].y = 0.0;
                    }
                    *reinterpret_cast<double2*>(bufR + k * CH + 2 * tid) = v2[k];
                }
                head_swap(hstR, bufR, j - 1, v2);
                // the barrier at the top of the next iteration publishes R_{j-1}
            }
        }
    }
}

}  // namespace vw
