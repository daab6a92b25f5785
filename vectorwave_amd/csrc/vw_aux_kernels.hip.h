// vw_aux_kernels.hip.h -- the non-tiled gfx950 kernels of the MODWT / SWT engine: generic one-level
// transforms (any boundary, any N / L), thresholding, finite check, exact median (universal threshold)
// and the WaveletDenoiser threshold rules.  Same arithmetic contract as vw_kernels.hip.h.
#pragma once
#include "vw_kernels.hip.h"

#pragma clang fp contract(off)

namespace vw {

// ---------------------------------------------------------------------------
// generic one-level kernels (any boundary, any N / L, truncated filters):
// one lane per output sample, operands through the cache hierarchy.  Used for
// SYMMETRIC, for filters longer than the signal and for halos that exceed LDS.
// ---------------------------------------------------------------------------

struct LevelArgs {
    const double* in;   // forward: V_{j-1}; inverse: R_j
    int64_t in_stride;
    const double* din;  // inverse only: W_j
    int64_t din_stride;
    double* out_lo;     // forward: V_j; inverse: R_{j-1}
    int64_t out_lo_stride;
    double* out_hi;     // forward: W_j (may be nullptr)
    int64_t out_hi_stride;
    int64_t n;
    int64_t nrows;
    int64_t stride;     // 2^(j-1)
    int32_t llo, lhi;   // taps used of lo / hi (after the reference's truncation)
    int32_t variant;    // inverse SYMMETRIC flavour, see k_level_inv
    int32_t use_in, use_din;  // 0 = treat the operand as all zeros
    // inverse only: W_j is thresholded as it is read (apply_thr mode; 0 = none), threshold of row r at thr[r * thr_row_stride + thr_off]
    int32_t thr_mode;
    const double* thr;
    int64_t thr_row_stride, thr_off;
    Taps taps;
};

template <int MODE>
__global__ void __launch_bounds__(kThreads) k_level_fwd(const LevelArgs a) {
    const int64_t gid = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
    const int64_t per_row = (a.n + kThreads - 1) / kThreads * kThreads;
    const int64_t row = gid / per_row;
    const int64_t t = gid - row * per_row;
    if (row >= a.nrows || t >= a.n) return;
    const double* __restrict__ x = a.in + row * a.in_stride;
    const int64_t n = a.n;
    const int lmax = a.llo > a.lhi ? a.llo : a.lhi;
    double slo = 0.0, shi = 0.0;
    for (int l = 0; l < lmax; ++l) {
        int64_t idx = t - static_cast<int64_t>(l) * a.stride;
        bool ok = true;
        if (MODE == kPeriodic) {
            if (idx < 0) idx = (idx >= -n) ? idx + n : mod_wrap(idx, n);
        } else if (MODE == kZeroPad) {
            ok = idx >= 0;
        } else {
            idx = sym_reflect(idx, n);
        }
        if (ok) {
            const double v = x[idx];
            if (l < a.llo) slo = slo + v * a.taps.lo[l];
            if (l < a.lhi) shi = shi + v * a.taps.hi[l];
        }
    }
    if (a.out_lo != nullptr) a.out_lo[row * a.out_lo_stride + t] = slo;
    if (a.out_hi != nullptr) a.out_hi[row * a.out_hi_stride + t] = shi;
}

// variant (SYMMETRIC only):
//   0  MultiLevelMODWTTransform.applyScaledInverseMODWT: reflect(t - l*s), the two
//      products added to the sum one after the other, per-filter length guards
//   1  MODWTTransform.inverse: reflect(t - l), pair-sum, l < llo
//   2  MODWTTransform.inverseBatchOptimized: reflect(t + l), pair-sum, l < llo
template <int MODE>
__global__ void __launch_bounds__(kThreads) k_level_inv(const LevelArgs a) {
    const int64_t gid = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
    const int64_t per_row = (a.n + kThreads - 1) / kThreads * kThreads;
    const int64_t row = gid / per_row;
    const int64_t t = gid - row * per_row;
    if (row >= a.nrows || t >= a.n) return;
    const double* __restrict__ ap = a.in + row * a.in_stride;
    const double* __restrict__ dp = a.din + row * a.din_stride;
    const int64_t n = a.n;
    const bool shrink = a.thr_mode != 0 && a.use_din;
    const double thr = shrink ? a.thr[row * a.thr_row_stride + a.thr_off] : 0.0;
    double sum = 0.0;
    if (MODE == kSymmetric && a.variant == 0) {
        const int lmax = a.llo > a.lhi ? a.llo : a.lhi;
        for (int l = 0; l < lmax; ++l) {
            const int64_t idx = sym_reflect(t - static_cast<int64_t>(l) * a.stride, n);
            if (l < a.llo) sum = sum + a.taps.lo[l] * (a.use_in ? ap[idx] : 0.0);
            if (l < a.lhi) {
                double dv = a.use_din ? dp[idx] : 0.0;
                if (shrink) dv = apply_thr(dv, thr, a.thr_mode);
                sum = sum + a.taps.hi[l] * dv;
            }
        }
    } else {
        for (int l = 0; l < a.llo; ++l) {
            int64_t idx;
            bool ok = true;
            if (MODE == kPeriodic) {
                idx = (t + static_cast<int64_t>(l) * a.stride) % n;
            } else if (MODE == kZeroPad) {
                idx = t + static_cast<int64_t>(l) * a.stride;
                ok = idx < n;
            } else if (a.variant == 1) {
                idx = sym_reflect(t - static_cast<int64_t>(l) * a.stride, n);
            } else {
                idx = sym_reflect(t + static_cast<int64_t>(l) * a.stride, n);
            }
            if (ok) {
                const double av = a.use_in ? ap[idx] : 0.0;
                double dv = a.use_din ? dp[idx] : 0.0;
                if (shrink) dv = apply_thr(dv, thr, a.thr_mode);
                sum = sum + (a.taps.lo[l] * av + a.taps.hi[l] * dv);
            }
        }
    }
    a.out_lo[row * a.out_lo_stride + t] = sum;
}

// ---------------------------------------------------------------------------
// element-wise kernels
// ---------------------------------------------------------------------------

__global__ void __launch_bounds__(kThreads) k_threshold(const double* __restrict__ in, double* __restrict__ out,
                                                        int64_t count, double thr, int mode) {
    int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
    const int64_t step = static_cast<int64_t>(gridDim.x) * kThreads;
    for (; i < count; i += step) out[i] = apply_thr(in[i], thr, mode);
}

__global__ void __launch_bounds__(kThreads) k_first_non_finite(const double* __restrict__ in, int64_t count,
                                                               unsigned long long* first_bad) {
    int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
    const int64_t step = static_cast<int64_t>(gridDim.x) * kThreads;
    unsigned long long best = ~0ull;
    for (; i < count; i += step)
        if (!finite_f64(in[i]) && static_cast<unsigned long long>(i) < best) best = static_cast<unsigned long long>(i);
    if (best != ~0ull) atomicMin(first_bad, best);
}

// ---------------------------------------------------------------------------
// universal threshold: exact median of |d_1| per row by radix select
// ---------------------------------------------------------------------------
//
// |x| of a finite double orders like its bit pattern as an unsigned integer, so
// the k-th order statistic is found by an 8-bit-per-pass radix descent over the
// keys.  One workgroup per row; the 256-bin histogram of a pass lives in LDS, the
// bin that holds rank k is located with a wavefront prefix scan (DPP shuffles)
// plus a 4-entry cross-wave fix-up.  Even N needs ranks N/2-1 and N/2 (the
// reference averages the two middle elements of the sorted array): the second
// one is either the same key (duplicates) or the smallest key above the first.

__device__ __forceinline__ unsigned wave_inclusive_scan(unsigned v) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned up = __shfl_up(v, d, 64);
        if ((threadIdx.x & 63) >= d) v += up;
    }
    return v;
}

// Key source over a row of doubles: key = bit pattern of | |v_i| - sub | (with sub = 0 the keys are |v_i|).
// LDS only names the address space the pointer lives in: one instantiation per space keeps the loads
// ds_read / global_load instead of flat.
template <bool LDS>
struct KeySrc {
    const double* v;
    double sub;
    __device__ __forceinline__ double raw(int64_t i) const { return v[i]; }
    __device__ __forceinline__ unsigned long long get(int64_t i) const {
        return static_cast<unsigned long long>(__double_as_longlong(fabs(fabs(v[i]) - sub))) & 0x7fffffffffffffffull;
    }
};

// Copies (|.| of) a row into LDS with eight loads per lane in flight.
template <int NT, bool ABS>
__device__ __forceinline__ void stage_row(double* dst, const double* __restrict__ src, int64_t count) {
    int64_t i = threadIdx.x;
    for (; i + 7 * NT < count; i += 8 * NT) {
        double t[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) t[j] = src[i + j * NT];
#pragma unroll
        for (int j = 0; j < 8; ++j) dst[i + j * NT] = ABS ? fabs(t[j]) : t[j];
    }
    for (; i < count; i += NT) dst[i] = ABS ? fabs(src[i]) : src[i];
}

// Returns the key of rank k (0-based) among n keys; all threads get the result.
template <typename Src>
__device__ unsigned long long radix_select(const Src& src, int64_t n, int64_t k, unsigned* hist /*256*/,
                                           unsigned* wsum /*4*/, unsigned long long* bcast /*2*/) {
    const int tid = threadIdx.x;
    unsigned long long prefix = 0;
    int64_t kk = k;
    for (int shift = 56; shift >= 0; shift -= 8) {
        hist[tid] = 0;
        __syncthreads();
        const unsigned long long himask = (shift == 56) ? 0ull : (~0ull << (shift + 8));
        for (int64_t i = tid; i < n; i += kThreads) {
            const unsigned long long key = src.get(i);
            if ((key & himask) == prefix) atomicAdd(&hist[(key >> shift) & 0xff], 1u);
        }
        __syncthreads();
        // inclusive scan of the 256 bins: bin `tid` belongs to lane tid
        const unsigned cnt = hist[tid];
        unsigned inc = wave_inclusive_scan(cnt);
        if ((tid & 63) == 63) wsum[tid >> 6] = inc;
        __syncthreads();
        unsigned add = 0;
        for (int w = 0; w < (tid >> 6); ++w) add += wsum[w];
        inc += add;
        const unsigned exc = inc - cnt;
        if (cnt != 0 && static_cast<int64_t>(exc) <= kk && kk < static_cast<int64_t>(inc)) {
            bcast[0] = prefix | (static_cast<unsigned long long>(tid) << shift);
            bcast[1] = static_cast<unsigned long long>(kk - exc);
        }
        __syncthreads();
        prefix = bcast[0];
        kk = static_cast<int64_t>(bcast[1]);
        __syncthreads();
    }
    return prefix;
}

// Median of n non-negative keys the way the reference takes it from a sorted copy: the middle element,
// or the mean of the two middle elements for even n.  All threads call it and all get the result.
template <typename Src>
__device__ double median_select(const Src& src, int64_t n, unsigned* hist /*256*/, unsigned* wsum /*4*/,
                                unsigned long long* bcast /*2*/, unsigned long long* red /*kThreads*/) {
    const int tid = threadIdx.x;
    const int64_t k1 = (n % 2 == 0) ? n / 2 - 1 : n / 2;
    const unsigned long long v1 = radix_select(src, n, k1, hist, wsum, bcast);
    if (n % 2 != 0) return __longlong_as_double(static_cast<long long>(v1));
    // rank k1+1: v1 again if at least k1+2 keys are <= v1, else the smallest key above v1
    unsigned long long le = 0, above = ~0ull;
    for (int64_t i = tid; i < n; i += kThreads) {
        const unsigned long long key = src.get(i);
        if (key <= v1)
            ++le;
        else if (key < above)
            above = key;
    }
    red[tid] = le;
    __syncthreads();
    for (int w = kThreads / 2; w > 0; w >>= 1) {
        if (tid < w) red[tid] += red[tid + w];
        __syncthreads();
    }
    const unsigned long long le_total = red[0];
    __syncthreads();
    red[tid] = above;
    __syncthreads();
    for (int w = kThreads / 2; w > 0; w >>= 1) {
        if (tid < w && red[tid + w] < red[tid]) red[tid] = red[tid + w];
        __syncthreads();
    }
    const unsigned long long v2 = (static_cast<int64_t>(le_total) >= k1 + 2) ? v1 : red[0];
    __syncthreads();
    return (__longlong_as_double(static_cast<long long>(v1)) + __longlong_as_double(static_cast<long long>(v2))) / 2.0;
}


// wavefront-wide reductions (every lane gets the result)
__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        const unsigned long long o = __shfl_xor(v, d, 64);
        v = o < v ? o : v;
    }
    return v;
}

// LDS scratch of the 256-lane selection helpers
template <int NT>
struct SelLds {
    __align__(16) unsigned hist[256];
    unsigned wsum[4];
    unsigned long long bcast[2];
    unsigned long long red[kThreads];
};

template <int NT, typename Src>
__device__ __forceinline__ double median_nt(const Src& src, int64_t n, SelLds<NT>& l) {
    static_assert(NT == kThreads, "the LDS-histogram selection runs with 256 lanes");
    return median_select(src, n, l.hist, l.wsum, l.bcast, l.red);
}

// sum of one value per lane over the workgroup; every lane gets it
template <int NT>
__device__ __forceinline__ unsigned long long sum_nt(unsigned long long v, SelLds<NT>& l) {
    const int tid = threadIdx.x;
    l.red[tid] = v;
    __syncthreads();
    for (int s = kThreads / 2; s > 0; s >>= 1) {
        if (tid < s) l.red[tid] += l.red[tid + s];
        __syncthreads();
    }
    const unsigned long long r = l.red[0];
    __syncthreads();
    return r;
}

// ---------------------------------------------------------------------------
// selection over a row held in registers (one wavefront per row, rows of up to 64 * KPL samples)
// ---------------------------------------------------------------------------
//
// Lane L holds the non-negative doubles v[j] = row[j * 64 + L]; their bit patterns are the keys.  Entries past
// the end of the row hold the all-ones pattern, which is above every |double| and every threshold used here.
// The rank-k key is found by descending its bits: thresholds split the current range (in two, or in four with
// three thresholds), every key is compared with them, and ballots count the keys below each.  No LDS, no atomics (a
// histogram in LDS serialises the many lanes that hit one bin, and all the wavefronts of a CU share that
// unit), no barriers; bits that all keys share are skipped, and the descent stops as soon as one key is
// left in the range.

__device__ __forceinline__ double reg_pad() { return __longlong_as_double(-1ll); }

// Reductions over the wavefront that end on the scalar unit (every lane of the wavefront must be active): four DPP
// steps -- neighbour, other pair, mirrored half row, mirrored row -- leave each row of 16 lanes with its own result,
// and the four rows are combined from scalar registers.  No LDS traffic (a shuffle is an LDS instruction with its
// latency), and the result is known to the compiler as wave-uniform, so what is derived from it (the bit position,
// the prefix, the counts, the loop of the descent) lives in scalar registers and branches on the scalar unit.
template <int CTRL>
__device__ __forceinline__ unsigned dpp_u32(unsigned v) {
    return static_cast<unsigned>(__builtin_amdgcn_update_dpp(static_cast<int>(v), static_cast<int>(v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ unsigned lane_u32(unsigned v, int lane) {
    return static_cast<unsigned>(__builtin_amdgcn_readlane(static_cast<int>(v), lane));
}
constexpr int kDppSwap1 = 0xb1, kDppSwap2 = 0x4e, kDppHalfMirror = 0x141, kDppMirror = 0x140;

__device__ __forceinline__ unsigned wave_and_u32(unsigned v) {
    v &= dpp_u32<kDppSwap1>(v);
    v &= dpp_u32<kDppSwap2>(v);
    v &= dpp_u32<kDppHalfMirror>(v);
    v &= dpp_u32<kDppMirror>(v);
    return (lane_u32(v, 0) & lane_u32(v, 16)) & (lane_u32(v, 32) & lane_u32(v, 48));
}

__device__ __forceinline__ unsigned wave_or_u32(unsigned v) {
    v |= dpp_u32<kDppSwap1>(v);
    v |= dpp_u32<kDppSwap2>(v);
    v |= dpp_u32<kDppHalfMirror>(v);
    v |= dpp_u32<kDppMirror>(v);
    return (lane_u32(v, 0) | lane_u32(v, 16)) | (lane_u32(v, 32) | lane_u32(v, 48));
}

__device__ __forceinline__ unsigned wave_umin_u32(unsigned v) {
    v = min(v, dpp_u32<kDppSwap1>(v));
    v = min(v, dpp_u32<kDppSwap2>(v));
    v = min(v, dpp_u32<kDppHalfMirror>(v));
    v = min(v, dpp_u32<kDppMirror>(v));
    return min(min(lane_u32(v, 0), lane_u32(v, 16)), min(lane_u32(v, 32), lane_u32(v, 48)));
}

// keys below the wave-uniform threshold t; HI32: the low word of t is zero, so the high words decide (a 32-bit compare
// per key instead of a 64-bit one)
template <int KPL, bool HI32>
__device__ __forceinline__ unsigned count_below(const double (&v)[KPL], unsigned long long t) {
    unsigned c = 0;
    const unsigned th = static_cast<unsigned>(t >> 32);
#pragma unroll
    for (int j = 0; j < KPL; ++j) {
        const bool lt = HI32 ? static_cast<unsigned>(__double2hiint(v[j])) < th
                             : static_cast<unsigned long long>(__double_as_longlong(v[j])) < t;
        c += static_cast<unsigned>(__popcll(__builtin_amdgcn_ballot_w64(lt)));
    }
    return c;
}

// the key of rank k (0-based) among the n keys of the row; n and k are the same in every lane
template <int KPL>
__device__ __forceinline__ unsigned long long reg_select(const double (&v)[KPL], int64_t n, int64_t k) {
    unsigned ah = ~0u, al = ~0u, oh = 0u, ol = 0u;
#pragma unroll
    for (int j = 0; j < KPL; ++j) {
        const unsigned hi = static_cast<unsigned>(__double2hiint(v[j])), lo = static_cast<unsigned>(__double2loint(v[j]));
        const bool valid = static_cast<int>(hi) >= 0;  // the padding pattern has bit 63 set, no key has
        ah &= hi;  // padding is all ones: neutral here
        al &= lo;
        oh |= valid ? hi : 0u;
        ol |= valid ? lo : 0u;
    }
    const unsigned long long all_and = (static_cast<unsigned long long>(wave_and_u32(ah)) << 32) | wave_and_u32(al);
    const unsigned long long all_or = (static_cast<unsigned long long>(wave_or_u32(oh)) << 32) | wave_or_u32(ol);
    const unsigned long long diff = all_and ^ all_or;
    if (diff == 0) return all_and;  // every key is the same
    int pos = 64 - __clzll(static_cast<long long>(diff));  // bits [0, pos) are undecided
    unsigned long long prefix = pos >= 64 ? 0ull : (all_and & (~0ull << pos));
    unsigned below = 0;  // keys under the current range
    unsigned in_range = static_cast<unsigned>(__builtin_amdgcn_readfirstlane(static_cast<int>(n)));
    const unsigned kk = static_cast<unsigned>(__builtin_amdgcn_readfirstlane(static_cast<int>(k)));
#pragma unroll 1
    while (pos > 0 && in_range > 1) {
        --pos;
        const unsigned long long t = prefix | (1ull << pos);
        const unsigned c1 = (pos >= 32) ? count_below<KPL, true>(v, t) : count_below<KPL, false>(v, t);
        if (kk < c1) {
            in_range = c1 - below;
        } else {
            prefix = t;
            in_range = below + in_range - c1;
            below = c1;
        }
    }
    if (pos == 0) return prefix;
    // one key left in [prefix, prefix + 2^pos): one lane holds it in one slot
    const unsigned long long mask = ~0ull << pos;
    unsigned fh = 0u, fl = 0u;
    bool mine = false;
#pragma unroll
    for (int j = 0; j < KPL; ++j) {
        const unsigned long long key = static_cast<unsigned long long>(__double_as_longlong(v[j]));
        if ((key & mask) == prefix) {
            fh = static_cast<unsigned>(key >> 32);
            fl = static_cast<unsigned>(key);
            mine = true;
        }
    }
    const int src = __builtin_ctzll(__builtin_amdgcn_ballot_w64(mine));
    return (static_cast<unsigned long long>(lane_u32(fh, src)) << 32) | lane_u32(fl, src);
}

// median the way the reference takes it from a sorted copy (mean of the two middle elements for even n)
template <int KPL>
__device__ __forceinline__ double reg_median(const double (&v)[KPL], int64_t n) {
    const int64_t k1 = (n % 2 == 0) ? n / 2 - 1 : n / 2;
    const unsigned long long v1 = reg_select<KPL>(v, n, k1);
    if (n % 2 != 0) return __longlong_as_double(static_cast<long long>(v1));
    // rank k1 + 1: v1 again when at least k1 + 2 keys are <= v1, else the smallest key above v1
    unsigned le = 0;
    unsigned ah = ~0u, al = ~0u;  // the lane's smallest key above v1
#pragma unroll
    for (int j = 0; j < KPL; ++j) {
        const unsigned long long key = static_cast<unsigned long long>(__double_as_longlong(v[j]));
        const bool gt = key > v1;
        le += static_cast<unsigned>(__popcll(__builtin_amdgcn_ballot_w64(!gt)));
        const unsigned long long cur = (static_cast<unsigned long long>(ah) << 32) | al;
        if (gt && key < cur) {
            ah = static_cast<unsigned>(key >> 32);
            al = static_cast<unsigned>(key);
        }
    }
    unsigned long long v2 = v1;
    if (static_cast<int64_t>(le) < k1 + 2) {
        const unsigned mh = wave_umin_u32(ah);
        const unsigned ml = wave_umin_u32(ah == mh ? al : ~0u);
        v2 = (static_cast<unsigned long long>(mh) << 32) | ml;
    }
    return (__longlong_as_double(static_cast<long long>(v1)) + __longlong_as_double(static_cast<long long>(v2))) / 2.0;
}

// MAD with the streaming denoiser's guards (0 when no entry is finite or every finite entry is zero);
// v is overwritten with the absolute deviations
template <int KPL>
__device__ __forceinline__ double reg_mad(double (&v)[KPL], int64_t n) {
#pragma clang fp contract(off)
    const int lane = threadIdx.x;
    bool fin = false, nz = false;
#pragma unroll
    for (int j = 0; j < KPL; ++j)
        if (static_cast<int64_t>(j) * 64 + lane < n && finite_f64(v[j])) {
            fin = true;
            if (v[j] != 0.0) nz = true;
        }
    if (__builtin_amdgcn_ballot_w64(fin) == 0 || __builtin_amdgcn_ballot_w64(nz) == 0) return 0.0;
    const double med = reg_median<KPL>(v, n);
#pragma unroll
    for (int j = 0; j < KPL; ++j) v[j] = (static_cast<int64_t>(j) * 64 + lane < n) ? fabs(v[j] - med) : reg_pad();
    return reg_median<KPL>(v, n);
}

struct MedianArgs {
    const double* d1;      // band d_1 of row r at d1 + r*row_stride
    int64_t row_stride;
    int64_t n;
    int64_t nrows;
    double sqrt_2_ln_n;
    double* thr_out;       // [nrows]
    int32_t lds_cap;       // rows up to this length are copied into LDS
};

template <int NT>
__global__ void __launch_bounds__(NT) k_universal_threshold(const MedianArgs a) {
    extern __shared__ double2 vw_smem[];
    __shared__ SelLds<NT> sel;
    const int64_t row = blockIdx.x;
    const int64_t n = a.n;
    const double* __restrict__ d = a.d1 + row * a.row_stride;
    double* stage = reinterpret_cast<double*>(vw_smem);
    double median;
    if (n <= a.lds_cap) {
        stage_row<NT, true>(stage, d, n);
        __syncthreads();
        KeySrc<true> src;
        src.v = stage;
        src.sub = 0.0;
        median = median_nt<NT>(src, n, sel);
    } else {
        KeySrc<false> src;
        src.v = d;
        src.sub = 0.0;
        median = median_nt<NT>(src, n, sel);
    }
    if (threadIdx.x == 0) {
        const double sigma = median / 0.6745;
        a.thr_out[row] = sigma * a.sqrt_2_ln_n;
    }
}

// ---------------------------------------------------------------------------
// rows of up to NT * KPL samples in the registers of one workgroup (NT = 256 or 1024 lanes)
// ---------------------------------------------------------------------------
//
// The threshold-count descent of reg_select with the counts of a step summed over the waves of the workgroup: every
// wave counts its keys below the thresholds with ballots (scalars per wave) and the counts meet in LDS.  The row is
// read from HBM exactly once (the 256-lane radix select reads a row longer than its LDS stage nine times, and its LDS
// histogram serialises on the one bin that nearly all keys of a band share in the high-order passes: 3.07 ms for
// 16384 rows of 16384 samples).
// Exchange cells: one per combine, used once, zero when the kernel starts (k_universal_threshold_wg clears them), so no
// cell is ever reset and a combine needs one barrier.  Lane 0 of every wave folds the wave's value into the cell with
// an LDS atomic (add, or, max), and after the barrier every lane reads the one result -- summing sixteen per-wave cells
// in every lane cost more instructions than the compares of a step.  AND and MIN go through OR and MAX of the
// complement.
constexpr int kWgCells = 80;  // 2 (range) + 2 * 32 (two-bit steps) + 1 (odd last bit) + 1 (last key) + 2 (second order statistic)
struct WgCells {
    unsigned long long c[kWgCells];
};

enum WgOp { kWgAdd, kWgOr, kWgAnd, kWgMin };

template <WgOp OP>
__device__ __forceinline__ void wg_fold(unsigned long long* cell, unsigned long long x) {
    if (OP == kWgAdd) atomicAdd(cell, x);
    if (OP == kWgOr) atomicOr(cell, x);
    if (OP == kWgAnd) atomicOr(cell, ~x);
    if (OP == kWgMin) atomicMax(cell, ~x);
}
template <WgOp OP>
__device__ __forceinline__ unsigned long long wg_unfold(unsigned long long r) {
    return (OP == kWgAnd || OP == kWgMin) ? ~r : r;
}

// x and y are wave-uniform; every lane of the workgroup gets the combined values
template <WgOp OPX, WgOp OPY>
__device__ __forceinline__ void wg_combine2(unsigned long long& x, unsigned long long& y, WgCells& c, int& idx) {
    if ((threadIdx.x & 63) == 0) {
        wg_fold<OPX>(&c.c[idx], x);
        wg_fold<OPY>(&c.c[idx + 1], y);
    }
    __syncthreads();
    x = wg_unfold<OPX>(c.c[idx]);
    y = wg_unfold<OPY>(c.c[idx + 1]);
    idx += 2;
}

template <WgOp OP>
__device__ __forceinline__ unsigned long long wg_combine1(unsigned long long x, WgCells& c, int& idx) {
    if ((threadIdx.x & 63) == 0) wg_fold<OP>(&c.c[idx], x);
    __syncthreads();
    const unsigned long long r = wg_unfold<OP>(c.c[idx]);
    idx += 1;
    return r;
}

// the key of rank k (0-based) among the n keys of the row held by the whole workgroup (padding = reg_pad())
template <int NW, int KPL>
__device__ __forceinline__ unsigned long long wg_select(const double (&v)[KPL], int64_t n, int64_t k, WgCells& cells, int& par) {
    unsigned ah = ~0u, al = ~0u, oh = 0u, ol = 0u;
#pragma unroll
    for (int j = 0; j < KPL; ++j) {
        const unsigned hi = static_cast<unsigned>(__double2hiint(v[j])), lo = static_cast<unsigned>(__double2loint(v[j]));
        const bool valid = static_cast<int>(hi) >= 0;
        ah &= hi;
        al &= lo;
        oh |= valid ? hi : 0u;
        ol |= valid ? lo : 0u;
    }
    unsigned long long all_and = (static_cast<unsigned long long>(wave_and_u32(ah)) << 32) | wave_and_u32(al);
    unsigned long long all_or = (static_cast<unsigned long long>(wave_or_u32(oh)) << 32) | wave_or_u32(ol);
    wg_combine2<kWgAnd, kWgOr>(all_and, all_or, cells, par);
    const unsigned long long diff = all_and ^ all_or;
    if (diff == 0) return all_and;  // every key is the same
    int pos = 64 - __clzll(static_cast<long long>(diff));  // bits [0, pos) are undecided
    unsigned long long prefix = pos >= 64 ? 0ull : (all_and & (~0ull << pos));
    unsigned below = 0;
    unsigned in_range = static_cast<unsigned>(__builtin_amdgcn_readfirstlane(static_cast<int>(n)));
    const unsigned kk = static_cast<unsigned>(__builtin_amdgcn_readfirstlane(static_cast<int>(k)));
    // two bits per step while two are left: three thresholds, their three counts through ONE exchange (the exchange and
    // its barrier over up to sixteen waves cost more than the compares, so halving the steps pays here; in the
    // single-wave descent of reg_select it does not)
#pragma unroll 1
    while (pos > 1 && in_range > 1) {
        pos -= 2;
        const unsigned long long q1 = 1ull << pos;
        const unsigned long long t1 = prefix | q1, t2 = prefix | (q1 << 1), t3 = prefix | (q1 * 3);
        unsigned w1, w2, w3;
        if (pos >= 32) {
            w1 = count_below<KPL, true>(v, t1);
            w2 = count_below<KPL, true>(v, t2);
            w3 = count_below<KPL, true>(v, t3);
        } else {
            w1 = count_below<KPL, false>(v, t1);
            w2 = count_below<KPL, false>(v, t2);
            w3 = count_below<KPL, false>(v, t3);
        }
        unsigned long long x = (static_cast<unsigned long long>(w2) << 32) | w1, y = w3;  // n <= 32768: no carry between the halves
        wg_combine2<kWgAdd, kWgAdd>(x, y, cells, par);
        const unsigned c1 = static_cast<unsigned>(x), c2 = static_cast<unsigned>(x >> 32), c3 = static_cast<unsigned>(y);
        const unsigned end = below + in_range;  // keys below the end of the current range
        if (kk < c1) {
            in_range = c1 - below;
        } else if (kk < c2) {
            prefix = t1;
            in_range = c2 - c1;
            below = c1;
        } else if (kk < c3) {
            prefix = t2;
            in_range = c3 - c2;
            below = c2;
        } else {
            prefix = t3;
            in_range = end - c3;
            below = c3;
        }
    }
#pragma unroll 1
    while (pos > 0 && in_range > 1) {
        --pos;
        const unsigned long long t = prefix | (1ull << pos);
        const unsigned cw = (pos >= 32) ? count_below<KPL, true>(v, t) : count_below<KPL, false>(v, t);
        const unsigned c1 = static_cast<unsigned>(wg_combine1<kWgAdd>(cw, cells, par));
        if (kk < c1) {
            in_range = c1 - below;
        } else {
            prefix = t;
            in_range = below + in_range - c1;
            below = c1;
        }
    }
    if (pos == 0) return prefix;
    // one key left in [prefix, prefix + 2^pos): one lane of one wave holds it
    const unsigned long long mask = ~0ull << pos;
    unsigned fh = 0u, fl = 0u;
    bool mine = false;
#pragma unroll
    for (int j = 0; j < KPL; ++j) {
        const unsigned long long key = static_cast<unsigned long long>(__double_as_longlong(v[j]));
        if ((key & mask) == prefix) {
            fh = static_cast<unsigned>(key >> 32);
            fl = static_cast<unsigned>(key);
            mine = true;
        }
    }
    const unsigned long long who = __builtin_amdgcn_ballot_w64(mine);
    unsigned long long found = ~0ull;
    if (who != 0) {
        const int src = __builtin_ctzll(who);
        found = (static_cast<unsigned long long>(lane_u32(fh, src)) << 32) | lane_u32(fl, src);
    }
    return wg_combine1<kWgMin>(found, cells, par);
}

template <int NW, int KPL>
__device__ __forceinline__ double wg_median(const double (&v)[KPL], int64_t n, WgCells& cells, int& par) {
    const int64_t k1 = (n % 2 == 0) ? n / 2 - 1 : n / 2;
    const unsigned long long v1 = wg_select<NW, KPL>(v, n, k1, cells, par);
    if (n % 2 != 0) return __longlong_as_double(static_cast<long long>(v1));
    // rank k1 + 1: v1 again when at least k1 + 2 keys are <= v1, else the smallest key above v1
    unsigned le = 0;
    unsigned ah = ~0u, al = ~0u;
#pragma unroll
    for (int j = 0; j < KPL; ++j) {
        const unsigned long long key = static_cast<unsigned long long>(__double_as_longlong(v[j]));
        const bool gt = key > v1;
        le += static_cast<unsigned>(__popcll(__builtin_amdgcn_ballot_w64(!gt)));
        const unsigned long long cur = (static_cast<unsigned long long>(ah) << 32) | al;
        if (gt && key < cur) {
            ah = static_cast<unsigned>(key >> 32);
            al = static_cast<unsigned>(key);
        }
    }
    const unsigned mh = wave_umin_u32(ah);
    const unsigned ml = wave_umin_u32(ah == mh ? al : ~0u);
    unsigned long long le_all = le, above = (static_cast<unsigned long long>(mh) << 32) | ml;  // padding counts as above
    wg_combine2<kWgAdd, kWgMin>(le_all, above, cells, par);
    const unsigned long long v2 = (static_cast<int64_t>(le_all) >= k1 + 2) ? v1 : above;
    return (__longlong_as_double(static_cast<long long>(v1)) + __longlong_as_double(static_cast<long long>(v2))) / 2.0;
}

// (two workgroups of 1024 lanes per CU need at most 64 registers per lane: the descent is bound by the latency of its
// steps, so the second row in flight counts for more than a few moves around the sixteen keys of a lane)
template <int NT, int KPL>
__global__ void __launch_bounds__(NT, (KPL <= 16 && NT == 1024) ? 8 : 1) k_universal_threshold_wg(const MedianArgs a) {
    constexpr int NW = NT / 64;
    __shared__ WgCells cells;
    int par = 0;  // next unused exchange cell
    const int tid = threadIdx.x;
    if (tid < kWgCells) cells.c[tid] = 0ull;
    __syncthreads();
    const int64_t row = blockIdx.x;
    const int64_t n = a.n;
    const double* __restrict__ d = a.d1 + row * a.row_stride;
    double v[KPL];
#pragma unroll
    for (int j = 0; j < KPL; ++j) {
        const int64_t i = static_cast<int64_t>(j) * NT + tid;
        v[j] = i < n ? fabs(d[i]) : reg_pad();
    }
    const double median = wg_median<NW, KPL>(v, n, cells, par);
    if (tid == 0) {
        const double sigma = median / 0.6745;
        a.thr_out[row] = sigma * a.sqrt_2_ln_n;
    }
}

// rows of up to 64 * KPL samples: straight from global memory into registers, one wavefront per row
template <int KPL>
__global__ void __launch_bounds__(64) k_universal_threshold_reg(const MedianArgs a) {
    const int lane = threadIdx.x;
    const int64_t row = blockIdx.x;
    const int64_t n = a.n;
    const double* __restrict__ d = a.d1 + row * a.row_stride;
    double v[KPL];
#pragma unroll
    for (int j = 0; j < KPL; ++j) {
        const int64_t i = static_cast<int64_t>(j) * 64 + lane;
        v[j] = i < n ? fabs(d[i]) : reg_pad();
    }
    const double median = reg_median<KPL>(v, n);
    if (lane == 0) {
        const double sigma = median / 0.6745;
        a.thr_out[row] = sigma * a.sqrt_2_ln_n;
    }
}

// ---------------------------------------------------------------------------
// row-resident SWT denoise: forward cascade, exact median, thresholds, inverse cascade in one launch
// ---------------------------------------------------------------------------
//
// VectorWaveSwtAdapter.denoise (swt/VectorWaveSwtAdapter.java:296-312) for rows short enough that the whole
// decomposition of a row stays on chip: one wavefront per row, HBM traffic 16 bytes per sample (the row in, the
// row out).  LDS per row: two signal buffers (running approximation, ping-pong) with a halo on either side, and
// one buffer per detail band with a right halo.  The boundary rule is materialised in the halos -- copies of the
// other end of the row (PERIODIC) or zeros (ZERO_PADDING) written after each level -- so the tap loops index
// linearly, without a modulo.  A lane owns the pairs 2*(lane + 64*k); operands of dilated taps are aligned
// 16-byte reads, the stride-one level reads its window as LT/2 + 1 aligned pairs.  The universal threshold is the
// register-resident exact selection of k_universal_threshold_reg on |d_1| read back from LDS, so d_1 is
// thresholded in place after it and the coarser bands as they are written.  No workgroup barrier exists:
// the workgroup is the wavefront, whose LDS operations execute in order.
// Needs: both decomposition filters and the low-pass reconstruction filter of exactly LT taps (LT even), the
// reference's level limit (no filter longer than the row), n <= 64 * KPL.

// WaveletDenoiser.calculateThreshold for the rules that are functions of sigma alone
// (denoising/WaveletDenoiser.java:376-427): UNIVERSAL sigma * sqrt(2 ln n); MINIMAX piecewise in n
__device__ __forceinline__ double denoiser_sigma_rule(int method, double sigma, int64_t n, double log_n) {
    if (method == 2) {
        if (n <= 32) return 0;
        if (n <= 64) return sigma * 0.3936 + 0.1829 * sigma * log_n;
        return sigma * (0.4745 + 0.1148 * log_n);
    }
    return sigma * sqrt(2.0 * log_n);
}

struct RowDenoiseArgs {
    const double* x;   // rows of length n
    int64_t x_stride;
    double* y;
    int64_t y_stride;
    int64_t n;
    int64_t nrows;
    int32_t levels;
    int32_t thr_mode;       // apply_thr mode: 1 soft, 2 hard
    double threshold;       // >= 0 (or NaN): fixed; < 0: universal threshold from |d_1|
    double sqrt_2_ln_n;
    double* thr_out;        // rule 0: [nrows]; rule 1: [nrows][levels]; or nullptr
    int32_t vec;            // rows are 16-byte aligned and n is even: 16-byte global accesses
    // rule 0: one threshold per row for every level (VectorWaveSwtAdapter.denoise; WaveletDenoiser.denoiseFixed).
    // rule 1: WaveletDenoiser's per-level thresholds (denoising/WaveletDenoiser.java:121-167, 544-586): sigma =
    // median(|d_1|) / 0.6745, divided by sqrt(2^(level-1)) when level_scaled, then the rule of `method`
    // (VW_THRESHOLD_UNIVERSAL or VW_THRESHOLD_MINIMAX); `threshold` must be negative
    int32_t rule, method, level_scaled;
    double log_n;
    int32_t taps_small;     // every scaled tap is at most 2^20 in magnitude (the host checks): see shrink_finite
    Taps dec, rec;          // decomposition (lo = h, hi = g) and reconstruction (lo = h~, hi = g~) taps, scaled
};

template <int LT, int KPL, int MODE>
__global__ void __launch_bounds__(64) k_swt_denoise_row(const RowDenoiseArgs a) {
#pragma clang fp contract(off)
    extern __shared__ double2 vw_smem[];
    constexpr int KP = KPL / 2;          // pairs per lane
    constexpr int G = KP < 2 ? KP : 2;   // pairs whose operand windows are requested together
    constexpr int NW = LT / 2 + 1;       // aligned pairs in the window of a stride-one level
    const int lane = threadIdx.x;
    const int n = static_cast<int>(a.n);
    const int J = a.levels;
    const int np = (n + 1) & ~1;
    const int hm = ((((LT - 1) << (J - 1)) + 1) & ~1) + 2;  // widest halo, even, with slack for the pair that straddles n
    const int bs = hm + np + hm;                              // signal buffer: [left halo | row | right halo]
    const int ws = np + hm;                                   // band buffer: [row | right halo]
    double* lds = reinterpret_cast<double*>(vw_smem);
    double* bufA = lds + hm;                                  // index 0 of the row
    double* bufB = lds + bs + hm;
    double* wband = lds + 2 * bs;                             // band j at wband + (j-1)*ws
    const int64_t row = blockIdx.x;
    const double* __restrict__ xrow = a.x + row * a.x_stride;

    double dl[LT], dh[LT], rl[LT], rh[LT];
#pragma unroll
    for (int l = 0; l < LT; ++l) {
        dl[l] = a.dec.lo[l];
        dh[l] = a.dec.hi[l];
        rl[l] = a.rec.lo[l];
        rh[l] = a.rec.hi[l];
    }

    // halo writers: H samples (plus one for the straddling pair) of the other end of the row, or zeros
    auto fill_left = [&](double* b, int H) {
        for (int i = lane; i < H; i += 64) b[-1 - i] = (MODE == kPeriodic) ? b[n - 1 - i] : 0.0;
    };
    auto fill_right = [&](double* b, int H) {
        for (int i = lane; i <= H; i += 64) b[n + i] = (MODE == kPeriodic) ? b[i] : 0.0;
    };
    // pair owned by this lane in chunk k; lanes beyond the row compute on its last pair and do not write
    auto own = [&](int k, bool& live) {
        const int p = 2 * (lane + 64 * k);
        live = p < n;
        return live ? p : np - 2;
    };

    // ---- the row -> bufA ------------------------------------------------------------------------------
#pragma unroll
    for (int k = 0; k < KP; ++k) {
        const int p = 2 * (lane + 64 * k);
        if (p < n) {
            double2 v;
            if (a.vec) {
                v = *reinterpret_cast<const double2*>(xrow + p);
            } else {
                v.x = xrow[p];
                v.y = (p + 1 < n) ? xrow[p + 1] : 0.0;
            }
            *reinterpret_cast<double2*>(bufA + p) = v;
        }
    }
    __syncthreads();
    fill_left(bufA, LT - 1);
    __syncthreads();

    double thr = a.threshold;
    double sigma = 0.0;
    auto level_thr = [&](int j) -> double {
        if (a.rule == 0) return thr;
        const double sj = a.level_scaled ? sigma / sqrt(static_cast<double>(1 << (j - 1))) : sigma;
        return denoiser_sigma_rule(a.method, sj, a.n, a.log_n);
    };
    double* in = bufA;
    double* out = bufB;
    // ---- forward cascade --------------------------------------------------------------------------------
    for (int j = 1; j <= J; ++j) {
        const int s = 1 << (j - 1);
        double* wj = wband + (j - 1) * ws;
        const bool thr_known = (j > 1) || !(a.threshold < 0);
        if (j > 1 && a.rule != 0) {
            thr = level_thr(j);
            if (a.thr_out != nullptr && lane == 0) a.thr_out[row * J + (j - 1)] = thr;
        }
#pragma unroll
        for (int k0 = 0; k0 < KP; k0 += G) {
            bool live[G];
            int p[G];
            double av[G][2], aw[G][2];
#pragma unroll
            for (int g = 0; g < G; ++g) p[g] = own(k0 + g, live[g]);
            if (j == 1) {
                // samples in[p - LT .. p + 1] as aligned pairs; tap l of output p reads in[p - l]
                double2 xw[G][NW];
#pragma unroll
                for (int g = 0; g < G; ++g)
#pragma unroll
                    for (int q = 0; q < NW; ++q) xw[g][q] = *reinterpret_cast<const double2*>(in + p[g] - LT + 2 * q);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    av[g][0] = av[g][1] = aw[g][0] = aw[g][1] = 0.0;
#pragma unroll
                    for (int l = 0; l < LT; ++l) {
                        // in[p - l] is sample LT - l of the window, in[p + 1 - l] sample LT + 1 - l
                        const int i0 = LT - l, i1 = LT + 1 - l;
                        const double x0 = (i0 & 1) ? xw[g][i0 / 2].y : xw[g][i0 / 2].x;
                        const double x1 = (i1 & 1) ? xw[g][i1 / 2].y : xw[g][i1 / 2].x;
                        av[g][0] = av[g][0] + x0 * dl[l];
                        av[g][1] = av[g][1] + x1 * dl[l];
                        aw[g][0] = aw[g][0] + x0 * dh[l];
                        aw[g][1] = aw[g][1] + x1 * dh[l];
                    }
                }
            } else {
                double2 xt[G][LT];
#pragma unroll
                for (int g = 0; g < G; ++g)
#pragma unroll
                    for (int l = 0; l < LT; ++l) xt[g][l] = *reinterpret_cast<const double2*>(in + p[g] - l * s);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    av[g][0] = av[g][1] = aw[g][0] = aw[g][1] = 0.0;
#pragma unroll
                    for (int l = 0; l < LT; ++l) {
                        av[g][0] = av[g][0] + xt[g][l].x * dl[l];
                        av[g][1] = av[g][1] + xt[g][l].y * dl[l];
                        aw[g][0] = aw[g][0] + xt[g][l].x * dh[l];
                        aw[g][1] = aw[g][1] + xt[g][l].y * dh[l];
                    }
                }
            }
#pragma unroll
            for (int g = 0; g < G; ++g) {
                if (thr_known) {
                    aw[g][0] = apply_thr(aw[g][0], thr, a.thr_mode);
                    aw[g][1] = apply_thr(aw[g][1], thr, a.thr_mode);
                }
                if (live[g]) {
                    *reinterpret_cast<double2*>(out + p[g]) = make_double2(av[g][0], av[g][1]);
                    *reinterpret_cast<double2*>(wj + p[g]) = make_double2(aw[g][0], aw[g][1]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
        if (j == 1 && a.threshold < 0) {
            // sigma = median(|d_1|) / 0.6745 with the order statistics a full sort would give
            // (swt/VectorWaveSwtAdapter.java:377-395); thr = sigma * sqrt(2 ln n) (:263-264)
            double v[KPL];
#pragma unroll
            for (int q = 0; q < KPL; ++q) {
                const int i = q * 64 + lane;
                v[q] = i < n ? fabs(wj[i]) : reg_pad();
            }
            const double median = reg_median<KPL>(v, a.n);
            sigma = median / 0.6745;
            thr = (a.rule == 0) ? sigma * a.sqrt_2_ln_n : level_thr(1);
            if (a.rule != 0 && a.thr_out != nullptr && lane == 0) a.thr_out[row * J] = thr;
#pragma unroll
            for (int q = 0; q < KPL; ++q) {
                const int i = q * 64 + lane;
                if (i < n) wj[i] = apply_thr(wj[i], thr, a.thr_mode);
            }
            __syncthreads();
        }
        // halos: the band's right halo for its inverse level; the approximation's left halo for the next forward
        // level, or its right halo for the first inverse level
        fill_right(wj, (LT - 1) * s);
        if (j < J)
            fill_left(out, (LT - 1) * 2 * s);
        else
            fill_right(out, (LT - 1) * s);
        __syncthreads();
        double* t = in;
        in = out;
        out = t;
    }
    if (a.rule == 0 && a.thr_out != nullptr && lane == 0) a.thr_out[row] = thr;

    // ---- inverse cascade: `in` holds R_J ------------------------------------------------------------------
    double* __restrict__ yrow = a.y + row * a.y_stride;
    for (int j = J; j >= 1; --j) {
        const int s = 1 << (j - 1);
        const double* wj = wband + (j - 1) * ws;
#pragma unroll
        for (int k0 = 0; k0 < KP; k0 += G) {
            bool live[G];
            int p[G];
            double r[G][2];
#pragma unroll
            for (int g = 0; g < G; ++g) p[g] = own(k0 + g, live[g]);
            if (j == 1) {
                double2 rw[G][NW], ww[G][NW];
#pragma unroll
                for (int g = 0; g < G; ++g)
#pragma unroll
                    for (int q = 0; q < NW; ++q) {
                        rw[g][q] = *reinterpret_cast<const double2*>(in + p[g] + 2 * q);
                        ww[g][q] = *reinterpret_cast<const double2*>(wj + p[g] + 2 * q);
                    }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    r[g][0] = r[g][1] = 0.0;
#pragma unroll
                    for (int l = 0; l < LT; ++l) {
                        const double a0 = (l & 1) ? rw[g][l / 2].y : rw[g][l / 2].x;
                        const double a1 = (l & 1) ? rw[g][l / 2 + 1].x : rw[g][l / 2].y;
                        const double d0 = (l & 1) ? ww[g][l / 2].y : ww[g][l / 2].x;
                        const double d1 = (l & 1) ? ww[g][l / 2 + 1].x : ww[g][l / 2].y;
                        r[g][0] = r[g][0] + (rl[l] * a0 + rh[l] * d0);
                        r[g][1] = r[g][1] + (rl[l] * a1 + rh[l] * d1);
                    }
                }
            } else {
                double2 rt[G][LT], wt[G][LT];
#pragma unroll
                for (int g = 0; g < G; ++g)
#pragma unroll
                    for (int l = 0; l < LT; ++l) {
                        rt[g][l] = *reinterpret_cast<const double2*>(in + p[g] + l * s);
                        wt[g][l] = *reinterpret_cast<const double2*>(wj + p[g] + l * s);
                    }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    r[g][0] = r[g][1] = 0.0;
#pragma unroll
                    for (int l = 0; l < LT; ++l) {
                        r[g][0] = r[g][0] + (rl[l] * rt[g][l].x + rh[l] * wt[g][l].x);
                        r[g][1] = r[g][1] + (rl[l] * rt[g][l].y + rh[l] * wt[g][l].y);
                    }
                }
            }
#pragma unroll
            for (int g = 0; g < G; ++g) {
                if (live[g]) {
                    if (j == 1) {
                        if (a.vec) {
                            *reinterpret_cast<double2*>(yrow + p[g]) = make_double2(r[g][0], r[g][1]);
                        } else {
                            yrow[p[g]] = r[g][0];
                            if (p[g] + 1 < n) yrow[p[g] + 1] = r[g][1];
                        }
                    } else {
                        *reinterpret_cast<double2*>(out + p[g]) = make_double2(r[g][0], r[g][1]);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (j > 1) {
            __syncthreads();
            fill_right(out, (LT - 1) * (s >> 1));
            __syncthreads();
            double* t = in;
            in = out;
            out = t;
        }
    }
}

// The same for rows of at most 512 samples and at most four levels (the streaming-block shape): the detail bands
// stay in REGISTERS between the forward and the inverse cascade (4 levels x 4 pairs per lane = 64 registers) and
// pass through one LDS band buffer only while their inverse level runs; the running approximation is updated in
// place (the forward looks left, so its chunks are taken right to left; the inverse the other way round).  LDS per
// row drops from about 26 KiB to 9 KiB, i.e. from 6 to 12 resident rows per CU (then bounded by registers).
template <int LT, int MODE>
__global__ void __launch_bounds__(64, 3) k_swt_denoise_rowreg(const RowDenoiseArgs a) {
#pragma clang fp contract(off)
    extern __shared__ double2 vw_smem[];
    constexpr int KP = 4, JMAX = 4;
    constexpr int GF = (LT > 4) ? 1 : 2;  // forward chunks whose operand windows are requested together (registers)
    constexpr int NW = LT / 2 + 1;
    const int lane = threadIdx.x;
    const int n = static_cast<int>(a.n);
    const int J = a.levels;
    const int np = (n + 1) & ~1;
    const int hm = ((((LT - 1) << (J - 1)) + 1) & ~1) + 2;
    double* lds = reinterpret_cast<double*>(vw_smem);
    double* sig = lds + hm;               // [left halo | row | right halo]
    double* wb = lds + hm + np + hm;      // [row | right halo]
    const int64_t row = blockIdx.x;
    const double* __restrict__ xrow = a.x + row * a.x_stride;

    double dl[LT], dh[LT];  // (the reconstruction taps are fetched where the inverse cascade starts: scalar registers)
#pragma unroll
    for (int l = 0; l < LT; ++l) {
        dl[l] = a.dec.lo[l];
        dh[l] = a.dec.hi[l];
    }
    auto fill_left = [&](double* b, int H) {
        for (int i = lane; i < H; i += 64) b[-1 - i] = (MODE == kPeriodic) ? b[n - 1 - i] : 0.0;
    };
    auto fill_right = [&](double* b, int H) {
        for (int i = lane; i <= H; i += 64) b[n + i] = (MODE == kPeriodic) ? b[i] : 0.0;
    };
    int p[KP];
    bool live[KP];
#pragma unroll
    for (int k = 0; k < KP; ++k) {
        const int q = 2 * (lane + 64 * k);
        live[k] = q < n;
        p[k] = live[k] ? q : np - 2;  // lanes beyond the row compute on its last pair and do not write
    }

    bool tame = true;
#pragma unroll
    for (int k = 0; k < KP; ++k) {
        if (live[k]) {
            double2 v;
            if (a.vec) {
                v = *reinterpret_cast<const double2*>(xrow + p[k]);
            } else {
                v.x = xrow[p[k]];
                v.y = (p[k] + 1 < n) ? xrow[p[k] + 1] : 0.0;
            }
            *reinterpret_cast<double2*>(sig + p[k]) = v;
            tame = tame && fabs(v.x) < 0x1p500 && fabs(v.y) < 0x1p500;
        }
    }
    __syncthreads();
    fill_left(sig, LT - 1);
    __syncthreads();
    // Samples below 2^500 and taps of at most 2^20: eight levels of sums of at most sixteen products stay below
    // 2^(500 + 8 * 24), so every coefficient of this row is finite and shrink_finite applies whenever the threshold is
    // finite and not negative (a NaN sample fails the comparison above).
    const bool calm = a.taps_small != 0 && __builtin_amdgcn_ballot_w64(!tame) == 0;
    const bool soft = a.thr_mode != 2;

    double thr = a.threshold;
    double sigma = 0.0;
    auto level_thr = [&](int j) -> double {
        if (a.rule == 0) return thr;
        const double sj = a.level_scaled ? sigma / sqrt(static_cast<double>(1 << (j - 1))) : sigma;
        return denoiser_sigma_rule(a.method, sj, a.n, a.log_n);
    };
    // wave-uniform: the current threshold allows the short form
    auto quick_now = [&]() -> bool {
        return __builtin_amdgcn_readfirstlane(static_cast<int>(calm && thr >= 0.0 && thr <= 0x1.fffffffffffffp1023)) != 0;
    };
    bool quick = quick_now();
    // `quick` is wave-uniform: the general rule stays behind a scalar branch (the empty asm keeps the compiler from
    // evaluating both forms for every coefficient and selecting afterwards)
    auto shrink = [&](double c) -> double {
        if (quick) return shrink_finite(c, thr, soft);
        asm volatile("" ::: "memory");
        return apply_thr(c, thr, a.thr_mode);
    };
    double wreg[JMAX][KP][2];
    // ---- forward cascade, in place, chunk groups right to left ---------------------------------------------
#pragma unroll
    for (int j = 1; j <= JMAX; ++j) {
        if (j <= J) {
            const int s = 1 << (j - 1);
            const bool thr_known = (j > 1) || !(a.threshold < 0);
            if (j > 1 && a.rule != 0) {
                thr = level_thr(j);
                quick = quick_now();
                if (a.thr_out != nullptr && lane == 0) a.thr_out[row * J + (j - 1)] = thr;
            }
#pragma unroll
            for (int k0 = KP - GF; k0 >= 0; k0 -= GF) {
                double av[GF][2];
                if (j == 1) {
                    double2 xw[GF][NW];
#pragma unroll
                    for (int g = 0; g < GF; ++g)
#pragma unroll
                        for (int q = 0; q < NW; ++q) xw[g][q] = *reinterpret_cast<const double2*>(sig + p[k0 + g] - LT + 2 * q);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int g = 0; g < GF; ++g) {
                        double w0 = 0.0, w1 = 0.0;
                        av[g][0] = av[g][1] = 0.0;
#pragma unroll
                        for (int l = 0; l < LT; ++l) {
                            const int i0 = LT - l, i1 = LT + 1 - l;
                            const double x0 = (i0 & 1) ? xw[g][i0 / 2].y : xw[g][i0 / 2].x;
                            const double x1 = (i1 & 1) ? xw[g][i1 / 2].y : xw[g][i1 / 2].x;
                            av[g][0] = av[g][0] + x0 * dl[l];
                            av[g][1] = av[g][1] + x1 * dl[l];
                            w0 = w0 + x0 * dh[l];
                            w1 = w1 + x1 * dh[l];
                        }
                        wreg[j - 1][k0 + g][0] = w0;
                        wreg[j - 1][k0 + g][1] = w1;
                    }
                } else {
                    double2 xt[GF][LT];
#pragma unroll
                    for (int g = 0; g < GF; ++g)
#pragma unroll
                        for (int l = 0; l < LT; ++l) xt[g][l] = *reinterpret_cast<const double2*>(sig + p[k0 + g] - l * s);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int g = 0; g < GF; ++g) {
                        double w0 = 0.0, w1 = 0.0;
                        av[g][0] = av[g][1] = 0.0;
#pragma unroll
                        for (int l = 0; l < LT; ++l) {
                            av[g][0] = av[g][0] + xt[g][l].x * dl[l];
                            av[g][1] = av[g][1] + xt[g][l].y * dl[l];
                            w0 = w0 + xt[g][l].x * dh[l];
                            w1 = w1 + xt[g][l].y * dh[l];
                        }
                        wreg[j - 1][k0 + g][0] = thr_known ? shrink(w0) : w0;
                        wreg[j - 1][k0 + g][1] = thr_known ? shrink(w1) : w1;
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int g = 0; g < GF; ++g)
                    if (live[k0 + g]) *reinterpret_cast<double2*>(sig + p[k0 + g]) = make_double2(av[g][0], av[g][1]);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (j == 1) {
                if (a.threshold < 0) {
                    double v[2 * KP];
#pragma unroll
                    for (int k = 0; k < KP; ++k) {
                        const int q = 2 * (lane + 64 * k);
                        v[2 * k] = q < n ? fabs(wreg[0][k][0]) : reg_pad();
                        v[2 * k + 1] = q + 1 < n ? fabs(wreg[0][k][1]) : reg_pad();
                    }
                    const double median = reg_median<2 * KP>(v, a.n);
                    sigma = median / 0.6745;
                    thr = (a.rule == 0) ? sigma * a.sqrt_2_ln_n : level_thr(1);
                    quick = quick_now();
                    if (a.rule != 0 && a.thr_out != nullptr && lane == 0) a.thr_out[row * J] = thr;
                }
#pragma unroll
                for (int k = 0; k < KP; ++k) {
                    wreg[0][k][0] = shrink(wreg[0][k][0]);
                    wreg[0][k][1] = shrink(wreg[0][k][1]);
                }
            }
            __syncthreads();
            if (j < J)
                fill_left(sig, (LT - 1) * 2 * s);
            else
                fill_right(sig, (LT - 1) * s);
            __syncthreads();
        }
    }
    if (a.rule == 0 && a.thr_out != nullptr && lane == 0) a.thr_out[row] = thr;

    // ---- inverse cascade, in place, chunks left to right ------------------------------------------------------
    double* __restrict__ yrow = a.y + row * a.y_stride;
    double rl[LT], rh[LT];
#pragma unroll
    for (int l = 0; l < LT; ++l) {
        rl[l] = a.rec.lo[l];
        rh[l] = a.rec.hi[l];
    }
#pragma unroll
    for (int j = JMAX; j >= 1; --j) {
        if (j <= J) {
            const int s = 1 << (j - 1);
#pragma unroll
            for (int k = 0; k < KP; ++k)
                if (live[k]) *reinterpret_cast<double2*>(wb + p[k]) = make_double2(wreg[j - 1][k][0], wreg[j - 1][k][1]);
            __syncthreads();
            fill_right(wb, (LT - 1) * s);
            __syncthreads();
#pragma unroll
            for (int k = 0; k < KP; ++k) {
                double r0 = 0.0, r1 = 0.0;
                if (j == 1) {
                    double2 rw[NW], ww[NW];
#pragma unroll
                    for (int q = 0; q < NW; ++q) {
                        rw[q] = *reinterpret_cast<const double2*>(sig + p[k] + 2 * q);
                        ww[q] = *reinterpret_cast<const double2*>(wb + p[k] + 2 * q);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int l = 0; l < LT; ++l) {
                        const double a0 = (l & 1) ? rw[l / 2].y : rw[l / 2].x;
                        const double a1 = (l & 1) ? rw[l / 2 + 1].x : rw[l / 2].y;
                        const double d0 = (l & 1) ? ww[l / 2].y : ww[l / 2].x;
                        const double d1 = (l & 1) ? ww[l / 2 + 1].x : ww[l / 2].y;
                        r0 = r0 + (rl[l] * a0 + rh[l] * d0);
                        r1 = r1 + (rl[l] * a1 + rh[l] * d1);
                    }
                } else {
                    // the window in two halves: with all four bands still in registers, sixteen operand pairs at once
                    // would not fit the 128 registers that four waves per SIMD leave a lane
                    constexpr int LH = (LT + 1) / 2;
#pragma unroll
                    for (int h0 = 0; h0 < LT; h0 += LH) {
                        double2 rt[LH], wt[LH];
#pragma unroll
                        for (int l = 0; l < LH; ++l) {
                            if (h0 + l < LT) {
                                rt[l] = *reinterpret_cast<const double2*>(sig + p[k] + (h0 + l) * s);
                                wt[l] = *reinterpret_cast<const double2*>(wb + p[k] + (h0 + l) * s);
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int l = 0; l < LH; ++l) {
                            if (h0 + l < LT) {
                                r0 = r0 + (rl[h0 + l] * rt[l].x + rh[h0 + l] * wt[l].x);
                                r1 = r1 + (rl[h0 + l] * rt[l].y + rh[h0 + l] * wt[l].y);
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                if (live[k]) {
                    if (j == 1) {
                        if (a.vec) {
                            *reinterpret_cast<double2*>(yrow + p[k]) = make_double2(r0, r1);
                        } else {
                            yrow[p[k]] = r0;
                            if (p[k] + 1 < n) yrow[p[k] + 1] = r1;
                        }
                    } else {
                        *reinterpret_cast<double2*>(sig + p[k]) = make_double2(r0, r1);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (j > 1) {
                __syncthreads();
                fill_right(sig, (LT - 1) * (s >> 1));
                __syncthreads();
            }
        }
    }
}

// ---------------------------------------------------------------------------
// streaming denoiser: noise window, noise level and multiplier-path threshold, one workgroup per stream
// ---------------------------------------------------------------------------
//
// Every stream keeps a ring of |detail| samples.  A block writes all of its details into the ring when
// there are at most `w` of them, else exactly `w` picked stratum by stratum (up to ten equal strata, evenly
// spaced inside each), so the ring position after a block is the same for every stream.  The noise level is
// MAD(ring)/0.6745 with MAD = median(|v - median(v)|), or the sample standard deviation summed in index
// order.  With a threshold multiplier the block's threshold comes from that level (or, when there is none
// yet, from the MAD of the block's own |details|).

struct StreamNoiseArgs {
    const double* details;  // band d_1 of stream r at details + r*row_stride
    int64_t row_stride;
    int64_t n;
    int64_t nrows;
    double* window;         // [nrows][w]
    int64_t w;
    int64_t index0;         // ring position before this block
    int32_t estimation;     // 0 MAD, 1 STD, 2 FIXED: selects the estimator and the sigma rule of the threshold
    int32_t do_update;      // 1: the block's picks enter the ring
    int32_t do_level;       // 1: the level is re-estimated here (MAD; STD when no other kernel does it)
    int32_t want_thr;       // 1: write thr = base threshold * multiplier
    int32_t method;         // 0 UNIVERSAL, 1 SURE (0.8 * universal), 2 MINIMAX, 4 FIXED (sigma itself)
    double multiplier;
    double log_n;           // Math.log(n)
    double sqrt_2_ln_n;     // Math.sqrt(2.0 * Math.log(n))
    double* level;          // [nrows] in/out
    double* thr;            // [nrows]
    int32_t lds_cap;        // a ring or a block up to this length is copied into LDS
};

// threshold of the multiplier path before the multiplier: UNIVERSAL, SURE (0.8 of it), MINIMAX (piecewise in
// n), FIXED (sigma itself)
__device__ __forceinline__ double stream_base_threshold(const StreamNoiseArgs& a, double sigma) {
#pragma clang fp contract(off)
    if (a.method == 0) return sigma * a.sqrt_2_ln_n;
    if (a.method == 1) return sigma * a.sqrt_2_ln_n * 0.8;
    if (a.method == 2) {
        if (a.n <= 32) return 0.0;
        if (a.n <= 64) return sigma * (0.3936 + 0.1829 * a.log_n);
        return sigma * (0.4745 + 0.1148 * a.log_n);
    }
    return sigma;
}

// MAD of |v_0..n-1| with the reference's guards: 0 when no entry is finite or every finite entry is zero
template <int NT, typename Src>
__device__ double stream_mad(Src& src, int64_t n, SelLds<NT>& sel) {
    int fin = 0, nz = 0;
    for (int64_t i = threadIdx.x; i < n; i += NT) {
        const double v = src.raw(i);
        if (finite_f64(v)) {
            fin = 1;
            if (v != 0.0) nz = 1;
        }
    }
    const int any_fin = __syncthreads_or(fin);
    const int any_nz = __syncthreads_or(nz);
    if (!any_fin || !any_nz) return 0.0;
    src.sub = 0.0;
    const double med = median_nt<NT>(src, n, sel);
    src.sub = med;
    return median_nt<NT>(src, n, sel);
}

// Sample standard deviation of v_0..w-1, summed in index order by the calling lane.  The loads of a group of
// 16 are issued together so that the adder chain, not the memory latency, sets the pace.
__device__ double sequential_std(const double* v, int64_t w) {
#pragma clang fp contract(off)
    double sum = 0.0;
    int64_t i = 0;
    for (; i + 16 <= w; i += 16) {
        double t[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) t[j] = v[i + j];
#pragma unroll
        for (int j = 0; j < 16; ++j) sum += t[j];
    }
    for (; i < w; ++i) sum += v[i];
    const double mean = sum / static_cast<double>(w);
    double ssd = 0.0;
    i = 0;
    for (; i + 16 <= w; i += 16) {
        double t[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const double diff = v[i + j] - mean;
            t[j] = diff * diff;
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) ssd += t[j];
    }
    for (; i < w; ++i) {
        const double diff = v[i] - mean;
        ssd += diff * diff;
    }
    return sqrt(ssd / static_cast<double>(w - 1));
}

// noise level of a ring: MAD / 0.6745 (estimation 0) or the standard deviation (1); every lane gets it
template <int NT, typename Src>
__device__ double ring_level(Src& src, int64_t w, int estimation, SelLds<NT>& sel, double* bc) {
    if (estimation == 0) return stream_mad<NT>(src, w, sel) / 0.6745;
    unsigned long long fin = 0;
    for (int64_t i = threadIdx.x; i < w; i += NT) fin += finite_f64(src.raw(i)) ? 1 : 0;
    fin = sum_nt<NT>(fin, sel);
    if (threadIdx.x == 0) *bc = fin >= 2 ? sequential_std(src.v, w) : 0.0;  // fewer than two finite entries: 0
    __syncthreads();
    const double r = *bc;
    __syncthreads();
    return r;
}

// index of the k-th |detail| a block of n > w samples contributes to a ring of w: strata 0..extra-1 give
// per+1 picks each, the rest per; picks are evenly spaced inside their stratum
__device__ __forceinline__ unsigned stratified_pick(unsigned k, unsigned n, unsigned w) {
    const unsigned strata = w < 10u ? w : 10u;
    const unsigned per = w / strata, extra = w % strata, size = n / strata;
    unsigned s, i;
    if (k < extra * (per + 1u)) {
        s = k / (per + 1u);
        i = k - s * (per + 1u);
    } else {
        const unsigned kk = k - extra * (per + 1u);
        s = extra + kk / per;
        i = kk - (kk / per) * per;
    }
    const unsigned take = per + (s < extra ? 1u : 0u);
    const unsigned start = s * size;
    const unsigned end = (s == strata - 1u) ? n : (s + 1u) * size;
    const unsigned len = end - start;
    const unsigned step = len / take > 1u ? len / take : 1u;
    return start + (i * step) % len;
}

// 256 lanes per stream: rings and blocks too long for the register-resident kernels below, and the ring update
// of the STD estimator.  n and w are below 2^31 (checked by the host).
template <int NT>
__global__ void __launch_bounds__(NT) k_stream_noise(const StreamNoiseArgs a) {
#pragma clang fp contract(off)
    extern __shared__ double2 vw_smem[];
    __shared__ SelLds<NT> sel;
    __shared__ double bc;
    const int tid = threadIdx.x;
    const int64_t row = blockIdx.x;
    const int64_t n = a.n, w = a.w;
    const double* __restrict__ d = a.details + row * a.row_stride;
    double* win = a.window + row * w;
    double* stage = reinterpret_cast<double*>(vw_smem);
    const bool ring_in_lds = a.do_level && w <= a.lds_cap;
    double lvl;

    if (a.do_update) {
        // the old ring goes to LDS while the block's picks are fetched; the picks then patch both copies
        if (ring_in_lds) stage_row<NT, false>(stage, win, w);
        const unsigned m = static_cast<unsigned>(n <= w ? n : w);
        const unsigned un = static_cast<unsigned>(n), uw = static_cast<unsigned>(w), i0 = static_cast<unsigned>(a.index0);
        if (ring_in_lds) __syncthreads();
        for (unsigned k0 = 0; k0 < m; k0 += 4 * NT) {
            double t[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned k = k0 + j * NT + tid;
                if (k < m) t[j] = fabs(d[un > uw ? stratified_pick(k, un, uw) : k]);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned k = k0 + j * NT + tid;
                if (k < m) {
                    unsigned slot = i0 + k;  // both below w
                    if (slot >= uw) slot -= uw;
                    win[slot] = t[j];
                    if (ring_in_lds) stage[slot] = t[j];
                }
            }
        }
        if (!a.do_level) return;  // the level comes from another kernel
        __threadfence_block();
        __syncthreads();
        if (ring_in_lds) {
            KeySrc<true> src;
            src.v = stage;
            src.sub = 0.0;
            lvl = ring_level<NT>(src, w, a.estimation, sel, &bc);
        } else {
            KeySrc<false> src;
            src.v = win;
            src.sub = 0.0;
            lvl = ring_level<NT>(src, w, a.estimation, sel, &bc);
        }
        if (tid == 0) a.level[row] = lvl;
    } else {
        lvl = a.level[row];
    }
    if (!a.want_thr) return;

    double sigma = lvl;
    if (sigma <= 0.0 || a.estimation == 2) {
        __syncthreads();  // the ring's LDS copy is dead from here on
        if (n <= a.lds_cap) {
            stage_row<NT, true>(stage, d, n);
            __syncthreads();
            KeySrc<true> src;
            src.v = stage;
            src.sub = 0.0;
            sigma = stream_mad<NT>(src, n, sel) / 0.6745;
        } else {
            // signed details: the guards see the same finite / non-zero pattern and the keys take |v|
            KeySrc<false> src;
            src.v = d;
            src.sub = 0.0;
            sigma = stream_mad<NT>(src, n, sel) / 0.6745;
        }
    }
    if (tid == 0) a.thr[row] = stream_base_threshold(a, sigma) * a.multiplier;
}

// The same for rings and blocks of up to 64 * KPL samples, straight from global memory into the registers of
// one wavefront per stream.  Two kernels (ring + MAD level; threshold of the multiplier path) so that each
// carries one copy of the selection code.
template <int KPL>
__global__ void __launch_bounds__(64) k_stream_ring_mad_reg(const StreamNoiseArgs a) {
#pragma clang fp contract(off)
    const int lane = threadIdx.x;
    const int64_t row = blockIdx.x;
    const int64_t n = a.n, w = a.w;
    const double* __restrict__ d = a.details + row * a.row_stride;
    double* win = a.window + row * w;
    double v[KPL];
    const unsigned m = static_cast<unsigned>(n <= w ? n : w);
    const unsigned un = static_cast<unsigned>(n), uw = static_cast<unsigned>(w), i0 = static_cast<unsigned>(a.index0);
    if (un > uw) {
        // a block longer than the ring replaces every slot: write the picks, then read the ring back
        for (unsigned k = lane; k < m; k += 64u) {
            unsigned slot = i0 + k;
            if (slot >= uw) slot -= uw;
            win[slot] = fabs(d[stratified_pick(k, un, uw)]);
        }
        __threadfence_block();
        __syncthreads();
#pragma unroll
        for (int j = 0; j < KPL; ++j) {
            const unsigned i = static_cast<unsigned>(j) * 64u + lane;
            v[j] = i < uw ? win[i] : reg_pad();
        }
    } else {
#pragma unroll
        for (int j = 0; j < KPL; ++j) {
            const unsigned i = static_cast<unsigned>(j) * 64u + lane;
            v[j] = reg_pad();
            if (i < uw) {
                const unsigned rel = i >= i0 ? i - i0 : i + uw - i0;  // position of slot i in this block's run of picks
                const double* src = rel < m ? d + rel : win + i;     // a new pick, or the slot's old value (>= 0)
                v[j] = fabs(*src);
                if (rel < m) win[i] = v[j];
            }
        }
    }
    const double lvl = reg_mad<KPL>(v, w) / 0.6745;
    if (lane == 0) a.level[row] = lvl;
}

template <int KPL>
__global__ void __launch_bounds__(64) k_stream_threshold_reg(const StreamNoiseArgs a) {
#pragma clang fp contract(off)
    const int lane = threadIdx.x;
    const int64_t row = blockIdx.x;
    const int64_t n = a.n;
    const double* __restrict__ d = a.details + row * a.row_stride;
    double sigma = a.level[row];
    if (sigma <= 0.0 || a.estimation == 2) {
        // no level to use: the block's own |details| give sigma
        double v[KPL];
#pragma unroll
        for (int j = 0; j < KPL; ++j) {
            const int64_t i = static_cast<int64_t>(j) * 64 + lane;
            v[j] = i < n ? fabs(d[i]) : reg_pad();
        }
        sigma = reg_mad<KPL>(v, n) / 0.6745;
    }
    if (lane == 0) a.thr[row] = stream_base_threshold(a, sigma) * a.multiplier;
}

// ---------------------------------------------------------------------------
// WaveletDenoiser threshold selection: one workgroup per (row, level)
// ---------------------------------------------------------------------------
//
// UNIVERSAL and MINIMAX are closed forms of the row's sigma.  BAYES needs the mean and variance of the
// band summed in index order (one lane adds, the others stage the band through LDS so that the adder
// never waits for HBM).  SURE evaluates the risk of every candidate threshold |c_k| -- O(n^2) like the
// reference -- one candidate per lane, each risk summed in index order, then the smallest risk (ties:
// the smallest threshold, which is what scanning the sorted candidates for the first minimum yields).

struct DenoiserThrArgs {
    const double* coeffs;  // detail band j of row r at coeffs + r*row_stride + (j-1)*band_stride
    int64_t row_stride, band_stride;
    int64_t n;
    int64_t nrows;
    int32_t levels;
    int32_t method;        // 0 UNIVERSAL, 1 SURE, 2 MINIMAX, 3 BAYES
    int32_t level_scaled;  // 1: sigma_j = sigma / sqrt(1 << (j-1)) (multi-level form)
    double log_n;          // Math.log(n), computed by the host
    const double* sigma;   // [nrows] MAD estimate from the finest band
    double* thr;           // [nrows][levels]
};

constexpr int kThrChunk = 4096;  // doubles of a band staged in LDS at a time

// UNIVERSAL and MINIMAX need nothing but sigma: one lane per (row, level)
__global__ void __launch_bounds__(kThreads) k_denoiser_sigma_rules(const DenoiserThrArgs a) {
    const int64_t id = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
    if (id >= a.nrows * a.levels) return;
    const int64_t row = id / a.levels;
    const int level = static_cast<int>(id - row * a.levels) + 1;
    double sigma = a.sigma[row];
    if (a.level_scaled) sigma = sigma / sqrt(static_cast<double>(1 << (level - 1)));
    a.thr[id] = denoiser_sigma_rule(a.method, sigma, a.n, a.log_n);
}

__global__ void __launch_bounds__(kThreads) k_denoiser_thresholds(const DenoiserThrArgs a) {
    __shared__ double stage[kThrChunk];
    __shared__ double red_r[kThreads];
    __shared__ double red_t[kThreads];
    __shared__ double bc[2];
    const int tid = threadIdx.x;
    const int64_t row = blockIdx.x / a.levels;
    const int level = static_cast<int>(blockIdx.x - row * a.levels) + 1;
    const int64_t n = a.n;
    const double* __restrict__ c = a.coeffs + row * a.row_stride + static_cast<int64_t>(level - 1) * a.band_stride;
    double sigma = a.sigma[row];
    if (a.level_scaled) sigma = sigma / sqrt(static_cast<double>(1 << (level - 1)));
    double* out = a.thr + row * a.levels + (level - 1);
    const double universal = denoiser_sigma_rule(0, sigma, n, a.log_n);

    if (a.method == 0 || a.method == 2) {
        if (tid == 0) *out = denoiser_sigma_rule(a.method, sigma, n, a.log_n);
        return;
    }
    if (a.method == 3) {
        // pass 0: mean; pass 1: variance around it -- each a running sum in index order
        double mean = 0.0, variance = 0.0;
        for (int pass = 0; pass < 2; ++pass) {
            double acc = 0.0;
            for (int64_t base = 0; base < n; base += kThrChunk) {
                const int cnt = static_cast<int>((n - base) < kThrChunk ? (n - base) : kThrChunk);
                for (int i = tid; i < cnt; i += kThreads) stage[i] = c[base + i];
                __syncthreads();
                if (tid == 0) {
                    if (pass == 0) {
                        for (int i = 0; i < cnt; ++i) acc += stage[i];
                    } else {
                        for (int i = 0; i < cnt; ++i) {
                            const double diff = stage[i] - mean;
                            acc += diff * diff;
                        }
                    }
                }
                __syncthreads();
            }
            if (tid == 0) {
                if (pass == 0) {
                    mean = acc / static_cast<double>(n);
                    bc[0] = mean;
                } else {
                    variance = acc / static_cast<double>(n);
                }
            }
            __syncthreads();
            mean = bc[0];
        }
        if (tid == 0) {
            const double sigma2 = sigma * sigma;
            const double d = variance - sigma2;
            const double sx2 = d > 0.0 ? d : 0.0;
            const double sx = sqrt(sx2 + 1e-10);
            *out = sigma2 / sx;
        }
        return;
    }
    // SURE
    const double sigma2 = sigma * sigma;
    double best_risk = __builtin_inf(), best_t = 0.0;
    const bool in_lds = n <= kThrChunk;
    if (in_lds) {
        for (int i = tid; i < n; i += kThreads) stage[i] = c[i];
        __syncthreads();
    }
    for (int64_t k = tid; k < n; k += kThreads) {
        const double t = fabs(in_lds ? stage[k] : c[k]);
        double risk = -static_cast<double>(n) * sigma2;
        for (int64_t i = 0; i < n; ++i) {
            const double v = in_lds ? stage[i] : c[i];
            const double av = fabs(v);
            if (av <= t) {
                risk += v * v;
            } else {
                risk += sigma2 + (av - t) * (av - t);
            }
        }
        risk = risk / static_cast<double>(n);
        if (risk < best_risk || (risk == best_risk && t < best_t)) {
            best_risk = risk;
            best_t = t;
        }
    }
    red_r[tid] = best_risk;
    red_t[tid] = best_t;
    __syncthreads();
    for (int w = kThreads / 2; w > 0; w >>= 1) {
        if (tid < w) {
            const double r2 = red_r[tid + w], t2 = red_t[tid + w];
            if (r2 < red_r[tid] || (r2 == red_r[tid] && t2 < red_t[tid])) {
                red_r[tid] = r2;
                red_t[tid] = t2;
            }
        }
        __syncthreads();
    }
    if (tid == 0) {
        double best = red_t[0];
        // the reference starts from minRisk = +inf, bestThreshold = 0: a band whose risks are all NaN or
        // +inf keeps 0
        if (!(red_r[0] < __builtin_inf())) best = 0.0;
        if (best > universal) best = universal;
        *out = best;
    }
}


// Mean and sum of squared deviations of 16 rows at once, one row per lane 0..15 of a single-wave workgroup.
// Both must be running sums in index order (that is what the Java loops round to), so a row is one dependent
// chain of adds; the parallelism is across rows.  The wave fetches the 16 rows' next 256 elements with 64
// coalesced 512-byte loads that stay in flight while the lanes add the previous 256 out of LDS (row stride
// 257 doubles: conflict-free for the 16 adding lanes).  A lane that walked one chain alone per wave would
// keep the FP64 pipe busy for a full wavefront per add, and a wave that waited for every load before adding
// would spend its time on memory latency.
//   my_ptr: address of this lane's row (lanes >= nq hold none); mean = (sum v_i) / n; ssd = sum (v_i - mean)^2;
//   nfinite = finite entries of the row (COUNT only).
constexpr int kChainRows = 16;
constexpr int kChainCols = 256;
constexpr int kChainStride = kChainCols + 1;

template <bool COUNT>
__device__ __forceinline__ void lane_chains(unsigned long long my_ptr, bool have, int nq, int64_t n,
                                            double* tile /*kChainRows * kChainStride*/, double& mean, double& ssd,
                                            unsigned& nfinite) {
#pragma clang fp contract(off)
    const int lane = threadIdx.x;
    const double* rowp[kChainRows];
#pragma unroll
    for (int r = 0; r < kChainRows; ++r) {
        const unsigned lo = __builtin_amdgcn_readlane(static_cast<int>(my_ptr & 0xffffffffu), r);
        const unsigned hi = __builtin_amdgcn_readlane(static_cast<int>(my_ptr >> 32), r);
        rowp[r] = reinterpret_cast<const double*>((static_cast<unsigned long long>(hi) << 32) | lo);
    }
    mean = 0.0;
    ssd = 0.0;
    nfinite = 0;
    double v[kChainRows * 4];
    auto fetch = [&](int64_t base) {
#pragma unroll
        for (int r = 0; r < kChainRows; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int64_t i = base + c * 64 + lane;
                v[r * 4 + c] = (r < nq && i < n) ? rowp[r][i] : 0.0;
            }
    };
    for (int pass = 0; pass < 2; ++pass) {
        double acc = 0.0;
        fetch(0);
        for (int64_t base = 0; base < n; base += kChainCols) {
            const int cnt = static_cast<int>((n - base) < kChainCols ? (n - base) : kChainCols);
#pragma unroll
            for (int r = 0; r < kChainRows; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c) tile[r * kChainStride + c * 64 + lane] = v[r * 4 + c];
            if (base + kChainCols < n) fetch(base + kChainCols);  // in flight while the lanes add
            __syncthreads();
            if (have) {
                const double* t = tile + lane * kChainStride;
                int i = 0;
                if (pass == 0) {
                    for (; i + 16 <= cnt; i += 16) {
                        double u[16];
#pragma unroll
                        for (int j = 0; j < 16; ++j) u[j] = t[i + j];
#pragma unroll
                        for (int j = 0; j < 16; ++j) {
                            acc += u[j];
                            if (COUNT) nfinite += finite_f64(u[j]) ? 1u : 0u;
                        }
                    }
                    for (; i < cnt; ++i) {
                        acc += t[i];
                        if (COUNT) nfinite += finite_f64(t[i]) ? 1u : 0u;
                    }
                } else {
                    for (; i + 16 <= cnt; i += 16) {
                        double u[16];
#pragma unroll
                        for (int j = 0; j < 16; ++j) {
                            const double diff = t[i + j] - mean;
                            u[j] = diff * diff;
                        }
#pragma unroll
                        for (int j = 0; j < 16; ++j) acc += u[j];
                    }
                    for (; i < cnt; ++i) {
                        const double diff = t[i] - mean;
                        acc += diff * diff;
                    }
                }
            }
            __syncthreads();
        }
        if (pass == 0)
            mean = acc / static_cast<double>(n);
        else
            ssd = acc;
    }
}

// BAYES thresholds, 16 (row, level) bands per single-wave workgroup
__global__ void __launch_bounds__(64) k_denoiser_bayes(const DenoiserThrArgs a) {
    __shared__ double tile[kChainRows * kChainStride];
    const int lane = threadIdx.x;
    const int64_t nbands = a.nrows * a.levels;
    const int64_t band0 = static_cast<int64_t>(blockIdx.x) * kChainRows;
    const int64_t n = a.n;
    const int64_t mine = band0 + lane;
    const bool have = lane < kChainRows && mine < nbands;
    const int64_t my_row = have ? mine / a.levels : 0;
    const int my_level = have ? static_cast<int>(mine - my_row * a.levels) + 1 : 1;
    const unsigned long long my_ptr = reinterpret_cast<unsigned long long>(
        a.coeffs + my_row * a.row_stride + static_cast<int64_t>(my_level - 1) * a.band_stride);
    const int nq = static_cast<int>((nbands - band0) < kChainRows ? (nbands - band0) : kChainRows);
    double mean, ssd;
    unsigned unused;
    lane_chains<false>(my_ptr, have, nq, n, tile, mean, ssd, unused);
    if (have) {
        const double variance = ssd / static_cast<double>(n);
        double sigma = a.sigma[my_row];
        if (a.level_scaled) sigma = sigma / sqrt(static_cast<double>(1 << (my_level - 1)));
        const double sigma2 = sigma * sigma;
        const double d = variance - sigma2;
        const double sx2 = d > 0.0 ? d : 0.0;
        const double sx = sqrt(sx2 + 1e-10);
        a.thr[my_row * a.levels + (my_level - 1)] = sigma2 / sx;
    }
}

// Streaming denoiser, STD estimator: level = sample standard deviation of each stream's ring, 16 streams per
// single-wave workgroup; 0 for a ring with fewer than two finite entries.
__global__ void __launch_bounds__(64) k_stream_std(const StreamNoiseArgs a) {
    __shared__ double tile[kChainRows * kChainStride];
    const int lane = threadIdx.x;
    const int64_t s0 = static_cast<int64_t>(blockIdx.x) * kChainRows;
    const int64_t mine = s0 + lane;
    const bool have = lane < kChainRows && mine < a.nrows;
    const unsigned long long my_ptr = reinterpret_cast<unsigned long long>(a.window + (have ? mine : 0) * a.w);
    const int nq = static_cast<int>((a.nrows - s0) < kChainRows ? (a.nrows - s0) : kChainRows);
    double mean, ssd;
    unsigned nfinite;
    lane_chains<true>(my_ptr, have, nq, a.w, tile, mean, ssd, nfinite);
    if (have) a.level[mine] = nfinite >= 2 ? sqrt(ssd / static_cast<double>(a.w - 1)) : 0.0;
}

// ---------------------------------------------------------------------------
// structure-of-arrays batch (modwt/BatchSIMDMODWT.java): sample-major storage, soa[t * B + b]
// ---------------------------------------------------------------------------
//
// One lane per (t, b), lanes along b: every tap reads a contiguous run of the batch.  The three
// arithmetic flavours of the reference are kept apart because they round differently: Haar with the
// literal constants 0.5 and no running sum, four-tap filters as a chain of products without a zero
// start, everything else as running sums from +0.0.

struct SoaArgs {
    const double* in;   // [N][B]
    double* approx;     // [N][B]
    double* detail;     // [N][B]
    int64_t batch, n;
    int32_t flen;
    int32_t flavour;    // 0 general, 1 Haar constants, 2 four taps
    Taps taps;          // scaled by 1/sqrt(2) (unused for flavour 1)
};

__global__ void __launch_bounds__(kThreads) k_soa_modwt(const SoaArgs a) {
    const int64_t gid = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
    if (gid >= a.n * a.batch) return;
    const int64_t t = gid / a.batch;
    const int64_t b = gid - t * a.batch;
    const int64_t n = a.n, B = a.batch;
    const double* __restrict__ x = a.in;
    double va, vd;
    if (a.flavour == 1) {
        const double s0 = x[t * B + b], s1 = x[mod_wrap(t - 1, n) * B + b];
        va = s0 * 0.5 + s1 * 0.5;
        vd = s0 * 0.5 + s1 * -0.5;
    } else if (a.flavour == 2) {
        const double s0 = x[t * B + b], s1 = x[mod_wrap(t - 1, n) * B + b], s2 = x[mod_wrap(t - 2, n) * B + b],
                     s3 = x[mod_wrap(t - 3, n) * B + b];
        va = ((s0 * a.taps.lo[0] + s1 * a.taps.lo[1]) + s2 * a.taps.lo[2]) + s3 * a.taps.lo[3];
        vd = ((s0 * a.taps.hi[0] + s1 * a.taps.hi[1]) + s2 * a.taps.hi[2]) + s3 * a.taps.hi[3];
    } else {
        double sa = 0.0, sd = 0.0;
        int64_t src = t;
        for (int l = 0; l < a.flen; ++l) {
            const double v = x[src * B + b];
            sa = sa + v * a.taps.lo[l];
            sd = sd + v * a.taps.hi[l];
            src = src - 1;
            if (src < 0) src = mod_wrap(src, n);
        }
        va = sa;
        vd = sd;
    }
    a.approx[gid] = va;
    a.detail[gid] = vd;
}

// The same transform with a register window: a lane owns VEC adjacent signals and T consecutive samples of them, loads the
// T + L - 1 rows of the sample-major matrix its outputs read once (every load of the workgroup is one contiguous run of
// the batch) and slides over them.  k_soa_modwt reads L rows per output and leaves the reuse to the caches: 3.3 TB/s of
// algorithmic traffic for eight taps against 5.6 for Haar.  L = filter length (FLAV 1: the Haar constants, two rows;
// FLAV 2: the four-tap chain; FLAV 0: running sums from +0.0, taps in ascending order as in k_soa_modwt).
// Blocks: batch chunks fastest, so neighbouring blocks stream the same rows.
template <int L, int FLAV, int T, int VEC>
__global__ void __launch_bounds__(kThreads) k_soa_modwt_run(const SoaArgs a) {
#pragma clang fp contract(off)
    static_assert(VEC == 1 || VEC == 2, "one signal or an aligned pair of signals per lane");
    static_assert(FLAV != 1 || L == 2, "the Haar flavour reads two rows");
    static_assert(FLAV != 2 || L == 4, "the chain flavour has four taps");
    const int64_t n = a.n, B = a.batch;
    const int64_t per_block = static_cast<int64_t>(kThreads) * VEC;
    const int64_t bchunks = (B + per_block - 1) / per_block;
    const int64_t tt = blockIdx.x / bchunks;
    const int64_t bc = blockIdx.x - tt * bchunks;
    const int64_t b = (bc * kThreads + threadIdx.x) * VEC;  // VEC == 2: B is even, so b + 1 < B as well
    if (b >= B) return;
    const int64_t t0 = tt * T;
    const double* __restrict__ x = a.in;
    constexpr int NW = T + L - 1;
    double w[NW][VEC];
    {
        // row of window position m: t0 - (L - 1) + m, wrapped into [0, n) (any number of times when the filter is
        // longer than the row); rows at or beyond n belong to outputs that do not exist
        int64_t ts = mod_wrap(t0 - (L - 1), n);
#pragma unroll
        for (int m = 0; m < NW; ++m) {
            if (t0 - (L - 1) + m < n) {
                const double* p = x + ts * B + b;
                if (VEC == 2) {
                    const double2 v = *reinterpret_cast<const double2*>(p);
                    w[m][0] = v.x;
                    w[m][VEC - 1] = v.y;
                } else {
                    w[m][0] = *p;
                }
            } else {
#pragma unroll
                for (int c = 0; c < VEC; ++c) w[m][c] = 0.0;
            }
            ts = ts + 1;
            if (ts == n) ts = 0;
        }
    }
    double lo[L], hi[L];
#pragma unroll
    for (int l = 0; l < L; ++l) {
        lo[l] = a.taps.lo[l];
        hi[l] = a.taps.hi[l];
    }
#pragma unroll
    for (int k = 0; k < T; ++k) {
        if (t0 + k < n) {
            double va[VEC], vd[VEC];
#pragma unroll
            for (int c = 0; c < VEC; ++c) {
                // sample t - l sits at window position k + (L - 1) - l
                if (FLAV == 1) {
                    const double s0 = w[k + 1][c], s1 = w[k][c];
                    va[c] = s0 * 0.5 + s1 * 0.5;
                    vd[c] = s0 * 0.5 + s1 * -0.5;
                } else if (FLAV == 2) {
                    const double s0 = w[k + 3][c], s1 = w[k + 2][c], s2 = w[k + 1][c], s3 = w[k][c];
                    va[c] = ((s0 * lo[0] + s1 * lo[1]) + s2 * lo[2]) + s3 * lo[3];
                    vd[c] = ((s0 * hi[0] + s1 * hi[1]) + s2 * hi[2]) + s3 * hi[3];
                } else {
                    double sa = 0.0, sd = 0.0;
#pragma unroll
                    for (int l = 0; l < L; ++l) {
                        const double v = w[k + (L - 1) - l][c];
                        sa = sa + v * lo[l];
                        sd = sd + v * hi[l];
                    }
                    va[c] = sa;
                    vd[c] = sd;
                }
            }
            const int64_t gid = (t0 + k) * B + b;
            if (VEC == 2) {
                *reinterpret_cast<double2*>(a.approx + gid) = make_double2(va[0], va[VEC - 1]);
                *reinterpret_cast<double2*>(a.detail + gid) = make_double2(vd[0], vd[VEC - 1]);
            } else {
                a.approx[gid] = va[0];
                a.detail[gid] = vd[0];
            }
        }
    }
}

// out[c][r] = in[r][c]; 32 x 32 tiles through LDS (row stride 33: conflict-free both ways)
__global__ void __launch_bounds__(kThreads) k_transpose(const double* __restrict__ in, double* __restrict__ out,
                                                        int64_t rows, int64_t cols) {
    __shared__ double tile[32][33];
    const int64_t tiles_c = (cols + 31) / 32;
    const int64_t tr = blockIdx.x / tiles_c, tc = blockIdx.x - tr * tiles_c;
    const int lx = threadIdx.x & 31, ly = threadIdx.x >> 5;  // 32 x 8
    for (int k = ly; k < 32; k += 8) {
        const int64_t r = tr * 32 + k, c = tc * 32 + lx;
        if (r < rows && c < cols) tile[k][lx] = in[r * cols + c];
    }
    __syncthreads();
    for (int k = ly; k < 32; k += 8) {
        const int64_t c = tc * 32 + k, r = tr * 32 + lx;
        if (r < rows && c < cols) out[c * rows + r] = tile[lx][k];
    }
}

// ---------------------------------------------------------------------------
// sparse detail bands (swt/SWTResult.java toSparse / toFull)
// ---------------------------------------------------------------------------
//
// One workgroup per detail band (row, level).  Pass 1 (indices == nullptr) counts the entries with
// |v| > threshold; pass 2 writes them in index order at offsets[band]: every lane owns four consecutive
// elements of a 1024-element chunk, a workgroup prefix sum of the lanes' counts gives each lane its place.
// Pass 2 never writes at or beyond offsets[band + 1]: when the band holds more entries than the caller's
// prefix sum left room for (a different threshold or different coefficients than in pass 1), the surplus is
// dropped, and the number of entries the band really holds goes to counts[band] (if given) so that the caller
// can tell.

struct SparsifyArgs {
    const double* coeffs;  // band (r, j) at coeffs + r*row_stride + j*band_stride
    int64_t row_stride, band_stride;
    int64_t n;
    int32_t levels;
    double threshold;
    int64_t* counts;         // [rows * levels]: pass 1 result; optional in pass 2 (entries found)
    const int64_t* offsets;  // [rows * levels + 1] (pass 2)
    int32_t* indices;        // pass 2
    double* values;          // pass 2
};

__global__ void __launch_bounds__(kThreads) k_sparsify(const SparsifyArgs a) {
    __shared__ unsigned wsum[4];
    __shared__ unsigned long long red[kThreads];
    const int tid = threadIdx.x;
    const int64_t band = blockIdx.x;
    const int64_t row = band / a.levels;
    const int level = static_cast<int>(band - row * a.levels);
    const double* __restrict__ c = a.coeffs + row * a.row_stride + static_cast<int64_t>(level) * a.band_stride;
    const int64_t n = a.n;
    if (a.indices == nullptr) {
        unsigned long long cnt = 0;
        for (int64_t i = tid; i < n; i += kThreads) cnt += fabs(c[i]) > a.threshold ? 1u : 0u;
        red[tid] = cnt;
        __syncthreads();
        for (int s = kThreads / 2; s > 0; s >>= 1) {
            if (tid < s) red[tid] += red[tid + s];
            __syncthreads();
        }
        if (tid == 0) a.counts[band] = static_cast<int64_t>(red[0]);
        return;
    }
    const int64_t start = a.offsets[band], end = a.offsets[band + 1];
    int64_t run = start;
    for (int64_t base = 0; base < n; base += 4 * kThreads) {
        const int64_t i0 = base + 4 * tid;
        double v[4];
        bool keep[4];
        unsigned mine = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            v[k] = (i0 + k < n) ? c[i0 + k] : 0.0;
            keep[k] = (i0 + k < n) && fabs(v[k]) > a.threshold;
            mine += keep[k] ? 1u : 0u;
        }
        unsigned inc = wave_inclusive_scan(mine);
        if ((tid & 63) == 63) wsum[tid >> 6] = inc;
        __syncthreads();
        unsigned before = 0, total = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            if (w < (tid >> 6)) before += wsum[w];
            total += wsum[w];
        }
        int64_t pos = run + before + (inc - mine);
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (keep[k]) {
                if (pos < end) {
                    a.indices[pos] = static_cast<int32_t>(i0 + k);
                    a.values[pos] = v[k];
                }
                ++pos;
            }
        run += total;
        __syncthreads();
    }
    if (tid == 0 && a.counts != nullptr) a.counts[band] = run - start;
}

struct DensifyArgs {
    const int64_t* offsets;  // [bands + 1]
    const int32_t* indices;
    const double* values;
    double* out;             // band b at out + (b / levels)*row_stride + (b % levels)*band_stride, zeroed beforehand
    int64_t row_stride, band_stride;
    int64_t n;
    int32_t levels;
};

__global__ void __launch_bounds__(kThreads) k_densify(const DensifyArgs a) {
    const int64_t band = blockIdx.x;
    const int64_t row = band / a.levels;
    const int level = static_cast<int>(band - row * a.levels);
    double* out = a.out + row * a.row_stride + static_cast<int64_t>(level) * a.band_stride;
    const int64_t e0 = a.offsets[band], e1 = a.offsets[band + 1];
    for (int64_t e = e0 + threadIdx.x; e < e1; e += kThreads) {
        const int64_t i = a.indices[e];
        if (i >= 0 && i < a.n) out[i] = a.values[e];  // an index outside the band is the caller's error, not a reason to write there
    }
}

}  // namespace vw
