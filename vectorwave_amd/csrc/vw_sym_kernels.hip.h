// vw_sym_kernels.hip.h -- the SYMMETRIC inverse on tiles.
//
// The reference's SYMMETRIC reconstruction is not the mirror image of its PERIODIC / ZERO_PADDING one: it reads
// LEFT of the output sample, R_{j-1}[t] = sum_l lo[l] R_j[t - l s] (+) hi[l] W_j[t - l s], reflects an index below
// zero into the row (half-sample reflection, sample -1 - i is sample i; util/MathUtils.java:31-52) and, in the
// multi-level transform, adds the two products to the sum one after the other instead of as a pair
// (modwt/MultiLevelMODWTTransform.java:664-685; the single-level MODWTTransform.inverse keeps the pair sum,
// modwt/MODWTTransform.java:269-287).  So it has the data flow of a forward level with two inputs: a tile plus a
// LEFT halo of 512 samples in LDS, levels lev1 .. lev0 in place, the halo of R_{j-1} recomputed where its own left
// context is still inside the halo -- except in the first tile of a row, whose halo is the mirror image of the
// tile's own first 512 samples at every level and is copied from there after each level (bands: mirrored on load).
// Rows of at least 512 samples; sum_j (L - 1) 2^(j-1) <= 510 for the level group.
#pragma once
#include "vw_kernels.hip.h"

#pragma clang fp contract(off)

namespace vw {

// samples (t, t + 1) of a row, t even: mirrored left of the row start, zeros from the row end on
template <bool VEC2>
__device__ __forceinline__ double2 sym_fetch_pair(const double* __restrict__ src, int64_t t, int64_t n) {
    double2 x = make_double2(0.0, 0.0);
    if (t >= 0) {
        if (VEC2) {
            if (t < n) x = *reinterpret_cast<const double2*>(src + t);
        } else {
            if (t < n) x.x = src[t];
            if (t + 1 < n) x.y = src[t + 1];
        }
    } else {
        if (VEC2) {
            const double2 m = *reinterpret_cast<const double2*>(src + (-t - 2));
            x = make_double2(m.y, m.x);
        } else {
            x.x = src[-t - 1];
            x.y = src[-t - 2];
        }
    }
    return x;
}

// one tap of one output pair: PAIRSUM sum += lo*r + hi*w; otherwise sum += lo*r, then sum += hi*w
template <bool PAIRSUM>
__device__ __forceinline__ void sym_tap(double (&acc)[2], double flo, double fhi, double r0, double r1, double w0, double w1,
                                        bool do_lo, bool do_hi) {
    if (PAIRSUM) {
        acc[0] = acc[0] + (flo * r0 + fhi * w0);
        acc[1] = acc[1] + (flo * r1 + fhi * w1);
    } else {
        if (do_lo) {
            acc[0] = acc[0] + flo * r0;
            acc[1] = acc[1] + flo * r1;
        }
        if (do_hi) {
            acc[0] = acc[0] + fhi * w0;
            acc[1] = acc[1] + fhi * w1;
        }
    }
}

// All taps of the output pair at buffer position `pos` (even): operands at pos - l*s.
// LT > 0: both filters have LT taps (unrolled); LT == 0: llo / lhi taps, PAIRSUM loops l < llo with both filters.
template <bool S1, bool PAIRSUM, int LT>
__device__ __forceinline__ void sym_pair(const double* __restrict__ bufR, const double* __restrict__ bufW, const Taps& taps,
                                         int llo, int lhi, int s, int pos, double (&acc)[2]) {
    acc[0] = acc[1] = 0.0;
    if (LT > 0) {
        if (S1) {
            // samples pos + 1 - LT .. pos + 1 of both operands, as aligned pairs
            constexpr int NWIN = LT / 2 + 1;
            double2 rw[NWIN], ww[NWIN];
#pragma unroll
            for (int q = 0; q < NWIN; ++q) {
                rw[q] = *reinterpret_cast<const double2*>(bufR + pos - LT + 2 * q);
                ww[q] = *reinterpret_cast<const double2*>(bufW + pos - LT + 2 * q);
            }
#pragma unroll
            for (int l = 0; l < LT; ++l) {
                const int i0 = LT - l, i1 = LT + 1 - l;  // window samples of pos - l and pos + 1 - l
                const double r0 = (i0 & 1) ? rw[i0 / 2].y : rw[i0 / 2].x;
                const double r1 = (i1 & 1) ? rw[i1 / 2].y : rw[i1 / 2].x;
                const double w0 = (i0 & 1) ? ww[i0 / 2].y : ww[i0 / 2].x;
                const double w1 = (i1 & 1) ? ww[i1 / 2].y : ww[i1 / 2].x;
                sym_tap<PAIRSUM>(acc, taps.lo[l], taps.hi[l], r0, r1, w0, w1, true, true);
            }
        } else {
            double2 r[LT > 0 ? LT : 1], w[LT > 0 ? LT : 1];
#pragma unroll
            for (int l = 0; l < LT; ++l) {
                r[l] = *reinterpret_cast<const double2*>(bufR + pos - l * s);
                w[l] = *reinterpret_cast<const double2*>(bufW + pos - l * s);
            }
#pragma unroll
            for (int l = 0; l < LT; ++l) sym_tap<PAIRSUM>(acc, taps.lo[l], taps.hi[l], r[l].x, r[l].y, w[l].x, w[l].y, true, true);
        }
        return;
    }
    const int lmax = PAIRSUM ? llo : (llo > lhi ? llo : lhi);
    if (S1) {
        double r1 = bufR[pos + 1], w1 = bufW[pos + 1];
        for (int l = 0; l < lmax; ++l) {
            const double r0 = bufR[pos - l], w0 = bufW[pos - l];
            sym_tap<PAIRSUM>(acc, taps.lo[l], taps.hi[l], r0, r1, w0, w1, l < llo, l < lhi);
            r1 = r0;
            w1 = w0;
        }
    } else {
        for (int l = 0; l < lmax; ++l) {
            const double2 r = *reinterpret_cast<const double2*>(bufR + pos - l * s);
            const double2 w = *reinterpret_cast<const double2*>(bufW + pos - l * s);
            sym_tap<PAIRSUM>(acc, taps.lo[l], taps.hi[l], r.x, r.y, w.x, w.y, l < llo, l < lhi);
        }
    }
}

// a.pend[j]: first halo position (even) from which R_j is valid; a.lh / a.lg: taps of lo / hi.
// Buffer position p <-> sample t0 - 512 + p.
// THR: a.thr_mode (a mode of apply_thr) is applied to W_j as it is staged, with the threshold of (row, level j).
template <int PPT, bool VEC2, bool PAIRSUM, int LT, bool THR = false>
__global__ void __launch_bounds__(kThreads) k_inv_sym(const InvArgs a) {
    extern __shared__ double2 vw_smem[];
    constexpr int HALO = kChunk;
    constexpr int T = PPT * kChunk;
    constexpr int M = HALO + T;
    double* bufR = reinterpret_cast<double*>(vw_smem);
    double* bufW = bufR + M;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    const int64_t id = xcd_remap(blockIdx.x, gridDim.x);
    const int64_t row = id / a.ntiles;
    const int tile = static_cast<int>(id - row * a.ntiles);
    const int64_t t0 = static_cast<int64_t>(tile) * T;
    const int64_t n = a.n;
    const bool sym0 = (tile == 0);  // the halo lies left of the row start
    const double* __restrict__ wbase = a.win + row * a.w_row_stride;

    double2 pre[PPT + 1];  // PPT tile pairs and the halo pair of this lane
    const bool interior = VEC2 && tile > 0 && t0 + T <= n;  // no mirror, no row end: plain aligned loads
    auto fetch = [&](const double* __restrict__ src, bool use, int ps) {
        if (!use) {
#pragma unroll
            for (int k = 0; k <= PPT; ++k) pre[k] = make_double2(0.0, 0.0);
        } else if (interior) {
#pragma unroll
            for (int k = 0; k < PPT; ++k) pre[k] = *reinterpret_cast<const double2*>(src + t0 + k * kChunk + 2 * tid);
            pre[PPT] = (2 * tid >= ps) ? *reinterpret_cast<const double2*>(src + t0 - HALO + 2 * tid) : make_double2(0.0, 0.0);
        } else {
#pragma unroll
            for (int k = 0; k < PPT; ++k) pre[k] = sym_fetch_pair<VEC2>(src, t0 + k * kChunk + 2 * tid, n);
            pre[PPT] = (2 * tid >= ps) ? sym_fetch_pair<VEC2>(src, t0 - HALO + 2 * tid, n) : make_double2(0.0, 0.0);
        }
    };
    auto check = [&](int band) {  // non-finite coefficients of the tile part just fetched
        const int64_t flat0 = row * a.w_row_stride + static_cast<int64_t>(band) * a.w_band_stride;
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            const int64_t t = t0 + k * kChunk + 2 * tid;
            if (t < n && !finite_f64(pre[k].x)) atomicMin(a.first_bad, static_cast<unsigned long long>(flat0 + t));
            if (t + 1 < n && !finite_f64(pre[k].y)) atomicMin(a.first_bad, static_cast<unsigned long long>(flat0 + t + 1));
        }
    };
    // W_j from the registers into LDS (thresholded on the way when denoising: the rule is element-wise, so the mirrored
    // halo agrees with the tile), then W_{j-1} on its way while level j is computed
    auto publish = [&](int j) {
        if (a.first_bad != nullptr) check(j - 1);
        if (THR) {
            const double thr = uniform_load_f64(a.thr + row * a.thr_row_stride + (j - 1) * a.thr_level_stride);
#pragma unroll
            for (int k = 0; k <= PPT; ++k) {
                pre[k].x = apply_thr(pre[k].x, thr, a.thr_mode);
                pre[k].y = apply_thr(pre[k].y, thr, a.thr_mode);
            }
        }
#pragma unroll
        for (int k = 0; k <= PPT; ++k) *reinterpret_cast<double2*>(bufW + (k < PPT ? HALO + k * kChunk : 0) + 2 * tid) = pre[k];
        lds_barrier();  // R_j and W_j visible
        if (j > a.lev0) fetch(wbase + static_cast<int64_t>(j - 2) * a.w_band_stride, ((a.band_mask >> (j - 2)) & 1u) != 0, a.pend[j - 1]);
    };
    auto store = [&](const double (&acc)[PPT][2]) {
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
    };

    // R_{lev1}
    fetch(a.rin + row * a.rin_stride, ((a.band_mask >> 31) & 1u) != 0, a.pend[a.lev1]);
    if (a.first_bad != nullptr && a.check_rin) check(a.lev1);
#pragma unroll
    for (int k = 0; k < PPT; ++k) *reinterpret_cast<double2*>(bufR + HALO + k * kChunk + 2 * tid) = pre[k];
    *reinterpret_cast<double2*>(bufR + 2 * tid) = pre[PPT];
    // W_{lev1}
    fetch(wbase + static_cast<int64_t>(a.lev1 - 1) * a.w_band_stride, ((a.band_mask >> (a.lev1 - 1)) & 1u) != 0, a.pend[a.lev1]);

    // levels with a stride of at least one pair; level 1 (odd offsets, always the last of its group) follows the loop
    int j = a.lev1;
    for (; j >= a.lev0 && j >= 2; --j) {
        const int s = 1 << (j - 1);
        publish(j);
        // the halo of R_{j-1}: only while lower levels remain, only where its own context is valid, never in the first
        // tile of a row (mirrored below); switched per wave, a lane left of the zone recomputes the zone's first pair
        const int vs = (j > a.lev0) ? a.pend[j - 1] : HALO;
        const bool hact = (j > a.lev0) && !sym0 && (128 * wave + 128 > vs);
        const bool hwr = hact && (2 * tid >= vs);
        const int hp = (2 * tid >= vs) ? 2 * tid : vs;
        double acc[PPT][2], ah[2] = {0.0, 0.0};
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            sym_pair<false, PAIRSUM, LT>(bufR, bufW, a.taps, a.lh, a.lg, s, HALO + k * kChunk + 2 * tid, acc[k]);
            // one pair's operand window at a time: left alone, the compiler requests the windows of all four pairs up
            // front (256 registers, one wave per SIMD); three workgroups per CU hide the LDS latency instead
            asm volatile("" : "+v"(acc[k][0]), "+v"(acc[k][1]) : : "memory");  // (the sums are due here, the next window not before)
            __builtin_amdgcn_sched_barrier(0);
        }
        if (hact) sym_pair<false, PAIRSUM, LT>(bufR, bufW, a.taps, a.lh, a.lg, s, hp, ah);

        if (j == a.lev0) {
            store(acc);
        } else {
            lds_barrier();  // every lane has finished reading R_j / W_j
#pragma unroll
            for (int k = 0; k < PPT; ++k) *reinterpret_cast<double2*>(bufR + HALO + k * kChunk + 2 * tid) = make_double2(acc[k][0], acc[k][1]);
            if (hwr) *reinterpret_cast<double2*>(bufR + 2 * tid) = make_double2(ah[0], ah[1]);
            if (sym0) {
                lds_barrier();
                // R_{j-1}[-1 - i] = R_{j-1}[i]: halo pair (2 tid, 2 tid + 1) is tile pair 510 - 2 tid, swapped
                const double2 m = *reinterpret_cast<const double2*>(bufR + HALO + (HALO - 2 - 2 * tid));
                *reinterpret_cast<double2*>(bufR + 2 * tid) = make_double2(m.y, m.x);
            }
            // the barrier inside the next publish() makes R_{j-1} visible
        }
    }
    if (j == 1 && a.lev0 == 1) {
        publish(1);
        double acc[PPT][2];
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            sym_pair<true, PAIRSUM, LT>(bufR, bufW, a.taps, a.lh, a.lg, 1, HALO + k * kChunk + 2 * tid, acc[k]);
            asm volatile("" : "+v"(acc[k][0]), "+v"(acc[k][1]) : : "memory");  // (the sums are due here, the next window not before)
            __builtin_amdgcn_sched_barrier(0);
        }
        store(acc);
    }
}

}  // namespace vw
