// vw_wave_kernels.hip.h -- the inverse cascade with one wavefront per run of a row, no workgroup barriers.
//
// k_inv_ring keeps a 2048-sample tile per 256-lane workgroup: two barriers per level, two workgroups per CU, a
// 512-sample halo loaded and partly recomputed per tile.  Its waves spend a third of their time parked at barriers
// and waits, and what bounds it is how badly memory time and fp64 time overlap, not either of them.  Here the unit
// is ONE wavefront: a workgroup of 64 lanes walks a run of 512-sample chunks of one row from right to left and owns
// everything it touches, so the only synchronisation left is the wave's own counters (vmcnt for the LDS-DMA, lgkmcnt
// for LDS), and every wave on a CU is in a different phase by itself.
//
//  * chunk = 256 pairs, four per lane; the register-blocked operand windows of k_inv_blk (a lane owns the pairs
//    P0 + k*sp, k = 0..3, of a level with stride sp pairs and walks the window P0 + c*sp, c = 0..LT+2, once);
//  * nothing is recomputed: the right context of level j -- the first (LT-1)*2^(j-1) samples of R_j and W_j of the
//    chunk done just before, its "heads" -- stays in LDS; a window operand right of the chunk is read from the head
//    storage (a per-lane address select that is loop invariant, so it costs registers, not instructions);
//  * a run starts `ndry` chunks early with the stores switched off: the cascade is FIR, after that many samples
//    every head is exact whatever it held before;
//  * bands travel HBM -> LDS by LDS-DMA, three bands ahead of the level that is being computed, into four slots;
//    the running reconstruction has a buffer of its own, so the next chunk's a_J can land while level 1 still reads;
//  * every LDS address of every level is computed once per run, not once per chunk.
//
// LDS per wave: 5 chunk buffers of 4 KiB + the heads of R and W of every level (LT = 8, six levels: 27.6 KB),
// five waves per CU.
#pragma once
#include "vw_kernels.hip.h"

#pragma clang fp contract(off)

namespace vw {

constexpr int kWvC = 256;              // pairs per chunk
constexpr int kWvSamples = 2 * kWvC;   // samples per chunk
constexpr int kWvBuf = kWvC * 16;      // bytes per chunk buffer
constexpr int kWvHeads = 5 * kWvBuf;   // head storage behind the running reconstruction and the four band slots

// pairs of right context at level j, and where the heads of level j start inside the head storage (in pairs)
constexpr int wv_hp(int lt, int j) { return (((lt - 1) << (j - 1)) + 1) >> 1; }
constexpr int wv_hoff(int lt, int j) {
    int s = 0;
    for (int i = 1; i < j; ++i) s += wv_hp(lt, i);
    return s;
}
constexpr int wv_thr_off(int lt, int J) { return kWvHeads + 2 * wv_hoff(lt, J + 1) * 16; }  // per-level thresholds of the row
constexpr int wv_lds_bytes(int lt, int J) { return wv_thr_off(lt, J) + 8 * 8; }
// A chunk's bands in the order they are consumed: a_J, W_J, ..., W_1; the band at position q lands in slot q mod 4.
constexpr int wv_slot_a() { return kWvBuf; }
constexpr int wv_slot_w(int J, int j) { return kWvBuf * (1 + ((J - j + 1) & 3)); }

// The four LDS-DMA instructions of one band chunk: samples [t0, t0 + 512) of `band_row` into the slot at LDS byte
// address `ldst`, in the layout of the level that reads it (`voff`: the lane's swizzled byte offset inside a 1 KiB block).
// (HINT: 0 default cache policy, 1 streaming; the tuning build also knows the other combinations of the three policy bits)
#define VW_WV_ISSUE_ASM(POLICY)                                                                  \
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"                          \
                 "global_load_lds_dwordx4 %1, %2" POLICY "\n\t"                                  \
                 "global_load_lds_dwordx4 %1, %2 offset:1024" POLICY "\n\t"                      \
                 "global_load_lds_dwordx4 %1, %2 offset:2048" POLICY "\n\t"                      \
                 "global_load_lds_dwordx4 %1, %2 offset:3072" POLICY "\n\t"                      \
                 "s_mov_b32 m0, %0"                                                              \
                 : "=&s"(keep)                                                                   \
                 : "v"(voff), "s"(tsrc), "s"(ldst)                                               \
                 : "memory")
template <int HINT = 0>
__device__ __forceinline__ void wv_issue(const double* __restrict__ band_row, int64_t t0, unsigned voff, unsigned ldst) {
    const double* tsrc = band_row + t0;  // wave-uniform
    // the instruction's immediate offset is added to the global address AND to the LDS address: one M0, one address
    // register and one offset register serve the four 1 KiB blocks
    unsigned keep;
    if (HINT == 1) VW_WV_ISSUE_ASM(" nt");
#ifdef VW_TUNING
    else if (HINT == 2) VW_WV_ISSUE_ASM(" sc1");
    else if (HINT == 3) VW_WV_ISSUE_ASM(" sc0 sc1");
    else if (HINT == 4) VW_WV_ISSUE_ASM(" sc1 nt");
    else if (HINT == 5) VW_WV_ISSUE_ASM(" sc0 sc1 nt");
    else if (HINT == 6) VW_WV_ISSUE_ASM(" sc0");
    else if (HINT == 7) VW_WV_ISSUE_ASM(" sc0 nt");
#endif
    else VW_WV_ISSUE_ASM("");
}

// 16-byte store to a wave-uniform base plus a 32-bit byte offset per lane.  Written out so that the address stays an SGPR
// pair and one VGPR: left to the compiler, the per-lane 64-bit address of the output row lives in a register pair across
// the whole chunk loop (and was spilled: a scratch reload per chunk, with a full vmcnt(0) drain in front of it).
// The s_nop belongs to the store: a store of more than 64 bits reads its data registers up to two wait states after it
// issues, and a VALU instruction that overwrites one of them inside that window changes what is stored.  The compiler
// keeps that distance for its own stores but does not look inside inline assembly, and it reuses the data registers,
// dead after this statement, at once (seen: the index of the next block's bounds test written into the first data
// register one instruction later -- the low half of one double of a few lanes came out as that index, now and then).
__device__ __forceinline__ void wv_store16(double* sbase, unsigned voff, double2 v, int nt) {
    vw_f64x2 x;
    x.x = v.x;
    x.y = v.y;
#ifdef VW_TUNING
    if (nt == 2) {
        asm volatile("global_store_dwordx4 %0, %1, %2 sc1\n\ts_nop 1" : : "v"(voff), "v"(x), "s"(sbase) : "memory");
        return;
    }
    if (nt == 3) {
        asm volatile("global_store_dwordx4 %0, %1, %2 sc0 sc1\n\ts_nop 1" : : "v"(voff), "v"(x), "s"(sbase) : "memory");
        return;
    }
    if (nt == 4) {
        asm volatile("global_store_dwordx4 %0, %1, %2 sc1 nt\n\ts_nop 1" : : "v"(voff), "v"(x), "s"(sbase) : "memory");
        return;
    }
    if (nt == 5) {
        asm volatile("global_store_dwordx4 %0, %1, %2 sc0 sc1 nt\n\ts_nop 1" : : "v"(voff), "v"(x), "s"(sbase) : "memory");
        return;
    }
#endif
    if (nt)
        asm volatile("global_store_dwordx4 %0, %1, %2 nt\n\ts_nop 1" : : "v"(voff), "v"(x), "s"(sbase) : "memory");
    else
        asm volatile("global_store_dwordx4 %0, %1, %2\n\ts_nop 1" : : "v"(voff), "v"(x), "s"(sbase) : "memory");
}

__device__ __forceinline__ void wv_wait_vm(int n) {  // n is wave-uniform
    switch (n) {
        case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
        case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
        case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
        case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
        case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

// loop-invariant LDS byte addresses of a blocked level (j >= 2)
template <int LT>
struct WvLevelAddr {
    int ra[LT + 3];  // R operand of window position c (chunk buffer or head storage)
    int dA, dB;      // W operand = R operand + D for c < 4, + dA for c in 4..7, + dB for c >= 8
    int wb[4];       // write-back of accumulator k as R_{j-1}, in the layout of level j-1
};

template <int LT, int J, int j>
__device__ __forceinline__ void wv_addr_init(int lane, WvLevelAddr<LT>& A) {
    constexpr int sl = j - 2, sp = 1 << sl, nb = 64 >> sl;
    constexpr int rbase = (j == J) ? wv_slot_a() : 0;
    constexpr int D = wv_slot_w(J, j) - rbase;
    constexpr int HR = kWvHeads + wv_hoff(LT, j) * 16;
    constexpr int E = wv_hoff(LT, J + 1) * 16;  // W heads sit behind all R heads
    const int P0 = ((lane >> sl) << (sl + 2)) + (lane & (sp - 1));
#pragma unroll
    for (int c = 0; c < LT + 3; ++c) {
        const int p = P0 + c * sp;
        A.ra[c] = (p < kWvC) ? rbase + (blk_slot<sl>(p) << 4) : HR + ((p - kWvC) << 4);
    }
    const int blk = lane >> sl;
    A.dA = (blk >= nb - 1) ? E : D;
    A.dB = (blk >= nb - 2) ? E : D;
#pragma unroll
    for (int k = 0; k < 4; ++k) A.wb[k] = blk_slot<sl - 1>(P0 + k * sp) << 4;
}

// One blocked level of one chunk.  `post` runs once the operand windows are in registers (the slots read here are
// dead from then on): it issues the LDS-DMA of a later band.
template <int LT, int J, int j, bool FMA, typename Post>
__device__ __forceinline__ void wv_level(char* lds, int lane, const WvLevelAddr<LT>& A, int vswz, const double (&tl)[LT],
                                         const double (&th)[LT], bool ablate_math, Post post) {
    constexpr int rbase = (j == J) ? wv_slot_a() : 0;
    constexpr int D = wv_slot_w(J, j) - rbase;
    constexpr int HR = kWvHeads + wv_hoff(LT, j) * 16;
    constexpr int E = wv_hoff(LT, J + 1) * 16;
    constexpr int HPJ = wv_hp(LT, j);
    constexpr int NH = (HPJ + 63) / 64;
    static_assert(NH <= 2, "at most two head pairs per lane");
    // The lanes of the wave exchange data through LDS (what one lane stored -- a write-back, the zeros or the thresholded
    // values of a landed band -- another lane reads here), and the compiler reasons about ONE lane: it may move a load above
    // a store that this lane cannot alias.  Every such hand-over therefore sits behind a compiler-level memory barrier (the
    // hardware serves a wave's LDS requests in order anyway).
    asm volatile("" ::: "memory");
    double2 r[LT + 3], w[LT + 3];
    double2 hr0 = make_double2(0.0, 0.0), hw0 = hr0, hr1 = hr0, hw1 = hr0;
#pragma unroll
    for (int c = 0; c < LT + 3; ++c) {
        r[c] = *reinterpret_cast<const double2*>(lds + A.ra[c]);
        const int wa = A.ra[c] + (c < 4 ? D : (c < 8 ? A.dA : A.dB));
        w[c] = *reinterpret_cast<const double2*>(lds + wa);
    }
    // this chunk's heads: the first HPJ pairs of R_j and W_j (lane i takes pairs i and i + 64).  They are requested with
    // the window and stored behind it: the LDS serves a wave's requests in order, so the window reads have taken the old
    // heads by then, and `post` may overwrite a slot this level has read.
    {
        const int src = rbase + vswz;  // swizzle of lane + 64 i = swizzle of lane, + 64 i
        hr0 = *reinterpret_cast<const double2*>(lds + src);
        hw0 = *reinterpret_cast<const double2*>(lds + src + D);
        if (NH > 1) {
            hr1 = *reinterpret_cast<const double2*>(lds + src + 1024);
            hw1 = *reinterpret_cast<const double2*>(lds + src + 1024 + D);
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    {
        const int dst = HR + (lane << 4);
        if (lane < HPJ) {
            *reinterpret_cast<double2*>(lds + dst) = hr0;
            *reinterpret_cast<double2*>(lds + dst + E) = hw0;
        }
        if (NH > 1 && lane + 64 < HPJ) {
            *reinterpret_cast<double2*>(lds + dst + 1024) = hr1;
            *reinterpret_cast<double2*>(lds + dst + 1024 + E) = hw1;
        }
    }
    post();
    double acc[4][2];
    if (ablate_math) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            acc[k][0] = r[k].x + w[k].x;
            acc[k][1] = r[k].y + w[k].y;
        }
    } else {
#pragma unroll
        for (int c = 0; c < LT + 3; ++c) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int l = c - k;
                if (l >= 0 && l < LT) {
                    acc[k][0] = inv_pair_mac<FMA>(l == 0 ? 0.0 : acc[k][0], tl[l], r[c].x, th[l], w[c].x);
                    acc[k][1] = inv_pair_mac<FMA>(l == 0 ? 0.0 : acc[k][1], tl[l], r[c].y, th[l], w[c].y);
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) *reinterpret_cast<double2*>(lds + A.wb[k]) = make_double2(acc[k][0], acc[k][1]);
}

// Level 1 (stride one sample): lane owns the pairs lane + 64 kk, kk = 0..3, and reads LT/2 + 1 aligned pairs per operand.
template <int LT>
struct WvLevel1Addr {
    int ra[LT / 2 + 1];  // R operand q of the last 64-pair block (the only one whose window leaves the chunk)
    int d[LT / 2 + 1];   // its W operand = R operand + d[q]
};

template <int LT, int J>
__device__ __forceinline__ void wv_addr1_init(int lane, WvLevel1Addr<LT>& A) {
    constexpr int rbase = (J == 1) ? wv_slot_a() : 0;
    constexpr int D = wv_slot_w(J, 1) - rbase;
    constexpr int HR = kWvHeads;
    constexpr int E = wv_hoff(LT, J + 1) * 16;
#pragma unroll
    for (int q = 0; q < LT / 2 + 1; ++q) {
        const int p = 3 * 64 + lane + q;
        A.ra[q] = (p < kWvC) ? rbase + (p << 4) : HR + ((p - kWvC) << 4);
        A.d[q] = (p < kWvC) ? D : E;
    }
}

template <int LT, int J, bool FMA, typename Post>
__device__ __forceinline__ void wv_level1(char* lds, int lane, const WvLevel1Addr<LT>& A, const double (&tl)[LT],
                                          const double (&th)[LT], bool ablate_math, double (&out)[4][2], Post post) {
    constexpr int NW = LT / 2 + 1;
    constexpr int rbase = (J == 1) ? wv_slot_a() : 0;
    constexpr int D = wv_slot_w(J, 1) - rbase;
    constexpr int HR = kWvHeads;
    constexpr int E = wv_hoff(LT, J + 1) * 16;
    constexpr int HPJ = wv_hp(LT, 1);
    static_assert(HPJ <= 64, "level-1 heads: one pair per lane");
    asm volatile("" ::: "memory");  // (as in wv_level: other lanes' stores precede these loads)
    // two 64-pair blocks at a time
#pragma unroll
    for (int k0 = 0; k0 < 4; k0 += 2) {
        double2 rr[2][NW], ww[2][NW], hr, hw;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int q = 0; q < NW; ++q) {
                if (k0 + kk == 3) {
                    rr[kk][q] = *reinterpret_cast<const double2*>(lds + A.ra[q]);
                    ww[kk][q] = *reinterpret_cast<const double2*>(lds + A.ra[q] + A.d[q]);
                } else {
                    const int adr = rbase + (((k0 + kk) * 64 + lane + q) << 4);
                    rr[kk][q] = *reinterpret_cast<const double2*>(lds + adr);
                    ww[kk][q] = *reinterpret_cast<const double2*>(lds + adr + D);
                }
            }
        if (k0 == 2) {  // the old heads have been requested by now (they belong to the last block): take the new ones
            hr = *reinterpret_cast<const double2*>(lds + rbase + (lane << 4));
            hw = *reinterpret_cast<const double2*>(lds + rbase + D + (lane << 4));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if (k0 == 2) {
            if (lane < HPJ) {
                *reinterpret_cast<double2*>(lds + HR + (lane << 4)) = hr;
                *reinterpret_cast<double2*>(lds + HR + E + (lane << 4)) = hw;
            }
            post();
        }
        if (ablate_math) {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                out[k0 + kk][0] = rr[kk][0].x + ww[kk][0].x;
                out[k0 + kk][1] = rr[kk][0].y + ww[kk][0].y;
            }
        } else {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int l = 0; l < LT; ++l) {
                    const double r0 = (l & 1) ? rr[kk][l / 2].y : rr[kk][l / 2].x;
                    const double r1 = (l & 1) ? rr[kk][l / 2 + 1].x : rr[kk][l / 2].y;
                    const double w0 = (l & 1) ? ww[kk][l / 2].y : ww[kk][l / 2].x;
                    const double w1 = (l & 1) ? ww[kk][l / 2 + 1].x : ww[kk][l / 2].y;
                    out[k0 + kk][0] = inv_pair_mac<FMA>(l == 0 ? 0.0 : out[k0 + kk][0], tl[l], r0, th[l], w0);
                    out[k0 + kk][1] = inv_pair_mac<FMA>(l == 0 ? 0.0 : out[k0 + kk][1], tl[l], r1, th[l], w1);
                }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// The same four LDS-DMA instructions for a chunk that reaches beyond its row (the last, partial chunk of a row and the
// warm-up chunks behind it): the source wraps (PERIODIC) or is clamped to valid memory (ZERO_PADDING: the values are
// replaced by zeros once they have landed) lane by lane.  S: log2 stride of the reading level (-1: level 1).
template <int MODE>
__device__ __forceinline__ void wv_issue_edge(const double* __restrict__ band_row, int64_t t0, int64_t n, int S, int lane, unsigned ldst) {
    // (an opaque copy of the lane index: without it the compiler computes the pair indices of every level once per run and
    // keeps them in registers across the chunk loop, for a path that one chunk per row takes)
    asm volatile("" : "+v"(lane));
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int pair = blk_slot_rt(64 * i + lane, S);
        int64_t t = t0 + 2 * pair;
        if (t >= n) t = (MODE == kPeriodic) ? t - n : n - 2;
        glds16(band_row + t, ldst + 1024u * i);
    }
}

// ZERO_PADDING: zeros right of the signal in a freshly landed band chunk (slot at byte offset `slot`, layout of the level
// with log2 stride S)
__device__ __forceinline__ void wv_zero_tail(char* lds, int slot, int64_t t0, int64_t n, int S, int lane) {
    asm volatile("" : "+v"(lane));  // (as in wv_issue_edge)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int pair = blk_slot_rt(64 * i + lane, S);
        if (t0 + 2 * pair >= n) *reinterpret_cast<double2*>(lds + slot + ((64 * i + lane) << 4)) = make_double2(0.0, 0.0);
    }
}

// Thresholding of a freshly landed detail band chunk in place (TH = mode of apply_thr: 1 soft / a NaN coefficient becomes
// zero, 2 hard, 3 soft / a NaN coefficient stays).  The level's threshold is constant along the row, so every chunk of
// the band gets the same rule and the heads, which are copied out of the slot afterwards, hold thresholded values.  A
// finite threshold >= 0 takes the three-instruction forms of consume_thr, anything else the reference's rule as written.
template <int TH>
__device__ __forceinline__ void wv_threshold_slot(char* lds, int slot, int thr_addr, int lane) {
    const double thr = *reinterpret_cast<const double*>(lds + thr_addr);  // same address in every lane: a broadcast
    double2 v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = *reinterpret_cast<const double2*>(lds + slot + ((64 * i + lane) << 4));
    const bool plain = finite_f64(thr) && thr >= 0.0;
    if (__builtin_amdgcn_ballot_w64(plain) != 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[i].x = consume_thr<TH>(v[i].x, thr);
            v[i].y = consume_thr<TH>(v[i].y, thr);
        }
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[i].x = apply_thr(v[i].x, thr, TH);
            v[i].y = apply_thr(v[i].y, thr, TH);
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<double2*>(lds + slot + ((64 * i + lane) << 4)) = v[i];
}

// Levels 6 .. 1 of rows of even length n >= 1024, PERIODIC or ZERO_PADDING, both filters of LT taps.
// a.ntiles = runs per row (counted from the right end of the row), a.nchunks = chunks per run (>= a.ndry),
// a.tiles_per_row = chunks per row (the last one may be partial), a.ndry = warm-up chunks.
// Chunk c covers the samples [512 c, 512 c + 512) of the row extended by its boundary rule; a PERIODIC run at the right
// end of the row warms up on the chunks behind the row end, i.e. on the wrapped start of the row; a ZERO_PADDING run
// there needs no warm-up: right of the row everything is zero at every level, and the heads start as zeros.
// TH != 0: a denoise that only thresholds -- the detail bands are thresholded as they land (a.thr: threshold of row r,
// level j at thr[r * thr_row_stride + (j - 1) * thr_level_stride]).
template <int LT, int MODE, bool FMA, int TH = 0>
__global__ void __launch_bounds__(64, 2) k_inv_wave6(const InvArgs a) {
    constexpr int J = 6;
    constexpr int THR = wv_thr_off(LT, J);
    extern __shared__ double2 vw_smem[];
    char* lds = reinterpret_cast<char*>(vw_smem);
    const unsigned lds_base = static_cast<unsigned>(reinterpret_cast<uintptr_t>(lds));
    const int lane = threadIdx.x;

    const int64_t id = blockIdx.x;
    const int64_t row = id / a.ntiles;
    const int run = static_cast<int>(id - row * a.ntiles);
    const int64_t n = a.n;
    const int nch_row = a.tiles_per_row;
    const int c_hi = nch_row - 1 - run * a.nchunks;
    const int nreal = (c_hi + 1 < a.nchunks) ? c_hi + 1 : a.nchunks;
    const int ndry = (MODE == kZeroPad && run == 0) ? 0 : a.ndry;
    const int nsteps = ndry + nreal;
    const bool nomath = VW_ABLATED(a, 0);
    const bool noload = VW_ABLATED(a, 1);
    const int nt_load = a.hoff[0];  // bands are read once: load them with the streaming hint

    double tl[LT], th[LT];
#pragma unroll
    for (int i = 0; i < LT; ++i) {
        tl[i] = a.taps.lo[i];
        th[i] = a.taps.hi[i];
    }
    const double* __restrict__ arow = a.rin + row * a.rin_stride;
    const double* __restrict__ wrow = a.win + row * a.w_row_stride;
    double* __restrict__ orow = a.rout + row * a.rout_stride;
    auto wband = [&](int j) { return wrow + static_cast<int64_t>(j - 1) * a.w_band_stride; };
    // first sample of the chunk of step st (wrapped into the row when it lies behind the row end), and whether the chunk
    // reaches beyond the row end from there
    auto chunk_t0 = [&](int st) -> int64_t {
        const int c = (st < ndry) ? c_hi + ndry - st : c_hi - (st - ndry);
        int64_t t0 = static_cast<int64_t>(c) * kWvSamples;
        if (t0 >= n) t0 -= n;  // PERIODIC warm-up behind the row end (n >= 2 chunks: one wrap)
        return t0;
    };

    // the lane's swizzled byte offset inside a 1 KiB block, per reading level (level 1 and level 6 read unswizzled)
    const int vs0 = blk_slot<0>(lane) << 4, vs1 = blk_slot<1>(lane) << 4, vs2 = blk_slot<2>(lane) << 4, vs3 = blk_slot<3>(lane) << 4;
    const int vsid = lane << 4;
    WvLevelAddr<LT> A6, A5, A4, A3, A2;
    WvLevel1Addr<LT> A1;
    wv_addr_init<LT, J, 6>(lane, A6);
    wv_addr_init<LT, J, 5>(lane, A5);
    wv_addr_init<LT, J, 4>(lane, A4);
    wv_addr_init<LT, J, 3>(lane, A3);
    wv_addr_init<LT, J, 2>(lane, A2);
    wv_addr1_init<LT, J>(lane, A1);

    // heads start as zeros (ZERO_PADDING at the row end needs exactly that; elsewhere any finite value would do: the
    // warm-up chunks replace them before they matter)
    {
        constexpr int NHP = 2 * wv_hoff(LT, J + 1);
        for (int i = lane; i < NHP; i += 64) *reinterpret_cast<double2*>(lds + kWvHeads + (i << 4)) = make_double2(0.0, 0.0);
    }

    if (TH != 0) {  // the row's six thresholds, kept in LDS (read back per level as a broadcast)
        if (lane < J) *reinterpret_cast<double*>(lds + THR + 8 * lane) = a.thr[row * a.thr_row_stride + lane * a.thr_level_stride];
    }

    // one band chunk on its way: S = log2 stride of the level that reads it, voff = the lane's offset in that layout
    auto issue = [&](const double* band_row, int64_t t0, int S, int voff, int slot) {
        if (noload) return;
        if (t0 + kWvSamples > n)
            wv_issue_edge<MODE>(band_row, t0, n, S, lane, lds_base + slot);
        else if (nt_load == 1)
            wv_issue<1>(band_row, t0, voff, lds_base + slot);
#ifdef VW_TUNING
        else if (nt_load == 2)
            wv_issue<2>(band_row, t0, voff, lds_base + slot);
        else if (nt_load == 3)
            wv_issue<3>(band_row, t0, voff, lds_base + slot);
        else if (nt_load == 4)
            wv_issue<4>(band_row, t0, voff, lds_base + slot);
        else if (nt_load == 5)
            wv_issue<5>(band_row, t0, voff, lds_base + slot);
        else if (nt_load == 6)
            wv_issue<6>(band_row, t0, voff, lds_base + slot);
        else if (nt_load == 7)
            wv_issue<7>(band_row, t0, voff, lds_base + slot);
#endif
        else
            wv_issue<0>(band_row, t0, voff, lds_base + slot);
    };

    // prologue: the first four bands of the first chunk
    {
        const int64_t t0 = chunk_t0(0);
        issue(arow, t0, 4, vsid, wv_slot_a());
        issue(wband(6), t0, 4, vsid, wv_slot_w(J, 6));
        issue(wband(5), t0, 3, vs3, wv_slot_w(J, 5));
        issue(wband(4), t0, 2, vs2, wv_slot_w(J, 4));
    }

    int pst = 0;  // store instructions of the previous chunk that the memory counter may still hold
    for (int st = 0; st < nsteps; ++st) {
        const bool dry = st < ndry, last = (st + 1 == nsteps);
        const int64_t t0 = chunk_t0(st);
        const int64_t tn = last ? 0 : chunk_t0(st + 1);
        const bool pf = !last;
        const bool ztail = (MODE == kZeroPad) && (t0 + kWvSamples > n);  // zeros right of the signal

        wv_wait_vm(noload ? 0 : 8 + pst);  // a_6 and W_6 have landed (W_5, W_4 and the stores may be in flight)
        if (ztail) {
            wv_zero_tail(lds, wv_slot_a(), t0, n, 4, lane);
            wv_zero_tail(lds, wv_slot_w(J, 6), t0, n, 4, lane);
        }
        if (TH != 0) wv_threshold_slot<TH>(lds, wv_slot_w(J, 6), THR + 8 * 5, lane);
        wv_level<LT, J, 6, FMA>(lds, lane, A6, vsid, tl, th, nomath, [&] { issue(wband(3), t0, 1, vs1, wv_slot_w(J, 3)); });
        wv_wait_vm(noload ? 0 : 8 + pst);  // W_5
        if (ztail) wv_zero_tail(lds, wv_slot_w(J, 5), t0, n, 3, lane);
        if (TH != 0) wv_threshold_slot<TH>(lds, wv_slot_w(J, 5), THR + 8 * 4, lane);
        wv_level<LT, J, 5, FMA>(lds, lane, A5, vs3, tl, th, nomath, [&] { issue(wband(2), t0, 0, vs0, wv_slot_w(J, 2)); });
        wv_wait_vm(noload ? 0 : 8 + pst);  // W_4
        if (ztail) wv_zero_tail(lds, wv_slot_w(J, 4), t0, n, 2, lane);
        if (TH != 0) wv_threshold_slot<TH>(lds, wv_slot_w(J, 4), THR + 8 * 3, lane);
        wv_level<LT, J, 4, FMA>(lds, lane, A4, vs2, tl, th, nomath, [&] { issue(wband(1), t0, -1, vsid, wv_slot_w(J, 1)); });
        wv_wait_vm(noload ? 0 : 8);  // W_3
        if (ztail) wv_zero_tail(lds, wv_slot_w(J, 3), t0, n, 1, lane);
        if (TH != 0) wv_threshold_slot<TH>(lds, wv_slot_w(J, 3), THR + 8 * 2, lane);
        wv_level<LT, J, 3, FMA>(lds, lane, A3, vs1, tl, th, nomath, [&] { if (pf) issue(arow, tn, 4, vsid, wv_slot_a()); });
        wv_wait_vm(noload ? 0 : (last ? 4 : 8));  // W_2
        if (ztail) wv_zero_tail(lds, wv_slot_w(J, 2), t0, n, 0, lane);
        if (TH != 0) wv_threshold_slot<TH>(lds, wv_slot_w(J, 2), THR + 8 * 1, lane);
        wv_level<LT, J, 2, FMA>(lds, lane, A2, vs0, tl, th, nomath, [&] { if (pf) issue(wband(6), tn, 4, vsid, wv_slot_w(J, 6)); });
        wv_wait_vm(noload ? 0 : (last ? 0 : 8));  // W_1
        if (ztail) wv_zero_tail(lds, wv_slot_w(J, 1), t0, n, -1, lane);
        if (TH != 0) wv_threshold_slot<TH>(lds, wv_slot_w(J, 1), THR + 8 * 0, lane);
        double out[4][2];
        wv_level1<LT, J, FMA>(lds, lane, A1, tl, th, nomath, out, [&] {
            if (pf) {
                issue(wband(5), tn, 3, vs3, wv_slot_w(J, 5));
                issue(wband(4), tn, 2, vs2, wv_slot_w(J, 4));
            }
        });
        // The counted waits of the next chunk allow for the store instructions issued here, so their number must be
        // exact: a 64-pair block of the partial chunk that lies wholly outside the row issues NO instruction (the
        // compiler branches around a store whose lanes are all off), and a wait that allowed for it would let the
        // next chunk's levels start before their bands have landed.
        pst = 0;
        if (!dry && t0 < a.n_store) {
            double* __restrict__ cb = orow + t0;  // wave-uniform
            if (t0 + kWvSamples <= n) {
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) wv_store16(cb, static_cast<unsigned>(vsid + 1024 * kk), make_double2(out[kk][0], out[kk][1]), a.nt_store);
                pst = 4;
            } else {  // the last, partial chunk of the row
                const int rem = static_cast<int>(n - t0);  // samples of the chunk inside the row (< 512)
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)
                    if (2 * (kk * 64 + lane) < rem) wv_store16(cb, static_cast<unsigned>(vsid + 1024 * kk), make_double2(out[kk][0], out[kk][1]), a.nt_store);
                pst = (rem + 127) >> 7;  // blocks with at least one pair inside the row
            }
        }
    }
}

}  // namespace vw
