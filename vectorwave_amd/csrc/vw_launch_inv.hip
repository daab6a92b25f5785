// vw_launch_inv.hip -- instantiates and launches the inverse tile kernels (see vw_launch.h).
#include <stdlib.h>

#include "vw_launch.h"

namespace vw {

template <int P, int H, int MODE>
static hipError_t launch_inv_pm(const InvArgs& a, bool vec2, bool fma, int64_t nblocks, size_t lds, hipStream_t stream) {
    if (vec2) return fma ? launch_tile(k_inv_fused<P, H, MODE, true, true>, a, nblocks, lds, stream)
                         : launch_tile(k_inv_fused<P, H, MODE, true, false>, a, nblocks, lds, stream);
    return fma ? launch_tile(k_inv_fused<P, H, MODE, false, true>, a, nblocks, lds, stream)
               : launch_tile(k_inv_fused<P, H, MODE, false, false>, a, nblocks, lds, stream);
}
#define VW_INV_CASE(P, H)                                                                                    \
    if (cfg.ppt == P && cfg.hpt == H && cfg.nt == kThreads)                                                  \
        return mode == kPeriodic ? launch_inv_pm<P, H, kPeriodic>(a, vec2, fma, nblocks, lds, stream)        \
                                 : launch_inv_pm<P, H, kZeroPad>(a, vec2, fma, nblocks, lds, stream);
// SYMMETRIC: single-level launches with the mirrored right halo (reference arithmetic only), one halo chunk
#define VW_INV_SYM_CASE(P)                                                                                    \
    if (cfg.ppt == P && cfg.hpt == 1 && cfg.nt == kThreads)                                                   \
        return vec2 ? launch_tile(k_inv_fused<P, 1, kSymmetric, true, false>, a, nblocks, lds, stream)        \
                    : launch_tile(k_inv_fused<P, 1, kSymmetric, false, false>, a, nblocks, lds, stream);
hipError_t launch_inv(const InvArgs& a, TileCfg cfg, int mode, bool vec2, bool fma, int64_t nblocks,
                             hipStream_t stream) {
    const size_t lds = cfg.lds_bytes(true);
    if (mode == kSymmetric) {
        if (a.lev0 != 1 || a.lev1 != 1) return hipErrorInvalidValue;
        VW_INV_SYM_CASE(1)
        VW_INV_SYM_CASE(2)
        VW_INV_SYM_CASE(4)
        VW_INV_SYM_CASE(8)
        return hipErrorInvalidValue;
    }
    VW_INV_CASE(1, 1)
    VW_INV_CASE(2, 1)
    VW_INV_CASE(4, 1)
    VW_INV_CASE(8, 1)
    VW_INV_CASE(4, 2)
    VW_INV_CASE(8, 2)
    VW_INV_CASE(8, 4)
    VW_INV_CASE(8, 10)
    return hipErrorInvalidValue;
}


// register-blocked tile kernel: both reconstruction filters walked over exactly LT taps
template <int LT, int MODE>
static hipError_t launch_inv_blk_pm(const InvArgs& a, bool vec2, bool fma, int64_t nblocks, size_t lds, hipStream_t stream) {
    if (vec2) return fma ? launch_tile(k_inv_blk<LT, MODE, true, true>, a, nblocks, lds, stream)
                         : launch_tile(k_inv_blk<LT, MODE, true, false>, a, nblocks, lds, stream);
    return fma ? launch_tile(k_inv_blk<LT, MODE, false, true>, a, nblocks, lds, stream)
               : launch_tile(k_inv_blk<LT, MODE, false, false>, a, nblocks, lds, stream);
}
bool inv_blk_supported(int taps, int top_level) { return (taps == 2 || taps == 4 || taps == 8) && top_level <= 8; }
#define VW_INV_BLK_CASE(LT)                                                                                   \
    if (a.lh == LT)                                                                                           \
        return mode == kPeriodic ? launch_inv_blk_pm<LT, kPeriodic>(a, vec2, fma, nblocks, lds, stream)       \
                                 : launch_inv_blk_pm<LT, kZeroPad>(a, vec2, fma, nblocks, lds, stream);
hipError_t launch_inv_blk(const InvArgs& a, int mode, bool vec2, bool fma, int64_t nblocks, hipStream_t stream) {
    const size_t lds = 2 * static_cast<size_t>(kBlkWOff);
    VW_INV_BLK_CASE(2)
    VW_INV_BLK_CASE(4)
    VW_INV_BLK_CASE(8)
    return hipErrorInvalidValue;
}

// LDS-DMA ring form of the register-blocked kernel (16-byte-aligned rows only)
template <int LT, int MODE>
static hipError_t launch_inv_ring_pm(const InvArgs& a, bool fma, bool plain, int64_t nblocks, size_t lds, hipStream_t stream) {
    // a denoise that only thresholds (every band used, no finite check), reference arithmetic: the threshold is
    // applied as the operands are consumed (thr_mode 1, 2, 3 -> the kernel's TH)
    if (!fma && a.thr_mode != 0 && a.first_bad == nullptr && a.band_mask == 0xffffffffu) {
        if (a.thr_mode == 1) return launch_tile(k_inv_ring<LT, MODE, false, true, 1>, a, nblocks, lds, stream);
        if (a.thr_mode == 2) return launch_tile(k_inv_ring<LT, MODE, false, true, 2>, a, nblocks, lds, stream);
        return launch_tile(k_inv_ring<LT, MODE, false, true, 3>, a, nblocks, lds, stream);
    }
    if (plain) return fma ? launch_tile(k_inv_ring<LT, MODE, true, true>, a, nblocks, lds, stream)
                          : launch_tile(k_inv_ring<LT, MODE, false, true>, a, nblocks, lds, stream);
    return fma ? launch_tile(k_inv_ring<LT, MODE, true, false>, a, nblocks, lds, stream)
               : launch_tile(k_inv_ring<LT, MODE, false, false>, a, nblocks, lds, stream);
}
#define VW_INV_RING_CASE(LT)                                                                                  \
    if (a.lh == LT)                                                                                           \
        return mode == kPeriodic ? launch_inv_ring_pm<LT, kPeriodic>(a, fma, plain, nblocks, lds, stream)     \
                                 : launch_inv_ring_pm<LT, kZeroPad>(a, fma, plain, nblocks, lds, stream);
hipError_t launch_inv_ring(const InvArgs& a, int mode, bool fma, int64_t nblocks, hipStream_t stream) {
    size_t lds = 4 * static_cast<size_t>(kRingBand);
    if (VW_ABLATED(a, 3)) lds += 24576;  // tuning build: one workgroup per CU instead of two
    const bool plain = a.thr_mode == 0 && a.first_bad == nullptr && a.band_mask == 0xffffffffu;
    VW_INV_RING_CASE(2)
    VW_INV_RING_CASE(4)
    VW_INV_RING_CASE(8)
    return hipErrorInvalidValue;
}

// run form of the ring kernel (right contexts carried from tile to tile)
template <int LT, int MODE>
static hipError_t launch_inv_carry_pm(const InvArgs& a, bool fma, bool plain, int64_t nblocks, size_t lds, hipStream_t stream) {
    if (plain) return fma ? launch_tile(k_inv_carry<LT, MODE, true, true>, a, nblocks, lds, stream)
                          : launch_tile(k_inv_carry<LT, MODE, false, true>, a, nblocks, lds, stream);
    return fma ? launch_tile(k_inv_carry<LT, MODE, true, false>, a, nblocks, lds, stream)
               : launch_tile(k_inv_carry<LT, MODE, false, false>, a, nblocks, lds, stream);
}
// Head storage: the top level's two heads take one 128-pair area each, the heads of the levels below share the other
// two areas; below the top level no band of a run's first tile may reach into the storage behind pair 1152.
bool inv_carry_supported(int taps, int lev0, int lev1) {
    if (!inv_blk_supported(taps, lev1) || lev0 < 1 || lev1 < lev0) return false;
    auto hp = [&](int j) { return (((taps - 1) << (j - 1)) + 1) >> 1; };
    if (hp(lev1) > kCarryCtx) return false;
    int below = 0, ctx_below = 0;
    for (int j = lev0; j < lev1; ++j) {
        below += 2 * hp(j);
        ctx_below += (taps - 1) << (j - 1);
    }
    return below <= 2 * kCarryCtx && ((ctx_below + 1) >> 1) <= kCarryCtx;
}
#define VW_INV_CARRY_CASE(LT)                                                                                 \
    if (a.lh == LT)                                                                                           \
        return mode == kPeriodic ? launch_inv_carry_pm<LT, kPeriodic>(a, fma, plain, nblocks, lds, stream)    \
                                 : launch_inv_carry_pm<LT, kZeroPad>(a, fma, plain, nblocks, lds, stream);
hipError_t launch_inv_carry(const InvArgs& a, int mode, bool fma, int64_t nblocks, hipStream_t stream) {
    const size_t lds = 4 * static_cast<size_t>(kRingBand);
    const bool plain = a.thr_mode == 0 && a.first_bad == nullptr && a.band_mask == 0xffffffffu;
    VW_INV_CARRY_CASE(2)
    VW_INV_CARRY_CASE(4)
    VW_INV_CARRY_CASE(8)
    return hipErrorInvalidValue;
}

// vm: global access mode of the kernel (0 one sample, 1 aligned pairs, 2 pairs at 8-byte alignment)
template <int NT, int P, int HB, int MODE>
static hipError_t launch_inv_sdry_pm(const InvArgs& a, int vm, bool fma, int64_t nblocks, size_t lds, hipStream_t stream) {
    if (vm == 1) return fma ? launch_tile_nt(k_inv_sdry<NT, P, HB, MODE, 1, true>, a, NT, nblocks, lds, stream)
                            : launch_tile_nt(k_inv_sdry<NT, P, HB, MODE, 1, false>, a, NT, nblocks, lds, stream);
    if (vm == 2) return fma ? launch_tile_nt(k_inv_sdry<NT, P, HB, MODE, 2, true>, a, NT, nblocks, lds, stream)
                            : launch_tile_nt(k_inv_sdry<NT, P, HB, MODE, 2, false>, a, NT, nblocks, lds, stream);
    return fma ? launch_tile_nt(k_inv_sdry<NT, P, HB, MODE, 0, true>, a, NT, nblocks, lds, stream)
               : launch_tile_nt(k_inv_sdry<NT, P, HB, MODE, 0, false>, a, NT, nblocks, lds, stream);
}
template <int LT, int MODE>
static hipError_t launch_inv_sdry_lt(const InvArgs& a, int vm, int64_t nblocks, size_t lds, hipStream_t stream) {
    if (vm == 1) return launch_tile_nt(k_inv_sdry<1024, 2, 4, MODE, 1, false, LT>, a, 1024, nblocks, lds, stream);
    if (vm == 2) return launch_tile_nt(k_inv_sdry<1024, 2, 4, MODE, 2, false, LT>, a, 1024, nblocks, lds, stream);
    return launch_tile_nt(k_inv_sdry<1024, 2, 4, MODE, 0, false, LT>, a, 1024, nblocks, lds, stream);
}
#define VW_INV_SDRY_LT(LT)                                                                          \
    if (a.lh == LT)                                                                                 \
        return mode == kPeriodic ? launch_inv_sdry_lt<LT, kPeriodic>(a, vec2, nblocks, lds, stream) \
                                 : launch_inv_sdry_lt<LT, kZeroPad>(a, vec2, nblocks, lds, stream);
hipError_t launch_inv_sdry(const InvArgs& a, int shape, int mode, int vec2, bool fma, bool unroll, int64_t nblocks,
                           hipStream_t stream) {
    const SdryShape& sh = kSdry[shape];
    const size_t lds = static_cast<size_t>(sh.hb * kChunk + sh.tile() + a.hist_cap) * 8 * 2;
    if (shape == 1)
        return mode == kPeriodic ? launch_inv_sdry_pm<512, 2, 4, kPeriodic>(a, vec2, fma, nblocks, lds, stream)
                                 : launch_inv_sdry_pm<512, 2, 4, kZeroPad>(a, vec2, fma, nblocks, lds, stream);
    if (shape == 2)
        return mode == kPeriodic ? launch_inv_sdry_pm<512, 4, 5, kPeriodic>(a, vec2, fma, nblocks, lds, stream)
                                 : launch_inv_sdry_pm<512, 4, 5, kZeroPad>(a, vec2, fma, nblocks, lds, stream);
    if (shape == 3 && !fma && unroll) {
        VW_INV_SDRY_LT(16)
        VW_INV_SDRY_LT(18)
        VW_INV_SDRY_LT(20)
    }
    if (shape == 3)
        return mode == kPeriodic ? launch_inv_sdry_pm<1024, 2, 4, kPeriodic>(a, vec2, fma, nblocks, lds, stream)
                                 : launch_inv_sdry_pm<1024, 2, 4, kZeroPad>(a, vec2, fma, nblocks, lds, stream);
    return hipErrorInvalidValue;
}



}  // namespace vw
