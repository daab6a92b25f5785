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
hipError_t launch_inv(const InvArgs& a, TileCfg cfg, int mode, bool vec2, bool fma, int64_t nblocks,
                             hipStream_t stream) {
    const size_t lds = cfg.lds_bytes(true);
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


template <int NT, int P, int HB, int MODE>
static hipError_t launch_inv_sdry_pm(const InvArgs& a, bool vec2, bool fma, int64_t nblocks, size_t lds, hipStream_t stream) {
    if (vec2) return fma ? launch_tile_nt(k_inv_sdry<NT, P, HB, MODE, true, true>, a, NT, nblocks, lds, stream)
                         : launch_tile_nt(k_inv_sdry<NT, P, HB, MODE, true, false>, a, NT, nblocks, lds, stream);
    return fma ? launch_tile_nt(k_inv_sdry<NT, P, HB, MODE, false, true>, a, NT, nblocks, lds, stream)
               : launch_tile_nt(k_inv_sdry<NT, P, HB, MODE, false, false>, a, NT, nblocks, lds, stream);
}
hipError_t launch_inv_sdry(const InvArgs& a, int shape, int mode, bool vec2, bool fma, int64_t nblocks,
                                  hipStream_t stream) {
    const SdryShape& sh = kSdry[shape];
    const size_t lds = static_cast<size_t>(sh.hb * kChunk + sh.tile() + a.hist_cap) * 8 * 2;
    if (shape == 1)
        return mode == kPeriodic ? launch_inv_sdry_pm<512, 2, 4, kPeriodic>(a, vec2, fma, nblocks, lds, stream)
                                 : launch_inv_sdry_pm<512, 2, 4, kZeroPad>(a, vec2, fma, nblocks, lds, stream);
    if (shape == 2)
        return mode == kPeriodic ? launch_inv_sdry_pm<512, 4, 5, kPeriodic>(a, vec2, fma, nblocks, lds, stream)
                                 : launch_inv_sdry_pm<512, 4, 5, kZeroPad>(a, vec2, fma, nblocks, lds, stream);
    return hipErrorInvalidValue;
}



}  // namespace vw
