// vw_launch_fwd.hip -- instantiates and launches the forward tile kernels (see vw_launch.h).
#include <stdlib.h>

#include "vw_launch.h"

namespace vw {

template <int P, int H, int MODE>
static hipError_t launch_fwd_pm(const FwdArgs& a, bool vec2, bool fma, int64_t nblocks, size_t lds, hipStream_t stream) {
    if (vec2) return fma ? launch_tile(k_fwd_fused<P, H, MODE, true, true>, a, nblocks, lds, stream)
                         : launch_tile(k_fwd_fused<P, H, MODE, true, false>, a, nblocks, lds, stream);
    return fma ? launch_tile(k_fwd_fused<P, H, MODE, false, true>, a, nblocks, lds, stream)
               : launch_tile(k_fwd_fused<P, H, MODE, false, false>, a, nblocks, lds, stream);
}
template <int NT, int P, int MODE, int LT>
static hipError_t launch_fwd_stream_lt(const FwdArgs& a, bool vec2, bool fma, int64_t nblocks, size_t lds, hipStream_t stream) {
    if (vec2) return fma ? launch_tile_nt(k_fwd_stream<NT, P, MODE, true, true, LT>, a, NT, nblocks, lds, stream)
                         : launch_tile_nt(k_fwd_stream<NT, P, MODE, true, false, LT>, a, NT, nblocks, lds, stream);
    return fma ? launch_tile_nt(k_fwd_stream<NT, P, MODE, false, true, 0>, a, NT, nblocks, lds, stream)
               : launch_tile_nt(k_fwd_stream<NT, P, MODE, false, false, 0>, a, NT, nblocks, lds, stream);
}
// filters of exactly 8 + 8 taps (db4, sym4, ...) on 16-byte-aligned rows take the fully unrolled tap loop
template <int NT, int P, int MODE>
static hipError_t launch_fwd_stream_pm(const FwdArgs& a, bool vec2, bool fma, bool unroll, int64_t nblocks, size_t lds, hipStream_t stream) {
    if (a.lh == 8 && a.lg == 8 && unroll)
        return launch_fwd_stream_lt<NT, P, MODE, 8>(a, vec2, fma, nblocks, lds, stream);
    return launch_fwd_stream_lt<NT, P, MODE, 0>(a, vec2, fma, nblocks, lds, stream);
}
#define VW_FWD_CASE(P, H)                                                                                    \
    if (cfg.ppt == P && cfg.hpt == H && cfg.nt == kThreads)                                                  \
        return mode == kPeriodic ? launch_fwd_pm<P, H, kPeriodic>(a, vec2, fma, nblocks, lds, stream)        \
                                 : launch_fwd_pm<P, H, kZeroPad>(a, vec2, fma, nblocks, lds, stream);
// (SYMMETRIC: the streaming kernel only -- its first tile of a row mirrors the halo level by level)
#define VW_FWD_STREAM_CASE(NT, P)                                                                            \
    if (cfg.ppt == P && cfg.hpt == 1 && cfg.nt == NT) {                                                      \
        if (mode == kSymmetric) return launch_fwd_stream_pm<NT, P, kSymmetric>(a, vec2, fma, unroll, nblocks, lds, stream); \
        return mode == kPeriodic ? launch_fwd_stream_pm<NT, P, kPeriodic>(a, vec2, fma, unroll, nblocks, lds, stream) \
                                 : launch_fwd_stream_pm<NT, P, kZeroPad>(a, vec2, fma, unroll, nblocks, lds, stream); \
    }

hipError_t launch_fwd(const FwdArgs& a, TileCfg cfg, int mode, bool vec2, bool fma, bool unroll, int64_t nblocks,
                      hipStream_t stream) {
    const size_t lds = cfg.lds_bytes(false);
    VW_FWD_STREAM_CASE(256, 1)
    VW_FWD_STREAM_CASE(256, 2)
    VW_FWD_STREAM_CASE(256, 4)
    if (mode == kSymmetric) return hipErrorInvalidValue;
    VW_FWD_CASE(4, 2)
    VW_FWD_CASE(8, 2)
    VW_FWD_CASE(8, 4)
    VW_FWD_CASE(8, 10)
    return hipErrorInvalidValue;
}

template <int NT, int P, int HB, int MODE>
static hipError_t launch_fwd_sdry_pm(const FwdArgs& a, bool vec2, bool fma, int64_t nblocks, size_t lds, hipStream_t stream) {
    if (vec2) return fma ? launch_tile_nt(k_fwd_sdry<NT, P, HB, MODE, 1, true>, a, NT, nblocks, lds, stream)
                         : launch_tile_nt(k_fwd_sdry<NT, P, HB, MODE, 1, false>, a, NT, nblocks, lds, stream);
    return fma ? launch_tile_nt(k_fwd_sdry<NT, P, HB, MODE, 2, true>, a, NT, nblocks, lds, stream)
               : launch_tile_nt(k_fwd_sdry<NT, P, HB, MODE, 2, false>, a, NT, nblocks, lds, stream);
}
template <int LT, int MODE>
static hipError_t launch_fwd_sdry_lt(const FwdArgs& a, bool vec2, int64_t nblocks, size_t lds, hipStream_t stream) {
    return vec2 ? launch_tile_nt(k_fwd_sdry<512, 2, 4, MODE, 1, false, LT>, a, 512, nblocks, lds, stream)
                : launch_tile_nt(k_fwd_sdry<512, 2, 4, MODE, 2, false, LT>, a, 512, nblocks, lds, stream);
}
#define VW_FWD_SDRY_LT(LT)                                                                          \
    if (a.lh == LT)                                                                                 \
        return mode == kPeriodic ? launch_fwd_sdry_lt<LT, kPeriodic>(a, vec2, nblocks, lds, stream) \
                                 : launch_fwd_sdry_lt<LT, kZeroPad>(a, vec2, nblocks, lds, stream);
hipError_t launch_fwd_sdry(const FwdArgs& a, int shape, int mode, bool vec2, bool fma, bool unroll, int64_t nblocks,
                           hipStream_t stream) {
    const SdryShape& sh = kSdry[shape];
    const size_t lds = static_cast<size_t>(sh.hb * kChunk + sh.tile() + a.hist_cap) * 8;
    // reference arithmetic, both filters of exactly 16 / 18 / 20 taps (sym8, coif3, db10 ...): fully unrolled tap loop
    if (shape == 1 && !fma && a.lh == a.lg && unroll) {
        VW_FWD_SDRY_LT(16)
        VW_FWD_SDRY_LT(18)
        VW_FWD_SDRY_LT(20)
    }
    if (shape == 1)
        return mode == kPeriodic ? launch_fwd_sdry_pm<512, 2, 4, kPeriodic>(a, vec2, fma, nblocks, lds, stream)
                                 : launch_fwd_sdry_pm<512, 2, 4, kZeroPad>(a, vec2, fma, nblocks, lds, stream);
    if (shape == 2)
        return mode == kPeriodic ? launch_fwd_sdry_pm<512, 4, 5, kPeriodic>(a, vec2, fma, nblocks, lds, stream)
                                 : launch_fwd_sdry_pm<512, 4, 5, kZeroPad>(a, vec2, fma, nblocks, lds, stream);
    return hipErrorInvalidValue;
}


}  // namespace vw
