// vw_launch_wave.hip -- instantiates and launches the wave-private streaming kernels (see vw_launch.h).
#include "vw_launch.h"
#include "vw_wave_kernels.hip.h"

namespace vw {

// one wavefront per run of 512-sample chunks (k_inv_wave6): levels 6..1, filters of 2, 4 or 8 taps, PERIODIC or
// ZERO_PADDING rows of even length
bool inv_wave_supported(int taps, int lev0, int lev1, int mode, int64_t n) {
    return (taps == 2 || taps == 4 || taps == 8) && lev0 == 1 && lev1 == 6 && (mode == kPeriodic || mode == kZeroPad) && n % 2 == 0 &&
           n >= 2 * kWvSamples;
}
int inv_wave_ndry(int taps, int lev1) {
    const int ctx = (taps - 1) * ((1 << lev1) - 1);
    return (ctx + kWvSamples - 1) / kWvSamples;
}
template <int LT, int MODE>
static hipError_t launch_inv_wave_th(const InvArgs& a, int64_t nblocks, size_t lds, hipStream_t stream) {
    if (a.thr_mode == 1) return launch_tile_nt(k_inv_wave6<LT, MODE, false, 1>, a, 64, nblocks, lds, stream);
    if (a.thr_mode == 2) return launch_tile_nt(k_inv_wave6<LT, MODE, false, 2>, a, 64, nblocks, lds, stream);
    return launch_tile_nt(k_inv_wave6<LT, MODE, false, 3>, a, 64, nblocks, lds, stream);
}
template <int LT>
static hipError_t launch_inv_wave_lt(const InvArgs& a, int mode, bool fma, int64_t nblocks, hipStream_t stream) {
    const size_t lds = wv_lds_bytes(LT, 6);
    if (a.thr_mode != 0)  // (the planner sends thresholding inverses here in the reference's arithmetic only)
        return mode == kPeriodic ? launch_inv_wave_th<LT, kPeriodic>(a, nblocks, lds, stream)
                                 : launch_inv_wave_th<LT, kZeroPad>(a, nblocks, lds, stream);
    if (mode == kPeriodic)
        return fma ? launch_tile_nt(k_inv_wave6<LT, kPeriodic, true>, a, 64, nblocks, lds, stream)
                   : launch_tile_nt(k_inv_wave6<LT, kPeriodic, false>, a, 64, nblocks, lds, stream);
    return fma ? launch_tile_nt(k_inv_wave6<LT, kZeroPad, true>, a, 64, nblocks, lds, stream)
               : launch_tile_nt(k_inv_wave6<LT, kZeroPad, false>, a, 64, nblocks, lds, stream);
}
hipError_t launch_inv_wave(const InvArgs& a, int mode, bool fma, int64_t nblocks, hipStream_t stream) {
    if (a.lh == 2) return launch_inv_wave_lt<2>(a, mode, fma, nblocks, stream);
    if (a.lh == 4) return launch_inv_wave_lt<4>(a, mode, fma, nblocks, stream);
    if (a.lh == 8) return launch_inv_wave_lt<8>(a, mode, fma, nblocks, stream);
    return hipErrorInvalidValue;
}

}  // namespace vw
