// vw_launch_wave.hip -- instantiates and launches the wave-private streaming kernels (see vw_launch.h).
#include "vw_launch.h"
#include "vw_wave_kernels.hip.h"

namespace vw {

// one wavefront per run of 512-sample chunks (k_inv_wave6): levels 6..1, 8-tap filters, PERIODIC rows of a multiple of 512 samples
bool inv_wave_supported(int taps, int lev0, int lev1, int mode, int64_t n) {
    return taps == 8 && lev0 == 1 && lev1 == 6 && mode == kPeriodic && n % kWvSamples == 0 && n >= 2 * kWvSamples;
}
int inv_wave_ndry(int taps, int lev1) {
    const int ctx = (taps - 1) * ((1 << lev1) - 1);
    return (ctx + kWvSamples - 1) / kWvSamples;
}
hipError_t launch_inv_wave(const InvArgs& a, int mode, bool fma, int64_t nblocks, hipStream_t stream) {
    (void)mode;
    const size_t lds = wv_lds_bytes(8, 6);
    return fma ? launch_tile_nt(k_inv_wave6<8, true>, a, 64, nblocks, lds, stream)
               : launch_tile_nt(k_inv_wave6<8, false>, a, 64, nblocks, lds, stream);
}

}  // namespace vw
