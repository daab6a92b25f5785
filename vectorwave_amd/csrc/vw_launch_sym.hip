// vw_launch_sym.hip -- instantiates and launches the SYMMETRIC inverse tile kernel (see vw_launch.h).
#include "vw_launch.h"
#include "vw_sym_kernels.hip.h"

namespace vw {

bool inv_sym_supported(int lh, int lg, int lev0, int lev1, int64_t n) {
    if (n < kChunk || lev0 < 1 || lev1 < lev0) return false;
    const int lmax = lh > lg ? lh : lg;
    int64_t need = 0;
    for (int j = lev0; j <= lev1; ++j) need += static_cast<int64_t>(lmax - 1) * (static_cast<int64_t>(1) << (j - 1));
    return need + 2 <= kChunk;
}

template <int PPT, bool PAIRSUM, bool THR>
static hipError_t launch_sym_p(const InvArgs& a, bool vec2, int64_t nblocks, hipStream_t stream) {
    const size_t lds = 2 * static_cast<size_t>(kChunk + PPT * kChunk) * 8;
    // filters of exactly 8 + 8 taps on 16-byte-aligned rows take the fully unrolled tap loop
    if (vec2 && a.lh == 8 && a.lg == 8) return launch_tile(k_inv_sym<PPT, true, PAIRSUM, 8, THR>, a, nblocks, lds, stream);
    if (vec2) return launch_tile(k_inv_sym<PPT, true, PAIRSUM, 0, THR>, a, nblocks, lds, stream);
    return launch_tile(k_inv_sym<PPT, false, PAIRSUM, 0, THR>, a, nblocks, lds, stream);
}
template <int PPT>
static hipError_t launch_sym_ppt(const InvArgs& a, bool vec2, bool pairsum, int64_t nblocks, hipStream_t stream) {
    if (a.thr_mode != 0)
        return pairsum ? launch_sym_p<PPT, true, true>(a, vec2, nblocks, stream) : launch_sym_p<PPT, false, true>(a, vec2, nblocks, stream);
    return pairsum ? launch_sym_p<PPT, true, false>(a, vec2, nblocks, stream) : launch_sym_p<PPT, false, false>(a, vec2, nblocks, stream);
}

hipError_t launch_inv_sym(const InvArgs& a, int ppt, bool vec2, bool pairsum, int64_t nblocks, hipStream_t stream) {
    if (a.thr_mode != 0 && a.thr == nullptr) return hipErrorInvalidValue;
    if (ppt == 4) return launch_sym_ppt<4>(a, vec2, pairsum, nblocks, stream);
    if (ppt == 1) return launch_sym_ppt<1>(a, vec2, pairsum, nblocks, stream);
    return hipErrorInvalidValue;
}

}  // namespace vw
