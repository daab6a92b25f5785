// vw_launch.h -- tile shapes and the launchers of the tiled kernels.  The kernel templates are
// instantiated in vw_launch_fwd.hip and vw_launch_inv.hip (separate translation units so that they
// compile in parallel); the planner in vw_capi.hip only picks shapes and fills argument blocks.
#pragma once
#include "vw_kernels.hip.h"

namespace vw {

struct TileCfg {
    int ppt, hpt, nt;  // tile = ppt chunks of 2*nt samples, halo = hpt * 512 samples, nt lanes per workgroup
    int tile() const { return ppt * 2 * nt; }
    // forward shapes with a one-chunk halo run the tile-streaming kernel (one buffer + level histories);
    // everything else runs the one-tile-per-workgroup kernels (two buffers)
    size_t lds_bytes(bool inverse) const {
        const size_t m = static_cast<size_t>(tile()) + static_cast<size_t>(hpt) * kChunk;
        if (!inverse && hpt == 1) return (m + kChunk) * 8;
        return 2 * m * 8;
    }
};

struct SdryShape {
    int nt, ppt, hb;
    int tile() const { return ppt * 2 * nt; }
};
// shape 3 (inverse only): 1024 lanes, 4096-sample tiles -- R and W buffers plus both history sets fill the 160 KiB of
// a CU exactly, one workgroup of sixteen waves per CU instead of one of eight
constexpr SdryShape kSdry[4] = {{0, 0, 0}, {512, 2, 4}, {512, 4, 5}, {1024, 2, 4}};


// unroll: filters of exactly 8 + 8 taps take the fully unrolled tap loop
hipError_t launch_fwd(const FwdArgs& a, TileCfg cfg, int mode, bool vec2, bool fma, bool unroll, int64_t nblocks, hipStream_t stream);
hipError_t launch_inv(const InvArgs& a, TileCfg cfg, int mode, bool vec2, bool fma, int64_t nblocks, hipStream_t stream);
// register-blocked inverse (k_inv_blk): shape (4 tile chunks, 1 halo chunk) only, see inv_blk_supported
bool inv_blk_supported(int taps, int top_level);
hipError_t launch_inv_blk(const InvArgs& a, int mode, bool vec2, bool fma, int64_t nblocks, hipStream_t stream);
// the same with the bands fed by LDS-DMA (k_inv_ring): additionally needs 16-byte-aligned rows and n >= 2560
hipError_t launch_inv_ring(const InvArgs& a, int mode, bool fma, int64_t nblocks, hipStream_t stream);
// the ring kernel over runs of tiles with carried right contexts (k_inv_carry): a.ntiles runs per row of a.nchunks tiles,
// a.tiles_per_row tiles in all; inv_carry_supported tells whether the heads of this filter / level group fit the storage
bool inv_carry_supported(int taps, int lev0, int lev1);
hipError_t launch_inv_carry(const InvArgs& a, int mode, bool fma, int64_t nblocks, hipStream_t stream);
// one wavefront per run of chunks, no workgroup barriers (k_inv_wave6): a.ntiles runs per row of a.nchunks chunks of 512
// samples, a.tiles_per_row chunks in all, a.ndry warm-up chunks
bool inv_wave_supported(int taps, int lev0, int lev1, int mode, int64_t n);
int inv_wave_ndry(int taps, int lev1);
hipError_t launch_inv_wave(const InvArgs& a, int mode, bool fma, int64_t nblocks, hipStream_t stream);
// SYMMETRIC inverse on tiles (k_inv_sym): reads left of the output sample; ppt = 1 or 4 chunks per tile, one halo chunk.
// pairsum: MODWTTransform.inverse's arithmetic (lo*r + hi*w added as a pair, l < a.lh); otherwise the multi-level form
// (the two products added one after the other, per-filter lengths a.lh / a.lg).  a.pend[j] = first valid halo position of R_j.
bool inv_sym_supported(int lh, int lg, int lev0, int lev1, int64_t n);
hipError_t launch_inv_sym(const InvArgs& a, int ppt, bool vec2, bool pairsum, int64_t nblocks, hipStream_t stream);
hipError_t launch_fwd_sdry(const FwdArgs& a, int shape, int mode, bool vec2, bool fma, bool unroll, int64_t nblocks, hipStream_t stream);
// vec2: global access mode (0 one sample per access, 1 aligned pairs, 2 pairs at 8-byte alignment)
hipError_t launch_inv_sdry(const InvArgs& a, int shape, int mode, int vec2, bool fma, bool unroll, int64_t nblocks, hipStream_t stream);

// shared by the launcher units
template <typename K, typename A>
static hipError_t launch_tile_nt(K kernel, const A& args, int nt, int64_t nblocks, size_t lds, hipStream_t stream) {
    if (lds > 65536) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kernel, dim3(static_cast<unsigned>(nblocks)), dim3(nt), lds, stream, args);
    return hipGetLastError();
}
template <typename K, typename A>
static hipError_t launch_tile(K kernel, const A& args, int64_t nblocks, size_t lds, hipStream_t stream) {
    return launch_tile_nt(kernel, args, kThreads, nblocks, lds, stream);
}

}  // namespace vw
