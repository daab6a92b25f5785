// Microbenchmark: HBM rate of the inverse's traffic (7 bands of a [B][7][N] block read by LDS-DMA, one row written) as a
// function of HOW a row is walked: a workgroup of WAVES waves walks one row from right to left, KB KiB per band and step,
// bands three ahead in a four-slot LDS ring, nothing computed.  Varies the granule (KB), the waves per workgroup and the
// workgroups per CU (through the LDS request).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>

__device__ __forceinline__ void glds16_so(const void* sbase, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
typedef double f64x2 __attribute__((ext_vector_type(2)));

template <int N> __device__ __forceinline__ void wait_vm();
#define WV(N) template <> __device__ __forceinline__ void wait_vm<N>() { asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); }
WV(0) WV(2) WV(3) WV(4) WV(5) WV(10) WV(6) WV(8) WV(12) WV(15) WV(16) WV(24) WV(30) WV(32) WV(48) WV(60)

// ORDER 0: right to left, 1: left to right.  BAR: barrier per band (multi-wave workgroups)
template <int WAVES, int KB, int ORDER, int AHEAD>
__global__ void __launch_bounds__(WAVES * 64) k(const double* __restrict__ in, double* __restrict__ out, long n, int rot) {
    extern __shared__ double2 smem[];
    constexpr int PER = KB / WAVES;  // DMA instructions per band and wave
    static_assert(PER >= 1 && PER * WAVES == KB, "");
    const unsigned lds_base = (unsigned)(uintptr_t)smem;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int lruns = rot >> 4;                      // log2 of the runs per row
    const bool skew = (rot >> 2) & 1, perm = (rot >> 3) & 1;
    const int rmode = rot & 3;
    long row = blockIdx.x >> lruns;
    const int run = (int)(blockIdx.x & ((1 << lruns) - 1));
    if (perm) row = (row * 40503L) % (gridDim.x >> lruns);   // odd multiplier: a permutation of the rows (65536 rows)
    const double* rowp = in + row * 7 * n;
    double* orow = out + row * n;
    const int steps_row = (int)(n * 8 / (KB * 1024));
    const int steps = steps_row >> lruns;
    const int base = run * steps;
    const int r0 = rmode ? (int)((rmode == 1 ? (row >> 3) * 5 + (row & 7) * 3 : (row >> 3)) % steps) : 0;
    constexpr int SLOT = KB * 1024;
    auto issue = [&](int q) {  // q-th band granule of the run (7 per step)
        const int st = q / 7, b = q % 7;
        int s = (st + r0 + (skew ? 5 * b : 0)) % steps;
        const int c = base + (ORDER ? s : steps - 1 - s);
        const double* src = rowp + (long)b * n + (long)c * (KB * 128);
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int blk = wave * PER + i;
            glds16_so(src, (unsigned)(blk * 1024 + lane * 16), lds_base + (q % (AHEAD + 1)) * SLOT + blk * 1024);
        }
    };
    const int total = steps * 7;
#pragma unroll
    for (int q = 0; q < AHEAD; ++q) issue(q);
    for (int q = 0; q < total; ++q) {
        if (q + AHEAD < total) {
            issue(q + AHEAD);
            wait_vm<AHEAD * PER>();
        } else {
            wait_vm<0>();
        }
        if (WAVES > 1) __builtin_amdgcn_s_barrier();
        if (q % 7 == 6) {
            const int st = q / 7;
            int s = (st + r0) % steps;
            const int c = base + (ORDER ? s : steps - 1 - s);
            double* dst = orow + (long)c * (KB * 128);
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                f64x2 v; v.x = (double)q; v.y = (double)lane;
                __builtin_nontemporal_store(v, reinterpret_cast<f64x2*>(dst + (wave * PER + i) * 128 + lane * 2));
            }
            // the stores count in vmcnt: let them drain before the counted waits go on (once per step)
            wait_vm<0>();
        }
    }
}

// The ring kernel's order: one (row, tile) per 256-lane workgroup, 16 KiB tile + HALO KiB halo per band, ids remapped so
// that an XCD walks consecutive (row, tile) ids; RUN > 1: a workgroup walks RUN tiles of its row right to left instead
// (halo only on the first).
__device__ __forceinline__ long xcd_remap(long bid, long nblk) {
    const long q = nblk >> 3, r = nblk & 7, x = bid & 7, kk = bid >> 3;
    const long base = (x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q;
    return base + kk;
}
template <int HALO, int RUN, int REMAP>
__global__ void __launch_bounds__(256) ktile(const double* __restrict__ in, double* __restrict__ out, long n) {
    extern __shared__ double2 smem[];
    const unsigned lds_base = (unsigned)(uintptr_t)smem;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int tiles = (int)(n / 2048), runs = tiles / RUN;
    const long id = REMAP ? xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x;
    const long row = id / runs;
    const int run = (int)(id - row * runs);
    const double* rowp = in + row * 7 * n;
    double* orow = out + row * n;
    constexpr int PER = (16 + HALO) / 4;
    constexpr int SLOT = (16 + HALO) * 1024;
    for (int t = 0; t < RUN; ++t) {
        const int tile = runs * RUN - 1 - (run * RUN + t);
        const int per = (t == 0) ? PER : 4;
        auto issue = [&](int b) {
            const double* src = rowp + (long)b * n + (long)tile * 2048;
            for (int i = 0; i < per; ++i) {
                const int blk = wave + 4 * i;
                const long off = blk * 1024 + lane * 16;
                const bool wrap = (long)tile * 16384 + blk * 1024 >= n * 8;  // wave-uniform: the halo block lies behind the row end
                glds16_so(wrap ? src - n : src, (unsigned)off, lds_base + (b % 3) * SLOT + blk * 1024);
            }
        };
        issue(0); issue(1);
        for (int b = 0; b < 7; ++b) {
            if (b + 2 < 7) issue(b + 2);
            if (b + 2 < 7) { if (per == 5) wait_vm<10>(); else wait_vm<8>(); }
            else if (b + 1 < 7) { if (per == 5) wait_vm<5>(); else wait_vm<4>(); }
            else wait_vm<0>();
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_s_barrier();
        }
        double* dst = orow + (long)tile * 2048;
        for (int i = 0; i < 4; ++i) {
            f64x2 v; v.x = (double)i; v.y = (double)lane;
            __builtin_nontemporal_store(v, reinterpret_cast<f64x2*>(dst + (wave + 4 * i) * 128 + lane * 2));
        }
    }
}
template <int HALO, int RUN, int REMAP>
void run_tile(const double* in, double* out, long rows, long n, size_t lds) {
    auto kern = ktile<HALO, RUN, REMAP>;
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const long nblk = rows * (n / 2048 / RUN);
    hipLaunchKernelGGL(kern, dim3(nblk), dim3(256), lds, 0, in, out, n);
    if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); exit(1); }
    const int reps = 3;
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(kern, dim3(nblk), dim3(256), lds, 0, in, out, n);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    printf("tile order: halo %d KiB  run %d  remap %d  lds %6zu (wg/cu %2d)  %.3f ms  %.2f TB/s (algorithmic)\n", HALO, RUN, REMAP, lds, (int)(163840 / lds), ms,
           64.0 * rows * n / ms / 1e9);
    fflush(stdout);
}

template <int WAVES, int KB, int ORDER, int AHEAD>
void run(const double* in, double* out, long rows, long n, size_t lds, int rot) {
    auto kern = k<WAVES, KB, ORDER, AHEAD>;
    if (lds < (size_t)(AHEAD + 1) * KB * 1024) lds = (size_t)(AHEAD + 1) * KB * 1024;
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const long grid = rows << (rot >> 4);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(WAVES * 64), lds, 0, in, out, n, rot);
    if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); exit(1); }
    const int reps = 3;
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(kern, dim3(grid), dim3(WAVES * 64), lds, 0, in, out, n, rot);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    printf("waves/wg %d  KiB/band/step %2d  ahead %d  order %d rot %d  lds %6zu (wg/cu %2d)  %.3f ms  %.2f TB/s\n", WAVES, KB, AHEAD, ORDER, rot, lds,
           (int)(163840 / lds), ms, 64.0 * rows * n / ms / 1e9);
    fflush(stdout);
}

int main(int argc, char** argv) {
    const long rows = 65536, n = argc > 1 ? atol(argv[1]) : 16384;
    double *in, *out;
    if (hipMalloc(&in, rows * 7 * n * 8) != hipSuccess || hipMalloc(&out, rows * n * 8) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(in, 0x3c, rows * 7 * n * 8);
    printf("n = %ld   (rot: bits 0-1 rotation, 4 band skew, 8 row permutation, >>4 log2 runs per row)\n", n);
    run<1, 4, 0, 3>(in, out, rows, n, 32768, 0);
    run<1, 4, 0, 3>(in, out, rows, n, 32768, 4);        // band skew (diagnostic: not what the transform needs)
    run<1, 4, 0, 3>(in, out, rows, n, 32768, 8);        // row permutation
    run<1, 4, 0, 3>(in, out, rows, n, 32768, 16);       // 2 runs per row
    run<1, 4, 0, 3>(in, out, rows, n, 32768, 32);       // 4 runs per row
    run<1, 4, 0, 3>(in, out, rows, n, 32768, 48);       // 8 runs per row
    run<1, 4, 0, 3>(in, out, rows, n, 32768, 64);       // 16 runs per row
    run<1, 4, 0, 2>(in, out, rows, n, 32768, 0);        // two bands ahead
    run<1, 4, 0, 6>(in, out, rows, n, 32768, 0);        // six bands ahead
    run<1, 4, 0, 6>(in, out, rows, n, 32768, 4);        // six ahead + skew
    return 0;
}
