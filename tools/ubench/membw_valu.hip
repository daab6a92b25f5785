// Microbenchmark: what HBM rate does gfx950 hold while the vector units execute fp64 work beside the stream?
// Traffic shaped like the inverse MODWT (7 planes read, 1 written, 16 B per lane per access, next iteration's loads
// in flight while the current one is "computed"); OPS unfused fp64 instructions (half mul, half add) per lane and
// iteration on values that depend on the loaded data.  The inverse executes 384 per iteration of this shape.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#pragma clang fp contract(off)

template <int OPS>
__global__ void __launch_bounds__(256) k(const double2* __restrict__ in, double2* __restrict__ out, long plane, long n2, double a,
                                         double b) {
    const long stride = (long)gridDim.x * 256;
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    double2 v[7], nx[7];
#pragma unroll
    for (int p = 0; p < 7; ++p) v[p] = in[p * plane + i];
    for (; i < n2; i += stride) {
        const long j = i + stride < n2 ? i + stride : i;
#pragma unroll
        for (int p = 0; p < 7; ++p) nx[p] = in[p * plane + j];
        double acc[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) acc[q] = (q < 7 ? v[q].x : v[0].y);
#pragma unroll
        for (int o = 0; o < OPS / 16; ++o) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const double p = acc[q] * a;
                acc[q] = p + b;
            }
        }
        double s0 = 0, s1 = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            s0 += acc[q];
            s1 += acc[q + 4];
        }
#pragma unroll
        for (int p = 0; p < 7; ++p) s1 += v[p].y;
        __builtin_nontemporal_store(s0, &out[i].x);
        __builtin_nontemporal_store(s1, &out[i].y);
#pragma unroll
        for (int p = 0; p < 7; ++p) v[p] = nx[p];
    }
}

template <int OPS>
void run(const double2* in, double2* out, long plane, long n2, int wgs) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OPS>, dim3(wgs), dim3(256), 0, 0, in, out, plane, n2, 0.999999, 1e-9);
    hipDeviceSynchronize();
    const int reps = 5;
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k<OPS>, dim3(wgs), dim3(256), 0, 0, in, out, plane, n2, 0.999999, 1e-9);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    const double bytes = 8.0 * 16.0 * n2;
    printf("ops/iter %4d  wgs %6d  %.3f ms  %.2f TB/s  fp64 %.1f T instr-lanes/s\n", OPS, wgs, ms, bytes / ms / 1e9,
           (double)OPS * n2 / ms / 1e9);
    fflush(stdout);
}

int main(int argc, char** argv) {
    const long n2 = 1L << 29;  // double2 per plane: 8 GiB per plane, 68.7 GB per launch like the headline
    const long pad = argc > 1 ? atol(argv[1]) : 0;  // extra double2 between planes (breaks the power-of-two plane stride)
    const long plane = n2 + pad;
    double2 *in, *out;
    if (hipMalloc(&in, 7 * plane * 16) != hipSuccess || hipMalloc(&out, plane * 16) != hipSuccess) {
        printf("alloc failed\n");
        return 1;
    }
    hipMemset(in, 0x3c, 7 * plane * 16);
    printf("plane stride %ld bytes\n", plane * 16);
    // 256, 512, 1024, ... workgroups = 1, 2, 4, ... per CU: 28 KiB of loads in flight per workgroup
    for (int wgs : {256, 512, 768, 1024, 1536, 2048, 4096}) {
        run<0>(in, out, plane, n2, wgs);
        run<384>(in, out, plane, n2, wgs);
    }
    return 0;
}
