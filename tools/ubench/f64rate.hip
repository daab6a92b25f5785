// Microbenchmark: issue rate of v_mul_f64 + v_add_f64 (unfused) vs v_fma_f64 on gfx950.
#include <hip/hip_runtime.h>
#include <stdio.h>
#pragma clang fp contract(off)
template <int MODE>
__global__ void __launch_bounds__(256) k(double* out, int iters, double a, double b) {
    double x[8];
    for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 1e-3 + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 0) { double p = x[i] * a; x[i] = p + b; }          // mul + add
            else if (MODE == 1) x[i] = __builtin_fma(x[i], a, b);          // fma
            else if (MODE == 2) x[i] = x[i] + b;                           // add only
            else x[i] = x[i] * a;                                          // mul only
        }
    }
    double s = 0;
    for (int i = 0; i < 8; ++i) s += x[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE>
void run(const char* name, int wgs_per_cu) {
    double* d; int nb = 256 * wgs_per_cu; hipMalloc(&d, nb * 256 * 8);
    int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(nb), dim3(256), 0, 0, d, 100, 0.999999, 1e-9);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(nb), dim3(256), 0, 0, d, iters, 0.999999, 1e-9);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double ops_per_lane = (double)iters * 8 * (MODE == 0 ? 2 : 1);
    double lane_ops = ops_per_lane * nb * 256;
    printf("%-8s wg/cu=%d  %.3f ms  %.2f T lane-instr/s  (per CU per clk @2.4GHz: %.1f)\n", name, wgs_per_cu, ms,
           lane_ops / ms / 1e9, lane_ops / (ms * 1e-3) / 256 / 2.4e9);
    hipFree(d);
}
int main() {
    for (int w : {1, 2, 4}) { run<0>("mul+add", w); run<1>("fma", w); run<2>("add", w); run<3>("mul", w); }
    return 0;
}
