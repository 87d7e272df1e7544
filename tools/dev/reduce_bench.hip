// Latency of one wave-wide fp64 sum, dependent chain: the DPP reduction the kernels use (wave_sum) against a reduction
// through two v_mfma_f64_16x16x4_f64 (ones as the B matrix).  One wave per SIMD, like the step kernel at D = 300.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -o tools/dev/_ab/reduce_bench tools/dev/reduce_bench.hip
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../hydromodel_amd/csrc/hc_device.h"

using namespace hc;
typedef double v4d __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double mfma_sum(double x)
{
    const v4d z = {0.0, 0.0, 0.0, 0.0};
    // D[i][j] = sum_k x[16 k + i]; lane l holds rows 4 (l / 16) + r, r = 0..3
    v4d d = __builtin_amdgcn_mfma_f64_16x16x4f64(x, 1.0, z, 0, 0, 0);
    const double p = (d[0] + d[1]) + (d[2] + d[3]);          // sum over the lanes whose (lane % 16) / 4 == lane / 16
    // A[i][k] = p of lane 16 k + i = the k-th quarter sum: D[i][j] = total, in every lane
    v4d t = __builtin_amdgcn_mfma_f64_16x16x4f64(p, 1.0, z, 0, 0, 0);
    return t[0];
}

template <int V>
__global__ __launch_bounds__(256, 1) void bench(int iters, unsigned long long *cycles, double *sink)
{
    const int lane = threadIdx.x % WAVE;
    double acc = 1.0 + lane * 1e-3;
    const unsigned long long t0 = clock64();
    for (int i = 0; i < iters; i++) {
        const double s = V == 0 ? wave_sum(acc) : mfma_sum(acc);
        acc = fma(s, 1e-9, acc);                     // the next sum depends on this one
    }
    const unsigned long long t1 = clock64();
    if (lane == 0) cycles[blockIdx.x * 4 + threadIdx.x / WAVE] = t1 - t0;
    sink[blockIdx.x * 256 + threadIdx.x] = acc;
}

__global__ void check(double *out)
{
    const int lane = threadIdx.x;
    const double x = 1.0 + lane * 0.5;
    out[lane] = wave_sum(x);
    out[64 + lane] = mfma_sum(x);
}

int main(int argc, char **argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    const int grid = 256;
    unsigned long long *cyc;
    double *sink, *chk;
    (void)hipMalloc(&cyc, grid * 4 * 8);
    (void)hipMalloc(&sink, grid * 256 * 8);
    (void)hipMalloc(&chk, 128 * 8);
    hipLaunchKernelGGL(check, dim3(1), dim3(64), 0, 0, chk);
    std::vector<double> c(128);
    (void)hipMemcpy(c.data(), chk, 128 * 8, hipMemcpyDeviceToHost);
    printf("sum of 1 + lane / 2: DPP %.1f (lane 0) %.1f (lane 63), MFMA %.1f (lane 0) %.1f (lane 37) %.1f (lane 63); exact %.1f\n", c[0], c[63],
           c[64], c[64 + 37], c[127], 64 + 0.5 * 63 * 64 / 2);
    for (int v = 0; v < 2; v++) {
        if (v == 0) hipLaunchKernelGGL(bench<0>, dim3(grid), dim3(256), 0, 0, iters, cyc, sink);
        else hipLaunchKernelGGL(bench<1>, dim3(grid), dim3(256), 0, 0, iters, cyc, sink);
        (void)hipDeviceSynchronize();
        std::vector<unsigned long long> h(grid * 4);
        (void)hipMemcpy(h.data(), cyc, grid * 4 * 8, hipMemcpyDeviceToHost);
        double s = 0;
        for (auto x : h) s += (double)x;
        printf("%s: %.1f cycles per dependent reduction (incl. one fma)\n", v == 0 ? "DPP wave_sum" : "two MFMA 16x16x4 f64", s / h.size() / iters);
    }
    return 0;
}
