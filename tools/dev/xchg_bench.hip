// Cost of one Comm<2>::xchg between the two waves of a pair, in isolation: every workgroup of four waves runs two pairs
// that do `iters` exchanges of N doubles with `work` dependent fp64 FMAs in between (per half: work_up / work_lo, to
// model the daylight skew).  Prints cycles per exchange.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -o tools/dev/_ab/xchg_bench tools/dev/xchg_bench.hip
//   tools/dev/_ab/xchg_bench [iters] [work_up] [work_lo]
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../hydromodel_amd/csrc/hc_device.h"

using namespace hc;

template <int N>
__global__ __launch_bounds__(256, 1) void bench(int iters, int work_up, int work_lo, unsigned long long *cycles,
                                                 double *sink, unsigned long long *fault)
{
    __shared__ PairBox boxes[2];
    if (threadIdx.x < 4) boxes[threadIdx.x >> 1].seq[threadIdx.x & 1] = 0;
    __syncthreads();
    const int wave = threadIdx.x / WAVE, lane = threadIdx.x % WAVE;
    Comm<2> comm;
    comm.half = wave & 1;
    comm.lane = lane;
    comm.k = 0;
    comm.dead = 0;
    comm.fault = fault;
    comm.box = (Comm<2>::LdsBox *)(boxes + (wave >> 1));
    const int work = comm.half == 0 ? work_up : work_lo;
    double acc = 1.0 + lane * 1e-9;
    const unsigned long long t0 = clock64();
    for (int i = 0; i < iters; i++) {
        for (int w = 0; w < work; w++) acc = fma(acc, 0.999999, 1e-7);
        double mine[N], theirs[N];
#pragma unroll
        for (int j = 0; j < N; j++) mine[j] = uniform_d(acc) + j;
        comm.xchg(mine, theirs);
#pragma unroll
        for (int j = 0; j < N; j++) acc += 1e-12 * theirs[j];
    }
    const unsigned long long t1 = clock64();
    if (lane == 0) cycles[blockIdx.x * 4 + wave] = t1 - t0;
    sink[blockIdx.x * 256 + threadIdx.x] = acc;
}


typedef __attribute__((address_space(3))) PairBox LBox;
#define AS3 __attribute__((address_space(3)))

// variant 1: the shipped loop without the sleep; variant 2: poll the sequence number alone (scalar-steered), then
// fetch the payload; variant 3: like 2, payload and sequence number written by ONE ds_write_b128 per two values
template <int N, int V>
__device__ __forceinline__ void xchg_v(LBox *box, int half, int lane, unsigned &k, const double (&mine)[N], double (&theirs)[N])
{
    const unsigned p = k & 1u;
    if (lane == 0) {
        if (V == 4 || V == 5) {
            typedef double d2 __attribute__((ext_vector_type(2)));
            volatile AS3 d2 *out2 = (volatile AS3 d2 *)box->data[p][half];
#pragma unroll
            for (int j = 0; j < (N + 1) / 2; j++) {
                d2 w;
                w[0] = mine[2 * j];
                w[1] = 2 * j + 1 < N ? mine[2 * j + 1] : 0.0;
                out2[j] = w;
            }
        } else {
            volatile AS3 double *out = box->data[p][half];
#pragma unroll
            for (int j = 0; j < N; j++) out[j] = mine[j];
        }
        *(volatile AS3 unsigned *)&box->seq[half] = k + 1u;
    }
    const volatile AS3 unsigned *seq = &box->seq[half ^ 1];
    const volatile AS3 double *in = box->data[p][half ^ 1];
    const int want = __builtin_amdgcn_readfirstlane((int)k) + 1;
    if (V == 4 || V == 5) {
        // payload moved two doubles at a time (ds_write_b128 / ds_read_b128); 5: with a scalar spin bound
        typedef double d2 __attribute__((ext_vector_type(2)));
        constexpr int N2 = (N + 1) / 2;
        const volatile AS3 d2 *in2 = (const volatile AS3 d2 *)in;
        int spins = 0, gone = 0;
        for (;;) {
            const int got = __builtin_amdgcn_readfirstlane((int)*seq);
            d2 v[N2];
#pragma unroll
            for (int j = 0; j < N2; j++) v[j] = in2[j];
            if ((got >= want) | gone) {
#pragma unroll
                for (int j = 0; j < N; j++) theirs[j] = uniform_d(v[j / 2][j % 2]);
                break;
            }
            if (V == 5) {
                spins = __builtin_amdgcn_readfirstlane(spins + 1);
                gone = spins > (1 << 22);
            }
        }
        if (gone) { k = 0xdead0000u; return; }
    } else if (V == 1) {
        for (;;) {
            const int got = __builtin_amdgcn_readfirstlane((int)*seq);
            double v[N];
#pragma unroll
            for (int j = 0; j < N; j++) v[j] = in[j];
            if (got >= want) {
#pragma unroll
                for (int j = 0; j < N; j++) theirs[j] = uniform_d(v[j]);
                break;
            }
        }
    } else {
        for (;;) {
            const int got = __builtin_amdgcn_readfirstlane((int)*seq);
            if (got >= want) break;
            if (V == 3) __builtin_amdgcn_s_sleep(1);
        }
#pragma unroll
        for (int j = 0; j < N; j++) theirs[j] = uniform_d(in[j]);
    }
    k = (unsigned)want;
}

template <int N, int V>
__global__ __launch_bounds__(256, 1) void bench_v(int iters, unsigned long long *cycles, double *sink)
{
    __shared__ PairBox boxes[2];
    if (threadIdx.x < 4) boxes[threadIdx.x >> 1].seq[threadIdx.x & 1] = 0;
    __syncthreads();
    const int wave = threadIdx.x / WAVE, lane = threadIdx.x % WAVE;
    LBox *box = (LBox *)(boxes + (wave >> 1));
    const int half = wave & 1;
    unsigned k = 0;
    double acc = 1.0 + lane * 1e-9;
    const unsigned long long t0 = clock64();
    for (int i = 0; i < iters; i++) {
        double mine[N], theirs[N];
#pragma unroll
        for (int j = 0; j < N; j++) mine[j] = uniform_d(acc) + j;
        xchg_v<N, V>(box, half, lane, k, mine, theirs);
#pragma unroll
        for (int j = 0; j < N; j++) acc += 1e-12 * theirs[j];
    }
    const unsigned long long t1 = clock64();
    if (lane == 0) cycles[blockIdx.x * 4 + wave] = t1 - t0;
    sink[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int N, int V>
static void run_v(int iters)
{
    const int grid = 256;
    unsigned long long *cyc;
    double *sink;
    hipMalloc(&cyc, grid * 4 * 8);
    hipMalloc(&sink, grid * 256 * 8);
    hipLaunchKernelGGL((bench_v<N, V>), dim3(grid), dim3(256), 0, 0, iters, cyc, sink);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(grid * 4);
    hipMemcpy(h.data(), cyc, grid * 4 * 8, hipMemcpyDeviceToHost);
    double s = 0;
    for (auto v : h) s += (double)v;
    printf("variant %d N=%d: %.1f ticks per exchange\n", V, N, s / h.size() / iters);
    hipFree(cyc);
    hipFree(sink);
}

template <int N>
static void run(int iters, int wu, int wl)
{
    const int grid = 256;
    unsigned long long *cyc, *fault;
    double *sink;
    hipMalloc(&cyc, grid * 4 * 8);
    hipMalloc(&fault, 8);
    hipMemset(fault, 0, 8);
    hipMalloc(&sink, grid * 256 * 8);
    hipLaunchKernelGGL(bench<N>, dim3(grid), dim3(256), 0, 0, iters, wu, wl, cyc, sink, fault);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(grid * 4);
    hipMemcpy(h.data(), cyc, grid * 4 * 8, hipMemcpyDeviceToHost);
    unsigned long long f = 0;
    hipMemcpy(&f, fault, 8, hipMemcpyDeviceToHost);
    double s = 0;
    for (auto v : h) s += (double)v;
    printf("N=%d iters=%d work_up=%d work_lo=%d: %.1f clock64 ticks per iteration (timeouts %llu)\n", N, iters, wu, wl,
           s / h.size() / iters, f);
    hipFree(cyc);
    hipFree(fault);
    hipFree(sink);
}

int main(int argc, char **argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    const int wu = argc > 2 ? atoi(argv[2]) : 0;
    const int wl = argc > 3 ? atoi(argv[3]) : 0;
    run<1>(iters, wu, wl);
    run<4>(iters, wu, wl);
    run<8>(iters, wu, wl);
    if (wu == 0 && wl == 0) {
        run_v<1, 1>(iters); run_v<4, 1>(iters); run_v<8, 1>(iters);
        run_v<1, 2>(iters); run_v<4, 2>(iters); run_v<8, 2>(iters);
        run_v<1, 3>(iters); run_v<4, 3>(iters); run_v<8, 3>(iters);
        run_v<1, 4>(iters); run_v<4, 4>(iters); run_v<8, 4>(iters);
        run_v<1, 5>(iters); run_v<4, 5>(iters); run_v<8, 5>(iters);
    }
    return 0;
}
