// pmc_calib.hip -- what FETCH_SIZE / WRITE_SIZE report for the access shapes of the step kernel, on KNOWN byte counts.
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/dev/_ab/pmc_calib tools/pmc_calib.hip
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d out/fetch -- tools/dev/_ab/pmc_calib
//   rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d out/write -- tools/dev/_ab/pmc_calib
//   (plain run: prints the bytes each kernel moved and its time; tools/pmc_constants.py --calib joins the two)
//
// Kernels (one launch each, a name of its own so that the counter CSV separates them):
//   cal_buf_store    raw_buffer_store_b64, 512 contiguous bytes per wave instruction, every byte written once  (1 GiB)
//   cal_buf_load     raw_buffer_load_b64, same shape, every byte read once                                      (1 GiB)
//   cal_psi_store    global_store_dwordx2 in the state's own layout: lane l writes doubles [m*300 + 5 l + c], c = 0..4
//   cal_psi_load     global_load_dwordx2, same layout
//   cal_cycle_<KB>   the TWO layout's pattern: each of 2 048 waves owns <KB> KiB and, `passes` times, stores it whole and
//                    loads it whole (512 B per instruction) -- footprint 2 048 x <KB> KiB: 32 MiB (the L2s), 96 MiB (the
//                    step kernel's working set), 192 MiB, 384 MiB and 768 MiB (beyond the 256 MiB Infinity Cache)
// Every kernel runs on the step kernel's launch shape: 256 workgroups x 512 threads.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                                   \
    do {                                                                                           \
        hipError_t e_ = (x);                                                                       \
        if (e_ != hipSuccess) {                                                                    \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));              \
            exit(1);                                                                               \
        }                                                                                          \
    } while (0)

typedef unsigned v2u_t __attribute__((ext_vector_type(2)));
constexpr int WG = 256, THREADS = 512, WAVES = WG * THREADS / 64;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(double *base, unsigned bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(base, 0, bytes, 0x00020000);
}

// every wave streams its own contiguous slice of `bytes_per_wave`, 512 B per instruction
__global__ __launch_bounds__(THREADS) void cal_buf_store(double *buf, unsigned bytes_per_wave)
{
    const int wave = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * (THREADS / 64) + threadIdx.x / 64)), lane = threadIdx.x % 64;
    const __amdgpu_buffer_rsrc_t r = rsrc_of(buf + (size_t)wave * (bytes_per_wave / 8), bytes_per_wave);
    v2u_t v;
    v.x = wave;
    v.y = lane;
    for (unsigned off = 0; off < bytes_per_wave; off += 512) __builtin_amdgcn_raw_buffer_store_b64(v, r, lane * 8, off, 0);
}
__global__ __launch_bounds__(THREADS) void cal_buf_load(double *buf, unsigned bytes_per_wave, unsigned *sink)
{
    const int wave = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * (THREADS / 64) + threadIdx.x / 64)), lane = threadIdx.x % 64;
    const __amdgpu_buffer_rsrc_t r = rsrc_of(buf + (size_t)wave * (bytes_per_wave / 8), bytes_per_wave);
    unsigned acc = 0;
    for (unsigned off = 0; off < bytes_per_wave; off += 512) {
        const v2u_t v = __builtin_amdgcn_raw_buffer_load_b64(r, lane * 8, off, 0);
        acc += v.x ^ v.y;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
// the state's layout: member m, node 5 l + c  (D = 300: lanes 0..59)
__global__ __launch_bounds__(THREADS) void cal_psi_store(double *psi, long long members)
{
    const long long wave = (long long)blockIdx.x * (THREADS / 64) + threadIdx.x / 64;
    const int lane = threadIdx.x % 64;
    for (long long m = wave; m < members; m += WAVES)
        if (lane < 60)
#pragma unroll
            for (int c = 0; c < 5; c++) psi[m * 300 + lane * 5 + c] = (double)c;
}
__global__ __launch_bounds__(THREADS) void cal_psi_load(const double *psi, long long members, double *sink)
{
    const long long wave = (long long)blockIdx.x * (THREADS / 64) + threadIdx.x / 64;
    const int lane = threadIdx.x % 64;
    double acc = 0.0;
    for (long long m = wave; m < members; m += WAVES)
        if (lane < 60)
#pragma unroll
            for (int c = 0; c < 5; c++) acc += psi[m * 300 + lane * 5 + c];
    if (acc == 0.123) sink[0] = acc;
}
// the TWO layout's pattern: a per-wave region stored whole, then loaded whole, `passes` times
template <int KB>
__global__ __launch_bounds__(THREADS) void cal_cycle(double *buf, int passes, unsigned *sink)
{
    const int wave = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * (THREADS / 64) + threadIdx.x / 64)), lane = threadIdx.x % 64;
    constexpr unsigned BYTES = KB * 1024u;
    const __amdgpu_buffer_rsrc_t r = rsrc_of(buf + (size_t)wave * (BYTES / 8), BYTES);
    unsigned acc = 0;
    for (int p = 0; p < passes; p++) {
        v2u_t v;
        v.x = p;
        v.y = lane;
        for (unsigned off = 0; off < BYTES; off += 512) __builtin_amdgcn_raw_buffer_store_b64(v, r, lane * 8, off, 0);
        for (unsigned off = 0; off < BYTES; off += 512) {
            const v2u_t w = __builtin_amdgcn_raw_buffer_load_b64(r, lane * 8, off, 0);
            acc += w.x ^ w.y;
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

template <class F>
static float timed(F &&launch)
{
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    CHECK(hipEventRecord(a));
    launch();
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    CHECK(hipGetLastError());
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, a, b));
    return ms;
}

int main()
{
    const size_t GIB = 1ull << 30;
    double *buf;
    unsigned *sink;
    CHECK(hipMalloc((void **)&buf, 2 * GIB));
    CHECK(hipMalloc((void **)&sink, 64));
    CHECK(hipMemset(buf, 0, 2 * GIB));
    CHECK(hipDeviceSynchronize());
    const unsigned per_wave = (unsigned)(GIB / WAVES);
    float ms;
    printf("kernel,bytes_stored,bytes_loaded,footprint_bytes,ms\n");
    ms = timed([&] { cal_buf_store<<<WG, THREADS>>>(buf, per_wave); });
    printf("cal_buf_store,%zu,0,%zu,%.3f\n", GIB, GIB, ms);
    ms = timed([&] { cal_buf_load<<<WG, THREADS>>>(buf, per_wave, sink); });
    printf("cal_buf_load,0,%zu,%zu,%.3f\n", GIB, GIB, ms);
    const long long members = 262144;
    ms = timed([&] { cal_psi_store<<<WG, THREADS>>>(buf, members); });
    printf("cal_psi_store,%lld,0,%lld,%.3f\n", members * 2400, members * 2400, ms);
    ms = timed([&] { cal_psi_load<<<WG, THREADS>>>(buf, members, (double *)sink); });
    printf("cal_psi_load,0,%lld,%lld,%.3f\n", members * 2400, members * 2400, ms);
#define CYCLE(KB, PASSES)                                                                                              \
    ms = timed([&] { cal_cycle<KB><<<WG, THREADS>>>(buf, PASSES, sink); });                                            \
    printf("cal_cycle_%d,%zu,%zu,%zu,%.3f\n", KB, (size_t)WAVES * KB * 1024 * PASSES, (size_t)WAVES * KB * 1024 * PASSES, \
           (size_t)WAVES * KB * 1024, ms);
    CYCLE(16, 768)
    CYCLE(48, 256)
    CYCLE(96, 128)
    CYCLE(192, 64)
    CYCLE(384, 32)
    CHECK(hipFree(buf));
    CHECK(hipFree(sink));
    return 0;
}
