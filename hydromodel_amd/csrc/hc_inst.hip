// hc_inst.hip -- the kernels of ONE cells-per-lane count and ONE cell model (compile with -DHC_INST_CPL=N, N = 2..10, and
// -DHC_INST_SPECIAL=1 | 0: default / generic exponents): the step kernel and the RHS hook, each for {monitoring, PREDICT
// lateral flow}.  See hc_launch.h.
#include <cstdlib>

#include "hc_launch.h"

#if !defined(HC_INST_CPL) && !defined(HC_INST_PAIR)
#error "compile with -DHC_INST_CPL=<cells per lane> -DHC_INST_SPECIAL=<0|1> or -DHC_INST_PAIR"
#endif
#if defined(HC_INST_CPL) && !defined(HC_INST_SPECIAL)
#error "compile with -DHC_INST_SPECIAL=1 (default exponents) or 0 (generic exponents) next to -DHC_INST_CPL"
#endif

#ifdef HC_INST_PAIR
// ---- the split-column step kernel (two waves per member): its own translation unit
namespace hc {
namespace {
template <bool SPECIAL, bool PREDICT>
hipError_t step_pair_one(const LaunchCfg &cfg, const StepArgs &A)
{
    constexpr int WPB = wpb_of(PAIR_CPL, 2);
    // (a handle with several parameter points takes the build with the chunk scheduler compiled in)
    auto kern = A.n_points > 1 ? step_kernel<PAIR_CPL, SPECIAL, WPB, PREDICT, 2, true> : step_kernel<PAIR_CPL, SPECIAL, WPB, PREDICT, 2, false>;
    const size_t lds = step_lds_bytes(PAIR_CPL, WPB, 2);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(cfg.grid), dim3(WPB * WAVE), lds, cfg.stream, A);
    return hipGetLastError();
}
}  // namespace

// RHS hook on the split-column path: two waves per member, the same rhs_eval<..., Comm<2>> the step kernel inlines
template <bool SPECIAL, bool PREDICT>
__global__ __launch_bounds__(wpb_of(PAIR_CPL, 2) * WAVE, 1) void rhs_pair_kernel(const StepArgs A, long long row, double *dydt)
{
    constexpr int CPL = PAIR_CPL, SLOTS = WAVE * CPL, TSLOTS = 2 * SLOTS, WPB = wpb_of(PAIR_CPL, 2);
    extern __shared__ double lds[];
    double *tab = lds;
    PairBox *boxes = reinterpret_cast<PairBox *>(tab + NTAB * TSLOTS);
    for (int k = threadIdx.x; k < NTAB * TSLOTS; k += WPB * WAVE) tab[k] = A.tab[k];
    if (threadIdx.x < 2 * (WPB / 2)) boxes[threadIdx.x >> 1].seq[threadIdx.x & 1] = 0;
    __syncthreads();
    const int wave = threadIdx.x / WAVE, lane = threadIdx.x % WAVE;
    const long long member = (long long)blockIdx.x * (WPB / 2) + (wave >> 1);
    if (member >= A.n_members) return;                     // (both waves of the pair leave together)
    const ColumnDev P = load_const(A.P);
    const IoArgs io = load_const(A.io);
    const int D = P.D;
    Comm<2> comm;
    comm.half = wave & 1;
    comm.lane = lane;
    comm.k = 0;
    comm.dead = 0;
    comm.fault = io.counters + 5;
    comm.box = (Comm<2>::LdsBox *)(boxes + (wave >> 1));
    const int hb = comm.half * SLOTS;
    RowDev R;
    R.precip = io.precip[row];
    R.atm = io.atm[row];
    R.daylight = io.daylight[row] & 1;
    R.wet = (io.daylight[row] >> 1) & 1;
    R.wtd_obs = io.wtd_obs[row];
    R.spinup = A.spinup;
    R.diag = 0;
    double y[CPL], rnd[CPL], f[CPL];
    double dtr = 0.0, dlf = 0.0;
#ifdef HC_PROFILE
    unsigned long long prof_dummy[96], prof_t = 0;
    unsigned long long *prof_lds = prof_dummy;
    int prof_slot = 0;
#endif
#pragma unroll
    for (int c = 0; c < CPL; c++) {
        const int i = hb + lane * CPL + c;
        y[c] = i < D ? io.psi[member * D + i] : 0.0;
        int idx = i >= 1 ? i - 1 : 0;                       // cell i reads n_rnd[max(i - 1, 0)]; the top-node cell n_rnd[0]
        idx = (i < D - 1) ? idx : 0;
        double z;
        if (A.host_noise)
            z = io.base_noise[member * D + idx];
        else
            z = philox_normal(io.seed, (unsigned long long)(io.member_offset + member), 0u, (unsigned)idx) * io.nscale[member];
        rnd[c] = tab[T_NOISEC * TSLOTS + hb + c * WAVE + lane] * z;
    }
    rhs_eval<CPL, SPECIAL, PREDICT>(P, R, tab + hb, lane, y, rnd, f, nullptr, dtr, dlf, comm HC_RHS_PROF_ARG);
#pragma unroll
    for (int c = 0; c < CPL; c++) {
        const int i = hb + lane * CPL + c;
        if (i < D) dydt[member * D + i] = f[c];
    }
}

namespace {
template <bool SPECIAL, bool PREDICT>
hipError_t rhs_pair_one(const LaunchCfg &cfg, const StepArgs &A, long long row, double *dydt)
{
    auto kern = rhs_pair_kernel<SPECIAL, PREDICT>;
    constexpr int WPB = wpb_of(PAIR_CPL, 2);
    const size_t lds = (size_t)NTAB * 2 * WAVE * PAIR_CPL * 8 + (WPB / 2) * sizeof(PairBox);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(cfg.grid), dim3(WPB * WAVE), lds, cfg.stream, A, row, dydt);
    return hipGetLastError();
}
}  // namespace

hipError_t launch_rhs_pair(const LaunchCfg &cfg, const StepArgs &A, long long row, double *dydt)
{
    if (cfg.special) return cfg.predict ? rhs_pair_one<true, true>(cfg, A, row, dydt) : rhs_pair_one<true, false>(cfg, A, row, dydt);
    return cfg.predict ? rhs_pair_one<false, true>(cfg, A, row, dydt) : rhs_pair_one<false, false>(cfg, A, row, dydt);
}

hipError_t launch_step_pair(const LaunchCfg &cfg, const StepArgs &A)
{
    if (cfg.special) return cfg.predict ? step_pair_one<true, true>(cfg, A) : step_pair_one<true, false>(cfg, A);
    return cfg.predict ? step_pair_one<false, true>(cfg, A) : step_pair_one<false, false>(cfg, A);
}
}  // namespace hc
#else
namespace hc {

// RHS hook: one wave per member, same device path as the stepper
template <int CPL, bool SPECIAL, int WPB, bool PREDICT>
__global__ __launch_bounds__(WPB *WAVE, 1) void rhs_kernel(const StepArgs A, long long row, double *dydt,
                                                            double *aux)
{
    constexpr int SLOTS = WAVE * CPL;
    extern __shared__ double lds[];
    double *tab = lds;
    double *nzbase = tab + NTAB * SLOTS;
    for (int k = threadIdx.x; k < NTAB * SLOTS; k += WPB * WAVE) tab[k] = A.tab[k];
    __syncthreads();
    const int wave = threadIdx.x / WAVE, lane = threadIdx.x % WAVE;
    const long long member = (long long)blockIdx.x * WPB + wave;
    if (member >= A.n_members) return;
    double *nz = nzbase + wave * SLOTS;
    const ColumnDev P = load_const(A.P);
    const IoArgs io = load_const(A.io);
    const int D = P.D;
    RowDev R;
    R.precip = io.precip[row];
    R.atm = io.atm[row];
    R.daylight = io.daylight[row] & 1;
    R.wet = (io.daylight[row] >> 1) & 1;
    R.wtd_obs = io.wtd_obs[row];
    R.spinup = A.spinup;
    R.diag = 0;
    double y[CPL], rnd[CPL], f[CPL];
    double dtr = 0.0, dlf = 0.0;
    Comm<1> comm;
    comm.half = 0;
#ifdef HC_PROFILE
    unsigned long long prof_dummy[96], prof_t = 0;     // rhs_eval's region stamps go nowhere in the hook (32 cycle sums + 64 entry counts)
    unsigned long long *prof_lds = prof_dummy;
    int prof_slot = 0;
#endif
#pragma unroll
    for (int c = 0; c < CPL; c++) {
        const int i = lane * CPL + c;
        y[c] = i < D ? io.psi[member * D + i] : 0.0;
        double z = 0.0;
        if (i < D) {
            if (A.host_noise)
                z = io.base_noise[member * D + i];
            else
                z = philox_normal(io.seed, (unsigned long long)(io.member_offset + member), 0u, (unsigned)i) *
                    io.nscale[member];
        }
        nz[c * WAVE + lane] = z;
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int c = 0; c < CPL; c++) {
        const int i = lane * CPL + c;
        int idx = i >= 1 ? i - 1 : 0;
        idx = (i < D - 1) ? idx : 0;
        rnd[c] = tab[T_NOISEC * SLOTS + c * WAVE + lane] * nz[(idx % CPL) * WAVE + idx / CPL];
    }
#ifdef HC_PROFILE
    // diagnostic build: repeat the evaluation (loop-carried through y) to time the RHS alone
    for (long long rep = 1; rep < A.n_rows; rep++) {
        rhs_eval<CPL, SPECIAL, PREDICT>(P, R, tab, lane, y, rnd, f, nullptr, dtr, dlf, comm HC_RHS_PROF_ARG);
#pragma unroll
        for (int c = 0; c < CPL; c++) y[c] = fma(f[c], 1e-300, y[c]);
    }
#endif
    rhs_eval<CPL, SPECIAL, PREDICT>(P, R, tab, lane, y, rnd, f, aux ? aux + member * (3 * (D - 1) + 1) : nullptr, dtr, dlf, comm HC_RHS_PROF_ARG);
#pragma unroll
    for (int c = 0; c < CPL; c++) {
        const int i = lane * CPL + c;
        if (i < D) dydt[member * D + i] = f[c];
    }
}


namespace {

template <int CPL, bool SPECIAL, bool PREDICT>
hipError_t step_one(const LaunchCfg &cfg, const StepArgs &A)
{
    constexpr int WPB = wpb_of(CPL, 1, SPECIAL);
    auto kern = step_kernel<CPL, SPECIAL, WPB, PREDICT>;
    const size_t lds = step_lds_bytes(CPL, WPB, 1, SPECIAL);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(cfg.grid), dim3(WPB * WAVE), lds, cfg.stream, A);
    return hipGetLastError();
}

template <int CPL, bool SPECIAL, bool PREDICT>
hipError_t rhs_one(const LaunchCfg &cfg, const StepArgs &A, long long row, double *dydt, double *aux)
{
    constexpr int WPB = wpb_of(CPL, 1, SPECIAL);
    auto kern = rhs_kernel<CPL, SPECIAL, WPB, PREDICT>;
#ifdef HC_PROFILE
    // diagnostic build: the step kernel's occupancy (one workgroup per CU) unless HYDROCOL_RHS_LDS_KB says otherwise
    size_t lds = step_lds_bytes(CPL, WPB, 1, SPECIAL);
    if (const char *e = getenv("HYDROCOL_RHS_LDS_KB")) lds = std::max(rhs_lds_bytes(CPL, WPB), (size_t)atoll(e) * 1024);
#else
    const size_t lds = rhs_lds_bytes(CPL, WPB);
#endif
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(cfg.grid), dim3(WPB * WAVE), lds, cfg.stream, A, row, dydt, aux);
    return hipGetLastError();
}

}  // namespace

template <>
hipError_t launch_step_part<HC_INST_CPL, HC_INST_SPECIAL != 0>(const LaunchCfg &cfg, const StepArgs &A)
{
    constexpr bool S = HC_INST_SPECIAL != 0;
    return cfg.predict ? step_one<HC_INST_CPL, S, true>(cfg, A) : step_one<HC_INST_CPL, S, false>(cfg, A);
}

template <>
hipError_t launch_rhs_part<HC_INST_CPL, HC_INST_SPECIAL != 0>(const LaunchCfg &cfg, const StepArgs &A, long long row, double *dydt,
                                                           double *aux)
{
    constexpr bool S = HC_INST_SPECIAL != 0;
    return cfg.predict ? rhs_one<HC_INST_CPL, S, true>(cfg, A, row, dydt, aux) : rhs_one<HC_INST_CPL, S, false>(cfg, A, row, dydt, aux);
}

}  // namespace hc
#endif  // HC_INST_PAIR
