// hc_launch.h -- seam between the C-ABI host code (hydrocol.hip) and the kernel instantiations (hc_inst.hip).
//
// The step kernel exists per cells-per-lane count CPL = 2..10 x {special, generic exponents} x {monitoring, PREDICT
// lateral flow}: 36 instantiations of a ~90 KB kernel.  One translation unit per CPL and cell model (hc_inst.hip compiled
// with -DHC_INST_CPL=N -DHC_INST_SPECIAL=0|1) keeps them independent, so the build compiles them in parallel and with
// per-unit scheduler settings; the host code only sees the launch functions below.
#pragma once
#include <hip/hip_runtime.h>

#include "hc_step.h"

namespace hc {

struct LaunchCfg {
    hipStream_t stream;
    unsigned grid;       // workgroups
    bool special;        // every parameter point has the default exponents (n = 2, m = 1/2, lambda = 1, vrettas_fung)
    bool predict;        // repaired PREDICT lateral flow compiled in
};

inline size_t step_lds_bytes(int cpl, int wpb, int halves = 1, bool special = true)
{
    const size_t slots = (size_t)WAVE * cpl;
    return ((size_t)ntab_lds((int)slots * halves, special) * slots * 8 + 4 * slots * 1) * halves +
           (size_t)wpb * ((size_t)lds_wave_doubles(cpl, halves, special) + WAVE_SCRATCH) * 8 +
           (halves == 2 ? (size_t)(wpb / 2) * sizeof(PairBox) : 0);
}
inline size_t rhs_lds_bytes(int cpl, int wpb)
{
    const size_t slots = (size_t)WAVE * cpl;
    return NTAB * slots * 8 + (size_t)wpb * slots * 8;
}

// defined in hc_inst.hip, one explicit specialisation per translation unit: a unit holds the kernels of one cells-per-lane
// count AND one cell model (specialised / generic exponents), so that each can be compiled with its own scheduler settings
template <int CPL, bool SPECIAL> hipError_t launch_step_part(const LaunchCfg &cfg, const StepArgs &A);
template <int CPL, bool SPECIAL> hipError_t launch_rhs_part(const LaunchCfg &cfg, const StepArgs &A, long long row, double *dydt,
                                                            double *aux);
template <int CPL> inline hipError_t launch_step_cpl(const LaunchCfg &cfg, const StepArgs &A)
{
    return cfg.special ? launch_step_part<CPL, true>(cfg, A) : launch_step_part<CPL, false>(cfg, A);
}
template <int CPL> inline hipError_t launch_rhs_cpl(const LaunchCfg &cfg, const StepArgs &A, long long row, double *dydt,
                                                    double *aux)
{
    return cfg.special ? launch_rhs_part<CPL, true>(cfg, A, row, dydt, aux) : launch_rhs_part<CPL, false>(cfg, A, row, dydt, aux);
}
// split column: two waves per member, PAIR_CPL nodes per lane each, D in (64 PAIR_CPL, 128 PAIR_CPL]  (hc_inst.hip with
// -DHC_INST_PAIR)
constexpr int PAIR_CPL = 5;
hipError_t launch_step_pair(const LaunchCfg &cfg, const StepArgs &A);
hipError_t launch_rhs_pair(const LaunchCfg &cfg, const StepArgs &A, long long row, double *dydt);   // test hook, no aux
#define HC_DECLARE_CPL(N)                                                                                        \
    template <> hipError_t launch_step_part<N, true>(const LaunchCfg &cfg, const StepArgs &A);                     \
    template <> hipError_t launch_step_part<N, false>(const LaunchCfg &cfg, const StepArgs &A);                    \
    template <> hipError_t launch_rhs_part<N, true>(const LaunchCfg &cfg, const StepArgs &A, long long row, double *dydt, \
                                                    double *aux);                                                 \
    template <> hipError_t launch_rhs_part<N, false>(const LaunchCfg &cfg, const StepArgs &A, long long row, double *dydt, \
                                                     double *aux);
HC_DECLARE_CPL(2) HC_DECLARE_CPL(3) HC_DECLARE_CPL(4) HC_DECLARE_CPL(5) HC_DECLARE_CPL(6)
HC_DECLARE_CPL(7) HC_DECLARE_CPL(8) HC_DECLARE_CPL(9) HC_DECLARE_CPL(10)
#undef HC_DECLARE_CPL

}  // namespace hc
