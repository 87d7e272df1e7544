// hydrocol.hip -- C-ABI (include/hydrocol.h) and kernel launches for gfx950.
// Host side is plain C++ over the HIP runtime; nothing here falls back to the CPU.
#include "../../include/hydrocol.h"

#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <climits>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "hc_launch.h"

using namespace hc;

// layouts the ctypes binding (hydromodel_amd/_lib.py) mirrors; tests/test_abi.py checks the Python side
static_assert(sizeof(hc_column_params) == 8 * 4 + 15 * 8 + 2 * 4, "hc_column_params layout");
static_assert(sizeof(hc_step_args) == 8 + 8 + 4 + 4 + 5 * 8 + 8 + 8, "hc_step_args layout");
static_assert(sizeof(hc_spinup_args) == 8 + 4 + 4 + 8 + 8 + 8 + 8, "hc_spinup_args layout");

namespace {

thread_local std::string g_err;

int fail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                       \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess)                                                               \
            return fail(HC_ERR_DEVICE, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                        __FILE__, __LINE__);                                                \
    } while (0)

template <typename T>
struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    int ensure(size_t count)
    {
        if (count <= n && p) return HC_OK;
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
        if (count == 0) return HC_OK;
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&p), count * sizeof(T)));
        n = count;
        return HC_OK;
    }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
    }
};

}  // namespace

struct hc_handle {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool have_column = false, have_forcing = false, have_noise = false;
    hc_column_params p{};        // parameter point 0; dim_d, n_groups and dz are shared by every point
    ColumnDev P{};               // point 0
    int cpl = 0, wpb = 0, slots = 0;
    bool special = false;        // every point is vrettas_fung with n = 2, m = 1/2, lambda = 1
    bool force_generic = false;  // hc_set_generic_exponents: never take the specialised cell model
    bool use_special() const { return special && !force_generic; }
    // parameter points (BASELINE config 5): host copies, uploaded by fill_args when `points_dirty`
    int n_points = 0, moments_points = 0;
    std::vector<ColumnDev> P_host;
    std::vector<double> tab_host, node_host;
    bool points_dirty = false;
    // split column (two waves per member, hc_device.h Comm<2>): columns of 513..640 nodes with one parameter point and
    // the root zone inside the upper half; its own slot layout of the tables (point 0 only)
    // HYDROCOL_SPLIT_COLUMN=0 keeps the one-wave kernels, =1 takes the split column wherever it applies (A/B, cross-checks).
    // Round 4 default: the split column from 10 cells per lane on (D = 577..640); the one-wave kernel of 9 cells per lane
    // (global region behind a buffer resource) had overtaken it at D = 513..576 (97.0 k against 95.7 k column-days/s).
    // Round 5: the split column on the TWO layout -- four pairs per CU -- runs 125 k at every depth it serves (generic
    // exponents 73 k) and is the default from 513 nodes on (use_pair below; profiles/r05_split_column_two.txt)
    bool pair_ok = false, no_split = false, force_split = false;
    std::vector<double> tab_pair_host;
    DevBuf<double> tab_pair;
    DevBuf<int> gtab_pair;
    // (round 5: the split column runs on the TWO layout, four pairs per CU -- 125 k column-days/s at D = 513 ... 640 against
    //  96 k of the one-wave kernel of 9 cells per lane at D = 541 -- so it is the default from 513 nodes on)
    bool use_pair() const { return pair_ok && !no_split && (force_split || cpl >= 9); }
    int chunk_members = 0;       // HYDROCOL_CHUNK_MEMBERS (0: derived from the member count)
    DevBuf<double> tab, node_tabs, precip, atm, psi, base, nscale, fresh, psi_rows, scratch_d, diag;
    DevBuf<double> wave_spill;   // per-wave vectors of deep columns that do not fit in LDS (hc_step.h WaveVecs)
    DevBuf<int> spin_iters;
    DevBuf<double> trace;        // diagnostic builds only
    int max_phase_iterations = MAX_PHASE_ITERATIONS;
    int scipy_152 = 0;           // hc_set_scipy_152
    bool strict_guard = false;   // HYDROCOL_STRICT_GUARD=1: a tripped iteration guard fails the call
    DevBuf<int> gtab, wtd_obs, draw_idx, stats, scratch_i;
    DevBuf<unsigned char> daylight, refresh;
    DevBuf<unsigned short> wtd_u16;
    DevBuf<long long> moments;
    DevBuf<unsigned long long> counters;
    DevBuf<ColumnDev> Pdev;
    DevBuf<IoArgs> iodev;
    // several parameter points: Philox key of each point's first member, walk order of the chunk ticket, per-point cost
    DevBuf<long long> point_base;
    DevBuf<int> point_order;
    DevBuf<unsigned long long> point_cost;
    std::vector<long long> base_host;            // set by hc_set_point_member_bases (empty: member_offset + k * members_per_point)
    std::vector<int> order_host;
    bool fixed_order = false;                    // HYDROCOL_POINT_ORDER=fixed: keep the walk in point order (A/B timing)
    std::vector<unsigned long long> cost_total;  // RHS evaluations per point since the points were installed
    IoArgs io_host{};
    std::vector<unsigned char> h_refresh;
    int64_t n_rows = 0, n_members = 0;
    bool philox = false;
    uint64_t seed = 0;
    int64_t member_offset = 0;
    int rows_per_launch = 0;     // 0: chosen from the member count (auto_rows_per_launch)
    int n_cu = 256;
    double jac_reject = NUM_JAC_DIFF_REJECT;
};

// ------------------------------------------------------------------ auxiliary kernels
namespace hc {

// per-row ensemble moments of the water-table index: one block per row, single writer
// (blockIdx.y = parameter point: its members are contiguous, its table is moments[point][3][n_forcing])
__global__ void moments_kernel(const unsigned short *wtd, const int *wtd_obs, long long n_members,
                               long long members_per_point, long long row_begin, long long n_forcing,
                               long long *moments_all)
{
    const int r = blockIdx.x;
    const long long row = row_begin + r;
    const long long first = (long long)blockIdx.y * members_per_point;
    long long *moments = moments_all + (size_t)blockIdx.y * 3 * n_forcing;
    long long s1 = 0, s2 = 0;
    for (long long k = first + threadIdx.x; k < first + members_per_point; k += blockDim.x) {
        const long long w = wtd[(size_t)r * n_members + k];
        s1 += w;
        s2 += w * w;
    }
    __shared__ long long sh1[256], sh2[256];
    sh1[threadIdx.x] = s1;
    sh2[threadIdx.x] = s2;
    __syncthreads();
    for (int o = blockDim.x / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
            sh1[threadIdx.x] += sh1[threadIdx.x + o];
            sh2[threadIdx.x] += sh2[threadIdx.x + o];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0 && wtd_obs[row] >= 0) {
        moments[row] += members_per_point;
        moments[n_forcing + row] += sh1[0];
        moments[2 * n_forcing + row] += sh2[0];
    }
}

__global__ void widen_u16(const unsigned short *in, int *out, size_t n)
{
    size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) out[k] = in[k];
}

// T_RDELTA of every point: the refined reciprocal of por - theta_res, by the cell model's own instruction sequence
__global__ void fill_rdelta(double *tab, const ColumnDev *P, int slots)
{
    double *t = tab + (size_t)blockIdx.x * NTAB * slots;
    const double theta_res = P[blockIdx.x].theta_res;
    for (int k = threadIdx.x; k < slots; k += blockDim.x)
        t[(size_t)T_RDELTA * slots + k] = refined_rcp(t[(size_t)T_POR * slots + k] - theta_res);
}
__global__ void fill_d(double *p, double v, size_t n)
{
    size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) p[k] = v;
}

// src holds one column per parameter point ([n_points][D]; one point: a single column for everybody)
__global__ void broadcast_state(const double *src, double *dst, int D, long long n_members, long long members_per_point)
{
    size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < (size_t)D * n_members) dst[k] = src[((k / D) / members_per_point) * D + k % D];
}

__global__ void philox_dump(unsigned long long seed, long long member, unsigned draw, int D, double *out)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < D) out[i] = philox_normal(seed, (unsigned long long)member, draw, (unsigned)i);
}

// plugin call on the nodes (diagnostics), one thread per (member, node)
__global__ void model_nodes_kernel(const StepArgs A, const double *node_tabs, int special, double *out,
                                   double *qinf)
{
    const IoArgs io = load_const(A.io);
    const int D = A.D;
    const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)A.n_members * D;
    if (k >= total) return;
    const long long member = k / D;
    const int i = (int)(k % D);
    const long long point = member / A.members_per_point;
    const ColumnDev P = A.P[point];
    node_tabs += (size_t)point * 3 * D;
    const double por = node_tabs[i], meank = node_tabs[D + i], noisec = node_tabs[2 * D + i];
    const double mk = meank == 0.0 ? 1.0e-7 : meank;
    double z;
    if (A.host_noise)
        z = io.base_noise[member * D + i];
    else
    {
        const long long gid = A.n_points > 1 ? io.point_base[point] + (member - point * A.members_per_point)
                                             : io.member_offset + member;
        z = philox_normal(io.seed, (unsigned long long)gid, 0u, (unsigned)i) * io.nscale[member];
    }
    double th, K, C, kb, pf;
    if (special)
        model_cell<true>(P, io.psi[k], por, 1.0 / (por - P.theta_res), log(mk), 1.0 / (mk * mk), noisec, noisec * z, th, K, C, kb, pf);
    else
        model_cell<false>(P, io.psi[k], por, 1.0 / (por - P.theta_res), log(mk), 1.0 / (mk * mk), noisec, noisec * z, th, K, C, kb, pf);
    out[k] = th;
    out[total + k] = K;
    out[2 * total + k] = C;
    out[3 * total + k] = kb;
    if (qinf && i == 0) qinf[member] = fmin(2.0 * (por - th) * P.dz, kb);
}

// Stateless plugin call on arbitrary depths: psi [n_cells][n_cols] (depth-major, the reference's [dim_d x dim_m]),
// per-cell tables and noise [n_cells]; out [4][n_cells][n_cols] = theta, K, C, K_bkg; qinf [n_cols] from row 0
// (vrettas_fung.py:51-257, vanGenuchten.py:23-126).
__global__ void plugin_kernel(const ColumnDev P, int special, long long n_cells, long long n_cols, const double *psi,
                              const double *por, const double *meank, const double *noisec, const double *n_rnd,
                              double *out, double *qinf)
{
    const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)n_cells * n_cols;
    if (k >= total) return;
    const long long i = k / n_cols;
    const double mk = meank[i] == 0.0 ? 1.0e-7 : meank[i];
    double th, K, C, kb, pf;
    if (special)
        model_cell<true>(P, psi[k], por[i], 0.0, log(mk), 1.0 / (mk * mk), noisec[i], noisec[i] * n_rnd[i], th, K, C, kb, pf);
    else
        model_cell<false>(P, psi[k], por[i], 0.0, log(mk), 1.0 / (mk * mk), noisec[i], noisec[i] * n_rnd[i], th, K, C, kb, pf);
    out[k] = th;
    out[total + k] = K;
    out[2 * total + k] = C;
    out[3 * total + k] = kb;
    if (i == 0) qinf[k] = fmin(2.0 * (por[0] - th) * P.dz, kb);
}

}  // namespace hc

// ------------------------------------------------------------------ launch dispatch
namespace {

// kernel launch through the per-CPL translation units (hc_launch.h).  HC_CPL_MASK (bit n = cells-per-lane count n is
// linked in) lets development builds carry a few depths only.
#ifndef HC_CPL_MASK
#define HC_CPL_MASK 0x7FC
#endif
hipError_t launch_step_cpl_any(int cpl, bool &known, const LaunchCfg &cfg, const StepArgs &A)
{
    known = true;
    switch (cpl) {
#if (HC_CPL_MASK >> 2) & 1
        case 2: return launch_step_cpl<2>(cfg, A);
#endif
#if (HC_CPL_MASK >> 3) & 1
        case 3: return launch_step_cpl<3>(cfg, A);
#endif
#if (HC_CPL_MASK >> 4) & 1
        case 4: return launch_step_cpl<4>(cfg, A);
#endif
#if (HC_CPL_MASK >> 5) & 1
        case 5: return launch_step_cpl<5>(cfg, A);
#endif
#if (HC_CPL_MASK >> 6) & 1
        case 6: return launch_step_cpl<6>(cfg, A);
#endif
#if (HC_CPL_MASK >> 7) & 1
        case 7: return launch_step_cpl<7>(cfg, A);
#endif
#if (HC_CPL_MASK >> 8) & 1
        case 8: return launch_step_cpl<8>(cfg, A);
#endif
#if (HC_CPL_MASK >> 9) & 1
        case 9: return launch_step_cpl<9>(cfg, A);
#endif
#if (HC_CPL_MASK >> 10) & 1
        case 10: return launch_step_cpl<10>(cfg, A);
#endif
        default: break;
    }
    known = false;
    return hipSuccess;
}
hipError_t launch_rhs_cpl_any(int cpl, bool &known, const LaunchCfg &cfg, const StepArgs &A, long long row, double *dydt, double *aux)
{
    known = true;
    switch (cpl) {
#if (HC_CPL_MASK >> 2) & 1
        case 2: return launch_rhs_cpl<2>(cfg, A, row, dydt, aux);
#endif
#if (HC_CPL_MASK >> 3) & 1
        case 3: return launch_rhs_cpl<3>(cfg, A, row, dydt, aux);
#endif
#if (HC_CPL_MASK >> 4) & 1
        case 4: return launch_rhs_cpl<4>(cfg, A, row, dydt, aux);
#endif
#if (HC_CPL_MASK >> 5) & 1
        case 5: return launch_rhs_cpl<5>(cfg, A, row, dydt, aux);
#endif
#if (HC_CPL_MASK >> 6) & 1
        case 6: return launch_rhs_cpl<6>(cfg, A, row, dydt, aux);
#endif
#if (HC_CPL_MASK >> 7) & 1
        case 7: return launch_rhs_cpl<7>(cfg, A, row, dydt, aux);
#endif
#if (HC_CPL_MASK >> 8) & 1
        case 8: return launch_rhs_cpl<8>(cfg, A, row, dydt, aux);
#endif
#if (HC_CPL_MASK >> 9) & 1
        case 9: return launch_rhs_cpl<9>(cfg, A, row, dydt, aux);
#endif
#if (HC_CPL_MASK >> 10) & 1
        case 10: return launch_rhs_cpl<10>(cfg, A, row, dydt, aux);
#endif
        default: break;
    }
    known = false;
    return hipSuccess;
}

int unknown_depth(hc_handle *h)
{
    return fail(HC_ERR_UNSUPPORTED, "D = %d needs %d cells per lane; this build covers D <= %d (depth mask 0x%x)",
                h->p.dim_d, h->cpl, HC_MAX_DEPTH_NODES, HC_CPL_MASK);
}

LaunchCfg launch_cfg(hc_handle *h, unsigned grid)
{
    // PREDICT is per point in the tables but one kernel serves the launch: any predictive point selects the build
    // with the branch compiled in; a monitoring-mode point inside such a launch keeps its own semantics (flag_predict)
    bool predict = false;
    for (const ColumnDev &P : h->P_host) predict = predict || P.flag_predict;
    return LaunchCfg{h->stream, grid, h->use_special(), predict};
}

int launch_step(hc_handle *h, const StepArgs &A)
{
    const int wpb = wpb_of(h->cpl, 1, h->use_special());
    HIP_TRY(hipMemsetAsync(h->counters.p + 63, 0, sizeof(unsigned long long), h->stream));
    if (h->use_pair()) {
        // split column: a workgroup runs two members at a time, two waves each
        const int pairs = wpb_of(PAIR_CPL, 2) / 2;
        const long long want = (A.n_members + pairs - 1) / pairs;
        const unsigned grid = (unsigned)std::min<long long>(want, (long long)h->n_cu);
        StepArgs B = A;
        B.tab = h->tab_pair.p;
        B.gtab = h->gtab_pair.p;
        HIP_TRY(launch_step_pair(launch_cfg(h, grid), B));
        return HC_OK;
    }
    // persistent grid: LDS admits one workgroup per CU; fewer workgroups when there are fewer members
    const long long want = (A.n_members + wpb - 1) / wpb;
    const unsigned grid = (unsigned)std::min<long long>(want, (long long)h->n_cu);
    bool known = false;
    const hipError_t err = launch_step_cpl_any(h->cpl, known, launch_cfg(h, grid), A);
    if (!known) return unknown_depth(h);
    HIP_TRY(err);
    return HC_OK;
}

int launch_rhs(hc_handle *h, const StepArgs &A, long long row, double *dydt, double *aux)
{
    const int wpb = wpb_of(h->cpl, 1, h->use_special());
    if (h->use_pair() && !aux) {          // the split-column code path (the c | s | f view stays with the one-wave hook)
        StepArgs B = A;
        B.tab = h->tab_pair.p;
        B.gtab = h->gtab_pair.p;
        const int pairs = wpb_of(PAIR_CPL, 2) / 2;
        const unsigned grid = (unsigned)((A.n_members + pairs - 1) / pairs);
        HIP_TRY(launch_rhs_pair(launch_cfg(h, grid), B, row, dydt));
        return HC_OK;
    }
    const unsigned grid = (unsigned)((A.n_members + wpb - 1) / wpb);
    bool known = false;
    const hipError_t err = launch_rhs_cpl_any(h->cpl, known, launch_cfg(h, grid), A, row, dydt, aux);
    if (!known) return unknown_depth(h);
    HIP_TRY(err);
    return HC_OK;
}

int push_io(hc_handle *h)
{
    HIP_TRY(hipMemcpyAsync(h->iodev.p, &h->io_host, sizeof(IoArgs), hipMemcpyHostToDevice, h->stream));
    return HC_OK;
}

int fill_args(hc_handle *h, StepArgs &A)
{
    if (!h->have_column) return fail(HC_ERR_ARG, "hc_set_column has not been called");
    if (!h->have_forcing) return fail(HC_ERR_ARG, "hc_set_forcing has not been called");
    if (h->n_members <= 0 || !h->psi.p) return fail(HC_ERR_ARG, "hc_set_members / hc_set_state has not been called");
    if (!h->have_noise) return fail(HC_ERR_ARG, "no noise source: call hc_set_noise_host or hc_set_noise_philox");
    memset(&A, 0, sizeof(A));
    IoArgs &io = h->io_host;
    memset(&io, 0, sizeof(io));
    const int NP = h->n_points;
    if (h->n_members % NP != 0)
        return fail(HC_ERR_ARG, "%lld members do not divide into %d parameter points", (long long)h->n_members, NP);
    if (h->Pdev.ensure((size_t)NP) || h->iodev.ensure(1)) return HC_ERR_DEVICE;
    if (h->points_dirty) {
        if (h->tab.ensure(h->tab_host.size()) || h->node_tabs.ensure(h->node_host.size())) return HC_ERR_DEVICE;
        HIP_TRY(hipMemcpy(h->tab.p, h->tab_host.data(), h->tab_host.size() * 8, hipMemcpyHostToDevice));
        if (h->pair_ok) {
            if (h->tab_pair.ensure(h->tab_pair_host.size())) return HC_ERR_DEVICE;
            HIP_TRY(hipMemcpy(h->tab_pair.p, h->tab_pair_host.data(), h->tab_pair_host.size() * 8, hipMemcpyHostToDevice));
        }
        HIP_TRY(hipMemcpy(h->node_tabs.p, h->node_host.data(), h->node_host.size() * 8, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(h->Pdev.p, h->P_host.data(), (size_t)NP * sizeof(ColumnDev), hipMemcpyHostToDevice));
        {   // the reciprocal table is the device's own arithmetic (hc_device.h T_RDELTA)
            const int S1 = WAVE * h->cpl;
            hipLaunchKernelGGL(fill_rdelta, dim3((unsigned)NP), dim3(256), 0, h->stream, h->tab.p, h->Pdev.p, S1);
            if (h->pair_ok)
                hipLaunchKernelGGL(fill_rdelta, dim3((unsigned)NP), dim3(256), 0, h->stream, h->tab_pair.p, h->Pdev.p, 2 * WAVE * PAIR_CPL);
            HIP_TRY(hipStreamSynchronize(h->stream));
        }
        h->points_dirty = false;
    }
    if (h->moments_points != NP) {      // one [3][T] table per point, zeroed when the number of points changes
        const size_t cnt = (size_t)NP * 3 * h->n_rows;
        if (h->moments.ensure(cnt)) return HC_ERR_DEVICE;
        HIP_TRY(hipMemset(h->moments.p, 0, cnt * 8));
        h->moments_points = NP;
    }
    A.n_points = NP;
    A.members_per_point = h->n_members / NP;
    {
        // chunks of one point's members for the multi-point scheduler: >= 8 chunks per workgroup when the ensemble
        // allows it, <= 32 members per wave (that bounds the idle time at a chunk's end to ~1.5 %) and never fewer
        // members than the workgroup has waves
        // members a workgroup advances at once: its waves, or its pairs of waves on the split column
        const long long waves = h->use_pair() ? wpb_of(PAIR_CPL, 2) / 2 : wpb_of(h->cpl, 1, h->use_special());
        long long chunk = h->chunk_members > 0 ? h->chunk_members : (h->n_members + 8LL * h->n_cu - 1) / (8LL * h->n_cu);
        chunk = std::max<long long>(waves, std::min<long long>(chunk, 32 * waves));
        chunk = std::min<long long>(chunk, A.members_per_point);
        A.chunk_members = (int)chunk;
        A.chunks_per_point = (int)((A.members_per_point + chunk - 1) / chunk);
        A.n_chunks = A.chunks_per_point * NP;
    }
    {
        // deep columns: room for the per-wave vectors LDS cannot hold, for every wave of the persistent grid
        const size_t per_wave = std::max(std::max((size_t)spill_doubles(h->cpl, 1, true), (size_t)spill_doubles(h->cpl, 1, false)),
                                         (size_t)spill_doubles(PAIR_CPL, 2));
        const size_t cnt = (size_t)h->n_cu * MAX_WAVES_PER_BLOCK * per_wave;
        if (h->wave_spill.ensure(cnt)) return HC_ERR_DEVICE;
        A.wave_spill = h->wave_spill.p;
    }
    A.P = h->Pdev.p;
    A.io = h->iodev.p;
    A.tab = h->tab.p;          // (launch_step switches to the split-column tables)
    A.gtab = h->gtab.p;
    A.n_members = h->n_members;
    A.D = h->P.D;
    A.n_groups = h->P.n_groups;
    A.host_noise = h->philox ? 0 : 1;
    A.psi_sat = h->P.psi_sat;
    A.jac_reject = h->jac_reject;
    A.max_phase_iterations = h->max_phase_iterations;
    A.scipy_152 = h->scipy_152;
    io.psi = h->psi.p;
    io.base_noise = h->philox ? nullptr : h->base.p;
    io.nscale = h->nscale.p;
    io.precip = h->precip.p;
    io.atm = h->atm.p;
    io.daylight = h->daylight.p;
    io.refresh = h->refresh.p;
    io.wtd_obs = h->wtd_obs.p;
    io.draw_idx = h->draw_idx.p;
    io.member_offset = h->member_offset;
    io.seed = h->seed;
    io.counters = h->counters.p;
    io.queue = h->counters.p + 63;
    if (NP > 1) {
        if (h->point_base.ensure((size_t)NP) || h->point_order.ensure((size_t)NP) || h->point_cost.ensure((size_t)NP))
            return HC_ERR_DEVICE;
        std::vector<long long> base(h->base_host);
        if ((int)base.size() != NP) {
            base.resize((size_t)NP);
            for (int k = 0; k < NP; k++) base[(size_t)k] = h->member_offset + (long long)k * A.members_per_point;
        }
        if ((int)h->order_host.size() != NP) {
            h->order_host.resize((size_t)NP);
            for (int k = 0; k < NP; k++) h->order_host[(size_t)k] = k;
        }
        if ((int)h->cost_total.size() != NP) h->cost_total.assign((size_t)NP, 0ull);
        HIP_TRY(hipMemcpy(h->point_base.p, base.data(), (size_t)NP * 8, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(h->point_order.p, h->order_host.data(), (size_t)NP * 4, hipMemcpyHostToDevice));
        io.point_base = h->point_base.p;
        io.point_order = h->point_order.p;
        io.point_cost = h->point_cost.p;
    }
#ifdef HC_PROFILE
    if (getenv("HYDROCOL_DEBUG_TRACE")) {
        if (h->trace.ensure((size_t)1 + 6 * HC_TRACE_N)) return HC_ERR_DEVICE;
        hipMemset(h->trace.p, 0, ((size_t)1 + 6 * HC_TRACE_N) * 8);
        io.trace = h->trace.p;
    }
#endif
    return HC_OK;
}

}  // namespace

// ------------------------------------------------------------------ C-ABI
extern "C" {

const char *hc_last_error(void) { return g_err.c_str(); }
#ifndef HC_KERNEL_HASH
#define HC_KERNEL_HASH "unknown"
#endif
// "... kernels <hash>": identity of the device code (sources + compile flags + compiler, __graft_entry__.kernel_hash)
const char *hc_version(void) { return "hydrocol 0.4 (gfx950) kernels " HC_KERNEL_HASH; }

int hc_create(int device_ordinal, hc_handle **out)
{
    if (!out) return fail(HC_ERR_ARG, "hc_create: out is NULL");
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return fail(HC_ERR_NO_DEVICE, "no HIP device visible (%s): the hydrocol stepper has no CPU path",
                    e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
    if (device_ordinal < 0 || device_ordinal >= count)
        return fail(HC_ERR_ARG, "device ordinal %d out of range [0,%d)", device_ordinal, count);
    HIP_TRY(hipSetDevice(device_ordinal));
    hc_handle *h = new hc_handle();
    h->device = device_ordinal;
    // everything acquired so far is released when a later step fails
    auto init = [&]() -> int {
        HIP_TRY(hipStreamCreate(&h->stream));
        HIP_TRY(hipEventCreate(&h->ev0));
        HIP_TRY(hipEventCreate(&h->ev1));
        if (h->counters.ensure(128) != HC_OK) return HC_ERR_DEVICE;
        HIP_TRY(hipMemset(h->counters.p, 0, 128 * sizeof(unsigned long long)));
        hipDeviceProp_t prop;
        HIP_TRY(hipGetDeviceProperties(&prop, device_ordinal));
        h->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        if (const char *e = getenv("HYDROCOL_DEBUG_CUS"))      // measurement hook: a smaller persistent grid
            if (atoi(e) > 0) h->n_cu = std::min(h->n_cu, atoi(e));
        return HC_OK;
    };
    if (const int rc = init()) {
        const std::string keep = g_err;
        hc_destroy(h);
        g_err = keep;
        return rc;
    }
    if (const char *cm = getenv("HYDROCOL_CHUNK_MEMBERS"))
        if (atoi(cm) > 0) h->chunk_members = atoi(cm);
    const char *rpl = getenv("HYDROCOL_ROWS_PER_LAUNCH");
    if (rpl && atoi(rpl) > 0) h->rows_per_launch = atoi(rpl);
    if (const char *sg = getenv("HYDROCOL_STRICT_GUARD")) h->strict_guard = atoi(sg) != 0;
    if (const char *po = getenv("HYDROCOL_POINT_ORDER")) h->fixed_order = strcmp(po, "fixed") == 0;
    if (const char *sc = getenv("HYDROCOL_SPLIT_COLUMN")) {
        h->no_split = atoi(sc) == 0;
        h->force_split = atoi(sc) == 1;
    }
    if (const char *sv = getenv("HYDROCOL_SCIPY_152")) h->scipy_152 = atoi(sv) != 0;     // (the whole product at once: CLI, Simulation)
    if (const char *mi = getenv("HYDROCOL_DEBUG_MAX_ITER"))    // test hook: forces abandoned attempts
        if (atoi(mi) > 0) h->max_phase_iterations = atoi(mi);
    const char *jr = getenv("HYDROCOL_DEBUG_JAC_REJECT");   // test hook: exercises num_jac's retry branch
    if (jr && atof(jr) > 0.0) h->jac_reject = atof(jr);
    *out = h;
    return HC_OK;
}

int hc_destroy(hc_handle *h)
{
    if (!h) return HC_OK;
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->stream);
    h->tab.release(); h->node_tabs.release(); h->precip.release(); h->atm.release(); h->psi.release();
    h->base.release(); h->nscale.release(); h->fresh.release(); h->psi_rows.release(); h->scratch_d.release();
    h->diag.release();
    h->wave_spill.release();
    h->spin_iters.release();
    h->trace.release();
    h->tab_pair.release(); h->gtab_pair.release();
    h->gtab.release(); h->wtd_obs.release(); h->draw_idx.release(); h->stats.release(); h->scratch_i.release();
    h->Pdev.release(); h->iodev.release();
    h->point_base.release(); h->point_order.release(); h->point_cost.release();
    h->daylight.release(); h->refresh.release(); h->wtd_u16.release(); h->moments.release(); h->counters.release();
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return HC_OK;
}

// One parameter point: ColumnDev + the slot tables of the stepper.  `first` fixes the geometry shared by all points.
static int build_point(hc_handle *h, const hc_column_params *p, const double *node_tabs, const double *mid_tabs,
                       bool first)
{
    const int D = p->dim_d;
    if (!(p->dz > 0.0) || !(p->n > 1.0) || !(p->alpha > 0.0)) return fail(HC_ERR_ARG, "bad dz / n / alpha");
    if (p->n_root_first < 0 || p->n_root_first > 1 || p->n_root_int < 0 || p->n_root_int > D - 2)
        return fail(HC_ERR_ARG, "bad root-zone cell counts");
    if (!first && (D != h->p.dim_d || p->n_groups != h->p.n_groups || p->dz != h->p.dz))
        return fail(HC_ERR_ARG, "a parameter point must share dim_d, n_groups and dz with point 0");
    // low_lim = k - (sat_cells - 1) must stay inside the k-cell slice pde_fun sees: sat_cells >= 1
    if (p->flag_predict && p->sat_cells < 1) return fail(HC_ERR_ARG, "PREDICT mode: sat_cells = %d, must be >= 1", p->sat_cells);
    ColumnDev P{};
    P.D = D; P.model = p->model; P.flag_et = p->flag_et; P.flag_lf = p->flag_lf; P.flag_hlift = p->flag_hlift;
    P.n_root_first = p->n_root_first; P.n_root_int = p->n_root_int; P.n_groups = p->n_groups;
    P.theta_res = p->theta_res; P.alpha = p->alpha; P.n = p->n; P.m = p->m; P.psi_sat = p->psi_sat;
    P.epsilon = p->epsilon; P.lambda = p->lambda_exp; P.sigma = p->sigma_noise; P.sat_soil = p->sat_soil;
    P.dz = p->dz; P.inv_dz = 1.0 / p->dz; P.ipsi50 = p->ipsi50; P.lai = p->lai; P.surface_evap = p->surface_evap;
    P.interception = p->interception; P.evap_delta_min = p->evap_delta_min;
    P.mn_alpha = (p->m * p->n) * p->alpha;
    P.inv_m = 1.0 / p->m;
    P.por_node0 = node_tabs[0];
    // repaired PREDICT mode: low_lim = dim_d - (sat_cells - 1) of each pde_fun call as an int, nothing drains when
    // it is not positive (the reference's np.linspace(1.5, 0.0, low_lim) raises for a float or a negative count)
    P.flag_predict = p->flag_predict ? 1 : 0;
    P.predict_low = std::min(D - 2, std::max(0, (D - 2) - (p->sat_cells - 1)));
    P.predict_first = (1 - (p->sat_cells - 1)) >= 1 ? 1 : 0;

    const int M = D - 1;
    // Slot tables for `halves` waves per member with `cpl` nodes per lane: node / midpoint i sits with wide lane
    // wl = i / cpl, cell c = i % cpl, i.e. in wave wl / 64, slot c * 64 + wl % 64 of that wave's 64 cpl slots.
    auto layout = [&](int cpl, int halves, std::vector<double> &tab) {
        const int SW = WAVE * cpl, S = SW * halves;
        tab.assign((size_t)NTAB * S, 0.0);
        auto put = [&](int slot, double por, double fc, double wlt, double root, double meank, double noisec) {
            const double mk = meank == 0.0 ? 1.0e-7 : meank;   // utilities.py:50
            tab[(size_t)T_POR * S + slot] = por;
            tab[(size_t)T_FC * S + slot] = fc;
            tab[(size_t)T_WLT * S + slot] = wlt;
            tab[(size_t)T_ROOT * S + slot] = root;
            tab[(size_t)T_LOGM * S + slot] = std::log(mk);
            tab[(size_t)T_INVM2 * S + slot] = 1.0 / (mk * mk);
            tab[(size_t)T_NOISEC * S + slot] = noisec;
            tab[(size_t)T_VALID * S + slot] = 1.0;
            const double d1 = por - wlt;                       // tree_roots.py:235-238
            tab[(size_t)T_INVD1 * S + slot] = 1.0 / (d1 == 0.0 ? 1.0 : d1);
        };
        for (int wl = 0; wl < WAVE * halves; wl++)
            for (int c = 0; c < cpl; c++) {
                const int i = wl * cpl + c, slot = (wl / WAVE) * SW + c * WAVE + wl % WAVE;
                if (i < M)
                    put(slot, mid_tabs[i], mid_tabs[M + i], mid_tabs[2 * M + i], mid_tabs[3 * M + i],
                        mid_tabs[4 * M + i], mid_tabs[5 * M + i]);
                else
                    put(slot, 0.3, 0.2, 0.1, 0.0, 1.0, 0.0);    // padding cell: benign, results masked
                if (i >= M) tab[(size_t)T_VALID * S + slot] = 0.0;
            }
        // virtual top-node cell in the always-free last slot of the last lane (of the last wave)
        const int top = (halves - 1) * SW + (cpl - 1) * WAVE + (WAVE - 1);
        put(top, node_tabs[0], 0.2, 0.1, 0.0, node_tabs[D + 0], node_tabs[2 * D + 0]);
        tab[(size_t)T_VALID * S + top] = 0.0;                   // its C / flux never enter the assembly
    };
    std::vector<double> tab;
    layout(h->cpl, 1, tab);
    for (double v : tab)
        if (!std::isfinite(v)) return fail(HC_ERR_ARG, "a column table entry is not finite");
    // split column: 513..640 nodes, the root zone (cells 1..n_root_int) of EVERY point inside the upper half; each point
    // brings its own tables in the two-halves layout (round 4: sweeps at these depths run on the split column too)
    const bool pair_here = D > WAVE * 8 && D <= 2 * WAVE * PAIR_CPL && p->n_root_int <= WAVE * PAIR_CPL - 1;
    if (first) {
        h->pair_ok = pair_here;
        h->tab_pair_host.clear();
    } else {
        h->pair_ok = h->pair_ok && pair_here;
    }
    if (h->pair_ok) {
        std::vector<double> tp;
        layout(PAIR_CPL, 2, tp);
        h->tab_pair_host.insert(h->tab_pair_host.end(), tp.begin(), tp.end());
    }
    const bool special = (p->model == HC_MODEL_VRETTAS_FUNG && p->n == 2.0 && p->m == 0.5 && p->lambda_exp == 1.0);
    if (first) {
        h->P_host.clear(); h->tab_host.clear(); h->node_host.clear();
        h->base_host.clear(); h->order_host.clear(); h->cost_total.clear();
        h->n_points = 0;
        h->p = *p;
        h->P = P;
        h->special = special;
    } else {
        h->special = h->special && special;
    }
    h->P_host.push_back(P);
    h->tab_host.insert(h->tab_host.end(), tab.begin(), tab.end());
    h->node_host.insert(h->node_host.end(), node_tabs, node_tabs + (size_t)3 * D);
    h->n_points++;
    h->points_dirty = true;
    return HC_OK;
}

int hc_set_column(hc_handle *h, const hc_column_params *p, const double *node_tabs, const double *mid_tabs,
                  const int32_t *groups)
{
    if (!h || !p || !node_tabs || !mid_tabs || !groups) return fail(HC_ERR_ARG, "hc_set_column: NULL argument");
    const int D = p->dim_d;
    if (D < 4 || D > HC_MAX_DEPTH_NODES) return fail(HC_ERR_ARG, "dim_d = %d outside [4, %d]", D, HC_MAX_DEPTH_NODES);
    if (p->n_groups < 1 || p->n_groups > 16) return fail(HC_ERR_ARG, "n_groups = %d outside [1,16]", p->n_groups);
    for (int i = 0; i < D; i++)
        if (groups[i] < 0 || groups[i] >= p->n_groups) return fail(HC_ERR_ARG, "groups[%d] out of range", i);
    HIP_TRY(hipSetDevice(h->device));
    int cpl = (D + WAVE - 1) / WAVE;
    if (cpl < 2) cpl = 2;
    h->cpl = cpl;
    h->wpb = 4;
    h->slots = WAVE * cpl;
    h->have_column = false;
    int rc = build_point(h, p, node_tabs, mid_tabs, true);
    if (rc) return rc;
    auto group_layout = [&](int cpl_, int halves, std::vector<int> &gt) {
        const int SW = WAVE * cpl_, S = SW * halves;
        gt.assign((size_t)NGTAB * S, -1);
        for (int wl = 0; wl < WAVE * halves; wl++)
            for (int c = 0; c < cpl_; c++) {
                const int i = wl * cpl_ + c, slot = (wl / WAVE) * SW + c * WAVE + wl % WAVE;
                if (i < D) {
                    gt[(size_t)G_SELF * S + slot] = groups[i];
                    gt[(size_t)G_PREV * S + slot] = i >= 1 ? groups[i - 1] : -1;
                    gt[(size_t)G_NEXT * S + slot] = i < D - 1 ? groups[i + 1] : -1;
                }
            }
    };
    std::vector<int> gt;
    group_layout(cpl, 1, gt);
    if (h->gtab.ensure(gt.size())) return HC_ERR_DEVICE;
    HIP_TRY(hipMemcpy(h->gtab.p, gt.data(), gt.size() * 4, hipMemcpyHostToDevice));
    if (h->pair_ok) {
        group_layout(PAIR_CPL, 2, gt);
        if (h->gtab_pair.ensure(gt.size())) return HC_ERR_DEVICE;
        HIP_TRY(hipMemcpy(h->gtab_pair.p, gt.data(), gt.size() * 4, hipMemcpyHostToDevice));
    }
    h->have_column = true;
    return HC_OK;
}

int hc_add_point(hc_handle *h, const hc_column_params *p, const double *node_tabs, const double *mid_tabs)
{
    if (!h || !p || !node_tabs || !mid_tabs) return fail(HC_ERR_ARG, "hc_add_point: NULL argument");
    if (!h->have_column) return fail(HC_ERR_ARG, "hc_set_column (point 0) must come first");
    return build_point(h, p, node_tabs, mid_tabs, false);
}

int hc_get_point_count(hc_handle *h)
{
    if (!h) return fail(HC_ERR_ARG, "NULL handle");
    return h->n_points;
}

int hc_set_forcing(hc_handle *h, int64_t n_rows, const double *precip, const double *atm,
                   const uint8_t *daylight, const int32_t *wtd_obs, const uint8_t *refresh)
{
    if (!h || n_rows < 1 || !precip || !atm || !daylight || !wtd_obs || !refresh)
        return fail(HC_ERR_ARG, "hc_set_forcing: bad argument");
    HIP_TRY(hipSetDevice(h->device));
    const size_t T = (size_t)n_rows;
    // A skipped row (wtd_obs < 0) consumes no noise draw: the reference `continue`s before drawing
    // (simulation.py:582-588 come before :599-602), so its refresh flag is dropped here for everybody downstream.
    std::vector<unsigned char> refresh_eff(refresh, refresh + T);
    for (size_t i = 0; i < T; i++)
        if (wtd_obs[i] < 0) refresh_eff[i] = 0;
    refresh = refresh_eff.data();
    std::vector<int> draw(T, 0);
    int cnt = 0;
    for (size_t i = 0; i < T; i++) {
        if (refresh[i]) cnt++;
        draw[i] = cnt;
    }
    if (h->precip.ensure(T) || h->atm.ensure(T) || h->daylight.ensure(T) || h->refresh.ensure(T) ||
        h->wtd_obs.ensure(T) || h->draw_idx.ensure(T))
        return HC_ERR_DEVICE;
    HIP_TRY(hipMemcpy(h->precip.p, precip, T * 8, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(h->atm.p, atm, T * 8, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(h->daylight.p, daylight, T, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(h->refresh.p, refresh, T, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(h->wtd_obs.p, wtd_obs, T * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(h->draw_idx.p, draw.data(), T * 4, hipMemcpyHostToDevice));
    h->h_refresh.assign(refresh, refresh + T);
    h->n_rows = n_rows;
    h->moments_points = 0;       // (re)allocated and zeroed by the next call that needs the moment tables
    h->have_forcing = true;
    return HC_OK;
}

int hc_set_forcing_row(hc_handle *h, int64_t row, double precip, double atm, uint8_t daylight, int32_t wtd_obs)
{
    if (!h || !h->have_forcing) return fail(HC_ERR_ARG, "hc_set_forcing_row: hc_set_forcing has not been called");
    if (row < 0 || row >= h->n_rows) return fail(HC_ERR_ARG, "hc_set_forcing_row: row %lld outside [0, %lld)", (long long)row,
                                                 (long long)h->n_rows);
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    const unsigned char zero = 0;
    HIP_TRY(hipMemcpy(h->precip.p + row, &precip, 8, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(h->atm.p + row, &atm, 8, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(h->daylight.p + row, &daylight, 1, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(h->wtd_obs.p + row, &wtd_obs, 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(h->refresh.p + row, &zero, 1, hipMemcpyHostToDevice));
    h->h_refresh[(size_t)row] = 0;
    return HC_OK;
}

int hc_set_members(hc_handle *h, int64_t n_members)
{
    if (!h || n_members < 1) return fail(HC_ERR_ARG, "hc_set_members: bad argument");
    if (!h->have_column) return fail(HC_ERR_ARG, "hc_set_column must come first");
    HIP_TRY(hipSetDevice(h->device));
    const size_t n = (size_t)n_members * h->p.dim_d;
    if (h->psi.ensure(n) || h->nscale.ensure((size_t)n_members)) return HC_ERR_DEVICE;
    hipLaunchKernelGGL(fill_d, dim3((unsigned)((n_members + 255) / 256)), dim3(256), 0, h->stream, h->nscale.p, 1.0,
                       (size_t)n_members);
    HIP_TRY(hipGetLastError());
    h->n_members = n_members;
    h->have_noise = false;
    return HC_OK;
}

int hc_set_state(hc_handle *h, const double *psi, int broadcast)
{
    if (!h || !psi) return fail(HC_ERR_ARG, "hc_set_state: bad argument");
    if (h->n_members <= 0) return fail(HC_ERR_ARG, "hc_set_members must come first");
    HIP_TRY(hipSetDevice(h->device));
    const int D = h->p.dim_d;
    if (broadcast < 0 || broadcast > 2) return fail(HC_ERR_ARG, "hc_set_state: broadcast must be 0, 1 or 2");
    if (broadcast == 2 && h->n_members % h->n_points != 0)
        return fail(HC_ERR_ARG, "%lld members do not divide into %d parameter points", (long long)h->n_members, h->n_points);
    const size_t n = (size_t)h->n_members * D;
    const size_t n_in = broadcast == 1 ? (size_t)D : (broadcast == 2 ? (size_t)h->n_points * D : n);
    for (size_t i = 0; i < n_in; i++)
        if (!std::isfinite(psi[i])) return fail(HC_ERR_ARG, "state entry %zu is not finite", i);
    if (broadcast) {
        if (h->scratch_d.ensure(n_in)) return HC_ERR_DEVICE;
        HIP_TRY(hipMemcpyAsync(h->scratch_d.p, psi, n_in * 8, hipMemcpyHostToDevice, h->stream));
        const long long per_point = broadcast == 2 ? h->n_members / h->n_points : h->n_members;
        hipLaunchKernelGGL(broadcast_state, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream,
                           h->scratch_d.p, h->psi.p, D, (long long)h->n_members, per_point);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(h->stream));
    } else {
        HIP_TRY(hipMemcpy(h->psi.p, psi, n * 8, hipMemcpyHostToDevice));
    }
    return HC_OK;
}

int hc_get_state(hc_handle *h, double *psi, int64_t first, int64_t count)
{
    if (!h || !psi || first < 0 || count < 0 || first + count > h->n_members)
        return fail(HC_ERR_ARG, "hc_get_state: bad range");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    const int D = h->p.dim_d;
    HIP_TRY(hipMemcpy(psi, h->psi.p + (size_t)first * D, (size_t)count * D * 8, hipMemcpyDeviceToHost));
    return HC_OK;
}

int hc_set_noise_host(hc_handle *h, const double *base)
{
    if (!h || !base) return fail(HC_ERR_ARG, "hc_set_noise_host: bad argument");
    if (h->n_members <= 0) return fail(HC_ERR_ARG, "hc_set_members must come first");
    HIP_TRY(hipSetDevice(h->device));
    const size_t n = (size_t)h->n_members * h->p.dim_d;
    if (h->base.ensure(n)) return HC_ERR_DEVICE;
    HIP_TRY(hipMemcpy(h->base.p, base, n * 8, hipMemcpyHostToDevice));
    h->philox = false;
    h->have_noise = true;
    return HC_OK;
}

int hc_get_noise_base(hc_handle *h, double *base, int64_t first, int64_t count)
{
    if (!h || !base || h->philox || !h->base.p || first < 0 || count < 0 || first + count > h->n_members)
        return fail(HC_ERR_ARG, "hc_get_noise_base: bad argument (host noise mode only)");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    const int D = h->p.dim_d;
    HIP_TRY(hipMemcpy(base, h->base.p + (size_t)first * D, (size_t)count * D * 8, hipMemcpyDeviceToHost));
    return HC_OK;
}

int hc_set_noise_philox(hc_handle *h, uint64_t seed, int64_t member_offset)
{
    if (!h || member_offset < 0) return fail(HC_ERR_ARG, "hc_set_noise_philox: bad argument");
    if (h->n_members <= 0) return fail(HC_ERR_ARG, "hc_set_members must come first");
    HIP_TRY(hipSetDevice(h->device));
    hipLaunchKernelGGL(fill_d, dim3((unsigned)((h->n_members + 255) / 256)), dim3(256), 0, h->stream, h->nscale.p,
                       1.0, (size_t)h->n_members);
    HIP_TRY(hipGetLastError());
    h->seed = seed;
    h->member_offset = member_offset;
    h->philox = true;
    h->have_noise = true;
    return HC_OK;
}

int hc_philox_normals(hc_handle *h, int64_t member, int64_t draw, double *out)
{
    if (!h || !out || member < 0 || draw < 0) return fail(HC_ERR_ARG, "hc_philox_normals: bad argument");
    if (!h->have_column) return fail(HC_ERR_ARG, "hc_set_column must come first");
    HIP_TRY(hipSetDevice(h->device));
    const int D = h->p.dim_d;
    if (h->scratch_d.ensure(D)) return HC_ERR_DEVICE;
    hipLaunchKernelGGL(philox_dump, dim3((D + 63) / 64), dim3(64), 0, h->stream, (unsigned long long)h->seed,
                       (long long)member, (unsigned)draw, D, h->scratch_d.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(h->stream));
    HIP_TRY(hipMemcpy(out, h->scratch_d.p, (size_t)D * 8, hipMemcpyDeviceToHost));
    return HC_OK;
}

// One simulated day per launch for very large ensembles; smaller ones get proportionally longer launches, so that a launch
// holds ~1 M member-days of work (in-kernel noise; round 5: 64 k before) and the tail behind its slowest wavefront stays
// small: 4 096 members x D = 200 over the year 308.5 k column-days/s at 16 days per launch, 319 k at 64, 324 k at 128, 327 k at
// 364; 65 536 members 324 k at 1 day, 331 k at 4, 334 k at 16; 16 384 members 286 k at 4 days, 292 k at 64
// (gpurun_out/r5zd).  With the caller's noise every refreshed row of a launch stages members x D doubles on the device:
// there the launches stay at ~64 k member-days (include/hydrocol.h).
static int auto_rows_per_launch(int64_t n_members, bool in_kernel_noise)
{
    const int64_t target = in_kernel_noise ? (int64_t(1) << 20) : 65536;
    const int64_t days = std::max<int64_t>(1, std::min<int64_t>(target / std::max<int64_t>(n_members, 1), 365));
    return (int)(48 * days);
}

int hc_step_rows(hc_handle *h, hc_step_args *a)
{
    if (!h || !a) return fail(HC_ERR_ARG, "hc_step_rows: NULL argument");
    StepArgs A;
    int rc = fill_args(h, A);
    if (rc) return rc;
    if (a->n_rows < 0) return fail(HC_ERR_ARG, "n_rows < 0");
    if (a->spinup && a->accumulate_moments)
        return fail(HC_ERR_ARG, "spin-up solves have no forcing row of their own: accumulate_moments must be 0");
    if (a->spinup) {
        if (a->row_begin < 0 || a->row_begin >= h->n_rows) return fail(HC_ERR_ARG, "spin-up forcing row out of range");
    } else if (a->row_begin < 1 || a->row_begin + a->n_rows > h->n_rows) {
        return fail(HC_ERR_ARG, "rows [%lld, %lld) outside [1, %lld)", (long long)a->row_begin,
                    (long long)(a->row_begin + a->n_rows), (long long)h->n_rows);
    }
    HIP_TRY(hipSetDevice(h->device));
    const int D = h->p.dim_d;
    const int64_t N = h->n_members;
    a->kernel_ms = 0.0;
    a->launches = 0;
    int64_t fresh_consumed = 0;
    // Per-row outputs are staged in device buffers of rows x members entries: a caller that asks for them gets shorter
    // launches, so that the largest (psi_rows: 8 D bytes per member-row) stays within ~1 GiB however long the request.
    int64_t out_bytes_per_row = 0;
    if (a->psi_rows_out) out_bytes_per_row += N * (int64_t)D * 8;
    if (a->stats_out) out_bytes_per_row += N * 6 * 4;
    if (a->diag_out) out_bytes_per_row += N * 2 * 8;
    if (a->wtd_out) out_bytes_per_row += N * 4;
    const int64_t rows_cap = out_bytes_per_row > 0 ? std::max<int64_t>(1, (int64_t(1) << 30) / out_bytes_per_row) : INT32_MAX;
    for (int64_t done = 0; done < a->n_rows;) {
        int per_launch = h->rows_per_launch > 0 ? h->rows_per_launch : auto_rows_per_launch(N, h->philox);
        if (h->rows_per_launch <= 0) per_launch = (int)std::min<int64_t>(per_launch, rows_cap);
        const int chunk = (int)std::min<int64_t>(per_launch, a->n_rows - done);
        const int64_t row0 = a->spinup ? a->row_begin : a->row_begin + done;
        int n_fresh = 0;
        if (!a->spinup)
            for (int r = 0; r < chunk; r++) n_fresh += h->h_refresh[(size_t)(row0 + r)] ? 1 : 0;
        if (!h->philox && n_fresh > 0) {
            if (!a->fresh_noise) return fail(HC_ERR_ARG, "host noise mode: fresh_noise is NULL but rows refresh");
            const size_t cnt = (size_t)n_fresh * N * D;
            if (h->fresh.ensure(cnt)) return HC_ERR_DEVICE;
            HIP_TRY(hipMemcpyAsync(h->fresh.p, a->fresh_noise + (size_t)fresh_consumed * N * D, cnt * 8,
                                   hipMemcpyHostToDevice, h->stream));
        }
        if (h->wtd_u16.ensure((size_t)chunk * N)) return HC_ERR_DEVICE;
        if (a->stats_out && h->stats.ensure((size_t)chunk * N * 6)) return HC_ERR_DEVICE;
        if (a->psi_rows_out && h->psi_rows.ensure((size_t)chunk * N * D)) return HC_ERR_DEVICE;
        if (a->diag_out && h->diag.ensure((size_t)chunk * N * 2)) return HC_ERR_DEVICE;
        h->io_host.fresh = h->fresh.p;
        h->io_host.row_begin = row0;
        A.n_rows = chunk;
        A.spinup = a->spinup;
        h->io_host.wtd_u16 = h->wtd_u16.p;
        h->io_host.stats = a->stats_out ? h->stats.p : nullptr;
        h->io_host.psi_rows = a->psi_rows_out ? h->psi_rows.p : nullptr;
        h->io_host.diag = a->diag_out ? h->diag.p : nullptr;
        rc = push_io(h);
        if (rc) return rc;
        if (h->n_points > 1) {
            HIP_TRY(hipMemcpyAsync(h->point_order.p, h->order_host.data(), (size_t)h->n_points * 4, hipMemcpyHostToDevice,
                                   h->stream));
            HIP_TRY(hipMemsetAsync(h->point_cost.p, 0, (size_t)h->n_points * 8, h->stream));
        }
        HIP_TRY(hipEventRecord(h->ev0, h->stream));
        rc = launch_step(h, A);
        if (rc) return rc;
        HIP_TRY(hipEventRecord(h->ev1, h->stream));
        if (a->accumulate_moments) {
            hipLaunchKernelGGL(moments_kernel, dim3(chunk, h->n_points), dim3(256), 0, h->stream, h->wtd_u16.p,
                               h->wtd_obs.p, (long long)N, (long long)(N / h->n_points), (long long)row0,
                               (long long)h->n_rows, h->moments.p);
            HIP_TRY(hipGetLastError());
        }
        if (a->wtd_out) {
            const size_t cnt = (size_t)chunk * N;
            if (h->scratch_i.ensure(cnt)) return HC_ERR_DEVICE;
            hipLaunchKernelGGL(widen_u16, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, h->stream, h->wtd_u16.p,
                               h->scratch_i.p, cnt);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipMemcpyAsync(a->wtd_out + (size_t)done * N, h->scratch_i.p, cnt * 4, hipMemcpyDeviceToHost,
                                   h->stream));
        }
        if (a->stats_out)
            HIP_TRY(hipMemcpyAsync(a->stats_out + (size_t)done * N * 6, h->stats.p, (size_t)chunk * N * 6 * 4,
                                   hipMemcpyDeviceToHost, h->stream));
        if (a->psi_rows_out)
            HIP_TRY(hipMemcpyAsync(a->psi_rows_out + (size_t)done * N * D, h->psi_rows.p, (size_t)chunk * N * D * 8,
                                   hipMemcpyDeviceToHost, h->stream));
        if (a->diag_out)
            HIP_TRY(hipMemcpyAsync(a->diag_out + (size_t)done * N * 2, h->diag.p, (size_t)chunk * N * 2 * 8,
                                   hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, h->ev0, h->ev1));
        a->kernel_ms += ms;
        a->launches++;
        fresh_consumed += n_fresh;
        done += chunk;
        if (h->n_points > 1 && !a->spinup) {
            // what each point cost in this launch orders the next one: costliest point first (results do not depend on it)
            std::vector<unsigned long long> cost((size_t)h->n_points);
            HIP_TRY(hipMemcpy(cost.data(), h->point_cost.p, cost.size() * 8, hipMemcpyDeviceToHost));
            for (int k = 0; k < h->n_points; k++) h->cost_total[(size_t)k] += cost[(size_t)k];
            if (!h->fixed_order)
                std::stable_sort(h->order_host.begin(), h->order_host.end(),
                                 [&](int x, int y) { return cost[(size_t)x] > cost[(size_t)y]; });
        }
    }
    unsigned long long cnt[6];
    HIP_TRY(hipMemcpy(cnt, h->counters.p, sizeof(cnt), hipMemcpyDeviceToHost));
    if (cnt[5] != 0)
        return fail(HC_ERR_DEVICE, "split-column kernel: %llu mailbox exchanges timed out (internal error; results are invalid)", cnt[5]);
    // An attempt that exhausts the kernel's iteration budget is abandoned like a solve that gave up (the x0.8
    // retry rule applies); it is counted ([2], last place in [3]) and only fatal on request.
    if (cnt[2] != 0 && h->strict_guard)
        return fail(HC_ERR_DEVICE, "%llu BDF attempts hit the kernel's iteration guard (last: member %llu, row %llu)",
                    cnt[2], cnt[3] >> 24, cnt[3] & 0xFFFFFFull);
    return HC_OK;
}

int hc_spinup(hc_handle *h, hc_spinup_args *a)
{
    if (!h || !a || !a->iterations_out) return fail(HC_ERR_ARG, "hc_spinup: NULL argument");
    StepArgs A;
    int rc = fill_args(h, A);
    if (rc) return rc;
    if (a->forcing_row < 0 || a->forcing_row >= h->n_rows) return fail(HC_ERR_ARG, "spin-up forcing row out of range");
    if (a->max_iterations < 1 || a->max_iterations > 1000000) return fail(HC_ERR_ARG, "max_iterations out of range");
    HIP_TRY(hipSetDevice(h->device));
    const int64_t N = h->n_members;
    if (h->spin_iters.ensure((size_t)N)) return HC_ERR_DEVICE;
    h->io_host.row_begin = a->forcing_row;
    h->io_host.spin_iters = h->spin_iters.p;
    A.n_rows = a->max_iterations;
    A.spinup = 1;
    A.spin_stop = 1;
    A.spin_zwtd = a->zwtd_cm;
    A.spin_z0 = a->z0_cm;
    A.spin_dz = h->p.dz;
    rc = push_io(h);
    if (rc) return rc;
    HIP_TRY(hipEventRecord(h->ev0, h->stream));
    rc = launch_step(h, A);
    if (rc) return rc;
    HIP_TRY(hipEventRecord(h->ev1, h->stream));
    HIP_TRY(hipMemcpyAsync(a->iterations_out, h->spin_iters.p, (size_t)N * 4, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, h->ev0, h->ev1));
    a->kernel_ms = ms;
    unsigned long long cnt[6];
    HIP_TRY(hipMemcpy(cnt, h->counters.p, sizeof(cnt), hipMemcpyDeviceToHost));
    if (cnt[5] != 0)
        return fail(HC_ERR_DEVICE, "split-column kernel: %llu mailbox exchanges timed out (internal error; results are invalid)", cnt[5]);
    // An attempt that exhausts the kernel's iteration budget is abandoned like a solve that gave up (the x0.8
    // retry rule applies); it is counted ([2], last place in [3]) and only fatal on request.
    if (cnt[2] != 0 && h->strict_guard)
        return fail(HC_ERR_DEVICE, "%llu BDF attempts hit the kernel's iteration guard (last: member %llu, row %llu)",
                    cnt[2], cnt[3] >> 24, cnt[3] & 0xFFFFFFull);
    return HC_OK;
}

int hc_get_counters(hc_handle *h, uint64_t *out4)
{
    if (!h || !out4) return fail(HC_ERR_ARG, "hc_get_counters: NULL argument");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    HIP_TRY(hipMemcpy(out4, h->counters.p, 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return HC_OK;
}

#ifdef HC_PROFILE
extern "C" int hc_debug_trace(hc_handle *h, double *out, int64_t n)
{
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (!h->trace.p) return fail(HC_ERR_ARG, "no trace (set HYDROCOL_DEBUG_TRACE)");
    HIP_TRY(hipMemcpy(out, h->trace.p, (size_t)n * 8, hipMemcpyDeviceToHost));
    return HC_OK;
}
extern "C" int hc_debug_profile(hc_handle *h, uint64_t *out32)
{
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    HIP_TRY(hipMemcpy(out32, h->counters.p + 8, 32 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return HC_OK;
}
extern "C" int hc_debug_profile_counts(hc_handle *h, uint64_t *out32)      // entries into each region
{
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    HIP_TRY(hipMemcpy(out32, h->counters.p + 64, 32 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return HC_OK;
}
extern "C" int hc_debug_profile_subcounts(hc_handle *h, uint64_t *out32)   // entries into sub-regions 32..63
{
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    HIP_TRY(hipMemcpy(out32, h->counters.p + 96, 32 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return HC_OK;
}
#endif

int hc_synchronize(hc_handle *h)
{
    if (!h) return fail(HC_ERR_ARG, "NULL handle");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return HC_OK;
}

// the moment tables exist once forcing and column are known: [n_points][3][n_rows]
static int ensure_moments(hc_handle *h)
{
    if (!h->have_forcing || !h->have_column) return fail(HC_ERR_ARG, "hc_set_column and hc_set_forcing must come first");
    if (h->moments_points != h->n_points) {
        const size_t cnt = (size_t)h->n_points * 3 * h->n_rows;
        if (h->moments.ensure(cnt)) return HC_ERR_DEVICE;
        HIP_TRY(hipMemset(h->moments.p, 0, cnt * 8));
        h->moments_points = h->n_points;
    }
    return HC_OK;
}

int hc_get_moments(hc_handle *h, int64_t *moments)
{
    if (!h || !moments) return fail(HC_ERR_ARG, "hc_get_moments: bad argument");
    HIP_TRY(hipSetDevice(h->device));
    if (int rc = ensure_moments(h)) return rc;
    HIP_TRY(hipStreamSynchronize(h->stream));
    HIP_TRY(hipMemcpy(moments, h->moments.p, (size_t)h->n_points * 3 * h->n_rows * 8, hipMemcpyDeviceToHost));
    return HC_OK;
}

int hc_export_moments(hc_handle *h, void *device_dst)
{
    if (!h || !device_dst) return fail(HC_ERR_ARG, "hc_export_moments: bad argument");
    HIP_TRY(hipSetDevice(h->device));
    if (int rc = ensure_moments(h)) return rc;
    HIP_TRY(hipStreamSynchronize(h->stream));
    HIP_TRY(hipMemcpy(device_dst, h->moments.p, (size_t)h->n_points * 3 * h->n_rows * 8, hipMemcpyDeviceToDevice));
    HIP_TRY(hipDeviceSynchronize());
    return HC_OK;
}

int hc_set_moments(hc_handle *h, const int64_t *moments)
{
    if (!h || !moments) return fail(HC_ERR_ARG, "hc_set_moments: bad argument");
    HIP_TRY(hipSetDevice(h->device));
    if (int rc = ensure_moments(h)) return rc;
    HIP_TRY(hipMemcpy(h->moments.p, moments, (size_t)h->n_points * 3 * h->n_rows * 8, hipMemcpyHostToDevice));
    return HC_OK;
}

int hc_reset_moments(hc_handle *h)
{
    if (!h) return fail(HC_ERR_ARG, "hc_reset_moments: bad argument");
    HIP_TRY(hipSetDevice(h->device));
    h->moments_points = 0;
    return ensure_moments(h);
}

// The path's one collective without torch: a single process that drives several devices (one handle each) sums the
// handles' moment tables over RCCL.  RCCL is bound at run time (dlopen), so the library loads on boxes without it and
// a process that already carries an RCCL (torch's) keeps using that one.
namespace {
struct Rccl {
    void *so = nullptr;
    int (*CommInitAll)(void **, int, const int *) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
};
int load_rccl(Rccl &r)
{
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        r.so = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        if (r.so) break;
    }
    if (!r.so) return fail(HC_ERR_UNSUPPORTED, "RCCL is not loadable (%s)", dlerror());
    r.CommInitAll = reinterpret_cast<decltype(r.CommInitAll)>(dlsym(r.so, "ncclCommInitAll"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(r.so, "ncclCommDestroy"));
    r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(dlsym(r.so, "ncclGroupStart"));
    r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(dlsym(r.so, "ncclGroupEnd"));
    r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(dlsym(r.so, "ncclAllReduce"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(r.so, "ncclGetErrorString"));
    if (!r.CommInitAll || !r.CommDestroy || !r.GroupStart || !r.GroupEnd || !r.AllReduce || !r.GetErrorString)
        return fail(HC_ERR_UNSUPPORTED, "the RCCL library lacks an entry point this call needs");
    return HC_OK;
}
}  // namespace

int hc_allreduce_moments(hc_handle **handles, int n)
{
    if (!handles || n < 1 || n > 64) return fail(HC_ERR_ARG, "hc_allreduce_moments: bad argument");
    size_t count = 0;
    std::vector<int> devs((size_t)n);
    for (int k = 0; k < n; k++) {
        hc_handle *h = handles[k];
        if (!h) return fail(HC_ERR_ARG, "handle %d is NULL", k);
        HIP_TRY(hipSetDevice(h->device));
        if (int rc = ensure_moments(h)) return rc;
        const size_t c = (size_t)h->n_points * 3 * h->n_rows;
        if (k == 0) count = c;
        if (c != count) return fail(HC_ERR_ARG, "handle %d holds a moment table of another shape", k);
        for (int j = 0; j < k; j++)
            if (handles[j]->device == h->device) return fail(HC_ERR_ARG, "handles %d and %d share device %d: one handle per device", j, k, h->device);
        devs[(size_t)k] = h->device;
        HIP_TRY(hipStreamSynchronize(h->stream));
    }
    Rccl r;
    if (int rc = load_rccl(r)) return rc;
    constexpr int NCCL_INT64 = 4, NCCL_SUM = 0;     // rccl.h: ncclInt64, ncclSum
    std::vector<void *> comms((size_t)n, nullptr);
    int e = r.CommInitAll(comms.data(), n, devs.data());
    if (e != 0) return fail(HC_ERR_DEVICE, "ncclCommInitAll failed: %s", r.GetErrorString(e));
    int bad = 0;
    bad = bad ? bad : r.GroupStart();
    for (int k = 0; k < n && !bad; k++) {
        (void)hipSetDevice(handles[k]->device);
        bad = r.AllReduce(handles[k]->moments.p, handles[k]->moments.p, count, NCCL_INT64, NCCL_SUM, comms[(size_t)k],
                          handles[k]->stream);
    }
    const int ge = r.GroupEnd();
    bad = bad ? bad : ge;
    hipError_t he = hipSuccess;
    for (int k = 0; k < n; k++) {
        (void)hipSetDevice(handles[k]->device);
        const hipError_t e1 = hipStreamSynchronize(handles[k]->stream);
        he = he == hipSuccess ? e1 : he;
    }
    for (int k = 0; k < n; k++) (void)r.CommDestroy(comms[(size_t)k]);
    if (bad) return fail(HC_ERR_DEVICE, "RCCL all-reduce of the moment tables failed: %s", r.GetErrorString(bad));
    HIP_TRY(he);
    return HC_OK;
}

int hc_get_noise_scale(hc_handle *h, double *scale, int64_t first, int64_t count)
{
    if (!h || !scale || first < 0 || count < 0 || first + count > h->n_members || !h->nscale.p)
        return fail(HC_ERR_ARG, "hc_get_noise_scale: bad argument");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    HIP_TRY(hipMemcpy(scale, h->nscale.p + first, (size_t)count * 8, hipMemcpyDeviceToHost));
    return HC_OK;
}

int hc_set_noise_scale(hc_handle *h, const double *scale, int64_t first, int64_t count)
{
    if (!h || !scale || first < 0 || count < 0 || first + count > h->n_members || !h->nscale.p)
        return fail(HC_ERR_ARG, "hc_set_noise_scale: bad argument");
    if (!h->philox) return fail(HC_ERR_ARG, "hc_set_noise_scale: call hc_set_noise_philox first (it resets the scales to 1)");
    for (int64_t k = 0; k < count; k++)
        if (!(scale[k] > 0.0) || !(scale[k] <= 1.0))
            return fail(HC_ERR_ARG, "noise scale %lld = %g is not a product of 0.8 factors in (0, 1]", (long long)(first + k), scale[k]);
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    HIP_TRY(hipMemcpy(h->nscale.p + first, scale, (size_t)count * 8, hipMemcpyHostToDevice));
    return HC_OK;
}

int hc_set_point_member_bases(hc_handle *h, const int64_t *base)
{
    if (!h) return fail(HC_ERR_ARG, "NULL handle");
    if (!h->have_column) return fail(HC_ERR_ARG, "hc_set_column must come first");
    h->base_host.clear();
    if (base) {
        for (int k = 0; k < h->n_points; k++)
            if (base[k] < 0) return fail(HC_ERR_ARG, "member base of point %d is negative", k);
        h->base_host.assign(base, base + h->n_points);
    }
    return HC_OK;
}

int hc_get_point_costs(hc_handle *h, uint64_t *cost)
{
    if (!h || !cost) return fail(HC_ERR_ARG, "hc_get_point_costs: bad argument");
    if (!h->have_column) return fail(HC_ERR_ARG, "hc_set_column must come first");
    for (int k = 0; k < h->n_points; k++)
        cost[k] = (size_t)k < h->cost_total.size() ? h->cost_total[(size_t)k] : 0ull;
    return HC_OK;
}

int hc_set_generic_exponents(hc_handle *h, int32_t on)
{
    if (!h) return fail(HC_ERR_ARG, "NULL handle");
    h->force_generic = on != 0;
    return HC_OK;
}

int hc_set_rows_per_launch(hc_handle *h, int32_t rows)
{
    if (!h || rows < 0) return fail(HC_ERR_ARG, "hc_set_rows_per_launch: bad argument");
    h->rows_per_launch = rows;
    return HC_OK;
}

int hc_set_iteration_budget(hc_handle *h, int32_t phase_steps)
{
    if (!h || phase_steps < 1) return fail(HC_ERR_ARG, "hc_set_iteration_budget: bad argument");
    h->max_phase_iterations = phase_steps;
    return HC_OK;
}

int hc_set_scipy_152(hc_handle *h, int32_t on)
{
    if (!h) return fail(HC_ERR_ARG, "hc_set_scipy_152: NULL handle");
    h->scipy_152 = on ? 1 : 0;
    return HC_OK;
}

int hc_rhs(hc_handle *h, int64_t row, int32_t spinup, double *dydt, double *aux)
{
    if (!h || !dydt) return fail(HC_ERR_ARG, "hc_rhs: NULL argument");
    StepArgs A;
    int rc = fill_args(h, A);
    if (rc) return rc;
    if (row < 0 || row >= h->n_rows) return fail(HC_ERR_ARG, "row out of range");
    if (h->n_points != 1) return fail(HC_ERR_UNSUPPORTED, "hc_rhs (test hook) serves one parameter point");
    HIP_TRY(hipSetDevice(h->device));
    const int D = h->p.dim_d;
    const size_t n = (size_t)h->n_members * D, na = (size_t)h->n_members * (3 * (D - 1) + 1);
    if (h->scratch_d.ensure(n + (aux ? na : 0))) return HC_ERR_DEVICE;
    A.spinup = spinup;
    rc = push_io(h);
    if (rc) return rc;
#ifdef HC_PROFILE
    if (const char *e = getenv("HYDROCOL_RHS_REPEAT")) A.n_rows = atoll(e);
#endif
    rc = launch_rhs(h, A, row, h->scratch_d.p, aux ? h->scratch_d.p + n : nullptr);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(h->stream));
    HIP_TRY(hipMemcpy(dydt, h->scratch_d.p, n * 8, hipMemcpyDeviceToHost));
    if (aux) HIP_TRY(hipMemcpy(aux, h->scratch_d.p + n, na * 8, hipMemcpyDeviceToHost));
    return HC_OK;
}

int hc_model_nodes(hc_handle *h, double *out, double *qinf)
{
    if (!h || !out) return fail(HC_ERR_ARG, "hc_model_nodes: NULL argument");
    StepArgs A;
    int rc = fill_args(h, A);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(h->device));
    const int D = h->p.dim_d;
    const size_t n = (size_t)h->n_members * D;
    if (h->scratch_d.ensure(4 * n + (size_t)h->n_members)) return HC_ERR_DEVICE;
    rc = push_io(h);
    if (rc) return rc;
    hipLaunchKernelGGL(model_nodes_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, A,
                       h->node_tabs.p, (int)h->use_special(), h->scratch_d.p, h->scratch_d.p + 4 * n);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(h->stream));
    HIP_TRY(hipMemcpy(out, h->scratch_d.p, 4 * n * 8, hipMemcpyDeviceToHost));
    if (qinf) HIP_TRY(hipMemcpy(qinf, h->scratch_d.p + 4 * n, (size_t)h->n_members * 8, hipMemcpyDeviceToHost));
    return HC_OK;
}

int hc_plugin_eval(int device_ordinal, const hc_column_params *p, int64_t n_cells, int64_t n_cols,
                   const double *psi, const double *por, const double *meank, const double *noisec,
                   const double *n_rnd, double *out, double *qinf)
{
    if (!p || !psi || !por || !meank || !noisec || !n_rnd || !out || !qinf || n_cells < 1 || n_cols < 1)
        return fail(HC_ERR_ARG, "hc_plugin_eval: bad argument");
    if (!(p->n > 1.0) || !(p->alpha > 0.0) || !(p->dz > 0.0)) return fail(HC_ERR_ARG, "bad dz / n / alpha");
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return fail(HC_ERR_NO_DEVICE, "no HIP device visible (%s): the plugin call has no CPU path",
                    e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
    if (device_ordinal < 0 || device_ordinal >= count) return fail(HC_ERR_ARG, "device ordinal out of range");
    HIP_TRY(hipSetDevice(device_ordinal));
    ColumnDev P{};
    P.model = p->model;
    P.theta_res = p->theta_res; P.alpha = p->alpha; P.n = p->n; P.m = p->m; P.psi_sat = p->psi_sat;
    P.epsilon = p->epsilon; P.lambda = p->lambda_exp; P.sigma = p->sigma_noise; P.sat_soil = p->sat_soil;
    P.dz = p->dz; P.inv_dz = 1.0 / p->dz;
    P.mn_alpha = (p->m * p->n) * p->alpha;
    P.inv_m = 1.0 / p->m;
    const int special = (p->model == HC_MODEL_VRETTAS_FUNG && p->n == 2.0 && p->m == 0.5 && p->lambda_exp == 1.0);
    const size_t nk = (size_t)n_cells, tot = nk * (size_t)n_cols;
    DevBuf<double> in, res;
    if (in.ensure(tot + 4 * nk) || res.ensure(4 * tot + (size_t)n_cols)) {
        in.release(); res.release();
        return HC_ERR_DEVICE;
    }
    int rc = [&]() -> int {
        HIP_TRY(hipMemcpy(in.p, psi, tot * 8, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(in.p + tot, por, nk * 8, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(in.p + tot + nk, meank, nk * 8, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(in.p + tot + 2 * nk, noisec, nk * 8, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(in.p + tot + 3 * nk, n_rnd, nk * 8, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(plugin_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, 0, P, special,
                           (long long)n_cells, (long long)n_cols, in.p, in.p + tot, in.p + tot + nk,
                           in.p + tot + 2 * nk, in.p + tot + 3 * nk, res.p, res.p + 4 * tot);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipMemcpy(out, res.p, 4 * tot * 8, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(qinf, res.p + 4 * tot, (size_t)n_cols * 8, hipMemcpyDeviceToHost));
        return HC_OK;
    }();
    in.release();
    res.release();
    return rc;
}

}  // extern "C"
