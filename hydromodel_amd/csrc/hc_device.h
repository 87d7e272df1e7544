// hc_device.h -- device-side building blocks of the gfx950 column stepper.
//
// Execution model: ONE 64-lane wavefront integrates ONE ensemble member.  Lane l owns the
// CPL consecutive depth nodes [l*CPL, (l+1)*CPL) ("chunked" layout), so the 3-point stencil,
// prefix sums and the tridiagonal solve are lane-local except at chunk edges, which move
// through cross-lane shuffles.  Per-wave vectors live in registers (double v[CPL]) or in LDS
// as v[c*64 + lane] (conflict-free for every CPL).
//
// Reference anchors (paths relative to /root/reference/code/src):
//   model_cell      models/vrettas_fung.py:108-257, models/vanGenuchten.py:71-121, utilities.py:4-54
//   rhs_eval        richards_pde.py:82-160 (__call__), :172-395 (pde_fun), :414-476 (bc_fun),
//                   tree_roots.py:179-292 (efficiency), utilities.py:56-99 (find_wtd)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace hc {

constexpr int WAVE = 64;

// slot tables staged in LDS, [NTAB][64*CPL] doubles
// T_RDELTA (round 5): the refined reciprocal of delta = por - theta_res, the divisor of the effective saturation -- a constant
// of the cell that every evaluation otherwise recomputes (v_rcp_f64 + two Newton steps: 5 of the ~94 VALU instructions of a
// cell, one of them a quarter-rate one).  Filled on the DEVICE (fill_rdelta in hydrocol.hip: the same instruction sequence,
// the same bits), never by the host.  It is the LAST table: the kernels that do not read it stage the first NTAB - 1 only.
// Read where it pays (rdelta_table below): up to 5 cells per lane on one wave per member, +2.1 ... 2.5 % at D = 300 and +1.5 % at
// 241; at 6 cells per lane the LDS it takes from the factorisation costs more than it saves (-5 %), the split column is as
// short of LDS.  -DHC_RDELTA_TABLE=0: nobody reads it (A/B builds).
#ifndef HC_RDELTA_TABLE
#define HC_RDELTA_TABLE 1
#endif
enum { T_POR = 0, T_FC, T_WLT, T_ROOT, T_LOGM, T_INVM2, T_NOISEC, T_VALID, T_INVD1, T_RDELTA, NTAB };
// `slots` = table row stride = 64 x cells per lane x waves per member
// (the generic-exponent cell model, twice the working set, LOSES 12 % with the table at 5 cells per lane: default exponents only)
__host__ __device__ constexpr bool rdelta_table(int slots, bool special = true) { return HC_RDELTA_TABLE && special && slots <= 64 * 5; }
__host__ __device__ constexpr int ntab_lds(int slots, bool special = true) { return rdelta_table(slots, special) ? (int)NTAB : (int)NTAB - 1; }
// round 5: masks as factors in the ET sums, the lateral-flow total only when it is asked for (+0.5 ... 2.3 %, same bits).  In
// the two-waves-per-SIMD kernels only (compiled with -ffp-contract=on: source-determined bits); the one-wave kernels are
// compiled with hipcc's default contraction, where ANY change of the code's shape may move a fusion decision and with it
// last bits (it did: the digests at D = 401 ... 512 changed with this edit and came back without it), so they keep their code.
// two selects of the default-exponent cell model replaced by exact identities (+0.4 ... 0.7 %, same bits; two-wave kernels only,
// like HC_RHS_DIET below)
#ifndef HC_MODEL_DIET
#define HC_MODEL_DIET 1
#endif
#ifndef HC_RHS_DIET
#define HC_RHS_DIET 1
#endif
// T_VALID = 1.0 in the slots of the D-1 midpoints, 0.0 beyond (a lane mask as data: a mask proper is an SGPR pair
// the compiler spills and reloads with two v_readlane per use), T_INVD1 = 1/(por - wlt) (a zero denominator counts as 1)
// per-slot integer tables, [NGTAB][64*CPL]
enum { G_SELF = 0, G_PREV, G_NEXT, NGTAB };

struct ColumnDev {
    int D, model, flag_et, flag_lf, flag_hlift, n_root_first, n_root_int, n_groups;
    double theta_res, alpha, n, m, psi_sat, epsilon, lambda, sigma, sat_soil, dz, inv_dz;
    double ipsi50, lai, surface_evap, interception, evap_delta_min;
    double mn_alpha;    // (m*n)*alpha
    double inv_m;       // 1/m
    double por_node0;   // porosity the top-node BC call sees
    // repaired PREDICT mode (richards_pde.py:312-351; an extension, the reference raises TypeError at :327-330):
    // low_lim = dim_d - (sat_cells - 1) of the interior call (k = D - 2 cells) as an int, 0 when not positive;
    // predict_first = 1 when the single-cell first-midpoint call has low_lim >= 1 (sat_cells <= 1)
    int flag_predict, predict_low, predict_first, pad_;   // read only by the kernels built with PREDICT = true
};

struct RowDev {
    double precip, atm;
    int daylight, wtd_obs, spinup;
    int diag;   // also integrate transpiration / lateral flow of the interior call (pde_model.arg_out)
    int wet;    // PREDICT mode: month in {10, 11, 12, 1, 2, 3} (richards_pde.py:315)
};

// ---------------------------------------------------------------- wave primitives
__device__ __forceinline__ double readlane_d(double v, int lane /*uniform*/)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, lane);
    hi = __builtin_amdgcn_readlane(hi, lane);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double uniform_d(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readfirstlane(lo);
    hi = __builtin_amdgcn_readfirstlane(hi);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ int uniform_i(int v) { return __builtin_amdgcn_readfirstlane(v); }

// DPP move of a double: lanes without a source keep `old`
template <int CTRL, int ROWMASK = 0xf>
__device__ __forceinline__ double dpp_d(double v, double old)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(__double2loint(old), lo, CTRL, ROWMASK, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(__double2hiint(old), hi, CTRL, ROWMASK, 0xf, false);
    return __hiloint2double(hi, lo);
}
constexpr int DPP_QUAD_XOR1 = 0xB1, DPP_QUAD_XOR2 = 0x4E, DPP_ROW_HALF_MIRROR = 0x141, DPP_ROW_MIRROR = 0x140,
              DPP_ROW_BCAST15 = 0x142, DPP_ROW_BCAST31 = 0x143, DPP_WAVE_SHR1 = 0x138, DPP_WAVE_SHL1 = 0x130;
template <int N> struct dpp_row_shr { static constexpr int value = 0x110 + N; };

// DPP move of a double where every lane without a source receives 0 (bound_ctrl): no `old` operand to
// initialise, which saves two v_mov_b32 per step over dpp_d<>(v, 0.0)
template <int CTRL, int ROWMASK = 0xf>
__device__ __forceinline__ double dpp_z(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROWMASK, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROWMASK, 0xf, true);
    return __hiloint2double(hi, lo);
}

// wave-wide sum, result uniform (no LDS traffic: 6 DPP steps + readlane)
__device__ __forceinline__ double wave_sum(double v)
{
    v += dpp_z<DPP_QUAD_XOR1>(v);
    v += dpp_z<DPP_QUAD_XOR2>(v);
    v += dpp_z<DPP_ROW_HALF_MIRROR>(v);
    v += dpp_z<DPP_ROW_MIRROR>(v);
    v += dpp_z<DPP_ROW_BCAST15, 0xa>(v);
    v += dpp_z<DPP_ROW_BCAST31, 0xc>(v);
    return readlane_d(v, WAVE - 1);
}
// two sums at once: the chains are written step by step so that each one fills the other's DPP wait states
__device__ __forceinline__ void wave_sum2(double &a, double &b)
{
    double u = a, v = b;
    u += dpp_z<DPP_QUAD_XOR1>(u);
    v += dpp_z<DPP_QUAD_XOR1>(v);
    u += dpp_z<DPP_QUAD_XOR2>(u);
    v += dpp_z<DPP_QUAD_XOR2>(v);
    u += dpp_z<DPP_ROW_HALF_MIRROR>(u);
    v += dpp_z<DPP_ROW_HALF_MIRROR>(v);
    u += dpp_z<DPP_ROW_MIRROR>(u);
    v += dpp_z<DPP_ROW_MIRROR>(v);
    u += dpp_z<DPP_ROW_BCAST15, 0xa>(u);
    v += dpp_z<DPP_ROW_BCAST15, 0xa>(v);
    u += dpp_z<DPP_ROW_BCAST31, 0xc>(u);
    v += dpp_z<DPP_ROW_BCAST31, 0xc>(v);
    a = readlane_d(u, WAVE - 1);
    b = readlane_d(v, WAVE - 1);
}
// three sums at once (the fused solve + norm exchange of the split-column kernel)
__device__ __forceinline__ void wave_sum3(double &a, double &b, double &c)
{
    double u = a, v = b, w = c;
#define HC_SUM3_STEP(CTRL, ...)           \
    u += dpp_z<CTRL, ##__VA_ARGS__>(u);   \
    v += dpp_z<CTRL, ##__VA_ARGS__>(v);   \
    w += dpp_z<CTRL, ##__VA_ARGS__>(w);
    HC_SUM3_STEP(DPP_QUAD_XOR1)
    HC_SUM3_STEP(DPP_QUAD_XOR2)
    HC_SUM3_STEP(DPP_ROW_HALF_MIRROR)
    HC_SUM3_STEP(DPP_ROW_MIRROR)
    HC_SUM3_STEP(DPP_ROW_BCAST15, 0xa)
    HC_SUM3_STEP(DPP_ROW_BCAST31, 0xc)
#undef HC_SUM3_STEP
    a = readlane_d(u, WAVE - 1);
    b = readlane_d(v, WAVE - 1);
    c = readlane_d(w, WAVE - 1);
}
// exclusive prefix sum across lanes (Hillis-Steele inside 16-lane rows, then row broadcasts)
__device__ __forceinline__ double wave_excl_scan(double v, int lane)
{
    (void)lane;
    v += dpp_z<dpp_row_shr<1>::value>(v);
    v += dpp_z<dpp_row_shr<2>::value>(v);
    v += dpp_z<dpp_row_shr<4>::value>(v);
    v += dpp_z<dpp_row_shr<8>::value>(v);
    v += dpp_z<DPP_ROW_BCAST15, 0xa>(v);
    v += dpp_z<DPP_ROW_BCAST31, 0xc>(v);
    return dpp_z<DPP_WAVE_SHR1>(v);
}
// exclusive prefix sum of `scan` (its grand total comes for free from lane 63) and the plain sum of `sum`,
// interleaved like wave_sum2
__device__ __forceinline__ void wave_scan_and_sum(double &scan, double &scan_total, double &sum)
{
    double v = scan, u = sum;
    v += dpp_z<dpp_row_shr<1>::value>(v);
    u += dpp_z<DPP_QUAD_XOR1>(u);
    v += dpp_z<dpp_row_shr<2>::value>(v);
    u += dpp_z<DPP_QUAD_XOR2>(u);
    v += dpp_z<dpp_row_shr<4>::value>(v);
    u += dpp_z<DPP_ROW_HALF_MIRROR>(u);
    v += dpp_z<dpp_row_shr<8>::value>(v);
    u += dpp_z<DPP_ROW_MIRROR>(u);
    v += dpp_z<DPP_ROW_BCAST15, 0xa>(v);
    u += dpp_z<DPP_ROW_BCAST15, 0xa>(u);
    v += dpp_z<DPP_ROW_BCAST31, 0xc>(v);
    u += dpp_z<DPP_ROW_BCAST31, 0xc>(u);
    scan_total = readlane_d(v, WAVE - 1);
    sum = readlane_d(u, WAVE - 1);
    scan = dpp_z<DPP_WAVE_SHR1>(v);
}
// value of lane-1 / lane+1 (`fill` at the wave edge)
__device__ __forceinline__ double shfl_up1(double v, int lane, double fill)
{
    (void)lane;
    return dpp_d<DPP_WAVE_SHR1>(v, fill);
}
__device__ __forceinline__ double shfl_down1(double v, int lane, double fill)
{
    (void)lane;
    return dpp_d<DPP_WAVE_SHL1>(v, fill);
}

// d = a*b + c with the addend in an SGPR pair.  hipcc otherwise turns every Horner step with a 64-bit
// literal into two v_mov_b32 plus v_fmac_f64; gfx950 VOP3 cannot encode a 64-bit literal, but it can read
// one SGPR pair, and two s_mov_b32 are SALU work.  Pure VALU instruction: no wait-state obligations.
__device__ __forceinline__ double fma_s(double a, double b, double c_uniform)
{
    double d;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(c_uniform));
    return d;
}

// a/b to ~1 ulp without the IEEE scale/fixup ladder (operands far from the exponent limits)
__device__ __forceinline__ double fast_div(double a, double b)
{
    double r = __builtin_amdgcn_rcp(b);
    r = fma(fma(-b, r, 1.0), r, r);
    r = fma(fma(-b, r, 1.0), r, r);
    const double q = a * r;
    return fma(fma(-b, q, a), r, q);
}
__device__ __forceinline__ double fast_rcp(double b)
{
    double r = __builtin_amdgcn_rcp(b);
    r = fma(fma(-b, r, 1.0), r, r);
    return fma(fma(-b, r, 1.0), r, r);
}
// the reciprocal the cell model divides by: fast_div's first five operations (model_cells_* spell them out per cell)
__device__ __forceinline__ double refined_rcp(double b)
{
    double r = __builtin_amdgcn_rcp(b);
    double d = fma(-b, r, 1.0);
    r = fma(d, r, r);
    d = fma(-b, r, 1.0);
    return fma(d, r, r);
}
// value of lane (byte_addr / 4) mod 64 -- raw ds_bpermute, no index arithmetic.  Callers pass
// (lane +- d) * 4 and mask the lanes whose source falls outside the wave themselves.
__device__ __forceinline__ double bpermute_d(int byte_addr, double v)
{
    const int lo = __builtin_amdgcn_ds_bpermute(byte_addr, __double2loint(v));
    const int hi = __builtin_amdgcn_ds_bpermute(byte_addr, __double2hiint(v));
    return __hiloint2double(hi, lo);
}

// log(x) for finite normal x > 0: fdlibm's e_log kernel, ~1 ulp, no special-case ladder
__device__ __forceinline__ double log_pos(double x)
{
    double m = __builtin_amdgcn_frexp_mant(x);          // [0.5, 1)
    int e = __builtin_amdgcn_frexp_exp(x);
    const bool lo = m < 0.70710678118654752440;
    m = lo ? m + m : m;
    e = lo ? e - 1 : e;
    const double f = m - 1.0;
    const double s = fast_div(f, 2.0 + f);
    const double z = s * s, w = z * z;
    const double t1 = w * fma_s(w, fma_s(w, 1.531383769920937332e-01, 2.222219843214978396e-01), 3.999999999940941908e-01);
    const double t2 = z * fma_s(w, fma_s(w, fma_s(w, 1.479819860511658591e-01, 1.818357216161805012e-01),
                                         2.857142874366239149e-01), 6.666666666666735130e-01);
    const double R = t2 + t1;
    const double hfsq = 0.5 * f * f;
    const double dk = (double)e;
    return dk * 6.93147180369123816490e-01 - ((hfsq - (s * (hfsq + R) + dk * 1.90821492927058770002e-10)) - f);
}
// exp(x) for |x| < 700 (no overflow / underflow ladder): x = k ln2 + r, Taylor to r^13 (< 1e-17 rel)
__device__ __forceinline__ double exp_mid(double x)
{
    const double k = __builtin_rint(x * 1.4426950408889634);
    double r = fma(-k, 6.93147180369123816490e-01, x);
    r = fma(-k, 1.90821492927058770002e-10, r);
    double p = fma_s(r, 1.0 / 6227020800.0, 1.0 / 479001600.0);
    p = fma_s(p, r, 1.0 / 39916800.0);
    p = fma_s(p, r, 1.0 / 3628800.0);
    p = fma_s(p, r, 1.0 / 362880.0);
    p = fma_s(p, r, 1.0 / 40320.0);
    p = fma_s(p, r, 1.0 / 5040.0);
    p = fma_s(p, r, 1.0 / 720.0);
    p = fma_s(p, r, 1.0 / 120.0);
    p = fma_s(p, r, 1.0 / 24.0);
    p = fma_s(p, r, 1.0 / 6.0);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return ldexp(p, (int)k);
}
// sqrt(x) for 0 <= x < ~1e300, no denormal scaling (Newton on v_rsq_f64, two residual corrections)
__device__ __forceinline__ double sqrt_pos(double x)
{
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = 0.5 * y;
    const double r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    double d = fma(-g, g, x);
    g = fma(d, h, g);
    d = fma(-g, g, x);
    g = fma(d, h, g);
    return x == 0.0 ? 0.0 : g;
}
// 1/sqrt(x) for 1 <= x < 1e300 (x = 1 + (alpha psi)^2)
__device__ __forceinline__ double rsqrt_ge1(double x)
{
    double y = __builtin_amdgcn_rsq(x);
    const double e = fma(-x * y, y, 1.0);
    return fma(y * e, fma(e, 0.375, 0.5), y);
}

// ---------------------------------------------------------------- one or two wavefronts per member
// Columns deeper than 512 nodes are SPLIT over two cooperating wavefronts of one workgroup (waves 2p and 2p + 1): half h
// owns the nodes [h * 64 CPL, (h + 1) * 64 CPL), again CPL consecutive nodes per lane.  One wave cannot hold the working
// set of a 640-node column (hipcc 7.2: 1.5 KB of scratch per lane at 10 cells per lane); two waves hold it in two
// register files with 0.2 KB.  What crosses the cut -- the stencil's edge values, every reduction, the
// coupling of the two tridiagonal blocks -- goes through a mailbox in LDS: double-buffered payload, one sequence counter
// per half (volatile mailbox accesses between a workgroup-scope release fence and acquire fence), no s_barrier (the other pair of the workgroup runs another
// member with its own control flow).  Both halves execute the same sequence of exchanges: every branch that contains
// one is decided by values both halves hold identically.
struct PairBox {
    double data[2][2][8];      // [exchange parity][half][value]
    unsigned seq[2];           // exchanges posted by each half
    unsigned pad_[2];
};

template <int HALVES>
struct Comm;

// one wave per member: everything stays inside the wave, the compiler sees exactly the code it saw before
template <>
struct Comm<1> {
    static constexpr int H = 1;
    int half;                  // always 0
    static constexpr bool dead = false;
    __device__ __forceinline__ double sum(double v) const { return wave_sum(v); }
    __device__ __forceinline__ void sum2(double &a, double &b) const { wave_sum2(a, b); }
    __device__ __forceinline__ double up1(double v, int lane, double fill) { return shfl_up1(v, lane, fill); }
    __device__ __forceinline__ double down1(double v, int lane, double fill) { return shfl_down1(v, lane, fill); }
    __device__ __forceinline__ double first_row(double v) { return readlane_d(v, 0); }
    // N quantities at once: up[j] = previous node's lane of lastc[j], down[j] = next node's lane of firstc[j]
    template <int N>
    __device__ __forceinline__ void edges(const double (&lastc)[N], const double (&firstc)[N], const double (&fill)[N],
                                          int lane, double (&up)[N], double (&down)[N])
    {
#pragma unroll
        for (int j = 0; j < N; j++) {
            up[j] = shfl_up1(lastc[j], lane, fill[j]);
            down[j] = shfl_down1(firstc[j], lane, fill[j]);
        }
    }
    __device__ __forceinline__ bool any(bool p) { return __any(p); }
    __device__ __forceinline__ int or_bits(int m) { return m; }
    __device__ __forceinline__ int max_int(int v) { return v; }
};

template <>
struct Comm<2> {
    static constexpr int H = 2;
    int half;                  // 0: upper half of the column (nodes from 0), 1: lower half
    int lane;
    unsigned k;                // exchanges done so far (identical in both halves)
    typedef __attribute__((address_space(3))) PairBox LdsBox;     // (a generic pointer would turn every access into a FLAT one)
    LdsBox *box;
    int dead;                  // an exchange timed out (a bug, never the data): stop waiting, let the member run out
    unsigned long long *fault; // device counter of such events
    static constexpr int SPIN_LIMIT = 1 << 22;

    // Every lane of both waves gets the partner's N values.  `mine` are wave-uniform.
    // Lane 0 stores payload then sequence number (LDS executes a wave's instructions in order); the reader issues the
    // sequence load and the payload loads back to back -- one LDS round trip per poll -- and keeps the payload only if the
    // sequence number it came with is the awaited one (a payload loaded after an up-to-date sequence number is up to date).
    template <int N>
    __device__ __forceinline__ void xchg(const double (&mine)[N], double (&theirs)[N])
    {
        static_assert(N <= 8, "mailbox holds eight values per half");
        const unsigned p = k & 1u;
        // Release: whatever this wave wrote to LDS before the exchange (row 0 of a group evaluation, its noise vector) is
        // ordered before the sequence number that announces it.  The mailbox accesses themselves are volatile (kept in
        // program order among themselves; LDS executes one wave's instructions in order); the fence is what keeps the
        // compiler from moving a PLAIN LDS access of the surrounding code across them.  At workgroup scope on one CU
        // it is an s_waitcnt, no cache operation.
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) {
            volatile __attribute__((address_space(3))) double *out = box->data[p][half];
#pragma unroll
            for (int j = 0; j < N; j++) out[j] = mine[j];
            *(volatile __attribute__((address_space(3))) unsigned *)&box->seq[half] = k + 1u;
        }
        const volatile __attribute__((address_space(3))) unsigned *seq = &box->seq[half ^ 1];
        const volatile __attribute__((address_space(3))) double *in = box->data[p][half ^ 1];
        // everything that steers the loop is scalar: the awaited number, the polled one (readfirstlane) and the spin
        // count.  No sleep between polls (tools/dev/xchg_bench.hip: 270 cycles per exchange of one double against 560
        // with `s_sleep 1` and a per-lane exit flag inside the loop).
        const int want = __builtin_amdgcn_readfirstlane((int)k) + 1;
        k = (unsigned)want;
        if (__builtin_amdgcn_readfirstlane(dead)) {          // (an earlier exchange timed out: the member is running out)
#pragma unroll
            for (int j = 0; j < N; j++) theirs[j] = 0.0;
            return;
        }
        double v[N];
        bool timed_out = false;
        for (int spins = 0;;) {
            const int got = __builtin_amdgcn_readfirstlane((int)*seq);
#pragma unroll
            for (int j = 0; j < N; j++) v[j] = in[j];
            if (got >= want) break;
            if (++spins > SPIN_LIMIT) {          // every wave must reach an exit
                timed_out = true;
                break;
            }
        }
        // Acquire: the partner's plain LDS writes that preceded its post (read later through row0[] / nz_partner) are not
        // read before the poll that saw its sequence number
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#pragma unroll
        for (int j = 0; j < N; j++) theirs[j] = uniform_d(v[j]);
        if (timed_out) {
            dead = 1;
            if (lane == 0) atomicAdd(fault, 1ull);
        }
    }
    // upper half's value first: both halves form the same sum, bit for bit
    __device__ __forceinline__ double sum(double v)
    {
        const double mine[1] = {wave_sum(v)};
        double theirs[1];
        xchg(mine, theirs);
        return half == 0 ? mine[0] + theirs[0] : theirs[0] + mine[0];
    }
    __device__ __forceinline__ void sum2(double &a, double &b)
    {
        wave_sum2(a, b);
        const double mine[2] = {a, b};
        double theirs[2];
        xchg(mine, theirs);
        a = half == 0 ? mine[0] + theirs[0] : theirs[0] + mine[0];
        b = half == 0 ? mine[1] + theirs[1] : theirs[1] + mine[1];
    }
    // value of the previous / next node's lane across the cut: the lower half's lane 0 takes the upper half's lane 63
    // and the other way round; the column's outer ends take `fill`
    __device__ __forceinline__ double up1(double v, int lane_, double fill)
    {
        const double mine[1] = {readlane_d(v, WAVE - 1)};
        double theirs[1];
        xchg(mine, theirs);
        return shfl_up1(v, lane_, half == 0 ? fill : theirs[0]);
    }
    __device__ __forceinline__ double down1(double v, int lane_, double fill)
    {
        const double mine[1] = {readlane_d(v, 0)};
        double theirs[1];
        xchg(mine, theirs);
        return shfl_down1(v, lane_, half == 0 ? theirs[0] : fill);
    }
    template <int N>
    __device__ __forceinline__ void edges(const double (&lastc)[N], const double (&firstc)[N], const double (&fill)[N],
                                          int lane_, double (&up)[N], double (&down)[N])
    {
        static_assert(N <= 8, "one exchange");
        double mine[N], theirs[N];
#pragma unroll
        for (int j = 0; j < N; j++) mine[j] = half == 0 ? readlane_d(lastc[j], WAVE - 1) : readlane_d(firstc[j], 0);
        xchg(mine, theirs);
#pragma unroll
        for (int j = 0; j < N; j++) {
            up[j] = shfl_up1(lastc[j], lane_, half == 0 ? fill[j] : theirs[j]);
            down[j] = shfl_down1(firstc[j], lane_, half == 0 ? theirs[j] : fill[j]);
        }
    }
    __device__ __forceinline__ double first_row(double v)      // node 0's value, in both halves
    {
        const double mine[1] = {readlane_d(v, 0)};
        double theirs[1];
        xchg(mine, theirs);
        return half == 0 ? mine[0] : theirs[0];
    }
    __device__ __forceinline__ int or_bits(int m)              // m wave-uniform
    {
        const double mine[1] = {(double)m};
        double theirs[1];
        xchg(mine, theirs);
        return m | (int)theirs[0];
    }
    __device__ __forceinline__ bool any(bool p) { return or_bits(__any(p) ? 1 : 0) != 0; }
    __device__ __forceinline__ int max_int(int v)              // v wave-uniform
    {
        const double mine[1] = {(double)v};
        double theirs[1];
        xchg(mine, theirs);
        const int o = (int)theirs[0];
        return v > o ? v : o;
    }
};

// ---------------------------------------------------------------- Philox4x32-10 + Box-Muller
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                              uint32_t k0, uint32_t k1, uint32_t (&out)[4])
{
#pragma unroll
    for (int r = 0; r < 10; r++) {
        uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
// standard normal for (seed, member, draw, depth index i)
__device__ __forceinline__ double philox_normal(uint64_t seed, uint64_t member, uint32_t draw, uint32_t i)
{
    uint32_t r[4];
    philox4x32_10(i >> 1, draw, (uint32_t)member, (uint32_t)(member >> 32), (uint32_t)seed,
                  (uint32_t)(seed >> 32), r);
    uint64_t a = ((uint64_t)r[1] << 32) | r[0], b = ((uint64_t)r[3] << 32) | r[2];
    double u1 = ((double)(a >> 11) + 0.5) * (1.0 / 9007199254740992.0);
    double u2 = ((double)(b >> 11) + 0.5) * (1.0 / 9007199254740992.0);
    double rad = sqrt(-2.0 * log(u1));
    double s, c;
    sincospi(2.0 * u2, &s, &c);
    return (i & 1) ? rad * s : rad * c;
}

// ---------------------------------------------------------------- the plugin, one cell
// SPECIAL: vrettas_fung with n = 2, m = 1/2, lambda = 1 (the reference's input_parameters.json);
// pow() disappears.  Generic path keeps pow() and the model switch.
template <bool SPECIAL>
__device__ __forceinline__ void model_cell(const ColumnDev &P, double psi, double por, double inv_delta,
                                           double logm, double invm2, double noisec, double rnd,
                                           double &theta, double &K, double &C, double &kb, double &pfac)
{
    (void)inv_delta;
    const double delta = por - P.theta_res;
    const bool sat = psi >= P.psi_sat;
    const double ap = P.alpha * fabs(psi);
    if (SPECIAL)
        pfac = rsqrt_ge1(fma(ap, ap, 1.0));
    else
        pfac = pow(1.0 + pow(ap, P.n), -P.m);
    double q = fma(delta, pfac, P.theta_res);
    q = sat ? por : q;
    // a true division (1 - s is ill-conditioned near saturation, a 1-ulp reciprocal shows up in K_bkg):
    // fast_div is the compiler's own f64 division sequence minus the exponent scaling / fix-up steps
    double s = fast_div(q - P.theta_res, delta);
    s = fmin(fmax(s, 0.0), 1.0);
    if (SPECIAL || P.model == 0) {
        // K_bkg = exp(log(m^2/sqrt(v+m^2)) + sqrt(log(v/m^2+1))*rnd)  ==  exp(log m - Lt/2 + sqrt(Lt)*rnd),
        // Lt = log(v/m^2 + 1)   (utilities.py:10-19 restructured: one log, one sqrt, one exp)
        const double var = P.sigma * (1.0 - s);
        const double t = fma(var, invm2, 1.0);
        const double Lt = log_pos(t);
        const double sig = sqrt_pos(Lt);
        kb = exp_mid(fma(sig, rnd, fma(-0.5, Lt, logm)));
        kb = noisec < 0.0 ? P.sat_soil : kb;   // cell in no layer: vrettas_fung.py:143
        const double sl = SPECIAL ? s : pow(s, P.lambda);
        K = sl * kb;                            // <= kb because 0 <= sl <= 1: np.minimum(K, kbkg) is a no-op
    } else {
        // vanGenuchten.py:91-98
        kb = P.sat_soil;
        const double mth = pow(s, P.inv_m);
        K = fmin(kb * sqrt(s) * pow(1.0 - pow(1.0 - mth, P.m), P.n), kb);
    }
    K = sat ? kb : K;
    double s3, apn;
    if (SPECIAL) {
        s3 = s * s * s;
        apn = ap;
    } else {
        s3 = pow(s, P.inv_m + 1.0);
        apn = pow(ap, P.n - 1.0);
    }
    double c = P.mn_alpha * delta * s3 * apn;
    // saturated, below epsilon, NaN or infinite -> epsilon (vrettas_fung.py:243-249)
    c = (int(!sat) & int(c >= P.epsilon) & int(c < INFINITY)) ? c : P.epsilon;   // no short-circuit: no branch
    theta = q;
    C = c;
}

// The special-exponent cell model (n = 2, m = 1/2, lambda = 1; model_cell<true>) for the N cells a lane owns,
// written "vertically": every statement is applied to all N cells before the next one, so that the N
// independent dependency chains are interleaved in program order.  With one wave per SIMD nothing else hides
// the latency of a dependent fp64 instruction, and the compiler does not interleave the chains by itself.
// Same operations in the same order per cell as model_cell<true>.
#ifndef HC_MODEL_BATCH
#define HC_MODEL_BATCH 5
#endif
#ifndef HC_GENERIC_BATCH
#define HC_GENERIC_BATCH 5
#endif
#define HC_V(...)                         \
    _Pragma("unroll") for (int c = 0; c < N; c++) { __VA_ARGS__; }
template <int N, int SLOTS, bool DIET = false>
__device__ __forceinline__ void model_cells_special(const ColumnDev &P, const double *tab, int slot0,
                                                    const double *psi, const double *rnd, double *theta, double *K,
                                                    double *C, double *kbo, double *pfac)
{
    // tab + slot0 addresses cell 0 of the batch; cell c is WAVE slots further.  Each table is read where it
    // is first needed, not at the top: four more live vectors would push caller state out of the VGPR file.
    double por[N], invm2[N], logm[N], noisec[N];
    HC_V(por[c] = tab[T_POR * SLOTS + slot0 + c * WAVE])
    double delta[N], ap[N], a[N], b[N], d[N], e[N], g[N], h[N], s[N], Lt[N], f[N], z[N], w[N], t1[N], t2[N], dk[N];
    bool sat[N], lo[N];
    int ex[N];
    HC_V(delta[c] = por[c] - P.theta_res)
    HC_V(sat[c] = psi[c] >= P.psi_sat)
    HC_V(ap[c] = P.alpha * fabs(psi[c]))
    // pfac = rsqrt_ge1(1 + ap^2)
    HC_V(a[c] = fma(ap[c], ap[c], 1.0))
    HC_V(b[c] = __builtin_amdgcn_rsq(a[c]))
    HC_V(d[c] = -a[c] * b[c])
    HC_V(e[c] = fma(d[c], b[c], 1.0))
    HC_V(d[c] = b[c] * e[c])
    HC_V(e[c] = fma(e[c], 0.375, 0.5))
    HC_V(pfac[c] = fma(d[c], e[c], b[c]))
    HC_V(a[c] = fma(delta[c], pfac[c], P.theta_res))
    HC_V(theta[c] = sat[c] ? por[c] : a[c])
    // s = (theta - theta_res) / delta   (the next table read is issued here, a division ahead of its use)
    HC_V(invm2[c] = tab[T_INVM2 * SLOTS + slot0 + c * WAVE])
    HC_V(a[c] = theta[c] - P.theta_res)
    if constexpr (rdelta_table(SLOTS)) {
        HC_V(b[c] = tab[T_RDELTA * SLOTS + slot0 + c * WAVE])
    } else {
        HC_V(b[c] = __builtin_amdgcn_rcp(delta[c]))
        HC_V(d[c] = fma(-delta[c], b[c], 1.0))
        HC_V(b[c] = fma(d[c], b[c], b[c]))
        HC_V(d[c] = fma(-delta[c], b[c], 1.0))
        HC_V(b[c] = fma(d[c], b[c], b[c]))
    }
    HC_V(d[c] = a[c] * b[c])
    HC_V(e[c] = fma(-delta[c], d[c], a[c]))
    HC_V(s[c] = fma(e[c], b[c], d[c]))
    HC_V(s[c] = fmin(fmax(s[c], 0.0), 1.0))
    // t = 1 + sigma (1 - s) / m^2;  Lt = log_pos(t)
    HC_V(a[c] = P.sigma * (1.0 - s[c]))
    HC_V(a[c] = fma(a[c], invm2[c], 1.0))
    HC_V(b[c] = __builtin_amdgcn_frexp_mant(a[c]))
    HC_V(ex[c] = __builtin_amdgcn_frexp_exp(a[c]))
    HC_V(lo[c] = b[c] < 0.70710678118654752440)
    if constexpr (DIET) {   // b + b = ldexp(b, 1), ex - 1: the flag as an integer feeds both (one select instead of three)
        int lo_i[N];
        HC_V(lo_i[c] = lo[c] ? 1 : 0)
        HC_V(b[c] = ldexp(b[c], lo_i[c]))
        HC_V(ex[c] = ex[c] - lo_i[c])
    } else {
        HC_V(b[c] = lo[c] ? b[c] + b[c] : b[c])
        HC_V(ex[c] = lo[c] ? ex[c] - 1 : ex[c])
    }
    HC_V(f[c] = b[c] - 1.0)
    HC_V(a[c] = 2.0 + f[c])
    HC_V(b[c] = __builtin_amdgcn_rcp(a[c]))
    HC_V(d[c] = fma(-a[c], b[c], 1.0))
    HC_V(b[c] = fma(d[c], b[c], b[c]))
    HC_V(d[c] = fma(-a[c], b[c], 1.0))
    HC_V(b[c] = fma(d[c], b[c], b[c]))
    HC_V(d[c] = f[c] * b[c])
    HC_V(e[c] = fma(-a[c], d[c], f[c]))
    HC_V(g[c] = fma(e[c], b[c], d[c]))                  // g = s of fdlibm's log kernel
    HC_V(z[c] = g[c] * g[c])
    HC_V(w[c] = z[c] * z[c])
    HC_V(t1[c] = fma_s(w[c], 1.531383769920937332e-01, 2.222219843214978396e-01))
    HC_V(t2[c] = fma_s(w[c], 1.479819860511658591e-01, 1.818357216161805012e-01))
    HC_V(t1[c] = fma_s(w[c], t1[c], 3.999999999940941908e-01))
    HC_V(t2[c] = fma_s(w[c], t2[c], 2.857142874366239149e-01))
    HC_V(t1[c] = w[c] * t1[c])
    HC_V(t2[c] = fma_s(w[c], t2[c], 6.666666666666735130e-01))
    HC_V(t2[c] = z[c] * t2[c])
    HC_V(a[c] = t2[c] + t1[c])                          // R
    HC_V(h[c] = 0.5 * f[c] * f[c])                      // hfsq
    HC_V(dk[c] = (double)ex[c])
    HC_V(b[c] = g[c] * (h[c] + a[c]) + dk[c] * 1.90821492927058770002e-10)
    HC_V(b[c] = (h[c] - b[c]) - f[c])
    HC_V(Lt[c] = dk[c] * 6.93147180369123816490e-01 - b[c])
    // sig = sqrt_pos(Lt)
    HC_V(logm[c] = tab[T_LOGM * SLOTS + slot0 + c * WAVE])
    HC_V(noisec[c] = tab[T_NOISEC * SLOTS + slot0 + c * WAVE])
    HC_V(a[c] = __builtin_amdgcn_rsq(Lt[c]))
    HC_V(g[c] = Lt[c] * a[c])
    HC_V(h[c] = 0.5 * a[c])
    HC_V(d[c] = fma(-h[c], g[c], 0.5))
    HC_V(g[c] = fma(g[c], d[c], g[c]))
    HC_V(h[c] = fma(h[c], d[c], h[c]))
    HC_V(d[c] = fma(-g[c], g[c], Lt[c]))
    HC_V(g[c] = fma(d[c], h[c], g[c]))
    HC_V(d[c] = fma(-g[c], g[c], Lt[c]))
    HC_V(g[c] = fma(d[c], h[c], g[c]))
    // (capping rsq at 1e300 instead of this select is exact too and two instructions shorter, but measured -1.7 % at D = 300,
    //  +1.6 % at 241: the compiler's placement, not the instruction count, decides at this size -- left alone)
    HC_V(g[c] = Lt[c] == 0.0 ? 0.0 : g[c])
    // kb = exp_mid(sig * rnd - Lt / 2 + log m)
    HC_V(a[c] = fma(g[c], rnd[c], fma(-0.5, Lt[c], logm[c])))
    HC_V(dk[c] = __builtin_rint(a[c] * 1.4426950408889634))
    HC_V(a[c] = fma(-dk[c], 6.93147180369123816490e-01, a[c]))
    HC_V(a[c] = fma(-dk[c], 1.90821492927058770002e-10, a[c]))
    HC_V(b[c] = fma_s(a[c], 1.0 / 6227020800.0, 1.0 / 479001600.0))
    HC_V(b[c] = fma_s(b[c], a[c], 1.0 / 39916800.0))
    HC_V(b[c] = fma_s(b[c], a[c], 1.0 / 3628800.0))
    HC_V(b[c] = fma_s(b[c], a[c], 1.0 / 362880.0))
    HC_V(b[c] = fma_s(b[c], a[c], 1.0 / 40320.0))
    HC_V(b[c] = fma_s(b[c], a[c], 1.0 / 5040.0))
    HC_V(b[c] = fma_s(b[c], a[c], 1.0 / 720.0))
    HC_V(b[c] = fma_s(b[c], a[c], 1.0 / 120.0))
    HC_V(b[c] = fma_s(b[c], a[c], 1.0 / 24.0))
    HC_V(b[c] = fma_s(b[c], a[c], 1.0 / 6.0))
    HC_V(b[c] = fma(b[c], a[c], 0.5))
    HC_V(b[c] = fma(b[c], a[c], 1.0))
    HC_V(b[c] = fma(b[c], a[c], 1.0))
    HC_V(b[c] = ldexp(b[c], (int)dk[c]))
    HC_V(kbo[c] = noisec[c] < 0.0 ? P.sat_soil : b[c])   // cell in no layer: vrettas_fung.py:143
    HC_V(a[c] = s[c] * kbo[c])                           // <= kb: np.minimum(K, kbkg) is a no-op
    if constexpr (DIET) {
        // A saturated cell has theta = por, so theta - theta_res IS delta bit for bit and the division above returns exactly
        // 1.0 (q = delta r = 1 + e, the correction -delta e r brings it to 1 - e e' -> 1.0): s kb = kb, no select needed
        HC_V(K[c] = a[c])
    } else {
        HC_V(K[c] = sat[c] ? kbo[c] : a[c])
    }
    // C = m n alpha delta s^3 (alpha |psi|), epsilon when saturated / below epsilon / not finite
    HC_V(a[c] = s[c] * s[c] * s[c])
    HC_V(a[c] = P.mn_alpha * delta[c] * a[c] * ap[c])
    HC_V(C[c] = (int(!sat[c]) & int(a[c] >= P.epsilon) & int(a[c] < INFINITY)) ? a[c] : P.epsilon)
}

// ---- vertical log / exp building blocks (same kernels as log_pos / exp_mid, applied to N cells per statement).
// Scratch arrays _a, _b, _d, _e, _g, _h, _z, _w, _t1, _t2, _dk (double[N]), _lo (bool[N]), _ex (int[N]) must be in scope.
#define HC_VLOG(OUT, IN)                                                                     \
    HC_V(_b[c] = __builtin_amdgcn_frexp_mant(IN[c]))                                          \
    HC_V(_ex[c] = __builtin_amdgcn_frexp_exp(IN[c]))                                          \
    HC_V(_lo[c] = _b[c] < 0.70710678118654752440)                                             \
    HC_V(_b[c] = _lo[c] ? _b[c] + _b[c] : _b[c])                                              \
    HC_V(_ex[c] = _lo[c] ? _ex[c] - 1 : _ex[c])                                               \
    HC_V(_h[c] = _b[c] - 1.0)                                                                 \
    HC_V(_a[c] = 2.0 + _h[c])                                                                 \
    HC_V(_b[c] = __builtin_amdgcn_rcp(_a[c]))                                                 \
    HC_V(_d[c] = fma(-_a[c], _b[c], 1.0))                                                     \
    HC_V(_b[c] = fma(_d[c], _b[c], _b[c]))                                                    \
    HC_V(_d[c] = fma(-_a[c], _b[c], 1.0))                                                     \
    HC_V(_b[c] = fma(_d[c], _b[c], _b[c]))                                                    \
    HC_V(_d[c] = _h[c] * _b[c])                                                               \
    HC_V(_e[c] = fma(-_a[c], _d[c], _h[c]))                                                   \
    HC_V(_g[c] = fma(_e[c], _b[c], _d[c]))                                                    \
    HC_V(_z[c] = _g[c] * _g[c])                                                               \
    HC_V(_w[c] = _z[c] * _z[c])                                                               \
    HC_V(_t1[c] = fma_s(_w[c], 1.531383769920937332e-01, 2.222219843214978396e-01))           \
    HC_V(_t2[c] = fma_s(_w[c], 1.479819860511658591e-01, 1.818357216161805012e-01))           \
    HC_V(_t1[c] = fma_s(_w[c], _t1[c], 3.999999999940941908e-01))                             \
    HC_V(_t2[c] = fma_s(_w[c], _t2[c], 2.857142874366239149e-01))                             \
    HC_V(_t1[c] = _w[c] * _t1[c])                                                             \
    HC_V(_t2[c] = fma_s(_w[c], _t2[c], 6.666666666666735130e-01))                             \
    HC_V(_t2[c] = _z[c] * _t2[c])                                                             \
    HC_V(_a[c] = _t2[c] + _t1[c])                                                             \
    HC_V(_e[c] = 0.5 * _h[c] * _h[c])                                                         \
    HC_V(_dk[c] = (double)_ex[c])                                                             \
    HC_V(_b[c] = _g[c] * (_e[c] + _a[c]) + _dk[c] * 1.90821492927058770002e-10)               \
    HC_V(_b[c] = (_e[c] - _b[c]) - _h[c])                                                     \
    HC_V(OUT[c] = _dk[c] * 6.93147180369123816490e-01 - _b[c])
#define HC_VEXP(OUT, IN)                                                                     \
    HC_V(_dk[c] = __builtin_rint(IN[c] * 1.4426950408889634))                                 \
    HC_V(_a[c] = fma(-_dk[c], 6.93147180369123816490e-01, IN[c]))                             \
    HC_V(_a[c] = fma(-_dk[c], 1.90821492927058770002e-10, _a[c]))                             \
    HC_V(_b[c] = fma_s(_a[c], 1.0 / 6227020800.0, 1.0 / 479001600.0))                         \
    HC_V(_b[c] = fma_s(_b[c], _a[c], 1.0 / 39916800.0))                                       \
    HC_V(_b[c] = fma_s(_b[c], _a[c], 1.0 / 3628800.0))                                        \
    HC_V(_b[c] = fma_s(_b[c], _a[c], 1.0 / 362880.0))                                         \
    HC_V(_b[c] = fma_s(_b[c], _a[c], 1.0 / 40320.0))                                          \
    HC_V(_b[c] = fma_s(_b[c], _a[c], 1.0 / 5040.0))                                           \
    HC_V(_b[c] = fma_s(_b[c], _a[c], 1.0 / 720.0))                                            \
    HC_V(_b[c] = fma_s(_b[c], _a[c], 1.0 / 120.0))                                            \
    HC_V(_b[c] = fma_s(_b[c], _a[c], 1.0 / 24.0))                                             \
    HC_V(_b[c] = fma_s(_b[c], _a[c], 1.0 / 6.0))                                              \
    HC_V(_b[c] = fma(_b[c], _a[c], 0.5))                                                      \
    HC_V(_b[c] = fma(_b[c], _a[c], 1.0))                                                      \
    HC_V(_b[c] = fma(_b[c], _a[c], 1.0))                                                      \
    HC_V(OUT[c] = ldexp(_b[c], (int)_dk[c]))

// The cell model for arbitrary exponents (n > 1, lambda > 0; both plugins), vertical like model_cells_special.
// Every power x^y is exp(y log x) on the in-house kernels above, with one log per distinct base; x = 0 is patched to
// the limit 0 (all exponents here are positive).  Against libm's pow the result differs by |y log x| * 2^-53 ~ a few
// 1e-15 relative, two decades inside the 1e-11 parity tier of theta, C and K; libm's pow costs ~3.6x the whole column-step.
// Two logs / exps fewer per cell than the literal form (round 2), by algebra that stays inside those tiers:
//   * (alpha|psi|)^(n-1) = (alpha|psi|)^n / (alpha|psi|): a division instead of a second exp of the same log;
//   * powers of S_e: an unsaturated cell has S_e = (1 + (alpha|psi|)^n)^-m up to the rounding of the reference's
//     theta_res + delta * pfac - theta_res round trip, so log S_e = -m log(1 + (alpha|psi|)^n) (already at hand),
//     S_e^(1/m + 1) = pfac / (1 + (alpha|psi|)^n) and S_e^(1/m) = 1 / (1 + (alpha|psi|)^n).  Where that round trip's
//     cancellation noise is not negligible (very dry cells, S_e < ~1e-3) C is below epsilon and clamped to it on both
//     sides, and K enters with an absolute tolerance; saturated cells (S_e = 1) take C = epsilon, K = K_bkg anyway.
template <int N, int SLOTS>
__device__ __forceinline__ void model_cells_generic(const ColumnDev &P, const double *tab, int slot0,
                                                    const double *psi, const double *rnd, double *theta, double *K,
                                                    double *C, double *kbo, double *pfac)
{
    double _a[N], _b[N], _d[N], _e[N], _g[N], _h[N], _z[N], _w[N], _t1[N], _t2[N], _dk[N];
    bool _lo[N];
    int _ex[N];
    double por[N], delta[N], ap[N], Lap[N], apn[N], q1[N], Lq[N], x[N], s[N], y[N], u[N];
    bool sat[N], apz[N], sz[N];
    HC_V(por[c] = tab[T_POR * SLOTS + slot0 + c * WAVE])
    HC_V(delta[c] = por[c] - P.theta_res)
    HC_V(sat[c] = psi[c] >= P.psi_sat)
    HC_V(ap[c] = P.alpha * fabs(psi[c]))
    HC_V(apz[c] = ap[c] == 0.0)
    HC_V(x[c] = apz[c] ? 1.0 : ap[c])
    HC_VLOG(Lap, x)
    // (alpha|psi|)^n, then pfac = (1 + that)^-m
    HC_V(y[c] = P.n * Lap[c])
    HC_VEXP(apn, y)
    HC_V(apn[c] = apz[c] ? 0.0 : apn[c])
    HC_V(q1[c] = 1.0 + apn[c])
    HC_VLOG(Lq, q1)
    HC_V(y[c] = -P.m * Lq[c])
    HC_VEXP(u, y)
    HC_V(pfac[c] = u[c])
    HC_V(y[c] = fma(delta[c], pfac[c], P.theta_res))
    HC_V(theta[c] = sat[c] ? por[c] : y[c])
    // S_e = (theta - theta_res) / delta, clipped
    HC_V(y[c] = theta[c] - P.theta_res)
    HC_V(_b[c] = __builtin_amdgcn_rcp(delta[c]))
    HC_V(_d[c] = fma(-delta[c], _b[c], 1.0))
    HC_V(_b[c] = fma(_d[c], _b[c], _b[c]))
    HC_V(_d[c] = fma(-delta[c], _b[c], 1.0))
    HC_V(_b[c] = fma(_d[c], _b[c], _b[c]))
    HC_V(_d[c] = y[c] * _b[c])
    HC_V(_e[c] = fma(-delta[c], _d[c], y[c]))
    HC_V(s[c] = fma(_e[c], _b[c], _d[c]))
    HC_V(s[c] = fmin(fmax(s[c], 0.0), 1.0))
    HC_V(sz[c] = s[c] == 0.0)
    // 1 / (1 + (alpha|psi|)^n) = S_e^(1/m) of an unsaturated cell
    HC_V(_b[c] = __builtin_amdgcn_rcp(q1[c]))
    HC_V(_d[c] = fma(-q1[c], _b[c], 1.0))
    HC_V(_b[c] = fma(_d[c], _b[c], _b[c]))
    HC_V(_d[c] = fma(-q1[c], _b[c], 1.0))
    HC_V(x[c] = fma(_d[c], _b[c], _b[c]))                // x = 1 / q1
    if (P.model == 0) {
        // K_bkg = exp(log m - Lt/2 + sqrt(Lt) eps), Lt = log(1 + sigma (1 - S_e) / m^2); K = S_e^lambda K_bkg
        double invm2[N], logm[N], noisec[N], Lt[N], t[N];
        HC_V(invm2[c] = tab[T_INVM2 * SLOTS + slot0 + c * WAVE])
        HC_V(y[c] = P.sigma * (1.0 - s[c]))
        HC_V(t[c] = fma(y[c], invm2[c], 1.0))
        HC_VLOG(Lt, t)
        HC_V(logm[c] = tab[T_LOGM * SLOTS + slot0 + c * WAVE])
        HC_V(noisec[c] = tab[T_NOISEC * SLOTS + slot0 + c * WAVE])
        HC_V(u[c] = sqrt_pos(Lt[c]))
        HC_V(y[c] = fma(u[c], rnd[c], fma(-0.5, Lt[c], logm[c])))
        HC_VEXP(u, y)
        HC_V(kbo[c] = noisec[c] < 0.0 ? P.sat_soil : u[c])
        if (P.lambda == 1.0) {
            HC_V(u[c] = s[c])
        } else {
            HC_V(y[c] = sat[c] ? 0.0 : -(P.lambda * P.m) * Lq[c])
            HC_VEXP(u, y)
            HC_V(u[c] = sz[c] ? 0.0 : u[c])
        }
        HC_V(y[c] = u[c] * kbo[c])
        HC_V(K[c] = sat[c] ? kbo[c] : y[c])
    } else {
        // vanGenuchten.py:91-98: K = K_sat sqrt(S_e) (1 - (1 - S_e^(1/m))^m)^n, capped at K_sat
        double mth[N], v[N], t[N];
        bool vz[N];
        HC_V(mth[c] = sz[c] ? 0.0 : (sat[c] ? 1.0 : x[c]))
        HC_V(v[c] = 1.0 - mth[c])
        HC_V(vz[c] = !(v[c] > 0.0))
        HC_V(t[c] = vz[c] ? 1.0 : v[c])
        HC_VLOG(u, t)
        HC_V(y[c] = P.m * u[c])
        HC_VEXP(u, y)
        HC_V(v[c] = 1.0 - (vz[c] ? 0.0 : u[c]))
        HC_V(vz[c] = !(v[c] > 0.0))
        HC_V(t[c] = vz[c] ? 1.0 : v[c])
        HC_VLOG(u, t)
        HC_V(y[c] = P.n * u[c])
        HC_VEXP(u, y)
        HC_V(u[c] = vz[c] ? 0.0 : u[c])
        HC_V(kbo[c] = P.sat_soil)
        HC_V(y[c] = fmin(P.sat_soil * sqrt_pos(s[c]) * u[c], P.sat_soil))
        HC_V(K[c] = sat[c] ? kbo[c] : y[c])
    }
    // C = m n alpha delta S_e^(1/m + 1) (alpha|psi|)^(n-1) = m n alpha delta (pfac / q1) ((alpha|psi|)^n / (alpha|psi|))
    HC_V(u[c] = sz[c] ? 0.0 : pfac[c] * x[c])
    HC_V(_b[c] = __builtin_amdgcn_rcp(apz[c] ? 1.0 : ap[c]))
    HC_V(_a[c] = apz[c] ? 1.0 : ap[c])
    HC_V(_d[c] = fma(-_a[c], _b[c], 1.0))
    HC_V(_b[c] = fma(_d[c], _b[c], _b[c]))
    HC_V(_d[c] = fma(-_a[c], _b[c], 1.0))
    HC_V(_b[c] = fma(_d[c], _b[c], _b[c]))
    HC_V(_d[c] = apn[c] * _b[c])
    HC_V(_e[c] = fma(-_a[c], _d[c], apn[c]))
    HC_V(x[c] = fma(_e[c], _b[c], _d[c]))                // apn / ap  (apn = 0 where ap = 0)
    HC_V(y[c] = P.mn_alpha * delta[c] * u[c] * x[c])
    HC_V(C[c] = (int(!sat[c]) & int(y[c] >= P.epsilon) & int(y[c] < INFINITY)) ? y[c] : P.epsilon)
}

// deepest cell index with pred true, or -1: cells are (lane, c) -> index lane*CPL + c
// ONE_BALLOT (round 5, the two-waves-per-SIMD kernels): one ballot -- the deepest cell sits in the highest lane that has
// any -- instead of one per cell slot and a scalar max over them.  The same index; +1.8 % at D = 300, +0.5 ... 0.7 % at
// 241 / 361 where a second wave covers the extra VALU work, -0.5 ... -2.6 % on the one-wave kernels (D = 401 / 541), which
// keep the scalar form.
template <int CPL, bool ONE_BALLOT = false>
__device__ __forceinline__ int deepest_true(const bool (&pred)[CPL])
{
    if constexpr (ONE_BALLOT) {
        int bestc = -1;
#pragma unroll
        for (int c = 0; c < CPL; c++) bestc = pred[c] ? c : bestc;
        const unsigned long long m1 = __ballot(bestc >= 0);
        if (!m1) return -1;
        const int hi1 = 63 - __clzll((long long)m1);
        return hi1 * CPL + __builtin_amdgcn_readlane(bestc, hi1);
    }
    int best = -1;
#pragma unroll
    for (int c = 0; c < CPL; c++) {
        unsigned long long m = __ballot(pred[c]);
        if (m) {
            int hi = 63 - __clzll((long long)m);
            int cand = hi * CPL + c;
            best = cand > best ? cand : best;
        }
    }
    return uniform_i(best);
}

// region stamps inside rhs_eval (development builds, see HC_STAMP in hc_step.h): 24 cell model, 25 flux / hydraulic
// lift, 26 evapo-transpiration, 27 lateral flow, 28 top boundary, 29 assembly
#ifdef HC_PROFILE
#define HC_RHS_PROF_PARAM , unsigned long long *rprof, unsigned long long &rprof_t, int &rprof_slot
#define HC_RHS_PROF_ARG , prof_lds, prof_t, prof_slot
#define HC_RSTAMP(slot)                                                  \
    {                                                                    \
        const unsigned long long now_ = clock64();                       \
        if (lane == 0) {                                                 \
            rprof[rprof_slot] += (unsigned)(now_ - rprof_t);             \
            rprof[32 + (slot)] += 1;                                     \
        }                                                                \
        rprof_t = now_;                                                  \
        rprof_slot = (slot);                                             \
    }
#define HC_RSUB(slot)                                \
    {                                                \
        if (lane == 0) rprof[32 + (slot)] += 1;      \
    }
#define HC_RSUB_END()
#else
#define HC_RHS_PROF_PARAM
#define HC_RHS_PROF_ARG
#ifdef HC_MARKS
#define HC_RSTAMP(slot) asm volatile("; HCMARK %0" ::"n"(slot))
#define HC_RSUB(slot) asm volatile("; HCMARK %0" ::"n"(slot))
#define HC_RSUB_END() asm volatile("; HCMARK -2")
#else
#define HC_RSTAMP(slot)
#define HC_RSUB(slot)
#define HC_RSUB_END()
#endif
#endif
// Sub-regions (ids 32..63: entries counted, cycles stay with the enclosing region): 33-37 change_D at order 1-5, 41-45
// accept_update, 49-53 predictor, 56 ET interior call, 57 its water_k > 0 block, 58 the tot_x > 1 renormalisation,
// 59 ET first-midpoint call, 60 lateral-flow sink, 61 hydraulic lift, 62 FD-Jacobian step sizes, 63 group scatter

// ---------------------------------------------------------------- RHS of the method of lines
// y[c]   : state at node lane*CPL + c
// rnd[c] : scaled noise of the cell evaluated in slot c (midpoint lane*CPL+c; the virtual top-node
//          cell sits in the always-free slot (lane 63, c = CPL-1))
// f[c]   : dy/dt at node lane*CPL + c (0 beyond the grid)
// aux    : optional global pointer [3*(D-1)+1] receiving c | s | f at the midpoints and pL
// PREDICT: the repaired predictive lateral flow is compiled in (a template parameter, not a run-time flag: the
// branch costs 1.4 % of the monitoring-mode kernel through register allocation alone when it is merely present)
// top boundary, richards_pde.py:414-476: pL from the plugin's values at the top node
__device__ __forceinline__ double top_flux(const ColumnDev &P, const RowDev &R, double y_top, double kb_top,
                                           double th_top, double pf_top)
{
    const double qinf = fmin(2.0 * (P.por_node0 - th_top) * P.dz, kb_top);
    const double net = (1.0 - P.interception) * fabs(R.precip);
    double p = (y_top < P.psi_sat) ? fmin(net, qinf) : 0.0;
    if (!R.spinup) {
        const double q_min = P.theta_res + P.evap_delta_min * pf_top;
        const bool allow = (th_top > P.theta_res) && (q_min > P.theta_res);
        p = (allow && R.daylight) ? p - P.surface_evap : p;
    }
    return p;
}

// `tab` points at this wave's first table slot (the lower half of a split column starts 64 CPL slots in); table rows are
// 64 CPL x CommT::H slots apart.  With CommT = Comm<2> the wave holds one half of the column (see Comm above): the cut
// costs two mailbox exchanges per evaluation -- the edge states at the start, and in the middle the water-table
// candidates, the top flux (evaluated by the lower half, whose last slot is free) and the upper half's last cell.
template <int CPL, bool SPECIAL, bool PREDICT, bool ONE_BALLOT = false, class CommT>
__device__ __forceinline__ void rhs_eval(const ColumnDev &P, const RowDev &R, const double *tab,
                                         int lane, const double (&y)[CPL], const double (&rnd)[CPL],
                                         double (&f)[CPL], double *aux, double &diag_tr, double &diag_lf, CommT &comm HC_RHS_PROF_PARAM)
{
    constexpr int H = CommT::H;
    constexpr bool DIET = ONE_BALLOT && HC_RHS_DIET;           // (ONE_BALLOT = "a two-waves-per-SIMD kernel": see HC_RHS_DIET)
    constexpr bool MDIET = ONE_BALLOT && HC_MODEL_DIET;
    constexpr int SLOTS = WAVE * CPL * H;                      // table row stride
    const int hb = H == 2 ? comm.half * (WAVE * CPL) : 0;      // index of this wave's first node
    const bool last_half = H == 1 || comm.half == H - 1;
    const int D = P.D;
    const double half = 0.5 * P.dz;
    if (R.diag) {
        diag_tr = 0.0;
        diag_lf = 0.0;
    }
    double ym[CPL], dym[CPL], th[CPL], Kc[CPL], Cc[CPL], fl[CPL], sk[CPL];
    double y_top = readlane_d(y[0], 0), y_edge = 0.0;          // node 0; the first node below this wave's last one
    if constexpr (H == 2) {
        const double mine[1] = {y_top};
        double theirs[1];
        comm.xchg(mine, theirs);
        y_edge = comm.half == 0 ? theirs[0] : 0.0;
        y_top = comm.half == 0 ? y_top : theirs[0];
    }
    const double y_nf = shfl_down1(y[0], lane, y_edge);
    double kb_top = 0.0, th_top = 0.0, pf_top = 0.0;
    {
        double kbv[CPL], pfv[CPL];
#pragma unroll
        for (int c = 0; c < CPL; c++) {
            // No mask for cells beyond the last midpoint: y is 0 in the padding slots, the tables hold benign values
            // there, and C / flux / sink of those cells are zeroed before the assembly below.  Only the virtual
            // top-node cell (last slot of lane 63) needs its own argument.
            const double yn = (c + 1 < CPL) ? y[c + 1 < CPL ? c + 1 : c] : y_nf;
            double psi = 0.5 * (y[c] + yn);
            if (c == CPL - 1) psi = (lane == WAVE - 1 && last_half) ? y_top : psi;
            ym[c] = psi;
            {
                // the gradient is a rounded value of its own, as in the reference (richards_pde.py:566): it must not
                // be fused into the flux K (dy - 1) further down
#pragma clang fp contract(off)
                dym[c] = (yn - y[c]) * P.inv_dz;
            }
        }
        HC_RSTAMP(24);
        if (SPECIAL) {
            // batches of HC_MODEL_BATCH cells: enough independent chains to hide the fp64 latency, few
            // enough to keep the working set in VGPRs
            constexpr int B = HC_MODEL_BATCH < CPL ? HC_MODEL_BATCH : CPL;
            constexpr int NB = CPL / B, REM = CPL - NB * B;
#pragma unroll
            for (int q = 0; q < NB; q++)
                model_cells_special<B, SLOTS, MDIET>(P, tab, q * B * WAVE + lane, ym + q * B, rnd + q * B, th + q * B,
                                                     Kc + q * B, Cc + q * B, kbv + q * B, pfv + q * B);
            if (REM > 0)
                model_cells_special<(REM > 0 ? REM : 1), SLOTS, MDIET>(P, tab, NB * B * WAVE + lane, ym + NB * B, rnd + NB * B,
                                                                th + NB * B, Kc + NB * B, Cc + NB * B, kbv + NB * B,
                                                                pfv + NB * B);
        } else {
            // generic exponents: batches of HC_GENERIC_BATCH cells (the working set of one cell is twice the special one's)
            constexpr int B = HC_GENERIC_BATCH < CPL ? HC_GENERIC_BATCH : CPL;
            constexpr int NB = CPL / B, REM = CPL - NB * B;
#pragma unroll
            for (int q = 0; q < NB; q++)
                model_cells_generic<B, SLOTS>(P, tab, q * B * WAVE + lane, ym + q * B, rnd + q * B, th + q * B,
                                              Kc + q * B, Cc + q * B, kbv + q * B, pfv + q * B);
            if (REM > 0)
                model_cells_generic<(REM > 0 ? REM : 1), SLOTS>(P, tab, NB * B * WAVE + lane, ym + NB * B, rnd + NB * B,
                                                                th + NB * B, Kc + NB * B, Cc + NB * B, kbv + NB * B,
                                                                pfv + NB * B);
        }
        HC_RSTAMP(25);
#pragma unroll
        for (int c = 0; c < CPL; c++) {
            fl[c] = Kc[c] * (dym[c] - 1.0);
            sk[c] = 0.0;
        }
        kb_top = kbv[CPL - 1];
        th_top = th[CPL - 1];
        pf_top = pfv[CPL - 1];
    }
    const bool normal_mode = !R.spinup;
    // ---- hydraulic lift (night only), richards_pde.py:234-254
    if (P.flag_hlift && normal_mode && !R.daylight) {
        HC_RSUB(61);
        const double c_sat = 1800.0 * P.lai;
#pragma unroll
        for (int c = 0; c < CPL; c++) {
            const int i = hb + lane * CPL + c;
            const bool isr = (i == 0) ? (P.n_root_first > 0) : (i <= P.n_root_int);
            const double root = tab[T_ROOT * SLOTS + c * WAVE + lane];
            const double t1 = 1.0 - P.ipsi50 * ym[c];
            const double c_hr = c_sat * (t1 * t1) * root;
            const double add = 0.5 * c_hr * (dym[c] * P.dz);
            fl[c] += (isr && i < D - 1) ? add : 0.0;
        }
    }
    // ---- evapo-transpiration (daylight only), richards_pde.py:258-302 + tree_roots.py:213-291
    HC_RSTAMP(26);
    if (P.flag_et && normal_mode && R.daylight && (H == 1 || comm.half == 0)) {   // the root zone lies in the upper half
        // (a) interior call: midpoints 1..n_root_int, normalised together.
        // alpha_02 (tree_roots.py:245-265) is exactly 1 where theta > field capacity and 0 elsewhere:
        // inside (wlt, fc] the reference's (theta - fc)/(fc - wlt) is <= 0 and gets clipped to 0.
        if (P.n_root_int > 0) {
            HC_RSUB(56);
            bool isr[CPL];
            double pre[CPL], t_wlt[CPL], t_fc[CPL], t_invd1[CPL], t_root[CPL];
            double s_w = 0.0, s_t = 0.0;
            // table reads first and unconditional: a read under a per-lane condition becomes a divergent
            // branch with its own LDS wait
#pragma unroll
            for (int c = 0; c < CPL; c++) {
                const int slot = c * WAVE + lane;
                t_wlt[c] = tab[T_WLT * SLOTS + slot];
                t_fc[c] = tab[T_FC * SLOTS + slot];
                t_invd1[c] = tab[T_INVD1 * SLOTS + slot];
                t_root[c] = tab[T_ROOT * SLOTS + slot];
            }
            if constexpr (DIET) {
                // the root-zone mask as a 0 / 1 factor: fma(m, x, s) rounds x + s exactly as the masked add does (m = 1) and
                // returns s (m = 0, x finite) -- one instruction per term instead of two selects and an add
#pragma unroll
                for (int c = 0; c < CPL; c++) {
                    const int i = lane * CPL + c;
                    isr[c] = int(i >= 1) & int(i <= P.n_root_int);
                    const double m = isr[c] ? 1.0 : 0.0;
                    const double dw = th[c] - t_wlt[c];
                    s_w = fma(m, dw, s_w);
                    s_t = fma(m, th[c], s_t);
                    pre[c] = s_t;
                }
            } else {
#pragma unroll
                for (int c = 0; c < CPL; c++) {
                    const int i = lane * CPL + c;
                    isr[c] = int(i >= 1) & int(i <= P.n_root_int);
                    const double dw = th[c] - t_wlt[c];
                    s_w += isr[c] ? dw : 0.0;
                    s_t += isr[c] ? th[c] : 0.0;
                    pre[c] = s_t;
                }
            }
            // one pass: exclusive scan of the lane totals of theta (grand total from its last lane) + sum of s_w
            double excl = s_t, tot_theta, sum_w = s_w;
            wave_scan_and_sum(excl, tot_theta, sum_w);
            double water_k = sum_w * P.dz;
            double x_out[CPL];
            double tot_x = 0.0;
            if (water_k > 0.0) {
                HC_RSUB(57);
                double total = tot_theta * P.dz;
                total = total == 0.0 ? 1.0 : total;
                const double inv_total = fast_rcp(total);
                bool a2[CPL];
                bool all_one = true;
#pragma unroll
                for (int c = 0; c < CPL; c++) {
                    a2[c] = th[c] > t_fc[c];
                    all_one = int(all_one) & (int(!isr[c]) | int(a2[c]));
                }
                const double a2v = __all(all_one) ? 0.1 : 1.0;   // "roots drowning" guard, :270-272
                double s_r = 0.0, s_rx = 0.0;
                double rho[CPL];
#pragma unroll
                for (int c = 0; c < CPL; c++) {
                    const double local = (excl + pre[c]) * P.dz;
                    const double a1 = fmax(th[c] * t_invd1[c], local * inv_total);
                    rho[c] = (int(isr[c]) & int(a2[c])) ? fabs(a1 * a2v) : 0.0;
                    s_r += rho[c];
                    s_rx = fma(rho[c], t_root[c], s_rx);
                }
                // sum(rho) and sum(rho * root) together: sum(x_out) = sum(rho * root) / tot
                wave_sum2(s_r, s_rx);
                double tot = s_r * P.dz;
                tot = tot == 0.0 ? 1.0 : tot;
                const double inv_tot = fast_rcp(tot);
#pragma unroll
                for (int c = 0; c < CPL; c++)
                    x_out[c] = (rho[c] * inv_tot) * t_root[c];
                tot_x = (s_rx * inv_tot) * P.dz;
            } else {
                water_k = 0.0;
#pragma unroll
                for (int c = 0; c < CPL; c++) x_out[c] = 0.0;
            }
            if (tot_x > 1.0) {
                HC_RSUB(58);
                const double inv_tx = fast_rcp(tot_x);
                double s_x = 0.0;
#pragma unroll
                for (int c = 0; c < CPL; c++) {
                    x_out[c] = x_out[c] * inv_tx;
                    s_x += isr[c] ? x_out[c] : 0.0;
                }
                tot_x = wave_sum(s_x) * P.dz;
            }
            if (tot_x > 0.0) {
                const double tr_pot = fast_div(fmin(R.atm, water_k), tot_x);
#pragma unroll
                for (int c = 0; c < CPL; c++) sk[c] = isr[c] ? -(tr_pot * x_out[c]) : sk[c];
                if (R.diag) {   // richards_pde.py:380-381: sum(uptake) * dz of the interior call
                    double s_u = 0.0;
#pragma unroll
                    for (int c = 0; c < CPL; c++) s_u += isr[c] ? tr_pot * x_out[c] : 0.0;
                    diag_tr = wave_sum(s_u) * P.dz;
                }
            }
        }
        // (b) first-midpoint call: one cell normalised on its own
        if (P.n_root_first > 0) {
            HC_RSUB(59);
            // evaluated by every lane on its own slot-0 cell (no divergent branch); only lane 0's result is kept
            const double t0 = th[0], w0 = tab[T_WLT * SLOTS + lane], f0 = tab[T_FC * SLOTS + lane];
            const double water_k = fmax((t0 - w0) * P.dz, 0.0);
            const double local = t0 * P.dz;
            const double ratio = local == 0.0 ? 0.0 : 1.0;              // local / (local or 1): exactly 0 or 1
            const double a1 = fmax(t0 * tab[T_INVD1 * SLOTS + lane], ratio);
            // single cell: the all-ones guard always applies, alpha_02 * 0.1 (tree_roots.py:270-272)
            const double rho = (water_k > 0.0 && t0 > f0) ? fabs(a1 * 0.1) : 0.0;
            double tot = rho * P.dz;
            tot = tot == 0.0 ? 1.0 : tot;
            double x0 = fast_div(rho, tot) * tab[T_ROOT * SLOTS + lane];
            double tot_x = x0 * P.dz;
            const bool big = tot_x > 1.0;
            const double x0n = fast_div(x0, big ? tot_x : 1.0);
            x0 = big ? x0n : x0;
            tot_x = big ? x0 * P.dz : tot_x;
            const bool on = tot_x > 0.0;
            const double s0 = -(fast_div(fmin(R.atm, water_k), on ? tot_x : 1.0) * x0);
            sk[0] = (lane == 0 && on) ? s0 : sk[0];
        }
    }
    // ---- lateral flow, monitoring mode, richards_pde.py:352-376 (interior slice only; the
    //      single-cell first call can never satisfy wtd_est < wtd_obs)
    HC_RSTAMP(27);
    // split column: the top flux is the lower half's to compute (its last slot holds the top-node cell) and the
    // upper half's to use; the upper half's last cell is the lower half's upper neighbour in the assembly
    double pL_pair = 0.0, edge_C = 0.0, edge_f = 0.0, edge_s = 0.0, edge_ym = 0.0;
    int jstar_other = -1, jstar_mine = -1;
    if constexpr (H == 2) {
        double pl = 0.0;                          // (the upper half is the critical path in daylight: it skips this)
        if (comm.half == 1) pl = readlane_d(top_flux(P, R, y_top, kb_top, th_top, pf_top), WAVE - 1);
        bool unsat0[CPL];
#pragma unroll
        for (int c = 0; c < CPL; c++) {
            const int i = hb + lane * CPL + c;
            unsat0[c] = P.flag_lf && (i >= 1) && (i <= D - 2) && !(ym[c] >= P.psi_sat);
        }
        const int j0 = deepest_true<CPL, ONE_BALLOT>(unsat0);
        jstar_mine = j0 < 0 ? -1 : hb + j0;
        const double mine[5] = {(double)jstar_mine, comm.half == 0 ? readlane_d(Cc[CPL - 1], WAVE - 1) : pl,
                                readlane_d(fl[CPL - 1], WAVE - 1), readlane_d(sk[CPL - 1], WAVE - 1),
                                readlane_d(ym[CPL - 1], WAVE - 1)};
        double theirs[5];
        comm.xchg(mine, theirs);
        jstar_other = (int)theirs[0];
        if (comm.half == 0) {
            pL_pair = theirs[1];
        } else {
            edge_C = theirs[1];
            edge_f = theirs[2];
            edge_s = theirs[3];
            edge_ym = theirs[4];
        }
    }
    if (P.flag_lf) {
        const int k = D - 2;
        int jstar;                                            // midpoint index, local position p = j-1
        if constexpr (H == 2) {
            jstar = jstar_mine > jstar_other ? jstar_mine : jstar_other;     // (found before the exchange above)
        } else {
            bool unsat[CPL];
#pragma unroll
            for (int c = 0; c < CPL; c++) {
                const int i = hb + lane * CPL + c;
                unsat[c] = (i >= 1) && (i <= D - 2) && !(ym[c] >= P.psi_sat);
            }
            jstar = deepest_true<CPL, ONE_BALLOT>(unsat);
        }
        int wtd_est = jstar < 0 ? 0 : jstar;                  // p* + 1
        wtd_est = wtd_est < k - 1 ? wtd_est : k - 1;
        const int wtd_obs = R.wtd_obs < k - 1 ? R.wtd_obs : k - 1;
        if (PREDICT && P.flag_predict) {   // (a sweep may mix predictive and monitoring-mode points)
            // Predictive mode, richards_pde.py:312-351 with `low_lim` as an int (clamped at 0): ONE cell, the
            // estimated water table itself, drains with alpha_low (1 - (j / low_lim) ** nu[j]),
            // nu = linspace(1.5, 0, low_lim).  All of it is wave-uniform scalar work.
            const double alpha_low = R.wet ? -2.5e-3 : -1.5e-3;
            if (wtd_est < P.predict_low) {
                double pw = 0.0;                                      // 0 ** 1.5
                if (wtd_est > 0) {
                    double nu;
                    {
#pragma clang fp contract(off)
                        const double step = -1.5 / (double)(P.predict_low - 1);   // numpy linspace: delta / (num - 1)
                        nu = (double)wtd_est * step + 1.5;
                    }
                    nu = wtd_est == P.predict_low - 1 ? 0.0 : nu;     // linspace pins the last sample to `stop`
                    pw = exp_mid(nu * log_pos((double)wtd_est / (double)P.predict_low));
                }
                const double alpha_lat = alpha_low * (1.0 - pw);
                double s_l = 0.0;
#pragma unroll
                for (int c = 0; c < CPL; c++) {
                    const int p = hb + lane * CPL + c - 1;
                    const bool in = p == wtd_est;
                    sk[c] = in ? fmin(alpha_lat * ym[c], sk[c]) : sk[c];
                    s_l += in ? fabs(sk[c]) : 0.0;
                }
                if (H == 2 && hb - 2 == wtd_est) edge_s = fmin(alpha_lat * edge_ym, edge_s);   // the upper half's last cell
                if (R.diag) diag_lf = comm.sum(s_l) * P.dz;
            }
            if (P.predict_first && (H == 1 || comm.half == 0))   // single-cell call: wtd_est = 0 < low_lim, alpha_lat = alpha_low (1 - 0 ** 1.5)
                sk[0] = lane == 0 ? fmin(alpha_low * ym[0], sk[0]) : sk[0];
        } else if (wtd_est < wtd_obs) {
            HC_RSUB(60);
            double s_l = 0.0;
#pragma unroll
            for (int c = 0; c < CPL; c++) {
                const int p = hb + lane * CPL + c - 1;
                const bool in = (p >= wtd_est) && (p < wtd_obs);
                sk[c] = in ? fmin(-2.5e-4 * ym[c], sk[c]) : sk[c];
                if constexpr (!DIET) s_l += in ? fabs(sk[c]) : 0.0;
            }
            if (H == 2 && hb - 2 >= wtd_est && hb - 2 < wtd_obs) edge_s = fmin(-2.5e-4 * edge_ym, edge_s);
            if constexpr (DIET) {
                if (R.diag) {   // the sink's total is a diagnostic: summed only when asked for (richards_pde.py:374,388)
#pragma unroll
                    for (int c = 0; c < CPL; c++) {
                        const int p = hb + lane * CPL + c - 1;
                        s_l += ((p >= wtd_est) && (p < wtd_obs)) ? fabs(sk[c]) : 0.0;
                    }
                    diag_lf = comm.sum(s_l) * P.dz;
                }
            } else {
                if (R.diag) diag_lf = comm.sum(s_l) * P.dz;   // richards_pde.py:374,388
            }
        }
    }
    // ---- top boundary, richards_pde.py:414-476 (computed in lane 63's spare slot)
    HC_RSTAMP(28);
    double pL;
    if constexpr (H == 2) {
        pL = pL_pair;                                           // (only the upper half uses it)
    } else {
        const double qinf = fmin(2.0 * (P.por_node0 - th_top) * P.dz, kb_top);
        const double net = (1.0 - P.interception) * fabs(R.precip);
        double p = (y_top < P.psi_sat) ? fmin(net, qinf) : 0.0;
        if (normal_mode) {
            const double q_min = P.theta_res + P.evap_delta_min * pf_top;
            const bool allow = (th_top > P.theta_res) && (q_min > P.theta_res);
            p = (allow && R.daylight) ? p - P.surface_evap : p;
        }
        pL = readlane_d(p, WAVE - 1);
    }
    // ---- assemble dy/dt, richards_pde.py:108-156.  One formula for every node:
    //   dy/dt_i = (f_i - f_{i-1} + h (s_i + s_{i-1})) / (h (c_i + c_{i-1})),   h = dz/2,
    // with (c, s, f)_{-1} = (0, 0, -pL) at the top and (c, s, f)_{D-1} = 0 at the bottom, which is
    // :119 and :155 after cancelling the signs; padding nodes come out as 0/1 = 0.
    HC_RSTAMP(29);
#pragma unroll
    for (int c = 0; c < CPL; c++) {
        const double valid = tab[T_VALID * SLOTS + c * WAVE + lane];    // 1.0 / 0.0; the cell values are finite
        Cc[c] *= valid;
        fl[c] *= valid;
    }
    const bool upper = H == 1 || comm.half == 0;              // this wave starts at the column's top
    const double cP0 = shfl_up1(Cc[CPL - 1], lane, upper ? 0.0 : edge_C);
    const double sP0 = shfl_up1(sk[CPL - 1], lane, upper ? 0.0 : edge_s);
    const double fP0 = shfl_up1(fl[CPL - 1], lane, upper ? -pL : edge_f);
#pragma unroll
    for (int c = 0; c < CPL; c++) {
        const int i = hb + lane * CPL + c;
        const double cP = c == 0 ? cP0 : Cc[c > 0 ? c - 1 : 0];
        const double sP = c == 0 ? sP0 : sk[c > 0 ? c - 1 : 0];
        const double fP = c == 0 ? fP0 : fl[c > 0 ? c - 1 : 0];
        double den = half * Cc[c] + half * cP;
        const double num = (fl[c] - fP) + (half * sk[c] + half * sP);
        den = den == 0.0 ? 1.0 : den;
        f[c] = fast_div(num, den);
        if (aux && i < D - 1) {
            aux[i] = Cc[c];
            aux[(D - 1) + i] = sk[c];
            aux[2 * (D - 1) + i] = fl[c];
        }
    }
    if (aux && lane == 0) aux[3 * (D - 1)] = pL;
}

}  // namespace hc
