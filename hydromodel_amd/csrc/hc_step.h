// hc_step.h -- the fused column-step kernel: rows x (<=5 attempts) x variable-order BDF.
//
// Restates, per wavefront = per ensemble member:
//   Simulation.run row loop            /root/reference/code/src/simulation.py:576-626
//   RichardsPDE.solve (5 attempts)     /root/reference/code/src/richards_pde.py:478-537
//   scipy solve_ivp(method='BDF')      scipy/integrate/_ivp/bdf.py (BDF.__init__, _step_impl,
//                                      solve_bdf_system, change_D), common.py (select_initial_step,
//                                      norm, num_jac/_sparse_num_jac) -- third-party, restated
//
// Structure: a single loop with ONE inlined RHS site; a wave-uniform phase variable says who
// consumes the RHS value (initial slope, step-size probe, FD-Jacobian column group, Newton
// iterate).  All step-control decisions are wave-uniform scalars.  The linear solves with
// I - c*J use a lane-partitioned (Wang) tridiagonal factorisation: lane-local elimination,
// a 64-unknown reduced system solved by parallel cyclic reduction through shuffles, lane-local
// back substitution.
#pragma once
#include "hc_device.h"

namespace hc {

constexpr double EPS = 2.220446049250313e-16;
constexpr double SQRT_EPS = 1.4901161193847656e-08;
constexpr double RTOL = 1.0e-3, ATOL = 1.0e-3;          // richards_pde.py:496
constexpr double NEWTON_TOL = 0.03;                      // max(10*EPS/rtol, min(0.03, sqrt(rtol)))
constexpr int NEWTON_MAXITER = 4;
constexpr int MAX_ORDER = 5;
constexpr double NUM_JAC_DIFF_REJECT = 2.0097183471152322e-14;
constexpr double NUM_JAC_DIFF_SMALL = 1.8189894035458565e-12;
constexpr double NUM_JAC_DIFF_BIG = 0.0001220703125;
constexpr double NUM_JAC_MIN_FACTOR = 2.220446049250313e-13;

// Region accounting of the development builds (tools/isa_account.py, tools/prof_phases.py):
//   -DHC_PROFILE: cycles spent in and entries into each region, summed over waves (clock64 stamps; costs ~3 %)
//   -DHC_MARKS  : the same region ids as assembler comments ("; HCMARK n") -- zero instructions -- so that the static
//                 instruction mix of each region can be read off the .s file of the production kernel
#ifdef HC_PROFILE
#define HC_STAMP(slot)                                                   \
    {                                                                    \
        const unsigned long long now_ = clock64();                       \
        if (lane == 0) {                                                 \
            prof_lds[prof_slot] += (unsigned)(now_ - prof_t);            \
            prof_lds[32 + (slot)] += 1;                                  \
        }                                                                \
        prof_t = now_;                                                   \
        prof_slot = (slot);                                              \
    }
#define HC_SUB(slot)                                                                              \
    {                                                                                             \
        if (lane == 0) reinterpret_cast<unsigned long long *>(ru + 160)[32 + (slot)] += 1;        \
    }
#define HC_SUB_END()
#elif defined(HC_MARKS)
#define HC_STAMP(slot) asm volatile("; HCMARK %0" ::"n"(slot))
#define HC_SUB(slot) asm volatile("; HCMARK %0" ::"n"(slot))
#define HC_SUB_END() asm volatile("; HCMARK -2")
#else
#define HC_STAMP(slot)
#define HC_SUB(slot)
#define HC_SUB_END()
#endif

// Per-wave vectors (each 64*CPL doubles): the accepted state, the base f of a Jacobian, the FD step factors, the BDF
// difference rows D[0..7] and the row's noise vector.
// V_Y0, the state a row starts from, is only needed again when an attempt fails: deep columns (CPL > 5) keep it in the
// global region instead of registers, shallower ones never use it.  (Parking the Jacobian rows and column steps there
// as well was measured at D = 401 / 461 / 581: 7 % slower -- the compiler's own scratch placement does better.)
enum { V_Y = 0, V_FP, V_FAC, V_D0, V_NZ = V_D0 + MAX_ORDER + 3, NVEC = V_NZ + 1, V_Y0 = NVEC,
       // TWO layout only: the Jacobian rows and FD steps, the factorisation
       V_JL, V_JD, V_JU, V_HJ, NVEC_TWO };
constexpr int TWO_GROUP_VECTORS = 16;    // f of each FD-Jacobian group evaluation (n_groups <= 16), behind the lane scalars
constexpr int TWO_LANE_SCALARS = 14;     // per-lane scalars of the factorisation (wx, 1/B, al[6], ga[6])
#ifdef HC_PROFILE
constexpr int WAVE_SCRATCH = 256;   // + per wave: 32 cycle sums, 32 entry counts, 32 sub-region entry counts
#else
constexpr int WAVE_SCRATCH = 160;
#endif
// Waves per workgroup = per CU (LDS admits one workgroup): one wave per SIMD, except the shallow columns.  At two or
// three cells per lane the step kernel wants ~350 / ~400 registers; held to 256 (two waves per SIMD) it spills ~100 /
// ~150 of them to scratch and still gains +48 % / +12 % (D <= 128 / <= 192, r03 measurement in DESIGN.md §5) because a
// second wave fills the dependency stalls of the first.  From four cells per lane on the spill traffic outweighs that
// (D = 256: 0.63x, D = 300: 0.59x), and three or four waves per SIMD lose at every depth (D = 128: 0.64x / 0.48x).
// -DHC_WAVES_PER_BLOCK=n forces one count everywhere (A/B builds).
// Round 4: two waves per SIMD at FOUR to SIX cells per lane with hand-placed state ("TWO" layout, -DHC_TWO_MASK=<bit per
// cells-per-lane count>).  Nothing but the Newton-hot vectors (iterate, its noise, psi, d, 1/scale) stays in registers
// across an RHS evaluation; the factorisation, the Jacobian rows, the FD steps and the predicted state live in the wave's
// LDS / global vectors and are loaded by the phase that uses them -- see WaveVecs and rank_two below.
// Measured (profiles/r04_two_layout_ab.txt, state digests identical under -ffp-contract=on): default exponents 4 cells
// per lane 1.25x, 5 cells 1.12x; generic exponents 4 cells 1.21x, 5 cells 0.92x -- 1.08x once the column parameters are
// re-read per evaluation (P_RELOAD below); 6 / 7 / 8 cells 0.84 / 0.77 / 0.49x (LDS holds 5 / 4 / 3 vectors a wave) --
// after the traffic cuts 6 cells 0.91x with the cell model in one batch of 5 + 1 and 1.065x in batches of 4 + 2
// (-DHC_MODEL_BATCH=4 for that unit, __graft_entry__.UNIT_FLAGS), 7 cells 0.91x at best, generic 6 cells 0.97x.
// Round 5, late: with the spills gone (machine LICM off + sink-to-avoid-spills, __graft_entry__.UNIT_FLAGS) SEVEN cells per lane pay
// too -- D = 401 (the reference's default well) 199.5 k against 182.5 k column-days/s on the one-wave kernel built the same way
// (174 k before), D = 448 the same; EIGHT cells 0.74 - 0.81x (LDS holds 3 of 48 factorisation slots); generic exponents at 7
// cells 1.03x at best: left on the one-wave kernel (gpurun_out/r5x, r5y).
#ifndef HC_TWO_MASK
#define HC_TWO_MASK ((1 << 4) | (1 << 5) | (1 << 6) | (1 << 7))      // default exponents: bit per cells-per-lane count
#endif
#ifndef HC_TWO_MASK_GENERIC
#define HC_TWO_MASK_GENERIC ((1 << 4) | (1 << 5) | (1 << 6))   // generic exponents (6 cells: 0.97x in round 4, 1.034x with round 5's placement)
#endif
// "TWO layout" = the hand placement AND two waves per SIMD: the placement alone, at one wave per SIMD, loses at every depth
// (0.74 - 0.93x: nothing hides the memory round trips; LAB_NOTES.md "Round 4")
// Round 5: the split column (halves == 2, 5 cells per lane and half) takes the TWO layout too -- its halves are exactly the
// shape where the layout pays -- i.e. FOUR member pairs per CU instead of two (-DHC_TWO_PAIR=0: the round-4 kernel).
#ifndef HC_TWO_PAIR
#define HC_TWO_PAIR 1
#endif
#ifndef HC_TWO_J_FRESH         // the first factorisation after a Jacobian takes the rows from registers
#define HC_TWO_J_FRESH 0
#endif
#ifndef HC_TWO_J_ALIAS         // Jacobian-phase vectors in the (then dead) factorisation's LDS slots: WaveVecs::ldJ
#define HC_TWO_J_ALIAS 1
#endif
__host__ __device__ constexpr bool two_of(int cpl, int halves = 1, bool special = true)
{
    if (halves == 2) return HC_TWO_PAIR && cpl == 5;
    return halves == 1 && cpl >= 4 && (((special ? HC_TWO_MASK : HC_TWO_MASK_GENERIC) >> cpl) & 1);
}
#ifdef HC_WAVES_PER_BLOCK
__host__ __device__ constexpr int wpb_of(int, int = 1, bool = true) { return HC_WAVES_PER_BLOCK; }
constexpr int MAX_WAVES_PER_BLOCK = HC_WAVES_PER_BLOCK;
#else
__host__ __device__ constexpr int wpb_of(int cpl, int halves = 1, bool special = true)
{
    return ((halves == 1 && cpl <= 3) || two_of(cpl, halves, special)) ? 8 : 4;
}
constexpr int MAX_WAVES_PER_BLOCK = 8;
#endif
constexpr int LDS_BYTES = 160 * 1024;
// Where they live.  At CPL = 4 and 5 (D <= 320) all twelve fit in LDS next to the shared tables with four waves per
// workgroup (at CPL = 3 with eight waves, eleven do).  Deeper columns keep four waves per CU -- every SIMD busy -- by moving the vectors that are touched least
// to a per-wave region in global memory (L2 / Infinity-Cache resident: 1 024 waves x <= 31 KB): first D[7], the FD
// factors (read and written by Jacobian evaluations only, ~1.2 per row), D[6], the Jacobian's base f and the accepted
// state (both written once per step or accepted step -- fire-and-forget stores -- and read once per Jacobian / row),
// then D[5], D[4].  Every lane only ever touches its own slots of these, so program order is all the ordering needed.
// The noise vector is read across lanes (cell j uses n_rnd[j-1]) and at every attempt: it is always in LDS.  (Round 1
// dropped it from LDS for deep columns and rebuilt it per attempt; with the overflow region in place keeping it is
// faster at every depth -- D = 401 +3 %, 541 +6 %, 581 +8 % -- and removes a second code path.)
// `halves` = 2: a column split over two waves (hc_device.h, Comm<2>) -- each wave holds 64 cpl nodes, the shared tables
// cover both halves, and the workgroup's two mailboxes sit behind the waves' vectors.
// bytes of LDS one wave has for its vectors (next to the workgroup's tables, the pair mailboxes and the wave's scratch)
__host__ __device__ constexpr int lds_wave_bytes(int cpl, int halves = 1, bool special = true)
{
    const int slots = 64 * cpl;
    const int tables = (ntab_lds(slots * halves, special) * slots * 8 + 4 * slots) * halves;
    const int wpb = wpb_of(cpl, halves, special);
    const int boxes = halves == 2 ? (wpb / 2) * (int)sizeof(PairBox) : 0;
    return (LDS_BYTES - tables - boxes) / wpb - WAVE_SCRATCH * 8;
}
// TWO layout (round 5): LDS holds the three difference rows an order-1 step touches, D[0..2], and behind them as many
// LANE-SLOTS (64 doubles) of the factorisation as fit -- the factorisation is 5 cpl + 8 lane-slots (the entries a solve
// reads: wf of cells 1.., wb of cells ..cpl-3, l / u / 1/b of cells ..cpl-2, and the 14 per-lane scalars of the cyclic
// reduction), placed slot by slot instead of vector by vector so that no LDS is left over: 16 of 33 at 5 cells per lane
// (round 4: two vectors = 7 useful slots), 20 of 28 at 4, 12 of 38 at 6.
constexpr int TWO_LDS_ROWS = 3;
// (split column: + the block's answer to a unit load at the cut, sp[cpl], and one slot for the five wave-uniform
//  coupling scalars, one per lane 0..4)
__host__ __device__ constexpr int two_f_slots(int cpl, int halves = 1) { return 5 * cpl + 8 + (halves == 2 ? cpl + 1 : 0); }
__host__ __device__ constexpr int two_f_lds(int cpl, int halves = 1, bool special = true)
{
    const int n = (lds_wave_bytes(cpl, halves, special) - TWO_LDS_ROWS * 64 * cpl * 8) / (64 * 8);
    return n < 0 ? 0 : (n < two_f_slots(cpl, halves) ? n : two_f_slots(cpl, halves));
}
__host__ __device__ constexpr int lds_vectors(int cpl, int halves = 1, bool special = true)   // how many of the twelve fit
{
    if (two_of(cpl, halves, special)) return TWO_LDS_ROWS;
    const int n = lds_wave_bytes(cpl, halves, special) / (64 * cpl * 8);
    return n < (int)NVEC ? n : (int)NVEC;
}
// doubles of LDS per wave ahead of its scratch area
__host__ __device__ constexpr int lds_wave_doubles(int cpl, int halves = 1, bool special = true);
// Room to spare (two cells per lane, eight waves per workgroup): the Jacobian's three rows and the FD steps -- per-lane
// arrays that are live from the first group evaluation of a Jacobian to the factorisation, across every RHS evaluation in
// between -- move from registers to four more per-wave vectors behind the wave's scratch area.  At two waves per SIMD
// (256 registers) each of them is otherwise a scratch round trip.
__host__ __device__ constexpr int lds_extra(int cpl, int halves = 1, bool special = true)
{
    // (CPL = 4 with four waves would have the room too: measured 1.5 % slower -- at one wave per SIMD a parked register
    //  is an AGPR move, cheaper than the LDS round trip)
    if (halves != 1 || wpb_of(cpl, halves, special) != 8) return 0;
    const int slots = 64 * cpl;
    const int tables = ntab_lds(slots, special) * slots * 8 + 4 * slots;
    const int n = ((LDS_BYTES - tables) / wpb_of(cpl, halves, special) - WAVE_SCRATCH * 8) / (slots * 8);
    return n >= NVEC + 4 ? 4 : 0;
}
__host__ __device__ constexpr int lds_listed(int cpl, int halves = 1, bool special = true)    // ... of the eleven vectors other than the noise
{
    const int n = lds_vectors(cpl, halves, special) - 1;
    return n < NVEC - 1 ? n : NVEC - 1;
}
__host__ __device__ constexpr int lds_wave_doubles(int cpl, int halves, bool special)
{
    return two_of(cpl, halves, special) ? TWO_LDS_ROWS * 64 * cpl + two_f_lds(cpl, halves, special) * 64
                                        : (lds_vectors(cpl, halves, special) + lds_extra(cpl, halves, special)) * 64 * cpl;
}
constexpr int TWO_GLOBAL_VECTORS = 14;   // TWO layout: JL JD JU HJ D3 Y FP D4 FAC D5 D6 D7 NZ Y0 (rank_two - TWO_LDS_ROWS)
__host__ __device__ constexpr int spill_vectors(int cpl, int halves = 1, bool special = true)   // incl. V_Y0
{
    return two_of(cpl, halves, special) ? TWO_GLOBAL_VECTORS : NVEC - lds_listed(cpl, halves, special);
}
// doubles of the global region one wave owns (TWO: the vectors, the factorisation's lane-slots LDS has no room for, the
// group evaluations of the Jacobian)
__host__ __device__ constexpr int spill_doubles(int cpl, int halves = 1, bool special = true)
{
    return spill_vectors(cpl, halves, special) * 64 * cpl +
           (two_of(cpl, halves, special) ? (two_f_slots(cpl, halves) - two_f_lds(cpl, halves, special)) * 64 + TWO_GROUP_VECTORS * 64 * cpl : 0);
}
// rank of a vector in the keep-in-LDS order D0..D5, Y, FP, D6, FAC, D7 (, Y0: never in LDS)
__host__ __device__ constexpr int vec_rank(int v)
{
    return v == V_Y0 ? 11 : v == V_Y ? 6 : v == V_FP ? 7 : v == V_FAC ? 9 : v == V_D0 + 6 ? 8 : v == V_D0 + 7 ? 10 : v - V_D0;
}

// TWO layout: keep-in-LDS order.  The noise vector (read across lanes), the difference rows an order-1 step touches
// (a row's integration restarts at order 1 and rarely leaves it: 5.9 of 6.1 steps per row at D = 300), then what a
// Newton iteration reads -- the factorisation in the order the solve consumes it, so that the parts that fall to the
// global region are the ones needed last -- then the Jacobian rows and the FD steps; the vectors the deep-column kernels
// already keep in the global region come last.  (The predicted state / Jacobian base point has no vector: HC_YP_LOAD.)
// Round 5: the noise vector no longer takes an LDS slot.  It is written when a row's noise is generated (or damped after a
// failed attempt) and read ONCE per attempt -- across lanes, which a wave may do on its own global region after waiting for
// its stores (one s_waitcnt per attempt; the CU's vector L1 is write-through and shared by the workgroup, so the wave's
// own earlier stores are what its later loads see).  Its LDS goes to the factorisation, which every Newton iteration reads.
__host__ __device__ constexpr int rank_two(int v)
{
    return v == V_D0 ? 0 : v == V_D0 + 1 ? 1 : v == V_D0 + 2 ? 2 : v == V_JL ? 3 : v == V_JD ? 4 : v == V_JU ? 5 : v == V_HJ ? 6 :
           v == V_D0 + 3 ? 7 : v == V_Y ? 8 : v == V_FP ? 9 : v == V_D0 + 4 ? 10 : v == V_FAC ? 11 :
           v == V_D0 + 5 ? 12 : v == V_D0 + 6 ? 13 : v == V_D0 + 7 ? 14 : v == V_NZ ? 15 : 16 /* V_Y0 */;
}
constexpr bool TWO_NZ_GLOBAL = true;

// A wave's vectors: `lds` holds the first lds_listed(CPL) of the order above (then the noise vector, if it is in LDS),
// `spill` the rest.  VEC is a compile-time id; the D-row accessors take the row as an unrolled loop index.
template <int CPL, int HALVES = 1, bool SP = true>
struct WaveVecs {
    static constexpr int SLOTS = WAVE * CPL;
    static constexpr bool TWO = two_of(CPL, HALVES, SP);
    static constexpr int N_LDS = TWO ? TWO_LDS_ROWS : lds_listed(CPL, HALVES, SP);
    static constexpr int NF = two_f_slots(CPL, HALVES), F_LDS = TWO ? two_f_lds(CPL, HALVES, SP) : 0;
    // the global region through a buffer resource (see gld / gst): the TWO layout and, since round 4, the deep columns
#ifndef HC_RSRC_MIN_CPL
#define HC_RSRC_MIN_CPL 8      // (measured: 6 cells per lane -5 %, 7 -1.4 %, 8 +5 %, 9 +11 %, 10 +42 %; digests unchanged)
#endif
    static constexpr bool RSRC = TWO || CPL >= HC_RSRC_MIN_CPL;
    double *lds;
    __attribute__((address_space(1))) double *spill;
    template <int VEC>
    __device__ __forceinline__ double ld(int slot) const
    {
        if constexpr (TWO) {
            if constexpr (rank_two(VEC) < N_LDS) return lds[rank_two(VEC) * SLOTS + slot];
            else return gld((rank_two(VEC) - N_LDS) * SLOTS * 8, slot * 8);
        } else {
            static_assert(VEC <= V_Y0, "a vector of the TWO layout");
            if constexpr (VEC == V_NZ) return lds[N_LDS * SLOTS + slot];
            else if constexpr (vec_rank(VEC) < N_LDS) return lds[vec_rank(VEC) * SLOTS + slot];
            else if constexpr (RSRC) return gld((vec_rank(VEC) - N_LDS) * SLOTS * 8, slot * 8);
            else return spill[(vec_rank(VEC) - N_LDS) * SLOTS + slot];
        }
    }
    template <int VEC>
    __device__ __forceinline__ void st(int slot, double v) const
    {
        if constexpr (TWO) {
            if constexpr (rank_two(VEC) < N_LDS) lds[rank_two(VEC) * SLOTS + slot] = v;
            else gst((rank_two(VEC) - N_LDS) * SLOTS * 8, slot * 8, v);
        } else {
            static_assert(VEC <= V_Y0, "a vector of the TWO layout");
            if constexpr (VEC == V_NZ) lds[N_LDS * SLOTS + slot] = v;
            else if constexpr (vec_rank(VEC) < N_LDS) lds[vec_rank(VEC) * SLOTS + slot] = v;
            else if constexpr (RSRC) gst((vec_rank(VEC) - N_LDS) * SLOTS * 8, slot * 8, v);
            else spill[(vec_rank(VEC) - N_LDS) * SLOTS + slot] = v;
        }
    }
    // TWO layout: lane-slot k of the factorisation (k is a constant after unrolling): the first F_LDS of them behind the
    // difference rows in LDS, the rest behind the vectors of the global region
    static constexpr int F_OFF = TWO_GLOBAL_VECTORS * SLOTS * 8;
    __device__ __forceinline__ double ldF(int k, int lane) const
    {
        if (k < F_LDS) return lds[TWO_LDS_ROWS * SLOTS + k * WAVE + lane];
        return gld(F_OFF, ((k - F_LDS) * WAVE + lane) * 8);
    }
    __device__ __forceinline__ void stF(int k, int lane, double v) const
    {
        if (k < F_LDS) lds[TWO_LDS_ROWS * SLOTS + k * WAVE + lane] = v;
        else gst(F_OFF, ((k - F_LDS) * WAVE + lane) * 8, v);
    }
    // Round 5: while a Jacobian is being evaluated the factorisation is DEAD (every Jacobian ends with have_lu = 0), so its
    // LDS slots hold the FD steps and as many Jacobian rows as fit (5 cells per lane: the steps + two rows in 15 of the 16
    // slots; 4 cells: all four vectors) for the 6 - 7 phases that scatter into, re-read and finalise them.  The finished
    // rows go to the global region once (later factorisations of the same Jacobian read them there); the FD steps die with
    // the Jacobian.  ldJ / stJ: the Jacobian-phase view of a vector; ld / st stay the persistent one.
    static constexpr int N_ALIAS = (TWO && HC_TWO_J_ALIAS) ? (F_LDS / CPL < 4 ? F_LDS / CPL : 4) : 0;
    static constexpr int alias_of(int v) { return v == V_HJ ? 0 : v == V_JL ? 1 : v == V_JD ? 2 : v == V_JU ? 3 : 99; }
    template <int VEC>
    __device__ __forceinline__ double ldJ(int slot) const
    {
        if constexpr (alias_of(VEC) < N_ALIAS) return lds[(TWO_LDS_ROWS + alias_of(VEC)) * SLOTS + slot];
        else return ld<VEC>(slot);
    }
    template <int VEC>
    __device__ __forceinline__ void stJ(int slot, double v) const
    {
        if constexpr (alias_of(VEC) < N_ALIAS) lds[(TWO_LDS_ROWS + alias_of(VEC)) * SLOTS + slot] = v;
        else st<VEC>(slot, v);
    }
    // lane-slot of each entry: wf[1..], wb[..CPL-3], l / u / 1/b [..CPL-2], then the 14 scalars
    static constexpr int K_WF = -1, K_WB = CPL - 1, K_L = 2 * CPL - 3, K_U = 3 * CPL - 4, K_IB = 4 * CPL - 5, K_S = 5 * CPL - 6;
    static constexpr int K_SP = 5 * CPL + 8, K_UNI = 6 * CPL + 8;      // split column only
    // TWO layout: f of FD-Jacobian group evaluation g (stored whole, no read-modify-write of the Jacobian rows per group);
    // the finalising phase gathers row entries by each lane's own group ids
    static constexpr int FG_OFF = (TWO_GLOBAL_VECTORS * SLOTS + (NF - F_LDS) * WAVE) * 8;
    __device__ __forceinline__ void stG(int g_uniform, int slot, double v) const { gst(FG_OFF + g_uniform * (SLOTS * 8), slot * 8, v); }
    __device__ __forceinline__ double ldG(int g_lane, int slot) const { return gld(FG_OFF, g_lane * (SLOTS * 8) + slot * 8); }
    // TWO layout: the wave's global region through a buffer resource -- scalar base + scalar vector offset + per-lane
    // byte offset + immediate, ONE address VGPR for every access (as plain global pointers hipcc keeps a 64-bit VGPR
    // address per vector and cell, ~170 registers, and spills them); out-of-range accesses are dropped by the hardware
    typedef unsigned v2u_t __attribute__((ext_vector_type(2)));
    __amdgpu_buffer_rsrc_t rsrc;
    __device__ __forceinline__ void bind(double *region_base)
    {
        rsrc = __builtin_amdgcn_make_buffer_rsrc(region_base, 0, spill_doubles(CPL, HALVES, SP) * 8, 0x00020000);
    }
    __device__ __forceinline__ double gld(int vec_bytes, int lane_bytes) const
    {
        const v2u_t r = __builtin_amdgcn_raw_buffer_load_b64(rsrc, lane_bytes, vec_bytes, 0);
        return __hiloint2double((int)r.y, (int)r.x);
    }
    // Loads that must see what OTHER lanes (or, on the split column, the partner wave) stored to the region: the noise
    // vector's cross-lane read.  sc0 | sc1: served from the L2, never from a line the vector L1 may still hold.
    __device__ __forceinline__ double gld_fresh(int vec_bytes, int lane_bytes) const
    {
        const v2u_t r = __builtin_amdgcn_raw_buffer_load_b64(rsrc, lane_bytes, vec_bytes, 17);
        return __hiloint2double((int)r.y, (int)r.x);
    }
    __amdgpu_buffer_rsrc_t rsrc_partner;       // split column: the other half's region
    __device__ __forceinline__ void bind_partner(double *region_base)
    {
        rsrc_partner = __builtin_amdgcn_make_buffer_rsrc(region_base, 0, spill_doubles(CPL, HALVES, SP) * 8, 0x00020000);
    }
    __device__ __forceinline__ double gld_partner_fresh(int vec_bytes, int lane_bytes) const
    {
        const v2u_t r = __builtin_amdgcn_raw_buffer_load_b64(rsrc_partner, lane_bytes, vec_bytes, 17);
        return __hiloint2double((int)r.y, (int)r.x);
    }
    static constexpr int NZ_OFF = TWO ? (rank_two(V_NZ) - TWO_LDS_ROWS) * SLOTS * 8 : 0;
    __device__ __forceinline__ void gst(int vec_bytes, int lane_bytes, double v) const
    {
        v2u_t r;
        r.x = (unsigned)__double2loint(v);
        r.y = (unsigned)__double2hiint(v);
        __builtin_amdgcn_raw_buffer_store_b64(r, rsrc, lane_bytes, vec_bytes, 0);
    }
    // D[k]: k is a constant after unrolling, the switch folds away
    __device__ __forceinline__ double ldD(int k, int slot) const
    {
        switch (k) {
            case 0: return ld<V_D0 + 0>(slot);
            case 1: return ld<V_D0 + 1>(slot);
            case 2: return ld<V_D0 + 2>(slot);
            case 3: return ld<V_D0 + 3>(slot);
            case 4: return ld<V_D0 + 4>(slot);
            case 5: return ld<V_D0 + 5>(slot);
            case 6: return ld<V_D0 + 6>(slot);
            default: return ld<V_D0 + 7>(slot);
        }
    }
    __device__ __forceinline__ void stD(int k, int slot, double v) const
    {
        switch (k) {
            case 0: st<V_D0 + 0>(slot, v); break;
            case 1: st<V_D0 + 1>(slot, v); break;
            case 2: st<V_D0 + 2>(slot, v); break;
            case 3: st<V_D0 + 3>(slot, v); break;
            case 4: st<V_D0 + 4>(slot, v); break;
            case 5: st<V_D0 + 5>(slot, v); break;
            case 6: st<V_D0 + 6>(slot, v); break;
            default: st<V_D0 + 7>(slot, v); break;
        }
    }
};
// Iteration budget of one BDF attempt (loop trips of the phase machine; a hard row needs a few hundred).  It is part of
// the semantics at scale: on a state that slides along a discontinuity of the RHS (psi_sat / lateral-flow switch) Newton
// only converges for h ~ 1e-11, the step controller cycles "halve, halve, halve, accept twice, x10" forever and time
// advances ~1e-11 per cycle -- the reference's integrator has no exit there short of h < min_step, which never comes.
// About one attempt in 10^7 does this (86 in 4.7e8 row solves of the 262 144-member run, tools/guard_hunt.py).  Such an
// attempt is abandoned and handled by the reference's own failure rule: noise x0.8, retry (richards_pde.py:509-533).
constexpr int MAX_PHASE_ITERATIONS = 20000;

// Kernel arguments.  Only what the hot loop needs stays in the kernarg segment; the column parameters
// and the row-loop I/O pointers live in device memory and are (re)loaded through constant-address-space
// scalar loads exactly where they are used (load_const below).  Keeping all of them in SGPRs for the
// whole kernel made hipcc spill ~300 SGPRs to VGPR lanes, i.e. a v_readlane per use.
struct IoArgs {
    double *psi;              // [N][D] in/out
    double *base_noise;       // [N][D] or null (Philox)
    const double *fresh;      // [n_fresh][N][D] or null
    double *nscale;           // [N] Philox-mode damping factor of the base vector
    const double *precip, *atm;
    const unsigned char *daylight, *refresh;
    const int *wtd_obs, *draw_idx;
    long long member_offset;
    long long row_begin;
    unsigned long long seed;
    unsigned short *wtd_u16;  // [n_rows][N]
    int *stats;               // [n_rows][N][6] or null
    double *psi_rows;         // [n_rows][N][D] or null
    double *diag;             // [n_rows][N][2] transpiration, lateral flow of the row's last RHS evaluation, or null
    unsigned long long *counters;   // [0] FD-Jacobian retry passes, [1] failed attempts, [2] loop-guard trips
    unsigned long long *queue;      // member ticket of the persistent grid, zeroed before every launch
    int *spin_iters;          // [N] spin-up with the stop rule: solves used (negative: cap reached), or null
    // parameter points (n_points > 1 only):
    const long long *point_base;     // [n_points] global id of each point's first member: the Philox stream key of member
                                     // j of point k is point_base[k] + j, whichever handle / rank / order runs the point
    const int *point_order;          // [n_points] the chunk ticket walks the points in this order (costliest first)
    unsigned long long *point_cost;  // [n_points] RHS evaluations spent on each point's members by this launch
    double *trace;            // diagnostic builds: [1 + 6 * HC_TRACE_N] phase trace of member 0, or null
};

constexpr unsigned PHILOX_DRAW_SPINUP = 0xFFFFFFFFu;
constexpr int HC_TRACE_N = 20000;

struct StepArgs {
    const ColumnDev *P;       // device memory, [n_points]
    const IoArgs *io;         // device memory
    double *wave_spill;       // [grid * wpb_of(CPL) waves][spill_vectors(CPL)][SLOTS]: per-wave vectors that do not fit in LDS, or null
    const double *tab;        // [n_points][NTAB][SLOTS]
    const int *gtab;          // [NGTAB][SLOTS]
    long long n_members;
    int n_rows, spinup, D, n_groups, host_noise;
    double psi_sat;
    double jac_reject;        // NUM_JAC_DIFF_REJECT = EPS**0.875 (debug override: HYDROCOL_DEBUG_JAC_REJECT)
    // spin-up with the per-member stop rule of simulation.py:468 evaluated in the kernel (hc_spinup)
    int spin_stop;
    double spin_zwtd, spin_z0, spin_dz;
    int max_phase_iterations; // MAX_PHASE_ITERATIONS (test hook: HYDROCOL_DEBUG_MAX_ITER lowers it to force abandoned attempts)
    // select_initial_step as the reference's pinned scipy==1.5.2 has it: neither h0 nor the returned step is clamped to the
    // interval (scipy >= 1.9 clamps both; 1.15.3 made the pinning vectors and is the default).  hc_set_scipy_152.
    int scipy_152;
    // parameter points (BASELINE config 5): P[n_points], tab[n_points][NTAB][SLOTS]; members are point-major, point k owns
    // members [k * members_per_point, (k + 1) * members_per_point).  With more than one point a workgroup works through
    // chunks of `chunk_members` members of ONE point (its tables sit in LDS); chunk c = (point c / chunks_per_point,
    // members (c % chunks_per_point) * chunk_members ... of that point).
    int n_points, chunk_members, chunks_per_point, n_chunks;
    long long members_per_point;
};

// Uniform struct load from device memory through the constant address space (s_load_dwordxN).  The empty
// asm makes the pointer opaque at this program point, so the loads cannot be hoisted out of the enclosing
// loop and pinned in SGPRs for the kernel's lifetime.
template <class T>
__device__ __forceinline__ T load_const(const T *p)
{
    asm volatile("" : "+s"(p));
    T out;
    __builtin_memcpy(&out, (const __attribute__((address_space(4))) T *)p, sizeof(T));
    return out;
}

// gamma = [0, cumsum(1/k)], alpha = (1 - kappa) * gamma, error_const = kappa * gamma + 1/(k+1)
// with kappa = [0, -0.1850, -1/9, -0.0823, -0.0415, 0]  (bdf.py BDF.__init__), as NumPy evaluates them
static __device__ const double GAMMA_TAB[6] = {0.0, 1.0, 1.5, 1.8333333333333333, 2.083333333333333, 2.283333333333333};
static __device__ const double ALPHA_TAB[6] = {0.0, 1.185, 1.6666666666666667, 1.9842166666666667, 2.1697916666666663, 2.283333333333333};
static __device__ const double ERRC_TAB[6] = {1.0, 0.315, 0.16666666666666666, 0.09911666666666669, 0.11354166666666668, 0.16666666666666666};
__device__ __forceinline__ double gamma_k(int k) { return GAMMA_TAB[k]; }
__device__ __forceinline__ double alpha_k(int k) { return ALPHA_TAB[k]; }
__device__ __forceinline__ double error_const_k(int k) { return ERRC_TAB[k]; }

// RMS norm of x[c]/scale[c] over the D valid nodes; is[c] = 1/scale[c]
template <int CPL, class CommT>
__device__ __forceinline__ double rms_ratio(const double (&x)[CPL], const double (&is)[CPL], int lane, int D,
                                            double inv_sqrt_d, CommT &comm)
{
    // no node mask: every vector this is applied to is exactly 0 in the padding slots beyond the grid (state,
    // differences and dy/dt are kept 0 there), and 1/scale is finite there
    (void)lane;
    (void)D;
    double acc = 0.0;
#pragma unroll
    for (int c = 0; c < CPL; c++) {
        const double r = x[c] * is[c];
        acc = fma(r, r, acc);
    }
    return sqrt_pos(comm.sum(acc)) * inv_sqrt_d;
}

// bdf.py change_D: D[:order+1] = (R(order,factor) @ R(order,1)).T @ D[:order+1].
// R[i][j] = prod_{q=1..i} (q-1-factor*j)/q (R[0][j] = 1, R[i>0][0] = 0); lanes 0..35 build R and U = R(.,1)
// in LDS, lanes 0..35 form RU, then every lane applies RU^T to its cells.
template <int CPL, int ORDER, class WV>
__device__ __forceinline__ void apply_RU(const WV &W, const double *ru, int lane)
{
    constexpr int SLOTS = WAVE * CPL;
    double m[ORDER + 1][ORDER + 1];
#pragma unroll
    for (int a = 0; a <= ORDER; a++)
#pragma unroll
        for (int b = 0; b <= ORDER; b++) m[a][b] = ru[72 + a * 6 + b];
#pragma unroll
    for (int c = 0; c < CPL; c++) {
        const int slot = c * WAVE + lane;
        double col[ORDER + 1];
#pragma unroll
        for (int k = 0; k <= ORDER; k++) col[k] = W.ldD(k, slot);
#pragma unroll
        for (int a = 0; a <= ORDER; a++) {
            double s = 0.0;
#pragma unroll
            for (int k = 0; k <= ORDER; k++) s += m[k][a] * col[k];
            W.stD(a, slot, s);
        }
    }
}

// U of bdf.py change_D does not depend on the step factor: computed once per wave into ru[36..71]
// (U[i][j] = prod_{q=1..i} (q - 1 - j) / q, U[0][:] = 1, U[i>=1][0] = 0)
__device__ __forceinline__ void change_D_init(double *ru, int lane)
{
    const int i = lane / 6, j = lane % 6;
    double u = 1.0;
    if (lane < 36) {
        if (i >= 1) {
            if (j == 0) {
                u = 0.0;
            } else {
                for (int q = 1; q <= i; q++) u *= ((double)(q - 1) - (double)j) / (double)q;
            }
        }
        ru[36 + lane] = u;
    }
    __builtin_amdgcn_wave_barrier();
}

template <int CPL, class WV>
__device__ __forceinline__ void change_D(const WV &W, double *ru, int order, double factor, int lane)
{
    // R[i][j] = prod_{q=1..i} (q - 1 - factor j) / q on lanes 0..35 (cumprod of bdf.py compute_R), RU = R U on the
    // same lanes with all twelve operands read before the first multiply, then D[:order+1] = RU^T D[:order+1]
    const int i = lane / 6, j = lane % 6;
    if (lane < 36) {
        double r = 1.0;
        const double fj = factor * (double)j;
#pragma unroll
        for (int q = 1; q <= 5; q++) {
            const double t = r * fast_div((double)(q - 1) - fj, (double)q);
            r = q <= i ? t : r;
        }
        r = (i >= 1 && j == 0) ? 0.0 : r;
        ru[lane] = r;
    }
    __builtin_amdgcn_wave_barrier();
    if (lane < 36) {
        double ra[6], ub[6];
#pragma unroll
        for (int k = 0; k < 6; k++) {
            ra[k] = ru[i * 6 + k];
            ub[k] = ru[36 + k * 6 + j];
        }
        double acc = 0.0;
#pragma unroll
        for (int k = 0; k < 6; k++) {
            const double t = fma(ra[k], ub[k], acc);
            acc = k <= order ? t : acc;
        }
        ru[72 + lane] = acc;
    }
    __builtin_amdgcn_wave_barrier();
    switch (order) {
        case 1: HC_SUB(33); apply_RU<CPL, 1>(W, ru, lane); break;
        case 2: HC_SUB(34); apply_RU<CPL, 2>(W, ru, lane); break;
        case 3: HC_SUB(35); apply_RU<CPL, 3>(W, ru, lane); break;
        case 4: HC_SUB(36); apply_RU<CPL, 4>(W, ru, lane); break;
        default: HC_SUB(37); apply_RU<CPL, 5>(W, ru, lane); break;
    }
    HC_SUB_END();
    __builtin_amdgcn_wave_barrier();
}

// bdf.py _step_impl after acceptance, for a compile-time order: D[order+2] = d - D[order+1]; D[order+1] = d;
// D[k] += D[k+1] for k = order..0.  All rows are read before any is written (independent LDS reads).
// TOP2_LATER (TWO layout, round 5): D[order+2] is not stored here.  Nothing reads it unless the ORDER RISES at this very
// step -- the next accepted step overwrites it, change_D and the predictor stop at row order(+1) -- and the caller, which
// holds it in d_ord2, stores it in that case only (0.2 of 6.1 steps per column-step at D = 300).
template <int CPL, int ORDER, bool TOP2_LATER, class WV>
__device__ __forceinline__ void accept_update(const WV &W, const double (&dd)[CPL], int lane, double (&d_ord)[CPL],
                                              double (&d_ord2)[CPL])
{
    constexpr int SLOTS = WAVE * CPL;
#pragma unroll
    for (int c = 0; c < CPL; c++) {
        const int slot = c * WAVE + lane;
        double r[ORDER + 2];
#pragma unroll
        for (int k = 0; k <= ORDER + 1; k++) r[k] = W.ldD(k, slot);
        const double top2 = dd[c] - r[ORDER + 1];
        if (!TOP2_LATER) W.stD(ORDER + 2, slot, top2);
        W.stD(ORDER + 1, slot, dd[c]);
        double acc = dd[c];
#pragma unroll
        for (int k = ORDER; k >= 0; k--) {
            acc += r[k];
            W.stD(k, slot, acc);
            if (k == ORDER) d_ord[c] = acc;
        }
        d_ord2[c] = top2;
    }
}
// predictor: y_predict = sum_k D[k], psi = sum_{k>=1} gamma_k D[k] / alpha_order
template <int CPL, int ORDER, class WV>
__device__ __forceinline__ void predict(const WV &W, int lane, double inv_alpha, double (&yp)[CPL],
                                        double (&psiv)[CPL])
{
    constexpr int SLOTS = WAVE * CPL;
#pragma unroll
    for (int c = 0; c < CPL; c++) {
        const int slot = c * WAVE + lane;
        double r[ORDER + 1];
#pragma unroll
        for (int k = 0; k <= ORDER; k++) r[k] = W.ldD(k, slot);
        double sy = r[0], p = 0.0;
#pragma unroll
        for (int k = 1; k <= ORDER; k++) {
            sy += r[k];
            p += r[k] * GAMMA_TAB[k];
        }
        yp[c] = sy;
        psiv[c] = p * inv_alpha;
    }
}

// the predictor's y alone (the base point of a Jacobian refresh, TWO layout): the additions of predict<> in its order
template <int CPL, int ORDER, class WV>
__device__ __forceinline__ void predict_y(const WV &W, int lane, double (&yp)[CPL])
{
#pragma unroll
    for (int c = 0; c < CPL; c++) {
        const int slot = c * WAVE + lane;
        double sy = W.ldD(0, slot);
#pragma unroll
        for (int k = 1; k <= ORDER; k++) sy += W.ldD(k, slot);
        yp[c] = sy;
    }
}

// ---- lane-partitioned tridiagonal factorisation of A = I - cc*J -------------------------
// H = 2 (split column): each wave factorises ITS block of the matrix with the coupling across the cut removed -- the
// upper half's last row loses its super-diagonal `couple` = c_last, the lower half's first row its sub-diagonal a_first --
// and keeps the block's answer to a unit load at the cut, sp = T^-1 e_edge.  With g = couple * sp[edge] of either half,
//   x_last(upper) = (z_last - g_u z_first) / (1 - g_u g_l),   x_first(lower) = z_first - g_l x_last,
// and every row follows from its own block: x = z - couple * x_edge(other half) * sp.  One exchange per solve.
template <int CPL, int H = 1>
struct TriLU {
    double wf[CPL], wb[CPL], l[CPL], u[CPL], ib[CPL];
    double wx, invB, al[6], ga[6];
    double sp[H == 2 ? CPL : 1];
    double couple, couple_other, g_up, g_lo, inv_den;
};

template <int CPL, int H>
__device__ __forceinline__ void lu_solve_block(const TriLU<CPL, H> &F, double (&x)[CPL], int lane);

template <int CPL, class CommT>
__device__ __forceinline__ void lu_factor(TriLU<CPL, CommT::H> &F, const double (&jl)[CPL], const double (&jd)[CPL],
                                          const double (&ju)[CPL], double cc, int lane, int D, CommT &comm)
{
    constexpr int H = CommT::H;
    const int hb = H == 2 ? comm.half * (WAVE * CPL) : 0;
    double a[CPL], b[CPL], cu[CPL];
#pragma unroll
    for (int c = 0; c < CPL; c++) {
        const bool v = hb + lane * CPL + c < D;
        a[c] = v ? -cc * jl[c] : 0.0;
        b[c] = v ? 1.0 - cc * jd[c] : 1.0;
        cu[c] = v ? -cc * ju[c] : 0.0;
    }
    if constexpr (H == 2) {
        if (comm.half == 0) {
            F.couple = readlane_d(cu[CPL - 1], WAVE - 1);
            cu[CPL - 1] = lane == WAVE - 1 ? 0.0 : cu[CPL - 1];
        } else {
            F.couple = readlane_d(a[0], 0);
            a[0] = lane == 0 ? 0.0 : a[0];
        }
    }
    // forward: remove the sub-diagonal inside the chunk; fill-in l[] couples to x_{s-1}
    F.l[0] = a[0];
    F.wf[0] = 0.0;
#pragma unroll
    for (int c = 1; c < CPL; c++) {
        const double w = fast_div(a[c], b[c - 1]);
        F.wf[c] = w;
        F.l[c] = -w * F.l[c - 1];
        b[c] -= w * cu[c - 1];
    }
    // backward: remove the super-diagonal of rows s..e-2; fill-in u[] couples to x_e
    F.u[CPL - 1] = cu[CPL - 1];   // row e: still coupled to the next chunk's first unknown
    F.wb[CPL - 1] = 0.0;
    if (CPL >= 2) {
        F.u[CPL - 2] = cu[CPL - 2];
        F.wb[CPL - 2] = 0.0;
    }
#pragma unroll
    for (int c = CPL - 3; c >= 0; c--) {
        const double w = fast_div(cu[c], b[c + 1]);
        F.wb[c] = w;
        F.l[c] -= w * F.l[c + 1];
        F.u[c] = -w * F.u[c + 1];
    }
    // last row of the chunk: eliminate x_{s'} with the next lane's first row (l,b,u)
    const double nl = shfl_down1(F.l[0], lane, 0.0);
    const double nb = shfl_down1(b[0], lane, 1.0);
    const double nu = shfl_down1(F.u[0], lane, 0.0);
    const double w = fast_div(F.u[CPL - 1], nb);
    F.wx = w;
    double L = F.l[CPL - 1], B = b[CPL - 1] - w * nl, U = -w * nu;
#pragma unroll
    for (int c = 0; c < CPL; c++) F.ib[c] = fast_div(1.0, b[c]);
    // parallel cyclic reduction on the 64 chunk-end unknowns
#pragma unroll
    for (int s = 0; s < 6; s++) {
        const int d = 1 << s;
        const int am = (lane - d) * 4, ap = (lane + d) * 4;
        // A lane without a neighbour at distance d reads another lane's (finite, non-zero pivot) values through
        // the wrapped permute address; its own coupling L (or U) is an exact zero by then -- lane 0 has no lower
        // coupling, lane 63 no upper one, and each stage multiplies the couplings of the next 2^s lanes by that
        // zero -- so al (ga) and everything it multiplies vanish without a mask.
        const double Bm = bpermute_d(am, B), Um = bpermute_d(am, U), Lm = bpermute_d(am, L);
        const double Bp = bpermute_d(ap, B), Lp = bpermute_d(ap, L), Up = bpermute_d(ap, U);
        const double al = fast_div(-L, Bm), ga = fast_div(-U, Bp);
        F.al[s] = al;
        F.ga[s] = ga;
        B = B + al * Um + ga * Lp;
        L = al * Lm;
        U = ga * Up;
    }
    F.invB = fast_div(1.0, B);
    if constexpr (H == 2) {
        double e[CPL];
#pragma unroll
        for (int c = 0; c < CPL; c++) e[c] = 0.0;
        if (comm.half == 0) e[CPL - 1] = lane == WAVE - 1 ? 1.0 : 0.0;
        else e[0] = lane == 0 ? 1.0 : 0.0;
        lu_solve_block<CPL, H>(F, e, lane);
#pragma unroll
        for (int c = 0; c < CPL; c++) F.sp[c] = e[c];
        const double mine[2] = {F.couple * (comm.half == 0 ? readlane_d(e[CPL - 1], WAVE - 1) : readlane_d(e[0], 0)),
                                F.couple};
        double theirs[2];
        comm.xchg(mine, theirs);
        F.g_up = comm.half == 0 ? mine[0] : theirs[0];
        F.g_lo = comm.half == 0 ? theirs[0] : mine[0];
        F.couple_other = theirs[1];
        F.inv_den = fast_div(1.0, 1.0 - F.g_up * F.g_lo);
    }
}

template <int CPL, int H>
__device__ __forceinline__ void lu_solve_block(const TriLU<CPL, H> &F, double (&x)[CPL], int lane)
{
#pragma unroll
    for (int c = 1; c < CPL; c++) x[c] -= F.wf[c] * x[c - 1];
#pragma unroll
    for (int c = CPL - 3; c >= 0; c--) x[c] -= F.wb[c] * x[c + 1];
    const double nd = shfl_down1(x[0], lane, 0.0);
    double Rr = x[CPL - 1] - F.wx * nd;
#pragma unroll
    for (int s = 0; s < 6; s++) {
        const int d = 1 << s;
        // A lane without a neighbour at distance d reads some other lane's (finite) value through the wrapped
        // permute address, but its coefficient is an exact zero there: lu_factor masks the neighbour's pivot to 1 and
        // its coupling to 0, and the couplings of the outermost 2^s lanes have been multiplied by that 0 since.
        const double Rm = bpermute_d((lane - d) * 4, Rr), Rp = bpermute_d((lane + d) * 4, Rr);
        Rr = Rr + F.al[s] * Rm + F.ga[s] * Rp;
    }
    const double xe = Rr * F.invB;
    const double xp = shfl_up1(xe, lane, 0.0);
    x[CPL - 1] = xe;
#pragma unroll
    for (int c = 0; c < CPL - 1; c++) x[c] = (x[c] - F.l[c] * xp - F.u[c] * xe) * F.ib[c];
}

template <int CPL, class CommT>
__device__ __forceinline__ void lu_solve(const TriLU<CPL, CommT::H> &F, double (&x)[CPL], int lane, CommT &comm)
{
    lu_solve_block<CPL, CommT::H>(F, x, lane);
    if constexpr (CommT::H == 2) {
        const double mine[1] = {comm.half == 0 ? readlane_d(x[CPL - 1], WAVE - 1) : readlane_d(x[0], 0)};
        double theirs[1];
        comm.xchg(mine, theirs);
        const double z_up = comm.half == 0 ? mine[0] : theirs[0], z_lo = comm.half == 0 ? theirs[0] : mine[0];
        const double x_up = (z_up - F.g_up * z_lo) * F.inv_den;          // last unknown of the upper half
        const double x_lo = z_lo - F.g_lo * x_up;                        // first unknown of the lower half
        const double coef = F.couple * (comm.half == 0 ? x_lo : x_up);
#pragma unroll
        for (int c = 0; c < CPL; c++) x[c] = fma(-coef, F.sp[c], x[c]);
    }
}

// Split column: the Newton solve and the RMS norm of its result in ONE exchange.  With x = z - coef sp in either half,
//   sum ((x_i / scale_i)^2) = S_zz - 2 coef S_zs + coef^2 S_ss,   S_ab = sum (a_i / scale_i)(b_i / scale_i),
// so each half sends its block's edge value of z together with its three partial sums; both halves then know both
// corrections and form the same norm.  (The expansion costs a few digits where z and coef sp nearly cancel; the norm
// only feeds threshold tests at a 3 % tolerance.)
template <int CPL>
__device__ __forceinline__ double lu_solve_norm(const TriLU<CPL, 2> &F, double (&x)[CPL], const double (&is)[CPL],
                                                int lane, double inv_sqrt_d, Comm<2> &comm)
{
    lu_solve_block<CPL, 2>(F, x, lane);
    double szz = 0.0, szs = 0.0, sss = 0.0;
#pragma unroll
    for (int c = 0; c < CPL; c++) {
        const double zi = x[c] * is[c], si = F.sp[c] * is[c];
        szz = fma(zi, zi, szz);
        szs = fma(zi, si, szs);
        sss = fma(si, si, sss);
    }
    wave_sum3(szz, szs, sss);
    const double mine[4] = {comm.half == 0 ? readlane_d(x[CPL - 1], WAVE - 1) : readlane_d(x[0], 0), szz, szs, sss};
    double theirs[4];
    comm.xchg(mine, theirs);
    const bool up = comm.half == 0;
    const double z_up = up ? mine[0] : theirs[0], z_lo = up ? theirs[0] : mine[0];
    const double x_up = (z_up - F.g_up * z_lo) * F.inv_den;
    const double x_lo = z_lo - F.g_lo * x_up;
    const double c_up = (up ? F.couple : F.couple_other) * x_lo;       // the upper half's correction factor
    const double c_lo = (up ? F.couple_other : F.couple) * x_up;       // the lower half's
    const double n_up = fma(c_up, fma(c_up, up ? mine[3] : theirs[3], -2.0 * (up ? mine[2] : theirs[2])), up ? mine[1] : theirs[1]);
    const double n_lo = fma(c_lo, fma(c_lo, up ? theirs[3] : mine[3], -2.0 * (up ? theirs[2] : mine[2])), up ? theirs[1] : mine[1]);
    const double coef = up ? c_up : c_lo;
#pragma unroll
    for (int c = 0; c < CPL; c++) x[c] = fma(-coef, F.sp[c], x[c]);
    double n2 = n_up + n_lo;
    n2 = n2 < 0.0 ? 0.0 : n2;                    // (a NaN stays a NaN: the caller's "not finite" test must see it)
    return sqrt_pos(n2) * inv_sqrt_d;
}

// _sparse_num_jac bookkeeping for ONE column j: max |f_new - f| over the stored rows j-1, j, j+1
// (first maximum in row order) and scale = max(|f|, |f_new|) at that row.  An all-zero column makes
// scipy's sparse argmax return row 0 (scipy/sparse/_data.py, _arg_min_or_max_axis).
__device__ __forceinline__ void col_stats(bool hasU, bool hasD, double fnU, double fbU, double fnM, double fbM,
                                          double fnD, double fbD, double fn_row0, double fb_row0, double &md,
                                          double &sc)
{
    const double dU = hasU ? fabs(fnU - fbU) : -1.0;
    const double dM = fabs(fnM - fbM);
    const double dD = hasD ? fabs(fnD - fbD) : -1.0;
    double sf = fbU, sn = fnU;
    md = dU;
    if (!(dU >= dM)) { md = dM; sf = fbM; sn = fnM; }
    if (hasD && dD > md) { md = dD; sf = fbD; sn = fnD; }
    if (!(md > 0.0)) {
        md = 0.0;
        sf = fb_row0;
        sn = fn_row0;
    }
    sc = fmax(fabs(sf), fabs(sn));
}


// PH_FBASE (TWO layout, round 5): f(y_predict) evaluated again when a Jacobian refresh needs it as its base value, instead
// of stored at the first Newton iterate of every step (see HC_TWO_FP_LAZY)
enum Phase {
    PH_F0 = 0, PH_F1, PH_JAC, PH_JAC_REDO, PH_NEWTON,
    C_JAC_FIN, C_STEP_BEGIN, C_STEP_TRY, C_NEWTON_BEGIN, C_NEWTON_FAIL, C_ERR_TEST, C_ACCEPT, PH_FBASE,
    C_SUCCESS, C_FAIL
};

// HALVES = 2: a member's column is split over the waves 2p and 2p + 1 of the workgroup (Comm<2>, hc_device.h): SLOTS is
// what ONE wave holds, the tables cover TSLOTS = 2 SLOTS slots, and each wave works on its half through `tabw` / `gtabw`.
// PMULTI (split column only): the several-points scheduler compiled in.  Its mere presence cost the single-point split
// kernel 3.4 % through register allocation (95.8 -> 92.5 k at D = 581), so a lone point keeps a build without it.
template <int CPL, bool SPECIAL, int WPB, bool PREDICT, int HALVES = 1, bool PMULTI = false>
__global__ __launch_bounds__(WPB * WAVE, WPB / 4) void step_kernel(const StepArgs A)
{
    constexpr int SLOTS = WAVE * CPL;
    constexpr int TSLOTS = SLOTS * HALVES;
    extern __shared__ double lds[];
    double *tab = lds;
    constexpr int NTAB_L = ntab_lds(TSLOTS, SPECIAL);         // tables this kernel stages (the last one, T_RDELTA, only where it is read)
    signed char *gtab = reinterpret_cast<signed char *>(tab + NTAB_L * TSLOTS);     // group ids < 16: a byte each
    double *wave_base = reinterpret_cast<double *>(gtab + 4 * TSLOTS);
    constexpr int NVEC_K = lds_vectors(CPL, HALVES, SPECIAL);
    constexpr int NEXTRA = lds_extra(CPL, HALVES, SPECIAL);
    constexpr bool J_LDS = NEXTRA == 4;
    constexpr bool TWO = two_of(CPL, HALVES, SPECIAL);       // hand-placed state, two waves per SIMD (see two_of)
// Round 5, stores of the TWO layout's global region that bought nothing (each -D...=0 restores the round-4 form for A/B):
//  * HC_TWO_FP_LAZY: the base value of a Jacobian REFRESH, f(y_predict), was stored at the first Newton iterate of every
//    step (6.1 per column-step) and read by 0.004 refreshes per column-step.  It is now recomputed when a refresh happens:
//    one more RHS evaluation at the predictor's state -- rebuilt from the difference rows by the same additions, so the
//    same bits -- that the solver statistics do not count (scipy's fun(t_new, y_predict) of that iterate is the same call).
//  * HC_TWO_J_ZERO_ONCE: the Jacobian rows were zeroed at the start of every attempt.  Every entry with a column is
//    rewritten by the group scatter before anything reads it; entries without one (row 0's sub-diagonal, the last row's
//    super-diagonal, the padding) are zero from the wave's first pass on (C_JAC_FIN stores 0.0 there) -- so they are
//    zeroed once per launch.  With the rows gathered from the group evaluations (4 cells per lane) nothing is needed.
//  * HC_TWO_Y_LAZY: the accepted-state vector is not rewritten with the row's start state at every attempt (it still holds
//    it: the TWO layout only stores an accepted state when it can be the row's answer), and the separate copy of the
//    row-start state is only kept where the spin-up stop rule reads it.
#ifndef HC_TWO_FP_LAZY
#define HC_TWO_FP_LAZY 1
#endif
#ifndef HC_TWO_J_ZERO_ONCE
#define HC_TWO_J_ZERO_ONCE 1
#endif
#ifndef HC_TWO_Y_LAZY
#define HC_TWO_Y_LAZY 1
#endif
#ifndef HC_TWO_TOP2_LATER      // see accept_update
#define HC_TWO_TOP2_LATER 1
#endif
#ifndef HC_TWO_PARTS
#define HC_TWO_PARTS 31       // development: bit 0 factorisation, 1 Jacobian rows, 2 predicted state, 3 group ids, 4 row-start state
#endif
    constexpr bool TWO_F = TWO && (HC_TWO_PARTS & 1), TWO_J = TWO && (HC_TWO_PARTS & 2), TWO_YP = TWO && (HC_TWO_PARTS & 4);
    constexpr bool FP_LAZY = TWO_YP && HC_TWO_FP_LAZY, J_ZERO_ONCE = TWO_J && HC_TWO_J_ZERO_ONCE;
    constexpr bool Y_LAZY = TWO && (HC_TWO_PARTS & 16) && HC_TWO_Y_LAZY;
    constexpr bool TOP2_LATER = TWO && HC_TWO_TOP2_LATER;
    // Group evaluations of the FD Jacobian stored whole and the rows gathered once (WaveVecs::stG / ldG) instead of a
    // read-modify-write of the three rows per group; the column parameters re-read from scalar memory at every RHS
    // evaluation instead of ~60 SGPRs held (and spilled) across the phases.  Measured at 65 536 members
    // (profiles/r04_two_layout_ab.txt, digests identical): 4 cells per lane +1.4 % with both (each alone: +0.2 % / -1.4 %);
    // 5 cells per lane +3.3 % with the reload alone, -13 % with the gather (its per-lane offsets cost 100 B of scratch;
    // -15 % again after the traffic cuts below).
    // Traffic of the global region (rocprofv3 FETCH / WRITE_SIZE: 188 + 221 KB per column-step at 5 cells per lane before):
    // the FD steps move only where they change, the Jacobian's base point is recomputed from the difference rows instead
    // of parked, the accepted state is stored when it can be the row's answer: +8.0 % at 5 cells, +3.2 % at 4, same bits.
    // (Non-temporal stores for the write-mostly vectors: -3 %.)
#ifndef HC_TWO_JGATHER_MAX_CPL
#define HC_TWO_JGATHER_MAX_CPL 4
#endif
    constexpr bool JG = TWO_J && CPL <= HC_TWO_JGATHER_MAX_CPL && !HC_TWO_J_ALIAS;   // (with the alias the scatter stays in LDS)
#ifdef HC_TWO_NO_P_RELOAD
    constexpr bool P_RELOAD = false;
#else
    constexpr bool P_RELOAD = TWO;
#endif
    constexpr int WAVE_LDS = lds_wave_doubles(CPL, HALVES, SPECIAL);      // the wave's vectors (TWO: + the factorisation's slots)
    constexpr int WSTRIDE = WAVE_LDS + WAVE_SCRATCH;                      // doubles per wave
    static_assert(WPB == wpb_of(CPL, HALVES, SPECIAL), "the workgroup size the LDS layout was sized for");
    // chunk bookkeeping of the multi-point mode lives in the spare fourth row of the group-id table:
    // [0] next member of the chunk, [1] its end (-1: no chunk left), [2] point whose tables are in LDS, [3] the chunk's point
    volatile int *chunk_state = reinterpret_cast<volatile int *>(gtab + NGTAB * TSLOTS);
#ifdef HC_SINGLE_POINT   // development builds: the scheduler of the multi-point mode compiled out (A/B timing)
    constexpr bool multi = false;
#else
    const bool multi = (HALVES == 1 || PMULTI) && A.n_points > 1;      // (round 4: the split column serves several points too)
#endif
    if (!multi)
        for (int k = threadIdx.x; k < NTAB_L * TSLOTS; k += WPB * WAVE) tab[k] = A.tab[k];
    for (int k = threadIdx.x; k < NGTAB * TSLOTS; k += WPB * WAVE) gtab[k] = (signed char)A.gtab[k];
    if (threadIdx.x == 0) {
        chunk_state[0] = 0;
        chunk_state[1] = 0;
        chunk_state[2] = -1;
        chunk_state[3] = 0;
    }
    __syncthreads();

    // No barrier below this line: every wave is an independent worker.  The grid is persistent (one
    // workgroup per CU); a wave that finishes its member takes the next one from a device-wide ticket,
    // so members of unequal cost never leave SIMDs idle behind a slow neighbour.  Exit: ticket >= N.
    const int wave = threadIdx.x / WAVE, lane = threadIdx.x % WAVE;
    const int hb = HALVES == 2 ? (wave & 1) * SLOTS : 0;          // this wave's first node
    const double *tabw = tab + hb;
    const signed char *gtabw = gtab + hb;
    double *V = wave_base + (size_t)wave * WSTRIDE;
    double *ru = V + (J_LDS ? NVEC_K * SLOTS : WAVE_LDS);
    double *jx = ru + WAVE_SCRATCH;       // [4][SLOTS] when J_LDS: Jacobian rows (sub, main, super) and FD steps
#define HC_J_LOAD()                                                        \
    if constexpr (J_LDS) {                                                 \
        _Pragma("unroll") for (int c = 0; c < CPL; c++) {                  \
            const int s_ = c * WAVE + lane;                                \
            jl[c] = jx[s_];                                                \
            jd[c] = jx[SLOTS + s_];                                        \
            ju[c] = jx[2 * SLOTS + s_];                                    \
            hj[c] = jx[3 * SLOTS + s_];                                    \
        }                                                                  \
    } else if constexpr (TWO_J) {                                          \
        _Pragma("unroll") for (int c = 0; c < CPL; c++) {                  \
            const int s_ = c * WAVE + lane;                                \
            jl[c] = W.template ldJ<V_JL>(s_);                              \
            jd[c] = W.template ldJ<V_JD>(s_);                              \
            ju[c] = W.template ldJ<V_JU>(s_);                              \
            hj[c] = W.template ldJ<V_HJ>(s_);                              \
        }                                                                  \
    }
#define HC_J_STORE()                                                       \
    if constexpr (J_LDS) {                                                 \
        _Pragma("unroll") for (int c = 0; c < CPL; c++) {                  \
            const int s_ = c * WAVE + lane;                                \
            jx[s_] = jl[c];                                                \
            jx[SLOTS + s_] = jd[c];                                        \
            jx[2 * SLOTS + s_] = ju[c];                                    \
            jx[3 * SLOTS + s_] = hj[c];                                    \
        }                                                                  \
    } else if constexpr (TWO_J) {                                          \
        _Pragma("unroll") for (int c = 0; c < CPL; c++) {                  \
            const int s_ = c * WAVE + lane;                                \
            W.template stJ<V_JL>(s_, jl[c]);                               \
            W.template stJ<V_JD>(s_, jd[c]);                               \
            W.template stJ<V_JU>(s_, ju[c]);                               \
            W.template stJ<V_HJ>(s_, hj[c]);                               \
        }                                                                  \
    }
// TWO layout: the three rows without the FD steps (a Jacobian's steps change only where they are computed), and the steps alone;
// with the rows in LDS (two cells per lane) these are the four-vector forms above
#define HC_J_LOAD3()                                                       \
    if constexpr (TWO_J) {                                                 \
        _Pragma("unroll") for (int c = 0; c < CPL; c++) {                  \
            const int s_ = c * WAVE + lane;                                \
            jl[c] = W.template ld<V_JL>(s_);                               \
            jd[c] = W.template ld<V_JD>(s_);                               \
            ju[c] = W.template ld<V_JU>(s_);                               \
        }                                                                  \
    } else {                                                               \
        HC_J_LOAD();                                                       \
    }
#define HC_J_STORE3()                                                      \
    if constexpr (TWO_J) {                                                 \
        _Pragma("unroll") for (int c = 0; c < CPL; c++) {                  \
            const int s_ = c * WAVE + lane;                                \
            W.template st<V_JL>(s_, jl[c]);                                \
            W.template st<V_JD>(s_, jd[c]);                                \
            W.template st<V_JU>(s_, ju[c]);                                \
        }                                                                  \
    } else {                                                               \
        HC_J_STORE();                                                      \
    }
#define HC_HJ_LOAD()                                                                               \
    if constexpr (TWO_J) {                                                                         \
        _Pragma("unroll") for (int c = 0; c < CPL; c++) hj[c] = W.template ldJ<V_HJ>(c * WAVE + lane); \
    }
// TWO layout: the factorisation is written once per lu_factor and read by every Newton iteration (no register of it is
// live across an RHS evaluation); the predicted state / Jacobian base point likewise
// (entries the solve never reads -- wf[0], wb of the last two cells, l / u / 1/b of the chunk's last row, which only feed
//  the reduced system inside lu_factor -- have no lane-slot: 6 of 25 per lane were stored for nothing in round 4)
#define HC_F_STORE()                                                       \
    if constexpr (TWO_F) {                                                   \
        _Pragma("unroll") for (int c = 0; c < CPL; c++) {                  \
            if (c >= 1) W.stF(W.K_WF + c, lane, F.wf[c]);                  \
            if (c <= CPL - 3) W.stF(W.K_WB + c, lane, F.wb[c]);            \
            if (c <= CPL - 2) {                                            \
                W.stF(W.K_L + c, lane, F.l[c]);                            \
                W.stF(W.K_U + c, lane, F.u[c]);                            \
                W.stF(W.K_IB + c, lane, F.ib[c]);                          \
            }                                                              \
        }                                                                  \
        W.stF(W.K_S + 0, lane, F.wx);                                      \
        W.stF(W.K_S + 1, lane, F.invB);                                    \
        _Pragma("unroll") for (int q = 0; q < 6; q++) {                    \
            W.stF(W.K_S + 2 + q, lane, F.al[q]);                           \
            W.stF(W.K_S + 8 + q, lane, F.ga[q]);                           \
        }                                                                  \
        if constexpr (HALVES == 2) {                                       \
            _Pragma("unroll") for (int c = 0; c < CPL; c++) W.stF(W.K_SP + c, lane, F.sp[c]); \
            const double uni_ = lane == 0 ? F.couple : lane == 1 ? F.couple_other : lane == 2 ? F.g_up : \
                                lane == 3 ? F.g_lo : F.inv_den;            \
            W.stF(W.K_UNI, lane, uni_);                                    \
        }                                                                  \
    }
#define HC_F_LOAD()                                                        \
    if constexpr (TWO_F) {                                                   \
        _Pragma("unroll") for (int c = 0; c < CPL; c++) {                  \
            F.wf[c] = c >= 1 ? W.ldF(W.K_WF + c, lane) : 0.0;              \
            F.wb[c] = c <= CPL - 3 ? W.ldF(W.K_WB + c, lane) : 0.0;        \
            if (c <= CPL - 2) {                                            \
                F.l[c] = W.ldF(W.K_L + c, lane);                           \
                F.u[c] = W.ldF(W.K_U + c, lane);                           \
                F.ib[c] = W.ldF(W.K_IB + c, lane);                         \
            } else {                                                       \
                F.l[c] = F.u[c] = 0.0;                                     \
                F.ib[c] = 1.0;                                             \
            }                                                              \
        }                                                                  \
        F.wx = W.ldF(W.K_S + 0, lane);                                     \
        F.invB = W.ldF(W.K_S + 1, lane);                                   \
        _Pragma("unroll") for (int q = 0; q < 6; q++) {                    \
            F.al[q] = W.ldF(W.K_S + 2 + q, lane);                          \
            F.ga[q] = W.ldF(W.K_S + 8 + q, lane);                          \
        }                                                                  \
        if constexpr (HALVES == 2) {                                       \
            _Pragma("unroll") for (int c = 0; c < CPL; c++) F.sp[c] = W.ldF(W.K_SP + c, lane); \
            const double uni_ = W.ldF(W.K_UNI, lane);                      \
            F.couple = readlane_d(uni_, 0);                                \
            F.couple_other = readlane_d(uni_, 1);                          \
            F.g_up = readlane_d(uni_, 2);                                  \
            F.g_lo = readlane_d(uni_, 3);                                  \
            F.inv_den = readlane_d(uni_, 4);                               \
        }                                                                  \
    }
// TWO layout: the Jacobian's base point is not kept anywhere: it is D[0] for the first Jacobian of an attempt (y0) and the
// predictor's sum of the difference rows -- unchanged since the step was entered -- for a refresh; the same additions in
// the same order as predict<>, so the same bits (round 4: 15 KB of stores per column-step less than parking it)
#define HC_YP_STORE()
#define HC_YP_LOAD()                                                                               \
    if constexpr (TWO_YP) {                                                                        \
        if (jac_init) {                                                                            \
            _Pragma("unroll") for (int c = 0; c < CPL; c++) yp[c] = W.ldD(0, c * WAVE + lane);       \
        } else {                                                                                   \
            switch (order) {                                                                       \
                case 1: predict_y<CPL, 1>(W, lane, yp); break;                                     \
                case 2: predict_y<CPL, 2>(W, lane, yp); break;                                     \
                case 3: predict_y<CPL, 3>(W, lane, yp); break;                                     \
                case 4: predict_y<CPL, 4>(W, lane, yp); break;                                     \
                default: predict_y<CPL, 5>(W, lane, yp); break;                                    \
            }                                                                                      \
        }                                                                                          \
    }
    // f_new[group][row 0], <= 16 groups: row 0 belongs to the upper half, the lower half reads it there
    double *row0 = (HALVES == 2 && (wave & 1) ? ru - WSTRIDE : ru) + 108;
    Comm<HALVES> comm;
    comm.half = HALVES == 2 ? (wave & 1) : 0;
    // the partner wave's noise vector (the lower half's first cell uses n_rnd of the upper half's last node, its
    // top-node cell n_rnd[0])
    const double *nz_partner = V + (HALVES == 2 ? ((wave & 1) ? -1 : 1) * WSTRIDE : 0) +
                               lds_listed(CPL, HALVES, SPECIAL) * SLOTS;
    if constexpr (HALVES == 2) {
        PairBox *boxes = reinterpret_cast<PairBox *>(wave_base + (size_t)WPB * WSTRIDE);
        comm.lane = lane;
        comm.k = 0;
        comm.dead = 0;
        comm.fault = load_const(A.io).counters + 5;
        comm.box = (Comm<2>::LdsBox *)(boxes + (wave >> 1));
        if (lane < 2) comm.box->seq[lane] = 0;       // both halves write the same zeros, before the first exchange of either
        __syncthreads();
    }
    int *flags_lds = reinterpret_cast<int *>(ru + 124);   // per-lane column flags of the FD-Jacobian retry pass
#ifdef HC_PROFILE
    unsigned long long *prof_lds = reinterpret_cast<unsigned long long *>(ru + 160);
    prof_lds[lane] = 0;          // [0..31] cycles per region, [32..63] entries per region, [64..95] sub-region entries
    if (lane < 32) prof_lds[64 + lane] = 0;
    int prof_slot = 31;
#endif
    WaveVecs<CPL, HALVES, SPECIAL> W;
    W.lds = V;
    // (TWO: the wave index as a scalar, so that the region's base is an SGPR pair and every access is "scalar base +
    //  lane offset + immediate" -- with a per-lane base hipcc keeps one 64-bit VGPR address per vector and cell,
    //  ~170 registers of them, and spills those)
    W.spill = (__attribute__((address_space(1))) double *)A.wave_spill +
              ((size_t)blockIdx.x * WPB + (W.RSRC ? uniform_i(wave) : wave)) * (size_t)spill_doubles(CPL, HALVES, SPECIAL);
    if constexpr (W.RSRC) W.bind(A.wave_spill + ((size_t)blockIdx.x * WPB + uniform_i(wave)) * (size_t)spill_doubles(CPL, HALVES, SPECIAL));
    if constexpr (TWO && HALVES == 2)
        W.bind_partner(A.wave_spill + ((size_t)blockIdx.x * WPB + (uniform_i(wave) ^ 1)) * (size_t)spill_doubles(CPL, HALVES, SPECIAL));
    change_D_init(ru, lane);
    if constexpr (J_ZERO_ONCE && !JG) {
#pragma unroll
        for (int c = 0; c < CPL; c++) {
            W.template st<V_JL>(c * WAVE + lane, 0.0);
            W.template st<V_JD>(c * WAVE + lane, 0.0);
            W.template st<V_JU>(c * WAVE + lane, 0.0);
        }
    }
    const int D = A.D;
    const double inv_sqrt_d = 1.0 / sqrt((double)D);

    for (;;) {
    long long member;
    int point = 0;
    if (!multi) {
        const IoArgs io = load_const(A.io);
        unsigned long long ticket = 0;
        if (lane == 0 && comm.half == 0) ticket = atomicAdd(io.queue, 1ull);
        member = (long long)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(ticket >> 32)) << 32) |
                             (unsigned)__builtin_amdgcn_readfirstlane((int)ticket));
        if constexpr (HALVES == 2) {       // the upper half draws the member, the lower half learns it
            const double mine[1] = {(double)member};
            double theirs[1];
            comm.xchg(mine, theirs);
            member = comm.half == 0 ? member : (long long)theirs[0];
            if (comm.dead) break;          // (an exchange timed out: see Comm<2>::xchg; the host reports it)
        }
        if (member >= A.n_members) break;
    } else {
        // Several parameter points: the waves draw members of the workgroup's current chunk from an LDS counter.
        // A wave that finds the chunk empty waits for the others (the only barriers of the kernel), then one
        // thread takes the next chunk from the device-wide ticket and, when its point differs from the one whose
        // tables are in LDS, the workgroup reloads them.  Exit: no chunk left, seen by all waves together.
        bool none_left = false;
        for (;;) {
            int m = 0;
            if (lane == 0 && comm.half == 0) m = __hip_atomic_fetch_add(const_cast<int *>(chunk_state), 1, __ATOMIC_RELAXED,
                                                                        __HIP_MEMORY_SCOPE_WORKGROUP);
            m = __builtin_amdgcn_readfirstlane(m);
            if constexpr (HALVES == 2) {
                // split column: the upper half draws, the lower half learns the draw; both then take the same path through
                // the barriers below (the chunk bookkeeping only changes between barriers that both halves take part in)
                const double mine[1] = {(double)m};
                double theirs[1];
                comm.xchg(mine, theirs);
                m = comm.half == 0 ? m : (int)theirs[0];
                if (comm.dead) {                 // (an exchange timed out: a bug, never the data -- leave; a terminated
                    none_left = true;            //  wave no longer counts at the workgroup's barriers)
                    break;
                }
            }
            // (LDS reads are per-lane values to the compiler: readfirstlane keeps the control flow scalar)
            if (m < __builtin_amdgcn_readfirstlane(chunk_state[1])) {
                member = m;
                point = __builtin_amdgcn_readfirstlane(chunk_state[3]);
                break;
            }
            __syncthreads();
            if (threadIdx.x == 0) {
                const IoArgs io = load_const(A.io);
                const unsigned long long c = atomicAdd(io.queue, 1ull);
                if (c < (unsigned long long)A.n_chunks) {
                    // longest expected first: the ticket walks the points in the order the host derived from the
                    // previous launch's per-point cost, so the launch does not end on the costliest point's chunks
                    const int pt = io.point_order[c / (unsigned)A.chunks_per_point];
                    const long long first = (long long)pt * A.members_per_point +
                                            (long long)(c % (unsigned)A.chunks_per_point) * A.chunk_members;
                    const long long last = (long long)(pt + 1) * A.members_per_point;
                    chunk_state[3] = pt;
                    chunk_state[0] = (int)first;
                    chunk_state[1] = (int)(first + A.chunk_members < last ? first + A.chunk_members : last);
                } else {
                    chunk_state[0] = 0;
                    chunk_state[1] = -1;
                }
            }
            __syncthreads();
            if (__builtin_amdgcn_readfirstlane(chunk_state[1]) < 0) {
                none_left = true;
                break;
            }
            const int pt = __builtin_amdgcn_readfirstlane(chunk_state[3]);
            if (pt != __builtin_amdgcn_readfirstlane(chunk_state[2])) {
                const double *src = A.tab + (size_t)pt * (NTAB * TSLOTS);
                for (int k = threadIdx.x; k < NTAB_L * TSLOTS; k += WPB * WAVE) tab[k] = src[k];
                __syncthreads();
                if (threadIdx.x == 0) chunk_state[2] = pt;   // next read: after the first barrier of the next chunk change
            }
        }
        if (none_left) break;
    }
    point = __builtin_amdgcn_readfirstlane(point);
    bool vnode[CPL];
#pragma unroll
    for (int c = 0; c < CPL; c++) vnode[c] = hb + lane * CPL + c < D;
    // FD-Jacobian column groups of this lane's nodes and of their neighbours.  Up to CPL = 5 they sit in 3 CPL registers
    // for the kernel's lifetime; deep columns (already spilling to scratch) read them from LDS at the top of each
    // Jacobian phase instead (HC_GROUPS: a batch of byte loads through a pointer made opaque at that point, so that
    // the loads are not hoisted back out of the loop).
    // Which of the two is used where is decided by measurement and by bit-equality with the plain layout on Philox
    // runs (tools/prof_depth.py prints a digest of the final state): group ids on demand for CPL 6-8 (+4..10 %),
    // the row-start state in the global region for CPL 9-10 (+4 %).
    // (measured neutral at CPL = 5; with the split column: 40 % less scratch, 0.8 % slower; at two waves per SIMD every
    //  register counts: D = 101 +3 %, D = 192 +11 %)
    constexpr bool DEEP = CPL >= 6 || CPL <= 3 || (TWO && (HC_TWO_PARTS & 8));     // (CPL = 4, one wave per SIMD: -1 %)
    // the row-start state out of registers: CPL 9-10 (with DEEP at CPL <= 3: D = 101 +1.6 %, D = 192 -9.5 %; at CPL = 8: -3 %).
    // At CPL 9-10 both are on since round 3 (+7.7 % / +3.5 % on the one-wave kernels; the combination round 2 saw
    // miscompiled belonged to a noise path that no longer exists -- DESIGN.md §5 "Deep columns";
    // test_one_wave_kernels_of_the_deepest_columns_keep_their_guards holds the two symptoms of that build against it).
    constexpr bool DEEPY = CPL >= 9 || (TWO && (HC_TWO_PARTS & 16));
    int gs_keep[CPL], gp_keep[CPL], gn_keep[CPL];
    if (!DEEP) {
#pragma unroll
        for (int c = 0; c < CPL; c++) {
            gs_keep[c] = gtabw[G_SELF * TSLOTS + c * WAVE + lane];
            gp_keep[c] = gtabw[G_PREV * TSLOTS + c * WAVE + lane];
            gn_keep[c] = gtabw[G_NEXT * TSLOTS + c * WAVE + lane];
        }
    }
#define HC_GROUPS()                                                                          \
    int gs[CPL], gp[CPL], gn[CPL];                                                           \
    {                                                                                        \
        auto gq = (const __attribute__((address_space(3))) signed char *)gtabw;              \
        if (DEEP) asm volatile("" : "+v"(gq));                                               \
        _Pragma("unroll") for (int c = 0; c < CPL; c++) {                                    \
            gs[c] = DEEP ? (int)gq[G_SELF * TSLOTS + c * WAVE + lane] : gs_keep[c];          \
            gp[c] = DEEP ? (int)gq[G_PREV * TSLOTS + c * WAVE + lane] : gp_keep[c];          \
            gn[c] = DEEP ? (int)gq[G_NEXT * TSLOTS + c * WAVE + lane] : gn_keep[c];          \
        }                                                                                    \
    }
    // state -> LDS
    double nscale = 1.0;
    {
        const IoArgs io = load_const(A.io);
#pragma unroll
        for (int c = 0; c < CPL; c++)
            W.template st<V_Y>(c * WAVE + lane, vnode[c] ? io.psi[member * D + hb + lane * CPL + c] : 0.0);
        if (!A.host_noise) nscale = io.nscale[member];
    }
    int fresh_seen = 0;
    bool nz_is_base = false;     // V_NZ holds this member's base vector exactly as a fresh generation would give it
    int cost_nfev = 0;           // RHS evaluations of this member in this launch (multi-point mode: per-point cost)

    for (int r = 0; r < A.n_rows; r++) {
        RowDev R;
        long long row;
        bool refresh;
        {
            const IoArgs io = load_const(A.io);
            row = A.spinup ? io.row_begin : io.row_begin + r;
            R.precip = io.precip[row];
            R.atm = io.atm[row];
            const int day_bits = io.daylight[row];   // bit 0: daylight, bit 1: wet season (PREDICT mode)
            R.daylight = day_bits & 1;
            R.wet = (day_bits >> 1) & 1;
            R.wtd_obs = io.wtd_obs[row];
            R.spinup = A.spinup;
            R.diag = io.diag != nullptr;
            refresh = !A.spinup && io.refresh[row];
        }
        double diag_tr = 0.0, diag_lf = 0.0, spin_mse = INFINITY;
        const double t0 = A.spinup ? 0.0 : (double)(row - 1);
        const double tf = t0 + 1.0;
        int st_nfev = 0, st_njev = 0, st_nlu = 0, st_nsteps = 0, attempts = 0, failed_row = 0;
        bool skip = (R.wtd_obs < 0) && !A.spinup;    // simulation.py:582-588
        if (!skip) {
            // ---- noise vector of this row -> V_NZ (simulation.py:592,599-602)
            // The base vector stays in LDS from row to row: it is only rebuilt after a refresh row replaced it or a
            // failed attempt damped it in place (x0.8 on the LDS copy and x0.8 on the scale round differently, and
            // the result must not depend on where a launch boundary falls).  Box-Muller per row cost 2-3 % before.
            unsigned draw_row = 0u;
            {
                const IoArgs io = load_const(A.io);
                draw_row = A.spinup ? PHILOX_DRAW_SPINUP : ((refresh && !A.host_noise) ? (unsigned)io.draw_idx[row] : 0u);
            }
            if (refresh || !nz_is_base) {
                const IoArgs io = load_const(A.io);
                // Philox draws mirror the reference's order (simulation.py:426,561,601): the spin-up vector has
                // its own index, 0 is the base vector, refresh row k uses draw k >= 1
                const unsigned draw = draw_row;
                // the member's global id keys its Philox stream; with several points each point has a base of its own
                long long gid = io.member_offset + member;
                if (multi && !A.host_noise) gid = io.point_base[point] + (member - (long long)point * A.members_per_point);
#pragma unroll
                for (int c = 0; c < CPL; c++) {
                    const int i = hb + lane * CPL + c;
                    double z = 0.0;
                    if (vnode[c]) {
                        if (A.host_noise) {
                            z = refresh ? io.fresh[((size_t)fresh_seen * A.n_members + member) * D + i]
                                        : io.base_noise[member * D + i];
                        } else {
                            z = philox_normal(io.seed, (unsigned long long)gid, draw, (unsigned)i);
                            z = refresh ? z : z * nscale;
                        }
                    }
                    W.template st<V_NZ>(c * WAVE + lane, z);
                }
                nz_is_base = !refresh;
            }
            __builtin_amdgcn_wave_barrier();
            int failed = 0;
            // every attempt restarts from the row's y0: V_Y holds it when the row begins; a copy goes to the global
            // region (fire-and-forget stores) for the rare retry of a deep column, instead of CPL register pairs held across
            // the integrator
            // (measured at CPL = 5: the register copy is 0.5 % faster, so it stays there)
            double yrow0[DEEPY ? 1 : CPL];
            if (!Y_LAZY || A.spin_stop) {
#pragma unroll
                for (int c = 0; c < CPL; c++) {
                    const double y0c = W.template ld<V_Y>(c * WAVE + lane);
                    if (DEEPY) W.template st<V_Y0>(c * WAVE + lane, y0c);
                    else yrow0[DEEPY ? 0 : c] = y0c;
                }
            }
            // ---- up to 5 attempts (richards_pde.py:509-533)
            for (;;) {
                attempts++;
                // scaled noise per evaluated cell: midpoint j uses n_rnd[max(j-1,0)]; the virtual
                // top-node cell (lane 63, last slot) uses n_rnd[0]   
                double rnd[CPL];
                if constexpr (HALVES == 2) {       // both halves' noise vectors are in place before either reads across the cut
                    // (TWO layout: "in place" = this wave's stores to its global region have completed)
                    if constexpr (TWO) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    const double mine[1] = {0.0};
                    double theirs[1];
                    comm.xchg(mine, theirs);
                }
                // (noise vector in the wave's global region: its stores -- generation, or the x0.8 of a failed attempt -- have
                //  landed before other lanes' slots are read; at workgroup scope this is an s_waitcnt, no cache operation)
                if constexpr (TWO && TWO_NZ_GLOBAL) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
#pragma unroll
                for (int c = 0; c < CPL; c++) {
                    const int i = hb + lane * CPL + c;
                    int idx = i >= 1 ? i - 1 : 0;
                    idx = (i < D - 1) ? idx : 0;
                    double z;
                    if constexpr (HALVES == 2) {
                        const int wl = idx / CPL, sl = (idx % CPL) * WAVE + (wl & (WAVE - 1));
                        double own, other;
                        if constexpr (TWO) {
                            own = W.gld_fresh(W.NZ_OFF, sl * 8);
                            other = W.gld_partner_fresh(W.NZ_OFF, sl * 8);
                        } else {
                            own = W.template ld<V_NZ>(sl);
                            other = nz_partner[sl];
                        }
                        z = (wl >> 6) == comm.half ? own : other;
                    } else if constexpr (TWO) {
                        z = W.gld_fresh(W.NZ_OFF, ((idx % CPL) * WAVE + idx / CPL) * 8);
                    } else {
                        z = W.template ld<V_NZ>((idx % CPL) * WAVE + idx / CPL);
                    }
                    rnd[c] = tabw[T_NOISEC * TSLOTS + c * WAVE + lane] * z;
                }
                // ================= one BDF integration over [t0, tf] =================
                double ycur[CPL], f[CPL], yp[CPL], psiv[CPL], scl[CPL] /* 1/scale */, dd[CPL];
                double jl[CPL], jd[CPL], ju[CPL], hj[CPL];
                TriLU<CPL, HALVES> F;
                double t = t0, h_abs = 0.0, h0 = 0.0, t_new = t0, cc = 0.0, min_step = 0.0;
                double dy_norm_old = -1.0, safety = 0.0, error_norm = 0.0;
                int order = 1, n_equal = 0, have_lu = 0, current_jac = 0, newton_k = 0, n_iter = 0;
                int g = 0, jac_init = 1, nfev = 0, njev = 0, nlu = 0, nsteps = 0, ok = 0;
                int redo_mask = 0, jac_stage = 0;
#pragma unroll
                for (int c = 0; c < CPL; c++) {
                    // sol.y[:, -1] before any accepted step is y0 itself
                    double y0c;
                    if (Y_LAZY) y0c = W.template ld<V_Y>(c * WAVE + lane);   // (still the row's start state: see HC_TWO_Y_LAZY)
                    else if (DEEPY) y0c = attempts == 1 ? W.template ld<V_Y>(c * WAVE + lane) : W.template ld<V_Y0>(c * WAVE + lane);
                    else y0c = yrow0[DEEPY ? 0 : c];
                    ycur[c] = y0c;
                    if (!Y_LAZY) W.template st<V_Y>(c * WAVE + lane, y0c);
                    W.stD(0, c * WAVE + lane, ycur[c]);
                    W.template st<V_FAC>(c * WAVE + lane, SQRT_EPS);
                    yp[c] = psiv[c] = dd[c] = jl[c] = jd[c] = ju[c] = hj[c] = 0.0;
                    scl[c] = 1.0;
                }
                if constexpr (!J_ZERO_ONCE) { HC_J_STORE(); }
                int phase = PH_F0;
                // column parameters: one scalar-memory read per attempt.  (Per RHS evaluation the lone wave sat out
                // the load latency 24 times per row; for the kernel's lifetime they cost ~60 SGPRs, see DESIGN.md.)
                const ColumnDev P = load_const(A.P + point);
                int guard = 0;                       // every wave must reach an exit: bound the phase loop
#ifdef HC_PROFILE
                unsigned long long prof_t = clock64();
#endif
                for (;;) {
                    HC_STAMP(31);   // loop top: budget check, dispatch to the RHS site
#ifdef HC_PROFILE
                    {   // diagnostic build: (phase, t, h_abs, order, n_equal + 100 * current_jac + 1000 * have_lu, last norm)
                        const IoArgs iot = load_const(A.io);
                        if (iot.trace && member == 0 && lane == 0) {
                            const int k = (int)iot.trace[0];
                            if (k < HC_TRACE_N) {
                                double *q = iot.trace + 1 + 6 * k;
                                q[0] = phase; q[1] = t; q[2] = h_abs; q[3] = order;
                                q[4] = n_equal + 100 * current_jac + 1000 * have_lu + 10000 * newton_k;
                                q[5] = error_norm;
                                iot.trace[0] = k + 1;
                            }
                        }
                    }
#endif
                    if (++guard > A.max_phase_iterations || comm.dead) {
                        if (lane == 0 && comm.half == 0) {
                            const IoArgs iog = load_const(A.io);
                            atomicAdd(&iog.counters[2], 1ull);
                            // where it happened (last writer wins): global member id << 24 | forcing row
                            const long long gidg = multi ? iog.point_base[point] + (member - (long long)point * A.members_per_point)
                                                         : iog.member_offset + member;
                            iog.counters[3] = ((unsigned long long)gidg << 24) | ((unsigned long long)row & 0xFFFFFFull);
                        }
                        phase = C_FAIL;
                    }
                    if (phase < C_SUCCESS) {
                        HC_STAMP(16);   // RHS prologue (midpoints); rhs_eval stamps its own regions 24..29
                        if constexpr (P_RELOAD) {
                            // two waves per SIMD: the column parameters come from scalar memory at every evaluation (the
                            // other wave covers the load) instead of ~60 SGPRs held -- and spilled -- across the phases
                            const ColumnDev Pe = load_const(A.P + point);
                            rhs_eval<CPL, SPECIAL, PREDICT, TWO>(Pe, R, tabw, lane, ycur, rnd, f, nullptr, diag_tr, diag_lf, comm HC_RHS_PROF_ARG);
                        } else {
                            rhs_eval<CPL, SPECIAL, PREDICT, TWO>(P, R, tabw, lane, ycur, rnd, f, nullptr, diag_tr, diag_lf, comm HC_RHS_PROF_ARG);
                        }
                        HC_STAMP(17);   // after the RHS: dispatch to the phase block
                    }
                    // Phases run in topological order inside ONE loop iteration: a block that hands over to a
                    // later block falls through to it; only blocks that need a fresh RHS value end the iteration.
                    bool have_f = true;
                    bool j_fresh = false;      // jl / jd / ju hold the Jacobian C_JAC_FIN has just finished (this trip)
                    if (phase == PH_F0 && have_f) {
                        have_f = false;
                        HC_STAMP(PH_F0);
                        // BDF.__init__: f0 = fun(t0, y0); select_initial_step part 1
                        nfev++;
                        double y0v[CPL];
#pragma unroll
                        for (int c = 0; c < CPL; c++) {
                            const int slot = c * WAVE + lane;
                            y0v[c] = ycur[c];
                            W.stD(1, slot, f[c]);      // parked here until h_abs is known
                            W.template st<V_FP>(slot, f[c]);    // base f of the first Jacobian
                            scl[c] = fast_div(1.0, ATOL + fabs(y0v[c]) * RTOL);
                        }
                        const double d0 = rms_ratio<CPL>(y0v, scl, lane, D, inv_sqrt_d, comm);
                        const double d1 = rms_ratio<CPL>(f, scl, lane, D, inv_sqrt_d, comm);
                        h0 = (d0 < 1e-5 || d1 < 1e-5) ? 1e-6 : 0.01 * d0 / d1;
                        h0 = A.scipy_152 ? h0 : fmin(h0, fabs(tf - t0));
                        cc = d1;                              // carried to PH_F1
#pragma unroll
                        for (int c = 0; c < CPL; c++) {
                            yp[c] = y0v[c];                   // Jacobian base point
                            ycur[c] = y0v[c] + h0 * f[c];
                        }
                        HC_YP_STORE();
                        phase = PH_F1;
                    }
                    if (phase == PH_F1 && have_f) {
                        have_f = false;
                        HC_STAMP(PH_F1);
                        // select_initial_step part 2 (order = 1)
                        nfev++;
                        double df[CPL];
#pragma unroll
                        for (int c = 0; c < CPL; c++) df[c] = f[c] - W.template ld<V_FP>(c * WAVE + lane);
                        const double d1 = cc;
                        const double d2 = rms_ratio<CPL>(df, scl, lane, D, inv_sqrt_d, comm) / h0;
                        double h1;
                        if (d1 <= 1e-15 && d2 <= 1e-15)
                            h1 = fmax(1e-6, h0 * 1e-3);
                        else
                            h1 = sqrt(0.01 / fmax(d1, d2));
                        h_abs = fmin(100.0 * h0, h1);
                        h_abs = A.scipy_152 ? h_abs : fmin(h_abs, fabs(tf - t0));    // (1.5.2: _step_impl's t_bound rule takes care of it)
                        // first Jacobian: num_jac at (t0, y0) with f0 (the reference re-evaluates f0; same value)
                        njev++;
                        jac_init = 1;
                        g = -1;
                        phase = PH_JAC;
                    }
                    if (phase == PH_NEWTON && have_f) {
                        have_f = false;
                        HC_STAMP(PH_NEWTON);
                        // solve_bdf_system, iteration newton_k
                        nfev++;
                        if (!FP_LAZY && newton_k == 0 && !current_jac) {   // f(y_predict): base value of a Jacobian refresh
#pragma unroll
                            for (int c = 0; c < CPL; c++) W.template st<V_FP>(c * WAVE + lane, f[c]);
                        }
                        bool converged = false, failed_newton = false;
                        HC_F_LOAD();
                        {
                            double dy[CPL];
#pragma unroll
                            for (int c = 0; c < CPL; c++) dy[c] = cc * f[c] - psiv[c] - dd[c];   // 0 in the padding slots
                            HC_STAMP(20);
                            double dy_norm;
                            if constexpr (HALVES == 2) {
                                dy_norm = lu_solve_norm<CPL>(F, dy, scl, lane, inv_sqrt_d, comm);
                            } else {
                                lu_solve<CPL>(F, dy, lane, comm);
                                HC_STAMP(21);
                                dy_norm = rms_ratio<CPL>(dy, scl, lane, D, inv_sqrt_d, comm);
                            }
                            HC_STAMP(22);
                            // scipy: `if not np.all(np.isfinite(f)): break`.  A non-finite f makes the solve and its
                            // norm non-finite, and the iterate is left untouched either way.
                            // The decisions are evaluated as flags (no short-circuit control flow: every `&&` on these
                            // wave-uniform doubles costs an exec-mask save / branch / restore triple otherwise).
                            const bool have_rate = dy_norm_old >= 0.0;
                            const double rate = have_rate ? fast_div(dy_norm, dy_norm_old) : 0.0;
                            // rate ** (NEWTON_MAXITER - k), k = 1, 2, 3 (iteration 0 has no rate): products in the
                            // order the power loop took them
                            const double r2 = rate * rate;
                            const double rp = newton_k <= 1 ? r2 * rate : (newton_k == 2 ? r2 : rate);
                            const double slack = NEWTON_TOL * (1.0 - rate);
                            // rate**(4-k) / (1 - rate) * dy_norm > tol, with 0 <= rate < 1 on the right-hand branch
                            const int fail = int(!(dy_norm < INFINITY)) |
                                             (int(have_rate) & (int(rate >= 1.0) | int(rp * dy_norm > slack)));
                            const int conv = int(!fail) & (int(dy_norm == 0.0) | (int(have_rate) & int(rate * dy_norm < slack)));
                            if (!fail) {
#pragma unroll
                                for (int c = 0; c < CPL; c++) {
                                    ycur[c] += dy[c];
                                    dd[c] += dy[c];
                                }
                            }
                            const int more = int(!fail) & int(!conv);
                            dy_norm_old = more ? dy_norm : dy_norm_old;
                            newton_k += more;
                            converged = conv;
                            failed_newton = fail | (more & int(newton_k == NEWTON_MAXITER));
                        }
                        if (converged) {
                            n_iter = newton_k + 1;
                            phase = C_ERR_TEST;
                        } else if (failed_newton) {
                            phase = C_NEWTON_FAIL;
                        } else {
                            continue;      // next Newton iterate: straight back to the RHS site
                        }
                    }
                    if (phase == C_NEWTON_FAIL) {
                        HC_STAMP(C_NEWTON_FAIL);
                        if (current_jac) {
                            h_abs *= 0.5;
                            change_D<CPL>(W, ru, order, 0.5, lane);
                            n_equal = 0;
                            have_lu = 0;
                            phase = C_STEP_TRY;
                        } else if constexpr (FP_LAZY) {
                            // J = jac(t_new, y_predict): its base value f(y_predict) is evaluated again (PH_FBASE below)
                            jac_init = 0;
                            HC_YP_LOAD();
#pragma unroll
                            for (int c = 0; c < CPL; c++) ycur[c] = yp[c];
                            guard--;       // (the extra trip is not the integrator's: the budget counts what round 4 counted)
                            phase = PH_FBASE;
                        } else {
                            // J = jac(t_new, y_predict): base point yp, base f = fun(y_predict) kept in V_FP
                            njev++;
                            jac_init = 0;
                            g = -1;
                            phase = PH_JAC;
                        }
                    }
                    if (phase == PH_FBASE && have_f) {
                        have_f = false;
                        HC_STAMP(PH_FBASE);
#pragma unroll
                        for (int c = 0; c < CPL; c++) W.template st<V_FP>(c * WAVE + lane, f[c]);
                        njev++;
                        g = -1;
                        phase = PH_JAC;
                    }
                    if (phase == PH_JAC_REDO && have_f) {
                        have_f = false;
                        HC_STAMP(PH_JAC_REDO);
                        HC_GROUPS();
                        HC_J_LOAD();
                        HC_YP_LOAD();
                        // f = fun(y + h_new * [column small and in group g]); keep the new column where
                        // max_diff * scale_new < max_diff_new * scale  (common.py _sparse_num_jac)
                        __builtin_amdgcn_wave_barrier();
                        double fb[CPL];
#pragma unroll
                        for (int c = 0; c < CPL; c++) fb[c] = W.template ld<V_FP>(c * WAVE + lane);
                        const double fbU0 = comm.up1(fb[CPL - 1], lane, 0.0), fbD0 = comm.down1(fb[0], lane, 0.0);
                        const double fU0 = comm.up1(f[CPL - 1], lane, 0.0), fD0 = comm.down1(f[0], lane, 0.0);
                        const double fnU0 = comm.up1(ju[CPL - 1], lane, 0.0), fnD0 = comm.down1(jl[0], lane, 0.0);
                        const double fb_row0 = comm.first_row(fb[0]), f_row0 = comm.first_row(f[0]);
                        int flags = flags_lds[lane];
                        double upd[CPL];
#pragma unroll
                        for (int c = 0; c < CPL; c++) {
                            const int i = hb + lane * CPL + c, slot = c * WAVE + lane;
                            const bool hasU = i >= 1, hasD = i < D - 1;
                            const double fbU = c == 0 ? fbU0 : fb[c > 0 ? c - 1 : 0];
                            const double fbD = c == CPL - 1 ? fbD0 : fb[c < CPL - 1 ? c + 1 : c];
                            const double fnU = c == 0 ? fnU0 : ju[c > 0 ? c - 1 : 0];
                            const double fnD = c == CPL - 1 ? fnD0 : jl[c < CPL - 1 ? c + 1 : c];
                            const double f2U = c == 0 ? fU0 : f[c > 0 ? c - 1 : 0];
                            const double f2D = c == CPL - 1 ? fD0 : f[c < CPL - 1 ? c + 1 : c];
                            double md, sc, md2, sc2;
                            col_stats(hasU, hasD, fnU, fbU, jd[c], fb[c], fnD, fbD, row0[gs[c] >= 0 ? gs[c] : 0], fb_row0,
                                      md, sc);
                            col_stats(hasU, hasD, f2U, fbU, f[c], fb[c], f2D, fbD, f_row0, fb_row0, md2, sc2);
                            const bool mine = ((flags >> c) & 1) && gs[c] == g;
                            const bool u = mine && (md * sc2 < md2 * sc);
                            upd[c] = u ? 1.0 : 0.0;
                            if (u) {
                                double fac = 10.0 * W.template ld<V_FAC>(slot);
                                const double ysc = (fb[c] >= 0.0 ? 1.0 : -1.0) * fmax(ATOL, fabs(yp[c]));
                                hj[c] = (yp[c] + fac * ysc) - yp[c];
                                if (md2 < NUM_JAC_DIFF_SMALL * sc2) fac *= 10.0;
                                if (md2 > NUM_JAC_DIFF_BIG * sc2) fac *= 0.1;
                                W.template st<V_FAC>(slot, fmax(fac, NUM_JAC_MIN_FACTOR));
                                flags |= 1 << (16 + c);
                            }
                        }
                        flags_lds[lane] = flags;
                        const double updU0 = comm.up1(upd[CPL - 1], lane, 0.0), updD0 = comm.down1(upd[0], lane, 0.0);
#pragma unroll
                        for (int c = 0; c < CPL; c++) {
                            const double uU = c == 0 ? updU0 : upd[c > 0 ? c - 1 : 0];
                            const double uD = c == CPL - 1 ? updD0 : upd[c < CPL - 1 ? c + 1 : c];
                            jl[c] = uU != 0.0 ? f[c] : jl[c];
                            jd[c] = upd[c] != 0.0 ? f[c] : jd[c];
                            ju[c] = uD != 0.0 ? f[c] : ju[c];
                        }
                        HC_J_STORE();
                        redo_mask &= ~(1 << g);
                        if (redo_mask != 0) {
                            g = __ffs(redo_mask) - 1;
#pragma unroll
                            for (int c = 0; c < CPL; c++) {
                                const double ysc = (fb[c] >= 0.0 ? 1.0 : -1.0) * fmax(ATOL, fabs(yp[c]));
                                const bool done = (flags >> (16 + c)) & 1;   // an updated column already holds 10x its factor
                                const double fac0 = W.template ld<V_FAC>(c * WAVE + lane);
                                const double hn = (yp[c] + 10.0 * fac0 * ysc) - yp[c];
                                ycur[c] = yp[c] + ((((flags >> c) & 1) && !done && gs[c] == g) ? hn : 0.0);
                            }
                        } else {
                            phase = C_JAC_FIN;
                        }
                    }
                    if (phase == PH_JAC) {
                        HC_STAMP(PH_JAC);
                        HC_GROUPS();
                        if constexpr (!TWO_J) { HC_J_LOAD(); }          // (rows in LDS at two cells per lane)
                        HC_YP_LOAD();
                        if (g < 0) {
                            HC_SUB(62);
                            // common.num_jac: step h per column from factor, f sign and |y|
#pragma unroll
                            for (int c = 0; c < CPL; c++) {
                                const int slot = c * WAVE + lane;
                                const double fb = W.template ld<V_FP>(slot);
                                double fac = W.template ld<V_FAC>(slot);
                                const double ysc = (fb >= 0.0 ? 1.0 : -1.0) * fmax(ATOL, fabs(yp[c]));
                                double h = (yp[c] + fac * ysc) - yp[c];
                                for (int it = 0; it < 64 && vnode[c] && h == 0.0; it++) {
                                    fac *= 10.0;
                                    h = (yp[c] + fac * ysc) - yp[c];
                                }
                                W.template st<V_FAC>(slot, fac);
                                hj[c] = vnode[c] ? h : 1.0;
                                if constexpr (TWO_J) W.template stJ<V_HJ>(slot, hj[c]);
                            }
                            g = 0;
                        } else {
                            HC_SUB(63);
                            // f holds fun(y + h * [group == g]); scatter into the three per-row slots
                            const double r0v = readlane_d(f[0], 0);
                            if (lane == 0 && comm.half == 0) row0[g] = r0v;
                            if constexpr (TWO_J && !JG) {
                                // rows in memory: an entry changes in exactly the one group evaluation that perturbed its
                                // column, so the scatter is three PREDICATED stores per cell -- no load, no rewrite of the
                                // entries that stay (the read-modify-write of the register form keeps them too)
#pragma unroll
                                for (int c = 0; c < CPL; c++) {
                                    const int s_ = c * WAVE + lane;
                                    if (gp[c] == g) W.template stJ<V_JL>(s_, f[c]);
                                    if (gs[c] == g) W.template stJ<V_JD>(s_, f[c]);
                                    if (gn[c] == g) W.template stJ<V_JU>(s_, f[c]);
                                }
                                HC_HJ_LOAD();
                            } else if constexpr (JG) {
#pragma unroll
                                for (int c = 0; c < CPL; c++) {
                                    W.stG(g, c * WAVE + lane, f[c]);
                                    hj[c] = W.template ldJ<V_HJ>(c * WAVE + lane);
                                }
                            } else {
#pragma unroll
                                for (int c = 0; c < CPL; c++) {
                                    jl[c] = (gp[c] == g) ? f[c] : jl[c];
                                    jd[c] = (gs[c] == g) ? f[c] : jd[c];
                                    ju[c] = (gn[c] == g) ? f[c] : ju[c];
                                }
                            }
                            g++;
                        }
                        if constexpr (!TWO_J) { HC_J_STORE(); }
                        HC_SUB_END();
                        if (g < A.n_groups) {
#pragma unroll
                            for (int c = 0; c < CPL; c++) ycur[c] = yp[c] + ((gs[c] == g) ? hj[c] : 0.0);
                        } else {
                            phase = C_JAC_FIN;
                        }
                    }
                    if (phase == C_JAC_FIN) {
                        HC_STAMP(C_JAC_FIN);
                        HC_GROUPS();
                        if constexpr (JG) {
                            if (jac_stage == 0) {
                                // entry (row i, column j) = f_i of the evaluation that perturbed column j's group; entries
                                // without a column (row 0's sub-diagonal, the last row's super-diagonal, padding) are masked
                                // below by hasU / hasD / vnode
#pragma unroll
                                for (int c = 0; c < CPL; c++) {
                                    const int s_ = c * WAVE + lane;
                                    jl[c] = W.ldG(gp[c] < 0 ? 0 : gp[c], s_);
                                    jd[c] = W.ldG(gs[c] < 0 ? 0 : gs[c], s_);
                                    ju[c] = W.ldG(gn[c] < 0 ? 0 : gn[c], s_);
                                    hj[c] = W.template ldJ<V_HJ>(s_);
                                }
                            } else {
                                HC_J_LOAD();        // after the retry pass: the rows as it left them
                            }
                        } else {
                            HC_J_LOAD();
                        }
                        HC_YP_LOAD();
                        // _sparse_num_jac: per-column max |diff| (rows j-1, j, j+1), its scale, factor update,
                        // J = diff / h.  jac_stage 0 = first look, 1 = after the retry pass below.
                        __builtin_amdgcn_wave_barrier();
                        double fb[CPL];
#pragma unroll
                        for (int c = 0; c < CPL; c++) fb[c] = W.template ld<V_FP>(c * WAVE + lane);
                        double fbU0, fnU0, fbD0, fnD0, hU0, hD0, fb_row0;
                        if constexpr (HALVES == 2) {      // the six edge values and row 0's base f in two exchanges
                            const double lastc[3] = {fb[CPL - 1], ju[CPL - 1], hj[CPL - 1]}, firstc[3] = {fb[0], jl[0], hj[0]};
                            const double fill[3] = {0.0, 0.0, 1.0};
                            double up[3], down[3];
                            comm.edges(lastc, firstc, fill, lane, up, down);
                            fbU0 = up[0]; fnU0 = up[1]; hU0 = up[2];
                            fbD0 = down[0]; fnD0 = down[1]; hD0 = down[2];
                            fb_row0 = comm.first_row(fb[0]);
                        } else {
                            fbU0 = comm.up1(fb[CPL - 1], lane, 0.0), fnU0 = comm.up1(ju[CPL - 1], lane, 0.0);
                            fbD0 = comm.down1(fb[0], lane, 0.0), fnD0 = comm.down1(jl[0], lane, 0.0);
                            hU0 = comm.up1(hj[CPL - 1], lane, 1.0), hD0 = comm.down1(hj[0], lane, 1.0);
                            fb_row0 = comm.first_row(fb[0]);
                        }
                        const int old_flags = jac_stage ? flags_lds[lane] : 0;     // bit c: small, bit 16+c: factor done
                        int small_bits = 0, my_groups = 0;
                        double njl[CPL], njd[CPL], nju[CPL], nfac[CPL];
#pragma unroll
                        for (int c = 0; c < CPL; c++) {
                            const int i = hb + lane * CPL + c;
                            const double fbU = c == 0 ? fbU0 : fb[c > 0 ? c - 1 : 0];
                            const double fnU = c == 0 ? fnU0 : ju[c > 0 ? c - 1 : 0];
                            const double fbD = c == CPL - 1 ? fbD0 : fb[c < CPL - 1 ? c + 1 : c];
                            const double fnD = c == CPL - 1 ? fnD0 : jl[c < CPL - 1 ? c + 1 : c];
                            const bool hasU = i >= 1, hasD = i < D - 1;
                            double md, sc;
                            col_stats(hasU, hasD, fnU, fbU, jd[c], fb[c], fnD, fbD, row0[gs[c] >= 0 ? gs[c] : 0], fb_row0,
                                      md, sc);
                            const bool small = vnode[c] && (md < A.jac_reject * sc);
                            small_bits |= small ? (1 << c) : 0;
                            my_groups |= small ? (1 << gs[c]) : 0;
                            double fac = W.template ld<V_FAC>(c * WAVE + lane);
                            if (!((old_flags >> (16 + c)) & 1)) {
                                if (md < NUM_JAC_DIFF_SMALL * sc) fac *= 10.0;
                                if (md > NUM_JAC_DIFF_BIG * sc) fac *= 0.1;
                                fac = fmax(fac, NUM_JAC_MIN_FACTOR);
                            }
                            nfac[c] = fac;
                            // J row i: (f_new[i] - f[i]) / h[column]
                            const double hU = c == 0 ? hU0 : hj[c > 0 ? c - 1 : 0];
                            const double hD = c == CPL - 1 ? hD0 : hj[c < CPL - 1 ? c + 1 : c];
                            njl[c] = hasU && vnode[c] ? fast_div(jl[c] - fb[c], hU) : 0.0;
                            njd[c] = vnode[c] ? fast_div(jd[c] - fb[c], hj[c]) : 0.0;
                            nju[c] = hasD ? fast_div(ju[c] - fb[c], hD) : 0.0;
                        }
                        if (jac_stage == 0 && comm.any(small_bits != 0)) {
                            // rare: some column moved f by less than EPS^0.875 of its size -> retry those columns
                            // with a 10x step, group by group (num_jac's diff_too_small branch); nothing committed yet
                            redo_mask = 0;
                            for (int q = 0; q < A.n_groups; q++)
                                if (__any((my_groups >> q) & 1)) redo_mask |= 1 << q;
                            redo_mask = comm.or_bits(uniform_i(redo_mask));
                            flags_lds[lane] = small_bits;
                            __builtin_amdgcn_wave_barrier();
                            if (lane == 0 && comm.half == 0) atomicAdd(&load_const(A.io).counters[0], 1ull);
                            if constexpr (JG) { HC_J_STORE(); }      // the retry pass works on the rows in memory
                            jac_stage = 1;
                            g = __ffs(redo_mask) - 1;
#pragma unroll
                            for (int c = 0; c < CPL; c++) {
                                const double ysc = (fb[c] >= 0.0 ? 1.0 : -1.0) * fmax(ATOL, fabs(yp[c]));
                                const double hn = (yp[c] + 10.0 * W.template ld<V_FAC>(c * WAVE + lane) * ysc) - yp[c];
                                ycur[c] = yp[c] + ((((small_bits >> c) & 1) && gs[c] == g) ? hn : 0.0);
                            }
                            phase = PH_JAC_REDO;
                        } else {
#pragma unroll
                            for (int c = 0; c < CPL; c++) {
                                jl[c] = njl[c];
                                jd[c] = njd[c];
                                ju[c] = nju[c];
                                if (vnode[c]) W.template st<V_FAC>(c * WAVE + lane, nfac[c]);
                            }
                            HC_J_STORE3();
                            j_fresh = TWO_J && HC_TWO_J_ALIAS && HC_TWO_J_FRESH;
                                jac_stage = 0;
                            if (jac_init) {
                                // rest of BDF.__init__: D[0] = y, D[1] = f0 * h_abs, order = 1
#pragma unroll
                                for (int c = 0; c < CPL; c++) {
                                    const int slot = c * WAVE + lane;
                                    W.stD(1, slot, W.ldD(1, slot) * h_abs);
                                }
                                order = 1;
                                n_equal = 0;
                                have_lu = 0;
                                phase = C_STEP_BEGIN;
                            } else {
                                have_lu = 0;
                                current_jac = 1;
                                if constexpr (TWO) {       // (C_NEWTON_BEGIN's `ycur = yp`, while yp is in registers)
#pragma unroll
                                    for (int c = 0; c < CPL; c++) ycur[c] = yp[c];
                                }
                                phase = C_NEWTON_BEGIN;
                            }
                        }
                    }
                    if (phase == C_ERR_TEST) {
                        HC_STAMP(C_ERR_TEST);
                        safety = 0.9 * (2 * NEWTON_MAXITER + 1) / (double)(2 * NEWTON_MAXITER + n_iter);
                        const double ec = error_const_k(order);
                        double e[CPL];
#pragma unroll
                        for (int c = 0; c < CPL; c++) {
                            scl[c] = fast_div(1.0, ATOL + RTOL * fabs(ycur[c]));
                            e[c] = ec * dd[c];
                        }
                        error_norm = rms_ratio<CPL>(e, scl, lane, D, inv_sqrt_d, comm);
                        if (error_norm > 1.0) {
                            const double factor = fmax(0.2, safety * exp_mid(-log_pos(error_norm) / (double)(order + 1)));
                            h_abs *= factor;
                            change_D<CPL>(W, ru, order, factor, lane);
                            n_equal = 0;
                            phase = C_STEP_TRY;
                        } else {
                            phase = C_ACCEPT;
                        }
                    }
                    if (phase == C_ACCEPT) {
                        HC_STAMP(C_ACCEPT);
                        n_equal++;
                        t = t_new;
                        nsteps++;
                        double d_ord[CPL], d_ord2[CPL];          // updated D[order], D[order+2]
                        // sol.y[:, -1] so far.  TWO layout: stored only when it can become the row's answer -- the step that
                        // reaches tf, or any step of the fifth attempt (a failed earlier attempt restarts from y0 and
                        // overwrites it): the same final bits, 5 of 6 global stores per row less
                        if (!TWO || t == tf || attempts >= 5) {
#pragma unroll
                            for (int c = 0; c < CPL; c++) W.template st<V_Y>(c * WAVE + lane, ycur[c]);
                        }
                        switch (order) {
                            case 1: HC_SUB(41); accept_update<CPL, 1, TOP2_LATER>(W, dd, lane, d_ord, d_ord2); break;
                            case 2: HC_SUB(42); accept_update<CPL, 2, TOP2_LATER>(W, dd, lane, d_ord, d_ord2); break;
                            case 3: HC_SUB(43); accept_update<CPL, 3, TOP2_LATER>(W, dd, lane, d_ord, d_ord2); break;
                            case 4: HC_SUB(44); accept_update<CPL, 4, TOP2_LATER>(W, dd, lane, d_ord, d_ord2); break;
                            default: HC_SUB(45); accept_update<CPL, 5, TOP2_LATER>(W, dd, lane, d_ord, d_ord2); break;
                        }
                        HC_SUB_END();
                        if (t == tf) {
                            phase = C_SUCCESS;
                        } else if (n_equal < order + 1) {
                            phase = C_STEP_BEGIN;
                        } else {
                            // order / step selection
                            double em[CPL], ep[CPL];
                            const double ecm = order > 1 ? error_const_k(order - 1) : 0.0;
                            const double ecp = order < MAX_ORDER ? error_const_k(order + 1) : 0.0;
#pragma unroll
                            for (int c = 0; c < CPL; c++) {
                                em[c] = ecm * d_ord[c];
                                ep[c] = ecp * d_ord2[c];
                            }
                            double nm = INFINITY, np_ = INFINITY;
                            if (order > 1) nm = rms_ratio<CPL>(em, scl, lane, D, inv_sqrt_d, comm);
                            if (order < MAX_ORDER) np_ = rms_ratio<CPL>(ep, scl, lane, D, inv_sqrt_d, comm);
                            // three pow() in three lanes at once
                            const double en = lane == 0 ? nm : (lane == 1 ? error_norm : np_);
                            // error_norms ** (-1 / (order + k)), k = 0, 1, 2: three lanes at once
                            double fk = exp_mid(-log_pos(en) / (double)(order + (lane < 3 ? lane : 2)));
                            fk = en == 0.0 ? INFINITY : (en < INFINITY ? fk : (en == INFINITY ? 0.0 : en));
                            const double f0 = readlane_d(fk, 0), f1 = readlane_d(fk, 1), f2 = readlane_d(fk, 2);
                            int best = 0;
                            double fbest = f0;
                            if (f1 > fbest) { best = 1; fbest = f1; }
                            if (f2 > fbest) { best = 2; fbest = f2; }
                            if (TOP2_LATER && best == 2) {      // the order rises: D[order + 2] of this step is read from now on
                                switch (order) {
                                    case 1: _Pragma("unroll") for (int c = 0; c < CPL; c++) W.stD(3, c * WAVE + lane, d_ord2[c]); break;
                                    case 2: _Pragma("unroll") for (int c = 0; c < CPL; c++) W.stD(4, c * WAVE + lane, d_ord2[c]); break;
                                    case 3: _Pragma("unroll") for (int c = 0; c < CPL; c++) W.stD(5, c * WAVE + lane, d_ord2[c]); break;
                                    default: _Pragma("unroll") for (int c = 0; c < CPL; c++) W.stD(6, c * WAVE + lane, d_ord2[c]); break;
                                }
                            }
                            order += best - 1;
                            const double factor = fmin(10.0, safety * fbest);
                            h_abs *= factor;
                            change_D<CPL>(W, ru, order, factor, lane);
                            n_equal = 0;
                            have_lu = 0;
                            phase = C_STEP_BEGIN;
                        }

                    }
                    if (phase == C_STEP_BEGIN) {
                        HC_STAMP(C_STEP_BEGIN);
                        // _step_impl entry
                        // nextafter(t, +inf) for the finite t >= 0 of this integrator: one integer increment of the bit
                        // pattern (the smallest subnormal from 0), not a libm call per step
                        const double t_up = t == 0.0 ? 4.9406564584124654e-324
                                                     : __longlong_as_double(__double_as_longlong(t) + 1ll);
                        min_step = 10.0 * fabs(t_up - t);
                        if (h_abs < min_step) {
                            change_D<CPL>(W, ru, order, min_step / h_abs, lane);
                            h_abs = min_step;
                            n_equal = 0;
                        }
                        current_jac = 0;
                        phase = C_STEP_TRY;
                    }
                    if (phase == C_STEP_TRY) {
                        HC_STAMP(C_STEP_TRY);
                        if (h_abs < min_step) {
                            phase = C_FAIL;
                        } else {
                            t_new = t + h_abs;
                            if (t_new - tf > 0.0) {
                                t_new = tf;
                                change_D<CPL>(W, ru, order, fabs(t_new - t) / h_abs, lane);
                                n_equal = 0;
                                have_lu = 0;
                            }
                            const double h = t_new - t;
                            h_abs = fabs(h);
                            const double inv_alpha = 1.0 / alpha_k(order);
                            switch (order) {
                                case 1: HC_SUB(49); predict<CPL, 1>(W, lane, inv_alpha, yp, psiv); break;
                                case 2: HC_SUB(50); predict<CPL, 2>(W, lane, inv_alpha, yp, psiv); break;
                                case 3: HC_SUB(51); predict<CPL, 3>(W, lane, inv_alpha, yp, psiv); break;
                                case 4: HC_SUB(52); predict<CPL, 4>(W, lane, inv_alpha, yp, psiv); break;
                                default: HC_SUB(53); predict<CPL, 5>(W, lane, inv_alpha, yp, psiv); break;
                            }
                            HC_SUB_END();
#pragma unroll
                            for (int c = 0; c < CPL; c++) scl[c] = fast_div(1.0, ATOL + RTOL * fabs(yp[c]));
                            cc = h / alpha_k(order);
                            HC_YP_STORE();
                            if constexpr (TWO) {           // (C_NEWTON_BEGIN's `ycur = yp`, while yp is in registers)
#pragma unroll
                                for (int c = 0; c < CPL; c++) ycur[c] = yp[c];
                            }
                            phase = C_NEWTON_BEGIN;
                        }
                    }
                    if (phase == C_NEWTON_BEGIN) {
                        HC_STAMP(C_NEWTON_BEGIN);
                        if (!have_lu) {
                            HC_STAMP(23);
                            if (!j_fresh) { HC_J_LOAD3(); }      // (the rows as C_JAC_FIN stored them, unless it has just done so)
                            lu_factor<CPL>(F, jl, jd, ju, cc, lane, D, comm);
                            HC_F_STORE();
                            HC_STAMP(C_NEWTON_BEGIN);
                            have_lu = 1;
                            nlu++;
                        }
#pragma unroll
                        for (int c = 0; c < CPL; c++) {
                            dd[c] = 0.0;
                            if constexpr (!TWO) ycur[c] = yp[c];
                        }
                        newton_k = 0;
                        dy_norm_old = -1.0;
                        phase = PH_NEWTON;
                    }
                    if (phase >= C_SUCCESS) {
                        ok = (phase == C_SUCCESS);
                        break;
                    }
                }
                st_nfev += nfev;
                cost_nfev += nfev;
                st_njev += njev;
                st_nlu += nlu;
                st_nsteps = nsteps;
                if (ok) break;
                // failed attempt: n_rnd *= 0.8 in place (richards_pde.py:522); restart from y0
                failed++;
#pragma unroll
                for (int c = 0; c < CPL; c++)
                    W.template st<V_NZ>(c * WAVE + lane, W.template ld<V_NZ>(c * WAVE + lane) * 0.8);
                if (!refresh) nscale *= 0.8;
                __builtin_amdgcn_wave_barrier();
                if (attempts >= 5) break;
            }
            if (A.spin_stop) {   // mean((y_j - y_{j-1})^2), simulation.py:457
                double sq = 0.0;
#pragma unroll
                for (int c = 0; c < CPL; c++) {
                    const double dlt = W.template ld<V_Y>(c * WAVE + lane) -
                                       (DEEPY ? W.template ld<V_Y0>(c * WAVE + lane) : yrow0[DEEPY ? 0 : c]);
                    sq = fma(dlt, dlt, sq);
                }
                spin_mse = comm.sum(sq) / (double)D;
            }
            failed_row = failed;
            if (failed) {
                nz_is_base = false;
                const IoArgs io = load_const(A.io);
                if (lane == 0 && comm.half == 0) atomicAdd(&io.counters[1], (unsigned long long)failed);
                if (!refresh && A.host_noise) {
#pragma unroll
                    for (int c = 0; c < CPL; c++)
                        if (vnode[c]) io.base_noise[member * D + hb + lane * CPL + c] = W.template ld<V_NZ>(c * WAVE + lane);
                }
            }
            if (refresh) fresh_seen++;
        }
        // ---- row epilogue: water table index (simulation.py:612), optional outputs
        bool unsat[CPL];
        double yv[CPL];
        // psi_sat of the member's point: one scalar load per row rather than an SGPR pair held across the integrator
        const double psi_sat_m = load_const(&A.P[point].psi_sat);
#pragma unroll
        for (int c = 0; c < CPL; c++) {
            yv[c] = W.template ld<V_Y>(c * WAVE + lane);
            unsat[c] = vnode[c] && !(yv[c] >= psi_sat_m);
        }
        int istar = deepest_true<CPL, TWO>(unsat);
        if constexpr (HALVES == 2) istar = comm.max_int(istar < 0 ? -1 : hb + istar);
        int w = istar < 0 ? 0 : istar + 1;
        w = w < D - 1 ? w : D - 1;
        if (skip) w = 0;
        if (A.spin_stop) {
            // Simulation.initial_conditions stop rule (simulation.py:452-468): water table within 2 dz of the
            // first observation and a mean squared change of the state <= 0.01; no per-row outputs
            const double abs_error = fabs(A.spin_zwtd - (A.spin_z0 + (double)w * A.spin_dz));
            const bool stop = (abs_error <= 2.0 * A.spin_dz) && (spin_mse <= 0.01);
            if (stop || r == A.n_rows - 1) {
                if (lane == 0 && comm.half == 0) load_const(A.io).spin_iters[member] = stop ? r + 1 : -(r + 1);
                break;
            }
        } else {
            const IoArgs io = load_const(A.io);
            if (lane == 0 && comm.half == 0) {
                io.wtd_u16[(size_t)r * A.n_members + member] = (unsigned short)w;
                if (io.stats) {
                    int *s = io.stats + ((size_t)r * A.n_members + member) * 6;
                    s[0] = st_nfev; s[1] = st_njev; s[2] = st_nlu; s[3] = st_nsteps; s[4] = attempts; s[5] = (int)refresh | (failed_row << 8);
                }
            }
            if (io.psi_rows) {
#pragma unroll
                for (int c = 0; c < CPL; c++)
                    if (vnode[c])
                        io.psi_rows[((size_t)r * A.n_members + member) * D + hb + lane * CPL + c] = skip ? 0.0 : yv[c];
            }
            if (io.diag && lane == 0 && comm.half == 0) {
                io.diag[((size_t)r * A.n_members + member) * 2 + 0] = skip ? 0.0 : diag_tr;
                io.diag[((size_t)r * A.n_members + member) * 2 + 1] = skip ? 0.0 : diag_lf;
            }
        }
    }
    {
        const IoArgs io = load_const(A.io);
#pragma unroll
        for (int c = 0; c < CPL; c++)
            if (vnode[c]) io.psi[member * D + hb + lane * CPL + c] = W.template ld<V_Y>(c * WAVE + lane);
        if (!A.host_noise && lane == 0 && comm.half == 0) io.nscale[member] = nscale;
        if (multi && lane == 0 && comm.half == 0) atomicAdd(&io.point_cost[point], (unsigned long long)cost_nfev);
#ifdef HC_PROFILE
        __builtin_amdgcn_wave_barrier();
        // split column: HC_PROFILE_HALF (default 0) says which half of the pairs reports -- the upper half is the critical path
#ifndef HC_PROFILE_HALF
#define HC_PROFILE_HALF 0
#endif
        if (HALVES == 1 || comm.half == HC_PROFILE_HALF) {
            if (lane < 32) atomicAdd(&io.counters[8 + lane], prof_lds[lane]);
            if (lane >= 32) atomicAdd(&io.counters[32 + lane], prof_lds[lane]);     // entries: counters[64..95]
            if (lane < 32) atomicAdd(&io.counters[96 + lane], prof_lds[64 + lane]);  // sub-region entries: counters[96..127]
        }
        prof_lds[lane] = 0;
        if (lane < 32) prof_lds[64 + lane] = 0;
#endif
    }
    }   // next member
}

}  // namespace hc
