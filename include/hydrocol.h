/*
 * hydrocol.h -- C-ABI of libhydrocol.so: MI355X (gfx950) ensemble stepper for the
 * HydroModel 1-D stochastic Richards soil column.
 *
 * The reference (vrettasm/HydroModel) is pure Python and has no FFI; the boundary it
 * exposes for this path is two Python call signatures.  Each entry point below names
 * the reference interface it replaces (paths relative to /root/reference/code):
 *
 *   hc_set_column      <- objects built by Simulation.setupModel (src/simulation.py:100-231):
 *                         Porosity / TreeRoots / VrettasFung|vanGenuchten / SoilProperties ...
 *                         flattened to per-depth tables by hydromodel_amd/digest.py
 *   hc_set_forcing     <- per-row args_i dict (src/simulation.py:591-602): precipitation, atm,
 *                         time(hour), wtd; refresh rule `precip > 0.5 or i % 48 == 0` (:599)
 *   hc_set_state/get   <- y0 / y_i vectors (src/simulation.py:514,609,626)
 *   hc_set_noise_*     <- n_rnd vectors drawn at src/simulation.py:426,561,601
 *   hc_step_rows       <- the row loop body: RichardsPDE.solve(t_span, y0, args_i)
 *                         (src/richards_pde.py:478-537 -> scipy solve_ivp BDF) followed by
 *                         find_wtd(y_i >= psi_sat) (src/simulation.py:609-612)
 *   hc_spinup          <- Simulation.initial_conditions (src/simulation.py:389-493), per member
 *   hc_rhs             <- RichardsPDE.__call__(t, y, args)   (src/richards_pde.py:82-160)
 *   hc_model_nodes     <- h_model(y_i, z, args_i) diagnostics call (src/simulation.py:623;
 *                         src/models/vrettas_fung.py:51 / vanGenuchten.py:23)
 *   hc_get_moments     <- (new) per-row ensemble moments of wtd_est (src/simulation.py:612)
 *   hc_add_point       <- the per-parameter-point objects of Simulation.setupModel (src/simulation.py:146-231:
 *                         SoilProperties / WaterContent / HydraulicConductivity / Porosity -> field capacity,
 *                         wilting point src/porosity.py:172-181, iPsi_50 src/simulation.py:336-339), one set per
 *                         point of a parameter sweep, all points stepped by ONE launch
 *   hc_plugin_eval     <- HydrologicalModel subclasses called directly: VrettasFung.__call__(psi, z, {"n_rnd": ..})
 *                         (src/models/vrettas_fung.py:51-257), vanGenuchten.__call__ (src/models/vanGenuchten.py:23-126)
 *   hc_get/set_noise_scale <- the in-place damping `args["n_rnd"] *= 0.8` of a failed attempt (src/richards_pde.py:522) as
 *                         it accumulates on a member's base vector; with hc_get_state / hc_get_moments the restart state
 *                         of an ensemble (the reference restarts one column from IC_Filename, src/simulation.py:358-385)
 *   hc_set_point_member_bases <- which realisation a member is: the reference seeds one generator per run
 *                         (src/simulation.py:66-70); here member j of sweep point k draws stream base[k] + j
 *   hc_allreduce_moments, hc_get_point_costs <- (new) the ensemble's one collective; per-point cost for scheduling
 *
 * Conventions: every function returns 0 on success or a negative hc_status; nothing throws
 * or aborts across the boundary; hc_last_error() gives the thread-local message.  Host
 * pointers are only read/written during the call.  The library owns all device memory.
 * State layout in HBM is member-major: psi[member][depth], fp64.  One handle drives one
 * device; a handle is not thread-safe; distinct handles are independent.
 */
#ifndef HYDROCOL_H
#define HYDROCOL_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct hc_handle hc_handle;

enum hc_status {
    HC_OK = 0,
    HC_ERR_ARG = -1,       /* bad argument / call order            */
    HC_ERR_DEVICE = -2,    /* HIP runtime error                    */
    HC_ERR_NO_DEVICE = -3, /* no gfx950 device visible             */
    HC_ERR_UNSUPPORTED = -4
};

#define HC_MODEL_VRETTAS_FUNG 0
#define HC_MODEL_VAN_GENUCHTEN 1
#define HC_MAX_DEPTH_NODES 640 /* 64 lanes x 10 cells; columns of 577..640 nodes whose parameter points all keep the root
                                  zone above node 320 run on two cooperating wavefronts per member (64 x 5 cells each),
                                  single points and sweeps alike; HYDROCOL_SPLIT_COLUMN in the environment at hc_create:
                                  0 keeps the one-wave kernels, 1 takes the two-wavefront kernel from 513 nodes on */

/* One soil column geometry + parameter point (static during a run). */
typedef struct {
    int32_t dim_d;        /* D depth nodes (z_grid.size)                              */
    int32_t model;        /* HC_MODEL_*                                               */
    int32_t flag_et;      /* Simulation_Flags.ET                                      */
    int32_t flag_lf;      /* Simulation_Flags.LF  (monitoring mode)                   */
    int32_t flag_hlift;   /* Simulation_Flags.HLIFT                                   */
    int32_t n_root_first; /* root-zone cells of the first-midpoint call (0 or 1)      */
    int32_t n_root_int;   /* root-zone cells of the interior call                     */
    int32_t n_groups;     /* FD-Jacobian column groups                                */
    double theta_res, alpha, n, m, psi_sat, epsilon, lambda_exp, sigma_noise, sat_soil, dz;
    double ipsi50, lai, surface_evap, interception, evap_delta_min;
    /* Repaired PREDICT mode (src/richards_pde.py:312-351) -- an EXTENSION with no reference oracle: the reference
     * raises TypeError at :327-330 (np.linspace with a float count).  Semantics here: low_lim = dim_d - (sat_cells - 1)
     * of each pde_fun call as an integer, no cell drains when it is <= 0; needs the wet-season bit in `daylight`. */
    int32_t flag_predict; /* Simulation_Flags.PREDICT (lateral flow in predictive mode, needs flag_lf)   */
    int32_t sat_cells;    /* ceil(sat_depth / dz), src/simulation.py:128                                  */
} hc_column_params;

/* node_tabs: [3][D] rows = porosity, mean-K, noise coefficient (-1 = no layer) at the nodes
 * mid_tabs : [6][D-1] rows = porosity, field capacity, wilting point, root pdf, mean-K,
 *            noise coefficient at the midpoints
 * groups   : [D] column group of every state entry (scipy group_columns for a tridiagonal) */
int hc_create(int device_ordinal, hc_handle **out);
int hc_destroy(hc_handle *h);
const char *hc_last_error(void);
const char *hc_version(void);   /* "hydrocol <v> (gfx950) kernels <hash>": <hash> identifies the device code (kernel
                                      sources + compile flags + compiler) -- the key of profiles/pmc_constants.json */

int hc_set_column(hc_handle *h, const hc_column_params *p, const double *node_tabs,
                  const double *mid_tabs, const int32_t *groups);
/* Parameter points (BASELINE config 5).  hc_set_column installs point 0 and the geometry (dim_d, groups, dz) and
 * drops any further points; hc_add_point appends one more point with tables of its own (same dim_d, n_groups, dz).
 * With P points the n_members of hc_set_members must be a multiple of P and are point-major: point k owns members
 * [k * n_members / P, (k + 1) * n_members / P).  One hc_step_rows / hc_spinup launch advances every point; each
 * member sees the column parameters and tables of its own point; moments are kept per point.  Adding a point changes
 * the shape of the moment table ([n_points][3][T]): whatever was accumulated before is dropped (the table is re-created,
 * zeroed, by the next call that needs it) -- install all points before the first hc_step_rows. */
int hc_add_point(hc_handle *h, const hc_column_params *p, const double *node_tabs, const double *mid_tabs);
int hc_get_point_count(hc_handle *h); /* >= 1 once hc_set_column has run; negative on error */
/* The cell model comes in two builds: one specialised for the reference's default exponents (vrettas_fung, n = 2,
 * m = 1/2, lambda = 1: every pow() folds into rsqrt / multiplies) and a generic one (powers as exp(y log x)); a handle
 * takes the first only when EVERY point qualifies.  The two agree to ~1e-15 relative per call, not bit for bit.
 * on != 0 pins the generic build, so that a default-exponent point gives the same bits whether it is stepped alone or
 * inside a sweep of other points. */
int hc_set_generic_exponents(hc_handle *h, int32_t on);

/* Forcing struct-of-arrays, n_rows entries each; wtd_obs < 0 marks a row to skip
 * (src/simulation.py:582-588): such a row is not solved and consumes NO noise draw (its refresh flag is ignored,
 * host-noise callers must not supply a vector for it).  The library numbers the draws itself: refresh row k
 * (1-based, skipped rows not counted) uses draw k; 0 = base vector.
 * daylight: bit 0 = daylight (6 <= hour <= 17, src/richards_pde.py:230); bit 1 = wet season, month in
 * {10,11,12,1,2,3} (src/richards_pde.py:315) -- read only in PREDICT mode. */
int hc_set_forcing(hc_handle *h, int64_t n_rows, const double *precip, const double *atm,
                   const uint8_t *daylight, const int32_t *wtd_obs, const uint8_t *refresh);

/* One row of the forcing replaced in place (row < n_rows of hc_set_forcing; its refresh flag is cleared): what the
 * row-by-row caller of src/simulation.py:590-609 knows only when it reaches the row -- args_i["wtd"], ["atm"],
 * ["time"], ["precipitation"] -- handed over just before `pde_model.solve(t_span, y0, args_i)` (hydromodel_amd/pde.py). */
int hc_set_forcing_row(hc_handle *h, int64_t row, double precip, double atm, uint8_t daylight, int32_t wtd_obs);

int hc_set_members(hc_handle *h, int64_t n_members);
/* psi: [n_members][D] (broadcast 0), [D] copied to every member (1), or [n_points][D] copied to the members of
 * each parameter point (2) */
int hc_set_state(hc_handle *h, const double *psi, int broadcast);
int hc_get_state(hc_handle *h, double *psi, int64_t first_member, int64_t count);

/* Noise source A (parity): host-supplied base vectors [n_members][D]; refresh-row vectors are
 * passed to hc_step_rows.  Mutated in place by the x0.8 retry rule (src/richards_pde.py:522). */
int hc_set_noise_host(hc_handle *h, const double *base);
int hc_get_noise_base(hc_handle *h, double *base, int64_t first_member, int64_t count);
/* Noise source B (throughput): counter-based Philox4x32-10 + Box-Muller generated in-kernel;
 * stream = (seed, member_offset + member, draw index, depth).  The retry damping is kept as a
 * per-member scale factor. */
int hc_set_noise_philox(hc_handle *h, uint64_t seed, int64_t member_offset);
/* draw indices: 0 = base vector, k >= 1 = k-th refresh row, HC_PHILOX_DRAW_SPINUP = the vector spin-up solves use
 * (the reference draws spin-up, base, refresh... in that order: src/simulation.py:426,561,601) */
#define HC_PHILOX_DRAW_SPINUP 0xFFFFFFFFll
/* What the Philox source yields for (member, draw): out[D] (test hook). */
int hc_philox_normals(hc_handle *h, int64_t member, int64_t draw, double *out);

typedef struct {
    int64_t row_begin;      /* first row to solve (row i integrates t in [i-1, i]); >= 1      */
    int64_t n_rows;
    int32_t spinup;         /* 1: SPINUP semantics (src/simulation.py:398): row_begin is used as
                               the forcing row for every solve, t_span = (0,1), noise never refreshed */
    int32_t accumulate_moments; /* add (count, sum idx, sum idx^2) of wtd_est per row          */
    const double *fresh_noise;  /* host-noise mode: [n_refresh_rows_in_range][n_members][D]    */
    int32_t *wtd_out;       /* nullable: [n_rows][n_members] wtd_est index per row             */
    int32_t *stats_out;     /* nullable: [n_rows][n_members][6] nfev,njev,nlu,nsteps,attempts, then
                               (refresh flag) | (failed attempts of the row << 8): each failed attempt scaled the
                               row's noise vector by 0.8 (src/richards_pde.py:522)                      */
    double *psi_rows_out;   /* nullable: [n_rows][n_members][D] state after every row          */
    double *diag_out;       /* nullable: [n_rows][n_members][2] = transpiration, lateral_flow as
                               pde_model.arg_out holds them after the row's solve
                               (src/richards_pde.py:380-391, src/simulation.py:629-630)        */
    double kernel_ms;       /* out: device time of the launch(es), HIP events on the stream    */
    int64_t launches;       /* out */
} hc_step_args;

int hc_step_rows(hc_handle *h, hc_step_args *a);

/* Simulation.initial_conditions (src/simulation.py:389-493) for every member in ONE launch: each member
 * repeats the solve of `forcing_row` over t in (0, 1) with SPINUP semantics and its own fixed noise vector,
 * starting from the state set by hc_set_state, until its own stop rule holds (src/simulation.py:468:
 * |zwtd_cm - z[wtd_est]| <= 2 dz and mean((y_j - y_{j-1})^2) <= 0.01) or max_iterations solves are done.
 * The members' states are left in place (hc_get_state). */
typedef struct {
    int64_t forcing_row;      /* 0 in the reference                                                */
    int32_t max_iterations;   /* burn_in = 1500 in the reference (src/simulation.py:420)           */
    double zwtd_cm;           /* first water-table observation, cm                                 */
    double z0_cm;             /* depth of node 0: z[i] = z0_cm + i * dz                            */
    int32_t *iterations_out;  /* [n_members] solves used; negative = stopped by max_iterations     */
    double kernel_ms;         /* out */
} hc_spinup_args;
int hc_spinup(hc_handle *h, hc_spinup_args *a);
int hc_synchronize(hc_handle *h);
/* event counters since hc_create: [0] FD-Jacobian passes that took num_jac's "difference too small ->
 * retry with a 10x step" branch, [1] failed BDF attempts (each scales the noise by 0.8), [2] attempts abandoned
 * by the kernel's iteration budget (default 20 000 trips of the phase loop, ~14x the costliest regular attempt seen;
 * a semantic deviation -- SciPy's BDF has no such cap -- handled like a solve that gave up, DESIGN.md "Iteration
 * budget"; with HYDROCOL_STRICT_GUARD=1 in the environment hc_step_rows / hc_spinup fail instead), [3] where the
 * last of those happened: global member id << 24 | forcing row.
 * Test hooks read from the environment at hc_create: HYDROCOL_DEBUG_MAX_ITER (same as hc_set_iteration_budget),
 * HYDROCOL_DEBUG_JAC_REJECT (raises num_jac's retry threshold), HYDROCOL_ROWS_PER_LAUNCH, HYDROCOL_CHUNK_MEMBERS
 * (members per scheduling chunk when several parameter points share a launch), HYDROCOL_DEBUG_CUS (a persistent grid of
 * fewer workgroups than the device has compute units: measurements only). */
int hc_get_counters(hc_handle *h, uint64_t *out4);
/* Rows one kernel launch of hc_step_rows covers; hc_step_rows splits longer requests.  Default (and rows = 0): with the
 * in-kernel noise 48 = one simulated day for ensembles of >= 1 048 576 members, proportionally more for smaller ones
 * (48 x 1 048 576 / members, at most a year; round 5: 65 536 before -- 4 096 members +5 %, 16 384 +2 %, 65 536 +3 %); with
 * the caller's noise (every refreshed row of a launch stages members x D doubles) 48 x 65 536 / members.
 * A member's rows of a launch are solved back to back by one wavefront with psi resident in LDS, and a launch ends when
 * its slowest wavefront does.  Large ensembles (>> 1 024 wavefronts' worth of members) balance within a day; a SMALL
 * ensemble (a few members per wavefront, e.g. 4 096) loses ~15 % to that tail per launch and is better served by long
 * launches (a year: the members' total costs are nearly equal).  Per-row device buffers grow with it
 * (2 B x rows x members for the water-table indices). Results do not depend on it. */
int hc_set_rows_per_launch(hc_handle *h, int32_t rows);
/* Budget of one BDF attempt in trips of the kernel's phase loop (>= 1). */
int hc_set_iteration_budget(hc_handle *h, int32_t phase_steps);
/* The integrator is SciPy's (third-party to the reference; /root/reference/requirements.txt:4 pins scipy==1.5.2, the pinning
 * vectors were made with 1.15.3).  On this path the two differ in ONE place: common.py select_initial_step clamps h0 and
 * the step it returns to the integration interval since scipy 1.9.  Default (0): the >= 1.9 form; on != 0: the 1.5.2 form
 * (also HYDROCOL_SCIPY_152=1 in the environment at hc_create).  How often the clamps bind and what that moves:
 * profiles/r05_scipy152_clamp.txt. */
int hc_set_scipy_152(hc_handle *h, int32_t on);

/* moments: [n_points][3][n_forcing_rows] int64 = count, sum(idx), sum(idx^2) of wtd_est over the members of each
 * parameter point ([3][n_forcing_rows] for the usual single point); not available for spin-up solves */
int hc_get_moments(hc_handle *h, int64_t *moments);
/* The same table copied device-to-device into caller-owned memory on the handle's device (n_points * 3 * n_rows
 * int64): the buffer a collective (RCCL all-reduce over the ranks' tables) works on, without a host round trip.
 * Complete when the call returns. */
int hc_export_moments(hc_handle *h, void *device_dst);
int hc_set_moments(hc_handle *h, const int64_t *moments);
int hc_reset_moments(hc_handle *h);

/* The path's one collective inside the library, for a single process that drives several devices with
 * one handle each: every handle's moment table is replaced by the sum over all n handles (ncclAllReduce, ncclInt64,
 * ncclSum over RCCL / xGMI, in place on device memory, on the handles' own streams).  Integer sums: the result does not
 * depend on the number of devices.  RCCL is bound at run time (dlopen); HC_ERR_UNSUPPORTED when it is not loadable.
 * One process per GPU (torch.distributed) callers use hc_export_moments + their own all-reduce instead. */
int hc_allreduce_moments(hc_handle **handles, int n);

/* Bit-exact resume of a Philox ensemble (the reference's single-column analogue is IC_Filename,
 * src/simulation.py:358-385): besides the state (hc_get_state / hc_set_state), the moment table (hc_get_moments /
 * hc_set_moments) and the next row, a stopped ensemble is defined by the damping every member's base noise vector has
 * collected from failed attempts, scale[member] = 0.8^k (src/richards_pde.py:522 applied to the base vector).
 * hc_set_noise_scale must follow hc_set_noise_philox (which resets the scales to 1); values must lie in (0, 1]. */
int hc_get_noise_scale(hc_handle *h, double *scale, int64_t first_member, int64_t count);
int hc_set_noise_scale(hc_handle *h, const double *scale, int64_t first_member, int64_t count);

/* Several parameter points in one handle: base[k] = global id of point k's first member, i.e. the Philox stream of
 * member j of point k is keyed by base[k] + j.  Default (and base = NULL): member_offset + k * members_per_point, the
 * points of this handle are consecutive points of the sweep.  A rank that is dealt non-consecutive points (round-robin
 * by cost) sets the bases so that a point's realisations do not depend on who runs it. */
int hc_set_point_member_bases(hc_handle *h, const int64_t *base);
/* cost[n_points]: RHS evaluations spent on each point's members by hc_step_rows since the points were installed (zeros
 * for a single point).  The library walks the points costliest-first from the second launch on (the order changes
 * no result; HYDROCOL_POINT_ORDER=fixed in the environment at hc_create keeps point order). */
int hc_get_point_costs(hc_handle *h, uint64_t *cost);

/* Test hooks -------------------------------------------------------------------------- */
/* dydt for every member's current state on forcing row `row` (noise = base vectors).
 * aux (nullable): [n_members][3*(D-1)+1] = c | s | f at the midpoints, then pL.  Columns the split-column kernel serves
 * are evaluated on that path (two wavefronts per member) when aux is NULL, on the one-wave path when it is requested. */
int hc_rhs(hc_handle *h, int64_t row, int32_t spinup, double *dydt, double *aux);
/* plugin call on the nodes for every member's current state: out [4][n_members][D]
 * = theta, K, C, K_bkg; qinf [n_members] (nullable) */
int hc_model_nodes(hc_handle *h, double *out, double *qinf);

/* Stateless plugin call (needs a device, no handle): psi [n_cells][n_cols] depth-major as the reference's
 * [dim_d x dim_m]; por / meank / noisec / n_rnd [n_cells] = porosity, layer-mean K (0 -> 1e-7, src/utilities.py:50),
 * noise coefficient (0.05 / 0.10 / 1.0, -1 = cell in no layer) and N(0,1) value of every cell.
 * out [4][n_cells][n_cols] = theta, K, C, K_bkg; qinf [n_cols] = q_inf_max from row 0 (src/models/vrettas_fung.py:254).
 * Only model, theta_res, alpha, n, m, psi_sat, epsilon, lambda_exp, sigma_noise, sat_soil and dz of `p` are read. */
int hc_plugin_eval(int device_ordinal, const hc_column_params *p, int64_t n_cells, int64_t n_cols,
                   const double *psi, const double *por, const double *meank, const double *noisec,
                   const double *n_rnd, double *out, double *qinf);

#ifdef __cplusplus
}
#endif
#endif
