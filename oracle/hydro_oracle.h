/*
 * hydro_oracle.h -- CPU restatement (plain C, scalar) of the HydroModel column stepper.
 *
 * TEST INFRASTRUCTURE ONLY.  This is the parity oracle for the HIP path in
 * hydromodel_amd/csrc/.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it; the product (hydromodel_amd) never does.
 *
 * Parity status: PINNED against outputs of the reference itself (golden vectors
 * G1-G6 under tests/golden/, produced by tests/golden/make_golden.py importing
 * /root/reference in the build container).  The reference's own tests pin only
 * find_wtd known answers (code/tests/test_utilities.py:56-86), also checked.
 *
 * The time integrator is third-party to the reference: scipy==1.5.2 pinned in
 * /root/reference/requirements.txt:4 (1.15.3 installed here, used for the vectors);
 * scipy.integrate.solve_ivp(method='BDF') is restated from its published algorithm
 * (Shampine & Reichelt 1997 NDF; Byrne & Hindmarsh 1975) following the call site
 * code/src/richards_pde.py:512-514.
 */
#ifndef HYDRO_ORACLE_H
#define HYDRO_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HO_MODEL_VRETTAS_FUNG 0
#define HO_MODEL_VAN_GENUCHTEN 1

/* which cells a plugin call covers (decides table set and the LOCAL noise index) */
#define HO_VIEW_NODES 0   /* all D nodes, noise index i        (simulation.py:623)          */
#define HO_VIEW_TOP 1     /* node 0 alone, noise index 0        (richards_pde.py:435)        */
#define HO_VIEW_FIRST 2   /* midpoint 0 alone, noise index 0    (richards_pde.py:100-103)    */
#define HO_VIEW_INTERIOR 3/* midpoints 1..D-2, noise index j-1  (richards_pde.py:123-126)    */

typedef struct {
    int32_t dim_d, model, flag_et, flag_lf, flag_hlift, n_root_first, n_root_int, n_groups;
    double theta_res, alpha, n, m, psi_sat, epsilon, lambda, sigma_noise, sat_soil, dz,
           ipsi50, lai, surface_evap, interception, evap_delta_min;
    const double *por_node, *meank_node, *noisec_node;                          /* [D]   */
    const double *por_mid, *fc_mid, *wlt_mid, *root_mid, *meank_mid, *noisec_mid; /* [D-1] */
    const int32_t *groups;                                                      /* [D]   */
    /* repaired PREDICT mode (richards_pde.py:312-351 with `low_lim` cast to int, clamped at 0): an extension,
     * the reference itself raises TypeError at :327-330 -- nothing pins this branch but its own formula */
    int32_t flag_predict, sat_cells;
} ho_column;

typedef struct {
    double precip, atm;
    int32_t daylight, wtd_obs, spinup;
    int32_t wet;   /* PREDICT mode: month in {10,11,12,1,2,3} (richards_pde.py:315) */
} ho_row;

typedef struct {
    int64_t nfev, njev, nlu, nsteps, attempts, success;
} ho_stats;

/* plugin call: out arrays sized for the view (D, 1, 1, D-2); qinf may be NULL */
void ho_model_eval(const ho_column *c, int view, const double *psi, const double *n_rnd,
                   double *q, double *K, double *C, double *kbkg, double *qinf);

/* inverse van Genuchten, hydrological_model.py:43-119 */
void ho_pressure_head(const ho_column *c, const double *theta, double *psi, double *s_eff);

double ho_logn_rnd(double mx, double vx, double en);   /* utilities.py:4-54 */
int ho_find_wtd(const uint8_t *sat, int n);            /* utilities.py:56-99 */

/* method-of-lines RHS, richards_pde.py:82-160; aux (nullable) receives
 * [c(D-1) | s(D-1) | f(D-1) | pL | transp_first | lf_first | transp_int | lf_int] */
void ho_rhs(const ho_column *c, const ho_row *r, const double *y, const double *n_rnd,
            double *dydt, double *aux);

/* one row: richards_pde.py:478-537 (<=5 BDF attempts, noise *= 0.8 in place per failure).
 * t_steps (nullable, capacity cap_steps) receives the accepted time points of the LAST attempt. */
void ho_solve_row(const ho_column *c, const ho_row *r, double t0, double t1, const double *y0,
                  double *n_rnd, double *y1, ho_stats *st, double *t_steps, int cap_steps);

/* row loop of Simulation.run (simulation.py:561-626) for ONE member.
 * rows: [T] (daylight: bit 0 = daylight, bit 1 = wet season); base_noise [D] (mutated); fresh [n_refresh][D]
 * consumed in row order (mutated);
 * outputs (nullable): wtd_est [T] (index), psi_out [T][D], per_row [T][6] = nfev,njev,nlu,nsteps,attempts,refresh */
void ho_run(const ho_column *c, int64_t T, const double *precip, const double *atm,
            const uint8_t *daylight, const int32_t *wtd_obs, const uint8_t *refresh,
            int64_t row_begin, int64_t row_end, double *psi /*[D] in/out*/,
            double *base_noise, double *fresh, int32_t *wtd_est, double *psi_out, int32_t *per_row);

/* spin-up, simulation.py:389-493: returns iterations used; psi [D] in (hydrostatic start) / out */
int ho_spinup(const ho_column *c, const ho_row *row0, double zwtd0_cm, const double *z,
              double *psi, double *n_rnd, int max_iter);

/* pde_model.arg_out as Simulation.run reads it after a solve: {transpiration, lateral_flow} of the interior
 * pde_fun call of the LAST RHS evaluation (richards_pde.py:380-391, simulation.py:629-630) */
void ho_last_arg_out(double *out2);
void ho_run_diag(const ho_column *c, int64_t T, const double *precip, const double *atm,
                 const uint8_t *daylight, const int32_t *wtd_obs, const uint8_t *refresh,
                 int64_t row_begin, int64_t row_end, double *psi, double *base_noise, double *fresh,
                 int32_t *wtd_est, double *diag);

/* scipy==1.5.2 (the reference's pin, requirements.txt:4) has no clamp of the initial step to the interval; on != 0 runs that
 * form.  ho_clamp_counts: [0] solves whose h0 exceeded the interval, [1] solves whose min(100 h0, h1) did, [2] all solves. */
void ho_set_scipy_152(int on);
void ho_clamp_counts(long *out3, int reset);

/* test hooks (num_jac retry branch) */
void ho_debug_set_jac_reject(double v);
long ho_debug_jac_retry_count(void);

#ifdef __cplusplus
}
#endif
#endif
