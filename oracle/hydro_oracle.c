/*
 * hydro_oracle.c -- see hydro_oracle.h.  TEST INFRASTRUCTURE, not product code.
 *
 * Every function cites the reference lines it restates ("ref:" = /root/reference/code/src,
 * "scipy:" = scipy/integrate/_ivp as installed, algorithm of the pinned scipy==1.5.2).
 * Plain scalar C on purpose: it is meant to be read next to the Python.
 */
#include "hydro_oracle.h"

static int g_scipy_152 = 0;              /* 1: select_initial_step as scipy 1.5.2 has it (no clamp to the interval) */
static __thread long g_clamp_counts[3] = {0, 0, 0};   /* per calling thread */

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define HO_MAXD 1024
#define EPS 2.220446049250313e-16

/* ---- NumPy semantics helpers ------------------------------------------------------- */
static inline double np_maximum(double a, double b)
{
    if (a != a) return a;
    if (b != b) return b;
    return a > b ? a : b;
}
static inline double np_minimum(double a, double b)
{
    if (a != a) return a;
    if (b != b) return b;
    return a < b ? a : b;
}
/* ndarray ** python-float: NumPy's scalar-exponent fast paths, else pow() */
static inline double np_power(double x, double p)
{
    if (p == 2.0) return x * x;
    if (p == 1.0) return x;
    if (p == 0.5) return sqrt(x);
    if (p == -1.0) return 1.0 / x;
    return pow(x, p);
}
/* np.sum over a contiguous 1-D array: pairwise summation (numpy/_core/src/umath/loops_utils) */
static double np_sum(const double *a, long n)
{
    if (n < 8) {
        double res = 0.0;
        for (long i = 0; i < n; i++) res += a[i];
        return res;
    } else if (n <= 128) {
        double r[8];
        long i;
        for (int k = 0; k < 8; k++) r[k] = a[k];
        for (i = 8; i < n - (n % 8); i += 8)
            for (int k = 0; k < 8; k++) r[k] += a[i + k];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) res += a[i];
        return res;
    } else {
        long n2 = n / 2;
        n2 -= n2 % 8;
        return np_sum(a, n2) + np_sum(a + n2, n - n2);
    }
}
/* scipy: common.norm = ||x||_2 / sqrt(n) */
static double rms_norm(const double *x, int n)
{
    double s = 0.0;
    for (int i = 0; i < n; i++) s += x[i] * x[i];
    return sqrt(s) / sqrt((double)n);
}

/* ---- utilities.py ------------------------------------------------------------------ */
double ho_logn_rnd(double mx, double vx, double en)
{
    /* ref: utilities.py:50 (zero mean guard), :10-19 (_local_fast) */
    if (mx == 0.0) mx = 1.0e-7;
    double mx_sq = mx * mx;
    double mu0 = log(mx_sq / sqrt(vx + mx_sq));
    double sig = sqrt(log(vx / mx_sq + 1.0));
    return exp(mu0 + sig * en);
}

int ho_find_wtd(const uint8_t *sat, int n)
{
    /* ref: utilities.py:83-98 -- scan upward from the bottom for the first unsaturated cell */
    int i = 0;
    for (int j = 0; j < n; j++) {
        if (!sat[n - 1 - j]) {
            i = n - j;
            break;
        }
    }
    return i < n - 1 ? i : n - 1;
}

/* ---- the plugin: one cell ------------------------------------------------------------ */
static void model_cell(const ho_column *c, double psi, double por, double meank, double noisec,
                       double nrnd, double *q_, double *K_, double *C_, double *kb_)
{
    /* ref: vrettas_fung.py:108-126 / vanGenuchten.py:71-89 (identical water retention part) */
    double delta_s = por - c->theta_res;
    int sat = psi >= c->psi_sat;
    double ap = c->alpha * fabs(psi);
    double q = c->theta_res + delta_s * np_power(1.0 + np_power(ap, c->n), -c->m);
    if (sat) q = por;
    double s_eff = (q - c->theta_res) / delta_s;
    s_eff = np_minimum(np_maximum(s_eff, 0.0), 1.0);
    double K, kb;
    if (c->model == HO_MODEL_VRETTAS_FUNG) {
        /* ref: vrettas_fung.py:140-236 */
        K = c->sat_soil * pow(s_eff, c->lambda);
        kb = c->sat_soil;
        if (noisec >= 0.0) {
            double rnd = noisec * nrnd;
            double var = c->sigma_noise * (1.0 - s_eff);
            kb = ho_logn_rnd(meank, var, rnd);
            K = pow(s_eff, c->lambda) * kb;
        }
        if (sat) K = kb;
        K = np_minimum(K, kb);
    } else {
        /* ref: vanGenuchten.py:91-104 */
        double mth = np_power(s_eff, 1.0 / c->m);
        kb = c->sat_soil;
        K = kb * sqrt(s_eff) * np_power(1.0 - np_power(1.0 - mth, c->m), c->n);
        if (sat) K = kb;
        K = np_minimum(K, kb);
    }
    /* ref: vrettas_fung.py:239-249 */
    double C = (c->m * c->n) * c->alpha * delta_s * np_power(s_eff, 1.0 / c->m + 1.0) *
               np_power(ap, c->n - 1.0);
    if (sat) C = c->epsilon;
    if (C < c->epsilon || !isfinite(C)) C = c->epsilon;
    *q_ = q;
    *K_ = K;
    *C_ = C;
    *kb_ = kb;
}

void ho_model_eval(const ho_column *c, int view, const double *psi, const double *n_rnd,
                   double *q, double *K, double *C, double *kbkg, double *qinf)
{
    int D = c->dim_d;
    if (view == HO_VIEW_NODES || view == HO_VIEW_TOP) {
        int k = view == HO_VIEW_TOP ? 1 : D;
        for (int i = 0; i < k; i++)
            model_cell(c, psi[i], c->por_node[i], c->meank_node[i], c->noisec_node[i], n_rnd[i],
                       &q[i], &K[i], &C[i], &kbkg[i]);
        /* ref: vrettas_fung.py:254 */
        if (qinf) *qinf = np_minimum(2.0 * (c->por_node[0] - q[0]) * c->dz, kbkg[0]);
    } else if (view == HO_VIEW_FIRST) {
        model_cell(c, psi[0], c->por_mid[0], c->meank_mid[0], c->noisec_mid[0], n_rnd[0],
                   &q[0], &K[0], &C[0], &kbkg[0]);
        if (qinf) *qinf = np_minimum(2.0 * (c->por_mid[0] - q[0]) * c->dz, kbkg[0]);
    } else {
        for (int p = 0; p < D - 2; p++) /* position p <-> grid midpoint p+1, noise n_rnd[p] */
            model_cell(c, psi[p], c->por_mid[p + 1], c->meank_mid[p + 1], c->noisec_mid[p + 1],
                       n_rnd[p], &q[p], &K[p], &C[p], &kbkg[p]);
        if (qinf) *qinf = np_minimum(2.0 * (c->por_mid[1] - q[0]) * c->dz, kbkg[0]);
    }
}

void ho_pressure_head(const ho_column *c, const double *theta, double *psi, double *s_eff_out)
{
    /* ref: hydrological_model.py:86-118.  porous(z_grid) interpolates AT the knots -> por_node. */
    int D = c->dim_d, nsat = 0;
    for (int i = 0; i < D; i++) {
        double por = c->por_node[i];
        double delta_s = por - c->theta_res;
        double q = np_minimum(np_maximum(theta[i], c->theta_res), por);
        double s = (q - c->theta_res) / delta_s;
        s = np_minimum(np_maximum(s, c->epsilon), 1.0);
        s_eff_out[i] = s;
        if (s >= 0.99998) {
            psi[i] = (double)nsat * c->dz; /* np.arange(0, n_sat) * dz, in order of appearance */
            nsat++;
        } else {
            psi[i] = -np_power(pow(s, -1.0 / c->m) - 1.0, 1.0 / c->n) / c->alpha;
        }
        if (!isfinite(psi[i])) psi[i] = -1.0e+5;
    }
}

/* ---- TreeRoots.efficiency, ref: tree_roots.py:213-291 ------------------------------- */
static double root_efficiency(const ho_column *c, int off, int nr, const double *theta, double *rho)
{
    const double *por = c->por_mid + off, *fc = c->fc_mid + off, *wlt = c->wlt_mid + off;
    double tmp[HO_MAXD];
    for (int i = 0; i < nr; i++) tmp[i] = theta[i] - wlt[i];
    double water_k = np_sum(tmp, nr) * c->dz;
    if (!(water_k > 0.0)) {
        for (int i = 0; i < nr; i++) rho[i] = 0.0;
        return 0.0;
    }
    double local[HO_MAXD], run = 0.0;
    for (int i = 0; i < nr; i++) {
        run += theta[i];
        local[i] = run * c->dz;
    }
    double total = local[nr - 1];
    if (total == 0.0) total = 1.0;
    int all_one = 1;
    double a2[HO_MAXD];
    for (int i = 0; i < nr; i++) {
        double v = 0.0;
        if (wlt[i] < theta[i] && theta[i] <= fc[i]) {
            double d2 = fc[i] - wlt[i];
            if (d2 == 0.0) d2 = 1.0;
            v = (theta[i] - fc[i]) / d2;
        }
        if (theta[i] > fc[i]) v = 1.0;
        v = np_minimum(np_maximum(v, 0.0), 1.0);
        a2[i] = v;
        if (!(v == 1.0)) all_one = 0;
    }
    for (int i = 0; i < nr; i++) {
        double d1 = por[i] - wlt[i];
        if (d1 == 0.0) d1 = 1.0;
        double a1 = np_maximum(theta[i] / d1, local[i] / total);
        double v = all_one ? a2[i] * 0.1 : a2[i];
        rho[i] = fabs(a1 * v);
    }
    double tot = np_sum(rho, nr) * c->dz;
    if (tot == 0.0) tot = 1.0;
    for (int i = 0; i < nr; i++) rho[i] = rho[i] / tot;
    return water_k;
}

static __thread double g_last_arg_out[2] = {0.0, 0.0};
void ho_last_arg_out(double *out2)
{
    out2[0] = g_last_arg_out[0];
    out2[1] = g_last_arg_out[1];
}

#define HO_MAX_EVALS_PER_ATTEMPT 20000

/* ---- RichardsPDE.pde_fun, ref: richards_pde.py:172-395 ------------------------------- */
static void pde_fun(const ho_column *c, const ho_row *r, int view, const double *y, const double *dydz,
                    const double *n_rnd, double *C, double *sink, double *flux, double *tr_lf)
{
    int D = c->dim_d;
    int k = view == HO_VIEW_FIRST ? 1 : D - 2;
    int off = view == HO_VIEW_FIRST ? 0 : 1;
    int nr = view == HO_VIEW_FIRST ? c->n_root_first : c->n_root_int;
    double theta[HO_MAXD], K[HO_MAXD], kb[HO_MAXD];
    ho_model_eval(c, view, y, n_rnd, theta, K, C, kb, NULL);
    for (int i = 0; i < k; i++) {
        flux[i] = K[i] * (dydz[i] - 1.0);
        sink[i] = 0.0;
    }
    const double *roots = c->root_mid + off;
    double transp = 0.0, lat = 0.0;
    if (!r->spinup) {
        if (c->flag_hlift && !r->daylight) {
            /* ref: :234-254 */
            double c_sat = 1800 * c->lai;
            for (int i = 0; i < nr; i++) {
                double t1 = 1.0 - c->ipsi50 * y[i];
                double c_hr = c_sat * (t1 * t1) * roots[i];
                flux[i] += 0.5 * c_hr * (dydz[i] * c->dz);
            }
        }
        if (c->flag_et && r->daylight) {
            /* ref: :258-302 */
            double rho[HO_MAXD], x_out[HO_MAXD];
            double water_k = root_efficiency(c, off, nr, theta, rho);
            for (int i = 0; i < nr; i++) x_out[i] = rho[i] * roots[i];
            double tot_x = np_sum(x_out, nr) * c->dz;
            if (tot_x > 1.0) {
                for (int i = 0; i < nr; i++) x_out[i] = x_out[i] / tot_x;
                tot_x = np_sum(x_out, nr) * c->dz;
            }
            if (tot_x > 0.0) {
                double tot_tr = np_minimum(r->atm, water_k);
                double tr_pot = tot_tr / tot_x;
                for (int i = 0; i < nr; i++) {
                    x_out[i] = tr_pot * x_out[i];
                    sink[i] = -x_out[i];
                }
                transp = np_sum(x_out, nr) * c->dz;
            }
        }
    }
    if (c->flag_lf) {
        /* ref: :352-376 monitoring mode; indices are LOCAL to the k-cell slice */
        uint8_t sat[HO_MAXD];
        for (int i = 0; i < k; i++) sat[i] = y[i] >= c->psi_sat;
        int wtd_obs = r->wtd_obs < k - 1 ? r->wtd_obs : k - 1;
        int wtd_est = ho_find_wtd(sat, k);
        if (c->flag_predict) {
            /* ref: :312-351, repaired: low_lim = dim_d - (sat_cells - 1) as an int; np.linspace(1.5, 0.0, low_lim)
             * with low_lim <= 0 is taken as "no cell may drain" (the reference raises: TypeError for the float count
             * it passes, ValueError for a negative one, which the single-cell first call would hit). */
            double alpha_low = r->wet ? -2.5e-3 : -1.5e-3;
            int low_lim = k - (c->sat_cells - 1);
            if (low_lim > k) low_lim = k;   /* sat_cells <= 0 (refused by the Python face): never beyond the slice */
            if (low_lim > 0 && wtd_est < low_lim) {
                int j = wtd_est;
                /* numpy.linspace: step = (stop - start) / (num - 1); y = arange(num) * step + start; y[-1] = stop */
                double nu = 1.5;
                if (low_lim > 1) {
                    double step = -1.5 / (double)(low_lim - 1);
                    nu = (double)j * step + 1.5;
                    if (j == low_lim - 1) nu = 0.0;
                }
                double alpha_lat = alpha_low * (1.0 - pow((double)j / (double)low_lim, nu));
                sink[j] = np_minimum(alpha_lat * y[j], sink[j]);
                lat = fabs(sink[j]) * c->dz;
            }
        } else if (wtd_est < k && wtd_est < wtd_obs) {
            double lf[HO_MAXD];
            for (int j = wtd_est; j < wtd_obs; j++) {
                sink[j] = np_minimum(-2.5e-4 * y[j], sink[j]);
                lf[j - wtd_est] = fabs(sink[j]);
            }
            lat = np_sum(lf, wtd_obs - wtd_est) * c->dz;
        }
    }
    if (tr_lf) {
        tr_lf[0] = transp;
        tr_lf[1] = lat;
    }
    /* ref: richards_pde.py:380-391 -- var_arg_out is overwritten by EVERY pde_fun call; Simulation.run reads
     * it after solve() (simulation.py:629-630), i.e. it sees the interior call of the last RHS evaluation */
    g_last_arg_out[0] = transp;
    g_last_arg_out[1] = lat;
}

/* ---- RichardsPDE.bc_fun, ref: richards_pde.py:414-476 -> pL (qL = qR = 1, pR = 0) ---- */
static double bc_top(const ho_column *c, const ho_row *r, double y_top, const double *n_rnd)
{
    double q, K, C, kb, qinf;
    ho_model_eval(c, HO_VIEW_TOP, &y_top, n_rnd, &q, &K, &C, &kb, &qinf);
    double net_input = (1.0 - c->interception) * fabs(r->precip);
    double p_left = 0.0;
    if (y_top < c->psi_sat) p_left = np_minimum(net_input, qinf);
    if (!r->spinup) {
        /* theta_left spans the whole profile when z[0] == 0 (Porosity.__call__, porosity.py:200);
         * every entry shares the retention factor of psi[0], so the smallest delta_s decides np.all() */
        double ap = c->alpha * fabs(y_top);
        double fac = np_power(1.0 + np_power(ap, c->n), -c->m);
        double q_min = c->theta_res + c->evap_delta_min * fac;
        int allow = (q > c->theta_res) && (q_min > c->theta_res);
        if (allow && r->daylight) p_left = p_left - c->surface_evap;
    }
    return p_left;
}

/* ---- RichardsPDE.__call__, ref: richards_pde.py:82-160 -------------------------------- */
void ho_rhs(const ho_column *c, const ho_row *r, const double *y, const double *n_rnd,
            double *dydt, double *aux)
{
    int D = c->dim_d, M = D - 1;
    double ym[HO_MAXD], dym[HO_MAXD], cc[HO_MAXD], ss[HO_MAXD], ff[HO_MAXD], trlf0[2], trlf1[2];
    double half = 0.5 * c->dz; /* zxmp = xzmp = dz/2 on the uniform grid (:67-79) */
    for (int j = 0; j < M; j++) {
        ym[j] = 0.5 * (y[j] + y[j + 1]);       /* midpoints(), :575 */
        dym[j] = (y[j + 1] - y[j]) / c->dz;    /* :591 */
    }
    pde_fun(c, r, HO_VIEW_FIRST, ym, dym, n_rnd, cc, ss, ff, trlf0);
    double pL = bc_top(c, r, y[0], n_rnd);
    pde_fun(c, r, HO_VIEW_INTERIOR, ym + 1, dym + 1, n_rnd, cc + 1, ss + 1, ff + 1, trlf1);
    double denom = 1.0 * half * cc[0];
    if (denom == 0.0) denom = 1.0;
    dydt[0] = (pL + 1.0 * (ff[0] + half * ss[0])) / denom;
    for (int i = 1; i <= D - 2; i++) {
        denom = half * cc[i] + half * cc[i - 1];
        if (denom == 0.0) denom = 1.0;
        dydt[i] = (ff[i] - ff[i - 1] + (half * ss[i] + half * ss[i - 1])) / denom;
    }
    denom = -1.0 * half * cc[M - 1];
    if (denom == 0.0) denom = 1.0;
    dydt[D - 1] = (0.0 + 1.0 * (ff[M - 1] - half * ss[M - 1])) / denom;
    if (aux) {
        memcpy(aux, cc, sizeof(double) * M);
        memcpy(aux + M, ss, sizeof(double) * M);
        memcpy(aux + 2 * M, ff, sizeof(double) * M);
        aux[3 * M] = pL;
        aux[3 * M + 1] = trlf0[0];
        aux[3 * M + 2] = trlf0[1];
        aux[3 * M + 3] = trlf1[0];
        aux[3 * M + 4] = trlf1[1];
    }
}

/* ==== scipy BDF restated ============================================================== */
#define MAX_ORDER 5
#define NEWTON_MAXITER 4
#define MIN_FACTOR 0.2
#define MAX_FACTOR 10.0
#define NUM_JAC_DIFF_REJECT 2.0097183471152322e-14 /* EPS**0.875 */
#define NUM_JAC_DIFF_SMALL 1.8189894035458565e-12  /* EPS**0.75  */
#define NUM_JAC_DIFF_BIG 0.0001220703125           /* EPS**0.25  */
#define NUM_JAC_MIN_FACTOR 2.220446049250313e-13   /* 1e3*EPS    */

typedef struct {
    const ho_column *c;
    const ho_row *r;
    const double *n_rnd;
    int n;
    long nfev, njev, nlu;
    double jl[HO_MAXD], jd[HO_MAXD], ju[HO_MAXD]; /* J[i][i-1], J[i][i], J[i][i+1] */
    double factor[HO_MAXD];
    int have_factor;
    /* pivoted tridiagonal LU of I - c*J (LAPACK dgttrf layout) */
    double ldl[HO_MAXD], ld[HO_MAXD], ldu[HO_MAXD], ldu2[HO_MAXD];
    int ipiv[HO_MAXD];
} bdf_t;

/* test hooks: override EPS**0.875 to exercise num_jac's retry branch; count how often it is taken */
static double g_jac_reject = NUM_JAC_DIFF_REJECT;
static long g_jac_retry_count = 0;
/* scipy 1.5.2 (the reference's pin) against >= 1.9 (installed): see select_initial_step in bdf_integrate */
void ho_set_scipy_152(int on) { g_scipy_152 = on != 0; }
/* [0] solves whose h0 exceeded the interval, [1] solves whose min(100 h0, h1) did, [2] solves counted; reset by a call with reset != 0 */
void ho_clamp_counts(long *out3, int reset)
{
    for (int k = 0; k < 3; k++) out3[k] = g_clamp_counts[k];
    if (reset) g_clamp_counts[0] = g_clamp_counts[1] = g_clamp_counts[2] = 0;
}

void ho_debug_set_jac_reject(double v) { g_jac_reject = v > 0.0 ? v : NUM_JAC_DIFF_REJECT; }
long ho_debug_jac_retry_count(void) { return g_jac_retry_count; }

static void fun(bdf_t *b, const double *y, double *f)
{
    ho_rhs(b->c, b->r, y, b->n_rnd, f, NULL);
}

/* scipy.sparse argmax over one stored column (rows r0..r1 of |diff|) -- see _arg_min_or_max_axis:
 * a positive maximum wins (first occurrence, row order); an all-zero column resolves to row 0. */
static int col_argmax(const double *absd, int r0, int r1)
{
    int best = r0;
    for (int r = r0 + 1; r <= r1; r++)
        if (absd[r - r0] > absd[best - r0]) best = r;
    if (absd[best - r0] > 0.0) return best;
    if (absd[best - r0] == 0.0) return 0;
    /* NaN maximum: first row index that is not stored */
    return r0 > 0 ? 0 : r1 + 1;
}

/* scipy: common.num_jac + _sparse_num_jac for the tridiagonal structure and column groups */
static void num_jac(bdf_t *b, const double *y, const double *f, double threshold)
{
    int n = b->n;
    const int32_t *groups = b->c->groups;
    int ng = b->c->n_groups;
    double h[HO_MAXD], y_scale[HO_MAXD], *factor = b->factor;
    if (!b->have_factor) {
        for (int i = 0; i < n; i++) factor[i] = sqrt(EPS);
        b->have_factor = 1;
    }
    for (int i = 0; i < n; i++) {
        double f_sign = f[i] >= 0 ? 1.0 : -1.0;
        y_scale[i] = f_sign * np_maximum(threshold, fabs(y[i]));
        h[i] = (y[i] + factor[i] * y_scale[i]) - y[i];
        while (h[i] == 0.0) {
            factor[i] *= 10;
            h[i] = (y[i] + factor[i] * y_scale[i]) - y[i];
        }
    }
    static __thread double f_new[16][HO_MAXD], f_new2[16][HO_MAXD];
    double yp[HO_MAXD];
    for (int g = 0; g < ng; g++) {
        for (int i = 0; i < n; i++) yp[i] = y[i] + (groups[i] == g ? h[i] : 0.0);
        fun(b, yp, f_new[g]);
    }
    /* diff[r][j] for r in {j-1,j,j+1}: stored as dcol[j][0..2] (slot = r-(j-1)) */
    static __thread double dcol[HO_MAXD][3];
    double max_diff[HO_MAXD], scale[HO_MAXD];
    int too_small[HO_MAXD], any_small = 0;
    for (int j = 0; j < n; j++) {
        int r0 = j > 0 ? j - 1 : 0, r1 = j < n - 1 ? j + 1 : n - 1;
        double absd[3];
        for (int r = r0; r <= r1; r++) {
            double d = f_new[groups[j]][r] - f[r];
            dcol[j][r - (j - 1)] = d;
            absd[r - r0] = fabs(d);
        }
        int mi = col_argmax(absd, r0, r1);
        max_diff[j] = (mi >= r0 && mi <= r1) ? absd[mi - r0] : 0.0;
        scale[j] = np_maximum(fabs(f[mi]), fabs(f_new[groups[j]][mi]));
        too_small[j] = max_diff[j] < g_jac_reject * scale[j];
        any_small |= too_small[j];
    }
    if (any_small) {
        g_jac_retry_count++;
        double new_factor[HO_MAXD], h_new[HO_MAXD];
        int used[16] = {0};
        for (int j = 0; j < n; j++) {
            h_new[j] = 0.0;
            if (too_small[j]) {
                new_factor[j] = 10 * factor[j];
                h_new[j] = (y[j] + new_factor[j] * y_scale[j]) - y[j];
                used[groups[j]] = 1;
            }
        }
        for (int g = 0; g < ng; g++) {
            if (!used[g]) continue;
            for (int i = 0; i < n; i++) yp[i] = y[i] + (groups[i] == g ? h_new[i] : 0.0);
            fun(b, yp, f_new2[g]);
        }
        for (int j = 0; j < n; j++) {
            if (!too_small[j]) continue;
            int r0 = j > 0 ? j - 1 : 0, r1 = j < n - 1 ? j + 1 : n - 1;
            double absd[3], dn[3];
            for (int r = r0; r <= r1; r++) {
                dn[r - r0] = f_new2[groups[j]][r] - f[r];
                absd[r - r0] = fabs(dn[r - r0]);
            }
            int mi = col_argmax(absd, r0, r1);
            double max_diff_new = (mi >= r0 && mi <= r1) ? absd[mi - r0] : 0.0;
            double scale_new = np_maximum(fabs(f[mi]), fabs(f_new2[groups[j]][mi]));
            if (max_diff[j] * scale_new < max_diff_new * scale[j]) {
                factor[j] = new_factor[j];
                h[j] = h_new[j];
                for (int r = r0; r <= r1; r++) dcol[j][r - (j - 1)] = dn[r - r0];
                scale[j] = scale_new;
                max_diff[j] = max_diff_new;
            }
        }
    }
    for (int j = 0; j < n; j++) {
        if (j > 0) b->ju[j - 1] = dcol[j][0] / h[j]; /* J[j-1][j] */
        b->jd[j] = dcol[j][1] / h[j];
        if (j < n - 1) b->jl[j + 1] = dcol[j][2] / h[j]; /* J[j+1][j] */
        if (max_diff[j] < NUM_JAC_DIFF_SMALL * scale[j]) factor[j] *= 10;
        if (max_diff[j] > NUM_JAC_DIFF_BIG * scale[j]) factor[j] *= 0.1;
        factor[j] = np_maximum(factor[j], NUM_JAC_MIN_FACTOR);
    }
    b->jl[0] = 0.0;
    b->ju[n - 1] = 0.0;
}

/* bdf.py jac_wrapped: njev += 1; f = fun_single(t, y) (not counted in nfev); num_jac(...) */
static void eval_jac(bdf_t *b, const double *y, double atol)
{
    double f[HO_MAXD];
    b->njev++;
    fun(b, y, f);
    num_jac(b, y, f, atol);
}

/* LU = splu(I - c*J): restated as a partially pivoted tridiagonal factorisation */
static void lu_factor(bdf_t *b, double cc)
{
    int n = b->n;
    double *dl = b->ldl, *d = b->ld, *du = b->ldu, *du2 = b->ldu2;
    b->nlu++;
    for (int i = 0; i < n; i++) {
        d[i] = 1.0 - cc * b->jd[i];
        if (i < n - 1) {
            du[i] = -cc * b->ju[i];
            dl[i] = -cc * b->jl[i + 1];
        }
        du2[i] = 0.0;
        b->ipiv[i] = i;
    }
    for (int i = 0; i < n - 1; i++) {
        if (fabs(d[i]) >= fabs(dl[i])) {
            if (d[i] != 0.0) {
                double fact = dl[i] / d[i];
                dl[i] = fact;
                d[i + 1] -= fact * du[i];
            }
        } else {
            double fact = d[i] / dl[i];
            d[i] = dl[i];
            dl[i] = fact;
            double temp = du[i];
            du[i] = d[i + 1];
            d[i + 1] = temp - fact * d[i + 1];
            if (i < n - 2) {
                du2[i] = du[i + 1];
                du[i + 1] = -fact * du[i + 1];
            }
            b->ipiv[i] = i + 1;
        }
    }
}

static void lu_solve(const bdf_t *b, double *x)
{
    int n = b->n;
    for (int i = 0; i < n - 1; i++) {
        if (b->ipiv[i] == i) {
            x[i + 1] -= b->ldl[i] * x[i];
        } else {
            double temp = x[i];
            x[i] = x[i + 1];
            x[i + 1] = temp - b->ldl[i] * x[i];
        }
    }
    x[n - 1] /= b->ld[n - 1];
    if (n > 1) x[n - 2] = (x[n - 2] - b->ldu[n - 2] * x[n - 1]) / b->ld[n - 2];
    for (int i = n - 3; i >= 0; i--)
        x[i] = (x[i] - b->ldu[i] * x[i + 1] - b->ldu2[i] * x[i + 2]) / b->ld[i];
}

/* scipy: bdf.py compute_R / change_D */
static void compute_R(int order, double factor, double R[6][6])
{
    double M[6][6];
    memset(M, 0, sizeof(M));
    for (int i = 1; i <= order; i++)
        for (int j = 1; j <= order; j++) M[i][j] = ((double)(i - 1) - factor * (double)j) / (double)i;
    for (int j = 0; j <= order; j++) M[0][j] = 1.0;
    for (int j = 0; j <= order; j++) {
        double p = 1.0;
        for (int i = 0; i <= order; i++) {
            p *= M[i][j];
            R[i][j] = p;
        }
    }
}

static void change_D(double D[][HO_MAXD], int n, int order, double factor)
{
    double R[6][6], U[6][6], RU[6][6];
    compute_R(order, factor, R);
    compute_R(order, 1.0, U);
    for (int i = 0; i <= order; i++)
        for (int j = 0; j <= order; j++) {
            double s = 0.0;
            for (int k = 0; k <= order; k++) s += R[i][k] * U[k][j];
            RU[i][j] = s;
        }
    for (int x = 0; x < n; x++) {
        double col[6];
        for (int i = 0; i <= order; i++) {
            double s = 0.0;
            for (int k = 0; k <= order; k++) s += RU[k][i] * D[k][x]; /* RU.T @ D */
            col[i] = s;
        }
        for (int i = 0; i <= order; i++) D[i][x] = col[i];
    }
}

static int all_finite(const double *f, int n)
{
    for (int i = 0; i < n; i++)
        if (!isfinite(f[i])) return 0;
    return 1;
}

/* scipy: solve_ivp(method='BDF', rtol=atol=1e-3, jac_sparsity=tridiag) on [t0, tf].
 * Returns 1 on success; y_out = last accepted state either way (sol.y[:, -1]). */
static int bdf_integrate(bdf_t *b, double t0, double tf, const double *y0, double *y_out,
                         long *nsteps_out, double *t_steps, int cap_steps)
{
    const double rtol = 1.0e-3, atol = 1.0e-3; /* ref: richards_pde.py:496 */
    int n = b->n;
    static __thread double D[MAX_ORDER + 3][HO_MAXD];
    double y[HO_MAXD], f[HO_MAXD], scale[HO_MAXD], tmp[HO_MAXD];
    double t = t0;
    long nsteps = 0;
    memcpy(y, y0, sizeof(double) * n);
    if (t_steps && cap_steps > 0) t_steps[0] = t0;

    /* --- BDF.__init__ --- */
    fun(b, y, f);
    b->nfev++;
    double h_abs;
    {   /* select_initial_step (common.py), order = 1 */
        double interval = fabs(tf - t0);
        for (int i = 0; i < n; i++) scale[i] = atol + fabs(y[i]) * rtol;
        for (int i = 0; i < n; i++) tmp[i] = y[i] / scale[i];
        double d0 = rms_norm(tmp, n);
        for (int i = 0; i < n; i++) tmp[i] = f[i] / scale[i];
        double d1 = rms_norm(tmp, n);
        double h0 = (d0 < 1e-5 || d1 < 1e-5) ? 1e-6 : 0.01 * d0 / d1;
        /* scipy >= 1.9 clamps h0 and the returned step to the interval (common.py select_initial_step(..., t_bound, ...));
         * the reference's pinned scipy==1.5.2 has neither clamp and leaves an over-long first step to _step_impl's
         * `t_new - t_bound > 0` rule (change_D by the ratio).  The golden vectors were made with 1.15.3; the switch
         * below runs the 1.5.2 form and the counters say how often the two differ on this path. */
        if (h0 > interval) g_clamp_counts[0]++;
        if (!g_scipy_152) h0 = h0 < interval ? h0 : interval;
        double y1[HO_MAXD], f1[HO_MAXD];
        for (int i = 0; i < n; i++) y1[i] = y[i] + h0 * 1.0 * f[i];
        fun(b, y1, f1);
        b->nfev++;
        for (int i = 0; i < n; i++) tmp[i] = (f1[i] - f[i]) / scale[i];
        double d2 = rms_norm(tmp, n) / h0;
        double h1;
        if (d1 <= 1e-15 && d2 <= 1e-15)
            h1 = np_maximum(1e-6, h0 * 1e-3);
        else
            h1 = pow(0.01 / np_maximum(d1, d2), 1.0 / 2.0);
        h_abs = 100 * h0;
        if (h1 < h_abs) h_abs = h1;
        g_clamp_counts[2]++;
        if (interval < h_abs) {
            g_clamp_counts[1]++;
            if (!g_scipy_152) h_abs = interval;
        }
    }
    double newton_tol = np_maximum(10 * EPS / rtol, np_minimum(0.03, sqrt(rtol)));
    b->have_factor = 0;
    eval_jac(b, y, atol);

    static const double kappa[6] = {0, -0.1850, -1.0 / 9, -0.0823, -0.0415, 0};
    double gamma[6], alpha[6], error_const[7];
    gamma[0] = 0.0;
    for (int k = 1; k <= MAX_ORDER; k++) gamma[k] = gamma[k - 1] + 1.0 / k;
    for (int k = 0; k <= MAX_ORDER; k++) {
        alpha[k] = (1 - kappa[k]) * gamma[k];
        error_const[k] = kappa[k] * gamma[k] + 1.0 / (k + 1);
    }
    for (int i = 0; i < n; i++) {
        D[0][i] = y[i];
        D[1][i] = f[i] * h_abs * 1.0;
    }
    int order = 1, n_equal_steps = 0, have_lu = 0, ok = 1;

    /* --- solve_ivp loop over OdeSolver.step / BDF._step_impl --- */
    while (t != tf) {
        double min_step = 10 * fabs(nextafter(t, INFINITY) - t);
        if (h_abs < min_step) {
            change_D(D, n, order, min_step / h_abs);
            h_abs = min_step;
            n_equal_steps = 0;
        }
        /* scipy: current_jac = (self.jac is None).  With a finite-difference Jacobian self.jac is
         * the wrapper, so it starts False: one Jacobian refresh per step is allowed. */
        int current_jac = 0;
        int step_accepted = 0;
        double t_new = t, y_new[HO_MAXD], d[HO_MAXD], y_predict[HO_MAXD], psi[HO_MAXD];
        double error_norm = 0.0, safety = 0.0;
        while (!step_accepted) {
            if (h_abs < min_step) {
                ok = 0;
                goto done;
            }
            /* Work budget of one attempt, mirroring the kernel's iteration budget (hc_step.h MAX_PHASE_ITERATIONS):
             * on a state sliding along a discontinuity of the RHS the step controller cycles for ever with
             * h ~ 1e-11 (SciPy's BDF has no exit there either).  RHS evaluations incl. the 5 per FD Jacobian. */
            if (b->nfev + 5 * b->njev > HO_MAX_EVALS_PER_ATTEMPT) {
                ok = 0;
                goto done;
            }
            double h = h_abs;
            t_new = t + h;
            if (t_new - tf > 0) {
                t_new = tf;
                change_D(D, n, order, fabs(t_new - t) / h_abs);
                n_equal_steps = 0;
                have_lu = 0;
            }
            h = t_new - t;
            h_abs = fabs(h);
            for (int i = 0; i < n; i++) {
                double s = D[0][i];
                for (int k = 1; k <= order; k++) s += D[k][i];
                y_predict[i] = s;
                scale[i] = atol + rtol * fabs(s);
                double p = 0.0;
                for (int k = 1; k <= order; k++) p += D[k][i] * gamma[k];
                psi[i] = p / alpha[order];
            }
            int converged = 0, n_iter = 0;
            double cc = h / alpha[order];
            while (!converged) {
                if (!have_lu) {
                    lu_factor(b, cc);
                    have_lu = 1;
                }
                /* solve_bdf_system */
                for (int i = 0; i < n; i++) {
                    d[i] = 0.0;
                    y_new[i] = y_predict[i];
                }
                double dy_norm_old = -1.0;
                int k;
                for (k = 0; k < NEWTON_MAXITER; k++) {
                    fun(b, y_new, f);
                    b->nfev++;
                    if (!all_finite(f, n)) break;
                    double dy[HO_MAXD];
                    for (int i = 0; i < n; i++) dy[i] = cc * f[i] - psi[i] - d[i];
                    lu_solve(b, dy);
                    for (int i = 0; i < n; i++) tmp[i] = dy[i] / scale[i];
                    double dy_norm = rms_norm(tmp, n);
                    int have_rate = dy_norm_old >= 0.0;
                    double rate = have_rate ? dy_norm / dy_norm_old : 0.0;
                    if (have_rate && (rate >= 1 ||
                                      pow(rate, NEWTON_MAXITER - k) / (1 - rate) * dy_norm > newton_tol))
                        break;
                    for (int i = 0; i < n; i++) {
                        y_new[i] += dy[i];
                        d[i] += dy[i];
                    }
                    if (dy_norm == 0 || (have_rate && rate / (1 - rate) * dy_norm < newton_tol)) {
                        converged = 1;
                        break;
                    }
                    dy_norm_old = dy_norm;
                }
                n_iter = k + 1 > NEWTON_MAXITER ? NEWTON_MAXITER : k + 1;
                if (!converged) {
                    if (current_jac) break;
                    eval_jac(b, y_predict, atol);
                    have_lu = 0;
                    current_jac = 1;
                }
            }
            if (!converged) {
                h_abs *= 0.5;
                change_D(D, n, order, 0.5);
                n_equal_steps = 0;
                have_lu = 0;
                continue;
            }
            safety = 0.9 * (2 * NEWTON_MAXITER + 1) / (double)(2 * NEWTON_MAXITER + n_iter);
            for (int i = 0; i < n; i++) {
                scale[i] = atol + rtol * fabs(y_new[i]);
                tmp[i] = error_const[order] * d[i] / scale[i];
            }
            error_norm = rms_norm(tmp, n);
            if (error_norm > 1) {
                double factor = np_maximum(MIN_FACTOR, safety * pow(error_norm, -1.0 / (order + 1)));
                h_abs *= factor;
                change_D(D, n, order, factor);
                n_equal_steps = 0;
            } else {
                step_accepted = 1;
            }
        }
        n_equal_steps++;
        t = t_new;
        memcpy(y, y_new, sizeof(double) * n);
        nsteps++;
        if (t_steps && nsteps < cap_steps) t_steps[nsteps] = t;
        for (int i = 0; i < n; i++) {
            D[order + 2][i] = d[i] - D[order + 1][i];
            D[order + 1][i] = d[i];
            for (int k = order; k >= 0; k--) D[k][i] += D[k + 1][i];
        }
        if (n_equal_steps < order + 1) continue;
        double error_m_norm = INFINITY, error_p_norm = INFINITY;
        if (order > 1) {
            for (int i = 0; i < n; i++) tmp[i] = error_const[order - 1] * D[order][i] / scale[i];
            error_m_norm = rms_norm(tmp, n);
        }
        if (order < MAX_ORDER) {
            for (int i = 0; i < n; i++) tmp[i] = error_const[order + 1] * D[order + 2][i] / scale[i];
            error_p_norm = rms_norm(tmp, n);
        }
        double en[3] = {error_m_norm, error_norm, error_p_norm}, factors[3];
        int best = 0;
        for (int k = 0; k < 3; k++) {
            factors[k] = pow(en[k], -1.0 / (order + k));
            if (factors[k] > factors[best]) best = k;
        }
        order += best - 1;
        double factor = np_minimum(MAX_FACTOR, safety * factors[best]);
        h_abs *= factor;
        change_D(D, n, order, factor);
        n_equal_steps = 0;
        have_lu = 0;
    }
done:
    memcpy(y_out, y, sizeof(double) * n);
    *nsteps_out = nsteps;
    return ok;
}

void ho_solve_row(const ho_column *c, const ho_row *r, double t0, double t1, const double *y0,
                  double *n_rnd, double *y1, ho_stats *st, double *t_steps, int cap_steps)
{
    /* ref: richards_pde.py:493-536 */
    static __thread bdf_t b;
    ho_stats s = {0, 0, 0, 0, 0, 0};
    int n_trials = 5;
    while (n_trials > 0) {
        b.c = c;
        b.r = r;
        b.n_rnd = n_rnd;
        b.n = c->dim_d;
        b.nfev = b.njev = b.nlu = 0;
        long nsteps = 0;
        int ok = bdf_integrate(&b, t0, t1, y0, y1, &nsteps, t_steps, cap_steps);
        s.nfev += b.nfev;
        s.njev += b.njev;
        s.nlu += b.nlu;
        s.nsteps = nsteps;
        s.attempts++;
        if (ok) {
            s.success = 1;
            break;
        }
        for (int i = 0; i < c->dim_d; i++) n_rnd[i] *= 0.8;
        n_trials--;
    }
    if (st) *st = s;
}

void ho_run(const ho_column *c, int64_t T, const double *precip, const double *atm,
            const uint8_t *daylight, const int32_t *wtd_obs, const uint8_t *refresh,
            int64_t row_begin, int64_t row_end, double *psi, double *base_noise, double *fresh,
            int32_t *wtd_est, double *psi_out, int32_t *per_row)
{
    /* ref: simulation.py:576-626.  Rows [row_begin, row_end), row 0 is the initial state. */
    int D = c->dim_d;
    int64_t k_fresh = 0;
    double y1[HO_MAXD];
    uint8_t sat[HO_MAXD];
    (void)T;
    for (int64_t i = row_begin; i < row_end; i++) {
        if (i == 0) continue;
        if (wtd_obs[i] < 0) continue; /* :582-588 observation not on the grid -> row skipped */
        ho_row r = {precip[i], atm[i], daylight[i] & 1, wtd_obs[i], 0, (daylight[i] >> 1) & 1};
        double *noise = base_noise;          /* :592 the base vector, by reference */
        if (refresh[i]) {                    /* :599-602 fresh vector lives for this row only */
            noise = fresh + k_fresh * D;
            k_fresh++;
        }
        ho_stats st;
        ho_solve_row(c, &r, (double)(i - 1), (double)i, psi, noise, y1, &st, NULL, 0);
        for (int k = 0; k < D; k++) sat[k] = y1[k] >= c->psi_sat;
        int w = ho_find_wtd(sat, D);
        if (wtd_est) wtd_est[i] = w;
        memcpy(psi, y1, sizeof(double) * D);
        if (psi_out) memcpy(psi_out + i * D, y1, sizeof(double) * D);
        if (per_row) {
            int32_t *p = per_row + i * 6;
            p[0] = (int32_t)st.nfev;
            p[1] = (int32_t)st.njev;
            p[2] = (int32_t)st.nlu;
            p[3] = (int32_t)st.nsteps;
            p[4] = (int32_t)st.attempts;
            p[5] = refresh[i];
        }
    }
}

/* same as ho_run, additionally recording transpiration / lateral_flow per row: diag [T][2] */
void ho_run_diag(const ho_column *c, int64_t T, const double *precip, const double *atm,
                 const uint8_t *daylight, const int32_t *wtd_obs, const uint8_t *refresh,
                 int64_t row_begin, int64_t row_end, double *psi, double *base_noise, double *fresh,
                 int32_t *wtd_est, double *diag)
{
    int D = c->dim_d;
    int64_t k_fresh = 0;
    double y1[HO_MAXD];
    uint8_t sat[HO_MAXD];
    (void)T;
    for (int64_t i = row_begin; i < row_end; i++) {
        if (i == 0 || wtd_obs[i] < 0) continue;
        ho_row r = {precip[i], atm[i], daylight[i] & 1, wtd_obs[i], 0, (daylight[i] >> 1) & 1};
        double *noise = base_noise;
        if (refresh[i]) {
            noise = fresh + k_fresh * D;
            k_fresh++;
        }
        ho_solve_row(c, &r, (double)(i - 1), (double)i, psi, noise, y1, NULL, NULL, 0);
        for (int k = 0; k < D; k++) sat[k] = y1[k] >= c->psi_sat;
        if (wtd_est) wtd_est[i] = ho_find_wtd(sat, D);
        memcpy(psi, y1, sizeof(double) * D);
        ho_last_arg_out(diag + 2 * i);
    }
}

int ho_spinup(const ho_column *c, const ho_row *row0, double zwtd0_cm, const double *z,
              double *psi, double *n_rnd, int max_iter)
{
    /* ref: simulation.py:444-480 (SPINUP flag set by the caller through row0->spinup) */
    int D = c->dim_d;
    double y1[HO_MAXD];
    uint8_t sat[HO_MAXD];
    for (int j = 0; j < max_iter; j++) {
        ho_solve_row(c, row0, 0.0, 1.0, psi, n_rnd, y1, NULL, NULL, 0);
        for (int k = 0; k < D; k++) sat[k] = y1[k] >= c->psi_sat;
        int w = ho_find_wtd(sat, D);
        double abs_error = fabs(zwtd0_cm - z[w]);
        double mse = 0.0;
        for (int k = 0; k < D; k++) mse += (y1[k] - psi[k]) * (y1[k] - psi[k]);
        mse /= D;
        memcpy(psi, y1, sizeof(double) * D);
        if (abs_error <= 2.0 * c->dz && mse <= 0.01) return j + 1;
    }
    return max_iter;
}
