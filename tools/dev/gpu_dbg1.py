import sys, numpy as np, time
import os; R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,R); sys.path.insert(0,R+'/tests')
from helpers import *
from hydromodel_amd.stepper import EnsembleStepper
from oracle.oracle import Oracle
for well in (200, 300, 1):
    params, cols, forcing = digest(well)
    g = golden(f'g34_states_{well}.npz')
    D = cols.dim_d
    names=[n for n in g['names'] if not n.startswith('hlift') and n not in ('spinup','no_et_day','no_lf')]
    N=len(names)
    st = EnsembleStepper(cols, forcing, N)
    Y = np.array([g[f'{n}_y'] for n in names]); st.set_state(Y)
    base = np.tile(g['n_rnd'],(N,1)); st.set_noise_host(base)
    o = Oracle(cols, forcing.surface_evap)
    # pick forcing rows: night row 2 (hour 1), day row 24 (hour 12)
    for row in (2, 24, 30):
        dydt, aux = st.rhs(row, want_aux=True)
        for k,n in enumerate(names):
            r = Oracle.row(forcing.precip[row], forcing.atm[row], forcing.daylight[row], forcing.wtd_obs[row])
            ref, ra = o.rhs(r, Y[k], g['n_rnd'], want_aux=True)
            e = np.max(np.abs(dydt[k]-ref)/np.maximum(1,np.abs(ref)))
            ec = rel_err(aux['c'][k], ra['c'], 1e-7); es=rel_err(aux['s'][k], ra['s']); ef=rel_err(aux['f'][k], ra['f'])
            print(well,row,f'{n:18s} dydt {e:.2e} c {ec:.1e} s {es:.1e} f {ef:.1e} pL {abs(aux["pL"][k]-ra["pL"]):.1e}')
    mn = st.model_nodes()
    for k,n in enumerate(names[:3]):
        q,K,C_,kb,qi = o.model_eval(0, Y[k], g['n_rnd'])
        print(' nodes', n, rel_err(mn['theta'][k],q), rel_err(mn['K'][k],K), rel_err(mn['C'][k],C_,1e-7), rel_err(mn['K_bkg'][k],kb), abs(mn['q_inf_max'][k]-qi))
    # single-row solves vs oracle
    for row in (2,24):
        st.set_state(Y); st.set_noise_host(base)
        t=time.time()
        out = st.step_rows(row, 1, fresh_noise=np.zeros((st.n_refresh(row,1),N,D)), want_wtd=True, want_stats=True)
        y1 = st.get_state()
        for k,n in enumerate(names):
            r = Oracle.row(forcing.precip[row], forcing.atm[row], forcing.daylight[row], forcing.wtd_obs[row])
            yo, so, _, _ = o.solve_row(r, row-1, row, Y[k], g['n_rnd'])
            e = np.max(np.abs(y1[k]-yo)/(1+np.abs(yo)))
            print(well,'solve row',row,f'{n:18s} err {e:.2e} gpu stats {out["stats"][0,k].tolist()} oracle {so["nfev"]},{so["njev"]},{so["nlu"]},{so["nsteps"]},{so["attempts"]}  kernel_ms {out["kernel_ms"]:.2f}')
    st.close()
