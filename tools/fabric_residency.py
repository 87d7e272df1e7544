"""Where does the step kernel's L2 <-> fabric traffic go -- Infinity Cache or HBM -- and does it bound the kernel?

    python tools/fabric_residency.py cus <n_cu> [members_per_cu]     one 48-row launch on a grid of n_cu workgroups
                                                                     (HYDROCOL_DEBUG_CUS), members scaled with the grid
    python tools/fabric_residency.py polluter [n_cu]                 the same launch on n_cu CUs, alone and while a
                                                                     device-to-device copy sweeps 2 x 1 GiB on the idle CUs

`cus` at 64 / 128 / 192 / 256 workgroups scales the working set of the per-wave regions (8 waves x ~35 KB touched per CU) from
~18 to ~72 MB and the demand on the fabric with it; a kernel bounded by fabric bandwidth loses per-CU rate as the grid grows, a
latency-bound one does not.  tools/gpu_r5_residency.sh runs each size under rocprofv3 (FETCH_SIZE / WRITE_SIZE / TCC hit rate).
`polluter` evicts the Infinity Cache under the kernel: a copy that streams 2 GiB per pass (eight times the cache) runs on the CUs
the reduced grid leaves idle; the step kernel's lines then come from HBM whenever their reuse distance exceeds what the cache
keeps under that stream.
"""
import os
import sys
import time

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np  # noqa: E402

mode = sys.argv[1]
n_cu = int(sys.argv[2]) if len(sys.argv) > 2 else 128
os.environ["HYDROCOL_DEBUG_CUS"] = str(n_cu)

from hydromodel_amd.digest import ColumnTables, ForcingDigest  # noqa: E402
from hydromodel_amd.ensemble import PHILOX_DRAW_SPINUP, spinup_on_gpu  # noqa: E402
from hydromodel_amd.stepper import EnsembleStepper  # noqa: E402
from hydromodel_amd.synthetic import default_parameters, synthetic_forcing_frame, synthetic_well  # noqa: E402

D = int(os.environ.get("HC_RES_DEPTH", "300"))
params = default_parameters()
cols = ColumnTables(params, synthetic_well(D))
forcing = ForcingDigest(params, synthetic_forcing_frame(1), cols)
cache = os.environ.get("HC_RES_IC", "")
if cache and os.path.exists(cache):
    ic = np.load(cache)["ic"]
else:
    probe = EnsembleStepper(cols, forcing, 1)
    probe.set_noise_philox(1, 0)
    n0 = probe.philox_normals(0, PHILOX_DRAW_SPINUP)
    probe.close()
    ic, _, _ = spinup_on_gpu(cols, forcing, n0)
    if cache:
        np.savez(cache, ic=ic)
        print("initial condition cached")
        sys.exit(0)


def launch(members):
    st = EnsembleStepper(cols, forcing, members)
    st.set_state(ic)
    st.set_noise_philox(42, 0)
    return st


if mode == "cus":
    per_cu = int(sys.argv[3]) if len(sys.argv) > 3 else 128
    members = per_cu * n_cu
    st = launch(members)
    out = st.step_rows(1, 48)
    rate = members / (out["kernel_ms"] * 1e-3)
    print(f"grid {n_cu} workgroups, {members} members ({per_cu} per CU) x 48 rows x D = {D}: kernel {out['kernel_ms']:.2f} ms, "
          f"{rate:.0f} column-days/s = {rate / n_cu:.1f} per CU")
    st.close()
else:
    import threading

    import torch
    members = 128 * n_cu
    res = {}
    for label in ("alone", "with the copy"):
        st = launch(members)
        st.step_rows(1, 48)                                    # warm-up day
        stop = threading.Event()
        copied = [0]
        if label != "alone":
            a = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
            b = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
            side = torch.cuda.Stream()

            def sweep():
                with torch.cuda.stream(side):
                    while not stop.is_set():
                        for _ in range(8):
                            b.copy_(a)
                            a.copy_(b)
                        side.synchronize()
                        copied[0] += 16
            th = threading.Thread(target=sweep)
            th.start()
            time.sleep(0.3)
        c0, t0 = copied[0], time.perf_counter()
        out = st.step_rows(49, 48)
        dt = time.perf_counter() - t0
        n_copies = copied[0] - c0
        stop.set()
        if label != "alone":
            th.join()
        res[label] = (out["kernel_ms"], n_copies * 2.0 * (1 << 30) / dt / 1e12)
        st.close()
    k0, k1 = res["alone"][0], res["with the copy"][0]
    print(f"grid {n_cu} workgroups, {members} members x 48 rows x D = {D}: kernel {k0:.2f} ms alone, {k1:.2f} ms while a copy "
          f"moves {res['with the copy'][1]:.2f} TB/s (read + write) through the other CUs: x{k1 / k0:.3f}")
