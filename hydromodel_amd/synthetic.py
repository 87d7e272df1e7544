"""Synthetic forcing and synthetic wells for benchmarks and parity tests.

The reference ships no data (``/root/reference/.gitignore:1-3``; README.md:9-11),
so every run here uses the forcing defined in BASELINE.md §3 / SURVEY.md §8(d):

* header-less CSV ``ID, Datenum, Precipitation_cm, WTD_m``
  (format read by ``code/berkeley_hydro_main.py:115-116``);
* 48 rows per day, ``Datenum = 733682 + k/48`` (2008-10-01 00:00, start of a water year);
* daily rain totals from ``numpy.random.default_rng(123)``: P(rain) = 0.35 while the
  day-of-water-year is < 182, 0.03 otherwise; depth ~ Exp(mean 1.6 cm); the daily
  total is spread uniformly over that day's 48 rows;
* ``WTD_m = -3.00`` constant; whole years only (365-day years, no leap days).
"""
import json
from pathlib import Path

import numpy as np

ROWS_PER_DAY = 48
DAYS_PER_YEAR = 365
DATENUM0 = 733682.0


def synthetic_daily_rain(n_days, seed=123):
    """Daily precipitation totals [cm] for ``n_days`` days (see module docstring)."""
    rng = np.random.default_rng(seed)
    u = rng.random(n_days)
    depth = rng.exponential(1.6, n_days)
    dowy = np.arange(n_days) % DAYS_PER_YEAR
    p_rain = np.where(dowy < 182, 0.35, 0.03)
    return np.where(u < p_rain, depth, 0.0)


def synthetic_forcing(n_years=1, seed=123, wtd_m=-3.00):
    """Return (ids, datenum, precipitation_cm, wtd_m) arrays, 17 520 rows per year."""
    n_days = int(n_years) * DAYS_PER_YEAR
    daily = synthetic_daily_rain(n_days, seed)
    n_rows = n_days * ROWS_PER_DAY
    k = np.arange(n_rows)
    datenum = DATENUM0 + k / float(ROWS_PER_DAY)
    precip = np.repeat(daily / float(ROWS_PER_DAY), ROWS_PER_DAY)
    wtd = np.full(n_rows, float(wtd_m))
    return k + 1, datenum, precip, wtd


def synthetic_forcing_frame(n_years=1, seed=123, wtd_m=-3.00):
    """Same as :func:`synthetic_forcing` but as the pandas.DataFrame the CLI builds."""
    import pandas as pd
    ids, datenum, precip, wtd = synthetic_forcing(n_years, seed, wtd_m)
    return pd.DataFrame({"ID": ids, "Datenum": datenum,
                         "Precipitation_cm": precip, "WTD_m": wtd})


def write_forcing_csv(path, n_years=1, seed=123, wtd_m=-3.00):
    """Write the header-less 4-column CSV that ``--data`` expects."""
    ids, datenum, precip, wtd = synthetic_forcing(n_years, seed, wtd_m)
    with open(path, "w") as fh:
        for i in range(ids.size):
            fh.write(f"{int(ids[i]):d},{float(datenum[i])!r},{float(precip[i])!r},{float(wtd[i])!r}\n")
    return Path(path)


def synthetic_well(dim_d):
    """Well record with ``dim_d`` depth nodes: max_depth = 5*(D-1) cm."""
    return {"soil": 0.0, "saprolite": 50.0, "weathered": 200.0,
            "max_depth": 5.0 * (int(dim_d) - 1), "sat_depth": 100.0}


def write_site_information(path, wells):
    """Write a site_information.json holding ``wells`` = {well_no: record}."""
    with open(path, "w") as fh:
        json.dump({"Well": {str(k): v for k, v in wells.items()}}, fh, indent=1)
    return Path(path)


def default_parameters():
    """The values of ``code/model_parameters/input_parameters.json`` (reference defaults)."""
    return {
        "IC_Filename": None,
        "Data_Filename": None,
        "Output_Name": "Sim_00",
        "Well_No": 10,
        "Site_Information": None,
        "Water_Content": {"Theta_Max": 0.30, "Theta_Min": 0.05, "Theta_Residual": 0.04,
                          "Wilting_Point_cm": -1500.0, "Field_Capacity_cm": 340.0},
        "Hydraulic_Conductivity": {"Sat_Soil": 8.50, "Sat_Saprolite": 3.20,
                                   "Sat_Fresh_Bedrock": 0.10, "Sigma_Noise": 2.0,
                                   "Lambda_Exponent": 1.0},
        "Environmental": {"Interception_pct": 0.1, "Evaporation_pct": 0.0,
                          "Wet_Season_pct": 1.0, "Atmospheric_Demand": 0.5},
        "Trees": {"Leaf_Area_Index": 4.0, "Max_Root_Depth_cm": 1000.0,
                  "Root_Pdf_Profile": "Negative_Exp"},
        "Hydrological_Model": {"Name": "vrettas_fung", "Porosity_Profile": "Stratified"},
        "Soil_Properties": {"n": 2.0, "a0": 0.009, "psi_sat": -0.0047, "epsilon": 1.0e-7},
        "Simulation_Flags": {"SPINUP": False, "ET": True, "LF": True,
                             "HLIFT": False, "PREDICT": False},
    }
