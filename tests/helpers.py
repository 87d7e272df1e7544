"""Shared builders for the tests: wells, digests, oracle columns."""
from functools import lru_cache
from pathlib import Path

import numpy as np

from hydromodel_amd.digest import ColumnTables, ForcingDigest
from hydromodel_amd.synthetic import default_parameters, synthetic_forcing_frame, synthetic_well

GOLDEN = Path(__file__).resolve().parent / "golden"

# well 1 of the reference's site_information.json (D = 101), the two synthetic wells and the reference's deepest well (D = 581)
WELLS = {1: {"soil": 0.0, "saprolite": 50.0, "weathered": 200.0, "max_depth": 500.0, "sat_depth": 100.0},
         200: synthetic_well(200), 300: synthetic_well(300),
         401: {"soil": 0.0, "saprolite": 50.0, "weathered": 200.0, "max_depth": 2000.0, "sat_depth": 125.0},   # well 10, the
         # one the reference's input_parameters.json selects
         581: synthetic_well(581)}      # = well 14 of the reference's site_information.json (max_depth 2 900 cm), its deepest


@lru_cache(maxsize=None)
def forcing_frame(n_years=1):
    return synthetic_forcing_frame(n_years)


@lru_cache(maxsize=None)
def digest(well, model="vrettas_fung", n_years=1):
    params = default_parameters()
    params["Hydrological_Model"]["Name"] = model
    cols = ColumnTables(params, WELLS[well])
    forcing = ForcingDigest(params, forcing_frame(n_years), cols)
    return params, cols, forcing


@lru_cache(maxsize=None)
def points():
    """Non-default parameter points of tests/golden/points.json (fixtures g1p/g2p/g34p, synthetic well D=200)."""
    import json
    with open(GOLDEN / "points.json") as fh:
        return json.load(fh)


@lru_cache(maxsize=None)
def digest_point(tag, model="vrettas_fung", well=200):
    params = default_parameters()
    params["Hydrological_Model"]["Name"] = model
    for section, values in points()[tag].items():
        params[section].update(values)
    cols = ColumnTables(params, WELLS[well])
    forcing = ForcingDigest(params, forcing_frame(1), cols)
    return params, cols, forcing


@lru_cache(maxsize=None)
def golden(name):
    return dict(np.load(GOLDEN / name))


def rel_err(a, b, floor=1.0):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return float(np.max(np.abs(a - b) / np.maximum(floor, np.abs(b)))) if a.size else 0.0
