"""Where the per-wave vectors of the step kernel live and what each moves per column-step, BY COUNT: bytes requested from the
wave's global region (what the L2 sees), from the layout rules of hc_step.h and the phase entry counts of a profile run
(profiles/r05_phases_d300.txt: 1.00 attempts, 1.00 Jacobians of 5 group evaluations, 3.73 factorisations, 12.22 Newton
iterations, 6.11 steps of which 0.18 raise the order, per column-step).  The counters (profiles/pmc_constants.json) give what
of that crosses the fabric.   python tools/traffic_by_count.py [> profiles/r05_traffic_by_count.txt]"""
LDS_BYTES, NTAB, SCRATCH, BOX = 160 * 1024, 9, 160 * 8, 272
E = dict(att=1.00, jac=1.00, grp=5.00, lu=3.73, nwt=12.22, stp=6.11, rise=0.18, tf=1.00)


def layout(cpl, halves=1):
    slots = 64 * cpl
    ntab = NTAB + (1 if slots * halves <= 320 else 0)        # + the reciprocal table where the cell model reads it (hc_device.h rdelta_table)
    tables = (ntab * slots * 8 + 4 * slots) * halves
    boxes = 4 * BOX if halves == 2 else 0
    per_wave = (LDS_BYTES - tables - boxes) // 8 - SCRATCH
    nf = 5 * cpl + 8 + (cpl + 1 if halves == 2 else 0)
    f_lds = max(0, min(nf, (per_wave - 3 * slots * 8) // 512))
    n_alias = min(4, f_lds // cpl)
    return dict(slots=slots, V=slots * 8, per_wave=per_wave, nf=nf, f_lds=f_lds, n_alias=n_alias)


def table(cpl, halves=1):
    L = layout(cpl, halves)
    V, a = L["V"], L["n_alias"]
    al = {"HJ": a >= 1, "JL": a >= 2, "JD": a >= 3, "JU": a >= 4}
    rows = []
    rows.append(("D[0..2] (difference rows of an order-1 step)", "LDS", 0.0))
    rows.append((f"factorisation, {L['f_lds']} of {L['nf']} lane-slots", "LDS", 0.0))
    rows.append((f"factorisation, the other {L['nf'] - L['f_lds']} lane-slots", "global",
                 (L["nf"] - L["f_lds"]) * 512 * (E["lu"] + E["nwt"])))
    for name in ("HJ", "JL", "JD", "JU"):
        if name == "HJ":
            jac = V * (E["jac"] + E["grp"] + E["jac"])            # store at the step sizes, load per group + at the finish
            keep = 0.0
        else:
            jac = V * (E["jac"] + E["jac"])                       # scatter (each entry once) + load at the finish
            keep = V * (E["jac"] + E["lu"])                       # finished row stored once, loaded by every factorisation
        where = "LDS (the dead factorisation's slots) while a Jacobian is evaluated" if al[name] else "global"
        label = {"HJ": "FD steps h_j", "JL": "Jacobian sub-diagonal", "JD": "Jacobian diagonal", "JU": "Jacobian super-diagonal"}[name]
        rows.append((label, where + ("; finished row: global" if name != "HJ" else ""), (0.0 if al[name] else jac) + keep))
    rows.append(("FD factors", "global", V * (E["att"] + 4 * E["jac"])))
    rows.append(("base f of the Jacobian", "global", V * (2 * E["att"] + 2 * E["jac"])))
    rows.append(("D[3] (stored when the order rises), D[4..7]", "global", V * 3 * E["rise"]))
    rows.append(("accepted state Y (the row's answer)", "global", V * (E["att"] + E["tf"] + 1.0)))
    rows.append(("noise vector", "global", V * (E["att"] + 1.0 / 48)))
    rows.append(("row-start state Y0", "global (spin-up stop rule only)", 0.0))
    return L, rows


def main():
    print(__doc__)
    for cpl, halves, what in ((4, 1, "D = 193..256"), (5, 1, "D = 257..320, the bench"), (6, 1, "D = 321..384"),
                              (7, 1, "D = 385..448, the reference's default well; default exponents only"),
                              (5, 2, "split column, D = 513..640: per HALF")):
        L, rows = table(cpl, halves)
        total = sum(r[2] for r in rows)
        print(f"\n== {cpl} cells per lane{' x 2 waves per member' if halves == 2 else ''} ({what}): vector = {L['V']} B, "
              f"{L['per_wave']} B of LDS per wave")
        for name, where, b in rows:
            print(f"  {name:52s} {where:78s} {b / 1000:7.1f} KB")
        print(f"  {'requests to the global region per column-step':52s} {'':78s} {total / 1000:7.1f} KB")


if __name__ == "__main__":
    main()
