"""Extract the MAGICC7 reference outputs the reference's own regression tests hold for GHG
forcing (tests/regression/data/ghg_forcing/*.csv + *_config.json under the reference tree) into
one JSON fixture: per scenario the MAGICC config keys tests/regression/test_ghg_forcing.py reads
and the World rows of the variables it compares (concentrations in, ERF|CO2/CH4/N2O, total ERF and
Surface Temperature out).  Data only.  Run in the build container:

    python tests/golden/make_ghg_goldens.py
"""
import csv
import glob
import json
import os

SRC = "/root/reference/tests/regression/data/ghg_forcing"
HERE = os.path.dirname(os.path.abspath(__file__))
KEEP = ("core_co2ch4n2o_rfmethod", "core_climatesensitivity", "core_delq2xco2", "core_rfrapidadjust_co2",
        "core_rfrapidadjust_ch4", "core_rfrapidadjust_n2o", "rf_total_runmodus", "startyear", "endyear")
VARS = ("Atmospheric Concentrations|CO2", "Atmospheric Concentrations|CH4", "Atmospheric Concentrations|N2O",
        "Effective Radiative Forcing|CO2", "Effective Radiative Forcing|CH4", "Effective Radiative Forcing|N2O",
        "Effective Radiative Forcing", "Surface Temperature")
SKIP = ("03_emissions_driven",)  # the reference marks its own test of this scenario xfail

out = {"_source": "reference tests/regression/data/ghg_forcing (MAGICC7 outputs), region World; "
                  "tolerances in tests/regression/test_ghg_forcing.py (ERF rtol 1e-5 atol 1e-6; "
                  "temperature phased 5e-2/3e-2/3e-2)"}
for f in sorted(glob.glob(os.path.join(SRC, "*_config.json"))):
    name = os.path.basename(f)[: -len("_config.json")]
    if name in SKIP:
        continue
    cfg = json.load(open(f))
    with open(os.path.join(SRC, name + ".csv")) as fh:
        rows = list(csv.reader(fh))
    hdr = rows[0]
    vi, ri = hdr.index("variable"), hdr.index("region")
    first = next(i for i, h in enumerate(hdr) if h[:2] in ("17", "18", "19", "20", "21") and "-" in h)
    entry = {"config": {k: cfg[k] for k in KEEP if k in cfg}, "years": [int(h[:4]) for h in hdr[first:]]}
    for v in VARS:
        row = [r for r in rows[1:] if r[vi] == v and r[ri] == "World"]
        if row:
            entry[v] = [float(x) for x in row[0][first:]]
    out[name] = entry
json.dump(out, open(os.path.join(HERE, "ghg_forcing_magicc7.json"), "w"), separators=(",", ":"))
print(len(out) - 1, "scenarios,", os.path.getsize(os.path.join(HERE, "ghg_forcing_magicc7.json")), "bytes")
