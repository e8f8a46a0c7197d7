"""Extract the MAGICC7 reference outputs the reference's own regression tests hold for
ClimateUDEB (tests/regression/data/ocean_udeb/*.csv + *_config.json under the reference tree)
into one small JSON fixture: per scenario the MAGICC config keys the test maps onto
ClimateUDEBParameters (tests/regression/test_ocean_udeb.py:57-109) and the World
"Surface Temperature" row.  Data only.  Run in the build container:

    python tests/golden/make_udeb_goldens.py
"""
import csv
import glob
import json
import os

SRC = "/root/reference/tests/regression/data/ocean_udeb"
HERE = os.path.dirname(os.path.abspath(__file__))
KEEP = ("core_climatesensitivity", "core_delq2xco2", "core_initial_upwelling_rate",
        "core_upwelling_variable_part", "core_ocn_depthdependent", "core_verticaldiff_top_dkdt",
        "core_landheatcapacity_apply", "core_landhc_effthickness", "core_heatxchange_landground",
        "core_heatxchange_northsouth", "core_feedback_cumtsensitivity", "core_feedback_qsensitivity",
        "rf_efficacy_apply", "rf_efficacy_co2", "startyear", "endyear", "file_co2_conc")

out = {"_source": "reference tests/regression/data/ocean_udeb (MAGICC7 outputs), variable "
                  "'Surface Temperature', region World; tolerances in tests/regression/test_ocean_udeb.py"}
for f in sorted(glob.glob(os.path.join(SRC, "*_config.json"))):
    name = os.path.basename(f)[: -len("_config.json")]
    cfg = json.load(open(f))
    with open(os.path.join(SRC, name + ".csv")) as fh:
        rows = list(csv.reader(fh))
    hdr = rows[0]
    vi, ri = hdr.index("variable"), hdr.index("region")
    first = next(i for i, h in enumerate(hdr) if h[:2] in ("18", "19", "20", "21") and "-" in h)
    years = [int(h[:4]) for h in hdr[first:]]
    row = next(r for r in rows[1:] if r[vi] == "Surface Temperature" and r[ri] == "World")
    out[name] = {"config": {k: cfg[k] for k in KEEP if k in cfg}, "years": years,
                 "surface_temperature": [float(x) for x in row[first:]]}
json.dump(out, open(os.path.join(HERE, "udeb_magicc7.json"), "w"), separators=(",", ":"))
print(len(out) - 1, "scenarios,", os.path.getsize(os.path.join(HERE, "udeb_magicc7.json")), "bytes")
