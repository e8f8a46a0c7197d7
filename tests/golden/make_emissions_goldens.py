"""Extract the emissions-driven MAGICC7 scenario the reference's regression suite holds
(tests/regression/data/ghg_forcing/03_emissions_driven.csv + _config.json under the reference tree:
SSP245 emissions in, concentrations / forcing / temperature out) into one JSON fixture: the World
rows, years 1750-2100.  Data only.  Run in the build container:

    python tests/golden/make_emissions_goldens.py
"""
import csv
import json
import os

SRC = "/root/reference/tests/regression/data/ghg_forcing"
HERE = os.path.dirname(os.path.abspath(__file__))
name = "03_emissions_driven"
cfg = json.load(open(os.path.join(SRC, name + "_config.json")))
with open(os.path.join(SRC, name + ".csv")) as fh:
    rows = list(csv.reader(fh))
hdr = rows[0]
vi, ri, ui = hdr.index("variable"), hdr.index("region"), hdr.index("unit")
first = next(i for i, h in enumerate(hdr) if h[:2] in ("17", "18", "19", "20", "21") and "-" in h)
out = {"_source": "reference tests/regression/data/ghg_forcing/03_emissions_driven (MAGICC7, SSP245, region World); the "
                  "reference's own test of it (test_ghg_forcing.py::test_03_emissions_driven, rtol 5e-2) is xfail upstream",
       "config": cfg, "years": [int(h[:4]) for h in hdr[first:]], "units": {}, "variables": {}}
for r in rows[1:]:
    if r[ri] == "World":
        out["variables"][r[vi]] = [float(x) for x in r[first:]]
        out["units"][r[vi]] = r[ui]
path = os.path.join(HERE, "magicc7_emissions_driven.json")
json.dump(out, open(path, "w"), separators=(",", ":"))
print(len(out["variables"]), "variables,", len(out["years"]), "years,", os.path.getsize(path), "bytes")
