"""A/B of two builds of the library on the ClimateUDEB whole-axis launch (65 536 members x 750 years, six launches):
`base` loads rscm_amd/librscm_gpu.so, `variant` a second build copied to rscm_amd/librscm_gpu_variant.so.
    for k in 1 2 3; do python scripts/ab_udeb_variant.py base; python scripts/ab_udeb_variant.py variant; done
Round 3 used it for the column solve with the refined reciprocal formed once per row (one f64 instruction less per row, 1887 instead
of 1983 per sub-step): 54.0-54.4 ms against 53.9-54.1 ms -- no gain, not kept."""
import os, sys
sys.path.insert(0, os.getcwd())
from rscm_amd import _lib
if sys.argv[1] != "base":
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.__file__), "librscm_gpu_" + sys.argv[1] + ".so")
import bench
e = bench.make_udeb_ensemble(65536, 0)
ms = []
for _ in range(6):
    e.rewind(); e.run(); ms.append(e.last_run_ms())
print(sys.argv[1], " ".join(f"{m:.2f}" for m in ms))
