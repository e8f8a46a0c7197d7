"""The execution order of the emissions-driven MAGICC graph (scripts/bench_magicc_chain.py, topological) and the kinds behind the names (run on the GPU box)."""
import sys; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scripts.bench_magicc_chain import build_chain
m = build_chain(64, 2, "topological", steps_per_year=12, series_window=96, output_stride=12)
print(m._order)
for name in m._order:
    e = m.ensembles[name]
    print(name, e.kind)
print(m._links if hasattr(m, "_links") else None)
m.close()
