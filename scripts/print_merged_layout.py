import ctypes as C, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from rscm_amd import _lib as L
from scripts.bench_magicc_chain import build_chain
lib = L.load()
for order in ("topological", "reference"):
    for spy in (1, 12):
        m = build_chain(256, 3, order, steps_per_year=spy)
        m.set_mode(L.MODE_FAST)
        m.run()
        out = (C.c_int32 * 27)()
        L.check(lib.rscm_gpu_lockstep_last_layout(out))
        v = list(out)
        print(order, spy, "ops", v[0], "first", v[1], "second", v[2], "kinds/off", [(v[3+2*k], v[4+2*k]) for k in range(v[0])], "order", m._order)
        m.close()
