import sys, numpy as np
sys.path.insert(0, ".")
import rscm_amd
t = np.arange(1750.0, 2501.0); b = np.append(t, 2501.0)
F = 4.0 * (1.0 - np.exp(-(t - 1750.0) / 120.0)) + 0.3 * np.sin(2 * np.pi * (t - 1750.0) / 11.0)
lo = np.array([0.8, 0.0, 1.0, 0.5, 5.0, 50.0]); hi = np.array([1.5, 0.1, 1.8, 1.0, 15.0, 200.0])
for kind, n in ((rscm_amd.KIND_TWO_LAYER, 100_000), (rscm_amd.KIND_TWO_LAYER, 150_001)):
    with rscm_amd.Ensemble(kind, n, b) as e:
        e.sample_lhs(7, lo, hi); e.set_forcing(F); e.set_initial(1, 0.0); e.set_initial(2, 0.0)
        ref = None
        for k in range(40):
            e.rewind(); e.run(sync=False); e.sync()
            rows = np.stack([e.get_series(1, r, r + 1)[0] for r in (63, 64, 65, 375, 750)])
            if ref is None: ref = rows; print(n, e.last_run_plan())
            assert np.array_equal(rows.view(np.uint64), ref.view(np.uint64)), k
print("stable")
