"""Plume statistics on the device (run on the GPU box): rscm_ens_summary_series and
rscm_ens_quantile_series over all 751 rows of a two-layer ensemble, against moving the series to the
host and calling numpy."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rscm_amd as ra  # noqa: E402
from tests.helpers import axis_values, f_syn, two_layer_params  # noqa: E402

t = axis_values()
b = np.append(t, 2501.0)
q = [0.05, 0.17, 0.5, 0.83, 0.95]
for n in (100_000, 1_000_000):
    with ra.Ensemble(ra.KIND_TWO_LAYER, n, b) as e:
        e.set_params(two_layer_params(n))
        e.set_forcing(f_syn(t))
        e.set_initial(1, 0.0)
        e.set_initial(2, 0.0)
        e.run()
        e.quantile_series(1, q, 0, 8)
        t0 = time.perf_counter(); s = e.summary_series(1); t_sum = time.perf_counter() - t0
        t0 = time.perf_counter(); g = e.quantile_series(1, q); t_q = time.perf_counter() - t0
        line = f"N={n}: summary_series {t_sum*1e3:.1f} ms, quantile_series (5 quantiles x 751 rows) {t_q*1e3:.1f} ms"
        if n <= 100_000:
            t0 = time.perf_counter(); ts = e.get_series(1); t_copy = time.perf_counter() - t0
            t0 = time.perf_counter()
            with np.errstate(all="ignore"):
                w = np.nanquantile(ts, q, axis=1).T
            t_np = time.perf_counter() - t0
            line += f"; host: D2H {t_copy*1e3:.0f} ms + numpy.nanquantile {t_np*1e3:.0f} ms; same bits: {np.array_equal(w, g['quantiles'], equal_nan=True)}"
        print(line + f"; median warming in 2500: {g['quantiles'][-1][2]:.3f} K", flush=True)
