"""Generate the build-supplied golden vectors under tests/golden/ from the CPU oracle.

The reference holds no numeric two-layer vector ("key gap") and cannot be
run here (Rust, no toolchain), so these fixtures are produced by oracle/rscm_oracle.c -- which
is itself pinned against the reference's known-answer tests in
tests/test_oracle_reference_goldens.py.  Run from the repo root:

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import cbind  # noqa: E402
from tests.helpers import (axis_values, coupled_params, emissions_syn, f_syn,  # noqa: E402
                           two_layer_params)

HERE = os.path.dirname(os.path.abspath(__file__))


def two_layer():
    t = axis_values(1750, 2500)
    b = cbind.bounds_from_values(t)
    P = two_layer_params(16, seed=20260327)
    # member 0 = configs/two-layer/defaults.toml, member 1 = tuning/high-ecs.toml-like linear case
    P[:, 0] = [1.0, 0.0, 1.0, 0.7, 8.0, 100.0]
    P[:, 1] = [0.8, 0.0, 1.3, 0.7, 8.0, 100.0]
    F = np.stack([f_syn(t), 0.5 * f_syn(t) + 0.2])
    scen = (np.arange(16) % 2).astype(np.int32)
    ts0 = np.linspace(0.0, 0.3, 16)
    td0 = np.linspace(0.0, -0.1, 16)
    out = {}
    for source in (0, 1):
        ts, td = cbind.two_layer_run(b, P, F, ts0, td0, scen=scen, source=source)
        out[f"ts_src{source}"] = ts
        out[f"td_src{source}"] = td
    np.savez_compressed(os.path.join(HERE, "two_layer_golden.npz"), time_values=t, params=P,
                        forcing=F, scen=scen, ts0=ts0, td0=td0, **out)


def coupled():
    t = axis_values(1750, 2100)  # configs/two-layer/defaults.toml [time]
    b = cbind.bounds_from_values(t)
    P = coupled_params(8, seed=20260327)
    # member 0 = docs/notebooks/coupled_model.py:360-384
    P[:, 0] = [1.1, 0.0, 1.3, 0.7, 8.0, 100.0, 25.0, 278.0, 0.1, 3.7]
    E = emissions_syn(t)
    init = dict(ts=0.0, td=0.0, conc=278.0, cum_uptake=0.0, cum_emis=0.0)
    out = cbind.coupled_run(b, P, E, init)
    np.savez_compressed(os.path.join(HERE, "coupled_golden.npz"), time_values=t, params=P,
                        emissions=E, **out)


if __name__ == "__main__":
    two_layer()
    coupled()
    for f in ("two_layer_golden.npz", "coupled_golden.npz"):
        print(f, os.path.getsize(os.path.join(HERE, f)), "bytes")
