"""Profiling driver: a few passes of the two-layer kernel through the C-ABI (no torch), for
rocprofv3 --kernel-trace --stats and --pmc runs.  Usage:
    python3 scripts/profile_two_layer.py [members] [mode 0|1] [passes] [kind 0|1|2|3]
kind 3 = GhgForcing; mode selects its method there (0 = IPCCTAR, 1 = OLBL).
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rscm_amd  # noqa: E402

members = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 0
passes = int(sys.argv[3]) if len(sys.argv) > 3 else 5
kind = int(sys.argv[4]) if len(sys.argv) > 4 else 0

t = np.arange(1750, 2501, dtype=np.float64)
b = np.append(t, 2501.0)
F = 4.0 * (1.0 - np.exp(-(t - 1750.0) / 120.0)) + 0.3 * np.sin(2 * np.pi * (t - 1750.0) / 11.0)
lo = np.array([0.8, 0.0, 1.0, 0.5, 5.0, 50.0, 15.0, 278.0, 0.0, 3.7])
hi = np.array([1.5, 0.1, 1.8, 1.0, 15.0, 200.0, 40.0, 278.0, 0.1, 3.7])
if kind == 2:  # ClimateUDEB: defaults, ECS / kappa / RLO / k_lo varied
    from rscm_amd import _lib
    lo = np.array(_lib.UD_DEFAULTS, dtype=float)
    hi = lo.copy()
    for name, (a_, b_) in dict(ecs=(2.0, 5.0), kappa=(0.5, 1.5), rlo=(1.2, 1.45), k_lo=(1.0, 2.0)).items():
        j = _lib.UD_PARAM_NAMES.index(name)
        lo[j], hi[j] = a_, b_
    if os.environ.get("UDEB_LAYERS"):  # any count >= 2 (<= 64: columns on chip; 20/30/40/50 with the count compiled in)
        j = _lib.UD_PARAM_NAMES.index("n_layers")
        lo[j] = hi[j] = float(os.environ["UDEB_LAYERS"])
    if os.environ.get("UDEB_NOFB"):  # constant ECS: no LAMCALC re-solve per year
        for name in ("feedback_q_sensitivity", "feedback_cumt_sensitivity"):
            j = _lib.UD_PARAM_NAMES.index(name)
            lo[j] = hi[j] = 0.0
if kind == 3:  # GhgForcing: pre-industrial values, CO2 sensitivity and adjustments varied
    from rscm_amd import _lib
    lo = np.array(_lib.GH_DEFAULTS, dtype=float)
    lo[0] = float(mode)
    hi = lo.copy()
    for name, (a_, b_) in dict(co2_pi=(275.0, 281.0), ch4_pi=(700.0, 740.0), n2o_pi=(265.0, 275.0), delq2xco2=(3.5, 4.0),
                               adjust_co2=(0.95, 1.1), adjust_ch4=(0.8, 0.95), adjust_n2o=(0.9, 1.05)).items():
        j = _lib.GH_PARAM_NAMES.index(name)
        lo[j], hi[j] = a_, b_
if kind == 11:  # OceanCarbon (3D-GFDL preset): gas exchange, temperature sensitivity, mixed layer varied
    from rscm_amd import _lib
    lo = np.array(_lib.OC_PRESETS["3D-GFDL"], dtype=float)
    hi = lo.copy()
    for name, (a_, b_) in dict(gas_exchange_tau=(6.0, 10.0), temp_sensitivity=(0.03, 0.045), mixed_layer_depth=(45.0, 60.0),
                               sst_pi=(16.0, 19.0)).items():
        j = _lib.OC_PARAM_NAMES.index(name)
        lo[j], hi[j] = a_, b_
P = {0: 6, 1: 10, 2: 37, 3: 21, 11: 24}[kind]
if not os.environ.get("PROFILE_CUT"):   # one launch = the whole axis: a summary describes ONE kernel launch (the cut runs' launches are chunks)
    from rscm_amd import _lib as _L
    _L.check(_L.load().rscm_gpu_set_run_plan(0))
with rscm_amd.Ensemble(kind, members, b) as e:
    if kind != 3:
        e.set_mode(mode)
    e.sample_lhs(20260327, lo[:P], hi[:P])
    if kind == 0:
        e.set_forcing(F)
        e.set_initial(1, 0.0)
        e.set_initial(2, 0.0)
    elif kind == 2:
        e.set_forcing(F)
        for v in (1, 2, 3, 4):
            e.set_initial(v, 0.0)
    elif kind == 11:
        yr = t - 1750.0
        e.set_forcing(np.stack([np.minimum(278.0 * 1.003 ** yr, 1100.0), np.minimum(0.006 * yr, 4.0)]))
        e.set_initial(1, 278.0)
        e.set_initial(2, 0.0)
    elif kind == 3:
        yr = t - 1750.0
        e.set_forcing(np.stack([278.0 * 1.0015 ** yr, 722.0 + 2.0 * yr, 270.0 + 0.1 * yr]))
    else:
        yrs = np.array([1750.0, 1850.0, 1950.0, 2000.0, 2020.0, 2050.0, 2100.0])
        e.set_forcing(np.interp(t, yrs, [0.0, 0.5, 3.0, 7.0, 10.0, 5.0, 1.0]))
        for v, x in ((1, 0.0), (2, 0.0), (3, 278.0), (4, 0.0), (5, 0.0)):
            e.set_initial(v, x)
    ms = []
    for _ in range(passes):
        e.rewind()
        e.run()
        ms.append(e.last_run_ms())
    print(f"kind={kind} members={members} mode={mode} launch ms: " + " ".join(f"{x:.3f}" for x in ms))
    print(f"member-years/s (best) = {members * 750 / (min(ms) * 1e-3):.4e}")
