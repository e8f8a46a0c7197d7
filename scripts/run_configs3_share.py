"""BASELINE.json configs[3], one GPU's share at full size (run on the GPU box): the emissions-driven MAGICC
graph (ten rscm-magicc components, Sum of eight forcings, FourBox transforms), 125 000 members (1e6 / 8 GPUs),
MONTHLY model steps 1750-2500 (9001 points, 9000 steps), ClimateUDEB and OceanCarbon at their 12 sub-steps per
model step, series in a window of --window rows (default 96: 3.5 GB of the 288; the window slides once every ~80 steps) with annual
(every 12th) rows of every variable kept.

    python scripts/run_configs3_share.py [--members 125000] [--years 750] [--exact]

Prints one JSON line: build / run wall time, HBM allocated, launches, member-years/s (a member-year = 12 model
steps here), ensemble statistics at the end, and -- the parity anchor at this size -- whether the first 64
members equal a 64-member run of the same graph, bit for bit, on every kept row of five variables."""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rscm_amd import _lib as L  # noqa: E402
from scripts.bench_magicc_chain import build_chain  # noqa: E402

NAMES = ["Sea Surface Temperature", "Atmospheric Concentration|CO2", "Effective Radiative Forcing", "Atmospheric Concentration|CH4",
         "Carbon Flux|Ocean"]


def build(members, years, exact, window=16, device=0, member_offset=0, members_total=None):
    """The share's graph, built and in its arithmetic mode, ready to run (bench.py's N > 1 leg times `model.run()` itself)."""
    model = build_chain(members, years, "topological", steps_per_year=12, device=device, member_offset=member_offset,
                        members_total=members_total, series_window=window, output_stride=12)
    if not exact:
        model.set_mode(L.MODE_FAST)
    return model


def run(members, years, exact, window=16, device=0, member_offset=0, members_total=None):
    free0, total = L.mem_info(device)
    t0 = time.perf_counter()
    model = build_chain(members, years, "topological", steps_per_year=12, device=device, member_offset=member_offset,
                        members_total=members_total, series_window=window, output_stride=12)
    if not exact:
        model.set_mode(L.MODE_FAST)
    build_s = time.perf_counter() - t0
    free1, _ = L.mem_info(device)
    L.check(L.load().rscm_gpu_lockstep_stats(None, None))
    t0 = time.perf_counter()
    model.run()
    run_s = time.perf_counter() - t0
    nl, ns = C.c_int64(), C.c_int64()
    L.check(L.load().rscm_gpu_lockstep_stats(C.byref(nl), C.byref(ns)))
    rows = {n: model.get_series(n, t_stride=12) for n in NAMES}
    T = years * 12 + 1
    warm = model.ensembles["Transform:Surface Temperature"].summary(1, T - 1)
    co2 = model.ensembles["CO2Budget"].summary(1, T - 1)
    status = int(model.ensembles["ClimateUDEB"].status().sum())
    model.close()
    return dict(build_s=build_s, run_s=run_s, hbm_gib=(free0 - free1) / 2**30, hbm_total_gib=total / 2**30, launches=int(nl.value),
                component_steps=int(ns.value), warm=warm, co2=co2, failed=status), rows


def first_64(members, years, exact, window=16, device=0, member_offset=0, members_total=None):
    """Parity anchor at this size: a 64-member ensemble that is GIVEN the first 64 members' parameters of the
    `members`-member one (build_chain draws whole vectors from one seeded generator: replayed here for the big
    ensemble's draws).  Returns the kept (annual) rows of NAMES."""
    import scripts.bench_magicc_chain as mod
    small = build_chain(64, years, "topological", steps_per_year=12, device=device, series_window=window, output_stride=12)
    rng = np.random.default_rng(20260327)
    total = members if members_total is None else members_total
    first = slice(member_offset, member_offset + 64)
    ecs = rng.uniform(2.0, 4.5, total)
    kappa = rng.uniform(0.5, 1.2, total)
    beta_f = rng.uniform(0.7, 1.3, total)
    ud = small.ensembles["ClimateUDEB"]
    P = ud.get_params()
    P[L.UD_PARAM_NAMES.index("ecs")] = ecs[first]
    P[L.UD_PARAM_NAMES.index("kappa")] = kappa[first]
    ud.set_params(P)
    tc = small.ensembles["TerrestrialCarbon"]
    Q = tc.get_params()
    base_beta = mod.chain_components()[7].param_vector()[L.TC_PARAM_NAMES.index("beta")]
    Q[L.TC_PARAM_NAMES.index("beta")] = base_beta * beta_f[first]
    tc.set_params(Q)
    if not exact:
        small.set_mode(L.MODE_FAST)
    small.run()
    rows = {n: small.get_series(n, t_stride=12) for n in NAMES}
    small.close()
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--members", type=int, default=125_000)
    ap.add_argument("--years", type=int, default=750)
    ap.add_argument("--exact", action="store_true", help="RSCM_MODE_EXACT: OceanCarbon's literal O(T^2) convolution")
    ap.add_argument("--window", type=int, default=96, help="rows of every series kept on the device (36 MB per row at 125 000 members)")
    ap.add_argument("--fusion", type=int, default=1, help="rscm_gpu_set_lockstep_fusion mode 0..3 (include/rscm_gpu_internal.h)")
    ap.add_argument("--no-anchor", action="store_true",
                    help="skip the 64-member parity anchor (profiling passes: half the dispatches; the anchor is checked by every un-profiled run)")
    args = ap.parse_args()
    L.check(L.load().rscm_gpu_set_lockstep_fusion(args.fusion))
    big, rows = run(args.members, args.years, args.exact, args.window)
    small_rows = {n: rows[n][:, :64] for n in NAMES} if args.no_anchor else first_64(args.members, args.years, args.exact, args.window)
    same = {}
    for n in NAMES:
        a, b = rows[n][:, :64], small_rows[n]
        same[n] = bool(np.array_equal(a.view(np.uint64), b.view(np.uint64)) or np.array_equal(a, b, equal_nan=True))
    steps = args.years * 12
    out = {"workload": f"BASELINE configs[3], one GPU's share: MAGICC graph, {args.members} members x {steps} monthly steps "
                       f"({args.years} years), window {args.window} rows + annual outputs of all 36 series, mode {'EXACT' if args.exact else 'FAST'}",
           "run_s": big["run_s"], "build_s": big["build_s"], "member_years_per_s": args.members * args.years / big["run_s"],
           "member_model_steps_per_s": args.members * steps / big["run_s"], "ms_per_model_step": big["run_s"] / steps * 1e3,
           "launches": big["launches"], "launches_per_step": big["launches"] / steps, "hbm_allocated_gib": big["hbm_gib"],
           "hbm_total_gib": big["hbm_total_gib"], "failed_members": big["failed"],
           "warming_end_K": big["warm"], "co2_end_ppm": big["co2"],
           "first_64_members_equal_a_64_member_run": same}
    print(json.dumps(out))
    sys.exit(0 if all(same.values()) and big["failed"] == 0 else 1)


if __name__ == "__main__":
    main()
