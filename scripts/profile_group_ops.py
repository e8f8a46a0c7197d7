"""Which component costs what inside the fused lock-step launch (run on the GPU box, alone or under
`rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES`): the coupled chain as linked ensembles with components left
out one at a time, every variant one multi-step `group_kernel` launch.  The differences between the
variants are the per-component costs, overheads included.

    python scripts/profile_group_ops.py [members]"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rscm_amd as ra  # noqa: E402
from rscm_amd import _lib as L  # noqa: E402
from rscm_amd.ensemble import run_lockstep  # noqa: E402
from tests.helpers import axis_values, coupled_params, emissions_syn  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
t = axis_values(1750, 2500)
b = np.append(t, t[-1] + 1.0)
E = emissions_syn(t)
F = 4.0 * (1.0 - np.exp(-(t - 1750.0) / 120.0))
CONC = 278.0 + 0.5 * (t - 1750.0)
P = coupled_params(N)
stream = C.c_void_p()
L.check(L.load().rscm_gpu_stream_create(0, C.byref(stream)))
# RSCM_GROUP_MODE: rscm_gpu_set_lockstep_fusion's argument (1: one launch with LDS slots, the default; 2: without them)
L.check(L.load().rscm_gpu_set_lockstep_fusion(int(os.environ.get("RSCM_GROUP_MODE", "1"))))


def variant(name):
    cc = ce = ag = tl = None
    members = []
    if "cc" in name:
        cc = ra.Ensemble(ra.KIND_CARBON_CYCLE, N, b); cc.set_stream(stream.value); cc.set_params(P[[6, 7, 8]])
        cc.set_forcing(np.stack([E, np.zeros(len(t))]))
        for var, v in (("Atmospheric Concentration|CO2", 278.0), ("Cumulative Land Uptake", 0.0), ("Cumulative Emissions|CO2", 0.0)):
            cc.set_initial(var, v)
        members.append(cc)
    if "ce" in name:
        ce = ra.Ensemble(ra.KIND_CO2_ERF, N, b); ce.set_stream(stream.value); ce.set_params(P[[9, 7]])
        if cc is not None: ce.link_input(0, cc, 1, ra.SRC_UPSTREAM)
        else: ce.set_forcing(CONC)
        members.append(ce)
    n_ag = name.count("ag")
    ags = []
    prev = ce
    for k in range(n_ag):
        ag = ra.Ensemble(ra.KIND_AGGREGATE, N, b); ag.set_stream(stream.value); ag.set_params(np.zeros((9, N)))
        if prev is not None: ag.link_input(0, prev, 1, ra.SRC_UPSTREAM)
        else: ag.set_forcing(np.stack([F] + [np.full(len(t), np.nan)] * 7))
        ags.append(ag); members.append(ag); prev = ag
    if "tl" in name:
        tl = ra.Ensemble(ra.KIND_TWO_LAYER, N, b); tl.set_stream(stream.value); tl.set_params(P[:6])
        tl.set_initial(1, 0.0); tl.set_initial(2, 0.0)
        if prev is not None: tl.link_input(0, prev, 1, ra.SRC_UPSTREAM)
        else: tl.set_forcing(F)
        if cc is not None: cc.link_input(1, tl, 1, ra.SRC_EXOGENOUS)
        members.append(tl)
    if len(members) < 2:  # a single handle keeps its own kernel: pair it with a trivial aggregate
        raise SystemExit("variants need two components")
    best = 1e9
    for rep in range(3):
        for e in members: e.rewind()
        members[-1].sync()
        t0 = time.perf_counter()
        run_lockstep(tuple(members))
        best = min(best, time.perf_counter() - t0)
    print(f"{name:>16}: {best*1e3:7.2f} ms  ({len(members)} components)", flush=True)
    if cc is not None and tl is not None: cc.unlink_input(1)
    for e in reversed(members): e.close()


for name in ("cc ce ag tl", "cc ce tl", "ce ag tl", "ag tl", "ag ag tl", "ag ag ag tl", "cc ce", "ce ag"):
    variant(name)
L.check(L.load().rscm_gpu_stream_destroy(0, stream))
