"""Measure the host-copy rates behind the PCIe-inclusive note in DESIGN.md (run on the GPU box)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rscm_amd
t = np.arange(1750, 2501, dtype=np.float64); b = np.append(t, 2501.0)
n = 100_000
F = 4.0 * (1.0 - np.exp(-(t - 1750.0) / 120.0))
lo = np.array([0.8, 0.0, 1.0, 0.5, 5.0, 50.0]); hi = np.array([1.5, 0.1, 1.8, 1.0, 15.0, 200.0])
with rscm_amd.Ensemble(0, n, b) as e:
    e.sample_lhs(1, lo, hi); e.set_forcing(F); e.set_initial(1, 0.0); e.set_initial(2, 0.0)
    e.run()
    for rep in range(3):
        t0 = time.perf_counter(); ts = e.get_series(1); dt = time.perf_counter() - t0
        print(f"get_series full [751][{n}] = {ts.nbytes/1e6:.0f} MB in {dt*1e3:.1f} ms = {ts.nbytes/dt/1e9:.1f} GB/s")
    from rscm_amd.ensemble import pinned_empty
    buf = pinned_empty((751, n))
    for rep in range(3):
        t0 = time.perf_counter(); e.get_series(1, out=buf); dt = time.perf_counter() - t0
        print(f"get_series into pinned buffer = {buf.nbytes/1e6:.0f} MB in {dt*1e3:.1f} ms = {buf.nbytes/dt/1e9:.1f} GB/s")
    assert np.array_equal(np.asarray(buf), ts, equal_nan=True)
    warm = np.empty((751, n)); warm[:] = 0.0
    t0 = time.perf_counter(); e.get_series(1, out=warm); dt = time.perf_counter() - t0
    print(f"get_series into pre-touched pageable buffer = {warm.nbytes/dt/1e9:.1f} GB/s")
    t0 = time.perf_counter(); s = e.get_series(1, 0, 751, 10); dt = time.perf_counter() - t0
    print(f"get_series stride 10 = {s.nbytes/1e6:.0f} MB in {dt*1e3:.1f} ms = {s.nbytes/dt/1e9:.1f} GB/s")
    P = e.get_params()
    t0 = time.perf_counter(); e.set_params(P); dt = time.perf_counter() - t0
    print(f"set_params {P.nbytes/1e6:.1f} MB in {dt*1e3:.2f} ms = {P.nbytes/dt/1e9:.1f} GB/s")
    t0 = time.perf_counter(); e.set_params_aos(np.ascontiguousarray(P.T)); dt = time.perf_counter() - t0
    print(f"set_params_aos (host transpose + H2D) in {dt*1e3:.2f} ms")
    tid = np.arange(100, 271, 10, dtype=np.int32)
    t0 = time.perf_counter(); ll = e.loglik(np.ones(len(tid), int), tid, np.ones(len(tid)), np.full(len(tid), 0.1)); dt = time.perf_counter() - t0
    print(f"loglik {len(tid)} obs x {n} members incl. D2H in {dt*1e3:.2f} ms")
