"""Time the exact two-layer kernel with librscm_gpu built at several workgroup sizes (one-off
tuning experiment; variants are built by hand with -DRSCM_BLOCK=N)."""
import ctypes as C, os, sys
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
t = np.arange(1750, 2501, dtype=np.float64); b = np.append(t, 2501.0)
F = 4.0 * (1.0 - np.exp(-(t - 1750.0) / 120.0)) + 0.3 * np.sin(2 * np.pi * (t - 1750.0) / 11.0)
lo = np.array([0.8, 0.0, 1.0, 0.5, 5.0, 50.0]); hi = np.array([1.5, 0.1, 1.8, 1.0, 15.0, 200.0])
dp = C.POINTER(C.c_double)
for tag in ("b64", "b128", "", "b512"):
    path = os.path.join(root, "rscm_amd", f"librscm_gpu{'_' + tag if tag else ''}.so")
    if not os.path.exists(path):
        continue
    lib = C.CDLL(path)
    for n in (100_000, 1_000_000):
        for mode in (0, 1):
            h = C.c_void_p()
            assert lib.rscm_ens_create(0, C.c_int64(n), 751, b.ctypes.data_as(dp), 0, C.byref(h)) == 0
            lib.rscm_ens_set_mode(h, mode)
            lib.rscm_ens_sample_lhs(h, C.c_uint64(1), lo.ctypes.data_as(dp), hi.ctypes.data_as(dp), C.c_int64(0), C.c_int64(n))
            lib.rscm_ens_set_forcing(h, 0, 1, F.ctypes.data_as(dp), None, 0)
            z = (C.c_double * 1)(0.0)
            lib.rscm_ens_set_initial(h, 1, z, C.c_int64(1)); lib.rscm_ens_set_initial(h, 2, z, C.c_int64(1))
            ms = []
            for _ in range(6):
                lib.rscm_ens_rewind(h); assert lib.rscm_ens_run(h, 0, 750) == 0
                f = C.c_float(); lib.rscm_ens_last_run_ms(h, C.byref(f)); ms.append(f.value)
            lib.rscm_ens_destroy(h)
            print(f"block={tag or 'b256':5s} n={n:8d} mode={mode} best={min(ms[1:]):.3f} ms median={sorted(ms[1:])[2]:.3f} ms")
