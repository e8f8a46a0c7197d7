"""ctypes loader for ``librscm_oracle.so`` (the C restatement).  TEST INFRASTRUCTURE ONLY.

Builds the library with ``make -C oracle`` when it is missing (gcc is in the image).  Nothing in
the product package imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "librscm_oracle.so")
    src = os.path.join(_HERE, "rscm_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.run(["make", "-C", _HERE, "-B", "librscm_oracle.so"], check=True,
                       capture_output=True)
    return so


def lib() -> C.CDLL:
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        L.orc_bounds_from_values.argtypes = [_dp, C.c_int32, _dp]
        L.orc_rk4_nsteps.argtypes = [C.c_double] * 3
        L.orc_rk4_nsteps.restype = C.c_int32
        L.orc_rk4_endtime_ok.argtypes = [C.c_double] * 3
        L.orc_two_layer_solve.argtypes = [_dp] + [C.c_double] * 4 + [_dp, _dp, _dp]
        L.orc_two_layer_solve.restype = None
        L.orc_two_layer_run.argtypes = [C.c_int64, C.c_int32, _dp, _dp, C.c_int32, _dp, _ip,
                                        C.c_int, C.c_double, C.c_int32, C.c_int32, _dp, _dp,
                                        C.c_int64, C.c_int64]
        L.orc_carbon_cycle_solve.argtypes = [_dp] + [C.c_double] * 5 + [_dp]
        L.orc_carbon_cycle_solve.restype = None
        L.orc_co2_erf.argtypes = [C.c_double] * 3
        L.orc_co2_erf.restype = C.c_double
        L.orc_aggregate_sum.argtypes = [_dp, C.c_int32]
        L.orc_aggregate_sum.restype = C.c_double
        L.orc_coupled_run.argtypes = [C.c_int64, C.c_int32, _dp, _dp, C.c_int32, _dp, _ip,
                                      C.c_double, C.c_double, C.c_int32, C.c_int32] + [_dp] * 7 + \
                                     [C.c_int64, C.c_int64]
        L.orc_gaussian_loglik.argtypes = [C.c_int64, C.c_int32, C.POINTER(_dp), C.c_int32, _ip, _ip,
                                          _dp, _dp, C.c_int, _dp, C.c_int64, C.c_int64]
        _LIB = L
    return _LIB


def _d(a):
    assert a.dtype == np.float64 and a.flags.c_contiguous
    return a.ctypes.data_as(_dp)


def _i(a):
    if a is None:
        return None
    assert a.dtype == np.int32 and a.flags.c_contiguous
    return a.ctypes.data_as(_ip)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _split(n, threads):
    threads = max(1, min(threads, n))
    edges = np.linspace(0, n, threads + 1).astype(np.int64)
    return [(int(edges[k]), int(edges[k + 1])) for k in range(threads) if edges[k + 1] > edges[k]]


def _pmap(fn, n, threads):
    chunks = _split(n, threads)
    if len(chunks) == 1:
        rcs = [fn(*chunks[0])]
    else:  # ctypes releases the GIL during the foreign call
        with ThreadPoolExecutor(len(chunks)) as ex:
            rcs = list(ex.map(lambda c: fn(*c), chunks))
    if any(rcs):
        raise RuntimeError(f"oracle returned {rcs}")


def bounds_from_values(values):
    v = _f64(values)
    b = np.empty(len(v) + 1)
    rc = lib().orc_bounds_from_values(_d(v), len(v), _d(b))
    if rc:
        raise ValueError(f"orc_bounds_from_values rc={rc}")
    return b


def rk4_nsteps(t0, t1, h):
    return lib().orc_rk4_nsteps(t0, t1, h)


def rk4_endtime_ok(t0, t1, h):
    return bool(lib().orc_rk4_endtime_ok(t0, t1, h))


def two_layer_solve(params, erf, t0, t1, h, ts, td):
    p = _f64(params)
    a, b, c = C.c_double(ts), C.c_double(td), C.c_double(0.0)
    lib().orc_two_layer_solve(_d(p), erf, t0, t1, h, C.byref(a), C.byref(b), C.byref(c))
    return a.value, b.value, c.value


def two_layer_run(bounds, params, forcing, ts0, td0, *, scen=None, source=0, h=0.1,
                  step_begin=0, step_end=None, threads=1, ts=None, td=None):
    """Returns (Ts[T][N], Td[T][N]); rows beyond the last executed step stay NaN."""
    bounds = _f64(bounds)
    T = len(bounds) - 1
    params = _f64(params)
    N = params.shape[1]
    forcing = np.atleast_2d(_f64(forcing))
    assert forcing.shape[1] == T and params.shape[0] == 6
    if scen is not None:
        scen = np.ascontiguousarray(scen, dtype=np.int32)
    if ts is None:
        ts = np.full((T, N), np.nan)
        td = np.full((T, N), np.nan)
        ts[0, :] = ts0
        td[0, :] = td0
    step_end = T - 1 if step_end is None else step_end
    L = lib()
    _pmap(lambda i0, i1: L.orc_two_layer_run(N, T, _d(bounds), _d(params), forcing.shape[0],
                                             _d(forcing), _i(scen), source, h, step_begin,
                                             step_end, _d(ts), _d(td), i0, i1), N, threads)
    return ts, td


COUPLED_VARS = ("ts", "td", "conc", "cum_uptake", "cum_emis", "erf_co2", "erf_total")


def coupled_run(bounds, params, emissions, init, *, scen=None, h_tl=0.1, h_cc=0.1, step_begin=0,
                step_end=None, threads=1):
    """init: dict ts, td, conc, cum_uptake, cum_emis -> scalar or [N].  Returns dict of [T][N]."""
    bounds = _f64(bounds)
    T = len(bounds) - 1
    params = _f64(params)
    N = params.shape[1]
    emissions = np.atleast_2d(_f64(emissions))
    assert emissions.shape[1] == T and params.shape[0] == 10
    if scen is not None:
        scen = np.ascontiguousarray(scen, dtype=np.int32)
    out = {k: np.full((T, N), np.nan) for k in COUPLED_VARS}
    for k, v in init.items():
        out[k][0, :] = v
    step_end = T - 1 if step_end is None else step_end
    L = lib()
    _pmap(lambda i0, i1: L.orc_coupled_run(N, T, _d(bounds), _d(params), emissions.shape[0],
                                           _d(emissions), _i(scen), h_tl, h_cc, step_begin,
                                           step_end, *[_d(out[k]) for k in COUPLED_VARS], i0, i1),
          N, threads)
    return out


def carbon_cycle_solve(params, emissions, temperature, t0, t1, h, y):
    p = _f64(params)
    yy = _f64(y).copy()
    lib().orc_carbon_cycle_solve(_d(p), emissions, temperature, t0, t1, h, _d(yy))
    return yy


def co2_erf(erf_2xco2, conc_pi, concentration):
    return lib().orc_co2_erf(erf_2xco2, conc_pi, concentration)


def aggregate_sum(values):
    v = _f64(values)
    return lib().orc_aggregate_sum(_d(v), len(v))


def gaussian_loglik(series, obs_series, obs_tidx, obs_value, obs_sigma, normalize=False,
                    threads=1):
    series = [_f64(s) for s in series]
    T, N = series[0].shape
    ptrs = (_dp * len(series))(*[_d(s) for s in series])
    os_ = np.ascontiguousarray(obs_series, dtype=np.int32)
    ot = np.ascontiguousarray(obs_tidx, dtype=np.int32)
    ov, sg = _f64(obs_value), _f64(obs_sigma)
    out = np.empty(N)
    L = lib()
    _pmap(lambda i0, i1: L.orc_gaussian_loglik(N, T, ptrs, len(ot), _i(os_), _i(ot), _d(ov),
                                               _d(sg), int(normalize), _d(out), i0, i1),
          N, threads)
    return out
