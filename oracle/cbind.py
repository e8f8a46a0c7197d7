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
    srcs = [os.path.join(_HERE, f) for f in ("rscm_oracle.c", "udeb_oracle.c", "ghg_oracle.c", "forcing_oracle.c", "chem_oracle.c", "carbon_oracle.c", "ocean_oracle.c", "halocarbon_oracle.c")]
    if force or not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(f) for f in srcs):
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
        L.orc_udeb_n_params.restype = C.c_int32
        L.orc_udeb_default_params.argtypes = [_dp]
        L.orc_udeb_default_params.restype = None
        L.orc_udeb_run.argtypes = [C.c_int64, C.c_int32, _dp, _dp, C.c_int32, _dp, _ip, _dp] + [_dp] * 7 + \
                                  [_ip, C.c_int64, C.c_int64]
        L.orc_ghg_n_params.restype = C.c_int32
        L.orc_ghg_default_params.argtypes = [_dp]
        L.orc_ghg_default_params.restype = None
        L.orc_ghg_forcings.argtypes = [_dp, C.c_double, C.c_double, C.c_double, _dp]
        L.orc_ghg_forcings.restype = None
        L.orc_ghg_run.argtypes = [C.c_int64, C.c_int32, _dp, C.c_int32, _dp, _ip, _dp, _dp, _dp,
                                  C.c_int64, C.c_int64]
        L.orc_ghg_run.restype = None
        for f in ("n_params", "n_inputs", "n_outputs"):
            getattr(L, "orc_pointwise_" + f).argtypes = [C.c_int32]
            getattr(L, "orc_pointwise_" + f).restype = C.c_int32
        L.orc_pointwise_default_params.argtypes = [C.c_int32, _dp]
        L.orc_pointwise_default_params.restype = None
        L.orc_pointwise_eval.argtypes = [C.c_int32, _dp, _dp, _dp]
        L.orc_pointwise_eval.restype = C.c_int32
        L.orc_pointwise_run.argtypes = [C.c_int32, C.c_int64, C.c_int32, _dp, _dp, _ip, _dp, C.c_int64, C.c_int64]
        L.orc_pointwise_run.restype = C.c_int32
        for f in ("n_params", "n_inputs"):
            getattr(L, "orc_chem_" + f).argtypes = [C.c_int32]
            getattr(L, "orc_chem_" + f).restype = C.c_int32
        L.orc_chem_default_params.argtypes = [C.c_int32, _dp]
        L.orc_chem_default_params.restype = None
        L.orc_ch4_solve_concentration.argtypes = [_dp] + [C.c_double] * 7 + [_dp]
        L.orc_ch4_solve_concentration.restype = None
        L.orc_n2o_solve_concentration.argtypes = [_dp] + [C.c_double] * 5 + [_dp]
        L.orc_n2o_solve_concentration.restype = None
        L.orc_chem_run.argtypes = [C.c_int32, C.c_int64, C.c_int32, _dp, _dp, _dp, _ip, _dp, _dp, C.c_int64, C.c_int64]
        L.orc_chem_run.restype = C.c_int32
        for f in ("n_params", "n_inputs", "n_states", "n_outputs"):
            getattr(L, "orc_carbon_" + f).argtypes = [C.c_int32]
            getattr(L, "orc_carbon_" + f).restype = C.c_int32
        L.orc_carbon_default_params.argtypes = [C.c_int32, _dp]
        L.orc_carbon_default_params.restype = None
        L.orc_co2_budget_solve.argtypes = [_dp, _dp, C.c_double, C.c_double, _dp]
        L.orc_co2_budget_solve.restype = None
        L.orc_terrestrial_taus.argtypes = [_dp, _dp]
        L.orc_terrestrial_taus.restype = None
        L.orc_terrestrial_solve_pools.argtypes = [_dp, _dp, _dp, C.c_double, _dp]
        L.orc_terrestrial_solve_pools.restype = None
        L.orc_carbon_run.argtypes = [C.c_int32, C.c_int64, C.c_int32, _dp, _dp, _dp, _ip, _dp, C.c_int64, C.c_int64]
        L.orc_carbon_run.restype = C.c_int32
        L.orc_ocean_n_params.restype = C.c_int32
        L.orc_ocean_default_params.argtypes = [C.c_int32, _dp]
        L.orc_ocean_default_params.restype = None
        L.orc_ocean_irf.argtypes = [_dp, C.c_double]
        L.orc_ocean_irf.restype = C.c_double
        L.orc_ocean_delta_pco2_from_dic.argtypes = [_dp, C.c_double]
        L.orc_ocean_delta_pco2_from_dic.restype = C.c_double
        L.orc_ocean_pco2.argtypes = [_dp, C.c_double, C.c_double]
        L.orc_ocean_pco2.restype = C.c_double
        L.orc_ocean_run.argtypes = [C.c_int64, C.c_int32, _dp, _dp, _dp, _ip, _dp, C.c_int64, C.c_int64]
        L.orc_ocean_run.restype = C.c_int32
        L.orc_ocean_solve_repeated.argtypes = [_dp] + [C.c_double] * 5 + [C.c_int32, _dp]
        L.orc_ocean_solve_repeated.restype = C.c_int32
        for f in ("n_params", "n_species", "n_fgases"):
            getattr(L, "orc_halo_" + f).restype = C.c_int32
        L.orc_halo_default_params.argtypes = [_dp]
        L.orc_halo_default_params.restype = None
        L.orc_halo_decay_species.argtypes = [_dp, C.c_int32, C.c_double, C.c_double, C.c_double]
        L.orc_halo_decay_species.restype = C.c_double
        L.orc_halo_aggregates.argtypes = [_dp, _dp, _dp]
        L.orc_halo_aggregates.restype = None
        L.orc_halo_run.argtypes = [C.c_int64, C.c_int32, _dp, _dp, _dp, _ip, _dp, C.c_int64, C.c_int64]
        L.orc_halo_run.restype = C.c_int32
        L.orc_udeb_lamcalc.argtypes = [_dp, C.c_double, _dp]
        L.orc_udeb_area_factors.argtypes = [_dp, _dp, _dp, _dp]
        L.orc_udeb_sst_to_air.argtypes = [_dp, C.c_double]
        L.orc_udeb_sst_to_air.restype = C.c_double
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


# ------------------------------------------------------------------------------ ClimateUDEB
UDEB_PARAM_NAMES = (
    "n_layers", "mixed_layer_depth", "layer_thickness", "kappa", "kappa_min", "kappa_dkdt",
    "w_initial", "w_variable_fraction", "w_threshold_temp_nh", "w_threshold_temp_sh", "ecs",
    "rf_2xco2", "rlo", "feedback_q_sensitivity", "feedback_cumt_sensitivity",
    "feedback_cumt_period", "k_lo", "k_ns", "amplify_ocean_to_land", "nh_land_fraction",
    "sh_land_fraction", "depth_dependent_area", "temp_adjust_alpha", "temp_adjust_gamma",
    "polar_sinking_ratio", "land_heat_capacity_enabled", "k_lg", "land_hc_eff_thickness",
    "rf_regions_co2_0", "rf_regions_co2_1", "rf_regions_co2_2", "rf_regions_co2_3",
    "efficacy_apply", "prescribed_efficacy_co2", "ocean_temp_profile", "steps_per_year",
    "max_temperature")
UDEB_VARS = ("st0", "st1", "st2", "st3", "heat_uptake", "ohc", "sst")


def udeb_default_params(**overrides):
    """[P] vector of ClimateUDEBParameters::default() with named overrides."""
    n = lib().orc_udeb_n_params()
    assert n == len(UDEB_PARAM_NAMES)
    p = np.empty(n)
    lib().orc_udeb_default_params(_d(p))
    for k, v in overrides.items():
        p[UDEB_PARAM_NAMES.index(k)] = float(v)
    return p


def udeb_run(bounds, params, erf, *, scen=None, st_init=(0.0, 0.0, 0.0, 0.0), threads=1):
    """params [P][N]; erf [S][T] on the model axis.  Returns (dict of [T][N], status[N])."""
    bounds = _f64(bounds)
    T = len(bounds) - 1
    params = _f64(params)
    if params.ndim == 1:
        params = params.reshape(-1, 1).copy()
    N = params.shape[1]
    erf = np.atleast_2d(_f64(erf))
    assert erf.shape[1] == T
    if scen is not None:
        scen = np.ascontiguousarray(scen, dtype=np.int32)
    out = {k: np.full((T, N), np.nan) for k in UDEB_VARS}
    status = np.zeros(N, dtype=np.int32)
    init = _f64(st_init)
    L = lib()
    _pmap(lambda i0, i1: L.orc_udeb_run(N, T, _d(bounds), _d(params), erf.shape[0], _d(erf), _i(scen),
                                        _d(init), *[_d(out[k]) for k in UDEB_VARS], _i(status), i0, i1),
          N, threads)
    return out, status


def udeb_lamcalc(params, ecs):
    out = np.empty(4)
    rc = lib().orc_udeb_lamcalc(_d(_f64(params)), ecs, _d(out))
    if rc:
        raise RuntimeError(f"lamcalc rc={rc}")
    return {"lambda_ocean": out[0], "lambda_land": out[1], "co2_internal_efficacy": out[2], "qfrac0": out[3]}


def udeb_area_factors(params):
    p = _f64(params)
    n = int(p[0])
    a, b, c = np.empty(n), np.empty(n), np.empty(n)
    rc = lib().orc_udeb_area_factors(_d(p), _d(a), _d(b), _d(c))
    if rc:
        raise RuntimeError(f"area_factors rc={rc}")
    return a, b, c


def udeb_sst_to_air(params, sst):
    return lib().orc_udeb_sst_to_air(_d(_f64(params)), sst)


# ---------------------------------------------------------------------------- GhgForcing
GHG_PARAM_NAMES = ("method", "co2_pi", "ch4_pi", "n2o_pi", "delq2xco2", "ch4_radeff", "n2o_radeff",
                   "olbl_co2_a1", "olbl_co2_b1", "olbl_co2_c1", "olbl_co2_d1",
                   "olbl_ch4_a3", "olbl_ch4_b3", "olbl_ch4_d3",
                   "olbl_n2o_a2", "olbl_n2o_b2", "olbl_n2o_c2", "olbl_n2o_d2",
                   "adjust_co2", "adjust_ch4", "adjust_n2o")
GHG_VARS = ("co2_erf", "ch4_erf", "n2o_erf")
GHG_METHODS = {"Ipcctar": 0.0, "Olbl": 1.0}


def ghg_default_params(**over) -> np.ndarray:
    """GhgForcingParameters::default() as the 21-entry vector; ``method`` may be a name."""
    p = np.empty(lib().orc_ghg_n_params())
    lib().orc_ghg_default_params(_d(p))
    for k, v in over.items():
        p[GHG_PARAM_NAMES.index(k)] = GHG_METHODS[v] if isinstance(v, str) else v
    return p


def ghg_forcings(params, co2, ch4, n2o):
    out = np.empty(3)
    lib().orc_ghg_forcings(_d(_f64(params)), co2, ch4, n2o, _d(out))
    return dict(zip(GHG_VARS, out))


def ghg_run(n_times, params, conc, *, scen=None, threads=1):
    """params [21][N]; conc [S][3][T] (CO2, CH4, N2O).  Returns dict of [T][N]; row 0 is NaN."""
    params = _f64(params)
    if params.ndim == 1:
        params = params.reshape(-1, 1).copy()
    N = params.shape[1]
    conc = _f64(conc)
    if conc.ndim == 2:
        conc = conc[None]
    assert conc.shape[1:] == (3, n_times)
    if scen is not None:
        scen = np.ascontiguousarray(scen, dtype=np.int32)
    out = {k: np.full((n_times, N), np.nan) for k in GHG_VARS}
    L = lib()
    _pmap(lambda i0, i1: L.orc_ghg_run(N, n_times, _d(params), conc.shape[0], _d(conc), _i(scen),
                                       *[_d(out[k]) for k in GHG_VARS], i0, i1), N, threads)
    return out


# ------------------------------------------------- OzoneForcing / AerosolDirect / AerosolIndirect
PW_OZONE, PW_AEROSOL_DIRECT, PW_AEROSOL_INDIRECT = 4, 5, 6
PW_FOURBOX_OHU, PW_OSPP = 13, 14  # rscm-components: FourBoxOceanHeatUptake, OceanSurfacePartialPressure
PW_PARAM_NAMES = {
    PW_OZONE: ("eesc_reference", "strat_o3_scale", "strat_cl_exponent", "trop_radeff", "trop_oz_ch4",
               "trop_oz_nox", "trop_oz_co", "trop_oz_voc", "ch4_pi", "nox_pi", "co_pi", "nmvoc_pi",
               "temp_feedback_scale"),
    PW_AEROSOL_DIRECT: ("sox_coefficient", "bc_coefficient", "oc_coefficient", "nitrate_coefficient")
    + tuple(f"{s}_regional_{i}" for s in ("sox", "bc", "oc", "nitrate") for i in range(4))
    + ("sox_pi", "bc_pi", "oc_pi", "nox_pi", "harmonize", "harmonize_year", "harmonize_target"),
    PW_AEROSOL_INDIRECT: ("cloud_albedo_coefficient", "reference_burden", "sox_weight", "oc_weight",
                          "sox_pi", "oc_pi", "harmonize", "harmonize_year", "harmonize_target"),
    PW_FOURBOX_OHU: ("northern_ocean_ratio", "northern_land_ratio", "southern_ocean_ratio", "southern_land_ratio"),
    PW_OSPP: ("ospp_preindustrial", "sensitivity_ospp_to_temperature", "sea_surface_temperature_preindustrial")
    + tuple(f"delta_ospp_offsets_{i}" for i in range(5)) + tuple(f"delta_ospp_coefficients_{i}" for i in range(5)),
}


def pointwise_default_params(kind, **over) -> np.ndarray:
    p = np.empty(lib().orc_pointwise_n_params(kind))
    assert len(p) == len(PW_PARAM_NAMES[kind])
    lib().orc_pointwise_default_params(kind, _d(p))
    for k, v in over.items():
        p[PW_PARAM_NAMES[kind].index(k)] = v
    return p


def pointwise_eval(kind, params, inputs) -> np.ndarray:
    out = np.empty(lib().orc_pointwise_n_outputs(kind))
    x = _f64(inputs)
    assert len(x) == lib().orc_pointwise_n_inputs(kind)
    assert lib().orc_pointwise_eval(kind, _d(_f64(params)), _d(x), _d(out)) == 0
    return out


def pointwise_run(kind, n_times, params, inputs, *, scen=None, threads=1) -> np.ndarray:
    """params [P][N]; inputs [S][n_inputs][T].  Returns [n_outputs][T][N]; row 0 is NaN."""
    params = _f64(params)
    if params.ndim == 1:
        params = params.reshape(-1, 1).copy()
    N = params.shape[1]
    inputs = _f64(inputs)
    if inputs.ndim == 2:
        inputs = inputs[None]
    L = lib()
    assert inputs.shape[1:] == (L.orc_pointwise_n_inputs(kind), n_times)
    if scen is not None:
        scen = np.ascontiguousarray(scen, dtype=np.int32)
    out = np.full((L.orc_pointwise_n_outputs(kind), n_times, N), np.nan)
    _pmap(lambda i0, i1: L.orc_pointwise_run(kind, N, n_times, _d(params), _d(inputs), _i(scen), _d(out), i0, i1),
          N, threads)
    return out


# ------------------------------------------------------------ CH4Chemistry / N2OChemistry
CHEM_CH4, CHEM_N2O = 7, 8
CHEM_PARAM_NAMES = {
    CHEM_CH4: ("ch4_pi", "natural_emissions", "tau_oh", "tau_soil", "tau_strat", "tau_trop_cl", "ch4_self_feedback",
               "oh_sensitivity_scale", "oh_nox_sensitivity", "oh_co_sensitivity", "oh_nmvoc_sensitivity",
               "temp_sensitivity", "include_temp_feedback", "include_emissions_feedback", "ppb_to_tg",
               "nox_reference", "co_reference", "nmvoc_reference"),
    CHEM_N2O: ("n2o_pi", "natural_emissions", "tau_n2o", "lifetime_feedback", "strat_delay", "ppb_to_tg"),
}


def chem_default_params(kind, **over) -> np.ndarray:
    p = np.empty(lib().orc_chem_n_params(kind))
    assert len(p) == len(CHEM_PARAM_NAMES[kind])
    lib().orc_chem_default_params(kind, _d(p))
    for k, v in over.items():
        p[CHEM_PARAM_NAMES[kind].index(k)] = float(v)
    return p


def ch4_solve_concentration(params, prev, current, emissions, temperature, nox, co, nmvoc):
    out = np.empty(2)
    lib().orc_ch4_solve_concentration(_d(_f64(params)), prev, current, emissions, temperature, nox, co, nmvoc, _d(out))
    return float(out[0]), float(out[1])


def n2o_solve_concentration(params, prev, current, lagged, emissions, dt):
    out = np.empty(2)
    lib().orc_n2o_solve_concentration(_d(_f64(params)), prev, current, lagged, emissions, dt, _d(out))
    return float(out[0]), float(out[1])


def chem_run(kind, bounds, params, inputs, conc0, *, scen=None, threads=1):
    """params [P][N]; inputs [S][n_inputs][T]; conc0 scalar or [N].  Returns (conc, lifetime), [T][N]."""
    bounds = _f64(bounds)
    T = len(bounds) - 1
    params = _f64(params)
    if params.ndim == 1:
        params = params.reshape(-1, 1).copy()
    N = params.shape[1]
    inputs = _f64(inputs)
    if inputs.ndim == 2:
        inputs = inputs[None]
    L = lib()
    assert inputs.shape[1:] == (L.orc_chem_n_inputs(kind), T)
    if scen is not None:
        scen = np.ascontiguousarray(scen, dtype=np.int32)
    conc = np.full((T, N), np.nan)
    conc[0] = conc0
    life = np.full((T, N), np.nan)
    _pmap(lambda i0, i1: L.orc_chem_run(kind, N, T, _d(bounds), _d(params), _d(inputs), _i(scen), _d(conc), _d(life),
                                        i0, i1), N, threads)
    return conc, life


# ------------------------------------------------------------ CO2Budget / TerrestrialCarbon
CARBON_BUDGET, CARBON_TERRESTRIAL = 9, 10
CARBON_PARAM_NAMES = {
    CARBON_BUDGET: ("gtc_per_ppm", "co2_pi"),
    CARBON_TERRESTRIAL: ("npp_pi", "co2_pi", "beta", "npp_temp_sensitivity", "resp_temp_sensitivity",
                         "detritus_temp_sensitivity", "soil_temp_sensitivity", "humus_temp_sensitivity",
                         "plant_pool_pi", "detritus_pool_pi", "soil_pool_pi", "humus_pool_pi", "respiration_pi",
                         "frac_npp_to_plant", "frac_npp_to_detritus", "frac_plant_to_detritus",
                         "frac_detritus_to_soil", "frac_soil_to_humus", "enable_fertilization", "enable_temp_feedback"),
}


def carbon_default_params(kind, **over) -> np.ndarray:
    p = np.empty(lib().orc_carbon_n_params(kind))
    assert len(p) == len(CARBON_PARAM_NAMES[kind])
    lib().orc_carbon_default_params(kind, _d(p))
    for k, v in over.items():
        p[CARBON_PARAM_NAMES[kind].index(k)] = float(v)
    return p


def co2_budget_solve(params, fossil, landuse, terrestrial, ocean, co2, dt):
    out = np.empty(3)
    lib().orc_co2_budget_solve(_d(_f64(params)), _d(_f64([fossil, landuse, terrestrial, ocean])), co2, dt, _d(out))
    return tuple(float(x) for x in out)


def terrestrial_taus(params):
    out = np.empty(4)
    lib().orc_terrestrial_taus(_d(_f64(params)), _d(out))
    return out


def terrestrial_solve_pools(params, co2, temperature, landuse, pools, dt):
    out = np.empty(5)
    lib().orc_terrestrial_solve_pools(_d(_f64(params)), _d(_f64([co2, temperature, landuse])), _d(_f64(pools)), dt, _d(out))
    return out[:4].copy(), float(out[4])


def carbon_run(kind, bounds, params, inputs, initial, *, scen=None, threads=1):
    """params [P][N]; inputs [S][n_inputs][T]; initial [n_states] or [n_states][N].
    Returns [n_states + n_outputs][T][N] (states first)."""
    bounds = _f64(bounds)
    T = len(bounds) - 1
    params = _f64(params)
    if params.ndim == 1:
        params = params.reshape(-1, 1).copy()
    N = params.shape[1]
    inputs = _f64(inputs)
    if inputs.ndim == 2:
        inputs = inputs[None]
    L = lib()
    ns, no = L.orc_carbon_n_states(kind), L.orc_carbon_n_outputs(kind)
    assert inputs.shape[1:] == (L.orc_carbon_n_inputs(kind), T)
    if scen is not None:
        scen = np.ascontiguousarray(scen, dtype=np.int32)
    series = np.full((ns + no, T, N), np.nan)
    init = _f64(initial)
    series[:ns, 0] = init.reshape(ns, -1)
    _pmap(lambda i0, i1: L.orc_carbon_run(kind, N, T, _d(bounds), _d(params), _d(inputs), _i(scen), _d(series), i0, i1),
          N, threads)
    return series


# ------------------------------------------------------------------------------ OceanCarbon
OCEAN_MODELS = {"3D-GFDL": 0, "2D-BERN": 1, "HILDA": 2}
OCEAN_PARAM_NAMES = (("model", "co2_pi", "pco2_pi", "gas_exchange_scale", "gas_exchange_tau", "temp_sensitivity",
                      "irf_scale", "mixed_layer_depth", "ocean_surface_area", "sst_pi", "steps_per_year",
                      "max_history_months", "irf_switch_time")
                     + tuple(f"delta_ospp_offsets_{i}" for i in range(5))
                     + tuple(f"delta_ospp_coefficients_{i}" for i in range(5)) + ("enable_temp_feedback",))


def ocean_default_params(model="3D-GFDL", **over) -> np.ndarray:
    p = np.empty(lib().orc_ocean_n_params())
    assert len(p) == len(OCEAN_PARAM_NAMES)
    lib().orc_ocean_default_params(OCEAN_MODELS[model], _d(p))
    for k, v in over.items():
        p[OCEAN_PARAM_NAMES.index(k)] = float(v)
    return p


def ocean_irf(params, t):
    return lib().orc_ocean_irf(_d(_f64(params)), float(t))


def ocean_delta_pco2_from_dic(params, d):
    return lib().orc_ocean_delta_pco2_from_dic(_d(_f64(params)), float(d))


def ocean_pco2(params, delta_pco2_dic, delta_sst):
    return lib().orc_ocean_pco2(_d(_f64(params)), float(delta_pco2_dic), float(delta_sst))


def ocean_solve_repeated(params, co2, delta_sst, pco2, cumulative, dt, n_calls):
    """n_calls x (pco2, cumulative, flux) of consecutive solve_ocean calls from an empty history."""
    out = np.empty((n_calls, 3))
    assert lib().orc_ocean_solve_repeated(_d(_f64(params)), co2, delta_sst, pco2, cumulative, dt, n_calls, _d(out)) == 0
    return out


def ocean_run(bounds, params, inputs, pco2_0, cumulative_0=0.0, *, scen=None, threads=1):
    """params [P][N]; inputs [S][2][T] (CO2, SST anomaly).  Returns [3][T][N]: pCO2, cumulative, flux."""
    bounds = _f64(bounds)
    T = len(bounds) - 1
    params = _f64(params)
    if params.ndim == 1:
        params = params.reshape(-1, 1).copy()
    N = params.shape[1]
    inputs = _f64(inputs)
    if inputs.ndim == 2:
        inputs = inputs[None]
    assert inputs.shape[1:] == (2, T)
    if scen is not None:
        scen = np.ascontiguousarray(scen, dtype=np.int32)
    series = np.full((3, T, N), np.nan)
    series[0, 0] = pco2_0
    series[1, 0] = cumulative_0
    L = lib()
    rc = []
    _pmap(lambda i0, i1: rc.append(L.orc_ocean_run(N, T, _d(bounds), _d(params), _d(inputs), _i(scen), _d(series), i0, i1)),
          N, threads)
    assert not any(rc)
    return series


# ------------------------------------------------------------------------- HalocarbonChemistry
HALO_SPECIES = ("CF4", "C2F6", "C3F8", "C4F10", "C5F12", "C6F14", "C7F16", "C8F18", "c-C4F8", "HFC-23", "HFC-32",
                "HFC-43-10mee", "HFC-125", "HFC-134a", "HFC-143a", "HFC-152a", "HFC-227ea", "HFC-236fa", "HFC-245fa",
                "HFC-365mfc", "NF3", "SF6", "SO2F2",
                "CFC-11", "CFC-12", "CFC-113", "CFC-114", "CFC-115", "HCFC-22", "HCFC-141b", "HCFC-142b", "CH3CCl3",
                "CCl4", "CH3Cl", "CH2Cl2", "CHCl3", "CH3Br", "Halon-1211", "Halon-1301", "Halon-2402", "Halon-1202")
HALO_GLOBALS = ("br_multiplier", "cfc11_release_normalisation", "eesc_delay", "air_molar_mass", "atmospheric_mass_tg",
                "mixing_box_fraction")
HALO_FIELDS = ("lifetime", "radiative_efficiency", "concentration_pi", "molecular_weight", "n_cl", "n_br",
               "fractional_release")


def halo_index(species, field):
    return len(HALO_GLOBALS) + HALO_SPECIES.index(species) * len(HALO_FIELDS) + HALO_FIELDS.index(field)


def halo_default_params(**over) -> np.ndarray:
    """Overrides: a global by name, or ``{"CFC-11.lifetime": 45.0}``-style species fields via ``species=``."""
    p = np.empty(lib().orc_halo_n_params())
    assert lib().orc_halo_n_species() == len(HALO_SPECIES)
    lib().orc_halo_default_params(_d(p))
    for k, v in over.pop("species", {}).items():
        sp, field = k.rsplit(".", 1)
        p[halo_index(sp, field)] = v
    for k, v in over.items():
        p[HALO_GLOBALS.index(k)] = v
    return p


def halo_decay_species(params, species, concentration, emissions, dt):
    return lib().orc_halo_decay_species(_d(_f64(params)), HALO_SPECIES.index(species), concentration, emissions, dt)


def halo_aggregates(params, conc):
    """conc: dict species -> ppt (missing species sit at their pre-industrial value).  Returns
    (total, fgas, montreal, eesc)."""
    p = _f64(params)
    c = np.array([conc.get(s, p[halo_index(s, "concentration_pi")]) for s in HALO_SPECIES], dtype=np.float64)
    out = np.empty(4)
    lib().orc_halo_aggregates(_d(p), _d(c), _d(out))
    return tuple(float(x) for x in out)


def halo_run(bounds, params, emissions, conc0, *, scen=None, threads=1):
    """params [P][N]; emissions [S][41][T]; conc0 [41] or [41][N].  Returns [45][T][N]."""
    bounds = _f64(bounds)
    T = len(bounds) - 1
    params = _f64(params)
    if params.ndim == 1:
        params = params.reshape(-1, 1).copy()
    N = params.shape[1]
    emissions = _f64(emissions)
    if emissions.ndim == 2:
        emissions = emissions[None]
    ns = len(HALO_SPECIES)
    assert emissions.shape[1:] == (ns, T)
    if scen is not None:
        scen = np.ascontiguousarray(scen, dtype=np.int32)
    series = np.full((ns + 4, T, N), np.nan)
    series[:ns, 0] = _f64(conc0).reshape(ns, -1)
    L = lib()
    _pmap(lambda i0, i1: L.orc_halo_run(N, T, _d(bounds), _d(params), _d(emissions), _i(scen), _d(series), i0, i1),
          N, threads)
    return series
