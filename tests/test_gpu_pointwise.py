"""OzoneForcing, AerosolDirect and AerosolIndirect on the GPU (csrc/pointwise.hip through the C
ABI) against the CPU oracle (oracle/forcing_oracle.c).

Tolerance: |gpu - oracle| <= 1e-12 * max(1, |oracle|) where a pow or log is involved (device math
library vs glibc, <= 1-2 ulp each on O(1) results); bit-exact where none is (AerosolDirect, the
ozone temperature feedback, members on the zero branches)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TOL = 1e-12


@pytest.fixture(scope="module")
def ra():
    import rscm_amd
    from rscm_amd import _lib
    _lib.load()
    assert _lib.device_count() >= 1
    return rscm_amd


@pytest.fixture(scope="module")
def orc():
    from oracle import cbind
    return cbind


def _inputs(kind, orc, T, rng):
    """Two scenarios per kind that visit every branch of the component."""
    yr = np.arange(T, dtype=float)
    if kind == orc.PW_OZONE:  # EESC crosses its reference, CH4 dips to zero in scenario 1
        a = np.stack([1000.0 + 6.0 * yr, 700.0 + 5.0 * yr, 0.2 * yr, 2.0 * yr, 0.5 * yr, 0.01 * yr])
        b = np.stack([2500.0 - 5.0 * yr, np.maximum(900.0 - 4.0 * yr, 0.0), 40.0 - 0.1 * yr, 500.0 - yr, 100.0 - 0.2 * yr,
                      np.sin(yr / 7.0)])
    elif kind == orc.PW_AEROSOL_DIRECT:  # scenario 1 sits exactly at pre-industrial for a while
        a = np.stack([1.0 + 0.3 * yr, 2.5 + 0.03 * yr, 10.0 + 0.1 * yr, 10.0 + 0.15 * yr])
        b = np.stack([np.where(yr < 20, 1.0, 80.0 - 0.2 * yr), np.where(yr < 20, 2.5, 9.0 - 0.01 * yr),
                      np.where(yr < 20, 10.0, 30.0), np.where(yr < 20, 10.0, 45.0 - 0.1 * yr)])
    elif kind == orc.PW_FOURBOX_OHU:
        a, b = (0.01 * yr)[None], (3.0 * np.sin(yr / 10.0))[None]
    elif kind == orc.PW_OSPP:  # SST anomaly, DIC anomaly
        a = np.stack([0.01 * yr, 0.2 * yr])
        b = np.stack([np.sin(yr / 8.0), 40.0 * np.cos(yr / 30.0)])
    else:  # burden above and below its pre-industrial value
        a = np.stack([1.0 + 0.3 * yr, 10.0 + 0.1 * yr])
        b = np.stack([np.maximum(60.0 - 0.4 * yr, 0.0), np.maximum(30.0 - 0.2 * yr, 0.0)])
    del rng
    return np.stack([a, b])


def _ensemble(kind, orc, n, rng):
    P = np.repeat(orc.pointwise_default_params(kind).reshape(-1, 1), n, axis=1)
    names = orc.PW_PARAM_NAMES[kind]
    vary = {orc.PW_OZONE: ("eesc_reference", "strat_o3_scale", "strat_cl_exponent", "trop_radeff", "trop_oz_ch4",
                           "trop_oz_nox", "ch4_pi", "temp_feedback_scale"),
            orc.PW_AEROSOL_DIRECT: ("sox_coefficient", "bc_coefficient", "oc_coefficient", "nitrate_coefficient",
                                    "sox_regional_1", "bc_regional_0", "oc_regional_3"),
            orc.PW_AEROSOL_INDIRECT: ("cloud_albedo_coefficient", "reference_burden", "sox_weight", "oc_weight"),
            orc.PW_FOURBOX_OHU: ("northern_ocean_ratio", "southern_ocean_ratio"),
            orc.PW_OSPP: ("ospp_preindustrial", "sensitivity_ospp_to_temperature", "delta_ospp_offsets_0",
                          "delta_ospp_coefficients_3")}[kind]
    for k in vary:
        j = names.index(k)
        P[j] = P[j] * rng.uniform(0.8, 1.25, n)
    return P


def _gpu(ra, kind, T, P, inputs, scen=None, chunks=()):
    b = np.arange(T + 1, dtype=float) + 1750.0
    with ra.Ensemble(kind, P.shape[1], b) as e:
        e.set_params(P)
        e.set_forcing(inputs, scen)
        for c in chunks:
            e.run(c)
        e.run()
        assert not e.status().any()
        return np.stack([e.get_series(v) for v in sorted(v for v in e.var_ids.values() if v > 0)])


@pytest.mark.parametrize("kind_name", ["PW_OZONE", "PW_AEROSOL_DIRECT", "PW_AEROSOL_INDIRECT", "PW_FOURBOX_OHU", "PW_OSPP"])
@pytest.mark.parametrize("n", [1, 63, 1000])
def test_pointwise_gpu_vs_oracle(ra, orc, kind_name, n):
    kind = getattr(orc, kind_name)
    rng = np.random.default_rng(100 * kind + n)
    T = 301
    inputs = _inputs(kind, orc, T, rng)
    P = _ensemble(kind, orc, n, rng)
    scen = (np.arange(n) % 2).astype(np.int32)
    want = orc.pointwise_run(kind, T, P, inputs, scen=scen, threads=8)
    got = _gpu(ra, kind, T, P, inputs, scen=scen)
    assert got.shape == want.shape
    assert (np.isnan(got) == np.isnan(want)).all() and np.isnan(got[:, 0]).all()
    ok = ~np.isnan(want)
    if kind == orc.PW_FOURBOX_OHU:
        assert np.array_equal(got[ok], want[ok])  # one multiply per region
    elif kind == orc.PW_OSPP:
        err = np.abs(got[ok] - want[ok]) / np.maximum(1.0, np.abs(want[ok]))
        assert err.max() <= TOL, f"max deviation {err.max():.3e}"
    elif kind == orc.PW_AEROSOL_DIRECT:
        assert np.array_equal(got[ok], want[ok])  # no transcendental: the same bits
        assert (want[:, 1:20, 1::2] == 0.0).all()  # scenario 1 starts exactly at pre-industrial: zeros
    else:
        err = np.abs(got[ok] - want[ok]) / np.maximum(1.0, np.abs(want[ok]))
        assert err.max() <= TOL, f"max deviation {err.max():.3e}"
        assert np.array_equal(got[ok][want[ok] == 0.0], want[ok][want[ok] == 0.0])  # the zero branches are exact
        assert (want[ok] == 0.0).any() or n == 1
    if kind == orc.PW_OZONE:
        assert np.array_equal(got[2][1:], want[2][1:])  # temperature feedback: one multiply
    # three launches give the same bits as one; one scenario without a map reads scenario 0
    assert np.array_equal(_gpu(ra, kind, T, P, inputs, scen=scen, chunks=(1, 77)), got, equal_nan=True)
    w0 = orc.pointwise_run(kind, T, P, inputs[:1])
    g0 = _gpu(ra, kind, T, P, inputs[:1])
    assert (np.abs(g0[:, 1:] - w0[:, 1:]) <= TOL * np.maximum(1.0, np.abs(w0[:, 1:]))).all()


def test_pointwise_through_the_reference_shaped_front(ra, orc):
    """The same builder calls a user of rscm.magicc makes, one component per model."""
    from rscm_amd import core
    from rscm_amd.magicc import AerosolDirectBuilder, AerosolIndirectBuilder, OzoneForcingBuilder
    years = np.arange(1900.0, 1951.0)
    axis = core.TimeAxis.from_bounds(np.append(years, 1951.0))
    T = len(years)
    rng = np.random.default_rng(3)

    def run(builder, params, series, kind):
        b = core.ModelBuilder().with_time_axis(axis).with_rust_component(builder.from_parameters(params).build())
        for name, vals in series.items():
            b = b.with_exogenous_variable(name, core.Timeseries(vals, axis, "", core.InterpolationStrategy.Previous))
        m = b.build()
        m.run()
        res = m.timeseries()
        m.close()
        return res

    oz_in = {"EESC": 1400.0 + 12.0 * np.arange(T), "Atmospheric Concentration|CH4": 900.0 + 15.0 * np.arange(T),
             "Emissions|NOx": 10.0 + rng.uniform(0, 1, T), "Emissions|CO": 300.0 + 2.0 * np.arange(T),
             "Emissions|NMVOC": 60.0 + np.arange(T), "Surface Temperature": 0.02 * np.arange(T)}
    res = run(OzoneForcingBuilder, {"strat_o3_scale": -0.005, "trop_radeff": 0.04}, oz_in, orc.PW_OZONE)
    P = orc.pointwise_default_params(orc.PW_OZONE, strat_o3_scale=-0.005, trop_radeff=0.04)
    want = orc.pointwise_run(orc.PW_OZONE, T, P, np.stack(list(oz_in.values())))
    for k, name in enumerate(("Stratospheric", "Tropospheric", "Temperature Feedback")):
        got = res.get_timeseries_by_name("Effective Radiative Forcing|O3|" + name).values()
        assert np.isnan(got[0]) and np.abs(got[1:] - want[k, 1:, 0]).max() <= TOL
    assert np.array_equal(res.get_timeseries_by_name("EESC").values(), oz_in["EESC"])

    ad_in = {"Emissions|SOx": 1.0 + 1.5 * np.arange(T), "Emissions|BC": 2.5 + 0.1 * np.arange(T),
             "Emissions|OC": 10.0 + 0.4 * np.arange(T), "Emissions|NOx": 10.0 + 0.5 * np.arange(T)}
    res = run(AerosolDirectBuilder, {"sox_regional": [0.2, 0.5, 0.1, 0.2], "bc_coefficient": 0.009}, ad_in, orc.PW_AEROSOL_DIRECT)
    P = orc.pointwise_default_params(orc.PW_AEROSOL_DIRECT, bc_coefficient=0.009, sox_regional_0=0.2, sox_regional_1=0.5,
                                     sox_regional_2=0.1, sox_regional_3=0.2)
    want = orc.pointwise_run(orc.PW_AEROSOL_DIRECT, T, P, np.stack(list(ad_in.values())))
    fb = res.get_fourbox_timeseries_by_name("Effective Radiative Forcing|Aerosol|Direct").values()
    assert fb.shape == (T, 4) and np.isnan(fb[0]).all() and np.array_equal(fb[1:], want[:, 1:, 0].T)

    ai_in = {"Emissions|SOx": 1.0 + 1.5 * np.arange(T), "Emissions|OC": 10.0 + 0.4 * np.arange(T)}
    res = run(AerosolIndirectBuilder, {"cloud_albedo_coefficient": -1.2}, ai_in, orc.PW_AEROSOL_INDIRECT)
    P = orc.pointwise_default_params(orc.PW_AEROSOL_INDIRECT, cloud_albedo_coefficient=-1.2)
    want = orc.pointwise_run(orc.PW_AEROSOL_INDIRECT, T, P, np.stack(list(ai_in.values())))
    got = res.get_timeseries_by_name("Effective Radiative Forcing|Aerosol|Indirect").values()
    assert np.isnan(got[0]) and np.abs(got[1:] - want[0, 1:, 0]).max() <= TOL
    with pytest.raises(ValueError, match="unknown field"):
        OzoneForcingBuilder.from_parameters({"ozone_scale": 1.0})
    with pytest.raises(ValueError, match="length 4"):
        AerosolDirectBuilder.from_parameters({"sox_regional": [0.5, 0.5]})


def test_pointwise_full_size_properties(ra, orc):
    """1e6 members x 751 years: AerosolDirect's four regions sum to the species total the
    reference calls calculate_global_forcing (aerosol_direct.rs:73-83) to rounding, AerosolIndirect
    is monotone in the SOx weight, and sampled members match the oracle."""
    n, T = 1_000_000, 751
    rng = np.random.default_rng(21)
    yr = np.arange(T, dtype=float)
    b = np.arange(T + 1, dtype=float) + 1750.0
    ad_in = np.stack([1.0 + 0.1 * yr, 2.5 + 0.01 * yr, 10.0 + 0.03 * yr, 10.0 + 0.05 * yr])
    P = np.repeat(orc.pointwise_default_params(orc.PW_AEROSOL_DIRECT).reshape(-1, 1), n, axis=1)
    P[:4] *= rng.uniform(0.8, 1.2, (4, n))
    with ra.Ensemble(ra.KIND_AEROSOL_DIRECT, n, b) as e:
        e.set_params(P)
        e.set_forcing(ad_in)
        e.run()
        box = np.stack([e.get_series(v, 100, T, 130) for v in (1, 2, 3, 4)])  # [4][6][n]
    rows = np.arange(100, T, 130)
    delta = ad_in[:, rows - 1] - np.array([1.0, 2.5, 10.0, 10.0])[:, None]  # output row r holds year r-1
    total = (P[:4, None, :] * delta[:, :, None]).sum(axis=0)
    assert np.abs(box.sum(axis=0) - total).max() <= 1e-13
    pick = rng.choice(n, 32, replace=False)
    want = orc.pointwise_run(orc.PW_AEROSOL_DIRECT, T, P[:, pick].copy(), ad_in)
    assert np.array_equal(box[:, :, pick], want[:, rows][:, :, :])

    ai_in = np.stack([1.0 + 0.1 * yr, 10.0 + 0.03 * yr])
    Q = np.repeat(orc.pointwise_default_params(orc.PW_AEROSOL_INDIRECT).reshape(-1, 1), n, axis=1)
    Q[2] = np.sort(rng.uniform(0.5, 2.0, n))  # sox_weight ascending with the member index
    with ra.Ensemble(ra.KIND_AEROSOL_INDIRECT, n, b) as e:
        e.set_params(Q)
        e.set_forcing(ai_in)
        e.run()
        last = e.get_series(1, T - 1, T)[0]
    assert (np.diff(last) <= 0.0).all() and last[0] < 0.0  # more CCN per Mt S: stronger cooling
    w = orc.pointwise_run(orc.PW_AEROSOL_INDIRECT, T, Q[:, pick].copy(), ai_in)[0, T - 1]
    assert np.abs(last[pick] - w).max() <= TOL


def test_rscm_components_pointwise_front(ra, orc):
    """FourBoxOceanHeatUptake and OceanSurfacePartialPressure with the reference's builder calls; the
    OSPP known answers of the reference's own test (339.089 / 381.003 ppm, rel 10e-5) through the GPU."""
    from rscm_amd import core
    from rscm_amd.components import FourBoxOceanHeatUptakeBuilder, OceanSurfacePartialPressureBuilder
    axis = core.TimeAxis.from_values(np.array([2020.0, 2021.0, 2022.0]))
    ts = lambda v: core.Timeseries(np.array(v), axis, "", core.InterpolationStrategy.Previous)  # noqa: E731
    for params, expected in (
            (dict(ospp_preindustrial=278.0, sensitivity_ospp_to_temperature=0.043, sea_surface_temperature_preindustrial=17.9,
                  delta_ospp_offsets=[1.5568, 7.4706, 1.2748, 2.4491, 1.5468],
                  delta_ospp_coefficients=[-0.013993, -0.20207, -0.12015, -0.12639, -0.15326]), 339.089),
            (dict(ospp_preindustrial=315.0, sensitivity_ospp_to_temperature=0.0423, sea_surface_temperature_preindustrial=17.9,
                  delta_ospp_offsets=[1.5, 7.5, 1.3, 2.5, 1.6], delta_ospp_coefficients=[-0.02, -0.2, -0.1, -0.14, -0.2]), 381.003)):
        m = (core.ModelBuilder().with_time_axis(axis)
             .with_rust_component(OceanSurfacePartialPressureBuilder.from_parameters(params).build())
             .with_exogenous_variable("Sea Surface Temperature", ts([4.0, 4.0, 4.0]))
             .with_exogenous_variable("Dissolved Inorganic Carbon", ts([5.0, 5.0, 5.0])).build())
        m.step()
        got = m.timeseries().get_timeseries_by_name("Ocean Surface Partial Pressure|CO2").values()
        m.close()
        assert np.isnan(got[0]) and got[1] == pytest.approx(expected, rel=10e-5) and np.isnan(got[2])
    m = (core.ModelBuilder().with_time_axis(axis)
         .with_rust_component(FourBoxOceanHeatUptakeBuilder.from_parameters(
             dict(northern_ocean_ratio=1.2, northern_land_ratio=0.6, southern_ocean_ratio=1.6, southern_land_ratio=0.6)).build())
         .with_exogenous_variable("Effective Radiative Forcing|Aggregated", ts([2.0, 3.0, 4.0])).build())
    m.run()
    fb = m.timeseries().get_fourbox_timeseries_by_name("Heat Uptake|Ocean").values()
    m.close()
    assert np.isnan(fb[0]).all() and np.array_equal(fb[1], 2.0 * np.array([1.2, 0.6, 1.6, 0.6])) and np.array_equal(fb[2], 3.0 * np.array([1.2, 0.6, 1.6, 0.6]))
    with pytest.raises(ValueError, match="average to 1.0"):
        FourBoxOceanHeatUptakeBuilder.from_parameters(dict(northern_ocean_ratio=2.0, northern_land_ratio=2.0, southern_ocean_ratio=2.0,
                                                            southern_land_ratio=2.0)).build()
    with pytest.raises(ValueError, match="missing field"):
        OceanSurfacePartialPressureBuilder.from_parameters({"ospp_preindustrial": 278.0})
