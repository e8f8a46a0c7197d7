"""The device stretch-move sampler (rscm_sampler_* of the C ABI, csrc/sampler.hip) against the
host sampler of rscm_amd.calibrate, which mirrors crates/rscm-calibrate/src/sampler/.  The
reference draws from thread_rng, so samplers compare by distribution: moments of a known
posterior, the acceptance rule, invariance of the prior, and -- exactly -- the scores it assigns."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

NAMES = ["lambda0", "a", "efficacy", "eta", "heat_capacity_surface", "heat_capacity_deep"]
FIXED = dict(lambda0=1.1, a=0.05, efficacy=1.3, eta=0.7, heat_capacity_surface=8.0, heat_capacity_deep=100.0)


@pytest.fixture(scope="module")
def setup():
    import rscm_amd  # noqa: F401
    from rscm_amd import _lib
    _lib.load()
    assert _lib.device_count() >= 1
    from rscm_amd import calibrate as cal
    from rscm_amd import core
    from rscm_amd.two_layer import TwoLayerBuilder
    t = np.arange(1750.0, 1901.0)
    axis = core.TimeAxis.from_values(t)
    F = 4.0 * (1.0 - np.exp(-(t - 1750.0) / 60.0))
    b = (core.ModelBuilder().with_time_axis(axis).with_rust_component(TwoLayerBuilder.from_parameters(FIXED).build())
         .with_exogenous_variable("Effective Radiative Forcing", core.Timeseries(F, axis, "W/m^2", core.InterpolationStrategy.Linear))
         .with_initial_values({"Surface Temperature": 0.0, "Deep Ocean Temperature": 0.0}))
    return cal, b


def _problem(cal, b, names, ranges, sigma=0.05, years=range(1780, 1901, 10)):
    runner = cal.ModelRunner(b, names, ["Surface Temperature"])
    truth = runner.run([FIXED[k] for k in names])["Surface Temperature"]
    target = cal.Target()
    for yr in years:
        target.add_observation("Surface Temperature", float(yr), truth[float(yr)], sigma)
    params = cal.ParameterSet()
    for k, (lo, hi) in zip(names, ranges):
        params.add(k, cal.Uniform(lo, hi))
    return runner, target, params


def test_device_sampler_scores_equal_host_scores(setup):
    """Initial log-probabilities (log prior + fused device likelihood) are the host sampler's, bit
    for bit, and walkers outside the prior support are -inf."""
    cal, b = setup
    names, ranges = ["lambda0", "efficacy"], [(0.8, 1.5), (1.0, 1.8)]
    runner, target, params = _problem(cal, b, names, ranges)
    lik = cal.GaussianLikelihood()
    rng = np.random.default_rng(0)
    pos = params.sample_random(64, rng)
    pos[5, 0] = 2.0  # outside Uniform(0.8, 1.5)
    dev = cal.DeviceEnsembleSampler(params, runner, lik, target)
    chain = dev.run(0, cal.WalkerInit.explicit(pos), n_walkers=64, seed=1)
    assert len(chain) == 0
    host = cal.EnsembleSampler(params, runner, lik, target)
    want = host.log_posterior_batch(pos)
    # one sweep, then compare only the walkers that did not move: their scores are the initial ones
    chain = dev.run(1, cal.WalkerInit.explicit(pos), n_walkers=64, seed=1)
    got_pos, got_lp = chain.flat_samples(), chain.flat_log_probs()
    same = (got_pos == pos).all(axis=1)
    assert same.any() and (~same).any()
    assert np.array_equal(got_lp[same], want[same]) and want[5] == -np.inf
    moved = host.log_posterior_batch(got_pos[~same])
    assert np.array_equal(got_lp[~same], moved)  # accepted proposals carry their own exact score
    assert np.isfinite(got_lp[~same]).all()
    runner.close()


def test_device_sampler_leaves_the_prior_invariant(setup):
    """No observations: the posterior is the prior.  Uniform x Normal priors, 4096 walkers: after
    60 sweeps from a tight ball the ensemble has the prior's mean and variance."""
    cal, b = setup
    runner = cal.ModelRunner(b, ["lambda0", "eta"], ["Surface Temperature"])
    params = cal.ParameterSet().add("lambda0", cal.Uniform(0.8, 1.6)).add("eta", cal.Normal(0.7, 0.05))
    dev = cal.DeviceEnsembleSampler(params, runner, cal.GaussianLikelihood(), cal.Target())
    chain = dev.run(60, cal.WalkerInit.ball([1.2, 0.7], 0.01), thin=20, n_walkers=4096, seed=7)
    assert chain.total_iterations == 60 and len(chain) == 3  # sweeps 1, 21, 41 (Chain.push keeps 1, 1+thin, ...)
    chain = dev.run(60, cal.WalkerInit.ball([1.2, 0.7], 0.01), n_walkers=4096, seed=7)
    x = chain.flat_samples(discard=59)  # the ensemble after the last sweep
    assert abs(x[:, 0].mean() - 1.2) < 0.02 and abs(x[:, 0].var() - 0.8 ** 2 / 12.0) < 0.004
    assert x[:, 0].min() >= 0.8 and x[:, 0].max() <= 1.6
    assert abs(x[:, 1].mean() - 0.7) < 0.004 and abs(x[:, 1].std() - 0.05) < 0.004
    assert 0.3 < dev.acceptance_rate() < 0.9 and (dev.n_proposed == 60).all()
    runner.close()


def test_device_sampler_matches_host_posterior(setup):
    """A two-parameter calibration problem: host and device samplers agree on the posterior mean
    and spread within their Monte-Carlo error, recover the truth, and the device run is
    reproducible from its seed."""
    cal, b = setup
    names, ranges = ["lambda0", "efficacy"], [(0.8, 1.5), (1.0, 1.8)]
    runner, target, params = _problem(cal, b, names, ranges)
    lik = cal.GaussianLikelihood()
    init = cal.WalkerInit.from_prior()
    dev = cal.DeviceEnsembleSampler(params, runner, lik, target)
    cd = dev.run(300, init, thin=10, n_walkers=512, seed=3, rng=np.random.default_rng(5))
    host = cal.EnsembleSampler(params, runner, lik, target)
    ch = host.run(300, init, thin=10, n_walkers=512, rng=np.random.default_rng(6))
    xd, xh = cd.flat_samples(discard=15), ch.flat_samples(discard=15)
    assert xd.shape == xh.shape == (15 * 512, 2)
    for j in range(2):
        sd = max(xd[:, j].std(), xh[:, j].std())
        assert abs(xd[:, j].mean() - xh[:, j].mean()) < 0.15 * sd
        assert 0.8 < xd[:, j].std() / xh[:, j].std() < 1.25
        assert abs(xd[:, j].mean() - FIXED[names[j]]) < 3 * sd
    assert abs(dev.acceptance_rate() - host.acceptance_rate()) < 0.05
    again = dev.run(300, init, thin=10, n_walkers=512, seed=3, rng=np.random.default_rng(5))
    assert np.array_equal(again.flat_samples(), cd.flat_samples())
    other = dev.run(300, init, thin=10, n_walkers=512, seed=4, rng=np.random.default_rng(5))
    assert not np.array_equal(other.flat_samples(), cd.flat_samples())
    # lambda0 and efficacy are strongly correlated in this posterior: 15 kept sweeps per walker
    # leave the split-chain R-hat of both samplers at the same, still elevated, level
    rd, rh = cd.r_hat(discard=15), ch.r_hat(discard=15)
    assert all(abs(rd[k] - rh[k]) < 0.15 and rd[k] < 1.5 for k in rd)
    runner.close()


def test_device_sampler_error_conventions(setup):
    cal, b = setup
    runner, target, params = _problem(cal, b, ["lambda0"], [(0.8, 1.5)])
    lik = cal.GaussianLikelihood()
    with pytest.raises(ValueError, match="must be > 1.0"):
        cal.DeviceEnsembleSampler(params, runner, lik, target, stretch_a=1.0)
    dev = cal.DeviceEnsembleSampler(params, runner, lik, target)
    with pytest.raises(ValueError, match="must be even"):
        dev.run(1, cal.WalkerInit.from_prior(), n_walkers=33)
    with pytest.raises(ValueError, match="at least 2 walkers"):
        dev.run(1, cal.WalkerInit.from_prior(), n_walkers=1)
    wrong = cal.ParameterSet().add("eta", cal.Uniform(0.5, 1.0))
    with pytest.raises(ValueError, match="runner's parameters"):
        cal.DeviceEnsembleSampler(wrong, runner, lik, target)
    # one sampled dimension works (the z^(d-1) factor is 1)
    chain = dev.run(40, cal.WalkerInit.from_prior(), n_walkers=256, seed=2)
    assert abs(chain.flat_samples(discard=39)[:, 0].mean() - 1.1) < 0.05
    runner.close()


def test_device_sampler_on_climate_udeb(setup):
    """The stored-series path: calibrating ECS and the vertical diffusivity of ClimateUDEB (the
    MAGICC climate core) against its own sea-surface temperatures, proposals and scoring on the
    device.  Initial scores equal the host's; the posterior concentrates on the truth."""
    cal, _ = setup
    from rscm_amd import core
    from rscm_amd.magicc import ClimateUDEBBuilder
    years = np.arange(1850.0, 1911.0)
    axis = core.TimeAxis.from_values(years)
    erf = 3.71 * np.minimum((years - 1850.0) / 40.0, 1.0)
    b = (core.ModelBuilder().with_time_axis(axis)
         .with_rust_component(ClimateUDEBBuilder.from_parameters({"ecs": 3.2, "kappa": 0.9}).build())
         .with_exogenous_variable("Effective Radiative Forcing", core.Timeseries(erf, axis, "W/m^2", core.InterpolationStrategy.Previous))
         .with_initial_values({"Surface Temperature": 0.0}))
    runner = cal.ModelRunner(b, ["ecs", "kappa"], ["Sea Surface Temperature"])
    truth = runner.run([3.2, 0.9])["Sea Surface Temperature"]
    target = cal.Target()
    for yr in range(1860, 1911, 5):
        target.add_observation("Sea Surface Temperature", float(yr), truth[float(yr)], 0.02)
    params = cal.ParameterSet().add("ecs", cal.Uniform(1.5, 6.0)).add("kappa", cal.Uniform(0.3, 2.0))
    lik = cal.GaussianLikelihood()
    dev = cal.DeviceEnsembleSampler(params, runner, lik, target)
    pos = params.sample_random(128, np.random.default_rng(9))
    chain = dev.run(1, cal.WalkerInit.explicit(pos), n_walkers=128, seed=1)
    got_pos, got_lp = chain.flat_samples(), chain.flat_log_probs()
    host = cal.EnsembleSampler(params, runner, lik, target)
    assert np.allclose(got_lp, host.log_posterior_batch(got_pos), rtol=1e-9, atol=1e-9)
    chain = dev.run(150, cal.WalkerInit.from_prior(), n_walkers=128, seed=2, rng=np.random.default_rng(3))
    x = chain.flat_samples(discard=100)
    assert abs(x[:, 0].mean() - 3.2) < 3 * x[:, 0].std() + 0.05 and x[:, 0].std() < 0.5
    assert abs(x[:, 1].mean() - 0.9) < 3 * x[:, 1].std() + 0.05
    assert 0.05 < dev.acceptance_rate() < 0.9 and dev.device_ms > 0
    runner.close()


def test_samplers_refuse_structural_parameters(setup):
    """A [u] row of rscm_gpu.h (here ClimateUDEB's mixed_layer_depth: the host builds the column geometry from it when the
    parameters are set) cannot be a sampled dimension of the device sampler -- its proposals are written on the device and
    would be scored with the base value's tables, silently -- nor vary over the members of the host sampler's batches.  Both
    say so; the reference rebuilds the component per parameter vector (model_runner.rs:257-266)."""
    cal, _ = setup
    from rscm_amd import core
    from rscm_amd.magicc import ClimateUDEBBuilder
    years = np.arange(1850.0, 1871.0)
    axis = core.TimeAxis.from_values(years)
    b = (core.ModelBuilder().with_time_axis(axis)
         .with_rust_component(ClimateUDEBBuilder.from_parameters({"ecs": 3.0}).build())
         .with_exogenous_variable("Effective Radiative Forcing", core.Timeseries(np.full(len(years), 2.0), axis, "W/m^2", core.InterpolationStrategy.Previous))
         .with_initial_values({"Surface Temperature": 0.0}))
    runner = cal.ModelRunner(b, ["ecs", "mixed_layer_depth"], ["Sea Surface Temperature"])
    target = cal.Target()
    target.add_observation("Sea Surface Temperature", 1860.0, 0.5, 0.1)
    params = cal.ParameterSet().add("ecs", cal.Uniform(1.5, 6.0)).add("mixed_layer_depth", cal.Uniform(40.0, 80.0))
    dev = cal.DeviceEnsembleSampler(params, runner, cal.GaussianLikelihood(), target)
    with pytest.raises(Exception, match="structural"):
        dev.run(1, cal.WalkerInit.from_prior(), n_walkers=64, seed=1)
    host = cal.EnsembleSampler(params, runner, cal.GaussianLikelihood(), target)
    with pytest.raises(Exception, match="same for every member"):
        host.log_posterior_batch(params.sample_random(8, np.random.default_rng(0)))
    runner.close()


def test_device_sampler_lognormal_and_bound_priors(setup):
    """No observations: the device sampler must reproduce a LogNormal prior and a Normal truncated
    by Bound; its initial scores are the host prior's."""
    cal, b = setup
    runner = cal.ModelRunner(b, ["heat_capacity_surface", "eta"], ["Surface Temperature"])
    params = (cal.ParameterSet().add("heat_capacity_surface", cal.LogNormal(2.0, 0.25))
              .add("eta", cal.Bound(cal.Normal(0.7, 0.2), 0.5, 1.0)))
    dev = cal.DeviceEnsembleSampler(params, runner, cal.GaussianLikelihood(), cal.Target())
    pos = params.sample_random(4096, np.random.default_rng(1))
    first = dev.run(1, cal.WalkerInit.explicit(pos), n_walkers=4096, seed=5)
    moved = (first.flat_samples() != pos).any(axis=1)
    assert np.allclose(first.flat_log_probs()[~moved], params.log_prior_batch(pos[~moved]), rtol=1e-13, atol=1e-13)
    chain = dev.run(80, cal.WalkerInit.explicit(pos), n_walkers=4096, seed=6)
    x = chain.flat_samples(discard=79)
    assert abs(np.log(x[:, 0]).mean() - 2.0) < 0.02 and abs(np.log(x[:, 0]).std() - 0.25) < 0.02
    assert x[:, 1].min() >= 0.5 and x[:, 1].max() <= 1.0
    ref = params.distributions()[1].sample_n(np.random.default_rng(2), 200_000)  # truncated normal by rejection
    assert abs(x[:, 1].mean() - ref.mean()) < 0.01 and abs(x[:, 1].std() - ref.std()) < 0.01
    runner.close()


def test_device_sampler_groups_are_independent(setup):
    """n_groups independent ensembles side by side: a group's trajectory does not depend on what
    the other groups hold, and every group converges to the same posterior."""
    cal, b = setup
    names, ranges = ["lambda0", "efficacy"], [(0.8, 1.5), (1.0, 1.8)]
    runner, target, params = _problem(cal, b, names, ranges)
    dev = cal.DeviceEnsembleSampler(params, runner, cal.GaussianLikelihood(), target)
    G, Wg = 64, 32
    pos = params.sample_random(G * Wg, np.random.default_rng(11))
    a = dev.run(40, cal.WalkerInit.explicit(pos), n_walkers=G * Wg, seed=8, n_groups=G)
    other = pos.copy()
    other[Wg:] = params.sample_random((G - 1) * Wg, np.random.default_rng(12))  # every group but the first
    c = dev.run(40, cal.WalkerInit.explicit(other), n_walkers=G * Wg, seed=8, n_groups=G)
    xa = np.stack(a._samples)  # [sweeps][walkers][dims]
    xc = np.stack(c._samples)
    assert np.array_equal(xa[:, :Wg], xc[:, :Wg]) and not np.array_equal(xa[:, Wg:], xc[:, Wg:])
    one = dev.run(40, cal.WalkerInit.explicit(pos), n_walkers=G * Wg, seed=8)  # a single big ensemble differs
    assert not np.array_equal(np.stack(one._samples)[:, :Wg], xa[:, :Wg])
    last = xa[-1].reshape(G, Wg, 2).mean(axis=1)  # per-group posterior means
    assert np.abs(last.mean(axis=0) - [1.1, 1.3]).max() < 0.1 and last.std(axis=0).max() < 0.15
    with pytest.raises(Exception, match="do not split"):
        dev.run(1, cal.WalkerInit.explicit(pos), n_walkers=G * Wg, seed=8, n_groups=3)
    runner.close()


def test_device_sampler_full_size_1e5_walkers():
    """BASELINE.json configs[4] on one GPU: 1e5 walkers, the 751-point axis, six two-layer
    parameters, 18 `Surface Temperature` observations (1850 ... 2020 step 10, sigma 0.1 K, values
    from the default-parameter run: SURVEY 8d, C5), three stretch-move sweeps.
      * walkers that did not move keep the score the fused run+likelihood launch gives their
        position, bit for bit (checked against ModelRunner.log_likelihood_batch + the host prior);
      * accepted walkers carry the exact score of their new position;
      * the acceptance fraction of a prior-wide ensemble against a tight likelihood is small but not
        zero, every walker was proposed to exactly once per sweep, and the run is reproducible."""
    import rscm_amd  # noqa: F401
    from rscm_amd import calibrate as cal
    from rscm_amd import core
    from rscm_amd.two_layer import TwoLayerBuilder
    from tests.helpers import TL_RANGES, axis_values, f_syn
    t = axis_values()
    axis = core.TimeAxis.from_values(t)
    defaults = dict(lambda0=1.0, a=0.0, efficacy=1.0, eta=0.7, heat_capacity_surface=8.0, heat_capacity_deep=100.0)
    b = (core.ModelBuilder().with_time_axis(axis).with_rust_component(TwoLayerBuilder.from_parameters(defaults).build())
         .with_exogenous_variable("Effective Radiative Forcing", core.Timeseries(f_syn(t), axis, "W/m^2", core.InterpolationStrategy.Linear))
         .with_initial_values({"Surface Temperature": 0.0, "Deep Ocean Temperature": 0.0}))
    runner = cal.ModelRunner(b, NAMES, ["Surface Temperature"])
    truth = runner.run([defaults[k] for k in NAMES])["Surface Temperature"]
    target = cal.Target()
    for yr in range(1850, 2021, 10):
        target.add_observation("Surface Temperature", float(yr), truth[float(yr)], 0.1)
    params = cal.ParameterSet()
    for k, (lo, hi) in zip(NAMES, TL_RANGES):
        params.add(k, cal.Uniform(lo, hi))
    lik = cal.GaussianLikelihood()
    W = 100_000
    pos = params.sample_random(W, np.random.default_rng(2026))
    dev = cal.DeviceEnsembleSampler(params, runner, lik, target)
    chain = dev.run(3, cal.WalkerInit.explicit(pos), thin=3, n_walkers=W, seed=17)   # keeps sweep 1 only
    assert len(chain) == 1 and chain.total_iterations == 3
    host = cal.EnsembleSampler(params, runner, lik, target)
    want0 = host.log_posterior_batch(pos)
    assert np.isfinite(want0).all()
    p1, lp1 = chain._samples[0], chain._log_probs[0]
    same = (p1 == pos).all(axis=1)
    assert 0.02 < (~same).mean() < 0.9
    assert np.array_equal(lp1[same], want0[same])
    assert np.array_equal(lp1[~same], host.log_posterior_batch(p1[~same]))
    assert (dev.n_proposed == 3).all() and 0.02 < dev.acceptance_rate() < 0.9
    assert (lp1[~same] > -np.inf).all()
    # detailed balance in expectation: accepted moves raise the mean score of a far-from-equilibrium ensemble
    assert lp1.mean() > want0.mean()
    again = dev.run(3, cal.WalkerInit.explicit(pos), thin=3, n_walkers=W, seed=17)
    assert np.array_equal(again._samples[0], p1) and np.array_equal(again._log_probs[0], lp1)
    print(f"1e5 walkers x 3 sweeps: {dev.device_ms:.2f} ms on the device, acceptance {dev.acceptance_rate():.3f}")
    runner.close()
