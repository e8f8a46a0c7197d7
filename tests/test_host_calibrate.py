"""Host-side calibration pieces that need no GPU: the LogNormal / Bound priors
(crates/rscm-calibrate/src/distribution.rs:281-530), Chain diagnostics and persistence
(sampler/diagnostics.rs:164-335, sampler/chain.rs:190-277), SamplerState (sampler/state.rs) and the
checkpointed sampler runs (sampler/ensemble.rs:272-410, 548-660), driven by an in-process linear
model like the one of the reference's own sampler tests (ensemble.rs:667-720)."""
import math

import numpy as np
import pytest

from rscm_amd import calibrate as cal


class LinearRunner:
    """y = a x + b at x = 0..9 (ensemble.rs:667-700), with the batch likelihood on the host."""
    param_names = ["a", "b"]
    output_variables = ["y"]

    def run_batch(self, param_sets):
        return [{"y": {float(x): a * x + b for x in range(10)}} for a, b in np.asarray(param_sets)]

    def log_likelihood_batch(self, param_sets, target, likelihood):
        return np.array([likelihood.ln_likelihood(o, target) for o in self.run_batch(param_sets)])


@pytest.fixture()
def problem():
    target = cal.Target()
    for x in range(10):
        target.add_observation("y", float(x), 2.0 * x + 1.0, 0.5)
    params = cal.ParameterSet().add("a", cal.Uniform(0.0, 5.0)).add("b", cal.Uniform(-2.0, 4.0))
    return cal.EnsembleSampler(params, LinearRunner(), cal.GaussianLikelihood(), target)


def test_lognormal_and_bound_priors():
    ln = cal.LogNormal(0.5, 0.4)
    z = (math.log(2.0) - 0.5) / 0.4
    assert ln.ln_pdf(2.0) == pytest.approx(-0.5 * z * z - math.log(2.0) - math.log(0.4) - 0.5 * math.log(2 * math.pi), rel=1e-15)
    assert ln.ln_pdf(0.0) == -math.inf and ln.ln_pdf(-1.0) == -math.inf and ln.bounds() == (0.0, math.inf)
    assert np.array_equal(ln.ln_pdf_n(np.array([2.0, 0.0, -3.0])), [ln.ln_pdf(2.0), -math.inf, -math.inf])
    with pytest.raises(ValueError, match="must be positive"):
        cal.LogNormal(0.0, 0.0)
    rng = np.random.default_rng(0)
    s = ln.sample_n(rng, 200_000)
    assert abs(np.log(s).mean() - 0.5) < 0.005 and abs(np.log(s).std() - 0.4) < 0.005
    bd = cal.Bound(cal.Normal(0.0, 1.0), -1.0, 2.0)
    assert bd.ln_pdf(0.3) == cal.Normal(0.0, 1.0).ln_pdf(0.3)  # unnormalised inner density
    assert bd.ln_pdf(-1.5) == -math.inf and bd.ln_pdf(2.5) == -math.inf and bd.bounds() == (-1.0, 2.0)
    x = bd.sample_n(rng, 50_000)
    assert x.min() >= -1.0 and x.max() <= 2.0 and -1.0 <= bd.sample(rng) <= 2.0
    with pytest.raises(ValueError, match="must be less than"):
        cal.Bound(cal.Normal(0.0, 1.0), 1.0, 1.0)
    ps = cal.ParameterSet().add("k", cal.LogNormal(0.0, 0.3)).add("m", bd)
    assert ps.log_prior([1.2, 0.5]) == pytest.approx(cal.LogNormal(0.0, 0.3).ln_pdf(1.2) + bd.ln_pdf(0.5))
    assert ps.log_prior([1.2, 3.0]) == -math.inf and ps.bounds() == ([0.0, -1.0], [math.inf, 2.0])


def test_chain_ess_and_autocorrelation_time():
    rng = np.random.default_rng(1)
    # AR(1) walkers with known autocorrelation rho^k: tau = (1 + rho) / (1 - rho)
    rho, n, w = 0.6, 4000, 8
    x = np.zeros((n, w, 1))
    for t in range(1, n):
        x[t] = rho * x[t - 1] + math.sqrt(1 - rho * rho) * rng.normal(size=(w, 1))
    c = cal.Chain(["x"], 1)
    for t in range(n):
        c.push(x[t], np.zeros(w))
    tau = c.autocorr_time()["x"]
    assert abs(tau - (1 + rho) / (1 - rho)) < 0.4
    assert c.ess()["x"] == pytest.approx(n * w / tau, rel=1e-12)
    white = cal.Chain(["x"], 1)
    for t in range(400):
        white.push(rng.normal(size=(w, 1)), np.zeros(w))
    assert 0.5 * 400 * w < white.ess()["x"] <= 400 * w and white.autocorr_time()["x"] < 1.5
    assert cal.Chain(["x"], 1).ess() == {} and white.ess(discard=395) == {}  # fewer than 10 kept samples
    const = cal.Chain(["x"], 1)
    for t in range(20):
        const.push(np.ones((w, 1)), np.zeros(w))
    assert const.autocorr_time()["x"] == 1.0  # zero variance: no autocorrelation (diagnostics.rs:311-313)


def test_chain_save_load_merge(tmp_path):
    rng = np.random.default_rng(2)
    a, b = cal.Chain(["p", "q"], 2), cal.Chain(["p", "q"], 2)
    for c in (a, b):
        for _ in range(5):
            c.push(rng.normal(size=(4, 2)), rng.normal(size=4))
    assert len(a) == 3 and a.total_iterations == 5  # thin 2 keeps sweeps 1, 3, 5
    a.save(tmp_path / "a.chain")
    back = cal.Chain.load(tmp_path / "a.chain")
    assert back.param_names == ["p", "q"] and back.thin == 2 and back.total_iterations == 5
    assert np.array_equal(back.flat_samples(), a.flat_samples()) and np.array_equal(back.flat_log_probs(), a.flat_log_probs())
    a.merge(b)
    assert len(a) == 6 and a.total_iterations == 10 and np.array_equal(a.flat_samples(3), b.flat_samples())
    with pytest.raises(ValueError, match="different parameter names"):
        a.merge(cal.Chain(["p", "z"], 2))
    with pytest.raises(ValueError, match="different thinning"):
        a.merge(cal.Chain(["p", "q"], 1))
    empty = cal.Chain(["p"], 1)
    empty.save(tmp_path / "e.chain")
    assert len(cal.Chain.load(tmp_path / "e.chain")) == 0


def test_sampler_state_and_checkpoints(tmp_path):
    st = cal.SamplerState(np.arange(8.0).reshape(4, 2), ["a", "b"])
    assert st.n_walkers() == 4 and st.n_params() == 2 and (st.log_probs == -np.inf).all()
    assert st.mean_acceptance_rate() == 0.0 and (st.acceptance_fraction() == 0.0).all()
    st.n_accepted[:] = [1, 2, 0, 3]
    st.n_proposed[:] = [4, 4, 0, 4]
    assert np.allclose(st.acceptance_fraction(), [0.25, 0.5, 0.0, 0.75]) and st.mean_acceptance_rate() == 0.5
    st.save_checkpoint(tmp_path / "s.state")
    back = cal.SamplerState.load_checkpoint(tmp_path / "s.state")
    assert np.array_equal(back.positions, st.positions) and np.array_equal(back.n_accepted, st.n_accepted) and back.param_names == ["a", "b"]
    with pytest.raises(ValueError, match="does not match"):
        cal.SamplerState(np.zeros((4, 2)), ["a"])
    with pytest.raises(ValueError, match="at least 2 walkers"):
        cal.SamplerState(np.zeros((1, 2)), ["a", "b"])


def test_checkpointed_run_and_resume(problem, tmp_path):
    base = tmp_path / "run"
    seen = []
    chain = problem.run_with_checkpoint(60, cal.WalkerInit.from_prior(), 1, 20, base, n_walkers=32,
                                        rng=np.random.default_rng(3), progress=lambda it, acc, lp: seen.append(it))
    assert len(chain) == 60 and seen == list(range(60))
    saved = cal.Chain.load(f"{base}.chain")
    assert saved.total_iterations == 60 and np.array_equal(saved.flat_samples(), chain.flat_samples())
    state = cal.SamplerState.load_checkpoint(f"{base}.state")
    assert np.array_equal(state.positions, chain.flat_samples(59)) and (state.n_proposed == 60).all()
    # resuming towards a total of 100 runs the 40 that are missing; towards 50 runs nothing
    more = problem.resume_from_checkpoint(100, 1, 20, base, rng=np.random.default_rng(4))
    assert more.total_iterations == 100 and np.array_equal(more.flat_samples()[: 60 * 32], chain.flat_samples())
    assert problem.resume_from_checkpoint(50, 1, 20, base).total_iterations == 100
    x = more.flat_samples(discard=40)
    assert abs(x[:, 0].mean() - 2.0) < 0.1 and abs(x[:, 1].mean() - 1.0) < 0.4  # y = 2x + 1 recovered
    with pytest.raises(ValueError, match="must be even"):
        problem.run_with_checkpoint(1, cal.WalkerInit.from_prior(), 1, 0, base, n_walkers=31)


def test_tutorial_quadratic_model_through_a_python_factory():
    """docs/notebooks/calibration_tutorial.py part 1 of the reference: a Python callable as the model
    (ModelRunner(model_factory=...)), random-search point estimate, stretch-move sampling with a
    progress tracker, the chain as a DataFrame."""
    rng = np.random.default_rng(42)
    true = dict(a=0.5, b=-1.0, c=2.0)
    x_obs = np.linspace(-3.0, 3.0, 15)
    y_obs = true["a"] * x_obs ** 2 + true["b"] * x_obs + true["c"] + rng.normal(0.0, 0.2, x_obs.size)

    def model_factory(p):
        return {"y": {float(x): float(p["a"] * x ** 2 + p["b"] * x + p["c"]) for x in x_obs}}

    runner = cal.ModelRunner(model_factory=model_factory, param_names=["a", "b", "c"], output_variables=["y"])
    assert isinstance(runner, cal.FactoryModelRunner) and runner.param_names == ["a", "b", "c"]
    assert runner.run([0.5, -1.0, 2.0])["y"][3.0] == 0.5 * 9 - 3 + 2
    with pytest.raises(ValueError, match="Expected 3 parameters, got 2"):
        runner.run([1.0, 2.0])
    params = cal.ParameterSet().add("a", cal.Uniform(-2.0, 2.0)).add("b", cal.Uniform(-3.0, 3.0)).add("c", cal.Uniform(0.0, 4.0))
    target = cal.Target()
    for x, y in zip(x_obs, y_obs):
        target.add_observation("y", float(x), float(y), 0.2)
    lik = cal.GaussianLikelihood()
    est = cal.PointEstimator(params, runner, lik, target)
    res = est.optimize(cal.Optimizer.random_search(), n_samples=600, rng=np.random.default_rng(1))
    assert est.n_evaluations == 600 and np.isfinite(res.best_log_likelihood)
    sampler = cal.EnsembleSampler(params, runner, lik, target)
    tracker = cal.ProgressTracker()
    chain = sampler.run_with_progress(300, cal.WalkerInit.from_prior(), thin=1, progress_callback=tracker,
                                      n_walkers=32, rng=np.random.default_rng(2))
    assert tracker.iterations == list(range(300)) and 0.1 < tracker.acceptance_rates[-1] < 0.9
    assert tracker.mean_log_probs[-1] > tracker.mean_log_probs[0]
    df = chain.to_dataframe(discard=150)
    assert list(df.columns) == ["a", "b", "c", "log_prob"] and df.shape == (150 * 32, 4)
    med = df[["a", "b", "c"]].median()
    assert abs(med["a"] - true["a"]) < 0.1 and abs(med["b"] - true["b"]) < 0.1 and abs(med["c"] - true["c"]) < 0.2
    # a failing member is -inf, the batch goes on
    def flaky(p):
        if p["a"] > 1.0:
            raise ValueError("diverged")
        return model_factory(p)
    ll = cal.ModelRunner(flaky, ["a", "b", "c"], ["y"]).log_likelihood_batch(np.array([[0.5, -1.0, 2.0], [1.5, 0.0, 0.0]]), target, lik)
    assert np.isfinite(ll[0]) and ll[1] == -np.inf
