"""Calibration front for the GPU ensemble -- mirror of ``rscm.calibrate`` for the hot path.

The reference evaluates a batch of parameter vectors with ``ModelRunner::run_batch`` (rayon,
crates/rscm-calibrate/src/model_runner.rs:261-266; the Python runner is sequential under the
GIL, crates/rscm-calibrate/src/python/model_runner.rs:274-278), then log-prior + Gaussian
log-likelihood per member (sampler/ensemble.rs:143-177).  Here ``ModelRunner.run_batch`` uploads
the whole batch as one ``[N][P]`` matrix, steps all members in one kernel launch and either
extracts ``{variable: {time: value}}`` per member (reference shape) or reduces the likelihood on
the device (``log_likelihood_batch``), which is what ``EnsembleSampler``/``PointEstimator`` use.

The sampler's proposal/accept bookkeeping is tiny and stays on the host
(sampler/ensemble.rs:496-547, sampler/moves.rs:16-128), as in the reference.  The reference
draws from ``thread_rng`` (not reproducible); here every random draw takes a ``numpy`` Generator.
"""
from __future__ import annotations

import math
import os
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import _lib as L
from .core import GraphModel, Model, ModelBuilder


# ------------------------------------------------------------------------------------ priors
class Uniform:
    """crates/rscm-calibrate/src/distribution.rs:114-180."""

    def __init__(self, low: float, high: float):
        if low >= high:
            raise ValueError(f"Uniform: low ({low}) must be less than high ({high})")
        self._low, self._high = float(low), float(high)

    low = property(lambda self: self._low)
    high = property(lambda self: self._high)

    def sample(self, rng: np.random.Generator) -> float:
        return self._low + rng.random() * (self._high - self._low)

    def ln_pdf(self, x: float) -> float:
        if x < self._low or x > self._high:
            return -math.inf
        return -math.log(self._high - self._low)

    def bounds(self) -> Optional[Tuple[float, float]]:
        return self._low, self._high

    def quantile(self, u):
        return self._low + u * (self._high - self._low)

    def sample_n(self, rng: np.random.Generator, n: int) -> np.ndarray:
        return self._low + rng.random(n) * (self._high - self._low)

    def ln_pdf_n(self, x: np.ndarray) -> np.ndarray:
        return np.where((x < self._low) | (x > self._high), -np.inf, -math.log(self._high - self._low))


class Normal:
    """distribution.rs:211-275."""

    def __init__(self, mean: float, std_dev: float):
        if not std_dev > 0:
            raise ValueError(f"Normal: std_dev ({std_dev}) must be positive")
        self.mean, self.std_dev = float(mean), float(std_dev)

    def sample(self, rng):
        return rng.normal(self.mean, self.std_dev)

    def ln_pdf(self, x):
        z = (x - self.mean) / self.std_dev
        return -0.5 * z * z - math.log(self.std_dev) - 0.5 * math.log(2.0 * math.pi)

    def bounds(self):
        return None

    def quantile(self, u):
        from scipy.special import ndtri  # exact inverse CDF (the reference approximates it by sampling)
        return self.mean + self.std_dev * ndtri(u)

    def sample_n(self, rng, n):
        return rng.normal(self.mean, self.std_dev, n)

    def ln_pdf_n(self, x):
        z = (np.asarray(x) - self.mean) / self.std_dev
        return -0.5 * z * z - math.log(self.std_dev) - 0.5 * math.log(2.0 * math.pi)


class LogNormal:
    """distribution.rs:281-415: ``ln(X) ~ Normal(mu, sigma)``."""

    def __init__(self, mu: float, sigma: float):
        if not sigma > 0:
            raise ValueError(f"LogNormal: sigma ({sigma}) must be positive")
        self.mu, self.sigma = float(mu), float(sigma)

    def sample(self, rng):
        return rng.lognormal(self.mu, self.sigma)

    def ln_pdf(self, x):
        if x <= 0.0:
            return -math.inf
        ln_x = math.log(x)
        z = (ln_x - self.mu) / self.sigma
        return -0.5 * z * z - ln_x - math.log(self.sigma) - 0.5 * math.log(2.0 * math.pi)

    def bounds(self):
        return 0.0, math.inf

    def quantile(self, u):
        from scipy.special import ndtri
        return np.exp(self.mu + self.sigma * ndtri(u))

    def sample_n(self, rng, n):
        return rng.lognormal(self.mu, self.sigma, n)

    def ln_pdf_n(self, x):
        x = np.asarray(x, dtype=np.float64)
        with np.errstate(all="ignore"):
            ln_x = np.log(x)
            z = (ln_x - self.mu) / self.sigma
            out = -0.5 * z * z - ln_x - math.log(self.sigma) - 0.5 * math.log(2.0 * math.pi)
        return np.where(x <= 0.0, -np.inf, out)


class Bound:
    """distribution.rs:417-530: another distribution truncated to ``[low, high]`` (rejection
    sampling; the log-density is the inner one, unnormalised, inside the bounds)."""

    def __init__(self, distribution, low: float, high: float):
        if low >= high:
            raise ValueError(f"Bound: low ({low}) must be less than high ({high})")
        self.distribution, self._low, self._high = distribution, float(low), float(high)

    low = property(lambda self: self._low)
    high = property(lambda self: self._high)

    def inner(self):
        return self.distribution

    def sample(self, rng):
        while True:
            x = self.distribution.sample(rng)
            if self._low <= x <= self._high:
                return x

    def ln_pdf(self, x):
        if x < self._low or x > self._high:
            return -math.inf
        return self.distribution.ln_pdf(x)

    def bounds(self):
        return self._low, self._high

    def quantile(self, u):
        raise NotImplementedError("Bound has no closed-form quantile (the reference samples it)")

    def sample_n(self, rng, n):
        out = np.empty(n)
        filled = 0
        while filled < n:
            x = self.distribution.sample_n(rng, max(16, 2 * (n - filled)))
            x = x[(x >= self._low) & (x <= self._high)][: n - filled]
            out[filled:filled + len(x)] = x
            filled += len(x)
        return out

    def ln_pdf_n(self, x):
        x = np.asarray(x, dtype=np.float64)
        return np.where((x < self._low) | (x > self._high), -np.inf, self.distribution.ln_pdf_n(x))


class ParameterSet:
    """parameter_set.rs: ordered named priors; ``sample_lhs`` per :207-233."""

    def __init__(self) -> None:
        self._names: List[str] = []
        self._dists: List[object] = []

    def add(self, name: str, distribution) -> "ParameterSet":
        self._names.append(name)
        self._dists.append(distribution)
        return self

    def __len__(self) -> int:
        return len(self._names)

    @property
    def param_names(self) -> List[str]:
        return list(self._names)

    def distributions(self) -> List[object]:
        return list(self._dists)

    def bounds(self) -> Tuple[List[float], List[float]]:
        lo, hi = [], []
        for d in self._dists:
            b = d.bounds()
            lo.append(b[0] if b else -math.inf)
            hi.append(b[1] if b else math.inf)
        return lo, hi

    def sample_random(self, n: int, rng: Optional[np.random.Generator] = None) -> np.ndarray:
        rng = rng or np.random.default_rng()
        return np.column_stack([d.sample_n(rng, n) for d in self._dists])

    def sample_lhs(self, n: int, rng: Optional[np.random.Generator] = None) -> np.ndarray:
        """Host Latin hypercube, ``[n][P]``: per dimension one draw per stratum
        (``i*(1/n) + U*(1/n)``), shuffled, through the inverse CDF."""
        rng = rng or np.random.default_rng()
        out = np.empty((n, len(self)))
        size = 1.0 / n
        for j, d in enumerate(self._dists):
            u = np.arange(n) * size + rng.random(n) * size
            rng.shuffle(u)
            out[:, j] = d.quantile(u)
        return out

    def log_prior(self, params: Sequence[float]) -> float:
        if len(params) != len(self):
            raise ValueError(f"Expected {len(self)} parameters, got {len(params)}")
        return float(sum(d.ln_pdf(x) for d, x in zip(self._dists, params)))

    def log_prior_batch(self, params: np.ndarray) -> np.ndarray:
        params = np.asarray(params, dtype=np.float64)
        if params.ndim != 2 or params.shape[1] != len(self):
            raise ValueError(f"Expected {len(self)} parameters, got {params.shape[-1]}")
        total = np.zeros(params.shape[0])
        for j, d in enumerate(self._dists):  # same left-to-right sum as log_prior
            total = total + d.ln_pdf_n(params[:, j])
        return total


# ------------------------------------------------------------------------------------ target
class Observation:
    def __init__(self, time: float, value: float, uncertainty: float):
        if not uncertainty > 0:  # target.rs:25-56
            raise ValueError(f"Observation uncertainty must be positive, got {uncertainty}")
        self.time, self.value, self.uncertainty = float(time), float(value), float(uncertainty)


class VariableTarget:
    def __init__(self, name: str):
        self.name = name
        self.observations: List[Observation] = []

    def add(self, time, value, uncertainty) -> "VariableTarget":
        self.observations.append(Observation(time, value, uncertainty))
        return self


class Target:
    """target.rs:80-230."""

    def __init__(self) -> None:
        self._vars: Dict[str, VariableTarget] = {}

    def add_variable(self, name: str) -> VariableTarget:
        return self._vars.setdefault(name, VariableTarget(name))

    def add_observation(self, variable: str, time, value, uncertainty) -> "Target":
        self.add_variable(variable).add(time, value, uncertainty)
        return self

    def get_variable(self, name: str) -> Optional[VariableTarget]:
        return self._vars.get(name)

    def variable_names(self) -> List[str]:
        return list(self._vars)

    def variables(self):
        return self._vars.items()

    def total_observations(self) -> int:
        return sum(len(v.observations) for v in self._vars.values())


def time_key(t: float) -> str:
    """likelihood.rs:40-42."""
    return f"{t:.6f}"


class GaussianLikelihood:
    """likelihood.rs:167-250.  ``ln_likelihood`` is the host form over the reference's
    ``{variable: {time: value}}`` output; the batch form runs on the device."""

    def __init__(self, normalize: bool = False):
        self.normalize = bool(normalize)

    def ln_likelihood(self, output: Dict[str, Dict[float, float]], target: Target) -> float:
        total = 0.0
        for name, vt in target.variables():
            if name not in output:
                raise KeyError(f"Model output missing variable: {name}")
            keyed = {time_key(t): v for t, v in output[name].items()}
            ln_l = 0.0
            for obs in vt.observations:
                k = time_key(obs.time)
                if k not in keyed:
                    raise KeyError(f"Model output missing time {obs.time} for variable {name}")
                m = keyed[k]
                if not math.isfinite(m):
                    raise ValueError(f"Model output contains non-finite value for {name} at time {obs.time}")
                residual = obs.value - m
                chi = (residual * residual) / (obs.uncertainty * obs.uncertainty)
                l = -0.5 * chi
                if self.normalize:
                    l -= 0.5 * math.log(2.0 * math.pi)
                    l -= math.log(obs.uncertainty)
                ln_l += l
            total += ln_l
        return total


# ------------------------------------------------------------------------------------ runner
class FactoryModelRunner:
    """``ModelRunner(model_factory=...)`` of the reference's Python front (crates/rscm-calibrate/src/python/
    model_runner.rs): any Python callable ``param_dict -> {variable: {time: value}}`` as the model, run on
    the host, one call per member.  The slow, general path -- e.g. the tutorial's quadratic toy model;
    models built from components go through the GPU ``ModelRunner(model_builder, ...)``."""

    def __init__(self, model_factory: Callable[[Dict[str, float]], Dict[str, Dict[float, float]]],
                 param_names: Sequence[str], output_variables: Sequence[str]):
        self._factory = model_factory
        self._param_names = list(param_names)
        self._outputs = list(output_variables)

    @property
    def param_names(self) -> List[str]:
        return list(self._param_names)

    @property
    def output_variables(self) -> List[str]:
        return list(self._outputs)

    def run(self, params: Sequence[float]) -> Dict[str, Dict[float, float]]:
        if len(params) != len(self._param_names):
            raise ValueError(f"Expected {len(self._param_names)} parameters, got {len(params)}")  # model_runner.rs:225-231
        out = self._factory(dict(zip(self._param_names, (float(x) for x in params))))
        missing = [v for v in self._outputs if v not in out]
        if missing:
            raise KeyError(f"Model output missing variable: {missing[0]}")
        return {v: {float(t): float(x) for t, x in out[v].items()} for v in self._outputs}

    def run_batch(self, param_sets) -> List[Dict[str, Dict[float, float]]]:
        return [self.run(list(p)) for p in param_sets]

    def log_likelihood_batch(self, param_sets, target: "Target", likelihood: "GaussianLikelihood") -> np.ndarray:
        out = np.full(len(param_sets), -np.inf)
        for k, p in enumerate(param_sets):
            try:  # a member that fails is -inf, the batch goes on (sampler/ensemble.rs:163-172)
                out[k] = likelihood.ln_likelihood(self.run(list(p)), target)
            except (KeyError, ValueError, ArithmeticError):
                pass
        out[np.isnan(out)] = -np.inf
        return out

    def close(self) -> None:
        pass


class ModelRunner:
    """GPU-batched ``ModelRunner`` (trait: model_runner.rs:38-85).

    ``model`` is a ``ModelBuilder`` describing the shared structure (axis, components, forcing,
    initial values); ``param_names`` name component parameters (``lambda0`` ... ``erf_2xco2``)
    that vary per member, in the order of the parameter vectors; every other parameter keeps the
    value of the builder's components.  One device ensemble is kept per batch size.
    """

    def __new__(cls, model=None, param_names=None, output_variables=None, *args, model_factory=None, **kwargs):
        if model_factory is not None or (model is not None and callable(model) and not isinstance(model, ModelBuilder)):
            return FactoryModelRunner(model_factory if model_factory is not None else model, param_names, output_variables)
        return super().__new__(cls)

    def __init__(self, model: ModelBuilder, param_names: Sequence[str],
                 output_variables: Sequence[str], mode: int = L.MODE_EXACT, execution_order: str = "reference"):
        self._builder = model
        self._param_names = list(param_names)
        self._outputs = list(output_variables)
        self._mode = mode
        self._models: Dict[int, Model] = {}
        self._lik_models: Dict[int, Model] = {}
        self._execution_order = execution_order
        probe = self._model(1)
        # a graph without a fused kernel: one linked ensemble per component (core.GraphModel); parameters
        # are addressed as "Component.name", or by the bare name where only one component has it
        self._graph = isinstance(probe, GraphModel)
        if self._graph:
            unknown = [p for p in self._param_names if p not in probe.param_home]
            if unknown:
                raise ValueError(f"unknown (or ambiguous) model parameter(s) {unknown}; known: {sorted(probe.param_home)}")
            for v in self._outputs:
                probe.variable_home(v)
            self._rows = []
            return
        unknown = [p for p in self._param_names if p not in probe.param_order]
        if unknown:
            raise ValueError(f"unknown model parameter(s) {unknown}; known: {list(probe.param_order)}")
        missing = [v for v in self._outputs if v not in probe.ensemble.var_ids]
        if missing:
            raise KeyError(f"Model output missing variable: {missing[0]}")
        self._rows = [probe.param_order.index(p) for p in self._param_names]

    @property
    def param_names(self) -> List[str]:
        return list(self._param_names)

    @property
    def output_variables(self) -> List[str]:
        return list(self._outputs)

    def _model(self, n: int) -> Model:
        if n not in self._models:
            if len(self._models) >= 4:  # bounded cache of device ensembles
                self._models.pop(next(iter(self._models))).close()
            m = self._builder.build(n_members=n, execution_order=self._execution_order)
            if isinstance(m, GraphModel):
                m.set_mode(self._mode)
            else:
                m.ensemble.set_mode(self._mode)
            self._models[n] = m
        return self._models[n]

    def _lik_model(self, n: int) -> Model:
        """Likelihood-only ensemble: no series buffers (RSCM_FLAG_NO_SERIES)."""
        if n not in self._lik_models:
            if len(self._lik_models) >= 4:
                self._lik_models.pop(next(iter(self._lik_models))).close()
            m = self._builder.build(n_members=n, store_series=False)
            m.ensemble.set_mode(self._mode)
            self._lik_models[n] = m
        return self._lik_models[n]

    def _load(self, m: Model, param_sets) -> None:
        p = np.asarray(param_sets, dtype=np.float64)
        if p.ndim != 2 or p.shape[1] != len(self._param_names):
            got = p.shape[1] if p.ndim == 2 else len(p)
            raise ValueError(f"Expected {len(self._param_names)} parameters, got {got}")  # :225-231
        if self._graph:
            m.set_member_params(self._param_names, p)
            m.rewind()
            return
        full = np.repeat(m.base_params[:, None], p.shape[0], axis=1)
        full[self._rows, :] = p.T
        m.ensemble.set_params(full)
        m.ensemble.rewind()

    def _run(self, param_sets: np.ndarray) -> Model:
        p = np.asarray(param_sets, dtype=np.float64)
        if p.ndim != 2:
            raise ValueError(f"Expected {len(self._param_names)} parameters, got {len(p)}")
        m = self._model(p.shape[0])
        self._load(m, p)
        if self._graph:
            m.run()
        else:
            m.ensemble.run()
        return m

    def run(self, params: Sequence[float]) -> Dict[str, Dict[float, float]]:
        return self.run_batch([list(params)])[0]

    def run_batch(self, param_sets) -> List[Dict[str, Dict[float, float]]]:
        """Order-preserving; per member ``{variable: {time: value}}`` of the non-NaN entries
        (extract_outputs, model_runner.rs:161-212)."""
        if len(param_sets) == 0:  # run_batch(&[]) -> vec![]
            return []
        m = self._run(np.asarray(param_sets, dtype=np.float64))
        times = m._axis.values()
        series = {v: (m.get_series(v) if self._graph else m.ensemble.get_series(v)) for v in self._outputs}
        out = []
        for i in range(len(param_sets)):
            member = {}
            for v, s in series.items():
                col = s[:, i]
                ok = ~np.isnan(col)
                member[v] = {float(t): float(x) for t, x in zip(times[ok], col[ok])}
            out.append(member)
        return out

    def log_likelihood_batch(self, param_sets, target: Target,
                             likelihood: GaussianLikelihood) -> np.ndarray:
        """Device path: run the batch and reduce the Gaussian log-likelihood per member without
        moving any series to the host.  Failed members (non-finite or never-computed values at an
        observation time) get ``-inf`` (sampler/ensemble.rs:163-172)."""
        p = np.asarray(param_sets, dtype=np.float64)
        if p.size == 0:
            return np.zeros(0)
        if p.ndim != 2:
            raise ValueError(f"Expected {len(self._param_names)} parameters, got {len(p)}")
        probe = self._model(1)
        ov, ot, val, sig = [], [], [], []
        fused = not self._graph and probe.ensemble.kind == L.KIND_TWO_LAYER
        for name, vt in target.variables():
            if self._graph:
                probe.variable_home(name)
            elif name not in probe.ensemble.var_ids or probe.ensemble.var_ids[name] == 0:
                raise KeyError(f"Model output missing variable: {name}")
            prev = -1
            for obs in vt.observations:
                idx = probe._axis.index_of(obs.time)
                if idx is None:  # "Model output missing time" -> Err -> -inf for every member
                    return np.full(p.shape[0], -np.inf)
                fused = fused and idx >= prev
                prev = idx
                ov.append(name)
                ot.append(idx)
                val.append(obs.value)
                sig.append(obs.uncertainty)
        # Multi-GPU: every rank holds the same batch (same seeds => same proposals, nothing to
        # scatter), evaluates its contiguous block of members and all-gathers 8 B per member.
        from .distributed import gather_members, is_distributed, shard_bounds
        n_total = p.shape[0]
        if is_distributed():
            import torch.distributed as dist
            off, cnt = shard_bounds(n_total, dist.get_rank(), dist.get_world_size())
            p = p[off:off + cnt]
        if p.shape[0] == 0:
            local = np.zeros(0)
        elif fused:  # run + likelihood in one kernel, nothing written to HBM but ln L
            m = self._lik_model(p.shape[0])
            self._load(m, p)
            local = m.ensemble.run_loglik(ov, ot, val, sig, likelihood.normalize, on_device=is_distributed())
        elif self._graph:
            # observations grouped by the ensemble that holds their variable, in the target's order;
            # a member that fails anywhere is -inf (-inf + finite)
            m = self._run(p)
            local = np.zeros(p.shape[0])
            groups: Dict[int, Tuple[object, list]] = {}
            for k, name in enumerate(ov):
                ens, vid = m.variable_home(name)
                groups.setdefault(id(ens), (ens, []))[1].append((vid, ot[k], val[k], sig[k]))
            for ens, obs in groups.values():
                v, t_, x, s = zip(*obs)
                local = local + ens.loglik(list(v), list(t_), list(x), list(s), likelihood.normalize)
        else:
            m = self._run(p)
            local = m.ensemble.loglik(ov, ot, val, sig, likelihood.normalize, on_device=is_distributed())
        return gather_members(local, n_total) if is_distributed() else local

    def close(self) -> None:
        for m in list(self._models.values()) + list(self._lik_models.values()):
            m.close()
        self._models.clear()
        self._lik_models.clear()


# ------------------------------------------------------------------------------------ sampler
class ProgressTracker:
    """A progress callback that keeps the history (python/rscm/calibrate/progress.py of the reference):
    ``iterations``, ``acceptance_rates``, ``mean_log_probs``."""

    def __init__(self) -> None:
        self.iterations: List[int] = []
        self.acceptance_rates: List[float] = []
        self.mean_log_probs: List[float] = []

    def __call__(self, iteration: int, acceptance_rate: float, mean_log_prob: float) -> None:
        self.iterations.append(int(iteration))
        self.acceptance_rates.append(float(acceptance_rate))
        self.mean_log_probs.append(float(mean_log_prob))


class WalkerInit:
    """sampler/init.rs:40-98."""

    def __init__(self, kind: str, **kw):
        self.kind, self.kw = kind, kw

    @staticmethod
    def from_prior() -> "WalkerInit":
        return WalkerInit("prior")

    @staticmethod
    def ball(center: Sequence[float], radius: float) -> "WalkerInit":
        return WalkerInit("ball", center=np.asarray(center, dtype=float), radius=float(radius))

    @staticmethod
    def explicit(positions) -> "WalkerInit":
        return WalkerInit("explicit", positions=np.asarray(positions, dtype=float))

    def initialize(self, n_walkers: int, params: ParameterSet, rng) -> np.ndarray:
        if self.kind == "prior":
            return params.sample_random(n_walkers, rng)
        if self.kind == "ball":
            c = self.kw["center"]
            if len(c) != len(params):
                raise ValueError(f"Ball center length {len(c)} does not match parameter count {len(params)}")
            return c[None, :] + (rng.random((n_walkers, len(params))) - 0.5) * self.kw["radius"]
        pos = self.kw["positions"]
        if pos.shape != (n_walkers, len(params)):
            raise ValueError(f"Explicit positions have shape {pos.shape}, expected {(n_walkers, len(params))}")
        return pos.copy()


class Chain:
    """sampler/chain.rs + diagnostics.rs:39-160 (split-chain R-hat)."""

    def __init__(self, param_names: Sequence[str], thin: int):
        self.param_names = list(param_names)
        self.thin = max(1, int(thin))
        self.total_iterations = 0
        self._samples: List[np.ndarray] = []
        self._log_probs: List[np.ndarray] = []

    def push(self, positions: np.ndarray, log_probs: np.ndarray) -> bool:
        self.total_iterations += 1
        if (self.total_iterations - 1) % self.thin == 0:
            self._samples.append(positions.copy())
            self._log_probs.append(log_probs.copy())
            return True
        return False

    def __len__(self) -> int:
        return len(self._samples)

    def flat_samples(self, discard: int = 0) -> np.ndarray:
        if not self._samples or discard >= len(self):
            return np.zeros((0, len(self.param_names)))
        return np.concatenate(self._samples[discard:], axis=0)

    def flat_log_probs(self, discard: int = 0) -> np.ndarray:
        if not self._samples or discard >= len(self):
            return np.zeros(0)
        return np.concatenate(self._log_probs[discard:])

    def to_dataframe(self, discard: int = 0):
        """One row per kept sample (walker-major within a sweep): the parameters and ``log_prob``."""
        import pandas as pd
        df = pd.DataFrame(self.flat_samples(discard), columns=self.param_names)
        df["log_prob"] = self.flat_log_probs(discard)
        return df

    def to_param_dict(self, discard: int = 0) -> Dict[str, np.ndarray]:
        flat = self.flat_samples(discard)
        return {n: flat[:, j] for j, n in enumerate(self.param_names)}

    def r_hat(self, discard: int = 0) -> Dict[str, float]:
        if not self._samples or discard >= len(self) or len(self) - discard < 4:
            return {}
        x = np.stack(self._samples[discard:])  # [keep][walkers][params]
        n_split = x.shape[0] // 2
        halves = np.concatenate([x[:n_split], x[n_split:2 * n_split]], axis=1)  # [n_split][2W][P]
        means = halves.mean(axis=0)
        var = halves.var(axis=0, ddof=1)
        w = var.mean(axis=0)
        b = n_split * ((means - means.mean(axis=0)) ** 2).sum(axis=0) / (halves.shape[1] - 1)
        var_plus = ((n_split - 1) * w + b) / n_split
        with np.errstate(all="ignore"):
            r = np.sqrt(var_plus / w)
        return {n: float(r[j]) for j, n in enumerate(self.param_names)}

    def _autocorr_sums(self, discard: int) -> Dict[str, float]:
        """Per parameter: the sum of the walker-averaged autocorrelations at lags 1, 2, ... up to the
        first non-positive one (diagnostics.rs:164-252, compute_autocorrelation :302-335)."""
        if not self._samples or discard >= len(self) or len(self) - discard < 10:
            return {}
        x = np.stack(self._samples[discard:])  # [keep][walkers][params]
        n = x.shape[0]
        max_lag = min(n // 2, 100)
        out = {}
        for j, name in enumerate(self.param_names):
            c = x[:, :, j]
            d = c - c.mean(axis=0)
            var = (d * d).sum(axis=0) / n
            avg = np.zeros(max_lag)
            for lag in range(1, max_lag + 1):
                cov = (d[: n - lag] * d[lag:]).sum(axis=0) / (n - lag)
                with np.errstate(all="ignore"):
                    ac = np.where(var == 0.0, 0.0, cov / var)
                avg[lag - 1] = ac.mean()  # each walker weighs 1 / n_walkers
            total = 0.0
            for ac in avg:
                if ac <= 0.0:
                    break
                total += ac
            out[name] = total
        return out

    def ess(self, discard: int = 0) -> Dict[str, float]:
        """Effective sample size, ``n_total / (1 + 2 sum rho_k)`` (diagnostics.rs:164-221)."""
        sums = self._autocorr_sums(discard)
        if not sums:
            return {}
        n_total = (len(self) - discard) * self._samples[0].shape[0]
        return {k: n_total / (1.0 + 2.0 * s) for k, s in sums.items()}

    def autocorr_time(self, discard: int = 0) -> Dict[str, float]:
        """Integrated autocorrelation time ``1 + 2 sum rho_k`` (diagnostics.rs:254-300)."""
        return {k: 1.0 + 2.0 * s for k, s in self._autocorr_sums(discard).items()}

    def save(self, path) -> None:
        """chain.rs:190-203.  The reference writes postcard; this writes a NumPy archive."""
        with open(path, "wb") as fh:
            np.savez(fh, param_names=np.array(self.param_names), thin=self.thin, total_iterations=self.total_iterations,
                     samples=np.stack(self._samples) if self._samples else np.zeros((0, 0, len(self.param_names))),
                     log_probs=np.stack(self._log_probs) if self._log_probs else np.zeros((0, 0)))

    @staticmethod
    def load(path) -> "Chain":
        if os.path.getsize(path) > 1 << 30:  # chain.rs:218-231 MAX_CHAIN_FILE_SIZE
            raise ValueError(f"Chain file too large: {os.path.getsize(path)} bytes (max {1 << 30} bytes)")
        with np.load(path, allow_pickle=False) as z:
            c = Chain([str(s) for s in z["param_names"]], int(z["thin"]))
            c.total_iterations = int(z["total_iterations"])
            c._samples = [a.copy() for a in z["samples"]]
            c._log_probs = [a.copy() for a in z["log_probs"]]
        return c

    def merge(self, other: "Chain") -> None:
        """chain.rs:256-277."""
        if self.param_names != other.param_names:
            raise ValueError(f"Cannot merge chains with different parameter names: {self.param_names} vs {other.param_names}")
        if self.thin != other.thin:
            raise ValueError(f"Cannot merge chains with different thinning intervals: {self.thin} vs {other.thin}")
        self._samples += [s.copy() for s in other._samples]
        self._log_probs += [s.copy() for s in other._log_probs]
        self.total_iterations += other.total_iterations

    def is_converged(self, discard: int = 0, threshold: float = 1.1) -> bool:
        r = self.r_hat(discard)
        return bool(r) and all(math.isfinite(v) and v < threshold for v in r.values())


class SamplerState:
    """sampler/state.rs: walker positions, their log-probabilities and acceptance counters."""

    def __init__(self, positions, param_names: Sequence[str]):
        positions = np.array(positions, dtype=np.float64)
        if positions.ndim != 2 or len(param_names) != positions.shape[1]:
            raise ValueError(f"Number of parameter names ({len(param_names)}) does not match positions dimension "
                             f"({positions.shape[1] if positions.ndim == 2 else positions.shape})")
        if positions.shape[0] < 2:
            raise ValueError("Must have at least 2 walkers for ensemble sampling")
        self.positions = positions
        self.log_probs = np.full(positions.shape[0], -np.inf)
        self.n_accepted = np.zeros(positions.shape[0], dtype=np.int64)
        self.n_proposed = np.zeros(positions.shape[0], dtype=np.int64)
        self.param_names = list(param_names)

    def n_walkers(self) -> int:
        return self.positions.shape[0]

    def n_params(self) -> int:
        return self.positions.shape[1]

    def acceptance_fraction(self) -> np.ndarray:
        with np.errstate(all="ignore"):
            return np.where(self.n_proposed > 0, self.n_accepted / np.maximum(self.n_proposed, 1), 0.0)

    def mean_acceptance_rate(self) -> float:
        return float(self.n_accepted.sum() / self.n_proposed.sum()) if self.n_proposed.sum() > 0 else 0.0

    def save_checkpoint(self, path) -> None:
        with open(path, "wb") as fh:
            np.savez(fh, positions=self.positions, log_probs=self.log_probs, n_accepted=self.n_accepted,
                     n_proposed=self.n_proposed, param_names=np.array(self.param_names))

    @staticmethod
    def load_checkpoint(path) -> "SamplerState":
        if os.path.getsize(path) > 1 << 30:
            raise ValueError(f"Checkpoint file too large: {os.path.getsize(path)} bytes (max {1 << 30} bytes)")
        with np.load(path, allow_pickle=False) as z:
            s = SamplerState(z["positions"], [str(n) for n in z["param_names"]])
            s.log_probs = z["log_probs"].copy()
            s.n_accepted = z["n_accepted"].copy()
            s.n_proposed = z["n_proposed"].copy()
        return s


class EnsembleSampler:
    """Affine-invariant stretch-move sampler (sampler/ensemble.rs:106-547, sampler/moves.rs)."""

    def __init__(self, params: ParameterSet, runner: ModelRunner, likelihood: GaussianLikelihood,
                 target: Target, stretch_a: float = 2.0):
        if stretch_a <= 1.0:
            raise ValueError(f"Stretch move scale parameter must be > 1.0, got {stretch_a}")
        self.params, self.runner, self.likelihood, self.target = params, runner, likelihood, target
        self.a = float(stretch_a)
        self.default_n_walkers = max(2 * len(params), 32)
        self.n_accepted = None
        self.n_proposed = None

    def log_posterior_batch(self, positions: np.ndarray) -> np.ndarray:
        """log prior + device log-likelihood; anything failing is -inf (ensemble.rs:143-177)."""
        lp = self.params.log_prior_batch(positions)
        outside = ~np.isfinite(lp)
        if outside.any() and not outside.all():
            # rejected whatever the model says: evaluate a valid walker in their place, so that
            # garbage parameters do not push wavefronts onto the kernel's slow replay path
            positions = np.array(positions, dtype=np.float64)
            positions[outside] = positions[np.flatnonzero(~outside)[0]]
        ll = self.runner.log_likelihood_batch(positions, self.target, self.likelihood)
        with np.errstate(invalid="ignore"):
            out = lp + ll
        out[~np.isfinite(lp) | np.isnan(out)] = -np.inf
        return out

    def _update_group(self, pos, logp, active, comp, rng):
        n_active, n_params = len(active), pos.shape[1]
        z = ((self.a - 1.0) * rng.random(n_active) + 1.0) ** 2 / self.a     # moves.rs:55-59
        c = pos[comp[rng.integers(0, len(comp), n_active)]]                  # moves.rs:118-121
        proposals = c + z[:, None] * (pos[active] - c)                        # y = c + z (x - c)
        new_lp = self.log_posterior_batch(proposals)
        with np.errstate(all="ignore"):
            log_ratio = (n_params - 1.0) * np.log(z) + (new_lp - logp[active])
            prob = np.minimum(np.exp(log_ratio), 1.0)
        prob[~np.isfinite(new_lp)] = 0.0                                      # moves.rs:84-87
        accept = rng.random(n_active) < prob
        self.n_proposed[active] += 1
        self.n_accepted[active[accept]] += 1
        pos[active[accept]] = proposals[accept]
        logp[active[accept]] = new_lp[accept]

    def run(self, n_iterations: int, init: WalkerInit, thin: int = 1,
            n_walkers: Optional[int] = None, rng: Optional[np.random.Generator] = None,
            progress: Optional[Callable[[int, float, float], None]] = None) -> Chain:
        n_walkers = n_walkers or self.default_n_walkers
        if n_walkers < 2:
            raise ValueError("Must have at least 2 walkers")
        if n_walkers % 2:
            raise ValueError("Number of walkers must be even")
        rng = rng or np.random.default_rng()
        pos = init.initialize(n_walkers, self.params, rng)
        logp = self.log_posterior_batch(pos)
        self.n_accepted = np.zeros(n_walkers, dtype=np.int64)
        self.n_proposed = np.zeros(n_walkers, dtype=np.int64)
        chain = Chain(self.params.param_names, thin)
        half = n_walkers // 2
        first, second = np.arange(half), np.arange(half, n_walkers)
        for it in range(n_iterations):
            self._update_group(pos, logp, first, second, rng)
            self._update_group(pos, logp, second, first, rng)
            chain.push(pos, logp)
            if progress:
                progress(it, float(self.n_accepted.sum() / max(1, self.n_proposed.sum())),
                         float(logp.mean()))
        return chain

    def acceptance_rate(self) -> float:
        return float(self.n_accepted.sum() / max(1, self.n_proposed.sum()))

    def run_with_progress(self, n_iterations: int, init: WalkerInit, thin: int = 1, progress_callback=None,
                          n_walkers: Optional[int] = None, rng: Optional[np.random.Generator] = None) -> Chain:
        """``run`` reporting ``(iteration, acceptance rate so far, mean log probability)`` after every sweep."""
        return self.run(n_iterations, init, thin, n_walkers=n_walkers, rng=rng, progress=progress_callback)

    # -- checkpointed runs (ensemble.rs:272-410, 548-660) -----------------------------------------
    def run_with_checkpoint(self, n_iterations: int, init: WalkerInit, thin: int, checkpoint_every: int,
                            checkpoint_path, progress=None, n_walkers: Optional[int] = None,
                            rng: Optional[np.random.Generator] = None) -> Chain:
        n_walkers = n_walkers or self.default_n_walkers
        if n_walkers < 2:
            raise ValueError("Must have at least 2 walkers")
        if n_walkers % 2:
            raise ValueError("Number of walkers must be even")
        rng = rng or np.random.default_rng()
        state = SamplerState(init.initialize(n_walkers, self.params, rng), self.params.param_names)
        return self._run_from_state(state, Chain(self.params.param_names, thin), n_iterations, checkpoint_every,
                                    checkpoint_path, progress, rng)

    def resume_from_checkpoint(self, n_iterations: int, thin: int, checkpoint_every: int, checkpoint_path,
                               progress=None, rng: Optional[np.random.Generator] = None) -> Chain:
        """``n_iterations`` is the total wanted: what the saved chain already holds is not repeated."""
        state = SamplerState.load_checkpoint(f"{checkpoint_path}.state")
        chain = Chain.load(f"{checkpoint_path}.chain")
        remaining = max(0, n_iterations - chain.total_iterations)
        if remaining == 0:
            return chain
        return self._run_from_state(state, chain, remaining, checkpoint_every, checkpoint_path, progress,
                                    rng or np.random.default_rng())

    def _run_from_state(self, state: SamplerState, chain: Chain, n_iterations: int, checkpoint_every: int,
                        checkpoint_path, progress, rng) -> Chain:
        if not np.isfinite(state.log_probs).any():  # not computed yet
            state.log_probs = self.log_posterior_batch(state.positions)
        self.n_accepted, self.n_proposed = state.n_accepted, state.n_proposed
        half = state.n_walkers() // 2
        first, second = np.arange(half), np.arange(half, state.n_walkers())
        for it in range(n_iterations):
            self._update_group(state.positions, state.log_probs, first, second, rng)
            self._update_group(state.positions, state.log_probs, second, first, rng)
            chain.push(state.positions, state.log_probs)
            if checkpoint_every > 0 and (it + 1) % checkpoint_every == 0:
                state.save_checkpoint(f"{checkpoint_path}.state")
                chain.save(f"{checkpoint_path}.chain")
            if progress:
                progress(it, state.mean_acceptance_rate(), float(state.log_probs.mean()))
        return chain


class DeviceEnsembleSampler:
    """``EnsembleSampler`` with the whole stretch-move loop on the GPU (``rscm_sampler_*`` of the C
    ABI, csrc/sampler.hip): proposals, priors, the fused run+likelihood of every half-ensemble and
    the accept step never leave the device, the host only fetches the walker positions it wants
    to keep.  Same interface as ``EnsembleSampler``; priors must be ``Uniform`` or ``Normal``.  A
    two-layer model is scored by the fused run+likelihood kernel, any other kind (the coupled
    chain, ClimateUDEB, ...) is run and scored from its stored series, still on the device; a graph
    of linked ensembles (``ModelRunner`` over a ``GraphModel``) is the evaluator as a whole
    (``rscm_sampler_create_graph``): every half-step the proposal kernel writes each proposed value into
    the parameter block of the component that owns it, the graph is stepped in lock-step to the last
    observed index and the likelihood kernel reads the observations where their owners store them.
    Random numbers are counter-based from ``seed``."""

    def __init__(self, params: ParameterSet, runner: ModelRunner, likelihood: GaussianLikelihood,
                 target: Target, stretch_a: float = 2.0):
        if stretch_a <= 1.0:
            raise ValueError(f"Stretch move scale parameter must be > 1.0, got {stretch_a}")
        if list(params.param_names) != runner.param_names:
            raise ValueError("the parameter set must name the runner's parameters, in its order")
        self.params, self.runner, self.likelihood, self.target = params, runner, likelihood, target
        self.a = float(stretch_a)
        self.default_n_walkers = max(2 * len(params), 32)
        self.n_accepted = None
        self.n_proposed = None
        self.device_ms = 0.0
        self.exchange_ms = 0.0
        self.exchange_bytes_per_half_step = 0
        kinds, pa, pb, plo, phi = [], [], [], [], []
        for d in params.distributions():
            lo, hi = -math.inf, math.inf
            if isinstance(d, Bound):
                lo, hi, d = d.low, d.high, d.inner()
            if isinstance(d, Uniform):
                kinds.append(0), pa.append(d.low), pb.append(d.high)
            elif isinstance(d, Normal):
                kinds.append(1), pa.append(d.mean), pb.append(d.std_dev)
            elif isinstance(d, LogNormal):
                kinds.append(2), pa.append(d.mu), pb.append(d.sigma)
            else:
                raise NotImplementedError(f"prior {type(d).__name__} has no device form")
            plo.append(lo), phi.append(hi)
        self._prior = (np.array(kinds, dtype=np.int32), L.f64(pa), L.f64(pb), L.f64(plo), L.f64(phi))

    def _observations(self, model: Model):
        ov, ot, val, sig = [], [], [], []
        for name, vt in self.target.variables():
            if name not in model.ensemble.var_ids or model.ensemble.var_ids[name] == 0:
                raise KeyError(f"Model output missing variable: {name}")
            for obs in vt.observations:
                idx = model._axis.index_of(obs.time)
                if idx is None:
                    raise KeyError(f"Model output missing time: {obs.time}")
                ov.append(model.ensemble.var_ids[name]), ot.append(idx), val.append(obs.value), sig.append(obs.uncertainty)
        return np.array(ov, dtype=np.int32), np.array(ot, dtype=np.int32), L.f64(val), L.f64(sig)

    def run(self, n_iterations: int, init: "WalkerInit", thin: int = 1, n_walkers: Optional[int] = None,
            rng: Optional[np.random.Generator] = None, seed: int = 0, n_groups: int = 1,
            shard: Optional[bool] = None) -> "Chain":
        """``n_groups`` > 1 runs that many independent ensembles of ``n_walkers / n_groups`` walkers
        side by side (consecutive blocks of the walker index), each a sampler of its own.
        ``shard``: split the walkers over the ranks of the initialised ``torch.distributed`` group
        (default: whenever there is one); the chain is the same for any number of ranks."""
        import ctypes as C
        self.exchange_ms, self.exchange_bytes_per_half_step = 0.0, 0
        n_walkers = n_walkers or self.default_n_walkers * n_groups
        if n_walkers < 2:
            raise ValueError("Must have at least 2 walkers")
        if n_walkers % 2:
            raise ValueError("Number of walkers must be even")
        rng = rng or np.random.default_rng(seed)
        pos = L.f64(init.initialize(n_walkers, self.params, rng))
        # One process per GPU (torch.distributed initialised): every rank holds a replica of the walkers and
        # owns a block of both halves; the blocks are all-gathered after every half-step.  All ranks must be
        # given the same initial positions (the same `rng` / `seed`), as with the host sampler.
        from .distributed import is_distributed
        rank, world = 0, 1
        if is_distributed() and shard is not False:
            import torch.distributed as dist
            rank, world = dist.get_rank(), dist.get_world_size()
            if (n_walkers // 2) % world:
                raise ValueError(f"half the walkers ({n_walkers // 2}) must split evenly over {world} ranks")
            if n_groups != 1:
                raise ValueError("a sharded sampler runs one ensemble (n_groups = 1)")
        n_eval = n_walkers // 2 // world
        kinds, pa, pb, plo, phi = self._prior
        h = C.c_void_p()
        if getattr(self.runner, "_graph", False):
            model = self.runner._model(n_eval)
            if model._host_nodes:
                raise NotImplementedError("a graph with Python components steps through the host: calibrate it with EnsembleSampler")
            model.rewind()
            order = list(model._order)
            handles = [model.ensembles[name] for name in order]
            ens, lib = handles[0], handles[0]._lib
            # the non-sampled parameters: the builder's values for every member (the proposal kernel overwrites the sampled rows)
            for owner in {model.param_home[p][0] for p in self.runner.param_names}:
                model.ensembles[owner].set_params(np.repeat(model.base_params[owner][:, None], n_eval, axis=1))
            p_owner = np.array([order.index(model.param_home[p][0]) for p in self.runner.param_names], dtype=np.int32)
            p_rows = np.array([model.param_home[p][1] for p in self.runner.param_names], dtype=np.int32)
            oo, ov, ot, val, sig = [], [], [], [], []
            for name, vt in self.target.variables():
                owner_ens, vid = model.variable_home(name)
                k = next(j for j, e in enumerate(handles) if e is owner_ens)
                for obs in vt.observations:
                    idx = model._axis.index_of(obs.time)
                    if idx is None:
                        raise KeyError(f"Model output missing time: {obs.time}")
                    oo.append(k), ov.append(vid), ot.append(idx), val.append(obs.value), sig.append(obs.uncertainty)
            oo, ov, ot = (np.array(x, dtype=np.int32) for x in (oo, ov, ot))
            val, sig = L.f64(val), L.f64(sig)
            arr = (C.c_void_p * len(handles))(*[e._h.value for e in handles])
            L.check(lib.rscm_sampler_create_graph(arr, len(handles), 1 if model._reads_unwritten else 0, n_walkers, len(p_rows),
                                                  L.iptr(p_owner), L.iptr(p_rows), L.iptr(kinds), L.dptr(pa), L.dptr(pb), L.dptr(plo),
                                                  L.dptr(phi), len(ov), L.iptr(oo), L.iptr(ov), L.iptr(ot), L.dptr(val), L.dptr(sig),
                                                  1 if self.likelihood.normalize else 0, self.a,
                                                  C.c_uint64(seed & 0xFFFFFFFFFFFFFFFF), rank, world, C.byref(h)))
        else:
            two_layer = self.runner._model(1).ensemble.kind == L.KIND_TWO_LAYER
            model = self.runner._lik_model(n_eval) if two_layer else self.runner._model(n_eval)
            ens, lib = model.ensemble, model.ensemble._lib
            ens.rewind()
            ov, ot, val, sig = self._observations(model)
            rows = np.array(self.runner._rows, dtype=np.int32)
            base = L.f64(model.base_params)
            L.check(lib.rscm_sampler_create_sharded(ens._h, n_walkers, len(rows), L.iptr(rows), L.dptr(base), L.iptr(kinds),
                                                    L.dptr(pa), L.dptr(pb), L.dptr(plo), L.dptr(phi), len(ov), L.iptr(ov), L.iptr(ot),
                                                    L.dptr(val), L.dptr(sig), 1 if self.likelihood.normalize else 0, self.a,
                                                    C.c_uint64(seed & 0xFFFFFFFFFFFFFFFF), rank, world, C.byref(h)))
        try:
            if world > 1 or (is_distributed() and shard is not False):
                return self._run_sharded(lib, h, ens, pos, n_iterations, thin, n_walkers, world)
            if n_groups != 1:
                L.check(lib.rscm_sampler_set_groups(h, n_groups))
            L.check(lib.rscm_sampler_set_positions(h, L.dptr(pos)))
            chain = Chain(self.params.param_names, thin)
            logp = np.empty(n_walkers)
            self.device_ms = 0.0
            kept = lambda k: (k - 1) % chain.thin == 0  # noqa: E731  (Chain.push keeps 1, 1+thin, ...)
            it = 0
            while it < n_iterations:  # iterate on the device up to the next sweep the chain keeps
                nxt = it + 1
                while not kept(nxt):
                    nxt += 1
                step = min(nxt, n_iterations) - it
                L.check(lib.rscm_sampler_iterate(h, step))
                ms = C.c_float()
                L.check(lib.rscm_sampler_last_ms(h, C.byref(ms)))
                self.device_ms += ms.value
                it += step
                if kept(it):
                    L.check(lib.rscm_sampler_get(h, L.dptr(pos), L.dptr(logp), None, None))
                    chain._samples.append(pos.copy())
                    chain._log_probs.append(logp.copy())
            chain.total_iterations = n_iterations
            acc = np.empty(n_walkers, dtype=np.int64)
            prop = np.empty(n_walkers, dtype=np.int64)
            L.check(lib.rscm_sampler_get(h, None, None, acc.ctypes.data_as(C.POINTER(C.c_int64)),
                                         prop.ctypes.data_as(C.POINTER(C.c_int64))))
            self.n_accepted, self.n_proposed = acc, prop
        finally:
            lib.rscm_sampler_destroy(h)
        return chain

    def _run_sharded(self, lib, h, ens, pos, n_iterations: int, thin: int, n_walkers: int, world: int) -> "Chain":
        """The sweep loop of a sharded sampler (rscm_sampler_create_sharded): per half-step the
        rank's block is updated on its GPU, the blocks are all-gathered (RCCL: device to device on the
        evaluator's stream order; gloo rehearsals: through the host) and unpacked into every replica."""
        import ctypes as C
        import time
        import torch
        import torch.distributed as dist
        send, recv, n = C.c_void_p(), C.c_void_p(), C.c_int64()
        L.check(lib.rscm_sampler_exchange_buffers(h, C.byref(send), C.byref(recv), C.byref(n)))   # also switches the sampler to driven mode
        per = n.value
        on_device = dist.get_backend() == "nccl"
        if on_device:
            from .ensemble import DeviceVector
            dev = torch.device("cuda", torch.cuda.current_device())
            t_send = torch.as_tensor(DeviceVector(send.value, per, np.float64, ens), device=dev)
            t_recv = torch.as_tensor(DeviceVector(recv.value, per * world, np.float64, ens), device=dev)
        else:
            h_send, h_recv = np.empty(per), np.empty(per * world)
            t_send, t_recv = torch.from_numpy(h_send), torch.from_numpy(h_recv)

        self.exchange_ms = 0.0   # host clock from "this rank's block is packed" to "every rank's block has landed here": the
        #                          all-gather itself plus the wait for the slowest rank's half-step

        def half_step(half: int, identity: int) -> None:
            L.check(lib.rscm_sampler_half_step(h, half, identity))
            L.check(lib.rscm_sampler_sync(h))        # the packed block is complete before the collective reads it
            t_x = time.perf_counter()
            if not on_device:
                L.check(lib.rscm_gpu_copy_to_host(ens.device, h_send.ctypes.data_as(C.c_void_p), send, h_send.nbytes))
            dist.all_gather_into_tensor(t_recv, t_send)
            if on_device:
                torch.cuda.current_stream().synchronize()   # the gathered blocks have landed before the unpack launch
            else:
                L.check(lib.rscm_gpu_copy_to_device(ens.device, recv, h_recv.ctypes.data_as(C.c_void_p), h_recv.nbytes))
            self.exchange_ms += (time.perf_counter() - t_x) * 1e3
            L.check(lib.rscm_sampler_apply_exchange(h, half))

        L.check(lib.rscm_sampler_set_positions(h, L.dptr(pos)))
        for half in (0, 1):
            half_step(half, 1)
        self.exchange_ms = 0.0   # the identity half-steps above scored the initial positions: not part of the sweeps
        self.exchange_bytes_per_half_step = per * 8 * world
        chain = Chain(self.params.param_names, thin)
        logp = np.empty(n_walkers)
        t0 = time.perf_counter()
        for it in range(1, n_iterations + 1):
            L.check(lib.rscm_sampler_begin_iteration(h))
            for half in (0, 1):
                half_step(half, 0)
            if (it - 1) % chain.thin == 0:  # Chain.push keeps sweeps 1, 1 + thin, ...
                L.check(lib.rscm_sampler_get(h, L.dptr(pos), L.dptr(logp), None, None))
                chain._samples.append(pos.copy())
                chain._log_probs.append(logp.copy())
        L.check(lib.rscm_sampler_sync(h))
        self.device_ms = (time.perf_counter() - t0) * 1e3   # wall time of the sweeps incl. the exchanges
        chain.total_iterations = n_iterations
        acc = np.empty(n_walkers, dtype=np.int64)
        prop = np.empty(n_walkers, dtype=np.int64)
        L.check(lib.rscm_sampler_get(h, None, None, acc.ctypes.data_as(C.POINTER(C.c_int64)),
                                     prop.ctypes.data_as(C.POINTER(C.c_int64))))
        # every rank counted its own walkers only: the sum over ranks is the whole ensemble's record
        counts = torch.from_numpy(np.stack([acc, prop]))
        if on_device:
            counts = counts.to(torch.device("cuda", torch.cuda.current_device()))
        dist.all_reduce(counts, op=dist.ReduceOp.SUM)
        counts = counts.cpu().numpy()
        self.n_accepted, self.n_proposed = counts[0], counts[1]
        return chain

    def acceptance_rate(self) -> float:
        return float(self.n_accepted.sum() / max(1, self.n_proposed.sum()))


# ------------------------------------------------------------------------------------ point estimate
class Optimizer:
    def __init__(self, kind: str):
        self.kind = kind

    @staticmethod
    def random_search() -> "Optimizer":
        return Optimizer("random_search")


class OptimizationResult:
    def __init__(self, best_params, best_ll, best_lp, n):
        self.best_params = list(best_params)
        self.best_log_likelihood = float(best_ll)
        self.best_log_posterior = float(best_lp)
        self.n_evaluations = int(n)
        self.converged = True  # "Random search always converges" (optimizer.rs:118)


class PointEstimator:
    """point_estimator.rs + optimizer.rs:85-124; the n_samples evaluations are one GPU batch."""

    def __init__(self, params: ParameterSet, runner: ModelRunner, likelihood: GaussianLikelihood,
                 target: Target):
        self.params, self.runner, self.likelihood, self.target = params, runner, likelihood, target
        self._evaluated_params: List[List[float]] = []
        self._evaluated_ll: List[float] = []

    @property
    def param_names(self) -> List[str]:
        return self.params.param_names

    @property
    def n_params(self) -> int:
        return len(self.params)

    @property
    def n_evaluations(self) -> int:
        return len(self._evaluated_params)

    def evaluated_params(self):
        return self._evaluated_params

    def evaluated_log_likelihoods(self):
        return self._evaluated_ll

    def evaluate_batch(self, samples: np.ndarray) -> np.ndarray:
        lp = self.params.log_prior_batch(samples)
        ll = self.runner.log_likelihood_batch(samples, self.target, self.likelihood)
        ll = np.where(np.isfinite(lp), ll, -np.inf)
        self._evaluated_params += [list(r) for r in samples]
        self._evaluated_ll += [float(x) for x in ll]
        with np.errstate(invalid="ignore"):
            post = lp + ll
        post[np.isnan(post)] = -np.inf
        return post

    def best(self):
        if not self._evaluated_ll:
            return None
        i = int(np.argmax(self._evaluated_ll))
        return self._evaluated_params[i], self._evaluated_ll[i]

    def optimize(self, optimizer: Optimizer, n_samples: int,
                 rng: Optional[np.random.Generator] = None) -> OptimizationResult:
        if optimizer.kind != "random_search":
            raise ValueError(f"unknown optimizer {optimizer.kind}")
        rng = rng or np.random.default_rng()
        lo, hi = self.params.bounds()  # optimizer.rs:127-: uniform within the prior bounds
        lo, hi = np.asarray(lo), np.asarray(hi)
        if not (np.isfinite(lo).all() and np.isfinite(hi).all()):
            samples = self.params.sample_random(n_samples, rng)
        else:
            samples = lo + rng.random((n_samples, len(lo))) * (hi - lo)
        start = len(self._evaluated_ll)
        post = self.evaluate_batch(samples)
        if not np.isfinite(post).any():
            raise RuntimeError("Random search found no valid samples")
        i = int(np.argmax(post))
        return OptimizationResult(samples[i], self._evaluated_ll[start + i], post[i], n_samples)
