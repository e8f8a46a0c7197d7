"""Pure-Python restatement of the reference's stepper semantics.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
anything under ``oracle/``; the product package ``rscm_amd`` never does.

This module restates, for SMALL cases (Python loops), the generic machinery that the fused
HIP kernels hard-code, so that the reference's own index-convention golden vectors can be
checked against it and the fused C oracle (``rscm_oracle.c``) can be checked against a generic
component-by-component execution:

* ``TimeAxis``                      crates/rscm-core/src/timeseries.rs:45-212
* ``interpolate`` Linear/Previous/Next + ``find_segment``
                                    crates/rscm-core/src/interpolate/strategies/{mod.rs:24-81,
                                    linear_spline.rs:33-96,previous.rs:58-80,next.rs:56-80}
* ``interpolate_into``              crates/rscm-core/src/timeseries.rs:586-609
* ``ModelBuilder.build`` variable-source classification, graph edges, collection init
                                    crates/rscm-core/src/model/builder.rs:418-860
* ``Model.step/run``                crates/rscm-core/src/model/runtime.rs:368-527
* window accessors                  crates/rscm-core/src/state/windows.rs:155-247
* RK4 (ode_solvers 0.6.1 ``Rk4``, crate source not in the reference tree; classical scheme)
                                    crates/rscm-core/src/ivp/mod.rs:73-102,245-253
* TwoLayer / CarbonCycle / CO2ERF   crates/rscm-two-layer/src/component.rs:159-251,
                                    crates/rscm-components/src/components/{carbon_cycle.rs:102-159,
                                    co2_erf.rs:57-80}
* Sum/Mean/Weighted aggregates      crates/rscm-core/src/schema.rs:760-802,886-901
* Latin hypercube                   crates/rscm-calibrate/src/parameter_set.rs:207-233

Python floats are IEEE binary64 and CPython never fuses a*b+c, so every expression below rounds
exactly like the Rust source it follows; ``math.exp``/``math.log`` call the platform libm like
Rust's ``f64::exp``/``ln``.
"""
from __future__ import annotations

import math
from bisect import bisect_left
from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Sequence, Tuple

NAN = float("nan")
GTC_PER_PPM = 2.13  # crates/rscm-components/src/constants.rs:37
T_THRESHOLD = 5e-3  # crates/rscm-core/src/ivp/mod.rs:73

EXOGENOUS, UPSTREAM_OUTPUT, OWN_STATE = "Exogenous", "UpstreamOutput", "OwnState"
INPUT, OUTPUT, STATE = "Input", "Output", "State"


# --------------------------------------------------------------------------- time axis
class TimeAxis:
    """timeseries.rs:45-212.  ``bounds`` has len()+1 entries; value i is the START of step i."""

    def __init__(self, bounds: Sequence[float]):
        b = [float(x) for x in bounds]
        assert all(b[i + 1] > b[i] for i in range(len(b) - 1)), "bounds must increase"
        self.bounds = b

    @classmethod
    def from_values(cls, values: Sequence[float]) -> "TimeAxis":
        v = [float(x) for x in values]
        assert len(v) >= 2
        step = v[-1] - v[-2]
        return cls(v + [v[-1] + step])

    @classmethod
    def from_bounds(cls, bounds: Sequence[float]) -> "TimeAxis":
        assert len(bounds) > 1
        return cls(bounds)

    def __len__(self) -> int:
        return len(self.bounds) - 1

    def values(self) -> List[float]:
        return self.bounds[: len(self)]

    def at(self, i: int) -> Optional[float]:
        return self.bounds[i] if 0 <= i < len(self) else None

    def at_bounds(self, i: int) -> Optional[Tuple[float, float]]:
        return (self.bounds[i], self.bounds[i + 1]) if 0 <= i < len(self) else None

    def contains(self, value: float) -> bool:
        return any(value == v for v in self.values())

    def index_of(self, value: float) -> Optional[int]:
        for i, v in enumerate(self.values()):  # linear scan, |dv| < 1e-10 (timeseries.rs:204-211)
            if abs(v - value) < 1e-10:
                return i
        return None


# --------------------------------------------------------------------------- interpolation
def _is_close(a: float, b: float) -> bool:
    # `is_close!` defaults (is_close crate, modelled on math.isclose): rel_tol 1e-9, abs_tol 0
    return math.isclose(a, b, rel_tol=1e-9, abs_tol=0.0)


def _find_segment(target: float, tb: Sequence[float], extrapolate: bool) -> Tuple[str, int]:
    """strategies/mod.rs:24-81.  binary search -> insertion point == lower bound for sorted,
    duplicate-free bounds."""
    idx = bisect_left(tb, target)
    fwd = idx == len(tb)
    back = (not fwd) and idx == 0
    if not fwd and _is_close(tb[idx], target):
        return "OnBoundary", idx
    if (fwd or back) and not extrapolate:
        raise ValueError("Extrapolation is not allowed")
    if back:
        return "ExtrapolateBackward", 0
    if fwd:
        return "ExtrapolateForward", len(tb)
    return "InSegment", idx


def interpolate(strategy: str, time: Sequence[float], y: Sequence[float], t: float,
                extrapolate: bool = True) -> float:
    """One query.  ``time`` is whatever the caller hands the strategy: the unit tests pass
    len(y)+1 bounds, ``interpolate_into`` passes the len(y) time VALUES (timeseries.rs:593)."""
    if strategy == "Linear":
        kind, idx = _find_segment(t, time[: len(time) - 1], extrapolate)  # trims the last entry
        idx = min(idx, len(y) - 1)
        if kind == "OnBoundary":
            return y[idx]
        if kind == "ExtrapolateBackward":
            t1, y1, t2, y2 = time[0], y[0], time[1], y[1]
        elif kind == "ExtrapolateForward":
            assert len(y) >= 2
            t1, y1, t2, y2 = time[len(y) - 2], y[len(y) - 2], time[len(y) - 1], y[len(y) - 1]
        else:
            t1, y1, t2, y2 = time[idx - 1], y[idx - 1], time[idx], y[idx]
        m = (y2 - y1) / (t2 - t1)
        return m * (t - t1) + y1
    if strategy == "Previous":
        kind, idx = _find_segment(t, time, extrapolate)
        if kind == "OnBoundary":
            return y[idx]
        if kind == "ExtrapolateBackward":
            return y[0]
        if kind == "ExtrapolateForward":
            return y[len(y) - 1]
        return y[idx - 1]
    if strategy == "Next":
        kind, idx = _find_segment(t, time, extrapolate)
        idx = min(idx, len(y) - 1)
        if kind == "OnBoundary":
            return y[idx]
        if kind == "ExtrapolateBackward":
            return y[0]
        if kind == "ExtrapolateForward":
            return y[len(y) - 1]
        return y[idx]
    raise ValueError(strategy)


def interpolate_into(strategy: str, src_axis: TimeAxis, y: Sequence[float],
                     new_axis: TimeAxis) -> List[float]:
    """timeseries.rs:586-609: Interp1d over the source axis VALUES, queried at the new VALUES.
    Python-exposed strategies are built with extrapolate=true
    (crates/rscm-core/src/python/timeseries.rs:62-74)."""
    tv = src_axis.values()
    return [interpolate(strategy, tv, list(y), t, True) for t in new_axis.values()]


# --------------------------------------------------------------------------- RK4
def rk4_nsteps(t0: float, t1: float, h: float) -> int:
    return int(math.ceil((t1 - t0) / h))


def rk4_integrate(f: Callable[[List[float]], List[float]], t0: float, t1: float, h: float,
                  y: List[float]) -> List[float]:
    """Classical RK4, ode_solvers 0.6.1 association; end-time check of ivp/mod.rs:90-102."""
    half, sixth = h / 2.0, h / 6.0
    m = rk4_nsteps(t0, t1, h)
    t = t0
    d = len(y)
    assert m >= 1, "get_last_step: assert!(y.len() > 1)"
    for _ in range(m):
        k1 = f(y)
        k2 = f([y[c] + k1[c] * half for c in range(d)])
        k3 = f([y[c] + k2[c] * half for c in range(d)])
        k4 = f([y[c] + k3[c] * h for c in range(d)])
        y = [y[c] + (((k1[c] + k2[c] * 2.0) + k3[c] * 2.0) + k4[c]) * sixth for c in range(d)]
        t = t + h
    assert abs(t - t1) < T_THRESHOLD, "get_last_step: assert!(t_distance < T_THRESHOLD)"
    return y


# --------------------------------------------------------------------------- windows / state
@dataclass
class Window:
    """Scalar TimeseriesWindow, state/windows.rs:155-247 (unit factor fixed at 1.0)."""
    series: List[float]
    index: int
    source: str = EXOGENOUS

    def at_start(self) -> float:
        return self.series[self.index] * 1.0

    def at_end(self) -> Optional[float]:
        j = self.index + 1
        return None if j >= len(self.series) else self.series[j] * 1.0

    def get(self) -> float:
        if self.source == UPSTREAM_OUTPUT:
            e = self.at_end()
            return self.at_start() if e is None else e
        return self.at_start()

    def previous(self) -> Optional[float]:
        return None if self.index == 0 else self.series[self.index - 1] * 1.0

    def at_offset(self, k: int) -> Optional[float]:
        j = self.index + k
        return self.series[j] * 1.0 if 0 <= j < len(self.series) else None


@dataclass
class Req:
    name: str
    kind: str  # INPUT / OUTPUT / STATE


class Component:
    """component.rs:350-437 reduced to scalars.  ``defs`` order: inputs, outputs, states
    (rscm-macros/src/lib.rs:645-651)."""
    type_name = "Component"
    defs: List[Req] = []

    def inputs(self) -> List[Req]:
        return [d for d in self.defs if d.kind in (INPUT, STATE)]

    def outputs(self) -> List[Req]:
        return [d for d in self.defs if d.kind in (OUTPUT, STATE)]

    def solve(self, t0: float, t1: float, w: Dict[str, Window]) -> Dict[str, float]:
        raise NotImplementedError


# --------------------------------------------------------------------------- in-scope components
class TwoLayer(Component):
    type_name = "TwoLayer"
    defs = [Req("Effective Radiative Forcing", INPUT), Req("Surface Temperature", STATE),
            Req("Deep Ocean Temperature", STATE)]

    def __init__(self, lambda0, a, efficacy, eta, heat_capacity_surface, heat_capacity_deep,
                 step=0.1):
        self.p = (lambda0, a, efficacy, eta, heat_capacity_surface, heat_capacity_deep)
        self.h = step

    def solve(self, t0, t1, w):
        lambda0, a, efficacy, eta, cs, cd = self.p

        def rhs(y):
            ts, td = y[0], y[1]
            erf = w["Effective Radiative Forcing"].get()
            diff = ts - td
            lambda_eff = lambda0 - a * ts
            hx_s = efficacy * eta * diff
            dts = (erf - lambda_eff * ts - hx_s) / cs
            hx_d = eta * diff
            dtd = hx_d / cd
            return [dts, dtd, cs * dts + cd * dtd]

        y0 = [w["Surface Temperature"].at_start(), w["Deep Ocean Temperature"].at_start(), 0.0]
        y = rk4_integrate(rhs, t0, t1, self.h, y0)
        self.last_heat = y[2]
        return {"Surface Temperature": y[0], "Deep Ocean Temperature": y[1]}


class CarbonCycle(Component):
    type_name = "CarbonCycle"
    defs = [Req("Emissions|CO2|Anthropogenic", INPUT), Req("Surface Temperature", INPUT),
            Req("Atmospheric Concentration|CO2", STATE), Req("Cumulative Emissions|CO2", STATE),
            Req("Cumulative Land Uptake", STATE)]

    def __init__(self, tau, conc_pi, alpha_temperature, step=0.1):
        self.p = (tau, conc_pi, alpha_temperature)
        self.h = step

    def solve(self, t0, t1, w):
        tau, conc_pi, alpha = self.p

        def rhs(y):
            emissions = w["Emissions|CO2|Anthropogenic"].get()
            temperature = w["Surface Temperature"].get()
            conc = y[0]
            lifetime = tau * math.exp(alpha * temperature)
            uptake = (conc - conc_pi) / lifetime
            return [emissions / GTC_PER_PPM - uptake, uptake * GTC_PER_PPM, emissions]

        y0 = [w["Atmospheric Concentration|CO2"].at_start(),
              w["Cumulative Land Uptake"].at_start(),
              w["Cumulative Emissions|CO2"].at_start()]
        y = rk4_integrate(rhs, t0, t1, self.h, y0)
        return {"Atmospheric Concentration|CO2": y[0], "Cumulative Land Uptake": y[1],
                "Cumulative Emissions|CO2": y[2]}


class CO2ERF(Component):
    type_name = "CO2ERF"
    defs = [Req("Atmospheric Concentration|CO2", INPUT),
            Req("Effective Radiative Forcing|CO2", OUTPUT)]

    def __init__(self, erf_2xco2, conc_pi):
        self.erf_2xco2, self.conc_pi = erf_2xco2, conc_pi

    def calculate_erf(self, c):
        return self.erf_2xco2 / math.log(2.0) * math.log(1.0 + (c - self.conc_pi) / self.conc_pi)

    def solve(self, t0, t1, w):
        return {"Effective Radiative Forcing|CO2":
                self.calculate_erf(w["Atmospheric Concentration|CO2"].get())}


# --------------------------------------------------------------------------- rscm-magicc adapters
class _StepOracle(Component):
    """A rscm-magicc component inside the generic stepper: the window reads are restated here
    (file:line per class), the arithmetic of one step is the C oracle's single-step function
    (``oracle/cbind.py``), so a graph of these exercises exactly the index conventions."""

    def __init__(self, params):
        self.params = list(params)


class CH4Chemistry(_StepOracle):
    """crates/rscm-magicc/src/chemistry/ch4.rs solve(): emissions / temperature / precursors through
    get(), own concentration at_start() and previous().unwrap_or(current)."""
    type_name = "CH4Chemistry"
    defs = [Req("Emissions|CH4", INPUT), Req("Surface Temperature", INPUT), Req("Emissions|NOx", INPUT),
            Req("Emissions|CO", INPUT), Req("Emissions|NMVOC", INPUT),
            Req("Lifetime|CH4", OUTPUT), Req("Atmospheric Concentration|CH4", STATE)]

    def solve(self, t0, t1, w):
        from oracle import cbind
        c = w["Atmospheric Concentration|CH4"]
        cur = c.at_start()
        prev = c.previous()
        conc, life = cbind.ch4_solve_concentration(
            self.params, cur if prev is None else prev, cur, w["Emissions|CH4"].get(), w["Surface Temperature"].get(),
            w["Emissions|NOx"].get(), w["Emissions|CO"].get(), w["Emissions|NMVOC"].get())
        return {"Atmospheric Concentration|CH4": conc, "Lifetime|CH4": life}


class N2OChemistry(_StepOracle):
    """crates/rscm-magicc/src/chemistry/n2o.rs:203-218: the stratospheric delay reads
    at_offset(-delay) (else previous) and at_offset(-(delay+1)) (else the former)."""
    type_name = "N2OChemistry"
    defs = [Req("Emissions|N2O", INPUT), Req("Lifetime|N2O", OUTPUT), Req("Atmospheric Concentration|N2O", STATE)]

    def solve(self, t0, t1, w):
        from oracle import cbind
        c = w["Atmospheric Concentration|N2O"]
        cur = c.at_start()
        prev = c.previous()
        prev = cur if prev is None else prev
        delay = max(int(self.params[4]), 1)
        t_delay = c.at_offset(-delay)
        t_delay = prev if t_delay is None else t_delay
        t_delay_m1 = c.at_offset(-(delay + 1))
        t_delay_m1 = t_delay if t_delay_m1 is None else t_delay_m1
        conc, life = cbind.n2o_solve_concentration(self.params, prev, cur, (t_delay + t_delay_m1) / 2.0,
                                                   w["Emissions|N2O"].get(), t1 - t0)
        return {"Atmospheric Concentration|N2O": conc, "Lifetime|N2O": life}


class GhgForcing(_StepOracle):
    """crates/rscm-magicc/src/forcing/ghg.rs:272-345: three concentrations through get()."""
    type_name = "GhgForcing"
    defs = [Req("Atmospheric Concentration|CO2", INPUT), Req("Atmospheric Concentration|CH4", INPUT),
            Req("Atmospheric Concentration|N2O", INPUT), Req("Effective Radiative Forcing|CO2", OUTPUT),
            Req("Effective Radiative Forcing|CH4", OUTPUT), Req("Effective Radiative Forcing|N2O", OUTPUT)]

    def solve(self, t0, t1, w):
        from oracle import cbind
        f = cbind.ghg_forcings(self.params, w["Atmospheric Concentration|CO2"].get(),
                               w["Atmospheric Concentration|CH4"].get(), w["Atmospheric Concentration|N2O"].get())
        return {"Effective Radiative Forcing|CO2": f["co2_erf"], "Effective Radiative Forcing|CH4": f["ch4_erf"],
                "Effective Radiative Forcing|N2O": f["n2o_erf"]}


class AerosolIndirect(_StepOracle):
    """crates/rscm-magicc/src/forcing/aerosol_indirect.rs:75-170: SOx and OC emissions through get()."""
    type_name = "AerosolIndirect"
    defs = [Req("Emissions|SOx", INPUT), Req("Emissions|OC", INPUT),
            Req("Effective Radiative Forcing|Aerosol|Indirect", OUTPUT)]

    def solve(self, t0, t1, w):
        from oracle import cbind
        out = cbind.pointwise_eval(cbind.PW_AEROSOL_INDIRECT, self.params,
                                   [w["Emissions|SOx"].get(), w["Emissions|OC"].get()])
        return {"Effective Radiative Forcing|Aerosol|Indirect": float(out[0])}


# --------------------------------------------------------------------------- aggregates
def compute_aggregate(values: Sequence[float], op: str,
                      weights: Optional[Sequence[float]] = None) -> float:
    """schema.rs:760-802 (NaN contributors are skipped; all-NaN -> NaN)."""
    if op == "Sum":
        s, n = 0.0, 0
        for v in values:
            if not math.isnan(v):
                s += v
                n += 1
        return s if n else NAN
    if op == "Mean":
        s, n = 0.0, 0
        for v in values:
            if not math.isnan(v):
                s += v
                n += 1
        return s / float(n) if n else NAN
    if op == "Weighted":  # plain sum of v*w over the non-NaN pairs (no renormalisation)
        ws, n = 0.0, 0
        for v, wt in zip(values, weights):
            if not math.isnan(v):
                ws += v * wt
                n += 1
        return ws if n else NAN
    raise ValueError(op)


class Aggregator(Component):
    """schema.rs:822-952, scalar case: contributors are read with at_end() (fallback at_start)."""

    def __init__(self, name: str, op: str, contributors: List[str], weights=None):
        self.type_name = f"Aggregator:{name}"
        self.name, self.op, self.contributors, self.weights = name, op, contributors, weights
        self.defs = [Req(c, INPUT) for c in contributors] + [Req(name, OUTPUT)]

    def solve(self, t0, t1, w):
        vals = []
        for c in self.contributors:
            e = w[c].at_end()
            vals.append(w[c].at_start() if e is None else e)
        return {self.name: compute_aggregate(vals, self.op, self.weights)}


# --------------------------------------------------------------------------- builder + model
@dataclass
class ExoSeries:
    values: List[float]
    axis: TimeAxis
    strategy: str = "Linear"


@dataclass
class Model:
    axis: TimeAxis
    order_nodes: List[Optional[Component]]
    edges_out: Dict[int, List[int]]
    data: Dict[str, List[float]]
    var_type: Dict[str, str]
    sources: Dict[Tuple[str, str], str]
    time_index: int = 0

    def _bfs(self) -> List[int]:
        # petgraph Bfs over Graph: neighbours come out most-recently-added edge first
        seen, out, queue = {0}, [], [0]
        while queue:
            n = queue.pop(0)
            out.append(n)
            for m in reversed(self.edges_out.get(n, [])):
                if m not in seen:
                    seen.add(m)
                    queue.append(m)
        return out

    def step(self) -> None:
        assert self.time_index < len(self.axis) - 1
        n = self.time_index
        t0, t1 = self.axis.at_bounds(n)
        for node in self._bfs():
            comp = self.order_nodes[node]
            if comp is None:  # NullComponent
                continue
            wins = {}
            for r in comp.inputs():
                src = self.sources.get((r.name, comp.type_name), EXOGENOUS)
                wins[r.name] = Window(self.data[r.name], n, src)
            for key, v in comp.solve(t0, t1, wins).items():
                self.data[key][n + 1] = v  # runtime.rs:480
        self.time_index += 1

    def run(self) -> None:
        while self.time_index < len(self.axis) - 1:
            self.step()

    def finished(self) -> bool:
        return self.time_index == len(self.axis) - 1


@dataclass
class ModelBuilder:
    axis: Optional[TimeAxis] = None
    components: List[Component] = field(default_factory=list)
    initial_values: Dict[str, float] = field(default_factory=dict)
    exogenous: Dict[str, ExoSeries] = field(default_factory=dict)
    aggregates: List[Tuple[str, str, List[str]]] = field(default_factory=list)
    schema_variables: List[str] = field(default_factory=list)

    def build(self) -> Model:
        """builder.rs:418-860 reduced to scalars and unit factor 1."""
        nodes: List[Optional[Component]] = [None]
        edges: Dict[int, List[int]] = {}
        endogenous: Dict[str, int] = {}
        exo_names: List[str] = []
        defs: Dict[str, str] = {}
        sources: Dict[Tuple[str, str], str] = {}
        agg_names = {a[0] for a in self.aggregates}
        pending: List[Tuple[int, str]] = []

        def add_edge(a, b):
            edges.setdefault(a, []).append(b)

        for comp in self.components:
            node = len(nodes)
            nodes.append(comp)
            has_dep = False
            for r in comp.inputs():  # classification, builder.rs:464-485
                if r.kind == STATE:
                    src = OWN_STATE
                elif r.name in endogenous or r.name in agg_names:
                    src = UPSTREAM_OUTPUT
                else:
                    src = EXOGENOUS
                sources[(r.name, comp.type_name)] = src
            for r in comp.inputs():  # edges, builder.rs:487-518
                defs.setdefault(r.name, r.kind)
                if r.name in endogenous:
                    add_edge(endogenous[r.name], node)
                    has_dep = True
                elif r.name in agg_names:
                    pending.append((node, r.name))
                    has_dep = True
                elif r.name not in exo_names:
                    exo_names.append(r.name)
            if not has_dep:
                add_edge(0, node)
            for r in comp.outputs():  # builder.rs:532-559
                if r.kind == STATE or r.name not in defs:
                    defs[r.name] = r.kind
                if r.name in endogenous:
                    add_edge(endogenous[r.name], node)
                endogenous[r.name] = node
        for name in self.schema_variables:
            if name not in defs:
                defs[name] = INPUT
                exo_names.append(name)
        for name, op, contributors in self.aggregates:  # builder.rs:632-701
            agg = Aggregator(name, op, contributors)
            node = len(nodes)
            nodes.append(agg)
            has_dep = False
            for c in contributors:
                if c in endogenous:
                    add_edge(endogenous[c], node)
                    has_dep = True
            if not has_dep:
                add_edge(0, node)
            endogenous[name] = node
            defs[name] = OUTPUT
        for node, name in pending:
            if name in endogenous:
                add_edge(endogenous[name], node)
        for name, kind in defs.items():  # builder.rs:704-717
            if kind == STATE and name not in self.initial_values:
                raise ValueError(f"MissingInitialValue: {name}")
        T = len(self.axis)
        data, var_type = {}, {}
        for name in defs:  # builder.rs:735-790
            var_type[name] = "Endogenous" if name in endogenous else "Exogenous"
            exo = self.exogenous.get(name) if name in exo_names else None
            if exo is not None:
                data[name] = interpolate_into(exo.strategy, exo.axis, exo.values, self.axis)
            else:
                ts = [NAN] * T
                if name in self.initial_values:
                    ts[0] = self.initial_values[name]
                data[name] = ts
        return Model(self.axis, nodes, edges, data, var_type, sources)


# --------------------------------------------------------------------------- ensemble helpers
def extract_outputs(model: Model, names: Sequence[str]) -> Dict[str, Dict[str, float]]:
    """model_runner.rs:161-212: non-NaN (time,value) pairs keyed by format!("{:.6}", t)."""
    out = {}
    for name in names:
        ts = model.data[name]
        out[name] = {f"{model.axis.at(i):.6f}": v for i, v in enumerate(ts) if not math.isnan(v)}
    return out


def lhs_unit(n: int, n_params: int, uniform01: Callable[[], float],
             shuffle: Callable[[List[float]], None]) -> List[List[float]]:
    """parameter_set.rs:207-233 on the unit cube: per dimension
    interval_size = 1/n; u_i = i*interval_size + U*interval_size; then a shuffle of that column
    (the inverse CDF of a bounded constant-pdf prior is low + u*(high-low), :322-334)."""
    cols = []
    for _ in range(n_params):
        size = 1.0 / float(n)
        col = [float(i) * size + uniform01() * size for i in range(n)]
        shuffle(col)
        cols.append(col)
    return [[cols[j][i] for j in range(n_params)] for i in range(n)]
