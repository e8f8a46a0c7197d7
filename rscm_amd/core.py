"""Host-side mirror of the reference's Python model-description surface for the hot path.

Same names, argument meaning and error behaviour as ``rscm._lib.core``
(python/rscm/_lib/core/__init__.pyi:19-175,322-463,563-628; implementations under
crates/rscm-core/src/python/), so a script that builds a two-layer or coupled model keeps working:

    model = (ModelBuilder().with_time_axis(axis).with_rust_component(two_layer)
             .with_exogenous_variable("Effective Radiative Forcing", erf)
             .with_initial_values({...}).build())
    model.run(); model.timeseries().get_timeseries_by_name("Surface Temperature").values()

``build()`` resolves the component graph exactly as the reference does
(crates/rscm-core/src/model/builder.rs:418-860: registration-order variable sources, schema
aggregates, missing-initial-value check, exogenous resampling onto the model axis) and maps the
result onto one of the fused HIP kernels.  Graphs outside the supported set raise
``NotImplementedError`` -- there is no generic CPU interpreter and no CPU fallback.

Everything numeric that happens per time step runs on the GPU; the only host arithmetic here is
the build-time resampling of exogenous series (cold path, crates/rscm-core/src/timeseries.rs:586-609).
"""
from __future__ import annotations

import ctypes as C
import enum
import math
from bisect import bisect_left
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import _lib as L
from .ensemble import Ensemble, run_lockstep

NAN = float("nan")


# ------------------------------------------------------------------------------------ time axis
class TimeAxis:
    """crates/rscm-core/src/timeseries.rs:45-212.  ``bounds`` has ``len()+1`` entries; value ``i``
    is the start of step ``i``."""

    def __init__(self, bounds: np.ndarray):
        b = np.ascontiguousarray(bounds, dtype=np.float64)
        if b.ndim != 1 or not np.all(b[1:] > b[:-1]):
            raise ValueError("time axis must be strictly increasing")  # assert!(is_monotonic)
        self._bounds = b

    @staticmethod
    def from_values(values) -> "TimeAxis":
        v = np.asarray(values, dtype=np.float64)
        if len(v) < 2:
            raise ValueError("from_values needs at least 2 values")
        step = v[-1] - v[-2]
        return TimeAxis(np.concatenate([v, [v[-1] + step]]))

    @staticmethod
    def from_bounds(bounds) -> "TimeAxis":
        b = np.asarray(bounds, dtype=np.float64)
        if len(b) < 2:
            raise ValueError("from_bounds needs at least 2 bounds")
        return TimeAxis(b)

    def values(self) -> np.ndarray:
        return self._bounds[:-1].copy()

    def bounds(self) -> np.ndarray:
        return self._bounds.copy()

    def __len__(self) -> int:
        return len(self._bounds) - 1

    def at(self, index: int) -> Optional[float]:
        return float(self._bounds[index]) if 0 <= index < len(self) else None

    def at_bounds(self, index: int) -> Optional[Tuple[float, float]]:
        if 0 <= index < len(self):
            return float(self._bounds[index]), float(self._bounds[index + 1])
        return None

    def contains(self, value: float) -> bool:
        return bool(np.any(self._bounds[:-1] == value))

    def index_of(self, value: float) -> Optional[int]:
        hit = np.nonzero(np.abs(self._bounds[:-1] - value) < 1e-10)[0]  # timeseries.rs:204-211
        return int(hit[0]) if len(hit) else None


class InterpolationStrategy(enum.Enum):
    Linear = enum.auto()
    Next = enum.auto()
    Previous = enum.auto()


class VariableType(enum.Enum):
    Exogenous = enum.auto()
    Endogenous = enum.auto()


# ------------------------------------------------------------------------------------ interpolation
def _find_segment(target: float, tb: np.ndarray, extrapolate: bool) -> Tuple[str, int]:
    """crates/rscm-core/src/interpolate/strategies/mod.rs:24-81."""
    idx = bisect_left(tb, target)
    fwd = idx == len(tb)
    back = (not fwd) and idx == 0
    if not fwd and math.isclose(tb[idx], target, rel_tol=1e-9, abs_tol=0.0):  # is_close! defaults
        return "OnBoundary", idx
    if (fwd or back) and not extrapolate:
        where = "start of" if back else "end of"
        edge = tb[0] if back else tb[-1]
        raise RuntimeError(f"Extrapolation is not allowed. Target={target} is before the "
                           f"{where} the interpolation range={edge}")
    if back:
        return "ExtrapolateBackward", 0
    if fwd:
        return "ExtrapolateForward", len(tb)
    return "InSegment", idx


def interpolate(strategy: InterpolationStrategy, time: np.ndarray, y: np.ndarray, t: float,
                extrapolate: bool = True) -> float:
    """One query of Interp1d (strategies/linear_spline.rs:33-96, previous.rs:58-80,
    next.rs:56-80).  ``time`` is what the caller hands over: ``interpolate_into`` and ``at_time``
    pass the len(y) time VALUES, so the Linear strategy's "trim the last bound" drops the last
    value and a query there goes through the forward-extrapolation formula."""
    n = len(y)
    if strategy is InterpolationStrategy.Linear:
        kind, idx = _find_segment(t, time[: len(time) - 1], extrapolate)
        idx = min(idx, n - 1)
        if kind == "OnBoundary":
            return float(y[idx])
        if kind == "ExtrapolateBackward":
            t1, y1, t2, y2 = time[0], y[0], time[1], y[1]
        elif kind == "ExtrapolateForward":
            if n < 2:
                raise RuntimeError("linear extrapolation needs two points")
            t1, y1, t2, y2 = time[n - 2], y[n - 2], time[n - 1], y[n - 1]
        else:
            t1, y1, t2, y2 = time[idx - 1], y[idx - 1], time[idx], y[idx]
        m = (y2 - y1) / (t2 - t1)
        return float(m * (t - t1) + y1)
    kind, idx = _find_segment(t, time, extrapolate)
    if strategy is InterpolationStrategy.Previous:
        if kind == "OnBoundary":
            return float(y[idx])
        if kind == "ExtrapolateBackward":
            return float(y[0])
        if kind == "ExtrapolateForward":
            return float(y[n - 1])
        return float(y[idx - 1])
    idx = min(idx, n - 1)  # Next
    if kind in ("OnBoundary", "InSegment"):
        return float(y[idx])
    return float(y[0] if kind == "ExtrapolateBackward" else y[n - 1])


# ------------------------------------------------------------------------------------ timeseries
class Timeseries:
    """Scalar timeseries (python/rscm/_lib/core/__init__.pyi:37-93).  Python-built strategies
    extrapolate (crates/rscm-core/src/python/timeseries.rs:62-74)."""

    def __init__(self, values, time_axis: TimeAxis, units: str,
                 interpolation_strategy: InterpolationStrategy):
        v = np.array(values, dtype=np.float64).reshape(-1)
        if len(v) != len(time_axis):
            raise ValueError(f"values ({len(v)}) and time axis ({len(time_axis)}) differ in length")
        self._values = v
        self._axis = time_axis
        self._units = units
        self._strategy = interpolation_strategy
        self._latest = len(v) - 1 if len(v) and not np.isnan(v[-1]) else self._last_valid(v)

    @staticmethod
    def _last_valid(v: np.ndarray) -> int:
        ok = np.nonzero(~np.isnan(v))[0]
        return int(ok[-1]) if len(ok) else 0

    @staticmethod
    def from_values(values, time) -> "Timeseries":
        axis = time if isinstance(time, TimeAxis) else TimeAxis.from_values(time)
        return Timeseries(values, axis, "", InterpolationStrategy.Linear)

    def with_interpolation_strategy(self, interpolation_strategy) -> "Timeseries":
        self._strategy = interpolation_strategy
        return self

    def __len__(self) -> int:
        return len(self._values)

    def set(self, time_index: int, value: float) -> None:
        self._values[time_index] = value
        self._latest = max(self._latest, time_index)

    def values(self) -> np.ndarray:
        return self._values.copy()

    @property
    def latest(self) -> int:
        return self._latest

    @property
    def units(self) -> str:
        return self._units

    @property
    def time_axis(self) -> TimeAxis:
        return self._axis

    @property
    def interpolation_strategy(self) -> InterpolationStrategy:
        return self._strategy

    def latest_value(self) -> Optional[float]:
        return float(self._values[self._latest]) if len(self._values) else None

    def at(self, time_index: int) -> Optional[float]:
        return float(self._values[time_index]) if 0 <= time_index < len(self) else None

    def at_time(self, time: float) -> float:
        return interpolate(self._strategy, self._axis.values(), self._values, float(time), True)

    def interpolate_into(self, new_axis: TimeAxis) -> "Timeseries":
        """crates/rscm-core/src/timeseries.rs:586-609."""
        tv = self._axis.values()
        out = np.array([interpolate(self._strategy, tv, self._values, float(t), True)
                        for t in new_axis.values()])
        return Timeseries(out, new_axis, self._units, self._strategy)


class FourBoxTimeseries:
    """FourBox grid timeseries: values (n_times, 4) in the order NorthernOcean, NorthernLand,
    SouthernOcean, SouthernLand (python/rscm/_lib/core/__init__.pyi:95-106)."""

    def __init__(self, values, time_axis: TimeAxis, units: str = ""):
        v = np.array(values, dtype=np.float64)
        if v.shape != (len(time_axis), 4):
            raise ValueError(f"FourBox values must be ({len(time_axis)}, 4), got {v.shape}")
        self._values, self._axis, self._units = v, time_axis, units

    def __len__(self) -> int:
        return len(self._values)

    def values(self) -> np.ndarray:
        return self._values.copy()

    @property
    def units(self) -> str:
        return self._units

    @property
    def time_axis(self) -> TimeAxis:
        return self._axis

    @property
    def latest(self) -> int:
        ok = np.nonzero(~np.isnan(self._values[:, 0]))[0]
        return int(ok[-1]) if len(ok) else 0


class GridType(enum.Enum):
    Scalar = enum.auto()
    FourBox = enum.auto()
    Hemispheric = enum.auto()


class TimeseriesCollection:
    """python/rscm/_lib/core/__init__.pyi:119-190; kept sorted by name like the reference's
    ``Vec<TimeseriesItem>`` (crates/rscm-core/src/timeseries_collection.rs:318-321)."""

    def __init__(self) -> None:
        self._items: Dict[str, Tuple[Timeseries, VariableType]] = {}
        self._fourbox: Dict[str, FourBoxTimeseries] = {}

    def add_fourbox_timeseries(self, name: str, timeseries: FourBoxTimeseries) -> None:
        self._fourbox[name] = timeseries

    def get_fourbox_timeseries_by_name(self, name: str) -> Optional[FourBoxTimeseries]:
        return self._fourbox.get(name)

    def add_timeseries(self, name: str, timeseries: Timeseries,
                       variable_type: VariableType = VariableType.Exogenous) -> None:
        if name in self._items:
            raise ValueError(f"timeseries {name!r} already exists")
        self._items[name] = (timeseries, variable_type)

    def get_timeseries_by_name(self, name: str) -> Optional[Timeseries]:
        item = self._items.get(name)
        if item is None:
            return None
        ts = item[0]
        return Timeseries(ts.values(), ts.time_axis, ts.units, ts.interpolation_strategy)

    def variable_type(self, name: str) -> Optional[VariableType]:
        item = self._items.get(name)
        return item[1] if item else None

    def names(self) -> List[str]:
        return sorted(list(self._items) + list(self._fourbox))

    def timeseries(self) -> List[Timeseries]:
        return [self.get_timeseries_by_name(n) for n in sorted(self._items)]

    def __contains__(self, name: str) -> bool:
        return name in self._items


# ------------------------------------------------------------------------------------ schema
class SchemaVariableDefinition:
    """python/rscm/_lib/core/__init__.pyi: one declared variable (name, unit, grid type)."""

    def __init__(self, name: str, unit: str, grid_type: "GridType"):
        self.name, self.unit, self.grid_type = name, unit, grid_type

    def __repr__(self) -> str:
        return f"SchemaVariableDefinition({self.name!r}, {self.unit!r}, {self.grid_type.name})"


class AggregateDefinition:
    """One aggregate of the schema.  Unpacks as ``(unit, operation_type, contributors, weights)``."""

    def __init__(self, name: str, unit: str, operation_type: str, contributors: List[str], weights: Optional[List[float]],
                 grid_type: "GridType"):
        self.name, self.unit, self.operation_type = name, unit, operation_type
        self.contributors, self.weights, self.grid_type = contributors, weights, grid_type

    def __iter__(self):
        return iter((self.unit, self.operation_type, self.contributors, self.weights))

    def __getitem__(self, k):
        return (self.unit, self.operation_type, self.contributors, self.weights)[k]


class VariableSchema:
    """python/rscm/_lib/core/__init__.pyi:322-404; validation as crates/rscm-core/src/schema.rs
    (undefined contributors, unit / grid type / weight-count mismatches, circular aggregates)."""

    def __init__(self) -> None:
        self.variables: Dict[str, SchemaVariableDefinition] = {}
        self.aggregates: Dict[str, AggregateDefinition] = {}

    @property
    def grid_types(self) -> Dict[str, "GridType"]:
        return {n: v.grid_type for n, v in self.variables.items()}

    def add_variable(self, name: str, unit: str, grid_type=None) -> "VariableSchema":
        self.variables[name] = SchemaVariableDefinition(name, unit, grid_type or GridType.Scalar)
        return self

    def add_aggregate(self, name: str, unit: str, operation: str, contributors: Sequence[str],
                      weights: Optional[Sequence[float]] = None, grid_type=None) -> "VariableSchema":
        if operation not in ("Sum", "Mean", "Weighted"):
            raise ValueError(f"Unknown operation {operation!r}: expected 'Sum', 'Mean' or 'Weighted'")
        if operation == "Weighted" and weights is None:
            raise ValueError("weights must be provided for the 'Weighted' operation")
        self.aggregates[name] = AggregateDefinition(name, unit, operation, list(contributors),
                                                    list(weights) if weights is not None else None, grid_type or GridType.Scalar)
        return self

    def contains(self, name: str) -> bool:
        return name in self.variables or name in self.aggregates

    def get_grid_type(self, name: str) -> Optional["GridType"]:
        if name in self.variables:
            return self.variables[name].grid_type
        return self.aggregates[name].grid_type if name in self.aggregates else None

    def validate(self) -> None:
        def norm(u: str) -> str:
            return "".join(str(u).split())
        for name, agg in self.aggregates.items():
            for c in agg.contributors:
                if not self.contains(c):
                    raise ValueError(f"Undefined contributor: aggregate {name!r}: contributor {c!r} is not in the schema")
                unit = self.variables[c].unit if c in self.variables else self.aggregates[c].unit
                if unit and agg.unit and norm(unit) != norm(agg.unit):
                    raise ValueError(f"Unit mismatch: contributor {c!r} has unit {unit!r}, aggregate {name!r} has {agg.unit!r}")
                if self.get_grid_type(c) != agg.grid_type:
                    raise ValueError(f"Grid type mismatch: contributor {c!r} is {self.get_grid_type(c).name}, "
                                     f"aggregate {name!r} is {agg.grid_type.name}")
            if agg.operation_type == "Weighted" and len(agg.weights or []) != len(agg.contributors):
                raise ValueError(f"Weight count mismatch: aggregate {name!r} has {len(agg.contributors)} contributors "
                                 f"and {len(agg.weights or [])} weights")
        state: Dict[str, int] = {}

        def visit(n: str) -> None:
            if state.get(n) == 1:
                raise ValueError(f"Circular dependency among the aggregates at {n!r}")
            if state.get(n) == 2 or n not in self.aggregates:
                return
            state[n] = 1
            for c in self.aggregates[n].contributors:
                visit(c)
            state[n] = 2

        for n in self.aggregates:
            visit(n)


# ------------------------------------------------------------------------------------ components
class Component:
    """A built Rust component as the reference's ``ModelBuilder.with_rust_component`` sees it:
    a type name, parameters, and its requirement definitions in macro order (inputs, outputs,
    states; crates/rscm-macros/src/lib.rs:645-651)."""

    type_name = "Component"
    definitions: List[Tuple[str, str, str]] = []  # (name, unit, "Input"|"Output"|"State")

    def __init__(self, parameters: Dict[str, float]):
        self.parameters = dict(parameters)

    def input_names(self) -> List[str]:
        return [n for n, _, k in self.definitions if k in ("Input", "State")]

    def output_names(self) -> List[str]:
        return [n for n, _, k in self.definitions if k in ("Output", "State")]


class ComponentBuilder:
    component_cls = Component
    required: Tuple[str, ...] = ()

    def __init__(self, parameters: Dict[str, float]):
        self._parameters = dict(parameters)

    @classmethod
    def from_parameters(cls, parameters: Dict[str, float]):
        missing = [k for k in cls.required if k not in parameters]
        if missing:  # pythonize/serde: "missing field `x`" (crates/rscm-core/src/python/component.rs:19-48)
            raise ValueError(f"missing field `{missing[0]}`")
        return cls({k: float(parameters[k]) for k in cls.required})

    def build(self):
        return self.component_cls(self._parameters)


# ------------------------------------------------------------------------------------ builder
SUPPORTED = ("any graph of the built-in components with scalar links (one linked ensemble per component, "
             "stepped in the reference's graph order); fused single-launch kernels for: "
             "[ClimateUDEB] with exogenous 'Effective Radiative Forcing'",
             "[GhgForcing] | [OzoneForcing] | [AerosolDirect] | [AerosolIndirect] | [CH4Chemistry] | "
             "[N2OChemistry] | [CO2Budget] | [TerrestrialCarbon] | [OceanCarbon] | [HalocarbonChemistry] | "
             "[FourBoxOceanHeatUptake] | [OceanSurfacePartialPressure] with their inputs as "
             "exogenous series",
             "[TwoLayer] with exogenous or upstream 'Effective Radiative Forcing'",
             "[CarbonCycle, CO2ERF, TwoLayer] + Sum aggregate 'Effective Radiative Forcing' "
             "over ['Effective Radiative Forcing|CO2'] (registration order as listed)")

# single-component graphs whose inputs are all exogenous rows of the input block
STATELESS_KINDS = {"GhgForcing": L.KIND_GHG_FORCING, "OzoneForcing": L.KIND_OZONE_FORCING,
                   "AerosolDirect": L.KIND_AEROSOL_DIRECT, "AerosolIndirect": L.KIND_AEROSOL_INDIRECT,
                   "CH4Chemistry": L.KIND_CH4_CHEMISTRY, "N2OChemistry": L.KIND_N2O_CHEMISTRY,
                   "CO2Budget": L.KIND_CO2_BUDGET, "TerrestrialCarbon": L.KIND_TERRESTRIAL_CARBON,
                   "OceanCarbon": L.KIND_OCEAN_CARBON, "HalocarbonChemistry": L.KIND_HALOCARBON,
                   "FourBoxOceanHeatUptake": L.KIND_FOURBOX_OHU, "OceanSurfacePartialPressure": L.KIND_OSPP}

TL_PARAM_ORDER = ("lambda0", "a", "efficacy", "eta", "heat_capacity_surface", "heat_capacity_deep")
CP_PARAM_ORDER = TL_PARAM_ORDER + ("tau", "conc_pi", "alpha_temperature", "erf_2xco2")


class ModelBuilder:
    """python/rscm/_lib/core/__init__.pyi:405-463 -> crates/rscm-core/src/model/builder.rs."""

    def __init__(self) -> None:
        self._axis: Optional[TimeAxis] = None
        self._components: List[Component] = []
        self._initial: Dict[str, float] = {}
        self._exogenous = TimeseriesCollection()
        self._schema: Optional[VariableSchema] = None
        self._device = 0
        self._grid_weights: Dict["GridType", List[float]] = {}

    def with_time_axis(self, time_axis: TimeAxis) -> "ModelBuilder":
        self._axis = time_axis
        return self

    def with_rust_component(self, component: Component) -> "ModelBuilder":
        self._components.append(component)
        return self

    with_component = with_rust_component

    def with_py_component(self, component) -> "ModelBuilder":
        """A component written in Python (``rscm_amd.component.PythonComponent.build(obj)``).  It cannot
        be part of a kernel: its ``solve`` runs on the host between the launches of the graph's linked
        ensembles, once per member and step (``GraphModel``)."""
        from .component import PythonComponent
        if not isinstance(component, PythonComponent):
            raise NotImplementedError("with_py_component takes PythonComponent.build(<rscm_amd.component.Component instance>)")
        self._components.append(component)
        return self

    def with_initial_values(self, initial_values: Dict[str, float]) -> "ModelBuilder":
        self._initial.update({k: float(v) for k, v in initial_values.items()})
        return self

    def with_exogenous_variable(self, name: str, timeseries: Timeseries) -> "ModelBuilder":
        self._exogenous.add_timeseries(name, timeseries, VariableType.Exogenous)
        return self

    def with_exogenous_collection(self, collection: TimeseriesCollection) -> "ModelBuilder":
        for name in collection.names():
            self._exogenous.add_timeseries(name, collection.get_timeseries_by_name(name),
                                           VariableType.Exogenous)
        return self

    def with_schema(self, schema: VariableSchema) -> "ModelBuilder":
        self._schema = schema
        return self

    def with_grid_weights(self, grid_type: "GridType", weights: Sequence[float]) -> "ModelBuilder":
        """builder.rs:56-103: area weights used wherever a FourBox variable is aggregated to a scalar
        (default 0.25 each, FourBoxGrid::magicc_standard)."""
        w = [float(x) for x in weights]
        if grid_type != GridType.FourBox:
            raise NotImplementedError("only FourBox weights are used on the device path")
        if len(w) != 4:
            raise ValueError(f"Weights length {len(w)} does not match FourBox grid size 4")
        if abs(sum(w) - 1.0) >= 1e-6:
            raise ValueError(f"Weights must sum to 1.0, got {sum(w)}")
        self._grid_weights[grid_type] = w
        return self

    def with_device(self, device: int) -> "ModelBuilder":
        """Extension: which GPU the model lives on."""
        self._device = device
        return self

    # -- graph resolution (builder.rs:448-560) ----------------------------------------------
    def _resolve(self):
        if self._axis is None:
            raise ValueError("no time axis")
        endogenous: Dict[str, str] = {}
        sources: Dict[Tuple[str, str], str] = {}
        exo_names: List[str] = []
        kinds: Dict[str, str] = {}
        aggregates = self._schema.aggregates if self._schema else {}
        if self._schema:
            self._schema.validate()
        for comp, node_name in zip(self._components, self._node_names()):
            for name, _, kind in comp.definitions:
                if kind not in ("Input", "State"):
                    continue
                if kind == "State":
                    src = "OwnState"
                elif name in endogenous or name in aggregates:
                    src = "UpstreamOutput"
                else:
                    src = "Exogenous"
                sources[(name, comp.type_name)] = src
                sources[(name, node_name)] = src  # "<type>#k" for all but the last component of a type
                kinds.setdefault(name, kind)
                if name not in endogenous and name not in aggregates and name not in exo_names:
                    exo_names.append(name)
            for name, _, kind in comp.definitions:
                if kind in ("Output", "State"):
                    if kind == "State" or name not in kinds:
                        kinds[name] = kind
                    endogenous[name] = comp.type_name
        for name in aggregates:
            endogenous[name] = f"Aggregator:{name}"
            kinds[name] = "Output"
        for name, kind in kinds.items():  # builder.rs:704-717
            if kind == "State" and name not in self._initial:
                raise ValueError(f"Missing initial value for state variable '{name}' "
                                 f"(component {endogenous.get(name, 'unknown')})")
        return endogenous, sources, exo_names, aggregates

    def _exogenous_on_axis(self, name: str, exo_names: List[str]) -> Optional[np.ndarray]:
        if name in exo_names and name in self._exogenous:
            ts = self._exogenous.get_timeseries_by_name(name)
            self._check_units(name, ts.units)
            return ts.interpolate_into(self._axis).values()
        return None

    def _check_units(self, name: str, supplied: str) -> None:
        """The reference converts between compatible units when a series, the schema and a component
        disagree (builder.rs:141-338).  There is no units registry on this path -- every factor is
        1.0 -- so a disagreement that is more than spelling is refused instead of computed wrongly."""
        def norm(u: str) -> str:
            return "".join(str(u).split()).replace("**", "^")
        wanted = {norm(u) for c in self._components for n, u, k in c.definitions if n == name and k in ("Input", "State") and u}
        if self._schema and name in self._schema.variables and self._schema.variables[name].unit:
            wanted.add(norm(self._schema.variables[name].unit))
        if supplied and wanted and norm(supplied) not in wanted:
            raise NotImplementedError(f"unit conversion is not available on the GPU path: {name!r} is supplied in "
                                      f"{supplied!r}, expected {sorted(wanted)}")

    def _node_names(self) -> List[str]:
        """One graph-node name per component: its type name, except that all but the LAST registered
        component of a type are "<type>#<k>".  The reference accepts several components that provide the
        same variables (builder.rs:531-559: the later one becomes the owner, with an edge from the earlier
        one); each of them writes index n+1 of the same series every step, so what stands is the write of
        whichever runs LAST in the execution order (runtime.rs:504-527).  For components of one built-in
        type -- same variables, same states -- the others' work is never seen: they keep their nodes (the
        edges shape the breadth-first order) but get no ensemble (_build_graph)."""
        last = {c.type_name: k for k, c in enumerate(self._components)}
        seen: Dict[str, int] = {}
        out = []
        for k, c in enumerate(self._components):
            if last[c.type_name] == k:
                out.append(c.type_name)
            else:
                seen[c.type_name] = seen.get(c.type_name, 0) + 1
                out.append(f"{c.type_name}#{seen[c.type_name]}")
        return out

    def _graph_order(self, aggregates, topological: bool = False) -> List[str]:
        """Execution order of the reference: nodes in registration order (root, components,
        aggregates), edges as ModelBuilder::build adds them, petgraph Bfs from the root (neighbours
        come out most-recently-added edge first).  builder.rs:448-701, runtime.rs:504-527.

        ``topological`` (an extension) orders the same nodes so that every edge points forwards:
        the order in which no component reads a value of the current step before it has been
        produced.  Every such order gives the same values; among the nodes that are ready the next
        one is of the class of the previous one if possible -- components that share a fused
        launch (csrc/lockstep.cpp) against those with a kernel of their own (ClimateUDEB,
        OceanCarbon, HalocarbonChemistry, host components) -- so that a step takes as few launches
        as the edges allow; within a class the breadth-first position decides."""
        names = ["<root>"]
        edges: Dict[int, List[int]] = {}
        endogenous: Dict[str, int] = {}
        pending: List[Tuple[int, str]] = []
        node_names = self._node_names()

        def add_edge(a: int, b: int) -> None:
            edges.setdefault(a, []).append(b)

        for comp, node_name in zip(self._components, node_names):
            node = len(names)
            names.append(node_name)
            has_dep = False
            for name, _, kind in comp.definitions:
                if kind not in ("Input", "State"):
                    continue
                if name in endogenous:
                    add_edge(endogenous[name], node)
                    has_dep = True
                elif name in aggregates:
                    pending.append((node, name))
                    has_dep = True
            if not has_dep:
                add_edge(0, node)
            for name, _, kind in comp.definitions:
                if kind in ("Output", "State"):
                    if name in endogenous:
                        add_edge(endogenous[name], node)
                    endogenous[name] = node
        for agg, (_, _, contributors, _) in aggregates.items():
            node = len(names)
            names.append(f"Aggregator:{agg}")
            has_dep = False
            for c in contributors:
                if c in endogenous:
                    add_edge(endogenous[c], node)
                    has_dep = True
            if not has_dep:
                add_edge(0, node)
            endogenous[agg] = node
        for node, name in pending:
            if name in endogenous:
                add_edge(endogenous[name], node)
        seen, order, queue = {0}, [], [0]
        while queue:
            n = queue.pop(0)
            if n:
                order.append(names[n])
            for m in reversed(edges.get(n, [])):
                if m not in seen:
                    seen.add(m)
                    queue.append(m)
        if not topological:
            return order
        rank = {name: k for k, name in enumerate(order)}
        for name in names[1:]:
            rank.setdefault(name, len(rank))
        indeg = {n: 0 for n in range(1, len(names))}
        for a, targets in edges.items():
            for b in set(targets):
                if a:
                    indeg[b] += 1
        own_kernel = {0: False}
        for k, comp in enumerate(self._components, start=1):
            own_kernel[k] = comp.type_name in OWN_KERNEL_TYPES or bool(getattr(comp, "is_python", False))
        ready = sorted((n for n, d in indeg.items() if d == 0), key=lambda n: rank[names[n]])
        topo: List[str] = []
        last = False
        while ready:
            ready.sort(key=lambda k: (own_kernel.get(k, False) != last, rank[names[k]]))
            n = ready.pop(0)
            last = own_kernel.get(n, False)
            topo.append(names[n])
            for m in set(edges.get(n, [])):
                indeg[m] -= 1
                if indeg[m] == 0:
                    ready.append(m)
        if len(topo) != len(names) - 1:
            raise ValueError("the component graph has a cycle")
        return topo

    def _build_graph(self, n_members: int, endogenous, sources, exo_names, aggregates,
                     execution_order: str = "reference", series_window: Optional[int] = None, output_stride: int = 0,
                     outputs: Optional[Sequence[str]] = None) -> "GraphModel":
        T, bounds = len(self._axis), self._axis.bounds()
        node_names = self._node_names()
        for c, node in zip(self._components, node_names):
            if c.type_name not in COMPONENT_KINDS and not getattr(c, "is_python", False):
                raise NotImplementedError(f"component {c.type_name} has no GPU kernel; supported: " + "; ".join(SUPPORTED))
            if node != c.type_name and getattr(c, "is_python", False):
                # host components of one class may declare different variables: they would all be live
                raise NotImplementedError(f"two Python components of the same type in one graph: {c.type_name}")
        if execution_order not in ("reference", "topological"):
            raise ValueError("execution_order must be 'reference' or 'topological'")
        order = self._graph_order(aggregates, topological=execution_order == "topological")
        missing = [n for n in node_names + [f"Aggregator:{a}" for a in aggregates] if n not in order]
        if missing:
            raise NotImplementedError(f"components not reachable from the graph root: {missing}")
        # of several components of one type the one that runs last stands (see _node_names): the others
        # take part in the ordering only, and the survivor goes by the plain type name from here on
        runs_last: Dict[str, str] = {}
        for n in order:
            if not n.startswith("Aggregator:"):
                runs_last[n.split("#")[0]] = n
        by_node = dict(zip(node_names, self._components))
        for at, n in enumerate(order):
            if n.startswith("Aggregator:") or runs_last[n.split("#")[0]] == n:
                continue
            # an overridden component's values ARE seen by whoever reads index n+1 between it and the survivor
            provided = {name for name, _, kind in by_node[n].definitions if kind in ("Output", "State")}
            for reader in order[at + 1:order.index(runs_last[n.split("#")[0]])]:
                if reader.startswith("Aggregator:"):
                    reads = set(aggregates[reader[len("Aggregator:"):]][2])
                else:
                    reads = {name for name, _, kind in by_node[reader].definitions
                             if kind == "Input" and sources.get((name, reader)) == "UpstreamOutput"}
                if provided & reads:
                    raise NotImplementedError(f"{reader} reads {sorted(provided & reads)} of {n} within the step, before "
                                              f"{runs_last[n.split('#')[0]]} overrides them")
        order = [n.split("#")[0] if not n.startswith("Aggregator:") else n for n in order
                 if n.startswith("Aggregator:") or runs_last[n.split("#")[0]] == n]
        components = [c for c, node in zip(self._components, node_names) if runs_last[c.type_name] == node]
        for c, node in zip(self._components, node_names):
            if runs_last[c.type_name] == node and node != c.type_name:
                sources = dict(sources)
                for name, _, kind in c.definitions:
                    if kind in ("Input", "State"):
                        sources[(name, c.type_name)] = sources[(name, node)]
        stream = C.c_void_p()
        L.check(L.load().rscm_gpu_stream_create(self._device, C.byref(stream)))
        ensembles: Dict[str, Ensemble] = {}
        var_home: Dict[str, Tuple[str, int]] = {}
        exogenous: Dict[str, np.ndarray] = {}
        links: List[Tuple[str, int]] = []
        model = GraphModel(self._axis, order, ensembles, var_home, links, exogenous, sources, stream, self._device, True)
        model._builder = self
        model._execution_order = execution_order
        windowed = series_window is not None and series_window < T
        model._windowed = windowed
        model._output_stride = int(output_stride) if windowed else 1
        want_out = None if outputs is None else set(outputs)

        def make(kind: int, out_names=None) -> Ensemble:
            """One ensemble of the graph; windowed graphs keep `series_window` rows of every series and
            every `output_stride`-th row of the requested outputs (all variables if none were named)."""
            if not windowed:
                return Ensemble(kind, n_members, bounds, device=self._device)
            ids = L.KIND_TABLE[kind][0]
            keep = None
            if want_out is not None:
                names = out_names if out_names is not None else [n for n, v in ids.items() if v > 0]
                fb = L.FOURBOX_VARS.get(kind)
                keep = [ids[n] for n in names if n in want_out or (fb and fb[0] in want_out and fb[1] <= ids[n] < fb[1] + 4)]
            return Ensemble(kind, n_members, bounds, device=self._device, window_rows=series_window,
                            output_stride=output_stride if keep is None or keep else 0, output_vars=keep)

        try:
            def params_of(values) -> np.ndarray:
                return np.repeat(np.array(values, dtype=np.float64)[:, None], n_members, axis=1)

            owner_of: Dict[str, str] = {}  # storage ensemble of a Python component's variable -> the component
            for comp in components:
                if getattr(comp, "is_python", False):
                    # a host component: its outputs live in one device series each (an aggregate-kind
                    # ensemble used as storage, never launched), so that GPU components can link to them
                    for name in comp.output_names():
                        sname = f"{comp.type_name}:{name}"
                        store = Ensemble(L.KIND_AGGREGATE, n_members, bounds, device=self._device)
                        ensembles[sname] = store
                        store.set_stream(stream.value)
                        store.set_params(params_of([0.0] * (1 + L.AG_NINPUTS)))
                        var_home[name] = (sname, 1)
                        owner_of[sname] = comp.type_name
                    model._host_nodes[comp.type_name] = _HostNode(comp)
                    model._feed_forward = False
                    continue
                ens = make(COMPONENT_KINDS[comp.type_name])
                ensembles[comp.type_name] = ens
                ens.set_stream(stream.value)
                base = _component_params(comp)
                ens.set_params(params_of(base))
                model.base_params[comp.type_name] = np.array(base, dtype=np.float64)
                for row, pname in enumerate(_component_param_names(comp)):
                    model.param_home[f"{comp.type_name}.{pname}"] = (comp.type_name, row)
                    # a bare name addresses the parameter only if no other component has one of that name
                    model.param_home[pname] = (comp.type_name, row) if pname not in model.param_home else ("", -1)
                if comp.type_name == "CarbonCycle":
                    ens.set_step_size(L.COMP_CARBON_CYCLE, comp.step_size)
                for name, _, kind in comp.definitions:
                    if kind in ("Output", "State"):
                        fb = L.FOURBOX_VARS.get(ens.kind)
                        if fb and name == fb[0]:
                            # A FourBox variable is four scalar series.  Whoever wants it as a scalar
                            # -- the schema (write transform: the scalar is what the collection stores,
                            # runtime.rs:452-470) or a reader (read transform, state/aggregating.rs:
                            # 162-176) -- gets sum(value * weight) over the boxes, formed by a Weighted
                            # aggregate ensemble that runs right after the producer, index by index.
                            tname = f"Transform:{name}"
                            tr = Ensemble(L.KIND_AGGREGATE, n_members, bounds, device=self._device,
                                          window_rows=series_window if windowed else None,
                                          output_stride=output_stride if (want_out is None or name in want_out) else 0)
                            ensembles[tname] = tr
                            tr.set_stream(stream.value)
                            w = self._grid_weights.get(GridType.FourBox, [0.25, 0.25, 0.25, 0.25])
                            tr.set_params(params_of([L.AG_OPERATIONS["Weighted"]] + list(w) + [0.0] * 4))
                            for k in range(4):
                                tr.link_input(k, ens, fb[1] + k, L.SRC_UPSTREAM)
                                links.append((tname, k))
                            at = order.index(comp.type_name) + 1
                            if execution_order == "topological":
                                # ... or, in the extension's order, after the own-kernel components that follow the
                                # producer and do not read the variable: the transform then shares their successors' launch
                                while (at < len(order) and order[at] in OWN_KERNEL_TYPES and order[at] in by_node and not any(
                                        v == name and kind in ("Input", "State") for v, _, kind in by_node[order[at]].definitions)):
                                    at += 1
                            order.insert(at, tname)
                            var_home[name] = (tname, 1)
                            stored_scalar = self._schema is not None and self._schema.grid_types.get(name) == GridType.Scalar
                            model._fourbox[name] = (comp.type_name, fb[1], stored_scalar)
                            continue
                        var_home[name] = (comp.type_name, ens.var_ids[name])
            # An aggregate ensemble takes eight contributors.  More are folded left to right through
            # partial-sum stages -- stage k adds up to seven further contributors to the partial of stage
            # k-1 -- which keeps compute_aggregate's order of additions (schema.rs:760-802: one running sum
            # over the contributors in declaration order, NaN skipped, all-NaN -> NaN), so the result
            # carries the same bits.  Weighted: the partial enters the next stage with weight 1.  A Mean of
            # more than eight is that Sum chain, a second chain that counts the non-NaN contributors, and a
            # last stage that divides the one by the other (sum / n as f64, NaN for n = 0).
            agg_stages: Dict[str, List[Tuple[str, List[str]]]] = {}

            def add_stage(agg: str, name: str, op: str, rows: List[str], weights: List[float], out_var: str, final: bool) -> None:
                ens = Ensemble(L.KIND_AGGREGATE, n_members, bounds, device=self._device,
                               window_rows=series_window if windowed else None,
                               output_stride=output_stride if (final and (want_out is None or agg in want_out)) else 0)
                ensembles[name] = ens
                ens.set_stream(stream.value)
                w = list(weights) + [0.0] * (L.AG_NINPUTS - len(weights))
                ens.set_params(params_of([L.AG_OPERATIONS[op]] + w))
                var_home[out_var] = (name, 1)
                agg_stages[agg].append((name, rows))

            def chain(agg: str, op: str, carry_op: str, contributors: List[str], wts: List[float], tag: str, final: bool) -> str:
                """Stages over `contributors`; returns the variable the last one writes."""
                k, n_stage, out_var = 0, 0, ""
                while True:
                    first = n_stage == 0
                    take = L.AG_NINPUTS if first else L.AG_NINPUTS - 1
                    last = len(contributors) - k <= take
                    chunk = contributors[k:k + take]
                    wchunk = wts[k:k + take] if wts else []
                    k += len(chunk)
                    is_final = final and last
                    name = f"Aggregator:{agg}" if is_final else f"Aggregator:{agg}#{tag}{n_stage}"
                    rows = chunk if first else [out_var] + chunk
                    wrow = wchunk if first else [1.0] + wchunk
                    out_var = agg if is_final else f"{agg}#partial{tag}{n_stage}"
                    add_stage(agg, name, op if first else carry_op, rows, wrow, out_var, is_final)
                    n_stage += 1
                    if last:
                        return out_var

            def scalar_aggregate(agg: str, op: str, contributors: List[str], weights) -> None:
                agg_stages[agg] = []
                if op == "Mean" and len(contributors) > L.AG_NINPUTS:
                    total = chain(agg, "Sum", "Sum", contributors, [], "", False)
                    count = chain(agg, "Count", "CountCarry", contributors, [], "count", False)
                    add_stage(agg, f"Aggregator:{agg}", "Quotient", [total, count], [], agg, True)
                else:
                    chain(agg, op, op, contributors, list(weights or []), "", True)

            for agg, (_, op, contributors, weights) in aggregates.items():
                contributors = list(contributors)
                grid = aggregates[agg].grid_type
                if grid == GridType.Hemispheric:
                    raise NotImplementedError(f"aggregate {agg!r}: no component of this package produces a Hemispheric variable")
                if grid == GridType.FourBox:
                    # AggregatorComponent::solve for a FourBox aggregate (schema.rs:902-923): compute_aggregate per region over the
                    # contributors' values of that region.  A FourBox variable is four scalar series here, so the aggregate is four
                    # scalar aggregates "<name>|box<k>" (k = 0..3: the reference's region order) over box k of every contributor,
                    # plus the scalar view of the result -- sum(value * weight) over the boxes, what a component that declares the
                    # variable as a scalar input reads (read transform, state/aggregating.rs:162-176) -- under the aggregate's name.
                    stages: List[Tuple[str, List[str]]] = []
                    homes = []
                    for b in range(4):
                        rows_b = []
                        for c in contributors:
                            cb = f"{c}|box{b}"
                            if cb not in var_home:
                                if c not in model._fourbox:
                                    raise ValueError(f"Grid type mismatch: contributor {c!r} of the FourBox aggregate {agg!r} is not a FourBox variable")
                                owner_c, first_c, _ = model._fourbox[c]
                                var_home[cb] = (owner_c, first_c + b)
                            rows_b.append(cb)
                        scalar_aggregate(f"{agg}|box{b}", op, rows_b, weights)
                        stages += agg_stages.pop(f"{agg}|box{b}")
                        homes.append(var_home[f"{agg}|box{b}"])
                    tname = f"Transform:{agg}"
                    tr = Ensemble(L.KIND_AGGREGATE, n_members, bounds, device=self._device,
                                  window_rows=series_window if windowed else None,
                                  output_stride=output_stride if (want_out is None or agg in want_out) else 0)
                    ensembles[tname] = tr
                    tr.set_stream(stream.value)
                    w = self._grid_weights.get(GridType.FourBox, [0.25, 0.25, 0.25, 0.25])
                    tr.set_params(params_of([L.AG_OPERATIONS["Weighted"]] + list(w) + [0.0] * 4))
                    var_home[agg] = (tname, 1)
                    stages.append((tname, [f"{agg}|box{b}" for b in range(4)]))
                    agg_stages[agg] = stages
                    model._fourbox_aggregates[agg] = homes
                    at = order.index(f"Aggregator:{agg}")
                    order[at:at + 1] = [n for n, _ in stages]
                    continue
                scalar_aggregate(agg, op, contributors, weights)
                stages = agg_stages[agg]
                if len(stages) > 1:  # the partial stages run right before the aggregate itself
                    at = order.index(f"Aggregator:{agg}")
                    order[at:at] = [n for n, _ in stages[:-1]]
            position = {name: k for k, name in enumerate(order)}

            def wire(owner: str, rows: Sequence[str], read_end: bool) -> None:
                ens = ensembles[owner]
                table = np.full((len(ens.input_rows) if ens.input_rows else 1, T), NAN)
                use_table = False
                for k, name in enumerate(rows):
                    if name in var_home:
                        prod, vid = var_home[name]
                        src = L.SRC_UPSTREAM if read_end or sources.get((name, owner)) == "UpstreamOutput" else L.SRC_EXOGENOUS
                        ens.link_input(k, ensembles[prod], vid, src)
                        links.append((owner, k))
                        if position[owner_of.get(prod, prod)] > position[owner]:
                            # lagged feedback, or -- the breadth-first order is not a topological one -- a
                            # component that runs before the producer of what it reads at n+1 and, like
                            # in the reference, finds NaN there
                            model._feed_forward = False
                            model._reads_unwritten = model._reads_unwritten or src == L.SRC_UPSTREAM
                            L.check(L.load().rscm_ens_set_link_order_check(ens._h, 0))
                    else:
                        if name in endogenous:
                            raise NotImplementedError(f"{name!r} is a FourBox variable: it cannot feed the scalar input of {owner}")
                        vals = self._exogenous_on_axis(name, exo_names + list(rows))
                        use_table = True
                        if vals is not None:
                            table[k] = vals
                        exogenous[name] = table[k].copy()  # never supplied: a NaN series (builder.rs:772-780)
                if use_table:
                    ens.set_forcing(table if ens.input_rows else table[0])

            model.param_home = {k: v for k, v in model.param_home.items() if v[1] >= 0}
            for comp in components:
                if getattr(comp, "is_python", False):
                    node = model._host_nodes[comp.type_name]
                    for name in comp.input_names():
                        node.sources[name] = sources.get((name, comp.type_name), "Exogenous")
                        node.series[name] = np.full((T, n_members), NAN)
                        if name in var_home:
                            node.device_reads[name] = var_home[name]
                            if name in self._initial:
                                node.series[name][0] = self._initial[name]
                        else:
                            if name in endogenous:
                                raise NotImplementedError(f"{name!r} is a FourBox variable: it cannot feed the scalar input of {comp.type_name}")
                            vals = self._exogenous_on_axis(name, exo_names + [name])
                            if vals is not None:
                                node.series[name][:] = vals[:, None]
                            exogenous[name] = node.series[name][:, 0].copy()
                    continue
                ens = ensembles[comp.type_name]
                rows = ens.input_rows or [n for n, v in ens.var_ids.items() if v == 0]
                wire(comp.type_name, list(rows), ens.kind == L.KIND_UDEB)
            for agg in aggregates:
                for name, rows in agg_stages[agg]:
                    wire(name, rows, True)
            for name, (owner, vid) in var_home.items():
                if name in self._initial:
                    ensembles[owner].set_initial(vid, self._initial[name])
            for agg, homes in model._fourbox_aggregates.items():
                if agg in self._initial:  # one scalar for the four regions (builder.rs:797-804); the scalar view was set above
                    w = self._grid_weights.get(GridType.FourBox, [0.25, 0.25, 0.25, 0.25])
                    x0 = self._initial[agg]
                    for owner_b, vid_b in homes:
                        ensembles[owner_b].set_initial(vid_b, x0)
                    s0 = 0.0
                    for wk in w:
                        s0 = s0 + x0 * wk
                    ensembles[f"Transform:{agg}"].set_initial(1, s0)
            for owner, ens in list(ensembles.items()):
                fb = L.FOURBOX_VARS.get(ens.kind)
                if fb and fb[0] in self._initial:  # a FourBox state initialised with one scalar (builder.rs:797-804)
                    w = self._grid_weights.get(GridType.FourBox, [0.25, 0.25, 0.25, 0.25])
                    x0 = self._initial[fb[0]]
                    for v in range(fb[1], fb[1] + 4):
                        ens.set_initial(v, x0)
                    s0 = 0.0
                    for wk in w:  # the scalar view of index 0, summed like aggregate_global
                        s0 = s0 + x0 * wk
                    ensembles[f"Transform:{fb[0]}"].set_initial(1, s0)
        except Exception:
            model.close()
            raise
        return model

    def build(self, n_members: int = 1, store_series: bool = True, execution_order: str = "reference",
              series_window: Optional[int] = None, output_stride: int = 0,
              outputs: Optional[Sequence[str]] = None) -> "Model":
        """``execution_order`` (graphs without a fused kernel only): "reference" steps the components
        in the reference's breadth-first order, which can run a component before the producer of a
        value it reads at the end of the step (it then reads NaN, which an aggregate skips);
        "topological" is the order in which that cannot happen.

        ``series_window`` (an extension, for long axes and large ensembles): run the model as a graph of
        linked ensembles that keep only a sliding window of that many rows of every series -- what a
        step and its consumers read -- plus every ``output_stride``-th row of ``outputs`` (variable
        names; None: every variable).  The reference holds whole collections, one member at a time
        (model/builder.rs:735-830); 1e5 members x 9001 monthly points x 36 series do not fit a GPU that
        way.  ``get_series(name, t_stride=output_stride)`` reads the kept rows."""
        endogenous, sources, exo_names, aggregates = self._resolve()
        if series_window is not None:
            if any(getattr(c, "is_python", False) for c in self._components):
                raise NotImplementedError("Python components read whole host series: no series_window for such graphs")
            return self._build_graph(n_members, endogenous, sources, exo_names, aggregates, execution_order,
                                     series_window, output_stride, outputs)
        if any(getattr(c, "is_python", False) for c in self._components):
            return self._build_graph(n_members, endogenous, sources, exo_names, aggregates, execution_order)
        types = [c.type_name for c in self._components]
        erf = "Effective Radiative Forcing"
        if types == ["TwoLayer"] and not aggregates:
            kind = L.KIND_TWO_LAYER
            forcing = self._exogenous_on_axis(erf, exo_names)
            if forcing is None:  # builder.rs:772-780: NaN series; run() then yields NaN
                forcing = np.full(len(self._axis), NAN)
            src = L.SRC_EXOGENOUS
            params = [self._components[0].parameters[k] for k in TL_PARAM_ORDER]
            h = {L.COMP_TWO_LAYER: 0.1}
        elif (types == ["CarbonCycle", "CO2ERF", "TwoLayer"] and list(aggregates) == [erf]
              and aggregates[erf][1] == "Sum"
              and aggregates[erf][2] == ["Effective Radiative Forcing|CO2"]
              # the fused coupled kernel carries ONE conc_pi row; two different pre-industrial
              # concentrations run as the same graph of linked ensembles (one launch as well: all light)
              and self._components[0].parameters["conc_pi"] == self._components[1].parameters["conc_pi"]):
            kind = L.KIND_COUPLED
            cc, ce, tl = self._components
            assert sources[("Surface Temperature", "CarbonCycle")] == "Exogenous"
            assert sources[(erf, "TwoLayer")] == "UpstreamOutput"
            forcing = self._exogenous_on_axis("Emissions|CO2|Anthropogenic", exo_names)
            if forcing is None:
                forcing = np.full(len(self._axis), NAN)
            src = L.SRC_EXOGENOUS
            params = ([tl.parameters[k] for k in TL_PARAM_ORDER] +
                      [cc.parameters["tau"], cc.parameters["conc_pi"],
                       cc.parameters["alpha_temperature"], ce.parameters["erf_2xco2"]])
            h = {L.COMP_TWO_LAYER: 0.1, L.COMP_CARBON_CYCLE: cc.step_size}
        elif types == ["ClimateUDEB"] and not aggregates:
            kind = L.KIND_UDEB
            forcing = self._exogenous_on_axis(erf, exo_names)
            if forcing is None:
                forcing = np.full(len(self._axis), NAN)
            src = L.SRC_EXOGENOUS
            params = self._components[0].param_vector()
            h = {}
        elif len(types) == 1 and types[0] in STATELESS_KINDS and not aggregates:
            kind = STATELESS_KINDS[types[0]]
            rows = [self._exogenous_on_axis(name, exo_names) for name in L.KIND_TABLE[kind][2]]
            forcing = np.stack([np.full(len(self._axis), NAN) if r is None else r for r in rows])
            src = L.SRC_EXOGENOUS
            params = self._components[0].param_vector()
            h = {}
        else:
            # any other graph of built-in components: one ensemble per component, linked on the device
            return self._build_graph(n_members, endogenous, sources, exo_names, aggregates, execution_order)
        if not store_series and kind != L.KIND_TWO_LAYER:
            store_series = True  # likelihood-only handles exist for the two-layer kind
        ens = Ensemble(kind, n_members, self._axis.bounds(), device=self._device,
                       store_series=store_series)
        for comp_id, step in h.items():
            ens.set_step_size(comp_id, step)
        ens.set_params(np.repeat(np.array(params, dtype=np.float64)[:, None], n_members, axis=1))
        ens.set_forcing(forcing, None, src)
        for name, vid in ens.var_ids.items():
            if vid > 0 and name in self._initial:
                ens.set_initial(vid, self._initial[name])
            elif vid > 0 and "|" in name and name.split("|")[0] in self._initial and kind == L.KIND_UDEB:
                # a FourBox state initialised with one scalar sets all four regions (builder.rs:797-804)
                ens.set_initial(vid, self._initial[name.split("|")[0]])
        param_order = {L.KIND_TWO_LAYER: TL_PARAM_ORDER, L.KIND_COUPLED: CP_PARAM_ORDER,
                       L.KIND_UDEB: L.UD_PARAM_NAMES, L.KIND_GHG_FORCING: L.GH_PARAM_NAMES,
                       L.KIND_OZONE_FORCING: L.OZ_PARAM_NAMES, L.KIND_AEROSOL_DIRECT: L.AD_PARAM_NAMES,
                       L.KIND_AEROSOL_INDIRECT: L.AI_PARAM_NAMES, L.KIND_CH4_CHEMISTRY: L.CH4_PARAM_NAMES,
                       L.KIND_N2O_CHEMISTRY: L.N2O_PARAM_NAMES, L.KIND_CO2_BUDGET: L.CB_PARAM_NAMES,
                       L.KIND_TERRESTRIAL_CARBON: L.TC_PARAM_NAMES, L.KIND_OCEAN_CARBON: L.OC_PARAM_NAMES,
                       L.KIND_HALOCARBON: L.HC_PARAM_NAMES, L.KIND_FOURBOX_OHU: L.FB_PARAM_NAMES,
                       L.KIND_OSPP: L.SP_PARAM_NAMES}[kind]
        model = Model(ens, self._axis, sources, endogenous, forcing, dict(self._initial), param_order,
                      np.array(params, dtype=np.float64))
        model._builder = self
        return model


def save_checkpoint(path, ck: Dict[str, object]) -> None:
    """A checkpoint (of an ``Ensemble`` or a ``GraphModel``) as one ``.npz`` of plain arrays: nested
    keys joined with ``/``; nothing is pickled, so loading executes nothing from the file."""
    flat: Dict[str, np.ndarray] = {}

    def walk(prefix: str, value) -> None:
        if isinstance(value, dict):
            flat[prefix + "/__dict__"] = np.array(len(value))
            for k, v in value.items():
                if "/" in str(k):
                    raise ValueError(f"key {k!r} contains '/'")
                walk(f"{prefix}/{k}", v)
        elif value is None:
            flat[prefix + "/__none__"] = np.array(0)
        elif isinstance(value, (list, tuple)) and all(isinstance(x, str) for x in value):
            flat[prefix + "/__strings__"] = np.array(list(value), dtype=np.str_)
        else:
            flat[prefix] = np.asarray(value)

    walk("", ck)
    np.savez(path, **{k.lstrip("/").replace("|", "__BAR__"): v for k, v in flat.items()})


def load_checkpoint(path) -> Dict[str, object]:
    out: Dict[str, object] = {}
    with np.load(path, allow_pickle=False) as z:
        for raw in z.files:
            parts = raw.replace("__BAR__", "|").split("/")
            node = out
            leaf = parts[-1]
            if leaf in ("__dict__", "__none__", "__strings__"):
                parts, marker = parts[:-1], leaf
            else:
                marker = None
            for p_ in parts[:-1]:
                node = node.setdefault(p_, {})
            if not parts:
                continue
            if marker == "__dict__":
                node.setdefault(parts[-1], {})
            elif marker == "__none__":
                node[parts[-1]] = None
            elif marker == "__strings__":
                node[parts[-1]] = [str(x) for x in z[raw]]
            else:
                v = z[raw]
                node[parts[-1]] = v.item() if v.ndim == 0 else v
    return out


# components whose step is a kernel of their own (csrc/lockstep.cpp, fusable()); the others share fused launches
OWN_KERNEL_TYPES = ("ClimateUDEB", "OceanCarbon", "HalocarbonChemistry")

# type name -> ensemble kind, for graphs assembled from linked ensembles
COMPONENT_KINDS = {"TwoLayer": L.KIND_TWO_LAYER, "ClimateUDEB": L.KIND_UDEB, "CarbonCycle": L.KIND_CARBON_CYCLE,
                   "CO2ERF": L.KIND_CO2_ERF, **STATELESS_KINDS}


def _component_param_names(comp) -> Tuple[str, ...]:
    if comp.type_name == "TwoLayer":
        return tuple(TL_PARAM_ORDER)
    if comp.type_name == "CarbonCycle":
        return tuple(L.CC_PARAM_NAMES)
    if comp.type_name == "CO2ERF":
        return tuple(L.CE_PARAM_NAMES)
    kind = COMPONENT_KINDS[comp.type_name]
    table = {L.KIND_UDEB: L.UD_PARAM_NAMES, L.KIND_GHG_FORCING: L.GH_PARAM_NAMES, L.KIND_OZONE_FORCING: L.OZ_PARAM_NAMES,
             L.KIND_AEROSOL_DIRECT: L.AD_PARAM_NAMES, L.KIND_AEROSOL_INDIRECT: L.AI_PARAM_NAMES,
             L.KIND_CH4_CHEMISTRY: L.CH4_PARAM_NAMES, L.KIND_N2O_CHEMISTRY: L.N2O_PARAM_NAMES,
             L.KIND_CO2_BUDGET: L.CB_PARAM_NAMES, L.KIND_TERRESTRIAL_CARBON: L.TC_PARAM_NAMES,
             L.KIND_OCEAN_CARBON: L.OC_PARAM_NAMES, L.KIND_HALOCARBON: L.HC_PARAM_NAMES,
             L.KIND_FOURBOX_OHU: L.FB_PARAM_NAMES, L.KIND_OSPP: L.SP_PARAM_NAMES}
    return tuple(table[kind])


def _component_params(comp) -> List[float]:
    if comp.type_name == "TwoLayer":
        return [comp.parameters[k] for k in TL_PARAM_ORDER]
    if comp.type_name == "CarbonCycle":
        return [comp.parameters[k] for k in L.CC_PARAM_NAMES]
    if comp.type_name == "CO2ERF":
        return [comp.parameters[k] for k in L.CE_PARAM_NAMES]
    return list(comp.param_vector())


class _HostNode:
    """A Python component inside a GraphModel: the series it reads (host copies, refreshed row by
    row from the device), where those come from, and the source classification of each."""

    def __init__(self, comp):
        self.comp = comp
        self.series: Dict[str, np.ndarray] = {}
        self.device_reads: Dict[str, Tuple[str, int]] = {}
        self.sources: Dict[str, str] = {}


class GraphModel:
    """A component graph the fused kernels do not cover, run as one ensemble per component (and per
    schema aggregate) whose inputs are linked on the device (``rscm_ens_link_input``) and which are
    stepped in the reference's order: petgraph BFS from the root over the edges
    ``ModelBuilder::build`` adds (builder.rs:487-560, runtime.rs:368-497).  Same surface as ``Model``.

    ``ensembles`` maps a component's type name (``"Aggregator:<name>"`` for aggregates) to its
    ``Ensemble``, e.g. to give the members different parameters before ``run()``."""

    def __init__(self, axis: TimeAxis, order: List[str], ensembles: Dict[str, Ensemble], var_home, links,
                 exogenous: Dict[str, np.ndarray], sources, stream, device: int, feed_forward: bool):
        self._axis = axis
        self._order = order
        self.ensembles = ensembles
        self._var_home = var_home          # variable name -> (owner, variable id)
        self._links = links                # (consumer, row) pairs, for teardown
        self._exogenous = exogenous
        self._sources = sources
        self._stream = stream
        self._device = device
        self._feed_forward = feed_forward
        self._host_nodes: Dict[str, "_HostNode"] = {}  # Python components, stepped on the host
        self._reads_unwritten = False  # some component reads index n+1 of a producer that runs after it
        self._fourbox: Dict[str, Tuple[str, int, bool]] = {}  # FourBox variable -> (producer, first id, stored as scalar)
        self._fourbox_aggregates: Dict[str, List[Tuple[str, int]]] = {}  # FourBox aggregate -> (ensemble, id) of each region
        self._windowed = False      # ensembles keep a sliding window of rows + strided outputs (build(series_window=...))
        self._output_stride = 1
        self.time_index = 0
        # component parameters: "Type.name" -> (owner, row); bare names too where they are unique
        self.param_home: Dict[str, Tuple[str, int]] = {}
        self.base_params: Dict[str, np.ndarray] = {}

    @property
    def n_members(self) -> int:
        return next(iter(self.ensembles.values())).n_members

    def set_mode(self, mode: int) -> None:
        """Arithmetic mode of every ensemble of the graph (``MODE_FAST``: fused multiply-adds in
        the two-layer RK4 and the OceanCarbon convolution; the other kinds compute the same either way)."""
        for ens in self.ensembles.values():
            ens.set_mode(mode)

    def rewind(self) -> None:
        """Back to a fresh model.  Where a component runs ahead of a producer it reads at n+1, the
        rows of the previous run must not be found there: they are cleared to NaN."""
        for ens in self.ensembles.values():
            if self._reads_unwritten:
                ens.clear_series()
            else:
                ens.rewind()
        self.time_index = 0

    _builder: Optional["ModelBuilder"] = None
    _execution_order = "reference"

    def to_toml(self) -> str:
        from . import serialise
        return serialise.dumps(serialise.describe(self._builder, self))

    from_toml = staticmethod(lambda text: Model.from_toml(text))

    def as_dot(self) -> str:
        from . import serialise
        return serialise.as_dot(self._builder)

    def debug_info(self, format: str = "plain") -> str:
        """Execution order, variable flow and source classification (Model::debug_info)."""
        from . import serialise
        return serialise.debug_info(self._builder, self, format)

    def checkpoint(self) -> Dict[str, object]:
        """Everything needed to continue from the current step in another model object built from
        the same builder: per ensemble the current row of every stored variable (linked consumers
        read outputs as well as states), the rows the chemistry looks back at, the internal
        component states (runtime.rs:270-282)."""
        if self._host_nodes:
            raise NotImplementedError("checkpoints of graphs with Python components are not available")
        for ens in self.ensembles.values():
            ens.sync()
        return {"time_index": self.time_index, "order": list(self._order),
                "ensembles": {name: ens.checkpoint(all_variables=True) for name, ens in self.ensembles.items()}}

    def restore(self, ck: Dict[str, object]) -> None:
        if ck["order"] != list(self._order) or set(ck["ensembles"]) != set(self.ensembles):
            raise ValueError("checkpoint does not match this graph")
        # In the reference's execution order a component can read index n+1 of a producer that runs
        # after it and must find NaN there, as in the collection the checkpoint was taken from: rows
        # this (already advanced) model wrote beyond the checkpoint must not survive the roll-back.
        for name, ens in self.ensembles.items():
            ens.restore(ck["ensembles"][name], clear_later_rows=self._reads_unwritten)
        self.time_index = int(ck["time_index"])

    def variable_home(self, name: str) -> Tuple[Ensemble, int]:
        """The ensemble and variable id that hold the scalar series of ``name``."""
        if name not in self._var_home:
            raise KeyError(f"Model output missing variable: {name}")
        owner, vid = self._var_home[name]
        return self.ensembles[owner], vid

    def get_series(self, name: str, **kw) -> np.ndarray:
        ens, vid = self.variable_home(name)
        return ens.get_series(vid, **kw)

    def set_member_params(self, names: Sequence[str], values) -> None:
        """``values[N][len(names)]``: per-member values of the named component parameters; every
        other parameter keeps the value its component was built with."""
        v = np.asarray(values, dtype=np.float64)
        if v.ndim != 2 or v.shape != (self.n_members, len(names)):
            raise ValueError(f"Expected {len(names)} parameters for {self.n_members} members, got {v.shape}")
        per_owner: Dict[str, np.ndarray] = {}
        for k, name in enumerate(names):
            owner, row = self.param_home[name]
            if owner not in per_owner:
                per_owner[owner] = np.repeat(self.base_params[owner][:, None], self.n_members, axis=1)
            per_owner[owner][row] = v[:, k]
        for owner, full in per_owner.items():
            self.ensembles[owner].set_params(full)

    def variable_sources(self) -> Dict[Tuple[str, str], str]:
        return dict(self._sources)

    def current_time(self) -> float:
        return self._axis.at(self.time_index)

    def current_time_bounds(self) -> Tuple[float, float]:
        return self._axis.at_bounds(self.time_index)

    def _sync(self) -> None:
        next(iter(self.ensembles.values())).sync()  # one stream for all

    def _host_step(self, name: str, n: int) -> None:
        """One step of a Python component: the rows it may read come back from the device, ``solve`` runs
        per member, the new rows go into the component's storage series on the device."""
        node = self._host_nodes[name]
        T = len(self._axis)
        for var, (owner, vid) in node.device_reads.items():
            hi = min(n + 2, T)
            node.series[var][n:hi] = self.ensembles[owner].get_series(vid, n, hi)
        t0, t1 = self._axis.at_bounds(n)
        outs = {v: np.full(self.n_members, NAN) for v in node.comp.output_names()}
        for m in range(self.n_members):
            for var, value in node.comp.solve_member(t0, t1, node.series, m, n, node.sources).items():
                if var not in outs:
                    raise KeyError(f"{name}.solve returned {var!r}, which it does not declare as an output")
                outs[var][m] = value
        for var, row in outs.items():
            store = self.ensembles[f"{name}:{var}"]
            store.set_state(1, n + 1, row)
            store.set_time_index(n + 1)

    def step(self) -> None:
        if not self.time_index < len(self._axis) - 1:
            raise RuntimeError("assertion failed: self.time_index < self.time_axis.len() - 1")
        if self._host_nodes:
            for name in self._order:
                if name in self._host_nodes:
                    self._host_step(name, self.time_index)
                else:
                    self.ensembles[name].run(self.time_index + 1, sync=False)
        else:  # one native call: consecutive light components share a launch (rscm_ens_run_lockstep)
            run_lockstep([self.ensembles[name] for name in self._order], self.time_index + 1, sync=False)
        self.time_index += 1
        self._sync()

    def run(self) -> None:
        last = len(self._axis) - 1
        if self._host_nodes:  # Python components: the host takes part in every step
            while self.time_index < last:
                self.step()
            return
        if self._feed_forward and not self._windowed:  # no edge points backwards: every producer can finish before its consumers start
            for name in self._order:
                self.ensembles[name].run(last, sync=False)
        else:
            run_lockstep([self.ensembles[name] for name in self._order], last, sync=False)
        self.time_index = last
        self._sync()

    def finished(self) -> bool:
        return self.time_index == len(self._axis) - 1

    def timeseries(self, member: int = 0) -> TimeseriesCollection:
        if self._windowed:
            raise NotImplementedError("a windowed model keeps every output_stride-th row only: read them with "
                                      "get_series(name, t_stride=output_stride)")
        coll = TimeseriesCollection()
        for name, vals in self._exogenous.items():
            coll.add_timeseries(name, Timeseries(vals, self._axis, "", InterpolationStrategy.Linear), VariableType.Exogenous)
        skip = set()
        for name, (owner, first, stored_scalar) in self._fourbox.items():
            if stored_scalar:
                continue  # the collection holds the aggregated scalar (write transform)
            ens = self.ensembles[owner]
            boxes = np.stack([ens.get_series(v, m_begin=member, m_end=member + 1)[:, 0] for v in range(first, first + 4)], axis=1)
            coll.add_fourbox_timeseries(name, FourBoxTimeseries(boxes, self._axis, "K" if ens.kind == L.KIND_UDEB else "W/m^2"))
            skip.add(name)
        for name, homes in self._fourbox_aggregates.items():
            boxes = np.stack([self.ensembles[o].get_series(v, m_begin=member, m_end=member + 1)[:, 0] for o, v in homes], axis=1)
            coll.add_fourbox_timeseries(name, FourBoxTimeseries(boxes, self._axis, ""))
            skip.add(name)
        for name, (owner, vid) in self._var_home.items():
            if name in skip or "#partial" in name or "|box" in name:  # partial sums and the per-region series of FourBox variables are internal
                continue
            vals = self.ensembles[owner].get_series(vid, m_begin=member, m_end=member + 1)[:, 0]
            coll.add_timeseries(name, Timeseries(vals, self._axis, "", InterpolationStrategy.Linear), VariableType.Endogenous)
        return coll

    def close(self) -> None:
        for consumer, row in self._links:
            if self.ensembles[consumer]._h:
                self.ensembles[consumer].unlink_input(row)
        self._links = []
        for ens in self.ensembles.values():
            ens.close()
        if self._stream is not None:
            L.check(L.load().rscm_gpu_stream_destroy(self._device, self._stream))
            self._stream = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Model:
    """python/rscm/_lib/core/__init__.pyi:563-628 over one device-resident ensemble.  With
    ``n_members == 1`` it behaves like the reference's ``Model``; with more it is the batch the
    calibration front drives (all members share the graph, axis, forcing and initial values and
    differ in parameters)."""

    def __init__(self, ens: Ensemble, axis: TimeAxis, sources, endogenous, forcing, initial,
                 param_order, base_params):
        self.ensemble = ens
        self._axis = axis
        self._sources = sources
        self._endogenous = endogenous
        self._forcing = forcing
        self._initial = initial
        self.param_order = tuple(param_order)
        self.base_params = base_params

    _builder: Optional["ModelBuilder"] = None

    @property
    def n_members(self) -> int:
        return self.ensemble.n_members

    @property
    def time_index(self) -> int:
        return self.ensemble.time_index

    def checkpoint(self) -> Dict[str, object]:
        return self.ensemble.checkpoint()

    def restore(self, ck: Dict[str, object]) -> None:
        self.ensemble.restore(ck)

    def to_toml(self) -> str:
        """Model::to_toml (runtime.rs:270-300): description + state as TOML text (rscm_amd/serialise.py)."""
        from . import serialise
        return serialise.dumps(serialise.describe(self._builder, self))

    @staticmethod
    def from_toml(text: str):
        """Model::from_toml: the model a ``to_toml`` text describes, at the step it was taken at."""
        from . import serialise
        return serialise.rebuild(serialise.loads(text))

    def as_dot(self) -> str:
        from . import serialise
        return serialise.as_dot(self._builder)

    def debug_info(self, format: str = "plain") -> str:
        """Execution order, variable flow and source classification (Model::debug_info)."""
        from . import serialise
        return serialise.debug_info(self._builder, self, format)

    def variable_sources(self) -> Dict[Tuple[str, str], str]:
        return dict(self._sources)

    def current_time(self) -> float:
        return self._axis.at(self.ensemble.time_index)

    def current_time_bounds(self) -> Tuple[float, float]:
        return self._axis.at_bounds(self.ensemble.time_index)

    def step(self) -> None:
        if not self.ensemble.time_index < len(self._axis) - 1:
            raise RuntimeError("assertion failed: self.time_index < self.time_axis.len() - 1")
        self.ensemble.step()

    def run(self) -> None:
        self.ensemble.run()

    def finished(self) -> bool:
        return self.ensemble.finished()

    def timeseries(self, member: int = 0) -> TimeseriesCollection:
        coll = TimeseriesCollection()
        input_name = [n for n, v in self.ensemble.var_ids.items() if v == 0][0]
        fourbox = L.FOURBOX_VARS.get(self.ensemble.kind)
        fb_ids = ()
        if fourbox:  # a FourBox variable is stored as four scalar series
            fb_ids = tuple(range(fourbox[1], fourbox[1] + 4))
            boxes = np.stack([self.ensemble.get_series(v, m_begin=member, m_end=member + 1)[:, 0]
                              for v in fb_ids], axis=1)
            unit = "K" if self.ensemble.kind == L.KIND_UDEB else "W/m^2"
            coll.add_fourbox_timeseries(fourbox[0], FourBoxTimeseries(boxes, self._axis, unit))
        for name, vid in self.ensemble.var_ids.items():
            if vid in fb_ids:
                continue
            if vid == 0 and self.ensemble.input_rows:
                for k, input_row in enumerate(self.ensemble.input_rows):  # the input block's series
                    coll.add_timeseries(input_row, Timeseries(self._forcing[k], self._axis, "",
                                                              InterpolationStrategy.Linear), VariableType.Exogenous)
                continue
            if vid == 0:
                vals, vt = self._forcing, VariableType.Exogenous
            else:
                vals = self.ensemble.get_series(vid, m_begin=member, m_end=member + 1)[:, 0]
                vt = VariableType.Endogenous
            coll.add_timeseries(name, Timeseries(vals, self._axis, "", InterpolationStrategy.Linear), vt)
        del input_name
        return coll

    def close(self) -> None:
        self.ensemble.close()
