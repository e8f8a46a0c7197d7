"""``TwoLayerBuilder`` -- mirror of ``rscm.two_layer`` (python/rscm/_lib/two_layer.pyi;
component: crates/rscm-two-layer/src/component.rs:38-251)."""
from __future__ import annotations

from typing import Dict

import numpy as np

from . import _lib as L
from .core import Component, ComponentBuilder, TimeseriesCollection
from .ensemble import Ensemble


class ScalarValue:
    """StateValue::Scalar as the Python binding returns it (``.as_scalar()``)."""

    def __init__(self, v: float):
        self._v = float(v)

    def as_scalar(self) -> float:
        return self._v

    def __float__(self) -> float:
        return self._v


class TwoLayer(Component):
    type_name = "TwoLayer"
    definitions = [("Effective Radiative Forcing", "W/m^2", "Input"),
                   ("Surface Temperature", "K", "State"),
                   ("Deep Ocean Temperature", "K", "State")]
    step_size = 0.1  # hard-coded in the reference (component.rs:240)

    def solve(self, t_current: float, t_next: float, collection: TimeseriesCollection,
              device: int = 0) -> Dict[str, ScalarValue]:
        """One ``Component::solve`` on the GPU (crates/rscm-core/src/python/component.rs:66-87):
        inputs are read from ``collection`` at ``t_current`` (Exogenous -> at_start)."""
        t0, t1 = float(t_current), float(t_next)
        vals = {}
        for name in self.input_names():
            ts = collection.get_timeseries_by_name(name)
            if ts is None:
                raise KeyError(f"No timeseries with variable='{name}'")
            idx = ts.time_axis.index_of(t0)
            if idx is None:
                raise RuntimeError(f"time {t0} not found in the time axis of {name!r}")
            vals[name] = ts.at(idx)
        order = ("lambda0", "a", "efficacy", "eta", "heat_capacity_surface", "heat_capacity_deep")
        with Ensemble(L.KIND_TWO_LAYER, 1, [t0, t1, t1 + (t1 - t0)], device=device) as e:
            e.set_params(np.array([[self.parameters[k]] for k in order]))
            e.set_forcing([vals["Effective Radiative Forcing"]] * 2)
            e.set_initial("Surface Temperature", vals["Surface Temperature"])
            e.set_initial("Deep Ocean Temperature", vals["Deep Ocean Temperature"])
            e.step()
            return {"Surface Temperature": ScalarValue(e.get_series(1, 1, 2)[0, 0]),
                    "Deep Ocean Temperature": ScalarValue(e.get_series(2, 1, 2)[0, 0])}


class TwoLayerBuilder(ComponentBuilder):
    component_cls = TwoLayer
    required = ("lambda0", "a", "efficacy", "eta", "heat_capacity_surface", "heat_capacity_deep")
