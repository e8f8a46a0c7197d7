"""``ClimateUDEBBuilder`` -- mirror of ``rscm.magicc`` for the climate core
(python/rscm/_lib/magicc.pyi; crates/rscm-magicc/src/climate/udeb/mod.rs,
crates/rscm-magicc/src/parameters/climate_udeb.rs).  Unspecified parameters take
``ClimateUDEBParameters::default()`` (``#[serde(default)]``)."""
from __future__ import annotations

from typing import Dict

from . import _lib as L
from .core import Component, ComponentBuilder


class ClimateUDEB(Component):
    type_name = "ClimateUDEB"
    definitions = [("Effective Radiative Forcing", "W/m^2", "Input"),
                   ("Heat Uptake", "W/m^2", "Output"),
                   ("Ocean Heat Content", "J/m^2", "Output"),
                   ("Sea Surface Temperature", "K", "Output"),
                   ("Surface Temperature", "K", "State")]  # FourBox grid

    def param_vector(self):
        return [float(self.parameters[k]) for k in L.UD_PARAM_NAMES]


class ClimateUDEBBuilder(ComponentBuilder):
    component_cls = ClimateUDEB

    @classmethod
    def from_parameters(cls, parameters: Dict[str, float]):
        p = dict(zip(L.UD_PARAM_NAMES, L.UD_DEFAULTS))
        for k, v in parameters.items():
            if k == "rf_regions_co2":
                for j, x in enumerate(v):
                    p[f"rf_regions_co2_{j}"] = float(x)
            elif k in p:
                p[k] = float(v)
            else:  # serde: unknown field
                raise ValueError(f"unknown field `{k}`")
        return cls(p)
