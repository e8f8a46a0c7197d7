"""``ClimateUDEBBuilder`` and ``GhgForcingBuilder`` -- mirror of ``rscm.magicc`` for the climate
core and the greenhouse-gas forcing (python/rscm/_lib/magicc.pyi;
crates/rscm-magicc/src/climate/udeb/mod.rs, crates/rscm-magicc/src/forcing/ghg.rs and their
parameter structs under crates/rscm-magicc/src/parameters/).  Unspecified parameters take the
structs' ``Default`` (``#[serde(default)]``)."""
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


class GhgForcing(Component):
    type_name = "GhgForcing"
    definitions = [("Atmospheric Concentration|CO2", "ppm", "Input"),
                   ("Atmospheric Concentration|CH4", "ppb", "Input"),
                   ("Atmospheric Concentration|N2O", "ppb", "Input"),
                   ("Effective Radiative Forcing|CO2", "W/m^2", "Output"),
                   ("Effective Radiative Forcing|CH4", "W/m^2", "Output"),
                   ("Effective Radiative Forcing|N2O", "W/m^2", "Output")]

    def param_vector(self):
        return [float(self.parameters[k]) for k in L.GH_PARAM_NAMES]


class GhgForcingBuilder(ComponentBuilder):
    component_cls = GhgForcing

    @classmethod
    def from_parameters(cls, parameters: Dict[str, float]):
        p = dict(zip(L.GH_PARAM_NAMES, L.GH_DEFAULTS))
        for k, v in parameters.items():
            if k not in p:  # serde: unknown field
                raise ValueError(f"unknown field `{k}`")
            if k == "method":
                if v not in L.GH_METHODS:  # serde: unknown variant
                    raise ValueError(f"unknown variant `{v}`, expected `Ipcctar` or `Olbl`")
                p[k] = L.GH_METHODS[v]
            else:
                p[k] = float(v)
        return cls(p)
