"""``CarbonCycleBuilder`` / ``CO2ERFBuilder`` -- mirror of ``rscm.components`` for the coupled
chain (python/rscm/_lib/components.pyi:23-120; crates/rscm-components/src/components/
carbon_cycle.rs:24-94, co2_erf.rs:17-52) -- and the two stand-alone pointwise components
``FourBoxOceanHeatUptakeBuilder`` (components.pyi:124; four_box_ocean_heat_uptake.rs) and
``OceanSurfacePartialPressureBuilder`` (ocean_carbon_cycle/ocean_surface_partial_pressure.rs)."""
from __future__ import annotations

from . import _lib as L
from .core import Component, ComponentBuilder


class CarbonCycle(Component):
    type_name = "CarbonCycle"
    definitions = [("Emissions|CO2|Anthropogenic", "GtC / yr", "Input"),
                   ("Surface Temperature", "K", "Input"),
                   ("Atmospheric Concentration|CO2", "ppm", "State"),
                   ("Cumulative Emissions|CO2", "Gt C", "State"),
                   ("Cumulative Land Uptake", "Gt C", "State")]
    step_size = 0.1  # SolverOptions default (carbon_cycle.rs:83)

    def with_solver_options(self, step_size: float) -> "CarbonCycle":
        self.step_size = float(step_size)
        return self


class CarbonCycleBuilder(ComponentBuilder):
    component_cls = CarbonCycle
    required = ("tau", "conc_pi", "alpha_temperature")


class CO2ERF(Component):
    type_name = "CO2ERF"
    definitions = [("Atmospheric Concentration|CO2", "ppm", "Input"),
                   ("Effective Radiative Forcing|CO2", "W/m^2", "Output")]


class CO2ERFBuilder(ComponentBuilder):
    component_cls = CO2ERF
    required = ("erf_2xco2", "conc_pi")


class FourBoxOceanHeatUptake(Component):
    type_name = "FourBoxOceanHeatUptake"
    definitions = [("Effective Radiative Forcing|Aggregated", "W/m^2", "Input"),
                   ("Heat Uptake|Ocean", "W/m^2", "Output")]  # FourBox grid

    def param_vector(self):
        return [float(self.parameters[k]) for k in L.FB_PARAM_NAMES]


class FourBoxOceanHeatUptakeBuilder(ComponentBuilder):
    component_cls = FourBoxOceanHeatUptake
    required = L.FB_PARAM_NAMES

    def build(self):
        avg = sum(self._parameters[k] for k in L.FB_PARAM_NAMES) / 4.0
        if not abs(avg - 1.0) < 0.01:  # four_box_ocean_heat_uptake.rs:70-76 assert!
            raise ValueError(f"Regional ratios must average to 1.0 with equal weights (got {avg})")
        return super().build()


class OceanSurfacePartialPressure(Component):
    type_name = "OceanSurfacePartialPressure"
    definitions = [("Sea Surface Temperature", "K", "Input"), ("Dissolved Inorganic Carbon", "micromol / kg", "Input"),
                   ("Ocean Surface Partial Pressure|CO2", "ppm", "Output")]

    def param_vector(self):
        return [float(self.parameters[k]) for k in L.SP_PARAM_NAMES]


class OceanSurfacePartialPressureBuilder(ComponentBuilder):
    component_cls = OceanSurfacePartialPressure

    @classmethod
    def from_parameters(cls, parameters):
        p = {}
        for k in ("ospp_preindustrial", "sensitivity_ospp_to_temperature", "sea_surface_temperature_preindustrial"):
            if k not in parameters:  # no serde default upstream: every field is required
                raise ValueError(f"missing field `{k}`")
            p[k] = float(parameters[k])
        for k in ("delta_ospp_offsets", "delta_ospp_coefficients"):
            if k not in parameters:
                raise ValueError(f"missing field `{k}`")
            if len(parameters[k]) != 5:
                raise ValueError(f"invalid length {len(parameters[k])}, expected an array of length 5")
            for j, x in enumerate(parameters[k]):
                p[f"{k}_{j}"] = float(x)
        extra = set(parameters) - {"ospp_preindustrial", "sensitivity_ospp_to_temperature",
                                   "sea_surface_temperature_preindustrial", "delta_ospp_offsets", "delta_ospp_coefficients"}
        if extra:
            raise ValueError(f"unknown field `{sorted(extra)[0]}`")
        return cls(p)
