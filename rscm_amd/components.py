"""``CarbonCycleBuilder`` / ``CO2ERFBuilder`` -- mirror of ``rscm.components`` for the coupled
chain (python/rscm/_lib/components.pyi:23-120; crates/rscm-components/src/components/
carbon_cycle.rs:24-94, co2_erf.rs:17-52)."""
from __future__ import annotations

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
