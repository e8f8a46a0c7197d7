"""``ClimateUDEBBuilder``, ``GhgForcingBuilder``, ``OzoneForcingBuilder``, ``AerosolDirectBuilder``
``AerosolIndirectBuilder``, ``CH4ChemistryBuilder``, ``N2OChemistryBuilder``, ``CO2BudgetBuilder``,
``TerrestrialCarbonBuilder``, ``OceanCarbonBuilder`` and ``HalocarbonChemistryBuilder`` -- mirror of ``rscm.magicc`` for the climate core, the forcing
components, the CH4 / N2O chemistry, the CO2 budget and the terrestrial carbon pools
(python/rscm/_lib/magicc.pyi; crates/rscm-magicc/src/climate/udeb/mod.rs,
crates/rscm-magicc/src/forcing/{ghg,ozone,aerosol_direct,aerosol_indirect}.rs,
crates/rscm-magicc/src/chemistry/{ch4,n2o}.rs, crates/rscm-magicc/src/carbon/{budget,terrestrial,ocean}.rs
and their
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


def _flat_parameters(names, defaults, parameters, arrays=()):
    """serde(default) semantics over a flat name list; ``arrays`` are [f64; 4] fields stored as
    name_0..name_3; booleans become 0/1."""
    p = dict(zip(names, defaults))
    for k, v in parameters.items():
        if k in arrays:
            n_expected = sum(1 for name in names if name.startswith(k + "_") and name[len(k) + 1:].isdigit())
            if len(v) != n_expected:
                raise ValueError(f"invalid length {len(v)}, expected an array of length {n_expected}")
            for j, x in enumerate(v):
                p[f"{k}_{j}"] = float(x)
        elif k in p:
            p[k] = float(v)
        else:  # serde: unknown field
            raise ValueError(f"unknown field `{k}`")
    return p


class OzoneForcing(Component):
    type_name = "OzoneForcing"
    definitions = ([(n, u, "Input") for n, u in zip(L.OZ_INPUTS, ("ppt", "ppb", "Mt N/yr", "Mt CO/yr", "Mt NMVOC/yr", "K"))]
                   + [(n, "W/m^2", "Output") for n, v in L.OZ_VARS.items() if v > 0])

    def param_vector(self):
        return [float(self.parameters[k]) for k in L.OZ_PARAM_NAMES]


class OzoneForcingBuilder(ComponentBuilder):
    component_cls = OzoneForcing

    @classmethod
    def from_parameters(cls, parameters: Dict[str, float]):
        return cls(_flat_parameters(L.OZ_PARAM_NAMES, L.OZ_DEFAULTS, parameters))


class AerosolDirect(Component):
    type_name = "AerosolDirect"
    definitions = ([(n, u, "Input") for n, u in zip(L.AD_INPUTS, ("Mt S/yr", "Mt BC/yr", "Mt OC/yr", "Mt N/yr"))]
                   + [("Effective Radiative Forcing|Aerosol|Direct", "W/m^2", "Output")])  # FourBox grid

    def param_vector(self):
        return [float(self.parameters[k]) for k in L.AD_PARAM_NAMES]


class AerosolDirectBuilder(ComponentBuilder):
    component_cls = AerosolDirect

    @classmethod
    def from_parameters(cls, parameters: Dict[str, float]):
        return cls(_flat_parameters(L.AD_PARAM_NAMES, L.AD_DEFAULTS, parameters,
                                    arrays=("sox_regional", "bc_regional", "oc_regional", "nitrate_regional")))


class AerosolIndirect(Component):
    type_name = "AerosolIndirect"
    definitions = ([(n, u, "Input") for n, u in zip(L.AI_INPUTS, ("Mt S/yr", "Mt OC/yr"))]
                   + [("Effective Radiative Forcing|Aerosol|Indirect", "W/m^2", "Output")])

    def param_vector(self):
        return [float(self.parameters[k]) for k in L.AI_PARAM_NAMES]


class AerosolIndirectBuilder(ComponentBuilder):
    component_cls = AerosolIndirect

    @classmethod
    def from_parameters(cls, parameters: Dict[str, float]):
        return cls(_flat_parameters(L.AI_PARAM_NAMES, L.AI_DEFAULTS, parameters))


class CH4Chemistry(Component):
    type_name = "CH4Chemistry"
    definitions = ([(n, u, "Input") for n, u in zip(L.CH4_INPUTS, ("Mt CH4/yr", "K", "Mt N/yr", "Mt CO/yr", "Mt NMVOC/yr"))]
                   + [("Atmospheric Concentration|CH4", "ppb", "State"), ("Lifetime|CH4", "yr", "Output")])

    def param_vector(self):
        return [float(self.parameters[k]) for k in L.CH4_PARAM_NAMES]


class CH4ChemistryBuilder(ComponentBuilder):
    component_cls = CH4Chemistry

    @classmethod
    def from_parameters(cls, parameters: Dict[str, float]):
        return cls(_flat_parameters(L.CH4_PARAM_NAMES, L.CH4_DEFAULTS, parameters))


class N2OChemistry(Component):
    type_name = "N2OChemistry"
    definitions = [("Emissions|N2O", "Mt N/yr", "Input"), ("Atmospheric Concentration|N2O", "ppb", "State"),
                   ("Lifetime|N2O", "yr", "Output")]

    def param_vector(self):
        return [float(self.parameters[k]) for k in L.N2O_PARAM_NAMES]


class N2OChemistryBuilder(ComponentBuilder):
    component_cls = N2OChemistry

    @classmethod
    def from_parameters(cls, parameters: Dict[str, float]):
        p = _flat_parameters(L.N2O_PARAM_NAMES, L.N2O_DEFAULTS, parameters)
        if p["strat_delay"] < 0 or p["strat_delay"] != int(p["strat_delay"]):  # serde: usize
            raise ValueError("invalid type: strat_delay must be a non-negative integer")
        return cls(p)


class CO2Budget(Component):
    type_name = "CO2Budget"
    definitions = ([(n, "GtC/yr", "Input") for n in L.CB_INPUTS]
                   + [("Atmospheric Concentration|CO2", "ppm", "State"), ("Emissions|CO2|Net", "GtC/yr", "Output"),
                      ("Airborne Fraction|CO2", "1", "Output")])

    def param_vector(self):
        return [float(self.parameters[k]) for k in L.CB_PARAM_NAMES]


class CO2BudgetBuilder(ComponentBuilder):
    component_cls = CO2Budget

    @classmethod
    def from_parameters(cls, parameters: Dict[str, float]):
        return cls(_flat_parameters(L.CB_PARAM_NAMES, L.CB_DEFAULTS, parameters))


class TerrestrialCarbon(Component):
    type_name = "TerrestrialCarbon"
    definitions = ([(n, u, "Input") for n, u in zip(L.TC_INPUTS, ("ppm", "K", "GtC/yr"))]
                   + [(n, "GtC", "State") for n, v in L.TC_VARS.items() if 1 <= v <= 4]
                   + [("Carbon Flux|Terrestrial", "GtC/yr", "Output")])

    def param_vector(self):
        return [float(self.parameters[k]) for k in L.TC_PARAM_NAMES]


class TerrestrialCarbonBuilder(ComponentBuilder):
    component_cls = TerrestrialCarbon

    @classmethod
    def from_parameters(cls, parameters: Dict[str, float]):
        return cls(_flat_parameters(L.TC_PARAM_NAMES, L.TC_DEFAULTS, parameters))


class OceanCarbon(Component):
    type_name = "OceanCarbon"
    definitions = [("Atmospheric Concentration|CO2", "ppm", "Input"), ("Sea Surface Temperature", "K", "Input"),
                   ("Ocean Surface pCO2", "ppm", "State"), ("Cumulative Ocean Uptake", "GtC", "State"),
                   ("Carbon Flux|Ocean", "GtC/yr", "Output")]

    def param_vector(self):
        return [float(self.parameters[k]) for k in L.OC_PARAM_NAMES]


class OceanCarbonBuilder(ComponentBuilder):
    """``model`` picks the preset the unspecified fields default to and the impulse-response
    coefficient sets (the reference's ``OceanCarbonParameters::{gfdl_3d, bern_2d, hilda}``; its
    serde default is 3D-GFDL).  Custom ``irf_early`` / ``irf_late`` forms are not supported on the
    device."""
    component_cls = OceanCarbon

    @classmethod
    def from_parameters(cls, parameters: Dict[str, float]):
        parameters = dict(parameters)
        model = parameters.pop("model", "3D-GFDL")
        if model not in L.OC_PRESETS:  # serde: unknown variant
            raise ValueError(f"unknown variant `{model}`, expected one of `3D-GFDL`, `2D-BERN`, `HILDA`")
        for k in ("irf_early", "irf_late"):
            if k in parameters:
                raise NotImplementedError(f"custom {k} forms are not supported by the device kernel; choose a model preset")
        return cls(_flat_parameters(L.OC_PARAM_NAMES, L.OC_PRESETS[model], parameters,
                                    arrays=("delta_ospp_offsets", "delta_ospp_coefficients")))


class HalocarbonChemistry(Component):
    type_name = "HalocarbonChemistry"
    definitions = ([(n, "kt/yr", "Input") for n in L.HC_INPUTS]
                   + [(f"Atmospheric Concentration|{s}", "ppt", "State") for s in L.HC_SPECIES]
                   + [("Forcing|Halocarbons", "W/m^2", "Output"), ("Forcing|F-gases", "W/m^2", "Output"),
                      ("Forcing|Montreal Gases", "W/m^2", "Output"), ("EESC", "ppt", "Output")])

    def param_vector(self):
        return [float(self.parameters[k]) for k in L.HC_PARAM_NAMES]


class HalocarbonChemistryBuilder(ComponentBuilder):
    """``fgases`` / ``montreal_gases`` take the reference's list-of-species form
    (``[{"name": "CF4", "lifetime": ..., ...}, ...]``); the device kernel is laid out for the 23 + 18
    species of ``HalocarbonParameters::default()`` in their default order, so the lists may change
    species properties but not the species set."""
    component_cls = HalocarbonChemistry

    @classmethod
    def from_parameters(cls, parameters: Dict[str, object]):
        p = dict(zip(L.HC_PARAM_NAMES, L.HC_DEFAULTS))
        for k, v in parameters.items():
            if k in ("fgases", "montreal_gases"):
                expected = [s[0] for s in (L.HC_FGASES if k == "fgases" else L.HC_MONTREAL)]
                if [sp.get("name") for sp in v] != expected:
                    raise NotImplementedError(f"{k}: the device kernel supports the default species set {expected}")
                for sp in v:
                    for f, x in sp.items():
                        if f == "name":
                            continue
                        if f not in L.HC_FIELDS:  # serde: unknown field
                            raise ValueError(f"unknown field `{f}`")
                        p[f"{sp['name']}.{f}"] = float(x)
            elif k in L.HC_GLOBALS:
                p[k] = float(v)
            else:
                raise ValueError(f"unknown field `{k}`")
        return cls(p)
