"""ctypes binding of ``librscm_gpu.so`` -- the C-ABI of include/rscm_gpu.h.

There is no CPU fallback: if the HIP library is missing or fails to load this module raises
``RscmGpuUnavailable`` and every product entry point fails loudly.  Build it with
``python -c "import __graft_entry__ as g; g.build()"`` or ``make -C rscm_amd/csrc``.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "librscm_gpu.so")

OK, ERR_INVALID, ERR_STATE, ERR_TIME_AXIS, ERR_DEVICE, ERR_NOMEM = range(6)
KIND_TWO_LAYER, KIND_COUPLED, KIND_UDEB, KIND_GHG_FORCING = 0, 1, 2, 3
KIND_OZONE_FORCING, KIND_AEROSOL_DIRECT, KIND_AEROSOL_INDIRECT = 4, 5, 6
KIND_CH4_CHEMISTRY, KIND_N2O_CHEMISTRY = 7, 8
KIND_CO2_BUDGET, KIND_TERRESTRIAL_CARBON = 9, 10
KIND_OCEAN_CARBON = 11
KIND_HALOCARBON = 12
KIND_FOURBOX_OHU, KIND_OSPP = 13, 14
KIND_CARBON_CYCLE, KIND_CO2_ERF, KIND_AGGREGATE = 15, 16, 17
SRC_EXOGENOUS, SRC_UPSTREAM = 0, 1
COMP_TWO_LAYER, COMP_CARBON_CYCLE = 0, 1
MODE_EXACT, MODE_FAST = 0, 1
FLAG_NO_SERIES = 1
FLAG_WINDOWED = 2

TL_VARS = {"Effective Radiative Forcing": 0, "Surface Temperature": 1, "Deep Ocean Temperature": 2}
CP_VARS = {"Emissions|CO2|Anthropogenic": 0, "Surface Temperature": 1, "Deep Ocean Temperature": 2,
           "Atmospheric Concentration|CO2": 3, "Cumulative Land Uptake": 4,
           "Cumulative Emissions|CO2": 5, "Effective Radiative Forcing|CO2": 6,
           "Effective Radiative Forcing": 7}

UD_VARS = {"Effective Radiative Forcing": 0, "Surface Temperature|NorthernOcean": 1,
           "Surface Temperature|NorthernLand": 2, "Surface Temperature|SouthernOcean": 3,
           "Surface Temperature|SouthernLand": 4, "Heat Uptake": 5, "Ocean Heat Content": 6,
           "Sea Surface Temperature": 7}
UD_PARAM_NAMES = (
    "n_layers", "mixed_layer_depth", "layer_thickness", "kappa", "kappa_min", "kappa_dkdt",
    "w_initial", "w_variable_fraction", "w_threshold_temp_nh", "w_threshold_temp_sh", "ecs",
    "rf_2xco2", "rlo", "feedback_q_sensitivity", "feedback_cumt_sensitivity",
    "feedback_cumt_period", "k_lo", "k_ns", "amplify_ocean_to_land", "nh_land_fraction",
    "sh_land_fraction", "depth_dependent_area", "temp_adjust_alpha", "temp_adjust_gamma",
    "polar_sinking_ratio", "land_heat_capacity_enabled", "k_lg", "land_hc_eff_thickness",
    "rf_regions_co2_0", "rf_regions_co2_1", "rf_regions_co2_2", "rf_regions_co2_3",
    "efficacy_apply", "prescribed_efficacy_co2", "ocean_temp_profile", "steps_per_year",
    "max_temperature")
# ClimateUDEBParameters::default() (crates/rscm-magicc/src/parameters/climate_udeb.rs)
UD_DEFAULTS = (50, 60.0, 100.0, 0.75, 0.1, -0.191, 3.5, 0.7, 8.0, 8.0, 3.0, 3.71, 1.317, 7.84e-9, 0.08,
               300.0, 1.44, 0.31, 1.02, 0.42, 0.21, 1.0, 1.04, -0.002, 0.2, 1.0, 0.1, 300.0,
               1.4089, 1.37045, 1.43333, 1.33257, 0.0, 1.0, 2.0, 12.0, 25.0)

# GhgForcing (crates/rscm-magicc/src/forcing/ghg.rs:69-83): variable 0 is the block of the three
# concentration rows GH_INPUTS, [S][3][T]
GH_INPUTS = ("Atmospheric Concentration|CO2", "Atmospheric Concentration|CH4", "Atmospheric Concentration|N2O")
GH_VARS = {"Atmospheric Concentration": 0, "Effective Radiative Forcing|CO2": 1,
           "Effective Radiative Forcing|CH4": 2, "Effective Radiative Forcing|N2O": 3}
GH_PARAM_NAMES = ("method", "co2_pi", "ch4_pi", "n2o_pi", "delq2xco2", "ch4_radeff", "n2o_radeff",
                  "olbl_co2_a1", "olbl_co2_b1", "olbl_co2_c1", "olbl_co2_d1",
                  "olbl_ch4_a3", "olbl_ch4_b3", "olbl_ch4_d3",
                  "olbl_n2o_a2", "olbl_n2o_b2", "olbl_n2o_c2", "olbl_n2o_d2",
                  "adjust_co2", "adjust_ch4", "adjust_n2o")
GH_METHODS = {"Ipcctar": 0.0, "Olbl": 1.0}
# GhgForcingParameters::default() (crates/rscm-magicc/src/parameters/ghg_forcing.rs)
GH_DEFAULTS = (1.0, 278.0, 722.0, 270.0, 3.71, 0.036, 0.12, -2.4785e-7, 7.5906e-4, -2.1492e-3, 5.2,
               -8.9603e-5, -1.2462e-4, 0.045, -3.4197e-4, 2.5455e-4, -2.4357e-4, 0.14, 1.05, 0.86, 1.0)

# OzoneForcing / AerosolDirect / AerosolIndirect (crates/rscm-magicc/src/forcing/*.rs): variable 0
# is the block of input rows, [S][n_inputs][T], in the order of the #[inputs(...)] declaration
OZ_INPUTS = ("EESC", "Atmospheric Concentration|CH4", "Emissions|NOx", "Emissions|CO", "Emissions|NMVOC",
             "Surface Temperature")
OZ_VARS = {"Ozone inputs": 0, "Effective Radiative Forcing|O3|Stratospheric": 1,
           "Effective Radiative Forcing|O3|Tropospheric": 2,
           "Effective Radiative Forcing|O3|Temperature Feedback": 3}
OZ_PARAM_NAMES = ("eesc_reference", "strat_o3_scale", "strat_cl_exponent", "trop_radeff", "trop_oz_ch4",
                  "trop_oz_nox", "trop_oz_co", "trop_oz_voc", "ch4_pi", "nox_pi", "co_pi", "nmvoc_pi",
                  "temp_feedback_scale")
OZ_DEFAULTS = (1420.0, -0.0043, 1.7, 0.032, 5.7, 0.168, 0.00396, 0.01008, 700.0, 0.0, 0.0, 0.0, -0.037)
FOURBOX_REGIONS = ("NorthernOcean", "NorthernLand", "SouthernOcean", "SouthernLand")
AD_INPUTS = ("Emissions|SOx", "Emissions|BC", "Emissions|OC", "Emissions|NOx")
AD_VARS = {"Aerosol direct inputs": 0,
           **{f"Effective Radiative Forcing|Aerosol|Direct|{r}": k + 1 for k, r in enumerate(FOURBOX_REGIONS)}}
AD_PARAM_NAMES = (("sox_coefficient", "bc_coefficient", "oc_coefficient", "nitrate_coefficient")
                  + tuple(f"{s}_regional_{i}" for s in ("sox", "bc", "oc", "nitrate") for i in range(4))
                  + ("sox_pi", "bc_pi", "oc_pi", "nox_pi", "harmonize", "harmonize_year", "harmonize_target"))
AD_DEFAULTS = (-0.0035, 0.0077, -0.002, -0.001, 0.15, 0.55, 0.10, 0.20, 0.15, 0.50, 0.15, 0.20,
               0.15, 0.45, 0.15, 0.25, 0.15, 0.50, 0.15, 0.20, 1.0, 2.5, 10.0, 10.0, 0.0, 2019.0, -0.22)
AI_INPUTS = ("Emissions|SOx", "Emissions|OC")
AI_VARS = {"Aerosol indirect inputs": 0, "Effective Radiative Forcing|Aerosol|Indirect": 1}
AI_PARAM_NAMES = ("cloud_albedo_coefficient", "reference_burden", "sox_weight", "oc_weight", "sox_pi", "oc_pi",
                  "harmonize", "harmonize_year", "harmonize_target")
AI_DEFAULTS = (-1.0, 50.0, 1.0, 0.3, 1.0, 10.0, 0.0, 2019.0, -0.89)

# CH4Chemistry / N2OChemistry (crates/rscm-magicc/src/chemistry/{ch4,n2o}.rs): variable 1 is the
# concentration state, variable 2 the lifetime output
CH4_INPUTS = ("Emissions|CH4", "Surface Temperature", "Emissions|NOx", "Emissions|CO", "Emissions|NMVOC")
CH4_VARS = {"CH4 chemistry inputs": 0, "Atmospheric Concentration|CH4": 1, "Lifetime|CH4": 2}
CH4_PARAM_NAMES = ("ch4_pi", "natural_emissions", "tau_oh", "tau_soil", "tau_strat", "tau_trop_cl",
                   "ch4_self_feedback", "oh_sensitivity_scale", "oh_nox_sensitivity", "oh_co_sensitivity",
                   "oh_nmvoc_sensitivity", "temp_sensitivity", "include_temp_feedback",
                   "include_emissions_feedback", "ppb_to_tg", "nox_reference", "co_reference", "nmvoc_reference")
CH4_DEFAULTS = (722.0, 209.0, 9.3, 150.0, 120.0, 200.0, -0.32, 0.72, 0.0042, -0.000105, -0.000315, 0.0316,
                1.0, 1.0, 2.75, 0.0, 0.0, 0.0)
N2O_INPUTS = ("Emissions|N2O",)
N2O_VARS = {"N2O chemistry inputs": 0, "Atmospheric Concentration|N2O": 1, "Lifetime|N2O": 2}
N2O_PARAM_NAMES = ("n2o_pi", "natural_emissions", "tau_n2o", "lifetime_feedback", "strat_delay", "ppb_to_tg")
N2O_DEFAULTS = (270.0, 11.0, 139.275, -0.04, 1.0, 4.79)

# CO2Budget / TerrestrialCarbon (crates/rscm-magicc/src/carbon/{budget,terrestrial}.rs)
CB_INPUTS = ("Emissions|CO2|Fossil", "Emissions|CO2|Land Use", "Carbon Flux|Terrestrial", "Carbon Flux|Ocean")
CB_VARS = {"CO2 budget inputs": 0, "Atmospheric Concentration|CO2": 1, "Emissions|CO2|Net": 2,
           "Airborne Fraction|CO2": 3}
CB_PARAM_NAMES = ("gtc_per_ppm", "co2_pi")
CB_DEFAULTS = (2.123, 278.0)
TC_INPUTS = ("Atmospheric Concentration|CO2", "Surface Temperature", "Emissions|CO2|Land Use")
TC_VARS = {"Terrestrial carbon inputs": 0, "Carbon Pool|Plant": 1, "Carbon Pool|Detritus": 2,
           "Carbon Pool|Soil": 3, "Carbon Pool|Humus": 4, "Carbon Flux|Terrestrial": 5}
TC_PARAM_NAMES = ("npp_pi", "co2_pi", "beta", "npp_temp_sensitivity", "resp_temp_sensitivity",
                  "detritus_temp_sensitivity", "soil_temp_sensitivity", "humus_temp_sensitivity",
                  "plant_pool_pi", "detritus_pool_pi", "soil_pool_pi", "humus_pool_pi", "respiration_pi",
                  "frac_npp_to_plant", "frac_npp_to_detritus", "frac_plant_to_detritus",
                  "frac_detritus_to_soil", "frac_soil_to_humus", "enable_fertilization", "enable_temp_feedback")
TC_DEFAULTS = (66.27, 278.0, 0.6486, 0.0107, 0.0685, 0.1358, 0.1541, 0.05, 884.86, 92.77, 1681.53, 836.0,
               12.26, 0.4483, 0.3998, 0.9989, 0.3, 0.1, 1.0, 1.0)

# OceanCarbon (crates/rscm-magicc/src/carbon/ocean.rs, parameters/ocean_carbon.rs presets)
OC_INPUTS = ("Atmospheric Concentration|CO2", "Sea Surface Temperature")
OC_VARS = {"Ocean carbon inputs": 0, "Ocean Surface pCO2": 1, "Cumulative Ocean Uptake": 2, "Carbon Flux|Ocean": 3}
OC_MODELS = {"3D-GFDL": 0.0, "2D-BERN": 1.0, "HILDA": 2.0}
OC_PARAM_NAMES = (("model", "co2_pi", "pco2_pi", "gas_exchange_scale", "gas_exchange_tau", "temp_sensitivity",
                   "irf_scale", "mixed_layer_depth", "ocean_surface_area", "sst_pi", "steps_per_year",
                   "max_history_months", "irf_switch_time")
                  + tuple(f"delta_ospp_offsets_{i}" for i in range(5))
                  + tuple(f"delta_ospp_coefficients_{i}" for i in range(5)) + ("enable_temp_feedback",))
_OC_OSPP = (1.5568, 7.4706, 1.2748, 2.4491, 1.5468, -0.013993, -0.20207, -0.12015, -0.12639, -0.15326, 1.0)
# gfdl_3d(), bern_2d(), hilda() (parameters/ocean_carbon.rs:88-196)
OC_PRESETS = {
    "3D-GFDL": (0.0, 278.0, 278.0, 1.833492, 7.66, 0.03717879, 0.9492864, 50.9, 3.55e14, 17.7, 12.0, 6000.0, 1.0) + _OC_OSPP,
    "2D-BERN": (1.0, 278.0, 278.0, 1.833492, 7.46, 0.03717879, 0.9492864, 50.0, 3.5375e14, 18.2997, 12.0, 6000.0, 9.9) + _OC_OSPP,
    "HILDA": (2.0, 278.0, 278.0, 1.833492, 9.06, 0.03717879, 0.9492864, 75.0, 3.62e14, 18.1716, 12.0, 6000.0, 2.0) + _OC_OSPP,
}

# HalocarbonChemistry (crates/rscm-magicc/src/chemistry/halocarbon.rs, parameters/halocarbon.rs:95-160):
# name, lifetime, radiative_efficiency, concentration_pi, molecular_weight, n_cl, n_br, fractional_release
HC_FGASES = (
    ("CF4", 50000.0, 0.09, 0.0, 88.0, 0, 0, 0.0), ("C2F6", 10000.0, 0.25, 0.0, 138.0, 0, 0, 0.0),
    ("C3F8", 2600.0, 0.28, 0.0, 188.0, 0, 0, 0.0), ("C4F10", 2600.0, 0.36, 0.0, 238.0, 0, 0, 0.0),
    ("C5F12", 4100.0, 0.41, 0.0, 288.0, 0, 0, 0.0), ("C6F14", 3100.0, 0.44, 0.0, 338.0, 0, 0, 0.0),
    ("C7F16", 3000.0, 0.50, 0.0, 388.0, 0, 0, 0.0), ("C8F18", 3000.0, 0.55, 0.0, 438.0, 0, 0, 0.0),
    ("c-C4F8", 3200.0, 0.32, 0.0, 200.0, 0, 0, 0.0), ("HFC-23", 228.0, 0.18, 0.0, 70.0, 0, 0, 0.0),
    ("HFC-32", 5.4, 0.11, 0.0, 52.0, 0, 0, 0.0), ("HFC-43-10mee", 17.0, 0.359, 0.0, 252.0, 0, 0, 0.0),
    ("HFC-125", 31.0, 0.23, 0.0, 120.0, 0, 0, 0.0), ("HFC-134a", 14.0, 0.16, 0.0, 102.0, 0, 0, 0.0),
    ("HFC-143a", 51.0, 0.16, 0.0, 84.0, 0, 0, 0.0), ("HFC-152a", 1.6, 0.10, 0.0, 66.0, 0, 0, 0.0),
    ("HFC-227ea", 36.0, 0.26, 0.0, 170.0, 0, 0, 0.0), ("HFC-236fa", 213.0, 0.24, 0.0, 152.0, 0, 0, 0.0),
    ("HFC-245fa", 7.9, 0.24, 0.0, 134.0, 0, 0, 0.0), ("HFC-365mfc", 8.9, 0.22, 0.0, 148.0, 0, 0, 0.0),
    ("NF3", 569.0, 0.20, 0.0, 71.0, 0, 0, 0.0), ("SF6", 850.0, 0.57, 0.0, 146.0, 0, 0, 0.0),
    ("SO2F2", 36.0, 0.20, 0.0, 102.0, 0, 0, 0.0))
HC_MONTREAL = (
    ("CFC-11", 52.0, 0.295, 0.0, 137.4, 3, 0, 0.47), ("CFC-12", 102.0, 0.364, 0.0, 120.9, 2, 0, 0.23),
    ("CFC-113", 93.0, 0.30, 0.0, 187.4, 3, 0, 0.29), ("CFC-114", 189.0, 0.31, 0.0, 170.9, 2, 0, 0.12),
    ("CFC-115", 540.0, 0.20, 0.0, 154.5, 1, 0, 0.04), ("HCFC-22", 11.9, 0.21, 0.0, 86.5, 1, 0, 0.13),
    ("HCFC-141b", 9.4, 0.16, 0.0, 116.9, 2, 0, 0.34), ("HCFC-142b", 18.0, 0.19, 0.0, 100.5, 1, 0, 0.17),
    ("CH3CCl3", 5.0, 0.07, 0.0, 133.4, 3, 0, 0.67), ("CCl4", 32.0, 0.174, 0.0, 153.8, 4, 0, 0.56),
    ("CH3Cl", 0.9, 0.004, 500.0, 50.5, 1, 0, 0.44), ("CH2Cl2", 0.5, 0.028, 0.0, 84.9, 2, 0, 0.0),
    ("CHCl3", 0.5, 0.07, 0.0, 119.4, 3, 0, 0.0), ("CH3Br", 0.8, 0.004, 5.0, 94.9, 0, 1, 0.60),
    ("Halon-1211", 16.0, 0.29, 0.0, 165.4, 1, 1, 0.62), ("Halon-1301", 72.0, 0.30, 0.0, 148.9, 0, 1, 0.28),
    ("Halon-2402", 28.0, 0.31, 0.0, 259.8, 0, 2, 0.65), ("Halon-1202", 2.5, 0.27, 0.0, 209.8, 0, 2, 0.62))
HC_SPECIES = tuple(s[0] for s in HC_FGASES + HC_MONTREAL)
HC_FIELDS = ("lifetime", "radiative_efficiency", "concentration_pi", "molecular_weight", "n_cl", "n_br",
             "fractional_release")
HC_GLOBALS = ("br_multiplier", "cfc11_release_normalisation", "eesc_delay", "air_molar_mass",
              "atmospheric_mass_tg", "mixing_box_fraction")
HC_GLOBAL_DEFAULTS = (60.0, 0.47, 3.0, 28.97, 5.133e9, 0.949)
HC_PARAM_NAMES = HC_GLOBALS + tuple(f"{s}.{f}" for s in HC_SPECIES for f in HC_FIELDS)
HC_DEFAULTS = HC_GLOBAL_DEFAULTS + tuple(float(x) for s in HC_FGASES + HC_MONTREAL for x in s[1:])
HC_INPUTS = tuple(f"Emissions|{s}" for s in HC_SPECIES)
HC_VARS = {"Halocarbon emissions": 0, **{f"Atmospheric Concentration|{s}": k + 1 for k, s in enumerate(HC_SPECIES)},
           "Forcing|Halocarbons": 42, "Forcing|F-gases": 43, "Forcing|Montreal Gases": 44, "EESC": 45}

# rscm-components: FourBoxOceanHeatUptake, OceanSurfacePartialPressure
FB_INPUTS = ("Effective Radiative Forcing|Aggregated",)
FB_VARS = {"Heat uptake input": 0, **{f"Heat Uptake|Ocean|{r}": k + 1 for k, r in enumerate(FOURBOX_REGIONS)}}
FB_PARAM_NAMES = ("northern_ocean_ratio", "northern_land_ratio", "southern_ocean_ratio", "southern_land_ratio")
FB_DEFAULTS = (1.2, 0.6, 1.6, 0.6)
SP_INPUTS = ("Sea Surface Temperature", "Dissolved Inorganic Carbon")
SP_VARS = {"OSPP inputs": 0, "Ocean Surface Partial Pressure|CO2": 1}
SP_PARAM_NAMES = (("ospp_preindustrial", "sensitivity_ospp_to_temperature", "sea_surface_temperature_preindustrial")
                  + tuple(f"delta_ospp_offsets_{i}" for i in range(5))
                  + tuple(f"delta_ospp_coefficients_{i}" for i in range(5)))

CC_INPUTS = ("Emissions|CO2|Anthropogenic", "Surface Temperature")
CC_VARS = {"CarbonCycle inputs": 0, "Atmospheric Concentration|CO2": 1, "Cumulative Land Uptake": 2,
           "Cumulative Emissions|CO2": 3}
CC_PARAM_NAMES = ("tau", "conc_pi", "alpha_temperature")
CE_INPUTS = ("Atmospheric Concentration|CO2",)
CE_VARS = {"CO2ERF input": 0, "Effective Radiative Forcing|CO2": 1}
CE_PARAM_NAMES = ("erf_2xco2", "conc_pi")
AG_NINPUTS = 8
AG_INPUTS = tuple(f"contributor_{k}" for k in range(AG_NINPUTS))
AG_VARS = {"contributors": 0, "aggregate": 1}
AG_PARAM_NAMES = ("operation",) + tuple(f"weight_{k}" for k in range(AG_NINPUTS))
AG_OPERATIONS = {"Sum": 0.0, "Mean": 1.0, "Weighted": 2.0, "Count": 3.0, "CountCarry": 4.0, "Quotient": 5.0}

# per kind: (variable ids, parameter names, input rows of variable 0 or None for a single series)
KIND_TABLE = {
    KIND_TWO_LAYER: (TL_VARS, 6, None), KIND_COUPLED: (CP_VARS, 10, None), KIND_UDEB: (UD_VARS, 37, None),
    KIND_GHG_FORCING: (GH_VARS, 21, GH_INPUTS), KIND_OZONE_FORCING: (OZ_VARS, 13, OZ_INPUTS),
    KIND_AEROSOL_DIRECT: (AD_VARS, 27, AD_INPUTS), KIND_AEROSOL_INDIRECT: (AI_VARS, 9, AI_INPUTS),
    KIND_CH4_CHEMISTRY: (CH4_VARS, 18, CH4_INPUTS), KIND_N2O_CHEMISTRY: (N2O_VARS, 6, N2O_INPUTS),
    KIND_CO2_BUDGET: (CB_VARS, 2, CB_INPUTS), KIND_TERRESTRIAL_CARBON: (TC_VARS, 20, TC_INPUTS),
    KIND_OCEAN_CARBON: (OC_VARS, 24, OC_INPUTS), KIND_HALOCARBON: (HC_VARS, len(HC_PARAM_NAMES), HC_INPUTS),
    KIND_FOURBOX_OHU: (FB_VARS, 4, FB_INPUTS), KIND_OSPP: (SP_VARS, 13, SP_INPUTS),
    KIND_CARBON_CYCLE: (CC_VARS, 3, CC_INPUTS), KIND_CO2_ERF: (CE_VARS, 2, CE_INPUTS),
    KIND_AGGREGATE: (AG_VARS, 9, AG_INPUTS)}
# FourBox variables stored as four scalar series: kind -> (name, first variable id)
FOURBOX_VARS = {KIND_UDEB: ("Surface Temperature", 1),
                KIND_AEROSOL_DIRECT: ("Effective Radiative Forcing|Aerosol|Direct", 1),
                KIND_FOURBOX_OHU: ("Heat Uptake|Ocean", 1)}

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)
_bp = C.POINTER(C.c_uint8)
_h = C.c_void_p

# name -> (restype, argtypes); must list every symbol include/rscm_gpu.h declares
SIGNATURES = {
    "rscm_gpu_abi_version": (C.c_int, []),
    "rscm_gpu_abi_minor": (C.c_int, []),
    "rscm_gpu_last_error": (C.c_char_p, []),
    "rscm_gpu_device_count": (C.c_int, [_ip]),
    "rscm_gpu_mem_info": (C.c_int, [C.c_int32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "rscm_ens_create": (C.c_int, [C.c_int32, C.c_int64, C.c_int32, _dp, C.c_int32, C.POINTER(_h)]),
    "rscm_ens_create_ex": (C.c_int, [C.c_int32, C.c_int64, C.c_int32, _dp, C.c_int32, C.c_uint32,
                                     C.POINTER(_h)]),
    "rscm_ens_create_windowed": (C.c_int, [C.c_int32, C.c_int64, C.c_int32, _dp, C.c_int32, C.c_uint32, C.c_int32, C.c_int32,
                                           C.c_int32, _ip, C.POINTER(_h)]),
    "rscm_ens_destroy": (C.c_int, [_h]),
    "rscm_ens_n_params": (C.c_int, [_h, _ip]),
    "rscm_ens_n_vars": (C.c_int, [_h, _ip]),
    "rscm_ens_n_inputs": (C.c_int, [_h, _ip]),
    "rscm_ens_n_members": (C.c_int, [_h, C.POINTER(C.c_int64)]),
    "rscm_ens_n_times": (C.c_int, [_h, _ip]),
    "rscm_ens_set_mode": (C.c_int, [_h, C.c_int32]),
    "rscm_ens_set_step_size": (C.c_int, [_h, C.c_int32, C.c_double]),
    "rscm_ens_set_params": (C.c_int, [_h, _dp]),
    "rscm_ens_set_params_aos": (C.c_int, [_h, _dp]),
    "rscm_ens_set_forcing": (C.c_int, [_h, C.c_int32, C.c_int32, _dp, _ip, C.c_int32]),
    "rscm_ens_link_input": (C.c_int, [_h, C.c_int32, _h, C.c_int32, C.c_int32]),
    "rscm_ens_set_link_order_check": (C.c_int, [_h, C.c_int32]),
    "rscm_ens_unlink_input": (C.c_int, [_h, C.c_int32]),
    "rscm_ens_set_initial": (C.c_int, [_h, C.c_int32, _dp, C.c_int64]),
    "rscm_ens_set_state": (C.c_int, [_h, C.c_int32, C.c_int32, _dp, C.c_int64]),
    "rscm_ens_set_time_index": (C.c_int, [_h, C.c_int32]),
    "rscm_gpu_stream_create": (C.c_int, [C.c_int32, C.POINTER(C.c_void_p)]),
    "rscm_gpu_stream_destroy": (C.c_int, [C.c_int32, C.c_void_p]),
    "rscm_ens_set_stream": (C.c_int, [_h, C.c_void_p]),
    "rscm_ens_run": (C.c_int, [_h, C.c_int32, C.c_int32]),
    "rscm_ens_run_lockstep": (C.c_int, [C.POINTER(C.c_void_p), C.c_int32, C.c_int32, C.c_int32]),
    "rscm_ens_run_async": (C.c_int, [_h, C.c_int32, C.c_int32]),
    # test / A-B hooks (include/rscm_gpu_internal.h).  Their state is PER CALLING THREAD: a switch set from one Python thread does
    # not reach lock-step runs or ClimateUDEB launches issued from another, and the launch counters are the calling thread's own.
    "rscm_gpu_set_lockstep_fusion": (C.c_int, [C.c_int32]),
    "rscm_gpu_set_udeb_variant": (C.c_int, [C.c_int32]),
    "rscm_gpu_fail_chunk_launch": (C.c_int, [C.c_int32]),
    "rscm_gpu_set_run_plan": (C.c_int, [C.c_int32]),
    "rscm_gpu_experiments_build": (C.c_int, []),
    "rscm_gpu_derive_launches": (C.c_int, [C.POINTER(C.c_int64)]),
    "rscm_gpu_lockstep_stats": (C.c_int, [C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "rscm_gpu_lockstep_split_launches": (C.c_int, [C.POINTER(C.c_int64)]),
    "rscm_gpu_lockstep_merged_launches": (C.c_int, [C.POINTER(C.c_int64)]),
    "rscm_gpu_lockstep_last_layout": (C.c_int, [C.POINTER(C.c_int32)]),
    "rscm_gpu_lockstep_own_cut_launches": (C.c_int, [C.POINTER(C.c_int64)]),
    "rscm_ens_sync": (C.c_int, [_h]),
    "rscm_ens_time_index": (C.c_int, [_h, _ip]),
    "rscm_ens_clear_series": (C.c_int, [_h]),
    "rscm_ens_clear_rows_after": (C.c_int, [_h, C.c_int32]),
    "rscm_ens_internal_state_size": (C.c_int, [_h, C.POINTER(C.c_int64)]),
    "rscm_ens_get_internal_state": (C.c_int, [_h, _dp]),
    "rscm_ens_set_internal_state": (C.c_int, [_h, _dp, C.c_int64, C.c_int32]),
    "rscm_ens_rewind": (C.c_int, [_h]),
    "rscm_ens_last_run_ms": (C.c_int, [_h, C.POINTER(C.c_float)]),
    "rscm_ens_last_run_plan": (C.c_int, [_h, _ip, _ip]),
    "rscm_ens_get_series": (C.c_int, [_h, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int64,
                                      C.c_int64, _dp]),
    "rscm_ens_series_devptr": (C.c_int, [_h, C.c_int32, C.POINTER(C.c_void_p)]),
    "rscm_ens_params_devptr": (C.c_int, [_h, C.POINTER(C.c_void_p)]),
    "rscm_ens_status": (C.c_int, [_h, _bp]),
    "rscm_ens_loglik": (C.c_int, [_h, C.c_int32, _ip, _ip, _dp, _dp, C.c_int32, _dp]),
    "rscm_ens_run_loglik": (C.c_int, [_h, C.c_int32, _ip, _ip, _dp, _dp, C.c_int32, _dp]),
    "rscm_ens_loglik_device": (C.c_int, [_h, C.c_int32, _ip, _ip, _dp, _dp, C.c_int32, C.POINTER(C.c_void_p)]),
    "rscm_ens_run_loglik_device": (C.c_int, [_h, C.c_int32, _ip, _ip, _dp, _dp, C.c_int32, C.POINTER(C.c_void_p)]),
    "rscm_ens_status_devptr": (C.c_int, [_h, C.POINTER(C.c_void_p)]),
    "rscm_ens_quantile_series": (C.c_int, [_h, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _dp, _dp, _dp]),
    "rscm_ens_summary_series": (C.c_int, [_h, C.c_int32, C.c_int32, C.c_int32, _dp]),
    "rscm_sampler_create": (C.c_int, [_h, C.c_int32, C.c_int32, _ip, _dp, _ip, _dp, _dp, _dp, _dp, C.c_int32, _ip, _ip,
                                      _dp, _dp, C.c_int32, C.c_double, C.c_uint64, C.POINTER(_h)]),
    "rscm_sampler_create_sharded": (C.c_int, [_h, C.c_int32, C.c_int32, _ip, _dp, _ip, _dp, _dp, _dp, _dp, C.c_int32, _ip, _ip,
                                              _dp, _dp, C.c_int32, C.c_double, C.c_uint64, C.c_int32, C.c_int32, C.POINTER(_h)]),
    "rscm_sampler_create_graph": (C.c_int, [C.POINTER(C.c_void_p), C.c_int32, C.c_int32, C.c_int32, C.c_int32, _ip, _ip, _ip, _dp, _dp, _dp, _dp,
                                            C.c_int32, _ip, _ip, _ip, _dp, _dp, C.c_int32, C.c_double, C.c_uint64, C.c_int32, C.c_int32,
                                            C.POINTER(_h)]),
    "rscm_sampler_begin_iteration": (C.c_int, [_h]),
    "rscm_sampler_half_step": (C.c_int, [_h, C.c_int32, C.c_int32]),
    "rscm_sampler_exchange_buffers": (C.c_int, [_h, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]),
    "rscm_sampler_apply_exchange": (C.c_int, [_h, C.c_int32]),
    "rscm_sampler_sync": (C.c_int, [_h]),
    "rscm_sampler_destroy": (C.c_int, [_h]),
    "rscm_sampler_set_groups": (C.c_int, [_h, C.c_int32]),
    "rscm_sampler_set_positions": (C.c_int, [_h, _dp]),
    "rscm_sampler_iterate": (C.c_int, [_h, C.c_int32]),
    "rscm_sampler_last_ms": (C.c_int, [_h, C.POINTER(C.c_float)]),
    "rscm_sampler_get": (C.c_int, [_h, _dp, _dp, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "rscm_ens_summary": (C.c_int, [_h, C.c_int32, C.c_int32, _dp]),
    "rscm_ens_get_params": (C.c_int, [_h, _dp]),
    "rscm_ens_sample_lhs": (C.c_int, [_h, C.c_uint64, _dp, _dp, C.c_int64, C.c_int64]),
    "rscm_gpu_host_alloc": (C.c_int, [C.c_int64, C.POINTER(C.c_void_p)]),
    "rscm_gpu_host_free": (C.c_int, [C.c_void_p]),
    "rscm_gpu_copy_to_host": (C.c_int, [C.c_int32, C.c_void_p, C.c_void_p, C.c_int64]),
    "rscm_gpu_copy_to_device": (C.c_int, [C.c_int32, C.c_void_p, C.c_void_p, C.c_int64]),
    "rscm_gpu_ocean_fit_selftest": (C.c_int, [C.c_int32, C.c_double, C.c_double, C.c_int64, _dp, _ip, _ip, _ip, _dp]),
    "rscm_ens_ocean_fast_info": (C.c_int, [_h, _ip, _dp]),
    "rscm_gpu_selftest_div": (C.c_int, [C.c_int32, C.c_int64, _dp, _dp, _dp, _dp, _bp]),
}


class RscmGpuUnavailable(RuntimeError):
    """The HIP extension is missing or cannot be loaded.  There is no CPU fallback."""


class RscmGpuError(RuntimeError):
    def __init__(self, code: int, text: str):
        super().__init__(f"rscm_gpu error {code}: {text}")
        self.code = code


_LIB = None


def load() -> C.CDLL:
    """Load the shared library and bind every declared symbol (no device call is made)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise RscmGpuUnavailable(
            f"{LIB_PATH} not found: build the HIP extension first (make -C rscm_amd/csrc). "
            "rscm_amd has no CPU fallback.")
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:  # e.g. libamdhip64 missing
        raise RscmGpuUnavailable(f"cannot load {LIB_PATH}: {e}") from e
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the header and the library diverge
        fn.restype = res
        fn.argtypes = args
    if lib.rscm_gpu_abi_version() != 1 or lib.rscm_gpu_abi_minor() < 3:
        raise RscmGpuUnavailable("ABI version mismatch (this front end needs version 1, minor >= 3): rebuild librscm_gpu.so")
    _LIB = lib
    return lib


def check(rc: int) -> None:
    if rc != OK:
        raise RscmGpuError(rc, load().rscm_gpu_last_error().decode())


def device_count() -> int:
    n = C.c_int32(0)
    rc = load().rscm_gpu_device_count(C.byref(n))
    return n.value if rc == OK else 0


def mem_info(device_id: int = 0):
    """(free, total) bytes of HBM on ``device_id``."""
    f, t = C.c_uint64(0), C.c_uint64(0)
    check(load().rscm_gpu_mem_info(device_id, C.byref(f), C.byref(t)))
    return f.value, t.value


def f64(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float64)


def dptr(a: np.ndarray):
    assert a.dtype == np.float64 and a.flags.c_contiguous
    return a.ctypes.data_as(_dp)


def iptr(a):
    if a is None:
        return None
    assert a.dtype == np.int32 and a.flags.c_contiguous
    return a.ctypes.data_as(_ip)


def bptr(a: np.ndarray):
    assert a.dtype == np.uint8 and a.flags.c_contiguous
    return a.ctypes.data_as(_bp)
