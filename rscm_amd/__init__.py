"""rscm_amd -- MI355X (gfx950) ensemble runner for RSCM's two-layer hot path.

The compute path is hand-written HIP behind the C-ABI of ``include/rscm_gpu.h``
(``rscm_amd/librscm_gpu.so``); this package is the host-side mirror of the reference's Python
surface for that path.  There is no CPU fallback: without the HIP library every entry point
raises ``RscmGpuUnavailable``.
"""
from ._lib import (KIND_AEROSOL_DIRECT, KIND_AEROSOL_INDIRECT, KIND_AGGREGATE, KIND_CARBON_CYCLE, KIND_CH4_CHEMISTRY,
                   KIND_CO2_BUDGET, KIND_CO2_ERF,
                   KIND_COUPLED, KIND_FOURBOX_OHU, KIND_GHG_FORCING, KIND_HALOCARBON, KIND_N2O_CHEMISTRY,
                   KIND_OCEAN_CARBON, KIND_OSPP,
                   KIND_OZONE_FORCING, KIND_TERRESTRIAL_CARBON, KIND_TWO_LAYER, KIND_UDEB, MODE_EXACT, MODE_FAST, SRC_EXOGENOUS,
                   SRC_UPSTREAM, RscmGpuError, RscmGpuUnavailable)
from .ensemble import Ensemble, run_lockstep

__all__ = ["Ensemble", "run_lockstep", "KIND_TWO_LAYER", "KIND_COUPLED", "KIND_UDEB", "KIND_GHG_FORCING", "KIND_OZONE_FORCING",
           "KIND_AEROSOL_DIRECT", "KIND_AEROSOL_INDIRECT", "KIND_CH4_CHEMISTRY", "KIND_N2O_CHEMISTRY",
           "KIND_CO2_BUDGET", "KIND_TERRESTRIAL_CARBON", "KIND_OCEAN_CARBON", "KIND_HALOCARBON", "KIND_FOURBOX_OHU", "KIND_OSPP",
           "KIND_CARBON_CYCLE", "KIND_CO2_ERF", "KIND_AGGREGATE", "MODE_EXACT", "MODE_FAST",
           "SRC_EXOGENOUS", "SRC_UPSTREAM", "RscmGpuError", "RscmGpuUnavailable"]
