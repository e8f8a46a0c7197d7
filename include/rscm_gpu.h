/*
 * rscm_gpu.h -- C ABI of the MI355X (gfx950) ensemble runner for RSCM's two-layer hot path.
 *
 * This is the drop-in boundary: every entry point is `extern "C"`, takes plain pointers and
 * sizes (no torch / C++ types), returns an int status (0 = ok) and never unwinds across the
 * ABI.  It is what a Rust `impl ModelRunner for GpuRunner` / `impl Component` shim, or the
 * Python front-end (ctypes), binds -- see INTEGRATION.md for the reference-side stubs.
 *
 * Reference interfaces replaced (file:line under the reference tree):
 *   - Model::run / step / step_model / step_model_component
 *         crates/rscm-core/src/model/runtime.rs:368-527             -> rscm_ens_run*
 *   - ModelBuilder::build collection initialisation (initial values at index 0,
 *     exogenous series on the model axis)
 *         crates/rscm-core/src/model/builder.rs:735-830             -> rscm_ens_set_initial,
 *                                                                      rscm_ens_set_forcing
 *   - VariableSource index rule (Exogenous/OwnState -> n, UpstreamOutput -> n+1)
 *         crates/rscm-core/src/state/windows.rs:229-234             -> `source` argument
 *   - TwoLayer::solve + IVP RHS   crates/rscm-two-layer/src/component.rs:159-251
 *   - CarbonCycle::solve          crates/rscm-components/src/components/carbon_cycle.rs:102-159
 *   - CO2ERF::solve               crates/rscm-components/src/components/co2_erf.rs:57-80
 *   - scalar Sum aggregate        crates/rscm-core/src/schema.rs:760-773,886-901
 *   - RK4 driver + end-time check crates/rscm-core/src/ivp/mod.rs:73-102,245-253
 *   - ModelRunner::run_batch      crates/rscm-calibrate/src/model_runner.rs:38-85,215-267
 *         (order-preserving, per-member failure)                    -> rscm_ens_set_params*,
 *                                                                      rscm_ens_run, rscm_ens_status
 *   - extract_outputs             crates/rscm-calibrate/src/model_runner.rs:161-212
 *                                                                   -> rscm_ens_get_series
 *   - GaussianLikelihood          crates/rscm-calibrate/src/likelihood.rs:167-250
 *                                                                   -> rscm_ens_loglik
 *   - ParameterSet::sample_lhs    crates/rscm-calibrate/src/parameter_set.rs:207-233
 *                                                                   -> rscm_ens_sample_lhs
 *
 * Data layout (device, HBM): structure-of-arrays, member index fastest.
 *   params   [P][N]      f64
 *   series   [V][T][N]   f64   (index 0 of a state variable holds its initial value; outputs of
 *                               step n are written at index n+1, runtime.rs:480; never-written
 *                               entries are NaN, builder.rs:772-780)
 *   forcing  [S][T]      f64   shared scenarios, staged in LDS by the kernels
 *
 * Environment: the library reads exactly two variables, both once per process, neither changes a result.
 *   RSCM_SPLIT_RUNS=0     whole-axis runs over more members than one wavefront per SIMD are issued as ONE plain launch
 *                         instead of two member blocks on two streams in chunks of model steps (rscm_ens_last_run_plan);
 *                         same kernels on the same operands, the same bits -- an A/B and debugging switch.  Default: on.
 *   RSCM_POISON_ALLOC=1   debug aid: every device allocation starts as 0xFF bytes (NaN as a double, 255 as a status byte) so
 *                         that a read of memory nobody wrote shows in the results.  Default: off (one fill per allocation).
 * Nothing else in the environment reaches a launch plan: the sweep knobs of earlier rounds exist only in the experiments
 * build (csrc/experiment_env.hpp, `make EXPERIMENTS=1`), which rscm_gpu_experiments_build() of the internal header identifies.
 *
 * Threading: a handle is not thread-safe; use one handle per device per thread
 * (the reference calls run_batch from one thread at a time, sampler/ensemble.rs:145).
 * Ownership: the caller owns every input buffer (copied/uploaded before the call returns);
 * the library owns device buffers until rscm_ens_destroy.
 */
#ifndef RSCM_GPU_H
#define RSCM_GPU_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RSCM_GPU_ABI_VERSION 1
/* Bumped whenever an entry point is added or a mode changes what a kind computes while the major version stays:
 *   1  rscm_sampler_create_graph; four hooks moved to rscm_gpu_internal.h; RSCM_MODE_FAST acts on ClimateUDEB
 *   2  RSCM_MODE_FAST acts on the coupled chain and on CarbonCycle; rscm_gpu_abi_minor itself
 *   3  rscm_ens_last_run_plan.  In the internal header (rscm_gpu_internal.h, not part of this ABI) the same release REMOVED
 *      rscm_gpu_graph_stamps, made rscm_gpu_set_udeb_variant(4) an error (the four-wavefront kernel is gone) and re-used
 *      rscm_gpu_set_lockstep_fusion mode 4 (was: the whole-graph launch; now: mode 1 without the two-wavefront op split): a
 *      host built against the round-3 internal header does not link, or gets the new meaning of mode 4
 *   4  ClimateUDEB keeps its columns on chip at EVERY n_layers <= 64 (no entry point changed: same results, the counts
 *      other than 20 / 30 / 40 / 50 are ~15x faster); internal header: rscm_gpu_fail_chunk_launch, rscm_gpu_set_run_plan,
 *      rscm_gpu_set_udeb_variant(3)
 *   5  no entry point changed.  The shipped library no longer reads RSCM_SPLIT_CHUNK / _CHUNK2 / _FIRST, RSCM_LOCKSTEP_SPLIT or
 *      RSCM_UDEB_VARIANT (section "Environment" above); internal header: rscm_gpu_experiments_build, and
 *      rscm_gpu_fail_chunk_launch is consumed by the next cut run whether or not k is reached */
#define RSCM_GPU_ABI_MINOR 5

#if defined(__GNUC__)
#define RSCM_API __attribute__((visibility("default")))
#else
#define RSCM_API
#endif

/* ---- status codes ------------------------------------------------------------------------- */
#define RSCM_OK 0
#define RSCM_ERR_INVALID 1   /* bad argument / size (cf. model_runner.rs:225-231)               */
#define RSCM_ERR_STATE 2     /* call order / time index (cf. runtime.rs:516 assert)               */
#define RSCM_ERR_TIME_AXIS 3 /* RK4 end time misses t_next by >= 5e-3 (ivp/mod.rs:97 panics)   */
#define RSCM_ERR_DEVICE 4    /* HIP runtime error (text in rscm_gpu_last_error)                 */
#define RSCM_ERR_NOMEM 5

/* ---- model kinds -------------------------------------------------------------------------- */
#define RSCM_KIND_TWO_LAYER 0 /* stand-alone TwoLayer with a forcing series                     */
#define RSCM_KIND_COUPLED 1   /* CarbonCycle -> CO2ERF -> Sum aggregate -> TwoLayer             */

#define RSCM_KIND_UDEB 2      /* rscm-magicc ClimateUDEB (4-box upwelling-diffusion EBM)          */
#define RSCM_KIND_GHG_FORCING 3 /* rscm-magicc GhgForcing (CO2/CH4/N2O concentrations -> ERF)     */
#define RSCM_KIND_OZONE_FORCING 4    /* rscm-magicc OzoneForcing                                  */
#define RSCM_KIND_AEROSOL_DIRECT 5   /* rscm-magicc AerosolDirect (FourBox output)                */
#define RSCM_KIND_AEROSOL_INDIRECT 6 /* rscm-magicc AerosolIndirect                               */
#define RSCM_KIND_CH4_CHEMISTRY 7    /* rscm-magicc CH4Chemistry (Prather iteration)              */
#define RSCM_KIND_N2O_CHEMISTRY 8    /* rscm-magicc N2OChemistry (stratospheric delay)            */
#define RSCM_KIND_CO2_BUDGET 9       /* rscm-magicc CO2Budget                                     */
#define RSCM_KIND_TERRESTRIAL_CARBON 10 /* rscm-magicc TerrestrialCarbon (four pools)             */
#define RSCM_KIND_OCEAN_CARBON 11    /* rscm-magicc OceanCarbon (impulse-response mixed layer)    */
#define RSCM_KIND_HALOCARBON 12      /* rscm-magicc HalocarbonChemistry (41 species)              */
#define RSCM_KIND_FOURBOX_OHU 13     /* rscm-components FourBoxOceanHeatUptake                    */
#define RSCM_KIND_OSPP 14            /* rscm-components OceanSurfacePartialPressure               */
/* the members of the coupled chain on their own, for graphs assembled with rscm_ens_link_input */
#define RSCM_KIND_CARBON_CYCLE 15    /* rscm-components CarbonCycle (RK4, three states)           */
#define RSCM_KIND_CO2_ERF 16         /* rscm-components CO2ERF                                    */
#define RSCM_KIND_AGGREGATE 17       /* rscm-core schema aggregate (Sum / Mean / Weighted)        */

/* variable ids, kind TWO_LAYER (V = 3) */
#define RSCM_TL_VAR_ERF 0 /* "Effective Radiative Forcing"  (input, [S][T] shared)              */
#define RSCM_TL_VAR_TS 1  /* "Surface Temperature"          (state)                             */
#define RSCM_TL_VAR_TD 2  /* "Deep Ocean Temperature"       (state)                             */

/* variable ids, kind COUPLED (V = 8) */
#define RSCM_CP_VAR_EMISSIONS 0 /* "Emissions|CO2|Anthropogenic" (input, [S][T] shared)         */
#define RSCM_CP_VAR_TS 1
#define RSCM_CP_VAR_TD 2
#define RSCM_CP_VAR_CONC 3       /* "Atmospheric Concentration|CO2"  (state)                    */
#define RSCM_CP_VAR_CUM_UPTAKE 4 /* "Cumulative Land Uptake"         (state)                    */
#define RSCM_CP_VAR_CUM_EMIS 5   /* "Cumulative Emissions|CO2"       (state)                    */
#define RSCM_CP_VAR_ERF_CO2 6    /* "Effective Radiative Forcing|CO2" (output)                  */
#define RSCM_CP_VAR_ERF 7        /* "Effective Radiative Forcing"     (aggregate output)        */

/* variable ids, kind UDEB (V = 8): crates/rscm-magicc/src/climate/udeb/mod.rs:80-91 */
#define RSCM_UD_VAR_ERF 0          /* "Effective Radiative Forcing" (input, [S][T] shared)      */
#define RSCM_UD_VAR_ST_NH_OCEAN 1  /* "Surface Temperature" FourBox state: NorthernOcean        */
#define RSCM_UD_VAR_ST_NH_LAND 2   /*                                      NorthernLand         */
#define RSCM_UD_VAR_ST_SH_OCEAN 3  /*                                      SouthernOcean        */
#define RSCM_UD_VAR_ST_SH_LAND 4   /*                                      SouthernLand         */
#define RSCM_UD_VAR_HEAT_UPTAKE 5  /* "Heat Uptake"              (output)                       */
#define RSCM_UD_VAR_OHC 6          /* "Ocean Heat Content"       (output)                       */
#define RSCM_UD_VAR_SST 7          /* "Sea Surface Temperature"  (output)                       */

/* GhgForcing (crates/rscm-magicc/src/forcing/ghg.rs:69-83).  The shared input of this kind is a
 * block of three rows per scenario, series[n_scen][3][n_times]: "Atmospheric Concentration|CO2"
 * (ppm), "...|CH4" (ppb), "...|N2O" (ppb). */
#define RSCM_GH_VAR_CONC 0     /* the three concentration rows (input, [S][3][T] shared)         */
#define RSCM_GH_VAR_ERF_CO2 1  /* "Effective Radiative Forcing|CO2" (output)                     */
#define RSCM_GH_VAR_ERF_CH4 2  /* "Effective Radiative Forcing|CH4" (output)                     */
#define RSCM_GH_VAR_ERF_N2O 3  /* "Effective Radiative Forcing|N2O" (output)                     */
/* GhgForcing parameter rows (P = 21): GhgForcingParameters field order
 * (crates/rscm-magicc/src/parameters/ghg_forcing.rs); method 0 = Ipcctar, 1 = Olbl, [u] uniform. */
#define RSCM_GH_NPARAMS 21
#define RSCM_GH_P_METHOD 0      /* [u] */
#define RSCM_GH_P_CO2_PI 1
#define RSCM_GH_P_CH4_PI 2
#define RSCM_GH_P_N2O_PI 3
#define RSCM_GH_P_DELQ2XCO2 4
#define RSCM_GH_P_CH4_RADEFF 5
#define RSCM_GH_P_N2O_RADEFF 6
#define RSCM_GH_P_OLBL_CO2_A1 7
#define RSCM_GH_P_OLBL_CO2_B1 8
#define RSCM_GH_P_OLBL_CO2_C1 9
#define RSCM_GH_P_OLBL_CO2_D1 10
#define RSCM_GH_P_OLBL_CH4_A3 11
#define RSCM_GH_P_OLBL_CH4_B3 12
#define RSCM_GH_P_OLBL_CH4_D3 13
#define RSCM_GH_P_OLBL_N2O_A2 14
#define RSCM_GH_P_OLBL_N2O_B2 15
#define RSCM_GH_P_OLBL_N2O_C2 16
#define RSCM_GH_P_OLBL_N2O_D2 17
#define RSCM_GH_P_ADJUST_CO2 18
#define RSCM_GH_P_ADJUST_CH4 19
#define RSCM_GH_P_ADJUST_N2O 20

/* The three stateless forcing components below follow the same convention: variable 0 is the
 * block of input rows per scenario, series[n_scen][n_inputs][n_times], in the order of the
 * component's #[inputs(...)] declaration; variables 1.. are the outputs; the parameter rows are
 * the fields of the parameter struct in declaration order (arrays expanded, booleans as 0/1).
 *
 * OzoneForcing (crates/rscm-magicc/src/forcing/ozone.rs:69-85, parameters/ozone_forcing.rs):
 *   inputs  EESC, Atmospheric Concentration|CH4, Emissions|NOx, Emissions|CO, Emissions|NMVOC,
 *           Surface Temperature
 *   outputs 1 ERF|O3|Stratospheric, 2 ERF|O3|Tropospheric, 3 ERF|O3|Temperature Feedback
 *   params  eesc_reference, strat_o3_scale, strat_cl_exponent, trop_radeff, trop_oz_ch4,
 *           trop_oz_nox, trop_oz_co, trop_oz_voc, ch4_pi, nox_pi, co_pi, nmvoc_pi,
 *           temp_feedback_scale */
#define RSCM_OZ_NINPUTS 6
#define RSCM_OZ_NPARAMS 13
/* AerosolDirect (forcing/aerosol_direct.rs:53-63, parameters/aerosol.rs:6-70):
 *   inputs  Emissions|SOx, Emissions|BC, Emissions|OC, Emissions|NOx
 *   outputs 1-4 ERF|Aerosol|Direct in NorthernOcean, NorthernLand, SouthernOcean, SouthernLand
 *   params  sox/bc/oc/nitrate_coefficient, sox_regional[4], bc_regional[4], oc_regional[4],
 *           nitrate_regional[4], sox_pi, bc_pi, oc_pi, nox_pi, harmonize, harmonize_year,
 *           harmonize_target (the last three are carried but unused by solve, as upstream) */
#define RSCM_AD_NINPUTS 4
#define RSCM_AD_NPARAMS 27
/* AerosolIndirect (forcing/aerosol_indirect.rs:52-60, parameters/aerosol.rs:75-117):
 *   inputs  Emissions|SOx, Emissions|OC;   output 1 ERF|Aerosol|Indirect
 *   params  cloud_albedo_coefficient, reference_burden, sox_weight, oc_weight, sox_pi, oc_pi,
 *           harmonize, harmonize_year, harmonize_target */
#define RSCM_AI_NINPUTS 2
#define RSCM_AI_NPARAMS 9
/* CH4Chemistry (crates/rscm-magicc/src/chemistry/ch4.rs:50-66, parameters/ch4_chemistry.rs):
 *   inputs  Emissions|CH4, Surface Temperature, Emissions|NOx, Emissions|CO, Emissions|NMVOC
 *   state   1 Atmospheric Concentration|CH4 (needs an initial value);  output 2 Lifetime|CH4
 *   params  ch4_pi, natural_emissions, tau_oh, tau_soil, tau_strat, tau_trop_cl,
 *           ch4_self_feedback, oh_sensitivity_scale, oh_nox_sensitivity, oh_co_sensitivity,
 *           oh_nmvoc_sensitivity, temp_sensitivity, include_temp_feedback,
 *           include_emissions_feedback, ppb_to_tg, nox_reference, co_reference, nmvoc_reference */
#define RSCM_CH4_NINPUTS 5
#define RSCM_CH4_NPARAMS 18
/* N2OChemistry (chemistry/n2o.rs:44-54, parameters/n2o_chemistry.rs):
 *   input   Emissions|N2O;  state 1 Atmospheric Concentration|N2O;  output 2 Lifetime|N2O
 *   params  n2o_pi, natural_emissions, tau_n2o, lifetime_feedback, strat_delay (an integer >= 0
 *           held in a double), ppb_to_tg */
#define RSCM_N2O_NINPUTS 1
#define RSCM_N2O_NPARAMS 6
#define RSCM_CHEM_VAR_CONC 1
#define RSCM_CHEM_VAR_LIFETIME 2
/* CO2Budget (crates/rscm-magicc/src/carbon/budget.rs:60-75, parameters/co2_budget.rs):
 *   inputs  Emissions|CO2|Fossil, Emissions|CO2|Land Use, Carbon Flux|Terrestrial, Carbon Flux|Ocean
 *   state   1 Atmospheric Concentration|CO2;  outputs 2 Emissions|CO2|Net, 3 Airborne Fraction|CO2
 *   params  gtc_per_ppm, co2_pi */
#define RSCM_CB_NINPUTS 4
#define RSCM_CB_NPARAMS 2
/* TerrestrialCarbon (carbon/terrestrial.rs:66-82, parameters/terrestrial_carbon.rs):
 *   inputs  Atmospheric Concentration|CO2, Surface Temperature, Emissions|CO2|Land Use
 *   states  1 Carbon Pool|Plant, 2 Carbon Pool|Detritus, 3 Carbon Pool|Soil, 4 Carbon Pool|Humus
 *   output  5 Carbon Flux|Terrestrial
 *   params  npp_pi, co2_pi, beta, npp/resp/detritus/soil/humus_temp_sensitivity,
 *           plant/detritus/soil/humus_pool_pi, respiration_pi, frac_npp_to_plant,
 *           frac_npp_to_detritus, frac_plant_to_detritus, frac_detritus_to_soil,
 *           frac_soil_to_humus, enable_fertilization, enable_temp_feedback */
#define RSCM_TC_NINPUTS 3
#define RSCM_TC_NPARAMS 20
/* OceanCarbon (crates/rscm-magicc/src/carbon/ocean.rs:46-62, parameters/ocean_carbon.rs:73-196):
 *   inputs  Atmospheric Concentration|CO2, Sea Surface Temperature (the anomaly the component
 *           uses as delta_sst)
 *   states  1 Ocean Surface pCO2, 2 Cumulative Ocean Uptake;  output 3 Carbon Flux|Ocean
 *   params  model [u] (0 = 3D-GFDL, 1 = 2D-BERN, 2 = HILDA: selects the two IrfForm coefficient
 *           sets of the reference's presets; other IrfForm contents are not supported), co2_pi,
 *           pco2_pi, gas_exchange_scale, gas_exchange_tau, temp_sensitivity, irf_scale [u],
 *           mixed_layer_depth, ocean_surface_area, sst_pi, steps_per_year [u] (12),
 *           max_history_months [u], irf_switch_time [u], delta_ospp_offsets[5],
 *           delta_ospp_coefficients[5], enable_temp_feedback.  [u] rows are uniform. */
#define RSCM_OC_NINPUTS 2
#define RSCM_OC_NPARAMS 24
#define RSCM_OC_P_MODEL 0
#define RSCM_OC_P_IRF_SCALE 6
#define RSCM_OC_P_STEPS_PER_YEAR 10
#define RSCM_OC_P_MAX_HISTORY_MONTHS 11
#define RSCM_OC_P_IRF_SWITCH_TIME 12
/* HalocarbonChemistry (crates/rscm-magicc/src/chemistry/halocarbon.rs:262-300,
 * parameters/halocarbon.rs:46-160) with the species list of HalocarbonParameters::default():
 * 23 F-gases then 18 Montreal gases, in that order (41 species; other list lengths are not
 * supported on the device).
 *   inputs  Emissions|<species> x 41 (kt/yr), in species order
 *   states  1..41 Atmospheric Concentration|<species> (ppt)
 *   outputs 42 Forcing|Halocarbons, 43 Forcing|F-gases, 44 Forcing|Montreal Gases, 45 EESC
 *   params  br_multiplier, cfc11_release_normalisation, eesc_delay, air_molar_mass,
 *           atmospheric_mass_tg, mixing_box_fraction, then per species: lifetime,
 *           radiative_efficiency, concentration_pi, molecular_weight, n_cl, n_br,
 *           fractional_release */
#define RSCM_HC_NSPECIES 41
#define RSCM_HC_NINPUTS 41
#define RSCM_HC_NPARAMS (6 + 41 * 7)
/* FourBoxOceanHeatUptake (crates/rscm-components/src/components/four_box_ocean_heat_uptake.rs):
 *   input   Effective Radiative Forcing|Aggregated;  outputs 1-4 Heat Uptake|Ocean in
 *   NorthernOcean, NorthernLand, SouthernOcean, SouthernLand
 *   params  northern_ocean_ratio, northern_land_ratio, southern_ocean_ratio, southern_land_ratio
 *           (from_parameters asserts they average to 1 within 0.01; the front-end mirrors that) */
#define RSCM_FB_NINPUTS 1
#define RSCM_FB_NPARAMS 4
/* OceanSurfacePartialPressure (.../ocean_carbon_cycle/ocean_surface_partial_pressure.rs):
 *   inputs  Sea Surface Temperature, Dissolved Inorganic Carbon (both anomalies)
 *   output  1 Ocean Surface Partial Pressure|CO2
 *   params  ospp_preindustrial, sensitivity_ospp_to_temperature,
 *           sea_surface_temperature_preindustrial, delta_ospp_offsets[5], delta_ospp_coefficients[5] */
#define RSCM_SP_NINPUTS 2
#define RSCM_SP_NPARAMS 13
/* CarbonCycle (crates/rscm-components/src/components/carbon_cycle.rs:102-159):
 *   inputs  Emissions|CO2|Anthropogenic, Surface Temperature
 *   states  1 Atmospheric Concentration|CO2, 2 Cumulative Land Uptake, 3 Cumulative Emissions|CO2
 *   params  tau, conc_pi, alpha_temperature; RK4 step via rscm_ens_set_step_size(RSCM_COMP_CARBON_CYCLE)
 *   Same arithmetic as inside RSCM_KIND_COUPLED: the linked graph reproduces the fused kind's bits. */
#define RSCM_CC_NINPUTS 2
#define RSCM_CC_NPARAMS 3
/* CO2ERF (co2_erf.rs:57-80): input Atmospheric Concentration|CO2; output 1 Effective Radiative
 *   Forcing|CO2; params erf_2xco2, conc_pi */
#define RSCM_CE_NINPUTS 1
#define RSCM_CE_NPARAMS 2
/* Schema aggregate (AggregatorComponent / compute_aggregate, crates/rscm-core/src/schema.rs:760-802,
 * 886-901): up to eight contributors, every one read at index n+1 (at_end(), whatever `source` is
 * passed), NaN contributors skipped, all-NaN -> NaN.  The input block starts out all-NaN, so only the
 * rows that are linked or set take part.  output 1 the aggregate.
 *   params  operation (0 Sum, 1 Mean, 2 Weighted = sum of value * weight, no renormalisation),
 *           weights[8]
 * More than eight contributors are chained through several such ensembles (a partial enters the next stage
 * as its row 0: the additions keep compute_aggregate's order).  For a Mean that takes three helper
 * operations: 3 = number of non-NaN rows, 4 = row 0 (a count carried in) + number of non-NaN rows 1..7,
 * 5 = row 0 / row 1 (sum / count; NaN when the count is 0). */
#define RSCM_AG_NINPUTS 8
#define RSCM_AG_NPARAMS 9

/* UDEB parameter rows (P = 37): ClimateUDEBParameters field order
 * (crates/rscm-magicc/src/parameters/climate_udeb.rs), booleans/enums/integers as doubles.
 * Rows marked [u] are structural and must be equal for every member.  n_layers: any count >= 2 as in the reference
 * (climate/udeb/mod.rs:162-165; at most 4096 here).  Up to 64 layers a member's two columns stay in registers + LDS for a
 * whole launch (20, 30, 40, 50 with the count compiled in; every other count in the next capacity's instance of the same
 * unrolled solve with the count at run time); 65 to 128 layers keep the column in registers and the solve's work array in LDS
 * (about 6x the per-layer cost of the counts up to 64), more than 128 run a slower kernel with the columns in HBM (same
 * arithmetic, same parity bar throughout).  Device memory per handle besides the series: 2 x max(64, n_layers) x N x 8 B of columns,
 * 11 x N x 8 B of scalars, T x N x 8 B of temperature history, and -- for counts other than 20 / 30 / 40 / 50 --
 * n_layers x N x 8 B of work array (e.g. n_layers = 4096, N = 1e5: 9.8 GB; n_layers = 49: 90 MB).  The reference's MAGICC7
 * files pin the 50-layer configuration only: at every other count parity is against the CPU restatement kept with the tests.
 * ocean_temp_profile = 2 (CMIP5) only. */
#define RSCM_UD_NPARAMS 37
#define RSCM_UD_P_N_LAYERS 0              /* [u] */
#define RSCM_UD_P_MIXED_LAYER_DEPTH 1     /* [u] */
#define RSCM_UD_P_LAYER_THICKNESS 2       /* [u] */
#define RSCM_UD_P_KAPPA 3
#define RSCM_UD_P_KAPPA_MIN 4
#define RSCM_UD_P_KAPPA_DKDT 5
#define RSCM_UD_P_W_INITIAL 6
#define RSCM_UD_P_W_VARIABLE_FRACTION 7
#define RSCM_UD_P_W_THRESHOLD_TEMP_NH 8
#define RSCM_UD_P_W_THRESHOLD_TEMP_SH 9
#define RSCM_UD_P_ECS 10
#define RSCM_UD_P_RF_2XCO2 11
#define RSCM_UD_P_RLO 12
#define RSCM_UD_P_FEEDBACK_Q_SENSITIVITY 13
#define RSCM_UD_P_FEEDBACK_CUMT_SENSITIVITY 14
#define RSCM_UD_P_FEEDBACK_CUMT_PERIOD 15   /* [u] */
#define RSCM_UD_P_K_LO 16
#define RSCM_UD_P_K_NS 17
#define RSCM_UD_P_AMPLIFY_OCEAN_TO_LAND 18
#define RSCM_UD_P_NH_LAND_FRACTION 19
#define RSCM_UD_P_SH_LAND_FRACTION 20
#define RSCM_UD_P_DEPTH_DEPENDENT_AREA 21 /* [u] */
#define RSCM_UD_P_TEMP_ADJUST_ALPHA 22
#define RSCM_UD_P_TEMP_ADJUST_GAMMA 23
#define RSCM_UD_P_POLAR_SINKING_RATIO 24
#define RSCM_UD_P_LAND_HC_ENABLED 25      /* [u] */
#define RSCM_UD_P_K_LG 26
#define RSCM_UD_P_LAND_HC_EFF_THICKNESS 27
#define RSCM_UD_P_RF_REGIONS_CO2_0 28     /* ..31: NorthernOcean, NorthernLand, SouthernOcean, SouthernLand */
#define RSCM_UD_P_EFFICACY_APPLY 32       /* [u] */
#define RSCM_UD_P_PRESCRIBED_EFFICACY_CO2 33
#define RSCM_UD_P_OCEAN_TEMP_PROFILE 34   /* [u] */
#define RSCM_UD_P_STEPS_PER_YEAR 35       /* [u] */
#define RSCM_UD_P_MAX_TEMPERATURE 36
/* rscm_ens_status for this kind: 0 ok, 2 invalid prescribed_efficacy_co2, 4 LAMCALC did not
 * converge (ClimateUDEB::from_parameters returns Err; mod.rs:161-205) -- all outputs NaN. */

/* parameter rows.  TWO_LAYER: P = 6, TwoLayerParameters field order (component.rs:38-90):
 *   lambda0, a, efficacy, eta, heat_capacity_surface, heat_capacity_deep
 * COUPLED: P = 10: the six above, then tau, conc_pi, alpha_temperature (carbon_cycle.rs:24-34),
 *   erf_2xco2 (co2_erf.rs:18-25; CO2ERF.conc_pi == conc_pi as in docs/notebooks/coupled_model.py) */
#define RSCM_TL_NPARAMS 6
#define RSCM_CP_NPARAMS 10

/* VariableSource of the shared input as seen by its consumer (state/mod.rs:156-170) */
#define RSCM_SRC_EXOGENOUS 0 /* read index n   */
#define RSCM_SRC_UPSTREAM 1  /* read index n+1 */

/* solver components for rscm_ens_set_step_size */
#define RSCM_COMP_TWO_LAYER 0    /* reference hard-codes 0.1 (component.rs:240)                 */
#define RSCM_COMP_CARBON_CYCLE 1 /* SolverOptions.step_size, default 0.1 (carbon_cycle.rs:83)   */

/* arithmetic modes */
#define RSCM_MODE_EXACT 0 /* op-for-op the reference's f64 expression order, no FMA contraction:
                             bit-identical to the CPU oracle for the two-layer kind            */
#define RSCM_MODE_FAST 1  /* FMA + reciprocal heat capacities; |rel diff| <= 1e-11 on bounded
                             trajectories (tests/test_gpu_parity.py states the tolerance).
                             RSCM_KIND_OCEAN_CARBON: the history convolution in O(T) -- the last 60 (2D-BERN:
                             120) monthly lags explicitly, the older ones through 21 decaying modes fitted to the
                             impulse response; within 2e-10 of EXACT (tests/test_gpu_ocean.py; the tiled
                             convolution with fused multiply-adds where the fit does not apply).
                             RSCM_KIND_UDEB: one refinement term of the column solve's row reciprocals instead
                             of two (1.4e-13 from the oracle instead of 5e-14; the same 1e-9 bar,
                             tests/test_gpu_udeb.py).
                             RSCM_KIND_COUPLED and RSCM_KIND_CARBON_CYCLE (ABI minor 2): the carbon box's RK4 step
                             in closed form -- its equation is linear over a model step, so the four stages
                             collapse to C += h phi(h/lifetime) (A - C/lifetime) -- with 1/lifetime =
                             exp(-alpha T)/tau (no division), the uptake integral from the concentration
                             increments, cumulative emissions in the reference's own association (bit-exact),
                             and the two-layer half as in the two-layer kind; within 1e-11 of the oracle on
                             bounded members (measured 3e-13; tests/test_gpu_parity.py).
                             The other kinds have one arithmetic.  */

typedef struct rscm_ens rscm_ens;

/* ---- library ------------------------------------------------------------------------------ */
RSCM_API int rscm_gpu_abi_version(void);
RSCM_API int rscm_gpu_abi_minor(void);  /* RSCM_GPU_ABI_MINOR the library was built with */
/* Thread-local text of the last error raised by any call on this thread ("" if none). */
RSCM_API const char* rscm_gpu_last_error(void);
RSCM_API int rscm_gpu_device_count(int32_t* out);
/* Free and total HBM of `device_id` in bytes (sizing an ensemble against the 288 GB of an
 * MI355X: docs in DESIGN.md give the bytes per member of each kind). */
RSCM_API int rscm_gpu_mem_info(int32_t device_id, uint64_t* free_bytes, uint64_t* total_bytes);

/* ---- lifecycle ---------------------------------------------------------------------------- */
/* time_bounds has n_times+1 entries (TimeAxis.bounds, timeseries.rs:66-77) and must increase
 * strictly; step n integrates over [bounds[n], bounds[n+1]]. */
RSCM_API int rscm_ens_create(int32_t kind, int64_t n_members, int32_t n_times, const double* time_bounds,
                    int32_t device_id, rscm_ens** out);
/* As rscm_ens_create, with flags.  RSCM_FLAG_NO_SERIES (two-layer kind): keep only the initial
 * row of every state series -- for likelihood-only work through rscm_ens_run_loglik, where no
 * time series is ever written to HBM (12 GB per 1e6 members otherwise). */
#define RSCM_FLAG_NO_SERIES 1u
RSCM_API int rscm_ens_create_ex(int32_t kind, int64_t n_members, int32_t n_times,
                                const double* time_bounds, int32_t device_id, uint32_t flags,
                                rscm_ens** out);
/* RSCM_FLAG_WINDOWED (any kind): keep only a sliding window of `window_rows` rows of every series
 * -- enough for what a step and its linked consumers read (indices n and n+1, plus the rows a
 * chemistry kind looks back at) -- instead of all n_times rows, and, if out_stride > 0, every
 * out_stride-th row (t = 0, out_stride, 2 out_stride, ...) of the `out_vars` (n_out_vars < 0: every stored
 * variable) in an output store.  This is what lets a graph of linked ensembles run a long axis: 36
 * series x 9001 monthly points are 2.6 MB per member stored whole, 4.6 KB in a 16-row window plus 216 KB
 * of annual outputs.  Semantics are unchanged: outputs of step n are written at index n+1, consumers
 * read n or n+1 (state/windows.rs:229-234), rows never written read as NaN.  A windowed handle is stepped
 * in ranges shorter than its window (lock-step graphs: one step per launch); rscm_ens_get_series,
 * rscm_ens_loglik and rscm_ens_summary serve the rows that are resident (output store or window) and
 * fail with RSCM_ERR_STATE for the others; rscm_ens_rewind puts the initial rows back.
 * window_rows >= 4 and >= 2 x (look-back + 1); window_rows >= n_times gives plain full storage. */
#define RSCM_FLAG_WINDOWED 2u
RSCM_API int rscm_ens_create_windowed(int32_t kind, int64_t n_members, int32_t n_times, const double* time_bounds,
                                      int32_t device_id, uint32_t flags, int32_t window_rows, int32_t out_stride,
                                      int32_t n_out_vars, const int32_t* out_vars, rscm_ens** out);
RSCM_API int rscm_ens_destroy(rscm_ens* h);

RSCM_API int rscm_ens_n_params(const rscm_ens* h, int32_t* out);
RSCM_API int rscm_ens_n_vars(const rscm_ens* h, int32_t* out);
/* Rows per scenario of the shared input block (variable 0): 1 for the first three kinds. */
RSCM_API int rscm_ens_n_inputs(const rscm_ens* h, int32_t* out);
RSCM_API int rscm_ens_n_members(const rscm_ens* h, int64_t* out);
RSCM_API int rscm_ens_n_times(const rscm_ens* h, int32_t* out);

/* ---- configuration ------------------------------------------------------------------------ */
RSCM_API int rscm_ens_set_mode(rscm_ens* h, int32_t mode);
RSCM_API int rscm_ens_set_step_size(rscm_ens* h, int32_t component, double step);
/* [P][N] structure-of-arrays. */
RSCM_API int rscm_ens_set_params(rscm_ens* h, const double* soa);
/* [N][P] row-major, the shape ModelRunner::run_batch receives (&[Vec<f64>]). */
RSCM_API int rscm_ens_set_params_aos(rscm_ens* h, const double* aos);
/* Shared input series already on the model axis: series[n_scen][n_times]
 * (kinds with several inputs -- RSCM_KIND_GHG_FORCING and the three after it -- take the block
 * series[n_scen][n_inputs][n_times], see RSCM_GH_VAR_CONC);
 * scenario_of_member[N] or NULL (all members use scenario 0). */
RSCM_API int rscm_ens_set_forcing(rscm_ens* h, int32_t var_id, int32_t n_scen, const double* series,
                         const int32_t* scenario_of_member, int32_t source);
/* Initial value(s) at time index 0 of a state variable: n_values == 1 (broadcast) or N.
 * Also rewinds the time index to 0. */
/* Linked input: row `input_row` of h's input block is read, member by member, from the stored series
 * `src_var` of another ensemble `src` (same n_members, n_times and device) instead of the shared
 * scenario table -- the edge of a component graph (ModelBuilder::build, builder.rs:487-518) kept on
 * the device.  `source` is the consumer's VariableSource for that variable: RSCM_SRC_EXOGENOUS reads
 * index n (also the reference's choice for a producer registered *after* the consumer: lagged
 * feedback), RSCM_SRC_UPSTREAM index n+1.  ClimateUDEB reads at_start / at_end (n and n+1) and the
 * aggregate kind always n+1; both ignore `source`.
 * Rows that are not linked keep coming from rscm_ens_set_forcing's block (which is only required if
 * such rows exist).  Both ensembles must run on the same stream (rscm_ens_set_stream), and a run
 * of h over [b, e) needs src to have been stepped to e - 1 + source first: for a chain without
 * feedback run the producers over the whole axis and then the consumers; with feedback step every
 * ensemble one step at a time in graph order (what Model::step does).  `src` must outlive the link:
 * rscm_ens_destroy(src) fails while links to it exist.  Not available for RSCM_KIND_COUPLED and
 * RSCM_KIND_HALOCARBON inputs, nor with RSCM_FLAG_NO_SERIES on either side. */
RSCM_API int rscm_ens_link_input(rscm_ens* h, int32_t input_row, rscm_ens* src, int32_t src_var, int32_t source);
/* The guard "src has been stepped far enough" can be switched off (enabled = 0) for callers that
 * reproduce the reference's execution order as it is: its breadth-first order is not a topological
 * one, so a component can run before the producer of a variable it reads at index n+1 and then
 * sees what the collection holds there -- NaN in a fresh model (builder.rs:772-780).  Default on. */
RSCM_API int rscm_ens_set_link_order_check(rscm_ens* h, int32_t enabled);
/* Row `input_row` reads the scenario table again. */
RSCM_API int rscm_ens_unlink_input(rscm_ens* h, int32_t input_row);

RSCM_API int rscm_ens_set_initial(rscm_ens* h, int32_t var_id, const double* values, int64_t n_values);
/* Checkpoint / resume (the reference serialises time_index + the whole collection,
 * crates/rscm-core/src/model/runtime.rs:270-282): a run can be resumed from
 * (time index k, row k of every state variable).  rscm_ens_set_state writes row `tidx` of a stored
 * series (1 value = broadcast, or N values); rscm_ens_set_time_index moves the stepper there. */
RSCM_API int rscm_ens_set_state(rscm_ens* h, int32_t var_id, int32_t tidx, const double* values,
                                int64_t n_values);
/* RSCM_KIND_UDEB and RSCM_KIND_OCEAN_CARBON keep the reference's internal ComponentState (ocean
 * columns, flux history) on the device: for them tidx must be 0 or the current index. */
RSCM_API int rscm_ens_set_time_index(rscm_ens* h, int32_t tidx);
/* The internal ComponentState of the kinds that have one -- what the reference serialises next to the
 * collection in a checkpoint (runtime.rs:270-282): RSCM_KIND_UDEB: ocean layer temperatures
 * [2][n_layers][N], the per-member scalars [11][N] and the temperature history rows 0..time_index;
 * RSCM_KIND_OCEAN_CARBON: the flux history of the time_index * 12 months so far, or of the last
 * max_history_months (+ a few) of them if that is fewer -- the convolution reads no further back, and the
 * device keeps the history as a ring of that length.  One flat block of
 * doubles whose length depends on the current time index (0 for every other kind).
 * rscm_ens_set_internal_state puts such a block back and moves the stepper to `time_index` (the
 * one it was taken at); the stored series rows are restored with rscm_ens_set_state. */
RSCM_API int rscm_ens_internal_state_size(rscm_ens* h, int64_t* n_doubles);
RSCM_API int rscm_ens_get_internal_state(rscm_ens* h, double* out);
RSCM_API int rscm_ens_set_internal_state(rscm_ens* h, const double* in, int64_t n_doubles, int32_t time_index);
/* Use an existing hipStream_t (as void*) for all launches and copies; NULL = own stream. */
RSCM_API int rscm_ens_set_stream(rscm_ens* h, void* hip_stream);
/* A non-blocking hipStream_t on `device_id` for callers without a HIP runtime of their own (linked
 * ensembles must share one stream); destroy it after the ensembles that use it. */
RSCM_API int rscm_gpu_stream_create(int32_t device_id, void** out_stream);
RSCM_API int rscm_gpu_stream_destroy(int32_t device_id, void* hip_stream);

/* ---- stepping (Model::step / run) --------------------------------------------------------- */
/* Execute steps n = step_begin .. step_end-1 (0 <= step_begin <= step_end <= n_times-1).
 * step_begin must equal the current time index (Model::step advances it by one).
 * Fails with RSCM_ERR_TIME_AXIS, before launching anything, if for the configured RK4 step
 * sizes any model step's end time would be missed by >= 5e-3 (ivp/mod.rs:90-102; the reference
 * panics inside solve()), and with RSCM_ERR_STATE if parameters, the shared input or a state's
 * initial value are missing (builder.rs:704-717 MissingInitialValue).
 * rscm_ens_run returns after the work has completed; rscm_ens_run_async only enqueues. */
RSCM_API int rscm_ens_run(rscm_ens* h, int32_t step_begin, int32_t step_end);
RSCM_API int rscm_ens_run_async(rscm_ens* h, int32_t step_begin, int32_t step_end);
/* Model::run over a graph of linked ensembles (runtime.rs:504-527): for every step n in
 * [step_begin, step_end) each handle, in the order given (the graph order), advances by that one
 * step -- n_handles asynchronous launches per step on the handles' common stream, without
 * returning to the caller in between.  Every handle must stand at step_begin; follow with
 * rscm_ens_sync on any of them.  If a launch is refused part-way (a state error of one handle), the
 * handles before it in the order have advanced one step further than those after it. */
RSCM_API int rscm_ens_run_lockstep(rscm_ens* const* handles, int32_t n_handles, int32_t step_begin, int32_t step_end);
/* rscm_ens_run_lockstep issues one launch per step for every run of consecutive light components (chemistry,
 * forcing formulas, budgets, aggregates, grid transforms, the RK4 box models) instead of one per component:
 * each thread runs the components' per-member bodies in graph order -- every graph edge is per member, so this
 * is the same computation, bit for bit.  ClimateUDEB, OceanCarbon and HalocarbonChemistry keep their own
 * launches.  A graph made of light components only runs ALL its steps in one launch, and between the steps
 * every component keeps its varying parameters, its state and what its consumers read in thread-private LDS
 * slots instead of reading them back from HBM (the series are still written every step).
 * Where the graph order ends a step with light components and begins the next with light components (the MAGICC graph:
 * [8 light] ClimateUDEB OceanCarbon [3 light]), the two runs are consecutive launches and go out as ONE when they fit a
 * table of twelve ops: three launches per model step.  Inside a call of two steps or more the handles of the first run
 * therefore stand one step ahead of the others between the launches -- exactly where they would stand after their own
 * launch of the next step; at the end of the call every handle stands at step_end.
 * (A/B switches and launch counters for tests: include/rscm_gpu_internal.h.) */
RSCM_API int rscm_ens_sync(rscm_ens* h);
RSCM_API int rscm_ens_time_index(const rscm_ens* h, int32_t* out);
/* Rewind to time index 0 keeping parameters, forcing and initial values (outputs are
 * overwritten by the next run). */
RSCM_API int rscm_ens_rewind(rscm_ens* h);
/* Back to a fresh collection: time index 0 and every stored row after index 0 NaN again
 * (builder.rs:772-780), index 0 (initial values) kept.  rscm_ens_rewind alone leaves the rows of the
 * previous run in place, which nothing reads before rewriting them -- except a linked consumer that
 * runs ahead of its producer (rscm_ens_set_link_order_check). */
RSCM_API int rscm_ens_clear_series(rscm_ens* h);
/* Every stored row after `tidx` NaN again, the time index untouched: what a collection restored from
 * a checkpoint taken at `tidx` holds there (runtime.rs:270-282 serialises the collection as it was).
 * Needed when an already advanced model is rolled back and some component reads index n+1 of a
 * producer that runs after it (see rscm_ens_set_link_order_check). */
RSCM_API int rscm_ens_clear_rows_after(rscm_ens* h, int32_t tidx);
/* Device time of the most recent rscm_ens_run* launch sequence, from HIP events recorded on
 * the launch stream (valid after a sync). */
RSCM_API int rscm_ens_last_run_ms(rscm_ens* h, float* out_ms);
/* How the most recent rscm_ens_run* was cut into launches (ABI minor 3).  A whole-axis run of the two-layer or the coupled kind over more
 * members than the chip holds wavefronts at one per SIMD, and over at least ~190 model steps, is issued as TWO member blocks on two
 * streams (the caller's and one of the handle's own, forked and joined with events), each in chunks of ~64 model steps: the same
 * kernels on the same operands -- the same bits -- and the wavefronts even out over the SIMDs (1e5 members x 750 years: 2.7 -> 2.3 ms).
 * An unlinked whole-axis ClimateUDEB run over more than 65 536 members is cut the same way (two HALVES, chunks of ~96 steps; each
 * chunk reloads and stores the block's ocean columns and scalars, which is how rscm_ens_run in pieces resumes anyway; both halves
 * take the kernel variant chosen for the whole ensemble's size, so one run is one kernel).
 * member_blocks x step_chunks launches in all; 1 x 1 otherwise.  To the caller the run is one asynchronous operation on its stream
 * either way: the helper stream and the fork / join events are the handle's own.  Environment RSCM_SPLIT_RUNS=0 turns the cut off. */
RSCM_API int rscm_ens_last_run_plan(rscm_ens* h, int32_t* member_blocks, int32_t* step_chunks);

/* ---- outputs ------------------------------------------------------------------------------ */
/* Copy series[var][t][m] for t in {t_begin, t_begin+t_stride, ...} < t_end and
 * m in [m_begin, m_end) into out, laid out [n_t][m_end-m_begin]. */
RSCM_API int rscm_ens_get_series(rscm_ens* h, int32_t var_id, int32_t t_begin, int32_t t_end,
                        int32_t t_stride, int64_t m_begin, int64_t m_end, double* out);
/* Device pointer of series[var] ([T][N] contiguous) for zero-copy consumers. */
RSCM_API int rscm_ens_series_devptr(rscm_ens* h, int32_t var_id, void** out);
/* Device pointer of the parameter block ([P][N]) for callers that fill it on the device (the device sampler's
 * proposal kernel, a torch view).  The pointer stays valid until rscm_ens_destroy and may be written at any
 * time between launches: from this call on the handle never again treats a parameter row as uniform over the
 * members (the kernels' shortcut for rows rscm_ens_set_params found to hold one value), whatever later
 * rscm_ens_set_params calls upload. */
RSCM_API int rscm_ens_params_devptr(rscm_ens* h, void** out);
/* Per-member status after the last run: bit0 = a state variable is non-finite at the current
 * time index (the reference's failed-member case: NaN/Inf -> Err -> -inf log-posterior). */
RSCM_API int rscm_ens_status(rscm_ens* h, uint8_t* out);

/* Gaussian log-likelihood per member against observations given by (variable id, time index,
 * value, sigma); observations must be grouped by variable.  Non-finite model value -> -inf.
 * out is a host buffer [N]. */
RSCM_API int rscm_ens_loglik(rscm_ens* h, int32_t n_obs, const int32_t* obs_var, const int32_t* obs_tidx,
                    const double* obs_value, const double* obs_sigma, int32_t normalize,
                    double* out);
/* As rscm_ens_loglik, the result left on the device: *out_dev is the device address of the [N] doubles
 * (owned by the handle, valid until its next likelihood call; the work has completed on return).  For
 * callers that reduce or all-gather the per-member values without a host round trip (RCCL all-gather
 * of 8 B per member in the sharded calibration loop). */
RSCM_API int rscm_ens_loglik_device(rscm_ens* h, int32_t n_obs, const int32_t* obs_var, const int32_t* obs_tidx,
                                    const double* obs_value, const double* obs_sigma, int32_t normalize,
                                    void** out_dev);
/* Device address of the [N] status bytes rscm_ens_status copies out. */
RSCM_API int rscm_ens_status_devptr(rscm_ens* h, void** out);
/* Fused Model::run + GaussianLikelihood for the calibration loop (two-layer kind): steps every
 * member from time index 0 to the end of the axis and accumulates ln L on the fly, writing no
 * series (the time index stays 0, status is updated).  Same value as rscm_ens_run followed by
 * rscm_ens_loglik, bit for bit.  Observations grouped by variable with ascending time indices
 * inside a group. */
RSCM_API int rscm_ens_run_loglik(rscm_ens* h, int32_t n_obs, const int32_t* obs_var,
                                 const int32_t* obs_tidx, const double* obs_value,
                                 const double* obs_sigma, int32_t normalize, double* out);
/* rscm_ens_run_loglik with the result left on the device (see rscm_ens_loglik_device). */
RSCM_API int rscm_ens_run_loglik_device(rscm_ens* h, int32_t n_obs, const int32_t* obs_var,
                                        const int32_t* obs_tidx, const double* obs_value,
                                        const double* obs_sigma, int32_t normalize, void** out_dev);
/* ---- device stretch-move sampler ------------------------------------------------------------
 * EnsembleSampler::run (crates/rscm-calibrate/src/sampler/ensemble.rs:496-547) with StretchMove
 * (sampler/moves.rs:40-125) and the ParameterSet prior kept on the GPU: per half-ensemble update
 * a proposal kernel writes y = c + z (x - c), z = ((a-1)u + 1)^2 / a, straight into the evaluating
 * ensemble's parameter block, the fused run+likelihood kernel scores it and an accept kernel
 * applies q = z^(d-1) p(y)/p(x).  Random numbers are counter-based (Philox, keyed by `seed`), so a
 * run is reproducible; the reference draws from thread_rng, so only distributions compare.
 *
 * `evaluator`: an ensemble of n_walkers/2 members with parameters (rscm_ens_set_params, once: it
 * configures the structural rows of kinds that have them), forcing and initial values set; it must
 * outlive the sampler and is used exclusively by it while iterating.  A two-layer evaluator whose
 * observations have ascending time indices inside each variable group is scored by the fused
 * run+likelihood kernel (RSCM_FLAG_NO_SERIES is enough), whose launches end at the last observed time index
 * (later steps cannot change ln L; the evaluator's status then refers to that index); any other kind, or observation order,
 * is run through rscm_ens_run_async and scored from its stored series.  Sampled dimension d drives parameter row param_rows[d]; the other rows hold
 * base_params[P].  prior_kind: 0 = Uniform(low = a, high = b), 1 = Normal(mean = a, std = b),
 * 2 = LogNormal(mu = a, sigma = b); prior_low / prior_high truncate dimension d to [low, high]
 * like the reference's Bound wrapper (both NULL, or -inf / +inf entries: no truncation)
 * (distribution.rs).  Observations as for rscm_ens_run_loglik. */
typedef struct rscm_sampler rscm_sampler;
RSCM_API int rscm_sampler_create(rscm_ens* evaluator, int32_t n_walkers, int32_t n_dims,
                                 const int32_t* param_rows, const double* base_params,
                                 const int32_t* prior_kind, const double* prior_a, const double* prior_b,
                                 const double* prior_low, const double* prior_high,
                                 int32_t n_obs, const int32_t* obs_var, const int32_t* obs_tidx,
                                 const double* obs_value, const double* obs_sigma, int32_t normalize,
                                 double stretch_a, uint64_t seed, rscm_sampler** out);
/* The same sampler sharded over n_ranks processes (one per GPU): every rank holds a replica of the walker
 * positions and owns the half-walkers [rank * n, (rank + 1) * n), n = n_walkers / 2 / n_ranks, of BOTH
 * halves; `evaluator` has n members.  Per half-step a rank proposes, evaluates and accepts its block and
 * packs the block's new positions and log probabilities ([n_dims + 1][n] doubles); the ranks all-gather
 * the blocks (RCCL over xGMI: 2.8 MB in all at 1e5 walkers x 6 dimensions) and unpack them into their
 * replicas.  Proposals, complementary walkers and acceptance draws are keyed on the global walker index,
 * so the chain is the same for every n_ranks, bit for bit.  The driver loop
 * (rscm_amd.calibrate.DeviceEnsembleSampler):
 *     rscm_sampler_set_positions(s, pos)                    -- the same positions on every rank
 *     for half in 0, 1: rscm_sampler_half_step(s, half, 1); all-gather; rscm_sampler_apply_exchange(s, half)
 *     per iteration: rscm_sampler_begin_iteration(s);
 *         for half in 0, 1: rscm_sampler_half_step(s, half, 0); all-gather; rscm_sampler_apply_exchange(s, half)
 * with the all-gather from *send into *recv of rscm_sampler_exchange_buffers (device memory, on the
 * evaluator's stream: call rscm_sampler_sync first unless the collective runs on that stream).
 * rscm_sampler_get then returns all positions and log probabilities on every rank and this rank's
 * acceptance counters (zero for walkers of other ranks).  n_groups must be 1. */
RSCM_API int rscm_sampler_create_sharded(rscm_ens* evaluator, int32_t n_walkers, int32_t n_dims,
                                         const int32_t* param_rows, const double* base_params,
                                         const int32_t* prior_kind, const double* prior_a, const double* prior_b,
                                         const double* prior_low, const double* prior_high,
                                         int32_t n_obs, const int32_t* obs_var, const int32_t* obs_tidx,
                                         const double* obs_value, const double* obs_sigma, int32_t normalize,
                                         double stretch_a, uint64_t seed, int32_t rank, int32_t n_ranks,
                                         rscm_sampler** out);
RSCM_API int rscm_sampler_begin_iteration(rscm_sampler* s);
/* One half-ensemble update of this rank's block, enqueued: propose (identity != 0: score the walkers where
 * they stand), evaluate, accept, pack.  Also usable on an unsharded sampler (no pack). */
RSCM_API int rscm_sampler_half_step(rscm_sampler* s, int32_t half, int32_t identity);
RSCM_API int rscm_sampler_exchange_buffers(rscm_sampler* s, void** send, void** recv, int64_t* doubles_per_rank);
RSCM_API int rscm_sampler_apply_exchange(rscm_sampler* s, int32_t half);
RSCM_API int rscm_sampler_sync(rscm_sampler* s);
/* The same sampler over a GRAPH of linked ensembles as the evaluator -- EnsembleSampler<R: ModelRunner, L> is generic over the
 * runner (sampler/ensemble.rs:86-106,143-177); here the runner is any component graph stepped by rscm_ens_run_lockstep.
 * handles[0 .. n_handles) in graph order, n_walkers / 2 / n_ranks members each, whole series stored, one stream.  Sampled
 * dimension d is parameter row param_rows[d] of handles[param_owner[d]] (every other parameter keeps what rscm_ens_set_params
 * gave it); observation j is variable obs_var[j] of handles[obs_owner[j]] at time index obs_tidx[j].  Every half-step: the
 * handles are rewound (clear_between_runs != 0: their stored rows NaN again, for graphs in which a consumer runs ahead of its
 * producer, rscm_ens_set_link_order_check), the proposal kernel writes each proposed value into its owner's parameter block,
 * the graph is stepped to the last observed index -- later steps cannot change ln L -- and the likelihood kernel sums over the
 * observation rows where their owners store them: no host round trip per sweep.  Everything else (priors, groups, sharding,
 * driving calls, reproducibility) as for rscm_sampler_create_sharded. */
RSCM_API int rscm_sampler_create_graph(rscm_ens* const* handles, int32_t n_handles, int32_t clear_between_runs, int32_t n_walkers,
                              int32_t n_dims, const int32_t* param_owner, const int32_t* param_rows, const int32_t* prior_kind,
                              const double* prior_a, const double* prior_b, const double* prior_low, const double* prior_high,
                              int32_t n_obs, const int32_t* obs_owner, const int32_t* obs_var, const int32_t* obs_tidx,
                              const double* obs_value, const double* obs_sigma, int32_t normalize, double stretch_a, uint64_t seed,
                              int32_t rank, int32_t n_ranks, rscm_sampler** out);
RSCM_API int rscm_sampler_destroy(rscm_sampler* s);
/* Split the walkers into n_groups independent ensembles of n_walkers / n_groups walkers each
 * (consecutive blocks of the walker index): every group is a sampler of its own -- its own two
 * halves, complementary walkers drawn from itself only -- and all groups advance in the same
 * launches.  This is how ensembles of the reference's usual size (tens of walkers) fill a GPU:
 * thousands of them side by side, e.g. for an R-hat across independent runs.  Default 1. */
RSCM_API int rscm_sampler_set_groups(rscm_sampler* s, int32_t n_groups);
/* positions[n_walkers][n_dims] row-major (the Chain layout); scores every walker and zeroes the
 * acceptance counters. */
RSCM_API int rscm_sampler_set_positions(rscm_sampler* s, const double* positions);
/* n_iterations full sweeps (first half against the second, then the second against the updated
 * first); synchronous.  rscm_sampler_last_ms reports the device time of the last call. */
RSCM_API int rscm_sampler_iterate(rscm_sampler* s, int32_t n_iterations);
RSCM_API int rscm_sampler_last_ms(const rscm_sampler* s, float* out);
/* Any output may be NULL.  positions[n_walkers][n_dims], log_prob[n_walkers] (log prior + log
 * likelihood, -inf for failed members), per-walker acceptance counters. */
RSCM_API int rscm_sampler_get(rscm_sampler* s, double* positions, double* log_prob, int64_t* n_accepted,
                              int64_t* n_proposed);

/* Ensemble summary of one variable at one time index over finite members:
 * out[0]=count_finite, out[1]=sum, out[2]=min, out[3]=max (wavefront + block reductions). */
RSCM_API int rscm_ens_summary(rscm_ens* h, int32_t var_id, int32_t tidx, double out[4]);
/* The same four numbers for every time index in [t_begin, t_end) in two launches:
 * out[(t_end - t_begin)][4].  Each row carries the bits rscm_ens_summary returns for it. */
RSCM_API int rscm_ens_summary_series(rscm_ens* h, int32_t var_id, int32_t t_begin, int32_t t_end, double* out);
/* Ensemble quantiles of a stored variable at every time index of [t_begin, t_end): the plume (median,
 * 5-95 % band ...) reduced on the device.  Definition: numpy.nanquantile(row, q, method="linear") --
 * NaN members left out, virtual index (n - 1) q, numpy's interpolation -- so the results carry numpy's
 * bits.  out[(t - t_begin)][n_q]; count[(t - t_begin)] (or NULL) = members that are not NaN.  Rows beyond
 * the current time index: count 0, quantiles NaN.  An extension: the reference has no ensemble
 * statistics; q in [0, 1]. */
RSCM_API int rscm_ens_quantile_series(rscm_ens* h, int32_t var_id, int32_t t_begin, int32_t t_end, int32_t n_q,
                                      const double* q, double* out, double* count);

/* Copy the parameter matrix back to the host as [P][N] (e.g. after rscm_ens_sample_lhs). */
RSCM_API int rscm_ens_get_params(rscm_ens* h, double* out_soa);

/* ---- device-side Latin hypercube ---------------------------------------------------------- */
/* Fill params[j][i] = low[j] + u * (high[j] - low[j]) with one sample per stratum and dimension:
 * u = (perm_j(g) + U_j(g)) / n_total, g = member_offset + i, perm_j a keyed bijection of
 * [0, n_total).  Counter-based, so ranks that own disjoint member blocks of one global
 * ensemble generate their rows with no communication. */
RSCM_API int rscm_ens_sample_lhs(rscm_ens* h, uint64_t seed, const double* low, const double* high,
                        int64_t member_offset, int64_t n_total);

/* ---- pinned host buffers -------------------------------------------------------------------- */
/* Page-locked host memory for the buffers handed to rscm_ens_get_series / rscm_ens_set_params:
 * the copy then runs as one DMA at PCIe rate instead of being staged through pageable memory
 * (measured 601 MB of Ts: 11 GB/s into a fresh pageable buffer vs the pinned rate quoted in
 * DESIGN.md section 6). */
RSCM_API int rscm_gpu_host_alloc(int64_t n_bytes, void** out);
/* Blocking copy of n_bytes from a device address handed out by this library (rscm_ens_*_devptr,
 * rscm_ens_loglik_device) into host memory, for callers without a HIP runtime of their own. */
RSCM_API int rscm_gpu_copy_to_host(int32_t device_id, void* host, const void* device_ptr, int64_t n_bytes);
RSCM_API int rscm_gpu_copy_to_device(int32_t device_id, void* device_ptr, const void* host, int64_t n_bytes);
RSCM_API int rscm_gpu_host_free(void* p);

/* ---- diagnostics --------------------------------------------------------------------------- */
/* Whether this OceanCarbon ensemble's RSCM_MODE_FAST runs the recurrence, and the fit's deviation. */
RSCM_API int rscm_ens_ocean_fast_info(rscm_ens* h, int32_t* uses_recurrence, double* fit_error);
#ifdef __cplusplus
}
#endif
#endif /* RSCM_GPU_H */
