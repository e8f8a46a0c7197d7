// Host-side internals shared by the translation units of librscm_gpu.so (rscm_gpu.cpp: handles, validation,
// marshalling; lockstep.cpp: the lock-step scheduler of component graphs).  Not part of the boundary
// (include/rscm_gpu.h) and not visible outside the library (-fvisibility=hidden).
#pragma once

#include "../../include/rscm_gpu.h"
#include "../../include/rscm_gpu_internal.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <limits>
#include <new>
#include <cstdlib>

// Every device allocation of the host layer goes through dev_malloc.  Debug aid (RSCM_POISON_ALLOC=1 in the environment): it is filled with 0xFF bytes -- NaN as a
// double, -1 as an integer, 255 as a status byte -- so that a kernel or a copy that reads memory nobody wrote shows up as a wrong
// result in the parity tests instead of passing on whatever a fresh allocation happens to hold (usually zeros).  Off by default: one
// fill per allocation.  The GPU tier is run once per round with it on.
namespace rscm {
template <class T>
inline hipError_t dev_malloc(T** p, size_t n)
{
    static const bool poison = [] { const char* e = getenv("RSCM_POISON_ALLOC"); return e && atoi(e) != 0; }();
    const hipError_t e = hipMalloc(reinterpret_cast<void**>(p), n);
    if (e != hipSuccess || !poison || n == 0) return e;
    const hipError_t f = hipMemset(*p, 0xFF, n);
    return f == hipSuccess ? hipDeviceSynchronize() : f;
}
}  // namespace rscm
#include <string>
#include <vector>

#include "rscm_device.hpp"

// thread-local error text behind rscm_gpu_last_error(); returns `code`
int fail(int code, const char* fmt, ...);

#define HIPCHK(expr)                                                                       \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess)                                                              \
            return fail(e_ == hipErrorOutOfMemory ? RSCM_ERR_NOMEM : RSCM_ERR_DEVICE,      \
                        "%s failed: %s", #expr, hipGetErrorString(e_));                    \
    } while (0)

#define GUARD_BEGIN try {
#define GUARD_END                                                                          \
    }                                                                                      \
    catch (const std::bad_alloc&) { return fail(RSCM_ERR_NOMEM, "host allocation failed"); } \
    catch (...) { return fail(RSCM_ERR_INVALID, "unexpected C++ exception"); }

#define NEED(h) \
    if (!(h)) return fail(RSCM_ERR_INVALID, "handle is NULL")

// Window upkeep of the handles of a lock-step run, collected while a model step (or a chunk of steps) is enqueued and issued as
// one or two launches at its end (rscm_device.hpp, WindowBatch).  first: moves and output-row copies; second: fills; then the
// windows' new first rows take effect on the host side.
struct WindowDeferral {
    std::vector<rscm::WindowOp> first, second;
    std::vector<std::pair<rscm_ens*, int32_t>> new_win0;
};

// Cached state of rscm_ens_run_lockstep for one list of handles (kept by the first of them): the device
// table of fused-launch operations (csrc/group.hip) and what it currently holds.
struct LockstepPlan {
    std::vector<rscm_ens*> handles;
    rscm::GroupOp* d_ops = nullptr;          // [handles.size()]
    std::vector<rscm::GroupOp> cached;       // the table's contents (step fields zeroed)
    std::vector<uint8_t> valid;
    rscm::GroupOp* staging = nullptr;        // page-locked ring the uploads are sourced from
    int32_t ring_pos = 0;
    static constexpr int32_t kRing = 128;
};

struct rscm_ens {
    int32_t kind = 0;
    int64_t N = 0;
    int32_t T = 0;
    int32_t rows = 0;  // stored rows per series: T, or 1 with RSCM_FLAG_NO_SERIES
    int32_t device = 0;
    int32_t P = 0, V = 0;
    int32_t mode = RSCM_MODE_EXACT;
    std::vector<double> bounds;
    double h_tl = 0.1, h_cc = 0.1;
    bool schedule_dirty = true;
    std::vector<int32_t> nsub_tl, nsub_cc;
    int32_t* d_nsub_tl = nullptr;
    int32_t* d_nsub_cc = nullptr;

    double* d_params = nullptr;  // [P][N]
    int32_t ag_rows_set = 0;     // aggregate kind: 1 + the highest contributor row rscm_ens_set_forcing has ever been given data for
    uint64_t uniform_rows = 0;   // bit j: parameter row j (< 64) holds one value for every member (the kernels then read element 0: param_at)
    // member constants of the kinds that have them (GhgForcing, TerrestrialCarbon; rscm_device.hpp, launch_*_derive): [kDerivedRows][N],
    // re-formed by ensure_derived() at the next run after anything wrote the parameter block
    // whole-axis runs as two member blocks on two streams in chunks of model steps (rscm_gpu.cpp, plan_member_split): the second stream
    // and the fork / join events, created with the first such run
    int32_t last_blocks = 1, last_chunks = 1;   // how the last run was cut (rscm_ens_last_run_plan)
    hipStream_t split_stream = nullptr;
    hipEvent_t split_fork = nullptr, split_join = nullptr;
    double* d_derived = nullptr;
    bool derived_dirty = true;
    bool params_exposed = false; // rscm_ens_params_devptr handed the block out: uniform_rows stays 0 for the life of the handle
    bool derived_hold = false;   // inside one rscm_ens_run_lockstep call: the member constants were formed at its start and serve all its steps
    double* d_series = nullptr;  // [(V-1)][T][N], variable v at slot v-1
    double* d_forcing = nullptr; // [S][n_inputs][T]
    int32_t n_inputs = 1;        // rows per scenario of the shared input block
    double* d_ghg_tables = nullptr;  // GhgForcing: [S][kGhgRows][T] derived scenario rows
    int32_t ghg_method = 1;
    // OceanCarbon: flux history (internal state) and the tabulated impulse response
    double* d_ocean_hist = nullptr;  // [ocean_hist_rows][N]: a ring, pulse j in row j mod ocean_hist_rows
    int64_t ocean_hist_rows = 0;     // min((T-1)*12, max_hist + slack): the convolution never looks further back
    double* d_ocean_irf = nullptr;   // [max(max_hist, 1)]
    double* d_ocean_partial = nullptr;  // [(tile years - 1) * steps][N] split-tile running sums (one-step launches)
    int32_t ocean_tile_base = -1;       // first step of the split tile d_ocean_partial belongs to, -1: none
    int32_t ocean_tile_years = 0;       // its length (depends on the arithmetic mode it was started in)
    int32_t ocean_steps = 0;
    int64_t ocean_max_hist = 0;
    bool ocean_ready = false;
    // RSCM_MODE_FAST: the far response as decaying modes (ocean_fit_modes) and their running sums
    rscm::OceanModes ocean_modes{};
    bool ocean_recur_ok = false;
    int32_t ocean_near = 0;
    double ocean_fit_error = 0.0;
    double* d_ocean_mode_state = nullptr;  // [kOceanModes][N]
    double* d_ocean_mode_table = nullptr;  // [3][kOceanModes]: d_q, c_q, e_q
    int32_t ocean_modes_at = -1;           // time index the running sums stand at (-1: re-form them from the history)
    int32_t* d_scen = nullptr;   // [N] or null
    int32_t n_scen = 0;
    int32_t source = RSCM_SRC_EXOGENOUS;
    uint8_t* d_status = nullptr;

    // ClimateUDEB internal state
    double* d_ocean = nullptr;    // [2][NL][N]
    int32_t udeb_ocean_layers = 0;   // layers d_ocean has room for
    double* d_udeb_work = nullptr;   // [NL][N]  the any-layer-count kernel's c' array (udeb_any_body.hpp)
    double* d_udeb_tables = nullptr; // [NL][6]  its geometry table
    int32_t udeb_work_layers = 0;
    double* d_scal = nullptr;     // [kUdebScalars][N]
    double* d_hist = nullptr;     // [T][N]
    double* d_tables = nullptr;   // geometry tables
    double* d_bounds = nullptr;   // [T+1]
    int32_t* d_win_kfull = nullptr;  // [T]
    double* d_win_partw = nullptr;   // [T]
    int32_t udeb_n_layers = 0, udeb_steps = 0, udeb_land_hc = 0, udeb_efficacy = 0;
    bool udeb_ready = false;
    std::vector<double> udeb_tables;

    double* d_partial = nullptr;  // summary scratch
    double* d_out4 = nullptr;
    double* d_loglik = nullptr;   // [N]
    // observations prepared for the fused run+likelihood kernel (prepare_obs)
    void* d_obs = nullptr;
    size_t obs_capacity = 0;
    int32_t obs_n = 0, obs_normalize = 0, obs_first_is_deep = 0;
    int32_t obs_last_tidx = 0;        // the latest time index any prepared observation refers to
    bool loglik_stop_at_last_obs = false;  // fused run+likelihood launches end there (the device sampler: only ln L is used)

    hipStream_t stream = nullptr;
    bool own_stream = false;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool timed = false;

    // linked inputs (rscm_ens_link_input)
    struct Link { rscm_ens* src = nullptr; int32_t var = 0; int32_t off = 0; };
    Link links[rscm::kMaxLinks];
    int32_t n_linked = 0;
    bool link_order_check = true;
    int32_t link_refs = 0;  // links of other ensembles into this one's series

    LockstepPlan* plan = nullptr;  // rscm_ens_run_lockstep with this handle first
    WindowDeferral* defer = nullptr;  // set while rscm_ens_run_lockstep collects this handle's window upkeep (lockstep.cpp)

    int32_t time_index = 0;
    bool params_set = false, forcing_set = false;
    std::vector<uint8_t> initial_set;  // per variable id

    // Windowed storage (RSCM_FLAG_WINDOWED): d_series holds rows [win0, win0 + rows) of every series.
    bool windowed = false;
    int32_t win0 = 0;            // absolute time index of the first stored row
    int32_t lookback = 0;        // own rows before the current one that a step reads (chemistry kinds)
    bool read_ahead = false;     // a linked consumer may read index n+1 before this producer wrote it: rows after a slide must be NaN
    double* d_row0 = nullptr;    // [V-1][N] the initial rows, saved when step 0 starts (rewind restores them)
    bool row0_saved = false;
    int32_t out_stride = 0;      // > 0: every out_stride-th row of the output variables is kept in d_out
    int32_t n_out = 0, out_rows = 0;
    std::vector<int32_t> out_vars;      // variable ids kept
    std::vector<int32_t> out_slot;      // per variable id: slot in d_out or -1
    int32_t* d_out_vars = nullptr;
    double* d_out = nullptr;     // [n_out][out_rows][N]
    int32_t keep_rows() const { return std::max(lookback + 1, 2); }

    // Base of series[var] such that row t lives at base + t * N (for a windowed handle the address of
    // the virtual row 0: only rows [win0, win0 + rows) exist).
    double* series(int32_t var) const
    {
        return d_series + ((int64_t)(var - 1) * (int64_t)rows - (int64_t)win0) * N;
    }
    // Device address of row t of a stored variable where it is resident: the output store (every
    // out_stride-th row of the output variables, up to the current index), else the window; nullptr
    // if the row is not held any more.
    const double* row_ptr(int32_t var, int32_t t) const
    {
        if (!windowed) return (rows == T || t == 0) ? series(var) + (size_t)t * N : nullptr;
        if (t >= win0 && t < win0 + rows) return series(var) + (size_t)t * N;  // the window is the live copy
        if (out_stride > 0 && out_slot[var] >= 0 && t % out_stride == 0 && t <= time_index)
            return d_out + ((size_t)out_slot[var] * out_rows + (size_t)(t / out_stride)) * N;
        return nullptr;
    }
    bool is_state(int32_t var) const
    {
        if (kind == RSCM_KIND_TWO_LAYER) return var == RSCM_TL_VAR_TS || var == RSCM_TL_VAR_TD;
        if (kind == RSCM_KIND_UDEB) return var >= RSCM_UD_VAR_ST_NH_OCEAN && var <= RSCM_UD_VAR_ST_SH_LAND;
        if (kind == RSCM_KIND_CH4_CHEMISTRY || kind == RSCM_KIND_N2O_CHEMISTRY) return var == RSCM_CHEM_VAR_CONC;
        if (kind == RSCM_KIND_CO2_BUDGET) return var == 1;
        if (kind == RSCM_KIND_TERRESTRIAL_CARBON) return var >= 1 && var <= 4;
        if (kind == RSCM_KIND_OCEAN_CARBON) return var == 1 || var == 2;
        if (kind == RSCM_KIND_HALOCARBON) return var >= 1 && var <= RSCM_HC_NSPECIES;
        if (kind == RSCM_KIND_CARBON_CYCLE) return var >= 1 && var <= 3;
        if (kind >= RSCM_KIND_GHG_FORCING) return false;  // stateless components
        return var >= RSCM_CP_VAR_TS && var <= RSCM_CP_VAR_CUM_EMIS;
    }
};

// Parameter rows from which the HOST derives state when rscm_ens_set_params / rscm_ens_sample_lhs configure an ensemble
// (ClimateUDEB's geometry tables and look-back windows, OceanCarbon's impulse-response table and mode fit, GhgForcing's
// method switch, the N2O look-back): marked [u] in rscm_gpu.h.  They must hold ONE value for all members, and nothing that
// writes parameters on the device (the samplers' proposals) may touch them -- the derived tables would silently stay those
// of the base value, where the reference rebuilds the component for every parameter vector (model_runner.rs:257-266).
// Returns the row's name for the error text, or nullptr.
inline const char* structural_row(int32_t kind, int32_t row)
{
    if (kind == RSCM_KIND_UDEB) {
        switch (row) {
            case RSCM_UD_P_N_LAYERS: return "n_layers";
            case RSCM_UD_P_MIXED_LAYER_DEPTH: return "mixed_layer_depth";
            case RSCM_UD_P_LAYER_THICKNESS: return "layer_thickness";
            case RSCM_UD_P_DEPTH_DEPENDENT_AREA: return "depth_dependent_area";
            case RSCM_UD_P_LAND_HC_ENABLED: return "land_heat_capacity_enabled";
            case RSCM_UD_P_EFFICACY_APPLY: return "efficacy_apply";
            case RSCM_UD_P_OCEAN_TEMP_PROFILE: return "ocean_temp_profile";
            case RSCM_UD_P_STEPS_PER_YEAR: return "steps_per_year";
            case RSCM_UD_P_FEEDBACK_CUMT_PERIOD: return "feedback_cumt_period";
            default: return nullptr;
        }
    }
    if (kind == RSCM_KIND_OCEAN_CARBON) {
        switch (row) {
            case RSCM_OC_P_MODEL: return "model";
            case RSCM_OC_P_IRF_SCALE: return "irf_scale";
            case RSCM_OC_P_STEPS_PER_YEAR: return "steps_per_year";
            case RSCM_OC_P_MAX_HISTORY_MONTHS: return "max_history_months";
            case RSCM_OC_P_IRF_SWITCH_TIME: return "irf_switch_time";
            default: return nullptr;
        }
    }
    if (kind == RSCM_KIND_GHG_FORCING && row == RSCM_GH_P_METHOD) return "method";
    if (kind == RSCM_KIND_N2O_CHEMISTRY && row == 4) return "stratospheric delay (sets the look-back)";
    return nullptr;
}

inline int set_device(const rscm_ens* h)
{
    HIPCHK(hipSetDevice(h->device));
    return RSCM_OK;
}

// One launch range of one handle, in pieces (rscm_gpu.cpp); rscm_ens_run_lockstep (lockstep.cpp) fuses the
// launches of several handles out of the same pieces.
extern "C" {
int step_check(rscm_ens* h, int32_t step_begin, int32_t step_end, bool derive = true);
void set_run_plan(int32_t mode);          // A/B hook: -1 default, 0 one plain launch, 1 the two-stream cut where it applies (calling thread)
void set_fail_chunk_launch(int32_t k);   // test hook: the k-th chunk launch of the calling thread's next cut run fails (0: off)
int64_t take_derive_launches();           // test hook: member-constant kernels launched by the calling thread since the last call
int ensure_derived(rscm_ens* h);   // (every run starts with current member constants: run_range after its first event, rscm_ens_run_lockstep once per call)
int step_window_pre(rscm_ens* h, int32_t step_begin, int32_t step_end);
int step_links(rscm_ens* h, int32_t step_begin, int32_t step_end, rscm::InputLinks& links, int32_t& linked_out);
// op_out: nothing is launched, the arguments go into a fused launch's table (kind -1: this handle cannot be fused)
int step_launch(rscm_ens* h, int32_t step_begin, int32_t step_end, const rscm::InputLinks& links, int32_t linked, rscm::GroupOp* op_out);
int step_finish(rscm_ens* h, int32_t step_begin, int32_t step_end);
int run_range(rscm_ens* h, int32_t step_begin, int32_t step_end, bool timed);
}
// issue what a WindowDeferral holds on `stream` and make the new windows current (rscm_gpu.cpp)
int window_flush(WindowDeferral* d, hipStream_t stream);

// the fused run+likelihood pieces the sampler (sampler_host.cpp) shares with rscm_ens_run_loglik (rscm_gpu.cpp)
extern "C" {
int prepare_obs(rscm_ens* h, int32_t n_obs, const int32_t* obs_var, const int32_t* obs_tidx, const double* obs_value, const double* obs_sigma,
                int32_t normalize);
int check_loglik_ready(rscm_ens* h);
hipError_t launch_loglik(rscm_ens* h);   // asynchronous, with the prepared observations; fills h->d_loglik
}
