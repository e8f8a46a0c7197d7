// C-ABI layer of the MI355X ensemble runner: handle management, validation mirroring the
// reference's error behaviour, host<->device plumbing.  The arithmetic of the hot path lives in
// the .hip kernels.  Two things are tabulated here, once per scenario / parameter set, with the
// host's libm: the concentration-only factors of GhgForcing (ghg_tables: ln, sqrt, pow of the shared
// scenario rows) and OceanCarbon's impulse response per lag (ocean_irf_table).  Both kinds are
// tolerance-parity kinds for that reason among others (tests/test_gpu_ghg.py, tests/test_gpu_ocean.py
// state the bounds; GhgForcing's linked-input path evaluates the same factors with the device
// library and is compared with the table path in tests/test_gpu_links.py).
#include "ens.hpp"
#include "experiment_env.hpp"
#include <functional>

#include "udeb_tables.hpp"

namespace {
thread_local std::string g_last_error;
}

int fail(int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

namespace {

constexpr double kTThreshold = 5e-3;  // crates/rscm-core/src/ivp/mod.rs:73

// ode_solvers Rk4::integrate: n = ceil((t1 - t0)/h) steps of t += h; the caller
// (get_last_step, ivp/mod.rs:90-102) asserts more than one stored point and
// |t_last - t1| < 5e-3.
bool rk4_schedule(const std::vector<double>& bounds, double h, std::vector<int32_t>& nsub,
                  int32_t* bad_step)
{
    const size_t K = bounds.size() - 2;  // T-1 steps
    nsub.assign(K, 0);
    for (size_t n = 0; n < K; ++n) {
        const double t0 = bounds[n], t1 = bounds[n + 1];
        const double m = std::ceil((t1 - t0) / h);
        if (!(m >= 1.0) || m > 1e7) {
            *bad_step = (int32_t)n;
            return false;
        }
        double t = t0;
        for (int32_t s = 0; s < (int32_t)m; ++s) t = t + h;
        if (!(std::fabs(t - t1) < kTThreshold)) {
            *bad_step = (int32_t)n;
            return false;
        }
        nsub[n] = (int32_t)m;
    }
    return true;
}

}  // namespace

// GhgForcing: everything that depends on the concentrations alone, once per scenario and year
// (forcing/ghg.rs:119-269 evaluates these per call; csrc/ghg.hip documents the factorisation).
// conc is [S][3][T] (CO2, CH4, N2O); the result [S][kGhgRows][T].
static std::vector<double> ghg_tables(const double* conc, int32_t n_scen, int32_t T)
{
    std::vector<double> t((size_t)n_scen * rscm::kGhgRows * T);
    for (int32_t s = 0; s < n_scen; ++s) {
        const double* c = conc + (size_t)s * 3 * T;
        double* o = t.data() + (size_t)s * rscm::kGhgRows * T;
        for (int32_t n = 0; n < T; ++n) {
            const double co2 = c[n], ch4 = c[(size_t)T + n], n2o = c[(size_t)2 * T + n];
            o[(size_t)rscm::kGhgCo2 * T + n] = co2;
            o[(size_t)rscm::kGhgLnCo2 * T + n] = std::log(co2);
            o[(size_t)rscm::kGhgSqrtCo2 * T + n] = std::sqrt(co2);
            o[(size_t)rscm::kGhgSqrtCh4 * T + n] = std::sqrt(ch4);
            o[(size_t)rscm::kGhgSqrtN2o * T + n] = std::sqrt(n2o);
            o[(size_t)rscm::kGhgCh4P75 * T + n] = std::pow(ch4, 0.75);
            o[(size_t)rscm::kGhgCh4TimesP152 * T + n] = ch4 * std::pow(ch4, 1.52);
            o[(size_t)rscm::kGhgN2oP75 * T + n] = std::pow(n2o, 0.75);
            o[(size_t)rscm::kGhgN2oP152 * T + n] = std::pow(n2o, 1.52);
        }
    }
    return t;
}

// OceanCarbon: the scaled mixed-layer impulse response at lag k/12 years, k = 0 .. n-1
// (OceanCarbonParameters::irf + scale_irf, parameters/ocean_carbon.rs:202-216, with the IrfForm
// coefficient sets of the gfdl_3d / bern_2d / hilda presets, :88-196).  The expressions are the
// reference's, evaluated once per lag instead of once per (pulse, sub-step) pair.
struct OceanIrfForm { bool poly; int n; double c[8]; double tau[8]; };

static void ocean_forms(int model, const OceanIrfForm** early_out, const OceanIrfForm** late_out)
{
    typedef OceanIrfForm Form;
    static const Form gfdl_e = {true, 7, {1.0, -2.2617, 14.002, -48.770, 82.986, -67.527, 21.037}, {0}};
    static const Form gfdl_l = {false, 6, {0.01481, 0.019439, 0.038344, 0.066485, 0.24966, 0.70367},
                                {1.0e10, 347.55, 65.359, 15.281, 2.3488, 0.70177}};
    static const Form bern_e = {false, 6, {0.058648, 0.07515, 0.079338, 0.41413, 0.24845, 0.12429},
                                {1.0e10, 9.6218, 9.2364, 0.7603, 0.16294, 0.0032825}};
    static const Form bern_l = {false, 6, {0.01369, 0.012456, 0.026933, 0.026994, 0.036608, 0.06738},
                                {1.0e10, 331.54, 107.57, 38.946, 11.677, 10.515}};
    static const Form hilda_e = {false, 5, {0.12935, 0.24093, 0.24071, 0.17003, 0.21898},
                                 {1.0e10, 4.9792, 0.96083, 0.26936, 0.034569}};
    static const Form hilda_l = {false, 6, {0.022936, 0.035549, 0.037820, 0.089318, 0.13963, 0.24278},
                                 {1.0e10, 232.30, 68.736, 18.601, 5.2528, 1.2679}};
    *early_out = model == 1 ? &bern_e : model == 2 ? &hilda_e : &gfdl_e;
    *late_out = model == 1 ? &bern_l : model == 2 ? &hilda_l : &gfdl_l;
}

static std::vector<double> ocean_irf_table(int model, double irf_scale, double switch_time, int64_t n)
{
    typedef OceanIrfForm Form;
    const Form *early_p, *late_p;
    ocean_forms(model, &early_p, &late_p);
    const Form &early = *early_p, &late = *late_p;
    auto eval = [](const Form& f, double t) {
        if (f.poly) {
            double r = 0.0;
            for (int i = f.n - 1; i >= 0; --i) r = r * t + f.c[i];
            return r;
        }
        double s = 0.0;
        for (int i = 0; i < f.n; ++i) s += f.c[i] * std::exp(-t / f.tau[i]);
        return s;
    };
    std::vector<double> tab((size_t)std::max<int64_t>(n, 1), 0.0);
    for (int64_t k = 0; k < n; ++k) {
        const double t = (double)k * (1.0 / 12.0);
        const double raw = t < switch_time ? eval(early, t) : eval(late, t);
        tab[(size_t)k] = (raw * irf_scale) / (raw * irf_scale + 1.0 - raw);
    }
    return tab;
}

// RSCM_MODE_FAST for OceanCarbon: fit  tab[lag] ~ sum_q c_q d_q^(lag - near)  for lag in [near, H) with
// d_q = exp(-rate_q / 12) and rate_q in {1/tau_i} u {1/tau_i + 1/tau_j} of the late form (the scaled
// response s raw / (s raw + 1 - raw) expands in powers of (1 - s) raw; first and second order carry it to
// ~1e-11).  Linear least squares for the amplitudes by Householder QR in long double (the columns are
// strongly correlated: condition numbers of 1e8 and more).  Returns the largest deviation from the table
// over the window, or a negative number if the parameters do not allow the recurrence.
static double ocean_fit_modes(const std::vector<double>& tab, int model, double switch_time, int64_t H, int32_t* near_out,
                              rscm::OceanModes* out)
{
    const OceanIrfForm *early, *late;
    ocean_forms(model, &early, &late);
    if (late->poly) return -1.0;
    const int64_t sw = (int64_t)std::ceil(switch_time * 12.0 - 1e-9);   // first lag of the late regime
    const int32_t near = sw <= 60 ? 60 : sw <= 120 ? 120 : 0;
    if (!near || H < 4 * (int64_t)near) return -1.0;   // short windows: the tiled convolution is cheap already
    std::vector<double> rates;
    auto add = [&](double r) {
        for (double x : rates)
            if (std::fabs(x - r) <= 1e-7 * (1.0 + std::fabs(r))) return;
        rates.push_back(r);
    };
    for (int i = 0; i < late->n; ++i) add(1.0 / late->tau[i]);
    for (int i = 0; i < late->n; ++i)
        for (int j = i; j < late->n; ++j) add(1.0 / late->tau[i] + 1.0 / late->tau[j]);
    std::sort(rates.begin(), rates.end());
    const int M = (int)rates.size();
    if (M > rscm::kOceanModes) return -1.0;
    const int64_t rows = std::min<int64_t>(H, 24000) - near;   // beyond 2000 years only the constant mode is left
    typedef long double ld;
    std::vector<ld> A((size_t)rows * M), b((size_t)rows);
    for (int64_t r = 0; r < rows; ++r) {
        for (int q = 0; q < M; ++q) A[(size_t)r * M + q] = std::exp(-(ld)rates[q] * (ld)r / 12.0L);
        b[(size_t)r] = tab[(size_t)(near + r)];
    }
    // Householder QR, applied to b on the fly
    std::vector<ld> R((size_t)M * M, 0.0L), v((size_t)rows);
    for (int k = 0; k < M; ++k) {
        ld norm = 0.0L;
        for (int64_t r = k; r < rows; ++r) norm += A[(size_t)r * M + k] * A[(size_t)r * M + k];
        norm = std::sqrt(norm);
        if (norm == 0.0L) return -1.0;
        const ld akk = A[(size_t)k * M + k];
        const ld alpha = akk > 0 ? -norm : norm;
        ld vnorm = 0.0L;
        for (int64_t r = k; r < rows; ++r) {
            v[(size_t)r] = A[(size_t)r * M + k] - (r == k ? alpha : 0.0L);
            vnorm += v[(size_t)r] * v[(size_t)r];
        }
        if (vnorm == 0.0L) return -1.0;
        for (int c = k; c < M; ++c) {
            ld dot = 0.0L;
            for (int64_t r = k; r < rows; ++r) dot += v[(size_t)r] * A[(size_t)r * M + c];
            const ld f = 2.0L * dot / vnorm;
            for (int64_t r = k; r < rows; ++r) A[(size_t)r * M + c] -= f * v[(size_t)r];
        }
        ld dot = 0.0L;
        for (int64_t r = k; r < rows; ++r) dot += v[(size_t)r] * b[(size_t)r];
        const ld f = 2.0L * dot / vnorm;
        for (int64_t r = k; r < rows; ++r) b[(size_t)r] -= f * v[(size_t)r];
        for (int c = k; c < M; ++c) R[(size_t)k * M + c] = A[(size_t)k * M + c];
    }
    std::vector<ld> x((size_t)M);
    for (int k = M - 1; k >= 0; --k) {
        ld acc = b[(size_t)k];
        for (int c = k + 1; c < M; ++c) acc -= R[(size_t)k * M + c] * x[(size_t)c];
        if (R[(size_t)k * M + k] == 0.0L) return -1.0;
        x[(size_t)k] = acc / R[(size_t)k * M + k];
    }
    *out = rscm::OceanModes{};
    for (int q = 0; q < M; ++q) {
        out->d[q] = (double)std::exp(-(ld)rates[q] / 12.0L);
        out->c[q] = (double)x[(size_t)q];
        out->e[q] = (double)std::exp(-(ld)rates[q] * (ld)(H - near) / 12.0L);
        if (!std::isfinite(out->c[q])) return -1.0;
    }
    out->n_modes = M;
    // rates ascend, so the exit weights e_q descend: the first n_exit modes still matter at lag H
    out->n_exit = 0;
    for (int q = 0; q < M; ++q)
        if (std::fabs(out->c[q]) * out->e[q] >= 1e-19) out->n_exit = q + 1;
    for (int q = out->n_exit; q < rscm::kOceanModes; ++q) out->e[q] = 0.0;  // the kernel applies a fixed number of exit terms
    // the fit as the device will evaluate it (double coefficients), against the whole window
    double worst = 0.0;
    for (int64_t lag = near; lag < H; ++lag) {
        ld acc = 0.0L;
        for (int q = 0; q < M; ++q) acc += (ld)out->c[q] * std::exp(-(ld)rates[q] * (ld)(lag - near) / 12.0L);
        worst = std::max(worst, std::fabs((double)acc - tab[(size_t)lag]));
    }
    *near_out = near;
    return worst;
}

constexpr double kOceanFitTolerance = 5e-10;  // largest deviation of the fitted far response accepted for RSCM_MODE_FAST

static_assert(rscm::kKindOzoneForcing == RSCM_KIND_OZONE_FORCING && rscm::kKindAerosolDirect == RSCM_KIND_AEROSOL_DIRECT &&
                  rscm::kKindAerosolIndirect == RSCM_KIND_AEROSOL_INDIRECT && rscm::kKindCh4Chemistry == RSCM_KIND_CH4_CHEMISTRY &&
                  rscm::kKindN2oChemistry == RSCM_KIND_N2O_CHEMISTRY && rscm::kKindCo2Budget == RSCM_KIND_CO2_BUDGET &&
                  rscm::kKindTerrestrialCarbon == RSCM_KIND_TERRESTRIAL_CARBON && rscm::kKindFourBoxOhu == RSCM_KIND_FOURBOX_OHU &&
                  rscm::kKindOspp == RSCM_KIND_OSPP && rscm::kKindCarbonCycle == RSCM_KIND_CARBON_CYCLE &&
                  rscm::kKindCo2Erf == RSCM_KIND_CO2_ERF && rscm::kKindAggregate == RSCM_KIND_AGGREGATE,
              "rscm_device.hpp and include/rscm_gpu.h disagree on a kind value");


namespace {

int refresh_schedule(rscm_ens* h)
{
    if (!h->schedule_dirty) return RSCM_OK;
    if (h->kind != RSCM_KIND_TWO_LAYER && h->kind != RSCM_KIND_COUPLED && h->kind != RSCM_KIND_CARBON_CYCLE) {  // no RK4 component
        h->schedule_dirty = false;
        return RSCM_OK;
    }
    int32_t bad = -1;
    if (h->kind == RSCM_KIND_CARBON_CYCLE) {
        h->nsub_tl.assign(h->bounds.size() - 2, 1);
    } else if (!rk4_schedule(h->bounds, h->h_tl, h->nsub_tl, &bad))
        return fail(RSCM_ERR_TIME_AXIS,
                    "TwoLayer RK4 step %.17g does not land on the end of model step %d "
                    "([%.17g, %.17g]) within 5e-3 (the reference panics in get_last_step)",
                    h->h_tl, bad, h->bounds[bad], h->bounds[bad + 1]);
    HIPCHK(hipMemcpyAsync(h->d_nsub_tl, h->nsub_tl.data(), h->nsub_tl.size() * sizeof(int32_t),
                          hipMemcpyHostToDevice, h->stream));
    if (h->kind == RSCM_KIND_COUPLED || h->kind == RSCM_KIND_CARBON_CYCLE) {
        if (!rk4_schedule(h->bounds, h->h_cc, h->nsub_cc, &bad))
            return fail(RSCM_ERR_TIME_AXIS,
                        "CarbonCycle RK4 step %.17g does not land on the end of model step %d "
                        "within 5e-3 (the reference panics in get_last_step)", h->h_cc, bad);
        HIPCHK(hipMemcpyAsync(h->d_nsub_cc, h->nsub_cc.data(), h->nsub_cc.size() * sizeof(int32_t),
                              hipMemcpyHostToDevice, h->stream));
    }
    HIPCHK(hipStreamSynchronize(h->stream));  // host vectors may be rebuilt afterwards
    h->schedule_dirty = false;
    return RSCM_OK;
}

// OceanCarbon: the rows that select the response table and the loop bounds must be uniform.
template <typename Row>
int configure_ocean(rscm_ens* h, int64_t n_check, Row row)
{
    static const int structural[] = {RSCM_OC_P_MODEL, RSCM_OC_P_IRF_SCALE, RSCM_OC_P_STEPS_PER_YEAR,
                                     RSCM_OC_P_MAX_HISTORY_MONTHS, RSCM_OC_P_IRF_SWITCH_TIME};
    for (int j : structural)
        for (int64_t i = 1; i < n_check; ++i)
            if (row(j, i) != row(j, 0))
                return fail(RSCM_ERR_INVALID, "OceanCarbon parameter row %d must be the same for every member", j);
    const double model = row(RSCM_OC_P_MODEL, 0), steps = row(RSCM_OC_P_STEPS_PER_YEAR, 0);
    const double max_hist = row(RSCM_OC_P_MAX_HISTORY_MONTHS, 0);
    if (model != 0.0 && model != 1.0 && model != 2.0)
        return fail(RSCM_ERR_INVALID, "OceanCarbon model must be 0 (3D-GFDL), 1 (2D-BERN) or 2 (HILDA), got %g", model);
    if (steps != 12.0) return fail(RSCM_ERR_INVALID, "OceanCarbon on the device supports steps_per_year = 12, got %g", steps);
    if (!(max_hist >= 0.0) || max_hist > 1e7 || max_hist != std::floor(max_hist))
        return fail(RSCM_ERR_INVALID, "max_history_months must be a non-negative integer, got %g", max_hist);
    const std::vector<double> tab = ocean_irf_table((int)model, row(RSCM_OC_P_IRF_SCALE, 0),
                                                    row(RSCM_OC_P_IRF_SWITCH_TIME, 0), (int64_t)max_hist);
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipFree(h->d_ocean_irf));
    h->d_ocean_irf = nullptr;
    HIPCHK(rscm::dev_malloc(&h->d_ocean_irf, tab.size() * sizeof(double)));
    HIPCHK(hipMemcpy(h->d_ocean_irf, tab.data(), tab.size() * sizeof(double), hipMemcpyHostToDevice));
    // The flux history is a ring: the convolution reads at most max_history_months pulses back, a tile writes
    // up to 48 new ones before it is done reading (ocean.hip), the O(T) recurrence reads the pulse that leaves
    // the window.  On a monthly axis that is 48 KB per member instead of 864 KB (108 000 pulses).
    const int64_t all_pulses = (int64_t)(h->T - 1) * 12;
    const int64_t ring = (((int64_t)max_hist + 60 + 11) / 12) * 12;
    const int64_t rows = std::min(all_pulses, ring);
    if (!h->d_ocean_hist || rows != h->ocean_hist_rows) {
        // the ring holds the pulses of the steps taken so far (internal state): a new length mid-run would
        // throw them away and leave the next convolution reading whatever the new allocation contains
        if (h->d_ocean_hist && h->time_index > 0)
            return fail(RSCM_ERR_STATE, "max_history_months changes the flux-history ring from %lld to %lld rows at time index %d: "
                        "rewind (rscm_ens_rewind / rscm_ens_set_time_index(0)) before changing it", (long long)h->ocean_hist_rows,
                        (long long)rows, h->time_index);
        HIPCHK(hipFree(h->d_ocean_hist));
        h->d_ocean_hist = nullptr;
        const hipError_t e = rscm::dev_malloc(&h->d_ocean_hist, (size_t)rows * h->N * sizeof(double));
        if (e != hipSuccess)
            return fail(e == hipErrorOutOfMemory ? RSCM_ERR_NOMEM : RSCM_ERR_DEVICE,
                        "flux history of %lld members x %lld months: %s", (long long)h->N, (long long)rows, hipGetErrorString(e));
        HIPCHK(hipMemset(h->d_ocean_hist, 0, (size_t)rows * h->N * sizeof(double)));
        h->ocean_hist_rows = rows;
    }
    if (!h->d_ocean_partial) {
        const hipError_t e = rscm::dev_malloc(&h->d_ocean_partial, (size_t)(rscm::kOceanSplitYears - 1) * 12 * h->N * sizeof(double));
        if (e != hipSuccess)
            return fail(e == hipErrorOutOfMemory ? RSCM_ERR_NOMEM : RSCM_ERR_DEVICE, "split-tile sums of %lld members: %s",
                        (long long)h->N, hipGetErrorString(e));
    }
    h->ocean_tile_base = -1;  // sums parked under another response table are void
    h->ocean_steps = 12;
    h->ocean_max_hist = (int64_t)max_hist;
    h->ocean_modes_at = -1;
    h->ocean_fit_error = ocean_fit_modes(tab, (int)model, row(RSCM_OC_P_IRF_SWITCH_TIME, 0), (int64_t)max_hist, &h->ocean_near, &h->ocean_modes);
    h->ocean_recur_ok = h->ocean_fit_error >= 0.0 && h->ocean_fit_error <= kOceanFitTolerance;
    if (h->ocean_recur_ok && !h->d_ocean_mode_state) {
        const hipError_t e = rscm::dev_malloc(&h->d_ocean_mode_state, (size_t)rscm::kOceanModes * h->N * sizeof(double));
        if (e != hipSuccess)
            return fail(e == hipErrorOutOfMemory ? RSCM_ERR_NOMEM : RSCM_ERR_DEVICE, "mode sums of %lld members: %s", (long long)h->N, hipGetErrorString(e));
    }
    if (h->ocean_recur_ok) {
        if (!h->d_ocean_mode_table) HIPCHK(rscm::dev_malloc(&h->d_ocean_mode_table, (size_t)3 * rscm::kOceanModes * sizeof(double)));
        double t3[3 * rscm::kOceanModes];
        for (int q = 0; q < rscm::kOceanModes; ++q) {
            t3[q] = h->ocean_modes.d[q];
            t3[rscm::kOceanModes + q] = h->ocean_modes.c[q];
            t3[2 * rscm::kOceanModes + q] = h->ocean_modes.e[q];
        }
        HIPCHK(hipMemcpy(h->d_ocean_mode_table, t3, sizeof t3, hipMemcpyHostToDevice));
    }
    h->ocean_ready = true;
    return RSCM_OK;
}

// ClimateUDEB: the structural parameters must be uniform over the ensemble (one set of geometry
// tables, one unrolled column length); `row(j, i)` reads parameter j of member i.
template <typename Row>
int configure_udeb(rscm_ens* h, int64_t n_check, Row row)
{
    static const int structural[] = {RSCM_UD_P_N_LAYERS, RSCM_UD_P_MIXED_LAYER_DEPTH, RSCM_UD_P_LAYER_THICKNESS,
                                     RSCM_UD_P_DEPTH_DEPENDENT_AREA, RSCM_UD_P_LAND_HC_ENABLED,
                                     RSCM_UD_P_EFFICACY_APPLY, RSCM_UD_P_OCEAN_TEMP_PROFILE, RSCM_UD_P_STEPS_PER_YEAR,
                                     RSCM_UD_P_FEEDBACK_CUMT_PERIOD};
    for (int j : structural)
        for (int64_t i = 1; i < n_check; ++i)
            if (row(j, i) != row(j, 0))
                return fail(RSCM_ERR_INVALID, "ClimateUDEB parameter row %d must be the same for every member", j);
    const double nl = row(RSCM_UD_P_N_LAYERS, 0), steps = row(RSCM_UD_P_STEPS_PER_YEAR, 0);
    // mod.rs:162-165: "invalid n_layers: must be >= 2".  Up to 64 layers a member's columns stay in registers + LDS (20 / 30 / 40 / 50
    // with the count compiled in, the others in the next capacity's runtime-count instance); more run the columns-in-HBM kernel
    // (csrc/udeb_any_body.hpp).  The upper bound is this library's.
    if (nl != std::floor(nl) || nl < 2.0) return fail(RSCM_ERR_INVALID, "invalid n_layers: must be >= 2, got %g", nl);
    if (nl > 4096.0) return fail(RSCM_ERR_INVALID, "n_layers = %g: the device path takes at most 4096 ocean layers", nl);
    if (h->udeb_ready && (int32_t)nl != h->udeb_n_layers && h->time_index > 0)
        return fail(RSCM_ERR_STATE, "n_layers changes the ocean columns (internal state) at time index %d: rewind first", h->time_index);
    if (row(RSCM_UD_P_OCEAN_TEMP_PROFILE, 0) != 2.0)
        return fail(RSCM_ERR_INVALID, "ClimateUDEB on the device supports ocean_temp_profile = 2 (CMIP5)");
    if (!(steps >= 1.0) || steps > 1000.0 || steps != std::floor(steps))
        return fail(RSCM_ERR_INVALID, "steps_per_year must be a positive integer, got %g", steps);
    const double eff = row(RSCM_UD_P_EFFICACY_APPLY, 0);
    if (eff != 0.0 && eff != 1.0 && eff != 2.0) return fail(RSCM_ERR_INVALID, "efficacy_apply must be 0, 1 or 2");
    h->udeb_n_layers = (int32_t)nl;
    h->udeb_steps = (int32_t)steps;
    h->udeb_land_hc = row(RSCM_UD_P_LAND_HC_ENABLED, 0) != 0.0 ? 1 : 0;
    h->udeb_efficacy = (int32_t)eff;
    const std::vector<double> t = rscm::udeb_tables(h->udeb_n_layers, row(RSCM_UD_P_MIXED_LAYER_DEPTH, 0),
                                                    row(RSCM_UD_P_LAYER_THICKNESS, 0), row(RSCM_UD_P_DEPTH_DEPENDENT_AREA, 0));
    h->udeb_tables = t;
    {   // the backwards walk of adjusted_ecs() (mod.rs:302-331) over the time axis, once per year
        const double period = row(RSCM_UD_P_FEEDBACK_CUMT_PERIOD, 0);
        std::vector<int32_t> kfull((size_t)h->T, 0);
        std::vector<double> partw((size_t)h->T, 0.0);
        for (int32_t n = 0; n < h->T; ++n) {
            double years_remaining = period;
            int32_t kf = n;
            double pw = 0.0;
            for (int32_t k = n - 1; k >= 0; --k) {
                if (years_remaining <= 0.0) break;
                const double dt = h->bounds[k + 1] - h->bounds[k];
                if (dt <= years_remaining) {
                    kf = k;
                    years_remaining -= dt;
                } else {
                    pw = years_remaining / dt;
                    years_remaining = 0.0;
                }
            }
            kfull[n] = kf;
            partw[n] = pw;
        }
        if (!h->d_win_kfull) HIPCHK(rscm::dev_malloc(&h->d_win_kfull, (size_t)h->T * sizeof(int32_t)));
        if (!h->d_win_partw) HIPCHK(rscm::dev_malloc(&h->d_win_partw, (size_t)h->T * sizeof(double)));
        HIPCHK(hipMemcpyAsync(h->d_win_kfull, kfull.data(), kfull.size() * sizeof(int32_t), hipMemcpyHostToDevice, h->stream));
        HIPCHK(hipMemcpyAsync(h->d_win_partw, partw.data(), partw.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
    }
    {   // the columns: room for 50 layers at least (any of the unrolled kernels), more if asked for
        const int32_t want = std::max(rscm::kUdebMaxOnChipLayers, h->udeb_n_layers);
        if (want > h->udeb_ocean_layers) {
            (void)hipFree(h->d_ocean);
            h->d_ocean = nullptr;
            h->udeb_ocean_layers = 0;
            const hipError_t e = rscm::dev_malloc(&h->d_ocean, (size_t)2 * want * h->N * sizeof(double));
            if (e != hipSuccess)
                return fail(e == hipErrorOutOfMemory ? RSCM_ERR_NOMEM : RSCM_ERR_DEVICE, "ocean columns of %d layers x %lld members: %s", want,
                            (long long)h->N, hipGetErrorString(e));
            h->udeb_ocean_layers = want;
        }
    }
    if (!rscm::udeb_layers_fixed(h->udeb_n_layers)) {   // the columns-in-HBM kernel (more than 64 layers; up to 64 only on request, as the
        // yardstick of the runtime-count kernels -- rscm_gpu_set_udeb_variant(3)): its c' array and its table live in device memory
        if (h->udeb_n_layers > h->udeb_work_layers) {
            (void)hipFree(h->d_udeb_work);
            (void)hipFree(h->d_udeb_tables);
            h->d_udeb_work = h->d_udeb_tables = nullptr;
            h->udeb_work_layers = 0;
            const hipError_t e = rscm::dev_malloc(&h->d_udeb_work, (size_t)h->udeb_n_layers * h->N * sizeof(double));
            if (e != hipSuccess)
                return fail(e == hipErrorOutOfMemory ? RSCM_ERR_NOMEM : RSCM_ERR_DEVICE, "work array of %d layers x %lld members: %s",
                            h->udeb_n_layers, (long long)h->N, hipGetErrorString(e));
            // (room for the largest on-chip capacity: the unrolled sweeps request the table rows of their whole capacity, and the rows past
            // the layer count must read as zeros -- they are what makes those rows of the column exact no-ops)
            HIPCHK(rscm::dev_malloc(&h->d_udeb_tables, (size_t)6 * std::max(h->udeb_n_layers, (int32_t)rscm::kUdebDevTableRows) * sizeof(double)));
            h->udeb_work_layers = h->udeb_n_layers;
        }
        HIPCHK(hipMemsetAsync(h->d_udeb_tables, 0, (size_t)6 * std::max(h->udeb_work_layers, (int32_t)rscm::kUdebDevTableRows) * sizeof(double), h->stream));
        HIPCHK(hipMemcpyAsync(h->d_udeb_tables, h->udeb_tables.data(), h->udeb_tables.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
    }
    h->udeb_ready = true;
    return RSCM_OK;
}

// ---- windowed series -----------------------------------------------------------------------
// Back to the start of the axis: window at row 0, the saved initial rows put back, the rest NaN if
// asked (a fresh collection) or if a consumer may read rows this producer has not written yet.
int window_reset(rscm_ens* h, bool clear)
{
    if (!h->windowed) return RSCM_OK;
    h->win0 = 0;
    if (h->row0_saved)
        HIPCHK(rscm::launch_scatter_rows(h->d_series, h->N, h->rows, 0, nullptr, h->V - 1, h->d_row0, 1, 0, h->stream));
    if (clear || h->read_ahead)
        HIPCHK(rscm::launch_fill_rows(h->d_series, h->N, h->rows, h->V - 1, 1, std::numeric_limits<double>::quiet_NaN(), h->stream));
    return RSCM_OK;
}

// Move the window forward to start at new_win0, keeping the rows both windows share.
int window_slide(rscm_ens* h, int32_t new_win0)
{
    const int32_t shift = new_win0 - h->win0;
    if (!h->windowed || shift <= 0) return RSCM_OK;
    const int32_t cnt = std::max(0, h->rows - shift);  // rows still inside the new window
    if (h->defer && shift >= cnt) {
        // a lock-step run: the move is issued with every other handle's at the end of the model step; until then this handle's
        // base address and win0 stay what they are (consumers later in the step read the rows where they still lie)
        for (const auto& pending : h->defer->new_win0)
            if (pending.first == h) return fail(RSCM_ERR_STATE, "two window slides of one handle in one lock-step flush");
        if (cnt > 0) {
            rscm::WindowOp op{};
            op.buf = h->d_series; op.N = h->N; op.R = h->rows; op.n_vars = h->V - 1; op.kind = 0; op.shift = shift; op.keep = cnt;
            h->defer->first.push_back(op);
        }
        if (h->read_ahead || cnt == 0) {
            rscm::WindowOp op{};
            op.buf = h->d_series; op.N = h->N; op.R = h->rows; op.n_vars = h->V - 1; op.kind = 2; op.fill_from = cnt;
            h->defer->second.push_back(op);
        }
        h->defer->new_win0.emplace_back(h, new_win0);
        return RSCM_OK;
    }
    HIPCHK(rscm::launch_slide_rows(h->d_series, h->N, h->rows, h->V - 1, shift, cnt, h->stream));
    if (h->read_ahead || cnt == 0)
        HIPCHK(rscm::launch_fill_rows(h->d_series, h->N, h->rows, h->V - 1, cnt, std::numeric_limits<double>::quiet_NaN(), h->stream));
    h->win0 = new_win0;
    return RSCM_OK;
}

// Re-position the window for a stepper that is put at time index k from outside (checkpoint restore):
// the contents are the caller's to fill in (rscm_ens_set_state); everything is NaN until then.
int window_seek(rscm_ens* h, int32_t k)
{
    if (!h->windowed) return RSCM_OK;
    if (k == 0) return window_reset(h, true);
    const int32_t w0 = std::max(0, k - h->keep_rows() + 1);
    if (k >= h->win0 && k < h->win0 + h->rows - 1 && w0 >= h->win0) return window_slide(h, w0);
    HIPCHK(rscm::launch_fill_rows(h->d_series, h->N, h->rows, h->V - 1, 0, std::numeric_limits<double>::quiet_NaN(), h->stream));
    h->win0 = w0;
    return RSCM_OK;
}

// every out_stride-th row into the output store
int window_store_row(rscm_ens* h, int32_t t)
{
    if (!h->windowed || h->n_out == 0 || t % h->out_stride != 0) return RSCM_OK;
    if (h->defer) {
        rscm::WindowOp op{};
        op.buf = h->d_series; op.N = h->N; op.R = h->rows; op.n_vars = h->V - 1; op.kind = 1;
        op.vars = h->d_out_vars; op.dst = h->d_out; op.src_row = t - h->win0; op.n_out = h->n_out; op.dst_rows = h->out_rows;
        op.dst_row = t / h->out_stride;
        h->defer->first.push_back(op);
        return RSCM_OK;
    }
    HIPCHK(rscm::launch_gather_rows(h->d_series, h->N, h->rows, t - h->win0, h->d_out_vars, h->n_out, h->d_out, h->out_rows,
                                    t / h->out_stride, h->stream));
    return RSCM_OK;
}

}  // namespace

int window_flush(WindowDeferral* d, hipStream_t stream)
{
    for (std::vector<rscm::WindowOp>* list : {&d->first, &d->second}) {
        for (size_t at = 0; at < list->size(); at += rscm::kMaxWindowOps) {
            const size_t n = std::min(list->size() - at, (size_t)rscm::kMaxWindowOps);
            rscm::WindowBatch batch;
            memset((void*)&batch, 0, sizeof batch);
            memcpy((void*)batch.ops, list->data() + at, n * sizeof(rscm::WindowOp));
            HIPCHK(rscm::launch_window_batch(batch, (int32_t)n, stream));
        }
        list->clear();
    }
    for (const auto& w : d->new_win0) w.first->win0 = w.second;
    d->new_win0.clear();
    return RSCM_OK;
}

extern "C" {

int rscm_gpu_abi_version(void) { return RSCM_GPU_ABI_VERSION; }
int rscm_gpu_abi_minor(void) { return RSCM_GPU_ABI_MINOR; }

const char* rscm_gpu_last_error(void) { return g_last_error.c_str(); }

int rscm_gpu_device_count(int32_t* out)
{
    GUARD_BEGIN
    if (!out) return fail(RSCM_ERR_INVALID, "out is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *out = 0;
        return fail(RSCM_ERR_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e));
    }
    *out = n;
    return RSCM_OK;
    GUARD_END
}

int rscm_gpu_mem_info(int32_t device_id, uint64_t* free_bytes, uint64_t* total_bytes)
{
    GUARD_BEGIN
    if (!free_bytes || !total_bytes) return fail(RSCM_ERR_INVALID, "output pointer is NULL");
    hipError_t e = hipSetDevice(device_id);
    if (e != hipSuccess) return fail(RSCM_ERR_DEVICE, "hipSetDevice(%d): %s", device_id, hipGetErrorString(e));
    size_t f = 0, t = 0;
    e = hipMemGetInfo(&f, &t);
    if (e != hipSuccess) return fail(RSCM_ERR_DEVICE, "hipMemGetInfo: %s", hipGetErrorString(e));
    *free_bytes = (uint64_t)f;
    *total_bytes = (uint64_t)t;
    return RSCM_OK;
    GUARD_END
}

int rscm_ens_create(int32_t kind, int64_t n_members, int32_t n_times, const double* time_bounds,
                    int32_t device_id, rscm_ens** out)
{
    return rscm_ens_create_ex(kind, n_members, n_times, time_bounds, device_id, 0, out);
}

int rscm_ens_create_ex(int32_t kind, int64_t n_members, int32_t n_times, const double* time_bounds,
                       int32_t device_id, uint32_t flags, rscm_ens** out)
{
    return rscm_ens_create_windowed(kind, n_members, n_times, time_bounds, device_id, flags, 16, 0, -1, nullptr, out);
}

int rscm_ens_create_windowed(int32_t kind, int64_t n_members, int32_t n_times, const double* time_bounds,
                             int32_t device_id, uint32_t flags, int32_t window_rows, int32_t out_stride,
                             int32_t n_out_vars, const int32_t* out_vars, rscm_ens** out)
{
    GUARD_BEGIN
    if (flags & ~(uint32_t)(RSCM_FLAG_NO_SERIES | RSCM_FLAG_WINDOWED)) return fail(RSCM_ERR_INVALID, "unknown flags 0x%x", flags);
    if ((flags & RSCM_FLAG_NO_SERIES) && kind != RSCM_KIND_TWO_LAYER)
        return fail(RSCM_ERR_INVALID, "RSCM_FLAG_NO_SERIES is only available for the two-layer kind");
    if ((flags & RSCM_FLAG_NO_SERIES) && (flags & RSCM_FLAG_WINDOWED))
        return fail(RSCM_ERR_INVALID, "RSCM_FLAG_NO_SERIES and RSCM_FLAG_WINDOWED exclude each other");
    const bool windowed = (flags & RSCM_FLAG_WINDOWED) != 0;
    if (windowed && (window_rows < 4 || out_stride < 0 || (n_out_vars > 0 && !out_vars)))
        return fail(RSCM_ERR_INVALID, "a windowed ensemble needs window_rows >= 4, out_stride >= 0 and a variable list if n_out_vars > 0");
    if (!out) return fail(RSCM_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (kind < RSCM_KIND_TWO_LAYER || kind > RSCM_KIND_AGGREGATE)
        return fail(RSCM_ERR_INVALID, "unknown kind %d", kind);
    if (n_members < 1) return fail(RSCM_ERR_INVALID, "n_members must be >= 1, got %lld", (long long)n_members);
    if (n_times < 2) return fail(RSCM_ERR_INVALID, "n_times must be >= 2 (TimeAxis::from_values asserts len >= 2)");
    if (!time_bounds) return fail(RSCM_ERR_INVALID, "time_bounds is NULL");
    for (int32_t i = 0; i < n_times; ++i)
        if (!(time_bounds[i + 1] > time_bounds[i]))  // timeseries.rs:47-52 assert!(is_monotonic)
            return fail(RSCM_ERR_INVALID, "time_bounds must be strictly increasing (index %d)", i);
    if ((uint64_t)n_members > (uint64_t)std::numeric_limits<int32_t>::max() * 256ull)
        return fail(RSCM_ERR_INVALID, "n_members too large for one launch grid");

    rscm_ens* h = new rscm_ens();
    h->kind = kind;
    h->N = n_members;
    h->T = n_times;
    h->rows = (flags & RSCM_FLAG_NO_SERIES) ? 1 : n_times;
    if (windowed && window_rows < n_times) {  // a window as long as the axis is plain full storage
        h->windowed = true;
        h->rows = window_rows;
    }
    h->device = device_id;
    static const int32_t kP[] = {RSCM_TL_NPARAMS, RSCM_CP_NPARAMS, RSCM_UD_NPARAMS, RSCM_GH_NPARAMS,
                                 RSCM_OZ_NPARAMS, RSCM_AD_NPARAMS, RSCM_AI_NPARAMS, RSCM_CH4_NPARAMS,
                                 RSCM_N2O_NPARAMS, RSCM_CB_NPARAMS, RSCM_TC_NPARAMS, RSCM_OC_NPARAMS,
                                 RSCM_HC_NPARAMS, RSCM_FB_NPARAMS, RSCM_SP_NPARAMS, RSCM_CC_NPARAMS,
                                 RSCM_CE_NPARAMS, RSCM_AG_NPARAMS};
    static const int32_t kV[] = {3, 8, 8, 4, 4, 5, 2, 3, 3, 4, 6, 4, RSCM_HC_NSPECIES + 5, 5, 2, 4, 2, 2};  // variable ids incl. the input block 0
    static const int32_t kInputs[] = {1, 1, 1, 3, RSCM_OZ_NINPUTS, RSCM_AD_NINPUTS, RSCM_AI_NINPUTS,
                                      RSCM_CH4_NINPUTS, RSCM_N2O_NINPUTS, RSCM_CB_NINPUTS, RSCM_TC_NINPUTS,
                                      RSCM_OC_NINPUTS, RSCM_HC_NINPUTS, RSCM_FB_NINPUTS, RSCM_SP_NINPUTS,
                                      RSCM_CC_NINPUTS, RSCM_CE_NINPUTS, RSCM_AG_NINPUTS};
    h->P = kP[kind];
    h->V = kV[kind];
    h->n_inputs = kInputs[kind];
    h->bounds.assign(time_bounds, time_bounds + n_times + 1);
    h->initial_set.assign(h->V, 0);
    h->out_slot.assign(h->V, -1);
    if (kind == RSCM_KIND_CH4_CHEMISTRY) h->lookback = 1;   // previous()
    if (kind == RSCM_KIND_N2O_CHEMISTRY) h->lookback = 2;   // at_offset(-(strat_delay + 1)), refreshed by set_params
    if (h->windowed && out_stride > 0) {
        h->out_stride = out_stride;
        h->out_rows = (n_times - 1) / out_stride + 1;
        if (n_out_vars < 0)
            for (int32_t v = 1; v < h->V; ++v) h->out_vars.push_back(v);
        else
            for (int32_t k = 0; k < n_out_vars; ++k) {
                if (out_vars[k] < 1 || out_vars[k] >= h->V || h->out_slot[out_vars[k]] >= 0) {
                    delete h;
                    return fail(RSCM_ERR_INVALID, "output variable %d: not a stored variable of this kind, or listed twice", out_vars[k]);
                }
                h->out_slot[out_vars[k]] = k;
                h->out_vars.push_back(out_vars[k]);
            }
        h->n_out = (int32_t)h->out_vars.size();
        for (int32_t k = 0; k < h->n_out; ++k) h->out_slot[h->out_vars[k]] = k;
    }

    auto cleanup = [&](int rc) {
        rscm_ens_destroy(h);
        return rc;
    };
    hipError_t e = hipSetDevice(device_id);
    if (e != hipSuccess) {
        delete h;
        return fail(RSCM_ERR_DEVICE, "hipSetDevice(%d): %s", device_id, hipGetErrorString(e));
    }
#define CK(expr)                                                                              \
    do {                                                                                      \
        hipError_t e2_ = (expr);                                                              \
        if (e2_ != hipSuccess)                                                                \
            return cleanup(fail(e2_ == hipErrorOutOfMemory ? RSCM_ERR_NOMEM : RSCM_ERR_DEVICE, \
                                "%s failed: %s", #expr, hipGetErrorString(e2_)));             \
    } while (0)
    CK(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    h->own_stream = true;
    CK(hipEventCreate(&h->ev0));
    CK(hipEventCreate(&h->ev1));
    const size_t series_elems = (size_t)(h->V - 1) * (size_t)h->rows * (size_t)h->N;
    CK(rscm::dev_malloc(&h->d_params, (size_t)h->P * h->N * sizeof(double)));
    CK(rscm::dev_malloc(&h->d_series, series_elems * sizeof(double)));
    CK(rscm::dev_malloc(&h->d_status, (size_t)h->N));
    if (h->windowed) {
        CK(rscm::dev_malloc(&h->d_row0, (size_t)(h->V - 1) * h->N * sizeof(double)));
        if (h->n_out > 0) {
            const size_t out_elems = (size_t)h->n_out * h->out_rows * h->N;
            CK(rscm::dev_malloc(&h->d_out, out_elems * sizeof(double)));
            CK(rscm::dev_malloc(&h->d_out_vars, (size_t)h->n_out * sizeof(int32_t)));
            CK(hipMemcpy(h->d_out_vars, h->out_vars.data(), (size_t)h->n_out * sizeof(int32_t), hipMemcpyHostToDevice));
            CK(rscm::launch_fill(h->d_out, (int64_t)out_elems, std::numeric_limits<double>::quiet_NaN(), h->stream));
        }
    }
    CK(rscm::dev_malloc(&h->d_nsub_tl, (size_t)(h->T - 1) * sizeof(int32_t)));
    CK(rscm::dev_malloc(&h->d_nsub_cc, (size_t)(h->T - 1) * sizeof(int32_t)));
    CK(rscm::dev_malloc(&h->d_partial, 4 * 1024 * sizeof(double)));
    CK(rscm::dev_malloc(&h->d_out4, 4 * sizeof(double)));
    if (kind == RSCM_KIND_UDEB) {
        CK(rscm::dev_malloc(&h->d_scal, (size_t)rscm::kUdebScalars * h->N * sizeof(double)));
        CK(rscm::dev_malloc(&h->d_hist, (size_t)h->T * h->N * sizeof(double)));
    }
    if (kind == RSCM_KIND_UDEB || kind == RSCM_KIND_N2O_CHEMISTRY || kind == RSCM_KIND_CO2_BUDGET ||
        kind == RSCM_KIND_TERRESTRIAL_CARBON || kind == RSCM_KIND_OCEAN_CARBON ||
        kind == RSCM_KIND_HALOCARBON) {  // kinds that use the step length
        CK(rscm::dev_malloc(&h->d_bounds, (size_t)(h->T + 1) * sizeof(double)));
        CK(hipMemcpyAsync(h->d_bounds, h->bounds.data(), (size_t)(h->T + 1) * sizeof(double), hipMemcpyHostToDevice, h->stream));
    }
    if (kind == RSCM_KIND_AGGREGATE) {  // contributors that are neither linked nor set stay NaN = skipped
        CK(rscm::dev_malloc(&h->d_forcing, (size_t)h->n_inputs * h->T * sizeof(double)));
        CK(rscm::launch_fill(h->d_forcing, (int64_t)h->n_inputs * h->T, std::numeric_limits<double>::quiet_NaN(), h->stream));
        h->n_scen = 1;
        h->forcing_set = true;
    }
    CK(hipMemsetAsync(h->d_status, 0, (size_t)h->N, h->stream));
    // never-written entries are NaN (builder.rs:772-780)
    CK(rscm::launch_fill(h->d_series, (int64_t)series_elems, std::numeric_limits<double>::quiet_NaN(),
                         h->stream));
    CK(hipStreamSynchronize(h->stream));
#undef CK
    *out = h;
    return RSCM_OK;
    GUARD_END
}

int rscm_ens_destroy(rscm_ens* h)
{
    if (!h) return RSCM_OK;
    if (h->link_refs > 0)
        return fail(RSCM_ERR_STATE, "%d linked input(s) of other ensembles still read this one's series: destroy or unlink them first", h->link_refs);
    for (auto& l : h->links)
        if (l.src) {
            --l.src->link_refs;
            l.src = nullptr;
        }
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    (void)hipFree(h->d_params);
    (void)hipFree(h->d_series);
    (void)hipFree(h->d_forcing);
    (void)hipFree(h->d_scen);
    (void)hipFree(h->d_status);
    (void)hipFree(h->d_nsub_tl);
    (void)hipFree(h->d_nsub_cc);
    (void)hipFree(h->d_ghg_tables);
    (void)hipFree(h->d_ocean_hist);
    (void)hipFree(h->d_ocean_irf);
    (void)hipFree(h->d_ocean_partial);
    (void)hipFree(h->d_ocean_mode_state);
    (void)hipFree(h->d_ocean_mode_table);
    (void)hipFree(h->d_ocean);
    (void)hipFree(h->d_udeb_work);
    (void)hipFree(h->d_udeb_tables);
    (void)hipFree(h->d_derived);
    if (h->split_stream) (void)hipStreamDestroy(h->split_stream);
    if (h->split_fork) (void)hipEventDestroy(h->split_fork);
    if (h->split_join) (void)hipEventDestroy(h->split_join);
    (void)hipFree(h->d_scal);
    (void)hipFree(h->d_hist);
    (void)hipFree(h->d_tables);
    (void)hipFree(h->d_bounds);
    (void)hipFree(h->d_win_kfull);
    (void)hipFree(h->d_win_partw);
    (void)hipFree(h->d_partial);
    (void)hipFree(h->d_out4);
    (void)hipFree(h->d_loglik);
    (void)hipFree(h->d_obs);
    if (h->plan) {
        (void)hipFree(h->plan->d_ops);
        if (h->plan->staging) (void)hipHostFree(h->plan->staging);
        delete h->plan;
        h->plan = nullptr;
    }
    (void)hipFree(h->d_row0);
    (void)hipFree(h->d_out);
    (void)hipFree(h->d_out_vars);
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    if (h->own_stream && h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return RSCM_OK;
}


int rscm_ens_n_params(const rscm_ens* h, int32_t* out) { NEED(h); *out = h->P; return RSCM_OK; }
int rscm_ens_n_vars(const rscm_ens* h, int32_t* out) { NEED(h); *out = h->V; return RSCM_OK; }
int rscm_ens_n_inputs(const rscm_ens* h, int32_t* out) { NEED(h); *out = h->n_inputs; return RSCM_OK; }
int rscm_ens_n_members(const rscm_ens* h, int64_t* out) { NEED(h); *out = h->N; return RSCM_OK; }
int rscm_ens_n_times(const rscm_ens* h, int32_t* out) { NEED(h); *out = h->T; return RSCM_OK; }
int rscm_ens_time_index(const rscm_ens* h, int32_t* out) { NEED(h); *out = h->time_index; return RSCM_OK; }

int rscm_ens_set_mode(rscm_ens* h, int32_t mode)
{
    NEED(h);
    if (mode != RSCM_MODE_EXACT && mode != RSCM_MODE_FAST) return fail(RSCM_ERR_INVALID, "unknown mode %d", mode);
    if (mode != h->mode) {  // sums parked / carried by the other arithmetic cannot be resumed
        h->ocean_tile_base = -1;
        h->ocean_modes_at = -1;
    }
    h->mode = mode;
    return RSCM_OK;
}

int rscm_ens_set_step_size(rscm_ens* h, int32_t component, double step)
{
    NEED(h);
    if (!(step > 0.0) || !std::isfinite(step)) return fail(RSCM_ERR_INVALID, "step must be positive and finite");
    if (component == RSCM_COMP_TWO_LAYER && h->kind != RSCM_KIND_CARBON_CYCLE)
        h->h_tl = step;
    else if (component == RSCM_COMP_CARBON_CYCLE && (h->kind == RSCM_KIND_COUPLED || h->kind == RSCM_KIND_CARBON_CYCLE))
        h->h_cc = step;
    else
        return fail(RSCM_ERR_INVALID, "component %d not part of this model kind", component);
    h->schedule_dirty = true;
    return RSCM_OK;
}

int rscm_gpu_stream_create(int32_t device_id, void** out_stream)
{
    GUARD_BEGIN
    if (!out_stream) return fail(RSCM_ERR_INVALID, "out_stream is NULL");
    HIPCHK(hipSetDevice(device_id));
    hipStream_t s = nullptr;
    HIPCHK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *out_stream = (void*)s;
    return RSCM_OK;
    GUARD_END
}

int rscm_gpu_stream_destroy(int32_t device_id, void* hip_stream)
{
    GUARD_BEGIN
    if (!hip_stream) return RSCM_OK;
    HIPCHK(hipSetDevice(device_id));
    HIPCHK(hipStreamSynchronize((hipStream_t)hip_stream));
    HIPCHK(hipStreamDestroy((hipStream_t)hip_stream));
    return RSCM_OK;
    GUARD_END
}

int rscm_ens_set_stream(rscm_ens* h, void* hip_stream)
{
    GUARD_BEGIN
    NEED(h);
    if (int rc = set_device(h)) return rc;
    HIPCHK(hipStreamSynchronize(h->stream));
    if (h->own_stream) {
        HIPCHK(hipStreamDestroy(h->stream));
        h->own_stream = false;
    }
    if (hip_stream) {
        h->stream = (hipStream_t)hip_stream;
    } else {
        HIPCHK(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
        h->own_stream = true;
    }
    return RSCM_OK;
    GUARD_END
}

int rscm_ens_set_params(rscm_ens* h, const double* soa)
{
    GUARD_BEGIN
    NEED(h);
    if (!soa) return fail(RSCM_ERR_INVALID, "params is NULL");
    if (int rc = set_device(h)) return rc;
    if (h->kind == RSCM_KIND_UDEB)
        if (int rc = configure_udeb(h, h->N, [&](int j, int64_t i) { return soa[(size_t)j * h->N + i]; })) return rc;
    if (h->kind == RSCM_KIND_OCEAN_CARBON)
        if (int rc = configure_ocean(h, h->N, [&](int j, int64_t i) { return soa[(size_t)j * h->N + i]; })) return rc;
    if (h->kind == RSCM_KIND_GHG_FORCING) {  // one forcing method per ensemble (one kernel instance)
        const double m = soa[(size_t)RSCM_GH_P_METHOD * h->N];
        if (m != 0.0 && m != 1.0) return fail(RSCM_ERR_INVALID, "GhgForcing method must be 0 (Ipcctar) or 1 (Olbl), got %g", m);
        for (int64_t i = 1; i < h->N; ++i)
            if (soa[(size_t)RSCM_GH_P_METHOD * h->N + i] != m)
                return fail(RSCM_ERR_INVALID, "GhgForcing parameter row %d (method) must be the same for every member", RSCM_GH_P_METHOD);
        h->ghg_method = (int32_t)m;
    }
    if (h->kind == RSCM_KIND_N2O_CHEMISTRY) {  // rows the stratospheric delay looks back (n2o.rs:203-218)
        double d = 1.0;
        for (int64_t i = 0; i < h->N; ++i) d = std::max(d, soa[(size_t)4 * h->N + i]);
        const int32_t lookback = (int32_t)std::min(d, 1e6) + 1;
        // a window that has already slid kept only keep_rows() of the OLD look-back: the rows a longer delay
        // reads are gone (the body would read before the window's first row)
        if (h->windowed && h->win0 > 0 && std::max(0, h->time_index - lookback) < h->win0 && lookback > h->lookback)
            return fail(RSCM_ERR_STATE, "a stratospheric delay that looks %d rows back needs row %d, but the window already starts at row %d: "
                        "rewind (rscm_ens_rewind) or restore a checkpoint before raising the delay", lookback,
                        std::max(0, h->time_index - lookback), h->win0);
        h->lookback = lookback;
    }
    HIPCHK(hipMemcpyAsync(h->d_params, soa, (size_t)h->P * h->N * sizeof(double), hipMemcpyHostToDevice, h->stream));
    {   // rows that hold the same bits for every member
        uint64_t uni = 0;
        for (int32_t j = 0; j < h->P && j < 64; ++j) {
            const double* row = soa + (size_t)j * h->N;
            bool same = true;
            for (int64_t i = 1; i < h->N && same; ++i) same = memcmp(&row[i], &row[0], sizeof(double)) == 0;
            if (same) uni |= 1ull << j;
        }
        h->uniform_rows = h->params_exposed ? 0 : uni;  // a caller holding the device pointer may rewrite any row
        h->derived_dirty = true;
    }
    HIPCHK(hipStreamSynchronize(h->stream));
    h->params_set = true;
    return RSCM_OK;
    GUARD_END
}

int rscm_ens_set_params_aos(rscm_ens* h, const double* aos)
{
    GUARD_BEGIN
    NEED(h);
    if (!aos) return fail(RSCM_ERR_INVALID, "params is NULL");
    std::vector<double> soa((size_t)h->P * h->N);
    for (int64_t i = 0; i < h->N; ++i)
        for (int32_t j = 0; j < h->P; ++j) soa[(size_t)j * h->N + i] = aos[(size_t)i * h->P + j];
    return rscm_ens_set_params(h, soa.data());
    GUARD_END
}

int rscm_ens_set_forcing(rscm_ens* h, int32_t var_id, int32_t n_scen, const double* series,
                         const int32_t* scenario_of_member, int32_t source)
{
    GUARD_BEGIN
    NEED(h);
    if (var_id != 0) return fail(RSCM_ERR_INVALID, "variable %d is not the shared input of this kind", var_id);
    if (n_scen < 1 || !series) return fail(RSCM_ERR_INVALID, "need n_scen >= 1 and a series pointer");
    if (source != RSCM_SRC_EXOGENOUS && source != RSCM_SRC_UPSTREAM) return fail(RSCM_ERR_INVALID, "unknown source %d", source);
    if (h->kind == RSCM_KIND_COUPLED && source != RSCM_SRC_EXOGENOUS)
        return fail(RSCM_ERR_INVALID, "emissions of the coupled chain are exogenous (no component produces them)");
    if (h->kind == RSCM_KIND_UDEB && source != RSCM_SRC_EXOGENOUS)
        return fail(RSCM_ERR_INVALID, "ClimateUDEB reads its forcing as an exogenous series (at_start / at_end)");
    if (h->kind >= RSCM_KIND_GHG_FORCING && source != RSCM_SRC_EXOGENOUS)
        return fail(RSCM_ERR_INVALID, "the stateless forcing kinds read their inputs as exogenous series");
    if (scenario_of_member)
        for (int64_t i = 0; i < h->N; ++i)
            if (scenario_of_member[i] < 0 || scenario_of_member[i] >= n_scen)
                return fail(RSCM_ERR_INVALID, "scenario_of_member[%lld] = %d out of range [0, %d)", (long long)i,
                            scenario_of_member[i], n_scen);
    if (int rc = set_device(h)) return rc;
    HIPCHK(hipStreamSynchronize(h->stream));
    if (n_scen != h->n_scen || !h->d_forcing) {
        HIPCHK(hipFree(h->d_forcing));
        h->d_forcing = nullptr;
        HIPCHK(rscm::dev_malloc(&h->d_forcing, (size_t)n_scen * h->n_inputs * h->T * sizeof(double)));
        if (h->kind == RSCM_KIND_GHG_FORCING) {
            HIPCHK(hipFree(h->d_ghg_tables));
            h->d_ghg_tables = nullptr;
            HIPCHK(rscm::dev_malloc(&h->d_ghg_tables, (size_t)n_scen * rscm::kGhgRows * h->T * sizeof(double)));
        }
    }
    HIPCHK(hipMemcpyAsync(h->d_forcing, series, (size_t)n_scen * h->n_inputs * h->T * sizeof(double), hipMemcpyHostToDevice, h->stream));
    std::vector<double> ghg_tab;  // must outlive the copy below (synchronised before return)
    if (h->kind == RSCM_KIND_GHG_FORCING) {
        ghg_tab = ghg_tables(series, n_scen, h->T);
        HIPCHK(hipMemcpyAsync(h->d_ghg_tables, ghg_tab.data(), ghg_tab.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
    }
    if (scenario_of_member) {
        if (!h->d_scen) HIPCHK(rscm::dev_malloc(&h->d_scen, (size_t)h->N * sizeof(int32_t)));
        HIPCHK(hipMemcpyAsync(h->d_scen, scenario_of_member, (size_t)h->N * sizeof(int32_t), hipMemcpyHostToDevice, h->stream));
    } else if (h->d_scen) {
        HIPCHK(hipFree(h->d_scen));
        h->d_scen = nullptr;
    }
    HIPCHK(hipStreamSynchronize(h->stream));
    h->n_scen = n_scen;
    h->source = source;
    h->forcing_set = true;
    if (h->kind == RSCM_KIND_AGGREGATE) {  // rows after the last non-NaN one of the block stay out of the sums anyway
        int32_t used = 0;
        for (int32_t sidx = 0; sidx < n_scen; ++sidx)
            for (int32_t k = 0; k < h->n_inputs; ++k)
                for (int32_t t = 0; t < h->T; ++t)
                    if (!std::isnan(series[((size_t)sidx * h->n_inputs + k) * h->T + t])) used = std::max(used, k + 1);
        h->ag_rows_set = used;
    }
    return RSCM_OK;
    GUARD_END
}

int rscm_ens_link_input(rscm_ens* h, int32_t input_row, rscm_ens* src, int32_t src_var, int32_t source)
{
    GUARD_BEGIN
    NEED(h);
    if (!src) return fail(RSCM_ERR_INVALID, "source ensemble is NULL");
    if (src == h) return fail(RSCM_ERR_INVALID, "an ensemble cannot link to itself (its own states are read in the kernel)");
    if (h->kind == RSCM_KIND_COUPLED || h->kind == RSCM_KIND_HALOCARBON)
        return fail(RSCM_ERR_INVALID, "the inputs of this kind cannot be linked (they are exogenous emissions)");
    if (input_row < 0 || input_row >= h->n_inputs || input_row >= rscm::kMaxLinks)
        return fail(RSCM_ERR_INVALID, "input row %d out of range [0, %d)", input_row, h->n_inputs);
    if (src_var < 1 || src_var >= src->V) return fail(RSCM_ERR_INVALID, "variable %d of the source has no stored series", src_var);
    if (source != RSCM_SRC_EXOGENOUS && source != RSCM_SRC_UPSTREAM) return fail(RSCM_ERR_INVALID, "unknown source %d", source);
    if (src->N != h->N || src->T != h->T || src->device != h->device)
        return fail(RSCM_ERR_INVALID, "linked ensembles need the same members, time points and device (%lld x %d on %d vs %lld x %d on %d)",
                    (long long)src->N, src->T, src->device, (long long)h->N, h->T, h->device);
    if ((!src->windowed && src->rows != src->T) || (!h->windowed && h->rows != h->T))
        return fail(RSCM_ERR_INVALID, "linked ensembles must store their series (no RSCM_FLAG_NO_SERIES)");
    if (!h->link_order_check) src->read_ahead = true;
    auto& l = h->links[input_row];
    if (l.src) --l.src->link_refs;
    else ++h->n_linked;
    l.src = src;
    l.var = src_var;
    l.off = source == RSCM_SRC_UPSTREAM ? 1 : 0;
    ++src->link_refs;
    return RSCM_OK;
    GUARD_END
}

int rscm_ens_set_link_order_check(rscm_ens* h, int32_t enabled)
{
    NEED(h);
    h->link_order_check = enabled != 0;
    if (!h->link_order_check)  // the producers' rows beyond what they have written must read as NaN
        for (auto& l : h->links)
            if (l.src) l.src->read_ahead = true;
    return RSCM_OK;
}

int rscm_ens_unlink_input(rscm_ens* h, int32_t input_row)
{
    NEED(h);
    if (input_row < 0 || input_row >= h->n_inputs || input_row >= rscm::kMaxLinks)
        return fail(RSCM_ERR_INVALID, "input row %d out of range [0, %d)", input_row, h->n_inputs);
    auto& l = h->links[input_row];
    if (l.src) {
        --l.src->link_refs;
        --h->n_linked;
        l = rscm_ens::Link{};
    }
    return RSCM_OK;
}

int rscm_ens_set_initial(rscm_ens* h, int32_t var_id, const double* values, int64_t n_values)
{
    GUARD_BEGIN
    NEED(h);
    if (var_id < 1 || var_id >= h->V) return fail(RSCM_ERR_INVALID, "variable %d has no stored series", var_id);
    if (!values || (n_values != 1 && n_values != h->N))
        return fail(RSCM_ERR_INVALID, "initial values: need 1 or n_members values, got %lld", (long long)n_values);
    if (int rc = set_device(h)) return rc;
    if (h->windowed) {
        if (h->win0 != 0)
            if (int rc = window_reset(h, false)) return rc;
        h->row0_saved = false;  // saved again when step 0 starts
    }
    if (n_values == 1)
        HIPCHK(rscm::launch_fill(h->series(var_id), h->N, values[0], h->stream));
    else
        HIPCHK(hipMemcpyAsync(h->series(var_id), values, (size_t)h->N * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    h->initial_set[var_id] = 1;
    h->time_index = 0;
    return RSCM_OK;
    GUARD_END
}

int rscm_ens_set_state(rscm_ens* h, int32_t var_id, int32_t tidx, const double* values, int64_t n_values)
{
    GUARD_BEGIN
    NEED(h);
    if (var_id < 1 || var_id >= h->V) return fail(RSCM_ERR_INVALID, "variable %d has no stored series", var_id);
    if (h->windowed ? (tidx < h->win0 || tidx >= h->win0 + h->rows) : (tidx < 0 || tidx >= h->rows))
        return fail(RSCM_ERR_INVALID, h->windowed ? "time index %d is outside the stored window (move the stepper there first: rscm_ens_set_time_index)"
                                                   : "time index %d out of range", tidx);
    if (!values || (n_values != 1 && n_values != h->N))
        return fail(RSCM_ERR_INVALID, "state values: need 1 or n_members values, got %lld", (long long)n_values);
    if (int rc = set_device(h)) return rc;
    if (h->windowed && tidx == 0) h->row0_saved = false;
    double* row = h->series(var_id) + (size_t)tidx * h->N;
    if (n_values == 1)
        HIPCHK(rscm::launch_fill(row, h->N, values[0], h->stream));
    else
        HIPCHK(hipMemcpyAsync(row, values, (size_t)h->N * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    if (tidx == 0) h->initial_set[var_id] = 1;
    return RSCM_OK;
    GUARD_END
}

int rscm_ens_set_time_index(rscm_ens* h, int32_t tidx)
{
    GUARD_BEGIN
    NEED(h);
    if (tidx < 0 || tidx > (h->windowed ? h->T : h->rows) - 1) return fail(RSCM_ERR_INVALID, "time index %d out of range", tidx);
    // kinds whose components keep internal state in the reference (ComponentState: the UDEB ocean
    // columns, the OceanCarbon flux history) can only continue from where that state stands
    if ((h->kind == RSCM_KIND_UDEB || h->kind == RSCM_KIND_OCEAN_CARBON) && tidx != 0 && tidx != h->time_index)
        return fail(RSCM_ERR_STATE, "this kind carries internal component state on the device: the time index can "
                                    "only be rewound to 0 or left at %d", h->time_index);
    if (tidx > 0)
        for (int32_t v = 1; v < h->V; ++v)
            if (h->is_state(v)) h->initial_set[v] = 1;  // a restored checkpoint carries its own state rows
    if (h->windowed && tidx != h->time_index) {
        if (int rc = set_device(h)) return rc;
        if (int rc = window_seek(h, tidx)) return rc;
    }
    h->time_index = tidx;
    h->ocean_tile_base = -1;
    h->ocean_modes_at = -1;
    return RSCM_OK;
    GUARD_END
}

typedef std::vector<std::pair<double*, int64_t>> StatePieces;

// the pieces of a handle's internal component state at time index k: (device pointer, doubles)
static void internal_pieces(const rscm_ens* h, int32_t k, StatePieces& p)
{
    if (h->kind == RSCM_KIND_UDEB && h->d_ocean && h->d_scal && h->d_hist) {
        p.emplace_back(h->d_ocean, (int64_t)2 * h->udeb_n_layers * h->N);
        p.emplace_back(h->d_scal, (int64_t)rscm::kUdebScalars * h->N);
        p.emplace_back(h->d_hist, (int64_t)(k + 1) * h->N);
    } else if (h->kind == RSCM_KIND_OCEAN_CARBON && h->d_ocean_hist) {
        // the pulses still held, oldest first: the ring unrolled into (at most) two pieces
        const int64_t total = (int64_t)k * h->ocean_steps, R = h->ocean_hist_rows;
        const int64_t cnt = std::min(total, R), first = (total - cnt) % R;
        const int64_t head = std::min(cnt, R - first);
        if (head > 0) p.emplace_back(h->d_ocean_hist + (size_t)first * h->N, head * h->N);
        if (cnt > head) p.emplace_back(h->d_ocean_hist, (cnt - head) * h->N);
    }
}

int rscm_ens_internal_state_size(rscm_ens* h, int64_t* n_doubles)
{
    NEED(h);
    if (!n_doubles) return fail(RSCM_ERR_INVALID, "n_doubles is NULL");
    int64_t n = 0;
    StatePieces pieces;
    internal_pieces(h, h->time_index, pieces);
    for (const auto& piece : pieces) n += piece.second;
    *n_doubles = n;
    return RSCM_OK;
}

int rscm_ens_get_internal_state(rscm_ens* h, double* out)
{
    GUARD_BEGIN
    NEED(h);
    if (!out) return fail(RSCM_ERR_INVALID, "out is NULL");
    if (int rc = set_device(h)) return rc;
    StatePieces pieces;
    internal_pieces(h, h->time_index, pieces);
    for (const auto& piece : pieces) {
        if (piece.second > 0)
            HIPCHK(hipMemcpyAsync(out, piece.first, (size_t)piece.second * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        out += piece.second;
    }
    HIPCHK(hipStreamSynchronize(h->stream));
    return RSCM_OK;
    GUARD_END
}

int rscm_ens_set_internal_state(rscm_ens* h, const double* in, int64_t n_doubles, int32_t time_index)
{
    GUARD_BEGIN
    NEED(h);
    if (time_index < 0 || time_index > (h->windowed ? h->T : h->rows) - 1) return fail(RSCM_ERR_INVALID, "time index %d out of range", time_index);
    if ((h->kind == RSCM_KIND_UDEB && !h->udeb_ready) || (h->kind == RSCM_KIND_OCEAN_CARBON && !h->ocean_ready))
        return fail(RSCM_ERR_STATE, "set the parameters first: they size the internal state");
    StatePieces pieces;
    internal_pieces(h, time_index, pieces);
    int64_t n = 0;
    for (const auto& piece : pieces) n += piece.second;
    if (n != n_doubles) return fail(RSCM_ERR_INVALID, "internal state at time index %d is %lld doubles, got %lld", time_index, (long long)n, (long long)n_doubles);
    if (n > 0 && !in) return fail(RSCM_ERR_INVALID, "in is NULL");
    if (int rc = set_device(h)) return rc;
    for (const auto& piece : pieces) {
        if (piece.second > 0)
            HIPCHK(hipMemcpyAsync(piece.first, in, (size_t)piece.second * sizeof(double), hipMemcpyHostToDevice, h->stream));
        in += piece.second;
    }
    HIPCHK(hipStreamSynchronize(h->stream));
    if (time_index > 0)
        for (int32_t v = 1; v < h->V; ++v)
            if (h->is_state(v)) h->initial_set[v] = 1;
    if (h->windowed && time_index != h->time_index)
        if (int rc = window_seek(h, time_index)) return rc;
    h->time_index = time_index;
    h->ocean_tile_base = -1;
    h->ocean_modes_at = -1;
    return RSCM_OK;
    GUARD_END
}

int rscm_ens_rewind(rscm_ens* h)
{
    GUARD_BEGIN
    NEED(h);
    if (h->windowed) {
        if (int rc = set_device(h)) return rc;
        if (int rc = window_reset(h, false)) return rc;
    }
    h->time_index = 0;
    h->ocean_tile_base = -1;
    h->ocean_modes_at = -1;
    return RSCM_OK;
    GUARD_END
}

// ---- one launch range of one handle, in pieces (rscm_ens_run_lockstep fuses the launches of several handles) ----
// (1) what must hold before anything is enqueued
// The member constants of GhgForcing, TerrestrialCarbon and ClimateUDEB (what their bodies used to form from the parameters alone at the
// top of every launch; ClimateUDEB: the base LAMCALC solve): one small kernel whenever the parameter block has been written since the last one -- rscm_ens_set_params*,
// rscm_ens_sample_lhs, a checkpoint restore, a sampler's proposals -- and before EVERY run of a handle whose block the caller may
// write directly (rscm_ens_params_devptr).
static bool has_derived(const rscm_ens* h)
{
    return h->kind == RSCM_KIND_GHG_FORCING || h->kind == RSCM_KIND_TERRESTRIAL_CARBON || h->kind == RSCM_KIND_UDEB;
}

// (test hook: derive launches issued by the calling thread, rscm_gpu_derive_launches)
static thread_local int64_t t_derive_launches = 0;
int64_t take_derive_launches()
{
    const int64_t n = t_derive_launches;
    t_derive_launches = 0;
    return n;
}

int ensure_derived(rscm_ens* h)
{
    if (!has_derived(h) || !h->params_set) return RSCM_OK;
    if (!h->derived_dirty && !h->params_exposed && h->d_derived) return RSCM_OK;
    // a block the caller may write directly is re-derived before every run -- once per API call: the steps of one
    // rscm_ens_run_lockstep call share the constants formed at its start (a one-step launch per model step used to pay the
    // derive kernel, ClimateUDEB's secant solve, every step)
    if (h->derived_hold && !h->derived_dirty && h->d_derived) return RSCM_OK;
    if (int rc = set_device(h)) return rc;
    if (!h->d_derived) {
        const hipError_t e = rscm::dev_malloc(&h->d_derived, (size_t)rscm::kDerivedRows * h->N * sizeof(double));
        if (e != hipSuccess)
            return fail(e == hipErrorOutOfMemory ? RSCM_ERR_NOMEM : RSCM_ERR_DEVICE, "member constants of %lld members: %s", (long long)h->N, hipGetErrorString(e));
    }
    if (h->kind == RSCM_KIND_GHG_FORCING) HIPCHK(rscm::launch_ghg_derive(h->d_params, h->uniform_rows, h->ghg_method, h->N, h->d_derived, h->stream));
    else if (h->kind == RSCM_KIND_UDEB) HIPCHK(rscm::launch_udeb_derive(h->d_params, h->uniform_rows, h->N, h->d_derived, h->stream));
    else HIPCHK(rscm::launch_terrestrial_derive(h->d_params, h->uniform_rows, h->N, h->d_derived, h->stream));
    ++t_derive_launches;
    h->derived_dirty = false;
    return RSCM_OK;
}

int step_check(rscm_ens* h, int32_t step_begin, int32_t step_end, bool derive)
{
    NEED(h);
    if (step_begin < 0 || step_end > h->T - 1 || step_begin > step_end)
        return fail(RSCM_ERR_STATE, "steps [%d, %d) outside [0, %d] (Model::step asserts time_index < len-1)",
                    step_begin, step_end, h->T - 1);
    if (step_begin != h->time_index)
        return fail(RSCM_ERR_STATE, "step_begin %d != current time index %d", step_begin, h->time_index);
    if (!h->windowed && h->rows != h->T && step_end > step_begin)
        return fail(RSCM_ERR_STATE, "this handle stores no series (RSCM_FLAG_NO_SERIES): use rscm_ens_run_loglik");
    const int32_t keep = h->keep_rows();
    if (h->windowed && step_end > step_begin) {
        if (h->rows < 2 * keep)
            return fail(RSCM_ERR_STATE, "a window of %d rows is too short for a component that reads %d of its own earlier rows (need >= %d)",
                        h->rows, h->lookback, 2 * keep);
        if (step_end - step_begin + keep > h->rows)
            return fail(RSCM_ERR_STATE, "steps [%d, %d) do not fit a window of %d rows (%d are kept for look-back): step in shorter ranges",
                        step_begin, step_end, h->rows, keep);
        // the rows the first step looks back at must be resident (they are after any slide made with this
        // look-back; not after the look-back was raised on a window that had already moved)
        if (std::max(0, step_begin - h->lookback) < h->win0)
            return fail(RSCM_ERR_STATE, "step %d reads row %d of its own series, the window starts at row %d", step_begin,
                        std::max(0, step_begin - h->lookback), h->win0);
    }
    if (!h->params_set) return fail(RSCM_ERR_STATE, "parameters not set");
    if (!h->forcing_set && h->n_linked < h->n_inputs) return fail(RSCM_ERR_STATE, "shared input series not set");
    for (int32_t v = 1; v < h->V; ++v)
        if (h->is_state(v) && !h->initial_set[v])  // builder.rs:704-717 MissingInitialValue
            return fail(RSCM_ERR_STATE, "state variable %d has no initial value (MissingInitialValue)", v);
    return derive ? ensure_derived(h) : RSCM_OK;
}

// (2) schedule tables and the handle's own window: room for the rows this range writes
int step_window_pre(rscm_ens* h, int32_t step_begin, int32_t step_end)
{
    const int32_t keep = h->keep_rows();
    if (int rc = set_device(h)) return rc;
    if (int rc = refresh_schedule(h)) return rc;
    if (h->windowed && step_end > step_begin) {
        // room for the rows this range writes must exist before its launch is enqueued: nothing here is left to a later flush
        struct Immediate {
            rscm_ens* h; WindowDeferral* d;
            explicit Immediate(rscm_ens* x) : h(x), d(x->defer) { h->defer = nullptr; }
            ~Immediate() { h->defer = d; }
        } now(h);
        if (step_begin == 0) {
            if (h->win0 != 0)
                if (int rc = window_reset(h, false)) return rc;
            if (!h->row0_saved) {  // the initial rows, for rewind
                HIPCHK(rscm::launch_gather_rows(h->d_series, h->N, h->rows, 0, nullptr, h->V - 1, h->d_row0, 1, 0, h->stream));
                h->row0_saved = true;
            }
            if (int rc = window_store_row(h, 0)) return rc;
        }
        if (step_end >= h->win0 + h->rows)
            if (int rc = window_slide(h, step_begin - keep + 1)) return rc;
        // the links of h were resolved against the producers' windows above; h's own base moved with the slide
    }

    return RSCM_OK;
}

// (3) the producing ensembles' series as they stand now (after every window of the graph has been moved)
int step_links(rscm_ens* h, int32_t step_begin, int32_t step_end, rscm::InputLinks& links, int32_t& linked_out)
{
    // linked inputs: the producing ensembles' series, in launch order on one stream
    links = rscm::InputLinks{};
    const int32_t linked = h->n_linked > 0 ? 1 : 0;
    linked_out = linked;
    for (int32_t k = 0; k < rscm::kMaxLinks && k < h->n_inputs; ++k) {
        const auto& l = h->links[k];
        if (!l.src) continue;
        if (l.src->stream != h->stream)
            return fail(RSCM_ERR_STATE, "input row %d is linked to an ensemble on another stream (rscm_ens_set_stream both to the same one)", k);
        // ClimateUDEB reads at_start / at_end, the aggregate at_end: index n+1 whatever `source` said
        const bool reads_end = h->kind == RSCM_KIND_UDEB || h->kind == RSCM_KIND_AGGREGATE;
        const int32_t need = step_end - 1 + (reads_end ? 1 : l.off);
        if (h->link_order_check && step_end > step_begin && l.src->time_index < need)
            return fail(RSCM_ERR_STATE, "input row %d reads index %d of its source, which has only been stepped to %d", k, need,
                        l.src->time_index);
        if (l.src->windowed && step_end > step_begin) {  // the rows this launch reads must be resident in the producer's window
            const int32_t lo = step_begin + (h->kind == RSCM_KIND_UDEB ? 0 : (reads_end ? 1 : l.off));
            const int32_t hi = step_end - 1 + (reads_end ? 1 : l.off);
            if (lo < l.src->win0 || hi >= l.src->win0 + l.src->rows)
                return fail(RSCM_ERR_STATE, "input row %d reads indices [%d, %d] of its source, whose window holds [%d, %d): step the graph in lock-step",
                            k, lo, hi, l.src->win0, l.src->win0 + l.src->rows);
        }
        links.row[k] = l.src->series(l.var);
        links.off[k] = h->kind == RSCM_KIND_AGGREGATE ? 0 : l.off;
    }
    return RSCM_OK;
}

// (4) the launch itself -- or, with op_out, its arguments for the group kernel (csrc/group.hip) if the kind can
// be fused with its neighbours (op_out->kind = -1 otherwise; nothing is launched either way)
// Whole-axis launches of the two-layer and the coupled kind as TWO member blocks on two streams, each in chunks of model steps issued
// in turn.  One launch of 1e5 members is 1564 wavefronts on 1024 SIMDs: the SIMDs that got two take twice as long as those that got one,
// and the launch takes the time of two (issue utilisation 0.66).  Cut into a block that fills the chip once
// (65 536 members) and the rest, each block on its own stream and in chunks of ~64 model steps, the same kernels resume from the rows they
// stored (as rscm_ens_run in pieces always could), a block's next chunk is dispatched while the other block's is still running, and the
// hardware's dispatcher evens out the SIMDs over the chunks: 2.86 -> 2.36 ms at 1e5 members x 750 years, 1.1-1.36x at every size between
// 1e5 and 3e5, 1.05x at 1e6, never slower with three chunks or more (scripts/multi_stream_two_layer.py).  Same kernels on the same
// operands: the same bits.  The caller's stream forks into the helper stream and joins it again with events: to the caller this is one
// asynchronous run on its stream, as before.  RSCM_SPLIT_RUNS=0 turns it off (A/B).
// A/B and test hook (include/rscm_gpu_internal.h, rscm_gpu_set_run_plan): how the calling thread's whole-axis runs go out --
// -1 by the environment and the sizes (default), 0 always one plain launch, 1 the two-stream cut where it applies.
static thread_local int32_t t_run_plan = -1;
void set_run_plan(int32_t mode) { t_run_plan = mode; }

struct MemberSplit {
    bool on = false;
    int64_t first = 0;     // members of the first block
    int32_t chunk = 0;     // model steps per launch
};
static MemberSplit plan_member_split(rscm_ens* h, int32_t step_begin, int32_t step_end, bool linked, bool halves = false)
{
    static const bool enabled = [] { const char* e = getenv("RSCM_SPLIT_RUNS"); return !e || atoi(e) != 0; }();
    // (tuning knobs of the experiments build only, experiment_env.hpp: model steps per chunk, members of the first block)
    static const int32_t chunk_env = (int32_t)rscm::experiment_env("RSCM_SPLIT_CHUNK", 0);
    static const int64_t first_env = (int64_t)rscm::experiment_env("RSCM_SPLIT_FIRST", 0);
    MemberSplit m;
    // two-layer / coupled: 32-64 steps per chunk 2.30 ms at 1e5 members, 96: 2.32, 192: 2.37 (scripts/sweep_split.sh); ClimateUDEB reloads
    // and stores its columns with every chunk: 96 (88 ms at 1e5 members against 90 with 64)
    const int32_t kChunk = chunk_env > 0 ? chunk_env : (halves ? 96 : 64);
    const int32_t len = step_end - step_begin;
    if ((t_run_plan >= 0 ? t_run_plan != 1 : !enabled) || linked || h->windowed || h->rows != h->T || len < 3 * kChunk) return m;
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, h->device) != hipSuccess || cus <= 0) return m;
    const int64_t per_round = (int64_t)cus * 4 * 64;   // one wavefront on every SIMD: 65 536 members on an MI355X
    if (h->N <= per_round) return m;                   // every wavefront has a SIMD to itself already
    m.first = halves ? (h->N / 2 + 63) / 64 * 64 : std::max(per_round, (h->N / 2) / per_round * per_round);
    if (first_env > 0) {   // (experiment knob) rounded up to whole wavefronts; ignored unless both blocks keep members
        const int64_t want = (first_env + 63) / 64 * 64;
        if (want > 0 && want < h->N) m.first = want;
    }
    if (!(m.first > 0 && m.first < h->N)) return m;   // never a block that reaches past the ensemble
    const int32_t n_chunks = (len + kChunk - 1) / kChunk;
    m.chunk = (len + n_chunks - 1) / n_chunks;
    m.on = true;
    return m;
}
static int member_split_streams(rscm_ens* h)
{
    if (!h->split_stream) HIPCHK(hipStreamCreateWithFlags(&h->split_stream, hipStreamNonBlocking));
    if (!h->split_fork) HIPCHK(hipEventCreateWithFlags(&h->split_fork, hipEventDisableTiming));
    if (!h->split_join) HIPCHK(hipEventCreateWithFlags(&h->split_join, hipEventDisableTiming));
    return RSCM_OK;
}
// Test hook (include/rscm_gpu_internal.h, rscm_gpu_fail_chunk_launch): the k-th chunk launch of the calling thread's next cut run
// reports a launch failure instead of being issued -- the only way to execute the join-after-failure path below.
static thread_local int32_t t_fail_chunk = 0;
void set_fail_chunk_launch(int32_t k) { t_fail_chunk = k; }

// issue(begin, end, first_member, count, stream) launches one chunk of one block; the fork and the join around all of them
static int run_member_split(rscm_ens* h, const MemberSplit& m, int32_t step_begin, int32_t step_end,
                            const std::function<hipError_t(int32_t, int32_t, int64_t, int64_t, hipStream_t)>& issue)
{
    if (int rc = member_split_streams(h)) return rc;
    h->last_blocks = 2;
    h->last_chunks = (step_end - step_begin + m.chunk - 1) / m.chunk;
    HIPCHK(hipEventRecord(h->split_fork, h->stream));
    HIPCHK(hipStreamWaitEvent(h->split_stream, h->split_fork, 0));
    // (experiments: RSCM_SPLIT_CHUNK2 gives the second block its own chunk length; the block that is behind is issued next)
    static const int32_t chunk2_env = (int32_t)rscm::experiment_env("RSCM_SPLIT_CHUNK2", 0);
    const int32_t c0 = m.chunk, c1 = chunk2_env > 0 ? chunk2_env : m.chunk;
    hipError_t err = hipSuccess;
    auto guarded = [&](int32_t b, int32_t e, int64_t m0, int64_t cnt, hipStream_t st) -> hipError_t {
        if (t_fail_chunk > 0 && --t_fail_chunk == 0) return hipErrorLaunchFailure;   // (test hook)
        return issue(b, e, m0, cnt, st);
    };
    for (int32_t b0 = step_begin, b1 = step_begin; err == hipSuccess && (b0 < step_end || b1 < step_end);) {
        if (b0 < step_end && (b0 <= b1 || b1 >= step_end)) {
            const int32_t e = std::min(step_end, b0 + c0);
            err = guarded(b0, e, (int64_t)0, m.first, h->stream);
            b0 = e;
        } else {
            const int32_t e = std::min(step_end, b1 + c1);
            err = guarded(b1, e, m.first, h->N - m.first, h->split_stream);
            b1 = e;
        }
    }
    t_fail_chunk = 0;   // (test hook) a cut run consumes it, whether k was reached or not: it never outlives the run it was set for
    // the join is made whatever happened: the caller's stream never runs ahead of what was issued on the helper stream
    HIPCHK(hipEventRecord(h->split_join, h->split_stream));
    HIPCHK(hipStreamWaitEvent(h->stream, h->split_join, 0));
    HIPCHK(err);
    return RSCM_OK;
}

int step_launch(rscm_ens* h, int32_t step_begin, int32_t step_end, const rscm::InputLinks& links, int32_t linked,
                rscm::GroupOp* op_out)
{
    const int32_t len = step_end - step_begin;
    const size_t lds_bytes = (size_t)h->n_scen * (size_t)len * sizeof(double);
    if (!op_out) h->last_blocks = h->last_chunks = 1;
    if (h->kind == RSCM_KIND_TWO_LAYER) {
        rscm::TwoLayerArgs a{};
        a.n_members = h->N;
        a.row_stride = h->N;
        a.n_times = h->T;
        a.step_begin = step_begin;
        a.step_end = step_end;
        a.n_scen = h->n_scen;
        a.src_off = h->source == RSCM_SRC_UPSTREAM ? 1 : 0;
        a.lds_forcing = lds_bytes <= (size_t)rscm::kMaxLds - 1024 ? 1 : 0;
        a.params = h->d_params;
        a.uniform_rows = h->uniform_rows;
        a.forcing = h->d_forcing;
        if (linked) {
            a.link = links.row[0];
            a.src_off = links.off[0];
            a.lds_forcing = 0;
            a.n_scen = 1;
        }
        a.scen = linked ? nullptr : h->d_scen;
        a.nsub = h->d_nsub_tl;
        a.h = h->h_tl;
        a.h_half = h->h_tl / 2.0;
        a.h_sixth = h->h_tl / 6.0;
        a.ts = h->series(RSCM_TL_VAR_TS);
        a.td = h->series(RSCM_TL_VAR_TD);
        a.status = h->d_status;
        if (op_out) {  // the arguments go into the fused launch's table instead (csrc/group.hip)
            a.lds_forcing = 0;
            op_out->kind = h->kind;
            op_out->variant = h->mode;
            op_out->u.tl = a;
            return RSCM_OK;
        }
        const MemberSplit ms = plan_member_split(h, step_begin, step_end, linked != 0);
        if (ms.on) {
            const int32_t mode = h->mode;
            if (int rc = run_member_split(h, ms, step_begin, step_end, [&](int32_t b, int32_t e, int64_t m0, int64_t cnt, hipStream_t st) {
                    rscm::TwoLayerArgs c = a;
                    c.n_members = cnt;
                    c.step_begin = b;
                    c.step_end = e;
                    c.lds_forcing = (size_t)h->n_scen * (size_t)(e - b) * sizeof(double) <= (size_t)rscm::kMaxLds - 1024 ? 1 : 0;
                    c.params = a.params + m0;
                    if (a.scen) c.scen = a.scen + m0;
                    c.ts = a.ts + m0;
                    c.td = a.td + m0;
                    c.status = a.status + m0;
                    return rscm::launch_two_layer(c, mode, st);
                }))
                return rc;
        } else {
            HIPCHK(rscm::launch_two_layer(a, h->mode, h->stream));
        }
    } else if (h->kind == RSCM_KIND_GHG_FORCING) {
        rscm::GhgArgs a{};
        a.n_members = h->N;
        a.n_times = h->T;
        a.step_begin = step_begin;
        a.step_end = step_end;
        a.rows = h->rows;
        a.method = h->ghg_method;
        a.params = h->d_params;
        a.uniform_rows = h->uniform_rows;
        a.derived = h->d_derived;
        a.derived_uniform = (h->uniform_rows & rscm::ghg_derive_sources(h->ghg_method)) == rscm::ghg_derive_sources(h->ghg_method) ? 1 : 0;
        a.tables = h->d_ghg_tables;
        a.scen = h->d_scen;
        a.conc = h->d_forcing;
        a.links = links;
        a.linked = linked;
        a.erf_co2 = h->series(RSCM_GH_VAR_ERF_CO2);
        a.erf_ch4 = h->series(RSCM_GH_VAR_ERF_CH4);
        a.erf_n2o = h->series(RSCM_GH_VAR_ERF_N2O);
        a.status = h->d_status;
        if (op_out) {  // the arguments go into the fused launch's table instead (csrc/group.hip)
            if (!linked) {  // the table path uses host-built rows: not fused (fusable() keeps such a handle out)
                op_out->kind = -1;
                return RSCM_OK;
            }
            op_out->kind = h->kind;
            op_out->variant = h->ghg_method;
            op_out->u.ghg = a;
            return RSCM_OK;
        }
        HIPCHK(rscm::launch_ghg(a, h->stream));
    } else if (h->kind == RSCM_KIND_HALOCARBON) {
        rscm::HaloArgs a{};
        a.n_members = h->N;
        a.n_times = h->T;
        a.step_begin = step_begin;
        a.step_end = step_end;
        a.params = h->d_params;
        a.uniform_rows = h->uniform_rows;
        a.emissions = h->d_forcing;
        a.scen = h->d_scen;
        a.bounds = h->d_bounds;
        a.rows = h->rows;
        a.series = h->series(1);
        a.status = h->d_status;
        if (op_out) { op_out->kind = -1; return RSCM_OK; }
        HIPCHK(rscm::launch_halocarbon(a, h->stream));
    } else if (h->kind == RSCM_KIND_OCEAN_CARBON) {
        if (!h->ocean_ready) return fail(RSCM_ERR_STATE, "OceanCarbon parameters not configured");
        rscm::OceanArgs a{};
        a.n_members = h->N;
        a.n_times = h->T;
        a.step_begin = step_begin;
        a.step_end = step_end;
        a.steps = h->ocean_steps;
        a.fused = h->mode == RSCM_MODE_FAST ? 1 : 0;
        a.max_hist = h->ocean_max_hist;
        a.params = h->d_params;
        a.uniform_rows = h->uniform_rows;
        a.inputs = h->d_forcing;
        a.scen = h->d_scen;
        a.bounds = h->d_bounds;
        a.links = links;
        a.linked = linked;
        a.irf = h->d_ocean_irf;
        a.hist = h->d_ocean_hist;
        a.hist_rows = (int32_t)h->ocean_hist_rows;
        // one step at a time (linked graphs, Model::step): pair the steps up so that the history is
        // read once per two steps, as the two-year tiles of a whole run do
        a.partial = h->d_ocean_partial;
        a.part = -1;
        a.recur = (h->mode == RSCM_MODE_FAST && h->ocean_recur_ok) ? 1 : 0;
        if (a.recur) {
            a.near = h->ocean_near;
            a.modes = h->ocean_modes;
            a.mode_state = h->d_ocean_mode_state;
            a.mode_table = h->d_ocean_mode_table;
            a.rebuild = h->ocean_modes_at == step_begin ? 0 : 1;
            h->ocean_modes_at = step_end;
        } else {
            h->ocean_modes_at = -1;  // this launch does not advance the running sums
        }
        const int32_t tile_years = rscm::kOceanSplitYears;
        if (a.recur) {
            h->ocean_tile_base = -1;
        } else if (step_end - step_begin == 1 && h->d_ocean_partial) {
            const int32_t p = step_begin - h->ocean_tile_base;
            if (h->ocean_tile_base >= 0 && h->ocean_tile_years == tile_years && p > 0 && p < tile_years) {
                a.part = p;                                   // the next year of the tile in flight
                if (p == tile_years - 1) h->ocean_tile_base = -1;
            } else if (step_begin + tile_years <= h->T - 1) {  // all its steps exist: start a tile here
                a.part = 0;
                h->ocean_tile_base = step_begin;
                h->ocean_tile_years = tile_years;
            } else {
                h->ocean_tile_base = -1;
            }
        } else {
            h->ocean_tile_base = -1;
        }
        a.rows = h->rows;
        a.series = h->series(1);
        a.status = h->d_status;
        if (op_out) {
            op_out->kind = -1;   // a heavy component: its own launch
            return RSCM_OK;
        }
        HIPCHK(rscm::launch_ocean(a, h->stream));
    } else if (h->kind == RSCM_KIND_CO2_BUDGET || h->kind == RSCM_KIND_TERRESTRIAL_CARBON || h->kind == RSCM_KIND_CARBON_CYCLE) {
        rscm::CarbonArgs a{};
        a.n_members = h->N;
        a.n_times = h->T;
        a.step_begin = step_begin;
        a.step_end = step_end;
        a.kind = h->kind;
        a.params = h->d_params;
        a.uniform_rows = h->uniform_rows;
        a.derived = h->d_derived;   // (TerrestrialCarbon only; nullptr for the other two kinds)
        a.derived_uniform = (h->uniform_rows & rscm::terrestrial_derive_sources()) == rscm::terrestrial_derive_sources() ? 1 : 0;
        a.inputs = h->d_forcing;
        a.scen = h->d_scen;
        a.links = links;
        a.linked = linked;
        a.bounds = h->d_bounds;
        a.nsub = h->d_nsub_cc;
        a.h = h->h_cc;
        a.h_half = h->h_cc / 2.0;
        a.h_sixth = h->h_cc / 6.0;
        a.rows = h->rows;
        a.series = h->series(1);
        a.status = h->d_status;
        if (op_out) {  // the arguments go into the fused launch's table instead (csrc/group.hip)
            op_out->kind = h->kind;
            op_out->variant = h->kind == RSCM_KIND_CARBON_CYCLE ? h->mode : 0;
            op_out->u.carbon = a;
            return RSCM_OK;
        }
        HIPCHK(rscm::launch_carbon(a, h->mode, h->stream));
    } else if (h->kind == RSCM_KIND_CH4_CHEMISTRY || h->kind == RSCM_KIND_N2O_CHEMISTRY) {
        rscm::ChemArgs a{};
        a.n_members = h->N;
        a.n_times = h->T;
        a.step_begin = step_begin;
        a.step_end = step_end;
        a.kind = h->kind;
        a.params = h->d_params;
        a.uniform_rows = h->uniform_rows;
        a.inputs = h->d_forcing;
        a.scen = h->d_scen;
        a.bounds = h->d_bounds;
        a.links = links;
        a.linked = linked;
        a.conc = h->series(RSCM_CHEM_VAR_CONC);
        a.lifetime = h->series(RSCM_CHEM_VAR_LIFETIME);
        a.status = h->d_status;
        if (op_out) {  // the arguments go into the fused launch's table instead (csrc/group.hip)
            op_out->kind = h->kind;
            op_out->variant = 0;
            op_out->u.chem = a;
            return RSCM_OK;
        }
        HIPCHK(rscm::launch_chem(a, h->stream));
    } else if ((h->kind >= RSCM_KIND_OZONE_FORCING && h->kind <= RSCM_KIND_AEROSOL_INDIRECT) ||
               h->kind == RSCM_KIND_FOURBOX_OHU || h->kind == RSCM_KIND_OSPP || h->kind == RSCM_KIND_CO2_ERF ||
               h->kind == RSCM_KIND_AGGREGATE) {
        rscm::PointwiseArgs a{};
        a.n_members = h->N;
        a.n_times = h->T;
        a.step_begin = step_begin;
        a.step_end = step_end;
        a.rows = h->rows;
        a.kind = h->kind;
        a.params = h->d_params;
        a.uniform_rows = h->uniform_rows;
        a.inputs = h->d_forcing;
        a.scen = h->d_scen;
        a.links = links;
        a.linked = linked;
        a.n_inputs_used = h->ag_rows_set;
        for (int32_t k = 0; k < rscm::kMaxLinks && k < h->n_inputs; ++k)
            if (h->links[k].src) a.n_inputs_used = std::max(a.n_inputs_used, k + 1);
        a.out = h->series(1);
        a.status = h->d_status;
        if (op_out) {  // the arguments go into the fused launch's table instead (csrc/group.hip)
            op_out->kind = h->kind;
            op_out->variant = 0;
            op_out->u.pw = a;
            return RSCM_OK;
        }
        HIPCHK(rscm::launch_pointwise(a, h->stream));
    } else if (h->kind == RSCM_KIND_UDEB) {
        if (!h->udeb_ready) return fail(RSCM_ERR_STATE, "ClimateUDEB parameters not configured");
        rscm::UdebArgs a{};
        a.n_members = h->N;
        a.row_stride = h->N;
        a.n_total = h->N;
        a.n_times = h->T;
        a.step_begin = step_begin;
        a.step_end = step_end;
        a.n_scen = h->n_scen;
        a.n_layers = h->udeb_n_layers;
        a.steps_per_year = h->udeb_steps;
        a.land_hc = h->udeb_land_hc;
        a.efficacy_apply = h->udeb_efficacy;
        a.fast = h->mode == RSCM_MODE_FAST ? 1 : 0;
        a.params = h->d_params;
        a.uniform_rows = h->uniform_rows;
        a.derived = h->d_derived;
        a.derived_uniform = (h->uniform_rows & rscm::udeb_derive_sources()) == rscm::udeb_derive_sources() ? 1 : 0;
        a.erf = h->d_forcing;
        a.link = linked ? links.row[0] : nullptr;
        a.scen = linked ? nullptr : h->d_scen;
        a.bounds = h->d_bounds;
        a.win_kfull = h->d_win_kfull;
        a.win_partw = h->d_win_partw;
        if (h->udeb_tables.size() != (size_t)6 * h->udeb_n_layers) return fail(RSCM_ERR_STATE, "ClimateUDEB tables not built");
        if (rscm::udeb_layers_unrolled(h->udeb_n_layers))   // by value, in the kernel-argument segment (rows past n_layers stay zero)
            memcpy(a.tables, h->udeb_tables.data(), h->udeb_tables.size() * sizeof(double));
        if (!rscm::udeb_layers_fixed(h->udeb_n_layers)) {
            if (!h->d_udeb_tables || !h->d_udeb_work || h->udeb_work_layers < h->udeb_n_layers)
                return fail(RSCM_ERR_STATE, "ClimateUDEB work arrays for %d layers not allocated", h->udeb_n_layers);
            a.tables_dev = h->d_udeb_tables;
            a.work = h->d_udeb_work;
        }
        a.ocean = h->d_ocean;
        a.scal = h->d_scal;
        a.hist = h->d_hist;
        a.st0 = h->series(RSCM_UD_VAR_ST_NH_OCEAN);
        a.st1 = h->series(RSCM_UD_VAR_ST_NH_LAND);
        a.st2 = h->series(RSCM_UD_VAR_ST_SH_OCEAN);
        a.st3 = h->series(RSCM_UD_VAR_ST_SH_LAND);
        a.heat_uptake = h->series(RSCM_UD_VAR_HEAT_UPTAKE);
        a.ohc = h->series(RSCM_UD_VAR_OHC);
        a.sst = h->series(RSCM_UD_VAR_SST);
        a.status = h->d_status;
        if (op_out) {
            op_out->kind = -1;
            return RSCM_OK;
        }
        // (ClimateUDEB runs one wavefront per SIMD, so "rounds" of 65 536 members: halves even out best -- 1e5 members x 750 years
        // 106 -> 86 ms, 2e5 213 -> 168 ms, nothing to gain at 125 000 = 1.91 rounds; scripts/multi_stream_udeb.py)
        const MemberSplit ms = plan_member_split(h, step_begin, step_end, linked != 0, /*halves=*/true);
        if (ms.on) {
            if (int rc = run_member_split(h, ms, step_begin, step_end, [&](int32_t b, int32_t e, int64_t m0, int64_t cnt, hipStream_t st) {
                    rscm::UdebArgs c = a;
                    c.n_members = cnt;
                    c.step_begin = b;
                    c.step_end = e;
                    c.params = a.params + m0;
                    c.derived = a.derived + m0;
                    if (a.scen) c.scen = a.scen + m0;
                    c.ocean = a.ocean + m0; c.scal = a.scal + m0; c.hist = a.hist + m0;
                    if (a.work) c.work = a.work + m0;
                    c.st0 = a.st0 + m0; c.st1 = a.st1 + m0; c.st2 = a.st2 + m0; c.st3 = a.st3 + m0;
                    c.heat_uptake = a.heat_uptake + m0; c.ohc = a.ohc + m0; c.sst = a.sst + m0;
                    c.status = a.status + m0;
                    return rscm::launch_udeb(c, st);
                }))
                return rc;
        } else {
            HIPCHK(rscm::launch_udeb(a, h->stream));
        }
    } else {
        rscm::CoupledArgs a{};
        a.n_members = h->N;
        a.row_stride = h->N;
        a.n_times = h->T;
        a.step_begin = step_begin;
        a.step_end = step_end;
        a.n_scen = h->n_scen;
        a.lds_forcing = lds_bytes <= (size_t)rscm::kMaxLds - 1024 ? 1 : 0;
        a.params = h->d_params;
        a.uniform_rows = h->uniform_rows;
        a.emissions = h->d_forcing;
        a.scen = h->d_scen;
        a.nsub_tl = h->d_nsub_tl;
        a.nsub_cc = h->d_nsub_cc;
        a.h_tl = h->h_tl;
        a.h_cc = h->h_cc;
        a.ts = h->series(RSCM_CP_VAR_TS);
        a.td = h->series(RSCM_CP_VAR_TD);
        a.conc = h->series(RSCM_CP_VAR_CONC);
        a.cum_uptake = h->series(RSCM_CP_VAR_CUM_UPTAKE);
        a.cum_emis = h->series(RSCM_CP_VAR_CUM_EMIS);
        a.erf_co2 = h->series(RSCM_CP_VAR_ERF_CO2);
        a.erf_total = h->series(RSCM_CP_VAR_ERF);
        a.status = h->d_status;
        if (op_out) { op_out->kind = -1; return RSCM_OK; }
        const MemberSplit ms = plan_member_split(h, step_begin, step_end, false);
        if (ms.on) {
            const int32_t mode = h->mode;
            if (int rc = run_member_split(h, ms, step_begin, step_end, [&](int32_t b, int32_t e, int64_t m0, int64_t cnt, hipStream_t st) {
                    rscm::CoupledArgs c = a;
                    c.n_members = cnt;
                    c.step_begin = b;
                    c.step_end = e;
                    c.lds_forcing = (size_t)h->n_scen * (size_t)(e - b) * sizeof(double) <= (size_t)rscm::kMaxLds - 1024 ? 1 : 0;
                    c.params = a.params + m0;
                    if (a.scen) c.scen = a.scen + m0;
                    c.ts = a.ts + m0; c.td = a.td + m0; c.conc = a.conc + m0; c.cum_uptake = a.cum_uptake + m0; c.cum_emis = a.cum_emis + m0;
                    c.erf_co2 = a.erf_co2 + m0; c.erf_total = a.erf_total + m0;
                    c.status = a.status + m0;
                    return rscm::launch_coupled(c, mode, st);
                }))
                return rc;
        } else {
            HIPCHK(rscm::launch_coupled(a, h->mode, h->stream));
        }
    }
    return RSCM_OK;
}

// (5) bookkeeping after the launch: time index, strided outputs, room for the next step
int step_finish(rscm_ens* h, int32_t step_begin, int32_t step_end)
{
    const int32_t keep = h->keep_rows();
    h->time_index = step_end;
    if (h->windowed && step_end > step_begin) {
        if (h->n_out > 0)
            for (int32_t t = step_begin + 1; t <= step_end; ++t)
                if (int rc = window_store_row(h, t)) return rc;
        // make room for the next step now: consumers that run before this producer in the next step
        // resolve their links against the window as it will be when they read
        if (step_end + 1 >= h->win0 + h->rows && step_end < h->T - 1)
            if (int rc = window_slide(h, step_end - keep + 1)) return rc;
    }
    return RSCM_OK;
}

// One launch of the kind's kernel over [step_begin, step_end).  `timed` brackets it with the events
// rscm_ens_last_run_ms reads; the lock-step loop of rscm_ens_run_lockstep leaves them out.
int run_range(rscm_ens* h, int32_t step_begin, int32_t step_end, bool timed)
{
    NEED(h);
    if (int rc = step_check(h, step_begin, step_end, false)) return rc;
    if (int rc = step_window_pre(h, step_begin, step_end)) return rc;
    rscm::InputLinks links{};
    int32_t linked = 0;
    if (int rc = step_links(h, step_begin, step_end, links, linked)) return rc;
    if (timed) HIPCHK(hipEventRecord(h->ev0, h->stream));
    if (int rc = ensure_derived(h)) return rc;   // (after the first event: rscm_ens_last_run_ms includes the derive launch)
    if (int rc = step_launch(h, step_begin, step_end, links, linked, nullptr)) return rc;
    if (int rc = step_finish(h, step_begin, step_end)) return rc;
    if (timed) {
        HIPCHK(hipEventRecord(h->ev1, h->stream));
        h->timed = true;
    }
    return RSCM_OK;
}

int rscm_ens_clear_series(rscm_ens* h)
{
    GUARD_BEGIN
    NEED(h);
    if (int rc = set_device(h)) return rc;
    if (h->windowed) {
        if (int rc = window_reset(h, true)) return rc;
        if (h->d_out)
            HIPCHK(rscm::launch_fill(h->d_out, (int64_t)h->n_out * h->out_rows * h->N, std::numeric_limits<double>::quiet_NaN(), h->stream));
    } else if (h->rows > 1)
        for (int32_t v = 1; v < h->V; ++v)
            HIPCHK(rscm::launch_fill(h->series(v) + h->N, (int64_t)(h->rows - 1) * h->N,
                                     std::numeric_limits<double>::quiet_NaN(), h->stream));
    h->time_index = 0;
    h->ocean_tile_base = -1;
    h->ocean_modes_at = -1;
    return RSCM_OK;
    GUARD_END
}

int rscm_ens_clear_rows_after(rscm_ens* h, int32_t tidx)
{
    GUARD_BEGIN
    NEED(h);
    if (tidx < 0 || tidx > h->T - 1) return fail(RSCM_ERR_INVALID, "time index %d out of range", tidx);
    if (int rc = set_device(h)) return rc;
    if (h->windowed) {
        const int32_t first = std::max(0, tidx + 1 - h->win0);
        HIPCHK(rscm::launch_fill_rows(h->d_series, h->N, h->rows, h->V - 1, first, std::numeric_limits<double>::quiet_NaN(), h->stream));
    } else if (h->rows == h->T && tidx < h->T - 1)
        for (int32_t v = 1; v < h->V; ++v)
            HIPCHK(rscm::launch_fill(h->series(v) + (size_t)(tidx + 1) * h->N, (int64_t)(h->T - 1 - tidx) * h->N,
                                     std::numeric_limits<double>::quiet_NaN(), h->stream));
    return RSCM_OK;
    GUARD_END
}

int rscm_ens_run_async(rscm_ens* h, int32_t step_begin, int32_t step_end)
{
    GUARD_BEGIN
    return run_range(h, step_begin, step_end, true);
    GUARD_END
}

int rscm_ens_sync(rscm_ens* h)
{
    GUARD_BEGIN
    NEED(h);
    if (int rc = set_device(h)) return rc;
    HIPCHK(hipStreamSynchronize(h->stream));
    return RSCM_OK;
    GUARD_END
}

int rscm_ens_run(rscm_ens* h, int32_t step_begin, int32_t step_end)
{
    if (int rc = rscm_ens_run_async(h, step_begin, step_end)) return rc;
    return rscm_ens_sync(h);
}

int rscm_ens_last_run_plan(rscm_ens* h, int32_t* member_blocks, int32_t* step_chunks)
{
    NEED(h);
    if (member_blocks) *member_blocks = h->last_blocks;
    if (step_chunks) *step_chunks = h->last_chunks;
    return RSCM_OK;
}

int rscm_ens_last_run_ms(rscm_ens* h, float* out_ms)
{
    GUARD_BEGIN
    NEED(h);
    if (!out_ms) return fail(RSCM_ERR_INVALID, "out_ms is NULL");
    if (!h->timed) return fail(RSCM_ERR_STATE, "no run has been launched yet");
    if (int rc = set_device(h)) return rc;
    HIPCHK(hipEventSynchronize(h->ev1));
    HIPCHK(hipEventElapsedTime(out_ms, h->ev0, h->ev1));
    return RSCM_OK;
    GUARD_END
}

int rscm_ens_get_series(rscm_ens* h, int32_t var_id, int32_t t_begin, int32_t t_end, int32_t t_stride,
                        int64_t m_begin, int64_t m_end, double* out)
{
    GUARD_BEGIN
    NEED(h);
    if (var_id < 1 || var_id >= h->V) return fail(RSCM_ERR_INVALID, "variable %d has no stored series", var_id);
    if (t_begin < 0 || t_end > h->T || t_begin > t_end || t_stride < 1)
        return fail(RSCM_ERR_INVALID, "bad time range [%d, %d) stride %d", t_begin, t_end, t_stride);
    if (!h->windowed && h->rows != h->T && t_end > 1)
        return fail(RSCM_ERR_STATE, "this handle stores only the initial row (RSCM_FLAG_NO_SERIES)");
    if (m_begin < 0 || m_end > h->N || m_begin > m_end || !out)
        return fail(RSCM_ERR_INVALID, "bad member range [%lld, %lld)", (long long)m_begin, (long long)m_end);
    const int64_t width = m_end - m_begin;
    if (width == 0 || t_begin == t_end) return RSCM_OK;
    if (int rc = set_device(h)) return rc;
    if (h->windowed) {  // row by row from wherever each row is resident: the output store or the window
        int64_t r = 0;
        for (int32_t t = t_begin; t < t_end; t += t_stride, ++r) {
            double* dst = out + r * width;
            if (t > h->time_index) {
                for (int64_t m = 0; m < width; ++m) dst[m] = std::numeric_limits<double>::quiet_NaN();
                continue;
            }
            const double* src = h->row_ptr(var_id, t);
            if (!src)
                return fail(RSCM_ERR_STATE, "row %d of variable %d is not resident: the window holds [%d, %d) and the output store every %d-th row%s",
                            t, var_id, h->win0, h->win0 + h->rows, h->out_stride, h->out_slot[var_id] < 0 ? " of other variables" : "");
            HIPCHK(hipMemcpyAsync(dst, src + m_begin, (size_t)width * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        }
        HIPCHK(hipStreamSynchronize(h->stream));
        return RSCM_OK;
    }
    // rows beyond the current time index were never computed by this model instance: NaN
    int64_t n_rows = 0, n_valid = 0;
    for (int32_t t = t_begin; t < t_end; t += t_stride) {
        ++n_rows;
        if (t <= h->time_index) ++n_valid;
    }
    if (n_valid > 0)
        HIPCHK(hipMemcpy2DAsync(out, (size_t)width * sizeof(double),
                                h->series(var_id) + (size_t)t_begin * h->N + m_begin,
                                (size_t)h->N * sizeof(double) * (size_t)t_stride, (size_t)width * sizeof(double),
                                (size_t)n_valid, hipMemcpyDeviceToHost, h->stream));
    for (int64_t r = n_valid; r < n_rows; ++r)
        for (int64_t m = 0; m < width; ++m) out[r * width + m] = std::numeric_limits<double>::quiet_NaN();
    HIPCHK(hipStreamSynchronize(h->stream));
    return RSCM_OK;
    GUARD_END
}

int rscm_ens_series_devptr(rscm_ens* h, int32_t var_id, void** out)
{
    NEED(h);
    if (!out || var_id < 1 || var_id >= h->V) return fail(RSCM_ERR_INVALID, "variable %d has no stored series", var_id);
    if (h->windowed) return fail(RSCM_ERR_STATE, "a windowed ensemble has no contiguous [T][N] series: read rows with rscm_ens_get_series");
    *out = h->series(var_id);
    return RSCM_OK;
}

int rscm_ens_params_devptr(rscm_ens* h, void** out)
{
    NEED(h);
    if (!out) return fail(RSCM_ERR_INVALID, "out is NULL");
    *out = h->d_params;
    // Whatever the caller writes there, now or later through a pointer it kept: from here on no row of this
    // handle is ever treated as uniform again (rscm_ens_set_params keeps the flag clear as well).
    h->params_exposed = true;
    h->uniform_rows = 0;
    h->params_set = true;  // the caller fills it on the device
    return RSCM_OK;
}

int rscm_ens_status(rscm_ens* h, uint8_t* out)
{
    GUARD_BEGIN
    NEED(h);
    if (!out) return fail(RSCM_ERR_INVALID, "out is NULL");
    if (int rc = set_device(h)) return rc;
    HIPCHK(hipMemcpyAsync(out, h->d_status, (size_t)h->N, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return RSCM_OK;
    GUARD_END
}

// Gaussian log-likelihood of every member into h->d_loglik (device), synchronised before return.
static int loglik_on_device(rscm_ens* h, int32_t n_obs, const int32_t* obs_var, const int32_t* obs_tidx,
                            const double* obs_value, const double* obs_sigma, int32_t normalize)
{
    if (n_obs < 0 || (n_obs > 0 && (!obs_var || !obs_tidx || !obs_value || !obs_sigma)))
        return fail(RSCM_ERR_INVALID, "bad observation arrays");
    bool uncomputed = false;
    for (int32_t j = 0; j < n_obs; ++j) {
        if (obs_var[j] < 1 || obs_var[j] >= h->V) return fail(RSCM_ERR_INVALID, "observation %d: variable %d has no stored series", j, obs_var[j]);
        if (obs_tidx[j] < 0 || obs_tidx[j] >= h->T) return fail(RSCM_ERR_INVALID, "observation %d: time index %d out of range", j, obs_tidx[j]);
        if (obs_tidx[j] > h->time_index) uncomputed = true;  // NaN there -> skipped by extract_outputs -> missing time -> Err
        if (j > 0 && obs_var[j] != obs_var[j - 1])
            for (int32_t k = 0; k < j; ++k)
                if (obs_var[k] == obs_var[j])
                    return fail(RSCM_ERR_INVALID, "observations must be grouped by variable");
    }
    if (int rc = set_device(h)) return rc;
    if (!h->d_loglik) HIPCHK(rscm::dev_malloc(&h->d_loglik, (size_t)h->N * sizeof(double)));
    if (uncomputed) {
        HIPCHK(rscm::launch_fill(h->d_loglik, h->N, -std::numeric_limits<double>::infinity(), h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        return RSCM_OK;
    }
    std::vector<const double*> ptrs(n_obs);
    for (int32_t j = 0; j < n_obs; ++j) {
        ptrs[j] = h->row_ptr(obs_var[j], obs_tidx[j]);
        if (!ptrs[j])
            return fail(RSCM_ERR_STATE, "observation %d: row %d of variable %d is not resident (NO_SERIES handle, or outside the window and the output stride)",
                        j, obs_tidx[j], obs_var[j]);
    }
    void* d_blob = nullptr;
    const size_t sz_ptr = (size_t)n_obs * sizeof(double*), sz_i = (size_t)n_obs * sizeof(int32_t),
                 sz_d = (size_t)n_obs * sizeof(double);
    const size_t off_val = sz_ptr, off_sig = off_val + sz_d, off_grp = off_sig + sz_d;
    std::vector<unsigned char> blob(off_grp + sz_i + 8);
    if (n_obs > 0) {
        memcpy(blob.data(), ptrs.data(), sz_ptr);
        memcpy(blob.data() + off_val, obs_value, sz_d);
        memcpy(blob.data() + off_sig, obs_sigma, sz_d);
        memcpy(blob.data() + off_grp, obs_var, sz_i);
    }
    HIPCHK(rscm::dev_malloc(&d_blob, blob.size()));
    hipError_t e = hipMemcpyAsync(d_blob, blob.data(), blob.size(), hipMemcpyHostToDevice, h->stream);
    rscm::LoglikArgs a{};
    a.n_members = h->N;
    a.n_obs = n_obs;
    a.normalize = normalize ? 1 : 0;
    a.obs_series = (const double* const*)d_blob;
    a.obs_value = (const double*)((char*)d_blob + off_val);
    a.obs_sigma = (const double*)((char*)d_blob + off_sig);
    a.obs_group = (const int32_t*)((char*)d_blob + off_grp);
    a.out = h->d_loglik;
    if (e == hipSuccess) e = rscm::launch_loglik(a, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    (void)hipFree(d_blob);
    if (e != hipSuccess) return fail(RSCM_ERR_DEVICE, "loglik: %s", hipGetErrorString(e));
    return RSCM_OK;
}

int rscm_ens_loglik(rscm_ens* h, int32_t n_obs, const int32_t* obs_var, const int32_t* obs_tidx,
                    const double* obs_value, const double* obs_sigma, int32_t normalize, double* out)
{
    GUARD_BEGIN
    NEED(h);
    if (!out) return fail(RSCM_ERR_INVALID, "out is NULL");
    if (int rc = loglik_on_device(h, n_obs, obs_var, obs_tidx, obs_value, obs_sigma, normalize)) return rc;
    HIPCHK(hipMemcpy(out, h->d_loglik, (size_t)h->N * sizeof(double), hipMemcpyDeviceToHost));
    return RSCM_OK;
    GUARD_END
}

int rscm_ens_loglik_device(rscm_ens* h, int32_t n_obs, const int32_t* obs_var, const int32_t* obs_tidx,
                           const double* obs_value, const double* obs_sigma, int32_t normalize, void** out_dev)
{
    GUARD_BEGIN
    NEED(h);
    if (!out_dev) return fail(RSCM_ERR_INVALID, "out_dev is NULL");
    *out_dev = nullptr;
    if (int rc = loglik_on_device(h, n_obs, obs_var, obs_tidx, obs_value, obs_sigma, normalize)) return rc;
    *out_dev = h->d_loglik;
    return RSCM_OK;
    GUARD_END
}

int rscm_ens_status_devptr(rscm_ens* h, void** out)
{
    NEED(h);
    if (!out) return fail(RSCM_ERR_INVALID, "out is NULL");
    *out = h->d_status;
    return RSCM_OK;
}


// Validate a set of observations for the fused run+likelihood kernel and keep it on the device
// (h->d_obs): groups of one variable each, ascending time indices inside a group.
int prepare_obs(rscm_ens* h, int32_t n_obs, const int32_t* obs_var, const int32_t* obs_tidx,
                const double* obs_value, const double* obs_sigma, int32_t normalize)
{
    if (h->kind != RSCM_KIND_TWO_LAYER) return fail(RSCM_ERR_INVALID, "run_loglik supports the two-layer kind");
    if (h->windowed) return fail(RSCM_ERR_INVALID, "run_loglik writes no series: use a plain or RSCM_FLAG_NO_SERIES handle, not a windowed one");
    if (n_obs < 0 || (n_obs > 0 && (!obs_var || !obs_tidx || !obs_value || !obs_sigma)))
        return fail(RSCM_ERR_INVALID, "bad observation arrays");
    const int32_t first_var = n_obs > 0 ? obs_var[0] : RSCM_TL_VAR_TS;
    for (int32_t j = 0; j < n_obs; ++j) {
        if (obs_var[j] != RSCM_TL_VAR_TS && obs_var[j] != RSCM_TL_VAR_TD)
            return fail(RSCM_ERR_INVALID, "observation %d: variable %d has no stored series", j, obs_var[j]);
        if (obs_tidx[j] < 0 || obs_tidx[j] >= h->T) return fail(RSCM_ERR_INVALID, "observation %d: time index %d out of range", j, obs_tidx[j]);
        if (j > 0 && obs_var[j] != obs_var[j - 1] && obs_var[j] == first_var)
            return fail(RSCM_ERR_INVALID, "observations must be grouped by variable");
        if (j > 0 && obs_var[j] == obs_var[j - 1] && obs_tidx[j] < obs_tidx[j - 1])
            return fail(RSCM_ERR_INVALID, "run_loglik needs ascending time indices inside a variable group "
                                          "(use rscm_ens_run + rscm_ens_loglik for arbitrary order)");
    }
    if (int rc = set_device(h)) return rc;
    // merge the (at most two) groups by time index; ties keep the first group's variable first
    std::vector<int32_t> order(n_obs);
    for (int32_t j = 0; j < n_obs; ++j) order[j] = j;
    std::stable_sort(order.begin(), order.end(), [&](int32_t x, int32_t y) { return obs_tidx[x] < obs_tidx[y]; });
    const size_t sz_i = (size_t)n_obs * sizeof(int32_t), sz_d = (size_t)n_obs * sizeof(double);
    std::vector<unsigned char> blob(2 * sz_d + 2 * sz_i + 16);
    double* bv = (double*)blob.data();
    double* bs = bv + n_obs;
    int32_t* bt = (int32_t*)(bs + n_obs);
    int32_t* bd = bt + n_obs;
    for (int32_t k = 0; k < n_obs; ++k) {
        const int32_t j = order[k];
        bv[k] = obs_value[j];
        bs[k] = obs_sigma[j];
        bt[k] = obs_tidx[j];
        bd[k] = obs_var[j] == RSCM_TL_VAR_TD ? 1 : 0;
    }
    HIPCHK(hipStreamSynchronize(h->stream));  // a launch may still be reading the previous plan
    if (blob.size() > h->obs_capacity) {
        HIPCHK(hipFree(h->d_obs));
        h->d_obs = nullptr;
        h->obs_capacity = 0;
        HIPCHK(rscm::dev_malloc(&h->d_obs, blob.size()));
        h->obs_capacity = blob.size();
    }
    HIPCHK(hipMemcpy(h->d_obs, blob.data(), blob.size(), hipMemcpyHostToDevice));
    h->obs_n = n_obs;
    h->obs_last_tidx = 0;
    for (int32_t j = 0; j < n_obs; ++j) h->obs_last_tidx = std::max(h->obs_last_tidx, obs_tidx[j]);
    h->obs_normalize = normalize ? 1 : 0;
    h->obs_first_is_deep = first_var == RSCM_TL_VAR_TD ? 1 : 0;
    if (!h->d_loglik) HIPCHK(rscm::dev_malloc(&h->d_loglik, (size_t)h->N * sizeof(double)));
    return RSCM_OK;
}

// Everything rscm_ens_run_loglik needs of the handle besides the observations.
int check_loglik_ready(rscm_ens* h)
{
    if (h->time_index != 0) return fail(RSCM_ERR_STATE, "run_loglik starts from time index 0 (call rscm_ens_rewind)");
    if (!h->params_set) return fail(RSCM_ERR_STATE, "parameters not set");
    if (h->n_linked) return fail(RSCM_ERR_STATE, "the fused run+likelihood launch takes no linked inputs: use rscm_ens_run and rscm_ens_loglik");
    if (!h->forcing_set) return fail(RSCM_ERR_STATE, "shared input series not set");
    for (int32_t v = 1; v < h->V; ++v)
        if (h->is_state(v) && !h->initial_set[v])
            return fail(RSCM_ERR_STATE, "state variable %d has no initial value (MissingInitialValue)", v);
    if (int rc = set_device(h)) return rc;
    return refresh_schedule(h);
}

// Asynchronous fused run+likelihood launch with the prepared observations; fills h->d_loglik.
hipError_t launch_loglik(rscm_ens* h)
{
    // Steps after the last observed index cannot change ln L (GaussianLikelihood reads the model at the
    // observation times only, likelihood.rs:206-226): a caller that uses nothing but ln L -- the device
    // sampler -- lets the launch end there (1850-2020 observations on a 1750-2500 axis: 270 of 750 steps).
    const int32_t len = h->loglik_stop_at_last_obs ? std::max(1, std::min(h->T - 1, h->obs_last_tidx)) : h->T - 1;
    const size_t lds_bytes = (size_t)h->n_scen * (size_t)len * sizeof(double);
    rscm::TwoLayerArgs a{};
    a.n_members = h->N;
    a.row_stride = h->N;
    a.n_times = h->T;
    a.step_begin = 0;
    a.step_end = len;
    a.n_scen = h->n_scen;
    a.src_off = h->source == RSCM_SRC_UPSTREAM ? 1 : 0;
    a.lds_forcing = lds_bytes <= (size_t)rscm::kMaxLds - 1024 ? 1 : 0;
    a.params = h->d_params;
        a.uniform_rows = h->uniform_rows;
    a.forcing = h->d_forcing;
    a.scen = h->d_scen;
    a.nsub = h->d_nsub_tl;
    a.h = h->h_tl;
    a.h_half = h->h_tl / 2.0;
    a.h_sixth = h->h_tl / 6.0;
    a.ts = h->series(RSCM_TL_VAR_TS);
    a.td = h->series(RSCM_TL_VAR_TD);
    a.status = h->d_status;
    a.n_obs = h->obs_n;
    a.normalize = h->obs_normalize;
    a.first_is_deep = h->obs_first_is_deep;
    a.obs_value = (const double*)h->d_obs;
    a.obs_sigma = a.obs_value + h->obs_n;
    a.obs_tidx = (const int32_t*)(a.obs_sigma + h->obs_n);
    a.obs_is_deep = a.obs_tidx + h->obs_n;
    a.loglik = h->d_loglik;
    return rscm::launch_two_layer_loglik(a, h->mode, h->stream);
}


static int run_loglik_impl(rscm_ens* h, int32_t n_obs, const int32_t* obs_var, const int32_t* obs_tidx,
                           const double* obs_value, const double* obs_sigma, int32_t normalize, double* out_host)
{
    if (int rc = prepare_obs(h, n_obs, obs_var, obs_tidx, obs_value, obs_sigma, normalize)) return rc;
    if (int rc = check_loglik_ready(h)) return rc;
    hipError_t e = hipEventRecord(h->ev0, h->stream);
    if (e == hipSuccess) e = launch_loglik(h);
    if (e == hipSuccess) e = hipEventRecord(h->ev1, h->stream);
    if (e == hipSuccess && out_host)
        e = hipMemcpyAsync(out_host, h->d_loglik, (size_t)h->N * sizeof(double), hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess) return fail(RSCM_ERR_DEVICE, "run_loglik: %s", hipGetErrorString(e));
    h->timed = true;
    return RSCM_OK;
}

int rscm_ens_run_loglik(rscm_ens* h, int32_t n_obs, const int32_t* obs_var, const int32_t* obs_tidx,
                        const double* obs_value, const double* obs_sigma, int32_t normalize, double* out)
{
    GUARD_BEGIN
    NEED(h);
    if (!out) return fail(RSCM_ERR_INVALID, "out is NULL");
    return run_loglik_impl(h, n_obs, obs_var, obs_tidx, obs_value, obs_sigma, normalize, out);
    GUARD_END
}

int rscm_ens_run_loglik_device(rscm_ens* h, int32_t n_obs, const int32_t* obs_var, const int32_t* obs_tidx,
                               const double* obs_value, const double* obs_sigma, int32_t normalize, void** out_dev)
{
    GUARD_BEGIN
    NEED(h);
    if (!out_dev) return fail(RSCM_ERR_INVALID, "out_dev is NULL");
    *out_dev = nullptr;
    if (int rc = run_loglik_impl(h, n_obs, obs_var, obs_tidx, obs_value, obs_sigma, normalize, nullptr)) return rc;
    *out_dev = h->d_loglik;
    return RSCM_OK;
    GUARD_END
}

int rscm_gpu_copy_to_device(int32_t device_id, void* device_ptr, const void* host, int64_t n_bytes)
{
    GUARD_BEGIN
    if (n_bytes < 0 || (n_bytes > 0 && (!host || !device_ptr))) return fail(RSCM_ERR_INVALID, "bad arguments");
    if (n_bytes == 0) return RSCM_OK;
    HIPCHK(hipSetDevice(device_id));
    HIPCHK(hipMemcpy(device_ptr, host, (size_t)n_bytes, hipMemcpyHostToDevice));
    return RSCM_OK;
    GUARD_END
}

int rscm_ens_summary_series(rscm_ens* h, int32_t var_id, int32_t t_begin, int32_t t_end, double* out)
{
    GUARD_BEGIN
    NEED(h);
    if (var_id < 1 || var_id >= h->V || t_begin < 0 || t_end > h->T || t_begin > t_end || !out)
        return fail(RSCM_ERR_INVALID, "bad variable / time range");
    if (h->windowed || h->rows != h->T) return fail(RSCM_ERR_STATE, "this handle does not store whole series (RSCM_FLAG_NO_SERIES / RSCM_FLAG_WINDOWED): use rscm_ens_summary row by row");
    const int32_t n_rows = t_end - t_begin;
    // rows beyond the current time index have not been computed: empty summaries, as rscm_ens_summary
    const int32_t computed = std::max(0, std::min(t_end, h->time_index + 1) - t_begin);
    for (int32_t r = computed; r < n_rows; ++r) {
        out[4 * r + 0] = 0.0; out[4 * r + 1] = 0.0;
        out[4 * r + 2] = std::numeric_limits<double>::infinity();
        out[4 * r + 3] = -std::numeric_limits<double>::infinity();
    }
    if (computed == 0) return RSCM_OK;
    if (int rc = set_device(h)) return rc;
    const int32_t nb = rscm::summary_blocks(h->N);
    double* d_partial = nullptr;
    double* d_out = nullptr;
    HIPCHK(rscm::dev_malloc(&d_partial, (size_t)computed * nb * 4 * sizeof(double)));
    hipError_t e = rscm::dev_malloc(&d_out, (size_t)computed * 4 * sizeof(double));
    if (e == hipSuccess)
        e = rscm::launch_summary_rows(h->series(var_id) + (size_t)t_begin * h->N, h->N, computed, d_partial, nb, d_out, h->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(out, d_out, (size_t)computed * 4 * sizeof(double), hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    (void)hipFree(d_partial);
    (void)hipFree(d_out);
    if (e != hipSuccess) return fail(RSCM_ERR_DEVICE, "summary_series: %s", hipGetErrorString(e));
    return RSCM_OK;
    GUARD_END
}

int rscm_ens_quantile_series(rscm_ens* h, int32_t var_id, int32_t t_begin, int32_t t_end, int32_t n_q, const double* q, double* out,
                             double* count)
{
    GUARD_BEGIN
    NEED(h);
    if (var_id < 1 || var_id >= h->V || t_begin < 0 || t_end > h->T || t_begin > t_end || !out || n_q < 1 || !q)
        return fail(RSCM_ERR_INVALID, "bad variable / time range / quantile list");
    for (int32_t k = 0; k < n_q; ++k)
        if (!(q[k] >= 0.0 && q[k] <= 1.0)) return fail(RSCM_ERR_INVALID, "Quantiles must be in the range [0, 1], got %g", q[k]);
    if (h->windowed || h->rows != h->T) return fail(RSCM_ERR_STATE, "this handle does not store whole series (RSCM_FLAG_NO_SERIES / RSCM_FLAG_WINDOWED)");
    const int32_t n_rows = t_end - t_begin;
    const int32_t computed = std::max(0, std::min(t_end, h->time_index + 1) - t_begin);
    for (int32_t r = computed; r < n_rows; ++r) {
        for (int32_t k = 0; k < n_q; ++k) out[(size_t)r * n_q + k] = std::numeric_limits<double>::quiet_NaN();
        if (count) count[r] = 0.0;
    }
    if (computed == 0) return RSCM_OK;
    if (int rc = set_device(h)) return rc;
    double* d_q = nullptr;
    double* d_out = nullptr;
    std::vector<double> host((size_t)computed * (n_q + 1));
    HIPCHK(rscm::dev_malloc(&d_q, (size_t)n_q * sizeof(double)));
    hipError_t e = rscm::dev_malloc(&d_out, host.size() * sizeof(double));
    if (e == hipSuccess) e = hipMemcpyAsync(d_q, q, (size_t)n_q * sizeof(double), hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess) e = rscm::launch_quantile_rows(h->series(var_id) + (size_t)t_begin * h->N, h->N, computed, d_q, n_q, d_out, h->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(host.data(), d_out, host.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    (void)hipFree(d_q);
    (void)hipFree(d_out);
    if (e != hipSuccess)
        return fail(e == hipErrorOutOfMemory ? RSCM_ERR_NOMEM : RSCM_ERR_DEVICE, "quantile_series: %s", hipGetErrorString(e));
    for (int32_t r = 0; r < computed; ++r) {
        if (count) count[r] = host[(size_t)r * (n_q + 1)];
        for (int32_t k = 0; k < n_q; ++k) out[(size_t)r * n_q + k] = host[(size_t)r * (n_q + 1) + 1 + k];
    }
    return RSCM_OK;
    GUARD_END
}

int rscm_ens_summary(rscm_ens* h, int32_t var_id, int32_t tidx, double out[4])
{
    GUARD_BEGIN
    NEED(h);
    if (var_id < 1 || var_id >= h->V || tidx < 0 || tidx >= h->T || !out)
        return fail(RSCM_ERR_INVALID, "bad variable/time index");
    if (tidx > h->time_index) {
        out[0] = 0.0; out[1] = 0.0;
        out[2] = std::numeric_limits<double>::infinity();
        out[3] = -std::numeric_limits<double>::infinity();
        return RSCM_OK;
    }
    if (int rc = set_device(h)) return rc;
    const int32_t nb = rscm::summary_blocks(h->N);
    const double* row = h->row_ptr(var_id, tidx);
    if (!row) return fail(RSCM_ERR_STATE, "row %d of variable %d is not resident on this handle", tidx, var_id);
    HIPCHK(rscm::launch_summary(row, h->N, h->d_partial, nb, h->d_out4, h->stream));
    HIPCHK(hipMemcpyAsync(out, h->d_out4, 4 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return RSCM_OK;
    GUARD_END
}

int rscm_ens_sample_lhs(rscm_ens* h, uint64_t seed, const double* low, const double* high,
                        int64_t member_offset, int64_t n_total)
{
    GUARD_BEGIN
    NEED(h);
    if (!low || !high) return fail(RSCM_ERR_INVALID, "low/high is NULL");
    if (member_offset < 0 || n_total < 1 || member_offset + h->N > n_total)
        return fail(RSCM_ERR_INVALID, "member block [%lld, %lld) outside the global ensemble of %lld",
                    (long long)member_offset, (long long)(member_offset + h->N), (long long)n_total);
    if (int rc = set_device(h)) return rc;
    if (h->kind != RSCM_KIND_N2O_CHEMISTRY)   // (N2O's delay may vary over the members: the look-back is sized for the largest)
        for (int32_t j = 0; j < h->P; ++j)
            if (const char* name = structural_row(h->kind, j))
                if (low[j] != high[j])
                    return fail(RSCM_ERR_INVALID, "parameter row %d (%s) is structural: low must equal high", j, name);
    if (h->kind == RSCM_KIND_UDEB)
        if (int rc = configure_udeb(h, 1, [&](int j, int64_t) { return low[j]; })) return rc;
    if (h->kind == RSCM_KIND_OCEAN_CARBON)
        if (int rc = configure_ocean(h, 1, [&](int j, int64_t) { return low[j]; })) return rc;
    if (h->kind == RSCM_KIND_N2O_CHEMISTRY) h->lookback = (int32_t)std::min(std::max(1.0, high[4]), 1e6) + 1;
    if (h->kind == RSCM_KIND_GHG_FORCING) {
        const double m = low[RSCM_GH_P_METHOD];
        if (m != 0.0 && m != 1.0) return fail(RSCM_ERR_INVALID, "GhgForcing method must be 0 or 1");
        h->ghg_method = (int32_t)m;
    }
    double* d_lh = nullptr;
    HIPCHK(rscm::dev_malloc(&d_lh, 2 * (size_t)h->P * sizeof(double)));
    hipError_t e = hipMemcpyAsync(d_lh, low, (size_t)h->P * sizeof(double), hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_lh + h->P, high, (size_t)h->P * sizeof(double), hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess) e = rscm::launch_lhs(h->d_params, h->P, h->N, seed, d_lh, d_lh + h->P, member_offset, n_total, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    (void)hipFree(d_lh);
    if (e != hipSuccess) return fail(RSCM_ERR_DEVICE, "sample_lhs: %s", hipGetErrorString(e));
    h->uniform_rows = 0;   // conservatively: low + u (high - low) need not reproduce low's bits for every u
    h->derived_dirty = true;
    h->params_set = true;
    return RSCM_OK;
    GUARD_END
}

int rscm_ens_get_params(rscm_ens* h, double* out_soa)
{
    GUARD_BEGIN
    NEED(h);
    if (!out_soa) return fail(RSCM_ERR_INVALID, "out is NULL");
    if (int rc = set_device(h)) return rc;
    HIPCHK(hipMemcpyAsync(out_soa, h->d_params, (size_t)h->P * h->N * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return RSCM_OK;
    GUARD_END
}

int rscm_gpu_copy_to_host(int32_t device_id, void* host, const void* device_ptr, int64_t n_bytes)
{
    GUARD_BEGIN
    if (n_bytes < 0 || (n_bytes > 0 && (!host || !device_ptr))) return fail(RSCM_ERR_INVALID, "bad arguments");
    if (n_bytes == 0) return RSCM_OK;
    HIPCHK(hipSetDevice(device_id));
    HIPCHK(hipMemcpy(host, device_ptr, (size_t)n_bytes, hipMemcpyDeviceToHost));
    return RSCM_OK;
    GUARD_END
}

int rscm_gpu_host_alloc(int64_t n_bytes, void** out)
{
    GUARD_BEGIN
    if (!out || n_bytes <= 0) return fail(RSCM_ERR_INVALID, "bad arguments");
    *out = nullptr;
    HIPCHK(hipHostMalloc(out, (size_t)n_bytes, hipHostMallocDefault));
    return RSCM_OK;
    GUARD_END
}

int rscm_gpu_host_free(void* p)
{
    GUARD_BEGIN
    if (p) HIPCHK(hipHostFree(p));
    return RSCM_OK;
    GUARD_END
}

int rscm_gpu_ocean_fit_selftest(int32_t model, double irf_scale, double irf_switch_time, int64_t max_history_months,
                                double* max_error, int32_t* n_modes, int32_t* near_lags, int32_t* n_exit, double* max_abs_coefficient)
{
    GUARD_BEGIN
    if (!max_error || !n_modes || !near_lags || !n_exit) return fail(RSCM_ERR_INVALID, "NULL output");
    if (model < 0 || model > 2 || max_history_months < 1 || max_history_months > 10000000)
        return fail(RSCM_ERR_INVALID, "bad model or history length");
    const std::vector<double> tab = ocean_irf_table(model, irf_scale, irf_switch_time, max_history_months);
    rscm::OceanModes m{};
    int32_t near = 0;
    *max_error = ocean_fit_modes(tab, model, irf_switch_time, max_history_months, &near, &m);
    *n_modes = m.n_modes;
    *near_lags = near;
    *n_exit = m.n_exit;
    if (max_abs_coefficient) {
        *max_abs_coefficient = 0.0;
        for (int q = 0; q < m.n_modes; ++q) *max_abs_coefficient = std::max(*max_abs_coefficient, std::fabs(m.c[q]));
    }
    return RSCM_OK;
    GUARD_END
}

int rscm_ens_ocean_fast_info(rscm_ens* h, int32_t* uses_recurrence, double* fit_error)
{
    NEED(h);
    if (h->kind != RSCM_KIND_OCEAN_CARBON || !h->ocean_ready) return fail(RSCM_ERR_STATE, "not a configured OceanCarbon ensemble");
    if (uses_recurrence) *uses_recurrence = h->ocean_recur_ok ? 1 : 0;
    if (fit_error) *fit_error = h->ocean_fit_error;
    return RSCM_OK;
}

int rscm_gpu_selftest_div(int32_t device_id, int64_t n, const double* num, const double* den,
                          double* out_ref, double* out_fast, uint8_t* used_fast)
{
    GUARD_BEGIN
    if (n < 0 || !num || !den || !out_ref || !out_fast || !used_fast) return fail(RSCM_ERR_INVALID, "bad arguments");
    if (n == 0) return RSCM_OK;
    HIPCHK(hipSetDevice(device_id));
    double* d = nullptr;
    uint8_t* du = nullptr;
    HIPCHK(rscm::dev_malloc(&d, 4 * (size_t)n * sizeof(double)));
    hipError_t e = rscm::dev_malloc(&du, (size_t)n);
    if (e == hipSuccess) e = hipMemcpy(d, num, (size_t)n * sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d + n, den, (size_t)n * sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = rscm::launch_divtest(d, d + n, d + 2 * n, d + 3 * n, du, n, nullptr);
    if (e == hipSuccess) e = hipMemcpy(out_ref, d + 2 * n, (size_t)n * sizeof(double), hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(out_fast, d + 3 * n, (size_t)n * sizeof(double), hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(used_fast, du, (size_t)n, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    (void)hipFree(du);
    if (e != hipSuccess) return fail(RSCM_ERR_DEVICE, "selftest_div: %s", hipGetErrorString(e));
    return RSCM_OK;
    GUARD_END
}

}  // extern "C"
