// Host-side geometry tables for the ClimateUDEB kernel (uniform over an ensemble): area factors
// (parameters/climate_udeb.rs compute_area_factors, ocean_area_at_depth), 1 - relative depth of
// layer_diffusivities (ocean_column.rs), and the CMIP5 initial ocean temperature profiles
// (data: climate_udeb.rs CMIP5_PROFILE_NH / CMIP5_PROFILE_SH).
#pragma once

#include <vector>

namespace rscm {

inline const double* cmip5_profile(int hemi)
{
    static const double NH[50] = {
        1.89503822e+01, 1.58484640e+01, 1.27692938e+01, 1.11237631e+01, 9.93378544e+00, 8.89700890e+00,
        8.01173782e+00, 7.24060631e+00, 6.58022213e+00, 5.99888515e+00, 5.47700644e+00, 5.02416515e+00,
        4.62269211e+00, 4.27446032e+00, 3.95875454e+00, 3.70120311e+00, 3.47130036e+00, 3.26678157e+00,
        3.08187413e+00, 2.93045211e+00, 2.79141068e+00, 2.66952801e+00, 2.55478907e+00, 2.44816899e+00,
        2.35198379e+00, 2.26331019e+00, 2.18005610e+00, 2.10292435e+00, 2.02744699e+00, 1.95637441e+00,
        1.89118743e+00, 1.82867718e+00, 1.76954043e+00, 1.71074319e+00, 1.65469503e+00, 1.60236323e+00,
        1.55269921e+00, 1.50864816e+00, 1.47147048e+00, 1.44045138e+00, 1.41173756e+00, 1.38347185e+00,
        1.35783422e+00, 1.33539736e+00, 1.31498563e+00, 1.29516900e+00, 1.27472460e+00, 1.25263810e+00,
        1.22954643e+00, 1.20586693e+00};
    static const double SH[50] = {
        1.62849369e+01, 1.35041571e+01, 1.10637445e+01, 9.45342350e+00, 8.30402851e+00, 7.37928152e+00,
        6.60113478e+00, 5.90550613e+00, 5.29829597e+00, 4.77080584e+00, 4.31242418e+00, 3.93976259e+00,
        3.62348270e+00, 3.35576391e+00, 3.11617875e+00, 2.93644977e+00, 2.77795982e+00, 2.63738632e+00,
        2.50925493e+00, 2.40222931e+00, 2.30221725e+00, 2.21322107e+00, 2.12794638e+00, 2.04543614e+00,
        1.96889246e+00, 1.89580762e+00, 1.82651293e+00, 1.75886285e+00, 1.69188118e+00, 1.62586987e+00,
        1.56049752e+00, 1.49373257e+00, 1.42720032e+00, 1.35796928e+00, 1.28947854e+00, 1.22542751e+00,
        1.16357803e+00, 1.10515058e+00, 1.05139232e+00, 1.00322735e+00, 9.58882809e-01, 9.15422320e-01,
        8.75476420e-01, 8.43416333e-01, 8.16016912e-01, 7.90101945e-01, 7.68699825e-01, 7.51805604e-01,
        7.36583769e-01, 7.25481987e-01};
    return hemi == 0 ? NH : SH;
}

inline double ocean_area_at_depth(double depth_m, double depth_dependent_area)
{
    static const double DEPTH[12] = {0.0, 200.0, 500.0, 1000.0, 1500.0, 2000.0, 2500.0, 3000.0, 3500.0, 4000.0, 4500.0, 5000.0};
    static const double AREA[12] = {1.0, 0.975, 0.95, 0.92, 0.91, 0.87, 0.81, 0.72, 0.55, 0.38, 0.18, 0.05};
    double hydro;
    if (depth_m <= DEPTH[0]) hydro = AREA[0];
    else if (depth_m >= DEPTH[11]) hydro = AREA[11];
    else {
        hydro = AREA[0];
        for (int i = 1; i < 12; ++i)
            if (depth_m <= DEPTH[i]) {
                const double frac = (depth_m - DEPTH[i - 1]) / (DEPTH[i] - DEPTH[i - 1]);
                hydro = AREA[i - 1] + frac * (AREA[i] - AREA[i - 1]);
                break;
            }
    }
    return 1.0 + depth_dependent_area * (hydro - 1.0);
}

// One row of six values per layer, [n][6] = {af_top, af_bot, af_diff, 1 - relative depth, G_nh, G_sh} (a solve reads its
// tables row by row: the values of a few rows are one contiguous scalar load), where G folds the initial
// profile and the polar sinking temperature (ClimateUDEBState::new: 1.0) into the weight the
// profile-advection term of row l carries (ocean_column.rs step_hemisphere):
//   G[0]     = (init[1] - T_polar) * af_bot[0]
//   G[l]     = init[l+1]*af_bot[l] - init[l]*af_top[l] + T_polar*af_diff[l]      0 < l < n-1
//   G[n-1]   = (T_polar - init[n-1]) * af_top[n-1]
// The BOTTOM row's af_bot is stored as 0 and its af_diff as af_top[n-1]: the last row of the tridiagonal system has no lower
// neighbour (ocean_column.rs:188-198: b = 1 + (diff_up + upwell) af_top, no c, entrainment weighted by af_top instead of
// af_diff), which is exactly what an interior row's formulas give with those two values -- 1 + tdd*0 is 1, c' is 0 -- so the
// kernels run ONE row formula for every row below the mixed layer and never ask which row is the last.  That is what lets the
// layer count be a run-time value in an unrolled sweep (udeb_body.hpp): rows past the end read all-zero table rows and stay
// exact no-ops (b = 1, c' = 0, d = the row's own zero).
constexpr int kUdebTableCols = 6;
inline std::vector<double> udeb_tables(int n, double dz_mix, double dz, double depth_dependent_area)
{
    std::vector<double> aft((size_t)n), afb((size_t)n), afd((size_t)n), omr((size_t)n, 0.0);
    for (int l = 0; l < n; ++l) {
        double z_top, z_bottom;
        if (l == 0) { z_top = 0.0; z_bottom = dz_mix; }
        else { z_top = dz_mix + ((double)l - 1.0) * dz; z_bottom = z_top + dz; }
        const double a_top = ocean_area_at_depth(z_top, depth_dependent_area);
        const double a_bottom = ocean_area_at_depth(z_bottom, depth_dependent_area);
        const double a_avg = (a_top + a_bottom) / 2.0;
        aft[l] = a_top / a_avg;
        afb[l] = a_bottom / a_avg;
        afd[l] = (a_top - a_bottom) / a_avg;
    }
    const double total_depth = dz_mix + ((double)n - 1.0) * dz;
    for (int l = 0; l < n - 1; ++l) {
        const double depth = dz_mix + (double)l * dz;
        omr[l] = 1.0 - depth / total_depth;
    }
    std::vector<double> t((size_t)kUdebTableCols * n, 0.0);
    const double t_polar = 1.0;
    for (int l = 0; l < n; ++l) {
        t[(size_t)l * kUdebTableCols + 0] = aft[l];
        t[(size_t)l * kUdebTableCols + 1] = afb[l];
        t[(size_t)l * kUdebTableCols + 2] = afd[l];
        t[(size_t)l * kUdebTableCols + 3] = omr[l];
    }
    for (int hemi = 0; hemi < 2; ++hemi) {
        auto init = [&](int l) { return l < 50 ? cmip5_profile(hemi)[l] : cmip5_profile(hemi)[49]; };
        auto G = [&](int l) -> double& { return t[(size_t)l * kUdebTableCols + 4 + hemi]; };
        G(0) = (init(1) - t_polar) * afb[0];
        for (int l = 1; l < n - 1; ++l) G(l) = init(l + 1) * afb[l] - init(l) * aft[l] + t_polar * afd[l];
        G(n - 1) = (t_polar - init(n - 1)) * aft[n - 1];
    }
    t[(size_t)(n - 1) * kUdebTableCols + 1] = 0.0;            // (after G, which is built from the true area factors)
    t[(size_t)(n - 1) * kUdebTableCols + 2] = aft[n - 1];
    return t;
}

}  // namespace rscm
