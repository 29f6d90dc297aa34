// The reference's table-driven probit functions (util/Random.h:95-124) for the ALS / MCMC learners:
//   fast_pnorm  -- Phi by linear interpolation on a 2861-point grid x_i = i / HINV over [0, 5.2003...], saturating above it
//   fast_dpnorm -- dnorm(x) / (1 - pnorm(x)) on the 40001-point grid -3 + i * 2e-4 over [-3, 5]; 0 below, an asymptote above
// The reference ships the grids as literals (util/RandomData.h, RandomData_.h); here they are generated once per engine on
// the host from their defining formulas (the ratio with the shipped values' own cancellation and 12-decimal rounding) and
// uploaded.  Agreement with the shipped values: 7e-16 / 2e-11 (the same construction is pinned against them on the CPU,
// tests/golden/probit_tables.json).
#pragma once
#include <cmath>

namespace fmx {

constexpr int PN_POINTS = 2861;
constexpr int DP_POINTS = 40001;
constexpr double PN_MAX = 5.20031455849973;
constexpr double PN_HINV = 549.966731401936;

struct ProbitTables {
  const double* pn_y;  // [PN_POINTS + 1]
  const double* dp_y;  // [DP_POINTS + 1]
};

__host__ __device__ inline double pn_x(int i) { return (double)i / PN_HINV; }
__host__ __device__ inline double dp_x(int i) { return (double)(-30000 + 2 * i) / 10000.0; }

// util/Random.h:95-111
__device__ __forceinline__ double fast_pnorm(const double* __restrict__ y, double x) {
  const double ax = x < 0 ? -x : x;
  double res;
  if (ax > PN_MAX) {
    res = 0.999999900524235;
  } else {
    const int i = (int)(ax * PN_HINV);
    const double w = (ax - pn_x(i)) * PN_HINV;
    res = w * y[i + 1] + (1.0 - w) * y[i];
  }
  return ax == x ? res : 1.0 - res;
}

// util/Random.h:113-124
__device__ __forceinline__ double fast_dpnorm(const double* __restrict__ y, double x) {
  const double ax = x < 0 ? -x : x;
  if (x < -3.0) return 0.0;
  if (x > 5.0) return 0.1943369 + 0.9754752 * x + 0.4136861 * sqrt(ax) - 0.5034295 * log(ax + 1e-07);
  const int i = (int)((x - -3.0) * 5000);
  const double w = (x - dp_x(i)) * 5000;
  return w * y[i + 1] + (1.0 - w) * y[i];
}

}  // namespace fmx
