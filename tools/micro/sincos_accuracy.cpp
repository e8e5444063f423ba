// Accuracy of dto::sincos_fast (csrc/dto_math.hpp) against long double on the host:
//   g++ -O2 -ffp-contract=off -I directtrajectoryoptimization.jl_amd/csrc tools/micro/sincos_accuracy.cpp -o /tmp/sincos_accuracy && /tmp/sincos_accuracy
#include <cmath>
#include <cstdio>
#include <random>
#include "dto_math.hpp"
int main() {
  std::mt19937_64 g(1);
  for (double range : {3.2, 50.0, 1e3, 1e5, 1.5e6, 1e8}) {
    std::uniform_real_distribution<double> u(-range, range);
    double worst = 0, worst_abs = 0;
    for (int i = 0; i < 4000000; ++i) {
      const double x = u(g);
      double s, c;
      dto::sincos_fast(x, &s, &c);
      const long double rs = sinl((long double)x), rc = cosl((long double)x);
      const double us = std::fabs((double)((s - rs) / (long double)std::fabs(std::nextafter((double)rs, 2.0) - (double)rs)));
      const double uc = std::fabs((double)((c - rc) / (long double)std::fabs(std::nextafter((double)rc, 2.0) - (double)rc)));
      worst = std::fmax(worst, std::fmax(us, uc));
      worst_abs = std::fmax(worst_abs, std::fmax(std::fabs((double)(s - rs)), std::fabs((double)(c - rc))));
    }
    std::printf("|x| < %-8g  worst error %.3f ulp, %.2e absolute\n", range, worst, worst_abs);
  }
  return 0;
}
