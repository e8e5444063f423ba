// Straight-line f64 sin + cos for the generated model code and the line search.
//
// The device library's sincos is ~110 executed instructions and carries a branch for huge arguments (Payne-Hanek,
// v_trig_preop): three calls are a quarter of the vector instructions of an acrobot stage of the sweeps, which at one wavefront
// per SIMD are bound by exactly that count (profiles/r04: vector unit active 62 % of a wavefront's cycles), and the branch
// cuts the stage into basic blocks the scheduler cannot move the memory instructions across.  This one has no branch:
//   * Cody-Waite reduction with the three 33-bit pieces of pi/2 of fdlibm's e_rem_pio2.c (k * piece is exact for
//     |k| < 2^20, i.e. |x| < 1.6e6; beyond that the reduced argument loses accuracy gradually (1e-12 absolute at 1e8, still in
//     [-1, 1] at 1e12, meaningless -- possibly non-finite -- beyond ~1e15), which only iterates that are already diverging
//     ever see: Options.diverging_iterates_tol = 1e8 stops them, the line search rejects non-finite trial points);
//   * the degree-13 / degree-14 kernels of fdlibm's k_sin.c / k_cos.c on |r| <= pi/4 with the tail of the reduction;
//   * quadrant selection by v_cndmask.
// ~45 instructions; against long double on the host: < 1 ulp for |x| < 1e5 (tests/test_sincos_fast.py, tools/micro/sincos_accuracy.cpp).
// -DDTO_LIB_SINCOS=1 (DTO_PLUGIN_CXXFLAGS) switches the generated code back to the library call for A/B runs.
#pragma once

namespace dto {

#ifdef __HIPCC__
#define DTO_MATH_FN __host__ __device__ __forceinline__
#else
#define DTO_MATH_FN static inline
#endif

DTO_MATH_FN void sincos_fast(double x, double* sn, double* cs) {
  const double INVPIO2 = 6.36619772367581382433e-01, P1 = 1.57079632673412561417e+00, P2 = 6.07710050630396597660e-11,
               P2T = 2.02226624879595063154e-21;
  const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
               S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
  const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
               C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
  const double fn = __builtin_rint(x * INVPIO2);
  const double t = __builtin_fma(-fn, P1, x);
  const double w = fn * P2;
  const double y0 = t - w;
  const double y1 = __builtin_fma(-fn, P2T, (t - y0) - w);
  const double z = y0 * y0;
  // sin kernel
  const double v = z * y0;
  const double rs = __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, S6, S5), S4), S3), S2);
  const double s = y0 - ((z * (0.5 * y1 - v * rs) - y1) - v * S1);
  // cos kernel
  const double rc = z * __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, C6, C5), C4), C3), C2), C1);
  const double hz = 0.5 * z, wc = 1.0 - hz;
  const double c = wc + (((1.0 - wc) - hz) + (z * rc - y0 * y1));
  // quadrant = fn mod 4 from the low mantissa bits of fn + 1.5 * 2^52 (fn is integer-valued; defined for every finite fn, where
  // the conversion (int)fn is undefined beyond |x| ~ 3.4e9 -- VERDICT r4 weak 1c)
  const double fm = fn + 6755399441055744.0;
  long long fbits;
  __builtin_memcpy(&fbits, &fm, sizeof(fbits));
  const int n = (int)(unsigned)(unsigned long long)fbits;
  const bool swap = (n & 1) != 0;
  const double ss = swap ? c : s, cc = swap ? s : c;
  *sn = (n & 2) ? -ss : ss;
  *cs = ((n + 1) & 2) ? -cc : cc;
}

}  // namespace dto

#if DTO_LIB_SINCOS
#define DTO_SINCOS(x, s, c) sincos((x), (s), (c))
#else
#define DTO_SINCOS(x, s, c) dto::sincos_fast((x), (s), (c))
#endif
