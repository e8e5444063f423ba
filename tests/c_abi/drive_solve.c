/*
 * A host program in plain C99 that drives one solve purely through include/dto.h (no Python, no ctypes): what a host
 * language binding (the Julia `ccall` shim of INTEGRATION.md) does.  Built and run by tests/test_c_abi_program_gpu.py.
 *
 *   drive_solve <problem.txt> <solution.bin>
 * problem.txt:  plugin path / horizon T / T stage kinds / num_variables / lower bounds / upper bounds / initial guess
 *               (whitespace separated; "inf", "-inf" allowed), written by the test from the Python mirror's Structure.
 * Mirrors: Solver(...) -> dto_problem_create (src/solver.jl:6-21), initialize_states!/controls! -> x0 (src/solver.jl:23-39),
 *          solve! -> dto_solve (src/solver.jl:45-47), MOI.eval_objective -> dto_eval_f (src/moi.jl:1-13).
 */
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "dto.h"

#define CHECK(call)                                                                  \
  do {                                                                               \
    int rc_ = (call);                                                                \
    if (rc_ != DTO_OK) {                                                             \
      fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, dto_last_error());          \
      return 2;                                                                      \
    }                                                                                \
  } while (0)

int main(int argc, char** argv) {
  if (argc < 3) return 1;
  FILE* f = fopen(argv[1], "r");
  if (!f) return 1;
  char plugin[4096];
  int T = 0;
  long nz = 0;
  if (fscanf(f, "%4095s %d", plugin, &T) != 2) return 1;
  int32_t* kinds = (int32_t*)malloc((size_t)T * sizeof(int32_t));
  for (int t = 0; t < T; ++t) { int k; if (fscanf(f, "%d", &k) != 1) return 1; kinds[t] = k; }
  if (fscanf(f, "%ld", &nz) != 1) return 1;
  double* lo = (double*)malloc((size_t)nz * sizeof(double));
  double* hi = (double*)malloc((size_t)nz * sizeof(double));
  double* x0 = (double*)malloc((size_t)nz * sizeof(double));
  for (long i = 0; i < nz; ++i) if (fscanf(f, "%lf", &lo[i]) != 1) return 1;
  for (long i = 0; i < nz; ++i) if (fscanf(f, "%lf", &hi[i]) != 1) return 1;
  for (long i = 0; i < nz; ++i) if (fscanf(f, "%lf", &x0[i]) != 1) return 1;
  fclose(f);

  dto_problem_spec spec;
  memset(&spec, 0, sizeof(spec));
  spec.abi_version = DTO_ABI_VERSION;
  spec.model_library = plugin;
  spec.horizon = T;
  spec.stage_kind = kinds;
  spec.variable_lower = lo;
  spec.variable_upper = hi;
  spec.parameters = NULL;
  spec.num_parameters = 0;
  spec.evaluate_hessian = 1;
  dto_problem* p = NULL;
  CHECK(dto_problem_create(&spec, &p));
  dto_sizes_t sz;
  CHECK(dto_sizes(p, &sz));
  if (sz.num_variables != nz) { fprintf(stderr, "num_variables mismatch\n"); return 3; }

  dto_options opt;
  CHECK(dto_options_default(&opt));             /* the reference's Options defaults (src/options.jl:6-36) */
  double* x = (double*)malloc((size_t)nz * sizeof(double));
  double* mu = (double*)malloc((size_t)(sz.num_constraint > 0 ? sz.num_constraint : 1) * sizeof(double));
  int32_t status = -1, iterations = -1;
  CHECK(dto_solve(p, &opt, x0, x, mu, &status, &iterations));
  double fval = 0.0;
  CHECK(dto_eval_f(p, x, &fval));
  printf("status %d iterations %d objective %.17g num_constraint %lld\n", (int)status, (int)iterations, fval,
         (long long)sz.num_constraint);
  FILE* o = fopen(argv[2], "wb");
  if (!o) return 1;
  fwrite(x, sizeof(double), (size_t)nz, o);
  fwrite(mu, sizeof(double), (size_t)sz.num_constraint, o);
  fclose(o);
  CHECK(dto_problem_destroy(p));
  free(kinds); free(lo); free(hi); free(x0); free(x); free(mu);
  return 0;
}
