// KKT / interior-point driver (filled in by the KKT milestone).
#include "dto_problem.hpp"
namespace dto {
struct SolverState {};
void Problem::free_solver() { delete solver; solver = nullptr; }
}  // namespace dto
