// KKT assembly + block-tridiagonal LDL^T kernels (filled in by the KKT milestone).
#pragma once
#include <hip/hip_runtime.h>
#include "dto_model_plugin.h"

struct dto_kkt_args { int op; };
struct dto_kkt_info { int supported; };

namespace dto {
template <class M>
int launch_kkt(int, const dto_kkt_args*, void*) { return -1; }
template <class M>
int kkt_info(dto_kkt_info* out) { out->supported = 0; return 0; }
}  // namespace dto
