// Host-side layout contract: variable / constraint / nonzero ordering of the flattened NLP.
//
// Restates, in O(nnz log nnz), what the reference computes with O(T^2) prefix sums and O(nnz^2)
// `findfirst` scans:
//   dimensions                          src/dynamics.jl:206-211
//   state/action/... index vectors      src/dynamics.jl:188-204
//   constraint_indices/jacobian_indices src/dynamics.jl:162-170, src/constraints.jl:141-166,
//                                       src/general_constraint.jl:118-120
//   sparsity_jacobian                   src/dynamics.jl:129-142, src/constraints.jl:106-120,
//                                       src/general_constraint.jl:93-103
//   sparsity_hessian / hessian_indices  src/costs.jl:75-104, src/dynamics.jl:144-186,
//                                       src/constraints.jl:122-183, src/general_constraint.jl:105-139
//   NLPData totals, key = sort(unique)  src/data.jl:150-220
//   primal_bounds / constraint_bounds   src/data.jl:123-148
// All public index values are 1-based like the reference; device tables are 0-based.
#pragma once

#include <algorithm>
#include <cstdint>
#include <limits>
#include <string>
#include <utility>
#include <vector>

#include "dto_model_plugin.h"

namespace dto {

struct Layout {
  int T = 0;
  const dto_model_vtable* vt = nullptr;
  bool hessian = false;
  std::vector<int> kind, nx, nu, nw;
  // 0-based device tables, all length T+1
  std::vector<int> zoff, woff, cdoff, ccoff, jdoff, jcoff, hoff;
  int64_t Nz = 0, Nw = 0, Ndyn = 0, Nstage = 0, Ngen = 0, Nc = 0;
  int64_t nnzJd = 0, nnzJs = 0, nnzJg = 0, nnzJ = 0, nnzH_raw = 0, nnzH = 0;
  int max_nx = 0, max_nu = 0;
  std::vector<int64_t> jac_rows, jac_cols;    // 1-based COO, reference order
  std::vector<int64_t> hess_rows, hess_cols;  // 1-based key (row-major sorted unique)
  // positions (1-based) in the key of each local nonzero, per stage (indices.*_hessians)
  std::vector<std::vector<int64_t>> obj_h, dyn_h, con_h;
  std::vector<int64_t> gen_h;
  // per-kind relative scatter maps for the Hessian kernel
  int hmap_stride = 1;
  std::vector<int> hmap_cost, hmap_dyn_own, hmap_dyn_next, hmap_con;
  std::vector<double> var_lo, var_hi, con_lo, con_hi, params;
  std::string error;

  bool build(const dto_model_vtable* vtab, int horizon, const int32_t* stage_kind, bool want_hessian,
             const double* lo, const double* hi, const double* par, int64_t n_par = -1) {
    vt = vtab;
    T = horizon;
    hessian = want_hessian;
    if (T < 2) return fail("horizon must be >= 2");
    if (want_hessian && !vt->evaluate_hessian) return fail("plugin was generated without Hessians (evaluate_hessian=false)");
    kind.assign(stage_kind, stage_kind + T);
    nx.assign(T, 0); nu.assign(T, 0); nw.assign(T, 0);
    for (int t = 0; t < T; ++t) {
      if (kind[t] < 0 || kind[t] >= vt->n_kind) return fail("stage_kind out of range");
      const dto_kind& k = vt->kinds[kind[t]];
      if ((t < T - 1) != (k.dyn >= 0)) return fail("dynamics must be present exactly for t < T");
      if ((t > 0) != (k.prev_dyn >= 0)) return fail("kind.prev_dyn inconsistent with stage position");
      if (t > 0 && k.prev_dyn != vt->kinds[kind[t - 1]].dyn) return fail("kind.prev_dyn does not match the previous stage");
      if (k.dyn >= 0) {
        nx[t] = vt->dyn[k.dyn].num_state;
        nu[t] = vt->dyn[k.dyn].num_action;
        nw[t] = vt->dyn[k.dyn].num_parameter;
      } else {
        nx[t] = vt->dyn[k.prev_dyn].num_next_state;
        nu[t] = 0;
      }
      if (k.prev_dyn >= 0 && vt->dyn[k.prev_dyn].num_next_state != nx[t]) return fail("state dimension mismatch between stages");
      const dto_cost_class& c = vt->cost[k.cost];
      if (c.num_state != nx[t] || c.num_action != nu[t]) return fail("cost dims do not match stage dims");
      nw[t] = std::max(nw[t], c.num_parameter);
      if (k.con >= 0) {
        const dto_con_class& q = vt->con[k.con];
        if (q.num_state != nx[t] || q.num_action != nu[t]) return fail("constraint dims do not match stage dims");
        nw[t] = std::max(nw[t], q.num_parameter);
      }
      max_nx = std::max(max_nx, nx[t]);
      max_nu = std::max(max_nu, nu[t]);
    }
    // offsets
    zoff.assign(T + 1, 0); woff.assign(T + 1, 0); cdoff.assign(T + 1, 0); ccoff.assign(T + 1, 0);
    jdoff.assign(T + 1, 0); jcoff.assign(T + 1, 0); hoff.assign(T + 1, 0);
    for (int t = 0; t < T; ++t) {
      const dto_kind& k = vt->kinds[kind[t]];
      zoff[t + 1] = zoff[t] + nx[t] + nu[t];
      woff[t + 1] = woff[t] + nw[t];
      cdoff[t + 1] = cdoff[t] + (k.dyn >= 0 ? vt->dyn[k.dyn].num_next_state : 0);
      jdoff[t + 1] = jdoff[t] + (k.dyn >= 0 ? vt->dyn[k.dyn].num_jacobian : 0);
    }
    Nz = zoff[T]; Nw = woff[T]; Ndyn = cdoff[T]; nnzJd = jdoff[T];
    ccoff[0] = (int)Ndyn; jcoff[0] = (int)nnzJd;
    for (int t = 0; t < T; ++t) {
      const dto_kind& k = vt->kinds[kind[t]];
      ccoff[t + 1] = ccoff[t] + (k.con >= 0 ? vt->con[k.con].num_constraint : 0);
      jcoff[t + 1] = jcoff[t] + (k.con >= 0 ? vt->con[k.con].num_jacobian : 0);
    }
    Nstage = ccoff[T] - Ndyn; nnzJs = jcoff[T] - nnzJd;
    const dto_general_class* g = vt->general;
    Ngen = g ? g->num_constraint : 0;
    nnzJg = g ? g->num_jacobian : 0;
    if (g && g->num_variables != Nz) return fail("general constraint num_variables != total variables");
    Nc = Ndyn + Nstage + Ngen;
    nnzJ = nnzJd + nnzJs + nnzJg;

    // Jacobian COO (src/data.jl:170-175)
    jac_rows.clear(); jac_cols.clear();
    jac_rows.reserve(nnzJ); jac_cols.reserve(nnzJ);
    for (int t = 0; t < T - 1; ++t) {
      const dto_dyn_class& d = vt->dyn[vt->kinds[kind[t]].dyn];
      for (int i = 0; i < d.num_jacobian; ++i) {
        jac_rows.push_back(d.jac_rows[i] + cdoff[t]);
        jac_cols.push_back(d.jac_cols[i] + zoff[t]);
      }
    }
    for (int t = 0; t < T; ++t) {
      const int kc = vt->kinds[kind[t]].con;
      if (kc < 0) continue;
      const dto_con_class& q = vt->con[kc];
      for (int i = 0; i < q.num_jacobian; ++i) {
        jac_rows.push_back(q.jac_rows[i] + ccoff[t]);
        jac_cols.push_back(q.jac_cols[i] + zoff[t]);
      }
    }
    if (g) {
      for (int i = 0; i < g->num_jacobian; ++i) {
        jac_rows.push_back(g->jac_rows[i] + Ndyn + Nstage);
        jac_cols.push_back(g->jac_cols[i]);
      }
    }

    // Hessian raw list, key and index maps (src/data.jl:178-187)
    obj_h.assign(T, {}); dyn_h.assign(T, {}); con_h.assign(T, {}); gen_h.clear();
    hess_rows.clear(); hess_cols.clear();
    nnzH_raw = 0; nnzH = 0;
    if (hessian) {
      typedef std::pair<int64_t, int64_t> RC;
      std::vector<RC> raw;
      for (int t = 0; t < T; ++t) {
        const dto_cost_class& c = vt->cost[vt->kinds[kind[t]].cost];
        for (int i = 0; i < c.num_hessian; ++i) raw.emplace_back(c.hess_rows[i] + zoff[t], c.hess_cols[i] + zoff[t]);
      }
      for (int t = 0; t < T - 1; ++t) {
        const dto_dyn_class& d = vt->dyn[vt->kinds[kind[t]].dyn];
        for (int i = 0; i < d.num_hessian; ++i) raw.emplace_back(d.hess_rows[i] + zoff[t], d.hess_cols[i] + zoff[t]);
      }
      for (int t = 0; t < T; ++t) {
        const int kc = vt->kinds[kind[t]].con;
        if (kc < 0) continue;
        const dto_con_class& q = vt->con[kc];
        for (int i = 0; i < q.num_hessian; ++i) raw.emplace_back(q.hess_rows[i] + zoff[t], q.hess_cols[i] + zoff[t]);
      }
      if (g) for (int i = 0; i < g->num_hessian; ++i) raw.emplace_back(g->hess_rows[i], g->hess_cols[i]);
      nnzH_raw = (int64_t)raw.size();
      std::vector<RC> key(raw);
      std::sort(key.begin(), key.end());
      key.erase(std::unique(key.begin(), key.end()), key.end());
      nnzH = (int64_t)key.size();
      hess_rows.resize(nnzH); hess_cols.resize(nnzH);
      for (int64_t i = 0; i < nnzH; ++i) { hess_rows[i] = key[i].first; hess_cols[i] = key[i].second; }
      auto pos = [&](int64_t r, int64_t c) -> int64_t {
        return (int64_t)(std::lower_bound(key.begin(), key.end(), RC(r, c)) - key.begin()) + 1;
      };
      for (int t = 0; t < T; ++t) {
        const dto_kind& k = vt->kinds[kind[t]];
        const dto_cost_class& c = vt->cost[k.cost];
        for (int i = 0; i < c.num_hessian; ++i) obj_h[t].push_back(pos(c.hess_rows[i] + zoff[t], c.hess_cols[i] + zoff[t]));
        if (k.dyn >= 0) {
          const dto_dyn_class& d = vt->dyn[k.dyn];
          for (int i = 0; i < d.num_hessian; ++i) dyn_h[t].push_back(pos(d.hess_rows[i] + zoff[t], d.hess_cols[i] + zoff[t]));
        }
        if (k.con >= 0) {
          const dto_con_class& q = vt->con[k.con];
          for (int i = 0; i < q.num_hessian; ++i) con_h[t].push_back(pos(q.hess_rows[i] + zoff[t], q.hess_cols[i] + zoff[t]));
        }
      }
      if (g) for (int i = 0; i < g->num_hessian; ++i) gen_h.push_back(pos(g->hess_rows[i], g->hess_cols[i]));
      // key slots owned by the rows of stage t
      for (int t = 0; t <= T; ++t) {
        const int64_t first_row = (t < T ? zoff[t] : Nz) + 1;
        hoff[t] = (int)(std::lower_bound(key.begin(), key.end(), RC(first_row, 0)) - key.begin());
      }
      if (!build_hmaps()) return false;
    }

    // bounds (src/data.jl:123-148)
    const double inf = std::numeric_limits<double>::infinity();
    var_lo.assign(Nz, -inf); var_hi.assign(Nz, inf);
    if (lo) var_lo.assign(lo, lo + Nz);
    if (hi) var_hi.assign(hi, hi + Nz);
    con_lo.assign(Nc, 0.0); con_hi.assign(Nc, 0.0);
    for (int t = 0; t < T; ++t) {
      const int kc = vt->kinds[kind[t]].con;
      if (kc < 0) continue;
      const dto_con_class& q = vt->con[kc];
      for (int i = 0; i < q.num_inequality; ++i) con_lo[ccoff[t] + q.indices_inequality[i] - 1] = -inf;
    }
    if (g) for (int i = 0; i < g->num_inequality; ++i) con_lo[Ndyn + Nstage + g->indices_inequality[i] - 1] = -inf;
    params.assign(Nw, 0.0);
    // the caller states how many doubles `par` holds: a short or long vector would silently shift every later stage's w_t
    if (par && n_par != Nw) return fail("dto_problem_spec.num_parameters does not match the model (sum over the stages of the "
                                        "largest num_parameter among the stage's dynamics, cost and constraint)");
    if (par && Nw) params.assign(par, par + Nw);
    return true;
  }

 private:
  bool fail(const char* msg) { error = msg; return false; }

  // Per-kind relative maps; every stage of a kind must produce the same map (checked).
  bool build_hmaps() {
    const int nk = vt->n_kind;
    int stride = 1;
    for (int i = 0; i < vt->n_dyn; ++i) stride = std::max(stride, vt->dyn[i].num_hessian);
    for (int i = 0; i < vt->n_cost; ++i) stride = std::max(stride, vt->cost[i].num_hessian);
    for (int i = 0; i < vt->n_con; ++i) stride = std::max(stride, vt->con[i].num_hessian);
    hmap_stride = stride;
    const int UNSET = -2;
    hmap_cost.assign((size_t)nk * stride, UNSET);
    hmap_dyn_own.assign((size_t)nk * stride, UNSET);
    hmap_dyn_next.assign((size_t)nk * stride, UNSET);
    hmap_con.assign((size_t)nk * stride, UNSET);
    auto put = [&](std::vector<int>& m, int k, int i, int v) -> bool {
      int& slot = m[(size_t)k * stride + i];
      if (slot == UNSET) { slot = v; return true; }
      return slot == v;
    };
    for (int t = 0; t < T; ++t) {
      const int k = kind[t];
      const dto_kind& kd = vt->kinds[k];
      if (hoff[t + 1] - hoff[t] > vt->max_key) return fail("stage owns more Hessian key slots than the plugin was compiled for");
      bool ok = true;
      for (size_t i = 0; i < obj_h[t].size(); ++i) ok &= put(hmap_cost, k, (int)i, (int)(obj_h[t][i] - 1 - hoff[t]));
      for (size_t i = 0; i < con_h[t].size(); ++i) ok &= put(hmap_con, k, (int)i, (int)(con_h[t][i] - 1 - hoff[t]));
      if (kd.dyn >= 0) {
        const dto_dyn_class& d = vt->dyn[kd.dyn];
        const int np = d.num_state + d.num_action;
        for (int i = 0; i < d.num_hessian; ++i) {
          const bool own = d.hess_rows[i] <= np;
          ok &= put(hmap_dyn_own, k, i, own ? (int)(dyn_h[t][i] - 1 - hoff[t]) : -1);
          ok &= put(hmap_dyn_next, kind[t + 1], i, own ? -1 : (int)(dyn_h[t][i] - 1 - hoff[t + 1]));
        }
      }
      if (!ok) return fail("stages of one kind do not share a Hessian scatter map (general-constraint Hessian entries are not supported)");
    }
    for (auto* m : {&hmap_cost, &hmap_dyn_own, &hmap_dyn_next, &hmap_con})
      for (int& v : *m) if (v == UNSET) v = -1;
    if (!gen_h.empty()) return fail("nonlinear general-constraint Hessians are not supported (the reference call is itself broken: src/general_constraint.jl:87)");
    return true;
  }
};

}  // namespace dto
