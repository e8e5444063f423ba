/*
 * ORACLE / CPU baseline (test infrastructure, never linked into the product): serial C port of the
 * interior-point iteration for the BASELINE stage problems (pendulum, acrobot, cartpole, car): equality and
 * inequality stage constraints, variable bounds (finite, one-sided, or equal = fixed).  It restates, for ONE
 * instance on ONE core:
 *   - the reference's stage loops: cost / gradient! / hessian!                    src/costs.jl:49-73
 *                                  constraints! / jacobian! / hessian_lagrangian!  src/dynamics.jl:103-127,
 *                                                                                  src/constraints.jl:80-104
 *   - the KKT system the reference sketches at examples/pendulum/pendulum.jl:138-198
 *         [ H + Sigma + dw I  J' ; J  -D ] [dz; dlam] = -[ grad L_mu ; c ]
 *     solved stage by stage (block-tridiagonal LDL^T),
 *   - the part the reference delegates to Ipopt (src/solver.jl:45-47, src/data.jl:229-255): barrier with monotone
 *     update and mu_target, fraction to the boundary, inertia correction, filter line search (Waechter & Biegler 2006)
 *     with second-order correction and watchdog, Ipopt's scaled termination test incl. the acceptable level, with the
 *     reference Options (src/options.jl:6-36).
 * Variables follow the reference order z = [x_1;u_1;...;x_T] (src/dynamics.jl:188-195); multipliers
 * [dynamics t=1..T-1; stage constraints t=1..T] (src/data.jl:64-75).
 * Used (a) as bench.py's cpu_baseline ("port") and (b) by tests to cross-check the GPU solver's iterates: it mirrors
 * csrc/dto_kkt_kernels.hpp decision for decision, written independently (plain C over sympy-generated model code).
 * PARITY: Ipopt itself cannot run here, so this pins the GPU path against an independent implementation of the same
 * algorithm, not against Ipopt's iterates.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "port_model.h"

/* largest dimensions among the generated models (pendulum 2+1, acrobot / cartpole 4+1, car 3+2): kept tight, the per-stage
 * workspace is what decides whether an instance's data stays in the cache when all cores run */
#define MAXN 4
#define MAXP 6
#define MAXQ PORT_MAXQ
#define MAXBD (MAXP + MAXQ + MAXN)
#define TRI(i, j) ((i) * ((i) + 1) / 2 + (j))
#define LS_NULL_STEP 100.0
#define LS_TRIALS 8
#define FILTER_CAP 24
#define QN_MAX 12

typedef struct {
  /* mirrors dto_options (include/dto.h) + the interior-point constants of csrc/dto_solver.cpp:default_opts */
  double tol, s_max, dual_inf_tol, constr_viol_tol, compl_inf_tol;
  int max_iter;
  double acceptable_tol, acceptable_dual_inf_tol, acceptable_constr_viol_tol, acceptable_compl_inf_tol, acceptable_obj_change_tol;
  int acceptable_iter;
  double diverging_iterates_tol, mu_target;
  double mu_init, kappa_eps, kappa_mu, theta_mu, tau_min, bound_push, bound_frac;
  double delta_c, delta_w_init, delta_w_min, delta_w_max, delta_w_exact_cap, kappa_w_minus, kappa_w_plus, kappa_w_plus_first, piv_tol;
  int max_refactor, watchdog_trigger, watchdog_trials, max_soc;
  int ls_penalty;        /* 1: l1-penalty line search while the iterate is far from the constraint manifold, then the filter */
  int pen_gn;            /* 1: Gauss-Newton Hessian model during the penalty phase (factor_solve) */
  double ls_switch;      /* ... until theta_inf <= ls_switch (dto_options.penalty_switch_theta) */
} port_options;

typedef struct {
  double W[MAXP * (MAXP + 1) / 2], WD[MAXP * (MAXP + 1) / 2], WC[MAXP * (MAXP + 1) / 2], V[MAXP * MAXN], YY[MAXN * (MAXN + 1) / 2];
  double F[MAXN * MAXP], E[MAXN * MAXN], G[MAXQ * MAXP], rp[MAXP], d[MAXN], c[MAXQ] /* residual incl. slack */;
  /* factors */
  double L[MAXBD * MAXBD], dinv[MAXBD], X[MAXBD * MAXN], w[MAXBD];
} stage_t;

typedef struct {
  const port_model* M;
  port_options o;
  int n, m, T, Nz, Nc, Ni, n_bnd;
  int* con;      /* [T] constraint class of the stage or -1 */
  int* ccoff;    /* [T+1] offsets of the stage rows inside lam */
  int* ioff;     /* [T+1] slack offsets */
  double *lo, *hi;
  double *z, *lam, *zl, *zu, *s, *zs, *dz, *dlam, *ds;
  double *soc_c, *soc_buf, *save_dz, *save_dlam, *save_ds;
  stage_t* st;
  int status, iter, nfact, nsoc, filter_n, ls_fail, full_streak, short_streak, watchdog, acc_count, ls_kind;
  double f, f_last, th1, thinf, dinf, compl, e0, logbar, xmax, mu, merit0;
  double szmax, iszmax, sumlam, sumz;
  double delta_lm;
  double piv_min;
  double delta_w, delta_last, gamma, alpha, alpha_pmax, alpha_dmax, gphid, theta_max, theta_min;
  double filt[2 * FILTER_CAP];
  /* limited-memory BFGS mode (the reference's default: Solver(...; evaluate_hessian=false) leaves Ipopt on
   * hessian_approximation=limited-memory, src/solver.jl:7): compact representation B = sigma I - W M^-1 W' (Byrd, Nocedal &
   * Schnabel 1994), the 2k columns of W a dense border of the block-tridiagonal system */
  int qn_mode, qn_m, qn_k, qn_skipped;
  double qn_sigma;
  double *qn_S, *qn_Y, *qn_gl, *qn_s;   /* [m][Nz] pairs, grad_x L(x_k, lam_{k+1}), alpha dz of the last step */
  int ls_mode; double nu_pen, ascale;   /* line-search phase (1 penalty, 2 filter), penalty parameter, trial-step scale of the penalty phase */
} port_solver;

static int np_of(const port_solver* S, int t) { return t < S->T - 1 ? S->n + S->m : S->n; }
static int ny_of(const port_solver* S, int t) { return t < S->T - 1 ? S->n : 0; }
static int q_of(const port_solver* S, int t) { return S->con[t] >= 0 ? S->M->cls[S->con[t]].nc : 0; }
static int zoff(const port_solver* S, int t) { return t * (S->n + S->m); }
static double* lam_dyn(port_solver* S, int t) { return S->lam + t * S->n; }
static double* dlam_dyn(port_solver* S, int t) { return S->dlam + t * S->n; }
static int finite_lo(double v) { return v > -1e300; }
static int finite_hi(double v) { return v < 1e300; }

void port_default_options(port_options* o) {
  o->tol = 1e-6; o->s_max = 100.0; o->dual_inf_tol = 1.0; o->constr_viol_tol = 1e-3; o->compl_inf_tol = 1e-3; o->max_iter = 1000;
  o->acceptable_tol = 1e-6; o->acceptable_iter = 15; o->acceptable_dual_inf_tol = 1e10; o->acceptable_constr_viol_tol = 1e-2;
  o->acceptable_compl_inf_tol = 1e-2; o->acceptable_obj_change_tol = 1e-5; o->diverging_iterates_tol = 1e8; o->mu_target = 1e-4;
  o->mu_init = 0.1; o->kappa_eps = 10.0; o->kappa_mu = 0.2; o->theta_mu = 1.5; o->tau_min = 0.99; o->bound_push = 1e-2; o->bound_frac = 1e-2;
  o->delta_c = 1e-8; o->delta_w_init = 1e-4; o->delta_w_min = 1e-20; o->delta_w_max = 1e20; o->delta_w_exact_cap = 100.0;
  o->kappa_w_minus = 1.0 / 3.0; o->kappa_w_plus = 8.0; o->kappa_w_plus_first = 100.0; o->piv_tol = 1e-9;
  o->max_refactor = 9; o->watchdog_trigger = 2; o->watchdog_trials = 4; o->max_soc = 0;
  if (getenv("DTO_WATCHDOG")) sscanf(getenv("DTO_WATCHDOG"), "%d,%d", &o->watchdog_trigger, &o->watchdog_trials);
  if (getenv("DTO_EXACT_CAP")) o->delta_w_exact_cap = atof(getenv("DTO_EXACT_CAP"));
  if (getenv("DTO_MAX_SOC")) o->max_soc = atoi(getenv("DTO_MAX_SOC"));
  o->ls_penalty = 1; o->ls_switch = 1.0; o->pen_gn = 1;
  if (getenv("DTO_PEN_GN")) o->pen_gn = atoi(getenv("DTO_PEN_GN"));
  if (getenv("DTO_LS_MERIT")) o->ls_penalty = atoi(getenv("DTO_LS_MERIT"));          /* experiment knobs */
  if (getenv("DTO_KW_MINUS")) o->kappa_w_minus = atof(getenv("DTO_KW_MINUS"));
  if (getenv("DTO_KW_PLUS")) o->kappa_w_plus = atof(getenv("DTO_KW_PLUS"));
  if (getenv("DTO_KW_PLUS_FIRST")) o->kappa_w_plus_first = atof(getenv("DTO_KW_PLUS_FIRST"));
  if (getenv("DTO_LS_SWITCH_INF")) o->ls_switch = atof(getenv("DTO_LS_SWITCH_INF"));
}

void port_set_int(port_solver* S, const char* name, int v);
port_solver* port_create(const char* model, int T, const int* con, const double* lo, const double* hi, int max_iter) {
  const port_model* M = NULL;
  for (int i = 0; i < PORT_NMODELS; ++i)
    if (!strcmp(PORT_MODELS[i].name, model)) M = &PORT_MODELS[i];
  if (!M || T < 2) return NULL;
  port_solver* S = (port_solver*)calloc(1, sizeof(port_solver));
  S->M = M; S->n = M->n; S->m = M->m; S->T = T;
  port_default_options(&S->o);
  S->o.max_iter = max_iter;
  S->Nz = T * (M->n + M->m) - M->m;
  S->con = (int*)calloc(T, sizeof(int));
  S->ccoff = (int*)calloc(T + 1, sizeof(int));
  S->ioff = (int*)calloc(T + 1, sizeof(int));
  S->ccoff[0] = (T - 1) * M->n;
  for (int t = 0; t < T; ++t) {
    S->con[t] = con ? con[t] : -1;
    if (S->con[t] >= M->n_class) { free(S->con); free(S->ccoff); free(S->ioff); free(S); return NULL; }
    int q = 0, qi = 0;
    if (S->con[t] >= 0) {
      q = M->cls[S->con[t]].nc;
      for (int j = 0; j < q; ++j) qi += M->cls[S->con[t]].ineq[j];
    }
    S->ccoff[t + 1] = S->ccoff[t] + q;
    S->ioff[t + 1] = S->ioff[t] + qi;
  }
  S->Nc = S->ccoff[T]; S->Ni = S->ioff[T];
  S->lo = (double*)malloc(S->Nz * sizeof(double)); S->hi = (double*)malloc(S->Nz * sizeof(double));
  S->n_bnd = S->Ni;
  for (int i = 0; i < S->Nz; ++i) {
    S->lo[i] = lo ? lo[i] : -INFINITY; S->hi[i] = hi ? hi[i] : INFINITY;
    if (S->lo[i] != S->hi[i]) { if (finite_lo(S->lo[i])) S->n_bnd++; if (finite_hi(S->hi[i])) S->n_bnd++; }
  }
#define ALLOC(p, n) S->p = (double*)calloc((n) > 0 ? (n) : 1, sizeof(double))
  ALLOC(z, S->Nz); ALLOC(dz, S->Nz); ALLOC(zl, S->Nz); ALLOC(zu, S->Nz); ALLOC(save_dz, S->Nz);
  ALLOC(lam, S->Nc); ALLOC(dlam, S->Nc); ALLOC(soc_c, S->Nc); ALLOC(soc_buf, S->Nc); ALLOC(save_dlam, S->Nc);
  ALLOC(s, S->Ni); ALLOC(zs, S->Ni); ALLOC(ds, S->Ni); ALLOC(save_ds, S->Ni);
#undef ALLOC
  S->st = (stage_t*)calloc(T, sizeof(stage_t));
  if (getenv("DTO_LBFGS")) port_set_int(S, "lbfgs", atoi(getenv("DTO_LBFGS")));   /* experiment knob */
  return S;
}

void port_destroy(port_solver* S) {
  if (!S) return;
  free(S->con); free(S->ccoff); free(S->ioff); free(S->lo); free(S->hi);
  free(S->z); free(S->dz); free(S->zl); free(S->zu); free(S->save_dz); free(S->lam); free(S->dlam); free(S->soc_c);
  free(S->soc_buf); free(S->save_dlam); free(S->s); free(S->zs); free(S->ds); free(S->save_ds); free(S->st);
  free(S->qn_S); free(S->qn_Y); free(S->qn_gl); free(S->qn_s); free(S);
}

void port_set_int(port_solver* S, const char* name, int v) {
  if (!strcmp(name, "max_soc")) S->o.max_soc = v;
  else if (!strcmp(name, "max_iter")) S->o.max_iter = v;
  else if (!strcmp(name, "watchdog_trigger")) S->o.watchdog_trigger = v;
  else if (!strcmp(name, "watchdog_trials")) S->o.watchdog_trials = v;
  else if (!strcmp(name, "acceptable_iter")) S->o.acceptable_iter = v;
  else if (!strcmp(name, "ls_penalty")) S->o.ls_penalty = v;
  else if (!strcmp(name, "lbfgs")) {   /* v = history length (Ipopt: limited_memory_max_history = 6); 0 = exact Hessians */
    S->qn_mode = v > 0 ? 2 : 0; S->qn_m = v > QN_MAX ? QN_MAX : v;
    if (v > 0 && !S->qn_S) {
      S->qn_S = (double*)calloc((size_t)QN_MAX * S->Nz, sizeof(double)); S->qn_Y = (double*)calloc((size_t)QN_MAX * S->Nz, sizeof(double));
      S->qn_gl = (double*)calloc(S->Nz, sizeof(double)); S->qn_s = (double*)calloc(S->Nz, sizeof(double));
    }
  }
  else if (!strcmp(name, "pen_gn")) S->o.pen_gn = v;
}
void port_set_double(port_solver* S, const char* name, double v) {
  if (!strcmp(name, "mu_target")) S->o.mu_target = v;
  else if (!strcmp(name, "tol")) S->o.tol = v;
  else if (!strcmp(name, "delta_w_exact_cap")) S->o.delta_w_exact_cap = v;
  else if (!strcmp(name, "ls_switch")) S->o.ls_switch = v;
  else if (!strcmp(name, "kappa_w_minus")) S->o.kappa_w_minus = v;
  else if (!strcmp(name, "kappa_w_plus")) S->o.kappa_w_plus = v;
  else if (!strcmp(name, "kappa_w_plus_first")) S->o.kappa_w_plus_first = v;
  else if (!strcmp(name, "delta_w_init")) S->o.delta_w_init = v;
  else if (!strcmp(name, "delta_w_min")) S->o.delta_w_min = v;
}

/* slack index of row j of stage t (-1: equality row) */
static int slack_of(const port_solver* S, int t, int j) {
  if (S->con[t] < 0 || !S->M->cls[S->con[t]].ineq[j]) return -1;
  int k = 0;
  for (int i = 0; i < j; ++i) k += S->M->cls[S->con[t]].ineq[i];
  return S->ioff[t] + k;
}

/* ---- initialisation (k_init): push the guess into the bounds, slacks from the inequality values, multipliers on the
 *      central path of mu_init (Waechter & Biegler 2006, section 3.6) ---- */
void port_begin(port_solver* S, const double* z0) {
  const port_options* o = &S->o;
  memcpy(S->z, z0, S->Nz * sizeof(double));
  memset(S->lam, 0, S->Nc * sizeof(double));
  for (int i = 0; i < S->Nz; ++i) {
    double v = S->z[i];
    const double lo = S->lo[i], hi = S->hi[i];
    double zl = 0, zu = 0;
    if (lo == hi) v = lo;
    else {
      const int fl = finite_lo(lo), fh = finite_hi(hi);
      if (fl && fh) {
        const double pl = fmin(o->bound_push * fmax(1.0, fabs(lo)), o->bound_frac * (hi - lo));
        const double pu = fmin(o->bound_push * fmax(1.0, fabs(hi)), o->bound_frac * (hi - lo));
        v = fmin(fmax(v, lo + pl), hi - pu);
      } else if (fl) v = fmax(v, lo + o->bound_push * fmax(1.0, fabs(lo)));
      else if (fh) v = fmin(v, hi - o->bound_push * fmax(1.0, fabs(hi)));
      if (fl) zl = o->mu_init / (v - lo);
      if (fh) zu = o->mu_init / (hi - v);
    }
    S->z[i] = v; S->zl[i] = zl; S->zu[i] = zu;
  }
  for (int t = 0; t < S->T; ++t) {
    if (S->con[t] < 0) continue;
    const port_con_class* C = &S->M->cls[S->con[t]];
    double c[MAXQ];
    const double* x = S->z + zoff(S, t);
    C->val(x, x + S->n, c);
    for (int j = 0; j < C->nc; ++j) {
      const int k = slack_of(S, t, j);
      if (k < 0) continue;
      const double sv = fmax(-c[j], o->bound_push * fmax(1.0, fabs(c[j])));
      S->s[k] = sv; S->zs[k] = o->mu_init / sv;
      S->lam[S->ccoff[t] + j] = S->zs[k];
    }
  }
  S->status = 0; S->iter = 0; S->nfact = 0; S->nsoc = 0; S->filter_n = 0; S->ls_fail = 0; S->full_streak = 0; S->short_streak = 0;
  S->watchdog = 0; S->acc_count = 0; S->ls_kind = 0;
  S->mu = o->mu_init; S->f_last = 1e300;
  S->delta_lm = 0; S->delta_w = 0; S->delta_last = 0; S->gamma = 1.0; S->alpha = 0; S->theta_max = -1; S->theta_min = -1;
  S->ls_mode = o->ls_penalty ? 1 : 2; S->nu_pen = 0.0; S->ascale = 1.0;
  S->qn_k = 0; S->qn_sigma = 1.0; S->qn_skipped = 0;
}

/* ---- derivative blocks of every stage + residual norms (k_stage_eval + k_conv on the GPU) ---- */
static void eval_all(port_solver* S) {
  const port_model* M = S->M;
  const int n = S->n, T = S->T;
  double f = 0, th1 = 0, thinf = 0, dinf = 0, szmax = 0, iszmax = 0, sumlam = 0, sumz = 0, logbar = 0, xmax = 0;
  for (int t = 0; t < T; ++t) {
    stage_t* s = &S->st[t];
    const double* x = S->z + zoff(S, t);
    const int np = np_of(S, t);
    double l;
    memset(s->WD, 0, sizeof(s->WD)); memset(s->WC, 0, sizeof(s->WC)); memset(s->V, 0, sizeof(s->V)); memset(s->YY, 0, sizeof(s->YY));
    if (t < T - 1) {
      const double* u = x + n;
      const double* y = S->z + zoff(S, t + 1);
      M->cost(x, u, &l, s->rp, s->W);
      M->dyn(x, u, y, lam_dyn(S, t), s->d, s->F, s->E, s->WD, s->V, s->YY);
      for (int i = 0; i < n; ++i)
        for (int j = 0; j < np; ++j) s->rp[j] += s->F[i * np + j] * lam_dyn(S, t)[i];
      for (int i = 0; i < n; ++i) { th1 += fabs(s->d[i]); thinf = fmax(thinf, fabs(s->d[i])); sumlam += fabs(lam_dyn(S, t)[i]); }
    } else {
      M->costT(x, &l, s->rp, s->W);
    }
    f += l;
    if (S->con[t] >= 0) {
      const port_con_class* C = &M->cls[S->con[t]];
      const double* nu = S->lam + S->ccoff[t];
      double c[MAXQ];
      C->f(x, x + n, nu, c, s->G, s->WC);
      for (int j = 0; j < C->nc; ++j) {
        for (int i = 0; i < C->np; ++i) s->rp[i] += s->G[j * C->np + i] * nu[j];
        double r = c[j];
        const int k = slack_of(S, t, j);
        if (k >= 0) {
          const double sv = S->s[k], zv = S->zs[k];
          r = c[j] + sv;
          dinf = fmax(dinf, fabs(nu[j] - zv));
          szmax = fmax(szmax, sv * zv); iszmax = fmax(iszmax, 1.0 / (sv * zv));
          sumz += fabs(zv); logbar += log(sv);
        }
        s->c[j] = r;
        th1 += fabs(r); thinf = fmax(thinf, fabs(r)); sumlam += fabs(nu[j]);
      }
    }
    if (t > 0) { /* E_{t-1}' lam_{t-1} lands in the x rows of stage t */
      const stage_t* sp = &S->st[t - 1];
      for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) s->rp[j] += sp->E[i * n + j] * lam_dyn(S, t - 1)[i];
    }
    for (int i = 0; i < np; ++i) {
      const int zi = zoff(S, t) + i;
      const double lo = S->lo[zi], hi = S->hi[zi], p = S->z[zi];
      xmax = fmax(xmax, fabs(p));
      if (lo == hi) continue;
      dinf = fmax(dinf, fabs(s->rp[i] - S->zl[zi] + S->zu[zi]));
      if (finite_lo(lo)) { const double v = (p - lo) * S->zl[zi]; szmax = fmax(szmax, v); iszmax = fmax(iszmax, 1.0 / v); sumz += fabs(S->zl[zi]); logbar += log(p - lo); }
      if (finite_hi(hi)) { const double v = (hi - p) * S->zu[zi]; szmax = fmax(szmax, v); iszmax = fmax(iszmax, 1.0 / v); sumz += fabs(S->zu[zi]); logbar += log(hi - p); }
    }
  }
  S->f = f; S->th1 = th1; S->thinf = thinf; S->dinf = dinf; S->logbar = logbar; S->xmax = xmax;
  S->szmax = szmax; S->iszmax = iszmax; S->sumlam = sumlam; S->sumz = sumz;
}

static double compl_at(const port_solver* S, double m) {
  const double szmin = S->iszmax > 0.0 ? 1.0 / S->iszmax : 1e300;
  return S->n_bnd > 0 ? fmax(S->szmax - m, m - szmin) : 0.0;
}

/* k_conv: Ipopt's scaled termination test, acceptable level, monotone barrier update with mu_target */
static void convergence(port_solver* S) {
  const port_options* o = &S->o;
  const int n_mult = S->Nc, n_bnd = S->n_bnd;
  const double sd = fmax(o->s_max, (S->sumlam + S->sumz) / (double)(n_mult + n_bnd > 0 ? n_mult + n_bnd : 1)) / o->s_max;
  const double scn = fmax(o->s_max, S->sumz / (double)(n_bnd > 0 ? n_bnd : 1)) / o->s_max;
  const double c0 = compl_at(S, o->mu_target);
  const double e0 = fmax(fmax(S->dinf / sd, S->thinf), c0 / scn);
  S->compl = c0; S->e0 = e0;
  const double f = S->f;
  const int nonfinite = !(f == f) || !(S->th1 == S->th1) || !(S->dinf == S->dinf) || fabs(f) > 1e300 || S->th1 > 1e300;
  const int acceptable = o->acceptable_iter > 0 && e0 <= o->acceptable_tol && S->dinf <= o->acceptable_dual_inf_tol &&
                         S->thinf <= o->acceptable_constr_viol_tol && c0 <= o->acceptable_compl_inf_tol &&
                         fabs(f - S->f_last) / fmax(1.0, fabs(f)) <= o->acceptable_obj_change_tol;
  S->acc_count = acceptable ? S->acc_count + 1 : 0;
  S->f_last = f;
  if (nonfinite) S->status = 3;
  else if (e0 <= o->tol && S->dinf <= o->dual_inf_tol && S->thinf <= o->constr_viol_tol && c0 <= o->compl_inf_tol) S->status = 1;
  else if (o->acceptable_iter > 0 && S->acc_count >= o->acceptable_iter) S->status = 4;
  else if (S->xmax > o->diverging_iterates_tol) S->status = 5;
  else if (S->iter >= o->max_iter) S->status = 2;
  else if (n_bnd > 0) {
    const double mu_floor = fmax(o->mu_target, fmin(o->tol, o->compl_inf_tol) / (o->kappa_eps + 1.0));
    int changed = 0;
    double mu = S->mu;
    for (int k = 0; k < 8; ++k) {
      const double emu = fmax(fmax(S->dinf / sd, S->thinf), compl_at(S, mu) / scn);
      if (!(emu <= o->kappa_eps * mu) || mu <= mu_floor) break;
      mu = fmax(mu_floor, fmin(o->kappa_mu * mu, pow(mu, o->theta_mu)));
      changed = 1;
    }
    if (changed) { S->mu = mu; S->filter_n = 0; }
  }
  if (S->theta_max < 0) { S->theta_max = 1e4 * fmax(1.0, S->th1); S->theta_min = 1e-4 * fmax(1.0, S->th1); }
  S->merit0 = f - S->mu * S->logbar;
  /* end of the penalty phase (line_search): decided here, before the factorisation, which uses the Gauss-Newton model while it lasts */
  if (S->ls_mode == 1 && S->thinf <= o->ls_switch) { S->ls_mode = 2; S->filter_n = 0; }
}

/* ---- block-tridiagonal LDL^T, forward sweep; returns 1 if the inertia is (Nz, Nc, 0) and no pivot is tiny.
 *      rhs_c: NULL = the residuals of the current iterate, else replacement constraint residuals (second-order
 *      correction: same matrix, other right-hand side) ---- */
static int forward_sweep(port_solver* S, double dw, double gam, const double* rhs_c) {
  const port_options* o = &S->o;
  const int n = S->n, T = S->T;
  const double mu = S->mu;
  double Pm[MAXN * MAXN], py[MAXN];
  memset(Pm, 0, sizeof(Pm)); memset(py, 0, sizeof(py));
  int ok = 1, nneg = 0;
  S->piv_min = 0.0;
  for (int t = 0; t < T; ++t) {
    stage_t* s = &S->st[t];
    const int np = np_of(S, t), q = q_of(S, t), ny = ny_of(S, t), bd = np + q + ny, z0 = zoff(S, t);
    double A[MAXBD][MAXBD], y[MAXBD], X[MAXBD][MAXN];
    int fixed[MAXP];
    int stage_ok = 1, stage_neg = 0;
    {
      memset(A, 0, sizeof(A));
      memset(X, 0, sizeof(X));
      if (S->qn_mode == 2) {   /* limited-memory mode: K0 carries sigma I, the low-rank part is a border (qn_factor_solve) */
        for (int i = 0; i < np; ++i) A[i][i] = S->qn_sigma;
      } else {
      for (int i = 0; i < np; ++i)
        for (int j = 0; j <= i; ++j) A[i][j] = s->W[TRI(i, j)] + gam * (s->WD[TRI(i, j)] + s->WC[TRI(i, j)]);
      }
      for (int i = 0; i < n; ++i)
        for (int j = 0; j <= i; ++j) A[i][j] += Pm[i * n + j];
      for (int i = 0; i < np; ++i) {
        double rp = s->rp[i], sig = dw;
        const double lo = S->lo[z0 + i], hi = S->hi[z0 + i], p = S->z[z0 + i];
        fixed[i] = (lo == hi);
        if (!fixed[i]) {
          if (finite_lo(lo)) { sig += S->zl[z0 + i] / (p - lo); rp -= mu / (p - lo); }
          if (finite_hi(hi)) { sig += S->zu[z0 + i] / (hi - p); rp += mu / (hi - p); }
        }
        A[i][i] += sig;
        y[i] = -rp - (i < n ? py[i] : 0.0);
      }
      for (int j = 0; j < q; ++j) {
        const port_con_class* C = &S->M->cls[S->con[t]];
        for (int i = 0; i < C->np; ++i) A[np + j][i] = s->G[j * C->np + i];
        double dc = o->delta_c;
        double r = rhs_c ? rhs_c[S->ccoff[t] + j] : s->c[j];
        const int k = slack_of(S, t, j);
        if (k >= 0) {
          const double sv = S->s[k], zv = S->zs[k], nu = S->lam[S->ccoff[t] + j];
          dc += sv / zv;
          r -= (sv / zv) * (nu - mu / sv);
        }
        A[np + j][np + j] = -dc;
        y[np + j] = -r;
      }
      for (int k = 0; k < ny; ++k) {
        for (int i = 0; i < np; ++i) A[np + q + k][i] = s->F[k * np + i];
        A[np + q + k][np + q + k] = -o->delta_c;
        y[np + q + k] = -(rhs_c ? rhs_c[t * n + k] : s->d[k]);
      }
      for (int i = 0; i < np; ++i) for (int c = 0; c < ny; ++c) X[i][c] = gam * s->V[i * ny + c];
      for (int k = 0; k < ny; ++k) for (int c = 0; c < ny; ++c) X[np + q + k][c] = s->E[k * ny + c];
      for (int i = 0; i < np; ++i) {
        if (!fixed[i]) continue;
        for (int r2 = 0; r2 < bd; ++r2) { if (r2 > i) A[r2][i] = 0.0; if (r2 < i) A[i][r2] = 0.0; }
        A[i][i] = 1.0; y[i] = 0.0;
        for (int c = 0; c < ny; ++c) X[i][c] = 0.0;
      }
      /* right-looking LDL^T, static order */
      for (int j = 0; j < bd; ++j) {
        double dj = A[j][j], cmax = 0;
        for (int i = j + 1; i < bd; ++i) cmax = fmax(cmax, fabs(A[i][j]));
        if (!(fabs(dj) > o->piv_tol * fmax(1.0, cmax))) { stage_ok = 0; dj = (dj < 0 ? -1.0 : 1.0) * fmax(fabs(dj), o->piv_tol); }
        if (dj < 0) ++stage_neg;
        if (j < np && dj < 0 && (S->piv_min == 0.0 || (getenv("DTO_PIV_MIN") && dj < S->piv_min))) S->piv_min = dj;   /* (experiment DTO_PIV_JUMP: first [most] negative primal pivot of the sweep) */
        const double inv = 1.0 / dj;
        s->dinv[j] = inv;
        for (int i = j + 1; i < bd; ++i) {
          const double lij = A[i][j] * inv;
          for (int k = j + 1; k <= i; ++k) A[i][k] -= lij * A[k][j];
        }
        for (int i = j + 1; i < bd; ++i) A[i][j] *= inv;
      }
    }
    if (!stage_ok) ok = 0;
    nneg += stage_neg;
    if (stage_neg != q + ny) ok = 0; /* the block's primal pivots come first: exactly its q + ny constraint pivots are negative */
    for (int i = 1; i < bd; ++i)
      for (int k = 0; k < i; ++k) {
        const double l = A[i][k];
        for (int c = 0; c < ny; ++c) X[i][c] -= l * X[k][c];
        y[i] -= l * y[k];
      }
    for (int c = 0; c < ny; ++c) {
      for (int e = 0; e <= c; ++e) {
        double acc = gam * s->YY[TRI(c, e)];
        for (int i = 0; i < bd; ++i) acc -= X[i][c] * X[i][e] * s->dinv[i];
        Pm[c * n + e] = acc;
      }
      double acc = 0;
      for (int i = 0; i < bd; ++i) acc += X[i][c] * s->dinv[i] * y[i];
      py[c] = acc;
    }
    for (int i = 0; i < bd; ++i) {
      for (int k = 0; k < i; ++k) s->L[i * MAXBD + k] = A[i][k];
      s->w[i] = y[i];
      for (int c = 0; c < ny; ++c) s->X[i * MAXN + c] = X[i][c];
    }
  }
  if (nneg != S->Nc) ok = 0;
  return ok;
}

/* k_kkt_bwd: back substitution, step of the eliminated slack / bound multipliers, fraction to the boundary, directional
 * derivative of the barrier objective.  res: the constraint residuals of the right-hand side that was solved (the
 * iterate's own, or the second-order-correction ones) */
static void backward_sweep(port_solver* S) {
  const port_options* o = &S->o;
  const int n = S->n, T = S->T;
  const double mu = S->mu, tau = fmax(o->tau_min, 1.0 - mu);
  double xn[MAXN];
  memset(xn, 0, sizeof(xn));
  double gphid = 0, apmax = 1.0, admax = 1.0;
  for (int t = T - 1; t >= 0; --t) {
    stage_t* s = &S->st[t];
    const int np = np_of(S, t), q = q_of(S, t), ny = ny_of(S, t), bd = np + q + ny, z0 = zoff(S, t);
    double v[MAXBD];
    for (int i = 0; i < bd; ++i) {
      double r = s->w[i];
      for (int c = 0; c < ny; ++c) r -= s->X[i * MAXN + c] * xn[c];
      v[i] = r * s->dinv[i];
    }
    for (int i = bd - 1; i >= 1; --i)
      for (int k = 0; k < i; ++k) v[k] -= s->L[i * MAXBD + k] * v[i];
    for (int i = 0; i < np; ++i) {
      const double dp = v[i];
      S->dz[z0 + i] = dp;
      gphid += s->rp[i] * dp;
      const double lo = S->lo[z0 + i], hi = S->hi[z0 + i], p = S->z[z0 + i];
      if (lo == hi) continue;
      if (finite_lo(lo)) {
        const double zl = S->zl[z0 + i], gap = p - lo, dzl = mu / gap - zl - (zl / gap) * dp;
        if (dp < 0.0) apmax = fmin(apmax, -tau * gap / dp);
        if (dzl < 0.0) admax = fmin(admax, -tau * zl / dzl);
        gphid -= mu / gap * dp;
      }
      if (finite_hi(hi)) {
        const double zu = S->zu[z0 + i], gap = hi - p, dzu = mu / gap - zu + (zu / gap) * dp;
        if (dp > 0.0) apmax = fmin(apmax, tau * gap / dp);
        if (dzu < 0.0) admax = fmin(admax, -tau * zu / dzu);
        gphid += mu / gap * dp;
      }
    }
    for (int j = 0; j < q; ++j) {
      const double dnu = v[np + j], nu = S->lam[S->ccoff[t] + j], r = s->c[j];
      S->dlam[S->ccoff[t] + j] = dnu;
      double dsv = 0.0;
      const int k = slack_of(S, t, j);
      if (k >= 0) {
        const double sv = S->s[k], zv = S->zs[k];
        dsv = -(sv / zv) * (nu - mu / sv + dnu);
        const double dzs = mu / sv - zv - (zv / sv) * dsv;
        S->ds[k] = dsv;
        if (dsv < 0.0) apmax = fmin(apmax, -tau * sv / dsv);
        if (dzs < 0.0) admax = fmin(admax, -tau * zv / dzs);
        gphid -= mu / sv * dsv;
      }
      gphid += nu * (r - o->delta_c * dnu + dsv);
    }
    for (int k = 0; k < ny; ++k) {
      dlam_dyn(S, t)[k] = v[np + q + k];
      gphid += lam_dyn(S, t)[k] * (s->d[k] - o->delta_c * v[np + q + k]);
    }
    for (int i = 0; i < n; ++i) xn[i] = v[i];
  }
  S->gphid = gphid; S->alpha_pmax = apmax; S->alpha_dmax = admax;
}

/* ---- limited-memory BFGS ---------------------------------------------------------------------------------------------- */
/* K0 v = (rx; rc) with the factors forward_sweep stored (same recursions as forward_sweep / backward_sweep) */
static void solve_stored(port_solver* S, const double* rx, const double* rc, double* vx, double* vc) {
  const int n = S->n, T = S->T;
  double py[MAXN], xn[MAXN];
  double* ys = (double*)malloc((size_t)T * MAXBD * sizeof(double));
  memset(py, 0, sizeof(py));
  for (int t = 0; t < T; ++t) {
    const stage_t* s = &S->st[t];
    const int np = np_of(S, t), q = q_of(S, t), ny = ny_of(S, t), bd = np + q + ny, z0 = zoff(S, t);
    double* y = ys + (size_t)t * MAXBD;
    for (int i = 0; i < np; ++i) y[i] = (S->lo[z0 + i] == S->hi[z0 + i]) ? 0.0 : rx[z0 + i] - (i < n ? py[i] : 0.0);
    for (int j = 0; j < q; ++j) y[np + j] = rc[S->ccoff[t] + j];
    for (int k = 0; k < ny; ++k) y[np + q + k] = rc[t * n + k];
    for (int i = 1; i < bd; ++i)
      for (int k = 0; k < i; ++k) y[i] -= s->L[i * MAXBD + k] * y[k];
    for (int c = 0; c < ny; ++c) {
      double acc = 0;
      for (int i = 0; i < bd; ++i) acc += s->X[i * MAXN + c] * s->dinv[i] * y[i];
      py[c] = acc;
    }
  }
  memset(xn, 0, sizeof(xn));
  for (int t = T - 1; t >= 0; --t) {
    const stage_t* s = &S->st[t];
    const int np = np_of(S, t), q = q_of(S, t), ny = ny_of(S, t), bd = np + q + ny, z0 = zoff(S, t);
    const double* y = ys + (size_t)t * MAXBD;
    double v[MAXBD];
    for (int i = 0; i < bd; ++i) {
      double r = y[i];
      for (int c = 0; c < ny; ++c) r -= s->X[i * MAXN + c] * xn[c];
      v[i] = r * s->dinv[i];
    }
    for (int i = bd - 1; i >= 1; --i)
      for (int k = 0; k < i; ++k) v[k] -= s->L[i * MAXBD + k] * v[i];
    for (int i = 0; i < np; ++i) vx[z0 + i] = v[i];
    for (int j = 0; j < q; ++j) vc[S->ccoff[t] + j] = v[np + j];
    for (int k = 0; k < ny; ++k) vc[t * n + k] = v[np + q + k];
    for (int i = 0; i < n; ++i) xn[i] = v[i];
  }
  free(ys);
}

/* eigenvalue signs of a small symmetric matrix (cyclic Jacobi): returns the number of negative eigenvalues, *tiny = 1 if one is
 * numerically zero */
static int small_inertia(const double* A, int m, int* tiny) {
  double B[2 * QN_MAX][2 * QN_MAX];
  double scale = 0;
  for (int i = 0; i < m; ++i) for (int j = 0; j < m; ++j) { B[i][j] = A[i * m + j]; scale = fmax(scale, fabs(B[i][j])); }
  for (int sweep = 0; sweep < 60; ++sweep) {
    double off = 0;
    for (int i = 0; i < m; ++i) for (int j = 0; j < i; ++j) off += B[i][j] * B[i][j];
    if (off <= 1e-30 * scale * scale) break;
    for (int p = 0; p < m; ++p)
      for (int q = p + 1; q < m; ++q) {
        if (fabs(B[p][q]) < 1e-300) continue;
        const double th = (B[q][q] - B[p][p]) / (2.0 * B[p][q]);
        const double t = (th >= 0 ? 1.0 : -1.0) / (fabs(th) + sqrt(th * th + 1.0)), c = 1.0 / sqrt(t * t + 1.0), sn = t * c;
        for (int k = 0; k < m; ++k) { const double a = B[k][p], b = B[k][q]; B[k][p] = c * a - sn * b; B[k][q] = sn * a + c * b; }
        for (int k = 0; k < m; ++k) { const double a = B[p][k], b = B[q][k]; B[p][k] = c * a - sn * b; B[q][k] = sn * a + c * b; }
      }
  }
  int neg = 0; *tiny = 0;
  for (int i = 0; i < m; ++i) { if (B[i][i] < 0) ++neg; if (fabs(B[i][i]) <= 1e-13 * scale) *tiny = 1; }
  return neg;
}

/* dense solve with partial pivoting, A (m x m, destroyed) x = b (in place) */
static void small_solve(double* A, double* b, int m) {
  for (int j = 0; j < m; ++j) {
    int p = j;
    for (int i = j + 1; i < m; ++i) if (fabs(A[i * m + j]) > fabs(A[p * m + j])) p = i;
    if (p != j) { for (int k = 0; k < m; ++k) { const double t = A[j * m + k]; A[j * m + k] = A[p * m + k]; A[p * m + k] = t; } const double t = b[j]; b[j] = b[p]; b[p] = t; }
    const double inv = 1.0 / A[j * m + j];
    for (int i = j + 1; i < m; ++i) {
      const double l = A[i * m + j] * inv;
      if (l == 0.0) continue;
      for (int k = j; k < m; ++k) A[i * m + k] -= l * A[j * m + k];
      b[i] -= l * b[j];
    }
  }
  for (int j = m - 1; j >= 0; --j) { double acc = b[j]; for (int k = j + 1; k < m; ++k) acc -= A[j * m + k] * b[k]; b[j] = acc / A[j * m + j]; }
}

/* after the step: grad_x L(x_k, lam_{k+1}) = grad_x L(x_k, lam_k) + alpha J(x_k)' dlam, from the stage blocks of x_k (still in
 * S->st), and the primal step -- the next iteration forms y_k = grad_x L(x_{k+1}, lam_{k+1}) - this (Ipopt's limited-memory
 * update uses the Lagrangian gradients at the new multipliers) */
static void qn_save(port_solver* S) {
  const int n = S->n, T = S->T;
  const double al = S->alpha;
  for (int t = 0; t < T; ++t) {
    const stage_t* s = &S->st[t];
    const int np = np_of(S, t), q = q_of(S, t), z0 = zoff(S, t);
    for (int i = 0; i < np; ++i) {
      double g = s->rp[i];
      if (t < T - 1) for (int k = 0; k < n; ++k) g += al * s->F[k * np + i] * dlam_dyn(S, t)[k];
      if (S->con[t] >= 0) { const port_con_class* C = &S->M->cls[S->con[t]]; if (i < C->np) for (int j = 0; j < q; ++j) g += al * s->G[j * C->np + i] * S->dlam[S->ccoff[t] + j]; }
      if (t > 0 && i < n) { const stage_t* sp = &S->st[t - 1]; for (int k = 0; k < n; ++k) g += al * sp->E[k * n + i] * dlam_dyn(S, t - 1)[k]; }
      S->qn_gl[z0 + i] = g;
      S->qn_s[z0 + i] = al * S->dz[z0 + i];
    }
  }
}

/* at the new point (after eval_all): push the pair (s, y) unless the curvature condition fails (Ipopt skips the update then) */
static void qn_update(port_solver* S) {
  const int Nz = S->Nz;
  if (S->iter == 0 || S->alpha <= 0.0) return;
  double* y = (double*)malloc(Nz * sizeof(double));
  double sy = 0, ss = 0, yy = 0;
  for (int t = 0; t < S->T; ++t) {
    const int np = np_of(S, t), z0 = zoff(S, t);
    for (int i = 0; i < np; ++i) {
      const int zi = z0 + i;
      const int fx = S->lo[zi] == S->hi[zi];
      y[zi] = fx ? 0.0 : S->st[t].rp[i] - S->qn_gl[zi];
      if (fx) S->qn_s[zi] = 0.0;
      sy += S->qn_s[zi] * y[zi]; ss += S->qn_s[zi] * S->qn_s[zi]; yy += y[zi] * y[zi];
    }
  }
  if (!(sy > 1.4901161193847656e-08 * sqrt(ss) * sqrt(yy))) {
    /* Ipopt: limited_memory_max_skipping = 2 consecutive skips, then the approximation starts again */
    if (++S->qn_skipped >= 2) { S->qn_k = 0; S->qn_skipped = 0; }
    free(y);
    return;
  }
  S->qn_skipped = 0;
  if (S->qn_k == S->qn_m) {
    memmove(S->qn_S, S->qn_S + Nz, (size_t)(S->qn_m - 1) * Nz * sizeof(double));
    memmove(S->qn_Y, S->qn_Y + Nz, (size_t)(S->qn_m - 1) * Nz * sizeof(double));
    S->qn_k--;
  }
  memcpy(S->qn_S + (size_t)S->qn_k * Nz, S->qn_s, Nz * sizeof(double));
  memcpy(S->qn_Y + (size_t)S->qn_k * Nz, y, Nz * sizeof(double));
  S->qn_k++;
  /* Ipopt limited_memory_initialization = scalar1: sigma = s'y / s's of the newest pair */
  S->qn_sigma = fmin(1e8, fmax(1e-8, sy / ss));
  free(y);
}

/* step of the eliminated slack multipliers, fraction to the boundary and the directional derivative of the barrier objective for
 * a step (dz, dlam) that did not come out of backward_sweep (same formulas) */
static void step_info(port_solver* S) {
  const port_options* o = &S->o;
  const int T = S->T;
  const double mu = S->mu, tau = fmax(o->tau_min, 1.0 - mu);
  double gphid = 0, apmax = 1.0, admax = 1.0;
  for (int t = T - 1; t >= 0; --t) {
    stage_t* s = &S->st[t];
    const int np = np_of(S, t), q = q_of(S, t), ny = ny_of(S, t), z0 = zoff(S, t);
    for (int i = 0; i < np; ++i) {
      const double dp = S->dz[z0 + i];
      gphid += s->rp[i] * dp;
      const double lo = S->lo[z0 + i], hi = S->hi[z0 + i], p = S->z[z0 + i];
      if (lo == hi) continue;
      if (finite_lo(lo)) {
        const double zl = S->zl[z0 + i], gap = p - lo, dzl = mu / gap - zl - (zl / gap) * dp;
        if (dp < 0.0) apmax = fmin(apmax, -tau * gap / dp);
        if (dzl < 0.0) admax = fmin(admax, -tau * zl / dzl);
        gphid -= mu / gap * dp;
      }
      if (finite_hi(hi)) {
        const double zu = S->zu[z0 + i], gap = hi - p, dzu = mu / gap - zu + (zu / gap) * dp;
        if (dp > 0.0) apmax = fmin(apmax, tau * gap / dp);
        if (dzu < 0.0) admax = fmin(admax, -tau * zu / dzu);
        gphid += mu / gap * dp;
      }
    }
    for (int j = 0; j < q; ++j) {
      const double dnu = S->dlam[S->ccoff[t] + j], nu = S->lam[S->ccoff[t] + j], r = s->c[j];
      double dsv = 0.0;
      const int k = slack_of(S, t, j);
      if (k >= 0) {
        const double sv = S->s[k], zv = S->zs[k];
        dsv = -(sv / zv) * (nu - mu / sv + dnu);
        const double dzs = mu / sv - zv - (zv / sv) * dsv;
        S->ds[k] = dsv;
        if (dsv < 0.0) apmax = fmin(apmax, -tau * sv / dsv);
        if (dzs < 0.0) admax = fmin(admax, -tau * zv / dzs);
        gphid -= mu / sv * dsv;
      }
      gphid += nu * (r - o->delta_c * dnu + dsv);
    }
    for (int k = 0; k < ny; ++k) gphid += lam_dyn(S, t)[k] * (s->d[k] - o->delta_c * dlam_dyn(S, t)[k]);
  }
  S->gphid = gphid; S->alpha_pmax = apmax; S->alpha_dmax = admax;
}

/* K = K0 - U M^-1 U', U = [W; 0], W = [sigma S, Y], M = [[sigma S'S, L], [L', -D]]:
 *   v = v0 + Z C^-1 U' v0,  v0 = K0^-1 b,  Z = K0^-1 U,  C = M - U' Z        (Sherman-Morrison-Woodbury)
 * inertia(K) = inertia(K0) + inertia(C) - inertia(M) (Haynsworth, both ways round): K0 must have (Nz, Nc, 0) and C as many
 * negative eigenvalues as M, else delta_w goes up the ladder -- with curvature-checked pairs that is the exception */
static void qn_factor_solve(port_solver* S) {
  const port_options* o = &S->o;
  const int Nz = S->Nz, Nc = S->Nc, k = S->qn_k, m2 = 2 * k;
  const double sig = S->qn_sigma;
  /* first delta_w: the rules of conv_body (the ladder state delta_last stays 0 in this mode: gam = 0 never records it) */
  double dw = 0.0;
  /* (round 6: after a failed line search the escalation starts from the delta_w the rejected direction was computed with -- with
   *  delta_last = 0 for ever the rule gave delta_w_init again and again: same point, same direction, same null step, for the rest of
   *  the iterations -- 3 of 512 acrobot T = 101 seeds here, 24 of 4 096 on the GPU) */
  if (S->ls_fail) dw = fmin(o->delta_w_exact_cap, fmax(10.0 * fmax(S->delta_last, S->delta_w), o->delta_w_init));
  else if (S->delta_last > 1.1 * o->delta_w_init && S->full_streak < 2) dw = fmax(o->delta_w_init, o->kappa_w_minus * S->delta_last);
  if (S->ls_mode == 1 && o->pen_gn) dw = fmax(dw, o->delta_w_init);
  double* U = (double*)malloc((size_t)(m2 > 0 ? m2 : 1) * Nz * sizeof(double));
  double* Zx = (double*)malloc((size_t)(m2 > 0 ? m2 : 1) * Nz * sizeof(double));
  double* zc = (double*)malloc((Nc > 0 ? Nc : 1) * sizeof(double));
  double* zero_c = (double*)calloc(Nc > 0 ? Nc : 1, sizeof(double));
  double Mm[4 * QN_MAX * QN_MAX], Cm[4 * QN_MAX * QN_MAX], rhs[2 * QN_MAX];
  for (int j = 0; j < k; ++j)
    for (int i = 0; i < Nz; ++i) { U[(size_t)j * Nz + i] = sig * S->qn_S[(size_t)j * Nz + i]; U[(size_t)(k + j) * Nz + i] = S->qn_Y[(size_t)j * Nz + i]; }
  for (int a = 0; a < k; ++a)
    for (int b = 0; b < k; ++b) {
      double ss = 0, sy = 0;
      for (int i = 0; i < Nz; ++i) { ss += S->qn_S[(size_t)a * Nz + i] * S->qn_S[(size_t)b * Nz + i]; sy += S->qn_S[(size_t)a * Nz + i] * S->qn_Y[(size_t)b * Nz + i]; }
      Mm[a * m2 + b] = sig * ss;
      Mm[a * m2 + k + b] = a > b ? sy : 0.0;           /* L: strictly lower part of S'Y */
      Mm[(k + b) * m2 + a] = a > b ? sy : 0.0;
      Mm[(k + a) * m2 + k + b] = a == b ? -sy : 0.0;   /* -D */
    }
  int tiny_m = 0;
  const int neg_m = m2 > 0 ? small_inertia(Mm, m2, &tiny_m) : 0;
  int ok = 0;
  for (int attempt = 0;; ++attempt) {
    ok = forward_sweep(S, dw, 0.0, NULL);
    S->nfact++;
    if (ok && m2 > 0) {
      for (int j = 0; j < m2; ++j) solve_stored(S, U + (size_t)j * Nz, zero_c, Zx + (size_t)j * Nz, zc);
      for (int a = 0; a < m2; ++a)
        for (int b = 0; b < m2; ++b) {
          double acc = 0;
          for (int i = 0; i < Nz; ++i) acc += U[(size_t)a * Nz + i] * Zx[(size_t)b * Nz + i];
          Cm[a * m2 + b] = Mm[a * m2 + b] - acc;
        }
      for (int a = 0; a < m2; ++a) for (int b = 0; b < a; ++b) { const double av = 0.5 * (Cm[a * m2 + b] + Cm[b * m2 + a]); Cm[a * m2 + b] = Cm[b * m2 + a] = av; }
      int tiny_c = 0;
      const int neg_c = small_inertia(Cm, m2, &tiny_c);
      if (neg_c != neg_m || tiny_c) ok = 0;
    }
    if (ok || attempt >= o->max_refactor) break;
    dw = dw == 0.0 ? o->delta_w_init : dw * o->kappa_w_plus;   /* retry_update_t, gam = 0 */
    if (dw > o->delta_w_max) dw = o->delta_w_max;
  }
  backward_sweep(S);                                   /* v0 = K0^-1 (-r) in (dz, dlam) */
  if (m2 > 0) {
    for (int a = 0; a < m2; ++a) { double acc = 0; for (int i = 0; i < Nz; ++i) acc += U[(size_t)a * Nz + i] * S->dz[i]; rhs[a] = acc; }
    small_solve(Cm, rhs, m2);
    /* the multiplier part of Z: one more pass per column would double the solves; dlam follows from dz through the stage rows
     * instead -- recompute it with ONE solve of K0 for the corrected right-hand side:  K0 v = b + U (M^-1 U' v)  and
     * M^-1 U' v = C^-1 U' v0 (push-through identity) */
    double* bx = (double*)malloc(Nz * sizeof(double));
    double* bc = (double*)malloc((Nc > 0 ? Nc : 1) * sizeof(double));
    double* vx = (double*)malloc(Nz * sizeof(double));
    /* b = -(r_p'; c'): rebuild it from K0 v0: cheaper to solve for the correction only: K0 dv = U q */
    for (int i = 0; i < Nz; ++i) { double acc = 0; for (int a = 0; a < m2; ++a) acc += U[(size_t)a * Nz + i] * rhs[a]; bx[i] = acc; }
    solve_stored(S, bx, zero_c, vx, bc);
    for (int i = 0; i < Nz; ++i) S->dz[i] += vx[i];
    for (int i = 0; i < Nc; ++i) S->dlam[i] += bc[i];
    free(bx); free(bc); free(vx);
    step_info(S);
  }
  S->delta_w = dw;
  if (dw == 0.0) S->delta_last = 0.0;
  S->gamma = 0.0;
  S->ls_fail = ok ? 0 : 1;
  free(U); free(Zx); free(zc); free(zero_c);
}

/* inertia correction: k_conv's choice of the first delta_w + k_kkt_sep's ladder */
/* floor of the decaying delta_w: Ipopt's delta_w^min = 1e-20 (IpPDPerturbationHandler), rounds 2 - 5 used delta_w_init = 1e-4 --
 * DTO_DW_FLOOR=1e-4 restores that (tools/port_stats.py: acrobot T = 1000, 256 seeds: 227 converge with 1e-4, 256 with 1e-20) */
static double dw_floor(const port_options* o) {
  static double env = -2;
  if (env < -1) env = getenv("DTO_DW_FLOOR") ? atof(getenv("DTO_DW_FLOOR")) : -1.0;
  return env > 0 ? env : o->delta_w_min;
}
static void factor_solve(port_solver* S) {
  if (S->qn_mode == 2) { qn_factor_solve(S); return; }
  const port_options* o = &S->o;
  const double dlast = S->delta_last;
  double dw = 0.0, gam = 1.0;
  if (S->ls_fail) dw = fmin(o->delta_w_exact_cap, fmax(10.0 * dlast, o->delta_w_init));
  else if (dlast > 1.1 * dw_floor(o) && S->full_streak < 2) dw = fmax(dw_floor(o), o->kappa_w_minus * dlast);
  {
    static double lm = -1, lup, ldn, lthr;
    if (lm < 0) { lm = getenv("DTO_LM") ? atof(getenv("DTO_LM")) : 0.0; lup = getenv("DTO_LM_UP") ? atof(getenv("DTO_LM_UP")) : 4.0;
      ldn = getenv("DTO_LM_DN") ? atof(getenv("DTO_LM_DN")) : 1.0 / 3.0; lthr = getenv("DTO_LM_THR") ? atof(getenv("DTO_LM_THR")) : 0.25; }
    if (lm > 0 && S->iter > 0) {
      if (S->alpha >= S->alpha_pmax) S->delta_lm = S->delta_lm * ldn > o->delta_w_init ? S->delta_lm * ldn : 0.0;
      else if (S->alpha <= lthr * S->alpha_pmax && (!getenv("DTO_LM_NEAR") || S->th1 <= atof(getenv("DTO_LM_NEAR")) * S->theta_min)) S->delta_lm = fmin(lm, fmax(o->delta_w_init, lup * fmax(S->delta_lm, S->delta_w)));
      if (dw < S->delta_lm) dw = S->delta_lm;
    }
  }
  {
    static double up = -1, thr = 0.2;
    if (up < 0) { up = getenv("DTO_SHORT_UP") ? atof(getenv("DTO_SHORT_UP")) : 0.0; if (getenv("DTO_SHORT_ALPHA")) thr = atof(getenv("DTO_SHORT_ALPHA")); }
    if (up > 0 && S->iter > 0 && !S->ls_fail && S->alpha < thr) dw = fmin(o->delta_w_exact_cap, fmax(o->delta_w_init, up * S->delta_w));
  }
  /* penalty phase: Gauss-Newton model -- the constraint curvature lam' d'' + nu' c'' is dropped, delta_w >= delta_w_init: far from
   * the manifold the exact Hessian is so indefinite that the ladder ends at delta_w ~ 10 .. 100 anyway (the first factorisation
   * that succeeds at iteration 1 of an acrobot solve has delta_w = 26 against a cost Hessian of 0.2), i.e. its curvature is swamped
   * while every probe costs a sweep; the Gauss-Newton matrix has the right inertia by construction (ONE factorisation) and, with
   * the penalty line search, takes fewer iterations (acrobot T=1000, 128 seeds: median 47 instead of 57; T=101: 38 / 46) */
  if (S->ls_mode == 1 && o->pen_gn) { gam = 0.0; dw = fmax(dw, o->delta_w_init); if (S->delta_lm > dw) dw = S->delta_lm; }
  int ok = 0;
  for (int attempt = 0;; ++attempt) {
    ok = forward_sweep(S, dw, gam, NULL);
    S->nfact++;
    if (ok || attempt >= o->max_refactor) break;
    if (gam != 0.0) {
      const int skip_ladder = (S->gamma == 0.0) && (S->iter % 4 != 0);
      const double dw_failed = dw;
      if (dw == 0.0 && !skip_ladder) dw = (dlast == 0.0) ? o->delta_w_init : fmax(dw_floor(o), o->kappa_w_minus * dlast);
      else if (!skip_ladder) dw *= (dlast == 0.0) ? o->kappa_w_plus_first : o->kappa_w_plus;
      {   /* experiment: jump the ladder to the level the most negative primal pivot of the failed sweep asks for */
        static double kj = -1; if (kj < 0) kj = getenv("DTO_PIV_JUMP") ? atof(getenv("DTO_PIV_JUMP")) : 0.0;
        if (kj > 0 && !skip_ladder && S->piv_min < 0) { const double want = dw_failed + kj * -S->piv_min; if (want > dw) dw = want; }
      }
      if (skip_ladder || dw > o->delta_w_exact_cap) { gam = 0.0; dw = o->delta_w_init; }
    } else {
      dw *= o->kappa_w_plus;
      if (dw > o->delta_w_max) dw = o->delta_w_max;
    }
  }
  backward_sweep(S);
  S->delta_w = dw;
  if (dw > 0.0 && gam != 0.0) S->delta_last = dw;
  if (dw == 0.0) S->delta_last = 0.0;
  S->gamma = gam;
  S->ls_fail = ok ? 0 : 1;
}

/* barrier objective and l1 constraint violation at (z + alpha dz, s + alpha ds); res (may be NULL): the residuals, laid
 * out like lam */
static void trial_point(port_solver* S, double alpha, double* phi, double* th, double* res) {
  const port_model* M = S->M;
  const int n = S->n, T = S->T;
  const double mu = S->mu;
  double f = 0, t1 = 0;
  double xk[MAXP], yk[MAXN], d[MAXN], c[MAXQ], l;
  for (int t = 0; t < T; ++t) {
    const int np = np_of(S, t), z0 = zoff(S, t);
    for (int i = 0; i < np; ++i) xk[i] = S->z[z0 + i] + alpha * S->dz[z0 + i];
    if (t < T - 1) {
      for (int i = 0; i < n; ++i) yk[i] = S->z[zoff(S, t + 1) + i] + alpha * S->dz[zoff(S, t + 1) + i];
      M->costval(xk, xk + n, &l);
    } else {
      M->costTval(xk, &l);
    }
    f += l;
    for (int i = 0; i < np; ++i) {
      const double lo = S->lo[z0 + i], hi = S->hi[z0 + i];
      if (lo == hi) continue;
      if (finite_lo(lo)) f -= mu * log(xk[i] - lo);
      if (finite_hi(hi)) f -= mu * log(hi - xk[i]);
    }
    if (t < T - 1) {
      M->dynres(xk, xk + n, yk, d);
      for (int i = 0; i < n; ++i) { t1 += fabs(d[i]); if (res) res[t * n + i] = d[i]; }
    }
    if (S->con[t] >= 0) {
      const port_con_class* C = &M->cls[S->con[t]];
      C->val(xk, xk + n, c);
      for (int j = 0; j < C->nc; ++j) {
        double r = c[j];
        const int k = slack_of(S, t, j);
        if (k >= 0) { const double sk = S->s[k] + alpha * S->ds[k]; r += sk; f -= mu * log(sk); }
        t1 += fabs(r);
        if (res) res[S->ccoff[t] + j] = r;
      }
    }
  }
  *phi = f; *th = t1;
}

static int filter_ok(const port_solver* S, double tk, double pk) {
  const double G_TH = 1e-5, G_PHI = 1e-8;
  const int nf = S->filter_n < FILTER_CAP ? S->filter_n : FILTER_CAP;
  for (int i = 0; i < nf; ++i) {
    const double tf = S->filt[2 * i], pf = S->filt[2 * i + 1];
    if (!(tk <= (1.0 - G_TH) * tf || pk <= pf - G_PHI * tf)) return 0;
  }
  return 1;
}

/* acceptance of one trial (theta, phi) at step size alpha against the current iterate (Waechter & Biegler, A-5.4) */
static int trial_ok(const port_solver* S, double alpha, double tk, double pk, int* ftype) {
  const double G_TH = 1e-5, G_PHI = 1e-8, S_TH = 1.1, S_PHI = 2.3, ETA = 1e-8, DELTA = 1.0;
  const double th0 = S->th1, phi0 = S->merit0, dphi = S->gphid;
  int ok = (tk == tk) && (pk == pk) && tk <= S->theta_max;
  const int sw = dphi < 0.0 && alpha * pow(-dphi, S_PHI) > DELTA * pow(th0, S_TH);
  const int armijo = pk <= phi0 + ETA * alpha * dphi + 1e-13 * fabs(phi0);
  if (ok) {
    if (sw && th0 <= S->theta_min) ok = armijo;
    else ok = (tk <= (1.0 - G_TH) * th0) || (pk <= phi0 - G_PHI * th0);
  }
  *ftype = sw && armijo;
  return ok;
}

static void filter_augment(port_solver* S) {
  const double G_TH = 1e-5, G_PHI = 1e-8;
  const int slot = S->filter_n % FILTER_CAP;
  S->filt[2 * slot] = (1.0 - G_TH) * S->th1;
  S->filt[2 * slot + 1] = S->merit0 - G_PHI * S->th1;
  S->filter_n++;
}

/* second-order correction (Waechter & Biegler 2006, section 2.4; Ipopt max_soc = 4, kappa_soc = 0.99): the full
 * fraction-to-the-boundary step was rejected and did not reduce the violation -> re-solve the same matrix with the
 * constraint residual alpha c(x_k) + c(x_k + alpha d); accept the corrected step if the filter takes it */
static int second_order_correction(port_solver* S, double th_first) {
  const port_options* o = &S->o;
  const double amax = S->alpha_pmax, admax0 = S->alpha_dmax, gph = S->gphid;
  memcpy(S->save_dz, S->dz, S->Nz * sizeof(double)); memcpy(S->save_dlam, S->dlam, S->Nc * sizeof(double));
  memcpy(S->save_ds, S->ds, S->Ni * sizeof(double));
  double phi, th;
  trial_point(S, amax, &phi, &th, S->soc_buf);           /* c(x_k + alpha d) */
  trial_point(S, 0.0, &phi, &th, S->soc_c);              /* c(x_k) */
  for (int i = 0; i < S->Nc; ++i) S->soc_c[i] = amax * S->soc_c[i] + S->soc_buf[i];
  double th_prev = th_first;
  for (int it = 0; it < o->max_soc; ++it) {
    forward_sweep(S, S->delta_w, S->gamma, S->soc_c);
    backward_sweep(S);                                    /* new dz, dlam, ds and their fraction-to-the-boundary limits */
    const double alpha_soc = S->alpha_pmax;
    S->gphid = gph;
    double pk, tk;
    trial_point(S, alpha_soc, &pk, &tk, S->soc_buf);
    int ft = 0;
    const int ok = trial_ok(S, alpha_soc, tk, pk, &ft) && filter_ok(S, tk, pk);
    if (ok) {
      S->alpha = alpha_soc; S->ls_fail = 0; S->ls_kind = 4;
      if (!ft) filter_augment(S);
      S->nsoc++;
      return 1;
    }
    if (!(tk < 0.99 * th_prev)) break;
    th_prev = tk;
    for (int i = 0; i < S->Nc; ++i) S->soc_c[i] = alpha_soc * S->soc_c[i] + S->soc_buf[i];
  }
  memcpy(S->dz, S->save_dz, S->Nz * sizeof(double)); memcpy(S->dlam, S->save_dlam, S->Nc * sizeof(double));
  memcpy(S->ds, S->save_ds, S->Ni * sizeof(double));
  S->alpha_pmax = amax; S->alpha_dmax = admax0; S->gphid = gph;
  return 0;
}

/* k_linesearch + k_ls_reduce */
static void line_search(port_solver* S) {
  double phi[LS_TRIALS], th[LS_TRIALS];
  const double amax = S->alpha_pmax * (S->ls_mode == 1 ? S->ascale : 1.0);
  double alpha = amax;
  for (int k = 0; k < LS_TRIALS; ++k) { trial_point(S, alpha, &phi[k], &th[k], NULL); alpha *= 0.5; }
  const double th0 = S->th1;
  if (getenv("DTO_LS_TRACE") && S->iter >= atoi(getenv("DTO_LS_TRACE")) && S->iter < atoi(getenv("DTO_LS_TRACE")) + 12) {
    double dn = 0; for (int i = 0; i < S->Nz; ++i) dn = fmax(dn, fabs(S->dz[i]));
    fprintf(stderr, "it %d th0 %.3e phi0 %.10e gphid %.3e amax %.3g |dz|inf %.3e dw %.2e thmin %.2e nfilt %d wd %d\n", S->iter, th0, S->merit0, S->gphid, amax, dn, S->delta_w, S->theta_min, S->filter_n, S->watchdog);
    double a2 = amax; for (int k = 0; k < LS_TRIALS; ++k) { int ft; int ok = trial_ok(S, a2, th[k], phi[k], &ft); fprintf(stderr, "   a %.4f th %.3e dphi %.3e ok %d ftype %d filt %d\n", a2, th[k], phi[k] - S->merit0, ok, ft, filter_ok(S, th[k], phi[k])); a2 *= 0.5; }
  }
  double chosen = -1.0;
  int ftype = 0, best = 0;
  /* Two-phase globalisation (round 5; DESIGN.md section 5, profiles/r05/third_party_cfg3_T1000.json): far from the constraint
   * manifold (theta_inf > ls_switch) the filter takes any step that lowers the violation, whatever it does to the objective --
   * from the reference's straight-line guesses that is a jump to objective values 20 x the guess's, followed by hundreds of
   * iterations back down along the manifold.  There the step size is chosen on the l1 exact-penalty function instead; once the
   * iterate is near the manifold the filter (with its fast local convergence) takes over for good. */
  if (S->ls_mode == 1) {
    /* l1 exact-penalty merit phi + nu theta, Armijo backtracking (Nocedal & Wright 18.3; Ipopt's line_search_method=penalty) */
    const double rho = 0.1, eta = 1e-4, dphi = S->gphid;
    if (th0 > 0.0) {
      const double need = dphi / ((1.0 - rho) * th0);
      if (S->nu_pen < need) S->nu_pen = need + 1.0;
    }
    const double nu = S->nu_pen, m0 = S->merit0 + nu * th0, D = dphi - nu * th0;
    alpha = amax;
    for (int k = 0; k < LS_TRIALS; ++k) {
      const double mk = phi[k] + nu * th[k];
      if (mk == mk && mk <= m0 + eta * alpha * D + 1e-13 * fabs(m0)) { chosen = alpha; break; }
      alpha *= 0.5;
    }
    if (chosen < 0.0) { chosen = 0.0; S->alpha_dmax = 0.0; S->ascale = fmax(S->ascale / 256.0, 1e-12); }
    else if (chosen >= amax) S->ascale = fmin(1.0, S->ascale * 4.0);
    S->ls_fail = 0;
    S->ls_kind = chosen > 0.0 ? 5 : -1;
    S->alpha = chosen;
    S->full_streak = (chosen >= amax) ? S->full_streak + 1 : 0;
    return;
  }
  const int wd_left = S->watchdog, watchdog = wd_left > 0;  /* rollback-free watchdog: see k_ls_reduce */
  if (S->o.max_soc > 0 && !watchdog) {
    int ft;
    const int ok0 = trial_ok(S, amax, th[0], phi[0], &ft) && filter_ok(S, th[0], phi[0]);
    if (!ok0 && th[0] >= th0 && second_order_correction(S, th[0])) {
      S->full_streak = S->full_streak + 1; S->short_streak = 0;
      return;
    }
  }
  alpha = amax;
  for (int k = 0; k < LS_TRIALS; ++k) {
    const double tk = th[k], pk = phi[k];
    if (th[k] < th[best] || !(th[best] == th[best])) best = k;
    int ft;
    int ok = trial_ok(S, alpha, tk, pk, &ft);
    /* watchdog steps skip the filter but may not let the violation explode (there is no rollback): theta <= 3 max(theta_0, 1) */
    if (watchdog) ok = (tk == tk) && (pk == pk) && tk <= S->theta_max && tk <= 3.0 * fmax(th0, getenv("DTO_WD_FLOOR") ? atof(getenv("DTO_WD_FLOOR")) : 1.0);
    if (ok && !watchdog) ok = filter_ok(S, tk, pk);
    if (ok) { chosen = alpha; ftype = ft; break; }
    alpha *= 0.5;
  }
  int augment;
  if (chosen < 0.0) {
    double ab = amax;
    for (int k = 0; k < best; ++k) ab *= 0.5;
    if (th[best] == th[best] && th[best] < th0) chosen = ab; else chosen = alpha * 2.0;
    /* every trial is catastrophic (even the most feasible one multiplies the violation by > 100): a direction like that is
     * not worth any step -- stay where we are, primal and dual, and let ls_fail regularise the next system more */
    if (!(th[best] <= LS_NULL_STEP * fmax(th0, 1.0))) { chosen = 0.0; S->alpha_dmax = 0.0; }
    if (getenv("DTO_NULL_ALWAYS") && !(th[best] < th0)) { chosen = 0.0; S->alpha_dmax = 0.0; }
    S->ls_fail = 1; augment = 1;
  } else { S->ls_fail = 0; augment = !ftype && !watchdog; }
  if (watchdog) { S->watchdog = wd_left - 1; S->short_streak = 0; }
  else if (S->o.watchdog_trigger > 0) {
    const int streak = (chosen < amax) ? S->short_streak + 1 : 0;
    if (streak >= S->o.watchdog_trigger) { S->watchdog = S->o.watchdog_trials; S->short_streak = 0; }
    else S->short_streak = streak;
  }
  if (augment) filter_augment(S);
  S->ls_kind = chosen < 0.0 ? -1 : (watchdog ? 3 : (ftype ? 1 : 2));
  S->alpha = chosen;
  S->full_streak = (chosen >= amax) ? S->full_streak + 1 : 0;
}

/* k_update */
static void update(port_solver* S) {
  const double mu = S->mu, al = S->alpha, ad = S->alpha_dmax, KSIG = 1e10;
  for (int i = 0; i < S->Nz; ++i) {
    const double p = S->z[i], dp = S->dz[i], pn = p + al * dp, lo = S->lo[i], hi = S->hi[i];
    if (lo != hi) {
      if (finite_lo(lo)) {
        const double zl = S->zl[i], gap = p - lo, dzl = mu / gap - zl - (zl / gap) * dp, gn = pn - lo;
        S->zl[i] = fmin(fmax(zl + ad * dzl, mu / (KSIG * gn)), KSIG * mu / gn);
      }
      if (finite_hi(hi)) {
        const double zu = S->zu[i], gap = hi - p, dzu = mu / gap - zu + (zu / gap) * dp, gn = hi - pn;
        S->zu[i] = fmin(fmax(zu + ad * dzu, mu / (KSIG * gn)), KSIG * mu / gn);
      }
    }
    S->z[i] = pn;
  }
  for (int k = 0; k < S->Ni; ++k) {
    const double sv = S->s[k], zv = S->zs[k], dsv = S->ds[k];
    const double dzs = mu / sv - zv - (zv / sv) * dsv, sn = sv + al * dsv;
    S->s[k] = sn;
    S->zs[k] = fmin(fmax(zv + ad * dzs, mu / (KSIG * sn)), KSIG * mu / sn);
  }
  for (int i = 0; i < S->Nc; ++i) S->lam[i] += al * S->dlam[i];
}

/* one iteration: EVAL -> CONV -> FACTOR_SOLVE -> LINESEARCH -> UPDATE; returns 1 if an iteration was executed */
int port_iterate(port_solver* S) {
  if (S->status != 0) return 0;
  eval_all(S);
  if (S->qn_mode == 2) qn_update(S);
  convergence(S);
  if (S->status != 0) return 0;
  factor_solve(S);
  line_search(S);
  if (S->qn_mode == 2) qn_save(S);
  update(S);
  S->iter++;
  return 1;
}

/* accessors for ctypes */
int port_status(const port_solver* S) { return S->status; }
int port_iterations(const port_solver* S) { return S->iter; }
int port_nfact(const port_solver* S) { return S->nfact; }
int port_nsoc(const port_solver* S) { return S->nsoc; }
int port_ls_kind(const port_solver* S) { return S->ls_kind; }
double port_qn_sigma(const port_solver* S) { return S->qn_sigma; }
int port_qn_pairs(const port_solver* S) { return S->qn_k; }
int port_num_variables(const port_solver* S) { return S->Nz; }
int port_num_constraint(const port_solver* S) { return S->Nc; }
int port_num_slacks(const port_solver* S) { return S->Ni; }
double port_objective(const port_solver* S) { return S->f; }
double port_constr_viol(const port_solver* S) { return S->thinf; }
double port_dual_inf(const port_solver* S) { return S->dinf; }
double port_alpha(const port_solver* S) { return S->alpha; }
double port_delta_w(const port_solver* S) { return S->delta_w; }
double port_mu(const port_solver* S) { return S->mu; }
const double* port_z(const port_solver* S) { return S->z; }
/* multipliers in the reference order: dynamics rows, then stage rows t = 1..T */
const double* port_lam(const port_solver* S) { return S->lam; }

/* batched driver for the CPU baseline: solves (iters_per_instance <= 0) or runs a fixed number of iterations of B
 * instances, OpenMP over instances when built with -fopenmp.  Z0: [B][Nz]; out_iters / out_status: [B]; returns the
 * total number of iterations executed. */
long port_run_batch(const char* model, int T, const int* con, const double* lo, const double* hi, int max_iter,
                    int iters_per_instance, int B, const double* Z0, int* out_iters, int* out_status, long* out_nfact) {
  long total = 0, nfact = 0;
  /* one solver per THREAD, reused for its instances (port_begin resets it): creating one per instance meant a calloc / free of
   * the ~3 MB stage workspace each time, i.e. page faults under the process-wide mmap lock -- with many threads the batch
   * then scaled with the kernel's page-fault path instead of the cores (8 threads: 1.9x one core) */
#pragma omp parallel reduction(+ : total, nfact)
  {
    port_solver* S = port_create(model, T, con, lo, hi, max_iter);
#pragma omp for schedule(dynamic)
    for (int b = 0; b < B; ++b) {
      if (!S) continue;
      port_begin(S, Z0 + (size_t)b * S->Nz);
      int k = 0;
      while ((iters_per_instance <= 0 || k < iters_per_instance) && port_iterate(S)) ++k;
      total += k; nfact += S->nfact;
      if (out_iters) out_iters[b] = S->iter;
      if (out_status) out_status[b] = S->status;
    }
    port_destroy(S);
  }
  if (out_nfact) *out_nfact = nfact;
  return total;
}
