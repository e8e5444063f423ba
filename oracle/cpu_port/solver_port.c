/*
 * ORACLE / CPU baseline (test infrastructure, never linked into the product): serial C port of the
 * interior-point / SQP iteration for equality-constrained stage problems with pinned end states
 * (BASELINE configs 1 and 3: pendulum, acrobot).  It restates, for ONE instance on ONE core:
 *   - the reference's stage loops: cost / gradient! / hessian!          src/costs.jl:49-73
 *                                  constraints! / jacobian! / hessian_lagrangian!  src/dynamics.jl:103-127
 *   - the KKT system the reference sketches at examples/pendulum/pendulum.jl:138-198
 *         [ H + dw I  J' ; J  -dc I ] [dz; dlam] = -[ grad L ; c ]
 *     solved stage by stage (block-tridiagonal LDL^T),
 *   - the part the reference delegates to Ipopt (src/solver.jl:45-47): inertia correction,
 *     filter line search (Waechter & Biegler 2006), convergence test with the reference Options
 *     (tol 1e-6, constr_viol_tol 1e-3, dual_inf_tol 1, src/options.jl:7-14).
 * Variables follow the reference order z = [x_1;u_1;...;x_T] (src/dynamics.jl:188-195).
 * Used (a) as bench.py's cpu_baseline ("port", 1 core) and (b) by tests to cross-check the GPU
 * solver's iterates.  PARITY: Ipopt itself cannot run here, so this pins the GPU path against an
 * independent implementation of the same algorithm, not against Ipopt's iterates.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define MAXN 8
#define MAXP 10
#define MAXBD 26
#define TRI(i, j) ((i) * ((i) + 1) / 2 + (j))
#define LS_TRIALS 8
#define FILTER_CAP 24

typedef void (*cost_fn)(const double*, const double*, double*, double*, double*);
typedef void (*costT_fn)(const double*, double*, double*, double*);
typedef void (*dyn_fn)(const double*, const double*, const double*, const double*, double*, double*, double*,
                       double*, double*, double*);
typedef void (*dynres_fn)(const double*, const double*, const double*, double*);
typedef void (*costval_fn)(const double*, const double*, double*);
typedef void (*costTval_fn)(const double*, double*);

typedef struct {
  int n, m, T;
  cost_fn cost; costT_fn costT; dyn_fn dyn; dynres_fn dynres; costval_fn costval; costTval_fn costTval;
  double x1[MAXN], xT[MAXN];
  /* options (mirrors csrc/dto_solver.cpp default_opts) */
  double tol, s_max, dual_inf_tol, constr_viol_tol, delta_c, delta_w_init, delta_w_max, delta_w_exact_cap, piv_tol;
  int max_iter, max_refactor, watchdog_trigger, watchdog_trials;
} port_problem;

typedef struct {
  double W[MAXP * (MAXP + 1) / 2], WD[MAXP * (MAXP + 1) / 2], V[MAXP * MAXN], YY[MAXN * (MAXN + 1) / 2];
  double F[MAXN * MAXP], E[MAXN * MAXN], rp[MAXP], d[MAXN], c[MAXN];
  /* factors */
  double L[MAXBD * MAXBD], dinv[MAXBD], X[MAXBD * MAXN], w[MAXBD];
} stage_t;

typedef struct {
  port_problem P;
  int Nz, Nc;
  double *z, *lam /* dyn rows then pin rows (first, last) */, *dz, *dlam;
  stage_t* st;
  int status, iter, nfact, filter_n, ls_fail, full_streak, short_streak, watchdog;
  double f, th1, thinf, dinf, delta_w, delta_last, gamma, alpha, gphid, theta_max, theta_min;
  double filt[2 * FILTER_CAP];
} port_solver;

static int np_of(const port_problem* P, int t) { return t < P->T - 1 ? P->n + P->m : P->n; }
static int q_of(const port_problem* P, int t) { return (t == 0 || t == P->T - 1) ? P->n : 0; }
static int ny_of(const port_problem* P, int t) { return t < P->T - 1 ? P->n : 0; }
static int zoff(const port_problem* P, int t) { return t * (P->n + P->m); }
static double* lam_dyn(port_solver* S, int t) { return S->lam + t * S->P.n; }
static double* lam_pin(port_solver* S, int which) { return S->lam + (S->P.T - 1) * S->P.n + which * S->P.n; }
static double* dlam_dyn(port_solver* S, int t) { return S->dlam + t * S->P.n; }
static double* dlam_pin(port_solver* S, int which) { return S->dlam + (S->P.T - 1) * S->P.n + which * S->P.n; }

static double g_rho = 0.0;
static int g_nsoc = 0;
int port_nsoc(void) { return g_nsoc; }
port_solver* port_create(const port_problem* P) {
  port_solver* S = (port_solver*)calloc(1, sizeof(port_solver));
  S->P = *P;
  S->Nz = P->T * (P->n + P->m) - P->m;
  S->Nc = (P->T - 1) * P->n + 2 * P->n;
  S->z = (double*)calloc(S->Nz, sizeof(double));
  S->dz = (double*)calloc(S->Nz, sizeof(double));
  S->lam = (double*)calloc(S->Nc, sizeof(double));
  S->dlam = (double*)calloc(S->Nc, sizeof(double));
  S->st = (stage_t*)calloc(P->T, sizeof(stage_t));
  return S;
}

void port_destroy(port_solver* S) {
  if (!S) return;
  free(S->z); free(S->dz); free(S->lam); free(S->dlam); free(S->st); free(S);
}

void port_begin(port_solver* S, const double* z0) {
  memcpy(S->z, z0, S->Nz * sizeof(double));
  memset(S->lam, 0, S->Nc * sizeof(double));
  S->status = 0; S->iter = 0; S->nfact = 0; S->filter_n = 0; S->ls_fail = 0; S->full_streak = 0; S->short_streak = 0; S->watchdog = 0;
  S->delta_w = 0; S->delta_last = 0; S->gamma = 1.0; S->alpha = 0; S->theta_max = -1; S->theta_min = -1;
  g_rho = 0.0;
  if (getenv("PORT_DW0")) S->delta_last = 3.0 * atof(getenv("PORT_DW0"));
}

/* ---- derivative blocks of every stage + residual norms (k_stage_eval + k_conv on the GPU) ---- */
static void eval_all(port_solver* S) {
  const port_problem* P = &S->P;
  const int n = P->n, m = P->m, T = P->T;
  double f = 0, th1 = 0, thinf = 0, dinf = 0;
  for (int t = 0; t < T; ++t) {
    stage_t* s = &S->st[t];
    const double* x = S->z + zoff(P, t);
    const int np = np_of(P, t);
    double l;
    memset(s->WD, 0, sizeof(s->WD)); memset(s->V, 0, sizeof(s->V)); memset(s->YY, 0, sizeof(s->YY));
    if (t < T - 1) {
      const double* u = x + n;
      const double* y = S->z + zoff(P, t + 1);
      P->cost(x, u, &l, s->rp, s->W);
      P->dyn(x, u, y, lam_dyn(S, t), s->d, s->F, s->E, s->WD, s->V, s->YY);
      for (int i = 0; i < n; ++i)
        for (int j = 0; j < np; ++j) s->rp[j] += s->F[i * np + j] * lam_dyn(S, t)[i];
      for (int i = 0; i < n; ++i) { th1 += fabs(s->d[i]); thinf = fmax(thinf, fabs(s->d[i])); }
    } else {
      P->costT(x, &l, s->rp, s->W);
    }
    f += l;
    if (t > 0) { /* E_{t-1}' lam_{t-1} lands in the x rows of stage t */
      const stage_t* sp = &S->st[t - 1];
      for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) s->rp[j] += sp->E[i * n + j] * lam_dyn(S, t - 1)[i];
    }
    if (q_of(P, t)) {
      const double* target = t == 0 ? P->x1 : P->xT;
      const double* nu = lam_pin(S, t == 0 ? 0 : 1);
      for (int i = 0; i < n; ++i) {
        s->c[i] = x[i] - target[i];
        s->rp[i] += nu[i];
        th1 += fabs(s->c[i]); thinf = fmax(thinf, fabs(s->c[i]));
      }
    }
    for (int i = 0; i < np; ++i) dinf = fmax(dinf, fabs(s->rp[i]));
  }
  S->f = f; S->th1 = th1; S->thinf = thinf; S->dinf = dinf;
}

static void convergence(port_solver* S) {
  const port_problem* P = &S->P;
  double slam = 0;
  for (int i = 0; i < S->Nc; ++i) slam += fabs(S->lam[i]);
  const double sd = fmax(P->s_max, slam / (double)S->Nc) / P->s_max;
  const double e0 = fmax(S->dinf / sd, S->thinf);
  if (!(S->f == S->f) || !(S->th1 == S->th1) || !(S->dinf == S->dinf)) S->status = 3;
  else if (e0 <= P->tol && S->dinf <= P->dual_inf_tol && S->thinf <= P->constr_viol_tol) S->status = 1;
  else if (S->iter >= P->max_iter) S->status = 2;
  if (S->theta_max < 0) { S->theta_max = 1e4 * fmax(1.0, S->th1); S->theta_min = 1e-4 * fmax(1.0, S->th1); }
}

/* ---- block-tridiagonal LDL^T, forward sweep; returns 1 if the inertia is (Nz, Nc, 0) and no pivot is tiny ---- */
static int forward_sweep(port_solver* S, double dw, double gam) {
  const port_problem* P = &S->P;
  const int n = P->n, T = P->T;
  double Pm[MAXN * MAXN], py[MAXN];
  memset(Pm, 0, sizeof(Pm)); memset(py, 0, sizeof(py));
  int ok = 1, nneg = 0;
  for (int t = 0; t < T; ++t) {
    stage_t* s = &S->st[t];
    const int np = np_of(P, t), q = q_of(P, t), ny = ny_of(P, t), bd = np + q + ny;
    double A[MAXBD][MAXBD], y[MAXBD], X[MAXBD][MAXN];
    memset(A, 0, sizeof(A));
    for (int i = 0; i < np; ++i)
      for (int j = 0; j <= i; ++j) A[i][j] = s->W[TRI(i, j)] + gam * s->WD[TRI(i, j)];
    for (int i = 0; i < n; ++i)
      for (int j = 0; j <= i; ++j) A[i][j] += Pm[i * n + j];
    for (int i = 0; i < np; ++i) { A[i][i] += dw; y[i] = -s->rp[i] - (i < n ? py[i] : 0.0); }
    for (int j = 0; j < q; ++j) { A[np + j][j] = 1.0; A[np + j][np + j] = -P->delta_c; y[np + j] = -s->c[j]; }
    for (int k = 0; k < ny; ++k) {
      for (int i = 0; i < np; ++i) A[np + q + k][i] = s->F[k * np + i];
      A[np + q + k][np + q + k] = -P->delta_c;
      y[np + q + k] = -s->d[k];
    }
    memset(X, 0, sizeof(X));
    for (int i = 0; i < np; ++i) for (int c = 0; c < ny; ++c) X[i][c] = gam * s->V[i * ny + c];
    for (int k = 0; k < ny; ++k) for (int c = 0; c < ny; ++c) X[np + q + k][c] = s->E[k * ny + c];
    /* right-looking LDL^T, static order */
    for (int j = 0; j < bd; ++j) {
      double dj = A[j][j], cmax = 0;
      for (int i = j + 1; i < bd; ++i) cmax = fmax(cmax, fabs(A[i][j]));
      if (!(fabs(dj) > P->piv_tol * fmax(1.0, cmax))) { ok = 0; dj = (dj < 0 ? -1.0 : 1.0) * fmax(fabs(dj), P->piv_tol); }
      if (dj < 0) ++nneg;
      const double inv = 1.0 / dj;
      s->dinv[j] = inv;
      for (int i = j + 1; i < bd; ++i) {
        const double lij = A[i][j] * inv;
        for (int k = j + 1; k <= i; ++k) A[i][k] -= lij * A[k][j];
      }
      for (int i = j + 1; i < bd; ++i) A[i][j] *= inv;
    }
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

static void backward_sweep(port_solver* S) {
  const port_problem* P = &S->P;
  const int n = P->n, T = P->T;
  double xn[MAXN];
  memset(xn, 0, sizeof(xn));
  double gphid = 0;
  for (int t = T - 1; t >= 0; --t) {
    stage_t* s = &S->st[t];
    const int np = np_of(P, t), q = q_of(P, t), ny = ny_of(P, t), bd = np + q + ny;
    double v[MAXBD];
    for (int i = 0; i < bd; ++i) {
      double r = s->w[i];
      for (int c = 0; c < ny; ++c) r -= s->X[i * MAXN + c] * xn[c];
      v[i] = r * s->dinv[i];
    }
    for (int i = bd - 1; i >= 1; --i)
      for (int k = 0; k < i; ++k) v[k] -= s->L[i * MAXBD + k] * v[i];
    for (int i = 0; i < np; ++i) { S->dz[zoff(P, t) + i] = v[i]; gphid += s->rp[i] * v[i]; }
    if (q) {
      double* dnu = dlam_pin(S, t == 0 ? 0 : 1);
      const double* nu = lam_pin(S, t == 0 ? 0 : 1);
      for (int j = 0; j < q; ++j) { dnu[j] = v[np + j]; gphid += nu[j] * (s->c[j] - P->delta_c * dnu[j]); }
    }
    for (int k = 0; k < ny; ++k) {
      dlam_dyn(S, t)[k] = v[np + q + k];
      gphid += lam_dyn(S, t)[k] * (s->d[k] - P->delta_c * v[np + q + k]);
    }
    for (int i = 0; i < n; ++i) xn[i] = v[i];
  }
  S->gphid = gphid;
}

static void factor_solve(port_solver* S) {
  const port_problem* P = &S->P;
  const double dlast = S->delta_last;
  double dw = 0.0, gam = 1.0;
  if (S->ls_fail) dw = fmin(P->delta_w_exact_cap, fmax(10.0 * dlast, P->delta_w_init));  /* capped: see k_conv */
  else if (dlast > 1.1 * P->delta_w_init && S->full_streak < 2) dw = fmax(P->delta_w_init, dlast / 3.0);  /* no delta_w = 0 probe: see k_conv */
  if (getenv("PORT_GN_THETA") && S->thinf > atof(getenv("PORT_GN_THETA"))) { gam = 0.0; if (dw == 0.0) dw = getenv("PORT_GN_DW") ? atof(getenv("PORT_GN_DW")) : P->delta_w_init; }
  int ok = 0;
  const int TR = getenv("PORT_TR") ? atoi(getenv("PORT_TR")) : 0;
  if (TR) {
    /* experiment: delta_w as a Levenberg-Marquardt / trust-region parameter driven by the accepted step length */
    const double up = getenv("PORT_TR_UP") ? atof(getenv("PORT_TR_UP")) : 4.0, down = getenv("PORT_TR_DOWN") ? atof(getenv("PORT_TR_DOWN")) : 1.0 / 3.0;
    const double lo = getenv("PORT_TR_LO") ? atof(getenv("PORT_TR_LO")) : 0.25;
    if (S->iter == 0) dw = getenv("PORT_TR_INIT") ? atof(getenv("PORT_TR_INIT")) : 0.0;
    else if (S->alpha >= 1.0) dw = (dlast > 1.1 * P->delta_w_init || S->full_streak < 2) ? fmax(dlast * down, (dlast > 0 ? P->delta_w_init : 0.0)) : 0.0;
    else if (S->alpha >= lo) dw = dlast;
    else dw = fmax(dlast, P->delta_w_init) * up;
    for (int attempt = 0; attempt <= 40; ++attempt) {
      ok = forward_sweep(S, dw, 1.0);
      S->nfact++;
      if (ok) break;
      dw = (dw == 0.0) ? fmax(P->delta_w_init, dlast * down) : dw * ((dlast == 0.0 && attempt == 1) ? 100.0 : 8.0);
    }
    backward_sweep(S);
    S->delta_w = dw; S->delta_last = dw; S->gamma = 1.0;
    if (!ok) S->ls_fail = 1;
    return;
  }
  for (int attempt = 0; attempt <= P->max_refactor; ++attempt) {
    ok = forward_sweep(S, dw, gam);
    S->nfact++;
    if (ok) break;
    if (gam != 0.0) {
      const int skip_ladder = (S->gamma == 0.0) && (S->iter % 4 != 0);
      if (dw == 0.0 && !skip_ladder) dw = (dlast == 0.0) ? P->delta_w_init : fmax(P->delta_w_init, dlast / 3.0);
      else if (!skip_ladder) dw *= (dlast == 0.0) ? 100.0 : 8.0;
      if (skip_ladder || dw > P->delta_w_exact_cap) { gam = 0.0; dw = P->delta_w_init; }
    } else {
      dw *= 8.0;
      if (dw > P->delta_w_max) dw = P->delta_w_max;
    }
  }
  backward_sweep(S);
  S->delta_w = dw;
  if (dw > 0.0 && gam != 0.0) S->delta_last = dw;
  if (dw == 0.0) S->delta_last = 0.0;
  S->gamma = gam;
  if (!ok) S->ls_fail = 1;
}

static double g_lc, g_cc;  /* experiment: (lam + alpha dlam)'c and c'c at the last trial point */
static void trial_point(port_solver* S, double alpha, double* phi, double* th) {
  const port_problem* P = &S->P;
  const int n = P->n, m = P->m, T = P->T;
  double f = 0, t1 = 0;
  g_lc = 0; g_cc = 0;
  double xk[MAXP], yk[MAXN], d[MAXN], l;
  for (int t = 0; t < T; ++t) {
    const int np = np_of(P, t);
    for (int i = 0; i < np; ++i) xk[i] = S->z[zoff(P, t) + i] + alpha * S->dz[zoff(P, t) + i];
    if (t < T - 1) {
      for (int i = 0; i < n; ++i) yk[i] = S->z[zoff(P, t + 1) + i] + alpha * S->dz[zoff(P, t + 1) + i];
      P->costval(xk, xk + n, &l);
      P->dynres(xk, xk + n, yk, d);
      for (int i = 0; i < n; ++i) {
        t1 += fabs(d[i]);
        g_lc += (lam_dyn(S, t)[i] + alpha * dlam_dyn(S, t)[i]) * d[i];
        g_cc += d[i] * d[i];
      }
    } else {
      P->costTval(xk, &l);
    }
    f += l;
    if (q_of(P, t)) {
      const double* target = t == 0 ? P->x1 : P->xT;
      for (int i = 0; i < n; ++i) {
        const double ci = xk[i] - target[i];
        t1 += fabs(ci);
        g_lc += (lam_pin(S, t == 0 ? 0 : 1)[i] + alpha * dlam_pin(S, t == 0 ? 0 : 1)[i]) * ci;
        g_cc += ci * ci;
      }
    }
  }
  (void)m;
  *phi = f; *th = t1;
}

static void line_search_al(port_solver* S) {
  /* experiment: augmented-Lagrangian merit M = f + lam'c + rho/2 c'c, Armijo backtracking in (z, lam) jointly */
  double phi, th;
  trial_point(S, 0.0, &phi, &th);
  const double f0 = phi, lc0 = g_lc, cc0 = g_cc;
  /* M'(0) = rp'd + c'dlam - rho c'c ;  rp'd + lam'(Jd) part is S->gphid-like: recompute directly */
  const port_problem* P = &S->P;
  double rpd = 0, cdl = 0;
  for (int t = 0; t < P->T; ++t) {
    stage_t* s = &S->st[t];
    const int np = np_of(P, t);
    for (int i = 0; i < np; ++i) rpd += s->rp[i] * S->dz[zoff(P, t) + i];
    if (t < P->T - 1) for (int i = 0; i < P->n; ++i) cdl += s->d[i] * dlam_dyn(S, t)[i];
    if (q_of(P, t)) for (int i = 0; i < P->n; ++i) cdl += s->c[i] * dlam_pin(S, t == 0 ? 0 : 1)[i];
  }
  /* rp = grad f + J'lam, so d/dalpha [f + lam'c] = rp'd  (c terms: lam'Jd inside rp'd), plus dlam'c */
  double slope0 = rpd + cdl;
  double rho = g_rho;
  const double want = -0.5 * fabs(rpd - cdl);  /* = -1/2 d'(H+dw)d when the KKT rows hold */
  if (cc0 > 1e-300 && slope0 - rho * cc0 > want) rho = fmax(2.0 * rho, 2.0 * (slope0 - want) / cc0);
  if (getenv("PORT_RHO_DECAY") && rho > 1.0) { const double need = cc0 > 1e-300 ? (slope0 - want) / cc0 : 0.0; if (need < 0.25 * rho) rho = fmax(need * 2.0, rho * atof(getenv("PORT_RHO_DECAY"))); }
  g_rho = rho;
  const double slope = slope0 - rho * cc0;
  const double M0 = f0 + lc0 + 0.5 * rho * cc0;
  double alpha = 1.0, chosen = -1.0;
  const int ntr = getenv("PORT_AL_TRIALS") ? atoi(getenv("PORT_AL_TRIALS")) : 20;
  for (int k = 0; k < ntr; ++k) {
    trial_point(S, alpha, &phi, &th);
    const double M = phi + g_lc + 0.5 * rho * g_cc;
    if (M == M && M <= M0 + 1e-4 * alpha * slope + 1e-13 * fabs(M0)) { chosen = alpha; break; }
    alpha *= 0.5;
  }
  if (getenv("PORT_DEBUG_IT") && S->iter >= atoi(getenv("PORT_DEBUG_IT")) && S->iter < atoi(getenv("PORT_DEBUG_IT")) + 6)
    fprintf(stderr, "it %d f0 %.6e lc0 %.3e cc0 %.3e rho %.3e slope %.3e (rpd %.3e cdl %.3e) alpha %.4g\n", S->iter, f0, lc0, cc0, rho, slope, rpd, cdl, chosen);
  if (chosen < 0.0) { chosen = alpha * 2.0; S->ls_fail = 1; } else S->ls_fail = 0;
  S->alpha = chosen;
  S->full_streak = (chosen >= 1.0) ? S->full_streak + 1 : 0;
}

/* constraint residuals at z + alpha dz, laid out like lam (dyn rows, first pin, last pin) */
static void residuals_at(port_solver* S, double alpha, double* c) {
  const port_problem* P = &S->P;
  const int n = P->n, T = P->T;
  double xk[MAXP], yk[MAXN];
  for (int t = 0; t < T; ++t) {
    const int np = np_of(P, t);
    for (int i = 0; i < np; ++i) xk[i] = S->z[zoff(P, t) + i] + alpha * S->dz[zoff(P, t) + i];
    if (t < T - 1) {
      for (int i = 0; i < n; ++i) yk[i] = S->z[zoff(P, t + 1) + i] + alpha * S->dz[zoff(P, t + 1) + i];
      P->dynres(xk, xk + n, yk, c + t * n);
    }
    if (q_of(P, t)) {
      const double* target = t == 0 ? P->x1 : P->xT;
      double* cp = c + (T - 1) * n + (t == 0 ? 0 : 1) * n;
      for (int i = 0; i < n; ++i) cp[i] = xk[i] - target[i];
    }
  }
}

/* re-solve with the stored factors and the constraint right-hand side replaced by csoc (second-order correction) */
static void soc_solve(port_solver* S, const double* csoc) {
  const port_problem* P = &S->P;
  const int n = P->n, T = P->T;
  double py[MAXN];
  memset(py, 0, sizeof(py));
  for (int t = 0; t < T; ++t) {
    stage_t* s = &S->st[t];
    const int np = np_of(P, t), q = q_of(P, t), ny = ny_of(P, t), bd = np + q + ny;
    double y[MAXBD];
    for (int i = 0; i < np; ++i) y[i] = -s->rp[i] - (i < n ? py[i] : 0.0);
    for (int j = 0; j < q; ++j) y[np + j] = -csoc[(T - 1) * n + (t == 0 ? 0 : 1) * n + j];
    for (int k = 0; k < ny; ++k) y[np + q + k] = -csoc[t * n + k];
    for (int i = 1; i < bd; ++i)
      for (int k = 0; k < i; ++k) y[i] -= s->L[i * MAXBD + k] * y[k];
    for (int c = 0; c < ny; ++c) {
      double acc = 0;
      for (int i = 0; i < bd; ++i) acc += s->X[i * MAXN + c] * s->dinv[i] * y[i];
      py[c] = acc;
    }
    for (int i = 0; i < bd; ++i) s->w[i] = y[i];
  }
  const double keep = S->gphid;
  backward_sweep(S);
  S->gphid = keep;
}

static void line_search(port_solver* S) {
  if (getenv("PORT_MERIT_AL")) { line_search_al(S); return; }
  const double G_TH = 1e-5, G_PHI = 1e-8, S_TH = 1.1, S_PHI = 2.3, ETA = 1e-8, DELTA = 1.0;
  double phi[LS_TRIALS], th[LS_TRIALS];
  double alpha = 1.0;
  for (int k = 0; k < LS_TRIALS; ++k) { trial_point(S, alpha, &phi[k], &th[k]); alpha *= 0.5; }
  const double th0 = S->th1, phi0 = S->f, dphi = S->gphid;
  const int nf = S->filter_n < FILTER_CAP ? S->filter_n : FILTER_CAP;
  if (getenv("PORT_DEBUG_IT") && S->iter >= atoi(getenv("PORT_DEBUG_IT")) && S->iter < atoi(getenv("PORT_DEBUG_IT")) + 6) {
    double dn = 0; for (int i = 0; i < S->Nz; ++i) dn = fmax(dn, fabs(S->dz[i]));
    fprintf(stderr, "it %d phi0 %.8e th0 %.3e dphi %.3e dw %.2e |dz|inf %.3e\n", S->iter, phi0, th0, dphi, S->delta_w, dn);
    for (int k = 0; k < LS_TRIALS; ++k) fprintf(stderr, "   k %d phi-phi0 %+.3e th %.3e\n", k, phi[k] - phi0, th[k]);
    for (int i = 0; i < nf; ++i) fprintf(stderr, "   filt %d th %.3e phi-phi0 %+.3e\n", i, S->filt[2*i], S->filt[2*i+1] - phi0);
  }
  double chosen = -1.0;
  int ftype = 0, best = 0;
  const int wd_left = S->watchdog, watchdog = wd_left > 0;  /* rollback-free watchdog: see k_ls_reduce */
  const int max_soc = getenv("PORT_SOC") ? atoi(getenv("PORT_SOC")) : 0;
  if (max_soc > 0 && !watchdog) {
    /* is the full step acceptable? */
    int ok0;
    {
      const double tk = th[0], pk = phi[0];
      ok0 = (tk == tk) && (pk == pk) && tk <= S->theta_max;
      const int sw = dphi < 0.0 && pow(-dphi, S_PHI) > DELTA * pow(th0, S_TH);
      if (ok0) {
        if (sw && th0 <= S->theta_min) ok0 = pk <= phi0 + ETA * dphi + 1e-13 * fabs(phi0);
        else ok0 = (tk <= (1.0 - G_TH) * th0) || (pk <= phi0 - G_PHI * th0);
      }
      if (ok0)
        for (int i = 0; i < nf; ++i) {
          const double tf = S->filt[2 * i], pf = S->filt[2 * i + 1];
          if (!(tk <= (1.0 - G_TH) * tf || pk <= pf - G_PHI * tf)) { ok0 = 0; break; }
        }
    }
    if (!ok0 && th[0] >= th0) {
      double* csoc = (double*)malloc(S->Nc * sizeof(double));
      double* ctr = (double*)malloc(S->Nc * sizeof(double));
      double* dz0 = (double*)malloc(S->Nz * sizeof(double));
      double* dl0 = (double*)malloc(S->Nc * sizeof(double));
      memcpy(dz0, S->dz, S->Nz * sizeof(double)); memcpy(dl0, S->dlam, S->Nc * sizeof(double));
      residuals_at(S, 0.0, csoc);
      residuals_at(S, 1.0, ctr);
      for (int i = 0; i < S->Nc; ++i) csoc[i] += ctr[i];
      double th_prev = th[0];
      int accepted = 0;
      for (int it = 0; it < max_soc; ++it) {
        soc_solve(S, csoc);
        double pk, tk;
        trial_point(S, 1.0, &pk, &tk);
        int ok = (tk == tk) && (pk == pk) && tk <= S->theta_max;
        const int sw = dphi < 0.0 && pow(-dphi, S_PHI) > DELTA * pow(th0, S_TH);
        int ft = 0;
        if (ok) {
          if (sw && th0 <= S->theta_min) { ok = pk <= phi0 + ETA * dphi + 1e-13 * fabs(phi0); ft = ok; }
          else ok = (tk <= (1.0 - G_TH) * th0) || (pk <= phi0 - G_PHI * th0);
        }
        if (ok)
          for (int i = 0; i < nf; ++i) {
            const double tf = S->filt[2 * i], pf = S->filt[2 * i + 1];
            if (!(tk <= (1.0 - G_TH) * tf || pk <= pf - G_PHI * tf)) { ok = 0; break; }
          }
        if (getenv("PORT_DEBUG_IT") && S->iter >= atoi(getenv("PORT_DEBUG_IT")) && S->iter < atoi(getenv("PORT_DEBUG_IT")) + 6)
          fprintf(stderr, "   soc %d: phi-phi0 %+.3e th %.3e ok %d\n", it, pk - phi0, tk, ok);
        if (ok) {
          accepted = 1;
          S->alpha = 1.0; S->ls_fail = 0;
          if (!ft) { const int slot = S->filter_n % FILTER_CAP; S->filt[2 * slot] = (1.0 - G_TH) * th0; S->filt[2 * slot + 1] = phi0 - G_PHI * th0; S->filter_n++; }
          S->full_streak += 1; S->short_streak = 0;
          g_nsoc++;
          break;
        }
        if (!(tk < 0.99 * th_prev)) break;
        th_prev = tk;
        residuals_at(S, 1.0, ctr);
        for (int i = 0; i < S->Nc; ++i) csoc[i] += ctr[i];
      }
      if (!accepted) { memcpy(S->dz, dz0, S->Nz * sizeof(double)); memcpy(S->dlam, dl0, S->Nc * sizeof(double)); }
      free(csoc); free(ctr); free(dz0); free(dl0);
      if (accepted) return;
    }
  }
  alpha = 1.0;
  for (int k = 0; k < LS_TRIALS; ++k) {
    const double tk = th[k], pk = phi[k];
    if (th[k] < th[best] || !(th[best] == th[best])) best = k;
    int ok = (tk == tk) && (pk == pk) && tk <= S->theta_max;
    const int sw = dphi < 0.0 && alpha * pow(-dphi, S_PHI) > DELTA * pow(th0, S_TH);
    if (ok) {
      if (sw && th0 <= S->theta_min) ok = pk <= phi0 + ETA * alpha * dphi + 1e-13 * fabs(phi0);
      else ok = (tk <= (1.0 - G_TH) * th0) || (pk <= phi0 - G_PHI * th0);
    }
    if (watchdog) ok = (tk == tk) && (pk == pk) && tk <= S->theta_max;
    if (ok && !watchdog)
      for (int i = 0; i < nf; ++i) {
        const double tf = S->filt[2 * i], pf = S->filt[2 * i + 1];
        if (!(tk <= (1.0 - G_TH) * tf || pk <= pf - G_PHI * tf)) { ok = 0; break; }
      }
    if (ok) { chosen = alpha; ftype = sw && (pk <= phi0 + ETA * alpha * dphi + 1e-13 * fabs(phi0)); break; }
    alpha *= 0.5;
  }
  int augment;
  if (chosen < 0.0) {
    double ab = 1.0;
    for (int k = 0; k < best; ++k) ab *= 0.5;
    if (th[best] == th[best] && th[best] < th0) chosen = ab; else chosen = alpha * 2.0;
    S->ls_fail = 1; augment = 1;
  } else { S->ls_fail = 0; augment = !ftype && !watchdog; }
  if (watchdog) { S->watchdog = wd_left - 1; S->short_streak = 0; }
  else if (S->P.watchdog_trigger > 0) {
    const int streak = (chosen < 1.0) ? S->short_streak + 1 : 0;
    if (streak >= S->P.watchdog_trigger) { S->watchdog = S->P.watchdog_trials; S->short_streak = 0; }
    else S->short_streak = streak;
  }
  if (augment) {
    const int slot = S->filter_n % FILTER_CAP;
    S->filt[2 * slot] = (1.0 - G_TH) * th0;
    S->filt[2 * slot + 1] = phi0 - G_PHI * th0;
    S->filter_n++;
  }
  S->alpha = chosen;
  S->full_streak = (chosen >= 1.0) ? S->full_streak + 1 : 0;
}

/* one iteration: EVAL -> CONV -> FACTOR_SOLVE -> LINESEARCH -> UPDATE; returns 1 if an iteration was executed */
int port_iterate(port_solver* S) {
  if (S->status != 0) return 0;
  eval_all(S);
  convergence(S);
  if (S->status != 0) return 0;
  factor_solve(S);
  line_search(S);
  for (int i = 0; i < S->Nz; ++i) S->z[i] += S->alpha * S->dz[i];
  for (int i = 0; i < S->Nc; ++i) S->lam[i] += S->alpha * S->dlam[i];
  S->iter++;
  return 1;
}

/* accessors for ctypes */
int port_status(const port_solver* S) { return S->status; }
int port_iterations(const port_solver* S) { return S->iter; }
int port_nfact(const port_solver* S) { return S->nfact; }
int port_num_variables(const port_solver* S) { return S->Nz; }
int port_num_constraint(const port_solver* S) { return S->Nc; }
double port_objective(const port_solver* S) { return S->f; }
double port_constr_viol(const port_solver* S) { return S->thinf; }
double port_dual_inf(const port_solver* S) { return S->dinf; }
double port_alpha(const port_solver* S) { return S->alpha; }
double port_delta_w(const port_solver* S) { return S->delta_w; }
const double* port_z(const port_solver* S) { return S->z; }
/* multipliers in the reference order: dynamics rows, then stage rows (first pin, last pin) */
const double* port_lam(const port_solver* S) { return S->lam; }

/* ---- model registry ---- */
#define DECL(name)                                                                                             \
  void name##_cost(const double*, const double*, double*, double*, double*);                                   \
  void name##_costT(const double*, double*, double*, double*);                                                 \
  void name##_dyn(const double*, const double*, const double*, const double*, double*, double*, double*, double*, \
                  double*, double*);                                                                           \
  void name##_dynres(const double*, const double*, const double*, double*);                                    \
  void name##_costval(const double*, const double*, double*);                                                  \
  void name##_costTval(const double*, double*);
DECL(acrobot)
DECL(pendulum)

port_solver* port_create_named(const char* model, int T, const double* x1, const double* xT, int max_iter) {
  port_problem P;
  memset(&P, 0, sizeof(P));
  if (!strcmp(model, "acrobot")) {
    P.n = 4; P.m = 1;
    P.cost = acrobot_cost; P.costT = acrobot_costT; P.dyn = acrobot_dyn; P.dynres = acrobot_dynres;
    P.costval = acrobot_costval; P.costTval = acrobot_costTval;
  } else if (!strcmp(model, "pendulum")) {
    P.n = 2; P.m = 1;
    P.cost = pendulum_cost; P.costT = pendulum_costT; P.dyn = pendulum_dyn; P.dynres = pendulum_dynres;
    P.costval = pendulum_costval; P.costTval = pendulum_costTval;
  } else {
    return NULL;
  }
  P.T = T;
  memcpy(P.x1, x1, P.n * sizeof(double));
  memcpy(P.xT, xT, P.n * sizeof(double));
  P.tol = 1e-6; P.s_max = 100.0; P.dual_inf_tol = 1.0; P.constr_viol_tol = 1e-3;
  P.delta_c = 1e-8; P.delta_w_init = 1e-4; P.delta_w_max = 1e20; P.delta_w_exact_cap = 1.0; P.piv_tol = 1e-9;
  P.max_iter = max_iter; P.max_refactor = 12; P.watchdog_trigger = 10; P.watchdog_trials = 3;
  if (getenv("DTO_WATCHDOG")) sscanf(getenv("DTO_WATCHDOG"), "%d,%d", &P.watchdog_trigger, &P.watchdog_trials);
  return port_create(&P);
}
