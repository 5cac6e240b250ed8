/*
 * oracle/admm_oracle.c -- CPU ORACLE (test infrastructure only; see osqp_oracle.h).
 *
 * Restates the ADMM driver of OSQP v0.6.0 as shipped in the reference, single instance:
 *   src/osqp.c   : osqp_set_default_settings :24-71, osqp_setup :76-283, osqp_solve :288-641,
 *                  osqp_update_lin_cost :752-790, osqp_update_bounds :792-841, osqp_warm_start :929-948,
 *                  osqp_update_P_A :1158-1266, osqp_update_rho :1268-1319
 *   src/auxil.c  : compute_rho_estimate :13-55, adapt_rho :57-77, set_rho_vec :79-101,
 *                  update_rho_vec :103-145, compute_rhs :164-178, update_xz_tilde :180-186,
 *                  update_x :188-201, update_z :203-215, update_y :217-228, compute_obj_val :230-241,
 *                  compute_pri_res :243-257, compute_pri_tol :259-288, compute_dua_res :290-321,
 *                  compute_dua_tol :323-362, is_primal_infeasible :364-424, is_dual_infeasible :426-515,
 *                  store_solution :527-565, update_info :567-626, check_termination :684-789
 *   src/proj.c   : project :4-14
 *   src/scaling.c: scale_data :44-156, unscale_data :160-173, unscale_solution :175-192
 *   src/lin_alg.c: mat_vec :241-271, mat_tpose_vec :273-322, norms :19-43, :325-382, quad_form :387-413
 * Deliberate, documented divergences:
 *   - adaptive_rho with adaptive_rho_interval == 0 uses the reference's PROFILING-off rule
 *     (osqp.c:266-279): 4*check_termination, or 100 when check_termination == 0.  The shipped
 *     default picks the interval from wall-clock time (osqp.c:459-485), which is not reproducible.
 *   (polish IS restated: src/polish.c :19-103, :212-350 -> `polish` below.)
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>
#include <time.h>
#include "osqp_oracle.h"

#define RHO_MIN 1e-06
#define RHO_MAX 1e06
#define RHO_EQ_OVER_RHO_INEQ 1e03
#define RHO_TOL 1e-04
#define MIN_SCALING 1e-04
#define MAX_SCALING 1e+04

static orc_float dmax(orc_float a, orc_float b) { return a > b ? a : b; }
static orc_float dmin(orc_float a, orc_float b) { return a < b ? a : b; }
static orc_float *dvec(orc_int n) { return (orc_float *)calloc((size_t)(n > 0 ? n : 1), sizeof(orc_float)); }

static orc_float norm_inf(const orc_float *v, orc_int n) {
  orc_int i; orc_float m = 0., a;
  for (i = 0; i < n; i++) { a = fabs(v[i]); if (a > m) m = a; }
  return m;
}
static orc_float scaled_norm_inf(const orc_float *S, const orc_float *v, orc_int n) {
  orc_int i; orc_float m = 0., a;
  for (i = 0; i < n; i++) { a = fabs(S[i] * v[i]); if (a > m) m = a; }
  return m;
}
/* plus_eq: 0 y = A x, 1 y += A x, -1 y -= A x (lin_alg.c:241-271) */
static void mat_vec(const orc_csc *A, const orc_float *x, orc_float *y, int plus_eq) {
  orc_int i, j;
  if (!plus_eq) for (i = 0; i < A->m; i++) y[i] = 0;
  if (plus_eq == -1) {
    for (j = 0; j < A->n; j++)
      for (i = A->p[j]; i < A->p[j + 1]; i++) y[A->i[i]] -= A->x[i] * x[j];
    return;
  }
  for (j = 0; j < A->n; j++)
    for (i = A->p[j]; i < A->p[j + 1]; i++) y[A->i[i]] += A->x[i] * x[j];
}
static void mat_tpose_vec(const orc_csc *A, const orc_float *x, orc_float *y, int plus_eq, int skip_diag) {
  orc_int i, j, k;
  if (!plus_eq) for (i = 0; i < A->n; i++) y[i] = 0;
  for (j = 0; j < A->n; j++)
    for (k = A->p[j]; k < A->p[j + 1]; k++) {
      i = A->i[k];
      if (skip_diag && i == j) continue;
      if (plus_eq == -1) y[j] -= A->x[k] * x[i];
      else y[j] += A->x[k] * x[i];
    }
}
static orc_float quad_form(const orc_csc *P, const orc_float *x) {
  orc_float q = 0.; orc_int i, j, p;
  for (j = 0; j < P->n; j++)
    for (p = P->p[j]; p < P->p[j + 1]; p++) {
      i = P->i[p];
      if (i == j) q += .5 * P->x[p] * x[i] * x[i];
      else if (i < j) q += P->x[p] * x[i] * x[j];
    }
  return q;
}

void orc_set_default_settings(orc_settings *s) { /* osqp.c:24-71, constants.h:59-115 */
  s->rho = 0.1; s->sigma = 1e-6; s->alpha = 1.6; s->eps_abs = 1e-3; s->eps_rel = 1e-3;
  s->eps_prim_inf = 1e-4; s->eps_dual_inf = 1e-4; s->max_iter = 4000; s->check_termination = 25;
  s->warm_start = 1; s->scaling = 10; s->scaled_termination = 0; s->adaptive_rho = 1;
  s->adaptive_rho_interval = 0; s->adaptive_rho_tolerance = 5;
  s->polish = 0; s->polish_refine_iter = 3; s->delta = 1e-6;
}

/* ---------------------------------------------------------------- scaling ---- */
static void limit_scaling(orc_float *D, orc_int n) {
  orc_int i;
  for (i = 0; i < n; i++) {
    D[i] = D[i] < MIN_SCALING ? 1.0 : D[i];
    D[i] = D[i] > MAX_SCALING ? MAX_SCALING : D[i];
  }
}
static void inf_norm_cols_sym_triu(const orc_csc *M, orc_float *E) {
  orc_int i, j, p; orc_float a;
  for (j = 0; j < M->n; j++) E[j] = 0.;
  for (j = 0; j < M->n; j++)
    for (p = M->p[j]; p < M->p[j + 1]; p++) {
      i = M->i[p]; a = fabs(M->x[p]);
      E[j] = dmax(a, E[j]);
      if (i != j) E[i] = dmax(a, E[i]);
    }
}
static void premult_diag(orc_csc *A, const orc_float *d) {
  orc_int j, i;
  for (j = 0; j < A->n; j++) for (i = A->p[j]; i < A->p[j + 1]; i++) A->x[i] *= d[A->i[i]];
}
static void postmult_diag(orc_csc *A, const orc_float *d) {
  orc_int j, i;
  for (j = 0; j < A->n; j++) for (i = A->p[j]; i < A->p[j + 1]; i++) A->x[i] *= d[j];
}

static void scale_data(orc_workspace *w) { /* scaling.c:44-156 */
  orc_int i, k, p, n = w->n, m = w->m;
  orc_float c_temp, nq;
  w->c = 1.0;
  for (i = 0; i < n; i++) { w->D[i] = 1.; w->Dinv[i] = 1.; }
  for (i = 0; i < m; i++) { w->E[i] = 1.; w->Einv[i] = 1.; }
  for (k = 0; k < w->settings.scaling; k++) {
    /* column norms of [P A'; A 0] */
    inf_norm_cols_sym_triu(w->P, w->D_temp);
    for (i = 0; i < n; i++) w->D_temp_A[i] = 0.;
    for (i = 0; i < n; i++)
      for (p = w->A->p[i]; p < w->A->p[i + 1]; p++) w->D_temp_A[i] = dmax(fabs(w->A->x[p]), w->D_temp_A[i]);
    for (i = 0; i < n; i++) w->D_temp[i] = dmax(w->D_temp[i], w->D_temp_A[i]);
    for (i = 0; i < m; i++) w->E_temp[i] = 0.;
    for (i = 0; i < n; i++)
      for (p = w->A->p[i]; p < w->A->p[i + 1]; p++)
        w->E_temp[w->A->i[p]] = dmax(fabs(w->A->x[p]), w->E_temp[w->A->i[p]]);
    limit_scaling(w->D_temp, n); limit_scaling(w->E_temp, m);
    for (i = 0; i < n; i++) w->D_temp[i] = 1. / sqrt(w->D_temp[i]);
    for (i = 0; i < m; i++) w->E_temp[i] = 1. / sqrt(w->E_temp[i]);
    premult_diag(w->P, w->D_temp); postmult_diag(w->P, w->D_temp);
    premult_diag(w->A, w->E_temp); postmult_diag(w->A, w->D_temp);
    for (i = 0; i < n; i++) w->q[i] *= w->D_temp[i];
    for (i = 0; i < n; i++) w->D[i] *= w->D_temp[i];
    for (i = 0; i < m; i++) w->E[i] *= w->E_temp[i];
    /* cost normalisation */
    inf_norm_cols_sym_triu(w->P, w->D_temp);
    c_temp = 0.;
    for (i = 0; i < n; i++) c_temp += w->D_temp[i];
    c_temp /= (orc_float)n;
    nq = norm_inf(w->q, n);
    limit_scaling(&nq, 1);
    c_temp = dmax(c_temp, nq);
    limit_scaling(&c_temp, 1);
    c_temp = 1. / c_temp;
    for (p = 0; p < w->P->p[n]; p++) w->P->x[p] *= c_temp;
    for (i = 0; i < n; i++) w->q[i] *= c_temp;
    w->c *= c_temp;
  }
  w->cinv = 1. / w->c;
  for (i = 0; i < n; i++) w->Dinv[i] = 1. / w->D[i];
  for (i = 0; i < m; i++) w->Einv[i] = 1. / w->E[i];
  for (i = 0; i < m; i++) { w->l[i] *= w->E[i]; w->u[i] *= w->E[i]; }
}

static void unscale_data(orc_workspace *w) { /* scaling.c:160-173 */
  orc_int p, i;
  for (p = 0; p < w->P->p[w->n]; p++) w->P->x[p] *= w->cinv;
  premult_diag(w->P, w->Dinv); postmult_diag(w->P, w->Dinv);
  for (i = 0; i < w->n; i++) w->q[i] *= w->cinv * w->Dinv[i];
  premult_diag(w->A, w->Einv); postmult_diag(w->A, w->Dinv);
  for (i = 0; i < w->m; i++) { w->l[i] *= w->Einv[i]; w->u[i] *= w->Einv[i]; }
}

/* ------------------------------------------------------------------ rho ---- */
static void set_rho_vec(orc_workspace *w) { /* auxil.c:79-101 */
  orc_int i;
  w->settings.rho = dmin(dmax(w->settings.rho, RHO_MIN), RHO_MAX);
  for (i = 0; i < w->m; i++) {
    if (w->l[i] < -ORC_INFTY * MIN_SCALING && w->u[i] > ORC_INFTY * MIN_SCALING) {
      w->constr_type[i] = -1; w->rho_vec[i] = RHO_MIN;
    } else if (w->u[i] - w->l[i] < RHO_TOL) {
      w->constr_type[i] = 1; w->rho_vec[i] = RHO_EQ_OVER_RHO_INEQ * w->settings.rho;
    } else {
      w->constr_type[i] = 0; w->rho_vec[i] = w->settings.rho;
    }
    w->rho_inv_vec[i] = 1. / w->rho_vec[i];
  }
}

/* update_rho_vec of whichever backend is in use (work->linsys_solver->update_rho_vec, osqp.c:1310-1318) */
static orc_int ws_update_rho_vec(orc_workspace *w) {
  if (w->ext_update_rho_vec) return w->ext_update_rho_vec(w->ext_self, w->rho_vec);
  return orc_linsys_update_rho_vec(w->linsys, w->rho_vec);
}

static orc_int update_rho_vec(orc_workspace *w) { /* auxil.c:103-145 */
  orc_int i, changed = 0;
  for (i = 0; i < w->m; i++) {
    if (w->l[i] < -ORC_INFTY * MIN_SCALING && w->u[i] > ORC_INFTY * MIN_SCALING) {
      if (w->constr_type[i] != -1) {
        w->constr_type[i] = -1; w->rho_vec[i] = RHO_MIN; w->rho_inv_vec[i] = 1. / RHO_MIN; changed = 1;
      }
    } else if (w->u[i] - w->l[i] < RHO_TOL) {
      if (w->constr_type[i] != 1) {
        w->constr_type[i] = 1; w->rho_vec[i] = RHO_EQ_OVER_RHO_INEQ * w->settings.rho;
        w->rho_inv_vec[i] = 1. / w->rho_vec[i]; changed = 1;
      }
    } else if (w->constr_type[i] != 0) {
      w->constr_type[i] = 0; w->rho_vec[i] = w->settings.rho;
      w->rho_inv_vec[i] = 1. / w->settings.rho; changed = 1;
    }
  }
  if (changed) return ws_update_rho_vec(w);
  return 0;
}

orc_int orc_update_rho(orc_workspace *w, orc_float rho_new) { /* osqp.c:1268-1319 */
  orc_int i;
  if (rho_new <= 0) return 1;
  w->settings.rho = dmin(dmax(rho_new, RHO_MIN), RHO_MAX);
  for (i = 0; i < w->m; i++) {
    if (w->constr_type[i] == 0) { w->rho_vec[i] = w->settings.rho; w->rho_inv_vec[i] = 1. / w->settings.rho; }
    else if (w->constr_type[i] == 1) {
      w->rho_vec[i] = RHO_EQ_OVER_RHO_INEQ * w->settings.rho; w->rho_inv_vec[i] = 1. / w->rho_vec[i];
    }
  }
  return ws_update_rho_vec(w);
}

static orc_float compute_rho_estimate(orc_workspace *w) { /* auxil.c:13-55 */
  orc_int n = w->n, m = w->m;
  orc_float pri = norm_inf(w->z_prev, m), dua = norm_inf(w->x_prev, n), nrm, est;
  nrm = dmax(norm_inf(w->z, m), norm_inf(w->Ax, m));
  pri /= (nrm + 1e-10);
  nrm = dmax(norm_inf(w->q, n), norm_inf(w->Aty, n));
  nrm = dmax(nrm, norm_inf(w->Px, n));
  dua /= (nrm + 1e-10);
  est = w->settings.rho * sqrt(pri / (dua + 1e-10));
  return dmin(dmax(est, RHO_MIN), RHO_MAX);
}

static orc_int adapt_rho(orc_workspace *w) { /* auxil.c:57-77 */
  orc_float rho_new = compute_rho_estimate(w);
  orc_int flag = 0;
  w->info.rho_estimate = rho_new;
  if (rho_new > w->settings.rho * w->settings.adaptive_rho_tolerance ||
      rho_new < w->settings.rho / w->settings.adaptive_rho_tolerance) {
    flag = orc_update_rho(w, rho_new);
    w->info.rho_updates += 1;
  }
  return flag;
}

/* ----------------------------------------------------------------- setup ---- */
orc_int orc_setup(orc_workspace **wp, const orc_csc *P, const orc_float *q, const orc_csc *A,
                  const orc_float *l, const orc_float *u, const orc_settings *settings,
                  const orc_int *perm_in) { /* osqp.c:76-283 */
  orc_workspace *w = (orc_workspace *)calloc(1, sizeof(orc_workspace));
  orc_int n = P->n, m = A->m, flag;
  *wp = w;
  w->n = n; w->m = m; w->settings = *settings;
  w->P = orc_csc_from_arrays(P->m, P->n, P->p, P->i, P->x);
  w->A = orc_csc_from_arrays(A->m, A->n, A->p, A->i, A->x);
  w->q = dvec(n); memcpy(w->q, q, sizeof(orc_float) * (size_t)n);
  w->l = dvec(m); w->u = dvec(m);
  if (m) { memcpy(w->l, l, sizeof(orc_float) * (size_t)m); memcpy(w->u, u, sizeof(orc_float) * (size_t)m); }
  w->rho_vec = dvec(m); w->rho_inv_vec = dvec(m);
  w->constr_type = (orc_int *)calloc((size_t)(m > 0 ? m : 1), sizeof(orc_int));
  w->x = dvec(n); w->z = dvec(m); w->xz_tilde = dvec(n + m); w->x_prev = dvec(n); w->z_prev = dvec(m);
  w->y = dvec(m); w->Ax = dvec(m); w->Px = dvec(n); w->Aty = dvec(n);
  w->delta_y = dvec(m); w->Atdelta_y = dvec(n); w->delta_x = dvec(n); w->Pdelta_x = dvec(n); w->Adelta_x = dvec(m);
  w->D = dvec(n); w->Dinv = dvec(n); w->E = dvec(m); w->Einv = dvec(m);
  w->D_temp = dvec(n); w->D_temp_A = dvec(n); w->E_temp = dvec(m);
  w->sol_x = dvec(n); w->sol_y = dvec(m);
  w->c = 1.; w->cinv = 1.;
  if (perm_in) {
    w->perm = (orc_int *)malloc(sizeof(orc_int) * (size_t)(n + m));
    memcpy(w->perm, perm_in, sizeof(orc_int) * (size_t)(n + m));
  }
  if (w->settings.scaling) scale_data(w);
  set_rho_vec(w);
  flag = orc_linsys_init(&w->linsys, w->P, w->A, w->settings.sigma, w->rho_vec, 0, w->perm);
  if (flag) { orc_cleanup(w); *wp = 0; return flag; }
  w->info.status_val = ORC_UNSOLVED; w->info.rho_updates = 0; w->info.rho_estimate = w->settings.rho;
  if (w->settings.adaptive_rho && !w->settings.adaptive_rho_interval) /* osqp.c:266-279 (PROFILING off rule) */
    w->settings.adaptive_rho_interval = w->settings.check_termination ? 4 * w->settings.check_termination : 100;
  return 0;
}

void orc_cleanup(orc_workspace *w) {
  if (!w) return;
  orc_csc_spfree(w->P); orc_csc_spfree(w->A);
  free(w->q); free(w->l); free(w->u); free(w->rho_vec); free(w->rho_inv_vec); free(w->constr_type);
  free(w->x); free(w->z); free(w->xz_tilde); free(w->x_prev); free(w->z_prev); free(w->y);
  free(w->Ax); free(w->Px); free(w->Aty); free(w->delta_y); free(w->Atdelta_y); free(w->delta_x);
  free(w->Pdelta_x); free(w->Adelta_x); free(w->D); free(w->Dinv); free(w->E); free(w->Einv);
  free(w->D_temp); free(w->D_temp_A); free(w->E_temp); free(w->sol_x); free(w->sol_y); free(w->perm);
  orc_linsys_free(w->linsys);
  free(w);
}

/* ------------------------------------------------------------ ADMM steps ---- */
static void swap(orc_float **a, orc_float **b) { orc_float *t = *b; *b = *a; *a = t; }

static void update_xz_tilde(orc_workspace *w) { /* auxil.c:164-186 */
  orc_int i, n = w->n, m = w->m;
  for (i = 0; i < n; i++) w->xz_tilde[i] = w->settings.sigma * w->x_prev[i] - w->q[i];
  for (i = 0; i < m; i++) w->xz_tilde[i + n] = w->z_prev[i] - w->rho_inv_vec[i] * w->y[i];
  if (w->ext_solve) w->ext_solve(w->ext_self, w->xz_tilde);      /* work->linsys_solver->solve(...), auxil.c:185 */
  else orc_linsys_solve(w->linsys, w->xz_tilde);
}
static void update_x(orc_workspace *w) { /* auxil.c:188-201 */
  orc_int i;
  for (i = 0; i < w->n; i++) w->x[i] = w->settings.alpha * w->xz_tilde[i] + (1.0 - w->settings.alpha) * w->x_prev[i];
  for (i = 0; i < w->n; i++) w->delta_x[i] = w->x[i] - w->x_prev[i];
}
static void update_z(orc_workspace *w) { /* auxil.c:203-215, proj.c:4-14 */
  orc_int i, n = w->n;
  for (i = 0; i < w->m; i++) {
    w->z[i] = w->settings.alpha * w->xz_tilde[i + n] + (1.0 - w->settings.alpha) * w->z_prev[i] +
              w->rho_inv_vec[i] * w->y[i];
    w->z[i] = dmin(dmax(w->z[i], w->l[i]), w->u[i]);
  }
}
static void update_y(orc_workspace *w) { /* auxil.c:217-228 */
  orc_int i, n = w->n;
  for (i = 0; i < w->m; i++) {
    w->delta_y[i] = w->rho_vec[i] * (w->settings.alpha * w->xz_tilde[i + n] +
                                     (1.0 - w->settings.alpha) * w->z_prev[i] - w->z[i]);
    w->y[i] += w->delta_y[i];
  }
}

static orc_float compute_obj_val(orc_workspace *w, const orc_float *x) { /* auxil.c:230-241 */
  orc_int i; orc_float o = quad_form(w->P, x);
  for (i = 0; i < w->n; i++) o += w->q[i] * x[i];
  if (w->settings.scaling) o *= w->cinv;
  return o;
}

static int unscaled_term(orc_workspace *w) { return w->settings.scaling && !w->settings.scaled_termination; }

static orc_float compute_pri_res(orc_workspace *w) { /* auxil.c:243-257; z_prev is scratch */
  orc_int i;
  mat_vec(w->A, w->x, w->Ax, 0);
  for (i = 0; i < w->m; i++) w->z_prev[i] = w->Ax[i] - w->z[i];
  if (unscaled_term(w)) return scaled_norm_inf(w->Einv, w->z_prev, w->m);
  return norm_inf(w->z_prev, w->m);
}
static orc_float compute_dua_res(orc_workspace *w) { /* auxil.c:290-321; x_prev is scratch */
  orc_int i, n = w->n;
  memcpy(w->x_prev, w->q, sizeof(orc_float) * (size_t)n);
  mat_vec(w->P, w->x, w->Px, 0);
  mat_tpose_vec(w->P, w->x, w->Px, 1, 1);
  for (i = 0; i < n; i++) w->x_prev[i] += w->Px[i];
  if (w->m > 0) {
    mat_tpose_vec(w->A, w->y, w->Aty, 0, 0);
    for (i = 0; i < n; i++) w->x_prev[i] += w->Aty[i];
  }
  if (unscaled_term(w)) return w->cinv * scaled_norm_inf(w->Dinv, w->x_prev, n);
  return norm_inf(w->x_prev, n);
}
static orc_float compute_pri_tol(orc_workspace *w, orc_float ea, orc_float er) { /* auxil.c:259-288 */
  orc_float r;
  if (unscaled_term(w)) r = dmax(scaled_norm_inf(w->Einv, w->z, w->m), scaled_norm_inf(w->Einv, w->Ax, w->m));
  else r = dmax(norm_inf(w->z, w->m), norm_inf(w->Ax, w->m));
  return ea + er * r;
}
static orc_float compute_dua_tol(orc_workspace *w, orc_float ea, orc_float er) { /* auxil.c:323-362 */
  orc_float r;
  if (unscaled_term(w)) {
    r = dmax(scaled_norm_inf(w->Dinv, w->q, w->n), scaled_norm_inf(w->Dinv, w->Aty, w->n));
    r = dmax(r, scaled_norm_inf(w->Dinv, w->Px, w->n));
    r *= w->cinv;
  } else {
    r = dmax(norm_inf(w->q, w->n), norm_inf(w->Aty, w->n));
    r = dmax(r, norm_inf(w->Px, w->n));
  }
  return ea + er * r;
}

static orc_int is_primal_infeasible(orc_workspace *w, orc_float eps) { /* auxil.c:364-424 */
  orc_int i, m = w->m; orc_float nd, lhs = 0.;
  for (i = 0; i < m; i++) {
    if (w->u[i] > ORC_INFTY * MIN_SCALING) {
      if (w->l[i] < -ORC_INFTY * MIN_SCALING) w->delta_y[i] = 0.0;
      else w->delta_y[i] = dmin(w->delta_y[i], 0.0);
    } else if (w->l[i] < -ORC_INFTY * MIN_SCALING) w->delta_y[i] = dmax(w->delta_y[i], 0.0);
  }
  if (unscaled_term(w)) {
    for (i = 0; i < m; i++) w->Adelta_x[i] = w->E[i] * w->delta_y[i];
    nd = norm_inf(w->Adelta_x, m);
  } else nd = norm_inf(w->delta_y, m);
  if (nd > eps) {
    for (i = 0; i < m; i++) lhs += w->u[i] * dmax(w->delta_y[i], 0) + w->l[i] * dmin(w->delta_y[i], 0);
    if (lhs < -eps * nd) {
      mat_tpose_vec(w->A, w->delta_y, w->Atdelta_y, 0, 0);
      if (unscaled_term(w)) for (i = 0; i < w->n; i++) w->Atdelta_y[i] *= w->Dinv[i];
      return norm_inf(w->Atdelta_y, w->n) < eps * nd;
    }
  }
  return 0;
}

static orc_int is_dual_infeasible(orc_workspace *w, orc_float eps) { /* auxil.c:426-515 */
  orc_int i, n = w->n, m = w->m; orc_float nd, cs, qd = 0.;
  if (unscaled_term(w)) { nd = scaled_norm_inf(w->D, w->delta_x, n); cs = w->c; }
  else { nd = norm_inf(w->delta_x, n); cs = 1.0; }
  if (nd > eps) {
    for (i = 0; i < n; i++) qd += w->q[i] * w->delta_x[i];
    if (qd < -cs * eps * nd) {
      mat_vec(w->P, w->delta_x, w->Pdelta_x, 0);
      mat_tpose_vec(w->P, w->delta_x, w->Pdelta_x, 1, 1);
      if (unscaled_term(w)) for (i = 0; i < n; i++) w->Pdelta_x[i] *= w->Dinv[i];
      if (norm_inf(w->Pdelta_x, n) < cs * eps * nd) {
        mat_vec(w->A, w->delta_x, w->Adelta_x, 0);
        if (unscaled_term(w)) for (i = 0; i < m; i++) w->Adelta_x[i] *= w->Einv[i];
        for (i = 0; i < m; i++)
          if ((w->u[i] < ORC_INFTY * MIN_SCALING && w->Adelta_x[i] > eps * nd) ||
              (w->l[i] > -ORC_INFTY * MIN_SCALING && w->Adelta_x[i] < -eps * nd)) return 0;
        return 1;
      }
    }
  }
  return 0;
}

static void update_info(orc_workspace *w, orc_int iter) { /* auxil.c:567-626 */
  w->info.iter = iter;
  w->info.pri_res = w->m == 0 ? 0. : compute_pri_res(w);
  w->info.dua_res = compute_dua_res(w);
}

static orc_int check_termination(orc_workspace *w, orc_int approximate) { /* auxil.c:684-789 */
  orc_float ea = w->settings.eps_abs, er = w->settings.eps_rel, epi = w->settings.eps_prim_inf,
            edi = w->settings.eps_dual_inf;
  orc_int prc = 0, drc = 0, pic = 0, dic = 0, i;
  if (w->info.pri_res > ORC_INFTY || w->info.dua_res > ORC_INFTY) {
    w->info.status_val = ORC_NON_CVX; w->info.obj_val = ORC_NAN; return 1;
  }
  if (approximate) { ea *= 10; er *= 10; epi *= 10; edi *= 10; }
  if (w->m == 0) prc = 1;
  else {
    if (w->info.pri_res < compute_pri_tol(w, ea, er)) prc = 1;
    else pic = is_primal_infeasible(w, epi);
  }
  if (w->info.dua_res < compute_dua_tol(w, ea, er)) drc = 1;
  else dic = is_dual_infeasible(w, edi);
  if (prc && drc) { w->info.status_val = approximate ? ORC_SOLVED_INACCURATE : ORC_SOLVED; return 1; }
  if (pic) {
    w->info.status_val = approximate ? ORC_PRIMAL_INFEASIBLE_INACCURATE : ORC_PRIMAL_INFEASIBLE;
    if (unscaled_term(w)) for (i = 0; i < w->m; i++) w->delta_y[i] *= w->E[i];
    w->info.obj_val = ORC_INFTY; return 1;
  }
  if (dic) {
    w->info.status_val = approximate ? ORC_DUAL_INFEASIBLE_INACCURATE : ORC_DUAL_INFEASIBLE;
    if (unscaled_term(w)) for (i = 0; i < w->n; i++) w->delta_x[i] *= w->D[i];
    w->info.obj_val = -ORC_INFTY; return 1;
  }
  return 0;
}

static int has_solution(const orc_info *info) {
  return info->status_val != ORC_PRIMAL_INFEASIBLE && info->status_val != ORC_PRIMAL_INFEASIBLE_INACCURATE &&
         info->status_val != ORC_DUAL_INFEASIBLE && info->status_val != ORC_DUAL_INFEASIBLE_INACCURATE &&
         info->status_val != ORC_NON_CVX;
}

static void store_solution(orc_workspace *w) { /* auxil.c:527-565 */
  orc_int i; orc_float nv;
  if (has_solution(&w->info)) {
    memcpy(w->sol_x, w->x, sizeof(orc_float) * (size_t)w->n);
    if (w->m) memcpy(w->sol_y, w->y, sizeof(orc_float) * (size_t)w->m);
    if (w->settings.scaling) { /* scaling.c:175-192 */
      for (i = 0; i < w->n; i++) w->sol_x[i] *= w->D[i];
      for (i = 0; i < w->m; i++) w->sol_y[i] *= w->E[i] * w->cinv;
    }
  } else {
    for (i = 0; i < w->n; i++) w->sol_x[i] = ORC_NAN;
    for (i = 0; i < w->m; i++) w->sol_y[i] = ORC_NAN;
    if (w->info.status_val == ORC_PRIMAL_INFEASIBLE || w->info.status_val == ORC_PRIMAL_INFEASIBLE_INACCURATE) {
      nv = norm_inf(w->delta_y, w->m);
      for (i = 0; i < w->m; i++) w->delta_y[i] *= 1. / nv;
    }
    if (w->info.status_val == ORC_DUAL_INFEASIBLE || w->info.status_val == ORC_DUAL_INFEASIBLE_INACCURATE) {
      nv = norm_inf(w->delta_x, w->n);
      for (i = 0; i < w->n; i++) w->delta_x[i] *= 1. / nv;
    }
    for (i = 0; i < w->n; i++) w->x[i] = 0.;
    for (i = 0; i < w->m; i++) { w->z[i] = 0.; w->y[i] = 0.; }
  }
}

/* ------------------------------------------------------------------ polish ----
 * src/polish.c: guess the active constraints from (z, y), solve the equality-constrained QP on them through the
 * backend's polish = 1 mode (delta-regularised reduced KKT, raw solution), refine, accept if the residuals improve. */
static orc_int polish(orc_workspace *w) {
  const orc_int n = w->n, m = w->m;
  orc_int *A_to_Alow = (orc_int *)malloc(sizeof(orc_int) * (size_t)(m + 1)), *A_to_Aupp = (orc_int *)malloc(sizeof(orc_int) * (size_t)(m + 1));
  orc_int *low_to_A = (orc_int *)malloc(sizeof(orc_int) * (size_t)(m + 1)), *upp_to_A = (orc_int *)malloc(sizeof(orc_int) * (size_t)(m + 1));
  orc_int n_low = 0, n_upp = 0, j, ptr, nnz = 0, mred, it, ok, i;
  orc_csc Ared;
  orc_linsys *plsh = 0;
  orc_float *rhs_red, *pol_sol, *rhs, *px, *pz, *py, *sx, *sz, *sy, pol_obj, pol_pri, pol_dua;
  /* form_Ared, polish.c:19-103 */
  for (j = 0; j < m; j++) {
    if (w->z[j] - w->l[j] < -w->y[j]) { low_to_A[n_low] = j; A_to_Alow[j] = n_low++; } else A_to_Alow[j] = -1;
  }
  for (j = 0; j < m; j++) {
    if (w->u[j] - w->z[j] < w->y[j]) { upp_to_A[n_upp] = j; A_to_Aupp[j] = n_upp++; } else A_to_Aupp[j] = -1;
  }
  mred = n_low + n_upp;
  for (j = 0; j < w->A->p[n]; j++)
    if (A_to_Alow[w->A->i[j]] != -1 || A_to_Aupp[w->A->i[j]] != -1) nnz++;
  Ared.m = mred; Ared.n = n; Ared.nzmax = nnz > 0 ? nnz : 1; Ared.nz = -1;
  Ared.p = (orc_int *)calloc((size_t)n + 1, sizeof(orc_int));
  Ared.i = (orc_int *)malloc(sizeof(orc_int) * (size_t)Ared.nzmax);
  Ared.x = (orc_float *)malloc(sizeof(orc_float) * (size_t)Ared.nzmax);
  nnz = 0;
  for (j = 0; j < n; j++) {
    Ared.p[j] = nnz;
    for (ptr = w->A->p[j]; ptr < w->A->p[j + 1]; ptr++) {
      if (A_to_Alow[w->A->i[ptr]] != -1) { Ared.i[nnz] = A_to_Alow[w->A->i[ptr]]; Ared.x[nnz++] = w->A->x[ptr]; }
      else if (A_to_Aupp[w->A->i[ptr]] != -1) { Ared.i[nnz] = A_to_Aupp[w->A->i[ptr]] + n_low; Ared.x[nnz++] = w->A->x[ptr]; }
    }
  }
  Ared.p[n] = nnz;
  /* reduced KKT, polish.c:228-243 */
  if (orc_linsys_init(&plsh, w->P, &Ared, w->settings.delta, 0, 1, 0)) {
    w->info.status_polish = -1;
    free(A_to_Alow); free(A_to_Aupp); free(low_to_A); free(upp_to_A); free(Ared.p); free(Ared.i); free(Ared.x);
    return 1;
  }
  rhs_red = dvec(n + mred); pol_sol = dvec(n + mred); rhs = dvec(n + mred);
  for (j = 0; j < n; j++) rhs_red[j] = -w->q[j];                                   /* form_rhs_red :105-121 */
  for (j = 0; j < n_low; j++) rhs_red[n + j] = w->l[low_to_A[j]];
  for (j = 0; j < n_upp; j++) rhs_red[n + n_low + j] = w->u[upp_to_A[j]];
  memcpy(pol_sol, rhs_red, sizeof(orc_float) * (size_t)(n + mred));
  orc_linsys_solve(plsh, pol_sol);
  for (it = 0; it < w->settings.polish_refine_iter; it++) {                        /* iterative_refinement :134-181 */
    memcpy(rhs, rhs_red, sizeof(orc_float) * (size_t)(n + mred));
    mat_vec(w->P, pol_sol, rhs, -1);
    mat_tpose_vec(w->P, pol_sol, rhs, -1, 1);
    mat_tpose_vec(&Ared, pol_sol + n, rhs, -1, 0);
    mat_vec(&Ared, pol_sol, rhs + n, -1);
    orc_linsys_solve(plsh, rhs);
    for (j = 0; j < n + mred; j++) pol_sol[j] += rhs[j];
  }
  px = dvec(n); pz = dvec(m); py = dvec(m);
  memcpy(px, pol_sol, sizeof(orc_float) * (size_t)n);
  mat_vec(w->A, px, pz, 0);
  for (j = 0; j < m; j++)                                                          /* get_ypol_from_yred :188-210 */
    py[j] = A_to_Alow[j] != -1 ? pol_sol[n + A_to_Alow[j]] : (A_to_Aupp[j] != -1 ? pol_sol[n + n_low + A_to_Aupp[j]] : 0.0);
  for (i = 0; i < m; i++) {                                                        /* project_normalcone, proj.c:16-29 */
    const orc_float t = pz[i] + py[i];
    pz[i] = dmin(dmax(t, w->l[i]), w->u[i]);
    py[i] = t - pz[i];
  }
  /* update_info(work, 0, 1, 1), auxil.c:567-626, on the polished point */
  sx = w->x; sz = w->z; sy = w->y;
  w->x = px; w->z = pz; w->y = py;
  pol_obj = compute_obj_val(w, w->x);
  pol_pri = m == 0 ? 0. : compute_pri_res(w);
  pol_dua = compute_dua_res(w);
  w->x = sx; w->z = sz; w->y = sy;
  ok = (pol_pri < w->info.pri_res && pol_dua < w->info.dua_res) || (pol_pri < w->info.pri_res && w->info.dua_res < 1e-10) ||
       (pol_dua < w->info.dua_res && w->info.pri_res < 1e-10);                      /* polish.c:298-311 */
  if (ok) {
    w->info.obj_val = pol_obj; w->info.pri_res = pol_pri; w->info.dua_res = pol_dua; w->info.status_polish = 1;
    memcpy(w->x, px, sizeof(orc_float) * (size_t)n);
    memcpy(w->z, pz, sizeof(orc_float) * (size_t)m);
    memcpy(w->y, py, sizeof(orc_float) * (size_t)m);
  } else w->info.status_polish = -1;
  orc_linsys_free(plsh);
  free(A_to_Alow); free(A_to_Aupp); free(low_to_A); free(upp_to_A); free(Ared.p); free(Ared.i); free(Ared.x);
  free(rhs_red); free(pol_sol); free(rhs); free(px); free(pz); free(py);
  return 0;
}

orc_int orc_solve(orc_workspace *w) { /* osqp.c:288-641 */
  orc_int iter, can_check = 0, i;
  if (!w->settings.warm_start) {
    for (i = 0; i < w->n; i++) w->x[i] = 0.;
    for (i = 0; i < w->m; i++) { w->z[i] = 0.; w->y[i] = 0.; }
  }
  for (iter = 1; iter <= w->settings.max_iter; iter++) {
    swap(&w->x, &w->x_prev);
    swap(&w->z, &w->z_prev);
    update_xz_tilde(w);
    update_x(w);
    update_z(w);
    update_y(w);
    can_check = w->settings.check_termination && (iter % w->settings.check_termination == 0);
    if (can_check) {
      update_info(w, iter);
      if (check_termination(w, 0)) break;
    }
    if (w->settings.adaptive_rho && w->settings.adaptive_rho_interval &&
        (iter % w->settings.adaptive_rho_interval == 0)) {
      if (!can_check) update_info(w, iter);
      if (adapt_rho(w)) return 1;
    }
  }
  if (!can_check) {
    update_info(w, iter - 1);
    check_termination(w, 0);
  }
  if (has_solution(&w->info)) w->info.obj_val = compute_obj_val(w, w->x);
  if (w->info.status_val == ORC_UNSOLVED)
    if (!check_termination(w, 1)) w->info.status_val = ORC_MAX_ITER_REACHED;
  w->info.rho_estimate = compute_rho_estimate(w);
  w->info.status_polish = 0;
  if (w->settings.polish && w->info.status_val == ORC_SOLVED) polish(w);             /* osqp.c:591-595 */
  store_solution(w);
  return 0;
}

/* ---------------------------------------------------------------- updates ---- */
static void reset_info(orc_info *info) { info->status_val = ORC_UNSOLVED; info->rho_updates = 0; }

orc_int orc_update_lin_cost(orc_workspace *w, const orc_float *q_new) { /* osqp.c:752-790 */
  orc_int i;
  memcpy(w->q, q_new, sizeof(orc_float) * (size_t)w->n);
  if (w->settings.scaling) for (i = 0; i < w->n; i++) w->q[i] *= w->D[i] * w->c;
  reset_info(&w->info);
  return 0;
}

orc_int orc_update_bounds(orc_workspace *w, const orc_float *l_new, const orc_float *u_new) { /* osqp.c:792-841 */
  orc_int i;
  for (i = 0; i < w->m; i++) if (l_new[i] > u_new[i]) return 1;
  memcpy(w->l, l_new, sizeof(orc_float) * (size_t)w->m);
  memcpy(w->u, u_new, sizeof(orc_float) * (size_t)w->m);
  if (w->settings.scaling) for (i = 0; i < w->m; i++) { w->l[i] *= w->E[i]; w->u[i] *= w->E[i]; }
  reset_info(&w->info);
  return update_rho_vec(w);
}

orc_int orc_warm_start(orc_workspace *w, const orc_float *x, const orc_float *y) { /* osqp.c:929-948 */
  orc_int i;
  if (!w->settings.warm_start) w->settings.warm_start = 1;
  memcpy(w->x, x, sizeof(orc_float) * (size_t)w->n);
  if (w->m) memcpy(w->y, y, sizeof(orc_float) * (size_t)w->m);
  if (w->settings.scaling) {
    for (i = 0; i < w->n; i++) w->x[i] *= w->Dinv[i];
    for (i = 0; i < w->m; i++) w->y[i] *= w->Einv[i] * w->c;
  }
  mat_vec(w->A, w->x, w->z, 0);
  return 0;
}

/* Full-vector form of osqp_update_P_A (osqp.c:1158-1266 with Px_new_idx == Ax_new_idx == NULL);
 * pass NULL to leave P (or A) unchanged, which gives osqp_update_P / osqp_update_A. */
orc_int orc_update_P_A(orc_workspace *w, const orc_float *Px_new, const orc_float *Ax_new) {
  orc_int flag;
  if (w->settings.scaling) unscale_data(w);          /* osqp.c:1211-1214 */
  if (Px_new) memcpy(w->P->x, Px_new, sizeof(orc_float) * (size_t)w->P->p[w->n]);
  if (Ax_new) memcpy(w->A->x, Ax_new, sizeof(orc_float) * (size_t)w->A->p[w->n]);
  if (w->settings.scaling) scale_data(w);            /* osqp.c:1241-1244: full re-equilibration */
  flag = orc_linsys_update_matrices(w->linsys, w->P, w->A);
  reset_info(&w->info);
  return flag;
}

orc_float *orc_ws_x(orc_workspace *w) { return w->x; }
orc_float *orc_ws_y(orc_workspace *w) { return w->y; }
orc_float *orc_ws_z(orc_workspace *w) { return w->z; }
orc_float *orc_ws_sol_x(orc_workspace *w) { return w->sol_x; }
orc_float *orc_ws_sol_y(orc_workspace *w) { return w->sol_y; }
orc_info  *orc_ws_info(orc_workspace *w) { return &w->info; }
orc_linsys *orc_ws_linsys(orc_workspace *w) { return w->linsys; }
void orc_use_external_linsys(orc_workspace *w, void *self, orc_int (*solve)(void *, orc_float *),
                             orc_int (*update_rho_vec)(void *, const orc_float *)) {
  w->ext_self = self; w->ext_solve = solve; w->ext_update_rho_vec = update_rho_vec;
}
const orc_csc *orc_ws_P(const orc_workspace *w) { return w->P; }
const orc_csc *orc_ws_A(const orc_workspace *w) { return w->A; }
const orc_float *orc_ws_rho_vec(const orc_workspace *w) { return w->rho_vec; }
orc_float *orc_ws_delta_x(orc_workspace *w) { return w->delta_x; }
orc_float *orc_ws_delta_y(orc_workspace *w) { return w->delta_y; }
orc_float *orc_ws_D(orc_workspace *w) { return w->D; }
orc_float *orc_ws_E(orc_workspace *w) { return w->E; }
orc_float  orc_ws_c(orc_workspace *w) { return w->c; }

/* ------------------------------------------------------------ CPU baseline ---- */
static double now_s(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

double orc_bench_shared_pattern(orc_int count, orc_int n, orc_int m, const orc_int *Pp, const orc_int *Pi,
                                const orc_float *Px_all, const orc_int *Ap, const orc_int *Ai,
                                const orc_float *Ax_all, const orc_float *q_all, const orc_float *l_all,
                                const orc_float *u_all, const orc_settings *settings,
                                const orc_int *perm_in, orc_float *x_out, orc_float *y_out,
                                double *t_factor, double *t_solve) {
  orc_int b, nnzP = Pp[n], nnzA = Ap[n];
  double t0, t1, t2, tf = 0., ts = 0.;
  orc_csc P, A;
  P.m = n; P.n = n; P.nzmax = nnzP; P.nz = -1; P.p = (orc_int *)Pp; P.i = (orc_int *)Pi;
  A.m = m; A.n = n; A.nzmax = nnzA; A.nz = -1; A.p = (orc_int *)Ap; A.i = (orc_int *)Ai;
  for (b = 0; b < count; b++) {
    orc_workspace *w;
    P.x = (orc_float *)(Px_all + b * nnzP);
    A.x = (orc_float *)(Ax_all + b * nnzA);
    t0 = now_s();
    if (orc_setup(&w, &P, q_all + b * n, &A, l_all + b * m, u_all + b * m, settings, perm_in)) return -1.;
    t1 = now_s();
    orc_solve(w);
    t2 = now_s();
    tf += t1 - t0; ts += t2 - t1;
    if (x_out) memcpy(x_out + b * n, w->sol_x, sizeof(orc_float) * (size_t)n);
    if (y_out) memcpy(y_out + b * m, w->sol_y, sizeof(orc_float) * (size_t)m);
    orc_cleanup(w);
  }
  if (t_factor) *t_factor = tf;
  if (t_solve) *t_solve = ts;
  return tf + ts;
}

/* The same per-instance work on `nthreads` host threads (bench.py, "all cores" leg of cpu_baseline): the reference is
 * single-threaded per instance (lin_sys/direct/qdldl/qdldl_interface.c:208-209), so the all-core figure is one instance
 * per task.  Thread t runs instances t, t + nthreads, ... (cyclically over the `ndata` instances given), count_per_thread of
 * them.  Returns the wall-clock seconds from before the first thread starts to after the last one has finished
 * (clock_gettime(CLOCK_MONOTONIC), as osqp_tic / osqp_toc, src/util.c:317-337), or -1. */
typedef struct {
  orc_int t, nthreads, count, ndata, n, m;
  const orc_int *Pp, *Pi, *Ap, *Ai, *perm;
  const orc_float *Px, *Ax, *q, *l, *u;
  const orc_settings *settings;
  int failed;
} orc_mt_job;

static void *orc_mt_worker(void *arg) {
  orc_mt_job *j = (orc_mt_job *)arg;
  orc_int k, nnzP = j->Pp[j->n], nnzA = j->Ap[j->n];
  orc_csc P, A;
  P.m = j->n; P.n = j->n; P.nzmax = nnzP; P.nz = -1; P.p = (orc_int *)j->Pp; P.i = (orc_int *)j->Pi;
  A.m = j->m; A.n = j->n; A.nzmax = nnzA; A.nz = -1; A.p = (orc_int *)j->Ap; A.i = (orc_int *)j->Ai;
  for (k = 0; k < j->count; k++) {
    const orc_int b = (j->t + k * j->nthreads) % j->ndata;
    orc_workspace *w;
    P.x = (orc_float *)(j->Px + b * nnzP);
    A.x = (orc_float *)(j->Ax + b * nnzA);
    if (orc_setup(&w, &P, j->q + b * j->n, &A, j->l + b * j->m, j->u + b * j->m, j->settings, j->perm)) { j->failed = 1; return 0; }
    orc_solve(w);
    orc_cleanup(w);
  }
  return 0;
}

double orc_bench_shared_pattern_mt(orc_int nthreads, orc_int count_per_thread, orc_int ndata, orc_int n, orc_int m,
                                   const orc_int *Pp, const orc_int *Pi, const orc_float *Px_all, const orc_int *Ap,
                                   const orc_int *Ai, const orc_float *Ax_all, const orc_float *q_all, const orc_float *l_all,
                                   const orc_float *u_all, const orc_settings *settings, const orc_int *perm_in) {
  pthread_t *th;
  orc_mt_job *jobs;
  orc_int t, started = 0;
  double t0, t1;
  int bad = 0;
  if (nthreads <= 0 || ndata <= 0 || count_per_thread <= 0) return -1.;
  th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)nthreads);
  jobs = (orc_mt_job *)calloc((size_t)nthreads, sizeof(orc_mt_job));
  if (!th || !jobs) { free(th); free(jobs); return -1.; }
  t0 = now_s();
  for (t = 0; t < nthreads; t++) {
    orc_mt_job *j = jobs + t;
    j->t = t; j->nthreads = nthreads; j->count = count_per_thread; j->ndata = ndata; j->n = n; j->m = m;
    j->Pp = Pp; j->Pi = Pi; j->Ap = Ap; j->Ai = Ai; j->perm = perm_in;
    j->Px = Px_all; j->Ax = Ax_all; j->q = q_all; j->l = l_all; j->u = u_all; j->settings = settings;
    if (pthread_create(th + t, 0, orc_mt_worker, j)) { bad = 1; break; }
    started++;
  }
  for (t = 0; t < started; t++) { pthread_join(th[t], 0); if (jobs[t].failed) bad = 1; }
  t1 = now_s();
  free(th); free(jobs);
  return bad ? -1. : t1 - t0;
}
