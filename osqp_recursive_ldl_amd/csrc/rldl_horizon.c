/*
 * rldl_horizon.c -- variable-horizon MPC: changing N between solves.
 *
 * What the reference does (src/recursive_ldl.c): osqp_setup_recursive(..., Nmax, N, ...) :2018-2230 sizes everything for
 * Nmax and sets the problem up at N; osqp_update_recursive(work, data, N') :1973-2016 then (1) changes n and m
 * (update_problem_size :204-211), (2) continues the stage recursion from the last stage both horizons share
 * (LDL_update_from_pivot :946-1110, iter_start = min(N, N') - 1) and (3) rewrites P and A from that stage on with the
 * nominal blocks Qi / Ai / Aij and the terminal blocks QN / AN (update_AP_matrices :1675-1778).  The vectors of the
 * workspace (q, l, u, x, z, y, rho_vec) are left as they are: their first n' / m' entries are simply what the next solve
 * reads, and the caller is expected to refresh q, l, u.  The bordered X/Z/Y "combine" variant (:2319-2856) calls
 * functions whose bodies are empty upstream (osqp_update_X_horizon :2757) and is not restated.
 *
 * What this build does: the permuted KKT matrices of two horizons share their leading stage blocks
 * (Q0, C0, ..., C_{p-1} with p = min(N, N')), and so do their factors.  Every horizon that has been visited keeps its
 * own device-resident workspace (patterns, plan, factor, iterates: a few hundred MB per horizon at batch 4096, i.e. the
 * whole range 1..Nmax fits in a corner of the 288 GB of HBM), so a horizon change costs no allocation and no symbolic
 * analysis after the first visit.  Moving from N to N':
 *   values   per-instance P / A values of stages < p travel with the instance (k_horizon_values); stages >= p are written
 *            again with the nominal blocks, as update_AP_matrices does;
 *   q, l, u  are taken from the call (the reference leaves refreshing them to the caller);
 *   rho      every instance keeps its current rho; rho_vec is rebuilt from the new bounds (set_rho_vec, auxil.c:79-101);
 *   factor   the columns of the shared blocks are copied from the old factor and the stage recursion restarts at block
 *            Q_p (k_horizon_adopt + k_stage_factor_r with a per-instance first block) -- one stage later than the
 *            reference's iter_start, because C_{p-1} and its coupling to x_p are the same in both matrices.  Instances
 *            whose rho_vec differs on the shared rows (a constraint changed its type), equilibrated problems
 *            (scaling > 0 moves every entry of the scaled KKT) and patterns without dense stage blocks are
 *            factorised from the first block instead;
 *   iterates x keeps its first min(n, n') entries, y the rows of the shared row blocks, the terminal multipliers move to
 *            the new terminal rows, new entries start at zero, z = A x (osqp_warm_start, osqp.c:907-950).  This is a
 *            design decision: the reference reads whatever the Nmax-sized vectors held.
 */
#include <hip/hip_runtime_api.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/osqp_rldl_hip.h"
#include "rldl_device.h"
#include "rldl_internal.h"
#include "rldl_symbolic.h"

#define HIP_OK(e) ((e) == hipSuccess)

struct osqp_horizon {
  c_int batch, Nmax, N;
  rldl_stage_dims dims;          /* nx, nu, ny, nt; N = current horizon */
  csc *blk[7];                   /* owned copies of Q0, Qi, QN, A0, Ai, Aij, AN (nominal values) */
  OSQPBatchSettings st;
  void *stream;
  osqp_batch **ws;               /* [Nmax + 1] workspace of every horizon visited so far */
  double **nomP, **nomA;         /* [Nmax + 1] device copies of the nominal P / A values of that horizon (one row) */
  signed char *prefix_ok;        /* [(Nmax + 1)^2] do the L patterns of two horizons agree on the shared blocks? -1 = not looked at */
  double *xs, *ys;               /* [batch][n(Nmax)], [batch][m(Nmax)] staging for the mapped iterates */
  int *b0v, *n_reused;           /* [batch] first block of the restart per instance; [RLDL_NACT_SLOTS] counters */
  int *h_reused;                 /* pinned */
  c_int last_pivot, last_reused, last_created;
};

static csc *csc_clone(const csc *M) {
  csc *C;
  c_int nz;
  if (!M || !M->p) return 0;
  nz = M->p[M->n];
  C = (csc *)calloc(1, sizeof(csc));
  if (!C) return 0;
  C->m = M->m; C->n = M->n; C->nzmax = nz > 0 ? nz : 1; C->nz = -1;
  C->p = (c_int *)malloc(sizeof(c_int) * (size_t)(M->n + 1));
  C->i = (c_int *)malloc(sizeof(c_int) * (size_t)C->nzmax);
  C->x = (c_float *)malloc(sizeof(c_float) * (size_t)C->nzmax);
  if (!C->p || !C->i || !C->x) { rldl_csc_free(C); return 0; }
  memcpy(C->p, M->p, sizeof(c_int) * (size_t)(M->n + 1));
  if (nz > 0) { memcpy(C->i, M->i, sizeof(c_int) * (size_t)nz); memcpy(C->x, M->x, sizeof(c_float) * (size_t)nz); }
  return C;
}

void osqp_horizon_free(osqp_horizon *h) {
  c_int k;
  if (!h) return;
  if (h->ws)
    for (k = 0; k <= h->Nmax; k++) osqp_batch_cleanup(h->ws[k]);
  if (h->nomP)
    for (k = 0; k <= h->Nmax; k++) if (h->nomP[k]) (void)hipFree(h->nomP[k]);
  if (h->nomA)
    for (k = 0; k <= h->Nmax; k++) if (h->nomA[k]) (void)hipFree(h->nomA[k]);
  free(h->ws); free(h->nomP); free(h->nomA); free(h->prefix_ok);
  for (k = 0; k < 7; k++) rldl_csc_free(h->blk[k]);
  if (h->xs) (void)hipFree(h->xs);
  if (h->ys) (void)hipFree(h->ys);
  if (h->b0v) (void)hipFree(h->b0v);
  if (h->n_reused) (void)hipFree(h->n_reused);
  if (h->h_reused) (void)hipHostFree(h->h_reused);
  free(h);
}

static c_int create_workspace(osqp_horizon *h, c_int N, const c_float *d_q, const c_float *d_l, const c_float *d_u) {
  rldl_stage_dims d = h->dims;
  csc *P = 0, *A = 0;
  c_int rc;
  d.N = N;
  rc = osqp_batch_setup_recursive(&h->ws[N], h->batch, &d, h->blk[0], h->blk[1], h->blk[2], h->blk[3], h->blk[4], h->blk[5], h->blk[6],
                                  d_q, d_l, d_u, &h->st, &P, &A, h->stream);
  if (rc) return rc;
  /* the nominal value rows stay on the device: update_AP_matrices (:1675-1778) writes them again from the pivot stage on */
  if (!HIP_OK(hipMalloc((void **)&h->nomP[N], sizeof(double) * (size_t)P->p[P->n] + 8)) ||
      !HIP_OK(hipMalloc((void **)&h->nomA[N], sizeof(double) * (size_t)A->p[A->n] + 8)) ||
      !HIP_OK(hipMemcpy(h->nomP[N], P->x, sizeof(double) * (size_t)P->p[P->n], hipMemcpyHostToDevice)) ||
      !HIP_OK(hipMemcpy(h->nomA[N], A->x, sizeof(double) * (size_t)A->p[A->n], hipMemcpyHostToDevice)))
    rc = RLDL_MEM_ALLOC_ERROR;
  rldl_csc_free(P); rldl_csc_free(A);
  if (rc) { osqp_batch_cleanup(h->ws[N]); h->ws[N] = 0; }
  return rc;
}

c_int osqp_horizon_setup(osqp_horizon **hp, c_int batch, const rldl_stage_dims *dims, c_int Nmax, const csc *Q0, const csc *Qi,
                         const csc *QN, const csc *A0, const csc *Ai, const csc *Aij, const csc *AN, const c_float *d_q,
                         const c_float *d_l, const c_float *d_u, const OSQPBatchSettings *settings, void *stream) {
  const csc *src[7];
  osqp_horizon *h;
  c_int k, rc;
  size_t nmax, mmax;
  if (hp) *hp = 0;
  if (!hp || !dims || !settings || batch <= 0) return 1;
  if (dims->N < 1 || dims->N > Nmax) return 1;                    /* :1978-1989 */
  if (!rldl_device_available()) return RLDL_NO_DEVICE_ERROR;
  h = (osqp_horizon *)calloc(1, sizeof(osqp_horizon));
  if (!h) return RLDL_MEM_ALLOC_ERROR;
  h->batch = batch; h->Nmax = Nmax; h->N = dims->N; h->dims = *dims; h->st = *settings; h->stream = stream;
  h->last_pivot = -1;
  src[0] = Q0; src[1] = Qi; src[2] = QN; src[3] = A0; src[4] = Ai; src[5] = Aij; src[6] = AN;
  for (k = 0; k < 7; k++) {
    h->blk[k] = csc_clone(src[k]);
    if (!h->blk[k]) { osqp_horizon_free(h); return src[k] ? RLDL_MEM_ALLOC_ERROR : 1; }
  }
  h->ws = (osqp_batch **)calloc((size_t)Nmax + 1, sizeof(osqp_batch *));
  h->nomP = (double **)calloc((size_t)Nmax + 1, sizeof(double *)); h->nomA = (double **)calloc((size_t)Nmax + 1, sizeof(double *));
  h->prefix_ok = (signed char *)malloc((size_t)(Nmax + 1) * (size_t)(Nmax + 1));
  if (!h->ws || !h->nomP || !h->nomA || !h->prefix_ok) { osqp_horizon_free(h); return RLDL_MEM_ALLOC_ERROR; }
  memset(h->prefix_ok, -1, (size_t)(Nmax + 1) * (size_t)(Nmax + 1));
  nmax = (size_t)(Nmax * (dims->nx + dims->nu)); mmax = (size_t)(Nmax * (dims->nx + dims->ny) + dims->nt);
  if (!HIP_OK(hipMalloc((void **)&h->xs, sizeof(double) * (size_t)batch * nmax + 8)) ||
      !HIP_OK(hipMalloc((void **)&h->ys, sizeof(double) * (size_t)batch * mmax + 8)) ||
      !HIP_OK(hipMalloc((void **)&h->b0v, sizeof(int) * (size_t)batch)) || !HIP_OK(hipMalloc((void **)&h->n_reused, sizeof(int) * RLDL_NACT_SLOTS)) ||
      !HIP_OK(hipHostMalloc((void **)&h->h_reused, sizeof(int) * RLDL_NACT_SLOTS, hipHostMallocDefault))) {
    osqp_horizon_free(h);
    return RLDL_MEM_ALLOC_ERROR;
  }
  rc = create_workspace(h, h->N, d_q, d_l, d_u);
  if (rc) { osqp_horizon_free(h); return rc; }
  *hp = h;
  return 0;
}

osqp_batch *osqp_horizon_workspace(osqp_horizon *h) { return h ? h->ws[h->N] : 0; }
c_int osqp_horizon_N(const osqp_horizon *h) { return h ? h->N : -1; }

c_int osqp_horizon_last_update(const osqp_horizon *h, c_int *pivot_stage, c_int *instances_reused, c_int *workspace_created) {
  if (!h) return 1;
  if (pivot_stage) *pivot_stage = h->last_pivot;
  if (instances_reused) *instances_reused = h->last_reused;
  if (workspace_created) *workspace_created = h->last_created;
  return 0;
}

/* Do the factors of horizons a and b have the same pattern in the columns before Q_p, p = min(a, b)?  (They do whenever
 * the coupling block reaches only the state part of a stage; a coupling into the inputs of the next stage makes the
 * last shared columns longer in the longer horizon.)  Looked at once per pair. */
static int prefix_agrees(osqp_horizon *h, c_int a, c_int b, int c0) {
  signed char *f = &h->prefix_ok[(size_t)a * (size_t)(h->Nmax + 1) + (size_t)b];
  if (*f < 0) {
    const rldl_symbolic *sa = h->ws[a]->ls->sym, *sb = h->ws[b]->ls->sym;
    int ok = c0 <= sa->N && c0 <= sb->N && sa->Lp[c0] == sb->Lp[c0];
    if (ok) ok = !memcmp(sa->Lp, sb->Lp, sizeof(int) * (size_t)(c0 + 1)) && !memcmp(sa->Li, sb->Li, sizeof(int) * (size_t)sa->Lp[c0]);
    *f = (signed char)ok;
    h->prefix_ok[(size_t)b * (size_t)(h->Nmax + 1) + (size_t)a] = (signed char)ok;
  }
  return *f;
}

c_int osqp_horizon_update(osqp_horizon *h, c_int Nnew, const c_float *d_q, const c_float *d_l, const c_float *d_u) {
  osqp_batch *o, *w;
  const rldl_stage_dims *d;
  hipStream_t st;
  c_int p, n_keep, m_keep, col_keep, rc;
  int c0, adopt;
  size_t B;
  if (!h) return 7;
  if (Nnew > h->Nmax || Nnew < 1) return -1;                      /* :1978-1989 */
  if (Nnew == h->N) return 0;                                     /* :1982-1985 */
  if (!d_q || !d_l || !d_u) return 1;
  d = &h->dims; st = (hipStream_t)h->stream; B = (size_t)h->batch;
  o = h->ws[h->N];
  if (o->loop_pending && osqp_batch_wait(o)) return 1;
  h->last_created = 0;
  if (!h->ws[Nnew]) {
    rc = create_workspace(h, Nnew, d_q, d_l, d_u);
    if (rc) return rc;
    h->last_created = 1;
  }
  w = h->ws[Nnew];
  if (osqp_batch_update_settings(w, &o->st)) return 1;            /* settings changed on the current workspace travel along */
  p = h->N < Nnew ? h->N : Nnew;                                   /* first stage that differs */
  n_keep = p * (d->nx + d->nu); m_keep = p * (d->nx + d->ny);
  col_keep = d->nu + (p - 1) * (d->nx + d->nu);                    /* columns of P and A before stage p */
  c0 = (int)(d->nu + (d->nx + d->ny) + (p - 1) * (2 * d->nx + d->nu + d->ny));   /* first permuted index of block Q_p */
  h->last_pivot = p; h->last_reused = 0;

  if (!HIP_OK(hipMemcpyAsync(w->W.rho_cur, o->W.rho_cur, sizeof(double) * B, hipMemcpyDeviceToDevice, st))) return 1;
  /* the problem data of the new horizon, unscaled: nominal blocks, the instance's own values on the shared stages, q / l / u of the call */
  if (rldl_launch_bcast_rows((int)h->batch, (int)w->nnzP, w->Px, h->nomP[Nnew], w->stream) ||
      rldl_launch_bcast_rows((int)h->batch, (int)w->nnzA, w->Ax, h->nomA[Nnew], w->stream))
    return 1;
  if (!HIP_OK(hipMemcpyAsync(w->q, d_q, sizeof(double) * B * (size_t)w->n, hipMemcpyDeviceToDevice, st)) ||
      !HIP_OK(hipMemcpyAsync(w->l, d_l, sizeof(double) * B * (size_t)w->m, hipMemcpyDeviceToDevice, st)) ||
      !HIP_OK(hipMemcpyAsync(w->u, d_u, sizeof(double) * B * (size_t)w->m, hipMemcpyDeviceToDevice, st)))
    return 1;
  if (rldl_launch_horizon_values(&o->ls->dsym, &o->W, o->Px, o->Ax, (int)col_keep, w->Px, (int)w->nnzP, w->Ax, (int)w->nnzA, w->stream))
    return 1;
  if (w->st.scaling && rldl_launch_scale_data(&w->ls->dsym, &w->W, w->Px, w->Ax, w->q, w->l, w->u, (int)w->st.scaling, w->stream)) return 1;
  if (rldl_launch_set_rho_vec(&w->ls->dsym, &w->W, 1, w->stream)) return 1;
  if (rldl_launch_kkt_assemble(&w->ls->dsym, &w->ls->num, w->Px, w->Ax, w->W.rho_vec, 0, 0, w->stream)) return 1;
  adopt = !w->st.scaling && w->ls->dsym.stage.nb > 0 && o->ls->dsym.stage.nb > 0 && !getenv("RLDL_NO_STAGE_FACTOR") &&
          !getenv("RLDL_HORIZON_FULL") && prefix_agrees(h, h->N, Nnew, c0);
  if (adopt) {
    (void)hipMemsetAsync(h->n_reused, 0, sizeof(int) * RLDL_NACT_SLOTS, st);
    /* the tiles of the product tri-solve travel with the adopted columns when both handles lay them out alike up to the pivot block */
    const int bp = (int)(2 * p);
    const int ti_prefix = w->ls->dsym.stage.pv_ok && o->ls->dsym.stage.pv_ok && w->ls->pv_tiD && o->ls->pv_tiD &&
                          bp <= w->ls->dsym.stage.nb && bp <= o->ls->dsym.stage.nb && w->ls->pv_tiD[bp] == o->ls->pv_tiD[bp] ? w->ls->pv_tiD[bp] : 0;
    if (rldl_launch_horizon_adopt(&o->ls->dsym, &o->ls->num, &w->ls->dsym, &w->ls->num, o->W.rho_vec, w->W.rho_vec, (int)m_keep, c0,
                                  bp, ti_prefix, h->b0v, h->n_reused, w->stream))
      return 1;
    if (rldl_launch_stage_factor_each(&w->ls->dsym, &w->ls->num, h->b0v, ti_prefix > 0, w->stream)) return 1;
    if (!HIP_OK(hipMemcpyAsync(h->h_reused, h->n_reused, sizeof(int) * RLDL_NACT_SLOTS, hipMemcpyDeviceToHost, st))) return 1;
  } else if (rldl_launch_factor(&w->ls->dsym, &w->ls->num, 0, w->stream)) return 1;
  if (rldl_batch_check_status(w->ls)) return RLDL_NONCVX_ERROR;   /* synchronises the stream */
  if (adopt) { int k; for (k = 0; k < RLDL_NACT_SLOTS; k++) h->last_reused += h->h_reused[k]; }
  (void)hipMemsetAsync(w->W.refactor, 0, sizeof(int) * B, st);

  if (o->st.warm_start) {
    if (rldl_launch_horizon_state(&o->ls->dsym, &o->W, (int)w->n, (int)w->m, (int)n_keep, (int)m_keep, (int)d->nt, h->xs, h->ys, w->stream))
      return 1;
    if (osqp_batch_warm_start(w, h->xs, h->ys)) return 1;
  }
  osqp_batch_reset_info(w);
  h->N = Nnew; h->dims.N = Nnew;
  return 0;
}
