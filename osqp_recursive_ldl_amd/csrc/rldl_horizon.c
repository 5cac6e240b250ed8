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
 *
 * SINGLE STORE (round 3; the reference's "combined" X / Z / Y variant, osqp_setup_combine_recursive :2359-2756 and
 * osqp_update_Z_horizon :2761-2856: everything is sized for Nmax once, a horizon change rebuilds only the trailing part Z, its
 * border V^ = V L^-T D^-1 and the Schur block Y^ = Y - V^ D V^T, and changes n and m).  In the stage-interleaved order the
 * border IS the coupling block L(2p, 2p-1) and Y^ the Schur complement the recursion forms at block 2p, so the restart at a
 * block is the Z / V^ / Y^ rebuild.  What the single store adds: ONE workspace at Nmax dimensions serves every horizon.
 * Variables and rows of horizon N are prefixes of those of Nmax (x_N sits where stage N's state sits, the nt terminal rows are
 * the first nt rows of row block N), so a horizon is a set of VALUES on the Nmax patterns: stages < N nominal or the instance's
 * own, stage N with QN on its state part and AN in its first nt rows (the patterns of the store are the unions Qi + QN,
 * Ai + AN), everything behind it zero -- decoupled dummy variables (pivot sigma) and free rows (l = -inf, u = +inf), whose
 * iterates stay exactly 0.  The factorisation, the tile inverses and the tri-solve stop at the last live block
 * (rldl_dev_stage.nb_act; the solve runs a per-horizon step program over the same tile store), so the cost follows N, not
 * Nmax.  A change N -> N' writes the nominal values of stages >= p = min(N, N'), q / l / u, rebuilds rho_vec and restarts the
 * recursion at block 2p of the SAME factor store: no second workspace, no adoption copy.  Used when scaling = 0 and the
 * product tri-solve is available; otherwise the per-horizon workspaces above.
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
  /* single store: one workspace at Nmax dimensions (ws[Nmax]) whatever the horizon */
  int single;
  csc *Pmax, *Amax;              /* host patterns of the store (union blocks) */
  int **mapP, **mapA;            /* [Nmax + 1] device: entry i of horizon N's P / A (rldl_setup_AP_matrices order) -> entry of Pmax / Amax */
  c_int *nnzPh, *nnzAh;          /* [Nmax + 1] entries of horizon N's P / A */
  int **progN, *nstepsN;         /* [Nmax + 1] device step programs of the product tri-solve over the live blocks, their step counts */
  const int *prog0; int nsteps0; /* the store's own program (all blocks), restored before the workspace is freed */
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
  if (h->single && h->ws && h->ws[h->Nmax]) {                   /* the store's own tables go back before it is freed */
    rldl_dev_stage *G = &h->ws[h->Nmax]->ls->dsym.stage;
    G->pv_prog = h->prog0; G->pv_nsteps = h->nsteps0; G->nb_act = 0; G->npos_skip = 0;
  }
  if (h->ws)
    for (k = 0; k <= h->Nmax; k++) osqp_batch_cleanup(h->ws[k]);
  for (k = 0; k <= h->Nmax; k++) {
    if (h->mapP && h->mapP[k]) (void)hipFree(h->mapP[k]);
    if (h->mapA && h->mapA[k]) (void)hipFree(h->mapA[k]);
    if (h->progN && h->progN[k]) (void)hipFree(h->progN[k]);
  }
  free(h->mapP); free(h->mapA); free(h->nnzPh); free(h->nnzAh); free(h->progN); free(h->nstepsN);
  rldl_csc_free(h->Pmax); rldl_csc_free(h->Amax);
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

/* ---------------------------------------------------------------------------------------------------------------------
 * Single store
 * --------------------------------------------------------------------------------------------------------------------- */
/* union of two CSC blocks with the second one placed at the top-left corner of the first (rows / columns [0, small->m) x [0, small->n));
 * values: the big block's, 0.0 where only the small one has an entry */
static csc *csc_union_corner(const csc *big, const csc *small) {
  c_int j, nz = 0, cap = big->p[big->n] + small->p[small->n];
  csc *U = (csc *)calloc(1, sizeof(csc));
  if (!U) return 0;
  U->m = big->m; U->n = big->n; U->nzmax = cap > 0 ? cap : 1; U->nz = -1;
  U->p = (c_int *)calloc((size_t)big->n + 1, sizeof(c_int));
  U->i = (c_int *)malloc(sizeof(c_int) * (size_t)U->nzmax);
  U->x = (c_float *)malloc(sizeof(c_float) * (size_t)U->nzmax);
  if (!U->p || !U->i || !U->x) { rldl_csc_free(U); return 0; }
  for (j = 0; j < big->n; j++) {
    c_int a = big->p[j], ae = big->p[j + 1], b = j < small->n ? small->p[j] : 0, be = j < small->n ? small->p[j + 1] : 0;
    U->p[j] = nz;
    while (a < ae || b < be) {                                   /* merge by row index (both sorted) */
      if (b >= be || (a < ae && big->i[a] <= small->i[b])) {
        if (b < be && big->i[a] == small->i[b]) b++;
        U->i[nz] = big->i[a]; U->x[nz] = big->x[a]; a++;
      } else { U->i[nz] = small->i[b]; U->x[nz] = 0.0; b++; }
      nz++;
    }
  }
  U->p[big->n] = nz;
  return U;
}

static int csc_rows_sorted(const csc *M) {
  c_int j, p;
  for (j = 0; j < M->n; j++)
    for (p = M->p[j] + 1; p < M->p[j + 1]; p++) if (M->i[p] <= M->i[p - 1]) return 0;
  return 1;
}

/* entry (r, j) of M, or -1 */
static c_int csc_find(const csc *M, c_int r, c_int j) {
  c_int p;
  for (p = M->p[j]; p < M->p[j + 1]; p++) if (M->i[p] == r) return p;
  return -1;
}

/* tables of horizon N on the store's patterns: value maps, the nominal value rows (zeros outside the horizon's entries), the step
 * program over the live blocks */
static c_int single_tables(osqp_horizon *h, c_int N) {
  rldl_stage_dims d = h->dims;
  csc *P = 0, *A = 0;
  int *mp = 0, *ma = 0;
  double *rowP = 0, *rowA = 0;
  c_int rc = 0, j, p, nzP, nzA, nzPm = h->Pmax->p[h->Pmax->n], nzAm = h->Amax->p[h->Amax->n];
  if (h->mapP[N]) return 0;
  d.N = N;
  rc = rldl_setup_AP_matrices(&d, h->blk[0], h->blk[1], h->blk[2], h->blk[3], h->blk[4], h->blk[5], h->blk[6], &P, &A, 0, 0, 0, 0, 0, 0);
  if (rc) return rc;
  nzP = P->p[P->n]; nzA = A->p[A->n];
  mp = (int *)malloc(sizeof(int) * (size_t)(nzP + 1)); ma = (int *)malloc(sizeof(int) * (size_t)(nzA + 1));
  rowP = (double *)calloc((size_t)nzPm + 1, sizeof(double)); rowA = (double *)calloc((size_t)nzAm + 1, sizeof(double));
  if (!mp || !ma || !rowP || !rowA) rc = RLDL_MEM_ALLOC_ERROR;
  for (j = 0; !rc && j < P->n; j++)
    for (p = P->p[j]; p < P->p[j + 1]; p++) {
      const c_int q = csc_find(h->Pmax, P->i[p], j);
      if (q < 0) { rc = 1; break; }
      mp[p] = (int)q; rowP[q] = P->x[p];
    }
  for (j = 0; !rc && j < A->n; j++)
    for (p = A->p[j]; p < A->p[j + 1]; p++) {
      const c_int q = csc_find(h->Amax, A->i[p], j);
      if (q < 0) { rc = 1; break; }
      ma[p] = (int)q; rowA[q] = A->x[p];
    }
  if (!rc) {
    if (!HIP_OK(hipMalloc((void **)&h->mapP[N], sizeof(int) * (size_t)(nzP + 1))) || !HIP_OK(hipMalloc((void **)&h->mapA[N], sizeof(int) * (size_t)(nzA + 1))) ||
        !HIP_OK(hipMalloc((void **)&h->nomP[N], sizeof(double) * (size_t)nzPm + 8)) || !HIP_OK(hipMalloc((void **)&h->nomA[N], sizeof(double) * (size_t)nzAm + 8)) ||
        !HIP_OK(hipMemcpy(h->mapP[N], mp, sizeof(int) * (size_t)nzP, hipMemcpyHostToDevice)) ||
        !HIP_OK(hipMemcpy(h->mapA[N], ma, sizeof(int) * (size_t)nzA, hipMemcpyHostToDevice)) ||
        !HIP_OK(hipMemcpy(h->nomP[N], rowP, sizeof(double) * (size_t)nzPm, hipMemcpyHostToDevice)) ||
        !HIP_OK(hipMemcpy(h->nomA[N], rowA, sizeof(double) * (size_t)nzAm, hipMemcpyHostToDevice)))
      rc = RLDL_MEM_ALLOC_ERROR;
    h->nnzPh[N] = nzP; h->nnzAh[N] = nzA;
  }
  if (!rc && N < h->Nmax && rldl_stage_prog_prefix(h->ws[h->Nmax]->ls, (int)(2 * N + 2), &h->progN[N], &h->nstepsN[N])) rc = 1;
  if (rc) {
    if (h->mapP[N]) { (void)hipFree(h->mapP[N]); h->mapP[N] = 0; }
    if (h->mapA[N]) { (void)hipFree(h->mapA[N]); h->mapA[N] = 0; }
    if (h->nomP[N]) { (void)hipFree(h->nomP[N]); h->nomP[N] = 0; }
    if (h->nomA[N]) { (void)hipFree(h->nomA[N]); h->nomA[N] = 0; }
  }
  free(mp); free(ma); free(rowP); free(rowA);
  rldl_csc_free(P); rldl_csc_free(A);
  return rc;
}

/* move the store to horizon Nnew; p = first stage whose values change (0: everything, the first activation) */
static c_int single_activate(osqp_horizon *h, c_int Nnew, c_int p, const c_float *d_q, const c_float *d_l, const c_float *d_u) {
  osqp_batch *w = h->ws[h->Nmax];
  const rldl_stage_dims *d = &h->dims;
  rldl_dev_stage *G = &w->ls->dsym.stage;
  hipStream_t st = (hipStream_t)h->stream;
  const c_int nN = Nnew * (d->nx + d->nu), mN = Nnew * (d->nx + d->ny) + d->nt;
  const c_int n_keep = p * (d->nx + d->nu), m_keep = p * (d->nx + d->ny);
  const c_int col_keep = p > 0 ? d->nu + (p - 1) * (d->nx + d->nu) : 0;      /* columns of P and A before stage p */
  const c_int cP = h->Pmax->p[col_keep], cA = h->Amax->p[col_keep];
  const c_int term_old = h->N * (d->nx + d->ny), term_new = Nnew * (d->nx + d->ny);
  c_int rc;
  int k;
  rc = single_tables(h, Nnew);
  if (rc) return rc;
  if (w->loop_pending && osqp_batch_wait(w)) return 1;
  h->last_created = 0; h->last_pivot = p; h->last_reused = 0;
  /* values of the stages >= p: the nominal row of the new horizon (zeros behind its terminal stage) */
  if (rldl_launch_bcast_range((int)h->batch, (int)w->nnzP, (int)cP, (int)(w->nnzP - cP), w->Px, h->nomP[Nnew], w->stream) ||
      rldl_launch_bcast_range((int)h->batch, (int)w->nnzA, (int)cA, (int)(w->nnzA - cA), w->Ax, h->nomA[Nnew], w->stream))
    return 1;
  if (rldl_launch_horizon_vectors((int)h->batch, (int)nN, (int)w->n, (int)mN, (int)w->m, d_q, d_l, d_u, w->q, w->l, w->u, w->stream)) return 1;
  (void)hipMemsetAsync(h->n_reused, 0, sizeof(int) * RLDL_NACT_SLOTS, st);
  if (rldl_launch_horizon_rho(&w->ls->dsym, &w->W, w->ls->num.status, (int)m_keep, (int)(2 * p), h->b0v, h->n_reused, w->stream)) return 1;
  /* the live blocks of the new horizon and the step program over their tiles */
  if (Nnew < h->Nmax) {
    G->nb_act = (int)(2 * Nnew + 2);
    G->npos_skip = (int)(w->n - d->nu - Nnew * (d->nx + d->nu));           /* variable positions of the blocks behind block 2 Nnew + 1 */
    G->pv_prog = h->progN[Nnew]; G->pv_nsteps = h->nstepsN[Nnew];
  } else { G->nb_act = 0; G->npos_skip = 0; G->pv_prog = h->prog0; G->pv_nsteps = h->nsteps0; }
  if (rldl_launch_kkt_assemble(&w->ls->dsym, &w->ls->num, w->Px, w->Ax, w->W.rho_vec, 0, 0, w->stream)) return 1;
  if (rldl_launch_stage_factor_each(&w->ls->dsym, &w->ls->num, h->b0v, 1, w->stream)) return 1;
  if (!HIP_OK(hipMemcpyAsync(h->h_reused, h->n_reused, sizeof(int) * RLDL_NACT_SLOTS, hipMemcpyDeviceToHost, st))) return 1;
  if (rldl_batch_check_status(w->ls)) return RLDL_NONCVX_ERROR;   /* synchronises the stream */
  if (p > 0) for (k = 0; k < RLDL_NACT_SLOTS; k++) h->last_reused += h->h_reused[k];
  if (p > 0 && w->st.warm_start) {
    if (rldl_launch_horizon_state_single(&w->W, (int)w->n, (int)w->m, (int)n_keep, (int)m_keep, (int)term_old, (int)term_new, (int)d->nt, w->stream)) return 1;
    if (rldl_launch_matvec_A(&w->ls->dsym, &w->W, w->W.x, w->W.z, w->stream)) return 1;   /* z = A x, osqp.c:945 */
  }
  osqp_batch_reset_info(w);
  h->N = Nnew; h->dims.N = Nnew;
  return 0;
}

/* 0: the store is up at horizon dims->N; 2: the problem does not qualify (the caller takes the per-horizon workspaces); else an error */
static c_int single_setup(osqp_horizon *h, const c_float *d_q, const c_float *d_l, const c_float *d_u) {
  rldl_stage_dims d = h->dims;
  csc *Qi_u = 0, *Ai_u = 0;
  double *qp = 0, *lp = 0, *up = 0;
  osqp_batch *w = 0;
  const c_int N0 = h->dims.N;
  const c_int nmax = h->Nmax * (d.nx + d.nu), mmax = h->Nmax * (d.nx + d.ny) + d.nt;
  c_int rc = 2;
  if (h->st.scaling || getenv("RLDL_HORIZON_MULTI") || getenv("RLDL_NO_STAGE_PROD") || getenv("RLDL_NO_STAGE_FACTOR") || getenv("RLDL_NO_STAGE_SOLVE") ||
      getenv("RLDL_HORIZON_FULL") || getenv("RLDL_STAGE_LDS"))
    return 2;
  if (d.nt > d.nx + d.ny || d.nt > 64 || h->blk[2]->n > h->blk[1]->n) return 2;
  for (rc = 0; rc < 7; rc++) if (!csc_rows_sorted(h->blk[rc])) return 2;
  rc = 2;
  Qi_u = csc_union_corner(h->blk[1], h->blk[2]);                  /* Qi + QN on the state part */
  Ai_u = csc_union_corner(h->blk[4], h->blk[6]);                  /* Ai + AN in the first nt rows of the state columns */
  if (!Qi_u || !Ai_u) { rc = RLDL_MEM_ALLOC_ERROR; goto out; }
  h->mapP = (int **)calloc((size_t)h->Nmax + 1, sizeof(int *)); h->mapA = (int **)calloc((size_t)h->Nmax + 1, sizeof(int *));
  h->progN = (int **)calloc((size_t)h->Nmax + 1, sizeof(int *)); h->nstepsN = (int *)calloc((size_t)h->Nmax + 1, sizeof(int));
  h->nnzPh = (c_int *)calloc((size_t)h->Nmax + 1, sizeof(c_int)); h->nnzAh = (c_int *)calloc((size_t)h->Nmax + 1, sizeof(c_int));
  if (!h->mapP || !h->mapA || !h->progN || !h->nstepsN || !h->nnzPh || !h->nnzAh) { rc = RLDL_MEM_ALLOC_ERROR; goto out; }
  if (!HIP_OK(hipMalloc((void **)&qp, sizeof(double) * (size_t)h->batch * (size_t)nmax + 8)) ||
      !HIP_OK(hipMalloc((void **)&lp, sizeof(double) * (size_t)h->batch * (size_t)mmax + 8)) ||
      !HIP_OK(hipMalloc((void **)&up, sizeof(double) * (size_t)h->batch * (size_t)mmax + 8))) { rc = RLDL_MEM_ALLOC_ERROR; goto out; }
  if (rldl_launch_horizon_vectors((int)h->batch, (int)(N0 * (d.nx + d.nu)), (int)nmax, (int)(N0 * (d.nx + d.ny) + d.nt), (int)mmax, d_q, d_l, d_u,
                                  qp, lp, up, h->stream) || !HIP_OK(hipStreamSynchronize((hipStream_t)h->stream))) { rc = 1; goto out; }
  d.N = h->Nmax;
  rc = osqp_batch_setup_recursive(&w, h->batch, &d, h->blk[0], Qi_u, h->blk[2], h->blk[3], Ai_u, h->blk[5], h->blk[6], qp, lp, up, &h->st,
                                  &h->Pmax, &h->Amax, h->stream);
  if (rc) goto out;
  h->ws[h->Nmax] = w;
  if (!w->ls->dsym.stage.pv_ok || w->ls->dsym.stage.nb != 2 * h->Nmax + 2 || !w->ls->num.Ti) {   /* no product tri-solve on this pattern */
    osqp_batch_cleanup(w); h->ws[h->Nmax] = 0; rldl_csc_free(h->Pmax); rldl_csc_free(h->Amax); h->Pmax = h->Amax = 0;
    rc = 2; goto out;
  }
  h->prog0 = w->ls->dsym.stage.pv_prog; h->nsteps0 = w->ls->dsym.stage.pv_nsteps;
  h->single = 1;
  h->N = h->Nmax;                                                 /* (what the store holds right now: the nominal problem at Nmax) */
  rc = single_activate(h, N0, 0, d_q, d_l, d_u);
out:
  if (qp) (void)hipFree(qp);
  if (lp) (void)hipFree(lp);
  if (up) (void)hipFree(up);
  rldl_csc_free(Qi_u); rldl_csc_free(Ai_u);
  return rc;
}

c_int osqp_horizon_is_single(const osqp_horizon *h) { return h ? h->single : 0; }

/* resident numeric workspaces (factor, tiles, iterates, problem data): 1 with the single store, one per visited horizon otherwise */
c_int osqp_horizon_workspaces(const osqp_horizon *h) {
  c_int k, cnt = 0;
  if (!h) return 0;
  for (k = 0; k <= h->Nmax; k++) cnt += h->ws[k] ? 1 : 0;
  return cnt;
}

/* row strides of the workspace's per-instance arrays: n and m of Nmax with the single store, of the current horizon otherwise */
c_int osqp_horizon_ld(const osqp_horizon *h, c_int *ld_n, c_int *ld_m) {
  const osqp_batch *w;
  if (!h) return 1;
  w = h->single ? h->ws[h->Nmax] : h->ws[h->N];
  if (ld_n) *ld_n = w->n;
  if (ld_m) *ld_m = w->m;
  return 0;
}

/* osqp_update_P_A for the current horizon: values in the order of ITS assembled P / A (rldl_setup_AP_matrices at N), device arrays
 * [batch][nnz]; either may be null */
c_int osqp_horizon_update_P_A(osqp_horizon *h, const c_float *d_Px, const c_float *d_Ax) {
  osqp_batch *w;
  c_int rc;
  if (!h) return 7;
  if (!h->single) return osqp_batch_update_P_A(h->ws[h->N], d_Px, d_Ax);
  w = h->ws[h->Nmax];
  if (w->loop_pending && osqp_batch_wait(w)) return 1;
  if (d_Px && rldl_launch_scatter_rows((int)h->batch, (int)h->nnzPh[h->N], (int)w->nnzP, h->mapP[h->N], d_Px, w->Px, w->stream)) return 1;
  if (d_Ax && rldl_launch_scatter_rows((int)h->batch, (int)h->nnzAh[h->N], (int)w->nnzA, h->mapA[h->N], d_Ax, w->Ax, w->stream)) return 1;
  rc = rldl_batch_update_matrices(w->ls, d_Px ? w->Px : 0, d_Ax ? w->Ax : 0);
  osqp_batch_reset_info(w);
  return rc;
}

/* osqp_warm_start for the current horizon: packed [batch][n_N], [batch][m_N] */
c_int osqp_horizon_warm_start(osqp_horizon *h, const c_float *d_x, const c_float *d_y) {
  osqp_batch *w;
  const rldl_stage_dims *d;
  if (!h || !d_x || !d_y) return 1;
  if (!h->single) return osqp_batch_warm_start(h->ws[h->N], d_x, d_y);
  w = h->ws[h->Nmax]; d = &h->dims;
  if (!w->st.warm_start) w->st.warm_start = 1;
  if (!HIP_OK(hipMemsetAsync(w->W.x, 0, sizeof(double) * (size_t)h->batch * (size_t)w->n, (hipStream_t)w->stream)) ||
      !HIP_OK(hipMemsetAsync(w->W.y, 0, sizeof(double) * (size_t)h->batch * (size_t)w->m, (hipStream_t)w->stream)) ||
      !HIP_OK(hipMemcpy2DAsync(w->W.x, sizeof(double) * (size_t)w->n, d_x, sizeof(double) * (size_t)(h->N * (d->nx + d->nu)),
                               sizeof(double) * (size_t)(h->N * (d->nx + d->nu)), (size_t)h->batch, hipMemcpyDeviceToDevice, (hipStream_t)w->stream)) ||
      !HIP_OK(hipMemcpy2DAsync(w->W.y, sizeof(double) * (size_t)w->m, d_y, sizeof(double) * (size_t)(h->N * (d->nx + d->ny) + d->nt),
                               sizeof(double) * (size_t)(h->N * (d->nx + d->ny) + d->nt), (size_t)h->batch, hipMemcpyDeviceToDevice, (hipStream_t)w->stream)))
    return 1;
  return rldl_launch_matvec_A(&w->ls->dsym, &w->W, w->W.x, w->W.z, w->stream) ? 1 : 0;
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
  rc = single_setup(h, d_q, d_l, d_u);                            /* one store for every horizon when the problem qualifies ... */
  if (rc == 2) { h->single = 0; h->N = dims->N; h->dims.N = dims->N; rc = create_workspace(h, h->N, d_q, d_l, d_u); }   /* ... else a workspace per visited horizon */
  if (rc) { osqp_horizon_free(h); return rc; }
  *hp = h;
  return 0;
}

osqp_batch *osqp_horizon_workspace(osqp_horizon *h) { return h ? h->ws[h->single ? h->Nmax : h->N] : 0; }
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
  if (h->single) return single_activate(h, Nnew, h->N < Nnew ? h->N : Nnew, d_q, d_l, d_u);
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
