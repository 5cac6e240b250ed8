/*
 * oracle/rldl_oracle.c -- CPU ORACLE (test infrastructure only; see osqp_oracle.h): the stage recursion.
 *
 * Restatement of what /root/reference/src/recursive_ldl.c does when it factorises the stage-interleaved KKT matrix of an
 * MPC problem block by block, with its helper products from src/cs_addon.c:
 *   pivot_even               recursive_ldl.c:686-808    cost block E: LDL', L21 = A E^-1 (L + I), next S = rho^-1 + A E^-1 A'
 *   pivot_odd                recursive_ldl.c:554-680    constraint block S (stored POSITIVE): LDL', Dinv written NEGATED
 *                                                       (:673-675), L21 = -Aij' S^-1 (L + I) (:616, :651-655)
 *   pivot_final              recursive_ldl.c:813-935    terminal constraint block, Dinv negated, no entry dropped
 *   LDL_factorize_recursive  recursive_ldl.c:1139-1318  the driver: block order, the closed-form permutation, X_even[] cache
 *   LDL_update_from_pivot    recursive_ldl.c:946-1110   restart from the cached S of stage `iter_start`
 *   A_times_B_plus_rho       cs_addon.c:146-218         S = Ybar' A' + diag(rho_inv), upper triangle, diagonal gets rho_inv with the
 *                                                       first product that lands on it
 *   A_times_B_plus_I         cs_addon.c:274-339         L21 = scale Ybar' (L + I): diagonal term first, then the column of L ascending
 *   A_minus_B                cs_addon.c:342-411         E_next = Q + sigma I - (columns ny.. of Ybar')' on the leading nx x nx upper part
 *   copy_csc_plus_sigma      cs_addon.c:58-74           sigma only where a diagonal entry is stored
 *   compute_Vhat             recursive_ldl.c:253-327    the border algebra of the "combined" X / Z / Y variant (:2359-2856):
 *                                                       V^ = V L^-T D^-1 against a finished factor, Y^ = Y - V^ D V^' (orc_rldl_border)
 *
 * The reference keeps every block as a sparse CSC matrix; here the blocks are small dense arrays and every loop runs in the
 * reference's order, so structural zeros contribute exact zeros and the floating-point results are the same.  The small LDL'
 * factorisations go through this oracle's QDLDL restatement (qdldl_oracle.c), as the reference's go through QDLDL.
 *
 * Quirks of the reference that are MIRRORED (mirror_drops = 1) and can be switched off (mirror_drops = 0) for exact algebra:
 *   - entries of L and of L21 with |x| <= 1e-10 are dropped when a block is written (:660-668, :789-799): the emitted pattern is
 *     value dependent;
 *   - Ybar' keeps only |x| > 1e-20 in pivot_odd (:633) and |x| > 10e-10 (= 1e-9, sic) in pivot_even (:762); L21 and the next
 *     block are formed from that thinned Ybar';
 * and one that is always mirrored because it changes the matrix being factorised:
 *   - the terminal constraint block reads its rho_inv at index (Nmax - 1) (nx + ny) (:1085, :1293), i.e. the values of the
 *     last interior row block, not its own; `terminal_rho_own` = 1 gives the consistent index N (nx + ny) instead (what the
 *     HIP path and the generic oracle use: the assembled KKT matrix has the terminal rows' own rho there).
 *
 * PINNING STATUS: no test of the reference inspects L, D or P of this path and the reference cannot be built here (QDLDL is an
 * empty submodule) -- "parity unpinned by the reference".  What pins this file: P K P' = L D L' against the assembled KKT
 * matrix, and equality with the generic oracle's factor of the same permuted matrix (tests/test_oracle_rldl.py).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "osqp_oracle.h"

typedef struct {
  orc_int n;                       /* block size */
  orc_int *Lp, *Li;                /* strictly lower CSC of the dense pattern */
  orc_float *Lx, *D, *Dinv;
} blk_ldl;

static void blk_free(blk_ldl *f) { free(f->Lp); free(f->Li); free(f->Lx); free(f->D); free(f->Dinv); memset(f, 0, sizeof(*f)); }

/* LDL' of the n x n block X (column major, leading dimension ld, upper triangle valid) through the QDLDL restatement,
 * as pivot_* do with QDLDL_etree + QDLDL_factor (:589-593, :722-726, :871-875).  Every upper entry is handed over, zeros
 * included (they change nothing numerically).  Returns 0, or -1 when the factorisation reports a zero pivot. */
static int blk_factor(orc_int n, const orc_float *X, orc_int ld, blk_ldl *f) {
  orc_int i, j, k = 0, nnz = n * (n + 1) / 2, sum;
  orc_int *Ap = (orc_int *)malloc(sizeof(orc_int) * (size_t)(n + 1)), *Ai = (orc_int *)malloc(sizeof(orc_int) * (size_t)nnz);
  orc_float *Ax = (orc_float *)malloc(sizeof(orc_float) * (size_t)nnz);
  orc_int *etree = (orc_int *)malloc(sizeof(orc_int) * (size_t)n), *Lnz = (orc_int *)malloc(sizeof(orc_int) * (size_t)n);
  orc_int *iwork = (orc_int *)malloc(sizeof(orc_int) * (size_t)(3 * n)), *bwork = (orc_int *)malloc(sizeof(orc_int) * (size_t)n);
  orc_float *fwork = (orc_float *)malloc(sizeof(orc_float) * (size_t)n);
  int rc = -1;
  memset(f, 0, sizeof(*f));
  f->n = n;
  for (j = 0; j < n; j++) {
    Ap[j] = k;
    for (i = 0; i <= j; i++) { Ai[k] = i; Ax[k++] = X[i + j * ld]; }
  }
  Ap[n] = k;
  sum = orc_qdldl_etree(n, Ap, Ai, iwork, Lnz, etree);
  if (sum >= 0) {
    f->Lp = (orc_int *)malloc(sizeof(orc_int) * (size_t)(n + 1));
    f->Li = (orc_int *)malloc(sizeof(orc_int) * (size_t)(sum + 1));
    f->Lx = (orc_float *)malloc(sizeof(orc_float) * (size_t)(sum + 1));
    f->D = (orc_float *)malloc(sizeof(orc_float) * (size_t)n);
    f->Dinv = (orc_float *)malloc(sizeof(orc_float) * (size_t)n);
    if (orc_qdldl_factor(n, Ap, Ai, Ax, f->Lp, f->Li, f->Lx, f->D, f->Dinv, Lnz, etree, bwork, iwork, fwork) >= 0) rc = 0;
  }
  free(Ap); free(Ai); free(Ax); free(etree); free(Lnz); free(iwork); free(bwork); free(fwork);
  return rc;
}

/* multi right-hand-side substitutions, loop order of QDLDL_Lsolve_mat / QDLDL_Ltsolve_mat (recursive_ldl.c:62-96):
 * R is n x nrhs, column major with leading dimension n */
static void blk_Lsolve_mat(const blk_ldl *f, orc_int nrhs, orc_float *R) {
  orc_int i, p, k, n = f->n;
  for (k = 0; k < nrhs; k++)
    for (i = 0; i < n; i++)
      for (p = f->Lp[i]; p < f->Lp[i + 1]; p++) R[f->Li[p] + k * n] -= f->Lx[p] * R[i + k * n];
}
static void blk_Ltsolve_mat(const blk_ldl *f, orc_int nrhs, orc_float *R) {
  orc_int i, p, k, n = f->n;
  for (k = 0; k < nrhs; k++)
    for (i = n - 1; i >= 0; i--)
      for (p = f->Lp[i]; p < f->Lp[i + 1]; p++) R[i + k * n] -= f->Lx[p] * R[f->Li[p] + k * n];
}

/* output factor under construction */
typedef struct {
  orc_int *Lp, *Li;
  orc_float *Lx, *Dinv;
  orc_int cap, nnz, col;           /* capacity, entries written, next column */
  int overflow;
} lout;

static void lout_push(lout *o, orc_int row, orc_float v) {
  if (o->nnz >= o->cap) { o->overflow = 1; return; }
  o->Li[o->nnz] = row; o->Lx[o->nnz++] = v;
}

/* write one pivot block: for column i the entries of the block's own L (dropped when |x| <= tol, tol < 0: keep all), then
 * row j of L21t (s2 x s1, column major: L21t[j + i * s2] = L21(j, i)) below them; Dinv with the given sign */
static void emit_block(lout *o, const blk_ldl *f, const orc_float *L21t, orc_int s2, orc_float tol, orc_float dsign) {
  orc_int i, p, j, s1 = f->n, start = o->col;
  for (i = 0; i < s1; i++) {
    for (p = f->Lp[i]; p < f->Lp[i + 1]; p++)
      if (tol < 0. || fabs(f->Lx[p]) > tol) lout_push(o, start + f->Li[p], f->Lx[p]);
    for (j = 0; j < s2; j++)
      if (L21t && (tol < 0. ? L21t[j + i * s2] != 0. : fabs(L21t[j + i * s2]) > tol)) lout_push(o, start + s1 + j, L21t[j + i * s2]);
    o->Lp[start + i + 1] = o->nnz;
    o->Dinv[start + i] = dsign * f->Dinv[i];
  }
  o->col += s1;
}

/* L21 = scale * YbT (L + I), the order of A_times_B_plus_I (cs_addon.c:274-339): per column ii the diagonal term first, then
 * the entries of column ii of L (ascending rows); columns < start_col stay empty.  YbT is s2 x s1 (column major, ld s2). */
static void ybt_times_L_plus_I(const blk_ldl *f, const orc_float *YbT, orc_int s2, orc_int start_col, orc_float scale, orc_float *C) {
  orc_int ii, p, j, s1 = f->n;
  memset(C, 0, sizeof(orc_float) * (size_t)(s1 * s2));
  /* (when L has no entry at all the reference copies YbT WITHOUT the scale, cs_addon.c:291-294 -- wrong for scale = -1 and
   *  never reached there, a constraint block is not diagonal; not mirrored: a 1 x 1 block gets scale * YbT here) */
  for (ii = start_col; ii < s1; ii++) {
    for (j = 0; j < s2; j++) C[j + ii * s2] = scale * YbT[j + ii * s2];
    for (p = f->Lp[ii]; p < f->Lp[ii + 1]; p++) {
      const orc_float coeff = f->Lx[p];
      const orc_int jj = f->Li[p];
      for (j = 0; j < s2; j++)
        if (YbT[j + jj * s2] != 0.) C[j + ii * s2] += coeff * scale * YbT[j + jj * s2];
    }
  }
}

/* S = YbT' ... the next constraint block, order of A_times_B_plus_rho (cs_addon.c:146-218): column ii of S walks row ii of
 * Ablk (ascending variable index jj) and adds Ablk(ii, jj) * YbT(jrow, jj) to S(jrow, ii), jrow <= ii; the diagonal receives
 * rho_inv(ii) together with the first product that creates it; a row of Ablk without entries gives the bare rho_inv.
 * Ablk: s2 x s1 (rows of this stage's A block, column major ld s2); YbT: s2 x s1; S: s2 x s2 (ld s2), upper triangle. */
static void next_constraint_block(const orc_float *Ablk, const orc_float *YbT, orc_int s1, orc_int s2, const orc_float *rho_inv, orc_float *S) {
  orc_int ii, jj, jrow;
  memset(S, 0, sizeof(orc_float) * (size_t)(s2 * s2));
  for (ii = 0; ii < s2; ii++) {
    int any = 0, diag_made = 0;
    char *made = (char *)calloc((size_t)s2, 1);
    for (jj = 0; jj < s1; jj++) {
      const orc_float coeff = Ablk[ii + jj * s2];
      if (coeff == 0.) continue;                              /* (not stored in the reference's CSC) */
      any = 1;
      for (jrow = 0; jrow <= ii; jrow++) {
        const orc_float yb = YbT[jrow + jj * s2];
        if (yb == 0.) continue;                               /* (not stored in Ybar') */
        if (!made[jrow]) {
          S[jrow + ii * s2] = coeff * yb;
          if (jrow == ii) { S[jrow + ii * s2] += rho_inv[ii]; diag_made = 1; }
          made[jrow] = 1;
        } else S[jrow + ii * s2] += coeff * yb;
      }
    }
    /* a diagonal that no product reached: the reference stores rho_inv alone only for an EMPTY row of A (:166-170) and would
     * otherwise leave the diagonal out (QDLDL then fails); a dense restatement cannot represent "absent", so it takes rho_inv */
    if (!any || !diag_made) S[ii + ii * s2] += rho_inv[ii];
    free(made);
  }
}

/* dense stage blocks of the assembled matrices: variables [u0 | x1,u1 | ... | x_{N-1},u_{N-1} | x_N], row block k = [ny ; nx],
 * terminal block nt (recursive_ldl.c:1898-1961) */
static orc_int var0(const orc_stage_dims *d, orc_int k) { return k == 0 ? 0 : d->nu + (k - 1) * (d->nx + d->nu); }
static orc_int varsz(const orc_stage_dims *d, orc_int k) { return k == 0 ? d->nu : (k == d->N ? d->nx : d->nx + d->nu); }
static orc_int row0(const orc_stage_dims *d, orc_int k) { return k * (d->nx + d->ny); }
static orc_int rowsz(const orc_stage_dims *d, orc_int k) { return k == d->N ? d->nt : d->nx + d->ny; }

/* E = Q_k + sigma I (copy_csc_plus_sigma: sigma only on STORED diagonal entries), upper triangle, column major ld s1 */
static void cost_block(const orc_stage_dims *d, const orc_csc *P, orc_int k, orc_float sigma, orc_float *E) {
  orc_int c0 = var0(d, k), s1 = varsz(d, k), j, p;
  memset(E, 0, sizeof(orc_float) * (size_t)(s1 * s1));
  for (j = 0; j < s1; j++)
    for (p = P->p[c0 + j]; p < P->p[c0 + j + 1]; p++) {
      const orc_int i = P->i[p] - c0;
      if (i < 0 || i > j) continue;
      E[i + j * s1] = P->x[p] + (i == j ? sigma : 0.);
    }
}
/* A_k as s2 x s1 (column major, ld s2) */
static void constr_block(const orc_stage_dims *d, const orc_csc *A, orc_int k, orc_float *Ab) {
  orc_int c0 = var0(d, k), s1 = varsz(d, k), r0 = row0(d, k), s2 = rowsz(d, k), j, p;
  memset(Ab, 0, sizeof(orc_float) * (size_t)(s1 * s2));
  for (j = 0; j < s1; j++)
    for (p = A->p[c0 + j]; p < A->p[c0 + j + 1]; p++) {
      const orc_int i = A->i[p] - r0;
      if (i >= 0 && i < s2) Ab[i + j * s2] = A->x[p];
    }
}
/* the coupling of row block k-1 to the variables of stage k must be the hard-coded Aij = [0 0; -I 0] of pivot_odd (:582-586) */
static int coupling_is_minus_identity(const orc_stage_dims *d, const orc_csc *A) {
  orc_int k, j, p;
  for (k = 1; k <= d->N; k++) {
    const orc_int c0 = var0(d, k), s1 = varsz(d, k), r0 = row0(d, k - 1), s2 = d->nx + d->ny;
    for (j = 0; j < s1; j++)
      for (p = A->p[c0 + j]; p < A->p[c0 + j + 1]; p++) {
        const orc_int i = A->i[p] - r0;
        if (i < 0 || i >= s2) continue;
        if (!(j < d->nx && i == d->ny + j && A->x[p] == -1.0)) return 0;
      }
  }
  return 1;
}

/* pivot_even (:686-808): E (s1 x s1) with right-hand side A' (A: s2 x s1); emits the block, leaves YbT = (E^-1 A')' (s2 x s1) */
static int pivot_even(lout *o, const orc_float *E, orc_int s1, const orc_float *Ab, orc_int s2, int mirror, orc_float *YbT) {
  blk_ldl f;
  orc_int i, j;
  orc_float *R = (orc_float *)malloc(sizeof(orc_float) * (size_t)(s1 * s2 + 1)), *L21 = (orc_float *)malloc(sizeof(orc_float) * (size_t)(s1 * s2 + 1));
  if (blk_factor(s1, E, s1, &f)) { free(R); free(L21); return -1; }
  for (i = 0; i < s1; i++)
    for (j = 0; j < s2; j++) R[i + j * s1] = Ab[j + i * s2];     /* copy_csc_transpose: A_f = A' (s1 x s2) */
  blk_Lsolve_mat(&f, s2, R);
  for (i = 0; i < s1; i++)
    for (j = 0; j < s2; j++) R[i + j * s1] *= f.Dinv[i];
  blk_Ltsolve_mat(&f, s2, R);                                    /* R = E^-1 A' = Ybar */
  for (i = 0; i < s1; i++)
    for (j = 0; j < s2; j++) {
      const orc_float v = R[i + j * s1];
      YbT[j + i * s2] = (!mirror || fabs(v) > 10e-10) ? v : 0.;   /* Ybar' keeps |x| > 10e-10 (:762) */
    }
  ybt_times_L_plus_I(&f, YbT, s2, 0, 1.0, L21);                  /* L21 = Ybar' (L + I) = A L^-T D^-1 */
  emit_block(o, &f, L21, s2, mirror ? 1e-10 : -1., 1.0);
  blk_free(&f); free(R); free(L21);
  return 0;
}

/* pivot_odd (:554-680): the POSITIVE constraint block S (s1 x s1), right-hand side Aij (s1 x s2v, -1 at (ny + i, i), i < nx);
 * emits the block with Dinv negated, leaves W = S^-1 Aij restricted to its first nx columns, as YbT (nx x s1, ld nx) */
static int pivot_odd(lout *o, const orc_float *S, orc_int s1, orc_int s2v, const orc_stage_dims *d, int mirror, orc_float *YbT) {
  blk_ldl f;
  orc_int i, j, nx = d->nx, ny = d->ny;
  orc_float *R = (orc_float *)calloc((size_t)(s1 * s2v + 1), sizeof(orc_float));
  orc_float *Yfull = (orc_float *)calloc((size_t)(s1 * s2v + 1), sizeof(orc_float)), *L21 = (orc_float *)malloc(sizeof(orc_float) * (size_t)(s1 * s2v + 1));
  if (blk_factor(s1, S, s1, &f)) { free(R); free(Yfull); free(L21); return -1; }
  for (i = 0; i < nx; i++) R[ny + i + i * s1] = -1.0;
  blk_Lsolve_mat(&f, s2v, R);
  for (i = 0; i < s1; i++)
    for (j = 0; j < s2v; j++) R[i + j * s1] *= f.Dinv[i];
  blk_Ltsolve_mat(&f, s2v, R);                                   /* R = S^-1 Aij = W */
  for (i = 0; i < s1; i++)
    for (j = 0; j < nx; j++) {                                   /* width is s2v, but zero beyond nx: ignored (:632) */
      const orc_float v = R[i + j * s1];
      const orc_float kept = (!mirror || fabs(v) > 1e-20) ? v : 0.;
      YbT[j + i * nx] = kept;
      Yfull[j + i * s2v] = kept;
    }
  ybt_times_L_plus_I(&f, Yfull, s2v, ny, -1.0, L21);             /* L21 = -W' (L + I), columns >= ny only (:651-655) */
  emit_block(o, &f, L21, s2v, mirror ? 1e-10 : -1., -1.0);
  blk_free(&f); free(R); free(Yfull); free(L21);
  return 0;
}

/* pivot_final (:813-935): terminal constraint block, nothing below it, nothing dropped, Dinv negated */
static int pivot_final(lout *o, const orc_float *S, orc_int nt) {
  blk_ldl f;
  if (blk_factor(nt, S, nt, &f)) return -1;
  emit_block(o, &f, 0, 0, -1., -1.0);
  blk_free(&f);
  return 0;
}

orc_int orc_rldl_xeven_stride(const orc_stage_dims *d) {
  const orc_int a = d->nx + d->ny, b = d->nt;
  return (a > b ? a : b) * (a > b ? a : b);
}

/* LDL_factorize_recursive (iter_start < 0) or LDL_update_from_pivot restarted at the cached block of stage iter_start >= 0.
 *   P (n x n upper CSC), A (m x n CSC): the assembled problem of horizon d->N; rho_inv[m]; Nmax: see the terminal-rho quirk.
 *   xeven: (N + 1) * orc_rldl_xeven_stride(d) doubles, the X_even[] cache (the positive constraint blocks), written by a
 *          full run, read (block iter_start) and rewritten (later blocks) by a restart.
 *   Lp [n + m + 1], Li / Lx [Lcap], Dinv [n + m], perm [n + m]: outputs; a restart keeps the columns before
 *          nu + iter_start (2 nx + nu + ny) (:969-970) and needs the arrays of the previous run.
 * Returns nnz(L), or -1 zero pivot, -2 L capacity, -3 the coupling blocks are not [0 0; -I 0], -4 bad arguments. */
orc_int orc_rldl_factor(const orc_stage_dims *d, const orc_csc *P, const orc_csc *A, orc_float sigma, const orc_float *rho_inv,
                        orc_int Nmax, orc_int mirror_drops, orc_int terminal_rho_own, orc_int iter_start, orc_float *xeven,
                        orc_int *Lp, orc_int *Li, orc_float *Lx, orc_int Lcap, orc_float *Dinv, orc_int *perm) {
  const orc_int N = d->N, nx = d->nx, nu = d->nu, ny = d->ny, nt = d->nt, nxy = nx + ny, nxu = nx + nu;
  const orc_int n = N * nxu, xs = orc_rldl_xeven_stride(d);
  const int mirror = mirror_drops != 0;
  orc_int k, i, pc = 0, qc = 0, ac = n, smax = nxy > nxu ? nxy : nxu;
  orc_float *E, *Ab, *YbT, *S;
  lout o;
  int rc = 0;
  if (N < 1 || P->n != n || A->n != n || A->m != N * nxy + nt || iter_start >= N) return -4;
  if (!coupling_is_minus_identity(d, A)) return -3;
  if (nt > smax) smax = nt;
  /* the closed-form permutation (pivot_even / pivot_final :720, :869 and the driver :1184, :1234; compute_permutations :1345-1363):
   * cost blocks take the variables in order, constraint blocks the rows, which start at n */
  for (k = 0; k <= N; k++) {
    for (i = 0; i < varsz(d, k); i++) perm[pc++] = qc++;
    for (i = 0; i < rowsz(d, k); i++) perm[pc++] = ac++;
  }
  E = (orc_float *)malloc(sizeof(orc_float) * (size_t)(smax * smax)); Ab = (orc_float *)malloc(sizeof(orc_float) * (size_t)(smax * smax));
  YbT = (orc_float *)malloc(sizeof(orc_float) * (size_t)(smax * smax)); S = (orc_float *)malloc(sizeof(orc_float) * (size_t)(smax * smax));
  o.Lp = Lp; o.Li = Li; o.Lx = Lx; o.Dinv = Dinv; o.cap = Lcap; o.overflow = 0;
  if (iter_start < 0) {
    o.nnz = 0; o.col = 0; Lp[0] = 0;
    k = 0;
  } else {                                                       /* keep Q_0, C_0, ..., Q_{iter_start}; resume with the cached S (:969-970, :995-997) */
    o.col = nu + iter_start * (nxu + nxy);
    o.nnz = Lp[o.col];
    memcpy(S, xeven + iter_start * xs, sizeof(orc_float) * (size_t)(nxy * nxy));
    k = iter_start + 1;
  }
  for (;;) {
    if (iter_start < 0) {
      /* cost block of stage k: E = Q_k + sigma I (- the part of W' that couples x_k to the previous constraint block) */
      const orc_int s1 = varsz(d, k), s2 = rowsz(d, k);
      const orc_float *ri = rho_inv + (k < N ? k * nxy : (terminal_rho_own ? N * nxy : (Nmax - 1) * nxy));
      cost_block(d, P, k, sigma, E);
      if (k > 0) {                                               /* A_minus_B(ny, ny + nx, ...) (:1220-1222): E(r, j) -= YbT(r, ny + j), r <= j < nx */
        orc_int j, r;
        for (j = 0; j < nx; j++)
          for (r = 0; r <= j; r++) E[r + j * s1] -= YbT[r + (ny + j) * nx];
      }
      constr_block(d, A, k, Ab);
      if (pivot_even(&o, E, s1, Ab, s2, mirror, YbT)) { rc = -1; break; }
      /* the constraint block of stage k, stored positive: rho_inv + A E^-1 A' (:1192-1196, :1241-1245, :1289-1293) */
      next_constraint_block(Ab, YbT, s1, s2, ri, S);
      memcpy(xeven + k * xs, S, sizeof(orc_float) * (size_t)(s2 * s2));   /* X_even[k] (:1206, :1252) */
      if (k == N) {
        if (pivot_final(&o, S, nt)) rc = -1;
        break;
      }
      k++;
    }
    iter_start = -1;                                             /* (a restart enters here, at the constraint block of the cached stage) */
    /* constraint block of stage k - 1 against the variables of stage k (nx + nu of them, nx for the terminal stage) */
    if (pivot_odd(&o, S, nxy, varsz(d, k), d, mirror, YbT)) { rc = -1; break; }
  }
  free(E); free(Ab); free(YbT); free(S);
  if (rc) return rc;
  if (o.overflow) return -2;
  return o.nnz;
}


/* The border algebra of the reference's "combined" variant (osqp_setup_combine_recursive :2603-2661, osqp_update_Z_horizon :2814-2816),
 * as compute_Vhat (:253-327) carries it out: a block whose rows V couple it to an ALREADY FACTORISED part (L, Dinv of that part, nf x nf,
 * in the part's own order) is bordered by
 *     W    = L^-1 V'            forward substitution, column by column of L, every right-hand side (row of V) alongside (:270-278)
 *     V^   = (D^-1 W)'          (:280-292; the reference stores only the entries with W != 0)
 *     Y^   = Y - V^ W           (:294-303), i.e. Y - V^ D V^'
 * and Y^ is what gets factorised next.  V: nrows x nf dense, row major (V[r * nf + c]), columns in the factored part's order;
 * Y: nrows x nrows dense row major, symmetric.  Outputs Vhat (same layout as V) and Yhat (same layout as Y).
 * compute_Uhat (:329-389) is the same algebra for the other border of the combined layout. */
void orc_rldl_border(orc_int nf, const orc_int *Lp, const orc_int *Li, const orc_float *Lx, const orc_float *Dinv, orc_int nrows,
                     const orc_float *V, const orc_float *Y, orc_float *Vhat, orc_float *Yhat) {
  orc_float *W = (orc_float *)malloc(sizeof(orc_float) * (size_t)(nf * nrows > 0 ? nf * nrows : 1));
  orc_int i, p, k, r, c;
  for (k = 0; k < nrows; k++)
    for (i = 0; i < nf; i++) W[i + k * nf] = V[k * nf + i];      /* Vtemp: column i of the factored part, right-hand side k (:262-266) */
  for (i = 0; i < nf; i++)                                         /* "Invert by L" (:269-278) */
    for (p = Lp[i]; p < Lp[i + 1]; p++)
      for (k = 0; k < nrows; k++)
        if (W[i + k * nf] != 0.0) W[Li[p] + k * nf] -= Lx[p] * W[i + k * nf];
  for (i = 0; i < nf; i++)                                         /* "Invert by D" (:280-292) */
    for (k = 0; k < nrows; k++) Vhat[k * nf + i] = W[i + k * nf] != 0.0 ? W[i + k * nf] * Dinv[i] : 0.0;
  for (r = 0; r < nrows; r++)
    for (c = 0; c < nrows; c++) Yhat[r * nrows + c] = Y[r * nrows + c];
  for (i = 0; i < nf; i++)                                         /* Yhat -= Vhat(:, i) W(i, :) (:294-303) */
    for (r = 0; r < nrows; r++) {
      const orc_float v = Vhat[r * nf + i];
      if (v == 0.0) continue;
      for (c = 0; c < nrows; c++) Yhat[c + r * nrows] -= v * W[i + c * nf];
    }
  free(W);
}
