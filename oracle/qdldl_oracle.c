/*
 * oracle/qdldl_oracle.c -- CPU ORACLE (test infrastructure only; see osqp_oracle.h).
 *
 * Restatement of the QDLDL v0.1.x contract.  QDLDL's own source is an EMPTY submodule in the
 * reference (.gitmodules:1-3, lin_sys/direct/qdldl/qdldl_sources/), so this follows the published
 * algorithm (up-looking LDL' driven by the elimination tree) and the call-site contract:
 *   QDLDL_etree  : qdldl_interface.c:59-67   (negative return = not upper triangular / empty column)
 *   QDLDL_factor : qdldl_interface.c:74-92   (returns #positive D entries, -1 on zero pivot;
 *                                             workspaces bwork[n], iwork[3n], fwork[n] :249-251)
 *   QDLDL_solve  : qdldl_interface.c:553     (loop bodies visible in the author's multi-RHS clones
 *                                             src/recursive_ldl.c:62-116)
 * L is unit lower triangular, stored strictly-lower CSC with ascending row indices per column.
 */
#include "osqp_oracle.h"

#define ORC_UNKNOWN (-1)

orc_int orc_qdldl_etree(orc_int n, const orc_int *Ap, const orc_int *Ai, orc_int *work,
                        orc_int *Lnz, orc_int *etree) {
  orc_int i, j, p, total = 0;
  for (i = 0; i < n; i++) {
    work[i] = 0; Lnz[i] = 0; etree[i] = ORC_UNKNOWN;
    if (Ap[i] == Ap[i + 1]) return -1; /* empty column => structurally zero diagonal */
  }
  for (j = 0; j < n; j++) {
    work[j] = j;
    for (p = Ap[j]; p < Ap[j + 1]; p++) {
      i = Ai[p];
      if (i > j) return -1;            /* entry below the diagonal */
      while (work[i] != j) {           /* walk towards the root, marking for column j */
        if (etree[i] == ORC_UNKNOWN) etree[i] = j;
        Lnz[i]++;
        work[i] = j;
        i = etree[i];
      }
    }
  }
  for (i = 0; i < n; i++) total += Lnz[i];
  return total;
}

orc_int orc_qdldl_factor(orc_int n, const orc_int *Ap, const orc_int *Ai, const orc_float *Ax,
                         orc_int *Lp, orc_int *Li, orc_float *Lx, orc_float *D, orc_float *Dinv,
                         const orc_int *Lnz, const orc_int *etree, orc_int *bwork, orc_int *iwork,
                         orc_float *fwork) {
  orc_int  *marked = bwork, *ypat = iwork, *path = iwork + n, *next_free = iwork + 2 * n;
  orc_float *y = fwork;
  orc_int i, k, p, positive = 0;

  Lp[0] = 0;
  for (i = 0; i < n; i++) {
    Lp[i + 1] = Lp[i] + Lnz[i];
    marked[i] = 0; y[i] = 0.0; D[i] = 0.0; next_free[i] = Lp[i];
  }
  /* row 0: only the diagonal */
  D[0] = Ax[0];
  if (D[0] == 0.0) return -1;
  if (D[0] > 0.0) positive++;
  Dinv[0] = 1.0 / D[0];

  for (k = 1; k < n; k++) {
    orc_int ny = 0;
    /* scatter column k of the upper triangle; pattern of row k of L = union of etree paths */
    for (p = Ap[k]; p < Ap[k + 1]; p++) {
      orc_int r = Ai[p];
      if (r == k) { D[k] = Ax[p]; continue; }
      y[r] = Ax[p];
      if (!marked[r]) {
        orc_int len = 0, t = r;
        marked[t] = 1; path[len++] = t; t = etree[t];
        while (t != ORC_UNKNOWN && t < k) {
          if (marked[t]) break;
          marked[t] = 1; path[len++] = t; t = etree[t];
        }
        while (len) ypat[ny++] = path[--len];
      }
    }
    /* sparse triangular solve, columns visited in reverse of the order they were listed */
    for (i = ny - 1; i >= 0; i--) {
      orc_int c = ypat[i], slot = next_free[c];
      orc_float yc = y[c];
      for (p = Lp[c]; p < slot; p++) y[Li[p]] -= Lx[p] * yc;
      Li[slot] = k;
      Lx[slot] = yc * Dinv[c];
      D[k] -= yc * Lx[slot];
      next_free[c]++;
      y[c] = 0.0; marked[c] = 0;
    }
    if (D[k] == 0.0) return -1;
    if (D[k] > 0.0) positive++;
    Dinv[k] = 1.0 / D[k];
  }
  return positive;
}

/* x <- L^{-1} x   (src/recursive_ldl.c:62-78 with one right-hand side) */
void orc_qdldl_Lsolve(orc_int n, const orc_int *Lp, const orc_int *Li, const orc_float *Lx, orc_float *x) {
  orc_int i, p;
  for (i = 0; i < n; i++)
    for (p = Lp[i]; p < Lp[i + 1]; p++) x[Li[p]] -= Lx[p] * x[i];
}

/* x <- L^{-T} x   (src/recursive_ldl.c:81-96) */
void orc_qdldl_Ltsolve(orc_int n, const orc_int *Lp, const orc_int *Li, const orc_float *Lx, orc_float *x) {
  orc_int i, p;
  for (i = n - 1; i >= 0; i--)
    for (p = Lp[i]; p < Lp[i + 1]; p++) x[i] -= Lx[p] * x[Li[p]];
}

/* x <- L^{-T} D^{-1} L^{-1} x   (src/recursive_ldl.c:98-116) */
void orc_qdldl_solve(orc_int n, const orc_int *Lp, const orc_int *Li, const orc_float *Lx,
                     const orc_float *Dinv, orc_float *x) {
  orc_int i;
  orc_qdldl_Lsolve(n, Lp, Li, Lx, x);
  for (i = 0; i < n; i++) x[i] *= Dinv[i];
  orc_qdldl_Ltsolve(n, Lp, Li, Lx, x);
}
