/*
 * oracle/linsys_oracle.c -- CPU ORACLE (test infrastructure only; see osqp_oracle.h).
 *
 * Restates, per instance:
 *   src/cs.c          : triplet_to_csc :55-88, csc_cumsum :124-138, csc_pinv :140-151, csc_symperm :153-206
 *   src/kkt.c         : form_KKT :6-177, update_KKT_P :184-203, update_KKT_A :205-212, update_KKT_param2 :214-222
 *   lin_sys/direct/qdldl/qdldl_interface.c :
 *                       LDL_factor :53-96, permute_KKT :99-166, init :170-316, LDLSolve :550-556,
 *                       solve :559-585, update_matrices :590-602, update_rho_vec :605-619, free :17-43
 */
#include <stdlib.h>
#include <string.h>
#include "osqp_oracle.h"

/* ------------------------------------------------------------------ csc ---- */
orc_csc *orc_csc_spalloc(orc_int m, orc_int n, orc_int nzmax, int values, int triplet) {
  orc_csc *A = (orc_csc *)calloc(1, sizeof(orc_csc));
  if (!A) return 0;
  if (nzmax < 1) nzmax = 1;
  A->m = m; A->n = n; A->nzmax = nzmax; A->nz = triplet ? 0 : -1;
  A->p = (orc_int *)malloc(sizeof(orc_int) * (size_t)(triplet ? nzmax : n + 1));
  A->i = (orc_int *)malloc(sizeof(orc_int) * (size_t)nzmax);
  A->x = values ? (orc_float *)malloc(sizeof(orc_float) * (size_t)nzmax) : 0;
  return A;
}

void orc_csc_spfree(orc_csc *A) {
  if (!A) return;
  free(A->p); free(A->i); free(A->x); free(A);
}

orc_csc *orc_csc_from_arrays(orc_int m, orc_int n, const orc_int *p, const orc_int *i, const orc_float *x) {
  orc_int nz = p[n];
  orc_csc *A = orc_csc_spalloc(m, n, nz, 1, 0);
  memcpy(A->p, p, sizeof(orc_int) * (size_t)(n + 1));
  if (nz) { memcpy(A->i, i, sizeof(orc_int) * (size_t)nz); memcpy(A->x, x, sizeof(orc_float) * (size_t)nz); }
  return A;
}

static orc_int cumsum(orc_int *p, orc_int *c, orc_int n) { /* src/cs.c:124-138 */
  orc_int i, nz = 0;
  for (i = 0; i < n; i++) { p[i] = nz; nz += c[i]; c[i] = p[i]; }
  p[n] = nz;
  return nz;
}

orc_csc *orc_triplet_to_csc(const orc_csc *T, orc_int *TtoC) { /* src/cs.c:55-88 */
  orc_int k, p, nz = T->nz;
  orc_csc *C = orc_csc_spalloc(T->m, T->n, nz, T->x != 0, 0);
  orc_int *w = (orc_int *)calloc((size_t)(T->n > 0 ? T->n : 1), sizeof(orc_int));
  for (k = 0; k < nz; k++) w[T->p[k]]++;
  cumsum(C->p, w, T->n);
  for (k = 0; k < nz; k++) {
    p = w[T->p[k]]++;
    C->i[p] = T->i[k];
    if (C->x) { C->x[p] = T->x[k]; if (TtoC) TtoC[k] = p; }
  }
  free(w);
  return C;
}

orc_int *orc_csc_pinv(const orc_int *p, orc_int n) { /* src/cs.c:140-151 */
  orc_int k, *pinv = (orc_int *)malloc(sizeof(orc_int) * (size_t)(n > 0 ? n : 1));
  for (k = 0; k < n; k++) pinv[p[k]] = k;
  return pinv;
}

orc_csc *orc_csc_symperm(const orc_csc *A, const orc_int *pinv, orc_int *AtoC, int values) { /* src/cs.c:153-206 */
  orc_int i, j, p, q, i2, j2, n = A->n;
  orc_csc *C = orc_csc_spalloc(n, n, A->p[n], values && A->x, 0);
  orc_int *w = (orc_int *)calloc((size_t)(n > 0 ? n : 1), sizeof(orc_int));
  for (j = 0; j < n; j++) {
    j2 = pinv ? pinv[j] : j;
    for (p = A->p[j]; p < A->p[j + 1]; p++) {
      i = A->i[p];
      if (i > j) continue;
      i2 = pinv ? pinv[i] : i;
      w[i2 > j2 ? i2 : j2]++;
    }
  }
  cumsum(C->p, w, n);
  for (j = 0; j < n; j++) {
    j2 = pinv ? pinv[j] : j;
    for (p = A->p[j]; p < A->p[j + 1]; p++) {
      i = A->i[p];
      if (i > j) continue;
      i2 = pinv ? pinv[i] : i;
      q = w[i2 > j2 ? i2 : j2]++;
      C->i[q] = i2 < j2 ? i2 : j2;
      if (C->x) C->x[q] = A->x[p];
      if (AtoC) AtoC[p] = q;
    }
  }
  free(w);
  return C;
}

/* ------------------------------------------------------------------ KKT ---- */
orc_csc *orc_form_KKT(const orc_csc *P, const orc_csc *A, orc_float param1, const orc_float *param2,
                      orc_int *PtoKKT, orc_int *AtoKKT, orc_int **Pdiag_idx, orc_int *Pdiag_n,
                      orc_int *param2toKKT) { /* src/kkt.c:6-177, CSC format only */
  orc_int nKKT = P->m + A->m, nnzmax = P->p[P->n] + P->m + A->p[A->n] + A->m;
  orc_int ptr, i, j, z = 0;
  orc_csc *T = orc_csc_spalloc(nKKT, nKKT, nnzmax, 1, 1), *KKT;
  orc_int *TtoC;
  if (Pdiag_idx) { *Pdiag_idx = (orc_int *)malloc(sizeof(orc_int) * (size_t)(P->m > 0 ? P->m : 1)); *Pdiag_n = 0; }

  for (j = 0; j < P->n; j++) {
    if (P->p[j] == P->p[j + 1]) { /* empty column: diagonal sigma only */
      T->i[z] = j; T->p[z] = j; T->x[z] = param1; z++;
    }
    for (ptr = P->p[j]; ptr < P->p[j + 1]; ptr++) {
      i = P->i[ptr];
      T->i[z] = i; T->p[z] = j; T->x[z] = P->x[ptr];
      if (PtoKKT) PtoKKT[ptr] = z;
      if (i == j) {
        T->x[z] += param1;
        if (Pdiag_idx) { (*Pdiag_idx)[*Pdiag_n] = ptr; (*Pdiag_n)++; }
      }
      z++;
      if (i < j && ptr + 1 == P->p[j + 1]) { /* column ended above the diagonal */
        T->i[z] = j; T->p[z] = j; T->x[z] = param1; z++;
      }
    }
  }
  for (j = 0; j < A->n; j++)
    for (ptr = A->p[j]; ptr < A->p[j + 1]; ptr++) {
      T->p[z] = P->m + A->i[ptr]; T->i[z] = j; T->x[z] = A->x[ptr];
      if (AtoKKT) AtoKKT[ptr] = z;
      z++;
    }
  for (j = 0; j < A->m; j++) {
    T->i[z] = j + P->n; T->p[z] = j + P->n; T->x[z] = -param2[j];
    if (param2toKKT) param2toKKT[j] = z;
    z++;
  }
  T->nz = z;

  TtoC = (orc_int *)malloc(sizeof(orc_int) * (size_t)(z > 0 ? z : 1));
  KKT = orc_triplet_to_csc(T, TtoC);
  if (PtoKKT) for (i = 0; i < P->p[P->n]; i++) PtoKKT[i] = TtoC[PtoKKT[i]];
  if (AtoKKT) for (i = 0; i < A->p[A->n]; i++) AtoKKT[i] = TtoC[AtoKKT[i]];
  if (param2toKKT) for (i = 0; i < A->m; i++) param2toKKT[i] = TtoC[param2toKKT[i]];
  free(TtoC);
  orc_csc_spfree(T);
  return KKT;
}

void orc_update_KKT_P(orc_csc *KKT, const orc_csc *P, const orc_int *PtoKKT, orc_float param1,
                      const orc_int *Pdiag_idx, orc_int Pdiag_n) { /* src/kkt.c:184-203 */
  orc_int i;
  for (i = 0; i < P->p[P->n]; i++) KKT->x[PtoKKT[i]] = P->x[i];
  for (i = 0; i < Pdiag_n; i++) KKT->x[PtoKKT[Pdiag_idx[i]]] += param1;
}

void orc_update_KKT_A(orc_csc *KKT, const orc_csc *A, const orc_int *AtoKKT) { /* src/kkt.c:205-212 */
  orc_int i;
  for (i = 0; i < A->p[A->n]; i++) KKT->x[AtoKKT[i]] = A->x[i];
}

void orc_update_KKT_param2(orc_csc *KKT, const orc_float *param2, const orc_int *param2toKKT, orc_int m) {
  orc_int i; /* src/kkt.c:214-222 */
  for (i = 0; i < m; i++) KKT->x[param2toKKT[i]] = -param2[i];
}

/* -------------------------------------------------------------- ordering ---- */
/* Exact minimum degree on the explicit elimination graph, ties to the lowest index.  The
 * reference calls its vendored AMD (qdldl_interface.c:110-114), which cannot be built here;
 * any fill-reducing permutation gives the same solve results, only L's pattern differs.  Tests
 * normally hand the PRODUCT's permutation to the oracle so L, D, etree compare entry by entry. */
void orc_min_degree_order(orc_int n, const orc_int *Ap, const orc_int *Ai, orc_int *perm) {
  char *adj = (char *)calloc((size_t)n * (size_t)n + 1, 1);
  char *gone = (char *)calloc((size_t)n + 1, 1);
  orc_int i, j, p, k;
  for (j = 0; j < n; j++)
    for (p = Ap[j]; p < Ap[j + 1]; p++) {
      i = Ai[p];
      if (i != j) { adj[i * n + j] = 1; adj[j * n + i] = 1; }
    }
  for (k = 0; k < n; k++) {
    orc_int best = -1, bestdeg = n + 1;
    for (i = 0; i < n; i++) {
      orc_int d = 0;
      if (gone[i]) continue;
      for (j = 0; j < n; j++) d += adj[i * n + j];
      if (d < bestdeg) { bestdeg = d; best = i; }
    }
    perm[k] = best; gone[best] = 1;
    for (i = 0; i < n; i++) {
      if (!adj[best * n + i]) continue;
      for (j = 0; j < n; j++)
        if (adj[best * n + j] && i != j) { adj[i * n + j] = 1; }
    }
    for (i = 0; i < n; i++) { adj[best * n + i] = 0; adj[i * n + best] = 0; }
  }
  free(adj); free(gone);
}

/* --------------------------------------------------------------- backend ---- */
void orc_linsys_free(orc_linsys *s) { /* qdldl_interface.c:17-43 */
  if (!s) return;
  orc_csc_spfree(s->L); orc_csc_spfree(s->KKT);
  free(s->P); free(s->D); free(s->Dinv); free(s->bp); free(s->sol); free(s->rho_inv_vec);
  free(s->Pdiag_idx); free(s->PtoKKT); free(s->AtoKKT); free(s->rhotoKKT);
  free(s->etree); free(s->Lnz); free(s->iwork); free(s->bwork); free(s->fwork);
  free(s);
}

static orc_int ldl_factor(orc_csc *K, orc_linsys *s, orc_int nvar) { /* qdldl_interface.c:53-96 */
  orc_int sum_Lnz = orc_qdldl_etree(K->n, K->p, K->i, s->iwork, s->Lnz, s->etree), st;
  if (sum_Lnz < 0) return sum_Lnz;
  s->L->i = (orc_int *)malloc(sizeof(orc_int) * (size_t)(sum_Lnz > 0 ? sum_Lnz : 1));
  s->L->x = (orc_float *)malloc(sizeof(orc_float) * (size_t)(sum_Lnz > 0 ? sum_Lnz : 1));
  s->L->nzmax = sum_Lnz;
  st = orc_qdldl_factor(K->n, K->p, K->i, K->x, s->L->p, s->L->i, s->L->x, s->D, s->Dinv, s->Lnz,
                        s->etree, s->bwork, s->iwork, s->fwork);
  if (st < 0) return st;
  if (st < nvar) return -2; /* fewer positive pivots than variables: non-convex */
  return 0;
}

static void permute_KKT(orc_csc **KKT, orc_linsys *s, orc_int Pnz, orc_int Anz, orc_int m,
                        orc_int *PtoKKT, orc_int *AtoKKT, orc_int *rhotoKKT, const orc_int *perm_in) {
  /* qdldl_interface.c:99-166, AMD replaced by perm_in / min degree */
  orc_int n = (*KKT)->n, i, *pinv, *KtoPKPt;
  orc_csc *C;
  if (perm_in) memcpy(s->P, perm_in, sizeof(orc_int) * (size_t)n);
  else orc_min_degree_order(n, (*KKT)->p, (*KKT)->i, s->P);
  pinv = orc_csc_pinv(s->P, n);
  if (!PtoKKT && !AtoKKT && !rhotoKKT) {
    C = orc_csc_symperm(*KKT, pinv, 0, 1);
  } else {
    KtoPKPt = (orc_int *)malloc(sizeof(orc_int) * (size_t)((*KKT)->p[n] > 0 ? (*KKT)->p[n] : 1));
    C = orc_csc_symperm(*KKT, pinv, KtoPKPt, 1);
    if (PtoKKT) for (i = 0; i < Pnz; i++) PtoKKT[i] = KtoPKPt[PtoKKT[i]];
    if (AtoKKT) for (i = 0; i < Anz; i++) AtoKKT[i] = KtoPKPt[AtoKKT[i]];
    if (rhotoKKT) for (i = 0; i < m; i++) rhotoKKT[i] = KtoPKPt[rhotoKKT[i]];
    free(KtoPKPt);
  }
  orc_csc_spfree(*KKT);
  *KKT = C;
  free(pinv);
}

orc_int orc_linsys_init(orc_linsys **sp, const orc_csc *P, const orc_csc *A, orc_float sigma,
                        const orc_float *rho_vec, orc_int polish, const orc_int *perm_in) {
  /* qdldl_interface.c:170-316 */
  orc_linsys *s = (orc_linsys *)calloc(1, sizeof(orc_linsys));
  orc_int i, N, st;
  orc_csc *K;
  *sp = s;
  s->n = P->n; s->m = A->m; N = s->n + s->m; s->sigma = sigma; s->polish = polish;
  s->L = (orc_csc *)calloc(1, sizeof(orc_csc));
  s->L->m = N; s->L->n = N; s->L->nz = -1;
  s->L->p = (orc_int *)malloc(sizeof(orc_int) * (size_t)(N + 1));
  s->Dinv = (orc_float *)malloc(sizeof(orc_float) * (size_t)(N + 1));
  s->D = (orc_float *)malloc(sizeof(orc_float) * (size_t)(N + 1));
  s->P = (orc_int *)malloc(sizeof(orc_int) * (size_t)(N + 1));
  s->bp = (orc_float *)malloc(sizeof(orc_float) * (size_t)(N + 1));
  s->sol = (orc_float *)malloc(sizeof(orc_float) * (size_t)(N + 1));
  s->rho_inv_vec = (orc_float *)malloc(sizeof(orc_float) * (size_t)(s->m + 1));
  s->etree = (orc_int *)malloc(sizeof(orc_int) * (size_t)(N + 1));
  s->Lnz = (orc_int *)malloc(sizeof(orc_int) * (size_t)(N + 1));
  s->iwork = (orc_int *)malloc(sizeof(orc_int) * (size_t)(3 * N + 1));
  s->bwork = (orc_int *)malloc(sizeof(orc_int) * (size_t)(N + 1));
  s->fwork = (orc_float *)malloc(sizeof(orc_float) * (size_t)(N + 1));

  if (polish) { /* :254-265 param2 = delta for every row, no maps kept */
    for (i = 0; i < s->m; i++) s->rho_inv_vec[i] = sigma;
    K = orc_form_KKT(P, A, sigma, s->rho_inv_vec, 0, 0, 0, 0, 0);
    permute_KKT(&K, s, 0, 0, 0, 0, 0, 0, perm_in);
  } else {      /* :266-285 */
    s->PtoKKT = (orc_int *)malloc(sizeof(orc_int) * (size_t)(P->p[P->n] + 1));
    s->AtoKKT = (orc_int *)malloc(sizeof(orc_int) * (size_t)(A->p[A->n] + 1));
    s->rhotoKKT = (orc_int *)malloc(sizeof(orc_int) * (size_t)(s->m + 1));
    for (i = 0; i < s->m; i++) s->rho_inv_vec[i] = 1. / rho_vec[i];
    K = orc_form_KKT(P, A, sigma, s->rho_inv_vec, s->PtoKKT, s->AtoKKT, &s->Pdiag_idx, &s->Pdiag_n, s->rhotoKKT);
    permute_KKT(&K, s, P->p[P->n], A->p[A->n], s->m, s->PtoKKT, s->AtoKKT, s->rhotoKKT, perm_in);
  }
  st = ldl_factor(K, s, P->n);
  if (st < 0) { /* :298-303 */
    orc_csc_spfree(K);
    orc_linsys_free(s);
    *sp = 0;
    return st == -2 || st == -1 ? ORC_NONCVX_ERROR : ORC_NONCVX_ERROR;
  }
  if (polish) orc_csc_spfree(K); else s->KKT = K;
  return 0;
}

static void ldl_solve(orc_float *x, const orc_float *b, orc_linsys *s) { /* :538-556 */
  orc_int j, N = s->L->n;
  for (j = 0; j < N; j++) s->bp[j] = b[s->P[j]];
  orc_qdldl_solve(N, s->L->p, s->L->i, s->L->x, s->Dinv, s->bp);
  for (j = 0; j < N; j++) x[s->P[j]] = s->bp[j];
}

orc_int orc_linsys_solve(orc_linsys *s, orc_float *b) { /* :559-585 */
  orc_int j;
  if (s->polish) { ldl_solve(b, b, s); return 0; }
  ldl_solve(s->sol, b, s);
  for (j = 0; j < s->n; j++) b[j] = s->sol[j];
  for (j = 0; j < s->m; j++) b[j + s->n] += s->rho_inv_vec[j] * s->sol[j + s->n];
  return 0;
}

static orc_int refactor(orc_linsys *s) {
  return orc_qdldl_factor(s->KKT->n, s->KKT->p, s->KKT->i, s->KKT->x, s->L->p, s->L->i, s->L->x,
                          s->D, s->Dinv, s->Lnz, s->etree, s->bwork, s->iwork, s->fwork) < 0;
}

orc_int orc_linsys_update_matrices(orc_linsys *s, const orc_csc *P, const orc_csc *A) { /* :590-602 */
  orc_update_KKT_P(s->KKT, P, s->PtoKKT, s->sigma, s->Pdiag_idx, s->Pdiag_n);
  orc_update_KKT_A(s->KKT, A, s->AtoKKT);
  return refactor(s);
}

orc_int orc_linsys_update_rho_vec(orc_linsys *s, const orc_float *rho_vec) { /* :605-619 */
  orc_int i;
  for (i = 0; i < s->m; i++) s->rho_inv_vec[i] = 1. / rho_vec[i];
  orc_update_KKT_param2(s->KKT, s->rho_inv_vec, s->rhotoKKT, s->m);
  return refactor(s);
}

orc_int orc_linsys_nnzL(orc_linsys *s) { return s->L->p[s->L->n]; }
orc_int orc_linsys_nnzKKT(orc_linsys *s) { return s->KKT ? s->KKT->p[s->KKT->n] : 0; }

void orc_linsys_export(orc_linsys *s, orc_int *P, orc_int *etree, orc_int *Lnz, orc_int *Lp,
                       orc_int *Li, orc_float *Lx, orc_float *D, orc_float *Dinv) {
  orc_int N = s->L->n, nz = s->L->p[N];
  if (P) memcpy(P, s->P, sizeof(orc_int) * (size_t)N);
  if (etree) memcpy(etree, s->etree, sizeof(orc_int) * (size_t)N);
  if (Lnz) memcpy(Lnz, s->Lnz, sizeof(orc_int) * (size_t)N);
  if (Lp) memcpy(Lp, s->L->p, sizeof(orc_int) * (size_t)(N + 1));
  if (Li) memcpy(Li, s->L->i, sizeof(orc_int) * (size_t)nz);
  if (Lx) memcpy(Lx, s->L->x, sizeof(orc_float) * (size_t)nz);
  if (D) memcpy(D, s->D, sizeof(orc_float) * (size_t)N);
  if (Dinv) memcpy(Dinv, s->Dinv, sizeof(orc_float) * (size_t)N);
}

void orc_linsys_export_KKT(orc_linsys *s, orc_int *Kp, orc_int *Ki, orc_float *Kx) {
  orc_int N = s->KKT->n, nz = s->KKT->p[N];
  memcpy(Kp, s->KKT->p, sizeof(orc_int) * (size_t)(N + 1));
  memcpy(Ki, s->KKT->i, sizeof(orc_int) * (size_t)nz);
  memcpy(Kx, s->KKT->x, sizeof(orc_float) * (size_t)nz);
}
