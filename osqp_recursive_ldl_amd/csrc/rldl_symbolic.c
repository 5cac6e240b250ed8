/*
 * rldl_symbolic.c -- host-side symbolic analysis (plain C99).  See rldl_symbolic.h.
 *
 * Everything here is integer work done once per sparsity pattern; the results are uploaded to the
 * device and shared by every instance of the batch.  Reference behaviour reproduced (not its code):
 * the KKT layout and scatter maps of src/kkt.c:6-177, the map composition of
 * qdldl_interface.c:136-150, the elimination tree contract of QDLDL_etree (qdldl_interface.c:59-67).
 */
#include "rldl_symbolic.h"

#include <stdlib.h>
#include <string.h>

#define SRC_P 0
#define SRC_A 1
#define SRC_RHO 2
#define SRC_SIG 3

typedef struct { int row, col, kind, idx; } trip;

static void *xmalloc(size_t n) { return malloc(n ? n : 1); }
static void *xcalloc(size_t n, size_t s) { return calloc(n ? n : 1, s ? s : 1); }

static int cmp_trip(const void *a, const void *b) {
  const trip *x = (const trip *)a, *y = (const trip *)b;
  if (x->col != y->col) return x->col < y->col ? -1 : 1;
  if (x->row != y->row) return x->row < y->row ? -1 : 1;
  return 0;
}

/* ------------------------------------------------------------------ ordering ---- */
typedef struct { int *v; int len, cap; } ivec;

static int ivec_push(ivec *a, int x) {
  if (a->len == a->cap) {
    int nc = a->cap ? 2 * a->cap : 8;
    int *nv = (int *)realloc(a->v, sizeof(int) * (size_t)nc);
    if (!nv) return -1;
    a->v = nv; a->cap = nc;
  }
  a->v[a->len++] = x;
  return 0;
}

/* Minimum-degree ordering on the explicit elimination graph.  Ties go to the lowest index, so the
 * result is deterministic.  (The reference calls its vendored AMD, qdldl_interface.c:110-114; any
 * fill-reducing order is a valid private choice of the backend, the API never exposes it.) */
int rldl_order_min_degree(int n, const int *Ap, const int *Ai, int *perm) {
  ivec *adj = (ivec *)xcalloc((size_t)n, sizeof(ivec));
  int *mark = (int *)xcalloc((size_t)n, sizeof(int));
  char *dead = (char *)xcalloc((size_t)n, 1);
  int i, j, p, k, rc = 0;
  if (!adj || !mark || !dead) { rc = -2; goto done; }
  for (j = 0; j < n; j++)
    for (p = Ap[j]; p < Ap[j + 1]; p++) {
      i = Ai[p];
      if (i == j) continue;
      if (ivec_push(&adj[i], j) || ivec_push(&adj[j], i)) { rc = -2; goto done; }
    }
  /* remove duplicate edges */
  for (i = 0; i < n; i++) {
    int w = 0;
    for (p = 0; p < adj[i].len; p++) {
      int u = adj[i].v[p];
      if (mark[u] != i + 1) { mark[u] = i + 1; adj[i].v[w++] = u; }
    }
    adj[i].len = w;
  }
  memset(mark, 0, sizeof(int) * (size_t)n);
  {
    int stamp = 0;
    for (k = 0; k < n; k++) {
      int best = -1, bestdeg = n + 1;
      ivec nb;
      for (i = 0; i < n; i++)
        if (!dead[i] && adj[i].len < bestdeg) { bestdeg = adj[i].len; best = i; }
      perm[k] = best;
      dead[best] = 1;
      nb = adj[best];
      /* the pivot's neighbours become a clique; the pivot leaves every list */
      for (p = 0; p < nb.len; p++) {
        int u = nb.v[p], w = 0, q;
        stamp++;
        for (q = 0; q < adj[u].len; q++) {
          int t = adj[u].v[q];
          if (t == best) continue;
          adj[u].v[w++] = t;
          mark[t] = stamp;
        }
        adj[u].len = w;
        for (q = 0; q < nb.len; q++) {
          int t = nb.v[q];
          if (t == u || mark[t] == stamp) continue;
          if (ivec_push(&adj[u], t)) { rc = -2; goto done; }
          mark[t] = stamp;
        }
      }
      free(adj[best].v);
      adj[best].v = 0; adj[best].len = adj[best].cap = 0;
    }
  }
done:
  if (adj) { for (i = 0; i < n; i++) free(adj[i].v); free(adj); }
  free(mark); free(dead);
  return rc;
}

void rldl_stage_permutation(long long N, long long nx, long long nu, long long ny, long long nt,
                            long long *perm) {
  /* variables [u0 | x1,u1 | ... | x_{N-1},u_{N-1} | x_N], rows [C_0 .. C_{N-1} | C_N];
   * interleave cost and constraint stages: Q0, C0, Q1, C1, ..., QN, CN  (src/recursive_ldl.c:1350-1362) */
  long long nvar = N * (nx + nu), k = 0, pc = 0, ac = nvar, it, i;
  for (i = 0; i < nu; i++) perm[k++] = pc++;
  for (i = 0; i < nx + ny; i++) perm[k++] = ac++;
  for (it = 0; it < N - 1; it++) {
    for (i = 0; i < nx + nu; i++) perm[k++] = pc++;
    for (i = 0; i < nx + ny; i++) perm[k++] = ac++;
  }
  for (i = 0; i < nx; i++) perm[k++] = pc++;
  for (i = 0; i < nt; i++) perm[k++] = ac++;
}

/* ------------------------------------------------------------------- helpers ---- */
static int find_in_col(const int *Li, int lo, int hi, int row) { /* rows ascending in [lo,hi) */
  while (lo < hi) {
    int mid = lo + (hi - lo) / 2;
    if (Li[mid] == row) return mid;
    if (Li[mid] < row) lo = mid + 1; else hi = mid;
  }
  return -1;
}

/* row-order access (CSR) of a CSC pattern: rp[m+1], rj[nnz] (column), rpos[nnz] (CSC slot) */
static int build_csr(int m, int n, const int *Cp, const int *Ci, int **rp_o, int **rj_o, int **rpos_o) {
  int nnz = Cp[n], i, j, p;
  int *rp = (int *)xcalloc((size_t)m + 2, sizeof(int));
  int *rj = (int *)xmalloc(sizeof(int) * (size_t)nnz), *rpos = (int *)xmalloc(sizeof(int) * (size_t)nnz);
  int *w;
  if (!rp || !rj || !rpos) { free(rp); free(rj); free(rpos); return -2; }
  for (p = 0; p < nnz; p++) rp[Ci[p] + 1]++;
  for (i = 0; i < m; i++) rp[i + 1] += rp[i];
  w = (int *)xmalloc(sizeof(int) * (size_t)(m + 1));
  if (!w) { free(rp); free(rj); free(rpos); return -2; }
  memcpy(w, rp, sizeof(int) * (size_t)(m + 1));
  for (j = 0; j < n; j++)
    for (p = Cp[j]; p < Cp[j + 1]; p++) { int q = w[Ci[p]]++; rj[q] = j; rpos[q] = p; }
  free(w);
  *rp_o = rp; *rj_o = rj; *rpos_o = rpos;
  return 0;
}

void rldl_symbolic_free(rldl_symbolic *s) {
  if (!s) return;
  free(s->perm); free(s->pinv); free(s->Kp); free(s->Ki); free(s->PtoK); free(s->Pisdiag);
  free(s->AtoK); free(s->rhotoK); free(s->sigK); free(s->etree); free(s->Lnz); free(s->Lp); free(s->Li);
  free(s->Rp); free(s->Rj); free(s->Rpos); free(s->KtoW); free(s->Up); free(s->Udst); free(s->Uab);
  free(s->Pp); free(s->Pi); free(s->Prp); free(s->Prj); free(s->Prpos);
  free(s->Ap); free(s->Ai); free(s->Arp); free(s->Arj); free(s->Arpos);
  free(s->LtoS); free(s->plan);
  free(s);
}

/* ------------------------------------------------------------------ analysis ---- */
int rldl_symbolic_create(rldl_symbolic **out, long long n64, long long m64, const long long *Pp,
                         const long long *Pi, const long long *Ap, const long long *Ai, int polish,
                         const long long *perm_in) {
  int n = (int)n64, m = (int)m64, N = n + m, nnzP = (int)Pp[n], nnzA = (int)Ap[n];
  int i, j, k, p, nt = 0, rc = 0, nsig = 0;
  rldl_symbolic *s = (rldl_symbolic *)xcalloc(1, sizeof(rldl_symbolic));
  trip *T = (trip *)xmalloc(sizeof(trip) * (size_t)(nnzP + n + nnzA + m));
  int *uAp = 0, *uAi = 0, *work = 0, *mark = 0, *stack = 0, *fill = 0;
  *out = 0;
  if (!s || !T) { rc = -2; goto fail; }
  s->n = n; s->m = m; s->N = N; s->nnzP = nnzP; s->nnzA = nnzA; s->polish = polish;

  /* copies of the problem patterns (int32) + row-order maps for the residual kernels */
  s->Pp = (int *)xmalloc(sizeof(int) * (size_t)(n + 1)); s->Pi = (int *)xmalloc(sizeof(int) * (size_t)nnzP);
  s->Ap = (int *)xmalloc(sizeof(int) * (size_t)(n + 1)); s->Ai = (int *)xmalloc(sizeof(int) * (size_t)nnzA);
  if (!s->Pp || !s->Pi || !s->Ap || !s->Ai) { rc = -2; goto fail; }
  for (j = 0; j <= n; j++) { s->Pp[j] = (int)Pp[j]; s->Ap[j] = (int)Ap[j]; }
  for (p = 0; p < nnzP; p++) s->Pi[p] = (int)Pi[p];
  for (p = 0; p < nnzA; p++) s->Ai[p] = (int)Ai[p];
  for (j = 0; j < n; j++) {
    for (p = s->Pp[j]; p < s->Pp[j + 1]; p++)
      if (s->Pi[p] < 0 || s->Pi[p] > j) { rc = -1; goto fail; } /* P must be upper triangular (auxil.c:842-851) */
    for (p = s->Ap[j]; p < s->Ap[j + 1]; p++)
      if (s->Ai[p] < 0 || s->Ai[p] >= m) { rc = -1; goto fail; }
  }
  if ((rc = build_csr(n, n, s->Pp, s->Pi, &s->Prp, &s->Prj, &s->Prpos))) goto fail;
  if ((rc = build_csr(m, n, s->Ap, s->Ai, &s->Arp, &s->Arj, &s->Arpos))) goto fail;

  /* ---- upper-triangular KKT entries in original coordinates (layout of src/kkt.c:45-122) ---- */
  for (j = 0; j < n; j++) {
    int has_diag = 0;
    for (p = s->Pp[j]; p < s->Pp[j + 1]; p++) {
      T[nt].row = s->Pi[p]; T[nt].col = j; T[nt].kind = SRC_P; T[nt].idx = p; nt++;
      if (s->Pi[p] == j) has_diag = 1;
    }
    if (!has_diag) { T[nt].row = j; T[nt].col = j; T[nt].kind = SRC_SIG; T[nt].idx = nsig++; nt++; }
  }
  for (j = 0; j < n; j++)
    for (p = s->Ap[j]; p < s->Ap[j + 1]; p++) {
      T[nt].row = j; T[nt].col = n + s->Ai[p]; T[nt].kind = SRC_A; T[nt].idx = p; nt++;
    }
  for (j = 0; j < m; j++) { T[nt].row = n + j; T[nt].col = n + j; T[nt].kind = SRC_RHO; T[nt].idx = j; nt++; }
  s->nnzK = nt; s->nsig = nsig;

  /* ---- ordering ---- */
  s->perm = (int *)xmalloc(sizeof(int) * (size_t)N); s->pinv = (int *)xmalloc(sizeof(int) * (size_t)N);
  if (!s->perm || !s->pinv) { rc = -2; goto fail; }
  if (perm_in) {
    for (k = 0; k < N; k++) s->pinv[k] = -1;
    for (k = 0; k < N; k++) {
      long long v = perm_in[k];
      if (v < 0 || v >= N || s->pinv[v] != -1) { rc = -3; goto fail; }
      s->perm[k] = (int)v; s->pinv[v] = k;
    }
  } else {
    /* CSC of the unpermuted upper pattern for the ordering routine */
    uAp = (int *)xcalloc((size_t)N + 2, sizeof(int)); uAi = (int *)xmalloc(sizeof(int) * (size_t)nt);
    if (!uAp || !uAi) { rc = -2; goto fail; }
    for (k = 0; k < nt; k++) uAp[T[k].col + 1]++;
    for (j = 0; j < N; j++) uAp[j + 1] += uAp[j];
    work = (int *)xmalloc(sizeof(int) * (size_t)(N + 1));
    if (!work) { rc = -2; goto fail; }
    memcpy(work, uAp, sizeof(int) * (size_t)(N + 1));
    for (k = 0; k < nt; k++) uAi[work[T[k].col]++] = T[k].row;
    free(work); work = 0;
    if ((rc = rldl_order_min_degree(N, uAp, uAi, s->perm))) goto fail;
    for (k = 0; k < N; k++) s->pinv[s->perm[k]] = k;
  }

  /* ---- permute: (i,j) -> (min, max) of (pinv i, pinv j); CSC with ascending rows ---- */
  for (k = 0; k < nt; k++) {
    int a = s->pinv[T[k].row], b = s->pinv[T[k].col];
    T[k].row = a < b ? a : b; T[k].col = a < b ? b : a;
  }
  qsort(T, (size_t)nt, sizeof(trip), cmp_trip);
  s->Kp = (int *)xcalloc((size_t)N + 2, sizeof(int)); s->Ki = (int *)xmalloc(sizeof(int) * (size_t)nt);
  s->PtoK = (int *)xmalloc(sizeof(int) * (size_t)nnzP); s->Pisdiag = (unsigned char *)xcalloc((size_t)nnzP, 1);
  s->AtoK = (int *)xmalloc(sizeof(int) * (size_t)nnzA); s->rhotoK = (int *)xmalloc(sizeof(int) * (size_t)m);
  s->sigK = (int *)xmalloc(sizeof(int) * (size_t)nsig);
  if (!s->Kp || !s->Ki || !s->PtoK || !s->Pisdiag || !s->AtoK || !s->rhotoK || !s->sigK) { rc = -2; goto fail; }
  for (k = 0; k < nt; k++) {
    s->Kp[T[k].col + 1]++;
    s->Ki[k] = T[k].row;
    switch (T[k].kind) {
      case SRC_P: s->PtoK[T[k].idx] = k; break;
      case SRC_A: s->AtoK[T[k].idx] = k; break;
      case SRC_RHO: s->rhotoK[T[k].idx] = k; break;
      default: s->sigK[T[k].idx] = k; break;
    }
  }
  for (j = 0; j < N; j++) s->Kp[j + 1] += s->Kp[j];
  for (j = 0; j < n; j++)
    for (p = s->Pp[j]; p < s->Pp[j + 1]; p++) if (s->Pi[p] == j) s->Pisdiag[p] = 1;

  /* ---- elimination tree, column counts (Liu; the QDLDL_etree contract) ---- */
  s->etree = (int *)xmalloc(sizeof(int) * (size_t)N); s->Lnz = (int *)xcalloc((size_t)N, sizeof(int));
  s->Lp = (int *)xcalloc((size_t)N + 1, sizeof(int));
  work = (int *)xmalloc(sizeof(int) * (size_t)(N + 1));
  if (!s->etree || !s->Lnz || !s->Lp || !work) { rc = -2; goto fail; }
  for (i = 0; i < N; i++) { s->etree[i] = -1; work[i] = -1; }
  for (j = 0; j < N; j++) {
    work[j] = j;
    for (p = s->Kp[j]; p < s->Kp[j + 1]; p++) {
      i = s->Ki[p];
      while (work[i] != j) {
        if (s->etree[i] == -1) s->etree[i] = j;
        s->Lnz[i]++;
        work[i] = j;
        i = s->etree[i];
      }
    }
  }
  for (j = 0; j < N; j++) s->Lp[j + 1] = s->Lp[j] + s->Lnz[j];
  s->nnzL = s->Lp[N];
  { /* tree height (longest root path), reported by the bench */
    int h = 0;
    for (i = N - 1; i >= 0; i--) { /* parents have larger index: depth[i] = depth[parent]+1 */
      work[i] = s->etree[i] < 0 ? 1 : work[s->etree[i]] + 1;
      if (work[i] > h) h = work[i];
    }
    s->etree_height = h;
  }

  /* ---- pattern of L: row k = etree reach of column k of K; append k to each reached column ---- */
  s->Li = (int *)xmalloc(sizeof(int) * (size_t)s->nnzL);
  s->Rp = (int *)xcalloc((size_t)N + 1, sizeof(int)); s->Rj = (int *)xmalloc(sizeof(int) * (size_t)s->nnzL);
  s->Rpos = (int *)xmalloc(sizeof(int) * (size_t)s->nnzL);
  mark = (int *)xmalloc(sizeof(int) * (size_t)(N + 1)); stack = (int *)xmalloc(sizeof(int) * (size_t)(N + 1));
  fill = (int *)xmalloc(sizeof(int) * (size_t)(N + 1));
  if (!s->Li || !s->Rp || !s->Rj || !s->Rpos || !mark || !stack || !fill) { rc = -2; goto fail; }
  for (i = 0; i < N; i++) { mark[i] = -1; fill[i] = s->Lp[i]; }
  {
    int rptr = 0;
    for (k = 0; k < N; k++) {
      int cnt = 0, a, b;
      s->Rp[k] = rptr;
      mark[k] = k;
      for (p = s->Kp[k]; p < s->Kp[k + 1]; p++) {
        i = s->Ki[p];
        while (i != -1 && i < k && mark[i] != k) { mark[i] = k; stack[cnt++] = i; i = s->etree[i]; }
      }
      /* ascending column order for the row (insertion sort; rows are short) */
      for (a = 1; a < cnt; a++) {
        int v = stack[a];
        for (b = a - 1; b >= 0 && stack[b] > v; b--) stack[b + 1] = stack[b];
        stack[b + 1] = v;
      }
      for (a = 0; a < cnt; a++) {
        int c = stack[a], slot = fill[c]++;
        s->Li[slot] = k;
        s->Rj[rptr] = c; s->Rpos[rptr] = slot; rptr++;
      }
    }
    s->Rp[N] = rptr;
    if (rptr != s->nnzL) { rc = -1; goto fail; }
  }

  /* ---- KKT slot -> factor-workspace slot ---- */
  s->KtoW = (int *)xmalloc(sizeof(int) * (size_t)nt);
  if (!s->KtoW) { rc = -2; goto fail; }
  for (j = 0; j < N; j++)
    for (p = s->Kp[j]; p < s->Kp[j + 1]; p++) {
      i = s->Ki[p];
      if (i == j) s->KtoW[p] = s->nnzL + j;
      else {
        int q = find_in_col(s->Li, s->Lp[i], s->Lp[i + 1], j);
        if (q < 0) { rc = -1; goto fail; }
        s->KtoW[p] = q;
      }
    }

  /* ---- right-looking update lists ---- */
  s->Up = (long long *)xcalloc((size_t)N + 1, sizeof(long long));
  if (!s->Up) { rc = -2; goto fail; }
  for (j = 0; j < N; j++) { long long c = s->Lnz[j]; s->Up[j + 1] = s->Up[j] + c * (c + 1) / 2; }
  s->npairs = s->Up[N];
  s->Udst = (int *)xmalloc(sizeof(int) * (size_t)s->npairs);
  s->Uab = (unsigned int *)xmalloc(sizeof(unsigned int) * (size_t)s->npairs);
  if (!s->Udst || !s->Uab) { rc = -2; goto fail; }
  for (j = 0; j < N; j++) {
    int c = s->Lnz[j], a, b, base = s->Lp[j];
    long long t = s->Up[j];
    if (c > 65535) { rc = -1; goto fail; }
    for (a = 0; a < c; a++) {
      int ra = s->Li[base + a];
      for (b = 0; b <= a; b++, t++) {
        int rb = s->Li[base + b];
        s->Uab[t] = (unsigned int)a | ((unsigned int)b << 16);
        if (a == b) s->Udst[t] = s->nnzL + ra;
        else {
          int q = find_in_col(s->Li, s->Lp[rb], s->Lp[rb + 1], ra);
          if (q < 0) { rc = -1; goto fail; }
          s->Udst[t] = q;
        }
      }
    }
  }

  free(T); free(uAp); free(uAi); free(work); free(mark); free(stack); free(fill);
  T = 0; uAp = uAi = work = mark = stack = fill = 0;
  if ((rc = rldl_plan_build(s))) goto fail;
  *out = s;
  return 0;
fail:
  free(T); free(uAp); free(uAi); free(work); free(mark); free(stack); free(fill);
  rldl_symbolic_free(s);
  return rc ? rc : -2;
}
