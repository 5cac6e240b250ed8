/*
 * rldl_recursive.c -- stage-recursive factorisation strategy for MPC-structured KKT matrices.
 *
 * What the reference does (src/recursive_ldl.c): LDL_factorize_recursive :1139-1318 factorises the
 * stage-interleaved KKT  [Q0+sI, C0, Q1+sI, C1, ..., QN+sI, CN]  block by block with the closed-form
 * permutation of compute_permutations :1350-1362, caches the per-stage pivots (X_even[], :1206/:1252)
 * and LDL_update_from_pivot :946-1110 restarts the recursion at a stage instead of refactorising.
 *
 * What this build does: the SAME permutation (bit-exact integers) goes to the batched backend together with a dense
 * stage-block view of the permuted matrix and of its factor (build_stage_maps / build_solve_tiles below).  The numeric
 * factorisation (k_stage_factor_r) and the tri-solve (stage_tri_solve) then run the block recursion itself -- Schur
 * complement of the previous block, dense LDL' of the block, coupling block -- on one wavefront per instance, and a
 * restart keeps the blocks before the first modified stage: the reference's "continue from the saved pivot" without the
 * X_even cache (L(b, b-1) and D_{b-1} are read back from the factor).  Patterns that do not qualify (blocks wider than
 * 32, not block tridiagonal) stay on the generic sparse kernels, which restart at a column.
 */
#include <hip/hip_runtime_api.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/osqp_rldl_hip.h"
#include "rldl_device.h"
#include "rldl_internal.h"
#include "rldl_symbolic.h"

/* ---------------------------------------------------------------------------------------------------
 * Stage blocks of the interleaved order  Q0, C0, Q1, C1, ..., Q_N, C_N  (compute_permutations,
 * src/recursive_ldl.c:1350-1362): sizes nu, (ny+nx, nx+nu) x (N-1), ny+nx, nx, nt.  The permuted KKT matrix and
 * its factor are block tridiagonal over these blocks; build_stage_maps lists, per block, where every KKT value and
 * every L entry sits in a dense tile, which is all k_stage_factor needs (csrc/rldl_device.h: rldl_dev_stage).
 * Host copy of the block starts: h->rec = int[nb + 2] = { nb, bs[0..nb] }.
 * --------------------------------------------------------------------------------------------------- */
#define STAGE_BLOCK_MAX 32

static int *upload_ints(const int *src, size_t count) {
  int *d = 0;
  if (hipMalloc((void **)&d, sizeof(int) * (count ? count : 1)) != hipSuccess) return 0;
  if (count && hipMemcpy(d, src, sizeof(int) * count, hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(d); return 0; }
  return d;
}

void rldl_stage_maps_free(rldl_batch *h) {
  rldl_dev_stage *G;
  if (!h) return;
  G = &h->dsym.stage;
#define FRI(p) if (p) (void)hipFree((void *)(p))
  FRI(G->bs); FRI(G->kd_ptr); FRI(G->kd_src); FRI(G->kd_pos); FRI(G->kc_ptr); FRI(G->kc_src); FRI(G->kc_pos);
  FRI(G->ld_ptr); FRI(G->ld_slot); FRI(G->ld_pos); FRI(G->lc_ptr); FRI(G->lc_slot); FRI(G->lc_pos);
  FRI(G->sv_pk); FRI(G->sv_prog);
#undef FRI
  memset(G, 0, sizeof(*G));
  free(h->rec); h->rec = 0;
}

/* Tables of the block tri-solve (k_plan_solve<.., true> and friends): per stage block the L entries of the diagonal block
 * and of the coupling block below it as (tile position << 16 | factor slot), sorted by slot so that consecutive lanes read
 * neighbouring factor entries, plus one table entry per (direction, block) naming the block and its two tiles:
 * forward  C = L(b, b-1), D = L_bb for b = 0..nb-1, backward  C = L(b+1, b), D = L_bb for b = nb-1..0 (empty ranges allowed). */
static int cmp_slot(const void *x, const void *y) {
  const unsigned a = *(const unsigned *)x & 0xffffu, b = *(const unsigned *)y & 0xffffu;
  return a < b ? -1 : (a > b ? 1 : 0);
}
#define SV_TILE_WORDS 384                                         /* SV_PF * 64 in rldl_kernels.hip */
static void build_solve_tiles(rldl_batch *h, const int *bs, int nb, int ld, const int *dptr, const int *dslot, const int *dpos, int dtot,
                              const int *cptr, const int *cslot, const int *cpos, int ctot) {
  static const int widths[] = {8, 16, 22, 24, 32};               /* instantiated block-width bounds SM of the solve kernels */
  rldl_dev_stage *G = &h->dsym.stage;
  unsigned *pk = 0, pad, tmp[SV_TILE_WORDS];
  int *prog = 0, b, e, k, lds = 0, sm = 0, rc = 0, rd = 0;
  const int zs = h->dsym.ldF - 1;                                /* spare slot of the factor row, always 0.0 */
  (void)dtot; (void)ctot;
  for (k = 0; k < 5 && !lds; k++) if (G->smax <= widths[k]) { sm = widths[k]; lds = sm + 1; }   /* tile rows SM + 1 wide (odd) */
  if (!lds || zs >= 65536) return;
  for (b = 0; b < nb; b++)
    if (dptr[b + 1] - dptr[b] > SV_TILE_WORDS || cptr[b + 1] - cptr[b] > SV_TILE_WORDS) return;
  pk = (unsigned *)malloc(sizeof(unsigned) * (size_t)SV_TILE_WORDS * (size_t)(2 * nb));
  prog = (int *)calloc((size_t)8 * (size_t)(2 * nb + 4), sizeof(int));
  if (!pk || !prog) goto out;
  pad = ((unsigned)(sm * 8) << 16) | (unsigned)zs;               /* pad column of tile row 0 <- 0.0 */
  for (b = 0; b < nb; b++) {                                       /* tile 2b = L(b+1, b), tile 2b + 1 = L_bb */
    unsigned *tcw = pk + (size_t)SV_TILE_WORDS * (size_t)(2 * b), *tdw = tcw + SV_TILE_WORDS;
    int nc = cptr[b + 1] - cptr[b], nd = dptr[b + 1] - dptr[b];
    for (e = 0; e < nc; e++) tcw[e] = ((unsigned)(((cpos[cptr[b] + e] / ld) * lds + cpos[cptr[b] + e] % ld) * 8) << 16) | (unsigned)cslot[cptr[b] + e];
    for (e = 0; e < nd; e++) tdw[e] = ((unsigned)(((dpos[dptr[b] + e] / ld) * lds + dpos[dptr[b] + e] % ld) * 8) << 16) | (unsigned)dslot[dptr[b] + e];
    qsort(tcw, (size_t)nc, sizeof(unsigned), cmp_slot);
    qsort(tdw, (size_t)nd, sizeof(unsigned), cmp_slot);
    for (e = nc; e < SV_TILE_WORDS; e++) tcw[e] = pad;
    for (e = nd; e < SV_TILE_WORDS; e++) tdw[e] = pad;
    if ((nc + 63) / 64 > rc) rc = (nc + 63) / 64;
    if ((nd + 63) / 64 > rd) rd = (nd + 63) / 64;
    /* entry i (slot order) belongs to lane i % 64, round i / 64; stored lane-major: word [lane][round] */
    for (k = 0; k < 2; k++) {
      unsigned *w = k ? tdw : tcw;
      for (e = 0; e < SV_TILE_WORDS; e++) tmp[(e % 64) * (SV_TILE_WORDS / 64) + e / 64] = w[e];
      memcpy(w, tmp, sizeof(tmp));
    }
  }
  /* entry k: { c0, s, o0, coupling tile or -1, diagonal tile or -1 }; forward blocks 0..nb-1, backward blocks nb-1..0 */
  for (k = 0; k < 2 * nb + 4; k++) {
    int *q = prog + 8 * k;
    const int fwd = k < nb;
    q[3] = q[4] = -1;
    if (k >= 2 * nb) continue;
    b = fwd ? k : 2 * nb - 1 - k;
    q[0] = bs[b]; q[1] = bs[b + 1] - bs[b];
    if (fwd && b > 0) { q[2] = bs[b - 1]; if (cptr[b - 1] < cptr[b]) q[3] = 2 * (b - 1); }
    if (!fwd && b + 1 < nb) { q[2] = bs[b + 1]; if (cptr[b] < cptr[b + 1]) q[3] = 2 * b; }
    if (dptr[b] < dptr[b + 1]) q[4] = 2 * b + 1;
  }
  G->sv_pk = (const unsigned *)upload_ints((const int *)pk, (size_t)SV_TILE_WORDS * (size_t)(2 * nb));
  G->sv_prog = upload_ints(prog, (size_t)8 * (size_t)(2 * nb + 4));
  if (G->sv_pk && G->sv_prog) { G->sv_ok = 1; G->sv_ld = lds; G->sv_coff = rc | (rd << 8); G->sv_ntiles = 2 * nb; }
out:
  free(pk); free(prog);
}

/* 0: maps built and uploaded (h->dsym.stage.nb > 0); 1: the pattern does not qualify (generic kernels stay in use) */
static int build_stage_maps(rldl_batch *h) {
  const rldl_symbolic *s = h->sym;
  const rldl_stage_dims *d = &h->stage;
  const int N = s->N, nb = 2 * (int)d->N + 2;
  int *bs = 0, *blk = 0, *cnt = 0, *ptr[4] = {0, 0, 0, 0}, *a[4] = {0, 0, 0, 0}, *b[4] = {0, 0, 0, 0}, *fill = 0;
  int i, j, k, p, smax = 0, ld, rc = 1, tot[4] = {0, 0, 0, 0};
  rldl_dev_stage G;
  memset(&G, 0, sizeof(G));
  bs = (int *)malloc(sizeof(int) * (size_t)(nb + 2));
  blk = (int *)malloc(sizeof(int) * (size_t)(N + 1));
  cnt = (int *)calloc((size_t)4 * (size_t)(nb + 1), sizeof(int));
  fill = (int *)calloc((size_t)4 * (size_t)(nb + 1), sizeof(int));
  if (!bs || !blk || !cnt || !fill) goto out;
  bs[0] = 0; k = 1;
  bs[1] = (int)d->nu;                                                                 /* Q0 */
  for (i = 1; i < d->N; i++) { bs[k + 1] = bs[k] + (int)(d->ny + d->nx); k++; bs[k + 1] = bs[k] + (int)(d->nx + d->nu); k++; }
  bs[k + 1] = bs[k] + (int)(d->ny + d->nx); k++;                                      /* C_{N-1} */
  bs[k + 1] = bs[k] + (int)d->nx; k++;                                                /* Q_N */
  bs[k + 1] = bs[k] + (int)d->nt; k++;                                                /* C_N */
  if (k != nb || bs[nb] != N) goto out;
  for (k = 0; k < nb; k++) {
    if (bs[k + 1] - bs[k] > smax) smax = bs[k + 1] - bs[k];
    if (bs[k + 1] <= bs[k]) goto out;                                                 /* empty blocks are not handled */
    for (i = bs[k]; i < bs[k + 1]; i++) blk[i] = k;
  }
  if (smax > STAGE_BLOCK_MAX) goto out;
  ld = ((smax + 7) & ~7) + 1;                                                         /* register-kernel bound SM = 8/16/24/32, tiles SM + 1 wide (odd) */
  /* pass 1: classify and count; lists: 0 = K diagonal, 1 = K coupling, 2 = L diagonal, 3 = L coupling */
  for (j = 0; j < N; j++)
    for (p = s->Kp[j]; p < s->Kp[j + 1]; p++) {
      i = s->Ki[p];
      if (blk[i] == blk[j]) cnt[0 * (nb + 1) + blk[j]]++;
      else if (blk[i] + 1 == blk[j]) cnt[1 * (nb + 1) + blk[i]]++;
      else goto out;                                                                  /* not block tridiagonal */
    }
  for (j = 0; j < N; j++)
    for (p = s->Lp[j]; p < s->Lp[j + 1]; p++) {
      i = s->Li[p];
      if (blk[i] == blk[j]) cnt[2 * (nb + 1) + blk[j]]++;
      else if (blk[i] == blk[j] + 1) cnt[3 * (nb + 1) + blk[j]]++;
      else goto out;
    }
  for (k = 0; k < 4; k++) {
    ptr[k] = (int *)malloc(sizeof(int) * (size_t)(nb + 1));
    if (!ptr[k]) goto out;
    ptr[k][0] = 0;
    for (i = 0; i < nb; i++) ptr[k][i + 1] = ptr[k][i] + cnt[k * (nb + 1) + i];
    tot[k] = ptr[k][nb];
    a[k] = (int *)malloc(sizeof(int) * (size_t)(tot[k] + 1));
    b[k] = (int *)malloc(sizeof(int) * (size_t)(tot[k] + 1));
    if (!a[k] || !b[k]) goto out;
  }
  /* pass 2: fill.  Tile positions are row * ld + col with the LATER index as the row (lower triangle / block below) */
  for (j = 0; j < N; j++)
    for (p = s->Kp[j]; p < s->Kp[j + 1]; p++) {
      i = s->Ki[p];                                                                   /* i <= j: entry K(j, i) of the lower part */
      if (blk[i] == blk[j]) { k = 0; }
      else k = 1;
      {
        const int bb = k == 0 ? blk[j] : blk[i];
        const int e = ptr[k][bb] + fill[k * (nb + 1) + bb]++;
        a[k][e] = p;
        b[k][e] = (j - bs[blk[j]]) * ld + (i - bs[blk[i]]);
      }
    }
  for (j = 0; j < N; j++)
    for (p = s->Lp[j]; p < s->Lp[j + 1]; p++) {
      i = s->Li[p];                                                                   /* i > j: entry L(i, j) */
      k = blk[i] == blk[j] ? 2 : 3;
      {
        const int bb = blk[j];
        const int e = ptr[k][bb] + fill[k * (nb + 1) + bb]++;
        a[k][e] = s->LtoS[p];
        b[k][e] = (i - bs[blk[i]]) * ld + (j - bs[blk[j]]);
      }
    }
  G.nb = nb; G.ld = ld; G.smax = smax;
  G.bs = upload_ints(bs, (size_t)nb + 1);
  G.kd_ptr = upload_ints(ptr[0], (size_t)nb + 1); G.kd_src = upload_ints(a[0], (size_t)tot[0]); G.kd_pos = upload_ints(b[0], (size_t)tot[0]);
  G.kc_ptr = upload_ints(ptr[1], (size_t)nb + 1); G.kc_src = upload_ints(a[1], (size_t)tot[1]); G.kc_pos = upload_ints(b[1], (size_t)tot[1]);
  G.ld_ptr = upload_ints(ptr[2], (size_t)nb + 1); G.ld_slot = upload_ints(a[2], (size_t)tot[2]); G.ld_pos = upload_ints(b[2], (size_t)tot[2]);
  G.lc_ptr = upload_ints(ptr[3], (size_t)nb + 1); G.lc_slot = upload_ints(a[3], (size_t)tot[3]); G.lc_pos = upload_ints(b[3], (size_t)tot[3]);
  h->dsym.stage = G;
  if (!G.bs || !G.kd_ptr || !G.kd_src || !G.kd_pos || !G.kc_ptr || !G.kc_src || !G.kc_pos || !G.ld_ptr || !G.ld_slot || !G.ld_pos ||
      !G.lc_ptr || !G.lc_slot || !G.lc_pos) { rldl_stage_maps_free(h); goto out; }
  build_solve_tiles(h, bs, nb, ld, ptr[2], a[2], b[2], tot[2], ptr[3], a[3], b[3], tot[3]);   /* optional: sv_ok stays 0 on failure */
  h->rec = malloc(sizeof(int) * (size_t)(nb + 2));
  if (!h->rec) { rldl_stage_maps_free(h); goto out; }
  ((int *)h->rec)[0] = nb;
  memcpy((int *)h->rec + 1, bs, sizeof(int) * (size_t)(nb + 1));
  rc = 0;
out:
  free(bs); free(blk); free(cnt); free(fill);
  for (k = 0; k < 4; k++) { free(ptr[k]); free(a[k]); free(b[k]); }
  return rc;
}

/* Mark a handle as stage-structured: enables restart-from-stage and, when the pattern qualifies, the dense
 * stage-block factorisation for every later numeric factorisation of the handle. */
void rldl_batch_enable_stage(rldl_batch *h, const rldl_stage_dims *dims) {
  h->stage = *dims;
  h->recursive = 1;
  if (!getenv("RLDL_NO_STAGE_FACTOR")) (void)build_stage_maps(h);
}

/* first column (in the permuted matrix) of cost block Q_k, k = 0..N */
static c_int stage_first_col(const rldl_stage_dims *d, c_int k) {
  if (k <= 0) return 0;
  return d->nu + (d->nx + d->ny) + (k - 1) * (2 * d->nx + d->nu + d->ny);
}

c_int rldl_batch_init_recursive(rldl_batch **hp, c_int batch, const rldl_stage_dims *dims, const csc *P, const csc *A,
                                const c_float *d_Px, const c_float *d_Ax, c_float sigma, const c_float *d_rho_vec,
                                void *stream) {
  c_int nvar, ncon, rc, *perm;
  if (hp) *hp = 0;
  if (!dims || !P || !A || dims->N < 1) return RLDL_LINSYS_SOLVER_INIT_ERROR;
  nvar = dims->N * (dims->nx + dims->nu);
  ncon = dims->N * (dims->nx + dims->ny) + dims->nt;
  if (P->n != nvar || A->m != ncon) return RLDL_LINSYS_SOLVER_INIT_ERROR;
  perm = (c_int *)malloc(sizeof(c_int) * (size_t)(nvar + ncon));
  if (!perm) return RLDL_MEM_ALLOC_ERROR;
  rldl_stage_permutation(dims->N, dims->nx, dims->nu, dims->ny, dims->nt, perm);
  rc = rldl_batch_init(hp, batch, P, A, d_Px, d_Ax, sigma, d_rho_vec, 0, perm, stream);
  free(perm);
  if (rc) return rc;
  rldl_batch_enable_stage(*hp, dims);
  if ((*hp)->dsym.stage.nb > 0) {                               /* the factor of record comes from the stage kernel */
    if (rldl_launch_stage_factor(&(*hp)->dsym, &(*hp)->num, 0, 0, (*hp)->stream) || rldl_batch_check_status(*hp)) {
      rldl_batch_free(*hp); *hp = 0;
      return RLDL_NONCVX_ERROR;
    }
  }
  return 0;
}

c_int rldl_batch_update_from_stage(rldl_batch *h, c_int first_stage, const c_float *d_Px, const c_float *d_Ax,
                                   const c_float *d_rho_vec) {
  c_int col0;
  if (!h || !h->recursive || first_stage < 0 || first_stage > h->stage.N) return 1;
  col0 = stage_first_col(&h->stage, first_stage);
  if (rldl_launch_kkt_assemble(&h->dsym, &h->num, d_Px, d_Ax, d_rho_vec, 0, 0, h->stream)) return 1;
  if (h->dsym.stage.nb > 0 && h->rec) {                         /* dense stage blocks: restart at the block of Q_first_stage */
    const int *bs = (const int *)h->rec + 1, nb = ((const int *)h->rec)[0];
    int b = 0;
    while (b < nb && bs[b] < col0) b++;
    if (b >= nb || bs[b] != col0) return 1;
    if (rldl_launch_stage_factor(&h->dsym, &h->num, 0, b, h->stream)) return 1;
  } else if (rldl_launch_factor_from(&h->dsym, &h->num, (int)col0, h->stream)) return 1;
  return rldl_batch_check_status(h);
}

/* =====================================================================================
 * Assembling the big P (upper triangular) and A from the seven stage blocks: the layout of
 * setup_AP_matrices (src/recursive_ldl.c:1873-1970).  Variables [u0 | x1,u1 | ... | x_{N-1},u_{N-1} | x_N],
 * row block k = [ny inequality rows ; nx dynamics rows], terminal block of nt rows.  Column block k >= 1
 * holds Aij (coupling into the dynamics rows of row block k-1) stacked over Ai (row block k); the last
 * column block (x_N, nx columns) holds the first nx columns of Aij over AN.
 * Besides the matrices the function reports, per stored value, which stage block it came from, so a caller
 * can rebuild the value arrays of a whole batch from per-stage data (update_AP_matrices, :1675-1778).
 * ===================================================================================== */
static csc *csc_new(c_int m, c_int n, c_int nz) {
  csc *M = (csc *)calloc(1, sizeof(csc));
  if (!M) return 0;
  M->m = m; M->n = n; M->nzmax = nz > 0 ? nz : 1; M->nz = -1;
  M->p = (c_int *)calloc((size_t)n + 1, sizeof(c_int));
  M->i = (c_int *)malloc(sizeof(c_int) * (size_t)M->nzmax);
  M->x = (c_float *)malloc(sizeof(c_float) * (size_t)M->nzmax);
  if (!M->p || !M->i || !M->x) { free(M->p); free(M->i); free(M->x); free(M); return 0; }
  return M;
}

void rldl_csc_free(csc *M) {
  if (!M) return;
  free(M->p); free(M->i); free(M->x); free(M);
}

static void put_col(csc *dst, c_int *nz, const csc *blk, c_int col, c_int row_off, c_int kind, c_int stage, c_int *src_kind,
                    c_int *src_stage, c_int *src_entry) {
  c_int j;
  for (j = blk->p[col]; j < blk->p[col + 1]; j++) {
    dst->i[*nz] = row_off + blk->i[j];
    dst->x[*nz] = blk->x[j];
    if (src_kind) { src_kind[*nz] = kind; src_stage[*nz] = stage; src_entry[*nz] = j; }
    (*nz)++;
  }
}

c_int rldl_setup_AP_matrices(const rldl_stage_dims *d, const csc *Q0, const csc *Qi, const csc *QN, const csc *A0, const csc *Ai,
                             const csc *Aij, const csc *AN, csc **P_out, csc **A_out, c_int *P_kind, c_int *P_stage,
                             c_int *P_entry, c_int *A_kind, c_int *A_stage, c_int *A_entry) {
  c_int N, nvar, ncon, nzP = 0, nzA = 0, col = 0, prow = 0, arow = 0, k, i, capP, capA;
  csc *P, *A;
  if (!d || !Q0 || !Qi || !QN || !A0 || !Ai || !Aij || !AN || !P_out || !A_out) return 1;
  N = d->N;
  if (N < 1 || Q0->n != d->nu || Qi->n != d->nx + d->nu || QN->n != d->nx || A0->m != d->nx + d->ny || A0->n != d->nu ||
      Ai->m != d->nx + d->ny || Ai->n != d->nx + d->nu || Aij->m != d->nx + d->ny || Aij->n != d->nx + d->nu ||
      AN->m != d->nt || AN->n != d->nx)
    return 1;
  nvar = N * (d->nx + d->nu); ncon = N * (d->nx + d->ny) + d->nt;
  capP = (N - 1) * Qi->p[Qi->n] + Q0->p[Q0->n] + QN->p[QN->n];
  capA = N * (Ai->p[Ai->n] + Aij->p[Aij->n]) + A0->p[A0->n] + AN->p[AN->n];
  P = csc_new(nvar, nvar, capP); A = csc_new(ncon, nvar, capA);
  if (!P || !A) { rldl_csc_free(P); rldl_csc_free(A); return RLDL_MEM_ALLOC_ERROR; }
  /* kinds: P 0=Q0 1=Qi 2=QN ; A 0=A0 1=Ai 2=Aij 3=AN */
  for (i = 0; i < Q0->n; i++, col++) {
    P->p[col] = nzP; put_col(P, &nzP, Q0, i, 0, 0, 0, P_kind, P_stage, P_entry);
    A->p[col] = nzA; put_col(A, &nzA, A0, i, 0, 0, 0, A_kind, A_stage, A_entry);
  }
  prow = Q0->m;
  for (k = 1; k < N; k++) {
    for (i = 0; i < Qi->n; i++, col++) {
      P->p[col] = nzP; put_col(P, &nzP, Qi, i, prow, 1, k, P_kind, P_stage, P_entry);
      A->p[col] = nzA;
      put_col(A, &nzA, Aij, i, arow, 2, k, A_kind, A_stage, A_entry);
      put_col(A, &nzA, Ai, i, arow + Aij->m, 1, k, A_kind, A_stage, A_entry);
    }
    prow += Qi->m; arow += Aij->m;
  }
  for (i = 0; i < QN->n; i++, col++) {
    P->p[col] = nzP; put_col(P, &nzP, QN, i, prow, 2, N, P_kind, P_stage, P_entry);
    A->p[col] = nzA;
    put_col(A, &nzA, Aij, i, arow, 2, N, A_kind, A_stage, A_entry);
    put_col(A, &nzA, AN, i, arow + Aij->m, 3, N, A_kind, A_stage, A_entry);
  }
  P->p[col] = nzP; A->p[col] = nzA;
  *P_out = P; *A_out = A;
  return 0;
}

/* osqp_setup_recursive (src/recursive_ldl.c:2018-2230): the caller hands over the seven stage blocks of an MPC problem
 * (OSQPDataRLDL, include/recursive_ldl.h:17-50) instead of P and A.  The blocks are assembled on the host
 * (rldl_setup_AP_matrices), their values are replicated to every instance as the nominal problem, and the workspace is
 * built on the stage-interleaved permutation, so that osqp_batch_update_recursive can restart the factorisation at a
 * stage.  Per-instance values go in afterwards through osqp_batch_update_P_A / osqp_batch_update_recursive in the
 * value order of the assembled matrices (P_out / A_out, owned by the caller: rldl_csc_free). */
c_int osqp_batch_setup_recursive(osqp_batch **wp, c_int batch, const rldl_stage_dims *dims, const csc *Q0, const csc *Qi,
                                 const csc *QN, const csc *A0, const csc *Ai, const csc *Aij, const csc *AN, const c_float *d_q,
                                 const c_float *d_l, const c_float *d_u, const OSQPBatchSettings *settings, csc **P_out,
                                 csc **A_out, void *stream) {
  csc *P = 0, *A = 0;
  double *d_nom = 0, *d_Px = 0, *d_Ax = 0;
  c_int rc, nzP, nzA, *perm = 0;
  if (wp) *wp = 0;
  if (!wp || !dims || batch <= 0) return 1;
  rc = rldl_setup_AP_matrices(dims, Q0, Qi, QN, A0, Ai, Aij, AN, &P, &A, 0, 0, 0, 0, 0, 0);
  if (rc) return rc;
  nzP = P->p[P->n]; nzA = A->p[A->n];
  perm = (c_int *)malloc(sizeof(c_int) * (size_t)(P->n + A->m));
  rc = RLDL_MEM_ALLOC_ERROR;
  if (!perm) goto out;
  if (hipMalloc((void **)&d_nom, sizeof(double) * (size_t)(nzP > nzA ? nzP : nzA) + 8) != hipSuccess) goto out;
  if (hipMalloc((void **)&d_Px, sizeof(double) * (size_t)batch * (size_t)nzP + 8) != hipSuccess) goto out;
  if (hipMalloc((void **)&d_Ax, sizeof(double) * (size_t)batch * (size_t)nzA + 8) != hipSuccess) goto out;
  if (hipMemcpyAsync(d_nom, P->x, sizeof(double) * (size_t)nzP, hipMemcpyHostToDevice, (hipStream_t)stream) != hipSuccess) goto out;
  if (rldl_launch_bcast_rows((int)batch, (int)nzP, d_Px, d_nom, stream)) goto out;
  if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) goto out;
  if (hipMemcpyAsync(d_nom, A->x, sizeof(double) * (size_t)nzA, hipMemcpyHostToDevice, (hipStream_t)stream) != hipSuccess) goto out;
  if (rldl_launch_bcast_rows((int)batch, (int)nzA, d_Ax, d_nom, stream)) goto out;
  if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) goto out;
  rldl_stage_permutation(dims->N, dims->nx, dims->nu, dims->ny, dims->nt, perm);
  rc = osqp_batch_setup(wp, batch, P, A, d_Px, d_Ax, d_q, d_l, d_u, settings, perm, stream);
  if (!rc) rldl_batch_enable_stage((*wp)->ls, dims);
out:
  free(perm);
  if (d_nom) (void)hipFree(d_nom);
  if (d_Px) (void)hipFree(d_Px);
  if (d_Ax) (void)hipFree(d_Ax);
  if (!rc && P_out) *P_out = P; else rldl_csc_free(P);
  if (!rc && A_out) *A_out = A; else rldl_csc_free(A);
  return rc;
}
