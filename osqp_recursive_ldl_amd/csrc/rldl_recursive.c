/*
 * rldl_recursive.c -- stage-recursive factorisation strategy for MPC-structured KKT matrices.
 *
 * What the reference does (src/recursive_ldl.c): LDL_factorize_recursive :1139-1318 factorises the
 * stage-interleaved KKT  [Q0+sI, C0, Q1+sI, C1, ..., QN+sI, CN]  block by block with the closed-form
 * permutation of compute_permutations :1350-1362, caches the per-stage pivots (X_even[], :1206/:1252)
 * and LDL_update_from_pivot :946-1110 restarts the recursion at a stage instead of refactorising.
 *
 * What this build does (round 1): the SAME permutation (bit-exact integers) is handed to the batched
 * sparse backend, whose right-looking device factorisation can RESTART at any column: the columns of
 * stages < first_stage keep their L and D, their contributions to the trailing part are replayed
 * from the stored factor, and only stages >= first_stage are eliminated again.  This reproduces the
 * reference's "continue from the saved pivot" semantics without the X_even cache.  Dense per-stage
 * block kernels (the MFMA candidate of SURVEY.md 8a-15) are the planned next step.
 */
#include <stdlib.h>

#include "../../include/osqp_rldl_hip.h"
#include "rldl_device.h"
#include "rldl_internal.h"
#include "rldl_symbolic.h"

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
  (*hp)->stage = *dims;
  (*hp)->recursive = 1;
  return 0;
}

c_int rldl_batch_update_from_stage(rldl_batch *h, c_int first_stage, const c_float *d_Px, const c_float *d_Ax,
                                   const c_float *d_rho_vec) {
  c_int col0;
  if (!h || !h->recursive || first_stage < 0 || first_stage > h->stage.N) return 1;
  col0 = stage_first_col(&h->stage, first_stage);
  if (rldl_launch_kkt_assemble(&h->dsym, &h->num, d_Px, d_Ax, d_rho_vec, 0, 0, h->stream)) return 1;
  if (rldl_launch_factor_from(&h->dsym, &h->num, (int)col0, h->stream)) return 1;
  return rldl_batch_check_status(h);
}
