/* rldl_internal.h -- private host structs shared by rldl_backend.c, rldl_admm.c, rldl_recursive.c */
#ifndef RLDL_INTERNAL_H
#define RLDL_INTERNAL_H

#include "../../include/osqp_rldl_hip.h"
#include "rldl_device.h"
#include "rldl_symbolic.h"

struct rldl_batch {
  c_int batch;
  rldl_symbolic *sym;      /* host symbolic analysis (owned) */
  rldl_dev_sym dsym;       /* device copy of the index arrays */
  rldl_dev_num num;        /* device numeric state */
  void *stream;            /* hipStream_t (0 = default stream) */
  int stream_owned;
  void *ev0, *ev1;         /* hipEvent_t pair for rldl_batch_time_solve */
  int *status_host;        /* [batch] */
  /* stage-recursive strategy (rldl_recursive.c) */
  rldl_stage_dims stage;
  int recursive;
  void *rec;               /* rldl_rec_state* */
  int *pv_tiD;             /* host, stage handles with product tiles: first Ti entry of every diagonal block's tile [nb + 1] */
  /* host copies of the product tri-solve's tables, for step programs over a prefix of the blocks (single-store horizon handles):
   * group descriptors in tile order [pv_ngrp][12], tile info [4 per tile], tile id of (block, kind) [2 nb] */
  int *pv_grp, pv_ngrp, pv_ntiles_h, *pv_tinfo_h, *pv_blk_h;
};

/* batched ADMM workspace (rldl_admm.c) */
struct osqp_batch {
  c_int batch, n, m, nnzP, nnzA;
  OSQPBatchSettings st;
  rldl_batch *ls;
  rldl_batch *pls;               /* polish = 1 backend on the same patterns (delta-regularised reduced KKT), or NULL */
  rldl_dev_admm W;
  double *Px, *Ax, *q, *l, *u;   /* owned device copies of the problem data (osqp.c:106-114) */
  void *stream;
  void *ev0, *ev1;
  void *evn[2];                  /* events behind the asynchronous reads of the active-instance counter */
  int *h_nact;                   /* [2][RLDL_NACT_SLOTS] pinned */
  int *h_tmp_i;                  /* [batch] host scratch */
  double *h_tmp_d;               /* [batch] host scratch */
  int loop_pending;              /* a solve loop was enqueued and its event pair not read yet */
  int refactor_pending;          /* osqp_batch_update_P_A_async: the verdict of the refactorisation has not been read yet */
  int *d_bounds;                 /* [2] device words of k_check_bounds: [0] verdict of the update being enqueued, [1] sticky until osqp_batch_wait */
  int bounds_pending;            /* osqp_batch_update_bounds_async: the sticky verdict has not been read yet */
  float last_loop_ms;
  c_int last_loop_launches;     /* ADMM iterations run by the last solve loop ... */
  c_int last_loop_groups;       /* ... in this many launch groups (one kernel launch each on the arrowhead path) */
};

int rldl_device_available(void);
void rldl_batch_enable_stage(rldl_batch *h, const rldl_stage_dims *dims);   /* rldl_recursive.c */
void rldl_stage_maps_free(rldl_batch *h);
/* step program of the product tri-solve over the first nb_act blocks only (device array in *d_prog, owned by the caller; its
 * step count in *nsteps): forward over the tiles of those blocks, backward over the same tiles in reverse; the coupling tile
 * from the last live block to the first dead one is left out.  0 ok, 1 not available. */
int rldl_stage_prog_prefix(const rldl_batch *h, int nb_act, int **d_prog, int *nsteps);
void osqp_batch_reset_info(osqp_batch *w);                                    /* rldl_admm.c: auxil.c:628-645 */
c_int osqp_batch_bounds_ok(osqp_batch *w, c_int count, const c_float *d_l, const c_float *d_u);   /* rldl_admm.c: 1 when l <= u everywhere */
c_int rldl_batch_update_matrices_async(rldl_batch *h, const c_float *d_Px, const c_float *d_Ax, c_float *keep_Px,
                                       c_float *keep_Ax, int *d_status_reset, int *d_rho_updates_reset);   /* rldl_backend.c */
c_int rldl_batch_check_status(rldl_batch *h); /* sync + qdldl_interface.c:80-92 verdict: 0 ok, 1 failed */

#endif
