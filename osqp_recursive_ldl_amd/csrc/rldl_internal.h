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
};

int rldl_device_available(void);
c_int rldl_batch_check_status(rldl_batch *h); /* sync + qdldl_interface.c:80-92 verdict: 0 ok, 1 failed */

#endif
