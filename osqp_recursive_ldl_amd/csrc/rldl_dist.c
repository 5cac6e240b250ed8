/*
 * rldl_dist.c -- the multi-GPU leg of the path in plain C: one process per GPU, the batch shards by contiguous ranges with no
 * data-path collective, and ONE all-gather of the packed per-instance result records over RCCL collects the solutions
 * (the reference itself has no communication of any kind).
 *
 * librccl.so is opened with dlopen at the first call, the way the reference loads its optional MKL Pardiso backend
 * (lin_sys/lib_handler.c:7-49, lin_sys/direct/pardiso/pardiso_loader.c:63-102): the backend library has no link-time dependency on
 * RCCL and a single-GPU user never loads it.  The communicator's 128-byte unique id is created by rank 0 (osqp_dist_unique_id) and
 * handed to the other ranks by whatever launcher started them (a file, MPI, the environment): that exchange is not this library's.
 */
#include <dlfcn.h>
#include <hip/hip_runtime_api.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/osqp_rldl_hip.h"
#include "rldl_device.h"
#include "rldl_internal.h"

#define HIP_OK(e) ((e) == hipSuccess)

typedef struct { char internal[128]; } rccl_unique_id;             /* ncclUniqueId (rccl.h:40-43) */
typedef int (*fn_get_unique_id)(rccl_unique_id *);
typedef int (*fn_comm_init_rank)(void **, int, rccl_unique_id, int);
typedef int (*fn_comm_destroy)(void *);
typedef int (*fn_all_gather)(const void *, void *, size_t, int, void *, void *);
#define RCCL_DOUBLE 8                                              /* ncclFloat64 (rccl.h:467) */

static void *g_lib;
static fn_get_unique_id p_get_unique_id;
static fn_comm_init_rank p_comm_init_rank;
static fn_comm_destroy p_comm_destroy;
static fn_all_gather p_all_gather;

static int load_rccl(void) {
  if (g_lib) return 0;
  {
    const char *names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so", 0};
    int i;
    for (i = 0; names[i] && !g_lib; i++) g_lib = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
  }
  if (!g_lib) return 1;
  p_get_unique_id = (fn_get_unique_id)dlsym(g_lib, "ncclGetUniqueId");
  p_comm_init_rank = (fn_comm_init_rank)dlsym(g_lib, "ncclCommInitRank");
  p_comm_destroy = (fn_comm_destroy)dlsym(g_lib, "ncclCommDestroy");
  p_all_gather = (fn_all_gather)dlsym(g_lib, "ncclAllGather");
  if (!p_get_unique_id || !p_comm_init_rank || !p_comm_destroy || !p_all_gather) { dlclose(g_lib); g_lib = 0; return 1; }
  return 0;
}

struct osqp_dist {
  void *comm;
  c_int rank, nranks;
  void *stream;
  double *rec;                                                     /* device: this rank's packed records, grown on demand */
  size_t rec_cap;
};

c_int osqp_dist_unique_id(char id[128]) {
  rccl_unique_id u;
  if (!id) return 1;
  if (load_rccl()) return RLDL_LINSYS_SOLVER_LOAD_ERROR;
  if (p_get_unique_id(&u)) return 1;
  memcpy(id, u.internal, 128);
  return 0;
}

c_int osqp_dist_init(osqp_dist **dp, const char id[128], c_int rank, c_int nranks, void *stream) {
  osqp_dist *d;
  rccl_unique_id u;
  if (dp) *dp = 0;
  if (!dp || !id || nranks < 1 || rank < 0 || rank >= nranks) return 1;
  if (!rldl_device_available()) return RLDL_NO_DEVICE_ERROR;
  if (load_rccl()) return RLDL_LINSYS_SOLVER_LOAD_ERROR;
  d = (osqp_dist *)calloc(1, sizeof(osqp_dist));
  if (!d) return RLDL_MEM_ALLOC_ERROR;
  memcpy(u.internal, id, 128);
  if (p_comm_init_rank(&d->comm, (int)nranks, u, (int)rank)) { free(d); return 1; }
  d->rank = rank; d->nranks = nranks; d->stream = stream;
  *dp = d;
  return 0;
}

void osqp_dist_free(osqp_dist *d) {
  if (!d) return;
  if (d->comm) (void)p_comm_destroy(d->comm);
  if (d->rec) (void)hipFree(d->rec);
  free(d);
}

c_int osqp_dist_record_len(const osqp_batch *w) { return w ? w->n + w->m + 5 : 0; }

/* The collective of the path: this rank's result records [x | y | obj | pri_res | dua_res | iter | status] (osqp_batch_get's arrays,
 * n + m + 5 doubles per instance) are packed on the device and all-gathered over RCCL into d_out [nranks * batch][n + m + 5]
 * (every rank holds the same number of instances; ranks in order).  Enqueued on the stream given to osqp_dist_init, behind the
 * workspace's own stream; returns without waiting. */
c_int osqp_dist_gather_results(osqp_dist *d, osqp_batch *w, c_float *d_out) {
  size_t need;
  if (!d || !w || !d_out) return 1;
  need = (size_t)w->batch * (size_t)(w->n + w->m + 5);
  if (need > d->rec_cap) {
    if (d->rec) (void)hipFree(d->rec);
    d->rec = 0; d->rec_cap = 0;
    if (!HIP_OK(hipMalloc((void **)&d->rec, sizeof(double) * need))) return RLDL_MEM_ALLOC_ERROR;
    d->rec_cap = need;
  }
  if (w->stream != d->stream && !HIP_OK(hipStreamSynchronize((hipStream_t)w->stream))) return 1;
  if (rldl_launch_pack_results(&w->W, (int)w->n, (int)w->m, d->rec, d->stream)) return 1;
  return p_all_gather(d->rec, d_out, need, RCCL_DOUBLE, d->comm, d->stream) ? 1 : 0;
}
