/*
 * osqp_rldl_hip.h -- C ABI of the MI355X-native batched direct KKT backend for OSQP's ADMM loop.
 *
 * Drop-in boundary.  Every entry point is plain C: pointers and sizes only,
 * no torch / HIP types in any signature (a HIP stream is passed as `void *`).  Each declaration
 * names the reference interface it replaces as file:line under laperss/osqp-recursive-ldl.
 *
 * Three layers, all exported by libosqp_rldl_hip.so:
 *   1. legacy single-instance plugin  (host pointers; what `init_linsys_solver` would dispatch to)
 *   2. batched plugin                 (device pointers; the same five operations with a leading batch dim)
 *   3. batched ADMM driver            (device-resident mirror of osqp_setup/solve/update_*)
 *   4. stage-recursive factorisation  (MPC stage blocks; alternative `init` strategy of layer 2)
 */
#ifndef OSQP_RLDL_HIP_H
#define OSQP_RLDL_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

/* ---- scalar types: include/glob_opts.h:74-85 with the reference defaults DLONG=ON, DFLOAT=OFF ---- */
typedef long long c_int;
typedef double    c_float;

/* ---- compressed-column matrix: include/types.h:21-29 (identical layout) ---- */
#ifndef OSQP_TYPES_H
typedef struct {
  c_int    nzmax;
  c_int    m;
  c_int    n;
  c_int   *p;
  c_int   *i;
  c_float *x;
  c_int    nz;
} csc;
#endif

/* ---- solver enum: include/constants.h:36 has { QDLDL_SOLVER, MKL_PARDISO_SOLVER };
 *      recursive_ldl.h:55 adds QDLDL_HORIZON_SOLVER=10.  We register value 20. ---- */
#define HIP_LDL_SOLVER 20

/* ---- error / status codes: include/constants.h:18-51 ---- */
#define RLDL_LINSYS_SOLVER_LOAD_ERROR 3 /* OSQP_LINSYS_SOLVER_LOAD_ERROR: a library opened at run time is missing (librccl.so for the multi-GPU gather) */
#define RLDL_LINSYS_SOLVER_INIT_ERROR 4 /* OSQP_LINSYS_SOLVER_INIT_ERROR */
#define RLDL_NONCVX_ERROR 5             /* OSQP_NONCVX_ERROR */
#define RLDL_MEM_ALLOC_ERROR 6          /* OSQP_MEM_ALLOC_ERROR */
#define RLDL_NO_DEVICE_ERROR 100        /* no HIP device / kernel image: the product never falls back to CPU */
#define RLDL_DEVICE_ERROR 101           /* a HIP runtime call failed inside an enqueue-only entry point (not a verdict about the data) */

/* =====================================================================================
 * 1. Legacy single-instance plugin (prefix-compatible with `struct linsys_solver`,
 *    include/types.h:298-319; mirror of qdldl_solver, lin_sys/direct/qdldl/qdldl_interface.h:16-74)
 * ===================================================================================== */
typedef struct hipldl hipldl_solver;
struct hipldl {
  int type;                                             /* enum linsys_solver_type        types.h:299 */
  c_int (*solve)(hipldl_solver *self, c_float *b);      /*                               types.h:300-301 */
  void (*free)(hipldl_solver *self);                    /*                               types.h:304 */
  c_int (*update_matrices)(hipldl_solver *self, const csc *P, const csc *A); /*          types.h:308-310 */
  c_int (*update_rho_vec)(hipldl_solver *self, const c_float *rho_vec);     /*          types.h:312-313 */
  c_int nthreads;                                       /*                               types.h:317 */
  void *impl;                                           /* private: one-instance rldl_batch + staging */
};

/* replaces init_linsys_solver_qdldl           lin_sys/direct/qdldl/qdldl_interface.c:170-316 */
c_int init_linsys_solver_hipldl(hipldl_solver **sp, const csc *P, const csc *A, c_float sigma,
                                const c_float *rho_vec, c_int polish);
/* replaces solve_linsys_qdldl                 qdldl_interface.c:559-585 */
c_int solve_linsys_hipldl(hipldl_solver *s, c_float *b);
/* replaces update_linsys_solver_matrices_qdldl qdldl_interface.c:590-602 */
c_int update_linsys_solver_matrices_hipldl(hipldl_solver *s, const csc *P, const csc *A);
/* replaces update_linsys_solver_rho_vec_qdldl  qdldl_interface.c:605-619 */
c_int update_linsys_solver_rho_vec_hipldl(hipldl_solver *s, const c_float *rho_vec);
/* replaces free_linsys_solver_qdldl            qdldl_interface.c:17-43 */
void free_linsys_solver_hipldl(hipldl_solver *s);

/* =====================================================================================
 * 2. Batched plugin: `batch` independent instances sharing ONE sparsity pattern of P (upper
 *    triangular n x n) and A (m x n).  Value arrays are DEVICE pointers, instance-major:
 *      d_Px[batch][nnzP], d_Ax[batch][nnzA], d_rho_vec[batch][m], d_b[batch][n+m].
 *    Semantics per instance are exactly those of layer 1.
 * ===================================================================================== */
typedef struct rldl_batch rldl_batch;

/* init (qdldl_interface.c:170-316): symbolic phase on the host (form_KKT src/kkt.c:6-177, ordering,
 * csc_symperm src/cs.c:153-206, QDLDL_etree), numeric factor on the device.
 * `P`/`A` give the pattern (their ->x is ignored); `perm` (host, length n+m) may be NULL.
 * polish != 0: d_rho_vec must be NULL, sigma is used as delta for all of param2 (:254-265).
 * Returns 0, RLDL_LINSYS_SOLVER_INIT_ERROR, or RLDL_NONCVX_ERROR when ANY instance has fewer than n
 * positive pivots or a zero pivot (:80-92); per-instance detail via rldl_batch_factor_status. */
c_int rldl_batch_init(rldl_batch **hp, c_int batch, const csc *P, const csc *A, const c_float *d_Px,
                      const c_float *d_Ax, c_float sigma, const c_float *d_rho_vec, c_int polish,
                      const c_int *perm, void *stream);
/* solve (qdldl_interface.c:559-585): in/out d_b; non-polish post-condition b[0:n]=x_tilde,
 * b[n+j] += rho_inv[j]*nu[j]; polish: raw KKT solution. */
c_int rldl_batch_solve(rldl_batch *h, c_float *d_b);
/* update_matrices (qdldl_interface.c:590-602): scatter through PtoKKT/AtoKKT (src/kkt.c:184-212), full
 * numeric refactor.  Either pointer may be NULL (= unchanged).  Returns 0 ok / 1 failed. */
c_int rldl_batch_update_matrices(rldl_batch *h, const c_float *d_Px, const c_float *d_Ax);
/* update_rho_vec (qdldl_interface.c:605-619): rho_inv, KKT diagonal (src/kkt.c:214-222), refactor.
 * d_mask (device, int32[batch]) may be NULL; otherwise only instances with mask != 0 are touched. */
c_int rldl_batch_update_rho_vec(rldl_batch *h, const c_float *d_rho_vec, const int *d_mask);
/* free (qdldl_interface.c:17-43) */
void rldl_batch_free(rldl_batch *h);

/* Host-only symbolic analysis (no device needed): form_KKT pattern + maps (src/kkt.c:6-177), ordering +
 * csc_symperm + map composition (qdldl_interface.c:99-166), QDLDL_etree (call site :59).  Any output
 * pointer may be NULL; call once with NULL arrays to obtain nnzKKT / nnzL, then again with buffers.
 * Returns 0 or RLDL_LINSYS_SOLVER_INIT_ERROR. */
c_int rldl_symbolic_analyze(const csc *P, const csc *A, c_int polish, const c_int *perm_in, c_int *nnzKKT,
                            c_int *nnzL, c_int *etree_height, c_int *perm, c_int *etree, c_int *Lnz, c_int *Lp,
                            c_int *Li, c_int *KKTp, c_int *KKTi, c_int *PtoKKT, c_int *AtoKKT, c_int *rhotoKKT);
/* Host-only export of the grouped triangular-solve plan (rldl_plan.c) for inspection / CPU emulation in tests.
 * meta: 48 entries (layout: csrc/rldl_backend.c). */
c_int rldl_plan_export(const csc *P, const csc *A, c_int polish, const c_int *perm_in, c_int *meta, int *blob, c_int blob_cap,
                       c_int *LtoS);
/* closed-form stage-interleaved permutation, compute_permutations src/recursive_ldl.c:1350-1362 */
void rldl_stage_permutation(c_int N, c_int nx, c_int nu, c_int ny, c_int nt, c_int *perm);

/* introspection for parity tests and for the roofline's algorithmic-byte count */
c_int rldl_batch_dims(const rldl_batch *h, c_int *n, c_int *m, c_int *nnzKKT, c_int *nnzL, c_int *batch);
c_int rldl_batch_export_symbolic(const rldl_batch *h, c_int *perm, c_int *etree, c_int *Lnz, c_int *Lp,
                                 c_int *Li, c_int *KKTp, c_int *KKTi, c_int *PtoKKT, c_int *AtoKKT,
                                 c_int *rhotoKKT);
c_int rldl_batch_export_factor(const rldl_batch *h, c_int inst, c_float *Lx, c_float *D, c_float *Dinv,
                               c_float *KKTx);                      /* host buffers, CSC order */
c_int rldl_batch_factor_status(const rldl_batch *h, c_int *status); /* host[batch]: #positive D, or -1 */
/* Stage handles: tables of the product tri-solve (csrc/rldl_device.h, rldl_dev_stage.pv_*) and the tile values of one instance,
 * copied back for inspection / CPU emulation in tests.  meta[8] = { usable, tiles, table words, Ti entries, nb, row length of the inverted tiles (positions in src), kmax, steps };
 * with null arrays only meta is filled; prog int32[12 steps], tinfo int32[4 (tiles + 1)], tab uint32[table words],
 * src uint16[Ti entries], blk int32[2 nb], Ti double[Ti entries].  Returns 1 when the handle has no such tables. */
c_int rldl_batch_export_prod(const rldl_batch *h, c_int inst, c_int *meta, int *prog, int *tinfo, unsigned *tab, unsigned short *src,
                             int *blk, c_float *Ti);
/* average device time (ms) of the last `solve` kernel launch group measured with HIP events on the
 * handle's stream; bench.py's live roofline measurement uses this */
c_int rldl_batch_time_solve(rldl_batch *h, c_float *d_b, c_int reps, c_float *ms_per_launch);
/* the same over a rotation of handles that share one stream (launch r solves hs[r % count] on d_b[r % count]): with a combined
 * working set beyond the 256 MB Infinity Cache every launch streams its factor from HBM */
c_int rldl_batch_time_solve_rotating(rldl_batch **hs, c_float **d_b, c_int count, c_int reps, c_float *ms_per_launch);
/* cache policy of the factor rows in `solve`: 0 automatic (by the size of the rows of one solve against the 256 MB Infinity Cache),
 * 1 keep them cached between solves (one factorisation, many solves: the ADMM loop through the plugin API), 2 stream them with
 * non-temporal loads (many handles in turn, or rows beyond the cache) */
c_int rldl_batch_set_cache_policy(rldl_batch *h, c_int policy);
/* tracing aid: wave timeline of one launch of the solve kernel, host_out[batch][8] int64 ticks of the 100 MHz device clock
 * (wave start, all loads landed, forward gather / forward product / backward product / scatter done, stores issued, 0); 2 = this handle's solve kernel carries no timeline */
c_int rldl_batch_trace_solve(rldl_batch *h, c_float *d_b, long long *host_out);
/* the same for one numeric factorisation of the values the handle holds (arrowhead factor kernel): wave start, KKT values in the
 * workspace, head contributions added, tail in registers, tail eliminated, factor row stored, triangle packed, tail inverse stored */
c_int rldl_batch_trace_factor(rldl_batch *h, long long *host_out);

/* =====================================================================================
 * 3. Batched ADMM driver (device-resident mirror of src/osqp.c + src/auxil.c step kernels)
 * ===================================================================================== */
typedef struct {           /* subset of OSQPSettings, include/types.h:139-176; defaults constants.h:59-115 */
  c_float rho, sigma, alpha, eps_abs, eps_rel, eps_prim_inf, eps_dual_inf;
  c_int   max_iter, check_termination, warm_start, scaling, scaled_termination;
  c_int   adaptive_rho, adaptive_rho_interval;
  c_float adaptive_rho_tolerance;
  c_int   polish, polish_refine_iter;   /* constants.h:77-78 (0, 3) */
  c_float delta;                        /* constants.h:76 (1e-6) */
} OSQPBatchSettings;

typedef struct osqp_batch osqp_batch;

void  osqp_batch_set_default_settings(OSQPBatchSettings *s);           /* osqp.c:24-71 */
/* osqp_setup (osqp.c:76-283).  All value arrays are DEVICE pointers, instance-major:
 * d_Px[batch][nnzP], d_Ax[batch][nnzA], d_q[batch][n], d_l/d_u[batch][m]. */
c_int osqp_batch_setup(osqp_batch **wp, c_int batch, const csc *P, const csc *A, const c_float *d_Px,
                       const c_float *d_Ax, const c_float *d_q, const c_float *d_l, const c_float *d_u,
                       const OSQPBatchSettings *settings, const c_int *perm, void *stream);
c_int osqp_batch_solve(osqp_batch *w);                                  /* osqp.c:288-641 */
/* the same, enqueued on the workspace's stream without waiting when the settings need no host decision inside the
 * loop (check_termination == 0 and adaptive_rho == 0); results are valid after osqp_batch_wait */
c_int osqp_batch_solve_async(osqp_batch *w);
c_int osqp_batch_wait(osqp_batch *w);
c_int osqp_batch_update_lin_cost(osqp_batch *w, const c_float *d_q);   /* osqp.c:752-790 */
c_int osqp_batch_update_bounds(osqp_batch *w, const c_float *d_l, const c_float *d_u); /* osqp.c:792-841 */
/* the same, only enqueued (no host round trip, no stream drain): the l <= u check runs on the device, a refused update leaves l, u
 * untouched and is reported by the next osqp_batch_wait (return code 1 = the reference's exitflag, osqp.c:805-813);
 * RLDL_DEVICE_ERROR when a HIP call fails.  The blocking calls refuse to run while such a verdict is unread. */
c_int osqp_batch_update_bounds_async(osqp_batch *w, const c_float *d_l, const c_float *d_u);
c_int osqp_batch_update_rho(osqp_batch *w, c_float rho_new);           /* osqp.c:1268-1319 */
/* osqp_update_max_iter, _eps_abs, _eps_rel, _eps_prim_inf, _eps_dual_inf, _alpha, _warm_start, _scaled_termination,
 * _check_termination, _polish_refine_iter, _delta (osqp.c:1321-1560) in one call; other fields of `s` are ignored */
c_int osqp_batch_update_settings(osqp_batch *w, const OSQPBatchSettings *s);
c_int osqp_batch_update_P_A(osqp_batch *w, const c_float *d_Px, const c_float *d_Ax); /* osqp.c:1158-1266 */
/* the same, only enqueued: a failed refactorisation is reported by the next osqp_batch_wait / osqp_batch_solve (RLDL_NONCVX_ERROR) */
c_int osqp_batch_update_P_A_async(osqp_batch *w, const c_float *d_Px, const c_float *d_Ax);
c_int osqp_batch_warm_start(osqp_batch *w, const c_float *d_x, const c_float *d_y);   /* osqp.c:929-948 */
/* results: device pointers owned by the workspace (valid until cleanup).  x, y are the OSQPSolution of
 * store_solution (auxil.c:527-565: unscaled, NaN when the instance is infeasible / non-convex); z is work->z. */
c_int osqp_batch_get(osqp_batch *w, c_float **d_x, c_float **d_y, c_float **d_z, int **d_status,
                     int **d_iter, c_float **d_obj, c_float **d_pri_res, c_float **d_dua_res);
/* work->x, y, z (scaled iterates) and work->delta_x, delta_y (the infeasibility certificates after a solve) */
c_int osqp_batch_get_iterates(osqp_batch *w, c_float **d_x, c_float **d_y, c_float **d_z, c_float **d_delta_x,
                              c_float **d_delta_y);
/* OSQPScaling (include/types.h:43-48): D[batch][n], E[batch][m], c[batch]; returns 1 when scaling is off */
c_int osqp_batch_get_scaling(osqp_batch *w, c_float **d_D, c_float **d_E, c_float **d_c);
/* info->status_polish per instance (types.h:95): 0 not performed, 1 successful, -1 unsuccessful */
c_int osqp_batch_get_polish_status(osqp_batch *w, int **d_status_polish);
/* settings->rho as each instance currently runs it (adapt_rho -> osqp_update_rho, src/auxil.c:41-77, src/osqp.c:1268-1319),
 * info->rho_estimate and info->rho_updates (types.h:100-101): [batch] each */
c_int osqp_batch_get_rho(osqp_batch *w, c_float **d_rho, c_float **d_rho_estimate, int **d_rho_updates);
rldl_batch *osqp_batch_linsys(osqp_batch *w);
/* timing of the fused ADMM-iteration kernel (HIP events on the workspace stream), for the roofline:
 * reps == 0: device time per ITERATION of the last solve loop; reps > 0: time `reps` single-iteration launches now */
c_int osqp_batch_time_iteration(osqp_batch *w, c_int reps, c_float *ms_per_launch);
/* device time of the last solve loop, the iterations it ran and in how many launch groups (one kernel launch runs a
 * whole group of iterations when the factor can stay on chip) */
c_int osqp_batch_last_loop(osqp_batch *w, c_float *ms, c_int *iterations, c_int *launch_groups);
/* wave timeline of one launch of `iters` fused iterations: host_out[batch][8] int64 ticks of the 100 MHz device clock;
 * slots 0..6 = last iteration (start, rhs, gather, forward sweep, backward sweep, scatter, update), slot 7 = wave start */
c_int osqp_batch_trace_iteration(osqp_batch *w, c_int iters, long long *host_out);
void  osqp_batch_cleanup(osqp_batch *w);                                /* osqp.c:646-744 */

/* Several workspaces as one set (batches whose instances fall into a few sparsity patterns: one workspace per pattern, the
 * "per-instance pattern" variant of BASELINE config 2).  With a fixed number of iterations (check_termination = 0, adaptive_rho = 0,
 * polish = 0) osqp_multi_solve runs osqp_solve (osqp.c:354-641) of ALL workspaces in a handful of launches over the stacked instances
 * (solve_begin, the fused iterations once per kernel instantiation the patterns select, the closing check)
 * and osqp_multi_get writes the OSQPSolution / OSQPInfo fields in the caller's instance order: dest[k] = caller row of stacked
 * instance k (workspace 0's instances first).  osqp_multi_create returns 2 when the set does not qualify (other settings, patterns
 * off the tile kernels): solve the workspaces one by one then; 1 when dest is not a permutation of 0 .. total-1.  osqp_multi_solve
 * re-checks the settings of every member (they can change through osqp_batch_update_settings) and returns 2 when the set no longer
 * qualifies.  The workspaces stay owned by the caller. */
typedef struct osqp_multi osqp_multi;
/* the key of the fused kernel instantiation the workspace's pattern selects (equal keys share launches), -1: off the tile kernels,
 * the workspace cannot join a set */
/* problems of different sparsity patterns (each osqp_setup of the reference owns its pattern, qdldl_interface.c:99-166): bucket them
 * by (pattern of P, pattern of A) -- group[i] = bucket of problem i, numbered by first appearance; returns the number of buckets or -1
 * -- build one workspace per bucket (osqp_batch_setup on the stacked values), then join the workspaces into a set below. */
c_int osqp_groups_bucket(c_int count, const csc *const *P, const csc *const *A, c_int *group);
c_int osqp_batch_multi_key(const osqp_batch *w);
c_int osqp_multi_create(osqp_multi **mp, osqp_batch **ws, c_int count, const c_int *dest, void *stream);
c_int osqp_multi_solve(osqp_multi *mm);
/* osqp_update_P_A of every workspace of the set in one chain (scatter, numeric factorisation, tail inverse over the stacked instances);
 * d_Px[g] / d_Ax[g] = new values of ws[g], device arrays.  Enqueue-only; a failed refactorisation surfaces at the next osqp_multi_solve
 * (RLDL_NONCVX_ERROR).  2: the set does not qualify (equilibration on, different factor kernels): update the workspaces one by one. */
c_int osqp_multi_update_P_A(osqp_multi *mm, const c_float *const *d_Px, const c_float *const *d_Ax);
c_int osqp_multi_get(osqp_multi *mm, c_float *d_x, c_float *d_y, c_float *d_z, int *d_status, int *d_iter, c_float *d_obj,
                     c_float *d_pri_res, c_float *d_dua_res);
void  osqp_multi_free(osqp_multi *mm);

/* =====================================================================================
 * 4. Stage-recursive LDL for MPC-structured KKT (src/recursive_ldl.c)
 * ===================================================================================== */
typedef struct {           /* stage sizes: include/recursive_ldl.h:17-50 (N, nx, nu, ny, nt) */
  c_int N, nx, nu, ny, nt;
} rldl_stage_dims;
/* Host-only export of the product tri-solve's tables of a stage-structured pattern (csrc/rldl_device.h: rldl_dev_stage.pv_*), for
 * inspection / CPU emulation in tests.  meta[8] and the arrays as in rldl_batch_export_prod (without tile values); LtoS[nnzL]
 * maps the CSC entries of L to the factor slots the coupling tiles name.  2: the pattern does not qualify. */
c_int rldl_stage_prod_export(const csc *P, const csc *A, const rldl_stage_dims *dims, c_int *meta, int *prog, int *tinfo, unsigned *tab,
                             unsigned short *src, int *blk, c_int *LtoS);

/* Replaces LDL_factorize_recursive (src/recursive_ldl.c:1139-1318) + init_linsys_solver_qdldl_recursive
 * (:1555-1672): the batch handle is built from the ASSEMBLED P, A of setup_AP_matrices (:1873-1970)
 * and factorised stage by stage with the closed-form interleaved permutation (:1350-1362). */
c_int rldl_batch_init_recursive(rldl_batch **hp, c_int batch, const rldl_stage_dims *dims, const csc *P,
                                const csc *A, const c_float *d_Px, const c_float *d_Ax, c_float sigma,
                                const c_float *d_rho_vec, void *stream);
/* Replaces LDL_update_from_pivot (:946-1110): re-run the stage recursion from stage `first_stage` on
 * (after P/A values of stages >= first_stage, or rho, changed). */
c_int rldl_batch_update_from_stage(rldl_batch *h, c_int first_stage, const c_float *d_Px, const c_float *d_Ax,
                                   const c_float *d_rho_vec);

/* Replaces setup_AP_matrices (src/recursive_ldl.c:1873-1970): assemble P (upper triangular) and A from the seven
 * stage blocks (include/recursive_ldl.h:30-36).  *P_out / *A_out are allocated here (release with rldl_csc_free).
 * The optional kind/stage/entry arrays (sized by the outputs' nnz; upper bounds: (N-1) nnz(Qi) + nnz(Q0) + nnz(QN) and
 * N (nnz(Ai) + nnz(Aij)) + nnz(A0) + nnz(AN)) name the source of every stored value
 * (P kinds: 0=Q0 1=Qi 2=QN; A kinds: 0=A0 1=Ai 2=Aij 3=AN), which is what a batched update_AP_matrices (:1675-1778) needs. */
/* ADMM-level mirror of the recursive entry points (include/recursive_ldl.h:52-74):
 *   osqp_setup_recursive (src/recursive_ldl.c:2018-2230)        -> osqp_batch_setup_recursive
 *   LDL_update_from_pivot at a fixed horizon (:946-1110)        -> osqp_batch_update_recursive (new values from a stage on)
 *   osqp_solve_recursive (:2867-...)                            -> osqp_batch_solve
 *   osqp_partial_update_bounds (:119-200)                       -> osqp_batch_partial_update_bounds
 *   get_L_dimensions (:1334-1342) / cleanup_rldl (:15-60)       -> rldl_batch_dims(osqp_batch_linsys(w)) / osqp_batch_cleanup
 *   osqp_update_recursive (:1973-2016, changes the horizon N)   -> osqp_horizon_update (section 5)
 * The stage blocks carry the NOMINAL values, replicated to every instance; per-instance values follow through
 * osqp_batch_update_P_A / osqp_batch_update_recursive in the value order of *P_out / *A_out (free with rldl_csc_free). */
c_int osqp_batch_setup_recursive(osqp_batch **wp, c_int batch, const rldl_stage_dims *dims, const csc *Q0, const csc *Qi,
                                 const csc *QN, const csc *A0, const csc *Ai, const csc *Aij, const csc *AN,
                                 const c_float *d_q, const c_float *d_l, const c_float *d_u,
                                 const OSQPBatchSettings *settings, csc **P_out, csc **A_out, void *stream);
c_int osqp_batch_update_recursive(osqp_batch *w, c_int first_stage, const c_float *d_Px, const c_float *d_Ax);
c_int osqp_batch_partial_update_bounds(osqp_batch *w, c_int start, c_int stop, const c_float *d_l, const c_float *d_u);
c_int osqp_batch_partial_update_bounds_async(osqp_batch *w, c_int start, c_int stop, const c_float *d_l, const c_float *d_u);

c_int rldl_setup_AP_matrices(const rldl_stage_dims *dims, const csc *Q0, const csc *Qi, const csc *QN, const csc *A0,
                             const csc *Ai, const csc *Aij, const csc *AN, csc **P_out, csc **A_out, c_int *P_kind,
                             c_int *P_stage, c_int *P_entry, c_int *A_kind, c_int *A_stage, c_int *A_entry);
void rldl_csc_free(csc *M);

/* =====================================================================================
 * 5. Variable horizon (osqp_setup_recursive with Nmax + osqp_update_recursive, src/recursive_ldl.c:2018-2230, :1973-2016)
 * ===================================================================================== */
typedef struct osqp_horizon osqp_horizon;

/* osqp_setup_recursive(workp, data, settings, Nmax, N, nx, nu, ny, nt): the problem is set up at horizon dims->N and may
 * later move anywhere in 1..Nmax.  The seven stage blocks are copied (nominal values); d_q/d_l/d_u: [batch][n(N)], [batch][m(N)].
 * Returns 0 or the codes of osqp_batch_setup; 1 when N is outside 1..Nmax. */
c_int osqp_horizon_setup(osqp_horizon **hp, c_int batch, const rldl_stage_dims *dims, c_int Nmax, const csc *Q0, const csc *Qi,
                         const csc *QN, const csc *A0, const csc *Ai, const csc *Aij, const csc *AN, const c_float *d_q,
                         const c_float *d_l, const c_float *d_u, const OSQPBatchSettings *settings, void *stream);
/* osqp_update_recursive(work, data, N) (:1973-2016): move every instance to horizon Nnew.  Returns -1 when Nnew is outside
 * 1..Nmax (as the reference), 0 when Nnew is the current horizon (nothing happens) or on success.  d_q/d_l/d_u are the vectors
 * of the NEW problem ([batch][Nnew (nx+nu)], [batch][Nnew (nx+ny) + nt]; the reference leaves their refresh to the caller).
 * P / A: stages before min(N, Nnew) keep each instance's values, later stages are the nominal blocks again (update_AP_matrices,
 * :1675-1778; upload per-instance values for them afterwards with osqp_batch_update_recursive).  The factor of the shared stages is
 * copied from the old horizon and the stage recursion restarts at stage min(N, Nnew) (LDL_update_from_pivot, :946-1110);
 * rho stays per instance; x / y carry over on the shared stages (terminal multipliers to the terminal rows) when
 * warm_start is set.  The workspace handle of the new horizon is osqp_horizon_workspace(h): solve, update and read
 * results through the osqp_batch_* calls; it stays owned by h (do not osqp_batch_cleanup it). */
c_int osqp_horizon_update(osqp_horizon *h, c_int Nnew, const c_float *d_q, const c_float *d_l, const c_float *d_u);
osqp_batch *osqp_horizon_workspace(osqp_horizon *h);
c_int osqp_horizon_N(const osqp_horizon *h);
/* what the last osqp_horizon_update did: the first stage that was refactorised, how many instances restarted there (the others
 * were factorised from the first block), and whether the workspace of that horizon had to be created (first visit) */
c_int osqp_horizon_last_update(const osqp_horizon *h, c_int *pivot_stage, c_int *instances_reused, c_int *workspace_created);
/* Single store (the reference's "combined" X / Z / Y variant, src/recursive_ldl.c:2359-2856: everything sized for Nmax once, a
 * horizon change rebuilds the trailing part and its border only).  With scaling = 0 on a pattern the product tri-solve covers,
 * osqp_horizon_setup builds ONE workspace at Nmax dimensions: horizon N lives on the leading entries of its arrays (row strides
 * osqp_horizon_ld: n and m of Nmax), the stages behind N are decoupled dummies that are neither factorised nor solved, and
 * osqp_horizon_update restarts the recursion inside the same factor store -- no second workspace, no copy of shared columns.
 * Values and iterates of the CURRENT horizon go through the two calls below (packed arrays in that horizon's own P / A value
 * order and sizes); results are read from osqp_horizon_workspace with the row strides of osqp_horizon_ld. */
c_int osqp_horizon_is_single(const osqp_horizon *h);
c_int osqp_horizon_workspaces(const osqp_horizon *h);                     /* resident numeric workspaces: 1 with the single store */
c_int osqp_horizon_ld(const osqp_horizon *h, c_int *ld_n, c_int *ld_m);
c_int osqp_horizon_update_P_A(osqp_horizon *h, const c_float *d_Px, const c_float *d_Ax);   /* osqp_update_P_A at the current horizon */
c_int osqp_horizon_warm_start(osqp_horizon *h, const c_float *d_x, const c_float *d_y);      /* osqp_warm_start at the current horizon */
void osqp_horizon_free(osqp_horizon *h);

/* ---------------------------------------------------------------------------------------------------------------------
 * 7. Multi-GPU in C: one process per GPU, the batch shards by contiguous ranges, the data path has no collective;
 *    ONE RCCL all-gather of the packed result records collects the solutions.  librccl.so is dlopen'ed at the first call (the
 *    reference loads its optional Pardiso backend the same way, lin_sys/lib_handler.c:7-49): no link-time dependency.
 *      rank 0:      osqp_dist_unique_id(id)          -> ship the 128 bytes to the other ranks (launcher's business)
 *      every rank:  osqp_dist_init(&d, id, rank, nranks, stream)
 *                   ... osqp_batch_solve(w) ...
 *                   osqp_dist_gather_results(d, w, d_out)   d_out [nranks * batch][n + m + 5] on the device:
 *                                                           x | y | obj | pri_res | dua_res | iter | status per instance, ranks in order
 * --------------------------------------------------------------------------------------------------------------------- */
typedef struct osqp_dist osqp_dist;
/* pack only (one launch on the workspace's stream, d_rec [batch][osqp_dist_record_len]): for callers with their own collective */
c_int osqp_batch_pack_results(osqp_batch *w, c_float *d_rec);
c_int osqp_dist_unique_id(char id[128]);
c_int osqp_dist_init(osqp_dist **dp, const char id[128], c_int rank, c_int nranks, void *stream);
c_int osqp_dist_record_len(const osqp_batch *w);                         /* n + m + 5 */
c_int osqp_dist_gather_results(osqp_dist *d, osqp_batch *w, c_float *d_out);
void  osqp_dist_free(osqp_dist *d);

const char *rldl_version(void);

#ifdef __cplusplus
}
#endif
#endif
