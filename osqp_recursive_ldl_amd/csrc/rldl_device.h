/*
 * rldl_device.h -- the thin C boundary between the plain-C host code and the HIP kernels.
 *
 * Everything below is `extern "C"`, takes plain pointers/sizes, and only ENQUEUES work on the given
 * stream (no allocation, no synchronisation), so a caller may capture the launchers in a hipGraph.
 */
#ifndef RLDL_DEVICE_H
#define RLDL_DEVICE_H

#ifdef __cplusplus
extern "C" {
#endif

/* device-resident copy of rldl_symbolic (all pointers are DEVICE pointers) */
/* Dense stage-block view of a block-tridiagonal (MPC) pattern for k_stage_factor: per diagonal block b the
 * entries of the permuted KKT matrix and of L that fall into block (b, b) and into the coupling block (b+1, b), each
 * with its position in a dense ld x ld LDS tile (row * ld + col, rows/cols local to their blocks). */
#define RLDL_PV_RING 4   /* groups of the product tri-solve whose loads are in flight + 1 (stage_prod_solve); pv_prog is padded to multiples of it */
typedef struct {
  int nb, ld, smax;
  /* single-store horizon handles (rldl_horizon.c): only the first nb_act blocks are live (0 = all nb); the blocks behind them
   * belong to stages beyond the current horizon, are decoupled (zero coupling values) and are neither factorised nor solved;
   * npos_skip = their variable positions, which count as positive pivots in the inertia verdict */
  int nb_act, npos_skip;
  const int *bs;                          /* [nb+1] first permuted index of each block */
  const int *kd_ptr, *kd_src, *kd_pos;    /* KKT values of the diagonal block: Kx index -> tile position (lower part) */
  const int *kc_ptr, *kc_src, *kc_pos;    /* KKT values of the coupling block (b+1, b) */
  const int *ld_ptr, *ld_slot, *ld_pos;   /* L entries inside block b: factor slot <- tile position */
  const int *lc_ptr, *lc_slot, *lc_pos;   /* L entries of L(b+1, b) */
  /* block tri-solve (stage_tri_solve): per tile (2b = L(b+1, b), 2b + 1 = L_bb) 384 words (tile byte offset << 16 | factor slot),
   * tile rows sv_ld = SM + 1 wide, padded, lane-major; sv_coff = rounds per coupling tile | rounds per diagonal tile << 8;
   * sv_prog[8 k] = { c0, s, o0, coupling tile, diagonal tile, 0, 0, 0 } per (direction, block), see rldl_recursive.c */
  const unsigned *sv_pk;
  const int *sv_prog;
  int sv_ok, sv_ld, sv_coff, sv_ntiles;
  /* product tri-solve (stage_prod_solve, k_stage_invert): the diagonal blocks are inverted behind every factorisation, so both
   * passes are chains of sparse tile products without a dependent sweep.  Tiles in forward order D_0, C_0, D_1, ..., D_{nb-1}
   * (D_b = -(strictly lower part of L_bb^-1), C_b = L(b+1, b); empty tiles are left out); rldl_dev_num.Ti holds their entries,
   * tile after tile, step after step, lane after lane (layout: rldl_recursive.c, build_prod_tiles).  Every row of a tile has its
   * own lanes, every lane takes <= 1 entry per step, steps come in groups of four:
   *   pv_prog[12 s] = the group of step s of the kernel's sequence (forward groups, padding, backward groups, padding):
   *                   { first Ti entry of the group, first table word, byte address in x of the tile's first column | first row << 16,
   *                     1 (first group of its tile) | 2 (last), then the 64-bit lane mask of each of the four steps }
   *   pv_tab        : one word per (group, lane) = columns of the lane's four entries (5 bits each, relative to the tile's first
   *                   column) | one bit per step << 20 (lane has an entry) | the lane's row (relative) << 24 | 1 << 29 (lane has entries)
   *   pv_tinfo[4 t] = { first Ti entry, entries, kind (0 = D, 1 = C), first group }
   *   pv_src[i]     = where Ti entry i comes from: position row * pv_ldT + col in the inverted diagonal tile (D) / factor slot (C)
   *   pv_dpos[e]    = ld_pos[e] in k_stage_invert's tile geometry (rows pv_ldT = its SM + 2 long)
   *   pv_cidx[pv_cptr[b] .. pv_cptr[b + 1]) = the Ti entries of C_b (one flat list over the blocks: one batched copy loop)
   *   pv_blk[2 b], pv_blk[2 b + 1] = tile id of D_b, C_b or -1 */
  const unsigned *pv_tab;
  const int *pv_prog, *pv_tinfo, *pv_blk, *pv_dpos, *pv_cptr, *pv_cidx;
  const unsigned short *pv_src;
  int pv_ok, pv_ntiles, pv_ngroups, pv_nsteps, pv_kmax, pv_nTi, pv_ldTi, pv_ntab, pv_ldT;
  /* pv_mode 2: the coupling tiles hold the ORIGINAL coupling blocks K(b+1, b) of the permuted KKT matrix (pv_src = Kx index) instead of
   * L(b+1, b) = K(b+1, b) L_bb^-T D_b^-1; every diagonal tile is then used twice per pass while it sits in its ring slot, with an
   * auxiliary block vector at byte offset pv_aux behind x in the wave's LDS (step kinds: rldl_recursive.c, pv_sequence) */
  int pv_mode, pv_aux;
} rldl_dev_stage;

typedef struct {
  int n, m, N, nnzP, nnzA, nnzK, nnzL, nsig, polish;
  const int *perm, *PtoK, *AtoK, *rhotoK, *sigK;
  const unsigned char *Pisdiag;
  const int *Lp, *Li, *Rp, *Rj, *Rpos, *KtoW, *Udst;
  const unsigned int *Uab;
  const long long *Up;
  const int *Pp, *Pi, *Prp, *Prj, *Prpos, *Ap, *Ai, *Arp, *Arj, *Arpos;
  /* entry-parallel access to P and A: (row | col << 16) per stored entry in CSC order (Pfl, Afl), and a second order for the
   * kernels that add per-row / per-column contributions with LDS atomics: the entries dealt into rounds of 64 so that a round
   * holds as few entries of one row or one column as possible (an LDS atomic instruction costs ~6 cycles times the largest
   * number of lanes on one address).  Pbl/Abl = row | col << 16, Pbp/Abp = CSC position (0xffffffff: padding), Pbr/Abr rounds.
   * flat_ok = n, m < 65536 */
  const unsigned *Pfl, *Afl, *Pbl, *Pbp, *Abl, *Abp;
  int Pbr, Abr;
  int flat_ok;
  /* factor storage layout + grouped solve plan (rldl_plan.c) */
  const int *LtoS;             /* [nnzL] CSC position -> storage slot */
  const int *plan;             /* packed blob, plan_words int32 */
  int nS, nO, ngroups, plan_ok, plan_words;
  int ldF;                     /* row stride of rldl_dev_num.F in doubles: nS + N rounded up to even (16-byte rows) */
  int po_gstart, po_gflag, po_gaptr, po_grptr, po_gToff, po_fsp, po_bsp, po_acol, po_aoff, po_arow, po_coloff, po_fsb,
      po_fsc, po_bsb, po_bsc, po_fsig, po_bsig, po_fcol, po_brs, po_perm, po_avmap, po_avcol, po_avrow;
  int nOp;                     /* nO rounded up to even: first triangle slot */
  int arrow_ok, arrow_group, arrow_steps; /* arrowhead specialisation (see rldl_plan.c) */
  int arrow_vsteps, arrow_vrows;          /* virtual rows: coupling rows cut into pieces of <= vsteps entries, one piece per lane */
  int tile_ok, tile_ta, tile_tq, tile_lanes, nTi, ldTi;   /* tail inverse by register tiles (rldl_symbolic.h); ldTi = nTi rounded up to even */
  int po_tlane, po_tmap, po_tislot, po_tmask, po_pinv, po_trc, po_spack;
  int tile_admm_ok, tile_scatter_ok, tile_vslots, tile_slots, po_tpos;   /* ADMM slots of the tile kernels (rldl_symbolic.h) */
  int tile_ck[3], tile_tk, tile_sp, po_cmap, po_crow;            /* backward coupling product gathered by the owner lane */
  int arrow_g0, arrow_g;       /* index range of the tail group */
  int arrow_tb;                /* its triangle base relative to slot nOp, or -1 */
  int arrow_dense;             /* 1: every head column has entries only in the tail group, which ends the matrix (k_arrow_factor) */
  const int *arrow_tpos;       /* [arrow_g][64]: CSC position of L(g0 + lane, g0 + c), or -1 */
  const unsigned *arrow_pab, *arrow_pdc;  /* head pair updates, flat: posA | posB << 16 and dst | column << 16 (workspace positions) */
  int arrow_npairs;
  const unsigned *arrow_out;   /* [nS] per factor slot: workspace position | head column << 16 (0xffff: padding slot / tail column): write-out in slot order */
  int arrow_cnt[32];           /* per virtual-row step: number of lanes with an entry (kernarg segment -> scalar loads) */
  rldl_dev_stage stage;        /* stage.nb > 0: block-tridiagonal pattern, numeric factorisation by dense stage blocks */
} rldl_dev_sym;

/* per-batch numeric state of the linear-system backend (DEVICE pointers, instance-major) */
typedef struct {
  int batch;
  double sigma;
  double *Kx;       /* [batch][nnzK]   permuted KKT values                     */
  double *F;        /* [batch][ldF]    factor in plan slot order (nS), then Dinv (N): what `solve` streams */
  double *Ti;       /* [batch][ldTi]   inverse of the tail's unit lower triangle in tile order (tile_ok), else NULL */
  double *D;        /* [batch][N]      pivots (inertia check, export)           */
  double *rho_inv;  /* [batch][m]      param2 of the KKT (delta when polishing) */
  int *status;      /* [batch]         #positive pivots, or -1 on a zero pivot (verdict of the LAST factorisation of the instance) */
  int *fail;        /* [1]             sticky: set by any factorisation whose verdict is bad (zero pivot or fewer than n positive
                     *                  pivots, qdldl_interface.c:80-92); only the host clears it when it reads the verdict, so a
                     *                  failure in a stream of asynchronous refactorisations is not overwritten by a later success */
  int nt_loads;     /* host-side launch hint: the solve kernel reads the factor rows with non-temporal loads (rldl_batch_set_cache_policy) */
} rldl_dev_num;

/* ADMM iterate state (DEVICE pointers, instance-major) + scalar settings */
typedef struct {
  int batch;
  double sigma, alpha, eps_abs, eps_rel, eps_prim_inf, eps_dual_inf, rho, adaptive_rho_tolerance;
  const double *Px, *Ax, *q, *l, *u;
  double *x, *z, *y, *delta_x, *delta_y, *rho_vec;
  int *constr_type;
  double *pri_res, *dua_res, *obj, *rho_cur, *rho_est; /* [batch] */
  int *status, *iter, *rho_updates;                    /* [batch] */
  int *refactor;                                       /* [batch] mask written by the adapt-rho step */
  int *n_active;                                       /* [RLDL_NACT_SLOTS] instances still iterating, counted per slot inst % RLDL_NACT_SLOTS (one counter would serialise the batch's atomics in L2) */
  int write_delta;                                     /* 1: this iteration also stores delta_x / delta_y (a check follows) */
  int begin_flags;                                     /* first launch of a solve on the tile kernel (rldl_admm_begin_in_kernel): 1 = status <- OSQP_UNSOLVED
                                                        * by the kernel itself, 2 = cold start (x = z = y = 0, auxil.c:158-162): no k_solve_begin launch */
  /* polish (src/polish.c): masked copy of A (inactive rows zeroed), rhs / solution / refinement vectors [batch][n+m] */
  double *pol_Ax, *pol_b, *pol_z, *pol_r;
  int *pol_mask, *status_polish;
  int trace_iter;                                      /* iteration of the launch whose phases are recorded (-1: the last) */
  long long *trace;                                    /* wave timeline [batch][8] (s_memrealtime ticks) or NULL; osqp_batch_trace_iteration */
  /* Ruiz equilibration (src/scaling.c): per-instance D[n], E[m], their inverses, cost scaling c; 0 iterations = off */
  int scaling, scaled_termination;
  double *sD, *sDinv, *sE, *sEinv, *sc, *scinv;
  double *sol_x, *sol_y;                               /* unscaled solution (store_solution, auxil.c:527-565) */
} rldl_dev_admm;

/* Several workspaces in ONE launch (batches whose instances fall into a few sparsity patterns, one workspace per pattern): the
 * kernels that take this descriptor look their workgroup up in first_* and run it on the structs of its group.  ngroups = 0: an
 * ordinary launch on the structs passed by value. */
typedef struct {
  int ngroups;
  int total_tiles, total_insts;           /* workgroups of the two kinds of grid */
  const int *first_tile;                  /* DEVICE [ngroups + 1]: first workgroup of each group in grids of TILE_WPB instances per workgroup (k_tile_admm) */
  const int *first_inst;                  /* DEVICE [ngroups + 1]: first workgroup in grids of one instance per workgroup (k_solve_begin, k_admm_check, k_multi_gather) */
  const int *xdw;                         /* DEVICE [ngroups]: per-wave LDS doubles of k_tile_admm */
  const rldl_dev_sym *S;                  /* device arrays [ngroups] */
  const rldl_dev_num *N;
  const rldl_dev_admm *W;
} rldl_dev_multi;
/* per-group value arrays of an update of all groups: the caller's new P / A values and the workspaces' own copies, which the scatter
 * writes on the way.  DEVICE arrays [ngroups] of device pointers. */
typedef struct { const double *const *Px, *const *Ax; double *const *keepP, *const *keepA; } rldl_dev_multi_pa;

/* shared-memory footprint (bytes) of the LDS-resident variants; the launchers pick the global-memory
 * variant by themselves when this exceeds RLDL_LDS_LIMIT */
#define RLDL_LDS_LIMIT (64 * 1024)
#define RLDL_NACT_SLOTS 64

int rldl_launch_kkt_assemble(const rldl_dev_sym *S, const rldl_dev_num *Nn, const double *d_Px, const double *d_Ax,
                             const double *d_rho_vec, int set_sigma_only, const int *d_mask, void *stream);
int rldl_launch_kkt_assemble_keep(const rldl_dev_sym *S, const rldl_dev_num *Nn, const double *d_Px, const double *d_Ax, double *keepP,
                                  double *keepA, int *d_status_reset, int *d_rho_updates_reset, void *stream);   /* the two int arrays (or NULL): reset_info rides along */
int rldl_launch_factor(const rldl_dev_sym *S, const rldl_dev_num *Nn, const int *d_mask, void *stream);
int rldl_launch_stage_factor(const rldl_dev_sym *S, const rldl_dev_num *Nn, const int *d_mask, int first_block, void *stream);
int rldl_launch_stage_invert(const rldl_dev_sym *S, const rldl_dev_num *Nn, void *stream);
/* stage recursion where instance b restarts at block d_b0v[b] (0 = from the first block) */
int rldl_launch_stage_factor_each(const rldl_dev_sym *S, const rldl_dev_num *Nn, const int *d_b0v, int tiles_adopted, void *stream);
/* horizon change (src/recursive_ldl.c:1973-2016): old horizon's workspace (So, Wo / No) -> new horizon's (Sn, Nn) */
int rldl_launch_horizon_values(const rldl_dev_sym *So, const rldl_dev_admm *Wo, const double *oPx, const double *oAx, int col_keep,
                               double *nPx, int ldP, double *nAx, int ldA, void *stream);
int rldl_launch_horizon_state(const rldl_dev_sym *So, const rldl_dev_admm *Wo, int n_new, int m_new, int n_keep, int m_keep, int nt,
                              double *xs, double *ys, void *stream);
int rldl_launch_horizon_adopt(const rldl_dev_sym *So, const rldl_dev_num *No, const rldl_dev_sym *Sn, const rldl_dev_num *Nn,
                              const double *rvo, const double *rvn, int m_keep, int c0, int b_pivot, int ti_prefix, int *d_b0v,
                              int *d_n_reused, void *stream);
int rldl_launch_factor_from(const rldl_dev_sym *S, const rldl_dev_num *Nn, int c_start, void *stream);
int rldl_launch_solve(const rldl_dev_sym *S, const rldl_dev_num *Nn, double *d_b, void *stream);
int rldl_launch_factor_trace(const rldl_dev_sym *S, const rldl_dev_num *Nn, long long *d_trace, void *stream);
int rldl_launch_solve_trace(const rldl_dev_sym *S, const rldl_dev_num *Nn, double *d_b, long long *d_trace, void *stream);
int rldl_admm_begin_in_kernel(const rldl_dev_sym *S, const rldl_dev_num *Nn, const rldl_dev_admm *W);   /* 1: rldl_launch_admm_iters honours W->begin_flags */
int rldl_launch_admm_iter(const rldl_dev_sym *S, const rldl_dev_num *Nn, const rldl_dev_admm *W, void *stream);
int rldl_launch_bcast_rows(int batch, int len, double *dst, const double *src, void *stream);
int rldl_launch_set_range(int batch, int ld, int start, int cnt, double *dst, const double *src, const double *s, void *stream);
int rldl_launch_set_range_guarded(int batch, int ld, int start, int cnt, double *dst, const double *src, const double *s, const int *d_skip,
                                  void *stream);
/* single-store horizon change (rldl_horizon.c) */
int rldl_launch_bcast_range(int batch, int ld, int start, int cnt, double *dst, const double *src_row, void *stream);
int rldl_launch_scatter_rows(int batch, int cnt, int ld, const int *map, const double *src, double *dst, void *stream);
int rldl_launch_horizon_vectors(int batch, int n_act, int n_max, int m_act, int m_max, const double *q_src, const double *l_src,
                                const double *u_src, double *q, double *l, double *u, void *stream);
int rldl_launch_horizon_rho(const rldl_dev_sym *S, const rldl_dev_admm *W, const int *d_status, int m_keep, int b_pivot, int *d_b0v,
                            int *d_n_reused, void *stream);
int rldl_launch_horizon_state_single(const rldl_dev_admm *W, int n_max, int m_max, int n_keep, int m_keep, int term_old, int term_new,
                                     int nt, void *stream);
int rldl_launch_polish_prep(const rldl_dev_sym *S, const rldl_dev_admm *W, void *stream);
int rldl_launch_polish_resid(const rldl_dev_sym *S, const rldl_dev_admm *W, int add_first, void *stream);
int rldl_launch_polish_finish(const rldl_dev_sym *S, const rldl_dev_admm *W, int add_last, void *stream);
int rldl_launch_admm_iters(const rldl_dev_sym *S, const rldl_dev_num *Nn, const rldl_dev_admm *W, int iters, void *stream);
int rldl_launch_admm_check(const rldl_dev_sym *S, const rldl_dev_admm *W, int iter, int approximate, int final_pass,
                           void *stream);
int rldl_launch_set_rho_vec(const rldl_dev_sym *S, const rldl_dev_admm *W, int init, void *stream);
/* start of a solve in one launch: status = OSQP_UNSOLVED, active counter = batch, optional cold start and rho_updates = 0 */
int rldl_launch_solve_begin(const rldl_dev_admm *W, int n, int m, int cold, int reset_rho_updates, void *stream);
/* scale_data / unscale_data (src/scaling.c:44-173) on the workspace's own copies of P, A, q, l, u */
int rldl_launch_scale_data(const rldl_dev_sym *S, const rldl_dev_admm *W, double *Px, double *Ax, double *q, double *l, double *u,
                           int iters, void *stream);
int rldl_launch_unscale_data(const rldl_dev_sym *S, const rldl_dev_admm *W, double *Px, double *Ax, double *q, double *l, double *u,
                             void *stream);
/* dst[b][i] = src[b][i] * s[b][i] * (c ? c[b] : 1) */
int rldl_launch_ew_scale(int batch, int len, double *dst, const double *src, const double *s, const double *c, void *stream);
int rldl_launch_matvec_A(const rldl_dev_sym *S, const rldl_dev_admm *W, const double *d_x, double *d_out, void *stream);
/* *flag |= 1 when l[i] > u[i] for some i < count (osqp.c:805-813) */
int rldl_launch_check_bounds(long long count, const double *l, const double *u, int *flag, void *stream);
/* the fixed-iteration solve of several workspaces in three launches (osqp_multi_*, rldl_admm.c).  S0 / N0 / W0 = host copies of one
 * group's structs (they select the kernel instantiation; rldl_multi_key tells which groups may share a launch) */
int rldl_multi_key(const rldl_dev_sym *S, const rldl_dev_num *Nn, const rldl_dev_admm *W);
int rldl_multi_tile_xdw(const rldl_dev_sym *S);
int rldl_multi_tile_wpb(void);
int rldl_launch_multi_solve_begin(const rldl_dev_multi *M, int total, int n, int m, int cold, int reset_rho_updates, void *stream);
/* new P / A values for every group in one chain: scatter (+ the workspaces' own copies), arrowhead factorisation, tail inverse */
int rldl_multi_update_key(const rldl_dev_sym *S, const rldl_dev_num *Nn);
int rldl_multi_update_lds(const rldl_dev_sym *S, int which);
int rldl_launch_multi_update_rho(const rldl_dev_multi *M, int total, int key, int factor_lds, int invert_lds, void *stream);
int rldl_launch_multi_nactive(const rldl_dev_multi *M, int *d_out, void *stream);
int rldl_launch_multi_check(const rldl_dev_multi *M, const rldl_dev_sym *S0, const rldl_dev_admm *W0, int total, int iter, int approximate,
                            int final_pass, int max_nm, void *stream);
int rldl_launch_multi_fail(const rldl_dev_multi *M, int *d_out, void *stream);
int rldl_launch_pack_results(const rldl_dev_admm *W, int n, int m, double *rec, void *stream);   /* rldl_dist.c */   /* d_out[g] = sticky factorisation verdict of group g, cleared */
int rldl_launch_multi_update(const rldl_dev_multi *M, const rldl_dev_multi_pa *PA, int total, int key, int factor_lds, int invert_lds, void *stream);
int rldl_launch_multi_admm_iters(const rldl_dev_multi *M, const rldl_dev_sym *S0, const rldl_dev_num *N0, const rldl_dev_admm *W0, int iters,
                                 int max_xdw, void *stream);
int rldl_launch_multi_check_final(const rldl_dev_multi *M, const rldl_dev_sym *S0, const rldl_dev_admm *W0, int total, int iter, int max_nm,
                                  void *stream);
/* results of all groups into caller-order arrays: dest[first_inst[g] + i] = row of instance i of group g */
int rldl_launch_multi_gather(const rldl_dev_multi *M, int total, int n, int m, const int *dest, double *x, double *y, double *z, int *status,
                             int *iter, double *obj, double *pri, double *dua, void *stream);
const char *rldl_kernel_arch(void);

#ifdef __cplusplus
}
#endif
#endif
