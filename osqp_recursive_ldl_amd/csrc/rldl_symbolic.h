/*
 * rldl_symbolic.h -- host-side (plain C) symbolic analysis shared by every kernel of the backend.
 *
 * One rldl_symbolic describes the sparsity of ONE pattern group: the permuted upper-triangular KKT
 * matrix, the value-scatter maps of the reference's update path, the elimination tree, the pattern
 * of L in CSC and CSR order, and the update lists that drive the right-looking device factorisation.
 * All index arrays are int32 (device copies are uploaded verbatim).
 *
 * Reference functions whose result this analysis reproduces (host, integer work):
 *   form_KKT            src/kkt.c:6-177            (pattern + PtoKKT/AtoKKT/param2toKKT/Pdiag_idx)
 *   permute_KKT         lin_sys/direct/qdldl/qdldl_interface.c:99-166 (ordering, csc_pinv, csc_symperm, map composition)
 *   QDLDL_etree         call site qdldl_interface.c:59 (etree, Lnz; Lp = cumsum as QDLDL_factor builds it)
 */
#ifndef RLDL_SYMBOLIC_H
#define RLDL_SYMBOLIC_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
  int n, m, N;                 /* variables, constraints, KKT dimension n+m */
  int nnzP, nnzA, nnzK, nnzL;
  int polish;
  /* permutation: row/col k of the permuted matrix is row/col perm[k] of the original */
  int *perm, *pinv;
  /* permuted KKT, upper triangular CSC (rows ascending inside a column) */
  int *Kp, *Ki;
  /* value scatter maps into the permuted KKT value array */
  int *PtoK;                   /* [nnzP] */
  unsigned char *Pisdiag;      /* [nnzP] 1 where P(i,i): sigma is added (src/kkt.c:69-79, :196-202) */
  int *AtoK;                   /* [nnzA] */
  int *rhotoK;                 /* [m]    */
  int nsig, *sigK;             /* KKT slots holding sigma alone (P column without a diagonal entry) */
  /* elimination tree and L pattern (strictly lower, CSC, rows ascending) */
  int *etree, *Lnz, *Lp, *Li;
  int etree_height;
  /* L in row order: entries of row i are columns Rj[Rp[i]..Rp[i+1]) ascending, value slot Rpos[] */
  int *Rp, *Rj, *Rpos;
  /* factor workspace W = [L values (nnzL, CSC order) | D (N)]; KtoW maps KKT slot -> W slot */
  int *KtoW;
  /* right-looking updates: for column j, pairs t in [Up[j], Up[j+1]): W[Udst[t]] -= l_a * l_b * D_j,
   * a = Uab[t] & 0xffff, b = Uab[t] >> 16 index the entries of column j (b <= a) */
  long long *Up;               /* [N+1] */
  int *Udst;
  unsigned int *Uab;
  long long npairs;
  /* ---- solve plan (see rldl_plan_build): storage layout of the factor + grouped schedule ---- */
  int plan_ok;                 /* 0: limits exceeded, only the generic (v1) kernels may be used */
  int nS;                      /* factor storage slots per instance (>= nnzL; padding slots stay zero) */
  int nO;                      /* out-of-group entries (stored first, CSC order) */
  int ngroups;
  int *LtoS;                   /* [nnzL] CSC position -> storage slot */
  int *plan;                   /* packed int32 blob uploaded to the device */
  int plan_words;
  int po_gstart, po_gflag, po_gaptr, po_grptr, po_gToff, po_fsp, po_bsp, po_acol, po_aoff, po_arow, po_coloff, po_fsb,
      po_fsc, po_bsb, po_bsc, po_fsig, po_bsig, po_fcol, po_brs, po_perm, po_avmap, po_avcol, po_avrow;
  int nOp;                     /* nO rounded up to even: first triangle slot */
  int arrow_ok, arrow_group, arrow_steps; /* arrowhead specialisation (see rldl_plan.c) */
  int arrow_vsteps, arrow_vrows;          /* virtual rows: coupling rows cut into pieces of <= vsteps entries, one piece per lane */
  /* tail inverse by register tiles (k_tile_* kernels): the g x g unit lower triangle of the tail group is inverted at
   * factor time and its strictly lower part is cut into ta x ta tiles, one tile per lane (tq tiles per side, full tiles
   * first, then the diagonal tiles).  Register k = s * ta + u of the lane that owns tile (I, J) holds
   * Linv(ta I + (s + J) % ta, ta J + (u + I) % ta): rows rotated by J, columns by I, so the LDS atomics of lanes that share a
   * block row / block column start at different addresses.  Values live in their own array Ti[batch][ldTi] in (k, lane)
   * order, structural zeros (diagonal tiles, rows >= g) are not stored.
   *   po_tlane  [64]                 I | J << 8, 0xffffffff for lanes without a tile
   *   po_tmap   [ceil(ta^2 / 2)][64] slot(k even) | slot(k odd) << 16, 0xffff = structural zero
   *   po_tislot [g][32]              u16 [g][64]: slot of Linv(i, c) for lane c, 0xffff where c >= i (inverse kernel)
   *   po_tmask  [ta^2][2]            po_tmap without a per-lane table: the 64-bit lane mask of register k (lo, hi); the slot of (k, lane) is
   *                                  (entries of the registers before k) + (mask bits below the lane); 16-byte aligned
   *   po_pinv   [3][64]              permuted position of every original index (inverse of perm); entries past N = xdw + lane, the
   *                                  lane's dummy word in the tile kernels' LDS geometry (po_avrow is padded the same way)
   *   po_trc    [ta][64]             word s of the lane = rotated row s | rotated column s << 16 of its tile, local to the tail group */
  int tile_ok, tile_ta, tile_tq, tile_lanes, nTi;
  int po_tlane, po_tmap, po_tislot, po_tmask, po_pinv, po_trc;
  int po_spack;                /* k_tile_solve3's per-lane table words as 16-byte records [chunk][64][4] (0: none) */
  /* ADMM slots of the tile kernels: the permuted positions are dealt to (slot, lane) so that a slot holds variables only or
   * constraints only (uniform code per slot, no per-lane role test): po_tpos [3][64] = permuted position or -1;
   * tile_vslots = number of leading variable slots, tile_slots = slots in use (<= 3; more -> tile_admm_ok = 0) */
  int tile_admm_ok, tile_scatter_ok, tile_vslots, tile_slots, po_tpos;   /* tile_scatter_ok: the slot layout alone (fused iterations with the backward coupling product as a scatter) */
  /* Backward coupling product as a GATHER by the owner of each head entry (scattered ds_add_f64 costs ~14 LDS cycles per
   * instruction, a read 2-5): the lane that owns position c in slot t also holds the entries L(r, c) of column c (a second
   * register copy of the coupling values) and sums L(r, c) x_r itself.  Positions are dealt to the slots of their kind by
   * decreasing column count, so slot t needs tile_ck[t] = (count of its first column) steps.  Head entries must all be
   * constraints (ck = 0 for the variable slots) in at most two constraint slots: the first one's columns take steps
   * [0, tile_sp), the second one's [tile_sp, tile_tk) -- a compile-time split (tile_sp = 3 tk / 4 or 2 tk / 3, even);
   * tile_tk = 16 / 24 / 32, the smallest that fits.
   *   po_cmap [tile_tk / 2][64]  factor slot(k even) | slot(k odd) << 16, 0xffff = none
   *   po_crow [tile_tk / 2][64]  same packing: row of the entry, local to the tail group */
  int tile_ck[3], tile_tk, tile_sp, po_cmap, po_crow;
  /* problem matrices for the residual kernels: CSC as given plus row-order (CSR) access maps */
  int *Pp, *Pi, *Prp, *Prj, *Prpos;   /* P upper triangular n x n */
  int *Ap, *Ai, *Arp, *Arj, *Arpos;   /* A m x n */
} rldl_symbolic;

/* Pp/Pi/Ap/Ai: 64-bit CSC pattern arrays as they come through the C ABI (csc.p / csc.i).
 * perm_in: optional user permutation (length n+m), NULL -> built-in minimum-degree ordering.
 * Returns 0, or a negative code: -1 not upper triangular / bad index, -2 out of memory,
 * -3 invalid permutation. */
int rldl_symbolic_create(rldl_symbolic **out, long long n, long long m, const long long *Pp,
                         const long long *Pi, const long long *Ap, const long long *Ai, int polish,
                         const long long *perm_in);
void rldl_symbolic_free(rldl_symbolic *s);

/* Build the grouped solve plan for s (called by rldl_symbolic_create).  Indices 0..N-1 are cut into
 * consecutive groups of at most 64 (a dense trailing block becomes one group).  L entries whose row and
 * column fall in the same group live in a packed, zero-padded dense triangle per group (swept with
 * wave-level broadcasts on the device); all other entries are "out-of-group" and are gathered per lane
 * (by row in the forward sweep, by column in the backward sweep).  Returns 0, or -2 on out of memory. */
int rldl_plan_build(rldl_symbolic *s);

/* fill-reducing ordering of a symmetric pattern given by its upper triangle (CSC, n columns) */
int rldl_order_min_degree(int n, const int *Ap, const int *Ai, int *perm);

/* closed-form stage-interleaved permutation of src/recursive_ldl.c:1350-1362 (compute_permutations) */
void rldl_stage_permutation(long long N, long long nx, long long nu, long long ny, long long nt,
                            long long *perm);

#ifdef __cplusplus
}
#endif
#endif
