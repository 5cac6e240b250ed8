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
  FRI(G->sv_pk); FRI(G->sv_prog); FRI(G->pv_tab); FRI(G->pv_prog); FRI(G->pv_tinfo); FRI(G->pv_blk); FRI(G->pv_src); FRI(G->pv_dpos); FRI(G->pv_cptr); FRI(G->pv_cidx);
#undef FRI
  memset(G, 0, sizeof(*G));
  free(h->rec); h->rec = 0;
  free(h->pv_tiD); h->pv_tiD = 0;
  free(h->pv_grp); free(h->pv_tinfo_h); free(h->pv_blk_h);
  h->pv_grp = h->pv_tinfo_h = h->pv_blk_h = 0; h->pv_ngrp = 0;
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

/* Tables of the product tri-solve (stage_prod_solve / k_stage_invert, layout in rldl_device.h).  The pattern of a diagonal
 * tile is the pattern of the inverse of L_bb (its transitive closure inside the block).  Per tile: a row with n entries gets
 * ceil(n / K) lanes of its own (K = steps of the tile = 4 x its groups, the fewest groups for which 64 lanes are enough) and
 * deals its entries round robin to them; a lane takes one entry per step.  The entries of a step sit in Ti in LANE ORDER, so
 * the kernel finds a lane's entry from the step's 64-bit lane mask alone (v_mbcnt), and the same mask predicates the backward
 * pass's atomics.  Steps come in GROUPS of four, the unit of the kernel's load pipeline.  Which of its entries a lane takes in
 * which step is chosen step by step against the column use of the step: the backward pass adds into the columns with LDS
 * atomics, whose cost grows with the number of lanes on one word.  pv_prog is the kernel's STEP SEQUENCE: the groups in
 * forward order, closing (empty) groups up to a multiple of the ring size, the groups in backward order, closing groups. */
#define PV_KMAX 12
#define PV_GS 4                                                     /* steps per group */
#define PV_DW 12                                                    /* descriptor words per group */
typedef struct {                                                     /* host arrays of the product tri-solve (owned; prod_tiles_free) */
  unsigned *tab; int *seq, *grp, *tinfo, *blk, *tiD, *dposT, *cidx, *cptr; unsigned short *src;
  int ntab, nsteps, ntiles, ngroups, kmax, nTi, ldT, mode;
} prod_tiles_t;
static void prod_tiles_free(prod_tiles_t *T) {
  free(T->tab); free(T->seq); free(T->grp); free(T->tinfo); free(T->blk); free(T->tiD); free(T->dposT); free(T->cidx); free(T->cptr); free(T->src);
  memset(T, 0, sizeof(*T));
}

/* byte offset (from x[0]) of the auxiliary block vector of the mode-2 product tri-solve in a wave's LDS: behind x and its spare words */
static int pv_aux_offset(int N) { return 8 * (((N + 1) & ~1) + 2); }

/* The sequence of groups the solve kernel walks, padded to whole rings: [forward part | padding | backward part | padding | one ring
 * of empty closing groups].  grp: the group descriptors in tile order (D_0, C_0, D_1, ...; PV_DW words each), tinfo / blk: tile tables;
 * only the first nb_act blocks take part (single-store horizon handles; the coupling tile from the last live block onward is left out).
 *   mode 1   forward = the tiles in order, backward = the same groups in reverse (the kernel multiplies by the transposed tiles);
 *   mode 2   the coupling tiles hold the ORIGINAL coupling blocks K(b+1, b) of the permuted KKT matrix instead of L(b+1, b) =
 *            K(b+1, b) L_bb^-T D_b^-1, so every diagonal tile is used twice per pass while it sits in its ring slot and nothing but
 *            D tiles and the sparse K blocks is streamed.  With t_b = L_bb^-T D_b^-1 y_b in an auxiliary block vector a:
 *              forward, b ascending    [K-fwd_{b-1}: x_b -= K(b, b-1) a]   D-fwd_b: x_b <- L_bb^-1 x_b (= y_b);  a <- L_bb^-T D_b^-1 y_b
 *              backward, b descending  [K-bwd_b: a <- -K(b+1, b)' x_{b+1}]  D-bwd_b: a <- L_bb^-1 a;  x_b <- L_bb^-T D_b^-1 (x_b + a)
 *            flags word: bits 0-1 first / last group of its tile, bits 2-3 kind (0 K-fwd, 1 D-fwd, 2 K-bwd, 3 D-bwd), bit 4 the D-bwd
 *            step starts from a = 0 (no coupling tile below the block), bits 8-13 size of the block whose vector a holds.  A D step
 *            applies its tile twice (L_bb^-1 to one vector, then L_bb^-T to the other): a one-group tile does both from its one ring
 *            slot; a tile of several groups is listed twice, first half (bit 5, groups in order, the scaling behind the last one) and
 *            second half (bit 6, groups in reverse).  Blocks without a diagonal tile get an empty group that carries the scaling. */
static int *pv_sequence(int mode, const int *grp, int ngrp, const int *tinfo, int ntiles, const int *blk, const int *bs, int nb, int nb_act,
                        int N, int *nsteps_out) {
  int *seq = 0, cnt = 0, NGp = 0, b, i, pass;
  if (mode != 2) {
    int g_act = ngrp, kind;
    for (b = nb_act - 1; b < nb && g_act == ngrp; b++)
      for (kind = (b == nb_act - 1 ? 1 : 0); kind < 2; kind++) {
        const int t = blk[2 * b + kind];
        if (t >= 0) { g_act = tinfo[4 * t + 3]; break; }
      }
    NGp = ((g_act + RLDL_PV_RING - 1) / RLDL_PV_RING) * RLDL_PV_RING;
    seq = (int *)calloc((size_t)PV_DW * (size_t)(2 * NGp + 2 * RLDL_PV_RING), sizeof(int));
    if (!seq) return 0;
    for (i = 0; i < g_act; i++) memcpy(seq + PV_DW * i, grp + PV_DW * i, sizeof(int) * PV_DW);
    for (i = 0; i < g_act; i++) memcpy(seq + PV_DW * (NGp + i), grp + PV_DW * (g_act - 1 - i), sizeof(int) * PV_DW);
    *nsteps_out = 2 * NGp;
    return seq;
  }
  for (pass = 0; pass < 2; pass++) {                                 /* pass 0 counts, pass 1 fills */
    int pos = 0, half;
    for (half = 0; half < 2; half++) {
      if (pass == 1 && half == 1) pos = NGp;
      for (i = 0; i < nb_act; i++) {
        const int bb = half == 0 ? i : nb_act - 1 - i, sz = bs[bb + 1] - bs[bb];
        const int cb = half == 0 ? bb - 1 : bb;                       /* the block whose vector a holds around the coupling step: the tile's columns */
        const int tk = half == 0 ? (bb > 0 ? blk[2 * (bb - 1) + 1] : -1) : (bb < nb_act - 1 ? blk[2 * bb + 1] : -1);   /* coupling tile in front of the D step */
        const int td = blk[2 * bb];
        if (tk >= 0) {
          const int g0 = tinfo[4 * tk + 3], gend = tk + 1 < ntiles ? tinfo[4 * (tk + 1) + 3] : ngrp;
          int gk;
          for (gk = 0; gk < gend - g0; gk++) {
            if (pass == 1) {
              int *q = seq + PV_DW * pos;
              memcpy(q, grp + PV_DW * (half == 0 ? g0 + gk : gend - 1 - gk), sizeof(int) * PV_DW);   /* (backward: a tile's last group first) */
              q[2] = pv_aux_offset(N) | ((8 * bs[cb + 1]) << 16);
              q[3] = (q[3] & 3) | ((half == 0 ? 0 : 2) << 2) | ((bs[cb + 1] - bs[cb]) << 8);
            }
            pos++;
          }
        }
        {                                                             /* the D step: one group = both products from one ring slot; more groups = listed twice */
          const int g0 = td >= 0 ? tinfo[4 * td + 3] : 0, gend = td >= 0 ? (td + 1 < ntiles ? tinfo[4 * (td + 1) + 3] : ngrp) : 1;
          const int ng = gend - g0, kindbits = ((half == 0 ? 1 : 3) << 2) | (sz << 8);
          int gk;
          if (ng == 1) {
            if (pass == 1) {
              int *q = seq + PV_DW * pos;
              if (td >= 0) memcpy(q, grp + PV_DW * g0, sizeof(int) * PV_DW);
              else { memset(q, 0, sizeof(int) * PV_DW); q[2] = (8 * bs[bb]) | ((8 * bs[bb]) << 16); }
              q[3] = 3 | kindbits | ((half == 1 && tk < 0) ? 16 : 0);
            }
            pos++;
          } else {
            for (gk = 0; gk < 2 * ng; gk++) {
              if (pass == 1) {
                int *q = seq + PV_DW * pos;
                const int gsrc = gk < ng ? g0 + gk : gend - 1 - (gk - ng);
                memcpy(q, grp + PV_DW * gsrc, sizeof(int) * PV_DW);
                q[3] = (q[3] & 3) | kindbits | (gk < ng ? 32 : 64) | ((half == 1 && tk < 0 && gk == 0) ? 16 : 0);
              }
              pos++;
            }
          }
        }
      }
      if (half == 0 && pass == 0) cnt = pos;
    }
    if (pass == 0) {
      NGp = ((cnt + RLDL_PV_RING - 1) / RLDL_PV_RING) * RLDL_PV_RING;
      seq = (int *)calloc((size_t)PV_DW * (size_t)(2 * NGp + 2 * RLDL_PV_RING), sizeof(int));
      if (!seq) return 0;
    }
  }
  *nsteps_out = 2 * NGp;
  return seq;
}

/* 0: built, 1: the pattern does not qualify */
static int *pv_sequence(int mode, const int *grp, int ngrp, const int *tinfo, int ntiles, const int *blk, const int *bs, int nb, int nb_act,
                        int N, int *nsteps_out);
static int prod_tiles_host(int mode, int smax, int ldF, int N, const int *bs, int nb, int ld, const int *dptr, const int *dpos, const int *cptr,
                           const int *cslot, const int *cpos, prod_tiles_t *out) {
  const int ntmax = 2 * nb, ngmax = 2 * nb * (PV_KMAX / PV_GS);
  const int ldT = (smax <= 8 ? 8 : smax <= 16 ? 16 : smax <= 24 ? 24 : 32) + 2;   /* row length of k_stage_invert's tiles (its SM + 2) */
  unsigned char *pat = 0;                       /* [smax][smax] pattern of the tile at hand */
  int *rows_e = 0, *cnt = 0, *used = 0, *prog = 0, *seq = 0, *blk = 0, *tinfo = 0, *ent_src = 0, *colcnt = 0, *order = 0, *tiD = 0;
  int ssrc[PV_KMAX * 64];                                            /* source of the entry of (step, lane) of the tile at hand, -1 = none */
  unsigned *tab = 0;
  unsigned short *src = 0;
  int b, t = 0, g = 0, nTi = 0, ntab = 0, kmax = 0, r, c, k, e, kind, ok = 0, i, NGp, nsteps = 0;
  size_t tabcap = (size_t)ngmax * 64 + 256, srccap = 0;
  memset(out, 0, sizeof(*out));
  if (ldF >= 65536 || (N + 2) * 8 >= 65536 || smax > 32) return 1;
  for (b = 0; b < nb; b++) srccap += (size_t)smax * (size_t)smax;
  srccap += (size_t)cptr[nb];
  srccap = 2 * srccap + 256;                                         /* (mode 2 pads its pair slots) */
  pat = (unsigned char *)malloc((size_t)smax * smax);
  rows_e = (int *)malloc(sizeof(int) * (size_t)smax * smax);     /* entries of the tile: column per (row, i) */
  ent_src = (int *)malloc(sizeof(int) * (size_t)smax * smax);    /* ... and where the value comes from */
  cnt = (int *)malloc(sizeof(int) * (size_t)smax);
  order = (int *)malloc(sizeof(int) * (size_t)smax);
  used = (int *)malloc(sizeof(int) * (size_t)smax * smax);
  colcnt = (int *)malloc(sizeof(int) * (size_t)smax);
  prog = (int *)calloc((size_t)PV_DW * (size_t)(ngmax + 1), sizeof(int));
  seq = (int *)calloc((size_t)PV_DW * (size_t)(2 * ngmax + 4 * RLDL_PV_RING), sizeof(int));
  tinfo = (int *)calloc((size_t)4 * (size_t)(ntmax + 1), sizeof(int));
  blk = (int *)malloc(sizeof(int) * (size_t)(2 * nb));
  tab = (unsigned *)calloc(tabcap, sizeof(unsigned));
  src = (unsigned short *)malloc(sizeof(unsigned short) * (srccap + 1));
  if (!pat || !rows_e || !ent_src || !cnt || !order || !used || !colcnt || !prog || !seq || !tinfo || !blk || !tab || !src) goto out;
  ntab = 64;                                                          /* words [0, 64): zero = the closing groups' table words (no entries) */
  tiD = (int *)malloc(sizeof(int) * (size_t)(nb + 1));
  if (!tiD) goto out;
  for (b = 0; b < nb; b++) {
    tiD[b] = nTi;
    for (kind = 0; kind < 2; kind++) {
      const int s = bs[b + 1] - bs[b];                                  /* columns of the tile */
      const int R = kind == 0 ? s : (b + 1 < nb ? bs[b + 2] - bs[b + 1] : 0);
      const int rowbase = kind == 0 ? bs[b] : bs[b + 1], colbase = bs[b];
      int E = 0, K = 0, nzr = 0, NG = 0, lanes = 0;
      int lane_row[64], lane_p[64], lane_h[64];
      blk[2 * b + kind] = -1;
      if (R <= 0) continue;
      /* entries per row: column + source */
      for (r = 0; r < R; r++) cnt[r] = 0;
      if (kind == 0) {
        memset(pat, 0, (size_t)smax * smax);
        for (e = dptr[b]; e < dptr[b + 1]; e++) pat[(dpos[e] / ld) * smax + dpos[e] % ld] = 1;
        for (r = 0; r < s; r++)                                           /* closure: inv(r, c) if L(r, k) and (k == c or inv(k, c)) */
          for (c = 0; c < r; c++) {
            int hit = pat[r * smax + c];
            for (k = c + 1; k < r && !hit; k++) if (pat[r * smax + k] == 1 && pat[k * smax + c]) hit = 1;
            if (hit && !pat[r * smax + c]) pat[r * smax + c] = 2;        /* 2: fill of the inverse (counts as an entry from here on) */
          }
        for (r = 0; r < s; r++)
          for (c = 0; c < r; c++)
            if (pat[r * smax + c]) { rows_e[r * smax + cnt[r]] = c; ent_src[r * smax + cnt[r]] = r * ldT + c; cnt[r]++; }
      } else {
        for (e = cptr[b]; e < cptr[b + 1]; e++) {
          r = cpos[e] / ld; c = cpos[e] % ld;
          rows_e[r * smax + cnt[r]] = c; ent_src[r * smax + cnt[r]] = cslot[e]; cnt[r]++;
        }
      }
      for (r = 0; r < R; r++) { E += cnt[r]; if (cnt[r]) order[nzr++] = r; }
      if (!E) continue;
      for (i = 1; i < nzr; i++) {                                         /* rows by entries descending (insertion sort, stable) */
        const int v = order[i];
        for (k = i; k > 0 && cnt[order[k - 1]] < cnt[v]; k--) order[k] = order[k - 1];
        order[k] = v;
      }
      for (NG = 1; NG <= PV_KMAX / PV_GS; NG++) {                         /* fewest groups for which 64 lanes are enough */
        K = NG * PV_GS;
        for (lanes = 0, i = 0; i < nzr; i++) lanes += (cnt[order[i]] + K - 1) / K;
        if (lanes <= 64) break;
      }
      if (NG > PV_KMAX / PV_GS || E >= 65535 || t >= ntmax || g + NG > ngmax || s > 32 || R > 32) goto out;
      if (K > kmax) kmax = K;
      for (lanes = 0, i = 0; i < nzr; i++) {
        const int H = (cnt[order[i]] + K - 1) / K;
        for (k = 0; k < H; k++) { lane_row[lanes] = order[i]; lane_p[lanes] = k; lane_h[lanes] = H; lanes++; }
      }
      /* one word per (group, lane): columns of the lane's four entries (5 bits each, relative to the tile's first column) |
       * one bit per step << 20 (the lane has an entry) | row of the lane (relative to the tile's first row) << 24 | 1 << 29 for
       * lanes with entries in this tile */
      memset(used, 0, sizeof(int) * (size_t)smax * smax);
      for (i = 0; i < PV_KMAX * 64; i++) ssrc[i] = -1;
      e = 0;                                                              /* running Ti offset inside the tile */
      for (k = 0; k < K; k++) {
        const int gi = g + k / PV_GS, j = k % PV_GS;
        int *q = prog + PV_DW * gi, lane;
        unsigned *gw;
        if (j == 0) {
          q[0] = nTi + e; q[1] = ntab; q[2] = (8 * colbase) | ((8 * rowbase) << 16);
          q[3] = (k == 0 ? 1 : 0) | (k / PV_GS == NG - 1 ? 2 : 0);
          for (i = 4; i < PV_DW; i++) q[i] = 0;
          for (lane = 0; lane < lanes; lane++) tab[ntab + lane] = ((unsigned)lane_row[lane] << 24) | (1u << 29);
          ntab += 64;
        }
        gw = tab + q[1];
        for (c = 0; c < s; c++) colcnt[c] = 0;
        for (lane = 0; lane < lanes; lane++) {
          const int rr = lane_row[lane], p = lane_p[lane], H = lane_h[lane];
          const int n = cnt[rr] > p ? (cnt[rr] - p + H - 1) / H : 0;
          int best = -1, ii;
          if (k >= n) continue;
          for (ii = p; ii < cnt[rr]; ii += H)
            if (!used[rr * smax + ii] && (best < 0 || colcnt[rows_e[rr * smax + ii]] < colcnt[rows_e[rr * smax + best]])) best = ii;
          used[rr * smax + best] = 1;
          c = rows_e[rr * smax + best];
          colcnt[c]++;
          gw[lane] |= ((unsigned)c << (5 * j)) | (1u << (20 + j));
          q[4 + 2 * j + (lane >> 5)] |= (int)(1u << (lane & 31));         /* the step's lane mask */
          src[nTi + e] = (unsigned short)ent_src[rr * smax + best];
          ssrc[k * 64 + lane] = ent_src[rr * smax + best];
          e++;
        }
      }
      if (e != E) goto out;
      if (mode == 2) {
        /* PAIR LAYOUT: the two entries a lane takes in steps (2 p, 2 p + 1) of a group sit side by side, lanes of step 2 p in lane order
         * (a lane's entries fill its steps from 0 up, so a lane of step 2 p + 1 is a lane of step 2 p; a lane without a second entry
         * gets a padding slot, source 0xffff, value 0.0): one 16-byte load per lane and pair instead of two 8-byte loads */
        int gi2, pp, lane, eo = 0;
        for (gi2 = 0; gi2 < NG; gi2++) {
          prog[PV_DW * (g + gi2)] = nTi + eo;
          for (pp = 0; pp < PV_GS / 2; pp++) {
            const int k0 = gi2 * PV_GS + 2 * pp;
            for (lane = 0; lane < lanes; lane++) {
              if (ssrc[k0 * 64 + lane] < 0) { if (ssrc[(k0 + 1) * 64 + lane] >= 0) goto out; continue; }
              src[nTi + eo++] = (unsigned short)ssrc[k0 * 64 + lane];
              src[nTi + eo++] = ssrc[(k0 + 1) * 64 + lane] >= 0 ? (unsigned short)ssrc[(k0 + 1) * 64 + lane] : (unsigned short)0xffffu;
            }
          }
        }
        E = eo;
        if ((size_t)(nTi + E) >= srccap) goto out;
      }
      tinfo[4 * t] = nTi; tinfo[4 * t + 1] = E; tinfo[4 * t + 2] = kind; tinfo[4 * t + 3] = g;
      blk[2 * b + kind] = t;
      nTi += E; t++; g += NG;
    }
  }
  /* the step sequence the kernel walks (pv_sequence), one ring of closing groups behind it for the loads issued ahead */
  free(seq);
  seq = pv_sequence(mode, prog, g, tinfo, t, blk, bs, nb, nb, N, &nsteps);
  if (!seq) goto out;
  (void)NGp;
  tiD[nb] = nTi;
  out->dposT = (int *)malloc(sizeof(int) * (size_t)(dptr[nb] + 1));    /* ld_pos in the geometry of k_stage_invert's tiles */
  if (!out->dposT) goto out;
  for (e = 0; e < dptr[nb]; e++) out->dposT[e] = (dpos[e] / ld) * ldT + dpos[e] % ld;
  out->ldT = ldT;
  /* the Ti entries of the coupling tiles, block after block: k_stage_invert copies them from their factor slots in one batched loop */
  out->cptr = (int *)malloc(sizeof(int) * (size_t)(nb + 1));
  {
    int tot = 1;
    for (b = 0; b < nb; b++) if (blk[2 * b + 1] >= 0) tot += tinfo[4 * blk[2 * b + 1] + 1];
    out->cidx = (int *)malloc(sizeof(int) * (size_t)tot);
  }
  if (!out->cptr || !out->cidx) { free(out->dposT); free(out->cptr); free(out->cidx); out->dposT = out->cptr = out->cidx = 0; goto out; }
  for (b = 0, k = 0; b < nb; b++) {
    out->cptr[b] = k;
    if (blk[2 * b + 1] >= 0) { const int tc = blk[2 * b + 1]; for (e = 0; e < tinfo[4 * tc + 1]; e++) out->cidx[k++] = tinfo[4 * tc] + e; }
  }
  out->cptr[nb] = k;
  out->tab = tab; out->seq = seq; out->tinfo = tinfo; out->blk = blk; out->tiD = tiD; out->src = src;
  out->grp = (int *)malloc(sizeof(int) * (size_t)PV_DW * (size_t)(g + 1));
  if (!out->grp) { free(out->dposT); free(out->cptr); free(out->cidx); out->dposT = out->cptr = out->cidx = 0; goto out; }
  memcpy(out->grp, prog, sizeof(int) * (size_t)PV_DW * (size_t)g);
  out->ntab = ntab; out->nsteps = nsteps; out->ntiles = t; out->ngroups = g; out->kmax = kmax; out->nTi = nTi; out->mode = mode;
  tab = 0; seq = 0; tinfo = 0; blk = 0; tiD = 0; src = 0;
  ok = 1;
out:
  free(pat); free(rows_e); free(ent_src); free(cnt); free(order); free(used); free(colcnt); free(prog); free(seq); free(tinfo); free(blk); free(tab); free(src); free(tiD);
  return ok ? 0 : 1;
}


/* the same tables on the device (optional: pv_ok stays 0 on failure) */
static void build_prod_tiles(rldl_batch *h, const int *bs, int nb, int ld, const int *dptr, const int *dpos, const int *cptr,
                             const int *cslot, const int *cpos, const int *kptr, const int *ksrc, const int *kpos) {
  rldl_dev_stage *G = &h->dsym.stage;
  prod_tiles_t T;
  G->pv_ok = 0; G->pv_mode = 0;
  if (getenv("RLDL_NO_STAGE_PROD")) return;
  /* mode 2 (coupling tiles = the original coupling blocks of the KKT matrix, every diagonal tile used twice per pass: about half the
   * bytes per solve) when every diagonal tile fits one group and the KKT value indices fit the 16-bit source table; else mode 1 */
  if (getenv("RLDL_PROD_V1") || h->sym->nnzK >= 65535 || prod_tiles_host(2, G->smax, h->dsym.ldF, h->sym->N, bs, nb, ld, dptr, dpos, kptr, ksrc, kpos, &T))
    if (prod_tiles_host(1, G->smax, h->dsym.ldF, h->sym->N, bs, nb, ld, dptr, dpos, cptr, cslot, cpos, &T)) return;
  G->pv_tab = (const unsigned *)upload_ints((const int *)T.tab, (size_t)T.ntab);
  G->pv_prog = upload_ints(T.seq, (size_t)PV_DW * (size_t)(T.nsteps + 2 * RLDL_PV_RING));
  G->pv_tinfo = upload_ints(T.tinfo, (size_t)4 * (size_t)(T.ntiles + 1));
  G->pv_blk = upload_ints(T.blk, (size_t)2 * (size_t)nb);
  G->pv_dpos = upload_ints(T.dposT, (size_t)dptr[nb]);
  G->pv_cptr = upload_ints(T.cptr, (size_t)nb + 1);
  G->pv_cidx = upload_ints(T.cidx, (size_t)T.cptr[nb]);
  {
    unsigned short *d = 0;
    if (hipMalloc((void **)&d, sizeof(unsigned short) * (size_t)(T.nTi + 1)) == hipSuccess) {
      if (hipMemcpy(d, T.src, sizeof(unsigned short) * (size_t)T.nTi, hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(d); d = 0; }
    } else d = 0;
    G->pv_src = d;
  }
  if (G->pv_tab && G->pv_prog && G->pv_tinfo && G->pv_blk && G->pv_src && G->pv_dpos && G->pv_cptr && G->pv_cidx) {
    G->pv_ldT = T.ldT;
    G->pv_ntiles = T.ntiles; G->pv_ngroups = T.ngroups; G->pv_nsteps = T.nsteps; G->pv_kmax = T.kmax; G->pv_nTi = T.nTi; G->pv_ntab = T.ntab;
    G->pv_ldTi = (T.nTi + 1) & ~1;
    free(h->pv_tiD); h->pv_tiD = T.tiD; T.tiD = 0;
    G->pv_ok = 1; G->pv_mode = T.mode; G->pv_aux = pv_aux_offset(h->sym->N);
    /* host copies for step programs over a prefix of the blocks (the first T.ngroups entries of seq are the groups in tile order) */
    free(h->pv_grp); free(h->pv_tinfo_h); free(h->pv_blk_h);
    h->pv_grp = (int *)malloc(sizeof(int) * (size_t)PV_DW * (size_t)(T.ngroups + 1));
    h->pv_tinfo_h = (int *)malloc(sizeof(int) * 4 * (size_t)(T.ntiles + 1));
    h->pv_blk_h = (int *)malloc(sizeof(int) * 2 * (size_t)nb);
    if (h->pv_grp && h->pv_tinfo_h && h->pv_blk_h) {
      memcpy(h->pv_grp, T.grp, sizeof(int) * (size_t)PV_DW * (size_t)T.ngroups);
      memcpy(h->pv_tinfo_h, T.tinfo, sizeof(int) * 4 * (size_t)T.ntiles);
      memcpy(h->pv_blk_h, T.blk, sizeof(int) * 2 * (size_t)nb);
      h->pv_ngrp = T.ngroups; h->pv_ntiles_h = T.ntiles;
    } else { free(h->pv_grp); free(h->pv_tinfo_h); free(h->pv_blk_h); h->pv_grp = h->pv_tinfo_h = h->pv_blk_h = 0; h->pv_ngrp = 0; }
  }
  prod_tiles_free(&T);
}

int rldl_stage_prog_prefix(const rldl_batch *h, int nb_act, int **d_prog, int *nsteps) {
  const rldl_dev_stage *G;
  int *seq;
  if (!h || !d_prog || !nsteps || !h->rec) return 1;
  G = &h->dsym.stage;
  if (!G->pv_ok || !h->pv_grp || nb_act < 1 || nb_act > G->nb) return 1;
  seq = pv_sequence(G->pv_mode, h->pv_grp, h->pv_ngrp, h->pv_tinfo_h, h->pv_ntiles_h, h->pv_blk_h, (const int *)h->rec + 1, G->nb, nb_act,
                    h->sym->N, nsteps);
  if (!seq) return 1;
  *d_prog = upload_ints(seq, (size_t)PV_DW * (size_t)(*nsteps + 2 * RLDL_PV_RING));
  free(seq);
  return *d_prog ? 0 : 1;
}

/* Host part of the stage-block view: block starts and, per block, the entries of the permuted KKT matrix and of L that fall into
 * the diagonal block and into the coupling block below it.  lists: 0 = K diagonal, 1 = K coupling, 2 = L diagonal, 3 = L coupling;
 * a[k][e] = source (Kx index / factor slot), b[k][e] = tile position row * ld + col.  0: ok, 1: the pattern does not qualify. */
typedef struct { int nb, smax, ld, *bs, *ptr[4], *a[4], *b[4], tot[4]; } stage_lists_t;
static void stage_lists_free(stage_lists_t *L) {
  int k;
  free(L->bs);
  for (k = 0; k < 4; k++) { free(L->ptr[k]); free(L->a[k]); free(L->b[k]); }
  memset(L, 0, sizeof(*L));
}
static int stage_lists(const rldl_symbolic *s, const rldl_stage_dims *d, stage_lists_t *L) {
  const int N = s->N, nb = 2 * (int)d->N + 2;
  int *bs = 0, *blk = 0, *cnt = 0, *fill = 0, **ptr = L->ptr, **a = L->a, **b = L->b, *tot = L->tot;
  int i, j, k, p, smax = 0, ld, rc = 1;
  memset(L, 0, sizeof(*L));
  bs = (int *)malloc(sizeof(int) * (size_t)(nb + 2));
  blk = (int *)malloc(sizeof(int) * (size_t)(N + 1));
  cnt = (int *)calloc((size_t)4 * (size_t)(nb + 1), sizeof(int));
  fill = (int *)calloc((size_t)4 * (size_t)(nb + 1), sizeof(int));
  L->bs = bs;
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
  /* pass 1: classify and count */
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
  L->nb = nb; L->smax = smax; L->ld = ld;
  rc = 0;
out:
  free(blk); free(cnt); free(fill);
  if (rc) stage_lists_free(L);
  return rc;
}

/* 0: maps built and uploaded (h->dsym.stage.nb > 0); 1: the pattern does not qualify (generic kernels stay in use) */
static int build_stage_maps(rldl_batch *h) {
  stage_lists_t L;
  rldl_dev_stage G;
  int rc = 1, nb, **ptr = L.ptr, **a = L.a, **b = L.b, *tot = L.tot;
  if (stage_lists(h->sym, &h->stage, &L)) return 1;
  nb = L.nb;
  memset(&G, 0, sizeof(G));
  G.nb = nb; G.ld = L.ld; G.smax = L.smax;
  G.bs = upload_ints(L.bs, (size_t)nb + 1);
  G.kd_ptr = upload_ints(ptr[0], (size_t)nb + 1); G.kd_src = upload_ints(a[0], (size_t)tot[0]); G.kd_pos = upload_ints(b[0], (size_t)tot[0]);
  G.kc_ptr = upload_ints(ptr[1], (size_t)nb + 1); G.kc_src = upload_ints(a[1], (size_t)tot[1]); G.kc_pos = upload_ints(b[1], (size_t)tot[1]);
  G.ld_ptr = upload_ints(ptr[2], (size_t)nb + 1); G.ld_slot = upload_ints(a[2], (size_t)tot[2]); G.ld_pos = upload_ints(b[2], (size_t)tot[2]);
  G.lc_ptr = upload_ints(ptr[3], (size_t)nb + 1); G.lc_slot = upload_ints(a[3], (size_t)tot[3]); G.lc_pos = upload_ints(b[3], (size_t)tot[3]);
  h->dsym.stage = G;
  if (!G.bs || !G.kd_ptr || !G.kd_src || !G.kd_pos || !G.kc_ptr || !G.kc_src || !G.kc_pos || !G.ld_ptr || !G.ld_slot || !G.ld_pos ||
      !G.lc_ptr || !G.lc_slot || !G.lc_pos) { rldl_stage_maps_free(h); goto out; }
  build_solve_tiles(h, L.bs, nb, L.ld, ptr[2], a[2], b[2], tot[2], ptr[3], a[3], b[3], tot[3]);   /* optional: sv_ok stays 0 on failure */
  if (h->dsym.stage.sv_ok) build_prod_tiles(h, L.bs, nb, L.ld, ptr[2], b[2], ptr[3], a[3], b[3], ptr[1], a[1], b[1]);   /* optional as well */
  h->rec = malloc(sizeof(int) * (size_t)(nb + 2));
  if (!h->rec) { rldl_stage_maps_free(h); goto out; }
  ((int *)h->rec)[0] = nb;
  memcpy((int *)h->rec + 1, L.bs, sizeof(int) * (size_t)(nb + 1));
  rc = 0;
out:
  stage_lists_free(&L);
  return rc;
}

/* Host-only export of the product tri-solve's tables for a stage-structured pattern (no device needed): the symbolic analysis on
 * the stage-interleaved order, the block lists and the tables exactly as a handle would upload them.  meta[8] = { usable, tiles,
 * table words, Ti entries, nb, ld (tile positions in src are row * ld + col), kmax, steps }; with null arrays only meta is filled.
 * src: D-tile entries = position in the block's tile, C-tile entries = factor slot (LtoS maps the CSC entries of L to slots). */
c_int rldl_stage_prod_export(const csc *P, const csc *A, const rldl_stage_dims *dims, c_int *meta, int *prog, int *tinfo, unsigned *tab,
                             unsigned short *src, int *blk, c_int *LtoS) {
  rldl_symbolic *s = 0;
  stage_lists_t L;
  prod_tiles_t T;
  c_int *perm, i, rc = 1;
  if (!P || !A || !dims || !meta) return 1;
  for (i = 0; i < 8; i++) meta[i] = 0;
  perm = (c_int *)malloc(sizeof(c_int) * (size_t)(P->n + A->m));
  if (!perm) return RLDL_MEM_ALLOC_ERROR;
  rldl_stage_permutation(dims->N, dims->nx, dims->nu, dims->ny, dims->nt, perm);
  if (rldl_symbolic_create(&s, P->n, A->m, P->p, P->i, A->p, A->i, 0, perm)) { free(perm); return RLDL_LINSYS_SOLVER_INIT_ERROR; }
  free(perm);
  if (stage_lists(s, dims, &L)) { rldl_symbolic_free(s); return 2; }
  if (!prod_tiles_host(1, L.smax, (s->nS + s->N + 1) & ~1, s->N, L.bs, L.nb, L.ld, L.ptr[2], L.b[2], L.ptr[3], L.a[3], L.b[3], &T)) {   /* (the mode-1 tables: L(b+1, b) in the coupling tiles) */
    meta[0] = 1; meta[1] = T.ntiles; meta[2] = T.ntab; meta[3] = T.nTi; meta[4] = L.nb; meta[5] = T.ldT; meta[6] = T.kmax; meta[7] = T.nsteps;
    if (prog) memcpy(prog, T.seq, sizeof(int) * PV_DW * (size_t)T.nsteps);
    if (tinfo) memcpy(tinfo, T.tinfo, sizeof(int) * 4 * (size_t)(T.ntiles + 1));
    if (tab) memcpy(tab, T.tab, sizeof(unsigned) * (size_t)T.ntab);
    if (src) memcpy(src, T.src, sizeof(unsigned short) * (size_t)T.nTi);
    if (blk) memcpy(blk, T.blk, sizeof(int) * 2 * (size_t)L.nb);
    if (LtoS) for (i = 0; i < s->nnzL; i++) LtoS[i] = s->LtoS[i];
    prod_tiles_free(&T);
    rc = 0;
  } else rc = 2;
  stage_lists_free(&L);
  rldl_symbolic_free(s);
  return rc;
}

/* Mark a handle as stage-structured: enables restart-from-stage and, when the pattern qualifies, the dense
 * stage-block factorisation for every later numeric factorisation of the handle. */
void rldl_batch_enable_stage(rldl_batch *h, const rldl_stage_dims *dims) {
  h->stage = *dims;
  h->recursive = 1;
  if (!getenv("RLDL_NO_STAGE_FACTOR")) (void)build_stage_maps(h);
  if (h->dsym.stage.pv_ok) {                                    /* tile values of the product tri-solve (k_stage_invert writes them) */
    const size_t bytes = sizeof(double) * (size_t)h->batch * (size_t)h->dsym.stage.pv_ldTi;
    if (h->num.Ti || hipMalloc((void **)&h->num.Ti, bytes) != hipSuccess) { if (!h->dsym.tile_ok) h->num.Ti = 0; h->dsym.stage.pv_ok = 0; }
    else if (hipMemset(h->num.Ti, 0, bytes) != hipSuccess) { (void)hipFree(h->num.Ti); h->num.Ti = 0; h->dsym.stage.pv_ok = 0; }
    /* a handle that was factorised before it became stage-structured (osqp_batch_setup_recursive): tiles from that factor */
    else if (rldl_launch_stage_invert(&h->dsym, &h->num, h->stream)) { (void)hipFree(h->num.Ti); h->num.Ti = 0; h->dsym.stage.pv_ok = 0; }
  }
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
