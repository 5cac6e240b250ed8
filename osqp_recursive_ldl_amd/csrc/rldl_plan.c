/*
 * rldl_plan.c -- host-side builder of the grouped triangular-solve plan (plain C99).
 *
 * The forward/back substitution of the reference (QDLDL_solve, call site qdldl_interface.c:553; loop
 * bodies src/recursive_ldl.c:62-116) walks L column by column.  On a 64-lane wavefront that is one
 * dependent LDS round trip per column.  The plan turns the same arithmetic into:
 *   - GROUPS of at most 64 consecutive indices (chosen by dynamic programming), one lane per index;
 *   - a per-lane GATHER over the entries that connect a group to other, already final, indices, laid out
 *     as jagged diagonals (rows sorted by length, so step t touches a dense prefix of lanes);
 *   - an in-group SWEEP over a packed dense triangle, where the pivot value travels between lanes with
 *     v_readlane instead of through memory.
 * The factor is stored on the device directly in the plan's slot order (what `solve` streams):
 *   slots [0, nO)   out-of-group entries in FORWARD jagged-diagonal order (group, step, lane)
 *   slots [nO, nS)  per-group triangles, ROW-MAJOR packed: local row il >= 1 holds its il entries L(il, 0..il-1)
 *                   at Tb + il (il-1)/2 (zero padded), so the forward sweep reads lane-constant base + step
 *                   (DS immediate offsets) and the backward sweep reads uniform base + lane
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "rldl_symbolic.h"

#define GROUP_MAX 64

typedef struct { int key, idx; } kv;
static int cmp_kv_desc(const void *a, const void *b) { /* by key descending, ties by index ascending */
  const kv *x = (const kv *)a, *y = (const kv *)b;
  if (x->key != y->key) return x->key > y->key ? -1 : 1;
  return x->idx < y->idx ? -1 : (x->idx > y->idx ? 1 : 0);
}

int rldl_plan_build(rldl_symbolic *s) {
  const int N = s->N;
  int *group_of = 0, *gstart = 0, *blob = 0, *rowcnt = 0, *colcnt = 0, *coloff = 0;
  int *fpos_of_row = 0, *bpos_of_col = 0, *fsteps_base = 0, *fsteps_cnt = 0, *bsteps_base = 0, *bsteps_cnt = 0,
      *fstep_ptr = 0, *bstep_ptr = 0, *slot_of_csr = 0;
  unsigned char *col_active = 0, *row_active = 0;
  kv *order = 0;
  int ng = 0, i, k, c, p, q, nO = 0, nOp = 0, tri = 0, na = 0, nr = 0, words, nfs = 0, nbs = 0, rc = -2;
  int arrow_k = -1, arrow_steps = 0, ngather = 0, ntri = 0, vpad = 0;

  s->plan_ok = 0; s->tile_ok = 0; s->tile_admm_ok = 0; s->tile_scatter_ok = 0;
  s->LtoS = (int *)malloc(sizeof(int) * (size_t)(s->nnzL > 0 ? s->nnzL : 1));
  group_of = (int *)malloc(sizeof(int) * (size_t)(N + 1));
  gstart = (int *)malloc(sizeof(int) * (size_t)(N + 2));
  col_active = (unsigned char *)calloc((size_t)N + 1, 1);
  row_active = (unsigned char *)calloc((size_t)N + 1, 1);
  coloff = (int *)malloc(sizeof(int) * (size_t)(N + 1));
  rowcnt = (int *)calloc((size_t)N + 2, sizeof(int));
  colcnt = (int *)calloc((size_t)N + 2, sizeof(int));
  fpos_of_row = (int *)malloc(sizeof(int) * (size_t)(N + 1));
  bpos_of_col = (int *)malloc(sizeof(int) * (size_t)(N + 1));
  order = (kv *)malloc(sizeof(kv) * (size_t)(GROUP_MAX + 1));
  slot_of_csr = (int *)malloc(sizeof(int) * (size_t)(s->nnzL > 0 ? s->nnzL : 1));
  if (!s->LtoS || !group_of || !gstart || !col_active || !row_active || !coloff || !rowcnt || !colcnt || !fpos_of_row ||
      !bpos_of_col || !order || !slot_of_csr)
    goto out;

  /* ---- grouping by dynamic programming over cut points ----
   * cost of a group [i, j): zero padding of its packed triangle + 4 per sweep step - 1 per in-group entry
   * (entries kept inside a group ride the register sweep instead of the per-lane gather) + a fixed
   * sequential overhead per group.  Dense blocks (the Schur-complement tail of a KKT matrix, the stage blocks
   * of an MPC problem) become groups; runs of mutually independent columns cost nothing wherever cut. */
  {
    double *best = (double *)malloc(sizeof(double) * (size_t)(N + 1));
    int *from = (int *)malloc(sizeof(int) * (size_t)(N + 1));
    int j;
    const double group_cost = getenv("RLDL_GROUP_COST") ? atof(getenv("RLDL_GROUP_COST")) : 24.0;
    if (!best || !from) { free(best); free(from); goto out; }
    best[0] = 0.0; from[0] = 0;
    for (j = 1; j <= N; j++) {
      double pad = 0.0, ein = 0.0, steps = 0.0;
      int lo = j - GROUP_MAX < 0 ? 0 : j - GROUP_MAX;
      best[j] = 1e300; from[j] = j - 1;
      for (i = j - 1; i >= lo; i--) {          /* grow the group [i, j) to the left by column i */
        int cnt = 0;
        for (p = s->Lp[i]; p < s->Lp[i + 1] && s->Li[p] < j; p++) cnt++;
        ein += cnt;
        {
          const double g = (double)(j - i);
          pad = ein > 0 ? 0.5 * g * (g - 1.0) - ein : 0.0;    /* the whole packed triangle is stored */
          steps = ein > 0 ? 2.0 * (g - 1.0) : 0.0;             /* forward + backward sweep steps */
          double cst = best[i] + pad + 4.0 * steps - ein + group_cost;
          if (cst < best[j]) { best[j] = cst; from[j] = i; }
        }
      }
    }
    ng = 0;
    for (j = N; j > 0; j = from[j]) ng++;
    k = ng;
    for (j = N; j > 0; j = from[j]) gstart[--k] = from[j];
    gstart[ng] = N;
    free(best); free(from);
  }
  for (k = 0; k < ng; k++)
    for (i = gstart[k]; i < gstart[k + 1]; i++) group_of[i] = k;

  /* ---- classify entries ---- */
  for (c = 0; c < N; c++)
    for (p = s->Lp[c]; p < s->Lp[c + 1]; p++) {
      int r = s->Li[p];
      if (group_of[r] == group_of[c]) { col_active[c] = 1; row_active[r] = 1; }
      else { nO++; rowcnt[r]++; colcnt[c]++; }
    }
  if (N >= 65536 || nO >= 65536) { rc = 0; goto out; }       /* packed 16-bit fields would overflow */

  /* triangle base slots (absolute); coloff[] is reused as "triangle base of the index's group" (-1: none).
   * The triangles start at an even slot so the part of the row behind the gather values is 16-byte aligned. */
  nOp = (nO + 1) & ~1;
  for (k = 0; k < ng; k++) {
    int g0 = gstart[k], g = gstart[k + 1] - g0, any = 0;
    for (i = 0; i < g; i++) any |= col_active[g0 + i];
    for (i = 0; i < g; i++) coloff[g0 + i] = any ? nOp + tri : -1;
    if (any) { tri += g * (g - 1) / 2; na += g - 1; nr += g - 1; }
  }
  s->nO = nO; s->nOp = nOp; s->nS = nOp + tri; s->ngroups = ng;

  /* ---- jagged-diagonal orders: count steps ---- */
  fstep_ptr = (int *)calloc((size_t)ng + 1, sizeof(int));
  bstep_ptr = (int *)calloc((size_t)ng + 1, sizeof(int));
  if (!fstep_ptr || !bstep_ptr) goto out;
  for (k = 0; k < ng; k++) {
    int g0 = gstart[k], g1 = gstart[k + 1], mf = 0, mb = 0;
    for (i = g0; i < g1; i++) { if (rowcnt[i] > mf) mf = rowcnt[i]; if (colcnt[i] > mb) mb = colcnt[i]; }
    fstep_ptr[k + 1] = fstep_ptr[k] + mf;
    bstep_ptr[k + 1] = bstep_ptr[k] + mb;
  }
  nfs = fstep_ptr[ng]; nbs = bstep_ptr[ng];
  fsteps_base = (int *)calloc((size_t)nfs + 1, sizeof(int)); fsteps_cnt = (int *)calloc((size_t)nfs + 1, sizeof(int));
  bsteps_base = (int *)calloc((size_t)nbs + 1, sizeof(int)); bsteps_cnt = (int *)calloc((size_t)nbs + 1, sizeof(int));
  if (!fsteps_base || !fsteps_cnt || !bsteps_base || !bsteps_cnt) goto out;

  /* ---- blob layout ---- */
  words = 0;
  s->po_gstart = words; words += ng + 1;
  s->po_gflag = words; words += ng + 1;
  s->po_gaptr = words; words += ng + 1;
  s->po_grptr = words; words += ng + 1;
  s->po_gToff = words; words += ng + 1;
  s->po_fsp = words; words += ng + 1;
  s->po_bsp = words; words += ng + 1;
  s->po_acol = s->po_aoff = s->po_arow = s->po_coloff = 0;   /* (generic column lists are gone: every triangle is full) */
  s->po_fsb = words; words += nfs;
  s->po_fsc = words; words += nfs;
  s->po_bsb = words; words += nbs;
  s->po_bsc = words; words += nbs;
  s->po_fsig = words; words += (N + 1) / 2;
  s->po_bsig = words; words += (N + 1) / 2;
  s->po_fcol = words; words += (nO + 1) / 2;
  s->po_brs = words; words += nO;
  s->po_perm = words; words += N;
  /* "arrowhead" patterns: exactly one group receives out-of-group entries (the dense Schur tail of a KKT matrix)
   * and no other group has a triangle.  They get a padded [step][64] column-index table so the device can keep
   * the coupling values and their indices in registers (k_arrow_* kernels). */
  for (k = 0; k < ng; k++) {
    if (fstep_ptr[k + 1] > fstep_ptr[k]) { ngather++; arrow_k = k; arrow_steps = fstep_ptr[k + 1] - fstep_ptr[k]; }
    if (coloff[gstart[k]] >= 0) ntri++;
  }
  s->arrow_ok = (ngather == 1 && ntri <= 1 && (ntri == 0 || coloff[gstart[arrow_k]] >= 0) && nO < 65535) ? 1 : 0;
  s->arrow_group = s->arrow_ok ? arrow_k : -1;
  s->arrow_steps = s->arrow_ok ? arrow_steps : 0;
  /* VIRTUAL ROWS: the coupling rows of the tail group are cut into pieces of at most T consecutive entries, one
   * piece per lane, with T the smallest length for which the pieces fit the 64 lanes.  A lane then keeps T values in
   * registers instead of max-row-length (24 -> 13 on the metric shape: the rows average 15 entries and 14 lanes sat
   * idle), forward results of the pieces of a row meet in its x slot through an LDS atomic add. */
  s->arrow_vsteps = 0; s->arrow_vrows = 0;
  if (s->arrow_ok) {
    int g0 = gstart[arrow_k], g = gstart[arrow_k + 1] - g0, T, nv = 0;
    for (T = 1; T <= arrow_steps; T++) {
      nv = 0;
      for (i = 0; i < g; i++) nv += (rowcnt[g0 + i] + T - 1) / T;
      if (nv <= 64) break;
    }
    if (T > 32) s->arrow_ok = 0;
    else { s->arrow_vsteps = T; s->arrow_vrows = nv; }
  }
  /* The tables are padded to the register bound the kernels are compiled for, so the device reads them without range
   * checks: 12, 18 or 24 steps for the tile kernels (launch_tile_*); the sweep kernels (launch_arrow_*: 8, 12, 14, 16, 18,
   * 24) read a prefix of the same table. */
  {
    const int T = s->arrow_vsteps;
    vpad = T <= 12 ? 12 : T <= 18 ? 18 : T <= 24 ? 24 : ((T + 1) & ~1);
  }
  s->po_avmap = words; words += s->arrow_ok ? ((vpad + 1) / 2) * 64 : 0;   /* dword [ceil(Tpad/2)][64]: slot(t even) | slot(t odd) << 16, 0xffff = none */
  s->po_avcol = words; words += s->arrow_ok ? ((vpad + 1) / 2) * 64 : 0;   /* same packing, column index (0 where there is no entry) */
  s->po_avrow = words; words += s->arrow_ok ? 64 : 0;                                 /* row (permuted index) of the lane's piece */
  /* tail inverse by register tiles (see rldl_symbolic.h): tile size from the set the kernels are compiled for */
  s->tile_ok = 0; s->tile_ta = 0; s->tile_tq = 0; s->tile_lanes = 0; s->nTi = 0;
  s->po_tlane = s->po_tmap = s->po_tislot = s->po_tmask = s->po_pinv = s->po_trc = s->po_spack = 0;
  if (s->arrow_ok && coloff[gstart[arrow_k]] == nOp) {
    static const int tas[4] = {2, 3, 5, 7};
    const int g = gstart[arrow_k + 1] - gstart[arrow_k];
    int a = 0, tq = 0;
    for (i = 0; i < 4; i++) {
      tq = (g + tas[i] - 1) / tas[i];
      if (tq * (tq + 1) / 2 <= 64) { a = tas[i]; break; }
    }
    if (a > 0 && g >= 2 && g <= 64) {
      s->tile_ok = 1; s->tile_ta = a; s->tile_tq = tq; s->tile_lanes = tq * (tq + 1) / 2; s->nTi = g * (g - 1) / 2;
      s->po_tlane = words; words += 64;
      s->po_tmap = words; words += ((a * a + 1) / 2) * 64;
      s->po_tislot = words; words += g * 32;
      words = (words + 3) & ~3;                                /* (16-byte aligned: the kernel reads a record with one scalar load) */
      s->po_tmask = words; words += ((a * a * 2 + 3) & ~3);
      s->po_pinv = words; words += 3 * 64;
      s->po_trc = words; words += a * 64;
      s->tile_vslots = (s->n + 63) / 64; s->tile_slots = s->tile_vslots + (s->m + 63) / 64;
      s->tile_admm_ok = s->tile_slots <= 3 && !s->polish;
      /* fused iterations with the backward coupling product as a SCATTER (no owner copies of the coupling values, fewer registers): what a
       * pattern runs whose owner-gather steps do not fit the compiled splits or the register file; needs only the slot layout */
      s->tile_scatter_ok = s->tile_slots <= 3 && s->tile_vslots == 1 && !s->polish;
      s->po_tpos = words; words += 3 * 64;
      /* owner gather of the backward coupling product: positions of a kind by decreasing column count, steps per slot */
      s->tile_ck[0] = s->tile_ck[1] = s->tile_ck[2] = 0; s->tile_tk = 16; s->tile_sp = 12;
      if (s->tile_admm_ok) {
        int kind, tot = 0;
        for (kind = 0; kind < 2; kind++) {
          int t0 = kind ? s->tile_vslots : 0, t1 = kind ? s->tile_slots : s->tile_vslots, t;
          for (t = t0; t < t1; t++) {                         /* largest count among the positions ranked [64 (t - t0), ...) of this kind */
            int rank = 64 * (t - t0), best = 0, nbig;
            /* the (rank+1)-th largest count: count how many positions of the kind have a count > v */
            for (best = N; best > 0; best--) {
              nbig = 0;
              for (i = 0; i < N; i++) if ((s->perm[i] < s->n) == (kind == 0) && colcnt[i] >= best) nbig++;
              if (nbig > rank) break;
            }
            s->tile_ck[t] = best; tot += best;
          }
        }
        /* compile-time split of the steps: the first constraint slot's columns in [0, sp), the second's in [sp, tk), sp = 3 tk / 4 or 2 tk / 3;
         * head entries must all be constraints (variable slots own tail entries only: ck = 0) */
        (void)tot;
        {
          const int k1 = s->tile_slots > s->tile_vslots ? s->tile_ck[s->tile_vslots] : 0;
          const int k2 = s->tile_slots > s->tile_vslots + 1 ? s->tile_ck[s->tile_vslots + 1] : 0;
          int t, okv = 1;
          for (t = 0; t < s->tile_vslots; t++) if (s->tile_ck[t] > 0) okv = 0;
          s->tile_tk = 0;
          s->tile_sp = 0;
          for (t = 16; t <= 32 && !s->tile_tk; t += 8) {       /* split after 3/4 or after 2/3 of the steps (what the kernels are compiled for) */
            const int spa = (3 * t) / 4, spb = ((2 * t) / 3) & ~1;
            if (k1 <= spa && k2 <= t - spa) { s->tile_tk = t; s->tile_sp = spa; }
            else if (k1 <= spb && k2 <= t - spb) { s->tile_tk = t; s->tile_sp = spb; }
          }
          if (!okv || !s->tile_tk || s->tile_vslots != 1 || s->tile_slots - s->tile_vslots > 2) { s->tile_admm_ok = 0; s->tile_tk = 16; s->tile_sp = 12; }
        }
      }
      s->po_cmap = words; words += (s->tile_tk / 2) * 64;
      s->po_crow = words; words += (s->tile_tk / 2) * 64;
      /* the solve kernel's per-lane table words (k_tile_solve3: pinv 3, avcol H, avmap H, trc a, tlane, avrow; H = vpad / 2) once more as
       * 16-byte records, [chunk][64 lanes][4 words]: one wide load per chunk instead of one 4-byte load per word (28 -> 7 on the metric shape) */
      s->po_spack = 0;
      if (vpad <= 24) {
        words = (words + 3) & ~3;
        s->po_spack = words; words += 4 * 64 * ((3 + 2 * ((vpad + 1) / 2) + a + 2 + 3) / 4);
      }
    }
  }
  blob = (int *)calloc((size_t)words + 4, sizeof(int));
  if (!blob) goto out;

  for (k = 0; k <= ng; k++) {
    blob[s->po_gstart + k] = gstart[k];
    blob[s->po_fsp + k] = fstep_ptr[k];
    blob[s->po_bsp + k] = bstep_ptr[k];
  }
  for (k = 0; k < ng; k++) {
    int g0 = gstart[k];
    blob[s->po_gaptr + k] = 0; blob[s->po_grptr + k] = 0;
    blob[s->po_gToff + k] = coloff[g0] < 0 ? s->nS : coloff[g0];
    blob[s->po_gflag + k] = coloff[g0] >= 0 ? 1 : 0;         /* 1: the group has a packed triangle to sweep */
  }
  for (i = 0; i < N; i++) blob[s->po_perm + i] = s->perm[i];
  /* ---- forward jagged diagonals: rows of each group by out-of-group count (descending) ---- */
  {
    unsigned short *fsig = (unsigned short *)(blob + s->po_fsig), *fcol = (unsigned short *)(blob + s->po_fcol);
    int slot = 0;
    for (k = 0; k < ng; k++) {
      int g0 = gstart[k], g = gstart[k + 1] - g0, nst = fstep_ptr[k + 1] - fstep_ptr[k], t;
      for (i = 0; i < g; i++) { order[i].key = rowcnt[g0 + i]; order[i].idx = g0 + i; }
      qsort(order, (size_t)g, sizeof(kv), cmp_kv_desc);
      for (i = 0; i < g; i++) { fsig[g0 + i] = (unsigned short)order[i].idx; fpos_of_row[order[i].idx] = i; }
      for (t = 0; t < nst; t++) {
        int cnt = 0;
        for (i = 0; i < g; i++) if (order[i].key > t) cnt++;
        fsteps_base[fstep_ptr[k] + t] = slot; fsteps_cnt[fstep_ptr[k] + t] = cnt;
        slot += cnt;
      }
    }
    /* slot of the t-th out-of-group entry (ascending column) of row r = base(step t of its group) + lane(r) */
    for (i = 0; i < N; i++) {
      int t = 0;
      for (q = s->Rp[i]; q < s->Rp[i + 1]; q++) {
        int cc = s->Rj[q];
        if (group_of[cc] == group_of[i]) { slot_of_csr[q] = -1; continue; }
        slot_of_csr[q] = fsteps_base[fstep_ptr[group_of[i]] + t] + fpos_of_row[i];
        fcol[slot_of_csr[q]] = (unsigned short)cc;
        t++;
      }
    }
    for (k = 0; k < nfs; k++) { blob[s->po_fsb + k] = fsteps_base[k]; blob[s->po_fsc + k] = fsteps_cnt[k]; }
    if (s->arrow_ok) {                                       /* virtual-row tables, pieces sorted by length (descending) */
      unsigned *vmap = (unsigned *)(blob + s->po_avmap), *vcol = (unsigned *)(blob + s->po_avcol);
      int *vrow = blob + s->po_avrow;
      const int T = s->arrow_vsteps, f0 = fstep_ptr[arrow_k], g0 = gstart[arrow_k], g = gstart[arrow_k + 1] - g0;
      kv pieces[64];
      int rows_of[64], e0_of[64], np = 0, t, l;
      for (i = 0; i < g; i++) {
        int len = rowcnt[g0 + i], e0;
        for (e0 = 0; e0 < len; e0 += T) {
          pieces[np].key = len - e0 < T ? len - e0 : T;
          pieces[np].idx = np;                               /* stable tie-break */
          rows_of[np] = g0 + i; e0_of[np] = e0;
          np++;
        }
      }
      qsort(pieces, (size_t)np, sizeof(kv), cmp_kv_desc);
      for (l = 0; l < 64; l++) vrow[l] = 0;
      for (t = 0; t < ((vpad + 1) / 2) * 64; t++) { vmap[t] = 0xffffffffu; vcol[t] = 0u; }
      /* Which of its (padded) steps a lane uses for which entry is free, and it decides how the backward pass behaves: there
       * every step is one LDS atomic add per lane into x[column], and an atomic instruction whose lanes collide on an address
       * costs ~3 cycles per colliding lane against ~5 cycles for the whole instruction when all addresses differ (measured,
       * scripts/ubench_ldsatomic.hip).  The entries are therefore placed by a proper edge colouring of the bipartite graph
       * lanes x columns (Koenig: max degree colours suffice): at every step all active lanes hit different columns.  When a
       * column has more entries than there are steps the colouring is greedy (fewest collisions). */
      {
        int *lane_at = (int *)malloc(sizeof(int) * (size_t)64 * (size_t)vpad);       /* [lane][colour] -> column or -1 */
        int *col_at = (int *)malloc(sizeof(int) * (size_t)(N + 1) * (size_t)vpad);    /* [column][colour] -> lane or -1 (exact mode) / count (greedy) */
        int *eslot = (int *)malloc(sizeof(int) * (size_t)64 * (size_t)vpad);          /* [lane][colour] -> factor slot */
        int *pl = (int *)malloc(sizeof(int) * (size_t)3 * (size_t)(64 * vpad + N + 2));
        int *cdeg = (int *)calloc((size_t)N + 1, sizeof(int)), maxc = 0, exact;
        if (!lane_at || !col_at || !eslot || !pl || !cdeg) { free(lane_at); free(col_at); free(eslot); free(pl); free(cdeg); goto out; }
        for (l = 0; l < np; l++) {
          const int src = pieces[l].idx, r = rows_of[src], e0 = e0_of[src], len = pieces[l].key;
          for (t = 0; t < len; t++) { const int c = fcol[fsteps_base[f0 + e0 + t] + fpos_of_row[r]]; if (++cdeg[c] > maxc) maxc = cdeg[c]; }
        }
        exact = maxc <= vpad;
        for (t = 0; t < 64 * vpad; t++) { lane_at[t] = -1; eslot[t] = -1; }
        for (t = 0; t < (N + 1) * vpad; t++) col_at[t] = exact ? -1 : 0;
        for (l = 0; l < np; l++) {
          const int src = pieces[l].idx, r = rows_of[src], e0 = e0_of[src], len = pieces[l].key;
          vrow[l] = r;
          for (t = 0; t < len; t++) {
            const int slot = fsteps_base[f0 + e0 + t] + fpos_of_row[r], c = fcol[slot];
            int a = -1, b = -1, col = -1, k2;
            for (k2 = 0; k2 < vpad; k2++) if (lane_at[l * vpad + k2] < 0) { a = k2; break; }
            if (exact) {
              for (k2 = 0; k2 < vpad; k2++) if (col_at[c * vpad + k2] < 0) { b = k2; break; }
              if (col_at[c * vpad + a] < 0) col = a;
              else {                                         /* free colour a at column c: swap a and b along the alternating path */
                int npth = 0, node = c, side = 0, want = a, k3;
                for (;;) {
                  if (side == 0) { const int l2 = col_at[node * vpad + want]; if (l2 < 0) break; pl[3 * npth] = l2; pl[3 * npth + 1] = node; pl[3 * npth + 2] = want; npth++; node = l2; side = 1; }
                  else { const int c2 = lane_at[node * vpad + want]; if (c2 < 0) break; pl[3 * npth] = node; pl[3 * npth + 1] = c2; pl[3 * npth + 2] = want; npth++; node = c2; side = 0; }
                  want = want == a ? b : a;
                }
                for (k3 = 0; k3 < npth; k3++) {              /* take the path's edges out (remember their slots) ... */
                  const int pl_l = pl[3 * k3], pl_c = pl[3 * k3 + 1], pc = pl[3 * k3 + 2];
                  pl[3 * k3 + 2] = pc | (eslot[pl_l * vpad + pc] << 8);
                  lane_at[pl_l * vpad + pc] = -1; col_at[pl_c * vpad + pc] = -1; eslot[pl_l * vpad + pc] = -1;
                }
                for (k3 = 0; k3 < npth; k3++) {              /* ... and put them back with the two colours exchanged */
                  const int pl_l = pl[3 * k3], pl_c = pl[3 * k3 + 1], pc = pl[3 * k3 + 2] & 0xff, sl = pl[3 * k3 + 2] >> 8;
                  const int nc = pc == a ? b : a;
                  lane_at[pl_l * vpad + nc] = pl_c; col_at[pl_c * vpad + nc] = pl_l; eslot[pl_l * vpad + nc] = sl;
                }
                col = a;
              }
              col_at[c * vpad + col] = l;
            } else {                                         /* greedy: the lane's free colour with the fewest entries of this column */
              int best = 1 << 30;
              for (k2 = 0; k2 < vpad; k2++)
                if (lane_at[l * vpad + k2] < 0 && col_at[c * vpad + k2] < best) { best = col_at[c * vpad + k2]; col = k2; }
              col_at[c * vpad + col]++;
            }
            lane_at[l * vpad + col] = c; eslot[l * vpad + col] = slot;
          }
        }
        for (l = 0; l < np; l++)
          for (t = 0; t < vpad; t++) {
            const int slot = eslot[l * vpad + t], sh = 16 * (t & 1);
            if (slot < 0) continue;
            vmap[(t >> 1) * 64 + l] = (vmap[(t >> 1) * 64 + l] & ~(0xffffu << sh)) | ((unsigned)slot << sh);
            vcol[(t >> 1) * 64 + l] |= (unsigned)fcol[slot] << sh;
          }
        free(lane_at); free(col_at); free(eslot); free(pl); free(cdeg);
      }
    }
  }
  /* storage map: CSC position -> slot */
  for (i = 0; i < N; i++)
    for (q = s->Rp[i]; q < s->Rp[i + 1]; q++) {
      int cc = s->Rj[q], pcsc = s->Rpos[q];
      { const int il = i - gstart[group_of[i]], jl = cc - gstart[group_of[i]];
        s->LtoS[pcsc] = slot_of_csr[q] >= 0 ? slot_of_csr[q] : coloff[i] + il * (il - 1) / 2 + jl; }
    }
  if (s->tile_ok) {                                           /* tile tables of the tail inverse */
    const int a = s->tile_ta, tq = s->tile_tq, g = gstart[arrow_k + 1] - gstart[arrow_k];
    unsigned *tl = (unsigned *)(blob + s->po_tlane), *tm = (unsigned *)(blob + s->po_tmap);
    unsigned short *ts = (unsigned short *)(blob + s->po_tislot);
    int tI[64], tJ[64], nl = 0, l, kk, slot = 0;
    for (i = 1; i < tq; i++)                                  /* full tiles first (I > J), then the diagonal tiles */
      for (k = 0; k < i; k++) { tI[nl] = i; tJ[nl] = k; nl++; }
    for (i = 0; i < tq; i++) { tI[nl] = i; tJ[nl] = i; nl++; }
    for (l = 0; l < 64; l++) tl[l] = l < nl ? (unsigned)tI[l] | ((unsigned)tJ[l] << 8) : 0xffffffffu;
    for (l = 0; l < ((a * a + 1) / 2) * 64; l++) tm[l] = 0xffffffffu;
    for (l = 0; l < g * 64; l++) ts[l] = 0xffffu;
    for (kk = 0; kk < a * a; kk++) {
      const int sr = kk / a, u = kk % a, sh = 16 * (kk & 1);
      for (l = 0; l < nl; l++) {
        const int row = a * tI[l] + (sr + tJ[l]) % a, col = a * tJ[l] + (u + tI[l]) % a;
        if (row >= g || col >= row) continue;                 /* structural zero: not stored */
        tm[(kk >> 1) * 64 + l] = (tm[(kk >> 1) * 64 + l] & ~(0xffffu << sh)) | ((unsigned)slot << sh);
        ts[row * 64 + col] = (unsigned short)slot;
        slot++;
      }
    }
    if (slot != s->nTi) { rc = -2; goto out; }                /* cannot happen: every strictly lower entry lies in exactly one tile */
    {                                                         /* the same map without a per-lane table: the 64-bit lane mask of every register k (lo, hi words);
                                                               * slot(k, lane) = (entries of the registers before k) + (set mask bits below the lane) -- Ti is in (k, lane) order */
      unsigned *tk = (unsigned *)(blob + s->po_tmask);
      int first = 0;
      for (kk = 0; kk < a * a; kk++) {
        unsigned long long msk = 0;
        int cnt = 0;
        for (l = 0; l < nl; l++) {
          const unsigned w = tm[(kk >> 1) * 64 + l], sl = (kk & 1) ? w >> 16 : w & 0xffffu;
          if (sl == 0xffffu) continue;
          if ((int)sl != first + cnt) { rc = -2; goto out; }  /* cannot happen: slots were handed out in (k, lane) order */
          msk |= 1ull << l; cnt++;
        }
        tk[2 * kk] = (unsigned)msk; tk[2 * kk + 1] = (unsigned)(msk >> 32);
        first += cnt;
      }
      {                                                       /* x slot (in doubles from the wave's x[0]) of original index i = its permuted position; indices
                                                               * past N point at the lane's dummy word (xdw + lane, xdw as in launch geometry: tile_per_wave) */
        const int need = gstart[arrow_k] + a * tq, xdw = ((need > N ? need : N) + 1) & ~1;
        for (i = 0; i < 3 * 64; i++) blob[s->po_pinv + i] = xdw + (i & 63);
        if (N <= 3 * 64) for (i = 0; i < N; i++) blob[s->po_pinv + s->perm[i]] = i;
        if (s->arrow_ok) for (l = s->arrow_vrows; l < 64; l++) blob[s->po_avrow + l] = xdw + l;   /* lanes without a virtual row: their dummy word */
      }
      {                                                       /* tile addresses without arithmetic: word s of the lane = (rotated row s | rotated column s << 16), local to the tail */
        unsigned *rc = (unsigned *)(blob + s->po_trc);
        int sidx;
        for (sidx = 0; sidx < a; sidx++)
          for (l = 0; l < 64; l++)
            rc[sidx * 64 + l] = l < nl ? (unsigned)(a * tI[l] + (sidx + tJ[l]) % a) | ((unsigned)(a * tJ[l] + (sidx + tI[l]) % a) << 16) : 0u;
      }
    }
    {                                                         /* ADMM slots: variables first, then constraints; inside a kind by decreasing
                                                               * number of coupling entries in the position's column (ties: ascending position) */
      int *tp = blob + s->po_tpos;
      unsigned *cm = (unsigned *)(blob + s->po_cmap), *cr = (unsigned *)(blob + s->po_crow);
      const int g0 = gstart[arrow_k];
      for (l = 0; l < 3 * 64; l++) tp[l] = -1;
      for (l = 0; l < (s->tile_tk / 2) * 64; l++) { cm[l] = 0xffffffffu; cr[l] = 0u; }
      if (s->tile_admm_ok || s->tile_scatter_ok) {
        kv *ord = (kv *)malloc(sizeof(kv) * (size_t)(N + 1));
        int kind, t;
        if (!ord) goto out;
        for (kind = 0; kind < 2; kind++) {
          int cntk = 0, base = kind ? 64 * s->tile_vslots : 0;
          for (i = 0; i < N; i++)
            if ((s->perm[i] < s->n) == (kind == 0)) { ord[cntk].key = colcnt[i]; ord[cntk].idx = i; cntk++; }
          qsort(ord, (size_t)cntk, sizeof(kv), cmp_kv_desc);
          for (i = 0; i < cntk; i++) tp[base + i] = ord[i].idx;
        }
        free(ord);
        if (s->tile_admm_ok)
        /* Entries of the owner's column on the steps [e0, e0 + ck[t]) of its slot.  WHICH step an entry takes is free, and it
         * decides the LDS bank conflicts of the gather: a ds_read_b64 serves 32 lanes per pass and two lanes of a pass collide
         * when they read different words of one bank pair ((word index) mod 32).  Greedy placement, lane by lane: an entry goes
         * to a free step of the lane where its bank pair is still unused in the lane's half-wave (or holds the same word); the
         * steps a lane leaves empty read the lane's own dummy word, which occupies its bank pair too. */
        {
          const int tk = s->tile_tk, need = g0 + s->tile_ta * s->tile_tq, xdw = ((need > N ? need : N) + 1) & ~1;
          int *used = (int *)malloc(sizeof(int) * (size_t)tk * 2 * 32);             /* [step][half][bank pair] -> word index or -1 */
          int *lane_step = (int *)malloc(sizeof(int) * (size_t)tk);
          if (!used || !lane_step) { free(used); free(lane_step); goto out; }
          for (l = 0; l < tk * 64; l++) used[l] = -1;
          for (t = 0; t < s->tile_slots; t++) {
            const int e0 = t <= s->tile_vslots ? 0 : s->tile_sp, K = s->tile_ck[t];
            for (l = 0; l < 64; l++) {
              const int cpos = tp[t * 64 + l], half = l >> 5, dmy = xdw + l;
              int kk2;
              if (cpos < 0) continue;
              for (kk2 = 0; kk2 < K; kk2++) lane_step[kk2] = -1;
              for (p = s->Lp[cpos]; p < s->Lp[cpos + 1]; p++) {
                const int r = s->Li[p], word = r, bank = r & 31;
                int best = -1, bestc = 3;
                if (group_of[r] == group_of[cpos]) continue;  /* (in-group entry of a tail column) */
                if (r < g0 || r >= gstart[arrow_k + 1]) { free(used); free(lane_step); rc = -2; goto out; }   /* cannot happen on an arrowhead plan */
                for (kk2 = 0; kk2 < K; kk2++) {
                  int u, c;
                  if (lane_step[kk2] >= 0) continue;
                  u = used[((e0 + kk2) * 2 + half) * 32 + bank];
                  c = (u < 0) ? 0 : (u == word ? 0 : 1);
                  if (c < bestc) { bestc = c; best = kk2; }
                }
                if (best < 0) { free(used); free(lane_step); rc = -2; goto out; }   /* more entries than steps: cannot happen (ck = max count) */
                lane_step[best] = p;
                if (used[((e0 + best) * 2 + half) * 32 + bank] < 0) used[((e0 + best) * 2 + half) * 32 + bank] = word;
              }
              for (kk2 = 0; kk2 < K; kk2++) {
                const int kstep = e0 + kk2, sh = 16 * (kstep & 1), pp = lane_step[kk2];
                if (pp < 0) { if (used[(kstep * 2 + half) * 32 + (dmy & 31)] < 0) used[(kstep * 2 + half) * 32 + (dmy & 31)] = dmy; continue; }
                cm[(kstep >> 1) * 64 + l] = (cm[(kstep >> 1) * 64 + l] & ~(0xffffu << sh)) | ((unsigned)s->LtoS[pp] << sh);
                cr[(kstep >> 1) * 64 + l] |= (unsigned)(s->Li[pp] - g0) << sh;
              }
            }
          }
          free(used); free(lane_step);
        }
      }
    }
  }

  /* ---- backward jagged diagonals: columns of each group by out-of-group count (descending) ---- */
  {
    unsigned short *bsig = (unsigned short *)(blob + s->po_bsig);
    int e = 0;
    for (k = 0; k < ng; k++) {
      int g0 = gstart[k], g = gstart[k + 1] - g0, nst = bstep_ptr[k + 1] - bstep_ptr[k], t;
      for (i = 0; i < g; i++) { order[i].key = colcnt[g0 + i]; order[i].idx = g0 + i; }
      qsort(order, (size_t)g, sizeof(kv), cmp_kv_desc);
      for (i = 0; i < g; i++) { bsig[g0 + i] = (unsigned short)order[i].idx; bpos_of_col[order[i].idx] = i; }
      for (t = 0; t < nst; t++) {
        int cnt = 0;
        for (i = 0; i < g; i++) if (order[i].key > t) cnt++;
        bsteps_base[bstep_ptr[k] + t] = e; bsteps_cnt[bstep_ptr[k] + t] = cnt;
        e += cnt;
      }
    }
    for (c = 0; c < N; c++) {
      int t = 0;
      for (p = s->Lp[c]; p < s->Lp[c + 1]; p++) {
        int r = s->Li[p];
        if (group_of[r] == group_of[c]) continue;
        blob[s->po_brs + bsteps_base[bstep_ptr[group_of[c]] + t] + bpos_of_col[c]] =
            (int)((unsigned)r | ((unsigned)s->LtoS[p] << 16));
        t++;
      }
    }
    for (k = 0; k < nbs; k++) { blob[s->po_bsb + k] = bsteps_base[k]; blob[s->po_bsc + k] = bsteps_cnt[k]; }
  }

  if (s->tile_ok && s->po_spack > 0) {                       /* every source table is filled: pack (order = the kernel's word index) */
    const int H = (vpad + 1) / 2, a = s->tile_ta, nw = 3 + 2 * H + a + 2;
    int q, l;
    for (q = 0; q < nw; q++) {
      const int src = q < 3 ? s->po_pinv + 64 * q : q < 3 + H ? s->po_avcol + 64 * (q - 3) : q < 3 + 2 * H ? s->po_avmap + 64 * (q - 3 - H)
                    : q < 3 + 2 * H + a ? s->po_trc + 64 * (q - 3 - 2 * H) : q == 3 + 2 * H + a ? s->po_tlane : s->po_avrow;
      for (l = 0; l < 64; l++) blob[s->po_spack + 256 * (q / 4) + 4 * l + (q % 4)] = blob[src + l];
    }
  }
  s->plan = blob; blob = 0;
  s->plan_words = words;
  s->plan_ok = 1;
  if (getenv("RLDL_VERBOSE")) {
    int full = 0;
    for (k = 0; k < ng; k++) full += s->plan[s->po_gflag + k];
    fprintf(stderr, "[rldl] plan: N=%d nnzL=%d -> nS=%d (nO=%d, triangles=%d, padding=%d), %d groups (%d closed-form):", N,
            s->nnzL, s->nS, nO, tri, s->nS - s->nnzL, ng, full);
    for (k = 0; k < ng && k < 16; k++) fprintf(stderr, " [%d,%d)", gstart[k], gstart[k + 1]);
    fprintf(stderr, "%s, sweep steps %d fwd / %d bwd, gather steps %d fwd / %d bwd, %d plan words\n", ng > 16 ? " ..." : "", na,
            nr, nfs, nbs, words);
  }
  rc = 0;

out:
  if (rc == 0 && !s->plan_ok) {                               /* identity layout for the generic kernels */
    for (p = 0; p < s->nnzL; p++) s->LtoS[p] = p;
    s->nS = s->nnzL; s->nO = s->nnzL; s->nOp = s->nnzL; s->ngroups = 0; s->arrow_ok = 0; s->tile_ok = 0;
  }
  free(group_of); free(gstart); free(col_active); free(row_active); free(coloff); free(rowcnt); free(colcnt);
  free(fpos_of_row); free(bpos_of_col); free(order); free(slot_of_csr);
  free(fsteps_base); free(fsteps_cnt); free(bsteps_base); free(bsteps_cnt); free(fstep_ptr); free(bstep_ptr); free(blob);
  return rc;
}
