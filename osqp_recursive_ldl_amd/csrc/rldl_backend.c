/*
 * rldl_backend.c -- host side (plain C) of the batched linear-system backend and the legacy
 * single-instance plugin object.  Talks to the GPU only through the HIP runtime C API (memory,
 * streams, events) and the extern "C" launchers of rldl_device.h.
 *
 * Mirrors lin_sys/direct/qdldl/qdldl_interface.c: init :170-316, solve :559-585,
 * update_matrices :590-602, update_rho_vec :605-619, free :17-43.
 * There is NO CPU fallback: without a HIP device every entry point fails with RLDL_NO_DEVICE_ERROR.
 */
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/osqp_rldl_hip.h"
#include "rldl_device.h"
#include "rldl_internal.h"
#include "rldl_symbolic.h"

#define HIP_OK(call) ((call) == hipSuccess)

const char *rldl_version(void) { return "osqp-rldl-hip 0.1 (gfx950)"; }

/* ------------------------------------------------------------------ device pool ---- */
static void *dev_upload(const void *src, size_t bytes, int *ok) {
  void *d = 0;
  if (!*ok) return 0;
  if (!HIP_OK(hipMalloc(&d, bytes ? bytes : 8))) { *ok = 0; return 0; }
  if (bytes && !HIP_OK(hipMemcpy(d, src, bytes, hipMemcpyHostToDevice))) { *ok = 0; }
  return d;
}
static void *dev_alloc(size_t bytes, int *ok) {
  void *d = 0;
  if (!*ok) return 0;
  if (!HIP_OK(hipMalloc(&d, bytes ? bytes : 8))) { *ok = 0; return 0; }
  return d;
}

int rldl_device_available(void) {
  int count = 0;
  if (!HIP_OK(hipGetDeviceCount(&count))) return 0;
  return count > 0;
}

/* Deal nnz entries (rows[], cols[]) into rounds of 64 so that no round holds many entries of one target: targets are the rows
 * and the columns, in one common index space when `same` is set (symmetric use of P: both indices address the same vector).
 * Greedy: entries of the busiest targets first, each into the round (with a free lane) where its targets are rarest.
 * Out: rc[rounds * 64], pos[rounds * 64] (0xffffffff = padding); returns rounds, or -1. */
static int deal_entries(int nnz, const int *rows, const int *cols, int nr, int nc, int same, unsigned **rc_out, unsigned **pos_out) {
  const int R = (nnz + 63) / 64 > 0 ? (nnz + 63) / 64 : 1;
  int *cr = (int *)calloc((size_t)(nr + 1) * (size_t)R, sizeof(int)), *cc = same ? 0 : (int *)calloc((size_t)(nc + 1) * (size_t)R, sizeof(int));
  int *fill = (int *)calloc((size_t)R, sizeof(int)), *deg = (int *)calloc((size_t)(nr + nc + 2), sizeof(int)), *ord = (int *)malloc(sizeof(int) * (size_t)(nnz + 1));
  unsigned *rc = (unsigned *)malloc(sizeof(unsigned) * (size_t)R * 64), *pos = (unsigned *)malloc(sizeof(unsigned) * (size_t)R * 64);
  int i, k, rv = -1;
  if (!cr || (!same && !cc) || !fill || !deg || !ord || !rc || !pos) goto done;
  if (same) cc = cr;
  for (i = 0; i < nnz; i++) { deg[rows[i]]++; deg[(same ? 0 : nr) + cols[i]]++; ord[i] = i; }
  /* insertion sort by decreasing max degree (nnz is a few thousand at most per pattern; stable) */
  {
    int *key = (int *)malloc(sizeof(int) * (size_t)(nnz + 1)), maxk = 0, *start;
    if (!key) goto done;
    for (i = 0; i < nnz; i++) { const int a = deg[rows[i]], b = deg[(same ? 0 : nr) + cols[i]]; key[i] = a > b ? a : b; if (key[i] > maxk) maxk = key[i]; }
    start = (int *)calloc((size_t)maxk + 2, sizeof(int));
    if (!start) { free(key); goto done; }
    for (i = 0; i < nnz; i++) start[maxk - key[i] + 1]++;                 /* counting sort, descending */
    for (k = 0; k < maxk + 1; k++) start[k + 1] += start[k];
    for (i = 0; i < nnz; i++) ord[start[maxk - key[i]]++] = i;
    free(key); free(start);
  }
  for (i = 0; i < R * 64; i++) { rc[i] = 0u; pos[i] = 0xffffffffu; }
  for (k = 0; k < nnz; k++) {
    const int e = ord[k], r = rows[e], c = cols[e];
    int best = -1, bm = 1 << 30, bs = 1 << 30, q;
    for (q = 0; q < R; q++) {
      int a, b, m2;
      if (fill[q] >= 64) continue;
      a = cr[r * R + q]; b = cc[c * R + q];
      m2 = a > b ? a : b;
      if (m2 < bm || (m2 == bm && a + b < bs)) { bm = m2; bs = a + b; best = q; }
    }
    rc[best * 64 + fill[best]] = (unsigned)r | ((unsigned)c << 16);
    pos[best * 64 + fill[best]] = (unsigned)e;
    fill[best]++; cr[r * R + best]++; if (!same || c != r) cc[c * R + best]++;
  }
  *rc_out = rc; *pos_out = pos; rc = 0; pos = 0; rv = R;
done:
  free(cr); if (!same) free(cc); free(fill); free(deg); free(ord); free(rc); free(pos);
  return rv;
}

static int upload_symbolic(rldl_batch *h) {
  const rldl_symbolic *s = h->sym;
  rldl_dev_sym *D = &h->dsym;
  int ok = 1;
  memset(D, 0, sizeof(*D));
  D->n = s->n; D->m = s->m; D->N = s->N; D->nnzP = s->nnzP; D->nnzA = s->nnzA; D->nnzK = s->nnzK;
  D->nnzL = s->nnzL; D->nsig = s->nsig; D->polish = s->polish;
#define UP(field, count, type) D->field = (const type *)dev_upload(s->field, sizeof(type) * (size_t)(count), &ok)
  UP(perm, s->N, int); UP(PtoK, s->nnzP, int); UP(AtoK, s->nnzA, int); UP(rhotoK, s->m, int); UP(sigK, s->nsig, int);
  UP(Pisdiag, s->nnzP, unsigned char);
  UP(Lp, s->N + 1, int); UP(Li, s->nnzL, int); UP(Rp, s->N + 1, int); UP(Rj, s->nnzL, int); UP(Rpos, s->nnzL, int);
  UP(KtoW, s->nnzK, int); UP(Udst, s->npairs, int); UP(Uab, s->npairs, unsigned int); UP(Up, s->N + 1, long long);
  UP(Pp, s->n + 1, int); UP(Pi, s->nnzP, int); UP(Prp, s->n + 1, int); UP(Prj, s->nnzP, int); UP(Prpos, s->nnzP, int);
  UP(Ap, s->n + 1, int); UP(Ai, s->nnzA, int); UP(Arp, s->m + 1, int); UP(Arj, s->nnzA, int); UP(Arpos, s->nnzA, int);
  UP(LtoS, s->nnzL, int);
  if (s->n < 65536 && s->m < 65536) {                            /* (row | col << 16) of every P / A entry, storage order */
    unsigned *t = (unsigned *)malloc(sizeof(unsigned) * (size_t)(s->nnzP + s->nnzA + 2));
    if (t) {
      unsigned *pfl = t, *afl = pfl + s->nnzP;
      int j, q;
      for (j = 0; j < s->n; j++) {
        for (q = s->Pp[j]; q < s->Pp[j + 1]; q++) pfl[q] = (unsigned)s->Pi[q] | ((unsigned)j << 16);
        for (q = s->Ap[j]; q < s->Ap[j + 1]; q++) afl[q] = (unsigned)s->Ai[q] | ((unsigned)j << 16);
      }
      D->Pfl = (const unsigned *)dev_upload(pfl, sizeof(unsigned) * (size_t)s->nnzP, &ok);
      D->Afl = (const unsigned *)dev_upload(afl, sizeof(unsigned) * (size_t)s->nnzA, &ok);
      {
        int *rw = (int *)malloc(sizeof(int) * (size_t)(s->nnzP + s->nnzA + 2)), *cl = (int *)malloc(sizeof(int) * (size_t)(s->nnzP + s->nnzA + 2));
        unsigned *brc = 0, *bps = 0;
        int fine = rw && cl;
        if (fine) {
          for (q = 0; q < s->nnzP; q++) { rw[q] = (int)(pfl[q] & 0xffffu); cl[q] = (int)(pfl[q] >> 16); }
          D->Pbr = deal_entries(s->nnzP, rw, cl, s->n, s->n, 1, &brc, &bps);
          if (D->Pbr > 0) {
            D->Pbl = (const unsigned *)dev_upload(brc, sizeof(unsigned) * (size_t)D->Pbr * 64, &ok);
            D->Pbp = (const unsigned *)dev_upload(bps, sizeof(unsigned) * (size_t)D->Pbr * 64, &ok);
          } else fine = 0;
          free(brc); free(bps); brc = bps = 0;
        }
        if (fine) {
          for (q = 0; q < s->nnzA; q++) { rw[q] = (int)(afl[q] & 0xffffu); cl[q] = (int)(afl[q] >> 16); }
          D->Abr = deal_entries(s->nnzA, rw, cl, s->m, s->n, 0, &brc, &bps);
          if (D->Abr > 0) {
            D->Abl = (const unsigned *)dev_upload(brc, sizeof(unsigned) * (size_t)D->Abr * 64, &ok);
            D->Abp = (const unsigned *)dev_upload(bps, sizeof(unsigned) * (size_t)D->Abr * 64, &ok);
          } else fine = 0;
          free(brc); free(bps);
        }
        free(rw); free(cl);
        D->flat_ok = ok && fine && s->nnzP > 0;
      }
      free(t);
    }
  }
  D->plan = (const int *)dev_upload(s->plan, sizeof(int) * (size_t)(s->plan_ok ? ((s->plan_words + 3) & ~3) : 0), &ok); /* blob is calloc'ed with 4 words of slack */
  D->ldF = (s->nS + s->N + 2) & ~1;             /* even, and at least one spare slot that stays 0.0 (ldF-1) */
  D->nS = s->nS; D->nO = s->nO; D->ngroups = s->ngroups; D->plan_ok = s->plan_ok; D->plan_words = s->plan_ok ? s->plan_words : 0;
  D->po_gstart = s->po_gstart; D->po_gflag = s->po_gflag; D->po_gaptr = s->po_gaptr; D->po_grptr = s->po_grptr;
  D->po_gToff = s->po_gToff; D->po_fsp = s->po_fsp; D->po_bsp = s->po_bsp; D->po_acol = s->po_acol; D->po_aoff = s->po_aoff;
  D->po_arow = s->po_arow; D->po_coloff = s->po_coloff; D->po_fsb = s->po_fsb; D->po_fsc = s->po_fsc; D->po_bsb = s->po_bsb;
  D->po_bsc = s->po_bsc; D->po_fsig = s->po_fsig; D->po_bsig = s->po_bsig; D->po_fcol = s->po_fcol; D->po_brs = s->po_brs;
  D->po_perm = s->po_perm; D->po_avmap = s->po_avmap; D->po_avcol = s->po_avcol; D->po_avrow = s->po_avrow; D->nOp = s->nOp;
  D->arrow_ok = s->plan_ok ? s->arrow_ok : 0; D->arrow_group = s->arrow_group; D->arrow_steps = s->arrow_steps;
  D->arrow_vsteps = s->arrow_vsteps; D->arrow_vrows = s->arrow_vrows;
  D->tile_ok = D->arrow_ok ? s->tile_ok : 0; D->tile_ta = s->tile_ta; D->tile_tq = s->tile_tq; D->tile_lanes = s->tile_lanes;
  D->nTi = s->nTi; D->ldTi = (s->nTi + 1) & ~1;
  D->po_tlane = s->po_tlane; D->po_tmap = s->po_tmap; D->po_tislot = s->po_tislot; D->po_tmask = s->po_tmask; D->po_pinv = s->po_pinv; D->po_trc = s->po_trc; D->po_spack = s->po_spack;
  D->tile_admm_ok = D->tile_ok ? s->tile_admm_ok : 0; D->tile_scatter_ok = D->tile_ok ? s->tile_scatter_ok : 0; D->tile_vslots = s->tile_vslots; D->tile_slots = s->tile_slots; D->po_tpos = s->po_tpos;
  D->tile_ck[0] = s->tile_ck[0]; D->tile_ck[1] = s->tile_ck[1]; D->tile_ck[2] = s->tile_ck[2]; D->tile_tk = s->tile_tk; D->tile_sp = s->tile_sp;
  D->po_cmap = s->po_cmap; D->po_crow = s->po_crow;
  if (D->arrow_ok) {
    int t, l;
    D->arrow_g0 = s->plan[s->po_gstart + s->arrow_group];
    D->arrow_g = s->plan[s->po_gstart + s->arrow_group + 1] - D->arrow_g0;
    D->arrow_tb = s->plan[s->po_gflag + s->arrow_group] ? s->plan[s->po_gToff + s->arrow_group] - s->nOp : -1;
    {                                                         /* k_arrow_factor: independent head columns, tail group at the very end */
      int c, p, okf = D->arrow_tb >= 0 && D->arrow_g0 + D->arrow_g == s->N && D->arrow_g <= 64;
      for (c = 0; c < D->arrow_g0 && okf; c++)
        for (p = s->Lp[c]; p < s->Lp[c + 1]; p++)
          if (s->Li[p] < D->arrow_g0) { okf = 0; break; }
      D->arrow_dense = 0;
      if (okf) {                                              /* workspace position of L(g0 + r, g0 + c), or -1: table [g][64] */
        int *tp = (int *)malloc(sizeof(int) * (size_t)D->arrow_g * 64);
        if (tp) {
          for (p = 0; p < D->arrow_g * 64; p++) tp[p] = -1;
          for (c = 0; c < D->arrow_g; c++)
            for (p = s->Lp[D->arrow_g0 + c]; p < s->Lp[D->arrow_g0 + c + 1]; p++) tp[c * 64 + (s->Li[p] - D->arrow_g0)] = p;
          D->arrow_tpos = (const int *)dev_upload(tp, sizeof(int) * (size_t)D->arrow_g * 64, &ok);
          free(tp);
          D->arrow_dense = D->arrow_tpos != 0;
        }
        /* flat list of the head columns' pair updates with absolute workspace positions:
         * pab = posA | posB << 16, pdc = dst | column << 16 (all < 65536) */
        if (D->arrow_dense && s->nnzL + s->N < 65536 && s->Up[D->arrow_g0] < (1ll << 30)) {
          const long long P = s->Up[D->arrow_g0];
          unsigned *pab = (unsigned *)malloc(sizeof(unsigned) * (size_t)(P + 1)), *pdc = (unsigned *)malloc(sizeof(unsigned) * (size_t)(P + 1));
          long long t;
          if (pab && pdc) {
            for (c = 0; c < D->arrow_g0; c++)
              for (t = s->Up[c]; t < s->Up[c + 1]; t++) {
                pab[t] = (unsigned)(s->Lp[c] + (int)(s->Uab[t] & 0xffffu)) | ((unsigned)(s->Lp[c] + (int)(s->Uab[t] >> 16)) << 16);
                pdc[t] = (unsigned)s->Udst[t] | ((unsigned)c << 16);
              }
            D->arrow_pab = (const unsigned *)dev_upload(pab, sizeof(unsigned) * (size_t)P, &ok);
            D->arrow_pdc = (const unsigned *)dev_upload(pdc, sizeof(unsigned) * (size_t)P, &ok);
            D->arrow_npairs = (int)P;
          }
          free(pab); free(pdc);
          if (!D->arrow_pab || !D->arrow_pdc) D->arrow_dense = 0;
          if (D->arrow_dense) {                                /* write-out table in slot order (coalesced stores) */
            unsigned *ot = (unsigned *)malloc(sizeof(unsigned) * (size_t)(s->nS + 1));
            if (ot) {
              for (p = 0; p < s->nS; p++) ot[p] = 0xffffffffu;
              for (c = 0; c < s->N; c++)
                for (p = s->Lp[c]; p < s->Lp[c + 1]; p++)
                  ot[s->LtoS[p]] = (unsigned)p | ((unsigned)(c < D->arrow_g0 ? c : 0xffff) << 16);
              D->arrow_out = (const unsigned *)dev_upload(ot, sizeof(unsigned) * (size_t)s->nS, &ok);
              free(ot);
            }
            if (!D->arrow_out) D->arrow_dense = 0;
          }
        } else D->arrow_dense = 0;
      }
    }
    for (t = 0; t < 32; t++) {                                /* lanes whose virtual row has an entry at step t */
      const unsigned *vm = (const unsigned *)(s->plan + s->po_avmap);
      D->arrow_cnt[t] = 0;
      if (t < s->arrow_vsteps)
        for (l = 0; l < 64; l++)
          if (((vm[(t >> 1) * 64 + l] >> (16 * (t & 1))) & 0xffffu) != 0xffffu) D->arrow_cnt[t]++;
    }
  }
#undef UP
  return ok ? 0 : -1;
}

static void free_dev_symbolic(rldl_dev_sym *D) {
#define FR(f) if (D->f) (void)hipFree((void *)D->f)
  FR(perm); FR(PtoK); FR(AtoK); FR(rhotoK); FR(sigK); FR(Pisdiag); FR(Lp); FR(Li); FR(Rp); FR(Rj); FR(Rpos);
  FR(KtoW); FR(Udst); FR(Uab); FR(Up); FR(Pp); FR(Pi); FR(Prp); FR(Prj); FR(Prpos); FR(Ap); FR(Ai); FR(Arp);
  FR(Arj); FR(Arpos); FR(LtoS); FR(Pfl); FR(Afl); FR(Pbl); FR(Pbp); FR(Abl); FR(Abp); FR(plan); FR(arrow_tpos); FR(arrow_pab); FR(arrow_pdc); FR(arrow_out);
#undef FR
  memset(D, 0, sizeof(*D));
}

void rldl_batch_free(rldl_batch *h) {
  if (!h) return;
  if (h->stream_owned && h->stream) (void)hipStreamSynchronize((hipStream_t)h->stream);
  rldl_stage_maps_free(h);
  free_dev_symbolic(&h->dsym);
  if (h->num.Kx) (void)hipFree(h->num.Kx);
  if (h->num.F) (void)hipFree(h->num.F);
  if (h->num.D) (void)hipFree(h->num.D);
  if (h->num.Ti) (void)hipFree(h->num.Ti);
  if (h->num.rho_inv) (void)hipFree(h->num.rho_inv);
  if (h->num.status) (void)hipFree(h->num.status);
  if (h->num.fail) (void)hipFree(h->num.fail);
  if (h->ev0) (void)hipEventDestroy((hipEvent_t)h->ev0);
  if (h->ev1) (void)hipEventDestroy((hipEvent_t)h->ev1);
  rldl_symbolic_free(h->sym);
  if (h->status_host) (void)hipHostFree(h->status_host);
  free(h);
}

/* numeric factor of every (masked) instance + host-side verdict, qdldl_interface.c:80-92 */
static c_int collect_status(rldl_batch *h) {
  c_int b, bad = 0;
  int sticky = 0;
  if (!HIP_OK(hipMemcpyAsync(h->status_host, h->num.status, sizeof(int) * (size_t)h->batch, hipMemcpyDeviceToHost,
                             (hipStream_t)h->stream)))
    return RLDL_LINSYS_SOLVER_INIT_ERROR;
  /* the sticky flag covers every factorisation enqueued since the last verdict, not only the last one per instance */
  if (!HIP_OK(hipMemcpyAsync(&sticky, h->num.fail, sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)h->stream))) return RLDL_LINSYS_SOLVER_INIT_ERROR;
  if (!HIP_OK(hipMemsetAsync(h->num.fail, 0, sizeof(int), (hipStream_t)h->stream))) return RLDL_LINSYS_SOLVER_INIT_ERROR;
  if (!HIP_OK(hipStreamSynchronize((hipStream_t)h->stream))) return RLDL_LINSYS_SOLVER_INIT_ERROR;
  for (b = 0; b < h->batch; b++)
    if (h->status_host[b] < 0 || h->status_host[b] < h->sym->n) bad = 1;
  return (bad || sticky) ? RLDL_NONCVX_ERROR : 0;
}

static c_int factor_and_check(rldl_batch *h, const int *d_mask) {
  if (rldl_launch_factor(&h->dsym, &h->num, d_mask, h->stream)) return RLDL_LINSYS_SOLVER_INIT_ERROR;
  return collect_status(h);
}

c_int rldl_batch_check_status(rldl_batch *h) { return collect_status(h) ? 1 : 0; }

static c_int batch_create(rldl_batch **hp, c_int batch, const csc *P, const csc *A, c_float sigma, c_int polish,
                          const c_int *perm, void *stream) {
  rldl_batch *h;
  int rc, ok = 1;
  *hp = 0;
  if (!rldl_device_available()) return RLDL_NO_DEVICE_ERROR;
  if (!P || !A || batch <= 0 || P->n != A->n || P->m != P->n) return RLDL_LINSYS_SOLVER_INIT_ERROR;
  h = (rldl_batch *)calloc(1, sizeof(rldl_batch));
  if (!h) return RLDL_MEM_ALLOC_ERROR;
  h->batch = batch; h->stream = stream;
  rc = rldl_symbolic_create(&h->sym, P->n, A->m, P->p, P->i, A->p, A->i, polish != 0, perm);
  if (rc) { free(h); return rc == -2 ? RLDL_MEM_ALLOC_ERROR : RLDL_LINSYS_SOLVER_INIT_ERROR; }
  if (upload_symbolic(h)) { rldl_batch_free(h); return RLDL_MEM_ALLOC_ERROR; }
  h->num.batch = (int)batch; h->num.sigma = sigma;
  h->num.Kx = (double *)dev_alloc(sizeof(double) * (size_t)batch * (size_t)h->sym->nnzK, &ok);
  h->num.F = (double *)dev_alloc(sizeof(double) * (size_t)batch * (size_t)h->dsym.ldF, &ok);
  h->num.D = (double *)dev_alloc(sizeof(double) * (size_t)batch * (size_t)h->sym->N, &ok);
  if (h->dsym.tile_ok) {                                        /* inverse of the tail triangle in tile order (k_tile_* kernels) */
    h->num.Ti = (double *)dev_alloc(sizeof(double) * (size_t)batch * (size_t)h->dsym.ldTi, &ok);
    if (ok && !HIP_OK(hipMemset(h->num.Ti, 0, sizeof(double) * (size_t)batch * (size_t)h->dsym.ldTi))) ok = 0;
  }
  /* cache policy of the solve kernel's factor-row loads, automatic choice: rows that cannot stay in the 256 MB Infinity Cache from one
   * solve to the next anyway (more than 3/4 of it per solve) are read with non-temporal loads; rldl_batch_set_cache_policy overrides */
  h->num.nt_loads = sizeof(double) * (double)batch * (double)(h->dsym.nOp + h->sym->N + (h->dsym.tile_ok ? h->dsym.ldTi : 0)) > 0.75 * 256e6;
  /* padding slots of the plan's dense triangles are never written by the factor kernel: zero them once */
  if (ok && !HIP_OK(hipMemset(h->num.F, 0, sizeof(double) * (size_t)batch * (size_t)h->dsym.ldF))) ok = 0;
  h->num.rho_inv = (double *)dev_alloc(sizeof(double) * (size_t)batch * (size_t)h->sym->m, &ok);
  h->num.status = (int *)dev_alloc(sizeof(int) * (size_t)batch, &ok);
  h->num.fail = (int *)dev_alloc(sizeof(int), &ok);
  if (ok && !HIP_OK(hipMemset(h->num.fail, 0, sizeof(int)))) ok = 0;
  if (ok && !HIP_OK(hipHostMalloc((void **)&h->status_host, sizeof(int) * (size_t)batch, hipHostMallocDefault))) { h->status_host = 0; ok = 0; }
  if (!ok || !h->status_host) { rldl_batch_free(h); return RLDL_MEM_ALLOC_ERROR; }   /* pinned: the status read-back can be asynchronous */
  if (!HIP_OK(hipEventCreate((hipEvent_t *)&h->ev0)) || !HIP_OK(hipEventCreate((hipEvent_t *)&h->ev1))) {
    rldl_batch_free(h);
    return RLDL_MEM_ALLOC_ERROR;
  }
  *hp = h;
  return 0;
}

c_int rldl_batch_init(rldl_batch **hp, c_int batch, const csc *P, const csc *A, const c_float *d_Px,
                      const c_float *d_Ax, c_float sigma, const c_float *d_rho_vec, c_int polish, const c_int *perm,
                      void *stream) {
  rldl_batch *h;
  c_int rc = batch_create(&h, batch, P, A, sigma, polish, perm, stream);
  if (hp) *hp = 0;
  if (rc) return rc;
  if (!polish && !d_rho_vec && A->m > 0) { rldl_batch_free(h); return RLDL_LINSYS_SOLVER_INIT_ERROR; }
  if (rldl_launch_kkt_assemble(&h->dsym, &h->num, d_Px, d_Ax, polish ? 0 : d_rho_vec, 1, 0, stream)) {
    rldl_batch_free(h);
    return RLDL_LINSYS_SOLVER_INIT_ERROR;
  }
  rc = factor_and_check(h, 0);
  if (rc) { /* qdldl_interface.c:298-303: free, *sp = NULL, error code */
    rldl_batch_free(h);
    return rc;
  }
  *hp = h;
  return 0;
}

c_int rldl_batch_solve(rldl_batch *h, c_float *d_b) {
  if (!h || !d_b) return 1;
  return rldl_launch_solve(&h->dsym, &h->num, d_b, h->stream) ? 1 : 0;
}

c_int rldl_batch_update_matrices(rldl_batch *h, const c_float *d_Px, const c_float *d_Ax) {
  if (!h || h->sym->polish) return 1;
  if (rldl_launch_kkt_assemble(&h->dsym, &h->num, d_Px, d_Ax, 0, 0, 0, h->stream)) return 1;
  return factor_and_check(h, 0) ? 1 : 0; /* reference returns (QDLDL_factor < 0), :598-600 */
}

/* update_matrices without the host round trip: scatter + refactor are only enqueued; the QDLDL verdict (qdldl_interface.c:598-600)
 * is delivered by the next rldl_batch_check_status (which synchronises the stream).  keep_Px / keep_Ax (optional): arrays that
 * receive a copy of the incoming values in the same pass. */
c_int rldl_batch_update_matrices_async(rldl_batch *h, const c_float *d_Px, const c_float *d_Ax, c_float *keep_Px, c_float *keep_Ax,
                                       int *d_status_reset, int *d_rho_updates_reset) {
  if (!h || h->sym->polish) return 1;
  if (rldl_launch_kkt_assemble_keep(&h->dsym, &h->num, d_Px, d_Ax, keep_Px, keep_Ax, d_status_reset, d_rho_updates_reset, h->stream)) return 1;
  return rldl_launch_factor(&h->dsym, &h->num, 0, h->stream) ? 1 : 0;
}
c_int rldl_batch_update_rho_vec(rldl_batch *h, const c_float *d_rho_vec, const int *d_mask) {
  if (!h || (!d_rho_vec && h->sym->m > 0) || h->sym->polish) return 1;
  if (rldl_launch_kkt_assemble(&h->dsym, &h->num, 0, 0, d_rho_vec, 0, d_mask, h->stream)) return 1;
  if (d_mask) { /* asynchronous masked refactor: status is collected by the caller when it needs it */
    return rldl_launch_factor(&h->dsym, &h->num, d_mask, h->stream) ? 1 : 0;
  }
  return factor_and_check(h, 0) ? 1 : 0;
}

c_int rldl_batch_dims(const rldl_batch *h, c_int *n, c_int *m, c_int *nnzKKT, c_int *nnzL, c_int *batch) {
  if (!h) return 1;
  if (n) *n = h->sym->n;
  if (m) *m = h->sym->m;
  if (nnzKKT) *nnzKKT = h->sym->nnzK;
  if (nnzL) *nnzL = h->sym->nnzL;
  if (batch) *batch = h->batch;
  return 0;
}

static void copy_i2ll(c_int *dst, const int *src, size_t cnt) {
  size_t i;
  if (!dst) return;
  for (i = 0; i < cnt; i++) dst[i] = src[i];
}

static void export_sym(const rldl_symbolic *s, c_int *perm, c_int *etree, c_int *Lnz, c_int *Lp, c_int *Li, c_int *KKTp,
                       c_int *KKTi, c_int *PtoKKT, c_int *AtoKKT, c_int *rhotoKKT) {
  copy_i2ll(perm, s->perm, (size_t)s->N); copy_i2ll(etree, s->etree, (size_t)s->N);
  copy_i2ll(Lnz, s->Lnz, (size_t)s->N); copy_i2ll(Lp, s->Lp, (size_t)s->N + 1);
  copy_i2ll(Li, s->Li, (size_t)s->nnzL); copy_i2ll(KKTp, s->Kp, (size_t)s->N + 1);
  copy_i2ll(KKTi, s->Ki, (size_t)s->nnzK); copy_i2ll(PtoKKT, s->PtoK, (size_t)s->nnzP);
  copy_i2ll(AtoKKT, s->AtoK, (size_t)s->nnzA); copy_i2ll(rhotoKKT, s->rhotoK, (size_t)s->m);
}

c_int rldl_symbolic_analyze(const csc *P, const csc *A, c_int polish, const c_int *perm_in, c_int *nnzKKT,
                            c_int *nnzL, c_int *etree_height, c_int *perm, c_int *etree, c_int *Lnz, c_int *Lp,
                            c_int *Li, c_int *KKTp, c_int *KKTi, c_int *PtoKKT, c_int *AtoKKT, c_int *rhotoKKT) {
  rldl_symbolic *s = 0;
  if (!P || !A || P->n != A->n || P->m != P->n) return RLDL_LINSYS_SOLVER_INIT_ERROR;
  if (rldl_symbolic_create(&s, P->n, A->m, P->p, P->i, A->p, A->i, polish != 0, perm_in)) return RLDL_LINSYS_SOLVER_INIT_ERROR;
  if (nnzKKT) *nnzKKT = s->nnzK;
  if (nnzL) *nnzL = s->nnzL;
  if (etree_height) *etree_height = s->etree_height;
  export_sym(s, perm, etree, Lnz, Lp, Li, KKTp, KKTi, PtoKKT, AtoKKT, rhotoKKT);
  rldl_symbolic_free(s);
  return 0;
}

/* Host-only export of the solve plan (tests emulate the device schedule on the CPU with it).
 * meta[0..47] = {plan_ok, nS, nO, ngroups, plan_words, po_gstart, po_gflag, po_gToff, po_fsp, po_bsp, po_fsb, po_fsc,
 *                po_bsb, po_bsc, po_fsig, po_bsig, po_fcol, po_brs, po_perm, N, nnzL, arrow_ok, arrow_group, arrow_vsteps,
 *                arrow_vrows, po_avmap, po_avcol, po_avrow, nOp, tile_ok, tile_ta, tile_tq, tile_lanes, nTi, po_tlane, po_tmap,
 *                po_tislot, tile_admm_ok, tile_vslots, tile_slots, po_tpos, tile_ck[0..2], tile_tk, po_cmap, po_crow, tile_sp};
 *                blob/LtoS may be NULL. */
c_int rldl_plan_export(const csc *P, const csc *A, c_int polish, const c_int *perm_in, c_int *meta, int *blob, c_int blob_cap,
                       c_int *LtoS) {
  rldl_symbolic *s = 0;
  c_int i;
  if (!P || !A || !meta) return 1;
  if (rldl_symbolic_create(&s, P->n, A->m, P->p, P->i, A->p, A->i, polish != 0, perm_in)) return RLDL_LINSYS_SOLVER_INIT_ERROR;
  for (i = 0; i < 48; i++) meta[i] = 0;
  meta[0] = s->plan_ok; meta[1] = s->nS; meta[2] = s->nO; meta[3] = s->ngroups; meta[4] = s->plan_ok ? s->plan_words : 0;
  meta[5] = s->po_gstart; meta[6] = s->po_gflag; meta[7] = s->po_gToff; meta[8] = s->po_fsp; meta[9] = s->po_bsp;
  meta[10] = s->po_fsb; meta[11] = s->po_fsc; meta[12] = s->po_bsb; meta[13] = s->po_bsc; meta[14] = s->po_fsig;
  meta[15] = s->po_bsig; meta[16] = s->po_fcol; meta[17] = s->po_brs; meta[18] = s->po_perm; meta[19] = s->N; meta[20] = s->nnzL;
  meta[21] = s->plan_ok ? s->arrow_ok : 0; meta[22] = s->arrow_group; meta[23] = s->arrow_vsteps; meta[24] = s->arrow_vrows;
  meta[25] = s->po_avmap; meta[26] = s->po_avcol; meta[27] = s->po_avrow; meta[28] = s->nOp;
  meta[29] = s->plan_ok ? s->tile_ok : 0; meta[30] = s->tile_ta; meta[31] = s->tile_tq; meta[32] = s->tile_lanes; meta[33] = s->nTi;
  meta[34] = s->po_tlane; meta[35] = s->po_tmap; meta[36] = s->po_tislot;
  meta[37] = s->plan_ok && s->tile_ok ? s->tile_admm_ok : 0; meta[38] = s->tile_vslots; meta[39] = s->tile_slots; meta[40] = s->po_tpos;
  meta[41] = s->tile_ck[0]; meta[42] = s->tile_ck[1]; meta[43] = s->tile_ck[2]; meta[44] = s->tile_tk; meta[45] = s->po_cmap; meta[46] = s->po_crow; meta[47] = s->tile_sp;
  if (blob && s->plan_ok && blob_cap >= s->plan_words) memcpy(blob, s->plan, sizeof(int) * (size_t)s->plan_words);
  if (LtoS) for (i = 0; i < s->nnzL; i++) LtoS[i] = s->LtoS[i];
  rldl_symbolic_free(s);
  return 0;
}

c_int rldl_batch_export_symbolic(const rldl_batch *h, c_int *perm, c_int *etree, c_int *Lnz, c_int *Lp, c_int *Li,
                                 c_int *KKTp, c_int *KKTi, c_int *PtoKKT, c_int *AtoKKT, c_int *rhotoKKT) {
  const rldl_symbolic *s;
  if (!h) return 1;
  s = h->sym;
  export_sym(s, perm, etree, Lnz, Lp, Li, KKTp, KKTi, PtoKKT, AtoKKT, rhotoKKT);
  return 0;
}

c_int rldl_batch_export_factor(const rldl_batch *h, c_int inst, c_float *Lx, c_float *D, c_float *Dinv, c_float *KKTx) {
  const rldl_symbolic *s;
  size_t nF;
  double *tmp;
  int p;
  if (!h || inst < 0 || inst >= h->batch) return 1;
  s = h->sym;
  nF = (size_t)h->dsym.ldF;
  if (!HIP_OK(hipStreamSynchronize((hipStream_t)h->stream))) return 1;
  tmp = (double *)malloc(sizeof(double) * (nF ? nF : 1));
  if (!tmp) return 1;
  if (!HIP_OK(hipMemcpy(tmp, h->num.F + (size_t)inst * nF, sizeof(double) * nF, hipMemcpyDeviceToHost))) { free(tmp); return 1; }
  if (Lx) for (p = 0; p < s->nnzL; p++) Lx[p] = tmp[s->LtoS[p]];       /* back to CSC order */
  if (Dinv) memcpy(Dinv, tmp + s->nS, sizeof(double) * (size_t)s->N);
  free(tmp);
  if (D && !HIP_OK(hipMemcpy(D, h->num.D + (size_t)inst * s->N, sizeof(double) * (size_t)s->N, hipMemcpyDeviceToHost))) return 1;
  if (KKTx && !HIP_OK(hipMemcpy(KKTx, h->num.Kx + (size_t)inst * s->nnzK, sizeof(double) * (size_t)s->nnzK, hipMemcpyDeviceToHost))) return 1;
  return 0;
}

c_int rldl_batch_export_prod(const rldl_batch *h, c_int inst, c_int *meta, int *prog, int *tinfo, unsigned *tab, unsigned short *src,
                             int *blk, c_float *Ti) {
  const rldl_dev_stage *G;
  if (!h || !meta || inst < 0 || inst >= h->batch) return 1;
  G = &h->dsym.stage;
  memset(meta, 0, sizeof(c_int) * 8);
  if (!G->pv_ok || !h->num.Ti) return 1;
  if (!HIP_OK(hipStreamSynchronize((hipStream_t)h->stream))) return 1;
  meta[0] = G->pv_mode == 2 ? 2 : 1; meta[1] = G->pv_ntiles; meta[2] = G->pv_ntab; meta[3] = G->pv_nTi; meta[4] = G->nb; meta[5] = G->pv_ldT; meta[6] = G->pv_kmax;
  meta[7] = G->pv_nsteps;                                        /* meta[0]: 1 = mode-1 tables (L(b+1, b) in the coupling tiles), 2 = mode 2 (K(b+1, b)) */
  if (prog && !HIP_OK(hipMemcpy(prog, G->pv_prog, sizeof(int) * 12 * (size_t)G->pv_nsteps, hipMemcpyDeviceToHost))) return 1;
  if (tinfo && !HIP_OK(hipMemcpy(tinfo, G->pv_tinfo, sizeof(int) * 4 * (size_t)(G->pv_ntiles + 1), hipMemcpyDeviceToHost))) return 1;
  if (tab && !HIP_OK(hipMemcpy(tab, G->pv_tab, sizeof(unsigned) * (size_t)G->pv_ntab, hipMemcpyDeviceToHost))) return 1;
  if (src && G->pv_nTi && !HIP_OK(hipMemcpy(src, G->pv_src, sizeof(unsigned short) * (size_t)G->pv_nTi, hipMemcpyDeviceToHost))) return 1;
  if (blk && !HIP_OK(hipMemcpy(blk, G->pv_blk, sizeof(int) * 2 * (size_t)G->nb, hipMemcpyDeviceToHost))) return 1;
  if (Ti && G->pv_nTi && !HIP_OK(hipMemcpy(Ti, h->num.Ti + (size_t)inst * (size_t)G->pv_ldTi, sizeof(double) * (size_t)G->pv_nTi, hipMemcpyDeviceToHost))) return 1;
  return 0;
}

c_int rldl_batch_factor_status(const rldl_batch *h, c_int *status) {
  c_int b;
  if (!h || !status) return 1;
  if (!HIP_OK(hipStreamSynchronize((hipStream_t)h->stream))) return 1;
  if (!HIP_OK(hipMemcpy(h->status_host, h->num.status, sizeof(int) * (size_t)h->batch, hipMemcpyDeviceToHost))) return 1;
  for (b = 0; b < h->batch; b++) status[b] = h->status_host[b];
  return 0;
}

c_int rldl_batch_time_solve(rldl_batch *h, c_float *d_b, c_int reps, c_float *ms_per_launch) {
  c_int r;
  float ms = 0.f;
  if (!h || !d_b || reps <= 0) return 1;
  if (!HIP_OK(hipEventRecord((hipEvent_t)h->ev0, (hipStream_t)h->stream))) return 1;
  for (r = 0; r < reps; r++)
    if (rldl_launch_solve(&h->dsym, &h->num, d_b, h->stream)) return 1;
  if (!HIP_OK(hipEventRecord((hipEvent_t)h->ev1, (hipStream_t)h->stream))) return 1;
  if (!HIP_OK(hipEventSynchronize((hipEvent_t)h->ev1))) return 1;
  if (!HIP_OK(hipEventElapsedTime(&ms, (hipEvent_t)h->ev0, (hipEvent_t)h->ev1))) return 1;
  if (ms_per_launch) *ms_per_launch = (c_float)ms / (c_float)reps;
  return 0;
}

/* Cache policy of the factor rows in the plugin `solve` (k_tile_solve3).  A caller that solves again and again with ONE factorisation
 * (an ADMM loop through the plugin API, qdldl_interface.c:538-585 once per iteration) wants the rows to stay in the Infinity Cache
 * between solves: policy 1.  A caller that cycles through many handles, or whose rows exceed the cache, reads each row once before it
 * is evicted: policy 2 = non-temporal loads, which do not displace what the cache holds (measured: from HBM 19.1 -> 18.0 us at
 * B = 4096, 33.6 -> 30.4 us at 8192; cache-resident 15.1 -> 15.8 us, so it is not the default below the cache size).
 * 0 = automatic by the size of the rows of one solve (the choice made at init). */
c_int rldl_batch_set_cache_policy(rldl_batch *h, c_int policy) {
  if (!h || policy < 0 || policy > 2) return 1;
  if (policy == 0)
    h->num.nt_loads = sizeof(double) * (double)h->batch * (double)(h->dsym.nOp + h->sym->N + (h->dsym.tile_ok ? h->dsym.ldTi : 0)) > 0.75 * 256e6;
  else h->num.nt_loads = policy == 2;
  return 0;
}

/* The same timing over a ROTATION of handles: launch r solves handle r % count on its right-hand side d_b[r % count], all on the
 * stream of hs[0] (the handles must share it).  With count x (factor + tile bytes per handle) beyond the 256 MB Infinity Cache every
 * launch streams its factor rows from HBM instead of finding them cached from the previous launch of the same handle. */
c_int rldl_batch_time_solve_rotating(rldl_batch **hs, c_float **d_b, c_int count, c_int reps, c_float *ms_per_launch) {
  c_int r;
  float ms = 0.f;
  if (!hs || !d_b || count <= 0 || reps <= 0) return 1;
  for (r = 0; r < count; r++) if (!hs[r] || !d_b[r] || hs[r]->stream != hs[0]->stream) return 1;
  if (!HIP_OK(hipEventRecord((hipEvent_t)hs[0]->ev0, (hipStream_t)hs[0]->stream))) return 1;
  for (r = 0; r < reps; r++)
    if (rldl_launch_solve(&hs[r % count]->dsym, &hs[r % count]->num, d_b[r % count], hs[0]->stream)) return 1;
  if (!HIP_OK(hipEventRecord((hipEvent_t)hs[0]->ev1, (hipStream_t)hs[0]->stream))) return 1;
  if (!HIP_OK(hipEventSynchronize((hipEvent_t)hs[0]->ev1))) return 1;
  if (!HIP_OK(hipEventElapsedTime(&ms, (hipEvent_t)hs[0]->ev0, (hipEvent_t)hs[0]->ev1))) return 1;
  if (ms_per_launch) *ms_per_launch = (c_float)ms / (c_float)reps;
  return 0;
}

/* Wave timeline of ONE launch of the plugin solve: host_out[batch][8] int64 ticks of the 100 MHz device clock per instance (= wave):
 * wave start, every global load landed, forward gather done, forward tile product done, backward tile product done, scatter done,
 * result stores issued, 0.  Returns 2 when the handle's solve kernel carries no
 * timeline (only the tile kernel of arrowhead patterns does).  The traced launch is the second of two
 * back-to-back launches (warm instruction cache), so d_b is solved in place TWICE: pass a scratch right-hand side. */
c_int rldl_batch_trace_solve(rldl_batch *h, c_float *d_b, long long *host_out) {
  long long *d = 0;
  size_t bytes;
  c_int rc = 1;
  if (!h || !d_b || !host_out) return 1;
  bytes = sizeof(long long) * 8 * (size_t)h->batch;
  if (!HIP_OK(hipMalloc((void **)&d, bytes))) return RLDL_MEM_ALLOC_ERROR;
  if (HIP_OK(hipMemsetAsync(d, 0, bytes, (hipStream_t)h->stream))) {
    int lr = rldl_launch_solve_trace(&h->dsym, &h->num, d_b, d, h->stream);   /* (first launch: instruction cache warm-up; the stamps of the second stay) */
    if (lr == 0) lr = rldl_launch_solve_trace(&h->dsym, &h->num, d_b, d, h->stream);
    rc = lr < 0 ? 2 : (lr ? 1 : 0);
    if (!rc && !(HIP_OK(hipMemcpyAsync(host_out, d, bytes, hipMemcpyDeviceToHost, (hipStream_t)h->stream)) &&
                 HIP_OK(hipStreamSynchronize((hipStream_t)h->stream)))) rc = 1;
  }
  (void)hipFree(d);
  return rc;
}

/* Wave timeline of one numeric factorisation of the values the handle holds (rldl_batch_refactor's launch):
 * host_out[batch][8] = 100 MHz stamps of the arrowhead factor kernel (0 start, 1 KKT values in the workspace, 2 head
 * contributions added, 3 tail in registers, 4 tail eliminated, 5 factor row stored, 6 triangle packed, 7 tail inverse stored).
 * 2: the pattern is factorised by another kernel (no timeline).  Second of two launches, as rldl_batch_trace_solve. */
c_int rldl_batch_trace_factor(rldl_batch *h, long long *host_out) {
  long long *d = 0;
  size_t bytes;
  c_int rc = 1;
  if (!h || !host_out) return 1;
  bytes = sizeof(long long) * 8 * (size_t)h->batch;
  if (!HIP_OK(hipMalloc((void **)&d, bytes))) return RLDL_MEM_ALLOC_ERROR;
  if (HIP_OK(hipMemsetAsync(d, 0, bytes, (hipStream_t)h->stream))) {
    int lr = rldl_launch_factor_trace(&h->dsym, &h->num, d, h->stream);
    if (lr == 0) lr = rldl_launch_factor_trace(&h->dsym, &h->num, d, h->stream);
    rc = lr < 0 ? 2 : (lr ? 1 : 0);
    if (!rc && !(HIP_OK(hipMemcpyAsync(host_out, d, bytes, hipMemcpyDeviceToHost, (hipStream_t)h->stream)) &&
                 HIP_OK(hipStreamSynchronize((hipStream_t)h->stream)))) rc = 1;
  }
  (void)hipFree(d);
  return rc;
}

/* =====================================================================================
 * Legacy single-instance plugin: a batch of one with host<->device staging per call.
 * ===================================================================================== */
typedef struct {
  rldl_batch *batch;
  c_int n, m, nnzP, nnzA;
  double *d_Px, *d_Ax, *d_rho, *d_b;
} hipldl_impl;

static void legacy_free_impl(hipldl_impl *im) {
  if (!im) return;
  rldl_batch_free(im->batch);
  if (im->d_Px) (void)hipFree(im->d_Px);
  if (im->d_Ax) (void)hipFree(im->d_Ax);
  if (im->d_rho) (void)hipFree(im->d_rho);
  if (im->d_b) (void)hipFree(im->d_b);
  free(im);
}

void free_linsys_solver_hipldl(hipldl_solver *s) {
  if (!s) return;
  legacy_free_impl((hipldl_impl *)s->impl);
  free(s);
}

c_int solve_linsys_hipldl(hipldl_solver *s, c_float *b) {
  hipldl_impl *im = (hipldl_impl *)s->impl;
  size_t bytes = sizeof(double) * (size_t)(im->n + im->m);
  if (!HIP_OK(hipMemcpy(im->d_b, b, bytes, hipMemcpyHostToDevice))) return 1;
  if (rldl_batch_solve(im->batch, im->d_b)) return 1;
  if (!HIP_OK(hipStreamSynchronize((hipStream_t)im->batch->stream))) return 1;
  if (!HIP_OK(hipMemcpy(b, im->d_b, bytes, hipMemcpyDeviceToHost))) return 1;
  return 0;
}

c_int update_linsys_solver_matrices_hipldl(hipldl_solver *s, const csc *P, const csc *A) {
  hipldl_impl *im = (hipldl_impl *)s->impl;
  if (im->nnzP && !HIP_OK(hipMemcpy(im->d_Px, P->x, sizeof(double) * (size_t)im->nnzP, hipMemcpyHostToDevice))) return 1;
  if (im->nnzA && !HIP_OK(hipMemcpy(im->d_Ax, A->x, sizeof(double) * (size_t)im->nnzA, hipMemcpyHostToDevice))) return 1;
  return rldl_batch_update_matrices(im->batch, im->d_Px, im->d_Ax);
}

c_int update_linsys_solver_rho_vec_hipldl(hipldl_solver *s, const c_float *rho_vec) {
  hipldl_impl *im = (hipldl_impl *)s->impl;
  if (im->m && !HIP_OK(hipMemcpy(im->d_rho, rho_vec, sizeof(double) * (size_t)im->m, hipMemcpyHostToDevice))) return 1;
  return rldl_batch_update_rho_vec(im->batch, im->d_rho, 0);
}

c_int init_linsys_solver_hipldl(hipldl_solver **sp, const csc *P, const csc *A, c_float sigma, const c_float *rho_vec,
                                c_int polish) {
  hipldl_solver *s;
  hipldl_impl *im;
  int ok = 1;
  c_int rc;
  *sp = 0;
  if (!rldl_device_available()) return RLDL_NO_DEVICE_ERROR;
  s = (hipldl_solver *)calloc(1, sizeof(hipldl_solver));
  im = (hipldl_impl *)calloc(1, sizeof(hipldl_impl));
  if (!s || !im) { free(s); free(im); return RLDL_MEM_ALLOC_ERROR; }
  s->type = HIP_LDL_SOLVER;
  s->solve = &solve_linsys_hipldl;
  s->free = &free_linsys_solver_hipldl;
  s->update_matrices = &update_linsys_solver_matrices_hipldl;
  s->update_rho_vec = &update_linsys_solver_rho_vec_hipldl;
  s->nthreads = 1;
  s->impl = im;
  im->n = P->n; im->m = A->m; im->nnzP = P->p[P->n]; im->nnzA = A->p[A->n];
  im->d_Px = (double *)dev_upload(P->x, sizeof(double) * (size_t)im->nnzP, &ok);
  im->d_Ax = (double *)dev_upload(A->x, sizeof(double) * (size_t)im->nnzA, &ok);
  im->d_rho = polish ? (double *)dev_alloc(sizeof(double) * (size_t)im->m, &ok)
                     : (double *)dev_upload(rho_vec, sizeof(double) * (size_t)im->m, &ok);
  im->d_b = (double *)dev_alloc(sizeof(double) * (size_t)(im->n + im->m), &ok);
  if (!ok) { free_linsys_solver_hipldl(s); return RLDL_MEM_ALLOC_ERROR; }
  rc = rldl_batch_init(&im->batch, 1, P, A, im->d_Px, im->d_Ax, sigma, polish ? 0 : im->d_rho, polish, 0, 0);
  if (rc) { free_linsys_solver_hipldl(s); return rc; }
  *sp = s;
  return 0;
}
