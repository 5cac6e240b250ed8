// rldl_kernels.hip -- hand-written gfx950 (CDNA4) kernels of the batched direct KKT backend.
//
// Execution model: ONE 64-lane wavefront owns ONE QP instance (workgroup = 1 wave), so every
// cross-lane dependency is resolved inside a wave (LDS is in-order per wave) and the batch dimension
// supplies the parallelism: batch=4096 -> 4096 workgroups = 16 waves per CU on 256 CUs.
// Per-instance value arrays are instance-major and contiguous, so a wave's loads are fully coalesced
// 512-byte rows; all index arrays describe the shared sparsity pattern and stay L2-resident.
//
// Kernels and the reference code each one replaces:
//   k_kkt_assemble   : update_KKT_P/A/param2            src/kkt.c:184-222  (+ rho_inv = 1/rho, qdldl_interface.c:609-611)
//   k_factor         : QDLDL_factor (numeric LDL'), generic sparse, right-looking; restart at a column
//                      call sites qdldl_interface.c:74-77, :598-600, :616-618
//   k_stage_factor_r : the same factor by dense stage blocks for MPC patterns (k_stage_factor: LDS-resident variant)
//                      src/recursive_ldl.c:554-935, :1139-1318; restart at a block :946-1110
//   k_arrow_solve / k_plan_solve / k_solve : permute_x, QDLDL_solve, permutet_x, z-tilde epilogue
//                      qdldl_interface.c:538-585 (arrowhead plan / grouped plan / generic column sweep)
//   k_arrow_admm / k_plan_admm / k_plan_admm_loop / k_admm_iter : compute_rhs, solve, update_x, update_z + project,
//                      update_y   src/auxil.c:164-228, src/proj.c:4-14 (k_arrow_admm: a group of iterations per launch)
//   k_admm_check     : update_info, check_termination, adapt_rho, tail of osqp_solve, store_solution
//                      src/auxil.c:13-77, :243-515, :527-626, :684-789, src/osqp.c:541-641
//   k_set_rho_vec    : set_rho_vec / update_rho_vec      src/auxil.c:79-145
//   k_scale_data / k_unscale_data / k_ew_scale : scale_data, unscale_data, scaled updates   src/scaling.c:44-192
#include <hip/hip_runtime.h>
#include <utility>
#include <cstdio>
#include <cstdlib>

#include "rldl_device.h"

#define WAVE 64

#define OSQP_INFTY 1e30
#define OSQP_NAN_VALUE 2143289344.0 /* (c_float)0x7fc00000UL, include/constants.h:96 (sic) */
#define MIN_SCALING 1e-04
#define RHO_MIN 1e-06
#define RHO_MAX 1e06
#define RHO_TOL 1e-04
#define RHO_EQ_OVER_RHO_INEQ 1e03

#define ST_SOLVED 1
#define ST_SOLVED_INACCURATE 2
#define ST_PRIMAL_INFEASIBLE_INACCURATE 3
#define ST_DUAL_INFEASIBLE_INACCURATE 4
#define ST_MAX_ITER_REACHED (-2)
#define ST_PRIMAL_INFEASIBLE (-3)
#define ST_DUAL_INFEASIBLE (-4)
#define ST_NON_CVX (-7)
#define ST_UNSOLVED (-10)

namespace {

// Wave-wide reductions without LDS: four DPP steps reduce every row of 16 lanes (quad swaps, then the half-row and row mirrors: all
// lanes of a row end with the row's value), v_readlane fetches the four row values and the rest is scalar-operand arithmetic.
// (__shfl_xor on a double is two ds_bpermute_b32 + address arithmetic per step, six dependent LDS round trips per reduction; the
// residual kernel does sixteen of them.)
template <int CTRL>
__device__ __forceinline__ double dpp_perm_f64(double v) {
  const int lo = __double2loint(v), hi = __double2hiint(v);
  return __hiloint2double(__builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false), __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ double wave_rows_f64(double v, int row) {   // value of lane 16 * row (row uniform)
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 16 * row), __builtin_amdgcn_readlane(__double2loint(v), 16 * row));
}
__device__ __forceinline__ double wave_max(double v) {
  double t;
  t = dpp_perm_f64<0xB1>(v); v = t > v ? t : v;                  // quad_perm [1, 0, 3, 2]
  t = dpp_perm_f64<0x4E>(v); v = t > v ? t : v;                  // quad_perm [2, 3, 0, 1]
  t = dpp_perm_f64<0x141>(v); v = t > v ? t : v;                 // row_half_mirror
  t = dpp_perm_f64<0x140>(v); v = t > v ? t : v;                 // row_mirror
  const double a = wave_rows_f64(v, 0), b = wave_rows_f64(v, 1), c = wave_rows_f64(v, 2), d = wave_rows_f64(v, 3);
  const double ab = b > a ? b : a, cd = d > c ? d : c;
  return cd > ab ? cd : ab;
}
__device__ __forceinline__ double wave_sum(double v) {
  v += dpp_perm_f64<0xB1>(v);
  v += dpp_perm_f64<0x4E>(v);
  v += dpp_perm_f64<0x141>(v);
  v += dpp_perm_f64<0x140>(v);
  return (wave_rows_f64(v, 0) + wave_rows_f64(v, 1)) + (wave_rows_f64(v, 2) + wave_rows_f64(v, 3));
}
__device__ __forceinline__ int wave_any(int p) { return __any(p); }

// ------------------------------------------------------------------------------------------------
// KKT value scatter: Kx[PtoK[i]] = Px[i] (+sigma on the diagonal), Kx[AtoK[i]] = Ax[i],
// Kx[rhotoK[j]] = -1/rho[j], sigma-only slots.  One workgroup of 256 threads per instance.
// ------------------------------------------------------------------------------------------------
// Several workspaces in one launch (rldl_dev_multi): group of workgroup `bid` in a grid whose groups start at first[]
__device__ __forceinline__ int multi_group(const int *__restrict__ first, int ng, int bid) {   // largest g with first[g] <= bid (bisection: a set may hold thousands of groups)
  int lo = 0, hi = ng;
  while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (first[mid] <= bid) lo = mid; else hi = mid; }
  return lo;
}
__device__ __forceinline__ void kkt_assemble_body(const rldl_dev_sym &S, const rldl_dev_num &Nn, const double *__restrict__ Px,
                                                  const double *__restrict__ Ax, const double *__restrict__ rho_vec,
                                                  int set_sigma_only, const int *__restrict__ mask,
                                                  double *__restrict__ keepP, double *__restrict__ keepA, int inst) {
  if (mask && !mask[inst]) return;
  double *K = Nn.Kx + (size_t)inst * S.nnzK;
  // four rounds of loads (value, slot, diagonal flag) in flight per thread before the first store
  if (Px) {                                                      // keepP / keepA: the caller's own copy of the values is written on the way
    const double *p = Px + (size_t)inst * S.nnzP;
    double *kp = keepP ? keepP + (size_t)inst * S.nnzP : nullptr;
    for (int i0 = 0; i0 < S.nnzP; i0 += 4 * blockDim.x) {
      double v[4];
      int k[4], dg[4];
#pragma unroll
      for (int u = 0; u < 4; u++) { const int i = min(i0 + u * (int)blockDim.x + (int)threadIdx.x, S.nnzP - 1); v[u] = p[i]; k[u] = S.PtoK[i]; dg[u] = S.Pisdiag[i]; }
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int i = i0 + u * (int)blockDim.x + (int)threadIdx.x;
        if (i < S.nnzP) { K[k[u]] = v[u] + (dg[u] ? Nn.sigma : 0.0); if (kp) kp[i] = v[u]; }
      }
    }
  }
  if (Ax) {
    const double *a = Ax + (size_t)inst * S.nnzA;
    double *ka = keepA ? keepA + (size_t)inst * S.nnzA : nullptr;
    for (int i0 = 0; i0 < S.nnzA; i0 += 4 * blockDim.x) {
      double v[4];
      int k[4];
#pragma unroll
      for (int u = 0; u < 4; u++) { const int i = min(i0 + u * (int)blockDim.x + (int)threadIdx.x, S.nnzA - 1); v[u] = a[i]; k[u] = S.AtoK[i]; }
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int i = i0 + u * (int)blockDim.x + (int)threadIdx.x;
        if (i < S.nnzA) { K[k[u]] = v[u]; if (ka) ka[i] = v[u]; }
      }
    }
  }
  if (set_sigma_only)
    for (int i = threadIdx.x; i < S.nsig; i += blockDim.x) K[S.sigK[i]] = Nn.sigma;
  if (rho_vec || S.polish) {
    double *ri = Nn.rho_inv + (size_t)inst * S.m;
    for (int j = threadIdx.x; j < S.m; j += blockDim.x) {
      // polish: param2 = delta (= sigma) for every row, qdldl_interface.c:254-258
      double v = S.polish ? Nn.sigma : 1.0 / rho_vec[(size_t)inst * S.m + j];
      ri[j] = v;
      K[S.rhotoK[j]] = -v;
    }
  }
}
__global__ __launch_bounds__(256) void k_kkt_assemble(rldl_dev_sym S, rldl_dev_num Nn, const double *__restrict__ Px,
                                                      const double *__restrict__ Ax, const double *__restrict__ rho_vec,
                                                      int set_sigma_only, const int *__restrict__ mask,
                                                      double *__restrict__ keepP, double *__restrict__ keepA,
                                                      int *__restrict__ status_reset, int *__restrict__ rho_updates_reset) {
  kkt_assemble_body(S, Nn, Px, Ax, rho_vec, set_sigma_only, mask, keepP, keepA, blockIdx.x);
  // reset_info of osqp_update_P_A (auxil.c:628-645: status = OSQP_UNSOLVED, rho_updates = 0) rides along: no launch of its own
  if (status_reset && threadIdx.x == 0) { status_reset[blockIdx.x] = ST_UNSOLVED; rho_updates_reset[blockIdx.x] = 0; }
}
// New P AND A values (osqp_update_P_A): the whole KKT row of the instance is rebuilt in LDS -- P (+ sigma on its diagonal), A, the
// -1/rho entries from rho_inv, the bare sigma entries -- and written out front to back: coalesced 512-byte stores instead of one
// scattered 8-byte store per value (k_kkt_assemble's scatter into Kx: 980 of them per instance of the metric shape).  The workspace's
// own copies of the values and reset_info ride along as in k_kkt_assemble.  Dynamic LDS: nnzK doubles.
__global__ __launch_bounds__(256) void k_kkt_assemble_full(rldl_dev_sym S, rldl_dev_num Nn, const double *__restrict__ Px, const double *__restrict__ Ax,
                                                           double *__restrict__ keepP, double *__restrict__ keepA,
                                                           int *__restrict__ status_reset, int *__restrict__ rho_updates_reset) {
  extern __shared__ double kr[];
  const int inst = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
  const double *p = Px + (size_t)inst * S.nnzP, *a = Ax + (size_t)inst * S.nnzA, *ri = Nn.rho_inv + (size_t)inst * S.m;
  double *kp = keepP ? keepP + (size_t)inst * S.nnzP : nullptr, *ka = keepA ? keepA + (size_t)inst * S.nnzA : nullptr;
  constexpr int U = 4;                                           // rounds of loads in flight per thread
  for (int i0 = 0; i0 < S.nnzP; i0 += U * nt) {
    double v[U];
    int k[U], dg[U];
#pragma unroll
    for (int u = 0; u < U; u++) { const int i = min(i0 + u * nt + tid, S.nnzP - 1); v[u] = p[i]; k[u] = S.PtoK[i]; dg[u] = S.Pisdiag[i]; }
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int i = i0 + u * nt + tid;
      if (i < S.nnzP) { kr[k[u]] = v[u] + (dg[u] ? Nn.sigma : 0.0); if (kp) kp[i] = v[u]; }
    }
  }
  for (int i0 = 0; i0 < S.nnzA; i0 += U * nt) {
    double v[U];
    int k[U];
#pragma unroll
    for (int u = 0; u < U; u++) { const int i = min(i0 + u * nt + tid, S.nnzA - 1); v[u] = a[i]; k[u] = S.AtoK[i]; }
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int i = i0 + u * nt + tid;
      if (i < S.nnzA) { kr[k[u]] = v[u]; if (ka) ka[i] = v[u]; }
    }
  }
  for (int j = tid; j < S.m; j += nt) kr[S.rhotoK[j]] = -ri[j];
  for (int i = tid; i < S.nsig; i += nt) kr[S.sigK[i]] = Nn.sigma;
  __syncthreads();
  double *K = Nn.Kx + (size_t)inst * S.nnzK;
  for (int k = tid; k < S.nnzK; k += nt) K[k] = kr[k];
  if (status_reset && tid == 0) { status_reset[inst] = ST_UNSOLVED; rho_updates_reset[inst] = 0; }
}
// rho_vec of the instances whose rho moved (W.refactor) into rho_inv and the KKT values, every group of a set (update_rho_vec, qdldl_interface.c:605-619)
__global__ __launch_bounds__(256) void k_kkt_assemble_multi_rho(rldl_dev_multi M) {
  const int g = multi_group(M.first_inst, M.ngroups, blockIdx.x);
  kkt_assemble_body(M.S[g], M.N[g], nullptr, nullptr, M.W[g].rho_vec, 0, M.W[g].refactor, nullptr, nullptr, (int)blockIdx.x - M.first_inst[g]);
}
// instances of all groups that are still iterating, per counter slot (the groups keep their own counters; the host reads one array)
__global__ __launch_bounds__(RLDL_NACT_SLOTS) void k_multi_nactive(rldl_dev_multi M, int *__restrict__ out) {
  int acc = 0;
  for (int g = 0; g < M.ngroups; g++) acc += M.W[g].n_active[threadIdx.x];
  out[threadIdx.x] = acc;
}
// the same for the stacked instances of several workspaces: new P / A values of every group (osqp_multi_update_P_A)
__global__ __launch_bounds__(256) void k_kkt_assemble_multi(rldl_dev_multi M, rldl_dev_multi_pa PA) {
  const int g = multi_group(M.first_inst, M.ngroups, blockIdx.x);
  kkt_assemble_body(M.S[g], M.N[g], PA.Px[g], PA.Ax[g], nullptr, 0, nullptr, PA.keepP[g], PA.keepA[g], (int)blockIdx.x - M.first_inst[g]);
}

// ------------------------------------------------------------------------------------------------
// Numeric LDL' (right-looking, one wave per instance).  Workspace W = [L (CSC order) | D] lives in
// LDS (USE_LDS) or directly in the output array.  Column j: d = D[j]; every pair (a >= b) of the
// column's entries updates one later slot W[dst] -= w_a * w_b / d (dst precomputed on the host, all
// distinct inside a column -> no atomics), then the column is scaled by 1/d.
// Returns the QDLDL_factor contract: status = #positive pivots, or -1 on a zero pivot.
// ------------------------------------------------------------------------------------------------
template <bool USE_LDS>
__global__ __launch_bounds__(WAVE) void k_factor(rldl_dev_sym S, rldl_dev_num Nn, const int *__restrict__ mask, int c_start) {
  const int inst = blockIdx.x;
  if (mask && !mask[inst]) return;
  extern __shared__ double sh[];
  const int lane = threadIdx.x;
  const int nW = S.nnzL + S.N;
  double *F = Nn.F + (size_t)inst * S.ldF;          // plan slot order, then Dinv
  double *Dg = Nn.D + (size_t)inst * S.N;
  const double *K = Nn.Kx + (size_t)inst * S.nnzK;
  int npos = 0, bad = 0;
  // workspace element i: L entry (CSC position i) for i < nnzL, pivot D[i - nnzL] otherwise
  auto W = [&](int i) -> double & {
    if (USE_LDS) return sh[i];
    return i < S.nnzL ? F[S.LtoS[i]] : Dg[i - S.nnzL];
  };

  if (c_start <= 0) {
    for (int i = lane; i < nW; i += WAVE) W(i) = 0.0;
    __syncthreads();
    for (int k = lane; k < S.nnzK; k += WAVE) W(S.KtoW[k]) = K[k];
    __syncthreads();
  } else {
    // Restart (LDL_update_from_pivot semantics, src/recursive_ldl.c:946-1110): columns < c_start keep
    // their L and D; the trailing part is rebuilt from the KKT values plus the replayed contributions
    // of the kept columns.
    const int l0 = S.Lp[c_start];
    if (USE_LDS) {
      for (int i = lane; i < S.nnzL; i += WAVE) sh[i] = F[S.LtoS[i]];
      for (int j = lane; j < S.N; j += WAVE) sh[S.nnzL + j] = Dg[j];
    }
    __syncthreads();
    for (int i = l0 + lane; i < S.nnzL; i += WAVE) W(i) = 0.0;
    for (int j = c_start + lane; j < S.N; j += WAVE) W(S.nnzL + j) = 0.0;
    __syncthreads();
    for (int k = lane; k < S.nnzK; k += WAVE) {
      const int w = S.KtoW[k];
      if ((w >= l0 && w < S.nnzL) || w >= S.nnzL + c_start) W(w) = K[k];
    }
    __syncthreads();
    for (int c = 0; c < c_start; c++) {
      const double d = W(S.nnzL + c);
      if (d > 0.0) npos++;
      const int base = S.Lp[c], cnt = S.Lp[c + 1] - base;
      if (cnt == 0 || S.Li[base + cnt - 1] < c_start) continue;   // column does not reach the trailing part
      const long long t0 = S.Up[c], t1 = S.Up[c + 1];
      for (long long t = t0 + lane; t < t1; t += WAVE) {
        const int dst = S.Udst[t];
        if ((dst >= l0 && dst < S.nnzL) || dst >= S.nnzL + c_start) {
          const unsigned ab = S.Uab[t];
          W(dst) -= W(base + (ab & 0xffffu)) * (W(base + (ab >> 16)) * d);   // stored L is already scaled by 1/d
        }
      }
      __syncthreads();
    }
  }

  for (int j = c_start > 0 ? c_start : 0; j < S.N; j++) {
    const double d = W(S.nnzL + j);
    if (d == 0.0) { bad = 1; break; }
    if (d > 0.0) npos++;
    const int base = S.Lp[j], c = S.Lp[j + 1] - base;
    if (c == 0) continue;
    const double dinv = 1.0 / d;
    const long long t0 = S.Up[j], t1 = S.Up[j + 1];
    for (long long t = t0 + lane; t < t1; t += WAVE) {
      const unsigned ab = S.Uab[t];
      const double wa = W(base + (ab & 0xffffu)), wb = W(base + (ab >> 16));
      W(S.Udst[t]) -= wa * (wb * dinv);
    }
    __syncthreads();
    for (int a = lane; a < c; a += WAVE) W(base + a) *= dinv;
    // no barrier needed: later columns never read column j again (only the write-out below does)
  }
  __syncthreads();
  if (USE_LDS) {
    for (int i = lane; i < S.nnzL; i += WAVE) F[S.LtoS[i]] = sh[i];
    for (int j = lane; j < S.N; j += WAVE) Dg[j] = sh[S.nnzL + j];
  }
  for (int j = lane; j < S.N; j += WAVE) F[S.nS + j] = 1.0 / W(S.nnzL + j);   // Dinv rides behind the factor
  if (lane == 0) { Nn.status[inst] = bad ? -1 : npos; if (Nn.fail && (bad || npos < S.n)) atomicOr(Nn.fail, 1); }   // sticky verdict, see rldl_dev_num.fail
}

// ------------------------------------------------------------------------------------------------
// x <- L^-T D^-1 L^-1 x for one instance; xs[N] in LDS, Lv = L values (LDS copy or global).
// Forward: column sweep (lanes over the entries of column j).  Backward: the same axpy form on L'
// using the row-order access (Rp/Rj/Rpos), so neither sweep needs a cross-lane reduction.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void tri_solve(const rldl_dev_sym &S, const double *Lv, const double *__restrict__ Dinv,
                                          double *xs, int lane) {
  // generic fallback: Lv is in plan slot order, reached through LtoS
  for (int j = 0; j < S.N; j++) {
    const int base = S.Lp[j], c = S.Lp[j + 1] - base;
    if (c == 0) continue;
    const double xj = xs[j];
    for (int a = lane; a < c; a += WAVE) xs[S.Li[base + a]] -= Lv[S.LtoS[base + a]] * xj;
    __syncthreads();
  }
  for (int j = lane; j < S.N; j += WAVE) xs[j] *= Dinv[j];
  __syncthreads();
  for (int i = S.N - 1; i >= 0; i--) {
    const int base = S.Rp[i], c = S.Rp[i + 1] - base;
    if (c == 0) continue;
    const double xi = xs[i];
    for (int a = lane; a < c; a += WAVE) xs[S.Rj[base + a]] -= Lv[S.LtoS[S.Rpos[base + a]]] * xi;
    __syncthreads();
  }
}

template <bool USE_LDS>
__global__ __launch_bounds__(WAVE) void k_solve(rldl_dev_sym S, rldl_dev_num Nn, double *__restrict__ b_all) {
  const int inst = blockIdx.x, lane = threadIdx.x;
  extern __shared__ double sh[];
  double *xs = sh;               // [N]
  double *Ls = sh + S.N;         // [nS] when USE_LDS
  const double *Lg = Nn.F + (size_t)inst * S.ldF;
  const double *Dinv = Lg + S.nS;
  double *b = b_all + (size_t)inst * S.N;
  if (USE_LDS)
    for (int i = lane; i < S.nS; i += WAVE) Ls[i] = Lg[i];
  for (int j = lane; j < S.N; j += WAVE) xs[j] = b[S.perm[j]];          // permute_x  :538-541
  __syncthreads();
  tri_solve(S, USE_LDS ? Ls : Lg, Dinv, xs, lane);
  if (S.polish) {
    for (int j = lane; j < S.N; j += WAVE) b[S.perm[j]] = xs[j];        // permutet_x :544-547, raw solution :563-565
  } else {
    const double *ri = Nn.rho_inv + (size_t)inst * S.m;
    for (int j = lane; j < S.N; j += WAVE) {
      const int o = S.perm[j];
      if (o < S.n) b[o] = xs[j];                                        // x_tilde :572-574
      else b[o] += ri[o - S.n] * xs[j];                                 // z_tilde :577-579
    }
  }
}

// ------------------------------------------------------------------------------------------------
// One fused ADMM iteration per launch (auxil.c:164-228): rhs -> permuted LDS vector -> tri-solve ->
// x/z/y updates, all for one instance per wave.  Instances whose status left UNSOLVED are skipped.
// ------------------------------------------------------------------------------------------------
template <bool USE_LDS>
__global__ __launch_bounds__(WAVE) void k_admm_iter(rldl_dev_sym S, rldl_dev_num Nn, rldl_dev_admm W) {
  const int inst = blockIdx.x, lane = threadIdx.x;
  if (W.status[inst] != ST_UNSOLVED) return;
  extern __shared__ double sh[];
  double *xs = sh;               // [N] permuted rhs / solution
  double *Ls = sh + S.N;         // [nS]
  const int n = S.n, m = S.m;
  const double *Lg = Nn.F + (size_t)inst * S.ldF;
  const double *Dinv = Lg + S.nS;
  const double *ri = Nn.rho_inv + (size_t)inst * m;
  double *x = W.x + (size_t)inst * n, *z = W.z + (size_t)inst * m, *y = W.y + (size_t)inst * m;
  const double *q = W.q + (size_t)inst * n;

  if (USE_LDS)
    for (int i = lane; i < S.nS; i += WAVE) Ls[i] = Lg[i];
  // compute_rhs (auxil.c:164-178) gathered straight into permuted order (permute_x)
  for (int j = lane; j < S.N; j += WAVE) {
    const int o = S.perm[j];
    xs[j] = o < n ? W.sigma * x[o] - q[o] : z[o - n] - ri[o - n] * y[o - n];
  }
  __syncthreads();
  tri_solve(S, USE_LDS ? Ls : Lg, Dinv, xs, lane);
  // un-permute + epilogue + update_x / update_z / update_y
  const double alpha = W.alpha;
  double *dx = W.delta_x + (size_t)inst * n, *dy = W.delta_y + (size_t)inst * m;
  const double *l = W.l + (size_t)inst * m, *u = W.u + (size_t)inst * m;
  const double *rv = W.rho_vec + (size_t)inst * m;
  for (int j = lane; j < S.N; j += WAVE) {
    const int o = S.perm[j];
    const double s = xs[j];
    if (o < n) {
      const double xp = x[o];
      const double xn = alpha * s + (1.0 - alpha) * xp;       // update_x :188-201
      x[o] = xn;
      if (W.write_delta) dx[o] = xn - xp;
    } else {
      const int i = o - n;
      const double zp = z[i], yi = y[i], r = ri[i];
      const double zt = (zp - r * yi) + r * s;                 // z_tilde, qdldl_interface.c:577-579
      const double mix = alpha * zt + (1.0 - alpha) * zp;
      double zn = mix + r * yi;                                // update_z :203-215
      zn = fmin(fmax(zn, l[i]), u[i]);                         // project, proj.c:4-14
      const double d = rv[i] * (mix - zn);                     // update_y :217-228
      z[i] = zn;
      if (W.write_delta) dy[i] = d;
      y[i] = yi + d;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Residuals, termination and adaptive rho for one instance per wave.
// ------------------------------------------------------------------------------------------------
struct ResidualData {
  double pri_res, dua_res, nz, nAx, nq, nAty, nPx;
};

// LDS vectors: vx[n], vy[m], Ax[m], Px[n], Aty[n]
__device__ __forceinline__ void spmv_A(const rldl_dev_sym &S, const double *Av, const double *v, double *out, int lane) {
  for (int i = lane; i < S.m; i += WAVE) {       // row-wise: out = A v (mat_vec, lin_alg.c:241-271)
    double acc = 0.0;
    for (int p = S.Arp[i]; p < S.Arp[i + 1]; p++) acc += Av[S.Arpos[p]] * v[S.Arj[p]];
    out[i] = acc;
  }
}
__device__ __forceinline__ void spmv_At(const rldl_dev_sym &S, const double *Av, const double *v, double *out, int lane) {
  for (int j = lane; j < S.n; j += WAVE) {       // column-wise: out = A' v (mat_tpose_vec, lin_alg.c:273-322)
    double acc = 0.0;
    for (int p = S.Ap[j]; p < S.Ap[j + 1]; p++) acc += Av[p] * v[S.Ai[p]];
    out[j] = acc;
  }
}
__device__ __forceinline__ void spmv_Psym(const rldl_dev_sym &S, const double *Pv, const double *v, double *out, int lane) {
  for (int j = lane; j < S.n; j += WAVE) {       // P upper-tri: P v + P' v without the diagonal twice (auxil.c:299-303)
    double acc = 0.0;
    for (int p = S.Prp[j]; p < S.Prp[j + 1]; p++) acc += Pv[S.Prpos[p]] * v[S.Prj[p]];          // row j of the upper part
    for (int p = S.Pp[j]; p < S.Pp[j + 1]; p++) { const int i = S.Pi[p]; if (i != j) acc += Pv[p] * v[i]; }
    out[j] = acc;
  }
}
// The three products of update_info at once, entry-parallel: every lane takes one stored entry per round and adds its
// contributions with LDS atomics, so no lane walks a row (the row loops above cost one dependent index load per entry and
// leave the wave waiting on memory for every one of them).  One pass per matrix in storage order -- coalesced value loads,
// one table word (row | col << 16) per entry, four rounds of loads in flight; lanes that hit the same destination (the
// entries of one column) are serialised by the LDS unit, which costs a few cycles against the microseconds saved.
#define FLAT_U 16                       // rounds of loads in flight: 1024 entries per batch (the metric shape needs one batch per matrix)
struct FlatBatch { unsigned rc[FLAT_U]; double v[FLAT_U]; };
// (storage order: the value loads are coalesced and do not wait for a table word.  The dealt order of k_scale_data_flat would
//  make the atomics cheaper but costs a dependent, scattered value load per entry: 58 vs 52 us for this kernel, measured.)
__device__ __forceinline__ void flat_load(const unsigned *__restrict__ tab, const double *__restrict__ val, int nnz, int p0, FlatBatch &B, int lane) {
#pragma unroll
  for (int u = 0; u < FLAT_U; u++)
    if (p0 + u * WAVE < nnz) {                                   // uniform
      const int p = min(p0 + u * WAVE + lane, nnz - 1);
      B.rc[u] = tab[p]; B.v[u] = val[p];
    }
}
// which: bit 0 = A x -> vAx, bit 1 = A' y -> vAty, bit 2 = P x -> vPx (uniform; products that are off are left untouched)
__device__ __forceinline__ void spmv3_flat(const rldl_dev_sym &S, const double *__restrict__ Pg, const double *__restrict__ Ag,
                                           const double *vx, const double *vy, double *vAx, double *vPx, double *vAty, int lane,
                                           int which = 7) {
  FlatBatch A, P;
  const int nA = (which & 3) ? S.nnzA : 0, nP = (which & 4) ? S.nnzP : 0;
  flat_load(S.Afl, Ag, nA, 0, A, lane);                          // both first batches are on their way before anything waits
  flat_load(S.Pfl, Pg, nP, 0, P, lane);
  if (which & 1) for (int i = lane; i < S.m; i += WAVE) vAx[i] = 0.0;
  if (which & 2) for (int i = lane; i < S.n; i += WAVE) vAty[i] = 0.0;
  if (which & 4) for (int i = lane; i < S.n; i += WAVE) vPx[i] = 0.0;
  __syncthreads();
  for (int p0 = 0; p0 < nA; p0 += FLAT_U * WAVE) {               // A x and A' y (mat_vec / mat_tpose_vec, lin_alg.c:241-322)
    if (p0) flat_load(S.Afl, Ag, nA, p0, A, lane);
#pragma unroll
    for (int u = 0; u < FLAT_U; u++)
      if (p0 + u * WAVE + lane < nA) {
        const unsigned r = A.rc[u] & 0xffffu, c = A.rc[u] >> 16;
        if (which & 1) unsafeAtomicAdd(&vAx[r], A.v[u] * vx[c]);
        if (which & 2) unsafeAtomicAdd(&vAty[c], A.v[u] * vy[r]);
      }
  }
  for (int p0 = 0; p0 < nP; p0 += FLAT_U * WAVE) {               // P upper-tri: P x + P' x without the diagonal twice (auxil.c:299-303)
    if (p0) flat_load(S.Pfl, Pg, nP, p0, P, lane);
#pragma unroll
    for (int u = 0; u < FLAT_U; u++)
      if (p0 + u * WAVE + lane < nP) {
        const unsigned r = P.rc[u] & 0xffffu, c = P.rc[u] >> 16;
        unsafeAtomicAdd(&vPx[r], P.v[u] * vx[c]);
        if (r != c) unsafeAtomicAdd(&vPx[c], P.v[u] * vx[r]);
      }
  }
}
// The same products with the entries handed out in CHUNKS: lane t owns the R = ceil(nnz / 64) consecutive stored entries
// [t R, (t + 1) R).  In storage (CSC) order a chunk lies in one or two columns, so the contributions to the per-column results
// (A' y, the transposed part of P x) are summed in a register and leave as one or two atomics per lane, and at a given round
// neighbouring lanes sit in different columns: the atomic instructions no longer pile ~15 lanes on one word (~100 cycles
// each, see scripts/ubench_ldsatomic.hip).  Needs nnz <= FLAT_U * 64 per matrix; spmv3_flat is the general version.
__device__ __forceinline__ void chunk_load(const unsigned *__restrict__ tab, const double *__restrict__ val, int nnz, int R, FlatBatch &B, int lane) {
#pragma unroll
  for (int u = 0; u < FLAT_U; u++)
    if (u < R) {                                                 // uniform
      const unsigned p = (unsigned)min(lane * R + u, nnz - 1);
      B.rc[u] = tab[p]; B.v[u] = val[p];
    }
}
__device__ __forceinline__ void spmv3_chunk(const rldl_dev_sym &S, const double *__restrict__ Pg, const double *__restrict__ Ag,
                                            const double *vx, const double *vy, double *vAx, double *vPx, double *vAty, int lane,
                                            int which = 7) {
  FlatBatch A, P;
  const int nA = (which & 3) ? S.nnzA : 0, nP = (which & 4) ? S.nnzP : 0;
  const int RA = (nA + WAVE - 1) / WAVE, RP = (nP + WAVE - 1) / WAVE;
  chunk_load(S.Afl, Ag, nA, RA, A, lane);
  chunk_load(S.Pfl, Pg, nP, RP, P, lane);
  if (which & 1) for (int i = lane; i < S.m; i += WAVE) vAx[i] = 0.0;
  if (which & 2) for (int i = lane; i < S.n; i += WAVE) vAty[i] = 0.0;
  if (which & 4) for (int i = lane; i < S.n; i += WAVE) vPx[i] = 0.0;
  __syncthreads();
  {
    double acc = 0.0;
    unsigned cur = 0xffffffffu;
#pragma unroll
    for (int u = 0; u < FLAT_U; u++)
      if (u < RA && lane * RA + u < nA) {
        const unsigned r = A.rc[u] & 0xffffu, c = A.rc[u] >> 16;
        if (which & 1) unsafeAtomicAdd(&vAx[r], A.v[u] * vx[c]);
        if (which & 2) {
          if (c != cur) { if (cur != 0xffffffffu) unsafeAtomicAdd(&vAty[cur], acc); cur = c; acc = 0.0; }
          acc += A.v[u] * vy[r];
        }
      }
    if ((which & 2) && cur != 0xffffffffu) unsafeAtomicAdd(&vAty[cur], acc);
  }
  {
    double acc = 0.0;
    unsigned cur = 0xffffffffu;
#pragma unroll
    for (int u = 0; u < FLAT_U; u++)
      if (u < RP && lane * RP + u < nP) {
        const unsigned r = P.rc[u] & 0xffffu, c = P.rc[u] >> 16;
        unsafeAtomicAdd(&vPx[r], P.v[u] * vx[c]);
        if (c != cur) { if (cur != 0xffffffffu) unsafeAtomicAdd(&vPx[cur], acc); cur = c; acc = 0.0; }
        if (r != c) acc += P.v[u] * vx[r];
      }
    if (cur != 0xffffffffu) unsafeAtomicAdd(&vPx[cur], acc);
  }
}
__device__ __forceinline__ void spmv3_auto(const rldl_dev_sym &S, const double *__restrict__ Pg, const double *__restrict__ Ag,
                                           const double *vx, const double *vy, double *vAx, double *vPx, double *vAty, int lane,
                                           int which = 7) {
  if (S.nnzA <= FLAT_U * WAVE && S.nnzP <= FLAT_U * WAVE) spmv3_chunk(S, Pg, Ag, vx, vy, vAx, vPx, vAty, lane, which);
  else spmv3_flat(S, Pg, Ag, vx, vy, vAx, vPx, vAty, lane, which);
}
// scaled infinity norm max_i |s_i v_i| (s == nullptr: plain norm)
__device__ __forceinline__ double norm_inf_s(const double *s, const double *v, int len, int lane) {
  double mx = 0.0;
  for (int i = lane; i < len; i += WAVE) { const double a = fabs(s ? s[i] * v[i] : v[i]); mx = a > mx ? a : mx; }
  return wave_max(mx);
}

// is_primal_infeasible (auxil.c:364-424); dyp = projected delta_y (LDS scratch), tmp[n] scratch.
// E / Dinv: per-instance scaling vectors, or nullptr when the termination test runs on scaled quantities.
__device__ __forceinline__ int primal_infeasible(const rldl_dev_sym &S, const double *Av, const double *l, const double *u,
                                                 const double *dy, double *dyp, double *tmp, double eps, const double *E,
                                                 const double *Dinv, int lane) {
  for (int i = lane; i < S.m; i += WAVE) {
    double d = dy[i];
    if (u[i] > OSQP_INFTY * MIN_SCALING) {
      if (l[i] < -OSQP_INFTY * MIN_SCALING) d = 0.0; else d = fmin(d, 0.0);
    } else if (l[i] < -OSQP_INFTY * MIN_SCALING) d = fmax(d, 0.0);
    dyp[i] = d;
  }
  __syncthreads();
  const double nd = norm_inf_s(E, dyp, S.m, lane);              // ||E delta_y|| when unscaling (:394-400)
  if (!(nd > eps)) return 0;
  double lhs = 0.0;
  for (int i = lane; i < S.m; i += WAVE) lhs += u[i] * fmax(dyp[i], 0.0) + l[i] * fmin(dyp[i], 0.0);
  lhs = wave_sum(lhs);
  if (!(lhs < -eps * nd)) return 0;
  spmv_At(S, Av, dyp, tmp, lane);
  __syncthreads();
  return norm_inf_s(Dinv, tmp, S.n, lane) < eps * nd;           // Dinv A' delta_y (:412-418)
}

// is_dual_infeasible (auxil.c:426-515); tmpn[n], tmpm[m] scratch; D/Dinv/Einv/c as above (c = 1 when scaled)
__device__ __forceinline__ int dual_infeasible(const rldl_dev_sym &S, const double *Pv, const double *Av, const double *q,
                                               const double *l, const double *u, const double *dx, double *tmpn,
                                               double *tmpm, double eps, const double *D, const double *Dinv,
                                               const double *Einv, double c, int lane) {
  const double nd = norm_inf_s(D, dx, S.n, lane);
  if (!(nd > eps)) return 0;
  double qd = 0.0;
  for (int i = lane; i < S.n; i += WAVE) qd += q[i] * dx[i];
  qd = wave_sum(qd);
  if (!(qd < -c * eps * nd)) return 0;
  spmv_Psym(S, Pv, dx, tmpn, lane);
  __syncthreads();
  if (!(norm_inf_s(Dinv, tmpn, S.n, lane) < c * eps * nd)) return 0;
  spmv_A(S, Av, dx, tmpm, lane);
  __syncthreads();
  int viol = 0;
  for (int i = lane; i < S.m; i += WAVE) {
    const double adx = Einv ? Einv[i] * tmpm[i] : tmpm[i];
    if ((u[i] < OSQP_INFTY * MIN_SCALING && adx > eps * nd) || (l[i] > -OSQP_INFTY * MIN_SCALING && adx < -eps * nd)) viol = 1;
  }
  return !wave_any(viol);
}

// mode bits
#define CHK_TERMINATION 1
#define CHK_ADAPT 2
#define CHK_FINAL 4          /* tail of osqp_solve */
#define CHK_FINAL_NEEDS_INFO 8

// STAGED: the instance's P and A values are copied to LDS first (coalesced), so the three SpMVs of update_info and
// the ones of the infeasibility tests walk LDS instead of issuing dependent global loads entry by entry.
// (4 waves per SIMD with 10 spilled registers: 3 waves per SIMD without spills measured 64 instead of 47 us per 4096 final checks)
template <bool STAGED, bool MULTI = false>
__global__ __launch_bounds__(WAVE, 4) void k_admm_check(rldl_dev_sym S, rldl_dev_admm W, int iter, int mode, rldl_dev_multi M) {
  int inst = blockIdx.x;
  const int lane = threadIdx.x;
  if (MULTI) { const int g = multi_group(M.first_inst, M.ngroups, inst); S = M.S[g]; W = M.W[g]; inst -= M.first_inst[g]; }
  const int n = S.n, m = S.m;
  extern __shared__ double sh[];
  double *vx = sh, *vy = vx + n, *vz = vy + m, *vAx = vz + m, *vPx = vAx + m, *vAty = vPx + n;
  double *vdx = vAty + n, *vdy = vdx + n, *t_n = vdy + m, *t_m = t_n + n, *dyp = t_m + m;
  int st = W.status[inst];
  const int active = st == ST_UNSOLVED;
  if (!active && !(mode & CHK_FINAL)) return;

  const double *Pg = W.Px + (size_t)inst * S.nnzP, *Ag = W.Ax + (size_t)inst * S.nnzA;
  double *Pl = dyp + m, *Al = Pl + S.nnzP;
  if (STAGED) {
    for (int p = lane; p < S.nnzP; p += WAVE) Pl[p] = Pg[p];
    for (int p = lane; p < S.nnzA; p += WAVE) Al[p] = Ag[p];
  }
  const double *Pv = STAGED ? Pl : Pg, *Av = STAGED ? Al : Ag;
  const double *q = W.q + (size_t)inst * n, *l = W.l + (size_t)inst * m, *u = W.u + (size_t)inst * m;
  double *x = W.x + (size_t)inst * n, *z = W.z + (size_t)inst * m, *y = W.y + (size_t)inst * m;
  double *dxg = W.delta_x + (size_t)inst * n, *dyg = W.delta_y + (size_t)inst * m;
  // termination on unscaled quantities when the data was equilibrated (settings->scaling && !scaled_termination)
  const bool uns = W.scaling && !W.scaled_termination;
  const double *sD = uns ? W.sD + (size_t)inst * n : nullptr, *sDinv = uns ? W.sDinv + (size_t)inst * n : nullptr;
  const double *sE = uns ? W.sE + (size_t)inst * m : nullptr, *sEinv = uns ? W.sEinv + (size_t)inst * m : nullptr;
  const double sc = uns ? W.sc[inst] : 1.0, scinv = uns ? W.scinv[inst] : 1.0;

  for (int i = lane; i < n; i += WAVE) { vx[i] = x[i]; vdx[i] = dxg[i]; }
  for (int i = lane; i < m; i += WAVE) { vy[i] = y[i]; vz[i] = z[i]; vdy[i] = dyg[i]; }
  __syncthreads();

  // update_info (auxil.c:567-626): residuals
  if (!STAGED && S.flat_ok) spmv3_auto(S, Pg, Ag, vx, vy, vAx, vPx, vAty, lane);
  else {
    spmv_A(S, Av, vx, vAx, lane);
    spmv_Psym(S, Pv, vx, vPx, lane);
    spmv_At(S, Av, vy, vAty, lane);
  }
  __syncthreads();
  for (int i = lane; i < m; i += WAVE) t_m[i] = vAx[i] - vz[i];                     // primal residual vector (z_prev in the reference)
  for (int i = lane; i < n; i += WAVE) t_n[i] = q[i] + vPx[i] + vAty[i];            // dual residual vector (x_prev in the reference)
  __syncthreads();
  // raw (scaled-problem) norms: what compute_rho_estimate uses (auxil.c:13-55)
  const double prr = m ? norm_inf_s(nullptr, t_m, m, lane) : 0.0, drr = norm_inf_s(nullptr, t_n, n, lane);
  const double nzr = norm_inf_s(nullptr, vz, m, lane), nAxr = norm_inf_s(nullptr, vAx, m, lane);
  const double nqr = norm_inf_s(nullptr, q, n, lane), nAtyr = norm_inf_s(nullptr, vAty, n, lane), nPxr = norm_inf_s(nullptr, vPx, n, lane);
  // norms of the termination test (auxil.c:243-362)
  double pr = prr, dr = drr, nz = nzr, nAx = nAxr, nq = nqr, nAty = nAtyr, nPx = nPxr;
  if (uns) {
    pr = m ? norm_inf_s(sEinv, t_m, m, lane) : 0.0;
    dr = scinv * norm_inf_s(sDinv, t_n, n, lane);
    nz = norm_inf_s(sEinv, vz, m, lane); nAx = norm_inf_s(sEinv, vAx, m, lane);
    nq = scinv * norm_inf_s(sDinv, q, n, lane); nAty = scinv * norm_inf_s(sDinv, vAty, n, lane);
    nPx = scinv * norm_inf_s(sDinv, vPx, n, lane);
  }
  __syncthreads();

  const int do_info = active && (!(mode & CHK_FINAL) || (mode & CHK_FINAL_NEEDS_INFO));
  if (do_info && lane == 0) { W.pri_res[inst] = pr; W.dua_res[inst] = dr; W.iter[inst] = iter; }
  if (!do_info) { pr = W.pri_res[inst]; dr = W.dua_res[inst]; }

  // check_termination (auxil.c:684-789); up to two passes in the final call (exact, then approximate)
  const int npass = (mode & CHK_FINAL) ? 2 : ((mode & CHK_TERMINATION) ? 1 : 0);
  double obj_special = 0.0;
  int have_special = 0;
  for (int pass = 0; pass < npass && st == ST_UNSOLVED; pass++) {
    const int approx = (mode & CHK_FINAL) ? pass : 0;
    if ((mode & CHK_FINAL) && pass == 0 && !(mode & CHK_FINAL_NEEDS_INFO)) continue; // exact check already ran this iteration
    if (pr > OSQP_INFTY || dr > OSQP_INFTY) { st = ST_NON_CVX; obj_special = OSQP_NAN_VALUE; have_special = 1; break; }
    const double f = approx ? 10.0 : 1.0;
    const double ea = W.eps_abs * f, er = W.eps_rel * f, epi = W.eps_prim_inf * f, edi = W.eps_dual_inf * f;
    int prc = 0, drc = 0, pic = 0, dic = 0;
    if (m == 0) prc = 1;
    else if (pr < ea + er * fmax(nz, nAx)) prc = 1;
    else pic = primal_infeasible(S, Av, l, u, vdy, dyp, t_n, epi, sE, sDinv, lane);
    if (dr < ea + er * fmax(fmax(nq, nAty), nPx)) drc = 1;
    else dic = dual_infeasible(S, Pv, Av, q, l, u, vdx, t_n, t_m, edi, sD, sDinv, sEinv, sc, lane);
    if (prc && drc) st = approx ? ST_SOLVED_INACCURATE : ST_SOLVED;
    else if (pic) {
      st = approx ? ST_PRIMAL_INFEASIBLE_INACCURATE : ST_PRIMAL_INFEASIBLE;
      obj_special = OSQP_INFTY; have_special = 1;
      for (int i = lane; i < m; i += WAVE) dyg[i] = sE ? sE[i] * dyp[i] : dyp[i];   // certificate, unscaled (:757-760)
    } else if (dic) {
      st = approx ? ST_DUAL_INFEASIBLE_INACCURATE : ST_DUAL_INFEASIBLE;
      obj_special = -OSQP_INFTY; have_special = 1;
      if (sD) for (int i = lane; i < n; i += WAVE) { vdx[i] *= sD[i]; }              // (:772-775)
      __syncthreads();
    }
  }
  if ((mode & CHK_FINAL) && st == ST_UNSOLVED) st = ST_MAX_ITER_REACHED;   // osqp.c:567-571

  if (active && st != ST_UNSOLVED && lane == 0) {
    W.status[inst] = st;
    if (have_special) W.obj[inst] = obj_special;
    if (!(mode & CHK_FINAL)) atomicSub(&W.n_active[inst & (RLDL_NACT_SLOTS - 1)], 1);   // (nobody reads the counters after the last pass)
  }
  if (active && (st == ST_DUAL_INFEASIBLE || st == ST_DUAL_INFEASIBLE_INACCURATE))
    for (int i = lane; i < n; i += WAVE) dxg[i] = vdx[i];

  // adapt_rho (auxil.c:13-77); only for instances that keep iterating
  if ((mode & CHK_ADAPT) && st == ST_UNSOLVED) {
    double prn = prr / (fmax(nzr, nAxr) + 1e-10);
    double drn = drr / (fmax(fmax(nqr, nAtyr), nPxr) + 1e-10);
    const double rho = W.rho_cur[inst];
    double est = rho * sqrt(prn / (drn + 1e-10));
    est = fmin(fmax(est, RHO_MIN), RHO_MAX);
    if (lane == 0) W.rho_est[inst] = est;
    if (est > rho * W.adaptive_rho_tolerance || est < rho / W.adaptive_rho_tolerance) {
      // osqp_update_rho (osqp.c:1268-1319)
      const double rn = fmin(fmax(est, RHO_MIN), RHO_MAX);
      double *rv = W.rho_vec + (size_t)inst * m;
      const int *ct = W.constr_type + (size_t)inst * m;
      for (int i = lane; i < m; i += WAVE) {
        if (ct[i] == 0) rv[i] = rn;
        else if (ct[i] == 1) rv[i] = RHO_EQ_OVER_RHO_INEQ * rn;
      }
      if (lane == 0) { W.rho_cur[inst] = rn; W.rho_updates[inst] += 1; W.refactor[inst] = 1; }
    }
  }

  // tail of osqp_solve: objective, rho estimate, store_solution (osqp.c:541-633, auxil.c:527-565)
  if (mode & CHK_FINAL) {
    const int has_sol = st != ST_PRIMAL_INFEASIBLE && st != ST_PRIMAL_INFEASIBLE_INACCURATE && st != ST_DUAL_INFEASIBLE &&
                        st != ST_DUAL_INFEASIBLE_INACCURATE && st != ST_NON_CVX;
    double *sx = W.sol_x + (size_t)inst * n, *sy = W.sol_y + (size_t)inst * m;
    if (has_sol) {
      double o = 0.0;
      for (int i = lane; i < n; i += WAVE) o += (0.5 * vPx[i] + q[i]) * vx[i];   // 1/2 x'Px + q'x (auxil.c:230-241)
      o = wave_sum(o);
      if (W.scaling) o *= W.scinv[inst];
      if (lane == 0) W.obj[inst] = o;
      // unscale_solution (scaling.c:175-192)
      const double *D = W.scaling ? W.sD + (size_t)inst * n : nullptr, *E = W.scaling ? W.sE + (size_t)inst * m : nullptr;
      const double ci = W.scaling ? W.scinv[inst] : 1.0;
      for (int i = lane; i < n; i += WAVE) sx[i] = D ? D[i] * vx[i] : vx[i];
      for (int i = lane; i < m; i += WAVE) sy[i] = E ? E[i] * vy[i] * ci : vy[i];
    }
    {
      double prn = prr / (fmax(nzr, nAxr) + 1e-10), drn = drr / (fmax(fmax(nqr, nAtyr), nPxr) + 1e-10);
      double est = W.rho_cur[inst] * sqrt(prn / (drn + 1e-10));
      if (lane == 0) W.rho_est[inst] = fmin(fmax(est, RHO_MIN), RHO_MAX);
    }
    if (!has_sol) {
      for (int i = lane; i < n; i += WAVE) sx[i] = OSQP_NAN_VALUE;
      for (int i = lane; i < m; i += WAVE) sy[i] = OSQP_NAN_VALUE;
      // normalised certificates (auxil.c:543-557), iterates cold-started (:560-562)
      if (st == ST_PRIMAL_INFEASIBLE || st == ST_PRIMAL_INFEASIBLE_INACCURATE) {
        __syncthreads();
        double nv = 0.0;
        for (int i = lane; i < m; i += WAVE) { const double a = fabs(dyg[i]); nv = a > nv ? a : nv; }
        nv = wave_max(nv);
        for (int i = lane; i < m; i += WAVE) dyg[i] *= 1.0 / nv;
      }
      if (st == ST_DUAL_INFEASIBLE || st == ST_DUAL_INFEASIBLE_INACCURATE) {
        __syncthreads();
        double nv = 0.0;
        for (int i = lane; i < n; i += WAVE) { const double a = fabs(dxg[i]); nv = a > nv ? a : nv; }
        nv = wave_max(nv);
        for (int i = lane; i < n; i += WAVE) dxg[i] *= 1.0 / nv;
      }
      for (int i = lane; i < n; i += WAVE) x[i] = 0.0;
      for (int i = lane; i < m; i += WAVE) { z[i] = 0.0; y[i] = 0.0; }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Polish (src/polish.c) for every instance whose status is OSQP_SOLVED, one wave per instance.  The reference builds a
// reduced matrix Ared from the rows guessed active and factorises [P + delta I, Ared'; Ared, -delta I] from scratch.
// Here the reduced system keeps the SHARED pattern: the rows of A that are not active are zeroed (pol_Ax), their
// right-hand side is 0, so they decouple (-delta y_i = 0) and the backend's polish = 1 mode factorises all instances
// with one symbolic analysis.  k_polish_prep: form_Ared + form_rhs_red (:19-121); k_polish_resid: the right-hand side
// b - K z of one refinement step (:134-181, K without the delta terms); k_polish_finish: polished (x, z, y), normal
// cone projection (proj.c:16-29), residuals and objective at that point (update_info, auxil.c:567-626), acceptance
// test and commit (polish.c:283-325) plus store_solution for the accepted instances.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(WAVE) void k_polish_prep(rldl_dev_sym S, rldl_dev_admm W) {
  const int inst = blockIdx.x, lane = threadIdx.x, n = S.n, m = S.m;
  const int go = W.status[inst] == ST_SOLVED;
  if (lane == 0) { W.pol_mask[inst] = go; W.status_polish[inst] = 0; }
  if (!go) return;
  extern __shared__ double sh[];
  int *act = reinterpret_cast<int *>(sh);
  const double *z = W.z + (size_t)inst * m, *y = W.y + (size_t)inst * m, *l = W.l + (size_t)inst * m, *u = W.u + (size_t)inst * m;
  const double *q = W.q + (size_t)inst * n, *Av = W.Ax + (size_t)inst * S.nnzA;
  double *b = W.pol_b + (size_t)inst * S.N, *z0 = W.pol_z + (size_t)inst * S.N, *Ar = W.pol_Ax + (size_t)inst * S.nnzA;
  for (int i = lane; i < m; i += WAVE) {
    const int a = (z[i] - l[i] < -y[i]) ? 1 : ((u[i] - z[i] < y[i]) ? 2 : 0);      // lower-active / upper-active / free
    act[i] = a;
    const double v = a == 1 ? l[i] : (a == 2 ? u[i] : 0.0);
    b[n + i] = v; z0[n + i] = v;
  }
  for (int j = lane; j < n; j += WAVE) { const double v = -q[j]; b[j] = v; z0[j] = v; }
  __syncthreads();
  for (int p = lane; p < S.nnzA; p += WAVE) Ar[p] = act[S.Ai[p]] ? Av[p] : 0.0;
}

__global__ __launch_bounds__(WAVE) void k_polish_resid(rldl_dev_sym S, rldl_dev_admm W, int add_first) {
  const int inst = blockIdx.x, lane = threadIdx.x, n = S.n, m = S.m;
  if (!W.pol_mask[inst]) return;
  extern __shared__ double sh[];
  double *zx = sh, *zy = zx + n, *t1 = zy + m, *t2 = t1 + n, *t3 = t2 + n;
  const double *Pv = W.Px + (size_t)inst * S.nnzP, *Ar = W.pol_Ax + (size_t)inst * S.nnzA, *b = W.pol_b + (size_t)inst * S.N;
  double *z = W.pol_z + (size_t)inst * S.N, *r = W.pol_r + (size_t)inst * S.N;
  for (int j = lane; j < S.N; j += WAVE) {
    const double v = add_first ? z[j] + r[j] : z[j];                               // z <- z + dz of the previous step
    if (add_first) z[j] = v;
    if (j < n) zx[j] = v; else zy[j - n] = v;
  }
  __syncthreads();
  if (S.flat_ok) spmv3_auto(S, Pv, Ar, zx, zy, t3, t1, t2, lane);    // entry-parallel (see k_admm_check)
  else {
    spmv_Psym(S, Pv, zx, t1, lane);
    spmv_At(S, Ar, zy, t2, lane);
    spmv_A(S, Ar, zx, t3, lane);
  }
  __syncthreads();
  for (int j = lane; j < n; j += WAVE) r[j] = (b[j] - t1[j]) - t2[j];
  for (int i = lane; i < m; i += WAVE) r[n + i] = b[n + i] - t3[i];
}

__global__ __launch_bounds__(WAVE) void k_polish_finish(rldl_dev_sym S, rldl_dev_admm W, int add_last) {
  const int inst = blockIdx.x, lane = threadIdx.x, n = S.n, m = S.m;
  if (!W.pol_mask[inst]) return;
  extern __shared__ double sh[];
  double *px = sh, *py = px + n, *pz = py + m, *vAx = pz + m, *vPx = vAx + m, *vAty = vPx + n, *t_n = vAty + n, *t_m = t_n + n;
  const double *Pv = W.Px + (size_t)inst * S.nnzP, *Av = W.Ax + (size_t)inst * S.nnzA;
  const double *q = W.q + (size_t)inst * n, *l = W.l + (size_t)inst * m, *u = W.u + (size_t)inst * m;
  const double *zs = W.pol_z + (size_t)inst * S.N, *r = W.pol_r + (size_t)inst * S.N;
  for (int j = lane; j < n; j += WAVE) px[j] = add_last ? zs[j] + r[j] : zs[j];
  for (int i = lane; i < m; i += WAVE) py[i] = add_last ? zs[n + i] + r[n + i] : zs[n + i];     // rows that are not active carry 0
  __syncthreads();
  if (S.flat_ok) spmv3_auto(S, Pv, Av, px, py, vAx, vPx, vAty, lane, 5);             // pol->z = A pol->x (and P x, which needs px only)
  else spmv_A(S, Av, px, vAx, lane);
  __syncthreads();
  for (int i = lane; i < m; i += WAVE) {                                            // project_normalcone
    const double t = vAx[i] + py[i];
    const double zi = fmin(fmax(t, l[i]), u[i]);
    pz[i] = zi; py[i] = t - zi;
  }
  __syncthreads();
  if (S.flat_ok) spmv3_auto(S, Pv, Av, px, py, vAx, vPx, vAty, lane, 2);             // A' y with the projected y
  else {
    spmv_Psym(S, Pv, px, vPx, lane);
    spmv_At(S, Av, py, vAty, lane);
  }
  __syncthreads();
  for (int i = lane; i < m; i += WAVE) t_m[i] = vAx[i] - pz[i];
  for (int j = lane; j < n; j += WAVE) t_n[j] = q[j] + vPx[j] + vAty[j];
  __syncthreads();
  const bool uns = W.scaling && !W.scaled_termination;
  const double *sDinv = uns ? W.sDinv + (size_t)inst * n : nullptr, *sEinv = uns ? W.sEinv + (size_t)inst * m : nullptr;
  const double scinv = W.scaling ? W.scinv[inst] : 1.0;
  const double ppri = m ? norm_inf_s(sEinv, t_m, m, lane) : 0.0;
  const double pdua = (uns ? scinv : 1.0) * norm_inf_s(sDinv, t_n, n, lane);
  double o = 0.0;
  for (int j = lane; j < n; j += WAVE) o += (0.5 * vPx[j] + q[j]) * px[j];
  o = wave_sum(o) * (W.scaling ? scinv : 1.0);
  const double ipri = W.pri_res[inst], idua = W.dua_res[inst];
  const bool ok = (ppri < ipri && pdua < idua) || (ppri < ipri && idua < 1e-10) || (pdua < idua && ipri < 1e-10);
  if (lane == 0) W.status_polish[inst] = ok ? 1 : -1;
  if (!ok) return;
  if (lane == 0) { W.obj[inst] = o; W.pri_res[inst] = ppri; W.dua_res[inst] = pdua; }
  double *x = W.x + (size_t)inst * n, *z = W.z + (size_t)inst * m, *y = W.y + (size_t)inst * m;
  double *sx = W.sol_x + (size_t)inst * n, *sy = W.sol_y + (size_t)inst * m;
  const double *D = W.scaling ? W.sD + (size_t)inst * n : nullptr, *E = W.scaling ? W.sE + (size_t)inst * m : nullptr;
  for (int j = lane; j < n; j += WAVE) { x[j] = px[j]; sx[j] = D ? D[j] * px[j] : px[j]; }
  for (int i = lane; i < m; i += WAVE) { z[i] = pz[i]; y[i] = py[i]; sy[i] = E ? E[i] * py[i] * scinv : py[i]; }
}

// ------------------------------------------------------------------------------------------------
// Ruiz equilibration of one instance per wave (scale_data, src/scaling.c:44-156): `iters` passes of
// D, E <- 1/sqrt(inf-norm of the KKT columns) applied to P, A, q, followed by the cost normalisation
// step; ends with Dinv, Einv, cinv and the scaling of l, u.  Works in place on the workspace's own
// copies; column norms use the CSC and row-order maps of the shared pattern, so there are no atomics.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double limit_scaling(double v) {      // scaling.c:7-14
  v = v < MIN_SCALING ? 1.0 : v;
  return v > 1e4 ? 1e4 : v;
}
__device__ __forceinline__ double colnorm_P_sym(const rldl_dev_sym &S, const double *P, int j) {   // mat_inf_norm_cols_sym_triu
  double d = 0.0;
  for (int p = S.Pp[j]; p < S.Pp[j + 1]; p++) d = fmax(d, fabs(P[p]));
  for (int p = S.Prp[j]; p < S.Prp[j + 1]; p++) d = fmax(d, fabs(P[S.Prpos[p]]));
  return d;
}

// `STAGED`: the instance's P, A, q, D, E live in LDS for all `iters` passes (one coalesced read and one coalesced write
// of the data instead of ~6 dependent global round trips per pass); the fallback works in place in global memory.
template <bool STAGED>
__global__ __launch_bounds__(WAVE) void k_scale_data(rldl_dev_sym S, rldl_dev_admm W, double *Px_all, double *Ax_all, double *q_all,
                                                     double *l_all, double *u_all, int iters) {
  const int inst = blockIdx.x, lane = threadIdx.x, n = S.n, m = S.m;
  extern __shared__ double sh[];
  double *Dt = sh, *Et = Dt + n;
  double *Pg = Px_all + (size_t)inst * S.nnzP, *Ag = Ax_all + (size_t)inst * S.nnzA, *qg = q_all + (size_t)inst * n;
  double *l = l_all + (size_t)inst * m, *u = u_all + (size_t)inst * m;
  double *Dg = W.sD + (size_t)inst * n, *Dinv = W.sDinv + (size_t)inst * n, *Eg = W.sE + (size_t)inst * m, *Einv = W.sEinv + (size_t)inst * m;
  double *D = STAGED ? Et + m : Dg, *E = STAGED ? D + n : Eg, *q = STAGED ? E + m : qg;
  double *P = STAGED ? q + n : Pg, *A = STAGED ? P + S.nnzP : Ag;
  if (STAGED) {
    for (int p = lane; p < S.nnzP; p += WAVE) P[p] = Pg[p];
    for (int p = lane; p < S.nnzA; p += WAVE) A[p] = Ag[p];
    for (int j = lane; j < n; j += WAVE) q[j] = qg[j];
  }
  for (int j = lane; j < n; j += WAVE) D[j] = 1.0;
  for (int i = lane; i < m; i += WAVE) E[i] = 1.0;
  double c = 1.0;
  __syncthreads();
  for (int it = 0; it < iters; it++) {
    for (int j = lane; j < n; j += WAVE) {                       // norms of the columns of [P A'; A 0] (scaling.c:27-42)
      double d = colnorm_P_sym(S, P, j);
      for (int p = S.Ap[j]; p < S.Ap[j + 1]; p++) d = fmax(d, fabs(A[p]));
      Dt[j] = 1.0 / sqrt(limit_scaling(d));
    }
    for (int i = lane; i < m; i += WAVE) {
      double e = 0.0;
      for (int p = S.Arp[i]; p < S.Arp[i + 1]; p++) e = fmax(e, fabs(A[S.Arpos[p]]));
      Et[i] = 1.0 / sqrt(limit_scaling(e));
    }
    __syncthreads();
    for (int j = lane; j < n; j += WAVE) {                       // P <- D P D, A <- E A D, q <- D q (scaling.c:91-101)
      for (int p = S.Pp[j]; p < S.Pp[j + 1]; p++) P[p] = (P[p] * Dt[S.Pi[p]]) * Dt[j];
      for (int p = S.Ap[j]; p < S.Ap[j + 1]; p++) A[p] = (A[p] * Et[S.Ai[p]]) * Dt[j];
      q[j] *= Dt[j];
      D[j] *= Dt[j];
    }
    for (int i = lane; i < m; i += WAVE) E[i] *= Et[i];
    __syncthreads();
    double sum = 0.0, nq = 0.0;                                  // cost normalisation (scaling.c:110-141)
    for (int j = lane; j < n; j += WAVE) { sum += colnorm_P_sym(S, P, j); nq = fmax(nq, fabs(q[j])); }
    sum = wave_sum(sum); nq = wave_max(nq);
    double ct = fmax(sum / (double)n, limit_scaling(nq));
    ct = 1.0 / limit_scaling(ct);
    __syncthreads();
    for (int p = lane; p < S.nnzP; p += WAVE) P[p] *= ct;
    for (int j = lane; j < n; j += WAVE) q[j] *= ct;
    c *= ct;
    __syncthreads();
  }
  if (STAGED) {
    for (int p = lane; p < S.nnzP; p += WAVE) Pg[p] = P[p];
    for (int p = lane; p < S.nnzA; p += WAVE) Ag[p] = A[p];
    for (int j = lane; j < n; j += WAVE) { qg[j] = q[j]; Dg[j] = D[j]; }
    for (int i = lane; i < m; i += WAVE) Eg[i] = E[i];
  }
  for (int j = lane; j < n; j += WAVE) Dinv[j] = 1.0 / D[j];
  for (int i = lane; i < m; i += WAVE) { const double e = E[i]; Einv[i] = 1.0 / e; l[i] *= e; u[i] *= e; }
  if (lane == 0) { W.sc[inst] = c; W.scinv[inst] = 1.0 / c; }
}

// scale_data, entry-parallel: every lane keeps its share of the entries of P and A (value + row | col << 16) in registers
// for all Ruiz iterations; the column / row norms are LDS atomic maxima over the bit patterns of |v| (exact and
// order-independent, so D, E, c come out bit-identical to the column-loop version above), the rescaling of an entry
// needs two LDS reads.  PR / AR = rounds of 64 entries of P / A (compile-time: the entries live in registers).
// The column-loop version spends one dependent index load per entry and iteration on the wave's critical path
// (~1 ms per 4096 instances and 10 iterations on the metric shape); this one ~4x less.
__device__ __forceinline__ void lds_max_abs(double *dst, double v) {
  atomicMax(reinterpret_cast<unsigned long long *>(dst), (unsigned long long)__double_as_longlong(fabs(v)));
}
template <int PR, int AR, int WPE>
__global__ __launch_bounds__(WAVE, WPE) void k_scale_data_flat(rldl_dev_sym S, rldl_dev_admm W, double *Px_all, double *Ax_all, double *q_all,
                                                          double *l_all, double *u_all, int iters) {
  const int inst = blockIdx.x, lane = threadIdx.x, n = S.n, m = S.m;
  extern __shared__ double sh[];
  double *Dt = sh, *Et = Dt + n, *D = Et + m, *E = D + n, *q = E + m, *cn = q + n;
  double *Pg = Px_all + (size_t)inst * S.nnzP, *Ag = Ax_all + (size_t)inst * S.nnzA, *qg = q_all + (size_t)inst * n;
  double *l = l_all + (size_t)inst * m, *u = u_all + (size_t)inst * m;
  double *Dg = W.sD + (size_t)inst * n, *Dinv = W.sDinv + (size_t)inst * n, *Eg = W.sE + (size_t)inst * m, *Einv = W.sEinv + (size_t)inst * m;
  double vP[PR], vA[AR];
  unsigned rcP[PR], rcA[AR];
  // padding entries carry 0.0 and point at a slot of their own lane (lane mod n / m), so that they do not pile up on one address
  const unsigned padP = (unsigned)(lane % n) * 0x10001u, padA = (unsigned)(lane % (m > 0 ? m : 1)) | ((unsigned)(lane % n) << 16);
#pragma unroll
  for (int r = 0; r < PR; r++) {                                 // dealt order (few entries of a row / column per round), padding: v = 0
    const unsigned i = (unsigned)(min(r, S.Pbr - 1) * WAVE + lane), ps = S.Pbp[i];
    const bool on = r < S.Pbr && ps != 0xffffffffu;
    rcP[r] = on ? S.Pbl[i] : padP; vP[r] = on ? Pg[on ? ps : 0u] : 0.0;
  }
#pragma unroll
  for (int r = 0; r < AR; r++) {
    const unsigned i = (unsigned)(min(r, S.Abr - 1) * WAVE + lane), ps = S.Abp[i];
    const bool on = r < S.Abr && ps != 0xffffffffu;
    rcA[r] = on ? S.Abl[i] : padA; vA[r] = on ? Ag[on ? ps : 0u] : 0.0;
  }
  for (int j = lane; j < n; j += WAVE) { q[j] = qg[j]; D[j] = 1.0; }
  for (int i = lane; i < m; i += WAVE) E[i] = 1.0;
  double c = 1.0;
  for (int it = 0; it < iters; it++) {
    // (the table words are made opaque per iteration: otherwise every LDS address derived from them is hoisted out of the
    //  loop and kept in a register of its own)
#pragma unroll
    for (int r = 0; r < PR; r++) asm volatile("" : "+v"(rcP[r]));
#pragma unroll
    for (int r = 0; r < AR; r++) asm volatile("" : "+v"(rcA[r]));
    for (int j = lane; j < n; j += WAVE) Dt[j] = 0.0;
    for (int i = lane; i < m; i += WAVE) Et[i] = 0.0;
    __syncthreads();
    // norms of the columns of [P A'; A 0] (scaling.c:27-42); padding entries hold 0.0 at (0, 0) and change nothing
#pragma unroll
    for (int r = 0; r < PR; r++) {
      const unsigned ro = rcP[r] & 0xffffu, co = rcP[r] >> 16;
      lds_max_abs(&Dt[co], vP[r]);
      if (ro != co) lds_max_abs(&Dt[ro], vP[r]);
    }
#pragma unroll
    for (int r = 0; r < AR; r++) { lds_max_abs(&Dt[rcA[r] >> 16], vA[r]); lds_max_abs(&Et[rcA[r] & 0xffffu], vA[r]); }
    __syncthreads();
    for (int j = lane; j < n; j += WAVE) Dt[j] = 1.0 / sqrt(limit_scaling(Dt[j]));
    for (int i = lane; i < m; i += WAVE) Et[i] = 1.0 / sqrt(limit_scaling(Et[i]));
    __syncthreads();
    // P <- D P D, A <- E A D, q <- D q (scaling.c:91-101)
#pragma unroll
    for (int r = 0; r < PR; r++) vP[r] = (vP[r] * Dt[rcP[r] & 0xffffu]) * Dt[rcP[r] >> 16];
#pragma unroll
    for (int r = 0; r < AR; r++) vA[r] = (vA[r] * Et[rcA[r] & 0xffffu]) * Dt[rcA[r] >> 16];
    for (int j = lane; j < n; j += WAVE) { q[j] *= Dt[j]; D[j] *= Dt[j]; cn[j] = 0.0; }
    for (int i = lane; i < m; i += WAVE) E[i] *= Et[i];
    __syncthreads();
    // cost normalisation (scaling.c:110-141): mean column norm of the scaled P, inf-norm of the scaled q
#pragma unroll
    for (int r = 0; r < PR; r++) {
      const unsigned ro = rcP[r] & 0xffffu, co = rcP[r] >> 16;
      lds_max_abs(&cn[co], vP[r]);
      if (ro != co) lds_max_abs(&cn[ro], vP[r]);
    }
    __syncthreads();
    double sum = 0.0, nq = 0.0;
    for (int j = lane; j < n; j += WAVE) { sum += cn[j]; nq = fmax(nq, fabs(q[j])); }
    sum = wave_sum(sum); nq = wave_max(nq);
    double ct = fmax(sum / (double)n, limit_scaling(nq));
    ct = 1.0 / limit_scaling(ct);
#pragma unroll
    for (int r = 0; r < PR; r++) vP[r] *= ct;
    for (int j = lane; j < n; j += WAVE) q[j] *= ct;
    c *= ct;
    __syncthreads();
  }
  int lane_o = lane;
  asm volatile("" : "+v"(lane_o));                               // (keeps the store addresses from being formed before the loop)
#pragma unroll
  for (int r = 0; r < PR; r++)
    if (r < S.Pbr) { const unsigned ps = S.Pbp[(unsigned)(r * WAVE + lane_o)]; if (ps != 0xffffffffu) Pg[ps] = vP[r]; }
#pragma unroll
  for (int r = 0; r < AR; r++)
    if (r < S.Abr) { const unsigned ps = S.Abp[(unsigned)(r * WAVE + lane_o)]; if (ps != 0xffffffffu) Ag[ps] = vA[r]; }
  for (int j = lane; j < n; j += WAVE) { const double d = D[j]; qg[j] = q[j]; Dg[j] = d; Dinv[j] = 1.0 / d; }
  for (int i = lane; i < m; i += WAVE) { const double e = E[i]; Eg[i] = e; Einv[i] = 1.0 / e; l[i] *= e; u[i] *= e; }
  if (lane == 0) { W.sc[inst] = c; W.scinv[inst] = 1.0 / c; }
}

// unscale_data (scaling.c:160-173)
__global__ __launch_bounds__(WAVE) void k_unscale_data(rldl_dev_sym S, rldl_dev_admm W, double *Px_all, double *Ax_all, double *q_all,
                                                       double *l_all, double *u_all) {
  const int inst = blockIdx.x, lane = threadIdx.x, n = S.n, m = S.m;
  double *P = Px_all + (size_t)inst * S.nnzP, *A = Ax_all + (size_t)inst * S.nnzA, *q = q_all + (size_t)inst * n;
  double *l = l_all + (size_t)inst * m, *u = u_all + (size_t)inst * m;
  const double *Dinv = W.sDinv + (size_t)inst * n, *Einv = W.sEinv + (size_t)inst * m;
  const double cinv = W.scinv[inst];
  if (S.flat_ok) {                                               // entry-parallel, four rounds of loads in flight (same arithmetic per entry)
    for (int p0 = 0; p0 < S.nnzP; p0 += 4 * WAVE) {
      unsigned rc[4];
      double v[4];
#pragma unroll
      for (int k = 0; k < 4; k++) { const unsigned p = (unsigned)min(p0 + k * WAVE + lane, S.nnzP - 1); rc[k] = S.Pfl[p]; v[k] = P[p]; }
#pragma unroll
      for (int k = 0; k < 4; k++) v[k] = ((v[k] * cinv) * Dinv[rc[k] & 0xffffu]) * Dinv[rc[k] >> 16];
#pragma unroll
      for (int k = 0; k < 4; k++) { const int p = p0 + k * WAVE + lane; if (p < S.nnzP) P[(unsigned)p] = v[k]; }
    }
    for (int p0 = 0; p0 < S.nnzA; p0 += 4 * WAVE) {
      unsigned rc[4];
      double v[4];
#pragma unroll
      for (int k = 0; k < 4; k++) { const unsigned p = (unsigned)min(p0 + k * WAVE + lane, S.nnzA - 1); rc[k] = S.Afl[p]; v[k] = A[p]; }
#pragma unroll
      for (int k = 0; k < 4; k++) v[k] = (v[k] * Einv[rc[k] & 0xffffu]) * Dinv[rc[k] >> 16];
#pragma unroll
      for (int k = 0; k < 4; k++) { const int p = p0 + k * WAVE + lane; if (p < S.nnzA) A[(unsigned)p] = v[k]; }
    }
    for (int j = lane; j < n; j += WAVE) q[j] = (q[j] * cinv) * Dinv[j];
  } else {
    for (int j = lane; j < n; j += WAVE) {
      for (int p = S.Pp[j]; p < S.Pp[j + 1]; p++) P[p] = ((P[p] * cinv) * Dinv[S.Pi[p]]) * Dinv[j];
      for (int p = S.Ap[j]; p < S.Ap[j + 1]; p++) A[p] = (A[p] * Einv[S.Ai[p]]) * Dinv[j];
      q[j] = (q[j] * cinv) * Dinv[j];
    }
  }
  for (int i = lane; i < m; i += WAVE) { l[i] *= Einv[i]; u[i] *= Einv[i]; }
}

__global__ __launch_bounds__(256) void k_ew_scale(int len, double *dst, const double *src, const double *s, const double *c) {
  const int inst = blockIdx.x;
  const double cc = c ? c[inst] : 1.0;
  for (int i = threadIdx.x; i < len; i += blockDim.x) dst[(size_t)inst * len + i] = src[(size_t)inst * len + i] * s[(size_t)inst * len + i] * cc;
}

// set_rho_vec (auxil.c:79-101) when init != 0, update_rho_vec (auxil.c:103-145) otherwise:
// classify rows, fill rho_vec, raise the refactor mask when a constraint type changed.
// dst[b][i] = src[i] (one nominal row to every instance)
__global__ __launch_bounds__(256) void k_bcast_rows(int len, double *dst, const double *src) {
  const int inst = blockIdx.x;
  for (int i = threadIdx.x; i < len; i += blockDim.x) dst[(size_t)inst * len + i] = src[i];
}
// dst[b][start + i] = src[b][i] * (s ? s[b][start + i] : 1), i < cnt: a column range of instance-major rows
__global__ __launch_bounds__(256) void k_set_range(int ld, int start, int cnt, double *dst, const double *src, const double *s, const int *skip) {
  const int inst = blockIdx.x;
  if (skip && *skip) return;                                     // refused update (l > u somewhere): nothing changes
  for (int i = threadIdx.x; i < cnt; i += blockDim.x) {
    const size_t o = (size_t)inst * ld + start + i;
    dst[o] = src[(size_t)inst * cnt + i] * (s ? s[o] : 1.0);
  }
}

__global__ __launch_bounds__(WAVE) void k_set_rho_vec(rldl_dev_sym S, rldl_dev_admm W, int init) {
  const int inst = blockIdx.x, lane = threadIdx.x, m = S.m;
  const double *l = W.l + (size_t)inst * m, *u = W.u + (size_t)inst * m;
  double *rv = W.rho_vec + (size_t)inst * m;
  int *ct = W.constr_type + (size_t)inst * m;
  const double rho = W.rho_cur[inst];
  int changed = 0;
  for (int i = lane; i < m; i += WAVE) {
    int t;
    double r;
    if (l[i] < -OSQP_INFTY * MIN_SCALING && u[i] > OSQP_INFTY * MIN_SCALING) { t = -1; r = RHO_MIN; }
    else if (u[i] - l[i] < RHO_TOL) { t = 1; r = RHO_EQ_OVER_RHO_INEQ * rho; }
    else { t = 0; r = rho; }
    if (init || ct[i] != t) { ct[i] = t; rv[i] = r; changed = 1; }
  }
  changed = wave_any(changed);
  if (lane == 0) W.refactor[inst] = init ? 1 : changed;
}

__global__ __launch_bounds__(WAVE) void k_matvec_A(rldl_dev_sym S, rldl_dev_admm W, const double *__restrict__ xin,
                                                   double *__restrict__ out) {
  const int inst = blockIdx.x, lane = threadIdx.x;
  const double *Av = W.Ax + (size_t)inst * S.nnzA, *v = xin + (size_t)inst * S.n;
  double *o = out + (size_t)inst * S.m;
  if (S.flat_ok) {                                               // entry-parallel through LDS (dynamic LDS: n + m doubles), see k_admm_check
    extern __shared__ double sh[];
    double *vx = sh, *vo = sh + S.n;
    for (int j = lane; j < S.n; j += WAVE) vx[j] = v[j];
    __syncthreads();
    spmv3_auto(S, Av, Av, vx, vx, vo, vo, vo, lane, 1);
    __syncthreads();
    for (int i = lane; i < S.m; i += WAVE) o[i] = vo[i];
    return;
  }
  for (int i = lane; i < S.m; i += WAVE) {
    double acc = 0.0;
    for (int p = S.Arp[i]; p < S.Arp[i + 1]; p++) acc += Av[S.Arpos[p]] * v[S.Arj[p]];
    o[i] = acc;
  }
}


// ================================================================================================
// v2: plan-driven triangular solve (rldl_plan.c).  WPB instances per workgroup share one LDS copy of
// the plan; each wave stages its instance's factor (plan slot order) + Dinv in LDS with 16-byte loads
// and then runs, per group of <= 64 indices (one lane per index):
//   forward : per-lane gather over the out-of-group entries of its row, then the in-group dense
//             triangle swept column by column -- the pivot value is broadcast with v_readlane, the
//             column of L comes from LDS (addresses independent of x, so the reads pipeline);
//   backward: x *= Dinv folded into the load, per-lane gather over the out-of-group entries of its
//             column, then the in-group triangle swept row by row (descending) the same way.
// No atomics, no cross-lane reductions, deterministic summation order.
// ================================================================================================
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
typedef double lds_d2 __attribute__((ext_vector_type(2)));          // 16-byte LDS reads (ds_read_b128)
constexpr int LCH = 16;                                              // entries read ahead of their use in the factor kernels' LDS streams
// LDS accesses by byte address (shb = base of the wave's LDS region)
__device__ __forceinline__ double lds_ld(const char *shb, unsigned a) { return *reinterpret_cast<const double *>(shb + a); }
__device__ __forceinline__ void lds_st(char *shb, unsigned a, double v) { *reinterpret_cast<double *>(shb + a) = v; }
// The LDS double atomic of the tile / arrow kernels.  Lanes hand values to OTHER lanes through these adds, a dependence the compiler
// cannot see: as a plain intrinsic it may move a lane's later ds_read above the lane's own add whenever it proves the two addresses
// differ -- legal per thread, wrong for lane-to-lane traffic.  asm volatile with a memory clobber pins every LDS access of the wave on
// its side of the add (the wave's LDS operations then execute in program order), which is what makes the products' phases --
// reads, adds, reads of the sums -- correct; the wave_sync() fences between phases stay as documentation and cost no instruction.
__device__ __forceinline__ void lds_add(char *shb, unsigned a, double v) {
  asm volatile("ds_add_f64 %0, %1" : : "v"((unsigned)(unsigned long long)shb + a), "v"(v) : "memory");
}
__device__ __forceinline__ double readlane_f64(double v, int src) {   // src must be wave-uniform
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}


// In-group sweeps over a row-major packed triangle T (base pointer Tp, group size g); one lane per index.
// U = sweep steps whose L reads are kept in flight per batch (two batches, ping-pong).
// Branch-free inside a batch: the step's lane mask is a SCALAR (s_lshl_b64, no v_cmp) that selects the MULTIPLIER,
// which is off the dependency chain, so the chain per step is v_readlane x2 -> v_fma_f64 only.
//   forward : step a eliminates local column a; lane i > a reads L(i, a) = rowp[a], rowp = Tp + i (i-1)/2
//   backward: step s uses local row il = g-1-s; lane j < il reads L(il, j) = Tp[il (il-1)/2 + j]
// The g-1 steps are run as (g-1) mod U single steps first and whole batches after, so no step needs a range check;
// reads past a row / past the triangle return other finite values of the wave's LDS and are masked.
// Lanes >= g are never touched (they may carry live values of rows handled elsewhere).
// Multiplier of a masked-out lane: only the HIGH dword is cleared (one v_cndmask instead of two).  What is left is
// a denormal below 2^-1042, whose product with the pivot vanishes against any normal accumulator; an accumulator
// that is exactly 0 picks up at most ~1e-314 |pivot|, far below every tolerance of the path.
__device__ __forceinline__ double mask_hi(unsigned long long lanes, double t) {
  const int hi = __builtin_amdgcn_inverse_ballot_w64(lanes) ? __double2hiint(t) : 0;
  return __hiloint2double(hi, __double2loint(t));
}
template <int U>
__device__ __forceinline__ double sweep_fwd(const double *Tp, int g, int lane, double acc) {
  const int nst = g - 1, rem = nst % U, nb = nst / U;
  const int lc = lane < g ? lane : g - 1;
  const double *rowp = Tp + ((lc * (lc - 1)) >> 1);
  const unsigned long long live = g >= 64 ? ~0ull : ((1ull << g) - 1ull);
  double tA[U], tB[U];
  auto load = [&](int s0, double (&tb)[U]) {
#pragma unroll
    for (int u = 0; u < U; u++) tb[u] = rowp[s0 + u];
  };
  auto step = [&](int a, double t) {
    const unsigned long long mk = (~1ull << a) & live;           // lanes (a, g)
    const double tm = mask_hi(mk, t);
    const double xj = readlane_f64(acc, a);
    acc = fma(-tm, xj, acc);
  };
  auto proc = [&](int s0, const double (&tb)[U]) {
#pragma unroll
    for (int u = 0; u < U; u++) step(s0 + u, tb[u]);
  };
  if (rem) {
    load(0, tA);
#pragma unroll
    for (int u = 0; u < U - 1; u++)
      if (u < rem) step(u, tA[u]);
  }
  int b = 0;
  if (nb > 0) load(rem, tA);
  while (b + 2 <= nb) {
    load(rem + (b + 1) * U, tB);
    proc(rem + b * U, tA);
    if (b + 2 < nb) load(rem + (b + 2) * U, tA);
    proc(rem + (b + 1) * U, tB);
    b += 2;
  }
  if (b < nb) proc(rem + b * U, tA);
  return acc;
}
template <int U>
__device__ __forceinline__ double sweep_bwd(const double *Tp, int g, int lane, double acc) {
  const int nst = g - 1, rem = nst % U, nb = nst / U;
  const double *lanep = Tp + lane;
  double tA[U], tB[U];
  // step s uses row il = nst - s >= 1; the packed offset il (il-1)/2 is carried along, off(il-1) = off(il) - (il-1)
  auto load = [&](int s0, double (&tb)[U]) {
    int il = nst - s0;
    int off = (il * (il - 1)) >> 1;
#pragma unroll
    for (int u = 0; u < U; u++) {
      tb[u] = lanep[off];
      il -= 1;
      off -= il;                                                 // (rows below 1 only in unused slots of the odd batch)
    }
  };
  auto step = [&](int il, double t) {
    const unsigned long long mk = ~(~0ull << il);                // lanes [0, il)
    const double tm = mask_hi(mk, t);
    const double xi = readlane_f64(acc, il);
    acc = fma(-tm, xi, acc);
  };
  auto proc = [&](int s0, const double (&tb)[U]) {
#pragma unroll
    for (int u = 0; u < U; u++) step(nst - (s0 + u), tb[u]);
  };
  if (rem) {
    load(0, tA);
#pragma unroll
    for (int u = 0; u < U - 1; u++)
      if (u < rem) step(nst - u, tA[u]);
  }
  int b = 0;
  if (nb > 0) load(rem, tA);
  while (b + 2 <= nb) {
    load(rem + (b + 1) * U, tB);
    proc(rem + b * U, tA);
    if (b + 2 < nb) load(rem + (b + 2) * U, tA);
    proc(rem + (b + 1) * U, tB);
    b += 2;
  }
  if (b < nb) proc(rem + b * U, tA);
  return acc;
}

#define SU 8   // sweep depth of the plan kernels

#define GU 4   // gather steps per batch
// w: plan blob (LDS), Sv: [nS factor slots | N Dinv] (LDS), xs: [N] permuted rhs in / solution out (LDS)
__device__ __forceinline__ void plan_tri_solve(const rldl_dev_sym &S, const int *w, const double *Sv, double *xs, int lane) {
  const int ng = S.ngroups;
  const double *Dinv = Sv + S.nS;
  const int zs = S.ldF - 1;                                      // spare slot of the factor row, always 0.0
  const unsigned short *fsig = reinterpret_cast<const unsigned short *>(w + S.po_fsig);
  const unsigned short *bsig = reinterpret_cast<const unsigned short *>(w + S.po_bsig);
  const unsigned short *fcol = reinterpret_cast<const unsigned short *>(w + S.po_fcol);
  // ================= forward: L y = b =================
  for (int k = 0; k < ng; k++) {
    const int g0 = __builtin_amdgcn_readfirstlane(w[S.po_gstart + k]);
    const int g = __builtin_amdgcn_readfirstlane(w[S.po_gstart + k + 1]) - g0;
    const int fs0 = __builtin_amdgcn_readfirstlane(w[S.po_fsp + k]);
    const int nfs = __builtin_amdgcn_readfirstlane(w[S.po_fsp + k + 1]) - fs0;
    const int na = !__builtin_amdgcn_readfirstlane(w[S.po_gflag + k]) ? 0 : g - 1;
    if (nfs == 0 && na == 0) continue;                           // nothing flows into this group
    const bool act = lane < g;
    const int r = g0 + lane;
    if (nfs > 0) {
      // jagged-diagonal gather: lane i owns the row with the i-th most out-of-group entries; step t touches
      // lanes [0, cnt_t): value slot base_t + lane (consecutive -> conflict-free), column index alongside
      const int jr = act ? fsig[r] : 0;
      double ga = act ? xs[jr] : 0.0;
      for (int t0 = 0; t0 < nfs; t0 += WAVE) {
        const int nb = min(WAVE, nfs - t0);
        const int vb = lane < nb ? w[S.po_fsb + fs0 + t0 + lane] : 0;
        const int vc = lane < nb ? w[S.po_fsc + fs0 + t0 + lane] : 0;
        for (int t = 0; t < nb; t += GU) {
          double v[GU], xv[GU];
#pragma unroll
          for (int u = 0; u < GU; u++) {
            const bool ok = t + u < nb;
            const int ts = ok ? t + u : 0;
            const int base = __builtin_amdgcn_readlane(vb, ts), cnt = __builtin_amdgcn_readlane(vc, ts);
            const bool on = ok && lane < cnt;
            v[u] = Sv[(unsigned)(on ? base + lane : zs)];
            xv[u] = xs[fcol[on ? base + lane : 0]];
          }
#pragma unroll
          for (int u = 0; u < GU; u++) ga = fma(-v[u], xv[u], ga);
        }
      }
      if (act) xs[jr] = ga;
      wave_sync();
    }
    if (na > 0) {
      double acc = act ? xs[r] : 0.0;
      acc = sweep_fwd<SU>(Sv + __builtin_amdgcn_readfirstlane(w[S.po_gToff + k]), g, lane, acc);
      if (act) xs[r] = acc;
      wave_sync();
    }
  }
  // ================= backward: D^-1 then L' x = y =================
  for (int k = ng - 1; k >= 0; k--) {
    const int g0 = __builtin_amdgcn_readfirstlane(w[S.po_gstart + k]);
    const int g = __builtin_amdgcn_readfirstlane(w[S.po_gstart + k + 1]) - g0;
    const int bs0 = __builtin_amdgcn_readfirstlane(w[S.po_bsp + k]);
    const int nbs = __builtin_amdgcn_readfirstlane(w[S.po_bsp + k + 1]) - bs0;
    const int nr = !__builtin_amdgcn_readfirstlane(w[S.po_gflag + k]) ? 0 : g - 1;
    const bool act = lane < g;
    const int c = g0 + lane;
    if (nbs == 0 && nr == 0) {                                   // only the diagonal scaling is left
      if (act) xs[c] *= Dinv[c];
      wave_sync();
      continue;
    }
    if (nbs > 0) {
      const int jc = act ? bsig[c] : 0;
      double gb = act ? xs[jc] * Dinv[jc] : 0.0;
      for (int t0 = 0; t0 < nbs; t0 += WAVE) {
        const int nb = min(WAVE, nbs - t0);
        const int vb = lane < nb ? w[S.po_bsb + bs0 + t0 + lane] : 0;
        const int vc = lane < nb ? w[S.po_bsc + bs0 + t0 + lane] : 0;
        for (int t = 0; t < nb; t += GU) {
          double v[GU], xv[GU];
#pragma unroll
          for (int u = 0; u < GU; u++) {
            const bool ok = t + u < nb;
            const int ts = ok ? t + u : 0;
            const int base = __builtin_amdgcn_readlane(vb, ts), cnt = __builtin_amdgcn_readlane(vc, ts);
            const bool on = ok && lane < cnt;
            const unsigned rs = (unsigned)w[S.po_brs + (on ? base + lane : 0)];   // row | slot << 16
            v[u] = Sv[on ? (rs >> 16) : (unsigned)zs];
            xv[u] = xs[rs & 0xffffu];
          }
#pragma unroll
          for (int u = 0; u < GU; u++) gb = fma(-v[u], xv[u], gb);
        }
      }
      if (act) xs[jc] = gb;
      wave_sync();
    }
    if (nr > 0) {
      double acc = act ? (nbs > 0 ? xs[c] : xs[c] * Dinv[c]) : 0.0;
      acc = sweep_bwd<SU>(Sv + __builtin_amdgcn_readfirstlane(w[S.po_gToff + k]), g, lane, acc);
      if (act) xs[c] = acc;
      wave_sync();
    }
  }
}

// ------------------------------------------------------------------------------------------------
// stage_tri_solve -- the same L D L' solve for block-tridiagonal (MPC) patterns, by dense stage blocks (what
// QDLDL_solve does on the factor of LDL_factorize_recursive, src/recursive_ldl.c:1139-1318): one tile at a time,
// the entries of a coupling block C = L(b, b-1) / L(b+1, b) or of a diagonal block D = L_bb travel from the factor's
// slots into an LDS tile (host-built table, rldl_recursive.c: build_solve_tiles), then every lane owns one row
// (forward) or one column (backward) of the tile:
//   forward   y_b = b_b - L(b, b-1) y_{b-1}     one fma per column of the coupling tile, y_{b-1}[c] as an LDS broadcast
//             y_b <- L_bb^-1 y_b                 column sweep: pivot by v_readlane, one fma per column
//   backward  x_b = D_b^-1 y_b - L(b+1, b)' x_{b+1},  x_b <- L_bb^-T x_b      the same with the tile read by columns
// SM = compile-time bound on the block width: all inner loops run to SM without range checks or lane masks -- the
// tile is zeroed before it is filled, so whatever lies outside the live block contributes exact zeros.
// Two register sets hold the entries of the next C and the next D tile while the current ones are in use (a set is
// refilled as soon as it has been written to LDS): two tiles of global loads are in flight per wave at any time.
// F: factor row in global memory [nS slots | N Dinv], xs: [N] permuted rhs in / solution out (LDS), T: tile (LDS)
// ------------------------------------------------------------------------------------------------
#define SV_PF 6                        // rounds of 64 entries per tile (host pads every tile's table to SV_PF * 64 words)
#define SV_TW (SV_PF * WAVE)
typedef const __attribute__((address_space(4))) int *sv_cptr_t;   // constant address space: uniform reads become s_load

struct SvTile { double v[SV_PF]; unsigned p[SV_PF]; };

// table words: (byte offset inside the tile << 16) | factor slot; padding words point at the row's spare zero slot and
// at the pad column of tile row 0, so no lane needs a predicate anywhere
// (words of a tile are stored lane-major: lane t owns words [SV_PF t, SV_PF t + SV_PF) = its entries of rounds 0..SV_PF-1, so
//  the table comes in with two wide loads per lane; nr = rounds this kind of tile needs on this pattern, uniform)
__device__ __forceinline__ void sv_load_map(const unsigned *__restrict__ pk, int tile, unsigned (&m)[SV_PF], int lane) {
  const uint2 *q = reinterpret_cast<const uint2 *>(pk + (unsigned)(tile * SV_TW + lane * SV_PF));
#pragma unroll
  for (int r = 0; r < SV_PF / 2; r++) { const uint2 w = q[r]; m[2 * r] = w.x; m[2 * r + 1] = w.y; }
}
__device__ __forceinline__ void sv_load_val(const double *F, const unsigned (&m)[SV_PF], SvTile &P, int nr) {
#pragma unroll
  for (int r = 0; r < SV_PF; r++) {
    P.p[r] = m[r];
    if (r < nr) P.v[r] = F[m[r] & 0xffffu];
  }
}
template <int SM>
__device__ __forceinline__ void sv_commit(double *T, const SvTile &P, int lane, int nr) {
  double2 *T2 = reinterpret_cast<double2 *>(T);
#pragma unroll
  for (int i = 0; i < (SM * (SM + 1) / 2 + WAVE - 1) / WAVE; i++)
    if (i * WAVE + lane < SM * (SM + 1) / 2) T2[i * WAVE + lane] = make_double2(0.0, 0.0);
  wave_sync();
  char *Tb = reinterpret_cast<char *>(T);
#pragma unroll
  for (int r = 0; r < SV_PF; r++)
    if (r < nr) *reinterpret_cast<double *>(Tb + (P.p[r] >> 16)) = P.v[r];
  wave_sync();
}

// per (direction, block): sv_prog[8 k ..] = { c0, s, o0 (first index of the block the coupling tile connects to), tile id of
// the coupling tile or -1, tile id of the diagonal tile or -1, 0, 0, 0 }; forward blocks, backward blocks, 3 closing entries of -1 tiles
template <int SM>
__device__ __forceinline__ void stage_tri_solve(const rldl_dev_sym &S, const double *F, double *xs, double *T, int lane) {
  const rldl_dev_stage &G = S.stage;
  const int nb = G.nb;
  constexpr int ld = SM + 1;
  const double *Dinv = F + S.nS;
  const unsigned *pk = G.sv_pk;
  sv_cptr_t prog = (sv_cptr_t)(unsigned long long)G.sv_prog;
  SvTile PC, PD;                                               // entries of the current block's tiles (values + table words)
  unsigned mC[SV_PF], mD[SV_PF];                               // table words of the next block's tiles
  // pipeline: table words two blocks ahead of their use, factor values one block ahead
  int c0 = prog[0], s = prog[1], o0 = prog[2], tc = prog[3], td = prog[4];
  int n_c0 = prog[8], n_s = prog[9], n_o0 = prog[10], n_tc = prog[11], n_td = prog[12];
  const int nrc = G.sv_coff & 0xff, nrd = G.sv_coff >> 8;        // rounds per coupling / diagonal tile
  if (tc >= 0) { sv_load_map(pk, tc, mC, lane); sv_load_val(F, mC, PC, nrc); }
  if (td >= 0) { sv_load_map(pk, td, mD, lane); sv_load_val(F, mD, PD, nrd); }
  if (n_tc >= 0) sv_load_map(pk, n_tc, mC, lane);
  if (n_td >= 0) sv_load_map(pk, n_td, mD, lane);
  for (int k = 0; k < 2 * nb; k++) {
    if (k == nb) {                                               // ================= D^-1 between the passes =================
      for (int j0 = 0; j0 < S.N; j0 += 4 * WAVE) {
        double d[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { const int j = j0 + u * WAVE + lane; d[u] = j < S.N ? Dinv[j] : 0.0; }
#pragma unroll
        for (int u = 0; u < 4; u++) { const int j = j0 + u * WAVE + lane; if (j < S.N) xs[j] *= d[u]; }
      }
      wave_sync();
    }
    const bool fwd = k < nb;
    const int nn_tc = prog[8 * (k + 2) + 3], nn_td = prog[8 * (k + 2) + 4];
    const int lr = lane < s ? lane : 0;
    double acc = xs[c0 + lr];
    if (tc >= 0) sv_commit<SM>(T, PC, lane, nrc);
    if (n_tc >= 0) sv_load_val(F, mC, PC, nrc);                       // values of the next block's coupling tile ...
    if (nn_tc >= 0) sv_load_map(pk, nn_tc, mC, lane);            // ... and the table words of the one after
    if (tc >= 0) {
      // the neighbour block's vector sits in one register (lane c = entry c) and is broadcast by v_readlane: the LDS
      // pipe, shared by all waves of the CU, is what bounds this kernel, the VALU has room
      const double xo = xs[o0 + (lane < SM ? lane : SM - 1)];     // (lanes past the block read finite junk that meets zero tile entries)
      if (fwd) {                                                 // y_b -= L(b, b-1) y_{b-1}: own row of the tile
        const double *row = T + lr * ld;
#pragma unroll
        for (int c = 0; c < SM; c++) acc = fma(-row[c], readlane_f64(xo, c), acc);
      } else {                                                   // x_b -= L(b+1, b)' x_{b+1}: own column
        const double *col = T + lr;
#pragma unroll
        for (int c = 0; c < SM; c++) acc = fma(-col[c * ld], readlane_f64(xo, c), acc);
      }
    }
    if (td >= 0) sv_commit<SM>(T, PD, lane, nrd);
    if (n_td >= 0) sv_load_val(F, mD, PD, nrd);
    if (nn_td >= 0) sv_load_map(pk, nn_td, mD, lane);
    if (td >= 0) {
      if (fwd) {
        const double *row = T + lr * ld;
#pragma unroll
        for (int j = 0; j < SM - 1; j++) acc = fma(-row[j], readlane_f64(acc, j), acc);
      } else {
        const double *col = T + lr;
#pragma unroll
        for (int j = SM - 1; j >= 1; j--) acc = fma(-col[j * ld], readlane_f64(acc, j), acc);
      }
    }
    if ((tc >= 0 || td >= 0) && lane < s) xs[c0 + lane] = acc;
    wave_sync();
    c0 = n_c0; s = n_s; o0 = n_o0; tc = n_tc; td = n_td;
    n_c0 = prog[8 * (k + 2)]; n_s = prog[8 * (k + 2) + 1]; n_o0 = prog[8 * (k + 2) + 2]; n_tc = nn_tc; n_td = nn_td;
  }
}

// ------------------------------------------------------------------------------------------------
// stage_prod_solve -- the block tri-solve without dependent sweeps: k_stage_invert has replaced every diagonal block by the
// inverse of its unit triangle, so  y_b = L_bb^-1 (b_b - L(b, b-1) y_{b-1})  and its transpose are chains of sparse tile
// PRODUCTS (tiles D_0, C_0, D_1, ... of rldl_dev_stage.pv_*; forward pass = tiles in order, backward pass = the same tiles in
// reverse order, transposed).  Every row of a tile has its own lanes; a lane takes one entry of its row per step, streamed from
// Ti by plain coalesced loads whose addresses come from uniform data only (the step's lane mask, v_mbcnt):
//   forward   every lane gathers x[col] of its entries from LDS and sums in a register; at the tile's last step the lanes of a
//             row add their sums into x[row] (LDS atomic, as many lanes on a word as the row has lanes);
//   backward  every lane reads x[row] once and ADDS  -v x[row]  into x[col] with an LDS atomic (the host orders a lane's
//             entries so that the lanes of a step meet few equal columns).
// LDS operations of a wave execute in order, so consecutive tiles need no wait between them.  Per wave x is the only LDS
// array (stage_tri_solve above needs x + one dense tile and three passes over every tile's data in LDS).
// Load pipeline: steps come in groups of four; the loads of a group (4 values and 1 table word per lane) are issued
// RLDL_PV_RING - 1 groups ahead of their use into a register ring -- every group issues the same number of loads, in
// straight-line code, so the waits are exact counts.  The host lays the groups out as the sequence the kernel walks
// (forward, backward), so a step's descriptors are two scalar loads at consecutive addresses, fetched one step ahead.
// ------------------------------------------------------------------------------------------------
struct PvGrp { double v[4]; unsigned w; };                       // one ring slot: four values + the lane's word of the group
struct PvGI { int ti0, wo; unsigned long long m[4]; };           // what issuing a group's loads needs, uniform (SGPRs); m = lane mask per step
struct PvGC { int base, fl; unsigned long long m[4]; };          // what its products need
typedef unsigned pv_v2u __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned long long pv_u64(int lo, int hi) { return (unsigned long long)(unsigned)lo | ((unsigned long long)(unsigned)hi << 32); }
__device__ __forceinline__ PvGI pv_gdesc_i(sv_cptr_t q) {
  PvGI d;
  d.ti0 = q[0]; d.wo = q[1];
#pragma unroll
  for (int j = 0; j < 4; j++) d.m[j] = pv_u64(q[4 + 2 * j], q[5 + 2 * j]);
  return d;
}
__device__ __forceinline__ PvGC pv_gdesc_c(sv_cptr_t q, bool fwd) {
  PvGC d;
  d.base = q[2]; d.fl = q[3];
#pragma unroll
  for (int j = 0; j < 4; j++) d.m[j] = fwd ? 0ull : pv_u64(q[4 + 2 * j], q[5 + 2 * j]);   // (only the backward products use the masks)
  return d;
}
// set lanes of the uniform mask get a, the others b: one v_cndmask with the mask as its SGPR-pair condition
__device__ __forceinline__ unsigned pv_select(unsigned long long mask, unsigned a, unsigned b) {
  unsigned r;
  asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(r) : "v"(b), "v"(a), "s"(mask));
  return r;
}
// LDS double atomic add by the lanes of a uniform mask (lds32 = LDS byte address)
__device__ __forceinline__ void pv_masked_add(unsigned long long mask, unsigned lds32, double v) {
  unsigned long long keep;
  asm volatile("s_and_saveexec_b64 %0, %3\n\tds_add_f64 %1, %2\n\ts_mov_b64 exec, %0" : "=&s"(keep) : "v"(lds32), "v"(v), "s"(mask) : "memory", "scc");
}
// The loads of a group.  A lane's entry of a step sits at (entries of the earlier steps) + (set mask bits below the lane);
// lanes outside the mask get an out-of-range offset: the buffer load returns 0.0 for them without touching memory.
__device__ __forceinline__ void pv_issue(__amdgpu_buffer_rsrc_t rTi, __amdgpu_buffer_rsrc_t rTab, const PvGI &d, unsigned lane4, PvGrp &P) {
  unsigned sb = 8u * (unsigned)d.ti0;
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const unsigned mb = __builtin_amdgcn_mbcnt_hi((unsigned)(d.m[j] >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)d.m[j], 0u));
    const pv_v2u r = __builtin_amdgcn_raw_buffer_load_b64(rTi, pv_select(d.m[j], (mb << 3) + sb, 0xffffffffu), 0, 0);
    P.v[j] = __hiloint2double((int)r.y, (int)r.x);
    sb += 8u * (unsigned)__builtin_popcountll(d.m[j]);
  }
  P.w = __builtin_amdgcn_raw_buffer_load_b32(rTab, lane4 + 4u * (unsigned)d.wo, 0, 0);
}
template <bool FWD>
__device__ __forceinline__ void pv_group(char *xb, unsigned xb32, const PvGC &d, const PvGrp &P, double &acc, double &own) {
  const unsigned cb = (unsigned)d.base & 0xffffu, ra = ((unsigned)d.base >> 16) + (__builtin_amdgcn_ubfe(P.w, 24, 5) << 3);
  if (FWD) {
    if (d.fl & 1) acc = 0.0;
    double g[4];
#pragma unroll
    for (int j = 0; j < 4; j++) g[j] = lds_ld(xb, (__builtin_amdgcn_ubfe(P.w, 5 * j, 5) << 3) + cb);
#pragma unroll
    for (int j = 0; j < 4; j++) acc = fma(P.v[j], g[j], acc);    // (0.0 for lanes without an entry)
    if (d.fl & 2)                                                // last group of the tile: the lanes of a row add their sums into x[row]
      pv_masked_add(__builtin_amdgcn_ballot_w64((P.w >> 29) & 1u), xb32 + ra, -acc);
  } else {
    if (d.fl & 2) own = lds_ld(xb, ra);                          // (the backward pass meets a tile's last group first)
#pragma unroll
    for (int j = 0; j < 4; j++) pv_masked_add(d.m[j], (__builtin_amdgcn_ubfe(P.w, 5 * j, 5) << 3) + (xb32 + cb), -(P.v[j] * own));
  }
}
__device__ __forceinline__ void stage_prod_solve(const rldl_dev_sym &S, const double *F, const double *Ti, double *xs, int lane) {
  const rldl_dev_stage &G = S.stage;
  const int NS = G.pv_nsteps;                                     // forward steps [0, NS / 2), backward steps [NS / 2, NS), both multiples of the ring
  sv_cptr_t prog = (sv_cptr_t)(unsigned long long)G.pv_prog;
  const double *Dinv = F + S.nS;
  char *xb = reinterpret_cast<char *>(xs);
  const unsigned xb32 = (unsigned)(unsigned long long)xb;         // low half of a generic LDS address = LDS byte address
  const unsigned lane4 = 4u * (unsigned)lane;
  const __amdgpu_buffer_rsrc_t rTi = __builtin_amdgcn_make_buffer_rsrc((void *)Ti, 0, 8 * G.pv_nTi, 0x00020000);
  const __amdgpu_buffer_rsrc_t rTab = __builtin_amdgcn_make_buffer_rsrc((void *)G.pv_tab, 0, 4 * G.pv_ntab, 0x00020000);
  PvGrp P0, P1, P2, P3;
  double acc = 0.0, own = 0.0;
  {
    const PvGI e0 = pv_gdesc_i(prog), e1 = pv_gdesc_i(prog + 12), e2 = pv_gdesc_i(prog + 24);
    pv_issue(rTi, rTab, e0, lane4, P0); pv_issue(rTi, rTab, e1, lane4, P1); pv_issue(rTi, rTab, e2, lane4, P2);
  }
  sv_cptr_t qI = prog + 12 * (RLDL_PV_RING - 1), qC = prog;      // descriptors of the group to issue / to multiply
  PvGI dI = pv_gdesc_i(qI);
  PvGC dC = pv_gdesc_c(qC, true);
  // one step: issue the loads of step s + RING - 1 into the slot that came free, then the products of step s; both descriptors of
  // the NEXT step are fetched first (scalar loads), so no step starts by waiting for its descriptors
#define PV_STEP(FWD, PC, PN)                                                                                                    \
  {                                                                                                                            \
    qI += 12; qC += 12;                                                                                                        \
    const PvGI nI = pv_gdesc_i(qI);                                                                                            \
    const PvGC nC = pv_gdesc_c(qC, FWD);                                                                                       \
    pv_issue(rTi, rTab, dI, lane4, PN);                                                                                        \
    pv_group<FWD>(xb, xb32, dC, PC, acc, own);                                                                                 \
    dI = nI; dC = nC;                                                                                                          \
  }
#define PV_ROUND(FWD) PV_STEP(FWD, P0, P3) PV_STEP(FWD, P1, P0) PV_STEP(FWD, P2, P1) PV_STEP(FWD, P3, P2)
  for (int s = 0; s < NS / 2; s += RLDL_PV_RING) { PV_ROUND(true) }
  for (int j0 = 0; j0 < S.N; j0 += 4 * WAVE) {                   // D^-1 between the passes
    double dd[4];
#pragma unroll
    for (int u = 0; u < 4; u++) { const int j = j0 + u * WAVE + lane; dd[u] = Dinv[min(j, S.N - 1)]; }
#pragma unroll
    for (int u = 0; u < 4; u++) { const int j = j0 + u * WAVE + lane; if (j < S.N) xs[j] *= dd[u]; }
  }
  dC = pv_gdesc_c(qC, false);                                    // (the backward products use the lane masks as well)
  for (int s = NS / 2; s < NS; s += RLDL_PV_RING) { PV_ROUND(false) }
#undef PV_ROUND
#undef PV_STEP
  wave_sync();
}

// ------------------------------------------------------------------------------------------------
// stage_prod_solve2 (round 3, rldl_dev_stage.pv_mode == 2) -- the same chain of tile products on HALF the bytes.  The coupling block
// of the factor is L(b+1, b) = K(b+1, b) L_bb^-T D_b^-1 with K(b+1, b) the ORIGINAL coupling block of the permuted KKT matrix (block
// tridiagonal: no elimination touches it) -- a few dozen entries (a stage's A_k, the -I of the dynamics) against the dense
// L(b+1, b).  So the coupling tiles hold K(b+1, b) and every diagonal tile is applied twice per pass while it sits in its ring slot:
//   forward,  b ascending    [K-fwd: x_b -= K(b, b-1) a]    D-fwd: x_b <- L_bb^-1 x_b (= y_b);  a <- L_bb^-T D_b^-1 y_b
//   backward, b descending   [K-bwd: a <- -K(b+1, b)' x_{b+1}]  D-bwd: a <- L_bb^-1 a;  x_b <- L_bb^-T D_b^-1 (x_b + a)
// with a an auxiliary block vector (<= 32 doubles) behind x in the wave's LDS (y_b = L_bb^-1 (r_b - L(b, b-1) y_{b-1}) and its
// transpose, regrouped).  Per solve a wave streams the D tiles and the sparse K tiles twice -- on the MPC shape of BASELINE config 3
// about 57 KB per pass instead of 84.5 KB.  Same load pipeline as stage_prod_solve (groups of four steps, register ring, lane masks +
// v_mbcnt, buffer loads, descriptors as a sequential program: rldl_recursive.c, pv_sequence); a ring slot carries one more value, the
// lane's Dinv of the block (lanes < block size, D steps only).  A D tile of one group does both of its products from its one ring
// slot; a tile of several groups is walked twice (pv_sequence).
// ------------------------------------------------------------------------------------------------
typedef unsigned pv_v4u __attribute__((ext_vector_type(4)));
struct PvGrp2 { double v[4], d; unsigned w; };
struct PvGI2 { int ti0, wo, rb, s; unsigned long long m[4]; };   // rb: byte offset of the block's first entry in x / Dinv (D steps), s: its size (0 for K steps)
struct PvGC2 { int base, fl; unsigned long long m[4]; };
__device__ __forceinline__ PvGI2 pv_gdesc_i2(sv_cptr_t q) {
  PvGI2 d;
  const int fl = q[3];
  d.ti0 = q[0]; d.wo = q[1]; d.rb = (int)((unsigned)q[2] >> 16); d.s = (fl & 4) ? (fl >> 8) & 63 : 0;   // (kinds 1 and 3: bit 2)
#pragma unroll
  for (int j = 0; j < 4; j++) d.m[j] = pv_u64(q[4 + 2 * j], q[5 + 2 * j]);
  return d;
}
__device__ __forceinline__ PvGC2 pv_gdesc_c2(sv_cptr_t q) {
  PvGC2 d;
  d.base = q[2]; d.fl = q[3];
#pragma unroll
  for (int j = 0; j < 4; j++) d.m[j] = pv_u64(q[4 + 2 * j], q[5 + 2 * j]);
  return d;
}
__device__ __forceinline__ void pv_issue2(__amdgpu_buffer_rsrc_t rTi, __amdgpu_buffer_rsrc_t rTab, __amdgpu_buffer_rsrc_t rD, const PvGI2 &d,
                                          unsigned lane4, PvGrp2 &P) {
  unsigned sb = 8u * (unsigned)d.ti0;
#pragma unroll
  for (int pp = 0; pp < 2; pp++) {                                 // pair layout (rldl_recursive.c): the lane's entries of steps 2 pp, 2 pp + 1 side by side
    const unsigned long long mk = d.m[2 * pp];
    const unsigned mb = __builtin_amdgcn_mbcnt_hi((unsigned)(mk >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mk, 0u));
    const pv_v4u r = __builtin_amdgcn_raw_buffer_load_b128(rTi, pv_select(mk, (mb << 4) + sb, 0xffffffffu), 0, 0);
    P.v[2 * pp] = __hiloint2double((int)r.y, (int)r.x);
    P.v[2 * pp + 1] = __hiloint2double((int)r.w, (int)r.z);
    sb += 16u * (unsigned)__builtin_popcountll(mk);
  }
  P.w = __builtin_amdgcn_raw_buffer_load_b32(rTab, lane4 + 4u * (unsigned)d.wo, 0, 0);
  const pv_v2u rd = __builtin_amdgcn_raw_buffer_load_b64(rD, lane4 < 4u * (unsigned)d.s ? (unsigned)d.rb + 2u * lane4 : 0xffffffffu, 0, 0);
  P.d = __hiloint2double((int)rd.y, (int)rd.x);
}
// one tile product: FWD  dst[rb + row] -= sum_j v_j src[cb + col_j]  (lanes of a row meet in one LDS atomic at the tile's last group),
// else the transposed one  dst[cb + col_j] -= v_j src[rb + row]
template <bool FWD>
__device__ __forceinline__ void pv_prod2(char *xb, unsigned xb32, unsigned cb, unsigned rb, int fl, const unsigned long long (&m)[4], const PvGrp2 &P,
                                         double &acc, double &own) {
  const unsigned ra = rb + (__builtin_amdgcn_ubfe(P.w, 24, 5) << 3);
  if (FWD) {
    if (fl & 1) acc = 0.0;
    double g[4];
#pragma unroll
    for (int j = 0; j < 4; j++) g[j] = lds_ld(xb, (__builtin_amdgcn_ubfe(P.w, 5 * j, 5) << 3) + cb);
#pragma unroll
    for (int j = 0; j < 4; j++) acc = fma(P.v[j], g[j], acc);
    if (fl & 2) pv_masked_add(__builtin_amdgcn_ballot_w64((P.w >> 29) & 1u), xb32 + ra, -acc);
  } else {
    if (fl & 2) own = lds_ld(xb, ra);
#pragma unroll
    for (int j = 0; j < 4; j++) pv_masked_add(m[j], (__builtin_amdgcn_ubfe(P.w, 5 * j, 5) << 3) + (xb32 + cb), -(P.v[j] * own));
  }
}
__device__ __forceinline__ void pv_step2(char *xb, unsigned xb32, unsigned aux, const PvGC2 &d, const PvGrp2 &P, double &acc, double &own, int lane) {
  const int kind = (d.fl >> 2) & 3, s = (d.fl >> 8) & 63;
  const unsigned cb = (unsigned)d.base & 0xffffu, rb = (unsigned)d.base >> 16, l8 = 8u * (unsigned)lane;
  if (kind == 0) {                                               // K-fwd: x_b -= K(b, b-1) a
    pv_prod2<true>(xb, xb32, cb, rb, d.fl, d.m, P, acc, own);
  } else if (kind == 2) {                                        // K-bwd: a <- -K(b+1, b)' x_{b+1}
    if (d.fl & 2) {                                              // (the backward pass meets a tile's last group first)
      if (lane < s) lds_st(xb, aux + l8, 0.0);
      asm volatile("" ::: "memory");
    }
    pv_prod2<false>(xb, xb32, cb, rb, d.fl, d.m, P, acc, own);
  } else if (kind == 1) {                                        // D-fwd: x_b <- L_bb^-1 x_b;  a <- L_bb^-T D_b^-1 x_b
    const int f2 = (d.fl & 96) ? d.fl : 3;                         // (one-group tile: both halves from this slot)
    if (!(d.fl & 64)) {
      pv_prod2<true>(xb, xb32, rb, rb, f2, d.m, P, acc, own);
      if (f2 & 2) {
        asm volatile("" ::: "memory");
        if (lane < s) lds_st(xb, aux + l8, P.d * lds_ld(xb, rb + l8));
        asm volatile("" ::: "memory");
      }
    }
    if (!(d.fl & 32)) pv_prod2<false>(xb, xb32, aux, aux, f2, d.m, P, acc, own);
  } else {                                                       // D-bwd: a <- L_bb^-1 a;  x_b <- L_bb^-T D_b^-1 (x_b + a)
    const int f2 = (d.fl & 96) ? d.fl : 3;
    if (d.fl & 16) {                                             // no coupling tile below this block: a starts at 0
      if (lane < s) lds_st(xb, aux + l8, 0.0);
      asm volatile("" ::: "memory");
    }
    if (!(d.fl & 64)) {
      pv_prod2<true>(xb, xb32, aux, aux, f2, d.m, P, acc, own);
      if (f2 & 2) {
        asm volatile("" ::: "memory");
        if (lane < s) lds_st(xb, rb + l8, P.d * (lds_ld(xb, rb + l8) + lds_ld(xb, aux + l8)));
        asm volatile("" ::: "memory");
      }
    }
    if (!(d.fl & 32)) pv_prod2<false>(xb, xb32, rb, rb, f2, d.m, P, acc, own);
  }
}
__device__ __forceinline__ void stage_prod_solve2(const rldl_dev_sym &S, const double *F, const double *Ti, double *xs, int lane) {
  const rldl_dev_stage &G = S.stage;
  const int NS = G.pv_nsteps;                                     // forward steps [0, NS / 2), backward steps [NS / 2, NS), both multiples of the ring
  sv_cptr_t prog = (sv_cptr_t)(unsigned long long)G.pv_prog;
  char *xb = reinterpret_cast<char *>(xs);
  const unsigned xb32 = (unsigned)(unsigned long long)xb, aux = (unsigned)G.pv_aux;
  const unsigned lane4 = 4u * (unsigned)lane;
  const __amdgpu_buffer_rsrc_t rTi = __builtin_amdgcn_make_buffer_rsrc((void *)Ti, 0, 8 * G.pv_nTi, 0x00020000);
  const __amdgpu_buffer_rsrc_t rTab = __builtin_amdgcn_make_buffer_rsrc((void *)G.pv_tab, 0, 4 * G.pv_ntab, 0x00020000);
  const __amdgpu_buffer_rsrc_t rD = __builtin_amdgcn_make_buffer_rsrc((void *)(F + S.nS), 0, 8 * S.N, 0x00020000);
  PvGrp2 P0, P1, P2, P3;
  double acc = 0.0, own = 0.0;
  {
    const PvGI2 e0 = pv_gdesc_i2(prog), e1 = pv_gdesc_i2(prog + 12), e2 = pv_gdesc_i2(prog + 24);
    pv_issue2(rTi, rTab, rD, e0, lane4, P0); pv_issue2(rTi, rTab, rD, e1, lane4, P1); pv_issue2(rTi, rTab, rD, e2, lane4, P2);
  }
  sv_cptr_t qI = prog + 12 * (RLDL_PV_RING - 1), qC = prog;
  PvGI2 dI = pv_gdesc_i2(qI);
  PvGC2 dC = pv_gdesc_c2(qC);
#define PV2_STEP(PC, PN)                                                                                                        \
  {                                                                                                                            \
    qI += 12; qC += 12;                                                                                                        \
    const PvGI2 nI = pv_gdesc_i2(qI);                                                                                          \
    const PvGC2 nC = pv_gdesc_c2(qC);                                                                                          \
    pv_issue2(rTi, rTab, rD, dI, lane4, PN);                                                                                   \
    pv_step2(xb, xb32, aux, dC, PC, acc, own, lane);                                                                           \
    dI = nI; dC = nC;                                                                                                          \
  }
  for (int s = 0; s < NS; s += RLDL_PV_RING) { PV2_STEP(P0, P3) PV2_STEP(P1, P0) PV2_STEP(P2, P1) PV2_STEP(P3, P2) }
#undef PV2_STEP
  wave_sync();
}

// k_stage_invert -- behind every stage factorisation of a handle with pv_ok: per diagonal block, L_bb goes from the factor's
// slots into a dense LDS tile, lane c solves  L_bb X = e_c  by forward substitution with the rows of L_bb as LDS broadcast
// reads (one address for all lanes, two entries per read), X goes back to LDS and its entries -X(r, c), r > c, to their places
// in Ti (pv_src names the tile position behind every Ti entry); the coupling tiles are copied from their factor slots.
// Blocks before the restart block keep their tiles.  SM = compile-time bound on the block width (blocks of <= 16 take the
// 16-wide instance of the substitution).  One wave per instance.
template <int SB>
__device__ __forceinline__ void invert_block(const double *T, double *X, int ldT, int lane, int s) {
  // column `lane` of X = L_bb^-1: x[r] = [r == lane] - sum_{k < r} L(r, k) x[k]   (x[k] = 0 for k < lane falls out by itself)
  double x[SB];
#pragma unroll
  for (int r = 0; r < SB; r++) {
    double acc = r == lane ? 1.0 : 0.0;
    const double *row = T + r * ldT;                               // ldT even, T 16-byte aligned: pairs at even k are aligned
#pragma unroll
    for (int k = 0; k + 1 < r; k += 2) {
      const double2 l2 = *reinterpret_cast<const double2 *>(row + k);
      acc = fma(-l2.x, x[k], acc);
      acc = fma(-l2.y, x[k + 1], acc);
    }
    if (r & 1) acc = fma(-row[r - 1], x[r - 1], acc);
    x[r] = acc;
  }
  if (lane < s) {
#pragma unroll
    for (int r = 0; r < SB; r++) X[r * ldT + lane] = x[r];
  }
}
// (4 waves per SIMD with 14 spilled registers on the 24-wide instance: 3 waves per SIMD without spills measured 2 % slower end to end)
template <int SM>
__global__ __launch_bounds__(WAVE, 4) void k_stage_invert(rldl_dev_sym S, rldl_dev_num Nn, const int *__restrict__ mask, int b0,
                                                          const int *__restrict__ b0v) {
  const int inst = blockIdx.x, lane = threadIdx.x;
  if (mask && !mask[inst]) return;
  if (b0v) b0 = __builtin_amdgcn_readfirstlane(b0v[inst]);
  const rldl_dev_stage &G = S.stage;
  const int nb = G.nb_act > 0 ? G.nb_act : G.nb;                  // (single-store horizon handles: the live blocks)
  constexpr int ldT = SM + 2;                                     // even (= G.pv_ldT)
  constexpr int PR = (SM * (SM - 1) / 2 + WAVE - 1) / WAVE;       // rounds of 64 entries that cover a diagonal block
  extern __shared__ __attribute__((aligned(16))) double sh[];
  double *T = sh, *X = sh + SM * ldT;                             // [SM][ldT] each
  const double *F = Nn.F + (size_t)inst * S.ldF;
  double *To = Nn.Ti + (size_t)inst * G.pv_ldTi;
  sv_cptr_t tinfo = (sv_cptr_t)(unsigned long long)G.pv_tinfo;
  sv_cptr_t blk = (sv_cptr_t)(unsigned long long)G.pv_blk;
  sv_cptr_t ldp = (sv_cptr_t)(unsigned long long)G.ld_ptr;
  // The index words of block b + 1 (factor slot and tile position of every entry of L_bb, source position of every tile entry) are
  // fetched while block b is inverted: a block then waits for ONE memory round trip (its factor values), not for two in a row.
  int slotN[PR], posN[PR];
  auto load_idx = [&](int b) {
    const int e0 = b < nb ? ldp[b] : 0, e1 = b < nb ? ldp[b + 1] : 0;
#pragma unroll
    for (int r = 0; r < PR; r++) {
      const int e = e0 + r * WAVE + lane, ec = min(e, max(e1, 1) - 1);
      slotN[r] = G.ld_slot[ec];
      posN[r] = e < e1 ? G.pv_dpos[ec] : -1;                      // (the host's positions are in this kernel's tile geometry)
    }
  };
  load_idx(b0);
  {                                                              // coupling tiles L(b + 1, b), b >= b0: factor slot -> Ti, one flat list, eight rounds of loads in flight
    sv_cptr_t cpt = (sv_cptr_t)(unsigned long long)G.pv_cptr;
    const double *Cs = G.pv_mode == 2 ? Nn.Kx + (size_t)inst * S.nnzK : F;   // mode 2: the coupling tiles hold the KKT matrix's own coupling blocks
    const int k1 = cpt[b0 < nb - 1 ? nb - 1 : b0];               // C_b for b0 <= b < nb - 1 (the last live block couples to nothing live)
    const int bc0 = G.pv_mode == 2 && b0 > 0 ? b0 - 1 : b0;      // (mode 2: K(b0, b0-1) is data of the restart block's own columns)
    for (int k0 = cpt[bc0]; k0 < k1; k0 += 8 * WAVE) {
      int ti[8], sl[8];
      double v[8];
#pragma unroll
      for (int u = 0; u < 8; u++) ti[u] = G.pv_cidx[min(k0 + u * WAVE + lane, k1 - 1)];
#pragma unroll
      for (int u = 0; u < 8; u++) sl[u] = G.pv_src[ti[u]];
#pragma unroll
      for (int u = 0; u < 8; u++) { const double cv = Cs[sl[u] != 0xffff ? sl[u] : 0]; v[u] = sl[u] != 0xffff ? cv : 0.0; }   // (0xffff: padding slot of the pair layout)
#pragma unroll
      for (int u = 0; u < 8; u++) if (k0 + u * WAVE + lane < k1) To[ti[u]] = v[u];
    }
  }
  for (int b = b0; b < nb; b++) {
    const int td = blk[2 * b];
    const int t0 = td >= 0 ? tinfo[4 * td] : 0, E = td >= 0 ? tinfo[4 * td + 1] : 0;
    double val[PR];
    int pos[PR], src[PR];
#pragma unroll
    for (int r = 0; r < PR; r++) {                               // this block's factor values start travelling (and the sources of its tile entries)
      val[r] = F[slotN[r]]; pos[r] = posN[r];
      src[r] = r * WAVE + lane < E ? (int)G.pv_src[t0 + min(r * WAVE + lane, max(E, 1) - 1)] : -1;
    }
    load_idx(b + 1);
    if (td < 0) continue;
    const int s = G.bs[b + 1] - G.bs[b];
    for (int p = lane; p < SM * ldT; p += WAVE) T[p] = 0.0;
    wave_sync();
#pragma unroll
    for (int r = 0; r < PR; r++) if (pos[r] >= 0) T[pos[r]] = val[r];
    wave_sync();
    if (SM > 16 && s <= 16) invert_block<16>(T, X, ldT, lane, s);
    else invert_block<SM>(T, X, ldT, lane, s);
    wave_sync();
#pragma unroll
    for (int r = 0; r < PR; r++) if (src[r] >= 0) To[t0 + r * WAVE + lane] = src[r] != 0xffff ? -X[src[r]] : 0.0;
    for (int e = PR * WAVE + lane; e < E; e += WAVE) {             // (the pair layout's padding can push a tile past PR rounds)
      const int sx = (int)G.pv_src[t0 + e];
      To[t0 + e] = sx != 0xffff ? -X[sx] : 0.0;
    }
    wave_sync();
  }
}

// cooperative staging: plan blob -> LDS by LDS-DMA, 16-byte pieces spread over the workgroup's waves
// (the device copy of the blob is padded to a multiple of 4 words)
__device__ __forceinline__ void stage_plan(const rldl_dev_sym &S, int *wl) {
  typedef __attribute__((address_space(1))) const void *gptr_t;
  typedef __attribute__((address_space(3))) void *lptr_t;
  const int np = (S.plan_words + 3) >> 2, lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x >> 6;
  for (int base = wv * WAVE; base < np; base += blockDim.x) {
    const int i = base + lane;
    if (i < np) __builtin_amdgcn_global_load_lds((gptr_t)(S.plan + 4 * (size_t)i), (lptr_t)(wl + 4 * (size_t)base), 16, 0, 0);
  }
}
// factor row -> LDS with LDS-DMA: every 1 KiB piece is one global_load_lds_dwordx4 wave-instruction, all
// pieces are issued back to back (no VGPR staging, one vmcnt wait for the whole row).  Rows are 16-byte
// aligned (ldF even, hipMalloc base), the LDS destination of lane t is base + 16 t.
__device__ __forceinline__ void stage_factor_dma(const rldl_dev_sym &S, const double *Fg, double *Sv, int lane) {
  const int n2 = S.ldF >> 1;                                   // number of 16-byte pieces in the row
  typedef __attribute__((address_space(1))) const void *gptr_t;
  typedef __attribute__((address_space(3))) void *lptr_t;
  for (int base = 0; base < n2; base += WAVE) {
    const int i = base + lane;
    if (i < n2)
      __builtin_amdgcn_global_load_lds((gptr_t)(Fg + 2 * (size_t)i), (lptr_t)(Sv + 2 * (size_t)base), 16, 0, 0);
  }
}
__device__ __forceinline__ void wait_dma() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// STAGE: the factor row is copied to LDS (small patterns); otherwise the sweeps and gathers read it straight from
// global memory / L2 (their addresses do not depend on x, so the batched reads still pipeline) and only the
// plan and x live in LDS -- the variant for patterns whose row would leave one wave per CU.
// BLK > 0: block-tridiagonal pattern solved by stage_tri_solve<BLK> (no plan blob; LDS per wave = x + one tile)
template <bool STAGE, int BLK = 0, bool PROD = false>
__global__ __launch_bounds__(PROD ? 256 : 1024, PROD ? 4 : 1) void k_plan_solve(rldl_dev_sym S, rldl_dev_num Nn, double *__restrict__ b_all, int per_wave) {
  extern __shared__ __attribute__((aligned(16))) double sh[];
  const int lane = threadIdx.x & (WAVE - 1), wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), wpb = blockDim.x >> 6;
  const int inst = blockIdx.x * wpb + wv;
  int *wl = reinterpret_cast<int *>(sh + (size_t)wpb * per_wave);
  double *Sl = sh + (size_t)wv * per_wave;                     // STAGE: [ldF] factor + Dinv (16-byte aligned), then xs
  double *xs = STAGE ? Sl + S.ldF : Sl;                        // [N]
  const bool live = inst < Nn.batch;
  const double *Sv = STAGE ? Sl : Nn.F + (size_t)(live ? inst : 0) * S.ldF;
  double *b = b_all + (size_t)(live ? inst : 0) * S.N;
  if (!BLK) stage_plan(S, wl);
  if (live) {
    if (STAGE) stage_factor_dma(S, Nn.F + (size_t)inst * S.ldF, Sl, lane);
    const int *permg = S.plan + S.po_perm;                     // global copy: lets the rhs gather start before the barrier
    for (int j0 = 0; j0 < S.N; j0 += 4 * WAVE) {               // permute_x  qdldl_interface.c:538-541
      double v[4];
      int o[4];
      // clamped indices instead of conditional loads: the four index loads, then the four value loads, are in flight together
#pragma unroll
      for (int t = 0; t < 4; t++) o[t] = permg[min(j0 + t * WAVE + lane, S.N - 1)];
#pragma unroll
      for (int t = 0; t < 4; t++) v[t] = b[o[t]];
#pragma unroll
      for (int t = 0; t < 4; t++) { const int j = j0 + t * WAVE + lane; if (j < S.N) xs[j] = v[t]; }
    }
  }
  wait_dma();
  __syncthreads();
  if (!live) return;
  const int *perm = BLK ? S.plan + S.po_perm : wl + S.po_perm;
  if constexpr (PROD) { if (S.stage.pv_mode == 2) stage_prod_solve2(S, Sv, Nn.Ti + (size_t)inst * S.stage.pv_ldTi, xs, lane); else stage_prod_solve(S, Sv, Nn.Ti + (size_t)inst * S.stage.pv_ldTi, xs, lane); }
  else if constexpr (BLK > 0) stage_tri_solve<BLK>(S, Sv, xs, xs + ((S.N + 1) & ~1), lane);
  else plan_tri_solve(S, wl, Sv, xs, lane);
  if (S.polish) {
    for (int j = lane; j < S.N; j += WAVE) b[perm[j]] = xs[j];  // permutet_x :544-547, raw solution :563-565
  } else {
    const double *ri = Nn.rho_inv + (size_t)inst * S.m;
    const double *rsafe = S.m > 0 ? ri : b;                     // m == 0: no row is a constraint, the load just needs a valid address
    for (int j0 = 0; j0 < S.N; j0 += 4 * WAVE) {
      double bo[4], rr[4];
      int oo[4];
#pragma unroll
      for (int t = 0; t < 4; t++) {
        const int j = j0 + t * WAVE + lane;
        const int o = perm[min(j, S.N - 1)];
        oo[t] = j < S.N ? o : -1;
        const bool con = o >= S.n;                              // (unconditional loads from clamped addresses, selected afterwards)
        bo[t] = b[o];
        rr[t] = rsafe[con ? o - S.n : 0];
      }
#pragma unroll
      for (int t = 0; t < 4; t++) {
        const int j = j0 + t * WAVE + lane;
        if (oo[t] < 0) continue;
        if (oo[t] < S.n) b[oo[t]] = xs[j];                      // x_tilde :572-574
        else b[oo[t]] = bo[t] + rr[t] * xs[j];                  // z_tilde :577-579
      }
    }
  }
}

// One fused ADMM iteration (auxil.c:164-228).  All global loads a wave needs (factor row via LDS-DMA,
// x/q or z/y/rho_inv/l/u/rho per owned position) are issued before the first wait, so a wave pays about
// two memory latencies per iteration instead of one per loop trip.
template <int TMAX, bool STAGE>   // positions per lane held in registers: covers N <= 64*TMAX
__global__ __launch_bounds__(1024) void k_plan_admm(rldl_dev_sym S, rldl_dev_num Nn, rldl_dev_admm W, int per_wave) {
  extern __shared__ __attribute__((aligned(16))) double sh[];
  const int lane = threadIdx.x & (WAVE - 1), wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), wpb = blockDim.x >> 6;
  const int inst = blockIdx.x * wpb + wv;
  int *wl = reinterpret_cast<int *>(sh + (size_t)wpb * per_wave);
  double *Sl = sh + (size_t)wv * per_wave;
  double *xs = STAGE ? Sl + S.ldF : Sl;
  const int st = inst < Nn.batch ? W.status[inst] : 0;          // latency overlaps with the plan / perm loads below
  stage_plan(S, wl);
  const int *permg = S.plan + S.po_perm;
  int oo[TMAX];
#pragma unroll
  for (int t = 0; t < TMAX; t++) { const int j = t * WAVE + lane; oo[t] = j < S.N ? permg[j] : -1; }
  const bool live = inst < Nn.batch && st == ST_UNSOLVED;
  const int n = S.n, m = S.m;
  const size_t io = (size_t)(live ? inst : 0);
  const double *ri = Nn.rho_inv + io * m;
  double *x = W.x + io * n, *z = W.z + io * m, *y = W.y + io * m;
  const double *q = W.q + io * n, *l = W.l + io * m, *u = W.u + io * m, *rv = W.rho_vec + io * m;
  // per owned permuted position: o = perm[j]; variables carry (x_prev, q), constraints (z_prev, y, rho_inv, l, u, rho)
  double va[TMAX], vb[TMAX], vr[TMAX], vl[TMAX], vu[TMAX], vrho[TMAX];
  const double *Sv = STAGE ? Sl : Nn.F + io * S.ldF;
  if (live) {
    if (STAGE) stage_factor_dma(S, Nn.F + io * S.ldF, Sl, lane);
#pragma unroll
    for (int t = 0; t < TMAX; t++) {
      // branch-free: every lane loads from a valid address chosen by pointer select, so all loads of all
      // positions are in flight together (a conditional load would split the block and wait per branch)
      const int o = oo[t];
      const bool con = o >= n;
      const int iv = con || o < 0 ? 0 : o, ic = con ? o - n : 0;
      const double *pa = con ? z + ic : x + iv;
      const double *pb = con ? y + ic : q + iv;
      va[t] = *pa;
      vb[t] = *pb;
      vr[t] = ri[ic];
      vl[t] = l[ic];
      vu[t] = u[ic];
      vrho[t] = rv[ic];
    }
    // compute_rhs (auxil.c:164-178) straight into permuted order (permute_x)
#pragma unroll
    for (int t = 0; t < TMAX; t++) {
      const int j = t * WAVE + lane;
      if (oo[t] >= 0) xs[j] = oo[t] < n ? W.sigma * va[t] - vb[t] : va[t] - vr[t] * vb[t];
    }
  }
  wait_dma();
  __syncthreads();
  if (!live) return;
  plan_tri_solve(S, wl, Sv, xs, lane);
  const double alpha = W.alpha;
  double *dx = W.delta_x + io * n, *dy = W.delta_y + io * m;
#pragma unroll
  for (int t = 0; t < TMAX; t++) {
    const int o = oo[t];
    if (o < 0) continue;
    const double s = xs[t * WAVE + lane];
    if (o < n) {
      const double xp = va[t];
      const double xn = alpha * s + (1.0 - alpha) * xp;       // update_x :188-201
      x[o] = xn;
      if (W.write_delta) dx[o] = xn - xp;
    } else {
      const int i = o - n;
      const double zp = va[t], yi = vb[t], r = vr[t];
      const double zt = (zp - r * yi) + r * s;                 // z_tilde, qdldl_interface.c:577-579
      const double mix = alpha * zt + (1.0 - alpha) * zp;
      double zn = mix + r * yi;                                // update_z :203-215
      zn = fmin(fmax(zn, vl[t]), vu[t]);                       // project, proj.c:4-14
      const double d = vrho[t] * (mix - zn);                   // update_y :217-228
      z[i] = zn;
      if (W.write_delta) dy[i] = d;
      y[i] = yi + d;
    }
  }
}


// Large-N variant of the fused iteration: the per-position vectors do not fit in registers, so the right-hand side is
// built and the x/z/y update applied in batches of four positions per lane (two extra memory latencies per batch).
template <bool STAGE, int BLK = 0, bool PROD = false>
__global__ __launch_bounds__(PROD ? 256 : 1024, PROD ? 4 : 1) void k_plan_admm_loop(rldl_dev_sym S, rldl_dev_num Nn, rldl_dev_admm W, int per_wave, int iters) {
  extern __shared__ __attribute__((aligned(16))) double sh[];
  const int lane = threadIdx.x & (WAVE - 1), wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), wpb = blockDim.x >> 6;
  const int inst = blockIdx.x * wpb + wv;
  int *wl = reinterpret_cast<int *>(sh + (size_t)wpb * per_wave);
  double *Sl = sh + (size_t)wv * per_wave;
  double *xs = STAGE ? Sl + S.ldF : Sl;
  const int st = inst < Nn.batch ? W.status[inst] : 0;
  if (!BLK) stage_plan(S, wl);
  const bool live = inst < Nn.batch && st == ST_UNSOLVED;
  const int n = S.n, m = S.m;
  const size_t io = (size_t)(live ? inst : 0);
  const double *ri = Nn.rho_inv + io * m;
  double *x = W.x + io * n, *z = W.z + io * m, *y = W.y + io * m;
  const double *q = W.q + io * n, *l = W.l + io * m, *u = W.u + io * m, *rv = W.rho_vec + io * m;
  const double *Sv = STAGE ? Sl : Nn.F + io * S.ldF;
  const int *permg = S.plan + S.po_perm;
  if (live) {
    if (STAGE) stage_factor_dma(S, Nn.F + io * S.ldF, Sl, lane);
    for (int j0 = 0; j0 < S.N; j0 += 4 * WAVE) {               // compute_rhs (auxil.c:164-178) in permuted order
      int oo[4];
      double va[4], vb[4], vr[4];
#pragma unroll
      for (int t = 0; t < 4; t++) { const int j = j0 + t * WAVE + lane; oo[t] = j < S.N ? permg[j] : -1; }
#pragma unroll
      for (int t = 0; t < 4; t++) {
        const int o = oo[t];
        const bool con = o >= n;
        const int iv = con || o < 0 ? 0 : o, ic = con ? o - n : 0;
        const double *pa = con ? z + ic : x + iv;
        const double *pb = con ? y + ic : q + iv;
        va[t] = *pa; vb[t] = *pb; vr[t] = ri[ic];
      }
#pragma unroll
      for (int t = 0; t < 4; t++) {
        const int j = j0 + t * WAVE + lane;
        if (oo[t] >= 0) xs[j] = oo[t] < n ? W.sigma * va[t] - vb[t] : va[t] - vr[t] * vb[t];
      }
    }
  }
  wait_dma();
  __syncthreads();
  if (!live) return;
  const double alpha = W.alpha;
  double *dx = W.delta_x + io * n, *dy = W.delta_y + io * m;
  const int *perm = BLK ? permg : wl + S.po_perm;
  // `iters` iterations per launch (the product tri-solve of stage patterns: the launcher passes the whole group; else 1): the
  // update of an iteration leaves the NEXT right-hand side in xs, so only the first one is built from global memory above and
  // x / z / y / rho_inv are read once per iteration instead of twice.  Same arithmetic on the same values as one launch per
  // iteration: bit-identical iterates.
  for (int it = 0; it < iters; it++) {
    const bool last = it + 1 == iters;
    if constexpr (PROD) { if (S.stage.pv_mode == 2) stage_prod_solve2(S, Sv, Nn.Ti + io * S.stage.pv_ldTi, xs, lane); else stage_prod_solve(S, Sv, Nn.Ti + io * S.stage.pv_ldTi, xs, lane); }
    else if constexpr (BLK > 0) stage_tri_solve<BLK>(S, Sv, xs, xs + ((S.N + 1) & ~1), lane);
    else plan_tri_solve(S, wl, Sv, xs, lane);
    for (int j0 = 0; j0 < S.N; j0 += 4 * WAVE) {
      int oo[4];
      double va[4], vb[4], vr[4], vl[4], vu[4], vrho[4];
#pragma unroll
      for (int t = 0; t < 4; t++) {
        const int j = j0 + t * WAVE + lane;
        const int o = j < S.N ? perm[j] : -1;
        oo[t] = o;
        const bool con = o >= n;
        const int iv = con || o < 0 ? 0 : o, ic = con ? o - n : 0;
        const double *pa = con ? z + ic : x + iv;
        const double *pb = con ? y + ic : q + iv;                  // (q: only the next right-hand side needs it)
        va[t] = *pa; vb[t] = *pb; vr[t] = ri[ic]; vl[t] = l[ic]; vu[t] = u[ic]; vrho[t] = rv[ic];
      }
#pragma unroll
      for (int t = 0; t < 4; t++) {
        const int o = oo[t], j = j0 + t * WAVE + lane;
        if (o < 0) continue;
        const double s = xs[j];
        if (o < n) {
          const double xp = va[t];
          const double xn = alpha * s + (1.0 - alpha) * xp;     // update_x :188-201
          x[o] = xn;
          if (last && W.write_delta) dx[o] = xn - xp;
          if (!last) xs[j] = W.sigma * xn - vb[t];                // compute_rhs of the next iteration (auxil.c:164-178)
        } else {
          const int i = o - n;
          const double zp = va[t], yi = vb[t], r = vr[t];
          const double zt = (zp - r * yi) + r * s;               // z_tilde, qdldl_interface.c:577-579
          const double mix = alpha * zt + (1.0 - alpha) * zp;
          double zn = mix + r * yi;                              // update_z :203-215
          zn = fmin(fmax(zn, vl[t]), vu[t]);                     // project, proj.c:4-14
          const double d = vrho[t] * (mix - zn);                 // update_y :217-228
          const double yn = yi + d;
          z[i] = zn;
          if (last && W.write_delta) dy[i] = d;
          y[i] = yn;
          if (!last) xs[j] = zn - r * yn;
        }
      }
    }
    if (!last) wave_sync();
  }
}

// ================================================================================================
// k_stage_factor -- numeric LDL' of a block-tridiagonal (MPC stage-interleaved) KKT matrix by dense stage blocks,
// the arithmetic of the reference's recursion (src/recursive_ldl.c: pivot_even :686-808, pivot_odd :554-680,
// pivot_final :813-935, driver LDL_factorize_recursive :1139-1318) on one wavefront per instance:
//   S_b = K_bb - L(b,b-1) D_{b-1} L(b,b-1)'     Schur complement of the previous block (dense, LDS)
//   S_b = L_bb D_b L_bb'                          dense LDL' in place (right-looking, column by column)
//   L(b+1,b) = K(b+1,b) L_bb^-T D_b^-1            one row per lane
// Where the reference works on the positive form of the odd pivots and flips signs afterwards (:616, :651-655,
// :673-675), the blocks here are the quasi-definite ones themselves -- same L, same D.  Inputs come from the
// permuted KKT values, outputs go straight into the factor's slot layout through host-built maps, so the solve
// kernels, the restart logic and export see no difference to k_factor.  first block b0 > 0: blocks < b0 keep
// their L and D (LDL_update_from_pivot, :946-1110); L(b0,b0-1) and D_{b0-1} are read back from the factor.
// ================================================================================================
__global__ __launch_bounds__(WAVE) void k_stage_factor(rldl_dev_sym S, rldl_dev_num Nn, const int *__restrict__ mask, int b0,
                                                       const int *__restrict__ b0v) {
  const int inst = blockIdx.x, lane = threadIdx.x;
  if (mask && !mask[inst]) return;
  if (b0v) b0 = b0v[inst];                                       // per-instance restart block (horizon change)
  const rldl_dev_stage &G = S.stage;
  const int ld = G.ld, nb = G.nb_act > 0 ? G.nb_act : G.nb;     // (single-store horizon handles: the live blocks)
  extern __shared__ double sh[];
  // Wp: panel [S_b ; C_b] (rows of block b, then rows of block b+1), 2*smax rows x ld;  Lc: L(b, b-1), smax rows x ld
  double *Wp = sh, *Lc = Wp + 2 * G.smax * ld, *dg = Lc + G.smax * ld, *dc = dg + ld;
  const double *Kx = Nn.Kx + (size_t)inst * S.nnzK;
  double *F = Nn.F + (size_t)inst * S.ldF, *Dv = Nn.D + (size_t)inst * S.N;
  const int rl = lane >> 2, cl = lane & 3;                        // 16 row lanes x 4 column lanes for the rank-1 updates
  int npos = 0, zero = 0;
  for (int j = lane; j < G.bs[b0]; j += WAVE) npos += Dv[j] > 0.0 ? 1 : 0;        // pivots kept from the blocks before b0
  if (b0 > 0) {
    const int sprev = G.bs[b0] - G.bs[b0 - 1], scur = G.bs[b0 + 1] - G.bs[b0];
    for (int p = lane; p < scur * ld; p += WAVE) Lc[p] = 0.0;
    wave_sync();
    for (int e = G.lc_ptr[b0 - 1] + lane; e < G.lc_ptr[b0]; e += WAVE) Lc[G.lc_pos[e]] = F[G.lc_slot[e]];
    for (int c = lane; c < sprev; c += WAVE) dg[c] = Dv[G.bs[b0 - 1] + c];
    wave_sync();
  }
  for (int b = b0; b < nb; b++) {
    const int bs = G.bs[b], s = G.bs[b + 1] - bs, sp = b > 0 ? bs - G.bs[b - 1] : 0;
    const int sn = b + 1 < nb ? G.bs[b + 2] - G.bs[b + 1] : 0, R = s + sn;
    // 1. dense copy of the diagonal block (lower part incl. the diagonal) and of the coupling block below it
    for (int p = lane; p < R * ld; p += WAVE) Wp[p] = 0.0;
    wave_sync();
    for (int e = G.kd_ptr[b] + lane; e < G.kd_ptr[b + 1]; e += WAVE) Wp[G.kd_pos[e]] = Kx[G.kd_src[e]];
    if (sn)
      for (int e = G.kc_ptr[b] + lane; e < G.kc_ptr[b + 1]; e += WAVE) Wp[s * ld + G.kc_pos[e]] = Kx[G.kc_src[e]];
    wave_sync();
    // 2. Schur complement of the previous block, 1 x 4 tiles on and below the diagonal (8 tile columns per row)
    if (b > 0) {
      for (int tile = lane; tile < s * 8; tile += WAVE) {
        const int i = tile >> 3, j0 = (tile & 7) * 4;
        if (j0 > i) continue;
        const double *ri = Lc + i * ld, *r0 = Lc + j0 * ld;
        const double *r1 = Lc + (j0 + 1 < s ? j0 + 1 : j0) * ld, *r2 = Lc + (j0 + 2 < s ? j0 + 2 : j0) * ld, *r3 = Lc + (j0 + 3 < s ? j0 + 3 : j0) * ld;
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll 4
        for (int c = 0; c < sp; c++) {
          const double w = ri[c] * dg[c];
          a0 = fma(w, r0[c], a0); a1 = fma(w, r1[c], a1); a2 = fma(w, r2[c], a2); a3 = fma(w, r3[c], a3);
        }
        double *o = Wp + i * ld + j0;
        o[0] -= a0;
        if (j0 + 1 <= i) o[1] -= a1;
        if (j0 + 2 <= i) o[2] -= a2;
        if (j0 + 3 <= i) o[3] -= a3;
      }
      wave_sync();
    }
    // 3. right-looking elimination of the panel: L_bb, D_b and L(b+1, b) = C L_bb^-T D_b^-1 in one sweep.
    //    Column j: W[i][k] -= W[i][j] W[k][j] / d for j < k < s, k <= i < R (rows of C take every k), then W[i][j] /= d.
    for (int j = 0; j < s; j++) {
      const double d = Wp[j * ld + j];
      if (d == 0.0) zero = 1;                                     // QDLDL contract: zero pivot -> -1
      if (lane == 0 && d > 0.0) npos++;
      const double dinv = 1.0 / d;
      for (int i = j + 1 + rl; i < R; i += 16) {
        const double a = Wp[i * ld + j];
        const int kend = i < s ? i : s - 1;
        for (int k = j + 1 + cl; k <= kend; k += 4) Wp[i * ld + k] -= a * (Wp[k * ld + j] * dinv);
      }
      wave_sync();
      for (int i = j + 1 + lane; i < R; i += WAVE) Wp[i * ld + j] *= dinv;
      if (lane == 0) dc[j] = d;
      wave_sync();
    }
    // 4. this block's D, Dinv, L_bb and the coupling block into the factor
    for (int j = lane; j < s; j += WAVE) { const double d = dc[j]; Dv[bs + j] = d; F[S.nS + bs + j] = 1.0 / d; }
    for (int e = G.ld_ptr[b] + lane; e < G.ld_ptr[b + 1]; e += WAVE) F[G.ld_slot[e]] = Wp[G.ld_pos[e]];
    if (sn) {
      for (int e = G.lc_ptr[b] + lane; e < G.lc_ptr[b + 1]; e += WAVE) F[G.lc_slot[e]] = Wp[s * ld + G.lc_pos[e]];
      for (int p = lane; p < sn * ld; p += WAVE) Lc[p] = Wp[s * ld + p];             // keep L(b+1, b) for the next Schur complement
      double *tp = dg; dg = dc; dc = tp;
      wave_sync();
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) npos += __shfl_xor(npos, o);
  npos += G.nb_act > 0 ? G.npos_skip : 0;                       // variable positions of the blocks beyond the live ones
  if (lane == 0) { Nn.status[inst] = zero ? -1 : npos; if (Nn.fail && (zero || npos < S.n)) atomicOr(Nn.fail, 1); }
}

// ------------------------------------------------------------------------------------------------
// k_stage_factor_r -- the same block recursion with the panel [S_b ; C_b] held ROW-PER-LANE IN REGISTERS (lane r owns
// row r, SM = compile-time bound on the block width).  The rank-1 updates then cost one fma per column for the whole
// panel; what a lane needs from other rows travels as LDS broadcast reads (one address for all lanes) or v_readlane,
// so the LDS traffic of the LDS-resident version (which bounds it) drops by an order of magnitude.
//   Schur   : w[k] -= sum_c (Lc[r][c] d_c) Lc[k][c]      own Lc row in registers, Lc[k][c] broadcast from LDS
//   column j: d = w[j] of lane j (v_readlane); l_r = w[j] / d; w[k] -= w[j] l_k for k > j (l_k by v_readlane from lane k)
// Entries above the diagonal of a row pick up garbage and are never read.  1/d: v_rcp_f64 + 3 Newton steps.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double recip_nr(double d) {
  double x = __builtin_amdgcn_rcp(d);
#pragma unroll
  for (int it = 0; it < 3; it++) { const double e = fma(-d, x, 1.0); x = fma(x, e, x); }
  return x;
}

// MFMA: the Schur complement S_b -= L(b, b-1) D_{b-1} L(b, b-1)' on the matrix cores.  S_b is cut into NT x NT tiles of 16 x 16
// (NT = ceil(SM / 16)), K = SM in steps of 4: NT^2 SM / 4 v_mfma_f64_16x16x4_f64 per block with both operands read from the
// LDS copy of L(b, b-1) (one double per lane: A(i, k) = L(i, k) d_k, B(k, j) = L(j, k)); the result tiles go through LDS (the
// staging tile T is free at that point) and every lane subtracts its row.  On the MPC shape (22-wide blocks) this step is
// 3.8 x faster than the row-per-lane fma form with LDS broadcast reads, hand-over included (scripts/ubench_mfma_f64.hip:
// 6.45 -> 1.72 us per block update per wave at 8 waves per CU); the f64 MFMA rate equals the f64 vector rate on gfx950, what
// is saved are the SM^2 broadcast reads and the issue slots of SM^2 scalar fmas.  The elimination itself stays on the VALUs.
// One column step of the tail elimination (arrow_factor_body, LP): J is a template parameter so that every step is its own
// straight-line code (a 56-step loop with the chunk logic inside is beyond the unroller's size limit and w[] would live in scratch).
template <int SM, int J, int CH>
__device__ __forceinline__ void elim_step(double (&w)[SM], double *pc, int g, int lane, double &dcur, double &lcur, int &zero, int &npos) {
  if (J < g) {                                                   // uniform
    if (dcur == 0.0) zero = 1;
    if (lane == 0 && dcur > 0.0) npos++;
    const lds_d2 *cur = reinterpret_cast<const lds_d2 *>(pc + (J & 1) * 64);
    double *nxt = pc + ((J + 1) & 1) * 64;
    if (lane > J) w[J] = lcur;                                   // l_r = a_r / d_J; the update below is w[k] -= l_r a_k with a_k, UNSCALED, from the buffer
    wave_sync();                                                 // (compiler fence: the reads below stay behind the write of cur)
    const double lme = lcur;
    constexpr int k0 = (J + 1) & ~1, nch = (SM - k0 + CH - 1) / CH;
    lds_d2 cb[2][CH / 2];
#pragma unroll
    for (int q = 0; q < CH / 2; q++) if (k0 + 2 * q < SM) cb[0][q] = cur[k0 / 2 + q];
#pragma unroll
    for (int c = 0; c < nch; c++) {
#pragma unroll
      for (int q = 0; q < CH / 2; q++) if (c + 1 < nch && k0 + (c + 1) * CH + 2 * q < SM) cb[(c + 1) & 1][q] = cur[(k0 + (c + 1) * CH) / 2 + q];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int e = 0; e < CH; e++) {
        const int k = k0 + c * CH + e;
        if (k > J && k < SM) {
          w[k] = fma(-lme, cb[c & 1][e >> 1][e & 1], w[k]);
          if (k == J + 1 && J + 1 < g) {                         // uniform
            nxt[lane] = lane < g ? w[k] : 0.0;                   // the next pivot column goes out unscaled, at once: its LDS round trip runs
            dcur = readlane_f64(w[k], k);                        // beside the reciprocal of the next pivot instead of behind it
            lcur = w[k] * recip_nr(dcur);
          }
        }
      }
    }
  }
}
template <int SM, int CH, int... Js>
__device__ __forceinline__ void elim_steps(std::integer_sequence<int, Js...>, double (&w)[SM], double *pc, int g, int lane, double &dcur, double &lcur,
                                           int &zero, int &npos) {
  (elim_step<SM, Js, CH>(w, pc, g, lane, dcur, lcur, zero, npos), ...);
}
// STAGE_LP: the elimination of k_stage_factor_r with the pivot column through LDS and read-ahead (elim_step) instead of v_readlane;
// same box, B = 4096, MPC shape: 1.553 -> 1.502 ms per full factorisation (at 3 waves per SIMD; forced to 4 by launch bounds: 1.573)
#ifndef STAGE_LP
#define STAGE_LP 1
#endif
typedef double v4d __attribute__((ext_vector_type(4)));
template <int SM, bool MFMA>
__global__ __launch_bounds__(WAVE) void k_stage_factor_r(rldl_dev_sym S, rldl_dev_num Nn, const int *__restrict__ mask, int b0,
                                                         const int *__restrict__ b0v, int lt_rows) {
  const int inst = blockIdx.x, lane = threadIdx.x;
  if (mask && !mask[inst]) return;
  if (b0v) b0 = __builtin_amdgcn_readfirstlane(b0v[inst]);      // per-instance restart block (horizon change)
  const rldl_dev_stage &G = S.stage;
  const int ld = G.ld, nb = G.nb_act > 0 ? G.nb_act : G.nb;     // (single-store horizon handles: the live blocks)
  constexpr int NT = (SM + 15) / 16, SO_LD = 16 * NT + 1;
  extern __shared__ double sh[];
  // T: staging tile of the panel (2 smax rows; MFMA: also the 16 NT x SO_LD result tiles); Lt: L(b, b-1) as broadcast source /
  // MFMA operand (lt_rows = smax, or 16 NT zero-padded rows)
  double *T = sh, *Lt = T + 2 * G.smax * ld, *dprev = Lt + lt_rows * ld, *dcur = dprev + ld;
  const double *Kx = Nn.Kx + (size_t)inst * S.nnzK;
  double *F = Nn.F + (size_t)inst * S.ldF, *Dv = Nn.D + (size_t)inst * S.N;
  double w[SM];
  int npos = 0, zero = 0;
  // every inner loop below runs to SM without range checks (checks would fence the LDS reads one by one): the tiles are
  // SM + 1 wide (host: ld), zero outside the live block, so the surplus terms are exact zeros
  for (int p = lane; p < (2 * G.smax + lt_rows) * ld + 2 * ld + 130; p += WAVE) sh[p] = 0.0;
  wave_sync();
  for (int j = lane; j < G.bs[b0]; j += WAVE) npos += Dv[j] > 0.0 ? 1 : 0;        // pivots kept from the blocks before b0
  if (b0 > 0) {                                                  // L(b0, b0-1) and D_{b0-1} back from the stored factor
    const int sprev = G.bs[b0] - G.bs[b0 - 1];
    for (int e = G.lc_ptr[b0 - 1] + lane; e < G.lc_ptr[b0]; e += WAVE) Lt[G.lc_pos[e]] = F[G.lc_slot[e]];
    for (int c = lane; c < sprev; c += WAVE) dprev[c] = Dv[G.bs[b0 - 1] + c];
    wave_sync();
  }
  for (int b = b0; b < nb; b++) {
    const int bs = G.bs[b], s = G.bs[b + 1] - bs;
    const int sn = b + 1 < nb ? G.bs[b + 2] - G.bs[b + 1] : 0, R = s + sn;
    // 1. KKT values of the diagonal block and of the coupling block below it -> staging tile -> own row
    for (int p = lane; p < R * ld; p += WAVE) T[p] = 0.0;
    wave_sync();
    for (int e = G.kd_ptr[b] + lane; e < G.kd_ptr[b + 1]; e += WAVE) T[G.kd_pos[e]] = Kx[G.kd_src[e]];
    if (sn)
      for (int e = G.kc_ptr[b] + lane; e < G.kc_ptr[b + 1]; e += WAVE) T[s * ld + G.kc_pos[e]] = Kx[G.kc_src[e]];
    wave_sync();
    const double *myT = T + (lane < R ? lane : 0) * ld, *myL = Lt + (lane < s ? lane : 0) * ld;
#pragma unroll
    for (int c = 0; c < SM; c++) w[c] = myT[c];
    // 2. Schur complement of the previous block (rows of S_b only; the rows of the coupling block C_b stay as they are)
    if (b > 0) {
      const bool srow = lane < s;
      if (MFMA) {
        wave_sync();                                             // own rows are in registers: T becomes the result buffer
        v4d acc[NT * NT];
#pragma unroll
        for (int t = 0; t < NT * NT; t++) acc[t] = v4d{0.0, 0.0, 0.0, 0.0};
        double av[NT][SM / 4], bv[NT][SM / 4];
#pragma unroll
        for (int g = 0; g < NT; g++)
#pragma unroll
          for (int kk = 0; kk < SM / 4; kk++) {
            const int k = 4 * kk + (lane >> 4);
            const double v = Lt[(16 * g + (lane & 15)) * ld + k];
            bv[g][kk] = v; av[g][kk] = v * dprev[k];
          }
#pragma unroll
        for (int r = 0; r < NT; r++)
#pragma unroll
          for (int cc = 0; cc < NT; cc++)
#pragma unroll
            for (int kk = 0; kk < SM / 4; kk++)
              acc[r * NT + cc] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[r][kk], bv[cc][kk], acc[r * NT + cc], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < NT * NT; t++)                        // register q of lane l holds tile element (l / 16 + 4 q, l % 16) (measured on gfx950)
#pragma unroll
          for (int q = 0; q < 4; q++) T[(16 * (t / NT) + (lane >> 4) + 4 * q) * SO_LD + 16 * (t % NT) + (lane & 15)] = acc[t][q];
        wave_sync();
        const double *mine = T + (srow ? lane : 0) * SO_LD;
#pragma unroll
        for (int k = 0; k < SM; k++) { const double a = mine[k]; if (srow) w[k] -= a; }
      } else {
        double lcd[SM];
#pragma unroll
        for (int c = 0; c < SM; c++) lcd[c] = myL[c] * dprev[c];  // own row of L(b, b-1) times D_{b-1}
#pragma unroll
        for (int k = 0; k < SM; k++) {
          if (k < s) {                                           // uniform
            const double *lk = Lt + k * ld;
            double acc = 0.0;
#pragma unroll
            for (int c = 0; c < SM; c++) acc = fma(lcd[c], lk[c], acc);   // lk[c]: one address for the whole wave
            if (srow) w[k] -= acc;
          }
        }
      }
    }
    wave_sync();                                                 // T, Lt are free again
    // 3. right-looking elimination of the panel
    if (STAGE_LP) {                                              // pivot column through LDS with read-ahead (elim_step, as k_arrow_factor)
      double *pc = sh + (((2 * G.smax + lt_rows) * ld + 2 * ld + 1) & ~1);
      double dpv = readlane_f64(w[0], 0), lpv = w[0] * recip_nr(dpv);
      pc[lane] = lane < s ? w[0] : 0.0;                          // (unscaled column, see elim_step)
      elim_steps<SM, 8>(std::make_integer_sequence<int, SM>(), w, pc, s, lane, dpv, lpv, zero, npos);
#pragma unroll
      for (int j = 0; j < SM; j++) if (lane == j && j < s) dcur[j] = w[j];
    } else
#pragma unroll
    for (int j = 0; j < SM; j++) {
      if (j < s) {                                               // uniform
        const double d = readlane_f64(w[j], j);
        if (d == 0.0) zero = 1;                                  // QDLDL contract: zero pivot -> -1
        if (lane == 0) { if (d > 0.0) npos++; dcur[j] = d; }
        const double dinv = recip_nr(d);
        const double a = w[j];                                   // unscaled W[r][j]
        const double l = a * dinv;
        if (lane > j) w[j] = l;
#pragma unroll
        for (int k = j + 1; k < SM; k++) w[k] = fma(-a, readlane_f64(l, k), w[k]);   // pivot column broadcast by v_readlane; rows <= j only touch their dead upper part
      }
    }
    // 4. outputs: rows back to the tile, then D, Dinv, L_bb and L(b+1, b) into the factor's slots
    wave_sync();
    if (MFMA) {                                                  // the result tiles of step 2 used another row length: clear what they left
      for (int p = lane; p < 2 * G.smax * ld; p += WAVE) T[p] = 0.0;
      wave_sync();
    }
    if (lane < R) {
      double *o = T + lane * ld;
#pragma unroll
      for (int c = 0; c < SM; c++) o[c] = c < s ? w[c] : 0.0;   // (keeps the tile zero outside the block)
    }
    wave_sync();
    for (int j = lane; j < s; j += WAVE) { const double d = dcur[j]; Dv[bs + j] = d; F[S.nS + bs + j] = 1.0 / d; }
    for (int e = G.ld_ptr[b] + lane; e < G.ld_ptr[b + 1]; e += WAVE) F[G.ld_slot[e]] = T[G.ld_pos[e]];
    if (sn) {
      for (int e = G.lc_ptr[b] + lane; e < G.lc_ptr[b + 1]; e += WAVE) F[G.lc_slot[e]] = T[s * ld + G.lc_pos[e]];
      for (int p = lane; p < lt_rows * ld; p += WAVE) Lt[p] = p < sn * ld ? T[s * ld + p] : 0.0;   // L(b+1, b): source of the next block's Schur complement (zero padded)
      double *tp = dprev; dprev = dcur; dcur = tp;
    }
    wave_sync();
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) npos += __shfl_xor(npos, o);
  npos += G.nb_act > 0 ? G.npos_skip : 0;                       // variable positions of the blocks beyond the live ones
  if (lane == 0) { Nn.status[inst] = zero ? -1 : npos; if (Nn.fail && (zero || npos < S.n)) atomicOr(Nn.fail, 1); }
}

// ------------------------------------------------------------------------------------------------
// k_arrow_factor -- numeric LDL' for arrowhead patterns (S.arrow_dense): the head columns have entries only in the
// dense tail group, so they do not see each other:
//   head : D_c = K_cc, and all their rank-1 contributions to the tail are accumulated in ONE pass with LDS double
//          atomics (the generic kernel walks the 100 head columns of the metric shape one barrier at a time);
//   tail : the g x g Schur complement moves to registers, row per lane, and is eliminated like a stage block
//          (pivot by v_readlane, pivot column through a 64-entry LDS buffer, one fma per column for the whole block).
// Same outputs as k_factor (factor in plan slot order, D, Dinv, status); summation order of the Schur complement differs.
// ------------------------------------------------------------------------------------------------
template <int SM>
__device__ __forceinline__ void tile_invert_lds(const rldl_dev_sym &S, const rldl_dev_num &Nn, int inst, double *sh, int lane);
// INV: the tail's inverse (the tile store Ti of the solve kernels) is formed in the same launch, from the triangle while it is
// still in registers -- no second launch that reads it back from the factor row (k_tile_invert, kept for the other factor kernels)
template <int SM, bool INV, bool LP>
__device__ __forceinline__ void arrow_factor_body(const rldl_dev_sym &S, const rldl_dev_num &Nn, const int *__restrict__ mask, int inst,
                                                  long long *__restrict__ trace = nullptr) {
  const int lane = threadIdx.x;
  if (mask && !mask[inst]) return;
  // wave timeline (rldl_batch_trace_factor): 0 start, 1 KKT values in the workspace, 2 head contributions added, 3 tail in registers,
  // 4 tail eliminated, 5 factor row and D stored, 6 triangle packed, 7 tail inverse stored
  long long *tr = trace && lane == 0 ? trace + 8 * (size_t)inst : nullptr;
  if (tr) tr[0] = wall_clock64();
  extern __shared__ double sh[];                                  // W = [L values, CSC order | D] as in k_factor, then scratch
  const int nW = S.nnzL + S.N, g0 = S.arrow_g0, g = S.arrow_g;
  double *Wd = sh + S.nnzL, *dih = sh + nW;
  double *F = Nn.F + (size_t)inst * S.ldF, *Dg = Nn.D + (size_t)inst * S.N;
  const double *K = Nn.Kx + (size_t)inst * S.nnzK;
  // Every loop below that reads an index table from global memory takes FB rounds at a time: the FB index loads (and the
  // value loads that do not depend on them) are in flight together instead of one memory round trip per round.
  // (FBV / FBP rounds while the row registers are still free, FBO with the eliminated rows live)
  constexpr int FBV = 16, FBP = 16, FBO = 16;
  for (int i = lane; i < nW; i += WAVE) sh[i] = 0.0;
  wave_sync();
  for (int k0 = 0; k0 < S.nnzK; k0 += FBV * WAVE) {
    int ix[FBV];
    double v[FBV];
#pragma unroll
    for (int u = 0; u < FBV; u++) { const int k = min(k0 + u * WAVE + lane, S.nnzK - 1); ix[u] = S.KtoW[k]; v[u] = K[k]; }
#pragma unroll
    for (int u = 0; u < FBV; u++) if (k0 + u * WAVE + lane < S.nnzK) sh[ix[u]] = v[u];
  }
  wave_sync();
  if (tr) tr[1] = wall_clock64();
  int npos = 0, zero = 0;
  for (int j = lane; j < g0; j += WAVE) {                        // head pivots are final as they come
    const double d = Wd[j];
    if (d == 0.0) zero = 1;
    if (d > 0.0) npos++;
    dih[j] = 1.0 / d;
  }
  wave_sync();
  // head: every pair (a, b) of column j adds -l_a l_b d_j to a tail entry; columns are independent -> no barrier, atomics
  {                                                              // flat over all head columns: independent iterations
    // Rounds of FBP x 64 pairs; the index words of round k + 1 are fetched while round k is processed (two register sets), and a
    // round is straight-line: its 3 LDS reads per pair are independent of its atomics (they read head entries, the atomics write
    // tail entries), eight pairs at a time.  A pair index past the end is clamped to the last pair and adds 0.0 there (a predicate
    // per pair made every pair its own basic block: read, wait, multiply, add, one after the other).
    const int np = S.arrow_npairs;
    auto fetch = [&](unsigned (&ab)[FBP], unsigned (&dc)[FBP], int t0) {
#pragma unroll
      for (int u = 0; u < FBP; u++) { const int t = min(t0 + u * WAVE + lane, np - 1); ab[u] = S.arrow_pab[t]; dc[u] = S.arrow_pdc[t]; }
    };
    auto apply = [&](const unsigned (&ab)[FBP], const unsigned (&dc)[FBP], int t0) {
#pragma unroll
      for (int u0 = 0; u0 < FBP; u0 += 8) {
        double pv[8];
#pragma unroll
        for (int u = 0; u < 8; u++) pv[u] = -(sh[ab[u0 + u] & 0xffffu] * (sh[ab[u0 + u] >> 16] * dih[dc[u0 + u] >> 16]));
#pragma unroll
        for (int u = 0; u < 8; u++) unsafeAtomicAdd(&sh[dc[u0 + u] & 0xffffu], t0 + (u0 + u) * WAVE + lane < np ? pv[u] : 0.0);
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    unsigned ab0[FBP], dc0[FBP], ab1[FBP], dc1[FBP];
    if (np > 0) fetch(ab0, dc0, 0);
    for (int t0 = 0; t0 < np; t0 += 2 * FBP * WAVE) {
      const bool second = t0 + FBP * WAVE < np;                  // uniform
      if (second) fetch(ab1, dc1, t0 + FBP * WAVE);
      apply(ab0, dc0, t0);
      if (t0 + 2 * FBP * WAVE < np) fetch(ab0, dc0, t0 + 2 * FBP * WAVE);
      if (second) apply(ab1, dc1, t0 + FBP * WAVE);
    }
  }
  // (l_rc = K_rc / d_c of the head columns is applied on the way out: nothing below reads the head entries again)
  // tail: Schur complement -> registers, row per lane; positions from the [g][64] table (-1: structural zero / c >= r)
  double w[SM];
  const int r = lane < g ? lane : g - 1;
  {
    int tp[SM];
#pragma unroll
    for (int c = 0; c < SM; c++) tp[c] = c < g ? S.arrow_tpos[c * 64 + r] : -1;
    wave_sync();
    if (tr) tr[2] = wall_clock64();
#pragma unroll
    for (int c = 0; c < SM; c++) {
      const double lo = sh[tp[c] >= 0 ? tp[c] : 0];
      w[c] = tp[c] >= 0 ? lo : (c == r ? Wd[g0 + r] : 0.0);
    }
  }
  wave_sync();
  if (tr) tr[3] = wall_clock64();
  if (LP) {
  // The pivot column of step j goes through a 64-entry LDS buffer and every lane reads the entries it multiplies with as broadcast
  // reads, two per instruction (2 v_readlane + hazard nop + fma per update was 3.5 issue slots; this is 1.5).  The kernel runs at
  // 2 waves per SIMD and is bound by latency, not issue slots (timeline: rldl_batch_trace_factor), so
  //   * look-ahead: column j + 1 is final after its first update, its scaled copy is written (other buffer) before the rest of
  //     step j's updates -- the LDS round trip of the next step hides behind them;
  //   * the reads of a step are issued a chunk of LCH entries ahead of the fmas that consume them (the compiler's own schedule
  //     keeps 2-3 reads in flight and every pair of fmas waits a full LDS latency).
  double *pc = sh + ((nW + g0 + 1) & ~1);
  double dcur = readlane_f64(w[0], 0), lcur = w[0] * recip_nr(dcur);
  pc[lane] = lane < g ? w[0] : 0.0;                              // (unscaled column, see elim_step)
  elim_steps<SM, LCH>(std::make_integer_sequence<int, SM>(), w, pc, g, lane, dcur, lcur, zero, npos);
  }
  if (tr) tr[4] = wall_clock64();
  // back to the CSC workspace, then the common coalesced write-out in plan slot order (the positions are fetched again: 56 registers
  // held across the elimination cost the second wave per SIMD)
  if (lane < g) {
    int tp[SM];
#pragma unroll
    for (int c = 0; c < SM; c++) tp[c] = c < g ? __builtin_nontemporal_load(&S.arrow_tpos[c * 64 + lane]) : -1;
#pragma unroll
    for (int c = 0; c < SM; c++) {
      if (tp[c] >= 0) sh[tp[c]] = w[c];
      else if (c == lane) Wd[g0 + lane] = w[c];
    }
  }
  wave_sync();
  // write-out in SLOT order: consecutive lanes store consecutive words of the factor row; the table names the workspace position
  // behind every slot and, for head columns, the column whose 1/d still has to be applied (padding slots are rewritten with 0.0)
  for (int s0 = 0; s0 < S.nS; s0 += FBO * WAVE) {
    unsigned ot[FBO];
#pragma unroll
    for (int u = 0; u < FBO; u++) ot[u] = S.arrow_out[min(s0 + u * WAVE + lane, S.nS - 1)];
#pragma unroll
    for (int u = 0; u < FBO; u++) {
      const int sl = s0 + u * WAVE + lane;
      const unsigned pos = ot[u] & 0xffffu, hc = ot[u] >> 16;
      double v = pos != 0xffffu ? sh[pos] : 0.0;
      if (hc != 0xffffu) v *= dih[hc];
      if (sl < S.nS) F[(unsigned)sl] = v;
    }
  }
  for (int j = lane; j < S.N; j += WAVE) { const double d = Wd[j]; Dg[j] = d; F[S.nS + j] = 1.0 / d; }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { npos += __shfl_xor(npos, o); zero |= __shfl_xor(zero, o); }
  if (lane == 0) { Nn.status[inst] = zero ? -1 : npos; if (Nn.fail && (zero || npos < S.n)) atomicOr(Nn.fail, 1); }
  if (tr) tr[5] = wall_clock64();
  if (INV) {
    wave_sync();                                                 // the workspace is dead: its first words become the packed triangle
    if (lane < g) {
      double *row = sh + ((lane * (lane - 1)) >> 1);             // row-major packed, as the factor row holds it
#pragma unroll
      for (int c = 0; c + 1 < SM; c++)
        if (c < lane) row[c] = w[c];
    }
    wave_sync();
    if (tr) tr[6] = wall_clock64();
    tile_invert_lds<SM>(S, Nn, inst, sh, lane);
    if (tr) tr[7] = wall_clock64();
  }
}
template <int SM, bool INV, bool LP>
__global__ __launch_bounds__(WAVE) void k_arrow_factor(rldl_dev_sym S, rldl_dev_num Nn, const int *__restrict__ mask, long long *__restrict__ trace) {
  arrow_factor_body<SM, INV, LP>(S, Nn, mask, blockIdx.x, trace);
}
template <int SM>
__global__ __launch_bounds__(WAVE) void k_arrow_factor_multi(rldl_dev_multi M, int masked) {   // masked: only the instances whose W.refactor is set (rho adaptation)
  const int g = multi_group(M.first_inst, M.ngroups, blockIdx.x), inst = (int)blockIdx.x - M.first_inst[g];
  int *mask = masked ? M.W[g].refactor : nullptr;
  arrow_factor_body<SM, true, true>(M.S[g], M.N[g], mask, inst);
  if (mask && threadIdx.x == 0) mask[inst] = 0;                  // (last reader of the mask of this rho update)
}

// ================================================================================================
// Arrowhead specialisation (plan->arrow_ok): all out-of-group entries feed ONE dense group (the Schur
// tail of a KKT matrix ordered by minimum degree) and no other group has a triangle.  Then
//   * the coupling values (jagged-diagonal order, one row per lane) are staged by LDS-DMA, moved to
//     REGISTERS, and the same LDS bytes are overwritten by the second DMA phase (the packed triangle);
//     their column indices come from a padded [step][64] table;
//   * forward: the tail rows gather from the (already final) head entries out of registers, then the
//     packed triangle is swept forward and -- without leaving registers -- scaled by Dinv and swept back;
//   * the last `rr` rows of the triangle and Dinv never touch LDS: lane a holds L(i, a) of such a row, so
//     the forward step of row i is a wave reduction and its backward step a plain fma;
//   * backward for the head columns is the transposed gather: every tail lane scatters L(r,c) x_r into
//     x_c with LDS double atomics (ds_add_f64) from the same registers.
// LDS per wave is the triangle without its last rr rows + x (10 224 B at the metric shape with rr = 2, which
// is what lets 16 waves = 16 instances share a CU: the whole 4096-instance batch is resident at once), there
// is no shared plan copy and therefore no workgroup barrier.
// ================================================================================================
#define ARROW_RR_MAX 2
#define SA 4   // sweep depth of the arrow kernels (register budget: 128 VGPRs for 4 waves per SIMD)

template <int TG>
struct ArrowRegs {
  double v[TG];            // coupling values of this lane's virtual row (piece of a row), step-major; 0 where there is none
};
template <int TG>
struct ArrowIdx {
  unsigned ix[(TG + 1) / 2];  // packed column indices: step 2k in the low half, 2k+1 in the high half (padding: column 0)
};
struct ArrowDiag {
  double dtail;            // Dinv[g0 + lane]
  double row[ARROW_RR_MAX];   // L(g - rr + k, lane) for lane < g - rr + k, else 0
};

// The index table is shared by all instances (L2-resident): it is re-read by the gather and by the scatter of every
// iteration instead of living in 12 VGPRs across the sweeps.  `ap` is made opaque so the loads stay where they are.
template <int TG>
__device__ __forceinline__ void arrow_load_idx(const rldl_dev_sym &S, int lane, ArrowIdx<TG> &I) {
  int off = S.po_avcol;                                          // (opaque offset, not pointer: the loads stay global_load)
  asm volatile("" : "+s"(off));
  const unsigned *ap = reinterpret_cast<const unsigned *>(S.plan + off);
#pragma unroll
  for (int t2 = 0; t2 < (TG + 1) / 2; t2++) I.ix[t2] = ap[t2 * 64 + lane];   // (table padded to TG steps by the host)
}
// Dinv of the tail and the register-resident triangle rows: straight from the factor row in HBM (coalesced)
__device__ __forceinline__ void arrow_load_diag(const rldl_dev_sym &S, const double *Fg, int g0, int g, int rr, int lane, ArrowDiag &Dg) {
  const double *Dinv = Fg + S.nS;
  Dg.dtail = Dinv[g0 + (lane < g ? lane : 0)];
#pragma unroll
  for (int k = 0; k < ARROW_RR_MAX; k++) {
    const int i = g - rr + k;                                    // local row index
    double v = 0.0;
    if (k < rr) {                                                // uniform
      v = Fg[S.nOp + ((i * (i - 1)) >> 1) + (lane < i ? lane : 0)];
      v = lane < i ? v : 0.0;
    }
    Dg.row[k] = v;
  }
}
// coupling values: LDS (slots [0, nOp) staged by LDS-DMA at the start of the wave's buffer) -> registers, through the
// virtual-row slot table (vm: this lane's packed slots, fetched before the DMA wait)
template <int TG>
__device__ __forceinline__ void arrow_load_map(const rldl_dev_sym &S, int lane, ArrowIdx<TG> &M) {
  const unsigned *vm = reinterpret_cast<const unsigned *>(S.plan + S.po_avmap);
#pragma unroll
  for (int t2 = 0; t2 < (TG + 1) / 2; t2++) M.ix[t2] = vm[t2 * 64 + lane];   // (table padded to TG steps by the host)
}
template <int TG>
__device__ __forceinline__ void arrow_load_val(const double *Ov, const ArrowIdx<TG> &M, ArrowRegs<TG> &R) {
#pragma unroll
  for (int t = 0; t < TG; t++) {
    const unsigned slot = (t & 1) ? M.ix[t >> 1] >> 16 : M.ix[t >> 1] & 0xffffu;
    const double val = Ov[slot != 0xffffu ? slot : 0u];
    R.v[t] = slot != 0xffffu ? val : 0.0;
  }
}

// Tv: LDS triangle rows [0, g - rr); xs: LDS permuted rhs in / solution out.  Part 1: forward gather, both sweeps;
// leaves the solution of the tail group in xs and ZERO in every head slot.  Part 2 (arrow_scatter) accumulates
// -sum_r L(r,c) x_r into the head slots; the caller adds y_c Dinv_c (it still holds y_c in registers).
template <int TG>
__device__ __forceinline__ void arrow_tri_solve(const rldl_dev_sym &S, const ArrowRegs<TG> &R, const ArrowDiag &Dg, const double *Tv,
                                                double *xs, int g0, int g, int rr, int jr, int lane, long long *tr = nullptr) {
  const bool act = lane < g;
  const int gp = g - rr;                                         // rows whose L entries live in LDS
  // ---- forward gather out of registers: every lane sums its piece of a tail row, the pieces of a row meet in its
  //      x slot (which holds b) through an LDS atomic add ----
  double ga = 0.0;
  {
    ArrowIdx<TG> I;
    arrow_load_idx<TG>(S, lane, I);
#pragma unroll
    for (int t = 0; t < TG; t++) {                               // steps without an entry carry value 0 and column 0
      const unsigned col = (t & 1) ? I.ix[t >> 1] >> 16 : I.ix[t >> 1] & 0xffffu;
      ga = fma(-R.v[t], xs[col], ga);
    }
  }
  wave_sync();                                                   // all reads of the head values are done
  if (lane < S.arrow_vrows) unsafeAtomicAdd(&xs[jr], ga);
  wait_dma();                                                    // triangle (second DMA phase) streamed in behind the gather
  wave_sync();
  if (tr && lane == 0) tr[2] = wall_clock64();                   // gather done (first iteration: triangle has arrived)
  double acc = act ? xs[g0 + lane] : 0.0;
  for (int j = lane; j < S.N; j += WAVE)                         // head slots become scatter accumulators
    if (j < g0 || j >= g0 + g) xs[j] = 0.0;
  const bool tri = S.arrow_tb >= 0 && g > 1;
  if (tri) {
    if (gp > 1) acc = sweep_fwd<SA>(Tv, gp, lane, acc);
    if (rr > 0) {                                                // rows gp.. : y_i = b_i - sum_a L(i,a) y_a as wave reductions
      const double y = lane < gp ? acc : 0.0;
      const double s0 = wave_sum(Dg.row[0] * y);
      if (lane == gp) acc -= s0;
      if (rr > 1) {
        const double s1 = wave_sum(Dg.row[1] * y);
        const double l10 = readlane_f64(Dg.row[1], gp), y0 = readlane_f64(acc, gp);
        if (lane == gp + 1) acc = (acc - s1) - l10 * y0;
      }
    }
  }
  if (tr && lane == 0) tr[3] = wall_clock64();                   // forward sweep + register rows done
  if (act) acc *= Dg.dtail;                                      // D^-1 without leaving registers
  if (tri) {
#pragma unroll
    for (int k = ARROW_RR_MAX - 1; k >= 0; k--)
      if (k < rr) {                                              // x_j -= L(i, j) x_i, i = gp + k; row[k] is 0 on lanes >= i
        const double xi = readlane_f64(acc, gp + k);
        acc = fma(-Dg.row[k], xi, acc);
      }
    if (gp > 1) acc = sweep_bwd<SA>(Tv, gp, lane, acc);
  }
  if (act) xs[g0 + lane] = acc;
  if (tr && lane == 0) tr[4] = wall_clock64();                   // backward sweep done
  wave_sync();
}
// ---- transposed gather: scatter L(r, c) x_r into the head columns with LDS double atomics ----
template <int TG>
__device__ __forceinline__ void arrow_scatter(const rldl_dev_sym &S, const ArrowRegs<TG> &R, double *xs, int g, int jr, int lane) {
  const double xr = lane < S.arrow_vrows ? xs[jr] : 0.0;
  ArrowIdx<TG> I;
  arrow_load_idx<TG>(S, lane, I);
#pragma unroll
  for (int t = 0; t < TG; t++) {
    const unsigned col = (t & 1) ? I.ix[t >> 1] >> 16 : I.ix[t >> 1] & 0xffffu;
    const double pr = R.v[t] * xr;
    if (R.v[t] != 0.0) unsafeAtomicAdd(&xs[col], -pr);           // padding steps and lanes beyond a step's count hold 0
  }
  wave_sync();
}

// LDS-DMA of one part of the factor row into the wave's staging buffer (16-byte pieces; both parts start at an
// even slot).  first != 0: coupling values, slots [0, nOp); else the triangle rows kept in LDS, `cnt2` pieces.
__device__ __forceinline__ void arrow_stage(const rldl_dev_sym &S, const double *Fg, double *Tv, int lane, int first, int cnt2) {
  typedef __attribute__((address_space(1))) const void *gptr_t;
  typedef __attribute__((address_space(3))) void *lptr_t;
  const int n2 = first ? S.nOp >> 1 : cnt2;
  const double *src = first ? Fg : Fg + S.nOp;
  for (int base = 0; base < n2; base += WAVE) {
    const int i = base + lane;
    if (i < n2) __builtin_amdgcn_global_load_lds((gptr_t)(src + 2 * (size_t)i), (lptr_t)(Tv + 2 * (size_t)base), 16, 0, 0);
  }
}

// doubles per wave, offset of x, 16-B pieces of the LDS triangle, register rows
struct ArrowGeom { int per_wave, xoff, tri2, rr; };

template <int TMAX, int TG>
__global__ __launch_bounds__(256, 4) void k_arrow_solve(rldl_dev_sym S, rldl_dev_num Nn, double *__restrict__ b_all, ArrowGeom G) {
  extern __shared__ __attribute__((aligned(16))) double sh[];
  const int lane = threadIdx.x & (WAVE - 1), wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), wpb = blockDim.x >> 6;
  const int inst = blockIdx.x * wpb + wv;
  if (inst >= Nn.batch) return;
  double *Tv = sh + (size_t)wv * G.per_wave;
  double *xs = Tv + G.xoff;
  const double *Fg = Nn.F + (size_t)inst * S.ldF;
  double *b = b_all + (size_t)inst * S.N;
  const int *permg = S.plan + S.po_perm;
  const int g0 = S.arrow_g0, g = S.arrow_g;
  ArrowRegs<TG> R;
  ArrowDiag Dg;
  arrow_stage(S, Fg, Tv, lane, 1, 0);                           // coupling values first ...
  const int jr = (S.plan + S.po_avrow)[lane];                   // row of this lane's piece (lanes >= arrow_vrows: unused)
  ArrowIdx<TG> M;
  arrow_load_map<TG>(S, lane, M);
  int oo[TMAX];
  double vb[TMAX];
#pragma unroll
  for (int t = 0; t < TMAX; t++) { const int j = t * WAVE + lane; oo[t] = j < S.N ? permg[j] : -1; }
#pragma unroll
  for (int t = 0; t < TMAX; t++) vb[t] = b[oo[t] >= 0 ? oo[t] : 0];   // permute_x  qdldl_interface.c:538-541
  arrow_load_diag(S, Fg, g0, g, G.rr, lane, Dg);
  wait_dma();
  arrow_load_val<TG>(Tv, M, R);                                 // ... into registers ...
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  arrow_stage(S, Fg, Tv, lane, 0, G.tri2);                      // ... then the triangle over the same LDS
#pragma unroll
  for (int t = 0; t < TMAX; t++) { const int j = t * WAVE + lane; if (oo[t] >= 0) xs[j] = vb[t]; }
  wave_sync();                                                  // (the triangle DMA is awaited inside, behind the gather)
  arrow_tri_solve<TG>(S, R, Dg, Tv, xs, g0, g, G.rr, jr, lane);
  double dv[TMAX], rr_[TMAX];                                   // head Dinv and rho_inv: fetched under the scatter
  const double *ri = Nn.rho_inv + (size_t)inst * S.m;
#pragma unroll
  for (int t = 0; t < TMAX; t++) {
    const int j = t * WAVE + lane;
    dv[t] = Fg[S.nS + (j < S.N ? j : 0)];
    rr_[t] = S.polish ? 0.0 : ri[oo[t] >= S.n ? oo[t] - S.n : 0];
  }
  arrow_scatter<TG>(S, R, xs, g, jr, lane);
#pragma unroll
  for (int t = 0; t < TMAX; t++) {
    if (oo[t] < 0) continue;
    const int j = t * WAVE + lane;
    const bool head = j < g0 || j >= g0 + g;
    const double xv = head ? fma(vb[t], dv[t], xs[j]) : xs[j];  // x_c = y_c Dinv_c - sum_r L(r,c) x_r
    if (S.polish || oo[t] < S.n) b[oo[t]] = xv;
    else b[oo[t]] = vb[t] + rr_[t] * xv;                        // qdldl_interface.c:568-579
  }
}

// `iters` ADMM iterations per launch: an instance belongs to one wave, nothing couples instances, and the whole
// factor row of the instance sits in that wave's registers + LDS -- so the wave simply keeps going.  The factor, q
// and the iterates are read from HBM once per launch instead of once per iteration; only l, u, rho and the head's
// Dinv (3.2 KB, L2-resident) are re-fetched per iteration because the register file has no room for them during
// the sweeps.  x, z, y (and delta_x / delta_y when a check follows) are stored after the last iteration.
// TRACE: instantiation with the wave-timeline stores (osqp_batch_trace_iteration); the production kernel carries none.
template <int TMAX, int TG, bool TRACE>
__global__ __launch_bounds__(256, 4) void k_arrow_admm(rldl_dev_sym S, rldl_dev_num Nn, rldl_dev_admm W, ArrowGeom G, int iters) {
  extern __shared__ __attribute__((aligned(16))) double sh[];
  const int lane = threadIdx.x & (WAVE - 1), wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), wpb = blockDim.x >> 6;
  const int inst = blockIdx.x * wpb + wv;
  if (inst >= Nn.batch) return;
  const int st = W.status[inst];                                // latency overlaps with the index loads below
  long long *tr = TRACE && W.trace ? W.trace + 8 * (size_t)inst : nullptr;
  if (TRACE && tr && lane == 0) tr[7] = wall_clock64();         // wave start; slots 0..6 belong to ONE iteration of the launch
  double *Tv = sh + (size_t)wv * G.per_wave;
  double *xs = Tv + G.xoff;
  const int *permg = S.plan + S.po_perm;
  int oo[TMAX];
#pragma unroll
  for (int t = 0; t < TMAX; t++) { const int j = t * WAVE + lane; oo[t] = j < S.N ? permg[j] : -1; }
  const int g0 = S.arrow_g0, g = S.arrow_g;
  const int jr = (S.plan + S.po_avrow)[lane];                   // row of this lane's piece (lanes >= arrow_vrows: unused)
  if (st != ST_UNSOLVED) return;
  const int n = S.n, m = S.m;
  const size_t io = (size_t)inst;
  const double *Fg = Nn.F + io * S.ldF;
  const double *ri = Nn.rho_inv + io * m;
  double *x = W.x + io * n, *z = W.z + io * m, *y = W.y + io * m;
  const double *q = W.q + io * n, *l = W.l + io * m, *u = W.u + io * m, *rv = W.rho_vec + io * m;
  ArrowRegs<TG> R;
  ArrowDiag Dg;
  arrow_stage(S, Fg, Tv, lane, 1, 0);                           // coupling values first ...
  double va[TMAX], vb[TMAX];
#pragma unroll
  for (int t = 0; t < TMAX; t++) {
    const int o = oo[t];
    const bool con = o >= n;
    const int iv = con || o < 0 ? 0 : o, ic = con ? o - n : 0;
    const double *pa = con ? z + ic : x + iv;
    const double *pb = con ? y + ic : q + iv;
    va[t] = *pa; vb[t] = *pb;
  }
  arrow_load_diag(S, Fg, g0, g, G.rr, lane, Dg);
  wait_dma();
  {
    ArrowIdx<TG> M;
    arrow_load_map<TG>(S, lane, M);
    arrow_load_val<TG>(Tv, M, R);                               // ... into registers ...
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  arrow_stage(S, Fg, Tv, lane, 0, G.tri2);                      // ... then the triangle over the same LDS
  const double alpha = W.alpha;
  auto ldg = [](const double *base, unsigned idx) {              // scalar base + 32-bit lane offset (no 64-bit lane arithmetic)
    return *reinterpret_cast<const double *>(reinterpret_cast<const char *>(base) + (size_t)(idx * 8u));
  };
#pragma clang loop unroll(disable)
  for (int it = 0; it < iters; it++) {
    const bool last = it + 1 == iters;
    const bool trit = TRACE && tr && (W.trace_iter < 0 ? last : it == W.trace_iter);
    if (trit && lane == 0) tr[0] = wall_clock64();
    // keep the per-lane index data opaque per iteration: otherwise every LDS / global address derived from it is
    // hoisted out of the loop and the 24 gather addresses alone cost 24 VGPRs of the 128 a wave may hold
    int ln = lane;                                              // (same for everything derived from the lane id)
    asm volatile("" : "+v"(ln));
#pragma unroll
    for (int t = 0; t < TMAX; t++) asm volatile("" : "+v"(oo[t]));
#pragma unroll
    for (int t = 0; t < TG; t++) asm volatile("" : "+v"(R.v[t]));
    {
      // rho_inv, l, u, rho and the head's Dinv are constant over the launch but are re-read (L2) where they are used:
      // the 128 registers of a wave cannot hold them across the sweeps next to the coupling values
      int zero = 0;
      asm volatile("" : "+s"(zero));
      const double *rip = ri + zero;
      double vr[TMAX];
#pragma unroll
      for (int t = 0; t < TMAX; t++) vr[t] = ldg(rip, oo[t] >= n ? (unsigned)(oo[t] - n) : 0u);
#pragma unroll
      for (int t = 0; t < TMAX; t++) {                          // compute_rhs (auxil.c:164-178) in permuted order
        const int j = t * WAVE + ln;
        if (oo[t] >= 0) xs[j] = oo[t] < n ? W.sigma * va[t] - vb[t] : va[t] - vr[t] * vb[t];
      }
    }
    wave_sync();                                                // (the triangle DMA is awaited inside, behind the gather)
    if (tr && ln == 0 && trit) tr[1] = wall_clock64();          // rhs in LDS
    arrow_tri_solve<TG>(S, R, Dg, Tv, xs, g0, g, G.rr, jr, ln, trit ? tr : nullptr);
    // bounds, rho and the head's Dinv are needed only from here on: fetched behind the sweeps, under the scatter
    double vl[TMAX], vu[TMAX], vrho[TMAX], dv[TMAX], vr[TMAX];
#pragma unroll
    for (int t = 0; t < TMAX; t++) {
      const unsigned ic = oo[t] >= n ? (unsigned)(oo[t] - n) : 0u, j = (unsigned)(t * WAVE + ln);
      vl[t] = ldg(l, ic); vu[t] = ldg(u, ic); vrho[t] = ldg(rv, ic); vr[t] = ldg(ri, ic);
      dv[t] = ldg(Fg + S.nS, j < (unsigned)S.N ? j : 0u);
    }
    arrow_scatter<TG>(S, R, xs, g, jr, ln);
    if (tr && ln == 0 && trit) tr[5] = wall_clock64();          // scatter done
#pragma unroll
    for (int t = 0; t < TMAX; t++) {
      const int o = oo[t], j = t * WAVE + ln;
      if (o < 0) continue;
      double sv = xs[j];
      if (j < g0 || j >= g0 + g) {                             // head: x_c = y_c Dinv_c - sum_r L(r,c) x_r, y_c = rhs_c
        const double rhs = o < n ? W.sigma * va[t] - vb[t] : va[t] - vr[t] * vb[t];
        sv = fma(rhs, dv[t], sv);
      }
      if (o < n) {
        const double xp = va[t];
        const double xn = alpha * sv + (1.0 - alpha) * xp;    // update_x :188-201
        va[t] = xn;
        if (last) {                                            // (output pointers are formed here, not kept across the loop)
          (W.x + io * n)[o] = xn;
          if (W.write_delta) (W.delta_x + io * n)[o] = xn - xp;
        }
      } else {
        const int i = o - n;
        const double zp = va[t], yi = vb[t], r = vr[t];
        const double zt = (zp - r * yi) + r * sv;              // z_tilde, qdldl_interface.c:577-579
        const double mix = alpha * zt + (1.0 - alpha) * zp;
        double zn = mix + r * yi;                              // update_z :203-215
        zn = fmin(fmax(zn, vl[t]), vu[t]);                     // project, proj.c:4-14
        const double d = vrho[t] * (mix - zn);                 // update_y :217-228
        va[t] = zn; vb[t] = yi + d;
        if (last) {
          (W.z + io * m)[i] = zn;
          if (W.write_delta) (W.delta_y + io * m)[i] = d;
          (W.y + io * m)[i] = yi + d;
        }
      }
    }
    if (trit && ln == 0 && W.trace_iter >= 0) tr[6] = wall_clock64();
  }
  if (tr && lane == 0 && W.trace_iter < 0) tr[6] = wall_clock64();
}

// ================================================================================================
// Tile kernels (S.tile_ok): the arrowhead solve WITHOUT dependent sweeps.
// The two sweeps over the tail's packed triangle are a chain of g - 1 dependent (v_readlane x2 -> v_fma_f64) steps each
// (94 of them at the metric shape: 4.5 of the 7.4 us a lone wave needs per solve).  Here the unit lower triangle L22 of
// the tail group is inverted once per factorisation (k_tile_invert) and x <- L22^-T D^-1 L22^-1 x becomes two
// dependency-free products with Linv = L22^-1:
//   * Linv's strictly lower part is cut into ta x ta tiles, ONE TILE PER LANE, ta^2 values in registers (25 at the metric
//     shape: 55 of 64 lanes hold a tile, the 10 diagonal tiles are half empty);
//   * forward  y = Linv c : the lane of tile (I, J) reads the ta entries of c in block column J, forms its ta partial row
//     sums with ta^2 independent fmas and adds them to block row I of x with ta LDS atomics (ds_add_f64);
//   * backward x = Linv' w: the same registers, read by columns: ta reads from block row I, ta^2 fmas, ta atomics into block column J;
//   * rows are stored rotated by J and columns by I inside a tile, so lanes that share a block row (column) start their
//     atomics at different words (at most ceil(tq / ta) lanes meet on one word).
// The coupling part (tail rows x head columns) is the virtual-row gather / scatter of the arrow kernels, with its values
// loaded straight from the factor row in HBM into registers (per-lane slot table), no LDS staging.  LDS per wave is x only.
// Same result as QDLDL_solve up to rounding (the products sum in a different order and use the explicit inverse):
// tests hold it to the same 1e-10 / 1e-8 tolerances as the sweep kernels.  RLDL_NO_TILE=1 selects the sweep kernels.
// ================================================================================================
template <int TA>
struct TileRegs { double v[TA * TA]; };
// tile values: global (Ti row of the instance, (k, lane) order) -> registers through the slot table
template <int TA>
__device__ __forceinline__ void tile_load(const rldl_dev_sym &S, const double *__restrict__ Tg, int lane, TileRegs<TA> &T) {
  const unsigned *tm = reinterpret_cast<const unsigned *>(S.plan + S.po_tmap);
  unsigned w[(TA * TA + 1) / 2];
#pragma unroll
  for (int k2 = 0; k2 < (TA * TA + 1) / 2; k2++) w[k2] = tm[k2 * 64 + lane];
#pragma unroll
  for (int k = 0; k < TA * TA; k++) {
    const unsigned slot = (k & 1) ? w[k >> 1] >> 16 : w[k >> 1] & 0xffffu;
    const double val = Tg[slot != 0xffffu ? slot : 0u];
    T.v[k] = slot != 0xffffu ? val : 0.0;
  }
}
// coupling values: factor row in global memory -> registers through the virtual-row slot table
template <int TG>
__device__ __forceinline__ void arrow_load_val_global(const rldl_dev_sym &S, const double *__restrict__ Fg, int lane, ArrowRegs<TG> &R) {
  ArrowIdx<TG> M;
  arrow_load_map<TG>(S, lane, M);
#pragma unroll
  for (int t = 0; t < TG; t++) {
    const unsigned slot = (t & 1) ? M.ix[t >> 1] >> 16 : M.ix[t >> 1] & 0xffffu;
    const double val = Fg[slot != 0xffffu ? slot : 0u];
    R.v[t] = slot != 0xffffu ? val : 0.0;
  }
}
// All LDS traffic of the tile kernels goes through absolute byte addresses that are built once per launch and kept PACKED,
// two 16-bit addresses per register (a workgroup's dynamic LDS stays below 64 KiB): one v_and / v_lshrrev per access.
__device__ __forceinline__ unsigned pk_lo(unsigned w) { return w & 0xffffu; }
__device__ __forceinline__ unsigned pk_hi(unsigned w) { return w >> 16; }

template <int TG>
struct TileCols { unsigned a[(TG + 1) / 2]; };                   // byte address of x[column] per virtual-row step, packed
template <int TA>
struct TileAddr { unsigned rc[TA]; bool act; };                 // low half: address of the tile's (rotated) row s, high half: of its column s

// xb = byte address of the wave's x[0]; steps without an entry point at the lane's own dummy word (dmy), so neither the
// gather nor the scatter needs a predicate and the padding lanes of a step never meet on one address
template <int TG>
__device__ __forceinline__ void tile_cols(const rldl_dev_sym &S, int lane, unsigned xb, unsigned dmy, TileCols<TG> &C) {
  const unsigned *ap = reinterpret_cast<const unsigned *>(S.plan + S.po_avcol), *mp = reinterpret_cast<const unsigned *>(S.plan + S.po_avmap);
#pragma unroll
  for (int t2 = 0; t2 < (TG + 1) / 2; t2++) {
    const unsigned cw = ap[t2 * 64 + lane], mw = mp[t2 * 64 + lane];
    const unsigned lo = (mw & 0xffffu) != 0xffffu ? xb + 8u * (cw & 0xffffu) : dmy, hi = (mw >> 16) != 0xffffu ? xb + 8u * (cw >> 16) : dmy;
    C.a[t2] = lo | (hi << 16);                                    // (both halves stay below 2^16)
  }
}
template <int TA>
__device__ __forceinline__ void tile_addr(const rldl_dev_sym &S, int lane, unsigned xtb, TileAddr<TA> &A) {
  const unsigned w = reinterpret_cast<const unsigned *>(S.plan + S.po_tlane)[lane];
  const int I = (int)(w & 0xffu), J = (int)((w >> 8) & 0xffu);
  A.act = w != 0xffffffffu;
#pragma unroll
  for (int s = 0; s < TA; s++) {
    const unsigned r = A.act ? (unsigned)(TA * I + (s + J) % TA) : 0u, c = A.act ? (unsigned)(TA * J + (s + I) % TA) : 0u;
    A.rc[s] = (xtb + 8u * r) | ((xtb + 8u * c) << 16);
  }
}
// y = Linv c in place (strictly lower part times c, added to c) on the tail's x in LDS
template <int TA>
__device__ __forceinline__ void tile_fwd(const TileRegs<TA> &T, char *shb, const TileAddr<TA> &A) {
  double c[TA], p[TA];
  if (A.act) {
#pragma unroll
    for (int u = 0; u < TA; u++) c[u] = lds_ld(shb, pk_hi(A.rc[u]));
#pragma unroll
    for (int s = 0; s < TA; s++) {
      double acc = 0.0;
#pragma unroll
      for (int u = 0; u < TA; u++) acc = fma(T.v[s * TA + u], c[u], acc);
      p[s] = acc;
    }
  }
  wave_sync();                                                   // every read of c precedes the first add
  if (A.act) {
#pragma unroll
    for (int s = 0; s < TA; s++) lds_add(shb, pk_lo(A.rc[s]), p[s]);
  }
  wave_sync();
}
// x = Linv' w in place
template <int TA>
__device__ __forceinline__ void tile_bwd(const TileRegs<TA> &T, char *shb, const TileAddr<TA> &A) {
  double w[TA], p[TA];
  if (A.act) {
#pragma unroll
    for (int s = 0; s < TA; s++) w[s] = lds_ld(shb, pk_lo(A.rc[s]));
#pragma unroll
    for (int u = 0; u < TA; u++) {
      double acc = 0.0;
#pragma unroll
      for (int s = 0; s < TA; s++) acc = fma(T.v[s * TA + u], w[s], acc);
      p[u] = acc;
    }
  }
  wave_sync();
  if (A.act) {
#pragma unroll
    for (int u = 0; u < TA; u++) lds_add(shb, pk_hi(A.rc[u]), p[u]);
  }
  wave_sync();
}
// The whole permuted solve on the wave's x (LDS): rhs in; on return the tail slots hold their solution and every head slot
// holds -sum_r L(r, c) x_r (the caller adds y_c Dinv_c, which it still has in registers).  Straight-line code: no predicate
// besides the tile lanes' (one per phase).
//   jra: byte address of x[row of this lane's virtual row] (the dummy word for lanes without one); za[t]: byte address of the
//   caller's t-th x entry if that is a head entry, else the dummy word; dta: address of tail entry `lane` (dummy for lane >= g)
template <int TG, int TA, int TS, bool SCATTER>
__device__ __forceinline__ void tile_tri_solve(const ArrowRegs<TG> &R, const TileCols<TG> &C, const TileRegs<TA> &T, const TileAddr<TA> &A,
                                               double dtail, char *shb, unsigned jra, unsigned dta, const unsigned (&za)[TS], int lane,
                                               long long *tr = nullptr) {
  double ga[3] = {0.0, 0.0, 0.0};                                // three partial sums: the gather is not one dependent fma chain
#pragma unroll
  for (int t = 0; t < TG; t++)
  {
    ga[t % 3] = fma(-R.v[t], lds_ld(shb, (t & 1) ? pk_hi(C.a[t >> 1]) : pk_lo(C.a[t >> 1])), ga[t % 3]);
    if (t % 9 == 8) __builtin_amdgcn_sched_barrier(0);           // at most 9 reads in flight: their values need registers
  }
  wave_sync();                                                   // all reads of the head values are done
  lds_add(shb, jra, (ga[0] + ga[1]) + ga[2]);
  if (SCATTER) {
#pragma unroll
    for (int t = 0; t < TS; t++) lds_st(shb, za[t], 0.0);         // head slots become scatter accumulators
  }
  wave_sync();
  if (tr && lane == 0) tr[2] = wall_clock64();                   // gather done
  tile_fwd<TA>(T, shb, A);
  if (tr && lane == 0) tr[3] = wall_clock64();                   // forward product done
  lds_st(shb, dta, lds_ld(shb, dta) * dtail);                    // D^-1
  wave_sync();
  tile_bwd<TA>(T, shb, A);
  if (tr && lane == 0) tr[4] = wall_clock64();                   // backward product done
  if (SCATTER) {                                                 // SCATTER = false: the caller gathers the head entries itself (col_gather)
    const double xr = -lds_ld(shb, jra);
#pragma unroll
    for (int t = 0; t < TG; t++)                                 // transposed gather: -L(r, c) x_r into the head columns
      lds_add(shb, (t & 1) ? pk_hi(C.a[t >> 1]) : pk_lo(C.a[t >> 1]), R.v[t] * xr);
    wave_sync();
  }
}

// Backward coupling product gathered by the owner: lane (slot t) holds the entries L(r, c) of the column c it owns in slot t
// (second register copy of the coupling values, host tables po_cmap / po_crow) and the LDS addresses of their x_r, packed.
template <int TK>
struct ColRegs { double v[TK]; unsigned a[TK / 2]; };
template <int TK>
__device__ __forceinline__ void col_load(const rldl_dev_sym &S, const double *__restrict__ Fg, int lane, unsigned xtb, unsigned dmy, ColRegs<TK> &Q) {
  const unsigned *cm = reinterpret_cast<const unsigned *>(S.plan + S.po_cmap), *cr = reinterpret_cast<const unsigned *>(S.plan + S.po_crow);
#pragma unroll
  for (int k2 = 0; k2 < TK / 2; k2++) {
    const unsigned mw = cm[k2 * 64 + lane], rw = cr[k2 * 64 + lane];
    const unsigned s0 = mw & 0xffffu, s1 = mw >> 16;
    const double v0 = Fg[s0 != 0xffffu ? s0 : 0u], v1 = Fg[s1 != 0xffffu ? s1 : 0u];
    Q.v[2 * k2] = s0 != 0xffffu ? v0 : 0.0;
    Q.v[2 * k2 + 1] = s1 != 0xffffu ? v1 : 0.0;
    Q.a[k2] = (s0 != 0xffffu ? xtb + 8u * (rw & 0xffffu) : dmy) | ((s1 != 0xffffu ? xtb + 8u * (rw >> 16) : dmy) << 16);
  }
}
// steps [0, SP): the column this lane owns in the first constraint slot, [SP, TK): in the second (compile-time split)
template <int TK, int SP>
__device__ __forceinline__ void col_gather(const ColRegs<TK> &Q, const char *shb, double &r1, double &r2) {
  double a[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int k = 0; k < TK; k++) {
    const double xv = lds_ld(shb, (k & 1) ? pk_hi(Q.a[k >> 1]) : pk_lo(Q.a[k >> 1]));
    const int j = (k < SP ? 0 : 2) + (k & 1);
    a[j] = fma(Q.v[k], xv, a[j]);
    if (k % 8 == 7) __builtin_amdgcn_sched_barrier(0);           // at most 8 reads in flight: their values need registers
  }
  r1 = a[0] + a[1]; r2 = a[2] + a[3];
}

// Inverse of the tail's unit lower triangle, once per factorisation: lane c solves L22 X = e_c by forward substitution with
// the rows of L22 as LDS broadcast reads (one address per wave), X in registers; the columns then go through LDS into
// the (k, lane) tile order of Ti.  SM = compile-time bound on g.  One wave per instance.
// row I of the forward substitution of tile_invert_lds (template parameter for the same reason as elim_step)
template <int SM, int I>
__device__ __forceinline__ void invert_row(double (&X)[SM], lds_d2 (&rb)[2][LCH / 2], const lds_d2 *st, int g) {
  if (I < g) {                                                   // uniform
    double a0 = 0.0, a1 = 0.0;
#pragma unroll
    for (int j = 0; j < I; j++) {
      const int p = ((I * (I - 1)) >> 1) + j, c = p / LCH, e = p % LCH;
      if (e == 0) {
#pragma unroll
        for (int q = 0; q < LCH / 2; q++) rb[(c + 1) & 1][q] = st[(c + 1) * (LCH / 2) + q];
        __builtin_amdgcn_sched_barrier(0);
      }
      const double lv = rb[c & 1][e >> 1][e & 1];
      if (j & 1) a1 = fma(lv, X[j], a1); else a0 = fma(lv, X[j], a0);
    }
    X[I] -= a0 + a1;                                             // lanes c >= I: all X[j < I] are 0, X[I] stays delta_Ic
  }
}
template <int SM, int... Is>
__device__ __forceinline__ void invert_rows(std::integer_sequence<int, Is...>, double (&X)[SM], lds_d2 (&rb)[2][LCH / 2], const lds_d2 *st, int g) {
  (invert_row<SM, Is + 1>(X, rb, st, g), ...);
}
// The same inverse on the matrix cores (TILE_INVERT_MFMA, the default): X = L22^-1 by block forward substitution over 16 x 16 blocks,
//   X_JJ = L_JJ^-1,   X_IJ = -X_II (sum_{J <= K < I} L_IK X_KJ)   (I > J),
// every block product four v_mfma_f64_16x16x4_f64.  The 16 x 16 diagonal inverses come from the register substitution above, four
// blocks at a time (lane group d = lane / 16 inverts block d, column per lane) and go to LDS as dense blocks; the A operands (L_IK from
// the packed triangle, -X_II from those dense blocks) are LDS reads with compile-time offsets, the B operands never leave registers:
// register q of a lane's 16 x 16 result (rows lane / 16 + 4 q, column lane % 16) IS the B operand of k-chunk q.  All ten result blocks
// of a block column stay in registers until the column is stored in tile order.  64 MFMAs + 120 fmas per lane instead of 1225 fmas per lane:
// the vector-unit version spends g^2 / 2 fma slots per lane on a product of which g^3 / 6 multiply-adds are not zero.
// LDS: [0, tri) the triangle, zo = even(tri): 64 zero words (rows >= g read them), then NT dense blocks.
typedef double inv_v4d __attribute__((ext_vector_type(4)));
template <int SM>
__device__ __forceinline__ void tile_invert_mfma(const rldl_dev_sym &S, const rldl_dev_num &Nn, int inst, double *sh, int lane) {
  constexpr int NT = (SM + 15) / 16;
  const int g = S.arrow_g, tri = (g * (g - 1)) >> 1;
  double *Z = sh + ((tri + 1) & ~1), *Dd = Z + 64;
  const int lr = lane & 15, lq = lane >> 4;
  Z[lane] = 0.0;
  wave_sync();
  {                                                              // 1. the diagonal blocks, four at a time
    double X[16];
#pragma unroll
    for (int i = 0; i < 16; i++) X[i] = i == lr ? 1.0 : 0.0;
#pragma unroll
    for (int i = 1; i < 16; i++) {
      const int R = 16 * lq + i;                                 // row of L22; rows >= g are rows of the identity
      const double *row = R < g ? sh + ((R * (R - 1)) >> 1) + 16 * lq : Z;
      double a0 = 0.0, a1 = 0.0;
#pragma unroll
      for (int j = 0; j < i; j++) { if (j & 1) a1 = fma(row[j], X[j], a1); else a0 = fma(row[j], X[j], a0); }
      X[i] -= a0 + a1;
    }
    if (lq < NT) {
#pragma unroll
      for (int i = 0; i < 16; i++) Dd[lq * 256 + i * 16 + lr] = X[i];
    }
  }
  wave_sync();
  // 2. block forward substitution, one block column of X at a time; a finished column goes straight to the tile store (slot of
  //    X(row, col) from the host's table; 8-byte stores, the row is 9.8 KB and L2 merges them) -- holding all ten blocks for a staged
  //    write-out would cost 80 registers next to the factor phases' 200
  const unsigned short *ts = reinterpret_cast<const unsigned short *>(S.plan + S.po_tislot);
  double *To = Nn.Ti + (size_t)inst * S.ldTi;
  const double *lrow[NT];
#pragma unroll
  for (int I = 0; I < NT; I++) { const int r = 16 * I + lr; lrow[I] = (r < g ? sh + ((r * (r - 1)) >> 1) : Z) + lq; }
#pragma unroll
  for (int J = 0; J < NT; J++) {
    inv_v4d Xc[NT];                                              // blocks X_IJ, I >= J, in the MFMA result layout
    unsigned short sl[NT][4];
#pragma unroll
    for (int I = J; I < NT; I++)
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int row = 16 * I + lq + 4 * q, col = 16 * J + lr;
        const unsigned short v = ts[(row < g ? row : 0) * 64 + col];
        sl[I][q] = row < g && col < row ? v : (unsigned short)0xffffu;
      }
#pragma unroll
    for (int q = 0; q < 4; q++) Xc[J][q] = Dd[J * 256 + (lq + 4 * q) * 16 + lr];
#pragma unroll
    for (int I = J + 1; I < NT; I++) {
      inv_v4d acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int K = J; K < I; K++)
#pragma unroll
        for (int kk = 0; kk < 4; kk++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(lrow[I][16 * K + 4 * kk], Xc[K][kk], acc, 0, 0, 0);
      inv_v4d xi = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int kk = 0; kk < 4; kk++) xi = __builtin_amdgcn_mfma_f64_16x16x4f64(-Dd[I * 256 + lr * 16 + 4 * kk + lq], acc[kk], xi, 0, 0, 0);
      Xc[I] = xi;
    }
#pragma unroll
    for (int I = J; I < NT; I++)
#pragma unroll
      for (int q = 0; q < 4; q++)
        if (sl[I][q] != 0xffffu) To[sl[I][q]] = Xc[I][q];
  }
}
#ifndef TILE_INVERT_MFMA
#define TILE_INVERT_MFMA 1
#endif
// (the triangle is in sh[0, g (g - 1) / 2), row-major packed; the same words then become the staging buffer)
template <int SM>
__device__ __forceinline__ void tile_invert_lds(const rldl_dev_sym &S, const rldl_dev_num &Nn, int inst, double *sh, int lane) {
  if (TILE_INVERT_MFMA) { tile_invert_mfma<SM>(S, Nn, inst, sh, lane); return; }
  const int g = S.arrow_g;
  wave_sync();
  double X[SM];
#pragma unroll
  for (int i = 0; i < SM; i++) X[i] = i == lane ? 1.0 : 0.0;
  // the packed triangle is consumed front to back: LCH entries are read ahead of the fmas that use them (see arrow_factor_body;
  // the reads may run up to 2 LCH doubles past the triangle -- the launchers size the LDS for that)
  const lds_d2 *st = reinterpret_cast<const lds_d2 *>(sh);
  lds_d2 rb[2][LCH / 2];
#pragma unroll
  for (int q = 0; q < LCH / 2; q++) rb[0][q] = st[q];
  invert_rows<SM>(std::make_integer_sequence<int, SM - 1>(), X, rb, st, g);
  unsigned short sl[SM];                                         // (fetched here, not ahead of the substitution: 55 registers)
  {
    const unsigned short *ts = reinterpret_cast<const unsigned short *>(S.plan + S.po_tislot);
#pragma unroll
    for (int i = 1; i < SM; i++) sl[i] = i < g ? ts[i * 64 + lane] : (unsigned short)0xffffu;
  }
  wave_sync();                                                   // the triangle is dead: same LDS becomes the tile-order buffer
#pragma unroll
  for (int i = 1; i < SM; i++)
    if (sl[i] != 0xffffu) sh[sl[i]] = X[i];
  wave_sync();
  double *To = Nn.Ti + (size_t)inst * S.ldTi;
  for (int p = lane; p < S.nTi; p += WAVE) To[p] = sh[p];
}
template <int SM>
__device__ __forceinline__ void tile_invert_body(const rldl_dev_sym &S, const rldl_dev_num &Nn, const int *__restrict__ mask, int inst) {
  const int lane = threadIdx.x;
  if (mask && !mask[inst]) return;
  extern __shared__ double sh[];                                 // g (g - 1) / 2 doubles: the triangle, then the staging buffer
  const int g = S.arrow_g, tri = (g * (g - 1)) >> 1;
  const double *Lg = Nn.F + (size_t)inst * S.ldF + S.nOp;        // row-major packed triangle of the tail group (plan slot order)
  for (int i = lane; i < tri; i += WAVE) sh[i] = Lg[i];
  tile_invert_lds<SM>(S, Nn, inst, sh, lane);
}
template <int SM>
__global__ __launch_bounds__(WAVE) void k_tile_invert(rldl_dev_sym S, rldl_dev_num Nn, const int *__restrict__ mask) {
  tile_invert_body<SM>(S, Nn, mask, blockIdx.x);
}

// LDS per wave of the tile kernels (pws doubles): x (xdw doubles, incl. the padding rows of the last block row), 64 dummy words
// (one per lane, always 0: the target of every access that has no entry), then -- k_tile_admm only -- the per-slot constants of
// the ADMM step, [5][3 * 64]: l, u, rho, rho_inv (constraint slots) and the head's Dinv (0 for tail entries).  The constants
// are private to a lane: LDS is used as a second register file, reads are conflict-free.
template <int TMAX, int TG, int TA>
__global__ __launch_bounds__(256, 2) void k_tile_solve(rldl_dev_sym S, rldl_dev_num Nn, double *__restrict__ b_all, int xdw) {
  extern __shared__ __attribute__((aligned(16))) double sh[];
  const int lane = threadIdx.x & (WAVE - 1), wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), wpb = blockDim.x >> 6;
  const int inst = blockIdx.x * wpb + wv;
  if (inst >= Nn.batch) return;
  char *shb = reinterpret_cast<char *>(sh);
  const unsigned xb = (unsigned)wv * (unsigned)(xdw + WAVE) * 8u, dmy = xb + 8u * (unsigned)(xdw + lane);
  const double *Fg = Nn.F + (size_t)inst * S.ldF;
  double *b = b_all + (size_t)inst * S.N;
  const int *permg = S.plan + S.po_perm;
  const int g0 = S.arrow_g0, g = S.arrow_g;
  ArrowRegs<TG> R;
  TileRegs<TA> T;
  arrow_load_val_global<TG>(S, Fg, lane, R);
  tile_load<TA>(S, Nn.Ti + (size_t)inst * S.ldTi, lane, T);
  TileCols<TG> C;
  TileAddr<TA> A;
  tile_cols<TG>(S, lane, xb, dmy, C);
  tile_addr<TA>(S, lane, xb + 8u * (unsigned)g0, A);
  const unsigned jra = lane < S.arrow_vrows ? xb + 8u * (unsigned)(S.plan + S.po_avrow)[lane] : dmy;
  const unsigned dta = lane < g ? xb + 8u * (unsigned)(g0 + lane) : dmy;
  const double dtail = Fg[S.nS + g0 + (lane < g ? lane : 0)];
  int oo[TMAX];
  unsigned xa[TMAX], za[TMAX];
  double vb[TMAX];
#pragma unroll
  for (int t = 0; t < TMAX; t++) {
    const int j = t * WAVE + lane;
    oo[t] = j < S.N ? permg[j] : -1;
    xa[t] = j < S.N ? xb + 8u * (unsigned)j : dmy;
    za[t] = j < S.N && (j < g0 || j >= g0 + g) ? xa[t] : dmy;
  }
#pragma unroll
  for (int t = 0; t < TMAX; t++) vb[t] = oo[t] >= 0 ? b[oo[t]] : 0.0;   // permute_x  qdldl_interface.c:538-541
  double dv[TMAX], rr_[TMAX];                                   // head Dinv (0 for tail entries) and rho_inv
  const double *ri = Nn.rho_inv + (size_t)inst * S.m;
#pragma unroll
  for (int t = 0; t < TMAX; t++) {
    const int j = t * WAVE + lane;
    const double d = Fg[S.nS + (j < S.N ? j : 0)];
    dv[t] = za[t] != dmy ? d : 0.0;
    rr_[t] = S.polish ? 0.0 : ri[oo[t] >= S.n ? oo[t] - S.n : 0];
  }
  for (int j = S.N + lane; j < xdw + WAVE; j += WAVE) lds_st(shb, xb + 8u * (unsigned)j, 0.0);   // padding rows, dummy words
  wave_sync();
#pragma unroll
  for (int t = 0; t < TMAX; t++) lds_st(shb, xa[t], vb[t]);
  wave_sync();
  tile_tri_solve<TG, TA, TMAX, true>(R, C, T, A, dtail, shb, jra, dta, za, lane);
#pragma unroll
  for (int t = 0; t < TMAX; t++) {
    if (oo[t] < 0) continue;
    const double xv = fma(vb[t], dv[t], lds_ld(shb, xa[t]));    // head: x_c = y_c Dinv_c - sum_r L(r,c) x_r
    if (S.polish || oo[t] < S.n) b[oo[t]] = xv;
    else b[oo[t]] = vb[t] + rr_[t] * xv;                        // qdldl_interface.c:568-579
  }
}

// Round 3 form of the plugin `solve` on tile handles.  k_tile_solve above starts every wave with a chain of DEPENDENT loads
// (slot tables -> gathered factor values, perm -> b) and holds the coupling values in registers (164 VGPRs: 12 waves per CU, so
// 4096 instances are 1 1/3 rounds).  Here every global load is issued at wave start from addresses that need no table:
//   * coupling values: the slots [0, nOp) of the factor row are ONE contiguous piece -> coalesced LDS-DMA (16 bytes per lane),
//     they stay in LDS and are read where they are used (gather and scatter) through the virtual-row slot table;
//   * Ti: slot of (register k, lane) = first slot of k + number of set bits below the lane in k's lane mask (po_tmask: scalar
//     loads + v_mbcnt), i.e. coalesced loads without a per-lane table;
//   * b, rho_inv in ORIGINAL order and Dinv in permuted order: plain coalesced rows; b enters x through the inverse
//     permutation as an LDS scatter and leaves through the same addresses, so permute_x / permutet_x cost no dependent load.
// The tables (L2-resident, shared by all instances) only feed LDS addresses and arrive under the value loads.  The coupling values
// move from LDS to registers once they have landed (a variant that left them in LDS to fit 128 registers / 16 waves per CU spent
// 0.9 us per wave in each of the gather and the scatter on dependent LDS round trips and was no faster: DESIGN 7.1).  Same products as k_tile_solve; the head's y_c Dinv_c enters as the
// initial value of the scatter accumulator instead of a closing fma (last-bit differences; RLDL_SOLVE_V2=1 selects the old kernel).  TRACE: wave timeline (rldl_batch_trace_solve).
// NT: the factor rows (tile of Ti, coupling values) are read with non-temporal loads (cache-policy bit nt): rows that will not be read
// again before they are evicted do not displace the Infinity Cache's contents (rldl_batch_set_cache_policy).
template <int TMAX, int TG, int TA, bool TRACE, bool NT>
__global__ __launch_bounds__(256, (TA <= 5 && TG <= 18) ? 3 : 2) void k_tile_solve3(rldl_dev_sym S, rldl_dev_num Nn, double *__restrict__ b_all, int xdw, int cwp,
                                                         long long *__restrict__ trace) {
  typedef __attribute__((address_space(3))) void *lptr_t;
  extern __shared__ __attribute__((aligned(16))) double sh[];
  const int lane = threadIdx.x & (WAVE - 1), wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), wpb = blockDim.x >> 6;
  const int inst = blockIdx.x * wpb + wv;
  if (inst >= Nn.batch) return;
  long long *tr = TRACE ? trace + 8 * (size_t)inst : nullptr;
  if (TRACE && lane == 0) tr[0] = wall_clock64();
  char *shb = reinterpret_cast<char *>(sh);
  const unsigned pws = (unsigned)(cwp + xdw + WAVE);              // doubles per wave: coupling values | x | one dummy word per lane
  const unsigned cb = (unsigned)wv * pws * 8u, xb = cb + 8u * (unsigned)cwp, dmy = xb + 8u * (unsigned)(xdw + lane);
  const double *Fg = Nn.F + (size_t)inst * S.ldF;
  const int g0 = S.arrow_g0, g = S.arrow_g;
  // Every load below is a BUFFER load: 32-bit offsets, and a lane whose offset lies past the end of its array receives 0 (a store is
  // dropped) -- no range selects, so the compiler has nothing to turn into branches around loads.
  const __amdgpu_buffer_rsrc_t rP = __builtin_amdgcn_make_buffer_rsrc((void *)S.plan, 0, 4 * S.plan_words, 0x00020000);
  const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc((void *)(b_all + (size_t)inst * S.N), 0, 8 * S.N, 0x00020000);
  const __amdgpu_buffer_rsrc_t rD = __builtin_amdgcn_make_buffer_rsrc((void *)(Fg + S.nS), 0, 8 * S.N, 0x00020000);
  const __amdgpu_buffer_rsrc_t rR = __builtin_amdgcn_make_buffer_rsrc((void *)(Nn.rho_inv + (size_t)inst * S.m), 0, S.polish ? 0 : 8 * S.m, 0x00020000);
  const __amdgpu_buffer_rsrc_t rTi = __builtin_amdgcn_make_buffer_rsrc((void *)(Nn.Ti + (size_t)inst * S.ldTi), 0, 8 * S.nTi, 0x00020000);
  const unsigned l8 = 8u * (unsigned)lane;
  // ---- phase A: issue every load of the wave.  Tables first (L2-resident, needed first; vmcnt retires in order), then the rows
  // of this instance: b, Dinv, the coupling values (LDS-DMA), the lane's tile of Linv. ----
  // (the table words of a lane sit side by side in 16-byte records, [chunk][lane][4]: one load per chunk, rldl_plan.c po_spack)
  unsigned pw[TMAX], cw[(TG + 1) / 2], mw[(TG + 1) / 2], rcw[TA], tlw, jrw;
  {
    constexpr int H = (TG + 1) / 2, NWP = TMAX + 2 * H + TA + 2, NCH = (NWP + 3) / 4;
    pv_v4u rk[NCH];
#pragma unroll
    for (int c = 0; c < NCH; c++) rk[c] = __builtin_amdgcn_raw_buffer_load_b128(rP, 16u * (unsigned)lane + 4u * (unsigned)(S.po_spack + 256 * c), 0, 0);
#pragma unroll
    for (int t = 0; t < TMAX; t++) pw[t] = rk[t / 4][t % 4];
#pragma unroll
    for (int t2 = 0; t2 < H; t2++) { cw[t2] = rk[(TMAX + t2) / 4][(TMAX + t2) % 4]; mw[t2] = rk[(TMAX + H + t2) / 4][(TMAX + H + t2) % 4]; }
#pragma unroll
    for (int sx = 0; sx < TA; sx++) rcw[sx] = rk[(TMAX + 2 * H + sx) / 4][(TMAX + 2 * H + sx) % 4];
    tlw = rk[(TMAX + 2 * H + TA) / 4][(TMAX + 2 * H + TA) % 4];
    jrw = rk[(TMAX + 2 * H + TA + 1) / 4][(TMAX + 2 * H + TA + 1) % 4];
  }
  pv_v2u blr[TMAX], dlr[TMAX], rrr[TMAX];
#pragma unroll
  for (int t = 0; t < TMAX; t++) {
    blr[t] = __builtin_amdgcn_raw_buffer_load_b64(rB, l8 + 512u * (unsigned)t, 0, 0);    // b by original index (0 past N)
    dlr[t] = __builtin_amdgcn_raw_buffer_load_b64(rD, l8 + 512u * (unsigned)t, 0, 0);    // Dinv by permuted position
    rrr[t] = __builtin_amdgcn_raw_buffer_load_b64(rR, l8 + 512u * (unsigned)t - 8u * (unsigned)S.n, 0, 0);   // rho_inv of row i - n (0 for variables and when polishing)
  }
  const pv_v2u dtr = __builtin_amdgcn_raw_buffer_load_b64(rD, lane < g ? 8u * (unsigned)(g0 + lane) : 0xffffffffu, 0, 0);
  {                                                               // coupling values: slots [0, nOp) of the factor row -> LDS, one coalesced stream of
    const __amdgpu_buffer_rsrc_t rC = __builtin_amdgcn_make_buffer_rsrc((void *)Fg, 0, 8 * S.nOp, 0x00020000);   // 1 KB pieces (lanes past the end write 0.0 into the padding)
    lptr_t dst = (lptr_t)(sh + (size_t)wv * pws);
    const unsigned l16 = 16u * (unsigned)lane;
    for (int pc = 0; pc < cwp; pc += 128)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rC, (lptr_t)((__attribute__((address_space(3))) double *)dst + pc), 16, l16, 8 * pc, 0, NT ? 2 : 0);
  }
  TileRegs<TA> T;
  {                                                               // slot of (register k, lane) = entries of the registers before k + set bits of k's lane mask below the lane
    sv_cptr_t tk = (sv_cptr_t)(unsigned long long)(S.plan + S.po_tmask);   // (scalar loads: the masks are uniform)
    unsigned long long msk[TA * TA];
#pragma unroll
    for (int k = 0; k < TA * TA; k++) msk[k] = (unsigned long long)(unsigned)tk[2 * k] | ((unsigned long long)(unsigned)tk[2 * k + 1] << 32);
    __builtin_amdgcn_sched_barrier(0);                            // all masks with ONE wait, not one wait per pair of registers
    unsigned first = 0;
#pragma unroll
    for (int k = 0; k < TA * TA; k++) {
      const unsigned rank = __builtin_amdgcn_mbcnt_hi((unsigned)(msk[k] >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)msk[k], 0u));
      const pv_v2u r = __builtin_amdgcn_raw_buffer_load_b64(rTi, pv_select(msk[k], (first + rank) << 3, 0xffffffffu), 0, NT ? 2 : 0);
      T.v[k] = __hiloint2double((int)r.y, (int)r.x);
      first += (unsigned)__builtin_popcountll(msk[k]);
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  if (TRACE && lane == 0) tr[7] = wall_clock64();                 // every load is issued
  // ---- phase B: LDS addresses from the tables; b into x through the inverse permutation ----
  for (int j = S.N + lane; j < xdw + WAVE; j += WAVE) lds_st(shb, xb + 8u * (unsigned)j, 0.0);   // padding rows, dummy words
  wave_sync();
  unsigned za[TMAX];                                              // x slot of permuted position j = t * 64 + lane when j is a head position, else the dummy word
#pragma unroll
  for (int t = 0; t < TMAX; t++) {
    const int j = t * WAVE + lane;
    za[t] = j < S.N && (j < g0 || j >= g0 + g) ? xb + l8 + 512u * (unsigned)t : dmy;
    lds_st(shb, xb + 8u * pw[t], __hiloint2double((int)blr[t].y, (int)blr[t].x));       // permute_x  qdldl_interface.c:538-541 (indices past N: 0.0 to the dummy word)
  }
  TileCols<TG> C, V;                                              // C: byte address of x[column] per step; V: of the step's coupling value
#pragma unroll
  for (int t2 = 0; t2 < (TG + 1) / 2; t2++) {
    const bool l0 = (mw[t2] & 0xffffu) != 0xffffu, h0 = (mw[t2] >> 16) != 0xffffu;
    C.a[t2] = (l0 ? xb + 8u * (cw[t2] & 0xffffu) : dmy) | ((h0 ? xb + 8u * (cw[t2] >> 16) : dmy) << 16);
    V.a[t2] = (l0 ? cb + 8u * (mw[t2] & 0xffffu) : dmy) | ((h0 ? cb + 8u * (mw[t2] >> 16) : dmy) << 16);
  }
  TileAddr<TA> A;
  A.act = tlw != 0xffffffffu;
  {
    const unsigned xt2 = (xb + 8u * (unsigned)g0) * 0x10001u;    // both halves: byte address of the tail's first entry
#pragma unroll
    for (int sx = 0; sx < TA; sx++) A.rc[sx] = xt2 + (rcw[sx] << 3);
  }
  const unsigned jra = xb + 8u * jrw;                             // (lanes without a virtual row: the table holds their dummy word)
  const unsigned dta = lane < g ? xb + 8u * (unsigned)(g0 + lane) : dmy;
  const double dtail = __hiloint2double((int)dtr.y, (int)dtr.x);
  wave_sync();
  double hd[TMAX];                                                // head entries are final after the (empty) forward pass of the head: y_c Dinv_c
#pragma unroll
  for (int t = 0; t < TMAX; t++) hd[t] = lds_ld(shb, za[t]) * __hiloint2double((int)dlr[t].y, (int)dlr[t].x);   // (no head position: dummy word, 0)
  wait_dma();                                                     // the coupling values have landed
  wave_sync();
  if (TRACE && lane == 0) tr[1] = wall_clock64();
  ArrowRegs<TG> R;                                                // the coupling values of the lane's virtual row: LDS -> registers, all reads in flight at once
#pragma unroll
  for (int t = 0; t < TG; t++) R.v[t] = lds_ld(shb, (t & 1) ? pk_hi(V.a[t >> 1]) : pk_lo(V.a[t >> 1]));
  {
    double ga[3] = {0.0, 0.0, 0.0};
#pragma unroll
    for (int t = 0; t < TG; t++) {
      ga[t % 3] = fma(-R.v[t], lds_ld(shb, (t & 1) ? pk_hi(C.a[t >> 1]) : pk_lo(C.a[t >> 1])), ga[t % 3]);
      if (t % 9 == 8) __builtin_amdgcn_sched_barrier(0);         // at most 9 reads in flight: their values need registers
    }
    wave_sync();                                                  // all reads of the head values are done
    lds_add(shb, jra, (ga[0] + ga[1]) + ga[2]);
#pragma unroll
    for (int t = 0; t < TMAX; t++) lds_st(shb, za[t], za[t] != dmy ? hd[t] : 0.0);   // head slots become the accumulators of x_c = y_c Dinv_c - sum_r L(r, c) x_r
    wave_sync();
  }
  if (TRACE && lane == 0) tr[2] = wall_clock64();
  tile_fwd<TA>(T, shb, A);
  if (TRACE && lane == 0) tr[3] = wall_clock64();
  lds_st(shb, dta, lds_ld(shb, dta) * dtail);                    // D^-1
  wave_sync();
  tile_bwd<TA>(T, shb, A);
  if (TRACE && lane == 0) tr[4] = wall_clock64();
  {
    const double xr = -lds_ld(shb, jra);
#pragma unroll
    for (int t = 0; t < TG; t++)                                 // transposed gather: -L(r, c) x_r into the head columns
      lds_add(shb, (t & 1) ? pk_hi(C.a[t >> 1]) : pk_lo(C.a[t >> 1]), R.v[t] * xr);
    wave_sync();
  }
  if (TRACE && lane == 0) tr[5] = wall_clock64();
#pragma unroll
  for (int t = 0; t < TMAX; t++) {                                // permutet_x + the z~ epilogue, coalesced in original order (stores past N are dropped)
    const int i = t * WAVE + lane;
    const double xv = lds_ld(shb, xb + 8u * pw[t]);
    const double bv = __hiloint2double((int)blr[t].y, (int)blr[t].x), rv = __hiloint2double((int)rrr[t].y, (int)rrr[t].x);
    const double out = (S.polish || i < S.n) ? xv : fma(rv, xv, bv);      // qdldl_interface.c:568-579
    pv_v2u o;
    o.x = (unsigned)__double2loint(out); o.y = (unsigned)__double2hiint(out);
    __builtin_amdgcn_raw_buffer_store_b64(o, rB, l8 + 512u * (unsigned)t, 0, 0);
  }
  if (TRACE && lane == 0) tr[6] = wall_clock64();
}

// `iters` fused ADMM iterations per launch (as k_arrow_admm).  Everything the loop touches is on chip: both copies of the
// coupling values, the tile of Linv, the packed LDS addresses, x / z / y / q and rho_inv in registers, x~ and the other
// per-slot constants in LDS -- no global memory access inside the loop and no branch besides the tile lanes' predicate.
// The permuted positions are dealt to (slot, lane) by the host (po_tpos): slot 0 holds the variables, slots 1 and 2 the
// constraints, so the step code of a slot is uniform (S.tile_vslots == 1 is a condition of tile_admm_ok).
#define TILE_SLOTS 3
template <int TG, int TA, int TK, int SP, bool TRACE, bool MULTI = false>
__global__ __launch_bounds__(256, 2) void k_tile_admm(rldl_dev_sym S, rldl_dev_num Nn, rldl_dev_admm W, int xdw, int iters, rldl_dev_multi M) {
  extern __shared__ __attribute__((aligned(16))) double sh[];
  const int lane = threadIdx.x & (WAVE - 1), wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), wpb = blockDim.x >> 6;
  int bid = blockIdx.x;
  if (MULTI) { const int g = multi_group(M.first_tile, M.ngroups, bid); S = M.S[g]; Nn = M.N[g]; W = M.W[g]; xdw = M.xdw[g]; bid -= M.first_tile[g]; }
  const int inst = bid * wpb + wv;
  if (inst >= Nn.batch) return;
  // first launch of a solve without termination checks: the start of osqp_solve (status = OSQP_UNSOLVED, cold start) is done here
  // instead of by a k_solve_begin launch in front
  const int bf = MULTI ? 0 : W.begin_flags;
  if (bf & 1) { if (lane == 0) W.status[inst] = ST_UNSOLVED; }
  else if (W.status[inst] != ST_UNSOLVED) return;
  const bool cold = (bf & 2) != 0;
  long long *tr = TRACE && W.trace ? W.trace + 8 * (size_t)inst : nullptr;
  if (TRACE && tr && lane == 0) tr[7] = wall_clock64();
  char *shb = reinterpret_cast<char *>(sh);
  constexpr int TS = TILE_SLOTS, CW = TS * WAVE;                 // doubles per constant array
  const unsigned xb = (unsigned)wv * (unsigned)(xdw + WAVE + 4 * CW) * 8u, dmy = xb + 8u * (unsigned)(xdw + lane);
  const unsigned cb = xb + 8u * (unsigned)(xdw + WAVE + lane);   // this lane's column of the constant arrays: l, u, rho, Dinv
  const int *permg = S.plan + S.po_perm, *tpos = S.plan + S.po_tpos;
  const int n = S.n, m = S.m, g0 = S.arrow_g0, g = S.arrow_g;
  const size_t io = (size_t)inst;
  const double *Fg = Nn.F + io * S.ldF;
  ArrowRegs<TG> R;
  TileRegs<TA> T;
  arrow_load_val_global<TG>(S, Fg, lane, R);
  tile_load<TA>(S, Nn.Ti + io * S.ldTi, lane, T);
  TileCols<TG> C;
  TileAddr<TA> A;
  tile_cols<TG>(S, lane, xb, dmy, C);
  tile_addr<TA>(S, lane, xb + 8u * (unsigned)g0, A);
  const unsigned jra = lane < S.arrow_vrows ? xb + 8u * (unsigned)(S.plan + S.po_avrow)[lane] : dmy;
  const unsigned dta = lane < g ? xb + 8u * (unsigned)(g0 + lane) : dmy;
  const double dtail = Fg[S.nS + g0 + (lane < g ? lane : 0)];
  // TK > 0: the owner's copy of the coupling values (backward product as a gather).  TK == 0: no copies -- the backward product is the
  // scatter of tile_tri_solve<.., true> (atomics into the head slots of x), for patterns whose owner-gather steps fit no compiled split
  ColRegs<(TK > 0 ? TK : 2)> Q;
  if constexpr (TK > 0) col_load<TK>(S, Fg, lane, xb + 8u * (unsigned)g0, dmy, Q);
  unsigned zh[TS];                                               // TK == 0: address of the slot's x entry when that is a head entry, else the dummy word
  int oo[TS];                                                    // index of the slot's entry in its own arrays (variable 0..n-1, constraint 0..m-1), -1: none
  unsigned xz[TS];                                               // address of x[position] | address where the solve leaves the entry's x (dummy: head entry) << 16
  double va[TS], vb[TS], rinv[TS];                               // slot 0: x, q; slots 1, 2: z, y, rho_inv
  for (int j = S.N + lane; j < xdw + WAVE; j += WAVE) lds_st(shb, xb + 8u * (unsigned)j, 0.0);   // padding rows, dummy words
  {
    const double *ri = Nn.rho_inv + io * m, *x = W.x + io * n, *z = W.z + io * m, *y = W.y + io * m;
    const double *q = W.q + io * n, *l = W.l + io * m, *u = W.u + io * m, *rv = W.rho_vec + io * m;
#pragma unroll
    for (int t = 0; t < TS; t++) {
      const int jp = tpos[t * WAVE + lane];
      const bool on = jp >= 0, var = t == 0, head = on && (jp < g0 || jp >= g0 + g);
      const int o = on ? permg[jp] : 0, i = var ? o : o - n;      // (host: slot 0 holds perm < n, the others perm >= n)
      oo[t] = on ? i : -1;
      const unsigned xa = on ? xb + 8u * (unsigned)jp : dmy;
      xz[t] = xa | ((head && TK > 0 ? dmy : xa) << 16);            // (scatter variant: a head slot of x ends as -sum_r L(r, c) x_r and is read like a tail entry)
      zh[t] = head && TK == 0 ? xa : dmy;
      va[t] = !on || cold ? 0.0 : var ? x[i] : z[i];
      vb[t] = !on ? 0.0 : var ? q[i] : cold ? 0.0 : y[i];
      const bool con = on && !var;
      rinv[t] = con ? ri[i] : 0.0;
      const unsigned ca = cb + 8u * (unsigned)(t * WAVE);
      lds_st(shb, ca, con ? l[i] : 0.0);
      lds_st(shb, ca + 8u * CW, con ? u[i] : 0.0);
      lds_st(shb, ca + 16u * CW, con ? rv[i] : 0.0);
      lds_st(shb, ca + 24u * CW, head ? Fg[S.nS + jp] : 0.0);
    }
  }
  const double alpha = W.alpha, sigma = W.sigma;
#pragma clang loop unroll(disable)
  for (int it = 0; it < iters; it++) {
    const bool last = it + 1 == iters;
    const bool trit = TRACE && tr && (W.trace_iter < 0 ? last : it == W.trace_iter);
    if (trit && lane == 0) tr[0] = wall_clock64();
    // the packed addresses stay packed: without this every unpacked LDS address becomes a loop-invariant register of its own
#pragma unroll
    for (int t2 = 0; t2 < (TG + 1) / 2; t2++) asm volatile("" : "+v"(C.a[t2]));
#pragma unroll
    for (int s2 = 0; s2 < TA; s2++) asm volatile("" : "+v"(A.rc[s2]));
#pragma unroll
    for (int t = 0; t < TS; t++) asm volatile("" : "+v"(xz[t]), "+v"(oo[t]));   // (oo: else the 15 store addresses of the last iteration are hoisted)
    if constexpr (TK > 0) {
#pragma unroll
      for (int k2 = 0; k2 < TK / 2; k2++) asm volatile("" : "+v"(Q.a[k2]));
    }
    unsigned cbo = cb;
    asm volatile("" : "+v"(cbo));
    const unsigned za[TS] = {zh[0], zh[1], zh[2]};               // (TK > 0: all dummy words, no scatter)
    double rhs[TS];                                              // compute_rhs (auxil.c:164-178) in permuted order
    rhs[0] = sigma * va[0] - vb[0];
#pragma unroll
    for (int t = 1; t < TS; t++) rhs[t] = va[t] - rinv[t] * vb[t];
#pragma unroll
    for (int t = 0; t < TS; t++) lds_st(shb, pk_lo(xz[t]), rhs[t]);
    wave_sync();
    if (trit && lane == 0) tr[1] = wall_clock64();              // rhs in LDS
    tile_tri_solve<TG, TA, TS, TK == 0>(R, C, T, A, dtail, shb, jra, dta, za, lane, trit ? tr : nullptr);
    double hs[TS] = {0.0, 0.0, 0.0};                             // sum_r L(r, c) x_r of the head entry this lane owns in slot t (0: tail entry)
    if constexpr (TK > 0) col_gather<TK, SP>(Q, shb, hs[1], hs[2]);
    if (trit && lane == 0) tr[5] = wall_clock64();              // backward coupling product done
    double xs_[TS], dinv[TS], lo[TS], hi[TS], rho[TS], dl[TS];
#pragma unroll
    for (int t = 0; t < TS; t++) {                               // all reads first: one LDS round trip for the whole update
      const unsigned ca = cbo + 8u * (unsigned)(t * WAVE);
      xs_[t] = lds_ld(shb, pk_hi(xz[t]));                        // tail entry: its solution; head entry: 0
      dinv[t] = lds_ld(shb, ca + 24u * CW);                      // head: x_c = y_c Dinv_c - sum_r L(r,c) x_r, y_c = rhs_c; tail: 0
      if (t > 0) { lo[t] = lds_ld(shb, ca); hi[t] = lds_ld(shb, ca + 8u * CW); rho[t] = lds_ld(shb, ca + 16u * CW); }
    }
    {                                                            // slot 0: update_x (auxil.c:188-201)
      const double xp = va[0];
      const double sv = fma(rhs[0], dinv[0], xs_[0] - hs[0]);
      const double xn = alpha * sv + (1.0 - alpha) * xp;
      va[0] = xn; dl[0] = xn - xp;
    }
#pragma unroll
    for (int t = 1; t < TS; t++) {                               // slots 1, 2: z_tilde, update_z, update_y
      const double zp = va[t], yi = vb[t], r = rinv[t];
      const double sv = fma(rhs[t], dinv[t], xs_[t] - hs[t]);
      const double zt = rhs[t] + r * sv;                       // z_tilde = (z_prev - rho_inv y) + rho_inv nu, qdldl_interface.c:577-579
      const double mix = alpha * zt + (1.0 - alpha) * zp;
      double zn = mix + r * yi;                                // update_z :203-215
      zn = fmin(fmax(zn, lo[t]), hi[t]);                       // project, proj.c:4-14
      const double d = rho[t] * (mix - zn);                    // update_y :217-228
      va[t] = zn; vb[t] = yi + d; dl[t] = d;
    }
    if (last) {                                                  // uniform: the only stores of the launch
      if (oo[0] >= 0) {
        (W.x + io * n)[oo[0]] = va[0];
        if (W.write_delta) (W.delta_x + io * n)[oo[0]] = dl[0];
      }
#pragma unroll
      for (int t = 1; t < TS; t++)
        if (oo[t] >= 0) {
          (W.z + io * m)[oo[t]] = va[t];
          (W.y + io * m)[oo[t]] = vb[t];
          if (W.write_delta) (W.delta_y + io * m)[oo[t]] = dl[t];
        }
    }
    if (trit && lane == 0 && W.trace_iter >= 0) tr[6] = wall_clock64();
  }
  if (tr && lane == 0 && W.trace_iter < 0) tr[6] = wall_clock64();
}

// ------------------------------------------------------------------------------------------------
// Horizon change (osqp_update_recursive, src/recursive_ldl.c:1973-2016; update_AP_matrices :1675-1778;
// LDL_update_from_pivot :946-1110).  Every horizon N <= Nmax has its own resident workspace; moving from one to another
//   k_horizon_values : carries the per-instance P / A values of the stages before the pivot (unscaled on the way),
//   k_horizon_state  : maps the unscaled iterates (x prefix; y of the kept row blocks; terminal rows -> terminal rows),
//   k_horizon_adopt  : copies the factor columns of the blocks before the pivot from the old horizon's factor (both
//                      patterns share that prefix of L) and names, per instance, the block the recursion restarts at.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(WAVE) void k_horizon_values(rldl_dev_sym So, rldl_dev_admm Wo, const double *__restrict__ oPx,
                                                         const double *__restrict__ oAx, int col_keep, double *__restrict__ nPx, int ldP,
                                                         double *__restrict__ nAx, int ldA) {
  const int inst = blockIdx.x, lane = threadIdx.x;
  const double *P = oPx + (size_t)inst * So.nnzP, *A = oAx + (size_t)inst * So.nnzA;
  double *Pn = nPx + (size_t)inst * ldP, *An = nAx + (size_t)inst * ldA;
  const bool sc = Wo.scaling != 0;
  const double *Dinv = sc ? Wo.sDinv + (size_t)inst * So.n : nullptr, *Einv = sc ? Wo.sEinv + (size_t)inst * So.m : nullptr;
  const double cinv = sc ? Wo.scinv[inst] : 1.0;
  for (int j = lane; j < col_keep; j += WAVE) {
    for (int p = So.Pp[j]; p < So.Pp[j + 1]; p++) Pn[p] = sc ? ((P[p] * cinv) * Dinv[So.Pi[p]]) * Dinv[j] : P[p];   // k_unscale_data's order
    for (int p = So.Ap[j]; p < So.Ap[j + 1]; p++) An[p] = sc ? (A[p] * Einv[So.Ai[p]]) * Dinv[j] : A[p];
  }
}

__global__ __launch_bounds__(WAVE) void k_horizon_state(rldl_dev_sym So, rldl_dev_admm Wo, int n_new, int m_new, int n_keep, int m_keep,
                                                        int nt, double *__restrict__ xs, double *__restrict__ ys) {
  const int inst = blockIdx.x, lane = threadIdx.x, n = So.n, m = So.m;
  const double *x = Wo.x + (size_t)inst * n, *y = Wo.y + (size_t)inst * m;
  const bool sc = Wo.scaling != 0;
  const double *D = sc ? Wo.sD + (size_t)inst * n : nullptr, *E = sc ? Wo.sE + (size_t)inst * m : nullptr;
  const double cinv = sc ? Wo.scinv[inst] : 1.0;
  for (int i = lane; i < n_new; i += WAVE) xs[(size_t)inst * n_new + i] = i < n_keep ? (sc ? D[i] * x[i] : x[i]) : 0.0;
  for (int i = lane; i < m_new; i += WAVE) {
    const int src = i < m_keep ? i : (i >= m_new - nt ? m - nt + (i - (m_new - nt)) : -1);
    ys[(size_t)inst * m_new + i] = src < 0 ? 0.0 : (sc ? (E[src] * y[src]) * cinv : y[src]);    // store_solution's unscaling (auxil.c:527-565)
  }
}

__global__ __launch_bounds__(WAVE) void k_horizon_adopt(rldl_dev_sym So, rldl_dev_num No, rldl_dev_sym Sn, rldl_dev_num Nn,
                                                        const double *__restrict__ rvo, const double *__restrict__ rvn, int m_keep, int c0,
                                                        int b_pivot, int ti_prefix, int *__restrict__ b0v, int *__restrict__ n_reused) {
  const int inst = blockIdx.x, lane = threadIdx.x;
  const double *ro = rvo + (size_t)inst * So.m, *rn = rvn + (size_t)inst * Sn.m;
  int differ = No.status[inst] < 0 ? 1 : 0;                      // an old factor with a zero pivot is not adopted
  for (int i = lane; i < m_keep; i += WAVE) differ |= ro[i] != rn[i] ? 1 : 0;
  differ = wave_any(differ);
  if (lane == 0) { b0v[inst] = differ ? 0 : b_pivot; if (!differ) atomicAdd(&n_reused[inst & (RLDL_NACT_SLOTS - 1)], 1); }   // (slots: see rldl_dev_admm.n_active)
  if (differ) return;
  const double *Fo = No.F + (size_t)inst * So.ldF, *Do = No.D + (size_t)inst * So.N;
  double *Fn = Nn.F + (size_t)inst * Sn.ldF, *Dn = Nn.D + (size_t)inst * Sn.N;
  const int nl = So.Lp[c0];                                      // the two L patterns agree on columns < c0 (host check)
  for (int p = lane; p < nl; p += WAVE) Fn[Sn.LtoS[p]] = Fo[So.LtoS[p]];
  for (int j = lane; j < c0; j += WAVE) { Dn[j] = Do[j]; Fn[Sn.nS + j] = Fo[So.nS + j]; }
  if (ti_prefix > 0) {                                           // the product tri-solve's tiles of the adopted blocks (same order in both handles: host check)
    const double *To = No.Ti + (size_t)inst * So.stage.pv_ldTi;
    double *Tn = Nn.Ti + (size_t)inst * Sn.stage.pv_ldTi;
    for (int p = lane; p < ti_prefix; p += WAVE) Tn[p] = To[p];
  }
}

}  // namespace

// ---- single-store horizon change (rldl_horizon.c): ONE workspace at Nmax dimensions for every horizon; the stages beyond the
// current horizon are decoupled dummies (zero values, free rows), so only VALUES change and nothing is copied between workspaces ----
// dst[b][start + i] = src_row[start + i], i < cnt: a column range of one nominal row to every instance
__global__ __launch_bounds__(256) void k_bcast_range(int ld, int start, int cnt, double *dst, const double *__restrict__ src_row) {
  const int inst = blockIdx.x;
  for (int i = threadIdx.x; i < cnt; i += blockDim.x) dst[(size_t)inst * ld + start + i] = src_row[start + i];
}
// dst[b][map[i]] = src[b][i], i < cnt: values given in one horizon's pattern order into the Nmax (union) pattern
__global__ __launch_bounds__(256) void k_scatter_rows(int cnt, int ld, const int *__restrict__ map, const double *__restrict__ src, double *dst) {
  const int inst = blockIdx.x;
  for (int i = threadIdx.x; i < cnt; i += blockDim.x) dst[(size_t)inst * ld + map[i]] = src[(size_t)inst * cnt + i];
}
// q, l, u of a horizon (packed [n_act] / [m_act] rows) into the Nmax-sized rows: linear cost 0 and free rows behind them
__global__ __launch_bounds__(256) void k_horizon_vectors(int n_act, int n_max, int m_act, int m_max, const double *__restrict__ qs,
                                                         const double *__restrict__ ls, const double *__restrict__ us, double *q, double *l, double *u) {
  const int inst = blockIdx.x;
  for (int i = threadIdx.x; i < n_max; i += blockDim.x) q[(size_t)inst * n_max + i] = i < n_act ? qs[(size_t)inst * n_act + i] : 0.0;
  for (int i = threadIdx.x; i < m_max; i += blockDim.x) {
    l[(size_t)inst * m_max + i] = i < m_act ? ls[(size_t)inst * m_act + i] : -OSQP_INFTY;
    u[(size_t)inst * m_max + i] = i < m_act ? us[(size_t)inst * m_act + i] : OSQP_INFTY;
  }
}
// set_rho_vec (auxil.c:79-101) for the new bounds + the verdict of the horizon change per instance: rho_vec of the rows the two
// horizons share is part of the shared factor columns, so an instance whose rho_vec changes there (a row changed its type) or
// whose old factor has a zero pivot restarts at block 0, the others at the pivot block
__global__ __launch_bounds__(WAVE) void k_horizon_rho(rldl_dev_sym S, rldl_dev_admm W, const int *__restrict__ fstatus, int m_keep, int b_pivot,
                                                      int *__restrict__ b0v, int *__restrict__ n_reused) {
  const int inst = blockIdx.x, lane = threadIdx.x, m = S.m;
  const double *l = W.l + (size_t)inst * m, *u = W.u + (size_t)inst * m;
  double *rv = W.rho_vec + (size_t)inst * m;
  int *ct = W.constr_type + (size_t)inst * m;
  const double rho = W.rho_cur[inst];
  int differ = fstatus[inst] < 0 ? 1 : 0;
  for (int i = lane; i < m; i += WAVE) {
    int t;
    double r;
    if (l[i] < -OSQP_INFTY * MIN_SCALING && u[i] > OSQP_INFTY * MIN_SCALING) { t = -1; r = RHO_MIN; }
    else if (u[i] - l[i] < RHO_TOL) { t = 1; r = RHO_EQ_OVER_RHO_INEQ * rho; }
    else { t = 0; r = rho; }
    if (i < m_keep && rv[i] != r) differ = 1;
    ct[i] = t; rv[i] = r;
  }
  differ = wave_any(differ);
  if (lane == 0) {
    W.refactor[inst] = 0;
    b0v[inst] = differ ? 0 : b_pivot;
    if (!differ) atomicAdd(&n_reused[inst & (RLDL_NACT_SLOTS - 1)], 1);
  }
}
// the iterates across a horizon change, in place: x keeps its first n_keep entries, y the rows of the shared row blocks, the
// terminal multipliers move from the old terminal rows to the new ones, everything else starts at zero (z = A x follows)
__global__ __launch_bounds__(WAVE) void k_horizon_state_single(rldl_dev_admm W, int n_max, int m_max, int n_keep, int m_keep, int term_old,
                                                               int term_new, int nt) {
  const int inst = blockIdx.x, lane = threadIdx.x;
  double *x = W.x + (size_t)inst * n_max, *y = W.y + (size_t)inst * m_max;
  double tv = 0.0;
  if (lane < nt) tv = y[term_old + lane];                         // (nt <= 64: checked by the host)
  wave_sync();
  for (int i = n_keep + lane; i < n_max; i += WAVE) x[i] = 0.0;
  for (int i = m_keep + lane; i < m_max; i += WAVE) y[i] = 0.0;
  wave_sync();
  if (lane < nt) y[term_new + lane] = tv;
}

// ================================================================================================
// extern "C" launchers (enqueue only)
// ================================================================================================
static inline int launch_status() { return hipGetLastError() == hipSuccess ? 0 : -1; }

// ---- plan kernels: geometry ----
#define LDS_PER_CU (160 * 1024)
static int plan_per_wave_doubles(const rldl_dev_sym *S, bool stage) { return (stage ? S->ldF : 0) + ((S->N + 1) & ~1); }
static size_t plan_lds_bytes(const rldl_dev_sym *S, int wpb, bool stage) {
  return sizeof(double) * (size_t)wpb * (size_t)plan_per_wave_doubles(S, stage) + sizeof(int) * (size_t)((S->plan_words + 3) & ~3);
}
// waves per workgroup (1..16) that maximise resident waves per CU; asks the runtime what actually fits
// (the usable LDS per CU is below the nominal 160 KiB).  RLDL_WPB forces a value for experiments.
static int plan_pick_wpb_for(const rldl_dev_sym *S, const void *kernel, bool stage, int *waves_out) {
  int best = 0, best_waves = 0;
  const char *force = getenv("RLDL_WPB");
  for (int wpb = 1; wpb <= 16; wpb *= 2) {
    const size_t b = plan_lds_bytes(S, wpb, stage);
    if (b > (size_t)LDS_PER_CU) continue;
    if (force && atoi(force) != wpb) continue;
    if (b > 64 * 1024 && hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)b) != hipSuccess) continue;
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, wpb * WAVE, b) != hipSuccess) continue;
    const int waves = nb * wpb;
    if (waves > best_waves) { best = wpb; best_waves = waves; }
  }
  (void)hipGetLastError();
  if (waves_out) *waves_out = best_waves;
  return best;
}
// staged (factor row in LDS) when that still leaves >= 4 waves per CU, streamed from global memory otherwise
struct PlanGeom { int wpb; bool stage; size_t lds; };
static PlanGeom plan_geometry(const rldl_dev_sym *S, const void *k_staged, const void *k_stream) {
  static thread_local const void *ck = 0; static thread_local int cw = -1, cl = -1, cn = -1; static thread_local PlanGeom cg = {0, false, 0};
  if (ck == k_staged && cw == S->plan_words && cl == S->ldF && cn == S->N) return cg;
  int ws = 0, wg = 0;
  const int wpb_s = plan_pick_wpb_for(S, k_staged, true, &ws), wpb_g = plan_pick_wpb_for(S, k_stream, false, &wg);
  PlanGeom g;
  if (wpb_s > 0 && (ws >= 4 || wpb_g <= 0) && !getenv("RLDL_NO_STAGE")) { g.wpb = wpb_s; g.stage = true; }
  else { g.wpb = wpb_g; g.stage = false; }
  g.lds = g.wpb > 0 ? plan_lds_bytes(S, g.wpb, g.stage) : 0;
  if (getenv("RLDL_VERBOSE"))
    fprintf(stderr, "[rldl] plan kernel: %s, wpb=%d, %d waves/CU, %zu B LDS per workgroup\n", g.stage ? "factor staged in LDS" : "factor read from global",
            g.wpb, g.stage ? ws : wg, g.lds);
  ck = k_staged; cw = S->plan_words; cl = S->ldF; cn = S->N; cg = g;
  return g;
}
static int plan_pick_wpb(const rldl_dev_sym *S) {   // feasibility only: the global-memory variant needs plan + x in LDS
  for (int wpb = 1; wpb <= 16; wpb *= 2)
    if (plan_lds_bytes(S, wpb, false) <= (size_t)LDS_PER_CU) return wpb;
  return 0;
}
static bool plan_usable(const rldl_dev_sym *S) { return S->plan_ok && S->ngroups > 0 && plan_pick_wpb(S) > 0; }
static bool plan_admm_usable(const rldl_dev_sym *S) { return plan_usable(S); }

// ---- block tri-solve on stage patterns: LDS per wave = x (N, even) + one tile (SM rows of SM + 1) ----
static int blk_per_wave_doubles(const rldl_dev_sym *S) { return ((S->N + 1) & ~1) + (S->stage.sv_ld - 1) * S->stage.sv_ld; }
static int blk_pick_wpb(const rldl_dev_sym *S) {
  const size_t pw = sizeof(double) * (size_t)blk_per_wave_doubles(S);
  for (int wpb = 4; wpb >= 1; wpb >>= 1)
    if (pw * wpb <= 64 * 1024) return wpb;
  return 0;
}
static bool blk_usable(const rldl_dev_sym *S) {
  return S->stage.nb > 0 && S->stage.sv_ok && !S->polish && blk_pick_wpb(S) > 0 && !getenv("RLDL_NO_STAGE_SOLVE") && !getenv("RLDL_NO_STAGE_FACTOR");
}
#define BLK_DISPATCH(KERNEL, ...)                                                                                              \
  switch (S->stage.sv_ld - 1) {                                                                                                \
    case 8: hipLaunchKernelGGL((KERNEL<false, 8>), dim3(grid), dim3(wpb * WAVE), lds, (hipStream_t)stream, __VA_ARGS__); break;   \
    case 16: hipLaunchKernelGGL((KERNEL<false, 16>), dim3(grid), dim3(wpb * WAVE), lds, (hipStream_t)stream, __VA_ARGS__); break; \
    case 22: hipLaunchKernelGGL((KERNEL<false, 22>), dim3(grid), dim3(wpb * WAVE), lds, (hipStream_t)stream, __VA_ARGS__); break; \
    case 24: hipLaunchKernelGGL((KERNEL<false, 24>), dim3(grid), dim3(wpb * WAVE), lds, (hipStream_t)stream, __VA_ARGS__); break; \
    case 32: hipLaunchKernelGGL((KERNEL<false, 32>), dim3(grid), dim3(wpb * WAVE), lds, (hipStream_t)stream, __VA_ARGS__); break; \
    default: return -1;                                                                                                        \
  }
// product form (stage_prod_solve): Ti holds the tiles, x is the only LDS array
static bool prod_usable(const rldl_dev_sym *S, const rldl_dev_num *Nn) {
  static const int off = getenv("RLDL_NO_STAGE_PROD") ? 1 : 0;
  return !off && S->stage.pv_ok && Nn->Ti;
}
#define PROD_WPB 4
static int prod_per_wave_doubles(const rldl_dev_sym *S) { return ((S->N + 1) & ~1) + 2 + 34; }   // x, its spare words, the auxiliary block vector of mode 2
// dynamic LDS per workgroup of the product kernels: what the waves need, or RLDL_PROD_LDS bytes when that is more (a diagnostic:
// fewer resident workgroups per CU)
static size_t prod_lds_bytes(const rldl_dev_sym *S, const void *kernel) {
  static const long forced = getenv("RLDL_PROD_LDS") ? atol(getenv("RLDL_PROD_LDS")) : 0;
  size_t lds = sizeof(double) * (size_t)prod_per_wave_doubles(S) * PROD_WPB;
  if (forced > 0 && (size_t)forced > lds) lds = (size_t)forced;
  if (lds > 64 * 1024) (void)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  return lds;
}
static int launch_blk_solve(const rldl_dev_sym *S, const rldl_dev_num *Nn, double *d_b, void *stream) {
  if (prod_usable(S, Nn)) {
    const int pw = prod_per_wave_doubles(S), grid = (Nn->batch + PROD_WPB - 1) / PROD_WPB;
    hipLaunchKernelGGL((k_plan_solve<false, 1, true>), dim3(grid), dim3(PROD_WPB * WAVE), prod_lds_bytes(S, (const void *)k_plan_solve<false, 1, true>),
                       (hipStream_t)stream, *S, *Nn, d_b, pw);
    return launch_status();
  }
  const int wpb = blk_pick_wpb(S), pw = blk_per_wave_doubles(S), grid = (Nn->batch + wpb - 1) / wpb;
  const size_t lds = sizeof(double) * (size_t)pw * wpb;
  BLK_DISPATCH(k_plan_solve, *S, *Nn, d_b, pw)
  return launch_status();
}
static int launch_blk_admm(const rldl_dev_sym *S, const rldl_dev_num *Nn, const rldl_dev_admm *W, void *stream, int iters = 1) {
  if (prod_usable(S, Nn)) {
    const int pw = prod_per_wave_doubles(S), grid = (Nn->batch + PROD_WPB - 1) / PROD_WPB;
    hipLaunchKernelGGL((k_plan_admm_loop<false, 1, true>), dim3(grid), dim3(PROD_WPB * WAVE), prod_lds_bytes(S, (const void *)k_plan_admm_loop<false, 1, true>),
                       (hipStream_t)stream, *S, *Nn, *W, pw, iters);
    return launch_status();
  }
  if (iters != 1) return -1;
  const int wpb = blk_pick_wpb(S), pw = blk_per_wave_doubles(S), grid = (Nn->batch + wpb - 1) / wpb;
  const size_t lds = sizeof(double) * (size_t)pw * wpb;
  BLK_DISPATCH(k_plan_admm_loop, *S, *Nn, *W, pw, 1)
  return launch_status();
}

static int launch_plan_solve(const rldl_dev_sym *S, const rldl_dev_num *Nn, double *d_b, void *stream) {
  const PlanGeom g = plan_geometry(S, (const void *)k_plan_solve<true>, (const void *)k_plan_solve<false>);
  if (g.wpb <= 0) return -1;
  const void *k = g.stage ? (const void *)k_plan_solve<true> : (const void *)k_plan_solve<false>;
  if (g.lds > 64 * 1024 && hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)g.lds) != hipSuccess) return -1;
  const int grid = (Nn->batch + g.wpb - 1) / g.wpb, pw = plan_per_wave_doubles(S, g.stage);
  if (g.stage) hipLaunchKernelGGL(k_plan_solve<true>, dim3(grid), dim3(g.wpb * WAVE), g.lds, (hipStream_t)stream, *S, *Nn, d_b, pw);
  else hipLaunchKernelGGL(k_plan_solve<false>, dim3(grid), dim3(g.wpb * WAVE), g.lds, (hipStream_t)stream, *S, *Nn, d_b, pw);
  return launch_status();
}
template <int TMAX>
static int launch_plan_admm_t(const rldl_dev_sym *S, const rldl_dev_num *Nn, const rldl_dev_admm *W, void *stream) {
  const PlanGeom g = plan_geometry(S, (const void *)k_plan_admm<TMAX, true>, (const void *)k_plan_admm<TMAX, false>);
  if (g.wpb <= 0) return -1;
  const void *k = g.stage ? (const void *)k_plan_admm<TMAX, true> : (const void *)k_plan_admm<TMAX, false>;
  if (g.lds > 64 * 1024 && hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)g.lds) != hipSuccess) return -1;
  const int grid = (Nn->batch + g.wpb - 1) / g.wpb, pw = plan_per_wave_doubles(S, g.stage);
  if (g.stage) hipLaunchKernelGGL((k_plan_admm<TMAX, true>), dim3(grid), dim3(g.wpb * WAVE), g.lds, (hipStream_t)stream, *S, *Nn, *W, pw);
  else hipLaunchKernelGGL((k_plan_admm<TMAX, false>), dim3(grid), dim3(g.wpb * WAVE), g.lds, (hipStream_t)stream, *S, *Nn, *W, pw);
  return launch_status();
}
static int launch_plan_admm_loop(const rldl_dev_sym *S, const rldl_dev_num *Nn, const rldl_dev_admm *W, void *stream, int iters = 1) {
  const PlanGeom g = plan_geometry(S, (const void *)k_plan_admm_loop<true>, (const void *)k_plan_admm_loop<false>);
  if (g.wpb <= 0) return -1;
  const void *k = g.stage ? (const void *)k_plan_admm_loop<true> : (const void *)k_plan_admm_loop<false>;
  if (g.lds > 64 * 1024 && hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)g.lds) != hipSuccess) return -1;
  const int grid = (Nn->batch + g.wpb - 1) / g.wpb, pw = plan_per_wave_doubles(S, g.stage);
  if (g.stage) hipLaunchKernelGGL(k_plan_admm_loop<true>, dim3(grid), dim3(g.wpb * WAVE), g.lds, (hipStream_t)stream, *S, *Nn, *W, pw, iters);
  else hipLaunchKernelGGL(k_plan_admm_loop<false>, dim3(grid), dim3(g.wpb * WAVE), g.lds, (hipStream_t)stream, *S, *Nn, *W, pw, iters);
  return launch_status();
}
static int launch_plan_admm(const rldl_dev_sym *S, const rldl_dev_num *Nn, const rldl_dev_admm *W, void *stream) {
  if (S->N <= 2 * WAVE) return launch_plan_admm_t<2>(S, Nn, W, stream);
  if (S->N <= 4 * WAVE) return launch_plan_admm_t<4>(S, Nn, W, stream);
  if (S->N <= 8 * WAVE) return launch_plan_admm_t<8>(S, Nn, W, stream);
  return launch_plan_admm_loop(S, Nn, W, stream);
}

// ---- arrowhead kernels: geometry ----
// Per wave: staging buffer (coupling values, then the LDS part of the triangle; even length) followed by x.
// rr = triangle rows kept in registers: the smallest count (<= ARROW_RR_MAX) that lets 16 waves share a CU's LDS,
// else 0.  RLDL_ARROW_RR overrides (timing experiments).
static ArrowGeom arrow_geometry_rr(const rldl_dev_sym *S, int rr) {
  const int g = S->arrow_tb >= 0 ? S->arrow_g : 1;
  rr = rr < 0 ? 0 : (rr > ARROW_RR_MAX ? ARROW_RR_MAX : rr);
  if (rr > g - 1) rr = g - 1 > 0 ? g - 1 : 0;
  const int gp = g - rr, tri = (gp * (gp - 1)) / 2, trip = (tri + 1) & ~1;
  const int stage = trip > S->nOp ? trip : S->nOp;
  ArrowGeom G;
  G.rr = rr; G.tri2 = trip >> 1; G.xoff = stage; G.per_wave = stage + ((S->N + 1) & ~1);
  return G;
}
static ArrowGeom arrow_geometry(const rldl_dev_sym *S) {
  const char *force = getenv("RLDL_ARROW_RR");
  if (force) return arrow_geometry_rr(S, atoi(force));
  for (int rr = 0; rr <= ARROW_RR_MAX; rr++) {
    const ArrowGeom G = arrow_geometry_rr(S, rr);
    if (sizeof(double) * (size_t)G.per_wave * 16 <= (size_t)LDS_PER_CU) return G;
  }
  return arrow_geometry_rr(S, 0);
}
static int arrow_pick_wpb(const rldl_dev_sym *S, const void *kernel, const ArrowGeom &G, size_t *lds_out) {
  static thread_local const void *ck = 0; static thread_local int cpw = -1, cbest = 0; static thread_local size_t clds = 0;
  if (ck == kernel && cpw == G.per_wave) { *lds_out = clds; return cbest; }
  int best = 0, best_waves = 0; size_t best_lds = 0;
  const char *force = getenv("RLDL_WPB");
  for (int wpb = 1; wpb <= 4; wpb *= 2) {
    const size_t b = sizeof(double) * (size_t)wpb * (size_t)G.per_wave;
    if (b > (size_t)LDS_PER_CU) continue;
    if (force && atoi(force) != wpb) continue;
    if (b > 64 * 1024 && hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)b) != hipSuccess) continue;
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, wpb * WAVE, b) != hipSuccess) continue;
    if (nb * wpb > best_waves) { best = wpb; best_waves = nb * wpb; best_lds = b; }
  }
  (void)hipGetLastError();
  ck = kernel; cpw = G.per_wave; cbest = best; clds = best_lds;
  if (getenv("RLDL_VERBOSE"))
    fprintf(stderr, "[rldl] arrow kernel: wpb=%d, %d waves/CU, %zu B LDS per workgroup, %d triangle rows in registers\n", best, best_waves,
            best_lds, G.rr);
  *lds_out = best_lds;
  return best;
}
static bool arrow_usable(const rldl_dev_sym *S) {
  return S->plan_ok && S->arrow_ok && S->arrow_vsteps >= 1 && S->arrow_vsteps <= 24 && S->arrow_tb <= 0 && S->arrow_g <= WAVE && S->N <= 8 * WAVE &&
         !getenv("RLDL_NO_ARROW") && sizeof(double) * (size_t)arrow_geometry(S).per_wave <= (size_t)LDS_PER_CU;
}
template <int TMAX, int TG>
static int launch_arrow_solve_t(const rldl_dev_sym *S, const rldl_dev_num *Nn, double *d_b, void *stream) {
  size_t lds = 0;
  const ArrowGeom G = arrow_geometry(S);
  const int wpb = arrow_pick_wpb(S, (const void *)k_arrow_solve<TMAX, TG>, G, &lds);
  if (wpb <= 0) return -1;
  if (lds > 64 * 1024 && hipFuncSetAttribute((const void *)k_arrow_solve<TMAX, TG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -1;
  hipLaunchKernelGGL((k_arrow_solve<TMAX, TG>), dim3((Nn->batch + wpb - 1) / wpb), dim3(wpb * WAVE), lds, (hipStream_t)stream, *S, *Nn, d_b, G);
  return launch_status();
}
template <int TMAX>
static int launch_arrow_solve_g(const rldl_dev_sym *S, const rldl_dev_num *Nn, double *d_b, void *stream) {
  // (the step counts the host pads the virtual-row tables to: the edge colouring may use every padded step)
  if (S->arrow_vsteps <= 12) return launch_arrow_solve_t<TMAX, 12>(S, Nn, d_b, stream);
  if (S->arrow_vsteps <= 18) return launch_arrow_solve_t<TMAX, 18>(S, Nn, d_b, stream);
  return launch_arrow_solve_t<TMAX, 24>(S, Nn, d_b, stream);
}
static int launch_arrow_solve(const rldl_dev_sym *S, const rldl_dev_num *Nn, double *d_b, void *stream) {
  if (S->N <= 2 * WAVE) return launch_arrow_solve_g<2>(S, Nn, d_b, stream);
  if (S->N <= 3 * WAVE) return launch_arrow_solve_g<3>(S, Nn, d_b, stream);
  if (S->N <= 4 * WAVE) return launch_arrow_solve_g<4>(S, Nn, d_b, stream);
  return launch_arrow_solve_g<8>(S, Nn, d_b, stream);
}
static bool tile_admm_usable(const rldl_dev_sym *S, const rldl_dev_num *Nn, const rldl_dev_admm *W);
static int launch_tile_admm(const rldl_dev_sym *S, const rldl_dev_num *Nn, const rldl_dev_admm *W, int iters, void *stream,
                            const rldl_dev_multi *M = nullptr, int multi_grid = 0, int multi_xdw = 0);
template <int TMAX, int TG>
static int launch_arrow_admm_t(const rldl_dev_sym *S, const rldl_dev_num *Nn, const rldl_dev_admm *W, int iters, void *stream) {
  size_t lds = 0;
  const ArrowGeom G = arrow_geometry(S);
  const int wpb = arrow_pick_wpb(S, (const void *)k_arrow_admm<TMAX, TG, false>, G, &lds);
  if (wpb <= 0) return -1;
  if (lds > 64 * 1024 && hipFuncSetAttribute(W->trace ? (const void *)k_arrow_admm<TMAX, TG, true> : (const void *)k_arrow_admm<TMAX, TG, false>,
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -1;
  if (W->trace)
    hipLaunchKernelGGL((k_arrow_admm<TMAX, TG, true>), dim3((Nn->batch + wpb - 1) / wpb), dim3(wpb * WAVE), lds, (hipStream_t)stream, *S, *Nn, *W, G, iters);
  else
    hipLaunchKernelGGL((k_arrow_admm<TMAX, TG, false>), dim3((Nn->batch + wpb - 1) / wpb), dim3(wpb * WAVE), lds, (hipStream_t)stream, *S, *Nn, *W, G, iters);
  return launch_status();
}
template <int TMAX>
static int launch_arrow_admm_g(const rldl_dev_sym *S, const rldl_dev_num *Nn, const rldl_dev_admm *W, int iters, void *stream) {
  if (S->arrow_vsteps <= 12) return launch_arrow_admm_t<TMAX, 12>(S, Nn, W, iters, stream);
  if (S->arrow_vsteps <= 18) return launch_arrow_admm_t<TMAX, 18>(S, Nn, W, iters, stream);
  return launch_arrow_admm_t<TMAX, 24>(S, Nn, W, iters, stream);
}
static int launch_arrow_admm(const rldl_dev_sym *S, const rldl_dev_num *Nn, const rldl_dev_admm *W, int iters, void *stream) {
  if (tile_admm_usable(S, Nn, W)) return launch_tile_admm(S, Nn, W, iters, stream);
  if (S->N <= 2 * WAVE) return launch_arrow_admm_g<2>(S, Nn, W, iters, stream);
  if (S->N <= 3 * WAVE) return launch_arrow_admm_g<3>(S, Nn, W, iters, stream);
  if (S->N <= 4 * WAVE) return launch_arrow_admm_g<4>(S, Nn, W, iters, stream);
  return launch_arrow_admm_g<8>(S, Nn, W, iters, stream);
}

// ---- tile kernels: geometry and dispatch ----
// LDS per wave: x, long enough for the padding rows of the last block row (g0 + ta * tq >= N when the tail ends the matrix)
#define TILE_WPB 4
static int tile_per_wave(const rldl_dev_sym *S) {
  const int need = S->arrow_g0 + S->tile_ta * S->tile_tq;
  return ((need > S->N ? need : S->N) + 1) & ~1;
}
static bool tile_usable(const rldl_dev_sym *S, const rldl_dev_num *Nn) {
  static const int off = getenv("RLDL_NO_TILE") ? 1 : 0;
  return !off && S->tile_ok && Nn->Ti && S->arrow_tb == 0 && S->arrow_vsteps <= 24 && S->N <= 3 * WAVE &&
         sizeof(double) * (size_t)(tile_per_wave(S) + WAVE + 5 * 3 * WAVE) * TILE_WPB < 65536;
}
// fused ADMM iterations: also needs the slot table (at most 3 slots of variables-only / constraints-only positions); the wave
// timeline (W->trace) exists for the metric shape's instantiation only
// owner-gather variant: its steps fit a compiled split and the register file
static bool tile_admm_gather(const rldl_dev_sym *S) {
  static const int off = getenv("RLDL_TILE_SCATTER") ? 1 : 0;      // diagnostic: every pattern on the scatter variant
  const int tg = S->arrow_vsteps <= 12 ? 12 : S->arrow_vsteps <= 18 ? 18 : 24;
  return !off && S->tile_admm_ok && 2 * tg + 2 * S->tile_ta * S->tile_ta + (26 * S->tile_tk) / 10 + 89 <= 256;
}
static bool tile_admm_usable(const rldl_dev_sym *S, const rldl_dev_num *Nn, const rldl_dev_admm *W) {
  // register need of the instantiation the launcher would pick (measured fit: 2 TG + 2 TA^2 + 2.6 TK + 89): beyond 256 the
  // kernel would spill inside its loop -- those shapes stay on the sweep kernels
  if (!tile_usable(S, Nn)) return false;
  if (tile_admm_gather(S)) return !W->trace || (S->arrow_vsteps > 12 && S->arrow_vsteps <= 18 && S->tile_ta == 5 && S->tile_tk == 24 && S->tile_sp == 18);
  return S->tile_scatter_ok && !W->trace;                          // the scatter variant (k_tile_admm<.., TK = 0>)
}
#define TILE_TA_SWITCH(CALL)                  \
  switch (S->tile_ta) {                       \
    case 2: CALL(2); break;                   \
    case 3: CALL(3); break;                   \
    case 5: CALL(5); break;                   \
    case 7: CALL(7); break;                   \
    default: return -1;                       \
  }
// round-3 solve kernel (k_tile_solve3): coupling values staged in LDS (cwp doubles per wave, whole 64-lane DMA pieces); every LDS
// byte address of the workgroup must fit 16 bits
static int tile_solve3_cwp(const rldl_dev_sym *S) { return ((S->nOp + 127) / 128) * 128; }
static bool tile_solve3_usable(const rldl_dev_sym *S) {
  static const int off = getenv("RLDL_SOLVE_V2") ? 1 : 0;
  return !off && S->po_tmask > 0 && S->po_spack > 0 && sizeof(double) * (size_t)(tile_solve3_cwp(S) + tile_per_wave(S) + WAVE) * TILE_WPB < 65536;
}
static int launch_tile_solve3(const rldl_dev_sym *S, const rldl_dev_num *Nn, double *d_b, long long *d_trace, void *stream) {
  const int pw = tile_per_wave(S), cwp = tile_solve3_cwp(S), grid = (Nn->batch + TILE_WPB - 1) / TILE_WPB;
  const size_t lds = sizeof(double) * (size_t)(cwp + pw + WAVE) * TILE_WPB;
#define TS3(TG, TA) do { if (d_trace) hipLaunchKernelGGL((k_tile_solve3<3, TG, TA, true, false>), dim3(grid), dim3(TILE_WPB * WAVE), lds, (hipStream_t)stream, *S, *Nn, d_b, pw, cwp, d_trace); \
    else if (Nn->nt_loads) hipLaunchKernelGGL((k_tile_solve3<3, TG, TA, false, true>), dim3(grid), dim3(TILE_WPB * WAVE), lds, (hipStream_t)stream, *S, *Nn, d_b, pw, cwp, d_trace); \
    else hipLaunchKernelGGL((k_tile_solve3<3, TG, TA, false, false>), dim3(grid), dim3(TILE_WPB * WAVE), lds, (hipStream_t)stream, *S, *Nn, d_b, pw, cwp, d_trace); } while (0)
  if (S->arrow_vsteps <= 12) {
#define C(TA) TS3(12, TA)
    TILE_TA_SWITCH(C)
#undef C
  } else if (S->arrow_vsteps <= 18) {
#define C(TA) TS3(18, TA)
    TILE_TA_SWITCH(C)
#undef C
  } else {
#define C(TA) TS3(24, TA)
    TILE_TA_SWITCH(C)
#undef C
  }
#undef TS3
  return launch_status();
}
static int launch_tile_solve(const rldl_dev_sym *S, const rldl_dev_num *Nn, double *d_b, void *stream) {
  if (tile_solve3_usable(S)) return launch_tile_solve3(S, Nn, d_b, nullptr, stream);
  const int pw = tile_per_wave(S), grid = (Nn->batch + TILE_WPB - 1) / TILE_WPB;
  static const size_t pad = getenv("RLDL_SOLVE_LDS_PAD") ? (size_t)atol(getenv("RLDL_SOLVE_LDS_PAD")) : 0;   // occupancy experiments: LDS bytes per workgroup
  const size_t lds0 = sizeof(double) * (size_t)(pw + WAVE) * TILE_WPB, lds = pad > lds0 ? pad : lds0;
#define TS(TMAX, TG, TA) do { if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void *)k_tile_solve<TMAX, TG, TA>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    hipLaunchKernelGGL((k_tile_solve<TMAX, TG, TA>), dim3(grid), dim3(TILE_WPB * WAVE), lds, (hipStream_t)stream, *S, *Nn, d_b, pw); } while (0)
  if (S->arrow_vsteps <= 12) {
#define C(TA) TS(3, 12, TA)
    TILE_TA_SWITCH(C)
#undef C
  } else if (S->arrow_vsteps <= 18) {
#define C(TA) TS(3, 18, TA)
    TILE_TA_SWITCH(C)
#undef C
  } else {
#define C(TA) TS(3, 24, TA)
    TILE_TA_SWITCH(C)
#undef C
  }
#undef TS
  return launch_status();
}
// M: several workspaces in one launch (S / Nn / W then only select the instantiation: every group must have the same rldl_multi_key);
// multi_grid = workgroups of all groups, multi_xdw = the largest per-wave LDS need among them
static const rldl_dev_multi k_no_multi = {};
static int launch_tile_admm(const rldl_dev_sym *S, const rldl_dev_num *Nn, const rldl_dev_admm *W, int iters, void *stream,
                            const rldl_dev_multi *M, int multi_grid, int multi_xdw) {
  const int pw = tile_per_wave(S), grid = M ? multi_grid : (Nn->batch + TILE_WPB - 1) / TILE_WPB;
  const size_t lds = sizeof(double) * (size_t)((M ? multi_xdw : pw) + WAVE + 4 * TILE_SLOTS * WAVE) * TILE_WPB;
  const rldl_dev_multi &MM = M ? *M : k_no_multi;
  if (W->trace) {                                                 // the wave timeline exists for the metric shape's instantiation only
    if (M) return -1;
    hipLaunchKernelGGL((k_tile_admm<18, 5, 24, 18, true>), dim3(grid), dim3(TILE_WPB * WAVE), lds, (hipStream_t)stream, *S, *Nn, *W, pw, iters, MM);
    return launch_status();
  }
  int launched = 0;
  // one instantiation per (virtual-row steps, tile size, owner-gather steps, split); combinations whose register need exceeds
  // 256 (tile_admm_usable rules them out) are not compiled
#define TA_(TG, TA, TK, SP) do { if constexpr (2 * TG + 2 * TA * TA + (26 * TK) / 10 + 89 <= 256) { \
    if (M) hipLaunchKernelGGL((k_tile_admm<TG, TA, TK, SP, false, true>), dim3(grid), dim3(TILE_WPB * WAVE), lds, (hipStream_t)stream, *S, *Nn, *W, pw, iters, MM); \
    else hipLaunchKernelGGL((k_tile_admm<TG, TA, TK, SP, false>), dim3(grid), dim3(TILE_WPB * WAVE), lds, (hipStream_t)stream, *S, *Nn, *W, pw, iters, MM); launched = 1; } } while (0)
#define TK_(TG, TA) switch (tile_admm_gather(S) ? S->tile_tk * 100 + S->tile_sp : 0) { case 0: TA_(TG, TA, 0, 0); break; case 1612: TA_(TG, TA, 16, 12); break; case 1610: TA_(TG, TA, 16, 10); break; \
    case 2418: TA_(TG, TA, 24, 18); break; case 2416: TA_(TG, TA, 24, 16); break; case 3224: TA_(TG, TA, 32, 24); break; case 3220: TA_(TG, TA, 32, 20); break; default: return -1; }
  if (S->arrow_vsteps <= 12) {
#define C(TA) TK_(12, TA)
    TILE_TA_SWITCH(C)
#undef C
  } else if (S->arrow_vsteps <= 18) {
#define C(TA) TK_(18, TA)
    TILE_TA_SWITCH(C)
#undef C
  } else {
#define C(TA) TK_(24, TA)
    TILE_TA_SWITCH(C)
#undef C
  }
#undef TK_
#undef TA_
  return launched ? launch_status() : -1;
}
// inverse of the tail triangle behind every numeric factorisation of a tile handle (any factor kernel: it reads the factor row)
static int launch_tile_invert(const rldl_dev_sym *S, const rldl_dev_num *Nn, const int *d_mask, void *stream) {
  if (!S->tile_ok || !Nn->Ti || S->arrow_tb != 0) return 0;
  const int g = S->arrow_g;
  const size_t lds = sizeof(double) * (size_t)((g * (g - 1)) / 2 + 2 + 2 * LCH + 64 + 256 * ((g + 15) / 16));
  const dim3 grid(Nn->batch), blk(WAVE);
  if (g <= 16) hipLaunchKernelGGL(k_tile_invert<16>, grid, blk, lds, (hipStream_t)stream, *S, *Nn, d_mask);
  else if (g <= 32) hipLaunchKernelGGL(k_tile_invert<32>, grid, blk, lds, (hipStream_t)stream, *S, *Nn, d_mask);
  else if (g <= 48) hipLaunchKernelGGL(k_tile_invert<48>, grid, blk, lds, (hipStream_t)stream, *S, *Nn, d_mask);
  else if (g <= 56) hipLaunchKernelGGL(k_tile_invert<56>, grid, blk, lds, (hipStream_t)stream, *S, *Nn, d_mask);
  else hipLaunchKernelGGL(k_tile_invert<64>, grid, blk, lds, (hipStream_t)stream, *S, *Nn, d_mask);
  return launch_status();
}

extern "C" int rldl_launch_kkt_assemble(const rldl_dev_sym *S, const rldl_dev_num *Nn, const double *d_Px,
                                        const double *d_Ax, const double *d_rho_vec, int set_sigma_only,
                                        const int *d_mask, void *stream) {
  if (Nn->batch <= 0) return 0;
  hipLaunchKernelGGL(k_kkt_assemble, dim3(Nn->batch), dim3(256), 0, (hipStream_t)stream, *S, *Nn, d_Px, d_Ax, d_rho_vec,
                     set_sigma_only, d_mask, (double *)0, (double *)0, (int *)0, (int *)0);
  return launch_status();
}
// the same scatter that also stores the incoming values in the caller's own arrays (osqp_update_P_A keeps a copy of the data)
extern "C" int rldl_launch_kkt_assemble_keep(const rldl_dev_sym *S, const rldl_dev_num *Nn, const double *d_Px, const double *d_Ax,
                                             double *keepP, double *keepA, int *d_status_reset, int *d_rho_updates_reset, void *stream) {
  if (Nn->batch <= 0) return 0;
  int *sr = d_status_reset && d_rho_updates_reset ? d_status_reset : (int *)0;
  static const int scatter = getenv("RLDL_ASSEMBLE_SCATTER") ? 1 : 0;   // (A/B: the scatter kernel also for full updates)
  if (d_Px && d_Ax && !S->polish && !scatter && sizeof(double) * (size_t)S->nnzK <= 48 * 1024) {   // both value sets: the row is rebuilt in LDS
    hipLaunchKernelGGL(k_kkt_assemble_full, dim3(Nn->batch), dim3(256), sizeof(double) * (size_t)S->nnzK, (hipStream_t)stream, *S, *Nn, d_Px, d_Ax,
                       keepP, keepA, sr, d_rho_updates_reset);
    return launch_status();
  }
  hipLaunchKernelGGL(k_kkt_assemble, dim3(Nn->batch), dim3(256), 0, (hipStream_t)stream, *S, *Nn, d_Px, d_Ax, (const double *)0, 0,
                     (const int *)0, d_Px ? keepP : (double *)0, d_Ax ? keepA : (double *)0, sr, d_rho_updates_reset);
  return launch_status();
}
// start of osqp_solve for the whole batch in one launch: status = OSQP_UNSOLVED, rho_updates = 0 (reset_info, auxil.c:628-645),
// the active-instance counter, and cold_start (auxil.c:158-162) when warm starting is off
__global__ __launch_bounds__(256) void k_solve_begin(rldl_dev_admm W, int n, int m, int cold, int reset_rho_updates, rldl_dev_multi M) {
  int inst = blockIdx.x;
  if (M.ngroups > 0) { const int g = multi_group(M.first_inst, M.ngroups, inst); W = M.W[g]; inst -= M.first_inst[g]; }
  if (threadIdx.x == 0) {
    W.status[inst] = ST_UNSOLVED;
    if (reset_rho_updates) W.rho_updates[inst] = 0;
    if (inst < RLDL_NACT_SLOTS) W.n_active[inst] = (W.batch - inst + RLDL_NACT_SLOTS - 1) / RLDL_NACT_SLOTS;   // instances inst, inst + SLOTS, ...
    for (int s = W.batch + inst; s < RLDL_NACT_SLOTS; s += W.batch) W.n_active[s] = 0;                          // batch < SLOTS
  }
  if (cold) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) W.x[(size_t)inst * n + i] = 0.0;
    for (int i = threadIdx.x; i < m; i += blockDim.x) { W.z[(size_t)inst * m + i] = 0.0; W.y[(size_t)inst * m + i] = 0.0; }
  }
}
extern "C" int rldl_launch_solve_begin(const rldl_dev_admm *W, int n, int m, int cold, int reset_rho_updates, void *stream) {
  if (W->batch <= 0) return 0;
  hipLaunchKernelGGL(k_solve_begin, dim3(W->batch), dim3(256), 0, (hipStream_t)stream, *W, n, m, cold, reset_rho_updates, k_no_multi);
  return launch_status();
}

static int launch_factor(const rldl_dev_sym *S, const rldl_dev_num *Nn, const int *d_mask, int c_start, void *stream, long long *d_trace = nullptr) {
  if (Nn->batch <= 0) return 0;
  if (c_start <= 0 && S->arrow_ok && S->arrow_dense && S->arrow_g <= 64 && !getenv("RLDL_NO_ARROW_FACTOR")) {
    const size_t al = sizeof(double) * (size_t)(S->nnzL + S->N + S->arrow_g0 + 2 + 128);
    if (al <= RLDL_LDS_LIMIT) {
      const dim3 grid(Nn->batch), blk(WAVE);
      const int g = S->arrow_g;
      static const int split = getenv("RLDL_SPLIT_INVERT") ? 1 : 0;            // (A/B: tail inverse as its own launch, as in round 2)
      const size_t il = sizeof(double) * (size_t)((g * (g - 1)) / 2 + 2 + 2 * LCH + 64 + 256 * ((g + 15) / 16));
      if (S->tile_ok && Nn->Ti && S->arrow_tb == 0 && !split) {               // (the conditions of launch_tile_invert)
        const size_t fl = al > il ? al : il;
#define AF(SMV) hipLaunchKernelGGL((k_arrow_factor<SMV, true, true>), grid, blk, fl, (hipStream_t)stream, *S, *Nn, d_mask, d_trace)
        if (g <= 16) AF(16); else if (g <= 32) AF(32); else if (g <= 48) AF(48); else if (g <= 56) AF(56); else AF(64);
#undef AF
        return launch_status();
      }
#define AF(SMV) hipLaunchKernelGGL((k_arrow_factor<SMV, false, true>), grid, blk, al, (hipStream_t)stream, *S, *Nn, d_mask, d_trace)
      if (g <= 16) AF(16); else if (g <= 32) AF(32); else if (g <= 48) AF(48); else if (g <= 56) AF(56); else AF(64);
#undef AF
      if (launch_status()) return -1;
      return launch_tile_invert(S, Nn, d_mask, stream);
    }
  }
  const size_t lds = sizeof(double) * (size_t)(S->nnzL + S->N);
  if (lds <= RLDL_LDS_LIMIT)
    hipLaunchKernelGGL(k_factor<true>, dim3(Nn->batch), dim3(WAVE), lds, (hipStream_t)stream, *S, *Nn, d_mask, c_start);
  else
    hipLaunchKernelGGL(k_factor<false>, dim3(Nn->batch), dim3(WAVE), 0, (hipStream_t)stream, *S, *Nn, d_mask, c_start);
  if (launch_status()) return -1;
  return launch_tile_invert(S, Nn, d_mask, stream);
}

// tiles of the product tri-solve (k_stage_invert) behind a stage factorisation.  Per-instance restart blocks come from a horizon
// change: when the adopted columns did not bring their tiles along (tiles_adopted = 0), every block is redone.
static int launch_stage_invert(const rldl_dev_sym *S, const rldl_dev_num *Nn, const int *d_mask, int first_block, const int *d_b0v,
                               int tiles_adopted, void *stream) {
  const rldl_dev_stage *G = &S->stage;
  if (!G->pv_ok || !Nn->Ti || G->smax > 32) return 0;
  const int sm = G->smax <= 8 ? 8 : G->smax <= 16 ? 16 : G->smax <= 24 ? 24 : 32;
  if (G->pv_ldT != sm + 2) return -1;
  const size_t lds = sizeof(double) * (size_t)(2 * sm * (sm + 2));
  const dim3 grid(Nn->batch), blk(WAVE);
  const int b0 = d_b0v ? 0 : first_block;
  const int *bv = tiles_adopted ? d_b0v : (const int *)0;
  if (sm == 8) hipLaunchKernelGGL(k_stage_invert<8>, grid, blk, lds, (hipStream_t)stream, *S, *Nn, d_mask, b0, bv);
  else if (sm == 16) hipLaunchKernelGGL(k_stage_invert<16>, grid, blk, lds, (hipStream_t)stream, *S, *Nn, d_mask, b0, bv);
  else if (sm == 24) hipLaunchKernelGGL(k_stage_invert<24>, grid, blk, lds, (hipStream_t)stream, *S, *Nn, d_mask, b0, bv);
  else hipLaunchKernelGGL(k_stage_invert<32>, grid, blk, lds, (hipStream_t)stream, *S, *Nn, d_mask, b0, bv);
  return launch_status();
}
static int launch_stage_factor(const rldl_dev_sym *S, const rldl_dev_num *Nn, const int *d_mask, int first_block, const int *d_b0v,
                               int tiles_adopted, void *stream) {
  if (Nn->batch <= 0) return 0;
  const rldl_dev_stage *G = &S->stage;
  if (G->nb <= 0 || first_block < 0 || first_block >= G->nb) return -1;
  if (getenv("RLDL_STAGE_LDS") || G->smax > 32) {                // LDS-resident panel (reference version of the same recursion)
    const size_t lds = sizeof(double) * (size_t)(3 * G->smax * G->ld + 2 * G->ld);
    hipLaunchKernelGGL(k_stage_factor, dim3(Nn->batch), dim3(WAVE), lds, (hipStream_t)stream, *S, *Nn, d_mask, first_block, d_b0v);
    if (launch_status()) return -1;
    return launch_stage_invert(S, Nn, d_mask, first_block, d_b0v, tiles_adopted, stream);
  }
  // matrix-core Schur complement (see k_stage_factor_r) when the result tiles fit the staging tile; RLDL_NO_MFMA=1: fma form
  static const int no_mfma = getenv("RLDL_NO_MFMA") ? 1 : 0;
  const int sm = G->smax <= 8 ? 8 : G->smax <= 16 ? 16 : G->smax <= 24 ? 24 : 32, nt16 = 16 * ((sm + 15) / 16);
  const bool mfma = !no_mfma && nt16 * (nt16 + 1) <= 2 * G->smax * G->ld && G->ld >= sm + 1;
  const int lt_rows = mfma ? (nt16 > G->smax ? nt16 : G->smax) : G->smax;
  const size_t lds = sizeof(double) * (size_t)((2 * G->smax + lt_rows) * G->ld + 2 * G->ld + 130);
  const dim3 grid(Nn->batch), blk(WAVE);
#define SF(SMV) do { if (mfma) hipLaunchKernelGGL((k_stage_factor_r<SMV, true>), grid, blk, lds, (hipStream_t)stream, *S, *Nn, d_mask, first_block, d_b0v, lt_rows); \
                     else hipLaunchKernelGGL((k_stage_factor_r<SMV, false>), grid, blk, lds, (hipStream_t)stream, *S, *Nn, d_mask, first_block, d_b0v, lt_rows); } while (0)
  if (sm == 8) SF(8);
  else if (sm == 16) SF(16);
  else if (sm == 24) SF(24);
  else SF(32);
#undef SF
  if (launch_status()) return -1;
  return launch_stage_invert(S, Nn, d_mask, first_block, d_b0v, tiles_adopted, stream);
}
// tiles of the product tri-solve from the factor as it stands (a handle that becomes stage-structured after its first factorisation)
extern "C" int rldl_launch_stage_invert(const rldl_dev_sym *S, const rldl_dev_num *Nn, void *stream) {
  return Nn->batch > 0 ? launch_stage_invert(S, Nn, 0, 0, 0, 0, stream) : 0;
}
extern "C" int rldl_launch_stage_factor(const rldl_dev_sym *S, const rldl_dev_num *Nn, const int *d_mask, int first_block, void *stream) {
  return launch_stage_factor(S, Nn, d_mask, first_block, 0, 0, stream);
}
// every instance restarts at its own block d_b0v[inst] (0 = full factorisation)
extern "C" int rldl_launch_stage_factor_each(const rldl_dev_sym *S, const rldl_dev_num *Nn, const int *d_b0v, int tiles_adopted, void *stream) {
  return d_b0v ? launch_stage_factor(S, Nn, 0, 0, d_b0v, tiles_adopted, stream) : -1;
}

// the factorisation with the wave timeline of the arrowhead kernel ([batch][8] stamps, arrow_factor_body); -1: another kernel serves the pattern
extern "C" int rldl_launch_factor_trace(const rldl_dev_sym *S, const rldl_dev_num *Nn, long long *d_trace, void *stream) {
  if (!(S->arrow_ok && S->arrow_dense && S->arrow_g <= 64) || getenv("RLDL_NO_ARROW_FACTOR")) return -1;
  if (sizeof(double) * (size_t)(S->nnzL + S->N + S->arrow_g0 + 2 + 128) > RLDL_LDS_LIMIT) return -1;
  return launch_factor(S, Nn, 0, 0, stream, d_trace) ? 1 : 0;
}
extern "C" int rldl_launch_factor(const rldl_dev_sym *S, const rldl_dev_num *Nn, const int *d_mask, void *stream) {
  if (S->stage.nb > 0 && !getenv("RLDL_NO_STAGE_FACTOR")) return rldl_launch_stage_factor(S, Nn, d_mask, 0, stream);
  return launch_factor(S, Nn, d_mask, 0, stream);
}

extern "C" int rldl_launch_factor_from(const rldl_dev_sym *S, const rldl_dev_num *Nn, int c_start, void *stream) {
  if (c_start < 0 || c_start > S->N) return -1;
  return launch_factor(S, Nn, 0, c_start, stream);
}

extern "C" int rldl_launch_solve(const rldl_dev_sym *S, const rldl_dev_num *Nn, double *d_b, void *stream) {
  if (Nn->batch <= 0) return 0;
  if (arrow_usable(S) && tile_usable(S, Nn)) return launch_tile_solve(S, Nn, d_b, stream);
  if (arrow_usable(S)) return launch_arrow_solve(S, Nn, d_b, stream);
  if (blk_usable(S)) return launch_blk_solve(S, Nn, d_b, stream);
  if (plan_usable(S)) return launch_plan_solve(S, Nn, d_b, stream);
  const size_t lds = sizeof(double) * (size_t)(S->nS + S->N);
  if (lds <= RLDL_LDS_LIMIT)
    hipLaunchKernelGGL(k_solve<true>, dim3(Nn->batch), dim3(WAVE), lds, (hipStream_t)stream, *S, *Nn, d_b);
  else
    hipLaunchKernelGGL(k_solve<false>, dim3(Nn->batch), dim3(WAVE), sizeof(double) * (size_t)S->N, (hipStream_t)stream, *S,
                       *Nn, d_b);
  return launch_status();
}

// wave timeline of one launch of the plugin solve (round-3 tile kernel only): d_trace[batch][8] s_memrealtime ticks (100 MHz):
// wave start, all loads landed, forward gather done, forward product done, backward product done, scatter done, stores issued, 0.  -1: this handle does not run that kernel.
extern "C" int rldl_launch_solve_trace(const rldl_dev_sym *S, const rldl_dev_num *Nn, double *d_b, long long *d_trace, void *stream) {
  if (Nn->batch <= 0 || !d_trace) return -1;
  if (!(arrow_usable(S) && tile_usable(S, Nn) && tile_solve3_usable(S))) return -1;
  return launch_tile_solve3(S, Nn, d_b, d_trace, stream);
}

// `iters` ADMM iterations of every active instance.  The arrowhead kernel runs them inside one launch with the factor
// kept on chip; the other kernels are launched once per iteration (W->write_delta applies to the last one).
// 1 when rldl_launch_admm_iters runs the tile kernel, which starts a solve itself (W->begin_flags): the caller then skips k_solve_begin
extern "C" int rldl_admm_begin_in_kernel(const rldl_dev_sym *S, const rldl_dev_num *Nn, const rldl_dev_admm *W) {
  static const int off = (getenv("RLDL_ITERS_PER_LAUNCH") || getenv("RLDL_SOLVE_BEGIN_LAUNCH")) ? 1 : 0;
  return !off && Nn->batch > 0 && arrow_usable(S) && S->N <= 8 * WAVE && tile_admm_usable(S, Nn, W) && !W->trace;
}
extern "C" int rldl_launch_admm_iters(const rldl_dev_sym *S, const rldl_dev_num *Nn, const rldl_dev_admm *W, int iters, void *stream) {
  if (Nn->batch <= 0 || iters <= 0) return 0;
  static const int one = getenv("RLDL_ITERS_PER_LAUNCH") ? atoi(getenv("RLDL_ITERS_PER_LAUNCH")) : 0;   // timing experiments
  if (arrow_usable(S) && S->N <= 8 * WAVE) {
    if (one <= 0) return launch_arrow_admm(S, Nn, W, iters, stream);
    rldl_dev_admm Wi = *W;
    for (int done = 0; done < iters; done += one) {
      const int k = iters - done < one ? iters - done : one;
      Wi.write_delta = done + k == iters ? W->write_delta : 0;
      if (launch_arrow_admm(S, Nn, &Wi, k, stream)) return -1;
    }
    return 0;
  }
  if (one <= 0 && blk_usable(S) && prod_usable(S, Nn)) return launch_blk_admm(S, Nn, W, stream, iters);   // the whole group in one launch
  // generic patterns on the grouped plan: the whole group in one launch of the loop kernel (the factor row, when it fits, stays in LDS
  // across the iterations instead of being streamed once per iteration)
  if (one <= 0 && iters > 1 && !blk_usable(S) && plan_admm_usable(S)) return launch_plan_admm_loop(S, Nn, W, stream, iters);
  rldl_dev_admm Wi = *W;
  for (int k = 0; k < iters; k++) {
    Wi.write_delta = k + 1 == iters ? W->write_delta : 0;
    if (rldl_launch_admm_iter(S, Nn, &Wi, stream)) return -1;
  }
  return 0;
}

extern "C" int rldl_launch_admm_iter(const rldl_dev_sym *S, const rldl_dev_num *Nn, const rldl_dev_admm *W, void *stream) {
  if (Nn->batch <= 0) return 0;
  if (arrow_usable(S) && S->N <= 8 * WAVE) return launch_arrow_admm(S, Nn, W, 1, stream);
  if (blk_usable(S)) return launch_blk_admm(S, Nn, W, stream);
  if (plan_admm_usable(S)) return launch_plan_admm(S, Nn, W, stream);
  const size_t lds = sizeof(double) * (size_t)(S->nS + S->N);
  if (lds <= RLDL_LDS_LIMIT)
    hipLaunchKernelGGL(k_admm_iter<true>, dim3(Nn->batch), dim3(WAVE), lds, (hipStream_t)stream, *S, *Nn, *W);
  else
    hipLaunchKernelGGL(k_admm_iter<false>, dim3(Nn->batch), dim3(WAVE), sizeof(double) * (size_t)S->N, (hipStream_t)stream,
                       *S, *Nn, *W);
  return launch_status();
}

static size_t check_lds(const rldl_dev_sym *S) { return sizeof(double) * (size_t)(6 * S->n + 6 * S->m + 8); }
static size_t check_lds_staged(const rldl_dev_sym *S) { return check_lds(S) + sizeof(double) * (size_t)(S->nnzP + S->nnzA); }

extern "C" int rldl_launch_admm_check(const rldl_dev_sym *S, const rldl_dev_admm *W, int iter, int approximate,
                                      int final_pass, void *stream) {
  if (W->batch <= 0) return 0;
  // approximate: bit0 = run termination check, bit1 = adapt rho; final_pass: 0 none, 1 final, 2 final + needs info
  int mode = 0;
  if (approximate & 1) mode |= CHK_TERMINATION;
  if (approximate & 2) mode |= CHK_ADAPT;
  if (final_pass) mode |= CHK_FINAL;
  if (final_pass == 2) mode |= CHK_FINAL_NEEDS_INFO;
  // entry-parallel products straight from global memory when the tables exist; else the row loops, on LDS copies of P / A if they fit
  if ((!S->flat_ok || getenv("RLDL_CHECK_STAGED")) && check_lds_staged(S) <= 40 * 1024)
    hipLaunchKernelGGL(k_admm_check<true>, dim3(W->batch), dim3(WAVE), check_lds_staged(S), (hipStream_t)stream, *S, *W, iter, mode, k_no_multi);
  else
    hipLaunchKernelGGL(k_admm_check<false>, dim3(W->batch), dim3(WAVE), check_lds(S), (hipStream_t)stream, *S, *W, iter, mode, k_no_multi);
  return launch_status();
}


extern "C" int rldl_launch_set_rho_vec(const rldl_dev_sym *S, const rldl_dev_admm *W, int init, void *stream) {
  if (W->batch <= 0) return 0;
  hipLaunchKernelGGL(k_set_rho_vec, dim3(W->batch), dim3(WAVE), 0, (hipStream_t)stream, *S, *W, init);
  return launch_status();
}


extern "C" int rldl_launch_polish_prep(const rldl_dev_sym *S, const rldl_dev_admm *W, void *stream) {
  if (W->batch <= 0) return 0;
  hipLaunchKernelGGL(k_polish_prep, dim3(W->batch), dim3(WAVE), sizeof(int) * (size_t)(S->m + 2), (hipStream_t)stream, *S, *W);
  return launch_status();
}
extern "C" int rldl_launch_polish_resid(const rldl_dev_sym *S, const rldl_dev_admm *W, int add_first, void *stream) {
  if (W->batch <= 0) return 0;
  hipLaunchKernelGGL(k_polish_resid, dim3(W->batch), dim3(WAVE), sizeof(double) * (size_t)(3 * S->n + 2 * S->m + 2), (hipStream_t)stream,
                     *S, *W, add_first);
  return launch_status();
}
extern "C" int rldl_launch_polish_finish(const rldl_dev_sym *S, const rldl_dev_admm *W, int add_last, void *stream) {
  if (W->batch <= 0) return 0;
  hipLaunchKernelGGL(k_polish_finish, dim3(W->batch), dim3(WAVE), sizeof(double) * (size_t)(4 * S->n + 5 * S->m + 2), (hipStream_t)stream,
                     *S, *W, add_last);
  return launch_status();
}

extern "C" int rldl_launch_scale_data(const rldl_dev_sym *S, const rldl_dev_admm *W, double *Px, double *Ax, double *q, double *l,
                                      double *u, int iters, void *stream) {
  if (W->batch <= 0) return 0;
  if (S->flat_ok && S->nnzP > 0 && S->nnzA > 0 && S->nnzP <= 16 * WAVE && S->nnzA <= 16 * WAVE && !getenv("RLDL_SCALE_LOOPS")) {
    const size_t lds = sizeof(double) * (size_t)(4 * S->n + 2 * S->m + 2);   // entries in registers: up to 16 rounds of 64 per matrix
    const int rounds = S->Pbr > S->Abr ? S->Pbr : S->Abr;
    const dim3 grid(W->batch), blk(WAVE);
    if (rounds <= 8) hipLaunchKernelGGL((k_scale_data_flat<8, 8, 4>), grid, blk, lds, (hipStream_t)stream, *S, *W, Px, Ax, q, l, u, iters);
    else if (rounds <= 12) hipLaunchKernelGGL((k_scale_data_flat<12, 12, 2>), grid, blk, lds, (hipStream_t)stream, *S, *W, Px, Ax, q, l, u, iters);
    else if (rounds <= 14) hipLaunchKernelGGL((k_scale_data_flat<14, 14, 2>), grid, blk, lds, (hipStream_t)stream, *S, *W, Px, Ax, q, l, u, iters);
    else hipLaunchKernelGGL((k_scale_data_flat<16, 16, 2>), grid, blk, lds, (hipStream_t)stream, *S, *W, Px, Ax, q, l, u, iters);
    return launch_status();
  }
  const size_t small = sizeof(double) * (size_t)(S->n + S->m + 2);
  const size_t staged = small + sizeof(double) * (size_t)(2 * S->n + S->m + S->nnzP + S->nnzA);
  if (staged <= 64 * 1024)
    hipLaunchKernelGGL(k_scale_data<true>, dim3(W->batch), dim3(WAVE), staged, (hipStream_t)stream, *S, *W, Px, Ax, q, l, u, iters);
  else
    hipLaunchKernelGGL(k_scale_data<false>, dim3(W->batch), dim3(WAVE), small, (hipStream_t)stream, *S, *W, Px, Ax, q, l, u, iters);
  return launch_status();
}
extern "C" int rldl_launch_unscale_data(const rldl_dev_sym *S, const rldl_dev_admm *W, double *Px, double *Ax, double *q, double *l,
                                        double *u, void *stream) {
  if (W->batch <= 0) return 0;
  hipLaunchKernelGGL(k_unscale_data, dim3(W->batch), dim3(WAVE), 0, (hipStream_t)stream, *S, *W, Px, Ax, q, l, u);
  return launch_status();
}
extern "C" int rldl_launch_ew_scale(int batch, int len, double *dst, const double *src, const double *s, const double *c, void *stream) {
  if (batch <= 0 || len <= 0) return 0;
  hipLaunchKernelGGL(k_ew_scale, dim3(batch), dim3(256), 0, (hipStream_t)stream, len, dst, src, s, c);
  return launch_status();
}

extern "C" int rldl_launch_bcast_rows(int batch, int len, double *dst, const double *src, void *stream) {
  if (batch <= 0 || len <= 0) return 0;
  hipLaunchKernelGGL(k_bcast_rows, dim3(batch), dim3(256), 0, (hipStream_t)stream, len, dst, src);
  return launch_status();
}
extern "C" int rldl_launch_bcast_range(int batch, int ld, int start, int cnt, double *dst, const double *src_row, void *stream) {
  if (batch <= 0 || cnt <= 0) return 0;
  hipLaunchKernelGGL(k_bcast_range, dim3(batch), dim3(256), 0, (hipStream_t)stream, ld, start, cnt, dst, src_row);
  return launch_status();
}
extern "C" int rldl_launch_scatter_rows(int batch, int cnt, int ld, const int *map, const double *src, double *dst, void *stream) {
  if (batch <= 0 || cnt <= 0) return 0;
  hipLaunchKernelGGL(k_scatter_rows, dim3(batch), dim3(256), 0, (hipStream_t)stream, cnt, ld, map, src, dst);
  return launch_status();
}
extern "C" int rldl_launch_horizon_vectors(int batch, int n_act, int n_max, int m_act, int m_max, const double *q_src, const double *l_src,
                                           const double *u_src, double *q, double *l, double *u, void *stream) {
  if (batch <= 0) return 0;
  hipLaunchKernelGGL(k_horizon_vectors, dim3(batch), dim3(256), 0, (hipStream_t)stream, n_act, n_max, m_act, m_max, q_src, l_src, u_src, q, l, u);
  return launch_status();
}
extern "C" int rldl_launch_horizon_rho(const rldl_dev_sym *S, const rldl_dev_admm *W, const int *d_status, int m_keep, int b_pivot, int *d_b0v,
                                       int *d_n_reused, void *stream) {
  if (W->batch <= 0) return 0;
  hipLaunchKernelGGL(k_horizon_rho, dim3(W->batch), dim3(WAVE), 0, (hipStream_t)stream, *S, *W, d_status, m_keep, b_pivot, d_b0v, d_n_reused);
  return launch_status();
}
extern "C" int rldl_launch_horizon_state_single(const rldl_dev_admm *W, int n_max, int m_max, int n_keep, int m_keep, int term_old, int term_new,
                                                int nt, void *stream) {
  if (W->batch <= 0) return 0;
  hipLaunchKernelGGL(k_horizon_state_single, dim3(W->batch), dim3(WAVE), 0, (hipStream_t)stream, *W, n_max, m_max, n_keep, m_keep, term_old, term_new, nt);
  return launch_status();
}
extern "C" int rldl_launch_set_range(int batch, int ld, int start, int cnt, double *dst, const double *src, const double *s, void *stream) {
  if (batch <= 0 || cnt <= 0) return 0;
  hipLaunchKernelGGL(k_set_range, dim3(batch), dim3(256), 0, (hipStream_t)stream, ld, start, cnt, dst, src, s, (const int *)0);
  return launch_status();
}
// the same guarded by a device word: no write when *d_skip != 0 (the verdict of k_check_bounds, read without a host round trip)
extern "C" int rldl_launch_set_range_guarded(int batch, int ld, int start, int cnt, double *dst, const double *src, const double *s, const int *d_skip,
                                             void *stream) {
  if (batch <= 0 || cnt <= 0) return 0;
  hipLaunchKernelGGL(k_set_range, dim3(batch), dim3(256), 0, (hipStream_t)stream, ld, start, cnt, dst, src, s, d_skip);
  return launch_status();
}

extern "C" int rldl_launch_matvec_A(const rldl_dev_sym *S, const rldl_dev_admm *W, const double *d_x, double *d_out,
                                    void *stream) {
  if (W->batch <= 0) return 0;
  const size_t lds = sizeof(double) * (size_t)(S->n + S->m + 2);
  rldl_dev_sym Sk = *S;
  if (lds > 64 * 1024) Sk.flat_ok = 0;                           // vectors too long for LDS: row loops
  hipLaunchKernelGGL(k_matvec_A, dim3(W->batch), dim3(WAVE), Sk.flat_ok ? lds : 0, (hipStream_t)stream, Sk, *W, d_x, d_out);
  return launch_status();
}

// l <= u for every entry (osqp_update_bounds, src/osqp.c:805-813; osqp_partial_update_bounds, src/recursive_ldl.c:137-145):
// *flag becomes 1 when a lower bound exceeds its upper bound anywhere in the batch
__global__ __launch_bounds__(256) void k_check_bounds(long long count, const double *__restrict__ l, const double *__restrict__ u, int *flag) {
  int bad = 0;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (long long)gridDim.x * blockDim.x) bad |= l[i] > u[i];
  if (__any(bad) && (threadIdx.x & (WAVE - 1)) == 0) { atomicOr(flag, 1); atomicOr(flag + 1, 1); }   // word 0: this update, word 1: sticky until read
}
extern "C" int rldl_launch_check_bounds(long long count, const double *l, const double *u, int *flag, void *stream) {
  if (count <= 0) return 0;
  const long long blocks = (count + 255) / 256;
  hipLaunchKernelGGL(k_check_bounds, dim3((unsigned)(blocks > 4096 ? 4096 : blocks)), dim3(256), 0, (hipStream_t)stream, count, l, u, flag);
  return launch_status();
}

// ---- several workspaces in one launch (osqp_multi_*, rldl_admm.c) ----
// which groups may share the launches of an update of all groups (new P / A values -> scatter, numeric factorisation, tail inverse):
// the arrowhead factor kernel's instantiation, or -1 (another factor path)
extern "C" int rldl_multi_update_key(const rldl_dev_sym *S, const rldl_dev_num *Nn) {
  if (!(S->arrow_ok && S->arrow_dense && S->arrow_g <= 64) || getenv("RLDL_NO_ARROW_FACTOR") || S->polish) return -1;
  if (sizeof(double) * (size_t)(S->nnzL + S->N + S->arrow_g0 + 2 + 128) > RLDL_LDS_LIMIT) return -1;
  if (!S->tile_ok || !Nn->Ti || S->arrow_tb != 0) return -1;
  const int g = S->arrow_g;
  return g <= 16 ? 16 : g <= 32 ? 32 : g <= 48 ? 48 : g <= 56 ? 56 : 64;
}
extern "C" int rldl_multi_update_lds(const rldl_dev_sym *S, int which) {     // dynamic LDS bytes of the factor (0) / tail inverse (1) kernel
  const int g = S->arrow_g;
  return which == 0 ? (int)(sizeof(double) * (size_t)(S->nnzL + S->N + S->arrow_g0 + 2 + 128)) : (int)(sizeof(double) * (size_t)((g * (g - 1)) / 2 + 2 + 2 * LCH + 64 + 256 * ((g + 15) / 16)));
}
extern "C" int rldl_launch_multi_update(const rldl_dev_multi *M, const rldl_dev_multi_pa *PA, int total, int key, int factor_lds, int invert_lds,
                                        void *stream) {
  if (total <= 0) return 0;
  const dim3 grid(total), blk(WAVE);
  hipLaunchKernelGGL(k_kkt_assemble_multi, grid, dim3(256), 0, (hipStream_t)stream, *M, *PA);
  if (launch_status()) return -1;
  // (factorisation and tail inverse are one launch; the larger of the two LDS sizes covers both)
#define MU(SMV, MASKED) hipLaunchKernelGGL(k_arrow_factor_multi<SMV>, grid, blk, (size_t)(factor_lds > invert_lds ? factor_lds : invert_lds), (hipStream_t)stream, *M, MASKED)
  switch (key) { case 16: MU(16, 0); break; case 32: MU(32, 0); break; case 48: MU(48, 0); break; case 56: MU(56, 0); break; case 64: MU(64, 0); break; default: return -1; }
  return launch_status();
}
// osqp_update_rho of every group of a set after an adapt_rho step (osqp.c:1268-1319 -> update_rho_vec): only the instances whose rho moved
extern "C" int rldl_launch_multi_update_rho(const rldl_dev_multi *M, int total, int key, int factor_lds, int invert_lds, void *stream) {
  if (total <= 0) return 0;
  const dim3 grid(total), blk(WAVE);
  hipLaunchKernelGGL(k_kkt_assemble_multi_rho, grid, dim3(256), 0, (hipStream_t)stream, *M);
  if (launch_status()) return -1;
  switch (key) { case 16: MU(16, 1); break; case 32: MU(32, 1); break; case 48: MU(48, 1); break; case 56: MU(56, 1); break; case 64: MU(64, 1); break; default: return -1; }
#undef MU
  return launch_status();
}
extern "C" int rldl_launch_multi_nactive(const rldl_dev_multi *M, int *d_out, void *stream) {
  hipLaunchKernelGGL(k_multi_nactive, dim3(1), dim3(RLDL_NACT_SLOTS), 0, (hipStream_t)stream, *M, d_out);
  return launch_status();
}
// a termination check and / or an adapt_rho step of every group (rldl_launch_admm_check's arguments)
extern "C" int rldl_launch_multi_check(const rldl_dev_multi *M, const rldl_dev_sym *S0, const rldl_dev_admm *W0, int total, int iter, int approximate,
                                       int final_pass, int max_nm, void *stream) {
  if (total <= 0) return 0;
  int mode = 0;
  if (approximate & 1) mode |= CHK_TERMINATION;
  if (approximate & 2) mode |= CHK_ADAPT;
  if (final_pass) mode |= CHK_FINAL;
  if (final_pass == 2) mode |= CHK_FINAL_NEEDS_INFO;
  const size_t lds = sizeof(double) * (size_t)(6 * max_nm + 8);
  hipLaunchKernelGGL((k_admm_check<false, true>), dim3(total), dim3(WAVE), lds, (hipStream_t)stream, *S0, *W0, iter, mode, *M);
  return launch_status();
}
// which groups may share the launches: the instantiation of k_tile_admm their pattern selects (-1: not on the tile kernels) and
// the entry-parallel check kernel
extern "C" int rldl_multi_key(const rldl_dev_sym *S, const rldl_dev_num *Nn, const rldl_dev_admm *W) {
  if (!(arrow_usable(S) && S->N <= 8 * WAVE && tile_admm_usable(S, Nn, W)) || W->trace || !S->flat_ok || getenv("RLDL_CHECK_STAGED")) {
    if (getenv("RLDL_VERBOSE"))
      fprintf(stderr, "[rldl] pattern off the tile chain: arrow_ok %d vsteps %d vrows %d g %d tb %d | tile_ok %d ta %d admm_ok %d vslots %d slots %d ck %d %d %d tk %d sp %d | flat %d\n",
              S->arrow_ok, S->arrow_vsteps, S->arrow_vrows, S->arrow_g, S->arrow_tb, S->tile_ok, S->tile_ta, S->tile_admm_ok, S->tile_vslots, S->tile_slots,
              S->tile_ck[0], S->tile_ck[1], S->tile_ck[2], S->tile_tk, S->tile_sp, S->flat_ok);
    return -1;
  }
  const int tg = S->arrow_vsteps <= 12 ? 12 : S->arrow_vsteps <= 18 ? 18 : 24;
  return tile_admm_gather(S) ? ((tg * 16 + S->tile_ta) * 64 + S->tile_tk) * 64 + S->tile_sp : ((tg * 16 + S->tile_ta) * 64) * 64;   // (scatter variant: tk = sp = 0)
}
extern "C" int rldl_multi_tile_xdw(const rldl_dev_sym *S) { return tile_per_wave(S); }
extern "C" int rldl_multi_tile_wpb(void) { return TILE_WPB; }
extern "C" int rldl_launch_multi_solve_begin(const rldl_dev_multi *M, int total, int n, int m, int cold, int reset_rho_updates, void *stream) {
  if (total <= 0) return 0;
  rldl_dev_admm W0 = {};
  hipLaunchKernelGGL(k_solve_begin, dim3(total), dim3(256), 0, (hipStream_t)stream, W0, n, m, cold, reset_rho_updates, *M);
  return launch_status();
}
extern "C" int rldl_launch_multi_admm_iters(const rldl_dev_multi *M, const rldl_dev_sym *S0, const rldl_dev_num *N0, const rldl_dev_admm *W0,
                                            int iters, int max_xdw, void *stream) {
  if (iters <= 0 || M->ngroups <= 0) return 0;
  return launch_tile_admm(S0, N0, W0, iters, stream, M, M->total_tiles, max_xdw);
}
extern "C" int rldl_launch_multi_check_final(const rldl_dev_multi *M, const rldl_dev_sym *S0, const rldl_dev_admm *W0, int total, int iter,
                                             int max_nm, void *stream) {
  if (total <= 0) return 0;
  const size_t lds = sizeof(double) * (size_t)(6 * max_nm + 8);   // (check_lds of the largest group; all groups share n and m)
  hipLaunchKernelGGL((k_admm_check<false, true>), dim3(total), dim3(WAVE), lds, (hipStream_t)stream, *S0, *W0, iter, CHK_FINAL | CHK_FINAL_NEEDS_INFO, *M);
  return launch_status();
}
// verdicts of the factorisations enqueued since the last read, one word per group: the sticky flags of rldl_dev_num.fail, read and cleared
__global__ void k_multi_fail(rldl_dev_multi M, int *__restrict__ out) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g < M.ngroups) out[g] = M.N[g].fail ? atomicExch(M.N[g].fail, 0) : 0;
}
extern "C" int rldl_launch_multi_fail(const rldl_dev_multi *M, int *d_out, void *stream) {
  if (M->ngroups <= 0) return 0;
  hipLaunchKernelGGL(k_multi_fail, dim3((M->ngroups + 63) / 64), dim3(64), 0, (hipStream_t)stream, *M, d_out);
  return launch_status();
}
// results of every group into caller-order arrays, one workgroup per instance
__global__ __launch_bounds__(WAVE) void k_multi_gather(rldl_dev_multi M, int n, int m, const int *__restrict__ dest, double *__restrict__ x,
                                                       double *__restrict__ y, double *__restrict__ z, int *__restrict__ status,
                                                       int *__restrict__ iter, double *__restrict__ obj, double *__restrict__ pri,
                                                       double *__restrict__ dua) {
  const int bid = blockIdx.x, lane = threadIdx.x;
  const int g = multi_group(M.first_inst, M.ngroups, bid), i = bid - M.first_inst[g], o = dest[bid];
  const rldl_dev_admm &W = M.W[g];
  const double *sx = W.sol_x, *sy = W.sol_y;                    // OSQPSolution (store_solution, auxil.c:527-565): what osqp_batch_get hands out
  for (int k = lane; k < n; k += WAVE) x[(size_t)o * n + k] = sx[(size_t)i * n + k];
  for (int k = lane; k < m; k += WAVE) { y[(size_t)o * m + k] = sy[(size_t)i * m + k]; if (z) z[(size_t)o * m + k] = W.z[(size_t)i * m + k]; }
  if (lane == 0) {
    status[o] = W.status[i]; iter[o] = W.iter[i]; obj[o] = W.obj[i]; pri[o] = W.pri_res[i]; dua[o] = W.dua_res[i];
  }
}
extern "C" int rldl_launch_multi_gather(const rldl_dev_multi *M, int total, int n, int m, const int *dest, double *x, double *y, double *z,
                                        int *status, int *iter, double *obj, double *pri, double *dua, void *stream) {
  if (total <= 0) return 0;
  hipLaunchKernelGGL(k_multi_gather, dim3(total), dim3(WAVE), 0, (hipStream_t)stream, *M, n, m, dest, x, y, z, status, iter, obj, pri, dua);
  return launch_status();
}

// the result record of every instance, packed for the all-gather of a multi-GPU run: [x | y | obj | pri_res | dua_res | iter | status]
__global__ __launch_bounds__(WAVE) void k_pack_results(rldl_dev_admm W, int n, int m, double *__restrict__ rec) {
  const int inst = blockIdx.x, lane = threadIdx.x, L = n + m + 5;
  double *r = rec + (size_t)inst * L;
  for (int k = lane; k < n; k += WAVE) r[k] = W.sol_x[(size_t)inst * n + k];
  for (int k = lane; k < m; k += WAVE) r[n + k] = W.sol_y[(size_t)inst * m + k];
  if (lane == 0) {
    r[n + m] = W.obj[inst]; r[n + m + 1] = W.pri_res[inst]; r[n + m + 2] = W.dua_res[inst];
    r[n + m + 3] = (double)W.iter[inst]; r[n + m + 4] = (double)W.status[inst];
  }
}
extern "C" int rldl_launch_pack_results(const rldl_dev_admm *W, int n, int m, double *rec, void *stream) {
  if (W->batch <= 0) return 0;
  hipLaunchKernelGGL(k_pack_results, dim3(W->batch), dim3(WAVE), 0, (hipStream_t)stream, *W, n, m, rec);
  return launch_status();
}

extern "C" const char *rldl_kernel_arch(void) { return "gfx950"; }

extern "C" int rldl_launch_horizon_values(const rldl_dev_sym *So, const rldl_dev_admm *Wo, const double *oPx, const double *oAx,
                                          int col_keep, double *nPx, int ldP, double *nAx, int ldA, void *stream) {
  if (Wo->batch <= 0 || col_keep <= 0) return 0;
  if (col_keep > So->n) return -1;
  hipLaunchKernelGGL(k_horizon_values, dim3(Wo->batch), dim3(WAVE), 0, (hipStream_t)stream, *So, *Wo, oPx, oAx, col_keep, nPx, ldP, nAx, ldA);
  return launch_status();
}
extern "C" int rldl_launch_horizon_state(const rldl_dev_sym *So, const rldl_dev_admm *Wo, int n_new, int m_new, int n_keep, int m_keep,
                                         int nt, double *xs, double *ys, void *stream) {
  if (Wo->batch <= 0) return 0;
  if (n_keep > So->n || n_keep > n_new || m_keep + nt > So->m || m_keep + nt > m_new) return -1;
  hipLaunchKernelGGL(k_horizon_state, dim3(Wo->batch), dim3(WAVE), 0, (hipStream_t)stream, *So, *Wo, n_new, m_new, n_keep, m_keep, nt, xs, ys);
  return launch_status();
}
extern "C" int rldl_launch_horizon_adopt(const rldl_dev_sym *So, const rldl_dev_num *No, const rldl_dev_sym *Sn, const rldl_dev_num *Nn,
                                         const double *rvo, const double *rvn, int m_keep, int c0, int b_pivot, int ti_prefix, int *d_b0v,
                                         int *d_n_reused, void *stream) {
  if (ti_prefix > 0 && (!No->Ti || !Nn->Ti || ti_prefix > So->stage.pv_nTi || ti_prefix > Sn->stage.pv_nTi)) return -1;
  if (Nn->batch <= 0) return 0;
  if (No->batch != Nn->batch || c0 <= 0 || c0 > So->N || c0 > Sn->N || m_keep > So->m || m_keep > Sn->m) return -1;
  hipLaunchKernelGGL(k_horizon_adopt, dim3(Nn->batch), dim3(WAVE), 0, (hipStream_t)stream, *So, *No, *Sn, *Nn, rvo, rvn, m_keep, c0, b_pivot, ti_prefix,
                     d_b0v, d_n_reused);
  return launch_status();
}
