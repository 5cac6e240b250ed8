// Micro-benchmark: LDS-array cost per wave-instruction (per CU, all CUs busy) of the access shapes of the tile kernels:
// ds_read_b64 / ds_write_b64 / ds_add_f64 with consecutive, random and random-distinct double indices in a 100-entry vector,
// all 64 lanes or only some of them active.  Diagnostic only.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/ubench_lds2.hip -o scripts/ubench_lds2 && scripts/ubench_lds2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <numeric>

#define U 8
// OP 0: read, 1: atomic add, 2: write.  idx: [U][64] double index per lane and unrolled step (host-made), nact: active lanes
template <int OP>
__global__ __launch_bounds__(256) void kb(const int *__restrict__ idx, double *out, long long *cyc, int reps, int nact) {
  __shared__ double sh[4 * 160];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  double *x = sh + wv * 160;
  for (int i = lane; i < 160; i += 64) x[i] = 0.0;
  int ix[U];
#pragma unroll
  for (int u = 0; u < U; u++) ix[u] = idx[u * 64 + lane];
  __syncthreads();
  double acc = 0.0;
  const double v = 1e-3 * (lane + 1);
  const bool on = lane < nact;
  const long long t0 = wall_clock64();
  for (int r = 0; r < reps; r++) {
#pragma unroll
    for (int u = 0; u < U; u++) {
      if (on) {
        if (OP == 0) acc += x[ix[u]];
        else if (OP == 1) unsafeAtomicAdd(&x[ix[u]], v);
        else x[ix[u]] = v;
      }
    }
    if (OP == 2) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  }
  __syncthreads();
  const long long t1 = wall_clock64();
  out[blockIdx.x * 256 + threadIdx.x] = x[lane] + acc;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
static int *d_idx; static double *dout; static long long *dcyc;
template <int OP>
static void run(const char *name, const std::vector<int> &idx, int blocks, int reps, int nact) {
  hipMemcpy(d_idx, idx.data(), sizeof(int) * U * 64, hipMemcpyHostToDevice);
  for (int it = 0; it < 2; it++) hipLaunchKernelGGL((kb<OP>), dim3(blocks), dim3(256), 0, 0, d_idx, dout, dcyc, reps, nact);
  hipDeviceSynchronize();
  std::vector<long long> c(blocks);
  hipMemcpy(c.data(), dcyc, sizeof(long long) * blocks, hipMemcpyDeviceToHost);
  double s = 0; for (auto v : c) s += v;
  const double ns = s / blocks * 10.0, wpc = (double)blocks * 4 / 256.0;
  printf("%-58s %2.0f waves/CU  %6.2f ns per wave-instruction per CU (~%5.1f cycles at 2.4 GHz)\n", name, wpc, ns / (wpc * reps * U), ns / (wpc * reps * U) * 2.4);
}
int main() {
  hipMalloc(&d_idx, sizeof(int) * U * 64); hipMalloc(&dout, sizeof(double) * 256 * 4096); hipMalloc(&dcyc, sizeof(long long) * 4096);
  const int reps = 2000;
  srand(7);
  std::vector<int> cons(U * 64), rnd(U * 64), dist(U * 64), dist32(U * 64), bankfree(U * 64);
  for (int u = 0; u < U; u++) {
    std::vector<int> p(100); std::iota(p.begin(), p.end(), 0);
    for (int i = 99; i > 0; i--) std::swap(p[i], p[rand() % (i + 1)]);
    for (int l = 0; l < 64; l++) {
      cons[u * 64 + l] = l;
      rnd[u * 64 + l] = rand() % 100;
      dist[u * 64 + l] = p[l];                                  // 64 distinct of 100: what an edge-coloured step looks like
    }
    // distinct AND bank-conflict-free per half-wave: lanes 0-31 and 32-63 each cover all residues mod 32 once
    for (int h = 0; h < 2; h++) {
      std::vector<int> res(32); std::iota(res.begin(), res.end(), 0);
      for (int i = 31; i > 0; i--) std::swap(res[i], res[rand() % (i + 1)]);
      for (int l = 0; l < 32; l++) bankfree[u * 64 + h * 32 + l] = res[l] + 32 * ((l + h + u) % 3);   // indices < 96
    }
  }
  for (int blocks : {256, 768, 1024}) {
    run<0>("ds_read_b64  consecutive", cons, blocks, reps, 64);
    run<0>("ds_read_b64  random of 100", rnd, blocks, reps, 64);
    run<0>("ds_read_b64  64 distinct of 100", dist, blocks, reps, 64);
    run<0>("ds_read_b64  distinct, residues mod 32 distinct per half", bankfree, blocks, reps, 64);
    run<2>("ds_write_b64 consecutive", cons, blocks, reps, 64);
    run<1>("ds_add_f64   consecutive", cons, blocks, reps, 64);
    run<1>("ds_add_f64   random of 100 (same-address collisions)", rnd, blocks, reps, 64);
    run<1>("ds_add_f64   64 distinct of 100", dist, blocks, reps, 64);
    run<1>("ds_add_f64   distinct, residues mod 32 distinct per half", bankfree, blocks, reps, 64);
    run<1>("ds_add_f64   42 active lanes, distinct of 100", dist, blocks, reps, 42);
    run<1>("ds_add_f64   32 active lanes, distinct of 100", dist, blocks, reps, 32);
    run<1>("ds_add_f64   16 active lanes, distinct of 100", dist, blocks, reps, 16);
  }
  return 0;
}
