// Micro-benchmark: throughput of LDS f64 atomic adds (ds_add_f64, no return) per CU for several address patterns, against
// plain ds_read_b64 + ds_write_b64 pairs.  Diagnostic only.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/ubench_ldsatomic.hip -o scripts/ubench_ldsatomic && scripts/ubench_ldsatomic
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

// MODE 0: atomic, lane i -> slot i (conflict-free)   1: atomic, all lanes one slot   2: atomic, 16 lanes per slot
// MODE 3: atomic, pseudo-random distinct slots (stride 17)   4: read + add + write (non-atomic), lane i -> slot i
// MODE 5: atomic, only 32 lanes active, distinct   6 / 7 / 8: 2 / 4 lanes per slot, 8 lanes colliding pairwise
template <int MODE>
__global__ __launch_bounds__(256) void kb(double *out, long long *cyc, int reps) {
  __shared__ double sh[4 * 128];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  double *x = sh + wv * 128;
  x[lane] = 0.0; x[64 + lane] = 0.0;
  __syncthreads();
  int slot = lane;
  if (MODE == 1) slot = 0;
  if (MODE == 2) slot = lane >> 4;
  if (MODE == 3) slot = (lane * 17) & 63;
  if (MODE == 6) slot = lane >> 1;                               // 2 lanes per slot
  if (MODE == 7) slot = lane >> 2;                               // 4 lanes per slot
  if (MODE == 8) slot = lane < 8 ? (lane >> 1) : lane;           // 8 lanes collide pairwise, 56 are alone
  const double v = 1e-3 * (lane + 1);
  const long long t0 = wall_clock64();
  for (int r = 0; r < reps; r++) {
#pragma unroll
    for (int u = 0; u < 8; u++) {
      if (MODE == 4) { x[slot] = x[slot] + v; __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); }
      else if (MODE == 5) { if (lane < 32) unsafeAtomicAdd(&x[slot], v); }
      else unsafeAtomicAdd(&x[slot], v);
    }
  }
  __syncthreads();
  const long long t1 = wall_clock64();
  out[blockIdx.x * 256 + threadIdx.x] = x[lane];
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int MODE>
static void run(const char *name, double *dout, long long *dcyc, int blocks, int reps) {
  for (int it = 0; it < 2; it++) hipLaunchKernelGGL((kb<MODE>), dim3(blocks), dim3(256), 0, 0, dout, dcyc, reps);
  hipDeviceSynchronize();
  std::vector<long long> c(blocks);
  hipMemcpy(c.data(), dcyc, sizeof(long long) * blocks, hipMemcpyDeviceToHost);
  double s = 0; for (auto v : c) s += v;
  // wall_clock64 ticks at 100 MHz: ns = ticks * 10; instructions per CU in that time = waves_per_cu * reps * 8
  const double ns = s / blocks * 10.0, wpc = (double)blocks * 4 / 256.0;
  printf("%-44s blocks=%5d  %.2f ns per wave-instruction per CU (~%.1f cycles at 2.4 GHz)\n", name, blocks, ns / (wpc * reps * 8), ns / (wpc * reps * 8) * 2.4);
}
int main() {
  double *dout; long long *dcyc;
  hipMalloc(&dout, sizeof(double) * 256 * 4096); hipMalloc(&dcyc, sizeof(long long) * 4096);
  const int reps = 2000;
  for (int blocks : {256, 1024}) {   // 4 and 16 waves per CU
    run<0>("atomic add, 64 distinct slots", dout, dcyc, blocks, reps);
    run<3>("atomic add, 64 distinct slots (stride 17)", dout, dcyc, blocks, reps);
    run<5>("atomic add, 32 active lanes, distinct", dout, dcyc, blocks, reps);
    run<8>("atomic add, 8 of 64 lanes collide pairwise", dout, dcyc, blocks, reps);
    run<6>("atomic add, 2 lanes per slot", dout, dcyc, blocks, reps);
    run<7>("atomic add, 4 lanes per slot", dout, dcyc, blocks, reps);
    run<2>("atomic add, 16 lanes per slot", dout, dcyc, blocks, reps);
    run<1>("atomic add, all lanes one slot", dout, dcyc, blocks, reps);
    run<4>("read + add + write, 64 distinct slots", dout, dcyc, blocks, reps);
  }
  return 0;
}
