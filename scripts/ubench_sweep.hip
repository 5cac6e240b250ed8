// Micro-benchmark of the in-group triangle sweeps (the dependent chain of the tri-solve): cycles per sweep step for
// several code shapes, alone on a SIMD and with 4 waves per SIMD.  Diagnostic only (not part of the library).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/ubench_sweep.hip -o scripts/ubench_sweep && scripts/ubench_sweep
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define WAVE 64
__device__ __forceinline__ double readlane_f64(double v, int src) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}

// MODE 0: scalar mask + select (library code)   1: no mask (wrong, timing)   2: mask only the high dword
// MODE 3: values from registers instead of LDS (no ds_read in the loop), scalar mask
template <int U, int MODE>
__device__ __forceinline__ double sweep_fwd(const double *Tp, int g, int lane, double acc) {
  const int nst = g - 1, nb = (nst + U - 1) / U;
  const int lc = lane < g ? lane : g - 1;
  const double *rowp = Tp + ((lc * (lc - 1)) >> 1);
  const unsigned long long live = g >= 64 ? ~0ull : ((1ull << g) - 1ull);
  double tA[U], tB[U];
  auto load = [&](int s0, double (&tb)[U]) {
#pragma unroll
    for (int u = 0; u < U; u++) tb[u] = MODE == 3 ? 1e-3 * (double)(s0 + u + lane) : rowp[s0 + u];
  };
  auto proc = [&](int s0, const double (&tb)[U]) {
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int a = s0 + u;
      double tm;
      if (MODE == 1) tm = tb[u];
      else if (MODE == 2) {
        const unsigned long long mk = (~1ull << a) & live;
        const int hi = __builtin_amdgcn_inverse_ballot_w64(mk) ? __double2hiint(tb[u]) : 0;
        tm = __hiloint2double(hi, __double2loint(tb[u]));
      } else {
        const unsigned long long mk = a < nst ? ((~1ull << a) & live) : 0ull;
        tm = __builtin_amdgcn_inverse_ballot_w64(mk) ? tb[u] : 0.0;
      }
      const double xj = readlane_f64(acc, a & 63);
      acc = fma(-tm, xj, acc);
    }
  };
  int b = 0;
  if (nb > 0) load(0, tA);
  while (b + 2 <= nb) {
    load((b + 1) * U, tB);
    proc(b * U, tA);
    if (b + 2 < nb) load((b + 2) * U, tA);
    proc((b + 1) * U, tB);
    b += 2;
  }
  if (b < nb) proc(b * U, tA);
  return acc;
}

template <int U, int MODE>
__global__ __launch_bounds__(256) void kb(const double *Tg, double *out, long long *cyc, int g, int reps, int tri) {
  extern __shared__ double sh[];
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  double *Tv = sh + (size_t)wv * tri;
  for (int i = lane; i < tri; i += WAVE) Tv[i] = Tg[i];
  __syncthreads();
  double acc = 1.0 + lane;
  const long long w0 = wall_clock64();
  const long long t0 = clock64();
  for (int r = 0; r < reps; r++) acc = sweep_fwd<U, MODE>(Tv, g, lane, acc);
  const long long t1 = clock64();
  const long long w1 = wall_clock64();
  const int w = blockIdx.x * (blockDim.x >> 6) + wv;
  out[(size_t)w * 64 + lane] = acc;
  if (lane == 0) { cyc[2 * w] = t1 - t0; cyc[2 * w + 1] = w1 - w0; }
}

template <int U, int MODE>
static void run(const char *name, const double *dT, double *dout, long long *dcyc, int g, int reps, int tri, int blocks, int wpb) {
  const size_t lds = sizeof(double) * (size_t)tri * wpb;
  hipFuncSetAttribute((const void *)kb<U, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  for (int it = 0; it < 2; it++) hipLaunchKernelGGL((kb<U, MODE>), dim3(blocks), dim3(64 * wpb), lds, 0, dT, dout, dcyc, g, reps, tri);
  hipDeviceSynchronize();
  const int nw = blocks * wpb;
  std::vector<long long> c(2 * nw);
  hipMemcpy(c.data(), dcyc, sizeof(long long) * 2 * nw, hipMemcpyDeviceToHost);
  double sc = 0, sw = 0;
  for (int i = 0; i < nw; i++) { sc += c[2 * i]; sw += c[2 * i + 1]; }
  const double steps = (double)reps * (g - 1);
  printf("%-34s waves=%5d  memtime ticks/step %7.2f   ns/step %7.2f\n", name, nw, sc / nw / steps, sw / nw * 10.0 / steps);
}

int main() {
  const int g = 48, tri = g * (g - 1) / 2, reps = 200;
  std::vector<double> T(tri);
  for (int i = 0; i < tri; i++) T[i] = 1e-3 * ((i * 7919) % 13 - 6);
  double *dT, *dout; long long *dcyc;
  hipMalloc(&dT, sizeof(double) * tri); hipMalloc(&dout, sizeof(double) * 64 * 4096); hipMalloc(&dcyc, sizeof(long long) * 2 * 4096);
  hipMemcpy(dT, T.data(), sizeof(double) * tri, hipMemcpyHostToDevice);
  for (int cfg = 0; cfg < 3; cfg++) {
    const int blocks = cfg == 0 ? 1 : (cfg == 1 ? 256 : 1024), wpb = cfg == 0 ? 1 : 4;     // 1 wave; 1 wave/SIMD; 4 waves/SIMD
    printf("--- %d workgroups x %d waves\n", blocks, wpb);
    run<4, 0>("U=4 scalar mask (library)", dT, dout, dcyc, g, reps, tri, blocks, wpb);
    run<8, 0>("U=8 scalar mask", dT, dout, dcyc, g, reps, tri, blocks, wpb);
    run<8, 1>("U=8 no mask", dT, dout, dcyc, g, reps, tri, blocks, wpb);
    run<8, 2>("U=8 high-dword mask", dT, dout, dcyc, g, reps, tri, blocks, wpb);
    run<8, 3>("U=8 scalar mask, no LDS reads", dT, dout, dcyc, g, reps, tri, blocks, wpb);
    run<16, 0>("U=16 scalar mask", dT, dout, dcyc, g, reps, tri, blocks, wpb);
  }
  return 0;
}
