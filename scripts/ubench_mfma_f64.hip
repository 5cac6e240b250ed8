// Micro-benchmark for the question SURVEY.md section 7 step 6 asks: does v_mfma_f64_16x16x4_f64 pay for the dense stage blocks?
// One wave = one instance; the work is the Schur update of one stage block of the MPC shape,
//     S (s x s) -= L (s x sp) diag(d) L',   s = sp = 22 (zero padded to 24 / 32),
// in the two formulations:
//   A  what k_stage_factor_r does: row per lane in registers, the other rows as LDS broadcast reads, SM^2 fmas per lane;
//   B  MFMA: S as 2 x 2 tiles of 16 x 16, K = 24 in 6 steps of 4 -> 24 v_mfma_f64_16x16x4_f64, operands read from the same LDS tile
//      (one double per lane per operand), result tiles written back to LDS (what the row-per-lane elimination needs next).
// Prints cycles per block update per wave with every CU busy.  Diagnostic only.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/ubench_mfma_f64.hip -o scripts/ubench_mfma_f64 && scripts/ubench_mfma_f64
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define SM 24
#define LD 25
typedef double v4d __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void kb(double *out, long long *cyc, int reps) {
  __shared__ double sh[4 * (32 * LD + 32 + 32 * 33)];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  double *Lt = sh + wv * (32 * LD + 32 + 32 * 33), *dv = Lt + 32 * LD, *So = dv + 32;
  for (int p = lane; p < 32 * LD; p += 64) Lt[p] = (p / LD < 22 && p % LD < 22) ? 1e-3 * ((p * 7) % 13 - 6) : 0.0;
  if (lane < 32) dv[lane] = lane < 22 ? 1.0 + 0.01 * lane : 0.0;
  __syncthreads();
  double w[SM];
#pragma unroll
  for (int c = 0; c < SM; c++) w[c] = 0.001 * (lane + c);
  v4d acc[4];
#pragma unroll
  for (int t = 0; t < 4; t++) acc[t] = v4d{0, 0, 0, 0};
  const long long t0 = wall_clock64();
  for (int r = 0; r < reps; r++) {
    if (MODE == 0) {
      const double *myL = Lt + (lane < 22 ? lane : 0) * LD;
      double lcd[SM];
#pragma unroll
      for (int c = 0; c < SM; c++) lcd[c] = myL[c] * dv[c];
#pragma unroll
      for (int k = 0; k < SM; k++) {
        const double *lk = Lt + k * LD;
        double a = 0.0;
#pragma unroll
        for (int c = 0; c < SM; c++) a = fma(lcd[c], lk[c], a);
        w[k] -= a;
      }
    } else {
      // operands: A(i, k) = L(R0 + i, k) d(k), lane holds i = lane % 16, k = 4 kk + lane / 16;  B(k, j) = L(C0 + j, k)
      double a[2][6], b[2][6];
#pragma unroll
      for (int g = 0; g < 2; g++)
#pragma unroll
        for (int kk = 0; kk < 6; kk++) {
          const int k = 4 * kk + (lane >> 4);
          const double v = Lt[(16 * g + (lane & 15)) * LD + k];
          b[g][kk] = v; a[g][kk] = v * dv[k];
        }
#pragma unroll
      for (int R = 0; R < 2; R++)
#pragma unroll
        for (int Cc = 0; Cc < 2; Cc++)
#pragma unroll
          for (int kk = 0; kk < 6; kk++) acc[2 * R + Cc] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[R][kk], b[Cc][kk], acc[2 * R + Cc], 0, 0, 0);
      // result tile element (lane / 16 + 4 q, lane % 16) -> LDS (row major 32 x 33), as the row-per-lane elimination wants it
#pragma unroll
      for (int t = 0; t < 4; t++)
#pragma unroll
        for (int q = 0; q < 4; q++) So[(16 * (t >> 1) + (lane >> 4) + 4 * q) * 33 + 16 * (t & 1) + (lane & 15)] = acc[t][q];
      if (MODE == 2) {                                            // ... and each lane picks its row up again (the full hand-over)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const double *mine = So + (lane < 32 ? lane : 0) * 33;
#pragma unroll
        for (int c = 0; c < SM; c++) w[c] -= mine[c];
      }
    }
    asm volatile("" ::: "memory");
  }
  __syncthreads();
  const long long t1 = wall_clock64();
  double s = 0.0;
#pragma unroll
  for (int c = 0; c < SM; c++) s += w[c];
#pragma unroll
  for (int t = 0; t < 4; t++) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
  out[blockIdx.x * 256 + threadIdx.x] = s + So[lane];
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int MODE>
static void run(const char *name, double *dout, long long *dcyc, int blocks, int reps) {
  for (int it = 0; it < 2; it++) hipLaunchKernelGGL((kb<MODE>), dim3(blocks), dim3(256), 0, 0, dout, dcyc, reps);
  (void)hipDeviceSynchronize();
  std::vector<long long> c(blocks);
  (void)hipMemcpy(c.data(), dcyc, sizeof(long long) * blocks, hipMemcpyDeviceToHost);
  double s = 0; for (auto v : c) s += v;
  const double ns = s / blocks * 10.0, wpc = (double)blocks * 4 / 256.0;
  printf("%-72s %2.0f waves/CU  %8.1f ns per block update per wave  = %7.1f ns of CU time per update (~%6.0f cycles)\n", name, wpc, ns / reps,
         ns / reps / wpc, ns / reps / wpc * 2.4);
}
int main() {
  double *dout; long long *dcyc;
  (void)hipMalloc(&dout, sizeof(double) * 256 * 1024); (void)hipMalloc(&dcyc, sizeof(long long) * 1024);
  const int reps = 500;
  for (int blocks : {256, 512}) {   // 4 and 8 waves per CU (the LDS of this test holds 2 workgroups per CU)
    run<0>("A  row per lane, LDS broadcast rows, 576 v_fma_f64 per lane", dout, dcyc, blocks, reps);
    run<1>("B  24 x v_mfma_f64_16x16x4_f64, operands from LDS, result tiles to LDS", dout, dcyc, blocks, reps);
    run<2>("B' the same + every lane reads its row back (hand-over to the elimination)", dout, dcyc, blocks, reps);
  }
  return 0;
}
