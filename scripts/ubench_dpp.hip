// Semantics check of the DPP row shifts used by the tile kernels' segmented reduction (gfx950): lane i of
// update_dpp(old, v, 0x110 + n) must hold v of lane i - n when that lane is in the same row of 16, else old.
// build: hipcc --offload-arch=gfx950 -O3 scripts/ubench_dpp.hip -o /tmp/ubench_dpp && /tmp/ubench_dpp
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int *out) {
  const int lane = threadIdx.x;
  out[lane] = __builtin_amdgcn_update_dpp(-1, lane, 0x111, 0xf, 0xf, false);
  out[64 + lane] = __builtin_amdgcn_update_dpp(-1, lane, 0x112, 0xf, 0xf, false);
  out[128 + lane] = __builtin_amdgcn_update_dpp(-1, lane, 0x114, 0xf, 0xf, false);
  out[192 + lane] = __builtin_amdgcn_update_dpp(-1, lane, 0x118, 0xf, 0xf, false);
}
int main() {
  int *d, h[256];
  hipMalloc(&d, sizeof(h));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int bad = 0;
  const int sh[4] = {1, 2, 4, 8};
  for (int s = 0; s < 4; s++)
    for (int i = 0; i < 64; i++) {
      const int want = (i % 16) >= sh[s] ? i - sh[s] : -1;
      if (h[64 * s + i] != want) { if (bad < 8) printf("shift %d lane %d: got %d want %d\n", sh[s], i, h[64 * s + i], want); bad++; }
    }
  printf("row_shr semantics %s\n", bad ? "DIFFER" : "as assumed");
  return bad != 0;
}
