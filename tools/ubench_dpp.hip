// semantics probe (MI355X): wave_shl:1 / wave_shr:1 DPP moves with bound_ctrl and EXEC-masked source lanes,
// and v_permlane32_swap.  Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_dpp.hip -o tools/ubench_dpp
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void probe(int* out, int nact) {
  const int lane = threadIdx.x;
  int shl = -7, shr = -7, sw0 = -7, sw1 = -7;
  const int slot = lane & 31;
  if (slot < nact) {
    const int v = 100 + lane;
    shl = __builtin_amdgcn_mov_dpp(v, 0x130, 0xF, 0xF, true);
    shr = __builtin_amdgcn_mov_dpp(v, 0x138, 0xF, 0xF, true);
#if __has_builtin(__builtin_amdgcn_permlane32_swap)
    auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    sw0 = r[0];
    sw1 = r[1];
#endif
  }
  out[lane] = shl;
  out[64 + lane] = shr;
  out[128 + lane] = sw0;
  out[192 + lane] = sw1;
}

int main() {
  int* d;
  hipMalloc(&d, 256 * 4);
  int h[256];
  for (int nact : {32, 25}) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, nact);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("nact=%d\n", nact);
    const char* nm[4] = {"shl", "shr", "sw0", "sw1"};
    for (int a = 0; a < 4; ++a) {
      printf("%s:", nm[a]);
      for (int l = 0; l < 64; ++l) printf(" %d", h[64 * a + l]);
      printf("\n");
    }
  }
  return 0;
}
