// Which XCD does workgroup b run on?  HW_REG_XCC_ID (hardware register 20, bits 3:0) against blockIdx % 8, for a grid of 768
// workgroups of 256 threads (the zy kernel's shape).   hipcc --offload-arch=gfx950 -O3 scripts/mb/xcc_probe.hip -o scripts/mb/xcc_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(int* out) {
  if (threadIdx.x == 0) out[blockIdx.x] = (int)(__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 15u);
}
int main() {
  const int n = 768;
  int* d = nullptr;
  if (hipMalloc(&d, n * sizeof(int)) != hipSuccess) return 1;
  hipLaunchKernelGGL(probe, dim3(n), dim3(256), 0, 0, d);
  int h[n];
  if (hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return 2;
  int hist[16] = {0}, same = 0;
  for (int i = 0; i < n; ++i) { hist[h[i] & 15]++; same += (h[i] == i % 8); }
  printf("XCC_ID histogram:");
  for (int i = 0; i < 16; ++i) printf(" %d", hist[i]);
  printf("\nworkgroups with XCC_ID == blockIdx %% 8: %d of %d\n", same, n);
  return 0;
}
