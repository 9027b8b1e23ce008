// Probe: what do HW_REG_LDS_ALLOC / HW_REG_HW_ID read for co-resident workgroups?
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
  extern __shared__ float lds[];
  lds[threadIdx.x] = 1.f;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned lds_alloc = __builtin_amdgcn_s_getreg((32 - 1) << 11 | 0 << 6 | 6);   // HW_REG_LDS_ALLOC full
    unsigned hw_id = __builtin_amdgcn_s_getreg((32 - 1) << 11 | 0 << 6 | 4);       // HW_REG_HW_ID
    unsigned xcc = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20);         // HW_REG_XCC_ID
    out[blockIdx.x * 4 + 0] = lds_alloc; out[blockIdx.x * 4 + 1] = hw_id; out[blockIdx.x * 4 + 2] = xcc;
  }
  // keep blocks resident for a while so two share a CU
  for (int i = 0; i < 2000; ++i) __builtin_amdgcn_s_sleep(32);
}
int main() {
  const int nb = 512;
  unsigned* d; hipMalloc(&d, nb * 16); hipMemset(d, 0, nb * 16);
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 60000);
  k<<<nb, 256, 57000>>>(d);
  hipDeviceSynchronize();
  unsigned h[nb * 4]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  for (int b : {0, 1, 2, 7, 8, 255, 256, 257, 300, 511})
    printf("block %3d lds_alloc=0x%08x (base=%u size=%u) hw_id=0x%08x (cu=%u sh=%u se=%u) xcc=%u\n", b, h[b*4], h[b*4] & 0xff, (h[b*4] >> 12) & 0x1ff,
           h[b*4+1], (h[b*4+1] >> 8) & 0xf, (h[b*4+1] >> 12) & 1, (h[b*4+1] >> 13) & 7, h[b*4+2]);
  int nz = 0; for (int b = 0; b < nb; ++b) nz += (h[b*4] & 0xff) != 0;
  printf("blocks with lds base != 0: %d of %d\n", nz, nb);
  return 0;
}
