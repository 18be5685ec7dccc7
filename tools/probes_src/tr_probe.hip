// Developer probe: what ds_read_b64_tr_b16 returns.  LDS holds u16 element i at element index i; every lane passes its own byte address.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k(const int* addr, uint16_t* out) {
  __shared__ uint16_t lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (uint16_t)i;
  __syncthreads();
  unsigned a = (unsigned)(unsigned long)(__attribute__((address_space(3))) void*)lds + addr[threadIdx.x];
  uint2 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
  out[threadIdx.x * 4 + 0] = v.x & 0xffff; out[threadIdx.x * 4 + 1] = v.x >> 16;
  out[threadIdx.x * 4 + 2] = v.y & 0xffff; out[threadIdx.x * 4 + 3] = v.y >> 16;
}
int main() {
  int h_addr[64]; uint16_t h_out[256];
  int* d_addr; uint16_t* d_out;
  hipMalloc(&d_addr, sizeof(h_addr)); hipMalloc(&d_out, sizeof(h_out));
  for (int pat = 0; pat < 3; ++pat) {
    for (int l = 0; l < 64; ++l) {
      if (pat == 0) h_addr[l] = l * 8;                          // consecutive 8-byte pieces
      if (pat == 1) h_addr[l] = (l & 15) * 128 + (l >> 4) * 8;  // lane (l&15) -> row of 64 elements, group -> 4-element column block
      if (pat == 2) h_addr[l] = (l & 3) * 256 + ((l >> 2) & 3) * 8 + (l >> 4) * 2048;
    }
    hipMemcpy(d_addr, h_addr, sizeof(h_addr), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d_addr, d_out);
    hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost);
    printf("pattern %d (lane: byte address -> 4 element indices returned)\n", pat);
    for (int l = 0; l < 64; ++l) printf("  lane %2d addr %5d (elem %4d): %4d %4d %4d %4d\n", l, h_addr[l], h_addr[l] / 2, h_out[l * 4], h_out[l * 4 + 1], h_out[l * 4 + 2], h_out[l * 4 + 3]);
  }
  return 0;
}
