// Developer probe: issue cost (shader cycles per wave64 instruction) of the VALU operations an attention softmax is made of, for 1 / 2 / 3 waves
// per SIMD.  Each wave runs REP x 32 independent instructions of one kind between two s_memtime reads.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP 256
template <int OP>
__global__ void k(float* out, unsigned long* cyc, float seed) {
  float a[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) a[i] = seed + i * 0.01f + threadIdx.x * 1e-4f;
  const unsigned long t0 = __builtin_amdgcn_s_memtime();
  for (int r = 0; r < REP; ++r) {
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      if (OP == 0) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
      if (OP == 1) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(a[i]));
      if (OP == 2) asm volatile("v_add_f32 %0, %0, %0" : "+v"(a[i]));
      if (OP == 3) asm volatile("v_max3_f32 %0, %0, %0, %0" : "+v"(a[i]));
      if (OP == 4) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %0" : "+v"(a[i]));
      if (OP == 5) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
      if (OP == 6) asm volatile("v_mul_f32 %0, %0, %0" : "+v"(a[i]));
      if (OP == 7) asm volatile("v_mov_b32 %0, %0" : "+v"(a[i]));
    }
  }
  asm volatile("s_nop 0" ::: "memory");
  const unsigned long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
#pragma unroll
  for (int i = 0; i < 32; ++i) s += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}
template <int OP>
__global__ void kpk(float* out, unsigned long* cyc, float seed) {
  typedef __attribute__((ext_vector_type(2))) float f2;
  f2 a[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) a[i] = f2{seed + i * 0.01f, seed + threadIdx.x * 1e-4f};
  const unsigned long t0 = __builtin_amdgcn_s_memtime();
  for (int r = 0; r < REP; ++r) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (OP == 0) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(a[i]));
      if (OP == 1) asm volatile("v_pk_add_f32 %0, %0, %0" : "+v"(a[i]));
      if (OP == 2) asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(a[i]));
    }
  }
  asm volatile("s_nop 0" ::: "memory");
  const unsigned long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += a[i].x + a[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}
template <class F>
void run(const char* name, F kern, int per_rep) {
  float* d_out; unsigned long* d_cyc; unsigned long h[4096];
  hipMalloc(&d_out, 256 * 1024 * 4 * 4); hipMalloc(&d_cyc, sizeof(h));
  for (int wps = 1; wps <= 3; ++wps) {  // waves per SIMD: block = 256 * wps threads, one block per CU
    hipLaunchKernelGGL(kern, dim3(256), dim3(256 * wps), 0, 0, d_out, d_cyc, 0.5f);
    hipLaunchKernelGGL(kern, dim3(256), dim3(256 * wps), 0, 0, d_out, d_cyc, 0.5f);
    hipDeviceSynchronize();
    hipMemcpy(h, d_cyc, 256 * 4 * wps * sizeof(unsigned long), hipMemcpyDeviceToHost);
    double s = 0; for (int i = 0; i < 256 * 4 * wps; ++i) s += h[i];
    s /= 256 * 4 * wps;
    printf("%-22s %d wave(s)/SIMD: %7.2f cycles per instruction per wave  -> %6.2f cycles of SIMD time per instruction\n", name, wps, s / (REP * per_rep), s / (REP * per_rep) / wps);
  }
  hipFree(d_out); hipFree(d_cyc);
}
int main() {
  run("v_exp_f32", k<0>, 32); run("v_fma_f32", k<1>, 32); run("v_add_f32", k<2>, 32); run("v_max3_f32", k<3>, 32);
  run("v_cvt_pk_bf16_f32", k<4>, 32); run("v_rcp_f32", k<5>, 32); run("v_mul_f32", k<6>, 32); run("v_mov_b32", k<7>, 32);
  run("v_pk_fma_f32", kpk<0>, 16); run("v_pk_add_f32", kpk<1>, 16); run("v_pk_mul_f32", kpk<2>, 16);
  return 0;
}
