// Developer probe: do the MFMAs of one wave and the VALU work of ANOTHER wave on the same SIMD overlap?  512-thread blocks (two waves per SIMD),
// one block per CU: waves 0-3 issue only v_mfma_f32_32x32x16_bf16 (4 independent accumulators), waves 4-7 only one kind of VALU instruction;
// each wave measures its own cycles per instruction, alone (the other half exits at once) and together.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
#define REP 512
template <int VOP>
__global__ __launch_bounds__(512) void k(float* out, unsigned long* cyc, int mode) {  // mode 0: both, 1: MFMA waves only, 2: VALU waves only
  const int wave = threadIdx.x >> 6;
  unsigned long t0 = 0, t1 = 0;
  float sink = 0;
  if (wave < 4) {
    if (mode == 2) return;
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(0.01f * threadIdx.x); b[e] = (__bf16)(0.02f * e); }
    t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < REP; ++r) {
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    }
    for (int i = 0; i < 4; ++i) sink += acc[i][0];
    asm volatile("s_nop 0" ::"v"(sink));
    t1 = __builtin_amdgcn_s_memtime();
  } else {
    if (mode == 1) return;
    float x[16];
    for (int i = 0; i < 16; ++i) x[i] = 0.5f + i * 0.01f + threadIdx.x * 1e-4f;
    t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < REP; ++r) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        if (VOP == 0) asm volatile("v_exp_f32 %0, %0" : "+v"(x[i]));
        if (VOP == 1) asm volatile("v_add_f32 %0, %0, %0" : "+v"(x[i]));
        if (VOP == 2) asm volatile("v_max3_f32 %0, %0, %0, %0" : "+v"(x[i]));
        if (VOP == 3) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %0" : "+v"(x[i]));
      }
    }
    for (int i = 0; i < 16; ++i) sink += x[i];
    asm volatile("s_nop 0" ::"v"(sink));
    t1 = __builtin_amdgcn_s_memtime();
  }
  out[blockIdx.x * 512 + threadIdx.x] = sink;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}
template <class F>
void run(const char* name, F kern) {
  float* d_out; unsigned long* d_cyc; unsigned long h[2048];
  hipMalloc(&d_out, 256 * 512 * 4); hipMalloc(&d_cyc, sizeof(h));
  const char* modes[3] = {"together", "MFMA waves alone", "VALU waves alone"};
  for (int mode = 0; mode < 3; ++mode) {
    hipMemset(d_cyc, 0, sizeof(h));
    hipLaunchKernelGGL(kern, dim3(256), dim3(512), 0, 0, d_out, d_cyc, mode);
    hipDeviceSynchronize();
    hipMemcpy(h, d_cyc, sizeof(h), hipMemcpyDeviceToHost);
    double sm = 0, sv = 0;
    for (int b = 0; b < 256; ++b) for (int w = 0; w < 8; ++w) (w < 4 ? sm : sv) += h[b * 8 + w];
    printf("%-10s %-18s  MFMA wave: %6.1f cycles per MFMA   VALU wave: %6.2f cycles per instruction\n", name, modes[mode], sm / 1024 / (REP * 4), sv / 1024 / (REP * 16));
  }
}
int main() {
  run("v_exp_f32", k<0>); run("v_add_f32", k<1>); run("v_max3_f32", k<2>); run("v_cvt_pk", k<3>);
  return 0;
}
