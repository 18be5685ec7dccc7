// common.h — shared device helpers for the gfx950 kernels (wave64, MFMA 32x32x16 bf16).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/lvdhip.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define LVD_DEV __device__ __forceinline__

LVD_DEV float bf2f(uint16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
// fp32 -> bf16 with the hardware converter (v_cvt_pk_bf16_f32: round-to-nearest-even, one instruction per pair)
LVD_DEV uint16_t f2bf(float f) { return __builtin_bit_cast(uint16_t, (__bf16)f); }
LVD_DEV uint32_t pack2bf(float lo, float hi) {
  bf16x2 v = {(__bf16)lo, (__bf16)hi};
  return __builtin_bit_cast(uint32_t, v);
}
// raw v_exp_f32 (2^x): inputs far below the denormal range flush to 0, which is what the softmax wants
LVD_DEV float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
LVD_DEV float bflo(uint32_t u) { return __uint_as_float(u << 16); }
LVD_DEV float bfhi(uint32_t u) { return __uint_as_float(u & 0xffff0000u); }

LVD_DEV uint4 ldg16(const void* p) { return *reinterpret_cast<const uint4*>(p); }
LVD_DEV uint2 ldg8(const void* p) { return *reinterpret_cast<const uint2*>(p); }
LVD_DEV void stg16(void* p, uint4 v) { *reinterpret_cast<uint4*>(p) = v; }
LVD_DEV void stg8(void* p, uint2 v) { *reinterpret_cast<uint2*>(p) = v; }

LVD_DEV bf16x8 as_bf16x8(uint4 v) { return __builtin_bit_cast(bf16x8, v); }

// SiLU and its derivative with the hardware reciprocal (v_rcp_f32, 1 ulp) instead of an IEEE division (a ten-instruction sequence):
// these run once or twice per element inside kernels that otherwise only stream bf16 rows, and every consumer rounds to bf16
LVD_DEV float sigmoid_f(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
LVD_DEV float silu_f(float x) { return x * sigmoid_f(x); }
LVD_DEV float silu_grad_f(float x) {
  float s = sigmoid_f(x);
  return s * (1.f + x * (1.f - s));
}
// erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, far below the bf16 resolution of every consumer): one v_rcp and one
// v_exp instead of libm's two-regime erff, which made the GEGLU epilogue longer than its GEMM main loop.
LVD_DEV float erf_as_f(float x) {
  const float a = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, a, 1.f));
  float pl = fmaf(1.061405429f, t, -1.453152027f);
  pl = fmaf(pl, t, 1.421413741f);
  pl = fmaf(pl, t, -0.284496736f);
  pl = fmaf(pl, t, 0.254829592f);
  const float e = fast_exp2(-1.4426950408889634f * a * a);
  return copysignf(fmaf(-pl * t, e, 1.f), x);
}
LVD_DEV float gelu_erf_f(float x) { return 0.5f * x * (1.f + erf_as_f(x * 0.70710678118654752f)); }
// GELU for the GEGLU epilogues of the GEMM kernels, two values per lane in packed fp32 (v_pk_fma_f32 / v_pk_mul_f32: two FMAs per issue slot).
// The A&S form above costs ~21 issue slots per value (two quarter-rate transcendentals: v_rcp, v_exp) and the epilogue of a feed-forward
// tile evaluates it 20 480 times per wave: with the GELU compiled out the level-0 / 1 / 2 GEGLU products run 331 -> 274, 255 -> 222,
// 242 -> 219 us (profiles/r06_geglu_epilogue.txt).  Here: Phi(x) = 1/2 + x Q(x^2) on |x| <= 4.25 with Q of degree 8 (weighted minimax
// fit of x Phi(x), tools/gelu_fit.py), x clamped into the interval for Phi only: ~7 slots per value, no transcendental.
// |gelu - exact| <= 3.7e-5 for |x| <= 4.25 and <= 5e-5 out to |x| = 12: an order below the fp16 rounding of the reference's own
// activations (2^-11 relative) and two below the bf16 rounding of the stored product.  The element-wise kernels keep the A&S form.
typedef float f32x2 __attribute__((ext_vector_type(2)));
LVD_DEV f32x2 gelu_pk2(f32x2 x) {
#ifdef LVD_GELU_ABL  // developer ablation (never in the shipped library): what the GELU arithmetic costs inside the GEGLU epilogues
  return x;
#else
  const float L = 4.25f;
  const f32x2 xc = {__builtin_amdgcn_fmed3f(x.x, -L, L), __builtin_amdgcn_fmed3f(x.y, -L, L)};
  const f32x2 s = xc * xc;
  f32x2 q = (f32x2)(4.5474263243860946e-11f);
  q = __builtin_elementwise_fma(q, s, (f32x2)(-4.515573248653482e-09f));
  q = __builtin_elementwise_fma(q, s, (f32x2)(1.9862437738993322e-07f));
  q = __builtin_elementwise_fma(q, s, (f32x2)(-5.147656338522211e-06f));
  q = __builtin_elementwise_fma(q, s, (f32x2)(8.849051664583385e-05f));
  q = __builtin_elementwise_fma(q, s, (f32x2)(-0.0010790855158120394f));
  q = __builtin_elementwise_fma(q, s, (f32x2)(0.009718801826238632f));
  q = __builtin_elementwise_fma(q, s, (f32x2)(-0.06619030982255936f));
  q = __builtin_elementwise_fma(q, s, (f32x2)(0.398819237947464f));
  return x * __builtin_elementwise_fma(xc, q, (f32x2)(0.5f));
#endif
}
// LayerNorm fold of four adjacent columns, rstd * (v - mean * colsum) + bias, as two v_pk_fma_f32 pairs (same roundings as the scalar
// fmaf(rstd, fmaf(-mean, s, v), b) it replaces: 1 issue slot per value instead of 2)
template <class V4>
LVD_DEV V4 ln_fold4(const V4& v, float mean, float rstd, const V4& s, const V4& b) {
  return __builtin_elementwise_fma((V4)(rstd), __builtin_elementwise_fma((V4)(-mean), s, v), b);
}
// hidden * gelu(gate) for four adjacent columns -> four bf16 (the GEGLU of models/attention.py:355-376 as the epilogues apply it)
template <class V4>
LVD_DEV uint2 geglu4(const V4& h, const V4& g) {
  const f32x2 a = (f32x2){h[0], h[1]} * gelu_pk2((f32x2){g[0], g[1]});
  const f32x2 b = (f32x2){h[2], h[3]} * gelu_pk2((f32x2){g[2], g[3]});
  uint2 o;
  o.x = pack2bf(a.x, a.y);
  o.y = pack2bf(b.x, b.y);
  return o;
}
LVD_DEV float gelu_erf_grad_f(float x) {
  float cdf = 0.5f * (1.f + erf_as_f(x * 0.70710678118654752f));
  float pdf = 0.3989422804014327f * __expf(-0.5f * x * x);
  return cdf + x * pdf;
}
// gelu(x) and gelu'(x) together: the exponential inside the erf approximation IS exp(-x^2/2), the density's — one v_exp, one v_rcp and
// one polynomial for both (the GEGLU backward needs both per element)
LVD_DEV void gelu_erf_both(float x, float& val, float& grad) {
  const float a = fabsf(x) * 0.70710678118654752f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, a, 1.f));
  float pl = fmaf(1.061405429f, t, -1.453152027f);
  pl = fmaf(pl, t, 1.421413741f);
  pl = fmaf(pl, t, -0.284496736f);
  pl = fmaf(pl, t, 0.254829592f);
  const float e = fast_exp2(-1.4426950408889634f * a * a);  // exp(-x^2 / 2)
  const float cdf = 0.5f * (1.f + copysignf(fmaf(-pl * t, e, 1.f), x));
  val = x * cdf;
  grad = fmaf(x * 0.3989422804014327f, e, cdf);
}

LVD_DEV float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
LVD_DEV float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// host-side error plumbing (capi.cpp)
void lvd_set_error(const char* fmt, ...);
#define LVD_CHECK(cond, ...)            \
  do {                                  \
    if (!(cond)) {                      \
      lvd_set_error(__VA_ARGS__);       \
      return 1;                         \
    }                                   \
  } while (0)
#define LVD_LAUNCH_CHECK()                                             \
  do {                                                                 \
    hipError_t e_ = hipGetLastError();                                 \
    if (e_ != hipSuccess) {                                            \
      lvd_set_error("launch failed: %s", hipGetErrorString(e_));       \
      return 2;                                                        \
    }                                                                  \
  } while (0)
