// elementwise.hip — layout, activation and sampler-update kernels (all HBM-bound, vectorised).
#include "common.h"

namespace {

#define GRID_STRIDE(i, n) for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += (long)gridDim.x * blockDim.x)

inline int nblocks(long n, int per = 256) {
  long b = (n + per - 1) / per;
  if (b > 16384) b = 16384;
  if (b < 1) b = 1;
  return (int)b;
}

// latents (B,C,F,HW) fp32 -> tokens [(b,f,p), cpad] bf16
__global__ void latents_to_tokens_kernel(const float* lat, lvd_bf16* tok, int B, int C, int F, int HW, int cpad, float scale) {
  long n = (long)B * F * HW;
  GRID_STRIDE(i, n) {
    long b = i / ((long)F * HW);
    long rem = i - b * F * HW;
    int f = (int)(rem / HW), px = (int)(rem - (long)f * HW);
    for (int c = 0; c < cpad; ++c) {
      float v = c < C ? lat[((b * C + c) * F + f) * HW + px] * scale : 0.f;
      tok[i * cpad + c] = f2bf(v);
    }
  }
}

__global__ void tokens_to_latents_kernel(const float* tok, int ld, float* lat, int B, int C, int F, int HW) {
  long n = (long)B * C * F * HW;
  GRID_STRIDE(i, n) {
    long px = i % HW;
    long t = i / HW;
    int f = (int)(t % F); t /= F;
    int c = (int)(t % C);
    long b = t / C;
    lat[i] = tok[((b * F + f) * HW + px) * ld + c];
  }
}

__global__ void tokens_grad_to_latents_kernel(const lvd_bf16* tok, int ld, float* lat, int B, int C, int F, int HW, float scale) {
  long n = (long)B * C * F * HW;
  GRID_STRIDE(i, n) {
    long px = i % HW;
    long t = i / HW;
    int f = (int)(t % F); t /= F;
    int c = (int)(t % C);
    long b = t / C;
    lat[i] = bf2f(tok[((b * F + f) * HW + px) * ld + c]) * scale;
  }
}

__global__ void add_kernel(const lvd_bf16* a, int lda, const lvd_bf16* b, int ldb, lvd_bf16* y, int ldy, int rows, int c) {
  int vpr = c >> 3;
  long n = (long)rows * vpr;
  GRID_STRIDE(i, n) {
    long row = i / vpr;
    int cc = (int)(i - row * vpr) * 8;
    uint4 x = ldg16(a + row * lda + cc), z = ldg16(b + row * ldb + cc);
    uint4 o;
    o.x = pack2bf(bflo(x.x) + bflo(z.x), bfhi(x.x) + bfhi(z.x));
    o.y = pack2bf(bflo(x.y) + bflo(z.y), bfhi(x.y) + bfhi(z.y));
    o.z = pack2bf(bflo(x.z) + bflo(z.z), bfhi(x.z) + bfhi(z.z));
    o.w = pack2bf(bflo(x.w) + bflo(z.w), bfhi(x.w) + bfhi(z.w));
    stg16(y + row * ldy + cc, o);
  }
}

// Temporal (3,1,1) convolution as an N-expanded product + this combine (engine._tconv, small M): Y[m, t*N + n] = x[m] . W_t[n] for the three
// taps at once (a plain-loader GEMM with 3N output columns: three times the tiles, no K split), then
//   out[m, n] = bias[n] + res[m, n] + Y[m - hw, n] (frame > 0) + Y[m, N + n] + Y[m + hw, 2N + n] (frame < F - 1),   summed in fp32 in tap order.
__global__ void tconv_combine_kernel(const float* y, int ldy, const float* bias, const lvd_bf16* res, int ldres, lvd_bf16* out, int ldo,
                                     int rows, int n, int frames, int hw, int accumulate) {
  const int vpr = n >> 2;
  const long tot = (long)rows * vpr;
  GRID_STRIDE(i, tot) {
    const long row = i / vpr;
    const int c = (int)(i - row * vpr) * 4;
    const int f = (int)((row / hw) % frames);
    // clamped addresses + masks instead of branches around the loads
    const long rp = f > 0 ? row - hw : row, rn = f < frames - 1 ? row + hw : row;
    const f32x4 a = *reinterpret_cast<const f32x4*>(y + rp * ldy + c);
    const f32x4 b = *reinterpret_cast<const f32x4*>(y + row * ldy + n + c);
    const f32x4 d = *reinterpret_cast<const f32x4*>(y + rn * ldy + 2 * n + c);
    f32x4 v = (f > 0 ? 1.f : 0.f) * a;
    v += b;
    v += (f < frames - 1 ? 1.f : 0.f) * d;
    if (bias) v += *reinterpret_cast<const f32x4*>(bias + c);
    if (res) {
      const uint2 r = ldg8(res + row * ldres + c);
      v[0] += bflo(r.x); v[1] += bfhi(r.x); v[2] += bflo(r.y); v[3] += bfhi(r.y);
    }
    lvd_bf16* o = out + row * ldo + c;
    if (accumulate) {
      const uint2 r = ldg8(o);
      v[0] += bflo(r.x); v[1] += bfhi(r.x); v[2] += bflo(r.y); v[3] += bfhi(r.y);
    }
    uint2 w;
    w.x = pack2bf(v[0], v[1]);
    w.y = pack2bf(v[2], v[3]);
    stg8(o, w);
  }
}

// pre: [rows, 2*n_out] with hidden/gate interleaved in blocks of 32 (the GEMM's W' row order)
__global__ void geglu_fwd_kernel(const lvd_bf16* pre, int ldp, lvd_bf16* y, int ldy, int rows, int n_out) {
  int vpr = n_out >> 2;
  long n = (long)rows * vpr;
  GRID_STRIDE(i, n) {
    long row = i / vpr;
    int oc = (int)(i - row * vpr) * 4;
    int blk = oc >> 5, w = oc & 31;
    const lvd_bf16* pr = pre + row * ldp + blk * 64 + w;
    uint2 h = ldg8(pr), g = ldg8(pr + 32);
    uint2 o;
    o.x = pack2bf(bflo(h.x) * gelu_erf_f(bflo(g.x)), bfhi(h.x) * gelu_erf_f(bfhi(g.x)));
    o.y = pack2bf(bflo(h.y) * gelu_erf_f(bflo(g.y)), bfhi(h.y) * gelu_erf_f(bfhi(g.y)));
    stg8(y + row * ldy + oc, o);
  }
}

__global__ void geglu_bwd_kernel(const lvd_bf16* pre, int ldp, const lvd_bf16* dy, int lddy, lvd_bf16* dpre, int lddp, int rows, int n_out) {
  int vpr = n_out >> 2;
  long n = (long)rows * vpr;
  GRID_STRIDE(i, n) {
    long row = i / vpr;
    int oc = (int)(i - row * vpr) * 4;
    int blk = oc >> 5, w = oc & 31;
    const lvd_bf16* pr = pre + row * ldp + blk * 64 + w;
    uint2 h = ldg8(pr), g = ldg8(pr + 32), d = ldg8(dy + row * lddy + oc);
    float hv[4] = {bflo(h.x), bfhi(h.x), bflo(h.y), bfhi(h.y)};
    float gv[4] = {bflo(g.x), bfhi(g.x), bflo(g.y), bfhi(g.y)};
    float dv[4] = {bflo(d.x), bfhi(d.x), bflo(d.y), bfhi(d.y)};
    float dh[4], dg[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float ge, gg;
      gelu_erf_both(gv[e], ge, gg);
      dh[e] = dv[e] * ge;
      dg[e] = dv[e] * hv[e] * gg;
    }
    lvd_bf16* dp = dpre + row * lddp + blk * 64 + w;
    uint2 o1, o2;
    o1.x = pack2bf(dh[0], dh[1]); o1.y = pack2bf(dh[2], dh[3]);
    o2.x = pack2bf(dg[0], dg[1]); o2.y = pack2bf(dg[2], dg[3]);
    stg8(dp, o1);
    stg8(dp + 32, o2);
  }
}

// dx[n,y,x,:] (+)= sum of the 2x2 dy block
__global__ void upsample2x_bwd_kernel(const lvd_bf16* dy, lvd_bf16* dx, int nimg, int h, int w, int c, int accumulate) {
  int vpr = c >> 3;
  long n = (long)nimg * h * w * vpr;
  GRID_STRIDE(i, n) {
    long pix = i / vpr;
    int cc = (int)(i - pix * vpr) * 8;
    int x = (int)(pix % w);
    long t = pix / w;
    int y = (int)(t % h);
    long im = t / h;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int dyy = 0; dyy < 2; ++dyy)
#pragma unroll
      for (int dxx = 0; dxx < 2; ++dxx) {
        long src = ((im * (2 * h) + (2 * y + dyy)) * (2 * w) + (2 * x + dxx));
        uint4 r = ldg16(dy + src * c + cc);
        acc[0] += bflo(r.x); acc[1] += bfhi(r.x); acc[2] += bflo(r.y); acc[3] += bfhi(r.y);
        acc[4] += bflo(r.z); acc[5] += bfhi(r.z); acc[6] += bflo(r.w); acc[7] += bfhi(r.w);
      }
    lvd_bf16* o = dx + pix * c + cc;
    if (accumulate) {
      uint4 r = ldg16(o);
      acc[0] += bflo(r.x); acc[1] += bfhi(r.x); acc[2] += bflo(r.y); acc[3] += bfhi(r.y);
      acc[4] += bflo(r.z); acc[5] += bfhi(r.z); acc[6] += bflo(r.w); acc[7] += bfhi(r.w);
    }
    uint4 wv;
    wv.x = pack2bf(acc[0], acc[1]); wv.y = pack2bf(acc[2], acc[3]); wv.z = pack2bf(acc[4], acc[5]); wv.w = pack2bf(acc[6], acc[7]);
    stg16(o, wv);
  }
}

// diffusers Timesteps(dim, flip_sin_to_cos=True, downscale_freq_shift=0): [cos | sin]
__global__ void timestep_embedding_kernel(const float* t, lvd_bf16* out, int n, int dim) {
  int half = dim / 2;
  long tot = (long)n * half;
  GRID_STRIDE(i, tot) {
    int r = (int)(i / half), j = (int)(i % half);
    float freq = __expf(-9.210340371976184f * (float)j / (float)half);
    float a = t[r] * freq;
    out[(long)r * dim + j] = f2bf(cosf(a));
    out[(long)r * dim + half + j] = f2bf(sinf(a));
  }
}

__global__ void silu_kernel(const lvd_bf16* x, lvd_bf16* y, long n) {
  GRID_STRIDE(i, n) y[i] = f2bf(silu_f(bf2f(x[i])));
}

__global__ void cfg_dpm_step_kernel(const float* eu, const float* ec, float gs, float* x, float* x0p, float alpha_t,
                                    float sigma_t, float c_x, float c_0, float c_1, long n) {
  GRID_STRIDE(i, n) {
    float e = eu[i] + gs * (ec[i] - eu[i]);
    float xv = x[i];
    float x0 = (xv - sigma_t * e) / alpha_t;
    float xn = c_x * xv + c_0 * x0 + c_1 * x0p[i];
    x0p[i] = x0;
    x[i] = xn;
  }
}

__global__ void axpy_kernel(float* x, const float* g, float scale, long n) {
  GRID_STRIDE(i, n) x[i] -= scale * g[i];
}

// single-block deterministic reduction
__global__ void reduce_sum_kernel(const float* x, long n, float scale, float* out) {
  __shared__ float red[16];
  float s = 0.f;
  for (long i = threadIdx.x; i < n; i += blockDim.x) s += x[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += red[w];
    out[0] = t * scale;
  }
}

// ---- row softmax of an fp32 score matrix -> bf16 probabilities (VAE mid-block attention: one head of dim 512, scores
// by the GEMM family, AutoencoderKL's Attention block).  One wave per row, scores kept in registers (cols <= 64*SM_MAXV).
constexpr int SM_MAXV = 64;
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ x, int ldx, lvd_bf16* __restrict__ y, int ldy, int rows, int cols) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + row * ldx;
  float v[SM_MAXV];
  float m = -3.0e38f;
#pragma unroll
  for (int j = 0; j < SM_MAXV; ++j) {
    const int c = lane + 64 * j;
    v[j] = c < cols ? xr[c] : -3.0e38f;
    m = fmaxf(m, v[j]);
  }
  m = wave_max(m);
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < SM_MAXV; ++j) {
    v[j] = fast_exp2((v[j] - m) * 1.4426950408889634f);
    s += v[j];
  }
  const float inv = 1.f / wave_sum(s);
  lvd_bf16* yr = y + row * ldy;
#pragma unroll
  for (int j = 0; j < SM_MAXV; ++j) {
    const int c = lane + 64 * j;
    if (c < cols) yr[c] = f2bf(v[j] * inv);
  }
}

// Rows wider than the register budget (the 72x128 = 9216-token mid attention of a 1024x576 decode / encode): one 256-thread
// block per row, three passes over the row (it stays in L2: 37 KB), block reductions through LDS.
__global__ __launch_bounds__(256) void softmax_rows_wide_kernel(const float* __restrict__ x, int ldx, lvd_bf16* __restrict__ y, int ldy, int cols) {
  __shared__ float red[4];
  const float* xr = x + (long)blockIdx.x * ldx;
  lvd_bf16* yr = y + (long)blockIdx.x * ldy;
  const int t = threadIdx.x;
  float m = -3.0e38f;
  for (int c = t; c < cols; c += 256) m = fmaxf(m, xr[c]);
  m = wave_max(m);
  if ((t & 63) == 0) red[t >> 6] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float s = 0.f;
  for (int c = t; c < cols; c += 256) s += fast_exp2((xr[c] - m) * 1.4426950408889634f);
  s = wave_sum(s);
  if ((t & 63) == 0) red[t >> 6] = s;
  __syncthreads();
  const float inv = 1.f / (red[0] + red[1] + red[2] + red[3]);
  for (int c = t; c < cols; c += 256) yr[c] = f2bf(fast_exp2((xr[c] - m) * 1.4426950408889634f) * inv);
}

// ---- decoded image tokens [(f,y,x), >=3 channels] bf16 -> video float32 (f, y, x, 3) in [0,1]:  x/2 + 0.5 clamped
// (VaeImageProcessor.postprocess + tensor2vid, controllable_pipeline_text_to_video_synth.py:66-88,374-400)
__global__ void tokens_to_video_kernel(const lvd_bf16* __restrict__ t, int ld, float* __restrict__ video, long rows) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < rows; i += (long)gridDim.x * blockDim.x) {
    uint2 r = ldg8(t + i * ld);
    float c0 = bflo(r.x), c1 = bfhi(r.x), c2 = bflo(r.y);
    video[i * 3 + 0] = fminf(fmaxf(c0 * 0.5f + 0.5f, 0.f), 1.f);
    video[i * 3 + 1] = fminf(fmaxf(c1 * 0.5f + 0.5f, 0.f), 1.f);
    video[i * 3 + 2] = fminf(fmaxf(c2 * 0.5f + 0.5f, 0.f), 1.f);
  }
}

// ---- GELU (CLIP text MLP): exact erf form or quick_gelu
__global__ void gelu_kernel(const lvd_bf16* __restrict__ x, lvd_bf16* __restrict__ y, long n, int mode) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float v = bf2f(x[i]);
    y[i] = f2bf(mode == 0 ? gelu_erf_f(v) : v / (1.f + __expf(-1.702f * v)));
  }
}

}  // namespace

#define ST ((hipStream_t)stream)

extern "C" int lvdhip_latents_to_tokens(const float* latents, lvd_bf16* tokens, int32_t B, int32_t C, int32_t F, int32_t HW,
                                        int32_t cpad, float scale, void* stream) {
  LVD_CHECK(latents && tokens && cpad >= C, "latents_to_tokens: bad args");
  long n = (long)B * F * HW;
  hipLaunchKernelGGL(latents_to_tokens_kernel, dim3(nblocks(n)), dim3(256), 0, ST, latents, tokens, B, C, F, HW, cpad, scale);
  LVD_LAUNCH_CHECK();
  return 0;
}
extern "C" int lvdhip_tokens_to_latents(const float* tokens, int32_t ld, float* latents, int32_t B, int32_t C, int32_t F,
                                        int32_t HW, void* stream) {
  LVD_CHECK(latents && tokens, "tokens_to_latents: bad args");
  long n = (long)B * C * F * HW;
  hipLaunchKernelGGL(tokens_to_latents_kernel, dim3(nblocks(n)), dim3(256), 0, ST, tokens, ld, latents, B, C, F, HW);
  LVD_LAUNCH_CHECK();
  return 0;
}
extern "C" int lvdhip_tokens_grad_to_latents(const lvd_bf16* tokens, int32_t ld, float* latents, int32_t B, int32_t C,
                                             int32_t F, int32_t HW, float scale, void* stream) {
  LVD_CHECK(latents && tokens, "tokens_grad_to_latents: bad args");
  long n = (long)B * C * F * HW;
  hipLaunchKernelGGL(tokens_grad_to_latents_kernel, dim3(nblocks(n)), dim3(256), 0, ST, tokens, ld, latents, B, C, F, HW, scale);
  LVD_LAUNCH_CHECK();
  return 0;
}
extern "C" int lvdhip_add(const lvd_bf16* a, int32_t lda, const lvd_bf16* b, int32_t ldb, lvd_bf16* y, int32_t ldy, int32_t rows,
                          int32_t c, void* stream) {
  LVD_CHECK(a && b && y && c % 8 == 0, "add: bad args");
  long n = (long)rows * (c / 8);
  hipLaunchKernelGGL(add_kernel, dim3(nblocks(n)), dim3(256), 0, ST, a, lda, b, ldb, y, ldy, rows, c);
  LVD_LAUNCH_CHECK();
  return 0;
}
extern "C" int lvdhip_tconv_combine(const float* y, int32_t ldy, const float* bias, const lvd_bf16* res, int32_t ldres, lvd_bf16* out, int32_t ldo,
                                    int32_t rows, int32_t n, int32_t frames, int32_t hw, int32_t accumulate, void* stream) {
  LVD_CHECK(y && out && n % 4 == 0 && ldy % 4 == 0 && ldo % 4 == 0 && (!res || ldres % 4 == 0), "tconv_combine: bad args");
  LVD_CHECK(frames >= 1 && hw >= 1 && rows % (frames * hw) == 0, "tconv_combine: rows %d is not a multiple of frames * hw = %d", rows, frames * hw);
  long tot = (long)rows * (n / 4);
  hipLaunchKernelGGL(tconv_combine_kernel, dim3(nblocks(tot)), dim3(256), 0, ST, y, ldy, bias, res, ldres, out, ldo, rows, n, frames, hw, accumulate);
  LVD_LAUNCH_CHECK();
  return 0;
}
extern "C" int lvdhip_geglu_fwd(const lvd_bf16* pre, int32_t ldp, lvd_bf16* y, int32_t ldy, int32_t rows, int32_t n_out, void* stream) {
  LVD_CHECK(pre && y && n_out % 32 == 0, "geglu_fwd: bad args");
  long n = (long)rows * (n_out / 4);
  hipLaunchKernelGGL(geglu_fwd_kernel, dim3(nblocks(n)), dim3(256), 0, ST, pre, ldp, y, ldy, rows, n_out);
  LVD_LAUNCH_CHECK();
  return 0;
}
extern "C" int lvdhip_geglu_bwd(const lvd_bf16* pre, int32_t ldp, const lvd_bf16* dy, int32_t lddy, lvd_bf16* dpre, int32_t lddp,
                                int32_t rows, int32_t n_out, void* stream) {
  LVD_CHECK(pre && dy && dpre && n_out % 32 == 0, "geglu_bwd: bad args");
  long n = (long)rows * (n_out / 4);
  hipLaunchKernelGGL(geglu_bwd_kernel, dim3(nblocks(n)), dim3(256), 0, ST, pre, ldp, dy, lddy, dpre, lddp, rows, n_out);
  LVD_LAUNCH_CHECK();
  return 0;
}
extern "C" int lvdhip_upsample2x_bwd(const lvd_bf16* dy, lvd_bf16* dx, int32_t n, int32_t h, int32_t w, int32_t c, int32_t accumulate,
                                     void* stream) {
  LVD_CHECK(dy && dx && c % 8 == 0, "upsample2x_bwd: bad args");
  long tot = (long)n * h * w * (c / 8);
  hipLaunchKernelGGL(upsample2x_bwd_kernel, dim3(nblocks(tot)), dim3(256), 0, ST, dy, dx, n, h, w, c, accumulate);
  LVD_LAUNCH_CHECK();
  return 0;
}
extern "C" int lvdhip_timestep_embedding(const float* t, lvd_bf16* out, int32_t n, int32_t dim, void* stream) {
  LVD_CHECK(t && out && dim % 2 == 0, "timestep_embedding: bad args");
  hipLaunchKernelGGL(timestep_embedding_kernel, dim3(nblocks((long)n * dim / 2)), dim3(256), 0, ST, t, out, n, dim);
  LVD_LAUNCH_CHECK();
  return 0;
}
extern "C" int lvdhip_gelu(const lvd_bf16* x, lvd_bf16* y, int64_t n, int32_t mode, void* stream) {
  LVD_CHECK(x && y && n > 0 && (mode == 0 || mode == 1), "gelu: bad arguments");
  long blocks = (n + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(gelu_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, y, (long)n, mode);
  LVD_LAUNCH_CHECK();
  return 0;
}

extern "C" int lvdhip_silu(const lvd_bf16* x, lvd_bf16* y, int64_t n, void* stream) {
  LVD_CHECK(x && y, "silu: bad args");
  hipLaunchKernelGGL(silu_kernel, dim3(nblocks(n)), dim3(256), 0, ST, x, y, (long)n);
  LVD_LAUNCH_CHECK();
  return 0;
}
extern "C" int lvdhip_cfg_dpm_step(const float* eps_uncond, const float* eps_cond, float guidance_scale, float* x, float* x0_prev,
                                   float alpha_t, float sigma_t, float c_x, float c_0, float c_1, int64_t n, void* stream) {
  LVD_CHECK(eps_uncond && eps_cond && x && x0_prev, "cfg_dpm_step: bad args");
  hipLaunchKernelGGL(cfg_dpm_step_kernel, dim3(nblocks(n)), dim3(256), 0, ST, eps_uncond, eps_cond, guidance_scale, x, x0_prev,
                     alpha_t, sigma_t, c_x, c_0, c_1, (long)n);
  LVD_LAUNCH_CHECK();
  return 0;
}
extern "C" int lvdhip_axpy(float* x, const float* g, float scale, int64_t n, void* stream) {
  LVD_CHECK(x && g, "axpy: bad args");
  hipLaunchKernelGGL(axpy_kernel, dim3(nblocks(n)), dim3(256), 0, ST, x, g, scale, (long)n);
  LVD_LAUNCH_CHECK();
  return 0;
}
extern "C" int lvdhip_reduce_sum(const float* x, int64_t n, float scale, float* out, void* stream) {
  LVD_CHECK(x && out, "reduce_sum: bad args");
  hipLaunchKernelGGL(reduce_sum_kernel, dim3(1), dim3(1024), 0, ST, x, (long)n, scale, out);
  LVD_LAUNCH_CHECK();
  return 0;
}

extern "C" int lvdhip_softmax_rows(const float* x, int32_t ldx, lvd_bf16* y, int32_t ldy, int32_t rows, int32_t cols, void* stream) {
  LVD_CHECK(x && y && rows > 0 && cols > 0, "softmax_rows: bad arguments");
  if (cols <= 64 * SM_MAXV)
    hipLaunchKernelGGL(softmax_rows_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, ldx, y, ldy, rows, cols);
  else
    hipLaunchKernelGGL(softmax_rows_wide_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, x, ldx, y, ldy, cols);
  LVD_LAUNCH_CHECK();
  return 0;
}

extern "C" int lvdhip_tokens_to_video(const lvd_bf16* tokens, int32_t ld, float* video, int64_t rows, void* stream) {
  LVD_CHECK(tokens && video && rows > 0 && ld >= 4 && ld % 4 == 0, "tokens_to_video: bad arguments (ld=%d)", ld);
  long blocks = (rows + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(tokens_to_video_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, tokens, ld, video, (long)rows);
  LVD_LAUNCH_CHECK();
  return 0;
}
