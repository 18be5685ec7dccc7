// guidance_maps.hip — the map-level options of add_ca_loss_per_attn_map_to_loss (utils/guidance.py:209-226): `smooth_attn` (a 3x3 Gaussian
// over the (position, token) plane of every (frame, head) map, reflect-padded) and `attn_renorm` (a second softmax, over the probabilities
// of tokens 1 .. num_tokens-2 times renorm_scale).  Both spread the gradient of the energy over text tokens that are not object tokens, so
// they cannot use the object-token fast path of guidance_loss.hip (probabilities of the object columns only, dQ from those columns).
// They run on WHOLE maps instead, fp32 [rows = frames * heads, P, T] as lvdhip_ca_probs_full writes them:
//     map  = ca_probs_full(Q, K)                       -> [smooth] -> [renorm] -> gather the object columns -> lvdhip_ca_select (unchanged)
//     dmap = scatter of the selected columns' gradient -> [renorm backward] -> [smooth adjoint] -> softmax backward (dS) -> dQ = dS . K
// with dQ = lvdhip_ca_apply_probs(dS, K).  Nothing here is on the timed path (no entry point of the reference switches these options on);
// one thread per (row, position) walks the T <= 128 tokens of its row: plain, deterministic, no atomics.
#include "common.h"

namespace {

constexpr int MAXT = 128;

// 1-D index lists of the reflect-padded 3-tap correlation on n points: forward tap a of target x reads refl(x + a - 1); the adjoint lists,
// for a target x, every (source, tap) pair whose read lands on x
LVD_DEV int refl(int x, int n) { return x < 0 ? -x : (x >= n ? 2 * n - 2 - x : x); }

__global__ void map_smooth_kernel(const float* __restrict__ in, float* __restrict__ out, long rows, int P, int T, const float* __restrict__ w9, int adjoint) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= rows * P) return;
  const long r = idx / P;
  const int i = (int)(idx - r * P);
  float w[9];
#pragma unroll
  for (int q = 0; q < 9; ++q) w[q] = w9[q];
  const float* base = in + r * P * T;
  float* o = out + idx * T;
  if (!adjoint) {
    const float* rowp[3] = {base + (long)refl(i - 1, P) * T, base + (long)i * T, base + (long)refl(i + 1, P) * T};
    for (int j = 0; j < T; ++j) {
      const int jj[3] = {refl(j - 1, T), j, refl(j + 1, T)};
      float acc = 0.f;
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) acc += w[a * 3 + b] * rowp[a][jj[b]];
      o[j] = acc;
    }
    return;
  }
  // adjoint: every forward read (i', a) -> refl(i' + a - 1) that lands on i, times every (j', b) that lands on j
  int si[5], sa[5], ni = 0;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const int s = i - (a - 1);
    if (s >= 0 && s < P) { si[ni] = s; sa[ni] = a; ++ni; }
  }
  if (i == 1) { si[ni] = 0; sa[ni] = 0; ++ni; }              // source 0, tap 0 reads padded index -1 = position 1
  if (i == P - 2) { si[ni] = P - 1; sa[ni] = 2; ++ni; }       // source P-1, tap 2 reads padded index P = position P-2
  for (int j = 0; j < T; ++j) {
    int sj[5], sb[5], nj = 0;
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      const int s = j - (b - 1);
      if (s >= 0 && s < T) { sj[nj] = s; sb[nj] = b; ++nj; }
    }
    if (j == 1) { sj[nj] = 0; sb[nj] = 0; ++nj; }
    if (j == T - 2) { sj[nj] = T - 1; sb[nj] = 2; ++nj; }
    float acc = 0.f;
    for (int x = 0; x < ni; ++x)
      for (int y = 0; y < nj; ++y) acc += w[sa[x] * 3 + sb[y]] * base[(long)si[x] * T + sj[y]];
    o[j] = acc;
  }
}

// mode 0: out[., u] = softmax_u(s * in[., lo + u]) for u < n, 0 beyond (the renormalised map drops token 0 and the tokens from num_tokens-1 on:
//         column u of the output is token lo + u of the input, utils/guidance.py:222-226 with lo = 1, n = num_tokens - 2)
// mode 1: out[., lo + u] = s * y_u * (dout_u - sum_v y_v dout_v), 0 for the tokens outside [lo, lo + n)
__global__ void map_renorm_kernel(const float* __restrict__ in, const float* __restrict__ dout, float* __restrict__ out, long npos, int T, int lo, int n, float s,
                                  int mode) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= npos) return;
  const float* x = in + idx * T + lo;
  float* o = out + idx * T;
  float m = -3.0e38f;
  for (int u = 0; u < n; ++u) m = fmaxf(m, s * x[u]);
  float z = 0.f;
  for (int u = 0; u < n; ++u) z += __expf(s * x[u] - m);
  const float inv = 1.f / z;
  if (mode == 0) {
    for (int u = 0; u < T; ++u) o[u] = u < n ? __expf(s * x[u] - m) * inv : 0.f;
    return;
  }
  const float* d = dout + idx * T;
  float dot = 0.f;
  for (int u = 0; u < n; ++u) dot += __expf(s * x[u] - m) * inv * d[u];
  for (int t = 0; t < T; ++t) {
    const int u = t - lo;
    o[t] = (u >= 0 && u < n) ? s * (__expf(s * x[u] - m) * inv) * (d[u] - dot) : 0.f;
  }
}

// out[r, c, i] = map[r, i, cols[c]]   (the [frames, heads, ntok, P] layout lvdhip_ca_select reads)
__global__ void map_gather_cols_kernel(const float* __restrict__ map, const int* __restrict__ cols, int ncols, float* __restrict__ out, long rows, int P, int T) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= rows * P) return;
  const long r = idx / P;
  const int i = (int)(idx - r * P);
  for (int c = 0; c < ncols; ++c) out[(r * ncols + c) * P + i] = map[idx * T + cols[c]];
}

// dmap[r, i, :] = 0; dmap[r, i, cols[c]] += dcols[r, c, i] in column order (two objects may name the same token)
__global__ void map_scatter_cols_kernel(const float* __restrict__ dcols, const int* __restrict__ cols, int ncols, float* __restrict__ dmap, long rows, int P, int T) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= rows * P) return;
  const long r = idx / P;
  const int i = (int)(idx - r * P);
  float* o = dmap + idx * T;
  for (int t = 0; t < T; ++t) o[t] = 0.f;
  for (int c = 0; c < ncols; ++c) o[cols[c]] += dcols[(r * ncols + c) * P + i];
}

// dS = scale * A o (dA - rowsum(A o dA)): the backward of softmax(scale * Q K^T) w.r.t. the scores, times the score scale
__global__ void map_softmax_bwd_kernel(const float* __restrict__ probs, const float* __restrict__ dprobs, float* __restrict__ ds, long npos, int T, float scale) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= npos) return;
  const float* a = probs + idx * T;
  const float* d = dprobs + idx * T;
  float dot = 0.f;
  for (int t = 0; t < T; ++t) dot += a[t] * d[t];
  for (int t = 0; t < T; ++t) ds[idx * T + t] = scale * a[t] * (d[t] - dot);
}

inline dim3 grid_for(long n) { return dim3((unsigned)((n + 255) / 256)); }

}  // namespace

extern "C" int lvdhip_ca_map_smooth(const float* in, float* out, int64_t rows, int32_t P, int32_t T, const float* w9, int32_t adjoint, void* stream) {
  LVD_CHECK(in && out && w9 && in != out, "ca_map_smooth: null or aliased pointers");
  LVD_CHECK(rows > 0 && P >= 2 && T >= 2 && T <= MAXT, "ca_map_smooth: rows=%ld P=%d T=%d (P, T >= 2: reflect padding; T <= %d)", (long)rows, P, T, MAXT);
  hipLaunchKernelGGL(map_smooth_kernel, grid_for(rows * P), dim3(256), 0, (hipStream_t)stream, in, out, (long)rows, P, T, w9, adjoint);
  LVD_LAUNCH_CHECK();
  return 0;
}

extern "C" int lvdhip_ca_map_renorm(const float* in, const float* dout, float* out, int64_t rows, int32_t P, int32_t T, int32_t tok_lo, int32_t tok_n,
                                    float renorm_scale, int32_t backward, void* stream) {
  LVD_CHECK(in && out && (!backward || dout), "ca_map_renorm: null pointer");
  LVD_CHECK(rows > 0 && P > 0 && tok_lo >= 0 && tok_n >= 1 && tok_lo + tok_n <= T && T <= MAXT, "ca_map_renorm: tokens [%d, %d) outside the map's %d", tok_lo,
            tok_lo + tok_n, T);
  hipLaunchKernelGGL(map_renorm_kernel, grid_for(rows * P), dim3(256), 0, (hipStream_t)stream, in, dout, out, (long)rows * P, T, tok_lo, tok_n, renorm_scale,
                     backward);
  LVD_LAUNCH_CHECK();
  return 0;
}

extern "C" int lvdhip_ca_map_gather_cols(const float* map, const int32_t* cols, int32_t ncols, float* out, int64_t rows, int32_t P, int32_t T, void* stream) {
  LVD_CHECK(map && cols && out && rows > 0 && P > 0 && T > 0 && ncols > 0, "ca_map_gather_cols: bad arguments");
  hipLaunchKernelGGL(map_gather_cols_kernel, grid_for(rows * P), dim3(256), 0, (hipStream_t)stream, map, cols, ncols, out, (long)rows, P, T);
  LVD_LAUNCH_CHECK();
  return 0;
}

extern "C" int lvdhip_ca_map_scatter_cols(const float* dcols, const int32_t* cols, int32_t ncols, float* dmap, int64_t rows, int32_t P, int32_t T, void* stream) {
  LVD_CHECK(dcols && cols && dmap && rows > 0 && P > 0 && T > 0 && ncols > 0, "ca_map_scatter_cols: bad arguments");
  hipLaunchKernelGGL(map_scatter_cols_kernel, grid_for(rows * P), dim3(256), 0, (hipStream_t)stream, dcols, cols, ncols, dmap, (long)rows, P, T);
  LVD_LAUNCH_CHECK();
  return 0;
}

extern "C" int lvdhip_ca_map_softmax_bwd(const float* probs, const float* dprobs, float* ds, int64_t rows, int32_t P, int32_t T, float scale, void* stream) {
  LVD_CHECK(probs && dprobs && ds && rows > 0 && P > 0 && T > 0, "ca_map_softmax_bwd: bad arguments");
  hipLaunchKernelGGL(map_softmax_bwd_kernel, grid_for(rows * P), dim3(256), 0, (hipStream_t)stream, probs, dprobs, ds, (long)rows * P, T, scale);
  LVD_LAUNCH_CHECK();
  return 0;
}
