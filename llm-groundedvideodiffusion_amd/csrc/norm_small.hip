// norm_small.hip — GroupNorm(+SiLU) forward and input-gradient in ONE launch for small samples.
//
// The two-stage kernels of norm.hip (partial sums per row chunk -> finalize -> apply) are right for the 40x72 / 20x36 feature
// maps, where a tensor is tens of MB.  On the deep levels (10x18, 5x9: a whole tensor is 1-10 MB) each of the three launches
// sits at the launch floor and the norm costs three floors.  Here one workgroup owns one (sample, group) slab —
// rows_per_sample x (c/groups) elements, at most 128 KB, so its second pass hits L2 — and does statistics and apply back
// to back.  Thread (row lane, channel pair): the two channels' gamma/beta/scale/shift stay in registers, a row segment of the
// group is read by consecutive lanes, GNS_U rows are in flight per thread.  Workgroups are numbered so that all groups of a
// sample run on one XCD (round-robin dispatch: XCD = workgroup id mod 8): the 128-byte lines that neighbouring groups share
// are fetched into one L2, not eight.
#include "common.h"

namespace {

constexpr int GNS_U = 8;

LVD_DEV void block_sum2(float& a, float& b, float (*red)[4]) {
  a = wave_sum(a);
  b = wave_sum(b);
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = a; red[1][threadIdx.x >> 6] = b; }
  __syncthreads();
  a = red[0][0] + red[0][1] + red[0][2] + red[0][3];
  b = red[1][0] + red[1][1] + red[1][2] + red[1][3];
}

__global__ __launch_bounds__(256) void gn_fused_small_kernel(const lvd_gn_stats_params p, lvd_bf16* __restrict__ y, int ldy, int silu, int hp,
                                                             int RL, int samples) {
  __shared__ float red[2][4];
  const int xcd = blockIdx.x & 7, k = blockIdx.x >> 3;
  const int s = (k / p.groups) * 8 + xcd, g = k % p.groups;
  if (s >= samples) return;
  const int t = threadIdx.x, j = t % hp, rl = t / hp;
  const bool live = rl < RL;
  const int rps = p.rows_per_sample, cpg = 2 * hp;
  const int c = g * cpg + 2 * j;
  const bool first = c < p.c1;
  const lvd_bf16* xb = first ? p.x1 + c : p.x2 + (c - p.c1);
  const int ldx = first ? p.ld1 : p.ld2;
  const long row0 = (long)s * rps;
  float s1 = 0.f, s2 = 0.f;
  if (live) {
    for (int r0 = rl; r0 < rps; r0 += GNS_U * RL) {
      uint32_t raw[GNS_U];
#pragma unroll
      for (int u = 0; u < GNS_U; ++u) raw[u] = *reinterpret_cast<const uint32_t*>(xb + (row0 + min(r0 + u * RL, rps - 1)) * ldx);
#pragma unroll
      for (int u = 0; u < GNS_U; ++u) {
        const float w = r0 + u * RL < rps ? 1.f : 0.f;  // clamped duplicates are masked out of the sums
        const float a = bflo(raw[u]), b = bfhi(raw[u]);
        s1 += w * (a + b);
        s2 += w * (a * a + b * b);
      }
    }
  }
  block_sum2(s1, s2, red);
  const float cnt = (float)cpg * (float)rps;
  const float mean = s1 / cnt;
  const float rstd = rsqrtf(fmaxf(s2 / cnt - mean * mean, 0.f) + p.eps);
  if (t == 0 && p.mean_rstd) {
    p.mean_rstd[((long)s * p.groups + g) * 2] = mean;
    p.mean_rstd[((long)s * p.groups + g) * 2 + 1] = rstd;
  }
  if (!live) return;
  const float sc0 = rstd * p.gamma[c], sc1 = rstd * p.gamma[c + 1];
  const float sh0 = p.beta[c] - mean * sc0, sh1 = p.beta[c + 1] - mean * sc1;
  lvd_bf16* yb = y + c;
  for (int r0 = rl; r0 < rps; r0 += GNS_U * RL) {
    uint32_t raw[GNS_U];
#pragma unroll
    for (int u = 0; u < GNS_U; ++u) raw[u] = *reinterpret_cast<const uint32_t*>(xb + (row0 + min(r0 + u * RL, rps - 1)) * ldx);
#pragma unroll
    for (int u = 0; u < GNS_U; ++u) {
      const int r = r0 + u * RL;
      if (r >= rps) break;
      float a = bflo(raw[u]) * sc0 + sh0, b = bfhi(raw[u]) * sc1 + sh1;
      if (silu) { a = silu_f(a); b = silu_f(b); }
      *reinterpret_cast<uint32_t*>(yb + (row0 + r) * ldy) = pack2bf(a, b);
    }
  }
}

// dx = rstd * (g - mean_g(g) - xhat * mean_g(g * xhat)),  g = dy * silu'(xhat*gamma+beta) * gamma   (norm.hip gn_bwd_*)
__global__ __launch_bounds__(256) void gn_bwd_fused_small_kernel(const lvd_gn_bwd_apply_params p, int hp, int RL, int samples) {
  __shared__ float red[2][4];
  const int xcd = blockIdx.x & 7, k = blockIdx.x >> 3;
  const int s = (k / p.groups) * 8 + xcd, g = k % p.groups;
  if (s >= samples) return;
  const int t = threadIdx.x, j = t % hp, rl = t / hp;
  const bool live = rl < RL;
  const int rps = p.rows_per_sample, cpg = 2 * hp;
  const int c = g * cpg + 2 * j;
  const bool first = c < p.c1;
  const lvd_bf16* xb = first ? p.x1 + c : p.x2 + (c - p.c1);
  const int ldx = first ? p.ld1 : p.ld2;
  const lvd_bf16* dyb = p.dy + c;
  const long row0 = (long)s * rps;
  const float mean = p.mean_rstd[((long)s * p.groups + g) * 2], rstd = p.mean_rstd[((long)s * p.groups + g) * 2 + 1];
  float ga0 = 0.f, ga1 = 0.f, be0 = 0.f, be1 = 0.f;
  if (live) { ga0 = p.gamma[c]; ga1 = p.gamma[c + 1]; be0 = p.beta[c]; be1 = p.beta[c + 1]; }
  constexpr int U = GNS_U / 2;  // two operands per row
  float s1 = 0.f, s2 = 0.f;
  if (live) {
    for (int r0 = rl; r0 < rps; r0 += U * RL) {
      uint32_t rx[U], rd[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const long row = row0 + min(r0 + u * RL, rps - 1);
        rx[u] = *reinterpret_cast<const uint32_t*>(xb + row * ldx);
        rd[u] = *reinterpret_cast<const uint32_t*>(dyb + row * p.lddy);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const float w = r0 + u * RL < rps ? 1.f : 0.f;
        const float xh0 = (bflo(rx[u]) - mean) * rstd, xh1 = (bfhi(rx[u]) - mean) * rstd;
        float g0 = bflo(rd[u]), g1 = bfhi(rd[u]);
        if (p.silu) { g0 *= silu_grad_f(xh0 * ga0 + be0); g1 *= silu_grad_f(xh1 * ga1 + be1); }
        g0 *= ga0 * w; g1 *= ga1 * w;
        s1 += g0 + g1;
        s2 += g0 * xh0 + g1 * xh1;
      }
    }
  }
  block_sum2(s1, s2, red);
  if (!live) return;
  const float cnt = (float)cpg * (float)rps;
  const float m1 = s1 / cnt, m2 = s2 / cnt;
  lvd_bf16* ob = first ? p.dx1 + c : p.dx2 + (c - p.c1);
  const int ldo = first ? p.lddx1 : p.lddx2;
  for (int r0 = rl; r0 < rps; r0 += U * RL) {
    uint32_t rx[U], rd[U], ra[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long row = row0 + min(r0 + u * RL, rps - 1);
      rx[u] = *reinterpret_cast<const uint32_t*>(xb + row * ldx);
      rd[u] = *reinterpret_cast<const uint32_t*>(dyb + row * p.lddy);
      ra[u] = 0;
      if (p.accumulate) ra[u] = *reinterpret_cast<const uint32_t*>(ob + row * ldo);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int r = r0 + u * RL;
      if (r >= rps) break;
      const float xh0 = (bflo(rx[u]) - mean) * rstd, xh1 = (bfhi(rx[u]) - mean) * rstd;
      float g0 = bflo(rd[u]), g1 = bfhi(rd[u]);
      if (p.silu) { g0 *= silu_grad_f(xh0 * ga0 + be0); g1 *= silu_grad_f(xh1 * ga1 + be1); }
      g0 *= ga0; g1 *= ga1;
      const float d0 = rstd * (g0 - m1 - xh0 * m2) + bflo(ra[u]), d1 = rstd * (g1 - m1 - xh1 * m2) + bfhi(ra[u]);
      *reinterpret_cast<uint32_t*>(ob + (row0 + r) * ldo) = pack2bf(d0, d1);
    }
  }
}

int small_geometry(int c, int c1, int groups, int* hp, int* RL) {
  if (groups <= 0 || c % groups) return 1;
  const int cpg = c / groups;
  if (cpg % 2 || c1 % 2 || cpg / 2 > 256) return 1;
  *hp = cpg / 2;
  *RL = 256 / *hp;
  return 0;
}

}  // namespace

extern "C" int lvdhip_groupnorm_fused(const lvd_gn_stats_params* s, const lvd_gn_apply_params* a, void* stream) {
  LVD_CHECK(s && a && s->x1 && a->y && s->gamma && s->beta, "gn_fused: null pointer");
  LVD_CHECK(a->x1 == s->x1 && a->x2 == s->x2 && a->c == s->c && a->c1 == s->c1 && a->rows == s->rows && a->rows_per_sample == s->rows_per_sample,
            "gn_fused: the stats and apply descriptions differ");
  LVD_CHECK(s->rows_per_sample > 0 && s->rows % s->rows_per_sample == 0, "gn_fused: rows %% rows_per_sample");
  LVD_CHECK(s->x2 != nullptr || s->c1 >= s->c, "gn_fused: missing second source");
  int hp, RL;
  LVD_CHECK(small_geometry(s->c, s->c1, s->groups, &hp, &RL) == 0, "gn_fused: channels per group must be even and <= 512 (c=%d groups=%d)", s->c, s->groups);
  const int samples = s->rows / s->rows_per_sample;
  const unsigned grid = (unsigned)(((samples + 7) / 8) * 8 * s->groups);
  hipLaunchKernelGGL(gn_fused_small_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, *s, a->y, a->ldy, a->silu, hp, RL, samples);
  LVD_LAUNCH_CHECK();
  return 0;
}

extern "C" int lvdhip_groupnorm_bwd_fused(const lvd_gn_bwd_apply_params* p, void* stream) {
  LVD_CHECK(p && p->x1 && p->dy && p->dx1 && p->mean_rstd && p->gamma && p->beta, "gn_bwd_fused: null pointer");
  LVD_CHECK(p->x2 == nullptr || p->dx2 != nullptr, "gn_bwd_fused: dx2 missing");
  LVD_CHECK(p->rows_per_sample > 0 && p->rows % p->rows_per_sample == 0, "gn_bwd_fused: rows %% rows_per_sample");
  int hp, RL;
  LVD_CHECK(small_geometry(p->c, p->c1, p->groups, &hp, &RL) == 0, "gn_bwd_fused: channels per group must be even and <= 512");
  const int samples = p->rows / p->rows_per_sample;
  const unsigned grid = (unsigned)(((samples + 7) / 8) * 8 * p->groups);
  hipLaunchKernelGGL(gn_bwd_fused_small_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, *p, hp, RL, samples);
  LVD_LAUNCH_CHECK();
  return 0;
}
