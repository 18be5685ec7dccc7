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

// ------------------------------------------------------------------ whole slab in registers (1024 threads)
// The norms in between: a (sample, group) slab of 30-200 KB — the 2-D norms of the 40x72 / 20x36 levels, the 5-D norms of the temporal
// layers on the 5x9 level, the norm over [x, skip] at 2560 channels — is too long a walk for the 256-thread kernel above (20-90 rows per
// thread, one memory round trip per eight), and as two launches of norm.hip it is read twice (and each launch has its floor: 7-10 us for
// a few MB).  Here a workgroup of 1024 threads loads the WHOLE slab at once, all loads of a thread in flight together — one memory round
// trip —, reduces, and normalises what it holds in registers: x is read once and there is one launch.
// Thread = (row lane, vector of V channels of the group); V = 8 / 4 / 2 (16- / 8- / 4-byte loads): the widest that divides the channels
// per group (40 and 80 -> 8, 20 and 60 -> 4, 10 and 30 -> 2); at most 12 / 8 / 8 loads per thread (slab_maxu).
// With many slabs in flight it streams: 88 MB in 32.7 us (5.4 TB/s over read + write) against 49.1 us for the two launches.
// [Longer slabs were tried — 22 loads of 16 bytes per thread, 345 KB per workgroup: the 5-D norms of the 10x18 level — and lose to the
// two launches (28.7 against 21.0 us for 11 MB): one workgroup moves its slab at ~25 GB/s however many loads it has in flight.  On the
// way: a version that spilled 152 bytes per lane took 47 us — scratch limits how many waves the device runs at once.]
template <int V>
struct SlabVec { uint32_t w[V / 2]; };
template <int V>
LVD_DEV SlabVec<V> slab_load(const lvd_bf16* p) {
  SlabVec<V> r;
  if constexpr (V == 8) { const uint4 v = ldg16(p); r.w[0] = v.x; r.w[1] = v.y; r.w[2] = v.z; r.w[3] = v.w; }
  else if constexpr (V == 4) { const uint2 v = ldg8(p); r.w[0] = v.x; r.w[1] = v.y; }
  else r.w[0] = *reinterpret_cast<const uint32_t*>(p);
  return r;
}
template <int V>
LVD_DEV void slab_store(lvd_bf16* p, const SlabVec<V>& r) {
  if constexpr (V == 8) stg16(p, make_uint4(r.w[0], r.w[1], r.w[2], r.w[3]));
  else if constexpr (V == 4) stg8(p, make_uint2(r.w[0], r.w[1]));
  else *reinterpret_cast<uint32_t*>(p) = r.w[0];
}
template <int V>
LVD_DEV void slab_sums(const SlabVec<V>& v, float w, float& s1, float& s2) {
  float a1 = 0.f, a2 = 0.f;
#pragma unroll
  for (int e = 0; e < V / 2; ++e) {
    const float lo = bflo(v.w[e]), hi = bfhi(v.w[e]);
    a1 += lo + hi;
    a2 += lo * lo + hi * hi;
  }
  s1 += w * a1;
  s2 += w * a2;
}
template <int V>
LVD_DEV SlabVec<V> slab_apply(const SlabVec<V>& v, const float (&sc)[V], const float (&sh)[V], int silu) {
  SlabVec<V> o;
#pragma unroll
  for (int e = 0; e < V / 2; ++e) {
    float a = bflo(v.w[e]) * sc[2 * e] + sh[2 * e], b = bfhi(v.w[e]) * sc[2 * e + 1] + sh[2 * e + 1];
    if (silu) { a = silu_f(a); b = silu_f(b); }
    o.w[e] = pack2bf(a, b);
  }
  return o;
}

// OPG = vectors per group row (cpg / V), RLN = row lanes (1024 / OPG)
template <int U, int V>
__global__ __launch_bounds__(1024) void gn_slab_kernel(const lvd_gn_stats_params p, lvd_bf16* __restrict__ y, int ldy, int silu, int OPG, int RLN,
                                                       int samples) {
  constexpr int T = 1024;
  __shared__ float red[2][T / 64];
  __shared__ float gb[2][1024];  // the group's gamma / beta: parked in LDS across the reduction, not in registers
  // Eight or more samples: all groups of a sample on one XCD, as above.  Fewer (the 5-D norms: one or two samples of many rows): that
  // would put the whole tensor through one or two of the eight L2s and their links (10.0 -> 8.1 us at 2.8 MB), so the (sample, group) slabs go
  // round the XCDs in launch order instead (neighbouring groups share a few 128-byte lines across two L2s: cheap).
  int s, g;
  if (samples >= 8) {
    const int xcd = blockIdx.x & 7, k = blockIdx.x >> 3;
    s = (k / p.groups) * 8 + xcd;
    g = k % p.groups;
  } else {
    s = blockIdx.x / p.groups;
    g = blockIdx.x % p.groups;
  }
  if (s >= samples) return;
  const int t = threadIdx.x, rl = t / OPG, j = t - rl * OPG;
  const bool live = rl < RLN;
  const int rps = p.rows_per_sample, cpg = V * OPG;
  const int c = g * cpg + V * j;
  const bool first = c < p.c1;  // a group lies in one source (c1 % cpg == 0)
  const lvd_bf16* xb = first ? p.x1 + c : p.x2 + (c - p.c1);
  const int ldx = first ? p.ld1 : p.ld2;
  const long row0 = (long)s * rps;
  const int r0 = live ? rl : 0;
  SlabVec<V> raw[U];
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int u = 0; u < U; ++u) raw[u] = slab_load<V>(xb + (row0 + min(r0 + u * RLN, rps - 1)) * ldx);  // clamped, not predicated
  for (int i = t; i < cpg; i += T) { gb[0][i] = p.gamma[g * cpg + i]; gb[1][i] = p.beta[g * cpg + i]; }
#pragma unroll
  for (int u = 0; u < U; ++u) slab_sums<V>(raw[u], (live && r0 + u * RLN < rps) ? 1.f : 0.f, s1, s2);  // clamped duplicates, idle lanes: masked
  // the rows stay packed across the reduction (the compiler would rather keep the unpacked floats: twice the registers)
#pragma unroll
  for (int u = 0; u < U; ++u)
#pragma unroll
    for (int e = 0; e < V / 2; ++e) asm volatile("" : "+v"(raw[u].w[e]));
  s1 = wave_sum(s1);
  s2 = wave_sum(s2);
  if ((t & 63) == 0) { red[0][t >> 6] = s1; red[1][t >> 6] = s2; }
  __syncthreads();
  s1 = 0.f; s2 = 0.f;
#pragma unroll
  for (int q = 0; q < T / 64; ++q) { s1 += red[0][q]; s2 += red[1][q]; }
  const float cnt = (float)cpg * (float)rps;
  const float mean = s1 / cnt;
  const float rstd = rsqrtf(fmaxf(s2 / cnt - mean * mean, 0.f) + p.eps);
  if (t == 0 && p.mean_rstd) {
    p.mean_rstd[((long)s * p.groups + g) * 2] = mean;
    p.mean_rstd[((long)s * p.groups + g) * 2 + 1] = rstd;
  }
  if (!live) return;
  float sc[V], sh[V];
#pragma unroll
  for (int e = 0; e < V; ++e) {
    sc[e] = rstd * gb[0][V * j + e];
    sh[e] = gb[1][V * j + e] - mean * sc[e];
  }
  lvd_bf16* yb = y + c;
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int r = r0 + u * RLN;
    if (r >= rps) break;
    slab_store<V>(yb + (row0 + r) * ldy, slab_apply<V>(raw[u], sc, sh, silu));
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

// backward counterpart of gn_slab_kernel: x AND dy of the slab in registers, one launch: sums of (g, g*xhat) over the slab, then
// dx = rstd * (g - mean(g) - xhat * mean(g * xhat)) from what the thread holds (g is formed twice, as in the two-launch kernels).
// gamma / beta are fetched from LDS again for every row, behind a compiler fence: kept in registers across the rows they, the held rows and
// the SiLU-gradient temporaries do not fit.  [Read one by one as volatiles they became flat loads with a full wait each.]
template <int U, int V>
__global__ __launch_bounds__(1024) void gn_bwd_slab_kernel(const lvd_gn_bwd_apply_params p, int OPG, int RLN, int samples) {
  constexpr int T = 1024;
  __shared__ float red[2][T / 64];
  __shared__ float gb[2][1024];
  int s, g;
  if (samples >= 8) {
    const int xcd = blockIdx.x & 7, k = blockIdx.x >> 3;
    s = (k / p.groups) * 8 + xcd;
    g = k % p.groups;
  } else {
    s = blockIdx.x / p.groups;
    g = blockIdx.x % p.groups;
  }
  if (s >= samples) return;
  const int t = threadIdx.x, rl = t / OPG, j = t - rl * OPG;
  const bool live = rl < RLN;
  const int rps = p.rows_per_sample, cpg = V * OPG;
  const int c = g * cpg + V * j;
  const bool first = c < p.c1;
  const lvd_bf16* xb = first ? p.x1 + c : p.x2 + (c - p.c1);
  const int ldx = first ? p.ld1 : p.ld2;
  const lvd_bf16* dyb = p.dy + c;
  const long row0 = (long)s * rps;
  const int r0 = live ? rl : 0;
  SlabVec<V> rx[U], rd[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const long row = row0 + min(r0 + u * RLN, rps - 1);  // clamped, not predicated
    rx[u] = slab_load<V>(xb + row * ldx);
    rd[u] = slab_load<V>(dyb + row * p.lddy);
  }
  for (int i = t; i < cpg; i += T) { gb[0][i] = p.gamma[g * cpg + i]; gb[1][i] = p.beta[g * cpg + i]; }
  const float mean = p.mean_rstd[((long)s * p.groups + g) * 2], rstd = p.mean_rstd[((long)s * p.groups + g) * 2 + 1];
  __syncthreads();
  const float* gam = &gb[0][V * j];
  const float* bet = &gb[1][V * j];
  const int do_silu = p.silu;
  auto fetch = [&](float (&ga)[V], float (&be)[V]) {
    asm volatile("" ::: "memory");
#pragma unroll
    for (int e = 0; e < V; ++e) { ga[e] = gam[e]; be[e] = bet[e]; }
  };
  auto pair = [&](uint32_t xw, uint32_t dw, const float* ga, const float* be, float& xh0, float& xh1, float& g0, float& g1) {
    xh0 = (bflo(xw) - mean) * rstd; xh1 = (bfhi(xw) - mean) * rstd;
    g0 = bflo(dw); g1 = bfhi(dw);
    if (do_silu) { g0 *= silu_grad_f(xh0 * ga[0] + be[0]); g1 *= silu_grad_f(xh1 * ga[1] + be[1]); }
    g0 *= ga[0]; g1 *= ga[1];
  };
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const float w = (live && r0 + u * RLN < rps) ? 1.f : 0.f;
    float ga[V], be[V], a1 = 0.f, a2 = 0.f;
    fetch(ga, be);
#pragma unroll
    for (int e = 0; e < V / 2; ++e) {
      float xh0, xh1, g0, g1;
      pair(rx[u].w[e], rd[u].w[e], ga + 2 * e, be + 2 * e, xh0, xh1, g0, g1);
      a1 += g0 + g1;
      a2 += g0 * xh0 + g1 * xh1;
    }
    s1 += w * a1;
    s2 += w * a2;
  }
#pragma unroll
  for (int u = 0; u < U; ++u)
#pragma unroll
    for (int e = 0; e < V / 2; ++e) asm volatile("" : "+v"(rx[u].w[e]), "+v"(rd[u].w[e]));
  s1 = wave_sum(s1);
  s2 = wave_sum(s2);
  if ((t & 63) == 0) { red[0][t >> 6] = s1; red[1][t >> 6] = s2; }
  __syncthreads();
  if (!live) return;
  s1 = 0.f; s2 = 0.f;
#pragma unroll
  for (int q = 0; q < T / 64; ++q) { s1 += red[0][q]; s2 += red[1][q]; }
  const float cnt = (float)cpg * (float)rps;
  const float m1 = s1 / cnt, m2 = s2 / cnt;
  lvd_bf16* ob = first ? p.dx1 + c : p.dx2 + (c - p.c1);
  const int ldo = first ? p.lddx1 : p.lddx2;
  const int accumulate = p.accumulate;
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int r = r0 + u * RLN;
    if (r >= rps) break;
    SlabVec<V> acc, o;
#pragma unroll
    for (int e = 0; e < V / 2; ++e) acc.w[e] = 0;
    if (accumulate) acc = slab_load<V>(ob + (row0 + r) * ldo);
    float ga[V], be[V];
    fetch(ga, be);
#pragma unroll
    for (int e = 0; e < V / 2; ++e) {
      float xh0, xh1, g0, g1;
      pair(rx[u].w[e], rd[u].w[e], ga + 2 * e, be + 2 * e, xh0, xh1, g0, g1);
      o.w[e] = pack2bf(rstd * (g0 - m1 - xh0 * m2) + bflo(acc.w[e]), rstd * (g1 - m1 - xh1 * m2) + bfhi(acc.w[e]));
    }
    slab_store<V>(ob + (row0 + r) * ldo, o);
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

namespace {
// loads a thread holds, by channels per load: 12 x 16 bytes, 8 x 8 or 4 bytes.  [With 15 narrow loads per thread — the 2-D norms of the 40x72
// level: 10 channels per group, 2880 rows — the kernel ties with the two launches at 44 MB and loses at 88 MB (61.7 against 48.3 us).]
constexpr int slab_maxu(int v) { return v == 8 ? 12 : 8; }
// channels per load (8 / 4 / 2: the widest that divides the channels per group and the row pitches), vectors per group row, row lanes and
// loads per thread of the slab kernel; 0 loads = the shape does not qualify
int slab_geometry(int c, int c1, int groups, int rows_per_sample, int pitch_gcd, int* V, int* OPG, int* RLN) {
  if (groups <= 0 || c % groups) return 0;
  const int cpg = c / groups;
  if (cpg > 1024 || (c1 < c && c1 % cpg)) return 0;
  for (*V = 8; *V >= 2; *V /= 2) {
    if (cpg % *V || pitch_gcd % *V) continue;
    *OPG = cpg / *V;
    *RLN = 1024 / *OPG;
    const int u = (rows_per_sample + *RLN - 1) / *RLN;
    return u <= slab_maxu(*V) ? u : 0;
  }
  return 0;
}
template <int U, int V>
void launch_slab(const lvd_gn_stats_params* s, const lvd_gn_apply_params* a, unsigned grid, int OPG, int RLN, int samples, hipStream_t st) {
  hipLaunchKernelGGL((gn_slab_kernel<U, V>), dim3(grid), dim3(1024), 0, st, *s, a->y, a->ldy, a->silu, OPG, RLN, samples);
}
template <int V>
void launch_slab_u(int u, const lvd_gn_stats_params* s, const lvd_gn_apply_params* a, unsigned grid, int OPG, int RLN, int samples, hipStream_t st) {
  if (u <= 2) launch_slab<2, V>(s, a, grid, OPG, RLN, samples, st);
  else if (u <= 4) launch_slab<4, V>(s, a, grid, OPG, RLN, samples, st);
  else if (u <= 8) launch_slab<8, V>(s, a, grid, OPG, RLN, samples, st);
  else if constexpr (V == 8) launch_slab<12, V>(s, a, grid, OPG, RLN, samples, st);
}
}  // namespace

extern "C" int lvdhip_groupnorm_slab_loads(int32_t c, int32_t c1, int32_t groups, int32_t rows_per_sample) {
  int V, OPG, RLN;
  return slab_geometry(c, c1, groups, rows_per_sample, 8, &V, &OPG, &RLN);
}

extern "C" int lvdhip_groupnorm_slab(const lvd_gn_stats_params* s, const lvd_gn_apply_params* a, void* stream) {
  LVD_CHECK(s && a && s->x1 && a->y && s->gamma && s->beta, "gn_slab: null pointer");
  LVD_CHECK(a->x1 == s->x1 && a->x2 == s->x2 && a->c == s->c && a->c1 == s->c1 && a->rows == s->rows && a->rows_per_sample == s->rows_per_sample,
            "gn_slab: the stats and apply descriptions differ");
  LVD_CHECK(s->rows_per_sample > 0 && s->rows % s->rows_per_sample == 0, "gn_slab: rows %% rows_per_sample");
  LVD_CHECK(s->x2 != nullptr || s->c1 >= s->c, "gn_slab: missing second source");
  LVD_CHECK(s->c1 % 8 == 0 && s->ld1 % 8 == 0 && (s->x2 == nullptr || s->ld2 % 8 == 0) && a->ldy % 8 == 0, "gn_slab: row pitches must be multiples of 8");
  int V, OPG, RLN;
  const int u = slab_geometry(s->c, s->c1, s->groups, s->rows_per_sample, 8, &V, &OPG, &RLN);
  LVD_CHECK(u > 0, "gn_slab: shape does not qualify (c=%d c1=%d groups=%d rows_per_sample=%d; lvdhip_groupnorm_slab_loads)", s->c, s->c1, s->groups,
            s->rows_per_sample);
  const int samples = s->rows / s->rows_per_sample;
  const unsigned grid = (unsigned)(((samples + 7) / 8) * 8 * s->groups);
  hipStream_t st = (hipStream_t)stream;
  if (V == 8) launch_slab_u<8>(u, s, a, grid, OPG, RLN, samples, st);
  else if (V == 4) launch_slab_u<4>(u, s, a, grid, OPG, RLN, samples, st);
  else launch_slab_u<2>(u, s, a, grid, OPG, RLN, samples, st);
  LVD_LAUNCH_CHECK();
  return 0;
}

namespace {
constexpr int slab_bwd_maxu(int v) { return v == 8 ? 2 : (v == 4 ? 4 : 8); }  // pairs of loads (x, dy) a thread holds without spilling: 64 bytes
template <int U, int V>
void launch_bwd_slab(const lvd_gn_bwd_apply_params* p, unsigned grid, int OPG, int RLN, int samples, hipStream_t st) {
  hipLaunchKernelGGL((gn_bwd_slab_kernel<U, V>), dim3(grid), dim3(1024), 0, st, *p, OPG, RLN, samples);
}
template <int V>
void launch_bwd_slab_u(int u, const lvd_gn_bwd_apply_params* p, unsigned grid, int OPG, int RLN, int samples, hipStream_t st) {
  if (u <= 2) launch_bwd_slab<2, V>(p, grid, OPG, RLN, samples, st);
  else if constexpr (V <= 4) {
    if (u <= 4) launch_bwd_slab<4, V>(p, grid, OPG, RLN, samples, st);
    else if constexpr (V == 2) launch_bwd_slab<8, V>(p, grid, OPG, RLN, samples, st);
  }
}
}  // namespace

extern "C" int lvdhip_groupnorm_bwd_slab_loads(int32_t c, int32_t c1, int32_t groups, int32_t rows_per_sample) {
  int V, OPG, RLN;
  const int u = slab_geometry(c, c1, groups, rows_per_sample, 8, &V, &OPG, &RLN);
  return u <= slab_bwd_maxu(V) ? u : 0;
}

extern "C" int lvdhip_groupnorm_bwd_slab(const lvd_gn_bwd_apply_params* p, void* stream) {
  LVD_CHECK(p && p->x1 && p->dy && p->dx1 && p->mean_rstd && p->gamma && p->beta, "gn_bwd_slab: null pointer");
  LVD_CHECK(p->x2 == nullptr || p->dx2 != nullptr, "gn_bwd_slab: dx2 missing");
  LVD_CHECK(p->rows_per_sample > 0 && p->rows % p->rows_per_sample == 0, "gn_bwd_slab: rows %% rows_per_sample");
  LVD_CHECK(p->c1 % 8 == 0 && p->ld1 % 8 == 0 && (p->x2 == nullptr || (p->ld2 % 8 == 0 && p->lddx2 % 8 == 0)) && p->lddy % 8 == 0 && p->lddx1 % 8 == 0,
            "gn_bwd_slab: row pitches must be multiples of 8");
  int V, OPG, RLN;
  const int u = slab_geometry(p->c, p->c1, p->groups, p->rows_per_sample, 8, &V, &OPG, &RLN);
  LVD_CHECK(u > 0 && u <= slab_bwd_maxu(V), "gn_bwd_slab: shape does not qualify (c=%d c1=%d groups=%d rows_per_sample=%d; lvdhip_groupnorm_bwd_slab_loads)",
            p->c, p->c1, p->groups, p->rows_per_sample);
  const int samples = p->rows / p->rows_per_sample;
  const unsigned grid = (unsigned)(((samples + 7) / 8) * 8 * p->groups);
  hipStream_t st = (hipStream_t)stream;
  if (V == 8) launch_bwd_slab_u<8>(u, p, grid, OPG, RLN, samples, st);
  else if (V == 4) launch_bwd_slab_u<4>(u, p, grid, OPG, RLN, samples, st);
  else launch_bwd_slab_u<2>(u, p, grid, OPG, RLN, samples, st);
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
