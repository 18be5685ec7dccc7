// norm.hip — GroupNorm(+SiLU) and LayerNorm, forward and input-gradient, on token matrices.
//
// HBM-bound kernels (roofline: bytes moved / 8 TB/s).  GroupNorm over a channels-last token
// matrix: a "sample" is rows_per_sample consecutive rows (H*W rows for the 2-D norms of
// ResnetBlock2D / Transformer2DModel, F*H*W rows for the 5-D norms of TemporalConvLayer /
// TransformerTemporalModel — the (B·F,C,H,W)->(B,C,F,H,W) permute of the reference,
// models/transformer_temporal.py:148-153, is just a different rows_per_sample here).
// Statistics are two-stage and deterministic: per-(row chunk, group) partial sums (coalesced 16-byte loads, each
// thread owns 8 channels; the channels of a group are folded inside the workgroup), and the APPLY kernel folds the
// chunks of its sample itself (fixed order, a few KB from L2 per workgroup) — there is no finalize launch between
// the two passes (round 3: 2328 launches of 5.7 us per 8 steps that did nothing else).
#include <cstdlib>
#include "common.h"

namespace {

LVD_DEV void load4(const lvd_bf16* x1, const lvd_bf16* x2, int ld1, int ld2, int c1, long row, int c, float v[4]) {
  uint2 r = (c < c1) ? ldg8(x1 + row * ld1 + c) : ldg8(x2 + row * ld2 + (c - c1));
  v[0] = bflo(r.x); v[1] = bfhi(r.x); v[2] = bflo(r.y); v[3] = bfhi(r.y);
}
LVD_DEV void unpack8(uint4 r, float v[8]) {
  v[0] = bflo(r.x); v[1] = bfhi(r.x); v[2] = bflo(r.y); v[3] = bfhi(r.y);
  v[4] = bflo(r.z); v[5] = bfhi(r.z); v[6] = bflo(r.w); v[7] = bfhi(r.w);
}
LVD_DEV void load8(const lvd_bf16* x1, const lvd_bf16* x2, int ld1, int ld2, int c1, long row, int c, float v[8]) {
  unpack8((c < c1) ? ldg16(x1 + row * ld1 + c) : ldg16(x2 + row * ld2 + (c - c1)), v);
}

// Fold the RL row-lanes of a block and then the channels of each group: red is [RL][VC][16] (8 first-moment, 8 second-moment
// sums per 8-channel vector); out = the (sum1, sum2) pair of every group for this (sample, chunk).
LVD_DEV void gn_fold_rows_groups(float* red, float s1[8], float s2[8], int VC, int RL, int vcid, int rl, bool live, int c, int groups,
                                 float* out) {
  if (live) {
#pragma unroll
    for (int e = 0; e < 8; ++e) { red[(rl * VC + vcid) * 16 + e] = s1[e]; red[(rl * VC + vcid) * 16 + 8 + e] = s2[e]; }
  }
  __syncthreads();
  if (live && rl == 0) {
    for (int q = 1; q < RL; ++q)
#pragma unroll
      for (int e = 0; e < 8; ++e) { s1[e] += red[(q * VC + vcid) * 16 + e]; s2[e] += red[(q * VC + vcid) * 16 + 8 + e]; }
    // row lane 0 owns slot [0][vcid] (nobody else reads it): per-channel totals of the chunk
#pragma unroll
    for (int e = 0; e < 8; ++e) { red[vcid * 16 + e] = s1[e]; red[vcid * 16 + 8 + e] = s2[e]; }
  }
  __syncthreads();
  const int cpg = c / groups;
  for (int g = threadIdx.x; g < groups; g += blockDim.x) {
    float a = 0.f, b = 0.f;
    for (int cc = 0; cc < cpg; ++cc) {
      const int ch = g * cpg + cc;
      a += red[(ch >> 3) * 16 + (ch & 7)];
      b += red[(ch >> 3) * 16 + 8 + (ch & 7)];
    }
    out[g * 2] = a; out[g * 2 + 1] = b;
  }
}

// Totals of sample s over its row chunks, for every group, by the whole block, in a fixed order (thread (part, g) adds the chunks
// part, part + parts, ...; then the parts are added in sequence): tot[g] = (sum1, sum2).  `partial` is [samples][chunks][groups][2].
constexpr int GN_PARTS = 16, GN_MAXG = 128;
LVD_DEV void gn_fold_chunks(const float* partial, int chunks, int groups, int s, float2* tot) {
  __shared__ float2 scratch[GN_PARTS * GN_MAXG];
  const int T = blockDim.x;
  int parts = T / groups;
  parts = parts < 1 ? 1 : (parts > GN_PARTS ? GN_PARTS : parts);
  if (parts > chunks) parts = chunks;
  const float2* base = reinterpret_cast<const float2*>(partial) + (long)s * chunks * groups;
  for (int idx = threadIdx.x; idx < parts * groups; idx += T) {
    const int part = idx / groups, g = idx - part * groups;
    float a = 0.f, b = 0.f;
    // eight chunks of this thread in flight at once (clamped index, masked value): the partials come from the memory side and the fold
    // sits in front of the whole workgroup's work — at most two round trips (256 chunks over 16 parts)
    if (chunks > 4 * parts) {
      for (int ch0 = part; ch0 < chunks; ch0 += 8 * parts) {
        float2 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = base[(long)min(ch0 + u * parts, chunks - 1) * groups + g];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const float w = ch0 + u * parts < chunks ? 1.f : 0.f;
          a += w * v[u].x; b += w * v[u].y;
        }
      }
    } else {
      float2 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = base[(long)min(part + u * parts, chunks - 1) * groups + g];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float w = part + u * parts < chunks ? 1.f : 0.f;
        a += w * v[u].x; b += w * v[u].y;
      }
    }
    scratch[idx] = make_float2(a, b);
  }
  __syncthreads();
  for (int g = threadIdx.x; g < groups; g += T) {
    float2 t = scratch[g];
    for (int q = 1; q < parts; ++q) { t.x += scratch[q * groups + g].x; t.y += scratch[q * groups + g].y; }
    tot[g] = t;
  }
  __syncthreads();
}

// ------------------------------------------------------------------ GroupNorm forward stats
// grid (chunks, samples); block = VC*RL threads (VC = c/8 channel-octets: 16-byte loads, RL row lanes)
template <int U>  // rows in flight per thread and trip: 8 when a row lane walks >= 8 rows (few, fat chunks: the 5-D norms), else 4
__global__ void gn_partial_kernel(const lvd_gn_stats_params p, int VC, int RL) {
  extern __shared__ float red[];  // [RL][VC][16]
  const int t = threadIdx.x;
  const int vcid = t % VC, rl = t / VC;
  const int chunk = blockIdx.x, s = blockIdx.y;
  const int rps = p.rows_per_sample;
  const int rpc = (rps + p.chunks - 1) / p.chunks;
  const int rbeg = chunk * rpc, rend = min(rps, rbeg + rpc);
  const int c = vcid * 8;
  float s1[8] = {0, 0, 0, 0, 0, 0, 0, 0}, s2[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const bool live = rl < RL;
  if (live) {
    // four rows per trip, row index clamped instead of predicated (a load inside a lane branch is waited for at the end of
    // the branch, one memory round trip per row); the clamped duplicates are masked out of the sums
    const bool first = c < p.c1;  // select the source pointer once: one load instruction, no branch around it
    const lvd_bf16* xb = first ? p.x1 + c : p.x2 + (c - p.c1);
    const int ldx = first ? p.ld1 : p.ld2;
    for (int r0 = rbeg + rl; r0 < rend; r0 += U * RL) {
      uint4 raw[U];
#pragma unroll
      for (int u = 0; u < U; ++u) raw[u] = ldg16(xb + ((long)s * rps + min(r0 + u * RL, rend - 1)) * ldx);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        float v[8];
        unpack8(raw[u], v);
        const float w = r0 + u * RL < rend ? 1.f : 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) { s1[e] += w * v[e]; s2[e] += w * v[e] * v[e]; }
      }
    }
  }
  gn_fold_rows_groups(red, s1, s2, VC, RL, vcid, rl, live, p.c, p.groups, p.partial + ((long)s * p.chunks + chunk) * p.groups * 2);
}

// ------------------------------------------------------------------ GroupNorm apply (+SiLU)
// grid (chunks, samples); block = VC*RL threads.  Prologue: the block folds the chunk partials of its sample into the group
// statistics (gn_fold_chunks) and every thread forms scale = rstd*gamma, shift = beta - mean*rstd*gamma of its 8 channels in
// registers; then it walks rows.  Block (0, s) also writes mean / rstd of sample s for the backward.
__global__ __launch_bounds__(1024, 6) void gn_apply_kernel(const lvd_gn_apply_params p, int VC, int RL, int chunks) {
  __shared__ float2 tot[GN_MAXG];
  const int t = threadIdx.x;
  const int vcid = t % VC, rl = t / VC;
  const int chunk = blockIdx.x, s = blockIdx.y;
  const int rps = p.rows_per_sample;
  const int cpg = p.c / p.groups;
  const bool live = rl < RL;
  const int c = live ? vcid * 8 : 0;
  // gamma / beta do not depend on the fold: their loads overlap it
  f32x4 gv[2], bv[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    gv[q] = *reinterpret_cast<const f32x4*>(p.gamma + c + 4 * q);
    bv[q] = *reinterpret_cast<const f32x4*>(p.beta + c + 4 * q);
  }
  // the first trip's rows do not depend on the statistics either
  const int rpc = (rps + chunks - 1) / chunks;
  const int rbeg = chunk * rpc, rend = min(rps, rbeg + rpc);
  const bool first = c < p.c1;
  const lvd_bf16* xb = first ? p.x1 + c : p.x2 + (c - p.c1);
  const int ldx = first ? p.ld1 : p.ld2;
  uint4 raw[4];
  {
    const int r0 = min(rbeg + (live ? rl : 0), rend - 1);
#pragma unroll
    for (int u = 0; u < 4; ++u) raw[u] = ldg16(xb + ((long)s * rps + min(r0 + u * RL, rend - 1)) * ldx);  // clamped, not predicated
  }
  gn_fold_chunks(p.partial, p.chunks, p.groups, s, tot);
  const float cnt = (float)cpg * (float)rps;
  if (chunk == 0 && p.mean_rstd) {
    for (int g = t; g < p.groups; g += blockDim.x) {
      const float mean = tot[g].x / cnt;
      const float var = fmaxf(tot[g].y / cnt - mean * mean, 0.f);
      p.mean_rstd[((long)s * p.groups + g) * 2] = mean;
      p.mean_rstd[((long)s * p.groups + g) * 2 + 1] = rsqrtf(var + p.eps);
    }
  }
  if (!live) return;
  float sc[8], sh[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float2 ab = tot[(c + e) / cpg];
    const float mean = ab.x / cnt;
    const float var = fmaxf(ab.y / cnt - mean * mean, 0.f);
    const float rstd = rsqrtf(var + p.eps);
    const float ga = gv[e >> 2][e & 3], be = bv[e >> 2][e & 3];
    sc[e] = rstd * ga;
    sh[e] = be - mean * rstd * ga;
  }
  // four rows per trip: the loads are issued together (the stores may alias them for all the compiler knows)
  for (int r0 = rbeg + rl; r0 < rend; r0 += 4 * RL) {
    if (r0 != rbeg + rl) {
#pragma unroll
      for (int u = 0; u < 4; ++u) raw[u] = ldg16(xb + ((long)s * rps + min(r0 + u * RL, rend - 1)) * ldx);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int r = r0 + u * RL;
      if (r >= rend) break;
      float y[8];
      unpack8(raw[u], y);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        y[e] = y[e] * sc[e] + sh[e];
        if (p.silu) y[e] = silu_f(y[e]);
      }
      uint4 o;
      o.x = pack2bf(y[0], y[1]); o.y = pack2bf(y[2], y[3]); o.z = pack2bf(y[4], y[5]); o.w = pack2bf(y[6], y[7]);
      stg16(p.y + ((long)s * rps + r) * p.ldy + c, o);
    }
  }
}

// ------------------------------------------------------------------ GroupNorm backward
// per-(chunk, group) partials of (g, g*xhat), g = dy * silu'(yhat) * gamma
__global__ void gn_bwd_partial_kernel(const lvd_gn_bwd_stats_params p, int VC, int RL) {
  extern __shared__ float red[];
  const int t = threadIdx.x;
  const int vcid = t % VC, rl = t / VC;
  const int chunk = blockIdx.x, s = blockIdx.y;
  const int rps = p.rows_per_sample;
  const int rpc = (rps + p.chunks - 1) / p.chunks;
  const int rbeg = chunk * rpc, rend = min(rps, rbeg + rpc);
  const int c = vcid * 8;
  const int cpg = p.c / p.groups;
  float s1[8] = {0, 0, 0, 0, 0, 0, 0, 0}, s2[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const bool live = rl < RL;
  if (live) {
    float mean[8], rstd[8], ga[8], be[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      int g = (c + e) / cpg;
      mean[e] = p.mean_rstd[((long)s * p.groups + g) * 2];
      rstd[e] = p.mean_rstd[((long)s * p.groups + g) * 2 + 1];
      ga[e] = p.gamma[c + e]; be[e] = p.beta[c + e];
    }
    const bool first = c < p.c1;
    const lvd_bf16* xb = first ? p.x1 + c : p.x2 + (c - p.c1);
    const int ldx = first ? p.ld1 : p.ld2;
    for (int r0 = rbeg + rl; r0 < rend; r0 += 2 * RL) {
      uint4 rx[2], rdy[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {  // clamped rows, duplicates masked below
        const long row = (long)s * rps + min(r0 + u * RL, rend - 1);
        rx[u] = ldg16(xb + row * ldx);
        rdy[u] = ldg16(p.dy + row * p.lddy + c);
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        float v[8], dy[8];
        unpack8(rx[u], v);
        unpack8(rdy[u], dy);
        const float w = r0 + u * RL < rend ? 1.f : 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float xh = (v[e] - mean[e]) * rstd[e];
          float g = dy[e];
          if (p.silu) g *= silu_grad_f(xh * ga[e] + be[e]);
          g *= ga[e] * w;
          s1[e] += g; s2[e] += g * xh;
        }
      }
    }
  }
  gn_fold_rows_groups(red, s1, s2, VC, RL, vcid, rl, live, p.c, p.groups, p.partial + ((long)s * p.chunks + chunk) * p.groups * 2);
}

// grid (chunks, samples); the block folds the backward partials of its sample first (as gn_apply_kernel does for the forward)
__global__ void gn_bwd_apply_kernel(const lvd_gn_bwd_apply_params p, int VC, int RL, int chunks) {
  __shared__ float2 tot[GN_MAXG];
  const int t = threadIdx.x;
  const int vcid = t % VC, rl = t / VC;
  const int chunk = blockIdx.x, s = blockIdx.y;
  const int rps = p.rows_per_sample;
  const int cpg = p.c / p.groups;
  const bool live = rl < RL;
  const int c = live ? vcid * 8 : 0;
  float mean[8], rstd[8], m1[8], m2[8], ga[8], be[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int g = (c + e) / cpg;
    float2 mr = *reinterpret_cast<const float2*>(p.mean_rstd + ((long)s * p.groups + g) * 2);
    mean[e] = mr.x; rstd[e] = mr.y;
    ga[e] = p.gamma[c + e]; be[e] = p.beta[c + e];
  }
  gn_fold_chunks(p.partial, p.chunks, p.groups, s, tot);
  if (!live) return;
  const float cnt = (float)cpg * (float)rps;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float2 ab = tot[(c + e) / cpg];
    m1[e] = ab.x / cnt; m2[e] = ab.y / cnt;
  }
  const int rpc = (rps + chunks - 1) / chunks;
  const int rbeg = chunk * rpc, rend = min(rps, rbeg + rpc);
  const bool first = c < p.c1;
  const lvd_bf16* xb = first ? p.x1 + c : p.x2 + (c - p.c1);
  const int ldx = first ? p.ld1 : p.ld2;
  lvd_bf16* ob = first ? p.dx1 + c : p.dx2 + (c - p.c1);
  const int ldo = first ? p.lddx1 : p.lddx2;
  for (int r0 = rbeg + rl; r0 < rend; r0 += 2 * RL) {
    uint4 rx[2], rdy[2], rac[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const long row = (long)s * rps + min(r0 + u * RL, rend - 1);  // clamped, not predicated
      rx[u] = ldg16(xb + row * ldx);
      rdy[u] = ldg16(p.dy + row * p.lddy + c);
      rac[u] = make_uint4(0, 0, 0, 0);
      if (p.accumulate) rac[u] = ldg16(ob + row * ldo);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int r = r0 + u * RL;
      if (r >= rend) break;
      float v[8], dy[8], dx[8];
      unpack8(rx[u], v);
      unpack8(rdy[u], dy);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float xh = (v[e] - mean[e]) * rstd[e];
        float gg = dy[e];
        if (p.silu) gg *= silu_grad_f(xh * ga[e] + be[e]);
        gg *= ga[e];
        dx[e] = rstd[e] * (gg - m1[e] - xh * m2[e]);
      }
      if (p.accumulate) {
        float q[8];
        unpack8(rac[u], q);
#pragma unroll
        for (int e = 0; e < 8; ++e) dx[e] += q[e];
      }
      uint4 w;
      w.x = pack2bf(dx[0], dx[1]); w.y = pack2bf(dx[2], dx[3]); w.z = pack2bf(dx[4], dx[5]); w.w = pack2bf(dx[6], dx[7]);
      stg16(ob + ((long)s * rps + r) * ldo, w);
    }
  }
}

// ------------------------------------------------------------------ LayerNorm (one wave per row)
constexpr int LN_MAXV = 4;  // 8-channel vectors per lane -> c <= 2048

__global__ void ln_fwd_kernel(const lvd_ln_params p) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= p.rows) return;
  const int nv = p.c >> 3;
  float x[LN_MAXV][8];
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < LN_MAXV; ++j) {
    int v = lane + 64 * j;
    if (v < nv) {
      uint4 r = ldg16(p.x + row * p.ldx + v * 8);
      x[j][0] = bflo(r.x); x[j][1] = bfhi(r.x); x[j][2] = bflo(r.y); x[j][3] = bfhi(r.y);
      x[j][4] = bflo(r.z); x[j][5] = bfhi(r.z); x[j][6] = bflo(r.w); x[j][7] = bfhi(r.w);
#pragma unroll
      for (int e = 0; e < 8; ++e) s += x[j][e];
    }
  }
  float mean = wave_sum(s) / (float)p.c;
  float q = 0.f;
#pragma unroll
  for (int j = 0; j < LN_MAXV; ++j) {
    int v = lane + 64 * j;
    if (v < nv) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { float d = x[j][e] - mean; q += d * d; }
    }
  }
  float rstd = rsqrtf(wave_sum(q) / (float)p.c + p.eps);
  if (lane == 0 && p.mean_rstd) { p.mean_rstd[row * 2] = mean; p.mean_rstd[row * 2 + 1] = rstd; }
  if (!p.y) return;  // statistics only
#pragma unroll
  for (int j = 0; j < LN_MAXV; ++j) {
    int v = lane + 64 * j;
    if (v < nv) {
      float y[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) y[e] = (x[j][e] - mean) * rstd * p.gamma[v * 8 + e] + p.beta[v * 8 + e];
      uint4 o;
      o.x = pack2bf(y[0], y[1]); o.y = pack2bf(y[2], y[3]); o.z = pack2bf(y[4], y[5]); o.w = pack2bf(y[6], y[7]);
      stg16(p.y + row * p.ldy + v * 8, o);
    }
  }
}

// LayerNorm forward for the model's channel counts (C = 40·LPR, LPR = 8 / 16 / 32 lanes per row: 320 / 640 / 1280): a wave
// holds 64/LPR rows at once, every lane five 16-byte vectors of its row (consecutive lanes = consecutive 16 bytes: full
// 128-byte lines), reductions are log2(LPR) shuffles, and gamma/beta stay in registers over LN_BATCH row batches.  The
// one-wave-per-row kernel above keeps 40 of 64 lanes busy at C = 320, one load in flight per lane, and re-reads gamma /
// beta (4x the bytes of the row itself) for every row.
constexpr int LN_BATCH = 4;
// YOUT = false: statistics only (p.y == NULL) — the (mean, rstd) rows a LayerNorm-folded GEMM reads (lvd_gemm_params.ln_mean_rstd)
template <int LPR, bool YOUT = true>
__global__ __launch_bounds__(256) void ln_fwd_rows_kernel(const lvd_ln_params p, const int nb) {
  constexpr int RPW = 64 / LPR;  // rows per wave per batch
  const int lane = threadIdx.x & 63;
  const int r = lane / LPR, c = lane % LPR;
  const long row0 = ((long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * (RPW * nb);
  float ga[YOUT ? 5 : 1][8], be[YOUT ? 5 : 1][8];
#pragma unroll
  for (int j = 0; j < (YOUT ? 5 : 0); ++j) {
    const int ch = (c + j * LPR) * 8;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const f32x4 g = *reinterpret_cast<const f32x4*>(p.gamma + ch + 4 * q);
      const f32x4 b = *reinterpret_cast<const f32x4*>(p.beta + ch + 4 * q);
#pragma unroll
      for (int e = 0; e < 4; ++e) { ga[j][4 * q + e] = g[e]; be[j][4 * q + e] = b[e]; }
    }
  }
  const float inv_c = 1.f / (float)p.c;
#pragma unroll 1
  for (int bt = 0; bt < nb; ++bt) {
    const long row = row0 + bt * RPW + r;
    const long rowc = row < p.rows ? row : p.rows - 1;  // clamped, not predicated (the store is)
    uint4 raw[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) raw[j] = ldg16(p.x + rowc * p.ldx + (c + j * LPR) * 8);
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      float v[8];
      unpack8(raw[j], v);
#pragma unroll
      for (int e = 0; e < 8; ++e) s += v[e];
    }
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    const float mean = s * inv_c;
    float q2 = 0.f;
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      float v[8];
      unpack8(raw[j], v);
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float d = v[e] - mean; q2 += d * d; }
    }
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) q2 += __shfl_xor(q2, o, 64);
    const float rstd = rsqrtf(q2 * inv_c + p.eps);
    if (row < p.rows) {
      if (c == 0 && p.mean_rstd) { p.mean_rstd[row * 2] = mean; p.mean_rstd[row * 2 + 1] = rstd; }
#pragma unroll
      for (int j = 0; j < (YOUT ? 5 : 0); ++j) {
        float v[8];
        unpack8(raw[j], v);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (v[e] - mean) * rstd * ga[j][e] + be[j][e];
        uint4 o;
        o.x = pack2bf(v[0], v[1]); o.y = pack2bf(v[2], v[3]); o.z = pack2bf(v[4], v[5]); o.w = pack2bf(v[6], v[7]);
        stg16(p.y + row * p.ldy + (c + j * LPR) * 8, o);
      }
    }
  }
}

__global__ void ln_bwd_kernel(const lvd_ln_bwd_params p) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= p.rows) return;
  const int nv = p.c >> 3;
  const float mean = p.mean_rstd[row * 2], rstd = p.mean_rstd[row * 2 + 1];
  float xh[LN_MAXV][8], g[LN_MAXV][8];
  float a = 0.f, b = 0.f;
#pragma unroll
  for (int j = 0; j < LN_MAXV; ++j) {
    int v = lane + 64 * j;
    if (v < nv) {
      uint4 r = ldg16(p.x + row * p.ldx + v * 8);
      uint4 d = ldg16(p.dy + row * p.lddy + v * 8);
      float xv[8] = {bflo(r.x), bfhi(r.x), bflo(r.y), bfhi(r.y), bflo(r.z), bfhi(r.z), bflo(r.w), bfhi(r.w)};
      float dv[8] = {bflo(d.x), bfhi(d.x), bflo(d.y), bfhi(d.y), bflo(d.z), bfhi(d.z), bflo(d.w), bfhi(d.w)};
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        xh[j][e] = (xv[e] - mean) * rstd;
        g[j][e] = dv[e] * p.gamma[v * 8 + e];
        a += g[j][e]; b += g[j][e] * xh[j][e];
      }
    }
  }
  float m1 = wave_sum(a) / (float)p.c, m2 = wave_sum(b) / (float)p.c;
#pragma unroll
  for (int j = 0; j < LN_MAXV; ++j) {
    int v = lane + 64 * j;
    if (v < nv) {
      float dx[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) dx[e] = rstd * (g[j][e] - m1 - xh[j][e] * m2);
      lvd_bf16* o = p.dx + row * p.lddx + v * 8;
      if (p.accumulate) {
        uint4 r = ldg16(o);
        dx[0] += bflo(r.x); dx[1] += bfhi(r.x); dx[2] += bflo(r.y); dx[3] += bfhi(r.y);
        dx[4] += bflo(r.z); dx[5] += bfhi(r.z); dx[6] += bflo(r.w); dx[7] += bfhi(r.w);
      }
      uint4 w;
      w.x = pack2bf(dx[0], dx[1]); w.y = pack2bf(dx[2], dx[3]); w.z = pack2bf(dx[4], dx[5]); w.w = pack2bf(dx[6], dx[7]);
      stg16(o, w);
    }
  }
}

// backward counterpart of ln_fwd_rows_kernel (same row/lane mapping, gamma resident over the row batches)
template <int LPR>
__global__ __launch_bounds__(256) void ln_bwd_rows_kernel(const lvd_ln_bwd_params p, const int nb) {
  constexpr int RPW = 64 / LPR;
  const int lane = threadIdx.x & 63;
  const int r = lane / LPR, c = lane % LPR;
  const long row0 = ((long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * (RPW * nb);
  float ga[5][8];
#pragma unroll
  for (int j = 0; j < 5; ++j)
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const f32x4 g = *reinterpret_cast<const f32x4*>(p.gamma + (c + j * LPR) * 8 + 4 * q);
#pragma unroll
      for (int e = 0; e < 4; ++e) ga[j][4 * q + e] = g[e];
    }
  const float inv_c = 1.f / (float)p.c;
#pragma unroll 1
  for (int bt = 0; bt < nb; ++bt) {
    const long row = row0 + bt * RPW + r;
    const long rowc = row < p.rows ? row : p.rows - 1;
    uint4 rx[5], rd[5], ra[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      rx[j] = ldg16(p.x + rowc * p.ldx + (c + j * LPR) * 8);
      rd[j] = ldg16(p.dy + rowc * p.lddy + (c + j * LPR) * 8);
    }
    if (p.accumulate) {
#pragma unroll
      for (int j = 0; j < 5; ++j) ra[j] = ldg16(p.dx + rowc * p.lddx + (c + j * LPR) * 8);
    }
    const float2 mr = *reinterpret_cast<const float2*>(p.mean_rstd + rowc * 2);
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      float xv[8], dv[8];
      unpack8(rx[j], xv);
      unpack8(rd[j], dv);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float g = dv[e] * ga[j][e];
        a += g;
        b += g * (xv[e] - mr.x) * mr.y;
      }
    }
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
    const float m1 = a * inv_c, m2 = b * inv_c;
    if (row < p.rows) {
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        float xv[8], dv[8], dx[8];
        unpack8(rx[j], xv);
        unpack8(rd[j], dv);
#pragma unroll
        for (int e = 0; e < 8; ++e) dx[e] = mr.y * (dv[e] * ga[j][e] - m1 - (xv[e] - mr.x) * mr.y * m2);
        if (p.accumulate) {
          float q[8];
          unpack8(ra[j], q);
#pragma unroll
          for (int e = 0; e < 8; ++e) dx[e] += q[e];
        }
        uint4 w;
        w.x = pack2bf(dx[0], dx[1]); w.y = pack2bf(dx[2], dx[3]); w.z = pack2bf(dx[4], dx[5]); w.w = pack2bf(dx[6], dx[7]);
        stg16(p.dx + row * p.lddx + (c + j * LPR) * 8, w);
      }
    }
  }
}

// row chunks per sample for the elementwise passes: ~3 workgroups per CU over the whole launch (one resident round: every workgroup
// of a sample repeats the fold of its statistics partials, so few fat workgroups beat many thin ones), >= 4 rows per row lane
int gn_row_chunks(int samples, int rows_per_sample, int RL) {
  static const int target = [] { const char* e = getenv("LVD_GN_APPLY_WGS"); return e ? atoi(e) : 768; }();  // developer knob
  int want = (target + samples - 1) / samples;
  if (want > 512) want = 512;
  int most = rows_per_sample / (4 * RL);
  if (want > most) want = most;
  return want < 1 ? 1 : want;
}

int gn_geometry(int c, int* VC, int* RL, int* threads) {
  *VC = c / 8;
  if (*VC > 1024) return 1;
  *RL = 512 / *VC;
  if (*RL < 1) *RL = 1;
  *threads = ((*VC * *RL + 63) / 64) * 64;
  return 0;
}

}  // namespace

extern "C" int lvdhip_groupnorm_stats(const lvd_gn_stats_params* p, void* stream) {
  LVD_CHECK(p && p->x1 && p->partial, "gn_stats: null pointer");
  LVD_CHECK(p->groups >= 1 && p->groups <= GN_MAXG, "gn_stats: groups=%d unsupported (1..%d)", p->groups, GN_MAXG);
  LVD_CHECK(p->c % 8 == 0 && p->c1 % 8 == 0 && p->c % p->groups == 0, "gn_stats: bad channels c=%d c1=%d groups=%d", p->c, p->c1, p->groups);
  LVD_CHECK(p->rows % p->rows_per_sample == 0 && p->chunks > 0, "gn_stats: rows %% rows_per_sample");
  LVD_CHECK(p->x2 != nullptr || p->c1 >= p->c, "gn_stats: missing second source");
  int VC, RL, threads;
  LVD_CHECK(gn_geometry(p->c, &VC, &RL, &threads) == 0, "gn_stats: c too large");
  int samples = p->rows / p->rows_per_sample;
  hipStream_t s = (hipStream_t)stream;
  const int rpc = (p->rows_per_sample + p->chunks - 1) / p->chunks;
  if (rpc >= 8 * RL) hipLaunchKernelGGL(gn_partial_kernel<8>, dim3(p->chunks, samples), dim3(threads), VC * RL * 16 * sizeof(float), s, *p, VC, RL);
  else hipLaunchKernelGGL(gn_partial_kernel<4>, dim3(p->chunks, samples), dim3(threads), VC * RL * 16 * sizeof(float), s, *p, VC, RL);
  LVD_LAUNCH_CHECK();
  return 0;
}

extern "C" int lvdhip_groupnorm_apply(const lvd_gn_apply_params* p, void* stream) {
  LVD_CHECK(p && p->x1 && p->y && p->partial && p->gamma && p->beta, "gn_apply: null pointer");
  LVD_CHECK(p->c % 8 == 0 && p->c1 % 8 == 0, "gn_apply: channels must be multiples of 8");
  LVD_CHECK(p->groups >= 1 && p->groups <= GN_MAXG && p->c % p->groups == 0 && p->chunks > 0, "gn_apply: bad groups=%d / chunks=%d", p->groups, p->chunks);
  LVD_CHECK(p->rows % p->rows_per_sample == 0, "gn_apply: rows %% rows_per_sample");
  int VC, RL, threads;
  LVD_CHECK(gn_geometry(p->c, &VC, &RL, &threads) == 0, "gn_apply: c too large");
  const int samples = p->rows / p->rows_per_sample;
  const int chunks = gn_row_chunks(samples, p->rows_per_sample, RL);
  hipLaunchKernelGGL(gn_apply_kernel, dim3(chunks, samples), dim3(threads), 0, (hipStream_t)stream, *p, VC, RL, chunks);
  LVD_LAUNCH_CHECK();
  return 0;
}

extern "C" int lvdhip_groupnorm_bwd_stats(const lvd_gn_bwd_stats_params* p, void* stream) {
  LVD_CHECK(p && p->x1 && p->dy && p->partial && p->mean_rstd, "gn_bwd_stats: null pointer");
  LVD_CHECK(p->c % 8 == 0 && p->c1 % 8 == 0 && p->groups >= 1 && p->groups <= GN_MAXG && p->c % p->groups == 0, "gn_bwd_stats: bad channels");
  int VC, RL, threads;
  LVD_CHECK(gn_geometry(p->c, &VC, &RL, &threads) == 0, "gn_bwd_stats: c too large");
  int samples = p->rows / p->rows_per_sample;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(gn_bwd_partial_kernel, dim3(p->chunks, samples), dim3(threads), VC * RL * 16 * sizeof(float), s, *p, VC, RL);
  LVD_LAUNCH_CHECK();
  return 0;
}

extern "C" int lvdhip_groupnorm_bwd_apply(const lvd_gn_bwd_apply_params* p, void* stream) {
  LVD_CHECK(p && p->x1 && p->dy && p->dx1 && p->partial && p->mean_rstd, "gn_bwd_apply: null pointer");
  LVD_CHECK(p->groups >= 1 && p->groups <= GN_MAXG && p->c % p->groups == 0 && p->chunks > 0, "gn_bwd_apply: bad groups=%d / chunks=%d", p->groups, p->chunks);
  LVD_CHECK(p->x2 == nullptr || p->dx2 != nullptr, "gn_bwd_apply: dx2 missing");
  LVD_CHECK(p->c % 8 == 0 && p->c1 % 8 == 0 && p->rows % p->rows_per_sample == 0, "gn_bwd_apply: bad shape");
  int VC, RL, threads;
  LVD_CHECK(gn_geometry(p->c, &VC, &RL, &threads) == 0, "gn_bwd_apply: c too large");
  const int samples = p->rows / p->rows_per_sample;
  const int chunks = gn_row_chunks(samples, p->rows_per_sample, RL);
  hipLaunchKernelGGL(gn_bwd_apply_kernel, dim3(chunks, samples), dim3(threads), 0, (hipStream_t)stream, *p, VC, RL, chunks);
  LVD_LAUNCH_CHECK();
  return 0;
}

namespace {
// Row batches a wave walks in sequence (gamma / beta stay in registers over them): LN_BATCH when that still leaves two workgroups per
// CU, fewer for the short matrices of the deep levels — at 1080-8640 rows four sequential round trips per wave on 34-270 workgroups
// were the launch (6 us for < 1 us of traffic).
int ln_batches(int rows, int lpr) {
  static const int forced = [] { const char* e = getenv("LVD_LN_BATCHES"); return e ? atoi(e) : 0; }();  // developer knob
  if (forced >= 1 && forced <= LN_BATCH) return forced;
  const int rows_per_batch_block = 4 * (64 / lpr);
  for (int nb = LN_BATCH; nb > 1; nb /= 2)
    if (rows / (rows_per_batch_block * nb) >= 512) return nb;
  return 1;
}
}  // namespace

extern "C" int lvdhip_layernorm(const lvd_ln_params* p, void* stream) {
  LVD_CHECK(p && p->x && ((p->y && p->gamma && p->beta) || p->mean_rstd), "layernorm: null pointer");  // y == NULL: statistics only
  LVD_CHECK(p->c % 8 == 0 && p->c <= 512 * LN_MAXV, "layernorm: c=%d unsupported (need c%%8==0, c<=%d)", p->c, 512 * LN_MAXV);
  hipStream_t s = (hipStream_t)stream;
  const int lpr = p->c % 40 == 0 ? p->c / 40 : 0;
  if (lpr == 8 || lpr == 16 || lpr == 32) {
    const int nb = ln_batches(p->rows, lpr);
    const int rows_per_block = 4 * (64 / lpr) * nb;
    const dim3 grid((unsigned)((p->rows + rows_per_block - 1) / rows_per_block));
    if (!p->y) {
      if (lpr == 8) hipLaunchKernelGGL((ln_fwd_rows_kernel<8, false>), grid, dim3(256), 0, s, *p, nb);
      else if (lpr == 16) hipLaunchKernelGGL((ln_fwd_rows_kernel<16, false>), grid, dim3(256), 0, s, *p, nb);
      else hipLaunchKernelGGL((ln_fwd_rows_kernel<32, false>), grid, dim3(256), 0, s, *p, nb);
    } else if (lpr == 8) hipLaunchKernelGGL(ln_fwd_rows_kernel<8>, grid, dim3(256), 0, s, *p, nb);
    else if (lpr == 16) hipLaunchKernelGGL(ln_fwd_rows_kernel<16>, grid, dim3(256), 0, s, *p, nb);
    else hipLaunchKernelGGL(ln_fwd_rows_kernel<32>, grid, dim3(256), 0, s, *p, nb);
    LVD_LAUNCH_CHECK();
    return 0;
  }
  int blocks = (p->rows + 3) / 4;
  hipLaunchKernelGGL(ln_fwd_kernel, dim3(blocks), dim3(256), 0, s, *p);
  LVD_LAUNCH_CHECK();
  return 0;
}

extern "C" int lvdhip_layernorm_bwd(const lvd_ln_bwd_params* p, void* stream) {
  LVD_CHECK(p && p->x && p->dy && p->dx && p->gamma && p->mean_rstd, "layernorm_bwd: null pointer");
  LVD_CHECK(p->c % 8 == 0 && p->c <= 512 * LN_MAXV, "layernorm_bwd: c=%d unsupported", p->c);
  hipStream_t s = (hipStream_t)stream;
  const int lpr = p->c % 40 == 0 ? p->c / 40 : 0;
  if (lpr == 8 || lpr == 16 || lpr == 32) {
    const int nb = ln_batches(p->rows, lpr);
    const int rows_per_block = 4 * (64 / lpr) * nb;
    const dim3 grid((unsigned)((p->rows + rows_per_block - 1) / rows_per_block));
    if (lpr == 8) hipLaunchKernelGGL(ln_bwd_rows_kernel<8>, grid, dim3(256), 0, s, *p, nb);
    else if (lpr == 16) hipLaunchKernelGGL(ln_bwd_rows_kernel<16>, grid, dim3(256), 0, s, *p, nb);
    else hipLaunchKernelGGL(ln_bwd_rows_kernel<32>, grid, dim3(256), 0, s, *p, nb);
    LVD_LAUNCH_CHECK();
    return 0;
  }
  int blocks = (p->rows + 3) / 4;
  hipLaunchKernelGGL(ln_bwd_kernel, dim3(blocks), dim3(256), 0, s, *p);
  LVD_LAUNCH_CHECK();
  return 0;
}
