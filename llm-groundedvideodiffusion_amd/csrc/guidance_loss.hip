// guidance_loss.hip — fused cross-attention-energy guidance loss, forward + backward, for gfx950.
//
// Reference path being replaced (per guidance key and iteration):
//   AttnProcessor slow path materialises softmax(scale·QK^T) as (frames·heads, HW, 77) fp16, clones it into
//   save_attn_to_dict (models/attention_processor.py:515-586); utils/guidance.py:231-524 then loops over
//   objects × frames × tokens in Python launching mask fills, column gathers, two top-k and reductions each;
//   autograd later walks all of it backwards (models/pipelines.py:120).
// Here: the probability maps are never materialised.  Only the object-token columns are kept (fp32,
// [frames, heads, ntok, P]) and everything is three launches per key:
//   1. ca_probs  : MFMA K·Q^T over the 77 text keys -> row LSE + probabilities of the object tokens
//   2. ca_select : per (frame, head, token) block: exact top-k_fg / top-k_bg by radix select (ties broken by
//                  lowest index), centre-of-mass + velocity terms, loss partial and d(loss)/d(prob)
//   3. ca_dq     : softmax backward over the 77 keys (only the token columns carry gradient) and
//                  dQ = scale · dS · K with K^T staged through LDS — HBM traffic = read Q + write dQ.
// All loss arithmetic is fp32 (the reference's mask is fp32, so A·mask promotes: utils/guidance.py:239,339-353).
#include <algorithm>
#include "common.h"

namespace {

constexpr int TP = 18;
constexpr int MAXTOK = 16;

LVD_DEV void stage_transposed(uint32_t* lds, const lvd_bf16* r0, const lvd_bf16* r1, int vj, int vdc) {
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    int d0 = vdc * 8 + 32 * half;
    uint4 a = ldg16(r0 + d0);
    uint4 b = ldg16(r1 + d0);
    uint32_t aw[4] = {a.x, a.y, a.z, a.w}, bw[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      lds[(d0 + 2 * e) * TP + vj] = (aw[e] & 0xffffu) | (bw[e] << 16);
      lds[(d0 + 2 * e + 1) * TP + vj] = (aw[e] >> 16) | (bw[e] & 0xffff0000u);
    }
  }
}
LVD_DEV bf16x8 frag_transposed(const uint32_t* lds, int d, int ks2, int hi) {
  const uint32_t* r = lds + d * TP + ks2 * 8 + 2 * hi;
  uint2 lo = *reinterpret_cast<const uint2*>(r);
  uint2 h2 = *reinterpret_cast<const uint2*>(r + 4);
  return as_bf16x8(make_uint4(lo.x, lo.y, h2.x, h2.y));
}
LVD_DEV float dot8(uint4 a, uint4 b) {
  return bflo(a.x) * bflo(b.x) + bfhi(a.x) * bfhi(b.x) + bflo(a.y) * bflo(b.y) + bfhi(a.y) * bfhi(b.y) +
         bflo(a.z) * bflo(b.z) + bfhi(a.z) * bfhi(b.z) + bflo(a.w) * bflo(b.w) + bfhi(a.w) * bfhi(b.w);
}

// ------------------------------------------------------------------------------------ 1. probs
// grid (frames * ceil(P/32), heads), one wave per 32-query tile
LVD_DEV void ca_probs_body(const lvd_ca_probs_params& p, const int bx, const int by) {
  const int lane = threadIdx.x, l31 = lane & 31, hi = lane >> 5;
  const int nqt = (p.P + 31) >> 5;
  const int f = bx / nqt, qt = bx - f * nqt, h = by;
  const int qi = qt * 32 + l31;
  const int qic = min(qi, p.P - 1);
  const lvd_bf16* qp = p.q + ((long)f * p.P + qic) * p.ldq + h * 64 + hi * 8;
  uint4 qraw[4];
  bf16x8 qf[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) { qraw[ks] = ldg16(qp + ks * 16); qf[ks] = as_bf16x8(qraw[ks]); }
  const float sc = p.scale * 1.4426950408889634f;
  float m = -1e30f, lsum = 0.f;
  for (int kt = 0; kt * 32 < p.ntext; ++kt) {
    f32x16 st;
#pragma unroll
    for (int e = 0; e < 16; ++e) st[e] = 0.f;
    int kk = min(kt * 32 + l31, p.ntext - 1);
    const lvd_bf16* kp = p.k + (long)kk * p.ldk + h * 64 + hi * 8;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(ldg16(kp + ks * 16)), qf[ks], st, 0, 0, 0);
    float v[16], tmax = -1e30f;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      int kidx = kt * 32 + (e & 3) + 8 * (e >> 2) + 4 * hi;
      v[e] = kidx < p.ntext ? st[e] * sc : -1e30f;
      tmax = fmaxf(tmax, v[e]);
    }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
    float mn = fmaxf(m, tmax), rs = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) rs += fast_exp2(v[e] - mn);
    lsum = lsum * fast_exp2(m - mn) + rs;
    m = mn;
  }
  float ltot = lsum + __shfl_xor(lsum, 32, 64);
  float lse2 = m + log2f(ltot);
  const long row = ((long)f * p.heads + h);
  if (hi == 0 && qi < p.P) p.lse[row * p.P + qi] = lse2 * 0.6931471805599453f;
  for (int t = 0; t < p.ntok; ++t) {
    const lvd_bf16* kp = p.k + (long)p.tok_ids[t] * p.ldk + h * 64 + hi * 8;
    float s = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) s += dot8(qraw[ks], ldg16(kp + ks * 16));
    s += __shfl_xor(s, 32, 64);
    if (hi == 0 && qi < p.P) p.probs[(row * p.ntok + t) * p.P + qi] = fast_exp2(s * sc - lse2);
  }
}

// ------------------------------------------------------------------------------------ 2a. centre of mass
// grid frames*heads*ntok, block 256: com_ws[.,4] = (sum A, com_y, com_x, 0)
LVD_DEV void ca_com_body(const lvd_ca_select_params& p, const int b) {
  __shared__ float red[3][4];
  const float* A = p.probs + (long)b * p.P;
  float s = 0.f, sy = 0.f, sx = 0.f;
  for (int i = threadIdx.x; i < p.P; i += blockDim.x) {
    float a = A[i];
    int y = i / p.W, x = i - y * p.W;
    s += a; sy += a * (float)y; sx += a * (float)x;
  }
  s = wave_sum(s); sy = wave_sum(sy); sx = wave_sum(sx);
  int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { red[0][w] = s; red[1][w] = sy; red[2][w] = sx; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float S = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    float Y = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    float X = red[2][0] + red[2][1] + red[2][2] + red[2][3];
    p.com_ws[(long)b * 4 + 0] = S;
    p.com_ws[(long)b * 4 + 1] = Y / S;
    p.com_ws[(long)b * 4 + 2] = X / S;
    p.com_ws[(long)b * 4 + 3] = 0.f;
  }
}

// ------------------------------------------------------------------------------------ 2b. select
// Exact k-th largest of 44-bit keys (value bits << 12 | (4095 - index)), for the entries inside the box (flag 1, k = k1) and outside
// it (flag 0, k = k0) at once: six 8-bit passes, both histograms built in the same sweep, and the bin that holds the k-th largest found
// by a 256-thread suffix scan (one bin per thread, top bin first) instead of a serial walk over the bins.  thr[w] = key of the k-th
// largest of class w (entries with key >= thr[w] are its top-k set); a class with k <= 0 or no members keeps ~0 ("nothing selected").
// blockDim.x must be 256.
LVD_DEV void radix_select2(const float* vals, const unsigned char* flag, int n, int k0, int k1, bool on0, bool on1,
                           unsigned int (*hist)[256], int (*sh)[2], int (*wtot)[4], unsigned long long thr[2]) {
  unsigned long long prefix[2] = {0, 0}, maskbits = 0;
  int need[2] = {k0, k1};
  const bool on[2] = {on0, on1};
  const int tb = threadIdx.x;
  for (int pass = 5; pass >= 0; --pass) {
    hist[0][tb] = 0;
    hist[1][tb] = 0;
    __syncthreads();
    for (int i = tb; i < n; i += 256) {
      const int w = flag[i];
      if (!on[w]) continue;
      const unsigned long long key = ((unsigned long long)__float_as_uint(vals[i]) << 12) | (unsigned)(4095 - i);
      if ((key & maskbits) == prefix[w]) atomicAdd(&hist[w][(key >> (8 * pass)) & 255], 1u);
    }
    __syncthreads();
    int c[2], incl[2];
#pragma unroll
    for (int w = 0; w < 2; ++w) {
      c[w] = (int)hist[w][255 - tb];
      incl[w] = c[w];
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl[w], o, 64);
        if ((tb & 63) >= o) incl[w] += v;
      }
      if ((tb & 63) == 63) wtot[w][tb >> 6] = incl[w];
    }
    __syncthreads();
#pragma unroll
    for (int w = 0; w < 2; ++w) {
      for (int q = 0; q < (tb >> 6); ++q) incl[w] += wtot[w][q];
      const int excl = incl[w] - c[w];
      // the first bin from the top whose running count reaches `need` (bin 0 takes whatever is left, as the serial walk did)
      if ((incl[w] >= need[w] && excl < need[w]) || (tb == 255 && incl[w] < need[w])) {
        sh[w][0] = 255 - tb;
        sh[w][1] = need[w] - excl;
      }
    }
    __syncthreads();
#pragma unroll
    for (int w = 0; w < 2; ++w) {
      prefix[w] |= (unsigned long long)sh[w][0] << (8 * pass);
      need[w] = sh[w][1];
    }
    maskbits |= 0xffull << (8 * pass);
  }
  thr[0] = on0 ? prefix[0] : ~0ull;
  thr[1] = on1 ? prefix[1] : ~0ull;
}

LVD_DEV void ca_select_body(const lvd_ca_select_params& p, const int b) {
  extern __shared__ unsigned char smem[];
  float* vals = reinterpret_cast<float*>(smem);                         // [P]
  unsigned char* flag = smem + (size_t)p.P * 4;                         // [P] 1 = inside the box
  __shared__ unsigned int hist[2][256];
  __shared__ int sh[2][2];
  __shared__ int wtot[2][4];
  __shared__ float red[8];

  const int t = b % p.ntok;
  const int fh = b / p.ntok;
  const int h = fh % p.heads, f = fh / p.heads;
  const int obj = p.tok_obj[t];
  const int* bx = p.boxes + ((long)obj * p.frames + f) * 6;
  const int x0 = bx[0], y0 = bx[1], x1 = bx[2], y1 = bx[3], kfg = bx[4], kbg = bx[5];
  const int nmask = max(0, x1 - x0) * max(0, y1 - y0);
  const float* A = p.probs + (long)b * p.P;
  float* dA = p.dprobs + (long)b * p.P;
  const float wt = p.tok_weight[t];
  const float c = p.grad_scale * wt;

  for (int i = threadIdx.x; i < p.P; i += blockDim.x) {
    int y = i / p.W, x = i - y * p.W;
    vals[i] = A[i];
    flag[i] = (y >= y0 && y < y1 && x >= x0 && x < x1) ? 1 : 0;
  }
  __syncthreads();

  unsigned long long thr_fg = ~0ull, thr_bg = ~0ull;  // "nothing selected"
  const bool ratio = p.use_ratio_loss != 0;
  if (!ratio) {
    unsigned long long thr[2];
    radix_select2(vals, flag, p.P, kbg, kfg, p.P - nmask > 0, nmask > 0, hist, sh, wtot, thr);
    thr_bg = thr[0];
    thr_fg = thr[1];
  }
  // ratio-based energy (utils/guidance.py:312-323): act = sum(A*mask) / (sum(A) + eps); loss = mean over heads of (1 - act)^2
  float r_in = 0.f, r_out = 0.f, ratio_loss = 0.f;  // dL/dA[p] = r_in inside the box, r_out outside
  if (ratio) {
    float sa = 0.f, sm = 0.f;
    for (int i = threadIdx.x; i < p.P; i += blockDim.x) { sa += vals[i]; sm += flag[i] ? vals[i] : 0.f; }
    sa = wave_sum(sa); sm = wave_sum(sm);
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = sa; red[4 + (threadIdx.x >> 6)] = sm; }
    __syncthreads();
    sa = red[0] + red[1] + red[2] + red[3]; sm = red[4] + red[5] + red[6] + red[7];
    __syncthreads();
    const float den = sa + p.ratio_eps, act = sm / den;
    ratio_loss = (1.f - act) * (1.f - act) / (float)p.heads;
    const float k2 = -2.f * (1.f - act) / ((float)p.heads * den * den);
    r_in = k2 * (den - sm);
    r_out = k2 * (-sm);
  }
  // attention sync (:401-430).  The reference crops BOTH frames with the box of the NEXT frame (its x_min..y_max were
  // overwritten by the t1 loop), per head: w * mean_box((A_f - A_f+1)^2).  This block owns frame f: the pair (f, f+1) with
  // box(f+1), and — as the second frame of the pair (f-1, f) — the gradient that pair sends to frame f, cropped with box(f).
  // An empty next-frame box is a NaN in the reference (mean of an empty crop); here the pair contributes nothing.
  const float sw = p.attn_sync_weight;
  const float* Anext = nullptr;
  const float* Aprev = nullptr;
  int nx0 = 0, ny0 = 0, nx1 = 0, ny1 = 0;
  float snext = 0.f, sprev = 0.f;
  if (sw != 0.f) {
    const long fstride = (long)p.heads * p.ntok * p.P;
    if (f + 1 < p.frames) {
      const int* b1 = p.boxes + ((long)obj * p.frames + f + 1) * 6;
      nx0 = b1[0]; ny0 = b1[1]; nx1 = b1[2]; ny1 = b1[3];
      const int n1 = max(0, nx1 - nx0) * max(0, ny1 - ny0);
      if (n1 > 0) { Anext = A + fstride; snext = sw / (float)n1; }
    }
    if (f >= 1 && nmask > 0) { Aprev = A - fstride; sprev = sw / (float)nmask; }
  }

  // centre-of-mass gradient coefficients: dL/dA[p] = gy*(y - cy)/S + gx*(x - cx)/S  (+ same with the t1 roles)
  float gy = 0.f, gx = 0.f, cy = 0.f, cx = 0.f, S = 1.f, com_loss = 0.f;
  if (p.com_loss_scale > 0.f && nmask > 0) {
    const float* cw = p.com_ws + (long)b * 4;
    S = cw[0]; cy = cw[1]; cx = cw[2];
    float my = 0.5f * (float)(y0 + y1 - 1), mx = 0.5f * (float)(x0 + x1 - 1);
    float ey = cy - my, ex = cx - mx;
    const float kh = p.com_loss_scale / (float)p.heads;
    com_loss += kh * (ey * ey + ex * ex);
    gy += 2.f * kh * ey; gx += 2.f * kh * ex;
    // velocity term with the next frame (utils/guidance.py:489-522); f1 = min(f+1, F-1)
    if (f + 1 < p.frames) {
      const int* b1 = p.boxes + ((long)obj * p.frames + f + 1) * 6;
      int n1 = max(0, b1[2] - b1[0]) * max(0, b1[3] - b1[1]);
      if (n1 > 0) {
        const float* c1 = p.com_ws + ((long)b + (long)p.heads * p.ntok) * 4;
        float my1 = 0.5f * (float)(b1[1] + b1[3] - 1), mx1 = 0.5f * (float)(b1[0] + b1[2] - 1);
        float vy = (c1[1] - cy) - (my1 - my), vx = (c1[2] - cx) - (mx1 - mx);
        com_loss += kh * (vy * vy + vx * vx);
        gy -= 2.f * kh * vy; gx -= 2.f * kh * vx;
      }
    }
    // this frame acting as "t1" of the previous frame's velocity term
    if (f >= 1) {
      const int* b0 = p.boxes + ((long)obj * p.frames + f - 1) * 6;
      int n0 = max(0, b0[2] - b0[0]) * max(0, b0[3] - b0[1]);
      if (n0 > 0) {
        const float* c0 = p.com_ws + ((long)b - (long)p.heads * p.ntok) * 4;
        float my0 = 0.5f * (float)(b0[1] + b0[3] - 1), mx0 = 0.5f * (float)(b0[0] + b0[2] - 1);
        float vy = (cy - c0[1]) - (my - my0), vx = (cx - c0[2]) - (mx - mx0);
        gy += 2.f * kh * vy; gx += 2.f * kh * vx;
      }
    }
  }

  float sfg = 0.f, sbg = 0.f, ssync = 0.f;
  const float gfg = -p.fg_weight / (float)kfg, gbg = p.bg_weight / (float)kbg;
  for (int i = threadIdx.x; i < p.P; i += blockDim.x) {
    float a = vals[i];
    unsigned long long key = ((unsigned long long)__float_as_uint(a) << 12) | (unsigned)(4095 - i);
    float g = 0.f;
    if (ratio) {
      g += flag[i] ? r_in : r_out;
    } else if (flag[i]) {
      if (key >= thr_fg) { sfg += a; g += gfg; }
    } else {
      if (key >= thr_bg) { sbg += a; g += gbg; }
    }
    const int y = i / p.W, x = i - y * p.W;
    if (gy != 0.f || gx != 0.f) g += (gy * ((float)y - cy) + gx * ((float)x - cx)) / S;
    if (Anext && y >= ny0 && y < ny1 && x >= nx0 && x < nx1) {
      const float d = a - Anext[i];
      ssync += d * d;
      g += 2.f * snext * d;
    }
    if (Aprev && flag[i]) g -= 2.f * sprev * (Aprev[i] - a);
    dA[i] = c * g;
  }
  sfg = wave_sum(sfg); sbg = wave_sum(sbg); ssync = wave_sum(ssync);
  int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { red[w] = sfg; red[4 + w] = sbg; }
  __syncthreads();
  const float tf = red[0] + red[1] + red[2] + red[3], tb = red[4] + red[5] + red[6] + red[7];
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = ssync;
  __syncthreads();
  const float tsync = red[0] + red[1] + red[2] + red[3];
  float loss = ratio ? ratio_loss : p.fg_weight * (1.f - tf / (float)kfg) + p.bg_weight * (tb / (float)kbg);
  loss += com_loss + snext * tsync;

  // BoxDiff corner constraint (:240-287, 433-465): |max over rows/columns of A - max of the mask| on the corner columns/rows;
  // the gradient goes to the (first) arg-max of each column / row.  dA was written above: the adds below are ordered by barriers.
  if (p.boxdiff_loss_scale > 0.f) {
    const int L = p.boxdiff_L, Hh = p.H, Ww = p.W;
    float cc = 0.f;
    __syncthreads();
    for (int pass = 0; pass < 2; ++pass) {  // pass 0: columns (max over y), pass 1: rows (max over x)
      const int n = pass == 0 ? Ww : Hh, m = pass == 0 ? Hh : Ww;
      const int lo = pass == 0 ? x0 : y0, hi2 = pass == 0 ? x1 : y1;
      const float norm = p.boxdiff_normed ? 1.f / ((float)p.heads * (float)n) : 1.f;
      for (int j = threadIdx.x; j < n; j += blockDim.x) {
        const bool corner = (j >= max(lo - L, 0) && j < min(lo + L + 1, n)) || (j >= max(hi2 - L, 0) && j < min(hi2 + L + 1, n));
        if (!corner) continue;
        float best = -1.f;
        int arg = 0;
        for (int q = 0; q < m; ++q) {
          const int idx = pass == 0 ? q * Ww + j : j * Ww + q;
          if (vals[idx] > best) { best = vals[idx]; arg = idx; }
        }
        const float target = (nmask > 0 && j >= lo && j < hi2) ? 1.f : 0.f;
        const float diff = best - target;
        cc += fabsf(diff) * norm;
        const float sg = diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f);
        dA[arg] += c * p.boxdiff_loss_scale * norm * sg;
      }
      __syncthreads();
    }
    cc = wave_sum(cc);
    if ((threadIdx.x & 63) == 0) red[w] = cc;
    __syncthreads();
    loss += p.boxdiff_loss_scale * (red[0] + red[1] + red[2] + red[3]);
  }
  if (threadIdx.x == 0) p.loss_partial[b] = wt * loss;
}

// ------------------------------------------------------------------------------------ 3. dQ
// NT = compile-time bound on the object tokens of the launch (2 / 4 / 8 / 16): the per-score test "is this key one of the object tokens"
// costs 16 x NT selects per key tile, and with the bound fixed at 16 it was the kernel (768 VALU operations per tile against 8 MFMAs).
template <int NT>
LVD_DEV void ca_dq_body(const lvd_ca_dq_params& p, const int bx, const int by) {
  __shared__ uint32_t kt_lds[64 * TP];
  const int lane = threadIdx.x, l31 = lane & 31, hi = lane >> 5;
  const int nqt = (p.P + 31) >> 5;
  const int f = bx / nqt, qt = bx - f * nqt, h = by;
  const int qi = qt * 32 + l31;
  const int qic = min(qi, p.P - 1);
  const lvd_bf16* qp = p.q + ((long)f * p.P + qic) * p.ldq + h * 64 + hi * 8;
  bf16x8 qf[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) qf[ks] = as_bf16x8(ldg16(qp + ks * 16));
  const long row = ((long)f * p.heads + h);
  const float lse2 = p.lse[row * p.P + qic] * 1.4426950408889634f;
  const float sc = p.scale * 1.4426950408889634f;
  // per-query token-column gradients and c = sum_t A_t dA_t; every load is issued unconditionally (clamped token index, masked
  // value): a load inside a branch is waited for inside that branch, one memory round trip per token
  float da[NT], pa[NT];
  int tk[NT];
  float cq = 0.f;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int tt = min(t, p.ntok - 1);
    const long idx = (row * p.ntok + tt) * p.P + qic;
    da[t] = p.dprobs[idx];
    pa[t] = p.probs[idx];
    tk[t] = p.tok_ids[tt];
  }
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    if (t >= p.ntok) { da[t] = 0.f; tk[t] = -1; }
    cq += pa[t] * da[t];
  }
  f32x16 dq0, dq1;
#pragma unroll
  for (int e = 0; e < 16; ++e) { dq0[e] = 0.f; dq1[e] = 0.f; }
  const int vj = lane & 15, vdc = lane >> 4;
  for (int kt = 0; kt * 32 < p.ntext; ++kt) {
    f32x16 st;
#pragma unroll
    for (int e = 0; e < 16; ++e) st[e] = 0.f;
    {
      int kk = min(kt * 32 + l31, p.ntext - 1);
      const lvd_bf16* kp = p.k + (long)kk * p.ldk + h * 64 + hi * 8;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(ldg16(kp + ks * 16)), qf[ks], st, 0, 0, 0);
    }
    {
      int k0 = min(kt * 32 + 2 * vj, p.ntext - 1), k1 = min(kt * 32 + 2 * vj + 1, p.ntext - 1);
      stage_transposed(kt_lds, p.k + (long)k0 * p.ldk + h * 64, p.k + (long)k1 * p.ldk + h * 64, vj, vdc);
    }
    __syncthreads();
    float ds[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      int kidx = kt * 32 + (e & 3) + 8 * (e >> 2) + 4 * hi;
      float pr = kidx < p.ntext ? fast_exp2(st[e] * sc - lse2) : 0.f;
      float g = -cq;
#pragma unroll
      for (int t = 0; t < NT; ++t) g += (tk[t] == kidx) ? da[t] : 0.f;
      ds[e] = pr * g;
    }
#pragma unroll
    for (int ks2 = 0; ks2 < 2; ++ks2) {
      uint4 w;
      w.x = pack2bf(ds[ks2 * 8 + 0], ds[ks2 * 8 + 1]); w.y = pack2bf(ds[ks2 * 8 + 2], ds[ks2 * 8 + 3]);
      w.z = pack2bf(ds[ks2 * 8 + 4], ds[ks2 * 8 + 5]); w.w = pack2bf(ds[ks2 * 8 + 6], ds[ks2 * 8 + 7]);
      bf16x8 dsf = as_bf16x8(w);
      dq0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_transposed(kt_lds, l31, ks2, hi), dsf, dq0, 0, 0, 0);
      dq1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_transposed(kt_lds, 32 + l31, ks2, hi), dsf, dq1, 0, 0, 0);
    }
    __syncthreads();
  }
  if (qi < p.P) {
    lvd_bf16* op = p.dq + ((long)f * p.P + qi) * p.lddq + h * 64 + 4 * hi;
    float* ap = p.acc32 + ((long)f * p.P + qi) * p.ldacc + h * 64 + 4 * hi;  // token chunks are summed in fp32 (acc_mode, lvdhip.h)
    const float fs = p.scale;
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
      f32x4 v0, v1;
#pragma unroll
      for (int e = 0; e < 4; ++e) { v0[e] = dq0[rq * 4 + e] * fs; v1[e] = dq1[rq * 4 + e] * fs; }
      if (p.acc_mode >= 2) {
        v0 += *reinterpret_cast<const f32x4*>(ap + 8 * rq);
        v1 += *reinterpret_cast<const f32x4*>(ap + 32 + 8 * rq);
      }
      if (p.acc_mode == 1 || p.acc_mode == 2) {
        *reinterpret_cast<f32x4*>(ap + 8 * rq) = v0;
        *reinterpret_cast<f32x4*>(ap + 32 + 8 * rq) = v1;
      } else {
        uint2 w0, w1;
        w0.x = pack2bf(v0[0], v0[1]); w0.y = pack2bf(v0[2], v0[3]);
        w1.x = pack2bf(v1[0], v1[1]); w1.y = pack2bf(v1[2], v1[3]);
        stg8(op + 8 * rq, w0);
        stg8(op + 32 + 8 * rq, w1);
      }
    }
  }
}

// ------------------------------------------------------------------------------------ kernels: one key, or all keys of an iteration
// The six guidance keys of an iteration are independent and small (180-720 query positions x 10-20 heads): launched one by one they are
// 18 launches of latency-bound one-wave workgroups.  The *_multi kernels take the per-key parameter blocks as ONE kernel argument and
// blockIdx.z picks the key; a block outside its key's own grid leaves at once (the grid is the largest key's).
template <class P>
struct KeyTable {
  P k[LVD_CA_MAX_KEYS];
};
LVD_DEV int probs_gx(const lvd_ca_probs_params& p) { return ((p.P + 31) >> 5) * p.frames; }
LVD_DEV int dq_gx(const lvd_ca_dq_params& p) { return ((p.P + 31) >> 5) * p.frames; }

__global__ __launch_bounds__(64) void ca_probs_kernel(const lvd_ca_probs_params p) { ca_probs_body(p, blockIdx.x, blockIdx.y); }
__global__ __launch_bounds__(64) void ca_probs_multi_kernel(const KeyTable<lvd_ca_probs_params> tab) {
  const lvd_ca_probs_params& p = tab.k[blockIdx.z];
  if ((int)blockIdx.x >= probs_gx(p) || (int)blockIdx.y >= p.heads) return;
  ca_probs_body(p, blockIdx.x, blockIdx.y);
}
__global__ void ca_com_kernel(const lvd_ca_select_params p) { ca_com_body(p, blockIdx.x); }
__global__ void ca_com_multi_kernel(const KeyTable<lvd_ca_select_params> tab) {
  const lvd_ca_select_params& p = tab.k[blockIdx.y];
  if ((int)blockIdx.x >= p.frames * p.heads * p.ntok) return;
  ca_com_body(p, blockIdx.x);
}
__global__ __launch_bounds__(256) void ca_select_kernel(const lvd_ca_select_params p) { ca_select_body(p, blockIdx.x); }
__global__ __launch_bounds__(256) void ca_select_multi_kernel(const KeyTable<lvd_ca_select_params> tab) {
  const lvd_ca_select_params& p = tab.k[blockIdx.y];
  if ((int)blockIdx.x >= p.frames * p.heads * p.ntok) return;
  ca_select_body(p, blockIdx.x);
}
template <int NT>
__global__ __launch_bounds__(64) void ca_dq_kernel(const lvd_ca_dq_params p) { ca_dq_body<NT>(p, blockIdx.x, blockIdx.y); }
template <int NT>
__global__ __launch_bounds__(64) void ca_dq_multi_kernel(const KeyTable<lvd_ca_dq_params> tab) {
  const lvd_ca_dq_params& p = tab.k[blockIdx.z];
  if ((int)blockIdx.x >= dq_gx(p) || (int)blockIdx.y >= p.heads) return;
  ca_dq_body<NT>(p, blockIdx.x, blockIdx.y);
}

}  // namespace

namespace {
int check_probs(const lvd_ca_probs_params* p) {
  LVD_CHECK(p && p->q && p->k && p->tok_ids && p->probs && p->lse, "ca_probs: null pointer");
  LVD_CHECK(p->ntok > 0 && p->ntok <= MAXTOK, "ca_probs: ntok=%d outside 1..%d", p->ntok, MAXTOK);
  LVD_CHECK(p->ldq % 8 == 0 && p->ldk % 8 == 0, "ca_probs: leading dims");
  return 0;
}
int check_select(const lvd_ca_select_params* p) {
  LVD_CHECK(p && p->probs && p->dprobs && p->tok_obj && p->boxes && p->tok_weight && p->loss_partial && p->com_ws, "ca_select: null pointer");
  LVD_CHECK(p->P == p->H * p->W && p->P <= 4096, "ca_select: P=%d must equal H*W and be <= 4096", p->P);
  return 0;
}
int check_dq(const lvd_ca_dq_params* p) {
  LVD_CHECK(p && p->q && p->k && p->tok_ids && p->probs && p->dprobs && p->lse && p->dq, "ca_dq: null pointer");
  LVD_CHECK(p->ntok > 0 && p->ntok <= MAXTOK, "ca_dq: ntok=%d outside 1..%d", p->ntok, MAXTOK);
  LVD_CHECK(p->acc_mode >= 0 && p->acc_mode <= 3 && (p->acc_mode == 0 || (p->acc32 && p->ldacc % 4 == 0)), "ca_dq: acc_mode %d needs an fp32 accumulator", p->acc_mode);
  return 0;
}
}  // namespace

// All keys of a guidance iteration in one launch each (3 launches instead of 3 per key; 4 with the centre-of-mass term).  `keys` is a
// host array of `nkeys` (<= LVD_CA_MAX_KEYS) parameter blocks, each exactly what the single-key entry point takes.
extern "C" int lvdhip_ca_probs_multi(const lvd_ca_probs_params* keys, int32_t nkeys, void* stream) {
  LVD_CHECK(keys && nkeys >= 1 && nkeys <= LVD_CA_MAX_KEYS, "ca_probs_multi: 1..%d keys", LVD_CA_MAX_KEYS);
  KeyTable<lvd_ca_probs_params> tab;
  int gx = 0, gy = 0;
  for (int i = 0; i < nkeys; ++i) {
    if (int rc = check_probs(keys + i)) return rc;
    tab.k[i] = keys[i];
    gx = std::max(gx, ((keys[i].P + 31) / 32) * keys[i].frames);
    gy = std::max(gy, keys[i].heads);
  }
  hipLaunchKernelGGL(ca_probs_multi_kernel, dim3(gx, gy, nkeys), dim3(64), 0, (hipStream_t)stream, tab);
  LVD_LAUNCH_CHECK();
  return 0;
}

extern "C" int lvdhip_ca_select_multi(const lvd_ca_select_params* keys, int32_t nkeys, void* stream) {
  LVD_CHECK(keys && nkeys >= 1 && nkeys <= LVD_CA_MAX_KEYS, "ca_select_multi: 1..%d keys", LVD_CA_MAX_KEYS);
  KeyTable<lvd_ca_select_params> tab;
  int blocks = 0;
  size_t smem = 0;
  bool com = false;
  for (int i = 0; i < nkeys; ++i) {
    if (int rc = check_select(keys + i)) return rc;
    tab.k[i] = keys[i];
    blocks = std::max(blocks, keys[i].frames * keys[i].heads * keys[i].ntok);
    smem = std::max(smem, (size_t)keys[i].P * 5 + 16);
    com = com || keys[i].com_loss_scale > 0.f;
    LVD_CHECK((keys[i].com_loss_scale > 0.f) == (keys[0].com_loss_scale > 0.f), "ca_select_multi: the centre-of-mass term is on for all keys or for none");
  }
  hipStream_t s = (hipStream_t)stream;
  if (com) {
    hipLaunchKernelGGL(ca_com_multi_kernel, dim3(blocks, nkeys), dim3(256), 0, s, tab);
    LVD_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(ca_select_multi_kernel, dim3(blocks, nkeys), dim3(256), smem, s, tab);
  LVD_LAUNCH_CHECK();
  return 0;
}

extern "C" int lvdhip_ca_dq_multi(const lvd_ca_dq_params* keys, int32_t nkeys, void* stream) {
  LVD_CHECK(keys && nkeys >= 1 && nkeys <= LVD_CA_MAX_KEYS, "ca_dq_multi: 1..%d keys", LVD_CA_MAX_KEYS);
  KeyTable<lvd_ca_dq_params> tab;
  int gx = 0, gy = 0, nt = 0;
  for (int i = 0; i < nkeys; ++i) {
    if (int rc = check_dq(keys + i)) return rc;
    tab.k[i] = keys[i];
    gx = std::max(gx, ((keys[i].P + 31) / 32) * keys[i].frames);
    gy = std::max(gy, keys[i].heads);
    nt = std::max(nt, keys[i].ntok);
  }
  const dim3 grid(gx, gy, nkeys);
  hipStream_t s = (hipStream_t)stream;
  if (nt <= 2) hipLaunchKernelGGL(ca_dq_multi_kernel<2>, grid, dim3(64), 0, s, tab);
  else if (nt <= 4) hipLaunchKernelGGL(ca_dq_multi_kernel<4>, grid, dim3(64), 0, s, tab);
  else if (nt <= 8) hipLaunchKernelGGL(ca_dq_multi_kernel<8>, grid, dim3(64), 0, s, tab);
  else hipLaunchKernelGGL(ca_dq_multi_kernel<MAXTOK>, grid, dim3(64), 0, s, tab);
  LVD_LAUNCH_CHECK();
  return 0;
}

extern "C" int lvdhip_ca_probs(const lvd_ca_probs_params* p, void* stream) {
  if (int rc = check_probs(p)) return rc;
  dim3 grid(((p->P + 31) / 32) * p->frames, p->heads);
  hipLaunchKernelGGL(ca_probs_kernel, grid, dim3(64), 0, (hipStream_t)stream, *p);
  LVD_LAUNCH_CHECK();
  return 0;
}

extern "C" int lvdhip_ca_select(const lvd_ca_select_params* p, void* stream) {
  if (int rc = check_select(p)) return rc;
  hipStream_t s = (hipStream_t)stream;
  int blocks = p->frames * p->heads * p->ntok;
  if (p->com_loss_scale > 0.f) {
    hipLaunchKernelGGL(ca_com_kernel, dim3(blocks), dim3(256), 0, s, *p);
    LVD_LAUNCH_CHECK();
  }
  size_t smem = (size_t)p->P * 5 + 16;
  hipLaunchKernelGGL(ca_select_kernel, dim3(blocks), dim3(256), smem, s, *p);
  LVD_LAUNCH_CHECK();
  return 0;
}

extern "C" int lvdhip_ca_dq(const lvd_ca_dq_params* p, void* stream) {
  if (int rc = check_dq(p)) return rc;
  dim3 grid(((p->P + 31) / 32) * p->frames, p->heads);
  if (p->ntok <= 2) hipLaunchKernelGGL(ca_dq_kernel<2>, grid, dim3(64), 0, (hipStream_t)stream, *p);
  else if (p->ntok <= 4) hipLaunchKernelGGL(ca_dq_kernel<4>, grid, dim3(64), 0, (hipStream_t)stream, *p);
  else if (p->ntok <= 8) hipLaunchKernelGGL(ca_dq_kernel<8>, grid, dim3(64), 0, (hipStream_t)stream, *p);
  else hipLaunchKernelGGL(ca_dq_kernel<MAXTOK>, grid, dim3(64), 0, (hipStream_t)stream, *p);
  LVD_LAUNCH_CHECK();
  return 0;
}
