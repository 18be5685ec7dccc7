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

// The head's keys, row-major, staged once per 256-thread workgroup: 96 rows (positions past the prompt repeat its last row; their scores
// are masked) of 64 channels, 144-byte row pitch (16-byte fragment reads of 16 consecutive rows hit 16 different bank groups).  Loaded
// this way a key row is one cache line shared by eight lanes; as MFMA fragments straight from global memory every lane of a load touches
// its own line, and each wave of the workgroup repeats the twelve loads.
constexpr int KROW = 36;  // dwords
struct KStage { uint4 v[3]; };
LVD_DEV KStage kstage_load(const lvd_bf16* k, int ldk, int ntext, int h) {
  KStage r;
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    const int row = min(s * 32 + ((int)threadIdx.x >> 3), ntext - 1);
    r.v[s] = ldg16(k + (long)row * ldk + h * 64 + (threadIdx.x & 7) * 8);
  }
  return r;
}
LVD_DEV void kstage_store(uint32_t* k_lds, const KStage& r) {
#pragma unroll
  for (int s = 0; s < 3; ++s) *reinterpret_cast<uint4*>(k_lds + (s * 32 + ((int)threadIdx.x >> 3)) * KROW + (threadIdx.x & 7) * 4) = r.v[s];
}
LVD_DEV bf16x8 kstage_frag(const uint32_t* k_lds, int kt, int ks, int l31, int hi) {
  return as_bf16x8(*reinterpret_cast<const uint4*>(k_lds + (kt * 32 + l31) * KROW + ks * 8 + hi * 4));
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

// Prompts of up to 96 text positions (three 32-key tiles; CLIP has 77).  A wave here is bound by latency and by the load path, not by
// HBM: the general body re-reads the head's keys tile by tile and then once more, row by row, for the object tokens (28 KB of L2 reads
// for 4 KB of queries).  Here every global load of the wave — the query rows and the twelve key fragments — is issued before the first
// MFMA (one memory round trip), and the object tokens' scores are taken from the S tiles the MFMAs already produced instead of being
// recomputed from re-loaded key rows.  Same result as ca_probs_body up to summation order (the token scores are the MFMA's fp32 sums).
LVD_DEV void ca_probs_body3(const lvd_ca_probs_params& p, const int bx, const int by) {
  __shared__ uint32_t k_lds[96 * KROW];
  const int lane = threadIdx.x & 63, l31 = lane & 31, hi = lane >> 5;
  const int nqt = (p.P + 31) >> 5;
  const bool live = bx < nqt * p.frames;
  const int bxc = live ? bx : nqt * p.frames - 1;
  const int f = bxc / nqt, qt = bxc - f * nqt, h = by;
  const int qi = qt * 32 + l31;
  const int qic = min(qi, p.P - 1);
  const lvd_bf16* qp = p.q + ((long)f * p.P + qic) * p.ldq + h * 64 + hi * 8;
  uint4 qraw[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) qraw[ks] = ldg16(qp + ks * 16);
  kstage_store(k_lds, kstage_load(p.k, p.ldk, p.ntext, h));
  __syncthreads();
  if (!live) return;
  const float sc = p.scale * 1.4426950408889634f;
  float v[3][16], m = -1e30f;
#pragma unroll
  for (int kt = 0; kt < 3; ++kt) {
    f32x16 st;
#pragma unroll
    for (int e = 0; e < 16; ++e) st[e] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kstage_frag(k_lds, kt, ks, l31, hi), as_bf16x8(qraw[ks]), st, 0, 0, 0);
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int kidx = kt * 32 + (e & 3) + 8 * (e >> 2) + 4 * hi;
      v[kt][e] = kidx < p.ntext ? st[e] * sc : -1e30f;
      m = fmaxf(m, v[kt][e]);
    }
  }
  m = fmaxf(m, __shfl_xor(m, 32, 64));
  float lsum = 0.f;
#pragma unroll
  for (int kt = 0; kt < 3; ++kt)
#pragma unroll
    for (int e = 0; e < 16; ++e) lsum += fast_exp2(v[kt][e] - m);
  lsum += __shfl_xor(lsum, 32, 64);
  const float lse2 = m + log2f(lsum);
  const long row = ((long)f * p.heads + h);
  if (hi == 0 && qi < p.P) p.lse[row * p.P + qi] = lse2 * 0.6931471805599453f;
  for (int t = 0; t < p.ntok; ++t) {
    // score of text position tk for this lane's query: tile tk / 32, accumulator slot e with (e & 3) + 8 (e >> 2) + 4 hi = tk % 32,
    // held by the half-wave hi = bit 2 of tk
    const int tk = p.tok_ids[t], kt = tk >> 5, r = tk & 31;
    float s = -1e30f;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const float cand = kt == 0 ? v[0][e] : (kt == 1 ? v[1][e] : v[2][e]);
      s = ((e & 3) + 8 * (e >> 2) + 4 * hi == r) ? cand : s;
    }
    s = fmaxf(s, __shfl_xor(s, 32, 64));  // the other half-wave holds -1e30
    if (hi == 0 && qi < p.P) p.probs[(row * p.ntok + t) * p.P + qi] = fast_exp2(s - lse2);
  }
}

// ------------------------------------------------------------------------------------ 1b. full maps (AttnProcessor's saved-probabilities branch)
// The reference's slow path (models/attention_processor.py:515-552 -> Attention.get_attention_scores :222-258) materialises
// softmax(scale * Q K^T) over ALL text positions and stores it as (batch, heads, HW, tokens).  The guidance loss above never needs that
// tensor; visualisation, `return_attntion_probs` and `attn_process_fn` callers do.  Same staging and MFMA tiles as ca_probs_body3 (prompts of
// up to 96 positions), one 32-query tile per wave; a lane ends up with its query's scores of keys kt*32 + 8 j + 4 hi + (0..3), normalises
// them in registers and writes them as runs of four.  out[((s * heads + h) * P + q) * ntext + key], fp32.
__global__ __launch_bounds__(256) void ca_probs_full_kernel(const lvd_ca_probs_full_params p) {
  __shared__ uint32_t k_lds[96 * KROW];
  const int lane = threadIdx.x & 63, l31 = lane & 31, hi = lane >> 5;
  const int nqt = (p.P + 31) >> 5;
  const int h = blockIdx.x % p.heads;
  const int tile = (blockIdx.x / p.heads) * 4 + (threadIdx.x >> 6);
  // the four tiles of a workgroup must share their keys: workgroups never straddle samples (tiles of a sample are padded to a multiple of 4)
  const int tiles_per_sample = (nqt + 3) & ~3;
  const int s = tile / tiles_per_sample, qt = tile - s * tiles_per_sample;
  const bool live = qt < nqt;
  const int qi = qt * 32 + l31;
  const int qic = min(qi, p.P - 1);
  const lvd_bf16* qp = p.q + ((long)s * p.P + qic) * p.ldq + h * 64 + hi * 8;
  uint4 qraw[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) qraw[ks] = ldg16(qp + ks * 16);
  const lvd_bf16* kb = p.k + (long)(s / p.samples_per_key) * p.ntext * p.ldk;
  kstage_store(k_lds, kstage_load(kb, p.ldk, p.ntext, h));
  __syncthreads();
  if (!live) return;
  const float sc = p.scale * 1.4426950408889634f;
  const float* kbias = p.key_bias ? p.key_bias + (long)s * p.ld_key_bias : nullptr;  // wave-uniform
  float v[3][16], m = -1e30f;
#pragma unroll
  for (int kt = 0; kt < 3; ++kt) {
    f32x16 st;
#pragma unroll
    for (int e = 0; e < 16; ++e) st[e] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kstage_frag(k_lds, kt, ks, l31, hi), as_bf16x8(qraw[ks]), st, 0, 0, 0);
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int kidx = kt * 32 + (e & 3) + 8 * (e >> 2) + 4 * hi;
      float sv = st[e] * sc;
      if (kbias) sv += kbias[min(kidx, p.ntext - 1)] * 1.4426950408889634f;
      v[kt][e] = kidx < p.ntext ? sv : -1e30f;
      m = fmaxf(m, v[kt][e]);
    }
  }
  m = fmaxf(m, __shfl_xor(m, 32, 64));
  float lsum = 0.f;
#pragma unroll
  for (int kt = 0; kt < 3; ++kt)
#pragma unroll
    for (int e = 0; e < 16; ++e) { v[kt][e] = fast_exp2(v[kt][e] - m); lsum += v[kt][e]; }
  lsum += __shfl_xor(lsum, 32, 64);
  const float inv = 1.f / lsum;
  if (qi >= p.P) return;
  float* o = p.probs + (((long)s * p.heads + h) * p.P + qi) * p.ntext;
#pragma unroll
  for (int kt = 0; kt < 3; ++kt)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int kidx = kt * 32 + (e & 3) + 8 * (e >> 2) + 4 * hi;
      if (kidx < p.ntext) o[kidx] = v[kt][e] * inv;
    }
}

// ------------------------------------------------------------------------------------ 1c. (processed) probabilities x V
// The other half of the reference's slow path: hidden = bmm(attention_probs, value) (models/attention_processor.py:549), where the
// probabilities may have been rewritten by the caller's `attn_process_fn` (:537-548) — so they come from memory, fp32, not from the
// fused attention kernel.  64 query rows per workgroup: the rows' probabilities and the head's V rows are staged in LDS (coalesced
// loads), each thread owns (row, 16 channels).  HBM-bound by the fp32 map (ntext * 4 bytes per (row, head) against 128 bytes of output).
__global__ __launch_bounds__(256) void ca_apply_probs_kernel(const lvd_ca_apply_probs_params p) {
  extern __shared__ float apply_lds[];
  float* pl = apply_lds;                    // [64][ntext] probabilities of the tile, row pitch ntext | 1 (odd: conflict-free columns)
  const int pitch = p.ntext | 1;
  float* vl = apply_lds + 64 * pitch;       // [ntext][64] V of this (sample, head)
  const int sh = blockIdx.y, s = sh / p.heads, h = sh - s * p.heads;
  const int row0 = blockIdx.x * 64, rows = min(64, p.P - row0);
  const float* src = p.probs + ((long)sh * p.P + row0) * p.ntext;
  for (int i = threadIdx.x; i < rows * p.ntext; i += 256) {
    const int r = i / p.ntext, t = i - r * p.ntext;
    pl[r * pitch + t] = src[i];
  }
  const lvd_bf16* vb = p.v + (long)(s / p.samples_per_key) * p.ntext * p.ldv + h * 64;
  for (int i = threadIdx.x; i < p.ntext * 8; i += 256) {
    const int t = i >> 3, c = (i & 7) * 8;
    const uint4 raw = ldg16(vb + (long)t * p.ldv + c);
    float* d = vl + t * 64 + c;
    d[0] = bflo(raw.x); d[1] = bfhi(raw.x); d[2] = bflo(raw.y); d[3] = bfhi(raw.y);
    d[4] = bflo(raw.z); d[5] = bfhi(raw.z); d[6] = bflo(raw.w); d[7] = bfhi(raw.w);
  }
  __syncthreads();
  const int r = threadIdx.x >> 2, c0 = (threadIdx.x & 3) * 16;
  if (r >= rows) return;
  float acc[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  for (int t = 0; t < p.ntext; ++t) {
    const float a = pl[r * pitch + t];
    const float4* vv = reinterpret_cast<const float4*>(vl + t * 64 + c0);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float4 x = vv[e];
      acc[4 * e] += a * x.x; acc[4 * e + 1] += a * x.y; acc[4 * e + 2] += a * x.z; acc[4 * e + 3] += a * x.w;
    }
  }
  lvd_bf16* o = p.out + ((long)s * p.P + row0 + r) * p.ldo + h * 64 + c0;
  stg16(o, make_uint4(pack2bf(acc[0], acc[1]), pack2bf(acc[2], acc[3]), pack2bf(acc[4], acc[5]), pack2bf(acc[6], acc[7])));
  stg16(o + 8, make_uint4(pack2bf(acc[8], acc[9]), pack2bf(acc[10], acc[11]), pack2bf(acc[12], acc[13]), pack2bf(acc[14], acc[15])));
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
// One WAVE per (frame, head, token) map, four maps per workgroup, no workgroup barrier anywhere: the map (P <= 64 E values) lives in the
// lanes' registers, entry i in lane i % 64.  [The first version gave each map a 256-thread workgroup; at P = 180 — four of the six
// guidance keys — most of those threads had no entry and the kernel was bound by the instructions of its scans and barriers.]
LVD_DEV void wave_lds_sync() {  // orders this wave's LDS (and, with the waitcnt the fences imply, global) accesses across its lanes
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
// Wave-wide inclusive scan on the DPP path (row shifts inside the rows of 16 lanes, then the two row broadcasts): six VALU operations.
// __shfl_up goes through ds_bpermute, an LDS round trip per step — and this kernel is one long dependent chain per wave.
LVD_DEV int wave_scan_incl(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);  // row_shr:1
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);  // row_shr:2
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);  // row_shr:4
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);  // row_shr:8
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);  // row_bcast:15 into rows 1 and 3
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);  // row_bcast:31 into rows 2 and 3
  return v;
}
LVD_DEV float dpp_add(float v, int moved) { return v + __int_as_float(moved); }
LVD_DEV float wave_total(float v) {  // sum over the wave, same value in every lane
  v = dpp_add(v, __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x111, 0xf, 0xf, false));
  v = dpp_add(v, __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x112, 0xf, 0xf, false));
  v = dpp_add(v, __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x114, 0xf, 0xf, false));
  v = dpp_add(v, __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x118, 0xf, 0xf, false));
  v = dpp_add(v, __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x142, 0xa, 0xf, false));
  v = dpp_add(v, __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x143, 0xc, 0xf, false));
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
LVD_DEV unsigned long long select_key(float a, int i) { return ((unsigned long long)__float_as_uint(a) << 12) | (unsigned)(4095 - i); }

// Exact top-k sets by radix select on 44-bit keys (value bits << 12 | (4095 - index): ties go to the lowest index), for the entries inside
// the box (class 1, k = k1) and outside it (class 0, k = k0) at once: 8-bit digits from the top, both histograms (hist[2][256], private to
// the wave) built in the same sweep, the bin holding the k-th largest found by a suffix scan (four bins per lane, top bin first).  A class
// stops as soon as the bin it lands in is wanted whole: every key sharing the prefix so far is then in the set, and `key >= prefix` (low
// digits zero) selects exactly what the exact k-th key would — distinct float values leave a one-entry bin after two or three digits, so
// the usual case is 3-4 sweeps, not 6.  thr[w]: entries of class w with key >= thr[w] are its top-k set; ~0 = nothing selected (k <= 0
// or no members).  cls[j]: 0 / 1, or 2 for a lane slot past the end of the map.
template <int E>
LVD_DEV void radix_select2_wave(const float (&a)[E], const int (&cls)[E], int n, int lane, int k0, int k1, bool on0, bool on1, unsigned int* hist,
                                unsigned long long thr[2]) {
  unsigned long long prefix[2] = {0, 0}, maskbits = 0;
  int need[2] = {k0, k1};
  bool act[2] = {on0, on1};  // class still being refined (wave-uniform)
  for (int pass = 5; pass >= 0 && (act[0] || act[1]); --pass) {
    *reinterpret_cast<uint4*>(hist + 4 * lane) = make_uint4(0, 0, 0, 0);
    *reinterpret_cast<uint4*>(hist + 256 + 4 * lane) = make_uint4(0, 0, 0, 0);
    wave_lds_sync();
#pragma unroll
    for (int j = 0; j < E; ++j) {
      if (j * 64 >= n) break;  // wave-uniform: a short map in a kernel built for a longer one
      const int w = cls[j];
      if (w > 1 || !act[w]) continue;
      const unsigned long long key = select_key(a[j], j * 64 + lane);
      if ((key & maskbits) == prefix[w]) atomicAdd(&hist[w * 256 + (int)((key >> (8 * pass)) & 255)], 1u);
    }
    wave_lds_sync();
#pragma unroll
    for (int w = 0; w < 2; ++w) {
      if (!act[w]) continue;
      const uint4 c4 = *reinterpret_cast<const uint4*>(hist + w * 256 + 252 - 4 * lane);
      const int c[4] = {(int)c4.w, (int)c4.z, (int)c4.y, (int)c4.x};  // c[q] = entries in bin 255 - (4 lane + q)
      const int tot = c[0] + c[1] + c[2] + c[3];
      int run = wave_scan_incl(tot) - tot, fbin = -1, frem = 0, fcnt = 0;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int above = run;
        run += c[q];
        // the first bin from the top whose running count reaches `need`; bin 0 takes whatever is left when the class has fewer members
        if ((run >= need[w] && above < need[w]) || (lane == 63 && q == 3 && run < need[w])) { fbin = 255 - (4 * lane + q); frem = need[w] - above; fcnt = c[q]; }
      }
      const unsigned long long hit = __ballot(fbin >= 0);
      if (hit == 0) {  // k <= 0
        act[w] = false;
        prefix[w] = ~0ull;
        continue;
      }
      const int src = __ffsll((long long)hit) - 1;
      fbin = __builtin_amdgcn_readlane(fbin, src); frem = __builtin_amdgcn_readlane(frem, src); fcnt = __builtin_amdgcn_readlane(fcnt, src);
      prefix[w] |= (unsigned long long)fbin << (8 * pass);
      need[w] = frem;
      if (frem == fcnt) act[w] = false;
    }
    maskbits |= 0xffull << (8 * pass);
  }
  thr[0] = on0 ? prefix[0] : ~0ull;
  thr[1] = on1 ? prefix[1] : ~0ull;
}

// ws: this wave's LDS: hist[512] then vals[P] (only the BoxDiff term reads the map back by position)
template <int E>
LVD_DEV void ca_select_wave(const lvd_ca_select_params& p, const int b, unsigned int* ws) {
  unsigned int* hist = ws;
  float* vals = reinterpret_cast<float*>(ws + 512);
  const int lane = threadIdx.x & 63;
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
  const bool boxdiff = p.boxdiff_loss_scale > 0.f;
  (void)h;

  float a[E];
  int cls[E];  // 1 = inside the box, 0 = outside, 2 = past the end of the map
#pragma unroll
  for (int j = 0; j < E; ++j) {
    const int i = j * 64 + lane;
    const bool valid = i < p.P;
    a[j] = 0.f;
    cls[j] = 2;
    if (j * 64 >= p.P) continue;  // wave-uniform
    a[j] = valid ? A[i] : 0.f;
    const int y = i / p.W, x = i - y * p.W;
    cls[j] = !valid ? 2 : ((y >= y0 && y < y1 && x >= x0 && x < x1) ? 1 : 0);
    if (boxdiff && valid) vals[i] = a[j];
  }

  unsigned long long thr_fg = ~0ull, thr_bg = ~0ull;  // "nothing selected"
  const bool ratio = p.use_ratio_loss == 1;
  const bool ce = p.use_ratio_loss == 2;  // NLL form of the top-k energy (utils/guidance.py:363-399)
  if (!ratio) {
    unsigned long long thr[2];
    radix_select2_wave<E>(a, cls, p.P, lane, kbg, kfg, p.P - nmask > 0, nmask > 0, hist, thr);
    thr_bg = thr[0];
    thr_fg = thr[1];
  }
  // ratio-based energy (utils/guidance.py:312-323): act = sum(A*mask) / (sum(A) + eps); loss = mean over heads of (1 - act)^2
  float r_in = 0.f, r_out = 0.f, ratio_loss = 0.f;  // dL/dA[p] = r_in inside the box, r_out outside
  if (ratio) {
    float sa = 0.f, sm = 0.f;
#pragma unroll
    for (int j = 0; j < E; ++j) { sa += a[j]; sm += cls[j] == 1 ? a[j] : 0.f; }
    sa = wave_total(sa); sm = wave_total(sm);
    const float den = sa + p.ratio_eps, act = sm / den;
    ratio_loss = (1.f - act) * (1.f - act) / (float)p.heads;
    const float k2 = -2.f * (1.f - act) / ((float)p.heads * den * den);
    r_in = k2 * (den - sm);
    r_out = k2 * (-sm);
  }
  // attention sync (:401-430).  The reference crops BOTH frames with the box of the NEXT frame (its x_min..y_max were
  // overwritten by the t1 loop), per head: w * mean_box((A_f - A_f+1)^2).  This wave owns frame f: the pair (f, f+1) with
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
  const float gfg = -p.fg_weight / (float)kfg;
  float gbg = p.bg_weight / (float)kbg;
  // CE: the map is clamped to [eps, 1 - eps] first (gradient 1 inside the interval, 0 outside); selecting on the raw values picks the same
  // top-k sums (the clamp is monotone; entries it ties carry no gradient).  fg = mean over the set of -log(a), bg = -log(1 - mean of the
  // set): the background gradient needs that mean before any dA is written, hence the extra sweep.
  const float ce_lo = p.ratio_eps, ce_hi = 1.f - p.ratio_eps;
  if (ce) {
    float pre = 0.f;
#pragma unroll
    for (int j = 0; j < E; ++j) {
      if (j * 64 >= p.P) break;  // wave-uniform
      if (cls[j] == 0 && select_key(a[j], j * 64 + lane) >= thr_bg) pre += fminf(fmaxf(a[j], ce_lo), ce_hi);
    }
    gbg = p.bg_weight / ((1.f - wave_total(pre) / (float)kbg) * (float)kbg);
  }
#pragma unroll
  for (int j = 0; j < E; ++j) {
    const int i = j * 64 + lane;
    if (j * 64 >= p.P) break;  // wave-uniform
    if (cls[j] > 1) continue;
    const float av = a[j];
    const unsigned long long key = select_key(av, i);
    float g = 0.f;
    if (ratio) {
      g += cls[j] ? r_in : r_out;
    } else if (ce) {
      const float ac = fminf(fmaxf(av, ce_lo), ce_hi);
      const float inside = (av >= ce_lo && av <= ce_hi) ? 1.f : 0.f;
      if (cls[j]) {
        if (key >= thr_fg) { sfg -= logf(ac); g += inside * gfg / ac; }
      } else {
        if (key >= thr_bg) { sbg += ac; g += inside * gbg; }
      }
    } else if (cls[j]) {
      if (key >= thr_fg) { sfg += av; g += gfg; }
    } else {
      if (key >= thr_bg) { sbg += av; g += gbg; }
    }
    const int y = i / p.W, x = i - y * p.W;
    if (gy != 0.f || gx != 0.f) g += (gy * ((float)y - cy) + gx * ((float)x - cx)) / S;
    if (Anext && y >= ny0 && y < ny1 && x >= nx0 && x < nx1) {
      const float d = av - Anext[i];
      ssync += d * d;
      g += 2.f * snext * d;
    }
    if (Aprev && cls[j]) g -= 2.f * sprev * (Aprev[i] - av);
    dA[i] = c * g;
  }
  const float tf = wave_total(sfg), tb = wave_total(sbg), tsync = wave_total(ssync);
  float loss = ratio ? ratio_loss : p.fg_weight * (1.f - tf / (float)kfg) + p.bg_weight * (tb / (float)kbg);
  // CE with an empty box: the reference's top-1 of an all-zero masked map is clamped to eps (:381-386) — a constant -log(eps)
  if (ce) loss = p.fg_weight * (nmask > 0 ? tf / (float)kfg : -logf(ce_lo)) - p.bg_weight * logf(1.f - tb / (float)kbg);
  loss += com_loss + snext * tsync;

  // BoxDiff corner constraint (:240-287, 433-465): |max over rows/columns of A - max of the mask| on the corner columns/rows;
  // the gradient goes to the (first) arg-max of each column / row.  dA was written above by other lanes of this wave: the adds below
  // come behind a wave-level fence, and so does the second pass (a cell can be the arg-max of its column and of its row).
  if (boxdiff) {
    const int L = p.boxdiff_L, Hh = p.H, Ww = p.W;
    float cc = 0.f;
    for (int pass = 0; pass < 2; ++pass) {  // pass 0: columns (max over y), pass 1: rows (max over x)
      wave_lds_sync();
      const int n = pass == 0 ? Ww : Hh, m = pass == 0 ? Hh : Ww;
      const int lo = pass == 0 ? x0 : y0, hi2 = pass == 0 ? x1 : y1;
      const float norm = p.boxdiff_normed ? 1.f / ((float)p.heads * (float)n) : 1.f;
      for (int j = lane; j < n; j += 64) {
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
    }
    loss += p.boxdiff_loss_scale * wave_total(cc);
  }
  if (lane == 0) p.loss_partial[b] = wt * loss;
}

// ------------------------------------------------------------------------------------ 3. dQ
LVD_DEV void ca_dq_store(const lvd_ca_dq_params& p, int f, int qi, int h, int hi, const f32x16& dq0, const f32x16& dq1) {
  if (p.acc_mode == 0) {
    // The accumulator holds, per query, channels {8 r + 4 hi .. + 3}: 8-byte pieces interleaved between the two half-waves.  The halves
    // trade every other piece, so each lane stores 16 contiguous bytes and a store instruction covers half as many pieces of cache lines.
    const float fs = p.scale;
    lvd_bf16* op = p.dq + ((long)f * p.P + min(qi, p.P - 1)) * p.lddq + h * 64 + 8 * hi;
#pragma unroll
    for (int blk = 0; blk < 2; ++blk)
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        const f32x16& d = blk ? dq1 : dq0;
        uint2 lo, up;  // channels 16 m + 4 hi .. and 16 m + 8 + 4 hi ..
        lo.x = pack2bf(d[8 * m + 0] * fs, d[8 * m + 1] * fs); lo.y = pack2bf(d[8 * m + 2] * fs, d[8 * m + 3] * fs);
        up.x = pack2bf(d[8 * m + 4] * fs, d[8 * m + 5] * fs); up.y = pack2bf(d[8 * m + 6] * fs, d[8 * m + 7] * fs);
        const uint2 send = hi ? lo : up;
        uint2 recv;
        recv.x = __shfl_xor(send.x, 32, 64); recv.y = __shfl_xor(send.y, 32, 64);
        const uint4 out = hi ? make_uint4(recv.x, recv.y, up.x, up.y) : make_uint4(lo.x, lo.y, recv.x, recv.y);
        if (qi < p.P) stg16(op + blk * 32 + 16 * m, out);
      }
    return;
  }
  if (qi >= p.P) return;
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
  ca_dq_store(p, f, qi, h, hi, dq0, dq1);
}

// Short prompts (<= 96 text positions): four query tiles of one head per workgroup.  K^T of the head (all three key tiles) is staged once
// for the four waves — wave w stages key tile w — behind a single barrier, and every global load of a wave (query rows, LSE, the token
// columns, the twelve key fragments of S = K.Q^T) is in flight before that barrier; the one-wave body above pays two barriers, a staging
// pass and a dependent load per key tile and wave.  bx counts (frame, query tile) pairs; the last workgroup of a head may hold idle waves.
template <int NT>
LVD_DEV void ca_dq_body4(const lvd_ca_dq_params& p, const int bx, const int by) {
  __shared__ uint32_t kt_lds[3][64 * TP];
  __shared__ uint32_t k_lds[96 * KROW];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, l31 = lane & 31, hi = lane >> 5;
  const int nqt = (p.P + 31) >> 5;
  const bool live = bx < nqt * p.frames;
  const int bxc = live ? bx : nqt * p.frames - 1;
  const int f = bxc / nqt, qt = bxc - f * nqt, h = by;
  const int qi = qt * 32 + l31;
  const int qic = min(qi, p.P - 1);
  const lvd_bf16* qp = p.q + ((long)f * p.P + qic) * p.ldq + h * 64 + hi * 8;
  bf16x8 qf[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) qf[ks] = as_bf16x8(ldg16(qp + ks * 16));
  const long row = ((long)f * p.heads + h);
  const float lse2 = p.lse[row * p.P + qic] * 1.4426950408889634f;
  const float sc = p.scale * 1.4426950408889634f;
  float da[NT], pa[NT];
  int tk[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int tt = min(t, p.ntok - 1);
    const long idx = (row * p.ntok + tt) * p.P + qic;
    da[t] = p.dprobs[idx];
    pa[t] = p.probs[idx];
    tk[t] = p.tok_ids[tt];
  }
  const KStage krows = kstage_load(p.k, p.ldk, p.ntext, h);
  if (wave < 3) {
    const int vj = lane & 15, vdc = lane >> 4;
    const int k0 = min(wave * 32 + 2 * vj, p.ntext - 1), k1 = min(wave * 32 + 2 * vj + 1, p.ntext - 1);
    stage_transposed(kt_lds[wave], p.k + (long)k0 * p.ldk + h * 64, p.k + (long)k1 * p.ldk + h * 64, vj, vdc);
  }
  kstage_store(k_lds, krows);
  __syncthreads();
  if (!live) return;
  float cq = 0.f;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    if (t >= p.ntok) { da[t] = 0.f; tk[t] = -1; }
    cq += pa[t] * da[t];
  }
  f32x16 dq0, dq1;
#pragma unroll
  for (int e = 0; e < 16; ++e) { dq0[e] = 0.f; dq1[e] = 0.f; }
#pragma unroll
  for (int kt = 0; kt < 3; ++kt) {
    f32x16 st;
#pragma unroll
    for (int e = 0; e < 16; ++e) st[e] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kstage_frag(k_lds, kt, ks, l31, hi), qf[ks], st, 0, 0, 0);
    // d(score) = prob * (d(prob) - c): d(prob) is non-zero at the object tokens' text positions only.  A token's position is wave-uniform,
    // so only the tile that holds it tests its 16 accumulator slots (the kernel is VALU-bound: a test of every slot against every token
    // was 2/3 of its instructions)
    float g[16], ds[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) g[e] = -cq;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      if ((tk[t] >> 5) != kt) continue;  // also skips the unused slots (tk = -1)
      const int r = (tk[t] & 31) - 4 * hi;
#pragma unroll
      for (int e = 0; e < 16; ++e) g[e] += ((e & 3) + 8 * (e >> 2) == r) ? da[t] : 0.f;
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int kidx = kt * 32 + (e & 3) + 8 * (e >> 2) + 4 * hi;
      ds[e] = kidx < p.ntext ? fast_exp2(st[e] * sc - lse2) * g[e] : 0.f;
    }
#pragma unroll
    for (int ks2 = 0; ks2 < 2; ++ks2) {
      uint4 w;
      w.x = pack2bf(ds[ks2 * 8 + 0], ds[ks2 * 8 + 1]); w.y = pack2bf(ds[ks2 * 8 + 2], ds[ks2 * 8 + 3]);
      w.z = pack2bf(ds[ks2 * 8 + 4], ds[ks2 * 8 + 5]); w.w = pack2bf(ds[ks2 * 8 + 6], ds[ks2 * 8 + 7]);
      const bf16x8 dsf = as_bf16x8(w);
      dq0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_transposed(kt_lds[kt], l31, ks2, hi), dsf, dq0, 0, 0, 0);
      dq1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_transposed(kt_lds[kt], 32 + l31, ks2, hi), dsf, dq1, 0, 0, 0);
    }
  }
  ca_dq_store(p, f, qi, h, hi, dq0, dq1);
}

// ------------------------------------------------------------------------------------ kernels: one key, or all keys of an iteration
// The six guidance keys of an iteration are independent and small (180-720 query positions x 10-20 heads): launched one by one they are
// 18 launches of latency-bound one-wave workgroups.  The multi-key kernels take the per-key parameter blocks as ONE kernel argument and a
// FLAT grid: start[i] is the first workgroup of key i, so no workgroup is launched only to find itself outside its key (with a
// (largest key) x keys grid two thirds of the probs / dq workgroups were such no-ops).
template <class P>
struct KeyTable {
  P k[LVD_CA_MAX_KEYS];
  int start[LVD_CA_MAX_KEYS + 1];
  int nkeys;
};
template <class P>
LVD_DEV int key_of_block(const KeyTable<P>& tab, int b) {
  int key = 0;
#pragma unroll
  for (int i = 1; i < LVD_CA_MAX_KEYS; ++i) key += (i < tab.nkeys && b >= tab.start[i]) ? 1 : 0;
  return key;
}
template <class P>
LVD_DEV int tiles_of(const P& p) { return ((p.P + 31) >> 5) * p.frames; }  // (frame, 32-query tile) pairs of a key

// long prompts (> 96 text positions): the general one-wave bodies, workgroup = (tile, head)
__global__ __launch_bounds__(64) void ca_probs_multi_kernel(const KeyTable<lvd_ca_probs_params> tab) {
  const int key = key_of_block(tab, blockIdx.x);
  const lvd_ca_probs_params& p = tab.k[key];
  const int u = blockIdx.x - tab.start[key], tiles = tiles_of(p);
  ca_probs_body(p, u % tiles, u / tiles);
}
// Four query tiles of one head per workgroup; the head runs fastest over consecutive workgroups, so the 128-byte head slices of the same
// query rows are fetched at about the same time (whole rows, open DRAM pages): 168 -> 154 us for the three stages of the bench layout
// against tile-fastest order, once the key fragments came from LDS (before that the order made no difference: the loads were the limit).
__global__ __launch_bounds__(256) void ca_probs3_multi_kernel(const KeyTable<lvd_ca_probs_params> tab) {
  const int key = key_of_block(tab, blockIdx.x);
  const lvd_ca_probs_params& p = tab.k[key];
  const int u = blockIdx.x - tab.start[key];
  ca_probs_body3(p, (u / p.heads) * 4 + (threadIdx.x >> 6), u % p.heads);
}
__global__ void ca_com_multi_kernel(const KeyTable<lvd_ca_select_params> tab) {
  const int key = key_of_block(tab, blockIdx.x);
  ca_com_body(tab.k[key], blockIdx.x - tab.start[key]);
}
template <int E>
__global__ __launch_bounds__(256) void ca_select_multi_kernel(const KeyTable<lvd_ca_select_params> tab, const int ws_words) {
  extern __shared__ unsigned int select_ws[];
  const int wave = threadIdx.x >> 6;
  const int key = key_of_block(tab, blockIdx.x);
  const lvd_ca_select_params& p = tab.k[key];
  const int b = (blockIdx.x - tab.start[key]) * 4 + wave;
  if (b >= p.frames * p.heads * p.ntok) return;  // the waves of a workgroup never meet at a barrier
  ca_select_wave<E>(p, b, select_ws + wave * ws_words);
}
template <int NT>
__global__ __launch_bounds__(64) void ca_dq_multi_kernel(const KeyTable<lvd_ca_dq_params> tab) {
  const int key = key_of_block(tab, blockIdx.x);
  const lvd_ca_dq_params& p = tab.k[key];
  const int u = blockIdx.x - tab.start[key], tiles = tiles_of(p);
  ca_dq_body<NT>(p, u % tiles, u / tiles);
}
template <int NT>
__global__ __launch_bounds__(256) void ca_dq4_multi_kernel(const KeyTable<lvd_ca_dq_params> tab) {
  const int key = key_of_block(tab, blockIdx.x);
  const lvd_ca_dq_params& p = tab.k[key];
  const int u = blockIdx.x - tab.start[key];
  ca_dq_body4<NT>(p, (u / p.heads) * 4 + (threadIdx.x >> 6), u % p.heads);  // head fastest, as in ca_probs3_multi_kernel
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
  // the bf16 dQ rows leave as 16-byte stores at dq + row * lddq + head * 64 + 8 * k (acc_mode 0 and the closing pass of acc_mode 3)
  if (p->acc_mode == 0 || p->acc_mode == 3)
    LVD_CHECK(p->lddq % 8 == 0 && (reinterpret_cast<uintptr_t>(p->dq) & 15) == 0, "ca_dq: dq must be 16-byte aligned with lddq %% 8 == 0 (lddq=%d)", p->lddq);
  return 0;
}
}  // namespace

// All keys of a guidance iteration in one launch each (3 launches instead of 3 per key; 4 with the centre-of-mass term).  `keys` is a
// host array of `nkeys` (<= LVD_CA_MAX_KEYS) parameter blocks, each exactly what the single-key entry point takes; the single-key entry
// points are the nkeys = 1 case of the same kernels.
namespace {
template <class P, class F>
int fill_table(KeyTable<P>& tab, const P* keys, int nkeys, F blocks_of) {
  tab.nkeys = nkeys;
  int total = 0;
  for (int i = 0; i < LVD_CA_MAX_KEYS; ++i) {
    tab.start[i] = total;
    if (i < nkeys) {
      tab.k[i] = keys[i];
      total += blocks_of(keys[i]);
    }
  }
  tab.start[LVD_CA_MAX_KEYS] = total;
  return total;
}
template <class P>
int host_tiles(const P& p) { return ((p.P + 31) / 32) * p.frames; }
}  // namespace

extern "C" int lvdhip_ca_probs_multi(const lvd_ca_probs_params* keys, int32_t nkeys, void* stream) {
  LVD_CHECK(keys && nkeys >= 1 && nkeys <= LVD_CA_MAX_KEYS, "ca_probs_multi: 1..%d keys", LVD_CA_MAX_KEYS);
  bool brief = true;
  for (int i = 0; i < nkeys; ++i) {
    if (int rc = check_probs(keys + i)) return rc;
    brief = brief && keys[i].ntext <= 96;
  }
  KeyTable<lvd_ca_probs_params> tab;
  if (brief) {
    const int blocks = fill_table(tab, keys, nkeys, [](const lvd_ca_probs_params& p) { return ((host_tiles(p) + 3) / 4) * p.heads; });
    hipLaunchKernelGGL(ca_probs3_multi_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, tab);
  } else {
    const int blocks = fill_table(tab, keys, nkeys, [](const lvd_ca_probs_params& p) { return host_tiles(p) * p.heads; });
    hipLaunchKernelGGL(ca_probs_multi_kernel, dim3(blocks), dim3(64), 0, (hipStream_t)stream, tab);
  }
  LVD_LAUNCH_CHECK();
  return 0;
}

extern "C" int lvdhip_ca_select_multi(const lvd_ca_select_params* keys, int32_t nkeys, void* stream) {
  LVD_CHECK(keys && nkeys >= 1 && nkeys <= LVD_CA_MAX_KEYS, "ca_select_multi: 1..%d keys", LVD_CA_MAX_KEYS);
  int pmax = 0;
  bool com = false, boxdiff = false;
  for (int i = 0; i < nkeys; ++i) {
    if (int rc = check_select(keys + i)) return rc;
    pmax = std::max(pmax, keys[i].P);
    boxdiff = boxdiff || keys[i].boxdiff_loss_scale > 0.f;
    com = com || keys[i].com_loss_scale > 0.f;
    LVD_CHECK((keys[i].com_loss_scale > 0.f) == (keys[0].com_loss_scale > 0.f), "ca_select_multi: the centre-of-mass term is on for all keys or for none");
  }
  hipStream_t s = (hipStream_t)stream;
  if (com) {
    KeyTable<lvd_ca_select_params> tab;
    const int blocks = fill_table(tab, keys, nkeys, [](const lvd_ca_select_params& p) { return p.frames * p.heads * p.ntok; });
    hipLaunchKernelGGL(ca_com_multi_kernel, dim3(blocks), dim3(256), 0, s, tab);
    LVD_LAUNCH_CHECK();
  }
  KeyTable<lvd_ca_select_params> tab;
  const int blocks = fill_table(tab, keys, nkeys, [](const lvd_ca_select_params& p) { return (p.frames * p.heads * p.ntok + 3) / 4; });
  // per wave: two histograms, then (only when a key has the BoxDiff term, which reads the map back by position) the map itself
  const int ws_words = 512 + (boxdiff ? ((pmax + 3) & ~3) : 0);
  const size_t smem = (size_t)4 * ws_words * sizeof(unsigned int);
  const int e = (pmax + 63) / 64;  // map entries per lane
  auto launch = [&](auto kernel) {
    if (smem > 48 * 1024)  // 4 maps of up to 4096 positions: past the default dynamic-LDS limit, well inside the CU's 160 KB
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess) return 1;
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), smem, s, tab, ws_words);
    return 0;
  };
  int rc;
  if (e <= 3) rc = launch(ca_select_multi_kernel<3>);
  else if (e <= 12) rc = launch(ca_select_multi_kernel<12>);
  else if (e <= 24) rc = launch(ca_select_multi_kernel<24>);
  else rc = launch(ca_select_multi_kernel<64>);
  LVD_CHECK(rc == 0, "ca_select_multi: %zu bytes of LDS per workgroup refused", smem);
  LVD_LAUNCH_CHECK();
  return 0;
}

namespace {
template <int NT>
void launch_dq(const KeyTable<lvd_ca_dq_params>& tab, int blocks, bool brief, hipStream_t s) {
  if (brief) hipLaunchKernelGGL(ca_dq4_multi_kernel<NT>, dim3(blocks), dim3(256), 0, s, tab);
  else hipLaunchKernelGGL(ca_dq_multi_kernel<NT>, dim3(blocks), dim3(64), 0, s, tab);
}
}  // namespace

extern "C" int lvdhip_ca_dq_multi(const lvd_ca_dq_params* keys, int32_t nkeys, void* stream) {
  LVD_CHECK(keys && nkeys >= 1 && nkeys <= LVD_CA_MAX_KEYS, "ca_dq_multi: 1..%d keys", LVD_CA_MAX_KEYS);
  int nt = 0;
  bool brief = true;
  for (int i = 0; i < nkeys; ++i) {
    if (int rc = check_dq(keys + i)) return rc;
    nt = std::max(nt, keys[i].ntok);
    brief = brief && keys[i].ntext <= 96;
  }
  KeyTable<lvd_ca_dq_params> tab;
  const int blocks = brief ? fill_table(tab, keys, nkeys, [](const lvd_ca_dq_params& p) { return ((host_tiles(p) + 3) / 4) * p.heads; })
                           : fill_table(tab, keys, nkeys, [](const lvd_ca_dq_params& p) { return host_tiles(p) * p.heads; });
  hipStream_t s = (hipStream_t)stream;
  if (nt <= 2) launch_dq<2>(tab, blocks, brief, s);
  else if (nt <= 4) launch_dq<4>(tab, blocks, brief, s);
  else if (nt <= 8) launch_dq<8>(tab, blocks, brief, s);
  else launch_dq<MAXTOK>(tab, blocks, brief, s);
  LVD_LAUNCH_CHECK();
  return 0;
}

extern "C" int lvdhip_ca_probs_full(const lvd_ca_probs_full_params* p, void* stream) {
  LVD_CHECK(p && p->q && p->k && p->probs, "ca_probs_full: null pointer");
  LVD_CHECK(p->samples > 0 && p->heads > 0 && p->P > 0 && p->samples_per_key > 0, "ca_probs_full: empty problem");
  LVD_CHECK(p->ntext >= 1 && p->ntext <= 96, "ca_probs_full: %d text positions (1..96 supported; CLIP has 77)", p->ntext);
  LVD_CHECK(p->ldq % 8 == 0 && p->ldk % 8 == 0, "ca_probs_full: leading dims");
  const int tiles_per_sample = (((p->P + 31) / 32) + 3) & ~3;
  const long blocks = (long)p->samples * (tiles_per_sample / 4) * p->heads;
  LVD_CHECK(blocks < (1l << 31), "ca_probs_full: grid too large");
  hipLaunchKernelGGL(ca_probs_full_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, *p);
  LVD_LAUNCH_CHECK();
  return 0;
}

extern "C" int lvdhip_ca_apply_probs(const lvd_ca_apply_probs_params* p, void* stream) {
  LVD_CHECK(p && p->probs && p->v && p->out, "ca_apply_probs: null pointer");
  LVD_CHECK(p->samples > 0 && p->heads > 0 && p->P > 0 && p->samples_per_key > 0, "ca_apply_probs: empty problem");
  LVD_CHECK(p->ntext >= 1 && p->ntext <= 256, "ca_apply_probs: %d text positions (1..256 supported)", p->ntext);
  LVD_CHECK(p->ldv % 8 == 0 && p->ldo % 8 == 0 && (reinterpret_cast<uintptr_t>(p->out) & 15) == 0 && (reinterpret_cast<uintptr_t>(p->v) & 15) == 0,
            "ca_apply_probs: v / out rows must be 16-byte aligned");
  LVD_CHECK((long)p->samples * p->heads < 65536, "ca_apply_probs: samples x heads = %ld exceeds the grid", (long)p->samples * p->heads);
  const size_t smem = sizeof(float) * ((size_t)64 * (p->ntext | 1) + (size_t)p->ntext * 64);
  if (smem > 48 * 1024)
    LVD_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(ca_apply_probs_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) == hipSuccess,
              "ca_apply_probs: %zu bytes of LDS refused", smem);
  hipLaunchKernelGGL(ca_apply_probs_kernel, dim3((p->P + 63) / 64, p->samples * p->heads), dim3(256), smem, (hipStream_t)stream, *p);
  LVD_LAUNCH_CHECK();
  return 0;
}

extern "C" int lvdhip_ca_probs(const lvd_ca_probs_params* p, void* stream) { return lvdhip_ca_probs_multi(p, 1, stream); }
extern "C" int lvdhip_ca_select(const lvd_ca_select_params* p, void* stream) { return lvdhip_ca_select_multi(p, 1, stream); }
extern "C" int lvdhip_ca_dq(const lvd_ca_dq_params* p, void* stream) { return lvdhip_ca_dq_multi(p, 1, stream); }
