// attention.hip — flash-style attention forward for head_dim 64 on gfx950 (MFMA 32x32x16 bf16).
//
// One wave owns a 32-query tile of one (sample, head) and streams the keys in tiles of 32:
//   S^T[key][query] = K · Q^T          A operand = K rows (16-byte global loads, d contiguous)
//                                       B operand = Q rows (resident in registers)
//   => every lane holds 16 of the 32 scores of ITS query (lane&31): the online softmax is
//      lane-local plus one cross-half shuffle, and P never leaves registers.
//   O^T[d][query]  = V^T · P^T          A operand = V^T fragment (V tile transposed through LDS),
//                                       B operand = P registers as they are
//   The contraction (key) index of an MFMA operand can be permuted freely as long as A and B agree:
//   the accumulator layout gives lane (query, hi) the keys {4hi..4hi+3, 4hi+8..4hi+11} (+16·step),
//   so the V^T fragment is read in exactly that key order and no permlane/LDS round trip of P is needed.
// The row addressing (base + step·i per sample) lets the same kernel serve spatial self-attention,
// text cross-attention (77 keys), temporal attention (rows HW apart — the reference's
// (B·F,HW,C)<->(B·HW,F,C) permutes, models/transformer_temporal.py:154-156,175-182, are fused away)
// and the GLIGEN fuser (second key/value segment of 30 grounding tokens, models/attention.py:51-57).
#include <cstdlib>
#include "common.h"

namespace {

LVD_DEV long base_row(int s, int ninner, int os, int is) {
  int so = s / ninner;
  int si = s - so * ninner;
  return (long)so * os + (long)si * is;
}

constexpr float RESCALE_THR = 5.f;  // log2 units: deferred running-max update (see the softmax blocks)
constexpr int VT_PITCH = 18;  // dwords per d-row of the transposed V tile (16 key pairs + 2 pad)

__global__ __launch_bounds__(64, 4) void attn_fwd_kernel(const lvd_attn_params p) {
  __shared__ uint32_t vt[64 * VT_PITCH];
  __shared__ uint4 qk[32 * 8];  // Q tile, then one K tile at a time: [row][16-byte chunk ^ ((row >> 1) & 7)]

  const int lane = threadIdx.x;
  const int l31 = lane & 31, hi = lane >> 5;
  const int nqt = (p.sq + 31) >> 5;
  const int s = blockIdx.x / nqt, qt = blockIdx.x - s * nqt, h = blockIdx.y;

  const long qbase = base_row(s, p.q_ninner, p.q_os, p.q_is);
  const long kvbase = base_row(s, p.kv_ninner, p.kv_os, p.kv_is);
  const long kv2base = p.skv2 > 0 ? base_row(s, p.kv2_ninner, p.kv2_os, p.kv2_is) : 0;
  const int skv_tot = p.skv + p.skv2;

  const int qi = qt * 32 + l31;
  const int qic = min(qi, p.sq - 1);
  const long qrow = qbase + (long)qic * p.q_step;

  // Q and every K tile reach their MFMA fragments through a 4 KB LDS tile: the global loads then cover whole 128-byte head rows
  // (eight lanes per row) instead of 32 bytes of 32 different rows per instruction — four instructions re-touching the same lines,
  // which at 16 waves per CU do not survive in the L1.  XOR chunk swizzle: conflict-free ds_read_b128 in the fragment layout.
  const int sr = lane >> 3, scn = lane & 7;  // staging role: row sr (+8 per instruction), 16-byte chunk scn
  auto frag = [&](int ks) { return as_bf16x8(qk[l31 * 8 + ((ks * 2 + hi) ^ ((l31 >> 1) & 7))]); };
  bf16x8 qf[4];
  {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int r = it * 8 + sr;
      const long row = qbase + (long)min(qt * 32 + r, p.sq - 1) * p.q_step;
      qk[r * 8 + (scn ^ ((r >> 1) & 7))] = ldg16(p.q + row * p.ldq + h * 64 + scn * 8);
    }
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = frag(ks);
    __syncthreads();
  }

  f32x16 o0, o1;
#pragma unroll
  for (int e = 0; e < 16; ++e) { o0[e] = 0.f; o1[e] = 0.f; }
  float m = -1e30f, lsum = 0.f;
  const float sc = p.scale * 1.4426950408889634f;

  const int vj = lane & 15, vdc = lane >> 4;

  for (int kt = 0; kt * 32 < skv_tot; ++kt) {
    // ---- S^T = K · Q^T
    f32x16 st;
#pragma unroll
    for (int e = 0; e < 16; ++e) st[e] = 0.f;
    {
      uint4 kr[4];
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int kk = min(kt * 32 + it * 8 + sr, skv_tot - 1);
        const lvd_bf16* kp = (kk < p.skv) ? p.k + (kvbase + (long)kk * p.kv_step) * p.ldk
                                          : p.k2 + (kv2base + (long)(kk - p.skv) * p.kv2_step) * p.ldk2;
        kr[it] = ldg16(kp + h * 64 + scn * 8);
      }
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int r = it * 8 + sr;
        qk[r * 8 + (scn ^ ((r >> 1) & 7))] = kr[it];
      }
    }
    // ---- V tile -> LDS, transposed: vt[d][key pair]
    {
      int k0 = min(kt * 32 + 2 * vj, skv_tot - 1);
      int k1 = min(kt * 32 + 2 * vj + 1, skv_tot - 1);
      const lvd_bf16* v0 = (k0 < p.skv) ? p.v + (kvbase + (long)k0 * p.kv_step) * p.ldv
                                        : p.v2 + (kv2base + (long)(k0 - p.skv) * p.kv2_step) * p.ldv2;
      const lvd_bf16* v1 = (k1 < p.skv) ? p.v + (kvbase + (long)k1 * p.kv_step) * p.ldv
                                        : p.v2 + (kv2base + (long)(k1 - p.skv) * p.kv2_step) * p.ldv2;
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        int d0 = vdc * 8 + 32 * half;
        uint4 a = ldg16(v0 + h * 64 + d0);
        uint4 b = ldg16(v1 + h * 64 + d0);
        uint32_t aw[4] = {a.x, a.y, a.z, a.w}, bw[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          vt[(d0 + 2 * e) * VT_PITCH + vj] = (aw[e] & 0xffffu) | (bw[e] << 16);
          vt[(d0 + 2 * e + 1) * VT_PITCH + vj] = (aw[e] >> 16) | (bw[e] & 0xffff0000u);
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag(ks), qf[ks], st, 0, 0, 0);

    // ---- online softmax (lane-local over this lane's 16 keys + partner half).  The running max is only raised when
    // some query's tile maximum exceeds it by more than 2^RESCALE_THR (exact: p stays <= 2^THR, l and O share the scale)
    float pv[16];
    float tmax = -1e30f;
    const bool last = (kt + 1) * 32 >= skv_tot;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      float v = st[e] * sc;
      if (last || p.causal) {
        int kidx = kt * 32 + (e & 3) + 8 * (e >> 2) + 4 * hi;
        v = (kidx < skv_tot && !(p.causal && kidx > qi)) ? v : -1e30f;
      }
      pv[e] = v;
      tmax = fmaxf(tmax, v);
    }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
    if (__any(tmax > m + RESCALE_THR)) {
      float mn = fmaxf(m, tmax);
      float alpha = fast_exp2(m - mn);
      lsum *= alpha;
      m = mn;
#pragma unroll
      for (int e = 0; e < 16; ++e) { o0[e] *= alpha; o1[e] *= alpha; }
    }
    float rs = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) { pv[e] = fast_exp2(pv[e] - m); rs += pv[e]; }
    lsum += rs;

    // ---- O^T += V^T · P^T
#pragma unroll
    for (int ks2 = 0; ks2 < 2; ++ks2) {
      uint4 pw;
      pw.x = pack2bf(pv[ks2 * 8 + 0], pv[ks2 * 8 + 1]);
      pw.y = pack2bf(pv[ks2 * 8 + 2], pv[ks2 * 8 + 3]);
      pw.z = pack2bf(pv[ks2 * 8 + 4], pv[ks2 * 8 + 5]);
      pw.w = pack2bf(pv[ks2 * 8 + 6], pv[ks2 * 8 + 7]);
      bf16x8 pf = as_bf16x8(pw);
      int kd = ks2 * 8 + 2 * hi;  // dword offset of keys (16·ks2 + 4·hi)
      {
        const uint32_t* r = vt + l31 * VT_PITCH + kd;
        uint2 lo = *reinterpret_cast<const uint2*>(r);
        uint2 hi2 = *reinterpret_cast<const uint2*>(r + 4);
        bf16x8 vf = as_bf16x8(make_uint4(lo.x, lo.y, hi2.x, hi2.y));
        o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, o0, 0, 0, 0);
      }
      {
        const uint32_t* r = vt + (32 + l31) * VT_PITCH + kd;
        uint2 lo = *reinterpret_cast<const uint2*>(r);
        uint2 hi2 = *reinterpret_cast<const uint2*>(r + 4);
        bf16x8 vf = as_bf16x8(make_uint4(lo.x, lo.y, hi2.x, hi2.y));
        o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, o1, 0, 0, 0);
      }
    }
    __syncthreads();
  }

  float ltot = lsum + __shfl_xor(lsum, 32, 64);
  float inv = 1.f / ltot;
  // O tile back through the LDS tile (the loop's last barrier retired every read of it): whole 128-byte head rows per store
  // instruction instead of 16 bytes of 32 different rows
  {
    uint2* ot = reinterpret_cast<uint2*>(qk);  // [query][16-byte chunk ^ ((query >> 1) & 7)], two uint2 per chunk
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
      uint2 w0, w1;
      w0.x = pack2bf(o0[rq * 4 + 0] * inv, o0[rq * 4 + 1] * inv);
      w0.y = pack2bf(o0[rq * 4 + 2] * inv, o0[rq * 4 + 3] * inv);
      w1.x = pack2bf(o1[rq * 4 + 0] * inv, o1[rq * 4 + 1] * inv);
      w1.y = pack2bf(o1[rq * 4 + 2] * inv, o1[rq * 4 + 3] * inv);
      ot[(l31 * 8 + (rq ^ ((l31 >> 1) & 7))) * 2 + hi] = w0;        // d = 8 rq + 4 hi .. +3
      ot[(l31 * 8 + ((rq + 4) ^ ((l31 >> 1) & 7))) * 2 + hi] = w1;  // d = 32 + 8 rq + 4 hi .. +3
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int r = it * 8 + sr, q = qt * 32 + r;
      if (q < p.sq) stg16(p.o + (qbase + (long)q * p.q_step) * p.ldo + h * 64 + scn * 8, qk[r * 8 + (scn ^ ((r >> 1) & 7))]);
    }
  }
  if (qi < p.sq && p.lse && hi == 0) p.lse[((long)s * p.heads + h) * p.sq + qi] = (m + log2f(ltot)) * 0.6931471805599453f;
}

// ---------------------------------------------------------------------------------------------------------------
// v2: 4 waves share every K/V tile through LDS (self-attention over long sequences: 2880 / 720 keys at the upper UNet
// levels streams K/V once per 128 queries instead of once per 32).  64-key tiles, register-prefetched double buffer:
//   K tile  [64 keys][8 x 16 B] with the XOR chunk swizzle of the GEMM A tile (conflict-free ds_read_b128 fragments)
//   V tile  transposed [64 d][32 key pairs (+2 pad)] written as packed key pairs, read as two ds_read_b64 per fragment
constexpr int V2_VP = 34;

__global__ __launch_bounds__(256, 3) void attn_fwd_v2_kernel(const lvd_attn_params p) {
  __shared__ uint4 k_lds[2][64 * 8];
  __shared__ uint32_t vt_lds[2][64 * V2_VP];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hi = lane >> 5;
  const int nqt = (p.sq + 127) >> 7;
  const int s = blockIdx.x / nqt, qt = blockIdx.x - s * nqt, h = blockIdx.y;
  const long qbase = base_row(s, p.q_ninner, p.q_os, p.q_is);
  const long kvbase = base_row(s, p.kv_ninner, p.kv_os, p.kv_is);
  const int skv = p.skv;

  const int qi = qt * 128 + wave * 32 + l31;
  const int qic = min(qi, p.sq - 1);
  bf16x8 qf[4];
  {
    const lvd_bf16* qp = p.q + (qbase + (long)qic * p.q_step) * p.ldq + h * 64 + hi * 8;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = as_bf16x8(ldg16(qp + ks * 16));
  }
  f32x16 o0, o1;
#pragma unroll
  for (int e = 0; e < 16; ++e) { o0[e] = 0.f; o1[e] = 0.f; }
  float m = -1e30f, lsum = 0.f;
  const float sc = p.scale * 1.4426950408889634f;

  // staging roles
  const int kr = tid >> 3, kc = tid & 7;   // K: rows kr, kr+32; 16-byte chunk kc
  const int vj = tid & 31, vdc = tid >> 5; // V: keys 2vj, 2vj+1; d = 8*vdc..8*vdc+7
  uint4 rk[2], rv[2];
  auto load_tile = [&](int kt) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      int key = min(kt * 64 + kr + 32 * i, skv - 1);
      rk[i] = ldg16(p.k + (kvbase + (long)key * p.kv_step) * p.ldk + h * 64 + kc * 8);
      int vkey = min(kt * 64 + 2 * vj + i, skv - 1);
      rv[i] = ldg16(p.v + (kvbase + (long)vkey * p.kv_step) * p.ldv + h * 64 + vdc * 8);
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      int row = kr + 32 * i;
      k_lds[buf][row * 8 + (kc ^ ((row >> 1) & 7))] = rk[i];
    }
    uint32_t aw[4] = {rv[0].x, rv[0].y, rv[0].z, rv[0].w}, bw[4] = {rv[1].x, rv[1].y, rv[1].z, rv[1].w};
    uint32_t* vt = vt_lds[buf];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      vt[(vdc * 8 + 2 * e) * V2_VP + vj] = (aw[e] & 0xffffu) | (bw[e] << 16);
      vt[(vdc * 8 + 2 * e + 1) * V2_VP + vj] = (aw[e] >> 16) | (bw[e] & 0xffff0000u);
    }
  };

  const int nt = (skv + 63) >> 6;
  load_tile(0);
  store_tile(0);
  // Every global load issued so far (the Q fragments above all) is retired HERE: otherwise the compiler, which merges the loop's
  // entry state with its back edge, keeps a vmcnt wait for the Q registers in front of the first MFMAs of every iteration — and
  // that wait also drains the K/V prefetch issued a few instructions earlier, exposing its whole latency once per tile.
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
  __syncthreads();
  for (int kt = 0; kt < nt; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nt) load_tile(kt + 1);
    const uint4* kl = k_lds[buf];
    const uint32_t* vt = vt_lds[buf];
    f32x16 st[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int e = 0; e < 16; ++e) st[kb][e] = 0.f;
    // the two key blocks alternate: consecutive MFMAs never accumulate into the same registers (a dependent 32x32x16 waits for the
    // whole pass of its predecessor)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        const int row = kb * 32 + l31;
        bf16x8 kf = as_bf16x8(kl[row * 8 + ((ks * 2 + hi) ^ ((row >> 1) & 7))]);
        st[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], st[kb], 0, 0, 0);
      }
    // softmax on the raw scores: the scale is folded into the exponent FMA (max over raw scores, scale > 0)
    float pv[2][16];
    float tmax = -1e30f;
    const bool last = kt + 1 == nt;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        float v = st[kb][e];
        if (last) {
          int kidx = kt * 64 + kb * 32 + (e & 3) + 8 * (e >> 2) + 4 * hi;
          v = (kidx < skv) ? v : -1e30f;
          st[kb][e] = v;
        }
        tmax = fmaxf(tmax, v);
      }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64)) * sc;
    if (__any(tmax > m + RESCALE_THR)) {
      float mn = fmaxf(m, tmax);
      float alpha = fast_exp2(m - mn);
      lsum *= alpha;
      m = mn;
#pragma unroll
      for (int e = 0; e < 16; ++e) { o0[e] *= alpha; o1[e] *= alpha; }
    }
    float rs = 0.f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int e = 0; e < 16; ++e) { pv[kb][e] = fast_exp2(fmaf(st[kb][e], sc, -m)); rs += pv[kb][e]; }
    lsum += rs;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int ks2 = 0; ks2 < 2; ++ks2) {
        uint4 pw;
        pw.x = pack2bf(pv[kb][ks2 * 8 + 0], pv[kb][ks2 * 8 + 1]);
        pw.y = pack2bf(pv[kb][ks2 * 8 + 2], pv[kb][ks2 * 8 + 3]);
        pw.z = pack2bf(pv[kb][ks2 * 8 + 4], pv[kb][ks2 * 8 + 5]);
        pw.w = pack2bf(pv[kb][ks2 * 8 + 6], pv[kb][ks2 * 8 + 7]);
        bf16x8 pf = as_bf16x8(pw);
        const int kd = kb * 16 + ks2 * 8 + 2 * hi;
        {
          const uint32_t* r = vt + l31 * V2_VP + kd;
          uint2 lo = *reinterpret_cast<const uint2*>(r);
          uint2 h2 = *reinterpret_cast<const uint2*>(r + 4);
          o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(make_uint4(lo.x, lo.y, h2.x, h2.y)), pf, o0, 0, 0, 0);
        }
        {
          const uint32_t* r = vt + (32 + l31) * V2_VP + kd;
          uint2 lo = *reinterpret_cast<const uint2*>(r);
          uint2 h2 = *reinterpret_cast<const uint2*>(r + 4);
          o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(make_uint4(lo.x, lo.y, h2.x, h2.y)), pf, o1, 0, 0, 0);
        }
      }
    if (kt + 1 < nt) store_tile(buf ^ 1);
    __syncthreads();
  }

  float ltot = lsum + __shfl_xor(lsum, 32, 64);
  float inv = 1.f / ltot;
  if (qi < p.sq) {
    lvd_bf16* op = p.o + (qbase + (long)qi * p.q_step) * p.ldo + h * 64 + 4 * hi;
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
      uint2 w0, w1;
      w0.x = pack2bf(o0[rq * 4 + 0] * inv, o0[rq * 4 + 1] * inv);
      w0.y = pack2bf(o0[rq * 4 + 2] * inv, o0[rq * 4 + 3] * inv);
      w1.x = pack2bf(o1[rq * 4 + 0] * inv, o1[rq * 4 + 1] * inv);
      w1.y = pack2bf(o1[rq * 4 + 2] * inv, o1[rq * 4 + 3] * inv);
      stg8(op + 8 * rq, w0);
      stg8(op + 32 + 8 * rq, w1);
    }
    if (p.lse && hi == 0) p.lse[((long)s * p.heads + h) * p.sq + qi] = (m + log2f(ltot)) * 0.6931471805599453f;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// v3 (round 5): the v2 structure (4 waves share every 64-key K/V tile, register-prefetched double buffer) with the VALU work per tile cut
// from ~200 to ~120 instructions — at head dim 64 the softmax, not the MFMAs, is what a wave spends its issue slots on:
//   * exp-only softmax.  Q is scaled by scale*log2(e) ONCE (bf16 fragments in registers) and the score MFMAs start from an accumulator
//     that holds -m (the running, deferred maximum of the query row, a 16-register tuple that only changes on a rescale): the tile comes
//     out of the matrix pipe as  s - m  and the probability is ONE v_exp_f32 per score (v2: v_fma + v_exp).  The running maximum lags the
//     true one by at most 2^THR (deferred rescale, as before): a tile whose scores exceed the current shift by more than THR raises it,
//     rescales O and l, and corrects its own scores with one subtraction each — the rare path.
//   * V stays ROW-MAJOR in LDS (two 16-byte stores per thread and tile, as for K; v2 transposed it with ~20 VALU and eight 4-byte stores) and
//     the V^T fragments come from ds_read_b64_tr_b16: within a 16-lane group lane i receives, as element j, element (i & 3) of the 8 bytes
//     lane 4j + (i >> 2) addresses (measured: tools/probes_src/tr_probe.hip) — so lane p of a group addresses V[key0 + (p >> 2)][d0 + 4 (p & 3)]
//     and every lane ends up with four consecutive keys of ITS d row.  The two 64-byte halves of a key row are swapped when bit 1 of the key
//     is set: the four keys a 32-lane half touches then sit in four different 16-bank groups (conflict-free).
//   * per-thread K / V pointers advance by a tile stride (v2 recomputed 64-bit row addresses: ~35 VALU per tile); only a ragged last tile
//     clamps its rows.
//   * O leaves through the idle K tiles as whole 128-byte head rows.
// Level 0 (2880 keys, batch 2): 862 -> 726-755 us (0.59 -> 0.69 PF/s), level 1: 112 -> 102 us (tools/attn_bench.py, same box).
// Measured on the way and dropped (profiles/r05_attention_experiments.txt): a software-pipelined form (score MFMAs of tile j+1 fenced between
// the exponentials of tile j, row maxima between the P.V MFMAs; two score tiles live -> 2 waves per SIMD): correct, 794 us — the third wave
// per SIMD is worth more than the in-wave overlap.  tools/probes_src/{valu_rate,mfma_valu_overlap}.hip hold the unit costs behind that:
// v_exp_f32 5.5 cycles of SIMD time (v_add 2.0, v_max3 / v_cvt_pk 3.0), and an MFMA wave beside a VALU wave keeps its 32 cycles per MFMA.
constexpr float V3_THR = 5.f;

// -DLVD_ATTN_TRACE (developer build, tools/attn_trace.py): wave 0 of one workgroup of attn_fwd_v3_kernel stamps s_memtime at the phase edges of
// a few tiles into p.lse (which the probe over-allocates); never in the shipped library
#ifdef LVD_ATTN_TRACE
#define ATR(tag) do { if (tr_on && tr_n < 200) { const unsigned long t_ = __builtin_amdgcn_s_memtime(); if (lane == 0) { tr_out[tr_n] = (float)(unsigned)(t_ & 0xffffff); tr_out[tr_n + 1] = (float)(tag); } tr_n += 2; } } while (0)
#else
#define ATR(tag) do {} while (0)
#endif

LVD_DEV uint2 lds_tr_b16(unsigned addr, int off) {  // off must be a compile-time constant at every call site (inlined)
  uint2 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(off) : "memory");
  return v;
}

__global__ __launch_bounds__(256, 3) void attn_fwd_v3_kernel(const lvd_attn_params p) {
  __shared__ uint4 k_lds[2][64 * 8];
  __shared__ uint4 v_lds[2][64 * 8];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hi = lane >> 5;
  const int nqt = (p.sq + 127) >> 7;
  const int s = blockIdx.x / nqt, qt = blockIdx.x - s * nqt, h = blockIdx.y;
  const long qbase = base_row(s, p.q_ninner, p.q_os, p.q_is);
  const long kvbase = base_row(s, p.kv_ninner, p.kv_os, p.kv_is);
  const int skv = p.skv;

  const int qi = qt * 128 + wave * 32 + l31;
  const int qic = min(qi, p.sq - 1);
  const float sc = p.scale * 1.4426950408889634f;
  bf16x8 qf[4];
  {
    const lvd_bf16* qp = p.q + (qbase + (long)qic * p.q_step) * p.ldq + h * 64 + hi * 8;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      uint4 u = ldg16(qp + ks * 16);
      u.x = pack2bf(bflo(u.x) * sc, bfhi(u.x) * sc);
      u.y = pack2bf(bflo(u.y) * sc, bfhi(u.y) * sc);
      u.z = pack2bf(bflo(u.z) * sc, bfhi(u.z) * sc);
      u.w = pack2bf(bflo(u.w) * sc, bfhi(u.w) * sc);
      qf[ks] = as_bf16x8(u);
    }
  }
  f32x16 o0, o1, negm;
#pragma unroll
  for (int e = 0; e < 16; ++e) { o0[e] = 0.f; o1[e] = 0.f; negm[e] = 0.f; }
  float m = 0.f, lsum = 0.f;  // m: the shift the scores carry (log2 units); set from the first tile

  // staging roles: rows kr, kr + 32 of the tile; 16-byte chunk kc (K and V alike)
  const int kr = tid >> 3, kc = tid & 7;
  const long ktile = 64L * p.kv_step * p.ldk, khalf = 32L * p.kv_step * p.ldk;
  const long vtile = 64L * p.kv_step * p.ldv, vhalf = 32L * p.kv_step * p.ldv;
  const lvd_bf16* kpt = p.k + (kvbase + (long)kr * p.kv_step) * p.ldk + h * 64 + kc * 8;
  const lvd_bf16* vpt = p.v + (kvbase + (long)kr * p.kv_step) * p.ldv + h * 64 + kc * 8;
  uint4 rk[2], rv[2];
  auto load_tile = [&](int kt) {
    if (kt * 64 + 64 <= skv) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        rk[i] = ldg16(kpt + i * khalf);
        rv[i] = ldg16(vpt + i * vhalf);
      }
    } else {  // ragged last tile: rows past the end re-read the last key (their scores are masked)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int key = min(kt * 64 + kr + 32 * i, skv - 1);
        rk[i] = ldg16(p.k + (kvbase + (long)key * p.kv_step) * p.ldk + h * 64 + kc * 8);
        rv[i] = ldg16(p.v + (kvbase + (long)key * p.kv_step) * p.ldv + h * 64 + kc * 8);
      }
    }
    kpt += ktile;
    vpt += vtile;
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = kr + 32 * i;
      k_lds[buf][row * 8 + (kc ^ ((row >> 1) & 7))] = rk[i];
      v_lds[buf][row * 8 + (kc ^ (((row >> 1) & 1) << 2))] = rv[i];
    }
  };
  // transposed-read address of this lane inside a V tile (bytes): key (4 hi + q), d = 16 * ((lane >> 4) & 1) + 4 * (lane & 3)
  const int tq = (lane & 15) >> 2;
  const int tc = 2 * ((lane >> 4) & 1) + ((lane & 3) >> 1);
  const unsigned vaddr0 = (unsigned)(unsigned long)(__attribute__((address_space(3))) void*)&v_lds[0][0] +
                          (4 * hi + tq) * 128 + ((tc ^ (((tq >> 1) & 1) << 2)) * 16) + (lane & 1) * 8;

  const int nt = (skv + 63) >> 6;
#ifdef LVD_ATTN_TRACE
  float* tr_out = p.lse + (long)p.samples * p.heads * p.sq;
  int tr_n = 0;
  bool tr_on = false;
  const bool tr_blk = blockIdx.x == gridDim.x / 2 && blockIdx.y == 2 && wave == 0;
#endif
  load_tile(0);
  store_tile(0);
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): see v2
  __syncthreads();
  for (int kt = 0; kt < nt; ++kt) {
    const int buf = kt & 1;
#ifdef LVD_ATTN_TRACE
    tr_on = tr_blk && kt >= 8 && kt < 20;
#endif
    ATR(1);
    if (kt + 1 < nt) load_tile(kt + 1);
    const uint4* kl = k_lds[buf];
    f32x16 st[2];
    {
      // all eight K fragments first, ONE wait, then the MFMAs back to back: left to itself hipcc pairs every MFMA with its own ds_read and
      // waits for it (eight exposed LDS round trips per tile: 1100-1900 cycles from the top of a tile to its last score MFMA, tools/attn_trace.py)
#pragma unroll
      for (int half = 0; half < 2; ++half) {  // two batches of four fragments (eight at once do not fit 168 registers)
        bf16x8 kf[2][2];
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
          for (int kb = 0; kb < 2; ++kb) {
            const int row = kb * 32 + l31;
            kf[k2][kb] = as_bf16x8(kl[row * 8 + (((half * 2 + k2) * 2 + hi) ^ ((row >> 1) & 7))]);
          }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
          for (int kb = 0; kb < 2; ++kb)
            st[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[k2][kb], qf[half * 2 + k2], (half == 0 && k2 == 0) ? negm : st[kb], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (kt + 1 == nt && (skv & 63)) {  // ragged last tile: keys past the end out of the maximum and the sums
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int kidx = kt * 64 + kb * 32 + (e & 3) + 8 * (e >> 2) + 4 * hi;
          st[kb][e] = (kidx < skv) ? st[kb][e] : -1e30f;
        }
    }
    ATR(2);
    float tmax = -1e30f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int e = 0; e < 16; ++e) tmax = fmaxf(tmax, st[kb][e]);
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
    ATR(3);
    if (kt == 0 || __any(tmax > V3_THR)) {
      // raise the shift of the rows that need it (every row on the first tile): d = by how much
      const float d = kt == 0 ? tmax : fmaxf(tmax, 0.f);
      if (kt > 0) {
        const float alpha = fast_exp2(-d);
        lsum *= alpha;
#pragma unroll
        for (int e = 0; e < 16; ++e) { o0[e] *= alpha; o1[e] *= alpha; }
      }
      m += d;
#pragma unroll
      for (int e = 0; e < 16; ++e) { negm[e] = -m; st[0][e] -= d; st[1][e] -= d; }
    }
    float rs = 0.f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int e = 0; e < 16; ++e) { st[kb][e] = fast_exp2(st[kb][e]); rs += st[kb][e]; }
    lsum += rs;
    ATR(4);
    const unsigned va = vaddr0 + buf * 8192, vb = va ^ 64;  // d rows 0..31 / 32..63 (the swizzle swaps the row halves by XOR)
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      uint2 t0[2][2], t1[2][2];  // [ks2][first / second group of four keys]
#pragma unroll
      for (int ks2 = 0; ks2 < 2; ++ks2)
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          t0[ks2][r] = lds_tr_b16(va, (kb * 32 + ks2 * 16 + r * 8) * 128);
          t1[ks2][r] = lds_tr_b16(vb, (kb * 32 + ks2 * 16 + r * 8) * 128);
        }
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(t0[0][0]), "+v"(t0[0][1]), "+v"(t0[1][0]), "+v"(t0[1][1]), "+v"(t1[0][0]), "+v"(t1[0][1]), "+v"(t1[1][0]), "+v"(t1[1][1])
                   :
                   : "memory");
#pragma unroll
      for (int ks2 = 0; ks2 < 2; ++ks2) {
        uint4 pw;
        pw.x = pack2bf(st[kb][ks2 * 8 + 0], st[kb][ks2 * 8 + 1]);
        pw.y = pack2bf(st[kb][ks2 * 8 + 2], st[kb][ks2 * 8 + 3]);
        pw.z = pack2bf(st[kb][ks2 * 8 + 4], st[kb][ks2 * 8 + 5]);
        pw.w = pack2bf(st[kb][ks2 * 8 + 6], st[kb][ks2 * 8 + 7]);
        const bf16x8 pf = as_bf16x8(pw);
        o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(make_uint4(t0[ks2][0].x, t0[ks2][0].y, t0[ks2][1].x, t0[ks2][1].y)), pf, o0, 0, 0, 0);
        o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(make_uint4(t1[ks2][0].x, t1[ks2][0].y, t1[ks2][1].x, t1[ks2][1].y)), pf, o1, 0, 0, 0);
      }
    }
    ATR(5);
    if (kt + 1 < nt) store_tile(buf ^ 1);
    ATR(6);
    __syncthreads();
    ATR(7);
  }

  const float ltot = lsum + __shfl_xor(lsum, 32, 64);
  const float inv = 1.f / ltot;
  {
    // O through this wave's quarter of the (idle) K tiles: [query][16-byte chunk ^ ((query >> 1) & 7)], whole 128-byte head rows per store
    uint2* ot = reinterpret_cast<uint2*>(&k_lds[0][0]) + wave * 32 * 16;
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
      uint2 w0, w1;
      w0.x = pack2bf(o0[rq * 4 + 0] * inv, o0[rq * 4 + 1] * inv);
      w0.y = pack2bf(o0[rq * 4 + 2] * inv, o0[rq * 4 + 3] * inv);
      w1.x = pack2bf(o1[rq * 4 + 0] * inv, o1[rq * 4 + 1] * inv);
      w1.y = pack2bf(o1[rq * 4 + 2] * inv, o1[rq * 4 + 3] * inv);
      ot[(l31 * 8 + (rq ^ ((l31 >> 1) & 7))) * 2 + hi] = w0;        // d = 8 rq + 4 hi .. +3
      ot[(l31 * 8 + ((rq + 4) ^ ((l31 >> 1) & 7))) * 2 + hi] = w1;  // d = 32 + 8 rq + 4 hi .. +3
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const uint4* oq = reinterpret_cast<const uint4*>(ot);
    const int sr = lane >> 3, scn = lane & 7;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int r = it * 8 + sr, q = qt * 128 + wave * 32 + r;
      if (q < p.sq) stg16(p.o + (qbase + (long)q * p.q_step) * p.ldo + h * 64 + scn * 8, oq[r * 8 + (scn ^ ((r >> 1) & 7))]);
    }
  }
  if (qi < p.sq && p.lse && hi == 0) p.lse[((long)s * p.heads + h) * p.sq + qi] = (m + log2f(ltot)) * 0.6931471805599453f;
}

}  // namespace

extern "C" int lvdhip_attention_fwd(const lvd_attn_params* p, void* stream) {
  LVD_CHECK(p && p->q && p->k && p->v && p->o, "attention_fwd: null pointer");
  LVD_CHECK(p->sq > 0 && p->skv > 0 && p->samples > 0 && p->heads > 0, "attention_fwd: bad sizes");
  LVD_CHECK(p->skv2 == 0 || (p->k2 && p->v2), "attention_fwd: second KV segment pointers missing");
  LVD_CHECK(p->ldq % 8 == 0 && p->ldk % 8 == 0 && p->ldv % 8 == 0 && p->ldo % 8 == 0, "attention_fwd: leading dims must be multiples of 8");
  LVD_CHECK(p->heads <= 65535, "attention_fwd: too many heads");
  LVD_CHECK(p->q_ninner > 0 && p->kv_ninner > 0, "attention_fwd: ninner must be > 0");
  static int force = -1;
  if (force < 0) { const char* e = getenv("LVD_ATTN_VARIANT"); force = e ? atoi(e) : 0; }
  LVD_CHECK(!p->causal || p->skv2 == 0, "attention_fwd: causal mask with a second KV segment is not defined");
  const bool use_v2 = !p->causal && (force == 2 || force == 3 || (force == 0 && p->skv2 == 0 && p->sq >= 128 && p->skv >= 128));
  if (use_v2 && p->skv2 == 0) {
    dim3 grid(((p->sq + 127) / 128) * p->samples, p->heads);
    // LVD_ATTN_VARIANT: 1 = one-wave kernel, 2 = v2 (round-2 kernel, kept for A/B), 3 / default = v3
    if (force == 2) hipLaunchKernelGGL(attn_fwd_v2_kernel, grid, dim3(256), 0, (hipStream_t)stream, *p);
    else hipLaunchKernelGGL(attn_fwd_v3_kernel, grid, dim3(256), 0, (hipStream_t)stream, *p);
  } else {
    dim3 grid(((p->sq + 31) / 32) * p->samples, p->heads);
    hipLaunchKernelGGL(attn_fwd_kernel, grid, dim3(64), 0, (hipStream_t)stream, *p);
  }
  LVD_LAUNCH_CHECK();
  return 0;
}
