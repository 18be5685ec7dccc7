// attention_bwd.hip — flash-attention input gradients (dQ, dK, dV), head_dim 64, gfx950 MFMA 32x32x16.
//
// Replaces the autograd backward of scaled_dot_product_attention / baddbmm+softmax+bmm that
// torch.autograd.grad(loss, latents) walks in the reference (models/pipelines.py:120).  Same register
// discipline as the forward: the score tile is produced with the reduction-side index (query for dQ,
// key for dK/dV) on the lane axis, so P and dS stay in registers and feed the next MFMA directly as the
// B operand; the operand that must be contracted over its row index (K for dQ; Q and dO for dK/dV) is
// transposed through a small LDS tile in the (hi, j) key order the accumulator layout dictates.
//   dq kernel  : one wave per 32-query tile, loops over key tiles   (also emits delta = rowsum(dO∘O))
//   dkv kernel : one wave per 32-key tile,  loops over query tiles  (skipped for text cross-attention)
#include <cstdlib>
#include "common.h"

namespace {

LVD_DEV long base_row(int s, int ninner, int os, int is) {
  int so = s / ninner;
  int si = s - so * ninner;
  return (long)so * os + (long)si * is;
}

constexpr int TP = 18;  // LDS pitch (dwords) of a transposed [64 d][32 rows] tile

// Stage a 32-row x 64-col bf16 tile transposed: lds[d][row pair].  r0/r1 = this lane's two row pointers
// (rows 2*vj and 2*vj+1 of the tile, already offset to the head), vdc = lane>>4.
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

// A-operand fragment of the transposed tile: row d, k-slots = tile rows {16ks2+4hi+0..3, 16ks2+4hi+8..11}
LVD_DEV bf16x8 frag_transposed(const uint32_t* lds, int d, int ks2, int hi) {
  const uint32_t* r = lds + d * TP + ks2 * 8 + 2 * hi;
  uint2 lo = *reinterpret_cast<const uint2*>(r);
  uint2 h2 = *reinterpret_cast<const uint2*>(r + 4);
  return as_bf16x8(make_uint4(lo.x, lo.y, h2.x, h2.y));
}

LVD_DEV bf16x8 pack8(const float* v) {
  uint4 w;
  w.x = pack2bf(v[0], v[1]); w.y = pack2bf(v[2], v[3]); w.z = pack2bf(v[4], v[5]); w.w = pack2bf(v[6], v[7]);
  return as_bf16x8(w);
}

LVD_DEV float dot8(uint4 a, uint4 b) {
  return bflo(a.x) * bflo(b.x) + bfhi(a.x) * bfhi(b.x) + bflo(a.y) * bflo(b.y) + bfhi(a.y) * bfhi(b.y) +
         bflo(a.z) * bflo(b.z) + bfhi(a.z) * bfhi(b.z) + bflo(a.w) * bflo(b.w) + bfhi(a.w) * bfhi(b.w);
}

// Row-major 32-row x 128-byte LDS tile with an XOR chunk swizzle ([row][16-byte chunk ^ ((row >> 1) & 7)]): global loads and stores cover
// whole head rows (eight lanes per row) and the MFMA fragments are conflict-free ds_read_b128.  sr = lane >> 3 (row within an
// 8-row instruction), scn = lane & 7 (chunk).
LVD_DEV bf16x8 tile_frag(const uint4* tile, int l31, int hi, int ks) { return as_bf16x8(tile[l31 * 8 + ((ks * 2 + hi) ^ ((l31 >> 1) & 7))]); }
LVD_DEV void tile_put(uint4* tile, const uint4 v[4], int sr, int scn) {
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int r = it * 8 + sr;
    tile[r * 8 + (scn ^ ((r >> 1) & 7))] = v[it];
  }
}
// transposed [64 d][32 rows] staging of a tile that already sits row-major in LDS (instead of a second trip to global memory)
LVD_DEV void stage_transposed_from_tile(uint32_t* lds, const uint4* tile, int vj, int vdc) {
  const int r0 = 2 * vj, r1 = 2 * vj + 1;
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    const int c = vdc + 4 * half, d0 = c * 8;
    const uint4 a = tile[r0 * 8 + (c ^ ((r0 >> 1) & 7))];
    const uint4 b = tile[r1 * 8 + (c ^ ((r1 >> 1) & 7))];
    const uint32_t aw[4] = {a.x, a.y, a.z, a.w}, bw[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      lds[(d0 + 2 * e) * TP + vj] = (aw[e] & 0xffffu) | (bw[e] << 16);
      lds[(d0 + 2 * e + 1) * TP + vj] = (aw[e] >> 16) | (bw[e] & 0xffff0000u);
    }
  }
}

// ------------------------------------------------------------------------------------------- dQ
__global__ __launch_bounds__(64) void attn_bwd_dq_kernel(const lvd_attn_bwd_params bp) {
  __shared__ uint32_t kt_lds[64 * TP];
  __shared__ uint4 ta[32 * 8], tb[32 * 8];  // Q then K tiles / dO then V tiles (row-major, swizzled)
  __shared__ float dl[32];
  const lvd_attn_params& p = bp.f;
  const int lane = threadIdx.x;
  const int l31 = lane & 31, hi = lane >> 5;
  const int sr = lane >> 3, scn = lane & 7;
  const int nqt = (p.sq + 31) >> 5;
  const int s = blockIdx.x / nqt, qt = blockIdx.x - s * nqt, h = blockIdx.y;
  const long qbase = base_row(s, p.q_ninner, p.q_os, p.q_is);
  const long kvbase = base_row(s, p.kv_ninner, p.kv_os, p.kv_is);

  const int qi = qt * 32 + l31;
  const int qic = min(qi, p.sq - 1);

  // Q, dO, O as whole rows; delta = rowsum(dO . O) is reduced over the eight lanes of a row before anything is transposed
  bf16x8 qf[4], dof[4];
  {
    uint4 vq[4], vd[4];
    float part[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const long row = qbase + (long)min(qt * 32 + it * 8 + sr, p.sq - 1) * p.q_step;
      vq[it] = ldg16(p.q + row * p.ldq + h * 64 + scn * 8);
      vd[it] = ldg16(bp.d_o + row * bp.lddo + h * 64 + scn * 8);
      part[it] = dot8(vd[it], ldg16(p.o + row * p.ldo + h * 64 + scn * 8));
    }
    tile_put(ta, vq, sr, scn);
    tile_put(tb, vd, sr, scn);
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      float d = part[it];
      d += __shfl_xor(d, 1, 64); d += __shfl_xor(d, 2, 64); d += __shfl_xor(d, 4, 64);
      if (scn == 0) dl[it * 8 + sr] = d;
    }
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) { qf[ks] = tile_frag(ta, l31, hi, ks); dof[ks] = tile_frag(tb, l31, hi, ks); }
  }
  const float delta = dl[l31];
  const long sidx = ((long)s * p.heads + h) * p.sq + qic;
  if (hi == 0 && qi < p.sq) bp.delta[sidx] = delta;
  const float lse2 = p.lse[sidx] * 1.4426950408889634f;
  const float sc = p.scale * 1.4426950408889634f;
  __syncthreads();

  f32x16 dq0, dq1;
#pragma unroll
  for (int e = 0; e < 16; ++e) { dq0[e] = 0.f; dq1[e] = 0.f; }
  const int vj = lane & 15, vdc = lane >> 4;

  for (int kt = 0; kt * 32 < p.skv; ++kt) {
    f32x16 st, dpt;
#pragma unroll
    for (int e = 0; e < 16; ++e) { st[e] = 0.f; dpt[e] = 0.f; }
    {
      uint4 vk[4], vv[4];
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const long krow = kvbase + (long)min(kt * 32 + it * 8 + sr, p.skv - 1) * p.kv_step;
        vk[it] = ldg16(p.k + krow * p.ldk + h * 64 + scn * 8);
        vv[it] = ldg16(p.v + krow * p.ldv + h * 64 + scn * 8);
      }
      tile_put(ta, vk, sr, scn);
      tile_put(tb, vv, sr, scn);
    }
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tile_frag(ta, l31, hi, ks), qf[ks], st, 0, 0, 0);
      dpt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tile_frag(tb, l31, hi, ks), dof[ks], dpt, 0, 0, 0);
    }
    stage_transposed_from_tile(kt_lds, ta, vj, vdc);
    __syncthreads();
    float ds[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      int kidx = kt * 32 + (e & 3) + 8 * (e >> 2) + 4 * hi;
      float pr = (kidx < p.skv) ? fast_exp2(st[e] * sc - lse2) : 0.f;
      ds[e] = pr * (dpt[e] - delta);
    }
#pragma unroll
    for (int ks2 = 0; ks2 < 2; ++ks2) {
      bf16x8 dsf = pack8(ds + ks2 * 8);
      dq0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_transposed(kt_lds, l31, ks2, hi), dsf, dq0, 0, 0, 0);
      dq1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_transposed(kt_lds, 32 + l31, ks2, hi), dsf, dq1, 0, 0, 0);
    }
    __syncthreads();
  }

  {  // dQ tile back through LDS: whole rows per store instruction
    uint2* ot = reinterpret_cast<uint2*>(ta);
    const float f = p.scale;
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
      uint2 w0, w1;
      w0.x = pack2bf(dq0[rq * 4 + 0] * f, dq0[rq * 4 + 1] * f);
      w0.y = pack2bf(dq0[rq * 4 + 2] * f, dq0[rq * 4 + 3] * f);
      w1.x = pack2bf(dq1[rq * 4 + 0] * f, dq1[rq * 4 + 1] * f);
      w1.y = pack2bf(dq1[rq * 4 + 2] * f, dq1[rq * 4 + 3] * f);
      ot[(l31 * 8 + (rq ^ ((l31 >> 1) & 7))) * 2 + hi] = w0;
      ot[(l31 * 8 + ((rq + 4) ^ ((l31 >> 1) & 7))) * 2 + hi] = w1;
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int r = it * 8 + sr, q = qt * 32 + r;
      if (q < p.sq) stg16(bp.dq + (qbase + (long)q * p.q_step) * bp.lddq + h * 64 + scn * 8, ta[r * 8 + (scn ^ ((r >> 1) & 7))]);
    }
  }
}

// ------------------------------------------------------------------------------------------- dK, dV
__global__ __launch_bounds__(64) void attn_bwd_dkv_kernel(const lvd_attn_bwd_params bp) {
  __shared__ uint32_t qt_lds[64 * TP];
  __shared__ uint32_t dot_lds[64 * TP];
  __shared__ uint4 ta[32 * 8], tb[32 * 8];  // K then Q tiles / V then dO tiles (row-major, swizzled), dK / dV on the way out
  const lvd_attn_params& p = bp.f;
  const int lane = threadIdx.x;
  const int l31 = lane & 31, hi = lane >> 5;
  const int sr = lane >> 3, scn = lane & 7;
  const int nkt = (p.skv + 31) >> 5;
  const int s = blockIdx.x / nkt, ktile = blockIdx.x - s * nkt, h = blockIdx.y;
  const long qbase = base_row(s, p.q_ninner, p.q_os, p.q_is);
  const long kvbase = base_row(s, p.kv_ninner, p.kv_os, p.kv_is);

  bf16x8 kf[4], vf[4];
  {
    uint4 vk[4], vv[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const long krow = kvbase + (long)min(ktile * 32 + it * 8 + sr, p.skv - 1) * p.kv_step;
      vk[it] = ldg16(p.k + krow * p.ldk + h * 64 + scn * 8);
      vv[it] = ldg16(p.v + krow * p.ldv + h * 64 + scn * 8);
    }
    tile_put(ta, vk, sr, scn);
    tile_put(tb, vv, sr, scn);
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) { kf[ks] = tile_frag(ta, l31, hi, ks); vf[ks] = tile_frag(tb, l31, hi, ks); }
    __syncthreads();
  }
  const float sc = p.scale * 1.4426950408889634f;
  f32x16 dk0, dk1, dv0, dv1;
#pragma unroll
  for (int e = 0; e < 16; ++e) { dk0[e] = 0.f; dk1[e] = 0.f; dv0[e] = 0.f; dv1[e] = 0.f; }
  const int vj = lane & 15, vdc = lane >> 4;
  const long sbase = ((long)s * p.heads + h) * p.sq;

  for (int qt = 0; qt * 32 < p.sq; ++qt) {
    f32x16 sm, dpm;
#pragma unroll
    for (int e = 0; e < 16; ++e) { sm[e] = 0.f; dpm[e] = 0.f; }
    {
      uint4 vq[4], vd[4];
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const long qrow = qbase + (long)min(qt * 32 + it * 8 + sr, p.sq - 1) * p.q_step;
        vq[it] = ldg16(p.q + qrow * p.ldq + h * 64 + scn * 8);
        vd[it] = ldg16(bp.d_o + qrow * bp.lddo + h * 64 + scn * 8);
      }
      tile_put(ta, vq, sr, scn);
      tile_put(tb, vd, sr, scn);
    }
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      // S[query][key] = Q · K^T ; dP[query][key] = dO · V^T   (rows = queries, cols = this lane's key)
      sm = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tile_frag(ta, l31, hi, ks), kf[ks], sm, 0, 0, 0);
      dpm = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tile_frag(tb, l31, hi, ks), vf[ks], dpm, 0, 0, 0);
    }
    stage_transposed_from_tile(qt_lds, ta, vj, vdc);
    stage_transposed_from_tile(dot_lds, tb, vj, vdc);
    __syncthreads();
    float pr[16], ds[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      int qidx = qt * 32 + (e & 3) + 8 * (e >> 2) + 4 * hi;
      bool ok = qidx < p.sq;
      int qc = ok ? qidx : p.sq - 1;
      float lse2 = p.lse[sbase + qc] * 1.4426950408889634f;
      float dl = bp.delta[sbase + qc];
      float pe = ok ? fast_exp2(sm[e] * sc - lse2) : 0.f;
      pr[e] = pe;
      ds[e] = pe * (dpm[e] - dl);
    }
#pragma unroll
    for (int ks2 = 0; ks2 < 2; ++ks2) {
      bf16x8 pf = pack8(pr + ks2 * 8);
      bf16x8 dsf = pack8(ds + ks2 * 8);
      dv0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_transposed(dot_lds, l31, ks2, hi), pf, dv0, 0, 0, 0);
      dv1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_transposed(dot_lds, 32 + l31, ks2, hi), pf, dv1, 0, 0, 0);
      dk0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_transposed(qt_lds, l31, ks2, hi), dsf, dk0, 0, 0, 0);
      dk1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_transposed(qt_lds, 32 + l31, ks2, hi), dsf, dk1, 0, 0, 0);
    }
    __syncthreads();
  }

  {  // dK, dV tiles back through LDS: whole rows per store instruction
    uint2* ok = reinterpret_cast<uint2*>(ta);
    uint2* ov = reinterpret_cast<uint2*>(tb);
    const float f = p.scale;
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
      uint2 w;
      const int c0 = (l31 * 8 + (rq ^ ((l31 >> 1) & 7))) * 2 + hi, c1 = (l31 * 8 + ((rq + 4) ^ ((l31 >> 1) & 7))) * 2 + hi;
      w.x = pack2bf(dk0[rq * 4 + 0] * f, dk0[rq * 4 + 1] * f); w.y = pack2bf(dk0[rq * 4 + 2] * f, dk0[rq * 4 + 3] * f);
      ok[c0] = w;
      w.x = pack2bf(dk1[rq * 4 + 0] * f, dk1[rq * 4 + 1] * f); w.y = pack2bf(dk1[rq * 4 + 2] * f, dk1[rq * 4 + 3] * f);
      ok[c1] = w;
      w.x = pack2bf(dv0[rq * 4 + 0], dv0[rq * 4 + 1]); w.y = pack2bf(dv0[rq * 4 + 2], dv0[rq * 4 + 3]);
      ov[c0] = w;
      w.x = pack2bf(dv1[rq * 4 + 0], dv1[rq * 4 + 1]); w.y = pack2bf(dv1[rq * 4 + 2], dv1[rq * 4 + 3]);
      ov[c1] = w;
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int r = it * 8 + sr, k = ktile * 32 + r;
      if (k < p.skv) {
        const long krow = kvbase + (long)k * p.kv_step;
        stg16(bp.dk + krow * bp.lddk + h * 64 + scn * 8, ta[r * 8 + (scn ^ ((r >> 1) & 7))]);
        stg16(bp.dv + krow * bp.lddv + h * 64 + scn * 8, tb[r * 8 + (scn ^ ((r >> 1) & 7))]);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------- dQ, dK, dV in one pass
// Sequences of at most 32 queries and 32 keys (temporal attention: 24 frames): one wave owns a whole (sample, head), so the five
// operand tiles (Q, K, V, dO, O) are read ONCE — the dQ and dK/dV kernels above each read four of them and recompute S.
// Both orientations of the score tile are formed: S^T = K·Q^T with the query on the lane axis (softmax statistics lane-local ->
// dS^T -> dQ), and S = Q·K^T with the key on the lane axis (-> P, dS -> dV, dK).
__global__ __launch_bounds__(64) void attn_bwd_small_kernel(const lvd_attn_bwd_params bp) {
  __shared__ uint4 tq[32 * 8], tk[32 * 8], tv[32 * 8], td[32 * 8];  // row-major swizzled tiles: Q, K, V, dO
  __shared__ uint32_t xa[64 * TP], xb[64 * TP];                       // transposed: K^T, then Q^T / dO^T
  __shared__ float dl[32], ls[32];
  const lvd_attn_params& p = bp.f;
  const int lane = threadIdx.x;
  const int l31 = lane & 31, hi = lane >> 5;
  const int sr = lane >> 3, scn = lane & 7;
  const int s = blockIdx.x, h = blockIdx.y;
  const long qbase = base_row(s, p.q_ninner, p.q_os, p.q_is);
  const long kvbase = base_row(s, p.kv_ninner, p.kv_os, p.kv_is);
  const long sbase = ((long)s * p.heads + h) * p.sq;
  {
    uint4 vq[4], vd[4], vk[4], vv[4];
    float part[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const long qrow = qbase + (long)min(it * 8 + sr, p.sq - 1) * p.q_step;
      const long krow = kvbase + (long)min(it * 8 + sr, p.skv - 1) * p.kv_step;
      vq[it] = ldg16(p.q + qrow * p.ldq + h * 64 + scn * 8);
      vd[it] = ldg16(bp.d_o + qrow * bp.lddo + h * 64 + scn * 8);
      vk[it] = ldg16(p.k + krow * p.ldk + h * 64 + scn * 8);
      vv[it] = ldg16(p.v + krow * p.ldv + h * 64 + scn * 8);
      part[it] = dot8(vd[it], ldg16(p.o + qrow * p.ldo + h * 64 + scn * 8));
    }
    tile_put(tq, vq, sr, scn);
    tile_put(td, vd, sr, scn);
    tile_put(tk, vk, sr, scn);
    tile_put(tv, vv, sr, scn);
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      float d = part[it];
      d += __shfl_xor(d, 1, 64); d += __shfl_xor(d, 2, 64); d += __shfl_xor(d, 4, 64);
      if (scn == 0) dl[it * 8 + sr] = d;
    }
    if (lane < 32) ls[lane] = p.lse[sbase + min(lane, p.sq - 1)] * 1.4426950408889634f;
  }
  __syncthreads();
  if (lane < p.sq) bp.delta[sbase + lane] = dl[lane];
  const float sc = p.scale * 1.4426950408889634f;
  const int vj = lane & 15, vdc = lane >> 4;

  // ---- query on the lane axis: dS^T, dQ
  f32x16 dq0, dq1;
  {
    f32x16 st, dpt;
#pragma unroll
    for (int e = 0; e < 16; ++e) { st[e] = 0.f; dpt[e] = 0.f; dq0[e] = 0.f; dq1[e] = 0.f; }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tile_frag(tk, l31, hi, ks), tile_frag(tq, l31, hi, ks), st, 0, 0, 0);
      dpt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tile_frag(tv, l31, hi, ks), tile_frag(td, l31, hi, ks), dpt, 0, 0, 0);
    }
    stage_transposed_from_tile(xa, tk, vj, vdc);
    __syncthreads();
    const float lse2 = ls[l31], delta = dl[l31];
    float ds[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int kidx = (e & 3) + 8 * (e >> 2) + 4 * hi;
      const float pr = (kidx < p.skv) ? fast_exp2(st[e] * sc - lse2) : 0.f;
      ds[e] = pr * (dpt[e] - delta);
    }
#pragma unroll
    for (int ks2 = 0; ks2 < 2; ++ks2) {
      const bf16x8 dsf = pack8(ds + ks2 * 8);
      dq0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_transposed(xa, l31, ks2, hi), dsf, dq0, 0, 0, 0);
      dq1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_transposed(xa, 32 + l31, ks2, hi), dsf, dq1, 0, 0, 0);
    }
    __syncthreads();
  }

  // ---- key on the lane axis: P, dS, dV, dK
  f32x16 dk0, dk1, dv0, dv1;
  {
    f32x16 sm, dpm;
#pragma unroll
    for (int e = 0; e < 16; ++e) { sm[e] = 0.f; dpm[e] = 0.f; dk0[e] = 0.f; dk1[e] = 0.f; dv0[e] = 0.f; dv1[e] = 0.f; }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      sm = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tile_frag(tq, l31, hi, ks), tile_frag(tk, l31, hi, ks), sm, 0, 0, 0);
      dpm = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tile_frag(td, l31, hi, ks), tile_frag(tv, l31, hi, ks), dpm, 0, 0, 0);
    }
    stage_transposed_from_tile(xa, tq, vj, vdc);
    stage_transposed_from_tile(xb, td, vj, vdc);
    __syncthreads();
    float pr[16], ds[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int qidx = (e & 3) + 8 * (e >> 2) + 4 * hi;
      const float pe = (qidx < p.sq) ? fast_exp2(sm[e] * sc - ls[qidx]) : 0.f;
      pr[e] = pe;
      ds[e] = pe * (dpm[e] - dl[qidx]);
    }
#pragma unroll
    for (int ks2 = 0; ks2 < 2; ++ks2) {
      const bf16x8 pf = pack8(pr + ks2 * 8), dsf = pack8(ds + ks2 * 8);
      dv0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_transposed(xb, l31, ks2, hi), pf, dv0, 0, 0, 0);
      dv1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_transposed(xb, 32 + l31, ks2, hi), pf, dv1, 0, 0, 0);
      dk0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_transposed(xa, l31, ks2, hi), dsf, dk0, 0, 0, 0);
      dk1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_transposed(xa, 32 + l31, ks2, hi), dsf, dk1, 0, 0, 0);
    }
    __syncthreads();
  }

  {  // results through the row-major tiles: whole rows per store instruction
    uint2* oq = reinterpret_cast<uint2*>(tq);
    uint2* ok = reinterpret_cast<uint2*>(tk);
    uint2* ov = reinterpret_cast<uint2*>(tv);
    const float f = p.scale;
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
      uint2 w;
      const int c0 = (l31 * 8 + (rq ^ ((l31 >> 1) & 7))) * 2 + hi, c1 = (l31 * 8 + ((rq + 4) ^ ((l31 >> 1) & 7))) * 2 + hi;
      w.x = pack2bf(dq0[rq * 4 + 0] * f, dq0[rq * 4 + 1] * f); w.y = pack2bf(dq0[rq * 4 + 2] * f, dq0[rq * 4 + 3] * f);
      oq[c0] = w;
      w.x = pack2bf(dq1[rq * 4 + 0] * f, dq1[rq * 4 + 1] * f); w.y = pack2bf(dq1[rq * 4 + 2] * f, dq1[rq * 4 + 3] * f);
      oq[c1] = w;
      w.x = pack2bf(dk0[rq * 4 + 0] * f, dk0[rq * 4 + 1] * f); w.y = pack2bf(dk0[rq * 4 + 2] * f, dk0[rq * 4 + 3] * f);
      ok[c0] = w;
      w.x = pack2bf(dk1[rq * 4 + 0] * f, dk1[rq * 4 + 1] * f); w.y = pack2bf(dk1[rq * 4 + 2] * f, dk1[rq * 4 + 3] * f);
      ok[c1] = w;
      w.x = pack2bf(dv0[rq * 4 + 0], dv0[rq * 4 + 1]); w.y = pack2bf(dv0[rq * 4 + 2], dv0[rq * 4 + 3]);
      ov[c0] = w;
      w.x = pack2bf(dv1[rq * 4 + 0], dv1[rq * 4 + 1]); w.y = pack2bf(dv1[rq * 4 + 2], dv1[rq * 4 + 3]);
      ov[c1] = w;
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int r = it * 8 + sr, c = r * 8 + (scn ^ ((r >> 1) & 7));
      if (r < p.sq) stg16(bp.dq + (qbase + (long)r * p.q_step) * bp.lddq + h * 64 + scn * 8, tq[c]);
      if (r < p.skv) {
        const long krow = kvbase + (long)r * p.kv_step;
        stg16(bp.dk + krow * bp.lddk + h * 64 + scn * 8, tk[c]);
        stg16(bp.dv + krow * bp.lddv + h * 64 + scn * 8, tv[c]);
      }
    }
  }
}

// ===============================================================================================================
// v2 kernels for long self-attention sequences: 4 waves share every streamed tile through LDS (row-major copy for
// the A-operand fragments, transposed copy for the contraction over the tile's rows), register-prefetched double
// buffer, one barrier per tile.  Same math and register discipline as the one-wave kernels above.
constexpr int VP2 = 34;  // dwords per d-row of a transposed [64 d][64 rows] tile (32 row pairs + 2 pad)

LVD_DEV void store_rm(uint4* t, int kr, int kc, const uint4* r) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int row = kr + 32 * i;
    t[row * 8 + (kc ^ ((row >> 1) & 7))] = r[i];
  }
}
LVD_DEV void store_tr(uint32_t* t, int vj, int vdc, const uint4* r) {
  uint32_t aw[4] = {r[0].x, r[0].y, r[0].z, r[0].w}, bw[4] = {r[1].x, r[1].y, r[1].z, r[1].w};
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    t[(vdc * 8 + 2 * e) * VP2 + vj] = (aw[e] & 0xffffu) | (bw[e] << 16);
    t[(vdc * 8 + 2 * e + 1) * VP2 + vj] = (aw[e] >> 16) | (bw[e] & 0xffff0000u);
  }
}
LVD_DEV bf16x8 frag_rm(const uint4* t, int row, int c) { return as_bf16x8(t[row * 8 + (c ^ ((row >> 1) & 7))]); }
LVD_DEV bf16x8 frag_tr(const uint32_t* t, int d, int kd) {
  const uint32_t* r = t + d * VP2 + kd;
  uint2 lo = *reinterpret_cast<const uint2*>(r);
  uint2 h2 = *reinterpret_cast<const uint2*>(r + 4);
  return as_bf16x8(make_uint4(lo.x, lo.y, h2.x, h2.y));
}

__global__ __launch_bounds__(256, 3) void attn_bwd_dq_v2_kernel(const lvd_attn_bwd_params bp) {
  __shared__ uint4 k_rm[2][64 * 8];
  __shared__ uint4 v_rm[2][64 * 8];
  __shared__ uint32_t k_tr[2][64 * VP2];
  const lvd_attn_params& p = bp.f;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hi = lane >> 5;
  const int nqt = (p.sq + 127) >> 7;
  const int s = blockIdx.x / nqt, qt = blockIdx.x - s * nqt, h = blockIdx.y;
  const long qbase = base_row(s, p.q_ninner, p.q_os, p.q_is);
  const long kvbase = base_row(s, p.kv_ninner, p.kv_os, p.kv_is);
  const int skv = p.skv;

  const int qi = qt * 128 + wave * 32 + l31;
  const int qic = min(qi, p.sq - 1);
  const long qrow = qbase + (long)qic * p.q_step;
  bf16x8 qf[4], dof[4];
  float delta = 0.f;
  {
    const lvd_bf16* qp = p.q + qrow * p.ldq + h * 64 + hi * 8;
    const lvd_bf16* dp = bp.d_o + qrow * bp.lddo + h * 64 + hi * 8;
    const lvd_bf16* op = p.o + qrow * p.ldo + h * 64 + hi * 8;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      qf[ks] = as_bf16x8(ldg16(qp + ks * 16));
      uint4 d4 = ldg16(dp + ks * 16);
      dof[ks] = as_bf16x8(d4);
      delta += dot8(d4, ldg16(op + ks * 16));
    }
  }
  delta += __shfl_xor(delta, 32, 64);
  const long sidx = ((long)s * p.heads + h) * p.sq + qic;
  if (hi == 0 && qi < p.sq) bp.delta[sidx] = delta;
  const float lse2 = p.lse[sidx] * 1.4426950408889634f;
  const float sc = p.scale * 1.4426950408889634f;

  f32x16 dq0, dq1;
#pragma unroll
  for (int e = 0; e < 16; ++e) { dq0[e] = 0.f; dq1[e] = 0.f; }
  const int kr = tid >> 3, kc = tid & 7, vj = tid & 31, vdc = tid >> 5;
  uint4 rk[2], rv[2], rt[2];
  auto load_tile = [&](int kt) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      int key = min(kt * 64 + kr + 32 * i, skv - 1);
      long row = kvbase + (long)key * p.kv_step;
      rk[i] = ldg16(p.k + row * p.ldk + h * 64 + kc * 8);
      rv[i] = ldg16(p.v + row * p.ldv + h * 64 + kc * 8);
      int tkey = min(kt * 64 + 2 * vj + i, skv - 1);
      rt[i] = ldg16(p.k + (kvbase + (long)tkey * p.kv_step) * p.ldk + h * 64 + vdc * 8);
    }
  };
  auto store_tile = [&](int b) { store_rm(k_rm[b], kr, kc, rk); store_rm(v_rm[b], kr, kc, rv); store_tr(k_tr[b], vj, vdc, rt); };

  const int nt = (skv + 63) >> 6;
  load_tile(0);
  store_tile(0);
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): no pre-loop load may stay "pending" into the loop (see attention.hip, v2 forward)
  __syncthreads();
  for (int kt = 0; kt < nt; ++kt) {
    const int b = kt & 1;
    if (kt + 1 < nt) load_tile(kt + 1);
    const bool last = kt + 1 == nt;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      f32x16 st, dpt;
#pragma unroll
      for (int e = 0; e < 16; ++e) { st[e] = 0.f; dpt[e] = 0.f; }
      const int row = kb * 32 + l31;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rm(k_rm[b], row, ks * 2 + hi), qf[ks], st, 0, 0, 0);
        dpt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rm(v_rm[b], row, ks * 2 + hi), dof[ks], dpt, 0, 0, 0);
      }
      float ds[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        float pr = fast_exp2(st[e] * sc - lse2);
        if (last) {
          int kidx = kt * 64 + kb * 32 + (e & 3) + 8 * (e >> 2) + 4 * hi;
          pr = kidx < skv ? pr : 0.f;
        }
        ds[e] = pr * (dpt[e] - delta);
      }
#pragma unroll
      for (int ks2 = 0; ks2 < 2; ++ks2) {
        bf16x8 dsf = pack8(ds + ks2 * 8);
        const int kd = kb * 16 + ks2 * 8 + 2 * hi;
        dq0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr(k_tr[b], l31, kd), dsf, dq0, 0, 0, 0);
        dq1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr(k_tr[b], 32 + l31, kd), dsf, dq1, 0, 0, 0);
      }
    }
    if (kt + 1 < nt) store_tile(b ^ 1);
    __syncthreads();
  }
  if (qi < p.sq) {
    lvd_bf16* op = bp.dq + (qbase + (long)qi * p.q_step) * bp.lddq + h * 64 + 4 * hi;
    const float f = p.scale;
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
      uint2 w0, w1;
      w0.x = pack2bf(dq0[rq * 4 + 0] * f, dq0[rq * 4 + 1] * f); w0.y = pack2bf(dq0[rq * 4 + 2] * f, dq0[rq * 4 + 3] * f);
      w1.x = pack2bf(dq1[rq * 4 + 0] * f, dq1[rq * 4 + 1] * f); w1.y = pack2bf(dq1[rq * 4 + 2] * f, dq1[rq * 4 + 3] * f);
      stg8(op + 8 * rq, w0);
      stg8(op + 32 + 8 * rq, w1);
    }
  }
}

__global__ __launch_bounds__(256, 2) void attn_bwd_dkv_v2_kernel(const lvd_attn_bwd_params bp) {
  __shared__ uint4 q_rm[2][64 * 8];
  __shared__ uint4 do_rm[2][64 * 8];
  __shared__ uint32_t q_tr[2][64 * VP2];
  __shared__ uint32_t do_tr[2][64 * VP2];
  __shared__ float lse_s[2][64];
  __shared__ float dl_s[2][64];
  const lvd_attn_params& p = bp.f;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hi = lane >> 5;
  const int nkt = (p.skv + 127) >> 7;
  const int s = blockIdx.x / nkt, ktile = blockIdx.x - s * nkt, h = blockIdx.y;
  const long qbase = base_row(s, p.q_ninner, p.q_os, p.q_is);
  const long kvbase = base_row(s, p.kv_ninner, p.kv_os, p.kv_is);
  const int sq = p.sq;

  const int ki = ktile * 128 + wave * 32 + l31;
  const int kic = min(ki, p.skv - 1);
  const long krow = kvbase + (long)kic * p.kv_step;
  bf16x8 kf[4], vf[4];
  {
    const lvd_bf16* kp = p.k + krow * p.ldk + h * 64 + hi * 8;
    const lvd_bf16* vp = p.v + krow * p.ldv + h * 64 + hi * 8;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) { kf[ks] = as_bf16x8(ldg16(kp + ks * 16)); vf[ks] = as_bf16x8(ldg16(vp + ks * 16)); }
  }
  const float sc = p.scale * 1.4426950408889634f;
  f32x16 dk0, dk1, dv0, dv1;
#pragma unroll
  for (int e = 0; e < 16; ++e) { dk0[e] = 0.f; dk1[e] = 0.f; dv0[e] = 0.f; dv1[e] = 0.f; }
  const long sbase = ((long)s * p.heads + h) * sq;
  const int kr = tid >> 3, kc = tid & 7, vj = tid & 31, vdc = tid >> 5;
  uint4 rq[2], rd[2], rqt[2], rdt[2];
  float rstat = 0.f;
  auto load_tile = [&](int qt) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      int q = min(qt * 64 + kr + 32 * i, sq - 1);
      long row = qbase + (long)q * p.q_step;
      rq[i] = ldg16(p.q + row * p.ldq + h * 64 + kc * 8);
      rd[i] = ldg16(bp.d_o + row * bp.lddo + h * 64 + kc * 8);
      int tq = min(qt * 64 + 2 * vj + i, sq - 1);
      long trow = qbase + (long)tq * p.q_step;
      rqt[i] = ldg16(p.q + trow * p.ldq + h * 64 + vdc * 8);
      rdt[i] = ldg16(bp.d_o + trow * bp.lddo + h * 64 + vdc * 8);
    }
    if (tid < 128) {
      int q = min(qt * 64 + (tid & 63), sq - 1);
      rstat = tid < 64 ? p.lse[sbase + q] * 1.4426950408889634f : bp.delta[sbase + q];
    }
  };
  auto store_tile = [&](int b) {
    store_rm(q_rm[b], kr, kc, rq); store_rm(do_rm[b], kr, kc, rd);
    store_tr(q_tr[b], vj, vdc, rqt); store_tr(do_tr[b], vj, vdc, rdt);
    if (tid < 64) lse_s[b][tid] = rstat; else if (tid < 128) dl_s[b][tid - 64] = rstat;
  };

  const int nt = (sq + 63) >> 6;
  load_tile(0);
  store_tile(0);
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): no pre-loop load may stay "pending" into the loop (see attention.hip, v2 forward)
  __syncthreads();
  for (int qt = 0; qt < nt; ++qt) {
    const int b = qt & 1;
    if (qt + 1 < nt) load_tile(qt + 1);
    const bool last = qt + 1 == nt;
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      f32x16 sm, dpm;
#pragma unroll
      for (int e = 0; e < 16; ++e) { sm[e] = 0.f; dpm[e] = 0.f; }
      const int row = qb * 32 + l31;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        sm = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rm(q_rm[b], row, ks * 2 + hi), kf[ks], sm, 0, 0, 0);
        dpm = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rm(do_rm[b], row, ks * 2 + hi), vf[ks], dpm, 0, 0, 0);
      }
      float pr[16], ds[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        int qq = qb * 32 + (e & 3) + 8 * (e >> 2) + 4 * hi;
        float pe = fast_exp2(sm[e] * sc - lse_s[b][qq]);
        if (last) pe = (qt * 64 + qq < sq) ? pe : 0.f;
        pr[e] = pe;
        ds[e] = pe * (dpm[e] - dl_s[b][qq]);
      }
#pragma unroll
      for (int ks2 = 0; ks2 < 2; ++ks2) {
        bf16x8 pf = pack8(pr + ks2 * 8);
        bf16x8 dsf = pack8(ds + ks2 * 8);
        const int kd = qb * 16 + ks2 * 8 + 2 * hi;
        dv0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr(do_tr[b], l31, kd), pf, dv0, 0, 0, 0);
        dv1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr(do_tr[b], 32 + l31, kd), pf, dv1, 0, 0, 0);
        dk0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr(q_tr[b], l31, kd), dsf, dk0, 0, 0, 0);
        dk1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr(q_tr[b], 32 + l31, kd), dsf, dk1, 0, 0, 0);
      }
    }
    if (qt + 1 < nt) store_tile(b ^ 1);
    __syncthreads();
  }
  if (ki < p.skv) {
    lvd_bf16* dkp = bp.dk + krow * bp.lddk + h * 64 + 4 * hi;
    lvd_bf16* dvp = bp.dv + krow * bp.lddv + h * 64 + 4 * hi;
    const float f = p.scale;
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {
      uint2 w;
      w.x = pack2bf(dk0[r4 * 4 + 0] * f, dk0[r4 * 4 + 1] * f); w.y = pack2bf(dk0[r4 * 4 + 2] * f, dk0[r4 * 4 + 3] * f);
      stg8(dkp + 8 * r4, w);
      w.x = pack2bf(dk1[r4 * 4 + 0] * f, dk1[r4 * 4 + 1] * f); w.y = pack2bf(dk1[r4 * 4 + 2] * f, dk1[r4 * 4 + 3] * f);
      stg8(dkp + 32 + 8 * r4, w);
      w.x = pack2bf(dv0[r4 * 4 + 0], dv0[r4 * 4 + 1]); w.y = pack2bf(dv0[r4 * 4 + 2], dv0[r4 * 4 + 3]);
      stg8(dvp + 8 * r4, w);
      w.x = pack2bf(dv1[r4 * 4 + 0], dv1[r4 * 4 + 1]); w.y = pack2bf(dv1[r4 * 4 + 2], dv1[r4 * 4 + 3]);
      stg8(dvp + 32 + 8 * r4, w);
    }
  }
}

}  // namespace

extern "C" int lvdhip_attention_bwd(const lvd_attn_bwd_params* bp, void* stream) {
  LVD_CHECK(bp, "attention_bwd: null params");
  const lvd_attn_params* p = &bp->f;
  LVD_CHECK(p->q && p->k && p->v && p->o && p->lse && bp->d_o && bp->dq && bp->delta, "attention_bwd: null pointer");
  LVD_CHECK(p->skv2 == 0, "attention_bwd: second KV segment is forward-only (the guidance pass never has GLIGEN tokens)");
  LVD_CHECK((bp->dk == nullptr) == (bp->dv == nullptr), "attention_bwd: dk and dv must both be given or both be NULL");
  LVD_CHECK(p->ldq % 8 == 0 && p->ldk % 8 == 0 && p->ldv % 8 == 0 && p->ldo % 8 == 0 && bp->lddo % 8 == 0 && bp->lddq % 8 == 0,
            "attention_bwd: leading dims must be multiples of 8");
  hipStream_t s = (hipStream_t)stream;
  static int force = -1;
  if (force < 0) { const char* e = getenv("LVD_ATTN_VARIANT"); force = e ? atoi(e) : 0; }
  const bool v2 = force == 2 || (force == 0 && p->sq >= 128 && p->skv >= 128);
  if (!v2 && force != 1 && bp->dk && p->sq <= 32 && p->skv <= 32) {  // temporal attention: one pass for dQ, dK, dV
    LVD_CHECK(bp->lddk % 8 == 0 && bp->lddv % 8 == 0, "attention_bwd: lddk/lddv");
    hipLaunchKernelGGL(attn_bwd_small_kernel, dim3(p->samples, p->heads), dim3(64), 0, s, *bp);
    LVD_LAUNCH_CHECK();
    return 0;
  }
  if (v2) {
    dim3 gq(((p->sq + 127) / 128) * p->samples, p->heads);
    hipLaunchKernelGGL(attn_bwd_dq_v2_kernel, gq, dim3(256), 0, s, *bp);
  } else {
    dim3 gq(((p->sq + 31) / 32) * p->samples, p->heads);
    hipLaunchKernelGGL(attn_bwd_dq_kernel, gq, dim3(64), 0, s, *bp);
  }
  LVD_LAUNCH_CHECK();
  if (bp->dk) {
    LVD_CHECK(bp->lddk % 8 == 0 && bp->lddv % 8 == 0, "attention_bwd: lddk/lddv");
    if (v2) {
      dim3 gk(((p->skv + 127) / 128) * p->samples, p->heads);
      hipLaunchKernelGGL(attn_bwd_dkv_v2_kernel, gk, dim3(256), 0, s, *bp);
    } else {
      dim3 gk(((p->skv + 31) / 32) * p->samples, p->heads);
      hipLaunchKernelGGL(attn_bwd_dkv_kernel, gk, dim3(64), 0, s, *bp);
    }
    LVD_LAUNCH_CHECK();
  }
  return 0;
}
