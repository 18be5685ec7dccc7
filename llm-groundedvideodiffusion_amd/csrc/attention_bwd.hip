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

// ------------------------------------------------------------------------------------------- dQ
__global__ __launch_bounds__(64) void attn_bwd_dq_kernel(const lvd_attn_bwd_params bp) {
  __shared__ uint32_t kt_lds[64 * TP];
  const lvd_attn_params& p = bp.f;
  const int lane = threadIdx.x;
  const int l31 = lane & 31, hi = lane >> 5;
  const int nqt = (p.sq + 31) >> 5;
  const int s = blockIdx.x / nqt, qt = blockIdx.x - s * nqt, h = blockIdx.y;
  const long qbase = base_row(s, p.q_ninner, p.q_os, p.q_is);
  const long kvbase = base_row(s, p.kv_ninner, p.kv_os, p.kv_is);

  const int qi = qt * 32 + l31;
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
  const int vj = lane & 15, vdc = lane >> 4;

  for (int kt = 0; kt * 32 < p.skv; ++kt) {
    f32x16 st, dpt;
#pragma unroll
    for (int e = 0; e < 16; ++e) { st[e] = 0.f; dpt[e] = 0.f; }
    {
      int kk = min(kt * 32 + l31, p.skv - 1);
      long krow = kvbase + (long)kk * p.kv_step;
      const lvd_bf16* kp = p.k + krow * p.ldk + h * 64 + hi * 8;
      const lvd_bf16* vp = p.v + krow * p.ldv + h * 64 + hi * 8;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(ldg16(kp + ks * 16)), qf[ks], st, 0, 0, 0);
        dpt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(ldg16(vp + ks * 16)), dof[ks], dpt, 0, 0, 0);
      }
    }
    {
      int k0 = min(kt * 32 + 2 * vj, p.skv - 1), k1 = min(kt * 32 + 2 * vj + 1, p.skv - 1);
      stage_transposed(kt_lds, p.k + (kvbase + (long)k0 * p.kv_step) * p.ldk + h * 64,
                       p.k + (kvbase + (long)k1 * p.kv_step) * p.ldk + h * 64, vj, vdc);
    }
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

  if (qi < p.sq) {
    lvd_bf16* op = bp.dq + (qbase + (long)qi * p.q_step) * bp.lddq + h * 64 + 4 * hi;
    const float f = p.scale;
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
      uint2 w0, w1;
      w0.x = pack2bf(dq0[rq * 4 + 0] * f, dq0[rq * 4 + 1] * f);
      w0.y = pack2bf(dq0[rq * 4 + 2] * f, dq0[rq * 4 + 3] * f);
      w1.x = pack2bf(dq1[rq * 4 + 0] * f, dq1[rq * 4 + 1] * f);
      w1.y = pack2bf(dq1[rq * 4 + 2] * f, dq1[rq * 4 + 3] * f);
      stg8(op + 8 * rq, w0);
      stg8(op + 32 + 8 * rq, w1);
    }
  }
}

// ------------------------------------------------------------------------------------------- dK, dV
__global__ __launch_bounds__(64) void attn_bwd_dkv_kernel(const lvd_attn_bwd_params bp) {
  __shared__ uint32_t qt_lds[64 * TP];
  __shared__ uint32_t dot_lds[64 * TP];
  const lvd_attn_params& p = bp.f;
  const int lane = threadIdx.x;
  const int l31 = lane & 31, hi = lane >> 5;
  const int nkt = (p.skv + 31) >> 5;
  const int s = blockIdx.x / nkt, ktile = blockIdx.x - s * nkt, h = blockIdx.y;
  const long qbase = base_row(s, p.q_ninner, p.q_os, p.q_is);
  const long kvbase = base_row(s, p.kv_ninner, p.kv_os, p.kv_is);

  const int ki = ktile * 32 + l31;
  const int kic = min(ki, p.skv - 1);
  const long krow = kvbase + (long)kic * p.kv_step;
  bf16x8 kf[4], vf[4];
  {
    const lvd_bf16* kp = p.k + krow * p.ldk + h * 64 + hi * 8;
    const lvd_bf16* vp = p.v + krow * p.ldv + h * 64 + hi * 8;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      kf[ks] = as_bf16x8(ldg16(kp + ks * 16));
      vf[ks] = as_bf16x8(ldg16(vp + ks * 16));
    }
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
      int qq = min(qt * 32 + l31, p.sq - 1);
      long qrow = qbase + (long)qq * p.q_step;
      const lvd_bf16* qp = p.q + qrow * p.ldq + h * 64 + hi * 8;
      const lvd_bf16* dp = bp.d_o + qrow * bp.lddo + h * 64 + hi * 8;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        // S[query][key] = Q · K^T ; dP[query][key] = dO · V^T   (rows = queries, cols = this lane's key)
        sm = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(ldg16(qp + ks * 16)), kf[ks], sm, 0, 0, 0);
        dpm = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(ldg16(dp + ks * 16)), vf[ks], dpm, 0, 0, 0);
      }
    }
    {
      int q0 = min(qt * 32 + 2 * vj, p.sq - 1), q1 = min(qt * 32 + 2 * vj + 1, p.sq - 1);
      long r0 = qbase + (long)q0 * p.q_step, r1 = qbase + (long)q1 * p.q_step;
      stage_transposed(qt_lds, p.q + r0 * p.ldq + h * 64, p.q + r1 * p.ldq + h * 64, vj, vdc);
      stage_transposed(dot_lds, bp.d_o + r0 * bp.lddo + h * 64, bp.d_o + r1 * bp.lddo + h * 64, vj, vdc);
    }
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

  if (ki < p.skv) {
    lvd_bf16* dkp = bp.dk + krow * bp.lddk + h * 64 + 4 * hi;
    lvd_bf16* dvp = bp.dv + krow * bp.lddv + h * 64 + 4 * hi;
    const float f = p.scale;
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
      uint2 w;
      w.x = pack2bf(dk0[rq * 4 + 0] * f, dk0[rq * 4 + 1] * f); w.y = pack2bf(dk0[rq * 4 + 2] * f, dk0[rq * 4 + 3] * f);
      stg8(dkp + 8 * rq, w);
      w.x = pack2bf(dk1[rq * 4 + 0] * f, dk1[rq * 4 + 1] * f); w.y = pack2bf(dk1[rq * 4 + 2] * f, dk1[rq * 4 + 3] * f);
      stg8(dkp + 32 + 8 * rq, w);
      w.x = pack2bf(dv0[rq * 4 + 0], dv0[rq * 4 + 1]); w.y = pack2bf(dv0[rq * 4 + 2], dv0[rq * 4 + 3]);
      stg8(dvp + 8 * rq, w);
      w.x = pack2bf(dv1[rq * 4 + 0], dv1[rq * 4 + 1]); w.y = pack2bf(dv1[rq * 4 + 2], dv1[rq * 4 + 3]);
      stg8(dvp + 32 + 8 * rq, w);
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
  LVD_CHECK(p->ldq % 8 == 0 && p->ldk % 8 == 0 && p->ldv % 8 == 0 && p->ldo % 8 == 0 && bp->lddo % 8 == 0 && bp->lddq % 4 == 0,
            "attention_bwd: leading dims must be multiples of 8");
  hipStream_t s = (hipStream_t)stream;
  dim3 gq(((p->sq + 31) / 32) * p->samples, p->heads);
  hipLaunchKernelGGL(attn_bwd_dq_kernel, gq, dim3(64), 0, s, *bp);
  LVD_LAUNCH_CHECK();
  if (bp->dk) {
    LVD_CHECK(bp->lddk % 4 == 0 && bp->lddv % 4 == 0, "attention_bwd: lddk/lddv");
    dim3 gk(((p->skv + 31) / 32) * p->samples, p->heads);
    hipLaunchKernelGGL(attn_bwd_dkv_kernel, gk, dim3(64), 0, s, *bp);
    LVD_LAUNCH_CHECK();
  }
  return 0;
}
