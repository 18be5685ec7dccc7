// gemm_tile.h — pieces shared by the LDS-DMA GEMM kernels (gemm_ring.hip, conv_halo.hip): A-operand addressing for the
// four loader modes, counted vmcnt waits, and the register-direct epilogue.
#pragma once
#include "common.h"

namespace {

__device__ uint4 g_zero_page[4];

struct RowInfo {
  long off1, off2;
  int oy, ox;
  bool valid;
};

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

typedef __attribute__((ext_vector_type(4))) int v4i;

// One wave-wide LDS-DMA instruction: 64 lanes x 16 bytes from (descriptor base + per-lane byte offset + scalar byte offset)
// to the lane-linear 1 KB at LDS byte address `dst`.  Issued as inline assembly: with the builtin the compiler cannot prove
// that the DMA in flight does not alias the stage being read and puts s_waitcnt vmcnt(0) in front of the first ds_read of
// every phase, which turns the counted waits below into full drains.  M0 (the DMA's LDS base) is saved and restored inside
// the statement (it is compiler-reserved); s_nop 4 covers a VALU-written SGPR operand, s_nop 0 the M0 write.
LVD_DEV void dma16(v4i rsrc, int voff, int soff, unsigned dst) {
  unsigned keep;
  asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %4 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(voff), "s"(rsrc), "s"(dst), "s"(soff)
               : "memory");
}
LVD_DEV v4i make_rsrc(const void* base) {
  const unsigned long b = reinterpret_cast<unsigned long>(base);
  v4i r = {(int)(unsigned)b, (int)((b >> 32) & 0xffffu), 0x7fffffff, 0x00020000};
  return r;
}
LVD_DEV unsigned lds_addr(const void* q) { return (unsigned)(unsigned long)(lptr_t)q; }

// Source address of one 16-byte chunk of the A operand; written with selects (no divergent branches around the DMA).
template <int MODE>
LVD_DEV const lvd_bf16* a_src(const lvd_gemm_params& p, const RowInfo& r, int k0, int klim) {
  const lvd_bf16* z = reinterpret_cast<const lvd_bf16*>(g_zero_page);
  bool ok = r.valid && k0 < klim;
  const lvd_bf16* base;
  long off;
  if (MODE == LVD_A_PLAIN) {
    bool s2 = k0 >= p.c1;
    base = s2 ? p.a2 : p.a1;
    off = s2 ? r.off2 + (k0 - p.c1) : r.off1 + k0;
  } else if (MODE == LVD_A_CONV3X3) {
    int tap = k0 / p.cin;
    int c = k0 - tap * p.cin;
    int ky = tap / 3, kx = tap - 3 * ky;
    int iy = r.oy * p.stride + ky - 1, ix = r.ox * p.stride + kx - 1;
    ok = ok && iy >= 0 && iy < p.hin && ix >= 0 && ix < p.win;
    int ws = p.win >> p.upsample;
    iy >>= p.upsample; ix >>= p.upsample;
    long row = r.off1 + (long)iy * ws + ix;
    bool s2 = c >= p.c1;
    base = s2 ? p.a2 : p.a1;
    off = s2 ? row * p.lda2 + (c - p.c1) : row * p.lda1 + c;
  } else if (MODE == LVD_A_CONV3X3_T2) {
    int tap = k0 / p.cin;
    int c = k0 - tap * p.cin;
    int ky = tap / 3, kx = tap - 3 * ky;
    int ty = r.oy + 1 - ky, tx = r.ox + 1 - kx;
    ok = ok && ty >= 0 && tx >= 0 && !((ty | tx) & 1);
    ty >>= 1; tx >>= 1;
    ok = ok && ty < p.hin && tx < p.win;
    base = p.a1;
    off = (r.off1 + (long)ty * p.win + tx) * p.lda1 + c;
  } else {
    int tap = k0 / p.cin;
    int c = k0 - tap * p.cin;
    int ff = r.oy + tap - 1;
    ok = ok && ff >= 0 && ff < p.frames;
    long row = r.off1 + (long)(tap - 1) * p.hw;
    bool s2 = c >= p.c1;
    base = s2 ? p.a2 : p.a1;
    off = s2 ? row * p.lda2 + (c - p.c1) : row * p.lda1 + c;
  }
  return ok ? base + off : z;
}

template <int N>
LVD_DEV void wait_vmcnt() {
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if constexpr (N == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
  else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
  else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if constexpr (N == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
  else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else if constexpr (N == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
  else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if constexpr (N == 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
  else if constexpr (N == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
  else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  else if constexpr (N == 14) asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
  else if constexpr (N == 15) asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
  else if constexpr (N == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  else if constexpr (N == 18) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
  else if constexpr (N == 20) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
  else if constexpr (N == 21) asm volatile("s_waitcnt vmcnt(21)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// Decode token row m into the loader's per-row state (image / y / x for the convolutions, frame for the temporal conv).
template <int MODE>
LVD_DEV RowInfo make_row(const lvd_gemm_params& p, int m, bool live) {
  RowInfo r;
  r.valid = live && m < p.M;
  r.off1 = 0; r.off2 = 0; r.oy = 0; r.ox = 0;
  if (MODE == LVD_A_PLAIN) {
    r.off1 = (long)m * p.lda1;
    r.off2 = (long)m * p.lda2;
  } else if (MODE == LVD_A_CONV3X3 || MODE == LVD_A_CONV3X3_T2) {
    int plane = p.hout * p.wout;
    int nimg = m / plane;
    int rem = m - nimg * plane;
    r.oy = rem / p.wout;
    r.ox = rem - r.oy * p.wout;
    int hs = p.hin, ws = p.win;
    if (MODE == LVD_A_CONV3X3 && p.upsample) { hs >>= 1; ws >>= 1; }
    r.off1 = (long)nimg * hs * ws;
  } else {
    r.off1 = m;
    r.oy = (m / p.hw) % p.frames;
  }
  return r;
}

// Row map of a wave's accumulator tile: local row (0 .. FM*32) -> token row of the output matrix (>= p.M: nothing to store).
// The ring kernels' tiles are runs of consecutive rows; the temporal tap-GEMM (conv_halo.hip) owns P pixels x F frames.
struct RowsLinear {
  int mbase;
  LVD_DEV int operator()(int local) const { return mbase + local; }
};

// (Bias and temb row-bias are not applied by the epilogues below: the kernels start their accumulators from them, see
// ring_bias_init — the loads then overlap the DMA prologue instead of sitting, one waited-for load per 4 columns, in the
// epilogue, where they were the larger part of its time on the short-K layers.)
template <int FM, int FN, class RM>
LVD_DEV void ring_bias_init(const lvd_gemm_params& p, f32x16 (&acc)[FM][FN], const RM& rm, int nbase, int l31, int hi) {
  // The uniform tests wrap whole load loops: a load alone in a conditional block is waited for at the end of that block,
  // which would serialise the 20-40 loads of a wave (one L2 round trip each) in front of the first MFMA.
  if (p.bias) {
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int n = min(nbase + j * 32 + 8 * q + 4 * hi, p.N - 4);  // clamped: tail columns are never stored
        const f32x4 v = *reinterpret_cast<const f32x4*>(p.bias + n);
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[i][j][4 * q + e] = v[e];
      }
  } else {
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  }
  if (p.rowbias) {
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      const int m = min(rm(i * 32 + l31), p.M - 1);
      const float* rb = p.rowbias + (long)(m / p.rows_per_sample) * (p.ldrowbias ? p.ldrowbias : p.N);
#pragma unroll
      for (int j = 0; j < FN; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(rb + min(nbase + j * 32 + 8 * q + 4 * hi, p.N - 4));
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[i][j][4 * q + e] += v[e];
        }
    }
  }
}

template <int FM, int FN>
LVD_DEV void ring_bias_init(const lvd_gemm_params& p, f32x16 (&acc)[FM][FN], int mbase, int nbase, int l31, int hi) {
  ring_bias_init<FM, FN>(p, acc, RowsLinear{mbase}, nbase, l31, hi);
}

// Epilogue straight from registers.  The MFMAs are issued as D = W_frag · X_frag^T, so lane (l31) owns token row m and
// every 4 consecutive accumulator registers are 4 consecutive output channels: bias / temb row-bias / gate / residual /
// GEGLU are applied on 8-byte row-contiguous vectors with no LDS round trip and no barrier.
template <int FM, int FN, class RM>
LVD_DEV void ring_epilogue(const lvd_gemm_params& p, const f32x16 (&acc)[FM][FN], const RM& rm, int nbase, int l31, int hi) {
#pragma unroll
  for (int i = 0; i < FM; ++i) {
    const int m = rm(i * 32 + l31);
    if (m >= p.M) continue;
    if (p.act == LVD_ACT_GEGLU) {
      lvd_bf16* orow = reinterpret_cast<lvd_bf16*>(p.out) + (long)m * p.ldc;
#pragma unroll
      for (int b = 0; b < FN / 2; ++b)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int n = nbase + b * 64 + 8 * q + 4 * hi;  // hidden column in the interleaved W'; gate = n + 32
          if (n + 32 >= p.N) continue;
          f32x4 h, g;
#pragma unroll
          for (int e = 0; e < 4; ++e) { h[e] = acc[i][2 * b][4 * q + e]; g[e] = acc[i][2 * b + 1][4 * q + e]; }
          uint2 o;
          o = geglu4(h, g);
          stg8(orow + (nbase >> 1) + b * 32 + 8 * q + 4 * hi, o);
        }
      continue;
    }
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int n = nbase + j * 32 + 8 * q + 4 * hi;
        if (n >= p.N) continue;
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * q + e];
        if (p.alpha != 1.f) v *= p.alpha;  // wave-uniform: only the GLIGEN gates scale a product
        if (p.res) {
          uint2 r = ldg8(p.res + (long)m * p.ldres + n);
          v[0] += bflo(r.x); v[1] += bfhi(r.x); v[2] += bflo(r.y); v[3] += bfhi(r.y);
        }
        if (p.out_fp32) {
          float* o = reinterpret_cast<float*>(p.out) + (long)m * p.ldc + n;
          if (p.accumulate) v += *reinterpret_cast<const f32x4*>(o);
          *reinterpret_cast<f32x4*>(o) = v;
        } else {
          lvd_bf16* o = reinterpret_cast<lvd_bf16*>(p.out) + (long)m * p.ldc + n;
          if (p.accumulate) {
            uint2 r = ldg8(o);
            v[0] += bflo(r.x); v[1] += bfhi(r.x); v[2] += bflo(r.y); v[3] += bfhi(r.y);
          }
          uint2 w;
          w.x = pack2bf(v[0], v[1]);
          w.y = pack2bf(v[2], v[3]);
          stg8(o, w);
        }
      }
  }
}

// LayerNorm folded into the product (lvd_gemm_params.ln_mean_rstd): the MFMAs run on the raw rows x and on W' = gamma (.) W, and
// the epilogue turns the accumulator into  rstd_m * (acc - mean_m * colsum_n) + bias_n  (colsum_n = sum_k W'_nk, bias_n = b_n +
// sum_k beta_k W_nk, both prepared by the host) — the normalised activation is never written or read.  mean / rstd of the FM row
// blocks of this lane, and the tile's bias / colsum rows in LDS (already offset to this wave's first column).
template <int FM>
struct LnEpi {
  float mean[FM], rstd[FM];
  const float* b;
  const float* s;
};

// Coalesced epilogue.  The register-direct form above writes 16 bytes per token row per store instruction (32 rows, 32
// different cache lines): on the short-K layers, where the output is as large as the input, those partial-line stores
// were the longest phase of the kernel.  Here each wave transposes its accumulators through a private LDS strip
// (32 rows x W output columns, bf16, after bias / temb row-bias / alpha / GEGLU), then every lane moves 16 bytes so that
// consecutive lanes cover consecutive bytes of a row: full 128-byte lines for the store and for the residual /
// accumulate read.  The residual is added in fp32 to the bf16-rounded projection (what the reference's separate
// residual add does).  Wave-private: no workgroup barrier, the LDS queue keeps one wave's accesses in order.
template <int FM, int FN, bool GEGLU, class RM, bool LN = false>
LVD_DEV void ring_epilogue_rows(const lvd_gemm_params& p, const f32x16 (&acc)[FM][FN], const RM& rm, int nbase, int lane, uint32_t* buf,
                                const LnEpi<FM>* ln = nullptr) {
  constexpr int W = GEGLU ? FN * 16 : FN * 32;  // output columns of this wave
  constexpr int S = W / 2 + 2;                  // dwords per staged row: S = 2 (mod 4) -> the 32 rows of a ds_write_b64 hit 32 distinct
                                                // bank pairs (conflict-free); rows are 8-byte aligned, so the read side uses two b64
  constexpr int CPR = W / 8;                    // 16-byte chunks per row
  constexpr int PASSES = (32 * CPR + 63) / 64;
  const int l31 = lane & 31, hi = lane >> 5;
  const int col0 = GEGLU ? (nbase >> 1) : nbase;
  const int ncols = GEGLU ? (p.N >> 1) : p.N;
  lvd_bf16* out = reinterpret_cast<lvd_bf16*>(p.out);
#pragma unroll
  for (int i = 0; i < FM; ++i) {
    uint32_t* wrow = buf + l31 * S;
    // residual / accumulate operands of this 32-row block: all loads in flight before anything depends on them (the
    // stores below may alias them as far as the compiler knows, so it would otherwise chain load -> store -> load ...)
    // Addresses are clamped instead of predicated: a load inside a per-lane branch makes the compiler wait for it at the end
    // of that branch (vmcnt(0) after every load), which is exactly the serialisation this prefetch exists to avoid.
    uint4 rres[PASSES], racc[PASSES];
    static_assert((32 * CPR) % 64 == 0, "whole passes");
    constexpr bool SIDE = !LN;  // a LayerNorm-folded product has neither residual nor accumulate operand (lvdhip_gemm checks): no prefetch registers
    if (SIDE && p.res) {
#pragma unroll
      for (int ps = 0; ps < PASSES; ++ps) {
        const int idx = ps * 64 + lane;
        const int r = idx / CPR, c = idx - r * CPR;
        const int mm = min(rm(i * 32 + r), p.M - 1);
        const int n = min(col0 + c * 8, ncols - 8);
        rres[ps] = ldg16(p.res + (long)mm * p.ldres + n);
      }
    }
    if (SIDE && p.accumulate) {
#pragma unroll
      for (int ps = 0; ps < PASSES; ++ps) {
        const int idx = ps * 64 + lane;
        const int r = idx / CPR, c = idx - r * CPR;
        const int mm = min(rm(i * 32 + r), p.M - 1);
        const int n = min(col0 + c * 8, ncols - 8);
        racc[ps] = ldg16(out + (long)mm * p.ldc + n);
      }
    }
    if (GEGLU) {
#pragma unroll
      for (int b = 0; b < FN / 2; ++b)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int n = nbase + b * 64 + 8 * q + 4 * hi;  // hidden column in the interleaved W'; gate = n + 32
          f32x4 h, g;
#pragma unroll
          for (int e = 0; e < 4; ++e) { h[e] = acc[i][2 * b][4 * q + e]; g[e] = acc[i][2 * b + 1][4 * q + e]; }
          if constexpr (LN) {
            const int nl = b * 64 + 8 * q + 4 * hi;
            const f32x4 sh = *reinterpret_cast<const f32x4*>(ln->s + nl), sg = *reinterpret_cast<const f32x4*>(ln->s + nl + 32);
            const f32x4 bh = *reinterpret_cast<const f32x4*>(ln->b + nl), bg = *reinterpret_cast<const f32x4*>(ln->b + nl + 32);
            h = ln_fold4(h, ln->mean[i], ln->rstd[i], sh, bh);
            g = ln_fold4(g, ln->mean[i], ln->rstd[i], sg, bg);
          }
          uint2 o;
          o = geglu4(h, g);
          *reinterpret_cast<uint2*>(wrow + b * 16 + 4 * q + 2 * hi) = o;
        }
    } else {
#pragma unroll
      for (int j = 0; j < FN; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int n = nbase + j * 32 + 8 * q + 4 * hi;
          f32x4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * q + e];
          if constexpr (LN) {
            const int nl = j * 32 + 8 * q + 4 * hi;
            const f32x4 sv = *reinterpret_cast<const f32x4*>(ln->s + nl), bv = *reinterpret_cast<const f32x4*>(ln->b + nl);
            v = ln_fold4(v, ln->mean[i], ln->rstd[i], sv, bv);
          }
          if (p.alpha != 1.f) v *= p.alpha;  // wave-uniform: only the GLIGEN gates scale a product
          uint2 o;
          o.x = pack2bf(v[0], v[1]);
          o.y = pack2bf(v[2], v[3]);
          *reinterpret_cast<uint2*>(wrow + j * 16 + 4 * q + 2 * hi) = o;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) {
      const int idx = ps * 64 + lane;
      const int r = idx / CPR, c = idx - r * CPR;
      const int mm = rm(i * 32 + r);
      const int n = col0 + c * 8;
      if (mm < p.M && n < ncols) {
        const uint2 vlo = *reinterpret_cast<const uint2*>(buf + r * S + c * 4);
        const uint2 vhi = *reinterpret_cast<const uint2*>(buf + r * S + c * 4 + 2);
        uint4 v = make_uint4(vlo.x, vlo.y, vhi.x, vhi.y);
        lvd_bf16* o = out + (long)mm * p.ldc + n;
        if (SIDE && (p.res || p.accumulate)) {
          float f[8] = {bflo(v.x), bfhi(v.x), bflo(v.y), bfhi(v.y), bflo(v.z), bfhi(v.z), bflo(v.w), bfhi(v.w)};
          if (p.res) {
            uint4 t = rres[ps];
            f[0] += bflo(t.x); f[1] += bfhi(t.x); f[2] += bflo(t.y); f[3] += bfhi(t.y);
            f[4] += bflo(t.z); f[5] += bfhi(t.z); f[6] += bflo(t.w); f[7] += bfhi(t.w);
          }
          if (p.accumulate) {
            uint4 t = racc[ps];
            f[0] += bflo(t.x); f[1] += bfhi(t.x); f[2] += bflo(t.y); f[3] += bfhi(t.y);
            f[4] += bflo(t.z); f[5] += bfhi(t.z); f[6] += bflo(t.w); f[7] += bfhi(t.w);
          }
          v.x = pack2bf(f[0], f[1]); v.y = pack2bf(f[2], f[3]); v.z = pack2bf(f[4], f[5]); v.w = pack2bf(f[6], f[7]);
        }
        stg16(o, v);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}

// true when the coalesced epilogue applies: bf16 output whose rows (and the residual's) are 16-byte addressable
LVD_DEV bool rows_epilogue_ok(const lvd_gemm_params& p) {
  bool ok = !p.out_fp32 && (p.N & 15) == 0 && (p.ldc & 7) == 0 && (reinterpret_cast<uintptr_t>(p.out) & 15) == 0;
  if (p.res) ok = ok && (p.ldres & 7) == 0 && (reinterpret_cast<uintptr_t>(p.res) & 15) == 0;
  return ok;
}

template <int FM, int FN, class RM>
LVD_DEV void ring_epilogue_auto(const lvd_gemm_params& p, const f32x16 (&acc)[FM][FN], const RM& rm, int nbase, int lane, uint32_t* buf) {
  if (rows_epilogue_ok(p)) {
    if (p.act == LVD_ACT_GEGLU) {
      if constexpr (FN % 2 == 0) ring_epilogue_rows<FM, FN, true>(p, acc, rm, nbase, lane, buf);
    } else {
      ring_epilogue_rows<FM, FN, false>(p, acc, rm, nbase, lane, buf);
    }
  } else {
    ring_epilogue<FM, FN>(p, acc, rm, nbase, lane & 31, lane >> 5);
  }
}
template <int FM, int FN>
LVD_DEV void ring_epilogue_auto(const lvd_gemm_params& p, const f32x16 (&acc)[FM][FN], int mbase, int nbase, int lane, uint32_t* buf) {
  ring_epilogue_auto<FM, FN>(p, acc, RowsLinear{mbase}, nbase, lane, buf);
}

// K-split planning shared by the split-K launchers and the geometry choice.  Cost of c slices in "K elements of one
// tile's main loop": rounds of SLOTS resident workgroups, each K/c long plus a fixed fill/drain, plus c fp32 slabs per
// tile that are written and read back (measured: a 256x256 slab costs about as much as 450 K elements, 128x128 ~160).
inline int lvd_splitk_plan(long tiles, int K, int slots, int slab_cost, long* cost_out) {
  int kmax = K / 256 < 16 ? K / 256 : 16;
  if (kmax < 1) kmax = 1;
  long best = -1;
  int ks = 1;
  for (int c = 1; c <= kmax; ++c) {
    long rounds = (tiles * c + slots - 1) / slots;
    long cost = rounds * (K / c + 128) + (c > 1 ? (long)c * slab_cost : 0);
    if (best < 0 || cost < best) { best = cost; ks = c; }
  }
  if (cost_out) *cost_out = best;
  return ks;
}

}  // namespace

// Slab reduction tail shared by the K-split reduce kernels (gemm_ring.hip, conv_halo.hip): out[m, n..n+3] = epilogue(sum over slices).
// Every load of a quad is issued before the first use: the slices four at a time, the side operands (bias, temb row-bias, residual,
// accumulate target) unconditionally from a harmless address when absent — a load inside a (uniform) branch is waited for inside
// that branch, and one memory round trip per slice and per operand is what these small kernels used to cost (ks + 3 round trips).
// The summation order is the sequential one: ((s0 + s1) + s2) + ...
LVD_DEV void splitk_reduce_quad(const lvd_gemm_params& p, const float* s0, long sstride, int m, int n) {
  const float* safe = p.ws;
  const float* bp = p.bias ? p.bias + n : safe;
  const float* rbp = p.rowbias ? p.rowbias + (long)(m / (p.rows_per_sample > 0 ? p.rows_per_sample : 1)) * (p.ldrowbias ? p.ldrowbias : p.N) + n : safe;
  const lvd_bf16* rp = p.res ? p.res + (long)m * p.ldres + n : reinterpret_cast<const lvd_bf16*>(safe);
  float* of = reinterpret_cast<float*>(p.out) + (long)m * p.ldc + n;
  lvd_bf16* ob = reinterpret_cast<lvd_bf16*>(p.out) + (long)m * p.ldc + n;
  const bool acc32 = p.accumulate && p.out_fp32, acc16 = p.accumulate && !p.out_fp32;
  const f32x4 bv = *reinterpret_cast<const f32x4*>(bp);
  const f32x4 rbv = *reinterpret_cast<const f32x4*>(rbp);
  const uint2 rv = ldg8(rp);
  const f32x4 av32 = *reinterpret_cast<const f32x4*>(acc32 ? of : safe);
  const uint2 av16 = ldg8(acc16 ? ob : reinterpret_cast<const lvd_bf16*>(safe));
  f32x4 v = *reinterpret_cast<const f32x4*>(s0);
  int s = 1;
  for (; s + 7 < p.ksplit; s += 8) {  // eight slices per memory round trip (same sequential order of the additions)
    f32x4 t[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) t[u] = *reinterpret_cast<const f32x4*>(s0 + (long)(s + u) * sstride);
#pragma unroll
    for (int u = 0; u < 8; ++u) v += t[u];
  }
  for (; s + 3 < p.ksplit; s += 4) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(s0 + (long)s * sstride);
    const f32x4 b = *reinterpret_cast<const f32x4*>(s0 + (long)(s + 1) * sstride);
    const f32x4 c = *reinterpret_cast<const f32x4*>(s0 + (long)(s + 2) * sstride);
    const f32x4 d = *reinterpret_cast<const f32x4*>(s0 + (long)(s + 3) * sstride);
    v += a; v += b; v += c; v += d;
  }
  {  // up to three left: clamped slice index, masked value
    const int last = p.ksplit - 1;
    const f32x4 a = *reinterpret_cast<const f32x4*>(s0 + (long)min(s, last) * sstride);
    const f32x4 b = *reinterpret_cast<const f32x4*>(s0 + (long)min(s + 1, last) * sstride);
    const f32x4 c = *reinterpret_cast<const f32x4*>(s0 + (long)min(s + 2, last) * sstride);
    if (s < p.ksplit) v += a;
    if (s + 1 < p.ksplit) v += b;
    if (s + 2 < p.ksplit) v += c;
  }
  if (p.ln_mean_rstd) {  // LayerNorm folded into the product (LnEpi above): the slices hold x . W'^T
    const float2 mr = *reinterpret_cast<const float2*>(p.ln_mean_rstd + 2L * m);
    const f32x4 cs = *reinterpret_cast<const f32x4*>(p.ln_colsum + n);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = mr.y * fmaf(-mr.x, cs[e], v[e]);
  }
  if (p.bias) v += bv;
  if (p.rowbias) v += rbv;
  if (p.alpha != 1.f) v *= p.alpha;  // wave-uniform: only the GLIGEN gates scale a product
  if (p.res) { v[0] += bflo(rv.x); v[1] += bfhi(rv.x); v[2] += bflo(rv.y); v[3] += bfhi(rv.y); }
  if (p.out_fp32) {
    if (p.accumulate) v += av32;
    *reinterpret_cast<f32x4*>(of) = v;
  } else {
    if (p.accumulate) { v[0] += bflo(av16.x); v[1] += bfhi(av16.x); v[2] += bflo(av16.y); v[3] += bfhi(av16.y); }
    uint2 w;
    w.x = pack2bf(v[0], v[1]);
    w.y = pack2bf(v[2], v[3]);
    stg8(ob, w);
  }
}
