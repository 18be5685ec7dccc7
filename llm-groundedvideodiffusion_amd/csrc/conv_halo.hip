// conv_halo.hip — 3x3 convolution (stride 1, pad 1) as an MFMA GEMM whose im2col tile lives in LDS.
//
// The implicit-im2col loader of gemm_ring.hip stages every tap separately: a 256-row tile DMAs its ~(256 + 2·(W+1)) input
// rows nine times per channel chunk (9x the A-side staging bytes), and the nine passes over K are tap-major, so the re-reads
// fall out of the 4 MB L2 (PMC, round 1: 724 MB fetched for an 88 MB input).  The L2->LDS DMA path is what bounds the 8-wave
// tiles, so the lever is flop per staged byte:
//   * K runs chunk-major: for each 32-channel chunk the tile's rows PLUS their halo (W+1 rows on each side — one image row
//     and one pixel) are staged ONCE into a double-buffered LDS image; the nine taps are nine shifted views of that image:
//     the A fragment of tap (ky,kx) is read at row + (ky-1)·W + (kx-1), and lanes whose tap falls outside the image read a
//     zero line instead (per-lane 9-bit validity mask, one v_cndmask on the LDS address per fragment);
//   * with the activations resident, only the weights are staged per (chunk, tap), and flop per staged byte ~ BM: the tile is
//     512 rows x 160 (128) columns — eight waves stacked along M, each with the 64x160 register tile of the ring kernel — so a
//     K tile stages 10 KB of weights + 4.7 KB of halo for 5.2 MFLOP: 356 flop per staged byte instead of 142;
//   * weights keep the tap-major [Cout, 9·Cin] layout of the rest of the library (a (chunk, tap) tile is a 64-byte run
//     per output channel), so the same packed tensor feeds either kernel.
// Schedule: the 8-wave ping-pong of gemm_ring.hip — waves 4-7 run one phase behind waves 0-3, a K tile is an L phase
// (fragments LDS -> registers, DMA issue, counted wait) and an M phase (20 MFMAs), 3-deep weight ring, 2-deep halo image.
// The tap loop is unrolled (ring slot, tap shift and validity bit are compile-time), the chunk loop is the runtime loop.
// Split-K slices are ranges of chunks.  Epilogue: gemm_tile.h (bias / temb row-bias start the accumulators; coalesced rows).
#include "common.h"
#include "gemm_tile.h"

namespace {

constexpr int HALO_BM = 512;
constexpr int HALO_ROWS = 768;  // 48 wave-instructions of 16 rows: BM + 2·(W+1) <= 768 - 64  (W <= 95; the host asks W <= 87)

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

template <int FN, bool SPLITK>
__global__ __launch_bounds__(512, 2) void conv_halo_kernel(const lvd_gemm_params p) {
  constexpr int FM = 2, NW = 8, RCH = 4, RPI = 16, CH = 32;
  constexpr int BM = HALO_BM, BN = FN * 32;
  constexpr int ASZ = HALO_ROWS * RCH;          // uint4 per halo buffer
  constexpr int BSZ = BN * RCH;                 // uint4 per weight stage
  constexpr int AH = HALO_ROWS / RPI / NW;      // halo instructions per wave per chunk (6), one per tap 0..AH-1
  constexpr int BINS = BN / RPI;                // 10 (8) weight instructions per K tile
  constexpr int BPW = (BINS + NW - 1) / NW;     // per wave, padded (2 / 1)
  constexpr int ZOFF = 2 * ASZ + 3 * BSZ;       // 64-byte zero line
  static_assert((ZOFF & 3) == 0 && (ASZ & 3) == 0, "zero line must keep bit 1 of the chunk index free");
  static_assert(AH <= 9 && HALO_ROWS % (RPI * NW) == 0, "halo image = whole instructions per wave");
  __shared__ uint4 lds[ZOFF + 4];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;

  const int nb = gridDim.x;
  int id;
  {
    int bid = blockIdx.x;
    int q = nb >> 3, r = nb & 7;
    int xcd = bid & 7, idx = bid >> 3;
    id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tiles_n = (p.N + BN - 1) / BN;
  const int ntile = SPLITK ? nb / p.ksplit : nb;
  const int slice = SPLITK ? id / ntile : 0;
  if (SPLITK) id -= slice * ntile;
  const int tm = id / tiles_n, tn = id - tm * tiles_n;
  const int nchunk = p.cin / CH;
  int cbeg = 0, cend = nchunk;
  if (SPLITK) {
    const int per = (nchunk + p.ksplit - 1) / p.ksplit;
    cbeg = slice * per;
    cend = min(nchunk, cbeg + per);
  }

  const int W = p.win, H = p.hin;
  const int halo = W + 1;
  const int m0 = p.m_begin + tm * BM;
  const int cpos = lane & 3, rsub = lane >> 2;
  auto swz = [](int row, int c) { return c ^ ((row >> 2) & 3); };
  // DMA sources are raw buffer descriptors (base + 32-bit per-lane byte offset + scalar byte offset): no 64-bit address
  // VGPRs, no induction variables for the compiler to multiply.  Lanes / instructions with nothing to fetch (halo rows
  // outside the token matrix, weight rows >= N, pipeline overrun) read offset 0 — whatever lands in LDS for them is either
  // never read or masked by the validity bits / never stored.
  const v4i rsA = make_rsrc(p.a1), rsB = make_rsrc(p.w);

  // ---- A side: halo image rows [m0 - halo, m0 + BM + halo) of the token matrix; instruction s = wave·AH + q covers image
  // rows 16·s .. 16·s+15 (64 bytes each).  Offsets are affine in q; validity is re-derived at issue time (one VGPR each).
  const int arows = p.a_rows ? p.a_rows : p.M;
  const int hrow0 = wave * AH * RPI + rsub;
  const int g0 = m0 - halo + hrow0;                              // token row of q = 0
  const int aoff0 = (g0 * p.lda1 + swz(hrow0, cpos) * 8) * 2;    // bytes; swz(hrow0 + 16q) = swz(hrow0)
  const int lastc = nchunk - 1;
  auto stage_a = [&](int c, int q) {
    const int hrow = hrow0 + q * RPI, g = g0 + q * RPI;
    const bool ok = hrow < BM + 2 * halo && g >= 0 && g < arows;
    const int voff = ok ? aoff0 + q * RPI * p.lda1 * 2 : 0;
    dma16(rsA, voff, min(c, lastc) * (CH * 2), lds_addr(lds + ((c - cbeg) & 1) * ASZ + (wave * AH + q) * RPI * RCH));
  };

  // ---- B side: weight rows n of this tile; a (chunk, tap) K tile is the 64-byte run at k = tap·Cin + chunk·32.
  // Instruction b = wave + 8·t covers tile rows 16·b ..; a tile has BINS of them, so waves with wave + 8 >= BINS issue one
  // instruction less (wave-uniform: nbw), and wait for one less.
  const int brow0 = wave * RPI + rsub;
  const int woff0 = (tn * BN + brow0 < p.N) ? ((tn * BN + brow0) * p.K + swz(brow0, cpos) * 8) * 2 : 0;
  const bool two_b = BPW > 1 && wave + NW < BINS;
  const int woff1 = (BPW > 1 && tn * BN + brow0 + NW * RPI < p.N) ? woff0 + NW * RPI * p.K * 2 : 0;
  auto stage_b = [&](int c, int tap, int slot) {
    const int soff = (tap * p.cin + min(c, lastc) * CH) * 2;
    dma16(rsB, woff0, soff, lds_addr(lds + 2 * ASZ + slot * BSZ + wave * RPI * RCH));
    if (two_b) dma16(rsB, woff1, soff, lds_addr(lds + 2 * ASZ + slot * BSZ + (wave + NW) * RPI * RCH));
  };
  static_assert(BPW <= 2, "at most two weight instructions per wave");

  // ---- fragment addressing: rows of fragment i are rloc + 32·i; the validity masks of both fragments share one register
  const int rloc = wave * FM * 32 + l31 + halo;
  int vmask = 0;
  {
    const int plane = H * W;
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      const int m = m0 + wave * FM * 32 + i * 32 + l31;
      const int rem = m % plane;
      const int oy = rem / W, ox = rem - oy * W;
      const int vy = (oy > 0 ? 1 : 0) | 2 | (oy < H - 1 ? 4 : 0);
      const int vx = (ox > 0 ? 1 : 0) | 2 | (ox < W - 1 ? 4 : 0);
      int mk = 0;
#pragma unroll
      for (int t = 0; t < 9; ++t)
        if (((vy >> (t / 3)) & 1) && ((vx >> (t % 3)) & 1)) mk |= 1 << t;
      if (m < p.M) vmask |= mk << (16 * i);
    }
  }
  const int boff0 = l31 * RCH + swz(l31, hi);  // B fragment j sits 32 rows = 128 uint4 further (same swizzle)
  const int bx = (boff0 ^ 2) - boff0;          // second k-step: chunk index ^ 2 (+2 or -2 uint4)

  // prologue: halo of the first chunk, weight tiles 0 and 1
#pragma unroll
  for (int q = 0; q < AH; ++q) stage_a(cbeg, q);
  stage_b(cbeg, 0, 0);
  stage_b(cbeg, 1, 1);
  if (tid < 4) lds[ZOFF + tid] = make_uint4(0u, 0u, 0u, 0u);

  f32x16 acc[FM][FN];
  if (SPLITK) {
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  } else {
    ring_bias_init<FM, FN>(p, acc, m0 + wave * FM * 32, tn * BN, l31, hi);
  }

  const int group = wave >> 2;
  // halo 0 and weight tile 0 have landed (weight tile 1 may still be in flight)
  if (two_b) wait_vmcnt<2>();
  else wait_vmcnt<1>();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (group == 1) __builtin_amdgcn_s_barrier();

  for (int c = cbeg; c < cend; ++c) {
    const int abuf = ((c - cbeg) & 1) * ASZ;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int slot = t % 3, nslot = (t + 2) % 3;
      const int sh = (t / 3 - 1) * W + (t % 3 - 1);
      bf16x8 af[2][FM], bfr[2][FN];
      // opaque copies: keep the nine taps' A addresses and the 3 x 10 B addresses from being precomputed outside the chunk
      // loop (30+ VGPRs that then spill); with a fresh base per tap the fragment reads are base + immediate offset
      int r0 = rloc, b0 = boff0;
      asm volatile("" : "+v"(r0), "+v"(b0));
      const uint4* Bb = lds + 2 * ASZ + slot * BSZ + b0;
#pragma unroll
      for (int i = 0; i < FM; ++i) {
        const int row = r0 + sh + 32 * i;
        int off = abuf + row * RCH + swz(row, hi);
        off = ((vmask >> (16 * i + t)) & 1) ? off : ZOFF;
        af[0][i] = as_bf16x8(lds[off]);
        af[1][i] = as_bf16x8(lds[off ^ 2]);
      }
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        bfr[0][j] = as_bf16x8(Bb[j * 32 * RCH]);
        bfr[1][j] = as_bf16x8(Bb[j * 32 * RCH + bx]);
      }
      // refills: the halo of the next chunk (first AH taps), the weight tile two K tiles ahead
      if (t < AH) stage_a(c + 1, t);
      if (t + 2 < 9) stage_b(c, t + 2, nslot);
      else stage_b(c + 1, t + 2 - 9, nslot);
      // weight tile kt+1 (issued one L phase ago) must have landed: everything younger may stay in flight
      if (two_b) {
        if (t < AH) wait_vmcnt<3>();
        else wait_vmcnt<2>();
      } else {
        if (t < AH) wait_vmcnt<2>();
        else wait_vmcnt<1>();
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
          for (int j = 0; j < FN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[ks][j], af[ks][i], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  if (group == 0) __builtin_amdgcn_s_barrier();
  wait_vmcnt<0>();

  const int mbase = m0 + wave * FM * 32;
  const int nbase = tn * BN;
  if (SPLITK) {
    float* slab = p.ws + ((long)slice * (p.M - p.m_begin) - p.m_begin) * p.N;
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      const int m = mbase + i * 32 + l31;
      if (m >= p.M) continue;
#pragma unroll
      for (int j = 0; j < FN; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int n = nbase + j * 32 + 8 * q + 4 * hi;
          if (n >= p.N) continue;
          f32x4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * q + e];
          *reinterpret_cast<f32x4*>(slab + (long)m * p.N + n) = v;
        }
    }
    return;
  }
  constexpr int WAVE_DW = (ZOFF * 4) / NW;
  static_assert(WAVE_DW >= 32 * (FN * 16 + 2), "LDS too small for the epilogue strip");
  __builtin_amdgcn_s_barrier();
  ring_epilogue_auto<FM, FN>(p, acc, mbase, nbase, lane, reinterpret_cast<uint32_t*>(lds) + wave * WAVE_DW);
}

}  // namespace

void lvd_splitk_reduce_launch(const lvd_gemm_params* p, void* stream);  // gemm_ring.hip

// true when the halo kernel can run this product: stride-1 pad-1 conv from ONE source, 32-channel chunks, an image row + 1
// pixel of halo on each side within the LDS image, 32-bit element offsets
bool lvd_conv_halo_eligible(const lvd_gemm_params* p) {
  return p->mode == LVD_A_CONV3X3 && p->stride == 1 && p->upsample == 0 && p->a2 == nullptr && p->cin % 32 == 0 && p->c1 >= p->cin &&
         p->hin == p->hout && p->win == p->wout && p->win <= 87 && (long)(p->a_rows ? p->a_rows : p->M) * p->lda1 < (1L << 30) &&
         (long)p->N * p->K < (1L << 30);
}

// splitk = 0: one workgroup per 512 x (160|128) tile.  splitk = 1: K (channel chunks) split over workgroups into fp32 slabs +
// deterministic reduce (returns -1 when the product cannot be split: caller falls back to the unsplit launch).
int lvd_conv_halo_dispatch(const lvd_gemm_params* pp, void* stream, int splitk) {
  hipStream_t s = (hipStream_t)stream;
  lvd_gemm_params p = *pp;
  const bool n320 = p.N % 160 == 0;
  const int bn = n320 ? 160 : 128;
  const int rows = p.M - p.m_begin;
  const int tiles = ((rows + HALO_BM - 1) / HALO_BM) * ((p.N + bn - 1) / bn);
  if (!splitk) {
    if (n320) hipLaunchKernelGGL((conv_halo_kernel<5, false>), dim3(tiles), dim3(512), 0, s, p);
    else hipLaunchKernelGGL((conv_halo_kernel<4, false>), dim3(tiles), dim3(512), 0, s, p);
    return 0;
  }
  int ks = p.ksplit;
  if (ks <= 0) ks = lvd_splitk_plan(tiles, p.K, 256, HALO_BM * bn * 450 / 65536, nullptr);
  ks = min(ks, p.cin / 32);
  const long need = (long)ks * rows * p.N * 4;
  if (ks < 2 || p.act != LVD_ACT_NONE || !p.ws || p.ws_bytes < need) return -1;
  p.ksplit = ks;
  if (n320) hipLaunchKernelGGL((conv_halo_kernel<5, true>), dim3(tiles * ks), dim3(512), 0, s, p);
  else hipLaunchKernelGGL((conv_halo_kernel<4, true>), dim3(tiles * ks), dim3(512), 0, s, p);
  lvd_splitk_reduce_launch(&p, s);
  return 0;
}
