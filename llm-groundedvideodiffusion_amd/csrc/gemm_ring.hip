// gemm_ring.hip — LDS-DMA ring-buffered MFMA GEMM (second generation of gemm.hip, same lvd_gemm_params contract).
//
// PMC counters on the first kernel (register-staged, one K tile of prefetch, vmcnt(0)+barrier per tile) showed
// MFMA busy 17-34 %, waves parked in s_waitcnt/s_barrier ~40 % and zero LDS bank conflicts: the loads of tile k+1
// are issued one tile (~0.3 us of MFMA work) before they are needed, less than the loaded HBM/L2 latency.
// This kernel keeps STAGES-1 K tiles in flight:
//   * both operands go global -> LDS by global_load_lds (16 B per lane, no staging VGPRs, no ds_write issue slots);
//     conv padding and M/N/K tails are redirected to a 16-byte zero page instead of being predicated;
//   * counted s_waitcnt vmcnt(N) + raw s_barrier: only the tile about to be consumed is waited for, the
//     STAGES-2 younger tiles stay in flight across the barrier (hipcc's __syncthreads would drain them);
//   * one barrier per K tile; the slot that was consumed in iteration k-1 is refilled right after the barrier.
//   * what counted waits can and cannot do here: the compiler still puts vmcnt(0) in front of the first LDS read of an
//     iteration (it cannot prove the in-flight DMA does not alias the stage being read), so in practice one tile of
//     prefetch survives the barrier; issuing the DMA as inline asm removes that wait and changes nothing measurable —
//     the L2->LDS path is throughput-bound at ~10-12 TB/s.
// Geometry is a template: WMxWN waves, each owning an (FM*32)x(FN*32) accumulator tile.
//   128x128 (2x2 waves, 2x2 frags)  — small/odd shapes, most workgroups per CU; also the K-split slices
//   128x320 (2x2 waves, 2x5 frags)  — two workgroups per CU, N = 320·k exactly
//   256x160 / 256x128 (4x1 waves)   — 0.7 LDS fragment reads per MFMA instead of 1.0
//   256x320 / 256x256 (4x2 waves)   — 8 waves in two ping-pong groups, 142 / 128 flop per staged byte; 256x256 for GEGLU
//                                     (hidden/gate pairs need an even fragment count) and N not a multiple of 320
// Epilogue (gemm_tile.h): accumulators start from bias + temb row-bias; alpha / GEGLU in registers; a wave-private LDS
// strip turns the lane-owns-a-row layout into full 128-byte lines; residual operands are prefetched per 32-row block.
#include <cstdlib>
#include "common.h"

#include "gemm_tile.h"

namespace {

// Developer ablations of the ping-pong main loops (tools/build_ablations.sh builds side libraries with -DLVD_ABL=1|2; never defined in
// the shipped library): 1 = MFMAs compiled out (fragments kept live), 2 = in-loop DMA compiled out (the prologue tiles are re-read).
#ifndef LVD_ABL
#define LVD_ABL 0
#endif
template <class T>
LVD_DEV void keep_live(const T& v) { asm volatile("" ::"v"(v)); }
// -DLVD_TRACE (tools/build_ablations.sh, never in the shipped library): waves 0 and 4 of one workgroup stamp s_memtime at the phase
// edges of a few K tiles into a spare LDS strip and dump it to p.ws at the end (tools/phase_trace.py reads it back).
#ifdef LVD_TRACE
#define LVD_TRQ 128
#define LVD_STAMP()                                                         \
  do {                                                                      \
    if (tr_on && tr_n < 126) {                                              \
      const unsigned long tt_ = __builtin_amdgcn_s_memtime();               \
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                    \
      if (lane == 0) tr_buf[tr_n] = (unsigned)tt_;                          \
      ++tr_n;                                                               \
    }                                                                       \
  } while (0)
#else
#define LVD_TRQ 0
#define LVD_STAMP() do {} while (0)
#endif

// WM x WN waves (4 or 8); wave tile (FM*32) x (FN*32); STAGES-deep LDS ring
// ADMA (plain loader, K % RBK == 0): the LDS-DMA is issued as inline assembly from raw buffer descriptors (gemm_tile.h dma16),
// so the compiler neither sees a pending LDS write (no vmcnt(0) in front of every phase's first ds_read: the counted waits
// really leave STAGES-2 tiles in flight) nor keeps 64-bit source addresses; rows / columns past M / N are clamped (their
// products are never stored), K tiles past the end re-read the last one.
// LNF: the LayerNorm-folded product (lvd_gemm_params.ln_mean_rstd; gemm_tile.h LnEpi) — its own instantiation, not a runtime branch:
// one more epilogue path inside the 251-register FN = 5 kernels made the ordinary epilogue spill.
template <int MODE, int WM, int WN, int FM, int FN, int STAGES, int RBK, bool SPLITK = false, bool PP = false, bool ADMA = false, bool LNF = false>
__global__ __launch_bounds__(64 * WM * WN, (WM * WN == 8 ? 2 : 2)) void gemm_ring_kernel(const lvd_gemm_params p) {
  static_assert(!LNF || (ADMA && !SPLITK), "LayerNorm fold: asm-DMA, unsplit");
  static_assert(!ADMA || (MODE == LVD_A_PLAIN && (RBK == 32 || RBK == 64)), "asm DMA: plain loader, 32- or 64-wide K tiles");
  constexpr bool PP64 = PP && RBK == 64;                 // two-slot ring of 128-byte rows, consumed in two 32-deep halves (below)
  static_assert(!PP64 || (ADMA && STAGES == 2), "PP64: asm DMA, two slots");
  constexpr int NW = WM * WN;
  constexpr int RCH = RBK / 8;                            // 16-byte chunks per tile row (4: 64 B rows, 8: full 128 B lines)
  constexpr int RPI = 64 / RCH;                          // tile rows covered by one wave-wide glds instruction
  constexpr int BM = WM * FM * 32, BN = WN * FN * 32;
  constexpr int TILE = (BM + BN) * RCH;                  // uint4 per stage
  constexpr int AINS = BM / RPI, BINS = BN / RPI;        // wave-instructions per operand tile
  constexpr int APW = AINS / NW;                         // A instructions per wave (BM is a multiple of 16*NW)
  constexpr int BPW = (BINS + NW - 1) / NW;              // B instructions per wave (padded: every wave issues BPW)
  static_assert(AINS % NW == 0, "BM must be a multiple of RPI * waves");
  constexpr int LPS = APW + BPW;                         // glds per wave per stage (uniform -> one vmcnt immediate)
  constexpr int BIASQ = ADMA ? (BN * 4 + 1023) / 1024 * 64 : 0;               // ADMA: the tile's bias row is staged in LDS by the DMA engine too
  constexpr int LNQ = LNF ? BIASQ : 0;                                         // ... and, for a LayerNorm-folded product, its colsum row
  __shared__ uint4 lds[STAGES * TILE + BIASQ + LNQ + (PP ? LVD_TRQ : 0)];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave - wm * WN;
  const int l31 = lane & 31, hi = lane >> 5;
#ifdef LVD_TRACE
  unsigned* tr_buf = reinterpret_cast<unsigned*>(lds + STAGES * TILE + BIASQ + LNQ) + (wave >> 2) * 128;
  int tr_n = 0;
  const bool tr_blk = PP && (int)blockIdx.x == (int)(gridDim.x / 2) && (wave & 3) == 0;
  bool tr_on = false;
  const unsigned long tr_t0 = __builtin_amdgcn_s_memtime();
#endif

  const int nb = gridDim.x;
  int id;
  {
    int bid = blockIdx.x;
    int q = nb >> 3, r = nb & 7;
    int xcd = bid & 7, idx = bid >> 3;
    id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tiles_n = (p.N + BN - 1) / BN;
  // split-K: the K slice index is the slowest-varying part of the (XCD-remapped) tile id
  const int ntile = SPLITK ? nb / p.ksplit : nb;
  const int slice = SPLITK ? id / ntile : 0;
  if (SPLITK) id -= slice * ntile;
  const int tm = id / tiles_n, tn = id - tm * tiles_n;
  int kbeg = 0, klim = p.K;
  if (SPLITK) {
    int kchunk = (((p.K + p.ksplit - 1) / p.ksplit + RBK - 1) / RBK) * RBK;
    kbeg = slice * kchunk;
    klim = min(p.K, kbeg + kchunk);
  }

  // LDS image is lane-linear per instruction (row = rsub, position = cpos); the bank-conflict swizzle is applied to the
  // SOURCE chunk: RCH=4: pos ^ ((row>>2)&3), RCH=8: pos ^ ((row>>1)&7)  (row parity of 8-row instructions enters via bit 2)
  const int cpos = lane % RCH, rsub = lane / RCH;
  auto swz = [](int row, int c) { return RCH == 4 ? (c ^ ((row >> 2) & 3)) : (c ^ ((row >> 1) & 7)); };

  RowInfo ar[APW];
  long woff[BPW];
  bool wvalid[BPW];
  int bins[BPW];
  [[maybe_unused]] int av1[APW], av2[APW], bv[BPW];
  [[maybe_unused]] v4i rsA1, rsA2, rsB;
  if constexpr (ADMA) {
    rsA1 = make_rsrc(p.a1);
    rsA2 = make_rsrc(p.a2);
    rsB = make_rsrc(p.w);
#pragma unroll
    for (int q = 0; q < APW; ++q) {
      const int r = (wave * APW + q) * RPI + rsub;
      const int m = min(p.m_begin + tm * BM + r, p.M - 1);
      av1[q] = m * p.lda1 * 2 + swz(r, cpos) * 16;
      av2[q] = m * p.lda2 * 2 + swz(r, cpos) * 16;
    }
#pragma unroll
    for (int t = 0; t < BPW; ++t) {
      const int b = wave + NW * t;
      bins[t] = b < BINS ? b : BINS - 1;
      const int r = bins[t] * RPI + rsub;
      bv[t] = min(tn * BN + r, p.N - 1) * p.K * 2 + swz(r, cpos) * 16;
    }
  } else {
#pragma unroll
    for (int q = 0; q < APW; ++q) ar[q] = make_row<MODE>(p, p.m_begin + tm * BM + (wave * APW + q) * RPI + rsub, true);
#pragma unroll
    for (int t = 0; t < BPW; ++t) {
      int b = wave + NW * t;
      bool real = b < BINS;
      bins[t] = real ? b : BINS - 1;  // padding instruction re-stages the last 16 rows (same data, harmless)
      int n = tn * BN + bins[t] * RPI + rsub;
      wvalid[t] = n < p.N;
      woff[t] = (long)n * p.K;
    }
  }

  // one wave-wide LDS-DMA instruction of tile kt (idx < APW: A rows, else B rows)
  auto stage_one = [&](int kt, int slot, int idx) {
    uint4* A = lds + slot * TILE;
    uint4* B = A + BM * RCH;
    if constexpr (ADMA) {
      const int kb = min(kbeg + kt * RBK, p.K - RBK);  // tiles past the end (uniform vmcnt bookkeeping) re-read the last one
      if (idx < APW) {
        const bool s2 = kb >= p.c1;                    // a K tile lies in one source (c1 % RBK == 0)
        dma16(s2 ? rsA2 : rsA1, s2 ? av2[idx] : av1[idx], (s2 ? kb - p.c1 : kb) * 2, lds_addr(A + (wave * APW + idx) * RPI * RCH));
      } else {
        dma16(rsB, bv[idx - APW], kb * 2, lds_addr(B + bins[idx - APW] * RPI * RCH));
      }
    } else if (idx < APW) {
      const int q = idx;
      const int k0 = kbeg + kt * RBK + swz((wave * APW + q) * RPI + rsub, cpos) * 8;
      const lvd_bf16* src = a_src<MODE>(p, ar[q], k0, klim);
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(A + (wave * APW + q) * RPI * RCH), 16, 0, 0);
    } else {
      const int t = idx - APW;
      const int k0 = kbeg + kt * RBK + swz(bins[t] * RPI + rsub, cpos) * 8;
      const lvd_bf16* src = (wvalid[t] && k0 < klim) ? p.w + woff[t] + k0 : reinterpret_cast<const lvd_bf16*>(g_zero_page);
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(B + bins[t] * RPI * RCH), 16, 0, 0);
    }
  };
  auto stage = [&](int kt, int slot) {
#pragma unroll
    for (int idx = 0; idx < LPS; ++idx) stage_one(kt, slot, idx);
  };

  const int nk = (max(klim - kbeg, 0) + RBK - 1) / RBK;

  if constexpr (ADMA && !SPLITK) {
    if (p.bias && wave * 256 < BN) {
      const int off = min(tn * BN * 4 + wave * 1024 + lane * 16, p.N * 4 - 16);
      dma16(make_rsrc(p.bias), off, 0, lds_addr(lds + STAGES * TILE + wave * 64));
    }
    if constexpr (LNF) {
      if (wave * 256 < BN) {
        const int off = min(tn * BN * 4 + wave * 1024 + lane * 16, p.N * 4 - 16);
        dma16(make_rsrc(p.ln_colsum), off, 0, lds_addr(lds + STAGES * TILE + BIASQ + wave * 64));
      }
    }
  }
  // prologue: STAGES-1 tiles in flight (tiles beyond nk are staged from the zero page: uniform vmcnt bookkeeping)
#pragma unroll
  for (int s = 0; s < STAGES - 1; ++s) stage(s, s);
  // accumulators start from bias + temb row-bias (the K-split slices start from zero: their reduce kernel adds them); the
  // loads are issued behind the DMA prologue so both latencies overlap.  ADMA: a burst of ordinary loads between the asm
  // statements cannot be scheduled around them and spilled ~80 registers; the bias row of the tile (BN floats) rides the DMA
  // path instead — the oldest DMA of waves 0-1, so every counted wait below covers it — and is read from LDS after the
  // first barrier.
  f32x16 acc[FM][FN];
  constexpr bool LDS_BIAS = ADMA && !SPLITK;
  if (SPLITK || LDS_BIAS) {
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  } else {
    ring_bias_init<FM, FN>(p, acc, p.m_begin + tm * BM + wm * FM * 32, tn * BN + wn * FN * 32, l31, hi);
  }

  auto acc_from_lds_bias = [&]() {  // after a barrier that follows the waves' wait for the bias DMA
    if (p.bias && !LNF) {  // (a LayerNorm-folded product adds its bias behind the row scaling: ring_epilogue_rows)
      const uint4* bq = lds + STAGES * TILE + wn * FN * 8 + hi;
#pragma unroll
      for (int j = 0; j < FN; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 v = __builtin_bit_cast(f32x4, bq[j * 8 + 2 * q]);
#pragma unroll
          for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i][j][4 * q + e] = v[e];
        }
    }
  };

  if constexpr (PP64) {
    // 64-deep K tiles, two LDS slots (RBK = 64).  Every DMA instruction moves 8 rows x one full 128-byte line instead of 16 rows
    // x half a line: half the lines per staged byte through the address unit and the L1, and no line is fetched twice by
    // consecutive K tiles.  The registers hold the fragments of 32 of the 64 columns, so a tile is consumed as two halves, each a
    // ping-pong phase pair as in the 32-deep schedule further down (waves 4-7 = group 1 one phase behind waves 0-3 = group 0):
    //   phase:      4j        4j+1      4j+2      4j+3
    //   group 0:  L(j,h0)   M(j,h0)   L(j,h1)   M(j,h1)
    //   group 1:  M(j-1,h1) L(j,h0)   M(j,h0)   L(j,h1)
    // A slot is free once group 1 has read its second half AND HAS PASSED THE BARRIER BEHIND THAT PHASE (end of phase 4j+3): the four
    // waves of a group are not in step inside a phase, so a wave that refilled the slot right behind its own last reads would overwrite
    // rows a sibling wave is still reading (rare, and only when something else on the CU delays that sibling: a first version did
    // exactly that and produced a few thousand wrong elements per launch next to a co-tenant process — tests/probes/gemm_cotenant_probe.py).
    // The slot has to hold tile j+2 at phase 4j+8.  A wave's share of a tile (LPS instructions):
    //   group 0: batch 0 of tile j+1 in L(j,h0), batch 1 in L(j,h1)     -> waits vmcnt(0) behind the MFMAs of M(j,h1)
    //   group 1: its whole share of tile j+1 in L(j,h0) (phase 4j+1: both groups left that slot two barriers ago); nothing in L(j,h1),
    //            where a refill would have no phase left to land in    -> waits vmcnt(0) at the end of L(j,h1)
    // RAW: all of tile j+1 is waited for by the end of phase 4j+3, one barrier before its first reader.  WAR: every batch is issued
    // at least one barrier after the last read, by any wave, of the slot it overwrites.
    constexpr int NB0 = (LPS + 1) / 2;
    const int group = wave >> 2;
    auto batch0 = [&](int kt, int slot) {
      if (LVD_ABL == 2 && kt > 1) return;
#pragma unroll
      for (int idx = 0; idx < NB0; ++idx) stage_one(kt, slot, idx);
    };
    auto batch1 = [&](int kt, int slot) {
      if (LVD_ABL == 2 && kt > 1) return;
#pragma unroll
      for (int idx = NB0; idx < LPS; ++idx) stage_one(kt, slot, idx);
    };
    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    if constexpr (LDS_BIAS) acc_from_lds_bias();
    if (group == 1) __builtin_amdgcn_s_barrier();
    for (int j = 0; j < nk; ++j) {
      const int slot = j & 1;
      const uint4* A = lds + slot * TILE;
      const uint4* B = A + BM * RCH;
#ifdef LVD_TRACE
      tr_on = tr_blk && j >= 4 && j < 8;
#endif
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        LVD_STAMP();
        bf16x8 af[2][FM], bfr[2][FN];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const int c = (h * 2 + ks) * 2 + hi;
#pragma unroll
          for (int i = 0; i < FM; ++i) {
            int row = (wm * FM + i) * 32 + l31;
            af[ks][i] = as_bf16x8(A[row * RCH + swz(row, c)]);
          }
#pragma unroll
          for (int jn = 0; jn < FN; ++jn) {
            int row = (wn * FN + jn) * 32 + l31;
            bfr[ks][jn] = as_bf16x8(B[row * RCH + swz(row, c)]);
          }
        }
        if (h == 0) {
          if (j + 1 < nk) {
            batch0(j + 1, slot ^ 1);
            if (group == 1) batch1(j + 1, slot ^ 1);
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if (group == 0) {
          if (j + 1 < nk) batch1(j + 1, slot ^ 1);
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else {
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          wait_vmcnt<0>();
        }
        LVD_STAMP();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        LVD_STAMP();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int jn = 0; jn < FN; ++jn) {
              if (LVD_ABL == 1) { keep_live(bfr[ks][jn]); keep_live(af[ks][i]); continue; }
              acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[ks][jn], af[ks][i], acc[i][jn], 0, 0, 0);
            }
        __builtin_amdgcn_s_setprio(0);
        if (h == 1 && group == 0) wait_vmcnt<0>();
        LVD_STAMP();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (group == 0) __builtin_amdgcn_s_barrier();
  } else if constexpr (PP) {
    // Ping-pong schedule for the 8-wave geometries (two waves per SIMD).  In the lock-step loop below both waves of a
    // SIMD read their fragments at the same time and then compete for the MFMA pipe at the same time.  Here the K tile
    // is two phases — L: fragments LDS -> registers, refill DMA, waits;  M: nothing but MFMAs — and waves 4-7 run one
    // phase behind waves 0-3 (one extra barrier in front, one behind), so every SIMD always has one wave in M.
    //   RAW: a wave waits for ITS share of tile kt+1 at the end of L(kt); both groups have done so before the barrier
    //        that precedes the first L(kt+1).
    //   WAR: the refill issued in L(kt) overwrites tile kt-1, last read in the other group's L(kt-1) one phase earlier
    //        and retired there by lgkmcnt(0) before the barrier.
    const int group = wave >> 2;
    wait_vmcnt<(STAGES - 2) * LPS>();
    __builtin_amdgcn_s_barrier();
    if constexpr (LDS_BIAS) acc_from_lds_bias();
    if (group == 1) __builtin_amdgcn_s_barrier();
    int slot = 0;
    for (int kt = 0; kt < nk; ++kt) {
      const int nslot = slot == 0 ? STAGES - 1 : slot - 1;
      const uint4* A = lds + slot * TILE;
      const uint4* B = A + BM * RCH;
#ifdef LVD_TRACE
      tr_on = tr_blk && kt >= 8 && kt < 16;
#endif
      LVD_STAMP();
      bf16x8 af[RBK / 16][FM], bfr[RBK / 16][FN];
#pragma unroll
      for (int ks = 0; ks < RBK / 16; ++ks) {
        const int c = ks * 2 + hi;
#pragma unroll
        for (int i = 0; i < FM; ++i) {
          int row = (wm * FM + i) * 32 + l31;
          af[ks][i] = as_bf16x8(A[row * RCH + swz(row, c)]);
        }
#pragma unroll
        for (int j = 0; j < FN; ++j) {
          int row = (wn * FN + j) * 32 + l31;
          bfr[ks][j] = as_bf16x8(B[row * RCH + swz(row, c)]);
        }
      }
      if (!(LVD_ABL == 2 && kt > 0)) stage(kt + STAGES - 1, nslot);
      wait_vmcnt<(STAGES - 2) * LPS>();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      LVD_STAMP();
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      LVD_STAMP();
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int ks = 0; ks < RBK / 16; ++ks)
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
          for (int j = 0; j < FN; ++j) {
            if (LVD_ABL == 1) { keep_live(bfr[ks][j]); keep_live(af[ks][i]); continue; }
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[ks][j], af[ks][i], acc[i][j], 0, 0, 0);
          }
      __builtin_amdgcn_s_setprio(0);
      LVD_STAMP();
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      slot = slot + 1 == STAGES ? 0 : slot + 1;
    }
    if (group == 0) __builtin_amdgcn_s_barrier();
  } else {
    if constexpr (LDS_BIAS) {  // the bias DMA is older than the STAGES-1 prologue tiles
      wait_vmcnt<(STAGES - 1) * LPS>();
      __builtin_amdgcn_s_barrier();
      acc_from_lds_bias();
    }
    int slot = 0;
    for (int kt = 0; kt < nk; ++kt) {
      // tile kt has landed once at most (STAGES-2) younger stages are still outstanding
      wait_vmcnt<(STAGES - 2) * LPS>();
      __builtin_amdgcn_s_barrier();
      // refill the slot consumed in iteration kt-1 with tile kt+STAGES-1, one DMA instruction every IVL MFMAs: the
      // address arithmetic hides in the MFMA shadow and the L2 sees a steady request stream instead of a burst per barrier.
      const int nslot = slot == 0 ? STAGES - 1 : slot - 1;
      constexpr int NMF = (RBK / 16) * FM * FN;            // MFMAs per wave per K tile
      constexpr int IVL = NMF / LPS > 0 ? NMF / LPS : 1;   // one DMA instruction every IVL MFMAs: a steady request stream
      const uint4* A = lds + slot * TILE;
      const uint4* B = A + BM * RCH;
  #pragma unroll
      for (int ks = 0; ks < RBK / 16; ++ks) {
        bf16x8 af[FM], bfr[FN];
        const int c = ks * 2 + hi;
  #pragma unroll
        for (int i = 0; i < FM; ++i) {
          int row = (wm * FM + i) * 32 + l31;
          af[i] = as_bf16x8(A[row * RCH + swz(row, c)]);
        }
  #pragma unroll
        for (int j = 0; j < FN; ++j) {
          int row = (wn * FN + j) * 32 + l31;
          bfr[j] = as_bf16x8(B[row * RCH + swz(row, c)]);
        }
  #pragma unroll
        for (int i = 0; i < FM; ++i)
  #pragma unroll
          for (int j = 0; j < FN; ++j) {
            const int cnt = (ks * FM + i) * FN + j;
            if (cnt % IVL == 0 && cnt / IVL < LPS) stage_one(kt + STAGES - 1, nslot, cnt / IVL);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);  // D[n][m]: lane = token row
          }
      }
      if (LPS > NMF) {
  #pragma unroll
        for (int idx = NMF; idx < LPS; ++idx) stage_one(kt + STAGES - 1, nslot, idx);
      }
      slot = slot + 1 == STAGES ? 0 : slot + 1;
    }
  }
  // LayerNorm-folded product: (mean, rstd) of this lane's FM rows.  Loaded here, behind the main loop (the fragment registers are dead
  // now; four more live registers across the loop made the FN = 5 tiles spill), in front of the final wait + barrier that cover them.
  [[maybe_unused]] LnEpi<FM> lnepi;
  if constexpr (LNF) {
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      const int m = min(p.m_begin + tm * BM + (wm * FM + i) * 32 + l31, p.M - 1);
      const float2 mr = *reinterpret_cast<const float2*>(p.ln_mean_rstd + 2L * m);
      lnepi.mean[i] = mr.x; lnepi.rstd[i] = mr.y;
    }
  }
  wait_vmcnt<0>();
#ifdef LVD_TRACE
  if constexpr (PP) {
    if (tr_blk && p.ws) {
      const unsigned long tr_t1 = __builtin_amdgcn_s_memtime();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      unsigned* dst = reinterpret_cast<unsigned*>(p.ws) + (wave >> 2) * 128;
      if (lane == 0) { tr_buf[126] = (unsigned)(tr_t1 - tr_t0); tr_buf[127] = tr_n; }
      __builtin_amdgcn_wave_barrier();
      dst[lane] = tr_buf[lane];
      dst[lane + 64] = tr_buf[lane + 64];
    }
  }
#endif

  const int mbase = p.m_begin + tm * BM + wm * FM * 32;
  const int nbase = tn * BN + wn * FN * 32;
  if (SPLITK) {  // raw fp32 partial sums into this slice's slab; bias/residual/... happen in splitk_reduce_kernel
    float* slab = p.ws + ((long)slice * (p.M - p.m_begin) - p.m_begin) * p.N;  // slabs hold rows [m_begin, M)
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
  // the ring is idle now: every wave transposes its accumulators through its own share of it
  constexpr int WAVE_DW = STAGES * TILE * 4 / NW;
  static_assert(WAVE_DW >= 32 * (FN * 16 + 2), "ring too small for the epilogue strip");
  __builtin_amdgcn_s_barrier();  // all waves are done reading the last K tile (and every DMA has landed: vmcnt(0) above)
  if constexpr (LNF) {
    lnepi.b = reinterpret_cast<const float*>(lds + STAGES * TILE) + wn * FN * 32;
    lnepi.s = reinterpret_cast<const float*>(lds + STAGES * TILE + BIASQ) + wn * FN * 32;
    if (p.act == LVD_ACT_GEGLU) {
      if constexpr (FN % 2 == 0) ring_epilogue_rows<FM, FN, true, RowsLinear, true>(p, acc, RowsLinear{mbase}, nbase, lane, reinterpret_cast<uint32_t*>(lds) + wave * WAVE_DW, &lnepi);
    } else {
      ring_epilogue_rows<FM, FN, false, RowsLinear, true>(p, acc, RowsLinear{mbase}, nbase, lane, reinterpret_cast<uint32_t*>(lds) + wave * WAVE_DW, &lnepi);
    }
  } else {
    ring_epilogue_auto<FM, FN>(p, acc, mbase, nbase, lane, reinterpret_cast<uint32_t*>(lds) + wave * WAVE_DW);
  }
}

// deterministic slab reduction + the usual epilogue (bias, temb row-bias, gate, residual, accumulate)
__global__ void splitk_reduce_kernel(const lvd_gemm_params p) {
  const int rows = p.M - p.m_begin;
  const long quads = (long)rows * (p.N >> 2);
  const long slab = (long)rows * p.N;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < quads; i += (long)gridDim.x * blockDim.x) {
    const int lm = (int)(i / (p.N >> 2));
    const int n = (int)(i - (long)lm * (p.N >> 2)) * 4;
    const int m = p.m_begin + lm;
    splitk_reduce_quad(p, p.ws + (long)lm * p.N + n, slab, m, n);
  }
}

}  // namespace

// slab reduction of a K-split product whose slabs are in p->ws (p->ksplit slices); shared with conv_halo.hip
void lvd_splitk_reduce_launch(const lvd_gemm_params* p, void* stream) {
  const long quads = (long)(p->M - p->m_begin) * (p->N / 4);
  int rb = (int)((quads + 255) / 256);
  if (rb > 4096) rb = 4096;
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3(rb), dim3(256), 0, (hipStream_t)stream, *p);
}

namespace {

// K split over workgroups: SLOTS = workgroups resident on the device for this geometry (one full round, never a second
// partial one).  The wide ping-pong geometries stage 2.2x fewer bytes per flop than 128x128 and are what the small-M
// deep-level layers (M = 1080 ... 8640, K up to 23040) need once the K split gives them enough workgroups.
template <int WM, int WN, int FM, int FN, int STAGES, bool PP, int SLOTS, bool ADMA = false, int RBK = 32>
int launch_splitk(const lvd_gemm_params* pp, hipStream_t s) {
  constexpr int BM = WM * FM * 32, BN = WN * FN * 32;
  lvd_gemm_params p = *pp;
  const int rows = p.M - p.m_begin;
  const int tiles = ((rows + BM - 1) / BM) * ((p.N + BN - 1) / BN);
  int ks = p.ksplit;
  if (ks <= 0) ks = lvd_splitk_plan(tiles, p.K, SLOTS, BM * BN * 450 / 65536 * (SLOTS == 256 ? 1 : 0) + (SLOTS == 256 ? 0 : 160), nullptr);
  long need = (long)ks * rows * p.N * 4;
  if (ks < 2 || p.act != LVD_ACT_NONE || !p.ws || p.ws_bytes < need) return -1;  // caller falls back to the unsplit ring
  p.ksplit = ks;
  dim3 grid(tiles * ks), block(64 * WM * WN);
  switch (p.mode) {
    case LVD_A_PLAIN: hipLaunchKernelGGL((gemm_ring_kernel<LVD_A_PLAIN, WM, WN, FM, FN, STAGES, RBK, true, PP, ADMA>), grid, block, 0, s, p); break;
    default:
      if constexpr (RBK == 32) {
        switch (p.mode) {
          case LVD_A_CONV3X3: hipLaunchKernelGGL((gemm_ring_kernel<LVD_A_CONV3X3, WM, WN, FM, FN, STAGES, 32, true, PP>), grid, block, 0, s, p); break;
          case LVD_A_TCONV3: hipLaunchKernelGGL((gemm_ring_kernel<LVD_A_TCONV3, WM, WN, FM, FN, STAGES, 32, true, PP>), grid, block, 0, s, p); break;
          case LVD_A_CONV3X3_T2: hipLaunchKernelGGL((gemm_ring_kernel<LVD_A_CONV3X3_T2, WM, WN, FM, FN, STAGES, 32, true, PP>), grid, block, 0, s, p); break;
          default: return 1;
        }
      } else {
        return 1;
      }
  }
  lvd_splitk_reduce_launch(&p, s);
  return 0;
}

template <int WM, int WN, int FM, int FN, int STAGES, int RBK = 32, bool PP = false, bool ADMA = false>
int launch_ring(const lvd_gemm_params* p, hipStream_t s) {
  constexpr int BM = WM * FM * 32, BN = WN * FN * 32;
  int tiles = ((p->M - p->m_begin + BM - 1) / BM) * ((p->N + BN - 1) / BN);
  dim3 grid(tiles), block(64 * WM * WN);
  if constexpr (ADMA) {
    if (p->ln_mean_rstd) {  // LayerNorm-folded product (GEGLU needs an even fragment count: the FN = 5 tiles do not take it)
      if (p->mode != LVD_A_PLAIN || (p->act == LVD_ACT_GEGLU && FN % 2)) return 4;
      hipLaunchKernelGGL((gemm_ring_kernel<LVD_A_PLAIN, WM, WN, FM, FN, STAGES, RBK, false, PP, true, true>), grid, block, 0, s, *p);
      return 0;
    }
  } else {
    if (p->ln_mean_rstd) return 4;
  }
  switch (p->mode) {
    case LVD_A_PLAIN: hipLaunchKernelGGL((gemm_ring_kernel<LVD_A_PLAIN, WM, WN, FM, FN, STAGES, RBK, false, PP, ADMA>), grid, block, 0, s, *p); break;
    default:
      if constexpr (!ADMA) {  // the asm-DMA instantiations exist for the plain loader only
        switch (p->mode) {
          case LVD_A_CONV3X3: hipLaunchKernelGGL((gemm_ring_kernel<LVD_A_CONV3X3, WM, WN, FM, FN, STAGES, RBK, false, PP>), grid, block, 0, s, *p); break;
          case LVD_A_TCONV3: hipLaunchKernelGGL((gemm_ring_kernel<LVD_A_TCONV3, WM, WN, FM, FN, STAGES, RBK, false, PP>), grid, block, 0, s, *p); break;
          case LVD_A_CONV3X3_T2: hipLaunchKernelGGL((gemm_ring_kernel<LVD_A_CONV3X3_T2, WM, WN, FM, FN, STAGES, RBK, false, PP>), grid, block, 0, s, *p); break;
          default: return 1;
        }
      } else {
        return 1;
      }
  }
  return 0;
}

}  // namespace

// geometry: 0 = 128x128 (3 stages), 1 = 128x128 (4 stages), 2 = 256x160 (3 stages), 3 = 256x128 (3 stages),
//           4 = 256x320 8 waves (3 stages, ping-pong), 5 = 256x256 8 waves (3 stages, ping-pong),
//           8 = 256x256x64 8 waves (2 stages), 12 = 128x320 (2 stages), 20 = split-K 128x128
int lvd_gemm_ring_dispatch(const lvd_gemm_params* p, void* stream, int geometry) {
  hipStream_t s = (hipStream_t)stream;
  // +100: the asm-DMA instantiation of the same geometry (plain loader, K % 32 == 0, whole K tiles per source, no temb row-bias)
  const bool adma_ok = p->mode == LVD_A_PLAIN && p->K % 32 == 0 && p->rowbias == nullptr && p->N >= 4 &&
                       (p->a2 == nullptr || p->c1 % 32 == 0) &&
                       (long)p->M * (p->lda1 > p->lda2 ? p->lda1 : p->lda2) < (1L << 30) && (long)p->N * p->K < (1L << 30);
  if (geometry >= 200) {  // +200: 64-deep K tiles on the 8-wave geometries (two LDS slots, full-line DMA); anything else -> the +100 form
    geometry -= 200;
    const bool ok64 = adma_ok && p->K % 64 == 0 && p->K >= 128 && (p->a2 == nullptr || p->c1 % 64 == 0);
    if (ok64) {
      if (geometry == 24 || geometry == 25) {
        int rc = geometry == 24 ? launch_splitk<4, 2, 2, 5, 2, true, 256, true, 64>(p, s) : launch_splitk<4, 2, 2, 4, 2, true, 256, true, 64>(p, s);
        if (rc >= 0) return rc;
        geometry = geometry == 24 ? 4 : 5;
      }
      if (geometry == 4) return launch_ring<4, 2, 2, 5, 2, 64, true, true>(p, s);
      if (geometry == 5) return launch_ring<4, 2, 2, 4, 2, 64, true, true>(p, s);
      // round 4: the 4-wave geometries with 64-deep K tiles too (two LDS slots, one barrier per 64 columns instead of per 32): the deep
      // UNet levels run few, long K loops per CU, where the per-K-tile latency (barrier + counted wait + fragment reads) is the kernel
      if (geometry == 20) {
        int rc = launch_splitk<2, 2, 2, 2, 2, false, 512, true, 64>(p, s);
        if (rc >= 0) return rc;
        geometry = 0;
      }
      if (geometry == 0 || geometry == 1) return launch_ring<2, 2, 2, 2, 2, 64, false, true>(p, s);
      if (geometry == 2) return launch_ring<4, 1, 2, 5, 2, 64, false, true>(p, s);
      if (geometry == 3) return launch_ring<4, 1, 2, 4, 2, 64, false, true>(p, s);
      if (geometry == 12) return launch_ring<2, 2, 2, 5, 2, 64, false, true>(p, s);
    }
    geometry += 100;
  }
  if (geometry >= 100) {
    geometry -= 100;
    if (adma_ok) {
      if (geometry == 20) {
        int rc = launch_splitk<2, 2, 2, 2, 3, false, 768, true>(p, s);
        if (rc >= 0) return rc;
        geometry = 0;
      }
      if (geometry == 24 || geometry == 25) {
        int rc = geometry == 24 ? launch_splitk<4, 2, 2, 5, 3, true, 256, true>(p, s) : launch_splitk<4, 2, 2, 4, 3, true, 256, true>(p, s);
        if (rc >= 0) return rc;
        geometry = geometry == 24 ? 4 : 5;
      }
      switch (geometry) {
        case 0: return launch_ring<2, 2, 2, 2, 3, 32, false, true>(p, s);
        case 1: return launch_ring<2, 2, 2, 2, 4, 32, false, true>(p, s);
        case 2: return launch_ring<4, 1, 2, 5, 3, 32, false, true>(p, s);
        case 3: return launch_ring<4, 1, 2, 4, 3, 32, false, true>(p, s);
        case 4: return launch_ring<4, 2, 2, 5, 3, 32, true, true>(p, s);
        case 5: return launch_ring<4, 2, 2, 4, 3, 32, true, true>(p, s);
        case 12: return launch_ring<2, 2, 2, 5, 2, 32, false, true>(p, s);
        default: break;
      }
    }
  }
  if (geometry == 20) {
    int rc = launch_splitk<2, 2, 2, 2, 3, false, 768>(p, s);
    if (rc >= 0) return rc;
    geometry = 0;  // not splittable (no workspace / GEGLU / too little K): plain 128x128 ring
  }
  if (geometry == 24 || geometry == 25) {  // K split on the 8-wave ping-pong geometries (256x320 / 256x256), 1 workgroup per CU
    int rc = geometry == 24 ? launch_splitk<4, 2, 2, 5, 3, true, 256>(p, s) : launch_splitk<4, 2, 2, 4, 3, true, 256>(p, s);
    if (rc >= 0) return rc;
    geometry = geometry == 24 ? 4 : 5;
  }
  switch (geometry) {
    case 0: return launch_ring<2, 2, 2, 2, 3>(p, s);
    case 1: return launch_ring<2, 2, 2, 2, 4>(p, s);
    case 2: return launch_ring<4, 1, 2, 5, 3>(p, s);
    case 3: return launch_ring<4, 1, 2, 4, 3>(p, s);
    case 4: return launch_ring<4, 2, 2, 5, 3, 32, true>(p, s);
    case 5: return launch_ring<4, 2, 2, 4, 3, 32, true>(p, s);
    case 8: return launch_ring<4, 2, 2, 4, 2, 64>(p, s);
    case 12: return launch_ring<2, 2, 2, 5, 2>(p, s);
    default: return 1;
  }
}
