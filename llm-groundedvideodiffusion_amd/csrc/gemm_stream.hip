// gemm_stream.hip — persistent form of the 8-wave ping-pong asm-DMA ring GEMM (gemm_ring.hip, 256x320 / 256x256, 32-deep K tiles,
// three LDS slots): ONE workgroup per CU walks a list of output tiles and the ring of K tiles runs straight THROUGH the tile boundaries.
//
// Why (profiles/r05_gemm_roofline_model.txt): the short-K, large-M products of the two upper UNet levels (K = 320 / 640 at 138k / 35k token
// rows: to_qkv, to_q, to_out, the GEGLU feed-forward, proj_in / proj_out) ran at T_mfma + T_mem per tile, not max(T_mfma, T_mem).  A
// 256x320 tile of a K = 320 product is ten K tiles (about 10 us of matrix pipe), and with one workgroup per tile every tile also pays,
// in series and on every CU at the same time: workgroup dispatch, the first DMA round trip with nothing to multiply, the store phase
// (164 KB per CU at the ~15 B/clk a CU's stores drain) and the wait for the last write acknowledgement before the LDS can be handed to the next
// workgroup — with the matrix pipe idle through all of it (MFMAs compiled out, the round-3 ablation still took 111 of 134 us).
// Here the loader simply runs two K tiles ahead of the MFMAs in a flattened (item, k) iteration space, the stores of item i are never
// waited for (they drain under the K loop of item i+1; the counted vmcnt waits step over them), the accumulators are converted through
// LDS strips that are NOT part of the ring, and both wave groups convert in the same barrier interval.
//
// Round-1 history: a first persistent walker (gemm_pers.hip, removed) gained nothing — it used the compiler's LDS-DMA builtin (a hidden
// vmcnt(0) per K tile drained the prefetch AND the stores at every boundary) and the register-direct partial-line epilogue.
//
// Schedule at an item boundary (c = flat index of the item's last K tile; waves 4-7 = group 1 run one phase behind waves 0-3):
//   group 0:  ... M(c) | B2 | pre-issue DMA(c+3), vmcnt -> c+2 landed, EPILOGUE, acc <- bias' , L(c+1)      | B1 | M(c+1) ...
//   group 1:  ... L(c) | B1 | M(c), pre-issue DMA(c+3), vmcnt -> c+2 landed, EPILOGUE, acc <- bias'         | B2 | L(c+1) ...
// i.e. both epilogues sit between the same two barriers; no barrier is added or removed, so the ring protocol of the one-shot kernel
// (RAW: a wave waits for its share of tile t+1 before the barrier in front of L(t+1); WAR: a refill is issued at least one barrier
// after the last read of the slot by ANY wave) carries over.  DMA(c+3) goes into the slot tile c was read from: group 0 has passed
// the barrier behind group 1's L(c); group 1 has passed the barrier behind its own L(c) and group 0's L(c) is two barriers back.
// The wait in front of the epilogue (everything but the newest LPS pieces) retires the wave's share of tile c+2 while only loads are in
// the queue; the epilogue's stores are issued behind it and nothing waits for them until L(c+2), which needs tile c+3: the vmcnt
// there allows LPS + NSW operations (NSW = the stores of the last 32-row block — unconditional buffer stores, out-of-range lanes are
// dropped by the descriptor's bounds check — so the count is exact whatever the tile's raggedness).
//
// Tail: equal-cost tiles on G persistent workgroups take ceil(tiles / G) rounds.  When the last round holds at most G / 2 tiles, each of
// them is cut into two 128-row halves (every wave computes ONE 32-row block instead of two; same LDS image, the A rows of the unused blocks
// re-read their partner's lines) and the halves go to 2R workgroups: 540 tiles of 256x320 on 256 CUs cost 2.5 rounds instead of 3.
#include <cstdlib>
#include "common.h"

#include "gemm_tile.h"

namespace {

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

// -DLVD_TRACE (tools/build_ablations.sh, never in the shipped library): waves 0 and 4 of one workgroup stamp s_memtime at the phase edges
// of their second and third item into a spare LDS strip and dump it to p.ws at the end (tools/stream_trace.py reads it back).
#ifdef LVD_TRACE
#define LVD_TRQ 128
#define LVD_STAMP(tag)                                                      \
  do {                                                                      \
    if (tr_on && tr_n < 250) {                                              \
      const unsigned long tt_ = __builtin_amdgcn_s_memtime();               \
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                    \
      if ((threadIdx.x & 63) == 0) { tr_buf[tr_n] = (unsigned)tt_; tr_buf[tr_n + 1] = (tag); } \
      tr_n += 2;                                                            \
    }                                                                       \
  } while (0)
#else
#define LVD_TRQ 0
#define LVD_STAMP(tag) do {} while (0)
#endif

template <int FN, bool LNF, bool GEGLU, bool RES, int NSWK>
__global__ __launch_bounds__(512, 2) void gemm_stream_kernel(const lvd_gemm_params p, const int tiles, const int full_items, const int nitems) {
  constexpr int WN = 2, FM = 2, NW = 8, RBK = 32, RCH = 4, RPI = 16, STAGES = 3;
  constexpr int BM = 256, BN = WN * FN * 32;
  constexpr int TILE = (BM + BN) * RCH;                  // uint4 per ring slot
  constexpr int AINS = BM / RPI, BINS = BN / RPI;        // wave-instructions per operand tile
  constexpr int APW = AINS / NW, BPW = (BINS + NW - 1) / NW, LPS = APW + BPW;
  constexpr int BIASQ = (BN * 4 + 1023) / 1024 * 64;     // uint4 per staged bias / colsum row (whole 1 KB DMA pieces)
  constexpr int NSTRIP = LNF ? 2 : 1;
  constexpr int SETQ = NSTRIP * BIASQ;                   // one set = the rows of one item's n-tile; two sets (items alternate)
  constexpr int WO = GEGLU ? FN * 16 : FN * 32;          // output columns of a wave
  constexpr int CP = WO > 80 ? 2 : 1;                    // column passes of the strip epilogue
  constexpr int W = WO / CP;                             // output columns per pass (80 / 64)
  constexpr int S = W / 2 + 2;                           // dwords per strip row: S = 2 (mod 4) -> conflict-free ds_write_b64 of 32 rows
  constexpr int CPR = W / 8;                             // 16-byte chunks per strip row
  constexpr int PASSES = 32 * CPR / 64;                  // store instructions per (32-row block, column pass)
  constexpr int PP8 = W / 8;                             // 8-column groups per pass
  static_assert(!GEGLU || FN % 2 == 0, "GEGLU pairs hidden / gate fragments");
  static_assert((32 * CPR) % 64 == 0 && S % 4 == 2, "strip geometry");
  static_assert(!(LNF && RES) && !(GEGLU && RES), "no residual behind a LayerNorm-folded or GEGLU product");
  constexpr int STRIP_DW = 32 * S;
  constexpr int NSW = NSWK ? PASSES : 0;                 // stores known to sit between DMA(c+3) and DMA(c+4) in every wave's queue
  constexpr int RING_Q = STAGES * TILE;
  __shared__ uint4 lds[RING_Q + 2 * SETQ + (NW * STRIP_DW + 3) / 4 + LVD_TRQ];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, hi = lane >> 5;
  const int group = wave >> 2;
  auto swz = [](int row, int c) { return c ^ ((row >> 2) & 3); };
  // Lane-derived constants that only the once-per-item paths need (DMA row offsets, strip addresses, store offsets) must NOT be hoisted
  // out of the K-tile loop: at 250+ live registers they are spilled, and a scratch reload carries a compiler-inserted vmcnt(0) that
  // drains the hand-counted DMA queue.  Those paths start from an opaque copy of the lane id and recompute (a few VALU per item).
  auto opaque_lane = [&]() {
    int l = lane;
    asm volatile("" : "+v"(l));
    return l;
  };

  const int G = gridDim.x, bid = blockIdx.x;
#ifdef LVD_TRACE
  unsigned* tr_buf = reinterpret_cast<unsigned*>(lds + RING_Q + 2 * SETQ + (NW * STRIP_DW + 3) / 4) + (wave >> 2) * 256;
  int tr_n = 0;
  const bool tr_blk = (int)blockIdx.x == (int)(gridDim.x / 2) && (wave & 3) == 0;
  bool tr_on = false;
#endif
  const int tiles_n = (p.N + BN - 1) / BN;
  const int nk = p.K / RBK;
  const int my_items = (nitems - bid + G - 1) / G;  // >= 1 (G <= nitems)
  const int q8 = tiles >> 3, r8 = tiles & 7;
  // item j of this workgroup -> (m-tile, n-tile, half): half < 0 = the whole 256-row tile, 0 / 1 = its upper / lower 128 rows.  The tile
  // index goes through the XCD-aware bijection of the one-shot kernel: the G tiles in flight at any time are, per XCD, a contiguous run
  // of (m-tile, n-tile) pairs sharing their A rows (and the weights) in that XCD's L2.
  auto item_of = [&](int j, int& tm, int& tn, int& half) {
    const int it = bid + j * G;
    int t = it;
    half = -1;
    if (it >= full_items) {
      const int h = it - full_items;
      t = full_items + (h >> 1);
      half = h & 1;
    }
    const int xcd = t & 7, idx = t >> 3;
    const int id = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
    tm = id / tiles_n;
    tn = id - tm * tiles_n;
  };

  // ---- loader: (lj, lk) = the K tile issued next; two K tiles ahead of the MFMAs
  const v4i rsA = make_rsrc(p.a1), rsB = make_rsrc(p.w);
  int av[APW], bv[BPW], bins[BPW];
#pragma unroll
  for (int t = 0; t < BPW; ++t) {
    const int b = wave + NW * t;
    bins[t] = b < BINS ? b : BINS - 1;  // padding instruction re-stages the last 16 rows (same data, harmless)
  }
  int lj = 0, lk = 0, ltn = 0;
  auto set_loader_item = [&](int j) {
    int tm, tn, half;
    item_of(min(j, my_items - 1), tm, tn, half);  // past the end: the last item again (lands in a free slot, never read)
    ltn = tn;
    const int lo = opaque_lane();
    const int cpos = lo & 3, rsub = lo >> 2;
#pragma unroll
    for (int q = 0; q < APW; ++q) {
      const int r = (wave * APW + q) * RPI + rsub;
      const int rr = half < 0 ? r : half * 128 + (r >> 6) * 32 + (r & 31);
      const int m = min(tm * BM + rr, p.M - 1);
      av[q] = m * p.lda1 * 2 + swz(r, cpos) * 16;
    }
#pragma unroll
    for (int t = 0; t < BPW; ++t) {
      const int r = bins[t] * RPI + rsub;
      bv[t] = min(tn * BN + r, p.N - 1) * p.K * 2 + swz(r, cpos) * 16;
    }
  };
  auto stage = [&](int slot) {
    uint4* A = lds + slot * TILE;
    uint4* B = A + BM * RCH;
    // the bias (and colsum) row of an item's n-tile rides in front of its first K tile: older than that tile's pieces in the queue of
    // waves 0-1, so every counted wait that covers the tile covers it
    if (lk == 0 && lj < my_items && wave * 256 < BN) {
      const int off = min(ltn * BN * 4 + wave * 1024 + opaque_lane() * 16, p.N * 4 - 16);
      if (p.bias) dma16(make_rsrc(p.bias), off, 0, lds_addr(lds + RING_Q + (lj & 1) * SETQ + wave * 64));
      if constexpr (LNF) dma16(make_rsrc(p.ln_colsum), off, 0, lds_addr(lds + RING_Q + (lj & 1) * SETQ + BIASQ + wave * 64));
    }
    const int kb = lk * RBK * 2;
#pragma unroll
    for (int idx = 0; idx < APW; ++idx) dma16(rsA, av[idx], kb, lds_addr(A + (wave * APW + idx) * RPI * RCH));
#pragma unroll
    for (int t = 0; t < BPW; ++t) dma16(rsB, bv[t], kb, lds_addr(B + bins[t] * RPI * RCH));
    if (++lk == nk) {
      lk = 0;
      ++lj;
      set_loader_item(lj);
    }
  };

  f32x16 acc[FM][FN];
  auto acc_init = [&](int set) {  // accumulators start from the bias row (a LayerNorm-folded product adds it behind the row scaling)
    if (p.bias && !LNF) {
      const uint4* bq = lds + RING_Q + set * SETQ + wn * FN * 8 + (opaque_lane() >> 5);
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
    } else {
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    }
  };

  // ---- consumer state
  int ctm, ctn, chalf;
  item_of(0, ctm, ctn, chalf);
  const __amdgpu_buffer_rsrc_t rsOut = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, p.M * p.ldc * 2, 0x00020000);
  [[maybe_unused]] const __amdgpu_buffer_rsrc_t rsRes =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<lvd_bf16*>(RES ? p.res : reinterpret_cast<const lvd_bf16*>(p.out)), 0, RES ? p.M * p.ldres * 2 : 16, 0x00020000);
  uint32_t* const strip = reinterpret_cast<uint32_t*>(lds + RING_Q + 2 * SETQ) + wave * STRIP_DW;
  const int ncols = GEGLU ? (p.N >> 1) : p.N;

  // The epilogue of the consumer's item: rows mb + i*32 + (0..31) of this wave, FM (whole tile) or one (half tile) 32-row blocks, each in
  // CP column passes through the wave's strip.  mr = (mean, rstd) of the lane's rows (LayerNorm fold), loaded by the caller.
  typedef __attribute__((ext_vector_type(2))) float f32x2;
  auto epilogue = [&](int set, [[maybe_unused]] const f32x2 (&mr)[FM]) {
    const int lane = opaque_lane();
    const int l31 = lane & 31, hi = lane >> 5;
    const int mb = ctm * BM + (chalf < 0 ? wm * 64 : chalf * 128 + wm * 32);
    const int nbase = ctn * BN + wn * FN * 32;
    const int col0 = GEGLU ? (nbase >> 1) : nbase;
    [[maybe_unused]] const float* lnb = reinterpret_cast<const float*>(lds + RING_Q + set * SETQ) + wn * FN * 32;
    [[maybe_unused]] const float* lns = reinterpret_cast<const float*>(lds + RING_Q + set * SETQ + BIASQ) + wn * FN * 32;
    constexpr int NBLK = FM * CP;
    constexpr int UB = PP8 % 5 == 0 ? 5 : 4;  // 8-column groups converted per batch (their colsum / bias rows are read first, together)
    static_assert(PP8 % UB == 0, "whole batches");
    // residual rows of one (block, pass): issued RAHEAD blocks ahead (the accumulators a converted block releases are the registers the
    // next loads land in), always in front of an older block's stores, so that waiting for them never waits for a store (the queue is
    // in order); out-of-range lanes read the rejected offset (zeros, never used)
    constexpr int RAHEAD = 1;
    [[maybe_unused]] u32x4 rres[RAHEAD + 1][PASSES];
    auto res_issue = [&](int i, int cp, u32x4 (&dst)[PASSES]) {
#pragma unroll
      for (int ps = 0; ps < PASSES; ++ps) {
        const int idx = ps * 64 + lane;
        const int r = idx / CPR, c = idx - r * CPR;
        const int mm = mb + i * 32 + r;
        const int n = col0 + cp * W + c * 8;
        dst[ps] = __builtin_amdgcn_raw_buffer_load_b128(rsRes, (mm < p.M && n < ncols) ? (mm * p.ldres + n) * 2 : (int)0x80000000, 0, 0);
      }
    };
    if constexpr (RES) {
#pragma unroll
      for (int b = 0; b < RAHEAD && b < NBLK; ++b)
        if (!(b / CP > 0 && chalf >= 0)) res_issue(b / CP, b % CP, rres[b % (RAHEAD + 1)]);
    }
#pragma unroll
    for (int b = 0; b < NBLK; ++b) {
      const int i = b / CP, cp = b % CP;
      if (i > 0 && chalf >= 0) break;
      uint32_t* wrow = strip + l31 * S;
#pragma unroll
      for (int u0 = 0; u0 < PP8; u0 += UB) {
        [[maybe_unused]] f32x4 ls[UB][2], lb[UB][2];
        if constexpr (LNF) {
#pragma unroll
          for (int u = 0; u < UB; ++u) {
            const int c8 = cp * PP8 + u0 + u;
            const int nl = GEGLU ? (c8 >> 2) * 64 + 8 * (c8 & 3) + 4 * hi : (c8 >> 2) * 32 + 8 * (c8 & 3) + 4 * hi;
            ls[u][0] = *reinterpret_cast<const f32x4*>(lns + nl);
            lb[u][0] = *reinterpret_cast<const f32x4*>(lnb + nl);
            if constexpr (GEGLU) {
              ls[u][1] = *reinterpret_cast<const f32x4*>(lns + nl + 32);
              lb[u][1] = *reinterpret_cast<const f32x4*>(lnb + nl + 32);
            }
          }
        }
#pragma unroll
        for (int u = 0; u < UB; ++u) {
          const int c8 = cp * PP8 + u0 + u;  // 8-column group of the wave's output columns
          uint2 o;
          if constexpr (GEGLU) {
            const int bb = c8 >> 2, q = c8 & 3;  // hidden column block bb of the interleaved W'; gate = the fragment behind it
            f32x4 h, g;
#pragma unroll
            for (int e = 0; e < 4; ++e) { h[e] = acc[i][2 * bb][4 * q + e]; g[e] = acc[i][2 * bb + 1][4 * q + e]; }
            if constexpr (LNF) {
              h = ln_fold4(h, mr[i].x, mr[i].y, ls[u][0], lb[u][0]);
              g = ln_fold4(g, mr[i].x, mr[i].y, ls[u][1], lb[u][1]);
            }
            o = geglu4(h, g);
          } else {
            const int j = c8 >> 2, q = c8 & 3;
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * q + e];
            if constexpr (LNF) {
              v = ln_fold4(v, mr[i].x, mr[i].y, ls[u][0], lb[u][0]);
            }
            if (p.alpha != 1.f) v *= p.alpha;
            o.x = pack2bf(v[0], v[1]);
            o.y = pack2bf(v[2], v[3]);
          }
          *reinterpret_cast<uint2*>(wrow + (u0 + u) * 4 + 2 * hi) = o;
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      LVD_STAMP(13);
      if constexpr (RES) {
        if (b + RAHEAD < NBLK && !((b + RAHEAD) / CP > 0 && chalf >= 0)) res_issue((b + RAHEAD) / CP, (b + RAHEAD) % CP, rres[(b + RAHEAD) % (RAHEAD + 1)]);
      }
#pragma unroll
      for (int ps = 0; ps < PASSES; ++ps) {
        const int idx = ps * 64 + lane;
        const int r = idx / CPR, c = idx - r * CPR;
        const int mm = mb + i * 32 + r;
        const int n = col0 + cp * W + c * 8;
        // out-of-range rows / columns get an offset the descriptor's bounds check rejects: the store is issued and dropped
        const int ooff = (mm < p.M && n < ncols) ? (mm * p.ldc + n) * 2 : (int)0x80000000;
        const uint2 vlo = *reinterpret_cast<const uint2*>(strip + r * S + c * 4);
        const uint2 vhi = *reinterpret_cast<const uint2*>(strip + r * S + c * 4 + 2);
        u32x4 v = {vlo.x, vlo.y, vhi.x, vhi.y};
        if constexpr (RES) {  // the residual is added in fp32 to the bf16-rounded projection (what the reference's separate add does)
          const u32x4 t = rres[b % (RAHEAD + 1)][ps];
          v.x = pack2bf(bflo(v.x) + bflo(t.x), bfhi(v.x) + bfhi(t.x));
          v.y = pack2bf(bflo(v.y) + bflo(t.y), bfhi(v.y) + bfhi(t.y));
          v.z = pack2bf(bflo(v.z) + bflo(t.z), bfhi(v.z) + bfhi(t.z));
          v.w = pack2bf(bflo(v.w) + bflo(t.w), bfhi(v.w) + bfhi(t.w));
        }
        __builtin_amdgcn_raw_buffer_store_b128(v, rsOut, ooff, 0, 0);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      LVD_STAMP(14);
    }
  };

  // ---- prologue: two K tiles in flight
  set_loader_item(0);
  stage(0);
  stage(1);
  wait_vmcnt<LPS>();  // K tile 0 and the bias rows of item 0 (older) have landed
  __builtin_amdgcn_s_barrier();
  acc_init(0);
  if (group == 1) __builtin_amdgcn_s_barrier();

  int slot = 0, ck = 0, cj = 0;
  bool pre = false, post = false;
  const int total = my_items * nk;
  for (int c = 0; c < total; ++c) {
    const int nslot = slot == 0 ? STAGES - 1 : slot - 1;
    const uint4* A = lds + slot * TILE;
    const uint4* B = A + BM * RCH;
#ifdef LVD_TRACE
    tr_on = tr_blk && cj >= 1 && cj <= 2;
#endif
    LVD_STAMP(1);
    // ---- L: fragments LDS -> registers, refill DMA, counted wait
    bf16x8 af[RBK / 16][FM], bfr[RBK / 16][FN];
#pragma unroll
    for (int ks = 0; ks < RBK / 16; ++ks) {
      const int cc = ks * 2 + hi;
#pragma unroll
      for (int i = 0; i < FM; ++i) {
        const int row = (wm * FM + i) * 32 + l31;
        af[ks][i] = as_bf16x8(A[row * RCH + swz(row, cc)]);
      }
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const int row = (wn * FN + j) * 32 + l31;
        bfr[ks][j] = as_bf16x8(B[row * RCH + swz(row, cc)]);
      }
    }
    if (!pre) {
      stage(nslot);
      if (post) {
        wait_vmcnt<LPS + NSW>();
        post = false;
      } else {
        wait_vmcnt<LPS>();
      }
    }
    pre = false;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    LVD_STAMP(2);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    LVD_STAMP(3);
    // ---- M: nothing but MFMAs
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < RBK / 16; ++ks)
#pragma unroll
      for (int i = 0; i < FM; ++i) {
        if (i > 0 && chalf >= 0) continue;  // half tile: one 32-row block per wave
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[ks][j], af[ks][i], acc[i][j], 0, 0, 0);
      }
    __builtin_amdgcn_s_setprio(0);
    LVD_STAMP(4);
    const bool last = ck == nk - 1;
    auto boundary = [&]() {
      // (mean, rstd) of this lane's rows go out in FRONT of the pre-issued K tile, as asm loads the compiler does not count: the one
      // counted wait below retires them together with tile c+2 (a compiler-visible load would be waited for with vmcnt(0), i.e. behind
      // the pre-issued tile — and, for the second block, behind the first block's stores)
      [[maybe_unused]] f32x2 mr[FM];
      if constexpr (LNF) {
        const v4i rsLn = make_rsrc(p.ln_mean_rstd);
        const int mb = ctm * BM + (chalf < 0 ? wm * 64 : chalf * 128 + wm * 32) + (opaque_lane() & 31);
#pragma unroll
        for (int i = 0; i < FM; ++i)
          asm volatile("buffer_load_dwordx2 %0, %1, %2, 0 offen" : "=v"(mr[i]) : "v"(min(mb + i * 32, p.M - 1) * 8), "s"(rsLn) : "memory");
      }
      LVD_STAMP(10);
      stage(slot);  // K tile c+3 into the slot tile c was read from (see the header for the WAR argument)
      LVD_STAMP(11);
      if constexpr (LNF) {
        static_assert(FM == 2, "two row blocks");
        if constexpr (LPS == 5) asm volatile("s_waitcnt vmcnt(5)" : "+v"(mr[0]), "+v"(mr[1])::"memory");
        else asm volatile("s_waitcnt vmcnt(4)" : "+v"(mr[0]), "+v"(mr[1])::"memory");
        static_assert(LPS == 5 || LPS == 4, "vmcnt immediates above");
      } else {
        wait_vmcnt<LPS>();
      }
      LVD_STAMP(12);
      epilogue(cj & 1, mr);
      LVD_STAMP(19);
      if (cj + 1 < my_items) {
        item_of(cj + 1, ctm, ctn, chalf);
        acc_init((cj + 1) & 1);
      }
      LVD_STAMP(20);
    };
    if (last && group == 1) boundary();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (last && group == 0) boundary();
    LVD_STAMP(5);
    if (last) {
      pre = true;
      post = true;
      ck = 0;
      ++cj;
    } else {
      ++ck;
    }
    slot = slot + 1 == STAGES ? 0 : slot + 1;
  }
  if (group == 0) __builtin_amdgcn_s_barrier();
  wait_vmcnt<0>();
#ifdef LVD_TRACE
  if (tr_blk && p.ws) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    unsigned* dst = reinterpret_cast<unsigned*>(p.ws) + (wave >> 2) * 256;
    if (lane == 0) tr_buf[255] = tr_n;
    __builtin_amdgcn_wave_barrier();
    for (int q = 0; q < 4; ++q) dst[lane + 64 * q] = tr_buf[lane + 64 * q];
  }
#endif
}

int g_cus = 0;
int stream_cus() {
  if (!g_cus) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&g_cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (g_cus <= 0) g_cus = 256;
    g_cus &= ~7;  // multiple of 8: tile t and workgroup t % G sit on the same XCD
    if (g_cus < 8) g_cus = 8;
  }
  return g_cus;
}

template <int FN, bool LNF, bool GEGLU, bool RES>
int launch_stream(const lvd_gemm_params* p, hipStream_t s, int nsw) {
  constexpr int BN = 2 * FN * 32;
  const int tiles = ((p->M + 255) / 256) * ((p->N + BN - 1) / BN);
  const int cus = stream_cus();
  const int G = tiles < cus ? tiles : cus;
  int full_items = tiles, nitems = tiles;
  const int R = tiles % G;
  if (tiles > G && R > 0 && 2 * R <= G) {  // the last round would hold at most half the workgroups: cut its tiles into 128-row halves
    full_items = tiles - R;
    nitems = full_items + 2 * R;
  }
  if (nsw)
    hipLaunchKernelGGL((gemm_stream_kernel<FN, LNF, GEGLU, RES, 1>), dim3(G), dim3(512), 0, s, *p, tiles, full_items, nitems);
  else
    hipLaunchKernelGGL((gemm_stream_kernel<FN, LNF, GEGLU, RES, 0>), dim3(G), dim3(512), 0, s, *p, tiles, full_items, nitems);
  return 0;
}

}  // namespace

// true when the persistent kernel takes this product: plain single-source loader, whole 32-deep K tiles (at least four of them: the bias rows
// of item j+1 are staged while item j-1's set is long dead), bf16 output whose rows (and the residual's) are 16-byte addressable and below
// 2 GiB (buffer-descriptor offsets; 0x80000000 is the rejected offset), no temb row-bias / accumulate / fp32 output / row range.
bool lvd_gemm_stream_eligible(const lvd_gemm_params* p) {
  if (p->mode != LVD_A_PLAIN || p->a2 || p->rowbias || p->accumulate || p->out_fp32 || p->m_begin) return false;
  if (p->K % 32 || p->K < 128 || p->N % 16 || p->N < 16 || (p->ldc & 7) || (reinterpret_cast<uintptr_t>(p->out) & 15)) return false;
  if ((long)p->M * p->lda1 >= (1L << 30) || (long)p->N * p->K >= (1L << 30) || (long)p->M * p->ldc >= (1L << 30)) return false;
  if (p->res && ((p->ldres & 7) || (reinterpret_cast<uintptr_t>(p->res) & 15) || (long)p->M * p->ldres >= (1L << 30))) return false;
  if (p->res && (p->ln_mean_rstd || p->act == LVD_ACT_GEGLU)) return false;
  if (p->act == LVD_ACT_GEGLU && p->N % 32) return false;
  return true;
}

// geometry 0: 256x320 when N is a multiple of 320 (and the product is not GEGLU), else 256x256.  nsw: 1 = the wait behind an epilogue
// steps over the stores of its last block (production), 0 = it waits for every store (developer A/B).
int lvd_gemm_stream_dispatch(const lvd_gemm_params* p, void* stream, int nsw) {
  hipStream_t s = (hipStream_t)stream;
  const bool geglu = p->act == LVD_ACT_GEGLU, lnf = p->ln_mean_rstd != nullptr, res = p->res != nullptr;
  const bool n320 = !geglu && p->N % 320 == 0;
  if (geglu) return lnf ? launch_stream<4, true, true, false>(p, s, nsw) : launch_stream<4, false, true, false>(p, s, nsw);
  if (n320) {
    if (lnf) return launch_stream<5, true, false, false>(p, s, nsw);
    return res ? launch_stream<5, false, false, true>(p, s, nsw) : launch_stream<5, false, false, false>(p, s, nsw);
  }
  if (lnf) return launch_stream<4, true, false, false>(p, s, nsw);
  return res ? launch_stream<4, false, false, true>(p, s, nsw) : launch_stream<4, false, false, false>(p, s, nsw);
}
