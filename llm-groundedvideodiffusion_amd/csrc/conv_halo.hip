// conv_halo.hip — tap GEMMs (3x3 convolution stride 1 pad 1; temporal (3,1,1) convolution) whose im2col tile lives in LDS.
//
// The implicit-im2col loaders of gemm_ring.hip stage every tap separately: a 256-row tile DMAs its input rows once per tap
// and channel chunk (9x / 3x the A-side staging bytes) and walks K tap-major, so the re-reads fall out of the 4 MB L2 (PMC,
// round 1: 724 MB fetched for an 88 MB input).  The L2->LDS DMA path is what bounds the 8-wave tiles, so
// the lever is flop per staged byte:
//   * K runs chunk-major: for each 32-channel chunk the rows the tile needs for ALL taps are staged ONCE into a
//     double-buffered LDS image, and a tap is a shifted view of that image: the A fragment of tap t is read at
//     row + shift(t), and lanes whose tap falls outside the image / clip read a zero line instead (per-lane validity mask,
//     one v_cndmask on the LDS address per fragment);
//       3x3 conv:      image = the tile's 512 consecutive token rows + (W+1) rows of halo on each side (one image row and
//                      one pixel); shift(ky,kx) = (ky-1)·W + (kx-1)
//       temporal conv: the tile is P = 512/F pixels x all F frames, image row f·P + pixel; shift(dt) = (dt-1)·P, no halo;
//                      token rows are gathered / scattered through the row map (b, f, pixel) -> (b·F + f)·HW + pixel
//   * with the activations resident only the weights are staged per (chunk, tap) and flop per staged byte ~ BM: the tile is
//     512 rows x 160 (128) columns — eight waves stacked along M, each with the 64x160 register tile of the ring kernel —
//     so a K tile of the 3x3 conv stages 10 KB of weights + 4.7 KB of halo for 5.2 MFLOP: 356 flop per staged byte
//     instead of 142 (temporal conv: 10 KB + 10.7 KB: 250);
//   * weights keep the tap-major [Cout, T·Cin] layout of the rest of the library (a (chunk, tap) tile is a 64-byte run
//     per output channel), so the same packed tensor feeds either kernel.
// Schedule: the 8-wave ping-pong of gemm_ring.hip — waves 4-7 run one phase behind waves 0-3, a K tile is an L phase
// (fragments LDS -> registers, DMA issue, counted wait) and an M phase (20 MFMAs), 3-deep weight ring, 2-deep image.
// The tap loop is unrolled (ring slot, tap shift and validity bit are compile-time), the chunk loop is the runtime loop.
// DMA = buffer_load ... lds issued as inline assembly (the compiler would drain every DMA before the first ds_read of a
// phase).  Split-K slices are ranges of chunks writing compact fp32 slabs [slice][tile][512][BN]; the kernel's own reduce
// applies the epilogue through the same row map.  A launch covers tiles [tile_base, tile_base + grid): whole rounds of the
// 256 CUs go to the unsplit kernel, the remaining tiles to the split-K one.
#include "common.h"
#include "gemm_tile.h"

namespace {

constexpr int HALO_BM = 512;
// image rows (whole wave-instructions of 16 rows per wave): 3x3 conv 768 >= 512 + 2·(W+1) for W <= 87; temporal conv F·P <= 512
template <int MODE> constexpr int img_rows() { return MODE == LVD_A_CONV3X3 ? 768 : 512; }

// Geometry of one tile (wave-uniform scalars).  Image row r (0 .. img_rows) <-> token row; tile row l (0 .. 511) <-> token
// row of the output and image row l + lead.
template <int MODE>
struct TapGeo {
  int m0;      // conv: first output row of the tile
  int lead;    // conv: W + 1 halo rows in front of the tile; tconv: 0
  int need;    // image rows that hold data
  int P, base, pix0, pixn;  // tconv: pixels per tile, token row of (b, f = 0, pixel 0), first pixel, pixels in this tile
  int hw, magic;            // magic = ceil(2^20 / P): r / P == (r · magic) >> 20 exactly for r < 768, P <= 256
  int up, W, HWo;           // conv with a nearest-x2 upsampled source: output (= virtual input) width, rows per image
  LVD_DEV int frame_of(int r) const { return (r * magic) >> 20; }
  LVD_DEV void init(const lvd_gemm_params& p, int tm) {
    if (MODE == LVD_A_CONV3X3) {
      m0 = p.m_begin + tm * HALO_BM;
      lead = p.win + 1;
      need = HALO_BM + 2 * lead;
      P = base = pix0 = pixn = hw = magic = 0;
      up = p.upsample; W = p.win; HWo = p.hin * p.win;
    } else {
      P = HALO_BM / p.frames;
      magic = ((1 << 20) + P - 1) / P;
      hw = p.hw;
      const int nblk = (hw + P - 1) / P;
      const int b = tm / nblk;
      pix0 = (tm - b * nblk) * P;
      pixn = min(P, hw - pix0);
      base = b * p.frames * hw;
      need = p.frames * P;
      m0 = 0;
      lead = 0;
      up = W = HWo = 0;
    }
  }
  // token row of image row r, or -1
  LVD_DEV int src_row(int r, int arows) const {
    if (MODE == LVD_A_CONV3X3) {
      const int g = m0 - lead + r;
      if (!(r < need && g >= 0 && g < arows)) return -1;
      if (!up) return g;
      // the source is stored at half resolution: virtual pixel (iy, ix) of image n reads stored pixel (iy/2, ix/2)
      const int n = g / HWo, rem = g - n * HWo;
      const int iy = rem / W, ix = rem - iy * W;
      return n * (HWo >> 2) + (iy >> 1) * (W >> 1) + (ix >> 1);
    } else {
      const int f = frame_of(r), pp = r - f * P;
      return (r < need && pp < pixn) ? base + f * hw + pix0 + pp : -1;
    }
  }
  // token row of tile row l (>= M when the tile row is padding)
  LVD_DEV int out_row(int l) const {
    if (MODE == LVD_A_CONV3X3) return m0 + l;
    const int f = frame_of(l), pp = l - f * P;
    return (l < need && pp < pixn) ? base + f * hw + pix0 + pp : 0x7fffffff;
  }
};

template <int MODE>
struct TapRows {
  TapGeo<MODE> g;
  int l0;
  LVD_DEV int operator()(int local) const { return g.out_row(l0 + local); }
};

template <int MODE, int FN, bool SPLITK>
__global__ __launch_bounds__(512, 2) void tap_gemm_kernel(const lvd_gemm_params p, const int tile_base) {
  constexpr int T = MODE == LVD_A_CONV3X3 ? 9 : 3;  // taps
  constexpr int FM = 2, NW = 8, RCH = 4, RPI = 16, CH = 32;
  constexpr int BM = HALO_BM, BN = FN * 32;
  constexpr int HALO_ROWS = img_rows<MODE>();
  constexpr int ASZ = HALO_ROWS * RCH;          // uint4 per image buffer
  constexpr int BSZ = BN * RCH;                 // uint4 per weight stage
  constexpr int AH = HALO_ROWS / RPI / NW;      // image instructions per wave per chunk (6 / 4)
  constexpr int APT = (AH + T - 1) / T;         // ... issued APT per tap in the first AH / APT taps
  constexpr int BINS = BN / RPI;                // 10 (8) weight instructions per K tile
  constexpr int BPW = (BINS + NW - 1) / NW;     // per wave (2 / 1)
  constexpr int ZOFF = 2 * ASZ + 3 * BSZ;       // 64-byte zero line
  static_assert((ZOFF & 3) == 0 && (ASZ & 3) == 0, "zero line must keep bit 1 of the chunk index free");
  static_assert(AH % APT == 0 && AH / APT <= T && HALO_ROWS % (RPI * NW) == 0 && BPW <= 2, "instruction bookkeeping");
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
  const int ntile = SPLITK ? nb / p.ksplit : nb;  // tiles of this launch
  const int slice = SPLITK ? id / ntile : 0;
  if (SPLITK) id -= slice * ntile;
  const int tile = tile_base + id;
  const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
  const int nchunk = p.cin / CH;
  int cbeg = 0, cend = nchunk;
  if (SPLITK) {
    const int per = (nchunk + p.ksplit - 1) / p.ksplit;
    cbeg = slice * per;
    cend = min(nchunk, cbeg + per);
  }

  TapGeo<MODE> geo;
  geo.init(p, tm);
  const int cpos = lane & 3, rsub = lane >> 2;
  auto swz = [](int row, int c) { return c ^ ((row >> 2) & 3); };

  // DMA sources are raw buffer descriptors (base + 32-bit per-lane byte offset + scalar byte offset): no 64-bit address
  // VGPRs, no induction variables for the compiler to multiply.  Lanes with nothing to fetch (image rows outside the token
  // matrix / clip, weight rows >= N) read offset 0 — whatever lands in LDS for them is masked by the validity bits or feeds
  // output columns that are never stored; chunks past the end (pipeline overrun) re-read the last chunk.
  const v4i rsA = make_rsrc(p.a1), rsB = make_rsrc(p.w);
  const int arows = MODE == LVD_A_CONV3X3 ? p.M : 0;
  const int lastc = nchunk - 1;
  const int hrow0 = wave * AH * RPI + rsub;
  const int aswz = swz(hrow0, cpos) * 16;  // bytes; swz(hrow0 + 16q) = swz(hrow0)
  auto stage_a = [&](int c, int q) {
    const int g = geo.src_row(hrow0 + q * RPI, arows);
    const int voff = g >= 0 ? g * p.lda1 * 2 + aswz : 0;
    dma16(rsA, voff, min(c, lastc) * (CH * 2), lds_addr(lds + ((c - cbeg) & 1) * ASZ + (wave * AH + q) * RPI * RCH));
  };

  // ---- B side: weight rows n of this tile; a (chunk, tap) K tile is the 64-byte run at k = tap·Cin + chunk·32.
  // Instruction b = wave + 8·t covers tile rows 16·b ..; a tile has BINS of them, so waves with wave + 8 >= BINS issue one
  // instruction less (wave-uniform), and wait for one less.
  const int brow0 = wave * RPI + rsub;
  const int woff0 = (tn * BN + brow0 < p.N) ? ((tn * BN + brow0) * p.K + swz(brow0, cpos) * 8) * 2 : 0;
  const bool two_b = BPW > 1 && wave + NW < BINS;
  const int woff1 = (BPW > 1 && tn * BN + brow0 + NW * RPI < p.N) ? woff0 + NW * RPI * p.K * 2 : 0;
  auto stage_b = [&](int c, int tap, int slot) {
    const int soff = (tap * p.cin + min(c, lastc) * CH) * 2;
    dma16(rsB, woff0, soff, lds_addr(lds + 2 * ASZ + slot * BSZ + wave * RPI * RCH));
    if (two_b) dma16(rsB, woff1, soff, lds_addr(lds + 2 * ASZ + slot * BSZ + (wave + NW) * RPI * RCH));
  };

  // ---- fragment addressing: image rows of fragment i are rloc + 32·i; the validity masks of both fragments share one register
  const int rloc = wave * FM * 32 + l31 + geo.lead;
  int vmask = 0;
#pragma unroll
  for (int i = 0; i < FM; ++i) {
    const int l = wave * FM * 32 + i * 32 + l31;
    int mk = 0;
    if (MODE == LVD_A_CONV3X3) {
      const int W = p.win, H = p.hin;
      const int m = geo.m0 + l;
      const int rem = m % (H * W);
      const int oy = rem / W, ox = rem - oy * W;
      const int vy = (oy > 0 ? 1 : 0) | 2 | (oy < H - 1 ? 4 : 0);
      const int vx = (ox > 0 ? 1 : 0) | 2 | (ox < W - 1 ? 4 : 0);
#pragma unroll
      for (int t = 0; t < 9; ++t)
        if (((vy >> (t / 3)) & 1) && ((vx >> (t % 3)) & 1)) mk |= 1 << t;
      if (m >= p.M) mk = 0;
    } else {
      const int f = geo.frame_of(l), pp = l - f * geo.P;
      mk = (f > 0 ? 1 : 0) | 2 | (f < p.frames - 1 ? 4 : 0);
      if (l >= geo.need || pp >= geo.pixn) mk = 0;
    }
    vmask |= mk << (16 * i);
  }
  const int boff0 = l31 * RCH + swz(l31, hi);  // B fragment j sits 32 rows = 128 uint4 further (same swizzle)
  const int bx = (boff0 ^ 2) - boff0;          // second k-step: chunk index ^ 2 (+2 or -2 uint4)
  const int tapstep = MODE == LVD_A_CONV3X3 ? p.win : geo.P;

  // prologue: image of the first chunk, weight tiles 0 and 1
#pragma unroll
  for (int q = 0; q < AH; ++q) stage_a(cbeg, q);
  stage_b(cbeg, 0, 0);
  stage_b(cbeg, 1, 1);
  if (tid < 4) lds[ZOFF + tid] = make_uint4(0u, 0u, 0u, 0u);

  TapRows<MODE> rows{geo, wave * FM * 32};
  f32x16 acc[FM][FN];
  if (SPLITK) {
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  } else {
    ring_bias_init<FM, FN>(p, acc, rows, tn * BN, l31, hi);
  }

  const int group = wave >> 2;
  // image 0 and weight tile 0 have landed (weight tile 1 may still be in flight)
  if (two_b) wait_vmcnt<2>();
  else wait_vmcnt<1>();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (group == 1) __builtin_amdgcn_s_barrier();

  for (int c = cbeg; c < cend; ++c) {
    const int abuf = ((c - cbeg) & 1) * ASZ;
#pragma unroll
    for (int t = 0; t < T; ++t) {
      const int slot = t % 3, nslot = (t + 2) % 3;
      const int sh = MODE == LVD_A_CONV3X3 ? (t / 3 - 1) * tapstep + (t % 3 - 1) : (t - 1) * tapstep;
      bf16x8 af[2][FM], bfr[2][FN];
      // opaque copies: keep the taps' A addresses and the 3 x 10 B addresses from being precomputed outside the chunk
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
      // refills: the image of the next chunk (first taps), the weight tile two K tiles ahead
      const bool a_now = t * APT < AH;
      if (a_now) {
#pragma unroll
        for (int q = 0; q < APT; ++q) stage_a(c + 1, t * APT + q);
      }
      if (t + 2 < T) stage_b(c, t + 2, nslot);
      else stage_b(c + 1, t + 2 - T, nslot);
      // weight tile kt+1 (issued one L phase ago) must have landed: everything younger may stay in flight
      if (two_b) {
        if (a_now) wait_vmcnt<2 + APT>();
        else wait_vmcnt<2>();
      } else {
        if (a_now) wait_vmcnt<1 + APT>();
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

  const int nbase = tn * BN;
  if (SPLITK) {  // raw fp32 partial sums, compact: [slice][tile of this launch][512][BN]
    float* slab = p.ws + ((long)(slice * ntile + id) * BM + wave * FM * 32) * BN;
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          f32x4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * q + e];
          *reinterpret_cast<f32x4*>(slab + (long)(i * 32 + l31) * BN + j * 32 + 8 * q + 4 * hi) = v;
        }
    return;
  }
  constexpr int WAVE_DW = (ZOFF * 4) / NW;
  static_assert(WAVE_DW >= 32 * (FN * 16 + 2), "LDS too small for the epilogue strip");
  __builtin_amdgcn_s_barrier();
  ring_epilogue_auto<FM, FN>(p, acc, rows, nbase, lane, reinterpret_cast<uint32_t*>(lds) + wave * WAVE_DW);
}

// Slab reduction of the split-K launch + the usual epilogue (bias, temb row-bias, alpha, residual, accumulate), rows through
// the tile's row map.  One thread = 4 columns of one tile row.
template <int MODE, int FN>
__global__ void tap_reduce_kernel(const lvd_gemm_params p, const int tile_base, const int ntile) {
  constexpr int BN = FN * 32, QN = BN / 4;
  const int tiles_n = (p.N + BN - 1) / BN;
  const long total = (long)ntile * HALO_BM * QN;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int q = (int)(i % QN);
    const long rowi = i / QN;
    const int l = (int)(rowi % HALO_BM), t = (int)(rowi / HALO_BM);
    const int tile = tile_base + t;
    const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
    TapGeo<MODE> geo;
    geo.init(p, tm);
    const int m = geo.out_row(l);
    const int n = tn * BN + q * 4;
    if (m >= p.M || n >= p.N) continue;
    splitk_reduce_quad(p, p.ws + ((long)t * HALO_BM + l) * BN + q * 4, (long)ntile * HALO_BM * BN, m, n);
  }
}

template <int MODE, int FN>
void launch_tiles(const lvd_gemm_params& p, hipStream_t s, int tile_base, int ntile, int ks) {
  if (ntile <= 0) return;
  if (ks < 2) {
    hipLaunchKernelGGL((tap_gemm_kernel<MODE, FN, false>), dim3(ntile), dim3(512), 0, s, p, tile_base);
    return;
  }
  lvd_gemm_params q = p;
  q.ksplit = ks;
  hipLaunchKernelGGL((tap_gemm_kernel<MODE, FN, true>), dim3(ntile * ks), dim3(512), 0, s, q, tile_base);
  const long quads = (long)ntile * HALO_BM * (FN * 8);
  int rb = (int)((quads + 255) / 256);
  if (rb > 4096) rb = 4096;
  hipLaunchKernelGGL((tap_reduce_kernel<MODE, FN>), dim3(rb), dim3(256), 0, s, q, tile_base, ntile);
}

// Number of channel-chunk slices for `tiles` tiles on 256 CUs.  Times in microseconds, measured on MI355X: ~8 per chunk of
// the 3x3 conv (18 phases), ~2.7 per chunk of the temporal conv, ~6 of prologue + slab store per workgroup, slab reduction
// ~3 + bytes / 3 TB/s.  Slices are whole chunks: ks is what ceil-division really yields.
int plan_slices(long tiles, int nchunk, int taps, int bn, long ws_bytes) {
  const double per_chunk = taps == 9 ? 8.0 : 2.7;
  double best = 1e30;
  int best_ks = 1;
  for (int per = nchunk; per >= 1; --per) {
    const int ks = (nchunk + per - 1) / per;
    if (ks > 16) break;
    if (ks > 1 && (long)ks * tiles * HALO_BM * bn * 4 > ws_bytes) break;
    const long rounds = (tiles * ks + 255) / 256;
    double t = rounds * (per * per_chunk + 6.0);
    if (ks > 1) t += 3.0 + (double)ks * tiles * HALO_BM * bn * 4 / 3.0e6;
    if (t < best) { best = t; best_ks = ks; }
  }
  return best_ks;
}

template <int MODE>
int dispatch_mode(const lvd_gemm_params& p, hipStream_t s, int plan) {
  const bool n160 = p.N % 160 == 0;
  const int bn = n160 ? 160 : 128;
  const int tiles_n = (p.N + bn - 1) / bn;
  int tiles_m;
  if (MODE == LVD_A_CONV3X3) {
    tiles_m = (p.M - p.m_begin + HALO_BM - 1) / HALO_BM;
  } else {
    const int P = HALO_BM / p.frames;
    tiles_m = (p.M / (p.frames * p.hw)) * ((p.hw + P - 1) / P);
  }
  const int tiles = tiles_m * tiles_n;
  const int nchunk = p.cin / 32, taps = MODE == LVD_A_CONV3X3 ? 9 : 3;
  const bool can_split = p.ws && p.act == LVD_ACT_NONE && nchunk >= 2;
  auto go = [&](int base, int n, int ks) {
    if (n160) launch_tiles<MODE, 5>(p, s, base, n, ks);
    else launch_tiles<MODE, 4>(p, s, base, n, ks);
  };
  if (plan == 0 || !can_split) {  // one workgroup per tile
    go(0, tiles, 1);
    return 0;
  }
  if (plan == 1) {  // channel chunks split over workgroups wherever the plan says it pays
    int ks = p.ksplit > 1 ? min(p.ksplit, nchunk) : plan_slices(tiles, nchunk, taps, bn, p.ws_bytes);
    while (ks > 1 && (long)ks * tiles * HALO_BM * bn * 4 > p.ws_bytes) --ks;
    go(0, tiles, ks);
    return 0;
  }
  // plan 2: whole rounds of the 256 CUs unsplit, the remaining tiles split
  const int head = tiles / 256 * 256, rem = tiles - head;
  if (head == 0 || rem == 0 || rem * 10 > 256 * 6) {
    go(0, tiles, head == 0 ? plan_slices(tiles, nchunk, taps, bn, p.ws_bytes) : 1);
    return 0;
  }
  go(0, head, 1);
  go(head, rem, plan_slices(rem, nchunk, taps, bn, p.ws_bytes));
  return 0;
}

}  // namespace

// true when the LDS-resident kernel can run this product: 3x3 stride-1 pad-1 conv (W <= 87) or temporal conv (2 <= F <= 256)
// from ONE source, 32-channel chunks, 32-bit byte offsets
bool lvd_conv_halo_eligible(const lvd_gemm_params* p) {
  if (p->a2 != nullptr || p->cin % 32 != 0 || p->c1 < p->cin || p->cin < 32) return false;
  if ((long)p->M * p->lda1 >= (1L << 30) || (long)p->N * p->K >= (1L << 30)) return false;
  if (p->mode == LVD_A_CONV3X3)
    return p->stride == 1 && p->hin == p->hout && p->win == p->wout && p->win <= 87 && (!p->upsample || (p->hin % 2 == 0 && p->win % 2 == 0));
  if (p->mode == LVD_A_TCONV3) return p->frames >= 2 && p->frames <= 256 && p->m_begin == 0 && p->M % (p->frames * p->hw) == 0;
  return false;
}

// plan 0: one workgroup per 512 x (160|128) tile; 1: split-K where it pays; 2: whole rounds unsplit + split-K remainder
int lvd_conv_halo_dispatch(const lvd_gemm_params* p, void* stream, int plan) {
  hipStream_t s = (hipStream_t)stream;
  if (p->mode == LVD_A_CONV3X3) return dispatch_mode<LVD_A_CONV3X3>(*p, s, plan);
  return dispatch_mode<LVD_A_TCONV3>(*p, s, plan);
}
