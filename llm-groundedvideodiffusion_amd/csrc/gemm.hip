// gemm.hip — bf16 MFMA GEMM family for gfx950:  OUT[M,N] = epi( Aload[M,K] · W[N,K]^T ).
//
// One kernel template covers every dense contraction of the denoise step (SURVEY §2.3 K3-K8,
// K11-K14, K16): the A operand is generated on the fly by a loader —
//   PLAIN      token matrix (optionally the channel-concat of two matrices: the skip "torch.cat")
//   CONV3X3    implicit im2col of a 3x3 pad-1 conv (stride 1|2, optional nearest-x2 source)
//   TCONV3     3-tap temporal conv: the taps are the same token matrix shifted by ±HW rows
//   CONV3X3_T2 transposed stride-2 conv (input-gradient of Downsample2D)
// so no im2col buffer, no concat buffer and no (B·F,C,H,W)<->(B,C,F,H,W) permute ever touch HBM.
//
// Tile: 128x128x64 per 256-thread workgroup (4 waves, 2x2, each 64x64 = 2x2 MFMA 32x32x16 tiles),
// register-staged global->LDS double buffer (one barrier per K-tile), XOR-swizzled LDS rows so the
// ds_read_b128 fragment reads are conflict-free, XCD-aware block->tile map (consecutive tiles of an
// XCD share the A row-panel in that XCD's L2).  Epilogue goes through LDS so that bias / temb row
// bias / GEGLU / gate / residual are applied on 8-16 byte row-contiguous vectors.
#include <cstdlib>
#include "common.h"
#include "gemm_tile.h"

namespace {

constexpr int BM = 128, BN = 128;

struct ARow {
  long off1, off2;  // PLAIN: row offsets; CONV: image base row (off1) ; TCONV: row index m (off1)
  int oy, ox;       // CONV: output pixel; TCONV: oy = frame index
  bool valid;
};

template <int MODE>
LVD_DEV uint4 load_a(const lvd_gemm_params& p, const ARow& r, int k0) {
  uint4 z = make_uint4(0, 0, 0, 0);
  if (!r.valid || k0 >= p.K) return z;
  if (MODE == LVD_A_PLAIN) {
    if (k0 < p.c1) return ldg16(p.a1 + r.off1 + k0);
    return ldg16(p.a2 + r.off2 + (k0 - p.c1));
  } else if (MODE == LVD_A_CONV3X3) {
    int tap = k0 / p.cin;
    int c = k0 - tap * p.cin;
    int ky = tap / 3, kx = tap - 3 * ky;
    int iy = r.oy * p.stride + ky - 1, ix = r.ox * p.stride + kx - 1;
    if (iy < 0 || iy >= p.hin || ix < 0 || ix >= p.win) return z;
    int ws = p.win;
    if (p.upsample) { iy >>= 1; ix >>= 1; ws >>= 1; }
    long row = r.off1 + (long)iy * ws + ix;
    if (c < p.c1) return ldg16(p.a1 + row * p.lda1 + c);
    return ldg16(p.a2 + row * p.lda2 + (c - p.c1));
  } else if (MODE == LVD_A_CONV3X3_T2) {
    int tap = k0 / p.cin;
    int c = k0 - tap * p.cin;
    int ky = tap / 3, kx = tap - 3 * ky;
    int ty = r.oy + 1 - ky, tx = r.ox + 1 - kx;
    if (ty < 0 || tx < 0 || (ty & 1) || (tx & 1)) return z;
    ty >>= 1; tx >>= 1;
    if (ty >= p.hin || tx >= p.win) return z;
    long row = r.off1 + (long)ty * p.win + tx;
    return ldg16(p.a1 + row * p.lda1 + c);
  } else {  // TCONV3
    int tap = k0 / p.cin;
    int c = k0 - tap * p.cin;
    int ff = r.oy + tap - 1;
    if (ff < 0 || ff >= p.frames) return z;
    long row = r.off1 + (long)(tap - 1) * p.hw;
    if (c < p.c1) return ldg16(p.a1 + row * p.lda1 + c);
    return ldg16(p.a2 + row * p.lda2 + (c - p.c1));
  }
}

// BK = K-tile depth (64: 64 KiB LDS, 2 workgroups/CU; 32: 32 KiB LDS, up to 4 workgroups/CU), MINW = waves/SIMD the
// register allocator must leave room for.
template <int MODE, int BK, int MINW>
__global__ __launch_bounds__(256, MINW) void gemm_kernel(const lvd_gemm_params p) {
  constexpr int CH = BK / 8;            // 16-byte chunks per tile row
  constexpr int NLD = (BM * CH) / 256;  // 16-byte loads per thread per operand per K tile
  constexpr int RSTEP = 256 / CH;       // row distance between a thread's loads
  constexpr int TILE = (BM + BN) * CH;  // uint4 per buffer
  __shared__ uint4 lds[2 * TILE];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, hi = lane >> 5;

  // XCD-aware bijective remap: blocks b, b+8, b+16.. (same XCD) get consecutive tile ids
  const int nb = gridDim.x;
  int id;
  {
    int bid = blockIdx.x;
    int q = nb >> 3, r = nb & 7;
    int xcd = bid & 7, idx = bid >> 3;
    id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tiles_n = (p.N + BN - 1) / BN;
  const int tm = id / tiles_n, tn = id - tm * tiles_n;

  const int vc = tid % CH;  // 16-byte chunk (8 bf16) within the K tile
  const int r0 = tid / CH;

  ARow ar[NLD];
  long woff[NLD];
  bool wvalid[NLD];
#pragma unroll
  for (int i = 0; i < NLD; ++i) {
    int m = p.m_begin + tm * BM + r0 + RSTEP * i;
    ar[i].valid = m < p.M;
    ar[i].off1 = 0; ar[i].off2 = 0; ar[i].oy = 0; ar[i].ox = 0;
    if (MODE == LVD_A_PLAIN) {
      ar[i].off1 = (long)m * p.lda1;
      ar[i].off2 = (long)m * p.lda2;
    } else if (MODE == LVD_A_CONV3X3 || MODE == LVD_A_CONV3X3_T2) {
      int plane = p.hout * p.wout;
      int nimg = m / plane;
      int rem = m - nimg * plane;
      ar[i].oy = rem / p.wout;
      ar[i].ox = rem - ar[i].oy * p.wout;
      int hs = p.hin, ws = p.win;
      if (MODE == LVD_A_CONV3X3 && p.upsample) { hs >>= 1; ws >>= 1; }
      ar[i].off1 = (long)nimg * hs * ws;
    } else {
      ar[i].off1 = m;
      ar[i].oy = (m / p.hw) % p.frames;
    }
    int n = tn * BN + r0 + RSTEP * i;
    wvalid[i] = n < p.N;
    woff[i] = (long)n * p.K;
  }

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int nk = (p.K + BK - 1) / BK;
  uint4 ra[NLD], rb[NLD];

  // XOR swizzle of the 16-byte chunk index so that the 16-lane groups of ds_read_b128 hit 16 distinct slots of
  // the 256-byte bank row: BK=64 (128-byte rows): c ^ ((row>>1)&7);  BK=32 (64-byte rows): c ^ ((row>>2)&3)
  auto swz = [](int row, int c) { return BK == 64 ? (c ^ ((row >> 1) & 7)) : (c ^ ((row >> 2) & 3)); };

  auto load_tile = [&](int kt) {
    int k0 = kt * BK + vc * 8;
#pragma unroll
    for (int i = 0; i < NLD; ++i) ra[i] = load_a<MODE>(p, ar[i], k0);
#pragma unroll
    for (int i = 0; i < NLD; ++i)
      rb[i] = (wvalid[i] && k0 < p.K) ? ldg16(p.w + woff[i] + k0) : make_uint4(0, 0, 0, 0);
  };
  auto store_tile = [&](int buf) {
    uint4* A = lds + buf * TILE;
    uint4* B = A + BM * CH;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      int row = r0 + RSTEP * i;
      int sw = swz(row, vc);
      A[row * CH + sw] = ra[i];
      B[row * CH + sw] = rb[i];
    }
  };

  load_tile(0);
  store_tile(0);
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) load_tile(kt + 1);
    const uint4* A = lds + cur * TILE;
    const uint4* B = A + BM * CH;
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      bf16x8 af[2], bfr[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        int row = wm * 64 + i * 32 + l31;
        af[i] = as_bf16x8(A[row * CH + swz(row, ks * 2 + hi)]);
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        int row = wn * 64 + j * 32 + l31;
        bfr[j] = as_bf16x8(B[row * CH + swz(row, ks * 2 + hi)]);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nk) store_tile(cur ^ 1);
    __syncthreads();
  }

  // ---- epilogue: stage the wave's fp32 tile in LDS (RP rows per pass), then row-contiguous vector math ----
  constexpr int RP = BK;             // rows per pass: the wave's LDS share is TILE*2*16/4 bytes = RP rows x 64 fp32
  constexpr int NP = 64 / RP;        // passes (BK=64: 1, BK=32: 2)
  float* S = reinterpret_cast<float*>(lds) + wave * (RP * 64);
  const int mbase = p.m_begin + tm * BM + wm * 64;
  const int nbase = tn * BN + wn * 64;
#pragma unroll
  for (int ps = 0; ps < NP; ++ps) {
    if (ps > 0) __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (NP == 2 && i != ps) continue;
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          int rowl = (NP == 2 ? 0 : i * 32) + (e & 3) + 8 * (e >> 2) + 4 * hi;
          S[rowl * 64 + j * 32 + l31] = acc[i][j][e];
        }
    }
    __syncthreads();
    const int mrow0 = mbase + ps * RP;

    if (p.act == LVD_ACT_GEGLU) {
      const int ldc = p.ldc;
      lvd_bf16* out = reinterpret_cast<lvd_bf16*>(p.out);
#pragma unroll
      for (int it = 0; it < RP / 8; ++it) {
        int idx = it * 64 + lane;
        int row = idx >> 3, cq = idx & 7;
        int m = mrow0 + row;
        int n = nbase + cq * 4;  // hidden column in the interleaved W'; gate = n + 32
        if (m >= p.M || n + 32 >= p.N) continue;
        f32x4 h = *reinterpret_cast<const f32x4*>(&S[row * 64 + cq * 4]);
        f32x4 g = *reinterpret_cast<const f32x4*>(&S[row * 64 + 32 + cq * 4]);
        if (p.bias) {
          h += *reinterpret_cast<const f32x4*>(p.bias + n);
          g += *reinterpret_cast<const f32x4*>(p.bias + n + 32);
        }
        int oc = (nbase >> 1) + cq * 4;
        uint2 o;
        o = geglu4(h, g);
        stg8(out + (long)m * ldc + oc, o);
      }
      continue;
    }

    // residual / accumulate operands first, all in flight together: the stores below may alias them as far as the
    // compiler knows, so a load inside the store loop would be chained behind the previous store
    uint2 rres[RP / 4], racc[RP / 4];
    f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};  // this lane's 4 columns are the same in every pass: one load, outside the lane branch
    if (p.bias) bias4 = *reinterpret_cast<const f32x4*>(p.bias + min(nbase + (lane & 15) * 4, p.N - 4));
    if (p.res) {  // clamped, never predicated per lane: a load in a lane branch is waited for at the end of that branch
#pragma unroll
      for (int it = 0; it < RP / 4; ++it) {
        int idx = it * 64 + lane;
        int m = min(mrow0 + (idx >> 4), p.M - 1);
        int n = min(nbase + (idx & 15) * 4, p.N - 4);
        rres[it] = ldg8(p.res + (long)m * p.ldres + n);
      }
    }
    if (p.accumulate && !p.out_fp32) {
#pragma unroll
      for (int it = 0; it < RP / 4; ++it) {
        int idx = it * 64 + lane;
        int m = min(mrow0 + (idx >> 4), p.M - 1);
        int n = min(nbase + (idx & 15) * 4, p.N - 4);
        racc[it] = ldg8(reinterpret_cast<const lvd_bf16*>(p.out) + (long)m * p.ldc + n);
      }
    }
#pragma unroll
    for (int it = 0; it < RP / 4; ++it) {
      int idx = it * 64 + lane;
      int row = idx >> 4, cq = idx & 15;
      int m = mrow0 + row;
      int n = nbase + cq * 4;
      if (m >= p.M || n >= p.N) continue;
      f32x4 v = *reinterpret_cast<const f32x4*>(&S[row * 64 + cq * 4]);
      v += bias4;
      if (p.rowbias) v += *reinterpret_cast<const f32x4*>(p.rowbias + (long)(m / p.rows_per_sample) * (p.ldrowbias ? p.ldrowbias : p.N) + n);
      if (p.alpha != 1.f) v *= p.alpha;
      if (p.res) {
        uint2 r = rres[it];
        v[0] += bflo(r.x); v[1] += bfhi(r.x); v[2] += bflo(r.y); v[3] += bfhi(r.y);
      }
      if (p.out_fp32) {
        float* o = reinterpret_cast<float*>(p.out) + (long)m * p.ldc + n;
        if (p.accumulate) v += *reinterpret_cast<const f32x4*>(o);
        *reinterpret_cast<f32x4*>(o) = v;
      } else {
        lvd_bf16* o = reinterpret_cast<lvd_bf16*>(p.out) + (long)m * p.ldc + n;
        if (p.accumulate) {
          uint2 r = racc[it];
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

template <int BK, int MINW>
int launch_gemm(const lvd_gemm_params* p, dim3 grid, hipStream_t s) {
  switch (p->mode) {
    case LVD_A_PLAIN: hipLaunchKernelGGL((gemm_kernel<LVD_A_PLAIN, BK, MINW>), grid, dim3(256), 0, s, *p); break;
    case LVD_A_CONV3X3: hipLaunchKernelGGL((gemm_kernel<LVD_A_CONV3X3, BK, MINW>), grid, dim3(256), 0, s, *p); break;
    case LVD_A_TCONV3: hipLaunchKernelGGL((gemm_kernel<LVD_A_TCONV3, BK, MINW>), grid, dim3(256), 0, s, *p); break;
    case LVD_A_CONV3X3_T2: hipLaunchKernelGGL((gemm_kernel<LVD_A_CONV3X3_T2, BK, MINW>), grid, dim3(256), 0, s, *p); break;
    default: return 1;
  }
  return 0;
}

}  // namespace

int lvd_gemm_ring_dispatch(const lvd_gemm_params* p, void* stream, int geometry);  // gemm_ring.hip
bool lvd_conv_halo_eligible(const lvd_gemm_params* p);                               // conv_halo.hip
int lvd_conv_halo_dispatch(const lvd_gemm_params* p, void* stream, int plan);
bool lvd_gemm_stream_eligible(const lvd_gemm_params* p);                             // gemm_stream.hip
int lvd_gemm_stream_dispatch(const lvd_gemm_params* p, void* stream, int nsw);

namespace {

// 256x320 or 256x256 for the wide split-K: whichever the K-split plan predicts to be faster
bool wide320_fills_better(const lvd_gemm_params* p) {
  const int rows = p->M - p->m_begin;
  auto time_of = [&](int bn) {  // planned cost x tile area = relative time
    long tiles = (long)((rows + 255) / 256) * ((p->N + bn - 1) / bn), cost = 0;
    lvd_splitk_plan(tiles, p->K, 256, 256 * bn * 450 / 65536, &cost);
    return cost * bn;
  };
  return time_of(320) <= time_of(256);
}

int run_with_tail(const lvd_gemm_params* p, void* stream, int v);

// One launch (two for split-K) of a pinned or heuristic tile geometry over rows [m_begin, M).  Returns 3 for a variant code
// that names no geometry (a damaged autotune table must not silently pin some kernel).
int run_variant(const lvd_gemm_params* p, void* stream, int v) {
  int tiles = ((p->M - p->m_begin + BM - 1) / BM) * ((p->N + BN - 1) / BN);
  dim3 grid(tiles);
  hipStream_t s = (hipStream_t)stream;
  // + LVD_GEMM_V_ADMA / + LVD_GEMM_V_ADMA64: the asm buffer-DMA instantiations of a ring geometry (gemm_ring.hip); stripped first,
  // so that every base code below is looked at once
  int adma = 0;
  if (v >= 300) return 3;
  if (v >= LVD_GEMM_V_ADMA64) { adma = 200; v -= LVD_GEMM_V_ADMA64; }
  else if (v >= LVD_GEMM_V_ADMA) { adma = 100; v -= LVD_GEMM_V_ADMA; }
  if (v == 0 && !adma && p->ln_mean_rstd) { adma = 100; v = LVD_GEMM_V_RING128; }  // unpinned LayerNorm-folded product: the asm-DMA 128x128 ring
  if (v == 0) {
    if (adma) return 3;
    // measured on MI355X (tools/gemm_bench.py, profiles/r01_gemm_variants.txt):
    //   under-filled grids            -> 128x128x64 register-staged (fewest, longest tiles)
    //   conv / tconv / long-K linear  -> LDS-DMA ring, 3 stages, 3 workgroups per CU
    //   short-K linear                -> 128x128x32 register-staged, 4 workgroups per CU
    if (tiles < 400 && p->ws && p->K >= 1024 && p->act == LVD_ACT_NONE) v = 20;  // split-K (falls back if not splittable)
    else if (tiles < 400) v = 10;
    else if (p->mode != LVD_A_PLAIN || p->K >= 1024) v = 5;
    else v = 1;
  }
  const bool n320 = p->act != LVD_ACT_GEGLU && p->N % 320 == 0;
  // LDS-resident im2col (conv_halo.hip); products it cannot take fall through to the matching implicit-im2col geometry
  if (v == LVD_GEMM_V_CONV_HALO || v == LVD_GEMM_V_CONV_HALO_SPLITK || v == LVD_GEMM_V_CONV_HALO_TAIL) {
    if (adma) return 3;
    if (lvd_conv_halo_eligible(p))
      return lvd_conv_halo_dispatch(p, stream, v == LVD_GEMM_V_CONV_HALO ? 0 : (v == LVD_GEMM_V_CONV_HALO_SPLITK ? 1 : 2));
    if (v == LVD_GEMM_V_CONV_HALO_TAIL) return run_with_tail(p, stream, LVD_GEMM_V_RING256W_TAIL);
    v = v == LVD_GEMM_V_CONV_HALO ? LVD_GEMM_V_RING256W : LVD_GEMM_V_SPLITK_WIDE;
  }
  // persistent walker over the 8-wave ping-pong tiles (gemm_stream.hip; asm-DMA only: 161); products it cannot take run RING256W + ADMA
  if (v == LVD_GEMM_V_STREAM) {
    if (adma != 100) return 3;
    static const int nsw = [] {  // developer A/B: LVD_STREAM_NSW=0 makes the first wait behind an epilogue cover every store
      const char* e = getenv("LVD_STREAM_NSW");
      return e ? atoi(e) : 1;
    }();
    if (lvd_gemm_stream_eligible(p)) return lvd_gemm_stream_dispatch(p, stream, nsw);
    v = LVD_GEMM_V_RING256W;
  }
  const int a100 = adma ? 100 : 0;  // geometries without a 64-deep form take the 32-deep asm-DMA one
  if (v >= 5 && v <= 8) return lvd_gemm_ring_dispatch(p, stream, v - 5 + (v <= 6 ? (v == 5 ? adma : a100) : 0));
  if (v == 14 && !adma) return lvd_gemm_ring_dispatch(p, stream, 8);
  if (v == 20) return lvd_gemm_ring_dispatch(p, stream, 20 + adma);
  if (v == LVD_GEMM_V_SPLITK_WIDE) return lvd_gemm_ring_dispatch(p, stream, (n320 && wide320_fills_better(p) ? 24 : 25) + adma);
  if (v == 17) return lvd_gemm_ring_dispatch(p, stream, (n320 ? 12 : 0) + adma);
  if (v == 11) return lvd_gemm_ring_dispatch(p, stream, (n320 ? 4 : 5) + adma);
  if (v == 9) return lvd_gemm_ring_dispatch(p, stream, ((p->act != LVD_ACT_GEGLU && p->N % 160 == 0) ? 2 : 3) + adma);
  if (adma) return 3;
  if (p->ln_mean_rstd) return 4;  // the register-staged kernels have no LayerNorm-folded epilogue
  if (v == 1) return launch_gemm<32, 3>(p, grid, s);
  if (v == 2) return launch_gemm<32, 4>(p, grid, s);
  if (v == 10) return launch_gemm<64, 2>(p, grid, s);
  return 3;
}

// Tail-aware launch.  All tiles of one GEMM cost the same, so a grid of T tiles on S resident workgroup slots runs
// ceil(T/S) rounds however few tiles the last round holds: 540 tiles of 256x320 on 256 CUs take 3 rounds for 2.1 rounds
// of work.  The rows that fill whole rounds go to the wide geometry; the remaining rows (less than 0.6 of a round) are a
// second, K-split launch that spreads them over the whole chip again.  Row ranges are disjoint: results are identical to
// the unsplit product up to the fp32 summation order of the K slices.
int run_with_tail(const lvd_gemm_params* p, void* stream, int v) {
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (cus <= 0) cus = 256;
  }
  const bool n320 = p->act != LVD_ACT_GEGLU && p->N % 320 == 0;
  const int adma = v >= LVD_GEMM_V_ADMA64 ? LVD_GEMM_V_ADMA64 : (v >= LVD_GEMM_V_ADMA ? LVD_GEMM_V_ADMA : 0);
  v -= adma;
  const bool wide = v == LVD_GEMM_V_RING256W_TAIL;
  const int base = (wide ? LVD_GEMM_V_RING256W : LVD_GEMM_V_RING128x320) + adma;
  const int bm = wide ? 256 : 128;
  const int bn = n320 ? 320 : (wide ? 256 : 128);
  const int slots = cus * (wide ? 1 : (n320 ? 2 : 3));
  const int rows = p->M - p->m_begin;
  const int tiles_n = (p->N + bn - 1) / bn, tiles_m = (rows + bm - 1) / bm;
  const long total = (long)tiles_m * tiles_n;
  const long full = total / slots, rem = total - full * slots;
  const int head_mt = (int)(full * slots / tiles_n);
  if (full < 1 || rem == 0 || rem * 10 > (long)slots * 6 || head_mt < 1 || head_mt >= tiles_m) return run_variant(p, stream, base);
  lvd_gemm_params head = *p, tail = *p;
  head.M = p->m_begin + head_mt * bm;
  tail.m_begin = head.M;
  int rc = run_variant(&head, stream, base);
  if (rc) return rc;
  return run_variant(&tail, stream, ((p->ws && p->act == LVD_ACT_NONE) ? LVD_GEMM_V_SPLITK : LVD_GEMM_V_RING128) + (adma ? LVD_GEMM_V_ADMA : 0));
}

}  // namespace

extern "C" int lvdhip_gemm(const lvd_gemm_params* p, void* stream) {
  LVD_CHECK(p && p->a1 && p->w && p->out, "gemm: null pointer");
  LVD_CHECK(p->M > 0 && p->N > 0 && p->K > 0, "gemm: bad shape M=%d N=%d K=%d", p->M, p->N, p->K);
  LVD_CHECK(p->K % 8 == 0 && p->N % 4 == 0, "gemm: K%%8 / N%%4 violated (K=%d N=%d)", p->K, p->N);
  LVD_CHECK(p->cin % 8 == 0 && p->c1 % 8 == 0, "gemm: cin/c1 must be multiples of 8 (cin=%d c1=%d)", p->cin, p->c1);
  LVD_CHECK(p->lda1 % 8 == 0 && (p->a2 == nullptr || p->lda2 % 8 == 0), "gemm: lda must be a multiple of 8");
  LVD_CHECK(p->a2 != nullptr || p->c1 >= p->cin, "gemm: c1 < cin needs a second source");
  if (p->act == LVD_ACT_GEGLU) LVD_CHECK(p->N % 64 == 0 && !p->out_fp32 && !p->res, "gemm: GEGLU needs N%%64==0, bf16 out, no residual");
  if (p->rowbias) LVD_CHECK(p->rows_per_sample > 0, "gemm: rowbias needs rows_per_sample");
  if (p->mode == LVD_A_TCONV3) LVD_CHECK(p->frames > 0 && p->hw > 0 && p->K == 3 * p->cin, "gemm: bad tconv dims");
  if (p->mode == LVD_A_CONV3X3 || p->mode == LVD_A_CONV3X3_T2)
    LVD_CHECK(p->K == 9 * p->cin && p->hout > 0 && p->wout > 0 && p->hin > 0 && p->win > 0 && p->M % (p->hout * p->wout) == 0,
              "gemm: bad conv dims");
  LVD_CHECK(p->m_begin >= 0 && p->m_begin < p->M, "gemm: m_begin %d outside [0, M=%d)", p->m_begin, p->M);
  if (p->ln_mean_rstd)
    LVD_CHECK(p->ln_colsum && p->bias && p->mode == LVD_A_PLAIN && !p->out_fp32 && !p->rowbias && !p->res && !p->accumulate && !p->a2 && p->N % 16 == 0 &&
                  p->ldc % 8 == 0 && p->K % 32 == 0 && (reinterpret_cast<uintptr_t>(p->out) & 15) == 0,
              "gemm: a LayerNorm-folded product needs ln_colsum, a bias row (b + W beta: the folded epilogue always reads it), the plain single-source loader, K%%32==0, no residual / accumulate / temb row-bias "
              "and a 16-byte addressable bf16 output (N%%16, ldc%%8)");
  static const int variant = [] {  // developer knob for A/B runs (tools/gemm_bench.py); read once, thread-safe initialisation
    const char* e = getenv("LVD_GEMM_VARIANT");
    return e ? atoi(e) : 0;
  }();
  int v = p->variant ? p->variant : variant;
  int rc;
  LVD_CHECK(v >= 0 && v < 300, "gemm: variant %d names no tile geometry", v);  // before the tail / non-tail split: 331, 431, ... must not reach run_with_tail
  const int vb = v % 100;  // + LVD_GEMM_V_ADMA / ADMA64 select the main-loop generation of the same geometry
  if (vb == LVD_GEMM_V_RING256W_TAIL || vb == LVD_GEMM_V_RING128x320_TAIL) rc = run_with_tail(p, stream, v);
  else rc = run_variant(p, stream, v);
  LVD_CHECK(rc != 3, "gemm: variant %d names no tile geometry", v);
  LVD_CHECK(rc != 4, "gemm: variant %d cannot take a LayerNorm-folded product (asm-DMA ring and K-split variants only)", v);
  LVD_CHECK(rc == 0, "gemm: unknown mode %d", p->mode);
  LVD_LAUNCH_CHECK();
  return 0;
}

// Workspace the split-K plans can use for this product (caller-owned, passed back in p->ws / p->ws_bytes).  Under-filled
// grids (< 512 tiles of 128x128) may split the whole product into up to 16 fp32 slabs; larger products only ever split the
// last partial round of wide tiles (run_with_tail), for which 64 MiB covers every plan.  A smaller or absent workspace is
// legal: the launchers fall back to the unsplit geometry.
extern "C" int lvdhip_gemm_workspace_bytes(const lvd_gemm_params* p, int64_t* bytes) {
  LVD_CHECK(p && bytes && p->M > 0 && p->N > 0 && p->K > 0, "gemm_workspace_bytes: bad arguments");
  if (p->act != LVD_ACT_NONE || p->K < 512) {
    *bytes = 0;
    return 0;
  }
  const long tiles = ((long)(p->M + 127) / 128) * ((p->N + 127) / 128);
  *bytes = tiles < 512 ? 16L * p->M * p->N * 4 : 64L << 20;
  return 0;
}
