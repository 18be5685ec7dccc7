// gemm_pers.hip — persistent form of the LDS-DMA ring GEMM (gemm_ring.hip): one workgroup per CU slot walks a list of
// output tiles and the ring of K tiles runs straight THROUGH the tile boundaries.
//
// Why: the model's widest layers have short reductions (K = 320 / 640 at 138k / 35k token rows).  A 256x320 tile of a
// K = 320 linear is only 10 K-tile iterations (~3 us); with one workgroup per tile every tile pays the DMA pipeline fill
// (an L2/HBM round trip with nothing to compute) and the store tail, and the next workgroup cannot start before the
// previous one has retired.  Here the loader is simply STAGES-1 K tiles ahead of the MFMAs in a flattened
// (tile, k) iteration space: while the last K tiles of output tile j are multiplied, the first K tiles of tile j+1 are
// already landing in the ring, and they keep landing while tile j's accumulators are converted and stored.
//
// Tile order: workgroup b handles tiles b, b+G, b+2G, ... (G = grid size, a multiple of 8), each mapped through the same
// XCD-aware bijection as the one-shot kernel, so the G tiles in flight at any time are, per XCD, a contiguous run of
// (m-tile, n-tile) pairs sharing their A rows in that XCD's L2.
#include <cstdlib>
#include "gemm_tile.h"

namespace {

template <int MODE, int WM, int WN, int FM, int FN, int STAGES, int RBK, int WPS>
__global__ __launch_bounds__(64 * WM * WN, WPS) void gemm_pers_kernel(const lvd_gemm_params p, const int tiles) {
  constexpr int NW = WM * WN;
  constexpr int RCH = RBK / 8;                            // 16-byte chunks per tile row
  constexpr int RPI = 64 / RCH;                          // tile rows covered by one wave-wide glds instruction
  constexpr int BM = WM * FM * 32, BN = WN * FN * 32;
  constexpr int TILE = (BM + BN) * RCH;                  // uint4 per stage
  constexpr int AINS = BM / RPI, BINS = BN / RPI;
  constexpr int APW = AINS / NW;
  constexpr int BPW = (BINS + NW - 1) / NW;
  static_assert(AINS % NW == 0, "BM must be a multiple of RPI * waves");
  constexpr int LPS = APW + BPW;                         // glds per wave per stage (uniform -> one vmcnt immediate)
  __shared__ uint4 lds[STAGES * TILE];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave - wm * WN;
  const int l31 = lane & 31, hi = lane >> 5;
  const int cpos = lane % RCH, rsub = lane / RCH;
  auto swz = [](int row, int c) { return RCH == 4 ? (c ^ ((row >> 2) & 3)) : (c ^ ((row >> 1) & 7)); };

  const int G = gridDim.x, bid = blockIdx.x;
  const int tiles_n = (p.N + BN - 1) / BN;
  const int ntile = (tiles - bid + G - 1) / G;           // output tiles of this workgroup (>= 1: G <= tiles)
  const int nk = (p.K + RBK - 1) / RBK;
  const int q8 = tiles >> 3, r8 = tiles & 7;
  auto tile_of = [&](int j, int& tm, int& tn) {
    int t = bid + j * G;
    int xcd = t & 7, idx = t >> 3;
    int id = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
    tm = id / tiles_n;
    tn = id - tm * tiles_n;
  };

  // ---- loader state: the tile whose K tiles are being staged (runs STAGES-1 iterations ahead of the MFMAs)
  RowInfo ar[APW];
  long woff[BPW];
  bool wvalid[BPW];
  int bins[BPW];
#pragma unroll
  for (int t = 0; t < BPW; ++t) {
    int b = wave + NW * t;
    bins[t] = b < BINS ? b : BINS - 1;  // padding instruction re-stages the last rows (same data, harmless)
  }
  auto set_loader_tile = [&](int j) {
    const bool live = j < ntile;        // past the end: everything comes from the zero page (uniform vmcnt bookkeeping)
    int tm = 0, tn = 0;
    if (live) tile_of(j, tm, tn);
#pragma unroll
    for (int q = 0; q < APW; ++q) ar[q] = make_row<MODE>(p, p.m_begin + tm * BM + (wave * APW + q) * RPI + rsub, live);
#pragma unroll
    for (int t = 0; t < BPW; ++t) {
      int n = tn * BN + bins[t] * RPI + rsub;
      wvalid[t] = live && n < p.N;
      woff[t] = (long)n * p.K;
    }
  };
  int lj = 0, lk = 0;
  set_loader_tile(0);

  auto stage_one = [&](int slot, int idx) {
    uint4* A = lds + slot * TILE;
    uint4* B = A + BM * RCH;
    if (idx < APW) {
      const int q = idx;
      const int k0 = lk * RBK + swz((wave * APW + q) * RPI + rsub, cpos) * 8;
      const lvd_bf16* src = a_src<MODE>(p, ar[q], k0, p.K);
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(A + (wave * APW + q) * RPI * RCH), 16, 0, 0);
    } else {
      const int t = idx - APW;
      const int k0 = lk * RBK + swz(bins[t] * RPI + rsub, cpos) * 8;
      const lvd_bf16* src = (wvalid[t] && k0 < p.K) ? p.w + woff[t] + k0 : reinterpret_cast<const lvd_bf16*>(g_zero_page);
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(B + bins[t] * RPI * RCH), 16, 0, 0);
    }
  };
  auto advance_loader = [&]() {
    if (++lk == nk) {
      lk = 0;
      set_loader_tile(++lj);
    }
  };

  f32x16 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

#pragma unroll
  for (int s = 0; s < STAGES - 1; ++s) {
#pragma unroll
    for (int idx = 0; idx < LPS; ++idx) stage_one(s, idx);
    advance_loader();
  }

  int slot = 0, cj = 0, ck = 0;
  const int total = ntile * nk;
  for (int it = 0; it < total; ++it) {
    // the K tile in `slot` has landed once at most (STAGES-2) younger stages are outstanding.  After an epilogue the
    // stores are the youngest entries of the same in-order counter, so this wait also covers them; it never under-waits.
    wait_vmcnt<(STAGES - 2) * LPS>();
    __builtin_amdgcn_s_barrier();
    const int nslot = slot == 0 ? STAGES - 1 : slot - 1;
    constexpr int NMF = (RBK / 16) * FM * FN;
    constexpr int IVL = NMF / LPS > 0 ? NMF / LPS : 1;
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
          if (cnt % IVL == 0 && cnt / IVL < LPS) stage_one(nslot, cnt / IVL);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
        }
    }
    if (LPS > NMF) {
#pragma unroll
      for (int idx = NMF; idx < LPS; ++idx) stage_one(nslot, idx);
    }
    advance_loader();
    slot = slot + 1 == STAGES ? 0 : slot + 1;
    if (++ck == nk) {
      int tm, tn;
      tile_of(cj, tm, tn);
      ring_epilogue<FM, FN>(p, acc, p.m_begin + tm * BM + wm * FM * 32, tn * BN + wn * FN * 32, l31, hi);
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
      ck = 0;
      ++cj;
    }
  }
  wait_vmcnt<0>();
}

template <int MODE, int WM, int WN, int FM, int FN, int STAGES, int RBK, int WPS>
int launch_pers_mode(const lvd_gemm_params* p, hipStream_t s) {
  constexpr int BM = WM * FM * 32, BN = WN * FN * 32;
  auto kern = gemm_pers_kernel<MODE, WM, WN, FM, FN, STAGES, RBK, WPS>;
  static int slots = 0;  // resident workgroups on the whole device for this instantiation
  if (!slots) {
    int dev = 0, cus = 0, occ = 0;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, 64 * WM * WN, 0);
    if (occ < 1) occ = 1;
    slots = (cus * occ) & ~7;  // multiple of 8: tile t and workgroup t % G then sit on the same XCD
    if (slots < 8) slots = 8;
  }
  int tiles = ((p->M - p->m_begin + BM - 1) / BM) * ((p->N + BN - 1) / BN);
  int grid = tiles < slots ? tiles : slots;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * WM * WN), 0, s, *p, tiles);
  return 0;
}

template <int WM, int WN, int FM, int FN, int STAGES, int RBK, int WPS>
int launch_pers(const lvd_gemm_params* p, hipStream_t s) {
  switch (p->mode) {
    case LVD_A_PLAIN: return launch_pers_mode<LVD_A_PLAIN, WM, WN, FM, FN, STAGES, RBK, WPS>(p, s);
    case LVD_A_CONV3X3: return launch_pers_mode<LVD_A_CONV3X3, WM, WN, FM, FN, STAGES, RBK, WPS>(p, s);
    case LVD_A_TCONV3: return launch_pers_mode<LVD_A_TCONV3, WM, WN, FM, FN, STAGES, RBK, WPS>(p, s);
    case LVD_A_CONV3X3_T2: return launch_pers_mode<LVD_A_CONV3X3_T2, WM, WN, FM, FN, STAGES, RBK, WPS>(p, s);
    default: return 1;
  }
}

}  // namespace

// geometry: 0 = 128x128 S3 (3 WG/CU), 2 = 256x160 S3, 3 = 256x128 S3, 4 = 256x320 8 waves S3, 5 = 256x256 8 waves S3,
//           12 = 128x320 S2 (2 WG/CU), 13 = 128x256 S3 (2 WG/CU)
int lvd_gemm_pers_dispatch(const lvd_gemm_params* p, void* stream, int geometry) {
  hipStream_t s = (hipStream_t)stream;
  switch (geometry) {
    case 0: return launch_pers<2, 2, 2, 2, 3, 32, 3>(p, s);
    case 2: return launch_pers<4, 1, 2, 5, 3, 32, 2>(p, s);
    case 3: return launch_pers<4, 1, 2, 4, 3, 32, 2>(p, s);
    case 4: return launch_pers<4, 2, 2, 5, 3, 32, 2>(p, s);
    case 5: return launch_pers<4, 2, 2, 4, 3, 32, 2>(p, s);
    case 12: return launch_pers<2, 2, 2, 5, 2, 32, 2>(p, s);
    case 13: return launch_pers<2, 2, 2, 4, 3, 32, 2>(p, s);
    default: return 1;
  }
}
