/*
 * lvdhip.h — C ABI of the MI355X (gfx950) denoise-step kernel library.
 *
 * The reference (TonyLianLong/LLM-groundedVideoDiffusion) has NO native layer: every
 * op on its hot path is an ATen call made from Python.  This header therefore declares
 * the boundary the reference *would* bind if its hot path were native: one entry point
 * per ATen op-group of SURVEY.md §2.3 (K1..K18).  Each declaration cites the reference
 * call site (file:line under /root/reference) it replaces.
 *
 * Conventions
 *   - plain C types only; all tensors are caller-owned device pointers (the library never
 *     allocates, frees or synchronises); every call is asynchronous on `stream`
 *     (a hipStream_t passed as void*).
 *   - activations are "token matrices": row-major [rows, channels] bf16, rows ordered
 *     (batch, frame, y, x).  `ld*` = row stride in elements.
 *   - return value: 0 on success, non-zero on error; lvdhip_last_error() gives the text
 *     (thread-local).
 *   - bf16 storage is uint16_t; statistics / losses / gradients of the loss are fp32.
 */
#ifndef LVDHIP_H
#define LVDHIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef uint16_t lvd_bf16;

const char* lvdhip_last_error(void);
int lvdhip_version(void);

/* ------------------------------------------------------------------------------------------
 * GEMM family:  OUT[M,N] = epilogue( Aload(...)[M,K] · W[N,K]^T )
 * Replaces: nn.Linear (to_q/k/v/out, proj_in/out, GEGLU/FF, time_emb_proj, PositionNet)
 *   models/attention_processor.py:382-413, models/attention.py:355-376,
 *   models/transformer_2d.py:326,363, models/transformer_temporal.py:158,175;
 * nn.Conv2d 3x3 / 1x1 of ResnetBlock2D, Downsample2D, Upsample2D, conv_in/out
 *   models/unet_3d_condition.py:729,847, models/unet_3d_blocks.py:330-349 (diffusers resnet);
 * nn.Conv3d (3,1,1) of TemporalConvLayer  models/unet_3d_blocks.py:195-199;
 * torch.cat(dim=1) skip connection (two-source A)  models/unet_3d_blocks.py:641,736;
 * and their input-gradients (dgrad) in torch.autograd.grad  models/pipelines.py:120.
 * ------------------------------------------------------------------------------------------ */
enum {
  LVD_A_PLAIN = 0,      /* A[m,k] = X[m, k] (k < C1 from a1, else from a2) */
  LVD_A_CONV3X3 = 1,    /* implicit im2col, K = 9*Cin, tap-major; stride 1|2, optional nearest x2 upsampled source */
  LVD_A_TCONV3 = 2,     /* temporal 3-tap, K = 3*Cin, frames at row distance HW */
  LVD_A_CONV3X3_T2 = 3  /* transposed (dgrad of the stride-2 pad-1 conv): K = 9*Cin */
};
enum { LVD_ACT_NONE = 0, LVD_ACT_GEGLU = 1 };
/* tile geometries (all produce identical results up to fp32 summation order inside a K tile) */
enum {
  LVD_GEMM_V_REG32 = 1,     /* 128x128x32 register-staged, 4 workgroups/CU */
  LVD_GEMM_V_RING128 = 5,   /* 128x128x32 LDS-DMA ring (3 stages), 3 workgroups/CU */
  LVD_GEMM_V_RING256N = 9,  /* 256x160 / 256x128 LDS-DMA ring, 4 waves */
  LVD_GEMM_V_REG64 = 10,    /* 128x128x64 register-staged, 2 workgroups/CU */
  LVD_GEMM_V_RING256W = 11, /* 256x320 / 256x256 LDS-DMA ring, 8 waves in two ping-pong groups, 1 workgroup/CU */
  LVD_GEMM_V_RING256K64 = 14, /* 256x256x64 LDS-DMA double buffer, 8 waves */
  LVD_GEMM_V_RING128x320 = 17, /* 128x320x32 LDS-DMA double buffer (N = 320·k exactly), 2 workgroups/CU */
  LVD_GEMM_V_SPLITK = 20,    /* 128x128x32 ring, K split over workgroups + deterministic slab reduction (under-filled grids) */
  LVD_GEMM_V_SPLITK_WIDE = 25, /* K split on the 8-wave 256x320 / 256x256 geometries (small-M, long-K deep-level layers) */
  LVD_GEMM_V_RING256W_TAIL = 31,   /* RING256W on the rows that fill whole rounds of the 256 CUs, split-K on the remainder */
  LVD_GEMM_V_RING128x320_TAIL = 37, /* RING128x320 on whole rounds of 512 workgroup slots, split-K on the remainder */
  /* 3x3 stride-1 convolutions (optionally with the nearest-x2 upsample fused, even H/W) and temporal (3,1,1) convolutions with the
     im2col tile resident in LDS (conv_halo.hip): per 32-channel chunk the rows a 512 x 160|128 tile needs for all its taps are
     staged once (3x3: the tile's rows + one image row + one pixel of halo; temporal: 512/F pixels x all F frames) and the taps are
     shifted LDS views; 8 waves, 1 workgroup/CU.  Products the kernel cannot take (stride 2, two sources, W > 87, Cin % 32) run
     the RING256W equivalents. */
  LVD_GEMM_V_CONV_HALO = 41,
  LVD_GEMM_V_CONV_HALO_SPLITK = 45, /* channel chunks split over workgroups + deterministic slab reduction */
  LVD_GEMM_V_CONV_HALO_TAIL = 47,   /* CONV_HALO on whole rounds of the 256 CUs, CONV_HALO_SPLITK on the remaining tiles */
  /* Persistent walker (gemm_stream.hip; only as LVD_GEMM_V_STREAM + LVD_GEMM_V_ADMA = 161): ONE 8-wave workgroup per CU walks the
     256x320 / 256x256 tiles of the product, the three-slot ring of 32-deep K tiles runs through the tile boundaries, the epilogue's
     stores are never waited for and the last partial round is cut into 128-row half tiles.  Plain single-source loader, K % 32 == 0,
     K >= 128, bf16 output (bias, alpha, residual, GEGLU, LayerNorm fold); anything else runs RING256W + ADMA. */
  LVD_GEMM_V_STREAM = 61,
  /* + LVD_GEMM_V_ADMA on a RING128 / RING256N / RING256W / RING128x320 / SPLITK / SPLITK_WIDE / *_TAIL variant: the same
     geometry with its LDS-DMA issued from buffer descriptors in inline assembly, counted waits that really leave tiles in
     flight, bias row staged by the DMA engine (plain loader, K % 32 == 0; anything else runs the base variant) */
  LVD_GEMM_V_ADMA = 100,
  /* + LVD_GEMM_V_ADMA64 on RING256W / SPLITK_WIDE / RING256W_TAIL: 64-deep K tiles in a two-slot ring — every DMA instruction
     moves 8 rows x one full 128-byte line (plain loader, K % 64 == 0, K >= 128; anything else runs the + LVD_GEMM_V_ADMA form).
     Since round 4 also on RING128 / RING256N / RING128x320 / SPLITK (205 / 209 / 217 / 220): the same 4-wave loops with 64-deep tiles.
     A code that names no geometry (e.g. 141, 210) is an error, not a fallback. */
  LVD_GEMM_V_ADMA64 = 200
};

typedef struct {
  const lvd_bf16* a1;      /* first A source  [rows, lda1] */
  const lvd_bf16* a2;      /* second A source [rows, lda2] (channel concat) or NULL */
  const lvd_bf16* w;       /* [N, K] row-major, K contiguous */
  const float* bias;       /* [N] or NULL */
  const float* rowbias;    /* [M / rows_per_sample, N] or NULL (temb projection add) */
  const lvd_bf16* res;     /* [M, ldres] residual or NULL */
  void* out;               /* [M, ldc] bf16 (or fp32 if out_fp32) */
  int32_t M, N, K;
  int32_t lda1, lda2, c1;  /* c1 = channels taken from a1 (per tap); Cin = c1 + c2 */
  int32_t cin;             /* channels per tap (PLAIN: = K) */
  int32_t mode;            /* LVD_A_* */
  int32_t hin, win;        /* conv: input spatial dims (after optional upsample) */
  int32_t hout, wout;      /* conv: output spatial dims; rows m = (n, oy, ox) */
  int32_t stride;          /* conv stride 1|2 */
  int32_t upsample;        /* 1: a1 is stored at (hin/2, win/2) and nearest-upsampled on the fly */
  int32_t frames, hw;      /* tconv: frames per batch item, rows per frame */
  int32_t rows_per_sample; /* rowbias granularity */
  int32_t ldres, ldc;
  int32_t act;             /* LVD_ACT_* (GEGLU: W rows interleaved hidden/gate in blocks of 32; out has N/2 cols) */
  int32_t out_fp32;
  float alpha;             /* out = res + alpha * (acc + bias + rowbias) */
  int32_t accumulate;      /* 1: out += (bf16 read-modify-write; used for gradient accumulation) */
  int32_t variant;         /* 0 = library heuristic; >0 pins a tile geometry (LVD_GEMM_V_*), used by the host autotuner */
  int32_t ksplit;          /* LVD_GEMM_V_SPLITK only: number of K slices (0 = choose from the grid size) */
  float* ws;               /* split-K workspace, fp32 slabs [ksplit, M, N]; caller-owned */
  int64_t ws_bytes;
  int32_t m_begin;         /* rows [m_begin, M) are produced (0 = the whole product); row indices stay absolute, so a product
                              can be cut into row ranges (LVD_GEMM_V_*_TAIL variants do this internally) */
  int32_t ldrowbias;       /* row stride of rowbias in floats (0 = N): the temb projections of all ResnetBlock2Ds of a forward are ONE
                              product [B, sum of N] and each conv reads its column range of it */
  /* LayerNorm folded into a linear product (BasicTransformerBlock.norm1/2/3 -> to_q|k|v / ff.net.0, models/attention.py:113,140,153):
     with ln_mean_rstd set, A holds the UN-normalised rows x, `w` = gamma (.) W (bf16), `bias` = b + W beta, ln_colsum[n] = sum_k w[n,k]
     (fp32, of the bf16-rounded w), and OUT = alpha * ( rstd_m * (x . w^T - mean_m * ln_colsum) + bias ): the normalised
     activation is never materialised.  Plain single-source loader, no residual / accumulate / rowbias, bf16 output with N % 16 == 0;
     asm-DMA ring and K-split variants only. */
  const float* ln_mean_rstd;  /* [M, 2] fp32 (mean, rstd) per row of A (lvdhip_layernorm with y = NULL writes it), or NULL */
  const float* ln_colsum;     /* [N] fp32 */
} lvd_gemm_params;

int lvdhip_gemm(const lvd_gemm_params* p, void* stream);
/* Bytes of split-K workspace worth offering for this product (0 = none is ever used).  The caller allocates and owns it and
 * may share one buffer among all launches of a stream; passing less (or NULL) is legal and disables the K-split plans. */
int lvdhip_gemm_workspace_bytes(const lvd_gemm_params* p, int64_t* bytes);

/* ------------------------------------------------------------------------------------------
 * GroupNorm (+SiLU) over a token matrix.  A "sample" is rows_per_sample consecutive rows:
 *   2-D GroupNorm  (ResnetBlock2D.norm1/2, Transformer2DModel.norm  models/transformer_2d.py:314):
 *       rows_per_sample = H*W (statistics per frame)
 *   5-D GroupNorm  (TransformerTemporalModel.norm models/transformer_temporal.py:148-153,
 *       TemporalConvLayer norms): rows_per_sample = F*H*W (statistics across frames)
 * stats: per-(row chunk, group) partial sums (one launch); apply: every workgroup folds the chunk partials of its sample in a
 *   fixed order (a few KB from L2), forms scale = rstd*gamma, shift = beta - mean*rstd*gamma in registers and writes
 *   y = silu?( x * scale + shift ); the first workgroup of a sample also writes (mean, rstd) for the backward.  No finalize launch.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
  const lvd_bf16* x1; const lvd_bf16* x2;   /* channel concat of two sources (x2 may be NULL) */
  int32_t ld1, ld2, c1, c;                  /* c = total channels */
  int32_t rows, rows_per_sample, groups;
  float eps;
  const float* gamma; const float* beta;    /* [c] */
  float* partial;      /* out: [samples, chunks, groups, 2] fp32 (sum x, sum x^2) per row chunk and group */
  int32_t chunks;
  float* scale_shift;  /* unused by the two-stage path (kept for layout stability); may be NULL */
  float* mean_rstd;    /* single-launch variant only: out [samples, groups, 2] fp32 (the two-stage path writes it from the apply launch) */
} lvd_gn_stats_params;
int lvdhip_groupnorm_stats(const lvd_gn_stats_params* p, void* stream);

typedef struct {
  const lvd_bf16* x1; const lvd_bf16* x2;
  int32_t ld1, ld2, c1, c;
  int32_t rows, rows_per_sample;
  const float* partial;       /* [samples, chunks, groups, 2] from lvdhip_groupnorm_stats (unused by the single-launch variant) */
  int32_t silu;
  lvd_bf16* y; int32_t ldy;
  int32_t chunks, groups; float eps;
  const float* gamma; const float* beta;  /* [c] */
  float* mean_rstd;           /* out: [samples, groups, 2] fp32 (kept for backward) or NULL */
} lvd_gn_apply_params;
int lvdhip_groupnorm_apply(const lvd_gn_apply_params* p, void* stream);

/* GroupNorm backward (dgrad): given dy (grad wrt the (SiLU'd) output), x, saved mean/rstd:
 *   dx = rstd * ( g - mean_g(g) - xhat * mean_g(g * xhat) ),  g = dy * silu'(yhat) * gamma  */
typedef struct {
  const lvd_bf16* x1; const lvd_bf16* x2; int32_t ld1, ld2, c1, c;
  const lvd_bf16* dy; int32_t lddy;
  int32_t rows, rows_per_sample, groups;
  const float* gamma; const float* beta;
  const float* mean_rstd;     /* [samples, groups, 2] */
  float* partial; int32_t chunks;  /* out: [samples, chunks, groups, 2] = sums of (g, g*xhat) per row chunk and group */
  float* gsum;                /* unused (kept for layout stability); may be NULL */
  int32_t silu;
} lvd_gn_bwd_stats_params;
int lvdhip_groupnorm_bwd_stats(const lvd_gn_bwd_stats_params* p, void* stream);

typedef struct {
  const lvd_bf16* x1; const lvd_bf16* x2; int32_t ld1, ld2, c1, c;
  const lvd_bf16* dy; int32_t lddy;
  int32_t rows, rows_per_sample, groups;
  const float* gamma; const float* beta;
  const float* mean_rstd; const float* partial;  /* partial: [samples, chunks, groups, 2] from lvdhip_groupnorm_bwd_stats, folded by every workgroup */
  int32_t silu;
  lvd_bf16* dx1; lvd_bf16* dx2; int32_t lddx1, lddx2;  /* grads of the two sources */
  int32_t accumulate;         /* 1: dx += */
  int32_t chunks;             /* row chunks of `partial` (unused by the single-launch variant) */
} lvd_gn_bwd_apply_params;
int lvdhip_groupnorm_bwd_apply(const lvd_gn_bwd_apply_params* p, void* stream);

/* Single-launch variants for small samples (deep UNet levels: a tensor is a few MB and the three launches above cost three
 * launch floors): one workgroup per (sample, group) computes the statistics and applies them; the second pass over its slab
 * (rows_per_sample x c/groups elements, meant for <= 128 KB) hits L2.  Same results as stats+apply up to fp32 summation order.
 * `s->partial`, `s->chunks`, `s->scale_shift` and `p->gsum` are not used (may be NULL); `s->mean_rstd` is still written for the
 * backward.  Needs an even number of channels per group. */
int lvdhip_groupnorm_fused(const lvd_gn_stats_params* s, const lvd_gn_apply_params* a, void* stream);
int lvdhip_groupnorm_bwd_fused(const lvd_gn_bwd_apply_params* p, void* stream);
/* Single launch for the slabs in between (30-200 KB per (sample, group): the 2-D norms of the 40x72 / 20x36 levels, the 5-D norms of the
 * temporal layers on the 5x9 level, the norm over [x, skip] at 2560 channels): a 1024-thread workgroup loads its whole slab at once,
 * keeps it in registers — x is read once —, reduces and normalises.  A thread's load is the widest of 16 / 8 / 4 bytes (8 / 4 / 2
 * channels) that divides the channels per group.  Arguments as for lvdhip_groupnorm_fused (row pitches multiples of 8 elements).
 * lvdhip_groupnorm_slab_loads returns the loads per thread (>= 1) or 0 when the shape does not qualify: an even number of channels per
 * group, a group inside one source (c1 % (c/groups) == 0), and rows_per_sample <= U * (1024 / (c / groups / V)) with
 * (V channels per load, U loads) = (8, 12), (4, 8) or (2, 8). */
int lvdhip_groupnorm_slab_loads(int32_t c, int32_t c1, int32_t groups, int32_t rows_per_sample);
int lvdhip_groupnorm_slab(const lvd_gn_stats_params* s, const lvd_gn_apply_params* a, void* stream);
/* ... and its backward: x and dy of the slab in registers (at most 64 bytes of each per thread: lvdhip_groupnorm_bwd_slab_loads returns the
 * pairs of loads per thread, or 0).  Arguments as for lvdhip_groupnorm_bwd_fused. */
int lvdhip_groupnorm_bwd_slab_loads(int32_t c, int32_t c1, int32_t groups, int32_t rows_per_sample);
int lvdhip_groupnorm_bwd_slab(const lvd_gn_bwd_apply_params* p, void* stream);

/* ------------------------------------------------------------------------------------------
 * LayerNorm over channels (BasicTransformerBlock.norm1/2/3, GatedSelfAttentionDense.norm1/2
 *   models/attention.py:113,140,153,36-37), eps 1e-5, affine.  Saves (mean, rstd) per row.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
  const lvd_bf16* x; int32_t ldx;
  int32_t rows, c;
  const float* gamma; const float* beta; float eps;
  lvd_bf16* y; int32_t ldy;
  float* mean_rstd;   /* [rows, 2] or NULL */
} lvd_ln_params;
int lvdhip_layernorm(const lvd_ln_params* p, void* stream);

typedef struct {
  const lvd_bf16* x; int32_t ldx;
  const lvd_bf16* dy; int32_t lddy;
  int32_t rows, c;
  const float* gamma; const float* mean_rstd;
  lvd_bf16* dx; int32_t lddx;
  int32_t accumulate;
} lvd_ln_bwd_params;
int lvdhip_layernorm_bwd(const lvd_ln_bwd_params* p, void* stream);

/* ------------------------------------------------------------------------------------------
 * Attention, head_dim 64, flash-style (never materialises scores in HBM).
 * Replaces F.scaled_dot_product_attention  models/attention_processor.py:344-430 and the
 * baddbmm+softmax+bmm slow path  :222-258,515-536.
 * One "sample" s = (s_outer, s_inner); query i of sample s lives at token row
 *     q_os*s_outer + q_is*s_inner + q_step*i        (s_inner in [0, q_ninner))
 * and key j at kv_os*(s / kv_div) ... see fields.  This single addressing scheme covers
 *   spatial self-attn   (sample = frame, rows contiguous),
 *   temporal self-attn  (sample = (batch, pixel), rows HW apart: the (B·F,HW,C)<->(B·HW,F,C)
 *                        swap of models/transformer_temporal.py:154-156,175-182 is fused
 *                        into the loads/stores),
 *   text cross-attn     (K/V rows = 77 text tokens of the batch item),
 *   GLIGEN gated self-attn (second K/V segment = 30 grounding tokens, models/attention.py:44-57).
 * Every operand and result is moved as whole 16-byte pieces of a head's 128-byte row: all pointers 16-byte aligned, every leading
 * dimension (ldq, ldk, ldv, ldo, lddo, lddq, lddk, lddv) a multiple of 8 elements.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
  const lvd_bf16* q; int32_t ldq;       /* head h at column h*64 */
  const lvd_bf16* k; int32_t ldk;
  const lvd_bf16* v; int32_t ldv;
  const lvd_bf16* k2; const lvd_bf16* v2; int32_t ldk2, ldv2;  /* optional 2nd KV segment */
  lvd_bf16* o; int32_t ldo;
  float* lse;                           /* [samples, heads, sq] natural-log LSE or NULL */
  int32_t samples, heads, sq, skv, skv2;
  int32_t q_ninner, q_os, q_is, q_step;       /* query row addressing */
  int32_t kv_ninner, kv_os, kv_is, kv_step;   /* key/value row addressing (segment 1) */
  int32_t kv2_ninner, kv2_os, kv2_is, kv2_step;
  float scale;
  int32_t causal;                       /* 1: key j > query i is masked (CLIP text encoder); forward only, one-wave kernel */
} lvd_attn_params;
int lvdhip_attention_fwd(const lvd_attn_params* p, void* stream);

/* backward: dq (always), dk/dv (if dk != NULL).  delta = rowsum(do*o) is computed inside. */
typedef struct {
  lvd_attn_params f;                    /* forward description (o, lse must be valid) */
  const lvd_bf16* d_o; int32_t lddo;
  lvd_bf16* dq; int32_t lddq;
  lvd_bf16* dk; int32_t lddk;           /* segment-1 grads; NULL => skip (cross-attn: text has no grad) */
  lvd_bf16* dv; int32_t lddv;
  float* delta;                         /* workspace [samples, heads, sq] */
} lvd_attn_bwd_params;
int lvdhip_attention_bwd(const lvd_attn_bwd_params* p, void* stream);

/* ------------------------------------------------------------------------------------------
 * Cross-attention-energy guidance loss, fused forward+backward.
 * Replaces AttnProcessor slow path that materialises + saves probs
 *   (models/attention_processor.py:515-586) and utils/guidance.py:160-574
 *   (add_ca_loss_per_attn_map_to_loss / compute_ca_lossv3) and the autograd backward of both.
 * Phase 1: probs of the object tokens only   A[f,h,t,p]   (+ row LSE)
 * Phase 2: per (frame, head, object, token): top-k energy + centre-of-mass terms,
 *          loss partials and dA
 * Phase 3: softmax backward over the 77 keys and dQ = scale * dS * K
 * ------------------------------------------------------------------------------------------ */
typedef struct {
  const lvd_bf16* q; int32_t ldq;       /* [frames*P, heads*64] queries of the keyed layer */
  const lvd_bf16* k; int32_t ldk;       /* [ntext(77), heads*64] text keys */
  int32_t frames, heads, P, ntext;
  float scale;
  const int32_t* tok_ids;               /* [ntok] text-token index of every (object, token) pair */
  int32_t ntok;
  float* probs;                         /* out [frames, heads, ntok, P] fp32 */
  float* lse;                           /* out [frames, heads, P] */
} lvd_ca_probs_params;
int lvdhip_ca_probs(const lvd_ca_probs_params* p, void* stream);

/* The whole probability map softmax(scale * Q K^T) over ALL text positions, as AttnProcessor's slow path materialises and saves it
 * (models/attention_processor.py:515-552, Attention.get_attention_scores :222-258; stored at :553-586 as (batch, heads, HW, tokens)).
 * Not used by the guidance loss (which keeps only the object-token columns); used by the plug-in's `return_attntion_probs` /
 * `save_attn_to_dict` / `attn_process_fn` branch and by `return_saved_attn`.  Sample s reads the text keys of block s / samples_per_key
 * (1: every sample has its own keys, as the processor sees them; F: the F frames of a video share their prompt's keys). */
typedef struct {
  const lvd_bf16* q; int32_t ldq;       /* [samples*P, heads*64] */
  const lvd_bf16* k; int32_t ldk;       /* [(samples / samples_per_key) * ntext, heads*64] */
  int32_t samples, heads, P, ntext;     /* ntext <= 96 */
  int32_t samples_per_key;
  float scale;
  float* probs;                         /* out [samples, heads, P, ntext] fp32 */
  const float* key_bias;                /* optional [samples, ld_key_bias]: additive score bias per (sample, text position) — the (batch, 1, T)
                                           attention_mask Transformer2DModel builds from encoder_attention_mask (models/transformer_2d.py:303-307:
                                           (1 - mask) * -10000) and AttnProcessor adds in get_attention_scores (:222-258); NULL = none */
  int32_t ld_key_bias;
} lvd_ca_probs_full_params;
int lvdhip_ca_probs_full(const lvd_ca_probs_full_params* p, void* stream);

/* hidden = bmm(attention_probs, value) of the same slow path (models/attention_processor.py:549) for probabilities that come from memory —
 * i.e. after the caller's `attn_process_fn` rewrote them (:537-548).  out[(s*P + q), h*64 + d] = sum_t probs[s,h,q,t] * v[key block of s][t, h*64 + d]. */
typedef struct {
  const float* probs;                   /* [samples, heads, P, ntext] fp32 */
  const lvd_bf16* v; int32_t ldv;       /* [(samples / samples_per_key) * ntext, heads*64]; 16-byte aligned rows */
  int32_t samples, heads, P, ntext;     /* ntext <= 256 */
  int32_t samples_per_key;
  lvd_bf16* out; int32_t ldo;           /* [samples*P, heads*64]; 16-byte aligned rows */
} lvd_ca_apply_probs_params;
int lvdhip_ca_apply_probs(const lvd_ca_apply_probs_params* p, void* stream);

typedef struct {
  const float* probs;                   /* [frames, heads, ntok, P] */
  float* dprobs;                        /* out, same shape: d(loss)/d(probs), already scaled by grad_scale */
  int32_t frames, heads, P, ntok, H, W; /* P = H*W */
  const int32_t* tok_obj;               /* [ntok] object index of each token column */
  const int32_t* boxes;                 /* [nobj, frames, 6] = (x_min,y_min,x_max,y_max) at (H,W) via scale_proportion
                                           (utils/utils.py:82-103), then k_fg, k_bg (utils/guidance.py:328-337) */
  const float* tok_weight;              /* [ntok] = 1/|T_o| (utils/guidance.py:524) */
  int32_t nobj;
  float fg_weight, bg_weight, com_loss_scale;
  float grad_scale;                     /* loss_scale / (nobj * nkeys)  (utils/guidance.py:569-572, pipelines.py:101-112) */
  float* loss_partial;                  /* out [frames*heads*ntok]: per-(frame,head,token) loss terms, un-scaled */
  float* com_ws;                        /* workspace [frames, heads, ntok, 4] = (sum, com_y, com_x, -) */
  /* optional terms of add_ca_loss_per_attn_map_to_loss (all off = the max-based top-k energy of the entry points' defaults) */
  int32_t use_ratio_loss;               /* energy form: 0 = max-based top-k (default, :346-353); 1 = ratio-based (utils/guidance.py:312-323:
                                           (1 - sum(A*mask)/(sum(A)+eps))^2, mean over heads); 2 = CE / NLL over the same top-k sets (:363-399:
                                           A clamped to [eps, 1-eps], fg = mean(-log topk), bg = -log(1 - mean topk), summed over heads) */
  float ratio_eps;                      /* `eps` of the ratio and CE forms; 1e-2 in the reference */
  float attn_sync_weight;               /* :401-430: w * mean over the NEXT frame's box of (A_f - A_f+1)^2, summed over heads */
  float boxdiff_loss_scale;             /* :433-465: BoxDiff corner constraint on the row / column maxima */
  int32_t boxdiff_normed;               /* 1: mean over (heads, W|H); 0: sum */
  int32_t boxdiff_L;                    /* corner half-width (1 in the reference) */
} lvd_ca_select_params;
int lvdhip_ca_select(const lvd_ca_select_params* p, void* stream);

typedef struct {
  const lvd_bf16* q; int32_t ldq;
  const lvd_bf16* k; int32_t ldk;
  int32_t frames, heads, P, ntext;
  float scale;
  const int32_t* tok_ids; int32_t ntok;
  const float* probs; const float* dprobs; const float* lse;
  lvd_bf16* dq; int32_t lddq;           /* out [frames*P, heads*64]; 16-byte aligned, lddq % 8 == 0 (rows are written as 16-byte stores) */
  /* layouts with more object tokens than one launch holds run in chunks of tokens; dQ is linear in them and is summed in fp32:
     acc_mode 0: dq = value (one launch, acc32 unused)   1: acc32 = value   2: acc32 += value   3: dq = bf16(acc32 + value) */
  float* acc32; int32_t ldacc; int32_t acc_mode;
} lvd_ca_dq_params;
int lvdhip_ca_dq(const lvd_ca_dq_params* p, void* stream);
/* All keys of one guidance iteration (utils/guidance.py:529-574 loops over guidance_attn_keys) in ONE launch per stage instead of one per
 * key: `keys` is a host array of `nkeys` <= LVD_CA_MAX_KEYS parameter blocks, each exactly what the single-key entry point takes (its own
 * q / k / P / heads / buffers); the grid is flat over the keys' workgroups.  Same results, bit for bit, as nkeys single-key calls (which
 * are the nkeys = 1 case of the same kernels).  Prompts of up to 96 text positions take the staged-key bodies (four query tiles per
 * workgroup, keys read once into LDS); longer prompts the general one-wave bodies. */
#define LVD_CA_MAX_KEYS 8
int lvdhip_ca_probs_multi(const lvd_ca_probs_params* keys, int32_t nkeys, void* stream);
int lvdhip_ca_select_multi(const lvd_ca_select_params* keys, int32_t nkeys, void* stream);
int lvdhip_ca_dq_multi(const lvd_ca_dq_params* keys, int32_t nkeys, void* stream);

/* Map-level options of add_ca_loss_per_attn_map_to_loss (utils/guidance.py:209-226), on WHOLE maps — fp32 [rows = frames * heads, P, T] as
 * lvdhip_ca_probs_full writes them — because both spread the energy's gradient over text tokens that are not object tokens:
 *   smooth_attn  (:209-220): F.pad(map, (1,1,1,1), "reflect") then a 3x3 Gaussian (utils/attn.py GaussianSmoothing, kernel 3, sigma 0.5) over the
 *                (position, token) plane of every (frame, head): lvdhip_ca_map_smooth with w9 = the normalised kernel, row-major [position tap][token tap];
 *                adjoint = 1 applies the transpose of that linear map (the backward);
 *   attn_renorm  (:222-226): softmax over tokens [tok_lo, tok_lo + tok_n) = [1, num_tokens - 1) of renorm_scale * map; the output's column u is
 *                token tok_lo + u (columns from tok_n on are written as 0); backward = 1 takes the gradient w.r.t. that output (same layout) and
 *                writes the gradient w.r.t. the input map (0 outside the token range);
 *   gather / scatter: the object-token columns in the [frames, heads, ntok, P] layout lvdhip_ca_select reads, and their gradient back into a map;
 *   softmax_bwd: dS = scale * A o (dA - rowsum(A o dA)), the backward of softmax(scale * Q K^T); dQ = lvdhip_ca_apply_probs(dS, K). */
int lvdhip_ca_map_smooth(const float* in, float* out, int64_t rows, int32_t P, int32_t T, const float* w9, int32_t adjoint, void* stream);
int lvdhip_ca_map_renorm(const float* in, const float* dout, float* out, int64_t rows, int32_t P, int32_t T, int32_t tok_lo, int32_t tok_n,
                         float renorm_scale, int32_t backward, void* stream);
int lvdhip_ca_map_gather_cols(const float* map, const int32_t* cols, int32_t ncols, float* out, int64_t rows, int32_t P, int32_t T, void* stream);
int lvdhip_ca_map_scatter_cols(const float* dcols, const int32_t* cols, int32_t ncols, float* dmap, int64_t rows, int32_t P, int32_t T, void* stream);
int lvdhip_ca_map_softmax_bwd(const float* probs, const float* dprobs, float* ds, int64_t rows, int32_t P, int32_t T, float scale, void* stream);

/* ------------------------------------------------------------------------------------------
 * Element-wise / layout kernels.
 * ------------------------------------------------------------------------------------------ */
/* (B,4,F,h,w) fp32 latents -> token matrix [B*F*h*w, cpad] bf16 (channels zero-padded);
 * replaces the permute+reshape of models/unet_3d_condition.py:726-728 */
int lvdhip_latents_to_tokens(const float* latents, lvd_bf16* tokens, int32_t B, int32_t C, int32_t F,
                             int32_t HW, int32_t cpad, float scale, void* stream);
/* token matrix fp32 [B*F*HW, ld] -> (B,C,F,h,w) fp32; models/unet_3d_condition.py:850-854 */
int lvdhip_tokens_to_latents(const float* tokens, int32_t ld, float* latents, int32_t B, int32_t C,
                             int32_t F, int32_t HW, void* stream);
/* dgrad of latents_to_tokens: token grads bf16 [rows, ld] -> (B,C,F,h,w) fp32 */
int lvdhip_tokens_grad_to_latents(const lvd_bf16* tokens, int32_t ld, float* latents, int32_t B,
                                  int32_t C, int32_t F, int32_t HW, float scale, void* stream);
/* y = a + b (bf16) ; gradient accumulation and residual adds */
int lvdhip_add(const lvd_bf16* a, int32_t lda, const lvd_bf16* b, int32_t ldb, lvd_bf16* y, int32_t ldy,
               int32_t rows, int32_t c, void* stream);
/* Temporal (3,1,1) conv at small M as an N-expanded plain product + combine (models/unet_3d_blocks.py:195-199 via diffusers' TemporalConvLayer):
 * y [rows, 3n] fp32 = x . [W_0; W_1; W_2]^T from lvdhip_gemm; out[m] = bias + res[m] + y[m - hw, 0:n] + y[m, n:2n] + y[m + hw, 2n:3n] with the
 * out-of-clip taps dropped (frame of row m = (m / hw) % frames). */
int lvdhip_tconv_combine(const float* y, int32_t ldy, const float* bias, const lvd_bf16* res, int32_t ldres, lvd_bf16* out, int32_t ldo,
                         int32_t rows, int32_t n, int32_t frames, int32_t hw, int32_t accumulate, void* stream);
/* GEGLU pieces for the recorded (guidance) pass: pre = [hidden | gate] interleaved as the GEMM emits them */
int lvdhip_geglu_fwd(const lvd_bf16* pre, int32_t ldp, lvd_bf16* y, int32_t ldy, int32_t rows, int32_t n_out,
                     void* stream);
int lvdhip_geglu_bwd(const lvd_bf16* pre, int32_t ldp, const lvd_bf16* dy, int32_t lddy, lvd_bf16* dpre,
                     int32_t lddp, int32_t rows, int32_t n_out, void* stream);
/* SiLU backward is folded into groupnorm_bwd; nearest-upsample backward = 2x2 sum */
int lvdhip_upsample2x_bwd(const lvd_bf16* dy, lvd_bf16* dx, int32_t n, int32_t h, int32_t w, int32_t c,
                          int32_t accumulate, void* stream);
/* sinusoidal timestep embedding (diffusers Timesteps(flip_sin_to_cos=True, shift 0)) -> bf16 [n, dim] */
int lvdhip_timestep_embedding(const float* t, lvd_bf16* out, int32_t n, int32_t dim, void* stream);
/* gelu on a bf16 matrix: mode 0 = exact erf GELU, 1 = quick_gelu x*sigmoid(1.702x) (CLIP text MLP) */
int lvdhip_gelu(const lvd_bf16* x, lvd_bf16* y, int64_t n, int32_t mode, void* stream);
/* silu on a small bf16 matrix (temb) */
int lvdhip_silu(const lvd_bf16* x, lvd_bf16* y, int64_t n, void* stream);
/* CFG combine + DPM-Solver++(2M) update (models/controllable_pipeline_text_to_video_synth.py:926-950):
 *   eps = eu + s*(ec-eu);  x0 = (x - sigma_t*eps)/alpha_t;
 *   x' = c_x*x + c_0*x0 + c_1*x0_prev ; x0_prev <- x0       (all fp32, (B,4,F,h,w) layout) */
int lvdhip_cfg_dpm_step(const float* eps_uncond, const float* eps_cond, float guidance_scale, float* x,
                        float* x0_prev, float alpha_t, float sigma_t, float c_x, float c_0, float c_1,
                        int64_t n, void* stream);
/* latents -= scale * grad   (models/pipelines.py:124-132) */
int lvdhip_axpy(float* x, const float* g, float scale, int64_t n, void* stream);
/* deterministic sum of n floats -> out[0] (times scale) */
int lvdhip_reduce_sum(const float* x, int64_t n, float scale, float* out, void* stream);

/* ------------------------------------------------------------------------------------------
 * VAE decode + tensor2vid, the step after the denoising loop (SURVEY §8f row 1;
 * models/controllable_pipeline_text_to_video_synth.py:374-400 decode_latents, :66-88 tensor2vid; arithmetic in
 * diffusers 0.27.2 AutoencoderKL.decode).  The decoder is built from the GEMM / GroupNorm entry points above; these
 * two cover what is left: the single-head (dim 512) attention of the mid block computes its scores with lvdhip_gemm
 * (fp32 out) and normalises them here; the last one is VaeImageProcessor.postprocess.
 * ------------------------------------------------------------------------------------------ */
/* y[r, :cols] = softmax(x[r, :cols]) ; fp32 in, bf16 out (rows wider than 4096 take a block-per-row kernel) */
int lvdhip_softmax_rows(const float* x, int32_t ldx, lvd_bf16* y, int32_t ldy, int32_t rows, int32_t cols, void* stream);
/* tokens [(f,y,x), ld>=4] bf16 (channels 0..2 = RGB in [-1,1]) -> video fp32 [(f,y,x), 3] = clamp(x/2+0.5, 0, 1) */
int lvdhip_tokens_to_video(const lvd_bf16* tokens, int32_t ld, float* video, int64_t rows, void* stream);

/* ------------------------------------------------------------------------------------------
 * OWL-ViT benchmark scoring (scripts/eval_owl_vit.py:70-96).  The ViT / text towers and the class /
 * box heads run on the GEMM, LayerNorm, attention and GELU entry points above; these two are the ends of the detector.
 * ------------------------------------------------------------------------------------------ */
/* `processor(images=...)` (transformers 4.36.2 OwlViTImageProcessor) and `Image.resize` + `preprocess_video` of the upsampler
 * (scripts/upsample.py:15-28): frames uint8 [B,H,W,3] -> PIL-exact resize to SH x SW (two 8-bit passes; xbounds/ybounds = (first
 * tap, taps) per output coordinate, x/ycoef = [out, taps] coefficients with 22 fractional bits, computed by the caller for the
 * filter it wants: bicubic, Lanczos), 1/255 rescale, (v-mean)/std, written as the patch matrix [B*(SH/P)*(SW/P), ld >= 3*P*P]
 * bf16 with columns (channel, y, x) = Conv2d weight order (P = 1: a token matrix, columns >= 3 are left untouched).
 * `resized` (uint8 [B,SH,SW,3]) is optional. */
int lvdhip_frames_to_patches(const uint8_t* frames, int32_t B, int32_t H, int32_t W, int32_t SH, int32_t SW, int32_t P,
                             const int32_t* xbounds, const int32_t* xcoef, int32_t xtaps, const int32_t* ybounds,
                             const int32_t* ycoef, int32_t ytaps, const float* mean3, const float* std3,
                             lvd_bf16* patches, int32_t ld, uint8_t* resized, void* stream);
/* OwlViTClassPredictionHead / box_predictor tails + `processor.post_process`: per image token r
 *   logits[r,q] = (<e_r/(|e_r|+1e-6), queries_q> + shift_r) * (elu(scale_r)+1)   (finfo.min where query_mask[q]==0)
 *   scores[r] = sigmoid(max_q logits), labels[r] = argmax_q, boxes[r] = corners(sigmoid(box_raw_r + box_bias[r % tokens]))
 *   scaled by (img_w, img_h, img_w, img_h).  class_embeds fp32 [rows, ld>=D<=1024]; queries fp32 [Q,D] (already unit
 *   length + eps as the head applies it); shift_scale fp32 [rows, ld>=2] = (shift, raw scale); host arrays mean3/std3. */
int lvdhip_owl_detect_rows(const float* class_embeds, int32_t ld_embeds, int32_t D, const float* queries, int32_t Q,
                           const int32_t* query_mask, const float* shift_scale, int32_t ld_shift_scale, const float* box_raw,
                           int32_t ld_box, const float* box_bias, int32_t tokens_per_image, float img_w, float img_h,
                           int64_t rows, float* logits, float* scores, int64_t* labels, float* boxes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LVDHIP_H */
