// detect.hip — the two kernels the OWL-ViT scoring path needs besides the shared transformer kernels:
//   frames_to_patches : uint8 video frames -> resized, normalised, patchified bf16 matrix (input of the patch-embedding GEMM)
//   owl_detect_rows   : class / box heads' raw outputs -> logits, score, label and pixel box per image token
// Reference call sites: scripts/eval_owl_vit.py:70-96 (`processor(text, images)`, `model(**inputs)`,
// `processor.post_process(outputs, target_sizes)`); arithmetic in transformers 4.36.2 OwlViTImageProcessor (PIL bicubic
// resize, 1/255 rescale, CLIP mean/std) and OwlViTForObjectDetection.{class_predictor, box_predictor}.
#include "common.h"

namespace {

LVD_DEV int clip8(int acc) {
  int v = acc >> 22;  // PIL 8-bit resampling: coefficients carry 22 fractional bits, the accumulator starts at one half
  return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// One thread = one output pixel (3 channels).  Separable resampling with PIL's two-pass semantics: every tap row is first
// resampled horizontally and rounded to 8 bits, then the column of those is resampled vertically and rounded again, with
// the integer coefficient tables the host computed in double precision — bit-identical to Image.resize for uint8 RGB.
// The filter (bicubic for the OWL-ViT processor, Lanczos for the upsampler's init video) lives entirely in the tables.
__global__ void frames_to_patches_kernel(const uint8_t* __restrict__ frames, int B, int H, int W, int SH, int SW, int P,
                                         const int* __restrict__ xb, const int* __restrict__ xk, int xtaps,
                                         const int* __restrict__ yb, const int* __restrict__ yk, int ytaps,
                                         float m0, float m1, float m2, float s0, float s1, float s2,
                                         lvd_bf16* __restrict__ patches, int ld, uint8_t* __restrict__ resized) {
  const long total = (long)B * SH * SW;
  const int GH = SH / P, GW = SW / P;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int ox = (int)(i % SW), oy = (int)((i / SW) % SH), b = (int)(i / ((long)SH * SW));
    const int x0 = xb[2 * ox], xn = xb[2 * ox + 1], y0 = yb[2 * oy], yn = yb[2 * oy + 1];
    const int* kx = xk + (long)ox * xtaps;
    const int* ky = yk + (long)oy * ytaps;
    int a0 = 1 << 21, a1 = 1 << 21, a2 = 1 << 21;
    for (int ty = 0; ty < yn; ++ty) {
      const uint8_t* row = frames + (((long)b * H + (y0 + ty)) * W + x0) * 3;
      int h0 = 1 << 21, h1 = 1 << 21, h2 = 1 << 21;
      for (int tx = 0; tx < xn; ++tx) {
        const int k = kx[tx];
        h0 += row[tx * 3 + 0] * k;
        h1 += row[tx * 3 + 1] * k;
        h2 += row[tx * 3 + 2] * k;
      }
      const int k = ky[ty];
      a0 += clip8(h0) * k;
      a1 += clip8(h1) * k;
      a2 += clip8(h2) * k;
    }
    const int v0 = clip8(a0), v1 = clip8(a1), v2 = clip8(a2);
    if (resized) {
      uint8_t* r = resized + i * 3;
      r[0] = (uint8_t)v0, r[1] = (uint8_t)v1, r[2] = (uint8_t)v2;
    }
    // row = (frame, patch_y, patch_x), column = (channel, y in patch, x in patch): the flattening of Conv2d's [out, c, kh, kw]
    lvd_bf16* dst = patches + ((long)(b * GH + oy / P) * GW + ox / P) * ld + (oy % P) * P + (ox % P);
    const float inv = 1.f / 255.f;
    dst[0] = f2bf((v0 * inv - m0) / s0);
    dst[P * P] = f2bf((v1 * inv - m1) / s1);
    dst[2 * P * P] = f2bf((v2 * inv - m2) / s2);
  }
}

// One wave = one image token.  logits[q] = (<e/(|e|+1e-6), query_q> + shift) * (elu(scale) + 1); score = sigmoid(max_q),
// label = argmax_q (first maximum); box = sigmoid(raw + bias[token]) as (cx, cy, w, h) -> corners scaled to pixels.
constexpr int DET_MAX_D = 1024;
__global__ __launch_bounds__(256) void owl_detect_rows_kernel(const float* __restrict__ emb, int ld_emb, int D, const float* __restrict__ queries,
                                                              int Q, const int* __restrict__ qmask, const float* __restrict__ shsc, int ld_shsc,
                                                              const float* __restrict__ boxraw, int ld_box, const float* __restrict__ bias, int P,
                                                              float img_w, float img_h, long rows, float* __restrict__ logits,
                                                              float* __restrict__ scores, long long* __restrict__ labels, float* __restrict__ boxes) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= rows) return;
  float e[DET_MAX_D / 64];
  float ss = 0.f;
#pragma unroll
  for (int j = 0; j < DET_MAX_D / 64; ++j) {
    const int d = lane + 64 * j;
    e[j] = d < D ? emb[row * ld_emb + d] : 0.f;
    ss += e[j] * e[j];
  }
  const float rn = 1.f / (sqrtf(wave_sum(ss)) + 1e-6f);
  const float shift = shsc[row * ld_shsc], sraw = shsc[row * ld_shsc + 1];
  const float scale = (sraw > 0.f ? sraw : __expf(sraw) - 1.f) + 1.f;
  float best = -INFINITY;
  int arg = 0;
  for (int q = 0; q < Q; ++q) {
    float dot = 0.f;
#pragma unroll
    for (int j = 0; j < DET_MAX_D / 64; ++j) {
      const int d = lane + 64 * j;
      if (d < D) dot += e[j] * queries[(long)q * D + d];
    }
    float lg = (wave_sum(dot) * rn + shift) * scale;
    if (qmask && qmask[q] == 0) lg = -3.4028234663852886e38f;  // torch.finfo(float32).min for padded queries
    if (lane == 0) logits[row * Q + q] = lg;
    if (lg > best) best = lg, arg = q;
  }
  if (lane == 0) {
    scores[row] = 1.f / (1.f + __expf(-best));
    labels[row] = arg;
    const float* bb = bias + (row % P) * 4;
    float v[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) v[c] = 1.f / (1.f + __expf(-(boxraw[row * ld_box + c] + bb[c])));
    boxes[row * 4 + 0] = (v[0] - 0.5f * v[2]) * img_w;
    boxes[row * 4 + 1] = (v[1] - 0.5f * v[3]) * img_h;
    boxes[row * 4 + 2] = (v[0] + 0.5f * v[2]) * img_w;
    boxes[row * 4 + 3] = (v[1] + 0.5f * v[3]) * img_h;
  }
}

}  // namespace

extern "C" int lvdhip_frames_to_patches(const uint8_t* frames, int32_t B, int32_t H, int32_t W, int32_t SH, int32_t SW, int32_t P,
                                        const int32_t* xbounds, const int32_t* xcoef, int32_t xtaps, const int32_t* ybounds,
                                        const int32_t* ycoef, int32_t ytaps, const float* mean3, const float* std3,
                                        lvd_bf16* patches, int32_t ld, uint8_t* resized, void* stream) {
  LVD_CHECK(frames && patches && xbounds && xcoef && ybounds && ycoef && mean3 && std3, "frames_to_patches: null argument");
  LVD_CHECK(B > 0 && H > 0 && W > 0 && SH > 0 && SW > 0 && P > 0 && SH % P == 0 && SW % P == 0 && ld >= 3 * P * P && xtaps > 0 && ytaps > 0,
            "frames_to_patches: bad geometry (B=%d H=%d W=%d out=%dx%d P=%d ld=%d)", B, H, W, SH, SW, P, ld);
  const long total = (long)B * SH * SW;
  long blocks = (total + 255) / 256;
  if (blocks > 65536) blocks = 65536;
  hipLaunchKernelGGL(frames_to_patches_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, frames, B, H, W, SH, SW, P, xbounds, xcoef,
                     xtaps, ybounds, ycoef, ytaps, mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2], patches, ld, resized);
  LVD_LAUNCH_CHECK();
  return 0;
}

extern "C" int lvdhip_owl_detect_rows(const float* class_embeds, int32_t ld_embeds, int32_t D, const float* queries, int32_t Q,
                                      const int32_t* query_mask, const float* shift_scale, int32_t ld_shift_scale, const float* box_raw,
                                      int32_t ld_box, const float* box_bias, int32_t tokens_per_image, float img_w, float img_h,
                                      int64_t rows, float* logits, float* scores, int64_t* labels, float* boxes, void* stream) {
  LVD_CHECK(class_embeds && queries && shift_scale && box_raw && box_bias && logits && scores && labels && boxes, "owl_detect_rows: null argument");
  LVD_CHECK(D > 0 && D <= DET_MAX_D && Q > 0 && rows > 0 && tokens_per_image > 0 && ld_embeds >= D && ld_shift_scale >= 2 && ld_box >= 4,
            "owl_detect_rows: bad geometry (D=%d Q=%d rows=%ld)", D, Q, (long)rows);
  hipLaunchKernelGGL(owl_detect_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, class_embeds, ld_embeds, D,
                     queries, Q, query_mask, shift_scale, ld_shift_scale, box_raw, ld_box, box_bias, tokens_per_image, img_w, img_h, (long)rows,
                     logits, scores, (long long*)labels, boxes);
  LVD_LAUNCH_CHECK();
  return 0;
}
