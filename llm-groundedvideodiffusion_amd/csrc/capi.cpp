// capi.cpp — host-side plumbing of the C ABI (error string, version).
#include <cstdarg>
#include <cstdio>
#include "../../include/lvdhip.h"

static thread_local char g_err[512] = "";

void lvd_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* lvdhip_last_error(void) { return g_err; }
extern "C" int lvdhip_version(void) { return 108; }  // 108: lvd_ca_probs_full_params.key_bias (cross-attention attention_mask of the processor); 107: lvdhip_ca_map_* (smooth_attn / attn_renorm on whole maps); 106: CE form of the energy (use_ratio_loss = 2), lvdhip_ca_apply_probs; 105: lvdhip_ca_probs_full (the whole probability map of the processor's saved-probabilities branch); 104: LVD_GEMM_V_STREAM (gemm_stream.hip), a LayerNorm-folded product needs its bias row, ca_dq checks the 16-byte dQ rows; 103: lvdhip_groupnorm_slab(_loads); 102: GroupNorm apply kernels fold the statistics partials (lvd_gn_apply_params / lvd_gn_bwd_apply_params grew); 101: ldrowbias, acc_mode
