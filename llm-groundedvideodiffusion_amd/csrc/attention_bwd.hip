// attention_bwd.hip — placeholder, replaced below in this round.
#include "common.h"
extern "C" int lvdhip_attention_bwd(const lvd_attn_bwd_params* p, void* stream) {
  (void)p; (void)stream;
  LVD_CHECK(false, "attention_bwd: not implemented yet");
}
