#include "vu_attn_scores.h"
VU_SCORES_TU(vu_scores_bf16_plain, bf16_t, false)
