#include "vu_attn_scores.h"
VU_SCORES_TU(vu_scores_bf16_softmax, bf16_t, true)
