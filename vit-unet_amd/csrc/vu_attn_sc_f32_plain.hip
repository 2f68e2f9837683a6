#include "vu_attn_scores.h"
VU_SCORES_TU(vu_scores_f32_plain, float, false)
