// Host-side executor: parameter table, workspace layout, and the forward / backward launch
// sequences of ReAttention (model.py:150-164), SkipConnection (:244-259),
// ReAttentionTransformerEncoder (:201-207) and HViT_UNet.forward (:372-435), plus the
// extern "C" boundary declared in include/vit_unet_amd.h.  No device allocation happens here:
// every buffer is carved from the caller's workspace by a deterministic bump layout that the
// forward and the backward both recompute from (config, B).
#include <stdio.h>
#include <string.h>
#include <algorithm>
#include <vector>
#include <string>

#include "../../include/vit_unet_amd.h"
#include <stdlib.h>
#include "vu_kernels.h"
#include "vu_flash.h"

const char* vu_get_error();

#define VU_TRY(expr)            \
  do {                          \
    int _rc = (expr);           \
    if (_rc != VU_OK) return _rc; \
  } while (0)

// vu_ff2.hip: the pair as ONE kernel per direction where the shape is covered (level 2: D = 192, hidden = 32, no linear dropout)
int vu_ff2_forward_try(int dtype, const void* x, const void* w1, const float* b1, const void* w2, const float* b2, void* hpre, void* hact,
                       void* y, const void* resid, long long rows, int Din, int hid, hipStream_t st);
int vu_ff2_backward_try(int dtype, const void* dy, const void* w1, const void* w2, const void* hpre, void* gh, void* dx, const void* addend,
                        long long rows, int Din, int hid, hipStream_t st);

namespace {

inline size_t esize(int dtype) { return dtype == 0 ? 4 : 2; }
inline int round_up(int x, int a) { return (x + a - 1) / a * a; }

struct Bump {
  char* base; size_t off;
  // probability cache of the recompute attention (carve_attn): carved only for a TRAINING workspace, and only while the sum over
  // the modules (in carve order) stays within the process's budget (vu_set_flash_pcache_budget); a module without it recomputes
  bool pc_on = true; size_t pc_left = ~(size_t)0; size_t pc_total = 0;
  void* take(size_t bytes) {
    off = vu_align_up(off, 256);
    void* p = base ? base + off : nullptr;
    off += bytes;
    return p;
  }
  float* takef(size_t n) { return (float*)take(n * sizeof(float)); }
};

// ---------------------------------------------------------------------------------------------
// parameter plan
// ---------------------------------------------------------------------------------------------
struct AttnP { long long mixw, mixb, bng, bnb, wq, wk, wv, projw, projb; int bn; };
struct BlockP { AttnP at; long long ln1w, ln1b, ln2w, ln2b, w1, b1, w2, b2; int level; };
struct Level { int N, D, hid, s, ld; };
struct Plan {
  vu_config cfg;
  std::vector<Level> lv;
  long long pos = -1, outw = -1, outb = -1, total = 0;
  std::vector<BlockP> enc, bot, dec;
  std::vector<AttnP> skip;           // skip[j] merges at level depth-j-1
  std::vector<vu_param_entry> table;
  int nattn = 0;
};

int validate(const vu_config& c) {
  VU_REQUIRE(c.depth >= 0 && c.depth <= 6 && c.depth_te >= 1 && c.size_bottleneck >= 0, "config: bad depth / depth_te / size_bottleneck");
  VU_REQUIRE(c.patch_size > 0 && c.patch_size % (1 << c.depth) == 0, "Depth must be adjusted, final patch size is incompatible.");
  VU_REQUIRE(c.patch_size / (1 << c.depth) >= 4, "Depth must be adjusted, final patch size is too small (lower than 4).");
  VU_REQUIRE(c.im_size > 0 && c.im_size % c.patch_size == 0, "Patch size is not compatible with image size.");
  VU_REQUIRE((c.patch_size / (1 << c.depth)) % 4 == 0, "final patch size must be a multiple of 4 for the HIP path");
  VU_REQUIRE(c.num_channels >= 1 && c.num_channels <= 4, "num_channels must be 1..4");
  VU_REQUIRE(c.num_heads == 1 || c.num_heads == 2 || c.num_heads == 4 || c.num_heads == 8, "num_heads must be 1, 2, 4 or 8");
  const int dl = c.num_channels * (c.patch_size >> c.depth) * (c.patch_size >> c.depth);
  VU_REQUIRE(dl % c.num_heads == 0, "num_heads must divide the projection size at every level");
  VU_REQUIRE((c.hidden_dim >> c.depth) >= 1, "hidden_dim too small for depth");
  VU_REQUIRE(c.attn_drop >= 0.f && c.attn_drop < 1.f && c.proj_drop >= 0.f && c.proj_drop < 1.f && c.linear_drop >= 0.f && c.linear_drop < 1.f,
             "dropout must be in [0,1)");
  VU_REQUIRE(c.dtype == 0 || c.dtype == 1, "dtype must be 0 (fp32) or 1 (bf16)");
  VU_REQUIRE(c.attn_operands == 0 || c.attn_operands == 1, "attn_operands must be 0 (storage dtype) or 1 (OCP e4m3)");
  return VU_OK;
}

int build_plan(const vu_config& c, Plan& pl) {
  VU_TRY(validate(c));
  pl.cfg = c;
  const int n0 = (c.im_size / c.patch_size) * (c.im_size / c.patch_size);
  const int d0 = c.num_channels * c.patch_size * c.patch_size;
  for (int l = 0; l <= c.depth; ++l) {
    Level L;
    L.N = n0 << (2 * l); L.D = d0 >> (2 * l); L.hid = c.hidden_dim >> l; L.s = c.patch_size >> l;
    L.ld = round_up(L.N, 8);
    pl.lv.push_back(L);
  }
  long long off = 0;
  const int H = c.num_heads, C = c.num_channels;
  auto add = [&](const std::string& name, int ndim, int s0, int s1, int s2, int s3, int bn = -1) {
    vu_param_entry e;
    memset(&e, 0, sizeof(e));
    snprintf(e.name, sizeof(e.name), "%s", name.c_str());
    e.offset = off; e.ndim = ndim; e.shape[0] = s0; e.shape[1] = s1; e.shape[2] = s2; e.shape[3] = s3;
    e.bn_index = bn;
    long long n = 1;
    for (int i = 0; i < ndim; ++i) n *= e.shape[i];
    pl.table.push_back(e);
    const long long o = off;
    off += (n + 7) / 8 * 8;
    return o;
  };
  auto add_attn = [&](const std::string& pre, int D) {
    AttnP a;
    a.bn = pl.nattn++;
    a.mixw = add(pre + "reatten_matrix.weight", 4, H, H, 1, 1);
    a.mixb = add(pre + "reatten_matrix.bias", 1, H, 0, 0, 0);
    a.bng = add(pre + "var_norm.weight", 1, H, 0, 0, 0, a.bn);
    a.bnb = add(pre + "var_norm.bias", 1, H, 0, 0, 0);
    a.wq = add(pre + "qconv2d.weight", 4, C, C, 3, 3);
    a.wk = add(pre + "kconv2d.weight", 4, C, C, 3, 3);
    a.wv = add(pre + "vconv2d.weight", 4, C, C, 3, 3);
    a.projw = add(pre + "proj.weight", 2, D, D, 0, 0);
    a.projb = add(pre + "proj.bias", 1, D, 0, 0, 0);
    return a;
  };
  auto add_block = [&](const std::string& pre, int level) {
    const Level& L = pl.lv[level];
    BlockP b;
    b.level = level;
    b.at = add_attn(pre + "ReAttn.", L.D);
    b.ln1w = add(pre + "LN1.weight", 2, L.N, L.D, 0, 0);
    b.ln1b = add(pre + "LN1.bias", 2, L.N, L.D, 0, 0);
    b.ln2w = add(pre + "LN2.weight", 2, L.N, L.D, 0, 0);
    b.ln2b = add(pre + "LN2.bias", 2, L.N, L.D, 0, 0);
    b.w1 = add(pre + "FeedForward.net.0.weight", 2, L.hid, L.D, 0, 0);
    b.b1 = add(pre + "FeedForward.net.0.bias", 1, L.hid, 0, 0, 0);
    b.w2 = add(pre + "FeedForward.net.3.weight", 2, L.D, L.hid, 0, 0);
    b.b2 = add(pre + "FeedForward.net.3.bias", 1, L.D, 0, 0, 0);
    return b;
  };
  pl.pos = add("PE.position_embedding.weight", 2, pl.lv[0].N, pl.lv[0].D, 0, 0);
  int idx = 0;
  for (int l = 0; l < c.depth; ++l)
    for (int t = 0; t < c.depth_te; ++t) pl.enc.push_back(add_block("Encoders." + std::to_string(idx++) + ".", l));
  for (int t = 0; t < c.size_bottleneck; ++t) pl.bot.push_back(add_block("BottleNeck." + std::to_string(t) + ".", c.depth));
  idx = 0;
  for (int l = 0; l < c.depth; ++l)
    for (int t = 0; t < c.depth_te; ++t) pl.dec.push_back(add_block("Decoders." + std::to_string(idx++) + ".", c.depth - l));
  for (int l = 0; l < c.depth; ++l) pl.skip.push_back(add_attn("SkipConnections." + std::to_string(l) + ".", pl.lv[c.depth - l - 1].D));
  if (c.out_conv) {
    pl.outw = add("conv2d.weight", 4, C, C, 3, 3);
    pl.outb = add("conv2d.bias", 1, C, 0, 0, 0);
  }
  pl.total = off;
  return VU_OK;
}

// ---------------------------------------------------------------------------------------------
// ReAttention / SkipConnection
// ---------------------------------------------------------------------------------------------
struct AttnBuf { void *q, *k, *v, *Ps, *Ah, *O; float* stats; float *lse2, *rinv, *delta, *pk; void *qp = nullptr, *kp = nullptr, *vp = nullptr, *Op = nullptr; float* rinvb = nullptr; void* pc = nullptr; };
struct AttnScratch { void *dO, *dq, *dk, *dv, *dA; float* partials; int nblocks; void* pad = nullptr; };

struct AttnDims { int dtype, B, N, D, H, C, s, ld; int centered = 0; int flash = 0; };
// non-materialising form (vu_flash.hip): the maps are recomputed from q, k (and v, dO) in every pass
// flash: 0 = never, 1 = wherever the shape is covered, 2 = covered AND the launch fills enough of the chip (vu_flash_pays)
// Head dims the recompute kernels do not take (they want a multiple of 8) run on ZERO-PADDED copies of q, k, v (and dO): Lite's
// finest level has 4 heads of 12 features (N = 3136, D = 48) and runs as d = 16.  The pad features add 0 to every logit and meet
// zero columns of v / dO, so O, dq, dk, dv of the real features are what the unpadded arithmetic gives and the pad features of
// the results are 0 (dropped on the way back); the scale stays 12^-0.5.  Four small copies forward, seven backward (~10 MB each
// at 32 images) against map passes of 2.5 GB each in the materialised form.
inline int flash_dh(const AttnDims& d) { const int dh = d.D / d.H; return (d.H == 4 && dh == 12) ? 16 : dh; }
inline int flash_D(const AttnDims& d) { return flash_dh(d) * d.H; }
inline bool flash_padded(const AttnDims& d) { return flash_D(d) != d.D; }
inline bool flash_on(const AttnDims& d) {
  return d.flash && vu_flash_ok(d.dtype, d.B, d.N, flash_D(d), d.H) && (d.flash == 1 || vu_flash_pays(d.B, d.N));
}
// Which attention form runs is a PROCESS-LEVEL setting (vu_set_attn_form, include/vit_unet_amd.h): -1 = per level by the
// fill rule (model path) / materialised (stand-alone op), 0 = never the recompute form, 1 = wherever the shape is covered.
// The environment (VU_ATTN_FLASH, VU_ATTN_CENTERED) is read ONCE, as the initial value - never inside a launch path, so a
// forward and its backward cannot disagree on the workspace layout because somebody changed the environment in between.
struct AttnForm { int flash, centered; };
inline AttnForm& attn_form() {
  static AttnForm f = [] {
    const char* e = getenv("VU_ATTN_FLASH");
    return AttnForm{!e ? -1 : (e[0] == '0' ? 0 : 1), getenv("VU_ATTN_CENTERED") ? 1 : 0};
  }();
  return f;
}
inline int flash_switch() { const int f = attn_form().flash; return f < 0 ? 2 : f; }
// stand-alone op (vu_attn_forward / vu_attn_backward are separate calls that must agree on the form, and the forward may
// be asked for the map): the recompute form only when asked for (form 1)
inline int flash_switch_op() { return attn_form().flash == 1; }
// centred-map form (model path only; the stand-alone attention op returns the normalised map itself): the mixed map
// is stored centred, BatchNorm's affine part is applied inside the two products that consume it.  Needs the MFMA mix
// kernel and the streaming product kernels to cover the shape.
// Round 6: short rows too (ld <= 256: mix_center_small_kernel; Base / Large level 1 and level 0, whose 384-wide heads the streaming
// product kernels take as four 96-wide slices).  VU_ATTN_CENTERED_SMALL=0 keeps the round-1 pair (mix_stats + mix_apply) there.
inline bool centered_small() { static const bool v = [] { const char* e = getenv("VU_ATTN_CENTERED_SMALL"); return !(e && e[0] == '0'); }(); return v; }
inline bool centered_ok(const AttnDims& d) {
  const int dh = d.D / d.H;
  if (!(d.centered && d.dtype == 1 && d.H == 8 && d.ld <= 1024 && d.ld % 8 == 0)) return false;
  if (d.ld > 256) return dh <= 96 && d.N >= 64;
  return centered_small() && d.N >= 32 && (dh <= 96 || (dh % 96 == 0 && d.N <= 256));
}

inline int stats_blocks(const AttnDims& d) {
  long long t = (long long)d.B * d.N * (d.ld / 4);
  long long g = (t + 255) / 256;
  return (int)(g > 1024 ? 1024 : (g < 1 ? 1 : g));
}
void carve_attn(Bump& bp, const AttnDims& d, AttnBuf& a) {
  const size_t act = (size_t)d.B * d.N * d.D * esize(d.dtype);
  const size_t map = (size_t)d.B * d.H * d.N * d.ld * esize(d.dtype);
  a.q = bp.take(act); a.k = bp.take(act); a.v = bp.take(act); a.O = bp.take(act);
  if (flash_on(d)) { a.Ps = nullptr; a.Ah = nullptr; }          // the non-materialising form keeps no map at all
  else { a.Ps = bp.take(map); a.Ah = bp.take(map); }
  a.stats = bp.takef(VU_BN_STATS_FLOATS(d.H));
  a.lse2 = bp.takef((size_t)d.B * d.H * d.N);
  a.rinv = bp.takef((size_t)d.B * d.H * d.N);
  a.delta = bp.takef((size_t)d.B * d.H * d.N);
  a.rinvb = bp.takef((size_t)d.B * d.H * d.N);
  a.pk = flash_on(d) ? bp.takef((size_t)d.B * d.N * flash_D(d)) : nullptr;     // sum_k P k of the recompute form (vu_flash.h)
  {   // probability cache of the recompute form (written by the training forward's moments sweep, streamed by the four sweeps after it)
    size_t pcb = (bp.pc_on && flash_on(d)) ? vu_flash_pcache_bytes(d.B, d.N, flash_D(d), d.H) : 0;
    if (pcb > bp.pc_left) pcb = 0;        // over the budget: this module runs the recompute sweeps (bit-identical results)
    bp.pc_left -= pcb; bp.pc_total += pcb;
    a.pc = pcb ? bp.take(pcb) : nullptr;
  }
  if (flash_on(d) && flash_padded(d)) {      // zero-padded q, k, v (kept for the backward) and O of the recompute form
    const size_t actp = (size_t)d.B * d.N * flash_D(d) * esize(d.dtype);
    a.qp = bp.take(actp); a.kp = bp.take(actp); a.vp = bp.take(actp); a.Op = bp.take(actp);
  } else { a.qp = a.kp = a.vp = a.Op = nullptr; }
}
inline size_t attn_partials_floats(const AttnDims& d) {
  return std::max((size_t)std::max(stats_blocks(d), d.B) * 2 * d.H + 1024, vu_flash_partials_floats(d.B, d.N, d.H));
}
void fill_flash_args(vu_flash_args& fa, const AttnDims& d, const vu_attn_params& p, AttnBuf& a, float* partials, vu_rng ra, int training) {
  memset(&fa, 0, sizeof(fa));
  fa.B = d.B; fa.N = d.N; fa.D = d.D; fa.H = d.H; fa.scale = 1.0f / sqrtf((float)(d.D / d.H)); fa.training = training;
  fa.rng = vu_flash_quad_rng(ra);
  fa.q = a.q; fa.k = a.k; fa.v = a.v; fa.O = a.O; fa.lse2 = a.lse2; fa.rinv = a.rinv; fa.pk = a.pk; fa.pcache = a.pc; fa.rinv_b = a.rinvb; fa.delta = a.delta; fa.partials = partials; fa.stats = a.stats;
  fa.mix_w = p.mix_w; fa.mix_b = p.mix_b; fa.bn_w = p.bn_w; fa.bn_b = p.bn_b; fa.run_mean = p.run_mean; fa.run_var = p.run_var;
}

int attn_forward(const AttnDims& d, const vu_attn_params& p, const void* xq, const void* xkv, void* y,
                 AttnBuf& a, float* partials, float attn_drop, float proj_drop, int training, uint64_t seed,
                 uint64_t stream_id, const uint32_t* salt, hipStream_t st, const void* resid = nullptr) {
  const int dt = d.dtype, B = d.B, N = d.N, D = d.D, H = d.H, ld = d.ld;
  const int dh = D / H;
  const long long npatch = (long long)B * N;
  VU_TRY(vu_k_conv3x3_qkv_fwd(dt, xq, xkv, p.wq, p.wk, p.wv, a.q, a.k, a.v, npatch, d.C, d.s, st));
  // e4m3 attention operands: q, k, v are rounded where they are produced, so every product of the forward and of the
  // backward (which re-reads these buffers) sees the same fp8-valued operands; gradients pass straight through
  if (p.operands == 1) VU_TRY(vu_k_round_e4m3(dt, a.q, a.k, a.v, npatch * D, st));
  vu_rng ra = vu_make_rng(seed, 2 * stream_id, training ? attn_drop : 0.f);
  ra.salt = salt;
  if (flash_on(d)) {   // O = A^ v without ever forming a map
    vu_flash_args fa;
    fill_flash_args(fa, d, p, a, partials, ra, training);
    if (flash_padded(d)) {
      VU_REQUIRE(a.qp && a.Op, "attention: the padded recompute form needs its padded buffers (workspace carved for another form)");
      const void* src[4] = {a.q, a.k, a.v, nullptr};
      void* dst[4] = {a.qp, a.kp, a.vp, nullptr};
      VU_TRY(vu_k_head_pad(src, dst, 3, npatch, H, dh, flash_dh(d), st));
      fa.D = flash_D(d); fa.q = a.qp; fa.k = a.kp; fa.v = a.vp; fa.O = a.Op;
      VU_TRY(vu_k_flash_forward(fa, st));
      const void* s2[4] = {a.Op, nullptr, nullptr, nullptr};
      void* d2[4] = {a.O, nullptr, nullptr, nullptr};
      VU_TRY(vu_k_head_pad(s2, d2, 1, npatch, H, flash_dh(d), dh, st));
    } else VU_TRY(vu_k_flash_forward(fa, st));
  } else {
  const double count = (double)B * N * N;
  const bool cen = centered_ok(d);
  int nstat = stats_blocks(d);
  if (cen && vu_attn_f1_ok(dt, B, N, D, H, ld)) {
    // short rows: logits, softmax, dropout, the head mix and its batch moments in one launch (vu_attn_fused.hip)
    VU_TRY(vu_k_attn_f1(a.q, a.k, a.Ps, a.Ah, p.mix_w, partials, &nstat, B, N, D, ld, 1.0f / sqrtf((float)dh), ra, st));
    VU_TRY(vu_check_launch("vu_attn_f1"));
  } else {
  int fused = vu_k_attn_scores(dt, a.q, a.k, a.Ps, B, N, D, H, ld, 1.0f / sqrtf((float)dh), ra, st);
  if (fused < 0) return fused;
  if (fused == 1) {  // shape outside the fused kernel: S = scale * q k^T (model.py:155), then softmax + dropout
    vu_gemm_args g;
    memset(&g, 0, sizeof(g));
    g.A = a.q; g.B = a.k; g.C = a.Ps; g.M = N; g.N = N; g.K = dh;
    g.sAm = D; g.sAk = 1; g.sBk = 1; g.sBn = D; g.ldc = ld;
    g.Z1 = B; g.Z2 = H; g.sA1 = (long long)N * D; g.sA2 = dh; g.sB1 = (long long)N * D; g.sB2 = dh;
    g.sC1 = (long long)H * N * ld; g.sC2 = (long long)N * ld;
    g.alpha = 1.0f / sqrtf((float)dh);
    VU_TRY(vu_gemm_launch(dt, 0, g, st));
    VU_TRY(vu_k_softmax_dropout(dt, a.Ps, (long long)B * H * N, N, ld, ra, st));
  }
  if (cen) {   // one pass over P: batch statistics + the centred mixed map (a.Ah holds Ac, not Ahat)
    const int r = vu_k_mix_stats_mm(dt, a.Ps, p.mix_w, partials, a.Ah, stats_blocks(d), B, H, N, ld, ra.inv_keep, st);
    if (r != 0) { if (r > 0) vu_set_error("attention: centred-map form not available for this shape"); return r < 0 ? r : VU_EUNSUPPORTED; }
  } else if (training) VU_TRY(vu_k_mix_stats(dt, a.Ps, p.mix_w, p.mix_b, partials, stats_blocks(d), B, H, N, ld, ra.inv_keep, st));
  }
  VU_TRY(vu_k_bn_finalize(partials, nstat, p.mix_w, p.mix_b, p.bn_w, p.bn_b, p.run_mean, p.run_var,
                          a.stats, H, N, count, training, 0.1f, 1e-5f, st));
  if (!cen) VU_TRY(vu_k_mix_apply(dt, a.Ps, a.Ah, a.stats, B, H, N, ld, ra.inv_keep, st));
  const float* aff_sc = cen ? a.stats + VU_BN_STATS_SC(H) : nullptr;
  const float* aff_kp = cen ? a.stats + VU_BN_STATS_SC(H) + H : nullptr;
  int mp = vu_k_attn_map_prod(dt, 0, a.Ah, a.v, a.O, aff_sc, aff_kp, B, N, D, H, ld, st);   // streaming kernel for long rows
  if (cen && mp == 1) { vu_set_error("attention: centred-map form needs the streaming product kernel"); return VU_EUNSUPPORTED; }
  if (mp < 0) return mp;
  if (mp == 1) {  // O = Ahat v  (model.py:161)
    vu_gemm_args g;
    memset(&g, 0, sizeof(g));
    g.A = a.Ah; g.B = a.v; g.C = a.O; g.M = N; g.N = dh; g.K = N;
    g.sAm = ld; g.sAk = 1; g.sBk = D; g.sBn = 1; g.ldc = D;
    g.Z1 = B; g.Z2 = H; g.sA1 = (long long)H * N * ld; g.sA2 = (long long)N * ld;
    g.sB1 = (long long)N * D; g.sB2 = dh; g.sC1 = (long long)N * D; g.sC2 = dh;
    g.alpha = 1.f;
    VU_TRY(vu_gemm_launch(dt, 0, g, st));
  }
  }
  {  // y = dropout(O Wp^T + bp)  (model.py:162-163)
    vu_gemm_args g;
    memset(&g, 0, sizeof(g));
    g.A = a.O; g.B = p.proj_w; g.C = y; g.M = B * N; g.N = D; g.K = D;
    g.sAm = D; g.sAk = 1; g.sBk = 1; g.sBn = D; g.ldc = D; g.Z1 = 1; g.Z2 = 1;
    g.alpha = 1.f; g.bias = p.proj_b;
    g.addend = resid;   // block residual: y = dropout(O Wp^T + bp) + x leaves the GEMM in one pass
    g.rng = vu_make_rng(seed, 2 * stream_id + 1, training ? proj_drop : 0.f);
    g.rng.salt = salt;
    g.dropout = g.rng.thr != 0;
    VU_TRY(vu_gemm_launch(dt, 0, g, st));
  }
  return VU_OK;
}

// ---------------------------------------------------------------------------------------------
// Side lane of the backward: the weight gradients of a module do not feed its data-gradient chain, and most of them are
// small latency-bound streams (vu_tsgemm.hip, the convolution weight gradients) that leave most CUs idle, as do the
// data-gradient GEMMs they would otherwise wait behind.  They are enqueued on a second stream, forked from and joined back
// to the caller's stream with events (capturable: a stream capture of the caller's stream follows the fork), inside one
// module at a time so that no scratch buffer changes hands while the side lane reads it.  Per-thread library state created
// on first use.  MEASURED AND NOT KEPT: the two-lane step is 2 % slower than the single stream at 64 and at 16 images per
// GPU (14.59 vs 14.26 ms; the 72 fork / join edges per step cost more than the overlap returns), so it is opt-in
// (VU_SIDE_LANE=1) for experiments; off while the launch profiler runs (it times one stream).
// ---------------------------------------------------------------------------------------------
struct SideLane { hipStream_t s; hipEvent_t e[3]; bool ok; };
inline SideLane* side_lane() {
  static thread_local SideLane sl = {nullptr, {nullptr, nullptr, nullptr}, false};
  static thread_local bool tried = false;
  if (!tried) {
    tried = true;
    const char* ev = getenv("VU_SIDE_LANE");
    bool ok = ev && (ev[0] == '1' || ev[0] == '2');
    if (ok && ev[0] == '2') {          // 2: a LOW-PRIORITY side lane (its kernels only take what the main chain leaves idle)
      int lo = 0, hi = 0;
      ok = hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess && hipStreamCreateWithPriority(&sl.s, hipStreamNonBlocking, lo) == hipSuccess;
    } else {
      ok = ok && hipStreamCreateWithFlags(&sl.s, hipStreamNonBlocking) == hipSuccess;
    }
    for (int i = 0; ok && i < 3; ++i) ok = hipEventCreateWithFlags(&sl.e[i], hipEventDisableTiming) == hipSuccess;
    sl.ok = ok;
  }
  return (sl.ok && !vu_prof_on()) ? &sl : nullptr;
}
// side waits for everything enqueued on `from` so far (event slot i)
inline int lane_wait(SideLane* sl, int i, hipStream_t from, hipStream_t to) {
  if (hipEventRecord(sl->e[i], from) != hipSuccess || hipStreamWaitEvent(to, sl->e[i], 0) != hipSuccess) {
    vu_set_error("backward side lane: event record / wait failed");
    return VU_ELAUNCH;
  }
  return VU_OK;
}

// dz: gradient wrt the module output AFTER the projection dropout mask has been applied
// (i.e. gradient wrt O Wp^T + bp).  add_q / add_kv: tensors added to dxq / dxkv (residuals).
// If dxkv == nullptr the module is self-attention (xq == xkv) and everything lands in dxq.
int attn_backward(const AttnDims& d, const vu_attn_params& p, const vu_attn_grads& gr, const void* xq,
                  const void* xkv, const void* dz, const void* add_q, const void* add_kv, void* dxq, void* dxkv,
                  AttnBuf& a, AttnScratch& sc, float attn_drop, int training, uint64_t seed, uint64_t stream_id,
                  const uint32_t* salt, hipStream_t st) {
  const int dt = d.dtype, B = d.B, N = d.N, D = d.D, H = d.H, ld = d.ld;
  const int dh = D / H;
  const long long npatch = (long long)B * N, rows = (long long)B * N;
  const float inv_keep = (training && attn_drop > 0.f) ? 1.f / (1.f - attn_drop) : 1.f;
  SideLane* sl = side_lane();
  hipStream_t sw = sl ? sl->s : st;                     // where the weight gradients go
  if (sl) VU_TRY(lane_wait(sl, 0, st, sw));             // fork: dz (and everything before) is ready
  {  // dWp += dz^T O ; dbp += column sums of dz (carried by the same GEMM)
    vu_gemm_args g;
    memset(&g, 0, sizeof(g));
    g.A = dz; g.B = a.O; g.C = gr.proj_w; g.M = D; g.N = D; g.K = (int)rows;
    g.sAm = 1; g.sAk = D; g.sBk = D; g.sBn = 1; g.ldc = D; g.Z1 = 1; g.Z2 = 1; g.alpha = 1.f; g.accumulate = 1;
    g.colsum = gr.proj_b; g.colsum_side = 1;
    VU_TRY(vu_gemm_launch(dt, 1, g, sw));
  }
  {  // dO = dz Wp
    vu_gemm_args g;
    memset(&g, 0, sizeof(g));
    g.A = dz; g.B = p.proj_w; g.C = sc.dO; g.M = (int)rows; g.N = D; g.K = D;
    g.sAm = D; g.sAk = 1; g.sBk = D; g.sBn = 1; g.ldc = D; g.Z1 = 1; g.Z2 = 1; g.alpha = 1.f;
    VU_TRY(vu_gemm_launch(dt, 0, g, st));
  }
  if (flash_on(d)) {   // recompute sweeps: no map is read or written (vu_flash.hip)
    VU_TRY(vu_k_bn_bwd_small(dt, sc.dO, a.O, a.v, p.bn_w, p.bn_b, p.mix_w, p.mix_b, a.stats, gr.bn_w, gr.bn_b, sc.partials, B, N, D, H, training, st));
    vu_rng ra = vu_make_rng(seed, 2 * stream_id, training ? attn_drop : 0.f);
    ra.salt = salt;
    vu_flash_args fa;
    fill_flash_args(fa, d, p, a, sc.partials, ra, training);
    fa.dO = sc.dO; fa.dq = sc.dq; fa.dk = sc.dk; fa.dv = sc.dv; fa.d_mix_w = gr.mix_w; fa.d_mix_b = gr.mix_b;
    if (flash_padded(d)) {      // (a.qp, a.kp, a.vp: the forward's padded copies)
      VU_REQUIRE(a.qp && sc.pad, "attention: the padded recompute form needs its padded buffers (workspace carved for another form)");
      const size_t actp = vu_align_up((size_t)rows * flash_D(d) * esize(dt), 256);
      char* pb = (char*)sc.pad;
      const void* src[4] = {sc.dO, nullptr, nullptr, nullptr};
      void* dst[4] = {pb, nullptr, nullptr, nullptr};
      VU_TRY(vu_k_head_pad(src, dst, 1, rows, H, dh, flash_dh(d), st));
      fa.D = flash_D(d); fa.q = a.qp; fa.k = a.kp; fa.v = a.vp; fa.O = a.Op;
      fa.dO = pb; fa.dq = pb + actp; fa.dk = pb + 2 * actp; fa.dv = pb + 3 * actp;
      VU_TRY(vu_k_flash_backward(fa, st));
      const void* s2[4] = {fa.dq, fa.dk, fa.dv, nullptr};
      void* d2[4] = {sc.dq, sc.dk, sc.dv, nullptr};
      VU_TRY(vu_k_head_pad(s2, d2, 3, rows, H, flash_dh(d), dh, st));
    } else
    VU_TRY(vu_k_flash_backward(fa, st));
    if (sl) VU_TRY(lane_wait(sl, 1, st, sw));           // dq, dk, dv are ready
    VU_TRY(vu_k_conv3x3_qkv_wgrad(dt, sc.dq, sc.dk, sc.dv, xq, xkv, gr.wq, gr.wk, gr.wv, npatch, d.C, d.s, sw));
    VU_TRY(vu_k_conv3x3_qkv_dgrad(dt, sc.dq, sc.dk, sc.dv, p.wq, p.wk, p.wv, add_q, add_kv, dxq, dxkv, npatch, d.C, d.s, st));
    if (sl) VU_TRY(lane_wait(sl, 2, sw, st));           // join
    return VU_OK;
  }
  int fo = vu_k_attn_outer(dt, sc.dO, a.v, sc.dA, B, N, D, H, ld, 1.0f, st);   // dAhat = dO v^T (fused, vector stores)
  if (fo < 0) return fo;
  if (fo == 1) {
    vu_gemm_args g;
    memset(&g, 0, sizeof(g));
    g.A = sc.dO; g.B = a.v; g.C = sc.dA; g.M = N; g.N = N; g.K = dh;
    g.sAm = D; g.sAk = 1; g.sBk = 1; g.sBn = D; g.ldc = ld;
    g.Z1 = B; g.Z2 = H; g.sA1 = (long long)N * D; g.sA2 = dh; g.sB1 = (long long)N * D; g.sB2 = dh;
    g.sC1 = (long long)H * N * ld; g.sC2 = (long long)N * ld; g.alpha = 1.f;
    VU_TRY(vu_gemm_launch(dt, 0, g, st));
  }
  const bool cen = centered_ok(d);
  int mp = vu_k_attn_map_prod(dt, 1, a.Ah, sc.dO, sc.dv, cen ? a.stats + VU_BN_STATS_SC(H) : nullptr,
                              cen ? a.stats + VU_BN_STATS_SC(H) + H : nullptr, B, N, D, H, ld, st);
  if (cen && mp == 1) { vu_set_error("attention: centred-map form needs the streaming product kernel"); return VU_EUNSUPPORTED; }
  if (mp < 0) return mp;
  if (mp == 1) {  // dv = Ahat^T dO
    vu_gemm_args g;
    memset(&g, 0, sizeof(g));
    g.A = a.Ah; g.B = sc.dO; g.C = sc.dv; g.M = N; g.N = dh; g.K = N;
    g.sAm = 1; g.sAk = ld; g.sBk = D; g.sBn = 1; g.ldc = D;
    g.Z1 = B; g.Z2 = H; g.sA1 = (long long)H * N * ld; g.sA2 = (long long)N * ld;
    g.sB1 = (long long)N * D; g.sB2 = dh; g.sC1 = (long long)N * D; g.sC2 = dh; g.alpha = 1.f;
    VU_TRY(vu_gemm_launch(dt, 0, g, st));
  }
  const double count = (double)B * N * N;
  // BatchNorm-backward statistics from dO, O, v (vu_attn.hip) - no pass over the maps
  (void)count;
  VU_TRY(vu_k_bn_bwd_small(dt, sc.dO, a.O, a.v, p.bn_w, p.bn_b, p.mix_w, p.mix_b, a.stats, gr.bn_w, gr.bn_b, sc.partials, B, N, D, H, training, st));
  VU_TRY(vu_k_map_bwd(dt, a.Ps, sc.dA, p.mix_w, p.mix_b, p.bn_w, a.stats, gr.mix_w, gr.mix_b, B, H, N, ld, inv_keep,
                      1.0f / sqrtf((float)dh), st));
  mp = vu_k_attn_map_prod(dt, 0, sc.dA, a.k, sc.dq, nullptr, nullptr, B, N, D, H, ld, st);
  if (mp < 0) return mp;
  if (mp == 1) {  // dq = dS k
    vu_gemm_args g;
    memset(&g, 0, sizeof(g));
    g.A = sc.dA; g.B = a.k; g.C = sc.dq; g.M = N; g.N = dh; g.K = N;
    g.sAm = ld; g.sAk = 1; g.sBk = D; g.sBn = 1; g.ldc = D;
    g.Z1 = B; g.Z2 = H; g.sA1 = (long long)H * N * ld; g.sA2 = (long long)N * ld;
    g.sB1 = (long long)N * D; g.sB2 = dh; g.sC1 = (long long)N * D; g.sC2 = dh; g.alpha = 1.f;
    VU_TRY(vu_gemm_launch(dt, 0, g, st));
  }
  mp = vu_k_attn_map_prod(dt, 1, sc.dA, a.q, sc.dk, nullptr, nullptr, B, N, D, H, ld, st);
  if (mp < 0) return mp;
  if (mp == 1) {  // dk = dS^T q
    vu_gemm_args g;
    memset(&g, 0, sizeof(g));
    g.A = sc.dA; g.B = a.q; g.C = sc.dk; g.M = N; g.N = dh; g.K = N;
    g.sAm = 1; g.sAk = ld; g.sBk = D; g.sBn = 1; g.ldc = D;
    g.Z1 = B; g.Z2 = H; g.sA1 = (long long)H * N * ld; g.sA2 = (long long)N * ld;
    g.sB1 = (long long)N * D; g.sB2 = dh; g.sC1 = (long long)N * D; g.sC2 = dh; g.alpha = 1.f;
    VU_TRY(vu_gemm_launch(dt, 0, g, st));
  }
  if (sl) VU_TRY(lane_wait(sl, 1, st, sw));             // dq, dk, dv are ready
  VU_TRY(vu_k_conv3x3_qkv_wgrad(dt, sc.dq, sc.dk, sc.dv, xq, xkv, gr.wq, gr.wk, gr.wv, npatch, d.C, d.s, sw));
  VU_TRY(vu_k_conv3x3_qkv_dgrad(dt, sc.dq, sc.dk, sc.dv, p.wq, p.wk, p.wv, add_q, add_kv, dxq, dxkv, npatch, d.C, d.s, st));
  if (sl) VU_TRY(lane_wait(sl, 2, sw, st));             // join
  return VU_OK;
}

// ---------------------------------------------------------------------------------------------
// model workspace
// ---------------------------------------------------------------------------------------------
struct BlockBuf { AttnBuf at; void *z1, *x1, *hpre, *hact, *z2, *out; float *ln1s, *ln2s; };
struct ModelWS {
  void* tok0;
  std::vector<BlockBuf> enc, bot, dec;
  std::vector<AttnBuf> skip;
  std::vector<void*> skip_out, down_out, up_out;
  void* img;
  // forward scratch
  void *y_attn, *f_ff;
  // backward scratch
  void *gx0, *gx1, *ga, *gb, *gc, *gh;
  AttnScratch asc;
  std::vector<void*> dskip;
  float *partials, *lnp, *lnp2;
  void* wgs; size_t wgs_bytes;      // split-K slab of the tall-skinny weight gradients (bf16 storage; vu_gemm_set_scratch)
  void* wga; size_t wga_bytes;      // arena of the deferred (batched) reductions of those gradients (vu_tsgemm_set_arena)
  size_t bytes;
  size_t pcache_bytes;              // of which: probability caches
};

void carve_block(Bump& bp, const Plan& pl, int B, int level, BlockBuf& b) {
  const Level& L = pl.lv[level];
  const int dt = pl.cfg.dtype;
  AttnDims d{dt, B, L.N, L.D, pl.cfg.num_heads, pl.cfg.num_channels, L.s, L.ld, 1, flash_switch()};
  carve_attn(bp, d, b.at);
  const size_t act = (size_t)B * L.N * L.D * esize(dt), hh = (size_t)B * L.N * L.hid * esize(dt);
  b.z1 = bp.take(act); b.x1 = bp.take(act); b.hpre = bp.take(hh); b.hact = bp.take(hh);
  b.z2 = bp.take(act); b.out = bp.take(act);
  b.ln1s = bp.takef(2 * B); b.ln2s = bp.takef(2 * B);
}

// training = 0: the layout of an eval forward (and of a backward through it): no probability cache is carved - only the TRAINING
// forward's moments sweep fills one.  A forward and its backward must be given the same flag (they are: both take `training`).
void carve_model(const Plan& pl, int B, char* base, ModelWS& w, int training) {
  Bump bp{base, 0};
  bp.pc_on = training != 0;
  bp.pc_left = vu_flash_pcache_budget();
  const vu_config& c = pl.cfg;
  const int dt = c.dtype, H = c.num_heads;
  const long long P = (long long)c.num_channels * c.im_size * c.im_size;
  const size_t act = (size_t)B * P * esize(dt);
  w.tok0 = bp.take(act);
  w.enc.resize(pl.enc.size()); w.bot.resize(pl.bot.size()); w.dec.resize(pl.dec.size()); w.skip.resize(pl.skip.size());
  for (size_t i = 0; i < pl.enc.size(); ++i) carve_block(bp, pl, B, pl.enc[i].level, w.enc[i]);
  for (size_t i = 0; i < pl.bot.size(); ++i) carve_block(bp, pl, B, pl.bot[i].level, w.bot[i]);
  for (size_t i = 0; i < pl.dec.size(); ++i) carve_block(bp, pl, B, pl.dec[i].level, w.dec[i]);
  for (int j = 0; j < c.depth; ++j) {
    const Level& L = pl.lv[c.depth - j - 1];
    AttnDims d{dt, B, L.N, L.D, H, c.num_channels, L.s, L.ld, 1, flash_switch()};
    carve_attn(bp, d, w.skip[j]);
  }
  w.skip_out.resize(c.depth); w.down_out.resize(c.depth); w.up_out.resize(c.depth); w.dskip.resize(c.depth);
  for (int j = 0; j < c.depth; ++j) { w.skip_out[j] = bp.take(act); w.down_out[j] = bp.take(act); w.up_out[j] = bp.take(act); }
  w.img = bp.take(act);
  w.y_attn = bp.take(act); w.f_ff = bp.take(act);
  w.gx0 = bp.take(act); w.gx1 = bp.take(act); w.ga = bp.take(act); w.gb = bp.take(act); w.gc = bp.take(act);
  size_t hh = 0, map = 0, npart = 0, padb = 0;
  int nb = 1;
  for (const Level& L : pl.lv) {
    hh = std::max(hh, (size_t)B * L.N * L.hid * esize(dt));
    AttnDims d{dt, B, L.N, L.D, H, c.num_channels, L.s, L.ld, 1, flash_switch()};
    if (!flash_on(d)) map = std::max(map, (size_t)B * H * L.N * L.ld * esize(dt));     // dA^ / dS scratch of the materialised forms
    else if (flash_padded(d)) padb = std::max(padb, 4 * vu_align_up((size_t)B * L.N * flash_D(d) * esize(dt), 256));   // padded dO, dq, dk, dv
    nb = std::max(nb, stats_blocks(d));
    npart = std::max(npart, attn_partials_floats(d));
  }
  w.gh = bp.take(hh);
  w.asc.dO = bp.take(act); w.asc.dq = bp.take(act); w.asc.dk = bp.take(act); w.asc.dv = bp.take(act);
  w.asc.dA = bp.take(map);
  w.asc.pad = padb ? bp.take(padb) : nullptr;
  w.asc.nblocks = nb;
  for (int j = 0; j < c.depth; ++j) w.dskip[j] = bp.take(act);
  w.partials = bp.takef(std::max((size_t)std::max(nb, B) * 2 * H + 1024, npart));
  w.asc.partials = w.partials;
  w.lnp = bp.takef((size_t)B * vu_ln_nchunks(P) * 3);
  w.lnp2 = bp.takef((size_t)B * vu_ln_nbchunks(P) * 2);
  w.wgs_bytes = dt == 1 ? (size_t)40 << 20 : 0;      // (16 K slices of a 768 x 768 output)
  w.wgs = w.wgs_bytes ? bp.take(w.wgs_bytes) : nullptr;
  // (a Base backward queues ~100 MB of partial tiles at 16 images and ~490 MB at 64 - the K slices grow with the token count: sized
  // for ONE flush per call, 10 MB per image within [160 MB, 1 GB])
  w.wga_bytes = dt == 1 ? std::min((size_t)1 << 30, std::max((size_t)160 << 20, (size_t)B * ((size_t)10 << 20))) : 0;
  w.wga = w.wga_bytes ? bp.take(w.wga_bytes) : nullptr;
  w.bytes = vu_align_up(bp.off, 256);
  w.pcache_bytes = bp.pc_total;
}

vu_attn_params attn_params(const AttnP& a, const vu_config& c, const float* prm, const void* shadow, float* bn) {
  vu_attn_params p;
  p.mix_w = prm + a.mixw; p.mix_b = prm + a.mixb; p.bn_w = prm + a.bng; p.bn_b = prm + a.bnb;
  p.wq = prm + a.wq; p.wk = prm + a.wk; p.wv = prm + a.wv;
  p.proj_w = c.dtype == 0 ? (const void*)(prm + a.projw) : (const void*)((const bf16_t*)shadow + a.projw);
  p.proj_b = prm + a.projb;
  p.run_mean = bn ? bn + (long long)a.bn * 2 * c.num_heads : nullptr;
  p.run_var = bn ? bn + (long long)a.bn * 2 * c.num_heads + c.num_heads : nullptr;
  p.operands = c.attn_operands;
  return p;
}
vu_attn_grads attn_grads(const AttnP& a, float* g) {
  vu_attn_grads r;
  r.mix_w = g + a.mixw; r.mix_b = g + a.mixb; r.bn_w = g + a.bng; r.bn_b = g + a.bnb;
  r.wq = g + a.wq; r.wk = g + a.wk; r.wv = g + a.wv; r.proj_w = g + a.projw; r.proj_b = g + a.projb;
  return r;
}
inline const void* wptr(const vu_config& c, const float* prm, const void* shadow, long long off) {
  return c.dtype == 0 ? (const void*)(prm + off) : (const void*)((const bf16_t*)shadow + off);
}

struct Ctx {
  const Plan* pl; int B; const float* prm; const void* shadow; float* bn; float* grads;
  int training; uint64_t seed; const uint32_t* salt; hipStream_t st; ModelWS* w;
};

// ---------------------------------------------------------------------------------------------
// FeedForward (model.py:95-110): Linear(D,hid) -> GELU -> Dropout(linear_drop) -> Linear(hid,D) -> Dropout(linear_drop).
// hpre keeps the GELU pre-activation, hact the (dropped) activation.  The two dropout sites draw from the streams
// VU_FF_STREAM(stream_id, 0 / 1), which do not collide with the attention-map / projection streams 2 sid, 2 sid + 1.
// ---------------------------------------------------------------------------------------------
#define VU_FF_STREAM(sid, k) ((1ull << 32) + 2 * (uint64_t)(sid) + (k))
struct FFDims { int dtype; long long rows; int D, hid; };

int ff_forward(const FFDims& f, const void* x, const void* w1, const float* b1, const void* w2, const float* b2, void* hpre,
               void* hact, void* y, const void* resid, float linear_drop, int training, uint64_t seed, uint64_t stream_id,
               const uint32_t* salt, hipStream_t st) {
  if (!(training && linear_drop > 0.f)) {      // no dropout to draw: the fused pair where the shape is covered
    const int rc = vu_ff2_forward_try(f.dtype, x, w1, b1, w2, b2, hpre, hact, y, resid, f.rows, f.D, f.hid, st);
    if (rc < 0) return rc;
    if (rc > 0) return VU_OK;
  }
  {  // hact = dropout(gelu(x W1^T + b1))  (model.py:102-105)
    vu_gemm_args g;
    memset(&g, 0, sizeof(g));
    g.A = x; g.B = w1; g.C = hact; g.aux = hpre;
    g.M = (int)f.rows; g.N = f.hid; g.K = f.D; g.sAm = f.D; g.sAk = 1; g.sBk = 1; g.sBn = f.D; g.ldc = f.hid;
    g.Z1 = 1; g.Z2 = 1; g.alpha = 1.f; g.bias = b1; g.act = VU_ACT_GELU;
    g.rng = vu_make_rng(seed, VU_FF_STREAM(stream_id, 0), training ? linear_drop : 0.f);
    g.rng.salt = salt;
    g.dropout = g.rng.thr != 0;
    VU_TRY(vu_gemm_launch(f.dtype, 0, g, st));
  }
  {  // y = dropout(hact W2^T + b2) (+ resid)  (model.py:106-107)
    vu_gemm_args g;
    memset(&g, 0, sizeof(g));
    g.A = hact; g.B = w2; g.C = y;
    g.M = (int)f.rows; g.N = f.D; g.K = f.hid; g.sAm = f.hid; g.sAk = 1; g.sBk = 1; g.sBn = f.hid; g.ldc = f.D;
    g.Z1 = 1; g.Z2 = 1; g.alpha = 1.f; g.bias = b2; g.addend = resid;
    g.rng = vu_make_rng(seed, VU_FF_STREAM(stream_id, 1), training ? linear_drop : 0.f);
    g.rng.salt = salt;
    g.dropout = g.rng.thr != 0;
    VU_TRY(vu_gemm_launch(f.dtype, 0, g, st));
  }
  return VU_OK;
}

// dy: gradient wrt the module output; dym: the same with the output dropout mask applied (== dy when linear_drop is
// off).  dx = dh W1 (+ addend).  gh: scratch (rows, hid).  Weight gradients are accumulated.
int ff_backward(const FFDims& f, const void* x, const void* w1, const void* w2, const void* hpre, const void* hact,
                const void* dym, const void* addend, void* dx, float* dw1, float* db1, float* dw2, float* db2, void* gh,
                float linear_drop, int training, uint64_t seed, uint64_t stream_id, const uint32_t* salt, hipStream_t st) {
  SideLane* sl = side_lane();
  hipStream_t sw = sl ? sl->s : st;                     // weight gradients on the side lane (see SideLane)
  if (sl) VU_TRY(lane_wait(sl, 0, st, sw));             // fork: dym is ready
  {  // dW2 += dym^T hact ; db2 += column sums of dym
    vu_gemm_args g;
    memset(&g, 0, sizeof(g));
    g.A = dym; g.B = hact; g.C = dw2; g.M = f.D; g.N = f.hid; g.K = (int)f.rows;
    g.sAm = 1; g.sAk = f.D; g.sBk = f.hid; g.sBn = 1; g.ldc = f.hid; g.Z1 = 1; g.Z2 = 1; g.alpha = 1.f; g.accumulate = 1;
    g.colsum = db2; g.colsum_side = 1;
    VU_TRY(vu_gemm_launch(f.dtype, 1, g, sw));
  }
  int fused = 0;                                        // dh and dx in one kernel (vu_ff2.hip) where covered
  if (!(training && linear_drop > 0.f)) {
    fused = vu_ff2_backward_try(f.dtype, dym, w1, w2, hpre, gh, dx, addend, f.rows, f.D, f.hid, st);
    if (fused < 0) return fused;
  }
  if (!fused) {  // dh = dropout-mask * (dym W2) * gelu'(hpre)
    vu_gemm_args g;
    memset(&g, 0, sizeof(g));
    g.A = dym; g.B = w2; g.C = gh; g.aux = (void*)hpre;
    g.M = (int)f.rows; g.N = f.hid; g.K = f.D; g.sAm = f.D; g.sAk = 1; g.sBk = f.hid; g.sBn = 1; g.ldc = f.hid;
    g.Z1 = 1; g.Z2 = 1; g.alpha = 1.f; g.act = VU_ACT_DGELU;
    g.rng = vu_make_rng(seed, VU_FF_STREAM(stream_id, 0), training ? linear_drop : 0.f);
    g.rng.salt = salt;
    g.dropout = g.rng.thr != 0;
    VU_TRY(vu_gemm_launch(f.dtype, 0, g, st));
  }
  {  // dW1 += dh^T x ; db1 += column sums of dh
    vu_gemm_args g;
    memset(&g, 0, sizeof(g));
    g.A = gh; g.B = x; g.C = dw1; g.M = f.hid; g.N = f.D; g.K = (int)f.rows;
    g.sAm = 1; g.sAk = f.hid; g.sBk = f.D; g.sBn = 1; g.ldc = f.D; g.Z1 = 1; g.Z2 = 1; g.alpha = 1.f; g.accumulate = 1;
    g.colsum = db1; g.colsum_side = 1;
    if (sl) VU_TRY(lane_wait(sl, 1, st, sw));           // dh is ready
    VU_TRY(vu_gemm_launch(f.dtype, 1, g, sw));
  }
  if (!fused) {  // dx = dh W1 (+ addend)
    vu_gemm_args g;
    memset(&g, 0, sizeof(g));
    g.A = gh; g.B = w1; g.C = dx; g.addend = addend;
    g.M = (int)f.rows; g.N = f.D; g.K = f.hid; g.sAm = f.hid; g.sAk = 1; g.sBk = f.D; g.sBn = 1; g.ldc = f.D;
    g.Z1 = 1; g.Z2 = 1; g.alpha = 1.f;
    VU_TRY(vu_gemm_launch(f.dtype, 0, g, st));
  }
  if (sl) VU_TRY(lane_wait(sl, 2, sw, st));             // join
  return VU_OK;
}

int block_forward(Ctx& cx, const BlockP& bp, BlockBuf& bb, const void* x, uint64_t stream_id) {
  const vu_config& c = cx.pl->cfg;
  const Level& L = cx.pl->lv[bp.level];
  const int dt = c.dtype, B = cx.B;
  const long long P = (long long)L.N * L.D;
  AttnDims d{dt, B, L.N, L.D, c.num_heads, c.num_channels, L.s, L.ld, 1, flash_switch()};
  vu_attn_params ap = attn_params(bp.at, c, cx.prm, cx.shadow, cx.bn);
  VU_TRY(attn_forward(d, ap, x, x, bb.z1, bb.at, cx.w->partials, c.attn_drop, c.proj_drop, cx.training,
                      cx.seed, stream_id, cx.salt, cx.st, x));          // z1 = attn(x) + x
  VU_TRY(vu_k_add_ln_fwd(dt, bb.z1, nullptr, bb.z1, cx.prm + bp.ln1w, cx.prm + bp.ln1b, bb.x1, cx.w->lnp, bb.ln1s,
                         B, P, 1e-5f, cx.st));
  FFDims fd{dt, (long long)B * L.N, L.D, L.hid};
  VU_TRY(ff_forward(fd, bb.x1, wptr(c, cx.prm, cx.shadow, bp.w1), cx.prm + bp.b1, wptr(c, cx.prm, cx.shadow, bp.w2), cx.prm + bp.b2,
                    bb.hpre, bb.hact, bb.z2, bb.x1, c.linear_drop, cx.training, cx.seed, stream_id, cx.salt, cx.st));   // z2 = FF(x1) + x1
  VU_TRY(vu_k_add_ln_fwd(dt, bb.z2, nullptr, bb.z2, cx.prm + bp.ln2w, cx.prm + bp.ln2b, bb.out, cx.w->lnp, bb.ln2s,
                         B, P, 1e-5f, cx.st));
  return VU_OK;
}

// dout -> dx (both B*P T).  dout is clobbered.
int block_backward(Ctx& cx, const BlockP& bp, BlockBuf& bb, const void* x, void* dout, void* dx, uint64_t stream_id) {
  const vu_config& c = cx.pl->cfg;
  const Level& L = cx.pl->lv[bp.level];
  const int dt = c.dtype, B = cx.B;
  const long long P = (long long)L.N * L.D, rows = (long long)B * L.N;
  ModelWS& w = *cx.w;
  float* G = cx.grads;
  // LN2 backward: dz2 -> ga ; with linear_drop, the copy masked by the FF output dropout -> gc
  vu_rng rf = vu_make_rng(cx.seed, VU_FF_STREAM(stream_id, 1), cx.training ? c.linear_drop : 0.f);
  rf.salt = cx.salt;
  const bool ffmasked = rf.thr != 0;
  VU_TRY(vu_k_ln_bwd(dt, dout, bb.z2, cx.prm + bp.ln2w, bb.ln2s, G + bp.ln2w, G + bp.ln2b, w.lnp2, w.ga, ffmasked ? w.gc : nullptr, rf, B, P, cx.st));
  // FeedForward backward: dx1 = dh W1 + dz2 -> gb
  FFDims fd{dt, rows, L.D, L.hid};
  VU_TRY(ff_backward(fd, bb.x1, wptr(c, cx.prm, cx.shadow, bp.w1), wptr(c, cx.prm, cx.shadow, bp.w2), bb.hpre, bb.hact,
                     ffmasked ? w.gc : w.ga, w.ga, w.gb, G + bp.w1, G + bp.b1, G + bp.w2, G + bp.b2, w.gh, c.linear_drop, cx.training,
                     cx.seed, stream_id, cx.salt, cx.st));
  // LN1 backward: dz1 -> ga ; masked copy (projection dropout backward) -> gc
  vu_rng rp = vu_make_rng(cx.seed, 2 * stream_id + 1, cx.training ? c.proj_drop : 0.f);
  rp.salt = cx.salt;
  const bool masked = rp.thr != 0;
  VU_TRY(vu_k_ln_bwd(dt, w.gb, bb.z1, cx.prm + bp.ln1w, bb.ln1s, G + bp.ln1w, G + bp.ln1b, w.lnp2, w.ga,
                     masked ? w.gc : nullptr, rp, B, P, cx.st));
  AttnDims d{dt, B, L.N, L.D, c.num_heads, c.num_channels, L.s, L.ld, 1, flash_switch()};
  vu_attn_params ap = attn_params(bp.at, c, cx.prm, cx.shadow, cx.bn);
  vu_attn_grads ag = attn_grads(bp.at, G);
  VU_TRY(attn_backward(d, ap, ag, x, x, masked ? w.gc : w.ga, w.ga, nullptr, dx, nullptr, bb.at, w.asc, c.attn_drop,
                       cx.training, cx.seed, stream_id, cx.salt, cx.st));
  return VU_OK;
}

int model_forward(Ctx& cx, const float* x, float* y) {
  const Plan& pl = *cx.pl;
  const vu_config& c = pl.cfg;
  ModelWS& w = *cx.w;
  const int dt = c.dtype, B = cx.B, C = c.num_channels, im = c.im_size;
  // PatchEncoder (model.py:84-91): tokens + positional embedding
  VU_TRY(vu_k_retile(dt, 1, 0, x, w.tok0, cx.prm + pl.pos, B, C, im, im, c.patch_size, cx.st));
  const void* cur = w.tok0;
  uint64_t sid = 0;
  std::vector<const void*> skips(c.depth, nullptr);
  for (size_t i = 0; i < pl.enc.size(); ++i) {  // model.py:388-392
    VU_TRY(block_forward(cx, pl.enc[i], w.enc[i], cur, sid++));
    cur = w.enc[i].out;
    if ((i + 1) % c.depth_te == 0) {
      const int l = pl.enc[i].level;
      skips[l] = cur;
      VU_TRY(vu_k_retile(dt, 0, 0, cur, w.down_out[l], nullptr, B, C, im, pl.lv[l].s, pl.lv[l + 1].s, cx.st));
      cur = w.down_out[l];
    }
  }
  for (size_t i = 0; i < pl.bot.size(); ++i) {  // model.py:400-401
    VU_TRY(block_forward(cx, pl.bot[i], w.bot[i], cur, sid++));
    cur = w.bot[i].out;
  }
  for (size_t i = 0; i < pl.dec.size(); ++i) {  // model.py:410-418
    VU_TRY(block_forward(cx, pl.dec[i], w.dec[i], cur, sid++));
    cur = w.dec[i].out;
    if ((i + 1) % c.depth_te == 0) {
      const int j = (int)(i + 1) / c.depth_te - 1;  // SkipConnections[j]
      const int lfrom = pl.dec[i].level, lto = lfrom - 1;
      VU_TRY(vu_k_retile(dt, 0, 0, cur, w.up_out[j], nullptr, B, C, im, pl.lv[lfrom].s, pl.lv[lto].s, cx.st));
      const Level& L = pl.lv[lto];
      AttnDims d{dt, B, L.N, L.D, c.num_heads, C, L.s, L.ld, 1, flash_switch()};
      vu_attn_params ap = attn_params(pl.skip[j], c, cx.prm, cx.shadow, cx.bn);
      VU_TRY(attn_forward(d, ap, skips[lto], w.up_out[j], w.skip_out[j], w.skip[j], w.partials, c.attn_drop, c.proj_drop,
                          cx.training, cx.seed, sid++, cx.salt, cx.st));
      cur = w.skip_out[j];
    }
  }
  // model.py:425-428
  if (c.out_conv) {
    VU_TRY(vu_k_retile(dt, 0, 0, cur, w.img, nullptr, B, C, im, pl.lv[0].s, im, cx.st));
    VU_TRY(vu_k_conv3x3_fwd(dt, 1, w.img, cx.prm + pl.outw, cx.prm + pl.outb, y, B, C, im, cx.st));
  } else {
    VU_TRY(vu_k_retile(dt, 0, 1, cur, y, nullptr, B, C, im, pl.lv[0].s, im, cx.st));
  }
  return VU_OK;
}

// ---------------------------------------------------------------------------------------------
// backward, as a sequence of UNITS in reverse execution order so that the data-parallel engine can
// start the all-reduce of a gradient bucket as soon as the units that produce it are enqueued:
//   unit 0                     output conv (+ re-tiling of dy)                  grads: conv2d.*
//   per decoder block i = last..0:  [SkipConnections.j when block i closes a level]   Decoders.i
//   BottleNeck.i = last..0, Encoders.i = last..0 (the level's down-sampling / skip gradient first)
//   last unit                  positional-embedding gradient (+ dx)             PE.position_embedding.weight
// The running gradient ping-pongs between gx0 / gx1; a call that starts in the middle replays the
// pointer swaps of the units it skips (no launches), so any split into calls gives the same result.
// ---------------------------------------------------------------------------------------------
struct Unit { int kind; int idx; };   // kind: 0 head, 1 skip j, 2 dec i, 3 bot i, 4 enc i, 5 PE
std::vector<Unit> backward_units(const Plan& pl) {
  const vu_config& c = pl.cfg;
  std::vector<Unit> u;
  u.push_back({0, 0});
  for (int i = (int)pl.dec.size() - 1; i >= 0; --i) {
    if ((i + 1) % c.depth_te == 0) u.push_back({1, (i + 1) / c.depth_te - 1});
    u.push_back({2, i});
  }
  for (int i = (int)pl.bot.size() - 1; i >= 0; --i) u.push_back({3, i});
  for (int i = (int)pl.enc.size() - 1; i >= 0; --i) u.push_back({4, i});
  u.push_back({5, 0});
  return u;
}
// arena range [lo, hi) of the parameters whose gradients a unit produces
void unit_range(const Plan& pl, const Unit& u, long long& lo, long long& hi) {
  std::string pre;
  switch (u.kind) {
    case 0: pre = "conv2d."; break;
    case 1: pre = "SkipConnections." + std::to_string(u.idx) + "."; break;
    case 2: pre = "Decoders." + std::to_string(u.idx) + "."; break;
    case 3: pre = "BottleNeck." + std::to_string(u.idx) + "."; break;
    case 4: pre = "Encoders." + std::to_string(u.idx) + "."; break;
    default: pre = "PE."; break;
  }
  lo = -1; hi = -1;
  for (size_t t = 0; t < pl.table.size(); ++t) {
    const vu_param_entry& e = pl.table[t];
    if (strncmp(e.name, pre.c_str(), pre.size()) != 0) continue;
    const long long end = t + 1 < pl.table.size() ? pl.table[t + 1].offset : pl.total;
    if (lo < 0 || e.offset < lo) lo = e.offset;
    if (end > hi) hi = end;
  }
  if (lo < 0) { lo = 0; hi = 0; }     // (no output conv: unit 0 has no parameters)
}

int model_backward(Ctx& cx, const float* dy, float* dx, int first, int last) {
  const Plan& pl = *cx.pl;
  const vu_config& c = pl.cfg;
  ModelWS& w = *cx.w;
  const int dt = c.dtype, B = cx.B, C = c.num_channels, im = c.im_size;
  const long long P = (long long)C * im * im;
  float* G = cx.grads;
  // forward-order stream ids
  const uint64_t sid_enc0 = 0, sid_bot0 = pl.enc.size(), sid_dec0 = sid_bot0 + pl.bot.size();
  void* cur = w.gx0;
  void* oth = w.gx1;
  auto swap = [&]() { void* t = cur; cur = oth; oth = t; };
  const std::vector<Unit> units = backward_units(pl);
  for (int ui = 0; ui < (int)units.size() && ui <= last; ++ui) {
    const Unit& u = units[ui];
    const bool run = ui >= first;
    const int i = u.idx;
    switch (u.kind) {
      case 0:
        if (run) {
          if (c.out_conv) {
            VU_TRY(vu_k_conv3x3_wgrad(dt, 1, dy, w.img, G + pl.outw, G + pl.outb, B, C, im, cx.st));
            VU_TRY(vu_k_conv3x3_dgrad(dt, 1, dy, cx.prm + pl.outw, nullptr, w.ga, B, C, im, cx.st));
            VU_TRY(vu_k_retile(dt, 0, 0, w.ga, cur, nullptr, B, C, im, im, pl.lv[0].s, cx.st));
          } else {
            VU_TRY(vu_k_retile(dt, 1, 0, dy, cur, nullptr, B, C, im, im, pl.lv[0].s, cx.st));
          }
        }
        break;
      case 1: {   // SkipConnections[j] follows decoder block (j+1)*depth_te-1 in the forward
        if (!run) break;
        const int j = i, ib = (j + 1) * c.depth_te - 1;
        const uint64_t sid_skip = sid_dec0 + ib + (uint64_t)(ib / c.depth_te) + 1;
        const int lfrom = pl.dec[ib].level, lto = lfrom - 1;
        const Level& L = pl.lv[lto];
        AttnDims d{dt, B, L.N, L.D, c.num_heads, C, L.s, L.ld, 1, flash_switch()};
        vu_attn_params ap = attn_params(pl.skip[j], c, cx.prm, cx.shadow, cx.bn);
        vu_attn_grads ag = attn_grads(pl.skip[j], G);
        const void* dz = cur;
        vu_rng rp = vu_make_rng(cx.seed, 2 * sid_skip + 1, cx.training ? c.proj_drop : 0.f);
        rp.salt = cx.salt;
        if (rp.thr != 0) { VU_TRY(vu_k_dropout(dt, cur, w.gc, (long long)B * P, rp, cx.st)); dz = w.gc; }
        // encoder skip tensor at level lto is the output of the last encoder block of that level
        const void* enc_x = w.enc[(lto + 1) * c.depth_te - 1].out;
        VU_TRY(attn_backward(d, ap, ag, enc_x, w.up_out[j], dz, nullptr, nullptr, w.dskip[lto], w.ga, w.skip[j], w.asc,
                             c.attn_drop, cx.training, cx.seed, sid_skip, cx.salt, cx.st));
        // gradient of upsampling = retile back to the finer level
        VU_TRY(vu_k_retile(dt, 0, 0, w.ga, cur, nullptr, B, C, im, pl.lv[lto].s, pl.lv[lfrom].s, cx.st));
        break;
      }
      case 2: {
        if (run) {
          const uint64_t sid_blk = sid_dec0 + i + (uint64_t)(i / c.depth_te);   // dec blocks and skips interleave
          const void* xin = (i == 0) ? (pl.bot.empty() ? (c.depth > 0 ? w.down_out[c.depth - 1] : w.tok0) : w.bot.back().out)
                                     : ((i % c.depth_te == 0) ? w.skip_out[i / c.depth_te - 1] : w.dec[i - 1].out);
          VU_TRY(block_backward(cx, pl.dec[i], w.dec[i], xin, cur, oth, sid_blk));
        }
        swap();
        break;
      }
      case 3: {
        if (run) {
          const void* xin = (i == 0) ? (c.depth > 0 ? w.down_out[c.depth - 1] : w.tok0) : w.bot[i - 1].out;
          VU_TRY(block_backward(cx, pl.bot[i], w.bot[i], xin, cur, oth, sid_bot0 + i));
        }
        swap();
        break;
      }
      case 4: {
        if ((i + 1) % c.depth_te == 0) {
          if (run) {   // gradient of downsampling (retile back) + gradient arriving through the skip connection
            const int l = pl.enc[i].level;
            static const bool sep = [] { const char* e = getenv("VU_RETILE_ADD"); return e && e[0] == '0'; }();      // A/B switch: the sum as its own launch
            VU_TRY(vu_k_retile(dt, 0, 0, cur, oth, nullptr, B, C, im, pl.lv[l + 1].s, pl.lv[l].s, cx.st, sep ? nullptr : w.dskip[l]));
            if (sep) VU_TRY(vu_k_add(dt, oth, w.dskip[l], oth, (long long)B * P, cx.st));
          }
          swap();
        }
        if (run) {
          const void* xin = (i == 0) ? w.tok0 : ((i % c.depth_te == 0) ? w.down_out[pl.enc[i].level - 1] : w.enc[i - 1].out);
          VU_TRY(block_backward(cx, pl.enc[i], w.enc[i], xin, cur, oth, sid_enc0 + i));
        }
        swap();
        break;
      }
      default:
        if (run) {
          VU_TRY(vu_k_batch_sum(dt, cur, G + pl.pos, B, P, cx.st));
          if (dx) VU_TRY(vu_k_retile(dt, 0, 1, cur, dx, nullptr, B, C, im, c.patch_size, im, cx.st));
        }
        break;
    }
  }
  return VU_OK;
}

}  // namespace

// =============================================================================================
// extern "C" boundary
// =============================================================================================
extern "C" {

// 200: vu_config gained attn_operands, vu_attn_params gained operands (round 2) - callers check the version AND the struct size
int vu_version(void) { return 200; }
int vu_config_size(void) { return (int)sizeof(vu_config); }
int vu_set_attn_form(int flash, int centered) {
  if (flash < -1 || flash > 1 || centered < 0 || centered > 1) { vu_set_error("vu_set_attn_form: flash in {-1, 0, 1}, centered in {0, 1}"); return VU_EINVAL; }
  attn_form() = AttnForm{flash, centered};
  return VU_OK;
}
const char* vu_last_error(void) { return vu_get_error(); }

int vu_model_validate(const vu_config* cfg) { return cfg ? validate(*cfg) : VU_EINVAL; }
long long vu_model_param_elems(const vu_config* cfg) {
  Plan pl;
  if (!cfg || build_plan(*cfg, pl) != VU_OK) return -1;
  return pl.total;
}
int vu_model_num_params(const vu_config* cfg) {
  Plan pl;
  if (!cfg || build_plan(*cfg, pl) != VU_OK) return -1;
  return (int)pl.table.size();
}
int vu_model_param_table(const vu_config* cfg, vu_param_entry* out, int capacity) {
  Plan pl;
  if (!cfg) return VU_EINVAL;
  VU_TRY(build_plan(*cfg, pl));
  VU_REQUIRE((int)pl.table.size() <= capacity, "param table: capacity %d < %d", capacity, (int)pl.table.size());
  memcpy(out, pl.table.data(), pl.table.size() * sizeof(vu_param_entry));
  return (int)pl.table.size();
}
int vu_model_num_attn(const vu_config* cfg) {
  Plan pl;
  if (!cfg || build_plan(*cfg, pl) != VU_OK) return -1;
  return pl.nattn;
}
size_t vu_model_workspace_bytes_ex(const vu_config* cfg, int B, int training) {
  Plan pl;
  if (!cfg || B <= 0 || build_plan(*cfg, pl) != VU_OK) return 0;
  ModelWS w;
  carve_model(pl, B, nullptr, w, training);
  return w.bytes;
}
size_t vu_model_workspace_bytes(const vu_config* cfg, int B) { return vu_model_workspace_bytes_ex(cfg, B, 1); }
size_t vu_model_pcache_bytes(const vu_config* cfg, int B) {
  Plan pl;
  if (!cfg || B <= 0 || build_plan(*cfg, pl) != VU_OK) return 0;
  ModelWS w;
  carve_model(pl, B, nullptr, w, 1);
  return w.pcache_bytes;
}

// Diagnostic: the carve of the model workspace as text, one "name offset bytes" line per buffer in carve order (the forward's
// buffers come in execution order), for tools that diff two runs' workspaces (tools/nondet_check.py --ws-diff).  Returns the
// number of characters the full text needs (snprintf convention); `out` receives at most cap - 1 of them.
int vu_model_workspace_describe(const vu_config* cfg, int B, char* out, int cap) {
  Plan pl;
  if (!cfg || B <= 0 || build_plan(*cfg, pl) != VU_OK) return -1;
  ModelWS w;
  char* const fake = reinterpret_cast<char*>(4096);            // (a null base makes every take() return null: carve from a fake one)
  carve_model(pl, B, fake, w, 1);
  std::vector<std::pair<std::string, size_t>> ents;
  auto add = [&](const std::string& n, const void* p) { if (p) ents.push_back({n, (size_t)((const char*)p - fake)}); };
  auto add_attn = [&](const std::string& pre, const AttnBuf& a) {
    add(pre + "q", a.q); add(pre + "k", a.k); add(pre + "v", a.v); add(pre + "O", a.O); add(pre + "Ps", a.Ps); add(pre + "Ah", a.Ah);
    add(pre + "stats", a.stats); add(pre + "lse2", a.lse2); add(pre + "rinv", a.rinv); add(pre + "delta", a.delta); add(pre + "rinvb", a.rinvb);
    add(pre + "pk", a.pk); add(pre + "pc", a.pc); add(pre + "qp", a.qp); add(pre + "kp", a.kp); add(pre + "vp", a.vp); add(pre + "Op", a.Op);
  };
  auto add_block = [&](const std::string& pre, const BlockBuf& b) {
    add_attn(pre + "attn.", b.at);
    add(pre + "z1", b.z1); add(pre + "x1", b.x1); add(pre + "hpre", b.hpre); add(pre + "hact", b.hact); add(pre + "z2", b.z2);
    add(pre + "out", b.out); add(pre + "ln1s", b.ln1s); add(pre + "ln2s", b.ln2s);
  };
  ents.push_back({"tok0", 0});
  for (size_t i = 0; i < w.enc.size(); ++i) add_block("enc" + std::to_string(i) + ".", w.enc[i]);
  for (size_t i = 0; i < w.bot.size(); ++i) add_block("bot" + std::to_string(i) + ".", w.bot[i]);
  for (size_t i = 0; i < w.dec.size(); ++i) add_block("dec" + std::to_string(i) + ".", w.dec[i]);
  for (size_t j = 0; j < w.skip.size(); ++j) add_attn("skip" + std::to_string(j) + ".", w.skip[j]);
  for (size_t j = 0; j < w.skip_out.size(); ++j) {
    add("skip_out" + std::to_string(j), w.skip_out[j]); add("down_out" + std::to_string(j), w.down_out[j]); add("up_out" + std::to_string(j), w.up_out[j]);
  }
  add("img", w.img); add("y_attn", w.y_attn); add("f_ff", w.f_ff);
  add("gx0", w.gx0); add("gx1", w.gx1); add("ga", w.ga); add("gb", w.gb); add("gc", w.gc); add("gh", w.gh);
  add("asc.dO", w.asc.dO); add("asc.dq", w.asc.dq); add("asc.dk", w.asc.dk); add("asc.dv", w.asc.dv); add("asc.dA", w.asc.dA); add("asc.pad", w.asc.pad);
  for (size_t j = 0; j < w.dskip.size(); ++j) add("dskip" + std::to_string(j), w.dskip[j]);
  add("partials", w.partials); add("lnp", w.lnp); add("lnp2", w.lnp2); add("wgs", w.wgs); add("wga", w.wga);
  std::string text;
  for (size_t i = 0; i < ents.size(); ++i) {
    const size_t end = i + 1 < ents.size() ? ents[i + 1].second : w.bytes;
    text += ents[i].first + " " + std::to_string(ents[i].second) + " " + std::to_string(end - ents[i].second) + "\n";
  }
  if (out && cap > 0) { const size_t n = std::min(text.size(), (size_t)cap - 1); memcpy(out, text.data(), n); out[n] = 0; }
  return (int)text.size();
}

int vu_model_prefers_eager(const vu_config* cfg, int B) {
  Plan pl;
  if (!cfg || B <= 0 || build_plan(*cfg, pl) != VU_OK) return 0;
  const int dt = cfg->dtype;
  for (const Level& L : pl.lv) {
    AttnDims d{dt, B, L.N, L.D, cfg->num_heads, cfg->num_channels, L.s, L.ld, 1, flash_switch()};
    if (flash_on(d) && vu_flash_tail_overlap(B, L.N, cfg->num_heads)) return 1;
  }
  return 0;
}

int vu_model_forward(const vu_config* cfg, const float* params, const void* shadow, float* bn_state, const float* x,
                     float* y, void* ws, size_t ws_bytes, int B, int training, uint64_t seed, const uint32_t* rng_salt,
                     void* stream) {
  VU_REQUIRE(cfg && params && bn_state && x && y && ws && B > 0, "vu_model_forward: null argument");
  Plan pl;
  VU_TRY(build_plan(*cfg, pl));
  VU_REQUIRE(cfg->dtype == 0 || shadow, "vu_model_forward: bf16 mode needs the bf16 shadow arena");
  ModelWS w;
  carve_model(pl, B, (char*)ws, w, training);
  if (w.bytes > ws_bytes) { vu_set_error("workspace too small: need %zu, have %zu", w.bytes, ws_bytes); return VU_EWORKSPACE; }
  Ctx cx{&pl, B, params, shadow, bn_state, nullptr, training, seed, rng_salt, (hipStream_t)stream, &w};
  return model_forward(cx, x, y);
}

static int run_backward(const vu_config* cfg, const float* params, const void* shadow, const float* bn_state, float* grads,
                        const float* dy, float* dx, void* ws, size_t ws_bytes, int B, int training, uint64_t seed,
                        const uint32_t* rng_salt, int stage, int first, int last, void* stream) {
  Plan pl;
  VU_TRY(build_plan(*cfg, pl));
  VU_REQUIRE(cfg->dtype == 0 || shadow, "vu_model_backward: bf16 mode needs the bf16 shadow arena");
  ModelWS w;
  carve_model(pl, B, (char*)ws, w, training);
  if (w.bytes > ws_bytes) { vu_set_error("workspace too small: need %zu, have %zu", w.bytes, ws_bytes); return VU_EWORKSPACE; }
  const int nu = (int)backward_units(pl).size();
  if (stage >= 0) {   // stage API: 0 = all, 1 = head + decoders + skips, 2 = bottleneck, 3 = encoders + positional embedding
    const int n1 = 1 + (int)pl.dec.size() + cfg->depth, n2 = n1 + (int)pl.bot.size();
    first = stage <= 1 ? 0 : (stage == 2 ? n1 : n2);
    last = stage == 0 ? nu - 1 : (stage == 1 ? n1 - 1 : (stage == 2 ? n2 - 1 : nu - 1));
  }
  VU_REQUIRE(first >= 0 && last < nu && first <= last + 1, "vu_model_backward_units: unit range [%d,%d] outside [0,%d)", first, last, nu);
  Ctx cx{&pl, B, params, shadow, (float*)bn_state, grads, training, seed, rng_salt, (hipStream_t)stream, &w};
  vu_gemm_set_scratch(w.wgs, w.wgs_bytes);          // lent for this call: deterministic split-K of the skinny weight gradients
  vu_tsgemm_set_arena(w.wga, w.wga_bytes);          // ... and the arena of their deferred, batched reductions
  int rc = model_backward(cx, dy, dx, first, last);
  if (rc == VU_OK) rc = vu_tsgemm_flush((hipStream_t)stream);
  vu_gemm_set_scratch(nullptr, 0);
  vu_tsgemm_set_arena(nullptr, 0);
  return rc;
}

int vu_model_backward(const vu_config* cfg, const float* params, const void* shadow, const float* bn_state, float* grads,
                      const float* dy, float* dx, void* ws, size_t ws_bytes, int B, int training, uint64_t seed,
                      const uint32_t* rng_salt, int stage, void* stream) {
  VU_REQUIRE(cfg && params && grads && dy && ws && B > 0, "vu_model_backward: null argument");
  VU_REQUIRE(stage >= 0 && stage <= 3, "vu_model_backward: stage must be 0..3");
  return run_backward(cfg, params, shadow, bn_state, grads, dy, dx, ws, ws_bytes, B, training, seed, rng_salt, stage, 0, 0, stream);
}

int vu_model_num_backward_units(const vu_config* cfg) {
  Plan pl;
  if (!cfg || build_plan(*cfg, pl) != VU_OK) return -1;
  return (int)backward_units(pl).size();
}
int vu_model_backward_unit_ranges(const vu_config* cfg, long long* lo_hi, int capacity) {
  Plan pl;
  if (!cfg || !lo_hi) return VU_EINVAL;
  VU_TRY(build_plan(*cfg, pl));
  const std::vector<Unit> u = backward_units(pl);
  VU_REQUIRE((int)u.size() <= capacity, "unit ranges: capacity %d < %d", capacity, (int)u.size());
  for (size_t i = 0; i < u.size(); ++i) unit_range(pl, u[i], lo_hi[2 * i], lo_hi[2 * i + 1]);
  return (int)u.size();
}
int vu_model_backward_units(const vu_config* cfg, const float* params, const void* shadow, const float* bn_state, float* grads,
                            const float* dy, float* dx, void* ws, size_t ws_bytes, int B, int training, uint64_t seed,
                            const uint32_t* rng_salt, int first_unit, int last_unit, void* stream) {
  VU_REQUIRE(cfg && params && grads && dy && ws && B > 0, "vu_model_backward_units: null argument");
  return run_backward(cfg, params, shadow, bn_state, grads, dy, dx, ws, ws_bytes, B, training, seed, rng_salt, -1, first_unit, last_unit,
                      stream);
}

// ---- per-op ----
int vu_retile(int dtype, int in_f32, int out_f32, const void* in, void* out, const float* pos, int B, int C, int im,
              int s_in, int s_out, void* stream) {
  return vu_k_retile(dtype, in_f32, out_f32, in, out, pos, B, C, im, s_in, s_out, (hipStream_t)stream);
}
int vu_retile_add(int dtype, const void* in, const void* add, void* out, int B, int C, int im, int s_in, int s_out, void* stream) {
  VU_REQUIRE(in && add && out, "vu_retile_add: null argument");
  VU_REQUIRE(add != out && in != out, "vu_retile_add: the output must not alias an input");
  return vu_k_retile(dtype, 0, 0, in, out, nullptr, B, C, im, s_in, s_out, (hipStream_t)stream, add);
}
int vu_conv3x3_fwd(int dtype, int out_f32, const void* in, const float* w, const float* bias, void* out, long long npatch,
                   int C, int s, void* stream) {
  return vu_k_conv3x3_fwd(dtype, out_f32, in, w, bias, out, npatch, C, s, (hipStream_t)stream);
}
int vu_conv3x3_bwd(int dtype, int dout_f32, const void* dout, const void* in, const float* w, const void* add, void* din,
                   float* dw, float* dbias, long long npatch, int C, int s, void* stream) {
  if (dw) VU_TRY(vu_k_conv3x3_wgrad(dtype, dout_f32, dout, in, dw, dbias, npatch, C, s, (hipStream_t)stream));
  if (din) VU_TRY(vu_k_conv3x3_dgrad(dtype, dout_f32, dout, w, add, din, npatch, C, s, (hipStream_t)stream));
  return VU_OK;
}

int vu_conv3x3_qkv_fwd(int dtype, const void* xq, const void* xkv, const float* wq, const float* wk, const float* wv, void* q,
                       void* k, void* v, long long npatch, int C, int s, void* stream) {
  VU_REQUIRE(xq && xkv && wq && wk && wv && q && k && v, "vu_conv3x3_qkv_fwd: null argument");
  return vu_k_conv3x3_qkv_fwd(dtype, xq, xkv, wq, wk, wv, q, k, v, npatch, C, s, (hipStream_t)stream);
}
int vu_conv3x3_qkv_dgrad(int dtype, const void* dq, const void* dk, const void* dv, const float* wq, const float* wk,
                         const float* wv, const void* add_q, const void* add_kv, void* dxq, void* dxkv, long long npatch, int C,
                         int s, void* stream) {
  VU_REQUIRE(dq && dk && dv && wq && wk && wv && dxq, "vu_conv3x3_qkv_dgrad: null argument");
  return vu_k_conv3x3_qkv_dgrad(dtype, dq, dk, dv, wq, wk, wv, add_q, add_kv, dxq, dxkv, npatch, C, s, (hipStream_t)stream);
}
int vu_conv3x3_qkv_wgrad(int dtype, const void* dq, const void* dk, const void* dv, const void* xq, const void* xkv, float* dwq, float* dwk,
                         float* dwv, void* scratch, size_t scratch_bytes, long long npatch, int C, int s, void* stream) {
  VU_REQUIRE(dq && dk && dv && xq && xkv && dwq && dwk && dwv, "vu_conv3x3_qkv_wgrad: null argument");
  if (scratch) vu_gemm_set_scratch(scratch, scratch_bytes);       // lent for this call: per-block partial sums, added in a fixed order
  const int rc = vu_k_conv3x3_qkv_wgrad(dtype, dq, dk, dv, xq, xkv, dwq, dwk, dwv, npatch, C, s, (hipStream_t)stream);
  if (scratch) vu_gemm_set_scratch(nullptr, 0);
  return rc;
}

static void carve_attn_ws(Bump& bp, const AttnDims& d, AttnBuf& a, AttnScratch& sc, void** dzbuf) {
  carve_attn(bp, d, a);
  if (!a.pk) a.pk = bp.takef((size_t)d.B * d.N * flash_D(d));      // the op's form is chosen after the carve (test switch)
  if (!a.pc) { const size_t pcb = vu_flash_pcache_bytes(d.B, d.N, flash_D(d), d.H); if (pcb) a.pc = bp.take(pcb); }
  if (flash_padded(d) && !a.qp) {
    const size_t actp = (size_t)d.B * d.N * flash_D(d) * esize(d.dtype);
    a.qp = bp.take(actp); a.kp = bp.take(actp); a.vp = bp.take(actp); a.Op = bp.take(actp);
  }
  sc.pad = flash_padded(d) ? bp.take(4 * vu_align_up((size_t)d.B * d.N * flash_D(d) * esize(d.dtype), 256)) : nullptr;
  const size_t act = (size_t)d.B * d.N * d.D * esize(d.dtype);
  const size_t map = (size_t)d.B * d.H * d.N * d.ld * esize(d.dtype);
  sc.dO = bp.take(act); sc.dq = bp.take(act); sc.dk = bp.take(act); sc.dv = bp.take(act); sc.dA = bp.take(map);
  *dzbuf = bp.take(act);
  sc.nblocks = stats_blocks(d);
  sc.partials = bp.takef(attn_partials_floats(d));
}
size_t vu_attn_workspace_bytes(int dtype, int B, int N, int D, int H) {
  AttnDims d{dtype, B, N, D, H, 1, 4, round_up(N, 8)};
  Bump bp{nullptr, 0};
  AttnBuf a; AttnScratch sc; void* dz;
  carve_attn_ws(bp, d, a, sc, &dz);
  return vu_align_up(bp.off, 256);
}
static int attn_dims(int dtype, int B, int N, int D, int H, int C, AttnDims& d) {
  VU_REQUIRE(B > 0 && N > 0 && D > 0 && H > 0 && C > 0 && D % H == 0 && D % C == 0, "attention: bad dims");
  int s = 1;
  while ((long long)s * s * C < D) ++s;
  VU_REQUIRE((long long)s * s * C == D && s % 4 == 0, "attention: D must be C*s*s with s a multiple of 4");
  d = AttnDims{dtype, B, N, D, H, C, s, round_up(N, 8)};
  return VU_OK;
}
int vu_attn_forward(int dtype, const vu_attn_params* prm, const void* xq, const void* xkv, void* y, void* map_out,
                    void* ws, size_t ws_bytes, int B, int N, int D, int H, int C, float attn_drop, float proj_drop,
                    int training, uint64_t seed, uint64_t stream_id, void* stream) {
  AttnDims d;
  VU_TRY(attn_dims(dtype, B, N, D, H, C, d));
  Bump bp{(char*)ws, 0};
  AttnBuf a; AttnScratch sc; void* dz;
  carve_attn_ws(bp, d, a, sc, &dz);
  if (bp.off > ws_bytes) { vu_set_error("attention workspace too small"); return VU_EWORKSPACE; }
  // test switch: the stand-alone op in the model path's centred-map form (only when the map itself is not asked for)
  d.centered = (!map_out && attn_form().centered) ? 1 : 0;
  d.flash = (!map_out && flash_switch_op()) ? 1 : 0;
  VU_TRY(attn_forward(d, *prm, xq, xkv, y, a, sc.partials, attn_drop, proj_drop, training, seed, stream_id, nullptr,
                      (hipStream_t)stream));
  if (map_out) {  // the attn_next tensor of model.py:160 as (B,H,N,N) without row padding
    const size_t es = esize(dtype);
    hipError_t e = hipMemcpy2DAsync(map_out, (size_t)N * es, a.Ah, (size_t)d.ld * es, (size_t)N * es, (size_t)B * H * N,
                                    hipMemcpyDeviceToDevice, (hipStream_t)stream);
    if (e != hipSuccess) { vu_set_error("map copy: %s", hipGetErrorString(e)); return VU_ELAUNCH; }
  }
  return VU_OK;
}
int vu_attn_backward(int dtype, const vu_attn_params* prm, const vu_attn_grads* grd, const void* xq, const void* xkv,
                     const void* dy, void* dxq, void* dxkv, void* ws, size_t ws_bytes, int B, int N, int D, int H, int C,
                     float attn_drop, float proj_drop, int training, uint64_t seed, uint64_t stream_id, void* stream) {
  AttnDims d;
  VU_TRY(attn_dims(dtype, B, N, D, H, C, d));
  Bump bp{(char*)ws, 0};
  AttnBuf a; AttnScratch sc; void* dzb;
  carve_attn_ws(bp, d, a, sc, &dzb);
  if (bp.off > ws_bytes) { vu_set_error("attention workspace too small"); return VU_EWORKSPACE; }
  d.centered = attn_form().centered;      // must match what the forward call used (test switch)
  d.flash = flash_switch_op();
  const void* dz = dy;
  vu_rng rp = vu_make_rng(seed, 2 * stream_id + 1, training ? proj_drop : 0.f);
  if (rp.thr != 0) { VU_TRY(vu_k_dropout(dtype, dy, dzb, (long long)B * N * D, rp, (hipStream_t)stream)); dz = dzb; }
  return attn_backward(d, *prm, *grd, xq, xkv, dz, nullptr, nullptr, dxq, dxkv, a, sc, attn_drop, training, seed, stream_id, nullptr,
                       (hipStream_t)stream);
}

size_t vu_layernorm_workspace_floats(int B, long long P) {
  return (size_t)B * vu_ln_nchunks(P) * 3 + (size_t)B * vu_ln_nbchunks(P) * 2 + 64;
}
int vu_add_layernorm_fwd(int dtype, const void* a, const void* x, void* z, const float* w, const float* b, void* y,
                         float* ws, float* stats, int B, long long P, void* stream) {
  return vu_k_add_ln_fwd(dtype, a, x, z, w, b, y, ws, stats, B, P, 1e-5f, (hipStream_t)stream);
}
int vu_layernorm_bwd(int dtype, const void* dy, const void* z, const float* w, const float* stats, float* dw, float* db,
                     float* ws, void* dz, int B, long long P, void* stream) {
  vu_rng none = vu_make_rng(0, 0, 0.f);
  return vu_k_ln_bwd(dtype, dy, z, w, stats, dw, db, ws, dz, nullptr, none, B, P, (hipStream_t)stream);
}

int vu_ff_forward(int dtype, const void* x, const void* w1, const float* b1, const void* w2, const float* b2, void* hpre,
                  void* hact, void* y, long long rows, int D, int hid, float linear_drop, int training, uint64_t seed,
                  uint64_t stream_id, void* stream) {
  VU_REQUIRE(x && w1 && b1 && w2 && b2 && hpre && hact && y && rows > 0 && D > 0 && hid > 0, "vu_ff_forward: bad argument");
  VU_REQUIRE(linear_drop >= 0.f && linear_drop < 1.f, "dropout must be in [0,1)");
  FFDims f{dtype, rows, D, hid};
  return ff_forward(f, x, w1, b1, w2, b2, hpre, hact, y, nullptr, linear_drop, training, seed, stream_id, nullptr, (hipStream_t)stream);
}
int vu_ff_backward(int dtype, const void* x, const void* w1, const void* w2, const void* hpre, const void* hact, const void* dy,
                   void* dx, float* dw1, float* db1, float* dw2, float* db2, void* scratch, long long rows, int D, int hid,
                   float linear_drop, int training, uint64_t seed, uint64_t stream_id, void* stream) {
  VU_REQUIRE(x && w1 && w2 && hpre && hact && dy && dx && dw1 && db1 && dw2 && db2 && scratch, "vu_ff_backward: null argument");
  FFDims f{dtype, rows, D, hid};
  const size_t es = esize(dtype);
  char* sc = (char*)scratch;
  void* gh = sc;                                             // (rows, hid)
  void* dym = sc + vu_align_up((size_t)rows * hid * es, 256);  // (rows, D): dy under the output dropout mask
  const void* dyu = dy;
  vu_rng rf = vu_make_rng(seed, VU_FF_STREAM(stream_id, 1), training ? linear_drop : 0.f);
  if (rf.thr != 0) { VU_TRY(vu_k_dropout(dtype, dy, dym, rows * D, rf, (hipStream_t)stream)); dyu = dym; }
  return ff_backward(f, x, w1, w2, hpre, hact, dyu, nullptr, dx, dw1, db1, dw2, db2, gh, linear_drop, training, seed, stream_id,
                     nullptr, (hipStream_t)stream);
}
size_t vu_ff_scratch_bytes(int dtype, long long rows, int D, int hid) {
  return vu_align_up((size_t)rows * hid * esize(dtype), 256) + vu_align_up((size_t)rows * D * esize(dtype), 256);
}

int vu_gemm(int dtype, int c_float, const void* A, const void* Bm, void* C, int M, int N, int K, long long sAm,
            long long sAk, long long sBk, long long sBn, long long ldc, int Z1, int Z2, long long sA1, long long sA2,
            long long sB1, long long sB2, long long sC1, long long sC2, float alpha, const float* bias, int accumulate,
            void* stream) {
  vu_gemm_args g;
  memset(&g, 0, sizeof(g));
  g.A = A; g.B = Bm; g.C = C; g.M = M; g.N = N; g.K = K; g.sAm = sAm; g.sAk = sAk; g.sBk = sBk; g.sBn = sBn; g.ldc = ldc;
  g.Z1 = Z1; g.Z2 = Z2; g.sA1 = sA1; g.sA2 = sA2; g.sB1 = sB1; g.sB2 = sB2; g.sC1 = sC1; g.sC2 = sC2;
  g.alpha = alpha; g.bias = bias; g.accumulate = accumulate;
  return vu_gemm_launch(dtype, c_float, g, (hipStream_t)stream);
}

int vu_mse_loss(const float* out, const float* target, float* dout, float* loss, float* partials, long long n,
                float grad_scale, void* stream) {
  return vu_k_mse(out, target, dout, loss, partials, n, grad_scale, (hipStream_t)stream);
}
int vu_adamw(float* params, const float* grads, float* m, float* v, void* shadow_bf16, long long n, const float* hyper,
             int* step, float grad_scale, void* stream) {
  return vu_k_adamw(params, grads, m, v, shadow_bf16, n, hyper, step, grad_scale, (hipStream_t)stream);
}
int vu_cast_bf16(const float* in, void* out, long long n, void* stream) {
  return vu_k_cast_bf16(in, out, n, (hipStream_t)stream);
}
int vu_colsum(int dtype, const void* in, float* out, long long rows, int ncols, long long ld, void* stream) {
  VU_REQUIRE(dtype == 0 || dtype == 1, "dtype must be 0 (fp32) or 1 (bf16)");
  VU_REQUIRE(rows >= 0 && ncols >= 0 && ld >= ncols, "colsum: bad shape");
  return vu_k_colsum(dtype, in, out, rows, ncols, ld, (hipStream_t)stream);
}
int vu_round_e4m3(int dtype, void* x, long long n, void* stream) {
  VU_REQUIRE(dtype == 0 || dtype == 1, "dtype must be 0 (fp32) or 1 (bf16)");
  return vu_k_round_e4m3(dtype, x, nullptr, nullptr, n, (hipStream_t)stream);
}

}  // extern "C"
