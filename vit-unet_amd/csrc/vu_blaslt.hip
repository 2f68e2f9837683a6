// Plain big GEMMs through hipBLASLt (round 3).  The Linear layers of levels 0 / 1 (3072 x 3072 and 768 x 768 weights over
// 3136 / 12544 tokens; model.py:102-108,147-148) with NO fused epilogue beyond a bias - the projections' data gradients, the
// forward when no dropout rides on it - are exactly what the vendor library is tuned for: measured on this box with random
// operands, M 3136 N 3072 K 3072: vu_gemm 126 us (469 TFLOP/s) against 52.7 us (1122); M 12544 N 768 K 768: 34.8 against
// 28.0 forward, 32.8 against 20.8 for the data gradient (tools/gemm_lib_compare.py).  Everything with a fused epilogue
// (GELU, residual addend; dropout and bias-gradient column sums below the 3072 x 3072 class), every batched / per-head
// product, the skinny shapes and the fp32-accumulating weight gradients below 3072 x 3072 stay on csrc/vu_gemm.hip /
// vu_tsgemm.hip / vu_pgemm.hip.
// Row-major C (M x N) = A B is handed over as the column-major product C^T = B^T A^T.
// ONE 32 MB workspace serves every call: the products are issued on the caller's compute stream, one after the other (the side
// lanes of the backward carry no library product); a caller that put them on several streams at once would need one per stream.
// Plans (descriptors + the heuristic's first algorithm) are cached per shape; a plan is never CREATED while the stream is
// being captured (the heuristic query and the workspace allocation are not capturable) - such a call falls back to vu_gemm.
#include <hipblaslt/hipblaslt.h>
#include <map>
#include <mutex>
#include <stdio.h>
#include <stdlib.h>
#include <tuple>
#include "vu_kernels.h"

namespace {

typedef std::tuple<int, int, int, int, int, long long, long long, long long, int, int, int> Key;
struct Plan {
  hipblasLtMatmulDesc_t desc = nullptr;
  hipblasLtMatrixLayout_t a = nullptr, b = nullptr, c = nullptr;
  hipblasLtMatmulAlgo_t algo;
  size_t ws = 0;
  bool ok = false;
};
struct State {
  std::mutex mu;
  hipblasLtHandle_t handle = nullptr;
  void* ws = nullptr;
  size_t ws_bytes = 0;
  std::map<Key, Plan> plans;
};
State& state() { static State s; return s; }

inline int lt_mode() {        // VU_GEMM_LT: 0 = never, unset / 1 = where eligible
  static const int v = [] { const char* e = getenv("VU_GEMM_LT"); return (e && e[0] == '0') ? 0 : 1; }();
  return v;
}
inline bool post_forced() {      // VU_GEMM_LT_POST=1: every eligible product with a dropout / residual epilogue (measurements)
  static const bool v = [] { const char* e = getenv("VU_GEMM_LT_POST"); return e && e[0] == '1'; }();
  return v;
}
constexpr size_t WS_BYTES = 32u << 20;

bool make_plan(State& s, Plan& p, int m, int n, int k, hipblasOperation_t opA, hipblasOperation_t opB, long long lda, long long ldb,
               long long ldc, int c_float, int has_bias) {
  if (!s.handle && hipblasLtCreate(&s.handle) != HIPBLAS_STATUS_SUCCESS) return false;
  if (!s.ws) {
    if (hipMalloc(&s.ws, WS_BYTES) != hipSuccess) { s.ws = nullptr; return false; }
    s.ws_bytes = WS_BYTES;
  }
  if (hipblasLtMatmulDescCreate(&p.desc, HIPBLAS_COMPUTE_32F, HIP_R_32F) != HIPBLAS_STATUS_SUCCESS) return false;
  int32_t ta = (int32_t)opA, tb = (int32_t)opB;
  hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_TRANSA, &ta, sizeof(ta));
  hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_TRANSB, &tb, sizeof(tb));
  if (has_bias) {
    uint32_t epi = HIPBLASLT_EPILOGUE_BIAS;
    int32_t bt = (int32_t)HIP_R_32F;
    hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_EPILOGUE, &epi, sizeof(epi));
    hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_BIAS_DATA_TYPE, &bt, sizeof(bt));
  }
  // column-major operands: A' is (m x k) or, transposed, stored (k x m); B' is (k x n) or stored (n x k)
  const bool okl =
      hipblasLtMatrixLayoutCreate(&p.a, HIP_R_16BF, opA == HIPBLAS_OP_N ? m : k, opA == HIPBLAS_OP_N ? k : m, lda) == HIPBLAS_STATUS_SUCCESS &&
      hipblasLtMatrixLayoutCreate(&p.b, HIP_R_16BF, opB == HIPBLAS_OP_N ? k : n, opB == HIPBLAS_OP_N ? n : k, ldb) == HIPBLAS_STATUS_SUCCESS &&
      hipblasLtMatrixLayoutCreate(&p.c, c_float ? HIP_R_32F : HIP_R_16BF, m, n, ldc) == HIPBLAS_STATUS_SUCCESS;
  if (!okl) return false;
  hipblasLtMatmulPreference_t pref = nullptr;
  if (hipblasLtMatmulPreferenceCreate(&pref) != HIPBLAS_STATUS_SUCCESS) return false;
  uint64_t wsb = s.ws_bytes;
  hipblasLtMatmulPreferenceSetAttribute(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &wsb, sizeof(wsb));
  hipblasLtMatmulHeuristicResult_t res[1];
  int got = 0;
  const hipblasStatus_t hs = hipblasLtMatmulAlgoGetHeuristic(s.handle, p.desc, p.a, p.b, p.c, p.c, pref, 1, res, &got);
  hipblasLtMatmulPreferenceDestroy(pref);
  if (hs != HIPBLAS_STATUS_SUCCESS || got < 1 || res[0].state != HIPBLAS_STATUS_SUCCESS || res[0].workspaceSize > s.ws_bytes) return false;
  p.algo = res[0].algo;
  p.ws = res[0].workspaceSize;
  p.ok = true;
  return true;
}

// y = dropout(y) + addend in one pass behind the library's product (same mask as the fused epilogue of vu_gemm: element
// index m N + n through vu_keep; thr == 0: no dropout); 8 elements per thread
__global__ __launch_bounds__(256) void lt_post_kernel(bf16_t* __restrict__ y, const bf16_t* __restrict__ addend, long long n8, vu_rng rng_in) {
  const vu_rng rng = vu_rng_resolve(rng_in);
  for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < n8; t += (long long)gridDim.x * blockDim.x) {
    union U8 { uint4 u; bf16_t h[8]; } a, b;
    a.u = *reinterpret_cast<const uint4*>(y + t * 8);
    b.u = addend ? *reinterpret_cast<const uint4*>(addend + t * 8) : make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float v = (float)a.h[e];
      if (rng.thr) v = vu_keep(rng, (uint64_t)(t * 8 + e)) ? v * rng.inv_keep : 0.f;
      a.h[e] = (bf16_t)(v + (float)b.h[e]);
    }
    *reinterpret_cast<uint4*>(y + t * 8) = a.u;
  }
}

// colsum[n] += sum over the rows of in[row ld + n], one workgroup per 64 columns, every row, fixed order: the bias gradient
// behind the library's weight-gradient product must stay bit-reproducible (vu_k_colsum's row blocks end in float atomics)
__global__ __launch_bounds__(1024) void lt_colsum_kernel(const bf16_t* __restrict__ in, float* __restrict__ out, int rows, int ncols, long long ld) {
  __shared__ float red[128][65];
  const int v = threadIdx.x & 7, rl = threadIdx.x >> 3, c0 = blockIdx.x * 64 + v * 8;
  float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (c0 < ncols)
    for (int r = rl; r < rows; r += 128) {
      const bf16x8 x = *reinterpret_cast<const bf16x8*>(in + (long long)r * ld + c0);
#pragma unroll
      for (int i = 0; i < 8; ++i) a[i] += (float)x[i];
    }
#pragma unroll
  for (int i = 0; i < 8; ++i) red[rl][v * 8 + i] = a[i];
  __syncthreads();
  if (threadIdx.x < 64 && blockIdx.x * 64 + threadIdx.x < ncols) {
    float t = 0.f;
    for (int r = 0; r < 128; ++r) t += red[r][threadIdx.x];
    out[blockIdx.x * 64 + threadIdx.x] += t;
  }
}

}  // namespace

// 1 = done by the library, 0 = not eligible (caller goes on with vu_gemm), < 0 = error
int vu_lt_try(const vu_gemm_args& g, int c_float, hipStream_t st) {
  if (!lt_mode()) return 0;
  if (g.Z1 * g.Z2 != 1 || g.act != VU_ACT_NONE || g.aux || g.alpha != 1.f) return 0;
  if (!c_float && g.accumulate) return 0;
  if (c_float && (g.bias || g.dropout || g.addend)) return 0;
  // two epilogues are worth a pass of their own behind the library's product on the 3072 x 3072 WEIGHTS (K and N >= 2048,
  // any token count): the projection dropout + block residual (same mask: element index m N + n; the value is rounded to
  // bf16 once more before the 1 / keep scaling) and the bias-gradient column sums of the weight gradient.  Measured in the
  // Base step, 64 images: forward 130 -> 59 + 12 us, weight gradient 116 -> 80 + 8 us; 16 images: forward 79 -> 38 + 5 us.
  // The 768-class (K = 768) does not pay: 43.7 us against 26.7 + 10.
  const bool post = g.dropout || g.addend;          // one pass over C behind the product
  if (post && !(((g.K >= 2048 && g.N >= 2048) || post_forced()) && g.ldc == g.N && ((long long)g.M * g.N) % 8 == 0 && !(((uintptr_t)g.addend) & 15))) return 0;
  if (g.colsum && !(g.M >= 2048 && g.N >= 2048 && c_float && ((g.colsum_side == 1 && g.sAm == 1 && g.M % 8 == 0 && g.sAk % 8 == 0) ||
                                                                 (g.colsum_side == 2 && g.sBn == 1 && g.N % 8 == 0 && g.sBk % 8 == 0)))) return 0;
  // sizes where the library measured faster: both output extents >= 512, K >= 512, and for the fp32-accumulating weight
  // gradients only the 3072 x 3072 class (768 x 768: vu_gemm 45.7 us, library 71.9)
  if (g.K < 512 || g.N < 512 || g.M < 512) return 0;
  if (c_float ? ((long long)g.M * g.N < (4ll << 20)) : ((long long)g.M * g.N * g.K < (1ll << 32))) return 0;
  // operand storage: A (M,K) row-major (sAk == 1) or (K,M) (sAm == 1); B (K,N) row-major (sBn == 1) or (N,K) (sBk == 1)
  if (!((g.sAk == 1) != (g.sAm == 1)) || !((g.sBn == 1) != (g.sBk == 1))) return 0;
  const bool a_mk = g.sAk == 1, b_kn = g.sBn == 1;
  const long long lda_row = a_mk ? g.sAm : g.sAk, ldb_row = b_kn ? g.sBk : g.sBn;
  if (lda_row < (a_mk ? g.K : g.M) || ldb_row < (b_kn ? g.N : g.K) || g.ldc < g.N) return 0;
  if (((uintptr_t)g.A | (uintptr_t)g.B | (uintptr_t)g.C) & 15) return 0;
  // C^T (N x M, column-major, ld = ldc) = op(B') op(A'):  first operand from B, second from A
  const hipblasOperation_t opA = b_kn ? HIPBLAS_OP_N : HIPBLAS_OP_T;        // B stored (K,N) row-major = (N x K) column-major
  const hipblasOperation_t opB = a_mk ? HIPBLAS_OP_N : HIPBLAS_OP_T;        // A stored (M,K) row-major = (K x M) column-major
  const int m = g.N, n = g.M, k = g.K;
  State& s = state();
  std::lock_guard<std::mutex> lock(s.mu);
  const Key key{m, n, k, (int)opA, (int)opB, ldb_row, lda_row, g.ldc, c_float, g.accumulate, g.bias ? 1 : 0};
  auto it = s.plans.find(key);
  if (it == s.plans.end()) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) return 0;      // no plan yet: not inside a capture
    Plan p;
    if (!make_plan(s, p, m, n, k, opA, opB, ldb_row, lda_row, g.ldc, c_float, g.bias ? 1 : 0)) p.ok = false;
    it = s.plans.emplace(key, p).first;
  }
  Plan& p = it->second;
  if (!p.ok) return 0;
  if (g.bias) {
    const void* bp = g.bias;
    if (hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &bp, sizeof(bp)) != HIPBLAS_STATUS_SUCCESS) return 0;
  }
  const float alpha = 1.f, beta = (c_float && g.accumulate) ? 1.f : 0.f;
  const hipblasStatus_t rs = hipblasLtMatmul(s.handle, p.desc, &alpha, g.B, p.a, g.A, p.b, &beta, g.C, p.c, g.C, p.c, &p.algo, s.ws, p.ws, st);
  if (rs != HIPBLAS_STATUS_SUCCESS) {
    vu_set_error("vu_gemm (hipBLASLt route): hipblasLtMatmul failed with status %d for M=%d N=%d K=%d", (int)rs, g.M, g.N, g.K);
    return VU_ELAUNCH;
  }
  if (vu_prof_on()) {
    static const bool shapes = getenv("VU_PROF_SHAPES") != nullptr;
    char tag[96];
    if (shapes) snprintf(tag, sizeof(tag), "hipblaslt_gemm<%s> M%d N%d K%d%s%s", c_float ? "f32 acc" : "bf16", g.M, g.N, g.K, g.dropout ? " +dropout" : "", g.colsum ? " +colsum" : "");
    else snprintf(tag, sizeof(tag), "hipblaslt_gemm<%s>", c_float ? "f32 acc" : "bf16");
    vu_prof_note(tag, 2.0 * g.M * g.N * g.K, 2.0 * ((double)g.M * g.K + (double)g.K * g.N) + (c_float ? 8.0 : 2.0) * g.M * g.N);
  }
  int rc = vu_check_launch("vu_gemm (hipBLASLt route)");
  if (rc < 0) return rc;
  if (post) {
    const long long n8 = (long long)g.M * g.N / 8;
    vu_rng r = g.rng;
    if (!g.dropout) r.thr = 0;
    const long long grid = (n8 + 255) / 256;
    hipLaunchKernelGGL(lt_post_kernel, dim3((unsigned)(grid > 8192 ? 8192 : grid)), dim3(256), 0, st, (bf16_t*)g.C, (const bf16_t*)g.addend, n8, r);
    if (vu_prof_on()) vu_prof_note("lt_post_kernel", 0.0, (double)n8 * 16 * (g.addend ? 3 : 2));
    rc = vu_check_launch("vu_gemm (hipBLASLt route: dropout / residual pass)");
  }
  if (rc < 0) return rc;
  if (g.colsum) {
    const bool a_side = g.colsum_side == 1;
    const int nc = a_side ? g.M : g.N;
    hipLaunchKernelGGL(lt_colsum_kernel, dim3((unsigned)((nc + 63) / 64)), dim3(1024), 0, st, (const bf16_t*)(a_side ? g.A : g.B), g.colsum, g.K, nc, a_side ? g.sAk : g.sBk);
    if (vu_prof_on()) vu_prof_note("lt_colsum_kernel", 0.0, (double)g.K * nc * 2.0);
    rc = vu_check_launch("vu_gemm (hipBLASLt route: bias-gradient sums)");
  }
  return rc < 0 ? rc : 1;
}
