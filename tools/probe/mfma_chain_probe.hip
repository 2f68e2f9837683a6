// Reproducer for the "accumulator chained from v_mfma_f32_16x16x32_bf16 into v_mfma_f32_16x16x16_bf16" observation of round 3
// (vu_flash.hip, bwd2_dp): the same sum once as a chain through ONE accumulator (K=32 product, then K=16 product, back to back) and
// once as two independent products added on the VALU.  Exact inputs (small integers): both must agree bit for bit.
//   hipcc --offload-arch=gfx950 -O3 tools/probe/mfma_chain_probe.hip -o /tmp/mfma_chain_probe && /tmp/mfma_chain_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
__device__ __forceinline__ s16x4 lo4(const bf16x8& x) { bf16x8 t = x; return __builtin_bit_cast(s16x4, __builtin_shufflevector(t, t, 0, 1, 2, 3)); }
template <int ORDER, int WAIT> __global__ void probe(const float* in, float* out, int reps) {
  const int lane = threadIdx.x;
  bf16x8 a, b, a2, b2;
  for (int i = 0; i < 8; ++i) {
    a[i] = (__bf16)in[(lane * 8 + i) % 97]; b[i] = (__bf16)in[(lane * 5 + i * 3) % 89];
    a2[i] = (__bf16)in[(lane * 3 + i * 7) % 83]; b2[i] = (__bf16)in[(lane + i * 11) % 79];
  }
  f32x4 chain = {0.f, 0.f, 0.f, 0.f}, split = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
  for (int r = 0; r < reps; ++r) {   // the chain alone in its loop: the two matrix instructions are adjacent in the instruction stream
    if (ORDER == 0) {              // K = 32 product, then K = 16 into the same accumulator
      chain = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, chain, 0, 0, 0);
      if (WAIT) asm volatile("s_nop 15\n\ts_nop 15" : "+v"(chain));
      chain = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(lo4(a2), lo4(b2), chain, 0, 0, 0);
    } else if (ORDER == 1) {       // K = 16 first, then K = 32
      chain = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(lo4(a2), lo4(b2), chain, 0, 0, 0);
      if (WAIT) asm volatile("s_nop 15\n\ts_nop 15" : "+v"(chain));
      chain = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, chain, 0, 0, 0);
    } else {                       // control: the same sum as two K = 32 products (second operand pair zero-extended to 8 k-slots)
      chain = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, chain, 0, 0, 0);
      bf16x8 a3 = a2, b3 = b2;
      for (int i = 4; i < 8; ++i) { a3[i] = (__bf16)0.f; b3[i] = (__bf16)0.f; }
      chain = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a3, b3, chain, 0, 0, 0);
    }
    if (WAIT) asm volatile("s_nop 15\n\ts_nop 15" : "+v"(chain));
  }
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  const f32x4 p = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, z, 0, 0, 0);
  const f32x4 q = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(lo4(a2), lo4(b2), z, 0, 0, 0);
  for (int r = 0; r < reps; ++r)
    for (int j = 0; j < 4; ++j) split[j] += p[j] + q[j];
  for (int j = 0; j < 4; ++j) { out[lane * 8 + j] = chain[j]; out[lane * 8 + 4 + j] = split[j]; }
}
int main() {
  float h[128], *din, *dout, ho[512];
  for (int i = 0; i < 128; ++i) h[i] = (float)((i * 37) % 7 - 3);        // integers in [-3, 3]: every product and sum is exact
  (void)hipMalloc(&din, sizeof(h)); (void)hipMalloc(&dout, sizeof(ho));
  (void)hipMemcpy(din, h, sizeof(h), hipMemcpyHostToDevice);
  int bad_total = 0;
  const char* names[3] = {"K=32 -> K=16", "K=16 -> K=32", "K=32 -> K=32 (control)"};
  for (int wait = 0; wait < 2; ++wait)
    for (int order = 0; order < 3; ++order) {
      void (*kern)(const float*, float*, int) = wait ? (order == 0 ? probe<0, 1> : order == 1 ? probe<1, 1> : probe<2, 1>)
                                                     : (order == 0 ? probe<0, 0> : order == 1 ? probe<1, 0> : probe<2, 0>);
      hipLaunchKernelGGL(kern, dim3(1), dim3(64), 0, 0, din, dout, 4);
      (void)hipMemcpy(ho, dout, sizeof(ho), hipMemcpyDeviceToHost);
      int bad = 0;
      for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) bad += ho[l * 8 + j] != ho[l * 8 + 4 + j];
      printf("%-24s %s: %d of 256 accumulator elements differ\n", names[order], wait ? "with 32 wait states between the two" : "as hipcc schedules them          ", bad);
      if (!wait) bad_total += bad;
    }
  printf(bad_total ? "MFMA_CHAIN_PROBE MISMATCH\n" : "MFMA_CHAIN_PROBE CLEAN\n");
  return 0;
}
