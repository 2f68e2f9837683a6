// Ceiling of the probability-cache stream (csrc/vu_flash.hip "probability cache"): B x nt x nt tiles of 4 KB; a wave streams one
// ROW of nt tiles (query-major sweeps: contiguous) or one COLUMN (key-major sweeps: 4 KB pieces, stride nt tiles), 4 x 1 KB loads
// per tile, DEPTH tiles in flight per wave, plain or non-temporal loads, 4 waves per workgroup, 13 workgroups per sample as the
// sweeps launch them.  LDS padding sets the workgroups per CU (2 or 3 or 4).
// build: hipcc -O3 --offload-arch=gfx950 -o tile_stream_probe tile_stream_probe.hip ; run: ./tile_stream_probe [B=64] [nt=49]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef __attribute__((ext_vector_type(4))) unsigned u4;

template <int DEPTH, bool NT, bool COL>
__global__ __launch_bounds__(256) void stream(const u4* __restrict__ p, int B, int nt, unsigned* out) {
  extern __shared__ unsigned char pad[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int per = (nt + 3) / 4;
  const int id = blockIdx.x, x = id & 7, r = id >> 3;
  const int b = (B & 7) == 0 ? x + 8 * (r / per) : id / per, grp = (B & 7) == 0 ? r % per : id % per;
  const int t = grp * 4 + wave;
  if (t >= nt) return;
  auto tile = [&](int s) {
    const int ss = s < nt ? s : nt - 1;
    const long long ti = COL ? ((long long)b * nt + ss) * nt + t : ((long long)b * nt + t) * nt + ss;
    return p + ti * 256 + lane;
  };
  u4 v[DEPTH][4];
  unsigned acc = 0;
#pragma unroll
  for (int d = 0; d < DEPTH; ++d)
#pragma unroll
    for (int i = 0; i < 4; ++i) v[d][i] = NT ? __builtin_nontemporal_load(tile(d) + i * 64) : tile(d)[i * 64];
  for (int s = 0; s < nt; s += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
#pragma unroll
      for (int i = 0; i < 4; ++i) acc ^= v[d][i].x ^ v[d][i].y ^ v[d][i].z ^ v[d][i].w;
#pragma unroll
      for (int i = 0; i < 4; ++i) v[d][i] = NT ? __builtin_nontemporal_load(tile(s + d + DEPTH) + i * 64) : tile(s + d + DEPTH)[i * 64];
    }
  }
  if (acc == 0x12345678u) out[0] = acc + pad[0];
}

template <int DEPTH, bool NT, bool COL>
int run(const u4* buf, int B, int nt, unsigned* out, int lds, const char* name) {
  auto k = stream<DEPTH, NT, COL>;
  CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  const int grid = B * ((nt + 3) / 4);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, 0, buf, B, nt, out);
  CK(hipEventRecord(e0, 0));
  const int reps = 10;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, 0, buf, B, nt, out);
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double bytes = (double)B * nt * nt * 4096.0;
  printf("%-34s lds %6d  %8.1f us  %6.2f TB/s\n", name, lds, ms / reps * 1e3, bytes / (ms / reps * 1e-3) / 1e12);
  return 0;
}

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 64, nt = argc > 2 ? atoi(argv[2]) : 49;
  const size_t bytes = (size_t)B * nt * nt * 4096;
  u4* buf; unsigned* out;
  CK(hipMalloc(&buf, bytes + 65536)); CK(hipMalloc(&out, 64));
  CK(hipMemset(buf, 1, bytes + 65536));
  printf("B %d, %d x %d tiles: %.0f MB\n", B, nt, nt, bytes / 1e6);
  for (int lds : {80 * 1024, 53 * 1024, 40 * 1024, 20 * 1024}) {      // 2, 3, 4, 8 workgroups per CU
    run<1, false, false>(buf, B, nt, out, lds, "row  depth 1 plain");
    run<1, true, false>(buf, B, nt, out, lds, "row  depth 1 nt");
    run<2, false, false>(buf, B, nt, out, lds, "row  depth 2 plain");
    run<2, true, false>(buf, B, nt, out, lds, "row  depth 2 nt");
    run<4, true, false>(buf, B, nt, out, lds, "row  depth 4 nt");
    run<1, true, true>(buf, B, nt, out, lds, "col  depth 1 nt");
    run<2, false, true>(buf, B, nt, out, lds, "col  depth 2 plain");
    run<2, true, true>(buf, B, nt, out, lds, "col  depth 2 nt");
    run<4, true, true>(buf, B, nt, out, lds, "col  depth 4 nt");
  }
  return 0;
}
