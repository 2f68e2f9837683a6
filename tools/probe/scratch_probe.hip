// Does per-lane scratch (private) memory stay intact while ANOTHER PROCESS computes on the same GPU?  (round 4: the 4-head recompute
// sweeps compiled with a register cap - 12 to 68 bytes of scratch per lane inside their tile loops - lost bit-reproducibility under
// GPU sharing and kept it without the cap.)  Every lane fills a dynamically indexed private array (forced into scratch), does some
// arithmetic so that waves are switched, reads it back and counts mismatches.
//   hipcc --offload-arch=gfx950 -O3 tools/probe/scratch_probe.hip -o tools/probe/scratch_probe ; run it beside a load process
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
__global__ __launch_bounds__(256) void probe(unsigned* bad, int iters, int salt) {
  volatile unsigned buf[48];                        // dynamically indexed below: lives in scratch
  const unsigned id = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned errs = 0, acc = id;
  for (int it = 0; it < iters; ++it) {
    for (int i = 0; i < 48; ++i) buf[(i * 7 + it + salt) % 48] = id * 2654435761u + (unsigned)(it * 48 + i);
    for (int k = 0; k < 64; ++k) acc = acc * 1664525u + 1013904223u;            // time for other waves / queues to run
    for (int i = 0; i < 48; ++i) errs += buf[(i * 7 + it + salt) % 48] != id * 2654435761u + (unsigned)(it * 48 + i);
  }
  if (errs || acc == 0xdeadbeefu) atomicAdd(bad, errs ? errs : 1u);
}
int main(int argc, char** argv) {
  const int launches = argc > 1 ? atoi(argv[1]) : 400, iters = argc > 2 ? atoi(argv[2]) : 40;      // (iters 40: ~0.1 ms per launch; 2000: several ms)
  unsigned* bad; (void)hipMalloc(&bad, 4); (void)hipMemset(bad, 0, 4);
  unsigned total = 0, bad_launches = 0;
  for (int l = 0; l < launches; ++l) {
    hipLaunchKernelGGL(probe, dim3(4096), dim3(256), 0, 0, bad, iters, l);
    unsigned h = 0; (void)hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost);
    if (h) { ++bad_launches; total += h; (void)hipMemset(bad, 0, 4); }
  }
  printf("scratch probe: %u of %d launches saw mismatches (%u lane-words)\n", bad_launches, launches, total);
  printf(bad_launches ? "SCRATCH_PROBE MISMATCH\n" : "SCRATCH_PROBE CLEAN\n");
  return 0;
}
