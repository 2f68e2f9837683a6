// Stand-alone reproducers for the two "wrong data only while ANOTHER PROCESS uses the GPU" faults of round 4, outside the
// library:
//   smem   wave-uniform DATA fetched with scalar (SMEM) loads off a kernel-argument pointer inside a ROLLED loop - the weight path of
//          conv_fwd_kernel in rounds 1 - 3 (273 of 750 launches wrong beside a load process, 0 with the weights staged in LDS)
//   lds    rows of a small table prefetched from an LDS struct inside a ROLLED loop, one row ahead, with a compiler fence - the
//          transposed-mix table of the 4-head delta / dq / dk sweeps (last-bit differences in 1 - 2 % of the tiles under load)
// Each kernel forms sums whose exact value the host knows (small integers in float: every product and sum is exact), every launch is
// checked element for element, and the whole thing runs alone and beside a LOAD PROCESS (a child forked BEFORE this process touches
// the GPU; it initialises HIP itself and keeps the chip busy with a compute + stream kernel for the duration).
//   hipcc --offload-arch=gfx950 -O3 tools/probe/sharing_probe.hip -o tools/probe/sharing_probe
//   ./tools/probe/sharing_probe [launches = 1500] [load seconds = 120; 0 = the caller provides the load] [repetitions of the sum inside a launch = 60]
#include <hip/hip_runtime.h>
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <sys/wait.h>
#include <unistd.h>
#include <vector>

// ---- smem: out[i] = sum_o sum_t w[o][t] * x[(i + t) & mask] with w read by scalar loads inside the rolled (o) loop -----------------
template <int NT>
__global__ __launch_bounds__(256) void smem_kernel(const float* __restrict__ w, const float* __restrict__ x, float* __restrict__ out, int no, int mask, int outer, const int* zero) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  float first = 0.f, worst = 0.f;
#pragma unroll 1
  for (int rep = 0; rep < outer; ++rep) {           // the same sum `outer` times (a launch of milliseconds: long enough to be preempted)
    const int z = *reinterpret_cast<const volatile int*>(zero);      // 0, re-read every repetition: the sum cannot be hoisted
    float acc = 0.f;
#pragma unroll 1
    for (int o = 0; o < no; ++o) {                  // rolled: the s_load of w[o * NT ..] sits INSIDE the loop
      const float* wo = w + o * NT;                 // uniform address: hipcc emits s_load_dwordx{2,4,8}
      float part = 0.f;
#pragma unroll
      for (int t = 0; t < NT; ++t) part = fmaf(wo[t], x[(i + t + o + z) & mask], part);
      acc += part;
    }
    if (rep == 0) first = acc;
    else if (acc != first) worst = acc;             // any repetition that differs from the first is reported instead of it
  }
  out[i] = worst != 0.f ? worst : first;
}

// ---- lds: out[i] = sum_h sum_g tab[h][g] * x[(i + 4 h + g) & mask], the rows of tab prefetched from LDS one row ahead -------------
struct Tab { float a[8 * 8]; float pad[8]; };
#define LDS_FENCE() asm volatile("" ::: "memory")
__global__ __launch_bounds__(256) void lds_kernel(const float* __restrict__ tab, const float* __restrict__ x, float* __restrict__ out, int reps, int mask, int outer, const int* zero) {
  __shared__ Tab tb;
  __shared__ float ballast[12 * 1024];             // 48 KB: the workgroup owns a large LDS allocation, as the sweeps do
  if (threadIdx.x == 0 && outer < 0) ballast[reps] = 1.f;
  for (int k = threadIdx.x; k < 64; k += 256) tb.a[k] = tab[k];
  __syncthreads();
  const int i = blockIdx.x * 256 + threadIdx.x;
  float first = 0.f, worst = 0.f;
#pragma unroll 1
  for (int rep = 0; rep < outer; ++rep) {
  const int z = *reinterpret_cast<const volatile int*>(zero);
  float acc = 0.f;
#pragma unroll 1
  for (int r = 0; r < reps; ++r) {
    LDS_FENCE();
    float cur[8], nxt[8];
#pragma unroll
    for (int g = 0; g < 8; ++g) cur[g] = tb.a[g];
#pragma unroll 1
    for (int h = 0; h < 8; ++h) {                   // rolled head loop: row h + 1 is fetched before the arithmetic of row h
#pragma unroll
      for (int g = 0; g < 8; ++g) nxt[g] = tb.a[((h + 1) & 7) * 8 + g];
      LDS_FENCE();
#pragma unroll
      for (int g = 0; g < 8; ++g) acc = fmaf(cur[g], x[(i + 4 * h + g + r + z) & mask], acc);
#pragma unroll
      for (int g = 0; g < 8; ++g) cur[g] = nxt[g];
    }
  }
  if (rep == 0) first = acc;
  else if (acc != first) worst = acc;
  }
  out[i] = worst != 0.f ? worst : first;
}

// ---- the load: arithmetic + a stream over 256 MB, back to back ----------------------------------------------------------------------
__global__ __launch_bounds__(256) void load_kernel(float* __restrict__ buf, long long n, int iters) {
  __shared__ float lds[14 * 1024];                  // 56 KB per workgroup: the load competes for LDS as a train step does
  lds[threadIdx.x] = (float)threadIdx.x;
  __syncthreads();
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long stride = (long long)gridDim.x * 256;
  for (; i < n; i += stride) {
    float v = buf[i];
    for (int k = 0; k < iters; ++k) v = fmaf(v, 1.0000001f, 1e-7f);
    buf[i] = v + 0.f * lds[(threadIdx.x + 1) & 255];
  }
}
static int load_process(int seconds) {
  float* buf;
  const long long n = 64ll << 20;
  if (hipMalloc(&buf, n * 4) != hipSuccess) return 1;
  (void)hipMemset(buf, 0, n * 4);
  const time_t t0 = time(nullptr);
  while (time(nullptr) - t0 < seconds) {
    for (int k = 0; k < 20; ++k) hipLaunchKernelGGL(load_kernel, dim3(2048), dim3(256), 0, 0, buf, n, 512);
    (void)hipDeviceSynchronize();
  }
  return 0;
}

template <typename F>
static void run_probe(const char* name, int launches, const std::vector<float>& expect, float* d_out, size_t n, F launch) {
  std::vector<float> h(n);
  int bad_launches = 0;
  long long bad_elems = 0;
  for (int l = 0; l < launches; ++l) {
    launch();
    (void)hipMemcpy(h.data(), d_out, n * 4, hipMemcpyDeviceToHost);
    long long b = 0;
    for (size_t i = 0; i < n; ++i) b += h[i] != expect[i];
    if (b) { ++bad_launches; bad_elems += b; }
  }
  printf("  %-5s %d of %d launches wrong (%lld elements)\n", name, bad_launches, launches, bad_elems);
}

static int probe_process(int launches, int outer, const char* title) {
  const int n = 256 * 2048, mask = 4095, no = 27, NT = 9, reps = 6;
  std::vector<float> w(no * NT), tab(64), x(mask + 1), e_smem(n), e_lds(n);
  for (int i = 0; i < no * NT; ++i) w[i] = (float)((i * 7) % 11 - 5);
  for (int i = 0; i < 64; ++i) tab[i] = (float)((i * 5) % 13 - 6);
  for (int i = 0; i <= mask; ++i) x[i] = (float)((i * 3) % 17 - 8);
  for (int i = 0; i < n; ++i) {
    double a = 0;
    for (int o = 0; o < no; ++o) for (int t = 0; t < NT; ++t) a += (double)w[o * NT + t] * x[(i + t + o) & mask];
    e_smem[i] = (float)a;
    double b = 0;
    for (int r = 0; r < reps; ++r) for (int h = 0; h < 8; ++h) for (int g = 0; g < 8; ++g) b += (double)tab[h * 8 + g] * x[(i + 4 * h + g + r) & mask];
    e_lds[i] = (float)b;
  }
  float *dw, *dt, *dx, *dout;
  int* dzero;
  (void)hipMalloc(&dzero, 4); (void)hipMemset(dzero, 0, 4);
  if (hipMalloc(&dw, w.size() * 4) != hipSuccess || hipMalloc(&dt, 64 * 4) != hipSuccess || hipMalloc(&dx, x.size() * 4) != hipSuccess ||
      hipMalloc(&dout, (size_t)n * 4) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
  (void)hipMemcpy(dw, w.data(), w.size() * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(dt, tab.data(), 64 * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice);
  printf("%s\n", title);
  run_probe("smem", launches, e_smem, dout, n, [&] { hipLaunchKernelGGL(smem_kernel<9>, dim3(n / 256), dim3(256), 0, 0, dw, dx, dout, no, mask, outer, dzero); });
  run_probe("lds", launches, e_lds, dout, n, [&] { hipLaunchKernelGGL(lds_kernel, dim3(n / 256), dim3(256), 0, 0, dt, dx, dout, reps, mask, outer, dzero); });
  fflush(stdout);
  return 0;
}

// The parent makes NO HIP call: every process that touches the GPU is a child forked from it and initialises HIP itself.
int main(int argc, char** argv) {
  const int launches = argc > 1 ? atoi(argv[1]) : 1500, secs = argc > 2 ? atoi(argv[2]) : 120, outer = argc > 3 ? atoi(argv[3]) : 60;
  int st = 0;
  pid_t a = fork();
  if (a == 0) _exit(probe_process(launches, outer, "alone on the GPU:"));
  waitpid(a, &st, 0);
  if (secs == 0) {                                         // the caller runs its own load (e.g. a train step loop of the library in python)
    pid_t b = fork();
    if (b == 0) _exit(probe_process(launches, outer, "beside the caller's load:"));
    waitpid(b, &st, 0);
    return 0;
  }
  pid_t load = fork();
  if (load == 0) _exit(load_process(secs));
  sleep(4);                                                // the load child has its context and its first kernels in flight
  pid_t b = fork();
  if (b == 0) _exit(probe_process(launches, outer, "beside a load process:"));
  waitpid(b, &st, 0);
  kill(load, SIGTERM);
  waitpid(load, &st, 0);
  return 0;
}
