// Error text (thread-local) and the optional in-process launch profiler used by bench.py's
// `roofline` leg: when enabled, an event is recorded on the profiled stream after every launch,
// so the time between consecutive events is the device time of one launch (including its
// dispatch gap).  Launchers annotate each launch with its ALGORITHMIC flops / bytes.
// An interval between two events also contains any time the GPU WAITED FOR THE HOST to enqueue the launch: at 16 images per GPU
// a step is ~390 launches of 5 - 20 us and the eager host is slower than that (round 4: the instrumented step charged 48 us of host
// gap to every re-tiling launch).  vu_prof_gate() closes that hole: it enqueues a one-wave kernel that holds the stream for a given
// time, so the host runs a whole step ahead and the launches behind the gate execute back to back.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <map>
#include <string>
#include <vector>
#include "vu_common.h"

static thread_local char g_err[512] = "";

void vu_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
const char* vu_get_error() { return g_err; }

namespace {
struct Rec { std::string tag; double flops, bytes, strict, mapfree; };
bool g_prof = false;
hipStream_t g_stream = nullptr;
std::vector<hipEvent_t> g_events;   // g_events[0] = start marker, g_events[i+1] follows launch i
std::vector<Rec> g_recs;
std::string g_tag;
double g_flops = 0, g_bytes = 0, g_strict = -1, g_mapfree = -1;
std::string g_report;

// holds the stream for `ticks` of the constant 100 MHz counter (bounded: every wave leaves after that time)
__global__ void vu_prof_gate_kernel(long long ticks) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
}

hipEvent_t new_event() {
  hipEvent_t e;
  hipEventCreate(&e);
  return e;
}
}  // namespace

bool vu_prof_on() { return g_prof; }
void vu_prof_note(const char* tag, double flops, double bytes) {
  if (!g_prof) return;
  g_tag = tag; g_flops = flops; g_bytes = bytes; g_strict = -1; g_mapfree = -1;
}
// flops of the launch under SURVEY 8d's rule (the model's own products only: no recomputation, no padding); default = flops
void vu_prof_note_strict(double flops_strict) {
  if (g_prof) g_strict = flops_strict;
}

// bytes of the launch without any attention-map-sized stream (the probability cache is DESIGN traffic, not algorithmic); default = bytes
void vu_prof_note_mapfree(double bytes_mapfree) {
  if (g_prof) g_mapfree = bytes_mapfree;
}

int vu_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    vu_set_error("%s: %s", what, hipGetErrorString(e));
    return VU_ELAUNCH;
  }
  if (g_prof) {
    hipEvent_t ev = new_event();
    hipEventRecord(ev, g_stream);
    g_events.push_back(ev);
    g_recs.push_back(Rec{g_tag.empty() ? std::string(what) : g_tag, g_flops, g_bytes, g_strict < 0 ? g_flops : g_strict, g_mapfree < 0 ? g_bytes : g_mapfree});
    g_tag.clear(); g_flops = 0; g_bytes = 0; g_strict = -1; g_mapfree = -1;
  }
  return VU_OK;
}

extern "C" int vu_prof_enable(void* stream) {
  for (hipEvent_t e : g_events) hipEventDestroy(e);
  g_events.clear(); g_recs.clear();
  g_stream = (hipStream_t)stream;
  g_prof = true;
  hipEvent_t ev = new_event();
  hipEventRecord(ev, g_stream);
  g_events.push_back(ev);
  return VU_OK;
}

// Holds the profiled stream for `usec` microseconds (at most 200 ms) so that the launches enqueued behind it run without host gaps.
// The gate's own interval is not reported.
extern "C" int vu_prof_gate(int usec) {
  if (!g_prof) { vu_set_error("vu_prof_gate: the profiler is not enabled"); return VU_EINVAL; }
  if (usec < 0 || usec > 200000) { vu_set_error("vu_prof_gate: 0 .. 200000 us"); return VU_EINVAL; }
  hipLaunchKernelGGL(vu_prof_gate_kernel, dim3(1), dim3(64), 0, g_stream, (long long)usec * 100);
  g_tag = "(gate)";
  return vu_check_launch("vu_prof_gate");
}

// Stops profiling, waits for the stream and returns a JSON object
// {"<tag>": {"count": n, "ms": total, "flops": total, "bytes": total}, ...}
extern "C" const char* vu_prof_report(void) {
  g_prof = false;
  g_report = "{";
  if (!g_events.empty()) {
    hipEventSynchronize(g_events.back());
    struct Agg { long long n = 0; double ms = 0, flops = 0, bytes = 0, strict = 0, mapfree = 0; };
    std::map<std::string, Agg> agg;
    for (size_t i = 0; i < g_recs.size(); ++i) {
      float ms = 0.f;
      hipEventElapsedTime(&ms, g_events[i], g_events[i + 1]);
      if (g_recs[i].tag == "(gate)") continue;
      Agg& a = agg[g_recs[i].tag];
      a.n += 1; a.ms += ms; a.flops += g_recs[i].flops; a.bytes += g_recs[i].bytes; a.strict += g_recs[i].strict; a.mapfree += g_recs[i].mapfree;
    }
    bool first = true;
    char buf[512];
    for (auto& kv : agg) {
      snprintf(buf, sizeof(buf), "%s\"%s\": {\"count\": %lld, \"ms\": %.6f, \"flops\": %.6e, \"bytes\": %.6e, \"flops_strict\": %.6e, \"bytes_mapfree\": %.6e}",
               first ? "" : ", ", kv.first.c_str(), kv.second.n, kv.second.ms, kv.second.flops, kv.second.bytes, kv.second.strict, kv.second.mapfree);
      g_report += buf;
      first = false;
    }
  }
  g_report += "}";
  for (hipEvent_t e : g_events) hipEventDestroy(e);
  g_events.clear(); g_recs.clear();
  return g_report.c_str();
}
