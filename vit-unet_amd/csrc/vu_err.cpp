// Error text (thread-local) and the optional in-process launch profiler used by bench.py's
// `roofline` leg: when enabled, an event is recorded on the profiled stream after every launch,
// so the time between consecutive events is the device time of one launch (including its
// dispatch gap).  Launchers annotate each launch with its ALGORITHMIC flops / bytes.
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
struct Rec { std::string tag; double flops, bytes; };
bool g_prof = false;
hipStream_t g_stream = nullptr;
std::vector<hipEvent_t> g_events;   // g_events[0] = start marker, g_events[i+1] follows launch i
std::vector<Rec> g_recs;
std::string g_tag;
double g_flops = 0, g_bytes = 0;
std::string g_report;

hipEvent_t new_event() {
  hipEvent_t e;
  hipEventCreate(&e);
  return e;
}
}  // namespace

bool vu_prof_on() { return g_prof; }
void vu_prof_note(const char* tag, double flops, double bytes) {
  if (!g_prof) return;
  g_tag = tag; g_flops = flops; g_bytes = bytes;
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
    g_recs.push_back(Rec{g_tag.empty() ? std::string(what) : g_tag, g_flops, g_bytes});
    g_tag.clear(); g_flops = 0; g_bytes = 0;
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

// Stops profiling, waits for the stream and returns a JSON object
// {"<tag>": {"count": n, "ms": total, "flops": total, "bytes": total}, ...}
extern "C" const char* vu_prof_report(void) {
  g_prof = false;
  g_report = "{";
  if (!g_events.empty()) {
    hipEventSynchronize(g_events.back());
    struct Agg { long long n = 0; double ms = 0, flops = 0, bytes = 0; };
    std::map<std::string, Agg> agg;
    for (size_t i = 0; i < g_recs.size(); ++i) {
      float ms = 0.f;
      hipEventElapsedTime(&ms, g_events[i], g_events[i + 1]);
      Agg& a = agg[g_recs[i].tag];
      a.n += 1; a.ms += ms; a.flops += g_recs[i].flops; a.bytes += g_recs[i].bytes;
    }
    bool first = true;
    char buf[512];
    for (auto& kv : agg) {
      snprintf(buf, sizeof(buf), "%s\"%s\": {\"count\": %lld, \"ms\": %.6f, \"flops\": %.6e, \"bytes\": %.6e}",
               first ? "" : ", ", kv.first.c_str(), kv.second.n, kv.second.ms, kv.second.flops, kv.second.bytes);
      g_report += buf;
      first = false;
    }
  }
  g_report += "}";
  for (hipEvent_t e : g_events) hipEventDestroy(e);
  g_events.clear(); g_recs.clear();
  return g_report.c_str();
}
