"""Regenerates DESIGN.md section 5 from tools/design_section5.md.in; copies the bench lines / trace summaries of tools/gpu_round2.sh from gpurun_out/ into profiles/ (tracked) and fills the
R2_* placeholders of DESIGN.md section 5 from them.  usage: python tools/record_round.py [trace_tag]"""
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r02d"
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(R, "gpurun_out"), os.path.join(R, "profiles")


def line(name):
    with open(os.path.join(G, name)) as f:
        return json.loads(f.read().strip().splitlines()[-1])


lines = {}
for key, fn in [("base", "r02_bench_base.json"), ("lite", "r02_bench_lite_32.json"), ("large", "r02_bench_large_16.json"),
                ("seg32", "r02_bench_seg512_32.json"), ("segst", "r02_bench_seg512_32_storage.json"), ("seg8", "r02_bench_seg512_8.json"), ("b16", "r02_bench_base_16.json"),
                ("b32", "r02_bench_base_32.json"), ("b128", "r02_bench_base_128.json")]:
    lines[key] = line(fn)
    shutil.copy(os.path.join(G, fn), os.path.join(P, fn))
for src, dst in [(f"{tag}_kernel_stats.csv", "r02_kernel_stats.csv"), (f"{tag}_by_grid.csv", "r02_by_grid.csv")]:
    shutil.copy(os.path.join(G, src), os.path.join(P, dst))
b = lines["base"]
roof = b["roofline"]
sub = {
    "R2_VALUE": f"{b['value']:.0f}", "R2_MS": f"{b['ms_per_step']:.2f}",
    "R2_LITE_MS": f"{lines['lite']['ms_per_step']:.1f}", "R2_LITE": f"{lines['lite']['value']:.0f}",
    "R2_LARGE_MS": f"{lines['large']['ms_per_step']:.1f}", "R2_LARGE": f"{lines['large']['value']:.0f}",
    "R2_SEG32_MS": f"{lines['seg32']['ms_per_step']:.1f}", "R2_SEG32": f"{lines['seg32']['value']:.0f}",
    "R2_SEG8_MS": f"{lines['seg8']['ms_per_step']:.1f}", "R2_SEG8": f"{lines['seg8']['value']:.0f}",
    "R2_SEGST_MS": f"{lines['segst']['ms_per_step']:.1f}", "R2_SEGST": f"{lines['segst']['value']:.0f}",
    "R2_B16_MS": f"{lines['b16']['ms_per_step']:.2f}", "R2_B16": f"{lines['b16']['value']:.0f}",
    "R2_B32_MS": f"{lines['b32']['ms_per_step']:.2f}", "R2_B32": f"{lines['b32']['value']:.0f}",
    "R2_B128_MS": f"{lines['b128']['ms_per_step']:.2f}", "R2_B128": f"{lines['b128']['value']:.0f}",
    "R2_HOSTF32": f"{b['host_input']['float32_chw']:.0f}", "R2_HOSTU8": f"{b['host_input']['uint8_hwc_device_pipeline']:.0f}",
    "R2_CPU": f"{b['cpu_baseline']['value']:.1f}",
    "R2_DQX_US": f"{roof['avg_launch_us']:.0f}", "R2_DQX_TF": f"{roof['achieved']:.0f}", "R2_DQX_FRAC": f"{100 * roof['frac']:.1f} %",
    "R2_STEP_FRAC": f"{100 * roof['step_mfma_frac']:.1f} %",
}
d = os.path.join(R, "DESIGN.md")
s = open(d).read()
sec = open(os.path.join(R, "tools", "design_section5.md.in")).read()       # section 5 with R2_* placeholders
for k in sorted(sub, key=len, reverse=True):
    sec = sec.replace(k, sub[k])
a, b = s.index("## 5. Measurement, round 2"), s.index("## 5b. Measurement, round 1")
open(d, "w").write(s[:a] + sec + s[b:])
print(json.dumps(sub, indent=1))
print("dominant kernel:", roof["kernel"], "traffic:", roof.get("traffic"), roof.get("traffic_source"))
