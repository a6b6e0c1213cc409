#!/usr/bin/env python3
"""Dev tool: after tools/r02_refresh.sh ran on the GPU box, summarise the PMC passes and copy the artefacts judged under profiles/
(r02_final_*): bench lines (with the traffic figure of their own run's dominant kernel family from profiles/pmc_traffic.json,
which bench.py also reads), rocprofv3 kernel-stats CSVs, SQ counter summaries."""
import glob, json, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.chdir(ROOT)
# gpurun merges every refresh into the same local directories: keep only the newest run's files in each
for d in glob.glob("gpurun_out/r02f_*/*/"):
    fs = glob.glob(d + "*")
    newest = max(os.path.getmtime(f) for f in fs)
    for f in fs:
        if os.path.getmtime(f) < newest - 600:
            os.remove(f)
run = lambda *a: subprocess.run([sys.executable, *a], check=True, capture_output=True, text=True).stdout
for wl in ("srgan", "srresnet"):
    print(run("tools/pmc_summarize.py", wl, f"gpurun_out/r02f_pmc_fetch_{wl}", f"gpurun_out/r02f_pmc_write_{wl}"))
    f = glob.glob(f"gpurun_out/r02f_pmc_sq_{wl}/*/*counter_collection.csv")
    open(f"profiles/r02_final_pmc_sq_{wl}_b16.txt", "w").write(run("tools/pmc_sq_summary.py", f[0]))
    shutil.copy(glob.glob(f"gpurun_out/r02f_prof_{wl}/*/*kernel_stats.csv")[0], f"profiles/r02_final_graph_{wl}_b16_kernel_stats.csv")
shutil.copy(glob.glob("gpurun_out/r02f_prof_hr192/*/*kernel_stats.csv")[0], "profiles/r02_final_graph_srgan_hr192_b8_kernel_stats.csv")
t = json.load(open("profiles/pmc_traffic.json"))
for w, key in (("srgan", "srgan"), ("srresnet", "srresnet"), ("srgan_vgg", "srgan"), ("srgan_hr192", None)):
    j = json.load(open(f"gpurun_out/r02f_bench_{w}.json"))
    if key:
        j["roofline"]["traffic"] = t.get(key, {}).get(j["roofline"]["kernel"])
    json.dump(j, open(f"profiles/r02_final_bench_{w}.json", "w"))
    print(w, round(j["value"], 1), "img/s", round(j["ms_per_step"], 4), "ms; dominant", j["roofline"]["kernel"], round(j["roofline"]["frac"], 4),
          "traffic", j["roofline"]["traffic"])
