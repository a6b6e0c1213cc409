#!/usr/bin/env python3
"""Dev tool: after tools_refresh.sh ran on the GPU box, keep only the newest run in every gpurun_out profile directory,
summarise the PMC passes and copy the artefacts judged under profiles/ (bench lines get the traffic figure of their own run's
dominant kernel from profiles/pmc_traffic.json, which bench.py also reads)."""
import glob, json, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.chdir(ROOT)


def newest_only(d):
    files = glob.glob(os.path.join(d, "runc", "*_agent_info.csv"))
    if not files:
        return None
    files.sort(key=os.path.getmtime)
    keep = os.path.basename(files[-1]).split("_")[0]
    for f in glob.glob(os.path.join(d, "runc", "*")):
        if not os.path.basename(f).startswith(keep + "_"):
            os.remove(f)
    return keep


for d in ("prof_final", "pmcb_fetch", "pmcb_write", "pmcb_sq", "pmcb_fetch_srgan", "pmcb_write_srgan", "pmcb_sq_srgan"):
    newest_only(os.path.join("gpurun_out", d))
run = lambda *a: subprocess.run([sys.executable, *a], check=True, capture_output=True, text=True).stdout
run("tools/pmc_summarize.py", "srresnet", "gpurun_out/pmcb_fetch", "gpurun_out/pmcb_write")
if glob.glob("gpurun_out/pmcb_fetch_srgan/runc/*counter_collection.csv"):
    run("tools/pmc_summarize.py", "srgan", "gpurun_out/pmcb_fetch_srgan", "gpurun_out/pmcb_write_srgan")
for tag, out in (("pmcb_sq", "r01_pmc_sq_srresnet_b16.txt"), ("pmcb_sq_srgan", "r01_pmc_sq_srgan_b16.txt")):
    f = glob.glob(f"gpurun_out/{tag}/runc/*counter_collection.csv")
    if f:
        open(os.path.join("profiles", out), "w").write(run("tools/pmc_sq_summary.py", f[0]))
shutil.copy(glob.glob("gpurun_out/prof_final/runc/*kernel_stats.csv")[0], "profiles/r01_final_graph_srresnet_b16_kernel_stats.csv")
t = json.load(open("profiles/pmc_traffic.json"))
for w in ("srresnet", "srgan", "srgan_vgg"):
    j = json.load(open(f"gpurun_out/final_bench_{w}.json"))
    j["roofline"]["traffic"] = t.get(w, {}).get(j["roofline"]["kernel"])
    json.dump(j, open(f"profiles/r01_bench_{w}_b16.json", "w"))
    print(w, round(j["value"], 1), "img/s", round(j["ms_per_step"], 4), "ms; dominant", j["roofline"]["kernel"], round(j["roofline"]["frac"], 4),
          "traffic", j["roofline"]["traffic"])
