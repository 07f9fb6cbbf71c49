#!/usr/bin/env python3
"""Copy the summaries tools/refresh.sh left under gpurun_out/refresh_<tag>/ into profiles/ and rebuild
profiles/traffic_taf_mpx.json from the PMC summary (FETCH_SIZE doubled per MI355X_MICROARCH.md)."""
import csv, json, re, shutil, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
O = f"gpurun_out/refresh_{tag}"
per, cur = {}, None
for line in open(f"{O}/pmc_summary.txt"):
    if not line.startswith(" "):
        cur = line.strip(); continue
    m = re.match(r"\s+(\S+)\s+(\d+)", line)
    if m and cur: per.setdefault(cur, {})[m.group(1)] = int(m.group(2))
tot, out = 0, {}
for k, v in per.items():
    if "FETCH_SIZE" in v and k.startswith("k_"):
        f, w = v["FETCH_SIZE"] * 1024 * 2, v.get("WRITE_SIZE", 0) * 1024
        out[k.split("<")[0]] = {"fetch_bytes_corrected": f, "write_bytes": w}; tot += f + w
json.dump({"workload": "taf_mpx", "hbm_bytes_per_encode": tot, "algorithmic_bytes": 212710400,
           "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes (tools/pmc.sh), per-dispatch averages, KiB x1024, "
                     "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half of a wide coalesced read)",
           "per_kernel": out, "source": f"profiles/{tag}_taf_mpx_pmc_summary.txt"}, open("profiles/traffic_taf_mpx.json", "w"), indent=1)
for a, b in (("pmc_summary.txt", "taf_mpx_pmc_summary.txt"), ("taf_kernel_stats.csv", "bench_kernel_stats.csv"),
             ("hot_kernel_stats.csv", "taf_mpx_hotspot_kernel_stats.csv"), ("gen1_kernel_stats.csv", "taf_gen1_kernel_stats.csv"),
             ("det_kernel_stats.csv", "detector_kernel_stats.csv"), ("bench.json", "bench.json"), ("bench_hotspot.json", "bench_hotspot.json")):
    shutil.copy(f"{O}/{a}", f"profiles/{tag}_{b}")
try:
    shutil.copy("gpurun_out/train_prof/kernel_stats.csv", f"profiles/{tag}_train_step_b64_kernel_stats.csv")
except OSError:
    pass
d = json.load(open(f"profiles/{tag}_bench.json"))
print("traffic MB", round(tot / 1e6, 1), {k: (round(v["fetch_bytes_corrected"] / 1e6, 1), round(v["write_bytes"] / 1e6, 1)) for k, v in out.items()})
print("TAF", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["device_ms_per_encode"], [(a["value"], a["ms_per_step"]) for a in d["also"]])
print("det", d["detector"]["value"], d["detector"]["roofline"]["frac"], d["detector"]["ms_per_batch"], d["detector"]["fwd_plus_decode_nms_ms"])
t = d["train"]; print("train", t["value"], t["ms_per_step"], t["same_step_with_miopen_convs"], t["encode_plus_train_step"]["value"], t["encode_plus_train_step"]["encode_ms_per_batch"])
print("cpu", d["cpu_baseline"]["value"])
for name in ("taf", "gen1", "hot"):
    rows = list(csv.DictReader(open(f"{O}/{name}_kernel_stats.csv")))
    print(name, [(r["Name"].split("::")[-1][:18], round(float(r["AverageNs"]) / 1e3, 1)) for r in rows[:5]])
