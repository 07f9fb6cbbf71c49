#!/usr/bin/env python3
"""Copy the summaries tools/refresh_r02.sh left under gpurun_out/refresh_r02/ into profiles/ (prefix r02_) and rebuild
profiles/traffic_taf_mpx*.json from the PMC summaries: FETCH_SIZE (KiB) x 1024 x 2 (gfx950 reports half of a coalesced
read, MI355X_MICROARCH.md; checked against kf_hist, whose read is exactly 8 B per event), WRITE_SIZE (KiB) x 1024."""
import hashlib, json, os, re, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O = os.path.join(ROOT, "gpurun_out", "refresh_r02")
P = os.path.join(ROOT, "profiles")


def sha():
    h = hashlib.sha256()
    for name in ("taf_fast.hip", "partition.hip", "encoders.hip", "frlw_common.h"):
        h.update(open(os.path.join(ROOT, "frlw-evd_amd", "csrc", name), "rb").read())
    return h.hexdigest()[:16]


def traffic(work, tag, alg):
    per, cur = {}, None
    for line in open(os.path.join(O, f"taf_{work}_pmc_summary.txt")):
        if not line.startswith(" "):
            cur = line.strip(); continue
        m = re.match(r"\s+(\S+)\s+(\d+)", line)
        if m and cur:
            per.setdefault(cur, {})[m.group(1)] = int(m.group(2))
    tot, out = 0, {}
    for k, v in per.items():
        if "FETCH_SIZE" in v and k.startswith("kf_"):
            f, w = v["FETCH_SIZE"] * 1024 * 2, v.get("WRITE_SIZE", 0) * 1024
            out[k.split("<")[0]] = {"fetch_bytes_corrected": f, "write_bytes": w}
            tot += f + w
    json.dump({"workload": tag, "hbm_bytes_per_encode": tot, "algorithmic_bytes": alg, "ratio": round(tot / alg, 3),
               "kernel_source_sha": sha(),
               "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes (tools/refresh_r02.sh), per-dispatch "
                         "averages, KiB x 1024, FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half of a coalesced read)",
               "per_kernel": out, "source": f"profiles/r02_taf_{work}_pmc_summary.txt"},
              open(os.path.join(P, f"traffic_{tag}.json"), "w"), indent=1)
    print(tag, "traffic MB", round(tot / 1e6, 1), "ratio", round(tot / alg, 3),
          {k: (round(v["fetch_bytes_corrected"] / 1e6, 1), round(v["write_bytes"] / 1e6, 1)) for k, v in out.items()})


for f in os.listdir(O):
    if f.endswith(".csv") or f.endswith(".txt") or f.endswith(".json"):
        shutil.copy(os.path.join(O, f), os.path.join(P, "r02_" + f))
alg = 8 * 10_000_000 + 2 * 4 * 16 * 720 * 1280 + 16 * 720 * 1280
traffic("mpx", "taf_mpx", alg)
traffic("mpx_hot", "taf_mpx_hotspot", alg)
d = json.load(open(os.path.join(P, "r02_bench.json")))
print("TAF", d["value"], d["ms_per_step"], d["roofline"]["frac"], [(a["value"], a["ms_per_step"], a["roofline"]["frac"]) for a in d["also"]])
print("det", d["detector"]["value"], d["detector"]["roofline"]["frac"], "train", d["train"]["ms_per_step"], d["train"].get("same_step_with_miopen_convs"))
