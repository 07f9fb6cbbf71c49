#!/usr/bin/env python3
"""Copy the summaries tools/refresh_r06.sh left under gpurun_out/refresh_r06/ into profiles/ (prefix r06_) and rebuild
profiles/traffic_*.json from the PMC summaries: FETCH_SIZE (KiB) x 1024 x 2 (gfx950 reports half of a coalesced read,
MI355X_MICROARCH.md; checked against kf_hist, whose read is exactly 8 B per event), WRITE_SIZE (KiB) x 1024.  The traffic
files are tagged with the hash of the kernel sources: bench.py reports `roofline.traffic` only while that hash matches."""
import hashlib, json, os, re, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O = os.path.join(ROOT, "gpurun_out", "refresh_r06")
P = os.path.join(ROOT, "profiles")


def sha():
    h = hashlib.sha256()
    for name in ("taf_fast.hip", "partition.hip", "encoders.hip", "frlw_common.h"):
        h.update(open(os.path.join(ROOT, "frlw-evd_amd", "csrc", name), "rb").read())
    return h.hexdigest()[:16]


def traffic(cfg, tag, alg, prefixes):
    per, cur = {}, None
    for line in open(os.path.join(O, f"{cfg}_pmc_summary.txt")):
        if not line.startswith(" "):
            cur = line.strip(); continue
        m = re.match(r"\s+(\S+)\s+(\d+)", line)
        if m and cur:
            per.setdefault(cur, {})[m.group(1)] = int(m.group(2))
    tot, out = 0, {}
    for k, v in per.items():
        if "FETCH_SIZE" in v and k.startswith(prefixes) and "selftest" not in k:  # (the self-test runs once per process, not per encode)
            f, w = v["FETCH_SIZE"] * 1024 * 2, v.get("WRITE_SIZE", 0) * 1024
            out[k.split("<")[0]] = {"fetch_bytes_corrected": out.get(k.split("<")[0], {}).get("fetch_bytes_corrected", 0) + f,
                                    "write_bytes": out.get(k.split("<")[0], {}).get("write_bytes", 0) + w}
            tot += f + w
    json.dump({"workload": tag, "hbm_bytes_per_encode": tot, "algorithmic_bytes": alg, "ratio": round(tot / alg, 3),
               "kernel_source_sha": sha(),
               "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes over `build/enc_lab <lib> --cfg " + cfg +
                         "` (tools/refresh_r06.sh: the same kernels on a stream of the same shape), per-dispatch averages, KiB x 1024, "
                         "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half of a coalesced read)",
               "per_kernel": out, "source": f"profiles/r06_{cfg}_pmc_summary.txt"},
              open(os.path.join(P, f"traffic_{tag}.json"), "w"), indent=1)
    print(tag, "traffic MB", round(tot / 1e6, 1), "ratio", round(tot / alg, 3),
          {k: (round(v["fetch_bytes_corrected"] / 1e6, 1), round(v["write_bytes"] / 1e6, 1)) for k, v in out.items()})


for f in os.listdir(O):
    if f.endswith(".csv") or f.endswith(".txt") or f.endswith(".json"):
        shutil.copy(os.path.join(O, f), os.path.join(P, "r06_" + f))
taf = lambda n, H, W, K=8: 8 * n + 2 * 4 * 2 * K * H * W + 2 * K * H * W
ev = lambda n, H, W, b=5: 8 * n + 4 * 2 * b * H * W
traffic("mpx", "taf_mpx", taf(10_000_000, 720, 1280), ("kf_",))
traffic("mpx_hot", "taf_mpx_hotspot", taf(10_000_000, 720, 1280), ("kf_",))
traffic("gen1", "taf_gen1", taf(1_000_000, 240, 304), ("kf_",))
traffic("gen1x64", "taf_gen1_x64", 64 * taf(1_000_000, 240, 304), ("kf_",))   # (the lab's sequences are ragged: a few % fewer events)
traffic("evb1", "ev_gen1", ev(1_000_000, 240, 304), ("kf_",))
traffic("evb64", "ev_gen1_x64", 64 * ev(1_000_000, 240, 304), ("kf_",))
traffic("sae", "sae_gen1", 8 * 1_000_000 + 2 * 4 * 2 * 240 * 304 + 4 * 6 * 240 * 304, ("kf_",))
traffic("eci", "eci_gen1", 8 * 100_000 + 4 * 2 * 240 * 304, ("kf_",))
d = json.load(open(os.path.join(P, "r06_bench_detail.json")))
print("TAF", d["value"], d["ms_per_step"], d["roofline"]["frac"], [(a["value"], a["ms_per_step"], a["roofline"]["frac"]) for a in d["also"]])
print("det", d["detector"]["value"], d["detector"]["roofline"]["frac"], "train", d["train"]["ms_per_step"], d["train"].get("same_step_with_miopen_convs"))
print("line bytes", len(open(os.path.join(P, "r06_bench.json")).read()))
