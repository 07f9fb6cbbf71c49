"""Per-layer time / TFLOP/s of the detector plan: run under rocprofv3 --kernel-trace, then parse."""
import os, sys, csv, re, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "parse":
    import torch  # noqa
    from frlw_evd_amd.yolox import build_yolox
    from frlw_evd_amd.detector import DetectorEngine
    B = int(os.environ.get("B", "32"))
    e = DetectorEngine(build_yolox(10, 2).eval(), device="cpu"); e.build((10, 256, 320))
    meta = e.ops_meta
    rows = [r for r in csv.DictReader(open(sys.argv[2])) if "k_conv_mfma" in r["Kernel_Name"] or "k_focus" in r["Kernel_Name"] or "k_spp" in r["Kernel_Name"] or "k_upsample" in r["Kernel_Name"] or "k_pred_infer" in r["Kernel_Name"] or "k_focus_stem" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    n = len(meta)
    assert len(rows) % n == 0, (len(rows), n)
    reps = len(rows) // n
    t = [0.0] * n
    for i, r in enumerate(rows[n:]):  # skip the first forward
        t[i % n] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 / (reps - 1)
    agg = collections.OrderedDict()
    for (typ, hw, N, K, fl), us, r in zip(meta, t, rows[:n]):
        key = (typ, B * hw, N, K, re.search(r"<[^>]*>", r["Kernel_Name"]).group(0) if "<" in r["Kernel_Name"] else "")
        a = agg.setdefault(key, [0, 0.0, 0])
        a[0] += 1; a[1] += us; a[2] += fl * B
    tot = sum(a[1] for a in agg.values())
    print(f"{'op':8s} {'M':>8s} {'N':>5s} {'K':>5s} tile          n    us/each   TFLOP/s   share")
    for (typ, M, N, K, tile), (cnt, us, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{typ:8s} {M:8d} {N:5d} {K:5d} {tile:12s} {cnt:3d} {us/cnt:9.1f} {fl/us/1e6 if us else 0:9.1f} {us/tot*100:6.1f}%")
    print(f"total {tot:.0f} us per forward")
else:
    import torch
    from frlw_evd_amd.yolox import build_yolox
    from frlw_evd_amd.yolox.model import recipe_state_dict
    B = int(os.environ.get("B", "32"))
    m = build_yolox(10, 2); m.load_state_dict(recipe_state_dict(m)); m.eval().cuda()
    x = torch.rand(B, 10, 256, 320, device="cuda")
    for _ in range(6): m.engine().raw_outputs(x)
    torch.cuda.synchronize()
