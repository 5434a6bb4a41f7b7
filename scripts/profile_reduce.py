"""Reduce the rocprofv3 output of scripts/profile_search.sh to the small summaries kept under profiles/.

  <tag>_bench_kernel_stats.csv   per-kernel totals/averages from --kernel-trace --stats
  <tag>_scan8_traffic.json       HBM bytes per launch of the int8-copy scan kernel (+ SQ counters)
  <tag>_scan16_traffic.json      HBM bytes per launch of the bf16-copy scan kernel (+ SQ counters)
  <tag>_scan_traffic.json        same for the f32 scan kernel
FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE under-counts 16-B/lane streaming reads by
exactly 2x (MI355X_MICROARCH.md, HBM section), hence fetch_bytes = KB * 1024 * 2.
"""
import csv
import glob
import json
import os
import sys


def find(d, suffix):
    hits = glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True)
    return hits[0] if hits else None


def counters(d, want):
    """{counter: (mean per launch, launches, mean duration us)} for kernels whose name contains `want`."""
    path = find(d, "_counter_collection.csv")
    if not path:
        return {}
    acc = {}
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            if want not in row["Kernel_Name"]:
                continue
            key = row["Counter_Name"]
            disp = row["Dispatch_Id"]
            a = acc.setdefault(key, {})
            dur = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3
            v, _ = a.get(disp, (0.0, dur))
            a[disp] = (v + float(row["Counter_Value"]), dur)
    out = {}
    for key, per in acc.items():
        vals = [v for v, _ in per.values()]
        durs = [d_ for _, d_ in per.values()]
        out[key] = (sum(vals) / len(vals), len(vals), sum(durs) / len(durs))
    return out


def main():
    out_dir, tag = sys.argv[1], sys.argv[2]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    prof = os.path.join(root, "gpurun_out", "profiles_" + tag)
    os.makedirs(prof, exist_ok=True)
    stats = find(os.path.join(out_dir, "stats"), "_kernel_stats.csv")
    if stats:
        with open(stats) as f, open(os.path.join(prof, f"{tag}_bench_kernel_stats.csv"), "w") as g:
            g.write(f.read())
    enc = find(os.path.join(out_dir, "enc_stats"), "_kernel_stats.csv")
    if enc:
        with open(enc) as f, open(os.path.join(prof, f"{tag}_encoder_kernel_stats.csv"), "w") as g:
            g.write(f.read())
    encb = find(os.path.join(out_dir, "enc_bge_stats"), "_kernel_stats.csv")
    if encb:
        with open(encb) as f, open(os.path.join(prof, f"{tag}_encoder_bge_kernel_stats.csv"), "w") as g:
            g.write(f.read())
    et = os.path.join(out_dir, "encoder_bge_traffic.txt")
    if os.path.exists(et):
        with open(et) as f, open(os.path.join(prof, f"{tag}_encoder_bge_traffic.txt"), "w") as g:
            g.write(f.read())
    st768 = find(os.path.join(out_dir, "stats768"), "_kernel_stats.csv")
    if st768:
        with open(st768) as f, open(os.path.join(prof, f"{tag}_bench768_kernel_stats.csv"), "w") as g:
            g.write(f.read())
    for name in glob.glob(os.path.join(out_dir, "power_*.log")):
        with open(name) as f, open(os.path.join(prof, f"{tag}_" + os.path.basename(name)), "w") as g:
            g.write(f.read())
    ec = os.path.join(out_dir, "encoder_clock_mfma.txt")
    if os.path.exists(ec):
        with open(ec) as f, open(os.path.join(prof, f"{tag}_encoder_clock_mfma.txt"), "w") as g:
            g.write(f.read())
    tu = os.path.join(out_dir, "tail_ubench.txt")
    if os.path.exists(tu):
        with open(tu) as f, open(os.path.join(prof, f"{tag}_tail_ubench.txt"), "w") as g:
            g.write(f.read())
    for name in ("bench.json", "bench_under_rocprof.json", "bench_default.json", "bench768_under_rocprof.json"):
        src = os.path.join(out_dir, name)
        if os.path.exists(src):
            lines = [ln for ln in open(src).read().splitlines() if ln.startswith("{")]
            if lines:
                with open(os.path.join(prof, f"{tag}_{name}"), "w") as g:
                    g.write(lines[-1] + "\n")
    bench = {}
    try:
        bench = json.loads(open(os.path.join(prof, f"{tag}_bench.json")).read())
        # (round 6: the side legs' full reports live in a second file, bench.py --sides-out)
        side = os.path.join(out_dir, "bench_sides.json")
        if os.path.exists(side):
            bench.update(json.load(open(side)))
            with open(side) as f, open(os.path.join(prof, f"{tag}_bench_sides.json"), "w") as g:
                g.write(f.read())
    except Exception:
        pass
    el = find(os.path.join(out_dir, "enc_like_stats"), "_kernel_stats.csv")
    if el:
        with open(el) as f, open(os.path.join(prof, f"{tag}_enc_like_kernel_stats.csv"), "w") as g:
            g.write(f.read())
    for label, want, fname, pat in (("i8", "scan8_kernel<3, 1, 1, false>", f"{tag}_scan8_traffic.json", "pmc_*"),
                                    ("768", "scan8_kernel<6, 1, 1, false>", f"{tag}_scan8_768_traffic.json", "pmc768_*"),
                                    ("bf16", "scan16_kernel<3, 1>", f"{tag}_scan16_traffic.json", "pmc_*"),
                                    ("f32", "scan_kernel<3, 1>", f"{tag}_scan_traffic.json", "pmc_*"),
                                    ("768", "scan16_kernel<6, 1>", f"{tag}_scan16_768_traffic.json", "pmc768_*")):
        c = {}
        for d in sorted(glob.glob(os.path.join(out_dir, pat))):
            if os.path.isdir(d):
                c.update(counters(d, want))
        if "FETCH_SIZE" not in c:
            continue
        fetch = c["FETCH_SIZE"][0] * 1024.0 * 2.0
        write = c.get("WRITE_SIZE", (0.0, 0, 0.0))[0] * 1024.0
        if label == "768":
            rf = bench.get("cfg4_shard_10Mx768", {}).get("roofline", {})
            if ("scan8" in want) != ("scan8" in str(rf.get("kernel", ""))):
                rf = {}
        elif label == "f32":
            rf = bench.get("f32_rows", {}).get("roofline", {})          # (its own leg: N*D*4 bytes per launch)
        else:
            rf = bench.get("roofline", {}) if bench.get("scan") == label else bench.get("other_scan", {}).get("roofline", {})
        algo = rf.get("bytes_per_launch")
        res = {
            "kernel": "mx::" + want.replace(", ", ","),
            "launches_sampled": c["FETCH_SIZE"][1],
            "FETCH_SIZE_KB_avg": c["FETCH_SIZE"][0],
            "WRITE_SIZE_KB_avg": c.get("WRITE_SIZE", (0.0, 0, 0.0))[0],
            "fetch_bytes_corrected": fetch,
            "write_bytes": write,
            "traffic_bytes_per_launch": fetch + write,
            "algorithmic_bytes_per_launch": algo,
            "traffic_over_algorithmic": (fetch + write) / algo if algo else None,
            "note": "separate --pmc passes, each with --kernel-trace only; FETCH_SIZE KB x1024 x2 (gfx950 16-B/lane correction)",
        }
        sq = {k: v[0] for k, v in c.items() if k not in ("FETCH_SIZE", "WRITE_SIZE")}
        if "GRBM_GUI_ACTIVE" in c:
            dur_us = c["GRBM_GUI_ACTIVE"][2]
            sq["avg_duration_us_in_that_pass"] = dur_us
            # GRBM_GUI_ACTIVE is summed over the 8 XCDs
            sq["effective_sclk_GHz"] = c["GRBM_GUI_ACTIVE"][0] / 8.0 / dur_us / 1e3
            if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
                # busy cycles summed over 1024 SIMDs... reported per CU group by the SQ: normalise by
                # (256 CUs x 4 SIMDs) x kernel cycles
                cycles = c["GRBM_GUI_ACTIVE"][0] / 8.0
                sq["mfma_busy_frac_per_simd"] = c["SQ_VALU_MFMA_BUSY_CYCLES"][0] / (1024.0 * cycles)
        res["sq_counters"] = sq
        with open(os.path.join(prof, fname), "w") as g:
            json.dump(res, g, indent=1)
        print(fname, json.dumps(res)[:600])
    print("summaries in", prof)


if __name__ == "__main__":
    main()
