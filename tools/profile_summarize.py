"""Turns the rocprofv3 outputs under gpurun_out/ into the committed summaries under profiles/.

  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_X -- python bench.py ...      (timing)
  rocprofv3 --pmc FETCH_SIZE  --output-format csv -d gpurun_out/pmc_f -- python bench.py --steps 1 --warmup 0 ...
  rocprofv3 --pmc WRITE_SIZE  --output-format csv -d gpurun_out/pmc_w -- python bench.py --steps 1 --warmup 0 ...
  (+ the same two PMC passes on tools/pmc_calibrate.py -> gpurun_out/pmc_cal_f, pmc_cal_w)

usage: python tools/profile_summarize.py <round-tag> <stats-dir> [out-dir]   (out-dir defaults to profiles/)
"""
import collections, csv, glob, json, os, sys

tag, stats_dir = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out_dir = sys.argv[3] if len(sys.argv) > 3 else os.path.join(ROOT, "profiles")
os.makedirs(out_dir, exist_ok=True)

# ---- kernel timing summary
f = glob.glob(os.path.join(stats_dir, "*", "*kernel_stats.csv"))[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
with open(os.path.join(out_dir, f"rocprof_{tag}_kernel_stats.md"), "w") as o:
    o.write(f"# rocprofv3 --kernel-trace --stats ({tag})\n\nsource: {os.path.relpath(f, ROOT)}; total kernel time {tot/1e6:.1f} ms\n\n")
    o.write("| kernel | calls | total ms | avg us | min us | max us | % |\n|---|---|---|---|---|---|---|\n")
    for r in rows:
        o.write(f"| `{r['Name'][:90]}` | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.2f} | {float(r['AverageNs'])/1e3:.2f} | "
                f"{float(r['MinNs'])/1e3:.2f} | {float(r['MaxNs'])/1e3:.2f} | {r['Percentage']} |\n")
avg_us = {r["Name"].split("(")[0]: float(r["AverageNs"]) / 1e3 for r in rows}


def agg(d, counter):
    fs = glob.glob(os.path.join(ROOT, "gpurun_out", d, "*", "*counter_collection.csv"))
    if not fs:
        return {}
    out = collections.defaultdict(lambda: [0, 0.0])
    for x in csv.DictReader(open(fs[0])):
        if x["Counter_Name"] == counter:
            k = x["Kernel_Name"].split("(")[0]
            out[k][0] += 1
            out[k][1] += float(x["Counter_Value"])
    return out


cf, cw = agg("pmc_cal_f", "FETCH_SIZE"), agg("pmc_cal_w", "WRITE_SIZE")
KNOWN = 20209 * 1000 * 8.0          # bytes read and bytes written by one k_copy launch of tools/pmc_calibrate.py
fcal = wcal = None
for k, v in cf.items():
    if "k_copy" in k and v[1] / v[0] > 1e4:
        fcal = KNOWN / (v[1] / v[0] * 1024.0)
for k, v in cw.items():
    if "k_copy" in k and v[1] / v[0] > 1e4:
        wcal = KNOWN / (v[1] / v[0] * 1024.0)
bf, bw = agg("pmc_f", "FETCH_SIZE"), agg("pmc_w", "WRITE_SIZE")
traffic = {}
for k, v in bf.items():
    w = bw.get(k, [1, 0.0])
    traffic[k] = dict(launches=v[0], fetch_kb_per_launch=v[1] / v[0], write_kb_per_launch=w[1] / max(w[0], 1),
                      hbm_bytes_per_launch=(v[1] / v[0] * 1024.0 * (fcal or 1.0)) + (w[1] / max(w[0], 1) * 1024.0 * (wcal or 1.0)),
                      avg_us=avg_us.get(k))
json.dump(dict(calibration=dict(fetch_factor=fcal, write_factor=wcal,
                                note="bytes = counter*1024*factor; factors from a 161.7 MB k_copy (8 B/lane loads/stores), tools/pmc_calibrate.py"),
               kernels=traffic), open(os.path.join(out_dir, f"pmc_traffic_{tag}.json"), "w"), indent=1)
print("calibration", fcal, wcal, "kernels", len(traffic))
