"""Turns the rocprofv3 outputs under gpurun_out/ into the committed summaries under profiles/ (tools/collect_profiles.sh drives it).

  rocprof_<tag>_kernel_stats.md : rocprofv3 --kernel-trace --stats of `python bench.py ...` (per-kernel calls, total, average)
  pmc_traffic_<tag>.json        : per kernel and launch: FETCH_SIZE x 2 (gfx950 counts 128-B requests as 64 B) and WRITE_SIZE x 1 in bytes.
                                  These are L2-FABRIC bytes: requests the XCD L2s send towards memory, Infinity-Cache (MALL) hits included
                                  (MI355X_MICROARCH.md, HBM / Infinity Cache sections).  They are an upper bound of the HBM traffic and equal it
                                  only when the working set exceeds the 256 MiB Infinity Cache; the implied rate is printed next to the 8 TB/s HBM
                                  peak only with that caveat (field "fabric_GBps").
  mfma_util_<tag>.json          : per kernel: matrix-core busy fraction = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x SIMDs) and f64 MFMA flops
                                  (SQ_INSTS_VALU_MFMA_MOPS_F64 x 512) per launch, rate against the 78.6 TFLOP/s f64 matrix peak.

usage: python tools/profile_summarize.py <tag> <stats-dir> [out-dir]   (out-dir defaults to profiles/)
"""
import collections, csv, glob, json, os, sys

tag, stats_dir = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out_dir = sys.argv[3] if len(sys.argv) > 3 else os.path.join(ROOT, "profiles")
os.makedirs(out_dir, exist_ok=True)
SIMDS = 1024          # 256 CUs x 4

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
    out = collections.defaultdict(lambda: [0, 0.0])
    if not fs:
        return out
    for x in csv.DictReader(open(fs[0])):
        if x["Counter_Name"] == counter:
            k = x["Kernel_Name"].split("(")[0]
            out[k][0] += 1
            out[k][1] += float(x["Counter_Value"])
    return out


bf, bw = agg("pmc_f", "FETCH_SIZE"), agg("pmc_w", "WRITE_SIZE")
kern = {}
for k, v in bf.items():
    w = bw.get(k, [1, 0.0])
    fb = v[1] / v[0] * 1024.0 * 2.0                       # gfx950: FETCH_SIZE reports half of a coalesced streaming read
    wb = w[1] / max(w[0], 1) * 1024.0
    au = avg_us.get(k)
    kern[k] = dict(launches=v[0], fetch_kb_per_launch_raw=v[1] / v[0], write_kb_per_launch_raw=w[1] / max(w[0], 1),
                   fabric_bytes_per_launch=fb + wb, avg_us=au,
                   fabric_GBps=(fb + wb) / (au * 1e-6) / 1e9 if au else None)
run = None
try:          # the counter passes ran `bench.py --steps 1 --warmup 1` (+ the profiled extra solve): 3 solves of adi_iterations_per_solve each
    bj = json.loads(open(os.path.join(out_dir, f"bench_under_rocprof_{tag}.json")).read().strip().splitlines()[-1])
    run = dict(solves=3, adi_iterations_per_solve=bj["config"]["adi_iterations_per_solve"],
               note="counter passes: warm-up + 1 timed + 1 profiled solve; one shifted solve per ADI iteration")
except Exception:
    pass
json.dump(dict(run=run, note="bytes = FETCH_SIZE[KB] x 1024 x 2 + WRITE_SIZE[KB] x 1024 (MI355X_MICROARCH.md, HBM section).  L2-fabric traffic, Infinity-Cache hits "
                    "included: an upper bound of the HBM traffic, equal to it only for working sets beyond 256 MiB.  fabric_GBps may therefore exceed what "
                    "HBM delivers and is NOT an HBM bandwidth.",
               kernels=kern), open(os.path.join(out_dir, f"pmc_traffic_{tag}.json"), "w"), indent=1)

mb, ga, mo = agg("pmc_m", "SQ_VALU_MFMA_BUSY_CYCLES"), agg("pmc_m", "GRBM_GUI_ACTIVE"), agg("pmc_o", "SQ_INSTS_VALU_MFMA_MOPS_F64")
mf = {}
for k, v in mb.items():
    g = ga.get(k, [1, 0.0])
    o_ = mo.get(k, [1, 0.0])
    au = avg_us.get(k)
    fl = o_[1] / max(o_[0], 1) * 512.0
    mf[k] = dict(launches=v[0], mfma_busy_cycles_per_launch=v[1] / v[0], gui_active_per_launch=g[1] / max(g[0], 1),
                 mfma_util=(v[1] / v[0]) / max(g[1] / max(g[0], 1) * SIMDS, 1.0),
                 mfma_util_by_duration=((v[1] / v[0]) / (au * 1e-6 * 2.4e9 * SIMDS)) if au else None,
                 f64_mfma_flops_per_launch=fl, avg_us=au, f64_mfma_TFLOPs=fl / (au * 1e-6) / 1e12 if au else None,
                 frac_of_f64_matrix_peak=(fl / (au * 1e-6) / 1e12 / 78.6) if au else None)
json.dump(dict(note="mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x 1024 SIMDs) per launch (GRBM_GUI_ACTIVE is summed over the 8 XCDs by rocprofv3, so the "
                    "quotient reads LOW by up to 8x on short dispatches; mfma_util_by_duration = busy cycles / (kernel duration x 2.4 GHz x 1024 SIMDs) and the flop rate are the robust figures); f64 flops = SQ_INSTS_VALU_MFMA_MOPS_F64 x 512; "
                    "peak 78.6 TFLOP/s (f64 matrix).",
               kernels=mf), open(os.path.join(out_dir, f"mfma_util_{tag}.json"), "w"), indent=1)
print("wrote", os.listdir(out_dir))
