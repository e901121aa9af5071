"""Durations of the multifrontal sweep kernels grouped by grid size (= tree level) from a rocprofv3 --kernel-trace CSV.
usage: python tools/mf_levels.py <kernel_trace.csv>"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
print([k for k in rows[0].keys()])
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    nm = r["Kernel_Name"]
    if "k_mf_" not in nm and "k_top_" not in nm and "k_front" not in nm:
        continue
    short = nm.split("(")[0].replace("void ", "").replace("dre::", "")[:28]
    g = tuple(r.get(k, "") for k in ("Grid_Size_X", "Grid_Size_Y", "Workgroup_Size_X"))
    key = (short, g)
    agg[key][0] += 1
    agg[key][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for (short, g), (cnt, tot) in sorted(agg.items(), key=lambda kv: (kv[0][0], -kv[1][1])):
    print(f"{short:30s} grid={g} calls={cnt:5d} total={tot/1e3:8.3f} ms avg={tot/cnt:7.2f} us")
