"""Per-queue timeline summary of the LAST solve in a rocprofv3 --kernel-trace CSV (solves are separated by >= 30 ms of silence).
usage: python tools/timeline.py <kernel_trace.csv> [nsteps]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"] = int(r["Start_Timestamp"]); r["e"] = int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
# split at silences
segs, cur = [], [rows[0]]
last_end = rows[0]["e"]
for r in rows[1:]:
    if r["s"] - last_end > 30e6:
        segs.append(cur); cur = []
    cur.append(r); last_end = max(last_end, r["e"])
segs.append(cur)
seg = max(segs[-3:], key=len) if len(segs) >= 3 else segs[-1]
seg = segs[-1] if len(segs[-1]) > 100 else seg
t0, t1 = seg[0]["s"], max(r["e"] for r in seg)
print(f"columns: {list(rows[0].keys())[:14]}")
print(f"segments: {[len(s) for s in segs]}; analysed segment: {len(seg)} kernels, span {(t1-t0)/1e6:.2f} ms")
qkey = "Queue_Id" if "Queue_Id" in seg[0] else "Stream_Id"
byq = collections.defaultdict(list)
for r in seg:
    byq[r[qkey]].append(r)
for q, rs in byq.items():
    busy = sum(r["e"] - r["s"] for r in rs)
    gaps = [rs[i + 1]["s"] - rs[i]["e"] for i in range(len(rs) - 1)]
    small = [g for g in gaps if g < 20e3]
    big = [g for g in gaps if g >= 20e3]
    print(f"queue {q}: {len(rs)} kernels, busy {busy/1e6:.2f} ms, span {(rs[-1]['e']-rs[0]['s'])/1e6:.2f} ms, "
          f"gaps<20us: n={len(small)} sum={sum(small)/1e6:.2f} ms median={sorted(small)[len(small)//2]/1e3 if small else 0:.2f} us; gaps>=20us: n={len(big)} sum={sum(big)/1e6:.2f} ms")
    agg = collections.defaultdict(lambda: [0, 0])
    for r in rs:
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("dre::", "")[:40]
        agg[k][0] += 1; agg[k][1] += r["e"] - r["s"]
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
        print(f"     {k:42s} {v[0]:6d} {v[1]/1e6:8.3f} ms  {v[1]/v[0]/1e3:7.2f} us")

# condensed trace of one time step in the middle of the solve: the window between two consecutive k_eff_stack launches (one per step)
marks = [r for r in seg if "k_eff_stack" in r["Kernel_Name"]]
if len(marks) > 12:
    w0, w1 = marks[10]["s"], marks[11]["s"]
    print(f"\none time step (between two k_eff_stack launches): {(w1-w0)/1e3:.1f} us")
    last = {}
    for r in seg:
        if r["s"] < w0 or r["s"] >= w1:
            continue
        q = r[qkey]
        gap = (r["s"] - last[q]) / 1e3 if q in last else 0.0
        last[q] = r["e"]
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("dre::", "")[:34]
        print(f"  q{q} +{(r['s']-w0)/1e3:8.1f} us  dur {(r['e']-r['s'])/1e3:6.1f}  gap {gap:6.1f}  {name}")

# optional: the first N kernels of the analysed solve (both queues), e.g. the set-up step:  TIMELINE_HEAD=200
import os
nh = int(os.environ.get("TIMELINE_HEAD", "0"))
if nh:
    print(f"\nfirst {nh} kernels of the solve:")
    last = {}
    for r in seg[:nh]:
        q = r[qkey]
        gap = (r["s"] - last[q]) / 1e3 if q in last else 0.0
        last[q] = r["e"]
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("dre::", "")[:40]
        print(f"  q{q} +{(r['s']-t0)/1e3:8.1f} us  dur {(r['e']-r['s'])/1e3:6.1f}  gap {gap:6.1f}  {name}")
