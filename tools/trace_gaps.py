"""GPU idle gaps of a rocprofv3 kernel trace (dev tool): python tools/trace_gaps.py <kernel_trace.csv> <marker substring> [steps]
The marker is a kernel that runs once per step; the last `steps` steps are analysed."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:44]) for r in rows)
mark = [i for i, e in enumerate(ev) if sys.argv[2] in e[2]]
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
s0, s1 = mark[-steps - 1], mark[-1]
seg = ev[s0:s1]
span = (seg[-1][1] - seg[0][0]) / steps / 1e3
busy = sum(e[1] - e[0] for e in seg) / steps / 1e3
print("per step: span %.0f us, busy %.0f us, idle %.0f us, %d launches" % (span, busy, span - busy, len(seg) // steps))
gaps = {}
for a, b in zip(seg[:-1], seg[1:]):
    g = b[0] - a[1]
    if g > 10000:
        gaps.setdefault((a[2], b[2]), []).append(g / 1e3)
for k, v in sorted(gaps.items(), key=lambda kv: -sum(kv[1]))[:16]:
    print("%7.0f us/step  n=%d  %s -> %s" % (sum(v) / steps, len(v), k[0], k[1]))
