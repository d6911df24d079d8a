"""Timeline of ONE step from a rocprofv3 kernel trace (dev tool): python tools/step_timeline.py <kernel_trace.csv> <marker substring> [min_us]
Prints every kernel of the last complete step that lasts >= min_us: start offset, duration, queue / stream id, name."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:40], r.get('Queue_Id', r.get('Stream_Id', '?'))) for r in rows)
mark = [i for i, e in enumerate(ev) if sys.argv[2] in e[2]]
mn = float(sys.argv[3]) if len(sys.argv) > 3 else 20.0
s0, s1 = mark[-2], mark[-1]
t0 = ev[s0][0]
print("step span %.0f us" % ((ev[s1][0] - t0) / 1e3))
for e in ev[s0:s1]:
    d = (e[1] - e[0]) / 1e3
    if d >= mn:
        print("%8.0f us  +%7.0f us  q%-3s %s" % ((e[0] - t0) / 1e3, d, e[3], e[2]))
