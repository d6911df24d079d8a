"""Diagnose periodic host stalls in the e2e step: GC vs device allocator (dev tool)."""
import os, sys, gc, time
sys.argv = [sys.argv[0], "12"]
src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "e2e_perf.py")).read().split("# ---- phase breakdown")[0]
exec(compile(src, "e2e_perf.py", "exec"))
def run(tag, n=12):
    tr.start_step = 0
    s0 = torch.cuda.memory_stats()
    times = []
    orig = tr.update_step
    def upd(*a, **k):
        torch.cuda.synchronize(); t = time.time(); r = orig(*a, **k); torch.cuda.synchronize(); times.append((time.time() - t) * 1e3); return r
    tr.update_step = upd
    tr.train(max_steps=n)
    tr.update_step = orig
    s1 = torch.cuda.memory_stats()
    print(tag, "update_step ms:", [round(t, 1) for t in times])
    print("   device_alloc +%d  device_free +%d  retries +%d  reserved %.2f GB" % (
        s1["num_device_alloc"] - s0["num_device_alloc"], s1["num_device_free"] - s0["num_device_free"],
        s1["num_alloc_retries"] - s0["num_alloc_retries"], torch.cuda.memory_reserved() / 1e9), "gc counts", gc.get_count())
run("default")
gc.disable()
run("gc disabled")
gc.enable()
import collections
snap = torch.cuda.memory_snapshot()
hist = collections.Counter(round(s["total_size"] / 2**20) for s in snap)
print("segments (MiB: count):", sorted(hist.items())[-40:])
# where do new segments come from?  record allocator history for a few steps
torch.cuda.memory._record_memory_history(max_entries=200000)
tr.start_step = 0
tr.train(max_steps=6)
torch.cuda.synchronize()
s = torch.cuda.memory._snapshot()
evs = [e for tr_ in s["device_traces"] for e in tr_ if e["action"] == "segment_alloc"]
print("segment_alloc events:", len(evs))
for e in evs[:12]:
    fr = [f for f in e.get("frames", []) if "/root/repo" in f["filename"] or "neurofluid" in f["filename"]][:3]
    print(round(e["size"] / 2**20, 1), "MiB", [(os.path.basename(f["filename"]), f["line"]) for f in fr])
