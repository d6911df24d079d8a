import os, sys, gc, time
sys.argv = [sys.argv[0], "70"]
src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "e2e_perf.py")).read().split("# ---- phase breakdown")[0]
exec(compile(src, "e2e_perf.py", "exec"))
tr.start_step = 0
orig = tr.update_step
times = []
def upd(*a, **k):
    torch.cuda.synchronize(); t = time.time(); r = orig(*a, **k); torch.cuda.synchronize(); times.append((time.time() - t) * 1e3); return r
tr.update_step = upd
tr.train(max_steps=70)
print("spikes at", [(i, round(t)) for i, t in enumerate(times) if t > 20], "median", sorted(times)[len(times)//2])
