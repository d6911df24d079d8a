import os, sys, gc, time
sys.argv = [sys.argv[0], "40"]
src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "e2e_perf.py")).read().split("# ---- phase breakdown")[0]
exec(compile(src, "e2e_perf.py", "exec"))
from neurofluid_amd import autograd_bwd, ops, transmodel, _lib
last = [time.perf_counter(), "start"]
def mk(orig):
    def check(rc, what=""):
        now = time.perf_counter()
        if now - last[0] > 0.02:
            print(f"STALL {1e3*(now-last[0]):.1f} ms between '{last[1]}' and '{what}'")
        last[0], last[1] = now, what
        return orig(rc, what)
    return check
for m in (autograd_bwd, ops, transmodel):
    m.check = mk(m.check)
tr.start_step = 0
orig_upd = tr.update_step
def upd(loss, gs):
    t = time.perf_counter(); last[0], last[1] = t, "update_step begin"
    r = orig_upd(loss, gs)
    now = time.perf_counter()
    if now - last[0] > 0.02: print(f"STALL {1e3*(now-last[0]):.1f} ms between '{last[1]}' and 'update_step end'")
    last[0], last[1] = now, "update_step end"
    return r
tr.update_step = upd
tr.train(max_steps=40)
