import os, sys, gc, time
sys.argv = [sys.argv[0], "12"]
src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "e2e_perf.py")).read().split("# ---- phase breakdown")[0]
exec(compile(src, "e2e_perf.py", "exec"))
tr.start_step = 0
orig = tr.update_step
log = []
def upd(*a, **k):
    r = orig(*a, **k)
    torch.cuda.synchronize()
    log.append((round(torch.cuda.memory_allocated() / 2**20), round(torch.cuda.memory_reserved() / 2**20), torch.cuda.current_stream().cuda_stream))
    return r
tr.update_step = upd
tr.train(max_steps=10)
print(log)
import threading
print("threads", threading.active_count())
