"""cProfile of the train_e2e step's host side (dev tool)."""
import os, sys, cProfile, pstats, io
sys.argv = [sys.argv[0], "12"]
src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "e2e_perf.py")).read().split("# ---- phase breakdown")[0]
exec(compile(src, "e2e_perf.py", "exec"))
tr.start_step = 0
pr = cProfile.Profile()
pr.enable()
tr.train(max_steps=12)
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
print(s.getvalue()[:9000])
