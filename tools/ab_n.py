#!/usr/bin/env python3
"""Dev tool: A/B builds of the training MLP kernels (nf_mlp_n.hip: k_mlp_fwd_n / k_mlp_bwd_n; nf_mlp.hip: k_wgrad2) from -D switches.
  python tools/ab_n.py build base: timing:NF_N_TIMING deep:NH_DEPTH=6 ...   (here: neurofluid_amd/lib/variants/lib_<name>.so)
  python tools/ab_n.py run [rows ...]                                        (GPU box: every variant, alternately, twice)
All variants of a call run on one box; the un-suffixed library is not touched."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "neurofluid_amd", "csrc")
LIBD = os.path.join(ROOT, "neurofluid_amd", "lib")
VARD = os.path.join(LIBD, "variants")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-unused-value"]
FILES = ["nf_mlp_n", "nf_mlp"]


def build(specs):
    subprocess.call(["rm", "-rf", VARD])
    os.makedirs(VARD)
    objs = [os.path.join(LIBD, f) for f in os.listdir(LIBD) if f.endswith(".o") and f[:-2] not in FILES]
    procs = []
    for spec in specs:
        name, _, sw = spec.partition(":")
        defs = ["-D" + kv for kv in filter(None, sw.split(","))]
        d = os.path.join(VARD, "src_" + name)
        os.makedirs(d)
        mine = []
        for f in FILES:
            obj = os.path.join(d, f + ".o")
            mine.append(obj)
            cmd = ["/opt/rocm/bin/hipcc"] + FLAGS + defs + ["-I", os.path.join(ROOT, "include"), "-c", os.path.join(CSRC, f + ".hip"), "-o", obj]
            procs.append((name, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        procs.append((name, mine))
    pending = {}
    for name, p in procs:
        if isinstance(p, list):
            subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(VARD, "lib_%s.so" % name)] + p + objs)
            print("built", name)
            continue
        out, _ = p.communicate()
        if p.returncode:
            sys.stderr.write(out.decode())
            raise SystemExit("variant %s failed" % name)
    for f in os.listdir(VARD):
        if f.startswith("src_"):
            subprocess.call(["rm", "-rf", os.path.join(VARD, f)])


def run(args):
    libs = sorted(f for f in os.listdir(VARD) if f.endswith(".so"))
    for rep in range(2):
        for f in libs:
            env = dict(os.environ, NF_LIB_PATH=os.path.join(VARD, f))
            out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "n_bench.py")] + args, env=env, capture_output=True, text=True, timeout=900)
            print("== %s (pass %d)" % (f[4:-3], rep), flush=True)
            print(out.stdout.rstrip() or out.stderr[-2000:], flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build(sys.argv[2:])
    else:
        run(sys.argv[2:])
