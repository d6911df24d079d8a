#!/usr/bin/env python3
"""Dev tool: A/B builds of the generated fp16 MLP kernel (gen_mlp_ha.py reads NF_HA_* switches from the environment).
  python tools/ab_mlp_ha.py build name1:SW1=1,SW2=3 name2:...     (here: writes neurofluid_amd/lib/variants/lib_<name>.so)
  python tools/ab_mlp_ha.py run [rows] [iters]                   (GPU box: times every variant with tools/mlp_bench.py ... asm)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "neurofluid_amd", "csrc")
LIBD = os.path.join(ROOT, "neurofluid_amd", "lib")
VARD = os.path.join(LIBD, "variants")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-unused-value"]


def build(specs):
    os.makedirs(VARD, exist_ok=True)
    subprocess.call(["rm", "-rf", VARD])
    os.makedirs(VARD)
    objs = [os.path.join(LIBD, f) for f in os.listdir(LIBD) if f.endswith(".o") and f != "nf_mlp_ha.o"]
    procs = []
    for spec in specs:
        name, _, sw = spec.partition(":")
        env = dict(os.environ)
        for kv in filter(None, sw.split(",")):
            k, _, v = kv.partition("=")
            env["NF_HA_" + k] = v or "1"
        d = os.path.join(VARD, "src_" + name)
        os.makedirs(d, exist_ok=True)
        subprocess.check_call([sys.executable, os.path.join(CSRC, "gen_mlp_ha.py"), os.path.join(d, "nf_mlp_ha_body.inc")], env=env,
                              stderr=subprocess.DEVNULL)
        obj = os.path.join(d, "nf_mlp_ha.o")
        cmd = ["/opt/rocm/bin/hipcc"] + FLAGS + ["-I", d, "-I", CSRC, "-I", os.path.join(ROOT, "include"), "-iquote", d, "-c",
                                                 os.path.join(CSRC, "nf_mlp_ha.hip"), "-o", obj, "-DNF_HA_BODY=\"%s\"" % os.path.join(d, "nf_mlp_ha_body.inc")]
        procs.append((name, obj, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for name, obj, p in procs:
        out, _ = p.communicate()
        if p.returncode:
            sys.stderr.write(out.decode())
            raise SystemExit("variant %s failed" % name)
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(VARD, "lib_%s.so" % name), obj] + objs)
        print("built", name)
    for f in os.listdir(VARD):
        if f.startswith("src_"):
            subprocess.call(["rm", "-rf", os.path.join(VARD, f)])


def run(rows, iters):
    for f in sorted(os.listdir(VARD)):
        if not f.endswith(".so"):
            continue
        env = dict(os.environ, NF_LIB_PATH=os.path.join(VARD, f), NF_HA_TIMING_READ="1" if "timing" in f else "")
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "mlp_bench.py"), str(rows), str(iters), "fp16asmonly"], env=env,
                             capture_output=True, text=True, timeout=600).stdout
        ts = [float(l.split()[-2]) for l in out.splitlines() if l.startswith("ha iter")]
        eq = [l for l in out.splitlines() if l.startswith("ha vs")]
        tm = [l for l in out.splitlines() if l.startswith("timing")]
        print("%-24s %s  best %.1f TFLOP/s   %s  %s" % (f[4:-3], " ".join("%.1f" % t for t in ts), max(ts) if ts else 0, eq[0] if eq else "", tm[0] if tm else ""), flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build(sys.argv[2:])
    else:
        run(int(sys.argv[2]) if len(sys.argv) > 2 else 3280000, int(sys.argv[3]) if len(sys.argv) > 3 else 3)
