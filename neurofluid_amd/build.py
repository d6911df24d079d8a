"""Builds libneurofluid_hip.so for gfx950 with hipcc (in-tree, so the .so travels with the repo)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libneurofluid_hip.so")
SOURCES = ["nf_grid.hip", "nf_render.hip", "nf_mlp.hip", "nf_mlp_l.hip", "nf_mlp_a.hip", "nf_mlp_n.hip", "nf_mlp_h2.hip", "nf_mlp_ha.hip", "nf_mlp_s.hip", "nf_cconv.hip", "nf_cconv_gf.hip", "nf_trans.hip", "nf_host.hip", "nf_metrics.hip", "nf_embed.hip", "nf_gemm.hip"]
# per-file flags.  nf_mlp_h2.hip: MFMA accumulators in VGPRs (the finished blocks are converted by VALU instructions,
# which cannot read AGPRs: AGPR accumulators cost 16 v_accvgpr_read per block and tile)
EXTRA_FLAGS = {"nf_mlp_h2.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"], "nf_mlp_s.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"]}
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-unused-value"] + \
    os.environ.get("NF_EXTRA_DEFS", "").split()      # dev: extra -D switches for A/B builds of a kernel


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    os.makedirs(LIBDIR, exist_ok=True)
    import fcntl
    with open(os.path.join(LIBDIR, ".build.lock"), "w") as lock:      # one builder at a time (ranks starting together)
        fcntl.flock(lock, fcntl.LOCK_EX)
        return _build_locked(force, verbose)


# generated sources: (generator script, output) — the body of nf_mlp_a.hip's asm statement is written by gen_mlp_a.py
# one body per feature-row shape: (qx, qd) = 8-feature groups of the position-like / direction-like features
MLP_A_SHAPES = [(qx, qd) for qx in (8, 9, 16, 17, 24, 25) for qd in (4, 7)]
GENERATED = [("gen_mlp_a.py", "nf_mlp_a_body_%d_%d.inc" % s, [str(s[0]), str(s[1])]) for s in MLP_A_SHAPES] + \
    [("gen_mlp_ha.py", "nf_mlp_ha_body.inc", [])]       # the fp16 kernel's body (nf_mlp_ha.hip)


def _generate():
    procs = []
    for gen, out, args in GENERATED:
        gp, op = os.path.join(CSRC, gen), os.path.join(CSRC, out)
        if _stale(op, [gp]):
            procs.append(subprocess.Popen([sys.executable, gp, op] + args, stderr=subprocess.DEVNULL))
    for p in procs:
        if p.wait() != 0:
            raise RuntimeError("a kernel-body generator (gen_mlp_a.py / gen_mlp_ha.py) failed")


def _build_locked(force, verbose):
    _generate()
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h") or f.endswith(".inc")]
    headers.append(os.path.join(os.path.dirname(HERE), "include", "neurofluid_hip.h"))
    objs = []
    procs = []
    stamps = []
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        if not os.path.exists(sp):
            continue
        obj = os.path.join(LIBDIR, src.replace(".hip", ".o"))
        objs.append(obj)
        cmd = [HIPCC] + FLAGS + EXTRA_FLAGS.get(src, []) + ["-c", sp, "-o", obj]
        # an object is also stale when it was built with other flags (A/B builds with NF_EXTRA_DEFS must not leave their
        # objects behind for the next plain build): the command line is kept next to the object
        stamp = obj + ".cmd"
        same_cmd = os.path.exists(stamp) and open(stamp).read() == " ".join(cmd)
        if force or not same_cmd or _stale(obj, [sp] + headers):
            if os.path.exists(stamp):
                os.remove(stamp)
            stamps.append((stamp, " ".join(cmd)))
            if verbose:
                print(" ".join(cmd))
            procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    failed = False
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            failed = True
            sys.stderr.write(f"[neurofluid_amd.build] {src} failed:\n{out.decode()}\n")
        elif verbose and out:
            print(out.decode())
    if failed:
        raise RuntimeError("hipcc failed")
    # leftovers of removed sources / of `llvm-objdump --offloading` runs in this directory must not ship with the package.
    # Only after every compile job has finished, and never the compiler's own temporaries (`name-XXXX.o.tmp`): another process
    # building at the same moment (several ranks finding a stale library) must not lose its outputs mid-compile.
    keep = set(objs) | {o + ".cmd" for o in objs} | {LIB}
    for f in os.listdir(LIBDIR):
        fp = os.path.join(LIBDIR, f)
        if fp not in keep and not f.endswith(".tmp") and (f.endswith(".o") or f.endswith(".o.cmd") or ".hipv4-" in f or ".host-" in f):
            os.remove(fp)
    for stamp, text in stamps:
        with open(stamp, "w") as f:
            f.write(text)
    if force or procs or _stale(LIB, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
