#!/usr/bin/env python3
"""Timing experiments (round 6): libstep_hip_<tag>.so whose kernels OTHER than the encoder raise their wave priority at entry
(s_setprio N), built from patched copies of the sources; extra flags go to the encoder's translation unit.
usage: tools/build_prio_variant.py <tag> <prio or -1> [encoder flags...]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "step_amd", "csrc")
sys.path.insert(0, os.path.join(ROOT, "step_amd"))
from build import SOURCES
tag, prio, enc_flags = sys.argv[1], int(sys.argv[2]), sys.argv[3:]
out = os.path.join(ROOT, "step_amd", "build", "var_" + tag)
os.makedirs(out, exist_ok=True)
objs, procs = [], []
for s in SOURCES:
    src = os.path.join(CSRC, s)
    text = open(src).read()
    flags = []
    if s == "tsformer_encoder.hip":
        flags = enc_flags
    elif prio >= 0 and s.endswith(".hip"):
        # after the opening brace of every __global__ function
        def patch(m):
            return m.group(0) + " __builtin_amdgcn_s_setprio(%d);" % prio
        text, n = re.subn(r"__global__[^;{]*?\)\s*\{", patch, text, flags=re.S)
        print(s, "patched kernels:", n)
    dst = os.path.join(out, s)
    open(dst, "w").write(text)
    o = dst + ".o"
    objs.append(o)
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I", CSRC, "-x", "hip", "-c", dst, "-o", o] + flags
    procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
for s, p in procs:
    o, _ = p.communicate()
    if p.returncode:
        sys.stderr.write(o.decode()); raise SystemExit("failed: " + s)
lib = os.path.join(ROOT, "step_amd", "libstep_hip_%s.so" % tag)
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs + ["-ldl"])
print(lib)
