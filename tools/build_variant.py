"""Build an alternative libstep_hip with extra compiler defines, for same-box A/B measurements:

    python tools/build_variant.py lcg -DTSF_DROPOUT_LCG=1        # -> step_amd/libstep_hip_lcg.so
    STEP_HIP_LIB=step_amd/libstep_hip_lcg.so python tools/bench_encoder.py

Objects go to step_amd/build/<name>/; the default library is not touched.  Both files are git-ignored and travel to the GPU box.
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from step_amd import build as B          # noqa: E402


def main():
    name, defines = sys.argv[1], sys.argv[2:]
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    odir = os.path.join(B.HERE, "build", name)
    os.makedirs(odir, exist_ok=True)
    procs, objs = [], []
    for s in B.SOURCES:
        o = os.path.join(odir, s + ".o")
        objs.append(o)
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-x", "hip", *defines, "-c", os.path.join(B.CSRC, s), "-o", o]
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out.decode())
            raise SystemExit("hipcc failed on " + s)
    lib = os.path.join(B.HERE, f"libstep_hip_{name}.so")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs + ["-ldl"])
    print(lib)


if __name__ == "__main__":
    main()
