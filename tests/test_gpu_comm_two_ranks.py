"""csrc/comm.cpp + step_amd/comm.py with TWO ranks on the one GPU of a gpurun box.  RCCL itself refuses two ranks on one device, so
$STEP_RCCL_LIB points the library at tests/fake_rccl -- a stand-in (test infrastructure, built here with hipcc) that implements the seven
RCCL entry points through shared memory and host staging.  It says nothing about RCCL's transport or speed; it runs everything this
repository adds around it with nranks = 2: the id exchange, init, the self-check's closed forms, event ordering of the overlapped gradient
all-reduce against the compute stream, ncclAvg semantics, the time-sliced graph learner's sums in stream order, and that every rank ends a
training step with the same parameters (tests/comm_two_ranks_worker.py)."""
import os
import subprocess
import sys

import pytest

from tests.test_abi_and_host import _free_port

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAKE_DIR = os.path.join(ROOT, "tests", "fake_rccl")


def build_fake():
    src, lib = os.path.join(FAKE_DIR, "fake_rccl.cpp"), os.path.join(FAKE_DIR, "libfake_rccl.so")
    if not os.path.exists(lib) or os.path.getmtime(lib) < os.path.getmtime(src):
        subprocess.run(["/opt/rocm/bin/hipcc", "-O2", "-std=c++17", "-fPIC", "-shared", src, "-o", lib, "-lrt"], check=True, timeout=300)
    return lib


def test_native_collectives_with_two_ranks_on_one_device():
    lib = build_fake()
    script = os.path.join(ROOT, "tests", "comm_two_ranks_worker.py")
    world = 2
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), WORLD_SIZE=str(world), HSA_ENABLE_IPC_MODE_LEGACY="0",
               STEP_RCCL_LIB=lib)
    procs = [subprocess.Popen([sys.executable, script, ROOT], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
             for r in range(world)]
    outs = [p.communicate(timeout=900)[0].decode() for p in procs]
    print("\n".join(o[-2500:] for o in outs))
    for p, o in zip(procs, outs):
        assert p.returncode == 0 and "ok" in o, o[-4000:]
