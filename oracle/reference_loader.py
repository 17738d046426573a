"""TEST INFRASTRUCTURE (like everything under oracle/): where the reference's own Python sources are, when they are anywhere.

    $STEP_REFERENCE_ROOT            an explicit directory ("none": pretend the build container's checkout is not there)
    /root/reference                 the read-only checkout of the build container
    oracle/_ref/reference.tar.gz    what tools/stage_reference.sh packs for the GPU box: ONE binary artefact, git-ignored (.gitignore lists
                                    oracle/_ref/), shipped by gpurun like the built libraries -- no loose copy of a reference source file
                                    lies anywhere in this tree.  It is unpacked into the system's temporary directory on first use.

Only tests/, bench.py's cpu_baseline leg and the tools that generate goldens may use this; the product (step_amd/, include/) never does
(tests/test_abi_and_host.py::test_product_never_imports_the_oracle_or_the_reference)."""
import hashlib
import os
import tarfile
import tempfile

_HERE = os.path.dirname(os.path.abspath(__file__))
ARCHIVE = os.path.join(_HERE, "_ref", "reference.tar.gz")


def _is_root(c):
    return bool(c) and os.path.isdir(os.path.join(c, "step", "step_arch")) and os.path.isdir(os.path.join(c, "basicts"))


def _unpacked():
    """the archive's content under <tmp>/step_reference_<digest of the archive>, unpacked once per archive"""
    if not os.path.isfile(ARCHIVE):
        return None
    st = os.stat(ARCHIVE)
    tag = hashlib.sha1(f"{st.st_size}:{int(st.st_mtime)}:{ARCHIVE}".encode()).hexdigest()[:16]
    dst = os.path.join(tempfile.gettempdir(), f"step_reference_{tag}")
    if not _is_root(dst):
        part = f"{dst}.{os.getpid()}.part"
        with tarfile.open(ARCHIVE, "r:gz") as tf:
            tf.extractall(part)
        try:
            os.rename(part, dst)              # atomic: concurrent processes (the ranks of a test) race harmlessly
        except OSError:
            import shutil
            shutil.rmtree(part, ignore_errors=True)
    return dst if _is_root(dst) else None


def reference_root():
    """directory holding the reference's `step/` and `basicts/` packages, or None"""
    env = os.environ.get("STEP_REFERENCE_ROOT")
    if env and env != "none" and _is_root(env):
        return env
    if env != "none" and _is_root("/root/reference"):
        return "/root/reference"
    return _unpacked()
