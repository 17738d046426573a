"""TEST INFRASTRUCTURE (like everything under oracle/): where the reference's own Python sources are, when they are anywhere.

    /root/reference                 the read-only checkout of the build container
    oracle/_ref/reference           the copy tools/stage_reference.sh makes for the GPU box (git-ignored, shipped by gpurun)
    $STEP_REFERENCE_ROOT            anything else

Only tests/, bench.py's cpu_baseline leg and the tools that generate goldens may use this; the product (step_amd/, include/) never does
(tests/test_abi_and_host.py::test_product_never_imports_the_oracle_or_the_reference)."""
import os

_HERE = os.path.dirname(os.path.abspath(__file__))


def reference_root():
    """directory holding the reference's `step/` and `basicts/` packages, or None"""
    for c in (os.environ.get("STEP_REFERENCE_ROOT"), "/root/reference", os.path.join(_HERE, "_ref", "reference")):
        if c and os.path.isdir(os.path.join(c, "step", "step_arch")) and os.path.isdir(os.path.join(c, "basicts")):
            return c
    return None
