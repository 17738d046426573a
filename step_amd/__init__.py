"""step_amd: MI355X-native STEP training step behind the reference's ``step_arch`` module surface.

``from step_amd import STEP, TSFormer`` gives classes a reference config can assign to
``CFG.MODEL.ARCH`` (reference ``step/STEP_PEMS04.py:41``).  All arithmetic runs in
``libstep_hip.so`` (hand-written HIP for gfx950) through the C ABI in ``include/step_hip.h``.
"""


def __getattr__(name):
    if name in ("STEP", "TSFormer", "GraphWaveNet", "DiscreteGraphLearning"):
        from . import step_arch
        return getattr(step_arch, name)
    if name in ("LongHistoryRef", "DeviceWindowLoader"):          # device-resident dataset, index-only loader
        from .step_arch import step
        return getattr(step, name)
    if name == "GraphedTrainStep":          # one captured hipGraph per training step
        from .graphed import GraphedTrainStep
        return GraphedTrainStep
    raise AttributeError(name)
