from torch.nn.init import trunc_normal_          # step/step_arch/tsformer/tsformer.py:3  # noqa: F401
