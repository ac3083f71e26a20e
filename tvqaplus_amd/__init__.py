
from . import torch_ops as _torch_ops  # noqa: E402

_torch_ops.register()      # torch.ops.stage_hip.* (tvqaplus_amd/torch_ops.py)
