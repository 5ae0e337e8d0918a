"""vqacl_amd: MI355X-native engine for the VQACL VL-T5 training hot path (hand-written gfx950 kernels behind the
reference's `VLT5VQA.train_step` surface).  See DESIGN.md."""
from .modeling_vlt5 import VLT5, VLT5Config, param_layout  # noqa: F401
from .vqa_model import VLT5VQA  # noqa: F401
from .optim import FusedAdamW, reference_param_groups  # noqa: F401
