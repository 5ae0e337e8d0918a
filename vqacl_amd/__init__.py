"""vqacl_amd: MI355X-native engine for the VQACL VL-T5 training hot path (hand-written gfx950 kernels behind the
reference's `VLT5VQA.train_step` surface).  See DESIGN.md."""
from .modeling_vlt5 import VLT5, VLT5Config, param_layout  # noqa: F401
from .vqa_model import VLT5VQA  # noqa: F401
from .optim import FusedAdamW, reference_param_groups  # noqa: F401
# the callers either side of the path (SURVEY 8 f-2 .. f-4): host-side mirrors of the reference's modules
from .feed import FeatureStore, H5FeatureSource, StoreRef, collate, normalize_boxes  # noqa: F401
from .hdf5_io import H5File, write_feature_file  # noqa: F401   (the reference's HDF5 feature files without h5py: ctypes over libhdf5, loaded on first use)
from .loop import ContinualTrainer, ExemplarMemory, predict, test_seen_tasks  # noqa: F401   (loop.evaluate stays in its module: `evaluate` is the evaluator module)
from .evaluate import VQAEvaluator, evaluate_metric, normalize_answer  # noqa: F401
from .checkpoint import load_checkpoint, load_prototypes, save_checkpoint, save_prototypes  # noqa: F401
