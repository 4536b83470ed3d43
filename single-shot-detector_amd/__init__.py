"""MI355X-native RetinaNet inference path (drop-in for the reference's inference/detector.py
+ detector/ssd.py).  Import as `ssd_amd` (see ssd_amd.py at the repository root: the
directory name `single-shot-detector_amd` is not a Python identifier).

Host code is Python; all arithmetic runs in hand-written HIP kernels for gfx950 behind the
C ABI of include/ssd_hip.h (csrc/libssd_hip.so, loaded with ctypes).  PyTorch is used only
for device memory, streams and torch.distributed.  There is NO CPU fallback: every op
raises if the HIP library is missing or no GPU is present.
"""
from .config import load_config, INFERENCE_KEYS                       # noqa: F401
from .variables import (variable_shapes, synthetic_weights, save_weights,   # noqa: F401
                        load_weights)
from .pb_import import read_frozen_graph, load_pb_weights             # noqa: F401
from .ckpt_import import (read_checkpoint, read_checkpoint_index, resolve_checkpoint,   # noqa: F401
                          load_ckpt_weights, crc32c)
from ._lib import build, lib, lib_path, SsdError, set_option, get_option                       # noqa: F401
from .ssd import (SSD, AnchorGenerator, RetinaNetFeatureExtractor, RetinaNetBoxPredictor,     # noqa: F401
                  batch_multiclass_non_max_suppression, network_input_size, Engine)
from .detector import Detector                                         # noqa: F401
from . import coco_eval, coco_metric                                   # noqa: F401
from .distributed import shard_range, all_gather_detections, detect_sharded, bind_to_gpu_numa_node  # noqa: F401
