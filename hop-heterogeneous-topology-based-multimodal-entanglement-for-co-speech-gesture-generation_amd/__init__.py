"""hopmi -- MI355X-native hot path of the HOP co-speech gesture generator.

Host side (Python on PyTorch-ROCm) of the drop-in for the reference's
`model/HOP.py::Model`, `model/gwnet.py::gwnet`, `train_eval/train_llm.py::train_llm`
and `train_eval/train_gan.py::train_iter_gan`; the arithmetic of the graph-wavenet block,
the reprogramming attention and the fused BERT element-wise ops runs in hand-written
gfx950 kernels behind the C ABI of `include/hopmi.h` (`libhopmi.so`, loaded with ctypes).

The directory name is the one the build contract prescribes; `import hopmi` (the loader
module at the repository root) is the importable alias.
"""
__version__ = "0.1.0"

from .model import Model, ReprogrammingLayer          # noqa: F401
from .gwnet import gwnet, gcn, nconv, linear          # noqa: F401
from .nets import ConvDiscriminator, PoseGenerator    # noqa: F401
from .steps import train_llm, train_iter_gan, mixed_precision   # noqa: F401
from .infer import generate_long                      # noqa: F401
from .graph import GraphedTrainStep                   # noqa: F401
from .feeder import HostFeeder, log_melspec           # noqa: F401
from .tuning import use_tuned_gemms                   # noqa: F401
from .ops import gemm_parts, strict_fp32              # noqa: F401
