"""TEST INFRASTRUCTURE ONLY -- closed-form deterministic parameter / input fills.

Golden fixtures hold outputs only.  Parameters and inputs are regenerated from their
*names* with a CPU `torch.Generator` seeded by crc32(name), so the reference (in
`tools/make_golden.py`), the oracle and the HIP product path all see bit-identical
values no matter in which order their modules were constructed.  torch's CPU Philox
stream for a given seed is identical in the build container and on the GPU box
(same image, same wheel).
"""
import math
import zlib

import torch

# per-name-substring gain overrides: keep every stage of the path numerically "awake"
# (e.g. with plain 1/sqrt(fan_in) fills the 1500-way softmax would be almost uniform).
DEFAULT_GAINS = {
    "nodevec": 1.0,                 # gwnet.py:82-83 initialises these with randn
    "mapping_layer.weight": 12.0,
    "key_projection.weight": 4.0,
    "query_projection.weight": 4.0,
    "word_embeddings": 10.0,
}


def _gen(name: str, salt: int = 0) -> torch.Generator:
    g = torch.Generator(device="cpu")
    g.manual_seed((zlib.crc32(name.encode()) + 7919 * salt) & 0x7FFFFFFF)
    return g


def uniform(name: str, shape, scale: float = 1.0, salt: int = 0) -> torch.Tensor:
    """U(-scale, scale) float32 tensor that depends only on (name, shape, salt)."""
    u = torch.rand(tuple(shape), generator=_gen(name, salt), dtype=torch.float32)
    return u.mul_(2.0).sub_(1.0).mul_(scale)


def integers(name: str, shape, high: int, salt: int = 0) -> torch.Tensor:
    return torch.randint(0, high, tuple(shape), generator=_gen(name, salt), dtype=torch.int64)


def normal(name: str, shape, scale: float = 1.0, salt: int = 0) -> torch.Tensor:
    return torch.randn(tuple(shape), generator=_gen(name, salt), dtype=torch.float32).mul_(scale)


def _scale_for(name: str, shape, gains) -> float:
    gain = 1.0
    for key, g in gains.items():
        if key in name:
            gain = g
    if "nodevec" in name:
        return gain
    if len(shape) >= 2:
        fan_in = 1
        for d in shape[1:]:
            fan_in *= d
        return gain / math.sqrt(max(fan_in, 1))
    return 0.05 * gain


def fill_value(name: str, shape, salt: int = 0, gains=None) -> torch.Tensor:
    """The closed-form float32 value of state entry `name` with `shape`."""
    gains = DEFAULT_GAINS if gains is None else gains
    shape = tuple(shape)
    leaf = name.rsplit(".", 1)[-1]
    if leaf == "running_var":
        return uniform(name, shape, 1.0, salt).abs_().mul_(0.2).add_(1.0)
    if leaf == "running_mean":
        return uniform(name, shape, 0.05, salt)
    if len(shape) == 1 and leaf == "weight":          # BatchNorm / LayerNorm scales
        return uniform(name, shape, 0.1, salt).add_(1.0)
    return uniform(name, shape, _scale_for(name, shape, gains), salt)


@torch.no_grad()
def fill_state_(module: torch.nn.Module, salt: int = 0, gains=None) -> None:
    """Overwrite every float entry of `module.state_dict()` in sorted-key order."""
    sd = module.state_dict()
    for name in sorted(sd.keys()):
        t = sd[name]
        if not t.is_floating_point():
            continue                                   # num_batches_tracked, position_ids
        t.copy_(fill_value(name, t.shape, salt, gains).to(t.dtype))


def hot_path_inputs(B: int, V: int, vocab: int, n_spk: int, salt: int = 0):
    """Synthetic batch on the step's input contract; closed form."""
    return dict(
        in_audio=normal("in_audio", (B, 36267), 1.0, salt),
        log_melspec=normal("log_melspec", (B, 34, 128), 1.0, salt),
        text=integers("text", (B, 34), vocab, salt),
        target_dir_vec=normal("target_dir_vec", (B, 34, 3 * V), 0.1, salt),
        vid_indices=integers("vid_indices", (B,), n_spk, salt),
    )
