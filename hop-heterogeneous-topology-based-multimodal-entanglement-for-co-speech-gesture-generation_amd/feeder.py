"""Host -> device input stage of the training step.

The reference feeds `train_llm` from a DataLoader over LMDB samples (data_loader/lmdb_data_loader.py): per sample it
pads the raw audio to 36 267 samples (:211), computes the 34 x 128 log-mel feature with librosa on the host (:216-218),
and its collate (:47-62) stacks `text_token_padded` (a numpy float64 row per sample -> DoubleTensor; HOP.py:198 casts it),
`vec_seq` (float32, (34, 3V)), `audio_padded` (float32), `log_melspec` (float32) and `aux_info['vid']`.  LMDB / pyarrow /
the tokenizer stay with the caller; this module is what sits between those host arrays and the step:

  * `log_melspec(audio)`: the log-mel feature on the GPU (libhopmi `hopmi_logmel`, same parameters as the reference),
    so the host ships raw audio only;
  * `HostFeeder`: pinned, double-buffered staging -- batch k+1 is copied into pinned memory on the host while step k
    runs; `next()` enqueues its pinned -> device copy and the log-mel kernels in front of step k+1 and hands out device
    tensors in the collate's dtypes (text as float64, as the reference's collate produces it).
"""
import math

import numpy as np
import torch

from . import _lib

SR, N_FFT, HOP, N_MELS = 16000, 1024, 1096, 128          # lmdb_data_loader.py:216
_RUNS = {}


def _hz_to_mel(f):
    """Slaney scale (librosa.hz_to_mel, htk=False): linear below 1 kHz, logarithmic above."""
    f = np.asarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    min_log_hz, min_log_mel, logstep = 1000.0, 1000.0 / f_sp, math.log(6.4) / 27.0
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-30) / min_log_hz) / logstep, f / f_sp)


def _mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    min_log_hz, min_log_mel, logstep = 1000.0, 1000.0 / f_sp, math.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def mel_filter(sr=SR, n_fft=N_FFT, n_mels=N_MELS):
    """librosa.filters.mel(sr, n_fft, n_mels, fmin=0, fmax=sr/2, htk=False, norm='slaney') -> (n_mels, 1 + n_fft/2)
    float32 (computed in float64, as librosa does)."""
    fftfreqs = np.linspace(0.0, sr / 2.0, 1 + n_fft // 2)
    mel_f = _mel_to_hz(np.linspace(_hz_to_mel(0.0), _hz_to_mel(sr / 2.0), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fftfreqs[None, :]
    lower = -ramps[:-2] / fdiff[:-1, None]
    upper = ramps[2:] / fdiff[1:, None]
    w = np.maximum(0.0, np.minimum(lower, upper))
    w *= (2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels]))[:, None]
    return w.astype(np.float32)


def mel_filter_runs(device):
    """The filter bank as compact per-band runs for hopmi_logmel: (start, len, offset, weights) device tensors."""
    key = str(device)
    if key not in _RUNS:
        w = mel_filter()
        start, length, off, vals = [], [], [], []
        for m in range(w.shape[0]):
            nz = np.nonzero(w[m])[0]
            lo, hi = (int(nz[0]), int(nz[-1]) + 1) if nz.size else (0, 0)
            start.append(lo); length.append(hi - lo); off.append(len(vals))
            vals.extend(w[m, lo:hi].tolist())
        _RUNS[key] = tuple(torch.tensor(a, dtype=dt, device=device) for a, dt in
                           ((start, torch.int32), (length, torch.int32), (off, torch.int32), (vals, torch.float32)))
    return _RUNS[key]


def log_melspec(audio: torch.Tensor, out: torch.Tensor = None, ws: torch.Tensor = None) -> torch.Tensor:
    """(B, n_samples) float32 raw audio on a ROCm device -> (B, 1 + n_samples // 1096, 128) float32 log-mel, the
    `log_melspec` input of the step (lmdb_data_loader.py:216-218).  `out` / `ws`: optional preallocated result and
    workspace of that shape (the feeder reuses its own)."""
    if not audio.is_cuda:
        raise _lib.HopmiError(f"hopmi log_melspec: `audio` is on {audio.device}; the feature is computed on a ROCm device "
                              "(no CPU fallback)")
    audio = audio.contiguous().float()
    B, n = audio.shape
    frames = 1 + n // HOP
    start, length, off, vals = mel_filter_runs(audio.device)
    if ws is None:
        ws = torch.empty(B, frames, N_MELS, dtype=torch.float32, device=audio.device)
    if out is None:
        out = torch.empty(B, frames, N_MELS, dtype=torch.float32, device=audio.device)
    if tuple(out.shape) != (B, frames, N_MELS) or tuple(ws.shape) != (B, frames, N_MELS) or not (out.is_contiguous() and ws.is_contiguous()):
        raise _lib.HopmiError("hopmi log_melspec: `out` / `ws` must be contiguous float32 (B, frames, 128) tensors")
    _lib.check(_lib.lib().hopmi_logmel(audio.data_ptr(), B, n, HOP, start.data_ptr(), length.data_ptr(), off.data_ptr(),
                                       vals.data_ptr(), ws.data_ptr(), out.data_ptr(), torch.cuda.current_stream().cuda_stream),
               "hopmi_logmel")
    return out


class HostFeeder:
    """Pinned, double-buffered host -> device stage.  `source` is an iterator of host batches: dicts with `audio_padded`
    (B, 36267) float32, `text_token_padded` (B, 34) (float64 as the reference's collate makes it, or int64), `vec_seq`
    (B, 34, 3V) float32, `vid_indices` (B,) int64 -- numpy arrays or CPU tensors.  Iterating yields dicts of DEVICE tensors
    `in_audio, log_melspec, text, target_dir_vec, vid_indices`, the five inputs of `train_llm` (train_llm.py:9-11).

        batch = next(feeder);  step(batch...);  feeder.refill()

    `refill()` stages the NEXT host batch into a pinned slot on the host (the only part that costs host time: one memcpy,
    done while the device runs the step just issued).  `next()` enqueues that slot's pinned -> device copies (19 MB at
    B = 128: 0.37 ms at the measured 51 GB/s) and the log-mel kernels on the CURRENT stream, in front of the step that
    consumes them, into one reused set of device buffers -- stream order is the only synchronisation, nothing is allocated
    in the steady state.  Measured at BASELINE.json configs[1]: 23.07 ms per step fed from host memory against 22.81 ms
    with the batch resident in HBM (1.2 %); a copy stream could hide the remaining 0.4 ms and is not worth a second
    stream's ordering rules."""

    KEYS = ("audio_padded", "text_token_padded", "vec_seq", "vid_indices")
    OUT = dict(audio_padded="in_audio", text_token_padded="text", vec_seq="target_dir_vec", vid_indices="vid_indices")

    def __init__(self, source, device, depth=2):
        self.source = iter(source)
        self.device = torch.device(device)
        self.slots = [dict(pinned={}, copied=None) for _ in range(depth)]
        self.dev = {}
        self.depth, self.head, self.tail, self.staged = depth, 0, 0, 0
        self.exhausted = False
        for _ in range(depth):
            self.refill()

    @staticmethod
    def _like(store, key, t, **kw):
        buf = store.get(key)
        if buf is None or buf.shape != t.shape or buf.dtype != t.dtype:
            buf = store[key] = torch.empty(t.shape, dtype=t.dtype, **kw)
        return buf

    def refill(self):
        """Stage the next host batch into a free pinned slot (host memcpy).  Call it right after issuing a step, so that it
        runs while the device executes that step."""
        if self.exhausted or self.staged == self.depth:
            return
        try:
            host = next(self.source)
        except StopIteration:
            self.exhausted = True
            return
        slot = self.slots[self.tail]
        self.tail = (self.tail + 1) % self.depth
        self.staged += 1
        if slot["copied"] is not None:
            slot["copied"].synchronize()            # the slot's previous pinned -> device copy (issued a step ago) is done
        for k in self.KEYS:
            t = torch.as_tensor(host[k])
            if k in ("audio_padded", "vec_seq"):
                t = t.float()                       # collate dtypes (lmdb_data_loader.py:228-231)
            elif k == "vid_indices":
                t = t.long()
            # a plain single-threaded memcpy (numpy): torch's copy_ runs big host copies on its intra-op thread pool, whose
            # workers keep spinning after the copy and starve the ROCm runtime's completion-signal threads on a box with
            # few cores per GPU -- measured as 60-160 ms stalls of the training step every few steps
            np.copyto(self._like(slot["pinned"], k, t, pin_memory=True).numpy(), t.contiguous().numpy())

    def __iter__(self):
        return self

    def __next__(self):
        if self.staged == 0:
            raise StopIteration
        slot = self.slots[self.head]
        self.head = (self.head + 1) % self.depth
        self.staged -= 1
        dev = self.dev
        for k in self.KEYS:
            pin = slot["pinned"][k]
            self._like(dev, k, pin, device=self.device).copy_(pin, non_blocking=True)
        slot["copied"] = torch.cuda.Event()
        slot["copied"].record(torch.cuda.current_stream(self.device))
        B, n = dev["audio_padded"].shape
        shape = torch.empty(B, 1 + n // HOP, N_MELS, dtype=torch.float32, device="meta")
        log_melspec(dev["audio_padded"], out=self._like(dev, "log_melspec", shape, device=self.device),
                    ws=self._like(dev, "mel_ws", shape, device=self.device))
        batch = {self.OUT[k]: dev[k] for k in self.KEYS}
        batch["log_melspec"] = dev["log_melspec"]
        return batch
