"""Long-sequence inference of the reference's demo / evaluation scripts (test_checkpoint.py:395-472):

    for each window a (stride = 34 - n_pre frames):
        pre_seq  = first 16 target frames (a = 0)  |  last 16 generated frames of window a-1      (:448-451)
        outputs  = model(in_audio[a], log_melspec[a], text[a], pre_seq, vid)                       (:459)
        the first 4 frames of window a are cross-faded with the last 4 of window a-1               (:462-470)
    result = windows stacked, each but the last shortened by its 4 blended frames                  (:472)

`generate_long` keeps the whole loop on the device (feedback slice, cross-fade and concatenation are tensor ops,
one device->host copy at the end instead of one per window) and runs the model in eval mode under no_grad, which
puts gwnet on the fused no-autograd kernels with folded BatchNorm.  Audio / mel / token preparation
(librosa, tokenizer) stays with the caller, as in the reference.
"""
import torch

from . import ops as _ops


class _WindowGraph:
    """hipGraph of one batch-1 eval forward (static shapes, no dropout in eval mode): the ~250 launches of a window
    become one graph launch (2.8 -> 1.6 ms per window on an MI355X).  Inputs are copied into the graph's static buffers,
    the output is read from its static output.  Weights are read through their own storages, so in-place updates are
    seen; the cached prototype tensors are not, hence the graph is keyed on the model's prototype key as well."""

    def __init__(self, model, audio, mel, text, pre, vid):
        self.inputs = [t.clone() for t in (audio, mel, text, pre)] + ([vid.clone()] if vid is not None else [None])
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):                      # warm-up outside the capture: library handles, caches, workspaces
                model(*self.inputs)
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        sink, prev = [], _ops.STATUS_SINK
        _ops.STATUS_SINK = sink                     # status words of the recorded persistent GRU launches
        try:
            with torch.cuda.graph(self.graph):
                self.out = model(*self.inputs)[0]
                self.status = torch.stack([w.float().reshape(()) for w in sink]).sum() if sink else None
        finally:
            _ops.STATUS_SINK = prev

    def __call__(self, audio, mel, text, pre, vid):
        for dst, src in zip(self.inputs, (audio, mel, text, pre, vid)):
            if dst is not None:
                dst.copy_(src)
        self.graph.replay()
        return self.out.clone()


def _window_graph(model, audio, mel, text, pre, vid):
    key = (tuple(audio.shape), tuple(mel.shape), tuple(text.shape), tuple(pre.shape), vid is None, str(audio.device),
           model._prototype_key() if hasattr(model, "_prototype_key") else None)
    cache = model.__dict__.setdefault("_hopmi_window_graphs", {})
    g = cache.get(key)
    if g is None:
        cache.clear()                               # one live graph per model: older keys hold stale prototypes
        g = cache[key] = _WindowGraph(model, audio, mel, text, pre, vid)
    return g


@torch.no_grad()
def generate_long(model, in_audio, log_melspec, text_tokens, pre_seq, vid_indices=None, n_blend=4, use_graph=False):
    """in_audio (W, 36267), log_melspec (W, 34, 128), text_tokens (W, 34): one row per window; pre_seq (1, 16, 3V) the
    seed poses of window 0; vid_indices (1,) speaker id.  Returns (W*(34 - n_blend) + n_blend, 3V) direction vectors.
    `use_graph` replays a captured hipGraph of the window forward (same kernels, same results, one launch)."""
    if in_audio.shape[0] != log_melspec.shape[0] or in_audio.shape[0] != text_tokens.shape[0]:
        raise ValueError("hopmi generate_long: one audio / mel / text row per window expected")
    was_training = model.training
    model.eval()
    try:
        W = in_audio.shape[0]
        pre = pre_seq.float()
        chunks = []
        status = []                                 # hand-off status of the recorded window graphs (checked once, at the end)
        for a in range(W):
            if use_graph and in_audio.is_cuda:
                wg = _window_graph(model, in_audio[a:a + 1], log_melspec[a:a + 1], text_tokens[a:a + 1], pre, vid_indices)
                out = wg(in_audio[a:a + 1], log_melspec[a:a + 1], text_tokens[a:a + 1], pre, vid_indices)
                if wg.status is not None:
                    status.append(wg.status.clone())
            else:
                out, *_ = model(in_audio[a:a + 1], log_melspec[a:a + 1], text_tokens[a:a + 1], pre, vid_indices)
            seq = out[0]
            pre = out[:, -16:]                                              # test_checkpoint.py:449
            if chunks:
                last = chunks[-1][-n_blend:]
                chunks[-1] = chunks[-1][:-n_blend]                          # :463-464
                jn = torch.arange(n_blend, device=seq.device, dtype=seq.dtype).unsqueeze(1)
                head = last * (n_blend - jn) / (n_blend + 1) + seq[:n_blend] * (jn + 1) / (n_blend + 1)   # :466-470
                seq = torch.cat([head, seq[n_blend:]], 0)
            chunks.append(seq)
        result = torch.cat(chunks, 0)
        if in_audio.is_cuda:
            # no training step drains the persistent GRU kernels' status words here: read them back once (one tiny
            # device->host copy for the whole sequence) so that a hand-off time-out cannot return garbage poses silently
            _ops.check_status_now()
            if status and float(torch.stack(status).sum().item()) != 0.0:
                raise RuntimeError("hopmi generate_long: a persistent GRU kernel timed out waiting for a hand-off; "
                                   "set HOPMI_GRU_PERSISTENT=0 to use per-time-step launches")
        return result
    finally:
        model.train(was_training)
