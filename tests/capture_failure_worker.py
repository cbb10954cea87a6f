"""Child process of tests/test_gpu_graph.py::test_recording_failure_is_not_fatal: an exception while hopmi.GraphedTrainStep
records a step must come back as a Python exception, leave the process able to train (eager steps on the same object, with the
same results as train_llm on a twin), and never take the process down.

    python tests/capture_failure_worker.py pyerr|sync|devsync

pyerr  : a ValueError raised by the loss operator while the step is being recorded (the capture itself is healthy)
sync   : a host read-back under capture (illegal: the call raises)
devsync: torch.cuda.synchronize() under capture (illegal: the runtime INVALIDATES the capture and the call raises)
Prints RESULT {json} as its last line."""
import copy
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main(mode):
    import torch
    import hopmi
    from hopmi import ops, steps
    from oracle.golden_util import Accel, step_args
    from test_gpu_graph import _pair
    dev = torch.device("cuda:0")
    steps._randn_like = lambda t: torch.full_like(t, 0.5)
    steps._randperm = lambda n, device: torch.arange(n - 1, -1, -1, device=device)
    m1, d1, inp = _pair(9, dev)
    m2, d2 = copy.deepcopy(m1), copy.deepcopy(d1)
    m2._randn_like = m1._randn_like
    mk = lambda m, d: (torch.optim.Adam([p for p in m.parameters() if p.requires_grad], lr=1e-3, betas=(0.5, 0.999)),
                       torch.optim.Adam(d.parameters(), lr=1e-4, betas=(0.5, 0.999)))
    g1, o1 = mk(m1, d1)
    g2, o2 = mk(m2, d2)
    args = step_args(9)
    batch = (inp["in_audio"], inp["log_melspec"], inp["text"], inp["target_dir_vec"], inp["vid_indices"])
    graphed = hopmi.GraphedTrainStep(args, m2, d2, g2, o2, eager_calls=1)
    real = ops.hop_losses

    def failing(out, *a, **k):
        if steps._CAPTURE is not None:
            if mode == "pyerr":
                raise ValueError("injected failure while recording")
            if mode == "devsync":
                torch.cuda.synchronize()
            out.sum().item()                                 # a host read-back under capture
        return real(out, *a, **k)

    res = {"mode": mode, "raised": None, "losses_match": [], "enabled": None}
    for it in range(4):
        want = hopmi.train_llm(args, 0, *batch, m1, d1, g1, o1, Accel())
        if it == 1:
            ops.hop_losses = failing
            try:
                graphed(0, *batch)
                res["raised"] = "nothing"
            except BaseException as e:  # noqa: BLE001
                res["raised"] = type(e).__name__
            ops.hop_losses = real
            res["enabled"], res["broken"] = graphed.enabled, graphed.broken
            # the failed call did not train: bring the twin back in step by running the step the eager way
            got = graphed(0, *batch)
        else:
            got = graphed(0, *batch)
        res["losses_match"].append(sorted(got) == sorted(want) and all(abs(got[k] - want[k]) <= 2e-4 * max(abs(want[k]), 1e-6) for k in want))
    res["n_eager"], res["n_replay"] = graphed.n_eager, graphed.n_replay
    big = torch.empty(32 * 1024 * 1024, device=dev).fill_(1.0)        # a fresh allocation + work + sync still function
    res["after"] = float(big.sum().item())
    print("RESULT " + json.dumps(res), flush=True)


if __name__ == "__main__":
    main(sys.argv[1])
