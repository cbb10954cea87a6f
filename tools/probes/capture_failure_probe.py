#!/usr/bin/env python3
"""What the HIP runtime / torch do when an exception interrupts a stream capture, per way of failing and per way of cleaning up.

    python tools/probes/capture_failure_probe.py            # runs every (scenario, strategy) pair in a child process each

scenario: pyerr  -- a plain Python exception under a healthy capture
          sync   -- a host read-back (.item()) on the capturing stream: illegal, invalidates the capture
          legacy -- a launch on the NULL stream while a blocking stream captures (hipErrorStreamCaptureImplicit)
          event  -- torch.cuda.synchronize() under capture
strategy: torch_end -- CUDAGraph.capture_end()  (what graph.py did until round 3)
          abandon   -- hopmi_stream_capture_abandon (hipStreamEndCapture + destroy) + allocator pool hand-back, torch object left alone
          leave     -- nothing: the stream stays in capture mode
After the clean-up the child does ordinary eager work (a fresh 256 MB allocation = a real hipMalloc, a GEMM, a sync, a second
capture + replay on a new stream) and prints SURVIVED.  The parent prints one line per pair: return code and last line.
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def child(scenario, strategy):
    import torch
    import hopmi  # noqa: F401
    from hopmi import _lib
    L = _lib.lib()
    import ctypes
    dev = torch.device("cuda:0")
    x = torch.randn(1024, 1024, device=dev)
    (x @ x).sum().item()                  # (the GEMM library initialises itself in its first call: not under capture)
    L.hopmi_noop_launch(None)
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    pool = torch.cuda.graph_pool_handle()
    g = torch.cuda.CUDAGraph()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        g.capture_begin(pool=pool)
        try:
            y = x @ x
            if scenario == "pyerr":
                raise ValueError("boom")
            if scenario == "sync":
                y.sum().item()
            if scenario == "legacy":
                rc = L.hopmi_noop_launch(None)
                print("legacy launch rc", rc, L.hopmi_last_error().decode(), flush=True)
                if rc == 0:
                    torch.zeros(4, device=dev).sum().item()
            if scenario == "event":
                torch.cuda.synchronize()
            raise RuntimeError("scenario did not raise")
        except BaseException as e:  # noqa: BLE001
            print("raised:", type(e).__name__, str(e).splitlines()[0][:160], flush=True)
            st = ctypes.c_int(-1)
            rc = L.hopmi_stream_capture_status(s.cuda_stream, ctypes.byref(st))
            print("capture status rc", rc, "status", st.value, flush=True)
            if strategy == "torch_end":
                try:
                    g.capture_end()
                    print("capture_end returned", flush=True)
                except BaseException as e2:  # noqa: BLE001
                    print("capture_end raised:", type(e2).__name__, str(e2).splitlines()[0][:160], flush=True)
            elif strategy == "abandon":
                rc = L.hopmi_stream_capture_abandon(s.cuda_stream)
                print("abandon rc", rc, L.hopmi_last_error().decode() if rc else "", flush=True)
                torch._C._cuda_endAllocateToPool(dev.index, pool)
                torch._C._cuda_releasePool(dev.index, pool)
                GRAVEYARD.append(g)
            else:
                GRAVEYARD.append(g)
    # ordinary work afterwards
    big = torch.empty(64 * 1024 * 1024, dtype=torch.float32, device=dev)
    big.fill_(1.0)
    z = (x @ x).sum().item()
    torch.cuda.synchronize()
    print("eager work ok", z == z, flush=True)
    s2 = torch.cuda.Stream()
    g2 = torch.cuda.CUDAGraph()
    s2.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s2):
        g2.capture_begin()
        w = x @ x
        g2.capture_end()
    g2.replay()
    torch.cuda.synchronize()
    print("second capture ok", float(w.sum()) == float(w.sum()), flush=True)
    print("SURVIVED", flush=True)


GRAVEYARD = []

if __name__ == "__main__":
    if len(sys.argv) == 3:
        child(sys.argv[1], sys.argv[2])
        sys.stdout.flush()
        os._exit(0) if os.environ.get("PROBE_HARD_EXIT") == "1" else sys.exit(0)
    for scenario in ("pyerr", "sync", "legacy", "event"):
        for strategy in ("torch_end", "abandon", "leave"):
            r = subprocess.run([sys.executable, os.path.abspath(__file__), scenario, strategy], capture_output=True, text=True, timeout=180)
            lines = [ln for ln in (r.stdout + r.stderr).splitlines() if ln.strip() and "amdgpu.ids" not in ln]
            print(f"=== {scenario:7s} {strategy:9s} rc {r.returncode}")
            for ln in lines[-9:]:
                print("    " + ln[:200])
            sys.stdout.flush()
