#!/usr/bin/env python3
"""Diagnostic: where does one step of the persistent GRU forward go?  -DHOPMI_STAMPS build, s_memtime stamps (shader
cycles) of thread 0 of every workgroup at step T/2."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
PKG = os.path.join(ROOT, "hop-heterogeneous-topology-based-multimodal-entanglement-for-co-speech-gesture-generation_amd")
SO = os.path.join(ROOT, "tools", "probes", "libhopmi_stamps_gru.so")

def build():
    src = [os.path.join(PKG, "csrc", f) for f in ("api.hip", "gru.hip")]
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DHOPMI_STAMPS",
                    "-I" + os.path.join(ROOT, "include"), *src, "-o", SO], check=True)

def main():
    if "--build" in sys.argv:
        build(); return
    L = ctypes.CDLL(SO)
    L.hopmi_gru_ws_bytes.restype = ctypes.c_size_t
    dev = torch.device("cuda:0")
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    names = ["gi prefetch + load h_{t-1} rows until written", "split + LDS commit + sync", "MFMA (+ operand reads)",
             "partials -> LDS + sync", "gates + stores"]
    for B, T, H in ((128, 34, 350), (256, 34, 350)):
        gi = torch.randn(B, T, 2, 3 * H, device=dev) * 0.3
        whh = torch.randn(2, 3 * H, H, device=dev) / H ** 0.5
        bhh = torch.randn(2, 3 * H, device=dev) * 0.1
        y = torch.empty(B, T, 2 * H, device=dev); gates = torch.empty(B, T, 2, 4 * H, device=dev)
        ws = torch.empty(L.hopmi_gru_ws_bytes(B, T, H) // 4, dtype=torch.int32, device=dev)
        stamps = torch.zeros(1024 * 8, dtype=torch.int64, device=dev)
        assert L.hopmi_debug_set_stamps_gru(P(stamps)) == 0
        args = [P(gi), P(whh), P(bhh), P(y), P(gates), P(ws), B, T, H, None]
        for _ in range(3):
            stamps.zero_(); assert L.hopmi_gru_fwd(*args) == 0; torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): L.hopmi_gru_fwd(*args)
        e1.record(); torch.cuda.synchronize()
        st = stamps.view(-1, 8).cpu(); st = st[st[:, 0] > 0]
        d = (st[:, 1:6] - st[:, 0:5]).double()
        print(f"B={B} T={T} H={H}: {st.shape[0]} workgroups, {e0.elapsed_time(e1) * 50 / T:.2f} us per step; step total median {(st[:,5]-st[:,0]).double().median().item():.0f} cycles")
        for i in range(5):
            print(f"    {names[i]:34s} median {d[:, i].median().item():7.0f}  max {d[:, i].max().item():7.0f}")

if __name__ == "__main__":
    main()
