#!/usr/bin/env python3
"""Diagnostic: where does a wn_layer_fwd workgroup spend its time?  Builds libhopmi_stamps.so (-DHOPMI_STAMPS)
and prints median per-phase s_memtime deltas (100 MHz reference ticks -> ns) of wave 0 of every block."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
PKG = os.path.join(ROOT, "hop-heterogeneous-topology-based-multimodal-entanglement-for-co-speech-gesture-generation_amd")
SO = os.path.join(ROOT, "tools", "probes", "libhopmi_stamps.so")

def build():
    src = [os.path.join(PKG, "csrc", f) for f in ("api.hip", "gcn.hip", "wavenet.hip")]
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DHOPMI_STAMPS",
                    "-I" + os.path.join(ROOT, "include"), *src, "-o", SO], check=True)

def main():
    if "--build" in sys.argv:
        build(); return
    L = ctypes.CDLL(SO)
    L.hopmi_gcn_prep_floats.restype = ctypes.c_size_t
    L.hopmi_wn_layer_ws_floats.restype = ctypes.c_size_t
    dev = torch.device("cuda:0")
    P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    phases = ["issue x + wt loads", "wait, normalise, LDS, sync", "TCN mfma", "gate, fs stores, Wm issue, sync",
              "tail store, node mix, sync", "contraction mfma", "epilogue stores"]
    for V, B, T_in, d, grid, mt in ((9, 128, 16, 1, 256, 5), (9, 128, 16, 1, 512, 3), (9, 128, 10, 1, 256, 5), (42, 64, 16, 1, 256, 3)):
        os.environ["HOPMI_WN_GRID"] = str(grid); os.environ["HOPMI_WN_MAXMT"] = str(mt)
        T_out = T_in - d
        x = torch.randn(B, T_in, V, 64, device=dev); y = torch.empty(B, T_out, V, 64, device=dev)
        fs = torch.empty(B, T_out, V, 128, device=dev); ut = torch.empty(B, 4, V, 64, device=dev)
        A = torch.softmax(torch.randn(V, V, device=dev), 1); A2 = A @ A
        wt = torch.randn(64, 64, 1, 2, device=dev) / 11; wt2 = torch.randn(64, 64, 1, 2, device=dev) / 11; bt = torch.randn(64, device=dev); bt2 = torch.randn(64, device=dev)
        W = torch.randn(64, 192, device=dev) / 14; b = torch.randn(64, device=dev)
        scsh = torch.cat([torch.ones(64), torch.zeros(64)]).to(dev)
        stamps = torch.zeros(4096 * 8, dtype=torch.int64, device=dev)
        assert L.hopmi_debug_set_stamps_wn(P(stamps)) == 0
        prep = torch.empty(L.hopmi_gcn_prep_floats(V), device=dev)
        assert L.hopmi_gcn_prepare(P(A), P(A2), P(prep), V, None) == 0
        ws = torch.empty(L.hopmi_wn_layer_ws_floats(B, T_in, V, d), device=dev)
        args = [P(x), P(scsh), P(wt), P(wt2), P(bt), P(bt2), P(prep), P(W), P(b), P(y), P(fs), P(ut), 64, P(ws), B, T_in, V, d, 1, None]
        for _ in range(3):
            stamps.zero_(); assert L.hopmi_wn_layer_fwd(*args) == 0; torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): L.hopmi_wn_layer_fwd(*args)
        e1.record(); torch.cuda.synchronize()
        st = stamps.view(-1, 8).cpu()
        st = st[st[:, 0] > 0]
        dd = (st[:, 1:8] - st[:, 0:7]).double() * 10.0          # ns
        span = (st[:, 7].max() - st[:, 0].min()).item() * 10
        print(f"V={V} B={B} T_in={T_in} grid={grid} maxmt={mt}: {st.shape[0]} blocks; first-start..last-end {span} ns; "
              f"per-block median {((st[:,7]-st[:,0]).double()*10).median().item():.0f} ns; back-to-back {e0.elapsed_time(e1)*20:.1f} us/launch")
        for i, p in enumerate(phases):
            print(f"    {p:34s} median {dd[:, i].median().item():7.0f} ns  max {dd[:, i].max().item():7.0f}")
        print(f"    block start skew: {(st[:,0].max()-st[:,0].min()).item()*10} ns")

if __name__ == "__main__":
    main()
