#!/usr/bin/env python3
"""Diagnostic: where does a wn_layer_fwd workgroup spend its time?  Builds libhopmi_stamps.so (-DHOPMI_STAMPS)
and prints median per-phase s_memtime deltas (shader cycles) of wave 0 of every block."""
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
    L.hopmi_wn_weight_image_bytes.restype = ctypes.c_size_t
    dev = torch.device("cuda:0")
    P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    phases = ["prologue (weights, mix image, biases)", "x loads, normalise, split, LDS, sync", "TCN mfma", "gate, u stores, sync",
              "tail store, node mix, sync", "contraction mfma", "epilogue stores"]
    for V, B, T_in, d, grid, mt in ((9, 128, 16, 1, 256, 5), (9, 128, 6, 2, 256, 5), (9, 128, 16, 1, 512, 3), (42, 64, 16, 1, 256, 3)):
        os.environ["HOPMI_WN_GRID"] = str(grid); os.environ["HOPMI_WN_MAXMT"] = str(mt)
        L.hopmi_reload_env()
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
        img = torch.empty(L.hopmi_wn_weight_image_bytes(1), dtype=torch.uint8, device=dev)
        tab = lambda t: (ctypes.c_void_p * 1)(t.data_ptr())
        assert L.hopmi_wn_prepare_weights(tab(wt), tab(wt2), tab(W), 1, P(img), None) == 0
        args = [P(x), P(scsh), P(img), P(bt), P(bt2), P(prep), P(b), P(y), None, P(ut), 64, P(ws), B, T_in, V, d, 1, None]
        for _ in range(3):
            stamps.zero_(); assert L.hopmi_wn_layer_fwd(*args) == 0; torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): L.hopmi_wn_layer_fwd(*args)
        e1.record(); torch.cuda.synchronize()
        st = stamps.view(-1, 8).cpu()
        st = st[st[:, 0] > 0]
        dd = (st[:, 1:8] - st[:, 0:7]).double()                 # shader cycles
        span = (st[:, 7].max() - st[:, 0].min()).item()
        print(f"V={V} B={B} T_in={T_in} grid={grid} maxmt={mt}: {st.shape[0]} blocks; first-start..last-end {span} cycles; "
              f"per-block median {((st[:,7]-st[:,0]).double()).median().item():.0f} cycles; back-to-back {e0.elapsed_time(e1)*20:.1f} us/launch")
        for i, p in enumerate(phases):
            print(f"    {p:40s} median {dd[:, i].median().item():7.0f} cycles  max {dd[:, i].max().item():7.0f}")
        print(f"    block start skew: {(st[:,0].max()-st[:,0].min()).item()} cycles")

if __name__ == "__main__":
    main()
