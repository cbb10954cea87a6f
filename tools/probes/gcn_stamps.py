#!/usr/bin/env python3
"""Diagnostic: where does a gcn_fwd workgroup spend its cycles?  Builds libhopmi_stamps.so
(-DHOPMI_STAMPS) next to the product library, runs one launch per shape and prints the median
per-phase shader-cycle counts of wave 0 of every block (shares, not absolute run time)."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
PKG = os.path.join(ROOT, "hop-heterogeneous-topology-based-multimodal-entanglement-for-co-speech-gesture-generation_amd")
SO = os.path.join(ROOT, "tools", "probes", "libhopmi_stamps.so")

def build():
    src = [os.path.join(PKG, "csrc", f) for f in ("api.hip", "gcn.hip")]
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DHOPMI_STAMPS",
                    "-I" + os.path.join(ROOT, "include"), *src, "-o", SO], check=True)

def main():
    if "--build" in sys.argv:
        build(); return
    L = ctypes.CDLL(SO)
    dev = torch.device("cuda:0")
    phases = ["issue loads", "load wait+LDS commit+sync", "node mix+sync", "GEMM", "stores issued"]
    for V, B, T, S in ((9, 128, 15, 0), (9, 128, 15, 7), (9, 128, 4, 0), (42, 64, 15, 0)):
        os.environ["HOPMI_GCN_FWD_S"] = str(S)
        n_slabs = B * T
        x = torch.randn(n_slabs, V, 64, device=dev); h = torch.empty_like(x)
        A = torch.softmax(torch.randn(V, V, device=dev), 1); A2 = A @ A
        W = torch.randn(64, 192, device=dev) / 14; b = torch.randn(64, device=dev)
        stamps = torch.zeros(4096 * 8, dtype=torch.int64, device=dev)
        assert L.hopmi_debug_set_stamps(ctypes.c_void_p(stamps.data_ptr())) == 0
        L.hopmi_gcn_prep_floats.restype = ctypes.c_size_t
        prep = torch.empty(L.hopmi_gcn_prep_floats(V), device=dev)
        assert L.hopmi_gcn_prepare(ctypes.c_void_p(A.data_ptr()), ctypes.c_void_p(A2.data_ptr()), ctypes.c_void_p(prep.data_ptr()), V, None) == 0
        args = [ctypes.c_void_p(t.data_ptr()) for t in (x, prep, W, b, h)] + [n_slabs, V, None]
        for _ in range(3):
            stamps.zero_(); assert L.hopmi_gcn_fwd(*args) == 0; torch.cuda.synchronize()
        st = stamps.view(-1, 8).cpu()
        st = st[st[:, 0] > 0]
        d = (st[:, 1:6] - st[:, 0:5]).double()
        span = (st[:, 5].max() - st[:, 0].min()).item()
        print(f"V={V} B={B} T={T} S={S}: {st.shape[0]} blocks; kernel span {span} cycles; per-block total median {(st[:,5]-st[:,0]).double().median().item():.0f}")
        for i, p in enumerate(phases):
            print(f"    {p:28s} median {d[:, i].median().item():8.0f}  max {d[:, i].max().item():8.0f}")
        print(f"    block start skew: {(st[:,0].max()-st[:,0].min()).item()} cycles")

if __name__ == "__main__":
    main()
