#!/usr/bin/env python3
"""Diagnostic: where does a k-step of hopmi_gemm_split go?  -DHOPMI_STAMPS build of gemm.hip, s_memtime stamps of wave 0 of
every workgroup at the middle k-step (shader cycles).  In the stamped build the sched_group_barrier hints are off
(-DHOPMI_GEMM_NO_SCHED), so the MFMA / commit phases are separable."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
PKG = os.path.join(ROOT, "hop-heterogeneous-topology-based-multimodal-entanglement-for-co-speech-gesture-generation_amd")
SO = os.path.join(ROOT, "tools", "probes", "libhopmi_stamps_gemm.so")

def build():
    src = [os.path.join(PKG, "csrc", f) for f in ("api.hip", "gemm.hip")]
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DHOPMI_STAMPS", "-DHOPMI_GEMM_NO_SCHED",
                    "-I" + os.path.join(ROOT, "include"), *src, "-o", SO], check=True)

def main():
    if "--build" in sys.argv:
        build(); return
    L = ctypes.CDLL(SO)
    L.hopmi_gemm_split_image_bytes.restype = ctypes.c_size_t
    dev = torch.device("cuda:0")
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    M = 4352
    for N, K, mode in ((768, 768, 1), (768, 3072, 1), (2304, 768, 2), (3072, 768, 2), (3072, 768, 1)):
        os.environ["HOPMI_GEMM_TILE"] = str(mode); L.hopmi_reload_env()
        x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) / K ** 0.5
        img = torch.empty(L.hopmi_gemm_split_image_bytes(N, K, 3), dtype=torch.uint8, device=dev)
        assert L.hopmi_gemm_split_prepare(P(w), N, K, 3, P(img), None) == 0
        out = torch.empty(M, N, device=dev)
        stamps = torch.zeros(4096 * 8, dtype=torch.int64, device=dev)
        assert L.hopmi_debug_set_stamps_gemm(P(stamps)) == 0
        for _ in range(3):
            stamps.zero_(); assert L.hopmi_gemm_split(P(x), P(img), None, P(out), M, N, K, 3, None) == 0; torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): L.hopmi_gemm_split(P(x), P(img), None, P(out), M, N, K, 3, None)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 50
        st = stamps.view(-1, 8).cpu(); st = st[st[:, 0] > 0]
        names = ["loads issued, fragment reads, MFMAs issued", "(single buffer) barrier 1" if mode != 1 else "-", "commit (split + LDS stores)", "barrier"]
        print(f"M={M} N={N} K={K} mode={mode}: {us:.1f} us, {us * 2.4e3 / (K // 32):.0f} cycles@2.4GHz per k-step and tile round; stamped {st.shape[0]} workgroups")
        seq = [0, 1, 2, 3, 4] if mode != 1 else [0, 1, 3, 4]
        for a, b in zip(seq[:-1], seq[1:]):
            d = (st[:, b] - st[:, a]).double()
            print(f"    stamp {a}->{b}: median {d.median().item():7.0f}  max {d.max().item():7.0f}")
        print(f"    k-step total median {(st[:, 4] - st[:, 0]).double().median().item():.0f}")

if __name__ == "__main__":
    main()
