"""Timing experiments on the LDS-DMA GEMM form (wrong results, same schedule): HOPMI_LIB selects a build of csrc/gemm.hip with
-DHOPMI_AB_EXP=1 (no DMA in the k-loop), 2 (no matrix instructions), 3 (no fragment reads); both tile heights."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import hopmi
from hopmi import ops, _lib
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
M = 4352
L = _lib.lib()
st = torch.cuda.current_stream().cuda_stream


def timed(fn, iters=50):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


tag = os.path.basename(os.environ.get("HOPMI_LIB", "libhopmi.so"))
for N, K in ((2304, 768), (768, 768), (3072, 768), (768, 3072)):
    x = torch.randn(M, K, generator=g).to(dev)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev)
    b = torch.randn(N, generator=g).to(dev)
    img = ops.split_weight_image(w, 16)
    img_a = torch.empty(L.hopmi_rows_image_f16_bytes(M, K), dtype=torch.uint8, device=dev)
    sc2 = torch.empty(2, M, dtype=torch.float32, device=dev)
    L.hopmi_rows_image_f16(x.data_ptr(), M, K, img_a.data_ptr(), sc2.data_ptr(), st)
    out = torch.empty(M, N, device=dev)
    res = {}
    for bm in ("0", "2"):
        os.environ["HOPMI_GEMM_AB_BM64"] = bm
        L.hopmi_reload_env()
        res[bm] = timed(lambda: L.hopmi_gemm_f16x2_ab(img_a.data_ptr(), sc2.data_ptr(), img.data_ptr(), b.data_ptr(), out.data_ptr(), M, N, K, st))
    fl = 2.0 * M * N * K * 3
    print(f"{tag}: M={M} N={N} K={K}: 128-row tiles {res['0']:6.1f} us ({fl / res['0'] / 1e9:5.2f} PF/s of MFMA work) | 64-row tiles {res['2']:6.1f} us ({fl / res['2'] / 1e9:5.2f})", flush=True)
