"""The mapping layer's forward S = W E + b (1500 x 30522 x 768): library strided-batched fp32 GEMM vs the pieces of the split-K fp16 form."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import hopmi
from hopmi import ops, _lib
dev = torch.device("cuda:0")
hopmi.use_tuned_gemms()
g = torch.Generator().manual_seed(0)
M, K, N = 1500, 30522, 768
W = (torch.randn(M, K, generator=g) / K ** 0.5).to(dev)
E = torch.randn(K, N, generator=g).to(dev)
b = torch.randn(M, generator=g).to(dev)
L = _lib.lib()
st = torch.cuda.current_stream().cuda_stream


def timed(fn, iters=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def lib_path():
    kc = K // 6
    S = torch.bmm(W.view(M, 6, kc).transpose(0, 1), E.view(6, kc, N)).sum(0)
    S += b.unsqueeze(1)
    return S


img_b = ops.f16_weight_image(E, transpose=True, owners=(E,))
img_a = torch.empty(L.hopmi_rows_image_f16_bytes(M, K), dtype=torch.uint8, device=dev)
sc = torch.empty(2, M, dtype=torch.float32, device=dev)
out = torch.empty(M, N, device=dev)
f_img = lambda: _lib.check(L.hopmi_rows_image_f16(W.data_ptr(), M, K, img_a.data_ptr(), sc.data_ptr(), st), "img")
f_img()
print(f"library path (bmm + sum + bias): {timed(lib_path):7.1f} us")
print(f"image of W (183 MB in, 183 MB out): {timed(f_img):7.1f} us")
for sp in (3, 4, 5, 6, 8, 12):
    os.environ["HOPMI_GEMM_AB_SPLITS"] = str(sp); L.hopmi_reload_env()
    ws = torch.empty(L.hopmi_gemm_f16x2_ab_splitk_ws_floats(M, N, K), dtype=torch.float32, device=dev)
    f = lambda: _lib.check(L.hopmi_gemm_f16x2_ab_splitk(img_a.data_ptr(), sc.data_ptr(), img_b.data_ptr(), b.data_ptr(), out.data_ptr(), M, N, K, ws.data_ptr(), st), "sk")
    print(f"split-K GEMM + sum, {sp} slabs: {timed(f):7.1f} us")
os.environ.pop("HOPMI_GEMM_AB_SPLITS"); L.hopmi_reload_env()
print(f"whole f16 path: {timed(lambda: ops.f16_affine_splitk(W, E, b)):7.1f} us")
