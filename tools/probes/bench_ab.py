"""hopmi_gemm_f16x2_ab (both operands as fp16 hi/lo images, LDS-DMA) vs hopmi_gemm_f16x2 (A split in the k-loop) at the BERT shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import hopmi
from hopmi import ops, _lib
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
M = int(sys.argv[sys.argv.index('--m') + 1]) if '--m' in sys.argv else 4352
L = _lib.lib()
st = torch.cuda.current_stream().cuda_stream


def timed(fn, iters=50):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for N, K in ((2304, 768), (768, 768), (3072, 768), (768, 3072), (768, 2304)):
    x = torch.randn(M, K, generator=g).to(dev)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev)
    b = torch.randn(N, generator=g).to(dev)
    img = ops.split_weight_image(w, 16)
    sc = ops.row_scales(x)
    t_split = timed(lambda: ops._split_gemm(x, img, b, N, K, 16, a_part=sc))
    img_a = torch.empty(L.hopmi_rows_image_f16_bytes(M, K), dtype=torch.uint8, device=dev)
    sc2 = torch.empty(2, M, dtype=torch.float32, device=dev)
    t_img = timed(lambda: L.hopmi_rows_image_f16(x.data_ptr(), M, K, img_a.data_ptr(), sc2.data_ptr(), st))
    out = torch.empty(M, N, device=dev)
    res = {}
    ref = ops._split_gemm(x, img, b, N, K, 16, a_part=sc)
    eq = {}
    for waves in ("8", "4"):                          # (round 6: 4-wave workgroups of 32 x 64 per wave)
        os.environ["HOPMI_GEMM_AB_WAVES"] = waves
        L.hopmi_reload_env()
        out.zero_()
        res[waves] = timed(lambda: L.hopmi_gemm_f16x2_ab(img_a.data_ptr(), sc2.data_ptr(), img.data_ptr(), b.data_ptr(), out.data_ptr(), M, N, K, st))
        eq[waves] = torch.equal(out, ref)
    print(f"M={M} N={N} K={K}: split form {t_split:6.1f} us | ab 8 waves {res['8']:6.1f} 4 waves {res['4']:6.1f} us (image pass {t_img:5.1f}) "
          f"bit-identical to the split form: {eq}  TF-equiv(4w) {2 * M * N * K / res['4'] / 1e6:6.1f}", flush=True)
