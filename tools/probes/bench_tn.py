"""hopmi_gemm_f16x2_tn: the double-buffered form (round 6, HOPMI_GEMM_TN_DB=1) against the single-buffer form at the generator's
weight-gradient shapes: time per call (back to back) and bitwise equality of the results (incl. the column sums and a split shape)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import hopmi
from hopmi import ops, _lib
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
L = _lib.lib()


def timed(fn, iters=40):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for M, N, K in ((4352, 2100, 992), (4352, 2100, 700), (4352, 768, 1536), (2048, 1700, 3400), (2048, 170, 1700), (4352, 1050, 350), (4352, 1024, 128), (1000, 130, 70)):
    dy = (torch.randn(M, N, generator=g) * torch.logspace(-3, 1, M).unsqueeze(1)).to(dev)
    x = torch.randn(M, K, generator=g).to(dev)
    pad = lambda t: torch.nn.functional.pad(t, (0, (-t.shape[1]) % 4)).contiguous()
    rs_a, rs_b = ops.row_scales(pad(dy)), ops.row_scales(pad(x))
    res, t = {}, {}
    for db in ("0", "1", "2"):
        os.environ["HOPMI_GEMM_TN_DB"] = db
        L.hopmi_reload_env()
        t[db] = timed(lambda: ops.f16_mm_tn(dy, x, rs_a, rs_b))
        res[db] = ops.f16_mm_tn(dy, x, rs_a, rs_b, colsum=True)
    torch.cuda.synchronize()
    same = torch.equal(res["0"][0], res["1"][0]) and torch.equal(res["0"][1], res["1"][1]) and torch.equal(res["0"][0], res["2"][0])
    cs_ref = dy.double().sum(0)
    cs_err = ((res["2"][1].double() - cs_ref).norm() / cs_ref.norm()).item()
    ref = dy.double().t() @ x.double()
    err = ((res["1"][0].double() - ref).norm() / ref.norm()).item()
    print(f"M={M} N={N} K={K}: single buffer {t['0']:6.1f} us | double buffer {t['1']:6.1f} | + 16-byte loads {t['2']:6.1f} us ({2e-6 * M * N * K / t['2']:6.1f} TF-equiv) "
          f"products bit-identical {same}  rel err vs float64 {err:.2e}  column sums (16-byte form) {cs_err:.1e}", flush=True)
