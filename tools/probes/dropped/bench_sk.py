"""hopmi_gemm_f16x2_sk (stream-K, 128 x 256 tiles) against hopmi_gemm_f16x2_ab (64-row tiles) at the BERT shapes: time and agreement."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import hopmi
from hopmi import ops, _lib
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
L = _lib.lib()
st = torch.cuda.current_stream().cuda_stream
ws = torch.zeros(L.hopmi_gemm_f16x2_sk_ws_bytes() // 4, dtype=torch.int32, device=dev)


def timed(fn, iters=50):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


shapes = [(4096, 2048, 768), (4096, 4096, 768), (4096, 2048, 3072), (4352, 2304, 768), (4352, 768, 768), (4352, 3072, 768), (4352, 768, 3072), (4352, 768, 2304), (2176, 2304, 768), (2176, 768, 3072),
          (4352, 2100, 992), (1000, 130, 96), (129, 257, 32)]
for M, N, K in shapes:
    x = torch.randn(M, K, generator=g).to(dev)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev)
    b = torch.randn(N, generator=g).to(dev)
    img = ops.split_weight_image(w, 16)
    img_a = torch.empty(L.hopmi_rows_image_f16_bytes(M, K), dtype=torch.uint8, device=dev)
    sc2 = torch.empty(2, M, dtype=torch.float32, device=dev)
    L.hopmi_rows_image_f16(x.data_ptr(), M, K, img_a.data_ptr(), sc2.data_ptr(), st)
    out = torch.empty(M, N, device=dev)
    out2 = torch.full((M, N), float("nan"), device=dev)
    f_ab = lambda: _lib.check(L.hopmi_gemm_f16x2_ab(img_a.data_ptr(), sc2.data_ptr(), img.data_ptr(), b.data_ptr(), out.data_ptr(), M, N, K, st), "ab")
    f_sk = lambda: _lib.check(L.hopmi_gemm_f16x2_sk(img_a.data_ptr(), sc2.data_ptr(), img.data_ptr(), b.data_ptr(), out2.data_ptr(), None, None, None, M, N, K, 0,
                                                    None, None, None, 0.0, 0.0, ws.data_ptr(), st), "sk")
    f_ab(); f_sk(); torch.cuda.synchronize()
    ref = x.double() @ w.double().t() + b.double()
    e_ab = ((out.double() - ref).abs().max() / ref.abs().max()).item()
    e_sk = ((out2.double() - ref).abs().max() / ref.abs().max()).item()
    same = (out == out2).float().mean().item()
    r1 = out2.clone(); f_sk(); torch.cuda.synchronize()
    rep = torch.equal(r1, out2)
    t_ab, t_sk = timed(f_ab), timed(f_sk)
    dbg = {}
    for d in ('1', '2', '3'):
        os.environ['HOPMI_SK_DBG'] = d; L.hopmi_reload_env(); dbg[d] = timed(f_sk)
    os.environ['HOPMI_SK_DBG'] = '0'; L.hopmi_reload_env()
    fl = 2.0 * M * N * K * 3
    print(f"M={M} N={N} K={K}: ab {t_ab:6.1f} us | sk {t_sk:6.1f} us ({fl / t_sk / 1e9:5.2f} PF/s) | err vs f64 ab {e_ab:.2e} sk {e_sk:.2e} | bit-equal elements {same:.3f} | "
          f"reproducible {rep} | dbg no-store {dbg['1']:.1f} 2-steps {dbg['2']:.1f} both {dbg['3']:.1f} | status {int(ws[32])} flags left {int(ws[64:64+512].abs().sum())}", flush=True)
