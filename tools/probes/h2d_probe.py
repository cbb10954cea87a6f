"""Probe: host -> device copy bandwidth from pinned / pageable memory on this box."""
import time, torch
dev = torch.device("cuda:0")
for mb in (1, 19, 128):
    n = mb * (1 << 20) // 4
    src_p = torch.empty(n, dtype=torch.float32).pin_memory(); src_p.normal_()
    src_q = torch.empty(n, dtype=torch.float32, pin_memory=True); src_q.normal_()
    src = torch.randn(n)
    dst = torch.empty(n, device=dev)
    for name, s in (("pin_memory()", src_p), ("empty(pin_memory=True)", src_q), ("pageable", src)):
        for _ in range(2): dst.copy_(s, non_blocking=True); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter(); e0.record()
        for _ in range(5): dst.copy_(s, non_blocking=True)
        e1.record(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        print(f"{mb:4d} MB {name:24s}: is_pinned {s.is_pinned()}  host issue {1e3*(t1-t0)/5:.2f} ms/copy  device {e0.elapsed_time(e1)/5:.2f} ms/copy  = {mb/1024/(e0.elapsed_time(e1)/5e3):.1f} GB/s", flush=True)
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(5): dst.copy_(src_p, non_blocking=True)
        e1.record(st)
    torch.cuda.synchronize()
    print(f"{mb:4d} MB pinned on a side stream: {e0.elapsed_time(e1)/5:.2f} ms/copy")
