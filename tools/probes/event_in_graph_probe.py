import torch
x = torch.randn(1 << 24, device="cuda")
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
for ext in (False, True):
    try:
        kw = dict(enable_timing=True)
        if ext:
            kw["external"] = True
        e0, e1 = torch.cuda.Event(**kw), torch.cuda.Event(**kw)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(s):
            y = x * 2
            torch.cuda.synchronize()
            with torch.cuda.graph(g, stream=s):
                e0.record(s)
                y = x * 2
                y = y + 1
                e1.record(s)
            g.replay(); g.replay()
        torch.cuda.synchronize()
        print("external" if ext else "plain", "elapsed", e0.elapsed_time(e1))
    except Exception as ex:
        print("external" if ext else "plain", "FAILED", type(ex).__name__, str(ex)[:200])
