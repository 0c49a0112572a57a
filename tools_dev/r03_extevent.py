"""Does an EXTERNAL event recorded inside a captured HIP graph work on this ROCm / torch?  (the split exchange needs it)"""
import torch, time, sys
dev = "cuda"

def trial(name, use_event, on_side):
    a = torch.zeros(1 << 24, device=dev); b = torch.zeros(1 << 24, device=dev); c = torch.zeros(1 << 24, device=dev)
    ev = torch.cuda.Event(external=True) if use_event else None
    side, comm, s = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.stream(s):
            a.add_(1.0); torch.cuda.synchronize(); a.zero_()
            with torch.cuda.graph(g, stream=s):
                if on_side:
                    side.wait_stream(s)
                    with torch.cuda.stream(side):
                        a.add_(1.0)
                        if ev is not None: ev.record(side)
                        a.add_(0.0)
                else:
                    a.add_(1.0)
                    if ev is not None: ev.record(s)
                for _ in range(50):
                    b.mul_(1.0001).add_(1.0)
                if on_side:
                    s.wait_stream(side)
        torch.cuda.synchronize()
        for it in range(3):
            t0 = time.perf_counter()
            g.replay()
            if ev is not None:
                with torch.cuda.stream(comm):
                    comm.wait_event(ev)
                    c.copy_(a)
                torch.cuda.current_stream().wait_stream(comm)
            torch.cuda.synchronize()
            ms = round((time.perf_counter() - t0) * 1e3, 3)
        print(name, "OK a", float(a[0]), "c", float(c[0]), "ms", ms, flush=True)
    except Exception as e:
        print(name, "FAILED", type(e).__name__, str(e).splitlines()[0], flush=True)

which = sys.argv[1]
trial(which, which != "plain", which in ("plain", "side"))
