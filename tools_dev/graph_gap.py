"""Per-node overhead of a replayed HIP graph on this box: N dependent tiny kernels in one captured stream."""
import time
import torch
x = torch.zeros(64, device="cuda")
for n in (1, 10, 50, 100, 200):
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):
            x.add_(1.0)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            for _ in range(n):
                x.add_(1.0)
    torch.cuda.synchronize()
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 50
    for _ in range(reps):
        g.replay()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"nodes {n:4d}: {dt * 1e6:8.1f} us per replay, {dt * 1e6 / n:6.2f} us per node")
