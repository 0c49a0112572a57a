"""HIP-graph capture of a whole training step (forward + loss + backward).

A step on this path is ~45 short kernels, so eager launches are host-bound (~1 ms of Python per step).
Every kernel of libd3m_raster.so goes to torch's current stream and none allocates or synchronises, so a
step is capturable with torch.cuda.CUDAGraph.

One rule makes that robust: the step must ALWAYS run on the same stream -- eager steps, warm-up, capture and
replays.  Autograd binds a leaf's AccumulateGrad node to the stream that was current when the node was
created; if a step is warmed up (or ever run eagerly) on one stream and captured on another, autograd
inserts a cross-stream wait inside the capture, which either aborts the capture or leaves un-joined work
in it -- observed here as GPU memory faults on later replays.  `CapturedStep` owns a dedicated stream and
fences it against the caller's stream on entry and exit.

A second rule, found in round 4 as an intermittent `Fatal Python error: Aborted` in the test suite: NO GARBAGE
COLLECTION DURING CAPTURE.  The step's backward runs on autograd's worker thread; a collection that happens to
trigger there may finalise unrelated cyclic garbage -- another CapturedStep with its HIP graph and memory pool, side
streams, events -- and destroying those inside an open capture aborts the process.  capture() collects first and keeps
the collector off until the capture has ended."""
import gc

import torch


class CapturedStep:
    def __init__(self, fn, between=None):
        """fn(): runs one step and returns a tensor / tuple of tensors that stay valid until the next call.

        A step in SEVERAL PARTS: fn = [part_0, ..., part_n] (the last one returns the outputs) and between = [cb_0, ...,
        cb_{n-1}]: cb_i() runs behind part_i on the step's stream, OUTSIDE every capture -- each part becomes a HIP graph
        of its own (one memory pool for all: what part_i leaves for part_{i+1} lives in it), and cb_i is where a
        collective goes that should travel while the later parts still compute (multiview.py: the texture gradient's
        all-reduce between the texture side and the geometry side of the backward pass)."""
        self.parts = list(fn) if isinstance(fn, (list, tuple)) else [fn]
        self.between = list(between) if between else []
        if len(self.between) != len(self.parts) - 1:
            raise ValueError("a step of n parts takes n - 1 callbacks")
        self.stream = torch.cuda.Stream()
        self.graph = None               # the parts' graphs once captured
        self._static_out = None
        self._last_eager_out = None
        self._scratch_refs = None

    def _run_parts(self, run):
        out = None
        for i in range(len(self.parts)):
            out = run(i)
            if i < len(self.between):
                self.between[i]()
        return out

    def _enter(self):
        cur = torch.cuda.current_stream()
        # (a caller that already runs ON the dedicated stream -- `with torch.cuda.stream(step.stream):` around its loop --
        #  needs no fences: back-to-back replays then follow each other without a cross-stream event in between)
        if cur.cuda_stream != self.stream.cuda_stream:
            self.stream.wait_stream(cur)      # inputs written by the caller's stream are visible
        return cur

    def _exit(self, cur, out, eager):
        if cur.cuda_stream != self.stream.cuda_stream:
            cur.wait_stream(self.stream)      # results are visible to the caller's stream
        # Eager outputs were allocated on the dedicated stream but are consumed on the caller's.  Instead of
        # record_stream() (its deferred-free events are a hazard around graph capture) the last outputs are
        # simply kept alive until the next call, whose entry fence makes their release stream-ordered.
        self._last_eager_out = out if eager else None
        return out

    def __call__(self):
        cur = self._enter()
        eager = self.graph is None
        with torch.cuda.stream(self.stream):
            if eager:
                out = self._run_parts(lambda i: self.parts[i]())
            else:
                self._run_parts(lambda i: self.graph[i].replay())
                out = self._static_out
        return self._exit(cur, out, eager)

    def capture(self, warmup=3):
        """Warm up and capture on the dedicated stream.  Tensors the step reads (parameters, targets) must
        from now on be updated IN PLACE; the returned tensors are static buffers rewritten by every replay."""
        cur = self._enter()
        with torch.cuda.stream(self.stream):
            for _ in range(warmup):
                self._run_parts(lambda i: self.parts[i]())
        torch.cuda.synchronize()
        graphs = [torch.cuda.CUDAGraph() for _ in self.parts]
        gc.collect()                        # whatever is dead dies NOW, not inside the capture (see the module text)
        gc_was_on = gc.isenabled()
        gc.disable()

        def capture_part(i):
            # thread_local: other threads of the process (e.g. the RCCL watchdog) may touch the HIP runtime meanwhile
            pool = {} if i == 0 else {"pool": graphs[0].pool()}
            with torch.cuda.graph(graphs[i], stream=self.stream, capture_error_mode="thread_local", **pool):
                return self.parts[i]()
        try:
            with torch.cuda.stream(self.stream):     # (the callbacks between the parts run on the step's stream, uncaptured)
                self._static_out = self._run_parts(capture_part)
        finally:
            if gc_was_on:
                gc.enable()
        self.graph = graphs
        # the library's cached scratch buffers this capture baked into the graphs live as long as the graphs do, whatever the
        # (bounded) cache does with its entries meanwhile
        from .neural_renderer import rasterize_ops
        self._scratch_refs = rasterize_ops.take_captured_refs()
        cur.wait_stream(self.stream)
        return self

    def release(self):
        """Back to eager execution (still on the dedicated stream)."""
        self.graph = None
        self._static_out = None
        self._scratch_refs = None
