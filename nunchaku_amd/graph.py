"""HIP-graph capture of a whole denoising step (launch-overhead removal).

A FLUX step is ~900 kernel launches (665 from this library); captured once, it replays as one graph launch.  Every
launch of the library goes to the current torch stream, allocates nothing on the host side of the C ABI and the only
memset (the quantiser's ``lora_act`` zero-fill) is a stream operation, so ``torch.cuda.graph`` can capture it as is.
The launch profiler (``svdq_prof_*``) must be off during capture: event records on a capturing stream become graph
nodes and cannot be timed.
"""

from __future__ import annotations

import torch


class CapturedStep:
    """``fn(*static_inputs)`` captured into a HIP graph; ``__call__`` copies new inputs in and replays.

    ``fn`` must be shape-static and free of host synchronisation; inputs are copied into the captured buffers,
    the returned tensor(s) are the graph's output buffers (clone them if they must survive the next replay)."""

    def __init__(self, fn, example_inputs, warmup: int = 2):
        self.static_inputs = [x.clone() if isinstance(x, torch.Tensor) else x for x in example_inputs]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(warmup):  # lazy initialisation (repacks, workspaces, hipBLASLt handles) outside the capture
                fn(*self.static_inputs)
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        # capture on the stream the warm-up ran on: the GEMM's stream-K workspace is per (device, stream) (_C._workspace),
        # so it already exists and is not allocated (and re-zeroed on every replay) inside the graph
        with torch.cuda.graph(self.graph, stream=side), torch.no_grad():
            self.output = fn(*self.static_inputs)

    def __call__(self, *inputs):
        for dst, src in zip(self.static_inputs, inputs):
            if isinstance(dst, torch.Tensor) and src is not dst:
                dst.copy_(src)
        self.graph.replay()
        return self.output
