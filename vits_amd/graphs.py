"""HIP-graph replay of a fixed-shape call, for small batches where the ~80 launches of a
decoder pass are bounded by host launch cost rather than by the GPU.

``GraphedCall(fn, *example_inputs)`` runs ``fn`` a few times to warm up (weight packing,
one-time kernel attributes), captures one more run into a graph on a side stream, and then
``__call__`` copies new inputs into the captured input buffers and replays.  Every C-ABI entry
point only enqueues work on the stream it is given and never allocates or synchronises, which
is what makes the capture legal; intermediate tensors come from the graph's private pool.
"""
from __future__ import annotations

import torch


class GraphedCall:
    def __init__(self, fn, *example_inputs, warmup: int = 3):
        from . import native
        native.branch_streams(example_inputs[0].device)                 # before the capture's own streams exist
        self._static_in = [x.clone() for x in example_inputs]
        stream = torch.cuda.Stream(device=self._static_in[0].device)
        stream.wait_stream(torch.cuda.current_stream(self._static_in[0].device))
        with torch.cuda.stream(stream), torch.no_grad():
            for _ in range(warmup):
                fn(*self._static_in)
        torch.cuda.current_stream(self._static_in[0].device).wait_stream(stream)
        native.ensure_capture_workspace(self._static_in[0].device)      # scratch of split channel loops, made outside the capture
        self._graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(self._graph):
            self._static_out = fn(*self._static_in)

    def __call__(self, *inputs):
        for dst, src in zip(self._static_in, inputs):
            if dst.shape != src.shape:
                raise ValueError(f"graph captured for shape {tuple(dst.shape)}, got {tuple(src.shape)}")
            dst.copy_(src, non_blocking=True)
        self._graph.replay()
        return self._static_out
