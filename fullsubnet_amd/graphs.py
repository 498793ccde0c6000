"""hipGraph replay of a whole enhancement call for the launch-bound regime (one or a few utterances: ~100 kernels of
4 - 50 us around a few persistent launches, where the host's launch path is a tenth of the call).

Every entry of libfsn_hip.so only enqueues on the caller's stream - auxiliary and side streams are forked and joined with
events, workspaces come from PyTorch's allocator, the flags of the persistent kernels are cleared by kernel nodes - so a
call is capturable as it stands (``tests/test_gpu_streaming.py::test_enhance_is_capturable_in_a_hip_graph``).  ``GraphedCall``
wraps the warm-up / capture / static-buffer bookkeeping: one graph per input shape, replayed on new input.

    enhance = GraphedCall(model.enhance)      # or any ``fn(tensor) -> tensor`` built on the library
    y = enhance(noisy)                        # first call of a shape: eager warm-up + capture; afterwards: copy in, replay

The output tensor is the graph's static buffer: it is overwritten by the next call of the same shape (clone it to keep it).
Inference only.  Not in the reference (its inferencer is eager PyTorch); measured in ``tools/bench_graph_family.py``.

Weights: the captured kernels hold raw pointers into the re-tiled weight caches the eager warm-up made (``Model._packed``,
``SequenceModel._pair_cache``, ``MelScale._w``), which the eager code rebuilds - into NEW buffers - whenever a parameter
changes (``load_state_dict``, an optimizer step between two validations).  A graph is therefore keyed on a fingerprint of
the parameters and buffers of the module(s) behind ``fn`` as well: (data_ptr, version counter) of each.  A change drops the
stale graphs and the next call captures again.  Modules are found on ``fn`` itself (an ``nn.Module`` or a bound method of
one); pass ``modules=[...]`` when ``fn`` is a plain function closing over its models.  The sticky time-out record of the
stream (include/fsn_hip.h, "residency contract") is checked before every replay like the eager entries do at enqueue time.
"""
import torch

from . import _lib


class GraphedCall:
    def __init__(self, fn, max_graphs=8, modules=None):
        self.fn = fn
        self.max_graphs = max_graphs
        if modules is None:
            owner = fn if isinstance(fn, torch.nn.Module) else getattr(fn, "__self__", None)
            modules = [owner] if isinstance(owner, torch.nn.Module) else []
        self.modules = list(modules)
        self._graphs = {}  # (shape, dtype, device) -> (graph, static_in, static_out)
        self._weights = None

    def _fingerprint(self):
        return tuple((t.data_ptr(), t._version) for m in self.modules for t in list(m.parameters()) + list(m.buffers()))

    def invalidate(self):
        """Drop every captured graph (the next call of a shape captures again)."""
        self._graphs.clear()

    def __call__(self, x):
        if not x.is_cuda:
            raise RuntimeError("GraphedCall: the input must live on a ROCm device")
        fp = self._fingerprint()
        if fp != self._weights:  # a parameter was replaced or written: the graphs point at stale weight caches
            self.invalidate()
            self._weights = fp
        _lib.stream_status(x.device, synchronize=False)  # raises FsnTimeout if an earlier persistent launch ran out of time
        key = (tuple(x.shape), x.dtype, str(x.device))
        entry = self._graphs.get(key)
        if entry is None:
            entry = self._capture(x)
            if len(self._graphs) >= self.max_graphs:
                self._graphs.pop(next(iter(self._graphs)))
            self._graphs[key] = entry
        graph, static_in, static_out = entry
        static_in.copy_(x)
        graph.replay()
        return static_out

    @torch.no_grad()
    def _capture(self, x):
        static_in = x.clone()
        cur = torch.cuda.current_stream(x.device)
        side = torch.cuda.Stream(x.device)
        side.wait_stream(cur)
        with torch.cuda.stream(side):  # warm-up off the capture: streams, packed weights and allocator pools exist afterwards
            self.fn(static_in)
            self.fn(static_in)
        cur.wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            static_out = self.fn(static_in)
        return graph, static_in, static_out
