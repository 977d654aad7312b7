"""A whole optimisation step -- render(), loss, backward -- as ONE replayable HIP graph.

The step of the reference's optimisation tests and examples (src/mesh_renderer/mesh_renderer_test.py:238-262,
src/examples/example4.py:54-75) is ~10 kernel launches here; on small images the launches and the Python above them,
not the kernels, set the step time (configs[1]: 0.45 ms eager, 0.14 ms replayed).  `capture_step` records the step once
with torch.cuda.CUDAGraph (hipGraph on ROCm) and returns an object whose replay() re-runs it on the CURRENT values of
the tensors it read -- update parameters in place (optimizer.step() does) and replay.

The C ABI underneath neither allocates nor synchronises, so everything render() launches is capturable.  What cannot
be captured is host-side work: cameras must be DEVICE tensors (host cameras are turned into matrices on the host and
uploaded: a capture would bake that one upload's values in), and Python-side decisions (which kernel variant, which
tensors require grad) are those of the capture.

Known limitations (round 5's two HIP-graph anomalies):
 * the library issues NO memset nodes.  Round 6 reproduced the anomaly without any of this library
   (tools/graph_memset_repro.py, profiles/r06_graph_memset_repro.txt; ROCm 7.2, torch 2.10): a captured hipMemsetAsync
   of 64 bytes or more writes the bytes of a POINTER (0x7f66....) instead of the fill value on the second replay once
   eager work has run in between -- for buffers of the graph's pool and for buffers allocated before the capture
   alike, with no eager allocation at the node's address: the node's parameters, not pool memory handed out again.
   8-byte memsets are unaffected.  Every zero-fill of the library is therefore a kernel node (arguments by value).
   Capture nothing of your own that ends in hipMemsetAsync (torch.zeros / fill_ are kernels: fine).
 * capture ONE stream: a step that forks to other streams inside the capture (tools/overlap_probe.py's four lockstep
   chunks) crashed inside hipGraph capture; render()/loss/backward as written here stay on the current stream.
"""
import torch


class CapturedStep:
    """replay() re-runs the captured step; `outputs` are the tensors step() returned (static storage, rewritten by
    every replay), `parameters` the tensors whose .grad the replay rewrites."""

    def __init__(self, graph, outputs, parameters):
        self.graph, self.outputs, self.parameters = graph, outputs, parameters

    def replay(self):
        self.graph.replay()
        return self.outputs

    __call__ = replay


def capture_step(step, parameters, warmup=3):
    """step(): runs forward, loss and backward (e.g. `loss = torch.mean(torch.abs(render(...) - target));
    loss.backward(); return loss`) and returns the tensor(s) to keep; parameters: the leaf tensors it differentiates
    to (their .grad becomes static memory owned by the graph: read it after replay(), do not set it to None).

    Runs `warmup` eager steps on a side stream (allocator and library warm-up, cached maps), then captures one step.
    Host-side cameras are not detected here: the capture itself fails, loudly, at their upload."""
    if not torch.cuda.is_available():
        raise RuntimeError("capture_step needs the GPU: there is no CPU path")
    parameters = list(parameters)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(max(1, int(warmup))):
            for p in parameters:
                p.grad = None
            step()
    torch.cuda.current_stream().wait_stream(side)
    for p in parameters:
        p.grad = None
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        outputs = step()
    return CapturedStep(graph, outputs, parameters)
