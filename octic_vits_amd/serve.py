"""Inference as ONE hipGraph replay (static shapes): the serving-side counterpart of ``train.Trainer.capture``.

The reference measures inference throughput with eval / no_grad / autocast forwards in a Python loop
(experiments/complexity.py:40-56) and relies on ``torch.compile`` to cut the launch overhead; here the eval forward of an
``OcticVisionTransformer`` (or any module built from this package's layers) is captured once and replayed: ~700 kernel launches
become one graph launch, the host is out of the loop.  The compute-dtype weight copies the kernels read (LinearD8 preparations,
the standard half's bf16 / transposed copies) are made by the warm-up forwards and are then FROZEN inside the graph - parameters
must not change behind it: ``__call__`` checks the parameters' version counters and addresses and refuses stale graphs."""
import torch


class GraphedForward:
    def __init__(self, model, example, autocast_dtype=torch.bfloat16, warmup=2):
        if not example.is_cuda:
            raise RuntimeError("GraphedForward: a CUDA example input (its shape and dtype become the graph's)")
        self.model = model.eval()
        self.autocast_dtype = autocast_dtype
        self.static_in = example.detach().clone()
        side = torch.cuda.Stream(example.device)
        side.wait_stream(torch.cuda.current_stream(example.device))
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(max(1, warmup)):              # fills the weight caches: no cast / preparation launch is captured
                self._run()
        torch.cuda.current_stream(example.device).wait_stream(side)
        torch.cuda.synchronize(example.device)
        self.graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(self.graph):
            self.static_out = self._run()
        self._stamp = self._param_stamp()

    def _run(self):
        if self.autocast_dtype is None:
            return self.model(self.static_in)
        with torch.autocast("cuda", dtype=self.autocast_dtype):
            return self.model(self.static_in)

    def _param_stamp(self):
        return tuple((p.data_ptr(), p._version) for p in self.model.parameters())

    def __call__(self, x, copy_out=True):
        """x: same shape / dtype as the example.  Returns the model output (a copy unless copy_out=False: the graph's own
        output buffer is overwritten by the next call)."""
        if x.shape != self.static_in.shape or x.dtype != self.static_in.dtype:
            raise ValueError(f"GraphedForward: input {tuple(x.shape)} {x.dtype} differs from the captured "
                             f"{tuple(self.static_in.shape)} {self.static_in.dtype}")
        if self._param_stamp() != self._stamp:
            raise RuntimeError("GraphedForward: the model's parameters changed since the capture (the graph reads frozen "
                               "compute-dtype copies of them) - build a new GraphedForward")
        if self.model.training:
            raise RuntimeError("GraphedForward: the model was switched to training mode")
        self.static_in.copy_(x, non_blocking=True)
        self.graph.replay()
        out = self.static_out
        if not copy_out:
            return out
        return out.clone() if torch.is_tensor(out) else out
