"""Overlapped micro-batches: the all-to-alls of one micro-batch run under the kernels of the other.

With latitude sharding every spectral layer has four blocking all-to-all transposes (distributed.py); on one HIP stream
the GPU idles while they run.  ``MicroBatchRunner`` splits the local batch into ``n`` micro-batches, gives each its own
HIP stream and its own *lane* -- a separate copy of the process-group tree, i.e. a separate RCCL communicator with its
own collective queue (comm.add_lane) -- and issues them one after the other from the single Python thread.  Every rank
issues the same collectives in the same order on each communicator, the streams only order the work of their own
micro-batch, so the transposes of micro-batch 0 overlap the FFT / Legendre / GEMM kernels of micro-batch 1 and vice
versa, forward and backward (autograd replays every node on the stream, and here the lane, of its forward).

CAUTION (round 3): kernels of two streams that run at the same moment are NOT independent on the MI355X boxes this was built
on -- a dense bf16-MFMA workgroup (dhconv, weight gradients) beside an LDS-exchange workgroup (the FFT rows; rocFFT's just the
same) leaves single 16-lane register beats of the latter stale, a handful of wrong rows per hundred launches
(tools/ab/share_stress.py, profiles/r03_share_stress.txt).  The runner is therefore opt-in
(``MK_BENCH_MICROBATCH=2`` in bench.py); the default step keeps one stream.

The reference has no counterpart (its trainer runs one stream, makani/utils/trainer.py:742-763); results are identical
to the single-stream step up to the fp32 order of the gradient accumulation over the micro-batches.
"""
import torch

from . import comm


class MicroBatchRunner:
    def __init__(self, n):
        self.n = int(n)
        self.streams = [torch.cuda.Stream() for _ in range(self.n)]
        # parameters are accumulated on the stream they were created on (the calling stream) while their gradients
        # arrive from the micro-batch streams: intended here, the engine orders the two with events
        quiet = getattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch", None)
        if quiet is not None:
            quiet(False)
        if comm.get_world_size() > 1:
            while comm.num_lanes() < self.n:
                comm.add_lane()

    def forward(self, fn):
        """``fn(j)`` runs forward + loss of micro-batch j and returns its scalar loss; returns the summed loss
        (on the calling stream).  Call ``.backward()`` on the result as usual."""
        main = torch.cuda.current_stream()
        losses = []
        for j, s in enumerate(self.streams):
            s.wait_stream(main)                      # inputs / parameters written on the calling stream
            with comm.lane(j if comm.num_lanes() > j else 0), torch.cuda.stream(s):
                losses.append(fn(j))
        total = None
        for s, l in zip(self.streams, losses):
            main.wait_stream(s)
            l.record_stream(main)
            total = l if total is None else total + l
        return total

    def sync(self):
        """Make the calling stream wait for everything the micro-batch streams have been given (after backward)."""
        main = torch.cuda.current_stream()
        for s in self.streams:
            main.wait_stream(s)
