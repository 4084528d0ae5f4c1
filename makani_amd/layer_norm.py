"""Instance norm over a spatially sharded field.

Mirrors ``makani/mpu/layer_norm.py:27-114`` (``DistributedInstanceNorm2d``): local
(var, mean, count) per (b, c), merged across the ``spatial`` group with Welford's
update, then normalise + affine.  Differences, both documented in DESIGN.md:

* on the GPU the norm runs on the HIP instance-norm kernels in two phases (local row sums -> one all-reduce of
  ``[B*C, 2]`` float64 sums -> apply, optionally with the block's GELU fused), the torch-op formulation below is
  the CPU / unsupported-shape path (12 ms vs 1.2 ms per full-resolution norm, fwd+bwd, on one N = 8 shard);
* the three per-rank statistics travel in ONE all-gather (``[B, C, 3]``) instead of three;
* statistics keep their batch dimension (``[B, C, 1, 1]``); the reference reshapes to
  ``(1, -1, 1, 1)`` (layer_norm.py:85-86), which is only valid for local batch 1.
"""
import torch
import torch.distributed as dist
import torch.nn as nn

from . import comm
from .mappings import copy_to_parallel_region, gather_from_parallel_region


class DistributedInstanceNorm2d(nn.Module):
    def __init__(self, num_features, eps=1e-05, affine=False, device=None, dtype=None):
        super().__init__()
        self.eps = eps
        self.affine = affine
        self._counts = {}
        if self.affine:
            self.weight = nn.Parameter(torch.ones(num_features))
            self.bias = nn.Parameter(torch.zeros(num_features))
            self.weight.is_shared_mp = ["spatial"]
            self.bias.is_shared_mp = ["spatial"]

    def _stats_welford(self, x):
        var, mean = torch.var_mean(x, dim=(-2, -1), unbiased=False, keepdim=False)   # [B, C]
        count = torch.full_like(mean, float(x.shape[-2] * x.shape[-1]))
        stats = torch.stack([var, mean, count], dim=-1).unsqueeze(-1)               # [B, C, 3, 1]
        stats = gather_from_parallel_region(stats, -1, None, "spatial")              # [B, C, 3, P]
        vars_, means, counts = stats[:, :, 0], stats[:, :, 1], stats[:, :, 2]
        m2s = vars_ * counts
        mean, m2, count = means[..., 0], m2s[..., 0], counts[..., 0]
        for i in range(1, comm.get_size("spatial")):
            delta = means[..., i] - mean
            m2 = m2 + m2s[..., i] + delta**2 * count * counts[..., i] / (count + counts[..., i])
            if i == 1:
                mean = (mean * count + means[..., i] * counts[..., i]) / (count + counts[..., i])
            else:
                mean = mean + delta * counts[..., i] / (count + counts[..., i])
            count = count + counts[..., i]
        var = m2 / count
        return var.unsqueeze(-1).unsqueeze(-1), mean.unsqueeze(-1).unsqueeze(-1)

    def _global_count(self, x):
        """H*W summed over the spatial group (shards are uneven): one tiny all-reduce per local shape, cached."""
        key = (x.shape[-2], x.shape[-1])
        if key not in self._counts:
            t = torch.tensor([float(key[0] * key[1])], dtype=torch.float64, device=x.device)
            dist.all_reduce(t, group=comm.get_group("spatial"))
            self._counts[key] = int(round(t.item()))
        return self._counts[key]

    def forward(self, x, fuse_gelu=False, row_sums=None):
        from . import ops
        if x.is_cuda and x.dim() == 4 and ops.pointwise_supported(x) and comm.get_size("spatial") > 1:
            # HIP path: local row sums -> one all-reduce of [B*C, 2] doubles -> apply (+ fused GELU); same two
            # streaming passes as the single-GPU norm instead of ~10 elementwise torch passes in fp32
            return ops.instance_norm(x.contiguous(), self.weight if self.affine else None, self.bias if self.affine else None,
                                     self.eps, fuse_gelu, comm.get_group("spatial"), self._global_count(x), row_sums)
        y = self._forward_torch(x)
        return torch.nn.functional.gelu(y) if fuse_gelu else y

    def _forward_torch(self, x):
        with torch.autocast(device_type=x.device.type, enabled=False):
            dtype = x.dtype
            xf = x.float()
            var, mean = self._stats_welford(xf)
            # gradients of the (replicated) statistics are summed over the spatial group
            mean = copy_to_parallel_region(mean, "spatial")
            var = copy_to_parallel_region(var, "spatial")
        x = xf.to(dtype)
        mean = mean.to(dtype)
        var = var.to(dtype)
        x = (x - mean) / torch.sqrt(var + self.eps)
        if self.affine:
            x = self.weight.reshape(-1, 1, 1) * x + self.bias.reshape(-1, 1, 1)
        return x


class DistributedLayerNorm(nn.Module):
    """``makani/mpu/layer_norm.py:117-155``: layer norm over the CHANNEL axis of an NCHW field, per grid point (so it needs no
    communication under spatial sharding; the reference notes that it breaks equivariance).  Same parameters
    (``norm.weight`` / ``norm.bias``, shared over the ``model`` group) and the same arithmetic: transpose channels last,
    ``nn.LayerNorm``, transpose back.  Outside the benchmarked configuration (instance norm): torch ops, no HIP kernel."""

    def __init__(self, normalized_shape, eps=1e-05, elementwise_affine=True, bias=True, device=None, dtype=None):
        super().__init__()
        assert comm.get_size("matmul") == 1
        self.norm = nn.LayerNorm(normalized_shape, eps=eps, elementwise_affine=elementwise_affine, bias=bias, device=device,
                                 dtype=dtype)
        if elementwise_affine:
            self.norm.weight.is_shared_mp = ["model"]
            self.norm.weight.sharded_dims_mp = [None]
            if bias:
                self.norm.bias.is_shared_mp = ["model"]
                self.norm.bias.sharded_dims_mp = [None]

    def forward(self, x):
        return torch.transpose(self.norm(torch.transpose(x, 1, 3)), 1, 3).contiguous()
