"""Oracle: shard bookkeeping of the spatially distributed SHT.

TEST INFRASTRUCTURE ONLY -- see ``oracle/__init__.py``.

``compute_split_shapes`` / ``split_tensor_along_dim`` restate the helpers of
nvidia-modulus that the reference imports (``makani/mpu/layers.py:27-31``,
used at ``makani/mpu/layers.py:64-67`` and ``tests/distributed/tests_fft.py``),
following the rule written down in SURVEY.md section 8(c): chunk = ceil(size/P),
shards ``[chunk]*(P-1) + [size - chunk*(P-1)]``; if that last shard would be
<= 0 use chunk = floor(size/P) and give the remainder to the last shard.
A distributed transform's expected output on rank (h, w) is simply the
corresponding slice of the serial oracle's output.
"""
import torch


def compute_split_shapes(size, num_chunks):
    if num_chunks == 1:
        return [size]
    chunk = (size + num_chunks - 1) // num_chunks
    last = max(size - chunk * (num_chunks - 1), 0)
    if last == 0:
        chunk = size // num_chunks
        last = size - chunk * (num_chunks - 1)
    return [chunk] * (num_chunks - 1) + [last]


def split_tensor_along_dim(tensor, dim, num_chunks):
    assert dim < tensor.dim()
    assert tensor.shape[dim] >= num_chunks
    return torch.split(tensor, compute_split_shapes(tensor.shape[dim], num_chunks), dim=dim)


def shard(tensor, dim, num_chunks, rank):
    """The slice of ``tensor`` along ``dim`` that rank ``rank`` of ``num_chunks`` owns."""
    return split_tensor_along_dim(tensor, dim, num_chunks)[rank].contiguous()
