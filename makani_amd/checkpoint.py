"""Flexible (model-parallel-layout independent) checkpoints: ``makani/utils/trainer.py:971-1098``.

Saving gathers every parameter along the axes its ``sharded_dims_mp`` annotation names (uneven shards allowed) and rank 0
writes ONE file with full-size tensors; restoring splits each tensor for the layout of the loading job (shard size
``ceil(n / group size)``, the reference's rule -- which is also how ``compute_split_shapes`` deals rows).  A checkpoint
written under ``h = 4`` therefore loads under ``h = 1`` or ``h = 8`` and vice versa, and checkpoints of the reference
load here unchanged: parameter names and public shapes are the reference's (``state_dict`` compatibility is part of the
drop-in boundary).
"""
from collections import OrderedDict

import torch
import torch.distributed as dist

from . import comm


def gather_uneven(tensor, dim, comm_name):
    """All-gather ``tensor`` along ``dim`` over the group ``comm_name`` (shards may differ in size): mpu/helpers.py:33-56."""
    size = comm.get_size(comm_name)
    if size == 1:
        return tensor
    group = comm.get_group(comm_name)
    n = torch.tensor([tensor.shape[dim]], dtype=torch.int64, device=tensor.device)
    sizes = [torch.empty_like(n) for _ in range(size)]
    dist.all_gather(sizes, n, group=group)
    sizes = [int(s.item()) for s in sizes]
    big = max(sizes)
    t = tensor.contiguous()
    if t.shape[dim] < big:          # equal-size all_gather works on every backend: pad, trim after
        shp = list(t.shape)
        shp[dim] = big - t.shape[dim]
        t = torch.cat([t, t.new_zeros(shp)], dim=dim)
    parts = [torch.empty_like(t) for _ in range(size)]
    dist.all_gather(parts, t, group=group)
    return torch.cat([q.narrow(dim, 0, s) for q, s in zip(parts, sizes)], dim=dim)


def collect_flexible_state(model):
    """``OrderedDict`` name -> full-size CPU tensor of every parameter (collective over the model-parallel groups)."""
    state = OrderedDict()
    for k, v in model.named_parameters():
        weight = v.detach().clone()
        for d, group in enumerate(getattr(v, "sharded_dims_mp", [])):
            if group is not None:
                weight = gather_uneven(weight, d, group)
        state[k] = weight.to("cpu")
    return state


def save_flexible_checkpoint(checkpoint_path, model, iters=0, epoch=0, optimizer=None, scheduler=None, params=None):
    """Write ``checkpoint_path.format(mp_rank=0)`` from world rank 0; every rank of the model-parallel group must call."""
    state = collect_flexible_state(model)
    if params is not None and not isinstance(params, (dict, list, str, int, float)):
        params = params.to_dict() if hasattr(params, "to_dict") else dict(vars(params))   # plain data: loadable with weights_only=True
    store = {"iters": iters, "epoch": epoch, "model_state": state, "params": params}
    if optimizer is not None:
        store["optimizer_state_dict"] = optimizer.state_dict()
    if scheduler is not None:
        store["scheduler_state_dict"] = scheduler.state_dict()
    if comm.get_world_rank() == 0:
        torch.save(store, checkpoint_path.format(mp_rank=0))
    if dist.is_initialized() and comm.get_size("model") > 1:
        dist.barrier(group=comm.get_group("model"))


def restore_flexible_checkpoint(checkpoint_path, model, scheduler=None, load_optimizer=False, load_scheduler=False, logger=None,
                                trusted_pickle=False):
    """Load ``checkpoint_path.format(mp_rank=0)`` into ``model`` under the CURRENT model-parallel layout.  Returns the
    ``(iters, epoch)`` stored in the file.  Parameters missing from the file are left alone (and reported).

    The file is read with ``weights_only=True`` (tensors, numbers, strings, dicts, lists: everything this function uses);
    a file whose ``params`` entry holds pickled classes is refused unless the caller vouches for it with
    ``trusted_pickle=True`` (the reference's ``torch.load`` default, ``trainer.py:1062``, which executes what it loads)."""
    if load_optimizer:
        raise NotImplementedError("Error, restoring optimizer not supported for flexible checkpoint format yet")
    checkpoint = torch.load(checkpoint_path.format(mp_rank=0), map_location="cpu", weights_only=not trusted_pickle)
    state = checkpoint["model_state"]
    with torch.no_grad():
        for k, v in model.named_parameters():
            if k not in state:
                if logger is not None:
                    logger.warning(f"missing {k}")
                continue
            weight = state[k]
            for d, group in enumerate(getattr(v, "sharded_dims_mp", [])):
                if group is None or comm.get_size(group) == 1:
                    continue
                shard = (weight.shape[d] + comm.get_size(group) - 1) // comm.get_size(group)
                weight = torch.split(weight, split_size_or_sections=shard, dim=d)[comm.get_rank(group)]
            v.copy_(weight)
    if load_scheduler and scheduler is not None:
        scheduler.load_state_dict(checkpoint["scheduler_state_dict"])
    return checkpoint["iters"], checkpoint["epoch"]
