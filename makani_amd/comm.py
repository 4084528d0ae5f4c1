"""Process-group tree for spatial model parallelism over torch.distributed (RCCL on MI355X).

Mirrors the accessors of ``makani/utils/comm.py:32-153`` (``init``, ``get_size``,
``get_rank``, ``get_group``, ``get_names`` ...) without the modulus
``DistributedManager``: the same orthogonal tree

    world -> { model -> { spatial -> {h, w}, matmul -> {fin, fout} }, data }

is built directly from ``torch.distributed.new_group``.  Rank layout: row-major over
(data, fout, fin, w, h) with ``h`` fastest, so one model-parallel instance occupies
consecutive ranks (= the GPUs of one xGMI island) and data parallelism goes across.
One process per GPU; backend "nccl" (RCCL) on GPUs, "gloo" on CPU (tests).
"""
import math
import os

import torch
import torch.distributed as dist

_LEAVES = ["h", "w", "fin", "fout"]
_state = None  # dict: sizes, ranks, groups


def _require():
    return _state is not None and _state["world_size"] > 1


def get_size(name):
    return _state["sizes"][name] if _require() else 1


def get_rank(name):
    return _state["ranks"][name] if _require() else 0


def get_group(name):
    return _state["groups"].get(name) if _state is not None else None


def get_root(name):
    return _state["roots"].get(name, 0) if _require() else 0


def get_world_size():
    return _state["world_size"] if _state is not None else 1


def get_world_rank():
    return _state["world_rank"] if _state is not None else 0


def get_local_rank():
    return _state["local_rank"] if _state is not None else 0


def get_names():
    return list(_state["groups"].keys()) if _state is not None else []


def is_distributed(name):
    return _state is not None and name in _state["groups"]


def _coords(rank, sizes):
    """rank -> {name: coordinate}; h fastest, data slowest."""
    out = {}
    for name in ["h", "w", "fin", "fout", "data"]:
        out[name] = rank % sizes[name]
        rank //= sizes[name]
    return out


def init(model_parallel_sizes=(1, 1, 1, 1), model_parallel_names=("h", "w", "fin", "fout"), verbose=False,
         backend=None):
    """Initialise torch.distributed from the torchrun environment and build the group tree.

    Returns the model-parallel size, like the reference's ``comm.init`` (comm.py:97-153).
    """
    global _state
    if not dist.is_initialized():
        world_size = int(os.environ.get("WORLD_SIZE", "1"))
        if world_size == 1 and "RANK" not in os.environ:
            _state = None
            return 1
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend=backend, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)
    world_size, rank = dist.get_world_size(), dist.get_rank()
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))

    sizes = {n: 1 for n in _LEAVES}
    for n, s in zip(model_parallel_names, model_parallel_sizes):
        if n not in sizes:
            raise ValueError(f"unknown model-parallel group {n}")
        sizes[n] = int(s)
    model_size = math.prod(sizes.values())
    if world_size % model_size:
        raise ValueError(f"world size {world_size} not divisible by model-parallel size {model_size}")
    sizes["data"] = world_size // model_size
    sizes["spatial"] = sizes["h"] * sizes["w"]
    sizes["matmul"] = sizes["fin"] * sizes["fout"]
    sizes["model"] = model_size

    members = {"h": ["h"], "w": ["w"], "fin": ["fin"], "fout": ["fout"], "data": ["data"],
               "spatial": ["h", "w"], "matmul": ["fin", "fout"], "model": ["h", "w", "fin", "fout"]}
    all_coords = [_coords(r, sizes) for r in range(world_size)]
    groups, ranks, roots = {}, {}, {}
    for gname, axes in members.items():
        # ranks sharing every coordinate NOT in `axes` form one group; every rank creates all of them
        buckets = {}
        for r, c in enumerate(all_coords):
            key = tuple(c[a] for a in ["h", "w", "fin", "fout", "data"] if a not in axes)
            buckets.setdefault(key, []).append(r)
        for key in sorted(buckets):
            rl = buckets[key]
            grp = dist.new_group(rl) if len(rl) > 1 else None
            if rank in rl:
                groups[gname] = grp
                ranks[gname] = rl.index(rank)
                roots[gname] = min(rl)
    _state = {"sizes": sizes, "ranks": ranks, "groups": groups, "roots": roots, "world_size": world_size,
              "world_rank": rank, "local_rank": local_rank}
    if verbose and rank == 0:
        print(f"makani_amd.comm: world {world_size}, sizes {sizes}")
    return model_size


def cleanup():
    global _state
    _state = None
    if dist.is_initialized():
        dist.destroy_process_group()
