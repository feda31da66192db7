"""Frame sharding across the GPUs of one node: one process per GPU, no per-frame collectives.

The path shards by independent units (frames), so the only communication is ONE
broadcast of the packed constant kernels from rank 0 at init (RCCL when the backend is "nccl", gloo on
CPU in the tests).  The reference itself has no multi-device code at all
(slam_recognition/recognition_testing.py:64 hard-codes '/device:GPU:0').
"""
import os

import numpy as np

from .pipeline import default_constants, pack_constants, unpack_constants


def world():
    """(rank, world_size, local_rank) from the torchrun environment (1 process -> (0, 1, 0))."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", "0")))


def init(backend=None, device=None):
    """Initialise torch.distributed when launched with more than one rank.  Returns (rank, world_size, local_rank).
    ``device``: GPU index of this rank (default: LOCAL_RANK)."""
    rank, size, local = world()
    if device is not None:
        local = int(device)
    if size > 1 or os.environ.get("SILENT_DIST_FORCE") == "1":   # FORCE: a 1-rank group (exercises the RCCL calls on one GPU)
        import torch
        import torch.distributed as dist
        if not dist.is_initialized():
            if backend is None:
                backend = "nccl" if torch.cuda.is_available() else "gloo"
            if backend == "nccl":
                torch.cuda.set_device(local)
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group(backend=backend, rank=rank, world_size=size)
    return rank, size, local


def shard_frame_indices(n_frames_total, rank, world_size):
    """Frame i goes to rank i mod G."""
    return list(range(rank, n_frames_total, world_size))


def shard_stream_indices(n_streams, rank, world_size):
    """Stateful mode (boosting): a camera stream carries per-pixel state from frame to
    frame, so whole streams are sharded -- stream s lives on rank s mod G for all of its frames, in order."""
    return list(range(rank, n_streams, world_size))


def broadcast_constants(mode, n_orient=4, device=None):
    """Rank 0 generates the constant kernels; everyone receives one flat float32 blob (< 8 KB).

    The layout (names + shapes) is a pure function of (mode, n_orient), so only the floats travel."""
    import torch
    import torch.distributed as dist
    local = default_constants(mode, n_orient)
    blob, layout = pack_constants(local)
    if not (dist.is_available() and dist.is_initialized()):
        return unpack_constants(blob, layout)
    on_gpu = dist.get_backend() == "nccl"
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")) if device is None else device) if on_gpu else "cpu"
    if dist.get_rank() == 0:
        t = torch.from_numpy(blob.copy()).to(dev)
    else:
        t = torch.zeros(blob.shape[0], dtype=torch.float32, device=dev)    # receivers do NOT keep their own copy
    dist.broadcast(t, src=0)
    return unpack_constants(t.cpu().numpy(), layout)


def max_over_ranks(value):
    """MAX all-reduce of a Python float (step time of the slowest rank)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    on_gpu = dist.get_backend() == "nccl"
    t = torch.tensor([float(value)], dtype=torch.float64,
                     device=torch.device("cuda", torch.cuda.current_device()) if on_gpu else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def device_identity(local):
    """Who this rank's GPU is, for the scaling record: name, PCI bus id, uuid (what torch reports for device ``local``)."""
    import torch
    p = torch.cuda.get_device_properties(local)
    bus = "%04x:%02x:%02x" % (getattr(p, "pci_domain_id", 0), getattr(p, "pci_bus_id", -1) & 0xff, getattr(p, "pci_device_id", 0))
    return {"device_name": p.name, "pci_bus_id": bus, "uuid": str(getattr(p, "uuid", "")),
            "gcn_arch": getattr(p, "gcnArchName", "")}


def gather_records(record):
    """Every rank contributes one picklable record; every rank gets the list ordered by rank (all_gather_object)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return [record]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, record)
    return out


def backend_name():
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return "none (single process)"
    return "%s%s" % (dist.get_backend(), " (RCCL)" if dist.get_backend() == "nccl" else "")


def duplicate_devices(records):
    """Ranks that report the same GPU (same PCI bus id): [(rank_a, rank_b, bus)].  Empty = N distinct devices."""
    seen, dup = {}, []
    for r in records:
        key = r.get("pci_bus_id")
        if key in seen:
            dup.append((seen[key], r["rank"], key))
        else:
            seen[key] = r["rank"]
    return dup


def barrier():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        if dist.get_backend() == "nccl":
            import torch
            dist.barrier(device_ids=[torch.cuda.current_device()])      # this rank's GPU, not a guess from the rank
        else:
            dist.barrier()


def finalize():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()


def synthetic_frame(index, h, w, c):
    """noise frame with seed = global frame index."""
    return np.random.default_rng(index).integers(0, 256, (h, w, c)).astype(np.float32)
