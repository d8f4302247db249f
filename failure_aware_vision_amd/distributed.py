"""Batch sharding across the GPUs of one node.

Every frame is independent (BatchNorm is folded, dropout masks are keyed by the
GLOBAL frame index), so rank r classifies frames [start_r, stop_r) of the batch
with fully replicated weights and the only exchange is one small all-gather of
packed 8-byte (int32 label, fp32 confidence) records — 2 KiB for 256 frames,
latency-bound on xGMI, so a single ``all_gather_into_tensor`` (RCCL when the
backend is "nccl") and no ring/bucket machinery.  The reference is a single
process (SURVEY.md §5 "Distributed communication backend: None").
"""
from __future__ import annotations

import numpy as np


def shard_range(n_total: int, rank: int, world: int):
    """Contiguous shard [start, stop) of rank; the first n_total % world ranks get one extra frame."""
    q, r = divmod(int(n_total), int(world))
    start = rank * q + min(rank, r)
    return start, start + q + (1 if rank < r else 0)


def pack_records(labels, conf):
    """(int32[n], fp32[n]) torch tensors -> int32[n, 2] records (confidence bit-cast)."""
    import torch
    return torch.stack([labels.to(torch.int32), conf.to(torch.float32).view(torch.int32)], dim=1).contiguous()


def unpack_records(rec):
    import torch
    return rec[:, 0].contiguous(), rec[:, 1].contiguous().view(torch.float32)


def classify_sharded(classifier, local_frames, n_total: int, rank: int, world: int, group=None):
    """Each rank classifies its own shard (global frame indices [start, stop)) and all ranks receive the full
    (labels[n_total], conf[n_total]).

    ``classifier`` is a ``Backend`` on a GPU rank: its confidence head writes the packed 8-byte records straight into
    this rank's slot of the all-gather send buffer (``Backend.classify_records``), one ``all_gather_into_tensor``
    (RCCL) moves them, and the results are views of the receive buffer - no pack, pad or concatenate launches.
    A plain function ``classify_fn(frames, first_index=...) -> (labels, conf)`` is accepted too (CPU tensors / gloo:
    the tests that stand the oracle in for the per-rank classifier), and a ``Backend`` handed HOST frames (a NumPy
    array, or a tensor that is not on its GPU) goes through its ordinary ``classify`` (which uploads them).

    Lifetime of the results: on the direct path with equal shards (and with ``world == 1``) the two tensors are stride-2
    VIEWS of one freshly allocated [n, 2] int32 record buffer - no copy is made; call ``.contiguous()`` before handing a raw
    ``data_ptr()`` to something that assumes dense arrays.  Every call allocates its own buffers, so results stay valid
    across later calls."""
    import torch
    import torch.distributed as dist
    start, stop = shard_range(n_total, rank, world)
    n_local = stop - start
    if n_local > 0 and int(local_frames.shape[0]) != n_local:
        raise ValueError(f"rank {rank} owns frames [{start},{stop}) but was handed {int(local_frames.shape[0])}")
    cap = -(-n_total // world)  # every shard padded to the largest
    direct = hasattr(classifier, "classify_records") and hasattr(local_frames, "is_cuda") and bool(local_frames.is_cuda) and \
        local_frames.device.index == getattr(classifier, "device", local_frames.device.index)
    if not direct and hasattr(classifier, "classify_records"):
        backend = classifier
        classifier = lambda frames, first_index=0: backend.classify(frames, first_index=first_index)   # noqa: E731  (host frames: upload + classify)
    if direct:
        dev = local_frames.device
        send = torch.empty((cap, 2), dtype=torch.int32, device=dev) if n_local == cap else \
            torch.zeros((cap, 2), dtype=torch.int32, device=dev)
        if n_local > 0:
            classifier.classify_records(local_frames, first_index=start, out=send[:n_local])
    else:
        if n_local > 0:
            labels, conf = classifier(local_frames, first_index=start)
            if isinstance(labels, np.ndarray):
                labels, conf = torch.from_numpy(labels), torch.from_numpy(conf)
            rec = pack_records(labels, conf)
            dev = rec.device
        else:
            dev = local_frames.device if hasattr(local_frames, "device") and not isinstance(local_frames, np.ndarray) else "cpu"
            rec = torch.zeros((0, 2), dtype=torch.int32, device=dev)
        send = torch.zeros((cap, 2), dtype=torch.int32, device=dev)
        send[:n_local] = rec
    if world == 1:
        return unpack_records(send[:n_local]) if not direct else (send[:n_local, 0], send[:n_local, 1].view(torch.float32))
    recv = torch.empty((world * cap, 2), dtype=torch.int32, device=dev)
    if dist.get_backend(group) == "gloo" and send.is_cuda:
        # rehearsal only (several ranks sharing one GPU, where RCCL cannot run): the 8-byte records cross the host
        host = torch.empty((world * cap, 2), dtype=torch.int32)
        dist.all_gather_into_tensor(host, send.cpu(), group=group)
        recv.copy_(host)
    else:
        dist.all_gather_into_tensor(recv, send, group=group)
    if n_total == world * cap:                 # equal shards: the receive buffer IS the result
        return recv[:, 0], recv[:, 1].view(torch.float32)
    parts = []
    for r in range(world):
        s, e = shard_range(n_total, r, world)
        parts.append(recv[r * cap:r * cap + (e - s)])
    return unpack_records(torch.cat(parts, dim=0))
