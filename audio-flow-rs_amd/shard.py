"""Clip sharding across the GPUs of one node (one process per GPU).

Clips are independent end to end (every AudioResampler / VoiceActivityDetector
in the reference owns its own state: resampler.rs:12-21, vad.rs:60-70), so the
batch is split into contiguous clip ranges and NO collective sits on the data
path.  torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box, "gloo"
in the CPU tests) is only used for the barrier, the max-over-ranks time and the
optional gather of per-rank results.
"""
from __future__ import annotations

from typing import List, Tuple


def shard_range(n_items: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Contiguous, balanced [begin, end) of ``n_items`` for ``rank``: the first
    ``n_items % world_size`` ranks get one extra item."""
    if world_size <= 0 or not (0 <= rank < world_size):
        raise ValueError("bad world_size/rank")
    base, extra = divmod(n_items, world_size)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def all_shards(n_items: int, world_size: int) -> List[Tuple[int, int]]:
    return [shard_range(n_items, world_size, r) for r in range(world_size)]


def stream_shards(n_total: int, world_size: int, chunk: int, n_fft: int = 1024,
                  hop: int = 256) -> List[List[Tuple[int, int]]]:
    """Long-form stream (BASELINE config 5): the stream of ``n_total``
    PCM frames is cut into chunks of ``chunk`` frames (a multiple of 16*hop) that each carry a
    read-only right halo of ``n_fft - hop`` frames, and contiguous runs of chunks go to the
    ranks.  Returns, per rank, its ``(start, length)`` spans for
    ``FeaturePipeline.set_batch_spans``; spectrogram frame ``f`` is produced by the chunk that
    holds sample ``f*hop``, so the ranks' outputs, in rank order, are the one-shot framing.  No
    halo is exchanged: every rank reads its halo from the source like the rest of its span."""
    if chunk <= 0 or chunk % (16 * hop) != 0:
        raise ValueError("chunk must be a positive multiple of 16*hop")
    spans, pos = [], 0
    while pos < n_total:
        spans.append((pos, min(chunk + n_fft - hop, n_total - pos)))
        pos += chunk
    return [spans[b:e] for b, e in all_shards(len(spans), world_size)]


def chain_stream_shards(n_frames: int, world_size: int, first_step_frames: int, step_frames: int,
                        history_frames: int) -> List[Tuple[int, int, int, int, bool]]:
    """One long recording through the full chain on ``world_size`` GPUs (BASELINE config 5 on the chain): the
    recording's steps (``ChainStream``: step 0 takes ``first_step_frames`` frames, every later one ``step_frames``)
    are dealt to the ranks in contiguous ranges.  Returns, per rank, ``(step0, step1, frame_lo, frame_hi, flush)``:
    the rank calls ``ChainStream.seek(step0)`` (which answers ``frame_lo``), pushes the recording's frames
    ``[frame_lo, frame_hi)`` -- ``history_frames`` of them are the read-only history in front of step ``step0`` --
    and, when ``flush`` is set (the rank that holds the end of the recording), flushes.  The ranks' rows in rank order
    are the unbroken stream's rows; a rank with ``step0 == step1`` and no flush has nothing to do."""
    if min(first_step_frames, step_frames) <= 0 or not (0 < history_frames <= step_frames) or n_frames < 0:
        raise ValueError("bad step geometry")
    n_steps = 0 if n_frames < first_step_frames else 1 + (n_frames - first_step_frames) // step_frames

    def end_of(step):            # first frame NOT needed by steps < step
        return 0 if step == 0 else first_step_frames + (step - 1) * step_frames

    out = []
    for rank, (k0, k1) in enumerate(all_shards(n_steps, world_size)):
        last = rank == world_size - 1
        lo = 0 if k0 == 0 else end_of(k0) - history_frames
        hi = n_frames if last else end_of(k1)
        if k0 == k1 and not last:
            lo = hi = 0
        out.append((k0, k1, lo, hi, last))
    return out


def max_over_ranks(value: float, device=None) -> float:
    """MAX all-reduce of a python float (identity when not distributed)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value: float, device=None) -> float:
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def gather_rows(local_rows, n_rows_per_rank: List[int]):
    """All-gather of per-rank ``[n_rows_r, width]`` tensors (ragged first dim) into
    the full batch order -- the mel gather of SURVEY.md section 8(e).  Pads to the
    largest shard, one all_gather, then trims."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return local_rows
    world = dist.get_world_size()
    width = local_rows.shape[1]
    cap = max(n_rows_per_rank)
    padded = torch.zeros(cap, width, dtype=local_rows.dtype, device=local_rows.device)
    padded[: local_rows.shape[0]] = local_rows
    bufs = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(bufs, padded)
    return torch.cat([b[:n] for b, n in zip(bufs, n_rows_per_rank)], dim=0)


def scatter_rows(rows_on_root, n_rows_per_rank: List[int], width: int, dtype=None, device=None, src: int = 0):
    """Root-scatter of PCM shards: rank ``src`` holds ``[sum(n), width]``
    rows (e.g. one clip per row), every rank receives its contiguous ``[n_rows_r, width]`` slice.
    One padded ``dist.scatter``; the direct-load mode (every rank ingests its own shard) needs none
    of this and is what bench.py measures."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return rows_on_root
    world, rank = dist.get_world_size(), dist.get_rank()
    cap = max(n_rows_per_rank)
    if rank == src:
        dtype, device = rows_on_root.dtype, rows_on_root.device
        chunks, pos = [], 0
        for n in n_rows_per_rank:
            pad = torch.zeros(cap, width, dtype=dtype, device=device)
            pad[:n] = rows_on_root[pos:pos + n]
            chunks.append(pad)
            pos += n
    else:
        chunks = None
    out = torch.empty(cap, width, dtype=dtype, device=device)
    dist.scatter(out, chunks, src=src)
    return out[: n_rows_per_rank[rank]]
