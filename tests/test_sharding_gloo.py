"""N > 1 path on CPU: world_size-2 gloo processes shard a clip batch the way
bench.py does (contiguous clip ranges, no data-path collective), then use the
only collectives the job has: barrier, MAX of the step time, SUM of units and
the optional mel-row gather."""
import os
import socket
import sys

import numpy as np

from util import bits_equal
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_range_is_a_balanced_partition(af):
    from audio_flow_rs_amd.shard import all_shards, shard_range
    for n in (0, 1, 7, 256, 1000, 8192):
        for w in (1, 2, 3, 4, 8):
            sh = all_shards(n, w)
            assert sh[0][0] == 0 and sh[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(sh, sh[1:]))
            sizes = [e - b for b, e in sh]
            assert max(sizes) - min(sizes) <= 1
    assert shard_range(8192, 8, 3) == (3072, 4096)              # BASELINE config 4: 1024 / GPU
    with pytest.raises(ValueError):
        shard_range(10, 2, 2)


def test_stream_shards_reproduce_one_shot_framing(af, po):
    # config 5 in miniature: a stream cut into haloed chunks, chunks dealt to 3 ranks; the
    # checker computes every span on its own and the concatenation must be the one-shot result
    from audio_flow_rs_amd.shard import stream_shards
    from util import synth_clip
    hop, chunk = 256, 16 * 256
    for n_total in (chunk * 5 + 1234, chunk * 2, 1023, chunk + 767, chunk + 768):
        x = synth_clip(123, n_total, 1, 16000)
        one = po.features(x, 1, 16000.0)
        shards = stream_shards(n_total, 3, chunk)
        assert sum(len(s) for s in shards) == -(-n_total // chunk)
        rows = [po.features(x[s:s + n], 1, 16000.0) for spans in shards for s, n in spans]
        got = np.concatenate(rows) if rows else np.zeros((0, 80))
        assert got.shape == one.shape
        assert bits_equal(got, one)
    with pytest.raises(ValueError):
        stream_shards(1000, 2, 1000)


def test_chain_stream_shards_cover_the_recording():
    # the chain's time sharding (ChainStream.seek): every step belongs to exactly one rank, a rank's frames start one
    # history in front of its first step and end where its last step's input ends; the last rank holds the tail
    from audio_flow_rs_amd.shard import chain_stream_shards
    need0, step, hist = 186654, 184320, 3453
    for n_frames in (0, 1000, need0 - 1, need0, need0 + step - 1, need0 + step, need0 + 17 * step + 999):
        n_steps = 0 if n_frames < need0 else 1 + (n_frames - need0) // step
        for world in (1, 2, 3, 8, 32):
            shards = chain_stream_shards(n_frames, world, need0, step, hist)
            assert len(shards) == world
            assert [s[4] for s in shards] == [False] * (world - 1) + [True]
            pos = 0
            for k0, k1, lo, hi, flush in shards:
                assert k0 == pos and k1 >= k0
                pos = k1
                if k0 == k1 and not flush:
                    assert lo == hi == 0
                    continue
                assert lo == (0 if k0 == 0 else need0 + (k0 - 1) * step - hist)
                assert hi == (n_frames if flush else need0 + (k1 - 1) * step)
                assert 0 <= lo <= hi <= n_frames
            assert pos == n_steps
    with pytest.raises(ValueError):
        chain_stream_shards(100, 2, 10, 10, 11)


def _worker(rank, world, port, n_clips, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from audio_flow_rs_amd.shard import (gather_rows, max_over_ranks, scatter_rows, shard_range,
                                             sum_over_ranks)
        from oracle import pyoracle as po
        from util import synth_clip
        b, e = shard_range(n_clips, world, rank)
        # root-scatter mode: rank 0 holds every clip, each rank receives its contiguous range
        per = [shard_range(n_clips, world, r)[1] - shard_range(n_clips, world, r)[0] for r in range(world)]
        root = (torch.from_numpy(np.stack([synth_clip(c, 3000, 2, 48000) for c in range(n_clips)]))
                if rank == 0 else None)
        mine = scatter_rows(root, per, 6000, dtype=torch.float32, device=torch.device("cpu"))
        assert mine.shape == (e - b, 6000)
        for i, c in enumerate(range(b, e)):
            assert bits_equal(mine[i].numpy(), synth_clip(c, 3000, 2, 48000))
        # each rank computes ITS clips only (the checker stands in for the device here)
        rows = [po.features(mine[i].numpy(), 2, 48000.0) for i in range(e - b)]
        local = torch.from_numpy(np.concatenate(rows) if rows else np.zeros((0, 80)))
        counts = [(shard_range(n_clips, world, r)[1] - shard_range(n_clips, world, r)[0]) * 8
                  for r in range(world)]
        dist.barrier()
        full = gather_rows(local, counts)
        t = max_over_ranks(1.0 + rank)
        units = sum_over_ranks(float(local.shape[0]))
        if rank == 0:
            q.put((full.numpy(), t, units))
    finally:
        dist.destroy_process_group()


def test_two_rank_sharded_features_equal_single_process(po):
    from util import synth_clip
    n_clips, world = 5, 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_clips, q)) for r in range(world)]
    for p in procs:
        p.start()
    full, t, units = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = np.concatenate([po.features(synth_clip(c, 3000, 2, 48000), 2, 48000.0)
                           for c in range(n_clips)])
    assert full.shape == want.shape == (n_clips * 8, 80)
    assert bits_equal(full, want)          # sharding changes nothing, bit for bit
    assert t == 2.0                            # MAX over ranks of (1 + rank)
    assert units == n_clips * 8
