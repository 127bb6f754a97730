#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X DSP hot path.

Workload = BASELINE.json configs[1]: a batch of 256 x 30 s stereo 48 kHz clips
(f32 interleaved PCM, resident in HBM) -> to_mono -> 1024-point Hann STFT
(hop 256) -> |X|^2 -> 80-bin mel, fused in ONE HIP kernel (af_features_kernel).
One "step" = one pass of that kernel over the whole batch.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

N > 1: one process per GPU, every rank owns its own 256-clip shard (clips are
independent: weak scaling, no data-path collective); the barrier, the
max-over-ranks time and the unit count are the only collectives (RCCL).

Prints ONE JSON line on rank 0 (contract in the task statement):
  value     = spectrogram frames/s of the whole job, inputs resident in HBM
  roofline  = algorithmic bytes per launch / measured launch time vs 8 TB/s HBM
  cpu_baseline = the oracle's float32 port timed on this host (rank 0, N=1 only)
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SR = 48000
CHANNELS = 2
CLIP_SECONDS = 30
N_FFT, HOP, N_MELS = 1024, 256, 80
CLIPS_PER_GPU = 256
HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s
SEED0 = 0xA5D10F10             # seed = 0xA5D10F10 + clip index


def synth_batch(torch, dev, first_clip: int, n_clips: int, n_frames: int):
    """BASELINE.md synthetic PCM, generated on the device: per clip and channel three
    sines (log-uniform 50 Hz .. 0.45 sr, amplitude 0.05 .. 0.3, random phase) plus uniform
    noise +-0.01, clipped to [-1, 1]; f32 interleaved [clip][frame][channel]."""
    import numpy as np
    pcm = torch.empty(n_clips, n_frames, CHANNELS, device=dev, dtype=torch.float32)
    t = torch.arange(n_frames, device=dev, dtype=torch.float64) / SR
    group = 8                                    # clips synthesised per batch of torch kernels
    for c0 in range(0, n_clips, group):
        g = min(group, n_clips - c0)
        prm = np.empty((g, CHANNELS, 3, 3))
        for i in range(g):
            rng = np.random.default_rng(SEED0 + first_clip + c0 + i)
            for ch in range(CHANNELS):
                for k in range(3):
                    prm[i, ch, k] = (np.exp(rng.uniform(np.log(50.0), np.log(0.45 * SR))),
                                     rng.uniform(0.05, 0.3), rng.uniform(0, 2 * np.pi))
        gen = torch.Generator(device=dev)
        gen.manual_seed(SEED0 + first_clip + c0)
        x = (torch.rand(g, n_frames, CHANNELS, device=dev, generator=gen) * 2 - 1) * 0.01
        P = torch.from_numpy(prm).to(dev)         # [g, ch, 3, (f, a, phase)]
        for k in range(3):
            f = P[:, :, k, 0].unsqueeze(1)        # [g, 1, ch]
            a = P[:, :, k, 1].unsqueeze(1)
            ph = P[:, :, k, 2].unsqueeze(1)
            x += (a * torch.sin(2 * torch.pi * f * t.view(1, -1, 1) + ph)).to(torch.float32)
        pcm[c0:c0 + g] = x.clamp_(-1.0, 1.0)
    return pcm


def cpu_baseline(pcm_host, n_clips: int, n_frames: int) -> dict:
    """The oracle's float32 port (oracle/af_oracle_fast.c) on this host's cores, on a
    bounded sample of the same workload.  A reported baseline, not the target."""
    from oracle import pyoracle as po
    flat = pcm_host.reshape(-1)
    nf = po.stft_n_frames(n_frames, N_FFT, HOP)
    # whole passes over the sample until about 10 s of single-thread work have been timed
    passes, t_single = 0, 0.0
    while passes == 0 or (t_single < 10.0 and passes < 8):
        t0 = time.perf_counter()
        po.features_f32_batch(flat, n_clips, n_frames, CHANNELS, float(SR), N_FFT, HOP, N_MELS, threads=1)
        t_single += time.perf_counter() - t0
        passes += 1
    single = passes * n_clips * nf / t_single
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    threads = max(1, min(po.max_threads(), avail))
    t0 = time.perf_counter()
    po.features_f32_batch(flat, n_clips, n_frames, CHANNELS, float(SR), N_FFT, HOP, N_MELS,
                          threads=threads)
    t_multi = time.perf_counter() - t0
    multi = n_clips * nf / t_multi
    return {
        "value": single, "unit": "frames/s", "cores": 1, "kind": "port",
        "sample": f"{n_clips} of the {CLIPS_PER_GPU} clips ({n_clips * CLIP_SECONDS} s of stereo 48 kHz "
                  f"audio) x {passes} passes, float32 port of the oracle, single thread as in the "
                  f"reference ({t_single:.2f} s of CPU work)",
        "rtf": single * HOP / SR,
        "all_cores": {"value": multi, "unit": "frames/s", "cores": threads,
                      "seconds": round(t_multi, 3)},
    }


def main() -> None:
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on this pool (RCCL)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--clips", type=int, default=CLIPS_PER_GPU, help="clips per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-clips", type=int, default=256)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    import audio_flow_rs_amd as af
    from audio_flow_rs_amd.shard import max_over_ranks, sum_over_ranks

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    # AF_BENCH_REHEARSE=1: rehearse the N > 1 control flow on a one-GPU box (every rank on
    # device 0, gloo instead of RCCL); the numbers of such a run mean nothing
    rehearse = os.environ.get("AF_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    n_frames_pcm = CLIP_SECONDS * SR
    n_clips = args.clips
    pcm = synth_batch(torch, dev, rank * n_clips, n_clips, n_frames_pcm)

    pipe = af.FeaturePipeline(sample_rate=SR, channels=CHANNELS, n_fft=N_FFT, hop=HOP,
                              n_mels=N_MELS, device=local_rank)
    total_frames = pipe.set_batch_uniform(n_clips, n_frames_pcm)
    mel = torch.empty(total_frames, N_MELS, device=dev, dtype=torch.float32)
    stream = torch.cuda.current_stream().cuda_stream      # the kernel runs on torch's stream

    def step():
        pipe.run_device(pcm.data_ptr(), mel.data_ptr(), stream)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    ev0 = torch.cuda.Event(enable_timing=True)
    ev1 = torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        step()
    ev1.record()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t1 = time.perf_counter()

    wall = max_over_ranks(t1 - t0, dev)
    kernel_ms = max_over_ranks(ev0.elapsed_time(ev1) / args.steps, dev)   # HIP events, same stream
    frames_all = sum_over_ranks(float(total_frames), dev)
    audio_seconds_all = sum_over_ranks(float(n_clips * CLIP_SECONDS), dev)

    ms_per_step = wall / args.steps * 1e3
    value = frames_all / (wall / args.steps)
    # algorithmic bytes of ONE launch on one GPU: every stereo f32 input frame
    # once (8 B) + every mel row once (80 x 4 B)
    alg_bytes = n_clips * n_frames_pcm * CHANNELS * 4 + total_frames * N_MELS * 4
    achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
    traffic = None
    tp = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(tp):
        try:
            traffic = json.load(open(tp)).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None

    result = {
        "metric": "spectrogram frames/sec (whole node) + real-time factor, 48 kHz stereo batch",
        "value": value,
        "unit": "frames/s",
        "rtf": audio_seconds_all / (wall / args.steps),
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": f"BASELINE config 2: {n_clips} x {CLIP_SECONDS} s stereo {SR} Hz f32 per GPU "
                        f"-> to_mono -> STFT {N_FFT}/{HOP} Hann -> |X|^2 -> {N_MELS} mel (fused)",
            "clips_per_gpu": n_clips, "frames_per_gpu": total_frames,
            "sharding": "contiguous clip ranges, one process per GPU, no data-path collective",
        },
        "roofline": {
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
            "kernel": "af_features_kernel", "kernel_ms": kernel_ms,
            "algorithmic_bytes_per_launch": alg_bytes,
        },
    }

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        k = max(1, min(args.cpu_clips, n_clips))
        result["cpu_baseline"] = cpu_baseline(pcm[:k].cpu().numpy(), k, n_frames_pcm)
    elif rank == 0:
        result["cpu_baseline"] = None

    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
