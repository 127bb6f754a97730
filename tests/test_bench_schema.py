"""bench.py without a GPU: the JSON line's schema (contract of the task statement + the fields the
multi-GPU row needs: scaling mode, gather_ms / scatter_ms, ranks seen, post-timing check) and the
N > 1 control flow of the separately timed collectives at world size 2 over gloo."""
import json
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _meas(world, coll, scaling="weak"):
    return {"steps": 20, "warmup": 3, "world": world, "scaling": scaling, "wall_s": 0.03, "kernel_ms": 1.4,
            "frames_all": 1439232.0 * world, "audio_seconds_all": 7680.0 * world, "clips_local": 256,
            "clips_total": 256 * world, "frames_local": 1439232, "pcm_frames_per_clip": 1440000,
            "kernel": "af_features_v2_kernel", "collectives": coll,
            "check": {"clips": [0, 128, 255], "max_rel_err": 3e-7, "tol": 1e-5, "ok": True}}


def test_json_schema_single_gpu():
    import bench
    none = {"gather_ms": None, "scatter_ms": None, "ranks_seen": 1, "backend": None,
            "gather_bytes_per_rank": None, "scatter_bytes_per_rank": None}
    r = json.loads(json.dumps(bench.make_result(_meas(1, none))))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
              "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline",
              "gather_ms", "scatter_ms", "ranks_seen", "check", "secondary"):
        assert k in r
    assert r["n_gpus"] == 1 and r["gather_ms"] is None and r["scatter_ms"] is None    # "not measured"
    assert r["scaling"] == "weak" and r["higher_is_better"] is True and r["vs_baseline"] is None
    assert "workload" in r["config"] and "model" not in r["config"]
    rf = r["roofline"]
    assert rf["bound"] == "hbm" and rf["peak"] == 8000.0 and rf["unit"] == "GB/s"
    assert rf["algorithmic_bytes_per_launch"] == 256 * 1440000 * 8 + 1439232 * 320 == 3409674240
    assert abs(rf["achieved"] - 3409674240 / 1.4e-3 / 1e9) < 1e-6 and abs(rf["frac"] - rf["achieved"] / 8000) < 1e-12
    assert abs(r["value"] - 1439232 / (0.03 / 20)) < 1e-3 and abs(r["ms_per_step"] - 1.5) < 1e-9


def test_json_schema_strong_scaling_names_config4():
    import bench
    coll = {"gather_ms": 2.5, "scatter_ms": 40.0, "ranks_seen": 8, "backend": "nccl",
            "gather_bytes_per_rank": 8 * 460554240, "scatter_bytes_per_rank": 2949120000}
    m = _meas(8, coll, "strong")
    m["clips_total"], m["clips_local"] = 8192, 1024
    r = bench.make_result(m)
    assert r["scaling"] == "strong" and r["config"]["clips_total"] == 8192 and r["config"]["clips_per_gpu"] == 1024
    assert "config 4" in r["config"]["workload"] and r["ranks_seen"] == 8 and r["gather_ms"] == 2.5


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import bench
        dev = torch.device("cpu")
        mel = torch.full((50 + 0 * rank, 80), float(rank))
        pcm = torch.full((4, 1000, 2), float(rank))
        c = bench.time_collectives(torch, dist, dev, mel, pcm, world, rank)
        if rank == 0:
            q.put(c)
    finally:
        dist.destroy_process_group()


def test_collectives_control_flow_world2_gloo():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    c = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert c["ranks_seen"] == 2 and c["backend"] == "gloo"
    assert c["gather_ms"] > 0 and c["scatter_ms"] > 0
    assert c["gather_bytes_per_rank"] == 2 * 50 * 80 * 4 and c["scatter_bytes_per_rank"] == 4 * 1000 * 2 * 4


def _run_bench(args, extra_env, timeout=300):
    import subprocess
    env = dict(os.environ, **extra_env)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env,
                          capture_output=True, text=True, timeout=timeout)


def test_launcher_spawns_its_own_ranks_dryrun():
    """`bench.py --gpus 2` with no launcher around it starts 2 ranks itself (AF_BENCH_DRYRUN: gloo on
    the CPU, no kernel) and rank 0 prints ONE line that says n_gpus 2 / ranks_seen 2."""
    p = _run_bench(["--gpus", "2", "--steps", "3", "--warmup", "1"], {"AF_BENCH_DRYRUN": "1"})
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["ranks_seen"] == 2 and r["dryrun"] is True
    assert r["collective_backend"] == "gloo" and r["gather_ms"] > 0 and r["scatter_ms"] > 0
    assert r["steps"] == 3 and r["warmup"] == 1


@pytest.mark.skipif(torch.cuda.is_available() and torch.cuda.device_count() >= 8, reason="needs a node with < 8 GPUs")
def test_launcher_refuses_more_ranks_than_gpus():
    """--gpus 8 on a node with fewer GPUs: non-zero exit and a message, never an N = 1 line."""
    p = _run_bench(["--gpus", "8"], {})
    assert p.returncode != 0
    assert "refusing" in p.stderr and not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]


def test_rank_refuses_world_size_mismatch():
    """A launcher that set WORLD_SIZE=1 around `--gpus 2` gets an error, not a mislabelled line."""
    import subprocess
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", AF_BENCH_DRYRUN="1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env,
                       capture_output=True, text=True, timeout=120)
    assert p.returncode != 0 and "WORLD_SIZE=1" in p.stderr
