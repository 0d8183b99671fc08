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
    env.update(env_extra or {})
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
