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
