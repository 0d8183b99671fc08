#!/usr/bin/env python3
"""bench.py -- decoded frames/s of the MI355X-native OFDM receive path (BASELINE.json metric).

Workload (config.workload): BASELINE.json configs[2] -- a batch of 65536 mode-6 8 kHz frames,
2-channel analytic int16, AWGN at noise level -30 dB, resident in HBM before the timed region.
A step = one pass of the whole hot path (sync -> header/OSD -> 51 FFTs -> Theil-Sen -> soft demap
-> polar SCL -> CRC/pack) over the batch.  Frames are independent: with N GPUs every rank decodes
its own 65536 frames (weak scaling), no collective on the data path.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

B_FRAME_2CH = 95200 * 2 * 2 + 5380      # algorithmic bytes per frame, SURVEY 8(d): 386180
HBM_PEAK_GBS = 8000.0                   # MI355X_MICROARCH.md: HBM3E 8 TB/s
TRAFFIC_FILE = "r01_v10_traffic.json"      # PMC bytes of k_polar, written by tools/profile_round.sh


def cpu_baseline(pcm_sample, payload_ref, threads, ch=2):
    """the oracle (CPU restatement of decode.cc; the reference's own deps are absent) timed on the
    host cores on a bounded sample of the same workload.  This is the ONLY place bench.py touches oracle/."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import oracle_lib as O
    n = pcm_sample.shape[0]
    out = np.zeros((n, 5380), np.uint8)
    res = np.zeros(n * 56, np.uint8)
    lib = O.lib()
    pcm_sample = np.ascontiguousarray(pcm_sample)
    t0 = time.perf_counter()
    used = lib.orc_decode_batch(O.ptr(pcm_sample), O.FMT_S16, ch, pcm_sample.shape[1], pcm_sample.shape[1] * 2 * ch,
                                n, 8, O.ptr(out), O.ptr(res), threads)
    dt = time.perf_counter() - t0
    ok = int((out == payload_ref).all(axis=1).sum())
    return {"value": n / dt, "unit": "frames/s", "cores": int(used), "kind": "port",
            "sample": "%d frames of this batch (first %d), oracle = C restatement of decode.cc, list 8, "
                      "gcc -O2 strict IEEE, %d OpenMP threads, %d/%d payloads correct, %.1f s wall"
                      % (n, n, used, ok, n, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--frames", type=int, default=65536, help="frames per GPU per step")
    ap.add_argument("--noise-db", type=float, default=-30.0)
    ap.add_argument("--chunk", type=int, default=0, help="resident frames per pass (0 = library default)")
    ap.add_argument("--cpu-frames", type=int, default=-1, help="cpu_baseline sample size (-1 auto, 0 off)")
    ap.add_argument("--seed", type=int, default=2021)
    ap.add_argument("--unique", type=int, default=-1,
                    help="distinct payloads / clean frames made by the device transmitter (-1 = every frame distinct, "
                         "0 = tile the 4 committed fixture frames instead)")
    ap.add_argument("--channels", type=int, default=2, choices=[1, 2],
                    help="2 = configs[2] (analytic + AWGN, the headline workload); 1 = configs[1] flavour: clean 16-bit mono frames "
                         "(exercises the D1 front end)")
    ap.add_argument("--rate", type=int, default=8000, choices=[8000, 16000, 44100, 48000],
                    help="sample rate of the frames (the headline metric is 8000; the others are the N4 instantiations)")
    ap.add_argument("--mode", type=int, default=6, choices=range(6, 14), help="operation mode of the frames (headline: 6)")
    ap.add_argument("--impair", action="store_true",
                    help="configs[3]: every frame also goes through the device channel chain multipath -> CFO +234.567 Hz "
                         "-> SFO +147 ppm (README.md:49) before the AWGN")
    args = ap.parse_args()

    import numpy as np
    import torch
    import modem_amd
    import modem_amd.ofdmrx as M
    from modem_amd import shard

    rank, local_rank, world = shard.env_rank()
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    assert world == args.gpus or world == 1, "launch with torchrun --nproc-per-node == --gpus"
    n_gpus = world
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    B = args.frames
    ch = args.channels
    # ONE explicit (non-default) HIP stream shared by torch and the library: the default stream's handle is 0,
    # which the C ABI reads as "create your own stream", and two streams would race on the device buffers
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    rx = modem_amd.Receiver(device=local_rank, chunk_frames=args.chunk, stream=stream.cuda_stream, sample_rate=args.rate)
    spf = rx.tx_frame_samples(args.mode)
    d_in = torch.empty((B, spf, ch), dtype=torch.int16, device=dev)
    d_out = torch.zeros((B, 5380), dtype=torch.uint8, device=dev)
    d_res = torch.zeros((B, M.RESULT_DTYPE.itemsize), dtype=torch.uint8, device=dev)
    t_gen = time.perf_counter()
    if args.unique != 0:
        # synthetic batch made entirely on the device: random payloads -> device transmitter (N2) -> AWGN (N3).
        # Payload RNG and channel RNG are keyed by the GLOBAL frame index, so ranks never repeat each other.
        U = B if args.unique < 0 else min(args.unique, B)
        g = torch.Generator(device=dev)
        g.manual_seed(args.seed * 1000003 + rank)
        d_pay = torch.randint(0, 256, (U, 5380), dtype=torch.uint8, device=dev, generator=g)
        n_base = U
        if U == B:
            d_clean = d_in                        # transmit straight into the batch, add the noise in place
        else:
            d_clean = torch.empty((U, spf, ch), dtype=torch.int16, device=dev)
        rx.tx_encode(d_pay.data_ptr(), U, d_clean.data_ptr(), mode=args.mode, freq_off=2000, call_sign="ANONYMOUS", channels=ch)
        if ch == 2 and args.impair:
            d_imp = torch.empty((U, spf, ch), dtype=torch.int16, device=dev)
            taps = [(0, 1 + 0j), (5, 0.35 - 0.1j), (11, -0.2 + 0.2j), (23, 0.1 + 0.05j)]
            for lo in range(0, U, 8192):
                hi = min(lo + 8192, U)
                rx.channel(d_clean[lo:hi].data_ptr(), d_imp[lo:hi].data_ptr(), hi - lo, spf, cfo_hz=234.567, sfo_ppm=147.0,
                           multipath=taps)
            rx.awgn_tile(d_imp.data_ptr(), U, d_in.data_ptr(), B, spf, args.noise_db, args.seed,
                         shard.frame_seed_offset(B, rank))
            del d_imp
        elif ch == 2:
            rx.awgn_tile(d_clean.data_ptr(), U, d_in.data_ptr(), B, spf, args.noise_db, args.seed,
                         shard.frame_seed_offset(B, rank))
        elif U != B:
            idx = torch.arange(B, device=dev) % U
            for lo in range(0, B, 4096):
                d_in[lo:lo + 4096] = d_clean[idx[lo:lo + 4096]]
        source = "device transmitter, %d distinct random payloads" % U
        if ch == 2 and args.impair:
            source += ", device channel chain multipath(4 taps) -> CFO +234.567 Hz -> SFO +147 ppm (configs[3])"
    else:
        fx = np.load(os.path.join(ROOT, "tests", "golden", "base_frames_2ch.npz"))
        base = fx["pcm"]
        n_base = base.shape[0]
        d_pay = torch.from_numpy(fx["payload"]).to(dev)
        d_base = torch.from_numpy(base).to(dev)
        if ch == 2:
            rx.awgn_tile(d_base.data_ptr(), n_base, d_in.data_ptr(), B, spf, args.noise_db, args.seed,
                         shard.frame_seed_offset(B, rank))
        else:   # a 1-channel WAV from encode is the real part of the same stream (encode.cc:127-128)
            idx = torch.arange(B, device=dev) % n_base
            for lo in range(0, B, 4096):
                d_in[lo:lo + 4096, :, 0] = d_base[idx[lo:lo + 4096], :, 0]
        source = "%d committed fixture frames (tests/golden/base_frames_2ch.npz) tiled" % n_base
    rx.synchronize()
    torch.cuda.synchronize()
    gen_s = time.perf_counter() - t_gen

    def step():
        rx.decode_device(d_in.data_ptr(), M.FMT_S16, ch, spf, spf * 2 * ch, B, d_out.data_ptr(), d_res.data_ptr())

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    polar_ms, polar_launches = 0.0, 0
    stage_ms = {}
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        t = rx.timing()      # hipEvents on the stream the kernels run on; syncs the stream
        polar_ms += t["polar"][0]
        polar_launches += t["polar"][1]
        for k, v in t.items():
            stage_ms[k] = stage_ms.get(k, 0.0) + v[0]
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    secs = time.perf_counter() - t0

    res = d_res.cpu().numpy().view(M.RESULT_DTYPE).reshape(-1)
    pop = torch.tensor([bin(i).count("1") for i in range(256)], dtype=torch.int64, device=dev)
    frame_err = bit_err = 0
    for lo in range(0, B, 8192):
        ref = d_pay[torch.arange(lo, min(lo + 8192, B), device=dev) % n_base]
        be = pop[(d_out[lo:lo + 8192] ^ ref).long()].sum(dim=1)
        frame_err += int((be > 0).sum().item())
        bit_err += int(be.sum().item())
    ok_status = int((res["status"] == 0).sum())
    secs, (frames_total, frame_err, bit_err, ok_status) = shard.reduce_counters(
        (secs, [B * args.steps, frame_err, bit_err, ok_status]), world, dist, dev)

    if rank == 0:
        value = frames_total / secs
        frames_per_launch = B * args.steps / max(polar_launches, 1)
        avg_launch_s = polar_ms / 1e3 / max(polar_launches, 1)
        b_frame = spf * 2 * ch + 5380          # compulsory input + output bytes per frame (386180 for the headline)
        achieved = b_frame * frames_per_launch / avg_launch_s / 1e9 if avg_launch_s > 0 else 0.0
        # HBM bytes per k_polar launch from the committed PMC passes (bench.py cannot run the profiler on itself):
        # scaled to this run's frames per launch; null when the file is absent
        traffic, traffic_src = None, None
        try:
            with open(os.path.join(ROOT, "profiles", TRAFFIC_FILE)) as fh:
                tj = json.load(fh)
            # counted KiB -> bytes with the calibration of the same session (FETCH_SIZE counts half of this access
            # pattern's bytes on gfx950, WRITE_SIZE all of them: tools/pmc_calib.hip)
            traffic = ((tj["fetch_KiB"] * tj.get("fetch_scale", 1.0) + tj["write_KiB"] * tj.get("write_scale", 1.0))
                       * 1024.0 * frames_per_launch / tj["frames_per_launch"])
            traffic_src = ("profiles/%s (rocprofv3 PMC FETCH_SIZE x %.2f + WRITE_SIZE x %.2f, separate passes, calibrated on "
                           "tools/pmc_calib.hip), bytes per launch" % (TRAFFIC_FILE, tj.get("fetch_scale", 1.0), tj.get("write_scale", 1.0)))
        except (OSError, KeyError, ValueError):
            pass
        line = {
            "metric": "decoded frames/sec + BER, mode-6 8kHz OFDM, batch 65536, 1/2/4/8 MI355X",
            "value": value, "unit": "frames/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * secs / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": ("configs[%d]%s: batch %d analytic (2-ch int16) mode-%d %g kHz frames per GPU, AWGN noise "
                                    "level %g dB, inputs resident in HBM; %s, on-device noise keyed by frame index"
                                    % (3 if args.impair else 2, "" if (args.rate == 8000 and args.mode == 6) else " variant (not the headline workload)",
                                       B, args.mode, args.rate / 1000.0, args.noise_db, source)) if ch == 2 else
                                   ("configs[1] flavour: batch %d clean 16-bit mono mode-6 8 kHz frames per GPU, inputs resident "
                                    "in HBM; %s" % (B, source)),
                       "frames_per_gpu": B, "list_size": 8, "chunk_frames": rx.chunk_frames, "parallelism": "frames x%d" % n_gpus},
            "ber": bit_err / (43040.0 * B * n_gpus), "fer": frame_err / float(B * n_gpus),
            "frames_ok": ok_status, "frames": B * n_gpus,
            "roofline": {"bound": "hbm", "kernel": "k_polar (D9 SCL)", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "traffic_GBps": (traffic / avg_launch_s / 1e9) if (traffic and avg_launch_s > 0) else None,
                         "frames_per_launch": frames_per_launch, "avg_launch_ms": 1e3 * avg_launch_s,
                         "algorithmic_bytes_per_frame": b_frame},
            "stage_ms_per_step": {k: v / args.steps for k, v in stage_ms.items()},
            "input_generation_s": gen_s,
        }
        ncpu = args.cpu_frames
        if n_gpus == 1 and ncpu != 0:
            threads = min(os.cpu_count() or 1, 32)
            if ncpu < 0:
                ncpu = 4 * threads
            ncpu = min(ncpu, B)
            sample = d_in[:ncpu].cpu().numpy()
            ref_cpu = d_pay[torch.arange(ncpu, device=dev) % n_base].cpu().numpy()
            line["cpu_baseline"] = cpu_baseline(sample, ref_cpu, threads, ch)
        print(json.dumps(line), flush=True)
    rx.close()
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
