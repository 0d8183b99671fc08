#!/usr/bin/env python3
"""BER / FER sweep over the AWGN noise LEVEL (BASELINE.json configs[4]): Monte-Carlo throughput + BER curve.

Every point decodes `--frames` frames made entirely on the device: random payloads -> device transmitter (N2) ->
independent AWGN (N3, counter RNG keyed by the global frame index, never reused across points).
With torchrun the frames of every point are sharded over the ranks; the only reduction is the sum of four
integer counters per point (no collective on the data path).  Prints one JSON line per point and a summary.

  python tools/ber_sweep.py --frames 476190 --lo -40 --hi -20 --step 1        # 10^7 frames over 21 points
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=8192, help="frames per noise level (whole job)")
    ap.add_argument("--lo", type=float, default=-40.0)
    ap.add_argument("--hi", type=float, default=-20.0)
    ap.add_argument("--step", type=float, default=1.0)
    ap.add_argument("--batch", type=int, default=32768, help="frames resident per decode call and rank")
    ap.add_argument("--seed", type=int, default=777)
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
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # ONE explicit (non-default) HIP stream shared by torch and the library (the default stream's handle is 0,
    # which the C ABI reads as "create your own stream": torch.randint and the transmitter would then race)
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    rx = modem_amd.Receiver(device=local_rank, stream=stream.cuda_stream)
    spf = rx.tx_frame_samples(6)
    nb = min(args.batch, args.frames)
    d_in = torch.empty((nb, spf, 2), dtype=torch.int16, device=dev)
    d_out = torch.zeros((nb, 5380), dtype=torch.uint8, device=dev)
    d_res = torch.zeros((nb, M.RESULT_DTYPE.itemsize), dtype=torch.uint8, device=dev)
    pop = torch.tensor([bin(i).count("1") for i in range(256)], dtype=torch.int64, device=dev)
    gen = torch.Generator(device=dev)

    levels = np.arange(args.lo, args.hi + 1e-9, args.step)
    lo, hi = shard.block_range(args.frames, rank, world)
    summary = []
    t_all = time.perf_counter()
    for li, db in enumerate(levels):
        counters = [0, 0, 0, 0]          # frames, frame errors, bit errors, header/sync failures
        t0 = time.perf_counter()
        f = lo
        while f < hi:
            n = min(nb, hi - f)
            gidx = li * args.frames + f          # global frame index: distinct noise everywhere
            gen.manual_seed(args.seed * 7919 + gidx)
            d_pay = torch.randint(0, 256, (n, 5380), dtype=torch.uint8, device=dev, generator=gen)
            rx.tx_encode(d_pay.data_ptr(), n, d_in.data_ptr(), mode=6, freq_off=2000, call_sign="ANONYMOUS", channels=2)
            rx.awgn_tile(d_in.data_ptr(), n, d_in.data_ptr(), n, spf, float(db), args.seed, gidx)   # in place
            rx.decode_device(d_in.data_ptr(), M.FMT_S16, 2, spf, spf * 4, n, d_out.data_ptr(), d_res.data_ptr())
            rx.synchronize()
            biterr = torch.zeros(n, dtype=torch.int64, device=dev)
            for lo2 in range(0, n, 8192):
                hi2 = min(lo2 + 8192, n)
                biterr[lo2:hi2] = pop[(d_out[lo2:hi2] ^ d_pay[lo2:hi2]).long()].sum(dim=1)
            res = d_res[:n].cpu().numpy().view(M.RESULT_DTYPE).reshape(-1)
            counters[0] += n
            counters[1] += int((biterr > 0).sum().item())
            # a frame the decoder declares lost returns zeros: count its payload bits as 50 % wrong would
            # hide nothing - we count the actual differing bits against the transmitted payload
            counters[2] += int(biterr.sum().item())
            counters[3] += int((res["status"] != 0).sum())
            f += n
        secs = time.perf_counter() - t0
        secs, counters = shard.reduce_counters((secs, counters), world, dist, dev)
        if rank == 0:
            pt = {"noise_db": float(db), "frames": counters[0], "fer": counters[1] / counters[0],
                  "ber": counters[2] / (43040.0 * counters[0]), "declared_lost": counters[3],
                  "frames_per_s": counters[0] / secs}
            summary.append(pt)
            print(json.dumps(pt), flush=True)
    if rank == 0:
        tot = sum(p["frames"] for p in summary)
        print(json.dumps({"summary": "ber_sweep", "n_gpus": world, "total_frames": tot,
                          "frames_per_s": tot / (time.perf_counter() - t_all), "points": len(summary)}), flush=True)
    rx.close()
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
