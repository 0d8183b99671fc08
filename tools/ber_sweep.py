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
    ap.add_argument("--tx-reuse", type=int, default=1,
                    help="decode every transmitted batch this many times, each with fresh noise (1 = every frame freshly "
                         "transmitted, the default; the noise realisations are always distinct)")
    ap.add_argument("--levels", type=float, nargs="*", default=None, help="explicit noise levels in dB (overrides --lo/--hi/--step)")
    ap.add_argument("--resume", default=None,
                    help="file: every finished point is appended to it as a JSON line; points already in it are skipped "
                         "(a 10^7-frame sweep can be restarted; rank 0 reads and writes the file)")
    ap.add_argument("--dump", default=None, help="directory: keep every batch's PCM and payloads as .npy (tests, small runs only)")
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
    # a second handle + stream makes the inputs: the transmitter and the channel of batch b+1 run beside the decode of
    # batch b (the transmitter alone is 60 % of a decode), two input buffers in rotation
    tx_stream = torch.cuda.Stream(device=dev)
    tx = modem_amd.Receiver(device=local_rank, stream=tx_stream.cuda_stream)
    spf = rx.tx_frame_samples(6)
    nb = min(args.batch, args.frames)
    d_in = [torch.empty((nb, spf, 2), dtype=torch.int16, device=dev) for _ in range(2)]
    d_pay = [torch.empty((nb, 5380), dtype=torch.uint8, device=dev) for _ in range(2)]
    d_out = torch.zeros((nb, 5380), dtype=torch.uint8, device=dev)
    d_res = torch.zeros((nb, M.RESULT_DTYPE.itemsize), dtype=torch.uint8, device=dev)
    pop = torch.tensor([bin(i).count("1") for i in range(256)], dtype=torch.int64, device=dev)
    gen = torch.Generator(device=dev)
    ev_ready = [torch.cuda.Event() for _ in range(2)]    # inputs of the batch in buffer q are complete (tx stream)
    ev_free = [torch.cuda.Event() for _ in range(2)]     # the decode that read buffer q is done (decode stream)

    levels = np.asarray(args.levels) if args.levels else np.arange(args.lo, args.hi + 1e-9, args.step)
    done_pts = []
    if args.resume and os.path.exists(args.resume):
        with open(args.resume) as fh:
            done_pts = [json.loads(ln) for ln in fh if ln.startswith("{") and "noise_db" in ln]
        done_pts = [p for p in done_pts if p.get("frames") == args.frames and p.get("seed", args.seed) == args.seed]
    all_levels = levels
    skip = {round(p["noise_db"], 6) for p in done_pts}
    # (the noise seed of a frame is keyed by its level's index in the FULL list, so a resumed point equals a fresh one)
    level_index = {round(float(db), 6): i for i, db in enumerate(all_levels)}
    levels = np.asarray([db for db in all_levels if round(float(db), 6) not in skip])
    lo, hi = shard.block_range(args.frames, rank, world)
    # the whole job as one list of batches, so that the pipeline runs across noise levels too
    work = []
    for li, db in enumerate(levels):
        f = lo
        while f < hi:
            n = min(nb, hi - f)
            work.append((li, float(db), f, n))
            f += n
    gli = [level_index[round(float(db), 6)] for db in levels]   # position of each remaining level in the full list

    d_clean = torch.empty((nb, spf, 2), dtype=torch.int16, device=dev) if args.tx_reuse > 1 else None
    made, made_n = [0], [0]

    def make(w, q):
        """random payloads -> device transmitter -> AWGN for batch w into buffer q, on the tx stream.  With --tx-reuse K
        the clean waveforms (and payloads) of a batch serve K consecutive batches of the same size, each with its own noise"""
        li, db, f, n = w
        gidx = gli[li] * args.frames + f         # global frame index: distinct noise everywhere
        with torch.cuda.stream(tx_stream):
            tx_stream.wait_event(ev_free[q])
            fresh = args.tx_reuse <= 1 or made[0] % args.tx_reuse == 0 or n != made_n[0]
            if fresh:
                gen.manual_seed(args.seed * 7919 + gidx)
                d_pay[q][:n] = torch.randint(0, 256, (n, 5380), dtype=torch.uint8, device=dev, generator=gen)
                dst = d_clean if d_clean is not None else d_in[q]
                tx.tx_encode(d_pay[q].data_ptr(), n, dst.data_ptr(), mode=6, freq_off=2000, call_sign="ANONYMOUS", channels=2)
                made_n[0] = n
                made[0] = 0
            else:
                d_pay[q][:n] = d_pay[q ^ 1][:n]
            src = d_clean if d_clean is not None else d_in[q]
            tx.awgn_tile(src.data_ptr(), n, d_in[q].data_ptr(), n, spf, db, args.seed, gidx)   # in place when not reusing
            made[0] += 1
            ev_ready[q].record(tx_stream)

    summary = []
    acc = torch.zeros((len(levels), 4), dtype=torch.int64, device=dev)   # frames, frame errors, bit errors, declared lost
    t_all = time.perf_counter()
    t_level = [None] * len(levels)
    for q in range(2):
        ev_free[q].record(stream)
    if work:
        make(work[0], 0)
    for b, w in enumerate(work):
        li, db, f, n = w
        q = b & 1
        if t_level[li] is None:
            t_level[li] = time.perf_counter()
        stream.wait_event(ev_ready[q])
        rx.decode_device(d_in[q].data_ptr(), M.FMT_S16, 2, spf, spf * 4, n, d_out.data_ptr(), d_res.data_ptr())
        # counters stay on the device (no host synchronisation per batch): differing bits against the transmitted
        # payload (a frame the decoder declares lost returns zeros - its actual differing bits are counted)
        biterr = torch.zeros(n, dtype=torch.int64, device=dev)
        for lo2 in range(0, n, 8192):
            hi2 = min(lo2 + 8192, n)
            biterr[lo2:hi2] = pop[(d_out[lo2:hi2] ^ d_pay[q][lo2:hi2]).long()].sum(dim=1)
        status = d_res[:n].view(torch.int32)[:, 0]
        acc[li] += torch.stack([torch.tensor(n, device=dev), (biterr > 0).sum(), biterr.sum(), (status != 0).sum()])
        if args.dump:
            torch.cuda.synchronize()
            os.makedirs(args.dump, exist_ok=True)
            np.save(os.path.join(args.dump, "pcm_r%d_b%d.npy" % (rank, b)), d_in[q][:n].cpu().numpy())
            np.save(os.path.join(args.dump, "pay_r%d_b%d.npy" % (rank, b)), d_pay[q][:n].cpu().numpy())
            np.save(os.path.join(args.dump, "out_r%d_b%d.npy" % (rank, b)), d_out[:n].cpu().numpy())
        ev_free[q].record(stream)
        if b + 1 < len(work):
            make(work[b + 1], q ^ 1)             # blocks the host while the transmitter runs - beside the decode above
        last_of_level = b + 1 == len(work) or work[b + 1][0] != li
        if last_of_level:
            torch.cuda.synchronize()
            secs = time.perf_counter() - t_level[li]
            secs, counters = shard.reduce_counters((secs, acc[li].tolist()), world, dist, dev)
            if rank == 0:
                pt = {"noise_db": db, "frames": counters[0], "fer": counters[1] / counters[0],
                      "ber": counters[2] / (43040.0 * counters[0]), "declared_lost": counters[3],
                      "frames_per_s": counters[0] / secs, "seed": args.seed}
                summary.append(pt)
                print(json.dumps(pt), flush=True)
                if args.resume:
                    with open(args.resume, "a") as fh:
                        fh.write(json.dumps(pt) + "\n")
    tx.close()
    if rank == 0:
        summary = done_pts + summary
        tot = sum(p["frames"] for p in summary)
        print(json.dumps({"summary": "ber_sweep", "n_gpus": world, "total_frames": tot, "tx_reuse": args.tx_reuse,
                          "frames_per_s": tot / (time.perf_counter() - t_all), "points": len(summary)}), flush=True)
    rx.close()
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
