#!/usr/bin/env python3
"""bench.py -- decoded frames/s of the MI355X-native OFDM receive path (BASELINE.json metric).

Workload (config.workload): BASELINE.json configs[2] -- a batch of 65536 mode-6 8 kHz frames,
2-channel analytic int16, AWGN at noise level -30 dB (about +20 dB SNR: the transmitter's output sits near -9 dBFS,
encode.cc:109,135), resident in HBM before the timed region.
A step = one pass of the whole hot path (sync -> header/OSD -> 51 FFTs -> Theil-Sen -> soft demap
-> polar SCL -> CRC/pack -> payload bytes on the host) over the batch.  Frames are independent: no collective on the
data path.  --scaling weak (default): every rank decodes --frames frames; --scaling strong (configs[3] flavour):
--frames frames in total, split into contiguous blocks (modem_amd/shard.py).

  python bench.py --gpus N --steps K --warmup W        spawns N ranks by itself (torch.distributed.run as a child
                                                       process, before this process touches a GPU)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...      (the driver's form)

Prints ONE JSON line on rank 0.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# (GPU_MAX_HW_QUEUES is left at the runtime's default, 4: HIP maps streams onto that many hardware queues and kernels of streams that
# share one run in order.  With 8 or 16 the two-lane legs below gain more (+7 % instead of +3.5 %), but a list-decoder stream with a
# queue of its own lets the EMPTY k_polar launches of the one-lane path - 4096 workgroups that leave at once - contend with the front
# kernels: 0.9 ms per chunk whenever the mapping falls that way.  profiles/r06_hw_queues_and_two_lanes.txt)

B_FRAME_2CH = 95200 * 2 * 2 + 5380      # algorithmic bytes per frame, SURVEY 8(d): 386180
HBM_PEAK_GBS = 8000.0                   # MI355X_MICROARCH.md: HBM3E 8 TB/s
TRAFFIC_FILE = "r06_traffic.json"       # PMC bytes + VALU instructions per kernel and launch, written by tools/profile_round.sh
# Vector-instruction issue, two ceilings (round-4 verdict: the round-4 line priced a wave64 VALU instruction at 4 cycles and came out
# above 1.0): (i) MI355X_MICROARCH.md: a SIMD retires a wave64 fp32 VALU instruction in 2 cycles, 2.4 GHz nominal engine clock,
# 256 CUs x 4 SIMDs; (ii) what tools/ubench_issue.hip MEASURES on the box as ns per VALU wave-instruction per SIMD with every CU
# busy and 4 - 5 waves per SIMD (a time, so no clock enters it; profiles/<ISSUE_FILE>, re-run with the traffic passes)
SCLK_GHZ = 2.4
VALU_CYCLES_GUIDE = 2.0
N_SIMD = 1024
ISSUE_FILE = "r06_issue_rates_ubench.txt"
# stage of ofdmrx_get_timing -> (kernel, source file whose hash guards the committed traffic figure)
STAGE_KERNELS = {"front": ("k_mono_carries", "k_sync.hip"), "sync": ("k_sync", "k_sync.hip"), "header": ("k_header", "k_header.hip"), "demod": ("k_demod", "k_demod.hip"),
                 "theilsen": ("k_theil_sen", "k_theilsen.hip"), "llr": ("k_back", "k_finish.hip"),
                 "polar": ("k_polar", "k_polar.hip"), "finish": ("k_finish", "k_finish.hip"), "sc": ("k_sc", "k_sc.hip")}
CPU_FRAMES_DEFAULT = 2048               # cpu_baseline: 16 frames per thread on 128 threads (about 16 s of all-core work)
METRIC = "decoded frames/sec + BER, mode-6 8kHz OFDM, batch 65536, 1/2/4/8 MI355X"


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames", type=int, default=65536, help="frames per GPU per step (weak) / in total per step (strong)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak")
    ap.add_argument("--noise-db", type=float, default=-30.0)
    ap.add_argument("--chunk", type=int, default=0, help="resident frames per pass (0 = library default)")
    ap.add_argument("--cpu-frames", type=int, default=-1, help="cpu_baseline sample size (-1 auto, 0 off)")
    ap.add_argument("--host-frames", type=int, default=-1, help="frames of the host-pointer leg value_host (-1 auto, 0 off)")
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
    ap.add_argument("--list", type=int, default=8, choices=[4, 8], help="SCL list size (headline: 8 = the reference's AVX2 build)")
    ap.add_argument("--impair", action="store_true",
                    help="configs[3]: every frame also goes through the device channel chain multipath -> CFO +234.567 Hz "
                         "-> SFO +147 ppm (README.md:49) before the AWGN")
    ap.add_argument("--scl-steps", type=int, default=-1,
                    help="steps of the extra leg with the list decoder forced for every frame (value_scl_forced); -1 = min(steps, 5), 0 = off")
    ap.add_argument("--leg-steps", type=int, default=4,
                    help="steps of each of the two extra legs value_config3 / value_noise_m20 (one GPU, headline workload; 0 = off)")
    ap.add_argument("--dry-run", action="store_true",
                    help="no GPU: the ranks only rendezvous (gloo), shard the frames and reduce counters (launcher test)")
    return ap.parse_args(argv)


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(n, argv):
    """`python bench.py --gpus N` without a torchrun environment: start the N ranks as a CHILD process tree (never an
    exec) before this process has made any GPU call, pass their output through and exit with their code."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % n,
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    return subprocess.call(cmd, env=env)


def file_sha(path):
    try:
        with open(path, "rb") as fh:
            return hashlib.sha256(fh.read()).hexdigest()[:16]
    except OSError:
        return None


def cpu_baseline(pcm_sample, payload_ref, ch=2):
    """The oracle (CPU restatement of decode.cc; the reference's own deps are absent, so `decode` cannot be built)
    timed on THIS host's cores on a bounded sample of the same workload, as BASELINE.md section 3 asks: all cores and
    one thread, strict-IEEE -O3 -march=native and -Ofast -march=native (the reference's Makefile:2 flags) builds made
    here, plus the -O2 parity oracle.  This is the ONLY place bench.py touches oracle/."""
    import ctypes as C
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import oracle_lib as O
    n = pcm_sample.shape[0]
    pcm_sample = np.ascontiguousarray(pcm_sample)
    host_cores = os.cpu_count() or 1
    threads = min(host_cores, 128)

    def run(lib, frames, thr):
        out = np.zeros((frames, 5380), np.uint8)
        res = np.zeros(frames * 56, np.uint8)
        t0 = time.perf_counter()
        used = lib.orc_decode_batch(O.ptr(pcm_sample), O.FMT_S16, ch, pcm_sample.shape[1], pcm_sample.shape[1] * 2 * ch,
                                    frames, 8, O.ptr(out), O.ptr(res), thr)
        dt = time.perf_counter() - t0
        ok = int((out == payload_ref[:frames]).all(axis=1).sum())
        return {"frames_per_s": frames / dt, "frames": frames, "threads": int(used), "payloads_correct": ok, "wall_s": round(dt, 2)}

    variants = {}
    perf_dir = "/tmp/ofdmrx_oracle_perf_%d" % os.getpid()
    built = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "perf", "PERF_DIR=" + perf_dir],
                           capture_output=True, text=True)
    libs = {"o2_strict": O.lib()}
    if built.returncode == 0:
        for tag in ("o3_native", "ofast_native"):
            L = C.CDLL(os.path.join(perf_dir, "libmodem_oracle_%s.so" % tag.split("_")[0]))
            L.orc_decode_batch.restype = C.c_int
            L.orc_decode_batch.argtypes = O.lib().orc_decode_batch.argtypes
            libs[tag] = L
    # which build is the fastest: a short run each (4 frames per thread; the first, untimed, frame per thread populates the threads'
    # malloc arenas and tables); then the headline: that build on the WHOLE sample - 16 frames per thread by default (round-5
    # verdict, weak 4: with 4 frames per thread the all-core figure was 13 x the one-thread figure on 128 threads)
    n_short = min(n, 4 * threads)
    for tag, L in libs.items():
        run(L, min(n, threads), threads)
        variants[tag] = run(L, n_short, threads)
    full = {k: v for k, v in variants.items() if v["payloads_correct"] == v["frames"]} or dict(variants)
    best = max(full, key=lambda k: full[k]["frames_per_s"])        # the headline is the FASTEST build (round-2 verdict, weak 8)
    v = run(libs[best], n, threads) if n > n_short else full[best]
    variants[best + "_whole_sample"] = v
    one_tag = "o3_native" if "o3_native" in variants else "o2_strict"
    one = run(libs[one_tag], min(n, max(8, int(variants[one_tag]["frames_per_s"] / threads * 4))), 1)
    rates = sorted(x["frames_per_s"] for k, x in variants.items() if not k.endswith("_whole_sample"))
    variants[one_tag + "_1thread"] = one
    desc = {"o2_strict": "gcc -O2, strict IEEE, no contraction (the parity oracle)", "o3_native": "gcc -O3 -march=native, strict IEEE",
            "ofast_native": "gcc -Ofast -march=native (the reference's Makefile:2 flags; speed only)"}
    return {"value": v["frames_per_s"], "unit": "frames/s", "cores": v["threads"], "kind": "port", "host_cores": host_cores,
            "value_1thread": one["frames_per_s"], "frames_per_thread": v["frames"] / float(max(v["threads"], 1)),
            "speedup_over_1thread": v["frames_per_s"] / one["frames_per_s"] if one["frames_per_s"] > 0 else None,
            "sample": "the first %d frames of the headline batch (taken before any other leg touches the input buffer), %.0f frames per thread; "
                      "oracle = C restatement of decode.cc (the reference itself cannot be built: aicodix/dsp + aicodix/code absent), "
                      "list 8; headline = the fastest of the builds timed here (%d frames each: %.0f - %.0f frames/s): %s, re-timed on the "
                      "whole sample on %d OpenMP threads of %d host cores (%d/%d payloads correct, %.1f s); 1 thread (%s): %.2f frames/s"
                      % (n, v["frames"] / float(max(v["threads"], 1)), n_short, rates[0], rates[-1], desc.get(best, best), v["threads"],
                         host_cores, v["payloads_correct"], v["frames"], v["wall_s"], one_tag, one["frames_per_s"]),
            "variants": variants}


def dry_run(args):
    """launcher / sharding check without a GPU: gloo rendezvous, block ranges, counter reduction"""
    from modem_amd import shard
    rank, local_rank, world = shard.env_rank()
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("gloo")
    lo, hi = shard.block_range(args.frames, rank, world) if args.scaling == "strong" else (0, args.frames)
    secs, (frames, ranks) = shard.reduce_counters((1.0, [(hi - lo) * args.steps, 1]), world, dist, None)
    if rank == 0:
        print(json.dumps({"metric": METRIC, "value": frames / secs, "unit": "frames/s", "n_gpus": ranks, "steps": args.steps,
                          "warmup": args.warmup, "scaling": args.scaling, "dry_run": True, "frames": frames // args.steps,
                          "config": {"workload": "dry run (no GPU work)", "parallelism": "frames x%d" % ranks}}), flush=True)
    if dist:
        dist.barrier()
        dist.destroy_process_group()


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))
    from modem_amd import shard
    rank, local_rank, world = shard.env_rank()
    if world != args.gpus:
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%d (run `python bench.py --gpus N`, or torchrun with "
                         "--nproc-per-node == --gpus)\n" % (args.gpus, world))
        sys.exit(2)
    if args.dry_run:
        return dry_run(args)

    import numpy as np
    import torch
    import modem_amd
    import modem_amd.ofdmrx as M

    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    # frames this rank decodes per step, and the global index of its first frame (payload / channel RNG key)
    if args.scaling == "strong":
        lo, hi = shard.block_range(args.frames, rank, world)
        B, first = hi - lo, lo
    else:
        B, first = args.frames, shard.frame_seed_offset(args.frames, rank)
    ch = args.channels
    # ONE explicit (non-default) HIP stream shared by torch and the library: the default stream's handle is 0,
    # which the C ABI reads as "create your own stream", and two streams would race on the device buffers
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    copy_stream = torch.cuda.Stream(device=dev)
    rx = modem_amd.Receiver(device=local_rank, chunk_frames=args.chunk, stream=stream.cuda_stream, sample_rate=args.rate,
                            list_size=args.list)
    spf = rx.tx_frame_samples(args.mode)
    RES = M.RESULT_DTYPE.itemsize
    d_in = torch.empty((max(B, 1), spf, ch), dtype=torch.int16, device=dev)
    d_out = [torch.zeros((max(B, 1), 5380), dtype=torch.uint8, device=dev) for _ in range(2)]
    d_res = [torch.zeros((max(B, 1), RES), dtype=torch.uint8, device=dev) for _ in range(2)]
    h_out = torch.empty((max(B, 1), 5380), dtype=torch.uint8, pin_memory=True)
    h_res = torch.empty((max(B, 1), RES), dtype=torch.uint8, pin_memory=True)
    t_gen = time.perf_counter()
    state = {}

    def generate(rxh, noise_db, impair):
        """fill d_in for this rank: payloads -> device transmitter (N2) -> [channel chain] -> AWGN (N3), or the tiled fixture"""
        if args.unique != 0:
            # synthetic batch made entirely on the device: random payloads -> device transmitter (N2) -> AWGN (N3).
            # Payload RNG and channel RNG are keyed by the GLOBAL frame index, so ranks never repeat each other.
            U = B if args.unique < 0 else min(args.unique, B)
            if "d_pay" not in state:
                g = torch.Generator(device=dev)
                g.manual_seed(args.seed * 1000003 + first)
                state["d_pay"] = torch.randint(0, 256, (U, 5380), dtype=torch.uint8, device=dev, generator=g)
            d_pay = state["d_pay"]
            if U == B:
                d_clean = d_in                        # transmit straight into the batch, add the noise in place
            else:
                d_clean = torch.empty((U, spf, ch), dtype=torch.int16, device=dev)
            rxh.tx_encode(d_pay.data_ptr(), U, d_clean.data_ptr(), mode=args.mode, freq_off=2000, call_sign="ANONYMOUS", channels=ch)
            if ch == 2 and impair:
                d_imp = torch.empty((U, spf, ch), dtype=torch.int16, device=dev)
                taps = [(0, 1 + 0j), (5, 0.35 - 0.1j), (11, -0.2 + 0.2j), (23, 0.1 + 0.05j)]
                for lo_ in range(0, U, 8192):
                    hi_ = min(lo_ + 8192, U)
                    rxh.channel(d_clean[lo_:hi_].data_ptr(), d_imp[lo_:hi_].data_ptr(), hi_ - lo_, spf, cfo_hz=234.567, sfo_ppm=147.0,
                                multipath=taps)
                rxh.awgn_tile(d_imp.data_ptr(), U, d_in.data_ptr(), B, spf, noise_db, args.seed, first)
                rxh.synchronize()
                del d_imp
            elif ch == 2:
                rxh.awgn_tile(d_clean.data_ptr(), U, d_in.data_ptr(), B, spf, noise_db, args.seed, first)
            elif U != B:
                idx = torch.arange(B, device=dev) % U
                for lo_ in range(0, B, 4096):
                    d_in[lo_:lo_ + 4096] = d_clean[idx[lo_:lo_ + 4096]]
            source = "device transmitter, %d distinct random payloads" % U
            if ch == 2 and impair:
                source += ", device channel chain multipath(4 taps) -> CFO +234.567 Hz -> SFO +147 ppm (configs[3])"
            return d_pay, U, source
        fx = np.load(os.path.join(ROOT, "tests", "golden", "base_frames_2ch.npz"))
        base = fx["pcm"]
        n_base = base.shape[0]
        d_pay = torch.from_numpy(fx["payload"]).to(dev)
        d_base = torch.from_numpy(base).to(dev)
        if ch == 2:
            rxh.awgn_tile(d_base.data_ptr(), n_base, d_in.data_ptr(), B, spf, noise_db, args.seed, first)
        else:   # a 1-channel WAV from encode is the real part of the same stream (encode.cc:127-128)
            idx = torch.arange(B, device=dev) % n_base
            for lo_ in range(0, B, 4096):
                d_in[lo_:lo_ + 4096, :, 0] = d_base[idx[lo_:lo_ + 4096], :, 0]
        return d_pay, n_base, "%d committed fixture frames (tests/golden/base_frames_2ch.npz) tiled" % n_base

    d_pay, n_base, source = generate(rx, args.noise_db, args.impair)
    rx.synchronize()
    torch.cuda.synchronize()
    gen_s = time.perf_counter() - t_gen
    # cpu_baseline's sample: the first frames of THIS batch, taken now - the extra legs below regenerate d_in (round-5 verdict, weak 4)
    cpu_sample = cpu_ref = None
    if rank == 0 and world == 1 and args.cpu_frames != 0 and B:
        ncpu = min(CPU_FRAMES_DEFAULT if args.cpu_frames < 0 else args.cpu_frames, B)
        cpu_sample = d_in[:ncpu].cpu().numpy()
        cpu_ref = d_pay[torch.arange(ncpu, device=dev) % n_base].cpu().numpy()

    def step(s, to_host):
        """one pass of the hot path over the batch.  to_host: the payload bytes + result records go to pinned host memory (SURVEY
        8d's endpoint) - the device entry takes pinned host pointers for its outputs and copies every chunk out right behind it,
        beside the next chunk's kernels (include/ofdmrx.h, revision 1.4); else they stay in HBM (d_out / d_res)"""
        q = s & 1
        if B == 0:
            return
        if to_host:
            rx.decode_device(d_in.data_ptr(), M.FMT_S16, ch, spf, spf * 2 * ch, B, h_out.data_ptr(), h_res.data_ptr())
        else:
            rx.decode_device(d_in.data_ptr(), M.FMT_S16, ch, spf, spf * 2 * ch, B, d_out[q].data_ptr(), d_res[q].data_ptr())

    def timing_of(r):
        """ofdmrx_get_timing's stages plus the list-1 pass (ofdmrx_get_sc_timing: k_sc + k_sc_finish)"""
        t = dict(r.timing())
        t["sc"] = r.sc_timing()
        return t

    def routes(r, n):
        """who finished the frames of the handle's last call: the syndrome certificate in k_back (incl. frames without a header,
        which end there too), the list-1 pass, the list decoder"""
        ld, sc = r.list_decoded_frames(), r.sc_decided_frames()
        return {"certified": n - max(ld, 0) - max(sc, 0) if ld >= 0 else 0, "sc_decided": max(sc, 0), "list_decoded": ld if ld >= 0 else n}

    def fence():
        torch.cuda.synchronize()
        if dist:
            dist.barrier()
        torch.cuda.synchronize()

    for s in range(args.warmup):
        step(s, True)
    fence()
    # ---- timed region 1 (the headline `value`): samples resident in HBM -> payload bytes resident on the host
    stage_ms, stage_launches = {}, {}
    t0 = time.perf_counter()
    for s in range(args.steps):
        step(s, True)
        if B:
            # hipEvents on the streams the kernels run on; synchronises the handle's stream.  (Measured at the end of round 4: with the
            # steps enqueued back to back instead - no host synchronisation inside the timed region - a step takes 38.8 ms, not 36.9:
            # profiles/r04_v23_step_sync_and_deferred_join_ab.txt)
            t = timing_of(rx)
            for k, v in t.items():
                stage_ms[k] = stage_ms.get(k, 0.0) + v[0]
                stage_launches[k] = stage_launches.get(k, 0) + v[1]
    fence()
    secs = time.perf_counter() - t0
    last = (args.steps - 1) & 1
    list_decoded = rx.list_decoded_frames() if B else 0      # of the last step; -1: no certificate on this handle
    sc_decided = rx.sc_decided_frames() if B else 0          # ... finished by the list-1 pass; -1: that pass is off
    # one extra, untimed step with every kernel alone on the device (OFDMRX_NO_OVERLAP is read per call): under the pipeline a
    # stage's event span includes the time it shares the machine with others; the kernel that takes the most time ALONE is
    # the one the roofline object describes
    alone_ms = {}
    if B:
        prev = os.environ.get("OFDMRX_NO_OVERLAP")
        os.environ["OFDMRX_NO_OVERLAP"] = "1"
        step(0, False)
        alone_ms = {k: v[0] / max(v[1], 1) for k, v in timing_of(rx).items()}
        if prev is None:
            del os.environ["OFDMRX_NO_OVERLAP"]
        else:
            os.environ["OFDMRX_NO_OVERLAP"] = prev
    # ---- timed region 2: the kernels alone (payloads stay in HBM)
    t0 = time.perf_counter()
    for s in range(args.steps):
        step(s, False)
    fence()
    secs_k = time.perf_counter() - t0

    # ---- error counters from the HOST copy of the last timed step (that is what a consumer would see)
    res = h_res.numpy().view(M.RESULT_DTYPE).reshape(-1)[:B]
    frame_err = bit_err = 0
    pop = torch.tensor([bin(i).count("1") for i in range(256)], dtype=torch.int64, device=dev)
    d_last = h_out.to(dev, non_blocking=False)
    for lo_ in range(0, B, 8192):
        ref = d_pay[torch.arange(lo_, min(lo_ + 8192, B), device=dev) % n_base]
        be = pop[(d_last[lo_:lo_ + 8192] ^ ref).long()].sum(dim=1)
        frame_err += int((be > 0).sum().item())
        bit_err += int(be.sum().item())
    assert B == 0 or bool((d_last == d_out[last][:B]).all().item()), "host copy differs from the device buffer"
    ok_status = int((res["status"] == 0).sum())

    # ---- host-pointer leg: pinned host samples in -> payload bytes on the host (ofdmrx_decode_batch), bounded sample
    host_fps = None
    nh = args.host_frames
    if nh < 0:
        nh = min(B, 16384)
    nh = min(nh, B)
    if nh > 0 and ch == 2:
        h_in = torch.empty((nh, spf, ch), dtype=torch.int16, pin_memory=True)
        h_in.copy_(d_in[:nh])
        torch.cuda.synchronize()
        a_in = h_in.numpy()
        for it in range(2):                              # first call allocates the staging buffers
            th = time.perf_counter()
            o_h, r_h = rx.decode(a_in)
            dt_h = time.perf_counter() - th
        host_fps = nh / dt_h
        assert (o_h == h_out.numpy()[:nh]).all(), "host-pointer path differs from the device path"
        del h_in

    # ---- timed region 3: the same batch with the list decoder forced for every frame (cfg.flags bit 1): what the path costs
    # (the first handle is closed before: the second one's 8 GB of level stores then land where the first one's buffers were -
    # allocated behind them the list decoder measured 4 % slower)
    # when the syndrome certificate decides nothing (it decides every frame at this noise level, none from -24 dB on)
    chunk_frames_used = rx.chunk_frames
    rx.close()
    rx = None
    scl = None
    scl_steps = min(args.steps, 5) if args.scl_steps < 0 else args.scl_steps
    if scl_steps > 0 and not os.environ.get("OFDMRX_NO_CERT"):     # (the same decision on every rank: barriers inside)
        sm2, sl2 = {}, {}
        same = True
        if B:
            rx2 = modem_amd.Receiver(device=local_rank, chunk_frames=args.chunk, stream=stream.cuda_stream, sample_rate=args.rate,
                                     list_size=args.list, scl_always=True)
            d_out2 = torch.zeros((B, 5380), dtype=torch.uint8, device=dev)
            d_res2 = torch.zeros((B, RES), dtype=torch.uint8, device=dev)
            for _ in range(2):       # first call allocates, second warms the pipeline
                rx2.decode_device(d_in.data_ptr(), M.FMT_S16, ch, spf, spf * 2 * ch, B, d_out2.data_ptr(), d_res2.data_ptr())
        fence()
        t0 = time.perf_counter()
        for s in range(scl_steps):
            if B:
                rx2.decode_device(d_in.data_ptr(), M.FMT_S16, ch, spf, spf * 2 * ch, B, d_out2.data_ptr(), d_res2.data_ptr())
                t = timing_of(rx2)
                for k, v in t.items():
                    sm2[k] = sm2.get(k, 0.0) + v[0]
                    sl2[k] = sl2.get(k, 0) + v[1]
        fence()
        secs2 = time.perf_counter() - t0
        if B:
            same = bool((d_out2 == d_out[last][:B]).all().item()) and bool((d_res2 == d_res[last][:B]).all().item())
            rx2.close()
            del d_out2, d_res2
        scl = {"secs": secs2, "steps": scl_steps, "stage_ms": sm2, "stage_launches": sl2, "identical": same}

    # ---- two bounded legs outside `value` (one GPU, the headline workload only): what the README's own channel chain and a noise
    # level where the certificate decides nothing cost through the DEFAULT handle - configs[3] and one level of configs[4]
    extra = {}
    if world == 1 and B and ch == 2 and args.unique != 0 and not args.impair and args.leg_steps > 0:
        rx3 = modem_amd.Receiver(device=local_rank, chunk_frames=args.chunk, stream=stream.cuda_stream, sample_rate=args.rate,
                                 list_size=args.list)
        pop8 = torch.tensor([bin(i).count("1") for i in range(256)], dtype=torch.int64, device=dev)
        for name, db, imp, lch, what in (("value_config3", -30.0, True, 2, "configs[3] (README.md:49): multipath -> CFO +234.567 Hz -> SFO +147 ppm -> AWGN -30 dB"),
                                         ("value_noise_m20", -20.0, False, 2, "a configs[4] level where every frame has raw bit errors (the syndrome certificate decides nothing): AWGN noise level -20 dB"),
                                         ("value_config1_mono", None, False, 1, "configs[1] flavour: clean 16-bit MONO frames (the consumers form the analytic signal, DESIGN.md 4h)")):
            if lch == 2:
                generate(rx3, db, imp)
                d_leg = d_in
            else:                                                     # the same payloads as a real (1-channel) stream, encode.cc:127-128
                d_leg = torch.empty((B, spf, 1), dtype=torch.int16, device=dev)
                rx3.tx_encode(d_pay.data_ptr(), B, d_leg.data_ptr(), mode=args.mode, freq_off=2000, call_sign="ANONYMOUS", channels=1)
            rx3.synchronize()
            for _ in range(2):        # first call allocates, second warms the pipeline
                rx3.decode_device(d_leg.data_ptr(), M.FMT_S16, lch, spf, spf * 2 * lch, B, d_out[0].data_ptr(), d_res[0].data_ptr())
            fence()
            lsteps = args.leg_steps if lch == 2 else min(args.leg_steps, 2)
            lsm, lsl = {}, {}
            t0 = time.perf_counter()
            for _ in range(lsteps):
                rx3.decode_device(d_leg.data_ptr(), M.FMT_S16, lch, spf, spf * 2 * lch, B, d_out[0].data_ptr(), d_res[0].data_ptr())
                for k, v in timing_of(rx3).items():
                    lsm[k] = lsm.get(k, 0.0) + v[0]
                    lsl[k] = lsl.get(k, 0) + v[1]
            fence()
            dt = time.perf_counter() - t0
            ferr = 0
            for lo_ in range(0, B, 8192):
                ref = d_pay[torch.arange(lo_, min(lo_ + 8192, B), device=dev) % n_base]
                ferr += int((pop8[(d_out[0][lo_:lo_ + 8192] ^ ref).long()].sum(dim=1) > 0).sum().item())
            leg_listed, leg_routes = rx3.list_decoded_frames(), routes(rx3, B)     # (of the last full-size call)
            # the same workload as ONE call of 8192 frames: a GPU's share of this batch at N = 8 under strong scaling (configs[3])
            share = None
            if B > 8192:
                for _ in range(2):
                    rx3.decode_device(d_leg.data_ptr(), M.FMT_S16, lch, spf, spf * 2 * lch, 8192, d_out[1].data_ptr(), d_res[1].data_ptr())
                fence()
                t0 = time.perf_counter()
                for _ in range(4):
                    rx3.decode_device(d_leg.data_ptr(), M.FMT_S16, lch, spf, spf * 2 * lch, 8192, d_out[1].data_ptr(), d_res[1].data_ptr())
                fence()
                share = 8192 * 4 / (time.perf_counter() - t0)
            # the same batch through a handle with OFDMRX_FLAG_TWO_LANES (include/ofdmrx.h revision 1.6: the call's second half runs
            # through a second pipeline beside the first); opt-in, so outside the leg's `value`
            lanes2 = None
            if lch == 2 and B >= 4 * rx3.chunk_frames:
                rx4 = modem_amd.Receiver(device=local_rank, chunk_frames=args.chunk, stream=stream.cuda_stream, sample_rate=args.rate,
                                         list_size=args.list, two_lanes=True)
                for _ in range(2):
                    rx4.decode_device(d_leg.data_ptr(), M.FMT_S16, lch, spf, spf * 2 * lch, B, d_out[1].data_ptr(), d_res[1].data_ptr())
                fence()
                t0 = time.perf_counter()
                for _ in range(3):
                    rx4.decode_device(d_leg.data_ptr(), M.FMT_S16, lch, spf, spf * 2 * lch, B, d_out[1].data_ptr(), d_res[1].data_ptr())
                fence()
                lanes2 = {"value": B * 3 / (time.perf_counter() - t0), "steps": 3,
                          "outputs_identical_to_one_lane": bool((d_out[1] == d_out[0]).all().item()) and bool((d_res[1] == d_res[0]).all().item())}
                rx4.close()
            extra[name] = {"value": B * lsteps / dt, "unit": "frames/s", "steps": lsteps, "frames": B, "workload": what,
                           "value_one_call_of_8192_frames": share, "value_two_lanes": lanes2,
                           "list_decoded_frames": leg_listed, "routes": leg_routes, "fer": ferr / float(B),
                           "definition": "default handle, payloads left in HBM, outside `value`",
                           "stage_ms_per_step": {k: v / lsteps for k, v in lsm.items()}, "_stage": (lsm, lsl, lsteps, lch)}
            del d_leg
        rx3.close()

    secs_max, (frames_total, frame_err, bit_err, ok_status, ranks, frames_step) = shard.reduce_counters(
        (secs, [B * args.steps, frame_err, bit_err, ok_status, 1, B]), world, dist, dev)
    secs_k_max, _ = shard.reduce_counters((secs_k, [0]), world, dist, dev)
    # which rank decoded how many frames, and what the process group says about itself (RCCL, the world size it sees)
    frames_by_rank, dist_info = [B], {"backend": None, "world_size": 1}
    if dist:
        tb_ = torch.zeros(world, dtype=torch.int64, device=dev)
        tb_[rank] = B
        dist.all_reduce(tb_, op=dist.ReduceOp.SUM)
        frames_by_rank = [int(x) for x in tb_.tolist()]
        dist_info = {"backend": dist.get_backend(), "world_size": dist.get_world_size()}

    secs2_max = None
    if scl is not None:
        secs2_max, _ = shard.reduce_counters((scl["secs"], [0]), world, dist, dev)

    if rank == 0:
        value = frames_total / secs_max
        b_frame = spf * 2 * ch + 5380          # compulsory input + output bytes per frame (386180 for the headline)
        try:
            with open(os.path.join(ROOT, "profiles", TRAFFIC_FILE)) as fh:
                traffic_db = json.load(fh)
        except (OSError, ValueError):
            traffic_db = {}

        def roofline(sm, sl, steps, alone=None, fps=None, bf=None, dbkey="kernels"):
            """the roofline object of the stage that takes the most time in a run: achieved = ALGORITHMIC bytes of the whole path
            per launch / the kernel's average launch duration (hipEvents on its stream); traffic = HBM bytes per launch from the
            committed PMC passes (bench.py cannot run the profiler on itself), scaled to this run's frames per launch.
            fps: the run's frames/s (whole_path_frac, SURVEY 8d's own definition); bf: algorithmic bytes per frame when they are not
            the headline's (mono input); dbkey: which table of the traffic file holds the run's kernels"""
            bf = bf or b_frame
            cand = {k: v for k, v in sm.items() if k in STAGE_KERNELS and sl.get(k)}
            if not cand:
                return None
            pick = {k: alone[k] for k in cand if alone and k in alone} or cand
            st = max(pick, key=pick.get)
            kern, src = STAGE_KERNELS[st]
            launches = sl[st]
            fpl = B * steps / launches
            avg_s = sm[st] / 1e3 / launches
            ach = bf * fpl / avg_s / 1e9 if avg_s > 0 else 0.0
            tj = traffic_db.get(dbkey, {}).get(st)
            traffic = stale = tsrc = None
            if tj:
                traffic = (tj["fetch_KiB"] * traffic_db.get("fetch_scale", 2.0) + tj["write_KiB"] * traffic_db.get("write_scale", 1.0)) \
                    * 1024.0 * fpl / traffic_db["frames_per_launch"]
                stale = tj.get("src_sha") != file_sha(os.path.join(ROOT, "modem_amd", "csrc", src))
                tsrc = ("profiles/%s: rocprofv3 --pmc FETCH_SIZE x %.2f + WRITE_SIZE x %.2f (separate passes, one %d-frame chunk, kernels "
                        "back to back; scales calibrated in the same session on tools/pmc_calib.hip), scaled to this run's frames per launch"
                        % (TRAFFIC_FILE, traffic_db.get("fetch_scale", 2.0), traffic_db.get("write_scale", 1.0), traffic_db["frames_per_launch"]))
            # which resource the kernel is closer to: HBM (its real traffic against the peak) or vector-instruction issue (the
            # committed SQ_INSTS_VALU pass: wave instructions per launch on 1024 SIMDs against the launch time, priced at the guide's
            # 2 cycles each and at the measured ceiling of tools/ubench_issue.hip)
            valu = None
            if tj and tj.get("valu_insts") and avg_s > 0:
                insts = tj["valu_insts"] * fpl / traffic_db["frames_per_launch"]
                per_simd = insts / N_SIMD
                ns_meas = traffic_db.get("valu_ns_per_inst_per_simd")      # tools/ubench_issue.hip on the box of the traffic passes
                valu = {"insts_per_launch": insts, "simds": N_SIMD,
                        "frac_of_guide_peak": per_simd * VALU_CYCLES_GUIDE / (SCLK_GHZ * 1e9 * avg_s),
                        "guide_peak": "%.0f cycles per wave64 VALU instruction per SIMD at the nominal %.1f GHz (MI355X_MICROARCH.md)" % (VALU_CYCLES_GUIDE, SCLK_GHZ),
                        "frac_of_measured_ceiling": (per_simd * ns_meas * 1e-9 / avg_s) if ns_meas else None,
                        "measured_ceiling": ("%.3f ns per VALU wave-instruction per SIMD, every CU busy, the best of 2 - 5 waves per SIMD (profiles/%s; a measured "
                                             "time: no clock assumed; shader clock under that load %s GHz)"
                                             % (ns_meas, ISSUE_FILE, traffic_db.get("sclk_GHz_measured", "n/a"))) if ns_meas else None,
                        "source": "profiles/%s: rocprofv3 --pmc SQ_INSTS_VALU, scaled to this run's frames per launch" % TRAFFIC_FILE}
                valu["achieved_frac"] = valu["frac_of_measured_ceiling"] if ns_meas else valu["frac_of_guide_peak"]
            hbm_frac_real = (traffic / avg_s / 1e9 / HBM_PEAK_GBS) if (traffic and avg_s > 0) else None
            bound = "valu" if (valu and hbm_frac_real is not None and valu["achieved_frac"] > hbm_frac_real) else "hbm"
            return {"bound": bound, "kernel": kern, "stage": st, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": ach / HBM_PEAK_GBS,
                    "whole_path_frac": (fps * bf / 1e9 / HBM_PEAK_GBS) if fps else None,
                    "whole_path_frac_definition": "the run's frames/s x algorithmic bytes per frame / the HBM peak (SURVEY 8d): every stage's time counts, "
                                                  "not the dominant kernel's alone",
                    "valu_issue": valu,
                    "bound_definition": "the larger of traffic_GBps / peak (HBM) and valu_issue.achieved_frac (vector instruction issue against "
                                        "the MEASURED ceiling; frac_of_guide_peak prices the same count at the guide's 2 cycles); "
                                        "achieved / peak / frac are the HBM figures on ALGORITHMIC bytes as the contract defines them",
                    "achieved_definition": "ALGORITHMIC bytes of the whole path (B_frame x frames per launch) / average launch duration of "
                                           "the run's dominant kernel (hipEvents on the launch stream); its REAL HBM rate is traffic_GBps",
                    "traffic": traffic, "traffic_source": tsrc, "traffic_stale": stale,
                    "traffic_GBps": (traffic / avg_s / 1e9) if (traffic and avg_s > 0) else None,
                    "frames_per_launch": fpl, "avg_launch_ms": 1e3 * avg_s,
                    "avg_launch_ms_alone": alone.get(st) if alone else None, "algorithmic_bytes_per_frame": bf}

        for leg in extra.values():                                # the roofline object of each leg's dominant kernel (k_sc where the list-1 pass decides)
            lsm, lsl, lsteps, lch = leg.pop("_stage")
            leg["roofline"] = roofline(lsm, lsl, lsteps, fps=leg["value"], bf=spf * 2 * lch + 5380, dbkey="kernels" if lch == 2 else "kernels_mono")
        cert_note = ""
        if list_decoded >= 0:
            cert_note = ("; of rank 0's %d frames in the last step the syndrome certificate (hard decisions already a codeword with a valid "
                         "CRC-32 => the list decoder's answer is known, DESIGN.md 4g) decided %d, the list-1 pass (the sign-following path "
                         "provably is the list decoder's lane 0, DESIGN.md 4i) %d, the list decoder the other %d - value_scl_forced is the "
                         "same batch with the list decoder run for every frame"
                         % (B, B - list_decoded - max(sc_decided, 0), max(sc_decided, 0), list_decoded))
        line = {
            "metric": METRIC,
            "value": value, "unit": "frames/s", "n_gpus": ranks, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * secs_max / args.steps, "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "value_definition": "samples resident in HBM -> payload bytes + result records resident in pinned host memory (the device "
                                "entry with pinned host output pointers: every chunk is copied out behind its flush, beside the next "
                                "chunk's kernels); the library's default behaviour",
            "value_kernel_only": frames_total / secs_k_max,
            "value_scl_forced": (frames_step * scl["steps"] / secs2_max) if (scl and secs2_max) else None,
            "value_scl_forced_definition": ("the same batch, OFDMRX_FLAG_SCL_ALWAYS: polar SCL for every frame (what the reference does), "
                                            "%d steps, payloads left in HBM; outputs identical to the default path: %s"
                                            % (scl["steps"], scl["identical"])) if scl else None,
            "list_decoded_frames_rank0": list_decoded,
            "routes_rank0": {"certified": B - max(list_decoded, 0) - max(sc_decided, 0) if list_decoded >= 0 else 0,
                             "sc_decided": max(sc_decided, 0), "list_decoded": list_decoded if list_decoded >= 0 else B},
            "value_host": host_fps,
            "value_host_definition": ("ofdmrx_decode_batch: %d frames from PINNED host memory -> payload bytes on the host, "
                                      "PCIe both ways inside the time, rank 0 only" % nh) if host_fps else None,
            "config": {"workload": ("configs[%d]%s: batch %d analytic (2-ch int16) mode-%d %g kHz frames %s, AWGN noise "
                                    "level %g dB (a noise LEVEL: about +20 dB SNR), inputs resident in HBM; %s, on-device noise keyed "
                                    "by frame index%s"
                                    % (3 if args.impair else 2, "" if (args.rate == 8000 and args.mode == 6 and args.list == 8) else " variant (not the headline workload)",
                                       args.frames, args.mode, args.rate / 1000.0, "per GPU" if args.scaling == "weak" else "in total, sharded",
                                       args.noise_db, source, cert_note)) if ch == 2 else
                                   ("configs[1] flavour: batch %d clean 16-bit mono mode-6 8 kHz frames per GPU, inputs resident "
                                    "in HBM; %s%s" % (B, source, cert_note)),
                       "frames_per_step": frames_step, "frames_rank0": B, "frames_by_rank": frames_by_rank, "process_group": dist_info, "list_size": args.list, "chunk_frames": chunk_frames_used,
                       "parallelism": "frames x%d" % ranks},
            "ber": bit_err / (43040.0 * max(frames_step, 1)), "fer": frame_err / float(max(frames_step, 1)),
            "frames_ok": ok_status, "frames": frames_step,
            "roofline": roofline(stage_ms, stage_launches, args.steps, alone_ms, fps=value / max(ranks, 1), dbkey="kernels" if ch == 2 else "kernels_mono"),
            "roofline_scl_forced": roofline(scl["stage_ms"], scl["stage_launches"], scl["steps"],
                                            fps=frames_step * scl["steps"] / secs2_max / max(ranks, 1) if secs2_max else None) if scl else None,
            "stage_ms_per_step": {k: v / args.steps for k, v in stage_ms.items()},
            "stage_ms_per_launch_alone": alone_ms,
            "stage_ms_per_step_scl_forced": {k: v / scl["steps"] for k, v in scl["stage_ms"].items()} if scl else None,
            "input_generation_s": gen_s,
        }
        line.update(extra)
        if ranks == 1 and cpu_sample is not None:
            line["cpu_baseline"] = cpu_baseline(cpu_sample, cpu_ref, ch)
        print(json.dumps(line), flush=True)
    if rx is not None:
        rx.close()
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
