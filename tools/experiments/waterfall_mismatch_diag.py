import sys, os
ROOT = "/root/repo" if os.path.isdir("/root/repo") else os.getcwd()
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import oracle_lib as O, modem_amd, modem_amd.ofdmrx as M
n = 32768
dev = torch.device("cuda:0")
rx = modem_amd.Receiver(device=0, chunk_frames=96)
spf = rx.tx_frame_samples(6)
for li, db in enumerate([-14.5, -15.0]):
    g = torch.Generator(device=dev); g.manual_seed(1234 + li)
    d_pay = torch.randint(0, 256, (n, 5380), dtype=torch.uint8, device=dev, generator=g)
    d_in = torch.empty((n, spf, 2), dtype=torch.int16, device=dev)
    rx.tx_encode(d_pay.data_ptr(), n, d_in.data_ptr(), mode=6)
    rx.awgn_tile(d_in.data_ptr(), n, d_in.data_ptr(), n, spf, db, 99, li * n)
    rx.synchronize()
    d_out = torch.zeros((n, 5380), dtype=torch.uint8, device=dev)
    d_res = torch.zeros((n, M.RESULT_DTYPE.itemsize), dtype=torch.uint8, device=dev)
    rx.decode_device(d_in.data_ptr(), M.FMT_S16, 2, spf, spf * 4, n, d_out.data_ptr(), d_res.data_ptr())
    rx.synchronize()
    out = d_out.cpu().numpy(); res = d_res.cpu().numpy().view(M.RESULT_DTYPE).reshape(-1)
    pcm = np.ascontiguousarray(d_in.cpu().numpy())
    oout = np.zeros((n, 5380), np.uint8); ores = np.zeros(n * 56, np.uint8)
    O.lib().orc_decode_batch(O.ptr(pcm), O.FMT_S16, 2, spf, spf * 4, n, 8, O.ptr(oout), O.ptr(ores), 128)
    ores = ores.view(M.RESULT_DTYPE).reshape(-1)
    pay = d_pay.cpu().numpy()
    names = ("status", "best_lane", "sc_start", "symbol_pos", "oper_mode", "call_sign", "n_sync_rejects")
    bad = [i for i in range(n) if not (out[i] == oout[i]).all() or any(res[nm][i] != ores[nm][i] for nm in names[:6])]
    print(db, "frames that differ in something decided:", bad, flush=True)
    for i in bad:
        print("  frame", i, "payload equal:", bool((out[i] == oout[i]).all()), "gpu == sent:", bool((out[i] == pay[i]).all()), "oracle == sent:", bool((oout[i] == pay[i]).all()))
        for nm in names + ("cfo_rad", "cfo_fine", "bit_flips", "esn0_db_last"):
            print("     %-14s gpu %r oracle %r" % (nm, res[nm][i], ores[nm][i]))
