#!/usr/bin/env python3
"""GPU box: the gated decision-directed loop against the loop on the cyclic-prefix correlations, scenario by scenario
(tests/test_loop_gate.py's channels, several noise seeds): worst |offset - truth| after settling, calls gated, frames
decoded to the transmitted FIBs.  gate 2.5 = the default; 8 = the first guess of round 4; 0 = the quality gate off (round 3's behaviour, but for the
empty-sum guard and the branch hold).
usage: tools/loop_gate_table.py [seeds]"""
import os, sys, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sdrplusplus-dab-radio-plugin_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, dabgpu
from dabgpu import synth
SC = {"awgn_8dB": (8.0, 0.0, None), "awgn_5dB": (5.0, 0.0, None), "awgn_3dB": (3.0, 0.0, None), "awgn_0dB": (0.0, 0.0, None),
      "rayleigh_25Hz": (15.0, 25.0, None), "rayleigh_60Hz": (15.0, 60.0, None), "cw_minus10dBc": (15.0, 0.0, -10.0)}
USED, CFO, N = 76 * 2552, 0.07, 16
seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
ens = synth.Ensemble(seed=0xDAB0, n_frames=5); eiq = ens.iq()
dev = torch.device("cuda", 0)
print("scenario        /call  loop      worst |offset - truth| (carriers)   gated calls   frames decoded of %d x %d seeds" % (N, seeds))
for name, (snr, fading, cw) in SC.items():
    for per_call in (1, 4):
        rows = {"prefix": [0.0, 0, 0], "dd gate 2.5": [0.0, 0, 0], "dd gate 8": [0.0, 0, 0], "dd gate 0": [0.0, 0, 0]}
        for seed in range(seeds):
            rng = np.random.default_rng(zlib.crc32(name.encode()) + seed)
            tx = np.tile(eiq, (4, 1))[:N].ravel()
            if cw is not None:
                tx = tx + np.sqrt(10 ** (cw / 10)) * np.exp(2j * np.pi * (37.3 / 2048.0) * np.arange(tx.size) + 1.0j)
            rx = synth.channel(tx, snr_db=snr, cfo=CFO / 2048.0, rng=rng, fading_hz=fading, rice_k=0.0).reshape(N, -1)
            frames = np.ascontiguousarray(rx[:, synth.NB_NULL - 16:synth.NB_NULL - 16 + USED])
            d_iq = torch.from_numpy(frames).to(dev)
            soft = torch.zeros((per_call, dabgpu.NB_FRAME_BITS), dtype=torch.int8, device=dev)
            cyc = torch.zeros((per_call, 76), dtype=torch.complex64, device=dev)
            for label in rows:
                c = dabgpu.Context(0, 8); c.streams_reset(1)
                if label != "prefix":
                    c.set_stream_loop(decision_directed=True); c.set_loop_gate(float(label.split()[-1]))
                dev_off, ok_n = [], 0
                for k in range(N // per_call):
                    torch.cuda.synchronize()
                    c.ofdm_demod_streams_dev(d_iq.data_ptr() + k * per_call * USED * 8, USED, 1, per_call, 0.9, soft.data_ptr(),
                                             cyc.data_ptr() if label == "prefix" else None, None)
                    c.sync()
                    st = c.get_stats(0)
                    dev_off.append(abs(st.fine_freq_offset * 2048 + CFO))
                    fib, ok = c.fic_decode(soft.cpu().numpy())
                    for f in range(per_call):
                        ok_n += bool(ok[f].all()) and bool((fib[f] == ens.fibs[(k * per_call + f) % 5]).all())
                settle = 3 if per_call == 1 else 1
                rows[label][0] = max(rows[label][0], max(dev_off[settle:]))
                rows[label][1] += st.loop_gated; rows[label][2] += ok_n
                c.close()
        for label, (w, g, okn) in rows.items():
            print("%-15s %d      %-11s %.4f                               %4d          %4d" % (name, per_call, label, w, g, okn), flush=True)
