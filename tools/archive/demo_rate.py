#!/usr/bin/env python3
"""The host mirror end to end, as the plugin runs it: dab_host_demo (OFDM_Demod::Process on the main thread, the ring,
BasicRadio::Process on the radio thread, FIG database, DAB+ channels) over a synthetic stream of N frames from a file in
the page cache; wall clock per frame.  usage: tools/demo_rate.py [n_frames] [chunk_samples]"""
import os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sdrplusplus-dab-radio-plugin_amd")); sys.path.insert(0, ROOT)
import numpy as np
from dabgpu import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
chunk = sys.argv[2] if len(sys.argv) > 2 else "65536"
SERVICES = [("Radio One", 0xC221, 3, 0, 3, 64, 0), ("Jazz 24", 0xC222, 7, 0, 2, 48, 48), ("News", 0xC223, 9, 1, 2, 32, 200)]
ens = synth.ServiceEnsemble(1, SERVICES, n_frames=5)
base = synth.channel(np.tile(ens.iq().ravel(), 4), snr_db=20.0, cfo=1.2 / 2048, rng=np.random.default_rng(8)).astype(np.complex64)
L = synth.NB_FRAME_SAMPLES
host = os.path.join(ROOT, "sdrplusplus-dab-radio-plugin_amd", "host")
exe = os.path.join(host, "dab_host_demo")
if not os.path.exists(exe):
    subprocess.check_call(["make", "-C", host, "-j4"], stdout=subprocess.DEVNULL)
with tempfile.TemporaryDirectory() as d:
    path = os.path.join(d, "iq.cf32")
    with open(path, "wb") as f:
        f.write(base[-30000:].tobytes())
        for k in range((n + 19) // 20):
            f.write(base.tobytes())                       # 20 frames per repeat: the multiplex continues seamlessly
        f.write(base[:synth.NB_NULL + 5000].tobytes())
    frames = 20 * ((n + 19) // 20)
    for rep in range(2):
        r = subprocess.run([exe, path, os.path.join(d, "out"), chunk], stdout=subprocess.PIPE, text=True, timeout=600,
                           env=dict(os.environ, DAB_DEMO_PRELOAD="1"))
        assert r.returncode == 0, r.stderr
    if os.environ.get("DEMO_RATE_TRACE"):
        # kernel durations of the same run (rocprofv3 starts the demo itself; this process never touches the GPU)
        td = os.environ["DEMO_RATE_TRACE"]
        subprocess.run(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", td, "-o", "demo", "--", exe, path,
                        os.path.join(d, "out"), chunk], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=900,
                       env=dict(os.environ, DAB_DEMO_PRELOAD="1", TMPDIR="/tmp"), cwd="/tmp")
    line = r.stdout.strip().splitlines()[-1]
    read = int(line.split("frames_read=")[1].split()[0])
    dt = float(r.stdout.split("processing_s=")[1].split()[0])
    print(line)
    print(r.stdout.strip().splitlines()[-2])
    print("dab_host_demo: %d frames (three DAB+ services opened from the FIC), samples in memory, chunks of %s: %.3f s in the two threads = "
          "%.0f frames/s = %.0f x real time, %.1f us per frame" % (read, chunk, dt, read / dt, read / dt * 0.096, dt / read * 1e6))
