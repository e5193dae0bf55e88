#!/usr/bin/env python3
"""GPU box: the one overlap variant tools/overlap_time.py did not measure -- the channel decoder on a CU-MASKED stream
(hipExtStreamCreateWithCUMask) beside the front end of the next batch on the remaining CUs.  Two-stream overlap without
masks loses (decoder workgroups that share a CU with the front end take the LDS and registers its third
workgroup needs); with disjoint CU sets the two kernels cannot take each other's CU resources, only HBM bandwidth.
Pipeline: step k demodulates batch k into soft[k & 1] on stream A while stream B decodes soft[(k - 1) & 1].
usage: tools/overlap_cumask.py [ensembles] [frames]"""
import ctypes as C
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sdrplusplus-dab-radio-plugin_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch, dabgpu
E = int(sys.argv[1]) if len(sys.argv) > 1 else 64
F = int(sys.argv[2]) if len(sys.argv) > 2 else 256
n = E * F
dev = torch.device("cuda", 0)
torch.cuda.init()
hip = C.CDLL("libamdhip64.so")
hip.hipExtStreamCreateWithCUMask.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_uint32)]
hip.hipStreamDestroy.argtypes = [C.c_void_p]
hip.hipStreamSynchronize.argtypes = [C.c_void_p]
hip.hipEventCreate.argtypes = [C.POINTER(C.c_void_p)]
hip.hipEventRecord.argtypes = [C.c_void_p, C.c_void_p]
hip.hipStreamWaitEvent.argtypes = [C.c_void_p, C.c_void_p, C.c_uint]
NCU = torch.cuda.get_device_properties(0).multi_processor_count


def masked_stream(bits):
    """stream restricted to the CUs whose bit is set in `bits` (a Python int, bit i = CU i of the mask order)"""
    words = (NCU + 31) // 32
    arr = (C.c_uint32 * words)(*[(bits >> (32 * w)) & 0xFFFFFFFF for w in range(words)])
    s = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), words, arr)
    assert rc == 0, "hipExtStreamCreateWithCUMask -> %d" % rc
    return s


def event():
    e = C.c_void_p()
    assert hip.hipEventCreate(C.byref(e)) == 0
    return e


L = dabgpu.NB_FRAME_SAMPLES
iq = torch.empty((E, F * L, 2), dtype=torch.float32, device=dev).normal_()
soft = [torch.empty((n, dabgpu.NB_FRAME_BITS), dtype=torch.int8, device=dev) for _ in range(2)]
cyc = torch.zeros((n, 76, 2), dtype=torch.float32, device=dev)
fib = torch.zeros((n, 12, 32), dtype=torch.uint8, device=dev); ok = torch.zeros((n, 12), dtype=torch.uint8, device=dev)
sc = dabgpu.subchannel(0, 64, level=3)
msc = torch.zeros((E, F * 4, 192), dtype=torch.uint8, device=dev)
hist = [torch.zeros((E, 15, sc.length * 64), dtype=torch.int8, device=dev) for _ in range(2)]
torch.cuda.synchronize()


def run(dec_cus, fe_all, reps=8, serial=False):
    """dec_cus CUs for the decoder's stream; the front end gets the others (or every CU when fe_all)."""
    full = (1 << NCU) - 1
    dec_bits = 0
    # spread the decoder's CUs evenly over the mask order (every XCD / shader engine gives up the same share)
    if dec_cus:
        step = NCU / dec_cus
        for i in range(dec_cus):
            dec_bits |= 1 << int(i * step)
    sA = masked_stream(full if (fe_all or not dec_cus) else (full & ~dec_bits))
    sB = masked_stream(dec_bits if dec_cus else full)
    octx = dabgpu.Context(0, n); octx.streams_reset(E)
    dctx = dabgpu.Context(0, n)
    evA = [event(), event()]; evB = [event(), event()]

    def demod(k):
        octx.ofdm_demod_streams_dev(iq.data_ptr() + 2656 * 8, L, E, F, 0.5, soft[k & 1].data_ptr(), cyc.data_ptr(), None, sA.value)

    def decode(k, st):
        dctx.decode_frames_dev(soft[k & 1].data_ptr(), dabgpu.NB_FRAME_BITS, E, F, fib.data_ptr(), ok.data_ptr(), [sc],
                               [hist[k & 1].data_ptr()], [hist[(k & 1) ^ 1].data_ptr()], [msc.data_ptr()], st.value)

    def step(k):
        if serial:
            demod(k); decode(k, sA)
            return
        if k >= 2:
            hip.hipStreamWaitEvent(sA, evB[k & 1], 0)          # soft[k & 1] was decoded two steps ago
        demod(k)
        hip.hipEventRecord(evA[k & 1], sA)
        hip.hipStreamWaitEvent(sB, evA[k & 1], 0)
        decode(k, sB)
        hip.hipEventRecord(evB[k & 1], sB)

    for k in range(4): step(k)
    hip.hipStreamSynchronize(sA); hip.hipStreamSynchronize(sB)
    t0 = time.perf_counter()
    for k in range(reps): step(4 + k)
    hip.hipStreamSynchronize(sA); hip.hipStreamSynchronize(sB)
    dt = (time.perf_counter() - t0) / reps * 1e3
    octx.close(); dctx.close()
    hip.hipStreamDestroy(sA); hip.hipStreamDestroy(sB)
    return dt


print("device CUs: %d; step = front end + FIC/MSC decode of %d frames" % (NCU, n))
print("serial, one unmasked stream:            %.3f ms" % run(0, True, serial=True))
print("two unmasked streams (pipelined):       %.3f ms" % run(0, True))
for d in (16, 32, 48, 64, 96, 128):
    print("decoder on %3d CUs, front end on the other %3d: %.3f ms   | front end on all %d: %.3f ms" % (
        d, NCU - d, run(d, False), NCU, run(d, True)))
print("serial again:                            %.3f ms" % run(0, True, serial=True))
