"""Checker for externally dumped vectors (tests/external/*.npz; format in INTEGRATION.md section 6).

A file holds what the reference plugin sees at its two hand-over points:
  * the payload of On_OFDM_Frame (/root/reference/src/radio_block.cpp:25): `soft` int8 [n_frames][230400]
  * what BasicRadio::Process made of it (:42): `fib` uint8 [n_frames][12][32], `crc_ok` uint8 [n_frames][12], and per
    followed sub-channel `msc_<id>` uint8 [4 * n_frames][bytes per logical frame] (row t = the logical frame completed
    by CIF t of the dump; rows before `msc_valid_from` (default 15) are ignored) with its descriptor
    `subchannel_<id>` = [start_address, length, is_uep, eep_type, protection_level, bitrate_kbps]
and optionally the front end's input:
  * `iq` complex64 [n_frames][>= 76*2552], row f from the first sample the demodulator treated as PRS cyclic prefix,
    `freq_offset` float32 [n_frames] (cycles/sample, the correction applied; default 0)
Tolerances: FIB / MSC bytes and CRC flags bit-exact (FIB bytes only where the dump's own CRC flag is set); soft bits
within `soft_tol` (default 1) int8 LSB.  When the soft bits differ by a pure convention (sign, or a constant scale), the
report says so -- that is a one-line change of the quantiser, not a DSP difference.

backend: "oracle" (oracle/: CPU) or a dabgpu.Context (HIP through the C ABI).  Raises AssertionError with the report."""
import numpy as np

NB_FRAME_BITS = 230400
SYMS = 76 * 2552


def _subchannels(d):
    out = []
    for k in d.files:
        if k.startswith("subchannel_"):
            ident = k[len("subchannel_"):]
            out.append((ident, [int(v) for v in d[k]], d["msc_" + ident]))
    return out


def soft_convention(ours, theirs):
    """How two soft-bit arrays relate: ('equal', max |delta|) / ('sign', ...) / ('scale', factor) / ('different', ...)"""
    a, b = ours.astype(np.int32).ravel(), theirs.astype(np.int32).ravel()
    d = np.abs(a - b).max()
    if d <= 1:
        return "equal", int(d)
    if np.abs(a + b).max() <= 1:
        return "sign", int(np.abs(a + b).max())
    nz = (np.abs(a) > 16) & (np.abs(b) > 0)
    if nz.any():
        r = b[nz].astype(np.float64) / a[nz]
        med = float(np.median(r))
        if med != 0 and np.abs(b - np.rint(med * a)).max() <= max(2, int(abs(med))):
            return "scale", med
    agree = float(((a > 0) == (b > 0))[(a != 0) & (b != 0)].mean()) if ((a != 0) & (b != 0)).any() else 0.0
    return "different", agree


def check_file(path, backend, soft_tol=1):
    from oracle import oracle as O
    d = np.load(path)
    report = {"file": str(path)}
    hip = backend != "oracle"
    soft_given = d["soft"] if "soft" in d.files else None
    if "iq" in d.files:
        iq = np.ascontiguousarray(d["iq"], np.complex64)
        fo = np.asarray(d["freq_offset"], np.float32) if "freq_offset" in d.files else np.zeros(iq.shape[0], np.float32)
        if hip:
            soft, _, _ = backend.ofdm_demod_frames(iq, fo)
        else:
            soft = np.stack([O.ofdm_demod_frame(iq[f, :SYMS], float(fo[f]))[0] for f in range(iq.shape[0])])
        if soft_given is not None:
            kind, val = soft_convention(soft, soft_given[:soft.shape[0]])
            report["soft"] = (kind, val)
            assert kind == "equal" and val <= soft_tol, \
                "soft bits differ from the dump: %s (%r) -- 'sign' / 'scale' are pure quantiser conventions" % (kind, val)
        soft_for_decode = soft_given if soft_given is not None else soft
    else:
        assert soft_given is not None, "the file holds neither `iq` nor `soft`"
        soft_for_decode = soft_given
    soft_for_decode = np.ascontiguousarray(soft_for_decode, np.int8)
    n = soft_for_decode.shape[0]
    if "fib" in d.files:
        if hip:
            fib, ok = backend.fic_decode(soft_for_decode)
        else:
            r = [O.fic_decode(soft_for_decode[f]) for f in range(n)]
            fib, ok = np.stack([x[0] for x in r]), np.stack([x[1] for x in r])
        want_ok = np.asarray(d["crc_ok"]).astype(bool) if "crc_ok" in d.files else np.ones((n, 12), bool)
        report["fib_crc_flags_equal"] = bool((ok.astype(bool) == want_ok).all())
        report["fibs_compared"] = int(want_ok.sum())
        assert report["fib_crc_flags_equal"], "FIB CRC flags differ from the dump (%d of %d)" % (
            int((ok.astype(bool) != want_ok).sum()), want_ok.size)
        assert want_ok.any(), "no FIB of the dump has a passing CRC: nothing to compare"
        assert (fib[want_ok] == np.asarray(d["fib"])[want_ok]).all(), "FIB bytes differ from the dump"
    first = int(d["msc_valid_from"]) if "msc_valid_from" in d.files else 15
    for ident, desc, want in _subchannels(d):
        import dabgpu
        sc = dabgpu.Subchannel(*desc)
        if hip:
            got, _ = backend.msc_decode(sc, soft_for_decode, n_streams=1)
            got = got[0]
        else:
            if sc.is_uep:
                idx = next(i for i in range(64) if tuple(O.uep_profile(i)[:2]) == (sc.bitrate_kbps, sc.protection_level))
                mask, _, nsteps, _ = O.uep_puncture_mask(idx)
            else:
                mask, _, nsteps, _ = O.eep_puncture_mask(sc.eep_type, sc.protection_level, sc.bitrate_kbps)
            cifs = soft_for_decode[:, 9216:].reshape(n * 4, 55296)[:, sc.start_address * 64:(sc.start_address + sc.length) * 64]
            got = np.zeros((n * 4, (nsteps - 6) // 8), np.uint8)
            for t in range(15, n * 4):
                got[t] = O.msc_decode_lf(O.time_deinterleave(cifs[t - 15:t + 1]), mask, nsteps)
        report["msc_" + ident] = int(max(0, n * 4 - first))
        assert n * 4 > first, "the dump is too short for the 16-CIF de-interleaver (need > %d CIFs)" % first
        assert (got[first:] == np.asarray(want)[first:n * 4]).all(), "MSC bytes of sub-channel %s differ from the dump" % ident
    return report
