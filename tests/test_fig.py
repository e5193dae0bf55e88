"""FIC -> FIG -> database -> DAB+ channels (SURVEY.md 8f-4 with 8f-3 behind it).
CPU: the host FIG parser (C++, no GPU needed) against the Python restatement and against what the synthetic
multiplex says about itself; fuzzing with random CRC-valid FIBs.  GPU: the demo wired like Radio_Block discovers the
services of an unknown multiplex from IQ and hands out their access units."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT
from dabgpu import synth
from oracle import fig_oracle as FO

HOST = os.path.join(ROOT, "sdrplusplus-dab-radio-plugin_amd", "host")
SERVICES = [("Radio One", 0xC221, 3, 0, 3, 64, 0), ("Jazz 24", 0xC222, 7, 0, 2, 48, 48), ("News", 0xC223, 9, 1, 2, 32, 200)]


@pytest.fixture(scope="module")
def parser_exe():
    exe = os.path.join(HOST, "test_fig_parser")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", HOST, "test_fig_parser"], stdout=subprocess.DEVNULL)
    return exe


@pytest.fixture(scope="module")
def ensemble():
    return synth.ServiceEnsemble(1, SERVICES, n_frames=5)


def run_parser(exe, fibs, tmp_path):
    path = tmp_path / "fibs.bin"
    np.ascontiguousarray(fibs, np.uint8).tofile(path)
    out = subprocess.run([exe, str(path)], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0
    lines = out.stdout.splitlines()
    return [l for l in lines if not l.startswith("#")], [l for l in lines if l.startswith("#")][0]


def test_oracle_reads_back_what_the_multiplex_says(ensemble):
    db = FO.parse_fibs(ensemble.fibs.reshape(-1, 32))
    assert db.ensemble["id"] == 0xC181 and db.ensemble["label"] == "Synth Ensemble"
    assert sorted(db.subchannels) == [3, 7, 9]
    for (label, sid, scid, option, level, bitrate, start), size in zip(SERVICES, ensemble.sizes):
        s = db.subchannels[scid]
        assert (s["start_address"], s["length"], s["is_uep"], s["eep_type"], s["eep_prot_level"]) == (start, size, False, option, level - 1)
        sv = db.services[sid]
        assert sv["label"] == label
        (comp,) = sv["components"]
        assert (comp["subchannel_id"], comp["transport_mode"], comp["audio_service_type"], comp["is_primary"]) == (scid, 0, 63, True)
        assert comp["scids"] == 0 and comp["label"] == (label + " main")[:16] and comp["lang"] >= 0x09
    # the rest of what the multiplex says about itself (FIG 0/9, 0/17, 0/6, 0/21, 0/24)
    assert db.country == (0xE1, 10, 1)
    assert [db.services[sid]["pty"] for (_l, sid, *_r) in SERVICES] == [1, 2, 3] and db.services[0xC221]["lang"] == 0x09
    assert db.links == {0x123: {"active": 1, "hard": 1, "intl": 0, "service": 0xC221}}
    assert db.fm == {0xC479: {"lsn": 0x123, "tc": True, "freqs": [98300000, 101100000]}}
    assert db.other == {0xC182: {"cont": False, "freqs": [225648000], "services": [0xC221]}}


def test_date_time_round_trip():
    for (y, m, d) in ((2024, 2, 29), (1999, 12, 31), (2000, 1, 1), (2026, 10, 1), (2100, 2, 28)):
        assert FO.mjd_to_ymd(synth.mjd(y, m, d)) == (y, m, d)
    assert synth.mjd(1858, 11, 17) == 0 and synth.mjd(2000, 1, 1) == 51544      # known answers
    db = FO.parse_fibs(synth.pack_fibs([synth.fig0_0(0xC181, 4999), synth.fig0_10(2026, 10, 1, 13, 37, 42, 123)]))
    assert db.datetime == (2026, 10, 1, 13, 37, 42, 123) and db.lines()[-1] == "datetime 2026-10-01 13:37:42.123 cif=4999"
    db = FO.parse_fibs(synth.pack_fibs([synth.fig0_10(2026, 10, 1, 13, 37)]))         # short form: no seconds
    assert db.datetime == (2026, 10, 1, 13, 37, 0, 0)


def test_host_parser_equals_oracle(parser_exe, ensemble, tmp_path):
    fibs = ensemble.fibs.reshape(-1, 32).copy()
    got, stats = run_parser(parser_exe, fibs, tmp_path)
    assert got == FO.parse_fibs(fibs).lines()
    assert got[-1].startswith("datetime 2024-02-29 23:59:5")
    assert "fibs_bad_crc=0" in stats and "conflicts=0" in stats
    # FIBs damaged in transit must be ignored, not half-parsed
    bad = np.concatenate([fibs, fibs])             # the multiplex is cyclic: two rounds of it,
    bad[::7, 5] ^= 0x40                            # no FIB damaged in both (60 FIBs per round, 60 % 7 != 0)
    got, stats = run_parser(parser_exe, bad, tmp_path)
    assert got == FO.parse_fibs(bad).lines() == FO.parse_fibs(fibs).lines()
    assert "fibs_bad_crc=%d" % len(bad[::7]) in stats


def test_host_parser_other_forms(parser_exe, tmp_path):
    figs = [synth.fig0_0(0xE1C5, 4999),
            synth.fig0_1([{"id": 1, "start": 0, "uep_index": 37}, {"id": 62, "start": 700, "option": 1, "level": 4, "size": 90},
                          {"id": 5, "start": 100, "option": 3, "level": 1, "size": 12}]),       # reserved option: skipped
            synth.fig0(2, synth._bits((0xE0D12345, 32), (0, 1), (0, 3), (2, 4), (0, 2), (0, 6), (1, 6), (1, 1), (0, 1),
                                      (1, 2), (5, 6), (62, 6), (0, 1), (0, 1)), pd=1),           # 32-bit SId; one data component
            synth.fig0(9, bytes([0x12, 0x34, 0x56])), synth.fig0(13, bytes([0x12, 0x34, 0x56])),   # 0/13: not followed, skipped
            synth.fig1(1, 0x1234, "Short"), synth.fig1(5, 0x1234, "Other kind")]
    fibs = synth.pack_fibs(figs)
    got, _ = run_parser(parser_exe, fibs, tmp_path)
    want = FO.parse_fibs(fibs).lines()
    assert got == want
    assert "subchannel id=1 start=0 length=0 uep=1 uep_index=37 eep_type=0 eep_level=0" in got
    assert "subchannel id=62 start=700 length=90 uep=0 uep_index=0 eep_type=1 eep_level=3" in got
    assert not any(l.startswith("subchannel id=5 ") for l in got)
    assert "component service=E0D12345 subchannel=1 tmid=0 ascty=0 primary=1 scids=-1 lang=0 label=[]" in got
    assert sum(l.startswith("component") for l in got) == 1
    assert "service id=1234 label=[Short] pty=-1 lang=0 bits32=0" in got
    assert "service id=E0D12345 label=[] pty=-1 lang=0 bits32=1" in got
    assert "ensemble_info ecc=34 lto=90 inter_table=86" in got         # the FIG 0/9 above: 0x12 = +18 half hours


def test_host_parser_linking_and_frequency_forms(parser_exe, tmp_path):
    """Every form of the FIGs behind the GUI's service / linked-services / component tabs, including the ones the
    synthetic multiplex does not use: international and 32-bit id lists, DRM, a status-only linking entry, FIG 0/8
    long form and extension byte, FIG 0/17 with language and complementary code, labels that arrive too early."""
    S = synth
    figs = [S.fig0_0(0xE1C5, 17),
            S.fig1_4(0xC221, 2, "too early"),                                     # no FIG 0/8 yet: dropped
            S.fig0_2([{"sid": 0xC221, "components": [{"subchannel": 3, "ascty": 63}, {"subchannel": 4, "ascty": 0, "primary": False}]}]),
            S.fig0(2, S._bits((0xE0D12345, 32), (0, 1), (0, 3), (1, 4), (0, 2), (63, 6), (9, 6), (1, 1), (0, 1)), pd=1),
            S.fig0_8([(0xC221, 0, 3), (0xC221, 2, 4)]),
            S.fig0(8, S._bits((0xC221, 16), (1, 1), (0, 3), (5, 4), (1, 1), (0, 3), (0x123, 12), (0, 8))),   # long form + ext: skipped
            S.fig0_8([(0xE0D12345, 1, 9)], pd=1),
            S.fig1_4(0xC221, 2, "Second audio"), S.fig1_4(0xE0D12345, 1, "Data comp", pd=1), S.fig1_5(0xE0D12345, "Data service"),
            S.fig1_5(0xE0D99999, "unknown service"),                               # not announced by FIG 0/2: dropped
            S.fig0_9(0xE2, -7, 2),
            S.fig0_17([(0xC221, 14, 0x0F), (0xBEEF, 3, None)]),                    # second service unknown: dropped
            S.fig0(17, S._bits((0xC221, 16), (0, 1), (0, 1), (0, 1), (1, 1), (0, 4), (0, 3), (9, 5), (0, 3), (20, 5))),  # conflict + CC
            S.fig0_5([(3, 0x08), (4, 0x1D), (50, 0x22)]),
            S.fig0(5, S._bits((1, 1), (0, 3), (0x456, 12), (0x11, 8))),             # long form: skipped
            S.fig0_6(0x0AB, [0xC221, 0xC222], idlq=0), S.fig0_6(0x0AB, [0xD311, 0xD312], idlq=1),
            S.fig0_6(0x0AC, [0xE1C221, 0xE2D123], idlq=3, ils=1, hard=0), S.fig0_6(0x0AD, [0xE0D12345], pd=1, active=0),
            S.fig0(6, S._bits((0, 1), (1, 1), (0, 1), (0, 1), (0x0AE, 12))),        # status only
            S.fig0_21([(0xD311, 8, 1, [87600000, 107900000]), (0xE1C5, 0, 1, [174928000, 239200000])]),
            S.fig0_21([(0xE2D123, 6, 0, [6095000, 15120000]), (0xD311, 8, 0, [87600000, 99900000])]),
            S.fig0_24([(0xC221, [0xE1C6, 0xE1C7])]), S.fig0_24([(0xE0D12345, [0xE1C6])], pd=1)]
    fibs = S.pack_fibs(figs)
    got, stats = run_parser(parser_exe, fibs, tmp_path)
    assert got == FO.parse_fibs(fibs).lines()
    for want in ("ensemble_info ecc=E2 lto=-35 inter_table=2",
                 "service id=C221 label=[] pty=14 lang=15 bits32=0",
                 "component service=C221 subchannel=3 tmid=0 ascty=63 primary=1 scids=0 lang=8 label=[]",
                 "component service=C221 subchannel=4 tmid=0 ascty=0 primary=0 scids=2 lang=29 label=[Second audio]",
                 "service id=E0D12345 label=[Data service] pty=-1 lang=0 bits32=1",
                 "component service=E0D12345 subchannel=9 tmid=0 ascty=63 primary=1 scids=1 lang=0 label=[Data comp]",
                 "link lsn=171 active=1 hard=1 intl=0 service=C221",
                 "link lsn=172 active=1 hard=0 intl=1 service=none",
                 "link lsn=173 active=0 hard=1 intl=0 service=E0D12345",
                 "fm pi=D311 lsn=171 tc=1 freqs=87600000,107900000,99900000",
                 "fm pi=D312 lsn=171 tc=0 freqs=",
                 "drm code=E1C221 lsn=172 tc=0 freqs=",
                 "drm code=E2D123 lsn=172 tc=0 freqs=6095000,15120000",
                 "other_ensemble id=E1C5 cont=1 freqs=174928000,239200000 services=",
                 "other_ensemble id=E1C6 cont=0 freqs= services=C221,E0D12345"):
        assert want in got, want
    assert not any("lsn=174" in l for l in got) and not any("BEEF" in l or "E0D99999" in l for l in got)
    assert "conflicts=1" in stats


def test_host_parser_fuzz(parser_exe, tmp_path):
    """Random bytes with a valid CRC: whatever they mean, both parsers must read the same thing and neither may
    crash (lengths that overrun the FIB, truncated entries, reserved fields ...)."""
    rng = np.random.default_rng(99)
    fibs = synth.make_fibs(rng, 3000)
    # make FIG headers plausible more often: type 0/1 with assorted lengths at the start of the FIB
    fibs[:1500, 0] = rng.choice(np.array([0x05, 0x0D, 0x1D, 0x35, 0x15, 0x3F, 0x03], np.uint8), 1500)
    fibs[:1500, 1] &= 0x27
    # ... and the extensions behind the linking / component / programme-type entities: 0/5 0/6 0/8 0/9 0/17 0/21 0/24
    ext = rng.choice(np.array([5, 6, 8, 9, 17, 21, 24], np.uint8), 1200)
    fibs[1500:2700, 0] = rng.choice(np.array([0x09, 0x12, 0x1D, 0x1C], np.uint8), 1200)
    fibs[1500:2700, 1] = (fibs[1500:2700, 1] & 0x20) | ext
    fibs[1500:2700, 2:6] &= rng.choice(np.array([0xFF, 0x0F, 0xC3], np.uint8), (1200, 4))   # small ids collide more often
    fibs[2700:2900, 0] = 0x20 | rng.choice(np.array([22, 24, 25, 21], np.uint8), 200)      # FIG 1 with assorted lengths
    fibs[2700:2900, 1] = (fibs[2700:2900, 1] & 0xF0) | rng.choice(np.array([0, 1, 4, 5], np.uint8), 200)
    fibs[2700:2900, 2:7] &= 0x83
    for f in fibs:
        c = synth.crc16(f[:30])
        f[30], f[31] = c >> 8, c & 0xFF
    got, _ = run_parser(parser_exe, fibs, tmp_path)
    assert got == FO.parse_fibs(fibs).lines()


def test_host_parser_entity_lists_are_capped(parser_exe, tmp_path):
    """A FIC that announces thousands of distinct services, DRM / FM services, other ensembles and linkage sets (each
    FIB CRC-valid) must not grow the database without bound: every entity list stops at 1024 entries, the overflow is
    counted as conflicts, and both parsers agree on what is kept."""
    S = synth
    figs = [S.fig0_0(0xE1C5, 1)]
    for k in range(1500):
        figs.append(S.fig0_2([{"sid": 0x1000 + k, "components": [{"subchannel": k % 64, "ascty": 63}]}]))
        figs.append(S.fig0_21([(0x200000 + k, 6, 0, [6095000]), (0x3000 + k, 8, 0, [87600000]), (0x4000 + k, 0, 1, [174928000])]))
        figs.append(S.fig0_6(k & 0xFFF, [0x1000 + k], idlq=0))
    fibs = S.pack_fibs(figs)
    got, stats = run_parser(parser_exe, fibs, tmp_path)
    want = FO.parse_fibs(fibs).lines()
    assert got == want
    for kind in ("service ", "drm ", "fm ", "other_ensemble ", "link ", "component "):
        n = sum(l.startswith(kind) for l in got)
        assert n == 1024, (kind, n)
    assert int(stats.split("conflicts=")[1]) >= 5 * (1500 - 1024)


def test_host_parser_under_sanitizers(tmp_path):
    """The same fuzz input through an AddressSanitizer + UBSan build of the parser (CPU build only: the pool has no
    GPU sanitizers)."""
    exe = str(tmp_path / "test_fig_parser_asan")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                           "-I" + HOST, "-I" + os.path.join(ROOT, "include"), os.path.join(HOST, "tests", "test_fig_parser.cpp"),
                           os.path.join(HOST, "dab", "fic", "fic_parser.cpp"), "-o", exe])
    rng = np.random.default_rng(7)
    fibs = synth.make_fibs(rng, 6000)
    fibs[:4500, 0] = rng.choice(np.array([0x05, 0x0D, 0x1D, 0x35, 0x15, 0x3F, 0x03, 0x1F, 0x3E, 0x01, 0x21, 0x02], np.uint8), 4500)
    fibs[:4500, 1] &= 0x27
    for f in fibs:
        c = synth.crc16(f[:30])
        f[30], f[31] = c >> 8, c & 0xFF
    path = tmp_path / "fuzz.bin"
    fibs.tofile(path)
    r = subprocess.run([exe, str(path)], capture_output=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:].decode("latin-1")


@pytest.mark.gpu
def test_unknown_multiplex_from_iq_to_access_units(built, ensemble, tmp_path):
    if not os.path.exists(os.path.join(HOST, "dab_host_demo")):
        subprocess.check_call(["make", "-C", HOST, "-j4"], stdout=subprocess.DEVNULL)
    rng = np.random.default_rng(8)
    cycles = 3
    iq = np.tile(ensemble.iq().ravel(), cycles)
    iq = synth.channel(iq, snr_db=20.0, cfo=1.2 / 2048, rng=rng)
    path = tmp_path / "iq.cf32"
    np.concatenate([iq[-30000:], iq, iq[:synth.NB_NULL + 5000]]).astype(np.complex64).tofile(path)
    prefix = str(tmp_path / "out")
    r = subprocess.run([os.path.join(HOST, "dab_host_demo"), str(path), prefix, "40000"], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stderr
    assert "frames_desync=0" in r.stdout, r.stdout
    db = open(prefix + ".db").read().splitlines()
    want = FO.parse_fibs(ensemble.fibs.reshape(-1, 32)).lines()
    assert [l for l in db if not l.startswith(("channel", "datetime"))] == [l for l in want if not l.startswith("datetime")]
    chans = [l for l in db if l.startswith("channel")]
    assert len(chans) == 3 and all("firecode_error=0" in l and "rs_error=0" in l and "rate=48000 sbr=1 stereo=1" in l for l in chans)
    # access units, per sub-channel, in transmission order
    raw = np.fromfile(prefix + ".aus", np.uint8)
    got = {3: [], 7: [], 9: []}
    i = 0
    while i < raw.size:
        scid, idx, total, n = int(raw[i]), int(raw[i + 1]), int(raw[i + 2]), int(raw[i + 3]) | (int(raw[i + 4]) << 8)
        got[scid].append((idx, total, raw[i + 5:i + 5 + n]))
        i += 5 + n
    n_lf = 4 * 5 * cycles - 15                       # logical frames the de-interleaver completes
    for k, (label, sid, scid, *_r) in enumerate(SERVICES):
        sent = []
        for sf in range(n_lf // 5):
            aus = ensemble.aus[k][sf % len(ensemble.aus[k])]
            sent += [(a, len(aus), au) for a, au in enumerate(aus)]
        # the first frames may be lost while the fine-frequency loop settles: compare the tail
        assert len(got[scid]) >= len(sent) - 2 * 3
        tail = got[scid][-(len(sent) - 6):]
        for (gi, gt, gb), (si, st, sb) in zip(tail, sent[-len(tail):]):
            pay = sb[:-2]                            # the channel hands on the payload without its (checked) CRC16
            assert (gi, gt) == (si, st) and gb.size == pay.size and (gb == pay).all()


@pytest.mark.gpu
def test_dab_layer2_service_on_uep_subchannel(built, tmp_path):
    """A DAB (MPEG layer II) service announced with the FIG 0/1 short form: the receiver has to take bit rate,
    protection and size from the UEP table index, open the sub-channel and hand out the layer-II frames."""
    if not os.path.exists(os.path.join(HOST, "dab_host_demo")):
        subprocess.check_call(["make", "-C", HOST, "-j4"], stdout=subprocess.DEVNULL)
    ens = synth.ServiceEnsemble(3, [("Plus One", 0xC331, 2, 0, 3, 48, 0)], n_frames=5,
                                dab_services=[("Classic", 0xC332, 11, 17, 100)])      # 64 kbit/s UEP level 2: 58 CUs
    rng = np.random.default_rng(4)
    cycles = 3
    iq = synth.channel(np.tile(ens.iq().ravel(), cycles), snr_db=18.0, cfo=-0.4 / 2048, rng=rng)
    path = tmp_path / "iq.cf32"
    np.concatenate([iq[-30000:], iq, iq[:synth.NB_NULL + 5000]]).astype(np.complex64).tofile(path)
    prefix = str(tmp_path / "out")
    r = subprocess.run([os.path.join(HOST, "dab_host_demo"), str(path), prefix, "65536"], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0 and "frames_desync=0" in r.stdout, r.stdout + r.stderr
    db = open(prefix + ".db").read().splitlines()
    assert "subchannel id=11 start=100 length=58 uep=1 uep_index=17 eep_type=0 eep_level=0" in db   # size from the table
    assert any(l.startswith("service id=C332 label=[Classic] ") for l in db)
    assert any(l.startswith("component service=C332 subchannel=11 tmid=0 ascty=0 primary=1 ") for l in db)
    mp2 = [l for l in db if l.startswith("channel subchannel=11")]
    assert len(mp2) == 1 and "header_errors=0" in mp2[0] and "rate=48000 stereo=1" in mp2[0]
    raw = np.fromfile(prefix + ".aus", np.uint8)
    frames, i = [], 0
    while i < raw.size:
        scid, n = int(raw[i]), int(raw[i + 3]) | (int(raw[i + 4]) << 8)
        if scid == 11:
            frames.append(raw[i + 5:i + 5 + n])
        i += 5 + n
    n_lf = 4 * 5 * cycles - 15
    assert n_lf - 8 <= len(frames) <= n_lf
    sent = [ens.mp2_frames[0][k % 20] for k in range(n_lf)]
    for got, want in zip(frames[::-1], sent[::-1]):                 # compare from the end: the start may be lost to lock-in
        assert got.size == 192 and (got == want).all()
