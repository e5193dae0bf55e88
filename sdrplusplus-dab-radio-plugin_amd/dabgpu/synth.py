"""Synthetic DAB Mode-I transmitter (numpy) -- known-answer generator and bench input.

The reference ships no fixtures or tests; upstream DAB-Radio has a
`simulate_transmitter` example that is not in /root/reference.  This is an independent
restatement of the TRANSMIT side of ETSI EN 300 401 (clauses 5.2, 11, 12, 14), written
without sharing code or tables with oracle/ so that round trips cross-check both.

Frame layout produced: null symbol (2656 zeros) + PRS + 75 data symbols of 2552
samples = 196608 complex64 samples at 2.048 MSPS (what the plugin's VFO delivers,
/root/reference/src/dab_module.cpp:144-149).
"""
import numpy as np

NB_FFT = 2048
NB_CP = 504
NB_SYM = NB_FFT + NB_CP
NB_NULL = 2656
NB_SYMBOLS = 76
NB_CARRIERS = 1536
NB_SYM_BITS = 2 * NB_CARRIERS
NB_FRAME_BITS = 75 * NB_SYM_BITS
NB_FRAME_SAMPLES = NB_NULL + NB_SYMBOLS * NB_SYM
NB_FIC_BITS = 3 * NB_SYM_BITS
NB_CIF_BITS = 864 * 64
TDI_DELAY = np.array([0, 8, 4, 12, 2, 10, 6, 14, 1, 9, 5, 13, 3, 11, 7, 15])

# generator polynomials, octal, MSB = current bit (clause 11.1.1)
_GEN_OCT = (0o133, 0o171, 0o145, 0o133)


# ----------------------------------------------------------------------------- tables
def carrier_of_data_index():
    """k_n (carrier number in [-768,768]\\{0}) for data index n (clause 14.6)."""
    pi = np.zeros(NB_FFT, np.int64)
    for i in range(1, NB_FFT):
        pi[i] = (13 * pi[i - 1] + 511) % NB_FFT
    keep = pi[(pi >= 256) & (pi <= 1792) & (pi != 1024)]
    return keep - 1024


def prs_carriers():
    """z_{1,k} for k=-768..768 (k=0 -> 0): dict-free dense array indexed k+768 (clause 14.3.2)."""
    h = np.array([
        [0, 2, 0, 0, 0, 0, 1, 1, 2, 0, 0, 0, 2, 2, 1, 1] * 2,
        [0, 3, 2, 3, 0, 1, 3, 0, 2, 1, 2, 3, 2, 3, 3, 0] * 2,
        [0, 0, 0, 2, 0, 2, 1, 3, 2, 2, 0, 2, 2, 0, 1, 3] * 2,
        [0, 1, 2, 1, 0, 3, 3, 2, 2, 3, 2, 1, 2, 1, 3, 2] * 2,
    ])
    # (k', i, n) Mode I, table 39
    rows = [(-768, 0, 1), (-736, 1, 2), (-704, 2, 0), (-672, 3, 1), (-640, 0, 3), (-608, 1, 2),
            (-576, 2, 2), (-544, 3, 3), (-512, 0, 2), (-480, 1, 1), (-448, 2, 2), (-416, 3, 3),
            (-384, 0, 1), (-352, 1, 2), (-320, 2, 3), (-288, 3, 3), (-256, 0, 2), (-224, 1, 2),
            (-192, 2, 2), (-160, 3, 1), (-128, 0, 1), (-96, 1, 3), (-64, 2, 1), (-32, 3, 2),
            (1, 0, 3), (33, 3, 1), (65, 2, 1), (97, 1, 1), (129, 0, 2), (161, 3, 2), (193, 2, 1),
            (225, 1, 0), (257, 0, 2), (289, 3, 2), (321, 2, 3), (353, 1, 3), (385, 0, 0),
            (417, 3, 2), (449, 2, 1), (481, 1, 3), (513, 0, 3), (545, 3, 3), (577, 2, 3),
            (609, 1, 0), (641, 0, 3), (673, 3, 0), (705, 2, 1), (737, 1, 1)]
    z = np.zeros(2 * 768 + 1, np.complex128)
    for kp, i, n in rows:
        for j in range(32):
            z[kp + j + 768] = 1j ** int((h[i][j] + n) % 4)
    return z


def puncture_vector(pi):
    order = [0, 4, 2, 6, 1, 5, 3, 7]
    ones = [1] * 8
    for s in range(pi):
        ones[order[s % 8]] += 1
    return np.array([[1 if b < ones[g] else 0 for b in range(4)] for g in range(8)], np.uint8).ravel()


def _mask_from_profile(blocks):
    parts = [np.tile(puncture_vector(pi), 4 * L) for L, pi in blocks if L > 0]
    parts.append(np.tile(np.array([1, 1, 0, 0], np.uint8), 6))
    return np.concatenate(parts)


def fic_mask():
    return _mask_from_profile([(21, 16), (3, 15)])


def eep_profile(option, level, bitrate):
    """-> (blocks [(L,PI),...], size_cu).  Clause 11.3.2."""
    if option == 0:
        n = bitrate // 8
        assert bitrate == 8 * n
        if level == 1:
            return [(6 * n - 3, 24), (3, 23)], 12 * n
        if level == 2:
            if n == 1:
                return [(5, 13), (1, 12)], 8
            return [(2 * n - 3, 14), (4 * n + 3, 13)], 8 * n
        if level == 3:
            return [(6 * n - 3, 8), (3, 7)], 6 * n
        if level == 4:
            return [(4 * n - 3, 3), (2 * n + 3, 2)], 4 * n
    else:
        n = bitrate // 32
        assert bitrate == 32 * n
        pis = {1: (10, 9), 2: (6, 5), 3: (4, 3), 4: (2, 1)}[level]
        cu = {1: 27, 2: 21, 3: 18, 4: 15}[level] * n
        return [(24 * n - 3, pis[0]), (3, pis[1])], cu
    raise ValueError("bad EEP profile")


def eep_mask(option, level, bitrate):
    blocks, cu = eep_profile(option, level, bitrate)
    m = _mask_from_profile(blocks)
    assert m.size == 4 * (bitrate * 24 + 6) and int(m.sum()) == cu * 64
    return m, cu


# UEP protection profiles (clause 11.3.1, Table 8) in table-index order: bitrate level size L1 L2 L3 L4 PI1 PI2 PI3 PI4
# padding.  Restated from memory; uep_profile() asserts the two identities every row must satisfy.
_UEP_TABLE = [tuple(int(v) for v in line.split()) for line in """
    32 5 16 3 4 17 0 5 3 2 0 0
    32 4 21 3 3 18 0 11 6 5 0 0
    32 3 24 3 4 14 3 15 9 6 8 0
    32 2 29 3 4 14 3 22 13 8 13 0
    32 1 35 3 5 13 3 24 17 12 17 4
    48 5 24 4 3 26 3 5 4 2 3 0
    48 4 29 3 4 26 3 9 6 4 6 0
    48 3 35 3 4 26 3 15 10 6 9 4
    48 2 42 3 4 26 3 24 14 8 15 0
    48 1 52 3 5 25 3 24 18 13 18 0
    56 5 29 6 10 23 3 5 4 2 3 0
    56 4 35 6 10 23 3 9 6 4 5 0
    56 3 42 6 12 21 3 16 7 6 9 0
    56 2 52 6 10 23 3 23 13 8 13 8
    64 5 32 6 9 31 2 5 3 2 3 0
    64 4 42 6 9 33 0 11 6 5 0 0
    64 3 48 6 12 27 3 16 8 6 9 0
    64 2 58 6 10 29 3 23 13 8 13 8
    64 1 70 6 11 28 3 24 18 12 18 4
    80 5 40 6 10 41 3 6 3 2 3 0
    80 4 52 6 10 41 3 11 6 5 6 0
    80 3 58 6 11 40 3 16 8 6 7 0
    80 2 70 6 10 41 3 23 13 8 13 8
    80 1 84 6 10 41 3 24 17 12 18 4
    96 5 48 7 9 53 3 5 4 2 4 0
    96 4 58 7 10 52 3 9 6 4 6 0
    96 3 70 6 12 51 3 16 9 6 10 4
    96 2 84 6 10 53 3 22 12 9 12 0
    96 1 104 6 13 50 3 24 18 13 19 0
    112 5 58 14 17 50 3 5 4 2 5 0
    112 4 70 11 21 49 3 9 6 4 8 0
    112 3 84 11 23 47 3 16 8 6 9 0
    112 2 104 11 21 49 3 23 12 9 14 4
    128 5 64 12 19 62 3 5 3 2 4 0
    128 4 84 11 21 61 3 11 6 5 7 0
    128 3 96 11 22 60 3 16 9 6 10 4
    128 2 116 11 21 61 3 22 12 9 14 0
    128 1 140 11 20 62 3 24 17 13 19 8
    160 5 80 11 19 87 3 5 4 2 4 0
    160 4 104 11 23 83 3 11 6 5 9 0
    160 3 116 11 24 82 3 16 8 6 11 0
    160 2 140 11 21 85 3 22 11 9 13 0
    160 1 168 11 22 84 3 24 18 12 19 0
    192 5 96 11 20 110 3 6 4 2 5 0
    192 4 116 11 22 108 3 10 6 4 9 0
    192 3 140 11 24 106 3 16 10 6 11 0
    192 2 168 11 20 110 3 22 13 9 13 8
    192 1 208 11 21 109 3 24 20 13 24 0
    224 5 116 12 22 131 3 8 6 2 6 4
    224 4 140 12 26 127 3 12 8 4 11 0
    224 3 168 11 20 134 3 16 10 7 9 0
    224 2 208 11 22 132 3 24 16 10 15 0
    224 1 232 11 24 130 3 24 20 12 20 4
    256 5 128 11 24 154 3 6 5 2 5 0
    256 4 168 11 24 154 3 12 9 5 10 4
    256 3 192 11 27 151 3 16 10 7 10 0
    256 2 232 11 22 156 3 24 14 10 13 8
    256 1 280 11 26 152 3 24 19 14 18 4
    320 5 160 11 26 200 3 8 5 2 6 4
    320 4 208 11 25 201 3 13 9 5 10 8
    320 2 280 11 26 200 3 24 17 9 17 0
    384 5 192 11 27 247 3 8 6 2 7 0
    384 3 280 11 24 250 3 16 9 7 10 4
    384 1 416 12 28 245 3 24 20 14 23 8
""".strip().splitlines()]


def uep_index(bitrate, level):
    for i, r in enumerate(_UEP_TABLE):
        if r[0] == bitrate and r[1] == level:
            return i
    raise ValueError("no UEP profile for %d kbit/s level %d" % (bitrate, level))


def uep_profile(index):
    """-> (blocks [(L,PI) x4], size_cu, padding_bits, bitrate)."""
    br, level, size, L1, L2, L3, L4, P1, P2, P3, P4, pad = _UEP_TABLE[index]
    blocks = [(L1, P1), (L2, P2), (L3, P3), (L4, P4)]
    assert sum(L for L, _ in blocks) * 32 == br * 24
    assert sum(L * (32 + 4 * P) for L, P in blocks) + 12 + pad == size * 64
    return blocks, size, pad, br


def uep_mask(index):
    blocks, cu, pad, br = uep_profile(index)
    m = _mask_from_profile(blocks)
    assert m.size == 4 * (br * 24 + 6) and int(m.sum()) + pad == cu * 64
    return m, cu


# ----------------------------------------------------------------------------- bit level
def prbs(n):
    reg = [1] * 9  # reg[0] newest
    out = np.zeros(n, np.uint8)
    for i in range(n):
        b = reg[8] ^ reg[4]
        out[i] = b
        reg = [b] + reg[:8]
    return out


def crc16(data):
    crc = 0xFFFF
    for byte in bytes(data):
        crc ^= byte << 8
        for _ in range(8):
            crc = ((crc << 1) ^ 0x1021) if (crc & 0x8000) else (crc << 1)
            crc &= 0xFFFF
    return crc ^ 0xFFFF


def make_fibs(rng, n):
    """n random FIBs: 30 data bytes + CRC16 (clause 5.2.1)."""
    fibs = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    for f in fibs:
        c = crc16(f[:30])
        f[30] = c >> 8
        f[31] = c & 0xFF
    return fibs


def conv_encode(bits):
    """rate-1/4 mother code with 6 zero tail bits -> 4*(n+6) bits, order x0,x1,x2,x3 per input."""
    a = np.concatenate([np.zeros(6, np.uint8), np.asarray(bits, np.uint8), np.zeros(6, np.uint8)])
    n = a.size - 6
    out = np.zeros((n, 4), np.uint8)
    for p, g in enumerate(_GEN_OCT):
        acc = np.zeros(n, np.uint8)
        for k in range(7):          # octal MSB (bit 6) multiplies a[i], bit (6-k) multiplies a[i-k]
            if (g >> (6 - k)) & 1:
                acc ^= a[6 - k: 6 - k + n]
        out[:, p] = acc
    return out.ravel()


def fic_encode(fibs12):
    """12 FIBs -> 9216 punctured coded bits (4 groups of 3 FIBs)."""
    m = fic_mask().astype(bool)
    pr = prbs(768)
    out = []
    for g in range(4):
        bits = np.unpackbits(np.asarray(fibs12[3 * g:3 * g + 3], np.uint8).ravel())
        out.append(conv_encode(bits ^ pr)[m])
    return np.concatenate(out)


def msc_encode_lf(lf_bytes, mask):
    """one logical frame of a subchannel -> punctured coded bits."""
    bits = np.unpackbits(np.asarray(lf_bytes, np.uint8))
    return conv_encode(bits ^ prbs(bits.size))[mask.astype(bool)]


def time_interleave(coded, cyclic=True):
    """coded [R][nbits] logical frames -> tx [R][nbits] CIF contents:
    tx[(r + d(i%16)) % R][i] = coded[r][i] (cyclic) ; non-cyclic leaves the first CIFs partly zero."""
    R, nbits = coded.shape
    tx = np.zeros_like(coded)
    d = TDI_DELAY[np.arange(nbits) % 16]
    for r in range(R):
        dst = r + d
        if cyclic:
            tx[dst % R, np.arange(nbits)] = coded[r]
        else:
            ok = dst < R
            tx[dst[ok], np.arange(nbits)[ok]] = coded[r][ok]
    return tx


# ----------------------------------------------------------------------------- OFDM
_CAR = None
_PRS = None


def _tables():
    global _CAR, _PRS
    if _CAR is None:
        _CAR = carrier_of_data_index()
        _PRS = prs_carriers()
    return _CAR, _PRS


def modulate_frame(bits):
    """230400 bits -> complex64[196608] (null + PRS + 75 symbols), unit average symbol power."""
    car, prs = _tables()
    bits = np.asarray(bits, np.uint8).reshape(75, NB_SYM_BITS)
    out = np.zeros(NB_FRAME_SAMPLES, np.complex64)
    z = prs.copy()                       # indexed k+768
    pos = NB_NULL
    scale = NB_FFT / np.sqrt(NB_CARRIERS)
    for l in range(NB_SYMBOLS):
        if l > 0:
            p = bits[l - 1].astype(np.float64)
            q = ((1 - 2 * p[:NB_CARRIERS]) + 1j * (1 - 2 * p[NB_CARRIERS:])) / np.sqrt(2.0)
            y = np.ones(2 * 768 + 1, np.complex128)
            y[car + 768] = q
            z = z * y
        spec = np.zeros(NB_FFT, np.complex128)
        k = np.arange(-768, 769)
        spec[k % NB_FFT] = z
        spec[0] = 0
        t = np.fft.ifft(spec) * scale
        out[pos:pos + NB_CP] = t[-NB_CP:]
        out[pos + NB_CP:pos + NB_SYM] = t
        pos += NB_SYM
    return out


_RESAMPLE_TAPS, _RESAMPLE_PHASES = 24, 2048


def _resample_table():
    """[phases + 1][taps] windowed-sinc (Kaiser, beta 8) interpolation weights: row q interpolates at a fraction q / phases
    between sample `half - 1` and `half` of its taps.  The DAB signal occupies 75 % of the band; with 24 taps and the
    fraction quantised to 1/2048 sample the interpolation error stays below -60 dB."""
    taps, ph = _RESAMPLE_TAPS, _RESAMPLE_PHASES
    half = taps // 2
    k = np.arange(-half + 1, half + 1, dtype=np.float64)
    frac = np.arange(ph + 1, dtype=np.float64)[:, None] / ph
    arg = k[None, :] - frac
    w = np.sinc(arg) * np.i0(8.0 * np.sqrt(np.clip(1.0 - (arg / (half + 0.5)) ** 2, 0.0, 1.0))) / np.i0(8.0)
    return (w / w.sum(axis=1, keepdims=True)).astype(np.float32)


def resample(x, ppm, chunk=1 << 20):
    """The stream as a receiver whose sample clock runs `ppm` parts per million FAST sees it: y[n] = x((n + 12) / (1 +
    ppm e-6)) -- more samples per transmitted frame, the frame period grows to 196608 (1 + ppm e-6) samples.
    x: complex numpy array, or a torch tensor (complex64, any device: long streams are resampled on the GPU)."""
    taps, ph = _RESAMPLE_TAPS, _RESAMPLE_PHASES
    half = taps // 2
    ratio = 1.0 / (1.0 + ppm * 1e-6)
    n_out = int(np.floor((x.shape[0] - 1) / ratio)) - taps
    table = _resample_table()
    if not isinstance(x, np.ndarray):                                # torch
        import torch
        tab = torch.from_numpy(table).to(x.device)
        k = torch.arange(-half + 1, half + 1, device=x.device)
        out = torch.empty(n_out, dtype=torch.complex64, device=x.device)
        for a in range(0, n_out, chunk):
            b = min(n_out, a + chunk)
            t = (torch.arange(a, b, dtype=torch.float64, device=x.device) + half) * ratio
            i0 = torch.floor(t).to(torch.int64)
            q = torch.round((t - i0) * ph).to(torch.int64)
            out[a:b] = (x[i0[:, None] + k[None, :]] * tab[q]).sum(dim=1)
        return out
    x = np.asarray(x, np.complex64)
    k = np.arange(-half + 1, half + 1)
    out = np.empty(n_out, np.complex64)
    for a in range(0, n_out, chunk):
        b = min(n_out, a + chunk)
        t = (np.arange(a, b, dtype=np.float64) + half) * ratio       # positions in x (shifted so that no index is < 0)
        i0 = np.floor(t).astype(np.int64)
        q = np.rint((t - i0) * ph).astype(np.int64)
        out[a:b] = (x[i0[:, None] + k[None, :]] * table[q]).sum(axis=1)
    return out


def fading_gain(n, rng, doppler, rice_k=4.0, fs=2.048e6, oscillators=8):
    """Slow flat fading: a line-of-sight component plus a sum of `oscillators` sinusoids with Doppler shifts up to
    `doppler` Hz (Jakes), unit mean power; rice_k = LOS / scattered power."""
    t = np.arange(n, dtype=np.float64) / fs
    g = np.zeros(n, np.complex128)
    for _ in range(oscillators):
        fd = doppler * np.cos(rng.uniform(0, 2 * np.pi))
        g += np.exp(1j * (2 * np.pi * fd * t + rng.uniform(0, 2 * np.pi)))
    g *= np.sqrt(1.0 / (oscillators * (rice_k + 1.0)))
    return g + np.sqrt(rice_k / (rice_k + 1.0))


def channel(iq, snr_db=None, cfo=0.0, rng=None, phase0=0.0, paths=None, sco_ppm=0.0, fading_hz=0.0, rice_k=4.0, gain=None):
    """The synthetic channel.  In the order a signal meets them:
      paths      multipath: [(delay in samples, complex gain), ...]; the first path is usually (0, 1).  The total is
                 normalised to unit power, so snr_db keeps its meaning.  None = one path.
      fading_hz  slow flat fading with this maximum Doppler shift (0 = none), Rice factor rice_k
      gain       optional per-sample real gain (array, e.g. a level step between frames)
      sco_ppm    sample-clock offset of the receiver (resample(): the frame period becomes 196608 (1 + ppm e-6) samples)
      cfo        carrier offset in cycles/sample, phase0 in cycles
      snr_db     AWGN relative to unit signal power (None = noiseless)"""
    x = np.asarray(iq, np.complex64).astype(np.complex128)
    if paths:
        y = np.zeros_like(x)
        norm = np.sqrt(sum(abs(g) ** 2 for _, g in paths))
        for d, g in paths:
            d = int(d)
            y[d:] += (g / norm) * x[:x.size - d]
        x = y
    if fading_hz > 0.0:
        x = x * fading_gain(x.size, rng, fading_hz, rice_k)
    if gain is not None:
        x = x * np.asarray(gain, np.float64)
    if sco_ppm != 0.0:
        x = resample(x, sco_ppm).astype(np.complex128)
    n = np.arange(x.size)
    if cfo != 0.0 or phase0 != 0.0:
        x = x * np.exp(2j * np.pi * (cfo * n + phase0))
    if snr_db is not None:
        sigma = np.sqrt(0.5 * 10 ** (-snr_db / 10))
        x = x + sigma * (rng.standard_normal(x.size) + 1j * rng.standard_normal(x.size))
    return x.astype(np.complex64)


class Ensemble:
    """A small cyclic multiplex: FIC of random FIBs + one EEP subchannel + random filler.

    `n_frames` frames form a cyclically time-interleaved stream (4*n_frames CIFs) so that
    tiling the frames back to back is a valid continuous transmission."""

    def __init__(self, seed, n_frames=4, option=0, level=3, bitrate=64, start_cu=0, uep_index=None):
        rng = np.random.default_rng(seed)
        self.n_frames = n_frames
        if uep_index is None:
            self.mask, self.size_cu = eep_mask(option, level, bitrate)
        else:                                    # UEP: the table row fixes bit rate, level and size
            self.mask, self.size_cu = uep_mask(uep_index)
            bitrate = _UEP_TABLE[uep_index][0]
        self.start_cu = start_cu
        self.lf_bytes = bitrate * 3
        R = 4 * n_frames
        self.fibs = make_fibs(rng, 12 * n_frames).reshape(n_frames, 12, 32)
        self.msc_bytes = rng.integers(0, 256, size=(R, self.lf_bytes), dtype=np.uint8)
        coded = np.stack([msc_encode_lf(self.msc_bytes[r], self.mask) for r in range(R)])
        if coded.shape[1] < self.size_cu * 64:   # UEP padding bits (zeros) fill the sub-channel
            coded = np.concatenate([coded, np.zeros((R, self.size_cu * 64 - coded.shape[1]), np.uint8)], axis=1)
        tx = time_interleave(coded, cyclic=True)
        cifs = rng.integers(0, 2, size=(R, NB_CIF_BITS), dtype=np.uint8)
        a = start_cu * 64
        cifs[:, a:a + self.size_cu * 64] = tx
        self.frame_bits = np.zeros((n_frames, NB_FRAME_BITS), np.uint8)
        for f in range(n_frames):
            self.frame_bits[f, :NB_FIC_BITS] = fic_encode(self.fibs[f])
            self.frame_bits[f, NB_FIC_BITS:] = cifs[4 * f:4 * f + 4].ravel()

    def iq(self):
        """[n_frames][196608] complex64."""
        return np.stack([modulate_frame(self.frame_bits[f]) for f in range(self.n_frames)])


class MultiEnsemble:
    """Cyclic multiplex with several EEP subchannels: specs = [(option, level, bitrate, start_cu), ...]."""

    def __init__(self, seed, specs, n_frames=4):
        rng = np.random.default_rng(seed)
        self.n_frames = n_frames
        self.specs = specs
        R = 4 * n_frames
        self.fibs = make_fibs(rng, 12 * n_frames).reshape(n_frames, 12, 32)
        cifs = rng.integers(0, 2, size=(R, NB_CIF_BITS), dtype=np.uint8)
        self.masks, self.sizes, self.msc_bytes = [], [], []
        for (option, level, bitrate, start_cu) in specs:
            mask, size_cu = eep_mask(option, level, bitrate)
            data = rng.integers(0, 256, size=(R, bitrate * 3), dtype=np.uint8)
            coded = np.stack([msc_encode_lf(data[r], mask) for r in range(R)])
            cifs[:, start_cu * 64:(start_cu + size_cu) * 64] = time_interleave(coded, cyclic=True)
            self.masks.append(mask); self.sizes.append(size_cu); self.msc_bytes.append(data)
        self.frame_bits = np.zeros((n_frames, NB_FRAME_BITS), np.uint8)
        for f in range(n_frames):
            self.frame_bits[f, :NB_FIC_BITS] = fic_encode(self.fibs[f])
            self.frame_bits[f, NB_FIC_BITS:] = cifs[4 * f:4 * f + 4].ravel()

    def iq(self):
        return np.stack([modulate_frame(self.frame_bits[f]) for f in range(self.n_frames)])


# ----------------------------------------------------------------------------- DAB+ audio super-frame (TS 102 563)
def _gf256():
    exp = np.zeros(512, np.int64)
    log = np.zeros(256, np.int64)
    x = 1
    for i in range(255):
        exp[i] = x
        log[x] = i
        x <<= 1
        if x & 0x100:
            x ^= 0x11D
    exp[255:510] = exp[:255]
    return exp, log


_GF_EXP, _GF_LOG = _gf256()


def _gf_mul(a, b):
    return 0 if a == 0 or b == 0 else int(_GF_EXP[_GF_LOG[a] + _GF_LOG[b]])


def _rs_generator():
    g = [1]                                   # g[i] = coefficient of x^i
    for k in range(10):
        nx = [0] * (len(g) + 1)
        for i, c in enumerate(g):
            nx[i + 1] ^= c
            nx[i] ^= _gf_mul(c, int(_GF_EXP[k]))
        g = nx
    return g


_RS_G = _rs_generator()


def rs_parity(data110):
    """RS(120,110) parity: remainder of d(x) x^10 by prod_{k=0..9}(x - alpha^k), GF(2^8) / 0x11D."""
    rem = [0] * 10                            # rem[0] = highest degree
    for d in data110:
        fb = int(d) ^ rem[0]
        rem = [rem[j + 1] ^ _gf_mul(fb, _RS_G[9 - j]) for j in range(9)] + [_gf_mul(fb, _RS_G[0])]
    return np.array(rem, np.uint8)


def firecode16(data):
    crc = 0
    for byte in bytes(data):
        crc ^= byte << 8
        for _ in range(8):
            crc = ((crc << 1) ^ 0x782F) if (crc & 0x8000) else (crc << 1)
            crc &= 0xFFFF
    return crc


def build_superframe(rng, bitrate, dac_rate=1, sbr=0, channel_mode=1, ps=0, cuts=None):
    """-> (sf uint8[120*s], au_start list with the end appended, list of AU payloads incl. CRC).
    cuts: the start addresses of access units 1 .. n-1 (default: drawn at random)."""
    s = bitrate // 8
    size = 110 * s
    num_aus = {(0, 1): 2, (1, 1): 3, (0, 0): 4, (1, 0): 6}[(dac_rate, sbr)]
    first = {2: 5, 3: 6, 4: 8, 6: 11}[num_aus]
    if cuts is None:
        cuts = np.sort(rng.choice(np.arange(first + 8, size - 8, 4), num_aus - 1, replace=False))
    assert len(cuts) == num_aus - 1 and all(first + 3 <= c for c in cuts) and all(b - a >= 3 for a, b in zip(cuts, list(cuts[1:]) + [size]))
    starts = [first] + [int(c) for c in cuts] + [size]
    sf = np.zeros(120 * s, np.uint8)
    sf[2] = (dac_rate << 6) | (sbr << 5) | (channel_mode << 4) | (ps << 3)
    bits = []
    for a in range(1, num_aus):
        bits += [(starts[a] >> (11 - b)) & 1 for b in range(12)]
    bits += [0] * ((-len(bits)) % 8)
    hdr = np.packbits(np.array(bits, np.uint8)) if bits else np.zeros(0, np.uint8)
    sf[3:3 + hdr.size] = hdr
    aus = []
    for a in range(num_aus):
        body = rng.integers(0, 256, starts[a + 1] - starts[a] - 2, dtype=np.uint8)
        c = crc16(body)
        au = np.concatenate([body, np.array([c >> 8, c & 0xFF], np.uint8)])
        sf[starts[a]:starts[a + 1]] = au
        aus.append(au)
    fc = firecode16(sf[2:11])
    sf[0], sf[1] = fc >> 8, fc & 0xFF
    for j in range(s):
        sf[size + j::s] = rs_parity(sf[j:size:s])
    return sf, starts, aus


# ----------------------------------------------------------------------------- FIGs (EN 300 401 clauses 5.2, 6, 8)
# What a multiplex says about itself in the FIC: enough for a receiver to find and decode its audio services
# without being told anything (FIG 0/0 ensemble, 0/1 sub-channel organisation, 0/2 service organisation,
# 1/0 ensemble label, 1/1 service labels).
ASCTY_DAB = 0          # MPEG-1/2 layer II
ASCTY_DABPLUS = 63     # HE-AAC v2 super-frames (TS 102 563)


def _bits(*fields):
    """fields: (value, width) ... MSB first -> bytes (total width must be a multiple of 8)."""
    v, n = 0, 0
    for val, w in fields:
        assert 0 <= val < (1 << w), (val, w)
        v = (v << w) | val
        n += w
    assert n % 8 == 0
    return bytes((v >> (8 * (n // 8 - 1 - i))) & 0xFF for i in range(n // 8))


def fig0(ext, body, cn=0, oe=0, pd=0):
    data = _bits((cn, 1), (oe, 1), (pd, 1), (ext, 5)) + body
    assert len(data) <= 29
    return _bits((0, 3), (len(data), 5)) + data


def fig0_0(eid, cif_count, change=0, alarm=0):
    return fig0(0, _bits((eid, 16), (change, 2), (alarm, 1), ((cif_count // 250) % 20, 5), (cif_count % 250, 8)))


def fig0_1(subchannels):
    """subchannels: dicts {id, start, option, level, size} (long form, EEP) or {id, start, uep_index} (short form)."""
    body = b""
    for sc in subchannels:
        if "uep_index" in sc:
            body += _bits((sc["id"], 6), (sc["start"], 10), (0, 1), (0, 1), (sc["uep_index"], 6))
        else:
            body += _bits((sc["id"], 6), (sc["start"], 10), (1, 1), (sc["option"], 3), (sc["level"] - 1, 2), (sc["size"], 10))
    return fig0(1, body)


def fig0_2(services):
    """services: dicts {sid (16 bit), components: [{subchannel, ascty, primary}]} (programme services, MSC stream audio)."""
    body = b""
    for sv in services:
        comps = sv["components"]
        body += _bits((sv["sid"], 16), (0, 1), (0, 3), (len(comps), 4))
        for c in comps:
            body += _bits((0, 2), (c["ascty"], 6), (c["subchannel"], 6), (1 if c.get("primary", True) else 0, 1), (0, 1))
    return fig0(2, body)


def fig0_5(entries):
    """Service component language, short form: [(subchannel_id, language), ...] (clause 8.1.2)."""
    return fig0(5, b"".join(_bits((0, 1), (0, 1), (scid, 6), (lang, 8)) for scid, lang in entries))


def fig0_6(lsn, ids, idlq=0, active=1, hard=1, ils=0, pd=0):
    """Service linking (clause 8.1.15): one linkage set with an id list.  idlq 0 = DAB SIds, 1 = RDS PI codes,
    3 = DRM service ids; ils = 1: every id is (ECC, id) given as a 24-bit number; pd = 1: 32-bit SIds."""
    width = 32 if pd else (24 if ils else 16)
    body = _bits((1, 1), (active, 1), (hard, 1), (ils, 1), (lsn, 12), (0, 1), (idlq, 2), (0, 1), (len(ids), 4))
    body += b"".join(_bits((i, width)) for i in ids)
    return fig0(6, body, pd=pd)


def fig0_8(entries, pd=0):
    """Service component global definition, short form: [(sid, scids, subchannel_id), ...] (clause 6.3.5)."""
    w = 32 if pd else 16
    return fig0(8, b"".join(_bits((sid, w), (0, 1), (0, 3), (scids, 4), (0, 1), (0, 1), (scid, 6)) for sid, scids, scid in entries), pd=pd)


def fig0_9(ecc, lto_half_hours, inter_table=1):
    """Country, LTO and international table (clause 8.1.3.2); lto in half hours, negative = west of UTC."""
    return fig0(9, _bits((0, 1), (0, 1), (1 if lto_half_hours < 0 else 0, 1), (abs(lto_half_hours), 5), (ecc, 8), (inter_table, 8)))


def fig0_17(entries):
    """Programme type: [(sid, int_code, language or None), ...] (clause 8.1.5)."""
    body = b""
    for sid, code, lang in entries:
        body += _bits((sid, 16), (0, 1), (0, 1), (0 if lang is None else 1, 1), (0, 1), (0, 4))
        if lang is not None:
            body += _bits((lang, 8))
        body += _bits((0, 3), (code, 5))
    return fig0(17, body)


def fig0_21(entries, oe=0):
    """Frequency information (clause 8.1.8), one block: entries [(id, rm, continuity, frequencies in Hz), ...] with
    rm 0 = DAB ensemble (16 kHz steps), 8 = FM with RDS (87.5 MHz + 100 kHz steps), 6 = DRM (id is 24 bits, kHz)."""
    blk = b""
    for ident, rm, cont, freqs in entries:
        if rm == 0:
            fl = b"".join(_bits((0, 5), (f // 16000, 19)) for f in freqs)
        elif rm == 8:
            fl = bytes((f - 87500000) // 100000 for f in freqs)
        else:
            fl = bytes([ident >> 16]) + b"".join(_bits((0, 1), (f // 1000, 15)) for f in freqs)
        assert len(fl) <= 7
        blk += _bits((ident & 0xFFFF, 16), (rm, 4), (cont, 1), (len(fl), 3)) + fl
    return fig0(21, _bits((0, 11), (len(blk), 5)) + blk, oe=oe)


def fig0_24(entries, pd=0):
    """Services in other ensembles: [(sid, [eid, ...]), ...] (clause 8.1.10.2)."""
    w = 32 if pd else 16
    return fig0(24, b"".join(_bits((sid, w), (0, 1), (0, 3), (len(eids), 4)) + b"".join(_bits((e, 16)) for e in eids)
                             for sid, eids in entries), pd=pd, oe=1)


def fig1_4(sid, scids, label, pd=0, charset=0):
    """Service component label (clause 8.1.14.3)."""
    text = label.encode("latin-1")[:16].ljust(16, b" ")
    data = _bits((charset, 4), (0, 1), (4, 3), (pd, 1), (0, 3), (scids, 4), (sid, 32 if pd else 16)) + text + _bits((0xFF00, 16))
    return _bits((1, 3), (len(data), 5)) + data


def fig1_5(sid32, label, charset=0):
    """Data service label: 32-bit service identifier (clause 8.1.14.2)."""
    text = label.encode("latin-1")[:16].ljust(16, b" ")
    data = _bits((charset, 4), (0, 1), (5, 3), (sid32, 32)) + text + _bits((0xFF00, 16))
    return _bits((1, 3), (len(data), 5)) + data


def mjd(year, month, day):
    """Modified Julian date of a calendar date (inverse of the standard's annex formula)."""
    a = (14 - month) // 12
    y, m = year + 4800 - a, month + 12 * a - 3
    jdn = day + (153 * m + 2) // 5 + 365 * y + y // 4 - y // 100 + y // 400 - 32045
    return jdn - 2400001


def fig0_10(year, month, day, hours, minutes, seconds=None, milliseconds=0):
    """Date and time (clause 8.1.3.1): short form without seconds, long form with seconds and milliseconds."""
    long_form = seconds is not None
    body = _bits((0, 1), (mjd(year, month, day), 17), (0, 1), (0, 1), (1 if long_form else 0, 1), (hours, 5), (minutes, 6))
    if long_form:
        body += _bits((seconds, 6), (milliseconds, 10))
    return fig0(10, body)


def fig1(ext, ident, label, charset=0):
    text = label.encode("latin-1")[:16].ljust(16, b" ")
    data = _bits((charset, 4), (0, 1), (ext, 3), (ident, 16)) + text + _bits((0xFF00, 16))
    return _bits((1, 3), (len(data), 5)) + data


def pack_fibs(figs):
    """Greedy packing of FIGs (bytes) into 30-byte FIB data fields (end marker 0xFF, zero padding) + CRC."""
    fibs, cur = [], b""
    for f in figs:
        assert len(f) <= 30
        if len(cur) + len(f) > 30:
            fibs.append(cur)
            cur = b""
        cur += f
    if cur:
        fibs.append(cur)
    out = np.zeros((len(fibs), 32), np.uint8)
    for i, d in enumerate(fibs):
        d = d + (b"\xff" if len(d) < 30 else b"")
        d = d.ljust(30, b"\x00")
        c = crc16(d)
        out[i] = np.frombuffer(d + bytes([c >> 8, c & 0xFF]), np.uint8)
    return out


class ServiceEnsemble:
    """A cyclic multiplex that describes itself: DAB+ services on EEP sub-channels carrying real super-frames,
    FIC made of FIG 0/0, 0/1, 0/2, 1/0, 1/1.  services = [(label, sid, subchannel_id, option, level, bitrate,
    start_cu), ...].  n_frames must be a multiple of 5 so that whole super-frames (5 logical frames) tile."""

    def __init__(self, seed, services, n_frames=5, eid=0xC181, label="Synth Ensemble", dab_services=()):
        """dab_services: DAB (MPEG layer II) services on UEP sub-channels, [(label, sid, subchannel_id, uep_index,
        start_cu), ...]; every logical frame of such a sub-channel is one layer-II frame (header + random body)."""
        assert n_frames % 5 == 0
        rng = np.random.default_rng(seed)
        self.n_frames, self.eid, self.label, self.services = n_frames, eid, label, services
        R = 4 * n_frames
        cifs = rng.integers(0, 2, size=(R, NB_CIF_BITS), dtype=np.uint8)
        self.masks, self.sizes, self.msc_bytes, self.superframes, self.aus = [], [], [], [], []
        self.subchannels = []
        for (lab, sid, scid, option, level, bitrate, start_cu) in services:
            mask, size_cu = eep_mask(option, level, bitrate)
            sfs, aus = [], []
            for _ in range(R // 5):
                sf, starts, au = build_superframe(rng, bitrate, dac_rate=1, sbr=1, channel_mode=1, ps=0)
                sfs.append(sf); aus.append(au)
            data = np.concatenate(sfs).reshape(R, bitrate * 3)
            coded = np.stack([msc_encode_lf(data[r], mask) for r in range(R)])
            cifs[:, start_cu * 64:(start_cu + size_cu) * 64] = time_interleave(coded, cyclic=True)
            self.masks.append(mask); self.sizes.append(size_cu); self.msc_bytes.append(data)
            self.superframes.append(sfs); self.aus.append(aus)
            self.subchannels.append({"id": scid, "start": start_cu, "option": option, "level": level, "size": size_cu})
        self.dab_services, self.mp2_frames = list(dab_services), []
        for (lab, sid, scid, uidx, start_cu) in self.dab_services:
            mask, size_cu = uep_mask(uidx)
            bitrate = _UEP_TABLE[uidx][0]
            data = rng.integers(0, 256, size=(R, bitrate * 3), dtype=np.uint8)
            data[:, 0], data[:, 1] = 0xFF, 0xFD                     # sync, MPEG-1, layer II, no CRC
            data[:, 2] = (data[:, 2] & 0xF0) | 0x04                 # 48 kHz
            data[:, 3] &= 0x3F                                      # stereo
            coded = np.stack([msc_encode_lf(data[r], mask) for r in range(R)])
            coded = np.concatenate([coded, np.zeros((R, size_cu * 64 - coded.shape[1]), np.uint8)], axis=1)
            cifs[:, start_cu * 64:(start_cu + size_cu) * 64] = time_interleave(coded, cyclic=True)
            self.mp2_frames.append(data)
            self.subchannels.append({"id": scid, "start": start_cu, "uep_index": uidx})
        sv = [{"sid": sid, "components": [{"subchannel": scid, "ascty": ASCTY_DABPLUS}]}
              for (lab, sid, scid, *_rest) in services]
        sv += [{"sid": sid, "components": [{"subchannel": scid, "ascty": ASCTY_DAB}]}
               for (lab, sid, scid, *_rest) in self.dab_services]
        labels = [fig1(0, eid, label)] + [fig1(1, sid, lab) for (lab, sid, *_rest) in services]
        labels += [fig1(1, sid, lab) for (lab, sid, *_rest) in self.dab_services]
        # what else a multiplex says about itself (one of these per CIF, in rotation with the labels): country and
        # local time, component identifiers and labels, programme types, languages, a linkage set with its FM
        # alternative and that station's frequencies, the neighbouring ensemble carrying the first service
        all_sv = [(lab, sid, scid) for (lab, sid, scid, *_r) in services] + [(lab, sid, scid) for (lab, sid, scid, *_r) in self.dab_services]
        first_sid = all_sv[0][1]
        labels += [fig0_9(0xE1, 2, 1),
                   fig0_8([(sid, 0, scid) for (_l, sid, scid) in all_sv[:5]]),
                   fig0_17([(sid, 1 + k % 30, 0x09 if k == 0 else None) for k, (_l, sid, _s) in enumerate(all_sv[:5])]),
                   fig0_5([(scid, 0x09 + k) for k, (_l, _sid, scid) in enumerate(all_sv[:8])]),
                   fig0_6(0x123, [first_sid], idlq=0), fig0_6(0x123, [0xC479], idlq=1),
                   fig0_21([(0xC479, 8, 1, [98300000, 101100000]), (0xC182, 0, 0, [225648000])]),
                   fig0_24([(first_sid, [0xC182])])]
        labels += [fig1_4(sid, 0, (lab + " main")[:16]) for (lab, sid, _s) in all_sv[:3]]
        self.fibs = np.zeros((n_frames, 12, 32), np.uint8)
        k = 0
        for f in range(n_frames):
            for c in range(4):
                figs = [fig0_0(eid, 4 * f + c)]
                figs += [fig0_1(self.subchannels[i:i + 6]) for i in range(0, len(self.subchannels), 6)]
                figs += [fig0_2(sv[i:i + 5]) for i in range(0, len(sv), 5)]
                figs.append(labels[k % len(labels)]); k += 1
                if c == 3:                                    # once per frame: date and time, advancing 96 ms
                    ms = 96 * f
                    figs.append(fig0_10(2024, 2, 29, 23, 59, 58 + ms // 1000 if ms < 2000 else 59, ms % 1000))
                fibs = pack_fibs(figs)
                assert len(fibs) <= 3, "too many FIGs for one CIF's three FIBs"
                pad = pack_fibs([fig0_0(eid, 4 * f + c)])
                block = np.concatenate([fibs] + [pad] * (3 - len(fibs)))
                self.fibs[f, 3 * c:3 * c + 3] = block
        self.frame_bits = np.zeros((n_frames, NB_FRAME_BITS), np.uint8)
        for f in range(n_frames):
            self.frame_bits[f, :NB_FIC_BITS] = fic_encode(self.fibs[f])
            self.frame_bits[f, NB_FIC_BITS:] = cifs[4 * f:4 * f + 4].ravel()

    def iq(self):
        return np.stack([modulate_frame(self.frame_bits[f]) for f in range(self.n_frames)])
