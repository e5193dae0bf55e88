"""fig_oracle.py -- CPU restatement of FIB -> FIG -> database parsing, TEST INFRASTRUCTURE ONLY.

Stands behind what the reference's BasicRadio does with decoded FIBs (FIG parser + database updater in the absent
vendor/DAB-Radio sub-module; the GUI reads the result through radio.GetDatabase(),
/root/reference/src/render_radio_block.cpp:239-306, 490-752).  PARITY UNPINNED: restated from ETSI EN 300 401
clauses 5.2 (FIB / FIG structure), 6.2.1 (FIG 0/1), 6.3.1 (FIG 0/2), 6.4 (FIG 0/0), 8.1.3.1 (FIG 0/10),
8.1.13-14 (FIG 1/0, 1/1) from
memory; pinned only against this repo's own transmitter.  A field is written once (first value wins), as the
host-side updater does.

Only the product's tests import this file."""


def crc16(data):
    crc = 0xFFFF
    for byte in bytes(data):
        crc ^= byte << 8
        for _ in range(8):
            crc = ((crc << 1) ^ 0x1021) if (crc & 0x8000) else (crc << 1)
            crc &= 0xFFFF
    return crc ^ 0xFFFF


def esc(label):
    """Printable form of a label (any byte may arrive in a damaged or hostile FIB)."""
    return "".join(c if (0x20 <= ord(c) < 0x7F and c not in "\\[]") else "\\x%02X" % ord(c) for c in label)


class Database:
    def __init__(self):
        self.ensemble = {"id": None, "label": "", "cif_count": None}
        self.subchannels = {}     # id -> {start_address, length, is_uep, uep_prot_index, eep_type, eep_prot_level}
        self.services = {}        # sid -> {label, components: [{subchannel_id, transport_mode, audio_service_type, is_primary}]}
        self.datetime = None      # (year, month, day, hours, minutes, seconds, milliseconds) of the last FIG 0/10

    def lines(self):
        """Canonical text form, the same the host-side test program prints."""
        out = []
        e = self.ensemble
        if e["id"] is not None:
            out.append("ensemble id=%04X label=[%s]" % (e["id"], esc(e["label"])))
        for k in sorted(self.subchannels):
            s = self.subchannels[k]
            out.append("subchannel id=%d start=%d length=%d uep=%d uep_index=%d eep_type=%d eep_level=%d" % (
                k, s["start_address"], s["length"], int(s["is_uep"]), s["uep_prot_index"], s["eep_type"], s["eep_prot_level"]))
        for k in sorted(self.services):
            sv = self.services[k]
            out.append("service id=%04X label=[%s]" % (k, esc(sv["label"])))
            for c in sv["components"]:
                out.append("component service=%04X subchannel=%d tmid=%d ascty=%d primary=%d" % (
                    k, c["subchannel_id"], c["transport_mode"], c["audio_service_type"], int(c["is_primary"])))
        if self.datetime and self.datetime[0]:
            out.append("datetime %04d-%02d-%02d %02d:%02d:%02d.%03d cif=%d" % (self.datetime + (self.ensemble["cif_count"] or 0,)))
        return out


def mjd_to_ymd(mjd):
    yp = int((mjd - 15078.2) / 365.25)
    mp = int((mjd - 14956.1 - int(yp * 365.25)) / 30.6001)
    day = mjd - 14956 - int(yp * 365.25) - int(mp * 30.6001)
    k = 1 if mp in (14, 15) else 0
    return 1900 + yp + k, mp - 1 - 12 * k, day


def _fig0_10(d, db):
    if len(d) < 4:
        return
    w = int.from_bytes(bytes(d[:4]), "big")
    y, m, dd = mjd_to_ymd((w >> 14) & 0x1FFFF)
    hours, minutes, sec, ms = (w >> 6) & 0x1F, w & 0x3F, 0, 0
    if (w >> 11) & 1 and len(d) >= 6:
        sec, ms = d[4] >> 2, ((d[4] & 3) << 8) | d[5]
    if hours < 24 and minutes < 60 and sec < 61:
        db.datetime = (y, m, dd, hours, minutes, sec, ms)


def _fig0_0(d, db):
    if len(d) < 4:
        return
    if db.ensemble["id"] is None:
        db.ensemble["id"] = (d[0] << 8) | d[1]
    db.ensemble["cif_count"] = (d[2] & 0x1F) * 250 + d[3]


def _fig0_1(d, db):
    i = 0
    while i + 3 <= len(d):
        scid = d[i] >> 2
        start = ((d[i] & 3) << 8) | d[i + 1]
        if d[i + 2] & 0x80:                       # long form: EEP
            if i + 4 > len(d):
                return
            option = (d[i + 2] >> 4) & 7
            if option > 1:                        # reserved option: not an entry a receiver can use
                i += 4
                continue
            level = (d[i + 2] >> 2) & 3          # 0..3 as transmitted (the GUI prints level + 1)
            size = ((d[i + 2] & 3) << 8) | d[i + 3]
            db.subchannels.setdefault(scid, {"start_address": start, "length": size, "is_uep": False,
                                             "uep_prot_index": 0, "eep_type": option, "eep_prot_level": level})
            i += 4
        else:                                     # short form: index into the UEP table (size not restated here)
            db.subchannels.setdefault(scid, {"start_address": start, "length": 0, "is_uep": True,
                                             "uep_prot_index": d[i + 2] & 0x3F, "eep_type": 0, "eep_prot_level": 0})
            i += 3


def _fig0_2(d, pd, db):
    i = 0
    idlen = 4 if pd else 2
    while i + idlen + 1 <= len(d):
        sid = int.from_bytes(bytes(d[i:i + idlen]), "big")
        n = d[i + idlen] & 0x0F
        i += idlen + 1
        if i + 2 * n > len(d):
            return
        sv = db.services.setdefault(sid, {"label": "", "components": []})
        for _ in range(n):
            b0, b1 = d[i], d[i + 1]
            i += 2
            tmid = b0 >> 6
            if tmid != 0:                         # only MSC stream audio is followed
                continue
            comp = {"subchannel_id": b1 >> 2, "transport_mode": 0, "audio_service_type": b0 & 0x3F,
                    "is_primary": bool(b1 & 2)}
            if not any(c["subchannel_id"] == comp["subchannel_id"] for c in sv["components"]):
                sv["components"].append(comp)


def _label(d):
    return bytes(d[:16]).decode("latin-1").rstrip(" ")


def parse_fib(fib, db):
    """fib: 32 bytes (30 data + CRC).  Returns False (and changes nothing) when the CRC fails."""
    fib = bytes(fib)
    if len(fib) != 32 or crc16(fib[:30]) != ((fib[30] << 8) | fib[31]):
        return False
    d = fib[:30]
    i = 0
    while i < 30:
        hdr = d[i]
        if hdr == 0xFF:
            break
        ftype, flen = hdr >> 5, hdr & 0x1F
        if flen == 0 or i + 1 + flen > 30:
            break
        body = d[i + 1:i + 1 + flen]
        i += 1 + flen
        if ftype == 0:
            pd, ext = (body[0] >> 5) & 1, body[0] & 0x1F
            if ext == 0:
                _fig0_0(body[1:], db)
            elif ext == 1:
                _fig0_1(body[1:], db)
            elif ext == 2:
                _fig0_2(body[1:], pd, db)
            elif ext == 10:
                _fig0_10(body[1:], db)
        elif ftype == 1 and flen >= 21:
            ext = body[0] & 7
            ident = (body[1] << 8) | body[2]
            if ext == 0:
                if not db.ensemble["label"]:
                    db.ensemble["label"] = _label(body[3:19])
                if db.ensemble["id"] is None:
                    db.ensemble["id"] = ident
            elif ext == 1:
                sv = db.services.setdefault(ident, {"label": "", "components": []})
                if not sv["label"]:
                    sv["label"] = _label(body[3:19])
    return True


def parse_fibs(fibs):
    db = Database()
    for f in fibs:
        parse_fib(f, db)
    return db
