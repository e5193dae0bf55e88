"""fig_oracle.py -- CPU restatement of FIB -> FIG -> database parsing, TEST INFRASTRUCTURE ONLY.

Stands behind what the reference's BasicRadio does with decoded FIBs (FIG parser + database updater in the absent
vendor/DAB-Radio sub-module; the GUI reads the result through radio.GetDatabase(),
/root/reference/src/render_radio_block.cpp:239-306, 490-752).  PARITY UNPINNED: restated from ETSI EN 300 401
clauses 5.2 (FIB / FIG structure), 6.2.1 (FIG 0/1), 6.3.1 (FIG 0/2), 6.3.5 (FIG 0/8), 6.4 (FIG 0/0), 8.1.2 (FIG 0/5),
8.1.3 (FIG 0/9, 0/10), 8.1.5 (FIG 0/17), 8.1.8 (FIG 0/21), 8.1.10.2 (FIG 0/24), 8.1.13-14 (FIG 1/0, 1/1, 1/4, 1/5),
8.1.15 (FIG 0/6) from memory; pinned only against this repo's own transmitter.  A field is written once (first value wins), as the
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


MAX_ENTITIES = 1024      # cap of every entity list (the host-side updater's MAX_ENTITIES)


def _entity(table, key, fresh):
    """table[key], created from `fresh` if new -- unless the table is full: then a scratch entity nobody lists."""
    if key in table:
        return table[key]
    if len(table) >= MAX_ENTITIES:
        return fresh
    table[key] = fresh
    return fresh


class Database:
    def __init__(self):
        self.n_components = 0     # over all services (the updater keeps one capped list)
        self.ensemble = {"id": None, "label": "", "cif_count": None}
        self.subchannels = {}     # id -> {start_address, length, is_uep, uep_prot_index, eep_type, eep_prot_level}
        self.services = {}        # sid -> {label, components: [{subchannel_id, transport_mode, audio_service_type, is_primary}]}
        self.datetime = None      # (year, month, day, hours, minutes, seconds, milliseconds) of the last FIG 0/10
        self.country = None       # (ecc, lto in tenths of an hour, international table id) from FIG 0/9
        self.links = {}           # lsn -> {active, hard, intl, service}
        self.fm = {}              # RDS PI -> {lsn, tc, freqs}
        self.drm = {}             # 24-bit id -> {lsn, tc, freqs}
        self.other = {}           # EId -> {cont, freqs, services}

    def lines(self):
        """Canonical text form, the same the host-side test program prints."""
        out = []
        e = self.ensemble
        if e["id"] is not None:
            out.append("ensemble id=%04X label=[%s]" % (e["id"], esc(e["label"])))
        if self.country is not None:
            out.append("ensemble_info ecc=%02X lto=%d inter_table=%d" % self.country)
        for k in sorted(self.subchannels):
            s = self.subchannels[k]
            out.append("subchannel id=%d start=%d length=%d uep=%d uep_index=%d eep_type=%d eep_level=%d" % (
                k, s["start_address"], s["length"], int(s["is_uep"]), s["uep_prot_index"], s["eep_type"], s["eep_prot_level"]))
        for k in sorted(self.services):
            sv = self.services[k]
            out.append("service id=%04X label=[%s] pty=%d lang=%d bits32=%d" % (
                k, esc(sv["label"]), sv.get("pty", -1), sv.get("lang", 0), int(sv.get("bits32", False))))
            for c in sv["components"]:
                out.append("component service=%04X subchannel=%d tmid=%d ascty=%d primary=%d scids=%d lang=%d label=[%s]" % (
                    k, c["subchannel_id"], c["transport_mode"], c["audio_service_type"], int(c["is_primary"]),
                    c.get("scids", -1), c.get("lang", 0), esc(c.get("label", ""))))
        for k in sorted(self.links):
            l = self.links[k]
            out.append("link lsn=%d active=%d hard=%d intl=%d service=%s" % (
                k, l["active"], l["hard"], l["intl"], "none" if l["service"] is None else "%04X" % l["service"]))
        for k in sorted(self.fm):
            m = self.fm[k]
            out.append("fm pi=%04X lsn=%d tc=%d freqs=%s" % (k, -1 if m["lsn"] is None else m["lsn"], int(m["tc"]),
                                                             ",".join(str(f) for f in m["freqs"])))
        for k in sorted(self.drm):
            m = self.drm[k]
            out.append("drm code=%06X lsn=%d tc=%d freqs=%s" % (k, -1 if m["lsn"] is None else m["lsn"], int(m["tc"]),
                                                                ",".join(str(f) for f in m["freqs"])))
        for k in sorted(self.other):
            o = self.other[k]
            out.append("other_ensemble id=%04X cont=%d freqs=%s services=%s" % (
                k, int(o["cont"]), ",".join(str(f) for f in o["freqs"]), ",".join("%04X" % v for v in o["services"])))
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
        sv = _entity(db.services, sid, {"label": "", "components": [], "bits32": bool(pd)})
        for _ in range(n):
            b0, b1 = d[i], d[i + 1]
            i += 2
            tmid = b0 >> 6
            if tmid != 0:                         # only MSC stream audio is followed
                continue
            comp = {"subchannel_id": b1 >> 2, "transport_mode": 0, "audio_service_type": b0 & 0x3F,
                    "is_primary": bool(b1 & 2)}
            if not any(c["subchannel_id"] == comp["subchannel_id"] for c in sv["components"]) and db.n_components < MAX_ENTITIES:
                sv["components"].append(comp)
                db.n_components += 1


def _add_unique(lst, x):
    if x not in lst and len(lst) < 64:
        lst.append(x)


def _components(db):
    for sid in db.services:
        for c in db.services[sid]["components"]:
            yield sid, c


def _fig0_5(d, db):
    i = 0
    while i < len(d):
        if d[i] & 0x80:                           # long form (packet-mode component): not followed
            if i + 3 > len(d):
                return
            i += 3
            continue
        if i + 2 > len(d):
            return
        fic, scid, lang = (d[i] >> 6) & 1, d[i] & 0x3F, d[i + 1]
        i += 2
        if fic:
            continue
        for _, c in _components(db):
            if c["subchannel_id"] == scid and c.get("lang", 0) == 0 and lang != 0:
                c["lang"] = lang


def _fig0_6(d, pd, db):
    i = 0
    while i + 2 <= len(d):
        w = (d[i] << 8) | d[i + 1]
        has_list, la, hard, ils, lsn = (w >> 15) & 1, (w >> 14) & 1, (w >> 13) & 1, (w >> 12) & 1, w & 0x0FFF
        i += 2
        if not has_list:
            continue
        if i + 1 > len(d):
            return
        idlq, count = (d[i] >> 5) & 3, d[i] & 0x0F
        i += 1
        idlen = 4 if pd else (3 if ils else 2)
        if i + idlen * count > len(d):
            return
        link = _entity(db.links, lsn, {"active": la, "hard": hard, "intl": ils, "service": None})
        for k in range(count):
            ident = int.from_bytes(bytes(d[i:i + idlen]), "big")
            i += idlen
            if pd or idlq == 0:
                if k == 0 and link["service"] is None:
                    link["service"] = ident if pd else ident & 0xFFFF
            elif idlq == 1:
                m = _entity(db.fm, ident & 0xFFFF, {"lsn": None, "tc": False, "freqs": []})
                if m["lsn"] is None:
                    m["lsn"] = lsn
            elif idlq == 3:
                m = _entity(db.drm, ident & 0xFFFFFF, {"lsn": None, "tc": False, "freqs": []})
                if m["lsn"] is None:
                    m["lsn"] = lsn


def _fig0_8(d, pd, db):
    idlen = 4 if pd else 2
    i = 0
    while i + idlen + 2 <= len(d):
        sid = int.from_bytes(bytes(d[i:i + idlen]), "big")
        ext, scids = d[i + idlen] >> 7, d[i + idlen] & 0x0F
        c = d[i + idlen + 1]
        long_form = c >> 7
        size = idlen + 2 + (1 if long_form else 0) + (1 if ext else 0)
        if i + size > len(d):
            return
        i += size
        if long_form or ((c >> 6) & 1):
            continue
        for csid, comp in _components(db):
            if csid == sid and comp["subchannel_id"] == (c & 0x3F) and "scids" not in comp:
                comp["scids"] = scids


def _fig0_9(d, db):
    if len(d) < 3:
        return
    mag = d[0] & 0x1F
    if db.country is None:
        db.country = (d[1], -5 * mag if (d[0] >> 5) & 1 else 5 * mag, d[2])


def _fig0_17(d, db):
    i = 0
    while i + 4 <= len(d):
        sid = (d[i] << 8) | d[i + 1]
        has_lang, has_cc = (d[i + 2] >> 5) & 1, (d[i + 2] >> 4) & 1
        size = 4 + has_lang + has_cc
        if i + size > len(d):
            return
        lang = d[i + 3] if has_lang else 0
        code = d[i + 3 + has_lang] & 0x1F
        i += size
        sv = db.services.get(sid)
        if sv is None:
            continue
        sv.setdefault("pty", code)
        if has_lang and sv.get("lang", 0) == 0 and lang != 0:
            sv["lang"] = lang


def _fig0_21(d, db):
    i = 0
    while i + 2 <= len(d):
        fi_len = d[i + 1] & 0x1F
        i += 2
        if i + fi_len > len(d):
            return
        blk = d[i:i + fi_len]
        i += fi_len
        j = 0
        while j + 3 <= fi_len:
            ident = (blk[j] << 8) | blk[j + 1]
            rm, cont, n = blk[j + 2] >> 4, (blk[j + 2] >> 3) & 1, blk[j + 2] & 7
            j += 3
            if j + n > fi_len:
                break
            fl = blk[j:j + n]
            j += n
            if rm == 0:
                o = _entity(db.other, ident, {"cont": False, "freqs": [], "services": []})
                o["cont"] = o["cont"] or bool(cont)
                for k in range(0, n - 2, 3):
                    _add_unique(o["freqs"], (((fl[k] & 7) << 16) | (fl[k + 1] << 8) | fl[k + 2]) * 16000)
            elif rm == 8:
                m = _entity(db.fm, ident, {"lsn": None, "tc": False, "freqs": []})
                m["tc"] = m["tc"] or bool(cont)
                for k in range(n):
                    _add_unique(m["freqs"], 87500000 + 100000 * fl[k])
            elif rm == 6:
                if n < 1:
                    continue
                m = _entity(db.drm, (fl[0] << 16) | ident, {"lsn": None, "tc": False, "freqs": []})
                m["tc"] = m["tc"] or bool(cont)
                for k in range(1, n - 1, 2):
                    _add_unique(m["freqs"], (((fl[k] & 0x7F) << 8) | fl[k + 1]) * 1000)


def _fig0_24(d, pd, db):
    idlen = 4 if pd else 2
    i = 0
    while i + idlen + 1 <= len(d):
        sid = int.from_bytes(bytes(d[i:i + idlen]), "big")
        count = d[i + idlen] & 0x0F
        i += idlen + 1
        if i + 2 * count > len(d):
            return
        for _ in range(count):
            o = _entity(db.other, (d[i] << 8) | d[i + 1], {"cont": False, "freqs": [], "services": []})
            _add_unique(o["services"], sid)
            i += 2


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
            elif ext == 5:
                _fig0_5(body[1:], db)
            elif ext == 6:
                _fig0_6(body[1:], pd, db)
            elif ext == 8:
                _fig0_8(body[1:], pd, db)
            elif ext == 9:
                _fig0_9(body[1:], db)
            elif ext == 10:
                _fig0_10(body[1:], db)
            elif ext == 17:
                _fig0_17(body[1:], db)
            elif ext == 21:
                _fig0_21(body[1:], db)
            elif ext == 24:
                _fig0_24(body[1:], pd, db)
        elif ftype == 1:
            ext = body[0] & 7
            if ext in (0, 1) and flen >= 21:
                ident = (body[1] << 8) | body[2]
                if ext == 0:
                    if db.ensemble["id"] is None:
                        db.ensemble["id"] = ident
                    if not db.ensemble["label"]:
                        db.ensemble["label"] = _label(body[3:19])
                else:
                    sv = _entity(db.services, ident, {"label": "", "components": []})
                    if not sv["label"]:
                        sv["label"] = _label(body[3:19])
            elif ext == 5 and flen >= 23:
                sv = db.services.get(int.from_bytes(bytes(body[1:5]), "big"))
                if sv is not None and not sv["label"]:
                    sv["label"] = _label(body[5:21])
            elif ext == 4 and flen >= 2:
                cpd, scids = body[1] >> 7, body[1] & 0x0F
                idlen = 4 if cpd else 2
                if flen >= 2 + idlen + 18:
                    sid = int.from_bytes(bytes(body[2:2 + idlen]), "big")
                    for csid, comp in _components(db):
                        if csid == sid and comp.get("scids", -1) == scids and not comp.get("label"):
                            comp["label"] = _label(body[2 + idlen:18 + idlen])
    return True


def parse_fibs(fibs):
    db = Database()
    for f in fibs:
        parse_fib(f, db)
    return db
