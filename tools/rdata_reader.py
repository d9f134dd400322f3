"""Minimal reader for R's XDR serialisation (version 2/3), enough to pull the numeric columns out of
the reference's bundled data/ExomeCount.RData (a GRanges holding chr1 exon coordinates and four
count columns).  Used only by tests/golden/make_golden.py to turn that data file into a small numpy
fixture; nothing here runs in the product or on the GPU box.
"""
import gzip
import struct


class _Reader:
    def __init__(self, buf):
        self.b = buf
        self.i = 0
        self.refs = []

    def int(self):
        v = struct.unpack_from(">i", self.b, self.i)[0]
        self.i += 4
        return v

    def dbl(self):
        v = struct.unpack_from(">d", self.b, self.i)[0]
        self.i += 8
        return v

    def raw(self, n):
        v = self.b[self.i:self.i + n]
        self.i += n
        return v

    def item(self):
        flags = self.int()
        t = flags & 0xFF
        has_attr = bool(flags & 0x200)
        has_tag = bool(flags & 0x400)
        is_obj = bool(flags & 0x100)
        if t == 254:   # NILVALUE
            return None
        if t == 253:   # global env
            return "<globalenv>"
        if t == 242:   # base namespace / empty env etc.
            return "<emptyenv>"
        if t in (241, 247, 248, 249, 250, 251, 252):
            return "<special%d>" % t
        if t == 255:   # reference
            idx = flags >> 8
            if idx == 0:
                idx = self.int()
            return self.refs[idx - 1]
        if t == 1:     # SYMSXP
            name = self.item()
            self.refs.append(name)
            return name
        if t == 243 or t == 244:  # namespace / package spec: STRSXP follows (persistent names)
            self.int()
            n = self.int()
            v = [self.item() for _ in range(n)]
            self.refs.append(("<ns>", v))
            return ("<ns>", v)
        if t in (2, 6):  # LISTSXP / LANGSXP: pairlist node
            out = []
            while True:
                attr = self.item() if has_attr else None
                tag = self.item() if has_tag else None
                car = self.item()
                out.append((tag, car))
                # cdr
                flags = self.int()
                t2 = flags & 0xFF
                if t2 == 254:
                    break
                if t2 not in (2, 6):
                    self.i -= 4
                    out.append(("<cdr>", self.item()))
                    break
                has_attr = bool(flags & 0x200)
                has_tag = bool(flags & 0x400)
            return out
        if t == 9:     # CHARSXP
            n = self.int()
            if n == -1:
                return None
            return self.raw(n).decode("latin1")
        if t == 25:    # S4SXP
            attrs = self.item() if has_attr else []
            return {"<S4>": True, "attr": dict((k, v) for k, v in attrs)}
        if t == 4:     # ENVSXP
            self.int()
            env = {"<env>": True}
            self.refs.append(env)
            env["enclos"] = self.item(); env["frame"] = self.item(); env["hash"] = self.item(); env["attr"] = self.item()
            return env
        n = None
        if t in (10, 13):   # LGLSXP, INTSXP
            n = self.int()
            v = list(struct.unpack_from(">%di" % n, self.b, self.i)); self.i += 4 * n
        elif t == 14:       # REALSXP
            n = self.int()
            v = list(struct.unpack_from(">%dd" % n, self.b, self.i)); self.i += 8 * n
        elif t == 16:       # STRSXP
            n = self.int()
            v = [self.item() for _ in range(n)]
        elif t in (19, 20):  # VECSXP, EXPRSXP
            n = self.int()
            v = [self.item() for _ in range(n)]
        elif t == 24:       # RAWSXP
            n = self.int()
            v = self.raw(n)
        else:
            raise ValueError("unsupported SEXP type %d at offset %d" % (t, self.i))
        attrs = self.item() if has_attr else None
        if attrs or is_obj:
            return {"value": v, "attr": dict((k, a) for k, a in (attrs or []))}
        return v


def read_rdata(path):
    raw = gzip.open(path, "rb").read()
    assert raw[:5] == b"RDX2\n" or raw[:5] == b"RDX3\n", raw[:8]
    r = _Reader(raw)
    r.i = 5
    assert r.raw(2) == b"X\n"
    version = r.int(); r.int(); r.int()
    if version == 3:
        n = r.int(); r.raw(n)
    return r.item()


def unwrap(x):
    return x["value"] if isinstance(x, dict) and "value" in x else x


def exome_count(path):
    """Returns dict(start, width, Exome1..4, GC) as python lists from the bundled GRanges."""
    top = read_rdata(path)          # pairlist [(name, object)]
    name, obj = top[0]
    a = obj["attr"]
    ranges = a["ranges"]["attr"]
    md = a["elementMetadata"]["attr"]
    cols = unwrap(md["listData"])
    names = unwrap(md["listData"]["attr"]["names"]) if isinstance(md["listData"], dict) else None
    out = {"object": name, "start": unwrap(ranges["start"]), "width": unwrap(ranges["width"])}
    for k, v in zip(names, cols):
        out[k] = unwrap(v)
    return out
