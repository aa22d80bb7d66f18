"""Minimal reader for gzip'd R serialisation (RDX2, XDR) -- enough for data/prostate.RData.

Used once, in this container, by make_golden.py to turn the reference's own data file
(/root/reference/data/prostate.RData, the data set behind BASELINE config 1) into the
CSV fixture tests/golden/prostate.csv.  Not product code.
"""
import gzip
import struct


class _Reader:
    def __init__(self, buf):
        self.b = buf
        self.o = 0
        self.syms = []

    def i32(self):
        v = struct.unpack_from(">i", self.b, self.o)[0]
        self.o += 4
        return v

    def f64s(self, n):
        v = struct.unpack_from(">%dd" % n, self.b, self.o)
        self.o += 8 * n
        return list(v)

    def i32s(self, n):
        v = struct.unpack_from(">%di" % n, self.b, self.o)
        self.o += 4 * n
        return list(v)

    def item(self):
        flags = self.i32()
        typ = flags & 0xFF
        has_attr = bool(flags & (1 << 9))
        has_tag = bool(flags & (1 << 10))
        if typ == 254:  # NILVALUE
            return None
        if typ == 255:  # back reference to a symbol
            return self.syms[(flags >> 8) - 1]
        if typ == 1:  # symbol
            name = self.item()
            self.syms.append(name)
            return name
        if typ == 9:  # CHARSXP
            n = self.i32()
            if n < 0:
                return None
            s = self.b[self.o:self.o + n].decode("latin1")
            self.o += n
            return s
        if typ == 2:  # pairlist -> list of (tag, value)
            out = []
            while True:
                attr = self.item() if has_attr else None  # noqa: F841
                tag = self.item() if has_tag else None
                out.append((tag, self.item()))
                flags = self.i32()
                typ = flags & 0xFF
                has_attr = bool(flags & (1 << 9))
                has_tag = bool(flags & (1 << 10))
                if typ == 254:
                    return out
                if typ != 2:
                    raise ValueError("unexpected pairlist continuation type %d" % typ)
        if typ in (13, 10):  # INTSXP / LGLSXP
            val = self.i32s(self.i32())
        elif typ == 14:  # REALSXP
            val = self.f64s(self.i32())
        elif typ == 16:  # STRSXP
            val = [self.item() for _ in range(self.i32())]
        elif typ == 19:  # VECSXP
            val = [self.item() for _ in range(self.i32())]
        else:
            raise ValueError("unsupported SEXP type %d" % typ)
        attrs = dict(self.item()) if has_attr else {}
        return {"value": val, "attr": attrs} if attrs else val


def read_rdata(path):
    raw = gzip.open(path, "rb").read()
    if not raw.startswith(b"RDX2\nX\n"):
        raise ValueError("not an RDX2/XDR file")
    r = _Reader(raw)
    r.o = 7
    r.i32s(3)  # format version, writer version, min reader version
    return dict(r.item())


def read_dataframe(path, name):
    obj = read_rdata(path)[name]
    cols = obj["value"]
    names = obj["attr"]["names"]
    cols = [c["value"] if isinstance(c, dict) else c for c in cols]
    return names, cols
