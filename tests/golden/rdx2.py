"""Minimal reader for R's RDX2 / XDR serialization (``save()`` files, serialize v2/v3).

Dev/test tooling only: used once by ``make_example_fixture.py`` to dump the
reference's bundled *data files* (``data/example_sce.rda``) into small ``.npz``
fixtures.  It is NOT part of the product path and reads no reference source code.

Format notes follow R's published serialization layout (``src/main/serialize.c``
in the R sources, documented in "R Internals" §1.8): a stream of items, each
introduced by a flags word (type in the low byte, object/attr/tag bits 8-10,
gp "levels" from bit 12), pairlists stored car/cdr, environments and symbols
added to a back-reference table.
"""
import bz2
import gzip
import lzma
import struct

import numpy as np

NILVALUE, GLOBALENV, UNBOUND, MISSINGARG, BASENS = 254, 253, 252, 251, 250
NAMESPACESXP, PACKAGESXP, PERSISTSXP, REFSXP = 249, 248, 247, 255
CLASSREF, GENERICREF, BCREPDEF, BCREPREF, EMPTYENV, BASEENV = 246, 245, 244, 243, 242, 241
ATTRLANGSXP, ATTRLISTSXP, ALTREP = 240, 239, 238
SYMSXP, LISTSXP, CLOSXP, ENVSXP, PROMSXP, LANGSXP = 1, 2, 3, 4, 5, 6
SPECIALSXP, BUILTINSXP, CHARSXP, LGLSXP, INTSXP, REALSXP = 7, 8, 9, 10, 13, 14
CPLXSXP, STRSXP, DOTSXP, VECSXP, EXPRSXP, BCODESXP = 15, 16, 17, 19, 20, 21
EXTPTRSXP, WEAKREFSXP, RAWSXP, S4SXP = 22, 23, 24, 25


class RObj:
    """A decoded R value: ``kind`` + python ``value`` + optional ``attr`` dict."""

    __slots__ = ("kind", "value", "attr", "tag")

    def __init__(self, kind, value=None, attr=None, tag=None):
        self.kind, self.value, self.attr, self.tag = kind, value, attr, tag

    def __repr__(self):
        v = self.value
        if isinstance(v, np.ndarray):
            v = f"array{v.shape}"
        elif isinstance(v, (list, dict)):
            v = f"{type(v).__name__}[{len(v)}]"
        return f"RObj({self.kind}, {v}, attr={list(self.attr) if self.attr else None})"

    def names(self):
        if self.attr and "names" in self.attr:
            return list(self.attr["names"].value)
        return None

    def get(self, name):
        """Element of a named list (VECSXP) or pairlist-as-dict."""
        if self.kind == "list":
            return self.value[self.names().index(name)]
        raise KeyError(name)


class _Reader:
    def __init__(self, buf):
        self.b = buf
        self.p = 0
        self.refs = []

    def i32(self):
        v = struct.unpack_from(">i", self.b, self.p)[0]
        self.p += 4
        return v

    def length(self):
        n = self.i32()
        if n == -1:
            hi, lo = self.i32(), self.i32()
            n = (hi << 32) + (lo & 0xFFFFFFFF)
        return n

    def raw(self, n):
        v = bytes(self.b[self.p:self.p + n])
        self.p += n
        return v

    def attr_dict(self, pl):
        if pl is None or pl.kind == "NULL":
            return None
        return {k: v for k, v in pl.value}

    def pairlist_to_items(self, flags):
        """Read a LISTSXP-like chain iteratively; return list of (tag, value)."""
        items = []
        attr = None
        while True:
            t = flags & 0xFF
            if t not in (LISTSXP, LANGSXP, CLOSXP, PROMSXP, DOTSXP):
                # cdr was not a pairlist node (e.g. NILVALUE terminator)
                tail = self.item(flags)
                return items, attr, tail
            has_attr, has_tag = flags & (1 << 9), flags & (1 << 10)
            a = self.item() if has_attr else None
            if a is not None and attr is None:
                attr = self.attr_dict(a)
            tag = self.item() if has_tag else None
            car = self.item()
            tagname = tag.value if tag is not None and tag.kind == "symbol" else None
            items.append((tagname, car))
            flags = self.i32()

    def bclang(self, t, reps):
        if t == BCREPREF:
            return reps[self.i32()]
        if t in (BCREPDEF, LANGSXP, LISTSXP, ATTRLANGSXP, ATTRLISTSXP):
            pos = -1
            hasattr_ = False
            if t == BCREPDEF:
                pos = self.i32()
                t = self.i32()
            if t in (ATTRLANGSXP, ATTRLISTSXP):
                hasattr_ = True
            node = RObj("bclang", [])
            if pos >= 0:
                reps[pos] = node
            if hasattr_:
                self.item()
            self.item()  # tag
            car = self.bclang(self.i32(), reps)
            cdr = self.bclang(self.i32(), reps)
            node.value = [car, cdr]
            return node
        return self.item()

    def bc1(self, reps):
        code = self.item()
        n = self.i32()
        consts = []
        for _ in range(n):
            t = self.i32()
            if t == BCODESXP:
                consts.append(self.bc1(reps))
            elif t in (LANGSXP, LISTSXP, BCREPDEF, BCREPREF, ATTRLANGSXP, ATTRLISTSXP):
                consts.append(self.bclang(t, reps))
            else:
                consts.append(self.item())
        return RObj("bytecode", (code, consts))

    def item(self, flags=None):
        if flags is None:
            flags = self.i32()
        t = flags & 0xFF
        has_attr = flags & (1 << 9)
        if t == NILVALUE:
            return RObj("NULL")
        if t in (GLOBALENV, EMPTYENV, BASEENV, BASENS, UNBOUND, MISSINGARG):
            return RObj("special", t)
        if t == REFSXP:
            idx = flags >> 8
            if idx == 0:
                idx = self.i32()
            return self.refs[idx - 1]
        if t in (NAMESPACESXP, PACKAGESXP, PERSISTSXP):
            self.i32()
            n = self.i32()
            strs = [self.item().value for _ in range(n)]
            o = RObj("namespace", strs)
            self.refs.append(o)
            return o
        if t == SYMSXP:
            name = self.item().value
            o = RObj("symbol", name)
            self.refs.append(o)
            return o
        if t in (LISTSXP, LANGSXP, CLOSXP, PROMSXP, DOTSXP):
            items, attr, _tail = self.pairlist_to_items(flags)
            return RObj("pairlist", items, attr)
        if t == ENVSXP:
            self.i32()  # locked
            o = RObj("env", {})
            self.refs.append(o)
            enclos, frame, hashtab, attr = self.item(), self.item(), self.item(), self.item()
            vals = {}
            if frame.kind == "pairlist":
                vals.update({k: v for k, v in frame.value})
            if hashtab.kind == "list":
                for bucket in hashtab.value:
                    if bucket.kind == "pairlist":
                        vals.update({k: v for k, v in bucket.value})
            o.value = vals
            o.attr = self.attr_dict(attr)
            return o
        if t in (SPECIALSXP, BUILTINSXP):
            n = self.i32()
            return RObj("builtin", self.raw(n).decode())
        if t == CHARSXP:
            n = self.i32()
            if n == -1:
                return RObj("char", None)
            return RObj("char", self.raw(n).decode("utf-8", "replace"))
        if t in (LGLSXP, INTSXP):
            n = self.length()
            v = np.frombuffer(self.raw(4 * n), dtype=">i4").astype(np.int32)
            o = RObj("logical" if t == LGLSXP else "int", v)
        elif t == REALSXP:
            n = self.length()
            o = RObj("real", np.frombuffer(self.raw(8 * n), dtype=">f8").astype(np.float64))
        elif t == CPLXSXP:
            n = self.length()
            o = RObj("complex", np.frombuffer(self.raw(16 * n), dtype=">c16").astype(np.complex128))
        elif t == STRSXP:
            n = self.length()
            o = RObj("str", [self.item().value for _ in range(n)])
        elif t in (VECSXP, EXPRSXP):
            n = self.length()
            o = RObj("list", [self.item() for _ in range(n)])
        elif t == RAWSXP:
            n = self.length()
            o = RObj("raw", bytes(self.raw(n)))
        elif t == S4SXP:
            o = RObj("S4")
        elif t == BCODESXP:
            nreps = self.i32()
            o = self.bc1([None] * nreps)
        elif t == EXTPTRSXP:
            o = RObj("extptr")
            self.refs.append(o)
            self.item()
            self.item()
        elif t == WEAKREFSXP:
            o = RObj("weakref")
            self.refs.append(o)
        else:
            raise NotImplementedError(f"SEXP type {t} at offset {self.p}")
        if has_attr:
            o.attr = self.attr_dict(self.item())
        return o


def read_rda(path):
    """Return {name: RObj} for an R ``save()`` file (bzip2/gzip/xz or plain)."""
    raw = open(path, "rb").read()
    for mod in (bz2, gzip, lzma):
        try:
            raw = mod.decompress(raw)
            break
        except Exception:
            continue
    assert raw[:5] == b"RDX2\n", raw[:8]
    assert raw[5:7] == b"X\n", "only XDR format supported"
    r = _Reader(memoryview(raw))
    r.p = 7
    version = r.i32()
    r.i32()
    r.i32()
    if version == 3:
        n = r.i32()
        r.raw(n)
    top = r.item()
    assert top.kind == "pairlist"
    return {k: v for k, v in top.value}


def matrix(o):
    """R matrix (column-major + dim attr) -> numpy [nrow, ncol]."""
    d = o.attr["dim"].value
    return np.asarray(o.value).reshape(int(d[1]), int(d[0])).T
