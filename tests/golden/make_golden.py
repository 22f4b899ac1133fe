#!/usr/bin/env python3
"""Convert the reference's committed golden vectors (tests/**/*.gen.lua) into
compact JSON fixtures that travel with this repo (the GPU box has no
/root/reference).

The committed ``*.gen.lua`` files are the authoritative golden vectors: several
of the reference's numpy/scipy generators no longer run under scipy 1.15
(SURVEY.md section 0), so they are parsed, not regenerated.  Each number is kept as
the exact 8-decimal text the reference serialised (tests/generate.py:12,35-49),
re-parsed with Python ``float`` - no arithmetic happens here.

Run in the build container (needs /root/reference):

    python tests/golden/make_golden.py

Output: tests/golden/<spec>.json.gz, schema:
  BlockSpec : {"kind": "block", "block": name, "epsilon": str,
               "vectors": [{"desc", "args": [...], "inputs": [...], "outputs": [...]}]}
  RawSpec   : {"kind": "raw", "values": {name: value}}
where a sample vector is {"type": "ComplexFloat32"|"Float32"|..., "data": [...]}
(complex data as [[re, im], ...]).
"""
import gzip
import json
import os
import re
import struct
import sys

REF = os.environ.get("LUARADIO_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))

SPECS = [
    "blocks/signal/firfilter_spec",
    "blocks/signal/lowpassfilter_spec",
    "blocks/signal/highpassfilter_spec",
    "blocks/signal/bandpassfilter_spec",
    "blocks/signal/bandstopfilter_spec",
    "blocks/signal/frequencytranslator_spec",
    "blocks/signal/frequencydiscriminator_spec",
    "blocks/signal/downsampler_spec",
    "blocks/signal/iirfilter_spec",
    "blocks/signal/singlepolelowpassfilter_spec",
    "blocks/signal/fmdeemphasisfilter_spec",
    "blocks/signal/multiplyconjugate_spec",
    "blocks/signal/multiply_spec",
    "blocks/signal/add_spec",
    "blocks/signal/subtract_spec",
    "blocks/signal/multiplyconstant_spec",
    "blocks/signal/addconstant_spec",
    "blocks/signal/complexmagnitude_spec",
    "blocks/signal/complexphase_spec",
    "blocks/signal/complextoreal_spec",
    "blocks/signal/complextoimag_spec",
    "blocks/signal/complexconjugate_spec",
    "blocks/signal/realtocomplex_spec",
    "blocks/signal/absolutevalue_spec",
    "blocks/signal/delay_spec",
    "blocks/signal/agc_spec",
    "blocks/signal/powersquelch_spec",
    "blocks/signal/frequencymodulator_spec",
    "blocks/signal/pulsematchedfilter_spec",
    "blocks/signal/manchestermatchedfilter_spec",
    "blocks/signal/singlepolehighpassfilter_spec",
    "blocks/signal/fmpreemphasisfilter_spec",
    "blocks/signal/floattocomplex_spec",
    "blocks/signal/complextofloat_spec",
    "blocks/signal/hilberttransform_spec",
    "blocks/signal/upsampler_spec",
    "blocks/signal/complexbandpassfilter_spec",
    "blocks/signal/complexbandstopfilter_spec",
    "blocks/signal/rootraisedcosinefilter_spec",
    "composites/interpolator_spec",
    "composites/rationalresampler_spec",
    "blocks/sources/iqfile_spec",
    "blocks/sources/realfile_spec",
    "composites/decimator_spec",
    "composites/tuner_spec",
    "utilities/filter_utils_vectors",
    "utilities/window_utils_vectors",
    "utilities/spectrum_utils_vectors",
    "top_vectors",
]


class LuaLiteralParser:
    """Recursive-descent parser for the literal subset tests/generate.py emits."""

    TOKEN = re.compile(
        r"\s*(?:(?P<num>[-+]?(?:\d+\.\d*|\.\d+|\d+)(?:[eE][-+]?\d+)?)"
        r"|(?P<str>\"(?:[^\"\\]|\\.)*\")"
        r"|(?P<name>[A-Za-z_][A-Za-z_0-9.]*)"
        r"|(?P<punct>[{}(),=]))")

    def __init__(self, text, pos=0):
        self.text = text
        self.pos = pos

    def peek(self):
        m = self.TOKEN.match(self.text, self.pos)
        if not m:
            return None, None, self.pos
        return m.lastgroup, m.group(m.lastgroup), m.end()

    def take(self, expect=None):
        kind, val, end = self.peek()
        if kind is None:
            raise ValueError("unexpected end/char at %d: %r" % (self.pos, self.text[self.pos:self.pos + 40]))
        if expect is not None and val != expect:
            raise ValueError("expected %r got %r at %d" % (expect, val, self.pos))
        self.pos = end
        return kind, val

    def value(self):
        kind, val = self.take()
        if kind == "num":
            return float(val) if any(c in val for c in ".eE") else int(val)
        if kind == "str":
            return self._unescape(val[1:-1])
        if kind == "punct" and val == "{":
            return self.table()
        if kind == "name":
            if val == "true":
                return True
            if val == "false":
                return False
            if val == "nil":
                return None
            if val == "require":
                # require('tests.buffer').open("\x..") - an in-memory file (tests/buffer.lua); keep the bytes
                m2 = re.compile(r"\s*\(\s*'tests\.buffer'\s*\)\s*\.open\s*\(").match(self.text, self.pos)
                if not m2:
                    raise ValueError("unsupported require at %d" % self.pos)
                self.pos = m2.end()
                data = self.value()
                self.take(")")
                return data
            m = re.match(r"radio\.types\.(\w+)\.vector_from_array$", val)
            if m:
                self.take("(")
                data = self.value()
                self.take(")")
                return {"type": m.group(1), "data": data}
            m = re.match(r"radio\.types\.(\w+)$", val)
            if m:
                self.take("(")
                args = [self.value()]
                while self.peek()[1] == ",":
                    self.take(",")
                    args.append(self.value())
                self.take(")")
                return {"type": m.group(1), "scalar": args}
            raise ValueError("unknown name %r" % val)
        raise ValueError("unexpected token %r" % val)

    @staticmethod
    def _unescape(s):
        if "\\x" in s:
            raw = bytes(int(h, 16) for h in re.findall(r"\\x([0-9a-fA-F]{2})", s))
            return {"type": "bytes", "hex": raw.hex()}
        return s

    def table(self):
        """After '{'. Returns list (array part) or dict (if keys present)."""
        arr, rec = [], {}
        while True:
            kind, val, _ = self.peek()
            if val == "}":
                self.take("}")
                break
            if kind == "name" and not val.startswith("radio.") and val not in ("true", "false", "nil"):
                save = self.pos
                self.take()
                if self.peek()[1] == "=":
                    self.take("=")
                    rec[val] = self.value()
                else:
                    self.pos = save
                    arr.append(self.value())
            else:
                arr.append(self.value())
            if self.peek()[1] == ",":
                self.take(",")
        if rec and arr:
            rec["_array"] = arr
        return rec if rec else arr


def parse_block_spec(text):
    m = re.search(r"jigs\.TestBlock\(radio\.(\w+),\s*", text)
    name = m.group(1)
    p = LuaLiteralParser(text, m.end())
    p.take("{")
    vectors = p.table()
    eps = re.search(r"\{epsilon = (.*)\}\)\s*$", text[p.pos:], re.S).group(1).strip()
    return {"kind": "block", "block": name, "epsilon": eps, "vectors": vectors}


def parse_raw_spec(text):
    values = {}
    for m in re.finditer(r"^M\.(\w+) = ", text, re.M):
        p = LuaLiteralParser(text, m.end())
        values[m.group(1)] = p.value()
    return {"kind": "raw", "values": values}


def main():
    for spec in SPECS:
        src = os.path.join(REF, "tests", spec + ".gen.lua")
        with open(src) as f:
            text = f.read()
        doc = parse_block_spec(text) if "jigs.TestBlock" in text else parse_raw_spec(text)
        doc["source"] = "tests/" + spec + ".gen.lua"
        out = os.path.join(HERE, os.path.basename(spec) + ".json.gz")
        # mtime=0 keeps the gzip byte-stable across regenerations
        with gzip.GzipFile(out, "wb", mtime=0) as f:
            f.write(json.dumps(doc, separators=(",", ":")).encode())
        n = len(doc["vectors"]) if doc["kind"] == "block" else len(doc["values"])
        print("%-55s -> %s (%d entries)" % (doc["source"], os.path.basename(out), n))


if __name__ == "__main__":
    sys.exit(main())
