#!/usr/bin/env python3
"""Bind liblrhip.so into a LuaRadio checkout, mechanically.

    tools/apply_lua_binding.py <luaradio-checkout> [--out DIR] [--diff]

  * copies lua/radio/** (radio/core/lrhip.lua, the *_hip.lua device variants, radio/composites/devicechain.lua, devicefanout.lua) into the tree;
  * inserts ONE line directly above the final `return <Block>` of every block file that has a device variant:
        require('radio.core.lrhip').patch('<file>', <Block>)
    - after every top-level statement of the reference file, so that no later assignment can overwrite what the patch installs
    (radio/blocks/signal/firfilter.lua:400-402 / :488-490 assign process_fft_* after the dot-product ladder; block.factory(name, parent)
    copies the parent's functions when the DERIVED file loads, which is after the parent file returned - radio/core/class.lua:18-40);
  * inserts the DeviceChainBlock.collapse() hook into CompositeBlock:_prepare_to_run (radio/core/composite.lua:426-470).
Without --out the checkout is edited in place; with --out DIR only the touched files are written below DIR (same relative paths).
--diff prints a unified diff instead of writing anything.  tests/test_lua_glue.py runs `patched_sources()` against /root/reference
(when it is present) and models the load order of every patched file.
"""
import argparse
import difflib
import glob
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# reference block files (radio/blocks/signal/<name>.lua) that get a device variant; the patch name is the file's base name
BLOCK_FILES = [
    "firfilter", "frequencytranslator", "downsampler", "frequencydiscriminator", "iirfilter",
    "complexmagnitude", "complexphase", "complextoreal", "complextoimag", "complexconjugate", "realtocomplex", "absolutevalue",
    "addconstant", "multiplyconstant", "delay", "hilberttransform", "upsampler", "frequencymodulator", "agc", "powersquelch",
    "multiply", "multiplyconjugate", "add", "subtract", "floattocomplex",
]

COLLAPSE_HOOK = """
    -- Collapse every maximal linear run of device blocks into one DeviceChainBlock (liblrhip.so)
    local device_chains = {}
    if require('radio.core.lrhip').available then
        all_connections, device_chains = require('radio.composites.devicechain').collapse(all_connections)
        -- ... and give every output port that fans out into device chains ONE upload and GPU-to-GPU copies (one branch per GPU)
        all_connections, device_chains = require('radio.composites.devicefanout').collapse(all_connections, device_chains)
    end
"""
INIT_HOOK = "    for _, chain in ipairs(device_chains) do chain:initialize() end\n"


def patch_line(name, block_var):
    return "require('radio.core.lrhip').patch('%s', %s)\n" % (name, block_var)


def patch_block_source(name, text):
    """insert the patch line above the final `return <Var>` of a block file; returns (new_text, block_var)"""
    lines = text.splitlines(keepends=True)
    last = None
    for i, ln in enumerate(lines):
        m = re.match(r"^return\s+([A-Za-z_]\w*)\s*$", ln)
        if m:
            last = (i, m.group(1))
    if last is None:
        raise ValueError("%s: no top-level `return <Block>`" % name)
    i, var = last
    if any("radio.core.lrhip" in ln for ln in lines):
        raise ValueError("%s: already patched" % name)
    lines.insert(i, patch_line(name, var) + "\n")
    return "".join(lines), var


def patch_composite_source(text):
    """the two insertions into CompositeBlock:_prepare_to_run"""
    a = "    local all_connections = self:_crawl_connections()\n"
    b = "    self:_initialize()\n"
    if text.count(a) != 1:
        raise ValueError("composite.lua: `_crawl_connections()` call not found exactly once")
    head, tail = text.split(a)
    if tail.count(b) < 1:
        raise ValueError("composite.lua: `self:_initialize()` not found after the crawl")
    t0, t1 = tail.split(b, 1)
    return head + a + COLLAPSE_HOOK + t0 + b + INIT_HOOK + t1


def patched_sources(checkout):
    """{relative path: (old text or None, new text)} for everything the binding touches"""
    out = {}
    for name in BLOCK_FILES:
        rel = os.path.join("radio", "blocks", "signal", name + ".lua")
        old = open(os.path.join(checkout, rel)).read()
        out[rel] = (old, patch_block_source(name, old)[0])
    rel = os.path.join("radio", "core", "composite.lua")
    old = open(os.path.join(checkout, rel)).read()
    out[rel] = (old, patch_composite_source(old))
    for path in sorted(glob.glob(os.path.join(ROOT, "lua", "radio", "**", "*.lua"), recursive=True)):
        rel = os.path.relpath(path, os.path.join(ROOT, "lua"))
        out[rel] = (None, open(path).read())
    return out


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("checkout")
    ap.add_argument("--out")
    ap.add_argument("--diff", action="store_true")
    args = ap.parse_args()
    files = patched_sources(args.checkout)
    if args.diff:
        for rel, (old, new) in sorted(files.items()):
            sys.stdout.writelines(difflib.unified_diff((old or "").splitlines(keepends=True), new.splitlines(keepends=True),
                                                       "a/" + rel if old is not None else "/dev/null", "b/" + rel, n=2))
        return 0
    base = args.out or args.checkout
    for rel, (_, new) in files.items():
        dst = os.path.join(base, rel)
        os.makedirs(os.path.dirname(dst), exist_ok=True)
        with open(dst, "w") as f:
            f.write(new)
    print("%d files written below %s" % (len(files), base))
    return 0


if __name__ == "__main__":
    sys.exit(main())
