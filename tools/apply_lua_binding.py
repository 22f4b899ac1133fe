#!/usr/bin/env python3
"""Bind liblrhip.so into a LuaRadio checkout, mechanically.

    tools/apply_lua_binding.py <luaradio-checkout> [--out DIR] [--diff]

  * copies lua/radio/** (radio/core/lrhip.lua, the *_hip.lua device variants, radio/composites/devicechain.lua, devicefanout.lua) into the tree;
  * inserts ONE line directly above the final `return <Block>` of every block file that has a device variant:
        require('radio.core.lrhip').patch('<file>', <Block>)
    - after every top-level statement of the reference file, so that no later assignment can overwrite what the patch installs
    (radio/blocks/signal/firfilter.lua:400-402 / :488-490 assign process_fft_* after the dot-product ladder; block.factory(name, parent)
    copies the parent's functions when the DERIVED file loads, which is after the parent file returned - radio/core/class.lua:18-40);
  * does the same for the file sources / sinks and the spectrum sink (radio/blocks/sources/{iqfile,realfile}.lua, radio/blocks/sinks/{iqfile,realfile,
    gnuplotspectrum}.lua) and puts `require('radio.core.lrhip').patch_spectrum(DFT, IDFT, PSD)` above the final return of radio/utilities/spectrum_utils.lua;
  * inserts the collapse() hooks (devicegraph, devicechain, devicefanout) into CompositeBlock:_prepare_to_run (radio/core/composite.lua:426-470) and the
    parent-side close of the fan-out sockets after the fork loop of CompositeBlock:start (:638-642).
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
# ... and the file sources / sinks and the spectrum sink: (path below radio/blocks, patch name) - two directories hold an iqfile.lua
IO_FILES = [
    ("sources/iqfile", "iqfilesource"), ("sources/realfile", "realfilesource"),
    ("sinks/iqfile", "iqfilesink"), ("sinks/realfile", "realfilesink"), ("sinks/gnuplotspectrum", "gnuplotspectrum"),
]
# radio/utilities/spectrum_utils.lua returns a table of classes: its line goes above that `return {DFT = DFT, ...}`
SPECTRUM_RETURN = "return {DFT = DFT, IDFT = IDFT, PSD = PSD, fftshift = fftshift}\n"
SPECTRUM_LINE = "require('radio.core.lrhip').patch_spectrum(DFT, IDFT, PSD)\n"

COLLAPSE_HOOK = """
    -- Device blocks (liblrhip.so): a connected subgraph with a join becomes one DeviceGraphBlock, every maximal linear run of device blocks one
    -- DeviceChainBlock (a file source in front and a file sink behind included) ...
    local device_chains = {}
    if require('radio.core.lrhip').available then
        local graphs, chains
        all_connections, graphs = require('radio.composites.devicegraph').collapse(all_connections)
        all_connections, chains = require('radio.composites.devicechain').collapse(all_connections)
        -- ... and every output port that fans out into device chains gets ONE upload and GPU-to-GPU copies (one branch per GPU)
        all_connections, device_chains = require('radio.composites.devicefanout').collapse(all_connections, chains)
        for _, g in ipairs(graphs) do device_chains[#device_chains + 1] = g end
    end
"""
INIT_HOOK = "    for _, chain in ipairs(device_chains) do chain:initialize() end\n"
# a chain that absorbed BOTH its file source and its file sink has no port left, so it is in no connection and build_dependency_graph() cannot see it
ORDER_HOOK = """    for _, chain in ipairs(device_chains) do
        if #chain.inputs == 0 and #chain.outputs == 0 then evaluation_order[#evaluation_order + 1] = chain end
    end
"""
# CompositeBlock:start, after the fork loop, next to "Close all pipe inputs and outputs in the top-level process" (composite.lua:638-642): the parent's
# copies of the fan-out socket pairs go too, so that a dead branch / head is seen as EOF by its peer (devicefanout.lua close_parent_fds)
CLOSE_HOOK = """        for _, b in ipairs(evaluation_order) do
            if b.close_parent_fds then b:close_parent_fds() end
        end
"""


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


def patch_spectrum_source(text):
    if text.count(SPECTRUM_RETURN) != 1:
        raise ValueError("spectrum_utils.lua: final `return {DFT = ...}` not found exactly once")
    if "radio.core.lrhip" in text:
        raise ValueError("spectrum_utils.lua: already patched")
    return text.replace(SPECTRUM_RETURN, SPECTRUM_LINE + "\n" + SPECTRUM_RETURN)


def patch_composite_source(text):
    """the four insertions into radio/core/composite.lua: collapse + initialize + evaluation order in CompositeBlock:_prepare_to_run, and the parent's
    close of the fan-out sockets after the fork loop of CompositeBlock:start"""
    a = "    local all_connections = self:_crawl_connections()\n"
    b = "    self:_initialize()\n"
    c = "    local evaluation_order = build_evaluation_order(build_dependency_graph(all_connections))\n"
    d = "        -- Close all pipe inputs and outputs in the top-level process\n"
    if text.count(a) != 1:
        raise ValueError("composite.lua: `_crawl_connections()` call not found exactly once")
    head, tail = text.split(a)
    if tail.count(b) < 1:
        raise ValueError("composite.lua: `self:_initialize()` not found after the crawl")
    t0, t1 = tail.split(b, 1)
    if t1.count(c) != 1:
        raise ValueError("composite.lua: the global evaluation order is not built where expected")
    t1a, t1b = t1.split(c)
    if t1b.count(d) != 1:
        raise ValueError("composite.lua: the parent's pipe close after the fork loop not found")
    t1b0, t1b1 = t1b.split(d)
    return head + a + COLLAPSE_HOOK + t0 + b + INIT_HOOK + t1a + c + ORDER_HOOK + t1b0 + CLOSE_HOOK + d + t1b1


def patched_sources(checkout):
    """{relative path: (old text or None, new text)} for everything the binding touches"""
    out = {}
    for name in BLOCK_FILES:
        rel = os.path.join("radio", "blocks", "signal", name + ".lua")
        old = open(os.path.join(checkout, rel)).read()
        out[rel] = (old, patch_block_source(name, old)[0])
    for path, name in IO_FILES:
        rel = os.path.join("radio", "blocks", path + ".lua")
        old = open(os.path.join(checkout, rel)).read()
        out[rel] = (old, patch_block_source(name, old)[0])
    rel = os.path.join("radio", "utilities", "spectrum_utils.lua")
    old = open(os.path.join(checkout, rel)).read()
    out[rel] = (old, patch_spectrum_source(old))
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
