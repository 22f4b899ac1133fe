"""Static checks of the LuaJIT glue under lua/radio/** (LuaJIT is absent from the build image, so the files cannot be executed).

A small Lua tokenizer (comments and strings stripped) holds the glue to
  * the C ABI: the ffi.cdef in lua/radio/core/lrhip.lua declares every function of include/lrhip.h with the SAME prototype
    (return type and parameter types), and every `lib.lrhip_*(...)` call in any Lua file names a declared function and passes
    the declared number of arguments;
  * itself: every `lrhip.<helper>(...)` used is defined in lrhip.lua, every method called on a block (`x:method(...)`) is either
    defined by the glue (function X:method / X.method = / lrhip.device_block -> create_stage) or part of the reference's
    Block / Vector / Pipe / port API that the glue relies on, and every plain function called is a local, a parameter,
    a Lua builtin or a module table.
VERDICT r01 found `b:create_stage()` called but defined nowhere; this test fails on that class of defect.
"""
import glob
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LUA_FILES = sorted(glob.glob(os.path.join(ROOT, "lua", "radio", "**", "*.lua"), recursive=True))

# methods of the reference's objects that the glue calls (radio/core/block.lua, vector.lua, pipe.lua, class.lua, types)
REFERENCE_METHODS = {
    "add_type_signature", "differentiate", "get_input_type", "get_output_type", "get_rate", "initialize",   # Block
    "resize",                                                                                                # Vector
    "write",                                                                                                 # Pipe
    "vector",                                                                                                # data type .vector() is called with '.', listed for safety
}
LUA_BUILTINS = {"assert", "error", "ipairs", "pairs", "require", "tonumber", "tostring", "type", "setmetatable", "unpack", "pcall", "select", "print"}
LUA_KEYWORDS = {"and", "break", "do", "else", "elseif", "end", "false", "for", "function", "if", "in", "local", "nil", "not", "or", "repeat", "return",
                "then", "true", "until", "while"}


def strip_lua(text):
    """remove comments and string literals (replaced by "" / blanks), keep line structure; returns (code, cdef_bodies)"""
    out, cdefs, i, n = [], [], 0, len(text)
    while i < n:
        if text.startswith("--[[", i):
            j = text.find("]]", i)
            out.append("\n" * text.count("\n", i, j + 2))
            i = j + 2
        elif text.startswith("--", i):
            j = text.find("\n", i)
            j = n if j < 0 else j
            i = j
        elif text.startswith("[[", i):
            j = text.find("]]", i)
            body = text[i + 2:j]
            if out and "".join(out).rstrip().endswith("ffi.cdef"):
                cdefs.append(body)
            out.append('""' + "\n" * body.count("\n"))
            i = j + 2
        elif text[i] in "\"'":
            q, j = text[i], i + 1
            while text[j] != q:
                j += 2 if text[j] == "\\" else 1
            out.append('""')
            i = j + 1
        else:
            out.append(text[i])
            i += 1
    return "".join(out), cdefs


def c_prototypes(text):
    """{name: (normalised return type, [normalised parameter types])} of the function prototypes in a C fragment"""
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"//[^\n]*", " ", text)
    protos = {}
    for m in re.finditer(r"([A-Za-z_][\w\s\*]*?)\b(lrhip_\w+)\s*\(([^;{}]*?)\)\s*;", text, flags=re.S):
        ret, name, params = m.group(1), m.group(2), m.group(3)

        def norm(t):
            t = re.sub(r"\s+", " ", t.replace("*", " * ")).strip()
            return t

        plist = []
        params = params.strip()
        if params and params != "void":
            for prm in params.split(","):
                prm = norm(prm)
                # drop the parameter name (last identifier) when a type precedes it
                toks = prm.split(" ")
                if len(toks) > 1 and re.match(r"^[A-Za-z_]\w*$", toks[-1]) and toks[-1] not in ("int", "long", "unsigned", "float", "double", "char", "void"):
                    toks = toks[:-1]
                plist.append(" ".join(toks))
        protos[name] = (norm(ret.replace("extern", "")), plist)
    return protos


def call_args(code, open_paren):
    """number of top-level arguments of the call whose '(' is at index open_paren"""
    depth, i, args, seen = 0, open_paren, 0, False
    while True:
        ch = code[i]
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
            if depth == 0:
                return args + (1 if seen else 0)
        elif ch == "," and depth == 1:
            args += 1
        elif depth >= 1 and not ch.isspace():
            seen = True
        i += 1


def load():
    files = {}
    for path in LUA_FILES:
        code, cdefs = strip_lua(open(path).read())
        files[os.path.relpath(path, ROOT)] = (code, cdefs)
    return files


def test_lua_files_exist_and_balance():
    assert len(LUA_FILES) >= 4
    for path, (code, _) in load().items():
        for a, b in ("()", "[]", "{}"):
            assert code.count(a) == code.count(b), (path, a)
        toks = re.findall(r"\b\w+\b", code)
        opens = sum(toks.count(k) for k in ("function", "do", "then", "repeat")) - toks.count("elseif")     # elseif ... then shares its if's end
        closes = toks.count("end") + toks.count("until")
        assert opens == closes, (path, opens, closes)


def test_cdef_matches_the_header_prototype_for_prototype():
    header = c_prototypes(open(os.path.join(ROOT, "include", "lrhip.h")).read())
    _, cdefs = load()[os.path.join("lua", "radio", "core", "lrhip.lua")]
    assert len(cdefs) == 1
    cdef = c_prototypes(cdefs[0])
    assert len(header) > 60
    for name, proto in header.items():
        assert name in cdef, "cdef lacks %s" % name
        assert cdef[name] == proto, (name, cdef[name], proto)
    for name in cdef:
        assert name in header, "cdef declares %s, which include/lrhip.h does not" % name


def test_every_library_call_names_a_declared_function_with_its_argument_count():
    files = load()
    cdef = c_prototypes(files[os.path.join("lua", "radio", "core", "lrhip.lua")][1][0])
    used = set()
    for path, (code, _) in files.items():
        for m in re.finditer(r"\blib\.(lrhip_\w+)\s*(\()?", code):
            name = m.group(1)
            assert name in cdef, (path, name)
            used.add(name)
            if m.group(2):
                assert call_args(code, m.end() - 1) == len(cdef[name][1]), (path, name, call_args(code, m.end() - 1), cdef[name][1])
    # the boundary the path needs is actually bound: stage constructors, execute, chain with coalescing
    for name in ("lrhip_init", "lrhip_fir_create", "lrhip_rotator_create", "lrhip_downsampler_create", "lrhip_fmdiscrim_create", "lrhip_iir_create",
                 "lrhip_stage_execute", "lrhip_stage_max_output", "lrhip_stage_destroy", "lrhip_chain_create", "lrhip_chain_set_ring",
                 "lrhip_chain_push", "lrhip_chain_flush", "lrhip_chain_push_bound", "lrhip_chain_destroy", "lrhip_strerror"):
        assert name in used, name


def _definitions(files):
    methods, helpers = set(), set()
    for path, (code, _) in files.items():
        methods |= set(re.findall(r"\bfunction\s+\w+[:.](\w+)\s*\(", code))
        methods |= set(re.findall(r"\b[A-Z]\w*\.(\w+)\s*=", code))
        if path.endswith(os.path.join("core", "lrhip.lua")):
            helpers |= set(re.findall(r"\bfunction\s+M\.(\w+)\s*\(", code))
            helpers |= {"lib", "available"}
    return methods, helpers


def test_every_method_and_helper_the_glue_calls_is_defined():
    files = load()
    methods, helpers = _definitions(files)
    assert "create_stage" in methods, "lrhip.device_block must define Block:create_stage()"
    for path, (code, _) in files.items():
        for m in re.finditer(r"[\w\]\)]\s*:\s*(\w+)\s*\(", code):
            name = m.group(1)
            assert name in methods or name in REFERENCE_METHODS, "%s calls :%s(), defined nowhere" % (path, name)
        for m in re.finditer(r"\blrhip\.(\w+)", code):
            assert m.group(1) in helpers, "%s uses lrhip.%s, not defined in radio/core/lrhip.lua" % (path, m.group(1))


def test_every_plain_function_call_resolves():
    for path, (code, _) in load().items():
        local_names = set(re.findall(r"\blocal\s+function\s+(\w+)", code))
        for decl in re.findall(r"\blocal\s+([\w\s,]+?)\s*(?:=|\n|$)", code):
            local_names |= {t.strip() for t in decl.split(",")}
        for params in re.findall(r"\bfunction\b[^(\n]*\(([^)]*)\)", code):
            local_names |= {t.strip() for t in params.split(",") if t.strip()}
        for names in re.findall(r"\bfor\s+([\w\s,]+?)\s+in\b", code):
            local_names |= {t.strip() for t in names.split(",")}
        local_names |= set(re.findall(r"\bfor\s+(\w+)\s*=", code))
        for m in re.finditer(r"(?<![\w.:])([A-Za-z_]\w*)\s*\(", code):
            name = m.group(1)
            if name in LUA_KEYWORDS or name in LUA_BUILTINS:
                continue
            assert name in local_names, "%s calls %s(), which is neither local, a parameter nor a builtin" % (path, name)


def test_fir_mode_mapping_and_collapse_hook_are_documented():
    """the use_fft pass-through (0/1/2/3, automatic by default) and the _prepare_to_run hook are what INTEGRATION.md shows"""
    files = load()
    core = files[os.path.join("lua", "radio", "core", "lrhip.lua")][0]
    assert re.search(r"function M\.fir_mode\(use_fft\)", core)
    fir = files[os.path.join("lua", "radio", "blocks", "signal", "firfilter_hip.lua")][0]
    assert "lrhip.fir_mode(self.use_fft_argument)" in fir and "self.use_fft and 1 or 0" not in fir
    chain = files[os.path.join("lua", "radio", "composites", "devicechain.lua")][0]
    assert "function DeviceChainBlock.collapse(connections)" in chain and "return result, chains" in chain
    integ = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    assert "collapse(all_connections)" in integ and "chain:initialize()" in integ
