"""A small Lua 5.1 interpreter - TEST INFRASTRUCTURE, not product.

LuaJIT is not installed in the build image, so lua/radio/** could only be checked statically (tests/test_lua_glue.py: a tokenizer model).  This
interpreter runs those files: lexer, recursive-descent parser and a tree-walking evaluator for the subset the glue uses - locals, closures,
multiple assignment / returns, varargs, tables with metatables (__index, __newindex, __call, __tostring, __len, __eq, __concat, __mode ignored),
method calls, numeric and generic `for`, `while`, `repeat`, `if`, `break`, `return`, long strings / comments, the arithmetic, comparison, logical,
concatenation and length operators - plus the slice of the standard library they touch (string, table, math, os.getenv, io.stderr, pcall, error,
select, unpack, rawget / rawset, next, pairs / ipairs, tostring / tonumber, setmetatable / getmetatable, require with package.loaded / preload).

Not implemented (unused by the glue): goto, coroutines, string patterns beyond plain find / gsub with literal patterns, integer division, bit ops.
`ffi` and the reference's core modules are provided by tests/helpers/lua_mocks.py.
"""
import math
import os
import re

# ----------------------------------------------------------------------------------------------------------------------------------- lexer
KEYWORDS = {"and", "break", "do", "else", "elseif", "end", "false", "for", "function", "if", "in", "local", "nil", "not", "or", "repeat", "return",
            "then", "true", "until", "while"}
TOKEN_RE = re.compile(r"""
    (?P<ws>\s+)
  | (?P<lcomment>--\[(?P<lc_eq>=*)\[)
  | (?P<comment>--[^\n]*)
  | (?P<lstring>\[(?P<ls_eq>=*)\[)
  | (?P<number>0[xX][0-9a-fA-F]+|\d+\.?\d*(?:[eE][+-]?\d+)?|\.\d+(?:[eE][+-]?\d+)?)
  | (?P<name>[A-Za-z_][A-Za-z0-9_]*)
  | (?P<string>"(?:\\.|[^"\\\n])*"|'(?:\\.|[^'\\\n])*')
  | (?P<op>\.\.\.|\.\.|==|~=|<=|>=|[-+*/%^\#<>=(){}\[\];:,.])
""", re.X)
ESCAPES = {"n": "\n", "t": "\t", "r": "\r", "\\": "\\", '"': '"', "'": "'", "a": "\a", "b": "\b", "f": "\f", "v": "\v", "0": "\0", "\n": "\n"}


class LuaError(Exception):
    def __init__(self, value, traceback=None):
        Exception.__init__(self, tostr(value) if not isinstance(value, str) else value)
        self.value = value


def lex(src, chunk="?"):
    toks, i, line = [], 0, 1
    while i < len(src):
        m = TOKEN_RE.match(src, i)
        if not m:
            raise LuaError("%s:%d: unexpected character %r" % (chunk, line, src[i]))
        kind = m.lastgroup
        text = m.group(0)
        if kind in ("lcomment", "lstring"):
            eq = m.group("lc_eq") if kind == "lcomment" else m.group("ls_eq")
            close = "]" + eq + "]"
            j = src.find(close, m.end())
            if j < 0:
                raise LuaError("%s:%d: unfinished long bracket" % (chunk, line))
            body = src[m.end():j]
            if kind == "lstring":
                if body.startswith("\n"):
                    body = body[1:]
                toks.append(("string", body, line))
            line += src.count("\n", i, j + len(close))
            i = j + len(close)
            continue
        i = m.end()
        if kind in ("ws", "comment"):
            line += text.count("\n")
            continue
        if kind == "lc_eq" or kind == "ls_eq":
            continue
        if kind == "number":
            toks.append(("number", float(int(text, 16)) if text[:2].lower() == "0x" else float(text), line))
        elif kind == "name":
            toks.append(("kw" if text in KEYWORDS else "name", text, line))
        elif kind == "string":
            body, out, k = text[1:-1], [], 0
            while k < len(body):
                c = body[k]
                if c == "\\":
                    k += 1
                    e = body[k]
                    if e.isdigit():
                        d = re.match(r"\d{1,3}", body[k:]).group(0)
                        out.append(chr(int(d)))
                        k += len(d)
                        continue
                    out.append(ESCAPES.get(e, e))
                else:
                    out.append(c)
                k += 1
            toks.append(("string", "".join(out), line))
        else:
            toks.append(("op", text, line))
    toks.append(("eof", None, line))
    return toks


# ---------------------------------------------------------------------------------------------------------------------------------- parser
BINPRI = {"or": (1, 1), "and": (2, 2), "<": (3, 3), ">": (3, 3), "<=": (3, 3), ">=": (3, 3), "~=": (3, 3), "==": (3, 3), "..": (5, 4),
          "+": (6, 6), "-": (6, 6), "*": (7, 7), "/": (7, 7), "%": (7, 7), "^": (10, 9)}
UNARY_PRI = 8


class Parser:
    def __init__(self, src, chunk):
        self.toks, self.p, self.chunk = lex(src, chunk), 0, chunk

    def peek(self):
        return self.toks[self.p]

    def next(self):
        t = self.toks[self.p]
        self.p += 1
        return t

    def check(self, kind, val=None):
        t = self.toks[self.p]
        return t[0] == kind and (val is None or t[1] == val)

    def accept(self, kind, val=None):
        if self.check(kind, val):
            return self.next()
        return None

    def expect(self, kind, val=None):
        t = self.next()
        if t[0] != kind or (val is not None and t[1] != val):
            raise LuaError("%s:%d: expected %s, got %r" % (self.chunk, t[2], val or kind, t[1]))
        return t

    def block(self):
        stmts = []
        while True:
            t = self.peek()
            if t[0] == "eof" or (t[0] == "kw" and t[1] in ("end", "else", "elseif", "until")):
                break
            if t[0] == "kw" and t[1] == "return":
                self.next()
                exprs = []
                if not (self.check("eof") or (self.peek()[0] == "kw" and self.peek()[1] in ("end", "else", "elseif", "until")) or self.check("op", ";")):
                    exprs = self.exprlist()
                self.accept("op", ";")
                stmts.append(("return", exprs, t[2]))
                break
            s = self.statement()
            if s is not None:
                stmts.append(s)
        return stmts

    def statement(self):
        t = self.peek()
        line = t[2]
        if t[0] == "op" and t[1] == ";":
            self.next()
            return None
        if t[0] == "kw":
            k = t[1]
            if k == "if":
                self.next()
                clauses = []
                cond = self.expr()
                self.expect("kw", "then")
                clauses.append((cond, self.block()))
                orelse = None
                while True:
                    if self.accept("kw", "elseif"):
                        c = self.expr()
                        self.expect("kw", "then")
                        clauses.append((c, self.block()))
                    elif self.accept("kw", "else"):
                        orelse = self.block()
                        self.expect("kw", "end")
                        break
                    else:
                        self.expect("kw", "end")
                        break
                return ("if", clauses, orelse, line)
            if k == "while":
                self.next()
                c = self.expr()
                self.expect("kw", "do")
                b = self.block()
                self.expect("kw", "end")
                return ("while", c, b, line)
            if k == "do":
                self.next()
                b = self.block()
                self.expect("kw", "end")
                return ("do", b, line)
            if k == "for":
                self.next()
                n1 = self.expect("name")[1]
                if self.accept("op", "="):
                    a = self.expr()
                    self.expect("op", ",")
                    b = self.expr()
                    c = self.expr() if self.accept("op", ",") else None
                    self.expect("kw", "do")
                    body = self.block()
                    self.expect("kw", "end")
                    return ("fornum", n1, a, b, c, body, line)
                names = [n1]
                while self.accept("op", ","):
                    names.append(self.expect("name")[1])
                self.expect("kw", "in")
                exprs = self.exprlist()
                self.expect("kw", "do")
                body = self.block()
                self.expect("kw", "end")
                return ("forin", names, exprs, body, line)
            if k == "repeat":
                self.next()
                b = self.block()
                self.expect("kw", "until")
                return ("repeat", b, self.expr(), line)
            if k == "function":
                self.next()
                target = ("name", self.expect("name")[1])
                is_method = False
                while True:
                    if self.accept("op", "."):
                        target = ("index", target, ("const", self.expect("name")[1]))
                    elif self.accept("op", ":"):
                        target = ("index", target, ("const", self.expect("name")[1]))
                        is_method = True
                        break
                    else:
                        break
                return ("assign", [target], [self.funcbody(is_method, line)], line)
            if k == "local":
                self.next()
                if self.accept("kw", "function"):
                    name = self.expect("name")[1]
                    return ("localfunc", name, self.funcbody(False, line), line)
                names = [self.expect("name")[1]]
                while self.accept("op", ","):
                    names.append(self.expect("name")[1])
                exprs = self.exprlist() if self.accept("op", "=") else []
                return ("local", names, exprs, line)
            if k == "break":
                self.next()
                return ("break", line)
        # expression statement: call or assignment
        e = self.suffixedexp()
        if self.check("op", "=") or self.check("op", ","):
            targets = [e]
            while self.accept("op", ","):
                targets.append(self.suffixedexp())
            self.expect("op", "=")
            return ("assign", targets, self.exprlist(), line)
        if e[0] not in ("call", "method"):
            raise LuaError("%s:%d: syntax error near %r" % (self.chunk, line, self.peek()[1]))
        return ("exprstat", e, line)

    def funcbody(self, is_method, line):
        self.expect("op", "(")
        params, vararg = (["self"] if is_method else []), False
        if not self.check("op", ")"):
            while True:
                if self.accept("op", "..."):
                    vararg = True
                    break
                params.append(self.expect("name")[1])
                if not self.accept("op", ","):
                    break
        self.expect("op", ")")
        body = self.block()
        self.expect("kw", "end")
        return ("function", params, vararg, body, line)

    def exprlist(self):
        out = [self.expr()]
        while self.accept("op", ","):
            out.append(self.expr())
        return out

    def primaryexp(self):
        t = self.next()
        if t[0] == "name":
            return ("name", t[1])
        if t[0] == "op" and t[1] == "(":
            e = self.expr()
            self.expect("op", ")")
            return ("paren", e)
        raise LuaError("%s:%d: unexpected symbol %r" % (self.chunk, t[2], t[1]))

    def suffixedexp(self):
        e = self.primaryexp()
        while True:
            t = self.peek()
            if t[0] == "op" and t[1] == ".":
                self.next()
                e = ("index", e, ("const", self.expect("name")[1]))
            elif t[0] == "op" and t[1] == "[":
                self.next()
                k = self.expr()
                self.expect("op", "]")
                e = ("index", e, k)
            elif t[0] == "op" and t[1] == ":":
                self.next()
                name = self.expect("name")[1]
                e = ("method", e, name, self.callargs(), t[2])
            elif (t[0] == "op" and t[1] in ("(", "{")) or t[0] == "string":
                e = ("call", e, self.callargs(), t[2])
            else:
                return e

    def callargs(self):
        t = self.peek()
        if t[0] == "string":
            self.next()
            return [("const", t[1])]
        if t[0] == "op" and t[1] == "{":
            return [self.table()]
        self.expect("op", "(")
        args = [] if self.check("op", ")") else self.exprlist()
        self.expect("op", ")")
        return args

    def table(self):
        self.expect("op", "{")
        items = []
        while not self.check("op", "}"):
            if self.check("op", "["):
                self.next()
                k = self.expr()
                self.expect("op", "]")
                self.expect("op", "=")
                items.append(("kv", k, self.expr()))
            elif self.check("name") and self.toks[self.p + 1][0] == "op" and self.toks[self.p + 1][1] == "=":
                k = self.next()[1]
                self.next()
                items.append(("kv", ("const", k), self.expr()))
            else:
                items.append(("pos", self.expr()))
            if not (self.accept("op", ",") or self.accept("op", ";")):
                break
        self.expect("op", "}")
        return ("table", items)

    def simpleexp(self):
        t = self.peek()
        if t[0] == "number" or t[0] == "string":
            self.next()
            return ("const", t[1])
        if t[0] == "kw":
            if t[1] == "nil":
                self.next()
                return ("const", None)
            if t[1] == "true":
                self.next()
                return ("const", True)
            if t[1] == "false":
                self.next()
                return ("const", False)
            if t[1] == "function":
                self.next()
                return self.funcbody(False, t[2])
        if t[0] == "op" and t[1] == "...":
            self.next()
            return ("vararg",)
        if t[0] == "op" and t[1] == "{":
            return self.table()
        return self.suffixedexp()

    def expr(self, limit=0):
        t = self.peek()
        if (t[0] == "kw" and t[1] == "not") or (t[0] == "op" and t[1] in ("-", "#")):
            self.next()
            left = ("unop", t[1], self.expr(UNARY_PRI))
        else:
            left = self.simpleexp()
        while True:
            t = self.peek()
            op = t[1] if (t[0] == "op" or t[0] == "kw") and t[1] in BINPRI else None
            if op is None or BINPRI[op][0] <= limit:
                return left
            self.next()
            right = self.expr(BINPRI[op][1])
            left = ("binop", op, left, right, t[2])


def parse(src, chunk="?"):
    p = Parser(src, chunk)
    body = p.block()
    p.expect("eof")
    return body


# -------------------------------------------------------------------------------------------------------------------------------- runtime
class LuaTable:
    __slots__ = ("hash", "meta")

    def __init__(self):
        self.hash, self.meta = {}, None

    def get(self, k):
        if isinstance(k, float) and k == int(k):
            k = int(k)
        return self.hash.get(k)

    def set(self, k, v):
        if isinstance(k, float) and k == int(k):
            k = int(k)
        if k is None:
            raise LuaError("table index is nil")
        if v is None:
            self.hash.pop(k, None)
        else:
            self.hash[k] = v

    def length(self):
        n = 0
        while (n + 1) in self.hash:
            n += 1
        return n

    def __repr__(self):
        return "table: 0x%08x" % (id(self) & 0xffffffff)


class LuaFunction:
    __slots__ = ("params", "vararg", "body", "env", "interp", "name")

    def __init__(self, params, vararg, body, env, interp, name="?"):
        self.params, self.vararg, self.body, self.env, self.interp, self.name = params, vararg, body, env, interp, name

    def __call__(self, *args):
        return self.interp.call_function(self, list(args))

    def __repr__(self):
        return "function: 0x%08x" % (id(self) & 0xffffffff)


class Env:
    __slots__ = ("vars", "parent")

    def __init__(self, parent=None):
        self.vars, self.parent = {}, parent

    def lookup(self, name):
        e = self
        while e is not None:
            if name in e.vars:
                return e
            e = e.parent
        return None


class BreakLoop(Exception):
    pass


class ReturnValues(Exception):
    def __init__(self, values):
        self.values = values


def truthy(v):
    return v is not None and v is not False


def tostr(v):
    if v is None:
        return "nil"
    if v is True:
        return "true"
    if v is False:
        return "false"
    if isinstance(v, float):
        if v == int(v) and abs(v) < 1e15:
            return str(int(v))
        return repr(v)
    if isinstance(v, int):
        return str(v)
    if isinstance(v, str):
        return v
    if isinstance(v, LuaTable):
        mt = v.meta
        if mt is not None and mt.get("__tostring") is not None:
            r = call(mt.get("__tostring"), [v])
            return r[0] if r else "nil"
        return repr(v)
    if hasattr(v, "lua_tostring"):
        return v.lua_tostring()
    return repr(v)


def tonum(v, base=None):
    if isinstance(v, bool):
        return None
    if isinstance(v, (int, float)):
        return float(v)
    if isinstance(v, str):
        s = v.strip()
        try:
            if base is not None:
                return float(int(s, int(base)))
            return float(int(s, 16)) if s[:2].lower() == "0x" else float(s)
        except ValueError:
            return None
    if hasattr(v, "lua_tonumber"):
        return v.lua_tonumber()
    return None


def lua_type(v):
    if v is None:
        return "nil"
    if isinstance(v, bool):
        return "boolean"
    if isinstance(v, (int, float)):
        return "number"
    if isinstance(v, str):
        return "string"
    if isinstance(v, LuaTable):
        return "table"
    if isinstance(v, LuaFunction) or callable(v) and not hasattr(v, "lua_type"):
        return "function"
    return getattr(v, "lua_type", "userdata")


def call(f, args):
    """call a Lua value; always returns a list of results"""
    if isinstance(f, LuaFunction):
        return f.interp.call_function(f, args)
    if isinstance(f, LuaTable):
        mt = f.meta
        h = mt.get("__call") if mt is not None else None
        if h is None:
            raise LuaError("attempt to call a table value")
        return call(h, [f] + list(args))
    if hasattr(f, "lua_call"):
        return as_list(f.lua_call(*args))
    if callable(f):
        return as_list(f(*args))
    raise LuaError("attempt to call a %s value" % lua_type(f))


def as_list(r):
    if r is None:
        return []
    if isinstance(r, tuple):
        return list(r)
    if isinstance(r, list):
        return r
    return [r]


def index(obj, key):
    if isinstance(obj, LuaTable):
        v = obj.get(key)
        if v is not None:
            return v
        mt = obj.meta
        if mt is None:
            return None
        h = mt.get("__index")
        if h is None:
            return None
        if isinstance(h, LuaTable):
            return index(h, key)
        r = call(h, [obj, key])
        return r[0] if r else None
    if isinstance(obj, str):
        return STRING_LIB.get(key)
    if hasattr(obj, "lua_index"):
        return obj.lua_index(key)
    raise LuaError("attempt to index a %s value (key %s)" % (lua_type(obj), tostr(key)))


def setindex(obj, key, val):
    if isinstance(obj, LuaTable):
        if obj.meta is not None and obj.get(key) is None:
            h = obj.meta.get("__newindex")
            if h is not None:
                if isinstance(h, LuaTable):
                    return setindex(h, key, val)
                call(h, [obj, key, val])
                return
        obj.set(key, val)
        return
    if hasattr(obj, "lua_newindex"):
        return obj.lua_newindex(key, val)
    raise LuaError("attempt to index a %s value (key %s)" % (lua_type(obj), tostr(key)))


def lua_eq(a, b):
    if isinstance(a, bool) or isinstance(b, bool):
        return a is b
    if a is None or b is None:
        # LuaJIT: a NULL cdata pointer compares equal to nil
        other = b if a is None else a
        if other is None:
            return True
        return bool(getattr(other, "lua_is_null", lambda: False)())
    if isinstance(a, (int, float)) and isinstance(b, (int, float)):
        return a == b
    if isinstance(a, str) and isinstance(b, str):
        return a == b
    if a is b:
        return True
    if hasattr(a, "lua_eq"):
        return a.lua_eq(b)
    if isinstance(a, LuaTable) and isinstance(b, LuaTable) and a.meta is not None and a.meta is b.meta and a.meta.get("__eq") is not None:
        r = call(a.meta.get("__eq"), [a, b])
        return truthy(r[0] if r else None)
    return False


def arith(op, a, b):
    for v in (a, b):
        if hasattr(v, "lua_arith"):              # cdata pointers: pointer arithmetic, not arithmetic on the address
            return v.lua_arith(op, a, b)
    x, y = tonum(a), tonum(b)
    if x is None or y is None:
        for v in (a, b):
            if hasattr(v, "lua_arith"):
                return v.lua_arith(op, a, b)
            if isinstance(v, LuaTable) and v.meta is not None:
                h = v.meta.get({"+": "__add", "-": "__sub", "*": "__mul", "/": "__div", "%": "__mod", "^": "__pow"}[op])
                if h is not None:
                    r = call(h, [a, b])
                    return r[0] if r else None
        raise LuaError("attempt to perform arithmetic on a %s value" % lua_type(a if x is None else b))
    if op == "+":
        return x + y
    if op == "-":
        return x - y
    if op == "*":
        return x * y
    if op == "/":
        return x / y if y != 0 else (math.inf if x > 0 else -math.inf if x < 0 else math.nan)
    if op == "%":
        return x - math.floor(x / y) * y if y != 0 else math.nan
    return x ** y


class Interpreter:
    def __init__(self, search_paths=(), env_vars=None):
        self.globals = LuaTable()
        self.search_paths = list(search_paths)
        self.env_vars = dict(env_vars or {})
        self.package = LuaTable()
        self.loaded = LuaTable()
        self.preload = LuaTable()
        self.package.set("loaded", self.loaded)
        self.package.set("preload", self.preload)
        self.stderr = []
        install_stdlib(self)

    # ---- modules
    def register(self, name, value):
        self.loaded.set(name, value)

    def require(self, name):
        v = self.loaded.get(name)
        if v is not None:
            return v
        pre = self.preload.get(name)
        if pre is not None:
            r = call(pre, [name])
            v = r[0] if r and r[0] is not None else True
            self.loaded.set(name, v)
            return v
        rel = name.replace(".", os.sep)
        for base in self.search_paths:
            for cand in (os.path.join(base, rel + ".lua"), os.path.join(base, rel, "init.lua")):
                if os.path.exists(cand):
                    r = self.run_file(cand, [name])
                    v = r[0] if r and r[0] is not None else True
                    if self.loaded.get(name) is None:
                        self.loaded.set(name, v)
                    return self.loaded.get(name)
        raise LuaError("module '%s' not found" % name)

    def run_file(self, path, args=()):
        return self.run(open(path).read(), os.path.basename(path), args)

    def run(self, src, chunk="chunk", args=()):
        body = parse(src, chunk)
        fn = LuaFunction([], True, body, None, self, chunk)
        return self.call_function(fn, list(args))

    # ---- evaluation
    def call_function(self, fn, args):
        env = Env(fn.env)
        for i, p in enumerate(fn.params):
            env.vars[p] = args[i] if i < len(args) else None
        if fn.vararg:
            env.vars["..."] = args[len(fn.params):]
        try:
            self.exec_block(fn.body, env)
        except ReturnValues as r:
            return r.values
        return []

    def exec_block(self, stmts, env):
        for s in stmts:
            self.exec_stmt(s, env)

    def exec_stmt(self, s, env):
        k = s[0]
        if k == "local":
            vals = self.eval_list(s[2], env)
            for i, n in enumerate(s[1]):
                env.vars[n] = vals[i] if i < len(vals) else None
        elif k == "assign":
            vals = self.eval_list(s[2], env)
            for i, t in enumerate(s[1]):
                self.assign(t, vals[i] if i < len(vals) else None, env)
        elif k == "exprstat":
            self.eval_multi(s[1], env)
        elif k == "if":
            for cond, body in s[1]:
                if truthy(self.eval(cond, env)):
                    self.exec_block(body, Env(env))
                    return
            if s[2] is not None:
                self.exec_block(s[2], Env(env))
        elif k == "while":
            try:
                while truthy(self.eval(s[1], env)):
                    self.exec_block(s[2], Env(env))
            except BreakLoop:
                pass
        elif k == "repeat":
            try:
                while True:
                    e = Env(env)
                    self.exec_block(s[1], e)
                    if truthy(self.eval(s[2], e)):
                        break
            except BreakLoop:
                pass
        elif k == "fornum":
            a, b = tonum(self.eval(s[2], env)), tonum(self.eval(s[3], env))
            c = tonum(self.eval(s[4], env)) if s[4] is not None else 1.0
            if a is None or b is None or c is None:
                raise LuaError("'for' limits must be numbers")
            try:
                i = a
                while (c > 0 and i <= b) or (c < 0 and i >= b):
                    e = Env(env)
                    e.vars[s[1]] = i
                    self.exec_block(s[5], e)
                    i += c
            except BreakLoop:
                pass
        elif k == "forin":
            vals = self.eval_list(s[2], env)
            f, st, ctl = (vals + [None, None, None])[:3]
            try:
                while True:
                    r = call(f, [st, ctl])
                    if not r or r[0] is None:
                        break
                    ctl = r[0]
                    e = Env(env)
                    for i, n in enumerate(s[1]):
                        e.vars[n] = r[i] if i < len(r) else None
                    self.exec_block(s[3], e)
            except BreakLoop:
                pass
        elif k == "do":
            self.exec_block(s[1], Env(env))
        elif k == "localfunc":
            env.vars[s[1]] = None
            env.vars[s[1]] = self.make_function(s[2], env, s[1])
        elif k == "return":
            raise ReturnValues(self.eval_list(s[1], env))
        elif k == "break":
            raise BreakLoop()
        else:
            raise LuaError("unknown statement %s" % k)

    def assign(self, target, value, env):
        if target[0] == "name":
            e = env.lookup(target[1])
            if e is not None:
                e.vars[target[1]] = value
            else:
                self.globals.set(target[1], value)
        elif target[0] == "index":
            setindex(self.eval(target[1], env), self.eval(target[2], env), value)
        else:
            raise LuaError("cannot assign to this expression")

    def make_function(self, node, env, name="?"):
        return LuaFunction(node[1], node[2], node[3], env, self, name)

    def eval_list(self, exprs, env):
        out = []
        for i, e in enumerate(exprs):
            if i == len(exprs) - 1:
                out.extend(self.eval_multi(e, env))
            else:
                out.append(self.eval(e, env))
        return out

    def eval_multi(self, e, env):
        k = e[0]
        if k == "call":
            f = self.eval(e[1], env)
            try:
                return call(f, self.eval_list(e[2], env))
            except LuaError as err:
                if not getattr(err, "located", False):
                    err.located = True
                    err.args = ("%s (line %s)" % (err.args[0], e[3]),)
                raise
        if k == "method":
            obj = self.eval(e[1], env)
            f = index(obj, e[2])
            if f is None:
                raise LuaError("attempt to call method '%s' (a nil value) (line %s)" % (e[2], e[4]))
            return call(f, [obj] + self.eval_list(e[3], env))
        if k == "vararg":
            ev = env.lookup("...")
            return list(ev.vars["..."]) if ev is not None else []
        return [self.eval(e, env)]

    def eval(self, e, env):
        k = e[0]
        if k == "const":
            return e[1]
        if k == "name":
            ev = env.lookup(e[1])
            if ev is not None:
                return ev.vars[e[1]]
            return self.globals.get(e[1])
        if k == "index":
            return index(self.eval(e[1], env), self.eval(e[2], env))
        if k in ("call", "method", "vararg"):
            r = self.eval_multi(e, env)
            return r[0] if r else None
        if k == "paren":
            return self.eval(e[1], env)
        if k == "function":
            return self.make_function(e, env)
        if k == "table":
            t = LuaTable()
            n = 0
            for i, item in enumerate(e[1]):
                if item[0] == "kv":
                    t.set(self.eval(item[1], env), self.eval(item[2], env))
                elif i == len(e[1]) - 1:
                    for v in self.eval_multi(item[1], env):
                        n += 1
                        t.set(n, v)
                else:
                    n += 1
                    t.set(n, self.eval(item[1], env))
            return t
        if k == "unop":
            v = self.eval(e[2], env)
            if e[1] == "not":
                return not truthy(v)
            if e[1] == "-":
                x = tonum(v)
                if x is None:
                    if hasattr(v, "lua_arith"):
                        return v.lua_arith("unm", v, None)
                    raise LuaError("attempt to perform arithmetic on a %s value" % lua_type(v))
                return -x
            if isinstance(v, str):
                return float(len(v))
            if isinstance(v, LuaTable):
                if v.meta is not None and v.meta.get("__len") is not None:
                    r = call(v.meta.get("__len"), [v])
                    return r[0] if r else None
                return float(v.length())
            if hasattr(v, "lua_len"):
                return v.lua_len()
            raise LuaError("attempt to get length of a %s value" % lua_type(v))
        if k == "binop":
            op = e[1]
            if op == "and":
                a = self.eval(e[2], env)
                return self.eval(e[3], env) if truthy(a) else a
            if op == "or":
                a = self.eval(e[2], env)
                return a if truthy(a) else self.eval(e[3], env)
            a, b = self.eval(e[2], env), self.eval(e[3], env)
            if op == "==":
                return lua_eq(a, b)
            if op == "~=":
                return not lua_eq(a, b)
            if op == "..":
                if isinstance(a, (str, int, float)) and isinstance(b, (str, int, float)) and not isinstance(a, bool) and not isinstance(b, bool):
                    return tostr(a) + tostr(b)
                for v in (a, b):
                    if isinstance(v, LuaTable) and v.meta is not None and v.meta.get("__concat") is not None:
                        r = call(v.meta.get("__concat"), [a, b])
                        return r[0] if r else None
                raise LuaError("attempt to concatenate a %s value (line %s)" % (lua_type(a if not isinstance(a, (str, int, float)) else b), e[4]))
            if op in ("<", "<=", ">", ">="):
                if op in (">", ">="):
                    a, b, op = b, a, "<" if op == ">" else "<="
                if isinstance(a, str) and isinstance(b, str):
                    return a < b if op == "<" else a <= b
                x, y = tonum(a) if not isinstance(a, str) else None, tonum(b) if not isinstance(b, str) else None
                if x is None or y is None:
                    raise LuaError("attempt to compare %s with %s (line %s)" % (lua_type(a), lua_type(b), e[4]))
                return x < y if op == "<" else x <= y
            try:
                return arith(op, a, b)
            except LuaError as err:
                raise LuaError("%s (line %s)" % (err.args[0], e[4]))
        raise LuaError("unknown expression %s" % k)


# ------------------------------------------------------------------------------------------------------------------------------- stdlib
STRING_LIB = {}


def lua_format(fmt, *args):
    out, ai = [], 0
    for m in re.finditer(r"%([-+ #0]*\d*(?:\.\d+)?)([diuoxXeEfgGcsq%])|[^%]+", fmt):
        if m.group(2) is None:
            out.append(m.group(0))
            continue
        spec, conv = m.group(1), m.group(2)
        if conv == "%":
            out.append("%")
            continue
        v = args[ai] if ai < len(args) else None
        ai += 1
        if conv in "diuoxX":
            out.append(("%" + spec + ("d" if conv in "iu" else conv)) % int(tonum(v)))
        elif conv in "eEfgG":
            out.append(("%" + spec + conv) % tonum(v))
        elif conv == "c":
            out.append(chr(int(tonum(v))))
        elif conv == "q":
            out.append('"' + tostr(v).replace("\\", "\\\\").replace('"', '\\"') + '"')
        else:
            out.append(("%" + spec + "s") % tostr(v))
    return "".join(out)


def install_stdlib(I):
    G = I.globals

    def lua_error(msg=None, level=None):
        raise LuaError(msg)

    def lua_assert(*a):
        if not a or not truthy(a[0]):
            raise LuaError(a[1] if len(a) > 1 else "assertion failed!")
        return list(a)

    def lua_pcall(f, *a):
        try:
            return [True] + call(f, list(a))
        except LuaError as e:
            return [False, e.value if not isinstance(e.value, str) else e.args[0]]

    def lua_next(t, k=None):
        keys = list(t.hash.keys())
        if k is None:
            i = 0
        else:
            if isinstance(k, float) and k == int(k):
                k = int(k)
            i = keys.index(k) + 1
        if i >= len(keys):
            return [None]
        kk = keys[i]
        return [float(kk) if isinstance(kk, int) and not isinstance(kk, bool) else kk, t.hash[kk]]

    def lua_pairs(t):
        if not isinstance(t, LuaTable):
            raise LuaError("bad argument #1 to 'pairs' (table expected, got %s)" % lua_type(t))
        snapshot = list(t.hash.items())
        state = {"i": 0}

        def it(_s=None, _c=None):
            while state["i"] < len(snapshot):
                k, v = snapshot[state["i"]]
                state["i"] += 1
                if k in t.hash:
                    return [float(k) if isinstance(k, int) and not isinstance(k, bool) else k, t.hash[k]]
            return [None]
        return [it, t, None]

    def lua_ipairs(t):
        def it(tt, i):
            i = int(i) + 1
            v = index(tt, i)
            return [None] if v is None else [float(i), v]
        return [it, t, 0.0]

    def lua_select(n, *a):
        if n == "#":
            return float(len(a))
        n = int(tonum(n))
        return list(a[n - 1:]) if n > 0 else list(a[n:])

    def lua_unpack(t, i=1, j=None):
        i = int(tonum(i))
        j = t.length() if j is None else int(tonum(j))
        return [t.get(k) for k in range(i, j + 1)]

    def lua_setmetatable(t, mt):
        t.meta = mt
        return t

    def lua_getmetatable(t):
        if isinstance(t, LuaTable):
            return t.meta
        if isinstance(t, str):
            m = LuaTable()
            m.set("__index", G.get("string"))
            return m
        return None

    def lua_tostring(v=None):
        return tostr(v)

    def lua_tonumber(v=None, base=None):
        return tonum(v, base)

    def lua_print(*a):
        I.stderr.append("\t".join(tostr(x) for x in a))

    for name, f in (("error", lua_error), ("assert", lua_assert), ("pcall", lua_pcall), ("next", lua_next), ("pairs", lua_pairs), ("ipairs", lua_ipairs),
                    ("select", lua_select), ("unpack", lua_unpack), ("setmetatable", lua_setmetatable), ("getmetatable", lua_getmetatable),
                    ("tostring", lua_tostring), ("tonumber", lua_tonumber), ("type", lua_type), ("print", lua_print),
                    ("rawget", lambda t, k: t.get(k)), ("rawset", lambda t, k, v: (t.set(k, v), t)[1]), ("rawequal", lambda a, b: a is b or (a == b and type(a) == type(b))),
                    ("require", I.require)):
        G.set(name, f)
    G.set("_G", G)
    G.set("package", I.package)

    # string
    S = LuaTable()

    def s_sub(s, i=1, j=-1):
        n = len(s)
        i, j = int(tonum(i)), int(tonum(j))
        if i < 0:
            i = max(n + i + 1, 1)
        elif i == 0:
            i = 1
        if j < 0:
            j = n + j + 1
        elif j > n:
            j = n
        return s[i - 1:j] if i <= j else ""

    def s_find(s, pat, init=1, plain=None):
        i = s.find(pat, int(tonum(init)) - 1)      # literal patterns only
        return [None] if i < 0 else [float(i + 1), float(i + len(pat))]

    def s_gsub(s, pat, repl, n=None):
        if any(c in pat for c in "^$()%.[]*+-?") and not (len(pat) == 2 and pat[0] == "%"):
            if pat == "\n":
                pass
            else:
                raise LuaError("minilua: gsub with a Lua pattern is not implemented: %r" % pat)
        lit = pat[1] if len(pat) == 2 and pat[0] == "%" else pat
        cnt = s.count(lit)
        if isinstance(repl, str):
            return [s.replace(lit, repl.replace("%%", "%")), float(cnt)]
        raise LuaError("minilua: gsub with a function / table replacement is not implemented")

    for name, f in (("format", lua_format), ("sub", s_sub), ("len", lambda s: float(len(s))), ("rep", lambda s, n, sep="": sep.join([s] * int(tonum(n)))),
                    ("lower", lambda s: s.lower()), ("upper", lambda s: s.upper()), ("find", s_find), ("gsub", s_gsub),
                    ("byte", lambda s, i=1: float(ord(s[int(tonum(i)) - 1])) if s else None), ("char", lambda *a: "".join(chr(int(tonum(x))) for x in a)),
                    ("reverse", lambda s: s[::-1])):
        S.set(name, f)
        STRING_LIB[name] = f
    G.set("string", S)

    # table
    T = LuaTable()

    def t_insert(t, a, b=None):
        n = t.length()
        if b is None:
            t.set(n + 1, a)
        else:
            pos = int(tonum(a))
            for k in range(n, pos - 1, -1):
                t.set(k + 1, t.get(k))
            t.set(pos, b)

    def t_remove(t, pos=None):
        n = t.length()
        if n == 0:
            return None
        pos = n if pos is None else int(tonum(pos))
        v = t.get(pos)
        for k in range(pos, n):
            t.set(k, t.get(k + 1))
        t.set(n, None)
        return v

    def t_sort(t, cmp=None):
        import functools
        n = t.length()
        items = [t.get(k) for k in range(1, n + 1)]
        if cmp is None:
            items.sort()
        else:
            items.sort(key=functools.cmp_to_key(lambda a, b: -1 if truthy((call(cmp, [a, b]) or [None])[0]) else (1 if truthy((call(cmp, [b, a]) or [None])[0]) else 0)))
        for k, v in enumerate(items):
            t.set(k + 1, v)

    def t_concat(t, sep="", i=1, j=None):
        j = t.length() if j is None else int(tonum(j))
        return sep.join(tostr(t.get(k)) for k in range(int(tonum(i)), j + 1))

    for name, f in (("insert", t_insert), ("remove", t_remove), ("sort", t_sort), ("concat", t_concat)):
        T.set(name, f)
    G.set("table", T)

    # math
    M = LuaTable()
    for name, f in (("floor", lambda x: float(math.floor(tonum(x)))), ("ceil", lambda x: float(math.ceil(tonum(x)))), ("abs", lambda x: abs(tonum(x))),
                    ("max", lambda *a: max(tonum(x) for x in a)), ("min", lambda *a: min(tonum(x) for x in a)), ("sqrt", lambda x: math.sqrt(tonum(x))),
                    ("sin", lambda x: math.sin(tonum(x))), ("cos", lambda x: math.cos(tonum(x))), ("log", lambda x: math.log(tonum(x))),
                    ("exp", lambda x: math.exp(tonum(x))), ("fmod", lambda a, b: math.fmod(tonum(a), tonum(b))), ("pow", lambda a, b: tonum(a) ** tonum(b))):
        M.set(name, f)
    for name, f in (("tan", math.tan), ("atan", math.atan), ("asin", math.asin), ("acos", math.acos), ("log10", math.log10), ("sinh", math.sinh), ("cosh", math.cosh)):
        M.set(name, (lambda fn: lambda x: fn(tonum(x)))(f))
    M.set("atan2", lambda a, b: math.atan2(tonum(a), tonum(b)))
    M.set("pi", math.pi)
    M.set("huge", math.inf)
    G.set("math", M)

    # os / io
    O = LuaTable()
    O.set("getenv", lambda name: I.env_vars.get(name))
    O.set("time", lambda *a: float(int(__import__("time").time())))
    O.set("clock", lambda: __import__("time").process_time())
    G.set("os", O)
    IO = LuaTable()
    err = LuaTable()
    err.set("write", lambda self, *a: I.stderr.append("".join(tostr(x) for x in a)))
    IO.set("stderr", err)
    out = LuaTable()
    out.set("write", lambda self, *a: I.stderr.append("".join(tostr(x) for x in a)))
    IO.set("stdout", out)
    G.set("io", IO)
