"""Mechanical check of the UNCOMPILED Go shim against the C header (`ptr(id)` on an array).

No Go toolchain exists in this image, so `go vet` / `go build` cannot run. This tool does the part of their job that decides whether the
cgo calls can compile at all: for every `C.gpv_*( ... )` call site under bindings/go it derives the cgo type of each argument expression
with a small type inference over the Go source (conversions `C.T(x)`, pointer casts `(*C.T)(x)`, the helpers' declared return types,
`var` declarations, struct fields, function parameters, `make([]T, n)` slices, multi-value assignments from helper functions) and compares
it with the type cgo assigns to that parameter of the prototype in include/gpv.h (`const T*` -> `*C.T`, `void*` -> `unsafe.Pointer`,
`const void* const*` -> `*unsafe.Pointer`, `size_t` -> `C.size_t`, ...). It also checks the generic helper `ptr[T any](s []T)` is only
ever handed a slice (the round-4 regression), and that every header function is called somewhere.
Since round 5 also what a compiler rejects before type checking (`lexical_problems`): unbalanced brackets, imports that are never used, exported
names `pkg.Name` of the shim's own packages that the package does not declare, and exported methods / fields that no type of the shim declares
(receiver types are not resolved) -- that is how `Circuit.Dims()`, called by three packages and defined by none, was found.
And the argument COUNT of every call to one of the shim's own functions or methods (a method name only when all its declarations agree), the number
of values assigned from such a call, and variables declared with := that their function never mentions again.

    python tools/check_go_shim.py            # prints a summary, exit status 1 on any mismatch

tests/test_abi_cpu.py::test_go_shim_call_sites_match_the_header runs it.
"""
import re
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
GO_DIR = ROOT / "bindings" / "go"
NIL = "<nil>"
UNTYPED = "<untyped integer constant>"
INT_TYPES = {"C.int", "C.uint", "C.size_t", "C.int32_t", "C.uint32_t", "C.int64_t", "C.uint64_t", "C.uint8_t", "C.long", "C.ulong"}


# ---------------------------------------------------------------- header side
def header_prototypes(text=None):
    """{name: (return cgo type, [param cgo types], [param names])} for every function include/gpv.h declares."""
    if text is None:
        text = (ROOT / "include" / "gpv.h").read_text()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"//[^\n]*", " ", text)
    protos = {}
    for m in re.finditer(r"(?m)^\s*((?:const\s+)?[A-Za-z_][A-Za-z0-9_]*(?:\s*\*)*)\s*(gpv_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", text):
        ret, name, params = m.group(1), m.group(2), m.group(3)
        plist = [p.strip() for p in params.replace("\n", " ").split(",")] if params.strip() and params.strip() != "void" else []
        types, names = [], []
        for p in plist:
            t, nm = split_param(p)
            types.append(c_to_cgo(t))
            names.append(nm)
        protos[name] = (c_to_cgo(ret), types, names)
    return protos


def split_param(p):
    p = re.sub(r"\[[^\]]*\]", "", p).strip()  # `uint64_t out[4]` does not occur, but be safe
    m = re.match(r"^(.*?)([A-Za-z_][A-Za-z0-9_]*)$", p)
    if m and m.group(1).strip() and not m.group(1).strip().endswith(("const", "struct")):
        return m.group(1).strip(), m.group(2)
    return p, ""


def c_to_cgo(t):
    """The Go type cgo gives a C parameter type: const is dropped, `void*` is unsafe.Pointer, every other `T*` is `*C.T`."""
    t = re.sub(r"\bconst\b", " ", t)
    stars = t.count("*")
    base = " ".join(t.replace("*", " ").split())
    base = {"unsigned char": "uchar", "unsigned int": "uint", "unsigned": "uint", "unsigned long": "ulong", "long long": "longlong"}.get(base, base)
    if base == "void":
        if stars == 0:
            return "void"
        return "*" * (stars - 1) + "unsafe.Pointer"
    return "*" * stars + "C." + base


# ---------------------------------------------------------------- Go side
def strip_go_comments(src):
    out, i, n = [], 0, len(src)
    while i < n:
        c = src[i]
        if c == '"':
            j = i + 1
            while j < n and src[j] != '"':
                j += 2 if src[j] == "\\" else 1
            out.append('""')
            i = j + 1
        elif c == "`":
            j = src.index("`", i + 1)
            out.append('""')
            i = j + 1
        elif src.startswith("//", i):
            j = src.find("\n", i)
            i = n if j < 0 else j
        elif src.startswith("/*", i):
            j = src.find("*/", i + 2)
            seg = src[i:j + 2]
            out.append("\n" * seg.count("\n"))
            i = j + 2
        else:
            out.append(c)
            i += 1
    return "".join(out)


def split_args(s):
    args, depth, cur = [], 0, []
    for ch in s:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == "," and depth == 0:
            args.append("".join(cur).strip())
            cur = []
        else:
            cur.append(ch)
    last = "".join(cur).strip()
    if last:
        args.append(last)
    return args


def matching_paren(src, i):
    """index of the `)` that closes the `(` at src[i]"""
    depth = 0
    for j in range(i, len(src)):
        if src[j] == "(":
            depth += 1
        elif src[j] == ")":
            depth -= 1
            if depth == 0:
                return j
    raise ValueError("unbalanced parentheses")


GO_TYPE = r"(?:\*|\[\d*\])*(?:C\.\w+|unsafe\.Pointer|\w+(?:\.\w+)?)"


class GoFile:
    def __init__(self, path):
        self.path = path
        self.src = strip_go_comments(path.read_text())
        # top-level functions: name -> (params text, results text)
        self.funcs = {}
        self.func_spans = []  # (start, end, header text)
        for m in re.finditer(r"(?m)^func\s*(\([^)]*\))?\s*(\w+)(?:\[[^\]]*\])?\s*\(", self.src):
            open_i = m.end() - 1
            close_i = matching_paren(self.src, open_i)
            rest = self.src[close_i + 1:]
            rm = re.match(r"\s*(\([^)]*\)|[^{\n]*?)\s*\{", rest)
            results = rm.group(1).strip() if rm else ""
            self.funcs[m.group(2)] = (self.src[open_i + 1:close_i], results)
            self.func_spans.append((m.start(), m.group(1) or "", self.src[open_i + 1:close_i]))
        self.structs = {}
        for m in re.finditer(r"(?m)^type\s+(\w+)\s+struct\s*\{([^}]*)\}", self.src):
            fields = {}
            for line in m.group(2).replace(";", "\n").split("\n"):
                fm = re.match(r"\s*([\w, ]+?)\s+(" + GO_TYPE + r")\s*$", line)
                if fm:
                    for nm in fm.group(1).split(","):
                        fields[nm.strip()] = fm.group(2)
            self.structs[m.group(1)] = fields

    def enclosing(self, pos):
        """(receiver text, params text, body text up to pos) of the top-level function around pos"""
        best = None
        for start, recv, params in self.func_spans:
            if start <= pos:
                best = (start, recv, params)
        start, recv, params = best
        return recv, params, self.src[start:pos]


def result_types(results):
    results = results.strip()
    if not results:
        return []
    if results.startswith("("):
        return [r.strip().split()[-1] for r in split_args(results[1:-1])]
    return [results]


def param_types(params_text):
    """{name: type} for a Go parameter list `a, b []uint64, n int`"""
    out, pending = {}, []
    for p in split_args(params_text):
        parts = p.split()
        if len(parts) == 1:
            pending.append(parts[0])
        else:
            t = " ".join(parts[1:])
            for nm in pending + [parts[0]]:
                out[nm] = t
            pending = []
    return out


class Inference:
    def __init__(self, files):
        self.files = files
        self.funcs, self.structs = {}, {}
        for f in files:
            self.funcs.update(f.funcs)
            self.structs.update(f.structs)

    def var_type(self, name, gf, pos):
        recv, params, body = gf.enclosing(pos)
        # declared: var name T   /   var name T = ...
        m = None
        for m in re.finditer(r"\bvar\s+((?:\w+\s*,\s*)*\w+)\s+(" + GO_TYPE + r")", body):
            if name in [x.strip() for x in m.group(1).split(",")]:
                found = m.group(2)
                return found
        m = None
        # name := make([]T, ...)
        for m in re.finditer(r"\b" + re.escape(name) + r"\s*:?=\s*make\((\[\]" + GO_TYPE + r")\s*,", body):
            pass
        if m:
            return m.group(1)
        # closure: name := func(...) RET {
        m = re.search(r"\b" + re.escape(name) + r"\s*:=\s*func\s*\(([^)]*)\)\s*(" + GO_TYPE + r")\s*\{", body)
        if m:
            return "func:" + m.group(2)
        # a, b, c := helper(...)
        for m in re.finditer(r"(?m)^\s*([\w, ]+?)\s*:?=\s*(\w+)\(", body):
            names = [x.strip() for x in m.group(1).split(",")]
            if name in names and m.group(2) in self.funcs:
                rts = result_types(self.funcs[m.group(2)][1])
                if len(rts) == len(names):
                    return rts[names.index(name)]
        # name := C.T(...) / (*C.T)(...)
        m = re.search(r"\b" + re.escape(name) + r"\s*:=\s*([^\n]+)", body)
        if m:
            t = self.expr_type(m.group(1).strip(), gf, pos, depth=1)
            if t:
                return t
        # a parameter of the innermost closure around pos, then of the function
        for m in reversed(list(re.finditer(r"\bfunc\s*\(([^)]*)\)", body))):
            if body.count("{", m.end()) > body.count("}", m.end()):  # still inside that closure's body
                cp = param_types(m.group(1))
                if name in cp:
                    return cp[name]
        pt = param_types(params)
        if name in pt:
            return pt[name]
        rm = re.match(r"\(\s*(\w+)\s+(" + GO_TYPE + r")\s*\)", recv or "")
        if rm and rm.group(1) == name:
            return rm.group(2)
        return None

    def expr_type(self, e, gf, pos, depth=0):
        e = e.strip()
        if depth > 4:
            return None
        if e == "nil":
            return NIL
        if re.match(r"^(0x[0-9a-fA-F]+|\d+)$", e) or re.match(r"^C\.[A-Z][A-Z0-9_]*$", e):
            return UNTYPED  # integer literal, or a C enum / macro constant (cgo emits those as untyped Go constants)
        m = re.match(r"^C\.(\w+)\(", e)
        if m and matching_paren(e, m.end() - 1) == len(e) - 1:
            return "C." + m.group(1)
        m = re.match(r"^\((\*+)(C\.\w+|unsafe\.Pointer)\)\(", e)
        if m and matching_paren(e, m.end() - 1) == len(e) - 1:
            return m.group(1) + m.group(2)
        m = re.match(r"^(ptr|unsafe\.Pointer)\(", e)
        if m and matching_paren(e, m.end() - 1) == len(e) - 1:
            return "unsafe.Pointer"
        m = re.match(r"^&(\w+)\[0\]$", e)
        if m:
            t = self.var_type(m.group(1), gf, pos)
            return "*" + t[2:] if t and t.startswith("[]") else None
        m = re.match(r"^&(\w+)$", e)
        if m:
            t = self.var_type(m.group(1), gf, pos)
            return "*" + t if t else None
        m = re.match(r"^(\w+)\.(\w+)$", e)
        if m:
            t = self.var_type(m.group(1), gf, pos)
            if t:
                fields = self.structs.get(t.lstrip("*"), {})
                return fields.get(m.group(2))
            return None
        m = re.match(r"^(\w+)\(", e)
        if m and matching_paren(e, m.end() - 1) == len(e) - 1:
            name = m.group(1)
            t = self.var_type(name, gf, pos)
            if t and t.startswith("func:"):
                return t[5:]
            if name in self.funcs:
                rts = result_types(self.funcs[name][1])
                return rts[0] if len(rts) == 1 else None
            return None
        if re.match(r"^\w+$", e):
            return self.var_type(e, gf, pos)
        return None


def compatible(want, got):
    if got == NIL:
        return want.startswith("*") or want == "unsafe.Pointer"
    if got == UNTYPED:
        return want in INT_TYPES
    return want == got


def check(verbose=False, go_dir=None):
    go_dir = Path(go_dir) if go_dir else GO_DIR
    protos = header_prototypes()
    files = [GoFile(p) for p in sorted(go_dir.rglob("*.go"))]
    inf = Inference(files)
    problems, called, n_sites, n_args = [], set(), 0, 0
    for gf in files:
        rel = gf.path.relative_to(go_dir.parent.parent) if go_dir == GO_DIR else gf.path.relative_to(go_dir)
        for m in re.finditer(r"\bC\.(gpv_[a-z0-9_]+)\(", gf.src):
            name = m.group(1)
            line = gf.src.count("\n", 0, m.start()) + 1
            close = matching_paren(gf.src, m.end() - 1)
            args = split_args(gf.src[m.end():close])
            n_sites += 1
            if name not in protos:
                problems.append("%s:%d: C.%s is not declared in include/gpv.h" % (rel, line, name))
                continue
            called.add(name)
            ret, want, pnames = protos[name]
            if len(args) != len(want):
                problems.append("%s:%d: C.%s called with %d arguments, the header declares %d" % (rel, line, name, len(args), len(want)))
                continue
            for k, (a, w) in enumerate(zip(args, want)):
                n_args += 1
                got = inf.expr_type(a, gf, m.start())
                if got is None:
                    problems.append("%s:%d: C.%s argument %d (%s): cannot derive the type of `%s` (header wants %s)" % (rel, line, name, k + 1, pnames[k], a, w))
                elif not compatible(w, got):
                    problems.append("%s:%d: C.%s argument %d (%s): `%s` is %s, the header wants %s" % (rel, line, name, k + 1, pnames[k], a, got, w))
                elif verbose:
                    print("  ok %s:%d %s arg %d %s : %s" % (rel, line, name, k + 1, a, got))
        # the generic helper ptr[T any](s []T) infers T from a SLICE; an array argument does not compile
        for m in re.finditer(r"\bptr\((\w+)\)", gf.src):
            t = inf.var_type(m.group(1), gf, m.start())
            line = gf.src.count("\n", 0, m.start()) + 1
            if t is None:
                problems.append("%s:%d: ptr(%s): cannot derive the type of `%s`" % (rel, line, m.group(1), m.group(1)))
            elif not t.startswith("[]"):
                problems.append("%s:%d: ptr(%s): `%s` is %s, ptr[T any](s []T) needs a slice (use %s[:])" % (rel, line, m.group(1), m.group(1), t, m.group(1)))
    for name in protos:
        if name not in called:
            problems.append("include/gpv.h declares %s, no cgo call site under bindings/go" % name)
    return problems, {"header_functions": len(protos), "call_sites": n_sites, "arguments_checked": n_args}


# ---------------------------------------------------------------- lexical / cross-package checks (round 5)
def lexical_problems(go_dir=None):
    """What a Go compiler would reject before type checking, as far as it can be seen without one: unbalanced brackets, an imported package that is
    never used, and an exported name `pkg.Name` of one of the shim's own packages that the package does not declare (functions, types, constants,
    variables; methods and fields are not resolved)."""
    go_dir = Path(go_dir) if go_dir else GO_DIR
    problems = []
    files = {p: strip_go_comments(p.read_text()) for p in sorted(go_dir.rglob("*.go"))}
    declared = {}
    for path, src in files.items():
        names = declared.setdefault(path.parent.name, set())
        names.update(re.findall(r"^func (\w+)\s*[\[(]", src, re.M))
        names.update(re.findall(r"^type (\w+)", src, re.M))
        names.update(re.findall(r"^(?:const|var) (\w+)", src, re.M))
        for block in re.findall(r"^(?:const|var|type) \((.*?)^\)", src, re.M | re.S):
            names.update(re.findall(r"^\s+(\w+)", block, re.M))
    for path, src in files.items():
        rel = path.relative_to(go_dir)
        stack = []
        pairs = {")": "(", "]": "[", "}": "{"}
        for i, ch in enumerate(src):
            if ch in "([{":
                stack.append((ch, i))
            elif ch in ")]}":
                if not stack or stack[-1][0] != pairs[ch]:
                    problems.append("%s:%d: unbalanced `%s`" % (rel, src.count("\n", 0, i) + 1, ch))
                    stack = None
                    break
                stack.pop()
        if stack:
            problems.append("%s:%d: `%s` is never closed" % (rel, src.count("\n", 0, stack[-1][1]) + 1, stack[-1][0]))
        raw = path.read_text()
        imports = re.findall(r'^\s*(?:(\w+)\s+)?"([^"]+)"\s*$', "\n".join(re.findall(r"^import \((.*?)^\)", raw, re.M | re.S)), re.M)
        imports += [(a, q) for a, q in re.findall(r'^import (?:(\w+)\s+)?"([^"]+)"', raw, re.M)]
        for alias, ipath in imports:
            name = alias or ipath.rsplit("/", 1)[-1]
            if name in ("_", "C"):
                continue
            if not re.search(r"\b%s\." % re.escape(name), src):
                problems.append("%s: imports \"%s\" and never uses %s." % (rel, ipath, name))
            own = ipath.rsplit("/", 1)[-1]
            if "bindings/go/" in ipath and own in declared:
                for m in re.finditer(r"\b%s\.([A-Z]\w*)" % re.escape(name), src):
                    if m.group(1) not in declared[own]:
                        problems.append("%s:%d: %s.%s is not declared in package %s" % (rel, src.count("\n", 0, m.start()) + 1, name, m.group(1), own))
    # exported METHODS and FIELDS: `x.Name(` / `x.Name` on a value must be a method or a field that some type of the shim declares (receiver types are
    # not resolved), unless x is an imported package. Catches a helper that several packages call and nobody defined.
    methods, fields = set(), set()
    for src in files.values():
        methods.update(re.findall(r"^func \([^)]*\) (\w+)\s*\(", src, re.M))
        for body in re.findall(r"^type \w+ struct\s*\{(.*?)^\}", src, re.M | re.S) + re.findall(r"^type \w+ struct\s*\{([^\n}]*)\}", src, re.M):
            for line in body.replace(";", "\n").split("\n"):
                m = re.match(r"\s*((?:\w+\s*,\s*)*\w+)\s+[\w\[\]*.(){}]", line)
                if m:
                    fields.update(x.strip() for x in m.group(1).split(","))
                else:
                    m = re.match(r"\s*\*?(?:\w+\.)?(\w+)\s*$", line)   # embedded type
                    if m:
                        fields.add(m.group(1))
        for block in re.findall(r"interface\s*\{(.*?)\}", src, re.S):
            methods.update(re.findall(r"^\s*(\w+)\s*\(", block, re.M))
    STD_METHODS = {"Error", "String", "Add", "Done", "Wait", "Lock", "Unlock", "Unmarshal", "Decode", "Token", "Len", "Bytes", "Write", "WriteString", "Seconds",
                   "Pointer", "Slice", "SliceData", "Sizeof", "Cmp", "SetString", "Uint64", "IsUint64", "Int64", "Text", "Fatalf", "Errorf", "Helper", "Run", "Skip", "Logf",
                   "Fatal", "SetUint64", "Lsh", "Or", "Sign", "Mod", "Set",                                     # testing.T, math/big.Int
                   "Instructions", "Levels", "Blueprints", "BlueprintID", "DecompressHint", "HintID", "Unpack"}           # gnark v0.9.1 constraint.System (witness/adapter_gnark_v0_9.go)
    for path, src in files.items():
        rel = path.relative_to(go_dir)
        raw = path.read_text()
        aliases = {a or q.rsplit("/", 1)[-1] for a, q in re.findall(r'^\s*(?:(\w+)\s+)?"([^"]+)"\s*$', "\n".join(re.findall(r"^import \((.*?)^\)", raw, re.M | re.S)), re.M)}
        aliases |= {a or q.rsplit("/", 1)[-1] for a, q in re.findall(r'^import (?:(\w+)\s+)?"([^"]+)"', raw, re.M)} | {"C"}
        for m in re.finditer(r"(\w+|\)|\])\.([A-Z]\w*)(\s*\()?", src):
            recv, name, call = m.group(1), m.group(2), m.group(3)
            if recv in aliases:
                continue
            if name in methods or name in fields or name in STD_METHODS:
                continue
            problems.append("%s:%d: .%s%s: no type of the shim declares such a %s" % (rel, src.count("\n", 0, m.start()) + 1, name, "()" if call else "", "method" if call else "field"))
    # argument COUNTS of calls to the shim's own functions and methods (a method name is checked only when every type that declares it takes the same number)
    def n_params(text):
        ps = split_args(text)
        return len(ps), bool(ps) and "..." in ps[-1]

    pkg_funcs, method_sigs = {}, {}
    for path, src in files.items():
        for m in re.finditer(r"(?m)^func\s*(\([^)]*\))?\s*(\w+)(?:\[[^\]]*\])?\s*\(", src):
            close = matching_paren(src, m.end() - 1)
            sig = n_params(src[m.end():close])
            if m.group(1):
                method_sigs.setdefault(m.group(2), set()).add(sig)
            else:
                pkg_funcs.setdefault(path.parent.name, {})[m.group(2)] = sig
    for path, src in files.items():
        rel = path.relative_to(go_dir)
        raw = path.read_text()
        own = {}
        for a, q in re.findall(r'^\s*(?:(\w+)\s+)?"([^"]+)"\s*$', "\n".join(re.findall(r"^import \((.*?)^\)", raw, re.M | re.S)), re.M):
            if "bindings/go/" in q:
                own[a or q.rsplit("/", 1)[-1]] = q.rsplit("/", 1)[-1]
        for m in re.finditer(r"(?:(\w+|\)|\])\.)?\b([A-Za-z_]\w*)\(", src):
            recv, name = m.group(1), m.group(2)
            line_start = src.rfind("\n", 0, m.start()) + 1
            if re.match(r"\s*func\b", src[line_start:m.start()]) :   # a declaration, not a call
                continue
            if recv is None:
                sig = pkg_funcs.get(path.parent.name, {}).get(name)
                if src[max(0, m.start() - 1)] == ".":
                    continue
            elif recv in own:
                sig = pkg_funcs.get(own[recv], {}).get(name)
            else:
                sigs = method_sigs.get(name, set())
                sig = next(iter(sigs)) if len(sigs) == 1 and name[0].isupper() else None
            if sig is None:
                continue
            close = matching_paren(src, m.end() - 1)
            args = split_args(src[m.end():close])
            if len(args) == 1 and args[0].endswith(")") and sig[0] > 1:   # f(g()) with a multi-value g
                continue
            want, variadic = sig
            if (variadic and len(args) < want - 1) or (not variadic and len(args) != want):
                problems.append("%s:%d: %s%s() called with %d argument(s), its declaration takes %d" % (rel, src.count("\n", 0, m.start()) + 1, (recv + ".") if recv else "", name, len(args), want))
    # RESULT counts: `a, b := f(...)` against the declaration of f (own functions; methods when all declarations agree), and variables that are
    # declared with := and never mentioned again in their function (Go rejects both)
    def n_results(text):
        return len(result_types(text))

    fn_results, method_results = {}, {}
    for path, src in files.items():
        gf = GoFile(path)
        for m in re.finditer(r"(?m)^func\s*(\([^)]*\))?\s*(\w+)(?:\[[^\]]*\])?\s*\(", src):
            res = n_results(gf.funcs[m.group(2)][1]) if m.group(2) in gf.funcs else None
            if res is None:
                continue
            if m.group(1):
                method_results.setdefault(m.group(2), set()).add(res)
            else:
                fn_results.setdefault(path.parent.name, {})[m.group(2)] = res
    for path, src in files.items():
        rel = path.relative_to(go_dir)
        raw = path.read_text()
        own = {}
        for a, q in re.findall(r'^\s*(?:(\w+)\s+)?"([^"]+)"\s*$', "\n".join(re.findall(r"^import \((.*?)^\)", raw, re.M | re.S)), re.M):
            if "bindings/go/" in q:
                own[a or q.rsplit("/", 1)[-1]] = q.rsplit("/", 1)[-1]
        for m in re.finditer(r"(?m)^\s*((?:[\w.\[\]]+\s*,\s*)*[\w.\[\]]+)\s*:?=\s*(?:(\w+)\.)?(\w+)\(", src):
            lhs = [x.strip() for x in m.group(1).split(",")]
            recv, name = m.group(2), m.group(3)
            close = matching_paren(src, m.end() - 1)
            if src[close + 1:close + 2] not in ("\n", "", " ", "\t", "}") or src[close + 1:].lstrip(" \t").startswith((".", "[", "+", "-", "*", "/", "&", "|", "<", ">", "=", "!", "%")):
                continue   # the call is only part of the right-hand side
            if recv is None:
                want = fn_results.get(path.parent.name, {}).get(name)
            elif recv in own:
                want = fn_results.get(own[recv], {}).get(name)
            else:
                rs = method_results.get(name, set())
                want = next(iter(rs)) if len(rs) == 1 and name[0].isupper() else None
            if want is not None and want != len(lhs) and not (want == 0):
                problems.append("%s:%d: %d value(s) assigned from %s%s(), which returns %d" % (rel, src.count("\n", 0, m.start()) + 1, len(lhs), (recv + ".") if recv else "", name, want))
        gf = GoFile(path)
        spans = sorted(st for st, _, _ in gf.func_spans) + [len(src)]
        for a, b in zip(spans, spans[1:]):
            body = src[a:b]
            for m in re.finditer(r"(?m)(?:^|[;{]|\bif\s|\bfor\s|\bswitch\s)\s*((?:\w+\s*,\s*)*\w+)\s*:=", body):
                for nm in [x.strip() for x in m.group(1).split(",")]:
                    if nm == "_":
                        continue
                    uses = len(re.findall(r"\b%s\b" % re.escape(nm), body))
                    if uses < 2:
                        problems.append("%s:%d: `%s` is declared and never used" % (rel, src.count("\n", 0, a + m.start(1)) + 1, nm))
    # bare calls `name(...)`: a function or type of the package, a builtin, or something local to the function (:=, var, a parameter, a closure)
    builtin = set("len cap make new append copy delete panic recover print println min max close complex real imag int uint uint8 uint16 uint32 uint64 int8 int16 "
                  "int32 int64 uintptr float32 float64 string bool byte rune error func if for switch return go defer select case var type const range else chan "
                  "map struct interface".split())
    for path, src in files.items():
        rel = path.relative_to(go_dir)
        gf = GoFile(path)
        no_iface = re.sub(r"interface\s*\{.*?\}", lambda m: " " * len(m.group(0)) if "\n" not in m.group(0) else re.sub(r"[^\n]", " ", m.group(0)), src, flags=re.S)
        spans = sorted((st, params) for st, _, params in gf.func_spans) + [(len(src), "")]
        for (a, params), (b, _) in zip(spans, spans[1:]):
            body = no_iface[a:b]
            local = set(param_types(params))
            for m in re.finditer(r"((?:\w+\s*,\s*)*\w+)\s*:=", body):
                local.update(x.strip() for x in m.group(1).split(","))
            local |= set(re.findall(r"\bvar\s+(\w+)", body)) | set(re.findall(r"(\w+)\s+func\(", body[:max(0, body.find("{"))]))
            for m in re.finditer(r"\bfunc\s*\(([^)]*)\)", body):   # parameters of closures
                local |= set(param_types(m.group(1)))
            for m in re.finditer(r"(?<![\w.\])])\b([A-Za-z_]\w*)\(", body):
                name = m.group(1)
                ls = body.rfind("\n", 0, m.start()) + 1
                if re.match(r"\s*func\b", body[ls:m.start()]) and "{" not in body[ls:m.start()]:
                    continue
                if name in builtin or name in declared.get(path.parent.name, ()) or name in local:
                    continue
                problems.append("%s:%d: %s() is neither declared in package %s nor local to the function" % (rel, src.count("\n", 0, a + m.start()) + 1, name, path.parent.name))
    # composite literals of the shim's struct types: a positional literal names every field, a keyed one only fields that exist
    struct_fields = {}
    for path, src in files.items():
        for m in re.finditer(r"(?ms)^type (\w+) struct\s*\{(.*?)\}", src):
            fl = []
            for line in m.group(2).replace(";", "\n").split("\n"):
                line = line.strip()
                if not line:
                    continue
                fm = re.match(r"((?:\w+\s*,\s*)*\w+)\s+\S", line)
                fl += [x.strip() for x in fm.group(1).split(",")] if fm else [line.lstrip("*").split(".")[-1]]
            struct_fields[(path.parent.name, m.group(1))] = fl
    for path, src in files.items():
        rel = path.relative_to(go_dir)
        for m in re.finditer(r"(?:(\w+)\.)?\b(\w+)\{", src):
            key = (m.group(1) or path.parent.name, m.group(2))
            ls = src.rfind("\n", 0, m.start()) + 1
            if key not in struct_fields or re.match(r"\s*type\b", src[ls:m.start()]) or src[:m.start()].rstrip().endswith("]"):
                continue   # (`[]T{{a}, {b}}` is a slice literal with elided element types)
            depth, j = 0, m.end() - 1
            for j in range(m.end() - 1, len(src)):
                depth += src[j] == "{"
                depth -= src[j] == "}"
                if depth == 0:
                    break
            args = split_args(src[m.end():j])
            keyed = [a for a in args if re.match(r"^\w+\s*:", a)]
            line = src.count("\n", 0, m.start()) + 1
            if args and not keyed and len(args) != len(struct_fields[key]):
                problems.append("%s:%d: %s{...} lists %d value(s), the struct has %d field(s)" % (rel, line, m.group(2), len(args), len(struct_fields[key])))
            for a in keyed:
                if a.split(":")[0].strip() not in struct_fields[key]:
                    problems.append("%s:%d: %s{...} names the field %s, which the struct does not have" % (rel, line, m.group(2), a.split(":")[0].strip()))
    return problems


def main():
    lex = lexical_problems()
    for p in lex:
        print(p)
    if lex:
        print("check_go_shim: %d lexical / cross-package problem(s)" % len(lex))
        return 1
    problems, stats = check(verbose="-v" in sys.argv)
    for p in problems:
        print(p)
    print("check_go_shim: %(header_functions)d header functions, %(call_sites)d cgo call sites, %(arguments_checked)d arguments checked" % stats,
          "-- %d problem(s)" % len(problems))
    return 1 if problems else 0


if __name__ == "__main__":
    sys.exit(main())
