#!/usr/bin/env python3
"""Symbol-by-symbol check of the two (unbuilt) Rust crates against the reference source tree.

There is no Rust toolchain in this image, so nothing compiles rust/zkw-shim and rust/zkw-refdump.  This script is the
next best thing: it resolves every `zk_evm::...` path the crates import or spell out, every associated function they call
on an in-tree reference type, every method they call and every field they touch, against /root/reference/src — the file
and line where the item is declared `pub`, and its signature — and fails on anything it cannot find or whose arity does
not match.  Paths that lead into the two absent crates (`zkevm_opcode_defs`, `zk_evm_abstractions`, which the reference
re-exports) are listed as UNVERIFIABLE with the reference line that re-exports or itself uses them.

  python rust/check_api.py            # writes rust/API_CHECK.md, exit code 1 on any unverified in-tree symbol
  python rust/check_api.py --check    # same checks, compares with the committed rust/API_CHECK.md instead of writing

tests/test_rust_api_check.py runs it in the CPU suite (skipped where /root/reference does not exist).
"""
import os
import re
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("ZKW_REFERENCE", "/root/reference")
SRC = os.path.join(REF, "src")
CRATES = {"zkw-shim": ["src/lib.rs", "src/ffi.rs"], "zkw-refdump": ["src/main.rs"]}
EXTERNAL_ROOTS = {  # re-exports of absent crates at the reference's crate root (lib.rs)
    "zkevm_opcode_defs": "pub use zkevm_opcode_defs;",
    "zk_evm_abstractions": "pub use zk_evm_abstractions;",
    "ethereum_types": "pub use zkevm_opcode_defs::{bitflags, ethereum_types};",
    "blake2": "pub use zkevm_opcode_defs::blake2;",
    "aux_structures": "pub mod aux_structures {",
    "abstractions": "pub mod abstractions {",
}


# ---------------------------------------------------------------------------------------------------------------------
# index of the reference
# ---------------------------------------------------------------------------------------------------------------------
def strip_comments(text):
    text = re.sub(r"/\*.*?\*/", lambda m: "\n" * m.group(0).count("\n"), text, flags=re.S)
    return "\n".join(re.sub(r"//.*", "", l) for l in text.split("\n"))


def module_of(path):
    rel = os.path.relpath(path, SRC)[:-3]
    parts = rel.split(os.sep)
    if parts[-1] in ("mod", "lib"):
        parts = parts[:-1]
    return "::".join(parts)


class Item:
    def __init__(self, kind, name, module, file, line, sig, owner=None, is_pub=True):
        self.kind, self.name, self.module, self.file, self.line, self.sig, self.owner, self.is_pub = kind, name, module, file, line, sig, owner, is_pub

    def where(self):
        return "%s:%d" % (os.path.relpath(self.file, SRC), self.line)


def signature(lines, i):
    """declaration text from line i up to the opening brace / semicolon, whitespace-normalised"""
    out = []
    depth = 0
    for j in range(i, min(i + 40, len(lines))):
        l = lines[j].replace("->", "\u2192").replace("=>", "\u21d2")  # arrows are not brackets
        cut = None
        for k, ch in enumerate(l):
            if ch in "(<[":
                depth += 1
            elif ch in ")>]":
                depth -= 1
            elif ch in "{;" and depth <= 0:
                cut = k
                break
            elif ch == "=" and depth <= 0 and k + 1 < len(l) and l[k + 1] not in "=>" and (k == 0 or l[k - 1] not in "=!<>-"):
                # `const X: T = value` / `type X = ...`: keep the left side
                if re.match(r"\s*(pub(\([a-z]+\))?\s+)?(const|static|type)\b", lines[i]):
                    cut = k
                    break
        if cut is not None:
            out.append(l[:cut])
            break
        out.append(l)
    return re.sub(r"\s+", " ", " ".join(out)).strip().replace("\u2192", "->").replace("\u21d2", "=>")


def split_top(s, sep=","):
    parts, depth, cur = [], 0, ""
    s = s.replace("->", "\u2192").replace("=>", "\u21d2")
    for ch in s:
        if ch in "(<[{":
            depth += 1
        elif ch in ")>]}":
            depth -= 1
        if ch == sep and depth == 0:
            parts.append(cur)
            cur = ""
        else:
            cur += ch
    if cur.strip():
        parts.append(cur)
    return [p.strip().replace("\u2192", "->").replace("\u21d2", "=>") for p in parts]


def fn_params(sig):
    m = re.search(r"\bfn\s+\w+\s*(<.*?>)?\s*\(", sig)
    if not m:
        return None
    # find the matching parenthesis of the parameter list
    start = sig.index("(", m.start() + sig[m.start():].index("fn"))
    # the first '(' after the generics
    depth, k = 0, None
    gen = 0
    i = m.end() - 1
    depth = 0
    for j in range(i, len(sig)):
        if sig[j] == "(":
            depth += 1
        elif sig[j] == ")":
            depth -= 1
            if depth == 0:
                k = j
                break
    params = split_top(sig[i + 1:k])
    return params


def index_reference():
    items = {}    # "module::Name" -> Item (module-level)
    members = {}  # type or trait name -> {member name -> Item}
    globs = {}    # module -> [modules re-exported with `pub use ...::*`]
    files = []
    for root, _, fs in os.walk(SRC):
        for f in fs:
            if f.endswith(".rs"):
                files.append(os.path.join(root, f))
    for path in sorted(files):
        mod = module_of(path)
        raw = open(path).read()
        lines = strip_comments(raw).split("\n")
        depth = 0
        ctx = []  # stack of (depth_at_open, kind, name)
        opened = None  # a declaration whose opening brace has not been seen yet (multi-line generics / where clauses)
        for i, l in enumerate(lines):
            s = l.strip()
            cur = ctx[-1] if ctx else None
            at_mod_level = all(c[1] == "mod" for c in ctx)
            cur_mod = "::".join([mod] + [c[2] for c in ctx if c[1] == "mod"]).strip(":")
            m = re.match(r"(pub(\([a-z]+\))?\s+)?(?:(?:const|unsafe|async|extern\s+\"C\")\s+)*(struct|enum|trait|fn|const|static|type|mod)\s+(\w+)", s)
            mu = re.match(r"pub\s+use\s+(.*?);", s)
            if mu and at_mod_level:
                tgt = mu.group(1).strip()
                if tgt.endswith("::*"):
                    t = tgt[:-3]
                    t = re.sub(r"^(self|crate)::", "", t)
                    globs.setdefault(cur_mod, []).append((t if tgt.startswith("crate::") else ((cur_mod + "::" + t).strip(":") if tgt.startswith("self::") else t), i + 1, path))
                else:
                    # `pub use a::b::{X, Y}` / `pub use a::b::X;` — record as aliases
                    base, _, rest = tgt.rpartition("::")
                    names = split_top(rest.strip("{}")) if rest.startswith("{") else [rest]
                    for n in names:
                        n = n.strip()
                        if n:
                            items[(cur_mod + "::" + n.split(" as ")[-1]).strip(":")] = Item("use", n, cur_mod, path, i + 1, s)
            if m:
                is_pub = bool(m.group(1))
                kind, name = m.group(3), m.group(4)
                sig = signature(lines, i)
                if at_mod_level:
                    if is_pub or kind == "mod":
                        items[(cur_mod + "::" + name).strip(":")] = Item(kind, name, cur_mod, path, i + 1, sig, is_pub=is_pub)
                elif cur and cur[1] in ("impl", "trait") and kind in ("fn", "const", "type"):
                    # methods of trait impls are public through the trait; inherent ones need `pub`
                    if is_pub or cur[1] == "trait" or cur[3]:
                        members.setdefault(cur[2], {}).setdefault(name, Item(kind, name, cur_mod, path, i + 1, sig, owner=cur[2]))
                if kind in ("struct", "enum", "trait", "mod") and "{" in "".join(lines[i:i + 12]) and not s.rstrip().endswith(";"):
                    opened = (kind, name, False)
            mi = re.match(r"(unsafe\s+)?impl\b(.*)", s)
            if mi:
                head = signature(lines, i)
                head = re.sub(r"^(unsafe\s+)?impl\s*(<.*?>\s*)?", "", head) if not head.startswith("impl<") else head
                # strip the generic parameter list of the impl itself
                h = signature(lines, i)
                h = re.sub(r"^(unsafe\s+)?impl", "", h).strip()
                if h.startswith("<"):
                    d = 0
                    for k, ch in enumerate(h):
                        if ch == "<":
                            d += 1
                        elif ch == ">":
                            d -= 1
                            if d == 0:
                                h = h[k + 1:].strip()
                                break
                h = h.split(" where ")[0]
                is_trait_impl = " for " in h
                target = h.split(" for ")[-1].strip()
                tm = re.match(r"&?\s*(?:mut\s+)?([\w:]+)", target)
                if tm:
                    opened = ("impl", tm.group(1).split("::")[-1], is_trait_impl)
            # fields of a pub struct
            if cur and cur[1] == "struct" and depth == cur[0] + 1:
                mf = re.match(r"pub\s+(\w+)\s*:\s*(.*?),?$", s)
                if mf:
                    members.setdefault(cur[2], {})[mf.group(1)] = Item("field", mf.group(1), cur_mod, path, i + 1, "pub %s: %s" % (mf.group(1), mf.group(2)), owner=cur[2])
            if cur and cur[1] == "enum" and depth == cur[0] + 1:
                mv = re.match(r"(\w+)\s*(\(|\{|,|=|$)", s)
                if mv and not s.startswith("#"):
                    members.setdefault(cur[2], {})[mv.group(1)] = Item("variant", mv.group(1), cur_mod, path, i + 1, s.rstrip(","), owner=cur[2])
            # brace tracking
            for ch in l:
                if ch == "{":
                    if opened:
                        ctx.append((depth, opened[0], opened[1], opened[2]))
                        opened = None
                    depth += 1
                elif ch == "}":
                    depth -= 1
                    if ctx and ctx[-1][0] == depth:
                        ctx.pop()
    return items, members, globs


def resolve(path, items, globs, seen=None):
    """`a::b::Name` (relative to the crate root) -> Item, following `pub use x::*` re-exports"""
    if path in items:
        it = items[path]
        if it.kind == "use":
            return it
        return it
    mod, _, name = path.rpartition("::")
    seen = seen or set()
    for (g, line, file) in globs.get(mod, []):
        if (g, name) in seen:
            continue
        seen.add((g, name))
        r = resolve((g + "::" + name).strip(":"), items, globs, seen)
        if r:
            return r
    return None


# ---------------------------------------------------------------------------------------------------------------------
# the crates
# ---------------------------------------------------------------------------------------------------------------------
def expand_use(tree, prefix=""):
    tree = tree.strip()
    m = re.match(r"^(.*?)::\{(.*)\}$", tree, flags=re.S)
    if m and tree.count("{") >= 1 and tree.index("{") == len(m.group(1)) + 2:
        out = []
        for part in split_top(m.group(2)):
            out += expand_use(part, prefix + m.group(1) + "::")
        return out
    return [prefix + tree]


RUST_KEYWORDS_OR_STD = set("""
len iter map collect unwrap unwrap_or unwrap_or_default expect push pop clone cloned to_vec to_le_bytes from_le_bytes copy_from_slice
extend_from_slice try_into into to_string to_string_lossy into_owned as_ptr as_mut_ptr as_str as_bytes chunks enumerate position min max
take get insert contains resize reverse find is_empty last first sum flat_map chain once rev zip filter any all and_then ok_or
write_all read_to_end save section open create to_fixed_bytes from_ptr add saturating_sub clear iter_mut null_mut null as_ref
starts_with ends_with trim split join format contains_key entry or_default keys values to_owned as_slice concat fill swap
""".split())


def scan_crate(name, files):
    uses, inline_paths, aliases, calls, assoc, fields, impls, own = [], [], {}, [], [], [], [], set()
    for rel in files:
        path = os.path.join(HERE, name, rel)
        text = strip_comments(open(path).read())
        # strip string literals
        text_ns = re.sub(r'"(\\.|[^"\\])*"', '""', text)
        for m in re.finditer(r"\buse\s+(zk_evm(::[^;]*)?);", text_ns, flags=re.S):
            tree = re.sub(r"\s+", " ", m.group(1))
            line = text_ns[:m.start()].count("\n") + 1
            for p in expand_use(tree):
                p = p.strip()
                alias = None
                if " as " in p:
                    p, alias = [x.strip() for x in p.split(" as ")]
                uses.append((p, rel, line))
                aliases[alias or p.split("::")[-1]] = p
        for m in re.finditer(r"(?<![\w:])((?:zk_evm|defs)(?:::\w+)+)", text_ns):
            line = text_ns[:m.start()].count("\n") + 1
            if re.match(r"\s*use\b", text_ns.split("\n")[line - 1]):
                continue
            inline_paths.append((m.group(1), rel, line))
        for m in re.finditer(r"\bfn\s+(\w+)", text_ns):
            own.add(m.group(1))
        for m in re.finditer(r"\b(?:pub\s+)?(\w+)\s*:\s*[\w&\[<*(']", text_ns):
            own.add(m.group(1))  # struct fields / let bindings of the crates themselves
        for m in re.finditer(r"\b([A-Z]\w*)(?:::<[^>]*>)?::(\w+)\s*(\()?", text_ns):
            line = text_ns[:m.start()].count("\n") + 1
            args = None
            if m.group(3):
                args = call_args(text_ns, m.end() - 1)
            assoc.append((m.group(1), m.group(2), args, rel, line))
        for m in re.finditer(r"\.\s*(\w+)\s*(::<[^>]*>)?\s*(\()?", text_ns):
            line = text_ns[:m.start()].count("\n") + 1
            if m.group(1)[0].isdigit():
                continue
            if m.group(3):
                calls.append((m.group(1), call_args(text_ns, m.end() - 1), rel, line))
            else:
                fields.append((m.group(1), rel, line))
        for m in re.finditer(r"\bimpl\s*(<[^{]*?>)?\s*([\w:]+)\s*(<[^{]*?>)?\s+for\s+(\w+)[^{]*\{", text_ns):
            # body of the impl
            start = m.end() - 1
            d, k = 0, start
            for k in range(start, len(text_ns)):
                if text_ns[k] == "{":
                    d += 1
                elif text_ns[k] == "}":
                    d -= 1
                    if d == 0:
                        break
            body = text_ns[start:k]
            line0 = text_ns[:m.start()].count("\n") + 1
            for fm in re.finditer(r"\bfn\s+(\w+)\s*(<[^(]*>)?\s*\(", body):
                sig = signature(body[fm.start():].split("\n"), 0)
                impls.append((m.group(2).split("::")[-1], m.group(4), fm.group(1), fn_params(sig), rel, line0 + body[:fm.start()].count("\n")))
    return dict(uses=uses, inline=inline_paths, aliases=aliases, calls=calls, assoc=assoc, fields=fields, impls=impls, own=own)


def call_args(text, open_idx):
    d = 0
    for k in range(open_idx, min(len(text), open_idx + 4000)):
        if text[k] in "([{":
            d += 1
        elif text[k] in ")]}":
            d -= 1
            if d == 0:
                inner = text[open_idx + 1:k]
                return len([a for a in split_top(inner) if a])
    return None


def normalise_type(t):
    t = re.sub(r"\s+", "", t)
    t = t.replace("<N,E>", "<8,E>")
    return t


def main():
    check_only = "--check" in sys.argv
    if not os.path.isdir(SRC):
        print("reference tree not found at %s" % SRC)
        return 2
    items, members, globs = index_reference()
    out = ["# API_CHECK — the Rust crates against `/root/reference/src` (generated by `rust/check_api.py`)", "",
           "No Rust toolchain exists in the build image; this is a symbol-by-symbol resolution of what `rust/zkw-shim` and",
           "`rust/zkw-refdump` use from `zk_evm` v1.4.1 against the reference's own source.  `file:line` is relative to",
           "`/root/reference/src`.  UNVERIFIABLE = the path leaves the reference tree through one of its re-exports of the two",
           "absent crates (`zkevm_opcode_defs`, `zk_evm_abstractions` @ branch v1.4.1); the line shown is where the reference",
           "re-exports the crate or uses the same symbol itself.", ""]
    failures = []
    lib_rs = open(os.path.join(SRC, "lib.rs")).read().split("\n")

    def lib_line(snippet):
        for i, l in enumerate(lib_rs):
            if snippet in l:
                return i + 1
        return None

    def ref_usage(symbol):
        """a place where the reference itself names an external symbol"""
        for root, _, fs in os.walk(SRC):
            for f in sorted(fs):
                if not f.endswith(".rs"):
                    continue
                p = os.path.join(root, f)
                for i, l in enumerate(open(p).read().split("\n")):
                    if re.search(r"\b%s\b" % re.escape(symbol), l) and not l.strip().startswith("//"):
                        return "%s:%d" % (os.path.relpath(p, SRC), i + 1)
        return None

    for crate, files in CRATES.items():
        sc = scan_crate(crate, files)
        out += ["## `rust/%s`" % crate, "", "### Paths (`use zk_evm::…` and spelled-out paths)", "", "| path | used at | resolves to | declaration |", "|---|---|---|---|"]
        imported_types = {}  # local name -> in-tree type name
        seen_paths = set()
        all_paths = [(p, f, l) for (p, f, l) in sc["uses"]]
        for (p, f, l) in sc["inline"]:
            if p.startswith("defs::") and "defs" in sc["aliases"]:
                p = sc["aliases"]["defs"] + p[4:]
            all_paths.append((p, f, l))
        for (p, f, l) in all_paths:
            if p in seen_paths:
                continue
            seen_paths.add(p)
            segs = p.split("::")
            if segs[0] != "zk_evm":
                continue
            if len(segs) == 1:
                continue
            if segs[1] in EXTERNAL_ROOTS:
                ln = lib_line(EXTERNAL_ROOTS[segs[1]])
                if ln is None:
                    failures.append("re-export of `%s` not found in lib.rs" % segs[1])
                use = ref_usage(segs[-1]) if len(segs) > 2 else None
                out.append("| `%s` | %s:%d | UNVERIFIABLE (absent crate) | re-export lib.rs:%s%s |" % (p, f, l, ln, ("; the reference names `%s` at %s" % (segs[-1], use)) if use else "; **the reference never names `%s`**" % segs[-1] if len(segs) > 2 else ""))
                continue
            it = resolve("::".join(segs[1:]), items, globs)
            if it is None:
                failures.append("%s:%d: `%s` does not resolve in the reference" % (f, l, p))
                out.append("| `%s` | %s:%d | **NOT FOUND** | |" % (p, f, l))
                continue
            out.append("| `%s` | %s:%d | %s `%s` | %s — `%s` |" % (p, f, l, it.kind, it.name, it.where(), it.sig[:150]))
            if it.kind in ("struct", "enum", "trait"):
                imported_types[segs[-1]] = it.name
        for alias, p in sc["aliases"].items():
            segs = p.split("::")
            if len(segs) > 1 and segs[1] not in EXTERNAL_ROOTS:
                it = resolve("::".join(segs[1:]), items, globs)
                if it is not None and it.kind in ("struct", "enum", "trait"):
                    imported_types[alias] = it.name
        # types reachable through fields of the imported ones (e.g. Callstack via VmLocalState.callstack, Flags)
        in_tree_types = set(imported_types.values())
        for t in list(in_tree_types):
            for mem in members.get(t, {}).values():
                for w in re.findall(r"\b([A-Z]\w+)\b", mem.sig):
                    if w in members:
                        in_tree_types.add(w)
        out += ["", "### Associated functions and constructors on in-tree types", "", "| call | used at | declaration | arity |", "|---|---|---|---|"]
        done = set()
        for (t, fn, nargs, f, l) in sc["assoc"]:
            tt = imported_types.get(t)
            if tt is None or (t, fn) in done:
                continue
            done.add((t, fn))
            mem = members.get(tt, {}).get(fn)
            if mem is None:
                failures.append("%s:%d: `%s::%s` is not a public item of `%s` in the reference" % (f, l, t, fn, tt))
                out.append("| `%s::%s` | %s:%d | **NOT FOUND** | |" % (t, fn, f, l))
                continue
            ar = ""
            if mem.kind == "fn" and nargs is not None:
                ps = [p for p in (fn_params(mem.sig) or []) if p]
                expect = len([p for p in ps if not re.match(r"&?\s*(mut\s+)?self\b", p)])
                ar = "%d = %d" % (nargs, expect)
                if nargs != expect:
                    failures.append("%s:%d: `%s::%s` called with %d arguments, the reference takes %d (%s)" % (f, l, t, fn, nargs, expect, mem.where()))
                    ar = "**%d ≠ %d**" % (nargs, expect)
            out.append("| `%s::%s` | %s:%d | %s — `%s` | %s |" % (t, fn, f, l, mem.where(), mem.sig[:160], ar))
        # methods / fields
        out += ["", "### Methods called and fields touched (`.name`), resolved over the in-tree types in scope", "", "| member | kind | first use | owner(s) in the reference | arity |", "|---|---|---|---|---|"]
        own = sc["own"]
        seen = set()
        unknown = []
        for kind, lst in (("call", sc["calls"]), ("field", sc["fields"])):
            for entry in lst:
                if kind == "call":
                    n, nargs, f, l = entry
                else:
                    n, f, l = entry
                    nargs = None
                if (kind, n) in seen:
                    continue
                seen.add((kind, n))
                owners = [t for t in sorted(in_tree_types) if n in members.get(t, {}) and (members[t][n].kind == "fn") == (kind == "call")]
                if not owners:
                    if n in own or n in RUST_KEYWORDS_OR_STD:
                        continue
                    unknown.append((n, kind, f, l))
                    continue
                ar = ""
                if kind == "call" and nargs is not None:
                    expects = set()
                    for t in owners:
                        ps = [p for p in (fn_params(members[t][n].sig) or []) if p]
                        expects.add(len([p for p in ps if not re.match(r"&?\s*(mut\s+)?self\b", p)]))
                    ar = "%d ∈ %s" % (nargs, sorted(expects))
                    if nargs not in expects:
                        failures.append("%s:%d: `.%s(..)` called with %d arguments, the reference declares %s" % (f, l, n, nargs, sorted(expects)))
                        ar = "**%d ∉ %s**" % (nargs, sorted(expects))
                out.append("| `.%s` | %s | %s:%d | %s | %s |" % (n, kind, f, l, "; ".join("`%s` %s" % (t, members[t][n].where()) for t in owners), ar))
        # members that belong to neither an in-tree type, the crates themselves nor std: they are members of external-crate
        # types (U256, Address, OPCODES_TABLE entries, the query structs) — listed, unverifiable
        out += ["", "### Members of external-crate types (unverifiable here; where the reference uses the same member)", "", "| member | kind | first use | same member in the reference |", "|---|---|---|---|"]
        for (n, kind, f, l) in unknown:
            use = None
            for root, _, fs in os.walk(SRC):
                for ff in sorted(fs):
                    if ff.endswith(".rs") and use is None:
                        p = os.path.join(root, ff)
                        for i, ll in enumerate(open(p).read().split("\n")):
                            if re.search(r"\.\s*%s\b" % re.escape(n), ll) and not ll.strip().startswith("//"):
                                use = "%s:%d" % (os.path.relpath(p, SRC), i + 1)
                                break
            out.append("| `.%s` | %s | %s:%d | %s |" % (n, kind, f, l, use or "—"))
        # trait implementations: every method must exist in the trait with the same parameter list
        out += ["", "### Trait implementations for reference traits", "", "| impl | method | declared in the reference | parameters |", "|---|---|---|---|"]
        for (trait, ty, fn, params, f, l) in sc["impls"]:
            tname = imported_types.get(trait)
            if tname is None:
                continue
            mem = members.get(tname, {}).get(fn)
            if mem is None:
                failures.append("%s:%d: `impl %s for %s` defines `%s`, which the trait does not declare" % (f, l, trait, ty, fn))
                out.append("| `%s for %s` | `%s` | **NOT IN TRAIT** | |" % (trait, ty, fn))
                continue
            want = [normalise_type(p.split(":", 1)[1]) if ":" in p else normalise_type(p) for p in (fn_params(mem.sig) or [])]
            got = [normalise_type(p.split(":", 1)[1]) if ":" in p else normalise_type(p) for p in (params or [])]
            ok = want == got
            if not ok:
                failures.append("%s:%d: `%s::%s` parameters %s differ from the reference's %s (%s)" % (f, l, trait, fn, got, want, mem.where()))
            out.append("| `%s for %s` | `%s` | %s | %s |" % (trait, ty, fn, mem.where(), "identical (%d)" % len(want) if ok else "**%s ≠ %s**" % (got, want)))
        out.append("")
    out += ["## Result", "", ("**%d unverified in-tree symbol(s):**" % len(failures)) if failures else "Every in-tree symbol resolves; arities and trait signatures match.  0 failures.", ""]
    out += ["* " + x for x in failures]
    text = "\n".join(out) + "\n"
    target = os.path.join(HERE, "API_CHECK.md")
    if check_only:
        if not os.path.exists(target) or open(target).read() != text:
            print("rust/API_CHECK.md is stale: run python rust/check_api.py")
            return 3
    else:
        open(target, "w").write(text)
    for x in failures:
        print("FAIL", x)
    print("%d failures" % len(failures))
    return 1 if failures else 0


if __name__ == "__main__":
    sys.exit(main())
