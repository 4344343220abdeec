#!/usr/bin/env python3
"""Partial evaluation of preprocessor conditionals: flags in FIXED are replaced by their value in #if / #elif expressions
(`defined(X)` -> 1; flags in UNDEF: `defined(X)` -> 0, bare X -> 0); a conditional that evaluates to a constant keeps only its live
branch.  Conditionals that still mention other identifiers are left untouched (reported)."""
import re
import sys

FIXED = {
    "FLUX_BVH4_PERM": 1, "FLUX_BVH4_SORT": 0, "FLUX_BVH4_WAVE_TOTAL": 1, "FLUX_SHADE_TWO_PHASE": 1, "FLUX_RELOAD_PARAMS": 1,
    "FLUX_SCALAR_VOTES": 1, "FLUX_BVH4_TYP": 1, "FLUX_SPLIT_TYP": 1, "FLUX_SPLIT_MAX32": 1, "FLUX_EXP2_ARGS": 1, "FLUX_SET_ROWS": 1,
    "FLUX_SPLIT_UNIFORM_SUB": 1, "FLUX_SCALAR_LIVE": 0, "FLUX_SPLIT_OPAQUE_UNIFORMS": 3, "FLUX_SPLIT_PIXEL_CONSTS": 2,
    "FLUX_SPLIT_RELOAD_PARAMS": 1, "FLUX_BVH4_MAT_LIST": 0, "FLUX_BVH4_RELOAD_PARAMS": 1, "FLUX_BVH4_EARLY_REFILL": 1,
    "FLUX_BVH4_ENTRY": 1, "FLUX_BVH_LEAF_VOTE": 1, "FLUX_STRICT_BOX_HWMINMAX": 1, "FLUX_BVH4_LDS_SCENE": 1, "FLUX_TRI_FDIV": 0,
    "FLUX_TRAV_RCP32": 1, "FLUX_SPLIT_EARLY_SAMPLES": 0, "FLUX_SPLIT_LDS_SCENE": 2, "FLUX_STRICT_FILTER": 1,
    "FLUX_HEMI_AOS4": 1, "FLUX_SELF_SKIP": 1, "FLUX_Z_SLAB_RULE": 1, "FLUX_UNIT_DIRS": 1, "FLUX_ENV_SHORT": 1, "FLUX_BVH_WIDE": 1,
    "FLUX_SET_GROUPED": 1, "FLUX_GLOSS_TABLE": 1, "FLUX_FILTER32": 1, "FLUX_PRIMARY_UNIT": 1, "FLUX_BVH4_ARENA": 1,
}
UNDEF = {"FLUX_EXP_NO_PINS", "FLUX_EXP_SHFL_SUM", "FLUX_EXP_LDS_PAD_SPLIT", "FLUX_EXP_LDS_PAD", "FLUX_EXP_HIP_VOTES", "FLUX_EXP_EXTRA_SALU",
         "FLUX_EXP_VOTE_MACROS", "FLUX_EXP_HEMI_CMJ", "FLUX_EXP_NO_LAUNCH", "FLUX_NO_STAGE"}


def strip_comment(e):
    return re.sub(r"//.*$", "", re.sub(r"/\*.*?\*/", "", e)).strip()


def evaluate(expr):
    """-> True / False / None (not decidable)"""
    e = strip_comment(expr)
    def sub_defined(m):
        n = m.group(1)
        if n in FIXED:
            return "1"
        if n in UNDEF:
            return "0"
        return m.group(0)
    e = re.sub(r"defined\s*\(\s*(\w+)\s*\)", sub_defined, e)
    e = re.sub(r"defined\s+(\w+)", sub_defined, e)
    def sub_name(m):
        n = m.group(0)
        if n in FIXED:
            return str(FIXED[n])
        if n in UNDEF:
            return "0"
        return n
    e = re.sub(r"\b[A-Za-z_]\w*\b", sub_name, e)
    if re.search(r"[A-Za-z_]", e):
        # partially known: short-circuit forms  `0 && X`, `1 || X`
        py = e.replace("&&", " and ").replace("||", " or ")
        py = re.sub(r"!(?!=)", " not ", py)
        names = set(re.findall(r"\b[A-Za-z_]\w*\b", py)) - {"and", "or", "not"}
        res = set()
        for val in (0, 1):
            try:
                env = {n: val for n in names}
                res.add(bool(eval(py, {"__builtins__": {}}, env)))
            except Exception:
                return None
        # only if the result is the same for unknowns = 0 and = 1 AND the expression is purely boolean in the unknowns
        if len(res) == 1 and not re.search(r"[<>=&|^+\-*/](?![&|])", re.sub(r"&&|\|\||==|!=|<=|>=", "", e).replace("!", "")):
            return res.pop()
        return None
    py = e.replace("&&", " and ").replace("||", " or ")
    py = re.sub(r"!(?!=)", " not ", py)
    try:
        return bool(eval(py, {"__builtins__": {}}, {}))
    except Exception:
        return None


def process(lines):
    out = []
    # stack entries: dict(kind: 'keep'|'resolved', taken: bool (a branch already emitted), emitting: bool)
    stack = []
    undecided = []
    def emitting():
        return all(s["emitting"] for s in stack)
    i = 0
    while i < len(lines):
        line = lines[i]
        # join continuation lines of a directive
        m = re.match(r"^\s*#\s*(if|ifdef|ifndef|elif|else|endif)\b(.*)$", line.rstrip("\n"))
        if not m:
            if emitting():
                out.append(line)
            i += 1
            continue
        d, rest = m.group(1), m.group(2)
        if d in ("if", "ifdef", "ifndef"):
            if d == "ifdef":
                n = strip_comment(rest).split()[0]
                v = True if n in FIXED else (False if n in UNDEF else None)
            elif d == "ifndef":
                n = strip_comment(rest).split()[0]
                v = False if n in FIXED else (True if n in UNDEF else None)
            else:
                v = evaluate(rest)
            outer = emitting()
            if v is None:
                stack.append({"kind": "keep", "emitting": True, "outer": outer})
                if outer:
                    out.append(line)
                    undecided.append((i + 1, line.strip()))
            else:
                stack.append({"kind": "resolved", "emitting": v, "taken": v, "outer": outer, "became_keep": False})
        elif d == "elif":
            s = stack[-1]
            if s["kind"] == "keep":
                if all(x["emitting"] for x in stack[:-1]):
                    out.append(line)
            else:
                if s["taken"]:
                    s["emitting"] = False
                else:
                    v = evaluate(rest)
                    if v is None:
                        # becomes an open conditional: rewrite as #if
                        if all(x["emitting"] for x in stack[:-1]):
                            out.append(re.sub(r"#\s*elif", "#if", line, count=1))
                            undecided.append((i + 1, line.strip()))
                        s["kind"] = "keep"
                        s["emitting"] = True
                    else:
                        s["emitting"] = v
                        s["taken"] = v
        elif d == "else":
            s = stack[-1]
            if s["kind"] == "keep":
                if all(x["emitting"] for x in stack[:-1]):
                    out.append(line)
            else:
                s["emitting"] = not s["taken"]
                s["taken"] = True
        else:  # endif
            s = stack.pop()
            if s["kind"] == "keep" and emitting():
                out.append(line)
        i += 1
    assert not stack
    return out, undecided


if __name__ == "__main__":
    for path in sys.argv[1:]:
        lines = open(path).read().splitlines(keepends=True)
        out, und = process(lines)
        open(path, "w").write("".join(out))
        print(path, len(lines), "->", len(out), "lines;", len(und), "conditionals left")
        for n, l in und:
            print("   ", n, l[:110])
