#!/usr/bin/env python3
"""One-off source clean-up: removes the timing-experiment switches (results-invalid `LWS_DBG_*` builds and a few A/B knobs)
from lws_systolic.hip, keeping the production branch of each.  usage: strip_experiments.py file"""
import re, sys

UNDEF = {"LWS_DBG_TIMING", "LWS_DBG_NOSTORE", "LWS_DBG_NOFLOW", "LWS_SLEEP", "LWS_DBG_NOPUBLISH", "LWS_DBG_NOPROJECT",
         "LWS_PROJECT_NEWTON", "LWS_DBG_NOCARRY", "LWS_NO_PRIO", "LWS_DBG_ONLYROW", "LWS_DBG_NOPROLOG", "LWS_DBG_NOAMP",
         "LWS_DBG_NONYQ", "LWS_DBG_NOLOADER", "LWS_NO_R13", "LWS_NO_K0REAL"}
ZERO = {"LWS_DBG_NOLDS", "LWS_DBG_NOMATH", "LWS_DBG_NOIMG", "LWS_DBG_NOWRAP2", "LWS_DBG_NOSEL", "LWS_DBG_NOCPATCH"}
ONE = {"LWS_QUAD", "LWS_SERVICE_WAVE"}
KNOWN = UNDEF | ZERO | ONE


def decide(line):
    """None: unknown conditional (keep); else True/False for the branch taken"""
    m = re.match(r"\s*#\s*(ifdef|ifndef|if)\s+(!?)\s*(\w+)\s*(//.*)?$", line)
    if not m or m.group(3) not in KNOWN:
        return None
    kind, neg, name = m.group(1), m.group(2), m.group(3)
    if kind == "ifdef":
        return name not in UNDEF          # ZERO / ONE macros are "defined" once their default definition is folded
    if kind == "ifndef":
        return name in UNDEF
    val = 0 if (name in UNDEF or name in ZERO) else 1
    return bool(val) != bool(neg)


def strip(text):
    out, stack = [], []          # stack entries: [known(bool), emitting_now(bool), parent_emitting(bool)]
    lines = text.split("\n")
    i = 0
    while i < len(lines):
        ln = lines[i]
        emitting = all(e[1] for e in stack)
        # default definitions "#ifndef X / #define X v / #endif" of folded macros disappear entirely
        m = re.match(r"\s*#\s*ifndef\s+(\w+)", ln)
        if m and m.group(1) in (ZERO | ONE) and i + 2 < len(lines) and re.match(r"\s*#\s*define\s+" + m.group(1) + r"\b", lines[i + 1]) \
                and re.match(r"\s*#\s*endif", lines[i + 2]):
            i += 3
            continue
        if re.match(r"\s*#\s*(ifdef|ifndef|if)\b", ln):
            d = decide(ln)
            if d is None:
                stack.append([False, True, emitting])
                if emitting:
                    out.append(ln)
            else:
                stack.append([True, d, emitting])
            i += 1
            continue
        if re.match(r"\s*#\s*else\b", ln) and stack:
            if stack[-1][0]:
                stack[-1][1] = not stack[-1][1]
            elif all(e[1] for e in stack):
                out.append(ln)
            i += 1
            continue
        if re.match(r"\s*#\s*endif\b", ln) and stack:
            top = stack.pop()
            if not top[0] and all(e[1] for e in stack):
                out.append(ln)
            i += 1
            continue
        if emitting:
            out.append(ln)
        i += 1
    return "\n".join(out)


if __name__ == "__main__":
    path = sys.argv[1]
    src = open(path).read()
    res = strip(src)
    for name in ZERO:
        res = re.sub(r"\b" + name + r"\b", "0", res)
    for name in ONE:
        res = re.sub(r"\b" + name + r"\b", "1", res)
    open(path, "w").write(res)
