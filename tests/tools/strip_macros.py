"""Remove preprocessor branches of macros that are never defined (a small unifdef): `#ifdef M`, `#ifndef M`, `#if defined(M)`,
`#elif defined(M)`, `#if !defined(A) && !defined(B)` with every operand in the given set, their `#else` / `#endif`; anything
else is left as it is.  Used once in round 5 to take the A/B and knock-out branches out of the shipped kernel sources (the
numbers they produced are in DESIGN.md / profiles/); kept as the record of how.
usage: python tests/tools/strip_macros.py MACRO[,MACRO...] file [file ...]"""
import re
import sys

undef = set(sys.argv[1].split(","))


def cond_value(line):
    """True / False if the directive's condition is decided by the undefined set, None otherwise"""
    t = re.sub(r"//.*$|/\*.*?\*/", "", line).strip()
    m = re.match(r"#\s*ifdef\s+(\w+)\s*$", t)
    if m:
        return False if m.group(1) in undef else None
    m = re.match(r"#\s*ifndef\s+(\w+)\s*$", t)
    if m:
        return True if m.group(1) in undef else None
    m = re.match(r"#\s*(?:if|elif)\s+(.*)$", t)
    if m:
        e = m.group(1).strip()
        names = re.findall(r"defined\s*\(\s*(\w+)\s*\)", e)
        rest = re.sub(r"!?\s*defined\s*\(\s*\w+\s*\)|&&|\|\||[()\s]", "", e)
        if names and not rest and all(n in undef for n in names):
            py = re.sub(r"defined\s*\(\s*\w+\s*\)", "False", e).replace("&&", " and ").replace("||", " or ").replace("!", " not ")
            return bool(eval(py))
    return None


for path in sys.argv[2:]:
    out = []
    # stack entries: [kind, emitting_parent, state] kind 'keep' (directive left in the text) or 'cut' (resolved)
    stack = []
    emitting = True
    for line in open(path):
        s = line.lstrip()
        if re.match(r"#\s*(if|ifdef|ifndef)\b", s):
            v = cond_value(s) if emitting else None
            if v is None:
                stack.append(["keep", emitting, None])
                if emitting:
                    out.append(line)
            else:
                stack.append(["cut", emitting, v])
                emitting = emitting and v
            continue
        if re.match(r"#\s*elif\b", s) and stack:
            top = stack[-1]
            if top[0] == "cut":
                if top[2]:                       # a branch was taken already
                    emitting = False
                else:
                    v = cond_value(s)
                    if v is None:
                        raise SystemExit("%s: #elif of a resolved #if with an open condition: %s" % (path, s.strip()))
                    top[2] = v
                    emitting = top[1] and v
                continue
            if emitting:
                v = cond_value(s)
                if v is False:
                    # an undecided chain with a dead #elif branch: drop the branch (up to the next #elif / #else / #endif)
                    stack[-1] = ["keep_dead", top[1], None]
                    emitting = False
                    continue
            elif top[0] == "keep_dead":
                stack[-1] = ["keep", top[1], None]
                emitting = top[1]
            if emitting:
                out.append(line)
            continue
        if re.match(r"#\s*else\b", s) and stack:
            top = stack[-1]
            if top[0] == "cut":
                emitting = top[1] and not top[2]
                top[2] = True
                continue
            if top[0] == "keep_dead":
                stack[-1] = ["keep", top[1], None]
                emitting = top[1]
            if emitting:
                out.append(line)
            continue
        if re.match(r"#\s*endif\b", s) and stack:
            top = stack.pop()
            if top[0] == "keep_dead":
                emitting = top[1]
                if emitting:
                    out.append(line)
                continue
            if top[0] == "cut":
                emitting = top[1]
                continue
            emitting = top[1]
            if emitting:
                out.append(line)
            continue
        if emitting:
            out.append(line)
    open(path, "w").write("".join(out))
    print("%s: %d lines" % (path, len(out)))
