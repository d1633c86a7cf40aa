#!/usr/bin/env python3
"""List every public prototype of the reference headers the drop-in boundary cites and write the
inventory to tests/golden/reference_prototypes.json: name, header:line, and the prototype reduced to its types
("double(geodesic*,double)").  tests/test_capi_boundary.py holds sim5_amd/host/sim5lib.{h,c} to that list, minus its
explicit out-of-scope names.

Runs only in the build container (reads /root/reference/src/*.h as text).  TEST INFRASTRUCTURE ONLY.

    python oracle/list_prototypes.py
"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("SIM5_REFERENCE", "/root/reference")
HEADERS = ["sim5kerr-geod.h", "sim5kerr.h", "sim5raytrace.h", "sim5disk-nt.h", "sim5polarization.h", "sim5radiation.h",
           "sim5elliptic.h", "sim5math.h", "sim5polyroots.h"]


def reduce_signature(ret, args):
    """'double', 'geodesic *g, double P' -> 'double(geodesic*,double)'; array parameters decay to pointers"""
    def one(a):
        a = a.strip()
        if a in ("", "void"):
            return None
        dims = len(re.findall(r"\[[^\]]*\]", a))
        a = re.sub(r"\[[^\]]*\]", "", a)
        a = re.sub(r"\bconst\b", "", a)
        toks = re.findall(r"[A-Za-z_][A-Za-z_0-9]*|\*", a)
        stars = toks.count("*")
        words = [t for t in toks if t != "*"]
        if len(words) > 1 and words[-1] not in ("int", "double", "float", "long", "char", "unsigned", "sim5complex", "sim5metric",
                                                 "sim5tetrad", "geodesic", "raytrace_data"):
            words = words[:-1]                              # drop the parameter name
        ptr = stars + (1 if dims else 0)                    # T x[4][4][4] is passed as T (*)[4][4]: one level for the ABI
        return " ".join(words) + "*" * ptr
    parts = [p for p in (one(a) for a in args.split(",")) if p is not None]
    ret = " ".join(re.findall(r"[A-Za-z_][A-Za-z_0-9]*|\*", re.sub(r"\b(DEVICEFUNC|HOSTFUNC|INLINE|const)\b", "", ret)))
    return "%s(%s)" % (ret.replace(" *", "*"), ",".join(parts))


def prototypes(path, marked=True):
    """marked: the reference's headers prefix every prototype with DEVICEFUNC / HOSTFUNC; unmarked: plain C declarations"""
    txt = open(path).read()
    txt = re.sub(r"/\*.*?\*/", lambda m_: "\n" * m_.group(0).count("\n"), txt, flags=re.S)
    txt = re.sub(r"//[^\n]*", "", txt)
    out = []
    pat = (r"((?:DEVICEFUNC|HOSTFUNC)[\s\w\*]*?)\b([A-Za-z_]\w*)\s*\(([^()]*)\)\s*;" if marked else
           r"^[ \t]*((?:unsigned |long |const )*[A-Za-z_]\w*[\s\*]+?)([A-Za-z_]\w*)\s*\(([^()]*)\)\s*;")
    for m_ in re.finditer(pat, txt, flags=re.M):
        line = txt.count("\n", 0, m_.start(2)) + 1
        out.append({"name": m_.group(2), "line": line, "signature": reduce_signature(m_.group(1), m_.group(3))})
    return out


def main():
    inv = []
    for h in HEADERS:
        for p in prototypes(os.path.join(REF, "src", h)):
            p["header"] = "src/" + h
            if not any(q["name"] == p["name"] for q in inv):          # geodesic_P_int is declared twice
                inv.append(p)
    path = os.path.join(ROOT, "tests", "golden", "reference_prototypes.json")
    with open(path, "w") as fh:
        json.dump({"headers": ["src/" + h for h in HEADERS], "prototypes": inv}, fh, indent=1)
    print("%d prototypes -> %s" % (len(inv), path))
    return inv


if __name__ == "__main__":
    sys.exit(0 if main() else 1)
