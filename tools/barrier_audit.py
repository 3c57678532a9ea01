#!/usr/bin/env python3
"""Every __syncthreads() of the library with the function it is in and the control statements that enclose it (brace
matching on the source text): the raw material of the barrier audit in DESIGN.md — a workgroup barrier is only safe where
every wave of the workgroup reaches it, i.e. where the enclosing conditions are uniform over the workgroup.
  python tools/barrier_audit.py > profiles/r06_barrier_audit.txt"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "rust-kzg_amd", "csrc")


def strip(src):
    src = re.sub(r"//[^\n]*", lambda m: " " * len(m.group(0)), src)
    src = re.sub(r"/\*.*?\*/", lambda m: re.sub(r"[^\n]", " ", m.group(0)), src, flags=re.S)
    src = re.sub(r'"(\\.|[^"\\\n])*"', lambda m: '"' + " " * (len(m.group(0)) - 2) + '"', src)
    src = re.sub(r"^[ \t]*#[^\n]*", lambda m: " " * len(m.group(0)), src, flags=re.M)
    return src


def audit(path):
    raw = open(path).read()
    src = strip(raw)
    out = []
    for m in re.finditer(r"__syncthreads\(\)", src):
        pos = m.start()
        line = src.count("\n", 0, pos) + 1
        # walk back, collecting the statement in front of every unmatched '{'
        depth, i, heads = 0, pos, []
        while i > 0:
            i -= 1
            c = src[i]
            if c == "}":
                depth += 1
            elif c == "{":
                if depth:
                    depth -= 1
                    continue
                j = i
                # the head: back to the previous ';', '{' or '}' at this level
                k, par = j - 1, 0
                while k > 0:
                    ch = src[k]
                    if ch == ")":
                        par += 1
                    elif ch == "(":
                        par -= 1
                    elif ch in ";{}" and par == 0:
                        break
                    k -= 1
                head = " ".join(src[k + 1:j].split())
                heads.append(head)
                if "__global__" in head or "__device__" in head or "FF_HD" in head:
                    break  # a function header
        fn = heads[-1] if heads else "?"
        name = re.search(r"(\w+)\s*\(", re.sub(r"__launch_bounds__\([^)]*\)", "", fn))
        lb = re.search(r"__launch_bounds__\(([^)]*)\)", fn)
        conds = [h for h in heads[:-1] if h and not h.startswith("else") or h.startswith("else if")]
        conds = [h for h in heads[:-1] if h]
        out.append((os.path.basename(path), line, name.group(1) if name else fn[:40], lb.group(1) if lb else ("__global__" in fn and "?" or "device fn"),
                    [c[:110] for c in reversed(conds)]))
    return out


def main():
    rows = []
    for f in sorted(os.listdir(CSRC)):
        if f.endswith((".hip", ".h")):
            rows += audit(os.path.join(CSRC, f))
    for f, line, fn, lb, conds in rows:
        print("%s:%d  %s  [%s]" % (f, line, fn, lb))
        for c in conds:
            print("      inside: %s" % c)
    print("%d barriers" % len(rows))


if __name__ == "__main__":
    main()
