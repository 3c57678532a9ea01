#!/usr/bin/env python3
"""Static opcode histogram of a kernel of the library: the gfx950 code object is cut out of the .hip_fatbin section of a
host object (clang offload bundle), disassembled with llvm-objdump, and the VALU instructions of the named kernel are
counted by issue-cost class.  With the per-class issue cycles tools/ffbench.hip measured (profiles/r03_ffbench.log) this
gives the opcode-weighted issue bound bench.py prints next to the counter-based VALU fraction.

  python tools/isa_histogram.py [--object rust-kzg_amd/csrc/msm.o] [--kernel k_fbw_accum] > profiles/r05_isa_histogram.json"""
import argparse
import collections
import json
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
# issue-cost classes (profiles/r03_ffbench.log: wall cycles per wave-instruction at the nominal 2.4 GHz, 8 waves per SIMD)
CLASSES = [
    ("mad64", re.compile(r"^v_mad_(u64_u32|i64_i32)$"), "v_mad_u64_u32 (1 chain)"),
    ("mul32", re.compile(r"^v_mul_(lo|hi)_u32$"), "v_mul_lo_u32"),
    ("wide_or_three_operand", re.compile(r"^v_(lshrrev_b64|lshlrev_b64|lshl_add_u64|add3_u32|alignbit_b32|and_or_b32|lshl_or_b32|"
                                         r"lshl_add_u32|add_lshl_u32|bfe_u32|bfi_b32|mad_u32_u24|xad_u32|or3_b32|perm_b32|"
                                         r"cndmask_b32|addc_co_u32|subb_co_u32|add_co_u32|sub_co_u32|sad_u32|mov_b64).*$"), "v_add3_u32"),
    ("two_operand_32", re.compile(r"^v_.*$"), "v_add_u32"),
]


def code_object(obj):
    with tempfile.TemporaryDirectory() as td:
        fat = os.path.join(td, "fat.bin")
        subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", obj, fat])
        d = open(fat, "rb").read()
    assert d[:24] == b"__CLANG_OFFLOAD_BUNDLE__", "not an uncompressed clang offload bundle"
    n = struct.unpack_from("<Q", d, 24)[0]
    off = 32
    for _ in range(n):
        o, s, ts = struct.unpack_from("<QQQ", d, off)
        off += 24
        triple = d[off:off + ts].decode()
        off += ts
        if "gfx950" in triple:
            return d[o:o + s]
    raise SystemExit("no gfx950 code object in " + obj)


def ffbench_cycles(path):
    cyc = {}
    for line in open(path):
        m = re.match(r"^(.*?)\s+[\d.]+ ms\s+=>\s+([\d.]+) cyc/wave-instr", line)
        if m:
            cyc[m.group(1).strip()] = float(m.group(2))
    return cyc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--object", default=os.path.join(ROOT, "rust-kzg_amd", "csrc", "msm.o"))
    ap.add_argument("--kernel", default="k_fbw_accum")
    ap.add_argument("--ffbench", default=os.path.join(ROOT, "profiles", "r03_ffbench.log"))
    a = ap.parse_args()
    co = code_object(a.object)
    with tempfile.NamedTemporaryFile(suffix=".co") as f:
        f.write(co)
        f.flush()
        dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--mcpu=gfx950", f.name], stdout=subprocess.PIPE, check=True).stdout.decode()
    cyc = ffbench_cycles(a.ffbench)
    kernels = {}
    cur = None
    for line in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.*)>:$", line)
        if m:
            cur = m.group(1)
            continue
        if cur is None or a.kernel not in cur:
            continue
        m = re.match(r"^\s+(\w+)", line)
        if m:
            kernels.setdefault(cur, collections.Counter())[m.group(1)] += 1
    out = {"object": os.path.relpath(a.object, ROOT), "ffbench": os.path.relpath(a.ffbench, ROOT), "kernels": {}}
    for name, ops in kernels.items():
        valu = {k: v for k, v in ops.items() if k.startswith("v_") and not k.startswith("v_readlane") and not k.startswith("v_readfirstlane")}
        classes = collections.OrderedDict((c[0], 0) for c in CLASSES)
        for op, cnt in valu.items():
            for cname, rx, _ in CLASSES:
                if rx.match(op):
                    classes[cname] += cnt
                    break
        total = sum(classes.values())
        weighted = sum(classes[c[0]] * cyc[c[2]] for c in CLASSES)
        out["kernels"][name] = {
            "valu_instructions_static": total,
            "classes": classes,
            "class_issue_cycles": {c[0]: cyc[c[2]] for c in CLASSES},
            "mean_issue_cycles_per_valu_instruction": weighted / total if total else None,
            "top_opcodes": dict(collections.Counter(valu).most_common(12)),
            "other": {"salu": sum(v for k, v in ops.items() if k.startswith("s_")),
                      "vmem": sum(v for k, v in ops.items() if k.startswith(("global_", "buffer_", "flat_", "scratch_"))),
                      "lds": sum(v for k, v in ops.items() if k.startswith("ds_"))},
        }
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
