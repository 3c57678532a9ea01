"""Builds libkzg_mi355x.so (HIP kernels + C-ABI) for gfx950 with hipcc, in-tree."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libkzg_mi355x.so")
SOURCES = ["msm.hip", "ckzg.hip", "ckzg_verify.hip", "ckzg_7594.hip", "ntt.hip", "fftg1.hip", "multi.hip"]
HEADERS = ["ff.hip.h", "fp28.hip.h", "g1_28.hip.h", "g1_io.hip.h", "msm_internal.h", "ckzg_internal.h", "sha256.h", "host_g1.h", "host_pairing.h", "host_fp64.h", "ff28.hip.h", "fr29.hip.h", "ntt_internal.h", "device_guard.h", "fpw.hip.h", "g1w.hip.h", "glv.hip.h", "ntt_plan.h", "ckzg_shared.h", "config.h",
           os.path.join("..", "..", "include", "kzg_mi355x.h")]


# what the last build() / build_exact() / build_prefixed() of this process did, per library file name: "reused (content stamp
# matches the sources)" or "compiled: <sources>; linked" — __graft_entry__.build() prints it, so that a build record says
# whether hipcc ran
REPORT = {}


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    for f in SOURCES + HEADERS:
        p = os.path.join(CSRC, f)
        if os.path.exists(p) and os.path.getmtime(p) > t:
            return True
    return False


# c-kzg-4844 names this library shares with the reference's own C bindings (blst/src/eip_4844.rs, kzg/src/eth/c_bindings.rs).
# A process that also links the Rust staticlib (e.g. a test binary comparing the two backends) cannot have both sets
# under the same names: build_prefixed() links a second flavour,
# libkzg_mi355x_prefixed.so, in which every one of them is exported as kzgamd_ckzg_<name> instead
# (include/kzg_mi355x.h maps the plain names when KZG_MI355X_PREFIXED is defined).
CKZG_NAMES = ["load_trusted_setup", "load_trusted_setup_file", "free_trusted_setup", "blob_to_kzg_commitment", "compute_kzg_proof",
              "compute_blob_kzg_proof", "verify_kzg_proof", "verify_blob_kzg_proof", "verify_blob_kzg_proof_batch",
              "compute_challenge", "bytes_to_kzg_commitment", "bytes_from_bls_field", "compute_cells_and_kzg_proofs",
              "recover_cells_and_kzg_proofs", "verify_cell_kzg_proof_batch", "compute_verify_cell_kzg_proof_batch_challenge"]
LIB_PREFIXED = os.path.join(CSRC, "libkzg_mi355x_prefixed.so")


def build_prefixed(verbose=False):
    """libkzg_mi355x_prefixed.so from the objects of build(): c-kzg names renamed with llvm-objcopy, nothing recompiled."""
    build()
    objcopy = os.environ.get("OBJCOPY", "/opt/rocm/lib/llvm/bin/llvm-objcopy")
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    for s in SOURCES:
        o = os.path.join(CSRC, s.replace(".hip", ".o"))
        po = os.path.join(CSRC, s.replace(".hip", ".prefixed.o"))
        cmd = [objcopy] + ["--redefine-sym=%s=kzgamd_ckzg_%s" % (n, n) for n in CKZG_NAMES] + [o, po]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        objs.append(po)
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PREFIXED] + objs + ["-lpthread"], cwd=CSRC)
    for o in objs:
        os.remove(o)
    REPORT[os.path.basename(LIB_PREFIXED)] = "relinked from the product library's objects (symbols renamed with llvm-objcopy)"
    return LIB_PREFIXED


def _obj_stale(src, obj):
    """An object is rebuilt when its source, or any header the compiler listed in its .d file, is newer."""
    if not os.path.exists(obj):
        return True
    t = os.path.getmtime(obj)
    if os.path.getmtime(src) > t:
        return True
    dep = obj[:-2] + ".d"
    if not os.path.exists(dep):
        return True
    with open(dep) as f:
        names = f.read().replace("\\\n", " ").split()
    for name in names[1:]:
        if name.startswith("/opt/") or name.startswith("/usr/"):
            continue
        path = name if os.path.isabs(name) else os.path.join(CSRC, name)
        if not os.path.exists(path) or os.path.getmtime(path) > t:
            return True
    return False


def _content_hash(defines):
    """sha256 over the text of every source and header of the library and the compile flags: what the objects are a
    function of (with the pinned toolchain).  Modification times lie — an edit while a compilation runs leaves an object
    newer than its source — so a library is only reused when the stamp written next to it names this hash."""
    import hashlib

    h = hashlib.sha256(" ".join(defines).encode())
    names = sorted(set(SOURCES) | {f for f in os.listdir(CSRC) if f.endswith(".h")})
    for name in names + [os.path.join("..", "..", "include", "kzg_mi355x.h")]:
        p = os.path.join(CSRC, name)
        if os.path.exists(p):
            with open(p, "rb") as fh:
                h.update(name.encode() + b"\0" + fh.read() + b"\0")
    return h.hexdigest()


def _stamp_path(lib):
    return lib + ".built_from"


def _stamp_matches(lib, defines):
    try:
        with open(_stamp_path(lib)) as fh:
            return fh.read().strip() == _content_hash(defines)
    except OSError:
        return False


def _compile_and_link(lib, tag, defines, force, verbose):
    """Objects <source><tag>.o (parallel hipcc, rebuilt by their .d files) -> lib; then the content stamp."""
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    want = _content_hash(defines)  # before compiling: an edit during the compilation makes the stamp stale, as it should
    objs = []
    procs = []
    for s in srcs:
        o = os.path.join(CSRC, s.replace(".hip", tag + ".o"))
        objs.append(o)
        if not force and not _obj_stale(os.path.join(CSRC, s), o):
            continue
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"] + defines + ["-MD", "-MF", o[:-2] + ".d", "-c", os.path.join(CSRC, s), "-o", o]
        if verbose:
            print(" ".join(cmd))
        procs.append((s, subprocess.Popen(cmd, cwd=CSRC)))
    for s, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed on " + s)
    compiled = [s for s, _ in procs]
    linked = False
    if procs or not os.path.exists(lib) or any(os.path.getmtime(o) > os.path.getmtime(lib) for o in objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs + ["-lpthread"]
        subprocess.check_call(cmd, cwd=CSRC)
        linked = True
    REPORT[os.path.basename(lib)] = ("compiled with hipcc --offload-arch=gfx950: %s; " % (", ".join(compiled) if compiled else "nothing (objects up to date)")) + \
        ("linked" if linked else "library up to date")
    with open(_stamp_path(lib), "w") as fh:
        fh.write(want + "\n")
    return lib


def build(force=False, verbose=False):
    """The product library.  Reused only when its stamp names the content hash of the sources; objects are reused by
    their dependency files, except that a library whose stamp names ANOTHER hash while no object looks stale (an edit
    during a compilation) is recompiled in full.  KZGAMD_REBUILD=1 forces a full recompilation (what a fresh clone does)."""
    force = force or os.environ.get("KZGAMD_REBUILD") == "1"
    if not force and os.path.exists(LIB) and _stamp_matches(LIB, []):
        REPORT[os.path.basename(LIB)] = "reused (content stamp matches the sources: nothing to compile)"
        return LIB
    if not force and os.path.exists(LIB) and not _stale():
        force = True  # nothing looks newer than the library, yet it was built from other text
    return _compile_and_link(LIB, "", [], force, verbose)


# The forced-rare-path flavour: the same sources with -DKZGAMD_FORCE_EXACT_TESTS, in which the cheap filter in front of
# every exact "is this zero mod p" test (fp28::is_zero_mod_p, g1w::is_zero_mod_p: taken by 2.4e-7 of the values) is
# compiled out, so the exact comparison — the code a rare value reaches, with its LDS exchanges and wave-local
# synchronisation — runs on EVERY point addition of every kernel.  Results must be bit-identical to the product
# library's; the GPU suite and the fuzzers run against both (tests/conftest.py).  Test infrastructure, never shipped.
LIB_EXACT = os.path.join(CSRC, "libkzg_mi355x_exact.so")


def build_exact(force=False, verbose=False):
    force = force or os.environ.get("KZGAMD_REBUILD") == "1"
    defines = ["-DKZGAMD_FORCE_EXACT_TESTS"]
    if not force and os.path.exists(LIB_EXACT) and _stamp_matches(LIB_EXACT, defines):
        REPORT[os.path.basename(LIB_EXACT)] = "reused (content stamp matches the sources: nothing to compile)"
        return LIB_EXACT
    return _compile_and_link(LIB_EXACT, ".exact", defines, force, verbose)


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    if "--prefixed" in sys.argv:
        print(build_prefixed(verbose=True))
    if "--exact" in sys.argv:
        print(build_exact(force="--force" in sys.argv, verbose=True))
