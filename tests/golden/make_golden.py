#!/usr/bin/env python3
"""Generate tests/golden/* from the reference's own test data (run in the build container only).

Reads ONLY data files / constant tables the reference's tests hold:
  kzg-bench/src/trusted_setup.txt                     -> trusted_setup.txt (verbatim data file)
  kzg-bench/src/test_vectors/<fn>/kzg-mainnet/*/data.yaml
                                                      -> kzg_mainnet.json + blobs/<sha8>.bin.gz
  kzg-bench/src/tests/{eip_4844,fft_fr,das}.rs        -> kats.json (hard-coded known-answer constants)
  blst/src/consts.rs SCALE2_ROOT_OF_UNITY             -> kats.json["scale2_root_of_unity"]
Nothing here travels to the GPU box except the generated data.
"""
import gzip
import hashlib
import json
import os
import re
import shutil
import sys

import yaml

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
TV = os.path.join(REF, "kzg-bench/src/test_vectors")


def unhex(s):
    if s is None:
        return None
    assert s.startswith("0x")
    return bytes.fromhex(s[2:])


blobs = {}


def blob_ref(b):
    """store a blob once, keyed by sha256 prefix; non-131072-byte blobs (invalid length) are stored too"""
    h = hashlib.sha256(b).hexdigest()[:12]
    if h not in blobs:
        blobs[h] = b
    return h


def cases(fn):
    d = os.path.join(TV, fn, "kzg-mainnet")
    for name in sorted(os.listdir(d)):
        with open(os.path.join(d, name, "data.yaml")) as f:
            yield name, yaml.safe_load(f)


def main():
    os.makedirs(os.path.join(OUT, "blobs"), exist_ok=True)
    shutil.copyfile(os.path.join(REF, "kzg-bench/src/trusted_setup.txt"), os.path.join(OUT, "trusted_setup.txt"))
    out = {"source": "c-kzg-4844 kzg-mainnet vectors shipped in grandinetech/rust-kzg kzg-bench/src/test_vectors"}

    v = []
    for name, y in cases("blob_to_kzg_commitment"):
        v.append({"name": name, "blob": blob_ref(unhex(y["input"]["blob"])), "output": y["output"]})
    out["blob_to_kzg_commitment"] = v

    v = []
    for name, y in cases("compute_kzg_proof"):
        o = y["output"]
        v.append({"name": name, "blob": blob_ref(unhex(y["input"]["blob"])), "z": y["input"]["z"],
                  "output": None if o is None else [o[0], o[1]]})
    out["compute_kzg_proof"] = v

    v = []
    for name, y in cases("compute_blob_kzg_proof"):
        v.append({"name": name, "blob": blob_ref(unhex(y["input"]["blob"])), "commitment": y["input"]["commitment"],
                  "output": y["output"]})
    out["compute_blob_kzg_proof"] = v

    v = []
    for name, y in cases("compute_challenge"):
        v.append({"name": name, "blob": blob_ref(unhex(y["input"]["blob"])), "commitment": y["input"]["commitment"],
                  "output": y["output"]})
    out["compute_challenge"] = v

    # compute_cells pins the NTT (ifft 4096 + fft 8192).  Expected output = 128 cells x 2048 B;
    # keep its sha256 plus the first and last cell so a mismatch can be localised.
    v = []
    for name, y in cases("compute_cells"):
        o = y["output"]
        if o is None:
            v.append({"name": name, "blob": blob_ref(unhex(y["input"]["blob"])), "output": None})
            continue
        cells = b"".join(unhex(c) for c in o)
        assert len(cells) == 128 * 2048
        v.append({"name": name, "blob": blob_ref(unhex(y["input"]["blob"])),
                  "output": {"sha256": hashlib.sha256(cells).hexdigest(), "cell0": o[0], "cell127": o[127]}})
    out["compute_cells"] = v

    # compute_cells_and_kzg_proofs: 128 cells + 128 proofs per blob; digests + end points
    v = []
    for name, y in cases("compute_cells_and_kzg_proofs"):
        o = y["output"]
        if o is None:
            v.append({"name": name, "blob": blob_ref(unhex(y["input"]["blob"])), "output": None})
            continue
        cells = b"".join(unhex(c) for c in o[0])
        proofs = b"".join(unhex(c) for c in o[1])
        assert len(cells) == 128 * 2048 and len(proofs) == 128 * 48
        v.append({"name": name, "blob": blob_ref(unhex(y["input"]["blob"])),
                  "output": {"cells_sha256": hashlib.sha256(cells).hexdigest(),
                             "proofs_sha256": hashlib.sha256(proofs).hexdigest(),
                             "proof0": o[1][0], "proof1": o[1][1], "proof127": o[1][127]}})
    out["compute_cells_and_kzg_proofs"] = v

    # verification vectors (pin the pairing side and the batched-verification G1 work): expected = true / false / None
    v = []
    for name, y in cases("verify_kzg_proof"):
        i = y["input"]
        v.append({"name": name, "commitment": i["commitment"], "z": i["z"], "y": i["y"], "proof": i["proof"], "output": y["output"]})
    out["verify_kzg_proof"] = v
    v = []
    for name, y in cases("verify_blob_kzg_proof"):
        i = y["input"]
        v.append({"name": name, "blob": blob_ref(unhex(i["blob"])), "commitment": i["commitment"], "proof": i["proof"],
                  "output": y["output"]})
    out["verify_blob_kzg_proof"] = v
    v = []
    for name, y in cases("verify_blob_kzg_proof_batch"):
        i = y["input"]
        v.append({"name": name, "blobs": [blob_ref(unhex(b)) for b in i["blobs"]], "commitments": i["commitments"],
                  "proofs": i["proofs"], "output": y["output"]})
    out["verify_blob_kzg_proof_batch"] = v

    # trusted-setup text fixtures of the binding test-suite (kzg-bench/src/tests/c_bindings.rs:344-489): data files,
    # stored gzipped with the outcome load_trusted_setup_file must give
    fx = os.path.join(REF, "kzg-bench/src/tests/fixtures")
    os.makedirs(os.path.join(OUT, "setup_fixtures"), exist_ok=True)
    fixtures = {}
    for name in sorted(os.listdir(fx)):
        data = open(os.path.join(fx, name, "trusted_setup_fixture.txt"), "rb").read()
        with gzip.GzipFile(os.path.join(OUT, "setup_fixtures", name + ".txt.gz"), "wb", mtime=0) as f:
            f.write(data)
        fixtures[name] = {"expect": "ok" if name.startswith("valid_") else "badargs", "bytes": len(data),
                          "sha256": hashlib.sha256(data).hexdigest()}
    out["setup_fixtures"] = {"cite": "kzg-bench/src/tests/c_bindings.rs:344-489", "files": fixtures}

    # ---- EIP-7594 cell functions (c_bindings.rs:202-355, blst/src/eip_7594.rs:35-97): cells are 2048-byte strings, most
    # of them shared between cases; stored once each in cells_7594.bin.gz, the cases refer to them by position
    cell_pos = {}
    cell_list = []

    def cell_ref(hexstr):
        b = unhex(hexstr)
        if len(b) != 2048:
            return {"hex": hexstr}  # wrong-length input: kept inline
        k = hashlib.sha256(b).digest()
        if k not in cell_pos:
            cell_pos[k] = len(cell_list)
            cell_list.append(b)
        return cell_pos[k]

    out7 = {"source": out["source"], "cells_file": "cells_7594.bin.gz"}
    v = []
    for name, y in cases("verify_cell_kzg_proof_batch"):
        i = y["input"]
        v.append({"name": name, "commitments": i["commitments"], "cell_indices": i["cell_indices"],
                  "cells": [cell_ref(c) for c in i["cells"]], "proofs": i["proofs"], "output": y["output"]})
    out7["verify_cell_kzg_proof_batch"] = v
    v = []
    for name, y in cases("recover_cells_and_kzg_proofs"):
        i, o = y["input"], y["output"]
        e = {"name": name, "cell_indices": i["cell_indices"], "cells": [cell_ref(c) for c in i["cells"]], "output": None}
        if o is not None:
            cells = b"".join(unhex(c) for c in o[0])
            proofs = b"".join(unhex(c) for c in o[1])
            assert len(cells) == 128 * 2048 and len(proofs) == 128 * 48
            e["output"] = {"cells_sha256": hashlib.sha256(cells).hexdigest(), "proofs_sha256": hashlib.sha256(proofs).hexdigest(),
                           "cells": [cell_ref(c) for c in o[0]], "proof0": o[1][0], "proof1": o[1][1], "proof127": o[1][127]}
        v.append(e)
    out7["recover_cells_and_kzg_proofs"] = v
    v = []
    for name, y in cases("compute_verify_cell_kzg_proof_batch_challenge"):
        i = y["input"]
        v.append({"name": name, "commitments": i["commitments"], "commitment_indices": i["commitment_indices"],
                  "cell_indices": i["cell_indices"], "cells": [cell_ref("0x" + "".join(fe[2:] for fe in c)) for c in i["cosets_evals"]], "proofs": i["proofs"],
                  "output": y["output"]})
    out7["compute_verify_cell_kzg_proof_batch_challenge"] = v
    with gzip.GzipFile(os.path.join(OUT, "cells_7594.bin.gz"), "wb", mtime=0) as f:
        f.write(b"".join(cell_list))
    with open(os.path.join(OUT, "kzg_mainnet_7594.json"), "w") as f:
        json.dump(out7, f, indent=1)
    print("eip-7594 cells:", len(cell_list))

    for h, b in blobs.items():
        with gzip.GzipFile(os.path.join(OUT, "blobs", h + ".bin.gz"), "wb", mtime=0) as f:
            f.write(b)
    with open(os.path.join(OUT, "kzg_mainnet.json"), "w") as f:
        json.dump(out, f, indent=1)

    # ---- known-answer constants hard-coded in the reference's tests ----
    kats = {}
    src = open(os.path.join(REF, "kzg-bench/src/tests/eip_4844.rs")).read()
    m = re.search(r"const EXPECTED_POWERS.*?= \[(.*?)\];", src, re.S)
    kats["expected_powers"] = {"base": 32930439,
                               "cite": "kzg-bench/src/tests/eip_4844.rs:47-60",
                               "limbs": [[int(x) for x in re.findall(r"\d+", row)]
                                         for row in re.findall(r"\[([^\[\]]*)\]", m.group(1))]}
    fe = re.findall(r'TFr::from_hex\("(0x[0-9a-f]{64})"\)', src)
    g1 = [a + b for a, b in re.findall(r'TG1::from_hex\(\s*"(0x[0-9a-f]+)\\\s*([0-9a-f]+)"', src)]
    kats["blob_to_kzg_commitment_test"] = {"cite": "kzg-bench/src/tests/eip_4844.rs:85-121",
                                           "field_element": fe[0], "commitment": g1[0]}
    kats["compute_kzg_proof_test"] = {"cite": "kzg-bench/src/tests/eip_4844.rs:124-175",
                                      "field_element": fe[1], "z": fe[2], "proof": g1[1]}
    assert kats["blob_to_kzg_commitment_test"]["commitment"].startswith("0x91a5e1c1")
    assert kats["compute_kzg_proof_test"]["proof"].startswith("0xb21f8f9b")

    def limb_table(path, name):
        s = open(os.path.join(REF, path)).read()
        m = re.search(name + r".*?=\s*\[(.*?)\];", s, re.S)
        return [[int(x, 16) for x in re.findall(r"0x[0-9a-fA-F]+", row)]
                for row in re.findall(r"\[([^\[\]]*)\]", m.group(1))]

    kats["inverse_fft"] = {"cite": "kzg-bench/src/tests/fft_fr.rs:49-84", "scale": 4,
                           "expected": limb_table("kzg-bench/src/tests/fft_fr.rs", "inv_fft_expected")}
    kats["das_extension_known"] = {"cite": "kzg-bench/src/tests/das.rs:4-31", "scale": 4,
                                   "expected": limb_table("kzg-bench/src/tests/das.rs", "expected_u")}
    kats["scale2_root_of_unity"] = {"cite": "blst/src/consts.rs:17-50",
                                    "limbs": limb_table("blst/src/consts.rs", "SCALE2_ROOT_OF_UNITY")}
    assert len(kats["scale2_root_of_unity"]["limbs"]) == 32
    assert len(kats["inverse_fft"]["expected"]) == 16 and len(kats["das_extension_known"]["expected"]) == 8
    s = open(os.path.join(REF, "blst/src/consts.rs")).read()
    gen = re.search(r"G1_GENERATOR.*?\);", s, re.S).group(0)
    kats["g1_generator_mont_limbs"] = {"cite": "blst/src/consts.rs:52-84",
                                       "xyz": [int(x, 16) for x in re.findall(r"0x[0-9a-fA-F]+", gen)]}
    with open(os.path.join(OUT, "kats.json"), "w") as f:
        json.dump(kats, f, indent=1)
    tot = sum(os.path.getsize(os.path.join(OUT, "blobs", x)) for x in os.listdir(os.path.join(OUT, "blobs")))
    print("blobs:", len(blobs), "bytes gz:", tot)


if __name__ == "__main__":
    sys.exit(main())
