#!/usr/bin/env bash
# Real blst numbers for the CPU baseline, when a Rust toolchain and a checkout of the reference are at hand
# (BASELINE.md §2.2).  The build image and the GPU boxes have neither, so bench.py's
# `cpu_baseline` times the portable-C oracle instead (kind "port"); this script is what a maintainer runs on a box
# that has cargo to put the reference's own figure next to it.
#
#   bench/blst_baseline.sh /path/to/rust-kzg [out.json]
#
# Runs the reference's own criterion bench for the same workload as bench.py's headline
# (blob_to_kzg_commitment on one 4096-element blob, mainnet setup; kzg-bench/src/benches/eip_4844.rs) with the
# reference's default fixed-base algorithm (feature bgmw) and its thread pool (feature parallel), and prints
# one JSON object: {"commitments_per_s": ..., "ms_per_commitment": ..., "cores": ..., "kind": "reference"}.
set -euo pipefail
REF=${1:-}
OUT=${2:-/dev/stdout}
if [ -z "$REF" ] || [ ! -f "$REF/blst/Cargo.toml" ]; then
  echo "usage: $0 /path/to/rust-kzg [out.json]   (a checkout of grandinetech/rust-kzg)" >&2
  exit 2
fi
if ! command -v cargo >/dev/null 2>&1; then
  echo "cargo not found: no Rust toolchain on this box; bench.py reports the portable-C oracle instead" >&2
  exit 3
fi
CORES=$(nproc)
LOG=$(mktemp)
( cd "$REF" && cargo bench -p rust-kzg-blst --bench eip_4844 --features parallel,bgmw -- blob_to_kzg_commitment ) 2>&1 | tee "$LOG" >&2
# criterion prints e.g. "bench_blob_to_kzg_commitment  time:   [37.9 ms 38.1 ms 38.4 ms]": take the median
python3 - "$LOG" "$CORES" > "$OUT" <<'PY'
import json, re, sys
txt = open(sys.argv[1]).read()
m = re.search(r"blob_to_kzg_commitment[^\[]*\[\s*[\d.]+\s*\w+\s+([\d.]+)\s*(\w+)", txt)
if not m:
    sys.exit("could not find the criterion line for blob_to_kzg_commitment")
val, unit = float(m.group(1)), m.group(2)
ms = val * {"ns": 1e-6, "us": 1e-3, "µs": 1e-3, "ms": 1.0, "s": 1e3}[unit]
print(json.dumps({"commitments_per_s": 1e3 / ms, "ms_per_commitment": ms, "cores": int(sys.argv[2]), "kind": "reference",
                  "sample": "cargo bench -p rust-kzg-blst --bench eip_4844 --features parallel,bgmw (criterion median)"}))
PY
