#!/usr/bin/env bash
# The reference's own generic test-suite and criterion benches, instantiated for the MI355X backend
# (rust-kzg_amd/rust-backend/{tests,benches}: the counterpart of blst/tests and blst/benches), on a box that has a
# Rust toolchain, a checkout of the reference and an MI355X.  The build image has no cargo: there the same vectors
# run through tests/ (ctypes) and tests/c_abi_harness.c (C) instead.  SURVEY.md §4, /root/reference/run-c-kzg-4844-tests.sh.
#
#   bench/run_reference_suite.sh /path/to/rust-kzg [test|bench|all]
#
# What it does: copies the two crates next to blst/ in the checkout (rust-kzg-mi355x-sys -> <ref>/rust-kzg_amd/rust,
# rust-kzg-mi355x -> <ref>/mi355x), adds them to the workspace, builds libkzg_mi355x.so if it is missing, and runs
#   cargo test  -p rust-kzg-mi355x            (bls12_381, fft, eip_4844, eip_7594, kzg_proofs, c_bindings)
#   cargo bench -p rust-kzg-mi355x            (eip_4844, lincomb_fft)   next to   cargo bench -p rust-kzg-blst --bench eip_4844
set -euo pipefail
REF=${1:-}
WHAT=${2:-all}
HERE=$(cd "$(dirname "$0")/.." && pwd)
if [ -z "$REF" ] || [ ! -f "$REF/blst/Cargo.toml" ]; then
  echo "usage: $0 /path/to/rust-kzg [test|bench|all]   (a checkout of grandinetech/rust-kzg)" >&2
  exit 2
fi
command -v cargo >/dev/null 2>&1 || { echo "cargo not found: no Rust toolchain on this box" >&2; exit 3; }
[ -f "$HERE/rust-kzg_amd/csrc/libkzg_mi355x.so" ] || python3 "$HERE/rust-kzg_amd/build.py"
export KZG_MI355X_LIB_DIR="$HERE/rust-kzg_amd/csrc"
export LD_LIBRARY_PATH="$KZG_MI355X_LIB_DIR:/opt/rocm/lib:${LD_LIBRARY_PATH:-}"
mkdir -p "$REF/rust-kzg_amd" && rm -rf "$REF/rust-kzg_amd/rust" "$REF/mi355x"
cp -r "$HERE/rust-kzg_amd/rust" "$REF/rust-kzg_amd/rust"
cp -r "$HERE/rust-kzg_amd/rust-backend" "$REF/mi355x"
grep -q '"mi355x"' "$REF/Cargo.toml" || sed -i 's/members = \[/members = [\n    "mi355x",\n    "rust-kzg_amd\/rust",/' "$REF/Cargo.toml"
cd "$REF"
if [ "$WHAT" = test ] || [ "$WHAT" = all ]; then
  # one test binary at a time: every settings object builds HBM-sized tables
  KZGAMD_FBW_MAX_GB=${KZGAMD_FBW_MAX_GB:-40} cargo test -p rust-kzg-mi355x --release -- --test-threads 1
fi
if [ "$WHAT" = bench ] || [ "$WHAT" = all ]; then
  cargo bench -p rust-kzg-mi355x --bench eip_4844
  cargo bench -p rust-kzg-mi355x --bench lincomb_fft
  cargo bench -p rust-kzg-blst --bench eip_4844 --features parallel,bgmw   # the CPU figure beside it (bench/blst_baseline.sh)
fi
