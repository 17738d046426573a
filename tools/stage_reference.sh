#!/bin/bash
# Pack the reference's own Python sources for the GPU box: /root/reference exists only in the build container, and gpurun ships
# git-ignored files that .gpurunignore does not list -- oracle/_ref/ is such a path (.gitignore: "oracle/_ref/").  ONE binary artefact,
# oracle/_ref/reference.tar.gz, travels with the snapshot and never enters the history; no loose copy of a reference file is left in the
# tree (oracle/reference_loader.py unpacks it into the temporary directory of whatever box uses it).  Test infrastructure only (the same
# rule as for everything under oracle/): tests/test_gpu_reference_runner_live.py drives the reference's STEPRunner around the HIP module in
# ONE process, bench.py's cpu_baseline times the reference's modules on the GPU box's host cores (kind "reference"); nothing under
# step_amd/ or include/ may import it (tests/test_abi_and_host.py::test_product_never_imports_the_oracle_or_the_reference).
#   usage: tools/stage_reference.sh [reference root, default /root/reference]
set -e
src=${1:-/root/reference}
ref="$(cd "$(dirname "$0")/.." && pwd)/oracle/_ref"
if [ ! -d "$src/step/step_arch" ]; then echo "no reference checkout at $src"; exit 1; fi
mkdir -p "$ref"
[ -d "$ref/reference" ] && chmod -R u+w "$ref/reference" && rm -rf "$ref/reference"      # (the loose copy earlier revisions of this script made)
extra=""; [ -f "$src/LICENSE" ] && extra="LICENSE"
tar -C "$src" --exclude='__pycache__' --exclude='*.log' --exclude='*.pyc' -czf "$ref/reference.tar.gz.part" step basicts $extra
mv "$ref/reference.tar.gz.part" "$ref/reference.tar.gz"
echo "packed $(tar -tzf "$ref/reference.tar.gz" | grep -c '\.py$') python files of $src into $ref/reference.tar.gz ($(du -sh "$ref/reference.tar.gz" | cut -f1))"
