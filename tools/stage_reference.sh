#!/bin/bash
# Stage the reference's own Python sources for the GPU box: /root/reference exists only in the build container, and gpurun ships
# git-ignored files that .gpurunignore does not list -- oracle/_ref/ is such a path (.gitignore: "oracle/_ref/"), so what is copied here
# travels with the snapshot and never enters the history.  Test infrastructure only (the same rule as for everything under oracle/):
# tests/test_gpu_reference_runner_live.py drives the reference's STEPRunner around the HIP module in ONE process, bench.py's cpu_baseline
# times the reference's modules on the GPU box's host cores (kind "reference"); nothing under step_amd/ or include/ may import it
# (tests/test_abi_and_host.py::test_product_never_imports_the_oracle_or_the_reference).
#   usage: tools/stage_reference.sh [reference root, default /root/reference]
set -e
src=${1:-/root/reference}
dst="$(cd "$(dirname "$0")/.." && pwd)/oracle/_ref/reference"
if [ ! -d "$src/step/step_arch" ]; then echo "no reference checkout at $src"; exit 1; fi
[ -d "$dst" ] && chmod -R u+w "$dst"; rm -rf "$dst"; mkdir -p "$dst"
cp -r "$src/step" "$src/basicts" "$dst/"
chmod -R u+w "$dst"
find "$dst" -name "__pycache__" -type d -prune -exec rm -rf {} +
find "$dst" -name "*.log" -delete
[ -f "$src/LICENSE" ] && cp "$src/LICENSE" "$dst/" || true
echo "staged $(find "$dst" -name '*.py' | wc -l) python files of $src under $dst ($(du -sh "$dst" | cut -f1))"
