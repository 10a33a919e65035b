#!/bin/bash
# CPU check of the kernel experiments kept as patches (profiles/r03_*.patch): each is applied to a scratch copy of the tree,
# the emulated parity tests run on its source (tests/test_emulated.py, a selection) and its gfx950 assembly runs on the
# interpreter against the oracle (tests/emu/isa/check_kernels.py, then 40 random configurations of fuzz_kernels.py). No GPU.
# About half an hour.
#   bash tools/check_archived_variants.sh > profiles/r03_archived_variants_check.txt
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
check() {   # patch, options
    local patch=$1; shift
    local tmp; tmp=$(mktemp -d /tmp/rt_check_XXXX)
    git -C "$ROOT" worktree add --detach "$tmp/tree" HEAD >/dev/null 2>&1
    ( cd "$tmp/tree" && git apply "$ROOT/profiles/$patch" && make -s -C oracle librt_oracle.so >/dev/null 2>&1 &&
      cp "$ROOT/raytracing_amd/librt_mi355x.so" raytracing_amd/ && make -s -C tests/emu EXTRA="$*" >/dev/null 2>&1 &&
      echo "== $patch $* : emulated parity tests (source)" &&
      python3 -m pytest tests/test_emulated.py -q -x -p no:cacheprovider -k "radiance or llvmpipe_fixtures or primary_ids or deepest or grid_size or tile_order or several_rays" 2>&1 | tail -n 1 &&
      echo "== $patch $* : the compiled kernels on the interpreter" &&
      python3 tests/emu/isa/check_kernels.py "$@" 2>&1 | grep -v -i warning | tail -n 4 &&
      echo "== $patch $* : random configurations on the interpreter" &&
      python3 tests/emu/isa/fuzz_kernels.py --cases 40 --seed 21 "$@" 2>&1 | tail -n 1 )
    git -C "$ROOT" worktree remove --force "$tmp/tree"; rm -rf "$tmp"
}
check r03_gather_memory_lanes.patch -DRT_GLOBAL_AT=16
check r03_texel_pairs.patch
check r03_two_rays_per_lane.patch
git -C "$ROOT" worktree prune
