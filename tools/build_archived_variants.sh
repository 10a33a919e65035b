#!/bin/bash
# Builds the kernel experiments that are kept as patches (profiles/r03_*.patch, see profiles/EXPERIMENTS.md) into
# build_variants/<name>.so, here on the CPU, so that one GPU call can check and time them:
#
#   bash tools/build_archived_variants.sh
#   (take ./build_variants out of .gpurunignore for that call: it keeps the libraries off the GPU box otherwise)
#   gpurun -- 'bash tools/variant_parity.sh gather16 && bash tools/variant_check.sh r04 gather8 gather16 gather24 gather32 pairs gp16 gp16t gp24t tworay'
#
# Each patch is applied to a scratch copy of the work tree (git worktree), never to the tree itself.
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/build_variants"
mkdir -p "$OUT"
build() {   # name, patches (joined by +), extra compiler options
    local name=$1 patches=$2; shift 2
    local tmp; tmp=$(mktemp -d /tmp/rt_variant_XXXX)
    git -C "$ROOT" worktree add --detach "$tmp/tree" HEAD >/dev/null 2>&1
    ( cd "$tmp/tree" && for p in ${patches//+/ }; do git apply "$ROOT/profiles/$p"; done && cd raytracing_amd/csrc &&
      make -s HIPCC="/opt/rocm/bin/hipcc $*" >/dev/null 2>&1 && cp ../librt_mi355x.so "$OUT/$name.so" )
    git -C "$ROOT" worktree remove --force "$tmp/tree"; rm -rf "$tmp"
    echo "built build_variants/$name.so"
}
for t in 8 16 24 32; do build gather$t r03_gather_memory_lanes.patch -DRT_GLOBAL_AT=$t; done
build pairs r03_texel_pairs.patch
# the two together, and with a larger LDS copy of the tree top paid for by two stack entries (same LDS per work-group)
build gp16 r03_gather_memory_lanes.patch+r03_texel_pairs.patch -DRT_GLOBAL_AT=16
build gp16t r03_gather_memory_lanes.patch+r03_texel_pairs.patch -DRT_GLOBAL_AT=16 -DRT_LDS_STACK=10 -DRT_TOP_LDS=168
build gp24t r03_gather_memory_lanes.patch+r03_texel_pairs.patch -DRT_GLOBAL_AT=24 -DRT_LDS_STACK=10 -DRT_TOP_LDS=168
build tworay r03_two_rays_per_lane.patch
build tworay3 r03_two_rays_per_lane.patch -DRT_MIN_WAVES=3
git -C "$ROOT" worktree prune
