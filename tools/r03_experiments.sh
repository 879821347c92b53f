#!/bin/bash
# The timing-only elimination builds (NCA_EXP=<bits>) and the measured-and-switched-off kernel variants of rounds 2 and 3
# (NCA_BF_PIPE / NCA_BF_PIPE2, NCA_WGRAD_TR, NCA_ONCHIP_NR, NCA_CHAIN8, NCA_WAVES=4 ..., and the bf16-STAGED store with kernel modes
# 3 / 4) are not in the product sources any more (round 4: the shipped translation units hold only the code that runs).  They live in
# the repository's history under the tag `r03-kernels`; this script checks that tree out next to the product one and builds any of
# them THERE, with that round's own scripts -- so every A/B number of profiles/r02_* / r03_* and DESIGN.md 4.4 / 7 stays reproducible:
#
#   tools/r03_experiments.sh                                   # materialise ./_r03 (git worktree of r03-kernels; git-ignored, but it
#                                                              #  travels to the GPU box with the snapshot) and build its product library
#   tools/r03_experiments.sh elim 16 32 128                    # ./_r03/tools/elim_build.sh 16 32 128   (timing-only builds)
#   tools/r03_experiments.sh variant p2 "-DNCA_BF_PIPE2=1"     # ./_r03/tools/variant_build.sh p2 "-DNCA_BF_PIPE2=1"
#   tools/r03_experiments.sh variant_all c8 "-DNCA_CHAIN8=1"   # ./_r03/tools/variant_build_all.sh ...
#   tools/r03_experiments.sh patch tools/r03_c8_l0bf.patch     # apply a patch to ./_r03 first (this one: round 4's "layer 0 on bf16" form of the
#                                                              #  fp8 chain, then: variant_all c8l0 "-DNCA_CHAIN8=1 -DNCA_C8_L0BF=1")
# then, ON the GPU box:   cd _r03 && bash tools/ab_bench.sh 20 default p2      (its bench.py, its library, its ABI 8)
set -e
cd "$(dirname "$0")/.."
if [ ! -d _r03 ]; then
  # (the tag names commit 86ee5bb, the last commit of round 3's kernels: used directly where a clone did not bring the tag along)
  REV=r03-kernels; git rev-parse -q --verify "$REV^{commit}" > /dev/null || REV=86ee5bb4897b4d4f37e23bd602f48dc8e3332020
  git worktree add --detach _r03 $REV > /dev/null
fi
( cd _r03 && make -j4 > /dev/null && echo "_r03: round-3 product library built ($(git -C . rev-parse --short HEAD))" )
case "$1" in
  elim) shift; ( cd _r03 && tools/elim_build.sh "$@" ) ;;
  variant) shift; ( cd _r03 && tools/variant_build.sh "$@" ) ;;
  variant_all) shift; ( cd _r03 && tools/variant_build_all.sh "$@" ) ;;
  patch) ( cd _r03 && git apply "../$2" && echo "applied $2" ) ;;
  "") ;;
  *) echo "usage: see the head of this script"; exit 1 ;;
esac
