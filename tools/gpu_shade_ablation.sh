#!/bin/bash
# What the parts of k_shade cost, by taking them out (timing builds; the IMAGE CHANGES in every variant but `base`):
#   hash   the stratified sampler's SipHash-1-3 replaced by one multiply       (-DYK_ABLATE_HASH)
#   libm   sin / cos / tan / log: hardware approximations instead of the f64 recipe (-DYK_ABLATE_LIBM)
#   fast   the whole library with -ffast-math -ffp-contract=fast: approximate division / sqrt, fused multiply-adds, reassociation
# Build first:  tools/build_variant.sh abl_hash -DYK_ABLATE_HASH ; abl_libm -DYK_ABLATE_LIBM ; abl_fast "-ffast-math -ffp-contract=fast"
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_shade; mkdir -p $O; cd $R
for v in "" _abl_hash _abl_libm _abl_fast; do
  YK_LIB_PATH=$R/yuki_amd/libyuki_hip$v.so YK_DEBUG_BOUNCES=1 python3 tools/quick_bench.py cfg3 64 1920 1080 134217728 > $O/shade${v:-_base}.txt 2>&1
  echo "== ${v:-_base}"; grep -E "^bounce [0-3]|^wall" $O/shade${v:-_base}.txt | tail -5
done
