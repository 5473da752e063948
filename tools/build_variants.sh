#!/bin/bash
# A-B builds of the library: tools/build_variants.sh name1 "flags1" name2 "flags2" ...  ->  das3r_amd/libdas3r_hip.<name>.so
# (they travel to the GPU box; there:  cp das3r_amd/libdas3r_hip.<name>.so das3r_amd/libdas3r_hip.so  in front of each run).
# The default library is rebuilt at the end.
set -e
cd "$(dirname "$0")/../das3r_amd/csrc"
SRC="api.hip render_rows.hip render_fwd.hip render_bwd_blk.hip render_bwd.hip preprocess.hip preprocess_bwd.hip sort_onesweep.hip scan_emit.hip segsort.hip render_lanes.hip"
while [ $# -ge 2 ]; do
  touch $SRC
  make -s -j8 XFLAGS="$2" 2>&1 | grep -i " error" -A6 || true
  cp ../libdas3r_hip.so ../libdas3r_hip.$1.so
  shift 2
done
touch $SRC
make -s -j8 2>&1 | grep -i " error" -A6 || true
ls -la ../*.so
