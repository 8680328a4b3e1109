#!/bin/bash
# HBM traffic of the hot kernels on one shape: FETCH_SIZE and WRITE_SIZE in separate counter passes over tools/profile_shape.py.
# usage (GPU box, repo root): bash tools/pmc_shape.sh H W K [maps]   -> KB per dispatch (FETCH_SIZE x2 on gfx950 for wide reads, see MI355X_MICROARCH.md)
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-$OLDPWD}
TAG=$(echo "$@" | tr ' ' '_')
OUT=gpurun_out/pmc_shape_$TAG; mkdir -p $OUT
STEPS=2 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/f -o f -- python3 tools/profile_shape.py "$@" > $OUT/f.log 2>&1
STEPS=2 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/w -o w -- python3 tools/profile_shape.py "$@" > $OUT/w.log 2>&1
echo "== $@"; tail -1 $OUT/f.log | cut -c1-200
python3 tools/pmc_summary.py $(find $OUT/f -name "*counter_collection.csv") $(find $OUT/w -name "*counter_collection.csv") | grep -A1 "k_fast"
find $OUT -name "*.csv" -size +1M -delete; find $OUT -name "*.db" -delete
