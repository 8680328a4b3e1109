#!/bin/bash
# Profiles of the default bench (cfg3): kernel trace + stats, then FETCH_SIZE and WRITE_SIZE in
# separate counter passes, and profiles/traffic.json regenerated from them (stamped with the commit).
# Usage (on the GPU box, from the repo root): bash tools/profile_round.sh r02k <commit>
# then copy gpurun_out/prof_<tag>/{bench*.json,kernel_stats.csv,pmc_fetch_write.txt,traffic.json} into profiles/
set -e
TAG=${1:-r02x}
COMMIT=${2:-unknown}
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-$OLDPWD}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $OUT/bench_under_rocprof.json 2> $OUT/trace.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o fetch -- python3 bench.py --steps 1 --warmup 1 --no-clock-warm --no-cpu-baseline --no-extras > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o write -- python3 bench.py --steps 1 --warmup 1 --no-clock-warm --no-cpu-baseline --no-extras > /dev/null 2> $OUT/pmc_write.err
find $OUT -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
( echo "# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), bench.py --steps 1 --warmup 1 (cfg3 defaults)";
  echo "# units: KB per dispatch (mean over dispatches of that grid size); gfx950 correction: FETCH_SIZE x2 for wide coalesced reads (MI355X_MICROARCH.md, HBM section)";
  python3 tools/pmc_summary.py $(find $OUT/pmc_fetch -name "*counter_collection.csv") $(find $OUT/pmc_write -name "*counter_collection.csv") ) > $OUT/pmc_fetch_write.txt
python3 tools/make_traffic_json.py cfg3 $(find $OUT/pmc_fetch -name "*counter_collection.csv") $(find $OUT/pmc_write -name "*counter_collection.csv") $OUT/bench_under_rocprof.json $TAG $COMMIT $OUT/traffic.json > /dev/null
# the plain (un-profiled) bench line LAST, with the traffic figures just measured on this binary in place (roofline.traffic_is_of_this_binary)
cp $OUT/traffic.json profiles/traffic.json
python3 bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
find $OUT -name "*.csv" -size +2M -delete
find $OUT -name "*.db" -delete
ls -la $OUT
