#!/bin/bash
# round 3, GPU call 3: tests on the rebuilt library, the subdivision probe, the FETCH_SIZE calibration, an eighth of the headline frame
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03
mkdir -p $O
cd $R
timeout -k 10 400 python -m pytest tests -m gpu -x -q > $O/pytest2.log 2>&1 || { tail -30 $O/pytest2.log; exit 1; }
tail -2 $O/pytest2.log
timeout -k 10 300 python scripts/r03_probe2.py > $O/probe2.log 2>&1; tail -12 $O/probe2.log
export TMPDIR=/tmp
(cd /tmp && timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch_calib -- $R/build/fetch_calib > $O/fetch_calib.log 2>&1); tail -12 $O/fetch_calib.log
cd $R
timeout -k 10 120 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --share-of 8 > $O/bench_share8.json 2> $O/bench_share8.err; python -c "
import json; p=json.load(open('$O/bench_share8.json')); print('share8 ms', p['ms_per_step'], 'one lane', p['roofline']['kernel_avg_ms_one_lane'])"
timeout -k 10 120 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench1.json 2> $O/bench1.err; python -c "
import json; p=json.load(open('$O/bench1.json')); print('full ms', p['ms_per_step'], p['value'], 'crc', p['config']['frame_crc'], p['config']['d2h_ms'], p['config']['crc_ms'])"
