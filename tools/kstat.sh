#!/bin/bash
# usage: tools/kstat.sh <tag> <stage> [pattern]  -- per-kernel avg durations of one microbench stage (GPU box)
TAG=$1; STAGE=$2; PAT=${3:-.}
R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/ks_$TAG
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ks_$TAG -- python3 $R/tools/microbench.py $STAGE 5 > $R/gpurun_out/ks_$TAG.log 2>&1
cd $R && python - <<PY
import csv,glob,re
f=glob.glob('gpurun_out/ks_$TAG/**/*kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    if re.search(r'$PAT', r['Name']):
        print('$TAG', re.sub(r'\(.*','',r['Name'])[:60], 'calls', r['Calls'], 'avg us', round(float(r['AverageNs'])/1e3,2), 'min', round(float(r['MinNs'])/1e3,2), 'max', round(float(r['MaxNs'])/1e3,2))
PY
