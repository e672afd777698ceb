#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 150 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES --kernel-trace --output-format csv -d $OUT/pmc -- python3 $R/bench.py --config 5 --steps 3 --warmup 1 --launch eager --phased-variant 1 --no-steady-state --no-cpu-baseline --no-extras --no-live-traffic > $OUT/pmc.log 2>&1
python3 - $OUT <<'PY'
import csv, glob, sys, collections
v=collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(sys.argv[1]+'/pmc/**/*_counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'rsik::cont_' in r['Kernel_Name']:
            v[r['Kernel_Name'].split('rsik::')[1].split('<')[0]][r['Counter_Name']] += float(r['Counter_Value'])
for k,c in v.items():
    print(k, 'VALU per wave', round(c['SQ_INSTS_VALU']/c['SQ_WAVES'],1))
PY
rm -rf $OUT/pmc
