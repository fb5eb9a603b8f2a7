#!/bin/bash
# FETCH_SIZE per kernel of one command: tools/pmc_fetch_one.sh <tag> <cmd...>   (rocprofv3 --pmc FETCH_SIZE --kernel-trace; x2 gfx950 correction applied)
TAG=$1; shift
export TMPDIR=/tmp
OUT=gpurun_out/pmc_$TAG
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT -o f -- "$@" > $OUT/log.txt 2>&1
DB=$(find $OUT -name "*_results.db" | head -1)
python - "$DB" <<'PY'
import sqlite3, sys, re
cur = sqlite3.connect(sys.argv[1]).cursor()
rows = cur.execute("select kernel_name, count(*), sum(value), sum(end-start) from counters_collection where counter_name='FETCH_SIZE' group by kernel_name").fetchall()
for k, n, kb, ns in sorted(rows, key=lambda r: -r[2])[:12]:
    print(f"{re.sub(r'[(].*', '', k)[:70]:70s} n={n:5d} fetch x2 = {2 * kb / n / 1024:9.1f} MB/launch  {ns / n / 1e3:8.1f} us")
PY
