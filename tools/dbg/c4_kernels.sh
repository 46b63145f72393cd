#!/bin/bash
cd "$(dirname "$0")/../.."
R=$PWD
cd /tmp && export TMPDIR=/tmp
for cap in ""; do
  rm -rf /tmp/c4p
  timeout 600 rocprofv3 --kernel-trace --stats -f csv -d /tmp/c4p -o t -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-text-leg --no-deflate --no-verify $cap > /dev/null 2>&1
  echo "== cap '$cap'"
  python - <<PY
import csv, glob, collections
f = glob.glob("/tmp/c4p/**/*kernel_trace.csv", recursive=True)[0]
tot = collections.Counter(); cnt = collections.Counter()
for r in csv.DictReader(open(f)):
    nm = r["Kernel_Name"].split("(")[0][-40:]
    if "defl" in nm or "crc" in nm:
        tot[nm] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6; cnt[nm] += 1
for k, v in tot.most_common(): print("  %-42s %8.1f ms in %d launches" % (k, v, cnt[k]))
PY
done
