#!/bin/bash
# Run ON THE GPU BOX: per corpus file (512 copies, gzip level $1), ms of each deflate kernel and ns per input byte
R=$(cd "$(dirname "$0")/../.." && pwd)
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/ck
timeout 800 rocprofv3 --kernel-trace -f csv -d /tmp/ck -o t -- python $R/tools/dbg/corpus_kinds.py ${1:-4} > /tmp/ck.out 2>/dev/null
python - <<'PY'
import csv, glob
names = [l.split() for l in open("/tmp/ck.out") if l.startswith("FILE")]
f = glob.glob("/tmp/ck/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
calls, cur = [], {}
for r in rows:
    nm = r["Kernel_Name"]
    ms = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    if "deflate_plan" in nm:
        if cur: calls.append(cur)
        cur = {}
    for k in ("deflate_link", "deflate_match", "deflate_kernel"):
        if k in nm: cur[k] = cur.get(k, 0) + ms
if cur: calls.append(cur)
for n, c in zip(names, calls):
    size = int(n[2])
    print("%-14s %8d B ratio %s  link %6.2f  match %7.2f  seq %7.2f ms   match %.3f ns/B  seq(longest stream) %.1f ns/B" % (
        n[1], size, n[5], c.get("deflate_link", 0), c.get("deflate_match", 0), c.get("deflate_kernel", 0),
        c.get("deflate_match", 0) * 1e6 / (512 * size), c.get("deflate_kernel", 0) * 1e6 / size))
PY
