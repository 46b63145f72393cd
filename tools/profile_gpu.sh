#!/bin/bash
# Run ON THE GPU BOX (via gpurun): kernel-trace stats + HBM PMC passes for bench.py.
# usage: tools/profile_gpu.sh <tag> [bench args...]
# Summaries land in gpurun_out/prof_<tag>/ ; copy what you want judged to profiles/.
set -u
TAG=${1:-r04}; shift || true
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-text-leg --no-secondary --no-host-path --no-verify $*"
echo "== kernel trace: $BENCH"
rocprofv3 --kernel-trace --stats -f csv -d "$OUT/trace" -o trace -- $BENCH > "$OUT/bench_under_rocprof.json" 2> "$OUT/trace.log"
tail -1 "$OUT/bench_under_rocprof.json" | cut -c1-400
find "$OUT/trace" -name '*kernel_stats*.csv' -exec cp {} "$OUT/kernel_stats.csv" \;
# PMC passes (separate runs, no tracing domains besides kernel-trace): HBM bytes
for C in FETCH_SIZE WRITE_SIZE; do
  echo "== pmc $C"
  rocprofv3 --kernel-trace --pmc $C -f csv -d "$OUT/pmc_$C" -o pmc -- $BENCH > /dev/null 2> "$OUT/pmc_$C.log"
  find "$OUT/pmc_$C" -name '*counter_collection*.csv' -exec cp {} "$OUT/pmc_$C.csv" \;
done
python - "$OUT" "$BENCH" <<'PY'
import csv, sys, os, json, collections
out, cmd = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    p = os.path.join(out, "pmc_%s.csv" % c)
    if not os.path.exists(p):
        print(c, "missing"); continue
    for r in csv.DictReader(open(p)):
        if r.get("Counter_Name") == c and "md::" in r["Kernel_Name"]:
            k = r["Kernel_Name"].split("(")[0].split("<")[0].split("::")[-1].split(" ")[-1]
            agg[k][c].append(float(r["Counter_Value"]))
line = json.loads(open(os.path.join(out, "bench_under_rocprof.json")).read().strip().splitlines()[-1])
cfg = line["config"]
wide = {"inflate_wave_kernel": line["roofline"]["algorithmic_bytes_per_launch"] - cfg["streams_per_gpu"] * cfg["stream_bytes"]}
import subprocess
try:
    commit = subprocess.check_output(["git", "-C", os.path.dirname(out.rstrip("/")).rsplit("/gpurun_out", 1)[0], "rev-parse", "--short", "HEAD"], stderr=subprocess.DEVNULL).decode().strip()
except Exception:  # the GPU box gets a snapshot without .git: the commit is left in a file before the call
    try:
        commit = open(os.path.join(os.path.dirname(out.rstrip("/")).rsplit("/gpurun_out", 1)[0], "tools", ".profile_commit")).read().strip()
    except OSError:
        commit = "?"
sys.path.insert(0, os.path.dirname(out.rstrip("/")).rsplit("/gpurun_out", 1)[0])
import bench as _bench
summ = {"commit": commit, "csrc_digest": _bench._csrc_digest(), "deflate_calls": (line.get("deflate") or {}).get("steps", 0) + 1, "command": cmd.replace(os.path.dirname(out.rstrip("/")).rsplit("/gpurun_out", 1)[0] + "/", ""),
        "note": "FETCH_SIZE / WRITE_SIZE are KiB per dispatch (rocprofv3, separate --pmc passes); means over the "
                "dispatches of the run.  traffic_bytes_per_launch_raw = (FETCH + WRITE) x 1024.  On gfx950 FETCH_SIZE "
                "counts a wide coalesced read (16 bytes per lane) at half its bytes (MI355X_MICROARCH.md, HBM); the "
                "only such stream here is the inflate kernel's read of the compressed input (wide_read_bytes per "
                "launch, loaded with global_load_dwordx4), so traffic_bytes_per_launch = raw + wide_read_bytes / 2.  "
                "The other reads (8-byte loads of match sources, the deflate kernels' 4- to 16-byte loads) are "
                "taken as counted."}
for k, d in agg.items():
    f = sum(d["FETCH_SIZE"]) / max(1, len(d["FETCH_SIZE"]))
    w = sum(d["WRITE_SIZE"]) / max(1, len(d["WRITE_SIZE"]))
    raw = int((f + w) * 1024)
    summ[k] = {"launches": len(d["FETCH_SIZE"]), "fetch_KiB": round(f, 1), "write_KiB": round(w, 1),
               "traffic_bytes_per_launch_raw": raw, "wide_read_bytes": wide.get(k, 0),
               "traffic_bytes_per_launch": raw + wide.get(k, 0) // 2}
    print(k, summ[k])
json.dump(summ, open(os.path.join(out, "pmc_summary.json"), "w"), indent=1)
PY
ls -la "$OUT"
