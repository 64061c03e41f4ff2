#!/bin/bash
# rocprofv3 --pmc passes over a short wavefront run (one counter group per pass: the SQ / TA / TCP / TCC blocks have few slots each).
#   tools/gpu_pmc_passes.sh OUTDIR "<probe args>" PASS...        PASS = name:COUNTER,COUNTER,...
# e.g. tools/gpu_pmc_passes.sh gpurun_out/pmc "2309 4096 4 -- FTN_TRACE4=0 ''" sq1:SQ_WAVES,SQ_INSTS_VALU fetch:FETCH_SIZE
# Summaries per kernel: OUTDIR/<name>_summary.txt (tools/pmc_summary.py).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$1; shift
ARGS=$1; shift
case "$OUT" in /*) ;; *) OUT="$ROOT/$OUT";; esac
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for pass in "$@"; do
  name=${pass%%:*}; ctrs=${pass#*:}; ctrs=${ctrs//,/ }
  echo "== pass $name: $ctrs"
  rm -rf "$OUT/$name"
  # the program itself follows `--` (no env / bash -c hop: the profiler's preloaded library initialises the GPU first)
  eval "PROBE_REPS=1 timeout -k 10 300 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d '$OUT/$name' -o p -- python3 '$ROOT/tools/gpu_t4_probe.py' $ARGS" > "$OUT/$name.log" 2>&1
  rc=$?
  if [ $rc -ne 0 ]; then echo "pass $name: rc $rc"; tail -3 "$OUT/$name.log"; fi
  if [ $rc -ge 124 ]; then echo "pass $name was killed: stopping"; exit 1; fi
  python3 "$ROOT/tools/pmc_summary.py" "$OUT/$name" > "$OUT/${name}_summary.txt" 2>&1
  grep -h "^{" "$OUT/$name.log" | cut -c1-200
  find "$OUT/$name" -type f | head -5; du -sh "$OUT/$name" | cut -f1; find "$OUT/$name" -type f ! -name "*counter_collection.csv" -delete 2>/dev/null   # keep the counter csv (summaries are made from it), drop the traces
done
exit 0
