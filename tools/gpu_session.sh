#!/bin/bash
# One GPU-box session: named steps, each under its own timeout; a step that times out ends the session (no GPU work after a
# hang).  Usage: tools/gpu_session.sh <outdir> <<'STEPS'
#   name|timeout_s|command ...
# STEPS
set -o pipefail
out=${1:-gpurun_out/session}
mkdir -p "$out"
while IFS='|' read -r name to cmd; do
  [ -z "$name" ] && continue
  echo "== $name" | tee -a "$out/session.log"
  timeout -k 10 "$to" bash -c "$cmd" > "$out/$name.log" 2>&1
  rc=$?
  echo "== $name rc=$rc" | tee -a "$out/session.log"
  tail -n 6 "$out/$name.log"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timeout -> stop" | tee -a "$out/session.log"; exit 1; fi
done
exit 0
