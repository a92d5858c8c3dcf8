#!/bin/bash
# One GPU-box session: named steps, each under its own timeout; a step that times out ends the session (no GPU work after a
# hang).  Usage: tools/gpu_session.sh <outdir> <<'STEPS'
#   name|timeout_s|command ...
# STEPS
# The step list is read on fd 3 and every step runs with stdin from /dev/null, so a step that reads its stdin cannot swallow
# the steps behind it.
set -o pipefail
out=${1:-gpurun_out/session}
mkdir -p "$out"
while IFS='|' read -r -u 3 name to cmd; do
  [ -z "$name" ] && continue
  case "$to" in ''|*[!0-9]*) echo "== $name: bad timeout '$to' -> stop" | tee -a "$out/session.log"; exit 2;; esac
  echo "== $name" | tee -a "$out/session.log"
  timeout -k 10 "$to" bash -c "$cmd" > "$out/$name.log" 2>&1 < /dev/null
  rc=$?
  echo "== $name rc=$rc" | tee -a "$out/session.log"
  tail -n 6 "$out/$name.log"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timeout -> stop" | tee -a "$out/session.log"; exit 1; fi
  if [ $rc -ge 125 ] && [ $rc -le 127 ]; then echo "step could not be started -> stop" | tee -a "$out/session.log"; exit 2; fi
done 3<&0
exit 0
