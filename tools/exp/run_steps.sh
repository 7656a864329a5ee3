#!/bin/bash
# Run the commands given on stdin (one per line: "<seconds> <name> <command...>") one after the other on the GPU box.  A step that
# times out or is killed (exit 124 / 137) ends the script: no further GPU step is started after a hung one.  Ordinary failures are
# logged and the next step runs.  Output of each step: gpurun_out/<name>.log
mkdir -p gpurun_out
export TMPDIR=/tmp
while read -r secs name cmd; do
  [ -z "$secs" ] && continue
  case "$secs" in \#*) continue;; esac
  echo "=== $(date +%H:%M:%S) $name (limit ${secs}s): $cmd"
  timeout -k 10 "$secs" bash -c "$cmd" > "gpurun_out/$name.log" 2>&1
  rc=$?
  echo "    rc=$rc  $(tail -n 1 gpurun_out/$name.log | cut -c1-200)"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step $name hit its limit: stopping"; exit $rc; fi
done
echo "=== $(date +%H:%M:%S) all steps done"
