#!/bin/bash
# Runs groups of GPU tests one after another; stops at the first group that did not end in a normal pytest verdict
# (exit 0 = passed, 1 = some tests failed); anything else (crash, timeout, GPU fault) ends the run.
# usage: scripts/gpu_suite.sh <log> <timeout_s> <pytest args of group 1> -- <group 2> -- ...
log=$1; shift; tmo=$1; shift
mkdir -p "$(dirname "$log")"; : > "$log"
args=()
run() {
  echo "=== pytest ${args[*]}" >> "$log"
  timeout -k 10 "$tmo" python -m pytest "${args[@]}" >> "$log" 2>&1
  rc=$?; echo "=== rc=$rc" >> "$log"
  if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then echo "=== abnormal end, stopping" >> "$log"; tail -30 "$log"; exit $rc; fi
}
for a in "$@"; do
  if [ "$a" == "--" ]; then run; args=(); else args+=("$a"); fi
done
[ ${#args[@]} -gt 0 ] && run
grep -E "^=== |passed|failed" "$log" | tail -40
