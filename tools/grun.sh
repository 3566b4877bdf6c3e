#!/bin/bash
# tools/grun.sh <log name> [command...] -- on the GPU box: run a command with gpurun_out/r6 in place, output to gpurun_out/r6/<log name> and the terminal.
# Without a command: the round's standard check -- the GPU suite, smoke() and the driver's bench command (what profiles/r0N_gputest_*.txt record).
#   gpurun --timeout 3000 -- 'bash tools/grun.sh gputest.txt'
set -o pipefail
log="$1"; shift || { echo "usage: tools/grun.sh <log name> [command...]" >&2; exit 2; }
mkdir -p gpurun_out/r6
if [ $# -gt 0 ]; then
  "$@" 2>&1 | tee "gpurun_out/r6/$log"
else
  { python -m pytest tests -q -m gpu --durations=10 2>&1 | tail -40
    python -c 'import __graft_entry__ as g; g.smoke()' 2>&1 | tail -3
    python bench.py --steps 20 --warmup 5 2>&1 | tail -1; } | tee "gpurun_out/r6/$log"
fi
