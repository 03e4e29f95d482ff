#!/bin/bash
# Sample GPU clock / power while a command runs: tools/clock_watch.sh <out-file> <command...>
OUT=$1; shift
( while true; do /opt/rocm/bin/rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|Power|Temperature \(Sensor (edge|junction)" | tr '\n' ' '; echo; sleep 0.25; done ) > $OUT 2>&1 &
W=$!
"$@"
kill $W 2>/dev/null
wait $W 2>/dev/null
