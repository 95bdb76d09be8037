#!/bin/bash
# Sample rocm-smi (sclk, socket power) every 0.5 s while a command runs:  tools/smi_watch.sh <log> <command...>
LOG=$1; shift
( while true; do rocm-smi --showclocks --showpower --json 2>/dev/null | python3 -c "
import sys,json
try:
    d=json.load(sys.stdin)['card0']; print(d.get('sclk clock speed:'), d.get('Current Socket Graphics Package Power (W)'))
except Exception as e: pass
"; sleep 0.5; done ) > "$LOG" 2>/dev/null &
W=$!
"$@"
RC=$?
kill $W 2>/dev/null
exit $RC
