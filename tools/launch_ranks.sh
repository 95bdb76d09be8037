#!/bin/bash
# One process per GPU, no torch, no MPI: starts W copies of a rank program (default: the C client tests/c/abi_ranks.c built as
# ./abi_ranks) with --rank / --world / --id-file and a per-job nonce, waits for all of them, fails if any failed.
#   tools/launch_ranks.sh W [program [args...]]        e.g.  tools/launch_ranks.sh 8 ./abi_ranks --windows 100000
# DCE_LAUNCH_LATE="R:MS" starts rank R with --delay-ms MS (a rehearsal of a rank that comes late).
# Rank r runs on visible device r (the program's --device default); export HIP_VISIBLE_DEVICES to pick / permute the GPUs.
set -u
W=${1:?usage: launch_ranks.sh W [program [args...]]}; shift
PROG=${1:-./abi_ranks}; [ $# -gt 0 ] && shift
export DCE_COMM_NONCE="$$-$(date +%s%N | tail -c 10)"
export HSA_ENABLE_IPC_MODE_LEGACY=${HSA_ENABLE_IPC_MODE_LEGACY:-0}
IDFILE=${DCE_COMM_ID_FILE:-${TMPDIR:-/tmp}/dce_comm_id.$USER}
pids=()
for ((r = 0; r < W; ++r)); do
  LATE=${DCE_LAUNCH_LATE:-}
  if [ -n "$LATE" ] && [ "${LATE%%:*}" = "$r" ]; then
    "$PROG" --rank "$r" --world "$W" --id-file "$IDFILE" "$@" --delay-ms "${LATE##*:}" &
  else
    "$PROG" --rank "$r" --world "$W" --id-file "$IDFILE" "$@" &
  fi
  pids+=($!)
done
rc=0
for p in "${pids[@]}"; do wait "$p" || rc=1; done
[ $rc -eq 0 ] && echo "launch_ranks: all $W ranks OK" || echo "launch_ranks: FAILED" >&2
exit $rc
