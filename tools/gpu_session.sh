#!/bin/bash
# Helper for gpurun calls: runs the given steps (each a quoted command line) one after the other; a step that was killed by its
# timeout (exit code >= 124) ends the session -- no further GPU step is started behind a hung one -- while an ordinary failure
# (a failing test) is reported and the session goes on.   usage: bash tools/gpu_session.sh "cmd 1" "cmd 2" ...
for step in "$@"; do
  echo "[session] $step"
  bash -o pipefail -c "$step"
  rc=$?
  echo "[session] rc=$rc"
  if [ $rc -ge 124 ]; then echo "[session] step killed or timed out: stopping"; exit $rc; fi
done
exit 0
