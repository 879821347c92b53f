#!/bin/bash
# Run ON the GPU box: the given commands one after the other (each a single quoted argument), stopping at the first that was KILLED or
# TIMED OUT (exit code >= 124) -- a failed assertion does not stop the sequence, a hung or faulted GPU step does.
for C in "$@"; do
  echo "=== $C" ; bash -c "$C"; rc=$?
  echo "=== exit $rc"
  [ $rc -ge 124 ] && exit $rc
done
exit 0
