#!/bin/bash
# usage: tools/gpu_when_free.sh <timeout-seconds> '<command>'
# Runs gpurun; when no GPU slot / box is free (exit code 3: nothing ran, nothing was charged) waits and asks again.
# Any other outcome (the command ran, was refused, failed) ends the script with gpurun's exit code.
for attempt in $(seq 1 20); do
    /usr/local/graft/bin/gpurun --timeout "$1" -- "$2"
    rc=$?
    if [ $rc -ne 3 ]; then exit $rc; fi
    sleep 75
done
exit 3
