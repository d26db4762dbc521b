#!/bin/bash
# usage (GPU box): bash tools/gpu_suite.sh OUTDIR [pytest args...] — the -m gpu suite into OUTDIR/tests.txt; exit code 99 if the suite was killed by its time limit
# (a hang: nothing else should run on this box), else 0 so that measurement steps behind it still run after an ordinary test failure
out=$1; shift
mkdir -p $out
timeout -k 10 1000 python3 -m pytest tests -m gpu -q "$@" > $out/tests.txt 2>&1
rc=$?
echo "pytest rc=$rc" >> $out/tests.txt
tail -n 25 $out/tests.txt
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 99; fi
exit 0
