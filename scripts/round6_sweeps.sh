#!/bin/bash
# Round 6, final build: the randomised sweeps of rounds 4-5 once more (the coarse pass's arithmetic, its guard constant, the first E-step of
# a call and the M-step forms of short calls changed this round)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r6_sweeps.txt
: > $OUT
for cmd in "scripts/coarse_sweep.py 30 626" "scripts/em_call_sweep.py" "scripts/guarded_sweep.py 150" "scripts/guarded_drift.py" "scripts/forms_sweep.py 60" "scripts/parity_sweep.py"; do
  echo "## python3 $cmd" >> $OUT
  timeout 900 python3 $cmd 2>&1 | tail -12 >> $OUT
done
SWEEP_SMALL=1 timeout 900 python3 scripts/coarse_sweep.py 30 627 2>&1 | tail -4 >> $OUT
tail -70 $OUT | cut -c1-400
