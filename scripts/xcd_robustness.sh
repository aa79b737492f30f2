#!/bin/bash
# XCD-aware tile order: sensitivity to the idle-slot allowance (UPK_XCD_SLACK) with the in-tree tuning and with NO
# tuning (cost-model choices: the situation of a workload that was never tuned in situ)
cd $GRAFT_REPO_ROOT
for tune in intree none; do
  if [ $tune = none ]; then export UPGPT_TUNE_FILE=/nonexistent.json; else unset UPGPT_TUNE_FILE; fi
  for hw in 32x24 32x32; do
    for cfg in "0 1.2" "2 1.2" "2 1.0" "2 0"; do   # (slack 0 = the built-in rule); XCD_ONLY=1: the built-in rule only
      if [ -n "$XCD_ONLY" ] && [ "$cfg" != "2 0" ]; then continue; fi
      set -- $cfg
      AB_HW=$hw UPK_XCD_SLACK=$2 bash scripts/ab_env.sh UPK_XCD_MAP $1 2>/dev/null | tail -1 | sed "s/^/tuning $tune $hw slack $2: /"
    done
  done
done
