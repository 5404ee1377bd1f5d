#!/bin/bash
# Round 5, call 11: re-tune of the launch table under the round-5 library (tile order 3, tiles 59 / 79 in the candidate set); incumbents
# defend their entries with a 2 % margin; then old table vs new table on this box (two interleaved bench lines each).
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
T=controllora_amd/gemm_tuning_gfx950.json
cp $T /tmp/table_old.json
ALL=1,2,3,4,5,6,7,8,21,22,23,26,31,32,33,41,42,43,51,52,53,54,55,56,57,58,59
( time timeout 900 python tools/tune_gemm.py --merge --plain-only --infer-batch 0 --cfgs $ALL ) > gpurun_out/r05_retune_plain_train.log 2>&1
tail -3 gpurun_out/r05_retune_plain_train.log
( time timeout 600 python tools/tune_gemm.py --merge --patch-only --infer-batch 0 --cfgs 71,72,73,74,75,76,79,21,41,22,42 ) > gpurun_out/r05_retune_patch_train.log 2>&1
tail -3 gpurun_out/r05_retune_patch_train.log
cp $T gpurun_out/gemm_tuning_retuned.json
python - <<'PY'
import json
a = json.load(open("/tmp/table_old.json"))["table"]; b = json.load(open("controllora_amd/gemm_tuning_gfx950.json"))["table"]
ch = {k: (a.get(k), b[k]) for k in b if a.get(k) != b[k]}
print(len(ch), "entries changed"); [print(k, v) for k, v in list(ch.items())[:60]]
PY
Q="--no-roofline --no-cpu-baseline --no-ddim --no-full-step --no-calibration --steps 20 --warmup 5"
for arm in new old new old; do
  if [ $arm = old ]; then cp /tmp/table_old.json $T; else cp gpurun_out/gemm_tuning_retuned.json $T; fi
  python bench.py $Q 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$arm', d['ms_per_step'], d['timed_windows']['ms_per_step'])" | tee -a gpurun_out/r05_retune_ab.txt
done
