#!/bin/bash
# gpurun call 2: GPU test suite (all), GEMM variant pilot sweep, PMC diagnosis of the main loop
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -q -s -p no:cacheprovider ) > gpurun_out/gputest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/gputest.log
grep -E "FULL_TOPOLOGY|DDIM_LATENT|RANK256|CHAIN|passed|failed|rc=|^FAILED" gpurun_out/gputest.log | tail -40
( time timeout 900 python tools/gemm_pilot.py gpurun_out/gemm_pilot.json ) > gpurun_out/gemm_pilot.log 2>&1
tail -25 gpurun_out/gemm_pilot.log
cd /tmp
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" \
            "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC" ; do
  tag=$(echo $pass | cut -c4-12 | tr -d ' ')
  rm -rf /tmp/pmc_$tag
  timeout 600 rocprofv3 --pmc $pass -d /tmp/pmc_$tag -o pmc -- python $GRAFT_REPO_ROOT/tools/gemm_pilot.py --pmc > $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag.log 2>&1
  db=$(find /tmp/pmc_$tag -name "*.db" | head -1)
  python $GRAFT_REPO_ROOT/tools/pmc_dump.py $db $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag.json gemm_dma > $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag.txt 2>&1
done
cd $GRAFT_REPO_ROOT
head -30 gpurun_out/pmc_*.txt | cut -c1-400
