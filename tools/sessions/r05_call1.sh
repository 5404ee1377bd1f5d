#!/bin/bash
# Round 5, call 1: (a) same-box A/B of the library of commit 382a909 (round-4 closing evidence) against HEAD's -- VERDICT r04 item 1b --
# with the new calibration / three-window bench line; (b) the whole GPU suite under the new defaults (tile_order 3, tile 79 in the
# patch-conv cases); (c) tile 79 timed against the table entries of every patch-eligible signature; (d) one full bench line.
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python tools/box_calib.py 2>&1 | grep BOX_CALIB | tee gpurun_out/r05_box_calib_1.txt
Q="--no-roofline --no-cpu-baseline --no-ddim --no-full-step --steps 20 --warmup 5"
OLD="CLORA_LIB_PATH=$R/controllora_amd/_build_v_r04close/libclora.so CLORA_ABI_ANY=1"
: > gpurun_out/r05_ab_lib_382a909_vs_head.txt
for arm in head old headauto old head; do
  case $arm in
    head) env python bench.py $Q 2>/dev/null | grep '^{' > /tmp/line.json ;;
    headauto) env CLORA_TILE_ORDER=auto python bench.py $Q 2>/dev/null | grep '^{' > /tmp/line.json ;;
    old) env $OLD python bench.py $Q 2>/dev/null | grep '^{' > /tmp/line.json ;;
  esac
  python - "$arm" <<'PY' | tee -a gpurun_out/r05_ab_lib_382a909_vs_head.txt
import json, sys
d = json.load(open("/tmp/line.json"))
c = d.get("calibration") or {}
print(sys.argv[1], "ms_per_step", d["ms_per_step"], "windows", d["timed_windows"]["ms_per_step"], "calib_gemm_us", c.get("gemm8192_cfg1_us"),
      "copy_GBps", c.get("copy256MB_GBps"), "mfma_clock_MHz", c.get("mfma_clock_MHz"), "mfma_probe_TF", c.get("mfma_probe_TFLOPs"), "sysfs", c.get("sysfs"), "loss", d["loss"])
PY
  cp /tmp/line.json gpurun_out/r05_ab_line_$arm.json
done
( time timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider ) > gpurun_out/r05_gputest_1.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r05_gputest_1.log
grep -E "passed|failed|rc=|^FAILED|^ERROR" gpurun_out/r05_gputest_1.log | tail -6
( time timeout 420 python tools/tune_gemm.py --merge --patch-only --cfgs 79 ) > gpurun_out/r05_tune_tile79.log 2>&1
tail -25 gpurun_out/r05_tune_tile79.log
cp controllora_amd/gemm_tuning_gfx950.json gpurun_out/gemm_tuning_after_tile79.json
( time timeout 900 python bench.py --trace-out gpurun_out/r05_kernel_stats_1.json ) > gpurun_out/r05_bench_1.log 2>&1
grep '^{' gpurun_out/r05_bench_1.log > gpurun_out/r05_bench_1.json
head -c 600 gpurun_out/r05_bench_1.json; echo
python - <<'PY'
import json
d = json.load(open("gpurun_out/r05_bench_1.json"))
print({k: d[k] for k in ("ms_per_step", "value", "timed_windows", "calibration")})
print(d["roofline"]["frac"], d["roofline"]["family_ms_per_step"], d["roofline"]["traffic"], d["ddim50"])
PY
