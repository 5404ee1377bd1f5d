import os, sys, torch
sys.path.insert(0, os.getcwd())
from tests import full_cases as F
from controllora_amd import kernels as K, unet as U, ops
def run(tag):
    e = F.train_step_vs_fixture("cuda")
    print(tag, {k: (f"{v:.3e}" if isinstance(v, float) else v) for k, v in e.items() if k in ("pred","grads_sample","grads_sample2","grads_norm","param_norm_worst","param_norm_worst_name")}, flush=True)
run("all_on")
K.DEFER_FINISH=False; run("nodefer")
U.CAT_IN_PLACE=False; run("nodefer_nocat")
ops.GROUP_TEXT_KV=False; run("nodefer_nocat_nogroup")
ops.PERSISTENT_CONV_PACKS=False; K.DEFER_UNPACK=False; run("all_off")
run("all_off_again")
