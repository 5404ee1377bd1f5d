"""gpurun helper: parity figures of the full-size fixtures for whichever library CLORA_LIB_PATH selects"""
import os, sys
sys.path.insert(0, os.getcwd())
from tests import full_cases as F
tag = sys.argv[1]
e = F.train_step_vs_fixture("cuda")
print(tag, "train512", {k: f"{v:.3e}" for k, v in e.items() if k in ("pred", "grads_sample", "grads_sample2", "grads_norm", "param_norm_worst")}, flush=True)
e = F.train_step_vs_fixture("cuda", "full_train_512_bs8_v2.safetensors")
print(tag, "train512_v2", {k: f"{v:.3e}" for k, v in e.items() if k in ("pred", "grads_sample", "grads_sample2", "grads_norm", "param_norm_worst")}, flush=True)
e = F.infer32_vs_fixture("cuda")
print(tag, "infer32", {k: (f"{v:.3e}" if isinstance(v, float) else v) for k, v in e.items() if k != "oracle_seconds"}, flush=True)
e = F.ddim_vs_fixture("cuda")
print(tag, "ddim50", {k: f"{v:.3e}" for k, v in e.items() if k in ("latents", "eps_step01", "latents_step05", "latents_step20")}, flush=True)
e = F.vae_512_vs_fixture("cuda")
print(tag, "vae512", {k: f"{v:.3e}" for k, v in e.items() if isinstance(v, float)}, flush=True)
