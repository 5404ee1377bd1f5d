"""Extract one kernel's gfx950 ISA from a --save-temps .s file (hipcc --offload-arch=gfx950 -O3 --save-temps) and print its register /
LDS figures; `--skeleton` lists only loads, stores, waits, barriers, branches and MFMAs with their line numbers.
    python tools/kernel_isa.py file.s 'gemm_dma_kernelILi64ELi64ELi2ELi2ELi3ELi0ELi64ELi1ELi0E' [out.s] [--skeleton]"""
import re
import sys

path, pat = sys.argv[1], sys.argv[2]
out = next((a for a in sys.argv[3:] if not a.startswith("--")), None)
lines = open(path).read().split("\n")
names = [l.split(":")[0] for l in lines if re.match(r"^_Z\w+:", l) and pat in l]
if len(names) != 1:
    sys.exit(f"{len(names)} kernels match {pat!r}: {names[:6]}")
name = names[0]
beg = next(i for i, l in enumerate(lines) if l.startswith(name + ":"))
end = next(i for i in range(beg, len(lines)) if re.match(r"^\s*\.size\s+" + re.escape(name) + ",", lines[i]))
body = lines[beg:end]
if out:
    open(out, "w").write("\n".join(body))
info = {}
for l in lines:
    m = re.match(r"\s*\.set " + re.escape(name) + r"\.(num_vgpr|num_agpr|numbered_sgpr|private_seg_size), (\d+)", l)
    if m:
        info[m.group(1)] = int(m.group(2))
kd = "\n".join(lines[end:end + 120])
for key in ("group_segment_fixed_size", "next_free_vgpr", "accum_offset"):
    m = re.search(r"\.amdhsa_" + key + r" (\d+)", kd)
    if m:
        info[key] = int(m.group(1))
print(name, len(body), "lines", info, "mfma", sum("v_mfma" in l for l in body), "scratch", sum("scratch_" in l for l in body))
if "--skeleton" in sys.argv:
    for i, l in enumerate(body):
        if re.search(r"s_waitcnt|global_load|global_store|s_barrier|ds_read|ds_write|s_cbranch|^\.LBB|v_mfma|s_nop|s_endpgm|buffer_|global_atomic", l):
            print(f"{i:5d} {l.strip()[:120]}")
