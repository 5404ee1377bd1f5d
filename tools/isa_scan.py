"""Scan the compiled gfx950 ISA of every kernel for loads that are drained one at a time.

A streaming loop written with per-thread trip counts or with loads under `if (in_range)` compiles to
    global_load ... ; s_waitcnt vmcnt(0) ; use
per element: one 16-byte load in flight per thread, one L2 / fabric round trip per row.  This is how the GroupNorm passes, the
adapter weight-gradient kernel, the split-K finish and the partial folds lost 2-4x (DESIGN.md sections 5 and 7).  The script
compiles controllora_amd/csrc/*.hip with --save-temps into a scratch directory and prints, per kernel, the sequence of
    L global load   D LDS-DMA load   c scalar load   w counted vmcnt wait   0 vmcnt(0)   s store   A atomic
together with the number of single / double load -> vmcnt(0) groups.  A kernel with many of them is worth opening.
    python tools/isa_scan.py [--min 3] [--all] [file.hip ...]"""
import argparse, glob, os, re, shutil, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("files", nargs="*")
ap.add_argument("--min", type=int, default=3, help="report kernels with at least this many single/double load-drain groups")
ap.add_argument("--all", action="store_true", help="print every kernel's sequence")
args = ap.parse_args()

def demangle(name):
    for tool in ("c++filt", "/opt/rocm/lib/llvm/bin/llvm-cxxfilt"):
        if shutil.which(tool) or os.path.exists(tool):
            out = subprocess.run([tool, name], capture_output=True, text=True).stdout.strip()
            if out:
                return out.replace("(anonymous namespace)::", "")
    return name


tmp = tempfile.mkdtemp(prefix="isa_scan_")
try:
    os.makedirs(os.path.join(tmp, "controllora_amd", "csrc"))
    os.makedirs(os.path.join(tmp, "include"))
    for f in glob.glob(os.path.join(ROOT, "controllora_amd", "csrc", "*")):
        shutil.copy(f, os.path.join(tmp, "controllora_amd", "csrc"))
    shutil.copy(os.path.join(ROOT, "include", "clora.h"), os.path.join(tmp, "include"))
    wd = os.path.join(tmp, "controllora_amd", "csrc")
    srcs = [os.path.basename(f) for f in args.files] or sorted(os.path.basename(f) for f in glob.glob(os.path.join(wd, "*.hip")))
    for src in srcs:
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-c", src, "-o", src + ".o", "--save-temps"],
                       cwd=wd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        asm = os.path.join(wd, src[:-4] + "-hip-amdgcn-amd-amdhsa-gfx950.s")
        txt = open(asm).read()
        rows = []
        for name in dict.fromkeys(re.findall(r"^(_Z\w+):", txt, re.M)):
            m = re.search(r"^" + re.escape(name) + r":.*?$", txt, re.M)
            body = txt[m.start():txt.find("s_endpgm", m.start())]
            if "s_endpgm" not in txt[m.start():] or len(body) < 50:
                continue
            seq = ""
            for line in body.split("\n"):
                line = line.strip()
                if line.startswith("global_load_lds"): seq += "D"
                elif line.startswith(("global_load", "buffer_load")): seq += "L"
                elif line.startswith("s_load"): seq += "c"
                elif line.startswith("s_waitcnt") and "vmcnt(0)" in line: seq += "0"
                elif line.startswith("s_waitcnt") and "vmcnt" in line: seq += "w"
                elif line.startswith("global_store"): seq += "s"
                elif line.startswith("global_atomic"): seq += "A"
            vm = seq.replace("c", "")
            n1, n2 = len(re.findall(r"(?<![LD])L0", vm)), len(re.findall(r"(?<![LD])LL0", vm))
            meta = re.search(r"\.name:\s+" + re.escape(name) + r"\b.*?\.vgpr_count:\s+(\d+)", txt, re.S)
            spill = re.search(r"\.name:\s+" + re.escape(name) + r"\b.*?\.private_segment_fixed_size:\s+(\d+)", txt, re.S)
            rows.append((n1 + n2, n1, n2, name, seq, meta.group(1) if meta else "?", spill.group(1) if spill else "?"))
        print(f"== {src}: {len(rows)} kernels")
        for tot, n1, n2, name, seq, vg, sp in sorted(rows, reverse=True):
            if args.all or tot >= args.min:
                short = demangle(name)
                print(f"  {tot:3d} (single {n1}, double {n2})  vgpr {vg:>3s} spill {sp:>3s}  {short[:90]}\n        {seq[:160]}")
finally:
    shutil.rmtree(tmp, ignore_errors=True)
