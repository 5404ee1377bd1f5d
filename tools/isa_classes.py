"""One-line instruction-class string of every basic block of an extracted kernel ISA (tools/kernel_isa.py out.s) that holds many MFMAs:
M mfma  r ds_read  w ds_write  D global_load_lds  G global_load  S store  e v_exp  c v_cvt  v other VALU  s SALU  W s_waitcnt  B barrier
    python tools/isa_classes.py kernel.s [min_mfma]"""
import sys
lines = open(sys.argv[1]).read().split("\n")
mn = int(sys.argv[2]) if len(sys.argv) > 2 else 20
seq = []
for l in lines:
    t = l.strip()
    if not t or t.startswith(";") or t.startswith("."):
        if t.startswith(".LBB"):
            seq.append("\n" + t.split(":")[0] + ": ")
        continue
    op = t.split()[0]
    seq.append("M" if op.startswith("v_mfma") else "r" if op.startswith("ds_read") else "w" if op.startswith("ds_write") else
               "D" if op.startswith("global_load_lds") else "G" if op.startswith("global_load") else "S" if op.startswith("global_store") else
               "e" if op.startswith("v_exp") else "c" if op.startswith("v_cvt") else "W" if op.startswith("s_waitcnt") else
               "B" if op.startswith("s_barrier") else "n" if op.startswith("s_nop") else "J" if op.startswith(("s_cbranch", "s_branch")) else
               "v" if op.startswith("v_") else "s" if op.startswith("s_") else "?")
for blk in "".join(seq).split("\n"):
    if blk.count("M") >= mn:
        print(len(blk), blk)
