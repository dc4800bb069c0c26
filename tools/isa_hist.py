"""Instruction histogram of one kernel from the -save-temps assembly of tools/kernel_regs.py (/tmp/kregs):
    python tools/isa_hist.py decoder "fused_up_conv_kernel<32, 1, 2, 2, 1, 32, 8, 3" [top]"""
import collections, re, subprocess, sys
stem, sub = sys.argv[1], sys.argv[2]
top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
s = open(f"/tmp/kregs/{stem}-hip-amdgcn-amd-amdhsa-gfx950.s").read()
for m in re.finditer(r"^(_Z\w+):\s*; @", s, re.M):
    l = m.group(1)
    d = subprocess.run(["c++filt", l], capture_output=True, text=True).stdout.strip()
    if sub not in d:
        continue
    body = s[m.start():s.index(".Lfunc_end", m.start())]
    ins = [ln.strip().split()[0] for ln in body.split("\n") if ln.startswith("\t") and not ln.strip().startswith((".", ";"))]
    c = collections.Counter(ins)
    grp = lambda p, ex=(): sum(v for k, v in c.items() if k.startswith(p) and not k.startswith(ex))   # noqa: E731
    print(d[:100])
    print("total", sum(c.values()), "VALU", grp("v_", ("v_mfma",)), "SALU", grp("s_"), "MFMA", grp("v_mfma"), "DS", grp("ds_"),
          "VMEM", grp("global_") + grp("buffer_") + grp("scratch_"))
    for k, v in c.most_common(top):
        print(f"  {k:34s} {v}")
