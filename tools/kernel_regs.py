"""Register / spill table of every kernel in one csrc/*.hip file (compiles it with -save-temps into /tmp).
usage: python tools/kernel_regs.py nerf_bwd_fused.hip [substring]"""
import os, re, subprocess, sys
here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, here)
from cips_3dplusplus_amd.build import FLAGS, FILE_FLAGS      # the shipped flags, per-file ones included
src = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else ""
stem = os.path.splitext(src)[0]
os.makedirs("/tmp/kregs", exist_ok=True)
subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, *FILE_FLAGS.get(src, []), "-save-temps=obj", "-c",
                os.path.join(here, "cips_3dplusplus_amd/csrc", src), "-o", f"/tmp/kregs/{stem}.o"], check=True, cwd="/tmp/kregs")
s = open(f"/tmp/kregs/{stem}-hip-amdgcn-amd-amdhsa-gfx950.s").read()
for b in s.split("  - .agpr_count:")[1:]:
    g = lambda k: re.search(r"\." + k + r":\s+(\S+)", b).group(1)
    name = subprocess.run(["c++filt", g("name")], capture_output=True, text=True).stdout.strip()
    if sub in name:
        print(f"{name[:90]:90s} vgpr {g('vgpr_count'):>3} agpr {b.split()[0]:>3} spill {g('vgpr_spill_count'):>3} sgpr {g('sgpr_count'):>3} "
              f"scratch {g('private_segment_fixed_size'):>4} lds {g('group_segment_fixed_size')}")
