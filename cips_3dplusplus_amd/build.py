"""Builds libcips3d_hip.so (gfx950 code objects + C ABI) in-tree with hipcc.

    python -m cips_3dplusplus_amd.build [--force] [--keep-temps]

hipcc cross-compiles for gfx950 without a GPU, so this runs in the CPU-only container; the
resulting .so travels to the GPU box with the repository snapshot (it is git-ignored, not
gpurun-ignored).
"""
import concurrent.futures as cf
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(PKG, "libcips3d_hip.so")
ARCH = "gfx950"
SOURCES = ["bias_act.hip", "upfirdn2d.hip", "linear.hip", "camera.hip", "nerf.hip", "decoder.hip", "chain.hip", "conv3x3.hip", "forward.hip", "backward.hip", "nerf_bwd.hip", "nerf_bwd_fused.hip", "render_ops.hip", "rng.hip"]
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-ffp-contract=off", "-Wall",
         "-Wno-unused-function", "-fno-gpu-rdc", "-fgpu-flush-denormals-to-zero" if False else ""]
FLAGS = [f for f in FLAGS if f] + os.environ.get("CIPS3D_HIPCC_FLAGS", "").split()


def hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the HIP library cannot be built on this machine")


STAMP = LIB + ".srchash"      # content hash of the sources the library was built from (travels with the .so)


def source_hash():
    """sha256 over every csrc/*.hip, csrc/*.h, the public header, the flags and this file: file times do not survive a
    snapshot copy to the GPU box, contents do."""
    import hashlib
    files = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h")))
    files += [os.path.join(os.path.dirname(PKG), "include", "cips3d_hip.h"), os.path.abspath(__file__)]
    h = hashlib.sha256(" ".join(FLAGS).encode())
    for f in files:
        if os.path.exists(f):
            h.update(os.path.basename(f).encode())
            with open(f, "rb") as fh:
                h.update(fh.read())
    return h.hexdigest()


def up_to_date():
    if not (os.path.exists(LIB) and os.path.exists(STAMP)):
        return False
    with open(STAMP) as fh:
        return fh.read().strip() == source_hash()


# per-file flags (chain.hip: see the build note in its header)
FILE_FLAGS = {"chain.hip": ["-fno-slp-vectorize"]}


def _compile(src, keep_temps):
    obj = os.path.join(OBJ, src.replace(".hip", ".o"))
    cmd = [hipcc(), *FLAGS, *FILE_FLAGS.get(src, []), "-c", os.path.join(CSRC, src), "-o", obj]
    if keep_temps:
        cmd += ["-save-temps=obj", "-Rpass-analysis=kernel-resource-usage"]
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=OBJ)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed on {src}:\n{r.stdout}\n{r.stderr}")
    return obj, r.stderr


def build_library(force=False, keep_temps=False, verbose=False):
    if not force and up_to_date():
        return LIB
    os.makedirs(OBJ, exist_ok=True)
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    with cf.ThreadPoolExecutor(max_workers=min(6, len(srcs))) as ex:
        results = list(ex.map(lambda s: _compile(s, keep_temps), srcs))
    objs = [o for o, _ in results]
    if verbose:
        for _, log in results:
            sys.stderr.write(log)
    r = subprocess.run([hipcc(), "-shared", "-fPIC", f"--offload-arch={ARCH}", *objs, "-o", LIB + ".tmp"],
                       capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    os.replace(LIB + ".tmp", LIB)
    with open(STAMP, "w") as fh:
        fh.write(source_hash() + "\n")
    return LIB


if __name__ == "__main__":
    path = build_library(force="--force" in sys.argv, keep_temps="--keep-temps" in sys.argv,
                         verbose="--keep-temps" in sys.argv or "-v" in sys.argv)
    print(path)
