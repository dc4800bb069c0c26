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
SOURCES = ["bias_act.hip", "upfirdn2d.hip", "linear.hip", "camera.hip", "nerf.hip", "decoder.hip", "chain.hip", "conv3x3.hip", "forward.hip", "backward.hip", "decoder_grad.hip", "nerf_bwd.hip", "nerf_bwd_fused.hip", "render_ops.hip", "rng.hip", "optim.hip"]
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
    # (per-file flags included: a knob that only changes those -- CIPS3D_CHAIN_SLP -- must invalidate the library too; a probe of
    # round 4 compared two "builds" that were the same file because it did not)
    h = hashlib.sha256((" ".join(FLAGS) + " | " + repr(sorted(FILE_FLAGS.items()))).encode())
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


# per-file flags.  chain.hip: hipcc's SLP vectoriser packs the ToRGB fold's channel-1 / channel-2 accumulations into v_pk_fma_f32 chains
# one of whose forms -- op_sel:[0,1,0], the low lane reading the HIGH register of the src1 pair -- returns run-to-run different sums in
# the bf16 chain kernel (named and reproduced by hand in profiles/r05_slp_fold_cause.md); the shipped fold spells its FMAs in inline
# asm as well, and tests/test_host.py checks the kernel's assembly for the form
FILE_FLAGS = {"chain.hip": [] if os.environ.get("CIPS3D_CHAIN_SLP") == "1" else ["-fno-slp-vectorize"]}


def _compile(src, keep_temps):
    # per-process object names: two builders that slipped past the lock (different checkouts of one tree on a shared
    # file system) never write the same file
    obj = os.path.join(OBJ, os.path.basename(src).replace(".hip", f".{os.getpid()}.o"))
    cmd = [hipcc(), *FLAGS, *FILE_FLAGS.get(src, []), "-c", os.path.join(CSRC, src), "-o", obj]
    if keep_temps:
        cmd += ["-save-temps=obj", "-Rpass-analysis=kernel-resource-usage"]
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=OBJ)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed on {src}:\n{r.stdout}\n{r.stderr}")
    return obj, r.stderr


def build_library(force=False, keep_temps=False, verbose=False):
    """Compile + link under an exclusive lock.  Several ranks of one job (bench.py --gpus N, torchrun) may find the library
    stale at the same moment: the first one in builds, the others wait on the lock and then find it up to date.  `force`
    means "rebuild unless somebody else just did": the hash is re-checked after the lock is taken."""
    import fcntl
    import time
    if not force and up_to_date():
        return LIB
    os.makedirs(OBJ, exist_ok=True)
    with open(os.path.join(OBJ, ".build.lock"), "w") as lock:
        t_wait = time.time()
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            # built by another process while this one waited for the lock (or, unforced, nothing to do)
            if up_to_date() and (not force or os.path.getmtime(STAMP) >= t_wait):
                return LIB
            return _build_locked(keep_temps, verbose)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _sweep_objects():
    """Objects (and -save-temps files) of an earlier build that did not finish: the lock is held, so nothing in OBJ is in use."""
    for f in os.listdir(OBJ):
        if f.endswith((".o", ".tmp")):
            try:
                os.remove(os.path.join(OBJ, f))
            except OSError:
                pass


def _build_locked(keep_temps, verbose):
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    _sweep_objects()
    try:
        with cf.ThreadPoolExecutor(max_workers=min(6, len(srcs))) as ex:
            results = list(ex.map(lambda s: _compile(s, keep_temps), srcs))
    except BaseException:
        _sweep_objects()              # a failed or interrupted compile leaves nothing behind (the objects travel with gpurun snapshots)
        raise
    objs = [o for o, _ in results]
    if verbose:
        for _, log in results:
            sys.stderr.write(log)
    tmp = f"{LIB}.{os.getpid()}.tmp"
    r = subprocess.run([hipcc(), "-shared", "-fPIC", f"--offload-arch={ARCH}", *objs, "-o", tmp],
                       capture_output=True, text=True)
    for o in objs:
        try:
            os.remove(o)
        except OSError:
            pass
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    os.replace(tmp, LIB)                    # atomic: a concurrent dlopen sees the old or the new file, never a torn one
    with open(STAMP + f".{os.getpid()}", "w") as fh:
        fh.write(source_hash() + "\n")
    os.replace(STAMP + f".{os.getpid()}", STAMP)
    return LIB


if __name__ == "__main__":
    path = build_library(force="--force" in sys.argv, keep_temps="--keep-temps" in sys.argv,
                         verbose="--keep-temps" in sys.argv or "-v" in sys.argv)
    print(path)
