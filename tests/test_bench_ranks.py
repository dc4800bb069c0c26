"""bench.py's N > 1 leg: `--gpus N` starts N ranks itself (the launch form of the reference's multi-process scripts,
/root/reference/exp/tests/test_cips3dpp.py:814-820, scripts/gen_images.py:44-84).  CPU tests cover the launcher and
its loud failures; the GPU test runs the bench step with 2 ranks on the box's device(s) and checks the frames gathered on
rank 0 against two single-rank renders."""
import json
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(kw)
    return env


def test_gpus_flag_must_match_world_size():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "3"], env=_env(WORLD_SIZE="1", RANK="0"), capture_output=True,
                       text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr


@pytest.mark.skipif(torch.cuda.is_available(), reason="covered by the GPU test below")
def test_gpus_flag_spawns_ranks_that_fail_loudly_without_a_gpu():
    """No GPU here: the launcher must still start 2 ranks, each of which refuses to run (no CPU fallback)."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0"], env=_env(),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode != 0
    assert r.stderr.count("bench.py needs a GPU") >= 1, r.stderr[-2000:]     # the launcher may stop rank 1 before it prints
    assert "for 2 ranks" in r.stderr
    assert '"n_gpus"' not in r.stdout


@pytest.mark.gpu
def test_two_rank_bench_step_gathers_the_single_rank_frames(tmp_path):
    dump = str(tmp_path / "frames.pt")
    env = _env()
    if torch.cuda.device_count() < 2:
        env["CIPS3D_DIST_BACKEND"] = "gloo"           # one device: the two ranks share it, the exchange runs over gloo
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "3", "--warmup", "1", "--repeats", "2",
                        "--deterministic", "--no-cpu-baseline", "--dump-gathered", dump], env=env, capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and r.stdout.rstrip("\n").splitlines()[-1] == lines[0], r.stdout
    line = json.loads(lines[0])
    rccl = 2 if torch.cuda.device_count() >= 2 else 0      # `rccl_ranks` counts ranks that exchanged over RCCL: none over gloo
    assert line["n_gpus"] == 2 and line["ranks"] == 2 and line["rccl_ranks"] == rccl and line["scaling"] == "weak"
    assert line["config"]["parallelism"] == "views x2" and line["value"] > 0
    assert len(line["ms_per_step_repeats"]) == 2
    frames = torch.load(dump)
    assert frames.shape == (2, 3, 1024, 1024) and frames.dtype == torch.uint8
    sys.path.insert(0, ROOT)
    import bench
    from cips_3dplusplus_amd import hip
    dev = torch.device("cuda", 0)
    for rank in range(2):
        wl = bench.ForwardWorkload(dev, rank, 1, 1024, 2, 24, 1, "fp32", True)
        with torch.no_grad():
            exp = hip.rgb_to_uint8(wl.render()).cpu()
        assert torch.equal(frames[rank:rank + 1], exp), f"rank {rank}'s gathered frame differs from its single-rank render"
        del wl
    assert not torch.equal(frames[0], frames[1])


@pytest.mark.gpu
def test_the_drivers_eight_rank_command_runs_end_to_end():
    """The exact command form the driver uses for the 8-GPU leg -- `python -m torch.distributed.run --nnodes=1
    --nproc-per-node 8 --master-addr 127.0.0.1 --master-port P bench.py --gpus 8 --steps K --warmup W` -- on this box: with
    fewer than eight devices the ranks share them and the exchange runs over gloo (the line says so), so every rank's render,
    the uint8 conversion, the asynchronous gather to rank 0, the barrier / max-over-ranks timing and the one JSON line have
    run once in the eight-rank shape.  No rank may have initialised the GPU before binding its device (bench.py asserts)."""
    env = _env(MASTER_PORT="")
    env.pop("MASTER_PORT")
    n_dev = torch.cuda.device_count()
    if n_dev < 8:
        env["CIPS3D_DIST_BACKEND"] = "gloo"
    port = str(29800 + os.getpid() % 1000)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
           "--master-port", port, BENCH, "--gpus", "8", "--steps", "2", "--warmup", "1", "--repeats", "1"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 8 and line["ranks"] == 8 and line["scaling"] == "weak" and line["steps"] == 2
    assert line["rccl_ranks"] == (8 if n_dev >= 8 else 0)          # a gloo run must not claim RCCL ranks
    assert line["physical_gpus"] == n_dev
    assert line["dist_backend"].startswith("nccl" if n_dev >= 8 else "gloo")
    assert line["config"]["parallelism"] == "views x8" and line["value"] > 0 and line["ms_per_step"] > 0
    assert line["roofline"] is None or line["roofline"]["frac"] > 0
    assert "also" not in line and "cpu_baseline" not in line            # rank-0-at-N=1-only legs


@pytest.mark.gpu
def test_single_gpu_line_prices_every_big_kernel(tmp_path):
    """The N = 1 line as the driver reads it: the LAST stdout line, one compact JSON object under bench.LINE_LIMIT bytes, whose
    `roofline.kernels` prices the render kernel, the 64^2 chain GEMM and the four fused up-sampling stages -- each with an in-run
    launch time and a fraction of its bound in (0, 1), launch times adding up to no more than the step they are part of -- and
    the full record (bounds, flop / bytes per launch, provenance) in the detail file beside it."""
    sys.path.insert(0, ROOT)
    import bench
    detail = str(tmp_path / "detail.json")
    r = subprocess.run([sys.executable, BENCH, "--steps", "10", "--warmup", "3", "--repeats", "2", "--no-cpu-baseline", "--no-also",
                        "--detail", detail], env=_env(), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    last = r.stdout.rstrip("\n").splitlines()[-1]
    assert len(last) < bench.LINE_LIMIT, len(last)
    line = json.loads(last)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline"):
        assert k in line, k
    assert len(line["dtype"]) <= 80 and line["steps"] == 10 and line["value"] > 0
    rows = line["roofline"]["kernels"]
    assert rows[0]["kind"] == "render" and rows[0]["frac"] == line["roofline"]["frac"]
    kinds = [(k["kind"], k.get("c_in")) for k in rows]
    assert ("planes_gemm", 512) in kinds and {("fused_stage", c) for c in (32, 64, 128, 256)} <= set(kinds)
    total = 0.0
    for k in rows:
        assert k["us"] > 0 and k["n"] >= 1
        total += k["us"] * 1e-3 * k["n"]
        if k["kind"] in ("render", "planes_gemm", "lowres_gemm", "fused_stage"):
            assert 0.0 < k["frac"] < 1.0 and k["bound"] in ("mfma", "hbm")
            assert "traffic" in k               # measured bytes from a summary of THIS library build, or null
    # (the launch times are those of the kernels ALONE on the device: they add up to the one-stream step, not to the step of two
    # views in flight)
    step_ms = line["single_stream"]["ms_per_step"] if line.get("single_stream") else line["ms_per_step"]
    assert 0.6 * step_ms < total < 1.05 * step_ms
    # a replayed counter is keyed on the kernel's dominant shape: two GEMM rows of one kernel never carry the same bytes
    gemm = [k["traffic"] for k in rows if k["kind"] in ("planes_gemm", "lowres_gemm") and k["traffic"] is not None]
    assert len(gemm) == len(set(gemm))
    full = json.load(open(detail))
    assert full["value"] == pytest.approx(line["value"], rel=1e-4)
    for k in full["roofline"]["kernels"]:
        if k["kind"] in ("render", "planes_gemm", "lowres_gemm", "fused_stage"):
            assert k.get("flop_per_launch", 0) > 0 or k.get("algorithmic_bytes", 0) > 0
            assert k["achieved"] > 0 and k["avg_launch_ms"] > 0
