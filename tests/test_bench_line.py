"""The line bench.py prints for the driver must stay small: round 4's grew to 36 KB and the driver's record came back with
`parsed: null`.  These CPU tests hold `bench.compact` to the limit on the largest record the bench has ever produced (the
committed round-4 driver-form line) and on a synthetic worst case, and check that what the contract names survives."""
import copy
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline", "cpu_baseline")


def _r04_record():
    with open(os.path.join(ROOT, "profiles", "r04_bench_driver_form.json")) as fh:
        text = fh.read()
    assert len(text) > 30000                      # the line the driver could not parse
    return json.loads(text)


def test_compact_line_of_the_round4_record_is_small_and_complete():
    rec = _r04_record()
    text = bench.compact(rec, "bench_detail.json")
    assert len(text) < bench.LINE_LIMIT <= 6144 and "\n" not in text
    line = json.loads(text)
    for k in CONTRACT:
        assert k in line, k
    assert line["value"] == float(f"{rec['value']:.6g}") and line["unit"] == "views/s" and line["n_gpus"] == 1
    assert len(line["dtype"]) <= 80 and "workload" in line["config"]
    rf = line["roofline"]
    assert rf["bound"] == "mfma" and abs(rf["frac"] - rec["roofline"]["frac"]) < 1e-3 * rec["roofline"]["frac"]
    assert set(rf) >= {"kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms", "kernels"}
    kinds = {(k["kind"], k["c_in"]) for k in rf["kernels"]}
    assert ("render", None) in kinds and {("fused_stage", c) for c in (32, 64, 128, 256)} <= kinds
    for k in rf["kernels"]:
        assert set(k) == {"kind", "c_in", "res", "n", "us", "bound", "frac", "traffic"}
    assert set(line["cpu_baseline"]) >= {"value", "unit", "cores", "kind"}
    assert len(line["also"]) == len(rec["also"]) and "truncated" not in line
    for e in line["also"]:
        assert e["value"] > 0 and e["ms_per_step"] > 0 and len(e["workload"]) <= 48


def test_compact_line_never_exceeds_the_limit():
    """A synthetic worst case: forty `also` entries with long prose, errors, a huge kernel table, absurd strings."""
    rec = _r04_record()
    big = copy.deepcopy(rec)
    big["also"] = (rec["also"] * 4)[:40]
    big["also"][3] = {"what": "x" * 500, "error": "RuntimeError: " + "y" * 1000}
    big["roofline"]["kernels"] = rec["roofline"]["kernels"] * 6
    big["config"]["workload"] = "w" * 4000
    big["cpu_baseline"]["sample"] = "s" * 5000
    big["ms_per_step_repeats"] = [0.123456789] * 500
    text = bench.compact(big, "bench_detail.json")
    assert len(text) < bench.LINE_LIMIT
    line = json.loads(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "dtype", "config", "roofline"):
        assert k in line
    assert line["roofline"]["frac"] > 0


def test_compact_line_of_a_multi_rank_record_keeps_the_rank_fields():
    rec = {"metric": "m", "value": 1234.5678, "unit": "views/s", "n_gpus": 8, "steps": 20, "warmup": 5, "ms_per_step": 6.4,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic", "repeats": 5,
           "ms_per_step_repeats": [6.4] * 5, "config": {"workload": "w", "parallelism": "views x8"}, "roofline": None,
           "rccl_ranks": 8, "ranks": 8, "dist_backend": "nccl (RCCL)", "physical_gpus": 8, "nonfinite": float("nan")}
    line = json.loads(bench.compact(bench.finite(rec), None))
    assert line["rccl_ranks"] == 8 and line["ranks"] == 8 and line["physical_gpus"] == 8 and line["roofline"] is None
    assert line["config"]["parallelism"] == "views x8" and "cpu_baseline" not in line and "also" not in line


def test_emit_writes_the_detail_file_and_prints_one_line(tmp_path, capsys):
    rec = _r04_record()
    p = tmp_path / "detail.json"
    text = bench.emit(rec, str(p))
    out = capsys.readouterr().out
    assert out.count("\n") == 1 and out.strip() == text
    full = json.load(open(p))
    assert full["roofline"]["kernels"][1]["launch_ms_bounds"] == rec["roofline"]["kernels"][1]["launch_ms_bounds"]
    assert len(full["also"]) == len(rec["also"])


def test_replayed_traffic_stays_with_a_kernels_dominant_shape():
    rows = [{"kind": "render", "kernel": "nerf_render_kernel<16,4>", "launches_per_step": 1, "traffic": 9.6e6},
            {"kind": "planes_gemm", "kernel": "chain_gemm_kernel", "launches_per_step": 8, "traffic": 24.2e6, "c_in": 512},
            {"kind": "planes_gemm", "kernel": "chain_gemm_kernel", "launches_per_step": 1, "traffic": 24.2e6, "c_in": 256},
            {"kind": "lowres_gemm", "kernel": "modconv1x1_kernel", "launches_per_step": 1, "traffic": 24.6e6},
            {"kind": "fused_stage", "kernel": "fused_up_conv_kernel<32, ...>", "launches_per_step": 1, "traffic": 58.9e6},
            {"kind": "fused_stage", "kernel": "fused_up_conv_kernel<64, ...>", "launches_per_step": 1, "traffic": None},
            {"kind": "torgb", "kernel": "torgb", "launches_per_step": 1, "traffic": None}]
    out = bench.dedupe_traffic([dict(r) for r in rows])
    assert [r["traffic"] for r in out] == [9.6e6, 24.2e6, None, 24.6e6, 58.9e6, None, None]
    assert out[2]["traffic_source"].startswith("no counter pass")
