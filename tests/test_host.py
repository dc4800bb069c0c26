"""CPU-only checks: the C-ABI library loads and exports every symbol the header declares, the
modules keep the reference's checkpoint schema, and the product path refuses CPU tensors."""
import ctypes
import os
import re

import pytest
import torch

import cips_3dplusplus_amd as pkg
from cips_3dplusplus_amd import _lib, configs, weights

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(ROOT, "include", "cips3d_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(cips3d_[a-z0-9_]+)\s*\(", src)))


def test_library_builds_and_exports_header_symbols():
    lib = _lib.load()
    syms = header_symbols()
    assert len(syms) >= 18
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for s in syms:
        assert hasattr(raw, s), f"{s} declared in include/cips3d_hip.h but not exported"
    # every symbol the python binding uses is declared in the header
    assert set(_lib.EXPORTED) <= set(syms)
    m = re.search(r"#define\s+CIPS3D_ABI_VERSION\s+(\d+)", open(os.path.join(ROOT, "include", "cips3d_hip.h")).read())
    assert lib.cips3d_abi_version() == int(m.group(1)) == _lib.ABI_VERSION
    # every struct that crosses the boundary has the same size on both sides (load() already checked; spelled out here)
    for which, st in _lib._struct_table().items():
        assert lib.cips3d_sizeof_struct(which) == ctypes.sizeof(st), st.__name__
    assert lib.cips3d_sizeof_struct(99) == -1
    assert b"bad argument" in lib.cips3d_strerror(-1)


def test_argument_errors_do_not_launch():
    lib = _lib.load()
    assert lib.cips3d_fused_bias_act(None, None, None, None, 16, 1, 1, 3, 0, 0.2, 1.0, None) == -1
    assert lib.cips3d_upfirdn2d(None, None, None, 1, 4, 4, 1, 4, 4, 1, 1, 1, 1, 0, 0, 0, 0, None) == -1
    assert lib.cips3d_nerf_render(None, None) == -1
    assert lib.cips3d_generator_forward(None, None, None) == -1
    assert lib.cips3d_style_phase(None, None, 1, None) == -1
    import ctypes as C
    from cips_3dplusplus_amd import plan as PL
    p, io = PL.GeneratorPlan(), PL.ForwardIO()
    assert lib.cips3d_style_phase(C.byref(p), C.byref(io), 7, None) == -1        # no such mode
    assert lib.cips3d_style_phase(C.byref(p), C.byref(io), 1, None) == -1        # B = 0
    assert lib.cips3d_modconv1x1_supported(512, 512, 4096) == 1
    assert lib.cips3d_modconv1x1_supported(8, 12, 36) == 0
    assert lib.cips3d_nerf_suggest_chunks(1, 64, 24) == 8
    assert lib.cips3d_nerf_suggest_chunks(1, 64, 64) == 8
    assert lib.cips3d_nerf_suggest_chunks(8, 64, 24) == 1


@pytest.mark.parametrize("res,D,nkeys", [(256, 2, 181), (1024, 2, 185), (256, 6, 205), (1024, 8, 221)])
def test_state_dict_schema_counts(res, D, nkeys):
    # SURVEY.md 8(b) "Checkpoint schema": key counts measured on the imported reference
    G = pkg.Generator(**configs.ffhq_G_cfg(res, D))
    sd = G.state_dict()
    assert len(sd) == nkeys
    assert sd["renderer.network.views_linears.weight"].shape == (256, 259)
    assert sd["decoder.conv1.conv.weight"].shape == (1, 512, 256, 1, 1)
    assert sd["decoder.convs.8.conv.blur.kernel"].shape == (4, 4)
    assert sd["style_decoder.1.weight"].shape == (512, 256)
    assert G.decoder.n_latent == 18 and G.decoder.num_layers == 17 and G.z_dim == 256


@pytest.mark.parametrize("tag", ["h32_d2", "h32_d3", "h32_d2_k3"])
def test_state_dict_matches_reference_keys(golden, tag):
    fx = golden("tiny_generator")
    cfg = configs.tiny_G_cfg(32, 3 if "d3" in tag else 2, 3 if "k3" in tag else 1)
    G = pkg.Generator(**cfg)
    ref_keys = [str(k) for k in fx[f"{tag}.keys"]]
    assert list(G.state_dict().keys()) == ref_keys
    ref_sd = fx.sub(f"{tag}.sd.")
    for k, v in G.state_dict().items():
        assert tuple(v.shape) == tuple(ref_sd[k].shape), k
    G.load_state_dict(ref_sd, strict=True)


def test_full_size_golden_weights_reproduce(golden):
    """The full-size fixtures do not ship weights: they must be re-creatable from names + seed."""
    fx = golden("full_size")
    G = pkg.Generator(**configs.ffhq_G_cfg(256, 2))
    sd = weights.synth_state_dict({k: tuple(v.shape) for k, v in G.state_dict().items()}, seed=1)
    assert weights.state_dict_checksum(sd) == int(fx["r256_d2_n24.sd_checksum"])     # exact: closed-form fill, no RNG


def test_product_path_refuses_cpu_tensors():
    """Everything that launches a kernel refuses CPU tensors (no CPU fallback of the path) ..."""
    from cips_3dplusplus_amd import _lib, op
    with pytest.raises(RuntimeError, match="CUDA"):
        _lib.dev_ptr(torch.randn(2, 4), "x")
    with pytest.raises(RuntimeError, match="CUDA"):
        op.bias_act_raw(torch.randn(2, 4), torch.randn(4), None, 3, 0, 0.2, 1.0)
    G = pkg.Generator(**configs.tiny_G_cfg(32, 2, 1))
    with pytest.raises(RuntimeError, match="CUDA|GPU|HIP"):
        G.renderer.render(torch.zeros(1, 3, 4), torch.ones(1, 1, 1), torch.ones(1, 1, 1), torch.ones(1, 1, 1),
                          torch.zeros(1, 3, 32), 4, 2)


def test_op_level_api_dispatches_cpu_tensors_to_torch_like_the_reference(golden):
    """... except the two op-level entry points, whose CPU branch is part of the reference's API
    (/root/reference/exp/op/fused_act.py:105-116, /root/reference/exp/op/upfirdn2d.py:147-150, 160-201): package-own torch ops,
    checked against the reference-generated fixtures (tests/golden/ops.npz) and differentiable through torch's autograd."""
    fx = golden("ops")
    n = 0
    for name in fx["ufd_names"]:
        up, down, p0, p1 = [int(v) for v in fx[f"ufd_{name}_cfg"]]
        y = pkg.upfirdn2d(fx[f"ufd_{name}_x"], fx[f"ufd_{name}_k"], up=up, down=down, pad=(p0, p1))
        assert y.shape == fx[f"ufd_{name}_y"].shape, name
        assert float((y - fx[f"ufd_{name}_y"]).abs().max()) < 1e-5, name
        n += 1
    for name in ("2d_g1", "2d_gs", "4d", "4d_nob", "3d"):
        b = fx[f"flr_{name}_b"] if f"flr_{name}_b" in fx else None
        y = pkg.fused_leaky_relu(fx[f"flr_{name}_x"], b, scale=float(fx[f"flr_{name}_scale"]))
        assert float((y - fx[f"flr_{name}_y"]).abs().max()) < 1e-6, name
        n += 1
    assert n == 14
    x = torch.randn(1, 2, 5, 5, requires_grad=True)
    pkg.upfirdn2d(pkg.fused_leaky_relu(x, torch.zeros(2)), torch.ones(2, 2) / 4, up=2, pad=(1, 0)).sum().backward()
    assert x.grad is not None and bool(torch.isfinite(x.grad).all())


def test_product_never_imports_oracle():
    import sys
    pkg_dir = os.path.dirname(pkg.__file__)
    for fn in os.listdir(pkg_dir):
        if fn.endswith(".py"):
            txt = open(os.path.join(pkg_dir, fn)).read()
            assert "import oracle" not in txt and "from oracle" not in txt, fn


@pytest.mark.parametrize("res,D,B,S0", [(256, 6, 2, 64), (1024, 2, 1, 64)])
def test_decoder_grad_plan_layout(res, D, B, S0, monkeypatch):
    """Host side of the one-call decoder backward (decoder_grad.GradPlan) without a GPU: the python structs have the library's
    sizes, the workspace regions (amax rows | accumulators zeroed per backward | outputs cloned per backward | per-forward
    tensors) are ordered and disjoint, every pointer the plan hands to C lies inside its region, slot counts are powers of two
    and the slot-reduce table covers every slotted accumulator."""
    import ctypes as C
    from cips_3dplusplus_amd import decoder_grad as dg
    from cips_3dplusplus_amd.decoder import Decoder
    from cips_3dplusplus_amd import hip
    # (layout only, nothing is launched: the binding's CUDA-tensor guard is lifted for the table builder)
    monkeypatch.setattr(hip, "dev_ptr", lambda t, name="", optional=False, **kw: None if t is None else t.data_ptr())
    lib = _lib.load()
    assert lib.cips3d_sizeof_grad_plan() == C.sizeof(dg.DecoderGradPlan) and lib.cips3d_sizeof_grad_io() == C.sizeof(dg.DecoderGradIO)
    cfg = configs.ffhq_G_cfg(res, D)
    dec = Decoder(style_dim=cfg["mapping_decoder_cfg"]["style_dim"], **cfg["decoder_cfg"])
    plan = dg.GradPlan(dec, B, S0, S0, "cpu")
    p, ws = plan.plan, plan.ws
    base, end = ws.data_ptr(), ws.data_ptr() + 4 * ws.numel()
    a0, a1 = p.amax_base, p.amax_base + p.amax_bytes
    z0, z1 = p.zero_base, p.zero_base + p.zero_bytes
    o0, o1 = base + 4 * plan.out_range[0], base + 4 * plan.out_range[1]
    assert base <= a0 < a1 <= z0 < z1 <= end and z0 < o0 < z1 <= o1 <= end      # the zeroed outputs open the output block
    n_conv = 0
    for k in range(p.n_layers):
        L, i = p.layers[k], plan.info[k]
        assert (L.kind, L.Cin, L.Cout, L.H, L.W) == (i["kind"], i["Cin"], i["Cout"], i["H"], i["W"])
        assert z0 <= L.d_wm < z1 and base <= L.wm < end
        S = L.slots
        assert S >= 1 and S & (S - 1) == 0 and S <= 16
        if L.kind < 2:
            n_conv += 1
            # the activations' maxima are the forward's (zeroed with the amax block), the gradients' the backward's (zeroed with
            # its accumulators: a second backward over a retained graph must not see the previous one's maxima)
            assert a0 <= L.y_amax < a1 and z0 <= L.g_amax < o0 and (L.kind == 0 or z0 <= L.glo_amax < o0)
            assert z0 <= L.d_bias < z1 and z0 <= L.d_nw_part < z1 and o1 <= L.y < end and o1 <= L.wm_t < end
            assert L.slot_stride >= L.Cout and L.slot_stride % 32 == 0 and S * L.slot_stride <= p.nw_stride
        else:
            assert L.rgb_slot_stride >= B * 3 * L.Cin and L.slots == p.layers[k - 1].slots
            assert o0 <= L.d_bias < z1                                              # ToRGB bias gradient: zeroed, returned
    assert n_conv == plan.n_conv == dec.num_layers
    slotted = sum(1 for k in range(p.n_layers) if p.layers[k].slots > 1)
    assert p.slot_n == slotted and (slotted == 0) == (not p.slot_table)
    assert len(dg.parameters_of(dec)) == 3 * p.n_layers + 2 * n_conv + (p.n_layers - n_conv)
    # at 256^2 and above the row accumulators are slotted (one cache line per slot: L2 atomics serialise per line)
    assert any(p.layers[k].slots > 1 for k in range(p.n_layers))
    # the plan cache is keyed weakly by the decoder and a plan does not keep its decoder alive: dropping the module drops
    # the plan (and its workspace)
    import gc
    import weakref
    n0 = len(dg._PLANS)
    dg._PLANS.setdefault(dec, {})["k"] = plan
    assert len(dg._PLANS) == n0 + 1
    ws_ref = weakref.ref(plan.ws)
    del dec, plan, p, ws
    gc.collect()
    assert len(dg._PLANS) == n0 and ws_ref() is None


def test_chain_kernels_carry_no_high_register_broadcast_of_packed_fp32(tmp_path):
    """profiles/r05_slp_fold_cause.md: `v_pk_fma_f32 ... op_sel:[0,1,0]` (a packed-fp32 instruction whose low lane reads the HIGH
    register of its src1 pair) is what made the bf16 chain kernel return run-to-run different ToRGB partial sums when hipcc's SLP
    vectoriser packed its epilogue.  csrc/chain.hip is built with -fno-slp-vectorize and spells the fold's FMAs in inline asm; this
    test compiles it to assembly with the shipped flags (no GPU needed) and fails if the form shows up in a chain kernel again."""
    import re
    import subprocess
    from cips_3dplusplus_amd import build
    src = os.path.join(build.CSRC, "chain.hip")
    out = tmp_path / "chain.s"
    r = subprocess.run([build.hipcc(), *build.FLAGS, *build.FILE_FLAGS.get("chain.hip", []), "--cuda-device-only", "-S", src, "-o", str(out)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    text = out.read_text()
    assert "-fno-slp-vectorize" in build.FILE_FLAGS.get("chain.hip", []) or os.environ.get("CIPS3D_CHAIN_SLP") == "1"
    cur, bad, n_kernels = None, [], 0
    for line in text.splitlines():
        m = re.match(r"^(_Z\S*chain_gemm_kernel\S*):", line)
        if m:
            cur = m.group(1)
            n_kernels += 1
        elif re.match(r"^_Z\S+:", line):
            cur = None
        if cur and re.search(r"v_pk_[a-z]+_f32 .*op_sel:\[[01],1", line):
            bad.append((cur, line.strip()))
    assert n_kernels >= 3, "chain_gemm_kernel instantiations not found in the assembly"
    assert not bad, bad[:3]
