"""Import harness for the reference generator (CPU only, THIS container only).

Used exclusively by tests/golden/make_golden.py to produce golden vectors.
/root/reference never travels to the GPU box, so nothing under tests/ that
runs there may import this module.  Stubs carry no arithmetic: they only
satisfy import-time names (repr helpers, registry decorator, mesh libs).
"""
import sys, types

sys.dont_write_bytecode = True

REF_ROOT = "/root/reference"


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


class _Registry:
    def register(self, *a, **k):
        return lambda cls: cls


class _Anything:
    def __getattr__(self, k):
        return _Anything()

    def __call__(self, *a, **k):
        return _Anything()


def import_reference():
    if "exp.cips3d.models.model_v3" in sys.modules:
        return sys.modules["exp.cips3d.models.model_v3"]
    tu = _mod(
        "tl2.tl2_utils",
        get_class_repr=lambda self, prefix="": f"{prefix}.{type(self).__name__}",
        dict2string=lambda dict_obj, **k: str(dict_obj),
        print_repr=lambda self: None,
    )
    _mod("tl2", tl2_utils=tu)
    _mod("tl2.proj")
    _mod("tl2.proj.fvcore", MODEL_REGISTRY=_Registry())
    _mod("tl2.proj.pytorch", torch_utils=_mod("tl2.proj.pytorch.torch_utils"))
    for n in ["pytorch3d", "pytorch3d.io", "pytorch3d.renderer", "pytorch3d.structures",
              "pytorch3d.transforms", "trimesh", "skimage", "skimage.measure"]:
        _mod(n).__getattr__ = lambda k: _Anything()
    import torch.utils.cpp_extension as ce
    ce.load = lambda *a, **k: _Anything()  # no nvcc here: op/*.py take their own CPU branches
    sys.path[:0] = [REF_ROOT, REF_ROOT + "/exp"]
    import exp.cips3d.models.model_v3 as m
    return m
