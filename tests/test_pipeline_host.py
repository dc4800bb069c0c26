"""Host logic of pipeline.ViewPipeline that needs no device: on a CPU generator there are no streams and every call runs inline on
the caller (the two-stream behaviour itself is a GPU test: tests/test_io_and_bulk.py)."""
import gc
import weakref

import pytest
import torch

from cips_3dplusplus_amd.pipeline import ViewPipeline, pipeline_for


class _G(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.lin = torch.nn.Linear(2, 2)
        self.calls = []

    def forward(self, **kw):
        self.calls.append(kw)
        return {"rgb": torch.zeros(1) + len(self.calls)}


def test_cpu_pipeline_runs_inline_and_in_order():
    G = _G()
    pipe = ViewPipeline(G, lanes=2)
    assert pipe.lanes == 1 and pipe.streams == []          # no HIP device: nothing to alternate between
    outs = [pipe.submit(a=i) for i in range(3)]
    assert [int(o["rgb"]) for o in outs] == [1, 2, 3] and [c["a"] for c in G.calls] == [0, 1, 2]
    assert pipe.run(lambda: 7) == 7
    pipe.wait_lane(0); pipe.drain()                        # no-ops without streams


def test_lane_count_is_validated():
    with pytest.raises(ValueError):
        ViewPipeline(_G(), lanes=0)


def test_pipelines_are_cached_per_generator_and_do_not_pin_it():
    G = _G()
    p2 = pipeline_for(G, 2)
    assert pipeline_for(G, 2) is p2 and pipeline_for(G, 3) is not p2
    assert p2.G is G
    ref = weakref.ref(G)
    del G, p2
    gc.collect()
    assert ref() is None                                   # the cache is weakly keyed and the pipeline holds a weak reference
