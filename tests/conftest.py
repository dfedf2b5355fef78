import os
import sys

import pytest

# the product sets this at the first Net when HIP is not up yet (myolo.engine._ensure_hw_queues); the GPU tests touch the device through bare
# operator calls before any Net exists, so the harness sets it for the whole session (read by the HIP runtime when it initialises)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "mask-yolo_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built_library():
    """The C-ABI library is a build product (git-ignored).  If it is missing and hipcc is available
    (it cross-compiles gfx950 without a GPU), build it once per session, exactly as __graft_entry__.build() does."""
    lib = os.path.join(ROOT, "mask-yolo_amd", "myolo", "_lib", "libmyolo_hip.so")
    if not os.path.exists(lib) and os.path.exists(os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")):
        import __graft_entry__
        __graft_entry__.build()
    # run the whole suite against an alternative kernel choice:  MYOLO_TEST_OPTIONS=wino_x6=1 python -m pytest tests -m gpu
    opts = os.environ.get("MYOLO_TEST_OPTIONS", "")
    if opts:
        from myolo import _ext as X
        for kv in opts.split(","):
            name, _, val = kv.partition("=")
            if name.strip() == "wino_tiles":       # a config default, not a library switch:  MYOLO_TEST_OPTIONS=wino_tiles=f63
                from myolo import config as mcfg
                mcfg.Config.WINOGRAD_TILES = val.strip()
                continue
            X.set_option(name.strip(), int(val))
            if name.strip() == "wino_x6":          # a Net sets this switch from its config: make it the default there as well
                from myolo import config as mcfg
                mcfg.Config.FP32_MATMUL = "bf16x6" if int(val) else "native"
    yield
