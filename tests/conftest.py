import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run via gpurun)")


def pytest_sessionstart(session):
    """On a GPU box, before any test: is the BOX healthy?  (tests/box_canary.py: pure torch in a child
    process, retried.)  The verdict goes to stderr so that it is in the driver's log whatever happens next;
    an unhealthy box does not stop the run -- the tests then fail on their own, with this line above them."""
    if os.environ.get("TIMG_SKIP_CANARY") or not os.path.exists("/dev/kfd"):
        return
    import box_canary
    ok, text = box_canary.run_canary()
    sys.__stderr__.write(("[box canary] GPU BOX OK: " if ok else
                          "[box canary] GPU BOX UNHEALTHY (pure torch, none of this repository's code): ") + text + "\n")
    sys.__stderr__.flush()


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Make sure the checker (oracle/) and the product library exist.  Building
    is cheap and idempotent; on the GPU box the prebuilt files travel along."""
    need_oracle = not os.path.exists(os.path.join(ROOT, "oracle", "libtimg_oracle.so"))
    need_lib = not os.path.exists(os.path.join(ROOT, "timg_amd", "libtimg_hip.so"))
    if need_oracle or need_lib:
        subprocess.check_call([sys.executable, "-c", "import __graft_entry__ as g; g.build()"],
                              cwd=ROOT)


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    return oracle_lib.Oracle()


@pytest.fixture(scope="session")
def ref():
    import oracle_lib
    r = oracle_lib.Ref.try_load()
    if r is None:
        pytest.skip("oracle/_ref/libtimg_ref.so not built (no /root/reference here)")
    return r


@pytest.fixture(scope="session")
def hip():
    import timg_amd
    h = timg_amd.TimgHip(0)  # raises loudly without a GPU / without the .so
    yield h
    h.close()
