import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# The library only takes the incremental re-score path where its cost model says it beats a full
# evaluation, which on test-sized corpora is never: the tests ask for the plain rule "a token in at
# most a third of the chunks" so that the path is exercised (tests/test_gpu_toggles.py has one case
# on the default).
os.environ.setdefault("TOKSCORE_INCR_DIV", "3")
