import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """The shared library is a build product (git-ignored): make sure it exists and is newer than its sources before any test
    loads it.  A no-op when it is up to date; ~100 s of hipcc otherwise (cross-compiles without a GPU)."""
    try:
        import spatial_vae_amd
        spatial_vae_amd.build()
    except Exception as e:  # no hipcc here: the tests that need the library will say so themselves
        sys.stderr.write("conftest: could not (re)build libsvae_hip.so: %s\n" % e)
