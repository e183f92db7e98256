"""Lab: run GPU tests against the mask-free-set-up variant of the library (build_variant.py).
usage: python tools_dev/lab/maskfree_setup/pytest_variant.py <pytest args>"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
os.chdir(ROOT)
from occnet_amd import _lib                                         # noqa: E402
import pytest                                                       # noqa: E402

_lib.LIB_PATH = os.path.join(ROOT, "tools_dev", "bin", "libocc_amd_maskfree.so")
rc = pytest.main(sys.argv[1:])
print("variant library loaded:", _lib._lib is not None and _lib.LIB_PATH, flush=True)
sys.exit(rc)
