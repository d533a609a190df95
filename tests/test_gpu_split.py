"""The opt-in fp16x3 GEMM mode (SVAE_GEMM=fp16x3: hidden-layer GEMMs on the f16 matrix pipe with split operands,
spatial_vae_amd/csrc/split.h) must pass the SAME parity tests as the fp32-MFMA path: goldens from the reference, the
BASELINE-size cases, the odd geometries.  The mode is read once per process, hence the subprocess."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra_env, files):
    env = dict(os.environ, SVAE_GEMM="fp16x3", **extra_env)
    out = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider"] + files, cwd=ROOT,
                         env=env, capture_output=True, text=True, timeout=900)
    tail = out.stdout[-1500:] + out.stderr[-1500:]
    assert out.returncode == 0, tail
    assert " passed" in out.stdout and "failed" not in out.stdout, tail


def test_parity_suites_pass_in_fp16x3_mode():
    _run({}, ["tests/test_gpu_parity.py", "tests/test_gpu_fullsize.py", "tests/test_gpu_geometry.py"])


def test_parity_with_conversion_passes_instead_of_fused_producers():
    """SVAE_SPLIT_L0=0 / SVAE_SPLIT_OB=0: fp32 planes + conversion kernels feed the same GEMMs (deeper stacks use this)."""
    _run({"SVAE_SPLIT_L0": "0", "SVAE_SPLIT_OB": "0"}, ["tests/test_gpu_parity.py"])


def test_parity_with_fragments_only_a0():
    """SVAE_SPLIT_A0=1: the coordinate layer writes no fp32 plane, the fused first-layer epilogue reads act' from the column
    fragments (a memory option)."""
    _run({"SVAE_SPLIT_A0": "1"}, ["tests/test_gpu_parity.py"])
