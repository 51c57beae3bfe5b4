"""Round 4: the hand-scheduled stage loops come in two forms each (barrier at the end of a stage / in its middle with prefetched
fragments): conv_x6w_kernel (SG_X6W_VAR), pw_wide_kernel<3,float> and <1,bf16> (SG_PW_VAR), wgrad_pw_wide_kernel (SG_WPW_VAR).  Both forms
issue the same MFMAs in the same order, so every output must agree bit for bit.  The switches are read once per process: each
form runs scripts/pw_var_check.py in a child process (started before this process needs anything from it; the children use the
GPU one after the other) and the digests are compared line by line."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _digests(var: str):
    env = dict(os.environ, SG_PW_WIDE="2", SG_PW_VAR=var, SG_WPW_VAR=var, SG_X6W_VAR=var)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "pw_var_check.py")], env=env, cwd=ROOT, capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    return [ln for ln in out.stdout.splitlines() if "->" in ln]


def test_both_barrier_placements_give_the_same_bits():
    end, mid = _digests("0"), _digests("1")
    assert len(end) == 22 and len(mid) == 22, (len(end), len(mid))
    assert end == mid, [(a, b) for a, b in zip(end, mid) if a != b]


def _dw_digests(var: str):
    env = dict(os.environ, SG_DW_FSTRIP=var)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "dw_var_check.py")], env=env, cwd=ROOT, capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    return [ln for ln in out.stdout.splitlines() if ": " in ln and "x" in ln]


def test_depthwise_stencil_as_runs_and_as_strips_gives_the_same_bits():
    """dw_s1_run_kernel (SG_DW_FSTRIP=0) and dw_strip_kernel (2: every map) add the same products in the same order: forward
    (plain, pre-activation ReLU, BatchNormalization in the gather with / without ReLU) and dgrad (plain, ReLU mask, a collected
    gradient riding along) agree bit for bit in fp32 and bf16 storage over six map sizes."""
    runs, strips = _dw_digests("0"), _dw_digests("2")
    assert len(runs) == 12 and len(strips) == 12, (len(runs), len(strips))
    assert runs == strips, [(a, b) for a, b in zip(runs, strips) if a != b]


def _bn_digests(wide: str):
    env = dict(os.environ, SG_PW_WIDE=wide)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "pw_bn_check.py")], env=env, cwd=ROOT, capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    return [ln for ln in out.stdout.splitlines() if "->" in ln]


def test_256_wide_pointwise_tiles_give_the_bits_of_the_128_wide_kernels():
    """pw_wide_kernel<.., BN = 256> (1024 / 2048 / 256 output columns at many rows; SG_PW_WIDE=1) against conv_x6_kernel /
    conv_b16_kernel on the same layers (SG_PW_WIDE=3: 384-wide tiles only): forward + statistics + dgrad, fp32 and bf16 storage -
    the same products in the same order, so the same bits, whichever kernel a layer's batch size selects."""
    narrow, wide = _bn_digests("3"), _bn_digests("1")
    assert len(narrow) == 10 and len(wide) == 10, (len(narrow), len(wide))
    assert narrow == wide, [(a, b) for a, b in zip(narrow, wide) if a != b]
