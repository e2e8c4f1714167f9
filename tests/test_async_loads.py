"""Static check of the built ISA: the fp32 GEMM's inline-asm tile loads are never used before their wait.

k_bt_gemm_x6 and k_bt_dw_x6 (csrc/engine_batch.hip.inc) issue their tile loads through inline asm so that two tiles can stay in
flight; the compiler does not know those registers are pending, so a copy or a spill placed between a load and its
`s_waitcnt` would read stale data (seen once: a wait in each arm of an if / else, and a spill in the dual-product
variant).  scripts/check_async_loads.py walks the control-flow graph of the device assembly and reports any such use.
"""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")), reason="needs hipcc")
def test_no_use_of_a_tile_register_before_its_wait():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "check_async_loads.py")],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("0 early uses") == 14, r.stdout   # six k_bt_gemm_x6 variants, four k_bt_gemm_x6_pair, two k_bt_dw_many, two k_bt_gemm_bf16_big
