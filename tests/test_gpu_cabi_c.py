"""The C ABI from plain C on the GPU (`tools/cabi_spmm_bench.c`, built by `__graft_entry__.build()` with gcc):
graph build + normalized SpMM, checked inside the program against a host loop over the edge list."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("args", [["20000", "80000", "32"], ["3000", "9000", "8"], ["70000", "150000", "64"]])
def test_c_program_builds_a_plan_and_runs_the_spmm(dev, args):
    import __graft_entry__ as ge
    exe = ge.build_c_consumer()
    out = subprocess.run([exe] + args, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "gconv_spmm_csr_f32, plain call:" in out.stdout and "pre-scaled operands" in out.stdout and "GB/s" in out.stdout
