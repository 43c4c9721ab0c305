"""GPU: the cross-workgroup hand-off protocol of tools/handoff_repro.hip stays clean on the box the tests run on:
`sc1` write-through stores, every storing wave drains, one flag store; the consumer polls, ONE agent-scope acquire, plain LDS-DMA (or plain
loads), every word compared -- 40 configurations (store form, load form, stale-L1 pre-read, same / next XCD, ping-pong, with and without MFMA
kernels on three more streams), a short run of each. The long run is profiles/r06_handoff_repro.txt (1.2e12 checks, 0 wrong)."""
import os
import shutil
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_handoff_protocol_delivers_every_fragment(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    exe = str(tmp_path / "handoff_repro")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O2", os.path.join(ROOT, "tools", "handoff_repro.hip"), "-o", exe], check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    res = subprocess.run([exe, "0.05"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    out = res.stdout.decode()
    assert res.returncode == 0, out[-2000:]          # 1 = wrong words, 3 = a bounded spin timed out
    total = [l for l in out.splitlines() if l.startswith("TOTAL")]
    assert len(total) == 1 and total[0].endswith(" 0 wrong"), out[-2000:]
    assert out.count(" 0 wrong,") == 40, out[-3000:]       # every configuration ran and is clean
