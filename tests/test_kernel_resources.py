"""CPU (-m "not gpu"): register / scratch budget of the two kernels whose footprints let them share a CU (DESIGN.md 4,
"Sharing a CU"): two fused-module waves (<= 184 VGPRs each) and one BiLSTM cell wave (<= 96) per SIMD. A change that
pushes either over its budget costs 3 - 4 % of the 512-site throughput without failing any numerical test."""
import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cu_sharing_register_budgets():
    if not shutil.which("/opt/rocm/bin/hipcc"):
        pytest.skip("hipcc not available")
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import kernel_resources
    res = kernel_resources.kernel_resources()
    fused = [r for n, r in res.items() if "inception_fused_kernel<3>" in n]
    cell = [r for n, r in res.items() if "lstm_cell_lds_kernel<1>" in n]
    assert len(fused) == 1 and len(cell) == 1, sorted(res)
    assert fused[0]["vgprs"] <= 184 and fused[0]["scratch_bytes"] == 0, fused[0]
    assert cell[0]["vgprs"] <= 96 and cell[0]["scratch_bytes"] == 0 and cell[0]["static_lds_bytes"] == 0, cell[0]
    # round 3: the bf16 fused chain and the bf16 conv_layer2/3 kernel run TWO 512-thread workgroups per CU (4 waves per SIMD:
    # <= 128 VGPRs), the 128 x 128 bf16 BiLSTM tile three 256-thread workgroups (<= 168)
    for name, cap in (("inception_fused_bf16_kernel<3>", 128), ("inception_fused_bf16_kernel<2>", 128), ("stem23_bf16_kernel", 128),
                      ("lstm_cell_bf16_kernel<2, 2>", 168), ("lstm_cell_bf16_kernel<1, 1>", 96)):
        hit = [r for n, r in res.items() if name in n]
        assert len(hit) == 1 and hit[0]["vgprs"] <= cap, (name, hit)
    # round 5: the split-operand fused chain runs ONE 512-thread workgroup per CU (2 waves per SIMD: <= 256 registers)
    for tm in (1, 2, 3):
        hit = [r for n, r in res.items() if "inception_fused_split_kernel<%d>" % tm in n]
        assert len(hit) == 1 and hit[0]["vgprs"] <= 256, (tm, hit)
    # the split ring kernels count their LDS-DMA requests with a constant s_waitcnt vmcnt: a spill inside the K loop is a vector-memory
    # operation on the same counter and would break the count (the 256 x 192 dense tile: accumulators in the AGPR half)
    wide = [r for n, r in res.items() if "dense_split_kernel<4, 3, 2, 2>" in n]
    assert len(wide) == 1 and wide[0]["vgprs"] <= 512 and wide[0]["scratch_bytes"] == 0, wide
    # (lstm_cell_split_kernel<1, 2>: Engine(lstm_tiling="lds1"), the tile whose waves request unequal fragment counts and use the pad)
    # (lstm_cell_split_kernel<1, 2, 4, 2>: the 128 x 128 tile by eight waves, two per SIMD: <= 256 registers each)
    for name, cap in (("dense_split_kernel<1, 3, 4, 1>", 256), ("lstm_cell_split_kernel<2, 2, 2, 2>", 256), ("lstm_cell_split_kernel<1, 2, 2, 2>", 256),
                      ("lstm_cell_split_kernel<1, 2, 4, 2>", 256), ("lstm_cell_split_kernel<1, 1, 2, 2>", 128), ("dense_split_kernel<4, 3, 2, 2>", 512)):
        hit = [r for n, r in res.items() if name in n]
        assert len(hit) == 1 and hit[0]["vgprs"] <= cap and hit[0]["scratch_bytes"] == 0, (name, hit)
        # their pinned request sequences overwrite m0 between ONE save and ONE restore (SplitRing::request3): sound only while hipcc
        # itself has no use for m0 in between: no spill of any kind in these kernels
        assert hit[0]["vgpr_spills"] == 0 and hit[0]["scratch_bytes"] == 0 and hit[0]["sgpr_spills"] == 0, (name, hit)
    # no kernel of the library may spill
    assert all(r["scratch_bytes"] == 0 for r in res.values()), {n: r for n, r in res.items() if r["scratch_bytes"]}
