"""Diagnostic: per basic block of a kernel's gfx950 assembly (hipcc --save-temps), count MFMA / VALU / LDS / VMEM / SALU
instructions. On gfx950 a VALU instruction does not run in the shadow of an fp32 MFMA (tools/mfma_valu.hip), so the VALU
count of an MFMA loop is matrix-pipe time lost.   python tools/isa_blocks.py build/ds_kernels-...gfx950.s lstm_cell_lds_kernelILi1"""
import re, sys
s = open(sys.argv[1]).read()
pat = sys.argv[2]
minm = int(sys.argv[3]) if len(sys.argv) > 3 else 8
for m in re.finditer(r'^(_Z\w+):.*?\n(.*?)s_endpgm', s, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if pat not in name:
        continue
    print(name)
    cur = ["(entry)", 0, 0, 0, 0, 0]
    blocks = [cur]
    for l in body.split("\n"):
        l = l.strip()
        if re.match(r'\.LBB\d+_\d+:', l):
            cur = [l, 0, 0, 0, 0, 0]; blocks.append(cur)
        elif l.startswith("v_mfma"): cur[1] += 1
        elif l.startswith("v_"): cur[2] += 1
        elif l.startswith("ds_"): cur[3] += 1
        elif l.startswith(("global_", "buffer_", "flat_", "scratch_")): cur[4] += 1
        elif l.startswith("s_"): cur[5] += 1
    for b in blocks:
        if b[1] >= minm or (minm == 0 and b[2] > 20):
            print("   %-12s mfma %3d  valu %3d  lds %3d  vmem %3d  salu %3d   valu cycles/mfma cycles ~ %.1f %%" %
                  (b[0], b[1], b[2], b[3], b[4], b[5], 100.0 * b[2] * 4.5 / max(1, b[1] * 64)))
