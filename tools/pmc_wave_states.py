"""Where the waves of each kernel spend their cycles (rocprofv3 --pmc passes, SQ block; MI355X_MICROARCH.md "rocprofv3 PMC
slots": SQ_WAIT_ANY = parked at s_waitcnt / barrier, SQ_WAIT_INST_ANY = issue stall, SQ_ACTIVE_INST_ANY = issuing; the three
are disjoint and sum to ~SQ_WAVE_CYCLES; all in quad-cycles).
usage: pmc_wave_states.py <dir of pass 1> [<dir of pass 2> ...]   (each dir = rocprofv3 -d target)"""
import collections
import csv
import glob
import sys

agg = collections.OrderedDict()
for d in sys.argv[1:]:
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "ds::" not in r["Kernel_Name"]:
                continue
            name = r["Kernel_Name"].split("(")[0].replace("void ds::", "").replace("ds::", "")
            a = agg.setdefault(name, collections.defaultdict(float))
            a[r["Counter_Name"]] += float(r["Counter_Value"])
            a["_disp_" + r["Counter_Name"]] += 1
for name, a in agg.items():
    wc = a.get("SQ_WAVE_CYCLES", 0.0)
    if wc <= 0:
        continue
    nd = int(a["_disp_SQ_WAVE_CYCLES"])
    print("%-36s dispatches %3d  wave quad-cycles/dispatch %.3g" % (name, nd, wc / nd))
    parts = []
    for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS",
              "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_MISC", "SQ_WAIT_INST_LDS", "SQ_INST_CYCLES_VMEM",
              "SQ_ACTIVE_INST_FLAT", "SQ_ACTIVE_INST_EXP_GDS"):
        if c in a:
            parts.append("%s %.1f%%" % (c.replace("SQ_", ""), 100.0 * a[c] / wc))
    print("     share of wave cycles: " + "  ".join(parts))
    ins = []
    waves = a.get("SQ_WAVES", 0.0)
    for c in ("SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_INSTS_VALU_MFMA_MOPS_BF16", "SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_INSTS_SMEM",
              "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_VMEM", "SQ_INSTS_FLAT", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE",
              "SQ_BUSY_CYCLES", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_LDS_ADDR_CONFLICT", "SQ_LDS_UNALIGNED_STALL"):
        if c in a:
            ins.append("%s %.4g%s" % (c.replace("SQ_", ""), a[c] / nd, (" (%.0f/wave)" % (a[c] / waves)) if waves and "INSTS" in c else ""))
    if waves:
        ins.insert(0, "WAVES %.0f" % (waves / nd))
    if ins:
        print("     per dispatch: " + "  ".join(ins))
