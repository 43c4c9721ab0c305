"""Where the waves of each kernel spend their cycles (rocprofv3 --pmc passes, SQ block; MI355X_MICROARCH.md "rocprofv3 PMC
slots": SQ_WAIT_ANY = parked at s_waitcnt / barrier, SQ_WAIT_INST_ANY = issue stall, SQ_ACTIVE_INST_ANY = issuing; the three
are disjoint and sum to ~SQ_WAVE_CYCLES; all in quad-cycles). Every pass must carry SQ_WAVE_CYCLES: a counter is normalised
by the wave cycles of ITS OWN pass.
usage: pmc_wave_states.py <dir of pass 1> [<dir of pass 2> ...]   (each dir = rocprofv3 -d target)"""
import collections
import csv
import glob
import sys

agg = collections.OrderedDict()          # kernel -> counter -> [value sum, dispatches, wave cycles of the same pass]
for d in sys.argv[1:]:
    per = collections.OrderedDict()
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "ds::" not in r["Kernel_Name"]:
                continue
            name = r["Kernel_Name"].split("(")[0].replace("void ds::", "").replace("ds::", "")
            a = per.setdefault(name, collections.defaultdict(lambda: [0.0, 0]))
            a[r["Counter_Name"]][0] += float(r["Counter_Value"])
            a[r["Counter_Name"]][1] += 1
    for name, a in per.items():
        wc = a.get("SQ_WAVE_CYCLES", [0.0, 0])[0]
        out = agg.setdefault(name, {})
        for c, (v, n) in a.items():
            if c not in out:
                out[c] = [v, n, wc]
for name, a in agg.items():
    if "SQ_WAVE_CYCLES" not in a:
        continue
    v, nd, _ = a["SQ_WAVE_CYCLES"]
    print("%-36s dispatches %3d  wave quad-cycles/dispatch %.3g" % (name, nd, v / nd))
    parts = []
    for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS",
              "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_MISC", "SQ_WAIT_INST_LDS", "SQ_INST_CYCLES_VMEM",
              "SQ_ACTIVE_INST_FLAT", "SQ_ACTIVE_INST_EXP_GDS"):
        if c in a and a[c][2] > 0:
            parts.append("%s %.1f%%" % (c.replace("SQ_", ""), 100.0 * a[c][0] / a[c][2]))
    print("     share of wave cycles: " + "  ".join(parts))
    ins = []
    waves = a["SQ_WAVES"][0] if "SQ_WAVES" in a else 0.0
    wdisp = a["SQ_WAVES"][1] if "SQ_WAVES" in a else 0
    for c in ("SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR",
              "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_BUSY_CYCLES", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_LDS_ADDR_CONFLICT"):
        if c in a:
            per_disp = a[c][0] / a[c][1]
            per_wave = (" (%.0f/wave)" % (per_disp / (waves / wdisp))) if waves and "INSTS" in c else ""
            ins.append("%s %.4g%s" % (c.replace("SQ_", ""), per_disp, per_wave))
    if waves:
        ins.insert(0, "WAVES %.0f" % (waves / wdisp))
    if ins:
        print("     per dispatch: " + "  ".join(ins))
