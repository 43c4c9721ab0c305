# what the cache hierarchy reports for the operand streams: per-kernel L1->L2 read requests, their summed latency, L2 hits and misses
cd /tmp 2>/dev/null; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/l2lat; mkdir -p $O
rocprofv3 -L 2>/dev/null | grep -o -E "\b(TCP|TCC)_[A-Z0-9_]+" | sort -u > $O/counters.txt; wc -l $O/counters.txt
for prec in fp32 bf16x3; do
  rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum --output-format csv -d $O/a_$prec -- python3 tools/probe_engine.py $prec 512 threestep > /dev/null 2> $O/a_$prec.err
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum --output-format csv -d $O/b_$prec -- python3 tools/probe_engine.py $prec 512 threestep > /dev/null 2> $O/b_$prec.err
  rocprofv3 --pmc TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN2_sum TCP_TA_TCP_STATE_READ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum --output-format csv -d $O/c_$prec -- python3 tools/probe_engine.py $prec 512 threestep > /dev/null 2> $O/c_$prec.err
done
python3 - <<'PY'
import csv, glob, collections
for prec in ("fp32", "bf16x3"):
    agg = collections.OrderedDict()
    for p in "abc":
        fs = glob.glob("gpurun_out/l2lat/%s_%s/**/*counter_collection.csv" % (p, prec), recursive=True)
        if not fs: print("no file", p, prec); continue
        for r in csv.DictReader(open(fs[0])):
            if "ds::" not in r["Kernel_Name"]: continue
            n = r["Kernel_Name"].split("(")[0].replace("void ds::", "")[:40]
            a = agg.setdefault(n, collections.Counter()); a[r["Counter_Name"]] += float(r["Counter_Value"]); a["n_" + r["Counter_Name"]] += 1
    for n, a in agg.items():
        g = lambda c: a[c] / max(a["n_" + c], 1)
        rr = g("TCP_TCC_READ_REQ_sum")
        print("%-7s %-40s read_req %.3g lat/req %.0f clk | L2 hit %.3g miss %.3g ea_rd %.3g req %.3g | pend_stall %.3g gate2 %.3g tagconf %.3g" % (
            prec, n, rr, g("TCP_TCC_READ_REQ_LATENCY_sum") / max(rr, 1), g("TCC_HIT_sum"), g("TCC_MISS_sum"), g("TCC_EA0_RDREQ_sum"), g("TCC_REQ_sum"),
            g("TCP_PENDING_STALL_CYCLES_sum"), g("TCP_GATE_EN2_sum"), g("TCP_READ_TAGCONFLICT_STALL_CYCLES_sum")))
PY
tail -3 $O/a_fp32.err
