"""Host-only: per-rank parsing cost of the sharded call_mods reader. For world = 1, 2, 4, 8 every rank parses ONLY its own
byte ranges of the feature file (deepsignal_amd.call_modifications._call_mods_sharded's partition: ranges cut at read
boundaries by ds_tsv_align, range k -> rank k % world); each rank's parse is timed on its own with a fixed thread count,
so the figures are CPU work per rank, not a scaling claim for a box with fewer cores than ranks.
usage: shard_parse_bench.py [rows] [threads per rank] [out.json]"""
import json, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from deepsignal_amd import call_modifications as cm, fastio, synth
from deepsignal_amd.utils.process_utils import code2base_dna

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
threads = int(sys.argv[2]) if len(sys.argv) > 2 else 2
nuniq = min(rows, 2048)
f = synth.synthetic_features(nuniq, seed=5)
tails = ["\t".join(["".join(code2base_dna[int(c)] for c in f["kmer"][i]), ",".join("%.6f" % x for x in f["means"][i]),
                    ",".join("%.6f" % x for x in f["stds"][i]), ",".join(str(int(x)) for x in f["sanums"][i]),
                    ",".join("%.6f" % x for x in f["signals"][i]), "1"]) for i in range(nuniq)]
d = tempfile.mkdtemp(prefix="ds_shard_")
path = os.path.join(d, "features.tsv")
with open(path, "w") as w:
    for i in range(rows):
        w.write("chr1\t%d\t+\t%d\tread_%06d\tt\t%s\n" % (1000 + i, i, i // 20, tails[i % nuniq]))
size = os.path.getsize(path)
chunk = max(1 << 20, size // 64)          # ~64 ranges on this test file (the product uses 32 MB ranges)
out = {"rows": rows, "file_MB": round(size / 1e6, 1), "threads_per_rank": threads, "range_bytes": chunk, "worlds": {}}
for world in (1, 2, 4, 8):
    per_rank = []
    for rank in range(world):
        rd = fastio.FeatureReader(path, nthreads=threads)
        per = max(1, min(4096, rd.size // max(1, world * chunk)))
        cuts = rd.cut_points(world * per)
        t0 = time.perf_counter()
        n = 0
        for c in range(rank, world * per, world):
            rd.set_range(cuts[c], cuts[c + 1])
            for item in rd.items(50):
                n += len(item.labels)
        per_rank.append((time.perf_counter() - t0, n))
        rd.close()
    assert sum(n for _, n in per_rank) == rows
    out["worlds"][str(world)] = {"seconds_per_rank": [round(t, 3) for t, _ in per_rank], "rows_per_rank": [n for _, n in per_rank],
                                 "max_seconds": round(max(t for t, _ in per_rank), 3)}
    print("world %d: slowest rank parses %6d rows in %.3f s (ranks: %s)" % (world, max(n for _, n in per_rank), max(t for t, _ in per_rank),
                                                                          " ".join("%.3f" % t for t, _ in per_rank)))
base = out["worlds"]["1"]["max_seconds"]
for wd, v in out["worlds"].items():
    v["vs_world_1"] = round(v["max_seconds"] / base, 3)
print("slowest-rank parse time vs world 1:", {k: v["vs_world_1"] for k, v in out["worlds"].items()})
if len(sys.argv) > 3:
    json.dump(out, open(sys.argv[3], "w"), indent=1)
import shutil; shutil.rmtree(d, ignore_errors=True)
