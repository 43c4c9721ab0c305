"""In-kernel time stamps of the fp32 BiLSTM cell launches (Engine(debug_stamps=True)): where a diagonal's time goes.
usage: lstm_stamps.py [variant] [batch]"""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from deepsignal_amd import synth, weights as W
from deepsignal_amd.engine import Engine
var = sys.argv[1] if len(sys.argv) > 1 else "lds1"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 512
prec = sys.argv[3] if len(sys.argv) > 3 else "fp32"      # bf16_all: the bf16-operand cells (variant is ignored)
e = Engine(max_batch=B, slots=1, serial=True, debug_stamps=True, lstm_tiling=var, precision=prec); e.load_weights(W.random_weights(seed=1))
e.set_graph(False)
f = synth.synthetic_features(B, seed=2)
args = [f[k] for k in ("kmer", "means", "stds", "sanums", "signals")]
for _ in range(3): e.run(*args)
for d in range(19): e.intermediate("lstm_stamps%d" % d, (13,))      # clear
e.run(*args)
print("variant", var, "batch", B)
print("diag  wgs | entry->loop mean/max | loop mean/max | exit mean/max (cycles) | span us | entry spread us | wg life us | CUs | max wg/CU | clk GHz")
for d in range(19):
    s = e.intermediate("lstm_stamps%d" % d, (13,))
    print("%4d %4d | %8.0f %8.0f | %8.0f %8.0f | %7.0f %7.0f | %6.2f | %5.2f | %6.2f | %3d | %2d | %.2f" %
          (d, s[0], s[1], s[2], s[3], s[4], s[5], s[6], s[7], s[8], s[9], s[10], s[11], s[12]))
e.close()
