"""VGPRs, scratch and static LDS of every kernel in deepsignal_amd/csrc/*.hip (hipcc -S, device only, ~40 s; DS_KERNEL_SOURCES="ds_split.hip" restricts the files).

Two of the numbers carry a property the 512-site throughput depends on ("Sharing a CU"): the fused inception
module must stay at <= 184 VGPRs and the BiLSTM cell kernel at <= 96 (88 today), so that two module waves and one cell
wave fit a SIMD's 512 registers; tests/test_kernel_resources.py asserts them.

usage: python tools/kernel_resources.py [name filter]"""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def kernel_resources():
    text = ""
    only = os.environ.get("DS_KERNEL_SOURCES", "ds_kernels.hip ds_split.hip").split()
    with tempfile.TemporaryDirectory() as tmp:
        for i, name in enumerate(only):
            src = os.path.join(ROOT, "deepsignal_amd", "csrc", name)
            out = os.path.join(tmp, "k%d.s" % i)
            extra = ["-fno-slp-vectorize"] if name == "ds_split.hip" else []      # as deepsignal_amd/csrc/Makefile builds it
            subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-value", "--cuda-device-only", "-S"] + extra +
                           ["-x", "hip", "-o", out, src], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            text += open(out).read()
    res = {}
    for m in re.finditer(r"\.amdhsa_kernel (\S+)\n(.*?)\.end_amdhsa_kernel", text, re.S):
        body = m.group(2)
        get = lambda key: int(re.search(r"\.amdhsa_%s (\d+)" % key, body).group(1))
        name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip() or m.group(1)
        res[name] = {"vgprs": get("next_free_vgpr"), "scratch_bytes": get("private_segment_fixed_size"),
                     "static_lds_bytes": get("group_segment_fixed_size"), "sgpr_spills": 0, "vgpr_spills": 0}
        res[name]["mangled"] = m.group(1)
    # spill counts live in the amdhsa.kernels metadata (one YAML entry per kernel)
    by_mangled = {r.pop("mangled"): r for r in res.values()}
    for m in re.finditer(r"- \.agpr_count:.*?(?=\n  - \.agpr_count:|\namdhsa\.target|\Z)", text, re.S):
        ent = m.group(0)
        nm = re.search(r"\.name:\s+(\S+)", ent)
        if nm and nm.group(1) in by_mangled:
            for key in ("sgpr_spill", "vgpr_spill"):
                c = re.search(r"\.%s_count:\s+(\d+)" % key, ent)
                by_mangled[nm.group(1)][key + "s"] = int(c.group(1)) if c else -1
    return res


if __name__ == "__main__":
    flt = sys.argv[1] if len(sys.argv) > 1 else ""
    for name, r in sorted(kernel_resources().items()):
        if flt in name:
            print("%4d VGPRs  %5d B scratch  %6d B static LDS  %d/%d SGPR/VGPR spills  %s" % (r["vgprs"], r["scratch_bytes"], r["static_lds_bytes"], r["sgpr_spills"], r["vgpr_spills"], name[:110]))
