"""Which build a profile belongs to, recorded AT COLLECTION TIME on the GPU box (which has no .git): the commit the snapshot was
taken from -- written into build/COMMIT by tools/stamp_commit.sh before the gpurun call -- and a hash of the kernel / engine
sources as they lie in the snapshot, which anyone can recompute from the repository at that commit
(sha256 over ds_kernels.hip, ds_split.hip, ds_engine.cpp, ds_io.cpp, ds_internal.h, ds_device.h in this order, first 16 hex digits)."""
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SOURCES = ("ds_kernels.hip", "ds_split.hip", "ds_engine.cpp", "ds_io.cpp", "ds_internal.h", "ds_device.h")


def sources_sha16():
    h = hashlib.sha256()
    for name in SOURCES:
        with open(os.path.join(ROOT, "deepsignal_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def build_identity():
    commit = None
    try:
        commit = open(os.path.join(ROOT, "build", "COMMIT")).read().strip() or None
    except OSError:
        pass
    return {"commit": commit, "sources_sha16": sources_sha16()}


if __name__ == "__main__":
    print(build_identity())
