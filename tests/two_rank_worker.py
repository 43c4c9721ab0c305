"""Child of tests/test_gpu_two_procs.py: one rank of a `python -m torch.distributed.run --nproc-per-node N` job whose ranks
ALL use GPU 0 (the one GPU of a test box), each with its own Engine, collectives on gloo.

usage: two_rank_worker.py <mode> <features.tsv> <weights.dsw> <result.tsv> [chunk_bytes]
  mode ok     the product's sharded call_mods route, end to end
  mode raise  the same, but the last rank's engine raises on its third batch (every rank must exit non-zero, no hang)
  mode die    the same, but the last rank's process dies without a word on its third batch
DS_TEST_PRECISION (fp32 | bf16x3) picks the engines' precision.
"""
import datetime
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    mode, tsv, wfile, out = sys.argv[1:5]
    chunk = int(sys.argv[5]) if len(sys.argv) > 5 else 1 << 19
    import torch.distributed as dist
    from deepsignal_amd import call_modifications as cm
    from deepsignal_amd.engine import Engine
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=float(os.environ.get("DS_TEST_PG_TIMEOUT", "60"))))

    class Faulty(Engine):
        calls = 0

        def _trip(self):
            Faulty.calls += 1
            if Faulty.calls == 3:
                if mode == "die":
                    os._exit(17)
                raise RuntimeError("injected engine failure on rank %d" % rank)

        def submit(self, *a):
            self._trip()
            return Engine.submit(self, *a)

        def submit_parts(self, parts):
            self._trip()
            return Engine.submit_parts(self, parts)

    cls = Faulty if (mode in ("raise", "die") and rank == world - 1) else Engine
    eng = cls(device=0, max_batch=512, precision=os.environ.get("DS_TEST_PRECISION", "fp32"))
    eng.load_weights_file(wfile)
    cm.SHARD_CHUNK_BYTES = chunk
    n = cm.call_mods(tsv, "unused", out, 17, 360, 512, 0.001, 2, 1, True, True, True, True, None, engine=eng, dist=dist,
                     f5_batch_num=20)
    eng.close()
    with open(out + ".rank%d.count" % rank, "w") as f:
        f.write(str(n))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
