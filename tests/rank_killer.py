"""Test launcher hop (tests/test_rccl_multi_gpu.py): `rank_killer.py <rank> <seconds> <script> [args...]` runs <script> as
__main__ in this process; in rank <rank> (RANK of torch.distributed.run) a timer ends the process abruptly after <seconds> -
a rank that dies at an arbitrary point of a multi-rank run, possibly inside a collective.  Not part of the product or of bench.py."""
import os
import runpy
import sys
import threading

if __name__ == "__main__":
    rank, after, script = int(sys.argv[1]), float(sys.argv[2]), sys.argv[3]
    if int(os.environ.get("RANK", "0")) == rank:
        t = threading.Timer(after, lambda: os._exit(17))
        t.daemon = True
        t.start()
    sys.argv = [script] + sys.argv[4:]
    runpy.run_path(script, run_name="__main__")
