"""Starting the N ranks of bench.py on one node, and the one JSON line."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH_PY = os.path.join(ROOT, "bench.py")


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_command(n, argv, port=None):
    """the command that starts n ranks of this script on this node (one per GPU, rendezvous on the loopback address)"""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
            "--master-port", str(port or free_port()), BENCH_PY] + list(argv)


def self_launch(args, argv):
    """`python bench.py --gpus N` without a launcher: check the box, then start the N ranks ourselves"""
    if not args.selftest_cpu:
        import torch
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if os.environ.get("SIS3D_BENCH_SHARE_GPU") and have >= 1:
            have = args.gpus        # functional test hook: every rank on GPU 0, gloo instead of RCCL (see main())
        if have < args.gpus:
            sys.stderr.write("bench.py: --gpus %d requested but this box exposes %d GPU(s); refusing to run a smaller world\n"
                             % (args.gpus, have))
            return 2
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, min(16, (os.cpu_count() or 16) // args.gpus))))
    return subprocess.call(launch_command(args.gpus, argv), env=env)


def emit(line):
    """the JSON line must be the LAST thing on stdout: RCCL printf()s a version banner into C stdio's buffer, which would
    otherwise be flushed at exit, after our line"""
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stdout.flush()
    print(json.dumps(line), flush=True)
