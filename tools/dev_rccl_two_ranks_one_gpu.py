"""Experiment: does RCCL accept two ranks that share one GPU?  (bench.py with the nccl backend, both
ranks mapped to device 0.)  Prints what happened; never part of a test."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = r'''
import os, sys
sys.path.insert(0, %r)
import torch.distributed as dist
import plonk_amd as pa
from plonk_amd import sharded
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
ctx = pa.Context(0)
try:
    comm = sharded.Communicator(ctx, dist, rank, world, kind="rccl")
    print("rank", rank, "communicator created", flush=True)
except Exception as e:
    print("rank", rank, "failed:", e, flush=True)
''' % ROOT
procs = []
for rank in range(2):
    env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29541")
    procs.append(subprocess.Popen([sys.executable, "-c", WORKER], env=env))
for p in procs:
    try:
        p.wait(timeout=90)
    except subprocess.TimeoutExpired:
        p.kill()
        print("a rank hung; killed")
