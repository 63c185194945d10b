"""Two ranks, two GPUs, the RCCL communicator: the data plane a multi-GPU run uses (ncclAllGather over xGMI inside the
library; gloo carries only the RCCL id and the barriers).  Needs two devices, so it is skipped on a one-GPU box; it sits in
a file of its own, collected after the parity tests, because it is the first place where RCCL runs among several ranks."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT
from test_dist import GPU_WORKER


def _gpu_count():
    import torch
    return torch.cuda.device_count()   # counts devices without initialising the GPU runtime in this process


@pytest.mark.gpu
@pytest.mark.skipif(_gpu_count() < 2, reason="needs two GPUs: RCCL wants a device per rank")
@pytest.mark.parametrize("case", ["16", "20", "16g", "pi9"])
def test_sharded_prove_rccl_two_gpus(tmp_path, case):
    """plonk_prove_3wire_sharded / _grouped with two ranks in two fresh processes, a GPU each, the all-gathers
    by ncclAllGather inside the library: both ranks emit the oracle's proof bytes (golden 2^16, the 2^20-gate
    benchmark configuration, a grouped placement, a circuit with public inputs and a key swap)."""
    grouped = case.endswith("g")
    case = case.rstrip("g")
    script = tmp_path / "gpu_worker.py"
    script.write_text(GPU_WORKER % {"root": ROOT})
    procs = []
    port = 29900 + sum(map(ord, case)) % 16 + (16 if grouped else 0)
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   PLONK_TEST_CASE=case, PLONK_TEST_GROUPED="1" if grouped else "0", PLONK_TEST_KIND="rccl",
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=280)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out.decode())
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, out[-3000:]
        assert "rank %d ok" % rank in out
