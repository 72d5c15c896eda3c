"""Peer-mapped direct-write gather (ggq_peer_* + ggq.dist.PeerSlabGather): two / three processes that share the one GPU of the
test box map each other's gather buffer through HIP IPC and write their slabs into it; the result must equal the
one-process matmul bit for bit.  (On a multi-GPU node the same code crosses xGMI; none was available to this build.)"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("world", [2, 3])
def test_peer_slab_gather_processes_on_one_gpu(world):
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(29541 + world), os.path.join(here, "_peer_gather_worker.py")]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
    assert p.returncode == 0 and "peer gather ok" in p.stdout, p.stdout[-2000:] + p.stderr[-3000:]
