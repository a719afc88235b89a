"""GPU: the C-ABI collective (include/lipvq.h lipvq_comm_* / lipvq_allreduce_counts) binds RCCL and runs.  On a box with two
or more GPUs test_world2_rccl runs the real thing (two ranks, two devices).  One GPU is all
the development box has, so there the communicator has one rank (RCCL refuses two ranks on one device): that exercises the run-time
binding, the unique-id hand-off, communicator creation on the current device, the in-place int64/fp32 sums on a side
stream and the event ordering.  The N-rank arithmetic of the same call pattern is covered over gloo in
tests/test_distributed_cpu.py and tests/test_bench_launcher.py."""
import os
import socket
import subprocess
import sys
from pathlib import Path

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def _run_world(world, backend, timeout=300):
    """`world` fresh rank processes (children of this one, never an exec of it) of tests/rccl_world2_worker.py."""
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    procs = [subprocess.Popen([sys.executable, str(ROOT / "tests" / "rccl_world2_worker.py"), str(r), str(world), str(port), backend],
                              env=env, cwd=str(ROOT), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(world)]
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=timeout)[0])
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
                p.communicate()
    return [p.returncode for p in procs], outs


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (one rank per device: RCCL refuses two ranks on one)")
def test_world2_rccl():
    """The first N > 1 RCCL execution of this code base is a test, not a bench: two ranks, one GPU each, the batch sharded,
    the code-usage histogram summed (a) by torch.distributed over nccl (= RCCL) and (b) by the library's own
    lipvq_allreduce_counts / lipvq_allreduce_f32 -- both equal the single-process histogram, shards concatenate."""
    rcs, outs = _run_world(2, "nccl")
    assert rcs == [0, 0], "\n".join(o[-1500:] for o in outs)
    assert all("ok" in o for o in outs)


def test_world2_worker_rehearsal_over_gloo_on_one_gpu():
    """The same worker with both ranks on cuda:0 and gloo as the transport (what a one-GPU box can run): sharding, the delta
    all-reduce and the concatenation check execute; only the RCCL leg is skipped."""
    rcs, outs = _run_world(2, "gloo")
    assert rcs == [0, 0], "\n".join(o[-1500:] for o in outs)


def test_world1_worker_over_nccl():
    """The same worker with ONE rank over nccl (= RCCL): every leg -- ShardedTokenizer's delta all-reduce, lipvq_allreduce_counts /
    lipvq_allreduce_f32 on the library's own communicator, and bench.py's bucket pattern (an asynchronous [M][K] all-reduce issued
    while the next tokenize launches are in flight) through torch.distributed AND the C ABI -- runs on a real RCCL communicator."""
    rcs, outs = _run_world(1, "nccl")
    assert rcs == [0], "\n".join(o[-1500:] for o in outs)
    assert "ok" in outs[0]


def test_world1_communicator_allreduce_counts_and_f32():
    import lipvq_vae_amd  # noqa: F401
    from lipvq_vae_amd.sharded import RcclCounts
    torch.cuda.set_device(0)
    rc = RcclCounts()
    assert rc.world == 1 and rc.rank == 0
    K = 1024
    usage = torch.randint(0, 1000, (K,), device="cuda", dtype=torch.int64)
    want = usage.clone()
    for _ in range(3):                       # repeated, back to back, with compute queued behind the wait
        ev = rc.all_reduce(usage)
        rc.wait(ev)
        usage += 1
        want += 1
    torch.cuda.synchronize()
    assert torch.equal(usage, want)
    buf = torch.randn(4099, device="cuda")
    want_f = buf.clone()
    rc.wait(rc.all_reduce_f32(buf))
    torch.cuda.synchronize()
    assert torch.equal(buf, want_f)
    with pytest.raises(TypeError):
        rc.all_reduce(torch.zeros(4, device="cuda", dtype=torch.int32))
    rc.close()


def test_bad_arguments_are_reported_not_crashed():
    from lipvq_vae_amd import _capi
    assert _capi.lib.lipvq_allreduce_counts(None, 4, None, None) != 0
    assert b"allreduce_counts" in _capi.lib.lipvq_last_error()
    assert _capi.lib.lipvq_comm_destroy(None) == 0
