"""GPU: the C-ABI collective (include/lipvq.h lipvq_comm_* / lipvq_allreduce_counts) binds RCCL and runs.  One GPU is all
this box has, so the communicator has one rank (RCCL refuses two ranks on one device): that exercises the run-time
binding, the unique-id hand-off, communicator creation on the current device, the in-place int64/fp32 sums on a side
stream and the event ordering.  The N-rank arithmetic of the same call pattern is covered over gloo in
tests/test_distributed_cpu.py and tests/test_bench_launcher.py."""
import pytest
import torch

pytestmark = pytest.mark.gpu


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
