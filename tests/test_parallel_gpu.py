"""The gradient exchange on real GPUs: the C ABI's RCCL communicator (ctu_comm_*) and the two data-parallel step forms.

On the one-GPU box only the 1-rank cases run (communicator init / all-reduce / destroy through the C ABI on a real
device); the 2-rank cases need two GPUs (skipped otherwise; the driver's multi-GPU node runs them): they check that the
gradients every rank ends up with are the mean of the two single-rank gradients, for both the eager bucketed path
(UNetEngine.backward(sync=GradSync)) and the graph path (GraphedTrainStep(distributed=True))."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from util import gen, onehot_target

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_one_rank_communicator_through_the_c_abi():
    from ctunet_amd import parallel
    if not dist.is_initialized():
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29541", rank=0, world_size=1,
                                device_id=torch.device("cuda", 0))
    try:
        comm = parallel.get_communicator()
        assert comm.world == 1 and comm.rank == 0
        t = torch.randn(1000003, generator=gen(1)).cuda()
        ref = t.clone()
        comm.allreduce_(t, average=True)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        comm.allreduce_(t, average=False, stream=side)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        assert torch.equal(t, ref)
        with pytest.raises(RuntimeError):
            comm.allreduce_(torch.zeros(4))              # CPU tensor: no fallback
        # bucketed sync on one rank is a no-op that returns nothing
        s = parallel.GradSync()
        s.push([("a", t)])
        assert s.finish() == {}
    finally:
        parallel.close_communicators()
        dist.destroy_process_group()


def _rank_inputs(rank):
    x = torch.randn(1, 1, 32, 32, 32, generator=gen(100 + rank))
    t = onehot_target((1, 2, 32, 32, 32), 200 + rank, 0.2)
    return x, t


def _worker(rank, world, port, tmp):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "ct-unet_amd"), os.path.join(ROOT, "tests")]
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    try:
        import ctunet_amd
        from ctunet_amd import losses as L, optim as O2, parallel
        from ctunet_amd.graph import GraphedTrainStep

        def single_grads(r):
            torch.manual_seed(0)
            net = ctunet_amd.UNet(n_blocks=2, use_checkpoint=False).cuda().train()
            x, t = _rank_inputs(r)
            ce, dc = L.fused_ce_dice(net(x.cuda()), t.cuda(), 1.0, 1.0, False)
            (ce + dc).backward()
            return {n: (None if p.grad is None else p.grad.clone()) for n, p in net.named_parameters()}
        g = [single_grads(r) for r in range(world)]
        # (a) eager, bucketed, overlapped: tiny buckets force several collectives from inside backward
        torch.manual_seed(0)
        net = ctunet_amd.UNet(n_blocks=2, use_checkpoint=False).cuda().train()
        parallel.distribute(net, bucket_bytes=4096)
        x, t = _rank_inputs(rank)
        ce, dc = L.fused_ce_dice(net(x.cuda()), t.cuda(), 1.0, 1.0, False)
        (ce + dc).backward()
        # most bucket collectives were issued from push(), i.e. while the backward kernels were still being launched
        assert net.__dict__["_last_grad_sync"].launched_before_finish >= 2
        for n, p in net.named_parameters():
            if g[0][n] is None:
                assert p.grad is None, n
            else:
                exp = sum(gr[n] for gr in g) / world
                assert torch.allclose(p.grad, exp, rtol=1e-5, atol=1e-7), n
        # (b) graph path (chain of graph segments with the bucket all-reduces between them): two steps -- one eager
        #     warm-up inside the constructor, one replay -- must leave every rank with the same parameters as two steps of
        #     the eager bucketed path (a) from identical weights
        def two_steps(graph):
            torch.manual_seed(0)
            m = ctunet_amd.UNet(n_blocks=2, use_checkpoint=False).cuda().train()
            opt = O2.Adam(m.parameters(), lr=1e-2)
            xx, tt = _rank_inputs(rank)
            xx, tt = xx.cuda(), tt.cuda()
            if graph:
                parallel.broadcast_parameters(m)
                gs = GraphedTrainStep(m, opt, xx, [tt], 1.0, 1.0, warmup=1, distributed=True, bucket_bytes=4096)
                assert len(gs.flats) >= 3 and len(gs.segments) == len(gs.flats) + 1
                gs(xx, [tt])
            else:
                parallel.distribute(m, bucket_bytes=4096)
                for _ in range(2):
                    ce_, dc_ = L.fused_ce_dice(m(xx.clone().requires_grad_(True)), tt, 1.0, 1.0, False)
                    (ce_ + dc_).backward()
                    opt.step()
                    for p in m.parameters():
                        p.grad = None
            torch.cuda.synchronize()
            return m
        m_e, m_g = two_steps(False), two_steps(True)
        for (n, a), (_, b) in zip(m_e.state_dict().items(), m_g.state_dict().items()):
            assert torch.allclose(a.float(), b.float(), rtol=1e-4, atol=1e-6), n
        flat = torch.cat([p.detach().flatten() for p in m_g.parameters()])
        other = [torch.empty_like(flat) for _ in range(world)]
        dist.all_gather(other, flat)
        assert all(torch.equal(other[0], o) for o in other)            # the ranks stay in lock step
        open(os.path.join(tmp, f"ok{rank}"), "w").write("ok")
    finally:
        from ctunet_amd import parallel as par
        par.close_communicators()
        dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (RCCL refuses two ranks on one device)")
def test_two_ranks_gradients_are_the_mean(tmp_path):
    world = 2
    port = 29600 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(world))
