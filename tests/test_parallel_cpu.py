"""N > 1 path on CPU: two gloo ranks, gradient bucketing/averaging of ctunet_amd.parallel.

Checks the property the reference's DataParallel step defines: after the exchange every
rank holds the mean over ranks of each live gradient, parameters the graph never touches (None)
are skipped on every rank, and parameters/buffers start identical after ``distribute``."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _worker(rank, world, port, tmp):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "ct-unet_amd")]
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from ctunet_amd import parallel
        import ctunet_amd
        # 1. broadcast makes replicas identical
        torch.manual_seed(100 + rank)
        net = ctunet_amd.UNet(n_blocks=2, i_size=2)
        parallel.distribute(net)
        flat = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
        gathered = [torch.empty_like(flat) for _ in range(world)]
        dist.all_gather(gathered, flat)
        assert all(torch.equal(gathered[0], g) for g in gathered)
        # 2. block-wise push in backward order with tiny buckets (forces several collectives) + unused params
        names = [n for n, _ in net.named_parameters()]
        g = torch.Generator().manual_seed(7 + rank)
        grads = {n: (None if n.startswith("cblock.") else torch.randn(p.shape, generator=g))
                 for n, p in net.named_parameters()}
        sync = parallel.GradSync(None, bucket_bytes=256)
        live = [(n, t.clone()) for n, t in grads.items() if t is not None]
        for i in range(0, len(live), 3):
            sync.push(live[i:i + 3])
        assert sync.launched_before_finish >= 3          # collectives were issued from push(), i.e. under "backward"
        red = sync.finish()
        assert set(red) == {n for n, _ in live}
        # the default bucket size splits UNet()'s 3.3 MB into several buckets, most of them launched before finish()
        big = ctunet_amd.UNet()
        order = ["last_conv"] + [f"u_blocks.{j}." for j in (3, 2, 1, 0)] + [f"d_blocks.{i}." for i in (3, 2, 1, 0)]
        s2 = parallel.GradSync(None)
        for pre in order:
            s2.push([(n, torch.zeros_like(p)) for n, p in big.named_parameters() if n.startswith(pre)])
        assert s2.launched_before_finish >= 3, s2.launched_before_finish
        assert len(s2.finish()) == 58
        for n, t in live:
            exp = torch.zeros_like(t)
            for r in range(world):
                gr = torch.Generator().manual_seed(7 + r)
                for n2, p2 in net.named_parameters():
                    v = None if n2.startswith("cblock.") else torch.randn(p2.shape, generator=gr)
                    if n2 == n:
                        exp += v
            assert torch.allclose(red[n], exp / world, atol=1e-6), n
        # 3. stand-alone in-place form
        ts = [torch.full((5,), float(rank + 1)), None, torch.full((2, 3), float(10 * (rank + 1)))]
        parallel.allreduce_mean_(ts)
        assert torch.allclose(ts[0], torch.full((5,), (1 + world) / 2.0))
        assert ts[1] is None and torch.allclose(ts[2], torch.full((2, 3), 10 * (1 + world) / 2.0))
        # 4. graph._Segmenter (the sync object of the segmented graph step) under two ranks, in the engine's backward order
        # (head, decoder top -> bottom, encoder bottom -> top): with bucket_bytes = 1 EVERY push closes a bucket, with the
        # default size only finish() does; the boundary callback is the all-reduce.  Bucket order, layout and means must be
        # the same on every rank.
        from ctunet_amd.graph import _Segmenter
        order = ["last_conv"] + [f"u_blocks.{j}." for j in (1, 0)] + [f"d_blocks.{i}." for i in (1, 0)]
        for bucket_bytes, nb_expected in ((1, len(order)), (parallel.DEFAULT_BUCKET_BYTES, 1)):
            fired = []

            def boundary(k, flat):
                fired.append(k)
                dist.all_reduce(flat)
                flat /= world
            seg = _Segmenter(bucket_bytes, boundary)
            for pre in order:
                seg.push([(n, grads[n].clone()) for n in names if n.startswith(pre)])
            out = seg.finish()
            assert fired == list(range(nb_expected)) and len(seg.flats) == nb_expected, (fired, len(seg.flats))
            lay = [[(n, tuple(sh), off) for n, sh, off in bucket] for bucket in seg.layout]
            every = [None] * world
            dist.all_gather_object(every, lay)
            assert all(e == every[0] for e in every)                   # same buckets, same order, same offsets on every rank
            assert [n for bucket in lay for n, _, _ in bucket] == [n for pre in order for n in names if n.startswith(pre)]
            assert set(out) == {n for n, _ in live}
            for n, t in live:
                assert torch.allclose(out[n], red[n], atol=1e-6), n    # the same means GradSync produced
        open(os.path.join(tmp, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


def test_gloo_world2_gradient_mean(tmp_path):
    world = 2
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(world))


def test_single_process_is_noop():
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ct-unet_amd"))
    from ctunet_amd import parallel
    s = parallel.GradSync()
    s.push([("a", torch.ones(3))])
    assert s.finish() == {}


def test_graph_segmenter_cuts_backward_at_bucket_boundaries():
    """graph._Segmenter (the sync object of the segmented distributed step): buckets close when bucket_bytes are pending,
    `boundary` fires once per bucket in order, finish() returns views of the flat buffers with the pushed values."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ct-unet_amd"))
    from ctunet_amd.graph import _Segmenter
    seen = []
    seg = _Segmenter(100, lambda k, flat: seen.append((k, flat.numel())))
    vals = {f"t{i}": torch.full((4 + i,), float(i)) for i in range(8)}       # 16, 20, 24, ... bytes
    names = list(vals)
    seg.push([(n, vals[n]) for n in names[:3]])          # 60 bytes: open
    assert seen == []
    seg.push([(n, vals[n]) for n in names[3:5]])         # +60: closes bucket 0 (5 tensors)
    seg.push([(names[5], vals[names[5]])])               # 36: open
    out = seg.finish()                                   # closes bucket 1
    seg2 = seg.finish()
    assert seen == [(0, 4 + 5 + 6 + 7 + 8), (1, 9)] and set(out) == set(names[:6]) and set(seg2) == set(out)
    for n in names[:6]:
        assert torch.equal(out[n], vals[n]) and out[n].shape == vals[n].shape
