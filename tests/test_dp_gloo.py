"""World-size-2 `gloo` test (CPU) of the data-parallel gradient exchange used by the train-mode
backward: bucketed async all-reduce, averaging, unflattening (efficientat_amd/dp.py)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from efficientat_amd.dp import GradReducer


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        shapes = {"a.weight": (64, 16, 1, 1), "a.bias": (64,), "b.weight": (960, 160), "c": (7,), "d.weight": (3, 3)}
        red = GradReducer(bucket_bytes=16 << 10)          # small buckets: several collectives + a tail bucket
        gen = torch.Generator().manual_seed(100 + rank)
        mine = {k: torch.randn(v, generator=gen) for k, v in shapes.items()}
        for k in shapes:                                    # push order = production order in backward
            red.push(k, mine[k].clone())
        out = red.finish()
        # expected: mean over ranks of the same seeded tensors
        exp = {k: sum(torch.randn(v, generator=torch.Generator().manual_seed(100 + r)) if False else
                      _regen(shapes, r)[k] for r in range(world)) / world for k, v in shapes.items()}
        ok = all(torch.allclose(out[k], exp[k], atol=1e-6) and out[k].shape == torch.Size(shapes[k]) for k in shapes)
        # a second round must start clean
        red.push("c", torch.ones(7) * (rank + 1))
        out2 = red.finish()
        ok = ok and set(out2) == {"c"} and torch.allclose(out2["c"], torch.full((7,), (1 + world) / 2.0))
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


def _regen(shapes, rank):
    gen = torch.Generator().manual_seed(100 + rank)
    return {k: torch.randn(v, generator=gen) for k, v in shapes.items()}


def test_grad_reducer_world2_gloo():
    world, port = 2, _free_port()
    ret = mp.get_context("spawn").Manager().dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    assert dict(ret) == {0: True, 1: True}


def _flat_worker(rank, world, port, ret):
    """Round 5: the large gradients are PRODUCED inside the reducer's flat buffer (`alloc`) and all-reduced in place; small
    ones still go through the packed copy.  Pass 1 learns the buffer size (alloc -> None, everything packed), passes 2 and
    3 run in place - all three must return the mean over ranks, and `finish` must hand out the very memory it gave."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        big = {"l.weight": (960, 160), "p.weight": (160, 960), "se.fc2": (33, 7), "e.weight": (960, 160)}
        small = {"bn.weight": (160,), "bn.bias": (160,), "dw": (960, 9)}
        order = ["l.weight", "bn.weight", "bn.bias", "p.weight", "se.fc2", "dw", "e.weight"]
        red = GradReducer(bucket_bytes=256 << 10)
        ok, inplace = True, []
        for it in range(3):
            gen = torch.Generator().manual_seed(1000 * it + rank)
            vals = {k: torch.randn({**big, **small}[k], generator=gen) for k in order}
            handed = {}
            for k in order:
                mem = red.alloc(k, big[k], torch.device("cpu")) if k in big else None
                if mem is not None:
                    assert float(mem.abs().max()) == 0.0            # zero-filled: the kernels accumulate into it
                    mem.add_(vals[k])                               # "the kernel produces the gradient in its bucket"
                    handed[k] = mem
                    red.push(k, mem.view(*mem.shape, 1, 1) if k == "p.weight" else mem)   # pushed in the PARAMETER's shape
                else:
                    red.push(k, vals[k].clone())
            out = red.finish()
            for k in order:
                exp = sum(torch.randn({**big, **small}[k], generator=_gen_upto(1000 * it + r, order, k, {**big, **small}))
                          for r in range(world)) / world
                want = tuple({**big, **small}[k]) + ((1, 1) if (k == "p.weight" and k in handed) else ())
                ok = ok and torch.allclose(out[k].reshape(exp.shape), exp, atol=1e-6) and tuple(out[k].shape) == want
            inplace.append(sorted(k for k in handed if out[k].data_ptr() == handed[k].data_ptr()))
        # memory handed out but never pushed back: finish() must refuse (the bucket would be reduced without that gradient)
        red.alloc("l.weight", big["l.weight"], torch.device("cpu"))
        try:
            red.finish()
            ok = False
        except RuntimeError as e:
            ok = ok and "never pushed" in str(e)
        ret[rank] = (bool(ok), inplace)
    finally:
        dist.destroy_process_group()


def _gen_upto(seed, order, key, shapes):
    """Generator positioned where `key`'s values start in the stream the worker drew (same draw order)."""
    gen = torch.Generator().manual_seed(seed)
    for k in order:
        if k == key:
            return gen
        torch.randn(shapes[k], generator=gen)
    raise KeyError(key)


def test_grad_reducer_in_place_flat_buffer_world2_gloo():
    world, port = 2, _free_port()
    ret = mp.get_context("spawn").Manager().dict()
    mp.spawn(_flat_worker, args=(world, port, ret), nprocs=world, join=True)
    for r in range(world):
        ok, inplace = ret[r]
        assert ok
        assert inplace[0] == []                                                   # first pass: size unknown, packed copies
        assert inplace[1] == inplace[2] == ["e.weight", "l.weight", "p.weight", "se.fc2"]   # then produced in place


def _hook_worker(rank, world, port, ret):
    """install_grad_hooks(only=..., active=...): the head of an MN in trunk mode (its gradients come from torch autograd,
    not from the trunk Function) is averaged across ranks while the flag is up, and left alone while it is down."""
    import torch.nn as nn
    from efficientat_amd.dp import install_grad_hooks
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(0)
        m = nn.Module()
        m.features = nn.Linear(6, 5)
        m.classifier = nn.Sequential(nn.Linear(5, 4), nn.Linear(4, 3))
        flag = {"on": True}
        only = [n for n, _ in m.named_parameters() if n.startswith("classifier.")]
        install_grad_hooks(m, GradReducer(bucket_bytes=64), only=only, active=lambda: flag["on"])
        x = torch.randn(8, 6, generator=torch.Generator().manual_seed(10 + rank))

        def grads():
            m.zero_grad(set_to_none=True)
            m.classifier(m.features(x)).square().sum().backward()
            return {n: p.grad.clone() for n, p in m.named_parameters()}

        on = grads()
        flag["on"] = False
        off = grads()
        gathered = [None] * world
        dist.all_gather_object(gathered, {k: v for k, v in off.items()})
        ok = True
        for n in on:
            mean = sum(g[n] for g in gathered) / world
            if n in only:
                ok = ok and torch.allclose(on[n], mean, atol=1e-5) and (world == 1 or not torch.allclose(off[n], mean, atol=1e-5))
            else:
                ok = ok and torch.allclose(on[n], off[n])           # the trunk's parameters are not this reducer's business
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


def test_head_hooks_reduce_only_in_trunk_mode_world2_gloo():
    world, port = 2, _free_port()
    ret = mp.get_context("spawn").Manager().dict()
    mp.spawn(_hook_worker, args=(world, port, ret), nprocs=world, join=True)
    assert dict(ret) == {0: True, 1: True}


def test_grad_reducer_single_process_passthrough():
    red = GradReducer()
    t = torch.arange(6.0).view(2, 3)
    red.push("w", t)
    out = red.finish()
    assert out["w"] is t and red.finish() == {}


def _hook_worker(rank, world, port, ret):
    """Autograd-driven module (the DyMN path): post-accumulate hooks -> bucketed reducer -> averaged .grad"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from efficientat_amd.dp import enable_data_parallel
        torch.manual_seed(0)
        net = torch.nn.Sequential(torch.nn.Linear(20, 64), torch.nn.ReLU(), torch.nn.Linear(64, 5))
        enable_data_parallel(net, bucket_bytes=2 << 10)
        x = torch.randn(8, 20, generator=torch.Generator().manual_seed(10 + rank))
        net(x).square().mean().backward()
        mine = [p.grad.clone() for p in net.parameters()]
        # reference: every rank recomputes all ranks' local gradients and averages them
        ref = [torch.zeros_like(p) for p in net.parameters()]
        for r in range(world):
            net2 = torch.nn.Sequential(torch.nn.Linear(20, 64), torch.nn.ReLU(), torch.nn.Linear(64, 5))
            net2.load_state_dict(net.state_dict())
            xr = torch.randn(8, 20, generator=torch.Generator().manual_seed(10 + r))
            net2(xr).square().mean().backward()
            for a, p in zip(ref, net2.parameters()):
                a += p.grad / world
        ret[rank] = all(torch.allclose(a, b, atol=1e-6) for a, b in zip(mine, ref))
    finally:
        dist.destroy_process_group()


def test_grad_hooks_world2_gloo():
    world, port = 2, _free_port()
    ret = mp.get_context("spawn").Manager().dict()
    mp.spawn(_hook_worker, args=(world, port, ret), nprocs=world, join=True)
    assert dict(ret) == {0: True, 1: True}


def _worker_modes(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # local=True: no collective although a process group exists (a model that was never handed to
        # enable_data_parallel must not all-reduce behind the caller's back, e.g. under torch DDP or rank-0-only backward)
        red = GradReducer(local=True)
        t = torch.full((5,), float(rank + 1))
        if rank == 0:                      # only ONE rank pushes: a hidden collective would deadlock here
            red.push("w", t)
            out = red.finish()
            ok = out["w"] is t
        else:
            ok = True
        dist.barrier()
        # default reducer in the same group: averaged
        red2 = GradReducer(bucket_bytes=8)
        red2.push("w", t.clone())
        red2.push("v", torch.ones(3) * rank)
        out2 = red2.finish()
        ok = ok and torch.allclose(out2["w"], torch.full((5,), (1 + world) / 2.0)) and \
            torch.allclose(out2["v"], torch.full((3,), (world - 1) / 2.0))
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


def test_local_reducer_issues_no_collective_world2_gloo():
    world, port = 2, _free_port()
    ret = mp.get_context("spawn").Manager().dict()
    mp.spawn(_worker_modes, args=(world, port, ret), nprocs=world, join=True)
    assert dict(ret) == {0: True, 1: True}


def _worker_forced(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # force_buckets: the pack / all-reduce / unpack path with ONE rank must be the identity (what
        # EAT_BENCH_FORCE_DIST and the single-GPU reducer test rely on)
        red = GradReducer(bucket_bytes=64, force_buckets=True)
        g = torch.Generator().manual_seed(3)
        ts = {f"p{i}": torch.randn(7, i + 1, generator=g) for i in range(6)}
        for k, v in ts.items():
            red.push(k, v.clone())
        out = red.finish()
        ret[rank] = all(torch.equal(out[k], v) and out[k].shape == v.shape for k, v in ts.items())
    finally:
        dist.destroy_process_group()


def test_forced_bucketing_with_one_rank_is_identity():
    port = _free_port()
    ret = mp.get_context("spawn").Manager().dict()
    mp.spawn(_worker_forced, args=(1, port, ret), nprocs=1, join=True)
    assert dict(ret) == {0: True}
