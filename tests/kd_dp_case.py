"""BASELINE configs[4] pieces wired together (bodies of tests/test_gpu_trainloop.py::test_kd_*; own process because a
process-group failure aborts the interpreter):

  python tests/kd_dp_case.py two_rank   two ranks (one GPU each over RCCL when two are visible, else both on cuda:0 through
                                        gloo): `KDTrainer.loss_and_backward` on a model handed to `enable_data_parallel` leaves
                                        in `.grad` the MEAN over the ranks of the gradients a never-wrapped replica computes
                                        for the same shard with the same host draws (mel jitter, mixup permutation / lambdas,
                                        teacher rows) - the reference's DDP semantics (ex_pl_audioset.py:287-293) for the loop
                                        of ex_audioset.py:139-196.  Prints KD_TWO_RANK_OK.
  python tests/kd_dp_case.py graph_rccl one rank over RCCL with forced bucketing: `GraphedKDTrainer` (mel + mixup + forward +
                                        KD loss + backward + bucketed all-reduce + Adam in ONE hipGraph) follows the eager
                                        `KDTrainer` of the same seeds step by step.  Prints KD_GRAPH_RCCL_OK.
"""
import contextlib
import io
import os
import socket
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

B, L, NCLS, NTEACH = 6, 32000, 527, 40


def _quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def make_model(seed, dev):
    from efficientat_amd.mn import get_model
    torch.manual_seed(seed)
    m = _quiet(get_model, width_mult=0.5).to(dev)
    with torch.no_grad():
        for mod in m.modules():
            if isinstance(mod, torch.nn.Conv2d):
                fan_in = mod.weight.shape[1] * mod.weight.shape[2] * mod.weight.shape[3]
                mod.weight.normal_(0, (2.0 / fan_in) ** 0.5)
            if isinstance(mod, torch.nn.Dropout):
                mod.p = 0.0
    m.train()
    return m


def make_mel(dev, **kw):
    from efficientat_amd.preprocess import AugmentMelSTFT
    return _quiet(AugmentMelSTFT, **{"freqm": 0, "timem": 0, **kw}).to(dev).train()


def shard(rank, step, dev):
    g = torch.Generator().manual_seed(1000 * rank + step)
    x = (0.1 * torch.randn(B, 1, L, generator=g)).clamp_(-1, 1).to(dev)
    y = (torch.rand(B, NCLS, generator=g) < 0.01).float().to(dev)
    names = ["syn%07d" % (17 * rank + 3 * step + i) for i in range(B)]
    return x, names, y


def teacher_table():
    g = torch.Generator().manual_seed(5)
    preds = torch.randn(NTEACH, NCLS, generator=g) * 2 - 5
    f2i = {"syn%07d" % i: i % NTEACH for i in range(0, 60, 2)}             # every other file has a teacher row
    return preds, f2i


def _seed_host(s):
    torch.manual_seed(s)
    np.random.seed(s)


def _two_rank_worker(rank, world, port, multi_gpu):
    from efficientat_amd.dp import enable_data_parallel
    from efficientat_amd.train_loop import KDTrainer
    dev = torch.device(f"cuda:{rank}" if multi_gpu else "cuda:0")
    torch.cuda.set_device(dev)
    if multi_gpu:
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        probe = torch.ones(8, device=dev)
        dist.all_reduce(probe)
        assert float(probe[0]) == world
    except Exception as e:
        if rank == 0:
            print("KD_TWO_RANK_SKIP", repr(e)[:300], flush=True)
        dist.destroy_process_group()
        return
    preds, f2i = teacher_table()
    dp = make_model(7 + rank, dev)                                           # replicas start different: the broadcast fixes it
    enable_data_parallel(dp, bucket_bytes=64 << 10)
    ref = make_model(7 + rank, dev)
    ref.load_state_dict(dp.state_dict())
    mk = lambda m: KDTrainer(m, make_mel(dev), torch.optim.SGD(m.parameters(), lr=0.0), preds, f2i, kd_lambda=0.1,
                             mixup_alpha=0.3)
    t_dp, t_ref = mk(dp), mk(ref)
    x, names, y = shard(rank, 0, dev)
    _seed_host(100 + rank)
    l_ref = t_ref.loss_and_backward(x, names, y)
    exp = {}
    for n, p in ref.named_parameters():
        if p.grad is not None:
            t = p.grad.detach().clone()
            dist.all_reduce(t)
            exp[n] = t / world
    _seed_host(100 + rank)                                                   # the same mel jitter / mixup draws
    l_dp = t_dp.loss_and_backward(x, names, y)
    assert abs(float(l_ref) - float(l_dp)) < 1e-5 * max(1.0, abs(float(l_ref))), (float(l_ref), float(l_dp))
    gmax = max(float(v.abs().max()) for v in exp.values())
    worst, wname, loc = 0.0, None, 0.0
    for n, p in dp.named_parameters():
        if n not in exp:
            continue
        den = max(float(exp[n].abs().max()), 1e-3 * gmax)
        err = float((p.grad - exp[n]).abs().max()) / den
        loc = max(loc, float((ref.get_parameter(n).grad - exp[n]).abs().max()) / den)
        if err > worst:
            worst, wname = err, n
    print(f"rank {rank}: loss {float(l_dp):.6f}; worst |dp - mean of local KD gradients| = {worst:.2e} ({wname}); "
          f"local vs mean = {loc:.2e}", flush=True)
    ok = worst < 2e-3 and loc > 0.05
    flag = torch.tensor([1.0 if ok else 0.0], device=dev)
    dist.all_reduce(flag)
    torch.cuda.synchronize()
    if rank == 0 and float(flag[0]) == world:
        print("KD_TWO_RANK_OK", flush=True)
    dist.destroy_process_group()


def graph_rccl():
    from efficientat_amd.dp import enable_data_parallel
    from efficientat_amd.train_loop import GraphedKDTrainer, KDTrainer
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "0")
    dev = torch.device("cuda:0")
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
    preds, f2i = teacher_table()
    res = {}
    for tag in ("eager", "graph"):
        m = make_model(3, dev)
        enable_data_parallel(m, bucket_bytes=64 << 10, force_buckets=True)
        assert m._grad_reducer.bucketed
        opt = torch.optim.Adam(m.parameters(), lr=1e-3, capturable=True, fused=True)
        kw = dict(teacher_preds=preds, fname_to_index=f2i, kd_lambda=0.1, mixup_alpha=0.3)
        tr = (KDTrainer(m, make_mel(dev), opt, **kw) if tag == "eager"
              else GraphedKDTrainer(m, make_mel(dev), opt, B, L, **kw))
        _seed_host(11)
        losses = []
        for step in range(3):
            x, names, y = shard(0, step, dev)
            losses.append(float(tr.step(x, names, y)))
        torch.cuda.synchronize()
        res[tag] = (losses, torch.cat([p.detach().reshape(-1) for p in m.parameters()]).cpu(), tr.epoch_stats())
    le, lg = res["eager"][0], res["graph"][0]
    print("losses eager", le, "graph", lg, flush=True)
    assert all(abs(a - b) < 2e-5 * max(1.0, abs(a)) for a, b in zip(le, lg)), (le, lg)
    d = (res["eager"][1] - res["graph"][1]).abs()
    # three Adam steps at lr 1e-3 move a weight by <= 3e-3; the two runs differ by round-off in the gradients, which Adam's
    # normalisation turns into a bounded step difference for the (few) elements whose gradient is itself round-off
    frac = float((d > 1e-4).float().mean())
    print(f"parameters after 3 steps: max |eager - graph| = {float(d.max()):.2e}, fraction above 1e-4: {frac:.2e}", flush=True)
    assert float(d.max()) <= 6.1e-3 and frac < 0.02, (float(d.max()), frac)
    se, sg = res["eager"][2], res["graph"][2]
    assert all(abs(se[k] - sg[k]) < 2e-5 * max(1.0, abs(se[k])) for k in se), (se, sg)
    print("KD_GRAPH_RCCL_OK", flush=True)
    torch.cuda.synchronize()
    dist.destroy_process_group()


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "two_rank"
    if what == "graph_rccl":
        return graph_rccl()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_two_rank_worker, args=(2, port, torch.cuda.device_count() >= 2), nprocs=2, join=True)


if __name__ == "__main__":
    try:
        main()
    except BaseException:
        import traceback
        print("KD_CASE_FAILED\n" + traceback.format_exc(), flush=True)
        raise
