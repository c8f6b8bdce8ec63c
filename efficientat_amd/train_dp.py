"""Data-parallel knowledge-distillation training of an MN / DyMN on AudioSet: BASELINE configs[4] as one runnable program.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        -m efficientat_amd.train_dp --batch_size 256 [--model_name mn10_as] [...]          (one process per GPU, RCCL)
    python -m efficientat_amd.train_dp ...                                                  (one GPU, no process group)

What it must equal: the reference's training loop (`ex_audioset.py:123-220`: mel -> mixup -> model -> BCE + KD loss ->
backward -> Adam, per-epoch LambdaLR schedule, DyMN temperature update per epoch) under the reference's only multi-GPU
behaviour, PyTorch-Lightning DDP (`ex_pl_audioset.py:287-293`): one process per GPU, `--batch_size` clips PER GPU, local
BatchNorm statistics, gradients averaged over the ranks, the class-balancing sampler of `get_ft_weighted_sampler` cut into
per-rank shards (Lightning wraps a custom sampler into a DistributedSamplerWrapper: rank r takes draws r, r + N, ... of
the epoch's sample list).

Pipeline per rank: `datasets.audioset.get_full_training_set` (dropin/: the reference's tuple layout) -> DataLoader workers
-> `DevicePrefetcher` (pinned double buffering, copies on their own stream; `--transport int16` halves the PCIe bytes)
-> `GraphedKDTrainer` (the whole iteration incl. the bucketed RCCL all-reduce as ONE hipGraph replay; `--no_graph`: the
eager `KDTrainer`) with `enable_data_parallel`.  Out of scope here: wandb logging, mAP evaluation and
checkpoint rotation - rank 0 prints one line per epoch and saves `<out>/<model>_epoch_<e>.pt` when `--out` is given.
"""
import argparse
import contextlib
import io
import json
import os
import pickle
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

from .utils import NAME_TO_WIDTH, exp_warmup_linear_down


class RankShardSampler(torch.utils.data.Sampler):
    """Rank r's share of a base sampler's epoch: draws r, r + world, ... of the list every rank generates identically (the
    base sampler is seeded with `seed + epoch` on every rank) - what Lightning's DistributedSamplerWrapper does with the
    reference's WeightedRandomSampler.  The tail that does not fill a draw for every rank is dropped (equal step counts:
    an uneven last step would deadlock the collective)."""

    def __init__(self, base, rank, world, seed=0):
        self.base, self.rank, self.world, self.seed, self.epoch = base, rank, world, seed, 0

    def set_epoch(self, epoch):
        self.epoch = epoch

    def __len__(self):
        return len(self.base) // self.world

    def __iter__(self):
        g = torch.Generator().manual_seed(self.seed + self.epoch)
        if hasattr(self.base, "generator"):
            self.base.generator = g
        idx = list(iter(self.base))
        n = len(idx) // self.world * self.world
        return iter(idx[self.rank:n:self.world])


def parse_args(argv=None):
    p = argparse.ArgumentParser(description="DP KD training on the HIP path (ex_audioset.py's training arguments)")
    p.add_argument("--model_name", default="mn10_as")
    p.add_argument("--model_width", type=float, default=None, help="default: the width of --model_name")
    p.add_argument("--batch_size", type=int, default=120, help="clips per GPU per step (reference default 120)")
    p.add_argument("--num_workers", type=int, default=12)
    p.add_argument("--n_epochs", type=int, default=200)
    p.add_argument("--epoch_len", type=int, default=100000)
    p.add_argument("--mixup_alpha", type=float, default=0.3)
    p.add_argument("--roll", action="store_true")
    p.add_argument("--wavmix", action="store_true")
    p.add_argument("--gain_augment", type=int, default=0)
    p.add_argument("--weight_decay", type=float, default=0)
    p.add_argument("--adamw", action="store_true")
    p.add_argument("--max_lr", type=float, default=0.0008)
    p.add_argument("--warm_up_len", type=int, default=8)
    p.add_argument("--ramp_down_start", type=int, default=80)
    p.add_argument("--ramp_down_len", type=int, default=95)
    p.add_argument("--last_lr_value", type=float, default=0.01)
    p.add_argument("--teacher_preds", default=os.path.join("resources", "passt_enemble_logits_mAP_495.npy"))
    p.add_argument("--fname_to_index", default=os.path.join("resources", "fname_to_index.pkl"))
    p.add_argument("--temperature", type=float, default=1)
    p.add_argument("--kd_lambda", type=float, default=0.1)
    p.add_argument("--freqm", type=int, default=0)
    p.add_argument("--timem", type=int, default=0)
    p.add_argument("--fmin", type=int, default=0)
    p.add_argument("--fmax", type=int, default=None)
    p.add_argument("--fmin_aug_range", type=int, default=10)
    p.add_argument("--fmax_aug_range", type=int, default=2000)
    p.add_argument("--seed", type=int, default=0)
    p.add_argument("--no_graph", action="store_true", help="eager KDTrainer instead of the captured step")
    p.add_argument("--transport", choices=["fp32", "int16"], default="fp32", help="waveform format over PCIe")
    p.add_argument("--max_steps", type=int, default=0, help="stop after this many steps (benchmarks / tests); 0 = whole epochs")
    p.add_argument("--optimizer", default="eat", choices=["eat", "torch"],
                   help="eat: efficientat_amd.optim.FusedAdam (one launch per step); torch: torch.optim.Adam / AdamW (fused=True)")
    p.add_argument("--precision", default=None, help="model.train_precision (auto / fp32 / bf16)")
    p.add_argument("--act_storage", default=None, choices=["fp32", "bf16"],
                   help="model.act_storage; bf16 (with --precision bf16) = the 16-bit surface of ex_pl_audioset.py:287-293 "
                        "(precision=16): wide activations / gradients of every block in bf16 in HBM")
    p.add_argument("--out", default=None, help="directory for rank 0's per-epoch state dicts")
    p.add_argument("--json", action="store_true", help="rank 0 prints one JSON line with the run's throughput at the end")
    p.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                   help="process-group backend: nccl = RCCL, one GPU per rank (production); gloo = several ranks may share a GPU "
                        "(tests on a one-GPU box: collectives on device tensors through the host, eager trainer only)")
    return p.parse_args(argv)


def _quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def build(args, dev):
    from .preprocess import AugmentMelSTFT
    width = args.model_width if args.model_width is not None else NAME_TO_WIDTH(args.model_name)
    if args.model_name.startswith("dymn"):
        from .dymn import get_model
    else:
        from .mn import get_model
    model = _quiet(get_model, width_mult=width).to(dev)
    if args.precision:
        model.train_precision = args.precision
    if args.act_storage:
        model.act_storage = args.act_storage
    mel = _quiet(AugmentMelSTFT, freqm=args.freqm, timem=args.timem, fmin=args.fmin, fmax=args.fmax,
                 fmin_aug_range=args.fmin_aug_range, fmax_aug_range=args.fmax_aug_range).to(dev)
    return model, mel


def main(argv=None):
    args = parse_args(argv)
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if not torch.cuda.is_available():
        raise SystemExit("efficientat_amd.train_dp needs a GPU per rank: the package has no CPU path")
    if args.backend == "gloo":
        local %= torch.cuda.device_count()                                    # ranks share the visible GPUs
        if not args.no_graph and world > 1:
            raise SystemExit("--backend gloo cannot be captured into a hipGraph (host-side collectives): add --no_graph")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "0")        # required to capture collectives in a graph
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.backend == "gloo":
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(here, "dropin"))                          # datasets.audioset (reference module path)
    from datasets import audioset

    from .dp import enable_data_parallel
    from .input_pipeline import DevicePrefetcher
    from .train_loop import GraphedKDTrainer, KDTrainer

    torch.manual_seed(args.seed)                                              # same initial weights on every rank ...
    np.random.seed(args.seed)
    model, mel = build(args, dev)
    if world > 1:
        enable_data_parallel(model)                                           # ... and rank 0's broadcast anyway, like DDP
    torch.manual_seed(args.seed + 1000 * (rank + 1))                          # per-rank augmentation draws from here on
    np.random.seed(args.seed + 1000 * (rank + 1))

    ds = _quiet(audioset.get_full_training_set, resample_rate=32000, roll=args.roll, wavmix=args.wavmix,
                gain_augment=args.gain_augment)
    if args.transport == "int16":
        from .input_pipeline import Int16Waveform
        ds = Int16Waveform(ds)                                               # converted in the DataLoader workers
    sampler = RankShardSampler(_quiet(audioset.get_ft_weighted_sampler, args.epoch_len), rank, world, seed=args.seed)
    dl = torch.utils.data.DataLoader(ds, sampler=sampler, batch_size=args.batch_size, num_workers=args.num_workers,
                                     drop_last=world > 1, persistent_workers=args.num_workers > 0)

    graphed = not args.no_graph
    lr = torch.tensor(args.max_lr, device=dev) if graphed else args.max_lr     # tensor lr: the schedule needs no re-capture
    if args.optimizer == "torch":
        opt_cls = torch.optim.AdamW if args.adamw else torch.optim.Adam
        opt = opt_cls(model.parameters(), lr=lr, weight_decay=args.weight_decay, fused=True, capturable=graphed)
    else:                        # the same update as one launch over every parameter (optim.py, eat_adam_multi)
        from .optim import FusedAdam
        opt = FusedAdam(model.parameters(), lr=lr, weight_decay=args.weight_decay, decoupled=args.adamw, capturable=graphed)
    sched = torch.optim.lr_scheduler.LambdaLR(
        opt, exp_warmup_linear_down(args.warm_up_len, args.ramp_down_len, args.ramp_down_start, args.last_lr_value))

    teacher = f2i = None
    if args.kd_lambda > 0 and os.path.isfile(args.teacher_preds) and os.path.isfile(args.fname_to_index):
        teacher = torch.from_numpy(np.load(args.teacher_preds)).float()
        with open(args.fname_to_index, "rb") as f:
            f2i = pickle.load(f)
    elif rank == 0:
        print(f"[train_dp] no teacher predictions at {args.teacher_preds}: hard-label BCE only", file=sys.stderr)

    model.train()
    mel.train()
    clip = ds[0][0].shape[-1]
    common = dict(teacher_preds=teacher, fname_to_index=f2i, kd_lambda=args.kd_lambda, temperature=args.temperature,
                  mixup_alpha=args.mixup_alpha)
    trainer = (GraphedKDTrainer(model, mel, opt, args.batch_size, clip, **common) if graphed
               else KDTrainer(model, mel, opt, **common))

    steps_total, clips_total, t_train = 0, 0, 0.0
    done = False
    for epoch in range(args.n_epochs):
        sampler.set_epoch(epoch)
        if hasattr(model, "update_params"):                                   # DyMN temperature (ex_audioset.py:131-133)
            model.update_params(epoch)
            if graphed and epoch > 0:
                trainer.recapture()                                           # temperatures are launch constants
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n_ep = 0
        for batch in DevicePrefetcher(dl, dev, transport=args.transport):
            x, names, y = batch[0], batch[1], batch[2]
            trainer.step(x, names, y)
            n_ep += 1
            clips_total += x.shape[0]
            if args.max_steps and steps_total + n_ep >= args.max_steps:
                done = True
                break
        sched.step()
        stats = trainer.epoch_stats()                                         # the one host sync of the epoch
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        t_train += dt
        steps_total += n_ep
        if rank == 0:
            print(f"[train_dp] epoch {epoch + 1}/{args.n_epochs}: {n_ep} steps, {n_ep * args.batch_size * world / dt:.0f} clips/s "
                  f"over {world} GPU(s), train_loss {stats['train_loss']:.5f} (label {stats['label_loss']:.5f}, "
                  f"kd {stats['distillation_loss']:.5f}), lr {float(sched.get_last_lr()[0]):.2e}", file=sys.stderr, flush=True)
            if args.out:
                os.makedirs(args.out, exist_ok=True)
                torch.save(model.state_dict(), os.path.join(args.out, f"{args.model_name}_epoch_{epoch}.pt"))
        if done:
            break
    if world > 1:
        t = torch.tensor([t_train], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        t_train = float(t.item())
    digest = float(torch.cat([p.detach().double().reshape(-1) for p in model.parameters()]).abs().sum())
    if world > 1:                                                             # replicas must end identical (DDP semantics)
        dg = torch.tensor([digest, -digest], device=dev, dtype=torch.float64)
        dist.all_reduce(dg, op=dist.ReduceOp.MAX)
        spread = float(dg[0] + dg[1])                                         # max - min over the ranks
    else:
        spread = 0.0
    if rank == 0 and args.json:
        print(json.dumps({"what": "efficientat_amd.train_dp", "model": args.model_name, "n_gpus": world, "steps": steps_total,
                          "param_abs_sum": digest, "param_abs_sum_spread_over_ranks": spread, "backend": args.backend if world > 1 else None,
                          "batch_per_gpu": args.batch_size, "clips_per_s": round(clips_total * world / max(t_train, 1e-9), 1),
                          "launch": "hipGraph replay" if graphed else "eager", "transport": args.transport,
                          "final": stats}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
