"""Body of tests/test_gpu_train.py::test_mn_backward_through_bucketed_rccl_reducer_matches_local, run as a separate
process: `python tests/rccl_reducer_case.py` prints RCCL_REDUCER_OK when every check passed.  (Its own process because a
process-group failure aborts the interpreter: until round 4 the ProcessGroupNCCL watchdog thread's event polls could land
inside the hipGraph capture of the step - "operation not permitted when stream is capturing" - and terminate the process;
graphs.GraphedTrainStep captures with capture_error_mode="thread_local" now.)"""
import contextlib
import io
import os
import socket
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
import torch.nn.functional as F

from efficientat_amd.dp import enable_data_parallel
from efficientat_amd.graphs import GraphedTrainStep
from efficientat_amd.mn import get_model

DEV = torch.device("cuda:0")


def _rand(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


def _quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def main():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "0")
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=torch.device("cuda:0"))
    torch.manual_seed(0)
    x = _rand(4, 1, 128, 200, seed=3).to(DEV)
    y = (torch.rand(4, 10, generator=torch.Generator().manual_seed(1)) < 0.3).float().to(DEV)
    keep = torch.ones(4, 512, device=DEV)        # on the device: a host->device copy cannot be captured

    def make():
        torch.manual_seed(1)
        m = _quiet(get_model, width_mult=0.4, num_classes=10).to(DEV)
        with torch.no_grad():
            for mod in m.modules():
                if isinstance(mod, torch.nn.Conv2d):
                    fan_in = mod.weight.shape[1] * mod.weight.shape[2] * mod.weight.shape[3]
                    mod.weight.normal_(0, (2.0 / fan_in) ** 0.5)
        m.train()
        m._drop_mask_override = keep
        return m

    ref = make()
    F.binary_cross_entropy_with_logits(ref(x)[0], y).backward()
    dp = make()
    dp.load_state_dict(ref.state_dict())
    for b_ref, b_dp in zip(ref.buffers(), dp.buffers()):      # the forward above already moved ref's running stats
        pass
    enable_data_parallel(dp, bucket_bytes=64 << 10, force_buckets=True)
    assert dp._grad_reducer.bucketed and dp._grad_reducer.world == 1
    F.binary_cross_entropy_with_logits(dp(x)[0], y).backward()
    # per tensor, relative to max(|tensor|, 1e-3 * the largest gradient of the network): the project-BN biases have a TRUE
    # gradient of exactly zero (a per-channel shift in front of a train-mode BatchNorm cancels), what is computed for them
    # is round-off of the weight-gradient atomics and differs from run to run
    gmax = max(float(q.grad.abs().max()) for q in ref.parameters())

    def worst_of(model):
        return max(float((p.grad - q.grad).abs().max()) / max(float(q.grad.abs().max()), 1e-3 * gmax)
                   for p, q in zip(model.parameters(), ref.parameters()))
    worst = worst_of(dp)
    assert worst < 2e-3, worst           # atomics in the weight-gradient / Gram-matrix kernels: not bit-identical between runs
                                         # (4 clips: a few 10^4 elements per BatchNorm channel amplify the round-off)
    # the same step captured into a hipGraph with the collectives inside
    g = make()
    g.load_state_dict(ref.state_dict())
    enable_data_parallel(g, bucket_bytes=64 << 10, force_buckets=True)
    opt = torch.optim.SGD(g.parameters(), lr=0.0)              # lr 0: the replay leaves weights, keeps .grad
    step = GraphedTrainStep(g, opt, F.binary_cross_entropy_with_logits, x, y, warmup=2)
    loss = step(x, y)
    torch.cuda.synchronize()
    assert torch.isfinite(loss)
    worst = worst_of(g)
    assert worst < 2e-3, worst
    print("RCCL_REDUCER_OK", flush=True)
    torch.cuda.synchronize()
    dist.destroy_process_group()


if __name__ == "__main__":
    try:
        main()
    except BaseException:            # the parent test shows stdout: RCCL's teardown warnings flood stderr
        import traceback
        print("RCCL_REDUCER_FAILED\n" + traceback.format_exc(), flush=True)
        raise
