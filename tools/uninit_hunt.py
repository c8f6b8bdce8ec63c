"""torch.empty filled with NaN (deterministic-mode debugging aid): does any kernel of a training step read uninitialised memory?
python tools/uninit_hunt.py <model> [B]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
torch.use_deterministic_algorithms(True, warn_only=True)
torch.utils.deterministic.fill_uninitialized_memory = True
import bench
from efficientat_amd import _lib
name = sys.argv[1] if len(sys.argv) > 1 else "dymn20_bf16"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
dev = torch.device("cuda:0")
mel, _ = bench.build_model(dev)
model = bench.make_train_model(name, dev)
wave = (0.1 * torch.randn(B, 320000, device=dev)).clamp_(-1, 1)
y = (torch.rand(B, 527, device=dev) < 0.005).float()
opt = bench._adam(model.parameters(), capturable=False)
model.train(); mel.train()
# trace every library call: first call after which one of its float output tensors... we cannot see outputs generically, so just
# run the step and report where non-finite values sit
for step in range(3):
    opt.zero_grad(set_to_none=True)
    logits, _ = model(mel(wave).unsqueeze(1))
    loss = F.binary_cross_entropy_with_logits(logits, y)
    loss.backward()
    nf_g = [n for n, p in model.named_parameters() if p.grad is not None and not bool(torch.isfinite(p.grad).all())]
    opt.step()
    nf_p = [n for n, p in model.named_parameters() if not bool(torch.isfinite(p).all())]
    print(f"step {step}: loss {float(loss):.5f} logits finite {bool(torch.isfinite(logits).all())}; non-finite grads {len(nf_g)} {nf_g[:8]}; params {len(nf_p)} {nf_p[:4]}", flush=True)

# ---- the same inside the captured step (fills are captured too: every replay re-poisons what torch.empty hands out)
from efficientat_amd.graphs import GraphedTrainStep
model2 = bench.make_train_model(name, dev)
model2.train()
opt2 = bench._adam(model2.parameters(), capturable=True)
gstep = GraphedTrainStep(model2, opt2, F.binary_cross_entropy_with_logits, mel(wave).unsqueeze(1), y)
for s in range(6):
    loss = gstep(mel(wave, out=gstep.x).view_as(gstep.x), gstep.y)
    nf_p = [n for n, p in model2.named_parameters() if not bool(torch.isfinite(p).all())]
    nf_g = [(n, int((~torch.isfinite(p.grad)).sum()), p.grad.numel()) for n, p in model2.named_parameters() if p.grad is not None and not bool(torch.isfinite(p.grad).all())]
    print(f"graph step {s}: loss {float(loss):.5f}; non-finite params {len(nf_p)} {nf_p[:6]}; grads {nf_g[:10]}", flush=True)
    if nf_p:
        break
