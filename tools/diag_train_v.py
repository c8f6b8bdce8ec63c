"""GPU diagnostic: run-to-run and plan-to-plan (EAT_TRAIN_V 1 vs 2) agreement of the MN training step's gradients.
python tools/diag_train_v.py [tiny|full]"""
import contextlib, io, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from efficientat_amd import mn_train
from efficientat_amd.mn import get_model
DEV = torch.device("cuda:0")
mode = sys.argv[1] if len(sys.argv) > 1 else "tiny"
torch.manual_seed(0)
if mode == "tiny":
    wm, nc, shape = 0.4, 10, (4, 1, 128, 200)
else:
    wm, nc, shape = 1.0, 527, (16, 1, 128, 1000)
x = torch.randn(*shape, device=DEV)
y = (torch.rand(shape[0], nc, device=DEV) < 0.3).float()


def make():
    torch.manual_seed(1)
    with contextlib.redirect_stdout(io.StringIO()):
        m = get_model(width_mult=wm, num_classes=nc).to(DEV)
    with torch.no_grad():
        for mod in m.modules():
            if isinstance(mod, torch.nn.Conv2d):
                fan_in = mod.weight.shape[1] * mod.weight.shape[2] * mod.weight.shape[3]
                mod.weight.normal_(0, (2.0 / fan_in) ** 0.5)
    m.train()
    m._drop_mask_override = torch.ones(shape[0], m.classifier[2].out_features, device=DEV)
    return m


def grads(v):
    mn_train._TRAIN_V = v
    m = make()
    F.binary_cross_entropy_with_logits(m(x)[0], y).backward()
    torch.cuda.synchronize()
    return {n: p.grad.clone() for n, p in m.named_parameters()}


def cmp(a, b, tag):
    gmax = max(float(v.norm()) for v in b.values())
    worst = []
    for n in a:
        if float(b[n].norm()) < 1e-4 * gmax:        # zero-gradient project-BN biases: round-off only
            continue
        worst.append((float((a[n] - b[n]).norm() / b[n].norm()), n))
    worst.sort(reverse=True)
    med = worst[len(worst) // 2][0]
    print(tag, f"rel-L2 median {med:.2e} worst:", [(f"{d:.2e}", n) for d, n in worst[:4]], flush=True)


g1a, g1b = grads(1), grads(1)
cmp(g1a, g1b, "V1 run-to-run")
g2a, g2b, g2c = grads(2), grads(2), grads(2)
cmp(g2a, g2b, "V2 run-to-run (1st vs 2nd)")
cmp(g2b, g2c, "V2 run-to-run (2nd vs 3rd)")
cmp(g2a, g1a, "V2 vs V1")
cmp(g2c, g1a, "V2(3rd) vs V1")
