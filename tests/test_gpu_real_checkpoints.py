"""Real-weight pins ("with real weights: parity unpinned in this container").  The released checkpoints
(GitHub Releases of fschmid56/EfficientAT: mn10_as_mAP_471.pt, dymn10_as.pt, ...) cannot be fetched here or on the GPU box
(no network), so these tests RUN WHEN THE FILES EXIST and skip otherwise: put the `.pt` files (and, for the known-answer test,
the reference's `resources/metro_station-paris.wav`) under `<repo>/resources/` or point EAT_CHECKPOINT_DIR at them.

  * strict `load_state_dict` of a released checkpoint into the mirror built by `get_model` - the pretrained-weight layout
    contract of models/mn/model.py:282-310 / models/dymn/model.py:257-281;
  * the README's known answer (README.md:126-146): `inference.py --model_name=dymn10_as` on metro_station-paris.wav prints
    Train 0.747, Subway 0.599, Rail transport 0.493, Railroad car 0.445, Vehicle 0.360, ... - reproduced on the HIP path
    within 5e-3 on every listed probability (the reference's GPU path runs fp16 autocast, inference.py:51-53)."""
import contextlib
import csv
import io
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
if not torch.cuda.is_available():
    pytest.skip("no GPU", allow_module_level=True)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CKPT_DIR = os.environ.get("EAT_CHECKPOINT_DIR", os.path.join(ROOT, "resources"))
DEV = torch.device("cuda:0")

MN = {"mn10_as_mAP_471.pt": 1.0, "mn04_as_mAP_432.pt": 0.4, "mn20_as_mAP_478.pt": 2.0, "mn40_as_mAP_484.pt": 4.0}
DYMN = {"dymn10_as.pt": 1.0, "dymn04_as.pt": 0.4, "dymn20_as_mAP_493.pt": 2.0}
README_TOP10 = [("Train", 0.747), ("Subway, metro, underground", 0.599), ("Rail transport", 0.493),
                ("Railroad car, train wagon", 0.445), ("Vehicle", 0.360), ("Clickety-clack", 0.105), ("Speech", 0.053),
                ("Sliding door", 0.036), ("Outside, urban or manmade", 0.035), ("Music", 0.017)]


def _have(name):
    p = os.path.join(CKPT_DIR, name)
    if not os.path.isfile(p):
        pytest.skip(f"{name} is not under {CKPT_DIR} (released checkpoint; no network here)")
    return p


def _build(kind, width):
    if kind == "mn":
        from efficientat_amd.mn import get_model
    else:
        from efficientat_amd.dymn import get_model
    with contextlib.redirect_stdout(io.StringIO()):
        return get_model(width_mult=width)


@pytest.mark.parametrize("name", list(MN) + list(DYMN))
def test_released_checkpoint_loads_strictly(name):
    path = _have(name)
    kind, width = ("mn", MN[name]) if name in MN else ("dymn", DYMN[name])
    model = _build(kind, width)
    sd = torch.load(path, map_location="cpu")
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not missing and not unexpected, (missing[:5], unexpected[:5])
    model.load_state_dict(sd, strict=True)
    model.to(DEV).eval()
    with torch.no_grad():
        logits, _ = model(torch.randn(2, 1, 128, 1000, device=DEV))
    assert torch.isfinite(logits).all() and logits.shape == (2, 527)


def test_readme_known_answer_dymn10_metro_station():
    path = _have("dymn10_as.pt")
    wav = os.path.join(CKPT_DIR, "metro_station-paris.wav")
    labels_csv = os.environ.get("EAT_LABELS_CSV", os.path.join(CKPT_DIR, "class_labels_indices.csv"))
    if not os.path.isfile(wav) or not os.path.isfile(labels_csv):
        pytest.skip("metro_station-paris.wav / class_labels_indices.csv (reference resources/ and metadata/) are not staged")
    from efficientat_amd.audio_io import load_audio
    from efficientat_amd.preprocess import AugmentMelSTFT
    with open(labels_csv) as f:
        labels = [r[2] for r in list(csv.reader(f))[1:]]
    model = _build("dymn", 1.0)
    model.load_state_dict(torch.load(path, map_location="cpu"), strict=True)
    model.to(DEV).eval()
    with contextlib.redirect_stdout(io.StringIO()):
        mel = AugmentMelSTFT(n_mels=128, sr=32000, win_length=800, hopsize=320).to(DEV).eval()
    wave, _ = load_audio(wav, sr=32000, mono=True)                       # inference.py:45
    x = torch.from_numpy(wave[None, :]).to(DEV)
    with torch.no_grad():
        probs = torch.sigmoid(model(mel(x).unsqueeze(1))[0].float())[0].cpu().numpy()     # inference.py:50-56
    order = np.argsort(probs)[::-1][:10]
    got = [(labels[i], float(probs[i])) for i in order]
    print(got)
    for (name, p), (gname, gp) in zip(README_TOP10, got):
        assert name == gname and abs(p - gp) < 5e-3, (README_TOP10, got)
