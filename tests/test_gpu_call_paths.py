"""The reference's call paths on the HIP backend, self-contained (nothing under /root/reference is read; runs on the
driver's GPU box): the fp16-`autocast` inference sequence (inference.py:27-56), windowed inference
(windowed_inference.py:88-113), evaluate() (ex_audioset.py:259-321) and one KD training epoch (ex_audioset.py:123-220) -
restated in tests/callpaths/driver.py against the drop-in modules `models.*`, `helpers.*`, `datasets.audioset` (dropin/)
and the `librosa` / `wandb` stand-ins (tests/standins/), run in a scratch working directory laid out like the reference's
(./metadata, ./resources with a synthetic checkpoint under the released file name, teacher logits, file-name index).
Every section is checked against the CPU oracle on the same weights and inputs.  (tests/test_gpu_reference_scripts.py runs
the reference's own unmodified files the same way wherever they are available.)"""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
if not torch.cuda.is_available():
    pytest.skip("no GPU", allow_module_level=True)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import run_reference_scripts as R  # noqa: E402
from oracle import eat_oracle as O  # noqa: E402


@pytest.fixture(scope="module")
def callpaths():
    work = tempfile.mkdtemp(prefix="eat_callpaths_")
    env = R.build_workdir(work, n_train=48, n_test=527)      # >= 527 test clips: every class has a positive (per-class ROC)
    driver = os.path.join(ROOT, "tests", "callpaths", "driver.py")
    boot = ("import sys, runpy; sys.path[:0] = %r; sys.argv = %r; runpy.run_path(%r, run_name='__main__')"
            % ([os.path.join(ROOT, "tests", "standins"), os.path.join(ROOT, "dropin"), ROOT],
               [driver, "inference", "windowed", "evaluate", "kd_epoch"], driver))
    e = dict(os.environ, **env)
    e.pop("PYTHONPATH", None)
    p = subprocess.run([sys.executable, "-c", boot], cwd=work, env=e, capture_output=True, text=True, timeout=1200)
    assert p.returncode == 0, p.stderr[-4000:]
    res = {}
    for ln in p.stdout.splitlines():
        if ln.startswith("CALLPATH "):
            d = json.loads(ln[len("CALLPATH "):])
            res[d["section"]] = d
    from efficientat_amd.mn import _CKPT
    sd = torch.load(os.path.join(work, "resources", _CKPT["mn10_as"]))
    return work, res, {k: v.cpu() for k, v in sd.items()}


def test_inference_call_path_under_autocast(callpaths):
    work, res, sd = callpaths
    r = res["inference"]
    assert r["features_shape"] == [1, 960] and r["n_samples"] == 320000            # 10 s, resampled 44.1 -> 32 kHz
    probs = [p for _, p in r["top10"]]
    assert probs == sorted(probs, reverse=True) and all(0.0 <= p <= 1.0 for p in probs)
    # fp16 autocast does not touch our launchers (fp32 tensors in, fp32 out): the two passes differ by atomics order only
    assert r["autocast_vs_plain"] < 1e-4
    # the same clip through the CPU oracle (same stand-in loader, same checkpoint)
    from efficientat_amd.audio_io import load_audio
    wav, _ = load_audio(os.path.join(work, "resources", "synthetic_clip.wav"), sr=32000)
    with torch.no_grad():
        ref, _ = O.mn_forward(sd, O.mel_forward(torch.from_numpy(wav[None])).unsqueeze(1))
    assert float((torch.tensor(r["logits"]) - ref[0]).abs().max()) < 1e-3
    assert len(res["windowed"]["windows"]) == 3 and all(0.0 <= w["p"] <= 1.0 for w in res["windowed"]["windows"])


def test_evaluate_call_path(callpaths):
    work, res, sd = callpaths
    r = res["evaluate"]
    assert r["n"] == 527 and 0.0 <= r["mAP"] <= 1.0 and 0.0 <= r["ROC"] <= 1.0
    # first batch through the CPU oracle
    sys.path[:0] = [os.path.join(ROOT, "dropin")]
    cwd = os.getcwd()
    os.chdir(work)
    try:
        os.environ["EAT_SYNTH_AUDIOSET"], os.environ["EAT_SYNTH_AUDIOSET_TEST"] = "1", "527"
        from datasets.audioset import get_test_set
        ds = get_test_set(resample_rate=32000)
        x = torch.stack([torch.as_tensor(ds[i][0]) for i in range(4)])
    finally:
        os.chdir(cwd)
    with torch.no_grad():
        ref, _ = O.mn_forward(sd, O.mel_forward(x.reshape(4, -1)).unsqueeze(1))
    got = np.load(os.path.join(work, "eval_outputs.npy"))[:4]
    assert np.abs(got - ref.numpy()).max() < 1e-3


def test_kd_epoch_call_path(callpaths):
    work, res, sd = callpaths
    r = res["kd_epoch"]
    assert len(r["losses"]) == 4 and all(np.isfinite(r["losses"])) and r["num_batches_tracked"] == 4
    assert r["unknown_files"] >= 0 and r["lr"] > 0
    # first step on the CPU oracle: same mixed log-mel, mixed targets, teacher rows and dropout mask; train-mode BatchNorm
    first = torch.load(os.path.join(work, "kd_first_step.pt"), weights_only=False)
    st = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and not k.endswith(("running_mean", "running_var")) else v.clone())
          for k, v in first["state"].items()}
    logits, _ = O.mn_forward(st, first["x"], train=True, stats={}, drop_mask=first["keep"])
    lam, perm, y_soft = first["lam"], first["perm"], first["y_soft"]
    label = F.binary_cross_entropy_with_logits(logits, first["y_mix"], reduction="none").mean()
    bce = torch.nn.BCEWithLogitsLoss(reduction="none")
    soft = bce(logits, y_soft).mean(1) * lam + bce(logits, y_soft[perm]).mean(1) * (1 - lam)
    soft = torch.where(first["unknown"], torch.zeros_like(soft), soft)
    loss = 0.1 * label + 0.9 * soft.mean()
    loss.backward()
    assert abs(r["losses"][0] - float(loss)) < 1e-4 * max(1.0, abs(float(loss))), (r["losses"][0], float(loss))
    gmax = max(float(v.grad.norm()) for v in st.values() if getattr(v, "grad", None) is not None)
    rels = []
    for name, got in first["gnorm"].items():
        ref = float(st[name].grad.norm())
        if ref > 1e-4 * gmax:
            rels.append(abs(got - ref) / ref)
    assert len(rels) > 100 and max(rels) < 5e-2 and float(np.median(rels)) < 1e-2, (max(rels), float(np.median(rels)))
