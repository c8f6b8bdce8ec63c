"""Call sequences of the reference's entry scripts, restated against the drop-in modules (no reference file is read):

  inference         inference.py:27-56            get_model(pretrained_name=...) -> AugmentMelSTFT -> librosa.core.load ->
                                                  fp16 `autocast` around mel() and model() -> sigmoid -> top-10
  windowed          windowed_inference.py:88-113  the same per window of a padded waveform
  evaluate          ex_audioset.py:259-321        DataLoader(get_test_set) -> autocast + no_grad -> _mel_forward -> model ->
                                                  sklearn mAP / ROC
  kd_epoch          ex_audioset.py:123-220        mel.train(), model.train(), mixup on the log-mel, BCE + KD loss with the
                                                  teacher gather and the unknown-file mask, backward, Adam, LambdaLR

Run as a subprocess with the scratch directory of tools/run_reference_scripts.build_workdir as CWD (helpers.utils reads
./metadata/class_labels_indices.csv at import, the factories look for checkpoints under ./resources - both exactly as
in the reference) and sys.path = [tests/standins, dropin, repo root].  Prints one JSON line per section."""
import json
import os
import pickle
import sys
from contextlib import nullcontext

import numpy as np
import torch
import torch.nn.functional as F
from torch import autocast
from torch.utils.data import DataLoader

import librosa                                                    # tests/standins (backed by efficientat_amd.audio_io)
from datasets.audioset import get_test_set, get_full_training_set     # dropin: synthetic AudioSet, reference tuple layout
from helpers.init import worker_init_fn
from helpers.utils import NAME_TO_WIDTH, exp_warmup_linear_down, labels, mixup
from models.mn.model import get_model as get_mobilenet
from models.preprocess import AugmentMelSTFT

DEV = torch.device("cuda")


def _mel_forward(x, mel):                                         # ex_audioset.py:223-228
    old_shape = x.size()
    x = mel(x.reshape(-1, old_shape[2]))
    return x.reshape(old_shape[0], old_shape[1], x.shape[1], x.shape[2])


def _model_and_mel(name="mn10_as"):
    model = get_mobilenet(width_mult=NAME_TO_WIDTH(name), pretrained_name=name)      # loads ./resources/<released name>.pt
    model.to(DEV).eval()
    mel = AugmentMelSTFT(n_mels=128, sr=32000, win_length=800, hopsize=320)
    mel.to(DEV).eval()
    return model, mel


def inference(audio_path):
    model, mel = _model_and_mel()
    waveform, _ = librosa.core.load(audio_path, sr=32000, mono=True)
    waveform = torch.from_numpy(waveform[None, :]).to(DEV)
    with torch.no_grad(), autocast(device_type=DEV.type):
        spec = mel(waveform)
        preds, features = model(spec.unsqueeze(0))
    with torch.no_grad():                                         # the same without autocast: the launchers take fp32 as it is
        preds32, _ = model(mel(waveform).unsqueeze(0))
    p = torch.sigmoid(preds.float()).squeeze().cpu().numpy()
    order = np.argsort(p)[::-1]
    return {"section": "inference", "top10": [[labels[i], float(p[i])] for i in order[:10]],
            "logits": preds.float().squeeze().cpu().tolist(), "features_shape": list(features.shape),
            "autocast_vs_plain": float((preds.float() - preds32).abs().max()),
            "waveform": waveform.squeeze().cpu().tolist()[::97], "n_samples": int(waveform.shape[1])}


def windowed(audio_path, window_s=4.0, hop_s=3.0):
    model, mel = _model_and_mel()
    waveform, _ = librosa.core.load(audio_path, sr=32000, mono=True)
    waveform = torch.from_numpy(waveform[None, :]).to(DEV)
    win, hop = int(window_s * 32000), int(hop_s * 32000)
    n_windows = int(np.ceil((waveform.shape[1] - win) / hop)) + 1
    waveform = F.pad(waveform, (0, n_windows * hop + win - waveform.shape[1]))
    out = []
    with torch.no_grad(), autocast(device_type=DEV.type):
        for i in range(n_windows):
            spec = mel(waveform[:, i * hop:i * hop + win])
            preds, _ = model(spec.unsqueeze(0))
            p = torch.sigmoid(preds.float()).squeeze().cpu().numpy()
            out.append({"start": i * hop / 32000, "end": (i * hop + win) / 32000, "top": int(np.argmax(p)), "p": float(p.max())})
    return {"section": "windowed", "windows": out}


def evaluate(batch_size=31):
    from sklearn import metrics
    model, mel = _model_and_mel()
    dl = DataLoader(dataset=get_test_set(resample_rate=32000), worker_init_fn=worker_init_fn, num_workers=0, batch_size=batch_size)
    targets, outputs = [], []
    for x, _, y in dl:
        x = x.to(DEV)
        with autocast(device_type=DEV.type):
            with torch.no_grad():
                y_hat, _ = model(_mel_forward(x, mel))
        targets.append(y.numpy())
        outputs.append(y_hat.float().cpu().numpy())
    targets, outputs = np.concatenate(targets), np.concatenate(outputs)
    mAP = metrics.average_precision_score(targets, outputs, average=None)
    ROC = metrics.roc_auc_score(targets, outputs, average=None)
    np.save("eval_outputs.npy", outputs)
    return {"section": "evaluate", "n": int(len(outputs)), "mAP": float(mAP.mean()), "ROC": float(ROC.mean())}


def kd_epoch(batch_size=8, epoch_len=32, kd_lambda=0.1, mixup_alpha=0.3, lr=8e-4):
    torch.manual_seed(0)
    np.random.seed(0)
    model, mel = _model_and_mel()
    teacher_preds = torch.from_numpy(np.load("resources/passt_enemble_logits_mAP_495.npy")).float()
    teacher_preds = torch.sigmoid(teacher_preds / 1.0)            # ex_audioset.py:77-80 (temperature 1)
    with open("resources/fname_to_index.pkl", "rb") as fh:
        fname_to_index = pickle.load(fh)
    ds = torch.utils.data.Subset(get_full_training_set(add_index=True, resample_rate=32000, roll=False, wavmix=False, gain_augment=0),
                                 list(range(epoch_len)))
    dl = DataLoader(dataset=ds, worker_init_fn=worker_init_fn, num_workers=0, batch_size=batch_size, shuffle=False)
    optimizer = torch.optim.Adam(model.parameters(), lr=lr, weight_decay=0.0)
    scheduler = torch.optim.lr_scheduler.LambdaLR(optimizer, exp_warmup_linear_down(8, 95, 105, 0.01))
    distillation_loss = torch.nn.BCEWithLogitsLoss(reduction="none")
    mel.train()
    model.train()
    stats, first = [], None
    # test hook (not part of the reference's loop): the FIRST step's dropout mask is drawn on the CPU and handed to the
    # model, so that the CPU oracle can replay the step exactly; later steps draw their masks on the device as usual
    keep = (torch.rand(batch_size, model.classifier[2].out_features, generator=torch.Generator().manual_seed(11))
            < 1.0 - model.classifier[4].p).float()
    model._drop_mask_override = keep
    for x, f, y, i in dl:
        bs = x.size(0)
        x, y = x.to(DEV), y.to(DEV)
        x = _mel_forward(x, mel)
        rn_indices, lam = mixup(bs, mixup_alpha)
        lam = lam.to(x.device)
        x = x * lam.reshape(bs, 1, 1, 1) + x[rn_indices] * (1. - lam.reshape(bs, 1, 1, 1))
        if first is None:
            state0 = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
        y_hat, _ = model(x)
        model._drop_mask_override = None
        y_mix = y * lam.reshape(bs, 1) + y[rn_indices] * (1. - lam.reshape(bs, 1))
        label_loss = F.binary_cross_entropy_with_logits(y_hat, y_mix, reduction="none").mean()
        indices = torch.tensor([fname_to_index[n] if n in fname_to_index else -1 for n in f], dtype=torch.int64)
        unknown = indices == -1
        y_soft = teacher_preds[indices].to(y_hat.device).type_as(y_hat)
        soft = distillation_loss(y_hat, y_soft).mean(dim=1) * lam.reshape(bs) + \
            distillation_loss(y_hat, y_soft[rn_indices]).mean(dim=1) * (1. - lam.reshape(bs))
        soft[unknown] = soft[unknown] * 0
        loss = kd_lambda * label_loss + (1 - kd_lambda) * soft.mean()
        if first is None:       # everything the CPU oracle needs to recompute this step's loss and gradient norms
            first = dict(x=x.detach().cpu(), y_mix=y_mix.cpu(), y_soft=y_soft.cpu(), perm=rn_indices, lam=lam.cpu(), unknown=unknown,
                         keep=keep, state=state0)
        stats.append(float(loss.detach().cpu()))
        loss.backward()
        if len(stats) == 1:
            first["gnorm"] = {n: float(p.grad.norm()) for n, p in model.named_parameters()}
        optimizer.step()
        optimizer.zero_grad()
    scheduler.step()
    torch.save(first, "kd_first_step.pt")
    torch.save(model.state_dict(), "kd_epoch_state.pt")
    reloaded = get_mobilenet(width_mult=1.0)
    reloaded.load_state_dict(torch.load("kd_epoch_state.pt"), strict=True)
    return {"section": "kd_epoch", "losses": stats, "unknown_files": int(unknown.sum()), "lr": scheduler.get_last_lr()[0],
            "num_batches_tracked": int(model.state_dict()["features.0.1.num_batches_tracked"])}


if __name__ == "__main__":
    for sec in sys.argv[1:]:
        if sec == "inference":
            res = inference("resources/synthetic_clip.wav")
        elif sec == "windowed":
            res = windowed("resources/synthetic_clip.wav")
        elif sec == "evaluate":
            res = evaluate()
        elif sec == "kd_epoch":
            res = kd_epoch()
        else:
            raise SystemExit(f"unknown section {sec}")
        print("CALLPATH " + json.dumps(res), flush=True)
