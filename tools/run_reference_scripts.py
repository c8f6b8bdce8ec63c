"""Run the REFERENCE's own scripts, unmodified, on the HIP path ("drops into ex_audioset.py unchanged").

    python tools/run_reference_scripts.py --ref /root/reference [--out profiles/r2_reference_scripts.log]

`--ref` is a directory holding the reference's `inference.py` / `ex_audioset.py` (they are executed with runpy, never
copied or edited).  Everything those scripts import resolves to THIS repo: `models.*`, `helpers.*`, `datasets.audioset`
-> dropin/ (HIP-backed mirrors + the synthetic AudioSet), `wandb`, `librosa` -> tests/standins/ (test-only stubs).  The
working directory is a scratch folder with what the scripts expect relative to the CWD: metadata/class_labels_indices.csv
(synthetic label names), resources/<checkpoint>.pt (a synthetic state_dict under the released file name: torch.hub then
uses the cached file, no network), resources/<clip>.wav (44.1 kHz PCM, so the resampler runs), the KD teacher logits and
the file-name index.  Needs a GPU: the product has no CPU path.

Round 5: `--audio <wav>` runs inference.py / windowed_inference.py on a given recording (the reference's own
resources/metro_station-paris.wav: BASELINE configs[0]'s file) instead of the synthetic clip, and the log starts with the
commit it was asked to record (`--commit`, the GPU box has no .git) and a sha256 over the product tree it actually ran
(`tree_sha`: every file under efficientat_amd/ except build products, dropin/, include/).  `--verify <log>` recomputes
that hash on the tree it is run in and fails when it differs: the log of record must come from the committed state.
"""
import argparse
import csv
import os
import pickle
import re
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def tree_sha():
    """sha256 over the product sources (paths + contents, sorted): what a run of the reference scripts exercises."""
    import hashlib
    h = hashlib.sha256()
    files = []
    for top in ("efficientat_amd", "dropin", "include"):
        for d, _, fs in os.walk(os.path.join(ROOT, top)):
            if "__pycache__" in d:
                continue
            files += [os.path.join(d, f) for f in fs if f.endswith((".py", ".hip", ".h", ".cpp"))]
    for f in sorted(files):
        h.update(os.path.relpath(f, ROOT).encode() + b"\0")
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def build_workdir(work, n_train=48, n_test=527):
    import torch
    sys.path[:0] = [os.path.join(ROOT, "dropin"), ROOT]
    os.makedirs(os.path.join(work, "metadata"), exist_ok=True)
    os.makedirs(os.path.join(work, "resources"), exist_ok=True)
    with open(os.path.join(work, "metadata", "class_labels_indices.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["index", "mid", "display_name"])
        for i in range(527):
            w.writerow([i, "/m/syn%03d" % i, "synthetic class %03d" % i])
    cwd = os.getcwd()
    os.chdir(work)                                   # helpers.utils reads the csv relative to the CWD at import
    try:
        import contextlib
        import io
        from efficientat_amd.mn import get_model, _CKPT
        torch.manual_seed(0)
        with contextlib.redirect_stdout(io.StringIO()):
            model = get_model(width_mult=1.0)
        with torch.no_grad():
            for m in model.modules():
                if isinstance(m, torch.nn.Conv2d):
                    fan_in = m.weight.shape[1] * m.weight.shape[2] * m.weight.shape[3]
                    m.weight.normal_(0, (2.0 / fan_in) ** 0.5)
                elif isinstance(m, torch.nn.Linear):
                    m.weight.normal_(0, (1.0 / m.weight.shape[1]) ** 0.5)
            model.classifier[5].weight.mul_(0.05)        # keep the sigmoid outputs away from saturation
            model.classifier[5].bias.fill_(-2.0)
        torch.save(model.state_dict(), os.path.join(work, "resources", _CKPT["mn10_as"]))
        os.environ["EAT_SYNTH_AUDIOSET"] = "1"                        # explicit opt-in to the synthetic AudioSet stand-in
        os.environ["EAT_SYNTH_AUDIOSET_TRAIN"], os.environ["EAT_SYNTH_AUDIOSET_TEST"] = str(n_train), str(n_test)
        from datasets.audioset import dataset_config, synth_name
    finally:
        os.chdir(cwd)
    # 10 s, 44.1 kHz, 16-bit stereo clip: tones + noise (librosa.load -> mono, 32 kHz)
    from scipy.io import wavfile
    rng = np.random.default_rng(0)
    t = np.arange(441000) / 44100.0
    mono = 0.2 * np.sin(2 * np.pi * 440 * t) + 0.1 * np.sin(2 * np.pi * 3000 * t) + 0.05 * rng.standard_normal(t.size)
    wavfile.write(os.path.join(work, "resources", "synthetic_clip.wav"), 44100,
                  (np.stack([mono, 0.5 * mono], axis=1) * 32767).astype(np.int16))
    # KD teacher: logits N(-5, 2^2) for 90 % of the training files (the rest exercise the "unknown file" path)
    names = [synth_name(off + i) for _, n, off in (dataset_config["balanced_train_hdf5"], dataset_config["unbalanced_train_hdf5"])
             for i in range(n)]
    known = [n for i, n in enumerate(names) if i % 10 != 9]
    np.save(os.path.join(work, "resources", "passt_enemble_logits_mAP_495.npy"),
            (rng.standard_normal((len(known), 527)) * 2.0 - 5.0).astype(np.float32))
    with open(os.path.join(work, "resources", "fname_to_index.pkl"), "wb") as f:
        pickle.dump({n: i for i, n in enumerate(known)}, f)
    return dict(EAT_SYNTH_AUDIOSET="1", EAT_SYNTH_AUDIOSET_TRAIN=str(n_train), EAT_SYNTH_AUDIOSET_TEST=str(n_test))


def _drop_repr(text):
    """The reference's get_model prints the whole module tree: keep the log readable (one line instead of ~350)."""
    out, skip = [], False
    for line in text.splitlines():
        if not skip and re.match(r"^(MN|DyMN)\($", line):
            skip = True
            out.append(line + " ... module tree (printed by the reference's get_model) elided ... )")
        elif skip:
            skip = line != ")"
        else:
            out.append(line)
    return "\n".join(out)


def run_script(ref, script, argv, work, extra_env=None, timeout=900):
    path = os.path.join(ref, script)
    boot = ("import sys, runpy; sys.path[:0] = %r; sys.argv = %r; runpy.run_path(%r, run_name='__main__')"
            % ([os.path.join(ROOT, "tests", "standins"), os.path.join(ROOT, "dropin"), ROOT], [path] + argv, path))
    env = dict(os.environ, WANDB_STANDIN_DIR=os.path.join(work, "wandb_run"), **(extra_env or {}))
    env.pop("PYTHONPATH", None)
    p = subprocess.run([sys.executable, "-c", boot], cwd=work, env=env, capture_output=True, text=True, timeout=timeout)
    return p.returncode, p.stdout, p.stderr


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default=os.environ.get("EAT_REFERENCE_ROOT", "/root/reference"))
    ap.add_argument("--out", default=None)
    ap.add_argument("--audio", default=None, help="wav file for inference.py / windowed_inference.py (default: a synthetic clip)")
    ap.add_argument("--commit", default="unknown", help="commit hash to record in the log (the GPU box has no .git)")
    ap.add_argument("--verify", default=None, help="check that this log was produced from the tree this script runs in")
    ap.add_argument("--throughput", action="store_true",
                    help="also run the unmodified `ex_audioset.py --train --batch_size 120` for one epoch of 20 steps on the synthetic "
                         "AudioSet stand-in and record tqdm's steps/s (the eager loop a user of the reference's script gets)")
    a = ap.parse_args()
    if a.verify:
        m = re.search(r"tree_sha (\w+)", open(a.verify).read())
        ok = bool(m) and m.group(1) == tree_sha()
        print(f"{a.verify}: tree_sha {m.group(1) if m else None} vs this tree {tree_sha()}: {'MATCH' if ok else 'DIFFERENT'}")
        sys.exit(0 if ok else 1)
    work = tempfile.mkdtemp(prefix="eat_refscripts_")
    env = build_workdir(work, n_train=600 if a.throughput else 48)
    clip = "resources/synthetic_clip.wav"
    if a.audio:
        import shutil
        clip = "resources/" + os.path.basename(a.audio)
        shutil.copy(a.audio, os.path.join(work, clip))
    log = [f"# tools/run_reference_scripts.py: commit {a.commit}, tree_sha {tree_sha()}, audio {clip}: the reference's unmodified scripts\n"]
    runs = [("inference.py", ["--cuda", "--audio_path", clip]),
            ("ex_audioset.py", ["--cuda", "--batch_size", "31", "--num_workers", "0"]),               # evaluate(): fp16 autocast
            ("ex_audioset.py", ["--train", "--cuda", "--batch_size", "8", "--num_workers", "0", "--n_epochs", "1",
                                "--epoch_len", "32", "--pretrained"])]
    if os.path.exists(os.path.join(a.ref, "windowed_inference.py")):
        runs.append(("windowed_inference.py", ["--cuda", "--audio_path", clip, "--window_size", "4.0", "--hop_length", "3.0"]))
    if a.throughput:
        runs.append(("ex_audioset.py", ["--train", "--cuda", "--batch_size", "120", "--num_workers", "8", "--n_epochs", "1",
                                        "--epoch_len", "2400", "--pretrained"]))
    rc_all = 0
    for script, argv in runs:
        import time
        t0 = time.perf_counter()
        rc, out, err = run_script(a.ref, script, argv, work, env)
        el = time.perf_counter() - t0
        tail = "\n".join(err.strip().splitlines()[-6:])
        rates = re.findall(r"(\d+)/\1 \[[^\]]*?([\d.]+)(it/s|s/it)\]", err)              # tqdm's closing lines
        if rates:
            tail += f"\n[whole script {el:.1f} s; tqdm at completion: " + ", ".join(f"{n} steps at {v} {u}" for n, v, u in rates) + "]"
        log.append(f"$ python {script} {' '.join(argv)}   [unmodified reference script, dropin/ modules]\nrc={rc}\n{_drop_repr(out.strip())}\n--- stderr tail ---\n{tail}\n")
        rc_all |= rc
    wl = os.path.join(work, "wandb_run", "wandb_log.jsonl")
    if os.path.exists(wl):
        log.append("wandb_log.jsonl (stand-in logger): " + open(wl).read().strip())
    text = "\n".join(log)
    print(text)
    if a.out:
        with open(a.out, "w") as f:
            f.write(text + "\n")
    sys.exit(rc_all)


if __name__ == "__main__":
    main()
