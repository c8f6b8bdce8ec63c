"""TEST-ONLY stand-in for `wandb` (the reference's experiment logger, `ex_audioset.py:35-42,217-236`): init / log / run.dir.
`log` appends one JSON object per call to <run.dir>/wandb_log.jsonl so a test can read the training statistics back."""
import json
import os
import tempfile


class _Run:
    def __init__(self):
        self.dir = os.environ.get("WANDB_STANDIN_DIR") or tempfile.mkdtemp(prefix="wandb_standin_")
        os.makedirs(self.dir, exist_ok=True)


run = None


def init(**kwargs):
    global run
    run = _Run()
    return run


def log(d, **kwargs):
    with open(os.path.join(run.dir, "wandb_log.jsonl"), "a") as f:
        f.write(json.dumps({k: float(v) for k, v in d.items()}) + "\n")
