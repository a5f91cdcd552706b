"""Run logging: the `wandb_run.log(dict)` / `.watch` surface the trainers call (models/shelgon3/Trainer.py:345),
backed by wandb when it is installed and WANDB_MODE is not "disabled", by a JSONL file otherwise."""
from __future__ import annotations

import json
import os
import time


class JsonlRun:
    def __init__(self, run_path: str, config: dict | None = None):
        self.path = os.path.join(run_path, "metrics.jsonl") if run_path else None
        self.config = config or {}
        self.history = []

    def log(self, record: dict):
        rec = {k: (float(v) if hasattr(v, "__float__") else v) for k, v in record.items()}
        rec["_time"] = time.time()
        self.history.append(rec)
        if self.path:
            with open(self.path, "a") as fp:
                fp.write(json.dumps(rec) + "\n")

    def watch(self, *_a, **_k):
        pass

    def log_code(self, *_a, **_k):
        pass

    def finish(self):
        pass


def init_run(project, group, job_type, config, mode, run_path):
    if mode != "disabled":
        try:
            import wandb
            return wandb.init(project=project, group=group, job_type=job_type, config=config, mode=mode)
        except ImportError:
            pass
    return JsonlRun(run_path, config)
