#!/usr/bin/env python3
"""Throughput of the ENTRY POINTS: runs models/shelgon3/main.py and models/bagon/main.py as a user would --
synthetic dSentences corpus, bert-base encoder / decoder, batch 256, 32 tokens, TrainEngine + TokenCache, logging to the JSONL
run log -- and reports sentences/s of train() per epoch (the "perf/train_sentences_per_s" entries both trainers log: wall time
of the train stage, loop and all, closed by the epoch's device -> host read of the statistics), beside bench.py's number on the
same box.

    python tools/entrypoint_rate.py [--sentences 100000] [--epochs 3] [--out gpurun_out/entry] [--skip-bench]
Epoch 1 contains the engine's eager warm-up steps and the hipGraph capture; the steady-state figure is the mean of epochs >= 2.
Decoding sentences (a host loop over tokenizer.batch_decode, N_EPOCHS_TO_DECODE_AFTER) is switched off: it is the reference's
periodic logging step, not the training loop."""
import argparse
import glob
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "kindergarten-vq-vae_amd")


def run_main(script, out, sentences, epochs, extra):
    work = os.path.join(out, os.path.basename(os.path.dirname(script)))
    os.makedirs(work, exist_ok=True)
    data = os.path.join(out, "data")
    env = dict(os.environ, PYTHONPATH=PKG)
    env.update({"KVQ_SYNTHETIC_SENTENCES": str(sentences), "KVQ_N_EPOCHS": str(epochs), "KVQ_N_EPOCHS_TO_DECODE_AFTER": "1000000",
                "KVQ_RUNS_DIR": repr(os.path.join(work, "runs")), "KVQ_EXPORT_CHECKPOINT": "False", "KVQ_WANDB_MODE": "'disabled'",
                "KVQ_SENTENCES_PATH": repr(data + "/dSentences_sentences_clean.npy"), "KVQ_DATASET_PATH": repr(data + "/dSentences_sentences_clean.npy"),
                "KVQ_LATENT_CLASSES_LABELS_PATH": repr(data + "/dSentences_latent_classes_labels_clean.npy"),
                "KVQ_LATENT_CLASSES_ONE_HOT_PATH": repr(data + "/dSentences_latent_classes_one_hot_clean.npy")})
    env.update(extra)
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(PKG, script)], env=env, cwd=work, capture_output=True, text=True)
    open(os.path.join(work, "main.log"), "w").write(r.stdout[-20000:] + "\n--- stderr ---\n" + r.stderr[-20000:])
    if r.returncode != 0:
        raise SystemExit(f"{script} failed ({r.returncode}): see {work}/main.log\n{r.stderr[-2000:]}")
    run = max(glob.glob(os.path.join(work, "runs", "*")), key=os.path.getmtime)
    logs = [json.loads(l) for l in open(os.path.join(run, "metrics.jsonl"))]
    perf = [l for l in logs if "perf/train_sentences_per_s" in l]
    conf = json.load(open(os.path.join(run, "run_conf.json")))
    steady = [l["perf/train_sentences_per_s"] for l in perf[1:]] or [perf[0]["perf/train_sentences_per_s"]]
    return {"script": script, "wall_s": time.time() - t0, "batch_size": conf.get("batch_size"), "use_engine": conf.get("use_engine"),
            "token_cache": conf.get("token_cache"), "epochs": [{k: l[k] for k in ("epoch", "perf/train_s", "perf/train_steps", "perf/train_sentences_per_s")}
                                                               for l in perf],
            "train_sentences_per_s": sum(steady) / len(steady),
            "train_ms_per_step": 1e3 * sum(l["perf/train_s"] for l in perf[1:] or perf) / sum(l["perf/train_steps"] for l in perf[1:] or perf),
            "train_loss_recon": [l["train/loss_recon"] for l in logs if "train/loss_recon" in l]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sentences", type=int, default=100000)
    ap.add_argument("--epochs", type=int, default=3)
    ap.add_argument("--out", default="gpurun_out/entry")
    ap.add_argument("--skip-bench", action="store_true")
    ap.add_argument("--only", default="", help="shelgon3 | bagon")
    a = ap.parse_args()
    out = os.path.abspath(a.out)
    os.makedirs(out, exist_ok=True)
    res = {}
    if not a.skip_bench:
        for tag, flags in (("bench", []), ("bench_bagon", ["--bagon"])):
            if a.only and (("bagon" in tag) != (a.only == "bagon")):
                continue
            r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--steps", "200", "--warmup", "10"] + flags,
                               capture_output=True, text=True, cwd=ROOT)
            line = [l for l in r.stdout.splitlines() if l.startswith("{")]
            if r.returncode != 0 or not line:
                raise SystemExit(f"bench.py failed: {r.stderr[-2000:]}")
            j = json.loads(line[-1])
            res[tag] = {k: j.get(k) for k in ("value", "ms_per_step", "clock_mhz", "final_loss", "value_pack_in_step", "ms_per_step_pack_in_step")}
    if a.only in ("", "shelgon3"):
        res["shelgon3_main"] = run_main("models/shelgon3/main.py", out, a.sentences, a.epochs, {})
    if a.only in ("", "bagon"):
        res["bagon_main"] = run_main("models/bagon/main.py", out, a.sentences, a.epochs, {})
    for main_key, bench_key in (("shelgon3_main", "bench"), ("bagon_main", "bench_bagon")):
        if main_key in res and bench_key in res:
            res[main_key]["train_over_bench"] = res[main_key]["train_sentences_per_s"] / res[bench_key]["value"]
    print(json.dumps(res, indent=1))
    json.dump(res, open(os.path.join(out, "entrypoint_rate.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
