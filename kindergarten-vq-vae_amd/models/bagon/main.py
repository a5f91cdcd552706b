"""Entry point of the Bagon (BERT -> BERT sentence autoencoder) run on MI355X -- counterpart of models/bagon/main.py:37-163.

    PYTHONPATH=kindergarten-vq-vae_amd python3 kindergarten-vq-vae_amd/models/bagon/main.py
    (multi-GPU: python3 -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 <this file>)

Wiring as in the reference: dataset, 60/20/20 split (seed DS_GEN_SEED), loaders, Bagon, set_mode, encoder/decoder
tokenizers, Adam + MultiStepLR, run dir + run_conf.json, train, reload best-val checkpoint, test, feather dump.
The training step itself runs on kvq.engine.TrainEngine (USE_ENGINE, default on): hand-written HIP for every matrix product,
attention, LayerNorm, the loss and Adam; USE_ENGINE = False keeps the torch-autograd path (the checker)."""
import json
import os
import sys
from datetime import datetime

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(_HERE)))
sys.path.insert(0, _HERE)

from config import *  # noqa: E402,F401,F403

import torch  # noqa: E402
from torch.optim.adam import Adam  # noqa: E402
from torch.optim.lr_scheduler import MultiStepLR  # noqa: E402
from torch.utils.data import DataLoader, random_split  # noqa: E402
from torch.utils.data.distributed import DistributedSampler  # noqa: E402

from common.consts import *  # noqa: E402,F401,F403
from dsentences.dataset import dSentencesDataset  # noqa: E402
from dsentences.synthetic import write_corpus  # noqa: E402
from dsentences.token_cache import cache_of_split  # noqa: E402
from kvq import ddp  # noqa: E402
from kvq.engine import TrainEngine  # noqa: E402
from kvq.runlog import init_run  # noqa: E402
from kvq.tokenizer import load_tokenizer  # noqa: E402
from models.bagon.Bagon import Bagon  # noqa: E402
from models.bagon.Trainer import test, train  # noqa: E402


def main():
    rank, local_rank, world = ddp.init_distributed()
    is_main = rank == 0
    if not torch.cuda.is_available():
        raise SystemExit("models/bagon/main.py needs an MI355X: the step's kernels have no CPU fallback")
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)

    if not os.path.exists(DATASET_PATH):
        if is_main:
            write_corpus(os.path.dirname(DATASET_PATH), SYNTHETIC_SENTENCES, seed=DS_GEN_SEED)
        if world > 1:
            torch.distributed.barrier()
    ds = dSentencesDataset(DATASET_PATH, LATENT_CLASSES_LABELS_PATH, LATENT_CLASSES_ONE_HOT_PATH)
    n_tr, n_va = int(len(ds) * TRAIN_SPLIT_PCT), int(len(ds) * VAL_SPLIT_PCT)
    gen = torch.Generator()
    gen.manual_seed(DS_GEN_SEED)
    ds_train, ds_val, ds_test = random_split(ds, (n_tr, n_va, len(ds) - n_tr - n_va), gen)

    def loader(split, shuffle):
        sampler = DistributedSampler(split, world, rank, shuffle=shuffle, drop_last=True) if world > 1 else None
        return DataLoader(split, batch_size=BATCH_SIZE, num_workers=NUM_WORKERS, pin_memory=PIN_MEMORY,
                          shuffle=shuffle and sampler is None, sampler=sampler, drop_last=world > 1)
    dl_train, dl_val, dl_test = loader(ds_train, True), loader(ds_val, False), loader(ds_test, False)

    torch.manual_seed(0)
    model = Bagon(ENCODER_MODEL_NAME, DECODER_MODEL_NAME, CROSS_ATTN_MAKE_TRAINABLE,
                  compute_dtype=getattr(torch, COMPUTE_DTYPE)).to(device)
    model.set_mode(MODEL_MODE)
    ddp.broadcast_parameters(model)
    if is_main:
        model.model_params_summary_print()
    tok_enc = load_tokenizer(TOKENIZER_NAME_ENCODER)
    tok_dec = tok_enc if TOKENIZER_NAME_DECODER == TOKENIZER_NAME_ENCODER else load_tokenizer(TOKENIZER_NAME_DECODER)

    same_tok = tok_dec is tok_enc
    any_perturb = any(p != 0 for p in (ENCODER_PERTURB_TRAIN_PCT, ENCODER_PERTURB_VAL_PCT, ENCODER_PERTURB_TEST_PCT,
                                       DECODER_PERTURB_TRAIN_PCT, DECODER_PERTURB_VAL_PCT, DECODER_PERTURB_TEST_PCT))
    if TOKEN_CACHE and same_tok:
        # every split tokenised once and kept in HBM; a batch is an index_select on the device (no per-step tokenizer / H2D)
        caches = [cache_of_split(sp, tok_enc, TOKENIZED_SENTENCE_MAX_LENGTH, TOKENIZER_ADD_SPECIAL_TOKENS, device, keep_labels=True)
                  for sp in (ds_train, ds_val, ds_test)]
        dl_train = caches[0].loader(BATCH_SIZE, True, seed=DS_GEN_SEED, rank=rank, world=world)
        dl_val = caches[1].loader(BATCH_SIZE, False, rank=rank, world=world)
        dl_test = caches[2].loader(BATCH_SIZE, False, rank=rank, world=world)

    opt = Adam(params=[p for p in model.parameters()], lr=LR, weight_decay=WEIGHT_DECAY, amsgrad=AMSGRAD, fused=True)
    lr_sched = MultiStepLR(optimizer=opt, milestones=MILESTONES, gamma=GAMMA) if LR_SCHEDULER == "MultiStepLR" else None
    engine = grad_sync = None
    if USE_ENGINE and TrainEngine.supports(model, TOKENIZED_SENTENCE_MAX_LENGTH):
        # explicit forward/backward schedule on flat buffers (kvq/engine.py): own MFMA GEMMs and attention, fused loss, Adam, the
        # scheduler tick and the RCCL gradient exchange; `opt` above is then only the reference-shaped handle in run_conf.json
        engine = TrainEngine(model, lr=LR, weight_decay=WEIGHT_DECAY, amsgrad=AMSGRAD,
                             milestones=MILESTONES if LR_SCHEDULER == "MultiStepLR" else None, gamma=GAMMA, bucket_mib=GRAD_BUCKET_MIB,
                             fp8_forward=FP8_FORWARD if FP8_FORWARD else None)       # (None: the KVQ_FP8 environment switch decides)
        if TOKEN_CACHE and same_tok:
            for c in caches:
                c.packed_pad_id = engine.pad_idx if not any_perturb else "off"     # perturbed ids are sorted by the engine itself
    elif world > 1:
        grad_sync = ddp.GradSync(model.parameters(), bucket_mib=GRAD_BUCKET_MIB)

    console = None
    if is_main:
        from rich.console import Console
        console = Console()
    run_id = ddp.same_everywhere(datetime.now().strftime(RUN_ID_TIMESTAMP_FORMAT))     # one run directory for all ranks
    run_path = f"{RUNS_DIR}/{run_id}"
    run_conf = get_config()
    run_conf.update({"n_params": model.model_params_summary_dict(), "optimizer": str(opt), "run_id": run_id, "world_size": world})
    if is_main:
        os.makedirs(run_path, exist_ok=True)
        with open(f"{run_path}/run_conf.json", "w") as fp:
            json.dump(run_conf, fp)
    wandb_run = init_run(WANDB_PROJECT_NAME, WANDB_GROUP, WANDB_JOB_TYPE, run_conf, WANDB_MODE if is_main else "disabled",
                         run_path if is_main else None)

    decoded = []
    common = dict(tokenizer_encoder=tok_enc, tokenizer_decoder=tok_dec,
                  tokenizer_encoder_add_special_tokens=TOKENIZER_ADD_SPECIAL_TOKENS,
                  tokenized_encoder_sentence_max_length=TOKENIZED_SENTENCE_MAX_LENGTH,
                  tokenizer_decoder_add_special_tokens=TOKENIZER_ADD_SPECIAL_TOKENS,
                  tokenized_decoder_sentence_max_length=TOKENIZED_SENTENCE_MAX_LENGTH,
                  vocab_size_encoder=VOCAB_SIZE_ENCODER, vocab_size_decoder=VOCAB_SIZE_DECODER)
    train(prg=None, console=console, device=device, dl_train=dl_train, dl_val=dl_val,
          n_batches_train=int(len(dl_train) * LIM_BATCHES_TRAIN_PCT), n_batches_val=int(len(dl_val) * LIM_BATCHES_VAL_PCT),
          model=model, n_epochs_to_decode_after=N_EPOCHS_TO_DECODE_AFTER, decoded_sentences=decoded, opt=opt, lr_sched=lr_sched,
          n_epochs=N_EPOCHS, encoder_perturb_train_pct=ENCODER_PERTURB_TRAIN_PCT, encoder_perturb_val_pct=ENCODER_PERTURB_VAL_PCT,
          decoder_perturb_train_pct=DECODER_PERTURB_TRAIN_PCT, decoder_perturb_val_pct=DECODER_PERTURB_VAL_PCT, wandb_run=wandb_run,
          run_path=run_path, export_checkpoint=EXPORT_CHECKPOINT, grad_sync=grad_sync, engine=engine, is_main=is_main, **common)
    # The test stage runs on EVERY rank (each on its shard of the test split): test() ends in the stage's all-reduce of the
    # statistics (Trainer._sum_over_ranks), a collective every rank has to enter.  Rank 0 wrote the checkpoint; agree() puts a
    # barrier behind that write and hands every rank rank 0's answer to "is there a best checkpoint".
    best = f"{run_path}/bagon_ckpt_loss_recon_val_best.pth"
    if ddp.agree(EXPORT_CHECKPOINT and os.path.exists(best)):
        model.load_state_dict(torch.load(best, map_location=device)["model_state_dict"])
        if engine is not None:
            engine.sync_from_model()
        test(prg=None, console=console, device=device, dl_test=dl_test, n_batches_test=int(len(dl_test) * LIM_BATCHES_TEST_PCT),
             model=model, encoder_perturb_test_pct=ENCODER_PERTURB_TEST_PCT, decoder_perturb_test_pct=DECODER_PERTURB_TEST_PCT,
             decoded_sentences=decoded, epoch=N_EPOCHS, wandb_run=wandb_run, engine=engine, **common)
    decoded = ddp.gather_lists(decoded)            # every rank decoded its own shard: rank 0 writes them all
    if is_main:
        import pandas as pd
        try:
            pd.DataFrame(decoded).to_feather(f"{run_path}/decoded_sentences.feather")
        except ImportError as e:          # feather needs pyarrow
            print(f"[main] feather export unavailable ({e}); writing CSV instead")
            pd.DataFrame(decoded).to_csv(f"{run_path}/decoded_sentences.csv", index=False)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
