"""Entry point of the Shelgon (VQ sentence autoencoder) run on MI355X -- counterpart of models/shelgon3/main.py:40-187.

    PYTHONPATH=kindergarten-vq-vae_amd python3 kindergarten-vq-vae_amd/models/shelgon3/main.py             # 1 GPU
    PYTHONPATH=kindergarten-vq-vae_amd python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 8 \
        --master-addr 127.0.0.1 kindergarten-vq-vae_amd/models/shelgon3/main.py                            # 8 GPUs, RCCL

Same wiring as the reference: dataset -> 60/20/20 split with Generator(seed=DS_GEN_SEED) -> loaders ->
VectorQuantizer -> Shelgon -> set_mode -> tokenizer -> Adam + MultiStepLR -> run dir + run_conf.json -> train ->
reload best-val checkpoint -> test -> decoded sentences to feather.  Multi-GPU is plain data parallel
(kvq.ddp.GradSync); `model.compile()` of the reference (main.py:83) has no counterpart: the fused kernels are
hand written, there is no tracing compiler in this build.
"""
import json
import os
import sys
from datetime import datetime

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(_HERE)))      # package root (common/, kvq/, models/, dsentences/)
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
from models.shelgon3.Shelgon import Shelgon  # noqa: E402
from models.shelgon3.Trainer import test, train  # noqa: E402
from models.shelgon3.GumbelQuantizer import GumbelQuantizer  # noqa: E402
from models.shelgon3.MultiVectorQuantizer import MultiVectorQuantizer  # noqa: E402
from models.shelgon3.VectorQuantizer import VectorQuantizer  # noqa: E402


def main():
    rank, local_rank, world = ddp.init_distributed()
    is_main = rank == 0
    if not torch.cuda.is_available():
        raise SystemExit("models/shelgon3/main.py needs an MI355X: the VQ / loss kernels have no CPU fallback")
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)

    if not os.path.exists(SENTENCES_PATH):          # the corpus is git-ignored upstream and absent offline
        if is_main:
            write_corpus(os.path.dirname(SENTENCES_PATH), SYNTHETIC_SENTENCES, seed=DS_GEN_SEED)
        if world > 1:
            torch.distributed.barrier()
    ds = dSentencesDataset(SENTENCES_PATH, LATENT_CLASSES_LABELS_PATH, LATENT_CLASSES_ONE_HOT_PATH)
    ds_train_len = int(len(ds) * TRAIN_SPLIT_PCT)
    ds_val_len = int(len(ds) * VAL_SPLIT_PCT)
    ds_test_len = len(ds) - ds_train_len - ds_val_len
    ds_gen = torch.Generator()
    ds_gen.manual_seed(DS_GEN_SEED)
    ds_train, ds_val, ds_test = random_split(ds, (ds_train_len, ds_val_len, ds_test_len), ds_gen)

    def loader(split, shuffle):
        sampler = DistributedSampler(split, world, rank, shuffle=shuffle, drop_last=True) if world > 1 else None
        return DataLoader(split, batch_size=BATCH_SIZE, num_workers=NUM_WORKERS, pin_memory=PIN_MEMORY,
                          shuffle=shuffle and sampler is None, sampler=sampler, drop_last=world > 1)
    dl_train, dl_val, dl_test = loader(ds_train, True), loader(ds_val, False), loader(ds_test, False)

    if VQ_MODE == "VectorQuantizer":
        init = torch.load(VQ_CODEBOOK_INIT_VALUES_PATH)["codebook_init_values"] if VQ_CODEBOOK_INIT_VALUES_PATH else None
        vector_quantizer = VectorQuantizer(n_e=VQ_N_E, e_dim=VQ_E_DIM, beta=VQ_BETA, vq_codebook_init_values=init,
                                           ema_decay=VQ_EMA_DECAY)
        vector_quantizer.materialize_min_encodings = False          # the model drops min_encodings (Shelgon.py:58)
    elif VQ_MODE == "MultiVectorQuantizer":                         # extension (BASELINE.json configs[4]), off by default
        vector_quantizer = MultiVectorQuantizer(n_factors=VQ_N_FACTORS, n_e=VQ_N_E, e_dim=VQ_E_DIM, beta=VQ_BETA,
                                                ema_decay=VQ_EMA_DECAY)
    elif VQ_MODE == "GumbelQuantizer":                              # main.py:68-73
        vector_quantizer = GumbelQuantizer(enc_out_size=ENC_OUT_SIZE, n_embed=VQ_N_E, embedding_dim=VQ_E_DIM,
                                           temperature=VQ_TEMPERATURE, kl_div_scale=VQ_KL_DIV_SCALE,
                                           straight_through=VQ_STRAIGHT_THROUGH)
    else:
        raise ValueError(f"{VQ_MODE} vector quantizer mode NOT supported. Supported modalities: VectorQuantizer, GumbelQuantizer")

    torch.manual_seed(0)                                          # same init on every rank
    model = Shelgon(encoder_model_name=ENCODER_MODEL_NAME, vector_quantizer=vector_quantizer,
                    decoder_model_name=DECODER_MODEL_NAME, from_pretrained_bagon=FROM_PRETRAINED_BAGON,
                    cross_attn_make_trainable=CROSS_ATTN_MAKE_TRAINABLE,
                    compute_dtype=getattr(torch, COMPUTE_DTYPE)).to(device)
    model.set_mode(MODEL_MODE)
    ddp.broadcast_parameters(model)
    if is_main:
        model.model_params_summary_print()

    tokenizer = load_tokenizer(TOKENIZER_NAME)
    if TOKEN_CACHE:
        # every split tokenised once and kept in HBM; a batch is an index_select on the device (no per-step tokenizer / H2D)
        caches = [cache_of_split(sp, tokenizer, TOKENIZED_SENTENCE_MAX_LENGTH, TOKENIZER_ADD_SPECIAL_TOKENS, device)
                  for sp in (ds_train, ds_val, ds_test)]
        dl_train = caches[0].loader(BATCH_SIZE, True, seed=DS_GEN_SEED, rank=rank, world=world)
        dl_val = caches[1].loader(BATCH_SIZE, False, rank=rank, world=world)
        dl_test = caches[2].loader(BATCH_SIZE, False, rank=rank, world=world)
    opt = Adam(params=[p for p in model.parameters()], lr=LR, weight_decay=WEIGHT_DECAY, amsgrad=AMSGRAD, fused=True)
    lr_sched = MultiStepLR(optimizer=opt, milestones=MILESTONES, gamma=GAMMA) if LR_SCHEDULER == "MultiStepLR" else None
    engine = grad_sync = None
    if USE_ENGINE and TrainEngine.supports(model, TOKENIZED_SENTENCE_MAX_LENGTH):
        # explicit forward/backward schedule on flat buffers (kvq/engine.py): owns Adam, the scheduler tick and the
        # RCCL gradient exchange; `opt` above is then only the reference-shaped handle recorded in run_conf.json
        engine = TrainEngine(model, lr=LR, weight_decay=WEIGHT_DECAY, amsgrad=AMSGRAD,
                             milestones=MILESTONES if LR_SCHEDULER == "MultiStepLR" else None, gamma=GAMMA,
                             loss_recon_scale=LOSS_RECON_RESCALE_FACTOR * LOSS_RECON_WEIGHT,
                             loss_vq_scale=LOSS_VQ_RESCALE_FACTOR * LOSS_VQ_WEIGHT, bucket_mib=GRAD_BUCKET_MIB,
                             fp8_forward=FP8_FORWARD if FP8_FORWARD else None)       # (None: the KVQ_FP8 environment switch decides)
        if TOKEN_CACHE:
            for c in caches:          # the packed sort files the MODEL's padding row under -1 (not the tokenizer's pad id)
                c.packed_pad_id = engine.pad_idx
    elif world > 1:
        grad_sync = ddp.GradSync(model.parameters(), bucket_mib=GRAD_BUCKET_MIB)

    console = prg = None
    if is_main:
        from rich.console import Console
        from rich.progress import BarColumn, MofNCompleteColumn, Progress, TextColumn, TimeElapsedColumn, TimeRemainingColumn
        console = Console()
        prg = Progress(TextColumn("[progress.description]{task.description}"), BarColumn(), MofNCompleteColumn(),
                       TimeElapsedColumn(), TimeRemainingColumn(), TextColumn("[bold #5B4328]{task.speed} it/s"), console=console)

    run_id = ddp.same_everywhere(datetime.now().strftime(RUN_ID_TIMESTAMP_FORMAT))     # one run directory for all ranks
    run_path = f"{RUNS_DIR}/{run_id}"
    run_conf = get_config()
    run_conf.update({"n_params": model.model_params_summary_dict(), "optimizer": str(opt), "run_id": run_id, "world_size": world})
    if is_main:
        os.makedirs(run_path, exist_ok=True)
        console.print(f"Run ID: [bold {COLOR_RUN_ID}]{run_id}\n")
        with open(f"{run_path}/run_conf.json", "w") as fp:
            json.dump(run_conf, fp)
    os.environ["WANDB_SILENT"] = WANDB_SILENT
    wandb_run = init_run(WANDB_PROJECT_NAME, WANDB_GROUP, WANDB_JOB_TYPE, run_conf, WANDB_MODE if is_main else "disabled",
                         run_path if is_main else None)

    weights = dict(loss_recon_rescale_factor=LOSS_RECON_RESCALE_FACTOR, loss_recon_weight=LOSS_RECON_WEIGHT,
                   loss_vq_rescale_factor=LOSS_VQ_RESCALE_FACTOR, loss_vq_weight=LOSS_VQ_WEIGHT,
                   loss_perp_rescale_factor=LOSS_PERP_RESCALE_FACTOR, loss_perp_weight=LOSS_PERP_WEIGHT)
    n_batches_train = int(len(dl_train) * LIM_BATCHES_TRAIN_PCT)
    n_batches_val = int(len(dl_val) * LIM_BATCHES_VAL_PCT)
    decoded_sentences = []
    train(prg=prg, console=console, device=device, dl_train=dl_train, dl_val=dl_val, n_batches_train=n_batches_train,
          n_batches_val=n_batches_val, model=model, tokenizer=tokenizer, tokenizer_add_special_tokens=TOKENIZER_ADD_SPECIAL_TOKENS,
          n_epochs_to_decode_after=N_EPOCHS_TO_DECODE_AFTER, decoded_sentences=decoded_sentences, opt=opt, lr_sched=lr_sched,
          n_epochs=N_EPOCHS, vocab_size=VOCAB_SIZE, wandb_run=wandb_run, run_path=run_path, export_checkpoint=EXPORT_CHECKPOINT,
          max_length=TOKENIZED_SENTENCE_MAX_LENGTH, grad_sync=grad_sync, is_main=is_main, engine=engine, **weights)

    # The test stage runs on EVERY rank (each on its shard of the test split): test() ends in the stage's all-reduce of the
    # statistics (Trainer._sum_over_ranks), a collective every rank has to enter.  Rank 0 wrote the checkpoint; agree() puts a
    # barrier behind that write and hands every rank rank 0's answer to "is there a best checkpoint".
    best = f"{run_path}/shelgon_ckpt_loss_recon_val_best.pth"
    if ddp.agree(EXPORT_CHECKPOINT and os.path.exists(best)):
        model.load_state_dict(torch.load(best, map_location=device)["model_state_dict"])
        if engine is not None:
            engine.sync_from_model()
        n_batches_test = int(len(dl_test) * LIM_BATCHES_TEST_PCT)
        test(prg=prg, console=console, device=device, dl_test=dl_test, n_batches_test=n_batches_test, model=model,
             tokenizer=tokenizer, tokenizer_add_special_tokens=TOKENIZER_ADD_SPECIAL_TOKENS, decoded_sentences=decoded_sentences,
             vocab_size=VOCAB_SIZE, epoch=N_EPOCHS, wandb_run=wandb_run, max_length=TOKENIZED_SENTENCE_MAX_LENGTH, engine=engine,
             **weights)
    decoded_sentences = ddp.gather_lists(decoded_sentences)      # every rank decoded its own shard: rank 0 writes them all
    if is_main:
        if prg is not None:
            prg.stop()
        import pandas as pd
        try:
            pd.DataFrame(decoded_sentences).to_feather(f"{run_path}/decoded_sentences.feather")      # main.py:183-184
        except ImportError as e:          # feather needs pyarrow
            print(f"[main] feather export unavailable ({e}); writing CSV instead")
            pd.DataFrame(decoded_sentences).to_csv(f"{run_path}/decoded_sentences.csv", index=False)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
