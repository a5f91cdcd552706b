"""Run configuration of the Bagon (plain BERT->BERT autoencoder) entry point: UPPER_CASE constants + get_config().

Counterpart of the git-ignored models/bagon/config.py the reference imports with `from config import *`
(models/bagon/main.py:1,102); constant names reconstructed from their use in main.py / Trainer.py.
Override any of them from the environment with KVQ_<NAME>=<python literal>."""
import ast as _ast
import os as _os

DATASET_PATH = "./data/dSentences/dSentences_sentences_clean.npy"
LATENT_CLASSES_LABELS_PATH = "./data/dSentences/dSentences_latent_classes_labels_clean.npy"
LATENT_CLASSES_ONE_HOT_PATH = "./data/dSentences/dSentences_latent_classes_one_hot_clean.npy"
SYNTHETIC_SENTENCES = 65536
TRAIN_SPLIT_PCT = 0.6
VAL_SPLIT_PCT = 0.2
BATCH_SIZE = 256
NUM_WORKERS = 0
PIN_MEMORY = True
TOKEN_CACHE = True              # tokenise every split once, keep it in HBM, batches = device index_select (dsentences/token_cache.py)

ENCODER_MODEL_NAME = "bert-base-uncased"
DECODER_MODEL_NAME = "bert-base-uncased"
CROSS_ATTN_MAKE_TRAINABLE = True
MODEL_MODE = "full"
COMPUTE_DTYPE = "bfloat16"
TOKENIZER_NAME_ENCODER = "bert-base-uncased"
TOKENIZER_NAME_DECODER = "bert-base-uncased"
TOKENIZER_ADD_SPECIAL_TOKENS = False
TOKENIZED_SENTENCE_MAX_LENGTH = 32
ENCODER_PERTURB_TRAIN_PCT = 0.0
ENCODER_PERTURB_VAL_PCT = 0.0
ENCODER_PERTURB_TEST_PCT = 0.0
DECODER_PERTURB_TRAIN_PCT = 0.0
DECODER_PERTURB_VAL_PCT = 0.0
DECODER_PERTURB_TEST_PCT = 0.0
VOCAB_SIZE_ENCODER = 30522
VOCAB_SIZE_DECODER = 30522

LR = 1e-4
WEIGHT_DECAY = 0.0
AMSGRAD = False
LR_SCHEDULER = "MultiStepLR"
MILESTONES = [10000, 20000]
GAMMA = 0.1
N_EPOCHS = 1
N_EPOCHS_TO_DECODE_AFTER = 1
LIM_BATCHES_TRAIN_PCT = 1.0
LIM_BATCHES_VAL_PCT = 1.0
LIM_BATCHES_TEST_PCT = 1.0
GRAD_BUCKET_MIB = 64
USE_ENGINE = True               # kvq.engine.TrainEngine (explicit fwd/bwd on flat buffers, own HIP kernels) when the model shape allows
FP8_FORWARD = False             # extension (BASELINE.json configs[4]): forward GEMMs on the fp8 matrix cores -- False | True | "wide" | "all"

RUNS_DIR = "./runs/Bagon"
EXPORT_CHECKPOINT = True
WANDB_SILENT = "true"
WANDB_PROJECT_NAME = "kindergarten-vq-vae"
WANDB_GROUP = "Bagon"
WANDB_JOB_TYPE = "train"
WANDB_MODE = "disabled"
WANDB_WATCH_MODEL = False
WANDB_LOG_CODE = False

for _k in [k for k in list(globals()) if k.isupper()]:
    _v = _os.environ.get("KVQ_" + _k)
    if _v is not None:
        try:
            globals()[_k] = _ast.literal_eval(_v)
        except (ValueError, SyntaxError):
            globals()[_k] = _v


def get_config() -> dict:
    return {k.lower(): v for k, v in globals().items() if k.isupper() and not k.startswith("_")}
