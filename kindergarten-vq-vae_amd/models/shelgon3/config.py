"""Run configuration of the Shelgon (VQ) entry point: a flat module of UPPER_CASE constants + get_config().

The reference imports this file with `from config import *` (models/shelgon3/main.py:1) and tells users to edit it
(README.md:33), but git-ignores it (`**config.py`), so it is absent from the reference checkout; the constant names
below are the ones main.py / Trainer.py consume (reconstructed from usage: SURVEY.md §5.6).  Values are the
BASELINE.json benchmark configuration (K=512, D=768, seq_len 32, batch 256) on synthetic dSentences.
Any constant can be overridden from the environment as KVQ_<NAME>=<python literal> (used by tests and the bench).
"""
import ast as _ast
import os as _os

# --- data -----------------------------------------------------------------------------------------------------------
SENTENCES_PATH = "./data/dSentences/dSentences_sentences_clean.npy"
LATENT_CLASSES_LABELS_PATH = "./data/dSentences/dSentences_latent_classes_labels_clean.npy"
LATENT_CLASSES_ONE_HOT_PATH = "./data/dSentences/dSentences_latent_classes_one_hot_clean.npy"
SYNTHETIC_SENTENCES = 65536          # written to SENTENCES_PATH when the corpus is absent (it is git-ignored upstream)
TRAIN_SPLIT_PCT = 0.6
VAL_SPLIT_PCT = 0.2
BATCH_SIZE = 256                     # per GPU
NUM_WORKERS = 0
PIN_MEMORY = True
TOKEN_CACHE = True              # tokenise every split once, keep it in HBM, batches = device index_select (dsentences/token_cache.py)

# --- model ----------------------------------------------------------------------------------------------------------
ENCODER_MODEL_NAME = "bert-base-uncased"
DECODER_MODEL_NAME = "bert-base-uncased"
TOKENIZER_NAME = "bert-base-uncased"
TOKENIZER_ADD_SPECIAL_TOKENS = False
TOKENIZED_SENTENCE_MAX_LENGTH = 32   # the reference hard-codes 12 (Trainer.py:82); BASELINE.json uses 32
VOCAB_SIZE = 30522
FROM_PRETRAINED_BAGON = None
CROSS_ATTN_MAKE_TRAINABLE = False
MODEL_MODE = "full"                  # full | dec-head-ft | enc-head-ft-dec-head-ft | vq-ft
COMPUTE_DTYPE = "bfloat16"           # bfloat16 | float32
USE_ENGINE = True                    # kvq.engine.TrainEngine (explicit fwd/bwd on flat buffers) when the model shape allows
FP8_FORWARD = False                  # extension (BASELINE.json configs[4]): forward GEMMs on the fp8 matrix cores -- False | True | "wide" | "all"

VQ_MODE = "VectorQuantizer"          # VectorQuantizer | GumbelQuantizer | MultiVectorQuantizer (extension: VQ_N_FACTORS codebooks)
VQ_N_FACTORS = 1                     # MultiVectorQuantizer: codebooks = slices of the encoder output (must divide VQ_E_DIM)
VQ_EMA_DECAY = None                  # extension, default off: EMA codebook update instead of the codebook gradient (e.g. 0.99)
VQ_N_E = 512
VQ_E_DIM = 768
VQ_BETA = 0.25
VQ_CODEBOOK_INIT_VALUES_PATH = None
ENC_OUT_SIZE = 768                   # GumbelQuantizer-only knobs, kept for config compatibility
VQ_TEMPERATURE = 1.0
VQ_KL_DIV_SCALE = 5e-4
VQ_STRAIGHT_THROUGH = True

# --- optimisation ---------------------------------------------------------------------------------------------------
LR = 1e-4
WEIGHT_DECAY = 0.0
AMSGRAD = False
LR_SCHEDULER = "MultiStepLR"
MILESTONES = [10000, 20000]          # in optimiser STEPS: the reference ticks the scheduler per step (Trainer.py:114-115)
GAMMA = 0.1
N_EPOCHS = 1
N_EPOCHS_TO_DECODE_AFTER = 1
LIM_BATCHES_TRAIN_PCT = 1.0
LIM_BATCHES_VAL_PCT = 1.0
LIM_BATCHES_TEST_PCT = 1.0
LOSS_RECON_RESCALE_FACTOR = 1.0
LOSS_RECON_WEIGHT = 1.0
LOSS_VQ_RESCALE_FACTOR = 1.0
LOSS_VQ_WEIGHT = 1.0
LOSS_PERP_RESCALE_FACTOR = 1.0
LOSS_PERP_WEIGHT = 0.0
GRAD_BUCKET_MIB = 64                 # gradient all-reduce bucket size (multi-GPU)

# --- run / logging --------------------------------------------------------------------------------------------------
RUNS_DIR = "./runs/Shelgon"
EXPORT_CHECKPOINT = True
WANDB_SILENT = "true"
WANDB_PROJECT_NAME = "kindergarten-vq-vae"
WANDB_GROUP = "Shelgon"
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
    """JSON-serialisable dict keyed by the lower-cased constant names (what analyses/* read from run_conf.json)."""
    return {k.lower(): v for k, v in globals().items() if k.isupper() and not k.startswith("_")}
