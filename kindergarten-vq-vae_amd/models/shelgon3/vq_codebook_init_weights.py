"""Data-driven codebook initialisation on MI355X -- counterpart of models/shelgon3/vq_codebook_init_weights.py:1-115.

    PYTHONPATH=kindergarten-vq-vae_amd python3 kindergarten-vq-vae_amd/models/shelgon3/vq_codebook_init_weights.py

Same recipe as the reference: encode the TRAIN split (60/20/20, Generator(seed=DS_GEN_SEED)) with the Bagon encoder, flatten the
token embeddings to [N, e_dim], run k-means with `minit='points'`, 10 iterations, and save
{"codebook_init_values", "encoder_model_name", "decoder_model_name", "tokenizer_name"} where shelgon3/main.py reads it
(VQ_CODEBOOK_INIT_VALUES_PATH).  What changes: the embeddings never leave the GPU and the clustering is
kvq.functional.kmeans2_points -- scipy.cluster.vq.kmeans2's algorithm with the VQ arg-min kernel as its assignment step
(the reference copies every batch to the host and clusters 6.9 M x 768 floats on the CPU).
Settings come from models/shelgon3/config.py (KVQ_<NAME> environment overrides); N_E defaults to VQ_N_E.
"""
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(_HERE)))
sys.path.insert(0, _HERE)

from config import *  # noqa: E402,F401,F403

import torch  # noqa: E402
from torch.utils.data import DataLoader, random_split  # noqa: E402

from common.consts import *  # noqa: E402,F401,F403
from dsentences.dataset import dSentencesDataset  # noqa: E402
from dsentences.synthetic import write_corpus  # noqa: E402
from kvq.functional import kmeans2_points  # noqa: E402
from kvq.tokenizer import load_tokenizer  # noqa: E402
from models.bagon.Bagon import Bagon  # noqa: E402

CODEBOOK_INIT_BATCH_SIZE = int(os.environ.get("KVQ_CODEBOOK_INIT_BATCH_SIZE", 2048))       # reference :39
CODEBOOK_INIT_OUT = os.environ.get("KVQ_CODEBOOK_INIT_OUT") or os.path.join(os.path.dirname(SENTENCES_PATH),
                                                                             "dSentences_codebook_init_values.pth")
KMEANS_ITERS = int(os.environ.get("KVQ_KMEANS_ITERS", 10))                                   # kmeans2's default `iter`


@torch.no_grad()
def encode_split(model, tokenizer, loader, device, max_length):
    """[N_tokens, H] encoder outputs of every sentence of `loader`, kept on the GPU in the model's compute dtype."""
    chunks = []
    for batch in loader:
        tok = tokenizer(list(batch["sentence"]), return_tensors="pt", padding="max_length", max_length=max_length,
                        truncation=True, add_special_tokens=False)
        ids = tok.input_ids.to(device, non_blocking=True)
        mask = tok.attention_mask.to(device, non_blocking=True)
        chunks.append(model.encode(ids, mask).reshape(-1, model.encoder.config.hidden_size))
    return torch.cat(chunks)


def main():
    if not torch.cuda.is_available():
        raise SystemExit("vq_codebook_init_weights.py needs an MI355X: the clustering kernels have no CPU fallback")
    device = torch.device("cuda", 0)
    if not os.path.exists(SENTENCES_PATH):
        write_corpus(os.path.dirname(SENTENCES_PATH), SYNTHETIC_SENTENCES, seed=DS_GEN_SEED)
    ds = dSentencesDataset(SENTENCES_PATH, LATENT_CLASSES_LABELS_PATH, LATENT_CLASSES_ONE_HOT_PATH)
    n_train = int(len(ds) * TRAIN_SPLIT_PCT)
    n_val = int(len(ds) * VAL_SPLIT_PCT)
    gen = torch.Generator()
    gen.manual_seed(DS_GEN_SEED)
    ds_train, _, _ = random_split(ds, (n_train, n_val, len(ds) - n_train - n_val), gen)
    print(f"using {len(ds_train)} examples")
    loader = DataLoader(ds_train, batch_size=CODEBOOK_INIT_BATCH_SIZE, pin_memory=PIN_MEMORY)

    tokenizer = load_tokenizer(TOKENIZER_NAME)
    torch.manual_seed(0)
    model = Bagon(encoder_model_name=ENCODER_MODEL_NAME, decoder_model_name=DECODER_MODEL_NAME,
                  compute_dtype=getattr(torch, COMPUTE_DTYPE)).to(device).eval()
    if FROM_PRETRAINED_BAGON:                                   # reference :51 loads a trained Bagon checkpoint
        model.load_state_dict(torch.load(FROM_PRETRAINED_BAGON, map_location=device)["model_state_dict"])
    z = encode_split(model, tokenizer, loader, device, TOKENIZED_SENTENCE_MAX_LENGTH)
    print(f"encoded tokens: {tuple(z.shape)} {z.dtype}")

    print("running kmeans!!")
    rp = torch.randperm(z.shape[0], generator=torch.Generator().manual_seed(DS_GEN_SEED))     # reference :89 (unseeded there)
    codebook, labels = kmeans2_points(z, VQ_N_E, iters=KMEANS_ITERS, init_indices=rp[:VQ_N_E])
    used = torch.bincount(labels, minlength=VQ_N_E)
    print(f"kmeans done!! {int((used > 0).sum())}/{VQ_N_E} clusters in use")

    os.makedirs(os.path.dirname(CODEBOOK_INIT_OUT) or ".", exist_ok=True)
    torch.save({"codebook_init_values": codebook.cpu(), "encoder_model_name": ENCODER_MODEL_NAME,
                "decoder_model_name": DECODER_MODEL_NAME, "tokenizer_name": TOKENIZER_NAME}, CODEBOOK_INIT_OUT)
    print(f"values exported to {CODEBOOK_INIT_OUT}")
    return CODEBOOK_INIT_OUT


if __name__ == "__main__":
    main()
