"""Does a saved checkpoint load and reconstruct? -- counterpart of common/test_checkpoint_validity.py:1-44.

    PYTHONPATH=kindergarten-vq-vae_amd KVQ_CKPT_PATH="'./runs/Bagon/<run>/bagon_ckpt_loss_recon_train_best.pth'" \
        python3 kindergarten-vq-vae_amd/common/test_checkpoint_validity.py

Same steps as the reference script: torch.load -> Bagon(...) -> load_state_dict(ckpt["model_state_dict"]) -> tokenizer ->
the three probe sentences (:35-39) -> model.forward.  The reference stops at the logits (and calls forward with two arguments,
which its own Bagon.forward(:40-45, four arguments) does not accept); this one passes the decoder inputs and prints the arg-max
reconstruction next to each input.  The forward runs without autograd, i.e. on the engine's kernels.  Not a pytest module: the
name is the reference's.
"""
import ast
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(_HERE))

import torch  # noqa: E402

from kvq.tokenizer import load_tokenizer  # noqa: E402
from models.bagon.Bagon import Bagon  # noqa: E402

SUPPORTED_MODEL_NAMES = ["Bagon"]
MODEL_NAME = "Bagon"
CKPT_PATH = f"./runs/{MODEL_NAME}/2024_01_27_11_15_55/bagon_ckpt_loss_recon_train_best.pth"        # :16
ENCODER_MODEL_NAME = "bert-base-uncased"
DECODER_MODEL_NAME = "bert-base-uncased"
TOKENIZER_NAME = "bert-base-uncased"
COMPUTE_DTYPE = "bfloat16"
BATCH = ["he accepted the payment", "are you not ruining the holidays", "they were touring the lakes"]   # :35-39

for _k in [k for k in list(globals()) if k.isupper()]:
    _v = os.environ.get("KVQ_" + _k)
    if _v is not None:
        try:
            globals()[_k] = ast.literal_eval(_v)
        except (ValueError, SyntaxError):
            globals()[_k] = _v


def main():
    if not torch.cuda.is_available():
        raise SystemExit("needs an MI355X: the model's forward has no CPU path here")
    device = torch.device("cuda", 0)
    loaded_checkpoint = torch.load(CKPT_PATH, map_location=device)
    if MODEL_NAME != "Bagon":
        raise ValueError(f"Invalid model name \"{MODEL_NAME}\", supported values: {', '.join(SUPPORTED_MODEL_NAMES)}")
    model = Bagon(encoder_model_name=ENCODER_MODEL_NAME, decoder_model_name=DECODER_MODEL_NAME,
                  compute_dtype=getattr(torch, COMPUTE_DTYPE)).to(device)
    model.load_state_dict(loaded_checkpoint["model_state_dict"])
    model.eval()
    tokenizer = load_tokenizer(TOKENIZER_NAME)
    tokenized = tokenizer(BATCH, return_tensors="pt", padding=True, add_special_tokens=False)
    input_ids = tokenized.input_ids.to(device)
    attention_mask = tokenized.attention_mask.to(device)
    with torch.no_grad():
        logits_recon = model.forward(input_ids, attention_mask, input_ids, attention_mask)
    recon = tokenizer.batch_decode(logits_recon.argmax(-1) * attention_mask)
    for s, r in zip(BATCH, recon):
        print(f"{s!r} -> {r!r}")
    print(f"logits {tuple(logits_recon.shape)} {logits_recon.dtype}, finite: {bool(torch.isfinite(logits_recon.float()).all())}")
    return logits_recon


if __name__ == "__main__":
    main()
