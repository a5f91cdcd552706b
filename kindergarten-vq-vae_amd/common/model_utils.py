"""Parameter bookkeeping for the run summary (the surface of the reference's common/model_utils.py:9-27:
n_trainable_params / n_not_trainable_params / n_params / print_module_params_summary), built on one counting pass."""
from collections import namedtuple

from torch.nn import Module

ParamCounts = namedtuple("ParamCounts", "trainable frozen total")


def param_counts(module: Module) -> ParamCounts:
    """(trainable, frozen, total) element counts; shared (tied) parameters are counted once, as Module.parameters() does."""
    trainable = frozen = 0
    for p in module.parameters():
        if p.requires_grad:
            trainable += p.numel()
        else:
            frozen += p.numel()
    return ParamCounts(trainable, frozen, trainable + frozen)


def n_trainable_params(model: Module) -> int:
    return param_counts(model).trainable


def n_not_trainable_params(model: Module) -> int:
    return param_counts(model).frozen


def n_params(model: Module) -> int:
    return param_counts(model).total


def print_module_params_summary(module: Module, module_name: str, color_train: str, color_frozen: str, color_tot: str):
    from rich import print as rprint
    c = param_counts(module)
    denom = max(c.total, 1)
    rprint(f"{module_name} params summary:")
    for label, value, colour, pct in (("Trainable params", c.trainable, color_train, True),
                                      ("   Frozen params", c.frozen, color_frozen, True),
                                      ("      Tot params", c.total, color_tot, False)):
        share = f" {100.0 * value / denom:06.2f}%" if pct else ""
        rprint(f"{label}: [bold {colour}]{value:9d}{share}[/bold {colour}]")
    rprint()
