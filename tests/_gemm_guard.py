"""Shared by the GPU tests: every torch entry point that would multiply matrices with the vendor library raises while the bf16
engine runs, so that a passing step provably made all of its products in libkvq.so."""
import torch
import torch.nn.functional as F


def forbid_vendor_gemms(monkeypatch):
    def boom(name):
        def f(*a, **k):
            raise AssertionError(f"torch.{name} reached from the bf16 engine: a vendor-library GEMM")
        return f
    for name in ("mm", "addmm", "bmm", "matmul", "baddbmm", "einsum"):
        monkeypatch.setattr(torch, name, boom(name))
    monkeypatch.setattr(torch.Tensor, "addmm_", boom("Tensor.addmm_"))
    monkeypatch.setattr(torch.Tensor, "__matmul__", boom("Tensor.__matmul__"))
    monkeypatch.setattr(F, "linear", boom("nn.functional.linear"))
