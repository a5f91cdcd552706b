#!/usr/bin/env python3
"""bench.py -- dSentences train sentences/sec of the Shelgon (BERT-base enc/dec + VQ K=512, D=768) step on MI355X.

Contract (driver):  python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run)
Prints ONE JSON line on rank 0.  A "step" = tokenised batch already in HBM -> encoder -> fused VQ -> decoder ->
fused LM-head loss -> backward -> (RCCL gradient all-reduce, overlapped) -> Adam.  Workload = BASELINE.json
configs[1]: bf16, seq_len 32, per-GPU batch 256, `full` mode (all 247.8 M parameters trained), dropout active.
Weak scaling: per-GPU batch is fixed, global batch = 256*N.

Extra objects on the line:
  roofline     the VQ distance+argmin kernel (vq_dist_packed_kernel), timed with HIP events around every launch of the timed
               region on the stream it runs on; bound = f32 MFMA (exact-f32 distances), HBM figure beside it
  cpu_baseline oracle/step_oracle.py (CPU f32 restatement, "port") timed on this host's cores, rank 0 at N=1 only
  clock_mhz    the shader clock held over the timed region (kvq_clock_probe stamps bracket it); roofline.frac_at_clock prices the
               kernel against the peak at THAT clock, so lines from different boxes / rounds can be compared
  vs_cpu_port  value / cpu_baseline.value -- NOT a like-for-like speed-up (GPU batch 256 bf16 against the CPU restatement at
               batch 8 f32: BASELINE.json's pairing of configs[1] with configs[0]); `vs_baseline` stays null, BASELINE.md
               publishes no number for this metric
  (kernel time per family comes from the rocprofv3 trace, tools/step_breakdown.py -> profiles/rNN_step_breakdown.txt; the
   event-pair estimate of rounds 3 - 4 disagreed with it by 2 ms and is off the line: --family-steps N brings it back for debugging)
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "kindergarten-vq-vae_amd"))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

BERT_BASE = dict()   # BertConfig() defaults = bert-base-uncased architecture
F32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: dense f32 MFMA peak
BF16_MFMA_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA peak (the 2:1-sparsity headline figure is twice that)
HBM_PEAK_GBPS = 8000.0         # HBM3E spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=256, help="sentences per GPU")
    ap.add_argument("--seq-len", type=int, default=32)
    ap.add_argument("--codes", type=int, default=512)
    ap.add_argument("--mode", default="full")
    ap.add_argument("--dtype", default="bfloat16", choices=["bfloat16", "float32"])
    ap.add_argument("--model", default="bert-base-uncased")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=20)
    ap.add_argument("--bucket-mib", type=int, default=64)
    ap.add_argument("--fp8", action="store_true", help="BASELINE.json configs[4]: forward GEMMs on the fp8 matrix cores where that beats the "
                    "bf16 kernel -- the LM head and the all-layer cross-K/V projection (backward bf16)")
    ap.add_argument("--cpu-same-batch", action="store_true", help="also time the CPU restatement at THIS run's batch size (one warm-up + two "
                    "timed steps, ~1 minute of host time): `cpu_baseline_same_batch`, the like-for-like ratio `vs_cpu_port_same_batch`")
    ap.add_argument("--no-distance-phase", action="store_true", help="skip the 9 extra steps with the three-kernel quantiser forward that time "
                    "the distance / arg-min kernel alone (roofline.distance_phase); kernel traces of the step use this")
    ap.add_argument("--fp8-wide", action="store_true", help="round 4's scope: fp8 for the LM head and the all-layer cross-K/V projection only")
    ap.add_argument("--fp8-all", action="store_true", help="every forward GEMM on fp8 (rounds 2 - 3; slower: the per-layer GEMMs lose to their quantisation passes)")
    ap.add_argument("--factors", type=int, default=1, help="configs[4]: codebooks (MultiVectorQuantizer, K codes each); 1 = the reference's VectorQuantizer")
    ap.add_argument("--bagon", action="store_true", help="the plain Bagon step (models/bagon/main.py: no quantiser) with the decoder's ids "
                    "perturbed independently of the encoder's (models/bagon/Trainer.py:85,94); an extra line for profiles/, not the default")
    ap.add_argument("--family-steps", type=int, default=0, help="debug aid, off the default line: N eager steps after the timed region with "
                    "an event pair around every entry point (the chip idles between kernels: NOT comparable with the rocprofv3 trace)")
    ap.add_argument("--pack-in-step", action="store_true", help="also time K steps that build their batch inside the timed region "
                    "(TokenCache.batch: index_select + the ids' stable sort), as models/*/Trainer.train() does: `value_pack_in_step`")
    ap.add_argument("--path", default="engine", choices=["engine", "autograd"],
                    help="engine = kvq.engine.TrainEngine (explicit fwd/bwd over flat buffers); autograd = kvq.bert + torch autograd")
    return ap.parse_args()


def spawn_ranks(a):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves (one process per GPU, torch.distributed.run,
    rendezvous on 127.0.0.1) BEFORE this process has touched the GPU, hand their output through and exit with their code."""
    import socket
    import subprocess
    n_dev = torch.cuda.device_count()            # does not initialise the GPU
    if n_dev < a.gpus and os.environ.get("KVQ_DIST_BACKEND") != "gloo":
        print(f"[bench] --gpus {a.gpus} but only {n_dev} GPU(s) are visible", file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(a))
    from kvq import _ffi, ddp
    from dsentences.synthetic import random_token_batch
    from models.shelgon3.Shelgon import Shelgon
    from models.shelgon3.VectorQuantizer import VectorQuantizer

    rank, local, world = ddp.init_distributed()
    grouped = world > 1 or dist.is_initialized()          # (a one-rank group: KVQ_DP_SINGLE_RANK=1, rehearsal of the RCCL branch)
    if world != a.gpus:
        raise SystemExit(f"[bench] --gpus {a.gpus} but WORLD_SIZE={world}: refusing to report a number for a different job size")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    lib = _ffi.lib()

    dtype = getattr(torch, a.dtype)
    torch.manual_seed(0)
    from models.bagon.Bagon import LOCAL_BERT_CONFIGS
    hidden = LOCAL_BERT_CONFIGS[a.model].get("hidden_size", 768)
    vq = None
    if a.bagon:
        from models.bagon.Bagon import Bagon
        if a.path != "engine":
            raise SystemExit("--bagon is measured on the engine path")
        model = Bagon(a.model, a.model, True, compute_dtype=dtype).to(dev)
    else:
        if a.factors > 1:
            from models.shelgon3.MultiVectorQuantizer import MultiVectorQuantizer
            vq = MultiVectorQuantizer(n_factors=a.factors, n_e=a.codes, e_dim=hidden, beta=0.25)
        else:
            vq = VectorQuantizer(n_e=a.codes, e_dim=hidden, beta=0.25)
            vq.materialize_min_encodings = False
        model = Shelgon(a.model, vq, a.model, None, compute_dtype=dtype).to(dev)
        if model.encoder.config.hidden_size != vq.e_dim:
            raise SystemExit("model hidden size must equal the codebook dimension")
    model.set_mode(a.mode)
    model.train()                      # the reference trains with dropout on (Trainer.py:310)
    ddp.broadcast_parameters(model)
    engine = None
    if a.path == "engine":
        from kvq.engine import TrainEngine
        engine = TrainEngine(model, lr=1e-4, weight_decay=0.0, amsgrad=False, milestones=[10000, 20000], gamma=0.1,
                             bucket_mib=a.bucket_mib, fp8_forward="wide" if a.fp8_wide else ("all" if a.fp8_all else a.fp8))
    else:
        params = [p for p in model.parameters() if p.requires_grad]
        opt = torch.optim.Adam(params, lr=1e-4, weight_decay=0.0, amsgrad=False, fused=True)
        sched = torch.optim.lr_scheduler.MultiStepLR(opt, milestones=[10000, 20000], gamma=0.1)
        sync = ddp.GradSync(params, bucket_mib=a.bucket_mib) if world > 1 else None

    # synthetic dSentences-like ids, resident in HBM before the timed region (BASELINE.md §3 recipe)
    gen = torch.Generator().manual_seed(69 + rank)
    pool = [tuple(t.to(dev) for t in random_token_batch(a.batch, a.seq_len, gen)) for _ in range(8)]
    if a.bagon:
        # decoder ids = the same sentences with 15 % of the tokens replaced independently (Trainer.py:94); resident in HBM as well
        from common.tensor_utils import replace_pct_rand_values
        torch.manual_seed(69 + rank)
        pool = [(ids, mask, replace_pct_rand_values(ids, 0.15, 1000, 30000) * mask, mask) for ids, mask in pool]
        pool = [(e, em, d, dm, engine.pack_batch(e, em, d, dm)) for e, em, d, dm in pool]
    elif engine is not None:
        # a tokenised batch as the input pipeline hands it over (dsentences.token_cache): ids, mask and the ids' stable order
        # (what the word-embedding gradient needs) packed into one tensor, resident in HBM like the ids themselves
        pool = [(ids, mask, engine.pack_batch(ids, mask)) for ids, mask in pool]

    def one_step(i):
        ids, mask = pool[i % len(pool)][:2]
        if a.bagon:
            _, _, d, dm, pack = pool[i % len(pool)]
            out = engine.train_step(ids, mask, prepared=pack, dec_ids=d, dec_mask=dm)
            return out["loss_recon"], 0.0
        if engine is not None:
            out = engine.train_step(ids, mask, prepared=pool[i % len(pool)][2])
            return out["loss_recon"], out["loss_vq"]           # (added on the host after the timed region: no extra launch per step)
        loss_vq, perp, _idx, loss_recon, acc, _recon = model.forward_loss(ids, mask)
        loss = loss_recon + loss_vq
        if sync is not None:
            sync.zero_grad()
        else:
            opt.zero_grad(set_to_none=False)
        loss.backward()
        if sync is not None:
            sync.finish()
        opt.step()
        sched.step()
        return loss

    for i in range(a.warmup):
        one_step(i)
    torch.cuda.synchronize()
    lib.kvq_prof_enable(a.steps + 4)
    if engine is not None:
        engine.reset_comm_timing()
    if grouped:
        dist.barrier()
    torch.cuda.synchronize()
    from kvq import nnops
    t0 = time.perf_counter()
    probe0 = nnops.clock_probe()               # two single-wave-per-workgroup stamps bracket the K steps on their stream: the shader
    for i in range(a.steps):                   # clock the chip HELD over the timed region (a few microseconds of the region itself)
        loss = one_step(a.warmup + i)
    probe1 = nnops.clock_probe()
    torch.cuda.synchronize()
    if grouped:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    final_loss = float(loss[0]) + float(loss[1]) if isinstance(loss, tuple) else float(loss)
    rccl_ranks, exposed_ms = 1, 0.0
    if grouped:
        probe = torch.ones(1, device=dev)
        dist.all_reduce(probe)                                  # the ranks that actually take part in a collective
        rccl_ranks = int(probe.item())
        if engine is not None:
            exposed_ms = engine.exposed_comm_ms() / max(a.steps, 1)

    buf = (ctypes.c_float * (a.steps + 4))()
    n_ev = lib.kvq_prof_read(buf, a.steps + 4)
    lib.kvq_prof_enable(0)

    # the same K steps with the batch BUILT inside the timed region, as the trainers' loop does per step (dsentences.token_cache:
    # two index_selects on the resident split, the ids' stable sort, one stack) -- what `prepacked_batches` leaves out of `value`
    pack_elapsed = None
    if a.pack_in_step and engine is not None and not a.bagon:
        from dsentences.token_cache import TokenCache
        cache = TokenCache.from_ids(torch.cat([p[0] for p in pool]), pad_id=0, device=dev)
        cache.packed_pad_id = engine.pad_idx
        order = torch.arange(len(cache), device=dev)
        nb = len(cache) // a.batch

        def packed_step(i):
            b = cache.batch(order[(i % nb) * a.batch:(i % nb + 1) * a.batch])
            return engine.train_step(b["input_ids"], b["attention_mask"], prepared=b["packed"])
        for i in range(3):
            packed_step(i)
        torch.cuda.synchronize()
        if grouped:
            dist.barrier()
        tp = time.perf_counter()
        for i in range(a.steps):
            packed_step(3 + i)
        torch.cuda.synchronize()
        if grouped:
            dist.barrier()
        pack_elapsed = time.perf_counter() - tp
    vq_ms = sorted(buf[i] for i in range(n_ev))
    vq_avg_ms = sum(vq_ms) / max(len(vq_ms), 1) if vq_ms else float("nan")
    # the distance / arg-min phase alone: a few more steps with the quantiser forward split into its three kernels again
    # (kvq_vq_set_variant(0): the quantiser forward is an eager interlude between the step's graphs, so nothing is re-captured);
    # the event pair then brackets the distance kernel only -- same code, same operands, same place in the step
    dist_phase_ms = None
    if engine is not None and vq is not None and not grouped and os.environ.get("KVQ_VQ_FUSED", "1") != "0" and n_ev and not a.no_distance_phase:
        lib.kvq_vq_set_variant(0)
        one_step(a.warmup + a.steps)
        torch.cuda.synchronize()
        lib.kvq_prof_enable(12)
        for i in range(8):
            one_step(a.warmup + a.steps + 1 + i)
        torch.cuda.synchronize()
        b2 = (ctypes.c_float * 12)()
        n2 = lib.kvq_prof_read(b2, 12)
        lib.kvq_prof_enable(0)
        lib.kvq_vq_set_variant(1)
        if n2:
            dist_phase_ms = sum(b2[i] for i in range(n2)) / n2

    clock_mhz, clock_per_xcd = nnops.clock_mhz(probe0, probe1)
    # kernel time per family: a few EAGER steps after the timed region with an event pair around every entry point of libkvq.so
    # (launches captured in a hipGraph cannot carry events); the empty-pair time is subtracted per launch
    families = None
    if engine is not None and a.family_steps > 0 and not grouped:
        from kvq import _ffi as ffi
        eng_graph = engine.use_graph
        engine.use_graph = False
        one_step(a.warmup + a.steps)                                       # settle into eager launches
        torch.cuda.synchronize()
        ffi.family_profile_begin()
        fp0 = nnops.clock_probe()
        for i in range(a.family_steps):
            one_step(a.warmup + a.steps + 1 + i)
        fp1 = nnops.clock_probe()
        fam_ms, fam_n, empty_ms = ffi.family_profile_end()
        engine.use_graph = eng_graph
        families = {"ms_per_step": {k: v / a.family_steps for k, v in sorted(fam_ms.items(), key=lambda kv: -kv[1])},
                    "launches_per_step": {k: v / a.family_steps for k, v in fam_n.items()},
                    "sum_ms_per_step": sum(fam_ms.values()) / a.family_steps, "empty_event_pair_us": empty_ms * 1e3,
                    # these steps run with an event pair (a queue barrier) around every launch: the chip idles between kernels and
                    # may hold another clock than in the timed region -- compare families across runs at THIS clock
                    "clock_mhz": nnops.clock_mhz(fp0, fp1)[0],
                    "method": f"{a.family_steps} eager steps after the timed region, HIP event pair around every libkvq.so entry point, "
                              "empty-pair time subtracted"}

    graph_nodes = None
    if engine is not None and engine._graphs:
        try:
            graph_nodes = next(iter(engine._graphs.values())).node_census()
        except Exception as e:                                             # a reading aid: never fails the line
            graph_nodes = f"unavailable: {e}"

    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if grouped:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    if rank == 0:
        N_tok = a.batch * a.seq_len
        D = vq.e_dim if vq is not None else hidden
        CLOCK_NOMINAL_MHZ = 2400.0                     # the clock the MFMA peaks of MI355X_MICROARCH.md are quoted at
        at_clock = (clock_mhz / CLOCK_NOMINAL_MHZ) if clock_mhz else None
        es = 2 if dtype == torch.bfloat16 else 4
        flops = 2.0 * N_tok * a.codes * D                                      # distance contraction
        alg_bytes = N_tok * (2 * D * es + 8) + a.codes * D * 4                  # read z, write z_q, write idx, codebook once
        traffic = fwd_traffic = None
        tpath = os.path.join(ROOT, "profiles", "vq_fwd_traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                traffic = tj.get(f"N{N_tok}_K{a.codes}_D{D}_{a.dtype}")
                fwd_traffic = tj.get(f"forward_kernels_N{N_tok}_K{a.codes}_D{D}_{a.dtype}")
            except Exception:
                traffic = fwd_traffic = None
        # the distance kernel's own algorithmic bytes: z once, the codebook once, one 8-byte key per token; at the L2 <-> fabric
        # level (where FETCH_SIZE counts) each of the 8 XCD L2s needs its own copy of the codebook
        dist_alg = N_tok * D * es + a.codes * D * 4 + N_tok * 8
        dist_floor = N_tok * D * es + 8 * a.codes * D * 4 + N_tok * 8
        ach_tflops = flops / (vq_avg_ms * 1e-3) / 1e12 if vq_ms else None
        # matrix-product FLOPs of one step (forward + input gradients + weight gradients of the trainable part = 3 x forward in
        # `full` mode), from the model's dimensions: every nn.Linear, the two attention products per head, the LM head
        ce, cd = model.encoder.config, model.decoder.config
        H, Fi, S = ce.hidden_size, ce.intermediate_size, a.seq_len
        per_tok_enc = ce.num_hidden_layers * (2 * (4 * H * H + 2 * H * Fi) + 4 * S * H)
        per_tok_dec = cd.num_hidden_layers * (2 * (8 * H * H + 2 * H * Fi) + 8 * S * H) + 2 * H * H + 2 * H * cd.vocab_size
        if vq is None:
            flops = 0.0
        step_flops = 3.0 * N_tok * (per_tok_enc + per_tok_dec) + 3.0 * flops
        step_s = elapsed / a.steps
        out = {
            "metric": "dSentences train sentences/sec",
            "value": world * a.batch * a.steps / elapsed,
            "unit": "sentences/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": elapsed / a.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": ("fp8 forward GEMMs / bf16" if (a.fp8 or a.fp8_all or a.fp8_wide) else "bf16") if dtype == torch.bfloat16 else "f32", "data": "synthetic",
            "config": {"workload": (f"Bagon (no quantiser, models/bagon/main.py) {a.model} enc/dec, decoder ids perturbed 15 % independently "
                                    f"of the encoder's, " if a.bagon else
                                    f"Bagon VQ (Shelgon) {a.model} enc/dec, {str(a.factors) + ' x ' if a.factors > 1 else ''}K={a.codes} D={D} ")
                                   + f"seq_len={a.seq_len} batch={a.batch}/GPU, mode={a.mode}, Adam, dropout on, path={a.path}"
                                   + (", fp8 forward GEMMs (LM head + cross-K/V)" if a.fp8_wide else
                                      ", fp8 forward GEMMs (all, behind quantisation passes)" if a.fp8_all else
                                      ", fp8 forward GEMMs (all, inputs quantised by their producers)" if a.fp8 else ""),
                       "global_batch": world * a.batch, "seq_len": a.seq_len, "parallelism": f"dp{world}"},
            "final_loss": final_loss,
            # ids + mask + the ids' stable order (what the word-embedding gradient is summed in) are built with the batch, before the
            # timed region, as dsentences.token_cache hands them over in the training loop; the step starts from one device copy
            "prepacked_batches": engine is not None,
            **({"value_pack_in_step": world * a.batch * a.steps / pack_elapsed, "ms_per_step_pack_in_step": pack_elapsed / a.steps * 1e3}
               if pack_elapsed else {}),
            # shader clock held over the timed region (kvq_clock_probe before the first and after the last step, same stream; median
            # over the XCDs): the MFMA peaks are quoted at 2400 MHz, so achieved / (peak * clock_mhz / 2400) is the fraction of what
            # the chip could deliver at the clock it actually ran
            "clock_mhz": clock_mhz, "clock_mhz_per_xcd": clock_per_xcd, "clock_nominal_mhz": CLOCK_NOMINAL_MHZ,
            **({"families_eager_event_pairs": families} if families else {}),
            "graph": bool(engine is not None and engine._graphs),      # False = the step ran as ~800 eager launches (capture failed or off)
            # what the replayed step consists of, per graph of the chain (hipGraphGetNodes + hipGraphNodeGetType): kernel nodes only,
            # no memset / memcpy node (tests/test_graph_nodes_gpu.py, profiles/r05_graph_nodes.md)
            "graph_nodes": graph_nodes,
            "rccl_ranks": rccl_ranks, "dist_backend": (dist.get_backend() if grouped else None),
            "exposed_comm_ms_per_step": exposed_ms,
            # the whole step against the dense bf16 matrix-core peak (per GPU; a reading aid: `roofline` below is the contract's object)
            "step_mfma": {"flops_per_step": step_flops, "achieved": step_flops / step_s / 1e12, "peak": BF16_MFMA_PEAK_TFLOPS,
                          "unit": "TFLOP/s", "frac": step_flops / step_s / 1e12 / BF16_MFMA_PEAK_TFLOPS,
                          "frac_at_clock": (step_flops / step_s / 1e12 / (BF16_MFMA_PEAK_TFLOPS * at_clock)) if at_clock else None}
            if a.mode == "full" else None,
            "roofline": {
                "kernel": "vq_dist_packed_kernel (distances + arg-min + gather / straight-through / loss terms / histogram + final sums: "
                          "one launch, round 5)" if os.environ.get("KVQ_VQ_FUSED", "1") != "0" else "vq_dist_packed_kernel (distances + arg-min)",
                "bound": "mfma", "achieved": ach_tflops, "peak": F32_MFMA_PEAK_TFLOPS,
                "unit": "TFLOP/s", "frac": (ach_tflops / F32_MFMA_PEAK_TFLOPS) if ach_tflops else None,
                "frac_at_clock": (ach_tflops / (F32_MFMA_PEAK_TFLOPS * at_clock)) if (ach_tflops and at_clock) else None,
                "traffic": traffic, "traffic_source": "profiles/vq_fwd_traffic.json (rocprofv3 --pmc pass of this kernel, not this run)",
                "avg_launch_us": vq_avg_ms * 1e3 if vq_ms else None, "launches": len(vq_ms),
                # the matrix phase of the same kernel, timed in 8 further steps of this run with the tail split off again
                # (profiles/r05_vq_fused.md): what `frac` read in rounds 1 - 4, when the launch held nothing else
                "distance_phase": ({"avg_launch_us": dist_phase_ms * 1e3, "achieved": flops / (dist_phase_ms * 1e-3) / 1e12,
                                    "frac": flops / (dist_phase_ms * 1e-3) / 1e12 / F32_MFMA_PEAK_TFLOPS,
                                    "frac_at_clock": (flops / (dist_phase_ms * 1e-3) / 1e12 / (F32_MFMA_PEAK_TFLOPS * at_clock)) if at_clock else None,
                                    "method": "8 steps after the timed region with kvq_vq_set_variant(0): event pair around the distance / arg-min kernel alone"}
                                   if dist_phase_ms else None),
                "flops_per_launch": flops, "algorithmic_bytes_per_launch": dist_alg, "fabric_floor_bytes_per_launch": dist_floor,
                "forward": {"algorithmic_bytes": alg_bytes, "traffic": sum(fwd_traffic.values()) if fwd_traffic else None,
                            "kernels": fwd_traffic},
                "hbm": {"achieved": alg_bytes / (vq_avg_ms * 1e-3) / 1e9 if vq_ms else None, "peak": HBM_PEAK_GBPS,
                        "unit": "GB/s", "frac": alg_bytes / (vq_avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS if vq_ms else None},
            } if vq is not None else {
                # no quantiser in the plain Bagon step: the object describes the whole step against the bf16 matrix-core peak
                "kernel": "whole step (MFMA GEMM family, csrc/kvq_gemm2.hip)", "bound": "mfma", "achieved": step_flops / step_s / 1e12,
                "peak": BF16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": step_flops / step_s / 1e12 / BF16_MFMA_PEAK_TFLOPS,
                "frac_at_clock": (step_flops / step_s / 1e12 / (BF16_MFMA_PEAK_TFLOPS * at_clock)) if at_clock else None, "traffic": None},
        }
        if world == 1 and not a.no_cpu_baseline:
            from oracle import step_oracle
            cfg = dict(BERT_BASE) if a.model == "bert-base-uncased" else None
            if cfg is None:
                from models.bagon.Bagon import LOCAL_BERT_CONFIGS
                cfg = dict(LOCAL_BERT_CONFIGS[a.model])
            cores = step_oracle.host_cores()
            print(f"[bench] timing the CPU restatement on {cores} host cores ...", file=sys.stderr, flush=True)
            r = step_oracle.time_cpu_steps(cfg, batch=8, seq_len=a.seq_len, n_e=a.codes, e_dim=D, beta=0.25,  # (--bagon: same baseline)
                                           vocab_size=model.decoder.config.vocab_size, warmup=2, steps=a.cpu_steps,
                                           threads=cores, log=lambda m: print(m, file=sys.stderr, flush=True))
            out["cpu_baseline"] = {"value": r["sentences_per_s"], "unit": "sentences/s", "cores": r["threads"], "kind": "port",
                                   "sample": f"{r['steps']} timed steps (median) of oracle/step_oracle.py at batch=8 seq_len={a.seq_len} "
                                             f"f32 (BASELINE.json configs[0]), {r['s_per_step']:.2f} s/step"}
            # BASELINE.md publishes no number for this metric; north_star's target is stated against the reference on the host's cores,
            # so the ratio to the CPU restatement timed in this run stands in (different batch sizes: that is BASELINE.json's pairing)
            out["vs_cpu_port"] = None if a.bagon else out["value"] / r["sentences_per_s"]      # (--bagon: the CPU step timed is Shelgon's)
            out["vs_cpu_port_note"] = (f"value / cpu_baseline.value: GPU step at batch {a.batch} bf16 vs CPU restatement at batch 8 f32 on "
                                       f"{r['threads']} cores (BASELINE.json configs[1] vs configs[0]) -- different batch size and dtype, "
                                       f"Adam over 248 M parameters dominates a batch-8 CPU step: NOT a like-for-like speed-up; "
                                       f"north_star target >= 10")
            if a.cpu_same_batch and not a.bagon:
                print(f"[bench] timing the CPU restatement at batch {a.batch} ...", file=sys.stderr, flush=True)
                r2 = step_oracle.time_cpu_steps(cfg, batch=a.batch, seq_len=a.seq_len, n_e=a.codes, e_dim=D, beta=0.25,
                                                vocab_size=model.decoder.config.vocab_size, warmup=1, steps=2, threads=cores, budget_s=1e9,
                                                log=lambda m: print(m, file=sys.stderr, flush=True))
                out["cpu_baseline_same_batch"] = {"value": r2["sentences_per_s"], "unit": "sentences/s", "cores": r2["threads"], "kind": "port",
                                                  "sample": f"{r2['steps']} timed steps (median) of oracle/step_oracle.py at batch={a.batch} "
                                                            f"seq_len={a.seq_len} f32, {r2['s_per_step']:.2f} s/step"}
                out["vs_cpu_port_same_batch"] = out["value"] / r2["sentences_per_s"]      # same batch; dtype still bf16 against f32
        print(json.dumps(out), flush=True)
        # a line whose number means something else than it says is worse than no line: fail the run
        import math
        if not math.isfinite(final_loss):
            print(f"[bench] final loss is {final_loss}: the step diverged, the number above is not a training step", file=sys.stderr)
            bad = 3
        elif engine is not None and not out["graph"] and os.environ.get("KVQ_GRAPH", "1") != "0" and a.warmup + a.steps > 3:
            print("[bench] the step was NOT replayed from hipGraphs (capture failed) although KVQ_GRAPH=0 was not asked for",
                  file=sys.stderr)
            # one GPU: the line would describe another program than the one that is meant to be measured -> fail.  Several ranks: the
            # eager step is the documented fallback of a capture that the collective library refused; the line says "graph": false
            bad = 4 if world == 1 else 0
        else:
            bad = 0
    else:
        bad = 0
    if grouped:
        dist.barrier()
        dist.destroy_process_group()
    if bad:
        raise SystemExit(bad)


if __name__ == "__main__":
    main()
