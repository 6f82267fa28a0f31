#!/usr/bin/env python
"""Headline benchmark: image-pairs/s of the DUSt3R two-view forward, ViT-L/16 encoder + 12-block CroCo
cross-attention decoder + DPT pointmap heads + adaptor, 512x512 pairs, bf16 MFMA operands (BASELINE.json configs[1]).

    python bench.py --gpus 1 --steps 10 --warmup 3
    python bench.py --gpus 8 --steps 10 --warmup 3         (no torchrun: bench.py starts its own ranks, one process per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One process per GPU; every rank runs `--pairs` independent image pairs per step (weak scaling, no data-path
collective: pairs are independent in the forward).  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_BF16_TFLOPS = 2500.0  # dense bf16 MFMA peak of one MI355X (MI355X_MICROARCH.md §Chip-level parameters)
PEAK_HBM_GBS = 8000.0      # HBM3E peak (same guide)
PMC_TRAFFIC_FILE = "r6_pmc_traffic.json"
FWD_PAIRS = 128            # pairs per GPU per forward step: 56 GB; 64 -> 96 -> 128 pairs measured +0.7 / +0.9 % (same box), 256: see DESIGN section 7
TRAIN_PAIRS = 64           # pairs per GPU per training step: 216 GB of the 288 GB at 512x512 with DPT heads (32: 96-98 pairs/s, 64: 100-101, 80: 101.9 at 267 GB)


def gflop_enc_dec(img, patch=16, enc_dim=1024, enc_depth=24, dec_dim=768, dec_depth=12, n_extra=0):
    """Forward GFLOP of encoder + decoder per image PAIR (both views), SURVEY.md §8d:
    enc = depth.N.(24 D^2 + 4 N D) + 2.3.P^2.D.N per view; dec = depth.(N.28 D^2 + Nk.4 D^2 + 4 N^2 D + 4 N Nk D) + 2.Din.D.N per view.
    512^2 -> 2068.0, 224^2 -> 340.0, 1024^2 -> 12601.4, DINOv2 ViT-L/14 518^2 (n_extra = 1 cls token in the encoder) -> 2926.3."""
    n = (img // patch) ** 2
    ne = n + n_extra
    enc = enc_depth * ne * (24 * enc_dim ** 2 + 4 * ne * enc_dim) + 2 * 3 * patch ** 2 * enc_dim * n
    dec = dec_depth * (n * 28 * dec_dim ** 2 + n * 4 * dec_dim ** 2 + 4 * n * n * dec_dim + 4 * n * n * dec_dim) + 2 * enc_dim * dec_dim * n
    return 2.0 * (enc + dec) / 1e9


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--mode", default="fwd", choices=["fwd", "train"],
                    help="fwd: forward inference (BASELINE configs[1], the headline); train: forward + backward + gradient "
                         "all-reduce + AdamW step (BASELINE configs[2])")
    ap.add_argument("--pairs", type=int, default=None, help="image pairs per GPU per step (default FWD_PAIRS fwd, TRAIN_PAIRS train)")
    ap.add_argument("--encoder", default="croco", choices=["croco", "dinov2"],
                    help="croco: the DUSt3R factory model; dinov2: BASELINE configs[3] — DINOv2 ViT-L/14 encoder (frozen in "
                         "train mode) + the same decoder and heads, default 518x518")
    ap.add_argument("--img", type=int, default=None)
    ap.add_argument("--head", default="dpt", choices=["dpt", "linear"])
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--attention", default="bf16", choices=["bf16", "fp8"],
                    help="fp8: e4m3 K=64 MFMA attention kernel (BASELINE configs[4]; forward only)")
    ap.add_argument("--graph", action="store_true", help="replay the forward from a captured hipGraph (latency mode, small --pairs)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-reference-policy", action="store_true",
                    help="skip the extra legs: fp32-class heads beside the bf16 transformer, everything fp32-class, linear-head (enc+dec) run")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-extra-legs", action="store_true",
                    help="skip the `fwd_224` (224x224 pairs), `batch_sweep` (1 / 2 / 4 / 8 pairs), `other_configs` (BASELINE configs[3], [4]) and `train_step` (BASELINE configs[2]) legs of the default line")
    ap.add_argument("--train-pairs", type=int, default=TRAIN_PAIRS, help="pairs per GPU of the `train_step` leg inside the forward line")
    ap.add_argument("--train-steps", type=int, default=3, help="timed steps of the `train_step` leg")
    ap.add_argument("--single-stream", action="store_true",
                    help="no two-stream execution of independent sub-graphs at large batch (engine.concurrent(False)): what the "
                         "roofline pass and the committed kernel profiles use — per-kernel durations are only defined without overlap")
    ap.add_argument("--cpu-baseline-max-s", type=float, default=45.0)
    ap.add_argument("--sweep", default=None,
                    help="comma-separated batch sizes (pairs), e.g. 1,2,4,8,16,64: adds `batch_sweep` to the JSON line — the forward at each "
                         "size, timed like the reference's own harness (examples/models/dust3r/profile_dust3r.py:31-46: randn images, "
                         "no_grad, Timer.blocked_autorange().mean), eager and (up to 16 pairs) replayed from a hipGraph")
    return ap.parse_args()


def make_views(B, H, W, rank, dev):
    g = torch.Generator().manual_seed(1000 + rank)
    img1 = torch.randn(B, 3, H, W, generator=g).to(dev)
    img2 = torch.randn(B, 3, H, W, generator=g).to(dev)
    v1 = {"img": img1, "instance": [f"a{rank}_{i}" for i in range(B)], "data_norm_type": "dust3r"}
    v2 = {"img": img2, "instance": [f"b{rank}_{i}" for i in range(B)], "data_norm_type": "dust3r"}  # not symmetrized
    return v1, v2


def roofline_pass(step_fn, v1, precision, steps):
    """One instrumented single-stream pass: every launch of the path's kernel families is bracketed with HIP events on the launch
    stream (PyTorch's current stream — the one every uc_hip call is handed) and its ALGORITHMIC work related to the measured time.
    Returns (roofline of the dominant family: the dense bf16 MFMA GEMM, roofline_families: attention / 3x3 convolutions against the
    bf16 MFMA peak, the bandwidth-bound kernels against the HBM peak)."""
    from uniception_amd import ops

    rec = {}          # family -> list of (e0, e1, flops, bytes)
    saved = {}

    def bracket(family, fn, work):
        def wrapped(*args, **kw):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = fn(*args, **kw)
            e1.record()
            fam, fl, by = work(out, *args, **{k_: v_ for k_, v_ in kw.items() if k_ != "out"})
            rec.setdefault(fam or family, []).append((e0, e1, fl, by))
            return out
        return wrapped

    def nbytes(*ts):
        return float(sum(t.numel() * t.element_size() for t in ts if isinstance(t, torch.Tensor)))

    def gemm_work(out, a, w, *args, **kw):
        N_, K_ = w.shape
        if kw.get("conv") is not None:
            M_ = out.numel() // (4 if kw.get("tail") is not None else N_)
            return "conv3x3", 2.0 * M_ * N_ * K_, nbytes(a, w, out)
        if a.dtype not in (torch.bfloat16, torch.float16):     # (fp16: the heads' 1x1 / ConvTranspose GEMMs, same kernels)
            return "gemm_other", 2.0 * a.shape[0] * N_ * K_, nbytes(a, w, out)
        M_ = a.shape[0]
        by = 2.0 * (M_ * K_ + N_ * K_) + nbytes(out)                     # A + W + C ...
        if kw.get("residual") is not None:
            by += nbytes(kw["residual"])                                 # ... + the residual read by the epilogue
        if kw.get("vt") is not None:
            by += 2.0 * M_ * (N_ - out.shape[1])                         # ... + the V columns stored as packed VT
        side = getattr(out, "uc_ln", None)
        if side is not None:
            by += 8.0 * M_ * (N_ // 64)                                  # ... + the row statistics of the LayerNorm fold (producer)
            if side.twin is not out:
                by += 2.0 * M_ * N_                                      # ... + the bf16 twin of an fp32 stream (a bf16 stream is its own twin)
        if kw.get("ln") is not None:
            by += 8.0 * M_ + 4.0 * N_                                    # ... + (mean, rstd) per row and the column sums (consumer)
        return "dense_gemm", 2.0 * M_ * N_ * K_, by

    def attn_work(out, q, k, v, *args, **kw):
        B_, Nq, H_, D_ = q.shape
        Nk = k.shape[1]
        return None, 4.0 * B_ * H_ * Nq * Nk * D_, nbytes(out) + 2.0 * q.element_size() * B_ * H_ * D_ * (Nq / 2 + Nk)   # Q + K + V read, O written

    def attn_bwd_work(res, q, k, v, o, do, *args, **kw):     # the five algorithmic products (2.5 x the forward), whatever the kernels recompute
        B_, Nq, H_, D_ = q.shape
        Nk = k.shape[1]
        return None, 10.0 * B_ * H_ * Nq * Nk * D_, nbytes(q, k, v, o, do, *res)

    def gemm_tn_work(res, a, b, *args, **kw):                # weight gradients: sum over T tokens / output pixels of a[t,i] b[t,j]
        T_, I_ = a.shape
        conv = kw.get("conv")
        J_ = 9 * b.shape[-1] if conv is not None else b.shape[1]
        ws = res[0] if isinstance(res, tuple) else res
        return ("conv3x3_wgrad" if conv is not None else "gemm_wgrad"), 2.0 * T_ * I_ * J_, nbytes(a, b, ws)

    def io_work(out, x, *args, **kw):
        outs = out if isinstance(out, (tuple, list)) else (out,)
        return None, 0.0, nbytes(x, *outs)

    plan = {"gemm": ("dense_gemm", gemm_work), "attention": ("attention_fwd", attn_work), "attention_fp8": ("attention_fwd_fp8", attn_work),
            "bilinear_nhwc": ("bilinear", io_work), "pointmap_adaptor": ("adaptor", io_work), "adaptor_program": ("adaptor", io_work),
            "patch_gather": ("patch_gather", io_work), "convert": ("convert", io_work), "convt_scatter": ("convt_scatter", io_work),
            "pixel_shuffle": ("pixel_shuffle", io_work), "conv1x1_to4": ("conv1x1_to4", io_work), "layernorm": ("layernorm", io_work),
            "nchw_to_nhwc": ("layout", io_work), "nhwc_to_nchw": ("layout", io_work),
            # the training step's own families
            "attention_bwd": ("attention_bwd", attn_bwd_work), "gemm_tn": ("gemm_wgrad", gemm_tn_work)}
    for name, (fam, work) in plan.items():
        saved[name] = getattr(ops, name)
        setattr(ops, name, bracket(fam, saved[name], work))
    try:
        for _ in range(steps):
            step_fn()
        torch.cuda.synchronize()
    finally:
        for name, fn in saved.items():
            setattr(ops, name, fn)
    traffic, traffic_note = None, "no PMC record for this build"
    try:  # HBM bytes per launch from the committed PMC passes (profiles/): only when they were taken on this workload AND on
        # this very build of the kernels (fingerprint of csrc/ + flags recorded next to the counters)
        from uniception_amd import build as _build
        with open(os.path.join(ROOT, "profiles", PMC_TRAFFIC_FILE)) as f:
            pmc = json.load(f)
        c = pmc["config"]
        if (c["pairs_per_gpu"], c["img"], c["precision"]) != (v1["img"].shape[0], v1["img"].shape[-1], precision):
            traffic_note = "PMC record is for another workload"
        elif pmc.get("kernel_fingerprint") != _build.loaded_fingerprint():
            traffic_note = "PMC record was taken on another build of the kernels (stale): not reported"
        else:
            traffic, traffic_note = pmc["traffic_bytes_per_launch"], f"profiles/{PMC_TRAFFIC_FILE}"
    except (OSError, KeyError, ValueError):
        pass

    def summarize(records):
        t = sum(r[0].elapsed_time(r[1]) for r in records) * 1e-3
        return t, sum(r[2] for r in records), sum(r[3] for r in records), len(records)

    t, fl, alg_bytes, n = summarize(rec["dense_gemm"]) if rec.get("dense_gemm") else (1.0, 0.0, 0.0, 1)
    roof = {"bound": "mfma", "achieved": round(fl / t / 1e12, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
            "frac": round(fl / t / 1e12 / PEAK_BF16_TFLOPS, 4), "traffic": traffic, "traffic_source": traffic_note,
            "kernel": "gemm_bf16_glds_kernel<.., dense, ..> + gemm_bf16_glds8_kernel + gemm_bf16_glds4_kernel + gemm_bf16_co_kernel (dense bf16 MFMA GEMM, all tile variants and epilogue families)",
            "launches_per_step": n // steps, "avg_launch_us": round(t / n * 1e6, 2),
            "algorithmic_gflop_per_launch": round(fl / n / 1e9, 2), "algorithmic_bytes_per_launch": int(alg_bytes / n)}
    fams = {}
    total_t = sum(summarize(r)[0] for r in rec.values())
    for fam, records in sorted(rec.items()):
        ft, ffl, fby, fn_ = summarize(records)
        e = {"launches_per_step": fn_ // steps, "ms_per_step": round(ft / steps * 1e3, 3), "share_of_bracketed_time": round(ft / total_t, 4)}
        if ffl > 0:   # matrix-pipe families: algorithmic flops against the dense bf16 MFMA peak
            e.update({"bound": "mfma", "achieved": round(ffl / ft / 1e12, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                      "frac": round(ffl / ft / 1e12 / PEAK_BF16_TFLOPS, 4)})
        else:         # data movement: algorithmic bytes (tensors read + written once) against the HBM peak
            e.update({"bound": "hbm", "achieved": round(fby / ft / 1e9, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                      "frac": round(fby / ft / 1e9 / PEAK_HBM_GBS, 4)})
        fams[fam] = e
    return roof, fams


def cpu_model_name():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown CPU"


def cpu_topology():
    """(sockets, physical cores, logical cpus) from /proc/cpuinfo (falls back to os.cpu_count())."""
    phys, sockets, logical = set(), set(), 0
    try:
        pid = cid = None
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("processor"):
                    logical += 1
                elif line.startswith("physical id"):
                    pid = line.split(":")[1].strip(); sockets.add(pid)
                elif line.startswith("core id"):
                    cid = line.split(":")[1].strip(); phys.add((pid, cid))
    except OSError:
        pass
    n = os.cpu_count() or 1
    return max(1, len(sockets)), (len(phys) or n), (logical or n)


def _read(path):
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return None


def _cpulist(txt):
    "'0-3,8,10-11' -> [0, 1, 2, 3, 8, 10, 11]"
    out = []
    for part in (txt or "").split(","):
        part = part.strip()
        if not part:
            continue
        if "-" in part:
            a, b = part.split("-")
            out.extend(range(int(a), int(b) + 1))
        else:
            out.append(int(part))
    return out


def host_cpu_limits():
    """What bounds a CPU job in this container, beyond the core count of /proc/cpuinfo: the affinity mask, the cgroup CPU quota (v2
    cpu.max or v1 cfs_quota / cfs_period: quota / period = the number of CPUs' worth of time the cgroup may use), the NUMA nodes."""
    aff = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else list(range(os.cpu_count() or 1))
    quota = None
    v2 = _read("/sys/fs/cgroup/cpu.max")
    if v2:
        q, per = (v2.split() + ["100000"])[:2]
        quota = None if q == "max" else round(float(q) / float(per), 2)
    else:
        q, per = _read("/sys/fs/cgroup/cpu/cpu.cfs_quota_us"), _read("/sys/fs/cgroup/cpu/cpu.cfs_period_us")
        if q and per and int(q) > 0:
            quota = round(float(q) / float(per), 2)
    nodes = {}
    base = "/sys/devices/system/node"
    try:
        for n in sorted(os.listdir(base)):
            if n.startswith("node") and n[4:].isdigit():
                nodes[int(n[4:])] = _cpulist(_read(f"{base}/{n}/cpulist"))
    except OSError:
        pass
    siblings = {}
    for c in aff:
        t = _read(f"/sys/devices/system/cpu/cpu{c}/topology/thread_siblings_list")
        siblings[c] = min(_cpulist(t)) if t else c
    return {"affinity_cpus": len(aff), "cgroup_cpu_quota": quota, "numa_nodes": {str(k): len(v) for k, v in nodes.items()}}, aff, nodes, siblings


def cpu_baseline(model, H, W, head, max_s):
    """The oracle (CPU restatement of the reference path, validated against the real reference in tests/golden) timed on the host
    cores on ONE pair — a bounded sample of the same workload.  torch's default thread count (= logical CPUs) oversubscribes
    a multi-socket SMT host (round 3: 128 threads of an EPYC 9575F ran 4x slower than 8 Xeon vCPUs), and threads spread over both
    sockets pay the inter-socket fabric for every GEMM panel (round 4: 12 threads were the best of a sweep cut at 24).  So (round
    5): the process is PINNED to the allowed CPUs of one NUMA node, one hardware thread per core (os.sched_setaffinity), the thread
    count is swept over {16, 32, 64, cores of that node} on the ENCODER of one view (24 of the pair's 60 transformer blocks: a sixth
    of the forward, so the whole sweep fits the time cap), and the full pair is then timed n >= 3 times at the winner (median).
    The line carries what bounds the job here — affinity mask, cgroup CPU quota, NUMA nodes — so that a low figure can be read."""
    from oracle import dust3r_oracle as O

    sd = {k: v.detach().float().cpu() for k, v in model.state_dict().items()}
    img1, img2 = O.make_images(7, 1, H, W)
    sockets, phys, logical = cpu_topology()
    limits, aff, nodes, siblings = host_cpu_limits()
    default_threads = torch.get_num_threads()
    # one NUMA node's allowed CPUs, one hardware thread per physical core
    node_cpus = max(({n: [c for c in cpus if c in set(aff)] for n, cpus in nodes.items()} or {0: list(aff)}).items(), key=lambda kv: len(kv[1]))
    pin = sorted({siblings.get(c, c) for c in node_cpus[1]} & set(aff)) or list(aff)
    t_begin = time.perf_counter()
    pinned = False
    if hasattr(os, "sched_setaffinity"):
        try:
            os.sched_setaffinity(0, pin)
            pinned = True
        except OSError:
            pass
    avail = len(pin) if pinned else len(aff)
    if limits["cgroup_cpu_quota"]:
        avail = max(1, min(avail, int(limits["cgroup_cpu_quota"] + 0.5)))
    cands = sorted({c for c in (8, 16, 32, 64, avail) if 1 <= c <= avail}) or [1]

    def enc_one(nthreads):
        torch.set_num_threads(nthreads)
        with torch.no_grad():
            t0 = time.perf_counter()
            O.croco_encoder(img1, sd, "encoder.", depth=24, num_heads=16)
            return time.perf_counter() - t0

    def pair_one(nthreads):
        torch.set_num_threads(nthreads)
        with torch.no_grad():
            t0 = time.perf_counter()
            O.dust3r_forward(sd, img1, img2, head=head)
            return time.perf_counter() - t0

    sweep = {}
    try:
        enc_one(cands[-1])                # warm-up (allocator, oneDNN primitive caches)
        for c in reversed(cands):          # (largest first: the likely winner is measured even if the cap cuts the sweep)
            if time.perf_counter() - t_begin > 0.4 * max_s and sweep:
                break
            sweep[c] = enc_one(c)
        best = min(sweep, key=sweep.get)
        times = []
        while len(times) < 3 or (len(times) < 5 and time.perf_counter() - t_begin < max_s):
            times.append(pair_one(best))
    finally:
        torch.set_num_threads(default_threads)
        if pinned:
            try:
                os.sched_setaffinity(0, aff)
            except OSError:
                pass
    times.sort()
    t = times[len(times) // 2]
    return {"value": round(1.0 / t, 4), "unit": "image-pairs/s", "cores": best, "cpu": cpu_model_name(), "kind": "port",
            "n_samples": len(times), "host": {"sockets": sockets, "physical_cores": phys, "logical_cpus": logical, **limits,
                                              "pinned_to": f"{len(pin)} CPUs of NUMA node {node_cpus[0]} (one per core)" if pinned else "not pinned"},
            "thread_sweep_s_per_encoder_view": {str(k): round(v, 2) for k, v in sorted(sweep.items())},
            "sample": f"1 pair (2x{H}x{W}) fp32 forward incl. heads, pinned to one NUMA node: thread sweep {sorted(sweep)} on the encoder of one view, "
                      f"then n={len(times)} full forwards at the best count ({best} threads, median {t:.2f}s each), torch CPU"}


def timed(step, steps, world):
    """K steps between barrier + synchronize fences; max over ranks."""
    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = step()
    fence()
    return time.perf_counter() - t0, out


def _host_events():
    "Counters whose change inside a timed step names a one-time stall: caching-allocator traffic to the driver and Python's cyclic GC."
    import gc
    ms = torch.cuda.memory_stats() if torch.cuda.is_available() else {}
    return {"device_allocs": ms.get("num_device_alloc", 0), "device_frees": ms.get("num_device_free", 0),
            "alloc_retries": ms.get("num_alloc_retries", 0), "gc_gen2": gc.get_stats()[2]["collections"]}


def timed_each(step, steps):
    """The legs beside the headline: every step fenced and timed on its own; the MEAN over the K steps is reported (the headline's
    contract, and the reference harness's `blocked_autorange().mean`, utils/profile.py:4-6), the median and the sorted times beside it.
    Round 5 reported medians because a block of K steps right after a model switch now and then contained one step 60-250 ms long.
    Round 6 attached `stall_events` (per-step deltas of the caching allocator's hipMalloc / hipFree counts and of Python's
    generation-2 collections) and a `settle()` (collect + synchronize) before every leg's timed steps.  Root cause, caught with the
    counters (profiles/r6_z_bench_forward_pairs128.json, train_step: 595.9 / 682.6 / 596.8 ms, the slow step the one with
    `device_allocs: 1`): the caching allocator going to the driver for a new segment INSIDE a step — a hipMalloc with 216 of the 288 GB
    in use costs tens of ms; the same event on an emptier device costs nothing (most legs show it with no effect).  Python's
    collector is not it (generation-2 collections inside steps: no effect).  The leg's number stays the all-inclusive mean; the mean
    of the steps without allocator traffic is printed beside it."""
    ts, evs = [], []
    for _ in range(steps):
        e0 = _host_events()
        d1, _ = timed(step, 1, 1)
        e1 = _host_events()
        ts.append(d1)
        evs.append({k: e1[k] - e0[k] for k in e0 if e1[k] != e0[k]})
    mean = sum(ts) / len(ts)
    order = sorted(ts)
    med = order[len(order) // 2]
    st = {"n": steps, "ms_per_step_mean": round(mean * 1e3, 2), "ms_per_step_median": round(med * 1e3, 2),
          "ms_per_step_in_order": [round(t * 1e3, 1) for t in ts], "reported": "mean"}
    if any(evs):
        st["stall_events"] = evs
        quiet = [t for t, e in zip(ts, evs) if not e.get("device_allocs") and not e.get("device_frees")]
        if quiet and len(quiet) < len(ts):      # (for the reader: the steady state beside the all-inclusive mean the leg reports)
            st["ms_per_step_mean_without_allocator_growth"] = round(sum(quiet) / len(quiet) * 1e3, 2)
    return mean * steps, st


def settle():
    """Before a leg's timed steps: run Python's cyclic collector now (modules of the previous leg die in reference cycles: collected
    later, inside somebody's timed step, their tensors go back to the caching allocator mid-step) and let queued frees land."""
    import gc
    if os.environ.get("UNICEPTION_AMD_BENCH_NO_SETTLE", "0") == "1":      # (A/B of the root cause: profiles/r6_*stall*)
        return
    gc.collect()
    torch.cuda.synchronize()


def reference_policy_legs(model, v1, v2, args, dev):
    """Beside the headline (bf16 transformer + TF32-class heads: fp16 MFMA operands carry TF32's 10-bit mantissa, which is what the
    reference's fp32 heads — factory/dust3r.py:288-309 — multiply with in its own environment, allow_tf32 in libs/croco/blocks.py:15).
    (a) bf16 transformer + fp32-class heads (split bf16 operands on the matrix pipe: ~1e-5 from exact fp32) and + bf16 heads (round
    1-2's headline policy) at the headline batch; (b) EVERYTHING fp32-class
    (engine.precision("bf16x3"): split-operand GEMMs / convolutions / attention products on the matrix pipe) — the mode that meets
    the 1e-3 / 1e-2 gate (tests/test_precision_modes_gpu.py) — at the headline batch; (c) the encoder + decoder alone (linear
    head, 0.15 % of the FLOPs), the quantity the 40 % MFMA target is defined on; (d) the headline forward with an fp32 residual
    stream instead of the reference's bf16 one.  Every leg: steps fenced one by one, the MEAN reported (median beside it: timed_each)."""
    from uniception_amd import engine
    from uniception_amd.models.factory import DUSt3R
    out = {}
    steps = max(2, min(args.steps, 5))

    def fwd(vv1, vv2, mode, m=model):
        def f():
            with torch.no_grad(), engine.precision(mode), engine.attention_precision(args.attention):
                return m(vv1, vv2)
        return f

    def leg(f, n, extra):
        f(); f()          # (the first call of a shape runs its fork points one after the other)
        settle()
        dt, st = timed_each(f, n)
        e = {"pairs_per_s": round(args.pairs * n / dt, 2), "ms_per_step": round(dt / n * 1e3, 2), "pairs_per_gpu": args.pairs, "timing": st}
        e.update(extra)
        return e, args.pairs * n / dt
    if args.head == "dpt" and args.encoder == "croco":
        torch.manual_seed(0)
        lin = DUSt3R(name="bench_linear", img_size=(args.img, args.img), pred_head_type="linear").to(dev).eval()
        e, pps = leg(fwd(v1, v2, "bf16", lin), steps, {})
        e["enc_dec_mfma_frac"] = round(pps * gflop_enc_dec(args.img) / 1e3 / PEAK_BF16_TFLOPS, 4)
        e["enc_dec_mfma_frac_from"] = "mean step time x SURVEY section 8d flops per pair"
        out["enc_dec_linear_head"] = e
        del lin          # (no torch.cuda.empty_cache() here or anywhere between legs: on some boxes of the pool the leg that runs on freshly
                         #  hipMalloc'ed blocks reads 15-25 % low — 345 instead of 470 pairs/s for this one, 213 instead of 248 for the next —
                         #  while legs that reuse the caching allocator's blocks do not)
    with engine.head_precision("fp32"):
        out["bf16_transformer_fp32class_heads"], _ = leg(fwd(v1, v2, "bf16"), steps, {"heads": "bf16x3 split-operand MFMA, fp32 tensors (~1e-5 from exact fp32 heads)"})
    with engine.head_precision("follow"):     # round 1-2's headline policy: heads in the transformer's bf16
        out["bf16_transformer_bf16_heads"], _ = leg(fwd(v1, v2, "bf16"), steps, {"heads": "bf16 operands and maps (1.7e-2 from exact fp32 heads)"})
    if engine.bf16_stream_enabled() and args.encoder == "croco":
        # the same bf16 forward with the residual stream kept in fp32 (round 1's policy: more accurate than the reference's own bf16
        # stream under autocast, 10 instead of 4 bytes per element in the residual epilogues)
        with engine.bf16_stream(False):
            out["bf16_operands_fp32_residual_stream"], _ = leg(fwd(v1, v2, "bf16"), steps, {})
    out["everything_fp32class"], _ = leg(fwd(v1, v2, "bf16x3"), 3, {
        "mode": "bf16x3: every GEMM / convolution and both products of the attention as three bf16 MFMA products of "
                "split operands, fp32 accumulate, fp32 softmax, fp32 tensors",
        "meets": "rel-L2 < 1e-3 and max-abs < 1e-2 vs reference (tests/test_precision_modes_gpu.py)"})
    return out


def fwd_224_leg(args, dev, pairs_list=(64, 256)):
    """north_star's second workload: 224x224 pairs through the same ViT-L/16 + 12-block decoder + DPT model (the reference's own
    small-image case, encoders/utils.py:57-59), forward, bf16, the headline's policy.  Timed by the same fences as the headline."""
    from uniception_amd import engine
    from uniception_amd.models.factory import DUSt3R
    torch.manual_seed(0)
    m = DUSt3R(name="bench_224", img_size=(224, 224), pred_head_type=args.head).to(dev).eval()
    out = {"workload": f"ViT-L/16 CroCo encoder + 12-block decoder + {args.head} head, 224x224 pairs, forward, bf16", "by_pairs_per_gpu": []}
    gf = gflop_enc_dec(224)
    for b in pairs_list:
        a1, a2 = make_views(b, 224, 224, 0, dev)

        def f():
            with torch.no_grad(), engine.precision("bf16"), engine.attention_precision(args.attention):
                return m(a1, a2)
        f(); f(); f()
        n = 5
        settle()
        dt, st = timed_each(f, n)
        pps = b * n / dt
        out["by_pairs_per_gpu"].append({"pairs_per_gpu": b, "pairs_per_s": round(pps, 1), "ms_per_step": round(dt / n * 1e3, 3), "timing": st,
                                        "enc_dec_gflop_per_pair": round(gf, 1),
                                        "enc_dec_mfma_frac_lower_bound": round(pps * gf / 1e3 / PEAK_BF16_TFLOPS, 4)})
        del a1, a2
    del m
    return out


def other_configs_leg(args, dev):
    """BASELINE configs[3] and [4] inside the default line (round 5): DINOv2 ViT-L/14 at 518x518 (32 pairs) and the ViT-L/16 model at
    1024x1024 (8 pairs) with bf16 AND e4m3 attention — forward, bf16 transformer, the headline's head policy; five fenced steps each
    (mean reported, median beside it); `enc_dec_mfma_frac_lower_bound` = all-in pairs/s x the encoder + decoder flops of SURVEY section 8d over
    the bf16 peak (the heads' time is in the denominator, their flops are not in the numerator)."""
    from uniception_amd import engine
    from uniception_amd.models.encoders import encoder_factory
    from uniception_amd.models.factory import DUSt3R
    out = {}

    def run(m, img, pairs, attention, norm_type, gf):
        a1, a2 = make_views(pairs, img, img, 0, dev)
        a1["data_norm_type"] = a2["data_norm_type"] = norm_type

        def f():
            with torch.no_grad(), engine.precision("bf16"), engine.attention_precision(attention):
                return m(a1, a2)
        f(); f(); f()
        n = 5
        settle()
        dt, st = timed_each(f, n)
        pps = pairs * n / dt
        del a1, a2
        return {"pairs_per_gpu": pairs, "pairs_per_s": round(pps, 2), "ms_per_step": round(dt / n * 1e3, 2), "attention": attention, "timing": st,
                "enc_dec_gflop_per_pair": round(gf, 1), "enc_dec_mfma_frac_lower_bound": round(pps * gf / 1e3 / PEAK_BF16_TFLOPS, 4)}

    torch.cuda.empty_cache()
    torch.manual_seed(0)
    m = DUSt3R(name="bench_c3", img_size=(518, 518), pred_head_type=args.head)
    m.encoder = encoder_factory("dinov2", name="bench_dinov2", size="large")
    m = m.to(dev).eval()
    e = run(m, 518, 32, "bf16", "dinov2", gflop_enc_dec(518, patch=14, n_extra=1))
    e["workload"] = f"BASELINE configs[3]: DINOv2 ViT-L/14 encoder + 12-block CroCo decoder + {args.head} head, 518x518 pairs (37x37 patches + cls token), forward, bf16"
    out["dinov2_518"] = e
    del m
    torch.cuda.empty_cache()
    torch.manual_seed(0)
    m = DUSt3R(name="bench_c4", img_size=(1024, 1024), pred_head_type=args.head).to(dev).eval()
    for att in ("bf16", "fp8"):
        e = run(m, 1024, 8, att, "dust3r", gflop_enc_dec(1024))
        e["workload"] = (f"BASELINE configs[4]: ViT-L/16 encoder + 12-block decoder + {args.head} head, 1024x1024 pairs (4096 tokens), forward, bf16 transformer, "
                         f"{'e4m3 K=64 MFMA' if att == 'fp8' else 'bf16'} attention")
        out[f"vitl_1024_{att}_attention"] = e
    out["vitl_1024_fp8_over_bf16"] = round(out["vitl_1024_fp8_attention"]["pairs_per_s"] / out["vitl_1024_bf16_attention"]["pairs_per_s"], 3)
    del m
    torch.cuda.empty_cache()
    return out


def exchange_summary(stats, step_ms):
    """The multi-GPU part of a training leg: what `GradientBuckets.comm_stats()` measured, in the names the line carries.
    `rccl_ranks` is read back from the process group (dist.get_world_size()), not from --gpus."""
    return {"rccl_ranks": stats["ranks"], "backend": stats["backend"], "buckets": stats["buckets"], "bucket_bytes": stats["bucket_bytes"],
            "comm_ms_per_step": stats.get("comm_ms"), "exposed_comm_ms_per_step": stats.get("exposed_ms"),
            "overlapped_frac": stats.get("overlapped_frac"), "busbw_GBps": stats.get("busbw_GBps"),
            "exposed_frac_of_step": (round(stats["exposed_ms"] / step_ms, 4) if stats.get("exposed_ms") is not None and step_ms else None),
            "steps_measured": stats["steps"],
            "definition": "comm = sum over buckets of (all-reduce end - the point every producer of the bucket was done), events on the "
                          "communication stream; exposed = what the compute stream waited for the exchange after the backward's last kernel"}


def train_step_leg(args, dev, pairs=TRAIN_PAIRS, steps=3, rank=0, world=1):
    """BASELINE configs[2] inside the default line: forward + backward + gradient exchange + AdamW of the same ViT-L + DPT model
    at 512x512, `pairs` pairs per rank.  world == 1: `steps` individually fenced steps (mean reported, median beside it), with its own
    dense-GEMM roofline; the exchange is a no-op.  world > 1 (the driver's `bench.py --gpus N`): EVERY rank runs it — `Trainer` over the
    RCCL group, parameters broadcast from rank 0, K steps between barrier + synchronize fences, max over ranks — and the leg reports
    the bucketed all-reduce it ran (`exchange`: bucket count / bytes, communication per step, the part the backward did not hide)."""
    from uniception_amd import autograd, engine
    from uniception_amd.models.factory import DUSt3R
    from uniception_amd.training import Trainer
    torch.cuda.empty_cache()                 # the forward legs' cached blocks: the step below wants most of the HBM
    torch.cuda.reset_peak_memory_stats()
    torch.manual_seed(0)
    m = DUSt3R(name="bench_train", img_size=(args.img, args.img), pred_head_type=args.head).to(dev).train()
    trainer = Trainer(m, lr=1e-5, weight_decay=0.05)
    trainer.broadcast_parameters(0)
    a1, a2 = make_views(pairs, args.img, args.img, rank, dev)
    g = torch.Generator().manual_seed(2000 + rank)
    gt1 = torch.randn(pairs, args.img, args.img, 3, generator=g).to(dev)
    gt2 = torch.randn(pairs, args.img, args.img, 3, generator=g).to(dev)

    def step():
        trainer.zero_grad()
        with engine.precision("bf16"):
            r1, r2 = m(a1, a2)
            loss = autograd.conf_loss(r1["pts3d"], r1["conf"], gt1) + autograd.conf_loss(r2["pts3d_in_other_view"], r2["conf"], gt2)
        loss.backward()
        trainer.step()
        return loss.detach()
    step(); step(); step()      # (three: the allocator's segments of a two-stream training step settle late — see timed_each)
    exchange = None
    if world > 1:
        from uniception_amd.distributed import max_over_ranks
        trainer.enable_comm_timing(True)
        dt, loss = timed(step, steps, world)
        dt = max_over_ranks(dt, dev)
        st = {"n": steps, "reported": "K steps between barrier + synchronize fences, max over ranks"}
        exchange = exchange_summary(trainer.comm_stats(), dt / steps * 1e3)
        trainer.enable_comm_timing(False)
    else:
        settle()
        dt, st = timed_each(step, steps)
        loss = step()
    assert torch.isfinite(loss).all()
    pps = world * pairs * steps / dt
    gf = gflop_enc_dec(args.img)
    out = {"workload": f"BASELINE configs[2]: ViT-L/16 + 12-block decoder + {args.head} head, {args.img}x{args.img} pairs, forward + backward + AdamW, "
                       "synthetic pointmap targets, " + ("1 rank (the gradient exchange is a no-op at world 1)" if world == 1 else
                                                         f"{world} ranks (replicated model, bucketed in-place gradient all-reduce, weak scaling)"),
           "pairs_per_s": round(pps, 2), "ms_per_step": round(dt / steps * 1e3, 2), "pairs_per_gpu": pairs, "n_gpus": world, "timing": st,
           "peak_hbm_gb": round(torch.cuda.max_memory_allocated() / 1e9, 1),
           "heads": engine.train_head_dtype_name() if hasattr(engine, "train_head_dtype_name") else "bf16 kernels (forward and backward)",
           "streams": "2 (encoder views, decoder branches, heads: forward and backward)" if (engine.CONCURRENT and engine.TRAIN_CONCURRENT) else "1",
           "residual_stream": "bf16 (the reference's stream under autocast)" if engine._bf16_train_stream else "fp32",
           "enc_dec_mfma_frac_lower_bound": round(pps / world * gf * 3 / 1e3 / PEAK_BF16_TFLOPS, 4),
           "note": "fraction = pairs/s x 3 x forward enc+dec flops / peak: charges heads, optimizer and the whole step to the enc+dec flops"}
    if exchange is not None:
        out["exchange"] = exchange
    if not args.no_roofline and world == 1:
        with engine.concurrent(False):
            step()      # (untimed: the first single-stream step after two-stream ones draws fresh blocks from hipMalloc — inside the brackets otherwise)
            torch.cuda.synchronize()
            roof, fams = roofline_pass(step, a1, "bf16", 2)
        roof["traffic"], roof["traffic_source"] = None, "not collected for the training step"
        out["roofline"] = roof
        out["roofline_families"] = fams
    if world == 1 and not args.no_reference_policy and pairs >= 4:
        # VERDICT r5 #4e: the reference trains its heads in fp32 (TF32 under allow_tf32, libs/croco/blocks.py:15; dust3r.py:288-309) —
        # the same step with fp32-class heads (fp32 tensors, split bf16 operands forward and backward) beside the bf16-head step, both at
        # a quarter of the leg's batch (fp32 head maps + their split copies of the full batch do not fit beside the step's other tensors)
        import contextlib
        try:
            del a1, a2, gt1, gt2
            torch.cuda.empty_cache()
            b = max(1, pairs // 4)
            a1, a2 = make_views(b, args.img, args.img, rank, dev)
            g2 = torch.Generator().manual_seed(3000 + rank)
            gt1 = torch.randn(b, args.img, args.img, 3, generator=g2).to(dev)
            gt2 = torch.randn(b, args.img, args.img, 3, generator=g2).to(dev)
            pol = {"pairs_per_gpu": b, "note": "same model and trainer, a quarter of the leg's batch; mean of 2 fenced steps after 2 warm-up steps"}
            for name, mode in (("bf16_heads", None), ("fp32class_heads", "fp32")):
                with (engine.head_precision(mode) if mode else contextlib.nullcontext()):
                    step(); step()
                    settle()
                    d2, t2 = timed_each(step, 2)
                    pol[name] = {"pairs_per_s": round(b * 2 / d2, 2), "ms_per_step": round(d2 / 2 * 1e3, 2), "heads": engine.train_head_dtype_name(), "timing": t2}
                torch.cuda.empty_cache()
            pol["fp32class_over_bf16_heads"] = round(pol["fp32class_heads"]["pairs_per_s"] / pol["bf16_heads"]["pairs_per_s"], 4)
            out["reference_policy_heads"] = pol
        except Exception as e:      # (an extra leg must not take the line down: reported)
            out["reference_policy_heads"] = {"error": f"{type(e).__name__}: {e}"[:400]}
    del trainer, m, a1, a2, gt1, gt2
    torch.cuda.empty_cache()
    return out


def self_launch(args):
    """`python bench.py --gpus N` from a plain shell (no torchrun): start the N ranks ourselves — one process per GPU through
    torch.distributed.run on 127.0.0.1 and a free port, this very command line behind it — and pass the children's output (rank 0's
    ONE JSON line) and exit status through.  Under torchrun (WORLD_SIZE set) this is never reached."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // args.gpus)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def launch_check(rank, world):
    """UNICEPTION_AMD_BENCH_LAUNCH_CHECK=1 (tests, no GPU needed): rendezvous over gloo, the fences and the max-over-ranks reduction of
    the timing contract, rank 0's line — everything of the N > 1 control flow but the model.  The line says what it is."""
    from uniception_amd.distributed import max_over_ranks
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
        dist.barrier()
    t0 = time.perf_counter()
    time.sleep(0.01 * (rank + 1))
    dt = max_over_ranks(time.perf_counter() - t0)
    line = {"launch_check": True, "n_gpus": world, "max_rank_s": round(dt, 4), "note": "control flow only: not a measurement"}
    if world > 1:
        # the exchange the N > 1 line reports, on a toy CPU model: the same FlatParameters / GradientBuckets / comm_stats /
        # exchange_summary code the GPU ranks run (Trainer's optimizer kernel needs a GPU; the bucketed all-reduce does not)
        from uniception_amd.training import FlatParameters, GradientBuckets
        torch.manual_seed(3)
        toy = torch.nn.Sequential(torch.nn.Linear(6, 32), torch.nn.GELU(), torch.nn.Linear(32, 32), torch.nn.GELU(), torch.nn.Linear(32, 3))
        flat = FlatParameters(toy, 2048)
        buckets = GradientBuckets(flat)
        buckets.enable_comm_timing(True)
        x = torch.randn(5, 6, generator=torch.Generator().manual_seed(100 + rank))
        ts = time.perf_counter()
        for _ in range(2):
            flat.zero_grad()
            buckets.start_step()
            toy(x).square().mean().backward()
            buckets.finish()
        step_ms = (time.perf_counter() - ts) / 2 * 1e3
        line["rccl_ranks"] = dist.get_world_size()
        line["backend"] = dist.get_backend()
        line["train_step"] = {"workload": "toy CPU model over gloo: the exchange code path only", "n_gpus": world,
                              "exchange": exchange_summary(buckets.comm_stats(), step_ms)}
        dist.barrier()
    if rank == 0:
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


def batch_sweep(model, sizes, args, dev):
    """The reference's benchmark loop (profile_dust3r.py:31-46, utils/profile.py:4-6) over this build: batch sizes {1, 2, 4, 8} there."""
    import torch.utils.benchmark as tbench
    from uniception_amd import engine
    out = []
    for b in sizes:
        v1, v2 = make_views(b, args.img, args.img, 0, dev)

        def f():
            with torch.no_grad(), engine.precision(args.precision), engine.attention_precision(args.attention):
                return model(v1, v2)
        f(); f()
        m = tbench.Timer(stmt="f()", globals={"f": f}).blocked_autorange(min_run_time=1.0)
        e = {"pairs": b, "ms_per_batch": round(m.mean * 1e3, 3), "pairs_per_s": round(b / m.mean, 2)}
        if b <= 16:
            from uniception_amd.graphs import GraphedTwoView
            g = GraphedTwoView(model, v1, v2, precision=args.precision, attention=args.attention)
            g(v1, v2)
            mg = tbench.Timer(stmt="g(v1, v2)", globals={"g": g, "v1": v1, "v2": v2}).blocked_autorange(min_run_time=1.0)
            e["hipgraph_ms_per_batch"] = round(mg.mean * 1e3, 3)
            e["hipgraph_pairs_per_s"] = round(b / mg.mean, 2)
            del g
        out.append(e)
        del v1, v2          # (no empty_cache(): see reference_policy_legs)
    return out


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    if os.environ.get("UNICEPTION_AMD_BENCH_LAUNCH_CHECK", "0") == "1":
        return launch_check(rank, world)
    if not (os.environ.get("UNICEPTION_AMD_BENCH_SHARE_GPU", "0") == "1") and torch.cuda.device_count() < min(world, local_rank + 1):
        raise SystemExit(f"bench.py: rank {rank} needs cuda:{local_rank} but this node shows {torch.cuda.device_count()} GPU(s)")
    # UNICEPTION_AMD_BENCH_SHARE_GPU=1: a dry run of the N > 1 control flow on a box with fewer GPUs than ranks (ranks share
    # devices, gloo instead of RCCL — two ranks cannot open one device in an RCCL communicator); the line it prints is marked
    share = os.environ.get("UNICEPTION_AMD_BENCH_SHARE_GPU", "0") == "1"
    dev_index = local_rank % torch.cuda.device_count() if share else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    from uniception_amd import _lib, engine
    from uniception_amd.models.factory import DUSt3R

    _lib.load()  # fail loudly if the HIP extension is missing
    torch.manual_seed(0)
    if args.pairs is None:
        args.pairs = FWD_PAIRS if args.mode == "fwd" else TRAIN_PAIRS
    if args.img is None:
        args.img = 512 if args.encoder == "croco" else 518
    model = DUSt3R(name="bench", img_size=(args.img, args.img), pred_head_type=args.head)
    norm_type = "dust3r"
    if args.encoder == "dinov2":   # the factory hard-codes the CroCo encoder (factory/dust3r.py:115-122): swap it
        from uniception_amd.models.encoders import encoder_factory
        model.encoder = encoder_factory("dinov2", name="bench_dinov2", size="large")
        model.encoder.requires_grad_(False)
        norm_type = "dinov2"
    model = model.to(dev)
    v1, v2 = make_views(args.pairs, args.img, args.img, rank, dev)
    v1["data_norm_type"] = v2["data_norm_type"] = norm_type

    if args.mode == "train":
        from uniception_amd import autograd
        from uniception_amd.training import Trainer
        model.train()
        trainer = Trainer(model, lr=1e-5, weight_decay=0.05)
        trainer.broadcast_parameters(0)
        g = torch.Generator().manual_seed(2000 + rank)
        gt1 = torch.randn(args.pairs, args.img, args.img, 3, generator=g).to(dev)
        gt2 = torch.randn(args.pairs, args.img, args.img, 3, generator=g).to(dev)

        def step():
            trainer.zero_grad()
            with engine.precision(args.precision):
                r1, r2 = model(v1, v2)
                loss = autograd.conf_loss(r1["pts3d"], r1["conf"], gt1) + autograd.conf_loss(r2["pts3d_in_other_view"], r2["conf"], gt2)
            loss.backward()
            trainer.step()
            return ({"pts3d": loss.detach().reshape(1)},)
    else:
        model.eval()

        if args.graph:
            from uniception_amd.graphs import GraphedTwoView
            graphed = GraphedTwoView(model, v1, v2, precision=args.precision, attention=args.attention)

            def step():
                return graphed(v1, v2)
        else:
            def step():
                with torch.no_grad(), engine.precision(args.precision), engine.attention_precision(args.attention):
                    return model(v1, v2)

    if args.single_stream:
        engine.CONCURRENT = False
    for _ in range(args.warmup):
        step()

    if args.mode == "train" and world > 1:
        trainer.enable_comm_timing(True)
    dt, out = timed(step, args.steps, world)
    if world > 1:
        from uniception_amd.distributed import max_over_ranks
        dt = max_over_ranks(dt, dev)
    assert torch.isfinite(out[0]["pts3d"]).all()

    pairs_total = world * args.pairs * args.steps
    value = pairs_total / dt
    # encoder + decoder flops per pair from the SURVEY section 8d formulas (exact at every image size: attention's N^2 term included)
    gflop_pair = gflop_enc_dec(args.img) if args.encoder == "croco" else gflop_enc_dec(args.img, patch=14, n_extra=1)
    fwd = args.mode == "fwd"
    enc_name = "ViT-L/16" if args.encoder == "croco" else "DINOv2 ViT-L/14"
    line = {
        "metric": (f"image-pairs/sec fwd, {enc_name} two-view {args.img}x{args.img} (encoder + CroCo decoder + {args.head} heads + adaptor)" if fwd else
                   f"image-pairs/sec fwd+bwd training step, {enc_name} two-view {args.img}x{args.img} (forward, backward, gradient all-reduce, AdamW)"),
        "value": round(value, 3), "unit": "image-pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.precision, "data": "synthetic",
        "config": {"workload": (f"BASELINE configs[{(1 if fwd else 2) if args.encoder == 'croco' else 3}]: "
                                f"{'ViT-L/16 CroCo' if args.encoder == 'croco' else 'DINOv2 ViT-L/14'} encoder + 12-block CroCo decoder + {args.head} head, "
                                f"{args.img}x{args.img} pairs, " + ("forward" if fwd else "training step with synthetic pointmap targets")
                                + ", random-init weights"),
                   "pairs_per_gpu": args.pairs, "global_pairs_per_step": world * args.pairs, "img": args.img,
                   "head": args.head, "encoder": args.encoder, "attention": args.attention, "hipgraph": bool(args.graph),
                   "streams": ("1" if (not engine.CONCURRENT or (not fwd and not engine.TRAIN_CONCURRENT)) else
                               "2 (the two views through the encoder, the two decoder branches and the two heads run as concurrent HIP streams"
                               + ("" if fwd else ", forward and backward") + ")"),
                   "heads": (("TF32-class: fp16 MFMA operands (10-bit mantissa), fp32 accumulate, fp32 final layer + adaptor — the reference's "
                              "fp32 heads under allow_tf32 (2e-3 from exact-fp32 heads; bf16 heads: 1.7e-2)") if (fwd and args.precision == "bf16" and engine.head_dtype_name() == "fp16")
                             else (engine.head_dtype_name() if fwd else
                                   (engine.train_head_dtype_name() if hasattr(engine, "train_head_dtype_name") else "bf16 kernels (forward and backward)"))),
                   "residual_stream": ("bf16 (the reference's stream under autocast: bf16 sub-layer outputs added to a bf16 x)"
                                       if (fwd and args.precision == "bf16" and args.encoder == "croco" and engine.bf16_stream_enabled())
                                       else "fp32"),
                   "parallelism": (f"dp{world} (independent pairs per rank, no data-path collective)" if fwd else
                                   f"dp{world} (replicated model, bucketed in-place gradient all-reduce over RCCL)")},
        "enc_dec_mfma_frac": round(value / world * gflop_pair * (1 if fwd else 3) / 1e3 / PEAK_BF16_TFLOPS, 4),
        "enc_dec_gflop_per_pair": round(gflop_pair, 1),
        "peak_hbm_gb": round(torch.cuda.max_memory_allocated() / 1e9, 1),
    }
    if world > 1:
        # the multi-GPU line measures the multi-GPU design: forward pairs/s above is weak scaling with no data-path
        # collective; the bucketed gradient all-reduce only exists in the training step, so at N > 1 EVERY rank also runs it
        line["rccl_ranks"] = dist.get_world_size()
        line["backend"] = dist.get_backend()
        if not fwd:
            line["exchange"] = exchange_summary(trainer.comm_stats(), dt / args.steps * 1e3)
        elif not args.no_extra_legs and args.precision == "bf16" and not args.graph:
            del out
            try:
                leg = train_step_leg(args, dev, pairs=args.train_pairs, steps=args.train_steps, rank=rank, world=world)
            except Exception as e:      # (the forward's line must survive a failure of the extra leg: reported, not raised)
                leg = {"error": f"{type(e).__name__}: {e}"[:500], "n_gpus": world}
            line["train_step"] = leg
    if share:
        line["config"]["shared_gpu_dry_run"] = "ranks share devices, gloo process group: control-flow check only, not a measurement"
    if rank == 0 and world == 1 and not fwd and not args.no_roofline and args.precision == "bf16":
        # training: the same kernel family carries the forward and the data-gradient GEMMs (the TN weight-gradient kernel is
        # a separate, smaller share): all dense bf16 uc_gemm launches of a step, forward and backward
        with engine.concurrent(False):
            step()
            torch.cuda.synchronize()
            line["roofline"], line["roofline_families"] = roofline_pass(step, v1, args.precision, min(args.steps, 2))
    if rank == 0 and world == 1 and fwd:
        if not args.no_roofline and args.precision == "bf16":
            with engine.concurrent(False):   # per-launch durations are only defined when kernels do not overlap
                step()                       # (untimed: the allocator's blocks of the side streams are not reusable here — first step mallocs)
                torch.cuda.synchronize()
                line["roofline"], line["roofline_families"] = roofline_pass(step, v1, args.precision, min(args.steps, 3))
            line["roofline"]["schedule"] = ("single-stream pass (engine.concurrent(False), = bench.py --single-stream, the command of the "
                                            "committed profiles); the timed region runs two kernel streams")
        if not args.no_reference_policy and args.precision == "bf16" and not args.graph:
            line["reference_policy"] = reference_policy_legs(model, v1, v2, args, dev)
        if (not args.no_extra_legs and args.precision == "bf16" and not args.graph and args.encoder == "croco" and args.img == 512
                and args.attention == "bf16"):
            line["fwd_224"] = fwd_224_leg(args, dev)
            line["batch_sweep"] = batch_sweep(model, [1, 2, 4, 8], args, dev)      # the reference harness's own sizes (profile_dust3r.py:12-46)
            line["other_configs"] = other_configs_leg(args, dev)
            line["train_step"] = train_step_leg(args, dev, pairs=args.train_pairs, steps=args.train_steps)
        if args.sweep and "batch_sweep" not in line:
            line["batch_sweep"] = batch_sweep(model, [int(x) for x in args.sweep.split(",")], args, dev)
        if not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(model, args.img, args.img, args.head, args.cpu_baseline_max_s)
    if rank == 0:
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
