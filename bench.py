"""Headline benchmark: stage-3 CRDR training throughput (img/s) at 256x256 crops, bs 16 per GPU.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" = one full `MultirateBetaCondHrrGanRateDistortionTrainer.optimize_parameters` iteration (generator
forward + no-grad high-rate pass + LPIPS + discriminator passes + backward + clip + Adam for G, aux and D) on a
device-resident synthetic batch, fp32 end to end (the reference's precision).  Prints ONE JSON line on rank 0.
`roofline`: the implicit-GEMM conv kernel family (forward + input-gradient launches of igemm_kernel), algorithmic
FLOPs / HIP-event time per launch, against the dense fp32 MFMA peak.  `cpu_baseline`: the oracle's same step on
the host cores (bounded sample)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md, "Peak FP32 (matrix)"
# 3.8 F_G + 7.8 F_D + LPIPS (2 FLOP per MAC) = 644.6 GFLOP/img "as written"; the no-grad high-rate
# pass (0.8 of the steps) does not evaluate the Charm's scale transforms (8.12 GMAC) nor the hyper-decoder's scale branch
# (0.28 GMAC), whose results the reference discards: 644.6 - 0.8 * 2 * 8.40 = 631.2 GFLOP/img are actually required
GFLOP_PER_IMG_STAGE3 = 631.2
GFLOP_PER_IMG_STAGE1 = 434.0


def build_trainer(stage: int, bs: int, size: int, device: str, graphs: bool = True, precision: str = "fp32"):
    import torch
    from crdr_amd.trainer import build_trainer as _bt
    from crdr_amd.utils.options import BaseConfig, ConfigDict
    cfg, _, _ = BaseConfig._file2dict_yaml(os.path.join(ROOT, "config", f"crdr_stage_{stage}.yaml"))
    cfg.pop("pretrained_weight_path", None)
    cfg["device"] = device
    cfg["dataset"] = {"batch_size": bs, "train_dataset": {"type": "SyntheticDataset", "image_size": size}}
    cfg["path"] = None
    cfg["hip_graphs"] = graphs
    cfg["precision"] = precision
    if cfg.get("loss", {}).get("perceptual_loss"):  # throughput is weight-independent; stated in config.workload
        cfg["loss"]["perceptual_loss"]["allow_random_weights"] = True
    torch.manual_seed(0)
    return _bt(ConfigDict(cfg))


def _cpu_baseline_worker(stage: int, size: int, threads: int, budget_s: float) -> dict:
    """The oracle's training step (same losses, fwd + bwd for G and D) in stock torch on the host cores."""
    import torch
    from oracle import crdr_oracle as O
    from crdr_amd.models import build_comp_model
    from crdr_amd.models.discriminator import build_discriminator
    from crdr_amd.losses.perceptual_loss import LpipsAlex
    from crdr_amd.utils.options import BaseConfig, ConfigDict
    cfg, _, _ = BaseConfig._file2dict_yaml(os.path.join(ROOT, "config", f"crdr_stage_{stage}.yaml"))
    cfg["device"] = "cpu"
    torch.manual_seed(0)
    torch.set_num_threads(threads)
    g = {k: v.detach().clone().requires_grad_(v.is_floating_point()) for k, v in build_comp_model(ConfigDict(cfg)).state_dict().items() if v.numel() > 0}
    lp = {"lpips." + k: v.detach() for k, v in LpipsAlex().state_dict().items()}
    n = 2   # time N = 2 and scale linearly when a full batch does not fit the time budget
    x = torch.rand(n, 3, size, size) * 2 - 1
    ny, nz = torch.rand(n, 320, size // 16, size // 16) - 0.5, torch.rand(n, 192, size // 64, size // 64) - 0.5
    d = None
    if stage == 3:
        d = {k: v.detach().clone().requires_grad_(True) for k, v in build_discriminator(ConfigDict(cfg).discriminator).state_dict().items()}

    def step():
        if stage == 3:
            losses, out = O.stage3_g_losses(g, d, lp, x, 2, 2.56, ny, nz)
            losses["total"].backward()
            O.stage3_d_losses(d, x, out["fake_images"], 2)["d_total"].backward()
        else:
            losses, _ = O.stage1_losses(g, lp, x, ny, nz)
            losses["total"].backward()
    t0 = time.time()
    step()  # warm-up (allocations, oneDNN primitive creation)
    warm = time.time() - t0
    # the thread count that serves this host best: oneDNN's fork-join over these small (N = 2) convolutions stops scaling early, and the GPU
    # boxes share their cores (measured there, N = 4: 3.5 s at 16 threads, 5.2 at 32, 9.7 at 64, 26-30 at the default 128): one step each at
    # 16 / 32 / `threads`, the timed iterations at the fastest
    scan = {}
    for th in sorted({min(16, threads), min(32, threads), threads}):
        torch.set_num_threads(th)
        t0 = time.time()
        step()
        scan[th] = time.time() - t0
        if time.time() - t0 > budget_s / 2:
            break
    best = min(scan, key=scan.get)
    torch.set_num_threads(best)
    t0, k = time.time(), 0
    while k == 0 or (time.time() - t0 + warm + sum(scan.values()) < budget_s and k < 8):
        step()
        k += 1
    dt = (time.time() - t0) / k
    return {"value": round(n / dt, 4), "unit": "img/s", "cores": best, "kind": "port",
            "sample": f"oracle (stock torch fp32, oneDNN, {best} threads of {os.cpu_count()} logical CPUs: the fastest of "
                      f"{', '.join(f'{t}: {v:.1f} s' for t, v in scan.items())} per step) stage-{stage} step (G fwd + HR pass + LPIPS + D, "
                      f"all backward passes) at N={n} instead of {16 if stage == 3 else 8} images, {size}x{size}, q=2, img/s = N / step time "
                      f"(per-image cost scales linearly): {k} timed iteration(s) after 1 warm-up ({warm:.1f} s)"}


def cpu_baseline(stage: int, size: int, budget_s: float = 25.0, hard_timeout_s: float = 240.0) -> dict:
    """Runs the worker in a child process (bounded wall time; the GPU process is not disturbed by its threads)."""
    import subprocess
    # all host cores up to 64 threads: with more, oneDNN's fork-join over these small (N = 2) convolutions stops scaling --
    # measured on the 256-logical-CPU GPU box: one step does not finish within 4 minutes at 256 threads, ~2.5 s at 32-64
    threads = max(1, min(os.cpu_count() or 1, 64))
    code = (f"import json,sys; sys.path.insert(0, {ROOT!r}); import bench; "
            f"print('CPUBASE ' + json.dumps(bench._cpu_baseline_worker({stage}, {size}, {threads}, {budget_s})))")
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="", OMP_NUM_THREADS=str(threads))   # (the ceiling: the worker scans below it)
    try:
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=hard_timeout_s, env=env)
        for line in r.stdout.splitlines():
            if line.startswith("CPUBASE "):
                return json.loads(line[8:])
        return {"value": None, "unit": "img/s", "cores": threads, "kind": "port", "sample": "worker failed: " + r.stderr[-300:]}
    except subprocess.TimeoutExpired:
        return {"value": None, "unit": "img/s", "cores": threads, "kind": "port",
                "sample": f"one stage-{stage} oracle step at N=1 did not finish within {hard_timeout_s:.0f} s on {threads} threads"}


def launcher_argv(n: int, argv, port: int):
    """The command `python bench.py --gpus N` runs when no launcher set WORLD_SIZE: the driver's own torchrun line."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def self_launch(n: int, argv) -> int:
    """Start N ranks (one per GPU, RCCL over xGMI) as a child `torch.distributed.run` and relay rank 0's JSON line."""
    import socket
    import subprocess
    with socket.socket() as s:   # a free rendezvous port on the loopback interface
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // n)))
    r = subprocess.run(launcher_argv(n, argv, port), env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in r.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
    if line is None:
        sys.stdout.write(r.stdout)
        return r.returncode or 1
    print(line, flush=True)
    return r.returncode


def rccl_check(local: int):
    """What the collective backend actually is: observed world size + an all-reduce of ones over the ranks."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return {"backend": None, "world_size": 1, "allreduce_of_ones": None}
    t = torch.ones(1, device=f"cuda:{local}")
    dist.all_reduce(t)
    torch.cuda.synchronize()
    return {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "allreduce_of_ones": float(t.item())}


def _newest_profile(pattern: str):
    import glob
    fs = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)))
    return fs[-1] if fs else None


def run_stage(a, stage: int, bs: int, steps: int, warmup: int, profile_steps: int, shape_table=None, precision: str = "fp32") -> dict:
    """Build the stage's trainer, warm up (autotune + graph capture per rate index), time `steps` iterations with the rate
    index CYCLED deterministically (q = iteration mod rate levels: the expectation over the uniform draw of the reference,
    interpca_hyperprior_model.py:28-29, without sampling noise in the step mix), then `profile_steps` eager iterations
    with HIP events around every conv launch."""
    import ctypes as C
    import torch
    import torch.distributed as dist
    from crdr_amd.hip import ops
    from crdr_amd.trainer import dist as D
    ws = D.world_size()
    device = f"cuda:{torch.cuda.current_device()}"
    tr = build_trainer(stage, bs, a.size, device, graphs=not a.no_graph, precision=precision)
    loader = iter(tr.train_loader)
    lib = __import__("crdr_amd.hip.lib", fromlist=["load"]).load()
    levels = getattr(tr.comp_model, "rate_level", 0)

    def barrier():
        if ws > 1:
            dist.barrier()
        torch.cuda.synchronize()

    it = 0

    def step(q=None):
        nonlocal it
        it += 1
        d = next(loader)
        if levels:
            d = {**d, "rate_ind": (it % levels) if q is None else q}
        tr.optimize_parameters(it, d)
    # untimed preparation: autotune every shape eagerly and capture one graph set per rate index (each graph key needs
    # `graph_warmup` eager iterations first)
    for q in (range(levels) if levels else [None]):
        for _ in range(tr.graph_warmup + 1 if tr.graphs.enabled else 1):
            step(q)
    for _ in range(warmup):
        step()
    barrier()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]   # per-step device timestamps (no extra syncs): the median
    marks[0].record()
    t0 = time.perf_counter()
    for k in range(steps):
        step()
        marks[k + 1].record()   # the step's stream scope has made the current stream wait for the trainer's stream
    barrier()
    dt = time.perf_counter() - t0
    per_step = sorted(marks[k].elapsed_time(marks[k + 1]) for k in range(steps))
    median_ms = (per_step[(steps - 1) // 2] + per_step[steps // 2]) / 2 if steps else None
    # communication overlap (data parallel): timing events around every gradient all-reduce, over `comm_steps` extra graph-replayed
    # steps (the collectives run between the captured graphs, on the communication stream)
    comm = None
    if D.is_dist():
        D.TRACE = []
        for _ in range(min(10, max(steps, 1))):
            step()
        barrier()
        comm, D.TRACE = D.trace_summary(D.TRACE), None
    # per-kernel roofline: HIP events bracket every conv launch inside the library (they cannot be recorded inside a
    # graph replay), over `profile_steps` eager iterations of the same step mix right after the timed region
    graphs_on = tr.graphs.enabled
    tr.graphs.enabled = False
    lib.crdr_profile_enable(1)
    ops.PROFILE = {}
    for _ in range(profile_steps):
        step()
    torch.cuda.synchronize()
    lib.crdr_profile_enable(0)
    prof, ops.PROFILE = ops.PROFILE or {}, None
    tr.graphs.enabled = graphs_on
    if ws > 1:
        t = torch.tensor([dt], device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    def raw(kind):
        fl, ms, n = C.c_double(), C.c_double(), C.c_longlong()
        lib.crdr_profile_read(kind, C.byref(fl), C.byref(ms), C.byref(n))
        return fl.value, ms.value, n.value

    def fam(fl, ms, n):
        if n == 0 or ms <= 0:
            return None
        ps = max(profile_steps, 1)
        return {"launches_per_step": round(n / ps, 1), "avg_launch_us": round(ms * 1e3 / n, 2),
                "avg_gflop_per_launch": round(fl / n / 1e9, 3), "tflops": round(fl / (ms * 1e-3) / 1e12, 2),
                "ms_per_step": round(ms / ps, 2)}
    r0, r3, r5, r6 = raw(0), raw(3), raw(5), raw(6)   # implicit-GEMM / streaming launches, Winograd F(2x2) / F(4x4) / F(4x4)-on-5x5-stride-2 launches (all counted with the direct convolution's flops)
    r1, r4, r7, r8 = raw(1), raw(4), raw(7), raw(8)   # weight-gradient slab launches: direct, Winograd F(3x3, 2x2), F(3x3, 4x4) on 3x3 / 5x5 layers
    ig, wg, sk = (fam(r0[0] + r3[0] + r5[0] + r6[0], r0[1] + r3[1] + r5[1] + r6[1], r0[2] + r3[2] + r5[2] + r6[2]), fam(r1[0] + r4[0] + r7[0] + r8[0], r1[1] + r4[1] + r7[1] + r8[1], r1[2] + r4[2] + r7[2] + r8[2]),
                  fam(*raw(2)))
    if wg is not None and (r4[2] or r7[2] or r8[2]):
        if r4[2]:
            wg["winograd"] = dict(fam(*r4), executed_tflops=round(r4[0] / 2.25 / (r4[1] * 1e-3) / 1e12, 2))
        if r7[2]:
            wg["winograd_f3x3_4x4"] = dict(fam(*r7), executed_tflops=round(r7[0] / 4.0 / (r7[1] * 1e-3) / 1e12, 2))
        if r8[2]:
            wg["winograd_f3x3_4x4_k5"] = dict(fam(*r8), executed_tflops=round(r8[0] * 9.0 / 25.0 / (r8[1] * 1e-3) / 1e12, 2))
        wg["direct"] = fam(*r1)
        wg["executed_mfma_tflops"] = round((r1[0] + r4[0] / 2.25 + r7[0] / 4.0 + r8[0] * 9.0 / 25.0) / ((r1[1] + r4[1] + r7[1] + r8[1]) * 1e-3) / 1e12, 2)
    if ig is not None and (r3[2] or r5[2] or r6[2]):
        # what the matrix cores execute: a Winograd F(2x2, 3x3) launch does 16 multiply-accumulates per 2x2 outputs and channel pair
        # instead of 36, an F(4x4, 3x3) launch 36 per 4x4 outputs instead of 144 -- effective rates may exceed the MFMA peak, executed
        # rates may not
        if r3[2]:
            ig["winograd"] = dict(fam(*r3), executed_tflops=round(r3[0] / 2.25 / (r3[1] * 1e-3) / 1e12, 2))
        if r5[2]:
            ig["winograd_f4x4"] = dict(fam(*r5), executed_tflops=round(r5[0] / 4.0 / (r5[1] * 1e-3) / 1e12, 2))
        if r6[2]:   # a 5x5 stride-2 layer as four 3x3 sub-filters / output phases: 4 x 36 products per 16 outputs instead of 25 x 16
            ig["winograd_f4x4_k5s2"] = dict(fam(*r6), executed_tflops=round(r6[0] * 9.0 / 25.0 / (r6[1] * 1e-3) / 1e12, 2))
        ig["direct"] = fam(*r0)
        ig["executed_mfma_tflops"] = round((r0[0] + r3[0] / 2.25 + r5[0] / 4.0 + r6[0] * 9.0 / 25.0) / ((r0[1] + r3[1] + r5[1] + r6[1]) * 1e-3) / 1e12, 2)
    if ig is not None and sk is not None:
        ig["splitk_epilogue"] = {"launches_per_step": sk["launches_per_step"], "avg_launch_us": sk["avg_launch_us"], "ms_per_step": sk["ms_per_step"]}
    alg_bytes = None
    if prof.get("igemm"):
        alg_bytes = sum(r[4] for r in prof["igemm"]) / len(prof["igemm"])
    if shape_table:
        agg = {}
        for kind, rec in prof.items():
            for fl, e0, e1, label, nb in rec:
                t = agg.setdefault((kind, label), [0, 0.0, 0.0, 1e30, 0.0])
                dt_ = e0.elapsed_time(e1)
                t[0] += 1; t[1] += dt_; t[2] += fl; t[3] = min(t[3], dt_); t[4] = max(t[4], dt_)
        with open(shape_table, "w") as f:
            for (kind, label), (cnt, ms, fl, lo, hi) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
                f.write(f"{ms / max(profile_steps, 1):8.3f} ms/step  {cnt / max(profile_steps, 1):6.1f} calls/step  {fl / (ms * 1e-3) / 1e12:6.1f} TF  "
                        f"[{lo * 1e3:7.1f} .. {hi * 1e3:7.1f} us]  {kind:5s} {label}\n")
    out = {"value": ws * bs * steps / dt, "ms_per_step": dt / steps * 1e3, "median_ms": median_ms, "comm": comm, "igemm": ig, "wgrad": wg,
           "alg_bytes_per_igemm_launch": alg_bytes, "graphs": bool(graphs_on), "trainer": tr}
    return out


def fused_ops_roofline(tr) -> dict:
    """Achieved HBM bandwidth of the bandwidth-bound fused kernels at the step's own sizes, live (HIP events on the
    trainer's stream, 5 repetitions each): algorithmic bytes / time against the 8 TB/s HBM3E peak."""
    import torch
    from crdr_amd.hip import functional as HF
    from crdr_amd.hip import ops
    dev = torch.device(tr.device)
    out = {}

    side = torch.cuda.Stream()

    def timed(fn, reps=10):
        """Device time per call: `reps` invocations captured into one HIP graph and replayed (the Python / ctypes launch path
        and torch's allocator would otherwise bound these 5-50 us kernels), HIP events around the replay on its stream."""
        fn()
        torch.cuda.synchronize()
        ops.reserve_workspace(dev, side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            for _ in range(reps):
                fn()
        best = None
        for _ in range(5):   # (the fastest of five replays: these run right after the training loop, while the clocks still move)
            g.replay()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            g.replay()
            e1.record()
            torch.cuda.synchronize()
            t = e0.elapsed_time(e1) / reps * 1e-3
            best = t if best is None else min(best, t)
        return best

    def entry(name, nbytes, sec, note):
        out[name] = {"bytes": int(nbytes), "us": round(sec * 1e6, 1), "GBps": round(nbytes / sec / 1e9, 1), "frac_of_8TBps": round(nbytes / sec / 8e12, 4),
                     "what": note}
    from crdr_amd.hip import lib as L
    lib = L.load()
    opt = tr.g_optimizer
    g = next(gr for gr in opt.param_groups if gr["flat"] is not None)
    npar = g["flat"].numel()
    scratch = [g["flat"].clone(), torch.randn_like(g["flat"]) * 1e-3, torch.zeros_like(g["flat"]), torch.zeros_like(g["flat"])]
    dyn = torch.tensor([1e-4, 1.0, 0.0], device=dev)
    entry("adam_dyn_kernel", 28.0 * npar,
          timed(lambda: lib.crdr_adam_step_dyn(*[t.data_ptr() for t in scratch], npar, 0.9, 0.999, 1e-8, dyn.data_ptr(), None, 0.0, ops._stream())),
          f"fused Adam on a copy of the generator's flat buffers ({npar / 1e6:.1f} M parameters; 16 B read + 12 B written each)")
    del scratch
    pt = g["parts"][0] if g.get("parts") else None
    if pt is not None and pt.get("packs") is not None and pt["packs"].entries:
        pk = pt["packs"]
        nb = 8.0 * sum(e.T * e.rows * e.cols for e in pk.entries)
        entry("pack_weights_batched_kernel", nb, timed(pk.refill), f"all {len(pk.entries)} weight packs of the generator refilled by one launch (4 B read + 4 B written per packed element)")
    import ctypes as C

    def gc_call(n, c, h, w, noisy, want_lik):
        """crdr_gauss_cond_fwd2 on preallocated NHWC buffers (+ its fixed-order finishing pass): the launch pair the Charm issues"""
        m = n * h * w
        y, mu, sg = (torch.randn(m, c, device=dev) for _ in range(3))
        sg = sg.abs() + 0.05
        yh, lik = torch.empty(m, c, device=dev), (torch.empty(m, c, device=dev) if want_lik else None)
        bn, bq = torch.zeros(n, device=dev), torch.zeros(n, device=dev)
        ph = torch.tensor([1234, 0], dtype=torch.int64, device=dev)
        d = L.GcDesc2(N=n, HW=h * w, C=c, ldy=c, ldmu=c, ldsigma=c, ldyhat=c, ldlik=c, Ctot=c, c0=0, scale_bound=0.11, likelihood_bound=1e-9)
        io = L.GcIO(y=y.data_ptr(), mu=mu.data_ptr(), sigma=sg.data_ptr(), philox=ph.data_ptr() if noisy else None, yhat=yh.data_ptr(),
                    lik_quant=None if lik is None else lik.data_ptr(), bits_noisy=bn.data_ptr() if noisy else None, bits_quant=bq.data_ptr())
        keep = (y, mu, sg, yh, lik, bn, bq, ph)
        return (lambda: HF.gauss_cond_fwd2(d, io, dev)), keep
    fn, keep = gc_call(16, 320, 16, 16, True, False)
    entry("gauss_cond_fwd (training size)", 16.0 * 16 * 320 * 256, timed(fn),
          "16 x 320 x 16 x 16 latent, in-kernel Philox noise, both bit sums: 12 B read + 4 B written per element")
    fn, keep = gc_call(1, 320, 128, 86, False, True)
    entry("gauss_cond_fwd (2048x1365 codec size)", 20.0 * 320 * 128 * 86, timed(fn),
          "1 x 320 x 128 x 86 latent, y_hat + quantised likelihood written (12 B read + 8 B written per element), bit sum by per-block "
          "partials + fixed-order finish")
    x = torch.randn(16, 256, 128, 128, device=dev).contiguous(memory_format=torch.channels_last)
    o = torch.randn_like(x)
    s, t = torch.rand(256, device=dev) + 0.5, torch.rand(256, device=dev)
    entry("ebwd_kernel_v4 (affine + residual)", 12.0 * x.numel(),
          timed(lambda: ops.epilogue_bwd(x, o, L.EPI_AFFINE | L.EPI_RES, scale=s, shift=t, need_dz=False)),
          "InterpChAtt backward at the decoder's 256-channel 128x128 stage (read dout, read out, write the pre-affine gradient)")
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--stage", type=int, default=3)
    ap.add_argument("--bs", type=int, default=16)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the stage-1 bs 8 line and the fused-op bandwidth measurements")
    ap.add_argument("--shape-table", default=None, help="write per-shape conv timing of the timed region to this file")
    ap.add_argument("--no-autotune", action="store_true", help="use the library's built-in tile heuristic instead of timing candidates once per shape")
    ap.add_argument("--tune-log", default=None)
    ap.add_argument("--tune-db", default=None, help="perf database to preload (default: the one shipped in crdr_amd/hip); shapes it lacks are tuned live")
    ap.add_argument("--save-tune-db", default=None, help="write the tuner's choices after the run")
    ap.add_argument("--seed-tune-db", default=None, help="KINDS:PATH -- preload only these key kinds (e.g. w,wm) from a database of another "
                                                         "library version: a rebuild keeps the entries of kernels that did not change")
    ap.add_argument("--retune-k3", default=None, help="PATH -- preload a database of another library version except the 3x3 launches "
                                                      "(timed again: the Winograd kernels are new candidates for them)")
    ap.add_argument("--retune-k1", action="store_true", help="with --retune-k3: time the 1x1 launches again as well")
    ap.add_argument("--bf16x3", action="store_true", help="also time the step with precision: bf16x3 (second line `stage3_bf16x3`) even with --no-secondary")
    ap.add_argument("--bf16x6", action="store_true", help="also time the step with precision: bf16x6 (line `stage3_bf16x6`: fp32-equivalent split-bf16 products) even with --no-secondary")
    ap.add_argument("--precision", default="fp32", choices=["fp32", "bf16x3", "bf16x6"], help="matrix mode of the MAIN line (database builds of the opt-in modes: "
                                                                                            "tools/tune_bf16x6.sh); the default and the headline are exact fp32")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel from Python instead of replaying captured HIP graphs")
    ap.add_argument("--profile-steps", type=int, default=5, help="eager steps after the timed region used for the per-kernel roofline")
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start the N ranks ourselves.  This parent has not imported torch
        # or touched the GPU; the ranks run as a CHILD process tree (never exec from a GPU-initialised process).
        sys.exit(self_launch(a.gpus, sys.argv[1:]))

    # stdout carries exactly ONE line, the JSON: native libraries write to file descriptor 1 on their own (RCCL prints a version banner through C
    # stdio at communicator creation, which lands AFTER Python's line when it is flushed at exit) -- descriptor 1 is pointed at stderr for the
    # duration of the run and the line goes to the saved descriptor
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    import torch
    import torch.distributed as dist
    if torch.get_num_threads() > 16:
        # this process's host work is building the trainers (weight initialisation, packs): torch's default of one thread per physical core is the
        # slowest setting for it on the many-core GPU boxes (profiles/r6_oracle_threads.txt); never raised (torchrun sets OMP_NUM_THREADS for its ranks)
        torch.set_num_threads(16)
    from crdr_amd.hip import ops
    from crdr_amd.trainer import dist as D
    local = D.init_from_env()
    ws, rk = D.world_size(), D.rank()
    if ws != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={ws} (launch with torch.distributed.run --nproc-per-node {a.gpus}, "
                         f"or run `python bench.py --gpus {a.gpus}` without WORLD_SIZE set and let it start the ranks)")
    torch.cuda.set_device(local)
    rccl = rccl_check(local)
    ops.AUTOTUNE = not a.no_autotune  # the reference runs with cudnn.benchmark = True (base_trainer.py:20)
    if ops.AUTOTUNE and a.tune_db != "none":
        ops.load_tune_cache(a.tune_db or ops.DEFAULT_TUNE_DB)
    if ops.AUTOTUNE and a.seed_tune_db:
        kinds, path = a.seed_tune_db.split(":", 1)
        ops.load_tune_cache(path, only_kinds=tuple(kinds.split(",")), ignore_signature=True)
    if ops.AUTOTUNE and a.retune_k3:
        ops.load_tune_cache(a.retune_k3, ignore_signature=True, skip=lambda key: (3, 3) in key or (5, 5) in key or (a.retune_k1 and (1, 1) in key))   # (conv launches and weight gradients alike)
    main_run = run_stage(a, a.stage, a.bs, a.steps, a.warmup, a.profile_steps, a.shape_table, precision=a.precision)
    tr = main_run.pop("trainer")
    def save_tuning():   # (again at the end: the secondary runs tune their own shapes)
        if rk == 0 and a.save_tune_db:
            ops.save_tune_cache(a.save_tune_db)
        if a.tune_log and rk == 0:
            with open(a.tune_log, "w") as f:
                for key, best, t0, t1 in ops.TUNE_LOG:
                    f.write(f"{t0 * 1e3:9.1f}us -> {t1 * 1e3:9.1f}us  algo cfg={(best & 0xff) - 1} split={1 << ((best >> 8) & 0xf)}  {key}\n")
    save_tuning()
    if rk != 0:
        if dist.is_initialized():
            dist.destroy_process_group()
        return
    value, ig, wg = main_run["value"], main_run["igemm"], main_run["wgrad"]
    gflop = GFLOP_PER_IMG_STAGE3 if a.stage == 3 else GFLOP_PER_IMG_STAGE1
    # HBM bytes per launch of the dominant kernel come from separate rocprofv3 --pmc passes over one step (counters cannot
    # be read inside this process): tools/pmc_step.py + tools/pmc_families.py, newest result committed under profiles/
    traffic, traffic_src, traffic_nr = None, None, None
    tp = _newest_profile("r*_hbm_families.json")
    if a.stage == 3 and a.bs == 16 and a.size == 256 and tp:
        with open(tp) as f:
            doc = json.load(f)
        from crdr_amd.hip import lib as _L
        ver = int(_L.load().crdr_version())
        if doc.get("library_version") != ver:
            # counters cannot be read inside this process; a committed measurement of another library build is not evidence
            traffic_src = (f"STALE: {os.path.relpath(tp, ROOT)} was measured with library version {doc.get('library_version')}, this is {ver}: "
                           f"re-run tools/pmc_round.sh (tools/pmc_step.py + tools/pmc_families.py) and commit the result")
        else:
            fams_ = doc["families"]
            traffic = round(fams_.get("conv_fwd_dgrad", fams_["igemm_kernel"])["hbm_bytes_per_launch"])
            traffic_nr = round(fams_["conv_fwd_dgrad_no_rebuild"]["hbm_bytes_per_launch"]) if "conv_fwd_dgrad_no_rebuild" in fams_ else None
            traffic_src = (f"HBM bytes per launch, (2*FETCH_SIZE + WRITE_SIZE) KiB from two rocprofv3 --pmc passes over one eager step at q = 2 "
                           f"with this library version ({os.path.relpath(tp, ROOT)})")
    # `achieved` / `frac`: what the matrix cores EXECUTE (a Winograd launch counted with the products it really issues), so that the one-line
    # JSON cannot be read as MFMA utilisation where it is arithmetic reduction; the algorithmic (direct-convolution) rate SURVEY 8(d) defines
    # travels beside it as `achieved_effective` / `frac_effective`
    ex_tf = ig.get("executed_mfma_tflops", ig["tflops"]) if ig else None
    roof = {"bound": "mfma", "achieved": ex_tf, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(ex_tf / FP32_MFMA_PEAK_TFLOPS, 4) if ig else None,
            "achieved_effective": ig["tflops"] if ig else None, "frac_effective": round(ig["tflops"] / FP32_MFMA_PEAK_TFLOPS, 4) if ig else None,
            "traffic": traffic, "traffic_without_filter_rebuild": traffic_nr, "traffic_source": traffic_src,
            "algorithmic_bytes_per_launch": round(main_run["alg_bytes_per_igemm_launch"]) if main_run["alg_bytes_per_igemm_launch"] else None,
            "flop_convention": "dense: 2 * Cin * Cout * kh * kw per output pixel (input pixel for transposed convs), zero-padding taps at "
                               "the borders included (5x5 at 16x16: 14 % of the counted taps multiply padding)",
            "kernel": "igemm_kernel<*> + gemm1x1_kernel<*> + wino_kernel + wino4_kernel (conv / convT forward + input-gradient launches incl. grouped "
                      "ones: tiled implicit GEMM, the streaming 1x1 kernel and, where the tuner found them faster, the Winograd F(2x2,3x3) / "
                      "F(4x4,3x3) kernels with their filter transforms; v_mfma_f32_32x32x2_f32 / v_mfma_f32_16x16x4_f32)",
            "effective_rate_note": "`achieved` / `frac` are the multiply-accumulates the matrix cores EXECUTE over HIP-event time against the fp32 MFMA "
                                   "peak.  `achieved_effective` / `frac_effective` price the ALGORITHMIC (direct-convolution) flops of every launch, "
                                   "the unit SURVEY 8(d) defines: a Winograd launch executes 2.25x (F(2x2)), 4x (F(4x4)) or 25/9 x (F(4x4) on a 5x5 "
                                   "layer) fewer multiply-accumulates than it is credited with there, so the effective rate of a single Winograd "
                                   "launch may exceed 1.0.  detail.direct / detail.winograd* split the family.",
            "frac_executed": round(ig["executed_mfma_tflops"] / FP32_MFMA_PEAK_TFLOPS, 4) if ig and "executed_mfma_tflops" in ig else None,
            "measured_over": f"{a.profile_steps} eager steps after the timed region, rate index cycled (HIP events around each launch, on its stream)",
            "detail": ig, "wgrad_kernel": wg,
            "whole_step": {"algorithmic_gflop_per_img": gflop, "achieved": round(value / ws * gflop / 1e3, 2),
                           "frac": round(value / ws * gflop / 1e3 / FP32_MFMA_PEAK_TFLOPS, 4)}}
    line = {"metric": f"stage-{a.stage} training img/s at {a.size}x{a.size}", "value": round(value, 3), "unit": "img/s",
            "n_gpus": ws, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(main_run["ms_per_step"], 2),
            "value_convention": ("`value` = images / wall time over the K timed steps with the rate index CYCLED q = step mod 5 (the expectation over the reference's "
                                 "uniform draw, interpca_hyperprior_model.py:28-29; 4 of 5 steps carry the no-grad high-rate pass, so steps differ by design): "
                                 + ("K is a whole number of cycles, the mean IS the per-cycle cost" if a.steps % 5 == 0 else f"K = {a.steps} is not a whole number of cycles")
                                 + "; the median step (`ms_per_step_median`, `value_at_median`) is a q < 4 step and reads slower than the mean"),
            "ms_per_step_median": round(main_run["median_ms"], 2) if main_run["median_ms"] else None,
            "value_at_median": round(ws * a.bs / main_run["median_ms"] * 1e3, 3) if main_run["median_ms"] else None,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": a.precision, "data": "synthetic",
            "autotune": bool(ops.AUTOTUNE), "hip_graphs": main_run["graphs"], "rccl_ranks": rccl,
            "peak_device_mem_gb": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2),
            "config": {"workload": f"config/crdr_stage_{a.stage}.yaml -b {a.bs}: full GAN step (G + 5x CLIC21GVAE D + LPIPS-Alex, random-init weights), "
                                   f"rate index cycled 0..4" if a.stage == 3 else f"config/crdr_stage_1.yaml -b {a.bs}: R-D step (+LPIPS-Alex, random-init weights)",
                       "global_batch": ws * a.bs, "crop": a.size, "parallelism": f"dp{ws}"},
            "roofline": roof}
    if main_run.get("comm") is not None:
        line["comm_overlap"] = main_run["comm"]
    if ws == 1 and not a.no_secondary:
        try:
            roof["fused_ops"] = fused_ops_roofline(tr)
            fp = _newest_profile("r*_hbm_families.json")
            if fp:
                with open(fp) as f:
                    fams = json.load(f)["families"]
                roof["fused_ops"]["pmc_in_step"] = {"source": os.path.relpath(fp, ROOT) + " (HBM bytes by FETCH_SIZE / WRITE_SIZE over one eager step, kernel durations of that pass)",
                                                     **{k: {"GBps": v["achieved_GBps"], "frac_of_8TBps": v["frac_of_8TBps"], "launches": v["launches"], "ms": v["ms"]}
                                                        for k, v in fams.items() if k in ("adam_dyn", "ebwd", "wgrad_reduce_batched", "pack_weights_batched",
                                                                                          "colsum_finish", "gauss_cond_fwd", "gauss_cond_bwd", "igemm_splitk_epilogue")}}
        except Exception as e:  # the headline line must not depend on the secondary measurements
            roof["fused_ops"] = {"error": repr(e)[:300]}
        tr = None
        torch.cuda.empty_cache()
        if a.stage == 3:
            try:
                s1 = run_stage(a, 1, 8, 10, 3, 2)
                s1.pop("trainer")
                line["stage1_bs8"] = {"metric": f"stage-1 training img/s at {a.size}x{a.size}", "value": round(s1["value"], 3), "unit": "img/s",
                                      "ms_per_step": round(s1["ms_per_step"], 2), "steps": 10, "warmup": 3,
                                      "config": {"workload": "config/crdr_stage_1.yaml -b 8: R-D step (+LPIPS-Alex, random-init weights)"},
                                      "igemm_tflops": s1["igemm"]["tflops"] if s1["igemm"] else None,
                                      "whole_step_frac": round(s1["value"] * GFLOP_PER_IMG_STAGE1 / 1e3 / FP32_MFMA_PEAK_TFLOPS, 4)}
            except Exception as e:
                line["stage1_bs8"] = {"error": repr(e)[:300]}
    if ws == 1 and a.stage == 3 and (a.bf16x3 or not a.no_secondary):
        # second line, never the headline: the same step with `precision: bf16x3` (conv / weight-gradient products as split-bf16
        # triples on the bf16 matrix path, <= 3 * 2^-16 relative error per product, fp32 accumulate; everything else as above)
        try:
            tr = None
            torch.cuda.empty_cache()
            bx = run_stage(a, 3, a.bs, a.steps, a.warmup, a.profile_steps, precision="bf16x3")
            bx.pop("trainer")
            big = bx["igemm"]
            line["stage3_bf16x3"] = {
                "metric": f"stage-3 training img/s at {a.size}x{a.size}", "value": round(bx["value"], 3), "unit": "img/s",
                "ms_per_step": round(bx["ms_per_step"], 2), "steps": a.steps, "warmup": a.warmup, "dtype": "bf16x3 (fp32 accumulate)",
                "config": {"workload": f"config/crdr_stage_3.yaml -b {a.bs} + precision: bf16x3"},
                "roofline": {"bound": "mfma", "achieved": big["tflops"] if big else None, "peak": round(2500.0 / 3, 1), "unit": "TFLOP/s",
                             "frac": round(big["tflops"] / (2500.0 / 3), 4) if big else None,
                             "note": "conv forward / input-gradient family, dense fp32-equivalent FLOPs over HIP-event time; peak = dense bf16 "
                                     "MFMA peak / 3 (three bf16 MFMAs per product term); the 1x1 streaming kernel and RGB-input layers run exact fp32"},
                "wgrad_tflops": bx["wgrad"]["tflops"] if bx["wgrad"] else None,
                "parity": "tests/test_gpu_bf16x3.py: kernels within 3 * 2^-16 * sum|a||b|, step losses <= 1e-3, gradients <= 5e-3 vs the oracle"}
        except Exception as e:
            line["stage3_bf16x3"] = {"error": repr(e)[:300]}
    if ws == 1 and a.stage == 3 and (a.bf16x6 or not a.no_secondary):
        # the fp32-EQUIVALENT split: `precision: bf16x6` (three exact bf16 pieces per operand, six products of weight >= 2^-16, fp32 accumulate:
        # per-product error ~2^-23, one fp32 rounding) in the direct kernels -- tiled, streaming 1x1, direct weight gradients; the Winograd
        # kernels stay on the exact fp32 instruction and remain the tuner's choice where they are faster.  Held to the UNCHANGED fp32 parity gates
        # (tests/test_gpu_bf16x6.py); a second line, the headline stays the exact fp32 instruction.
        try:
            tr = None
            torch.cuda.empty_cache()
            b6 = run_stage(a, 3, a.bs, a.steps, a.warmup, a.profile_steps, precision="bf16x6")
            b6.pop("trainer")
            i6, w6 = b6["igemm"], b6["wgrad"]
            d6 = (i6.get("direct") or i6) if i6 else None
            line["stage3_bf16x6"] = {
                "metric": f"stage-3 training img/s at {a.size}x{a.size}", "value": round(b6["value"], 3), "unit": "img/s",
                "ms_per_step": round(b6["ms_per_step"], 2), "steps": a.steps, "warmup": a.warmup, "dtype": "bf16x6 (three exact bf16 pieces per fp32 operand, six products, fp32 accumulate)",
                "config": {"workload": f"config/crdr_stage_3.yaml -b {a.bs} + precision: bf16x6"},
                "roofline": {"bound": "mfma", "achieved": d6["tflops"] if d6 else None, "peak": round(2500.0 / 6, 1), "unit": "TFLOP/s",
                             "frac": round(d6["tflops"] / (2500.0 / 6), 4) if d6 else None,
                             "note": "DIRECT conv forward / input-gradient launches (tiled + streaming 1x1, the ones that run bf16x6), dense fp32-equivalent FLOPs "
                                     "over HIP-event time; peak = dense bf16 MFMA peak / 6 (six bf16 MFMAs per product term).  The family beside them:",
                             "family_effective_tflops": i6["tflops"] if i6 else None,
                             "winograd_f4x4": i6.get("winograd_f4x4") if i6 else None, "direct": d6},
                "wgrad": {"tflops": w6["tflops"] if w6 else None, "direct": (w6.get("direct") if w6 else None)},
                "parity": "tests/test_gpu_bf16x6.py: kernels <= 4e-6 of the output scale vs float64 (the exact-fp32 kernels' own worst) and within 2^-21 sum|a||b| "
                          "elementwise; the full-size stage-3 / stage-1 steps against the oracle at the fp32 gates of tests/test_gpu_step.py"}
        except Exception as e:
            line["stage3_bf16x6"] = {"error": repr(e)[:300]}
    if ws == 1 and a.stage == 3 and not a.no_secondary:
        # third line, for the record: the same fp32 step with the Winograd kernels taken out of the tuner's candidates (every conv on
        # the implicit-GEMM / streaming kernels, every weight gradient on the direct slab kernel) -- what the headline would be
        # without minimal filtering.  The 3x3 choices of the last database built before those kernels existed are preloaded.
        try:
            tr = None
            torch.cuda.empty_cache()
            lib_ = __import__("crdr_amd.hip.lib", fromlist=["x"]).load()
            wbase = lib_.crdr_conv2d_num_configs() + 1 + lib_.crdr_conv2d_num_stream_configs()
            wlast = lib_.crdr_conv2d_wgrad_num_configs()
            saved, saved_flag = dict(ops._algo_cache), ops.WINOGRAD
            try:
                ops.WINOGRAD = False
                for k_, v_ in list(ops._algo_cache.items()):
                    is_w = k_[0] in ("w", "wg", "ws", "wm")
                    if (is_w and (v_ & 0xff) >= wlast) or (not is_w and (v_ & 0xff) >= wbase):
                        del ops._algo_cache[k_]
                ops.load_tune_cache(os.path.join(ROOT, "tools", "data", "tune_r3_f.json"), ignore_signature=True)
                dx = run_stage(a, 3, a.bs, min(a.steps, 20), a.warmup, 0)
                dx.pop("trainer")
            finally:   # (also when the run raises: the process goes on with the headline's plan set, and --save-tune-db writes that one)
                ops._algo_cache.clear(); ops._algo_cache.update(saved); ops.WINOGRAD = saved_flag
            line["stage3_direct_only"] = {"metric": f"stage-3 training img/s at {a.size}x{a.size}", "value": round(dx["value"], 3), "unit": "img/s",
                                          "ms_per_step": round(dx["ms_per_step"], 2), "steps": min(a.steps, 20), "warmup": a.warmup, "dtype": "fp32",
                                          "config": {"workload": f"config/crdr_stage_3.yaml -b {a.bs}, CRDR_WINOGRAD=0 (no minimal-filtering kernels)"}}
        except Exception as e:
            line["stage3_direct_only"] = {"error": repr(e)[:300]}
    if ws == 1 and a.stage == 3 and not a.no_secondary:
        # fourth line: the same fp32 step with only the F(4x4, 3x3) / F(3x3, 4x4) kernels taken out (F(2x2) + direct kernels: the round-3 plan
        # set) -- what the round-4/5 kernels are worth on their own
        try:
            tr = None
            torch.cuda.empty_cache()
            lib_ = __import__("crdr_amd.hip.lib", fromlist=["x"]).load()
            w4 = lib_.crdr_conv2d_num_configs() + 1 + lib_.crdr_conv2d_num_stream_configs() + 2
            wg4 = lib_.crdr_conv2d_wgrad_num_configs() + 1
            saved = dict(ops._algo_cache)
            saved_w4 = ops.WINO4
            try:
                ops.WINO4 = False
                # the plan set tuned WITHOUT those kernels (tools/tune_round.sh, CRDR_WINO4=0: the tuner's choice among F(2x2) and the direct kernels for
                # every 3x3 / 5x5 shape); whatever it lacks keeps the shipped choice unless that is an F(4x4) id
                ops._algo_cache.clear()
                ops.load_tune_cache(os.path.join(ROOT, "tools", "data", "tune_r5_no_f4x4.json"), ignore_signature=True)
                for k_, v_ in saved.items():
                    is_w = k_[0] in ("w", "wg", "ws", "wm")
                    if not ((is_w and (v_ & 0xff) == wg4) or (not is_w and (v_ & 0xff) == w4)):
                        ops._algo_cache.setdefault(k_, v_)
                nx = run_stage(a, 3, a.bs, min(a.steps, 20), a.warmup, 0)
                nx.pop("trainer")
            finally:
                ops._algo_cache.clear(); ops._algo_cache.update(saved); ops.WINO4 = saved_w4
            line["stage3_no_f4x4"] = {"metric": f"stage-3 training img/s at {a.size}x{a.size}", "value": round(nx["value"], 3), "unit": "img/s",
                                      "ms_per_step": round(nx["ms_per_step"], 2), "steps": min(a.steps, 20), "warmup": a.warmup, "dtype": "fp32",
                                      "config": {"workload": f"config/crdr_stage_3.yaml -b {a.bs}, F(4x4, 3x3) / F(3x3, 4x4) kernels off (F(2x2) + direct)"}}
        except Exception as e:
            line["stage3_no_f4x4"] = {"error": repr(e)[:300]}
    if ops.TUNE_REJECTED:   # candidates the tuner refused because they disagreed with the built-in plan (a silently de-tuned database is visible)
        line["tune_rejected"] = {"count": len(ops.TUNE_REJECTED), "first": [[repr(k), int(a_), float(f"{d:.3g}")] for k, a_, d in ops.TUNE_REJECTED[:8]]}
    if ws == 1 and not a.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(a.stage, a.size)
    save_tuning()
    json_out.write(json.dumps(line) + "\n")
    json_out.flush()
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
