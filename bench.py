#!/usr/bin/env python3
"""bench.py - queries/sec of brute-force top-10 over an [N x 768] theorem-embedding matrix.

Contract (one JSON line on rank 0):
  python bench.py --gpus N --steps K --warmup W          (any N: for N > 1 this process starts the N ranks itself as
                                                          child processes, one per GPU - launch_ranks)
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   (N > 1 under a launcher: the same ranks)

Workload (BASELINE.json): configs[2] "10M x 768 bf16 corpus, batch-256 queries" - the shape the
north-star target is quoted on; it fits one GPU (15.36 GB).  One step = one batch of 256 queries
searched against the whole corpus (inputs resident in HBM).  With N > 1 the SAME corpus is
row-sharded over the ranks (strong scaling): every rank searches its shard, the per-shard top-10
are exchanged with one RCCL all-gather, and every rank merges them.
Other workloads: --workload c2 (1M x 768 fp32, batch 1), --workload c4 (50M x 768 bf16, batch 256).

Synthetic data: corpus chunk c (250,000 rows) = default_rng([1234, c]).standard_normal, rows
L2-normalised in fp32, rounded to bf16 for the bf16 workloads (synthetic.synth_chunk); queries from
default_rng([5678, 0]) the same way.  Rows are stored as given (metric = inner product on
normalised rows = cosine).
"""
import argparse
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (rows, dtype, nq)
    "c1": (1_000, "f32", 73),          # compare_embeddings.evaluate_retrieval end to end on ~1k theorems (BASELINE.json configs[0])
    "c2": (1_000_000, "f32", 1),
    "c2b": (1_000_000, "f32", 256),    # the same fp32 corpus, batch 256: the exact-fp32 MFMA path (kernels_mfma16.h, F32 mode)
    "c3": (10_000_000, "bf16", 256),
    "c4": (50_000_000, "bf16", 256),
    "c5": (10_000_000, "bf16", 256),   # encoder-in-loop: sentence-encoder forward feeds the C3 index
    # the production table of the reference: theorem_embedding_qwen, vector(1024) (streamlit_app.py:49,55, rds_schema.sql:50-56);
    # not a BASELINE.json config - the same batch-256 search at d = 1024 (--dim defaults to 1024 here)
    "c3q": (10_000_000, "bf16", 256),
}
D = 768   # overridden by --dim (diagnostic: the production table of the reference is vector(1024), rds_schema.sql:50)
K = 10
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "f32": 157.3}   # MI355X_MICROARCH.md: dense MFMA peaks (spec)
MFMA_RANDOM_DATA_GEMM_TFLOPS = 1247.0   # MI355X_MICROARCH.md "DVFS give-back" item 1: a bf16 GEMM on random data under DVFS


def log(rank, *a):
    if rank == 0:
        print("[bench]", *a, file=sys.stderr, flush=True)


class PowerSampler:
    """Package power and shader clock of one GPU, sampled from a side thread while a leg runs (the sustained leg, never the
    contract's timed K steps): the amdgpu hwmon files of the device (`power1_average` / `power1_input` in microwatts,
    `power1_cap`, `freq1_input` in Hz) every 50 ms, or - where the files cannot be read - `rocm-smi --showpower
    --showclocks --json` as often as it answers.  No GPU call, no effect on the stream being timed.
    Evidence for DESIGN.md section 3.2: the full pass of configs[2] holds the package at its power limit while the shader
    clock sits well below its maximum."""

    def __init__(self, pci_bus_id=None, device_index=0):
        import glob
        self.device_index = device_index
        self.samples = []          # (t, watts, sclk_mhz)
        self.cap_w = None
        self.source = None
        self._stop = False
        self._thread = None
        self._hw = None
        cands = []
        if pci_bus_id:
            cands += glob.glob(f"/sys/bus/pci/devices/{pci_bus_id.lower()}/hwmon/hwmon*")
        if not cands:
            cards = sorted(glob.glob("/sys/class/drm/card[0-9]*/device/hwmon/hwmon*"))
            if len(cards) == 1 or (cards and not pci_bus_id):
                cands = cards[device_index:device_index + 1] or cards[:1]
        for hw in cands:
            for name in ("power1_average", "power1_input"):
                if os.access(os.path.join(hw, name), os.R_OK):
                    self._hw, self._pfile = hw, os.path.join(hw, name)
                    self.source = f"hwmon {name}"
                    break
            if self._hw:
                break
        if self._hw:
            try:
                self.cap_w = int(open(os.path.join(self._hw, "power1_cap")).read()) / 1e6
            except (OSError, ValueError):
                self.cap_w = None

    def _read_hwmon(self):
        w = int(open(self._pfile).read()) / 1e6
        try:
            mhz = int(open(os.path.join(self._hw, "freq1_input")).read()) / 1e6
        except (OSError, ValueError):
            mhz = None
        return w, mhz

    def _read_smi(self):
        import subprocess
        out = subprocess.run(["rocm-smi", "-d", str(self.device_index), "--showpower", "--showclocks", "--showmaxpower", "--json"],
                             capture_output=True, text=True, timeout=10).stdout
        card = next(iter(json.loads(out).values()))
        w = mhz = None
        for k_, v in card.items():
            kl = k_.lower()
            try:
                if "power" in kl and "max" in kl:
                    self.cap_w = float(v)
                elif "power" in kl and "(w)" in kl:
                    w = float(v)
                elif kl.startswith("sclk clock speed"):
                    mhz = float(str(v).strip("()").lower().replace("mhz", ""))
            except ValueError:
                pass
        if w is None:
            raise RuntimeError("rocm-smi reported no power")
        return w, mhz

    def _run(self):
        read = self._read_hwmon if self._hw else self._read_smi
        if not self._hw:
            self.source = "rocm-smi --showpower --showclocks"
        while not self._stop:
            try:
                w, mhz = read()
                self.samples.append((time.time(), w, mhz))
            except Exception:                      # noqa: BLE001 - a sampler that cannot read stops quietly
                if not self.samples:
                    self.source = None
                return
            time.sleep(0.05)

    def __enter__(self):
        import threading
        self._thread = threading.Thread(target=self._run, daemon=True)
        self._thread.start()
        return self

    def __exit__(self, *exc):
        self._stop = True
        self._thread.join(timeout=15)
        return False

    def summary(self, t0, t1):
        """Mean over the samples taken inside [t0 + 20 %, t1] (the clock needs a moment to settle under load)."""
        lo = t0 + 0.2 * (t1 - t0)
        inside = [s_ for s_ in self.samples if lo <= s_[0] <= t1]
        if not inside:
            return None
        ws = [s_[1] for s_ in inside]
        fs = [s_[2] for s_ in inside if s_[2]]
        return {"avg_w": round(sum(ws) / len(ws), 1), "max_w": round(max(ws), 1), "cap_w": self.cap_w,
                "sclk_mhz": round(sum(fs) / len(fs), 0) if fs else None, "samples": len(inside), "source": self.source,
                "during": "the sustained leg (back-to-back steps behind the timed region)"}


def launch_ranks(n):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: this process starts the N ranks itself - one CHILD
    process per GPU with the environment `torch.distributed.run` would give it (RANK, LOCAL_RANK, WORLD_SIZE), the same
    argv - waits for them, hands rank 0's JSON line to its own stdout and exits with the first non-zero exit code.  The
    parent never imports torch and never touches HIP (children are started with `subprocess`, nothing is exec'ed over a
    process that has initialised the GPU).  A rank that dies takes the others with it (they would wait in a collective for
    ever): the survivors get SIGTERM, then SIGKILL, by pid; a parent that is killed outright takes them along too (every
    child asks for SIGTERM on its parent's death, `rank_dies_with_parent`).
    Rendezvous: a FILE store in a private temporary directory (TS_BENCH_INIT_FILE -> init_method file://...), not a TCP port
    picked here: a port found free by bind-and-close can be taken by another process before rank 0 listens on it.
    SURVEY.md 8e: the reference's only multi-device call is SentenceTransformer.encode_multi_process
    (ec2/generate_embeddings/embeddings.py:32) - a worker pool started by the library from one `python -m` command."""
    import shutil
    import signal
    import subprocess
    import tempfile
    import threading
    rdv_dir = tempfile.mkdtemp(prefix="ts_bench_rdv_")
    ncpu = len(os.sched_getaffinity(0))
    procs, lines = [], []
    lock = threading.Lock()

    def relay(p):
        for raw in p.stdout:                       # a rank's stdout: only rank 0 writes there, exactly one JSON line
            ln = raw.decode("utf-8", "replace")
            if ln.lstrip().startswith("{"):
                with lock:
                    lines.append(ln if ln.endswith("\n") else ln + "\n")
            elif ln.strip():
                sys.stderr.write(ln)

    for r in range(n):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n),
                    "TS_BENCH_INIT_FILE": os.path.join(rdv_dir, "store"), "TS_BENCH_LAUNCHED_BY": "bench.py"})
        for k_ in ("MASTER_ADDR", "MASTER_PORT"):
            env.pop(k_, None)
        # ROCr's IPC mode for memory shared between processes: 0 = dmabuf handles, the only kind the host driver of this
        # image's GPU pool supports.  RCCL maps its peers' buffers through it when the ranks are separate processes; with the
        # legacy mode its set-up fails with "hipIpcGetMemHandle: invalid argument".  The image exports the variable already;
        # this keeps it for ranks started from an environment that was scrubbed (a driver's `env -i`, a test's own dict).
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # host threads per rank: corpus generation, the fp64 truth and the gloo rehearsal share the node's cores
        env.setdefault("OMP_NUM_THREADS", str(max(1, ncpu // n)))
        p = subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, stdout=subprocess.PIPE)
        t = threading.Thread(target=relay, args=(p,), daemon=True)
        t.start()
        procs.append((p, t))

    def stop_all(sig):
        for p, _ in procs:
            if p.poll() is None:
                try:
                    p.send_signal(sig)
                except OSError:
                    pass

    signal.signal(signal.SIGTERM, lambda *_: (stop_all(signal.SIGTERM), sys.exit(143)))
    rc = 0
    try:
        while any(p.poll() is None for p, _ in procs):
            for p, _ in procs:
                if p.poll() not in (None, 0) and rc == 0:
                    rc = p.returncode
                    print(f"[bench] rank process {p.pid} exited with {rc}: stopping the other ranks", file=sys.stderr, flush=True)
                    stop_all(signal.SIGTERM)
                    t_end = time.time() + 20
                    while time.time() < t_end and any(q.poll() is None for q, _ in procs):
                        time.sleep(0.2)
                    stop_all(signal.SIGKILL)
            time.sleep(0.1)
    except KeyboardInterrupt:
        stop_all(signal.SIGTERM)
        rc = 130
    for p, t in procs:
        p.wait()
        t.join(timeout=5)
        if rc == 0 and p.returncode != 0:
            rc = p.returncode
    if rc == 0 and len(lines) != 1:
        print(f"[bench] expected one JSON line from rank 0, got {len(lines)}", file=sys.stderr, flush=True)
        rc = 1
    shutil.rmtree(rdv_dir, ignore_errors=True)
    if rc == 0:
        sys.stdout.write(lines[0])
        sys.stdout.flush()
    sys.exit(rc)


def decide_placement(overlap_ms, inline_ms, discard=1, margin=0.02):
    """The exchange-placement rule of the N > 1 trial: per mode the rounds' ms per step in the order they ran; the first
    `discard` rounds of each mode do not count (lazy connection set-up, clock ramp), the MEDIANS of the rest are compared, and
    `overlap` - the placement the design argues for - is kept unless `inline` wins by more than `margin`.
    Returns (use_overlap, median_overlap, median_inline)."""
    med_o = float(np.median(overlap_ms[discard:] if len(overlap_ms) > discard else overlap_ms))
    med_i = float(np.median(inline_ms[discard:] if len(inline_ms) > discard else inline_ms))
    return (not (med_i < (1.0 - margin) * med_o)), med_o, med_i


def rank_dies_with_parent():
    """A rank started by `launch_ranks`: ask the kernel for SIGTERM when the parent goes away (prctl PR_SET_PDEATHSIG), so a
    parent killed with SIGKILL (`timeout -k`) does not leave ranks holding their GPUs.  Called before torch is imported."""
    import ctypes
    import signal
    try:
        ctypes.CDLL("libc.so.6", use_errno=True).prctl(1, int(signal.SIGTERM), 0, 0, 0)      # 1 = PR_SET_PDEATHSIG
    except OSError:
        return
    if os.getppid() == 1:                                      # the parent went away before the request was in place
        raise SystemExit("rank: the launching process is gone")


def init_process_group(dist, backend, rank, world, **kw):
    """Under `launch_ranks` the ranks meet through a file store (TS_BENCH_INIT_FILE); under torch.distributed.run through
    the launcher's MASTER_ADDR / MASTER_PORT."""
    path = os.environ.get("TS_BENCH_INIT_FILE")
    if path:
        dist.init_process_group(backend=backend, init_method=f"file://{path}", rank=rank, world_size=world, **kw)
    else:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)


def rehearse_launch(args, real_stdout):
    """`--rehearse-launch`: the launch and host-collective plumbing of an N-rank run WITHOUT a device - what `--gpus 8` does
    around its searches on a node nobody here has seen: the rendezvous, barriers, the max-over-ranks all-reduce of the step
    time, the placement trial's all-reduces and `all_gather_object` of truth tables of the real shape
    (nq x (k + 64) fp64 + int64 per rank), over gloo.  No search runs and no number is reported as one: rank 0 prints one
    JSON line that says what was exercised.  (A one-GPU box may hold at most six processes on its card, so the eight-rank
    case cannot be rehearsed through the kernels; `--share-gpu` covers up to five ranks that way.)"""
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    init_process_group(dist, "gloo", rank, world)
    nq = args.nq or WORKLOADS[args.workload][2]
    t0 = time.perf_counter()
    dist.barrier()
    t = torch.tensor([float(rank)], dtype=torch.float64)
    for _ in range(16):                                       # the timed region's and the placement trial's reductions
        t.fill_(float(rank))
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        assert int(t.item()) == world - 1
    table = (np.full((nq, K + 64), float(rank)), np.full((nq, K + 64), rank, dtype=np.int64), np.full((nq, K), float(rank)), rank)
    objs = [None] * world
    dist.all_gather_object(objs, table)
    assert [o[3] for o in objs] == list(range(world)) and all(o[0].shape == (nq, K + 64) for o in objs)
    blk = torch.full(((nq * K * 4 + 7) // 8 * 8 + nq * K * 8,), rank, dtype=torch.uint8)
    allb = torch.empty(world * blk.numel(), dtype=torch.uint8)
    dist.all_gather_into_tensor(allb, blk)                    # the packed per-shard top-k block of one step
    assert allb.view(world, -1)[:, 0].tolist() == list(range(world))
    dist.barrier()
    if rank == 0:
        line = {"rehearsal": "launch + host collectives only, no device, no search", "n_ranks": world, "backend": "gloo",
                "launched_by": os.environ.get("TS_BENCH_LAUNCHED_BY", "env"), "rendezvous": "file" if os.environ.get("TS_BENCH_INIT_FILE") else "env",
                "omp_num_threads": os.environ.get("OMP_NUM_THREADS"), "hsa_enable_ipc_mode_legacy": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY"),
                "truth_table_bytes_per_rank": int(sum(a.nbytes for a in table[:3])), "packed_block_bytes": int(blk.numel()),
                "seconds": round(time.perf_counter() - t0, 3)}
        os.write(real_stdout, (json.dumps(line) + "\n").encode())
    dist.destroy_process_group()


def run_c1(args, real_stdout):
    """BASELINE.json configs[0]: `evaluate_retrieval` (compare_embeddings.py:55-92) on ~1k theorems and 73 queries - the
    [Q x N] cosine matrix and the six metrics, end to end.  The reference runs this on the host (numpy / torch-CPU); this
    product has no CPU compute path, so the SAME call runs through libtsearch on the GPU (util.cos_sim -> ts_scores) and the
    reference's host formulation is timed beside it as `cpu_baseline`.  One step = one evaluate_retrieval call; embeddings
    are precomputed (a model stub hands them out: the encoder forward is configs[4]'s subject, not this one's)."""
    import contextlib
    import io

    import torch  # noqa: F401  (the package's encoder module imports it)
    import synthetic
    from oracle import oracle
    from theoremsearch_amd import _ffi
    from theoremsearch_amd import compare_embeddings as ce
    if _ffi.device_count() <= 0:
        raise SystemExit("bench.py needs a HIP device (libtsearch has no CPU path)")
    n, _, nq = WORKLOADS["c1"]
    n, nq = args.rows or n, args.nq or nq
    rng = np.random.default_rng(7)
    s_emb = synthetic.synth_chunk(0, n, D)
    gold = rng.choice(n, nq, replace=False)
    q_emb = (s_emb[gold] + rng.standard_normal((nq, D)).astype(np.float32) * np.float32(0.9 / np.sqrt(D))).astype(np.float32)
    theorems = [(f"theorem {i}", f"paper {i // 4}") for i in range(n)]
    queries = [(f"query {j}", f"paper {int(gold[j]) // 4}") for j in range(nq)]
    # graded relevance as _generate_qrels makes it (compare_embeddings.py:175-183): the exact theorem 1.0, same paper 0.5
    qrels = {j: {int(i): (1.0 if i == gold[j] else 0.5) for i in range(int(gold[j]) // 4 * 4, min(n, int(gold[j]) // 4 * 4 + 4))}
             for j in range(nq)}

    class Stub:
        def encode(self, texts, **_):
            return s_emb if texts and texts[0].startswith("theorem") else q_emb

    def step():
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            ce.evaluate_retrieval(Stub(), theorems, queries, qrels, top_k_report=3)
        return buf.getvalue()

    for _ in range(args.warmup):
        report = step()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        report = step()
    dt = time.perf_counter() - t0
    # the checker: the oracle's restatement of the reference's own functions on the same embeddings
    sim = oracle.cos_sim(q_emb, s_emb)
    want = {"P@1": oracle.precision_at_k(sim, qrels, k=1), "H@3": oracle.hit_at_k(sim, qrels, k=3), "MRR@3": oracle.mrr_at_k(sim, qrels, k=3),
            "nDCG@3": oracle.ndcg_at_k(sim, qrels, k=3), "ERR@3": oracle.err_at_k(sim, qrels, k=3),
            "Q-measure@3": oracle.q_measure_at_k(sim, qrels, k=3)}
    got = {ln.split(" | ")[0]: float(ln.split(" | ")[1]) for ln in report.splitlines() if " | " in ln}
    viol = [k_ for k_, v in want.items() if abs(got.get(k_, float("nan")) - v) > 1e-9]
    t1 = time.perf_counter()
    reps = 0
    while time.perf_counter() - t1 < 3.0 or reps < 3:
        sim = oracle.cos_sim(q_emb, s_emb)
        for fn, kk in ((oracle.precision_at_k, 1), (oracle.hit_at_k, 3), (oracle.mrr_at_k, 3), (oracle.ndcg_at_k, 3),
                       (oracle.err_at_k, 3), (oracle.q_measure_at_k, 3)):
            fn(sim, qrels, k=kk)
        reps += 1
    t_cpu = (time.perf_counter() - t1) / reps
    line = {
        "metric": "queries/sec, evaluate_retrieval (cosine matrix + six retrieval metrics) over N x 768 theorem embeddings",
        "value": round(nq * args.steps / dt, 1), "unit": "queries/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"compare_embeddings.evaluate_retrieval on {n} theorems x {nq} queries, d = {D}, six metrics at k = 3 "
                               "(BASELINE.json configs[0]); runs through libtsearch on the GPU: the product has no CPU compute path",
                   "rows": n, "dim": D, "batch": nq, "k": 3},
        "parity": {"metrics_checked": len(want), "violations": len(viol), "tolerance": 1e-9, "metrics": got},
        "roofline": None,
        "cpu_baseline": {"value": round(nq / t_cpu, 1), "unit": "queries/s", "cores": 1, "kind": "port",
                         "sample": f"the oracle's restatement of the reference formulation (numpy cos_sim + six full argsorts, "
                                   f"compare_embeddings.py:61-92) on the same embeddings, {reps} repetitions, {t_cpu * 1e3:.2f} ms each, one core"},
    }
    os.write(real_stdout, (json.dumps(line) + "\n").encode())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--rows", type=int, default=0, help="override the corpus size (debug)")
    ap.add_argument("--nq", type=int, default=0, help="override the query batch (debug)")
    ap.add_argument("--algo", default="auto")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-recall", action="store_true")
    ap.add_argument("--gen-threads", type=int, default=0)
    ap.add_argument("--dim", type=int, default=0, help="embedding dimension (default 768, the BASELINE configs'; c3q: 1024)")
    ap.add_argument("--seq-len", type=int, default=32, help="c5: tokens per synthetic query")
    ap.add_argument("--encoder", default="bert", choices=["bert", "qwen", "gemma"],
                    help="c5: random-init stand-in of math-similarity/Bert-MLM_arXiv-MP-class_zbMath (BERT-base shape, 768-d: the "
                         "BASELINE config) or of Qwen/Qwen3-Embedding-0.6B (the production embedder, streamlit_app.py:55: 28 layers, "
                         "1024-d, grouped-query attention, last-token pooling; the index is then 10M x 1024) or of google/embeddinggemma-300m "
                         "(the reference's second embedder, ec2/generate_embeddings/embedders.py:1-4: Gemma3 text model, 24 layers, "
                         "768-d, mean pooling + two Dense modules)")
    ap.add_argument("--force-dist", action="store_true", help="debug: run the exchange + merge path even with one rank")
    ap.add_argument("--exchange", default="auto", choices=["auto", "native", "torch"],
                    help="N > 1: collective of the per-shard top-k: native = ncclAllGather inside libtsearch (ts_comm_*), "
                         "torch = torch.distributed.all_gather_into_tensor; auto = native on the nccl backend")
    ap.add_argument("--exchange-placement", default="auto", choices=["auto", "overlap", "inline"],
                    help="N > 1: where the all-gather + merge of a step run - 'overlap': on a side stream beside the next step's "
                         "search; 'inline': on the search's own stream; 'auto': both are tried for a few steps before the warm-up "
                         "and the faster one is used (every rank sees the same timings: max over ranks)")
    ap.add_argument("--cpu-baseline-full", action="store_true",
                    help="time the CPU baseline for EVERY query of the batch over the whole corpus, 16 queries at a time (minutes; "
                         "default: the whole corpus, as many 16-query chunks as fit in --cpu-baseline-seconds)")
    ap.add_argument("--cpu-baseline-seconds", type=float, default=20.0, help="host time the default CPU baseline may take")
    ap.add_argument("--sustained-steps", type=int, default=300,
                    help="steps of the sustained leg behind the timed region (0 = skip); reported as `sustained`")
    ap.add_argument("--no-ceiling", action="store_true", help="skip the measured-ceiling legs (tools/microbench)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="rehearsal of the N > 1 path on a one-GPU box: every rank uses device 0, exchange over gloo "
                         "through host memory (timings meaningless)")
    ap.add_argument("--mask-frac", type=float, default=0.0,
                    help="diagnostic: filtered search with this fraction of rows allowed (device bitmask)")
    ap.add_argument("--pipeline", type=int, default=1,
                    help="searches in flight: the timed loop alternates this many handles on the same rows (the index and "
                         "views of it, ts_index_view), each on its own stream, so that the small kernels at the head of "
                         "step i + 1 overlap the tail of step i (independent batches back to back, streamlit_app.py:165-173); "
                         "1 = one handle, one stream (default: measured on one box, two in flight gain 0.0 % on the 10M corpus and "
                         "2 % on an eighth of it - the full pass holds every CU, so the small kernels of the other search "
                         "only move, they do not hide - and the event brackets of overlapping passes stop being kernel time)")
    ap.add_argument("--encoder-graph", action="store_true",
                    help="c5: capture the encoder forward in a HIP graph per stream and replay it every step (default: eager "
                         "launches; measured on one box: 6.24 ms per step replayed, 6.19 eager - the forward is bound by its "
                         "kernels' device time, not by launch gaps)")
    ap.add_argument("--tune-gemms", action="store_true",
                    help="c5: let PyTorch's TunableOp pick the fastest hipBLASLt / rocBLAS solution for each of the encoder's "
                         "four GEMM shapes during the warm-up (seconds of tuning per shape)")
    ap.add_argument("--encoder-dtype", default="fp32", choices=["fp32", "fp32x3", "bf16"],
                    help="c5: the arithmetic of the encoder forward.  fp32 (default) is what the reference runs "
                         "(SentenceTransformer(name) without a dtype: streamlit_app.py:55,173, app_create_embeddings.py:22,81) and what "
                         "SentenceEncoder does for a real checkpoint; fp32x3 keeps fp32 weights and activations and runs the GEMMs on "
                         "the bf16 matrix pipe from bf16 pieces (SentenceEncoder(fp32_gemm='bf16x3'): 16 significant bits per factor, "
                         "fp32 accumulation); bf16 is the opt-in half-precision forward (DESIGN.md section 8 states how far the "
                         "embeddings of either lie from the fp32 ones)")
    ap.add_argument("--rehearse-launch", action="store_true",
                    help="N > 1: exercise only the launch + host-collective plumbing of an N-rank run over gloo, no device, no "
                         "search (rehearse_launch); prints what was exercised, not a benchmark line")
    ap.add_argument("--zero-queries", action="store_true", help="diagnostic: all-zero queries (power probe)")
    ap.add_argument("--zero-corpus", action="store_true", help="diagnostic: all-zero corpus (power probe)")
    args = ap.parse_args()
    global D
    D = args.dim or (1024 if (args.workload == "c3q" or (args.workload == "c5" and args.encoder == "qwen")) else 768)
    if args.workload != "c1" and args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher around this process: start the ranks as child processes (before torch is imported or HIP touched)
        return launch_ranks(args.gpus)
    if os.environ.get("TS_BENCH_LAUNCHED_BY") == "bench.py":
        rank_dies_with_parent()
    # Exactly ONE line goes to stdout (the JSON): libraries print banners there (RCCL prints its version / host /
    # library path on communicator creation), so fd 1 points at stderr until the result line is written.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    if args.workload == "c1":
        return run_c1(args, real_stdout)
    if args.rehearse_launch:
        return rehearse_launch(args, real_stdout)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 inside a launcher that set WORLD_SIZE=1: start `python bench.py --gpus N` bare (it starts "
                             "its own ranks) or under torch.distributed.run --nproc-per-node N")
        args.gpus = world

    import torch
    import torch.distributed as dist
    import synthetic                # input generator (plain numpy)
    from oracle import oracle       # checker legs only: recall@10 and cpu_baseline
    import theoremsearch_amd as ts
    from theoremsearch_amd import _ffi

    if _ffi.device_count() <= 0:
        raise SystemExit("bench.py needs a HIP device (libtsearch has no CPU path)")
    if args.share_gpu:
        local_rank = 0
    if local_rank >= _ffi.device_count():
        raise SystemExit(f"rank {rank}: local rank {local_rank} has no GPU ({_ffi.device_count()} visible); "
                         f"--share-gpu rehearses several ranks on one GPU")
    torch.cuda.set_device(local_rank)
    if world > 1 or args.force_dist:
        if args.share_gpu:
            init_process_group(dist, "gloo", rank, world)
        else:
            init_process_group(dist, "nccl", rank, world, device_id=torch.device("cuda", local_rank))

    rows_total, dtype, nq = WORKLOADS[args.workload]
    if args.rows:
        rows_total = args.rows
    if args.nq:
        nq = args.nq
    bf16 = dtype == "bf16"
    elem = 2 if bf16 else 4

    # ---- this rank's shard: contiguous rows [lo, hi), generated chunk by chunk -------------------
    lo = rows_total * rank // world
    hi = rows_total * (rank + 1) // world
    n_local = hi - lo
    CH = synthetic.CHUNK_ROWS
    chunks = list(range(lo // CH, (hi + CH - 1) // CH))
    ix = ts.TheoremIndex(n_local, D, dtype=dtype, metric="ip", device=local_rank, row_offset=lo)
    ncpu = len(os.sched_getaffinity(0))
    nthreads = args.gen_threads or max(1, min(16, ncpu // max(1, min(world, 8))))
    t_gen = time.time()
    cache = {}                                   # host copies for the recall check / cpu baseline
    cache_ok = n_local * D * elem <= (40 << 30) and not args.no_recall

    def make(c):
        data = synthetic.synth_chunk(c, CH, D, bf16=bf16) if not args.zero_corpus else np.zeros((CH, D), np.uint16 if bf16 else np.float32)
        a, b = max(lo, c * CH), min(hi, (c + 1) * CH)
        ix.upload(data[a - c * CH:b - c * CH], a - lo)
        if cache_ok or (c == 0 and rank == 0):
            cache[c] = data
        return c

    with ThreadPoolExecutor(nthreads) as ex:
        for i, c in enumerate(ex.map(make, chunks)):
            if (i + 1) % 8 == 0:
                log(rank, f"generated+uploaded {i + 1}/{len(chunks)} chunks ({time.time() - t_gen:.0f}s)")
    log(rank, f"corpus ready: {n_local} rows/rank x {D} {dtype} in {time.time() - t_gen:.1f}s ({nthreads} threads)")

    q_host = synthetic.synth_queries(0, nq, D, bf16=bf16)      # uint16 bits or float32
    if args.zero_queries:
        q_host = np.zeros_like(q_host)
    # The timed loop runs on an explicit stream (never on the default stream: the library's stream argument 0 means "the
    # index's own stream").  With N > 1 the step is ShardedSearcher.search_device: local search on `main`, ONE all-gather
    # of the packed per-shard top-k + the merge on a side stream, overlapping the next step's search.
    P = max(1, args.pipeline)
    main = torch.cuda.Stream()
    q_dev = torch.from_numpy(q_host.view(np.int16) if bf16 else q_host).cuda()
    idx_off = (nq * K * 4 + 7) // 8 * 8
    blk = idx_off + nq * K * 8
    nres = max(2, P)
    res = [torch.empty(blk, dtype=torch.uint8, device="cuda") for _ in range(nres)]   # N = 1: results of step i in res[i % nres]
    use_dist = world > 1 or args.force_dist
    searcher = None
    handles, lanes = [ix], [main]
    if use_dist:
        from theoremsearch_amd.distributed import ShardedSearcher
        searcher = ShardedSearcher(index=ix, exchange=("torch" if args.share_gpu else args.exchange), pipeline=P)
        log(rank, f"sharded search: exchange = {searcher.exchange}" + (" (RCCL inside libtsearch)" if searcher.exchange == "native" else f" (torch.distributed, backend {searcher.backend})"))
    else:
        for _ in range(P - 1):                     # N = 1: the loop itself alternates the handles
            handles.append(ix.view())
            lanes.append(torch.cuda.Stream())
    encoder = None
    if args.workload == "c5":
        # BASELINE.json configs[4]: encoder forward (PyTorch-ROCm, random-init BERT-base-shaped stand-in: no weights
        # offline) on synthetic token sequences, pooled + normalised on the device, handed to the search by pointer
        from theoremsearch_amd.encoder import SentenceEncoder
        if args.tune_gemms:
            import torch.cuda.tunable as tunable
            tunable.enable(True)
            tunable.tuning_enable(True)
            tunable.set_max_tuning_duration(30)
            tunable.set_max_tuning_iterations(20)
            tunable.set_filename(os.path.join(os.environ.get("TMPDIR", "/tmp"), f"ts_tunableop_rank{rank}.csv"))
        enc_name = {"qwen": "Qwen/Qwen3-Embedding-0.6B", "gemma": "google/embeddinggemma-300m"}.get(
            args.encoder, "math-similarity/Bert-MLM_arXiv-MP-class_zbMath")
        enc_dtype = torch.bfloat16 if args.encoder_dtype == "bf16" else torch.float32
        encoder = SentenceEncoder(enc_name, allow_random_init=True, dtype=enc_dtype,
                                  fp32_gemm="bf16x3" if args.encoder_dtype == "fp32x3" else "blas")
        if encoder.embedding_dim != D:
            raise SystemExit(f"the {args.encoder} encoder embeds into {encoder.embedding_dim} dimensions, the index has {D}")
        log(rank, f"encoder: {type(encoder.model).__name__} ({enc_name}, random init, {args.encoder_dtype} forward), fused forward: {type(encoder._fused).__name__}")
        g = torch.Generator(device="cpu").manual_seed(5678)
        tok_ids = torch.randint(1000, 30000, (nq, args.seq_len), generator=g).cuda()
        tok_ids[:, 0], tok_ids[:, -1] = 101, 102
        tok_mask = torch.ones_like(tok_ids)
    step_no = [0]
    last_out = [None]
    mask_ptr, mask_host = 0, None
    if args.mask_frac > 0:
        mask_host = np.random.default_rng(99 + rank).random(n_local) < args.mask_frac
        bits = np.packbits(mask_host, bitorder="little")
        words = np.zeros((n_local + 31) // 32 * 4, dtype=np.uint8)
        words[: bits.shape[0]] = bits
        mask_dev = torch.from_numpy(words).cuda()
        mask_ptr = mask_dev.data_ptr()
    torch.cuda.synchronize()

    def encode_queries():
        with torch.inference_mode():
            # fused BERT forward (QKV as one GEMM, add + LayerNorm as one kernel); the synthetic batch has no padding
            hidden = encoder.forward_hidden(tok_ids, tok_mask, no_padding=True)
            # fused mean-pool + L2 normalise + round to bf16 (ts_pool_normalize): the form the bf16 index multiplies, read in
            # place by the search - no fp32 round trip, no preparation launch
            return encoder.pool(hidden, tok_mask, True, out_dtype=torch.bfloat16 if bf16 else torch.float32)

    # The encoder forward is ~150 small launches (12 layers of GEMM + layer norm + GELU + add): captured ONCE per stream in a
    # HIP graph (fixed shapes: the token batch is resident) and replayed every step, so the launches stop paying host gaps.
    enc_graphs = {}

    def make_encoder_graph(lane):
        try:
            with torch.cuda.stream(lane):
                for _ in range(3):
                    encode_queries()
            lane.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=lane):
                out = encode_queries()
            return g, out
        except Exception as e:                     # noqa: BLE001 - an op that cannot be captured: eager forward instead
            log(rank, f"encoder graph capture failed ({type(e).__name__}: {e}); eager forward")
            torch.cuda.synchronize()
            return None

    def step():
        i = step_no[0]
        step_no[0] += 1
        b = i % nres
        lane = lanes[i % len(lanes)]
        with torch.cuda.stream(lane):
            if encoder is not None:
                g_ = enc_graphs.get(id(lane))
                if g_ is not None:
                    g_[0].replay()                            # on this step's stream: the search below is ordered behind it
                    emb = g_[1]
                else:
                    emb = encode_queries()
                    emb.record_stream(lane)
                qp, qd = emb.data_ptr(), ("bf16" if emb.dtype == torch.bfloat16 else "f32")
            else:
                qp, qd = q_dev.data_ptr(), dtype
            if searcher is not None:
                last_out[0] = searcher.search_device(qp, qd, nq, K, stream=lane, algo=args.algo, mask_ptr=mask_ptr, overlap=overlap[0])
            else:
                base = res[b].data_ptr()
                handles[i % len(handles)].search_device(qp, qd, nq, K, base, base + idx_off, lane.cuda_stream, algo=args.algo,
                                                        mask_ptr=mask_ptr)
                last_out[0] = b

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # Where the exchange of a step runs.  On a side stream it overlaps the next step's search - but the full pass of that
    # search wants every CU (one workgroup per CU, all registers), and a collective's kernel that sits on a CU waiting for a
    # slower rank holds the pass's last workgroup back; on the search's own stream it costs its latency every step and
    # nothing else.  Which is faster depends on the fabric and on how evenly the ranks run, so with more than one rank both
    # are tried.  The trial must not decide on noise (RCCL sets its connections up lazily during the first exchanges, and the
    # clock needs a second under load): a warm-up of >= 50 exchanges in each mode, then `rounds` rounds of `trial` steps per
    # mode in alternating order, the first round of each mode discarded, MEDIANS compared, and `overlap` - the placement the
    # design argues for - is kept unless `inline` wins by more than 2 %.  The time of a round is the slowest rank's.
    # One rank has no collective: overlap.
    overlap = [args.exchange_placement != "inline"]
    placement_trial = None
    if searcher is not None and world > 1 and args.exchange_placement == "auto":
        trial, rounds, warm = (10, 3, 5) if args.share_gpu else (200, 4, 60)      # gloo through host memory: a token trial
        times = {True: [], False: []}
        for mode in (True, False):
            overlap[0] = mode
            for _ in range(warm):
                step()
            torch.cuda.synchronize()
        barrier()
        for rnd in range(rounds):
            for mode in ((True, False) if rnd % 2 == 0 else (False, True)):
                overlap[0] = mode
                barrier()
                tt = time.perf_counter()
                for _ in range(trial):
                    step()
                torch.cuda.synchronize()
                barrier()
                t_ = torch.tensor([time.perf_counter() - tt], dtype=torch.float64, device="cuda")
                dist.all_reduce(t_, op=dist.ReduceOp.MAX)
                times[mode].append(float(t_.item()) / trial * 1e3)
        overlap[0], med_o_, med_i_ = decide_placement(times[True], times[False])
        med = {True: med_o_, False: med_i_}
        placement_trial = {"overlap_ms_per_step": [round(x, 4) for x in times[True]], "inline_ms_per_step": [round(x, 4) for x in times[False]],
                           "steps_per_round": trial, "rounds": rounds, "rounds_discarded": 1, "warmup_exchanges_per_mode": warm,
                           "median_overlap_ms": round(med[True], 4), "median_inline_ms": round(med[False], 4),
                           "rule": "overlap unless the median of inline is more than 2 % below the median of overlap"}
        log(rank, f"exchange placement: {'overlap' if overlap[0] else 'inline'} ({placement_trial})")

    if encoder is not None and args.encoder_graph:
        for lane_ in lanes:
            enc_graphs[id(lane_)] = make_encoder_graph(lane_)
        log(rank, f"encoder forward: {'HIP graph replay' if all(enc_graphs.values()) else 'eager'}")
    for _ in range(args.warmup):
        step()
    barrier()
    if searcher is not None and P > 1:
        searcher._lane(main)                       # the handles exist before profiling is switched on
    prof_handles = handles if searcher is None else [h for h, _ in (searcher._lanes or [(ix, None)])]
    # The hipEvent brackets around the dominant kernel are two marker packets per step on the search's stream (a few
    # microseconds each: 0.2 % of the 3.1 ms step of the whole corpus, 1.5 % of an eighth of it).  One rank: the timed region
    # carries them (the contract's "live, over the timed region").  Sharded runs: the timed region runs bare and the kernel
    # time comes from the bracketed leg right behind it (the sustained leg).
    bracket_timed = not use_dist
    for h in prof_handles:
        h.profile_enable(bracket_timed)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    prof = {"launches": 0, "total_ms": 0.0, "rows_per_launch": 0}
    for h in prof_handles:
        p_ = h.profile_read()
        h.profile_enable(False)
        prof["launches"] += p_["launches"]
        prof["total_ms"] += p_["total_ms"]
        prof["rows_per_launch"] = max(prof["rows_per_launch"], p_["rows_per_launch"])
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt / args.steps * 1e3
    qps = nq * args.steps / dt

    # ---- sustained: the same step, `--sustained-steps` more of them back to back (the timed region above is the
    # contract's K steps - 63 ms at the default; a second of back-to-back passes lets the clock settle) -----------------
    sustained, power = None, None
    sustained_steps = args.sustained_steps if (args.sustained_steps > 0 or bracket_timed) else max(20, args.steps)
    if args.sustained_steps >= 100 and ms_per_step > 0:
        # ... and at least a second of them: behind the idle gap between the legs the power controller needs ~20 steps to settle
        # (a 1.25M-row shard's pass: 363 us in the first steps, 557 us eight steps later, 420 us from step 20 on), and 300
        # steps of such a shard are 0.15 s - round 4's sustained leg read 13 % above its own timed region for that reason
        sustained_steps = max(sustained_steps, int(1000.0 / ms_per_step) + 1)

    def read_brackets():
        sp_ = {"launches": 0, "total_ms": 0.0, "rows_per_launch": 0}
        for h in prof_handles:
            p_ = h.profile_read()
            h.profile_enable(False)
            sp_["launches"] += p_["launches"]
            sp_["total_ms"] += p_["total_ms"]
            sp_["rows_per_launch"] = max(sp_["rows_per_launch"], p_["rows_per_launch"])
        return sp_

    if sustained_steps > 0 and args.workload != "c1":
        barrier()
        # the sustained leg runs as the timed region does: with the brackets on one rank, bare on a sharded run (whose kernel
        # time comes from a third, bracketed leg right behind; round 4 bracketed the sustained leg itself, and its step read
        # 68 us above the timed region's at 1.25M rows per rank: each bracket is a barrier packet with a timestamp write
        # between two kernels that otherwise dispatch back to back)
        for h in prof_handles:
            h.profile_enable(bracket_timed)
        try:
            props = torch.cuda.get_device_properties(local_rank)
            bus = f"{props.pci_domain_id:04x}:{props.pci_bus_id:02x}:{props.pci_device_id:02x}.0"
        except Exception:                          # noqa: BLE001
            bus = None
        sampler = PowerSampler(bus, local_rank) if rank == 0 else None
        tw0 = time.time()
        t1 = time.perf_counter()
        if sampler is not None:
            with sampler:
                for _ in range(sustained_steps):
                    step()
                torch.cuda.synchronize()
        else:
            for _ in range(sustained_steps):
                step()
            torch.cuda.synchronize()
        tw1 = time.time()
        barrier()
        dt_s = time.perf_counter() - t1
        power = sampler.summary(tw0, tw1) if sampler is not None else None
        sp = read_brackets()
        if world > 1:
            t = torch.tensor([dt_s], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt_s = float(t.item())
        sustained = {"steps": sustained_steps, "queries_per_s": round(nq * sustained_steps / dt_s, 1),
                     "ms_per_step": round(dt_s / sustained_steps * 1e3, 4), "kernel_brackets": bracket_timed,
                     "kernel_ms": round(sp["total_ms"] / max(1, sp["launches"]), 4) if bracket_timed else None,
                     "seconds": round(dt_s, 3)}
        if not bracket_timed:
            nb = min(sustained_steps, 100)
            for h in prof_handles:
                h.profile_enable(True)
            t2 = time.perf_counter()
            for _ in range(nb):
                step()
            torch.cuda.synchronize()
            dt_b = time.perf_counter() - t2
            barrier()
            prof = dict(read_brackets(), steps=nb)
            sustained["bracket_leg"] = {"steps": nb, "ms_per_step": round(dt_b / nb * 1e3, 4),
                                        "kernel_ms": round(prof["total_ms"] / max(1, prof["launches"]), 4),
                                        "note": "this rank's own clock; the two marker packets per step are what this leg's step has over the sustained leg's"}

    # ---- the exchange as the communicator saw it (N > 1): backend, world, one device per rank, and what ONE all-gather of
    # the packed per-shard top-k costs on the side stream (outside the timed region; every rank takes part) ----------------
    exchange = None
    if searcher is not None:
        exchange = searcher.measure_exchange(nq, K)
        exchange["placement"] = "overlap (side stream, beside the next step's search)" if overlap[0] else "inline (the search's own stream)"
        exchange["placement_trial"] = placement_trial
        exchange["launched_by"] = os.environ.get("TS_BENCH_LAUNCHED_BY", "torch.distributed.run" if "TORCHELASTIC_RUN_ID" in os.environ else "env")
        log(rank, f"exchange: {exchange}")

    # ---- how the search ran (one extra search outside the timed region): algorithm, levels, candidates ----------------
    search_stats, stats_algo = None, None
    if not mask_ptr and encoder is None:
        _, _, st_ = ix.search(q_host, K, algo=args.algo, return_stats=True)
        stats_algo = {1: "scan", 2: "mfma"}.get(st_["algo"])
        if st_["algo"] == 2:
            search_stats = {"levels": st_["levels"], "fallback_queries": st_["fallback_queries"],
                            "candidates_per_query": round(st_["candidates"] / float(nq), 1)}
            log(rank, f"threshold levels {st_['levels']}, candidates per query {search_stats['candidates_per_query']}, "
                      f"queries re-run exactly {st_['fallback_queries']}")

    # ---- roofline of the dominant kernel (live hipEvent brackets inside the library, on the launch stream) ---------
    # One launch of the dominant kernel reads the local corpus once for the whole batch: algorithmic bytes = rows * d * s
    # (SURVEY 8d), flops = 2 * batch * rows * d.  Both ceilings are reported; `bound` names the nearer one.
    kern_ms = prof["total_ms"] / max(1, prof["launches"])
    launches_per_step = prof["launches"] / max(1, prof.get("steps", args.steps))
    alg_bytes = prof["rows_per_launch"] * D * elem
    q_per_launch = nq / max(1.0, launches_per_step)
    alg_flops = 2.0 * q_per_launch * prof["rows_per_launch"] * D
    achieved = alg_bytes / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else 0.0
    tflops = alg_flops / (kern_ms * 1e-3) / 1e12 if kern_ms > 0 else 0.0
    mfma_peak = MFMA_PEAK_TFLOPS["bf16" if bf16 else "f32"]
    hbm_frac, mfma_frac = achieved / HBM_PEAK_GBS, tflops / mfma_peak
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath) and not args.nq and not args.rows and not mask_ptr:
        try:
            tb = json.load(open(tpath)).get(f"{args.workload}_n{world}")
            if tb:
                traffic = {"bytes": tb, "source": "profiles/traffic.json: offline rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of "
                                                  "this command (tools/run_profiles.sh), not an observation of this run"}
        except Exception:
            traffic = None
    kernel_name = {"mfma": "mfma16_topk_kernel", "scan": "scan_kernel"}.get(stats_algo or ("scan" if mask_ptr and nq <= 4 else None), "unknown")
    if stats_algo is None and encoder is not None:
        kernel_name = "mfma16_topk_kernel"
    # ---- the ceiling of this pass on THIS device, measured on the resident corpus right here (tools/microbench/
    # mfma_stream_ceiling.hip: the product's tile loop without epilogue / candidates, its DMA stream alone, its matrix work
    # alone, and the same MFMA count with operands in registers = the random-data matrix rate under this device's power cap)
    ceiling = None
    if (rank == 0 and world == 1 and not args.no_ceiling and bf16 and D == 768 and nq == 256 and encoder is None
            and not mask_ptr and n_local >= 64 * 512):
        try:
            import ctypes as C
            cl = C.CDLL(os.path.join(ROOT, "tools", "microbench", "build", "libts_ceiling.so"))
            cl.ts_ceiling_run.restype = C.c_int
            cl.ts_ceiling_run.argtypes = [C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_double), C.c_void_p]
            ms4 = (C.c_double * 4)()
            torch.cuda.synchronize()
            rows_c = n_local // 32 * 32
            if cl.ts_ceiling_run(local_rank, C.c_void_p(ix.info()["rows_ptr"]), rows_c, C.c_void_p(q_dev.data_ptr()), 3, 5, ms4, None) == 0:
                fl = 2.0 * 256 * rows_c * D
                ceiling = {"stream_plus_mfma_ms": round(ms4[0], 4), "stream_only_ms": round(ms4[1], 4),
                           "mfma_lds_only_ms": round(ms4[2], 4), "bare_mfma_ms": round(ms4[3], 4),
                           "measured_gemm_tflops": round(fl / (ms4[3] * 1e-3) / 1e12, 1),
                           "stream_plus_mfma_hbm_frac": round(rows_c * D * 2 / (ms4[0] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                           "kernel_over_stream_plus_mfma": round(kern_ms / ms4[0], 4) if ms4[0] > 0 else None,
                           "how": "same run, same device, the resident corpus: 3 interleaved rounds x 5 launches per leg of "
                                  "tools/microbench/mfma_stream_ceiling.hip"}
        except OSError as e:
            log(rank, f"ceiling legs skipped: {e}")
    by_mfma = mfma_frac > hbm_frac
    roofline = {"bound": "mfma" if by_mfma else "hbm", "kernel": kernel_name,
                "achieved": round(tflops if by_mfma else achieved, 1), "peak": mfma_peak if by_mfma else HBM_PEAK_GBS,
                "unit": "TFLOP/s" if by_mfma else "GB/s", "frac": round(mfma_frac if by_mfma else hbm_frac, 4),
                "hbm_achieved_gbs": round(achieved, 1), "hbm_peak_gbs": HBM_PEAK_GBS, "hbm_frac": round(hbm_frac, 4),
                "mfma_achieved_tflops": round(tflops, 1), "mfma_peak_tflops": mfma_peak, "mfma_frac": round(mfma_frac, 4),
                "mfma_frac_of_measured_gemm_rate": (round(tflops / ceiling["measured_gemm_tflops"], 4) if ceiling else None),
                "mfma_frac_of_guide_gemm_rate": round(tflops / MFMA_RANDOM_DATA_GEMM_TFLOPS, 4) if bf16 else None,
                "ceiling": ceiling,
                "power": power,
                "traffic": traffic,
                "kernel_ms": round(kern_ms, 4), "launches_per_step": launches_per_step,
                "kernel_ms_from": "hipEvent brackets over the timed region" if bracket_timed else
                                  "hipEvent brackets over a separate leg of <= 100 steps behind the timed region and the sustained leg (which both run without them)",
                "algorithmic_bytes_per_launch": alg_bytes, "algorithmic_flops_per_launch": alg_flops}

    # ---- recall@10 of EVERY query against the oracle (fp64 scores of the same bf16/fp32 values; oracle.ChunkedTruth
    # applies check_topk_against_truth's protocol: pinned ranks exact, near-tie runs as sets, scores within 1e-5) --------
    if searcher is not None:
        fs, fi, done = last_out[0]
        done.synchronize()
        res_s, res_i = fs.cpu().numpy(), fi.cpu().numpy()
    else:
        raw = res[last_out[0]].cpu().numpy()
        res_s = raw[: nq * K * 4].view(np.float32).reshape(nq, K)
        res_i = raw[idx_off: idx_off + nq * K * 8].view(np.int64).reshape(nq, K)
    recall, parity = None, None
    if not args.no_recall and encoder is not None:
        # what the index multiplied in the step that is checked: the static output of that step's graph, or a fresh forward
        g_ = enc_graphs.get(id(lanes[(step_no[0] - 1) % len(lanes)]))
        e_ = g_[1] if g_ is not None else encode_queries()
        if bf16:       # what a bf16 index multiplied: the encoder's bf16 output as it lies, or its fp32 output rounded as the library rounds it
            q_host = e_.view(torch.int16).cpu().numpy().view(np.uint16) if e_.dtype == torch.bfloat16 else oracle.f32_to_bf16_bits(e_.float().cpu().numpy())
        else:
            q_host = e_.float().cpu().numpy()
    if not args.no_recall:
        qf = oracle.bf16_bits_to_f32(q_host) if bf16 else q_host
        truth = oracle.ChunkedTruth(qf, res_i, K)
        t_chk = time.time()

        def local_scores(c):
            data = cache[c] if c in cache else synthetic.synth_chunk(c, CH, D, bf16=bf16)
            a, b = max(lo, c * CH), min(hi, (c + 1) * CH)
            part = data[a - c * CH:b - c * CH]
            part = oracle.bf16_bits_to_f32(part) if bf16 else part
            return a, b, truth.scores_of_chunk(part)

        with ThreadPoolExecutor(max(1, nthreads // 2)) as ex:
            for a, b, sc in ex.map(local_scores, chunks):
                truth.add_scores(sc, a, None if mask_host is None else mask_host[a - lo:b - lo])
        if world > 1:
            objs = [None] * world
            dist.all_gather_object(objs, (truth.best_s, truth.best_i, truth.got_s, truth.n))
            for r_, o in enumerate(objs):
                if r_ != rank:
                    truth.merge(*o)
        try:
            pstats = truth.check(res_s, gap=1e-6, score_tol=1e-5)
            recall = pstats["recall"]
            parity = {"queries_checked": nq, "pinned_ranks_exact": pstats["pinned"], "positions": pstats["positions"],
                      "max_score_error_allowed": 1e-5, "violations": 0}
        except AssertionError as e:
            recall = 0.0
            parity = {"queries_checked": nq, "violations": 1, "first_violation": str(e)[:300]}
        log(rank, f"parity of {nq} queries vs fp64 oracle: recall@10 {recall:.4f}, {parity} ({time.time() - t_chk:.0f}s)")

    # ---- CPU baseline: the reference's formulation on the host cores, MEASURED on a bounded sample (no scaling) --------
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        q_s = oracle.bf16_bits_to_f32(q_host) if bf16 else q_host
        mem_avail = 0
        try:
            for ln in open("/proc/meminfo"):
                if ln.startswith("MemAvailable:"):
                    mem_avail = int(ln.split()[1]) * 1024
        except OSError:
            pass
        fp32_corpus = rows_total * D * 4
        nq_timed = nq
        if len(cache) == len(chunks) and (not bf16 or mem_avail >= 3 * fp32_corpus) and rows_total * 16 * 12 < mem_avail:
            # the workload's own N: the [nq x N] fp32 matrix + its int64 argsort do not fit at once for N = 10M, so
            # 16 queries per call; every call re-normalises the corpus, as util.cos_sim does.  By default as many calls as
            # fit in --cpu-baseline-seconds (at least one), --cpu-baseline-full: all of them
            c_s = np.concatenate([oracle.bf16_bits_to_f32(cache[c]) if bf16 else cache[c] for c in sorted(cache)], axis=0)[:rows_total]
            t_cpu, nq_timed = 0.0, 0
            for q0 in range(0, nq, 16):
                _, t_ = oracle.cpu_reference_topk(q_s[q0:q0 + 16], c_s, K, threads=ncpu)
                t_cpu += t_
                nq_timed += len(q_s[q0:q0 + 16])
                if not args.cpu_baseline_full and t_cpu >= args.cpu_baseline_seconds:
                    break
            how = f"{nq_timed} of the batch's {nq} queries, 16 per call, over the whole corpus"
        else:
            nch = [c for c in sorted(cache) if c < 4]                     # up to 1M rows: ~10 s of host work at batch 256
            rows_s = np.concatenate([cache[c] for c in nch], axis=0)[: min(len(nch) * CH, rows_total)] if nch else cache[0]
            c_s = oracle.bf16_bits_to_f32(rows_s) if bf16 else rows_s
            # the oracle's port of the reference formulation: util.cos_sim + np.argsort(-S)[:, :10], all host cores
            _, t_cpu = oracle.cpu_reference_topk(q_s, c_s, K, threads=ncpu)
            how = f"all {nq} queries in one call (host memory did not allow the whole corpus: MemAvailable {mem_avail >> 30} GiB)"
        t_enc, enc_how = 0.0, ""
        if encoder is not None:
            # configs[4]: the reference's host path starts with model.encode (streamlit_app.py:173, app_showcase_model.py:92) -
            # the same stand-in (same seed, same weights) in fp32 on the host cores, forward + pooling + normalisation of the
            # very queries whose search was timed above; one untimed call on two queries first (thread pool, primitive caches)
            from theoremsearch_amd.encoder import SentenceEncoder
            host_enc = SentenceEncoder(enc_name, device="cpu", dtype=torch.float32, allow_random_init=True)
            # the reference asks for every core (torch.set_num_threads(cpu_count), ec2/generate_embeddings/embeddings.py:21); on a
            # 256-thread host that turns a 16-query forward into minutes of thread-pool contention (measured: 27 s for the BERT
            # shape, 168 s for the Qwen3 / Gemma3 shapes), so the baseline's encode runs on at most 32 threads - the faster
            # setting for the baseline - and says so
            enc_threads = max(1, min(ncpu, 32))
            torch.set_num_threads(enc_threads)
            ids_h, mask_h = tok_ids[:nq_timed].cpu(), tok_mask[:nq_timed].cpu()
            with torch.inference_mode():
                host_enc.pool(host_enc.forward_hidden(ids_h[:2], mask_h[:2]), mask_h[:2], True)
                te = time.perf_counter()
                emb_h = host_enc.pool(host_enc.forward_hidden(ids_h, mask_h), mask_h, True)
                t_enc = time.perf_counter() - te
            # the host's fp32 embeddings against the device's (the fused forward at --encoder-dtype): cosine per query
            e_dev = oracle.bf16_bits_to_f32(q_host[:nq_timed]) if bf16 else q_host[:nq_timed]
            cosines = np.sum(emb_h.numpy() * e_dev, axis=1) / np.maximum(np.linalg.norm(e_dev, axis=1), 1e-12)
            torch.set_num_threads(ncpu)
            enc_how = (f"; host model.encode of the same {nq_timed} x {args.seq_len}-token queries (same random-init weights, fp32, "
                       f"{enc_threads} threads): {t_enc:.2f}s, included; cosine of the device's embeddings ({args.encoder_dtype} forward) "
                       f"with the host's fp32 ones: min {float(cosines.min()):.6f}")
            del host_enc
            t_cpu += t_enc
        cpu = {"value": round(nq_timed / t_cpu, 3), "unit": "queries/s", "cores": ncpu, "kind": "port",
               "rows": int(c_s.shape[0]), "queries": int(nq_timed),
               "sample": f"MEASURED on {c_s.shape[0]} rows (of the workload's {rows_total}), {how}: fp32 torch-CPU "
                         f"cos_sim (F.normalize both sides + mm) + np.argsort(-S)[:, :10] in {t_cpu:.2f}s on {ncpu} host cores; "
                         f"nothing scaled{enc_how}"}
        if encoder is not None:
            cpu["encode_seconds"] = round(t_enc, 3)
            cpu["encode_threads"] = enc_threads
        del c_s

    if rank == 0:
        line = {
            "metric": f"queries/sec, brute-force top-10 over N x {D} theorem embeddings",
            "value": round(qps, 1), "unit": "queries/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": dtype, "data": "synthetic",
            "config": {"workload": f"{rows_total}x{D} {dtype} corpus, batch-{nq} queries, top-{K} "
                                   + (f"(BASELINE.json configs[{ {'c1': 0, 'c2': 1, 'c2b': 1, 'c3': 2, 'c4': 3, 'c5': 4}[args.workload] }])" if args.workload != "c3q" else
                                      "(the reference's production table shape, theorem_embedding_qwen vector(1024): streamlit_app.py:49, rds_schema.sql:50-56; not a BASELINE.json config)") + (f", {args.encoder_dtype} encoder forward in the loop ({args.seq_len} tokens/query, random-init " + ("Qwen3-Embedding-0.6B shape: the production embedder, streamlit_app.py:55)" if args.encoder == "qwen" else "embeddinggemma-300m shape: the reference's second embedder, embedders.py:1-4)" if args.encoder == "gemma" else "BERT-base shape)") if encoder is not None else ""),
                       "rows": rows_total, "dim": D, "batch": nq, "k": K, "searches_in_flight": P,
                       **({"encoder": args.encoder, "encoder_dtype": args.encoder_dtype, "seq_len": args.seq_len,
                           "encoder_forward": type(encoder._fused).__name__ if encoder._fused is not None else "the model's own"}
                          if encoder is not None else {}),
                       **({"mask_frac": args.mask_frac} if args.mask_frac > 0 else {}),
                       "parallelism": f"corpus row-sharded x{world}" + ((", gloo rehearsal on one GPU" if args.share_gpu else (", ncclAllGather of per-shard top-k inside libtsearch (ts_comm)" if searcher.exchange == "native" else ", torch.distributed all-gather of per-shard top-k (RCCL)")) if use_dist else "")},
            "recall_at_10": recall,
            "parity": parity,
            "search_stats": search_stats,
            "exchange": exchange,
            "sustained": sustained,
            "roofline": roofline,
            "cpu_baseline": cpu,
        }
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(line) + "\n").encode())
    if searcher is not None:
        searcher.close()
    for h in handles[1:]:
        h.close()                                  # views go before the index they view
    if world > 1 or args.force_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
