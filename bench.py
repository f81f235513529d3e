"""bench.py — Quick-ADC flat scan on MI355X: PQ codes scanned / second (+ Recall@100).

Workload (BASELINE.json configs[3] shape, which fits one GPU): flat database of 1B synthetic
16x4 PQ codes (8 B/code, counter-based generator), R = 100, keep = 1 %, every query scans the
whole list.  A step = one batch of NQ = 32 queries (the reference's documented `-b32`,
README.md:275-330) through the whole scanner_4::query_scan path (float pre-scan of the starts ->
qmax, quantizer, int8 scan of every code, candidate replay).  The queries of a batch are launched
as L2-sharing siblings: the codes cross the HBM interface about once per launch, not once per query.
With --gpus N the SAME 1B-code list is sharded over the N ranks in contiguous ranges ("strong"
scaling; keys are 32-bit as in the reference, so the list cannot grow past 2^32 anyway) and the
per-shard push streams are gathered once per batch over RCCL and replayed (pyqadc/sharded.py).

    python bench.py --gpus 1 --steps 20 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  Environment overrides for quick runs: QADC_BENCH_CODES, QADC_BENCH_NQ,
QADC_BENCH_M, QADC_BENCH_CPU_SECONDS (0 disables the CPU leg).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "quick-adc_amd"))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)


def make_tables(rng, codebooks, nq):
    """Float distance tables ||q_m - c||^2 of N(0,1) queries against N(0,1) codebooks [M][16][d]."""
    M, _, d = codebooks.shape
    q = rng.normal(size=(nq, M, 1, d)).astype(np.float32)
    t = ((q - codebooks[None]) ** 2).sum(-1, dtype=np.float32)
    return np.ascontiguousarray(t.reshape(nq, 1, M * 16), np.float32)


def cpu_baseline(M, n_total, seed, qtables, R, seconds):
    """Times the reference's own scan_avx_4<M> (oracle/_ref, built from /root/reference) — or, if
    that build is absent, the oracle's scalar C port — on a bounded prefix of the same synthetic
    list with the same int8 tables, one thread.  Reported, never the target."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle as po
    cs = M // 2
    n = int(min(n_total, 32 * 1024 * 1024))
    codes = po.fill_codes(0, (n * cs + 7) // 8, seed)[:n * cs].reshape(n, cs)
    kind = "reference" if po.have_ref() else "port"
    if kind == "reference":
        inter = po.ref_interleave(codes)
        run = lambda qt: po.ref_scan_interleaved(M, [inter], [n], None, qt, R)
    else:
        run = lambda qt: po.scan_i8(M, [codes], None, qt, R)
    run(qtables[0])  # warm
    t0, nqueries = time.perf_counter(), 0
    while time.perf_counter() - t0 < seconds:
        run(qtables[nqueries % len(qtables)])
        nqueries += 1
    dt = time.perf_counter() - t0
    return {"value": n * nqueries / dt, "unit": "codes/s", "cores": 1, "kind": kind,
            "sample": "%d queries x first %d codes of the same synthetic list, same int8 tables, R=%d, "
                      "1 thread, %s" % (nqueries, n, R, "scan_avx_4<%d> compiled from the reference" % M
                                        if kind == "reference" else "scalar C port (oracle)")}


def cpu_extra_legs(M, n_total, seed, qtables, R, seconds):
    """Two more CPU figures asked for by BASELINE.md §3 (reported, never the target):
    - the reference's AVX2 scan on ALL host cores (one query per thread: ctypes releases the GIL);
    - BASELINE config 1, PQ 8x8 float ADC over 1M codes (scanner_simple / scan_standard<uint8_t,8>), 1 thread,
      timed with the oracle's C port."""
    from concurrent.futures import ThreadPoolExecutor
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle as po
    out = {}
    if po.have_ref():
        cs = M // 2
        n = int(min(n_total, 32 * 1024 * 1024))
        codes = po.fill_codes(0, (n * cs + 7) // 8, seed)[:n * cs].reshape(n, cs)
        inter = po.ref_interleave(codes)
        cores = os.cpu_count() or 1
        run = lambda i: po.ref_scan_interleaved(M, [inter], [n], None, qtables[i % len(qtables)], R)
        with ThreadPoolExecutor(cores) as ex:
            list(ex.map(run, range(cores)))                       # warm
            t0, done = time.perf_counter(), 0
            while time.perf_counter() - t0 < seconds:
                list(ex.map(run, range(4 * cores)))
                done += 4 * cores
            dt = time.perf_counter() - t0
        out["cpu_baseline_all_cores"] = {"value": n * done / dt, "unit": "codes/s", "cores": cores, "kind": "reference",
                                         "sample": "%d queries x first %d codes, one query per thread" % (done, n)}
    rng = np.random.default_rng(5)
    codes8 = rng.integers(0, 256, (1000000, 8), dtype=np.uint8)
    tables8 = rng.random((1, 8, 256)).astype(np.float32)
    po.scan_standard_u8(8, [codes8], None, tables8, R)
    t0, done = time.perf_counter(), 0
    while time.perf_counter() - t0 < min(seconds, 3.0):
        po.scan_standard_u8(8, [codes8], None, tables8, R)
        done += 1
    dt = time.perf_counter() - t0
    out["cpu_config1_pq8x8_float_adc"] = {"value": 1000000 * done / dt, "unit": "codes/s", "cores": 1, "kind": "port",
                                          "sample": "%d queries x 1M codes, scan_standard<uint8_t,8> C port" % done}
    return out


def real_encode_recall(M, R, keep, n, nq, local_rank):
    """Recall@R on REAL encodings (SURVEY.md §8d "real-encode variant"): clustered synthetic 128-d vectors,
    codebooks = sampled sub-vectors, PQ-encoded on the GPU (qadc_pq_encode), queried through the device-side
    feeders (qadc_search); ground truth = exact float L2 nearest neighbour (torch, chunked).  Untimed."""
    import torch
    import pyqadc
    dev = torch.device("cuda", local_rank)
    g = torch.Generator(device=dev).manual_seed(7)
    dim, ds, cs = 128, 128 // M, M // 2
    C = max(2000, n // 300)     # ~300 points per cluster: neither trivial nor hopeless for a 100-entry shortlist
    centres = torch.randn(C, dim, device=dev, generator=g) * 3
    base = centres[torch.randint(0, C, (n,), device=dev, generator=g)] + torch.randn(n, dim, device=dev, generator=g)
    queries = centres[torch.randint(0, C, (nq,), device=dev, generator=g)] + torch.randn(nq, dim, device=dev, generator=g)
    ids = torch.randint(0, n, (M, 16), device=dev, generator=g)
    cb = torch.stack([base[ids[m], m * ds:(m + 1) * ds] for m in range(M)]).contiguous()      # [M][16][ds]
    raw = torch.zeros(n * cs + 64, dtype=torch.uint8, device=dev)                             # padded tail
    torch.cuda.synchronize()
    cb_host = cb.cpu().numpy()
    pyqadc.pq_encode_device(cb_host, base.data_ptr(), n, dim, raw.data_ptr(), local_rank)
    best_d = torch.full((nq,), float("inf"), device=dev)
    best_i = torch.zeros(nq, dtype=torch.long, device=dev)
    for lo in range(0, n, 1 << 20):
        d = torch.cdist(queries, base[lo:lo + (1 << 20)])
        dmin, imin = d.min(dim=1)
        upd = dmin < best_d
        best_d = torch.where(upd, dmin, best_d)
        best_i = torch.where(upd, imin + lo, best_i)
    gt = best_i.cpu().numpy()
    torch.cuda.synchronize()
    idx = pyqadc.Index(M, local_rank)
    idx.add_partition_device(raw.data_ptr(), n, keepalive=raw)
    idx.finalize(keep)
    idx.set_pq(cb_host)
    res = idx.search(queries.cpu().numpy(), 1, R)
    hits = sum(int(gt[q] in set(res["keys"][q][:res["sizes"][q]].tolist())) for q in range(nq))
    idx.close()
    return {"value": hits / nq, "codes": n, "queries": nq,
            "data": "synthetic 128-d vectors, %d clusters (3*N(0,1) centres + N(0,1)), codebooks = sampled sub-vectors, PQ %dx4 "
                    "encoded on the GPU; ground truth = exact float L2 NN" % (C, M)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    M = int(os.environ.get("QADC_BENCH_M", 16))
    N = int(float(os.environ.get("QADC_BENCH_CODES", 1e9)))
    NQ = int(os.environ.get("QADC_BENCH_NQ", 32))
    R, KEEP, SEED = 100, 0.01, 0x5EED0001
    cs = M // 2

    import torch
    import torch.distributed as dist
    import pyqadc
    from pyqadc import sharded

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the Quick-ADC engine has no CPU path")
    # test hooks (not used by the driver): run the multi-rank path on a 1-GPU box over gloo
    backend = os.environ.get("QADC_BENCH_BACKEND", "nccl")
    if os.environ.get("QADC_BENCH_ONE_GPU"):
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    cdev = dev if backend == "nccl" else torch.device("cpu")   # device of the collective buffers
    # QADC_BENCH_FORCE_DIST=1 takes the multi-rank code path (collectives included) even with one rank
    use_dist = world > 1 or bool(os.environ.get("QADC_BENCH_FORCE_DIST"))
    if use_dist:
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    # ---- database: this rank's contiguous shard of the synthetic list + replica of the starts ----
    first, local_n = sharded.shard_ranges(N, world)[rank]
    starts = max(1, int(np.float32(N) * np.float32(KEEP)))
    idx = pyqadc.Index(M, local_rank)
    idx.add_partition_synthetic_shard(N, first, local_n, SEED, starts)
    idx.finalize(KEEP)
    idx.set_option("profile", 1)
    for kv in filter(None, os.environ.get("QADC_BENCH_OPTS", "").split(",")):     # tuning experiments only
        idx.set_option(kv.split("=")[0], float(kv.split("=")[1]))

    rng = np.random.default_rng(1234)
    codebooks = rng.normal(size=(M, 16, 128 // M)).astype(np.float32)
    pool = [make_tables(rng, codebooks, NQ) for _ in range(4)]   # 4 distinct query batches, reused cyclically
    assign = np.zeros((NQ, 1), np.int32)

    def run_steps(k):
        """k pipelined steps, three batches in flight: while batch s scans, batch s-1 is collected and replayed on the
        host and the front of batch s+1 (upload, pre-scan, quantizer) already runs on its own stream."""
        if use_dist:
            return run_steps_dist(k)
        last = None
        pending = []
        for s in range(k):
            idx.submit(s % 3, assign, pool[s % len(pool)].copy(), R)
            pending.append(s % 3)
            if len(pending) == 3:
                last = idx.collect(pending.pop(0))
        while pending:
            last = idx.collect(pending.pop(0))
        return last

    def run_steps_dist(k):
        """Multi-rank steps.  Every rank pre-scans 1/world of the starts (the rest of the path is sharded by codes, the
        pre-scan by starts); ONE all-gather per step carries the finished batch's candidate streams and the pre-scan
        values of the batch three steps ahead.  Per iteration i: enqueue the sliced pre-scan of batch i+4, collect batch i,
        gather [streams of i | pre-scan values of i+3], replay, submit batch i+3 — batches i+1 and i+2 keep the GPU
        busy meanwhile (host jitter of a whole step is absorbed), and a pre-scan has a whole extra batch of lead (its kernels only find room at the boundaries
        of the long scan launches)."""
        last = None
        if k <= 0:
            return last
        tbs = {}
        LEAD = 3                                               # batches in flight besides the one being collected

        def prescan(b):                                        # batch b's sliced pre-scan -> pre-slot b % 2
            tbs[b % 6] = pool[b % len(pool)].copy()
            idx.prescan_submit(b % 2, assign, tbs[b % 6], R, rank, world)

        nb = min(LEAD, k)                                      # the first batches: one stand-alone gather for all
        pvs = []
        for b in range(nb):
            prescan(b)
            pvs.append(idx.prescan_collect(b % 2))
        g = sharded.gather_prescan(np.concatenate(pvs), cdev)
        for b in range(nb):
            idx.submit(b % 4, assign, tbs[b % 6], R, prescan=g[b * NQ:(b + 1) * NQ])
        if k > LEAD:
            prescan(LEAD)
        for i in range(k):                                     # batches i .. i+LEAD-1 are in flight; i is collected now
            if i + LEAD + 1 < k:
                prescan(i + LEAD + 1)                          # pre-slot of batch i+LEAD-1, collected an iteration ago
            res = idx.collect_candidates(i % 4)
            pv = idx.prescan_collect((i + LEAD) % 2) if i + LEAD < k else None   # batch i+LEAD's, enqueued an iteration ago
            out = sharded.merge_batch(res, NQ, R, res["status"], cdev, extra=pv)
            last = out[:3]
            if i + LEAD < k:
                idx.submit((i + LEAD) % 4, assign, tbs[(i + LEAD) % 6], R, prescan=out[3])
        return last

    def sync():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    run_steps(args.warmup)
    idx.profile_reset()
    sync()
    t0 = time.perf_counter()
    last = run_steps(args.steps)
    sync()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    prof = idx.profile()

    # ---- Recall@100 against the exact float-ADC nearest code, last batch, untimed ----
    tb = pool[(args.steps - 1) % len(pool)]
    keys = last[0] if use_dist else last["keys"]
    hits = 0
    for q in range(NQ):
        key, _, dist_q = idx.float_top1(0, tb[q, 0])
        if use_dist:
            cand = torch.tensor([dist_q, float(key)], dtype=torch.float64, device=cdev)
            allc = torch.empty(2 * world, dtype=torch.float64, device=cdev)
            dist.all_gather_into_tensor(allc, cand)
            allc = allc.cpu().numpy().reshape(world, 2)
            key = int(allc[np.lexsort((allc[:, 1], allc[:, 0]))[0], 1])   # min distance, lowest key on ties
        hits += int(key in set(keys[q].tolist()))
    recall = hits / NQ

    if rank == 0:
        total_codes = float(N) * NQ * args.steps
        scan_ms = prof["scan_ms"]
        achieved = prof["scan_codes"] * cs / (scan_ms * 1e-3) / 1e9 if scan_ms > 0 else 0.0
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "r01_hbm_traffic.json")
        if world == 1 and os.path.exists(pmc):
            pj = json.load(open(pmc))
        launches = max(prof["scan_launches"], 1)
        avg_ms = scan_ms / launches
        alg_bytes = prof["scan_codes"] * cs / launches
        if world == 1 and os.path.exists(pmc):
            # HBM bytes per launch = the PMC ratio (FETCH_SIZE*2 + WRITE_SIZE over algorithmic bytes, longest launch of
            # the profiled run of this same workload) x the algorithmic bytes of this run's average timed launch
            if pj.get("queries_per_step") == NQ and pj.get("codes") == N and pj.get("M") == M:
                traffic = pj.get("traffic_over_algorithmic") * alg_bytes
        mq = prof["mq_launches"] > 0
        # LDS-array cycles the launches need (MI355X_MICROARCH.md, LDS): multi-query kernel = one ds_read_b128 (4 cycles
        # per 64 lanes) per code nibble and pass; single-query kernel = one ds_read_u8 (2 cycles) per code byte and query
        lds_cycles = prof["pass_codes"] * (M * 4 if mq else (M // 2) * 2) / 64.0
        lds_frac = lds_cycles / (256 * 2.4e9 * scan_ms * 1e-3) if scan_ms > 0 else 0.0
        out = {
            "metric": "pq_codes_scanned_per_sec", "value": total_codes / elapsed, "unit": "codes/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed * 1e3 / args.steps, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "int8", "data": "synthetic",
            "config": {"workload": "flat DB, %d x %dx4 PQ codes (%d B/code), R=%d, keep=%.2f%%, %d queries/step, "
                                   "every query scans the whole list, sharded over %d GPU(s)" % (N, M, cs, R, KEEP * 100, NQ, world),
                       "codes": N, "M": M, "R": R, "keep": KEEP, "queries_per_step": NQ,
                       "parallelism": "shard%d" % world},
            "recall_at_100": recall,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": ("scan_i8_mq_kernel<%d,2> (8 queries per pass, %d passes per launch as L2-sharing siblings)"
                                    % (M, (NQ + 7) // 8)) if mq else
                                   ("scan_i8_kernel<%d,2> (sibling-major launch, %d queries share each tile)" % (M, NQ)
                                    if NQ > 1 else "scan_i8_kernel<%d,2,nt,chunk>" % M),
                         "launches": prof["scan_launches"], "avg_launch_ms": avg_ms,
                         "algorithmic_bytes_per_launch": alg_bytes,
                         # what actually crossed the HBM interface (PMC pass, profiles/): below the algorithmic
                         # bytes because the queries of a launch share tiles in L2 -- which is why frac can exceed 1
                         "hbm_actual": None if traffic is None else
                         {"achieved": traffic / (avg_ms * 1e-3) / 1e9, "unit": "GB/s",
                          "frac": traffic / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                          "traffic_over_algorithmic": traffic / alg_bytes},
                         # the limiter once a pass serves several queries: LDS-array cycles the lookups need over the
                         # LDS cycles available (256 CUs x 2.4 GHz x duration); the VALU pipe is equally loaded
                         "note": "achieved = algorithmic bytes (M/2 B per code and query) / launch time; the queries of a "
                                 "launch share the codes through L2 (8 per pass, passes as siblings), so the HBM interface "
                                 "moves far fewer bytes (hbm_actual) and frac exceeds 1; the launch is bound by LDS bandwidth (lds)",
                         "lds": {"achieved": lds_cycles / (scan_ms * 1e-3) / 1e9 if scan_ms > 0 else 0.0,
                                 "peak": 256 * 2.4, "unit": "G LDS cycles/s", "frac": lds_frac,
                                 "code_reads_per_launch": prof["pass_codes"] / launches}},
            "phases": {"prescan_quantize_ms_per_step": prof["start_ms"] / args.steps,
                       "scan_kernel_ms_per_step": scan_ms / args.steps,
                       "host_sort_replay_ms_per_step": prof["host_replay_ms"] / args.steps,
                       "candidates_per_query": prof["candidates"] / (NQ * args.steps), "regrows": prof["regrows"]},
        }
        n_real = int(float(os.environ.get("QADC_BENCH_REAL_CODES", 1e7)))
        if world == 1 and n_real > 0:
            out["recall_at_100_real_encode"] = real_encode_recall(M, R, KEEP, n_real, 64, local_rank)
        cpu_s = float(os.environ.get("QADC_BENCH_CPU_SECONDS", 15))
        if world == 1 and cpu_s > 0:
            res = idx.query_scan(assign, pool[0].copy(), R, want_qtables=True)
            out["cpu_baseline"] = cpu_baseline(M, N, SEED, res["qtables"][:, 0], R, cpu_s)
            out.update(cpu_extra_legs(M, N, SEED, res["qtables"][:, 0], R, min(cpu_s, 6.0)))
        print(json.dumps(out), flush=True)
    idx.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
